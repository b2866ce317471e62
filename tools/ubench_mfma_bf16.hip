// Micro-benchmark: v_mfma_f32_32x32x16_bf16 rate on gfx950, and whether f32 VALU work overlaps with it
// (it does not with the f32 MFMA, tools/ubench_mfma_valu.hip).  Basis for judging a split-bf16
// (bf16x3 / bf16x6) emulation of the f32 contractions.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_bf16.hip -o tools/ubench_mfma_bf16 && ./tools/ubench_mfma_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 1.f; a2[i] = 2.f; a3[i] = 3.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  bf16x8 p, q;
  for (int i = 0; i < 8; ++i) { p[i] = (__bf16)(x + i); q[i] = (__bf16)(y - i); }
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = x + i;
  for (int it = 0; it < iters; ++it) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a3, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = __builtin_fmaf(v[j], y, x);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV / 4, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  for (int j = 0; j < 32; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
void run(float* d, int blocks_per_cu, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 4 * (32.0 * 32 * 16 * 2);
  printf("bf16 32x32x16  NV=%2d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s (MFMA)  ns/iter/wave-slot %.1f\n", NV, blocks_per_cu, ms,
         flops / ms / 1e9, ms * 1e6 / iters);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  const int iters = 20000;
  for (int w = 1; w <= 2; ++w) {
    run<0>(d, w, iters); run<4>(d, w, iters); run<8>(d, w, iters); run<16>(d, w, iters); run<32>(d, w, iters);
  }
  return 0;
}
