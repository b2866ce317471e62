// Micro-benchmark: does f32 VALU work overlap with v_mfma_f32_32x32x2_f32 on one SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o tools/ubench_mfma_valu && ./tools/ubench_mfma_valu
// Each wave runs ITERS x { 4 MFMAs (independent accumulators) + NV v_fma_f32 on private registers }.
// Reported: cycles per iteration per wave-slot for NV = 0, 4, 8, 16, 32 at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

// same loop with NV/2 v_pk_fma_f32 (two results each) instead of NV v_fma_f32
template <int NV>
__global__ __launch_bounds__(256) void kpk(float* out, int iters) {
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 1.f; a2[i] = 2.f; a3[i] = 3.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  f32x2 v[16];
  const f32x2 y2 = {y, y}, x2 = {x, x * 0.5f};
  for (int i = 0; i < 16; ++i) v[i] = (f32x2){x + i, x - i};
  for (int it = 0; it < iters; ++it) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) v[j] = __builtin_elementwise_fma(v[j], y2, x2);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV / 8, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 1.f; a2[i] = 2.f; a3[i] = 3.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = x + i;
  for (int it = 0; it < iters; ++it) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
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
void runpk(float* d, int blocks_per_cu, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(kpk<NV>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kpk<NV>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("PK  NV=%2d results (%2d v_pk_fma) waves/SIMD=%d : %.3f ms  ns/iter/wave-slot %.1f\n", NV, NV / 2, blocks_per_cu, ms, ms * 1e6 / iters);
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
  const double flops = (double)grid * 4 * iters * 4 * 4096.0;
  printf("NV=%2d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s (MFMA)  ns/iter/wave-slot %.1f\n", NV, blocks_per_cu, ms, flops / ms / 1e9,
         ms * 1e6 / iters);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  const int iters = 20000;
  for (int w = 1; w <= 2; ++w) {
    run<0>(d, w, iters); run<4>(d, w, iters); run<8>(d, w, iters); run<16>(d, w, iters); run<32>(d, w, iters);
    runpk<16>(d, w, iters); runpk<32>(d, w, iters);
  }
  return 0;
}
