// Micro-benchmark: v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 as the engine of an LDS-fed f32 tile loop.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_shapes.hip -o tools/ubench_mfma_shapes && ./tools/ubench_mfma_shapes
// Each wave keeps a 64x64 accumulator tile (64 registers either way) and runs, per iteration, one 8-deep step of it:
//   32x32x2 : 2x2 blocks x 4 k-pairs  = 16 MFMA of 4096 FLOP, operands 2+2 registers per k-pair
//   16x16x4 : 4x4 blocks x 2 k-quads  = 32 MFMA of 2048 FLOP, operands 4+4 registers per k-quad
// with the operand fragments re-read from LDS every iteration (NL = ds_read_b128 per iteration: 4 for both shapes
// covers the step) plus NX extra ds_read_b128 and NV v_add_f32 as stand-ins for staging traffic and address math.
// Reported: TFLOP/s at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NX, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[4096 + 1024];
  for (int i = threadIdx.x; i < 5120; i += 256) lds[i] = (float)(i & 15) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4* lp = reinterpret_cast<const float4*>(lds) + lane;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = lane + i;
  float acc_sum = 0.f;
  if (SHAPE == 32) {
    f32x16 c[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const float4 a0 = lp[(it & 3) * 64], a1 = lp[256 + (it & 3) * 64], b0 = lp[512 + (it & 3) * 64], b1 = lp[768 + (it & 3) * 64];
      float4 x[NX > 0 ? NX : 1];
#pragma unroll
      for (int j = 0; j < NX; ++j) x[j] = lp[1024 + ((it + j) & 3) * 64];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 15] += 1.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float fa0 = e == 0 ? a0.x : e == 1 ? a0.y : e == 2 ? a0.z : a0.w, fa1 = e == 0 ? a1.x : e == 1 ? a1.y : e == 2 ? a1.z : a1.w;
        const float fb0 = e == 0 ? b0.x : e == 1 ? b0.y : e == 2 ? b0.z : b0.w, fb1 = e == 0 ? b1.x : e == 1 ? b1.y : e == 2 ? b1.z : b1.w;
        c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, c[0][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, c[0][1], 0, 0, 0);
        c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, c[1][0], 0, 0, 0);
        c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, c[1][1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NX; ++j) v[j & 15] += x[j].x;
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc_sum += c[i][j][r];
  } else {
    f32x4 c[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) c[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // one b128 per operand side and k-quad pair: register e of the float4 = block e's fragment for this k-quad
      const float4 a0 = lp[(it & 3) * 64], a1 = lp[256 + (it & 3) * 64], b0 = lp[512 + (it & 3) * 64], b1 = lp[768 + (it & 3) * 64];
      float4 x[NX > 0 ? NX : 1];
#pragma unroll
      for (int j = 0; j < NX; ++j) x[j] = lp[1024 + ((it + j) & 3) * 64];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 15] += 1.0f;
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
        const float4 a = kq ? a1 : a0, b = kq ? b1 : b0;
        const float fa[4] = {a.x, a.y, a.z, a.w}, fb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], c[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NX; ++j) v[j & 15] += x[j].x;
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc_sum += c[i][j][r];
  }
  for (int i = 0; i < 16; ++i) acc_sum += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc_sum;
}

template <int SHAPE, int NX, int NV>
void run(float* d, int wps) {
  const int iters = 20000, grid = 256 * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, NX, NV>), dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, NX, NV>), dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 65536.0;   // 64x64x8 x 2 per wave and iteration
  printf("shape %2dx%-2d  extra ds_read_b128 %2d  v_add %2d  waves/SIMD %d : %7.3f ms  %6.1f TFLOP/s\n", SHAPE, SHAPE, NX, NV, wps, ms, flops / ms / 1e9);
}

int main() {
  float* d; hipMalloc(&d, 512 * 256 * sizeof(float));
  for (int w = 1; w <= 2; ++w) {
    run<32, 0, 0>(d, w); run<16, 0, 0>(d, w);
    run<32, 4, 0>(d, w); run<16, 4, 0>(d, w);
    run<32, 8, 8>(d, w); run<16, 8, 8>(d, w);
    run<32, 0, 16>(d, w); run<16, 0, 16>(d, w);
  }
  return 0;
}
