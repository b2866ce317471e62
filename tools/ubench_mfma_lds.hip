// Micro-benchmark: cost of feeding v_mfma_f32_32x32x2_f32 operands from LDS.
//   MODE 0: operands in registers (baseline)
//   MODE 1: A operand of every MFMA from its own ds_read_b32 (the attention PV pattern)
//   MODE 2: A operands of 4 MFMAs from one ds_read_b128 (the attention QK^T pattern)
//   MODE 3: as 1 but two MFMAs share each ds_read_b32 (two query blocks per wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 68];
  for (int i = threadIdx.x; i < 64 * 68; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 1.f; a2[i] = 2.f; a3[i] = 3.f; }
  float y = 1.0001f + lane * 1e-6f, x = lane * 1e-3f;
  const float* row = &lds[(lane & 31) * 68 + 32 * (lane >> 5)];
  const float* col = &lds[(lane >> 5) * 4 * 68 + (lane & 31)];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(col[(4 * j + 0) * 68], y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(col[(4 * j + 1) * 68 + 32], y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(col[(4 * j + 2) * 68], y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(col[(4 * j + 3) * 68 + 32], y, a3, 0, 0, 0);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(row + 4 * j);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.y, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.z, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w, y, a3, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = col[j * 68 + 32 * (j & 1)];
        if (j & 1) { a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, x, a3, 0, 0, 0); }
        else { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, x, a1, 0, 0, 0); }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float* d, int bpc, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * bpc;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 16 * 4096.0;
  printf("MODE=%d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s  ns per MFMA per wave-slot %.2f\n", MODE, bpc, ms, flops / ms / 1e9,
         ms * 1e6 / iters / 16);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  const int iters = 5000;
  for (int w = 1; w <= 2; ++w) { run<0>(d, w, iters); run<1>(d, w, iters); run<2>(d, w, iters); run<3>(d, w, iters); }
  return 0;
}
