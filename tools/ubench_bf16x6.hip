// Prototype measurement for a split-bf16 ("bf16x6") emulation of the f32 contractions:
// C(32x32 per wave) = A(32x64) B^T(32x64) with every f32 operand split into three bf16 parts
// (hi, mid, lo) and the six products hh, hm, mh, hl, lh, mm accumulated in f32 by
// v_mfma_f32_32x32x16_bf16 -- against the exact-f32 v_mfma_f32_32x32x2_f32 the product uses today.
// Reports (1) the error of both against a double-precision reference, (2) MFMA-only throughput of both
// (operands already split and resident in registers: the upper bound of what the split can buy).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_bf16x6.hip -o tools/ubench_bf16x6 && ./tools/ubench_bf16x6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// one wave: C[32][32] = A[32][64] * B[32][64]^T, both row-major f32 in global memory
__global__ void k_bf16x6(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x & 63, r = lane & 31, kb = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < 64; k0 += 16) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {
      __bf16 h, m, l;
      split3(A[r * 64 + k0 + 8 * kb + j], h, m, l); ah[j] = h; am[j] = m; al[j] = l;
      split3(B[r * 64 + k0 + 8 * kb + j], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
    }
    // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * kb) * 32 + r] = acc[i];
}

__global__ void k_f32(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x & 63, r = lane & 31, kb = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k = 0; k < 64; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 64 + k + kb], B[r * 64 + k + kb], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * kb) * 32 + r] = acc[i];
}

// throughput: ITERS x one 32x32x64 tile per wave, operands in registers
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 1.f; }
  const float x = threadIdx.x * 1e-3f, y = 1.0001f;
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 8; ++j) { a[p][j] = (__bf16)(x + j + p); b[p][j] = (__bf16)(y - j - p); }
  for (int it = 0; it < iters; ++it) {
    if (SPLIT) {
#pragma unroll
      for (int k0 = 0; k0 < 4; ++k0) {   // 4 x K16 = K 64, six products each, two tiles interleaved
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc1, 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<float> A(32 * 64), B(32 * 64), C0(32 * 32), C1(32 * 32);
  srand(1);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : B) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *dA, *dB, *dC;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C0.size() * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_bf16x6, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  (void)hipMemcpy(C0.data(), dC, C0.size() * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  (void)hipMemcpy(C1.data(), dC, C1.size() * 4, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, mx = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double ref = 0;
      for (int k = 0; k < 64; ++k) ref += (double)A[m * 64 + k] * (double)B[n * 64 + k];
      e0 = fmax(e0, fabs(C0[m * 32 + n] - ref));
      e1 = fmax(e1, fabs(C1[m * 32 + n] - ref));
      mx = fmax(mx, fabs(ref));
    }
  printf("K=64 dot products, |values| <= 1, max |C| %.3f: max abs error  bf16x6 %.3e   f32 MFMA %.3e\n", mx, e0, e1);

  float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  const int iters = 2000;
  for (int split = 0; split < 2; ++split) {
    for (int w = 1; w <= 2; ++w) {
      hipEvent_t t0, t1;
      (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
      const int grid = 256 * w;
      if (split) hipLaunchKernelGGL(k_rate<true>, dim3(grid), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k_rate<false>, dim3(grid), dim3(256), 0, 0, d, iters);
      (void)hipEventRecord(t0);
      if (split) hipLaunchKernelGGL(k_rate<true>, dim3(grid), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k_rate<false>, dim3(grid), dim3(256), 0, 0, d, iters);
      (void)hipEventRecord(t1);
      (void)hipEventSynchronize(t1);
      float ms; (void)hipEventElapsedTime(&ms, t0, t1);
      const double tiles = (double)grid * 4 * iters * 2;            // 32x32x64 tiles
      const double flop = tiles * 2.0 * 32 * 32 * 64;               // algorithmic (f32-equivalent) FLOPs
      printf("%s  waves/SIMD %d: %.3f ms, %.1f algorithmic TFLOP/s\n", split ? "bf16x6 (6 x 32x32x16 bf16 per K16)" : "f32    (32 x 32x32x2 f32 per K64) ",
             w, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
