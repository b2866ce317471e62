// Micro-benchmark: do two waves on ONE SIMD overlap when one issues only v_mfma_f32_32x32x16_bf16 and the other only
// vector instructions (v_exp_f32 / v_fma_f32)?  A 512-thread workgroup puts waves w and w + 4 on the same SIMD; waves 0-3
// run role A, waves 4-7 role B.  Times: A alone (B exits), B alone, both.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_partner_overlap.hip -o /tmp/upo && /tmp/upo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bits: 1 = MFMA waves active, 2 = VALU waves active; SAME: both roles in every wave, interleaved (4 MFMA + NV VALU)
template <int NEXP, int NFMA>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 1.f; a2[i] = 2.f; a3[i] = 3.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  bf16x8 p, q;
  for (int i = 0; i < 8; ++i) { p[i] = (__bf16)(x + i); q[i] = (__bf16)(y - i); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = x + i;
  if (roleA) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a3, 0, 0, 0);
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NEXP; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
#pragma unroll
        for (int j = 0; j < NFMA; ++j) v[(j + NEXP) & 15] = __builtin_fmaf(v[(j + NEXP) & 15], y, x);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NEXP, int NFMA>
float run(float* d, int iters, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NEXP, NFMA>), dim3(256), dim3(512), 0, 0, d, iters, mode);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NEXP, NFMA>), dim3(256), dim3(512), 0, 0, d, iters, mode);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int NEXP, int NFMA>
void report(float* d) {
  const int iters = 20000;
  const float a = run<NEXP, NFMA>(d, iters, 1), b = run<NEXP, NFMA>(d, iters, 2), ab = run<NEXP, NFMA>(d, iters, 3);
  printf("per iteration: wave A 4 MFMA, wave B %2d v_exp + %2d v_fma:  A alone %.1f ns  B alone %.1f ns  both %.1f ns  (sum %.1f, max %.1f)\n", NEXP, NFMA,
         a * 1e6 / iters, b * 1e6 / iters, ab * 1e6 / iters, (a + b) * 1e6 / iters, (a > b ? a : b) * 1e6 / iters);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
  report<0, 16>(d); report<0, 32>(d); report<8, 0>(d); report<16, 0>(d); report<8, 16>(d); report<16, 16>(d);
  return 0;
}
