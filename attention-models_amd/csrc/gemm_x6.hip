// C (M, N) = A (M, K) B (N, K)^T (+ bias[N]) for f32 operands with SPLIT-BF16 products ("bf16x6").
//
// The projection / FFN GEMMs of the ViT-VQGAN step (x W^T of every nn.Linear, and dY W for its input
// gradient with W pre-transposed) are 36 % of the step on the vendor library's exact-f32 path, which
// already runs at 0.79 of the f32 MFMA peak: an exact-f32 kernel of our own cannot win there (SURVEY.md
// section 8f rank 1).  What can: every f32 operand x is split into three bf16 parts x = h + m + l
// (24 mantissa bits) and a product is the sum of six exact partial products accumulated in f32 by
// v_mfma_f32_32x32x16_bf16 -- the scheme of attn_fwd_x6.hip, with the same measured f32-level error
// (tools/ubench_bf16x6.hip) -- at 6 / 16 of the f32 MFMA's matrix-pipe time.  Its roofline is the bf16
// MFMA peak / 6 (417 TFLOP/s), NOT the 157.3 TFLOP/s f32 peak; bench.py reports it as a variant.
//
// Workgroup: 128 x 128 output tile, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles; K in steps of 32.
// Operands travel global -> registers (f32, buffer loads: rows past M / N read as zeros) -> split ->
// LDS as three bf16 planes per operand ([row][32 k], 80-byte rows: conflict-free 16-byte fragment
// reads) -> MFMA fragments.  The loads of step k+1 are in flight under the 48 MFMAs of step k.
#include "amk_common.h"

namespace amk_gemm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int RS = BK + 8;            // bf16 per LDS row: 80 bytes
constexpr int PLANE = 128 * RS;       // one plane of a 128-row tile

__device__ __forceinline__ f32x16 mfmab(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split4(const float4& x, bf16x4 (&pl)[3]) {
  const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)v[j];
    const float r1 = v[j] - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    pl[0][j] = h; pl[1][j] = m; pl[2][j] = l;
  }
}

struct Args {
  const float *A, *B, *bias;
  float* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
};

__global__ __launch_bounds__(256, 2) void gemm_x6_nt_kernel(Args g) {
  __shared__ __attribute__((aligned(16))) __bf16 Ap[3 * PLANE];
  __shared__ __attribute__((aligned(16))) __bf16 Bp[3 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // tiles: consecutive logical ids walk the N tiles of one M tile (they share the A rows): one XCD
  const int ntn = (g.N + BN - 1) / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;

  // staging: thread -> rows sr + 32 j (j < 4), 4 floats at column sc
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)(((int64_t)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)(((int64_t)(g.N - 1) * g.ldb + g.K) * 4), 0x00020000);
  int aoff[4], boff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ra = m0 + sr + 32 * j, rb = n0 + sr + 32 * j;
    aoff[j] = ra < g.M ? (int)(((int64_t)ra * g.lda + sc) * 4) : 0x7ffffff0;  // past the buffer: zeros
    boff[j] = rb < g.N ? (int)(((int64_t)rb * g.ldb + sc) * 4) : 0x7ffffff0;
  }
  float4 ast[4], bst[4];
  auto prefetch = [&](int k0) {
    // columns past K (the last, partial step) must read zeros, not the next row: out-of-range offset
    const bool kin = k0 + sc < g.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ast[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, kin ? aoff[j] + k0 * 4 : 0x7ffffff0, 0, 0));
      bst[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, kin ? boff[j] + k0 * 4 : 0x7ffffff0, 0, 0));
    }
  };
  auto commit = [&]() {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 pa[3], pb[3];
      split4(ast[j], pa);
      split4(bst[j], pb);
      const int o = (sr + 32 * j) * RS + sc;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        *reinterpret_cast<bf16x4*>(&Ap[p * PLANE + o]) = pa[p];
        *reinterpret_cast<bf16x4*>(&Bp[p * PLANE + o]) = pb[p];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

  const int nk = (g.K + BK - 1) / BK;
  prefetch(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // every wave is done with the previous tile's fragments
    commit();
    __syncthreads();
    prefetch(min(kt + 1, nk - 1) * BK);  // unconditional (the last step re-reads its own tile, unused)
    __builtin_amdgcn_sched_barrier(0);   // issued HERE, under the MFMAs below (the compiler sinks them otherwise)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = *reinterpret_cast<const bf16x8*>(&Ap[p * PLANE + (64 * wm + 32 * i + r) * RS + 16 * s + 8 * h]);
          b[i][p] = *reinterpret_cast<const bf16x8*>(&Bp[p * PLANE + (64 * wn + 32 * i + r) * RS + 16 * s + 8 * h]);
        }
      // six partial products per output tile, smallest first; the four tiles' chains interleave
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
        constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfmab(a[i][PA[q]], b[j][PB[q]], acc[i][j]);
      }
    }
  }
  // epilogue: register t of lane (r, h) is C[m0 + 64 wm + 32 i + acc_row(t, h)][n0 + 64 wn + 32 j + r]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 64 * wn + 32 * j + r;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = m0 + 64 * wm + 32 * i + acc_row(t, h);
        if (m < g.M) g.C[(int64_t)m * g.ldc + n] = acc[i][j][t] + bv;
      }
  }
}

}  // namespace amk_gemm

using namespace amk_gemm;

extern "C" int amk_gemm_x6_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                              float* C, int64_t ldc, int M, int N, int K, void* stream) {
  AMK_CHECK_ARG(A && B && C, "amk_gemm_x6_nt: null pointer");
  AMK_CHECK_ARG(M > 0 && N > 0 && K > 0, "amk_gemm_x6_nt: non-positive size M=%d N=%d K=%d", M, N, K);
  AMK_CHECK_SUPPORTED(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                          ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0,
                      "amk_gemm_x6_nt: K, lda, ldb must be multiples of 4 and A, B 16-byte aligned");
  AMK_CHECK_SUPPORTED(((int64_t)(M - 1) * lda + K) * 4 < 0x7ffffff0ll && ((int64_t)(N - 1) * ldb + K) * 4 < 0x7ffffff0ll,
                      "amk_gemm_x6_nt: an operand must span < 2 GiB");
  Args g;
  g.A = A; g.B = B; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  const int64_t nwg = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  AMK_CHECK_SUPPORTED(nwg < (1ll << 31), "amk_gemm_x6_nt: grid too large");
  hipLaunchKernelGGL(gemm_x6_nt_kernel, dim3((unsigned)nwg), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  AMK_CHECK_LAUNCH("amk_gemm_x6_nt");
  return AMK_OK;
}
