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
// B is the WEIGHT: small and shared by every row tile, so it is split ONCE per call into three bf16
// planes (split_planes_kernel: [plane][row][K rounded up to 32], zero padded) and the GEMM streams those;
// A (the activations, read once) is split on the fly between its global load and its LDS store.
//
// Workgroup: 128 x 128 output tile, 8 waves as 2 (m) x 4 (n), each 64 x 32 = two MFMA tiles; K in steps
// of 32; two workgroups per CU (four waves per SIMD: one wave's split / LDS traffic runs under another's
// MFMAs).  LDS: three bf16 planes per operand, [row][32 k] with 80-byte rows (conflict-free 16-byte
// fragment reads), 60 KB.  The loads of step k+1 are in flight under the MFMAs of step k.
// Measured at the ViT-VQGAN shapes (tools/kbench_gemm.py): 125-168 TFLOP/s = 0.30-0.40 of the bf16x6 bound,
// 1.15-1.4 x the tuned exact-f32 library GEMM.  Three other structures were built and measured within 8 % of
// it: 4 waves x (64 x 64) with both operands split on the fly; a 256 x 128 tile with the split hoisted out of
// the barrier window; two LDS stages with one barrier per step.  All sit at one barrier per ~768 matrix cycles
// per wave -- the level of the guide's two-barrier bf16 structure (912 of 2500 TFLOP/s); the next step is its
// 256 x 256 eight-phase pipeline, which needs N >= 512 to fill the chip at M = 32768.
#include "amk_common.h"

namespace amk_gemm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, NTHR = 512;
constexpr int RS = BK + 8;            // bf16 per LDS row: 80 bytes
constexpr int PLANE = 128 * RS;       // one plane of a 128-row tile

__device__ __forceinline__ f32x16 mfmab(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split1(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

__device__ __forceinline__ void split4(const float4& x, bf16x4 (&pl)[3]) {
  const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 h, m, l;
    split1(v[j], h, m, l);
    pl[0][j] = h; pl[1][j] = m; pl[2][j] = l;
  }
}

// planes[p][r][kp] = part p of W[r][k] (k < K), 0 for K <= kp < Kp
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ W, int64_t ldw, int R, int K, int Kp,
                                                           __bf16* __restrict__ planes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per 4 consecutive k
  const int kq = Kp / 4;
  if (i >= (int64_t)R * kq) return;
  const int r = (int)(i / kq), k = (int)(i % kq) * 4;
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) x = *reinterpret_cast<const float4*>(W + (int64_t)r * ldw + k);  // K % 4 == 0
  bf16x4 pl[3];
  split4(x, pl);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4*>(planes + ((int64_t)p * R + r) * Kp + k) = pl[p];
}

struct Args {
  const float *A, *bias;
  const __bf16* Bp;   // [3][N][Kp]
  float* C;
  int M, N, K, Kp;
  int64_t lda, ldc;
};

__global__ __launch_bounds__(NTHR, 4) void gemm_x6_nt_kernel(Args g) {
  __shared__ __attribute__((aligned(16))) __bf16 Ap[3 * PLANE];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves: rows 64 wm .., columns 32 wn ..
  // tiles: consecutive logical ids walk the N tiles of one M tile (they share the A rows): one XCD
  const int ntn = (g.N + BN - 1) / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;

  // A staging: thread -> rows ar + 64 j (j < 2), 4 floats at column ac
  const int ar = tid >> 3, ac = (tid & 7) * 4;
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)(((int64_t)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
  int aoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ra = m0 + ar + 64 * j;
    aoff[j] = ra < g.M ? (int)(((int64_t)ra * g.lda + ac) * 4) : 0x7ffffff0;  // past the buffer: zeros
  }
  // B staging: 3 planes x 128 rows x 4 pieces of 16 B: plane p, row tid / 4, piece tid % 4
  const int br = tid >> 2, bc = (tid & 3) * 8;
  const __amdgpu_buffer_rsrc_t b_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.Bp, 0, (int)((int64_t)3 * g.N * g.Kp * 2), 0x00020000);
  int boff[3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
    boff[p] = n0 + br < g.N ? (int)((((int64_t)p * g.N + n0 + br) * g.Kp + bc) * 2) : 0x7ffffff0;

  float4 ast[2];
  uint4 bst[3];
  auto prefetch = [&](int k0) {
    const bool kin = k0 + ac < g.K;  // columns past K (the last, partial step) must read zeros, not the next row
#pragma unroll
    for (int j = 0; j < 2; ++j)
      ast[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, kin ? aoff[j] + k0 * 4 : 0x7ffffff0, 0, 0));
#pragma unroll
    for (int p = 0; p < 3; ++p)
      bst[p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, boff[p] + k0 * 2, 0, 0));  // Kp is padded
  };
  auto commit = [&]() {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x4 pa[3];
      split4(ast[j], pa);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4*>(&Ap[p * PLANE + (ar + 64 * j) * RS + ac]) = pa[p];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(&Bs[p * PLANE + br * RS + bc]) = bst[p];
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[i][t] = 0.f;

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
      bf16x8 a[2][3], b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        b[p] = *reinterpret_cast<const bf16x8*>(&Bs[p * PLANE + (32 * wn + r) * RS + 16 * s + 8 * h]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i][p] = *reinterpret_cast<const bf16x8*>(&Ap[p * PLANE + (64 * wm + 32 * i + r) * RS + 16 * s + 8 * h]);
      }
      // six partial products per output tile, smallest first; the two tiles' chains interleave
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
        constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = mfmab(a[i][PA[q]], b[PB[q]], acc[i]);
      }
    }
  }
  // epilogue: register t of lane (r, h) is C[m0 + 64 wm + 32 i + acc_row(t, h)][n0 + 32 wn + r]
  const int n = n0 + 32 * wn + r;
  if (n < g.N) {
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = m0 + 64 * wm + 32 * i + acc_row(t, h);
        if (m < g.M) g.C[(int64_t)m * g.ldc + n] = acc[i][t] + bv;
      }
  }
}

}  // namespace amk_gemm

using namespace amk_gemm;

static int padded_k(int K) { return (K + BK - 1) / BK * BK; }

extern "C" int64_t amk_gemm_x6_planes_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (int64_t)3 * N * padded_k(K) * 2;
}

extern "C" int amk_gemm_x6_split(const float* W, int64_t ldw, int N, int K, void* planes, void* stream) {
  AMK_CHECK_ARG(W && planes, "amk_gemm_x6_split: null pointer");
  AMK_CHECK_ARG(N > 0 && K > 0, "amk_gemm_x6_split: non-positive size");
  AMK_CHECK_SUPPORTED(K % 4 == 0 && ldw % 4 == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(planes)) & 15) == 0,
                      "amk_gemm_x6_split: K, ldw multiples of 4 and 16-byte aligned pointers");
  const int Kp = padded_k(K);
  const int64_t n = (int64_t)N * (Kp / 4);
  AMK_CHECK_SUPPORTED((n + 255) / 256 < (1ll << 31), "amk_gemm_x6_split: grid too large");
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     W, ldw, N, K, Kp, static_cast<__bf16*>(planes));
  AMK_CHECK_LAUNCH("amk_gemm_x6_split");
  return AMK_OK;
}

extern "C" int amk_gemm_x6_nt(const float* A, int64_t lda, const void* b_planes, const float* bias,
                              float* C, int64_t ldc, int M, int N, int K, void* stream) {
  AMK_CHECK_ARG(A && b_planes && C, "amk_gemm_x6_nt: null pointer");
  AMK_CHECK_ARG(M > 0 && N > 0 && K > 0, "amk_gemm_x6_nt: non-positive size M=%d N=%d K=%d", M, N, K);
  AMK_CHECK_SUPPORTED(K % 4 == 0 && lda % 4 == 0 &&
                          ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(b_planes)) & 15) == 0,
                      "amk_gemm_x6_nt: K, lda must be multiples of 4 and A, the planes 16-byte aligned");
  AMK_CHECK_SUPPORTED(((int64_t)(M - 1) * lda + K) * 4 < 0x7ffffff0ll && amk_gemm_x6_planes_bytes(N, K) < 0x7ffffff0ll,
                      "amk_gemm_x6_nt: an operand must span < 2 GiB");
  Args g;
  g.A = A; g.Bp = static_cast<const __bf16*>(b_planes); g.bias = bias; g.C = C;
  g.M = M; g.N = N; g.K = K; g.Kp = padded_k(K); g.lda = lda; g.ldc = ldc;
  const int64_t nwg = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  AMK_CHECK_SUPPORTED(nwg < (1ll << 31), "amk_gemm_x6_nt: grid too large");
  hipLaunchKernelGGL(gemm_x6_nt_kernel, dim3((unsigned)nwg), dim3(NTHR), 0, static_cast<hipStream_t>(stream), g);
  AMK_CHECK_LAUNCH("amk_gemm_x6_nt");
  return AMK_OK;
}
