// Dense exact-f32 GEMMs for the nn.Linear layers around the attention core and the FFN, with the element-wise
// passes that surround them folded into operand staging and epilogues (SURVEY.md section 8f rank 1;
// reference: models/softmax_attention.py:30-42,80 -- q / kv / W_o projections --, models/vitvqgan.py:20-61 --
// pre-LN, SwiGLU FFN, residual adds).  Three products, all on v_mfma_f32_32x32x2_f32 (exact f32, the same
// rounding as an fmaf chain):
//   NT  C[m, n]  = sum_k A'[m, k] W[n, k] (+ bias[n]) (+ R[m, n])      F.linear; A' = A or LayerNorm(A) applied
//                                                                      while the tile is staged (mean / rstd given)
//       SwiGLU epilogue: W = w12 (2H, K); the tile pairs gate column j with column H + j and writes
//                        g = silu(a) * b (and (a | b) when the backward will need it)
//   NN  C[m, n]  = sum_k A[m, k] W[k, n]                               input gradient dX = dY W, the contraction cut
//                                                                      into two (A, W) segments (dq | dkv)
//       SwiGLU-backward epilogue: the tile is dG; it reads (a | b) and writes (dA | dB)
//   TN  C[n, k]  = sum_m Y[m, n] X'[m, k]                              weight gradient dW = dY^T X (X' = X or
//                                                                      LayerNorm(X)), bias gradient = column sums of
//                                                                      Y, the contraction (the M rows) cut into
//                                                                      chunks whose partial tiles are summed in order
// Tile: 128 x 128 outputs x BK deep (BK = 16 or 32), four waves as 2 x 2, 64 x 64 per wave (four 32x32
// accumulators).  Pipeline (as the expert GEMMs of moe.hip): two LDS stages, ONE barrier per step; during the
// MFMAs of step t every thread moves its 16-byte pieces of tile t+1 from registers to the other stage and
// refills them with tile t+2 through raw buffer loads (range-checked by the hardware: rows past the matrix read
// as zeros, no compares); no branch around a vector-memory instruction, so every wait is a counted one.
// Operand tiles come in two LDS images: "R" = [row][BK contraction values] (row stride BK + 4 floats, read as
// ds_read_b128: four MFMA steps per read) and "C" = [BK contraction rows][128 columns] (row stride 132, read as
// ds_read_b32, consecutive lanes on consecutive columns).  MFMA step (s4, x) of lane half hf contracts index
// (BK / 2) hf + 4 s4 + x of the BK -- the same for both operands, any order of the contraction is as good as another.
#include "amk_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#ifndef AMK_DENSE_DEFAULT_BK
#define AMK_DENSE_DEFAULT_BK 32
#endif

namespace amk_dense {

constexpr int LSC = 132;
constexpr unsigned ROW_PAST = 0x40000000u;  // added to an offset: beyond every buffer (records < 1 GiB)
constexpr unsigned K_PAST = 0x80000000u;    // ROW_PAST + K_PAST + offset does not wrap

template <int BK>
struct Geo {
  static constexpr int LSR = BK + 4;
  static constexpr int BREG = 128 * LSR;         // operand A region, then operand B region (each <= 128 * LSR floats)
  static constexpr int STAGE = 2 * BREG;
  static constexpr int NG = BK / 8;              // MFMA groups (16 MFMA each) per step
  static constexpr int NP = BK / 8;              // 16-byte pieces per operand, thread and step
  static constexpr int TPR = BK / 4;             // threads per row of an "R" image
  static constexpr int RPP = 256 / TPR;          // rows per pass of an "R" image
  static_assert(BK == 16 || BK == 32, "BK");
  static_assert(BK * LSC <= BREG, "the C image must fit the operand region");
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void bstore1(float v, __amdgpu_buffer_rsrc_t r, int off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes < 0 ? 0 : (bytes > 0x3FFFFFFF ? 0x3FFFFFFF : bytes)), 0x00020000);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// barrier that orders LDS only: the global loads in flight stay in flight (__syncthreads() would drain outstanding stores)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

enum { EPI_BIAS = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_SWIGLU_BWD = 3 };

struct Params {
  // operands (meaning per product, see the kernels)
  const float *a, *a2, *w, *w2, *bias, *bias2, *resid;
  float *c, *c2;
  const float *ln_mean, *ln_rstd, *ln_gamma, *ln_beta;
  const float* ab;   // SwiGLU backward: the forward's (a | b), (M, 2H)
  float* gate;       // SwiGLU forward: g (M, H)
  float* dbias;      // TN: bias-gradient partials / output
  float* ws;         // TN: partial tiles (chunks, N, K)
  int64_t lda, lda2, ldw, ldw2, ldc, ldc2, ldr, ldab, ldg;
  int64_t M;
  int N, K, split, H;
  int ntn, total;    // column tiles per row tile, workgroups
  int nchunk, steps_per_chunk;  // TN
  int stagger_mode, stagger_sleeps;  // see stagger()
};

// All workgroups of a launch start together and run tiles of equal length: the two (three) workgroups that share a
// CU load their first tiles, and later store their results, at the same time, with nobody's MFMAs to cover it.
// The workgroups of the first dispatch round that hold the "second" slot of a CU start a few microseconds late;
// their successors inherit the offset.  mode 1: slot = bit 5 of the workgroup's index inside its XCD (CUs filled
// breadth first); mode 2: bit 0 (depth first); mode 3: the XOR of both.
__device__ __forceinline__ void stagger(const Params& p, int slots) {
  if (p.stagger_mode == 0 || (int)blockIdx.x >= slots) return;
  const int k = blockIdx.x >> 3;
  const int late = p.stagger_mode == 1 ? (k >> 5) & 1 : (p.stagger_mode == 2 ? k & 1 : ((k >> 5) ^ k) & 1);
  if (late) {
    for (int i = 0; i < p.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  }
}

// Epilogue addressing.  A lane's 16 values of a 32x32 accumulator are rows e + 8 q + 4 hf (e, q = 0..3) of one column:
// per group of 8 rows (i, q) a descriptor of its own (scalar registers: base at the group's first row, records up to
// the last VALID row of the group -- rows past the matrix are dropped by the range check) and four lane offsets
// (rows 4 hf + e, the lane's column) shared by all groups: 4 vector registers per leading dimension instead of 32.
__device__ __forceinline__ void lane_offsets(unsigned (&vo)[4], int hf, int64_t ld, unsigned colbytes) {
#pragma unroll
  for (int e = 0; e < 4; ++e) vo[e] = (unsigned)((4 * hf + e) * ld * 4) + colbytes;  // (colbytes may carry K_PAST)
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t group_rsrc(const float* base, int64_t ld, int rows, int rg0, int col0, int width) {
  int n = rows - rg0;
  n = n > 8 ? 8 : n;
  return mkrsrc(base + (int64_t)rg0 * ld + col0, n > 0 ? ((int64_t)(n - 1) * ld + width) * 4 : 0);
}

// Instruction issue on a SIMD is arbitrated by priority, then age, and the f32 MFMA stream of the co-resident
// workgroup leaves few slots: a wave that loads its first tiles or stores its results next to a partner in its tile
// loop was measured at 5 us (prologue) and 13 us (epilogue) for a few hundred instructions.  Those two phases run at
// priority 3 (they hold the pipe for a few cycles per instruction); the tile loop at AMK_DENSE_MOVE_PRIO around
// its data movement and 0 around its MFMAs.
#ifndef AMK_DENSE_EDGE_PRIO
#define AMK_DENSE_EDGE_PRIO 3
#endif
#ifndef AMK_DENSE_MOVE_PRIO
#define AMK_DENSE_MOVE_PRIO 0
#endif
// ---- the step loop shared by the three products -------------------------------------------------------------
// ACT / BCT: the A / B operand tile is a "C" image.  a_rd / b_rd: this lane's read base inside a stage;
// a_blk / b_blk: distance between the wave's two 32-wide blocks of that operand.
// mv(i, nxt, kt): move piece i (0 .. 2 NP - 1) of tile kt+1 from registers into stage `nxt`, then load piece i of
// tile kt+2.
template <int BK, bool ACT, bool BCT, class Mover>
__device__ __forceinline__ void step_loop(f32x16 (&acc)[2][2], float* smem, int nk, int a_rd, int a_blk, int b_rd,
                                          int b_blk, Mover& mv) {
  using G = Geo<BK>;
  for (int kt = 0; kt < nk; ++kt) {
    const float* cur = smem + (kt & 1) * G::STAGE;
    float* nxt = smem + ((kt + 1) & 1) * G::STAGE;
    float a[2][4], b[2][4];
    auto frag = [&](int s4, float (&fa)[2][4], float (&fb)[2][4]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (!ACT) {
          const float4 v = ld4(cur + a_rd + i * a_blk + 4 * s4);
          fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
        } else {
#pragma unroll
          for (int x = 0; x < 4; ++x) fa[i][x] = cur[a_rd + i * a_blk + (4 * s4 + x) * LSC];
        }
        if constexpr (!BCT) {
          const float4 v = ld4(cur + G::BREG + b_rd + i * b_blk + 4 * s4);
          fb[i][0] = v.x; fb[i][1] = v.y; fb[i][2] = v.z; fb[i][3] = v.w;
        } else {
#pragma unroll
          for (int x = 0; x < 4; ++x) fb[i][x] = cur[G::BREG + b_rd + i * b_blk + (4 * s4 + x) * LSC];
        }
      }
    };
#ifndef AMK_ABLATE
#define AMK_ABLATE 0
#endif
    if (!(AMK_ABLATE & 4) || kt == 0) frag(0, a, b);
#pragma unroll
    for (int s4 = 0; s4 < G::NG; ++s4) {
      float an[2][4], bn[2][4];
      if (s4 + 1 < G::NG) {
        if (!(AMK_ABLATE & 4)) frag(s4 + 1, an, bn);
        else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int x = 0; x < 4; ++x) { an[i][x] = a[i][x]; bn[i][x] = b[i][x]; }
        }
      }
      if (AMK_DENSE_MOVE_PRIO) __builtin_amdgcn_s_setprio(AMK_DENSE_MOVE_PRIO);
      if (!(AMK_ABLATE & 1)) {
        mv(2 * s4, nxt, kt);
        mv(2 * s4 + 1, nxt, kt);
      }
      if (AMK_DENSE_MOVE_PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i][x], b[j][x], acc[i][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s4 + 1 < G::NG) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int x = 0; x < 4; ++x) { a[i][x] = an[i][x]; b[i][x] = bn[i][x]; }
        }
      }
    }
    if (!(AMK_ABLATE & 2)) __syncthreads();
  }
}

#define AMK_DENSE_BOUNDS(BK) __launch_bounds__(256, (BK) == 16 ? 3 : 2)

// =============================================================================================================
// NT: C = A' W^T.  A (M, K) "R" image, W (N, K) "R" image.
// Column segments: output columns [0, split) use (w, bias, c), columns [split, N) use (w2, bias2, c2) with their
// own leading dimensions -- q and kv projections in one launch (split a multiple of 128; 0 = one segment).
// SwiGLU: W = w12 (2H, K), N = H; tile nt covers gate columns [64 nt, 64 nt + 64): LDS rows [0, 64) = rows j of
// W, rows [64, 128) = rows H + j.
template <int BK, int EPI, bool LNA>
__global__ AMK_DENSE_BOUNDS(BK) void gemm_nt_kernel(Params p) {
  using G = Geo<BK>;
  constexpr int NP = G::NP, LSR = G::LSR;
  __shared__ __attribute__((aligned(16))) float smem[2 * G::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  stagger(p, (BK == 16 ? 3 : 2) * 256);
  __builtin_amdgcn_s_setprio(AMK_DENSE_EDGE_PRIO);
#ifdef AMK_DENSE_STAMPS
#define AMK_STAMP(i) do { if (tid == 0 && p.ws) reinterpret_cast<unsigned long long*>(p.ws)[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  AMK_STAMP(0);
  if (tid == 0 && p.ws) {
    reinterpret_cast<unsigned long long*>(p.ws)[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID
    reinterpret_cast<unsigned long long*>(p.ws)[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // XCC_ID
  }
#else
#define AMK_STAMP(i)
#endif
  const int u = xcd_remap(blockIdx.x, p.total);
  const int mt = u / p.ntn, nt = u - mt * p.ntn;
  const int64_t m0 = (int64_t)mt * 128;
  const int rows = (int)(p.M - m0 < 128 ? p.M - m0 : 128);
  constexpr bool SW = EPI == EPI_SWIGLU;
  // this tile's column segment
  int n0 = SW ? nt * 64 : nt * 128;
  const bool seg = !SW && p.split > 0 && n0 >= p.split;
  const float* Wb = seg ? p.w2 : p.w;
  const float* bias = seg ? p.bias2 : p.bias;
  float* Cb = seg ? p.c2 : p.c;
  const int64_t ldw = seg ? p.ldw2 : p.ldw, ldc = seg ? p.ldc2 : p.ldc;
  const int ncols = SW ? p.H : (p.split > 0 ? (seg ? p.N - p.split : p.split) : p.N);
  if (seg) n0 -= p.split;
  const int wrows = SW ? 2 * p.H : ncols;

  const int sr = tid / G::TPR, sc = (tid % G::TPR) * 4;
  const __amdgpu_buffer_rsrc_t a_rsrc = mkrsrc(p.a + m0 * p.lda, ((int64_t)(rows - 1) * p.lda + p.K) * 4);
  const __amdgpu_buffer_rsrc_t w_rsrc = mkrsrc(Wb, ((int64_t)(wrows - 1) * ldw + p.K) * 4);
  const __amdgpu_buffer_rsrc_t g_rsrc = mkrsrc(LNA ? p.ln_gamma : p.a, LNA ? (int64_t)p.K * 4 : 0);
  const __amdgpu_buffer_rsrc_t be_rsrc = mkrsrc(LNA ? p.ln_beta : p.a, LNA ? (int64_t)p.K * 4 : 0);
  unsigned goff[2 * NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) goff[j] = (unsigned)(((int64_t)(sr + G::RPP * j) * p.lda + sc) * 4);
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int rr = sr + G::RPP * j;  // LDS row of the W region
    int wr;                          // row of W
    bool ok;
    if (SW) { const int jj = n0 + (rr & 63); ok = jj < p.H; wr = (rr < 64 ? 0 : p.H) + jj; }
    else { wr = n0 + rr; ok = wr < ncols; }
    goff[NP + j] = ok ? (unsigned)(((int64_t)wr * ldw + sc) * 4) : ROW_PAST;
  }
  float rs[NP], mu[NP];
  if (LNA) {
    const __amdgpu_buffer_rsrc_t m_rsrc = mkrsrc(p.ln_mean + m0, (int64_t)rows * 4);
    const __amdgpu_buffer_rsrc_t r_rsrc = mkrsrc(p.ln_rstd + m0, (int64_t)rows * 4);
#pragma unroll
    for (int j = 0; j < NP; ++j) { mu[j] = bload1(m_rsrc, (sr + G::RPP * j) * 4); rs[j] = bload1(r_rsrc, (sr + G::RPP * j) * 4); }
  }
  float4 stg[2 * NP], gam, bet;
  const int nk = (p.K + BK - 1) / BK;
  auto kadd = [&](int kt) {
    const int k0 = (kt < nk ? kt : nk - 1) * BK;
    return (k0 + sc < p.K) ? (unsigned)(k0 * 4) : K_PAST;
  };
  auto gload = [&](int i, unsigned ka) {
    if (i < NP) stg[i] = bload4(a_rsrc, (int)(goff[i] + ka));
    else stg[i] = bload4(w_rsrc, (int)(goff[i] + ka));
  };
  auto lstore = [&](int i, float* stage) {
    float4 v = stg[i];
    if (LNA && i < NP) {
      v.x = fmaf((v.x - mu[i]) * rs[i], gam.x, bet.x); v.y = fmaf((v.y - mu[i]) * rs[i], gam.y, bet.y);
      v.z = fmaf((v.z - mu[i]) * rs[i], gam.z, bet.z); v.w = fmaf((v.w - mu[i]) * rs[i], gam.w, bet.w);
    }
    st4(&stage[(i < NP ? 0 : G::BREG) + (sr + G::RPP * (i % NP)) * LSR + sc], v);
  };
  auto gbload = [&](unsigned ka) {  // gamma / beta of a tile's 4 columns (zeros past K: the K tail stays zero)
    if (LNA) { gam = bload4(g_rsrc, (int)(ka + sc * 4)); bet = bload4(be_rsrc, (int)(ka + sc * 4)); }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

  unsigned ka = kadd(0);
  gbload(ka);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) gload(i, ka);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) lstore(i, smem);
  ka = kadd(1);
  gbload(ka);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) gload(i, ka);
  __syncthreads();
  AMK_STAMP(1);

  unsigned ka2 = 0;
  auto mv = [&](int i, float* nxt, int kt) {
    if (i == 0) ka2 = kadd(kt + 2);
    lstore(i, nxt);
    if (i == NP - 1) gbload(ka2);  // the A pieces of tile kt+1 are stored: their gamma / beta registers are free
    gload(i, ka2);
  };
  const int a_rd = (64 * wm + ln) * LSR + (BK / 2) * hf;
  const int b_rd = ((SW ? 32 : 64) * wn + ln) * LSR + (BK / 2) * hf;
  __builtin_amdgcn_s_setprio(0);
  step_loop<BK, false, false>(acc, smem, nk, a_rd, 32 * LSR, b_rd, (SW ? 64 : 32) * LSR, mv);
  __builtin_amdgcn_s_setprio(AMK_DENSE_EDGE_PRIO);
  AMK_STAMP(2);

  // ---- epilogue
  if constexpr (SW) {
    const int j = n0 + 32 * wn + ln;
    const bool ok = j < p.H;
    const float ba = (bias && ok) ? bias[j] : 0.f, bb = (bias && ok) ? bias[p.H + j] : 0.f;
    const unsigned cg = ok ? (unsigned)j * 4u : K_PAST;
    unsigned vg[4], vc[4];
    lane_offsets(vg, hf, p.ldg, cg);
    lane_offsets(vc, hf, ldc, cg);
    const float* gbase = p.gate + m0 * p.ldg;
    const float* abase = Cb ? Cb + m0 * ldc : p.gate;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rg0 = 64 * wm + 32 * i + 8 * q;
        const __amdgpu_buffer_rsrc_t gr = group_rsrc(gbase, p.ldg, rows, rg0, 0, p.H);
        const __amdgpu_buffer_rsrc_t ar = group_rsrc(abase, ldc, Cb ? rows : 0, rg0, 0, p.H);
        const __amdgpu_buffer_rsrc_t br = group_rsrc(abase, ldc, Cb ? rows : 0, rg0, p.H, p.H);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = acc[i][0][4 * q + e] + ba, bv = acc[i][1][4 * q + e] + bb;
          bstore1(av * sigmoidf_(av) * bv, gr, (int)vg[e]);
          if (Cb) { bstore1(av, ar, (int)vc[e]); bstore1(bv, br, (int)vc[e]); }
        }
      }
    }
    return;
  }
  const float* cbase = Cb + m0 * ldc;
  const float* rbase = EPI == EPI_RESID ? p.resid + m0 * p.ldr : p.a;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 64 * wn + 32 * j + ln;
    const bool ok = n < ncols;
    const float bv = (bias && ok) ? bias[n] : 0.f;
    const unsigned cn = ok ? (unsigned)n * 4u : K_PAST;
    unsigned vo[4], vr[4];
    lane_offsets(vo, hf, ldc, cn);
    if (EPI == EPI_RESID) lane_offsets(vr, hf, p.ldr, cn);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float res[16];
      if (EPI == EPI_RESID) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const __amdgpu_buffer_rsrc_t rr = group_rsrc(rbase, p.ldr, rows, 64 * wm + 32 * i + 8 * q, 0, ncols);
#pragma unroll
          for (int e = 0; e < 4; ++e) res[4 * q + e] = bload1(rr, (int)vr[e]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const __amdgpu_buffer_rsrc_t cr = group_rsrc(cbase, ldc, rows, 64 * wm + 32 * i + 8 * q, 0, ncols);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][4 * q + e] + bv;
          if (EPI == EPI_RESID) v += res[4 * q + e];
          bstore1(v, cr, (int)vo[e]);
        }
      }
    }
  }
  AMK_STAMP(3);
}

// =============================================================================================================
// NT, persistent form (K >= 4 steps).  Time stamps of the kernel above (tools/scratch/stamps.py) show where a 128 x
// 128 x 256 tile loses a fifth of its time: all workgroups of a launch move in step, so every CU loads its first tiles
// (2 us) and stores its results (4-5 us: 512 workgroups x 64 KiB hit HBM together) with nobody's MFMAs in flight,
// and offsetting the two workgroups of a CU does not help -- a wave storing its results next to a partner in its tile
// loop crawls (13 us).  Here a workgroup walks tiles u = blockIdx.x, + gridDim.x, ..., its global loads run two steps
// ahead ACROSS tiles, and the results of tile n leave DURING the first four steps of tile n+1: at the end of a tile
// the accumulators move to a second register set (64 v_mov), and each of the next 16 MFMA groups carries one
// 8-row group of them -- bias, residual (loaded two groups ahead), SwiGLU gate, four to twelve buffer stores.  The
// instruction stream of a wave is then the same all the way: 16 MFMA, a few loads / stores / LDS moves, 16 MFMA...
template <int EPI, bool LNA>
struct NtWalk {
  static constexpr int BK = 32;
  using G = Geo<32>;
  static constexpr int LSR = G::LSR;
  static constexpr bool SW = EPI == EPI_SWIGLU;
  const Params& p;
  int tid, sr, sc, ln, hf, wm, wn, nk;
  // ---- load side (the tile whose operands are being fetched)
  __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc, g_rsrc, be_rsrc, m_rsrc, r_rsrc;
  unsigned goff[8], ka;
  float4 stg[8], gam, bet;
  float mu[4], rs[4];
  // ---- drain side (the tile whose results sit in accp)
  f32x16 accp[2][2];
  const float *cbase_d, *rbase_d, *gbase_d;
  int64_t ldc_d;
  int rows_d, ncols_d;
  bool keep_ab_d;
  unsigned vo[4], vr[4], vg[4], d1;  // lane offsets (C / residual / gate) of column block 0; d1: what block 1 adds
  float bias_d[2];
  float res[2][4];

  __device__ __forceinline__ NtWalk(const Params& p_) : p(p_) {
    tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ln = lane & 31; hf = lane >> 5; wm = wave >> 1; wn = wave & 1;
    sr = tid >> 3; sc = (tid & 7) * 4;
    nk = (p.K + BK - 1) / BK;
    g_rsrc = mkrsrc(LNA ? p.ln_gamma : p.a, LNA ? (int64_t)p.K * 4 : 0);
    be_rsrc = mkrsrc(LNA ? p.ln_beta : p.a, LNA ? (int64_t)p.K * 4 : 0);
    rows_d = 0;  // nothing to drain yet: every drain access is out of range
    cbase_d = rbase_d = gbase_d = p.a;
    ldc_d = 0; ncols_d = 0; keep_ab_d = false; d1 = 0;
    bias_d[0] = bias_d[1] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { vo[e] = vr[e] = vg[e] = K_PAST; res[0][e] = res[1][e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) accp[i][j] = zero16();
  }
  __device__ __forceinline__ void decode(int u, int64_t& m0, int& rows, int& n0, bool& seg, int& ncols) const {
    const int v = xcd_remap(u, p.total);
    const int mt = v / p.ntn, nt = v - mt * p.ntn;
    m0 = (int64_t)mt * 128;
    rows = (int)(p.M - m0 < 128 ? p.M - m0 : 128);
    n0 = SW ? nt * 64 : nt * 128;
    seg = !SW && p.split > 0 && n0 >= p.split;
    ncols = SW ? p.H : (p.split > 0 ? (seg ? p.N - p.split : p.split) : p.N);
    if (seg) n0 -= p.split;
  }
  __device__ __forceinline__ void load_unit(int u) {
    int64_t m0; int rows, n0, ncols; bool seg;
    decode(u, m0, rows, n0, seg, ncols);
    const float* Wb = seg ? p.w2 : p.w;
    const int64_t ldw = seg ? p.ldw2 : p.ldw;
    const int wrows = SW ? 2 * p.H : ncols;
    a_rsrc = mkrsrc(p.a + m0 * p.lda, ((int64_t)(rows - 1) * p.lda + p.K) * 4);
    w_rsrc = mkrsrc(Wb, ((int64_t)(wrows - 1) * ldw + p.K) * 4);
    if (LNA) { m_rsrc = mkrsrc(p.ln_mean + m0, (int64_t)rows * 4); r_rsrc = mkrsrc(p.ln_rstd + m0, (int64_t)rows * 4); }
#pragma unroll
    for (int j = 0; j < 4; ++j) goff[j] = (unsigned)(((int64_t)(sr + 32 * j) * p.lda + sc) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = sr + 32 * j;
      int wr; bool ok;
      if (SW) { const int jj = n0 + (rr & 63); ok = jj < p.H; wr = (rr < 64 ? 0 : p.H) + jj; }
      else { wr = n0 + rr; ok = wr < ncols; }
      goff[4 + j] = ok ? (unsigned)(((int64_t)wr * ldw + sc) * 4) : ROW_PAST;
    }
  }
  __device__ __forceinline__ void gload(int i, int kt) {
    if (i == 0) { const int k0 = kt * BK; ka = (k0 + sc < p.K) ? (unsigned)(k0 * 4) : K_PAST; }
    if (i < 4) {
      stg[i] = bload4(a_rsrc, (int)(goff[i] + ka));
      // (mean / rstd of the piece's row: re-read every step -- a load under `if (new tile)` would be a branch around a
      //  vector-memory instruction, after which no wait can be counted)
      if (LNA) { mu[i] = bload1(m_rsrc, (sr + 32 * i) * 4); rs[i] = bload1(r_rsrc, (sr + 32 * i) * 4); }
      if (LNA && i == 3) { gam = bload4(g_rsrc, (int)(ka + sc * 4)); bet = bload4(be_rsrc, (int)(ka + sc * 4)); }
    } else stg[i] = bload4(w_rsrc, (int)(goff[i] + ka));
  }
  __device__ __forceinline__ void lstore(int i, float* stage) {
    float4 v = stg[i];
    if (LNA && i < 4) {
      v.x = fmaf((v.x - mu[i]) * rs[i], gam.x, bet.x); v.y = fmaf((v.y - mu[i]) * rs[i], gam.y, bet.y);
      v.z = fmaf((v.z - mu[i]) * rs[i], gam.z, bet.z); v.w = fmaf((v.w - mu[i]) * rs[i], gam.w, bet.w);
    }
    st4(&stage[(i < 4 ? 0 : G::BREG) + (sr + 32 * (i & 3)) * LSR + sc], v);
  }
  // the finished tile u becomes the one to drain (straight-line code: its loads may be waited for by count)
  __device__ __forceinline__ void retire(f32x16 (&acc)[2][2], int u) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) accp[i][j] = acc[i][j];
    int64_t m0; int n0; bool seg;
    decode(u, m0, rows_d, n0, seg, ncols_d);
    const float* bias = seg ? p.bias2 : p.bias;
    float* Cb = seg ? p.c2 : p.c;
    ldc_d = seg ? p.ldc2 : p.ldc;
    const __amdgpu_buffer_rsrc_t b_rsrc = mkrsrc(bias ? bias : p.a, bias ? (int64_t)(SW ? 2 * p.H : ncols_d) * 4 : 0);
    if constexpr (SW) {
      const int j = n0 + 32 * wn + ln;
      const bool ok = j < p.H;
      const unsigned cg = ok ? (unsigned)j * 4u : K_PAST;
      bias_d[0] = bload1(b_rsrc, (int)cg);
      bias_d[1] = bload1(b_rsrc, (int)(cg + (unsigned)p.H * 4u));
      lane_offsets(vg, hf, p.ldg, cg);
      lane_offsets(vo, hf, ldc_d, cg);
      gbase_d = p.gate + m0 * p.ldg;
      keep_ab_d = Cb != nullptr;
      cbase_d = Cb ? Cb + m0 * ldc_d : p.gate;
    } else {
      const int n = n0 + 64 * wn + ln;
      const bool ok0 = n < ncols_d, ok1 = n + 32 < ncols_d;
      const unsigned cn = ok0 ? (unsigned)n * 4u : K_PAST;
      d1 = ok1 ? 128u : (ok0 ? K_PAST : 0u);
      bias_d[0] = bload1(b_rsrc, (int)cn);
      bias_d[1] = bload1(b_rsrc, (int)(cn + d1));
      lane_offsets(vo, hf, ldc_d, cn);
      cbase_d = Cb + m0 * ldc_d;
      if (EPI == EPI_RESID) {
        lane_offsets(vr, hf, p.ldr, cn);
        rbase_d = p.resid + m0 * p.ldr;
        res_load(0, 0);
        res_load(0, 1);
      }
    }
  }
  // residual values of drain group (block d, row group q) into ring slot q & 1
  __device__ __forceinline__ void res_load(int d, int q) {
    const int i = d & 1, j = d >> 1;
    const __amdgpu_buffer_rsrc_t rr = group_rsrc(rbase_d, p.ldr, rows_d, 64 * wm + 32 * i + 8 * q, 0, ncols_d);
#pragma unroll
    for (int e = 0; e < 4; ++e) res[q & 1][e] = bload1(rr, (int)(vr[e] + (j ? d1 : 0u)));
  }
  // one of the 16 drain groups: block d = (i, j) (SwiGLU: i and a pair of row groups), MFMA group g
  __device__ __forceinline__ void drain(int d, int g) {
    if constexpr (SW) {
      const int i = d & 1, q = 2 * (d >> 1) + (g >> 1);
      const int rg0 = 64 * wm + 32 * i + 8 * q;
      const __amdgpu_buffer_rsrc_t gr = group_rsrc(gbase_d, p.ldg, rows_d, rg0, 0, p.H);
      const __amdgpu_buffer_rsrc_t ar = group_rsrc(cbase_d, ldc_d, keep_ab_d ? rows_d : 0, rg0, 0, p.H);
      const __amdgpu_buffer_rsrc_t br = group_rsrc(cbase_d, ldc_d, keep_ab_d ? rows_d : 0, rg0, p.H, p.H);
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int e = 2 * (g & 1) + e2;
        const float av = accp[i][0][4 * q + e] + bias_d[0], bv = accp[i][1][4 * q + e] + bias_d[1];
        bstore1(av * sigmoidf_(av) * bv, gr, (int)vg[e]);
        bstore1(av, ar, (int)vo[e]);  // (zero records without keep_ab: dropped)
        bstore1(bv, br, (int)vo[e]);
      }
    } else {
      const int i = d & 1, j = d >> 1, q = g;
      const __amdgpu_buffer_rsrc_t cr = group_rsrc(cbase_d, ldc_d, rows_d, 64 * wm + 32 * i + 8 * q, 0, ncols_d);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = accp[i][j][4 * q + e] + bias_d[j];
        if (EPI == EPI_RESID) v += res[q & 1][e];
        bstore1(v, cr, (int)(vo[e] + (j ? d1 : 0u)));
      }
      if (EPI == EPI_RESID) {  // refill the slot with the group two ahead
        const int nxt = 4 * d + g + 2;
        if (nxt < 16) res_load(nxt >> 2, nxt & 3);
      }
    }
  }
};

template <int EPI, bool LNA>
__global__ __launch_bounds__(256, 2) void gemm_nt_dkernel(Params p) {
  using W = NtWalk<EPI, LNA>;
  using G = Geo<32>;
  constexpr int LSR = G::LSR;
  constexpr bool SW = EPI == EPI_SWIGLU;
  __shared__ __attribute__((aligned(16))) float smem[2 * G::STAGE];
  W U(p);
  const int Gd = gridDim.x;
  int cu = blockIdx.x;
  if (cu >= p.total) return;
  // load cursor: two steps ahead of the MFMAs, across tiles
  int lu = cu, lk = 0;
  U.load_unit(lu);
  auto advance = [&]() {
    if (++lk == U.nk) {
      lk = 0;
      lu = lu + Gd < p.total ? lu + Gd : lu;  // past the end: the last tile again (loaded, never used)
      U.load_unit(lu);
    }
  };
  const int a_rd = (64 * U.wm + U.ln) * LSR + 16 * U.hf;
  const int b_rd = ((SW ? 32 : 64) * U.wn + U.ln) * LSR + 16 * U.hf;
  constexpr int a_blk = 32 * LSR, b_blk = (SW ? 64 : 32) * LSR;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
#pragma unroll
  for (int i = 0; i < 8; ++i) U.gload(i, lk);
  advance();
#pragma unroll
  for (int i = 0; i < 8; ++i) U.lstore(i, smem);
#pragma unroll
  for (int i = 0; i < 8; ++i) U.gload(i, lk);
  advance();
  lds_barrier();
  int s = 0;  // steps done: LDS stage parity
  auto frag = [&](const float* st, int s4, float (&fa)[2][4], float (&fb)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v = ld4(st + a_rd + i * a_blk + 4 * s4);
      fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
      const float4 w = ld4(st + G::BREG + b_rd + i * b_blk + 4 * s4);
      fb[i][0] = w.x; fb[i][1] = w.y; fb[i][2] = w.z; fb[i][3] = w.w;
    }
  };
  // one step; D >= 0: it also drains block D of the retired tile.  The step's barrier sits after its third MFMA
  // group (AMK_WALK_BARRIER_AT = 2; 3 = at the end): the fragments of the NEXT step's first group are read behind
  // it, under the fourth group's MFMAs, so a step does not begin by waiting for LDS.  All stores of a step into the
  // other stage happen in its groups 0-2, all its fragment reads of the current stage are issued before the barrier.
#ifndef AMK_WALK_BARRIER_AT
#define AMK_WALK_BARRIER_AT 2
#endif
  float a[2][4], b[2][4];
  frag(smem, 0, a, b);
  auto step = [&](auto dtag) {
    constexpr int D = decltype(dtag)::value;
    constexpr int BAT = AMK_WALK_BARRIER_AT;
    const float* cur = smem + (s & 1) * G::STAGE;
    float* nxt = smem + ((s + 1) & 1) * G::STAGE;
    auto mv = [&](int i) { U.lstore(i, nxt); U.gload(i, lk); };
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float an[2][4], bn[2][4];
      if (g + 1 < 4) frag(cur, g + 1, an, bn);
      if (BAT == 3) { mv(2 * g); mv(2 * g + 1); }
      else if (g == 0) { mv(0); mv(1); mv(2); }
      else if (g == 1) { mv(3); mv(4); mv(5); }
      else if (g == 2) { mv(6); mv(7); }
      if (g == BAT || (BAT == 3 && g == 3)) { if (g == (BAT == 3 ? 3 : 2)) advance(); }
      if (D >= 0) U.drain(D, g);
      if (BAT == 2 && g == 3) frag(nxt, 0, an, bn);  // (behind the barrier below group 2)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i][x], b[j][x], (D == 0 && g == 0 && x == 0) ? zero16() : acc[i][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (BAT == 2 && g == 2) lds_barrier();
      if (g + 1 < 4 || BAT == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int x = 0; x < 4; ++x) { a[i][x] = an[i][x]; b[i][x] = bn[i][x]; }
      }
    }
    if (BAT == 3) { lds_barrier(); frag(nxt, 0, a, b); }
    ++s;
  };
  for (;;) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    for (int kt = 4; kt < U.nk; ++kt) step(std::integral_constant<int, -1>{});
    U.retire(acc, cu);
    cu += Gd;
    if (cu >= p.total) break;
  }
  // the last tile's results
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) U.drain(d, g);
}

// =============================================================================================================
// NN: C[m, n] = sum_k A[m, k] W[k, n].  A (M, K) "R" image, W (K, N) "C" image.
// Contraction segments: k in [0, split) reads (a, w), k in [split, K) reads (a2, w2) at k - split (split = 0: one).
// SwiGLU backward: N = H, the tile is dG; c = (dA | dB) (M, 2H), ab = the forward's (a | b).
template <int BK, int EPI>
__global__ AMK_DENSE_BOUNDS(BK) void gemm_nn_kernel(Params p) {
  using G = Geo<BK>;
  constexpr int NP = G::NP, LSR = G::LSR;
  __shared__ __attribute__((aligned(16))) float smem[2 * G::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  stagger(p, (BK == 16 ? 3 : 2) * 256);
  __builtin_amdgcn_s_setprio(AMK_DENSE_EDGE_PRIO);
  const int u = xcd_remap(blockIdx.x, p.total);
  const int mt = u / p.ntn, nt = u - mt * p.ntn;
  const int64_t m0 = (int64_t)mt * 128;
  const int rows = (int)(p.M - m0 < 128 ? p.M - m0 : 128);
  const int n0 = nt * 128;
  const int K0 = p.split > 0 ? p.split : p.K, K1 = p.K - K0;
  const int nk0 = (K0 + BK - 1) / BK, nk1 = (K1 + BK - 1) / BK, nk = nk0 + nk1;

  const int sr = tid / G::TPR, sc = (tid % G::TPR) * 4;   // A pieces: rows sr + RPP j, 4 floats at column sc
  const int cr = tid >> 5, cc = (tid & 31) * 4;           // W pieces: contraction rows cr + 8 j, 4 floats at column cc
  const bool col_ok = n0 + cc < p.N;
  // byte offsets of this thread's pieces inside the row panel of A / the slab of W, per contraction segment
  unsigned aoff0[NP], aoff1[NP], woff0[NP], woff1[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    aoff0[j] = (unsigned)(((int64_t)(sr + G::RPP * j) * p.lda + sc) * 4);
    aoff1[j] = (unsigned)(((int64_t)(sr + G::RPP * j) * p.lda2 + sc) * 4);
    woff0[j] = col_ok ? (unsigned)(((int64_t)(cr + 8 * j) * p.ldw + n0 + cc) * 4) : ROW_PAST;
    woff1[j] = col_ok ? (unsigned)(((int64_t)(cr + 8 * j) * p.ldw2 + n0 + cc) * 4) : ROW_PAST;
  }
  float4 stg[2 * NP];
  // segment state of the tile being loaded (wave-uniform)
  auto seg_of = [&](int kt, int& s, int& k0) {
    const int t = kt < nk ? kt : nk - 1;
    s = t >= nk0;
    k0 = (s ? t - nk0 : t) * BK;
  };
  auto gload_tile_piece = [&](int i, int s, int k0) {
    const float* ab_ = s ? p.a2 : p.a;
    const int64_t lda = s ? p.lda2 : p.lda;
    const float* wb_ = s ? p.w2 : p.w;
    const int64_t ldw = s ? p.ldw2 : p.ldw;
    const int Ks = s ? K1 : K0;
    if (i < NP) {
      const __amdgpu_buffer_rsrc_t ar = mkrsrc(ab_ + m0 * lda, ((int64_t)(rows - 1) * lda + Ks) * 4);
      const unsigned off = (k0 + sc < Ks) ? (s ? aoff1[i] : aoff0[i]) + (unsigned)(k0 * 4) : K_PAST;
      stg[i] = bload4(ar, (int)off);
    } else {
      const __amdgpu_buffer_rsrc_t wr = mkrsrc(wb_, ((int64_t)(Ks - 1) * ldw + p.N) * 4);
      stg[i] = bload4(wr, (int)((s ? woff1[i - NP] : woff0[i - NP]) + (unsigned)(k0 * ldw * 4)));
    }
  };
  auto lstore = [&](int i, float* stage) {
    if (i < NP) st4(&stage[(sr + G::RPP * i) * LSR + sc], stg[i]);
    else st4(&stage[G::BREG + (cr + 8 * (i - NP)) * LSC + cc], stg[i]);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  int s, k0;
  seg_of(0, s, k0);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) gload_tile_piece(i, s, k0);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) lstore(i, smem);
  seg_of(1, s, k0);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) gload_tile_piece(i, s, k0);
  __syncthreads();
  int s2 = 0, k2 = 0;
  auto mv = [&](int i, float* nxt, int kt) {
    if (i == 0) seg_of(kt + 2, s2, k2);
    lstore(i, nxt);
    gload_tile_piece(i, s2, k2);
  };
  const int a_rd = (64 * wm + ln) * LSR + (BK / 2) * hf;
  const int b_rd = ((BK / 2) * hf) * LSC + 64 * wn + ln;
  __builtin_amdgcn_s_setprio(0);
  step_loop<BK, false, true>(acc, smem, nk, a_rd, 32 * LSR, b_rd, 32, mv);
  __builtin_amdgcn_s_setprio(AMK_DENSE_EDGE_PRIO);

  const float* cbase = p.c + m0 * p.ldc;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 64 * wn + 32 * j + ln;
    const unsigned cn = n < p.N ? (unsigned)n * 4u : K_PAST;
    unsigned vo[4], va[4];
    lane_offsets(vo, hf, p.ldc, cn);
    if (EPI == EPI_SWIGLU_BWD) lane_offsets(va, hf, p.ldab, cn);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (EPI == EPI_SWIGLU_BWD) {  // N = H: the tile is dGate; (a | b) in, (dA | dB) out
        const float* abbase = p.ab + m0 * p.ldab;
        float av[16], bv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rg0 = 64 * wm + 32 * i + 8 * q;
          const __amdgpu_buffer_rsrc_t ar = group_rsrc(abbase, p.ldab, rows, rg0, 0, p.H), br = group_rsrc(abbase, p.ldab, rows, rg0, p.H, p.H);
#pragma unroll
          for (int e = 0; e < 4; ++e) { av[4 * q + e] = bload1(ar, (int)va[e]); bv[4 * q + e] = bload1(br, (int)va[e]); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rg0 = 64 * wm + 32 * i + 8 * q;
          const __amdgpu_buffer_rsrc_t dar = group_rsrc(cbase, p.ldc, rows, rg0, 0, p.H), dbr = group_rsrc(cbase, p.ldc, rows, rg0, p.H, p.H);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = acc[i][j][4 * q + e], a_ = av[4 * q + e], sg = sigmoidf_(a_);
            bstore1(g * bv[4 * q + e] * (sg * (1.f + a_ * (1.f - sg))), dar, (int)vo[e]);
            bstore1(g * (a_ * sg), dbr, (int)vo[e]);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const __amdgpu_buffer_rsrc_t cr_ = group_rsrc(cbase, p.ldc, rows, 64 * wm + 32 * i + 8 * q, 0, p.N);
#pragma unroll
          for (int e = 0; e < 4; ++e) bstore1(acc[i][j][4 * q + e], cr_, (int)vo[e]);
        }
      }
    }
  }
}

// =============================================================================================================
// TN: C[n, k] = sum_m Y[m, n] X'[m, k]; Y (M, N) and X (M, K) both "C" images (the contraction is the row index).
// Row segments of the output: n in [0, split) reads Y = a (lda) and writes c; n in [split, N) reads a2 (lda2) and
// writes c2 (dq | dkv).  X = w (ldw), optionally LayerNorm(X) (mean / rstd per row m, gamma / beta per column k).
// The M rows are cut into nchunk chunks of steps_per_chunk BK-row steps; with nchunk > 1 every workgroup writes its
// partial tile to ws[chunk] (chunks, N, K) and tn_reduce_kernel sums the chunks in order.  dbias (column sums of
// Y) is accumulated by the workgroups of the first k tile from the pieces they stage.
template <int BK, bool LNX>
__global__ AMK_DENSE_BOUNDS(BK) void gemm_tn_kernel(Params p) {
  using G = Geo<BK>;
  constexpr int NP = G::NP;
  __shared__ __attribute__((aligned(16))) float smem[2 * G::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  stagger(p, (BK == 16 ? 3 : 2) * 256);
  __builtin_amdgcn_s_setprio(AMK_DENSE_EDGE_PRIO);
  // unit = (tile, chunk), chunks of one tile adjacent; tiles [n tile][k tile]
  const int u = xcd_remap(blockIdx.x, p.total);
  const int tile = u / p.nchunk, chunk = u - tile * p.nchunk;
  const int ntk = p.ntn;  // k tiles per n tile
  const int tn = tile / ntk, tk = tile - tn * ntk;
  int n0 = tn * 128;
  const int k0t = tk * 128;
  const bool seg = p.split > 0 && n0 >= p.split;
  const float* Yb = seg ? p.a2 : p.a;
  const int64_t ldy = seg ? p.lda2 : p.lda;
  const int nrows = p.split > 0 ? (seg ? p.N - p.split : p.split) : p.N;  // rows of this segment's output
  if (seg) n0 -= p.split;
  const int64_t mbeg = (int64_t)chunk * p.steps_per_chunk * BK;
  int64_t mend = mbeg + (int64_t)p.steps_per_chunk * BK;
  if (mend > p.M) mend = p.M;
  const int mrows = (int)(mend - mbeg);
  const int nk = (mrows + BK - 1) / BK;

  const int cr = tid >> 5, cc = (tid & 31) * 4;
  const __amdgpu_buffer_rsrc_t y_rsrc = mkrsrc(Yb + mbeg * ldy, ((int64_t)(mrows - 1) * ldy + nrows) * 4);
  const __amdgpu_buffer_rsrc_t x_rsrc = mkrsrc(p.w + mbeg * p.ldw, ((int64_t)(mrows - 1) * p.ldw + p.K) * 4);
  const __amdgpu_buffer_rsrc_t m_rsrc = mkrsrc(LNX ? p.ln_mean + mbeg : p.a, LNX ? (int64_t)mrows * 4 : 0);
  const __amdgpu_buffer_rsrc_t r_rsrc = mkrsrc(LNX ? p.ln_rstd + mbeg : p.a, LNX ? (int64_t)mrows * 4 : 0);
  const bool ycol_ok = n0 + cc < nrows, xcol_ok = k0t + cc < p.K;
  unsigned goff[2 * NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    goff[j] = ycol_ok ? (unsigned)(((int64_t)(cr + 8 * j) * ldy + n0 + cc) * 4) : ROW_PAST;
    goff[NP + j] = xcol_ok ? (unsigned)(((int64_t)(cr + 8 * j) * p.ldw + k0t + cc) * 4) : ROW_PAST;
  }
  const unsigned ystep = (unsigned)(BK * ldy * 4), xstep = (unsigned)(BK * p.ldw * 4);
  float4 gam = make_float4(0.f, 0.f, 0.f, 0.f), bet = gam;
  if (LNX && xcol_ok) { gam = ld4(p.ln_gamma + k0t + cc); bet = ld4(p.ln_beta + k0t + cc); }
  float4 stg[2 * NP];
  float mu[NP], rs[NP];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_bias = p.dbias != nullptr && tk == 0;
  auto gload = [&](int i, int kt) {  // (rows past the chunk: past the descriptor, zeros)
    const int t = kt < nk ? kt : nk - 1;
    if (i < NP) stg[i] = bload4(y_rsrc, (int)(goff[i] + (unsigned)t * ystep));
    else {
      stg[i] = bload4(x_rsrc, (int)(goff[i] + (unsigned)t * xstep));
      if (LNX) {
        mu[i - NP] = bload1(m_rsrc, (t * BK + cr + 8 * (i - NP)) * 4);
        rs[i - NP] = bload1(r_rsrc, (t * BK + cr + 8 * (i - NP)) * 4);
      }
    }
  };
  auto lstore = [&](int i, float* stage, bool count) {
    float4 v = stg[i];
    if (i < NP) {
      if (do_bias && count) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
      st4(&stage[(cr + 8 * i) * LSC + cc], v);
    } else {
      if (LNX) {
        const float m_ = mu[i - NP], r_ = rs[i - NP];
        v.x = fmaf((v.x - m_) * r_, gam.x, bet.x); v.y = fmaf((v.y - m_) * r_, gam.y, bet.y);
        v.z = fmaf((v.z - m_) * r_, gam.z, bet.z); v.w = fmaf((v.w - m_) * r_, gam.w, bet.w);
      }
      st4(&stage[G::BREG + (cr + 8 * (i - NP)) * LSC + cc], v);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) gload(i, 0);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) lstore(i, smem, true);
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) gload(i, 1);
  __syncthreads();
  auto mv = [&](int i, float* nxt, int kt) {
    lstore(i, nxt, kt + 1 < nk);  // (the re-read tile past the end is stored, never used, and not counted)
    gload(i, kt + 2);
  };
  const int a_rd = ((BK / 2) * hf) * LSC + 64 * wm + ln;
  const int b_rd = ((BK / 2) * hf) * LSC + 64 * wn + ln;
  __builtin_amdgcn_s_setprio(0);
  step_loop<BK, true, true>(acc, smem, nk, a_rd, 32, b_rd, 32, mv);
  __builtin_amdgcn_s_setprio(AMK_DENSE_EDGE_PRIO);

  // ---- epilogue: the partial tile
  float* Cb;
  int64_t ldc;
  if (p.nchunk > 1) { Cb = p.ws + ((int64_t)chunk * p.N + (seg ? p.split : 0)) * p.K; ldc = p.K; }
  else { Cb = seg ? p.c2 : p.c; ldc = seg ? p.ldc2 : p.ldc; }
  const int orow = nrows - n0 < 128 ? nrows - n0 : 128;  // valid rows of the tile
  const float* cbase = Cb + (int64_t)n0 * ldc;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kcol = k0t + 64 * wn + 32 * j + ln;
    const unsigned cn = kcol < p.K ? (unsigned)kcol * 4u : K_PAST;
    unsigned vo[4];
    lane_offsets(vo, hf, ldc, cn);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const __amdgpu_buffer_rsrc_t cr_ = group_rsrc(cbase, ldc, orow, 64 * wm + 32 * i + 8 * q, 0, p.K);
#pragma unroll
        for (int e = 0; e < 4; ++e) bstore1(acc[i][j][4 * q + e], cr_, (int)vo[e]);
      }
    }
  }
  if (do_bias) {  // fold the eight row groups (cr) of each column quad
    float* red = smem;  // (the step loop ended with a barrier)
    st4(&red[cr * 128 + cc], bsum);
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += red[j * 128 + tid];
      const int n = n0 + tid;
      if (n < nrows) p.dbias[(int64_t)(p.nchunk > 1 ? chunk : 0) * p.N + (seg ? p.split : 0) + n] = s;
    }
  }
}

// C = sum over chunks of ws (in chunk order).  Elements [0, split * K) go to c (rows of ldc), the rest to c2.
// One thread per 16-byte piece; the chunks are read eight at a time (independent loads in flight) and added in order.
__global__ __launch_bounds__(256) void tn_reduce_kernel(Params p, float* dbias_out) {
  const int64_t total4 = (int64_t)p.N * p.K / 4;
  const int64_t slab = (int64_t)p.N * p.K;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total4) {
    const float* src = p.ws + 4 * i;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = 0;
    for (; c + 8 <= p.nchunk; c += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ld4(src + (c + j) * slab);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    for (; c < p.nchunk; ++c) {
      const float4 v = ld4(src + c * slab);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int64_t e = 4 * i;
    const int64_t row = e / p.K;
    const int col = (int)(e - row * p.K);
    if (p.split > 0 && row >= p.split) st4(p.c2 + (row - p.split) * p.ldc2 + col, s);
    else st4(p.c + row * p.ldc + col, s);
  }
  if (dbias_out && blockIdx.x == gridDim.x - 1) {
    for (int n = threadIdx.x; n < p.N; n += 256) {
      float s = 0.f;
      for (int c = 0; c < p.nchunk; ++c) s += p.dbias[(int64_t)c * p.N + n];
      dbias_out[n] = s;
    }
  }
}

// mean / rstd of every row (two-pass form, as add_layernorm_fwd): one wave per row, D <= 4096
template <int NCH>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int64_t M, int D, float eps,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  const float inv_d = 1.f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < nch ? ld4(x + row * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[j].x + v[j].y + v[j].z + v[j].w;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        const float a = v[j].x - mean, b = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
        q += a * a + b * b + cc * cc + d * d;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rsqrtf(q * inv_d + eps); }
  }
}

}  // namespace amk_dense

using namespace amk_dense;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Step depth of the tile loop: 16 (40 KiB of LDS, three workgroups per CU) or 32 (72 KiB, two per CU).
// AMK_DENSE_BK overrides the default (A/B switch of tools/kbench_dense.py).
static int dense_bk() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("AMK_DENSE_BK");
    v = (e && atoi(e) == 32) ? 32 : ((e && atoi(e) == 16) ? 16 : AMK_DENSE_DEFAULT_BK);
  }
  return v;
}
// workgroups the chip holds at once
static int wg_slots() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return (dense_bk() == 16 ? 3 : 2) * cus;
}

static int tn_chunks(const amk_gemm_desc* d, int* steps_per_chunk) {
  const int bk = dense_bk();
  const int tiles = ((d->n + 127) / 128) * ((d->k + 127) / 128);
  const int64_t steps = (d->m + bk - 1) / bk;
  const int64_t min_steps = 256 / bk;  // at least 256 rows per chunk
  // workgroups the grid aims at: the chip's slots (two per CU) for gradients with many tiles, half of them for the
  // small ones, where the f32 partial tiles (chunks x N x K, written and re-read) weigh more than the second workgroup
  // per CU brings -- measured, 512 / 256 slots: q 121 / 109 us, kv 174 / 167, W_o 111 / 105, but w12 415 / 477, w3 210 /
  // 229.  AMK_DENSE_TN_SLOTS overrides.
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("AMK_DENSE_TN_SLOTS");
    forced = e && atoi(e) > 0 ? atoi(e) : 0;
  }
  const int slots = forced ? forced : (tiles <= 16 ? wg_slots() / 2 : wg_slots());
  int64_t chunks = slots / tiles;  // workgroup slots over the tiles
  if (chunks < 1) chunks = 1;
  if (chunks > steps / min_steps) chunks = steps / min_steps > 0 ? steps / min_steps : 1;
  int64_t spc = (steps + chunks - 1) / chunks;
  chunks = (steps + spc - 1) / spc;
  *steps_per_chunk = (int)spc;
  return (int)chunks;
}

extern "C" int64_t amk_gemm_f32_ws_bytes(const amk_gemm_desc* d) {
  if (!d || d->op != AMK_GEMM_TN) return 0;
  int spc;
  const int chunks = tn_chunks(d, &spc);
  if (chunks <= 1) return 0;
  return ((int64_t)chunks * d->n * d->k + (int64_t)chunks * d->n) * 4;
}

extern "C" int amk_row_stats(const float* x, int64_t M, int D, float eps, float* mean, float* rstd, void* stream) {
  AMK_CHECK_ARG(x && mean && rstd, "amk_row_stats: null pointer");
  AMK_CHECK_ARG(M > 0 && D > 0, "amk_row_stats: non-positive size");
  AMK_CHECK_SUPPORTED(D % 4 == 0 && D <= 4096 && a16(x), "amk_row_stats: D %% 4 == 0, D <= 4096 and a 16-byte aligned pointer required");
  const int64_t blocks = (M + 3) / 4;
  const dim3 grid((unsigned)(blocks < 16384 ? blocks : 16384));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (D <= 256) hipLaunchKernelGGL(row_stats_kernel<1>, grid, dim3(256), 0, st, x, M, D, eps, mean, rstd);
  else if (D <= 1024) hipLaunchKernelGGL(row_stats_kernel<4>, grid, dim3(256), 0, st, x, M, D, eps, mean, rstd);
  else hipLaunchKernelGGL(row_stats_kernel<16>, grid, dim3(256), 0, st, x, M, D, eps, mean, rstd);
  AMK_CHECK_LAUNCH("amk_row_stats");
  return AMK_OK;
}

extern "C" int amk_gemm_f32(const amk_gemm_desc* d, void* workspace, int64_t ws_bytes, void* stream) {
  AMK_CHECK_ARG(d, "amk_gemm_f32: null descriptor");
  const bool c_optional = d->op == AMK_GEMM_NT && d->epilogue == AMK_EPI_SWIGLU;  // (a | b) is written only on request
  AMK_CHECK_ARG(d->a && d->w && (d->c || c_optional), "amk_gemm_f32: null operand");
  AMK_CHECK_ARG(d->m > 0 && d->n > 0 && d->k > 0, "amk_gemm_f32: non-positive size");
  const bool two = d->split > 0;
  AMK_CHECK_SUPPORTED(d->k % 4 == 0 && d->lda % 4 == 0 && d->ldw % 4 == 0 && d->ldc % 4 == 0 && a16(d->a) && a16(d->w) && a16(d->c),
                      "amk_gemm_f32: K and leading dimensions must be multiples of 4, pointers 16-byte aligned");
  AMK_CHECK_SUPPORTED(d->m < (int64_t)1 << 31, "amk_gemm_f32: M too large");
  const int64_t lim = (int64_t)1 << 30;  // operand panels are addressed through 32-bit buffer offsets
  Params p = {};
  p.a = d->a; p.a2 = d->a2; p.w = d->w; p.w2 = d->w2; p.bias = d->bias; p.bias2 = d->bias2; p.resid = d->resid;
  p.c = d->c; p.c2 = d->c2;
  p.ln_mean = d->ln_mean; p.ln_rstd = d->ln_rstd; p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta;
  p.ab = d->ab; p.gate = d->gate;
  p.lda = d->lda; p.lda2 = d->lda2; p.ldw = d->ldw; p.ldw2 = d->ldw2; p.ldc = d->ldc; p.ldc2 = d->ldc2;
  p.ldr = d->ldr; p.ldab = d->ldab; p.ldg = d->ldg;
  p.M = d->m; p.N = d->n; p.K = d->k; p.split = d->split; p.H = d->n;
  {
    static int mode = -1, sleeps = 0;
    if (mode < 0) {
      const char* e = getenv("AMK_DENSE_STAGGER");   // "mode,sleeps"
      mode = e ? atoi(e) : 0;
      const char* c = e ? strchr(e, ',') : nullptr;
      sleeps = c ? atoi(c + 1) : 2;
    }
    p.stagger_mode = mode; p.stagger_sleeps = sleeps;
  }
  const bool ln = d->ln_mean != nullptr;
  if (ln) AMK_CHECK_ARG(d->ln_rstd && d->ln_gamma && d->ln_beta, "amk_gemm_f32: LayerNorm operand needs mean, rstd, gamma and beta");
  if (two) {
    AMK_CHECK_ARG(d->split < (d->op == AMK_GEMM_NN ? d->k : d->n), "amk_gemm_f32: split outside the segmented axis");
    AMK_CHECK_SUPPORTED(d->split % 128 == 0 || d->op == AMK_GEMM_NN, "amk_gemm_f32: output split must be a multiple of 128");
    AMK_CHECK_SUPPORTED(d->op != AMK_GEMM_NN || d->split % 4 == 0, "amk_gemm_f32: contraction split must be a multiple of 4");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t mt = (d->m + 127) / 128;
  if (d->op == AMK_GEMM_NT) {
    const bool sw = d->epilogue == AMK_EPI_SWIGLU;
    if (two) AMK_CHECK_ARG(!sw && d->w2 && d->c2 && d->ldw2 % 4 == 0 && d->ldc2 % 4 == 0, "amk_gemm_f32: second column segment incomplete");
    AMK_CHECK_SUPPORTED(128 * d->lda * 4 < lim && (int64_t)(sw ? 2 : 1) * d->n * d->ldw * 4 < lim && 128 * d->ldc * 4 < lim,
                        "amk_gemm_f32: operand panel beyond 1 GiB");
    if (sw) AMK_CHECK_ARG(d->gate && d->ldg % 4 == 0, "amk_gemm_f32: SwiGLU epilogue needs the gate output");
    if (d->epilogue == AMK_EPI_RESID) AMK_CHECK_ARG(d->resid && !two && d->ldr % 4 == 0, "amk_gemm_f32: residual epilogue needs resid (one column segment)");
    p.ntn = sw ? (d->n + 63) / 64 : (two ? d->split / 128 + (d->n - d->split + 127) / 128 : (d->n + 127) / 128);
#ifdef AMK_DENSE_STAMPS
    p.ws = static_cast<float*>(workspace);  // diagnostic build: per-workgroup time stamps
#endif
    const int64_t total = mt * p.ntn;
    AMK_CHECK_SUPPORTED(total < (int64_t)1 << 31, "amk_gemm_f32: grid too large");
    p.total = (int)total;
    const dim3 grid((unsigned)total), blk(256);
    const bool b16 = dense_bk() == 16;
    // persistent walk with the results drained under the next tile's MFMAs (K of at least four 32-deep steps);
    // AMK_DENSE_WALK=0: one workgroup per tile
    static int walk = -1;
    if (walk < 0) { const char* e = getenv("AMK_DENSE_WALK"); walk = (e && e[0] == '0') ? 0 : 1; }
    const bool use_walk = walk && !b16 && d->k > 96;
    const dim3 wgrid((unsigned)(total > wg_slots() ? wg_slots() : total));
#define AMK_NT(E, L) do { if (use_walk) hipLaunchKernelGGL((gemm_nt_dkernel<E, L>), wgrid, blk, 0, st, p); else if (b16) hipLaunchKernelGGL((gemm_nt_kernel<16, E, L>), grid, blk, 0, st, p); else hipLaunchKernelGGL((gemm_nt_kernel<32, E, L>), grid, blk, 0, st, p); } while (0)
    if (sw) { if (ln) AMK_NT(EPI_SWIGLU, true); else AMK_NT(EPI_SWIGLU, false); }
    else if (d->epilogue == AMK_EPI_RESID) { if (ln) AMK_NT(EPI_RESID, true); else AMK_NT(EPI_RESID, false); }
    else if (d->epilogue == AMK_EPI_BIAS) { if (ln) AMK_NT(EPI_BIAS, true); else AMK_NT(EPI_BIAS, false); }
    else AMK_CHECK_ARG(false, "amk_gemm_f32: epilogue %d is not an NT epilogue", d->epilogue);
#undef AMK_NT
    AMK_CHECK_LAUNCH("amk_gemm_f32(NT)");
    return AMK_OK;
  }
  if (d->op == AMK_GEMM_NN) {
    const bool swb = d->epilogue == AMK_EPI_SWIGLU_BWD;
    if (two) AMK_CHECK_ARG(d->a2 && d->w2 && d->lda2 % 4 == 0 && d->ldw2 % 4 == 0, "amk_gemm_f32: second contraction segment incomplete");
    AMK_CHECK_SUPPORTED(d->n % 4 == 0, "amk_gemm_f32(NN): N must be a multiple of 4");
    AMK_CHECK_SUPPORTED(128 * d->lda * 4 < lim && (int64_t)d->k * d->ldw * 4 < lim && 128 * d->ldc * 4 < lim &&
                        (!two || (128 * d->lda2 * 4 < lim && (int64_t)d->k * d->ldw2 * 4 < lim)),
                        "amk_gemm_f32: operand panel beyond 1 GiB");
    if (swb) AMK_CHECK_ARG(d->ab && d->ldab % 4 == 0 && 128 * d->ldab * 4 < lim, "amk_gemm_f32: SwiGLU backward epilogue needs (a | b)");
    else AMK_CHECK_ARG(d->epilogue == AMK_EPI_BIAS && !d->bias, "amk_gemm_f32(NN): epilogue must be AMK_EPI_BIAS without a bias, or AMK_EPI_SWIGLU_BWD");
    p.ntn = (d->n + 127) / 128;
    const int64_t total = mt * p.ntn;
    AMK_CHECK_SUPPORTED(total < (int64_t)1 << 31, "amk_gemm_f32: grid too large");
    p.total = (int)total;
    const dim3 grid((unsigned)total);
    const bool b16 = dense_bk() == 16;
    if (swb) { if (b16) hipLaunchKernelGGL((gemm_nn_kernel<16, EPI_SWIGLU_BWD>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((gemm_nn_kernel<32, EPI_SWIGLU_BWD>), grid, dim3(256), 0, st, p); }
    else { if (b16) hipLaunchKernelGGL((gemm_nn_kernel<16, EPI_BIAS>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((gemm_nn_kernel<32, EPI_BIAS>), grid, dim3(256), 0, st, p); }
    AMK_CHECK_LAUNCH("amk_gemm_f32(NN)");
    return AMK_OK;
  }
  AMK_CHECK_ARG(d->op == AMK_GEMM_TN, "amk_gemm_f32: unknown op %d", d->op);
  if (two) AMK_CHECK_ARG(d->a2 && d->c2 && d->lda2 % 4 == 0 && d->ldc2 % 4 == 0, "amk_gemm_f32: second row segment incomplete");
  AMK_CHECK_SUPPORTED(d->n % 4 == 0 && (!two || (d->n - d->split) % 4 == 0), "amk_gemm_f32(TN): N must be a multiple of 4");
  int spc;
  const int chunks = tn_chunks(d, &spc);
  AMK_CHECK_SUPPORTED((int64_t)spc * 32 * d->lda * 4 < lim && (int64_t)spc * 32 * d->ldw * 4 < lim && (!two || (int64_t)spc * 32 * d->lda2 * 4 < lim),
                      "amk_gemm_f32(TN): chunk panel beyond 1 GiB");
  AMK_CHECK_SUPPORTED((int64_t)d->n * d->k * 4 < lim, "amk_gemm_f32(TN): output beyond 1 GiB");
  if (chunks > 1) {
    AMK_CHECK_ARG(workspace && ws_bytes >= amk_gemm_f32_ws_bytes(d), "amk_gemm_f32(TN): workspace of amk_gemm_f32_ws_bytes() bytes required");
    AMK_CHECK_ARG(a16(workspace), "amk_gemm_f32(TN): workspace must be 16-byte aligned");
    p.ws = static_cast<float*>(workspace);
  }
  p.nchunk = chunks;
  p.steps_per_chunk = spc;
  p.ntn = (d->k + 127) / 128;
  float* dbias_out = d->dbias;
  p.dbias = d->dbias ? (chunks > 1 ? p.ws + (int64_t)chunks * d->n * d->k : d->dbias) : nullptr;
  const int64_t total = (int64_t)((d->n + 127) / 128) * p.ntn * chunks;
  if (two) AMK_CHECK_ARG(d->split % 128 == 0, "amk_gemm_f32(TN): split must be a multiple of 128");
  p.total = (int)total;
  {
    const dim3 grid((unsigned)total);
    const bool b16 = dense_bk() == 16;
    if (ln) { if (b16) hipLaunchKernelGGL((gemm_tn_kernel<16, true>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((gemm_tn_kernel<32, true>), grid, dim3(256), 0, st, p); }
    else { if (b16) hipLaunchKernelGGL((gemm_tn_kernel<16, false>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((gemm_tn_kernel<32, false>), grid, dim3(256), 0, st, p); }
  }
  AMK_CHECK_LAUNCH("amk_gemm_f32(TN)");
  if (chunks > 1) {
    const int64_t items = (int64_t)d->n * d->k / 4;
    const int64_t blocks = (items + 255) / 256;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, dbias_out);
    AMK_CHECK_LAUNCH("amk_gemm_f32(TN reduce)");
  }
  return AMK_OK;
}

// diagnostic (not part of the ABI): resident workgroups per CU the runtime computes for the NT kernel
extern "C" int amk_debug_dense_occupancy(int bk) {
  int n = -1;
  hipError_t e;
  if (bk == 16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_nt_kernel<16, EPI_BIAS, false>, 256, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_nt_kernel<32, EPI_BIAS, false>, 256, 0);
  return e == hipSuccess ? n : -(int)e;
}
