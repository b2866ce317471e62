// Weight (and bias) gradient of nn.Linear in the mixed-precision mode: dW[n, k] = sum_m dY[m, n] X[m, k] with dY
// (M, N) and X (M, K) in bf16 as the autocast GEMMs leave them, dW and db in f32 as the parameters' gradients are
// kept (reference: the backward of nn.Linear in models/softmax_attention.py:30-42,80 and models/vitvqgan.py:20-34
// under cfg/vitvqgan.yaml:73).  The vendor library runs these products -- tiny outputs, a contraction over all
// B*T rows -- at 70-300 TFLOP/s (90-150 us each at batch 32) and the bias gradients as separate reductions; they are
// HBM-bound: every byte of dY and X has to be read once and that is all.
//   tile 128 (n) x 256 (k), eight waves as 2 x 4 of 64 x 64 (128 x 128, four waves, for K < 256), v_mfma_f32_32x32x16_bf16;
//   the M rows in chunks so that tiles x chunks is one workgroup per CU; partial tiles (f32) through a workspace, summed in chunk order
//   (bitwise reproducible); both operands are staged AS STORED -- [32 rows][128 columns] bf16, row stride 320 B -- and
//   both MFMA operands (whose contraction index is the row) come from ds_read_b64_tr_b16, 4 rows x 64 B per
//   half-wave on 64 distinct banks; db = column sums of the staged dY pieces (first k tile only).
#include "amk_common.h"
#include <stdlib.h>

namespace amk_gemm16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int STR = 160;     // bf16 per LDS row: 128 + 32 (80 dwords = 16 mod 64: conflict-free transposing reads)
constexpr unsigned COL_PAST = 0x40000000u;

struct Params {
  const __bf16 *y, *x;
  float *c, *dbias, *ws, *dbias_ws;
  int64_t M, ldy, ldx, ldc;
  int N, K, ntk, total, nchunk, steps_per_chunk;
};

// barrier that orders LDS only: global loads and stores in flight stay in flight (__syncthreads() drains them, which
// would serialise the register prefetch of the step loops and the row stores of the persistent kernel's epilogue)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ bf16x4 tr_read(const __bf16* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
// 32x32x16 operand whose contraction index is the ROW of the LDS tile, natural order: lane (col = c0 + (l & 31), half)
// gets rows r0 + 8 half + (0..7)
template <int STRIDE = STR>
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* img, int r0, int c0, int lane) {
  const int hf = lane >> 5, grp = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  const __bf16* a = img + (r0 + 8 * hf + q) * STRIDE + c0 + 16 * grp + 4 * pp;
  const bf16x4 lo = tr_read(a), hi = tr_read(a + 4 * STRIDE);
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = lo[i]; r[4 + i] = hi[i]; }
  return r;
}

// NWN x NWK waves of 64 x 64: tile 64 NWN (n) x 64 NWK (k); 64 rows of the contraction per step
template <int NWN, int NWK>
__global__ __launch_bounds__(64 * NWN * NWK, 2) void gemm_tn_bf16_kernel(Params p) {
  constexpr int BKM = 64, NTH = 64 * NWN * NWK, TN = 64 * NWN, TK = 64 * NWK;
  constexpr int YSTR = TN + 32, XSTR = TK + 32;          // (TN / 2 + 16 dwords = 16 mod 64: conflict-free transposing reads)
  constexpr int YT = BKM * YSTR, STAGE = YT + BKM * XSTR;  // elements of the Y tile / of one stage {Y tile, X tile}
  constexpr int YCG = TN / 8, YRP = NTH / YCG, YNP = BKM / YRP;   // Y: 16-byte column groups, rows per pass, pieces per thread
  constexpr int XCG = TK / 8, XRP = NTH / XCG, XNP = BKM / XRP;
  static_assert(TN % 128 == 0 && TK % 128 == 0 && YNP >= 1 && XNP >= 1, "tile shape");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];   // 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  const int wn = wave / NWK, wk = wave % NWK;
  const int u = xcd_remap(blockIdx.x, p.total);
  // unit order (n tile, chunk, k tile): the k tiles that read the same dY panel are neighbours (one XCD, one L2)
  const int tk = u % p.ntk, rest = u / p.ntk;
  const int tn = rest / p.nchunk, chunk = rest - tn * p.nchunk;
  const int n0 = tn * TN, k0 = tk * TK;
  const int64_t mbeg = (int64_t)chunk * p.steps_per_chunk * BKM;
  int64_t mend = mbeg + (int64_t)p.steps_per_chunk * BKM;
  if (mend > p.M) mend = p.M;
  const int mrows = (int)(mend - mbeg);
  // an even number of steps (a step of zeros at the end if need be): the two register sets then alternate without a branch
  // in the loop -- behind a branch around memory instructions the compiler's waits stop counting and drain everything
  const int nk = ((mrows + BKM - 1) / BKM + 1) & ~1;

  // staging: thread -> column group (8 columns) of each operand tile, rows sr + RP i
  const int ycg = tid % YCG, ysr = tid / YCG, xcg = tid % XCG, xsr = tid / XCG;
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + mbeg * p.ldy), 0, (int)(((int64_t)(mrows - 1) * p.ldy + p.N) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + mbeg * p.ldx), 0, (int)(((int64_t)(mrows - 1) * p.ldx + p.K) * 2), 0x00020000);
  const bool yok = n0 + 8 * ycg < p.N, xok = k0 + 8 * xcg < p.K;   // (N and K multiples of 8: a piece is in or out)
  unsigned yoff[YNP], xoff[XNP];
#pragma unroll
  for (int i = 0; i < YNP; ++i) yoff[i] = yok ? (unsigned)(((int64_t)(ysr + YRP * i) * p.ldy + n0 + 8 * ycg) * 2) : COL_PAST;
#pragma unroll
  for (int i = 0; i < XNP; ++i) xoff[i] = xok ? (unsigned)(((int64_t)(xsr + XRP * i) * p.ldx + k0 + 8 * xcg) * 2) : COL_PAST;
  const unsigned ystep = (unsigned)(BKM * p.ldy * 2), xstep = (unsigned)(BKM * p.ldx * 2);
  struct Stg { float4 y[YNP], x[XNP]; };
  auto gload = [&](Stg& g, int t) {   // (steps past the chunk: rows past the descriptor's range, zeros)
#pragma unroll
    for (int i = 0; i < YNP; ++i)
      g.y[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, (int)(yoff[i] == COL_PAST ? COL_PAST : yoff[i] + (unsigned)t * ystep), 0, 0));
#pragma unroll
    for (int i = 0; i < XNP; ++i)
      g.x[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(xoff[i] == COL_PAST ? COL_PAST : xoff[i] + (unsigned)t * xstep), 0, 0));
  };
  const bool do_bias = p.dbias != nullptr && tk == 0;
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  auto lstore = [&](__bf16* stage, const Stg& g) {
#pragma unroll
    for (int i = 0; i < YNP; ++i) {
      *reinterpret_cast<float4*>(&stage[(ysr + YRP * i) * YSTR + 8 * ycg]) = g.y[i];
      if (do_bias) {   // (steps past the chunk carry zeros)
        const bf16x8 v = __builtin_bit_cast(bf16x8, g.y[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += (float)v[j];
      }
    }
#pragma unroll
    for (int i = 0; i < XNP; ++i) *reinterpret_cast<float4*>(&stage[YT + (xsr + XRP * i) * XSTR + 8 * xcg]) = g.x[i];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  // two tiles in flight in registers (sets ga, gb) beside the one being staged: a step is a few hundred cycles, HBM
  // latency several thousand
  Stg ga, gb;
  gload(ga, 0);
  gload(gb, 1);
  lstore(smem, ga);
  gload(ga, 2);
  lds_barrier();
  auto step = [&](int t, Stg& g) {
    const __bf16* cur = smem + (t & 1) * STAGE;
    lstore(smem + ((t + 1) & 1) * STAGE, g);   // tile t+1 (loaded two steps ago) -> the other stage; then fetch tile t+3
    gload(g, t + 3);
#pragma unroll
    for (int s = 0; s < BKM / 16; ++s) {
      const bf16x8 a0 = tr_frag<YSTR>(cur, 16 * s, 64 * wn, lane), a1 = tr_frag<YSTR>(cur, 16 * s, 64 * wn + 32, lane);
      const bf16x8 b0 = tr_frag<XSTR>(cur + YT, 16 * s, 64 * wk, lane), b1 = tr_frag<XSTR>(cur + YT, 16 * s, 64 * wk + 32, lane);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    lds_barrier();
  };
  for (int t = 0; t < nk; t += 2) {
    step(t, gb);
    step(t + 1, ga);
  }
  // ---- the partial tile: rows n0 + 64 wn + 32 i + (r & 3) + 8 (r >> 2) + 4 hf, column k0 + 64 wk + 32 j + ln
  float* Cb = p.nchunk > 1 ? p.ws + (int64_t)chunk * p.N * p.K : p.c;
  const int64_t ldc = p.nchunk > 1 ? p.K : p.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + 64 * wk + 32 * j + ln;
      if (kc < p.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + 64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf;
          if (n < p.N) Cb[(int64_t)n * ldc + kc] = acc[i][j][r];
        }
      }
    }
  }
  if (do_bias) {   // fold the YRP row groups of each column group
    float* red = reinterpret_cast<float*>(smem);   // (the loop ended with a barrier)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ysr * TN + 8 * ycg + j] = bsum[j];
    lds_barrier();
    if (tid < TN) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < YRP; ++r) s += red[r * TN + tid];
      const int n = n0 + tid;
      if (n < p.N) (p.nchunk > 1 ? p.dbias_ws + (int64_t)chunk * p.N : p.dbias)[n] = s;
    }
  }
}

__global__ __launch_bounds__(256) void tn_bf16_reduce_kernel(Params p) {
  const int64_t total4 = (int64_t)p.N * p.K / 4, slab = (int64_t)p.N * p.K;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total4) {
    const float* src = p.ws + 4 * i;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = 0;
    for (; c + 8 <= p.nchunk; c += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(src + (c + j) * slab);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    for (; c < p.nchunk; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(src + c * slab);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int64_t e = 4 * i, row = e / p.K;
    *reinterpret_cast<float4*>(p.c + row * p.ldc + (e - row * p.K)) = s;
  }
  if (p.dbias && i < p.N) {   // thread n also folds the bias partials of column n (the launch has at least N threads)
    float s = 0.f;
    for (int c = 0; c < p.nchunk; ++c) s += p.dbias_ws[(int64_t)c * p.N + i];
    p.dbias[i] = s;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Forward and input gradient of nn.Linear in the mixed-precision mode: C = A W^T (+ bias) [NT: W (N, K) as stored] and
// C = A W [NN: dX = dY W, W (K, N) as stored], A (M, K) and C (M, N) bf16, f32 accumulation, the bias (f32) added before
// the one rounding.  EPI 1 (NT only): the SwiGLU gate folded in -- the tile holds 64 gate columns j and the matching 64
// value columns H + j, and G[m, j] = silu(a) b leaves with (or instead of) the (a | b) tile.  EPI 2 (NN only): the
// gate's backward folded in -- the product is dG (M, H), and what leaves is (dA | dB) (M, 2 H) from the forward's (a | b):
// dG is rounded to bf16 as the unfused pair of launches would hand it over, then combined in the row pass.
//   tile 128 x 128, four waves as 2 x 2 of 64 x 64, contraction in steps of 32, two tiles of the step stream in flight
//   in registers beside the one in LDS, four workgroups per CU (K is 256 for most of these products: a tile is eight
//   steps, so its prologue and epilogue are hidden by the CU's other workgroups, not by its own loop).
//   The product is formed TRANSPOSED (W fragments as the MFMA's row operand), so a lane ends with four consecutive
//   output columns of one row; the tile is then turned through LDS into 16-byte row pieces.
#ifndef G16_ABLATE
#define G16_ABLATE 0   // timing experiments only (tools/ablate_gemm_bf16.sh): 1 no epilogue, 2 no MFMAs, 4 no operand reads from LDS
#endif
constexpr int FSTR = 32;            // bf16 per LDS row of a [128 rows][32 k] operand tile: no padding, the four 16-byte
                                    // chunks of row r sit at chunk ^ ((r >> 2) & 3) (writes and b128 reads conflict-free)
constexpr int FT = 32 * STR;        // elements per operand tile buffer (the [32 k][128 n] tile of the NN form; >= 128 * FSTR)
constexpr int OSTR = 136;           // bf16 per row of the output tile in LDS

struct FParams {
  const __bf16 *a, *w, *ab;
  const float* bias;
  __bf16 *c, *g;
  int64_t M, lda, ldw, ldc, ldg, ldab;
  int N, K, H, ntn, total;
};

template <bool BTR, int EPI>
__global__ __launch_bounds__(256, 4) void gemm_bf16_kernel(FParams p) {
  static_assert(FT >= 128 * FSTR, "the two B tile shapes share one buffer");
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 2 * FT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  const int u = xcd_remap(blockIdx.x, p.total);
  const int tm = u / p.ntn, tn = u - tm * p.ntn;   // neighbours share the A panel
  const int64_t m0 = (int64_t)tm * 128;
  const int n0 = EPI == 1 ? tn * 64 : tn * 128;    // (SwiGLU: first gate column)
  // local output column (0..127) -> column of C; 8-column groups stay together
  auto ncol = [&](int loc) {   // (>= N: not a column of this problem)
    if (EPI != 1) return n0 + loc;
    const int j = n0 + 32 * (loc >> 6) + (loc & 31);
    return j < p.H ? ((loc >> 5) & 1) * p.H + j : p.N;
  };
  const int nk = ((p.K + 31) / 32 + 1) & ~1;   // even (see the weight-gradient kernel): a step of zeros at the end if need be
  const int mrows = (int)(p.M - m0 < 128 ? p.M - m0 : 128);
  // ---- staging
  const int ar = tid >> 2, ach = tid & 3;          // A (and NT W): pieces (row ar + 64 i, 8 k at 8 ach)
  const int cg = tid & 15, sr = tid >> 4;          // NN W: pieces (k row sr + 16 i, 8 columns at 8 cg)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + m0 * p.lda), 0, (int)(((int64_t)(mrows - 1) * p.lda + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w, 0, (int)((BTR ? (int64_t)(p.K - 1) * p.ldw + p.N : (int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  unsigned aoff[2], woff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aoff[i] = (unsigned)(((int64_t)(ar + 64 * i) * p.lda + 8 * ach) * 2);
    if (BTR) {
      const int col = n0 + 8 * cg;
      woff[i] = col < p.N ? (unsigned)(((int64_t)(sr + 16 * i) * p.ldw + col) * 2) : COL_PAST;
    } else {
      const int n = ncol(ar + 64 * i);
      woff[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldw + 8 * ach) * 2) : COL_PAST;
    }
  }
  const unsigned wstep = BTR ? (unsigned)(32 * p.ldw * 2) : 64u;
  struct Stg { float4 a[2], w[2]; };
  auto gload = [&](Stg& g, int t) {
    const bool kin = 32 * t + 8 * ach < p.K;   // (K a multiple of 8: a piece is in or out)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool win = BTR ? (32 * t + sr + 16 * i < p.K && woff[i] != COL_PAST) : (kin && woff[i] != COL_PAST);
      g.a[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)(kin ? aoff[i] + (unsigned)t * 64u : COL_PAST), 0, 0));
      g.w[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(win ? woff[i] + (unsigned)t * wstep : COL_PAST), 0, 0));
    }
  };
  auto lstore = [&](__bf16* stage, const Stg& g) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sw = 8 * (ach ^ ((ar >> 2) & 3));   // (rows ar and ar + 64: the same swizzle)
      *reinterpret_cast<float4*>(&stage[(ar + 64 * i) * FSTR + sw]) = g.a[i];
      if (BTR) *reinterpret_cast<float4*>(&stage[FT + (sr + 16 * i) * STR + 8 * cg]) = g.w[i];
      else *reinterpret_cast<float4*>(&stage[FT + (ar + 64 * i) * FSTR + sw]) = g.w[i];
    }
  };
  f32x16 acc[2][2];   // [n block][m block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  Stg ga, gb;
  gload(ga, 0);
  gload(gb, 1);
  lstore(smem, ga);
  gload(ga, 2);
  lds_barrier();
  auto step = [&](int t, Stg& g) {
    const __bf16* cur = smem + (t & 1) * 2 * FT;
    lstore(smem + ((t + 1) & 1) * 2 * FT, g);
    gload(g, t + 3);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 xf[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (G16_ABLATE & 4) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { xf[i][e] = (__bf16)(float)(tid + e); wf[i][e] = (__bf16)(float)(lane - e); }
          continue;
        }
        const int sw = 8 * ((2 * s + hf) ^ ((ln >> 2) & 3));   // (row offsets 64 w + 32 i: multiples of 16, swizzle of ln)
        xf[i] = *reinterpret_cast<const bf16x8*>(&cur[(64 * wm + 32 * i + ln) * FSTR + sw]);
        wf[i] = BTR ? tr_frag(cur + FT, 16 * s, 64 * wn + 32 * i, lane)
                    : *reinterpret_cast<const bf16x8*>(&cur[FT + (64 * wn + 32 * i + ln) * FSTR + sw]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (G16_ABLATE & 2) acc[i][j][0] += (float)wf[i][0] * (float)xf[j][1];
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    lds_barrier();
  };
  for (int t = 0; t < nk; t += 2) {
    step(t, gb);
    step(t + 1, ga);
  }
  if (G16_ABLATE & 1) {
    if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.f) p.c[0] = (__bf16)1.f;
    return;
  }
  // ---- epilogue: acc[i][j][r] = C[m0 + 64 wm + 32 j + ln][ncol(64 wn + 32 i + (r & 3) + 8 (r >> 2) + 4 hf)]
  if (p.bias) {   // all eight requests first (clamped addresses, no branch): one memory latency, not eight
    float4 b4[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = ncol(64 * wn + 32 * i + 8 * g + 4 * hf);
        b4[i][g] = *reinterpret_cast<const float4*>(p.bias + (n < p.N ? n : 0));
        if (n >= p.N) b4[i][g] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][4 * g] += b4[i][g].x; acc[i][j][4 * g + 1] += b4[i][g].y; acc[i][j][4 * g + 2] += b4[i][g].z; acc[i][j][4 * g + 3] += b4[i][g].w;
        }
  }
  __bf16* ot = smem;   // [128][OSTR] (the loop ended with a barrier)
  auto put_tile = [&](int nblocks) {   // the accumulators (n blocks 0..nblocks-1) as bf16 into the LDS tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= nblocks) break;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (__bf16)acc[i][j][4 * g + e];
          const int loc = nblocks == 2 ? 64 * wn + 32 * i + 8 * g + 4 * hf : 32 * wn + 8 * g + 4 * hf;
          *reinterpret_cast<bf16x4*>(&ot[(64 * wm + 32 * j + ln) * OSTR + loc]) = v;
        }
    }
  };
  if (EPI == 0 || (EPI == 1 && p.c)) {
    put_tile(2);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pc = tid + 256 * i, row = pc >> 4, loc = 8 * (pc & 15);
      const int n = ncol(loc);
      if (row < mrows && n < p.N) *reinterpret_cast<float4*>(p.c + (m0 + row) * p.ldc + n) = *reinterpret_cast<const float4*>(&ot[row * OSTR + loc]);
    }
  }
  if (EPI == 2) {
    put_tile(2);
    lds_barrier();
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
      const int pc = tid + 256 * i, row = pc >> 4, loc = 8 * (pc & 15);
      const int j = n0 + loc;
      if (row < mrows && j < p.N) {
        const bf16x8 dg = *reinterpret_cast<const bf16x8*>(&ot[row * OSTR + loc]);
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(p.ab + (m0 + row) * p.ldab + j);
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(p.ab + (m0 + row) * p.ldab + p.N + j);
        bf16x8 da, db;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = (float)av[e], g = (float)dg[e];
          const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-a * AMK_LOG2E));
          da[e] = (__bf16)(g * (float)bv[e] * (sg * (1.f + a * (1.f - sg))));
          db[e] = (__bf16)(g * (a * sg));
        }
        *reinterpret_cast<bf16x8*>(p.c + (m0 + row) * p.ldc + j) = da;
        *reinterpret_cast<bf16x8*>(p.c + (m0 + row) * p.ldc + p.N + j) = db;
      }
    }
  }
  if (EPI == 1) {
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = acc[0][j][r];
        acc[0][j][r] = a * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-a * AMK_LOG2E)) * acc[1][j][r];
      }
    put_tile(1);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = tid + 256 * i, row = pc >> 3, loc = 8 * (pc & 7);
      const int jcol = n0 + 32 * (loc >> 5) + (loc & 31);
      if (row < mrows && jcol < p.H) *reinterpret_cast<float4*>(p.g + (m0 + row) * p.ldg + jcol) = *reinterpret_cast<const float4*>(&ot[row * OSTR + loc]);
    }
  }
}


}  // namespace amk_gemm16

using namespace amk_gemm16;

// tile width along K: 256 (eight waves as 2 x 4: dY is staged once for a whole K = 256, and a CU holds two waves per SIMD at
// one workgroup per CU) for the gradients with many tiles, else 128 (four waves): measured at the ViT shapes, 256 / 128:
// dW12 (2736 x 256) 72 / 95 us, dW3 (256 x 1368) 49 / 51, dWkv (1024 x 256) 43 / 40, dWq 29 / 26, dWo (256 x 512) 44 / 34 --
// with few tiles the chunks get short and the eight-wave workgroup's prologue and epilogue show.  AMK_TN16_TK overrides.
static int tile_k(int N, int K) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("AMK_TN16_TK");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 128 || forced == 256) return forced;
  return (K >= 256 && ((N + 127) / 128) * ((K + 255) / 256) >= 12) ? 256 : 128;
}

static int chunks_for(int64_t M, int N, int K, int* spc) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const int TK = tile_k(N, K);
  const int tiles = ((N + 127) / 128) * ((K + TK - 1) / TK);
  const int BKM = 64;
  const int64_t steps = (M + BKM - 1) / BKM;
  static int wgs = 0;
  if (wgs == 0) {
    const char* e = getenv("AMK_TN16_WGS");
    wgs = e && atoi(e) > 0 ? atoi(e) : 1;   // workgroups per CU the grid aims at (more: more partial tiles to write and re-read;
                                            // measured at the ViT shapes, 1 / 2 / 4 per CU: dW12 94 / 100 / 144 us, dWo 33 / 44 / 70 us)
  }
  int64_t chunks = ((int64_t)wgs * cus) / tiles;
  if (chunks < 1) chunks = 1;
  if (chunks > steps / 8) chunks = steps / 8 > 0 ? steps / 8 : 1;
  int64_t s = (steps + chunks - 1) / chunks;
  chunks = (steps + s - 1) / s;
  *spc = (int)s;
  return (int)chunks;
}

extern "C" int64_t amk_gemm_tn_bf16_ws_bytes(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int spc;
  const int chunks = chunks_for(M, N, K, &spc);
  return chunks > 1 ? ((int64_t)chunks * N * K + (int64_t)chunks * N) * 4 : 0;
}

extern "C" int amk_gemm_tn_bf16(const void* y, int64_t ldy, const void* x, int64_t ldx, float* c, int64_t ldc, float* dbias,
                                int64_t M, int N, int K, void* workspace, int64_t ws_bytes, void* stream) {
  AMK_CHECK_ARG(y && x && c, "amk_gemm_tn_bf16: null operand");
  AMK_CHECK_ARG(M > 0 && N > 0 && K > 0, "amk_gemm_tn_bf16: non-positive size");
  AMK_CHECK_SUPPORTED(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldc % 4 == 0,
                      "amk_gemm_tn_bf16: N, K and the bf16 leading dimensions must be multiples of 8, ldc of 4");
  AMK_CHECK_ARG(((uintptr_t)y & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)c & 15) == 0, "amk_gemm_tn_bf16: pointers must be 16-byte aligned");
  Params p = {};
  p.y = static_cast<const __bf16*>(y); p.x = static_cast<const __bf16*>(x); p.c = c; p.dbias = dbias;
  p.M = M; p.N = N; p.K = K; p.ldy = ldy; p.ldx = ldx; p.ldc = ldc;
  int spc;
  p.nchunk = chunks_for(M, N, K, &spc);
  p.steps_per_chunk = spc;
  const int BKM = 64;
  AMK_CHECK_SUPPORTED((int64_t)spc * BKM * ldy * 2 < (1ll << 30) && (int64_t)spc * BKM * ldx * 2 < (1ll << 30), "amk_gemm_tn_bf16: chunk panel beyond 1 GiB");
  if (p.nchunk > 1) {
    AMK_CHECK_ARG(workspace && ws_bytes >= amk_gemm_tn_bf16_ws_bytes(M, N, K) && ((uintptr_t)workspace & 15) == 0,
                  "amk_gemm_tn_bf16: workspace of amk_gemm_tn_bf16_ws_bytes() bytes (16-byte aligned) required");
    p.ws = static_cast<float*>(workspace);
    p.dbias_ws = p.ws + (int64_t)p.nchunk * N * K;
  }
  const int TK = tile_k(N, K);
  p.ntk = (K + TK - 1) / TK;
  const int64_t total = (int64_t)((N + 127) / 128) * p.ntk * p.nchunk;
  AMK_CHECK_SUPPORTED(total < (1ll << 31), "amk_gemm_tn_bf16: grid too large");
  p.total = (int)total;
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr size_t lds4 = (size_t)2 * 64 * ((128 + 32) + (128 + 32)) * 2, lds8 = (size_t)2 * 64 * ((128 + 32) + (256 + 32)) * 2;
  // the opt-in to > 64 KiB of dynamic LDS is per device: one flag per device ordinal, and a refusal is an error here, not
  // an opaque launch failure later
  static bool attr_set[64] = {};
  int dev_id = 0;
  AMK_CHECK_ARG(hipGetDevice(&dev_id) == hipSuccess && dev_id >= 0 && dev_id < 64, "amk_gemm_tn_bf16: no current device");
  if (!attr_set[dev_id]) {
    AMK_CHECK_SUPPORTED(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8) == hipSuccess,
                        "amk_gemm_tn_bf16: the device refused %zu bytes of dynamic LDS", lds8);
    attr_set[dev_id] = true;
  }
  if (TK == 256) hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 4>), dim3((unsigned)total), dim3(512), lds8, st, p);
  else hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 2>), dim3((unsigned)total), dim3(256), lds4, st, p);
  if (p.nchunk > 1) {
    int64_t blocks = ((int64_t)N * K / 4 + 255) / 256;
    if (blocks < (N + 255) / 256) blocks = (N + 255) / 256;   // (thread n also folds the bias partials of column n)
    hipLaunchKernelGGL(tn_bf16_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  }
  AMK_CHECK_LAUNCH("amk_gemm_tn_bf16");
  return AMK_OK;
}


// op 0: c = a w^T (+ bias), w (N, K); op 1: c = a w, w (K, N); epi 1 (op 0 only): w = (gate rows | value rows) (2 H, K),
// g (M, H) = silu(a w_gate^T + b_gate) * (a w_value^T + b_value), c (M, 2 H) optional
extern "C" int amk_gemm_bf16(int op, int epi, const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                             void* c, int64_t ldc, void* g, int64_t ldg, int64_t M, int N, int K, void* stream) {
  AMK_CHECK_ARG(a && w, "amk_gemm_bf16: null operand");
  AMK_CHECK_ARG((op == 0 || op == 1) && (epi == 0 || (epi == 1 && op == 0)), "amk_gemm_bf16: unknown op %d / epilogue %d", op, epi);
  AMK_CHECK_ARG(epi == 1 ? g != nullptr : c != nullptr, "amk_gemm_bf16: null output");
  AMK_CHECK_ARG(M > 0 && N > 0 && K > 0, "amk_gemm_bf16: non-positive size");
  AMK_CHECK_SUPPORTED(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && (!c || ldc % 8 == 0) && (epi == 0 || (ldg % 8 == 0 && N % 16 == 0)),
                      "amk_gemm_bf16: N, K and the leading dimensions must be multiples of 8 (N of 16 with the SwiGLU epilogue)");
  AMK_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)c & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)bias & 15) == 0,
                "amk_gemm_bf16: pointers must be 16-byte aligned");
  AMK_CHECK_SUPPORTED(128 * lda * 2 < (1ll << 30) && (op == 1 ? (int64_t)K * ldw : (int64_t)N * ldw) * 2 < (1ll << 30), "amk_gemm_bf16: operand panel beyond 1 GiB");
  FParams p = {};
  p.a = static_cast<const __bf16*>(a); p.w = static_cast<const __bf16*>(w); p.bias = bias;
  p.c = static_cast<__bf16*>(c); p.g = static_cast<__bf16*>(g);
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldg = ldg; p.H = N / 2;
  p.ntn = epi == 1 ? (p.H + 63) / 64 : (N + 127) / 128;
  const int64_t total = ((M + 127) / 128) * p.ntn;
  AMK_CHECK_SUPPORTED(total < (1ll << 31), "amk_gemm_bf16: grid too large");
  p.total = (int)total;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (op == 1) hipLaunchKernelGGL((gemm_bf16_kernel<true, 0>), dim3((unsigned)total), dim3(256), 0, st, p);
  else if (epi == 1) hipLaunchKernelGGL((gemm_bf16_kernel<false, 1>), dim3((unsigned)total), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false, 0>), dim3((unsigned)total), dim3(256), 0, st, p);
  AMK_CHECK_LAUNCH("amk_gemm_bf16");
  return AMK_OK;
}


// (dA | dB) (M, 2 H) = SwiGLU backward of dG = dy (M, K) w3 (K, H), with the forward's (a | b) (M, 2 H): the input gradient
// of the FFN's second projection and the gate's backward in one launch (dG never reaches HBM)
extern "C" int amk_gemm_bf16_swiglu_bwd(const void* dy, int64_t lddy, const void* w3, int64_t ldw, const void* ab, int64_t ldab,
                                        void* dab, int64_t lddab, int64_t M, int H, int K, void* stream) {
  AMK_CHECK_ARG(dy && w3 && ab && dab, "amk_gemm_bf16_swiglu_bwd: null operand");
  AMK_CHECK_ARG(M > 0 && H > 0 && K > 0, "amk_gemm_bf16_swiglu_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(H % 8 == 0 && K % 8 == 0 && lddy % 8 == 0 && ldw % 8 == 0 && ldab % 8 == 0 && lddab % 8 == 0,
                      "amk_gemm_bf16_swiglu_bwd: H, K and the leading dimensions must be multiples of 8");
  AMK_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)w3 & 15) == 0 && ((uintptr_t)ab & 15) == 0 && ((uintptr_t)dab & 15) == 0,
                "amk_gemm_bf16_swiglu_bwd: pointers must be 16-byte aligned");
  AMK_CHECK_SUPPORTED(128 * lddy * 2 < (1ll << 30) && (int64_t)K * ldw * 2 < (1ll << 30), "amk_gemm_bf16_swiglu_bwd: operand panel beyond 1 GiB");
  FParams p = {};
  p.a = static_cast<const __bf16*>(dy); p.w = static_cast<const __bf16*>(w3); p.ab = static_cast<const __bf16*>(ab);
  p.c = static_cast<__bf16*>(dab);
  p.M = M; p.N = H; p.K = K; p.lda = lddy; p.ldw = ldw; p.ldc = lddab; p.ldab = ldab; p.H = H;
  p.ntn = (H + 127) / 128;
  const int64_t total = ((M + 127) / 128) * p.ntn;
  AMK_CHECK_SUPPORTED(total < (1ll << 31), "amk_gemm_bf16_swiglu_bwd: grid too large");
  p.total = (int)total;
  hipLaunchKernelGGL((gemm_bf16_kernel<true, 2>), dim3((unsigned)total), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  AMK_CHECK_LAUNCH("amk_gemm_bf16_swiglu_bwd");
  return AMK_OK;
}
