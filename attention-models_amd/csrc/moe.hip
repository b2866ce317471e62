// Top-k expert routing and grouped expert GEMMs for gfx950 (exact-f32 MFMA).
//
// Replaces the per-expert Python loops of the reference -- MoELayer.forward
// (models/moe.py:23-38) and SwitchHeadAttention.moe_v / moe_out
// (models/switchhead_attention.py:58-88): for each expert a torch.where (host sync), a
// gather, a small GEMM and an index_put.  Here routing is three tiny launches with no host
// sync and every expert's GEMM runs in ONE launch:
//
//   route_topk     : per routed unit (a token, or a (token, head)), top-k of its E gate logits
//                    (descending, lowest index on ties = torch.topk on distinct values) and
//                    sigmoid of the selected logits.
//   route_local / route_scan / route_perm : block-local ranks + per-block histograms, per-expert scan
//                    of the histograms, then perm[offsets[e] + block base + rank] = pair -- pairs of an
//                    expert in ascending pair order (deterministic), no serial pass over the pairs.
//   grouped_nt     : Y[p,:]  = A[p/a_div,:] * W_e^T (+ b_e)          (forward of x W^T + b)
//   grouped_nn     : Y[p,:]  = s[p] * (A[p/a_div,:] * W_e)           (input gradient)
//   grouped_wgrad  : dW_e    = sum_{p in e} s[p] * G[p/a_div,:]^T (x) X[p/b_div,:]  (+ db_e)
//   combine        : out[g]  = sum_outer ( sum_{slots, ascending expert id} s[p] * Y[p] )
//   gate_grad      : dlogit[u, ids[p]] = sigmoid'(.) * <dOut[p/g_div], Y[p]>
//
// A "pair" p = unit*k + slot.  Two generations of GEMM kernels live here:
//   * grouped_{nt,nn}_wide_kernel, grouped_wgrad_wide_kernel (round 2; outputs >= 128 wide, weight gradients from
//     64 wide): ONE software pipeline -- 32-row blocks in tiles of 2..4 blocks (height chosen per launch by every
//     wave from the offsets) x 128 outputs x 32 deep, two LDS stages and one barrier per step, the tile movement
//     (register -> LDS of tile t+1, global -> register of tile t+2) between the MFMA groups of tile t, units decoded
//     by a wave scan, XCD-aware unit order, the last partial round as half-width units, range-checked buffer
//     loads / stores instead of compares.  DESIGN.md section 4 has the measurements behind each of these.
//   * grouped_{nt,nn,wgrad}_kernel<NB> (round 1): 64 pairs x 64 NB outputs x 32 deep, two barriers per step; still
//     used for outputs narrower than 128 (SwitchHead's 64-wide experts) and depths that are not multiples of 32.
// All of them: 4 waves, 32x32 f32 accumulators with the OUTPUT FEATURE ON THE LANE (coalesced 128-B row segments
// on store, per-lane bias), operands through LDS with row stride 36 floats (conflict-free ds_read_b128 row reads,
// ds_read_b32 column reads).
#include "amk_common.h"
#include <stdlib.h>

namespace amk_moe {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float f4(const float4& v, int e) {
  return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

constexpr int MAX_K = 8;

// ---------------------------------------------------------------------------------------
template <int KK>
__global__ __launch_bounds__(128) void route_topk_kernel(const float* __restrict__ logits, int64_t U, int E,
                                                         int64_t* __restrict__ ids, float* __restrict__ gate) {
  // A block takes 128 consecutive units = 128*E CONTIGUOUS logits; they are staged through LDS in
  // chunks of 32 experts by coalesced loads (consecutive threads read consecutive floats of a unit's
  // row; a thread reading its own row directly would touch 64 rows per instruction), then every thread
  // merges its unit's 32 values into a sorted top-k list: descending value, lowest index on ties --
  // the order torch.topk gives on distinct values.  k is a template parameter: the list lives in KK
  // registers and an insertion is KK compares (with a run-time k and an 8-entry list the kernel was 21 us
  // for the 33280 units of a SwitchHead layer).
  constexpr int EC = 32, UB = 128;
  __shared__ float tile[UB * (EC + 1)];
  const int tid = threadIdx.x;
  const int64_t u0 = (int64_t)blockIdx.x * UB;
  const int64_t u = u0 + tid;
  float bv[KK];
  int bi[KK];
#pragma unroll
  for (int s = 0; s < KK; ++s) { bv[s] = -INFINITY; bi[s] = -1; }
  const int nu = (int)min((int64_t)UB, U - u0);
  for (int e0 = 0; e0 < E; e0 += EC) {
    const int ne = min(EC, E - e0);
    __syncthreads();
    for (int f = tid; f < nu * ne; f += UB) {  // f -> (unit f / ne, expert f % ne): ne consecutive floats per row
      const int r = f / ne, c = f % ne;
      tile[r * (EC + 1) + c] = logits[(u0 + r) * E + e0 + c];
    }
    __syncthreads();
    if (u < U) {
      for (int c = 0; c < ne; ++c) {
        const float v = tile[tid * (EC + 1) + c];
        // insert (v, e0 + c) behind every entry with value >= v (equal values: the earlier index stays first)
        if (bi[KK - 1] < 0 || v > bv[KK - 1]) {
          int pos = KK - 1;
#pragma unroll
          for (int s = KK - 2; s >= 0; --s) {
            if (bi[s] < 0 || v > bv[s]) { bv[s + 1] = bv[s]; bi[s + 1] = bi[s]; pos = s; }
          }
#pragma unroll
          for (int s = 0; s < KK; ++s)
            if (s == pos) { bv[s] = v; bi[s] = e0 + c; }
        }
      }
    }
  }
  if (u < U) {
#pragma unroll
    for (int s = 0; s < KK; ++s) {
      ids[u * KK + s] = bi[s];
      gate[u * KK + s] = 1.f / (1.f + expf(-bv[s]));
    }
  }
}

// Deterministic expert-major ordering in three small launches (no serial scan over the pairs):
//   route_local : a block takes 1024 consecutive pairs; rank of a pair among the EARLIER pairs of the
//                 same expert inside the block (LDS broadcast compare loop) + the block's histogram.
//   route_scan  : per expert, exclusive scan of the block histograms -> block bases, counts, offsets.
//   route_perm  : perm[offsets[e] + base[block][e] + local rank] = pair.
// Pairs of one expert therefore appear in ascending pair order, whatever the launch timing.
constexpr int ROUTE_BLOCK = 1024;  // pairs per block of route_local / route_perm (16 waves): 65 blocks at a SwitchHead layer

__global__ __launch_bounds__(ROUTE_BLOCK) void route_local_kernel(const int64_t* __restrict__ ids, int64_t P, int E,
                                                                 int32_t* __restrict__ rank, int32_t* __restrict__ blockhist) {
  // inside a wave: one ballot per DISTINCT expert present (at most min(E, 64) rounds) gives every lane the set of
  // lanes with its expert -- rank among the earlier ones = popcount below the lane; across the sixteen waves the
  // per-wave counts go through LDS.  (A compare loop over the pairs of a 256-pair block was 8 us per launch, and
  // 260 blocks made the scan kernel's walk long.)
  constexpr int NW = ROUTE_BLOCK / 64;
  extern __shared__ int wcount[];  // [NW][E]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t p = (int64_t)blockIdx.x * ROUTE_BLOCK + tid;
  const int my = (p < P) ? (int)ids[p] : -1;
  for (int i = tid; i < NW * E; i += ROUTE_BLOCK) wcount[i] = 0;
  __syncthreads();
  unsigned long long remaining = __ballot(my >= 0), mine = 0;
  while (remaining) {
    const int leader = __builtin_ctzll(remaining);
    const int el = __shfl(my, leader, 64);
    const unsigned long long same = __ballot(my == el);
    if (my == el) mine = same;
    if (lane == leader) wcount[wave * E + el] = __builtin_popcountll(same);
    remaining &= ~same;
  }
  __syncthreads();
  if (my >= 0) {
    int r = __builtin_popcountll(mine & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) r += wcount[w * E + my];
    rank[p] = r;
  }
  int32_t* bh = blockhist + (int64_t)blockIdx.x * E;
  for (int e = tid; e < E; e += ROUTE_BLOCK) {
    int c = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) c += wcount[w * E + e];
    bh[e] = c;
  }
}

// One wave per expert (16 waves, experts dealt round-robin): the per-block counts of an expert are
// turned into exclusive bases 64 blocks at a time with a wave prefix sum (a serial walk over the
// 260 blocks of a ViTMoE layer cost 60 us); wave 0 then scans the expert totals into offsets.
__global__ __launch_bounds__(1024) void route_scan_kernel(int32_t* __restrict__ blockhist, int nblk, int E,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ offsets) {
  __shared__ int total[1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = wave; e < E; e += 16) {
    int running = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
      const int b = b0 + lane;
      const int c = b < nblk ? blockhist[(int64_t)b * E + e] : 0;
      int x = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
      }
      if (b < nblk) blockhist[(int64_t)b * E + e] = running + x - c;  // exclusive: base of this block inside the expert
      running += __shfl(x, 63, 64);
    }
    if (lane == 0) { counts[e] = running; total[e] = running; }
  }
  __syncthreads();
  if (wave == 0) {
    int running = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + lane;
      const int c = e < E ? total[e] : 0;
      int x = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
      }
      if (e < E) offsets[e] = running + x - c;
      running += __shfl(x, 63, 64);
    }
    if (lane == 0) offsets[E] = running;
  }
}

__global__ __launch_bounds__(256) void route_perm_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ blockbase, const int32_t* __restrict__ rank,
                                                         int64_t P, int E, int32_t* __restrict__ perm) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int e = (int)ids[p];
  perm[offsets[e] + blockbase[(p / ROUTE_BLOCK) * E + e] + rank[p]] = (int)p;
}

// ---------------------------------------------------------------------------------------
struct GemmParams {
  const float* A;        // rows addressed A + (p / a_div) * lda
  const float* B2;       // wgrad only: rows addressed B2 + (p / b_div) * ldb
  const float* W;        // (E, N, Kd) contiguous
  const float* bias;     // (E, N) or null            (nt)
  const float* scale;    // (P) or null               (nn, wgrad)
  float* Y;              // (P, N) for nt, (P, Kd) for nn; dW (E, N, Kd) for wgrad
  float* dbias;          // (E, N) or null            (wgrad)
  const int32_t* offsets;
  const int32_t* perm;
  int E, N, Kd;
  int ncol;              // output tiles per row tile (nt, nn: 1-D grid of units)
  int slots;             // workgroups the chip holds at once (2 per CU): tile-height choice of the wide kernels
  int a_div, b_div;
  int64_t lda, ldb;
  int64_t a_bytes;       // size of the A buffer (wide kernels: buffer-descriptor range)
  int64_t y_bytes;       // size of the Y buffer (wide kernels)
  int64_t b_bytes, s_bytes;  // wgrad: sizes of the X and scale buffers
  int a_shift, b_shift;  // log2 of a_div / b_div when they are powers of two, else -1
  int y_div;             // wide nt / nn: > 0 = accumulate mode, pair p ADDS into output row p / y_div (f32 atomics)
  int rsplit;            // wide wgrad: 2 = an expert's pairs cut in two, both halves ADD into a zeroed dW (0 + a + b: order-free)
};

// Locate this workgroup's unit = (expert, output tile, 64-pair row tile).  The grid is 1-D over the units in
// the order [expert][output tile][row tile]: the row tiles that read one weight panel W[e, tile, :] are
// adjacent, and the ids are remapped (xcd_remap over the ACTUAL number of units -- the grid is an upper bound,
// the surplus workgroups leave) so that adjacent units run on one XCD, behind one L2: as a 2-D grid the five
// row tiles of a panel were dealt to five XCDs and each fetched the panel from HBM (PMC: 693 MB read per
// forward for 185 MB of operands).
template <int TR = 64>   // pairs per tile
__device__ __forceinline__ bool find_unit(const int32_t* offsets, int E, int ncol, int bid, int& e, int& m0, int& cnt, int& ct) {
  int T = 0;
  for (int j = 0; j < E; ++j) T += (offsets[j + 1] - offsets[j] + TR - 1) / TR;
  const int total = T * ncol;
  if (bid >= total) return false;
  int u = xcd_remap(bid, total);
  for (int j = 0; j < E; ++j) {
    const int c = offsets[j + 1] - offsets[j];
    const int nt = (c + TR - 1) / TR;
    const int blk = nt * ncol;
    if (u < blk) { e = j; ct = u / nt; m0 = (u - ct * nt) * TR; cnt = c; return true; }
    u -= blk;
  }
  return false;
}

// All three GEMM kernels are templated on NB: every wave keeps NB 32x32 accumulators that share one
// operand fragment (NB = 2 halves the LDS reads, barriers and staging per MFMA; NB = 1 serves the
// 64-wide outputs of SwitchHead's V experts).

// Y[p, n] = sum_kk A[arow(p), kk] * W[e, n, kk] (+ bias[e, n])        tile: 64 pairs x 64*NB outputs
// SHORT (NB 1): 32 pairs x 64 outputs; the waves are 2 (output halves) x 2 (halves of every k-slab) and the two k-halves
// are added through LDS at the end, in a fixed order.  For launches of a few workgroups per CU -- SwitchHead's 64-wide
// experts at batch 64 are 840 tiles of 64 pairs on 256 CUs, all resident at once, and the kernel lasts as long as the CUs
// that drew four of them -- twice as many, half as long workgroups even the CUs out and cover each other's waits.
template <int NB, bool SHORT = false>
__global__ __launch_bounds__(256, 2) void grouped_nt_kernel(GemmParams g) {
  static_assert(!SHORT || NB == 1, "the short tile serves the 64-wide outputs");
  constexpr int LS = 36, BN = 64 * NB, TR = SHORT ? 32 : 64, AJ = TR / 32;
  __shared__ __attribute__((aligned(16))) float smem[(TR + BN) * LS];
  __shared__ int prow[TR];
  float* As = smem;
  float* Ws = smem + TR * LS;
  int e, m0, cnt, ct;
  if (!find_unit<TR>(g.offsets, g.E, g.ncol, blockIdx.x, e, m0, cnt, ct)) return;
  const int n0 = ct * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, hf = lane >> 5;
  const int wm = SHORT ? 0 : wave >> 1, wn = wave & 1, wk = SHORT ? wave >> 1 : 0;
  const bool rows_here = m0 + 32 * wm < cnt;  // wave-uniform
  if (tid < TR) prow[tid] = (m0 + tid < cnt) ? g.perm[g.offsets[e] + m0 + tid] : -1;
  __syncthreads();
  const float* We = g.W + (int64_t)e * g.N * g.Kd;

  // staging: rows sr + 32*j, 4 floats at column sc
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  // operands through buffer descriptors with the validity folded into the offset (absent pairs, rows past N, the K
  // tail: an offset past the range reads zeros): `cond ? *p : 0` compiles to a branch around the load, and behind a
  // branch around a memory instruction every wait drains everything in flight
  constexpr unsigned PASTO = 0x80000000u;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 0x7ffffffe, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)We, 0, 0x7ffffffe, 0x00020000);
  unsigned aoffs[AJ], woffs[2 * NB];   // (the host checks that A and one expert's weights span < 2 GiB)
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int pp = prow[sr + 32 * j];
    aoffs[j] = pp >= 0 ? (unsigned)(((int64_t)(pp / g.a_div) * g.lda + sc) * 4) : PASTO;
  }
#pragma unroll
  for (int j = 0; j < 2 * NB; ++j)
    woffs[j] = (n0 + sr + 32 * j < g.N) ? (unsigned)(((int64_t)(n0 + sr + 32 * j) * g.Kd + sc) * 4) : PASTO;
  float4 ast[AJ], wst[2 * NB];
  auto prefetch = [&](int k0) {   // (k0 past the contraction: every piece reads zeros)
    const bool kin = k0 + sc < g.Kd;
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      ast[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)((kin && aoffs[j] != PASTO) ? aoffs[j] + 4u * k0 : PASTO), 0, 0));
#pragma unroll
    for (int j = 0; j < 2 * NB; ++j)
      wst[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)((kin && woffs[j] != PASTO) ? woffs[j] + 4u * k0 : PASTO), 0, 0));
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j) st4(&As[(sr + 32 * j) * LS + sc], ast[j]);
#pragma unroll
    for (int j = 0; j < 2 * NB; ++j) st4(&Ws[(sr + 32 * j) * LS + sc], wst[j]);
  };
  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = zero16();
  prefetch(0);
  for (int k0 = 0; k0 < g.Kd; k0 += 32) {
    __syncthreads();
    commit();
    __syncthreads();
    prefetch(k0 + 32);
    if (!rows_here) continue;  // a tail tile with at most 32 pairs: this wave's row half is empty
    const float* ar = &As[(32 * wm + ln) * LS + 16 * hf + (SHORT ? 8 * wk : 0)];
    const float* wr = &Ws[(32 * NB * wn + ln) * LS + 16 * hf + (SHORT ? 8 * wk : 0)];
#pragma unroll
    for (int s4 = 0; s4 < (SHORT ? 2 : 4); ++s4) {
      const float4 a = ld4(ar + 4 * s4);
      float4 b[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = ld4(wr + 32 * j * LS + 4 * s4);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = mfma32(f4(a, x), f4(b[j], x), acc[j]);
      }
    }
  }
  if (SHORT) {  // acc(k-half 0) + acc(k-half 1), through the staging space
    __syncthreads();
    float* red = smem + wn * (16 * 64);
    if (wk == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (wk == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] += red[r * 64 + lane];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + 32 * NB * wn + 32 * j + ln;
    if (n < g.N) {
      const float bv = g.bias ? g.bias[(int64_t)e * g.N + n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = prow[32 * wm + acc_row(r, hf)];
        if (p >= 0) g.Y[(int64_t)p * g.N + n] = acc[j][r] + bv;
      }
    }
  }
}

// The same product for the wide shapes (N >= 128, Kd a multiple of 64).  An expert's pairs are cut into 32-row
// blocks and the blocks into tiles of RB = 1..4 blocks of (nearly) equal height -- 260 pairs are 9 blocks = three
// tiles of 96 rows, not four tiles of 64 and one of 4 (five passes over the weight panel, the last one for 4
// rows) -- x 128 outputs x 64 deep.  The four waves split the 128 OUTPUTS; every wave keeps RB accumulators that
// share its B (weight) fragment: 32 RB MFMAs per wave between barrier pairs.  Operands are staged through buffer
// descriptors (pairs beyond the tile and rows beyond N read as zeros: no compares or selects) and the tile loop
// has no branch around a vector-memory instruction, so every wait is a counted one.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Every wave decodes the unit by itself: lane j holds expert j's tile count, one scan, one ballot (a serial
// walk over the offsets is 2 E dependent scalar loads, microseconds before the first useful load).
// The tile height is chosen per launch: an expert's nb 32-row blocks are cut into ceil(nb / m) tiles of nearly
// equal height with m in 2..4 blocks.  Taller tiles stream the weight panel fewer times, but the units of a
// launch then fill the `slots` workgroup slots of the chip in few, long rounds (768 units of 3 blocks on 512
// slots take two rounds of which the second is half empty); every wave evaluates
//   rounds(m) x (mean blocks per unit + 0.35)      [0.35 blocks: a unit's fixed cost]
// for the three m from the same offsets and takes the smallest (ties: the taller tile).  With `split_tail` the units
// of the last, partial round run as TWO half-width workgroups each (half = 0 / 1: 64 of the 128 outputs, the four
// waves as 2 column x 2 row groups) when twice their number still fits the slots -- the round then costs half.
__device__ __forceinline__ bool find_unit_rb(const int32_t* offsets, int E, int ncol, int slots, bool split_tail, int bid, int& e, int& ct, int& m0, int& rb, int& cnt, int& half) {
  const int lane = threadIdx.x & 63;
  int U[4] = {0, 0, 0, 0}, nbtot = 0;
  for (int base = 0; base < E; base += 64) {
    const int j = base + lane;
    const int c = j < E ? offsets[j + 1] - offsets[j] : 0;
    const int nb = (c + 31) >> 5;
    nbtot += wave_sum(nb);
#pragma unroll
    for (int m = 1; m <= 4; ++m) U[m - 1] += wave_sum((nb + m - 1) / m);
  }
  int mbest = 4;
  float best = 3.0e38f;
#pragma unroll
  for (int m = 4; m >= 2; --m) {  // (m = 1 is left out: it would double the grid bound the host must launch)
    const int u = U[m - 1] * ncol;
    if (u == 0) return false;
    const float blocks = (float)(nbtot * ncol) / (float)u;
    const int full = u / slots, r = u - full * slots;
    // the last, partial round: its units run as two half-width workgroups each when they fit (split_tail)
    const float tail = r == 0 ? 0.f : ((split_tail && 2 * r <= slots) ? 0.6f * blocks + 0.35f : blocks + 0.35f);  // (0.6: both halves stage the whole A tile)
    const float est = (float)full * (blocks + 0.35f) + tail;
    if (est < best) { best = est; mbest = m; }
  }
  const int units = U[mbest - 1] * ncol;
  const int nfull = (units / slots) * slots, r = units - nfull;
  const bool split = split_tail && r > 0 && 2 * r <= slots;
  if (bid >= nfull + (split ? 2 * r : r)) return false;
  int u;
  half = -1;
  if (bid < nfull) {
    u = xcd_remap(bid, nfull);
  } else {
    const int v = xcd_remap(bid - nfull, split ? 2 * r : r);  // (nfull is a multiple of 8: the XCD of a workgroup is unchanged)
    u = nfull + (split ? v >> 1 : v);
    if (split) half = v & 1;
  }
  for (int base = 0; base < E; base += 64) {
    const int j = base + lane;
    const int c = j < E ? offsets[j + 1] - offsets[j] : 0;
    const int nb = (c + 31) >> 5, parts = (nb + mbest - 1) / mbest;
    const int incl = wave_incl_scan(parts * ncol, lane);
    const int tot = __builtin_amdgcn_readlane(incl, 63);
    if (u < tot) {
      const unsigned long long m = __ballot(incl > u);
      const int f = __builtin_ctzll(m);  // first expert whose running unit count passes u
      e = base + f;
      cnt = __shfl(c, f, 64);
      const int parts_e = __shfl(parts, f, 64), nb_e = (cnt + 31) >> 5;
      u -= __shfl(incl, f, 64) - parts_e * ncol;
      ct = u / parts_e;
      const int i = u - ct * parts_e, bs = nb_e / parts_e, extra = nb_e - bs * parts_e;
      rb = bs + (i < extra ? 1 : 0);
      m0 = 32 * (i * bs + min(i, extra));
      return true;
    }
    u -= tot;
  }
  return false;
}

template <int RB, bool HALF>
__device__ __forceinline__ void nt_panel_body(const GemmParams& g, int e, int ct, int m0, int cnt, int half, float* smem, int* prow) {
  constexpr int BK = 32, LS = BK + 4, BN = 128, ROWS = 32 * RB;
  constexpr int STAGE = (128 + BN) * LS;  // floats per LDS stage: A rows [0, 128), W rows [128, 256)
  const int n0 = ct * BN + (HALF ? 64 * half : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  // per row of the tile, computed once by one thread: byte offset of its source row in A and of its output row in
  // Y (rows past the expert's pairs: an offset past the buffers -- loads return zeros, stores are dropped)
  constexpr unsigned PAST = 0x80000000u;      // row sentinel; buffers are < 2 GB, so PAST + (any in-range offset) stays past them
  constexpr unsigned COL_PAST = 0x7FFF0000u;  // column sentinel: PAST + COL_PAST does not wrap
  int* arow_off = prow;
  int* yrow_off = prow + 128;
  if (tid < ROWS) {
    unsigned ao = PAST, yo = PAST;
    if (m0 + tid < cnt) {
      const int pp = g.perm[g.offsets[e] + m0 + tid];
      ao = (unsigned)((int64_t)(pp / g.a_div) * g.lda * 4);
      yo = (unsigned)((int64_t)(g.y_div > 0 ? pp / g.y_div : pp) * g.N * 4);
    }
    arow_off[tid] = (int)ao;
    yrow_off[tid] = (int)yo;
  }
  __syncthreads();
  // staging: 8 threads per row (16 B each): A rows sr + 32 j (j < RB), W rows sr + 32 j (j < 4)
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)g.a_bytes, 0x00020000);
  const float* We = g.W + (int64_t)e * g.N * g.Kd;
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)We, 0, (int)((int64_t)g.N * g.Kd * 4), 0x00020000);
  constexpr int NWS = HALF ? 2 : 4;  // W rows staged per thread: 128 outputs, 64 in a half-width unit
  constexpr int NS = RB + NWS;  // staged float4 per thread and step: slots [0, RB) = A, [RB, NS) = W
  int goff[NS];
#pragma unroll
  for (int j = 0; j < RB; ++j) goff[j] = arow_off[sr + 32 * j] + sc * 4;  // (PAST + sc * 4 stays past the buffer)
#pragma unroll
  for (int j = 0; j < NWS; ++j) goff[RB + j] = (int)(((int64_t)(n0 + sr + 32 * j) * g.Kd + sc) * 4);  // rows >= N: past the buffer
  float4 stg[NS];
  // (the k offset goes into the vector offset: that is the part the hardware range check covers)
  auto gload = [&](int i, int k0) {  // (two calls, not a select between the descriptors: that would make them "divergent")
    if (i < RB) stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, goff[i] + k0 * 4, 0, 0));
    else stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, goff[i] + k0 * 4, 0, 0));
  };
  auto lstore = [&](int i, float* stage) {
    const int row = i < RB ? sr + 32 * i : 128 + sr + 32 * (i - RB);
    st4(&stage[row * LS + sc], stg[i]);
  };
  const int nk = g.Kd / BK;
  // Weight panels start their walk over K at different slabs (the row tiles of one panel together), so that
  // workgroups that start in step do not all read the same slab of their rows at the same time.
  const int rot = (e * g.ncol + ct) % nk;
  auto kof = [&](int kt) { int k = min(kt, nk - 1) + rot; k -= (k >= nk) ? nk : 0; return k * BK; };

  // full unit: wave w owns outputs 32 w .. and all RB row blocks.  Half-width unit: wave (cw, rw) owns outputs
  // 32 cw .. and the row blocks rw, rw + 2.
  constexpr int NA = HALF ? (RB + 1) / 2 : RB;
  const int cw = HALF ? (wave & 1) : wave, rw = HALF ? (wave >> 1) : 0;
  constexpr int RSTEP = HALF ? 2 : 1;
  const bool last_ok = !HALF || (RB % 2 == 0) || rw == 0;  // odd RB: the second row group has one block fewer
  f32x16 acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = zero16();
  // pipeline: during step t (MFMAs on LDS stage t & 1) every thread moves its NS pieces of tile t + 1 from
  // registers to the other stage and refills each register with its piece of tile t + 2 -- one ds_write and
  // one buffer load between groups of MFMAs, ONE barrier per step, every vmcnt wait a counted one.
#pragma unroll
  for (int i = 0; i < NS; ++i) gload(i, kof(0));
#pragma unroll
  for (int i = 0; i < NS; ++i) lstore(i, smem);
#pragma unroll
  for (int i = 0; i < NS; ++i) gload(i, kof(1));
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop is entered in the state its back edge leaves
  __syncthreads();
  const int a_rd = (32 * rw + ln) * LS + 16 * hf, w_rd = (128 + 32 * cw + ln) * LS + 16 * hf;
  for (int kt = 0; kt < nk; ++kt) {
    const float* cur = smem + (kt & 1) * STAGE;
    float* nxt = smem + ((kt + 1) & 1) * STAGE;
    const int k2 = kof(kt + 2);
    float4 a[NA], b = ld4(cur + w_rd);
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = ld4(cur + a_rd + 32 * RSTEP * j * LS);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float4 an[NA], bn = b;
#pragma unroll
      for (int j = 0; j < NA; ++j) an[j] = a[j];
      if (s4 + 1 < 4) {
        bn = ld4(cur + w_rd + 4 * (s4 + 1));
#pragma unroll
        for (int j = 0; j < NA; ++j) an[j] = ld4(cur + a_rd + 32 * RSTEP * j * LS + 4 * (s4 + 1));
      }
      // this group's share of the tile movement
#pragma unroll
      for (int i = s4 * NS / 4; i < (s4 + 1) * NS / 4; ++i) {
        lstore(i, nxt);
        gload(i, k2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int j = 0; j < NA - 1; ++j) acc[j] = mfma32(f4(a[j], x), f4(b, x), acc[j]);
      }
      if (last_ok) {
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[NA - 1] = mfma32(f4(a[NA - 1], x), f4(b, x), acc[NA - 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      b = bn;
#pragma unroll
      for (int j = 0; j < NA; ++j) a[j] = an[j];
    }
    __syncthreads();
  }
  // epilogue: one add and one range-checked buffer store per element (rows past the pairs and columns past N
  // carry offsets past the buffer and are dropped by the hardware: no compares, branches or 64-bit products)
  const int n = n0 + 32 * cw + ln;
  const float bv = (g.bias && n < g.N) ? g.bias[(int64_t)e * g.N + n] : 0.f;
  const unsigned coff = n < g.N ? (unsigned)n * 4u : COL_PAST;
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.Y, 0, (int)g.y_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    if (j == NA - 1 && !last_ok) break;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned yo = (unsigned)yrow_off[32 * (rw + RSTEP * j) + acc_row(r, hf)];
      if (g.y_div > 0) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[j][r] + bv, y_rsrc, (int)(yo + coff), 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[j][r] + bv), y_rsrc, (int)(yo + coff), 0, 0);
    }
  }
}

__global__ __launch_bounds__(256, 2) void grouped_nt_wide_kernel(GemmParams g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (128 + 128) * 36];
  __shared__ int prow[256];
  int e, ct, m0, rb, cnt, half;
  if (!find_unit_rb(g.offsets, g.E, g.ncol, g.slots, true, blockIdx.x, e, ct, m0, rb, cnt, half)) return;
  // (wave-uniform by construction; said again here, or the weight panel's buffer descriptor counts as divergent
  // and every load through it is wrapped in a waterfall loop)
  e = __builtin_amdgcn_readfirstlane(e); ct = __builtin_amdgcn_readfirstlane(ct); m0 = __builtin_amdgcn_readfirstlane(m0);
  rb = __builtin_amdgcn_readfirstlane(rb); cnt = __builtin_amdgcn_readfirstlane(cnt); half = __builtin_amdgcn_readfirstlane(half);
  if (half < 0) {
    if (rb == 4) nt_panel_body<4, false>(g, e, ct, m0, cnt, 0, smem, prow);
    else if (rb == 3) nt_panel_body<3, false>(g, e, ct, m0, cnt, 0, smem, prow);
    else if (rb == 2) nt_panel_body<2, false>(g, e, ct, m0, cnt, 0, smem, prow);
    else nt_panel_body<1, false>(g, e, ct, m0, cnt, 0, smem, prow);
  } else {
    if (rb == 4) nt_panel_body<4, true>(g, e, ct, m0, cnt, half, smem, prow);
    else if (rb == 3) nt_panel_body<3, true>(g, e, ct, m0, cnt, half, smem, prow);
    else if (rb == 2) nt_panel_body<2, true>(g, e, ct, m0, cnt, half, smem, prow);
    else nt_panel_body<1, true>(g, e, ct, m0, cnt, half, smem, prow);
  }
}

// The input-gradient product Y[p, kk] = scale[p] * sum_n A[arow(p), n] * W[e, n, kk] on the same pipeline: tiles of
// RB 32-row blocks x 128 outputs (kk) x 32 deep (n).  W is consumed as stored, (n, kk) rows: the B operand of a
// step is a column of the staged (32 n x 128 kk) slab -- ds_read_b32, consecutive lanes on consecutive kk.
template <int RB, bool HALF>
__device__ __forceinline__ void nn_panel_body(const GemmParams& g, int e, int ct, int m0, int cnt, int half, float* smem, int* prow) {
  constexpr int BK = 32, LS = BK + 4, BN = 128, WS = BN + 4, ROWS = 32 * RB;
  constexpr int A_FLOATS = 128 * LS, STAGE = A_FLOATS + BK * WS;  // per stage: A rows [0, 128), then the W slab
  const int c0 = ct * BN + (HALF ? 64 * half : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  constexpr unsigned PAST = 0x80000000u, COL_PAST = 0x7FFF0000u;  // as in nt_panel_body
  int* arow_off = prow;
  int* yrow_off = prow + 128;
  float* srow = reinterpret_cast<float*>(prow + 256);
  if (tid < ROWS) {
    unsigned ao = PAST, yo = PAST;
    float sv = 0.f;
    if (m0 + tid < cnt) {
      const int pp = g.perm[g.offsets[e] + m0 + tid];
      ao = (unsigned)((int64_t)(pp / g.a_div) * g.lda * 4);
      yo = (unsigned)((int64_t)(g.y_div > 0 ? pp / g.y_div : pp) * g.Kd * 4);
      sv = g.scale ? g.scale[pp] : 1.f;
    }
    arow_off[tid] = (int)ao;
    yrow_off[tid] = (int)yo;
    srow[tid] = sv;
  }
  __syncthreads();
  // staging: A 8 threads per row (rows sr + 32 j, j < RB); W 32 threads per n row (rows wr + 8 j, j < 4)
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  constexpr int WTPR = HALF ? 16 : 32;      // threads per W slab row (16 B each): 128 outputs, 64 in a half-width unit
  constexpr int NWS = HALF ? 2 : 4;         // passes over the 32 n rows of the slab
  const int wr = tid / WTPR, wc = (tid % WTPR) * 4;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)g.a_bytes, 0x00020000);
  const float* We = g.W + (int64_t)e * g.N * g.Kd;
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)We, 0, (int)((int64_t)g.N * g.Kd * 4), 0x00020000);
  constexpr int NS = RB + NWS;
  int goff[NS];
#pragma unroll
  for (int j = 0; j < RB; ++j) goff[j] = arow_off[sr + 32 * j] + sc * 4;
  // columns >= Kd: the whole thread reads past the buffer (zeros)
#pragma unroll
  for (int j = 0; j < NWS; ++j) goff[RB + j] = c0 + wc < g.Kd ? (int)(((int64_t)(wr + (256 / WTPR) * j) * g.Kd + c0 + wc) * 4) : (int)PAST;
  const int wstep = g.Kd * 4;  // bytes per n row of W
  float4 stg[NS];
  auto gload = [&](int i, int k0) {
    if (i < RB) stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, goff[i] + k0 * 4, 0, 0));
    else stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, goff[i] + k0 * wstep, 0, 0));
  };
  auto lstore = [&](int i, float* stage) {
    if (i < RB) st4(&stage[(sr + 32 * i) * LS + sc], stg[i]);
    else st4(&stage[A_FLOATS + (wr + (256 / WTPR) * (i - RB)) * WS + wc], stg[i]);
  };
  const int nk = g.N / BK;
  const int rot = (e * g.ncol + ct) % nk;
  auto kof = [&](int kt) { int k = min(kt, nk - 1) + rot; k -= (k >= nk) ? nk : 0; return k * BK; };

  // half-width unit: wave (cw, rw) owns outputs 32 cw .. and the row blocks rw, rw + 2 (as in nt_panel_body)
  constexpr int NA = HALF ? (RB + 1) / 2 : RB;
  const int cw = HALF ? (wave & 1) : wave, rw = HALF ? (wave >> 1) : 0;
  constexpr int RSTEP = HALF ? 2 : 1;
  const bool last_ok = !HALF || (RB % 2 == 0) || rw == 0;
  f32x16 acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = zero16();
#pragma unroll
  for (int i = 0; i < NS; ++i) gload(i, kof(0));
#pragma unroll
  for (int i = 0; i < NS; ++i) lstore(i, smem);
#pragma unroll
  for (int i = 0; i < NS; ++i) gload(i, kof(1));
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop is entered in the state its back edge leaves
  __syncthreads();
  const int a_rd = (32 * rw + ln) * LS + 16 * hf, w_rd = A_FLOATS + (16 * hf) * WS + 32 * cw + ln;
  for (int kt = 0; kt < nk; ++kt) {
    const float* cur = smem + (kt & 1) * STAGE;
    float* nxt = smem + ((kt + 1) & 1) * STAGE;
    const int k2 = kof(kt + 2);
    float4 a[NA];
    float b[4];
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = ld4(cur + a_rd + 32 * RSTEP * j * LS);
#pragma unroll
    for (int x = 0; x < 4; ++x) b[x] = cur[w_rd + x * WS];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float4 an[NA];
      float bn[4];
#pragma unroll
      for (int j = 0; j < NA; ++j) an[j] = a[j];
#pragma unroll
      for (int x = 0; x < 4; ++x) bn[x] = b[x];
      if (s4 + 1 < 4) {
#pragma unroll
        for (int j = 0; j < NA; ++j) an[j] = ld4(cur + a_rd + 32 * RSTEP * j * LS + 4 * (s4 + 1));
#pragma unroll
        for (int x = 0; x < 4; ++x) bn[x] = cur[w_rd + (4 * (s4 + 1) + x) * WS];
      }
#pragma unroll
      for (int i = s4 * NS / 4; i < (s4 + 1) * NS / 4; ++i) {
        lstore(i, nxt);
        gload(i, k2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int j = 0; j < NA - 1; ++j) acc[j] = mfma32(f4(a[j], x), b[x], acc[j]);
      }
      if (last_ok) {
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[NA - 1] = mfma32(f4(a[NA - 1], x), b[x], acc[NA - 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NA; ++j) a[j] = an[j];
#pragma unroll
      for (int x = 0; x < 4; ++x) b[x] = bn[x];
    }
    __syncthreads();
  }
  const int c = c0 + 32 * cw + ln;
  const unsigned coff = c < g.Kd ? (unsigned)c * 4u : COL_PAST;
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.Y, 0, (int)g.y_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    if (j == NA - 1 && !last_ok) break;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * (rw + RSTEP * j) + acc_row(r, hf);
      if (g.y_div > 0) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[j][r] * srow[row], y_rsrc, (int)((unsigned)yrow_off[row] + coff), 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[j][r] * srow[row]), y_rsrc, (int)((unsigned)yrow_off[row] + coff), 0, 0);
    }
  }
}

__global__ __launch_bounds__(256, 2) void grouped_nn_wide_kernel(GemmParams g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (128 * 36 + 32 * 132)];
  __shared__ int prow[384];
  int e, ct, m0, rb, cnt, half;
  if (!find_unit_rb(g.offsets, g.E, g.ncol, g.slots, true, blockIdx.x, e, ct, m0, rb, cnt, half)) return;
  e = __builtin_amdgcn_readfirstlane(e); ct = __builtin_amdgcn_readfirstlane(ct); m0 = __builtin_amdgcn_readfirstlane(m0);
  rb = __builtin_amdgcn_readfirstlane(rb); cnt = __builtin_amdgcn_readfirstlane(cnt); half = __builtin_amdgcn_readfirstlane(half);
  if (half < 0) {
    if (rb == 4) nn_panel_body<4, false>(g, e, ct, m0, cnt, 0, smem, prow);
    else if (rb == 3) nn_panel_body<3, false>(g, e, ct, m0, cnt, 0, smem, prow);
    else if (rb == 2) nn_panel_body<2, false>(g, e, ct, m0, cnt, 0, smem, prow);
    else nn_panel_body<1, false>(g, e, ct, m0, cnt, 0, smem, prow);
  } else {
    if (rb == 4) nn_panel_body<4, true>(g, e, ct, m0, cnt, half, smem, prow);
    else if (rb == 3) nn_panel_body<3, true>(g, e, ct, m0, cnt, half, smem, prow);
    else if (rb == 2) nn_panel_body<2, true>(g, e, ct, m0, cnt, half, smem, prow);
    else nn_panel_body<1, true>(g, e, ct, m0, cnt, half, smem, prow);
  }
}

// Y[p, kk] = scale[p] * sum_n A[arow(p), n] * W[e, n, kk]              tile: 64 pairs x 64*NB outputs
// SHORT (NB 1): 32 pairs x 64 outputs, the waves as 2 output halves x 2 halves of every slab (see grouped_nt_kernel)
template <int NB, bool SHORT = false>
__global__ __launch_bounds__(256, 2) void grouped_nn_kernel(GemmParams g) {
  static_assert(!SHORT || NB == 1, "the short tile serves the 64-wide outputs");
  constexpr int LS = 36, BC = 64 * NB, WS = BC + 4, TR = SHORT ? 32 : 64, AJ = TR / 32;
  __shared__ __attribute__((aligned(16))) float smem[64 * LS + 32 * WS];
  __shared__ int prow[TR];
  float* As = smem;            // [TR pairs][32 n]
  float* Ws = smem + TR * LS;  // [32 n][BC kk]
  int e, m0, cnt, ct;
  if (!find_unit<TR>(g.offsets, g.E, g.ncol, blockIdx.x, e, m0, cnt, ct)) return;
  const int c0 = ct * BC;  // output (kk) tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, hf = lane >> 5;
  const int wm = SHORT ? 0 : wave >> 1, wn = wave & 1, wk = SHORT ? wave >> 1 : 0;
  const bool rows_here = m0 + 32 * wm < cnt;  // wave-uniform
  __shared__ float srow[TR];   // the rows' scale factors: one request per row here, not sixteen serial ones per lane at the end
  if (tid < TR) {
    const int pp = (m0 + tid < cnt) ? g.perm[g.offsets[e] + m0 + tid] : -1;
    prow[tid] = pp;
    srow[tid] = (pp >= 0 && g.scale) ? g.scale[pp] : 1.f;
  }
  __syncthreads();
  const float* We = g.W + (int64_t)e * g.N * g.Kd;

  const int sr = tid >> 3, sc = (tid & 7) * 4;      // A: rows sr, sr+32; cols sc
  constexpr int TPR = BC / 4;                        // threads per W row (16 B each)
  constexpr int RPP = 256 / TPR;                     // W rows per pass
  const int wrow_ = tid / TPR, wc = (tid % TPR) * 4; // W: rows wrow_ + RPP*j (n); cols wc (kk)
  // operands through buffer descriptors, validity folded into the offset (see grouped_nt_kernel)
  constexpr unsigned PASTO = 0x80000000u;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 0x7ffffffe, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)We, 0, 0x7ffffffe, 0x00020000);
  unsigned aoffs[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int pp = prow[sr + 32 * j];
    aoffs[j] = pp >= 0 ? (unsigned)(((int64_t)(pp / g.a_div) * g.lda + sc) * 4) : PASTO;
  }
  const bool cin = c0 + wc < g.Kd;
  float4 ast[AJ], wst[32 / RPP];
  auto prefetch = [&](int nb) {   // (nb past the contraction: every piece reads zeros)
    const bool nin = nb + sc < g.N;
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      ast[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)((nin && aoffs[j] != PASTO) ? aoffs[j] + 4u * nb : PASTO), 0, 0));
#pragma unroll
    for (int j = 0; j < 32 / RPP; ++j) {
      const int n = nb + wrow_ + RPP * j;
      wst[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)((cin && n < g.N) ? (unsigned)(((int64_t)n * g.Kd + c0 + wc) * 4) : PASTO), 0, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j) st4(&As[(sr + 32 * j) * LS + sc], ast[j]);
#pragma unroll
    for (int j = 0; j < 32 / RPP; ++j) st4(&Ws[(wrow_ + RPP * j) * WS + wc], wst[j]);
  };
  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = zero16();
  prefetch(0);
  for (int nb = 0; nb < g.N; nb += 32) {
    __syncthreads();
    commit();
    __syncthreads();
    prefetch(nb + 32);
    if (!rows_here) continue;  // a tail tile with at most 32 pairs: this wave's row half is empty
    const float* ar = &As[(32 * wm + ln) * LS + 16 * hf + (SHORT ? 8 * wk : 0)];
    const float* wc_ = &Ws[(16 * hf + (SHORT ? 8 * wk : 0)) * WS + 32 * NB * wn + ln];
#pragma unroll
    for (int s4 = 0; s4 < (SHORT ? 2 : 4); ++s4) {
      const float4 a = ld4(ar + 4 * s4);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = mfma32(f4(a, x), wc_[(4 * s4 + x) * WS + 32 * j], acc[j]);
      }
    }
  }
  if (SHORT) {  // acc(k-half 0) + acc(k-half 1), through the staging space
    __syncthreads();
    float* red = smem + wn * (16 * 64);
    if (wk == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (wk == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] += red[r * 64 + lane];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int c = c0 + 32 * NB * wn + 32 * j + ln;
    if (c < g.Kd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = prow[32 * wm + acc_row(r, hf)];
        if (p >= 0) g.Y[(int64_t)p * g.Kd + c] = acc[j][r] * srow[32 * wm + acc_row(r, hf)];
      }
    }
  }
}

// dW[e, n, kk] = sum_{p in e} scale[p] * G[p/a_div, n] * X[p/b_div, kk] ; dbias[e, n] = sum_p scale[p] * G[.., n]
// tile: 64 n x 64*NB kk
template <int NB>
__global__ __launch_bounds__(256, 2) void grouped_wgrad_kernel(GemmParams g) {
  constexpr int GS = 68, BC = 64 * NB, XS = BC + 4;
  __shared__ __attribute__((aligned(16))) float smem[32 * GS + 32 * XS];
  float* Gs = smem;            // [32 pairs][64 n]   (already scaled)
  float* Xs = smem + 32 * GS;  // [32 pairs][BC kk]
  const int e = blockIdx.z;
  const int n0 = blockIdx.y * 64, c0 = blockIdx.x * BC;
  const int beg = g.offsets[e], cnt = g.offsets[e + 1] - beg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;  // wm: n half, wn: kk half
  const int sr = tid >> 4, sc = (tid & 15) * 4;  // G: rows sr, sr+16 (pairs); cols sc
  constexpr int TPR = BC / 4, RPP = 256 / TPR;
  const int xr = tid / TPR, xc = (tid % TPR) * 4;  // X: rows xr + RPP*j
  // Per step the 32 pairs' source rows and scales come from a small LDS table that 32 threads fill ONE STEP
  // AHEAD (perm lookup, two integer divisions by run-time divisors, the scale): done per staged row by every
  // thread, that bookkeeping was 9.7 VALU instructions per MFMA (PMC) -- every one paid against the f32 MFMA pipe.
  __shared__ int tab_a[2][32], tab_x[2][32];
  __shared__ float tab_s[2][32];
  auto fill_table = [&](int step) {  // rows of pairs [32 step, 32 step + 32) of this expert; -1 = past the end
    if (tid < 32) {
      const int i = 32 * step + tid;
      int ra = -1, rx = -1;
      float sv = 0.f;
      if (i < cnt) {
        const int p = g.perm[beg + i];
        ra = p / g.a_div;
        rx = p / g.b_div;
        sv = g.scale ? g.scale[p] : 1.f;
      }
      tab_a[step & 1][tid] = ra; tab_x[step & 1][tid] = rx; tab_s[step & 1][tid] = sv;
    }
  };
  float4 gst[2], xst[32 / RPP];
  const bool g_in = n0 + sc < g.N, x_in = c0 + xc < g.Kd;
  auto prefetch = [&](int step) {
    const int b = step & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      gst[j] = zero4();
      const int ra = tab_a[b][sr + 16 * j];
      if (ra >= 0 && g_in) {
        const float sv = tab_s[b][sr + 16 * j];
        float4 t = ld4(g.A + (int64_t)ra * g.lda + n0 + sc);
        t.x *= sv; t.y *= sv; t.z *= sv; t.w *= sv;
        gst[j] = t;
      }
    }
#pragma unroll
    for (int j = 0; j < 32 / RPP; ++j) {
      xst[j] = zero4();
      const int rx = tab_x[b][xr + RPP * j];
      if (rx >= 0 && x_in) xst[j] = ld4(g.B2 + (int64_t)rx * g.ldb + c0 + xc);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) st4(&Gs[(sr + 16 * j) * GS + sc], gst[j]);
#pragma unroll
    for (int j = 0; j < 32 / RPP; ++j) st4(&Xs[(xr + RPP * j) * XS + xc], xst[j]);
  };
  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = zero16();
  float bsum = 0.f;
  const int nstep = (cnt + 31) / 32;
  fill_table(0);
  __syncthreads();
  if (nstep > 0) prefetch(0);
  fill_table(1);
  for (int st = 0; st < nstep; ++st) {
    __syncthreads();  // tile st consumed by every wave; table st + 1 complete
    commit();
    __syncthreads();
    if (st + 1 < nstep) prefetch(st + 1);
    fill_table(st + 2);  // into the buffer prefetch(st) read before the barriers above
    const float* gc = &Gs[(16 * hf) * GS + 32 * wm + ln];
    const float* xc_ = &Xs[(16 * hf) * XS + 32 * NB * wn + ln];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float gv = gc[s * GS];
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = mfma32(gv, xc_[s * XS + 32 * j], acc[j]);
      bsum += gv;
    }
  }
  float* dW = g.Y + (int64_t)e * g.N * g.Kd;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int c = c0 + 32 * NB * wn + 32 * j + ln;
    if (c < g.Kd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * wm + acc_row(r, hf);
        if (n < g.N) dW[(int64_t)n * g.Kd + c] = acc[j][r];
      }
    }
  }
  if (g.dbias && blockIdx.x == 0 && wn == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    const int n = n0 + 32 * wm + ln;
    if (hf == 0 && n < g.N) g.dbias[(int64_t)e * g.N + n] = bsum;
  }
}

// The weight gradient for the wide shapes (N, Kd >= 128) on the same pipeline: one workgroup owns a 128 (n) x 128
// (kk) tile of one expert's dW and walks the expert's pairs 32 at a time; 2 x 2 waves of 64 x 64 (four accumulators
// each), both operands read as columns of the staged slabs G[32 pairs][128 n] (scaled at the LDS store) and
// X[32 pairs][128 kk] (the upper 16 pairs of a slab are stored rotated by 32 columns, so the two half-waves of a
// ds_read_b32 use disjoint banks).  The source rows and scales of a step come from a four-deep ring of small LDS
// tables that 32 threads fill three steps ahead from registers loaded one (scale) and two (perm) steps earlier:
// no thread waits for a dependent load inside the pipeline, and no load sits behind a branch.
__device__ __forceinline__ int div_by(int x, int d, int shift) { return shift >= 0 ? x >> shift : x / d; }

// TNB / TKB: 32-column blocks per wave along n / kk -- tile (64 TNB) x (64 TKB): (2, 2) for the square experts,
// (1, 2) / (2, 1) for SwitchHead's (64 x D) and (D x 64) experts.
template <bool HAS_SCALE, int TNB, int TKB>
__global__ __launch_bounds__(256, 2) void grouped_wgrad_wide_kernel(GemmParams g) {
  constexpr int WG_ = 64 * TNB, WX_ = 64 * TKB;       // slab widths (columns of G and of X in the tile)
  constexpr int SG = WG_ + 4, SX = WX_ + 4;            // LDS row strides
  constexpr int GSLAB = 32 * SG, STAGE = GSLAB + 32 * SX;
  constexpr int NG = 2 * TNB, NX = 2 * TKB, NS = NG + NX;  // float4 per thread and step: G slots [0, NG), X slots [NG, NS)
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  __shared__ int tab_a[4][32], tab_x[4][32];
  __shared__ float tab_s[4][32];
  constexpr unsigned PAST = 0x80000000u, COL_PAST = 0x7FFF0000u;  // as in nt_panel_body
  const int ntn = (g.N + WG_ - 1) / WG_, ntk = g.ncol;  // tiles along n, kk
  const int u2 = xcd_remap(blockIdx.x, gridDim.x);
  // rsplit 2: neighbouring workgroups take the two halves of a tile's pairs.  With few tiles -- SwitchHead's 64-wide experts
  // make E x 8 = 256, one workgroup and one wave per SIMD on every CU -- nothing covers a workgroup's waits (MFMA pipe busy
  // 0.43-0.46, profiles/r03_vitmoe_pmc_digest.txt); two half-length workgroups per CU cover each other's.
  const int part = g.rsplit == 2 ? (u2 & 1) : 0;
  const int u = g.rsplit == 2 ? (u2 >> 1) : u2;
  const int e = __builtin_amdgcn_readfirstlane(u / (ntn * ntk)), rem = u - e * ntn * ntk;
  const int n0 = (rem / ntk) * WG_, c0 = (rem % ntk) * WX_;
  int beg = g.offsets[e], cnt = g.offsets[e + 1] - beg;
  if (g.rsplit == 2) {
    const int first = ((cnt + 63) >> 6) << 5;   // the first half, in whole 32-pair steps
    if (part) { beg += min(first, cnt); cnt = max(cnt - first, 0); }
    else cnt = min(cnt, first);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;  // wm: n half, wn: kk half
  const int nstep = (cnt + 31) >> 5;

  // every thread runs the table writers' loads (tid & 31 picks the pair; the range checks do the masking), only
  // the first 32 write the tables
  const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.perm + beg), 0, cnt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.scale, 0, HAS_SCALE ? (int)g.s_bytes : 0, 0x00020000);
  auto pair_of = [&](int i) {  // pair id of the expert's i-th pair, -1 past its end
    const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(p_rsrc, i * 4, 0, 0);
    return i < cnt ? v : -1;
  };
  auto scale_of = [&](int pp) {  // 0 for "no pair" (reads past the buffer)
    if (HAS_SCALE) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_rsrc, pp >= 0 ? pp * 4 : (int)PAST, 0, 0));
    return pp >= 0 ? 1.f : 0.f;
  };
  auto put = [&](int slot, int t, int pp, float sv) {
    tab_a[slot][t] = pp >= 0 ? (int)((int64_t)div_by(pp, g.a_div, g.a_shift) * g.lda * 4) : (int)PAST;
    tab_x[slot][t] = pp >= 0 ? (int)((int64_t)div_by(pp, g.b_div, g.b_shift) * g.ldb * 4) : (int)PAST;
    tab_s[slot][t] = sv;
  };
  {  // steps 0, 1, 2
    const int pp = pair_of(min(tid, 95));
    const float sv = scale_of(pp);
    if (tid < 96) put(tid >> 5, tid & 31, pp, sv);
  }
  // registers of the table writers: pair of step st + 3 with its scale, pair of step st + 4
  int pv1 = pair_of(96 + (tid & 31));
  float sv1 = scale_of(pv1);
  int pv0 = pair_of(128 + (tid & 31));
  __syncthreads();

  // staging: 16 TNB (16 TKB) threads per G (X) slab row, 16 B each; G rows gr + (16 / TNB) j, X rows xr + (16 / TKB) j
  const int gr = tid / (16 * TNB), gc = (tid % (16 * TNB)) * 4;
  const int xr = tid / (16 * TKB), xc = (tid % (16 * TKB)) * 4;
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.B2, 0, (int)g.b_bytes, 0x00020000);
  const unsigned gcol = n0 + gc < g.N ? (unsigned)(n0 + gc) * 4u : COL_PAST, xcol = c0 + xc < g.Kd ? (unsigned)(c0 + xc) * 4u : COL_PAST;
  float4 stg[NS];
  float ssv[NG];
  auto gload = [&](int i, int step) {
    const int slot = step & 3;
    if (i < NG) {
      const int row = gr + (16 / TNB) * i;
      stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, (int)((unsigned)tab_a[slot][row] + gcol), 0, 0));
      ssv[i] = tab_s[slot][row];
    } else {
      const int row = xr + (16 / TKB) * (i - NG);
      stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)((unsigned)tab_x[slot][row] + xcol), 0, 0));
    }
  };
  auto lstore = [&](int i, float* stage) {  // rows 16..31 of a slab (the second half of its slots): columns rotated by 32
    if (i < NG) {
      const int col = (gc + 32 * (i / TNB)) & (WG_ - 1);
      float4 t = stg[i];
      const float sv = ssv[i];
      t.x *= sv; t.y *= sv; t.z *= sv; t.w *= sv;
      st4(&stage[(gr + (16 / TNB) * i) * SG + col], t);
    } else {
      const int col = (xc + 32 * ((i - NG) / TKB)) & (WX_ - 1);
      st4(&stage[GSLAB + (xr + (16 / TKB) * (i - NG)) * SX + col], stg[i]);
    }
  };
  f32x16 acc[TNB][TKB];
#pragma unroll
  for (int i = 0; i < TNB; ++i)
#pragma unroll
    for (int j = 0; j < TKB; ++j) acc[i][j] = zero16();
  float bsum[TNB];
#pragma unroll
  for (int i = 0; i < TNB; ++i) bsum[i] = 0.f;
  const bool want_bias = g.dbias && c0 == 0 && wn == 0;

#pragma unroll
  for (int i = 0; i < NS; ++i) gload(i, 0);
#pragma unroll
  for (int i = 0; i < NS; ++i) lstore(i, smem);
#pragma unroll
  for (int i = 0; i < NS; ++i) gload(i, 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop is entered in the state its back edge leaves
  __syncthreads();
  // this lane's columns in the slabs, as stored for its half's 16 pairs
  int g_rd[TNB], x_rd[TKB];
#pragma unroll
  for (int i = 0; i < TNB; ++i) g_rd[i] = (16 * hf) * SG + ((32 * TNB * wm + 32 * i + ln + 32 * hf) & (WG_ - 1));
#pragma unroll
  for (int j = 0; j < TKB; ++j) x_rd[j] = GSLAB + (16 * hf) * SX + ((32 * TKB * wn + 32 * j + ln + 32 * hf) & (WX_ - 1));
  for (int st = 0; st < nstep; ++st) {
    const float* cur = smem + (st & 1) * STAGE;
    float* nxt = smem + ((st + 1) & 1) * STAGE;
    // table of step st + 3 from the registers, then the loads that refill them (scale of st + 4's pairs, pairs of st + 5)
    if (tid < 32) put((st + 3) & 3, tid, pv1, sv1);
    pv1 = pv0;
    sv1 = scale_of(pv1);
    pv0 = pair_of(32 * (st + 5) + (tid & 31));
    float a[TNB][4], b[TKB][4];
    auto rd = [&](int q, float (&av)[TNB][4], float (&bv)[TKB][4]) {  // pairs 4 q .. 4 q + 3 of this half
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int i = 0; i < TNB; ++i) av[i][x] = cur[g_rd[i] + (4 * q + x) * SG];
#pragma unroll
        for (int j = 0; j < TKB; ++j) bv[j][x] = cur[x_rd[j] + (4 * q + x) * SX];
      }
    };
    rd(0, a, b);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float an[TNB][4], bn[TKB][4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int i = 0; i < TNB; ++i) an[i][x] = a[i][x];
#pragma unroll
        for (int j = 0; j < TKB; ++j) bn[j][x] = b[j][x];
      }
      if (q + 1 < 4) rd(q + 1, an, bn);
#pragma unroll
      for (int i = q * NS / 4; i < (q + 1) * NS / 4; ++i) {
        lstore(i, nxt);
        gload(i, st + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int i = 0; i < TNB; ++i)
#pragma unroll
          for (int j = 0; j < TKB; ++j) acc[i][j] = mfma32(a[i][x], b[j][x], acc[i][j]);
      }
      if (want_bias) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int i = 0; i < TNB; ++i) bsum[i] += a[i][x];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int i = 0; i < TNB; ++i) a[i][x] = an[i][x];
#pragma unroll
        for (int j = 0; j < TKB; ++j) b[j][x] = bn[j][x];
      }
    }
    __syncthreads();
  }
  // epilogue through the expert's slab of dW as a buffer: rows >= N fall past it, columns >= Kd carry COL_PAST
  const __amdgpu_buffer_rsrc_t d_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(g.Y + (int64_t)e * g.N * g.Kd), 0, (int)((int64_t)g.N * g.Kd * 4), 0x00020000);
  const int row_bytes = g.Kd * 4;
#pragma unroll
  for (int i = 0; i < TNB; ++i) {
    const unsigned nbase = (unsigned)(n0 + 32 * TNB * wm + 32 * i + 4 * hf) * (unsigned)row_bytes;
#pragma unroll
    for (int j = 0; j < TKB; ++j) {
      const int c = c0 + 32 * TKB * wn + 32 * j + ln;
      const unsigned coff = c < g.Kd ? (unsigned)c * 4u : COL_PAST;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[i][j][r];  // (a copy: bit_cast of the vector-element lvalue itself reads element 0)
        if (g.rsplit == 2)
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, d_rsrc, (int)(nbase + (unsigned)(acc_row(r, 0) * row_bytes) + coff), 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), d_rsrc,
                                                (int)(nbase + (unsigned)(acc_row(r, 0) * row_bytes) + coff), 0, 0);
      }
    }
  }
  if (want_bias) {
#pragma unroll
    for (int i = 0; i < TNB; ++i) {
      const float t = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int n = n0 + 32 * TNB * wm + 32 * i + ln;
      if (hf == 0 && n < g.N) {
        if (g.rsplit == 2) atomicAdd(&g.dbias[(int64_t)e * g.N + n], t);
        else g.dbias[(int64_t)e * g.N + n] = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// out[g, :] = sum_{o < outer} ( sum over the k slots of unit (g*outer+o), ascending expert id,
//             of scale[p] * Y[p, :] )     -- the accumulation order of the reference loops.
// v_div > 0: Y is laid out per (row group, expert) -- pair p reads row (p / v_div) * E + ids[p] (the products of
// amk_moe_route_distinct's lists, computed once per distinct (group, expert)); v_div == 0: row p.
__global__ __launch_bounds__(256) void combine_kernel(const float* __restrict__ Y, const int64_t* __restrict__ ids,
                                                      const float* __restrict__ scale, int64_t G, int outer, int k,
                                                      int N, int v_div, int E, float* __restrict__ out) {
  const int nv = N >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= G * nv) return;
  const int64_t gi = idx / nv;
  const int c = (int)(idx % nv) * 4;
  float4 tot = zero4();
  for (int o = 0; o < outer; ++o) {
    const int64_t u = gi * outer + o;
    int order[MAX_K];
    for (int s = 0; s < k; ++s) order[s] = s;
    for (int a = 1; a < k; ++a)  // insertion sort of the slots by expert id
      for (int b = a; b > 0 && ids[u * k + order[b]] < ids[u * k + order[b - 1]]; --b) {
        const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t;
      }
    float4 acc = zero4();
    for (int s = 0; s < k; ++s) {
      const int64_t p = u * k + order[s];
      const float w = scale ? scale[p] : 1.f;
      const float4 y = ld4(Y + (v_div ? (p / v_div) * E + ids[p] : p) * N + c);
      // res += w * y with separate rounding of the product, as the eager reference does
      acc.x = __fadd_rn(acc.x, __fmul_rn(w, y.x));
      acc.y = __fadd_rn(acc.y, __fmul_rn(w, y.y));
      acc.z = __fadd_rn(acc.z, __fmul_rn(w, y.z));
      acc.w = __fadd_rn(acc.w, __fmul_rn(w, y.w));
    }
    if (outer == 1) { tot = acc; }
    else { tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w; }
  }
  st4(out + gi * N + c, tot);
}

// Z[g, e, :] = sum over the fan pairs of row g (p = g*fan + j) that chose expert e of scale[p] * A[p / a_div, :].
// With Z in hand the sum over a row's pairs of (pair row) x (its expert's matrix) is ONE dense product
// Z (G, E*d) x (E*d, N): no (pairs, N) intermediate.  One workgroup per row; a thread owns float4s of the row and walks
// the row's pairs in order (a fixed order: the result is reproducible).
__global__ __launch_bounds__(256) void expert_sums_kernel(const float* __restrict__ A, int64_t lda, int a_div,
                                                          const int64_t* __restrict__ ids, const float* __restrict__ scale,
                                                          int fan, int E, int d, float* __restrict__ Z) {
  extern __shared__ int es_lds[];
  int* sid = es_lds;                                       // [fan] expert of pair j
  float* ssc = reinterpret_cast<float*>(es_lds + fan);     // [fan] its scale
  int* order = es_lds + 2 * fan;                           // [fan] the pairs sorted by expert (stable: ascending j)
  int* start = es_lds + 3 * fan;                           // [E + 1] first entry of an expert in `order`
  const int64_t g = blockIdx.x, p0 = g * fan;
  for (int j = threadIdx.x; j < fan; j += 256) {
    sid[j] = (int)ids[p0 + j];
    ssc[j] = scale ? scale[p0 + j] : 1.f;
  }
  __syncthreads();
  // counting sort of the row's pairs by expert, one thread per expert: an output then walks only its expert's pairs
  // (0.5 on average at 16 pairs over 32 experts) instead of comparing against all of them
  for (int e = threadIdx.x; e < E; e += 256) {
    int first = 0, mine = 0;
    for (int j = 0; j < fan; ++j) { first += sid[j] < e; mine += sid[j] == e; }
    start[e] = first;
    if (e == E - 1) start[E] = first + mine;
    for (int j = 0; j < fan; ++j)
      if (sid[j] == e) order[first++] = j;
  }
  __syncthreads();
  const int dv = d >> 2, nv = E * dv;
  for (int idx = threadIdx.x; idx < nv; idx += 256) {
    const int e = idx / dv, c = (idx - e * dv) * 4;
    float4 acc = zero4();
    for (int i = start[e]; i < start[e + 1]; ++i) {
      const int j = order[i];
      const float w = ssc[j];
      const float4 a = ld4(A + ((p0 + j) / a_div) * lda + c);
      acc.x += w * a.x; acc.y += w * a.y; acc.z += w * a.z; acc.w += w * a.w;
    }
    st4(Z + g * (int64_t)E * d + (int64_t)idx * 4, acc);
  }
}

// dlogits[u, ids[p]] = gate[p]*(1-gate[p]) * <dOut[p / g_div, :], Y[p, :]>, the other entries of the row 0
__global__ __launch_bounds__(256) void gate_grad_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                        const int64_t* __restrict__ ids, const float* __restrict__ gate,
                                                        int64_t P, int k, int E, int N, int g_div, int v_div,
                                                        float* __restrict__ dlogits) {
  // 16 lanes per pair; v_div > 0: Y rows per (row group, expert) as in combine_kernel
  const int64_t p = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  float acc = 0.f;
  if (p < P) {
    const float* a = dOut + (p / g_div) * N;
    const float* y = Y + (v_div ? (p / v_div) * E + ids[p] : p) * N;
    for (int c = l * 4; c < N; c += 64) {
      const float4 u = ld4(a + c), v = ld4(y + c);
      acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
  }
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 4, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 1, 64);
  if (p < P && l == 0) {
    const float gt = gate[p];
    dlogits[(p / k) * E + ids[p]] = acc * gt * (1.f - gt);
  }
  // the entries of the unit's row that no slot selected are zero: written here by the lanes of the unit's first pair
  // (disjoint from the k selected entries, so no launch-wide memset and no ordering between the two kinds of stores)
  if (p < P && p % k == 0) {
    const int64_t u = p / k;
    for (int e = l; e < E; e += 16) {
      bool sel = false;
      for (int s2 = 0; s2 < k; ++s2) sel |= (int)ids[p + s2] == e;
      if (!sel) dlogits[u * E + e] = 0.f;
    }
  }
}

// Lists of the DISTINCT (row group, expert) combinations: group g = the fan pairs g*fan .. (one token's heads x slots in
// SwitchHead), mask[g] = the set of experts its pairs chose.  Where every pair of a group reads the same input row
// (moe_v: the token's row) or the products are summed over the group anyway (moe_out), the expert product is needed
// once per distinct (group, expert): 12.9 of 16 at E 32, h 8, top-2.  The lists use "virtual pair" ids g*E + e, so the
// grouped GEMMs run on them unchanged with a_div = E: input row g, output row g*E + e.
__global__ __launch_bounds__(256) void expert_mask_kernel(const int64_t* __restrict__ ids, int64_t P, int fan,
                                                          int groups_per_block, unsigned long long* __restrict__ mask, int64_t G) {
  // a block takes groups_per_block whole groups = consecutive pairs: one coalesced id per thread, OR-ed into the group's
  // mask in LDS
  __shared__ unsigned long long m[256];
  const int tid = threadIdx.x;
  m[tid] = 0;
  __syncthreads();
  const int64_t g0 = (int64_t)blockIdx.x * groups_per_block;
  const int span = groups_per_block * fan;
  for (int i = tid; i < span; i += 256) {
    const int64_t p = g0 * fan + i;
    if (p < P) atomicOr(&m[i / fan], 1ull << (int)ids[p]);
  }
  __syncthreads();
  if (tid < groups_per_block && g0 + tid < G) mask[g0 + tid] = m[tid];
}

// One workgroup per expert e.  Pass 1: offsets[e] = number of (group, expert') combinations with expert' < e, counted
// straight from the masks (no counters, no atomics); pass 2: the expert's groups in ascending order (ballot ranks inside
// a wave, wave totals through LDS).
__global__ __launch_bounds__(1024) void expert_lists_kernel(const unsigned long long* __restrict__ mask, int64_t G, int E,
                                                            int32_t* __restrict__ offsets, int32_t* __restrict__ perm) {
  __shared__ int wtot[16];
  __shared__ int wlow[16];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long low = (1ull << e) - 1ull;
  int below = 0, own = 0;
  for (int64_t g = tid; g < G; g += 1024) {
    const unsigned long long mk = mask[g];
    below += __builtin_popcountll(mk & low);
    own += (int)((mk >> e) & 1ull);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { below += __shfl_xor(below, o, 64); own += __shfl_xor(own, o, 64); }
  if (lane == 0) { wlow[wave] = below; wtot[wave] = own; }
  __syncthreads();
  int base = 0, count = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { base += wlow[w]; count += wtot[w]; }
  if (tid == 0) {
    offsets[e] = base;
    if (e == E - 1) offsets[E] = base + count;
  }
  __syncthreads();
  int running = base;
  for (int64_t g0 = 0; g0 < G; g0 += 1024) {
    const int64_t g = g0 + tid;
    const bool has = g < G && ((mask[g] >> e) & 1ull);
    const unsigned long long b = __ballot(has);
    if (lane == 0) wtot[wave] = __builtin_popcountll(b);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = wtot[w]; total += c; if (w < wave) before += c; }
    if (has) perm[running + before + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = (int32_t)(g * E + e);
    running += total;
    __syncthreads();
  }
}

}  // namespace amk_moe

using namespace amk_moe;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// workgroups of the wide kernels resident at once: two per CU (LDS-bound)
static int wg_slots() {
  static const int slots = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return 2 * cus;
  }();
  return slots;
}

extern "C" int64_t amk_moe_route_ws_ints(int64_t U, int E, int k) { return ((U * k + 255) / 256) * E; }

static void launch_topk(const float* logits, int64_t U, int E, int k, int64_t* ids, float* gate, hipStream_t st) {
  const dim3 tgrid((unsigned)((U + 127) / 128)), tblock(128);
  switch (k) {
    case 1: hipLaunchKernelGGL(route_topk_kernel<1>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
    case 2: hipLaunchKernelGGL(route_topk_kernel<2>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
    case 3: hipLaunchKernelGGL(route_topk_kernel<3>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
    case 4: hipLaunchKernelGGL(route_topk_kernel<4>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
    case 5: hipLaunchKernelGGL(route_topk_kernel<5>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
    case 6: hipLaunchKernelGGL(route_topk_kernel<6>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
    case 7: hipLaunchKernelGGL(route_topk_kernel<7>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
    default: hipLaunchKernelGGL(route_topk_kernel<8>, tgrid, tblock, 0, st, logits, U, E, ids, gate); break;
  }
}

extern "C" int amk_moe_topk(const float* logits, int64_t U, int E, int k, int64_t* ids, float* gate, void* stream) {
  AMK_CHECK_ARG(logits && ids && gate, "amk_moe_topk: null pointer");
  AMK_CHECK_ARG(U > 0 && E > 0 && k > 0 && k <= E, "amk_moe_topk: bad sizes U=%lld E=%d k=%d", (long long)U, E, k);
  AMK_CHECK_SUPPORTED(k <= MAX_K && U * k < (1ll << 31), "amk_moe_topk: sel_experts %d > %d or too many pairs", k, MAX_K);
  launch_topk(logits, U, E, k, ids, gate, static_cast<hipStream_t>(stream));
  AMK_CHECK_LAUNCH("amk_moe_topk");
  return AMK_OK;
}

extern "C" int amk_moe_route(const float* logits, int64_t U, int E, int k,
                             int64_t* ids, float* gate, int32_t* counts, int32_t* rank, int32_t* blockhist,
                             int32_t* offsets, int32_t* perm, void* stream) {
  AMK_CHECK_ARG(logits && ids && gate && counts && rank && blockhist && offsets && perm, "amk_moe_route: null pointer");
  AMK_CHECK_SUPPORTED(E <= 1024, "amk_moe_route: at most 1024 experts");
  AMK_CHECK_ARG(U > 0 && E > 0 && k > 0 && k <= E, "amk_moe_route: bad sizes U=%lld E=%d k=%d", (long long)U, E, k);
  AMK_CHECK_SUPPORTED(k <= MAX_K, "amk_moe_route: sel_experts %d > %d", k, MAX_K);
  const int64_t P = U * k;
  AMK_CHECK_SUPPORTED(P < (1ll << 31), "amk_moe_route: too many routed pairs");
  hipStream_t st = static_cast<hipStream_t>(stream);
  launch_topk(logits, U, E, k, ids, gate, st);
  const int nblk = (int)((P + ROUTE_BLOCK - 1) / ROUTE_BLOCK);
  hipLaunchKernelGGL(route_local_kernel, dim3(nblk), dim3(ROUTE_BLOCK), (size_t)(ROUTE_BLOCK / 64) * E * sizeof(int), st, ids, P, E, rank, blockhist);
  hipLaunchKernelGGL(route_scan_kernel, dim3(1), dim3(1024), 0, st, blockhist, nblk, E, counts, offsets);
  hipLaunchKernelGGL(route_perm_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, ids, offsets, blockhist, rank, P, E, perm);
  AMK_CHECK_LAUNCH("amk_moe_route");
  return AMK_OK;
}

static int check_gemm(const char* who, const void* A, const void* W, const void* Y, const void* offsets, const void* perm,
                      int64_t P, int E, int N, int Kd, int a_div, int64_t lda) {
  if (!(A && W && Y && offsets && perm)) { amk_set_error("%s: null pointer", who); return AMK_EINVAL; }
  if (!(P > 0 && E > 0 && N > 0 && Kd > 0 && a_div > 0)) { amk_set_error("%s: non-positive size", who); return AMK_EINVAL; }
  if ((N % 4) || (Kd % 4) || (lda % 4) || !a16(A) || !a16(W) || !a16(Y)) {
    amk_set_error("%s: N, Kd, lda must be multiples of 4 and pointers 16-byte aligned", who);
    return AMK_EUNSUPPORTED;
  }
  if (P >= (1ll << 31)) { amk_set_error("%s: too many pairs", who); return AMK_EUNSUPPORTED; }
  return AMK_OK;
}

static int grouped_nt_impl(const char* who, const float* A, int64_t lda, int a_div, const float* W, const float* bias,
                           const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                           float* Y, int y_div, void* stream) {
  const int rc = check_gemm(who, A, W, Y, offsets, perm, P, E, N, Kd, a_div, lda);
  if (rc) return rc;
  GemmParams g{};
  g.A = A; g.W = W; g.bias = bias; g.Y = Y; g.offsets = offsets; g.perm = perm;
  g.E = E; g.N = N; g.Kd = Kd; g.a_div = a_div; g.b_div = 1; g.lda = lda; g.y_div = y_div;
  const unsigned mt = (unsigned)((P + 63) / 64 + E);
  AMK_CHECK_SUPPORTED((uint64_t)mt * ((N + 63) / 64) < (1ull << 31), "%s: grid too large", who);
  g.a_bytes = ((P - 1) / a_div * lda + Kd) * 4;   // rows 0 .. (P-1)/a_div of A
  g.y_bytes = (y_div > 0 ? (P - 1) / y_div + 1 : P) * N * 4;
  const bool wide = N >= 128 && Kd % 32 == 0 && g.a_bytes < (1ll << 31) && g.y_bytes < 0x7FFF0000ll && (int64_t)N * Kd * 4 < (1ll << 31);
  AMK_CHECK_SUPPORTED(y_div == 0 || wide, "%s: the accumulating form needs N >= 128, Kd %% 32 == 0 and buffers below 2 GB", who);
  if (wide && (y_div > 0 || !getenv("AMK_MOE_NARROW")))
    { g.ncol = (N + 127) / 128; g.slots = wg_slots();  // grid: the bound for two-block tiles; the surplus workgroups leave at once
      hipLaunchKernelGGL(grouped_nt_wide_kernel, dim3((unsigned)((P + 63) / 64 + E) * g.ncol + (unsigned)g.slots / 2), dim3(256), 0, static_cast<hipStream_t>(stream), g); }
  else
    { AMK_CHECK_SUPPORTED(g.a_bytes < 0x7fff0000ll && (int64_t)N * Kd * 4 < 0x7fff0000ll, "%s: A and one expert's weights must span < 2 GiB", who);
      g.ncol = (N + 63) / 64;
      // few tiles per CU (all resident at once): 32-pair tiles, twice as many workgroups half as long (AMK_MOE_SHORT=0: never)
      static const bool short_ok = !(getenv("AMK_MOE_SHORT") && atoi(getenv("AMK_MOE_SHORT")) == 0);
      if (short_ok && N <= 64 && Kd >= 256 && (P + 63) / 64 < 16 * wg_slots() / 2)
        hipLaunchKernelGGL((grouped_nt_kernel<1, true>), dim3(2 * mt * g.ncol), dim3(256), 0, static_cast<hipStream_t>(stream), g);
      else
        hipLaunchKernelGGL(grouped_nt_kernel<1>, dim3(mt * g.ncol), dim3(256), 0, static_cast<hipStream_t>(stream), g); }
  AMK_CHECK_LAUNCH(who);
  return AMK_OK;
}

extern "C" int amk_grouped_gemm_nt(const float* A, int64_t lda, int a_div, const float* W, const float* bias,
                                   const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                                   float* Y, void* stream) {
  return grouped_nt_impl("amk_grouped_gemm_nt", A, lda, a_div, W, bias, offsets, perm, P, E, N, Kd, Y, 0, stream);
}

extern "C" int amk_grouped_gemm_nt_acc(const float* A, int64_t lda, int a_div, const float* W, const float* bias,
                                       const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                                       float* Y, int y_div, void* stream) {
  AMK_CHECK_ARG(y_div > 0, "amk_grouped_gemm_nt_acc: y_div must be positive");
  return grouped_nt_impl("amk_grouped_gemm_nt_acc", A, lda, a_div, W, bias, offsets, perm, P, E, N, Kd, Y, y_div, stream);
}

static int grouped_nn_impl(const char* who, const float* A, int64_t lda, int a_div, const float* W, const float* scale,
                           const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                           float* Y, int y_div, void* stream) {
  const int rc = check_gemm(who, A, W, Y, offsets, perm, P, E, N, Kd, a_div, lda);
  if (rc) return rc;
  GemmParams g{};
  g.A = A; g.W = W; g.scale = scale; g.Y = Y; g.offsets = offsets; g.perm = perm;
  g.E = E; g.N = N; g.Kd = Kd; g.a_div = a_div; g.b_div = 1; g.lda = lda; g.y_div = y_div;
  const unsigned mt = (unsigned)((P + 63) / 64 + E);
  AMK_CHECK_SUPPORTED((uint64_t)mt * ((Kd + 127) / 128) < (1ull << 31), "%s: grid too large", who);
  g.a_bytes = ((P - 1) / a_div * lda + N) * 4;   // rows 0 .. (P-1)/a_div of A
  g.y_bytes = (y_div > 0 ? (P - 1) / y_div + 1 : P) * Kd * 4;
  const bool wide = Kd >= 128 && Kd % 4 == 0 && N % 32 == 0 && g.a_bytes < (1ll << 31) && g.y_bytes < 0x7FFF0000ll && (int64_t)N * Kd * 4 < (1ll << 31);
  AMK_CHECK_SUPPORTED(y_div == 0 || wide, "%s: the accumulating form needs Kd >= 128, N %% 32 == 0 and buffers below 2 GB", who);
  if (wide && (y_div > 0 || !getenv("AMK_MOE_NARROW"))) {
    g.ncol = (Kd + 127) / 128; g.slots = wg_slots();
    hipLaunchKernelGGL(grouped_nn_wide_kernel, dim3(mt * g.ncol + (unsigned)g.slots / 2), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  } else if (g.a_bytes >= 0x7fff0000ll || (int64_t)N * Kd * 4 >= 0x7fff0000ll) {
    amk_set_error("%s: A and one expert's weights must span < 2 GiB", who);
    return AMK_EUNSUPPORTED;
  } else if (Kd > 64) { g.ncol = (Kd + 127) / 128; hipLaunchKernelGGL(grouped_nn_kernel<2>, dim3(mt * g.ncol), dim3(256), 0, static_cast<hipStream_t>(stream), g); }
  else {
    g.ncol = 1;
    static const bool short_ok = !(getenv("AMK_MOE_SHORT") && atoi(getenv("AMK_MOE_SHORT")) == 0);
    if (short_ok && N >= 256 && (P + 63) / 64 < 16 * wg_slots() / 2)
      hipLaunchKernelGGL((grouped_nn_kernel<1, true>), dim3(2 * mt), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    else
      hipLaunchKernelGGL(grouped_nn_kernel<1>, dim3(mt), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  }
  AMK_CHECK_LAUNCH(who);
  return AMK_OK;
}

extern "C" int amk_grouped_gemm_nn(const float* A, int64_t lda, int a_div, const float* W, const float* scale,
                                   const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                                   float* Y, void* stream) {
  return grouped_nn_impl("amk_grouped_gemm_nn", A, lda, a_div, W, scale, offsets, perm, P, E, N, Kd, Y, 0, stream);
}

extern "C" int amk_grouped_gemm_nn_acc(const float* A, int64_t lda, int a_div, const float* W, const float* scale,
                                       const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                                       float* Y, int y_div, void* stream) {
  AMK_CHECK_ARG(y_div > 0, "amk_grouped_gemm_nn_acc: y_div must be positive");
  return grouped_nn_impl("amk_grouped_gemm_nn_acc", A, lda, a_div, W, scale, offsets, perm, P, E, N, Kd, Y, y_div, stream);
}

extern "C" int amk_grouped_gemm_wgrad(const float* G, int64_t ldg, int g_div, const float* X, int64_t ldx, int x_div,
                                      const float* scale, const int32_t* offsets, const int32_t* perm,
                                      int64_t P, int E, int N, int Kd, float* dW, float* dbias, void* stream) {
  const int rc = check_gemm("amk_grouped_gemm_wgrad", G, X, dW, offsets, perm, P, E, N, Kd, g_div, ldg);
  if (rc) return rc;
  AMK_CHECK_ARG(x_div > 0 && (ldx % 4) == 0 && a16(X), "amk_grouped_gemm_wgrad: bad X addressing");
  AMK_CHECK_SUPPORTED(E <= 65535 && (N + 63) / 64 <= 65535, "amk_grouped_gemm_wgrad: grid too large");
  GemmParams g{};
  g.A = G; g.B2 = X; g.scale = scale; g.Y = dW; g.dbias = dbias; g.offsets = offsets; g.perm = perm;
  g.E = E; g.N = N; g.Kd = Kd; g.a_div = g_div; g.b_div = x_div; g.lda = ldg; g.ldb = ldx;
  g.a_bytes = ((P - 1) / g_div * ldg + N) * 4;   // G buffer
  g.b_bytes = ((P - 1) / x_div * ldx + Kd) * 4;  // X buffer
  g.s_bytes = P * 4;
  auto log2_of = [](int d) { int sft = 0; while ((1 << sft) < d) ++sft; return (1 << sft) == d ? sft : -1; };
  g.a_shift = log2_of(g_div); g.b_shift = log2_of(x_div);
  if (N >= 64 && Kd >= 64 && N + Kd >= 192 && g.a_bytes < 0x7FFF0000ll && g.b_bytes < 0x7FFF0000ll && g.s_bytes < (1ll << 31) &&
      (int64_t)N * Kd * 4 < 0x7FFF0000ll && !getenv("AMK_MOE_NARROW")) {
    // tile 128 x 128; 64 x 128 / 128 x 64 for SwitchHead's (64 x D) / (D x 64) experts
    const int tn = N >= 128 ? 128 : 64, tk = Kd >= 128 ? 128 : 64;
    g.ncol = (Kd + tk - 1) / tk;
    int64_t nwg = (int64_t)E * ((N + tn - 1) / tn) * g.ncol;
    AMK_CHECK_SUPPORTED(nwg < (1ll << 30), "amk_grouped_gemm_wgrad: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // fewer tiles than workgroup slots and long experts: the pairs of a tile in two halves that add into a zeroed dW
    // (AMK_MOE_WGRAD_SPLIT=0: never)
    static const bool split_ok = !(getenv("AMK_MOE_WGRAD_SPLIT") && atoi(getenv("AMK_MOE_WGRAD_SPLIT")) == 0);
    g.rsplit = 1;
    if (split_ok && 2 * nwg <= wg_slots() && P / E >= 256) {
      g.rsplit = 2;
      nwg *= 2;
      if (hipMemsetAsync(dW, 0, (size_t)E * N * Kd * sizeof(float), st) != hipSuccess ||
          (dbias && hipMemsetAsync(dbias, 0, (size_t)E * N * sizeof(float), st) != hipSuccess)) {
        amk_set_error("amk_grouped_gemm_wgrad: hipMemsetAsync failed");
        return AMK_ELAUNCH;
      }
    }
    const dim3 grid((unsigned)nwg), block(256);
#define AMK_WGRAD(S, TN_, TK_) hipLaunchKernelGGL((grouped_wgrad_wide_kernel<S, TN_, TK_>), grid, block, 0, st, g)
    if (tn == 128 && tk == 128) { if (scale) AMK_WGRAD(true, 2, 2); else AMK_WGRAD(false, 2, 2); }
    else if (tn == 64) { if (scale) AMK_WGRAD(true, 1, 2); else AMK_WGRAD(false, 1, 2); }
    else { if (scale) AMK_WGRAD(true, 2, 1); else AMK_WGRAD(false, 2, 1); }
#undef AMK_WGRAD
    AMK_CHECK_LAUNCH("amk_grouped_gemm_wgrad");
    return AMK_OK;
  }
  // same shape: NB 1 0.218 ms, NB 2 0.231 ms
  hipLaunchKernelGGL(grouped_wgrad_kernel<1>, dim3((Kd + 63) / 64, (N + 63) / 64, E), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g);
  AMK_CHECK_LAUNCH("amk_grouped_gemm_wgrad");
  return AMK_OK;
}

extern "C" int amk_moe_combine_rows(const float* Y, const int64_t* ids, const float* scale, int64_t G, int outer, int k,
                                    int N, int v_div, int E, float* out, void* stream) {
  AMK_CHECK_ARG(Y && ids && out, "amk_moe_combine: null pointer");
  AMK_CHECK_ARG(G > 0 && outer > 0 && k > 0 && N > 0 && v_div >= 0 && (v_div == 0 || E > 0), "amk_moe_combine: non-positive size");
  AMK_CHECK_SUPPORTED(k <= MAX_K && N % 4 == 0 && a16(Y) && a16(out), "amk_moe_combine: k <= %d, N %% 4 == 0, aligned pointers", MAX_K);
  const int64_t n = G * (N / 4);
  AMK_CHECK_SUPPORTED((n + 255) / 256 < (1ll << 31), "amk_moe_combine: grid too large");
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     Y, ids, scale, G, outer, k, N, v_div, E, out);
  AMK_CHECK_LAUNCH("amk_moe_combine");
  return AMK_OK;
}

extern "C" int amk_moe_combine(const float* Y, const int64_t* ids, const float* scale, int64_t G, int outer, int k,
                               int N, float* out, void* stream) {
  return amk_moe_combine_rows(Y, ids, scale, G, outer, k, N, 0, 0, out, stream);
}

extern "C" int amk_moe_route_distinct(const int64_t* ids, int64_t G, int fan, int E, uint64_t* mask,
                                      int32_t* offsets, int32_t* perm, void* stream) {
  AMK_CHECK_ARG(ids && mask && offsets && perm, "amk_moe_route_distinct: null pointer");
  AMK_CHECK_ARG(G > 0 && fan > 0 && E > 0, "amk_moe_route_distinct: non-positive size");
  AMK_CHECK_SUPPORTED(E <= 64 && G * E < (1ll << 31) && fan <= 4096, "amk_moe_route_distinct: at most 64 experts, G*E < 2^31, fan <= 4096");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int gpb = fan >= 256 ? 1 : 256 / fan;   // groups per block of the mask kernel
  hipLaunchKernelGGL(expert_mask_kernel, dim3((unsigned)((G + gpb - 1) / gpb)), dim3(256), 0, st, ids, G * fan, fan, gpb,
                     reinterpret_cast<unsigned long long*>(mask), G);
  hipLaunchKernelGGL(expert_lists_kernel, dim3((unsigned)E), dim3(1024), 0, st,
                     reinterpret_cast<const unsigned long long*>(mask), G, E, offsets, perm);
  AMK_CHECK_LAUNCH("amk_moe_route_distinct");
  return AMK_OK;
}

extern "C" int amk_moe_expert_sums(const float* A, int64_t lda, int a_div, const int64_t* ids, const float* scale,
                                   int64_t G, int fan, int E, int d, float* Z, void* stream) {
  AMK_CHECK_ARG(A && ids && Z, "amk_moe_expert_sums: null pointer");
  AMK_CHECK_ARG(G > 0 && fan > 0 && E > 0 && d > 0 && a_div > 0 && lda >= d, "amk_moe_expert_sums: non-positive size");
  AMK_CHECK_SUPPORTED(d % 4 == 0 && lda % 4 == 0 && a16(A) && a16(Z) && fan <= 4096 && E <= 4096 && G < (1ll << 31),
                      "amk_moe_expert_sums: d, lda multiples of 4, aligned pointers, fan and E <= 4096");
  hipLaunchKernelGGL(expert_sums_kernel, dim3((unsigned)G), dim3(256), ((size_t)fan * 3 + E + 1) * 4, static_cast<hipStream_t>(stream),
                     A, lda, a_div, ids, scale, fan, E, d, Z);
  AMK_CHECK_LAUNCH("amk_moe_expert_sums");
  return AMK_OK;
}

extern "C" int amk_moe_gate_grad_rows(const float* d_out, const float* Y, const int64_t* ids, const float* gate,
                                      int64_t P, int k, int E, int N, int g_div, int v_div, float* dlogits, void* stream) {
  AMK_CHECK_ARG(d_out && Y && ids && gate && dlogits, "amk_moe_gate_grad: null pointer");
  AMK_CHECK_ARG(P > 0 && k > 0 && E > 0 && N > 0 && g_div > 0 && v_div >= 0, "amk_moe_gate_grad: non-positive size");
  AMK_CHECK_SUPPORTED(N % 4 == 0 && a16(d_out) && a16(Y), "amk_moe_gate_grad: N %% 4 == 0, aligned pointers");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gate_grad_kernel, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, st, d_out, Y, ids, gate, P, k, E, N,
                     g_div, v_div, dlogits);
  AMK_CHECK_LAUNCH("amk_moe_gate_grad");
  return AMK_OK;
}

extern "C" int amk_moe_gate_grad(const float* d_out, const float* Y, const int64_t* ids, const float* gate,
                                 int64_t P, int k, int E, int N, int g_div, float* dlogits, void* stream) {
  return amk_moe_gate_grad_rows(d_out, Y, ids, gate, P, k, E, N, g_div, 0, dlogits, stream);
}
