// Fused softmax-attention backward for gfx950: ONE pass, dQ by f32 atomics.
//
// The two-kernel backward of attn_bwd.hip recomputes S twice and dP twice (7 products, bitwise
// reproducible, no atomics).  This kernel computes each product once (cdna guide, Appendix B
// "Attention backward"): a workgroup (NW waves) owns 32*NW keys of one (batch, head) -- v of a
// wave's 32 keys lives in registers, the K block in LDS, dK^T / dV^T accumulate in registers -- and sweeps the
// queries in tiles of 32:
//
//   S  = Q K^T                  32x32x2 MFMA (A = q rows from LDS, B = k rows from LDS) -- or, KEPT = true,
//                               read back from the 32x32 score tiles the forward left in HBM
//                               (amk_attn_fwd_keep; bitwise the same numbers): four products instead of five
//   dP = dO V^T                 32x32x2 MFMA, A = dO rows from LDS, B = v registers
//   P, dS                       key on the lane: fills are per-lane constants, row stats from LDS
//   dV^T += dO^T P, dK^T += Q^T dS   A = dO / q columns from LDS, B = the accumulators as they stand
//   dS -> LDS ([query][key], all keys of the workgroup), one barrier, then
//   dQ tile (32 x 64) = dS (32 x 32NW) K (32NW x 64) with v_mfma_f32_16x16x4_f32: every wave owns 8/NW
//        16x16 output blocks over ALL keys of the workgroup, so no cross-wave reduction; the result is added
//        to dq with global_atomic_add_f32 (dq is zeroed by the launcher).  Atomic volume is
//        8 KiB per (32 query x 32NW key) tile: every dq element receives J / (32 NW) adds.
// Two workgroup barriers per query tile (the next q / dO tile is committed under the dQ product);
// workgroups start at rotated query tiles so that co-resident ones are not in the same phase.
// Results differ from run to run in the last bits of dq only (f32 atomic arrival order); dk, dv are
// reproducible.  The two-kernel path stays available (AMK_ATTN_BWD_DKDV | AMK_ATTN_BWD_DQ).
#include "attn_common.h"
#include <stdlib.h>

// diagnostic builds only (tools/ablate_attn_bwd.sh): AMK_BWD_ABL bits switch parts of the tile loop off -- WRONG results,
// timing only: 1 no dq atomics / stores, 2 no dQ product, 4 no dS LDS writes, 8 no P / dS arithmetic, 16 no kept-score
// loads, 32 no dV / dK column reads (operands reused), 64 no q / dO staging after the first tile
#ifndef AMK_BWD_ABL
#define AMK_BWD_ABL 0
#endif

namespace amk_attn {

constexpr int TQ = 32;             // queries per tile

template <int NW>
struct FusedGeom {
  static constexpr int NT = 64 * NW;          // threads
  static constexpr int KB = 32 * NW;          // keys per workgroup
  static constexpr int DS_STRIDE = KB + 4;    // dS tile row stride (floats): 16-B aligned rows, b128 row reads
  static constexpr int LDS_FLOATS = 2 * TQ * LDS_STRIDE + KB * LDS_STRIDE + TQ * DS_STRIDE + 4 * TQ;
  static constexpr int NBLK = 8 / NW;         // 16x16 dQ blocks per wave
  static constexpr int KPG = KB / 4;          // keys per k-group of the 16x16x4 product
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  // D(16x16) += A(16x4) B(4x16): lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15];
  // accumulator register r of lane l is D[4 * (l >> 4) + r][l & 15].
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// DQ: 0 = f32 atomic adds into a zeroed dq (fastest; dq differs in the last bits from run to run);
//     1 = plain stores -- of dq itself when the workgroup holds every key of its (batch, head) (J <= 32 NW: no
//         atomics, no memset), else of this key block's partial into p.dq_part[kb], summed in key-block order
//         by attn_bwd_dq_reduce_kernel: bitwise reproducible.
// CAUSAL: p.causal_mask, an (I, J) byte mask shared by batch and heads (models/softmax_attention.py:62-66): a non-zero
// byte puts the fill value in place of the score (its dS is 0, its P still feeds dV), as the forward does
template <int NW, bool KEPT, int DQ, bool CAUSAL>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_fused_kernel(BwdParams p) {
  using G = FusedGeom<NW>;
  constexpr int NT = G::NT, KB = G::KB, DS_STRIDE = G::DS_STRIDE, KPG = G::KPG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                           // [TQ][LDS_STRIDE]  (q * scale * log2 e)
  float* Gs = Qs + TQ * LDS_STRIDE;           // [TQ][LDS_STRIDE]  dO
  float* Kc = Gs + TQ * LDS_STRIDE;           // [KB][LDS_STRIDE]  this workgroup's K rows
  float* dSl = Kc + KB * LDS_STRIDE;          // [TQ][DS_STRIDE]   dS of the current tile
  float* Ms = dSl + TQ * DS_STRIDE;
  float* Ls = Ms + TQ;
  float* Ds = Ls + TQ;    // MINUS delta: the initial accumulator of the dP chain (dP - delta leaves the matrix pipe)
  float* MLs = Ds + TQ;   // m + log2 l: P = exp2(S - MLs) in one subtraction where no fill can occur (plain tiles)

  const int tid = threadIdx.x;
  // wave-uniform by construction; readfirstlane tells the compiler (scalar registers, and no waterfall
  // loop around the loads through the per-wave score-tile descriptor below)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 31, hf = lane >> 5;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = wg % p.nkblk;
  const int bh = wg / p.nkblk;
  const int h = bh % p.H, b = bh / p.H;
  const int kj = kb * KB + wave * 32 + ln;  // this lane's key row
  const bool kvalid = kj < p.J;

  const float* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;

  // B operand held for the whole kernel: v of this lane's key (k comes from the K block in LDS).
  float vreg[32];
  {
    const float* vp = vbase + (int64_t)kj * p.vs.st + 32 * hf;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const float4 c = kvalid ? ld4(vp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      vreg[4 * s4 + 0] = c.x; vreg[4 * s4 + 1] = c.y; vreg[4 * s4 + 2] = c.z; vreg[4 * s4 + 3] = c.w;
    }
  }
  float kfill = 0.f;  // 0 keep, -1e9*log2e masked key, -inf beyond the sequence
  if (!kvalid) kfill = -INFINITY;
  else if (p.key_mask && p.key_mask[(int64_t)b * p.J + kj] == 0) kfill = AMK_FILL_MASKED;
  const bool plain = !CAUSAL && __all(kfill == 0.f);  // wave-uniform: none of this wave's scores is filled
  // CAUSAL: this lane's key column of the mask, one byte per query row of its 16 registers; the bytes of the coming
  // tile are requested a tile ahead (unconditional loads at clamped rows: no branch around memory instructions)
  const uint8_t* cm_col = CAUSAL ? p.causal_mask + min(kj, p.J - 1) : nullptr;
  unsigned cm_raw[16];
  auto load_cmask = [&](int i0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cm_raw[r] = cm_col[(int64_t)min(i0 + acc_row(r, hf), p.I - 1) * p.J];
  };

  // K rows of the whole workgroup -> LDS once (B operand of the dQ product, and of S when it is recomputed)
  {
    RowStagerT<KB / 2, NT> kl;  // two passes of KB/2 rows, 4 x 16 B per thread each
    const float* kblk = kbase + (int64_t)kb * KB * p.ks.st;
    kl.init(kblk, p.ks.st, min(KB, p.J - kb * KB), tid);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 t[4];
      kl.load(t);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) st4(&Kc[((KB / 2) * half + (tid >> 4) + (NT / 16) * ps) * LDS_STRIDE + (tid & 15) * 4], t[ps]);
    }
  }

  const float* qbase = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* gbase = p.d_o + (int64_t)b * p.dos.sb + (int64_t)h * p.dos.sh;
  const float* stbase = p.stats + ((int64_t)b * p.H + h) * p.I * 2;
  const float* dlbase = p.delta + ((int64_t)b * p.H + h) * p.I;
  float* dqbase = (DQ == 1 && p.nkblk > 1 ? p.dq_part + (int64_t)kb * p.B * p.I * p.H * D : p.dq) +
                  (int64_t)b * p.dqs.sb + (int64_t)h * p.dqs.sh;
  // dq rows of this (batch, head) through a buffer descriptor: adds to rows beyond the sequence are
  // dropped by the hardware range check, so the atomics need no branch either
  const __amdgpu_buffer_rsrc_t dq_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)dqbase, 0, (int)(((int64_t)(p.I - 1) * p.dqs.st + 64) * 4), 0x00020000);

  // kept scores: this wave's 32 keys x all query tiles are nqt consecutive 4-KiB tiles; a buffer
  // descriptor of zero records (key block beyond the forward's tiles) reads zeros
  __amdgpu_buffer_rsrc_t sc_rsrc;
  int sc_voff = 0;
  if (KEPT) {
    const ScoreTiles stl(p.I, p.J);
    const int kb32 = kb * NW + wave;
    const float* tiles = p.scores + ((int64_t)bh * stl.nkb + min(kb32, stl.nkb - 1)) * stl.nqt * 1024;
    sc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tiles, 0, kb32 < stl.nkb ? stl.nqt * 4096 : 0, 0x00020000);
    sc_voff = ln * 128 + hf * 16;  // row = this lane's key, 4 queries of register group g at +32 g bytes
  }
  float4 sk[4];  // KEPT: the score registers of the coming tile (16 queries of this lane's key)
  auto load_scores = [&](int qt) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      sk[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sc_rsrc, sc_voff + qt * 4096 + 32 * g, 0, 2));
  };

  constexpr int QNP = RowStagerT<TQ, NT>::NP;  // 16-B pieces per thread of a 32-row tile (2 or 1)
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  float4 qst[QNP], gst[QNP];
  RowStagerT<TQ, NT> qload, gload;
  qload.init(qbase, p.qs.st, p.I, tid);
  gload.init(gbase, p.dos.st, p.I, tid);
  // No branch around any vector-memory instruction of the tile loop: the compiler can then count, at
  // every wait, how many younger loads / atomics are in flight (s_waitcnt vmcnt(N) with N > 0); behind a
  // branch it must assume none and waits for everything, the dq atomics of the previous tile included.
  float2 ml_raw = make_float2(0.f, 1.f);
  float dl_raw = 0.f;
  bool row_ok = false;
  auto prefetch = [&](int i0) {  // loads only: whatever consumes them waits in commit(), a tile later
    qload.load(qst);
    gload.load(gst);
    const int i = i0 + (tid & (TQ - 1));
    const int ic = min(i, p.I - 1);
    ml_raw = *reinterpret_cast<const float2*>(stbase + 2 * ic);
    dl_raw = dlbase[ic];
    row_ok = i < p.I;
  };
  auto commit = [&]() {
    // Nothing below may be scheduled up to the loads (the compiler likes to): next to them it would sit
    // on the memory latency; here the loads were issued a tile ago.
    __builtin_amdgcn_sched_barrier(0);
    const float sc = p.scale * AMK_LOG2E;  // S comes out in the log2 domain; dK is scaled back by ln 2
#pragma unroll
    for (int ps = 0; ps < QNP; ++ps) {
      const int r = srow + (NT / 16) * ps;
      st4(&Qs[r * LDS_STRIDE + scol], make_float4(qst[ps].x * sc, qst[ps].y * sc, qst[ps].z * sc, qst[ps].w * sc));
      st4(&Gs[r * LDS_STRIDE + scol], gst[ps]);
    }
    // rows beyond the sequence: P = exp2(x - inf) * 0 = 0
    if (tid < TQ) {  // LDS stores only
      Ms[tid] = row_ok ? ml_raw.x : INFINITY;
      Ls[tid] = row_ok ? 1.f / ml_raw.y : 0.f;
      Ds[tid] = row_ok ? -dl_raw : 0.f;
      MLs[tid] = row_ok ? ml_raw.x + __builtin_amdgcn_logf(ml_raw.y) : INFINITY;   // (v_log_f32 is log2)
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

  // dQ product geometry: 16x16x4 MFMA, lane = (column c = l & 15, k-group kg = l >> 4)
  const int c16 = lane & 15, kg = lane >> 4;
  const int qhalf = wave & 1;                          // which 16 of the tile's 32 queries
  const int dcol0 = (16 * G::NBLK) * (wave >> 1);      // this wave's output columns (NBLK 16-wide blocks)
  const float* ds_row = &dSl[(16 * qhalf + c16) * DS_STRIDE + KPG * kg];  // A: dS[query][KPG*kg + s]
  const float* kc_col = &Kc[(KPG * kg) * LDS_STRIDE + dcol0 + c16];       // B: K[KPG*kg + s][dcol]

  const int ntile = (p.I + TQ - 1) / TQ;
  // Two barriers per query tile.  After the dS barrier every wave is past the phases that read the
  // q / dO / stats tiles, so the next tile is committed there -- under the dQ product, which only
  // reads dS and K -- and the barrier that closes the dQ product also publishes it.
  // Workgroups of different (batch, head) slices start at different query tiles (the order of the dQ atomics and
  // of the dK / dV accumulation does not matter): it keeps the MFMA and barrier phases of co-resident workgroups
  // from lining up.  The key blocks of ONE slice start at the same tile and walk q / dO in step -- they sit on one
  // XCD (xcd_remap), so a tile is fetched from HBM once and the other key blocks find it in the L2: FETCH_SIZE of
  // the kernel fell from 859 MB to 662 MB per launch (recompute variant: 227 -> 131 MB = the operands read once)
  // against a start tile per workgroup.
  const int rot = (int)(((unsigned)bh * 5u) % (unsigned)ntile);
  auto tile_of = [&](int t) { const int x = t + rot; return x >= ntile ? x - ntile : x; };
  qload.seek(rot, p.qs.st, tid);
  gload.seek(rot, p.dos.st, tid);
  prefetch(rot * TQ);
  if (KEPT) load_scores(rot);
  if (CAUSAL) load_cmask(rot * TQ);
  __syncthreads();  // the K block is in LDS
  commit();
  __syncthreads();
  {
    const int nx = tile_of(min(1, ntile - 1));
    qload.seek(nx, p.qs.st, tid);
    gload.seek(nx, p.dos.st, tid);
    prefetch(nx * TQ);
  }
  // Enter the loop with nothing in flight: the waits inside are then the back edge's counted ones
  // (one memory latency per workgroup here; vmcnt(0) with expcnt / lgkmcnt left alone).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int t = 0; t < ntile; ++t) {
    const int i0 = tile_of(t) * TQ;

    // ---- S and dP for the tile's 32 queries x this wave's 32 keys
    // dP's chain starts from -delta of the register's query row: the subtraction of softmax's backward costs no VALU
    f32x16 s = zero16(), dp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 d4 = ld4(&Ds[8 * g + 4 * hf]);
      dp[4 * g + 0] = d4.x; dp[4 * g + 1] = d4.y; dp[4 * g + 2] = d4.z; dp[4 * g + 3] = d4.w;
    }
    {
      const float* qr = &Qs[ln * LDS_STRIDE + 32 * hf];
      const float* gr = &Gs[ln * LDS_STRIDE + 32 * hf];
      const float* kr = &Kc[(32 * wave + ln) * LDS_STRIDE + 32 * hf];  // this lane's key row
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        const float4 c = ld4(gr + 4 * s4);
        if (!KEPT) {
          const float4 a = ld4(qr + 4 * s4);
          const float4 kk = ld4(kr + 4 * s4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s = mfma32(f4(a, e), f4(kk, e), s);
            dp = mfma32(f4(c, e), vreg[4 * s4 + e], dp);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dp = mfma32(f4(c, e), vreg[4 * s4 + e], dp);
        }
      }
    }
    if (KEPT && !(AMK_BWD_ABL & 16)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s[4 * g + 0] = sk[g].x; s[4 * g + 1] = sk[g].y; s[4 * g + 2] = sk[g].z; s[4 * g + 3] = sk[g].w;
      }
    }
    // ---- P and dS (register r of this lane is query acc_row(r, hf))
    if (AMK_BWD_ABL & 8) {
    } else if (plain) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // (no fill in this wave's columns, so every row has a live key here and m is a genuine score maximum: the
        // normaliser folds into the exponent, P = exp2(S - (m + log2 l)); the two-constant form below is for rows whose
        // m may be the fill value, where m + log2 l would swallow l)
        const float4 ml4 = ld4(&MLs[8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = __builtin_amdgcn_exp2f(s[r] - f4(ml4, e));
          s[r] = pr;
          dp[r] = pr * dp[r];
        }
      }
    } else {
      unsigned cbits = 0xffffu;   // bit r: the score of register r is kept
      if (CAUSAL) {
        cbits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) cbits |= (cm_raw[r] == 0 ? 1u : 0u) << r;   // (a non-zero byte masks)
      }
      const float fillv = kfill != 0.f ? kfill : AMK_FILL_MASKED;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 m4 = ld4(&Ms[8 * g + 4 * hf]);
        const float4 l4 = ld4(&Ls[8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const bool filled = kfill != 0.f || !((cbits >> r) & 1u);
          const float tt = filled ? fillv : s[r];
          const float pr = __builtin_amdgcn_exp2f(tt - f4(m4, e)) * f4(l4, e);
          s[r] = pr;
          dp[r] = filled ? 0.f : pr * dp[r];
        }
      }
    }
    // ---- dS -> LDS as [query][key] for the workgroup-wide dQ product
    if (!(AMK_BWD_ABL & 4)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dSl[acc_row(r, hf) * DS_STRIDE + 32 * wave + ln] = dp[r];
    }

    // the next tile's scores (consumed after the next tile's dP product: three MFMA phases from here;
    // the last iteration re-reads its own tile, unused)
    if (KEPT && !(AMK_BWD_ABL & 16)) {
      load_scores(tile_of(min(t + 1, ntile - 1)));
      __builtin_amdgcn_sched_barrier(0);  // issued HERE: the compiler otherwise sinks them below the MFMAs
    }
    if (CAUSAL) load_cmask(tile_of(min(t + 1, ntile - 1)) * TQ);

    // ---- dV^T += dO^T P ; dK^T += (q*scale*log2e)^T dS   (2 x 32 MFMAs)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* gc = &Gs[acc_row((AMK_BWD_ABL & 32) ? 0 : r, hf) * LDS_STRIDE + ln];
      const float* qc = &Qs[acc_row((AMK_BWD_ABL & 32) ? 0 : r, hf) * LDS_STRIDE + ln];
      dv0 = mfma32(gc[0], s[r], dv0);
      dv1 = mfma32(gc[32], s[r], dv1);
      dk0 = mfma32(qc[0], dp[r], dk0);
      dk1 = mfma32(qc[32], dp[r], dk1);
    }
    __syncthreads();  // every wave's dS columns are in LDS; the q / dO / stats tiles are dead
    if (!(AMK_BWD_ABL & 64)) {
    commit();         // (the last iteration commits a tile nobody reads)
    {
      const int nx = tile_of(min(t + 2, ntile - 1));
      qload.seek(nx, p.qs.st, tid);
      gload.seek(nx, p.dos.st, tid);
      prefetch(nx * TQ);
    }
    }

    // ---- dQ (16 queries x 16*NBLK columns per wave) = dS (16 x KB) K (KB x 16*NBLK): 64 MFMAs 16x16x4,
    //      two independent accumulator chains either way (two blocks, or even / odd k-steps of one)
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < ((AMK_BWD_ABL & 2) ? 0 : KPG / 4); ++s4) {
      const float4 a = ld4(ds_row + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* kc = kc_col + (4 * s4 + e) * LDS_STRIDE;
        if (G::NBLK == 2) {
          q0 = mfma16(f4(a, e), kc[0], q0);
          q1 = mfma16(f4(a, e), kc[16], q1);
        } else if (e & 1) {
          q1 = mfma16(f4(a, e), kc[0], q1);
        } else {
          q0 = mfma16(f4(a, e), kc[0], q0);
        }
      }
    }
    {
      const float sc = p.scale;
      const int qi0 = i0 + 16 * qhalf + 4 * kg;
      const int off = (int)(((int64_t)qi0 * p.dqs.st + dcol0 + c16) * 4);
      const int rstep = (int)(p.dqs.st * 4);
#pragma unroll
      for (int r = 0; r < ((AMK_BWD_ABL & 1) ? 0 : 4); ++r) {
        if (DQ == 0) {
          if (G::NBLK == 2) {
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(q0[r] * sc, dq_rsrc, off + r * rstep, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(q1[r] * sc, dq_rsrc, off + r * rstep + 64, 0, 0);
          } else {
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((q0[r] + q1[r]) * sc, dq_rsrc, off + r * rstep, 0, 0);
          }
        } else {  // plain stores through the same range-checked descriptor
          if (G::NBLK == 2) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q0[r] * sc), dq_rsrc, off + r * rstep, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q1[r] * sc), dq_rsrc, off + r * rstep + 64, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (q0[r] + q1[r]) * sc), dq_rsrc, off + r * rstep, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // dS consumed, the next q / dO tiles visible
  }

  if (kvalid) {
    float* dkp = p.dk + (int64_t)b * p.dks.sb + (int64_t)kj * p.dks.st + (int64_t)h * p.dks.sh + 4 * hf;
    float* dvp = p.dv + (int64_t)b * p.dvs.sb + (int64_t)kj * p.dvs.st + (int64_t)h * p.dvs.sh + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      st4(dkp + 8 * g, make_float4(dk0[4 * g] * AMK_LN2, dk0[4 * g + 1] * AMK_LN2, dk0[4 * g + 2] * AMK_LN2, dk0[4 * g + 3] * AMK_LN2));
      st4(dkp + 32 + 8 * g, make_float4(dk1[4 * g] * AMK_LN2, dk1[4 * g + 1] * AMK_LN2, dk1[4 * g + 2] * AMK_LN2, dk1[4 * g + 3] * AMK_LN2));
      st4(dvp + 8 * g, make_float4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]));
      st4(dvp + 32 + 8 * g, make_float4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]));
    }
  }
}

// dq[i] = sum over key blocks, in key-block order, of the partials the DQ = 1 kernel stored
__global__ __launch_bounds__(256) void attn_bwd_dq_reduce_kernel(const float* __restrict__ part, int nk, int64_t n4,
                                                                 float* __restrict__ dq) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 acc = ld4(part + 4 * i);
    for (int k = 1; k < nk; ++k) {
      const float4 x = ld4(part + ((int64_t)k * n4 + i) * 4);
      acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    st4(dq + 4 * i, acc);
  }
}

template <int NW, bool KEPT, int DQ, bool CAUSAL>
static bool launch_variant_c(BwdParams p, hipStream_t st) {
  using G = FusedGeom<NW>;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<NW, KEPT, DQ, CAUSAL>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  G::LDS_FLOATS * (int)sizeof(float)) == hipSuccess;
  if (!attr_ok) return false;
  p.nkblk = (p.J + G::KB - 1) / G::KB;
  const int64_t ndq = (int64_t)p.B * p.I * p.H * D;
  if (DQ == 0 && hipMemsetAsync(p.dq, 0, (size_t)ndq * sizeof(float), st) != hipSuccess) return false;
  const int64_t nk = (int64_t)p.B * p.H * p.nkblk;
  hipLaunchKernelGGL((attn_bwd_fused_kernel<NW, KEPT, DQ, CAUSAL>), dim3((unsigned)nk), dim3(G::NT), G::LDS_FLOATS * sizeof(float), st, p);
  if (DQ == 1 && p.nkblk > 1) {
    const int64_t n4 = ndq / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(attn_bwd_dq_reduce_kernel, dim3(grid), dim3(256), 0, st, p.dq_part, p.nkblk, n4, p.dq);
  }
  return true;
}

template <int NW, bool KEPT, int DQ>
static bool launch_variant(const BwdParams& p, hipStream_t st) {
  return p.causal_mask ? launch_variant_c<NW, KEPT, DQ, true>(p, st) : launch_variant_c<NW, KEPT, DQ, false>(p, st);
}

int fused_keys_per_wg(int J, int keys_per_wg) { return keys_per_wg ? keys_per_wg : (J >= 256 ? 256 : 128); }

// Launch the fused kernel.  dq must be the dense (B, I, H, 64) layout (one memset / one reduce pass);
// returns false (nothing launched) when the layout or the masks rule it out.
// keys_per_wg: 128 (4 waves, two workgroups per CU) or 256 (8 waves, one per CU; half the dq adds);
// 0 = pick (256 when the sequence has at least 256 keys).  p.dq_part != null: reproducible dq (DQ = 1).
bool launch_attn_bwd_fused(const BwdParams& p, int keys_per_wg, hipStream_t st) {
  if (!(p.dqs.sh == D && p.dqs.st == (int64_t)p.H * D && p.dqs.sb == (int64_t)p.I * p.H * D)) return false;
  keys_per_wg = fused_keys_per_wg(p.J, keys_per_wg);
  const bool det = p.dq_part != nullptr;
  if (keys_per_wg == 256) {
    if (det) return p.scores ? launch_variant<8, true, 1>(p, st) : launch_variant<8, false, 1>(p, st);
    return p.scores ? launch_variant<8, true, 0>(p, st) : launch_variant<8, false, 0>(p, st);
  }
  if (det) return p.scores ? launch_variant<4, true, 1>(p, st) : launch_variant<4, false, 1>(p, st);
  return p.scores ? launch_variant<4, true, 0>(p, st) : launch_variant<4, false, 0>(p, st);
}

}  // namespace amk_attn
