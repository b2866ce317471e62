// One-pass softmax-attention backward for head dims 32 and 128 (round 4): the structure of attn_bwd_fused.hip -- a
// workgroup owns 32*NW keys of one (batch, head), v of a wave's 32 keys in registers, the K block in LDS, dK^T / dV^T in
// accumulators, query tiles of 32, S read back from the forward's kept score tiles (or recomputed), dS through LDS once and
// dQ = dS K on v_mfma_f32_16x16x4_f32 added to dq by f32 atomics -- with the head dim as a template parameter.
//
//   DH = 32 : NW = 8 waves (256 keys per workgroup, two waves per SIMD) and query tiles of 64 (two 32-query sub-tiles per
//             barrier pair): the 64 x 32 dQ tile is eight 16 x 16 blocks, one per wave;
//   DH = 128: NW = 4 waves (128 keys per workgroup) at ONE wave per SIMD with the 512-register file: eight 32 x 32
//             accumulators for dK^T / dV^T alone are 128 registers; the 32 x 128 dQ tile is sixteen blocks, four per wave.
//
// Before this kernel those head dims ran the two recompute kernels of attn_generic.hip: 7 matrix products for the 4 the
// roofline credits.  Here 4 (kept scores; the unmasked forward of attn_generic.hip leaves them) or 5.  The reproducible-dq
// form (ordered partial sums) exists for head dim 64 only: under AMK_DETERMINISTIC these dims take the recompute pair.
#include "attn_common.h"

namespace amk_attn {

namespace {

// QS: 32-query sub-tiles per query tile (one barrier pair and one q / dO staging pass per TILE).  Head dim 32 uses 2: its
// sub-tile carries half the MFMA work of head dim 64 over the same softmax arithmetic, LDS round trip and barriers, its
// staging pass covers 64 rows anyway (512 threads x 16 B), and with 64 queries the dQ tile is eight 16 x 16 blocks -- one
// per wave, no split of the keys between waves.
template <int DH, int NW, int QS>
struct FGeom {
  static constexpr int GTQ = 32 * QS;            // queries per tile
  static constexpr int NT = 64 * NW;
  static constexpr int KB = 32 * NW;             // keys per workgroup
  static constexpr int HD = DH / 2;              // k-extent owned by one half-wave
  static constexpr int LS = DH + 4;              // LDS row stride of the q / dO / K images
  static constexpr int NTILE = DH / 32;          // 32-wide dim tiles of dK^T / dV^T
  static constexpr int F4R = DH / 4;             // threads per staged row (one float4 each)
  static constexpr int RP = NT / F4R;            // rows per staging pass of the workgroup
  static constexpr int QROWS = RP > GTQ ? RP : GTQ;   // rows of the q / dO images (threads past the tile stage zero rows)
  static constexpr int QNP = GTQ / RP > 0 ? GTQ / RP : 1;
  static constexpr int KNP = KB / RP;            // K-block staging passes
  static constexpr int DS_STRIDE = KB + 4;
  static constexpr int KPG = KB / 4;             // keys per k-group of the 16x16x4 product
  static constexpr int NCB = DH / 16;            // 16-wide column blocks of the dQ tile
  static constexpr int NQB = 2 * QS;             // 16-query row blocks of the dQ tile
  static constexpr int PAIRS = NW / NQB;         // waves per row block
  static constexpr int CBW = NCB / PAIRS > 0 ? NCB / PAIRS : 1;   // column blocks per wave
  static constexpr int KS = PAIRS / NCB > 0 ? PAIRS / NCB : 1;    // waves sharing a block (they split the keys)
  static constexpr int LDS_FLOATS = 2 * QROWS * LS + KB * LS + GTQ * DS_STRIDE + 4 * GTQ;
};

__device__ __forceinline__ f32x4 gmfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// rows [row0, row0 + nrows) x DH floats of one (batch, head) operand through a range-checked descriptor; thread ->
// (row = tid / F4R + RP * pass, 16 B at column 4 * (tid % F4R)); rows the tile does not have get an offset past the range
template <int DH, int NW, int QS>
struct FStager {
  using G = FGeom<DH, NW, QS>;
  __amdgpu_buffer_rsrc_t rsrc;
  int64_t st;
  int nrows_total;
  __device__ __forceinline__ void init(const float* base, int64_t row_stride, int nrows) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((int64_t)(nrows - 1) * row_stride + DH) * 4), 0x00020000);
    st = row_stride;
    nrows_total = nrows;
  }
  // the 16-byte piece of pass ps of a tile starting at row0 that has `have` rows (<= RP * passes)
  __device__ __forceinline__ float4 load(int row0, int have, int ps, int tid) const {
    const int r = tid / G::F4R + G::RP * ps;
    const unsigned off = (r < have) ? (unsigned)(((int64_t)(row0 + r) * st + (tid % G::F4R) * 4) * 4) : 0x80000000u;
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
  }
};

template <int DH, int NW, int QS, bool KEPT, bool CAUSAL>
__global__ __launch_bounds__(64 * NW, (DH >= 128 ? 1 : 2)) void attn_bwd_fused_gen_kernel(BwdParams p) {
  using G = FGeom<DH, NW, QS>;
  constexpr int KB = G::KB, HD = G::HD, LS = G::LS, NTILE = G::NTILE, DS_STRIDE = G::DS_STRIDE, KPG = G::KPG;
  constexpr int TQ = G::GTQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                           // [QROWS][LS]  (q * scale * log2 e)
  float* Gs = Qs + G::QROWS * LS;             // [QROWS][LS]  dO
  float* Kc = Gs + G::QROWS * LS;             // [KB][LS]     this workgroup's K rows
  float* dSl = Kc + KB * LS;                  // [TQ][DS_STRIDE]
  float* Ms = dSl + TQ * DS_STRIDE;
  float* Ls = Ms + TQ;
  float* Ds = Ls + TQ;                        // MINUS delta
  float* MLs = Ds + TQ;                       // m + log2 l

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 31, hf = lane >> 5;

  const int wg = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int kb = __builtin_amdgcn_readfirstlane(wg % p.nkblk);
  const int bh = __builtin_amdgcn_readfirstlane(wg / p.nkblk);
  const int h = __builtin_amdgcn_readfirstlane(bh % p.H), b = __builtin_amdgcn_readfirstlane(bh / p.H);
  const int kj = kb * KB + wave * 32 + ln;  // this lane's key row
  const bool kvalid = kj < p.J;

  const float* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;

  float vreg[HD];
  {
    const float* vp = vbase + (int64_t)kj * p.vs.st + HD * hf;
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
      const float4 c = kvalid ? ld4(vp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      vreg[4 * s4 + 0] = c.x; vreg[4 * s4 + 1] = c.y; vreg[4 * s4 + 2] = c.z; vreg[4 * s4 + 3] = c.w;
    }
  }
  float kfill = 0.f;  // 0 keep, -1e9*log2e masked key, -inf beyond the sequence
  if (!kvalid) kfill = -INFINITY;
  else if (p.key_mask && p.key_mask[(int64_t)b * p.J + kj] == 0) kfill = AMK_FILL_MASKED;
  const bool plain = !CAUSAL && __all(kfill == 0.f);  // wave-uniform
  const uint8_t* cm_col = CAUSAL ? p.causal_mask + min(kj, p.J - 1) : nullptr;
  unsigned cm_raw[QS][16];
  auto load_cmask = [&](int i0) {
#pragma unroll
    for (int u = 0; u < QS; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) cm_raw[u][r] = cm_col[(int64_t)min(i0 + 32 * u + acc_row(r, hf), p.I - 1) * p.J];
  };

  // K rows of the whole workgroup -> LDS once
  {
    FStager<DH, NW, QS> kl;
    kl.init(kbase, p.ks.st, p.J);
    const int have = min(KB, p.J - kb * KB);
    constexpr int HALFP = G::KNP >= 2 ? G::KNP / 2 : 1;
#pragma unroll
    for (int half = 0; half < G::KNP / HALFP; ++half) {
      float4 t[HALFP];
#pragma unroll
      for (int ps = 0; ps < HALFP; ++ps) t[ps] = kl.load(kb * KB, have, half * HALFP + ps, tid);
#pragma unroll
      for (int ps = 0; ps < HALFP; ++ps)
        st4(&Kc[(tid / G::F4R + G::RP * (half * HALFP + ps)) * LS + (tid % G::F4R) * 4], t[ps]);
    }
  }

  const float* qbase = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* gbase = p.d_o + (int64_t)b * p.dos.sb + (int64_t)h * p.dos.sh;
  const float* stbase = p.stats + ((int64_t)b * p.H + h) * p.I * 2;
  const float* dlbase = p.delta + ((int64_t)b * p.H + h) * p.I;
  float* dqbase = p.dq + (int64_t)b * p.dqs.sb + (int64_t)h * p.dqs.sh;
  const __amdgpu_buffer_rsrc_t dq_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)dqbase, 0, (int)(((int64_t)(p.I - 1) * p.dqs.st + DH) * 4), 0x00020000);

  __amdgpu_buffer_rsrc_t sc_rsrc;
  int sc_voff = 0;
  if (KEPT) {
    const ScoreTiles stl(p.I, p.J);
    const int kb32 = kb * NW + wave;
    const float* tiles = p.scores + ((int64_t)bh * stl.nkb + min(kb32, stl.nkb - 1)) * stl.nqt * 1024;
    sc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tiles, 0, kb32 < stl.nkb ? stl.nqt * 4096 : 0, 0x00020000);
    sc_voff = ln * 128 + hf * 16;
  }
  float4 sk[QS][4];
  auto load_scores = [&](int qt) {   // qt: query TILE; its sub-tiles are consecutive 4-KiB score tiles
#pragma unroll
    for (int u = 0; u < QS; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        sk[u][g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sc_rsrc, sc_voff + (qt * QS + u) * 4096 + 32 * g, 0, 2));
  };

  float4 qst[G::QNP], gst[G::QNP];
  FStager<DH, NW, QS> qload, gload;
  qload.init(qbase, p.qs.st, p.I);
  gload.init(gbase, p.dos.st, p.I);
  float2 ml_raw = make_float2(0.f, 1.f);
  float dl_raw = 0.f;
  bool row_ok = false;
  auto prefetch = [&](int i0) {
    const int have = min(TQ, p.I - i0);   // (<= 0 past the sequence: every piece reads zeros)
#pragma unroll
    for (int ps = 0; ps < G::QNP; ++ps) {
      qst[ps] = qload.load(i0, have, ps, tid);
      gst[ps] = gload.load(i0, have, ps, tid);
    }
    const int i = i0 + (tid & (TQ - 1));
    const int ic = min(max(i, 0), p.I - 1);
    ml_raw = *reinterpret_cast<const float2*>(stbase + 2 * ic);
    dl_raw = dlbase[ic];
    row_ok = i < p.I;
  };
  auto commit = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    const float sc = p.scale * AMK_LOG2E;
#pragma unroll
    for (int ps = 0; ps < G::QNP; ++ps) {
      const int r = tid / G::F4R + G::RP * ps, c = (tid % G::F4R) * 4;
      st4(&Qs[r * LS + c], make_float4(qst[ps].x * sc, qst[ps].y * sc, qst[ps].z * sc, qst[ps].w * sc));
      st4(&Gs[r * LS + c], gst[ps]);
    }
    if (tid < TQ) {
      Ms[tid] = row_ok ? ml_raw.x : INFINITY;
      Ls[tid] = row_ok ? 1.f / ml_raw.y : 0.f;
      Ds[tid] = row_ok ? -dl_raw : 0.f;
      MLs[tid] = row_ok ? ml_raw.x + __builtin_amdgcn_logf(ml_raw.y) : INFINITY;
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 dk[NTILE], dv[NTILE];
#pragma unroll
  for (int n = 0; n < NTILE; ++n) { dk[n] = zero16(); dv[n] = zero16(); }

  // dQ product geometry: 16x16x4 MFMA, lane = (column c = l & 15, k-group kg = l >> 4)
  const int c16 = lane & 15, kg = lane >> 4;
  const int qhalf = wave % G::NQB;                           // which 16 of the tile's queries
  const int pr = wave / G::NQB;                              // index among the waves of this row block
  const int colblk0 = G::KS > 1 ? pr % G::NCB : pr * G::CBW; // first 16-wide column block of this wave
  const int kpart = G::KS > 1 ? pr / G::NCB : 0;             // which part of the keys (waves sharing a block)
  constexpr int S4N = KPG / 4 / G::KS;                       // k-steps of 4 per wave
  const float* ds_row = &dSl[(16 * qhalf + c16) * DS_STRIDE + KPG * kg + 4 * S4N * kpart];
  const float* kc_col = &Kc[(KPG * kg + 4 * S4N * kpart) * LS + 16 * colblk0 + c16];

  const int ntile = (p.I + TQ - 1) / TQ;
  const int rot = (int)(((unsigned)bh * 5u) % (unsigned)ntile);
  auto tile_of = [&](int t) { const int x = t + rot; return x >= ntile ? x - ntile : x; };
  prefetch(rot * TQ);
  if (KEPT) load_scores(rot);
  if (CAUSAL) load_cmask(rot * TQ);
  __syncthreads();  // the K block is in LDS
  commit();
  __syncthreads();
  prefetch(tile_of(min(1, ntile - 1)) * TQ);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int t = 0; t < ntile; ++t) {
    const int i0 = tile_of(t) * TQ;

    // ---- per 32-query sub-tile: S and dP for its queries x this wave's 32 keys (dP's chain starts from -delta), P, dS,
    //      dS -> LDS, dV^T / dK^T accumulation
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      f32x16 s = zero16(), dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 d4 = ld4(&Ds[32 * u + 8 * g + 4 * hf]);
        dp[4 * g + 0] = d4.x; dp[4 * g + 1] = d4.y; dp[4 * g + 2] = d4.z; dp[4 * g + 3] = d4.w;
      }
      {
        const float* qr = &Qs[(32 * u + ln) * LS + HD * hf];
        const float* gr = &Gs[(32 * u + ln) * LS + HD * hf];
        const float* kr = &Kc[(32 * wave + ln) * LS + HD * hf];
#pragma unroll
        for (int s4 = 0; s4 < HD / 4; ++s4) {
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
      if (KEPT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          s[4 * g + 0] = sk[u][g].x; s[4 * g + 1] = sk[u][g].y; s[4 * g + 2] = sk[u][g].z; s[4 * g + 3] = sk[u][g].w;
        }
      }
      // P and dS (register r of this lane is query 32 u + acc_row(r, hf))
      if (plain) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 ml4 = ld4(&MLs[32 * u + 8 * g + 4 * hf]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float prb = __builtin_amdgcn_exp2f(s[r] - f4(ml4, e));
            s[r] = prb;
            dp[r] = prb * dp[r];
          }
        }
      } else {
        unsigned cbits = 0xffffu;
        if (CAUSAL) {
          cbits = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) cbits |= (cm_raw[u][r] == 0 ? 1u : 0u) << r;
        }
        const float fillv = kfill != 0.f ? kfill : AMK_FILL_MASKED;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 m4 = ld4(&Ms[32 * u + 8 * g + 4 * hf]);
          const float4 l4 = ld4(&Ls[32 * u + 8 * g + 4 * hf]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const bool filled = kfill != 0.f || !((cbits >> r) & 1u);
            const float tt = filled ? fillv : s[r];
            const float prb = __builtin_amdgcn_exp2f(tt - f4(m4, e)) * f4(l4, e);
            s[r] = prb;
            dp[r] = filled ? 0.f : prb * dp[r];
          }
        }
      }
      // dS -> LDS as [query][key]
#pragma unroll
      for (int r = 0; r < 16; ++r) dSl[(32 * u + acc_row(r, hf)) * DS_STRIDE + 32 * wave + ln] = dp[r];
      // the next tile's scores / mask bytes: requested HERE, ahead of the last sub-tile's dV / dK products (consumed after
      // the next tile's dP product: three MFMA phases away; the last iteration re-reads its own tile, unused)
      if (u == QS - 1) {
        if (KEPT) {
          load_scores(tile_of(min(t + 1, ntile - 1)));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (CAUSAL) load_cmask(tile_of(min(t + 1, ntile - 1)) * TQ);
      }
      // dV^T += dO^T P ; dK^T += (q*scale*log2e)^T dS
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* gc = &Gs[(32 * u + acc_row(r, hf)) * LS + ln];
        const float* qc = &Qs[(32 * u + acc_row(r, hf)) * LS + ln];
#pragma unroll
        for (int n = 0; n < NTILE; ++n) {
          dv[n] = mfma32(gc[32 * n], s[r], dv[n]);
          dk[n] = mfma32(qc[32 * n], dp[r], dk[n]);
        }
      }
    }
    __syncthreads();  // every wave's dS columns are in LDS; the q / dO / stats tiles are dead
    commit();
    prefetch(tile_of(min(t + 2, ntile - 1)) * TQ);

    // ---- dQ: this wave's 16 queries x CBW column blocks over its share of the workgroup's keys
    f32x4 qa[G::CBW];
#pragma unroll
    for (int cbk = 0; cbk < G::CBW; ++cbk) qa[cbk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < S4N; ++s4) {
      const float4 a = ld4(ds_row + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* kc = kc_col + (4 * s4 + e) * LS;
#pragma unroll
        for (int cbk = 0; cbk < G::CBW; ++cbk) qa[cbk] = gmfma16(f4(a, e), kc[16 * cbk], qa[cbk]);
      }
    }
    {
      const float sc = p.scale;
      const int qi0 = i0 + 16 * qhalf + 4 * kg;
      const int off = (int)(((int64_t)qi0 * p.dqs.st + 16 * colblk0 + c16) * 4);
      const int rstep = (int)(p.dqs.st * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cbk = 0; cbk < G::CBW; ++cbk)
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(qa[cbk][r] * sc, dq_rsrc, off + r * rstep + 64 * cbk, 0, 0);
    }
    __syncthreads();  // dS consumed, the next q / dO tiles visible
  }

  if (kvalid) {
    float* dkp = p.dk + (int64_t)b * p.dks.sb + (int64_t)kj * p.dks.st + (int64_t)h * p.dks.sh + 4 * hf;
    float* dvp = p.dv + (int64_t)b * p.dvs.sb + (int64_t)kj * p.dvs.st + (int64_t)h * p.dvs.sh + 4 * hf;
#pragma unroll
    for (int n = 0; n < NTILE; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        st4(dkp + 32 * n + 8 * g, make_float4(dk[n][4 * g] * AMK_LN2, dk[n][4 * g + 1] * AMK_LN2, dk[n][4 * g + 2] * AMK_LN2, dk[n][4 * g + 3] * AMK_LN2));
        st4(dvp + 32 * n + 8 * g, make_float4(dv[n][4 * g], dv[n][4 * g + 1], dv[n][4 * g + 2], dv[n][4 * g + 3]));
      }
  }
}

template <int DH, int NW, int QS, bool KEPT, bool CAUSAL>
bool launch_one(BwdParams p, hipStream_t st) {
  using G = FGeom<DH, NW, QS>;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_gen_kernel<DH, NW, QS, KEPT, CAUSAL>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  G::LDS_FLOATS * (int)sizeof(float)) == hipSuccess;
  if (!attr_ok) return false;
  p.nkblk = (p.J + G::KB - 1) / G::KB;
  const int64_t ndq = (int64_t)p.B * p.I * p.H * DH;
  if (hipMemsetAsync(p.dq, 0, (size_t)ndq * sizeof(float), st) != hipSuccess) return false;
  const int64_t nk = (int64_t)p.B * p.H * p.nkblk;
  hipLaunchKernelGGL((attn_bwd_fused_gen_kernel<DH, NW, QS, KEPT, CAUSAL>), dim3((unsigned)nk), dim3(G::NT), G::LDS_FLOATS * sizeof(float), st, p);
  return true;
}

template <int DH, int NW, int QS>
bool launch_dh(const BwdParams& p, hipStream_t st) {
  if (p.scores) return p.causal_mask ? launch_one<DH, NW, QS, true, true>(p, st) : launch_one<DH, NW, QS, true, false>(p, st);
  return p.causal_mask ? launch_one<DH, NW, QS, false, true>(p, st) : launch_one<DH, NW, QS, false, false>(p, st);
}

}  // namespace

// The one-pass backward for head dims 32 / 128 (dq by f32 atomics into the dense (B, I, H, Dh) layout, zeroed here).
// false = not applicable (layout, or the reproducible-dq form was asked for), nothing launched.
bool launch_attn_bwd_fused_gen(const BwdParams& p, int Dh, hipStream_t st) {
  if (p.dq_part) return false;
  if (!(p.dqs.sh == Dh && p.dqs.st == (int64_t)p.H * Dh && p.dqs.sb == (int64_t)p.I * p.H * Dh)) return false;
  if (Dh == 32) return launch_dh<32, 8, 2>(p, st);
  if (Dh == 128) return launch_dh<128, 4, 1>(p, st);
  return false;
}

}  // namespace amk_attn
