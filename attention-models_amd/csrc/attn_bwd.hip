// Fused softmax-attention backward for gfx950 (exact-f32 MFMA, scores recomputed).
//
// Autograd of models/softmax_attention.py:62-76 of the reference, where PyTorch keeps the
// (B,h,I,J) score and probability tensors for the backward.  Here P is recomputed from
// q, k and the saved row statistics {m, l}; three launches, no atomics, bitwise
// reproducible:
//
//   attn_bwd_delta : delta[b,h,i] = sum_d dO*O                     (HBM-bound, tiny)
//   attn_bwd_dq    : QUERY ON THE LANE (same skeleton as the forward)
//        S^T  = K Q^T, dP^T = V dO^T      (A = K / V rows from LDS, B = q / dO registers)
//        dS^T = P^T o (dP^T - delta)      (per-lane scalars m, 1/l, delta)
//        dQ^T += K^T dS^T                 (A = K columns from LDS, B = the dS^T accumulator)
//   attn_bwd_dkdv  : KEY ON THE LANE (a wave owns 32 keys, k / v live in registers)
//        S  = Q K^T,  dP = dO V^T         (A = q / dO rows from LDS, B = k / v registers)
//        dV^T += dO^T P,  dK^T += Q^T dS  (A = dO / q columns from LDS, B = accumulators)
//
// Gradients do not flow through positions the forward filled with -1e9 (masked_fill).
#include "attn_common.h"

namespace amk_attn {

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(BwdParams p) {
  // 16 lanes per (b,h,i) row: each adds 4 of the 64 products, then a 16-lane butterfly.
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t nrow = (int64_t)p.B * p.H * p.I;
  const int c = (threadIdx.x & 15) * 4;
  float acc = 0.f;
  if (row < nrow) {
    const int i = (int)(row % p.I);
    const int64_t bh = row / p.I;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const float4 a = ld4(p.o + (int64_t)b * p.os.sb + (int64_t)i * p.os.st + (int64_t)h * p.os.sh + c);
    const float4 g = ld4(p.d_o + (int64_t)b * p.dos.sb + (int64_t)i * p.dos.st + (int64_t)h * p.dos.sh + c);
    acc = a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
  }
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 4, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 1, 64);
  if (row < nrow && (threadIdx.x & 15) == 0) p.delta[row] = acc;
}

// ---------------------------------------------------------------------------------------
template <bool CAUSAL>
__global__ __launch_bounds__(WG, 2) void attn_bwd_dq_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE * LDS_STRIDE + TILE];
  float* Ks = smem;
  float* Vs = smem + TILE * LDS_STRIDE;
  float* Kfill = smem + 2 * TILE * LDS_STRIDE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wg % p.nqblk;
  const int bh = wg / p.nqblk;
  const int h = bh % p.H, b = bh / p.H;
  const int qi = qb * BLK + wave * 32 + ln;
  const bool qvalid = qi < p.I;

  // B operands held for the whole kernel: (q*scale*log2 e) and dO of this lane's query.
  const float qscale = p.scale * AMK_LOG2E;
  float qreg[32], greg[32];
  float m_q = INFINITY, linv_q = 0.f, delta_q = 0.f;
  {
    const float* qp = p.q + (int64_t)b * p.qs.sb + (int64_t)qi * p.qs.st + (int64_t)h * p.qs.sh + 32 * hf;
    const float* gp = p.d_o + (int64_t)b * p.dos.sb + (int64_t)qi * p.dos.st + (int64_t)h * p.dos.sh + 32 * hf;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const float4 t = qvalid ? ld4(qp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 g = qvalid ? ld4(gp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      qreg[4 * s4 + 0] = t.x * qscale; qreg[4 * s4 + 1] = t.y * qscale;
      qreg[4 * s4 + 2] = t.z * qscale; qreg[4 * s4 + 3] = t.w * qscale;
      greg[4 * s4 + 0] = g.x; greg[4 * s4 + 1] = g.y; greg[4 * s4 + 2] = g.z; greg[4 * s4 + 3] = g.w;
    }
    if (qvalid) {
      const int64_t row = ((int64_t)b * p.H + h) * p.I + qi;
      m_q = p.stats[2 * row];
      linv_q = 1.f / p.stats[2 * row + 1];
      delta_q = p.delta[row];
    }
  }

  const float* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;
  const uint8_t* kmask = p.key_mask ? p.key_mask + (int64_t)b * p.J : nullptr;
  const uint8_t* cmrow = CAUSAL ? p.causal_mask + (int64_t)qi * p.J : nullptr;

  const int srow = tid >> 4, scol = (tid & 15) * 4;
  float4 kst[4], vst[4];
  float fillst = 0.f;
  RowStager kload, vload;
  kload.init(kbase, p.ks.st, p.J, tid);
  vload.init(vbase, p.vs.st, p.J, tid);
  auto prefetch = [&](int j0) {
    kload.load(kst);
    vload.load(vst);
    if (tid < TILE) {
      const int j = j0 + tid;
      float f = 0.f;
      if (j >= p.J) f = -INFINITY;
      else if (kmask && kmask[j] == 0) f = AMK_FILL_MASKED;
      fillst = f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int r = srow + 16 * ps;
      st4(&Ks[r * LDS_STRIDE + scol], kst[ps]);
      st4(&Vs[r * LDS_STRIDE + scol], vst[ps]);
    }
    if (tid < TILE) Kfill[tid] = fillst;
  };

  f32x16 dq0 = zero16(), dq1 = zero16();
  const int ntile = (p.J + TILE - 1) / TILE;
  prefetch(0);
  for (int t = 0; t < ntile; ++t) {
    const int j0 = t * TILE;
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch(j0 + TILE);

    unsigned cbits0 = 0, cbits1 = 0;
    if (CAUSAL) {
      if (qvalid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ja = j0 + acc_row(r, hf), jb = ja + 32;
          if (ja < p.J && cmrow[ja]) cbits0 |= 1u << r;
          if (jb < p.J && cmrow[jb]) cbits1 |= 1u << r;
        }
      }
    }
    const bool plain = !CAUSAL && kmask == nullptr && (j0 + TILE <= p.J);  // wave-uniform

#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // S^T and dP^T for 32 keys (2 x 32 MFMAs)
      f32x16 s = zero16(), dp = zero16();
      const float* kr = &Ks[(32 * u + ln) * LDS_STRIDE + 32 * hf];
      const float* vr = &Vs[(32 * u + ln) * LDS_STRIDE + 32 * hf];
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        const float4 a = ld4(kr + 4 * s4);
        const float4 c = ld4(vr + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = mfma32(f4(a, e), qreg[4 * s4 + e], s);
          dp = mfma32(f4(c, e), greg[4 * s4 + e], dp);
        }
      }
      const unsigned cb = u ? cbits1 : cbits0;
      if (plain) {  // no fills in this tile: dS^T = P^T o (dP^T - delta)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = __builtin_amdgcn_exp2f(s[r] - m_q) * linv_q;
          s[r] = pr * (dp[r] - delta_q);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 f = ld4(&Kfill[32 * u + 8 * g + 4 * hf]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float fe = f4(f, e);
            bool filled = fe != 0.f;
            float tt = filled ? fe : s[r];
            if (CAUSAL) {
              if ((cb >> r) & 1u) { tt = AMK_FILL_MASKED; filled = true; }
            }
            const float pr = __builtin_amdgcn_exp2f(tt - m_q) * linv_q;
            s[r] = filled ? 0.f : pr * (dp[r] - delta_q);  // dS^T (no gradient through fills)
          }
        }
      }
      // dQ^T += K^T dS^T (32 MFMAs): A = K[key(r,half)][dim], B = dS^T register r
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* kc = &Ks[(32 * u + acc_row(r, hf)) * LDS_STRIDE + ln];
        dq0 = mfma32(kc[0], s[r], dq0);
        dq1 = mfma32(kc[32], s[r], dq1);
      }
    }
  }

  if (qvalid) {
    float* dp_ = p.dq + (int64_t)b * p.dqs.sb + (int64_t)qi * p.dqs.st + (int64_t)h * p.dqs.sh + 4 * hf;
    const float sc = p.scale;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      st4(dp_ + 8 * g, make_float4(dq0[4 * g] * sc, dq0[4 * g + 1] * sc, dq0[4 * g + 2] * sc, dq0[4 * g + 3] * sc));
      st4(dp_ + 32 + 8 * g, make_float4(dq1[4 * g] * sc, dq1[4 * g + 1] * sc, dq1[4 * g + 2] * sc, dq1[4 * g + 3] * sc));
    }
  }
}

// ---------------------------------------------------------------------------------------
template <bool CAUSAL>
__global__ __launch_bounds__(WG, CAUSAL ? 1 : 2) void attn_bwd_dkdv_kernel(BwdParams p) {
  // LDS: (q*scale) tile, dO tile (64 query rows each), then m, 1/l, delta of those rows.
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE * LDS_STRIDE + 3 * TILE];
  float* Qs = smem;
  float* Gs = smem + TILE * LDS_STRIDE;
  float* Ms = smem + 2 * TILE * LDS_STRIDE;
  float* Ls = Ms + TILE;
  float* Ds = Ls + TILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = wg % p.nkblk;
  const int bh = wg / p.nkblk;
  const int h = bh % p.H, b = bh / p.H;
  const int kj = kb * BLK + wave * 32 + ln;  // this lane's key row
  const bool kvalid = kj < p.J;

  // B operands held for the whole kernel: k and v of this lane's key.
  float kreg[32], vreg[32];
  {
    const float* kp = p.k + (int64_t)b * p.ks.sb + (int64_t)kj * p.ks.st + (int64_t)h * p.ks.sh + 32 * hf;
    const float* vp = p.v + (int64_t)b * p.vs.sb + (int64_t)kj * p.vs.st + (int64_t)h * p.vs.sh + 32 * hf;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const float4 a = kvalid ? ld4(kp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 c = kvalid ? ld4(vp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      kreg[4 * s4 + 0] = a.x; kreg[4 * s4 + 1] = a.y; kreg[4 * s4 + 2] = a.z; kreg[4 * s4 + 3] = a.w;
      vreg[4 * s4 + 0] = c.x; vreg[4 * s4 + 1] = c.y; vreg[4 * s4 + 2] = c.z; vreg[4 * s4 + 3] = c.w;
    }
  }
  // fill of this lane's key: 0 keep, -1e9*log2e masked key, -inf beyond the sequence
  float kfill = 0.f;
  if (!kvalid) kfill = -INFINITY;
  else if (p.key_mask && p.key_mask[(int64_t)b * p.J + kj] == 0) kfill = AMK_FILL_MASKED;
  const bool kfill_wave_plain = __all(kfill == 0.f);

  const float* qbase = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* gbase = p.d_o + (int64_t)b * p.dos.sb + (int64_t)h * p.dos.sh;
  const float* stbase = p.stats + ((int64_t)b * p.H + h) * p.I * 2;
  const float* dlbase = p.delta + ((int64_t)b * p.H + h) * p.I;

  const int srow = tid >> 4, scol = (tid & 15) * 4;
  float4 qst[4], gst[4];
  float mst = 0.f, lst = 0.f, dst = 0.f;
  RowStager qload, gload;
  qload.init(qbase, p.qs.st, p.I, tid);
  gload.init(gbase, p.dos.st, p.I, tid);
  auto prefetch = [&](int i0) {
    qload.load(qst);
    gload.load(gst);
    if (tid < TILE) {
      const int i = i0 + tid;
      if (i < p.I) {
        mst = stbase[2 * i];
        lst = 1.f / stbase[2 * i + 1];
        dst = dlbase[i];
      } else {  // rows beyond the sequence: P = exp2(x - inf) * 0 = 0
        mst = INFINITY; lst = 0.f; dst = 0.f;
      }
    }
  };
  auto commit = [&]() {
    const float sc = p.scale * AMK_LOG2E;  // S comes out in the log2 domain; dK is scaled back by ln 2
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int r = srow + 16 * ps;
      st4(&Qs[r * LDS_STRIDE + scol], make_float4(qst[ps].x * sc, qst[ps].y * sc, qst[ps].z * sc, qst[ps].w * sc));
      st4(&Gs[r * LDS_STRIDE + scol], gst[ps]);
    }
    if (tid < TILE) { Ms[tid] = mst; Ls[tid] = lst; Ds[tid] = dst; }
  };

  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const uint8_t* cmcol = CAUSAL ? p.causal_mask + kj : nullptr;

  const int ntile = (p.I + TILE - 1) / TILE;
  prefetch(0);
  for (int t = 0; t < ntile; ++t) {
    const int i0 = t * TILE;
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch(i0 + TILE);

    unsigned cbits0 = 0, cbits1 = 0;
    if (CAUSAL) {
      if (kvalid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ia = i0 + acc_row(r, hf), ib = ia + 32;
          if (ia < p.I && cmcol[(int64_t)ia * p.J]) cbits0 |= 1u << r;
          if (ib < p.I && cmcol[(int64_t)ib * p.J]) cbits1 |= 1u << r;
        }
      }
    }

#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // S and dP for 32 queries x this wave's 32 keys (2 x 32 MFMAs)
      f32x16 s = zero16(), dp = zero16();
      const float* qr = &Qs[(32 * u + ln) * LDS_STRIDE + 32 * hf];
      const float* gr = &Gs[(32 * u + ln) * LDS_STRIDE + 32 * hf];
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        const float4 a = ld4(qr + 4 * s4);
        const float4 c = ld4(gr + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = mfma32(f4(a, e), kreg[4 * s4 + e], s);
          dp = mfma32(f4(c, e), vreg[4 * s4 + e], dp);
        }
      }
      const unsigned cb = u ? cbits1 : cbits0;
      const bool plain = !CAUSAL && kfill_wave_plain;  // wave-uniform: none of this wave's keys is filled
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 m4 = ld4(&Ms[32 * u + 8 * g + 4 * hf]);
        const float4 l4 = ld4(&Ls[32 * u + 8 * g + 4 * hf]);
        const float4 d4 = ld4(&Ds[32 * u + 8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          if (plain) {
            const float pr = __builtin_amdgcn_exp2f(s[r] - f4(m4, e)) * f4(l4, e);
            s[r] = pr;                                   // P
            dp[r] = pr * (dp[r] - f4(d4, e));            // dS
          } else {
            bool filled = kfill != 0.f;
            float tt = filled ? kfill : s[r];
            if (CAUSAL) {
              if ((cb >> r) & 1u) { tt = AMK_FILL_MASKED; filled = true; }
            }
            const float pr = __builtin_amdgcn_exp2f(tt - f4(m4, e)) * f4(l4, e);
            s[r] = pr;                                             // P
            dp[r] = filled ? 0.f : pr * (dp[r] - f4(d4, e));       // dS
          }
        }
      }
      // dV^T += dO^T P ; dK^T += (q*scale)^T dS   (2 x 32 MFMAs)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* gc = &Gs[(32 * u + acc_row(r, hf)) * LDS_STRIDE + ln];
        const float* qc = &Qs[(32 * u + acc_row(r, hf)) * LDS_STRIDE + ln];
        dv0 = mfma32(gc[0], s[r], dv0);
        dv1 = mfma32(gc[32], s[r], dv1);
        dk0 = mfma32(qc[0], dp[r], dk0);
        dk1 = mfma32(qc[32], dp[r], dk1);
      }
    }
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

}  // namespace amk_attn

using namespace amk_attn;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool strides_ok(const Strides& s) { return (s.sb % 4 == 0) && (s.st % 4 == 0) && (s.sh % 4 == 0); }

static int attn_bwd_impl(const float* scores, const float* q, const float* k, const float* v, const float* o,
                            const float* stats, const float* d_o,
                            float* dq, float* dk, float* dv, float* delta_ws,
                            const uint8_t* key_mask, const uint8_t* causal_mask,
                            int B, int H, int I, int J, int Dh,
                            int64_t q_sb, int64_t q_st, int64_t q_sh,
                            int64_t k_sb, int64_t k_st, int64_t k_sh,
                            int64_t v_sb, int64_t v_st, int64_t v_sh,
                            int64_t o_sb, int64_t o_st, int64_t o_sh,
                            int64_t do_sb, int64_t do_st, int64_t do_sh,
                            int64_t dq_sb, int64_t dq_st, int64_t dq_sh,
                            int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                            int64_t dv_sb, int64_t dv_st, int64_t dv_sh,
                            float scale, int stages, void* stream) {
  AMK_CHECK_ARG(q && k && v && o && stats && d_o && dq && dk && dv && delta_ws, "amk_attn_bwd: null tensor pointer");
  AMK_CHECK_ARG(B > 0 && H > 0 && I > 0 && J > 0, "amk_attn_bwd: non-positive size B=%d H=%d I=%d J=%d", B, H, I, J);
  AMK_CHECK_SUPPORTED(Dh == D || attn_gen_supported(Dh), "amk_attn_bwd: head dim %d not supported (32, 64, 128)", Dh);
  AMK_CHECK_SUPPORTED(Dh == D || !scores || !(stages & AMK_ATTN_BWD_DQ_REPRO),
                      "amk_attn_bwd_kept: kept scores with the reproducible dq exist for head dim %d only", D);
  BwdParams p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.stats = stats; p.d_o = d_o;
  p.dq = dq; p.dk = dk; p.dv = dv; p.delta = delta_ws;
  p.key_mask = key_mask; p.causal_mask = causal_mask;
  p.B = B; p.H = H; p.I = I; p.J = J;
  p.qs = {q_sb, q_st, q_sh}; p.ks = {k_sb, k_st, k_sh}; p.vs = {v_sb, v_st, v_sh}; p.os = {o_sb, o_st, o_sh};
  p.dos = {do_sb, do_st, do_sh}; p.dqs = {dq_sb, dq_st, dq_sh}; p.dks = {dk_sb, dk_st, dk_sh}; p.dvs = {dv_sb, dv_st, dv_sh};
  p.scale = scale;
  p.pinf = INFINITY;
  p.nqblk = (I + BLK - 1) / BLK;
  p.nkblk = (J + BLK - 1) / BLK;
  p.scores = scores;
  // reproducible dq from the one-pass kernel: the partials live behind the deltas in the workspace
  p.dq_part = (stages & AMK_ATTN_BWD_DQ_REPRO) ? delta_ws + (((int64_t)B * H * I + 3) & ~(int64_t)3) : nullptr;
  AMK_CHECK_ARG(!scores || aligned16(scores), "amk_attn_bwd_kept: the scores buffer must be 16-byte aligned");
  AMK_CHECK_ARG(!scores || (stages & AMK_ATTN_BWD_FUSED), "amk_attn_bwd_kept: kept scores are read by the fused pass only");
  AMK_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(d_o) && aligned16(dq) &&
                    aligned16(dk) && aligned16(dv) && strides_ok(p.qs) && strides_ok(p.ks) && strides_ok(p.vs) &&
                    strides_ok(p.os) && strides_ok(p.dos) && strides_ok(p.dqs) && strides_ok(p.dks) && strides_ok(p.dvs),
                "amk_attn_bwd: pointers must be 16-byte aligned and strides multiples of 4 elements");
  const int64_t nrow = (int64_t)B * H * I;
  const int64_t nq = (int64_t)B * H * p.nqblk, nk = (int64_t)B * H * p.nkblk;
  AMK_CHECK_SUPPORTED(nq < (1ll << 31) && nk < (1ll << 31) && (nrow + 15) / 16 < (1ll << 31), "amk_attn_bwd: grid too large");
  AMK_CHECK_SUPPORTED(((int64_t)J + TILE) * k_st * 4 < (1ll << 31) && ((int64_t)J + TILE) * v_st * 4 < (1ll << 31) &&
                          ((int64_t)I + TILE) * q_st * 4 < (1ll << 31) && ((int64_t)I + TILE) * do_st * 4 < (1ll << 31),
                      "amk_attn_bwd: one (batch, head) slab must span < 2 GiB");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (Dh != D) {
    // head dims 32 / 128: the one-pass kernel of attn_bwd_fused_gen.hip (dq by atomics) when FUSED is asked for and the
    // layout allows it, else the two recompute kernels of attn_generic.hip (always for the reproducible dq)
    if (stages & AMK_ATTN_BWD_FUSED) {
      if (stages & AMK_ATTN_BWD_DELTA) launch_attn_bwd_gen(p, Dh, AMK_ATTN_BWD_DELTA, st);
      if (!launch_attn_bwd_fused_gen(p, Dh, st)) {
        AMK_CHECK_SUPPORTED(!scores, "amk_attn_bwd_kept: the one-pass kernel could not run (dq layout) and the recompute kernels do not read kept scores");
        launch_attn_bwd_gen(p, Dh, AMK_ATTN_BWD_DKDV | AMK_ATTN_BWD_DQ, st);
      }
    } else {
      launch_attn_bwd_gen(p, Dh, stages, st);
    }
    AMK_CHECK_LAUNCH("amk_attn_bwd");
    return AMK_OK;
  }
  if (stages & AMK_ATTN_BWD_DELTA)
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)((nrow + 15) / 16)), dim3(256), 0, st, p);
  if (stages & AMK_ATTN_BWD_FUSED) {
    // one-pass kernel when the layout / masks allow it, else the two recompute kernels
    const int keys = (stages & AMK_ATTN_BWD_KEYS256) ? 256 : ((stages & AMK_ATTN_BWD_KEYS128) ? 128 : 0);
    if (launch_attn_bwd_fused(p, keys, st)) stages &= ~(AMK_ATTN_BWD_DKDV | AMK_ATTN_BWD_DQ);
    else stages |= AMK_ATTN_BWD_DKDV | AMK_ATTN_BWD_DQ;
  }
  if (stages & AMK_ATTN_BWD_DKDV) {
    if (causal_mask) hipLaunchKernelGGL(attn_bwd_dkdv_kernel<true>, dim3((unsigned)nk), dim3(WG), 0, st, p);
    else hipLaunchKernelGGL(attn_bwd_dkdv_kernel<false>, dim3((unsigned)nk), dim3(WG), 0, st, p);
  }
  if (stages & AMK_ATTN_BWD_DQ) {
    if (causal_mask) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3((unsigned)nq), dim3(WG), 0, st, p);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3((unsigned)nq), dim3(WG), 0, st, p);
  }
  AMK_CHECK_LAUNCH("amk_attn_bwd");
  return AMK_OK;
}

extern "C" int64_t amk_attn_bwd_ws_floats(int B, int H, int I, int J, int stages) {
  if (B <= 0 || H <= 0 || I <= 0 || J <= 0) return 0;
  int64_t n = ((int64_t)B * H * I + 3) & ~(int64_t)3;
  if ((stages & AMK_ATTN_BWD_FUSED) && (stages & AMK_ATTN_BWD_DQ_REPRO)) {
    const int keys = fused_keys_per_wg(J, (stages & AMK_ATTN_BWD_KEYS256) ? 256 : ((stages & AMK_ATTN_BWD_KEYS128) ? 128 : 0));
    const int nk = (J + keys - 1) / keys;
    if (nk > 1) n += (int64_t)nk * B * I * H * D;
  }
  return n;
}

extern "C" int amk_attn_bwd(const float* q, const float* k, const float* v, const float* o,
                            const float* stats, const float* d_o,
                            float* dq, float* dk, float* dv, float* delta_ws,
                            const uint8_t* key_mask, const uint8_t* causal_mask,
                            int B, int H, int I, int J, int Dh,
                            int64_t q_sb, int64_t q_st, int64_t q_sh,
                            int64_t k_sb, int64_t k_st, int64_t k_sh,
                            int64_t v_sb, int64_t v_st, int64_t v_sh,
                            int64_t o_sb, int64_t o_st, int64_t o_sh,
                            int64_t do_sb, int64_t do_st, int64_t do_sh,
                            int64_t dq_sb, int64_t dq_st, int64_t dq_sh,
                            int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                            int64_t dv_sb, int64_t dv_st, int64_t dv_sh,
                            float scale, int stages, void* stream) {
  return attn_bwd_impl(nullptr, q, k, v, o, stats, d_o, dq, dk, dv, delta_ws, key_mask, causal_mask, B, H, I, J, Dh,
                       q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh, do_sb, do_st, do_sh,
                       dq_sb, dq_st, dq_sh, dk_sb, dk_st, dk_sh, dv_sb, dv_st, dv_sh, scale, stages, stream);
}

extern "C" int amk_attn_bwd_kept(const float* scores, const float* q, const float* k, const float* v, const float* o,
                                 const float* stats, const float* d_o,
                                 float* dq, float* dk, float* dv, float* delta_ws,
                                 const uint8_t* key_mask, const uint8_t* causal_mask,
                                 int B, int H, int I, int J, int Dh,
                                 int64_t q_sb, int64_t q_st, int64_t q_sh,
                                 int64_t k_sb, int64_t k_st, int64_t k_sh,
                                 int64_t v_sb, int64_t v_st, int64_t v_sh,
                                 int64_t o_sb, int64_t o_st, int64_t o_sh,
                                 int64_t do_sb, int64_t do_st, int64_t do_sh,
                                 int64_t dq_sb, int64_t dq_st, int64_t dq_sh,
                                 int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                                 int64_t dv_sb, int64_t dv_st, int64_t dv_sh,
                                 float scale, int stages, void* stream) {
  AMK_CHECK_ARG(scores, "amk_attn_bwd_kept: null scores buffer");
  return attn_bwd_impl(scores, q, k, v, o, stats, d_o, dq, dk, dv, delta_ws, key_mask, causal_mask, B, H, I, J, Dh,
                       q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh, do_sb, do_st, do_sh,
                       dq_sb, dq_st, dq_sh, dk_sb, dk_st, dk_sh, dv_sb, dv_st, dv_sh, scale, stages, stream);
}
