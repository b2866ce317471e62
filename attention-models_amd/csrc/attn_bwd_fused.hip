// Fused softmax-attention backward for gfx950: ONE pass, five products, dQ by f32 atomics.
//
// The two-kernel backward of attn_bwd.hip recomputes S twice and dP twice (7 products, bitwise
// reproducible, no atomics).  This kernel computes each product once (cdna guide, Appendix B
// "Attention backward"): a workgroup (4 waves) owns 128 keys of one (batch, head) -- v of a
// wave's 32 keys lives in registers, the K block in LDS, dK^T / dV^T accumulate in registers -- and sweeps the
// queries in tiles of 32:
//
//   S  = Q K^T, dP = dO V^T     32x32x2 MFMA, A = q / dO rows from LDS, B = k rows (LDS) / v registers
//   P, dS                       key on the lane: fills are per-lane constants, row stats from LDS
//   dV^T += dO^T P, dK^T += Q^T dS   A = dO / q columns from LDS, B = the accumulators as they stand
//   dS -> LDS ([query][key], all 128 keys of the workgroup), one barrier, then
//   dQ tile (32 x 64) = dS (32 x 128) K (128 x 64) with v_mfma_f32_16x16x4_f32: every wave owns two
//        16x16 output blocks over ALL 128 keys, so no cross-wave reduction; the result is added
//        to dq with global_atomic_add_f32 (dq is zeroed by the launcher).  Atomic volume is
//        8 KiB per (32 query x 128 key) tile = one byte per 320 FLOP.
// Two workgroup barriers per query tile (the next q / dO tile is committed under the dQ product);
// workgroups start at rotated query tiles so that co-resident ones are not in the same phase.
// Results differ from run to run in the last bits of dq only (f32 atomic arrival order); dk, dv are
// reproducible.  The two-kernel path stays available (AMK_ATTN_BWD_DKDV | AMK_ATTN_BWD_DQ).
#include "attn_common.h"

namespace amk_attn {

constexpr int TQ = 32;             // queries per tile
constexpr int DS_STRIDE = BLK + 4; // dS tile row stride (floats): 16-B aligned rows, b128 row reads
constexpr int FUSED_LDS_FLOATS = 2 * TQ * LDS_STRIDE + BLK * LDS_STRIDE + TQ * DS_STRIDE + 3 * TQ;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  // D(16x16) += A(16x4) B(4x16): lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15];
  // accumulator register r of lane l is D[4 * (l >> 4) + r][l & 15].
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(WG, 2) void attn_bwd_fused_kernel(BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                           // [TQ][LDS_STRIDE]  (q * scale * log2 e)
  float* Gs = Qs + TQ * LDS_STRIDE;           // [TQ][LDS_STRIDE]  dO
  float* Kc = Gs + TQ * LDS_STRIDE;           // [BLK][LDS_STRIDE] this workgroup's K rows
  float* dSl = Kc + BLK * LDS_STRIDE;         // [TQ][DS_STRIDE]   dS of the current tile
  float* Ms = dSl + TQ * DS_STRIDE;
  float* Ls = Ms + TQ;
  float* Ds = Ls + TQ;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = wg % p.nkblk;
  const int bh = wg / p.nkblk;
  const int h = bh % p.H, b = bh / p.H;
  const int kj = kb * BLK + wave * 32 + ln;  // this lane's key row
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
  const bool plain = __all(kfill == 0.f);  // wave-uniform: none of this wave's keys is filled

  // K rows of the whole workgroup -> LDS once (B operand of the dQ product)
  {
    RowStagerT<BLK / 2> kl;  // two passes of 64 rows
    const float* kblk = kbase + (int64_t)kb * BLK * p.ks.st;
    kl.init(kblk, p.ks.st, min(BLK, p.J - kb * BLK), tid);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 t[4];
      kl.load(t);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) st4(&Kc[(64 * half + (tid >> 4) + 16 * ps) * LDS_STRIDE + (tid & 15) * 4], t[ps]);
    }
  }

  const float* qbase = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* gbase = p.d_o + (int64_t)b * p.dos.sb + (int64_t)h * p.dos.sh;
  const float* stbase = p.stats + ((int64_t)b * p.H + h) * p.I * 2;
  const float* dlbase = p.delta + ((int64_t)b * p.H + h) * p.I;
  float* dqbase = p.dq + (int64_t)b * p.dqs.sb + (int64_t)h * p.dqs.sh;

  const int srow = tid >> 4, scol = (tid & 15) * 4;
  float4 qst[2], gst[2];
  float mst = 0.f, lst = 0.f, dst = 0.f;
  RowStagerT<TQ> qload, gload;
  qload.init(qbase, p.qs.st, p.I, tid);
  gload.init(gbase, p.dos.st, p.I, tid);
  auto prefetch = [&](int i0) {
    qload.load(qst);
    gload.load(gst);
    if (tid < TQ) {
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
    for (int ps = 0; ps < 2; ++ps) {
      const int r = srow + 16 * ps;
      st4(&Qs[r * LDS_STRIDE + scol], make_float4(qst[ps].x * sc, qst[ps].y * sc, qst[ps].z * sc, qst[ps].w * sc));
      st4(&Gs[r * LDS_STRIDE + scol], gst[ps]);
    }
    if (tid < TQ) { Ms[tid] = mst; Ls[tid] = lst; Ds[tid] = dst; }
  };

  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

  // dQ product geometry: 16x16x4 MFMA, lane = (column c = l & 15, k-group kg = l >> 4)
  const int c16 = lane & 15, kg = lane >> 4;
  const int qhalf = wave & 1;        // which 16 of the tile's 32 queries
  const int dcol0 = 32 * (wave >> 1); // this wave's 32 output columns (two 16-wide blocks)
  const float* ds_row = &dSl[(16 * qhalf + c16) * DS_STRIDE + 32 * kg];  // A: dS[query][32*kg + s]
  const float* kc_col = &Kc[(32 * kg) * LDS_STRIDE + dcol0 + c16];       // B: K[32*kg + s][dcol]

  const int ntile = (p.I + TQ - 1) / TQ;
  // Two barriers per query tile.  After the dS barrier every wave is past the phases that read the
  // q / dO / stats tiles, so the next tile is committed there -- under the dQ product, which only
  // reads dS and K -- and the barrier that closes the dQ product also publishes it.
  // Co-resident workgroups start at different query tiles (the order of the dQ atomics and of the
  // dK / dV accumulation does not matter): it keeps their MFMA and barrier phases from lining up.
  const int rot = (int)((blockIdx.x * 5u) % (unsigned)ntile);
  auto tile_of = [&](int t) { const int x = t + rot; return x >= ntile ? x - ntile : x; };
  qload.seek(rot, p.qs.st, tid);
  gload.seek(rot, p.dos.st, tid);
  prefetch(rot * TQ);
  __syncthreads();  // the K block is in LDS
  commit();
  __syncthreads();
  if (ntile > 1) {
    const int nx = tile_of(1);
    qload.seek(nx, p.qs.st, tid);
    gload.seek(nx, p.dos.st, tid);
    prefetch(nx * TQ);
  }
  for (int t = 0; t < ntile; ++t) {
    const int i0 = tile_of(t) * TQ;

    // ---- S and dP for the tile's 32 queries x this wave's 32 keys (2 x 32 MFMAs)
    f32x16 s = zero16(), dp = zero16();
    {
      const float* qr = &Qs[ln * LDS_STRIDE + 32 * hf];
      const float* gr = &Gs[ln * LDS_STRIDE + 32 * hf];
      const float* kr = &Kc[(32 * wave + ln) * LDS_STRIDE + 32 * hf];  // this lane's key row
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        const float4 a = ld4(qr + 4 * s4);
        const float4 c = ld4(gr + 4 * s4);
        const float4 kk = ld4(kr + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = mfma32(f4(a, e), f4(kk, e), s);
          dp = mfma32(f4(c, e), vreg[4 * s4 + e], dp);
        }
      }
    }
    // ---- P and dS (register r of this lane is query acc_row(r, hf))
    if (plain) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 m4 = ld4(&Ms[8 * g + 4 * hf]);
        const float4 l4 = ld4(&Ls[8 * g + 4 * hf]);
        const float4 d4 = ld4(&Ds[8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = __builtin_amdgcn_exp2f(s[r] - f4(m4, e)) * f4(l4, e);
          s[r] = pr;
          dp[r] = pr * (dp[r] - f4(d4, e));
        }
      }
    } else {
      const bool filled = kfill != 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 m4 = ld4(&Ms[8 * g + 4 * hf]);
        const float4 l4 = ld4(&Ls[8 * g + 4 * hf]);
        const float4 d4 = ld4(&Ds[8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float tt = filled ? kfill : s[r];
          const float pr = __builtin_amdgcn_exp2f(tt - f4(m4, e)) * f4(l4, e);
          s[r] = pr;
          dp[r] = filled ? 0.f : pr * (dp[r] - f4(d4, e));
        }
      }
    }
    // ---- dS -> LDS as [query][key] for the workgroup-wide dQ product
#pragma unroll
    for (int r = 0; r < 16; ++r) dSl[acc_row(r, hf) * DS_STRIDE + 32 * wave + ln] = dp[r];

    // ---- dV^T += dO^T P ; dK^T += (q*scale*log2e)^T dS   (2 x 32 MFMAs)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* gc = &Gs[acc_row(r, hf) * LDS_STRIDE + ln];
      const float* qc = &Qs[acc_row(r, hf) * LDS_STRIDE + ln];
      dv0 = mfma32(gc[0], s[r], dv0);
      dv1 = mfma32(gc[32], s[r], dv1);
      dk0 = mfma32(qc[0], dp[r], dk0);
      dk1 = mfma32(qc[32], dp[r], dk1);
    }
    __syncthreads();  // every wave's dS columns are in LDS; the q / dO / stats tiles are dead
    if (t + 1 < ntile) {
      commit();
      if (t + 2 < ntile) {
        const int nx = tile_of(t + 2);
        qload.seek(nx, p.qs.st, tid);
        gload.seek(nx, p.dos.st, tid);
        prefetch(nx * TQ);
      }
    }

    // ---- dQ (16 queries x 32 columns per wave) = dS (16 x 128) K (128 x 32): 64 MFMAs 16x16x4
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const float4 a = ld4(ds_row + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* kc = kc_col + (4 * s4 + e) * LDS_STRIDE;
        q0 = mfma16(f4(a, e), kc[0], q0);
        q1 = mfma16(f4(a, e), kc[16], q1);
      }
    }
    {
      const float sc = p.scale;
      const int qi0 = i0 + 16 * qhalf + 4 * kg;
      float* dst_ = dqbase + (int64_t)qi0 * p.dqs.st + dcol0 + c16;
      if (qi0 + 3 < p.I) {  // all four rows of this lane inside the sequence (the common case)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          atomicAdd(dst_ + (int64_t)r * p.dqs.st, q0[r] * sc);
          atomicAdd(dst_ + (int64_t)r * p.dqs.st + 16, q1[r] * sc);
        }
      } else {
        for (int r = 0; r < 4; ++r) {
          if (qi0 + r < p.I) {
            atomicAdd(dst_ + (int64_t)r * p.dqs.st, q0[r] * sc);
            atomicAdd(dst_ + (int64_t)r * p.dqs.st + 16, q1[r] * sc);
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

// Launch the fused kernel.  dq must be the dense (B, I, H, 64) layout so that it can be zeroed
// with one memset; returns false (nothing launched) when the layout or the masks rule it out.
bool launch_attn_bwd_fused(const BwdParams& p, hipStream_t st) {
  if (p.causal_mask) return false;
  if (!(p.dqs.sh == D && p.dqs.st == (int64_t)p.H * D && p.dqs.sb == (int64_t)p.I * p.H * D)) return false;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  FUSED_LDS_FLOATS * (int)sizeof(float)) == hipSuccess;
  if (!attr_ok) return false;
  if (hipMemsetAsync(p.dq, 0, (size_t)p.B * p.I * p.H * D * sizeof(float), st) != hipSuccess) return false;
  const int64_t nk = (int64_t)p.B * p.H * p.nkblk;
  hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3((unsigned)nk), dim3(WG), FUSED_LDS_FLOATS * sizeof(float), st, p);
  return true;
}

}  // namespace amk_attn
