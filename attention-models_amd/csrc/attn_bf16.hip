// Softmax attention on bf16 operands for the mixed-precision mode (SURVEY.md section 8a2 under the reference's shipped
// training precision: cfg/vitvqgan.yaml:73 trains under accelerate's bf16 autocast; trainers/vitgqgan.py:149,170).
// Under autocast the reference's q / kv projections and both einsums of models/softmax_attention.py:62-76 run in
// bf16 and the softmax in f32; here q, k, v, o (and dO, dq, dk, dv) are bf16 tensors, the two contractions run on
// v_mfma_f32_32x32x16_bf16 with f32 accumulators, scores / softmax / statistics stay f32 (the reference rounds the
// scores to bf16 before its softmax; this kernel does not).  Head dim 64; other head dims and f32 tensors take the
// exact-f32 kernels.  Key-padding and causal masks (models/softmax_attention.py:65-71: masked_fill(-1e9) on the scaled
// scores, True = keep for context_mask, True = masked for causal_mask) are template variants <MASKED> of both kernels:
// the unmasked loops are unchanged; the masked ones put -1e9 / scale in place of a raw score (so that c2 * fill is the
// reference's -1e9 in the log2 domain), form the exponent as (s * c2) - (m * c2) with two rounded operations -- exactly 0
// where s is the row maximum, also at 1.4e9, so a fully masked row keeps the reference's uniform weights -- and keep the
// normaliser 1 / l as its own factor in the backward (m + log2 l would swallow l on such a row).
//
// Forward: the structure of attn_fwd.hip -- 4 waves x 32 queries, query on the lane, S^T = K Q^T, the S^T
// accumulators packed to bf16 are the B operand of O^T += V^T P^T -- with
//   K tile [64 keys][64 d] bf16 in LDS, row stride 144 B: A fragments are 16-byte row reads (conflict-free);
//   V tile [64 keys][64 d] bf16 in LDS AS STORED, row stride 192 B: the A fragments of V^T (8 keys of one head
//   dim, in the key order of a packed accumulator) come from ds_read_b64_tr_b16, the hardware transposing read
//   (4 rows x 16 columns per 16 lanes; 4 rows x 64 B per half-wave land on 64 distinct banks at this stride).
// Backward: one pass with the key on the lane (as attn_bwd_fused.hip): a workgroup = 8 waves x 32 keys; K and V
// fragments of the wave's keys in registers, dK^T / dV^T in accumulators; per 32-query tile S = Q K^T and
// dP = dO V^T (A = rows of Q / dO from LDS), P and dS in registers, dV^T += dO^T P and dK^T += Q^T dS with the
// packed accumulators as B operands and transposing reads of the dO / Q tiles as A operands; dS crosses LDS once
// ([key][query] image) and dQ(32 x 64) = dS K over the workgroup's 256 keys runs on v_mfma_f32_16x16x32_bf16, one
// 16x16 output tile per wave.  dQ leaves as per-key-block f32 partials (plain stores) that a second launch sums in
// key-block order and rounds to bf16: no atomics, bitwise reproducible.
#include "attn_common.h"

namespace amk_attn_bf16 {

using amk_attn::Strides;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int KSTR = 72;   // bf16 per row of a row-read image (64 + 8): 144 B
constexpr int TSTR = 96;   // bf16 per row of a transposed-read image (64 + 32): 192 B

struct Params {
  const __bf16 *q, *k, *v, *o, *d_o;
  __bf16 *out, *dq, *dk, *dv;
  float* stats;        // (B, H, I, 2): row max in the log2 domain, row sum
  float* delta;        // (B, H, I) x 4: the backward row constants {-(m + log2 l) / c2, -rowsum(dO * O), m, 1 / l}
  const uint8_t *key_mask, *causal_mask;   // (B, J) 1 = keep; (I, J) 1 = masked; or null
  float* dq_part;      // (nkblk, B, I, H, 64) f32
  int B, H, I, J;
  Strides qs, ks, vs, os, dos, dqs, dks, dvs;
  float scale;
  float pinf;
  int nqblk, nkblk;
};

__device__ __forceinline__ f32x16 mfmab(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ float vmax(float a, float b, float pinf) { return __builtin_amdgcn_fmed3f(a, b, pinf); }
// v_max3_f32 (this file is built with -fno-honor-nans -mno-amdgpu-ieee, see the Makefile: no canonicalising v_max in front)
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
typedef float f32x2 __attribute__((ext_vector_type(2)));

// transposing read: this lane's 4 elements = one column of a 4-row x 16-column block of 16-bit values
__device__ __forceinline__ bf16x4 tr_read(const __bf16* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
  return r;
}
// registers 8t .. 8t+7 of a 32x32 accumulator as the 8 bf16 of a k-slot (the B operand of a product that sums over
// the accumulator's rows)
__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int t) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)s[8 * t + j];
  return r;
}
// A fragment of X^T for a product that sums over the ROWS of the LDS tile X [row][col], rows in packed-accumulator
// order: lane (col = c0 + (l & 31), half) gets rows 16 t + 4 half + (0..3) and + 8.
//   img: tile base; stride in elements; c0: first column of the 32-column block
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* img, int stride, int t, int c0, int lane) {
  const int hf = lane >> 5, grp = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  const __bf16* a = img + (16 * t + 4 * hf + q) * stride + c0 + 16 * grp + 4 * pp;
  return cat4(tr_read(a), tr_read(a + 8 * stride));
}

// barrier that orders LDS only (global loads in flight stay in flight)
__device__ __forceinline__ void lds_barrier_bf() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---------------------------------------------------------------------------------------------------------------
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_bf16_fwd_kernel(Params p) {
  // two LDS stages: tile t + 1 is committed while tile t is being read, ONE barrier per tile (round 4; the single-stage
  // loop had two, and at ~1700 cycles per tile the pair showed)
  __shared__ __attribute__((aligned(16))) __bf16 Ks2[2][64 * KSTR];
  __shared__ __attribute__((aligned(16))) __bf16 Vs2[2][64 * TSTR];
  __shared__ __attribute__((aligned(16))) float Kfill2[2][MASKED ? 64 : 4];   // MASKED: per key of the tile 0 keep / fill / -inf past J
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  // (wave-uniform; said explicitly: built without IEEE mode the compiler treats the quotients as divergent and wraps the loads
  // through the K / V descriptors in waterfall loops)
  const int wg = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int qb = __builtin_amdgcn_readfirstlane(wg % p.nqblk), bh = __builtin_amdgcn_readfirstlane(wg / p.nqblk);
  const int h = __builtin_amdgcn_readfirstlane(bh % p.H), b = __builtin_amdgcn_readfirstlane(bh / p.H);
  const int qi = qb * 128 + wave * 32 + ln;
  const bool qvalid = qi < p.I;

  // Q^T operand: lane (query, half) holds q[query][16 c + 8 half + j], j = 0..7, for the four k-blocks c
  bf16x8 qf[4];
  {
    const __bf16* qp = p.q + (int64_t)b * p.qs.sb + (int64_t)qi * p.qs.st + (int64_t)h * p.qs.sh + 8 * hf;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (qvalid) qf[c] = *reinterpret_cast<const bf16x8*>(qp + 16 * c);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[c][j] = (__bf16)0.f;
      }
    }
  }
  // staging: a 64-row tile of 128-byte rows = 512 pieces of 16 B: two per thread for K, two for V
  const __bf16* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const __bf16* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)(((int64_t)(p.J - 1) * p.ks.st + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)(((int64_t)(p.J - 1) * p.vs.st + 64) * 2), 0x00020000);
  const int srow = tid >> 3, sch = (tid & 7) * 8;   // rows srow and srow + 32, 8 bf16 at column sch
  int koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    koff[i] = (int)(((int64_t)(srow + 32 * i) * p.ks.st + sch) * 2);
    voff[i] = (int)(((int64_t)(srow + 32 * i) * p.vs.st + sch) * 2);
  }
  const int kstep = (int)(64 * p.ks.st * 2), vstep = (int)(64 * p.vs.st * 2);
  float4 kst[2], vst[2];
  const float fill_raw = -1.0e9f / p.scale;   // masked_fill(-1e9) acts on the SCALED scores; s here is the raw q . k
  const uint8_t* kmrow = MASKED && p.key_mask ? p.key_mask + (int64_t)b * p.J : nullptr;
  const uint8_t* cmrow = MASKED && p.causal_mask ? p.causal_mask + (int64_t)min(qi, p.I - 1) * p.J : nullptr;
  float fillst = 0.f;
  auto prefetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      kst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, koff[i] + t * kstep, 0, 0));
      vst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, voff[i] + t * vstep, 0, 0));
    }
    if (MASKED) {   // (all threads, clamped address: no branch around the load)
      const int j = t * 64 + (tid & 63);
      const uint8_t keep = kmrow ? kmrow[min(j, p.J - 1)] : (uint8_t)1;
      fillst = j >= p.J ? -INFINITY : (keep ? 0.f : fill_raw);
    }
  };
  auto commit = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(&Ks2[stage][(srow + 32 * i) * KSTR + sch]) = kst[i];
      *reinterpret_cast<float4*>(&Vs2[stage][(srow + 32 * i) * TSTR + sch]) = vst[i];
    }
    if (MASKED) Kfill2[stage][tid & 63] = fillst;   // (four waves write the same 64 values)
  };
  const float c2 = p.scale * AMK_LOG2E;
  f32x16 o0 = zero16(), o1 = zero16();
  float m_run = -INFINITY, l_run = 0.f;   // m_run: running max of the RAW scores (unmasked: a lazy reference, see below)
  float mc_run = 0.f;                     // m_run * c2
  const float lazy_tau = 8.f / c2;        // the reference moves when a row's maximum passes it by 2^8 in the exponent
  const int ntile = (p.J + 63) / 64;
  prefetch(0);
  commit(0);
  prefetch(1);         // (unconditional throughout: a tile past the sequence is outside the descriptors and reads zeros; a
                       //  branch around a memory instruction would make every wait of the loop a full drain)
  lds_barrier_bf();
  for (int t = 0; t < ntile; ++t) {
    const int j0 = t * 64;
    const __bf16* Ks = Ks2[t & 1];
    const __bf16* Vs = Vs2[t & 1];
    const float* Kfill = Kfill2[t & 1];
    // ---- S^T = K Q^T: two 32-key blocks x four 16-deep steps
    f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(&Ks[ln * KSTR + 16 * c + 8 * hf]);
      const bf16x8 kb = *reinterpret_cast<const bf16x8*>(&Ks[(32 + ln) * KSTR + 16 * c + 8 * hf]);
      s0 = mfmab(ka, qf[c], s0);
      s1 = mfmab(kb, qf[c], s1);
    }
    if (MASKED) {
      // fills: per key of the tile from LDS; the causal bytes of this lane's query row for its 2 x 16 keys
      unsigned cb0 = 0, cb1 = 0;
      if (cmrow) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ja = j0 + acc_row(r, hf);
          cb0 |= (cmrow[min(ja, p.J - 1)] ? 1u : 0u) << r;
          cb1 |= (cmrow[min(ja + 32, p.J - 1)] ? 1u : 0u) << r;
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 f0 = *reinterpret_cast<const float4*>(&Kfill[8 * g + 4 * hf]);
        const float4 f1 = *reinterpret_cast<const float4*>(&Kfill[32 + 8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float fa = amk_attn::f4(f0, e), fb = amk_attn::f4(f1, e);
          float t0 = ((cb0 >> r) & 1u) ? fill_raw : s0[r], t1 = ((cb1 >> r) & 1u) ? fill_raw : s1[r];
          s0[r] = fa == 0.f ? t0 : fa;     // (key-padding fill and the -inf past the sequence win over the causal fill: same value or -inf)
          s1[r] = fb == 0.f ? t1 : fb;
        }
      }
    } else if (j0 + 64 > p.J) {   // keys beyond the sequence (the last tile only): weight 0
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ja = j0 + acc_row(r, hf);
        if (ja >= p.J) s0[r] = -INFINITY;
        if (ja + 32 >= p.J) s1[r] = -INFINITY;
      }
    }
    if (!MASKED) {
      // Unmasked: the lazy reference of csrc/attn_fwd.hip's plain kernel.  softmax does not care which constant is
      // subtracted, so the reference m_run only has to stay within 2^8 of the true maximum: the 32 accumulator rescales
      // (paid in almost every tile when 32 independent rows share a wave) become rare, the row maximum is 16 v_max3, the
      // row sum 16 packed adds.  (m_run, l) leave as the statistics: the backward forms the same P from them.
      float mx = max3(s0[0], s1[0], s0[1]);
      mx = max3(mx, s1[1], s0[2]);
#pragma unroll
      for (int r = 2; r < 15; ++r) mx = max3(mx, s1[r], s0[r + 1]);
      mx = __builtin_fmaxf(mx, s1[15]);
      mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (t == 0 || __any(mx > m_run + lazy_tau)) {
        const float m_new = t == 0 ? mx : __builtin_fmaxf(m_run, mx);
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
          l_run *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        m_run = m_new;
        mc_run = m_new * c2;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c2, -mc_run));
        s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c2, -mc_run));
      }
      f32x2 la = {0.f, 0.f}, lb = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        la += (f32x2){s0[r], s0[r + 1]};
        lb += (f32x2){s1[r], s1[r + 1]};
      }
      la += lb;
      l_run += la.x + la.y;
    } else {
      float mx = vmax(s0[0], s1[0], p.pinf);
  #pragma unroll
      for (int r = 1; r < 16; ++r) mx = vmax(mx, vmax(s0[r], s1[r], p.pinf), p.pinf);
      mx = vmax(mx, __shfl_xor(mx, 32, 64), p.pinf);
      const float m_new = vmax(m_run, mx, p.pinf);
      const float mc = __fmul_rn(m_new, c2);
      float lsum = 0.f;
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        // MASKED: two rounded operations, so that s == m gives exactly 0 at any magnitude (a fused multiply-add leaves the
        // rounding error of m * c2, +-64 at the fill value's 1.4e9)
        const float p0 = __builtin_amdgcn_exp2f(MASKED ? __fsub_rn(__fmul_rn(s0[r], c2), mc) : __builtin_fmaf(s0[r], c2, -mc));
        const float p1 = __builtin_amdgcn_exp2f(MASKED ? __fsub_rn(__fmul_rn(s1[r], c2), mc) : __builtin_fmaf(s1[r], c2, -mc));
        s0[r] = p0;
        s1[r] = p1;
        lsum += p0 + p1;
      }
      if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f(MASKED ? __fsub_rn(__fmul_rn(m_run, c2), mc) : (m_run - m_new) * c2);
        l_run *= alpha;
  #pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        m_run = m_new;
      }
      l_run += lsum;
    }
    // ---- O^T += V^T P^T: four 16-key slots x two 32-dim blocks
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x8 pp = (g < 2) ? pack8(s0, g & 1) : pack8(s1, g & 1);
      const __bf16* vt = Vs + (32 * (g >> 1)) * TSTR;     // the 32-key block of this slot
      const bf16x8 va = tr_frag(vt, TSTR, g & 1, 0, lane);
      const bf16x8 vb = tr_frag(vt, TSTR, g & 1, 32, lane);
      o0 = mfmab(va, pp, o0);
      o1 = mfmab(vb, pp, o1);
    }
    commit((t + 1) & 1);   // tile t + 1 (requested a tile ago) into the other stage; past the last tile: zeros nobody reads
    prefetch(t + 2);
    lds_barrier_bf();      // every wave is done with stage t & 1 and has written its part of stage (t + 1) & 1
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (qvalid) {
    __bf16* op = p.out + (int64_t)b * p.os.sb + (int64_t)qi * p.os.st + (int64_t)h * p.os.sh + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = (__bf16)(o0[4 * g + e] * inv); c[e] = (__bf16)(o1[4 * g + e] * inv); }
      *reinterpret_cast<bf16x4*>(op + 8 * g) = a;
      *reinterpret_cast<bf16x4*>(op + 32 + 8 * g) = c;
    }
    if (hf == 0) {
      float* sp = p.stats + (((int64_t)b * p.H + h) * p.I + qi) * 2;
      sp[0] = __fmul_rn(m_run, c2);   // log2 domain, as the f32 kernels store it
      sp[1] = l_tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// delta[b, h, i] = sum_d dO[b, i, h, d] * O[b, i, h, d]   (one wave per 4 rows: 16 lanes x 4 elements per row)
__global__ __launch_bounds__(256) void attn_bf16_delta_kernel(Params p) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);   // over B * H * I
  const int part = threadIdx.x & 15;
  if (row >= (int64_t)p.B * p.H * p.I) return;
  const int i = (int)(row % p.I);
  const int64_t bh = row / p.I;
  const int h = (int)(bh % p.H), b = (int)(bh / p.H);
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.d_o + (int64_t)b * p.dos.sb + (int64_t)i * p.dos.st + (int64_t)h * p.dos.sh + 4 * part);
  const bf16x4 c = *reinterpret_cast<const bf16x4*>(p.o + (int64_t)b * p.os.sb + (int64_t)i * p.os.st + (int64_t)h * p.os.sh + 4 * part);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s += (float)a[e] * (float)c[e];
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (part == 0) {
    // the row constants as the backward's MFMA chains start from them: S' = q.k - (m + log2 l) / c2 (so that
    // P = exp2(c2 S') comes out normalised) and dP' = dO.v - delta
    const float m = p.stats[2 * row], l = p.stats[2 * row + 1];
    const float c2 = p.scale * AMK_LOG2E;
    reinterpret_cast<float4*>(p.delta)[row] = make_float4(-(m + __builtin_log2f(l)) / c2, -s, m, 1.f / l);
  }
}

#ifndef A16_ABLATE
#define A16_ABLATE 0   // timing experiments only (tools/ablate_attn_bf16.sh): bits switch parts of the backward off
#endif
constexpr int BW = 8;            // waves per backward workgroup
constexpr int BKEYS = 32 * BW;   // keys per workgroup
constexpr int DSTR = 48;         // bf16 per row of the dS^T image [key][32 queries]

__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// 16x16x32 operand out of an LDS tile X [row][col] by transposing reads, for a product that sums over X's ROWS in
// natural order: lane (col = c0 + (l & 15), kq = l >> 4) gets rows r0 + 8 kq + (0..7)
__device__ __forceinline__ bf16x8 tr_frag16(const __bf16* img, int stride, int r0, int c0, int lane) {
  const int kq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const __bf16* a = img + (r0 + 8 * kq + q) * stride + c0 + 4 * pp;
  return cat4(tr_read(a), tr_read(a + 4 * stride));
}

template <bool MASKED, bool CAUSAL>
__global__ __launch_bounds__(64 * BW, 1) void attn_bf16_bwd_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* Kt = reinterpret_cast<__bf16*>(smem_raw);                 // [256 keys][TSTR]
  __bf16* dSs = Kt + BKEYS * TSTR;                                  // 2 x [256 keys][DSTR]
  __bf16* tiles = dSs + 2 * BKEYS * DSTR;                           // 2 stages x {Qrow, Qtr, dOrow, dOtr}
  constexpr int STG = 2 * (32 * KSTR + 32 * TSTR);
  float* stat = reinterpret_cast<float*>(tiles + 2 * STG);          // 2 stages x {-(m + log2 l) / c2, -delta, m, 1 / l} x 32

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = lane & 31, hf = lane >> 5;
  const int wg = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int kb = __builtin_amdgcn_readfirstlane(wg % p.nkblk), bh = __builtin_amdgcn_readfirstlane(wg / p.nkblk);
  const int h = __builtin_amdgcn_readfirstlane(bh % p.H), b = __builtin_amdgcn_readfirstlane(bh / p.H);
  const int key0 = kb * BKEYS;
  const int key = key0 + 32 * wave + ln;
  const bool kvalid = key < p.J;
  const __bf16* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const __bf16* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;
  // this lane's key: K and V fragments for S = Q K^T and dP = dO V^T (B operands: k = head dim, column = key)
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (kvalid) {
      kf[c] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * p.ks.st + 16 * c + 8 * hf);
      vf[c] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * p.vs.st + 16 * c + 8 * hf);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { kf[c][j] = (__bf16)0.f; vf[c][j] = (__bf16)0.f; }
    }
  }
  // the workgroup's K rows as an LDS image for dQ = dS K (rows past the sequence: zeros)
  {
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)(((int64_t)(p.J - 1) * p.ks.st + 64) * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 64 * i, ch = (tid & 7) * 8;
      const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, (int)(((int64_t)(key0 + row) * p.ks.st + ch) * 2), 0, 0));
      *reinterpret_cast<float4*>(&Kt[row * TSTR + ch]) = v;
    }
  }
  // query-tile staging: threads 0..255 one 16-byte piece of Q, threads 256..511 one of dO; threads 0..31 the statistics
  const bool isq = wave < 4;   // (wave-uniform: a descriptor chosen per lane would be loaded through a waterfall loop)
  const int pt = tid & 255, prow = pt >> 3, pch = (pt & 7) * 8;
  const __bf16* tbase = isq ? p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh : p.d_o + (int64_t)b * p.dos.sb + (int64_t)h * p.dos.sh;
  const int64_t tst = isq ? p.qs.st : p.dos.st;
  const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tbase, 0, (int)(((int64_t)(p.I - 1) * tst + 64) * 2), 0x00020000);
  const int toff = (int)(((int64_t)prow * tst + pch) * 2), tstep = (int)(32 * tst * 2);
  const float4* rcp = reinterpret_cast<const float4*>(p.delta) + (int64_t)bh * p.I;
  float4 piece[2];                        // two tiles in flight (a tile is ~1 us of work, a load up to 2 us away)
  float4 st_rc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  const int ntile = (p.I + 31) / 32;
  const __amdgpu_buffer_rsrc_t rc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rcp, 0, p.I * 16, 0x00020000);
  // MASKED: this lane's key is filled for every query (key-padding mask); the causal bytes of its column come per tile
  const float fill_raw = -1.0e9f / p.scale;
  const bool kfilled = MASKED && p.key_mask && p.key_mask[(int64_t)b * p.J + min(key, p.J - 1)] == 0;
  // (through a buffer descriptor: 32-bit offsets, rows past I clamped -- the causal mask is (I, J) bytes, I * J < 2^31)
  const bool has_causal = CAUSAL;
  const __amdgpu_buffer_rsrc_t cm_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(has_causal ? p.causal_mask : (const uint8_t*)p.delta), 0,
                                                                           has_causal ? p.I * p.J : 0, 0x00020000);
  const int cm_key = min(key, p.J - 1);
  // No branch around a memory instruction anywhere in the tile loop (every thread requests the row constants, rows
  // past the sequence at a clamped address; dQ leaves through a range-checked descriptor): behind such a branch the
  // compiler's waits stop counting and every wait drains everything in flight -- the tile's dQ stores included.
  auto prefetch = [&](int t, int slot) {
    piece[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(t_rsrc, toff + t * tstep, 0, 0));
    const int i = t * 32 + (tid & 31);
    // rows past the sequence read zeros: P = exp2(0) = 1 there, next to dO = 0 and q = 0 rows: dV, dK get nothing, dS = 0
    if (MASKED) st_rc[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rc_rsrc, i * 16, 0, 0));
    else {
      const float2 two = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rc_rsrc, i * 16, 0, 0));
      st_rc[slot].x = two.x; st_rc[slot].y = two.y;
    }
  };
  auto commit = [&](int stage, int slot) {
    __bf16* base = tiles + stage * STG + (isq ? 0 : 32 * KSTR + 32 * TSTR);
    *reinterpret_cast<float4*>(&base[prow * KSTR + pch]) = piece[slot];                  // row-read image
    *reinterpret_cast<float4*>(&base[32 * KSTR + prow * TSTR + pch]) = piece[slot];      // transposed-read image
    {   // every thread writes the constants of row tid & 31 (sixteen copies of the same value): were it one wave only, the
        // compiler would move the request under that condition -- a branch around a memory instruction again
      float* sb = stat + stage * 128;
      sb[tid & 31] = st_rc[slot].x; sb[32 + (tid & 31)] = st_rc[slot].y;
      if (MASKED) { sb[64 + (tid & 31)] = st_rc[slot].z; sb[96 + (tid & 31)] = st_rc[slot].w; }
    }
  };
  const float c2 = p.scale * AMK_LOG2E;
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const bool tail_keys = key0 + BKEYS > p.J;   // some of the workgroup's keys are past the sequence
  prefetch(0, 0);
  commit(0, 0);
  prefetch(1, 1);   // (tiles past the end: the descriptor returns zeros and nothing is committed)
  prefetch(2, 0);
  __syncthreads();
  float* dqp = p.dq_part + (((int64_t)kb * p.B + b) * p.I * p.H + h) * 64;   // [kb][b][i][h][d]
  // dQ: wave w owns the 16 x 16 block (queries 16 (w & 1), dims 16 (w >> 1)) of every tile, so its K operands -- the
  // same for every tile -- stay in registers
  const int qblk = wave & 1, dblk = wave >> 1;
  // (MASKED: 32 registers the mask logic needs more -- there the operands are re-read from LDS per tile)
  bf16x8 kq[MASKED ? 1 : BKEYS / 32];
  if (!MASKED) {
#pragma unroll
    for (int ks = 0; ks < BKEYS / 32; ++ks) kq[ks] = tr_frag16(Kt, TSTR, 32 * ks, 16 * dblk, lane);
  }
  const __amdgpu_buffer_rsrc_t dq_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dqp, 0, (int)(((int64_t)(p.I - 1) * p.H * 64 + 64) * 4), 0x00020000);
  auto dq_tile = [&](int tt) {
    const __bf16* img = dSs + (tt & 1) * BKEYS * DSTR;
    // the dQ^T block (dims x queries), so that a lane ends with four consecutive dims of ONE query: a 16-byte store
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    int kofs = 0;
    if (MASKED) asm volatile("" : "+v"(kofs));   // (opaque: keeps the compiler from hoisting the K reads out of the tile loop, back into 32 registers)
#pragma unroll
    for (int ks = 0; ks < BKEYS / 32; ++ks)
      acc = mfma16(MASKED ? tr_frag16(Kt + kofs, TSTR, 32 * ks, 16 * dblk, lane) : kq[MASKED ? 0 : ks], tr_frag16(img, DSTR, 32 * ks, 16 * qblk, lane), acc);
    const int i = tt * 32 + 16 * qblk + (lane & 15);   // (tile -1 and rows past the sequence: outside the descriptor, dropped)
    const float4 o = make_float4(acc[0] * p.scale, acc[1] * p.scale, acc[2] * p.scale, acc[3] * p.scale);
    if (!(A16_ABLATE & 1))
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), dq_rsrc,
                                             (i * p.H * 64 + 16 * dblk + 4 * (lane >> 4)) * 4, 0, 0);
  };
  auto tile_step = [&](int t, int slot) {   // slot: where tile t + 1 waits in registers
    const int sg = t & 1;
    const __bf16* Qrow = tiles + sg * STG;
    const __bf16* Qtr = Qrow + 32 * KSTR;
    const __bf16* dOrow = Qtr + 32 * TSTR;
    const __bf16* dOtr = dOrow + 32 * KSTR;
    const float* sb = stat + sg * 128;
    // ---- S' = Q K^T - lse / c2, dP' = dO V^T - delta: rows = queries (registers r <-> rows 8 g + 4 half + e of the
    //      tile), columns = keys (this lane's key); the row constants are the chains' initial accumulators
    f32x16 s, dp;
    unsigned cbits = 0;   // MASKED: bit r = the causal mask fills (query of register r, this lane's key)
    if (CAUSAL) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cbits |= (__builtin_amdgcn_raw_buffer_load_b8(cm_rsrc, min(32 * t + acc_row(r, hf), p.I - 1) * p.J + cm_key, 0, 0) ? 1u : 0u) << r;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 d4 = *reinterpret_cast<const float4*>(&sb[32 + 8 * g + 4 * hf]);
      dp[4 * g] = d4.x; dp[4 * g + 1] = d4.y; dp[4 * g + 2] = d4.z; dp[4 * g + 3] = d4.w;
      if (MASKED) {   // raw scores: the fill replaces them below
        s[4 * g] = 0.f; s[4 * g + 1] = 0.f; s[4 * g + 2] = 0.f; s[4 * g + 3] = 0.f;
      } else {
        const float4 a4 = *reinterpret_cast<const float4*>(&sb[8 * g + 4 * hf]);
        s[4 * g] = a4.x; s[4 * g + 1] = a4.y; s[4 * g + 2] = a4.z; s[4 * g + 3] = a4.w;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(&Qrow[ln * KSTR + 16 * c + 8 * hf]);
      const bf16x8 da = *reinterpret_cast<const bf16x8*>(&dOrow[ln * KSTR + 16 * c + 8 * hf]);
      s = mfmab(qa, kf[c], s);
      dp = mfmab(da, vf[c], dp);
    }
    // ---- dQ of the previous tile (32 queries x 64 dims) = dS (32 x 256 keys) K (256 x 64): issued here, behind the
    //      S / dP chains in the matrix pipe, so that it runs under the exp work below
    if (!(A16_ABLATE & 2)) dq_tile(t - 1);   // (t = 0: an image nobody wrote, rows -32..-1: every store is dropped)
    // ---- P = exp2(c2 S'), dS / scale = P dP' (the scale goes onto dK and dQ where they are stored)
    if (MASKED) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 m4 = *reinterpret_cast<const float4*>(&sb[64 + 8 * g + 4 * hf]);
        const float4 l4 = *reinterpret_cast<const float4*>(&sb[96 + 8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const bool filled = kfilled || ((cbits >> r) & 1u);
          const float tt = filled ? fill_raw : s[r];
          // (s * c2) - m with two rounded operations, as the forward formed it: exactly 0 where the fill is the row maximum
          float pr = __builtin_amdgcn_exp2f(__fsub_rn(__fmul_rn(tt, c2), amk_attn::f4(m4, e))) * amk_attn::f4(l4, e);
          pr = kvalid ? pr : 0.f;
          s[r] = pr;                                 // a filled position still has its weight in P (dV), but passes no gradient
          dp[r] = filled ? 0.f : dp[r] * pr;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pr = (A16_ABLATE & 8) ? s[r] : __builtin_amdgcn_exp2f(s[r] * c2);
        if (tail_keys) pr = kvalid ? pr : 0.f;
        s[r] = pr;
        dp[r] *= pr;
      }
    }
    // ---- dS^T image for dQ: this lane's key row, its 16 queries as four 8-byte pieces (image t & 1: the dQ product
    //      of tile t runs in the NEXT iteration, beside that tile's S / dP work, so one barrier per tile is enough)
    {
      __bf16* row = dSs + (t & 1) * BKEYS * DSTR + (32 * wave + ln) * DSTR + 4 * hf;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (__bf16)dp[4 * g + e];
        *reinterpret_cast<bf16x4*>(row + 8 * g) = w;
      }
    }
    // ---- dV^T += dO^T P, dK^T += Q^T dS (sums over the tile's 32 queries: two 16-deep slots)
#pragma unroll
    for (int u = 0; u < ((A16_ABLATE & 4) ? 0 : 2); ++u) {
      const bf16x8 pb = pack8(s, u), db = pack8(dp, u);
      dv0 = mfmab(tr_frag(dOtr, TSTR, u, 0, lane), pb, dv0);
      dv1 = mfmab(tr_frag(dOtr, TSTR, u, 32, lane), pb, dv1);
      dk0 = mfmab(tr_frag(Qtr, TSTR, u, 0, lane), db, dk0);
      dk1 = mfmab(tr_frag(Qtr, TSTR, u, 32, lane), db, dk1);
    }
    commit((t + 1) & 1, slot);   // (past the last tile: zeros nobody reads)
    prefetch(t + 3, slot);
    __syncthreads();   // this tile's dS rows are in LDS, the next tile's images and statistics in place
  };
  const int ntile2 = (ntile + 1) & ~1;   // an even count: a tile past the sequence has q = 0, dO = 0 (dS = 0, nothing added)
  for (int t = 0; t < ntile2; t += 2) {
    tile_step(t, 1);
    tile_step(t + 1, 0);
  }
  dq_tile(ntile2 - 1);
  if (kvalid) {
    __bf16* kp = p.dk + (int64_t)b * p.dks.sb + (int64_t)key * p.dks.st + (int64_t)h * p.dks.sh + 4 * hf;
    __bf16* vp = p.dv + (int64_t)b * p.dvs.sb + (int64_t)key * p.dvs.st + (int64_t)h * p.dvs.sh + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 a, c, d, e4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = (__bf16)(dk0[4 * g + e] * p.scale); c[e] = (__bf16)(dk1[4 * g + e] * p.scale);
        d[e] = (__bf16)dv0[4 * g + e]; e4[e] = (__bf16)dv1[4 * g + e];
      }
      *reinterpret_cast<bf16x4*>(kp + 8 * g) = a;
      *reinterpret_cast<bf16x4*>(kp + 32 + 8 * g) = c;
      *reinterpret_cast<bf16x4*>(vp + 8 * g) = d;
      *reinterpret_cast<bf16x4*>(vp + 32 + 8 * g) = e4;
    }
  }
}

// dq[b, i, h, :] = sum over key blocks (in order) of the f32 partials, rounded to bf16 once
__global__ __launch_bounds__(256) void attn_bf16_dq_reduce_kernel(Params p) {
  const int64_t n4 = (int64_t)p.B * p.I * p.H * 16;   // float4 pieces of one partial
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= n4) return;
  const float4* src = reinterpret_cast<const float4*>(p.dq_part) + i4;
  float4 s = src[0];
  for (int kb = 1; kb < p.nkblk; ++kb) {
    const float4 v = src[kb * n4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int d4 = (int)(i4 & 15);
  const int64_t r = i4 >> 4;            // (b, i, h)
  const int h = (int)(r % p.H);
  const int64_t bi = r / p.H;
  const int i = (int)(bi % p.I), b = (int)(bi / p.I);
  bf16x4 w;
  w[0] = (__bf16)s.x; w[1] = (__bf16)s.y; w[2] = (__bf16)s.z; w[3] = (__bf16)s.w;
  *reinterpret_cast<bf16x4*>(p.dq + (int64_t)b * p.dqs.sb + (int64_t)i * p.dqs.st + (int64_t)h * p.dqs.sh + 4 * d4) = w;
}

}  // namespace amk_attn_bf16

using namespace amk_attn_bf16;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool st_ok(const Strides& s) { return s.sb % 8 == 0 && s.st % 8 == 0 && s.sh % 8 == 0; }

extern "C" int amk_attn_bf16_fwd(const void* q, const void* k, const void* v, void* o, float* stats,
                                 const uint8_t* key_mask, const uint8_t* causal_mask,
                                 int B, int H, int I, int J, int Dh,
                                 int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                                 int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                                 float scale, void* stream) {
  AMK_CHECK_ARG(q && k && v && o && stats, "amk_attn_bf16_fwd: null tensor pointer");
  AMK_CHECK_ARG(B > 0 && H > 0 && I > 0 && J > 0, "amk_attn_bf16_fwd: non-positive size B=%d H=%d I=%d J=%d", B, H, I, J);
  AMK_CHECK_SUPPORTED(Dh == 64, "amk_attn_bf16_fwd: head dim %d not supported (64)", Dh);
  Params p = {};
  p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.out = (__bf16*)o; p.stats = stats;
  p.key_mask = key_mask; p.causal_mask = causal_mask;
  p.B = B; p.H = H; p.I = I; p.J = J;
  p.qs = {q_sb, q_st, q_sh}; p.ks = {k_sb, k_st, k_sh}; p.vs = {v_sb, v_st, v_sh}; p.os = {o_sb, o_st, o_sh};
  p.scale = scale; p.pinf = INFINITY;
  p.nqblk = (I + 127) / 128;
  AMK_CHECK_ARG(al16(q) && al16(k) && al16(v) && al16(o) && st_ok(p.qs) && st_ok(p.ks) && st_ok(p.vs) && st_ok(p.os),
                "amk_attn_bf16_fwd: pointers must be 16-byte aligned and strides multiples of 8 elements");
  const int64_t nwg = (int64_t)B * H * p.nqblk;
  AMK_CHECK_SUPPORTED(nwg < (1ll << 31), "amk_attn_bf16_fwd: grid too large");
  AMK_CHECK_SUPPORTED(((int64_t)J + 64) * k_st * 2 < (1ll << 31) && ((int64_t)J + 64) * v_st * 2 < (1ll << 31),
                      "amk_attn_bf16_fwd: one (batch, head) K/V slab must span < 2 GiB");
  if (key_mask || causal_mask)
    hipLaunchKernelGGL(attn_bf16_fwd_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(attn_bf16_fwd_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  AMK_CHECK_LAUNCH("amk_attn_bf16_fwd");
  return AMK_OK;
}

extern "C" int64_t amk_attn_bf16_bwd_ws_floats(int B, int H, int I, int J) {
  if (B <= 0 || H <= 0 || I <= 0 || J <= 0) return 0;
  const int64_t nkblk = (J + BKEYS - 1) / BKEYS;
  return 4 * (int64_t)B * H * I + nkblk * B * I * H * 64;   // four row constants per query + dQ partials
}

extern "C" int amk_attn_bf16_bwd(const void* q, const void* k, const void* v, const void* o, const float* stats, const void* d_o,
                                 void* dq, void* dk, void* dv, float* ws,
                                 const uint8_t* key_mask, const uint8_t* causal_mask,
                                 int B, int H, int I, int J, int Dh,
                                 int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                                 int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                                 int64_t do_sb, int64_t do_st, int64_t do_sh, int64_t dq_sb, int64_t dq_st, int64_t dq_sh,
                                 int64_t dk_sb, int64_t dk_st, int64_t dk_sh, int64_t dv_sb, int64_t dv_st, int64_t dv_sh,
                                 float scale, void* stream) {
  AMK_CHECK_ARG(q && k && v && o && stats && d_o && dq && dk && dv && ws, "amk_attn_bf16_bwd: null tensor pointer");
  AMK_CHECK_ARG(B > 0 && H > 0 && I > 0 && J > 0, "amk_attn_bf16_bwd: non-positive size B=%d H=%d I=%d J=%d", B, H, I, J);
  AMK_CHECK_SUPPORTED(Dh == 64, "amk_attn_bf16_bwd: head dim %d not supported (64)", Dh);
  Params p = {};
  p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.o = (const __bf16*)o; p.d_o = (const __bf16*)d_o;
  p.dq = (__bf16*)dq; p.dk = (__bf16*)dk; p.dv = (__bf16*)dv;
  p.stats = const_cast<float*>(stats);
  p.key_mask = key_mask; p.causal_mask = causal_mask;
  p.B = B; p.H = H; p.I = I; p.J = J;
  p.qs = {q_sb, q_st, q_sh}; p.ks = {k_sb, k_st, k_sh}; p.vs = {v_sb, v_st, v_sh}; p.os = {o_sb, o_st, o_sh};
  p.dos = {do_sb, do_st, do_sh}; p.dqs = {dq_sb, dq_st, dq_sh}; p.dks = {dk_sb, dk_st, dk_sh}; p.dvs = {dv_sb, dv_st, dv_sh};
  p.scale = scale; p.pinf = INFINITY;
  p.nkblk = (J + BKEYS - 1) / BKEYS;
  p.delta = ws;
  p.dq_part = ws + 4 * (int64_t)B * H * I;
  AMK_CHECK_ARG(al16(q) && al16(k) && al16(v) && al16(o) && al16(d_o) && al16(dq) && al16(dk) && al16(dv) && al16(p.dq_part) &&
                    st_ok(p.qs) && st_ok(p.ks) && st_ok(p.vs) && st_ok(p.os) && st_ok(p.dos) && st_ok(p.dqs) && st_ok(p.dks) && st_ok(p.dvs),
                "amk_attn_bf16_bwd: pointers must be 16-byte aligned and strides multiples of 8 elements");
  AMK_CHECK_SUPPORTED(((int64_t)J + BKEYS) * k_st * 2 < (1ll << 31) && ((int64_t)J + BKEYS) * v_st * 2 < (1ll << 31) &&
                          ((int64_t)I + 32) * q_st * 2 < (1ll << 31) && ((int64_t)I + 32) * do_st * 2 < (1ll << 31),
                      "amk_attn_bf16_bwd: one (batch, head) slab must span < 2 GiB");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * H * I;
  hipLaunchKernelGGL(attn_bf16_delta_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, p);
  const int64_t nwg = (int64_t)B * H * p.nkblk;
  AMK_CHECK_SUPPORTED(nwg < (1ll << 31), "amk_attn_bf16_bwd: grid too large");
  constexpr size_t lds = (size_t)(BKEYS * TSTR + 2 * BKEYS * DSTR + 4 * (32 * KSTR + 32 * TSTR)) * 2 + 2 * 128 * 4;
  static bool attr_set[64] = {};   // per device ordinal (the opt-in to > 64 KiB of dynamic LDS is per device)
  int dev_id = 0;
  AMK_CHECK_ARG(hipGetDevice(&dev_id) == hipSuccess && dev_id >= 0 && dev_id < 64, "amk_attn_bf16_bwd: no current device");
  if (!attr_set[dev_id]) {
    AMK_CHECK_SUPPORTED(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf16_bwd_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf16_bwd_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf16_bwd_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                        "amk_attn_bf16_bwd: the device refused %zu bytes of dynamic LDS", lds);
    attr_set[dev_id] = true;
  }
  if (causal_mask) hipLaunchKernelGGL((attn_bf16_bwd_kernel<true, true>), dim3((unsigned)nwg), dim3(64 * BW), lds, st, p);
  else if (key_mask) hipLaunchKernelGGL((attn_bf16_bwd_kernel<true, false>), dim3((unsigned)nwg), dim3(64 * BW), lds, st, p);
  else hipLaunchKernelGGL((attn_bf16_bwd_kernel<false, false>), dim3((unsigned)nwg), dim3(64 * BW), lds, st, p);
  const int64_t n4 = (int64_t)B * I * H * 16;
  hipLaunchKernelGGL(attn_bf16_dq_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p);
  AMK_CHECK_LAUNCH("amk_attn_bf16_bwd");
  return AMK_OK;
}
