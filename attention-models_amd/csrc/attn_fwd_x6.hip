// Fused softmax-attention forward with SPLIT-BF16 products ("bf16x6") for gfx950.
//
// Same algorithm, masks, statistics and memory layout as attn_fwd.hip; only the two contractions
// are fed differently.  Every f32 operand x is split into three bf16 parts x = h + m + l
// (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 mantissa bits) and a product a*b is
// formed as the six partial products  a_m b_m, a_l b_h, a_h b_l, a_m b_h, a_h b_m, a_h b_h,
// each exact in f32, accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Measured
// (tools/ubench_bf16x6.hip): the error of a K = 64 dot product against double precision is
// 1.5e-6, the exact-f32 v_mfma_f32_32x32x2_f32 gives 2.6e-6 -- it is not a reduced-precision
// path -- and six of these MFMAs do a 32x32x16 block in 192 matrix cycles where the f32 MFMA
// needs 512 (399 against 155 algorithmic TFLOP/s).  The price is VALU work to split the operands
// (about 5.5 instructions per element, and the bf16 MFMA hides only about two VALU instructions
// per MFMA): so K and V are split ONCE per call by a pre-pass (attn_split_kv_kernel, three bf16
// planes per tile; V transposed and in accumulator key order, so that its A-operand fragments are
// single 16-byte reads) instead of once per query block, Q once per wave, and only P -- which
// exists nowhere else -- inside the tile loop, out of the S accumulators.
//
//   S^T (keys x queries) = K Q^T : A = K rows (LDS, 8 consecutive d per lane), B = Q^T (registers)
//   O^T (dims x queries) += V^T P^T : A = V^T rows (LDS, 8 keys per lane in accumulator order),
//                                     B = P^T out of the S^T accumulators (registers 8t..8t+7 of a
//                                     32-key block are the 8 keys of one k-slot)
// The 32x32x16 bf16 MFMA has the accumulator layout of the 32x32x2 f32 one, so fills, the online
// softmax, the rescale and the epilogue are those of attn_fwd.hip.
#include "attn_common.h"

namespace amk_attn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int PSTR = 72;                 // bf16 per LDS row (64 + 8): 144 B, conflict-free 16-byte row reads
constexpr int PLANE = TILE * PSTR;       // one plane of a 64-row tile

__device__ __forceinline__ f32x16 mfmab(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// the six partial products, smallest first
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
  c = mfmab(a[1], b[1], c);
  c = mfmab(a[2], b[0], c);
  c = mfmab(a[0], b[2], c);
  c = mfmab(a[1], b[0], c);
  c = mfmab(a[0], b[1], c);
  c = mfmab(a[0], b[0], c);
  return c;
}
// two independent accumulators fed by the same B planes, MFMAs alternating so that no MFMA waits on its predecessor
__device__ __forceinline__ void mfma6x2(const bf16x8 (&a0)[3], const bf16x8 (&a1)[3], const bf16x8 (&b)[3], f32x16& c0, f32x16& c1) {
  c0 = mfmab(a0[1], b[1], c0); c1 = mfmab(a1[1], b[1], c1);
  c0 = mfmab(a0[2], b[0], c0); c1 = mfmab(a1[2], b[0], c1);
  c0 = mfmab(a0[0], b[2], c0); c1 = mfmab(a1[0], b[2], c1);
  c0 = mfmab(a0[1], b[0], c0); c1 = mfmab(a1[1], b[0], c1);
  c0 = mfmab(a0[0], b[1], c0); c1 = mfmab(a1[0], b[1], c1);
  c0 = mfmab(a0[0], b[0], c0); c1 = mfmab(a1[0], b[0], c1);
}
__device__ __forceinline__ void split1(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&pl)[3]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 h, m, l;
    split1(x[j], h, m, l);
    pl[0][j] = h; pl[1][j] = m; pl[2][j] = l;
  }
}
__device__ __forceinline__ void split4(float a, float b, float c, float d, bf16x4 (&pl)[3]) {
  const float x[4] = {a, b, c, d};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 h, m, l;
    split1(x[j], h, m, l);
    pl[0][j] = h; pl[1][j] = m; pl[2][j] = l;
  }
}

template <bool CAUSAL>
__global__ __launch_bounds__(WG, 2) void attn_fwd_x6_kernel(FwdParams p) {
  __shared__ __attribute__((aligned(16))) __bf16 Kp[3 * PLANE];   // [plane][key][d]
  __shared__ __attribute__((aligned(16))) __bf16 Vt[3 * PLANE];   // [plane][d][key position]
  __shared__ __attribute__((aligned(16))) float Kfill[TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wg % p.nblk;
  const int bh = wg / p.nblk;
  const int h = bh % p.H, b = bh / p.H;

  const int qi = qb * BLK + wave * 32 + ln;  // this lane's query row
  const bool qvalid = qi < p.I;

  // Q^T operand: lane (query, half) holds, for each of the four 16-deep k-blocks c, the three planes of
  // (q * scale * log2 e)[query][16c + 8*half + j], j = 0..7
  bf16x8 qpl[4][3];
  {
    const float qscale = p.scale * AMK_LOG2E;
    const float* qp = p.q + (int64_t)b * p.qs.sb + (int64_t)qi * p.qs.st + (int64_t)h * p.qs.sh + 8 * hf;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 t0 = qvalid ? ld4(qp + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 t1 = qvalid ? ld4(qp + 16 * c + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float x[8] = {t0.x * qscale, t0.y * qscale, t0.z * qscale, t0.w * qscale,
                          t1.x * qscale, t1.y * qscale, t1.z * qscale, t1.w * qscale};
      split8(x, qpl[c]);
    }
  }

  const uint8_t* kmask = p.key_mask ? p.key_mask + (int64_t)b * p.J : nullptr;
  const uint8_t* cmrow = CAUSAL ? p.causal_mask + (int64_t)qi * p.J : nullptr;

  // staging: the pre-pass left, per (batch, head, 64-key tile), 24 KB of K planes [key][plane][64 d] and
  // 24 KB of V^T planes [dim][plane][64 key positions], contiguous: 1536 16-byte chunks each, six per thread.
  const int ntile = (p.J + TILE - 1) / TILE;
  // one buffer descriptor over this (batch, head)'s tiles; chunk c of tile t is at byte (t*3072 + c)*16
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.x6_ws) + (int64_t)bh * ntile * 2 * 24576, 0, ntile * 2 * 24576, 0x00020000);
  float4 kch[6], vch[6];
  int lds_off[6];   // bf16 index inside a 3-plane LDS tile for chunk tid + 256*n
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    const int c = tid + 256 * n;
    const int row = c / 24, plane = (c % 24) >> 3, k8 = c & 7;
    lds_off[n] = plane * PLANE + row * PSTR + 8 * k8;
  }
  float fillst = 0.f;
  auto prefetch = [&](int t) {
    const int base = t * 2 * 24576 + tid * 16;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      kch[n] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, base + 4096 * n, 0, 0));
      vch[n] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, base + 24576 + 4096 * n, 0, 0));
    }
    if (tid < TILE) {
      const int j = t * TILE + tid;
      float f = 0.f;
      if (j >= p.J) f = -INFINITY;                       // beyond the sequence: weight 0
      else if (kmask && kmask[j] == 0) f = AMK_FILL_MASKED;  // masked_fill(~context_mask, -1e9)
      fillst = f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      *reinterpret_cast<float4*>(&Kp[lds_off[n]]) = kch[n];
      *reinterpret_cast<float4*>(&Vt[lds_off[n]]) = vch[n];
    }
    if (tid < TILE) Kfill[tid] = fillst;
  };

  f32x16 o0 = zero16(), o1 = zero16();
  float m_run = -INFINITY, l_run = 0.f;

  prefetch(0);
  for (int t = 0; t < ntile; ++t) {
    const int j0 = t * TILE;
    __syncthreads();  // every wave is done reading the previous tile
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch(t + 1);

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

    // ---- S^T = K Q^T for the two 32-key halves of the tile: 2 x 4 k-blocks x 6 MFMAs ----
    f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 ka[3], kb2[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        ka[q] = *reinterpret_cast<const bf16x8*>(&Kp[q * PLANE + ln * PSTR + 16 * c + 8 * hf]);
        kb2[q] = *reinterpret_cast<const bf16x8*>(&Kp[q * PLANE + (32 + ln) * PSTR + 16 * c + 8 * hf]);
      }
      mfma6x2(ka, kb2, qpl[c], s0, s1);
    }

    // ---- fills and online softmax: as attn_fwd.hip ----
    const bool plain = !CAUSAL && kmask == nullptr && (j0 + TILE <= p.J);  // wave-uniform
    float mx;
    if (plain) {
      mx = vmax(s0[0], s1[0], p.pinf);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = vmax(mx, vmax(s0[r], s1[r], p.pinf), p.pinf);
    } else {
      mx = -INFINITY;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 f0 = ld4(&Kfill[8 * g + 4 * hf]);
        const float4 f1 = ld4(&Kfill[32 + 8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float t0 = s0[r], t1 = s1[r];
          const float fa = f4(f0, e), fb = f4(f1, e);
          t0 = (fa == 0.f) ? t0 : fa;
          t1 = (fb == 0.f) ? t1 : fb;
          if (CAUSAL) {
            t0 = ((cbits0 >> r) & 1u) ? AMK_FILL_MASKED : t0;
            t1 = ((cbits1 >> r) & 1u) ? AMK_FILL_MASKED : t1;
          }
          s0[r] = t0;
          s1[r] = t1;
          mx = vmax(mx, vmax(t0, t1, p.pinf), p.pinf);
        }
      }
    }
    mx = vmax(mx, __shfl_xor(mx, 32, 64), p.pinf);
    const float m_new = vmax(m_run, mx, p.pinf);
    f32x2 lsum2 = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p0 = __builtin_amdgcn_exp2f(s0[r] - m_new);
      const float p1 = __builtin_amdgcn_exp2f(s1[r] - m_new);
      s0[r] = p0;
      s1[r] = p1;
      lsum2 += (f32x2){p0, p1};
    }
    const float lsum = lsum2.x + lsum2.y;
    if (__any(m_new != m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
      m_run = m_new;
    }
    l_run += lsum;

    // ---- O^T += V^T P^T: 4 groups of 16 keys x 2 dim blocks x 6 MFMAs ----
    // group g = 2*blk + t: registers 8t..8t+7 of S^T block blk are keys 32*blk + 16*t + 4*half + (j&3) + 8*(j>>2)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (g < 2) ? s0[8 * (g & 1) + j] : s1[8 * (g & 1) + j];
      bf16x8 pp[3];
      split8(x, pp);
      bf16x8 va[3], vb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        va[q] = *reinterpret_cast<const bf16x8*>(&Vt[q * PLANE + ln * PSTR + 16 * g + 8 * hf]);
        vb[q] = *reinterpret_cast<const bf16x8*>(&Vt[q * PLANE + (32 + ln) * PSTR + 16 * g + 8 * hf]);
      }
      mfma6x2(va, vb, pp, o0, o1);
    }
  }

  // ---- epilogue: normalise, store O rows and the softmax statistics ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (qvalid) {
    float* op = p.o + (int64_t)b * p.os.sb + (int64_t)qi * p.os.st + (int64_t)h * p.os.sh + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      st4(op + 8 * g, make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
      st4(op + 32 + 8 * g, make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
    }
    if (hf == 0) {
      float* sp = p.stats + (((int64_t)b * p.H + h) * p.I + qi) * 2;
      sp[0] = m_run;
      sp[1] = l_tot;
    }
  }
}

// Pre-pass: one workgroup per (batch, head, 64-key tile) splits the K and V rows of the tile into bf16 planes.
//   K tile  -> [key 0..63][plane 0..2][64 d]
//   V tile  -> [dim 0..63][plane 0..2][64 key positions]; inside each group of 16 keys the positions are in
//              accumulator order [0..3, 8..11, 4..7, 12..15] (the k-slots of a P^T fragment)
// Rows beyond the sequence are written as zeros.
__global__ __launch_bounds__(WG) void attn_split_kv_kernel(FwdParams p) {
  const int tid = threadIdx.x;
  const int ntile = (p.J + TILE - 1) / TILE;
  const int t = blockIdx.x % ntile;
  const int bh = blockIdx.x / ntile;
  const int h = bh % p.H, b = bh / p.H;
  const float* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;
  __bf16* out = reinterpret_cast<__bf16*>(p.x6_ws) + ((int64_t)bh * ntile + t) * 2 * 1536 * 8;
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  RowStager kload, vload;
  kload.init(kbase, p.ks.st, p.J, tid);
  vload.init(vbase, p.vs.st, p.J, tid);
  kload.seek(t, p.ks.st, tid);
  vload.seek(t, p.vs.st, tid);
#pragma unroll
  for (int i = 0; i < 4; ++i) vload.voff[i] = (int)(((int64_t)(4 * srow + i) * p.vs.st + scol) * 4) + t * vload.step;
  float4 kst[4], vst[4];
  kload.load(kst);
  vload.load(vst);
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {   // K: row srow + 16*ps, dims scol..scol+3
    bf16x4 pl[3];
    split4(kst[ps].x, kst[ps].y, kst[ps].z, kst[ps].w, pl);
    const int r = srow + 16 * ps;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(out + (r * 3 + q) * 64 + scol) = pl[q];
  }
  // V: this thread holds keys 4*srow .. 4*srow+3 x dims scol..scol+3; position of the key quad in its 16-group
  const int vpos = 16 * (srow >> 2) + 4 * (((srow & 1) << 1) | ((srow >> 1) & 1));
  __bf16* vout = out + 1536 * 8;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    bf16x4 pl[3];
    split4(f4(vst[0], dd), f4(vst[1], dd), f4(vst[2], dd), f4(vst[3], dd), pl);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(vout + ((scol + dd) * 3 + q) * 64 + vpos) = pl[q];
  }
}

void launch_attn_fwd_x6(const FwdParams& p, int64_t nwg, hipStream_t st) {
  const int ntile = (p.J + TILE - 1) / TILE;
  hipLaunchKernelGGL(attn_split_kv_kernel, dim3((unsigned)((int64_t)p.B * p.H * ntile)), dim3(WG), 0, st, p);
  if (p.causal_mask)
    hipLaunchKernelGGL(attn_fwd_x6_kernel<true>, dim3((unsigned)nwg), dim3(WG), 0, st, p);
  else
    hipLaunchKernelGGL(attn_fwd_x6_kernel<false>, dim3((unsigned)nwg), dim3(WG), 0, st, p);
}

}  // namespace amk_attn
