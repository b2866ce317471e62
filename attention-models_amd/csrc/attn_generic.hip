// Softmax attention for head dims other than 64 (the reference takes any dim_head,
// models/softmax_attention.py:23): the kernels of attn_fwd.hip / attn_bwd.hip with the head dim as
// a template parameter, D = 32 or 128.  Same arithmetic, layouts, masks, statistics and MFMA
// structure (exact-f32 v_mfma_f32_32x32x2_f32, the reduction axis on the lane, accumulators reused
// as the next product's B operand); the D = 64 instantiations of those files stay the tuned
// ones (software-pipelined operand reads, kept scores, the one-pass backward) -- these are the
// plain form: forward, and the two reproducible recompute kernels for the backward.
#include "attn_common.h"
#include <type_traits>

namespace amk_attn {

template <int DH>
struct GenGeom {
  static constexpr int HD = DH / 2;                 // k-extent owned by one half-wave
  static constexpr int LS = DH + 4;                 // LDS row stride: conflict-free b128 row reads
  static constexpr int NT = DH / 32;                // 32-wide output tiles
  static constexpr int TL = DH <= 64 ? 64 : 32;     // rows of the streamed operand per LDS tile
  static constexpr int NS = TL / 32;                // 32-row sub-tiles per LDS tile
  static constexpr int F4R = DH / 4;                // float4 per row
  static constexpr int RP = 256 / F4R;              // rows per staging pass of 256 threads
  static constexpr int NP = TL / RP;                // passes per tile
};

// Streams TL rows x DH floats global -> registers through a range-checked buffer descriptor.
template <int DH>
struct GenStager {
  using G = GenGeom<DH>;
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[G::NP];
  int step;
  __device__ __forceinline__ void init(const float* base, int64_t row_stride, int nrows, int tid) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((int64_t)(nrows - 1) * row_stride + DH) * 4), 0x00020000);
    const int srow = tid / G::F4R, scol = (tid % G::F4R) * 4;
#pragma unroll
    for (int ps = 0; ps < G::NP; ++ps) voff[ps] = (int)(((int64_t)(srow + G::RP * ps) * row_stride + scol) * 4);
    step = (int)(G::TL * row_stride * 4);
  }
  __device__ __forceinline__ void load(float4 (&dst)[G::NP]) {
#pragma unroll
    for (int ps = 0; ps < G::NP; ++ps) {
      dst[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[ps], 0, 0));
      voff[ps] += step;
    }
  }
  // LDS float offset of pass ps of this thread inside a [TL][LS] tile
  static __device__ __forceinline__ int lds_off(int tid, int ps) {
    return (tid / G::F4R + G::RP * ps) * G::LS + (tid % G::F4R) * 4;
  }
};

// ------------------------------------------------------------------------------------------------
template <int DH, bool CAUSAL>
__global__ __launch_bounds__(WG, (DH <= 32 ? 2 : 1)) void attn_fwd_gen_kernel(FwdParams p) {
  using G = GenGeom<DH>;
  constexpr int HD = G::HD, LS = G::LS, NT = G::NT, TL = G::TL, NS = G::NS;
  __shared__ __attribute__((aligned(16))) float smem[2 * TL * LS + TL];
  float* Ks = smem;
  float* Vs = smem + TL * LS;
  float* Kfill = smem + 2 * TL * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;
  const int wg = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int qb = __builtin_amdgcn_readfirstlane(wg % p.nblk);
  const int bh = __builtin_amdgcn_readfirstlane(wg / p.nblk);
  const int h = __builtin_amdgcn_readfirstlane(bh % p.H), b = __builtin_amdgcn_readfirstlane(bh / p.H);
  const int qi = qb * BLK + wave * 32 + ln;
  const bool qvalid = qi < p.I;

  const float qscale = p.scale * AMK_LOG2E;
  float qreg[HD];
  {
    const float* qp = p.q + (int64_t)b * p.qs.sb + (int64_t)qi * p.qs.st + (int64_t)h * p.qs.sh + HD * hf;
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
      const float4 t = qvalid ? ld4(qp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      qreg[4 * s4 + 0] = t.x * qscale; qreg[4 * s4 + 1] = t.y * qscale;
      qreg[4 * s4 + 2] = t.z * qscale; qreg[4 * s4 + 3] = t.w * qscale;
    }
  }
  const float* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;
  const uint8_t* kmask = p.key_mask ? p.key_mask + (int64_t)b * p.J : nullptr;
  const uint8_t* cmrow = CAUSAL ? p.causal_mask + (int64_t)qi * p.J : nullptr;

  float4 kst[G::NP], vst[G::NP];
  float fillst = 0.f;
  GenStager<DH> kload, vload;
  kload.init(kbase, p.ks.st, p.J, tid);
  vload.init(vbase, p.vs.st, p.J, tid);
  auto prefetch = [&](int j0) {
    kload.load(kst);
    vload.load(vst);
    if (tid < TL) {
      const int j = j0 + tid;
      float f = 0.f;
      if (j >= p.J) f = -INFINITY;
      else if (kmask && kmask[j] == 0) f = AMK_FILL_MASKED;
      fillst = f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int ps = 0; ps < G::NP; ++ps) {
      st4(&Ks[GenStager<DH>::lds_off(tid, ps)], kst[ps]);
      st4(&Vs[GenStager<DH>::lds_off(tid, ps)], vst[ps]);
    }
    if (tid < TL) Kfill[tid] = fillst;
  };

  f32x16 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = zero16();
  float m_run = -INFINITY, l_run = 0.f;

  const int ntile = (p.J + TL - 1) / TL;
  prefetch(0);
  for (int t = 0; t < ntile; ++t) {
    const int j0 = t * TL;
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch(j0 + TL);

    // S^T for the NS sub-tiles of 32 keys, fills, tile maximum
    f32x16 s[NS];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      s[u] = zero16();
      const float* kr = &Ks[(32 * u + ln) * LS + HD * hf];
#pragma unroll
      for (int s4 = 0; s4 < HD / 4; ++s4) {
        const float4 a = ld4(kr + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[u] = mfma32(f4(a, e), qreg[4 * s4 + e], s[u]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 f = ld4(&Kfill[32 * u + 8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float fe = f4(f, e);
          float tt = (fe == 0.f) ? s[u][r] : fe;
          if (CAUSAL) {
            const int j = j0 + 32 * u + acc_row(r, hf);
            if (qvalid && j < p.J && cmrow[j]) tt = AMK_FILL_MASKED;
          }
          s[u][r] = tt;
          mx = vmax(mx, tt, p.pinf);
        }
      }
    }
    mx = vmax(mx, __shfl_xor(mx, 32, 64), p.pinf);
    const float m_new = vmax(m_run, mx, p.pinf);
    float lsum = 0.f;
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(s[u][r] - m_new);
        s[u][r] = pr;
        lsum += pr;
      }
    if (__any(m_new != m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
      m_run = m_new;
    }
    l_run += lsum;
    // O^T += V^T P^T
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vc = &Vs[(32 * u + acc_row(r, hf)) * LS + ln];
#pragma unroll
        for (int n = 0; n < NT; ++n) o[n] = mfma32(vc[32 * n], s[u][r], o[n]);
      }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (qvalid) {
    float* op = p.o + (int64_t)b * p.os.sb + (int64_t)qi * p.os.st + (int64_t)h * p.os.sh + 4 * hf;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(op + 32 * n + 8 * g, make_float4(o[n][4 * g] * inv, o[n][4 * g + 1] * inv, o[n][4 * g + 2] * inv, o[n][4 * g + 3] * inv));
    if (hf == 0) {
      float* sp = p.stats + (((int64_t)b * p.H + h) * p.I + qi) * 2;
      sp[0] = m_run;
      sp[1] = l_tot;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// The forward WITHOUT masks for head dims 32 / 128 (round 4): attn_fwd_plain_kernel of attn_fwd.hip with the head dim as a
// template parameter -- lazy softmax reference held in the MFMA accumulators (-m_ref is the C operand that opens every
// S^T chain; the reference moves only when a row's tile maximum passes it by 2^8), v_max3 row maxima, packed row sums,
// no fill path (the ragged last tile is a peeled copy of the tile body), operand fragments read a step ahead.  (m_ref, l)
// leave as the statistics; the recompute backward forms exp2(S - m_ref) / l, the same P.
__device__ __forceinline__ float gmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// KEEP: the raw scores also leave for the one-pass backward (ScoreTiles, as amk_attn_fwd_keep for head dim 64); that form
// forms its S^T chains from zero and subtracts the reference afterwards (the backward needs the raw numbers).
template <int DH, bool KEEP>
__global__ __launch_bounds__(WG, (DH <= 32 ? 2 : 1)) void attn_fwd_gen_plain_kernel(FwdParams p) {
  using G = GenGeom<DH>;
  constexpr int HD = G::HD, LS = G::LS, NT = G::NT, TL = G::TL, NS = G::NS;
  constexpr float TAU = 8.f;
  __shared__ __attribute__((aligned(16))) float smem[2 * TL * LS];
  float* Ks = smem;
  float* Vs = smem + TL * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;
  // (wave-uniform; said explicitly because this file is built without IEEE mode: see attn_fwd_plain_kernel)
  const int wg = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int qb = __builtin_amdgcn_readfirstlane(wg % p.nblk);
  const int bh = __builtin_amdgcn_readfirstlane(wg / p.nblk);
  const int h = __builtin_amdgcn_readfirstlane(bh % p.H), b = __builtin_amdgcn_readfirstlane(bh / p.H);
  const int qi = qb * BLK + wave * 32 + ln;
  const bool qvalid = qi < p.I;

  const float qscale = p.scale * AMK_LOG2E;
  float qreg[HD];
  {
    const float* qp = p.q + (int64_t)b * p.qs.sb + (int64_t)qi * p.qs.st + (int64_t)h * p.qs.sh + HD * hf;
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
      const float4 t = qvalid ? ld4(qp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      qreg[4 * s4 + 0] = t.x * qscale; qreg[4 * s4 + 1] = t.y * qscale;
      qreg[4 * s4 + 2] = t.z * qscale; qreg[4 * s4 + 3] = t.w * qscale;
    }
  }
  const float* kbase = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vbase = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;
  float4 kst[G::NP], vst[G::NP];
  GenStager<DH> kload, vload;
  kload.init(kbase, p.ks.st, p.J, tid);
  vload.init(vbase, p.vs.st, p.J, tid);
  auto prefetch = [&]() {
    kload.load(kst);
    vload.load(vst);
  };
  auto commit = [&]() {
#pragma unroll
    for (int ps = 0; ps < G::NP; ++ps) {
      st4(&Ks[GenStager<DH>::lds_off(tid, ps)], kst[ps]);
      st4(&Vs[GenStager<DH>::lds_off(tid, ps)], vst[ps]);
    }
  };

  f32x16 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = zero16();
  f32x16 negm = zero16();
  float mref = 0.f, l_run = 0.f;
  const int ntile = (p.J + TL - 1) / TL;
  const int nfull = p.J / TL;
  const ScoreTiles stl(p.I, p.J);
  const int64_t sc_kstep = (int64_t)stl.nqt * 1024;  // floats between consecutive 32-key blocks
  float* sc_ptr = nullptr;
  if (KEEP) sc_ptr = p.scores + ((int64_t)bh * stl.nkb * stl.nqt + (qb * NWAVE + wave)) * 1024 + 4 * hf * 32 + ln;

  auto tile = [&](const int t, auto ragged_c) {
    constexpr bool RAGGED = decltype(ragged_c)::value;
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch();

    f32x16 s[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      s[u] = KEEP ? zero16() : negm;
      const float* kr = &Ks[(32 * u + ln) * LS + HD * hf];
      float4 a = ld4(kr);
#pragma unroll
      for (int s4 = 0; s4 < HD / 4; ++s4) {
        float4 nx = a;
        if (s4 + 1 < HD / 4) nx = ld4(kr + 4 * (s4 + 1));   // the next k-block's fragment, one step ahead of its MFMAs
#pragma unroll
        for (int e = 0; e < 4; ++e) s[u] = mfma32(f4(a, e), qreg[4 * s4 + e], s[u]);
        a = nx;
      }
    }
    if (KEEP) {
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        float* t0 = sc_ptr + (int64_t)(NS * t + u) * sc_kstep;
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(s[u][r], t0 + acc_row(r, 0) * 32);
      }
    }
    if (RAGGED) {
#pragma unroll
      for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][r] = (t * TL + 32 * u + acc_row(r, hf) < p.J) ? s[u][r] : -INFINITY;
    }
    float mx = gmax3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int r = 3; r + 1 < 16; r += 2) mx = gmax3(mx, s[0][r], s[0][r + 1]);
    mx = __builtin_fmaxf(mx, s[0][15]);
#pragma unroll
    for (int u = 1; u < NS; ++u)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = gmax3(mx, s[u][r], s[u][r + 1]);
    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (KEEP) mx -= mref;
    if (t == 0 || __any(mx > TAU)) {   // rare after the first tile
      const float d = t == 0 ? mx : __builtin_fmaxf(mx, 0.f);
      if (t != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
      }
      mref += d;
      if (!KEEP) {
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[u][r] -= d;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -mref;
      }
    }
    f32x2 ls = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NS; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = __builtin_amdgcn_exp2f(KEEP ? s[u][r] - mref : s[u][r]);
#pragma unroll
      for (int r = 0; r < 16; r += 2) ls += (f32x2){s[u][r], s[u][r + 1]};
    }
    l_run += ls.x + ls.y;
    // O^T += V^T P^T, the V column fragments of step r + 1 read under the MFMAs of step r
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      float c[NT], nx[NT];
      {
        const float* vc = &Vs[(32 * u + acc_row(0, hf)) * LS + ln];
#pragma unroll
        for (int n = 0; n < NT; ++n) c[n] = vc[32 * n];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r + 1 < 16) {
          const float* vc = &Vs[(32 * u + acc_row(r + 1, hf)) * LS + ln];
#pragma unroll
          for (int n = 0; n < NT; ++n) nx[n] = vc[32 * n];
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) o[n] = mfma32(c[n], s[u][r], o[n]);
#pragma unroll
        for (int n = 0; n < NT; ++n) c[n] = nx[n];
      }
    }
  };

  prefetch();
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  if (nfull < ntile) tile(nfull, std::true_type{});

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (qvalid) {
    float* op = p.o + (int64_t)b * p.os.sb + (int64_t)qi * p.os.st + (int64_t)h * p.os.sh + 4 * hf;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(op + 32 * n + 8 * g, make_float4(o[n][4 * g] * inv, o[n][4 * g + 1] * inv, o[n][4 * g + 2] * inv, o[n][4 * g + 3] * inv));
    if (hf == 0) {
      float* sp = p.stats + (((int64_t)b * p.H + h) * p.I + qi) * 2;
      sp[0] = mref;
      sp[1] = l_tot;
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_delta_gen_kernel(BwdParams p) {
  // DH/4 lanes per (b,h,i) row
  constexpr int LPR = DH / 4;
  const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const int64_t nrow = (int64_t)p.B * p.H * p.I;
  const int c = (threadIdx.x % LPR) * 4;
  float acc = 0.f;
  if (row < nrow) {
    const int i = (int)(row % p.I);
    const int64_t bh = row / p.I;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const float4 a = ld4(p.o + (int64_t)b * p.os.sb + (int64_t)i * p.os.st + (int64_t)h * p.os.sh + c);
    const float4 g = ld4(p.d_o + (int64_t)b * p.dos.sb + (int64_t)i * p.dos.st + (int64_t)h * p.dos.sh + c);
    acc = a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
  }
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (row < nrow && (threadIdx.x % LPR) == 0) p.delta[row] = acc;
}

// ------------------------------------------------------------------------------------------------
// dQ: query on the lane (the forward's skeleton)
template <int DH, bool CAUSAL>
__global__ __launch_bounds__(WG, (DH <= 32 ? 2 : 1)) void attn_bwd_dq_gen_kernel(BwdParams p) {
  using G = GenGeom<DH>;
  constexpr int HD = G::HD, LS = G::LS, NT = G::NT, TL = G::TL, NS = G::NS;
  __shared__ __attribute__((aligned(16))) float smem[2 * TL * LS + TL];
  float* Ks = smem;
  float* Vs = smem + TL * LS;
  float* Kfill = smem + 2 * TL * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;
  const int wg = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int qb = __builtin_amdgcn_readfirstlane(wg % p.nqblk);
  const int bh = __builtin_amdgcn_readfirstlane(wg / p.nqblk);
  const int h = __builtin_amdgcn_readfirstlane(bh % p.H), b = __builtin_amdgcn_readfirstlane(bh / p.H);
  const int qi = qb * BLK + wave * 32 + ln;
  const bool qvalid = qi < p.I;

  const float qscale = p.scale * AMK_LOG2E;
  float qreg[HD], greg[HD];
  float m_q = INFINITY, linv_q = 0.f, delta_q = 0.f;
  {
    const float* qp = p.q + (int64_t)b * p.qs.sb + (int64_t)qi * p.qs.st + (int64_t)h * p.qs.sh + HD * hf;
    const float* gp = p.d_o + (int64_t)b * p.dos.sb + (int64_t)qi * p.dos.st + (int64_t)h * p.dos.sh + HD * hf;
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
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

  float4 kst[G::NP], vst[G::NP];
  float fillst = 0.f;
  GenStager<DH> kload, vload;
  kload.init(kbase, p.ks.st, p.J, tid);
  vload.init(vbase, p.vs.st, p.J, tid);
  auto prefetch = [&](int j0) {
    kload.load(kst);
    vload.load(vst);
    if (tid < TL) {
      const int j = j0 + tid;
      float f = 0.f;
      if (j >= p.J) f = -INFINITY;
      else if (kmask && kmask[j] == 0) f = AMK_FILL_MASKED;
      fillst = f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int ps = 0; ps < G::NP; ++ps) {
      st4(&Ks[GenStager<DH>::lds_off(tid, ps)], kst[ps]);
      st4(&Vs[GenStager<DH>::lds_off(tid, ps)], vst[ps]);
    }
    if (tid < TL) Kfill[tid] = fillst;
  };

  f32x16 dq[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) dq[n] = zero16();
  const int ntile = (p.J + TL - 1) / TL;
  prefetch(0);
  for (int t = 0; t < ntile; ++t) {
    const int j0 = t * TL;
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch(j0 + TL);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      f32x16 s = zero16(), dp = zero16();
      const float* kr = &Ks[(32 * u + ln) * LS + HD * hf];
      const float* vr = &Vs[(32 * u + ln) * LS + HD * hf];
#pragma unroll
      for (int s4 = 0; s4 < HD / 4; ++s4) {
        const float4 a = ld4(kr + 4 * s4);
        const float4 c = ld4(vr + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = mfma32(f4(a, e), qreg[4 * s4 + e], s);
          dp = mfma32(f4(c, e), greg[4 * s4 + e], dp);
        }
      }
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
            const int j = j0 + 32 * u + acc_row(r, hf);
            if (qvalid && j < p.J && cmrow[j]) { tt = AMK_FILL_MASKED; filled = true; }
          }
          const float pr = __builtin_amdgcn_exp2f(tt - m_q) * linv_q;
          s[r] = filled ? 0.f : pr * (dp[r] - delta_q);  // dS^T (no gradient through fills)
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* kc = &Ks[(32 * u + acc_row(r, hf)) * LS + ln];
#pragma unroll
        for (int n = 0; n < NT; ++n) dq[n] = mfma32(kc[32 * n], s[r], dq[n]);
      }
    }
  }
  if (qvalid) {
    float* dp_ = p.dq + (int64_t)b * p.dqs.sb + (int64_t)qi * p.dqs.st + (int64_t)h * p.dqs.sh + 4 * hf;
    const float sc = p.scale;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(dp_ + 32 * n + 8 * g, make_float4(dq[n][4 * g] * sc, dq[n][4 * g + 1] * sc, dq[n][4 * g + 2] * sc, dq[n][4 * g + 3] * sc));
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: key on the lane (a wave owns 32 keys; k, v in registers)
template <int DH, bool CAUSAL>
__global__ __launch_bounds__(WG, (DH <= 32 ? 2 : 1)) void attn_bwd_dkdv_gen_kernel(BwdParams p) {
  using G = GenGeom<DH>;
  constexpr int HD = G::HD, LS = G::LS, NT = G::NT, TL = G::TL, NS = G::NS;
  __shared__ __attribute__((aligned(16))) float smem[2 * TL * LS + 3 * TL];
  float* Qs = smem;
  float* Gs = smem + TL * LS;
  float* Ms = smem + 2 * TL * LS;
  float* Ls = Ms + TL;
  float* Ds = Ls + TL;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;
  const int wg = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int kb = __builtin_amdgcn_readfirstlane(wg % p.nkblk);
  const int bh = __builtin_amdgcn_readfirstlane(wg / p.nkblk);
  const int h = __builtin_amdgcn_readfirstlane(bh % p.H), b = __builtin_amdgcn_readfirstlane(bh / p.H);
  const int kj = kb * BLK + wave * 32 + ln;
  const bool kvalid = kj < p.J;

  float kreg[HD], vreg[HD];
  {
    const float* kp = p.k + (int64_t)b * p.ks.sb + (int64_t)kj * p.ks.st + (int64_t)h * p.ks.sh + HD * hf;
    const float* vp = p.v + (int64_t)b * p.vs.sb + (int64_t)kj * p.vs.st + (int64_t)h * p.vs.sh + HD * hf;
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
      const float4 a = kvalid ? ld4(kp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 c = kvalid ? ld4(vp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      kreg[4 * s4 + 0] = a.x; kreg[4 * s4 + 1] = a.y; kreg[4 * s4 + 2] = a.z; kreg[4 * s4 + 3] = a.w;
      vreg[4 * s4 + 0] = c.x; vreg[4 * s4 + 1] = c.y; vreg[4 * s4 + 2] = c.z; vreg[4 * s4 + 3] = c.w;
    }
  }
  float kfill = 0.f;
  if (!kvalid) kfill = -INFINITY;
  else if (p.key_mask && p.key_mask[(int64_t)b * p.J + kj] == 0) kfill = AMK_FILL_MASKED;

  const float* qbase = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* gbase = p.d_o + (int64_t)b * p.dos.sb + (int64_t)h * p.dos.sh;
  const float* stbase = p.stats + ((int64_t)b * p.H + h) * p.I * 2;
  const float* dlbase = p.delta + ((int64_t)b * p.H + h) * p.I;

  float4 qst[G::NP], gst[G::NP];
  float mst = 0.f, lst = 0.f, dst = 0.f;
  GenStager<DH> qload, gload;
  qload.init(qbase, p.qs.st, p.I, tid);
  gload.init(gbase, p.dos.st, p.I, tid);
  auto prefetch = [&](int i0) {
    qload.load(qst);
    gload.load(gst);
    if (tid < TL) {
      const int i = i0 + tid;
      if (i < p.I) {
        mst = stbase[2 * i];
        lst = 1.f / stbase[2 * i + 1];
        dst = dlbase[i];
      } else {
        mst = INFINITY; lst = 0.f; dst = 0.f;
      }
    }
  };
  auto commit = [&]() {
    const float sc = p.scale * AMK_LOG2E;
#pragma unroll
    for (int ps = 0; ps < G::NP; ++ps) {
      st4(&Qs[GenStager<DH>::lds_off(tid, ps)], make_float4(qst[ps].x * sc, qst[ps].y * sc, qst[ps].z * sc, qst[ps].w * sc));
      st4(&Gs[GenStager<DH>::lds_off(tid, ps)], gst[ps]);
    }
    if (tid < TL) { Ms[tid] = mst; Ls[tid] = lst; Ds[tid] = dst; }
  };

  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { dk[n] = zero16(); dv[n] = zero16(); }
  const uint8_t* cmcol = CAUSAL ? p.causal_mask + kj : nullptr;

  const int ntile = (p.I + TL - 1) / TL;
  prefetch(0);
  for (int t = 0; t < ntile; ++t) {
    const int i0 = t * TL;
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch(i0 + TL);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      f32x16 s = zero16(), dp = zero16();
      const float* qr = &Qs[(32 * u + ln) * LS + HD * hf];
      const float* gr = &Gs[(32 * u + ln) * LS + HD * hf];
#pragma unroll
      for (int s4 = 0; s4 < HD / 4; ++s4) {
        const float4 a = ld4(qr + 4 * s4);
        const float4 c = ld4(gr + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = mfma32(f4(a, e), kreg[4 * s4 + e], s);
          dp = mfma32(f4(c, e), vreg[4 * s4 + e], dp);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 m4 = ld4(&Ms[32 * u + 8 * g + 4 * hf]);
        const float4 l4 = ld4(&Ls[32 * u + 8 * g + 4 * hf]);
        const float4 d4 = ld4(&Ds[32 * u + 8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          bool filled = kfill != 0.f;
          float tt = filled ? kfill : s[r];
          if (CAUSAL) {
            const int i = i0 + 32 * u + acc_row(r, hf);
            if (kvalid && i < p.I && cmcol[(int64_t)i * p.J]) { tt = AMK_FILL_MASKED; filled = true; }
          }
          const float pr = __builtin_amdgcn_exp2f(tt - f4(m4, e)) * f4(l4, e);
          s[r] = pr;
          dp[r] = filled ? 0.f : pr * (dp[r] - f4(d4, e));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* gc = &Gs[(32 * u + acc_row(r, hf)) * LS + ln];
        const float* qc = &Qs[(32 * u + acc_row(r, hf)) * LS + ln];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          dv[n] = mfma32(gc[32 * n], s[r], dv[n]);
          dk[n] = mfma32(qc[32 * n], dp[r], dk[n]);
        }
      }
    }
  }
  if (kvalid) {
    float* dkp = p.dk + (int64_t)b * p.dks.sb + (int64_t)kj * p.dks.st + (int64_t)h * p.dks.sh + 4 * hf;
    float* dvp = p.dv + (int64_t)b * p.dvs.sb + (int64_t)kj * p.dvs.st + (int64_t)h * p.dvs.sh + 4 * hf;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        st4(dkp + 32 * n + 8 * g, make_float4(dk[n][4 * g] * AMK_LN2, dk[n][4 * g + 1] * AMK_LN2, dk[n][4 * g + 2] * AMK_LN2, dk[n][4 * g + 3] * AMK_LN2));
        st4(dvp + 32 * n + 8 * g, make_float4(dv[n][4 * g], dv[n][4 * g + 1], dv[n][4 * g + 2], dv[n][4 * g + 3]));
      }
  }
}

// ------------------------------------------------------------------------------------------------
template <int DH>
static void launch_fwd_gen(const FwdParams& p, int64_t nwg, hipStream_t st) {
  if (p.causal_mask) hipLaunchKernelGGL((attn_fwd_gen_kernel<DH, true>), dim3((unsigned)nwg), dim3(WG), 0, st, p);
  else if (!p.key_mask && p.scores) hipLaunchKernelGGL((attn_fwd_gen_plain_kernel<DH, true>), dim3((unsigned)nwg), dim3(WG), 0, st, p);
  else if (!p.key_mask) hipLaunchKernelGGL((attn_fwd_gen_plain_kernel<DH, false>), dim3((unsigned)nwg), dim3(WG), 0, st, p);
  else hipLaunchKernelGGL((attn_fwd_gen_kernel<DH, false>), dim3((unsigned)nwg), dim3(WG), 0, st, p);
}

bool attn_gen_supported(int Dh) { return Dh == 32 || Dh == 128; }

void launch_attn_fwd_gen(const FwdParams& p, int Dh, int64_t nwg, hipStream_t st) {
  if (Dh == 32) launch_fwd_gen<32>(p, nwg, st);
  else launch_fwd_gen<128>(p, nwg, st);
}

template <int DH>
static void launch_bwd_gen(const BwdParams& p, int stages, hipStream_t st) {
  const int64_t nrow = (int64_t)p.B * p.H * p.I;
  const int64_t nq = (int64_t)p.B * p.H * p.nqblk, nk = (int64_t)p.B * p.H * p.nkblk;
  constexpr int RPB = 256 / (DH / 4);
  if (stages & AMK_ATTN_BWD_DELTA)
    hipLaunchKernelGGL(attn_bwd_delta_gen_kernel<DH>, dim3((unsigned)((nrow + RPB - 1) / RPB)), dim3(256), 0, st, p);
  if (stages & AMK_ATTN_BWD_DKDV) {
    if (p.causal_mask) hipLaunchKernelGGL((attn_bwd_dkdv_gen_kernel<DH, true>), dim3((unsigned)nk), dim3(WG), 0, st, p);
    else hipLaunchKernelGGL((attn_bwd_dkdv_gen_kernel<DH, false>), dim3((unsigned)nk), dim3(WG), 0, st, p);
  }
  if (stages & AMK_ATTN_BWD_DQ) {
    if (p.causal_mask) hipLaunchKernelGGL((attn_bwd_dq_gen_kernel<DH, true>), dim3((unsigned)nq), dim3(WG), 0, st, p);
    else hipLaunchKernelGGL((attn_bwd_dq_gen_kernel<DH, false>), dim3((unsigned)nq), dim3(WG), 0, st, p);
  }
}

// stages as in amk_attn_bwd; the one-pass kernel exists for D = 64 only, so FUSED means DKDV | DQ here
void launch_attn_bwd_gen(const BwdParams& p, int Dh, int stages, hipStream_t st) {
  if (stages & AMK_ATTN_BWD_FUSED) stages |= AMK_ATTN_BWD_DKDV | AMK_ATTN_BWD_DQ;
  if (Dh == 32) launch_bwd_gen<32>(p, stages, st);
  else launch_bwd_gen<128>(p, stages, st);
}

}  // namespace amk_attn
