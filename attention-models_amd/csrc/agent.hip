// AgentAttention core for gfx950.
//
// Replaces models/agent_attention.py:55-73 of the reference: adaptive-avg-pool of q into p
// agent tokens per head, agent aggregation softmax((A*scale) K^T) V, agent broadcast
// softmax((q*scale) A^T) V_agent, plus a depthwise 3x3 convolution of v over the
// (head, token) plane.  With p <= 16 agents the work is O(T*p*d) per head -- HBM/VALU bound,
// nowhere near a GEMM -- so it is ONE workgroup per (batch, head) walking the sequence in
// 256-token chunks with two thread mappings per chunk:
//     phase A  thread <-> token : the p dot products of a q/k row against the agents
//                                 (agents broadcast from LDS), the p-wide softmaxes
//     phase B  lane   <-> channel: sums over tokens (coalesced 256-B row reads), the
//                                 depthwise convolution, coalesced O stores
// bias1 / bias2 are scalars added to every score of a softmax row: the softmax is invariant
// to them, so they do not enter the arithmetic and their gradient is exactly 0.
#include "amk_common.h"

namespace amk_agent {

constexpr int D = 64;
constexpr int MAXP = 16;
constexpr int CH = 256;  // tokens per chunk = threads per workgroup

struct Strides { int64_t sb, st, sh; };

struct Params {
  const float *q, *k, *v;          // (B,h,T,d) views
  const float *convw, *convb;      // (d,1,3,3), (d)
  float* o;                        // (B,h,T,d) view
  float *agents, *vagent, *stats1; // (B,h,p,d), (B,h,p,d), (B,h,p,2) saved for backward
  int B, H, T, P;
  Strides qs, ks, vs, os;
  float scale;
};

struct BwdParams {
  const float *q, *k, *v, *d_o, *convw;
  const float *agents, *vagent, *stats1;
  float *dq, *dk, *dv;             // (B,h,T,d) views, fully overwritten (dq lacks the pool term)
  float *dagents;                  // (B,h,p,d): gradient of the agent tokens (pool backward input)
  float *dconvw_part, *dconvb_part;// (B*h, 9, d), (B*h, d) partial sums
  int B, H, T, P;
  Strides qs, ks, vs, dos, dqs, dks, dvs;
  float scale;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ int bin_lo(int i, int T, int P) { return (int)(((int64_t)i * T) / P); }
__device__ __forceinline__ int bin_hi(int i, int T, int P) { return (int)((((int64_t)(i + 1)) * T + P - 1) / P); }

__device__ __forceinline__ void load_row(const float* p, bool ok, float (&r)[D]) {
#pragma unroll
  for (int c4 = 0; c4 < D / 4; ++c4) {
    const float4 t = ok ? ld4(p + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    r[4 * c4] = t.x; r[4 * c4 + 1] = t.y; r[4 * c4 + 2] = t.z; r[4 * c4 + 3] = t.w;
  }
}

__device__ __forceinline__ float dot_row(const float (&r)[D], const float* a) {
  float s = 0.f;
#pragma unroll
  for (int c4 = 0; c4 < D / 4; ++c4) {
    const float4 t = ld4(a + 4 * c4);  // same address in every lane: LDS broadcast
    s += r[4 * c4] * t.x + r[4 * c4 + 1] * t.y + r[4 * c4 + 2] * t.z + r[4 * c4 + 3] * t.w;
  }
  return s;
}

// depthwise 3x3 over the (head, token) plane, zero padded: channel c = lane
__device__ __forceinline__ float conv_at(const float* vb, const Strides& vs, int H, int T, int hh, int t, int c,
                                         const float (&w)[9], float bias) {
  float acc = bias;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int h2 = hh + a - 1;
    if (h2 < 0 || h2 >= H) continue;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int t2 = t + b - 1;
      if (t2 < 0 || t2 >= T) continue;
      acc += w[a * 3 + b] * vb[(int64_t)h2 * vs.sh + (int64_t)t2 * vs.st + c];
    }
  }
  return acc;
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(CH) void agent_fwd_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) float As[MAXP * D];    // agent tokens (unscaled)
  __shared__ __attribute__((aligned(16))) float Vas[MAXP * D];   // V_agent
  __shared__ float S[MAXP * CH];                                 // chunk scores / probabilities
  __shared__ float red[4 * MAXP * D];                            // cross-wave reductions
  __shared__ float mrun[MAXP], lrun[MAXP], alpha[MAXP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
  const int P = p.P, T = p.T;
  const float* qb = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* kb = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vb = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;

  // ---- agent tokens: mean of q over the adaptive bin (AdaptiveAvgPool2d over (t, h), h == p)
  for (int i = wave; i < P; i += 4) {
    const int lo = bin_lo(i, T, P), hi = bin_hi(i, T, P);
    float s = 0.f;
    for (int t = lo; t < hi; ++t) s += qb[(int64_t)t * p.qs.st + lane];
    As[i * D + lane] = s / (float)(hi - lo);
  }
  if (tid < P) { mrun[tid] = -INFINITY; lrun[tid] = 0.f; }
  __syncthreads();

  // ---- stage 1: V_agent = softmax((A*scale) K^T) V, online over 256-key chunks
  float acc[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) acc[i] = 0.f;
  for (int t0 = 0; t0 < T; t0 += CH) {
    {  // phase A: thread <-> key
      const int t = t0 + tid;
      float kr[D];
      load_row(kb + (int64_t)t * p.ks.st, t < T, kr);
      for (int i = 0; i < P; ++i) {
        float a_s = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; ++c4) {
          const float4 a = ld4(&As[i * D + 4 * c4]);
          a_s += (a.x * p.scale) * kr[4 * c4] + (a.y * p.scale) * kr[4 * c4 + 1] + (a.z * p.scale) * kr[4 * c4 + 2] +
                 (a.w * p.scale) * kr[4 * c4 + 3];
        }
        S[i * CH + tid] = (t < T) ? a_s : -INFINITY;
      }
    }
    __syncthreads();
    for (int i = wave; i < P; i += 4) {  // chunk max per agent: one wave per agent
      float m = fmaxf(fmaxf(S[i * CH + lane], S[i * CH + 64 + lane]), fmaxf(S[i * CH + 128 + lane], S[i * CH + 192 + lane]));
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (lane == 0) {
        const float mn = fmaxf(mrun[i], m);
        alpha[i] = expf(mrun[i] - mn);
        mrun[i] = mn;
      }
    }
    __syncthreads();
    for (int i = 0; i < P; ++i) {  // probabilities in place (every thread its own key)
      S[i * CH + tid] = expf(S[i * CH + tid] - mrun[i]);
    }
    __syncthreads();
    {  // phase B: lane <-> channel, this wave's 64 keys
      for (int i = 0; i < P; ++i) acc[i] *= alpha[i];
      const int tend = min(64, T - (t0 + 64 * wave));
      for (int tt = 0; tt < tend; ++tt) {
        const float vv = vb[(int64_t)(t0 + 64 * wave + tt) * p.vs.st + lane];
        for (int i = 0; i < P; ++i) acc[i] += S[i * CH + 64 * wave + tt] * vv;
      }
      for (int i = wave; i < P; i += 4) {  // row sums of this chunk, one wave per agent
        float s = S[i * CH + lane] + S[i * CH + 64 + lane] + S[i * CH + 128 + lane] + S[i * CH + 192 + lane];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) lrun[i] = lrun[i] * alpha[i] + s;
      }
    }
    __syncthreads();
  }
  for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = acc[i];
  __syncthreads();
  for (int i = wave; i < P; i += 4) {
    const float s = red[(0 * MAXP + i) * D + lane] + red[(1 * MAXP + i) * D + lane] + red[(2 * MAXP + i) * D + lane] +
                    red[(3 * MAXP + i) * D + lane];
    const float va = s / lrun[i];
    Vas[i * D + lane] = va;
    const int64_t row = ((int64_t)b * p.H + h) * P + i;
    p.vagent[row * D + lane] = va;
    p.agents[row * D + lane] = As[i * D + lane];
    if (lane == 0) { p.stats1[row * 2] = mrun[i]; p.stats1[row * 2 + 1] = lrun[i]; }
  }
  __syncthreads();

  // ---- stage 2: O = softmax((q*scale) A^T) V_agent + dwc(v)
  float w9[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) w9[j] = p.convw[lane * 9 + j];
  const float cb = p.convb[lane];
  float var[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) var[i] = (i < P) ? Vas[i * D + lane] : 0.f;
  const float* vbatch = p.v + (int64_t)b * p.vs.sb;
  float* ob = p.o + (int64_t)b * p.os.sb + (int64_t)h * p.os.sh;
  for (int t0 = 0; t0 < T; t0 += CH) {
    {  // phase A: thread <-> token: p scores, softmax over the agents
      const int t = t0 + tid;
      float qr[D];
      load_row(qb + (int64_t)t * p.qs.st, t < T, qr);
#pragma unroll
      for (int c = 0; c < D; ++c) qr[c] *= p.scale;
      float sc[MAXP];
      float m = -INFINITY;
      for (int i = 0; i < P; ++i) { sc[i] = dot_row(qr, &As[i * D]); m = fmaxf(m, sc[i]); }
      float l = 0.f;
      for (int i = 0; i < P; ++i) { sc[i] = expf(sc[i] - m); l += sc[i]; }
      for (int i = 0; i < P; ++i) S[i * CH + tid] = sc[i] / l;
    }
    __syncthreads();
    {  // phase B: lane <-> channel
      const int tend = min(64, T - (t0 + 64 * wave));
      for (int tt = 0; tt < tend; ++tt) {
        const int t = t0 + 64 * wave + tt;
        float o = 0.f;
        for (int i = 0; i < P; ++i) o += S[i * CH + 64 * wave + tt] * var[i];
        o += conv_at(vbatch, p.vs, p.H, T, h, t, lane, w9, cb);
        ob[(int64_t)t * p.os.st + lane] = o;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(CH) void agent_bwd_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float As[MAXP * D];
  __shared__ __attribute__((aligned(16))) float Vas[MAXP * D];
  __shared__ __attribute__((aligned(16))) float dVas[MAXP * D];
  __shared__ float S[MAXP * CH];    // P2 / P1 of the chunk
  __shared__ float DS[MAXP * CH];   // dS2 / dS1 of the chunk
  __shared__ float red[4 * MAXP * D];
  __shared__ float m1[MAXP], l1[MAXP], delta1[MAXP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
  const int P = p.P, T = p.T;
  const int64_t bh = (int64_t)b * p.H + h;
  const float* qb = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* kb = p.k + (int64_t)b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vb = p.v + (int64_t)b * p.vs.sb + (int64_t)h * p.vs.sh;
  const float* gb = p.d_o + (int64_t)b * p.dos.sb + (int64_t)h * p.dos.sh;
  float* dqb = p.dq + (int64_t)b * p.dqs.sb + (int64_t)h * p.dqs.sh;
  float* dkb = p.dk + (int64_t)b * p.dks.sb + (int64_t)h * p.dks.sh;
  float* dvb = p.dv + (int64_t)b * p.dvs.sb + (int64_t)h * p.dvs.sh;

  for (int i = wave; i < P; i += 4) {
    As[i * D + lane] = p.agents[(bh * P + i) * D + lane];
    Vas[i * D + lane] = p.vagent[(bh * P + i) * D + lane];
    if (lane == 0) { m1[i] = p.stats1[(bh * P + i) * 2]; l1[i] = p.stats1[(bh * P + i) * 2 + 1]; }
  }
  __syncthreads();

  float w9[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) w9[j] = p.convw[lane * 9 + j];
  const float* vbatch = p.v + (int64_t)b * p.vs.sb;
  const float* gbatch = p.d_o + (int64_t)b * p.dos.sb;

  // ---- stage-2 backward: dq (broadcast part), dV_agent, dA (broadcast part), conv weight grads
  float accva[MAXP], acca[MAXP], dw9[9], dbs = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) { accva[i] = 0.f; acca[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < 9; ++j) dw9[j] = 0.f;
  for (int t0 = 0; t0 < T; t0 += CH) {
    {  // phase A: thread <-> token
      const int t = t0 + tid;
      const bool ok = t < T;
      float qr[D], gr[D];
      load_row(qb + (int64_t)t * p.qs.st, ok, qr);
      load_row(gb + (int64_t)t * p.dos.st, ok, gr);
      float sc[MAXP], dp[MAXP];
      float m = -INFINITY;
      for (int i = 0; i < P; ++i) {
        sc[i] = dot_row(qr, &As[i * D]) * p.scale;
        dp[i] = dot_row(gr, &Vas[i * D]);
        m = fmaxf(m, sc[i]);
      }
      float l = 0.f;
      for (int i = 0; i < P; ++i) { sc[i] = expf(sc[i] - m); l += sc[i]; }
      float dl = 0.f;
      for (int i = 0; i < P; ++i) { sc[i] /= l; dl += sc[i] * dp[i]; }
      for (int i = 0; i < P; ++i) {
        const float ds = sc[i] * (dp[i] - dl);
        S[i * CH + tid] = ok ? sc[i] : 0.f;
        DS[i * CH + tid] = ok ? ds : 0.f;
        dp[i] = ds;
      }
      if (ok) {  // dq_t = scale * sum_i dS2[t,i] A_i   (row-per-lane store)
        float* dst = dqb + (int64_t)t * p.dqs.st;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; ++c4) {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int i = 0; i < P; ++i) {
            const float4 a = ld4(&As[i * D + 4 * c4]);
            o.x += dp[i] * a.x; o.y += dp[i] * a.y; o.z += dp[i] * a.z; o.w += dp[i] * a.w;
          }
          st4(dst + 4 * c4, make_float4(o.x * p.scale, o.y * p.scale, o.z * p.scale, o.w * p.scale));
        }
      }
    }
    __syncthreads();
    {  // phase B: lane <-> channel
      const int tend = min(64, T - (t0 + 64 * wave));
      for (int tt = 0; tt < tend; ++tt) {
        const int t = t0 + 64 * wave + tt;
        const float g = gb[(int64_t)t * p.dos.st + lane];
        const float qv = qb[(int64_t)t * p.qs.st + lane];
        for (int i = 0; i < P; ++i) {
          accva[i] += S[i * CH + 64 * wave + tt] * g;
          acca[i] += DS[i * CH + 64 * wave + tt] * qv;
        }
        dbs += g;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int h2 = h + a - 1;
          if (h2 < 0 || h2 >= p.H) continue;
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) {
            const int t2 = t + bb - 1;
            if (t2 < 0 || t2 >= T) continue;
            dw9[a * 3 + bb] += g * vbatch[(int64_t)h2 * p.vs.sh + (int64_t)t2 * p.vs.st + lane];
          }
        }
      }
    }
    __syncthreads();
  }
  // cross-wave reductions: dV_agent, dA (stage 2), conv partials
  for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = accva[i];
  __syncthreads();
  for (int i = wave; i < P; i += 4)
    dVas[i * D + lane] = red[(0 * MAXP + i) * D + lane] + red[(1 * MAXP + i) * D + lane] + red[(2 * MAXP + i) * D + lane] +
                         red[(3 * MAXP + i) * D + lane];
  __syncthreads();
  for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = acca[i];
  __syncthreads();
  float da2[MAXP];  // valid in the wave that owns agent i (i % 4 == wave)
#pragma unroll
  for (int i = 0; i < MAXP; ++i) da2[i] = 0.f;
  for (int i = wave; i < P; i += 4)
    da2[i] = p.scale * (red[(0 * MAXP + i) * D + lane] + red[(1 * MAXP + i) * D + lane] + red[(2 * MAXP + i) * D + lane] +
                        red[(3 * MAXP + i) * D + lane]);
  __syncthreads();
  for (int j = 0; j < 9; ++j) red[(wave * MAXP + j) * D + lane] = dw9[j];
  red[(wave * MAXP + 9) * D + lane] = dbs;
  __syncthreads();
  if (wave == 0) {
    for (int j = 0; j < 9; ++j)
      p.dconvw_part[(bh * 9 + j) * D + lane] = red[(0 * MAXP + j) * D + lane] + red[(1 * MAXP + j) * D + lane] +
                                               red[(2 * MAXP + j) * D + lane] + red[(3 * MAXP + j) * D + lane];
    p.dconvb_part[bh * D + lane] = red[(0 * MAXP + 9) * D + lane] + red[(1 * MAXP + 9) * D + lane] +
                                   red[(2 * MAXP + 9) * D + lane] + red[(3 * MAXP + 9) * D + lane];
  }
  // delta1[i] = <dV_agent_i, V_agent_i>
  for (int i = wave; i < P; i += 4) {
    float s = dVas[i * D + lane] * Vas[i * D + lane];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) delta1[i] = s;
  }
  __syncthreads();

  // ---- stage-1 backward: dk, dv (aggregation + transposed conv of dO), dA (aggregation part)
  float dva[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) { acca[i] = 0.f; dva[i] = (i < P) ? dVas[i * D + lane] : 0.f; }
  for (int t0 = 0; t0 < T; t0 += CH) {
    {  // phase A: thread <-> key
      const int t = t0 + tid;
      const bool ok = t < T;
      float kr[D], vr[D];
      load_row(kb + (int64_t)t * p.ks.st, ok, kr);
      load_row(vb + (int64_t)t * p.vs.st, ok, vr);
      float ds[MAXP];
      for (int i = 0; i < P; ++i) {
        const float s1 = dot_row(kr, &As[i * D]) * p.scale;
        const float pr = expf(s1 - m1[i]) / l1[i];
        const float dp = dot_row(vr, &dVas[i * D]);
        ds[i] = pr * (dp - delta1[i]);
        S[i * CH + tid] = ok ? pr : 0.f;
        DS[i * CH + tid] = ok ? ds[i] : 0.f;
      }
      if (ok) {  // dk_t = scale * sum_i dS1[i,t] A_i
        float* dst = dkb + (int64_t)t * p.dks.st;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; ++c4) {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int i = 0; i < P; ++i) {
            const float4 a = ld4(&As[i * D + 4 * c4]);
            o.x += ds[i] * a.x; o.y += ds[i] * a.y; o.z += ds[i] * a.z; o.w += ds[i] * a.w;
          }
          st4(dst + 4 * c4, make_float4(o.x * p.scale, o.y * p.scale, o.z * p.scale, o.w * p.scale));
        }
      }
    }
    __syncthreads();
    {  // phase B: lane <-> channel
      const int tend = min(64, T - (t0 + 64 * wave));
      for (int tt = 0; tt < tend; ++tt) {
        const int t = t0 + 64 * wave + tt;
        const float kv = kb[(int64_t)t * p.ks.st + lane];
        float dvv = 0.f;
        for (int i = 0; i < P; ++i) {
          acca[i] += DS[i * CH + 64 * wave + tt] * kv;
          dvv += S[i * CH + 64 * wave + tt] * dva[i];
        }
        // transposed depthwise conv: dv[h,t] += sum w[a][b] * dO[h-(a-1), t-(b-1)]
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int h2 = h - (a - 1);
          if (h2 < 0 || h2 >= p.H) continue;
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) {
            const int t2 = t - (bb - 1);
            if (t2 < 0 || t2 >= T) continue;
            dvv += w9[a * 3 + bb] * gbatch[(int64_t)h2 * p.dos.sh + (int64_t)t2 * p.dos.st + lane];
          }
        }
        dvb[(int64_t)t * p.dvs.st + lane] = dvv;
      }
    }
    __syncthreads();
  }
  for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = acca[i];
  __syncthreads();
  for (int i = wave; i < P; i += 4) {
    const float s = red[(0 * MAXP + i) * D + lane] + red[(1 * MAXP + i) * D + lane] + red[(2 * MAXP + i) * D + lane] +
                    red[(3 * MAXP + i) * D + lane];
    p.dagents[(bh * P + i) * D + lane] = da2[i] + p.scale * s;
  }
}

// dq[b,h,t,:] += sum over the bins i containing t of dA[b,h,i,:] / len(bin i)
__global__ __launch_bounds__(256) void agent_pool_bwd_kernel(const float* __restrict__ dagents, float* dq, Strides dqs,
                                                             int B, int H, int T, int P) {
  const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b,h,t)
  const int c = threadIdx.x & 63;
  if (idx >= (int64_t)B * H * T) return;
  const int t = (int)(idx % T);
  const int64_t bh = idx / T;
  const int h = (int)(bh % H), b = (int)(bh / H);
  float add = 0.f;
  for (int i = 0; i < P; ++i) {
    const int lo = bin_lo(i, T, P), hi = bin_hi(i, T, P);
    if (t >= lo && t < hi) add += dagents[(bh * P + i) * D + c] / (float)(hi - lo);
  }
  float* dst = dq + (int64_t)b * dqs.sb + (int64_t)h * dqs.sh + (int64_t)t * dqs.st + c;
  *dst += add;
}

}  // namespace amk_agent

using namespace amk_agent;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool sok(const Strides& s) { return s.sb % 4 == 0 && s.st % 4 == 0 && s.sh % 4 == 0; }

extern "C" int amk_agent_attn_fwd(const float* q, const float* k, const float* v, const float* conv_w, const float* conv_b,
                                  float* o, float* agents, float* vagent, float* stats1,
                                  int B, int H, int T, int Dh, int P,
                                  int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                                  int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                                  float scale, void* stream) {
  AMK_CHECK_ARG(q && k && v && conv_w && conv_b && o && agents && vagent && stats1, "amk_agent_attn_fwd: null pointer");
  AMK_CHECK_ARG(B > 0 && H > 0 && T > 0 && P > 0, "amk_agent_attn_fwd: non-positive size");
  AMK_CHECK_SUPPORTED(Dh == D, "amk_agent_attn_fwd: head dim %d not supported (built for %d)", Dh, D);
  AMK_CHECK_SUPPORTED(P <= MAXP && P <= T, "amk_agent_attn_fwd: agents per head %d > %d or > T", P, MAXP);
  Params p;
  p.q = q; p.k = k; p.v = v; p.convw = conv_w; p.convb = conv_b; p.o = o;
  p.agents = agents; p.vagent = vagent; p.stats1 = stats1;
  p.B = B; p.H = H; p.T = T; p.P = P;
  p.qs = {q_sb, q_st, q_sh}; p.ks = {k_sb, k_st, k_sh}; p.vs = {v_sb, v_st, v_sh}; p.os = {o_sb, o_st, o_sh};
  p.scale = scale;
  AMK_CHECK_ARG(a16(q) && a16(k) && a16(v) && a16(o) && sok(p.qs) && sok(p.ks) && sok(p.vs) && sok(p.os),
                "amk_agent_attn_fwd: pointers must be 16-byte aligned and strides multiples of 4");
  hipLaunchKernelGGL(agent_fwd_kernel, dim3((unsigned)(B * H)), dim3(CH), 0, static_cast<hipStream_t>(stream), p);
  AMK_CHECK_LAUNCH("amk_agent_attn_fwd");
  return AMK_OK;
}

extern "C" int amk_agent_attn_bwd(const float* q, const float* k, const float* v, const float* conv_w, const float* d_o,
                                  const float* agents, const float* vagent, const float* stats1,
                                  float* dq, float* dk, float* dv, float* dagents_ws, float* dconvw_part, float* dconvb_part,
                                  int B, int H, int T, int Dh, int P,
                                  int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                                  int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t do_sb, int64_t do_st, int64_t do_sh,
                                  int64_t dq_sb, int64_t dq_st, int64_t dq_sh, int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                                  int64_t dv_sb, int64_t dv_st, int64_t dv_sh, float scale, void* stream) {
  AMK_CHECK_ARG(q && k && v && conv_w && d_o && agents && vagent && stats1 && dq && dk && dv && dagents_ws && dconvw_part &&
                    dconvb_part, "amk_agent_attn_bwd: null pointer");
  AMK_CHECK_ARG(B > 0 && H > 0 && T > 0 && P > 0, "amk_agent_attn_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(Dh == D && P <= MAXP && P <= T, "amk_agent_attn_bwd: unsupported d=%d / p=%d", Dh, P);
  BwdParams p;
  p.q = q; p.k = k; p.v = v; p.d_o = d_o; p.convw = conv_w; p.agents = agents; p.vagent = vagent; p.stats1 = stats1;
  p.dq = dq; p.dk = dk; p.dv = dv; p.dagents = dagents_ws; p.dconvw_part = dconvw_part; p.dconvb_part = dconvb_part;
  p.B = B; p.H = H; p.T = T; p.P = P;
  p.qs = {q_sb, q_st, q_sh}; p.ks = {k_sb, k_st, k_sh}; p.vs = {v_sb, v_st, v_sh}; p.dos = {do_sb, do_st, do_sh};
  p.dqs = {dq_sb, dq_st, dq_sh}; p.dks = {dk_sb, dk_st, dk_sh}; p.dvs = {dv_sb, dv_st, dv_sh};
  p.scale = scale;
  AMK_CHECK_ARG(a16(q) && a16(k) && a16(v) && a16(d_o) && a16(dq) && a16(dk) && a16(dv) && sok(p.qs) && sok(p.ks) &&
                    sok(p.vs) && sok(p.dos) && sok(p.dqs) && sok(p.dks) && sok(p.dvs),
                "amk_agent_attn_bwd: pointers must be 16-byte aligned and strides multiples of 4");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(agent_bwd_kernel, dim3((unsigned)(B * H)), dim3(CH), 0, st, p);
  const int64_t rows = (int64_t)B * H * T;
  hipLaunchKernelGGL(agent_pool_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dagents_ws, dq, p.dqs, B, H, T, P);
  AMK_CHECK_LAUNCH("amk_agent_attn_bwd");
  return AMK_OK;
}
