// AgentAttention core for gfx950.
//
// Replaces models/agent_attention.py:55-73 of the reference: adaptive-avg-pool of q into p
// agent tokens per head, agent aggregation softmax((A*scale) K^T) V, agent broadcast
// softmax((q*scale) A^T) V_agent, plus a depthwise 3x3 convolution of v over the
// (head, token) plane.  With p <= 16 agents the work is O(T*p*d) per head -- HBM/VALU bound,
// nowhere near a GEMM.  The sequence is cut into 256-token chunks and every kernel runs one
// workgroup per (batch, head, chunk) -- B*h*ceil(T/256) workgroups, so a (2, 1024) call still
// spreads over 48 CUs and a (64, 1024) call over-subscribes the chip 6x -- with the sums over
// tokens (agent aggregation, dV_agent, dA, conv weight gradients) written as per-chunk partials
// and folded by tiny combine kernels in fixed chunk order (deterministic, no atomics):
//     forward : pool -> s1_partial -> s1_combine -> s2
//     backward: s2_bwd -> mid -> s1_bwd -> pool_bwd
// Two thread mappings per chunk:
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
constexpr int PSTR = D + 2;  // forward partial record per (chunk, agent): D sums, chunk max, chunk row sum

struct Strides { int64_t sb, st, sh; };

struct Params {
  const float *q, *k, *v;          // (B,h,T,d) views
  const float *convw, *convb;      // (d,1,3,3), (d)
  float* o;                        // (B,h,T,d) view
  float *agents, *vagent, *stats1; // (B,h,p,d), (B,h,p,d), (B,h,p,2) saved for backward
  float* part;                     // (B,h,NC,p,PSTR) stage-1 partials
  int B, H, T, P, NC;
  Strides qs, ks, vs, os;
  float scale;
};

struct BwdParams {
  const float *q, *k, *v, *d_o, *convw;
  const float *agents, *vagent, *stats1;
  float *dq, *dk, *dv;             // (B,h,T,d) views, fully overwritten
  float *pva, *pa2, *pa1;          // (B,h,NC,p,d) per-chunk partials: dV_agent, dA (stage 2), dA (stage 1)
  float *dva, *da2, *delta1;       // (B,h,p,d), (B,h,p,d), (B,h,p) folded by the mid kernel
  float *dconvw_part, *dconvb_part;// (B*h*NC, 9, d), (B*h*NC, d) partial sums
  int B, H, T, P, NC;
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

// sum of the four waves' entries red[(w*MAXP + i)*D + lane]
__device__ __forceinline__ float fold4(const float* red, int i, int lane) {
  return red[(0 * MAXP + i) * D + lane] + red[(1 * MAXP + i) * D + lane] + red[(2 * MAXP + i) * D + lane] +
         red[(3 * MAXP + i) * D + lane];
}

// blockIdx -> (b, h, chunk); chunk fastest so neighbouring workgroups share the conv halo rows in L2
struct Where { int b, h, ch, t0; int64_t bh; };
__device__ __forceinline__ Where where(int H, int NC) {
  Where w;
  w.ch = blockIdx.x % NC;
  const int bh = blockIdx.x / NC;
  w.h = bh % H; w.b = bh / H; w.bh = bh; w.t0 = w.ch * CH;
  return w;
}

// ---------------------------------------------------------------------------------------
// forward 0: agent tokens = mean of q over the adaptive bin (AdaptiveAvgPool2d over (t, h), h == p).
// One workgroup per (b, h, agent).
__global__ __launch_bounds__(256) void agent_pool_kernel(Params p) {
  __shared__ float red[4 * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x % p.P;
  const int bh = blockIdx.x / p.P;
  const int h = bh % p.H, b = bh / p.H;
  const float* qb = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const int lo = bin_lo(i, p.T, p.P), hi = bin_hi(i, p.T, p.P);
  float s = 0.f;
  for (int t = lo + wave; t < hi; t += 4) s += qb[(int64_t)t * p.qs.st + lane];
  red[wave * D + lane] = s;
  __syncthreads();
  if (wave == 0)
    p.agents[((int64_t)bh * p.P + i) * D + lane] =
        (red[lane] + red[D + lane] + red[2 * D + lane] + red[3 * D + lane]) / (float)(hi - lo);
}

// forward 1: per-chunk partial of V_agent = softmax((A*scale) K^T) V: chunk max, chunk row sum and
// the un-normalised sum over the chunk's keys.
__global__ __launch_bounds__(CH) void agent_s1_partial_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) float As[MAXP * D];
  __shared__ float S[MAXP * CH];
  __shared__ float red[4 * MAXP * D];
  __shared__ float mloc[MAXP], lloc[MAXP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, t0 = w.t0;
  const float* kb = p.k + (int64_t)w.b * p.ks.sb + (int64_t)w.h * p.ks.sh;
  const float* vb = p.v + (int64_t)w.b * p.vs.sb + (int64_t)w.h * p.vs.sh;

  for (int i = wave; i < P; i += 4) As[i * D + lane] = p.agents[(w.bh * P + i) * D + lane];
  __syncthreads();
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
    if (lane == 0) mloc[i] = m;  // finite: every chunk holds at least one key
  }
  __syncthreads();
  for (int i = 0; i < P; ++i) S[i * CH + tid] = expf(S[i * CH + tid] - mloc[i]);
  __syncthreads();
  {  // phase B: lane <-> channel, this wave's 64 keys
    float acc[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) acc[i] = 0.f;
    const int tend = min(64, T - (t0 + 64 * wave));
    for (int tt = 0; tt < tend; ++tt) {
      const float vv = vb[(int64_t)(t0 + 64 * wave + tt) * p.vs.st + lane];
      for (int i = 0; i < P; ++i) acc[i] += S[i * CH + 64 * wave + tt] * vv;
    }
    for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = acc[i];
    for (int i = wave; i < P; i += 4) {  // row sums of this chunk, one wave per agent
      float s = S[i * CH + lane] + S[i * CH + 64 + lane] + S[i * CH + 128 + lane] + S[i * CH + 192 + lane];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) lloc[i] = s;
    }
  }
  __syncthreads();
  for (int i = wave; i < P; i += 4) {
    float* rec = p.part + ((w.bh * p.NC + w.ch) * P + i) * PSTR;
    rec[lane] = fold4(red, i, lane);
    if (lane == 0) { rec[D] = mloc[i]; rec[D + 1] = lloc[i]; }
  }
}

// forward 2: fold the chunk partials (fixed chunk order): V_agent and the (max, sum) stats.
// One workgroup per (b, h).
__global__ __launch_bounds__(256) void agent_s1_combine_kernel(Params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bh = blockIdx.x;
  for (int i = wave; i < p.P; i += 4) {
    const float* rec0 = p.part + (bh * p.NC * p.P + i) * PSTR;
    float M = -INFINITY;
    for (int c = 0; c < p.NC; ++c) M = fmaxf(M, rec0[(int64_t)c * p.P * PSTR + D]);
    float L = 0.f, s = 0.f;
    for (int c = 0; c < p.NC; ++c) {
      const float* rec = rec0 + (int64_t)c * p.P * PSTR;
      const float a = expf(rec[D] - M);
      L += a * rec[D + 1];
      s += a * rec[lane];
    }
    const int64_t row = bh * p.P + i;
    p.vagent[row * D + lane] = s / L;
    if (lane == 0) { p.stats1[row * 2] = M; p.stats1[row * 2 + 1] = L; }
  }
}

// forward 3: O = softmax((q*scale) A^T) V_agent + dwc(v) for one chunk of tokens.
__global__ __launch_bounds__(CH) void agent_s2_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) float As[MAXP * D];
  __shared__ float S[MAXP * CH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, t0 = w.t0;
  const float* qb = p.q + (int64_t)w.b * p.qs.sb + (int64_t)w.h * p.qs.sh;

  for (int i = wave; i < P; i += 4) As[i * D + lane] = p.agents[(w.bh * P + i) * D + lane];
  float w9[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) w9[j] = p.convw[lane * 9 + j];
  const float cb = p.convb[lane];
  float var[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) var[i] = (i < P) ? p.vagent[(w.bh * P + i) * D + lane] : 0.f;
  __syncthreads();
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
    const float* vbatch = p.v + (int64_t)w.b * p.vs.sb;
    float* ob = p.o + (int64_t)w.b * p.os.sb + (int64_t)w.h * p.os.sh;
    const int tend = min(64, T - (t0 + 64 * wave));
    for (int tt = 0; tt < tend; ++tt) {
      const int t = t0 + 64 * wave + tt;
      float o = 0.f;
      for (int i = 0; i < P; ++i) o += S[i * CH + 64 * wave + tt] * var[i];
      o += conv_at(vbatch, p.vs, p.H, T, w.h, t, lane, w9, cb);
      ob[(int64_t)t * p.os.st + lane] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward 0: stage-2 backward of one chunk: dq (broadcast part), partials of dV_agent, of dA
// (broadcast part) and of the conv weight / bias gradients.
__global__ __launch_bounds__(CH) void agent_s2_bwd_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float As[MAXP * D];
  __shared__ __attribute__((aligned(16))) float Vas[MAXP * D];
  __shared__ float S[MAXP * CH];    // P2 of the chunk
  __shared__ float DS[MAXP * CH];   // dS2 of the chunk
  __shared__ float red[4 * MAXP * D];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, t0 = w.t0, h = w.h;
  const float* qb = p.q + (int64_t)w.b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* gb = p.d_o + (int64_t)w.b * p.dos.sb + (int64_t)h * p.dos.sh;
  float* dqb = p.dq + (int64_t)w.b * p.dqs.sb + (int64_t)h * p.dqs.sh;
  const float* vbatch = p.v + (int64_t)w.b * p.vs.sb;

  for (int i = wave; i < P; i += 4) {
    As[i * D + lane] = p.agents[(w.bh * P + i) * D + lane];
    Vas[i * D + lane] = p.vagent[(w.bh * P + i) * D + lane];
  }
  __syncthreads();
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
  float accva[MAXP], acca[MAXP], dw9[9], dbs = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) { accva[i] = 0.f; acca[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < 9; ++j) dw9[j] = 0.f;
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
  const int64_t cell = w.bh * p.NC + w.ch;
  for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = accva[i];
  __syncthreads();
  for (int i = wave; i < P; i += 4) p.pva[(cell * P + i) * D + lane] = fold4(red, i, lane);
  __syncthreads();
  for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = acca[i];
  __syncthreads();
  for (int i = wave; i < P; i += 4) p.pa2[(cell * P + i) * D + lane] = p.scale * fold4(red, i, lane);
  __syncthreads();
  for (int j = 0; j < 9; ++j) red[(wave * MAXP + j) * D + lane] = dw9[j];
  red[(wave * MAXP + 9) * D + lane] = dbs;
  __syncthreads();
  if (wave == 0) {
    for (int j = 0; j < 9; ++j) p.dconvw_part[(cell * 9 + j) * D + lane] = fold4(red, j, lane);
    p.dconvb_part[cell * D + lane] = fold4(red, 9, lane);
  }
}

// backward 1: fold the stage-2 partials in chunk order: dV_agent, dA (stage 2), delta1 = <dV_agent, V_agent>.
// One workgroup per (b, h).
__global__ __launch_bounds__(256) void agent_mid_kernel(BwdParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bh = blockIdx.x;
  for (int i = wave; i < p.P; i += 4) {
    float sva = 0.f, sa2 = 0.f;
    for (int c = 0; c < p.NC; ++c) {
      const int64_t o = (((bh * p.NC + c) * p.P) + i) * D + lane;
      sva += p.pva[o];
      sa2 += p.pa2[o];
    }
    const int64_t row = bh * p.P + i;
    p.dva[row * D + lane] = sva;
    p.da2[row * D + lane] = sa2;
    float s = sva * p.vagent[row * D + lane];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) p.delta1[row] = s;
  }
}

// backward 2: stage-1 backward of one chunk: dk, dv (aggregation + transposed conv of dO), partial
// of dA (aggregation part).
__global__ __launch_bounds__(CH) void agent_s1_bwd_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float As[MAXP * D];
  __shared__ __attribute__((aligned(16))) float dVas[MAXP * D];
  __shared__ float S[MAXP * CH];    // P1 of the chunk
  __shared__ float DS[MAXP * CH];   // dS1 of the chunk
  __shared__ float red[4 * MAXP * D];
  __shared__ float m1[MAXP], l1[MAXP], delta1[MAXP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, t0 = w.t0, h = w.h;
  const float* kb = p.k + (int64_t)w.b * p.ks.sb + (int64_t)h * p.ks.sh;
  const float* vb = p.v + (int64_t)w.b * p.vs.sb + (int64_t)h * p.vs.sh;
  float* dkb = p.dk + (int64_t)w.b * p.dks.sb + (int64_t)h * p.dks.sh;
  float* dvb = p.dv + (int64_t)w.b * p.dvs.sb + (int64_t)h * p.dvs.sh;
  const float* gbatch = p.d_o + (int64_t)w.b * p.dos.sb;

  for (int i = wave; i < P; i += 4) {
    const int64_t row = w.bh * P + i;
    As[i * D + lane] = p.agents[row * D + lane];
    dVas[i * D + lane] = p.dva[row * D + lane];
    if (lane == 0) { m1[i] = p.stats1[row * 2]; l1[i] = p.stats1[row * 2 + 1]; delta1[i] = p.delta1[row]; }
  }
  float w9[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) w9[j] = p.convw[lane * 9 + j];
  __syncthreads();
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
  float acca[MAXP], dva[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) { acca[i] = 0.f; dva[i] = (i < P) ? dVas[i * D + lane] : 0.f; }
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
  for (int i = 0; i < P; ++i) red[(wave * MAXP + i) * D + lane] = acca[i];
  __syncthreads();
  const int64_t cell = w.bh * p.NC + w.ch;
  for (int i = wave; i < P; i += 4) p.pa1[(cell * P + i) * D + lane] = p.scale * fold4(red, i, lane);
}

// backward 3: dq[b,h,t,:] += sum over the bins i containing t of dA[b,h,i,:] / len(bin i), with
// dA = dA(stage 2) + the chunk partials of dA(stage 1) folded in chunk order.
__global__ __launch_bounds__(256) void agent_pool_bwd_kernel(BwdParams p) {
  const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b,h,t)
  const int c = threadIdx.x & 63;
  const int T = p.T, P = p.P;
  if (idx >= (int64_t)p.B * p.H * T) return;
  const int t = (int)(idx % T);
  const int64_t bh = idx / T;
  const int h = (int)(bh % p.H), b = (int)(bh / p.H);
  float add = 0.f;
  for (int i = 0; i < P; ++i) {
    const int lo = bin_lo(i, T, P), hi = bin_hi(i, T, P);
    if (t >= lo && t < hi) {
      float da = p.da2[(bh * P + i) * D + c];
      for (int ch = 0; ch < p.NC; ++ch) da += p.pa1[(((bh * p.NC + ch) * P) + i) * D + c];
      add += da / (float)(hi - lo);
    }
  }
  float* dst = p.dq + (int64_t)b * p.dqs.sb + (int64_t)h * p.dqs.sh + (int64_t)t * p.dqs.st + c;
  *dst += add;
}

}  // namespace amk_agent

using namespace amk_agent;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool sok(const Strides& s) { return s.sb % 4 == 0 && s.st % 4 == 0 && s.sh % 4 == 0; }
static int nchunks(int T) { return (T + CH - 1) / CH; }

extern "C" int amk_agent_num_chunks(int T) { return T > 0 ? nchunks(T) : 0; }

extern "C" int64_t amk_agent_ws_floats(int B, int H, int T, int P, int backward) {
  if (B <= 0 || H <= 0 || T <= 0 || P <= 0) return 0;
  const int64_t cells = (int64_t)B * H * nchunks(T) * P, rows = (int64_t)B * H * P;
  return backward ? 3 * cells * D + 2 * rows * D + rows : cells * PSTR;
}

extern "C" int amk_agent_attn_fwd(const float* q, const float* k, const float* v, const float* conv_w, const float* conv_b,
                                  float* o, float* agents, float* vagent, float* stats1, float* ws,
                                  int B, int H, int T, int Dh, int P,
                                  int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                                  int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                                  float scale, void* stream) {
  AMK_CHECK_ARG(q && k && v && conv_w && conv_b && o && agents && vagent && stats1 && ws, "amk_agent_attn_fwd: null pointer");
  AMK_CHECK_ARG(B > 0 && H > 0 && T > 0 && P > 0, "amk_agent_attn_fwd: non-positive size");
  AMK_CHECK_SUPPORTED(Dh == D, "amk_agent_attn_fwd: head dim %d not supported (built for %d)", Dh, D);
  AMK_CHECK_SUPPORTED(P <= MAXP && P <= T, "amk_agent_attn_fwd: agents per head %d > %d or > T", P, MAXP);
  Params p;
  p.q = q; p.k = k; p.v = v; p.convw = conv_w; p.convb = conv_b; p.o = o;
  p.agents = agents; p.vagent = vagent; p.stats1 = stats1; p.part = ws;
  p.B = B; p.H = H; p.T = T; p.P = P; p.NC = nchunks(T);
  p.qs = {q_sb, q_st, q_sh}; p.ks = {k_sb, k_st, k_sh}; p.vs = {v_sb, v_st, v_sh}; p.os = {o_sb, o_st, o_sh};
  p.scale = scale;
  AMK_CHECK_ARG(a16(q) && a16(k) && a16(v) && a16(o) && sok(p.qs) && sok(p.ks) && sok(p.vs) && sok(p.os),
                "amk_agent_attn_fwd: pointers must be 16-byte aligned and strides multiples of 4");
  const int64_t cells = (int64_t)B * H * p.NC;
  AMK_CHECK_SUPPORTED(cells * P < (1ll << 31), "amk_agent_attn_fwd: B*h*chunks*p exceeds the grid limit");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(agent_pool_kernel, dim3((unsigned)(B * H * P)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(agent_s1_partial_kernel, dim3((unsigned)cells), dim3(CH), 0, st, p);
  hipLaunchKernelGGL(agent_s1_combine_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(agent_s2_kernel, dim3((unsigned)cells), dim3(CH), 0, st, p);
  AMK_CHECK_LAUNCH("amk_agent_attn_fwd");
  return AMK_OK;
}

extern "C" int amk_agent_attn_bwd(const float* q, const float* k, const float* v, const float* conv_w, const float* d_o,
                                  const float* agents, const float* vagent, const float* stats1,
                                  float* dq, float* dk, float* dv, float* ws, float* dconvw_part, float* dconvb_part,
                                  int B, int H, int T, int Dh, int P,
                                  int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                                  int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t do_sb, int64_t do_st, int64_t do_sh,
                                  int64_t dq_sb, int64_t dq_st, int64_t dq_sh, int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                                  int64_t dv_sb, int64_t dv_st, int64_t dv_sh, float scale, void* stream) {
  AMK_CHECK_ARG(q && k && v && conv_w && d_o && agents && vagent && stats1 && dq && dk && dv && ws && dconvw_part &&
                    dconvb_part, "amk_agent_attn_bwd: null pointer");
  AMK_CHECK_ARG(B > 0 && H > 0 && T > 0 && P > 0, "amk_agent_attn_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(Dh == D && P <= MAXP && P <= T, "amk_agent_attn_bwd: unsupported d=%d / p=%d", Dh, P);
  BwdParams p;
  p.q = q; p.k = k; p.v = v; p.d_o = d_o; p.convw = conv_w; p.agents = agents; p.vagent = vagent; p.stats1 = stats1;
  p.dq = dq; p.dk = dk; p.dv = dv; p.dconvw_part = dconvw_part; p.dconvb_part = dconvb_part;
  p.B = B; p.H = H; p.T = T; p.P = P; p.NC = nchunks(T);
  const int64_t cells = (int64_t)B * H * p.NC, rows = (int64_t)B * H * P;
  p.pva = ws; p.pa2 = p.pva + cells * P * D; p.pa1 = p.pa2 + cells * P * D;
  p.dva = p.pa1 + cells * P * D; p.da2 = p.dva + rows * D; p.delta1 = p.da2 + rows * D;
  p.qs = {q_sb, q_st, q_sh}; p.ks = {k_sb, k_st, k_sh}; p.vs = {v_sb, v_st, v_sh}; p.dos = {do_sb, do_st, do_sh};
  p.dqs = {dq_sb, dq_st, dq_sh}; p.dks = {dk_sb, dk_st, dk_sh}; p.dvs = {dv_sb, dv_st, dv_sh};
  p.scale = scale;
  AMK_CHECK_ARG(a16(q) && a16(k) && a16(v) && a16(d_o) && a16(dq) && a16(dk) && a16(dv) && sok(p.qs) && sok(p.ks) &&
                    sok(p.vs) && sok(p.dos) && sok(p.dqs) && sok(p.dks) && sok(p.dvs),
                "amk_agent_attn_bwd: pointers must be 16-byte aligned and strides multiples of 4");
  const int64_t tok = (int64_t)B * H * T;
  AMK_CHECK_SUPPORTED((tok + 3) / 4 < (1ll << 31), "amk_agent_attn_bwd: B*h*T exceeds the grid limit");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(agent_s2_bwd_kernel, dim3((unsigned)cells), dim3(CH), 0, st, p);
  hipLaunchKernelGGL(agent_mid_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(agent_s1_bwd_kernel, dim3((unsigned)cells), dim3(CH), 0, st, p);
  hipLaunchKernelGGL(agent_pool_bwd_kernel, dim3((unsigned)((tok + 3) / 4)), dim3(256), 0, st, p);
  AMK_CHECK_LAUNCH("amk_agent_attn_bwd");
  return AMK_OK;
}
