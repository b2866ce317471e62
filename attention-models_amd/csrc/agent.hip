// AgentAttention core for gfx950.
//
// Replaces models/agent_attention.py:55-73 of the reference: adaptive-avg-pool of q into p
// agent tokens per head, agent aggregation softmax((A*scale) K^T) V, agent broadcast
// softmax((q*scale) A^T) V_agent, plus a depthwise 3x3 convolution of v over the
// (head, token) plane.  With p <= 16 agents the work is O(T*p*d) per head -- HBM/VALU bound,
// nowhere near a GEMM.  The sequence is cut into 128-token chunks and every kernel runs one
// workgroup per (batch, head, chunk) -- B*h*ceil(T/128) workgroups, so a (2, 1024) call still
// spreads over 96 CUs and a (64, 1024) call over-subscribes the chip 12x -- with the sums over
// tokens (agent aggregation, dV_agent, dA, conv weight gradients) written as per-chunk partials
// and folded by tiny combine kernels in fixed chunk order (deterministic, no atomics):
//     forward : pool -> s1_partial -> s1_combine -> s2
//     backward: s2_bwd -> mid -> s1_bwd -> pool_bwd
// Every global access is a coalesced b128 with 16 lanes per 256-byte row.  Two thread mappings
// per chunk:
//     phase A  2 threads <-> token: the chunk's q/k/v/dO rows are staged in an LDS tile (row
//                                 stride 68 floats); each thread takes 32 channels of its token
//                                 against the agents (broadcast from LDS), the pair adds up with
//                                 one DPP swap; the p-wide softmaxes; dq / dk rows leave through
//                                 the same tile
//     phase B  16 lanes <-> row  : sums over tokens, the depthwise convolution (its 3x3 halo
//                                 comes from L2), O / dv stores; sums fold over the 4 row
//                                 groups of a wave by DPP and over the 4 waves through LDS
// bias1 / bias2 are scalars added to every score of a softmax row: the softmax is invariant
// to them, so they do not enter the arithmetic and their gradient is exactly 0.
#include "amk_common.h"
#include <stdlib.h>

namespace amk_agent {

constexpr int D = 64;
constexpr int MAXP = 16;
constexpr int CH = 128;      // tokens per chunk
constexpr int NT = 256;      // threads per workgroup: 2 per token in phase A, 16 per row in phase B
constexpr int TS = 68;       // LDS row stride of the staged tile (conflict-free b128 row reads)
constexpr int PSTR = D + 2;  // forward partial record per (chunk, agent): D sums, chunk max, chunk row sum
constexpr int HALF = D / 2;

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
__device__ __forceinline__ float4 f4(float x) { return make_float4(x, x, x, x); }
// Global rows through buffer descriptors with the validity folded into the OFFSET (an invalid row gets an offset past
// the descriptor's range: the load returns zeros, the store is dropped).  `cond ? *ptr : 0` compiles to a branch
// around the load, and behind such a branch every wait drains everything in flight: a tile's eight row requests
// became eight serial round trips (5.5 us to stage 32 KB; tools/scratch/stamps_agent.py).
constexpr unsigned PAST = 0x80000000u;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0);
}
// byte offset of row t (channels c4..) in a (T, D) view with row stride st, or PAST
__device__ __forceinline__ unsigned row_off(int t, int T, int64_t st, int c4) {
  return (t >= 0 && t < T) ? (unsigned)(((int64_t)t * st + c4) * 4) : PAST;
}
__device__ __forceinline__ void fma4(float4& a, float s, const float4& x) {
  a.x += s * x.x; a.y += s * x.y; a.z += s * x.z; a.w += s * x.w;
}
__device__ __forceinline__ void mad4(float4& a, const float4& w, const float4& x) {
  a.x += w.x * x.x; a.y += w.y * x.y; a.z += w.z * x.z; a.w += w.w * x.w;
}
// barrier that orders LDS only: global loads issued before it stay in flight across it (__syncthreads() drains them)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__host__ __device__ __forceinline__ int bin_lo(int i, int T, int P) { return (int)(((int64_t)i * T) / P); }
__host__ __device__ __forceinline__ int bin_hi(int i, int T, int P) { return (int)((((int64_t)(i + 1)) * T + P - 1) / P); }

// Workgroup-cooperative: rows [t0, t0+CH) of a (T, D) view -> tile[CH][TS]; rows >= T are zero.
// 16 lanes per 256-byte row, 16 rows per pass: coalesced b128 loads, conflict-free LDS writes.
__device__ __forceinline__ void stage_rows(float* tile, const float* base, int64_t st, int t0, int T, int tid) {
  const int c4 = (tid & 15) * 4;
  const __amdgpu_buffer_rsrc_t rs = slab(base);
  float4 pc[CH / (NT / 16)];
#pragma unroll
  for (int j = 0; j < CH / (NT / 16); ++j) pc[j] = bld4(rs, row_off(t0 + (NT / 16) * j + (tid >> 4), T, st, c4));   // all in flight
#pragma unroll
  for (int j = 0; j < CH / (NT / 16); ++j) st4(tile + ((NT / 16) * j + (tid >> 4)) * TS + c4, pc[j]);
}
// the same in two halves: request the pieces (registers), put them into the tile later -- the second tile of a kernel
// is in flight while the first one is worked on
constexpr int NPIECE = CH / (NT / 16);
__device__ __forceinline__ void fetch_rows(float4 (&pc)[NPIECE], const float* base, int64_t st, int t0, int T, int tid) {
  const int c4 = (tid & 15) * 4;
  const __amdgpu_buffer_rsrc_t rs = slab(base);
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) pc[j] = bld4(rs, row_off(t0 + (NT / 16) * j + (tid >> 4), T, st, c4));
}
__device__ __forceinline__ void put_rows(float* tile, const float4 (&pc)[NPIECE], int tid) {
  const int c4 = (tid & 15) * 4;
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) st4(tile + ((NT / 16) * j + (tid >> 4)) * TS + c4, pc[j]);
}
// tile[CH][TS] -> rows [t0, min(t0+CH, T)) of a (T, D) view
__device__ __forceinline__ void unstage_rows(const float* tile, float* base, int64_t st, int t0, int T, int tid) {
  const int c4 = (tid & 15) * 4;
  const __amdgpu_buffer_rsrc_t rs = slab(base);
#pragma unroll
  for (int r0 = 0; r0 < CH; r0 += NT / 16) {
    const int r = r0 + (tid >> 4);
    bst4(rs, row_off(t0 + r, T, st, c4), ld4(tile + r * TS + c4));
  }
}
// a value moved inside a row of 16 lanes by a DPP pattern (one VALU modifier; __shfl_xor compiles to ds_bpermute_b32, an
// LDS-crossbar round trip)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// phase A: thread (tok, half) owns channels [32*half, 32*half+32) of token tok
__device__ __forceinline__ void load_half(const float* tile, int tok, int half, float (&r)[HALF]) {
  const float* src = tile + tok * TS + HALF * half;
#pragma unroll
  for (int j = 0; j < HALF / 4; ++j) {
    const float4 t = ld4(src + 4 * j);
    r[4 * j] = t.x; r[4 * j + 1] = t.y; r[4 * j + 2] = t.z; r[4 * j + 3] = t.w;
  }
}
// <row half, a[32*half ...]> summed over the two halves of the token (lanes 2k, 2k+1)
__device__ __forceinline__ float dot_half(const float (&r)[HALF], const float* a, int half) {
  const float* src = a + HALF * half;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < HALF / 4; ++j) {
    const float4 t = ld4(src + 4 * j);
    s += r[4 * j] * t.x + r[4 * j + 1] * t.y + r[4 * j + 2] * t.z + r[4 * j + 3] * t.w;
  }
  return s + dpp_mov<0xB1>(s);   // quad_perm [1, 0, 3, 2]: the other half of the token
}
// sum over the four 16-lane row groups of a wave (phase B: lane bits 4, 5 select the token)
__device__ __forceinline__ float4 fold_subs(float4 a) {
  a.x += __shfl_xor(a.x, 16, 64); a.y += __shfl_xor(a.y, 16, 64); a.z += __shfl_xor(a.z, 16, 64); a.w += __shfl_xor(a.w, 16, 64);
  a.x += __shfl_xor(a.x, 32, 64); a.y += __shfl_xor(a.y, 32, 64); a.z += __shfl_xor(a.z, 32, 64); a.w += __shfl_xor(a.w, 32, 64);
  return a;
}
// sum of the four waves' entries red[(w*MAXP + i)*D + lane]
__device__ __forceinline__ float fold4(const float* red, int i, int lane) {
  return red[(0 * MAXP + i) * D + lane] + red[(1 * MAXP + i) * D + lane] + red[(2 * MAXP + i) * D + lane] +
         red[(3 * MAXP + i) * D + lane];
}

#ifdef AMK_AGENT_STAMPS   // diagnostic build (tools/scratch/stamps_agent.py): per-workgroup time stamps of s2_bwd
__device__ unsigned long long* g_stamps = nullptr;
#define AG_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AG_STAMP(i) do { } while (0)
#endif

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
// One workgroup per (b, h, agent); 16 rows per pass.
__global__ __launch_bounds__(NT) void agent_pool_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) float red[4 * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c4 = (tid & 15) * 4;
  const int i = blockIdx.x % p.P;
  const int bh = blockIdx.x / p.P;
  const int h = bh % p.H, b = bh / p.H;
  const float* qb = p.q + (int64_t)b * p.qs.sb + (int64_t)h * p.qs.sh;
  const int lo = bin_lo(i, p.T, p.P), hi = bin_hi(i, p.T, p.P);
  float4 s = f4(0.f);
  for (int t = lo + (tid >> 4); t < hi; t += NT / 16) {
    const float4 x = ld4(qb + (int64_t)t * p.qs.st + c4);
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  s = fold_subs(s);
  if (lane < 16) st4(&red[wave * D + c4], s);
  __syncthreads();
  if (wave == 0)
    p.agents[((int64_t)bh * p.P + i) * D + lane] =
        (red[lane] + red[D + lane] + red[2 * D + lane] + red[3 * D + lane]) / (float)(hi - lo);
}

// forward 1: per-chunk partial of V_agent = softmax((A*scale) K^T) V: chunk max, chunk row sum and
// the un-normalised sum over the chunk's keys.
template <int PM>
__global__ __launch_bounds__(NT) void agent_s1_partial_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) float As[PM * D];
  __shared__ __attribute__((aligned(16))) float tile[CH * TS];   // k rows, then the cross-wave reduction
  __shared__ float S[PM * CH];
  __shared__ float mloc[PM], lloc[PM];
  float* red = tile;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, t0 = w.t0;
  const float* kb = p.k + (int64_t)w.b * p.ks.sb + (int64_t)w.h * p.ks.sh;
  const float* vb = p.v + (int64_t)w.b * p.vs.sb + (int64_t)w.h * p.vs.sh;

  for (int i = wave; i < P; i += 4) As[i * D + lane] = p.agents[(w.bh * P + i) * D + lane] * p.scale;
  stage_rows(tile, kb, p.ks.st, t0, T, tid);
  // the value rows of phase B (16 lanes per row, 16 rows per pass) are requested NOW and arrive under phase A: the
  // barriers in between order LDS only
  const int sub = tid >> 4, c4 = (tid & 15) * 4;
  float4 vrow[CH / (NT / 16)];
  {
    const __amdgpu_buffer_rsrc_t vrs = slab(vb);
#pragma unroll
    for (int j = 0; j < CH / (NT / 16); ++j) vrow[j] = bld4(vrs, row_off(t0 + (NT / 16) * j + sub, T, p.vs.st, c4));
  }
  lds_barrier();
  {  // phase A: two threads per key
    const int tok = tid >> 1, half = tid & 1;
    float kr[HALF];
    load_half(tile, tok, half, kr);
#pragma unroll
    for (int i = 0; i < PM; ++i) {
      if (i < P) {
        const float s = dot_half(kr, &As[i * D], half);
        if (half == 0) S[i * CH + tok] = (t0 + tok < T) ? s : -INFINITY;
      }
    }
  }
  lds_barrier();
  for (int i = wave; i < P; i += 4) {  // chunk max per agent: one wave per agent
    float m = fmaxf(S[i * CH + lane], S[i * CH + 64 + lane]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) mloc[i] = m;  // finite: every chunk holds at least one key
  }
  lds_barrier();
  for (int e = tid; e < P * CH; e += NT) S[e] = expf(S[e] - mloc[e / CH]);   // (rows past T: exp(-inf) = 0)
  lds_barrier();
  {  // phase B: 16 lanes per value row, 16 rows of the chunk per pass
    float4 acc[PM];
#pragma unroll
    for (int i = 0; i < PM; ++i) acc[i] = f4(0.f);
#pragma unroll
    for (int j = 0; j < CH / (NT / 16); ++j) {
      const int r = (NT / 16) * j + sub;
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) fma4(acc[i], S[i * CH + r], vrow[j]);
    }
#pragma unroll
    for (int i = 0; i < PM; ++i) {
      if (i < P) {
        const float4 a = fold_subs(acc[i]);
        if (lane < 16) st4(&red[(wave * MAXP + i) * D + c4], a);
      }
    }
    for (int i = wave; i < P; i += 4) {  // row sums of this chunk, one wave per agent
      float s = S[i * CH + lane] + S[i * CH + 64 + lane];
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
// One workgroup per (b, h), one wave per agent.  The chunk stats sit one per lane (64 chunks per
// pass) so their loads and exponentials are independent instead of a serial chain.
__global__ __launch_bounds__(256) void agent_s1_combine_kernel(Params p) {
  __shared__ float a_s[4][64], l_s[4][64];  // per wave: rescale factor and row sum of 64 chunks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bh = blockIdx.x;
  for (int i = wave; i < p.P; i += 4) {
    const float* rec0 = p.part + (bh * p.NC * p.P + i) * PSTR;
    const int64_t cstr = (int64_t)p.P * PSTR;
    float M = -INFINITY;
    for (int c = lane; c < p.NC; c += 64) M = fmaxf(M, rec0[c * cstr + D]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
    float L = 0.f, s = 0.f;
    for (int c0 = 0; c0 < p.NC; c0 += 64) {
      const int cmine = c0 + lane;
      a_s[wave][lane] = cmine < p.NC ? expf(rec0[cmine * cstr + D] - M) : 0.f;
      l_s[wave][lane] = cmine < p.NC ? rec0[cmine * cstr + D + 1] : 0.f;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int n = min(64, p.NC - c0);
#pragma unroll 4
      for (int c = 0; c < n; ++c) {  // ascending chunk order; a wave reads only what it wrote
        const float a = a_s[wave][c];
        L += a * l_s[wave][c];
        s += a * rec0[(c0 + c) * cstr + lane];
      }
      __builtin_amdgcn_wave_barrier();
    }
    const int64_t row = bh * p.P + i;
    p.vagent[row * D + lane] = s / L;
    if (lane == 0) { p.stats1[row * 2] = M; p.stats1[row * 2 + 1] = L; }
  }
}

// depthwise 3x3 over the (head, token) plane at (hh, t), channels c4..c4+3, zero padded.
// wq[j] holds weight tap j of the four channels.  FLIP: transposed convolution (backward).
template <bool FLIP>
__device__ __forceinline__ void conv4(float4& acc, const float* xb, const Strides& xs, int H, int T, int hh, int t, int c4,
                                      const float4 (&wq)[9]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int h2 = FLIP ? hh - (a - 1) : hh + (a - 1);
    if (h2 < 0 || h2 >= H) continue;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int t2 = FLIP ? t - (b - 1) : t + (b - 1);
      if (t2 < 0 || t2 >= T) continue;
      mad4(acc, wq[a * 3 + b], ld4(xb + (int64_t)h2 * xs.sh + (int64_t)t2 * xs.st + c4));
    }
  }
}
__device__ __forceinline__ void load_taps(const float* convw, int c4, float4 (&wq)[9]) {
#pragma unroll
  for (int j = 0; j < 9; ++j)
    wq[j] = make_float4(convw[(c4 + 0) * 9 + j], convw[(c4 + 1) * 9 + j], convw[(c4 + 2) * 9 + j], convw[(c4 + 3) * 9 + j]);
}

// forward 3: O = softmax((q*scale) A^T) V_agent + dwc(v) for one chunk of tokens.
template <int PM>
__global__ __launch_bounds__(NT) void agent_s2_kernel(Params p) {
  __shared__ __attribute__((aligned(16))) float As[PM * D];
  __shared__ __attribute__((aligned(16))) float tile[CH * TS];   // q rows
  __shared__ float S[PM * CH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, t0 = w.t0;
  const float* qb = p.q + (int64_t)w.b * p.qs.sb + (int64_t)w.h * p.qs.sh;

  for (int i = wave; i < P; i += 4) As[i * D + lane] = p.agents[(w.bh * P + i) * D + lane] * p.scale;
  stage_rows(tile, qb, p.qs.st, t0, T, tid);
  // phase B mapping: 16 lanes per output row, and a 16-lane group takes CH / 16 CONSECUTIVE rows, so that the 3x3
  // window of the depthwise convolution slides: three new value rows (heads h-1, h, h+1) per output row instead of
  // nine.  The first rows of the window are requested now and arrive under phase A (LDS-only barriers in between).
  constexpr int RPG = CH / (NT / 16);   // rows per group
  const int grp = tid >> 4, c4 = (tid & 15) * 4;
  const __amdgpu_buffer_rsrc_t vrs = slab(p.v + (int64_t)w.b * p.vs.sb);
  auto ldv = [&](int a, int t) {   // value row of head h + a - 1 at token t, channels c4..c4+3; zero padding
    const int h2 = w.h + a - 1;
    return bld4(vrs, (h2 >= 0 && h2 < p.H && t >= 0 && t < T) ? (unsigned)(((int64_t)h2 * p.vs.sh + (int64_t)t * p.vs.st + c4) * 4) : PAST);
  };
  const int tfirst = t0 + RPG * grp;
  float4 win[3][4];   // [head offset][token slot]: tokens t-1, t, t+1 and the prefetched t+2
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    win[a][0] = ldv(a, tfirst - 1); win[a][1] = ldv(a, tfirst); win[a][2] = ldv(a, tfirst + 1); win[a][3] = ldv(a, tfirst + 2);
  }
  lds_barrier();
  {  // phase A: two threads per token: p scores, softmax over the agents
    const int tok = tid >> 1, half = tid & 1;
    float qr[HALF];
    load_half(tile, tok, half, qr);
    float sc[PM];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < PM; ++i)
      if (i < P) { sc[i] = dot_half(qr, &As[i * D], half); m = fmaxf(m, sc[i]); }
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < PM; ++i)
      if (i < P) { sc[i] = expf(sc[i] - m); l += sc[i]; }
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) S[i * CH + tok] = sc[i] / l;
    }
  }
  lds_barrier();
  {  // phase B
    float4 wq[9], var[PM];
    load_taps(p.convw, c4, wq);
    const float4 cb = ld4(p.convb + c4);
#pragma unroll
    for (int i = 0; i < PM; ++i) var[i] = (i < P) ? ld4(p.vagent + (w.bh * P + i) * D + c4) : f4(0.f);
    const __amdgpu_buffer_rsrc_t ors = slab(p.o + (int64_t)w.b * p.os.sb + (int64_t)w.h * p.os.sh);
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
      const int r = RPG * grp + j, t = t0 + r;
      float4 o = cb;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) mad4(o, wq[a * 3 + b], win[a][b]);
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) fma4(o, S[i * CH + r], var[i]);
      bst4(ors, row_off(t, T, p.os.st, c4), o);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        win[a][0] = win[a][1]; win[a][1] = win[a][2]; win[a][2] = win[a][3];
        win[a][3] = ldv(a, (j + 3 <= RPG) ? t + 3 : -1);   // (row t+3 is the last one the group's window needs)
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward 0: stage-2 backward of one chunk: dq (broadcast part), partials of dV_agent, of dA
// (broadcast part) and of the conv weight / bias gradients.
template <int PM>
__global__ __launch_bounds__(NT) void agent_s2_bwd_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float As[PM * D];
  __shared__ __attribute__((aligned(16))) float Vas[PM * D];
  __shared__ __attribute__((aligned(16))) float tile[CH * TS];   // q rows, dO rows, dq rows out, reductions
  __shared__ float S[PM * CH];    // P2 of the chunk
  __shared__ float DS[PM * CH];   // dS2 of the chunk
  float* red = tile;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, t0 = w.t0, h = w.h;
  const float* qb = p.q + (int64_t)w.b * p.qs.sb + (int64_t)h * p.qs.sh;
  const float* gb = p.d_o + (int64_t)w.b * p.dos.sb + (int64_t)h * p.dos.sh;
  float* dqb = p.dq + (int64_t)w.b * p.dqs.sb + (int64_t)h * p.dqs.sh;
  const float* vbatch = p.v + (int64_t)w.b * p.vs.sb;

  AG_STAMP(0);
  for (int i = wave; i < P; i += 4) {
    As[i * D + lane] = p.agents[(w.bh * P + i) * D + lane];
    Vas[i * D + lane] = p.vagent[(w.bh * P + i) * D + lane];
  }
  stage_rows(tile, qb, p.qs.st, t0, T, tid);
  float4 gpc[NPIECE];
  fetch_rows(gpc, gb, p.dos.st, t0, T, tid);   // the dO tile: in flight while the q tile is worked on
  // phase B (below): 16 lanes per row, a 16-lane group takes CH / 16 CONSECUTIVE rows: the 3x3 window of value rows behind
  // the convolution's weight gradient slides (three new rows per token instead of nine), rows two tokens ahead in
  // flight; its first rows are requested here, a whole phase early
  constexpr int RPG = CH / (NT / 16);
  const int grp = tid >> 4, c4 = (tid & 15) * 4;
  const __amdgpu_buffer_rsrc_t vrs = slab(vbatch), grs = slab(gb), qrs = slab(qb);
  auto ldv = [&](int a, int t) {
    const int h2 = h + a - 1;
    return bld4(vrs, (h2 >= 0 && h2 < p.H && t >= 0 && t < T) ? (unsigned)(((int64_t)h2 * p.vs.sh + (int64_t)t * p.vs.st + c4) * 4) : PAST);
  };
  const int tfirst = t0 + RPG * grp;
  float4 win[3][4], gq[2][3];   // value window [head offset][t-1, t, t+1, t+2]; dO / q rows [t, t+1, t+2]
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    win[a][0] = ldv(a, tfirst - 1); win[a][1] = ldv(a, tfirst); win[a][2] = ldv(a, tfirst + 1); win[a][3] = ldv(a, tfirst + 2);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { gq[0][k] = bld4(grs, row_off(tfirst + k, T, p.dos.st, c4)); gq[1][k] = bld4(qrs, row_off(tfirst + k, T, p.qs.st, c4)); }
  lds_barrier();
  AG_STAMP(1);
  {  // phase A: two threads per token
    const int tok = tid >> 1, half = tid & 1;
    const bool ok = t0 + tok < T;
    float sc[PM], dp[PM];
    {
      float qr[HALF];
      load_half(tile, tok, half, qr);
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) sc[i] = dot_half(qr, &As[i * D], half) * p.scale;
    }
    lds_barrier();
    put_rows(tile, gpc, tid);
    lds_barrier();
    AG_STAMP(2);
    {
      float gr[HALF];
      load_half(tile, tok, half, gr);
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) dp[i] = dot_half(gr, &Vas[i * D], half);
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < PM; ++i)
      if (i < P) m = fmaxf(m, sc[i]);
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < PM; ++i)
      if (i < P) { sc[i] = expf(sc[i] - m); l += sc[i]; }
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < PM; ++i)
      if (i < P) { sc[i] /= l; dl += sc[i] * dp[i]; }
#pragma unroll
    for (int i = 0; i < PM; ++i) {
      if (i < P) {
        const float ds = sc[i] * (dp[i] - dl);
        if (half == 0) {
          S[i * CH + tok] = ok ? sc[i] : 0.f;
          DS[i * CH + tok] = ok ? ds : 0.f;
        }
        dp[i] = ds * p.scale;
      }
    }
    // dq_t = scale * sum_i dS2[t,i] A_i: this thread's 32 channels into its own half row of the tile
    float* dst = tile + tok * TS + HALF * half;
#pragma unroll
    for (int j = 0; j < HALF / 4; ++j) {
      float4 o = f4(0.f);
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) fma4(o, dp[i], ld4(&As[i * D + HALF * half + 4 * j]));
      st4(dst + 4 * j, o);
    }
  }
  __syncthreads();
  AG_STAMP(3);
  unstage_rows(tile, dqb, p.dqs.st, t0, T, tid);
  float4 accva[PM], acca[PM], dw9[9], dbs = f4(0.f);
#pragma unroll
  for (int i = 0; i < PM; ++i) { accva[i] = f4(0.f); acca[i] = f4(0.f); }
#pragma unroll
  for (int j = 0; j < 9; ++j) dw9[j] = f4(0.f);
  {
#pragma unroll 1
    for (int j = 0; j < RPG; ++j) {
      const int r = RPG * grp + j, t = t0 + r;
      const float4 g = gq[0][0], qv = gq[1][0];   // (rows past T: zeros, and S / DS are zero there)
#pragma unroll
      for (int i = 0; i < PM; ++i) {
        if (i < P) {
          fma4(accva[i], S[i * CH + r], g);
          fma4(acca[i], DS[i * CH + r], qv);
        }
      }
      dbs.x += g.x; dbs.y += g.y; dbs.z += g.z; dbs.w += g.w;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) mad4(dw9[a * 3 + bb], g, win[a][bb]);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        win[a][0] = win[a][1]; win[a][1] = win[a][2]; win[a][2] = win[a][3];
        win[a][3] = ldv(a, (j + 3 <= RPG) ? t + 3 : -1);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        gq[k][0] = gq[k][1]; gq[k][1] = gq[k][2];
        gq[k][2] = bld4(k == 0 ? grs : qrs, (j + 3 < RPG) ? row_off(t + 3, T, k == 0 ? p.dos.st : p.qs.st, c4) : PAST);
      }
    }
  }
  const int64_t cell = w.bh * p.NC + w.ch;
  AG_STAMP(4);
  __syncthreads();  // the dq rows have left the tile
#pragma unroll
  for (int i = 0; i < PM; ++i) {
    if (i < P) {
      const float4 a = fold_subs(accva[i]);
      if (lane < 16) st4(&red[(wave * MAXP + i) * D + c4], a);
    }
  }
  __syncthreads();
  for (int i = wave; i < P; i += 4) p.pva[(cell * P + i) * D + lane] = fold4(red, i, lane);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PM; ++i) {
    if (i < P) {
      const float4 a = fold_subs(acca[i]);
      if (lane < 16) st4(&red[(wave * MAXP + i) * D + c4], a);
    }
  }
  __syncthreads();
  for (int i = wave; i < P; i += 4) p.pa2[(cell * P + i) * D + lane] = p.scale * fold4(red, i, lane);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const float4 a = fold_subs(dw9[j]);
    if (lane < 16) st4(&red[(wave * MAXP + j) * D + c4], a);
  }
  {
    const float4 a = fold_subs(dbs);
    if (lane < 16) st4(&red[(wave * MAXP + 9) * D + c4], a);
  }
  __syncthreads();
  if (wave == 0) {
    for (int j = 0; j < 9; ++j) p.dconvw_part[(cell * 9 + j) * D + lane] = fold4(red, j, lane);
    p.dconvb_part[cell * D + lane] = fold4(red, 9, lane);
  }
  AG_STAMP(5);
}

// backward 1: fold the stage-2 partials in chunk order: dV_agent, dA (stage 2), delta1 = <dV_agent, V_agent>.
// One workgroup per (b, h).
__global__ __launch_bounds__(256) void agent_mid_kernel(BwdParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bh = blockIdx.x;
  for (int i = wave; i < p.P; i += 4) {
    float sva = 0.f, sa2 = 0.f;
#pragma unroll 4
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
template <int PM>
__global__ __launch_bounds__(NT) void agent_s1_bwd_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float As[PM * D];
  __shared__ __attribute__((aligned(16))) float dVas[PM * D];
  __shared__ __attribute__((aligned(16))) float tile[CH * TS];   // k rows, v rows, dk rows out, reduction
  __shared__ float S[PM * CH];    // P1 of the chunk
  __shared__ float DS[PM * CH];   // dS1 of the chunk
  __shared__ float m1[PM], l1[PM], delta1[PM];
  float* red = tile;

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
  stage_rows(tile, kb, p.ks.st, t0, T, tid);
  float4 vpc[NPIECE];
  fetch_rows(vpc, vb, p.vs.st, t0, T, tid);   // the v tile: in flight while the k tile is worked on
  lds_barrier();
  {  // phase A: two threads per key
    const int tok = tid >> 1, half = tid & 1;
    const bool ok = t0 + tok < T;
    float pr[PM], ds[PM];
    {
      float kr[HALF];
      load_half(tile, tok, half, kr);
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) pr[i] = expf(dot_half(kr, &As[i * D], half) * p.scale - m1[i]) / l1[i];
    }
    lds_barrier();
    put_rows(tile, vpc, tid);
    lds_barrier();
    {
      float vr[HALF];
      load_half(tile, tok, half, vr);
#pragma unroll
      for (int i = 0; i < PM; ++i) {
        if (i < P) {
          const float dp = dot_half(vr, &dVas[i * D], half);
          const float d = pr[i] * (dp - delta1[i]);
          if (half == 0) {
            S[i * CH + tok] = ok ? pr[i] : 0.f;
            DS[i * CH + tok] = ok ? d : 0.f;
          }
          ds[i] = d * p.scale;
        }
      }
    }
    // dk_t = scale * sum_i dS1[i,t] A_i
    float* dst = tile + tok * TS + HALF * half;
#pragma unroll
    for (int j = 0; j < HALF / 4; ++j) {
      float4 o = f4(0.f);
#pragma unroll
      for (int i = 0; i < PM; ++i)
        if (i < P) fma4(o, ds[i], ld4(&As[i * D + HALF * half + 4 * j]));
      st4(dst + 4 * j, o);
    }
  }
  __syncthreads();
  unstage_rows(tile, dkb, p.dks.st, t0, T, tid);
  // phase B: 16 lanes per row, CH / 16 consecutive rows per 16-lane group: the 3x3 window of dO rows behind the transposed
  // depthwise convolution slides (three new rows per token instead of nine)
  constexpr int RPG = CH / (NT / 16);
  const int grp = tid >> 4, c4 = (tid & 15) * 4;
  float4 acca[PM];
  {
    float4 wq[9], dva[PM];
    load_taps(p.convw, c4, wq);
#pragma unroll
    for (int i = 0; i < PM; ++i) { acca[i] = f4(0.f); dva[i] = (i < P) ? ld4(&dVas[i * D + c4]) : f4(0.f); }
    // dv[h, t] += sum_{a, b} w[a][b] dO[h - (a - 1), t - (b - 1)]: window slot s holds dO[., t - 1 + s] of head h + 1 - a
    const __amdgpu_buffer_rsrc_t grs = slab(gbatch), krs = slab(kb), dvrs = slab(dvb);
    auto ldg = [&](int a, int t) {
      const int h2 = h - (a - 1);
      return bld4(grs, (h2 >= 0 && h2 < p.H && t >= 0 && t < T) ? (unsigned)(((int64_t)h2 * p.dos.sh + (int64_t)t * p.dos.st + c4) * 4) : PAST);
    };
    const int tfirst = t0 + RPG * grp;
    float4 win[3][4], krow[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      win[a][0] = ldg(a, tfirst - 1); win[a][1] = ldg(a, tfirst); win[a][2] = ldg(a, tfirst + 1); win[a][3] = ldg(a, tfirst + 2);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) krow[k] = bld4(krs, row_off(tfirst + k, T, p.ks.st, c4));
#pragma unroll 1
    for (int j = 0; j < RPG; ++j) {
      const int r = RPG * grp + j, t = t0 + r;
      const float4 kv = krow[0];
      float4 dvv = f4(0.f);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) mad4(dvv, wq[a * 3 + bb], win[a][2 - bb]);   // token t - (bb - 1) = slot 2 - bb
#pragma unroll
      for (int i = 0; i < PM; ++i) {
        if (i < P) {
          fma4(acca[i], DS[i * CH + r], kv);
          fma4(dvv, S[i * CH + r], dva[i]);
        }
      }
      bst4(dvrs, row_off(t, T, p.dvs.st, c4), dvv);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        win[a][0] = win[a][1]; win[a][1] = win[a][2]; win[a][2] = win[a][3];
        win[a][3] = ldg(a, (j + 3 <= RPG) ? t + 3 : -1);
      }
      krow[0] = krow[1]; krow[1] = krow[2];
      krow[2] = bld4(krs, (j + 3 < RPG) ? row_off(t + 3, T, p.ks.st, c4) : PAST);
    }
  }
  __syncthreads();  // the dk rows have left the tile
#pragma unroll
  for (int i = 0; i < PM; ++i) {
    if (i < P) {
      const float4 a = fold_subs(acca[i]);
      if (lane < 16) st4(&red[(wave * MAXP + i) * D + c4], a);
    }
  }
  __syncthreads();
  const int64_t cell = w.bh * p.NC + w.ch;
  for (int i = wave; i < P; i += 4) p.pa1[(cell * P + i) * D + lane] = p.scale * fold4(red, i, lane);
}

// ---------------------------------------------------------------------------------------
// Streaming forms of the two chunk kernels of the backward (P <= 8 agents per head).  The kernels above stage a
// chunk's rows in LDS and run two thread mappings over them (2 threads <-> token for the p-wide arithmetic, 16 lanes <->
// row for the sums): five barrier-separated phases per workgroup, the agents re-read from LDS for every token, and
// -- every workgroup of the chip being in the same phase at the same time -- HBM idle during the arithmetic.  Here
// ONE mapping serves everything: 16 lanes per row (4 channels each), a 16-lane group takes CH / 16 consecutive rows
// from load to store; the agents' channels of a lane sit in registers, a row's p scores are 4-channel partial dots
// folded over the 16 lanes by four DPP steps, and nothing goes through LDS but the final fold of the per-group sums.
// The convolution's weight / bias gradient moves to the stage-1 kernel, which holds the dO window anyway:
//   dw[a][b] = sum g[h][t] v[h+a-1][t+b-1] = sum v[h'][t'] g[h'-(a-1)][t'-(b-1)]   (h' = h+a-1, t' = t+b-1).
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// sum over the 16 lanes of a row group, every lane ends with it: four DPP moves inside the row (quad swaps, then the
// mirrored half-row and row: after the quad steps all four lanes of a quad agree, so a mirror is as good as an xor) --
// __shfl_xor compiles to ds_bpermute_b32, an LDS-crossbar round trip per step, 48 of them per token row here
__device__ __forceinline__ float fold16(float x) {
  x += dpp_mov<0xB1>(x);    // quad_perm [1, 0, 3, 2]
  x += dpp_mov<0x4E>(x);    // quad_perm [2, 3, 0, 1]
  x += dpp_mov<0x141>(x);   // row_half_mirror
  x += dpp_mov<0x140>(x);   // row_mirror
  return x;
}

template <int PM>
__global__ __launch_bounds__(NT) void agent_s2_bwd_stream_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float red[4 * 2 * PM * D];   // [wave][dV_agent | dA][agent][channel]
  constexpr int RPG = CH / (NT / 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = tid >> 4, c4 = (tid & 15) * 4;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, h = w.h;
  const __amdgpu_buffer_rsrc_t qrs = slab(p.q + (int64_t)w.b * p.qs.sb + (int64_t)h * p.qs.sh);
  const __amdgpu_buffer_rsrc_t grs = slab(p.d_o + (int64_t)w.b * p.dos.sb + (int64_t)h * p.dos.sh);
  const __amdgpu_buffer_rsrc_t dqrs = slab(p.dq + (int64_t)w.b * p.dqs.sb + (int64_t)h * p.dqs.sh);
  const int tfirst = w.t0 + RPG * grp;
  float4 qg[2][4];   // q / dO rows [t, t+1, t+2, t+3]: three rows of requests in flight ahead of the one being worked on
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    qg[0][k] = bld4(qrs, row_off(tfirst + k, T, p.qs.st, c4));
    qg[1][k] = bld4(grs, row_off(tfirst + k, T, p.dos.st, c4));
  }
  float4 A[PM], Va[PM], accva[PM], acca[PM];
#pragma unroll
  for (int i = 0; i < PM; ++i) {
    A[i] = i < P ? ld4(p.agents + (w.bh * P + i) * D + c4) : f4(0.f);
    Va[i] = i < P ? ld4(p.vagent + (w.bh * P + i) * D + c4) : f4(0.f);
    accva[i] = f4(0.f); acca[i] = f4(0.f);
  }
#pragma unroll 1
  for (int j = 0; j < RPG; ++j) {
    const float4 qv = qg[0][0], g = qg[1][0];
    const bool ok = tfirst + j < T;
    float sc[PM], dp[PM];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < PM; ++i) {
      sc[i] = fold16(dot4(qv, A[i])) * p.scale;
      dp[i] = fold16(dot4(g, Va[i]));
      if (i < P) m = fmaxf(m, sc[i]);
    }
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < PM; ++i) { sc[i] = i < P ? expf(sc[i] - m) : 0.f; l += sc[i]; }
    const float il = ok ? 1.f / l : 0.f;   // (rows past the sequence: P2 = 0, dS2 = 0)
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < PM; ++i) { sc[i] *= il; dl += sc[i] * dp[i]; }
    float4 dq = f4(0.f);
#pragma unroll
    for (int i = 0; i < PM; ++i) {
      const float ds = sc[i] * (dp[i] - dl);
      fma4(dq, ds * p.scale, A[i]);
      fma4(accva[i], sc[i], g);
      fma4(acca[i], ds, qv);
    }
    bst4(dqrs, row_off(tfirst + j, T, p.dqs.st, c4), dq);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      qg[k][0] = qg[k][1]; qg[k][1] = qg[k][2]; qg[k][2] = qg[k][3];
      qg[k][3] = bld4(k == 0 ? qrs : grs, (j + 4 < RPG) ? row_off(tfirst + j + 4, T, k == 0 ? p.qs.st : p.dos.st, c4) : PAST);
    }
  }
  // fold the 16 groups: 4 per wave by DPP, the 4 waves through LDS
#pragma unroll
  for (int i = 0; i < PM; ++i) {
    if (i < P) {
      const float4 a = fold_subs(accva[i]), b = fold_subs(acca[i]);
      if (lane < 16) { st4(&red[((wave * 2 + 0) * PM + i) * D + c4], a); st4(&red[((wave * 2 + 1) * PM + i) * D + c4], b); }
    }
  }
  __syncthreads();
  const int64_t cell = w.bh * p.NC + w.ch;
  for (int i = wave; i < P; i += 4) {
    float sva = 0.f, sa = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { sva += red[((ww * 2 + 0) * PM + i) * D + lane]; sa += red[((ww * 2 + 1) * PM + i) * D + lane]; }
    p.pva[(cell * P + i) * D + lane] = sva;
    p.pa2[(cell * P + i) * D + lane] = p.scale * sa;
  }
}

template <int PM>
__global__ __launch_bounds__(NT) void agent_s1_bwd_stream_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float red[4 * (PM + 10) * D];   // [wave][dA (PM) | dconv_w (9) | dconv_b][channel]
  __shared__ __attribute__((aligned(16))) float wqs[9 * D];                // the convolution's taps [tap][channel]
  __shared__ __attribute__((aligned(16))) float As[PM * D], dVas[PM * D];  // agents and dV_agent: 48 registers too many next
                                                                           // to the 36 of the convolution's gradient
  __shared__ float m1[PM], il1[PM], delta1[PM];
  constexpr int RPG = CH / (NT / 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = tid >> 4, c4 = (tid & 15) * 4;
  const Where w = where(p.H, p.NC);
  const int P = p.P, T = p.T, h = w.h;
  const __amdgpu_buffer_rsrc_t krs = slab(p.k + (int64_t)w.b * p.ks.sb + (int64_t)h * p.ks.sh);
  const __amdgpu_buffer_rsrc_t vrs = slab(p.v + (int64_t)w.b * p.vs.sb + (int64_t)h * p.vs.sh);
  const __amdgpu_buffer_rsrc_t grs = slab(p.d_o + (int64_t)w.b * p.dos.sb);
  const __amdgpu_buffer_rsrc_t dkrs = slab(p.dk + (int64_t)w.b * p.dks.sb + (int64_t)h * p.dks.sh);
  const __amdgpu_buffer_rsrc_t dvrs = slab(p.dv + (int64_t)w.b * p.dvs.sb + (int64_t)h * p.dvs.sh);
  const int tfirst = w.t0 + RPG * grp;
  // dO rows of head h - (a - 1) at token t (zero padding): the window of the transposed depthwise convolution
  auto ldg = [&](int a, int t) {
    const int h2 = h - (a - 1);
    return bld4(grs, (h2 >= 0 && h2 < p.H && t >= 0 && t < T) ? (unsigned)(((int64_t)h2 * p.dos.sh + (int64_t)t * p.dos.st + c4) * 4) : PAST);
  };
  float4 win[3][4], kv[2][3];   // dO window [head offset][t-1, t, t+1, t+2]; k / v rows [t, t+1, t+2]
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    win[a][0] = ldg(a, tfirst - 1); win[a][1] = ldg(a, tfirst); win[a][2] = ldg(a, tfirst + 1); win[a][3] = ldg(a, tfirst + 2);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { kv[0][k] = bld4(krs, row_off(tfirst + k, T, p.ks.st, c4)); kv[1][k] = bld4(vrs, row_off(tfirst + k, T, p.vs.st, c4)); }
  if (tid < 9 * 16) {   // taps: wqs[j][c] = convw[c][j]
    const int j = tid / 16, cc = (tid & 15) * 4;
    st4(&wqs[j * D + cc], make_float4(p.convw[(cc + 0) * 9 + j], p.convw[(cc + 1) * 9 + j], p.convw[(cc + 2) * 9 + j], p.convw[(cc + 3) * 9 + j]));
  }
  if (tid < P) {
    const int64_t row = w.bh * P + tid;
    m1[tid] = p.stats1[row * 2]; il1[tid] = 1.f / p.stats1[row * 2 + 1]; delta1[tid] = p.delta1[row];
  }
  for (int i = wave; i < P; i += 4) {
    As[i * D + lane] = p.agents[(w.bh * P + i) * D + lane];
    dVas[i * D + lane] = p.dva[(w.bh * P + i) * D + lane];
  }
  float4 acca[PM], dw9[9], dbs = f4(0.f);
#pragma unroll
  for (int i = 0; i < PM; ++i) acca[i] = f4(0.f);
#pragma unroll
  for (int j = 0; j < 9; ++j) dw9[j] = f4(0.f);
  __syncthreads();
#pragma unroll 1
  for (int j = 0; j < RPG; ++j) {
    const int t = tfirst + j;
    const float4 kr = kv[0][0], vr = kv[1][0];
    const bool ok = t < T;
    float4 dk = f4(0.f), dvv = f4(0.f);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        mad4(dvv, ld4(&wqs[(a * 3 + bb) * D + c4]), win[a][2 - bb]);   // dv[h, t] += w[a][b] dO[h - (a-1), t - (b-1)]
        mad4(dw9[a * 3 + bb], vr, win[a][2 - bb]);                      // (v is zero past the sequence)
      }
    dbs.x += win[1][1].x; dbs.y += win[1][1].y; dbs.z += win[1][1].z; dbs.w += win[1][1].w;   // dO[h, t]
#pragma unroll
    for (int i = 0; i < PM; ++i) {
      if (i < P) {
        const float4 Ai = ld4(&As[i * D + c4]), dVi = ld4(&dVas[i * D + c4]);
        const float s = fold16(dot4(kr, Ai)), dp = fold16(dot4(vr, dVi));
        const float pr = ok ? expf(s * p.scale - m1[i]) * il1[i] : 0.f;
        const float ds = pr * (dp - delta1[i]);
        fma4(dk, ds * p.scale, Ai);
        fma4(dvv, pr, dVi);
        fma4(acca[i], ds, kr);
      }
    }
    bst4(dkrs, row_off(t, T, p.dks.st, c4), dk);
    bst4(dvrs, row_off(t, T, p.dvs.st, c4), dvv);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      win[a][0] = win[a][1]; win[a][1] = win[a][2]; win[a][2] = win[a][3];
      win[a][3] = ldg(a, (j + 3 <= RPG) ? t + 3 : -1);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      kv[k][0] = kv[k][1]; kv[k][1] = kv[k][2];
      kv[k][2] = bld4(k == 0 ? krs : vrs, (j + 3 < RPG) ? row_off(t + 3, T, k == 0 ? p.ks.st : p.vs.st, c4) : PAST);
    }
  }
#pragma unroll
  for (int i = 0; i < PM; ++i) {
    if (i < P) {
      const float4 a = fold_subs(acca[i]);
      if (lane < 16) st4(&red[(wave * (PM + 10) + i) * D + c4], a);
    }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const float4 a = fold_subs(dw9[j]);
    if (lane < 16) st4(&red[(wave * (PM + 10) + PM + j) * D + c4], a);
  }
  {
    const float4 a = fold_subs(dbs);
    if (lane < 16) st4(&red[(wave * (PM + 10) + PM + 9) * D + c4], a);
  }
  __syncthreads();
  const int64_t cell = w.bh * p.NC + w.ch;
  auto fold = [&](int slot) {
    float x = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) x += red[(ww * (PM + 10) + slot) * D + lane];
    return x;
  };
  for (int i = wave; i < P; i += 4) p.pa1[(cell * P + i) * D + lane] = p.scale * fold(i);
  for (int j = wave; j < 10; j += 4) {
    if (j < 9) p.dconvw_part[(cell * 9 + j) * D + lane] = fold(PM + j);
    else p.dconvb_part[cell * D + lane] = fold(PM + 9);
  }
}

// backward 3: dq[b,h,t,:] += sum over the bins i containing t of dA[b,h,i,:] / len(bin i), with
// dA = dA(stage 2) + the chunk partials of dA(stage 1) folded in chunk order.  One workgroup per
// (b, h, 64-token block): dA / len of the bins that meet the block is rebuilt in LDS first.
constexpr int PB = 64;
__global__ __launch_bounds__(NT) void agent_pool_bwd_kernel(BwdParams p) {
  __shared__ __attribute__((aligned(16))) float dA[MAXP * D];
  __shared__ int lo_s[MAXP], hi_s[MAXP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = p.T, P = p.P;
  const int nblk = (T + PB - 1) / PB;
  const int blk = blockIdx.x % nblk;
  const int64_t bh = blockIdx.x / nblk;
  const int h = (int)(bh % p.H), b = (int)(bh / p.H);
  const int t0 = blk * PB, t1 = min(T, t0 + PB);
  for (int i = wave; i < P; i += 4) {
    const int lo = bin_lo(i, T, P), hi = bin_hi(i, T, P);
    if (lane == 0) { lo_s[i] = lo; hi_s[i] = hi; }
    if (lo < t1 && hi > t0) {
      float da = p.da2[(bh * P + i) * D + lane];
      for (int ch = 0; ch < p.NC; ++ch) da += p.pa1[(((bh * p.NC + ch) * P) + i) * D + lane];
      dA[i * D + lane] = da / (float)(hi - lo);
    }
  }
  __syncthreads();
  const int c4 = (tid & 15) * 4;
  float* dqb = p.dq + (int64_t)b * p.dqs.sb + (int64_t)h * p.dqs.sh;
#pragma unroll
  for (int r0 = 0; r0 < PB; r0 += NT / 16) {
    const int t = t0 + r0 + (tid >> 4);
    if (t < T) {
      float4 add = f4(0.f);
      for (int i = 0; i < P; ++i)
        if (t >= lo_s[i] && t < hi_s[i]) fma4(add, 1.f, ld4(&dA[i * D + c4]));
      float* dst = dqb + (int64_t)t * p.dqs.st + c4;
      const float4 cur = ld4(dst);
      st4(dst, make_float4(cur.x + add.x, cur.y + add.y, cur.z + add.z, cur.w + add.w));
    }
  }
}

}  // namespace amk_agent

using namespace amk_agent;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool sok(const Strides& s) { return s.sb % 4 == 0 && s.st % 4 == 0 && s.sh % 4 == 0; }
// the kernels address one batch entry's (h, T, d) slab through a 32-bit byte offset (offsets from 2^31 on mean "past the end")
static bool span_ok(const Strides& s, int H, int T) {
  return s.st >= 0 && s.sh >= 0 && ((int64_t)(H - 1) * s.sh + (int64_t)(T - 1) * s.st + D) * 4 < 0x7ffffffell;
}
static int nchunks(int T) { return (T + CH - 1) / CH; }

// kernels are instantiated for up to 8 and up to 16 agents per head (register arrays sized to that)
#define AMK_AGENT_LAUNCH(KERNEL, P_, ...)                                    \
  do {                                                                       \
    if ((P_) <= 8) hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__);               \
    else hipLaunchKernelGGL(KERNEL<16>, __VA_ARGS__);                        \
  } while (0)

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
  AMK_CHECK_SUPPORTED(span_ok(p.qs, H, T) && span_ok(p.ks, H, T) && span_ok(p.vs, H, T) && span_ok(p.os, H, T),
                      "amk_agent_attn_fwd: one batch entry of a tensor must span less than 2 GiB with non-negative strides");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(agent_pool_kernel, dim3((unsigned)(B * H * P)), dim3(NT), 0, st, p);
  AMK_AGENT_LAUNCH(agent_s1_partial_kernel, P, dim3((unsigned)cells), dim3(NT), 0, st, p);
  hipLaunchKernelGGL(agent_s1_combine_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, p);
  AMK_AGENT_LAUNCH(agent_s2_kernel, P, dim3((unsigned)cells), dim3(NT), 0, st, p);
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
  const int64_t pblk = (int64_t)B * H * ((T + PB - 1) / PB);
  AMK_CHECK_SUPPORTED(pblk < (1ll << 31), "amk_agent_attn_bwd: B*h*T exceeds the grid limit");
  AMK_CHECK_SUPPORTED(span_ok(p.qs, H, T) && span_ok(p.ks, H, T) && span_ok(p.vs, H, T) && span_ok(p.dos, H, T) &&
                          span_ok(p.dqs, H, T) && span_ok(p.dks, H, T) && span_ok(p.dvs, H, T),
                      "amk_agent_attn_bwd: one batch entry of a tensor must span less than 2 GiB with non-negative strides");
  hipStream_t st = static_cast<hipStream_t>(stream);
  static int stream_mode = -1;   // AMK_AGENT_STREAM=0: the LDS-staged chunk kernels also for P <= 8
  if (stream_mode < 0) {
    const char* e = getenv("AMK_AGENT_STREAM");
    stream_mode = e ? atoi(e) : 1;
  }
  const bool streaming = stream_mode && P <= 8;
#define AMK_AGENT_STREAM_LAUNCH(KERNEL)                                                                \
  do {                                                                                                  \
    if (P <= 4) hipLaunchKernelGGL(KERNEL<4>, dim3((unsigned)cells), dim3(NT), 0, st, p);              \
    else if (P <= 6) hipLaunchKernelGGL(KERNEL<6>, dim3((unsigned)cells), dim3(NT), 0, st, p);         \
    else hipLaunchKernelGGL(KERNEL<8>, dim3((unsigned)cells), dim3(NT), 0, st, p);                     \
  } while (0)
  if (streaming) AMK_AGENT_STREAM_LAUNCH(agent_s2_bwd_stream_kernel);
  else AMK_AGENT_LAUNCH(agent_s2_bwd_kernel, P, dim3((unsigned)cells), dim3(NT), 0, st, p);
  hipLaunchKernelGGL(agent_mid_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, p);
  if (streaming) AMK_AGENT_STREAM_LAUNCH(agent_s1_bwd_stream_kernel);
  else AMK_AGENT_LAUNCH(agent_s1_bwd_kernel, P, dim3((unsigned)cells), dim3(NT), 0, st, p);
#undef AMK_AGENT_STREAM_LAUNCH
  hipLaunchKernelGGL(agent_pool_bwd_kernel, dim3((unsigned)pblk), dim3(NT), 0, st, p);
  AMK_CHECK_LAUNCH("amk_agent_attn_bwd");
  return AMK_OK;
}

// dconvw[c][a*3+b] = sum over the rows of dconvw_part[row][a*3+b][c]; dconvb[c] = sum of dconvb_part[row][c]: one launch,
// a fixed order, the weight gradient written in the parameter's own (d, 1, 3, 3) layout -- instead of two library
// reductions and a transposing copy.  A workgroup takes one tap (or the bias) and 16 channels: 64 strided row walks of
// float4s in flight side by side, folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void agent_conv_reduce_kernel(const float* __restrict__ wpart, const float* __restrict__ bpart,
                                                                int64_t rows, float* __restrict__ dconvw, float* __restrict__ dconvb) {
  __shared__ float red[64][17];
  const int r = blockIdx.x;                       // 0..8: weight tap, 9: bias
  const int c4i = threadIdx.x & 3, part = threadIdx.x >> 2, c0 = blockIdx.y * 16;
  const float* src = (r < 9 ? wpart + (int64_t)r * amk_agent::D : bpart) + c0 + c4i * 4;
  const int64_t stride = r < 9 ? 9 * amk_agent::D : amk_agent::D;
  float4 acc = amk_agent::f4(0.f);
  for (int64_t i = part; i < rows; i += 64) {
    const float4 v = amk_agent::ld4(src + i * stride);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[part][c4i * 4 + 0] = acc.x; red[part][c4i * 4 + 1] = acc.y; red[part][c4i * 4 + 2] = acc.z; red[part][c4i * 4 + 3] = acc.w;
  __syncthreads();
  if (threadIdx.x < 16) {
    float t = 0.f;
    for (int j = 0; j < 64; ++j) t += red[j][threadIdx.x];
    const int c = c0 + threadIdx.x;
    if (r < 9) dconvw[c * 9 + r] = t;
    else dconvb[c] = t;
  }
}

extern "C" int amk_agent_conv_grad_reduce(const float* dconvw_part, const float* dconvb_part, int64_t rows, int Dh,
                                          float* dconvw, float* dconvb, void* stream) {
  AMK_CHECK_ARG(dconvw_part && dconvb_part && dconvw && dconvb && rows > 0, "amk_agent_conv_grad_reduce: null pointer or no rows");
  AMK_CHECK_SUPPORTED(Dh == amk_agent::D, "amk_agent_conv_grad_reduce: head dim %d not supported (built for %d)", Dh, amk_agent::D);
  AMK_CHECK_ARG((reinterpret_cast<uintptr_t>(dconvw_part) & 15) == 0 && (reinterpret_cast<uintptr_t>(dconvb_part) & 15) == 0,
                "amk_agent_conv_grad_reduce: partials must be 16-byte aligned");
  hipLaunchKernelGGL(agent_conv_reduce_kernel, dim3(10, amk_agent::D / 16), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dconvw_part, dconvb_part, rows, dconvw, dconvb);
  AMK_CHECK_LAUNCH("amk_agent_conv_grad_reduce");
  return AMK_OK;
}

// diagnostic (not part of the ABI): resident workgroups per CU the runtime computes for the chunk kernels
extern "C" int amk_debug_agent_occupancy(int which) {
  int n = -1;
  hipError_t e;
  if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, amk_agent::agent_s1_partial_kernel<8>, amk_agent::NT, 0);
  else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, amk_agent::agent_s2_kernel<8>, amk_agent::NT, 0);
  else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, amk_agent::agent_s2_bwd_kernel<8>, amk_agent::NT, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, amk_agent::agent_s1_bwd_kernel<8>, amk_agent::NT, 0);
  return e == hipSuccess ? n : -(int)e;
}

#ifdef AMK_AGENT_STAMPS
extern "C" int amk_debug_agent_set_stamps(void* buf) {
  unsigned long long* b = static_cast<unsigned long long*>(buf);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(amk_agent::g_stamps), &b, sizeof(b));
}
#endif
