// One step of the masked-token parallel decode as ONE pass over the logits.
//
// The reference (models/muse.py:211-236, models/maskgit.py:255-272) runs, per step and per position, over
// the (B, T, V = 8192) logits: the classifier-free-guidance combine, a softmax, a top-k (the "top-p"
// filter keeps the ceil((1-p) V) largest logits), a scatter into a -inf tensor, Gumbel noise, a division
// by the temperature, another softmax, an argmax, a gather of the chosen probability and two masked
// writes -- a dozen HBM passes over 33.5 MB per image.  Here a workgroup owns one (b, t) row:
//
//   s_j   = null_j + scale (l_j - null_j)              (or l_j without guidance), kept in LDS
//   m, Z  = max_j s_j, sum_j exp(s_j - m)              (the softmax the score is read from)
//   thr   = the keep-th largest s_j                    (radix select on order-preserving keys, LDS histograms)
//   pred  = argmax_{j : s_j >= thr} (s_j + g_j)        (= argmax softmax((filtered + g) / tau) for tau > 0;
//                                                        first index on ties)
//   score = exp(s_pred - m) / Z
//   ids[row] = pred where mask[row]; scores[row] = score (or `unmasked_score` where not masked: MaskGit)
//
// tau == 0 (the reference's LAST step: temperature = steps_until_x0 / timesteps = 0) divides by zero
// there: every entry of the Gumbel-softmax is NaN and argmax returns 0 -- reproduced: pred = 0.
// g_j is read from `gumbel` when given (parity tests pass torch's own noise), else drawn in-kernel from
// Philox4x32-10 keyed by (seed, offset; row, j / 4): g = -log(-log u) -- the same distribution as torch's
// -log(Exponential(1)), not the same stream (no RNG-stream parity is claimed, as for dropout).
#include "amk_common.h"

namespace amk_sample {

__device__ __forceinline__ uint32_t key_of(float x) {  // order-preserving: larger float -> larger key
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr_hi, uint64_t ctr_lo, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}

__device__ __forceinline__ float gumbel_of(uint32_t bits) {
  const float u = ((float)(bits >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
  return -__logf(-__logf(u));
}

struct Args {
  const float *logits, *null_logits, *gumbel;
  const uint8_t* mask;
  int64_t* ids;
  float* scores;
  uint64_t seed, offset;
  float cfg_scale, tau, unmasked_score;
  int V, keep;
};

__global__ __launch_bounds__(256) void sample_step_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float sv[];  // V scaled logits
  __shared__ float red[8];
  __shared__ int hist[256];
  __shared__ unsigned sel_prefix;
  __shared__ int sel_remaining;
  __shared__ float best_val[4];
  __shared__ int best_idx[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const float* l = a.logits + row * a.V;
  const float* nl = a.null_logits ? a.null_logits + row * a.V : nullptr;

  // ---- scaled logits -> LDS, row max
  float mx = -INFINITY;
  for (int j = tid * 4; j < a.V; j += 1024) {
    float4 x = *reinterpret_cast<const float4*>(l + j);
    if (nl) {
      const float4 n = *reinterpret_cast<const float4*>(nl + j);
      x.x = n.x + a.cfg_scale * (x.x - n.x); x.y = n.y + a.cfg_scale * (x.y - n.y);
      x.z = n.z + a.cfg_scale * (x.z - n.z); x.w = n.w + a.cfg_scale * (x.w - n.w);
    }
    *reinterpret_cast<float4*>(sv + j) = x;
    mx = fmaxf(fmaxf(mx, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  // ---- softmax denominator
  float z = 0.f;
  for (int j = tid; j < a.V; j += 256) z += __expf(sv[j] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
  if (lane == 0) red[4 + wave] = z;
  __syncthreads();
  z = (red[4] + red[5]) + (red[6] + red[7]);

  // ---- the keep-th largest value: radix select, 8 bits per pass from the top
  if (tid == 0) { sel_prefix = 0u; sel_remaining = a.keep; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = sel_prefix;
    const unsigned hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int j = tid; j < a.V; j += 256) {
      const unsigned kx = key_of(sv[j]);
      if ((kx & hi_mask) == prefix) atomicAdd(&hist[(kx >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {  // walk the bins from the largest digit down to the one holding the keep-th value
      int rem = sel_remaining, b = 255;
      for (; b > 0; --b) {
        if (hist[b] >= rem) break;
        rem -= hist[b];
      }
      sel_prefix = prefix | ((unsigned)b << shift);
      sel_remaining = rem;
    }
    __syncthreads();
  }
  const unsigned thr = sel_prefix;  // key of the keep-th largest value

  // ---- argmax over the kept set of (s + g); first index on ties
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  if (a.tau > 0.f) {
    for (int j0 = tid * 4; j0 < a.V; j0 += 1024) {
      float g4[4];
      if (a.gumbel) {
        const float4 g = *reinterpret_cast<const float4*>(a.gumbel + row * a.V + j0);
        g4[0] = g.x; g4[1] = g.y; g4[2] = g.z; g4[3] = g.w;
      } else {
        uint32_t r4[4];
        philox4(a.seed, a.offset, (uint64_t)row * (uint64_t)(a.V / 4) + (uint64_t)(j0 / 4), r4);
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = gumbel_of(r4[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sj = sv[j0 + e];
        if (key_of(sj) >= thr) {
          const float y = sj + g4[e];
          if (y > bv) { bv = y; bi = j0 + e; }  // a thread's indices ascend: strict > keeps the first
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { best_val[wave] = bv; best_idx[wave] = bi; }
    __syncthreads();
  }
  if (tid == 0) {
    int pred = 0;  // tau == 0: the reference's NaN softmax, whose argmax is index 0
    if (a.tau > 0.f) {
      bv = best_val[0]; bi = best_idx[0];
      for (int w = 1; w < 4; ++w)
        if (best_val[w] > bv || (best_val[w] == bv && best_idx[w] < bi)) { bv = best_val[w]; bi = best_idx[w]; }
      pred = bi;
    }
    const bool masked = a.mask ? a.mask[row] != 0 : true;
    if (masked) a.ids[row] = pred;
    float sc = __expf(sv[pred] - mx) / z;
    if (!masked && a.unmasked_score >= 0.f) sc = a.unmasked_score;
    a.scores[row] = sc;
  }
}

}  // namespace amk_sample

using namespace amk_sample;

extern "C" int amk_sample_step(const float* logits, const float* null_logits, float cfg_scale,
                               const float* gumbel, uint64_t seed, uint64_t offset, float tau,
                               int64_t R, int V, int keep, const uint8_t* mask, float unmasked_score,
                               int64_t* ids, float* scores, void* stream) {
  AMK_CHECK_ARG(logits && ids && scores, "amk_sample_step: null pointer");
  AMK_CHECK_ARG(R > 0 && V > 0 && keep > 0 && keep <= V, "amk_sample_step: bad sizes R=%lld V=%d keep=%d", (long long)R, V, keep);
  AMK_CHECK_SUPPORTED(V % 4 == 0 && V <= 36864, "amk_sample_step: V=%d must be a multiple of 4 and at most 36864 (one row in LDS)", V);
  AMK_CHECK_SUPPORTED(R < (1ll << 31), "amk_sample_step: too many rows");
  AMK_CHECK_ARG(((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(null_logits) |
                  reinterpret_cast<uintptr_t>(gumbel)) & 15) == 0, "amk_sample_step: pointers must be 16-byte aligned");
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_step_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4) == hipSuccess;
  AMK_CHECK_SUPPORTED(attr_ok, "amk_sample_step: cannot reserve the LDS row buffer");
  Args a;
  a.logits = logits; a.null_logits = null_logits; a.gumbel = gumbel; a.mask = mask; a.ids = ids; a.scores = scores;
  a.seed = seed; a.offset = offset; a.cfg_scale = cfg_scale; a.tau = tau; a.unmasked_score = unmasked_score;
  a.V = V; a.keep = keep;
  hipLaunchKernelGGL(sample_step_kernel, dim3((unsigned)R), dim3(256), (size_t)V * 4, static_cast<hipStream_t>(stream), a);
  AMK_CHECK_LAUNCH("amk_sample_step");
  return AMK_OK;
}
