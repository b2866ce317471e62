// Optimizer step over the flat gradient buckets of amk.dp.GradReducer:
// what trainers/vitgqgan.py:159-163,185-189 of the reference does per phase with four library
// calls -- accelerator.clip_grad_norm_ (a norm per tensor, a norm of norms, a scale per tensor),
// Adam.step (moment + parameter update) and zero_grad -- as two HBM-bound passes:
//
//   sumsq_partials_kernel   one read of the gradients: 256 fixed-range partial sums of squares per
//                           bucket (fixed ranges, fixed in-range order: bitwise reproducible)
//   adam_flat_kernel        every workgroup folds the partials (same order everywhere) into the
//                           global norm and the clip coefficient min(1, max_norm / (norm + 1e-6)),
//                           then per 256-element segment (one parameter each -- GradReducer aligns
//                           parameters to 1 KiB): g *= coef, Adam moments, parameter update, and the
//                           gradient is left ZEROED for the next step: 16 B read (p, g, m, v) and
//                           16 B written (p, 0, m, v) per element.
// Parameters that received no gradient this step (the reference's W_d, bias1 / bias2, ...) are
// skipped exactly as torch optimizers skip .grad None: the per-parameter table says so.
#include "amk_common.h"

namespace amk_opt {

constexpr int NPART = 256;  // partial sums per bucket
constexpr int SEG = 256;    // elements per segment (one wave x float4)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// partials[w] = sum of x[i]^2 over the w-th of NPART equal ranges of whole segments
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ x, int64_t nseg, float* __restrict__ partials) {
  __shared__ float red[4];
  const int64_t per = (nseg + NPART - 1) / NPART;  // segments per workgroup
  const int64_t s0 = (int64_t)blockIdx.x * per, s1 = min(nseg, s0 + per);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int64_t s = s0 + wave; s < s1; s += 4) {
    const float4 v = ld4(x + s * SEG + lane * 4);
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
  float *p, *g, *m, *v;
  int64_t nseg;
  const int32_t* seg_param;  // (nseg) parameter index of each segment
  const float* tab;          // (P, 4): active, lr / bias_correction1, sqrt(bias_correction2), weight-decay factor (decoupled & 2)
  const float* partials;     // (n_partials) squared-norm partials of ALL buckets of this optimizer
  int n_partials;
  float max_norm, beta1, beta2, eps, weight_decay, lr;
  int decoupled;             // bit 0: 0 Adam (L2 term added to the gradient), 1 AdamW (p -= lr * wd * p);
                             // bit 1: the factor comes per parameter from tab[.][3] (wd resp. lr * wd, computed where lr
                             // lives -- on the device for a graph-captured step) instead of the launch arguments
  float* norm_out;           // (1) the global gradient norm before clipping, or null
  __bf16* p16;               // bf16 copy of the parameters for the mixed-precision GEMMs, refreshed here, or null
};

__global__ __launch_bounds__(256) void adam_flat_kernel(AdamArgs a) {
  __shared__ float red[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // global norm: every workgroup folds the same partials in the same order
  float acc = 0.f;
  for (int i = threadIdx.x; i < a.n_partials; i += 256) acc += a.partials[i];
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  float coef = 1.f;
  if (a.max_norm > 0.f) coef = fminf(1.f, a.max_norm / (norm + 1e-6f));
  if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) *a.norm_out = norm;

  const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < a.nseg; s += (int64_t)gridDim.x * 4) {
    const int pid = a.seg_param[s];
    const float4 t = ld4(a.tab + 4 * pid);
    const int64_t off = s * SEG + lane * 4;
    float4 g = ld4(a.g + off);
    st4(a.g + off, make_float4(0.f, 0.f, 0.f, 0.f));
    if (t.x == 0.f) continue;  // no gradient this step: skipped like .grad None
    float4 p = ld4(a.p + off), m = ld4(a.m + off), v = ld4(a.v + off);
    const float step_size = t.y, bc2s = t.z;
    const bool adamw = (a.decoupled & 1) != 0;
    const float wdf = (a.decoupled & 2) ? t.w : (adamw ? a.lr * a.weight_decay : a.weight_decay);
#define AMK_ADAM_ONE(f)                                                   \
    {                                                                     \
      float gg = __fmul_rn(g.f, coef);   /* (rounded as clip_grad_norm_ leaves it, whatever gets fused below) */ \
      float pp = p.f;                                                     \
      if (wdf != 0.f) {                                                   \
        if (adamw) pp -= wdf * pp;                                        \
        else gg = __fmaf_rn(wdf, pp, gg);   /* grad.add(param, alpha=wd) */ \
      }                                                                   \
      const float mm = m.f + omb1 * (gg - m.f);                           \
      const float vv = a.beta2 * v.f + omb2 * gg * gg;                    \
      const float denom = sqrtf(vv) / bc2s + a.eps;                       \
      p.f = pp - step_size * (mm / denom);                                \
      m.f = mm;                                                           \
      v.f = vv;                                                           \
    }
    AMK_ADAM_ONE(x) AMK_ADAM_ONE(y) AMK_ADAM_ONE(z) AMK_ADAM_ONE(w)
#undef AMK_ADAM_ONE
    st4(a.p + off, p);
    if (a.p16) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 h;
      h[0] = (__bf16)p.x; h[1] = (__bf16)p.y; h[2] = (__bf16)p.z; h[3] = (__bf16)p.w;
      *reinterpret_cast<bf16x4*>(a.p16 + off) = h;
    }
    st4(a.m + off, m);
    st4(a.v + off, v);
  }
}

}  // namespace amk_opt

using namespace amk_opt;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int amk_opt_num_partials(void) { return NPART; }

extern "C" int amk_sumsq_partials(const float* x, int64_t n, float* partials, void* stream) {
  AMK_CHECK_ARG(x && partials, "amk_sumsq_partials: null pointer");
  AMK_CHECK_ARG(n > 0 && n % SEG == 0, "amk_sumsq_partials: n = %lld must be a positive multiple of %d", (long long)n, SEG);
  AMK_CHECK_ARG(aligned16(x), "amk_sumsq_partials: x must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(NPART), dim3(256), 0, static_cast<hipStream_t>(stream), x, n / SEG, partials);
  AMK_CHECK_LAUNCH("amk_sumsq_partials");
  return AMK_OK;
}

extern "C" int amk_adam_flat_step_shadow(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                         const int32_t* seg_param, const float* param_tab,
                                         const float* partials, int n_partials,
                                         float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                         int decoupled, float* norm_out, void* param_bf16, void* stream) {
  AMK_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && seg_param && param_tab, "amk_adam_flat_step: null pointer");
  AMK_CHECK_ARG(n > 0 && n % SEG == 0, "amk_adam_flat_step: n = %lld must be a positive multiple of %d", (long long)n, SEG);
  AMK_CHECK_ARG(max_norm <= 0.f || (partials && n_partials > 0), "amk_adam_flat_step: clipping needs the norm partials");
  AMK_CHECK_ARG(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq) && aligned16(param_tab),
                "amk_adam_flat_step: buffers must be 16-byte aligned");
  AdamArgs a;
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq;
  a.nseg = n / SEG;
  a.seg_param = seg_param; a.tab = param_tab;
  a.partials = partials; a.n_partials = partials ? n_partials : 0;
  a.max_norm = max_norm; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.lr = lr;
  a.decoupled = decoupled;
  a.norm_out = norm_out;
  a.p16 = static_cast<__bf16*>(param_bf16);
  AMK_CHECK_ARG((reinterpret_cast<uintptr_t>(param_bf16) & 7) == 0, "amk_adam_flat_step: the bf16 copy must be 8-byte aligned");
  const int64_t nwg = (a.nseg + 3) / 4;
  const unsigned grid = (unsigned)(nwg < 2048 ? nwg : 2048);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  AMK_CHECK_LAUNCH("amk_adam_flat_step");
  return AMK_OK;
}

extern "C" int amk_adam_flat_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const int32_t* seg_param, const float* param_tab,
                                  const float* partials, int n_partials,
                                  float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  int decoupled, float* norm_out, void* stream) {
  return amk_adam_flat_step_shadow(param, grad, exp_avg, exp_avg_sq, n, seg_param, param_tab, partials, n_partials, max_norm, lr, beta1,
                                   beta2, eps, weight_decay, decoupled, norm_out, nullptr, stream);
}
