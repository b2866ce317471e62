// Fused SwiGLU gate for the ViT-VQGAN feed-forward (epilogues around
// the hot path).  The FFN is w3(silu(a) * b) with (a | b) = w12(x) (the SwiGLU the reference's
// FeedForward keywords describe, models/vitvqgan.py:20-34).  Eager PyTorch runs it as strided
// chunk views -> silu -> mul (and chunk-backward = cat) : six HBM passes forward + backward more;
// here one pass each, 16 B per lane, reading the (M, 2H) GEMM output in place.
// The GEGLU gate of the transformer FFN (models/transformer.py:22-27: `gate * gelu(val)` with
// val, gate = chunk(2), exact erf GELU) is the same kernel with another activation.
#include "amk_common.h"

namespace amk_ew {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const float* __restrict__ ab, int64_t M, int H,
                                                         float* __restrict__ out) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / hv;
    const int c = (int)(i % hv) * 4;
    const float4 a = ld4(ab + row * 2 * H + c);
    const float4 b = ld4(ab + row * 2 * H + H + c);
    float4 o;
    o.x = a.x * sigmoidf_(a.x) * b.x;
    o.y = a.y * sigmoidf_(a.y) * b.y;
    o.z = a.z * sigmoidf_(a.z) * b.z;
    o.w = a.w * sigmoidf_(a.w) * b.w;
    st4(out + row * H + c, o);
  }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const float* __restrict__ ab, const float* __restrict__ d_out,
                                                         int64_t M, int H, float* __restrict__ d_ab) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / hv;
    const int c = (int)(i % hv) * 4;
    const float4 a = ld4(ab + row * 2 * H + c);
    const float4 b = ld4(ab + row * 2 * H + H + c);
    const float4 g = ld4(d_out + row * H + c);
    float4 da, db;
#define AMK_SWIGLU_ONE(f)                                   \
    {                                                       \
      const float s = sigmoidf_(a.f);                       \
      da.f = g.f * b.f * (s * (1.f + a.f * (1.f - s)));     \
      db.f = g.f * (a.f * s);                               \
    }
    AMK_SWIGLU_ONE(x) AMK_SWIGLU_ONE(y) AMK_SWIGLU_ONE(z) AMK_SWIGLU_ONE(w)
#undef AMK_SWIGLU_ONE
    st4(d_ab + row * 2 * H + c, da);
    st4(d_ab + row * 2 * H + H + c, db);
  }
}

__device__ __forceinline__ float gelu_(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
// d/dx gelu(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad_(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
}

__global__ __launch_bounds__(256) void geglu_fwd_kernel(const float* __restrict__ ab, int64_t M, int H,
                                                        float* __restrict__ out) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / hv;
    const int c = (int)(i % hv) * 4;
    const float4 a = ld4(ab + row * 2 * H + c);
    const float4 b = ld4(ab + row * 2 * H + H + c);
    st4(out + row * H + c, make_float4(gelu_(a.x) * b.x, gelu_(a.y) * b.y, gelu_(a.z) * b.z, gelu_(a.w) * b.w));
  }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ ab, const float* __restrict__ d_out,
                                                        int64_t M, int H, float* __restrict__ d_ab) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / hv;
    const int c = (int)(i % hv) * 4;
    const float4 a = ld4(ab + row * 2 * H + c);
    const float4 b = ld4(ab + row * 2 * H + H + c);
    const float4 g = ld4(d_out + row * H + c);
    st4(d_ab + row * 2 * H + c, make_float4(g.x * b.x * gelu_grad_(a.x), g.y * b.y * gelu_grad_(a.y),
                                            g.z * b.z * gelu_grad_(a.z), g.w * b.w * gelu_grad_(a.w)));
    st4(d_ab + row * 2 * H + H + c, make_float4(g.x * gelu_(a.x), g.y * gelu_(a.y), g.z * gelu_(a.z), g.w * gelu_(a.w)));
  }
}

}  // namespace amk_ew

using namespace amk_ew;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static unsigned grid_for(int64_t items) {
  const int64_t blocks = (items + 255) / 256;
  return (unsigned)(blocks < 8192 ? (blocks > 0 ? blocks : 1) : 8192);  // grid-stride beyond 32 blocks per CU
}

extern "C" int amk_swiglu_fwd(const float* ab, int64_t M, int H, float* out, void* stream) {
  AMK_CHECK_ARG(ab && out, "amk_swiglu_fwd: null pointer");
  AMK_CHECK_ARG(M > 0 && H > 0, "amk_swiglu_fwd: non-positive size");
  AMK_CHECK_SUPPORTED(H % 4 == 0 && a16(ab) && a16(out), "amk_swiglu_fwd: H %% 4 == 0 and 16-byte aligned pointers required");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), ab, M, H, out);
  AMK_CHECK_LAUNCH("amk_swiglu_fwd");
  return AMK_OK;
}

extern "C" int amk_swiglu_bwd(const float* ab, const float* d_out, int64_t M, int H, float* d_ab, void* stream) {
  AMK_CHECK_ARG(ab && d_out && d_ab, "amk_swiglu_bwd: null pointer");
  AMK_CHECK_ARG(M > 0 && H > 0, "amk_swiglu_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(H % 4 == 0 && a16(ab) && a16(d_out) && a16(d_ab), "amk_swiglu_bwd: H %% 4 == 0 and 16-byte aligned pointers required");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), ab, d_out, M, H, d_ab);
  AMK_CHECK_LAUNCH("amk_swiglu_bwd");
  return AMK_OK;
}

extern "C" int amk_geglu_fwd(const float* ab, int64_t M, int H, float* out, void* stream) {
  AMK_CHECK_ARG(ab && out, "amk_geglu_fwd: null pointer");
  AMK_CHECK_ARG(M > 0 && H > 0, "amk_geglu_fwd: non-positive size");
  AMK_CHECK_SUPPORTED(H % 4 == 0 && a16(ab) && a16(out), "amk_geglu_fwd: H %% 4 == 0 and 16-byte aligned pointers required");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), ab, M, H, out);
  AMK_CHECK_LAUNCH("amk_geglu_fwd");
  return AMK_OK;
}

extern "C" int amk_geglu_bwd(const float* ab, const float* d_out, int64_t M, int H, float* d_ab, void* stream) {
  AMK_CHECK_ARG(ab && d_out && d_ab, "amk_geglu_bwd: null pointer");
  AMK_CHECK_ARG(M > 0 && H > 0, "amk_geglu_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(H % 4 == 0 && a16(ab) && a16(d_out) && a16(d_ab), "amk_geglu_bwd: H %% 4 == 0 and 16-byte aligned pointers required");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), ab, d_out, M, H, d_ab);
  AMK_CHECK_LAUNCH("amk_geglu_bwd");
  return AMK_OK;
}
