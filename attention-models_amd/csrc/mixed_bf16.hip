// Element-wise kernels of the mixed-precision (bf16 autocast) mode: the passes between the bf16 GEMMs of a pre-LN
// block, reading and writing bf16 where the neighbouring GEMM wants bf16, so that no separate cast kernel runs.
// Reference: the same ops as layernorm.hip / elementwise.hip (models/vitvqgan.py:20-61) under the reference's shipped
// precision (cfg/vitvqgan.yaml:73: accelerate bf16 autocast -- nn.Linear in bf16, LayerNorm in f32).
//   swiglu_bf16_fwd / _bwd   (a | b) bf16 -> silu(a) * b bf16; arithmetic in f32.  Replaces, per FFN and direction,
//                            an upcast copy, the f32 gate kernel and a downcast copy: 1.3 GB -> 0.27 GB (forward)
//                            and 1.7 GB -> 0.45 GB (backward) of HBM traffic at batch 32.
//   add_layernorm_mixed_fwd  h = x (+ res) with x f32 or bf16 (a branch output) and res f32 (the residual stream);
//                            h f32, y = LN(h) * gamma + beta in bf16 (its consumers are bf16 GEMMs), mean / rstd f32
//   add_layernorm_mixed_bwd  dy bf16 or f32, h f32 -> dh f32 (+ dh_in), and a bf16 copy of dh for the bf16 branch
// One wave per row, 4 elements per lane and chunk, two-pass statistics -- the structure of layernorm.hip.
#include "amk_common.h"

namespace amk_mixed {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ldb4(const __bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void stb4(__bf16* p, float4 v) {
  bf16x4 w;
  w[0] = (__bf16)v.x; w[1] = (__bf16)v.y; w[2] = (__bf16)v.z; w[3] = (__bf16)v.w;
  *reinterpret_cast<bf16x4*>(p) = w;
}
__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void swiglu_bf16_fwd_kernel(const __bf16* __restrict__ ab, int64_t M, int H, __bf16* __restrict__ out) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / hv;
    const int c = (int)(i % hv) * 4;
    const float4 a = ldb4(ab + row * 2 * H + c), b = ldb4(ab + row * 2 * H + H + c);
    stb4(out + row * H + c, make_float4(a.x * sigmoidf_(a.x) * b.x, a.y * sigmoidf_(a.y) * b.y, a.z * sigmoidf_(a.z) * b.z,
                                        a.w * sigmoidf_(a.w) * b.w));
  }
}

__global__ __launch_bounds__(256) void swiglu_bf16_bwd_kernel(const __bf16* __restrict__ ab, const __bf16* __restrict__ d_out, int64_t M, int H,
                                                              __bf16* __restrict__ d_ab) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / hv;
    const int c = (int)(i % hv) * 4;
    const float4 a = ldb4(ab + row * 2 * H + c), b = ldb4(ab + row * 2 * H + H + c), g = ldb4(d_out + row * H + c);
    float4 da, db;
#define AMK_ONE(f)                                          \
    {                                                       \
      const float s = sigmoidf_(a.f);                       \
      da.f = g.f * b.f * (s * (1.f + a.f * (1.f - s)));     \
      db.f = g.f * (a.f * s);                               \
    }
    AMK_ONE(x) AMK_ONE(y) AMK_ONE(z) AMK_ONE(w)
#undef AMK_ONE
    stb4(d_ab + row * 2 * H + c, da);
    stb4(d_ab + row * 2 * H + H + c, db);
  }
}

constexpr int WAVES = 4;

// XB: x is bf16; HAS_RES: h = x + res
template <int NCH, bool XB, bool HAS_RES>
__global__ __launch_bounds__(64 * WAVES) void ln_mixed_fwd_kernel(const void* __restrict__ xv, const float* __restrict__ res,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int64_t M, int D, float eps, float* __restrict__ h,
                                                                  __bf16* __restrict__ y, float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  float4 g[NCH], b[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
    g[j] = c < nch ? ld4(gamma + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    b[j] = c < nch ? ld4(beta + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_d = 1.f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < M; row += (int64_t)gridDim.x * WAVES) {
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < nch) {
        v[j] = XB ? ldb4(static_cast<const __bf16*>(xv) + row * D + 4 * c) : ld4(static_cast<const float*>(xv) + row * D + 4 * c);
        if (HAS_RES) {
          const float4 r = ld4(res + row * D + 4 * c);
          v[j].x += r.x; v[j].y += r.y; v[j].z += r.z; v[j].w += r.w;
        }
        if (h) st4(h + row * D + 4 * c, v[j]);
        s += v[j].x + v[j].y + v[j].z + v[j].w;
      }
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch)
        stb4(y + row * D + 4 * c, make_float4(v[j].x * rstd * g[j].x + b[j].x, v[j].y * rstd * g[j].y + b[j].y,
                                              v[j].z * rstd * g[j].z + b[j].z, v[j].w * rstd * g[j].w + b[j].w));
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

// DB: dy is bf16.  dh = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dh_in); dh16: bf16 copy of dh or null
template <int NCH, bool DB, bool HAS_DH>
__global__ __launch_bounds__(64 * WAVES) void ln_mixed_bwd_kernel(const void* __restrict__ dyv, const float* __restrict__ h,
                                                                  const float* __restrict__ dh_in, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                  int64_t M, int D, float* __restrict__ dh, __bf16* __restrict__ dh16,
                                                                  float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // (WAVES, D), used for dgamma then dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  float4 g[NCH], dg[NCH], db[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
    g[j] = c < nch ? ld4(gamma + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_d = 1.f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < M; row += (int64_t)gridDim.x * WAVES) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 gy[NCH], xh[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      gy[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < nch) {
        const float4 d = DB ? ldb4(static_cast<const __bf16*>(dyv) + row * D + 4 * c) : ld4(static_cast<const float*>(dyv) + row * D + 4 * c);
        const float4 hv = ld4(h + row * D + 4 * c);
        xh[j] = make_float4((hv.x - mean) * rstd, (hv.y - mean) * rstd, (hv.z - mean) * rstd, (hv.w - mean) * rstd);
        gy[j] = make_float4(d.x * g[j].x, d.y * g[j].y, d.z * g[j].z, d.w * g[j].w);
        dg[j].x += d.x * xh[j].x; dg[j].y += d.y * xh[j].y; dg[j].z += d.z * xh[j].z; dg[j].w += d.w * xh[j].w;
        db[j].x += d.x; db[j].y += d.y; db[j].z += d.z; db[j].w += d.w;
        s1 += gy[j].x + gy[j].y + gy[j].z + gy[j].w;
        s2 += gy[j].x * xh[j].x + gy[j].y * xh[j].y + gy[j].z * xh[j].z + gy[j].w * xh[j].w;
      }
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float4 o = make_float4(rstd * (gy[j].x - c1 - xh[j].x * c2), rstd * (gy[j].y - c1 - xh[j].y * c2),
                               rstd * (gy[j].z - c1 - xh[j].z * c2), rstd * (gy[j].w - c1 - xh[j].w * c2));
        if (HAS_DH) {
          const float4 a = ld4(dh_in + row * D + 4 * c);
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        st4(dh + row * D + 4 * c, o);
        if (dh16) stb4(dh16 + row * D + 4 * c, o);
      }
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) st4(&red[wave * D + 4 * c], pass == 0 ? dg[j] : db[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 64 * WAVES) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) s += red[w * D + i];
      part[((int64_t)blockIdx.x * 2 + pass) * D + i] = s;
    }
    __syncthreads();
  }
}

}  // namespace amk_mixed

using namespace amk_mixed;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool a8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
static unsigned grid_for(int64_t items) {
  const int64_t blocks = (items + 255) / 256;
  return (unsigned)(blocks < 8192 ? (blocks > 0 ? blocks : 1) : 8192);
}

extern "C" int amk_swiglu_bf16_fwd(const void* ab, int64_t M, int H, void* out, void* stream) {
  AMK_CHECK_ARG(ab && out, "amk_swiglu_bf16_fwd: null pointer");
  AMK_CHECK_ARG(M > 0 && H > 0, "amk_swiglu_bf16_fwd: non-positive size");
  AMK_CHECK_SUPPORTED(H % 4 == 0 && a8(ab) && a8(out), "amk_swiglu_bf16_fwd: H %% 4 == 0 and 8-byte aligned pointers required");
  hipLaunchKernelGGL(swiglu_bf16_fwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const __bf16*>(ab), M, H, static_cast<__bf16*>(out));
  AMK_CHECK_LAUNCH("amk_swiglu_bf16_fwd");
  return AMK_OK;
}

extern "C" int amk_swiglu_bf16_bwd(const void* ab, const void* d_out, int64_t M, int H, void* d_ab, void* stream) {
  AMK_CHECK_ARG(ab && d_out && d_ab, "amk_swiglu_bf16_bwd: null pointer");
  AMK_CHECK_ARG(M > 0 && H > 0, "amk_swiglu_bf16_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(H % 4 == 0 && a8(ab) && a8(d_out) && a8(d_ab), "amk_swiglu_bf16_bwd: H %% 4 == 0 and 8-byte aligned pointers required");
  hipLaunchKernelGGL(swiglu_bf16_bwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const __bf16*>(ab), static_cast<const __bf16*>(d_out), M, H, static_cast<__bf16*>(d_ab));
  AMK_CHECK_LAUNCH("amk_swiglu_bf16_bwd");
  return AMK_OK;
}

static int parts_for(int64_t M) {
  const int64_t wg = (M + WAVES - 1) / WAVES;
  return (int)(wg < 2048 ? (wg > 0 ? wg : 1) : 2048);
}

#define AMK_MX_DISPATCH(D_, CALL)                     \
  do {                                                \
    if ((D_) <= 256) { CALL(1); }                     \
    else if ((D_) <= 512) { CALL(2); }                \
    else if ((D_) <= 1024) { CALL(4); }               \
    else if ((D_) <= 2048) { CALL(8); }               \
    else { CALL(16); }                                \
  } while (0)

extern "C" int amk_add_layernorm_mixed_fwd(const void* x, int x_is_bf16, const float* res, const float* gamma, const float* beta,
                                           int64_t M, int D, float eps, float* h, void* y_bf16, float* mean, float* rstd, void* stream) {
  AMK_CHECK_ARG(x && gamma && beta && y_bf16 && mean && rstd, "amk_add_layernorm_mixed_fwd: null pointer");
  AMK_CHECK_ARG(!res || h, "amk_add_layernorm_mixed_fwd: a residual needs the h output");
  AMK_CHECK_ARG(M > 0 && D > 0, "amk_add_layernorm_mixed_fwd: non-positive size");
  AMK_CHECK_SUPPORTED(D % 4 == 0 && D <= 4096, "amk_add_layernorm_mixed_fwd: width %d not supported (multiple of 4, <= 4096)", D);
  AMK_CHECK_ARG(a8(x) && (x_is_bf16 || a16(x)) && a16(res) && a16(gamma) && a16(beta) && a16(h) && a8(y_bf16),
                "amk_add_layernorm_mixed_fwd: misaligned pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t wg = (M + WAVES - 1) / WAVES;
  const dim3 grid((unsigned)(wg < 16384 ? wg : 16384)), block(64 * WAVES);
  __bf16* y = static_cast<__bf16*>(y_bf16);
#define CALL(NCH)                                                                                                                        \
  if (x_is_bf16 && res) hipLaunchKernelGGL((ln_mixed_fwd_kernel<NCH, true, true>), grid, block, 0, st, x, res, gamma, beta, M, D, eps, h, y, mean, rstd); \
  else if (x_is_bf16) hipLaunchKernelGGL((ln_mixed_fwd_kernel<NCH, true, false>), grid, block, 0, st, x, res, gamma, beta, M, D, eps, h, y, mean, rstd);  \
  else if (res) hipLaunchKernelGGL((ln_mixed_fwd_kernel<NCH, false, true>), grid, block, 0, st, x, res, gamma, beta, M, D, eps, h, y, mean, rstd);        \
  else hipLaunchKernelGGL((ln_mixed_fwd_kernel<NCH, false, false>), grid, block, 0, st, x, res, gamma, beta, M, D, eps, h, y, mean, rstd)
  AMK_MX_DISPATCH(D, CALL);
#undef CALL
  AMK_CHECK_LAUNCH("amk_add_layernorm_mixed_fwd");
  return AMK_OK;
}

extern "C" int amk_add_layernorm_mixed_bwd(const void* dy, int dy_is_bf16, const float* h, const float* dh_in, const float* gamma,
                                           const float* mean, const float* rstd, int64_t M, int D, float* dh, void* dh_bf16,
                                           float* dgb_part, void* stream) {
  AMK_CHECK_ARG(dy && h && gamma && mean && rstd && dh && dgb_part, "amk_add_layernorm_mixed_bwd: null pointer");
  AMK_CHECK_ARG(M > 0 && D > 0, "amk_add_layernorm_mixed_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(D % 4 == 0 && D <= 4096, "amk_add_layernorm_mixed_bwd: width %d not supported (multiple of 4, <= 4096)", D);
  AMK_CHECK_ARG(a8(dy) && (dy_is_bf16 || a16(dy)) && a16(h) && a16(dh_in) && a16(gamma) && a16(dh) && a8(dh_bf16) && a16(dgb_part),
                "amk_add_layernorm_mixed_bwd: misaligned pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(parts_for(M)), block(64 * WAVES);
  const size_t lds = (size_t)WAVES * D * sizeof(float);
  __bf16* d16 = static_cast<__bf16*>(dh_bf16);
#define CALL(NCH)                                                                                                                                   \
  if (dy_is_bf16 && dh_in) hipLaunchKernelGGL((ln_mixed_bwd_kernel<NCH, true, true>), grid, block, lds, st, dy, h, dh_in, gamma, mean, rstd, M, D, dh, d16, dgb_part); \
  else if (dy_is_bf16) hipLaunchKernelGGL((ln_mixed_bwd_kernel<NCH, true, false>), grid, block, lds, st, dy, h, dh_in, gamma, mean, rstd, M, D, dh, d16, dgb_part);   \
  else if (dh_in) hipLaunchKernelGGL((ln_mixed_bwd_kernel<NCH, false, true>), grid, block, lds, st, dy, h, dh_in, gamma, mean, rstd, M, D, dh, d16, dgb_part);         \
  else hipLaunchKernelGGL((ln_mixed_bwd_kernel<NCH, false, false>), grid, block, lds, st, dy, h, dh_in, gamma, mean, rstd, M, D, dh, d16, dgb_part)
  AMK_MX_DISPATCH(D, CALL);
#undef CALL
  AMK_CHECK_LAUNCH("amk_add_layernorm_mixed_bwd");
  return AMK_OK;
}
