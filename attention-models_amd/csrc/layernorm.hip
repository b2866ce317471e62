// Residual-add + LayerNorm, forward and backward, and a column-sum (bias gradient) for gfx950.
//
// SURVEY.md section 8f rank 1 (epilogues around the hot path): the pre-LN blocks of the reference
// (models/vitvqgan.py:44-61, models/transformer.py:11-19,58-76) run `x = x + branch(LN(x))` as
// separate HBM passes -- add, LayerNorm (34 us for 32768 x 256 in eager PyTorch = 2 TB/s), and in
// the backward LayerNorm-backward (67 us) + a gamma/beta partial-sum kernel (31 us) + the
// residual gradient add.  Here:
//   add_layernorm_fwd : h = x (+ res);  y = LN(h) * gamma + beta     one pass: reads x, res; writes h, y
//   add_layernorm_bwd : dh = LN'(dy) (+ dh_in);  partial dgamma / dbeta  one pass: reads dy, h, dh_in; writes dh
//   colsum            : per-column sums of a row-major matrix (the bias gradient of a Linear)
// One wave per row; a lane keeps its 16-byte chunks of the row in registers between the two
// reductions (mean, then centred variance -- the two-pass form), so every byte moves once.
// All HBM-bound: the roofline is bytes / 8 TB/s.
#include "amk_common.h"

namespace amk_ln {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

constexpr int WAVES = 4;  // rows in flight per workgroup

// NCH = 16-byte chunks per lane: D <= 256 * NCH
template <int NCH, bool HAS_RES>
__global__ __launch_bounds__(64 * WAVES) void add_layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, int64_t M, int D, float eps, float* __restrict__ h, float* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out) {
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
        v[j] = ld4(x + row * D + 4 * c);
        if (HAS_RES) {
          const float4 r = ld4(res + row * D + 4 * c);
          v[j].x += r.x; v[j].y += r.y; v[j].z += r.z; v[j].w += r.w;
          st4(h + row * D + 4 * c, v[j]);
        }
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
        st4(y + row * D + 4 * c, make_float4(v[j].x * rstd * g[j].x + b[j].x, v[j].y * rstd * g[j].y + b[j].y,
                                             v[j].z * rstd * g[j].z + b[j].z, v[j].w * rstd * g[j].w + b[j].w));
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}

// dh = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dh_in), g = dy * gamma, xhat = (h - mean) * rstd;
// per-workgroup partial sums of dgamma = dy * xhat and dbeta = dy into part (gridDim.x, 2, D).
template <int NCH, bool HAS_DH>
__global__ __launch_bounds__(64 * WAVES) void add_layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ h, const float* __restrict__ dh_in,
    const float* __restrict__ gamma, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    int64_t M, int D, float* __restrict__ dh, float* __restrict__ part) {
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
        const float4 d = ld4(dy + row * D + 4 * c);
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
      }
    }
  }
  // fold the four waves' column sums, one partial row pair per workgroup
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

// per-column sums of x (M, N): lanes take 16-byte column chunks, waves and workgroups stride the rows;
// part (gridDim.x, N) holds one partial row per workgroup.
template <int NCH>
__global__ __launch_bounds__(64 * WAVES) void colsum_kernel(const float* __restrict__ x, int64_t M, int N,
                                                            float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // (WAVES, N)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = N >> 2;
  float4 acc[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < M; row += (int64_t)gridDim.x * WAVES) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        const float4 v = ld4(x + row * N + 4 * c);
        acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) st4(&red[wave * N + 4 * c], acc[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 64 * WAVES) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += red[w * N + i];
    part[(int64_t)blockIdx.x * N + i] = s;
  }
}

constexpr int MAX_D = 4096;       // 16 chunks per lane
constexpr int MAX_PARTS = 2048;   // partial rows = workgroups of the reducing kernels (8 per CU: 32 rows in flight per CU)

}  // namespace amk_ln

using namespace amk_ln;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static int parts_for(int64_t M) {
  const int64_t wg = (M + WAVES - 1) / WAVES;
  return (int)(wg < MAX_PARTS ? (wg > 0 ? wg : 1) : MAX_PARTS);
}
static unsigned fwd_grid(int64_t M) {
  const int64_t wg = (M + WAVES - 1) / WAVES;
  return (unsigned)(wg < 16384 ? (wg > 0 ? wg : 1) : 16384);
}

extern "C" int amk_rowsum_num_partials(int64_t M) { return M > 0 ? parts_for(M) : 0; }

#define AMK_LN_DISPATCH(D_, CALL)                     \
  do {                                                \
    if ((D_) <= 256) { CALL(1); }                     \
    else if ((D_) <= 512) { CALL(2); }                \
    else if ((D_) <= 1024) { CALL(4); }               \
    else if ((D_) <= 2048) { CALL(8); }               \
    else { CALL(16); }                                \
  } while (0)

extern "C" int amk_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                                     int64_t M, int D, float eps, float* h, float* y, float* mean, float* rstd,
                                     void* stream) {
  AMK_CHECK_ARG(x && gamma && beta && y && mean && rstd, "amk_add_layernorm_fwd: null pointer");
  AMK_CHECK_ARG((res == nullptr) == (h == nullptr), "amk_add_layernorm_fwd: res and h go together");
  AMK_CHECK_ARG(M > 0 && D > 0, "amk_add_layernorm_fwd: non-positive size");
  AMK_CHECK_SUPPORTED(D % 4 == 0 && D <= MAX_D, "amk_add_layernorm_fwd: width %d not supported (multiple of 4, <= %d)", D, MAX_D);
  AMK_CHECK_ARG(a16(x) && a16(res) && a16(gamma) && a16(beta) && a16(h) && a16(y), "amk_add_layernorm_fwd: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(fwd_grid(M)), block(64 * WAVES);
#define CALL(NCH)                                                                                                     \
  if (res) hipLaunchKernelGGL((add_layernorm_fwd_kernel<NCH, true>), grid, block, 0, st, x, res, gamma, beta, M, D, eps, h, y, mean, rstd); \
  else hipLaunchKernelGGL((add_layernorm_fwd_kernel<NCH, false>), grid, block, 0, st, x, res, gamma, beta, M, D, eps, h, y, mean, rstd)
  AMK_LN_DISPATCH(D, CALL);
#undef CALL
  AMK_CHECK_LAUNCH("amk_add_layernorm_fwd");
  return AMK_OK;
}

extern "C" int amk_add_layernorm_bwd(const float* dy, const float* h, const float* dh_in, const float* gamma,
                                     const float* mean, const float* rstd, int64_t M, int D, float* dh,
                                     float* dgb_part, void* stream) {
  AMK_CHECK_ARG(dy && h && gamma && mean && rstd && dh && dgb_part, "amk_add_layernorm_bwd: null pointer");
  AMK_CHECK_ARG(M > 0 && D > 0, "amk_add_layernorm_bwd: non-positive size");
  AMK_CHECK_SUPPORTED(D % 4 == 0 && D <= MAX_D, "amk_add_layernorm_bwd: width %d not supported (multiple of 4, <= %d)", D, MAX_D);
  AMK_CHECK_ARG(a16(dy) && a16(h) && a16(dh_in) && a16(gamma) && a16(dh) && a16(dgb_part), "amk_add_layernorm_bwd: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(parts_for(M)), block(64 * WAVES);
  const size_t lds = (size_t)WAVES * D * sizeof(float);
#define CALL(NCH)                                                                                                       \
  if (dh_in) hipLaunchKernelGGL((add_layernorm_bwd_kernel<NCH, true>), grid, block, lds, st, dy, h, dh_in, gamma, mean, rstd, M, D, dh, dgb_part); \
  else hipLaunchKernelGGL((add_layernorm_bwd_kernel<NCH, false>), grid, block, lds, st, dy, h, dh_in, gamma, mean, rstd, M, D, dh, dgb_part)
  AMK_LN_DISPATCH(D, CALL);
#undef CALL
  AMK_CHECK_LAUNCH("amk_add_layernorm_bwd");
  return AMK_OK;
}

extern "C" int amk_colsum(const float* x, int64_t M, int N, float* part, void* stream) {
  AMK_CHECK_ARG(x && part, "amk_colsum: null pointer");
  AMK_CHECK_ARG(M > 0 && N > 0, "amk_colsum: non-positive size");
  AMK_CHECK_SUPPORTED(N % 4 == 0 && N <= MAX_D, "amk_colsum: width %d not supported (multiple of 4, <= %d)", N, MAX_D);
  AMK_CHECK_ARG(a16(x) && a16(part), "amk_colsum: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(parts_for(M)), block(64 * WAVES);
  const size_t lds = (size_t)WAVES * N * sizeof(float);
#define CALL(NCH) hipLaunchKernelGGL(colsum_kernel<NCH>, grid, block, lds, st, x, M, N, part)
  AMK_LN_DISPATCH(N, CALL);
#undef CALL
  AMK_CHECK_LAUNCH("amk_colsum");
  return AMK_OK;
}
