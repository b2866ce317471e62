// Tile geometry and parameter blocks shared by the attention forward/backward kernels.
#pragma once
#include "amk_common.h"

namespace amk_attn {

constexpr int D = 64;        // head dim the kernels are built for
constexpr int NWAVE = 4;     // waves per workgroup (one per SIMD)
constexpr int WG = NWAVE * AMK_WAVE;
constexpr int BLK = 32 * NWAVE;  // rows (queries or keys) owned by one workgroup
constexpr int TILE = 64;     // rows of the streamed operand staged per LDS tile
// LDS row stride (floats).  68 = 64 + 4: a ds_read_b128 row read (lane -> row,
// 16-lane groups) then starts each row 4 banks after the previous one, so the
// 16 lanes of a group cover all 64 banks; ds_read_b32 column reads (lanes ->
// consecutive columns of one row) are conflict-free at any stride.
constexpr int LDS_STRIDE = 68;

// fill values in the log2 domain (scores are multiplied by log2(e) before exp2)
#define AMK_FILL_MASKED (-1.0e9f * AMK_LOG2E)

struct Strides {
  int64_t sb, st, sh;
};

struct FwdParams {
  const float *q, *k, *v;
  float *o, *stats;
  const uint8_t *key_mask, *causal_mask;
  int B, H, I, J;
  Strides qs, ks, vs, os;
  float scale;
  int nblk;  // ceil(I / BLK)
};

struct BwdParams {
  const float *q, *k, *v, *o, *stats, *d_o;
  float *dq, *dk, *dv, *delta;
  const uint8_t *key_mask, *causal_mask;
  int B, H, I, J;
  Strides qs, ks, vs, os, dos, dqs, dks, dvs;
  float scale;
  int nqblk, nkblk;  // ceil(I / BLK), ceil(J / BLK)
};

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float f4(const float4& v, int e) {
  return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
}

}  // namespace amk_attn
