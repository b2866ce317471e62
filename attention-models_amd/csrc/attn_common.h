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
  float pinf;  // +infinity, passed at run time (see vmax)
  int nblk;  // ceil(I / BLK)
  void* x6_ws;  // split-bf16 forward: K / V planes, B*H*ceil(J/64) tiles of 48 KiB (attn_fwd_x6.hip)
  float* scores;  // kept scores (see ScoreTiles) or null
};

struct BwdParams {
  const float *q, *k, *v, *o, *stats, *d_o;
  float *dq, *dk, *dv, *delta;
  const uint8_t *key_mask, *causal_mask;
  int B, H, I, J;
  Strides qs, ks, vs, os, dos, dqs, dks, dvs;
  float scale;
  float pinf;  // +infinity, passed at run time (see vmax)
  int nqblk, nkblk;  // ceil(I / BLK), ceil(J / BLK)
  const float* scores;  // kept scores of the forward (see ScoreTiles) or null: S is recomputed
  float* dq_part;       // fused pass, reproducible dq: per-key-block partials (nkblk, B, I, H, 64), or null: atomics
};

// Kept scores: the forward can leave S^T = (q*scale*log2e) k^T (before any fill) in HBM for the backward,
// which then skips its fifth product.  Layout: 32x32 tiles of 4 KiB, [b][h][key block of 32][query
// block of 32][key][query] -- the forward (query on the lane) stores 128 contiguous bytes per
// half-wave and register, the backward (key on the lane) reads its 16 queries of one key as four
// 16-byte pieces of one 128-byte line, and a backward wave walks consecutive tiles.
// Both counts are padded to what the forward's workgroups cover (128 queries, 64-key tiles).
struct ScoreTiles {
  int nqt, nkb;
  __host__ __device__ ScoreTiles(int I, int J) : nqt(4 * ((I + 127) / 128)), nkb(2 * ((J + 63) / 64)) {}
  __host__ __device__ int64_t floats(int B, int H) const { return (int64_t)B * H * nkb * nqt * 1024; }
};

// max without the canonicalising v_max the compiler puts in front of fmaxf on MFMA outputs:
// med3(a, b, +inf) = max(a, b) for non-NaN inputs, one v_med3_f32.
// `pinf` must be a RUNTIME +infinity (a kernel argument): with a literal the compiler folds the
// med3 back into maxnum and re-inserts the canonicalisation.
__device__ __forceinline__ float vmax(float a, float b, float pinf) { return __builtin_amdgcn_fmed3f(a, b, pinf); }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Streams ROWS rows x 64 floats of one (batch, head) operand, global -> registers, 16 B per thread
// and pass (row = tid/16 + (NT/16)*pass, 16 B at column 4*(tid%16)), through a raw buffer descriptor:
// the hardware range check returns zeros for rows beyond the sequence, so a tile costs four
// address adds and no compares / selects / 64-bit multiplies (every VALU instruction is paid
// at ~4 cycles against the f32 MFMA pipe: tools/ubench_mfma_valu.hip).
template <int ROWS, int NT = 256>
struct RowStagerT {
  static constexpr int RP = NT / 16;    // rows covered by one pass of the workgroup's NT threads
  static constexpr int NP = ROWS / RP;
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[NP];
  int step;
  __device__ __forceinline__ void init(const float* base, int64_t row_stride, int nrows, int tid) {
    // bytes up to the end of the last row's 64-float slice; base / size are wave-uniform
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((int64_t)(nrows - 1) * row_stride + 64) * 4), 0x00020000);
    const int srow = tid >> 4, scol = (tid & 15) * 4;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) voff[ps] = (int)(((int64_t)(srow + RP * ps) * row_stride + scol) * 4);
    step = (int)(ROWS * row_stride * 4);
  }
  __device__ __forceinline__ void load(float4 (&dst)[NP]) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      // the builtin returns one 128-bit value: bit_cast it (assigning it to a 4 x u32 vector
      // would splat its low dword)
      dst[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[ps], 0, 0));
      voff[ps] += step;
    }
  }
  // jump so that the next load() fetches row block `blk` (blocks of ROWS rows)
  __device__ __forceinline__ void seek(int blk, int64_t row_stride, int tid) {
    const int srow = tid >> 4, scol = (tid & 15) * 4;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) voff[ps] = (int)(((int64_t)(srow + RP * ps) * row_stride + scol) * 4) + blk * step;
  }
};
typedef RowStagerT<TILE> RowStager;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float f4(const float4& v, int e) {
  return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
}

// attn_fwd_x6.hip: the forward with split-bf16 products
void launch_attn_fwd_x6(const FwdParams& p, int64_t nwg, hipStream_t st);

// attn_bwd_fused.hip: one-pass backward (dQ by atomics); false = not applicable, nothing launched.
bool launch_attn_bwd_fused(const BwdParams& p, int keys_per_wg, hipStream_t st);
int fused_keys_per_wg(int J, int keys_per_wg);

// attn_generic.hip: head dims 32 and 128 (forward + the two recompute kernels of the backward)
bool attn_gen_supported(int Dh);
void launch_attn_fwd_gen(const FwdParams& p, int Dh, int64_t nwg, hipStream_t st);
void launch_attn_bwd_gen(const BwdParams& p, int Dh, int stages, hipStream_t st);
// attn_bwd_fused_gen.hip: the one-pass backward for head dims 32 / 128 (dq by atomics); false = nothing launched
bool launch_attn_bwd_fused_gen(const BwdParams& p, int Dh, hipStream_t st);

}  // namespace amk_attn
