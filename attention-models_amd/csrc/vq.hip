// ViTVQGAN codebook nearest-neighbour lookup for gfx950 (exact-f32 MFMA + in-register argmin).
//
// Replaces Codebook.forward of the reference (models/vitvqgan.py:151-171), which builds the
// (N, K) distance matrix in HBM (33.5 MB per image at K = 8192) and re-reads it for the
// adds and the argmin.  Here the matrix never leaves the register file:
//
//   vq_prep_codebook : en = l2norm(E), ee[k] = sum en[k]^2                     (K rows, tiny)
//   vq_argmin        : Z-ROW ON THE LANE.  A workgroup owns 128 rows (32 per wave) and a
//        contiguous slice of the codebook.  Per 32-code tile:
//          dot^T (32 codes x 32 rows) = en_tile (A operand, LDS rows, ds_read_b128)
//                                       x zn^T  (B operand, C/2 registers, loaded once)
//          argmin_k (zz + ee[k] - 2 dot[k]) = argmax_k (dot[k] - ee[k]/2): the accumulator starts
//          at -ee/2 (read from LDS straight into the MFMA's C operand), so the score comes out of
//          the matrix pipe with no VALU work.  Every VALU instruction is paid against the f32
//          MFMA pipe (4 cycles each, measured: 1024 + 4*n cycles per tile), so the sweep only keeps
//          a running max over GROUPS of 4 consecutive codes -- nb = max(best, the group's four) as two v_max3, then
//          "nb != best" selects the group index: 16 instructions per 32-code tile (round 4; 22 with a separate group
//          maximum, 86 for a per-code argmin); strict: the first maximum wins over ascending groups; the argmax of a
//          sub-tile stands behind the first MFMAs of the next one, past the MFMA-result hazard; the two half-waves of a
//          row are merged at the end.
//   vq_finalize      : merges the per-slice (max, group) partials (lowest group wins ties), ranks
//        the 4 codes of the winning group (strict '>' ascending = torch.argmin's first-minimum
//        rule), gathers E[idx], re-normalises, forms the straight-through output and the
//        squared-error partials.
//   vq_bwd           : straight-through + commitment gradient to z through the l2-norm Jacobian;
//        codebook rows receive 2(zq - zn)/(N*C) through their own Jacobian by f32 atomics
//        (one 128-B segment per row per wave-instruction: the full-rate shape).
#include "amk_common.h"

namespace amk_vq {

constexpr int ROWS_WG = 128;   // z rows per workgroup in vq_argmin (4 waves x 32)
// codes staged per LDS tile: 4 MFMA tiles of 32 up to C = 64, fewer for wide codes (tile <= 36 KB)
template <int C> constexpr int codes_lds() { return C <= 64 ? 128 : (C == 128 ? 64 : 32); }
constexpr int FIN_ROWS = 16;   // z rows per workgroup in vq_finalize
constexpr float EPS = 1e-12f;  // F.normalize default eps

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float f4(const float4& v, int e) {
  return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void vq_prep_codebook_kernel(const float* __restrict__ E, int K, int Kpad,
                                                               float* __restrict__ en, float* __restrict__ ee) {
  // C/4 lanes per code row, one float4 each.  Rows K .. Kpad-1 (K rounded up to whole 32-code tiles per
  // slice) are padding: en = 0, ee = +inf, so their score <zn, en> - ee/2 = -inf is never the maximum.
  constexpr int LPR = C / 4;
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const int c = (threadIdx.x % LPR) * 4;
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < K) x = ld4(E + (int64_t)row * C + c);
  float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float denom = fmaxf(sqrtf(ss), EPS);
  const float4 y = make_float4(x.x / denom, x.y / denom, x.z / denom, x.w / denom);
  float s2 = y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  if (row < K) {
    st4(en + (int64_t)row * C + c, y);
    if (c == 0) ee[row] = s2;
  } else if (row < Kpad) {
    st4(en + (int64_t)row * C + c, make_float4(0.f, 0.f, 0.f, 0.f));
    if (c == 0) ee[row] = INFINITY;
  }
}

// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256, (C <= 64 ? 2 : 1)) void vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ en,
                                                          const float* __restrict__ ee, int64_t N, int K, int nsplit,
                                                          float* __restrict__ zn_out, float* __restrict__ pmin,
                                                          int32_t* __restrict__ pidx) {
  constexpr int HC = C / 2;       // k-extent owned by one half-wave
  constexpr int CODES_LDS = codes_lds<C>();
  constexpr int LS = C + 4;       // LDS row stride: conflict-free ds_read_b128 row reads
  __shared__ __attribute__((aligned(16))) float smem[CODES_LDS * LS + CODES_LDS];
  float* Es = smem;
  float* EEs = smem + CODES_LDS * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hf = lane >> 5;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  // (readfirstlane: the division runs on the vector unit; without it the loads through the slice's buffer
  // descriptor below are wrapped in waterfall loops)
  const int split = __builtin_amdgcn_readfirstlane(wg % nsplit);  // consecutive ids share the z rows, walk the slices
  const int64_t rb = __builtin_amdgcn_readfirstlane(wg / nsplit);
  const int64_t row = rb * ROWS_WG + wave * 32 + ln;
  const bool rvalid = row < N;

  // zn^T operand: lane (row, half) holds zn[row][HC*half + s]; l2norm needs the other half's sum.
  float zreg[HC];
  {
    const float* zp = z + row * C + HC * hf;
    float ss = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < HC / 4; ++s4) {
      const float4 t = rvalid ? ld4(zp + 4 * s4) : make_float4(0.f, 0.f, 0.f, 0.f);
      zreg[4 * s4 + 0] = t.x; zreg[4 * s4 + 1] = t.y; zreg[4 * s4 + 2] = t.z; zreg[4 * s4 + 3] = t.w;
      ss += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    ss += __shfl_xor(ss, 32, 64);
    const float denom = fmaxf(sqrtf(ss), EPS);
#pragma unroll
    for (int s = 0; s < HC; ++s) zreg[s] = zreg[s] / denom;
  }
  if (split == 0 && rvalid) {
    float* op = zn_out + row * C + HC * hf;
#pragma unroll
    for (int s4 = 0; s4 < HC / 4; ++s4)
      st4(op + 4 * s4, make_float4(zreg[4 * s4], zreg[4 * s4 + 1], zreg[4 * s4 + 2], zreg[4 * s4 + 3]));
  }

  const int kper = K / nsplit;          // codes in this slice (multiple of 32)
  const int kbeg = split * kper;
  const int ntile = (kper + CODES_LDS - 1) / CODES_LDS;

  // staging: a tile is CODES_LDS*C CONTIGUOUS floats of the slice = (CODES_LDS*C/4) float4 over 256
  // threads, fetched through a buffer descriptor that ends with the slice: rows past it read as zeros
  // (hardware range check), so a tile costs NF4 loads and no compares, selects or 64-bit address math
  // (every VALU instruction is paid against the f32 MFMA pipe).
  constexpr int NF4 = CODES_LDS * C / 4 / 256;
  const __amdgpu_buffer_rsrc_t en_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(en + (int64_t)kbeg * C), 0, kper * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ee_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(ee + kbeg), 0, kper * 4, 0x00020000);
  float4 est[NF4];
  float eest = 0.f;
  bool ee_ok = false;
  auto prefetch = [&](int t) {
    const int tile_off = t * CODES_LDS * C * 4;  // bytes; added to the VECTOR offset, the part the range check covers
#pragma unroll
    for (int ps = 0; ps < NF4; ++ps)
      est[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(en_rsrc, (tid + 256 * ps) * 16 + tile_off, 0, 0));
    // one ee per code (the range check returns 0 past the slice; commit() puts the padding value in)
    eest = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ee_rsrc, (tid & (CODES_LDS - 1)) * 4 + t * CODES_LDS * 4, 0, 0));
    ee_ok = t * CODES_LDS + (tid & (CODES_LDS - 1)) < kper;
  };
  auto commit = [&]() {
    // nothing that consumes the loads may be scheduled up to them (it would wait out the memory latency
    // in front of a tile's MFMAs): the loads were issued a tile ago
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ps = 0; ps < NF4; ++ps) {
      const int f = tid + 256 * ps;
      const int r = f / (C / 4), c4 = f % (C / 4);
      st4(&Es[r * LS + 4 * c4], est[ps]);
    }
    // codes past the slice: ee = +inf -> score -inf, never chosen
    if (tid < CODES_LDS) EEs[tid] = ee_ok ? -0.5f * eest : -INFINITY;
  };

  // vq.hip is compiled with -fno-honor-nans -mno-amdgpu-ieee (Makefile): fmaxf is then a bare
  // v_max_f32 / v_max3_f32 with no canonicalising v_max(x, x) in front of it
  auto vmax = [&](float a, float b) { return fmaxf(a, b); };
  float best = -INFINITY;
  int bbase = kbeg, bsub = 0;
  prefetch(0);
  for (int t = 0; t < ntile; ++t) {
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < ntile) prefetch(t + 1);
    const int c0 = kbeg + t * CODES_LDS;
    const int nsub = min(CODES_LDS, kper - t * CODES_LDS) / 32;
    // -ee/2 of this half-wave's 16 codes of sub-tile u, in accumulator order: the loads go
    // straight into the registers the MFMA chain uses as its C operand
    auto load_init = [&](int u) {
      f32x16 a;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 n = ld4(&EEs[32 * u + 8 * g + 4 * hf]);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * g + e] = f4(n, e);
      }
      return a;
    };
    struct AHead { float4 a0, a1; };  // first half of the A operand (codes) of a sub-tile
    auto load_head = [&](int u) {
      const float* er = &Es[(32 * u + ln) * LS + HC * hf];
      return AHead{ld4(er), ld4(er + 4)};
    };
    auto chain = [&](int u, f32x16 acc, const AHead& hd, int s_lo, int s_hi) {   // k-steps [4 s_lo, 4 s_hi) of sub-tile u's MFMA chain
      const float* er = &Es[(32 * u + ln) * LS + HC * hf];
#pragma unroll
      for (int s4 = s_lo; s4 < s_hi; ++s4) {
        const float4 a = s4 == 0 ? hd.a0 : (s4 == 1 ? hd.a1 : ld4(er + 4 * s4));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(f4(a, e), zreg[4 * s4 + e], acc);
      }
      return acc;
    };
    // acc[r] = dot - ee/2 for code (r&3) + 8*(r>>2) + 4*half of the sub-tile: registers 4g..4g+3
    // are 4 consecutive codes.  Running max over groups; the index is (sub-tile base, group).
    auto argmax4 = [&](int u, const f32x16& acc) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // the running maximum folded into the group's max chain: nb = max(best, group) in two v_max3, and nb != best exactly
        // when the group's maximum is strictly greater (the first maximum wins, groups ascend per lane): 4 instructions per
        // group (round 4; 5 with a separate group maximum, compare and running max)
        const float nb = vmax(vmax(vmax(vmax(acc[4 * g], acc[4 * g + 1]), acc[4 * g + 2]), acc[4 * g + 3]), best);
        bsub = nb != best ? 4 * u + g : bsub;  // an inline constant in the unrolled tile
        best = nb;
      }
    };
    auto subtile = [&](int u, f32x16 acc, const AHead& hd) { argmax4(u, chain(u, acc, hd, 0, HC / 4)); };
    const float before = best;
    if (nsub == CODES_LDS / 32) {
      // full tile, sub-tiles software-pipelined: the argmax of sub-tile u stands BEHIND the first MFMAs of sub-tile u + 1 in
      // program order, so the wave does not idle through the MFMA-result hazard (the s_nop 15 + s_nop 1 the compiler puts
      // between a chain's last MFMA and the first v_max that reads it)
      f32x16 done = chain(0, load_init(0), load_head(0), 0, HC / 4);
#pragma unroll
      for (int u = 0; u < CODES_LDS / 32; ++u) {
        f32x16 nxt = done;
        if (u + 1 < CODES_LDS / 32) {
          nxt = chain(u + 1, load_init(u + 1), load_head(u + 1), 0, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        argmax4(u, done);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < CODES_LDS / 32) nxt = chain(u + 1, nxt, load_head(u + 1), 1, HC / 4);
        done = nxt;
      }
    } else {
      for (int u = 0; u < nsub; ++u) subtile(u, load_init(u), load_head(u));
    }
    bbase = (best != before) ? c0 : bbase;  // the tile base moves once per LDS tile
  }
  // first code of the winning group: tile base + 32 * sub-tile + 8 * group + 4 * half
  int bidx = bbase + 8 * bsub + 4 * hf;
  // merge the two half-waves of a row (they own interleaved groups): lowest group on ties
  const float ob = __shfl_xor(best, 32, 64);
  const int oi = __shfl_xor(bidx, 32, 64);
  if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  if (rvalid && hf == 0) {
    pmin[row * nsplit + split] = best;
    pidx[row * nsplit + split] = bidx;
  }
}

// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(FIN_ROWS * C / 4) void vq_finalize_kernel(
    const float* __restrict__ E, const float* __restrict__ en, const float* __restrict__ ee,
    const float* __restrict__ zn, const float* __restrict__ pmin,
    const int32_t* __restrict__ pidx, int64_t N, int nsplit, int64_t* __restrict__ idx,
    float* __restrict__ out, float* __restrict__ zq_out, float* __restrict__ sqerr_partial) {
  constexpr int LPR = C / 4;
  __shared__ float red[FIN_ROWS];
  const int rl = threadIdx.x / LPR;
  const int64_t row = (int64_t)blockIdx.x * FIN_ROWS + rl;
  const int c = (threadIdx.x % LPR) * 4;
  float err = 0.f;
  if (row < N) {
    // all slices' partials first (independent loads in flight together; a load -> compare -> load loop was a
    // chain of nsplit memory latencies in a 7-us kernel), then the merge: slices hold ascending code ranges,
    // strict '>' keeps the first maximum
    constexpr int MAXS = 16;
    float pd[MAXS];
    int pi[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      const bool on = s < nsplit;
      pd[s] = on ? pmin[row * nsplit + s] : -INFINITY;
      pi[s] = on ? pidx[row * nsplit + s] : 0;
    }
    float best = pd[0];
    int bg = pi[0];
#pragma unroll
    for (int s = 1; s < MAXS; ++s)
      if (pd[s] > best) { best = pd[s]; bg = pi[s]; }
    for (int s = MAXS; s < nsplit; ++s) {  // (more than 16 slices: not chosen by vq_nsplit, kept correct)
      const float d = pmin[row * nsplit + s];
      if (d > best) { best = d; bg = pidx[row * nsplit + s]; }
    }
    // rank the 4 consecutive codes of the winning group: score = <zn, en[k]> - ee[k]/2, the lanes
    // of the row each take 4 channels; every lane uses the first lane's sums so the row agrees.
    const float4 zv = ld4(zn + row * C + c);
    int bi = bg;
    float bs = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 ev = ld4(en + (int64_t)(bg + j) * C + c);
      float sc = zv.x * ev.x + zv.y * ev.y + zv.z * ev.z + zv.w * ev.w;
#pragma unroll
      for (int o = LPR / 2; o >= 1; o >>= 1) sc += __shfl_xor(sc, o, 64);
      sc = __shfl(sc, (threadIdx.x & 63) & ~(LPR - 1), 64) - 0.5f * ee[bg + j];
      if (sc > bs) { bs = sc; bi = bg + j; }
    }
    const float4 e = ld4(E + (int64_t)bi * C + c);
    float ss = e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w;
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float denom = fmaxf(sqrtf(ss), EPS);
    const float4 q = make_float4(e.x / denom, e.y / denom, e.z / denom, e.w / denom);
    const float4 df = make_float4(q.x - zv.x, q.y - zv.y, q.z - zv.z, q.w - zv.w);
    st4(zq_out + row * C + c, q);
    st4(out + row * C + c, make_float4(zv.x + df.x, zv.y + df.y, zv.z + df.z, zv.w + df.w));
    if (c == 0) idx[row] = bi;
    err = df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
  }
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) err += __shfl_xor(err, o, 64);
  if (threadIdx.x % LPR == 0) red[rl] = err;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FIN_ROWS; ++i) s += red[i];
    sqerr_partial[blockIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void vq_bwd_kernel(const float* __restrict__ z, const float* __restrict__ E,
                                                     const float* __restrict__ zn, const float* __restrict__ zq,
                                                     const int64_t* __restrict__ idx, const float* __restrict__ g_out,
                                                     const float* __restrict__ g_loss, float beta, int64_t N, int K,
                                                     float* __restrict__ dz, float* __restrict__ dE, float* __restrict__ ge_rows) {
  // one lane per element, C lanes per row: the codebook atomics of a wave then form
  // 64/C whole 4*C-byte row segments per instruction.
  constexpr int RPB = 256 / C;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / C;
  const int c = threadIdx.x % C;
  const bool valid = row < N;
  const int64_t off = valid ? row * C + c : 0;
  const int64_t code = valid ? min(max(idx[row], (int64_t)0), (int64_t)K - 1) : 0;  // never outside the codebook
  const float zr = valid ? z[off] : 0.f;
  const float er = valid ? E[code * C + c] : 0.f;
  const float znv = valid ? zn[off] : 0.f;
  const float zqv = valid ? zq[off] : 0.f;
  const float go = valid ? g_out[off] : 0.f;
  const float coef = g_loss[0] * 2.f / (float)((double)N * C);

  const float dzn = go + coef * beta * (znv - zqv);
  const float dzq = coef * (zqv - znv);
  float nz = zr * zr, ne = er * er, pz = znv * dzn, pq = zqv * dzq;
#pragma unroll
  for (int o = C / 2; o >= 1; o >>= 1) {
    nz += __shfl_xor(nz, o, 64);
    ne += __shfl_xor(ne, o, 64);
    pz += __shfl_xor(pz, o, 64);
    pq += __shfl_xor(pq, o, 64);
  }
  nz = sqrtf(nz);
  ne = sqrtf(ne);
  // y = x / max(|x|, eps): Jacobian (I - y y^T)/|x| above the clamp, I/eps below it
  const float gz = (nz > EPS) ? (dzn - znv * pz) / nz : dzn / EPS;
  const float ge = (ne > EPS) ? (dzq - zqv * pq) / ne : dzq / EPS;
  if (valid) {
    dz[off] = gz;
    if (ge_rows) ge_rows[off] = ge;  // the caller adds the rows into the codebook gradient in a fixed order
    else atomicAdd(dE + code * C + c, ge);
  }
}

// ---------------------------------------------------------------------------------------
// Same gradient for wide codes (C = 128, 256): C/4 lanes per row, 16 bytes per lane.
template <int C>
__global__ __launch_bounds__(256) void vq_bwd_wide_kernel(const float* __restrict__ z, const float* __restrict__ E,
                                                          const float* __restrict__ zn, const float* __restrict__ zq,
                                                          const int64_t* __restrict__ idx, const float* __restrict__ g_out,
                                                          const float* __restrict__ g_loss, float beta, int64_t N, int K,
                                                          float* __restrict__ dz, float* __restrict__ dE, float* __restrict__ ge_rows) {
  constexpr int LPR = C / 4;
  const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const int c = (threadIdx.x % LPR) * 4;
  const bool valid = row < N;
  const int64_t off = valid ? row * C + c : 0;
  const int64_t code = valid ? min(max(idx[row], (int64_t)0), (int64_t)K - 1) : 0;  // never outside the codebook
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 zr = valid ? ld4(z + off) : zero, er = valid ? ld4(E + code * C + c) : zero;
  const float4 znv = valid ? ld4(zn + off) : zero, zqv = valid ? ld4(zq + off) : zero;
  const float4 go = valid ? ld4(g_out + off) : zero;
  const float coef = g_loss[0] * 2.f / (float)((double)N * C);
  float dzn[4], dzq[4];
  const float znA[4] = {znv.x, znv.y, znv.z, znv.w}, zqA[4] = {zqv.x, zqv.y, zqv.z, zqv.w};
  const float goA[4] = {go.x, go.y, go.z, go.w};
  float nz = zr.x * zr.x + zr.y * zr.y + zr.z * zr.z + zr.w * zr.w;
  float ne = er.x * er.x + er.y * er.y + er.z * er.z + er.w * er.w;
  float pz = 0.f, pq = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    dzn[e] = goA[e] + coef * beta * (znA[e] - zqA[e]);
    dzq[e] = coef * (zqA[e] - znA[e]);
    pz += znA[e] * dzn[e];
    pq += zqA[e] * dzq[e];
  }
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) {
    nz += __shfl_xor(nz, o, 64);
    ne += __shfl_xor(ne, o, 64);
    pz += __shfl_xor(pz, o, 64);
    pq += __shfl_xor(pq, o, 64);
  }
  nz = sqrtf(nz);
  ne = sqrtf(ne);
  if (valid) {
    float gz[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gz[e] = (nz > EPS) ? (dzn[e] - znA[e] * pz) / nz : dzn[e] / EPS;
      const float ge = (ne > EPS) ? (dzq[e] - zqA[e] * pq) / ne : dzq[e] / EPS;
      if (ge_rows) ge_rows[off + e] = ge;
      else atomicAdd(dE + code * C + c + e, ge);
    }
    st4(dz + off, make_float4(gz[0], gz[1], gz[2], gz[3]));
  }
}

// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void vq_gather_kernel(const int64_t* __restrict__ idx, const float* __restrict__ E,
                                                        int64_t N, int K, float* __restrict__ out, int32_t* __restrict__ bad) {
  constexpr int LPR = C / 4;
  const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const int c = (threadIdx.x % LPR) * 4;
  float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < N) {
    // indices come from the caller: never read outside the codebook; out-of-range ones are counted
    // (torch raises IndexError for them; the binding does the same from the count)
    int64_t code = idx[row];
    if (code < 0 || code >= K) {
      if (bad && c == 0) atomicAdd(bad, 1);
      code = code < 0 ? 0 : K - 1;
    }
    e = ld4(E + code * C + c);
  }
  float ss = e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w;
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float denom = fmaxf(sqrtf(ss), EPS);
  if (row < N) st4(out + row * C + c, make_float4(e.x / denom, e.y / denom, e.z / denom, e.w / denom));
}

// ---------------------------------------------------------------------------------------
template <int C>
int launch_fwd(const float* z, const float* E, int64_t N, int K, int Kpad, int nsplit, float* en, float* ee, float* pmin,
               int32_t* pidx, int64_t* idx, float* out, float* zq, float* zn, float* sqerr, hipStream_t st) {
  constexpr int LPR = C / 4;
  hipLaunchKernelGGL(vq_prep_codebook_kernel<C>, dim3((Kpad + 256 / LPR - 1) / (256 / LPR)), dim3(256), 0, st, E, K, Kpad, en, ee);
  const int64_t nrb = (N + ROWS_WG - 1) / ROWS_WG;
  hipLaunchKernelGGL(vq_argmin_kernel<C>, dim3((unsigned)(nrb * nsplit)), dim3(256), 0, st, z, en, ee, N, Kpad, nsplit, zn,
                     pmin, pidx);
  hipLaunchKernelGGL(vq_finalize_kernel<C>, dim3((unsigned)((N + FIN_ROWS - 1) / FIN_ROWS)), dim3(FIN_ROWS * LPR), 0, st,
                     E, en, ee, zn, pmin, pidx, N, nsplit, idx, out, zq, sqerr);
  return 0;
}

}  // namespace amk_vq

using namespace amk_vq;

extern "C" int64_t amk_vq_num_partials(int64_t N) { return (N + FIN_ROWS - 1) / FIN_ROWS; }

// rows of the en / ee workspaces: K rounded up to whole 32-code tiles in each of the nsplit slices
extern "C" int amk_vq_padded_codes(int K, int nsplit) {
  if (K <= 0 || nsplit <= 0) return 0;
  const int q = 32 * nsplit;
  return (K + q - 1) / q * q;
}

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int amk_vq_lookup_fwd(const float* z, const float* codebook, int64_t N, int K, int C, int nsplit,
                                 float* en_ws, float* ee_ws, float* pmin_ws, int32_t* pidx_ws,
                                 int64_t* idx, float* out, float* zq, float* zn, float* sqerr_partial,
                                 void* stream) {
  AMK_CHECK_ARG(z && codebook && en_ws && ee_ws && pmin_ws && pidx_ws && idx && out && zq && zn && sqerr_partial,
                "amk_vq_lookup_fwd: null pointer");
  AMK_CHECK_ARG(N > 0 && K > 0 && nsplit > 0, "amk_vq_lookup_fwd: non-positive size N=%lld K=%d nsplit=%d", (long long)N, K, nsplit);
  AMK_CHECK_SUPPORTED(C == 32 || C == 64 || C == 128 || C == 256, "amk_vq_lookup_fwd: codebook_dim %d not supported (32, 64, 128, 256)", C);
  const int Kpad = amk_vq_padded_codes(K, nsplit);  // en_ws / ee_ws hold Kpad rows
  AMK_CHECK_SUPPORTED((int64_t)Kpad * C * 4 < (1ll << 31), "amk_vq_lookup_fwd: codebook too large");
  AMK_CHECK_ARG(a16(z) && a16(codebook) && a16(en_ws) && a16(out) && a16(zq) && a16(zn), "amk_vq_lookup_fwd: pointers must be 16-byte aligned");
  const int64_t nwg = ((N + ROWS_WG - 1) / ROWS_WG) * nsplit;
  AMK_CHECK_SUPPORTED(nwg < (1ll << 31) && (N + FIN_ROWS - 1) / FIN_ROWS < (1ll << 31), "amk_vq_lookup_fwd: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C == 32) launch_fwd<32>(z, codebook, N, K, Kpad, nsplit, en_ws, ee_ws, pmin_ws, pidx_ws, idx, out, zq, zn, sqerr_partial, st);
  else if (C == 64) launch_fwd<64>(z, codebook, N, K, Kpad, nsplit, en_ws, ee_ws, pmin_ws, pidx_ws, idx, out, zq, zn, sqerr_partial, st);
  else if (C == 128) launch_fwd<128>(z, codebook, N, K, Kpad, nsplit, en_ws, ee_ws, pmin_ws, pidx_ws, idx, out, zq, zn, sqerr_partial, st);
  else launch_fwd<256>(z, codebook, N, K, Kpad, nsplit, en_ws, ee_ws, pmin_ws, pidx_ws, idx, out, zq, zn, sqerr_partial, st);
  AMK_CHECK_LAUNCH("amk_vq_lookup_fwd");
  return AMK_OK;
}

static int vq_bwd_impl(const char* who, const float* z, const float* codebook, const float* zn, const float* zq,
                       const int64_t* idx, const float* g_out, const float* g_loss, float beta,
                       int64_t N, int K, int C, float* dz, float* dcodebook, float* ge_rows, void* stream) {
  AMK_CHECK_ARG(z && codebook && zn && zq && idx && g_out && g_loss && dz && (dcodebook || ge_rows), "%s: null pointer", who);
  AMK_CHECK_ARG(N > 0 && K > 0, "%s: non-positive size", who);
  AMK_CHECK_SUPPORTED(C == 32 || C == 64 || C == 128 || C == 256, "%s: codebook_dim %d not supported (32, 64, 128, 256)", who, C);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!ge_rows && hipMemsetAsync(dcodebook, 0, (size_t)K * C * sizeof(float), st) != hipSuccess) {
    amk_set_error("%s: hipMemsetAsync failed", who);
    return AMK_ELAUNCH;
  }
  const int rpb = C <= 64 ? 256 / C : 256 / (C / 4);
  const int64_t nb = (N + rpb - 1) / rpb;
  AMK_CHECK_SUPPORTED(nb < (1ll << 31), "%s: grid too large", who);
  AMK_CHECK_ARG(C <= 64 || (a16(z) && a16(codebook) && a16(zn) && a16(zq) && a16(g_out) && a16(dz)), "%s: pointers must be 16-byte aligned", who);
  if (C == 32)
    hipLaunchKernelGGL(vq_bwd_kernel<32>, dim3((unsigned)nb), dim3(256), 0, st, z, codebook, zn, zq, idx, g_out, g_loss, beta, N, K, dz, dcodebook, ge_rows);
  else if (C == 64)
    hipLaunchKernelGGL(vq_bwd_kernel<64>, dim3((unsigned)nb), dim3(256), 0, st, z, codebook, zn, zq, idx, g_out, g_loss, beta, N, K, dz, dcodebook, ge_rows);
  else if (C == 128)
    hipLaunchKernelGGL(vq_bwd_wide_kernel<128>, dim3((unsigned)nb), dim3(256), 0, st, z, codebook, zn, zq, idx, g_out, g_loss, beta, N, K, dz, dcodebook, ge_rows);
  else
    hipLaunchKernelGGL(vq_bwd_wide_kernel<256>, dim3((unsigned)nb), dim3(256), 0, st, z, codebook, zn, zq, idx, g_out, g_loss, beta, N, K, dz, dcodebook, ge_rows);
  AMK_CHECK_LAUNCH(who);
  return AMK_OK;
}

extern "C" int amk_vq_lookup_bwd(const float* z, const float* codebook, const float* zn, const float* zq,
                                 const int64_t* idx, const float* g_out, const float* g_loss, float beta,
                                 int64_t N, int K, int C, float* dz, float* dcodebook, void* stream) {
  AMK_CHECK_ARG(dcodebook, "amk_vq_lookup_bwd: null pointer");
  return vq_bwd_impl("amk_vq_lookup_bwd", z, codebook, zn, zq, idx, g_out, g_loss, beta, N, K, C, dz, dcodebook, nullptr, stream);
}

extern "C" int amk_vq_lookup_bwd_rows(const float* z, const float* codebook, const float* zn, const float* zq,
                                      const int64_t* idx, const float* g_out, const float* g_loss, float beta,
                                      int64_t N, int K, int C, float* dz, float* ge_rows, void* stream) {
  AMK_CHECK_ARG(ge_rows, "amk_vq_lookup_bwd_rows: null pointer");
  return vq_bwd_impl("amk_vq_lookup_bwd_rows", z, codebook, zn, zq, idx, g_out, g_loss, beta, N, K, C, dz, nullptr, ge_rows, stream);
}

extern "C" int amk_vq_gather(const int64_t* idx, const float* codebook, int64_t N, int K, int C, float* out,
                             int32_t* bad_count, void* stream) {
  AMK_CHECK_ARG(idx && codebook && out, "amk_vq_gather: null pointer");
  AMK_CHECK_ARG(N > 0 && K > 0, "amk_vq_gather: non-positive size");
  AMK_CHECK_SUPPORTED(C == 32 || C == 64 || C == 128 || C == 256, "amk_vq_gather: codebook_dim %d not supported (32, 64, 128, 256)", C);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpb = 256 / (C / 4);
  const int64_t nb = (N + rpb - 1) / rpb;
  AMK_CHECK_SUPPORTED(nb < (1ll << 31), "amk_vq_gather: grid too large");
  if (C == 32) hipLaunchKernelGGL(vq_gather_kernel<32>, dim3((unsigned)nb), dim3(256), 0, st, idx, codebook, N, K, out, bad_count);
  else if (C == 64) hipLaunchKernelGGL(vq_gather_kernel<64>, dim3((unsigned)nb), dim3(256), 0, st, idx, codebook, N, K, out, bad_count);
  else if (C == 128) hipLaunchKernelGGL(vq_gather_kernel<128>, dim3((unsigned)nb), dim3(256), 0, st, idx, codebook, N, K, out, bad_count);
  else hipLaunchKernelGGL(vq_gather_kernel<256>, dim3((unsigned)nb), dim3(256), 0, st, idx, codebook, N, K, out, bad_count);
  AMK_CHECK_LAUNCH("amk_vq_gather");
  return AMK_OK;
}
