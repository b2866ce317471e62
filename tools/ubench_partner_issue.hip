// Micro-benchmark: what does an instruction cost a wave whose SIMD partner streams v_mfma_f32_32x32x2_f32?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_partner_issue.hip -o tools/ubench_partner_issue && ./tools/ubench_partner_issue
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run back-to-back MFMAs for the whole launch; waves 4-7
// time N instructions of one kind with s_memtime: v_add_f32, v_pk_add_f32, buffer_store_dword (coalesced 256 B),
// ds_write_b32, ds_read_b32, s_add_u32, and one MFMA of their own -- at priority 0 and 3, and with the MFMA waves idle
// as the baseline.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int N = 256;

template <int KIND, int PRIO, bool BUSY>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, float* sink, int iters) {
  __shared__ float lds[4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    if (!BUSY) return;
    f32x16 a0, a1, a2, a3;
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 1.f; a2[i] = 2.f; a3[i] = 3.f; }
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int it = 0; it < iters; ++it) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    return;
  }
  // timed waves: let the MFMA waves get going first
  __builtin_amdgcn_s_sleep(100);
  __builtin_amdgcn_s_setprio(PRIO);
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
  f32x2 pv[8];
  for (int i = 0; i < 8; ++i) pv[i] = (f32x2){(float)threadIdx.x, (float)i};
  const f32x2 one2 = {1.f, 1.f};
  f32x16 m;
  for (int i = 0; i < 16; ++i) m[i] = 0.f;
  unsigned sacc = blockIdx.x;
  float* dst = sink + (size_t)(blockIdx.x * 4 + (wave - 4)) * 64 * N + (threadIdx.x & 63);
  float* dst4 = sink + (size_t)(blockIdx.x * 4 + (wave - 4)) * 256 * N + (threadIdx.x & 63) * 4;
  const long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(1.0f));
    if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[j & 7]) : "v"(one2));
    if (KIND == 2) asm volatile("global_store_dword %0, %1, off" :: "v"(dst + 64 * j), "v"(v[j & 15]) : "memory");
    if (KIND == 3) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)((threadIdx.x & 63) * 4 + (j & 15) * 256)), "v"(v[j & 15]) : "memory");
    if (KIND == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j & 15]) : "v"((unsigned)((threadIdx.x & 63) * 4 + (j & 15) * 256)) : "memory");
    if (KIND == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
    if (KIND == 6 && (j & 15) == 0) m = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], m, 0, 0, 0);
    if (KIND == 7) { typedef float f4 __attribute__((ext_vector_type(4))); const f4 q = {v[0], v[1], v[2], v[3]};
      asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst4 + 256 * j), "v"(q) : "memory"); }
  }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t2 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = (float)sacc;
  for (int i = 0; i < 16; ++i) s += v[i] + m[i];
  for (int i = 0; i < 8; ++i) s += pv[i].x + pv[i].y;
  lds[threadIdx.x] = s;
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 4 + (wave - 4)) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + (wave - 4)) * 2 + 1] = t2 - t0; }
}

template <int KIND, int PRIO, bool BUSY>
void run(const char* name, float* out, long long* cyc, float* sink) {
  const int grid = 256;
  hipMemset(cyc, 0, grid * 8 * sizeof(long long));
  hipLaunchKernelGGL((k<KIND, PRIO, BUSY>), dim3(grid), dim3(512), 0, 0, out, cyc, sink, 4000);
  hipDeviceSynchronize();
  static long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < grid * 4; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
  const int n = (KIND == 6) ? N / 16 : N;
  printf("%-22s prio %d partner %s : %7.1f cycles per instruction to issue (%7.1f incl. drain)\n", name, PRIO, BUSY ? "MFMA" : "idle",
         a / (grid * 4) / n, b / (grid * 4) / n);
}

int main() {
  float *out, *sink; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * sizeof(long long)); hipMalloc(&sink, (size_t)256 * 4 * 64 * N * 4 * 4);
#define ALL(K, NAME) run<K, 0, false>(NAME, out, cyc, sink); run<K, 0, true>(NAME, out, cyc, sink); run<K, 3, true>(NAME, out, cyc, sink);
  ALL(0, "v_add_f32") ALL(1, "v_pk_add_f32") ALL(2, "global_store_dword") ALL(7, "global_store_dwordx4") ALL(3, "ds_write_b32") ALL(4, "ds_read_b32")
  ALL(5, "s_add_u32") ALL(6, "v_mfma_f32_32x32x2")
  return 0;
}
