#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* o, int nbytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
  o[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, v.x);
  o[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, v.y);
  o[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, v.z);
  o[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, v.w);
}
int main() {
  float *d, *o; hipMalloc(&d, 1024 * 4); hipMalloc(&o, 1024 * 4);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 128 * 4);
  hipMemcpy(h, o, 256 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 256; i += 17) printf("%d:%g ", i, h[i]);
  printf("\n");
  return 0;
}
