#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -7.0f;     // stale marker
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 4, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x >= 32) voff = 0x80000000u;                                // out of range for half the lanes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = lds[i];
}
int main() {
  float *s, *o; hipMalloc(&s, 4096 * 4); hipMalloc(&o, 256 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i + 1;
  hipMemcpy(s, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, s, o, 4096);
  float r[256]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  printf("in-range lanes: %g %g %g %g ... lane31: %g | out-of-range lanes: %g %g %g ... %g\n", r[0], r[1], r[2], r[3], r[127], r[128], r[129], r[130], r[255]);
  return 0;
}
