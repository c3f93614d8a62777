// Diagnostic: where do the workgroups of a 2-per-CU kernel land, and what does HW_REG_LDS_ALLOC read?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
  extern __shared__ char lds[];
  long long t0 = __builtin_amdgcn_s_memtime();
  unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  unsigned ldsa = __builtin_amdgcn_s_getreg((31 << 11) | 6);    // HW_REG_LDS_ALLOC
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
  lds[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hwid; out[blockIdx.x * 4 + 1] = ldsa; out[blockIdx.x * 4 + 2] = xcc;
    out[blockIdx.x * 4 + 3] = (unsigned)(t0 & 0xffffffff);
  }
}
int main() {
  const int G = 1536;
  unsigned *d; hipMalloc(&d, G * 16);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 79488);
  hipLaunchKernelGGL(k, dim3(G), dim3(256), 79488, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(G * 4); hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
  unsigned tmin = 0xffffffff; for (int i = 0; i < G; ++i) if (h[i*4+3] < tmin) tmin = h[i*4+3];
  for (int i = 0; i < G; ++i) {
    unsigned hw = h[i*4], cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    if (i < 24 || (i >= 250 && i < 270) || (i >= 508 && i < 530) || (i>=1020 && i<1030))
      printf("wg %4d xcc %u se %u sh %u cu %2u  lds_alloc %08x  t %u\n", i, h[i*4+2] & 0xf, se, sh, cu, h[i*4+1], h[i*4+3] - tmin);
  }
  return 0;
}
