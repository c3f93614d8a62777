// Micro-benchmark: issue rate of bf16 MFMA shapes on gfx950 (one wave per SIMD, operands in registers).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
  s4 a4 = {1, 2, 3, (short)threadIdx.x}, b4 = {4, 5, 6, (short)threadIdx.x};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      else acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[0], 0, 0, 0);  // ONE accumulation chain: measured 18.0 cycles per MFMA for a lone wave (2.09 PFLOP/s chip)
      // (rotating over several accumulators makes hipcc shuffle AGPRs inside this loop: read the ISA before trusting KIND 0 / 1)
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, double flop_per_mfma) {
  float *out; long long *cyc;
  const int grid = 256, iters = 20000;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 100, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n = (double)iters * 8;
  printf("%s: %.2f cycles/MFMA/wave(SIMD), wall %.2f ms, %.1f TFLOP/s chip, eff clock %.2f GHz\n", name, c / n, ms,
         n * 4 * grid * flop_per_mfma / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9);
}
int main() {
  run<0>("v_mfma_f32_16x16x32_bf16", 16384.0);
  run<1>("v_mfma_f32_16x16x16_bf16 (1k)", 8192.0);
  run<0>("v_mfma_f32_16x16x32_bf16", 16384.0);
  run<2>("v_mfma_f32_16x16x32_bf16, ONE accumulation chain", 16384.0);
  return 0;
}
