// What bounds the bf16 transition-matrix chain GEMM (csrc/gemm_bf16.hip) at n = 4096?  This program times the library's OWN main loop
// -- the file is included, not copied -- in its timing-only DIAG variants (parts of the loop left out; results meaningless):
//     MFMAs alone  ->  + LDS fragment reads  ->  + barrier per k-tile  ->  + the LDS-DMA operand stream (= the whole loop of rounds 1-3;
//     the library's default at 256 x 256 staggers the requests of the two waves of a SIMD: last row)
// and the DMA stream alone / with the barrier, for every tile the kernel has (128 x 128 on 4 waves, two workgroups per CU;
// 256 x 256 on 8 waves, one per CU), plain bf16 and hi/lo pairs, in the two operand layouts the chain uses most.
// It prints MFMA flops executed per second (what bench.py's roofline_chain_n4096* lines quote) and the time per k-tile and CU.
//
//   build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I radar-sounder-crw_amd/csrc tools/ubench/gemm_ceiling.hip -o tools/ubench/gemm_ceiling
//   run  :  tools/ubench/gemm_ceiling [n=4096] [batch=4]          (on the GPU box; output committed as profiles/r04_gemm_ceiling.log)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm_bf16.hip"

namespace crw {
thread_local int g_last_hip_error = 0;
}
using namespace crw;

static const char *MODE[] = {"whole loop, all request at barrier", "MFMAs alone", "+ LDS fragment reads", "+ barrier per k-tile", "+ LDS-DMA stream (all)",
                             "LDS-DMA stream alone", "LDS-DMA stream + barrier", "whole loop (again, last)", "whole loop, staggered requests"};

template <int SPLIT, int TB, bool AKC, bool BKC, int DIAG, bool RING5 = false>
static double time_one(const GemmGroup &g, int iters) {
  for (int i = 0; i < 2; ++i) (void)launch_one<SPLIT, TB, AKC, BKC, false, DIAG, RING5>(g, nullptr);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; ++i) (void)launch_one<SPLIT, TB, AKC, BKC, false, DIAG, RING5>(g, nullptr);
  hipEventRecord(e1, nullptr);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms / iters;
}

template <int SPLIT, int TB, bool AKC, bool BKC>
static void sweep(const GemmGroup &g, const char *layout) {
  const double flops = 2.0 * g.n * (double)g.n * g.n * g.batch * (SPLIT == 3 ? 3 : 1);
  const int bk = tile_bk<SPLIT, TB>();
  const double ktiles_per_cu = (double)(g.n / TB) * (g.n / TB) * g.batch * (g.n / bk) / 256.0;  // k-tiles a CU works through
  double ms[9];
  (void)time_one<SPLIT, TB, AKC, BKC, 0>(g, 40);  // warm-up: the chip settles into the clock it holds under this kernel
  ms[0] = time_one<SPLIT, TB, AKC, BKC, 0>(g, 10);
  ms[1] = time_one<SPLIT, TB, AKC, BKC, 1>(g, 10);
  ms[2] = time_one<SPLIT, TB, AKC, BKC, 2>(g, 10);
  ms[3] = time_one<SPLIT, TB, AKC, BKC, 3>(g, 10);
  ms[4] = time_one<SPLIT, TB, AKC, BKC, 4>(g, 10);
  ms[5] = time_one<SPLIT, TB, AKC, BKC, 5>(g, 10);
  ms[6] = time_one<SPLIT, TB, AKC, BKC, 6>(g, 10);
  ms[7] = time_one<SPLIT, TB, AKC, BKC, 0>(g, 10);
  ms[8] = TB == 256 ? time_one<SPLIT, TB, AKC, BKC, 8>(g, 10) : 0.0;  // the library's default at 256 x 256 since round 4
  const double stage_kb = (SPLIT == 3 ? 4 : 2) * TB * 2.0 * bk / 1024.0;
  printf("tile %3d x %3d, BK %2d, %s, %s  (%.0f KiB of operands per k-tile, %d workgroup(s) per CU)\n", TB, TB, bk,
         SPLIT == 3 ? "hi/lo pairs (3 MFMAs per product)" : "plain bf16", layout, stage_kb, TB == 128 && SPLIT == 1 ? 2 : 1);
  for (int m = 0; m < (TB == 256 ? 9 : 8); ++m) {
    const double us_tile = ms[m] * 1e3 / ktiles_per_cu;
    if (m <= 4 || m >= 7)
      printf("    %-28s %8.1f us/launch  %7.1f TFLOP/s executed = %.3f of 2500   %6.3f us per k-tile and CU\n", MODE[m], ms[m] * 1e3,
             flops / (ms[m] * 1e-3) / 1e12, flops / (ms[m] * 1e-3) / 1e12 / 2500.0, us_tile);
    else
      printf("    %-28s %8.1f us/launch  %7.1f GB/s per CU into LDS = %5.2f TB/s chip      %6.3f us per k-tile and CU\n", MODE[m],
             ms[m] * 1e3, stage_kb * 1024.0 / (us_tile * 1e-6) / 1e9, stage_kb * 1024.0 / (us_tile * 1e-6) / 1e12 * 256, us_tile);
  }
}

// the ring of five 32-deep half-tiles (plain bf16, 256 x 256; gemm_bf16.hip mainloop_ring5): whole loop, without its DMA stream
// (both buffers of a pair hold real tiles), and its DMA stream + barrier alone
template <bool AKC, bool BKC>
static void sweep_ring5(const GemmGroup &g, const char *layout) {
  const double flops = 2.0 * g.n * (double)g.n * g.n * g.batch;
  const double ktiles_per_cu = (double)(g.n / 256) * (g.n / 256) * g.batch * (g.n / 64) / 256.0;
  (void)time_one<1, 256, AKC, BKC, 0, true>(g, 40);
  const double ms[3] = {time_one<1, 256, AKC, BKC, 0, true>(g, 10), time_one<1, 256, AKC, BKC, 3, true>(g, 10), time_one<1, 256, AKC, BKC, 6, true>(g, 10)};
  static const char *nm[3] = {"shipped kernel", "without the LDS-DMA stream", "LDS-DMA stream + barrier"};
  printf("tile 256 x 256, ring of five 32-deep half-tiles (one barrier per 64 deep), plain bf16, %s  (64 KiB of operands per 64 deep, 1 workgroup per CU)\n", layout);
  for (int m = 0; m < 3; ++m) {
    const double us_tile = ms[m] * 1e3 / ktiles_per_cu;
    if (m < 2)
      printf("    %-28s %8.1f us/launch  %7.1f TFLOP/s executed = %.3f of 2500   %6.3f us per 64 deep and CU\n", nm[m], ms[m] * 1e3,
             flops / (ms[m] * 1e-3) / 1e12, flops / (ms[m] * 1e-3) / 1e12 / 2500.0, us_tile);
    else
      printf("    %-28s %8.1f us/launch  %7.1f GB/s per CU into LDS = %5.2f TB/s chip      %6.3f us per 64 deep and CU\n", nm[m], ms[m] * 1e3,
             65536.0 / (us_tile * 1e-6) / 1e9, 65536.0 / (us_tile * 1e-6) / 1e12 * 256, us_tile);
  }
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096, batch = argc > 2 ? atoi(argv[2]) : 4;
  if (n % 256 || n < 256 || batch < 1) {
    fprintf(stderr, "n must be a multiple of 256\n");
    return 1;
  }
  const size_t elems = (size_t)n * n * batch;
  uint16_t *Ah, *Al, *Bh, *Bl;
  float *Cf;
  hipMalloc(&Ah, elems * 2); hipMalloc(&Al, elems * 2); hipMalloc(&Bh, elems * 2); hipMalloc(&Bl, elems * 2);
  hipMalloc(&Cf, elems * 4);
  {  // probability-like operands: uniform [0, 1) as bf16, lo planes ~2^-9 of them
    std::vector<uint16_t> h(elems), l(elems);
    unsigned s = 12345u;
    for (size_t i = 0; i < elems; ++i) {
      s = s * 1664525u + 1013904223u;
      const float v = (float)(s >> 8) / 16777216.0f;
      unsigned u;
      memcpy(&u, &v, 4);
      h[i] = (uint16_t)(u >> 16);
      const float lo = v * (1.0f / 512.0f);
      memcpy(&u, &lo, 4);
      l[i] = (uint16_t)(u >> 16);
    }
    hipMemcpy(Ah, h.data(), elems * 2, hipMemcpyHostToDevice); hipMemcpy(Bh, h.data(), elems * 2, hipMemcpyHostToDevice);
    hipMemcpy(Al, l.data(), elems * 2, hipMemcpyHostToDevice); hipMemcpy(Bl, l.data(), elems * 2, hipMemcpyHostToDevice);
  }
  GemmGroup g{};
  g.n = n; g.batch = batch; g.nprob = 1;
  GemmProb &p = g.p[0];
  p.A = Ah; p.Al = Al; p.B = Bh; p.Bl = Bl; p.C = Cf;
  p.sA = p.sB = p.sC = (long)n * n;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("%s, %d CUs; n = %d, batch = %d; MFMA flops EXECUTED (hi/lo pairs: 3 per product) against the 2500 TFLOP/s dense bf16 peak\n",
         prop.name, prop.multiProcessorCount, n, batch);
  // layouts: A k-contiguous & B k-contiguous (C = A B^T: ta = 0, tb = 1) ; A r-contiguous & B r-contiguous... the chain's X <- P X is
  // A k-contiguous, B r-contiguous (ta = 0, tb = 0)
  sweep<1, 256, true, true>(g, "A [m][k], B [n][k] (ds_read_b128 both)");
  sweep<1, 256, true, false>(g, "A [m][k], B [k][n] (B by ds_read_b64_tr_b16)");
  sweep_ring5<true, true>(g, "A [m][k], B [n][k]");
  sweep_ring5<true, false>(g, "A [m][k], B [k][n]");
  sweep_ring5<false, false>(g, "A [k][m], B [k][n]");
  sweep<1, 128, true, true>(g, "A [m][k], B [n][k]");
  sweep<1, 128, true, false>(g, "A [m][k], B [k][n]");
  sweep<3, 256, true, true>(g, "A [m][k], B [n][k]");
  sweep<3, 256, true, false>(g, "A [m][k], B [k][n]");
  sweep<3, 128, true, false>(g, "A [m][k], B [k][n]");
  return 0;
}
