// Micro-benchmark: the k-step of the 3x3 weight-gradient kernel in isolation (LDS already filled, no global traffic):
// which wave tile / waves per SIMD / LDS row order keeps the matrix pipe busy?  Cycles per k-step by s_memtime.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/wgrad_kstep.hip -o tools/ubench/wgrad_kstep
// Variants (template parameters):
//   NW      waves per workgroup (4 = one per SIMD, 8 = two per SIMD)
//   NCOW    co tiles per wave (wave tile = NCOW co tiles x 1 ci tile x 9 taps)
//   ORDER   0: lane group g reads virtual rows 4g+q, +16 (the round-1 order);  1: even rows to one half-wave, odd rows
//           to the other (conflict-free on the 2C+16 stride for consecutive rows)
//   READS   1: all fragment reads;  0: no LDS reads at all (pure MFMA ceiling)
//   PRE     1: the next k-step's dY fragments + first X tap are read during the last taps (software pipelining)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>
#include <vector>
#include <algorithm>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char *lds_cp;

constexpr int PAD_W = 12, NPIX = 100, NPAD = 144, XTAIL = 24, YROWS_ALL = 109;
constexpr int XS = 144, YS = 144, XPL = (XTAIL + NPAD) * XS, YPL = YROWS_ALL * YS;

template <int IMM>
__device__ inline s4v tr_read(uint32_t a) {
  s4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(IMM) : "memory");
  return v;
}
template <int IMM>
__device__ inline bf8 tr_frag(uint32_t lo, uint32_t hi) {
  const s4v a = tr_read<IMM>(lo), b = tr_read<IMM>(hi);
  const s8v v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf8, v);
}
template <int N, class F>
__device__ inline void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int NW, int NCOW, int ORDER, int READS, int PRE, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void kstep_kernel(float *out, int iters, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  constexpr int NTH = NW * 64;
  // fill LDS with small pseudo-random bf16 values
  for (int i = tid; i < (2 * XPL + 2 * YPL) / 4; i += NTH) {
    uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 13;
    const uint32_t a = 0x3c00u | (h & 0x1ffu) | ((h >> 9 & 1u) << 15), b = 0x3c00u | (h >> 16 & 0x1ffu) | ((h >> 10 & 1u) << 15);
    reinterpret_cast<uint32_t *>(lds)[i] = a | (b << 16);
  }
  __syncthreads();
  constexpr int WCI = 4, WCO = NW / WCI;
  const int wci = wave % WCI, wco = wave / WCI;
  const int co0w = wco * NCOW * 16;
  const uint32_t xs_a = (uint32_t)(uintptr_t)(lds_cp)lds, ys_a = xs_a + 2 * XPL;
  const int t16 = lane & 15, q4 = t16 >> 2, pq = t16 & 3;
  const uint32_t lane_col = 8 * (pq & 1) + 16 * (pq >> 1);

  f32x4 acc[9][NCOW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NCOW; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto row_addr = [&](int v, uint32_t &ya, uint32_t &xa) {
    const int i = v;  // pixel 0..95
    ya = ys_a + i * YS + lane_col + 2 * (co0w % 64);
    xa = xs_a + (XTAIL + (i / 10) * PAD_W + (i % 10)) * XS + lane_col + 32 * wci;
  };
  auto slot_rows = [&](int ks, int &v0, int &v1) {
    if (ORDER == 0) {
      v0 = 32 * ks + 4 * g + q4;
      v1 = v0 + 16;
    } else {  // half-wave (g >> 1) takes the rows of one parity; first read rows 0..15 of the k-step, second read 16..31
      const int s = 4 * (g & 1) + q4;  // 0..7
      v0 = 32 * ks + 2 * s + (g >> 1);
      v1 = v0 + 16;
    }
  };
  bf8 ah[NCOW], al[NCOW], bh[2], bl[2];
  uint32_t ya_lo, ya_hi, xa_lo, xa_hi;
  auto read_a = [&]() {
    static_for<NCOW>([&](auto JC) {
      constexpr int j = decltype(JC)::value;
      ah[j] = tr_frag<32 * j>(ya_lo, ya_hi);
      al[j] = tr_frag<YPL + 32 * j>(ya_lo, ya_hi);
    });
  };
  auto read_b = [&](auto TC, auto BC) {  // tap TC into register buffer BC
    constexpr int tap = decltype(TC)::value, buf = decltype(BC)::value;
    constexpr int XO = ((tap / 3) * PAD_W + (tap % 3)) * XS;
    bh[buf] = tr_frag<XO>(xa_lo, xa_hi);
    bl[buf] = tr_frag<XPL + XO>(xa_lo, xa_hi);
  };
  if (!READS) {
    for (int j = 0; j < NCOW; ++j)
      for (int e = 0; e < 8; ++e) { ah[j][e] = (__bf16)(0.01f * (lane + e + j)); al[j][e] = (__bf16)(0.001f * (lane + e)); }
    for (int b = 0; b < 2; ++b)
      for (int e = 0; e < 8; ++e) { bh[b][e] = (__bf16)(0.02f * (lane + e + b)); bl[b][e] = (__bf16)(0.002f * (lane + e)); }
  }
  // PAR: tap t of this k-step uses X register buffer (t + PAR) & 1 (with PRE the next k-step's tap 0 is read during
  // tap 8, so consecutive k-steps alternate the parity)
  auto kstep = [&](auto PARC, int ks, int ks_next, bool first) {
    constexpr int PAR = decltype(PARC)::value;
    using std::integral_constant;
    int v0, v1;
    if (READS && (!PRE || first)) {
      slot_rows(ks, v0, v1);
      row_addr(v0, ya_lo, xa_lo);
      row_addr(v1, ya_hi, xa_hi);
      read_a();
      read_b(integral_constant<int, 0>{}, integral_constant<int, PAR>{});
    }
    bf8 nah[NCOW], nal[NCOW];
    uint32_t nya_lo = 0, nya_hi = 0, nxa_lo = 0, nxa_hi = 0;
    static_for<9>([&](auto TC) {
      constexpr int tap = decltype(TC)::value;
      constexpr int cur = (tap + PAR) & 1, nxt = cur ^ 1;
      if (READS) {
        if constexpr (tap < 8) {
          // (lgkmcnt holds 4 bits: the next k-step's dY fragments go out in two batches, hi planes during tap 6, lo planes
          // during tap 7, each BEFORE that tap's X reads so that the counted wait covers the X fragment of this tap)
          if (PRE && tap == 6) {
            slot_rows(ks_next, v0, v1);
            row_addr(v0, nya_lo, nxa_lo);
            row_addr(v1, nya_hi, nxa_hi);
            static_for<NCOW>([&](auto JC) { nah[decltype(JC)::value] = tr_frag<32 * decltype(JC)::value>(nya_lo, nya_hi); });
          }
          if (PRE && tap == 7)
            static_for<NCOW>([&](auto JC) { nal[decltype(JC)::value] = tr_frag<YPL + 32 * decltype(JC)::value>(nya_lo, nya_hi); });
          read_b(integral_constant<int, tap + 1>{}, integral_constant<int, nxt>{});
          asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else {
          if (PRE) {  // first X tap of the next k-step, into the buffer tap 7 has released
            xa_lo = nxa_lo; xa_hi = nxa_hi; ya_lo = nya_lo; ya_hi = nya_hi;
            read_b(integral_constant<int, 0>{}, integral_constant<int, nxt>{});
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NCOW; ++j) {
        acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j], bh[cur], acc[tap][j], 0, 0, 0);
        acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bl[cur], acc[tap][j], 0, 0, 0);
        acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bh[cur], acc[tap][j], 0, 0, 0);
      }
      if (READS && PRE && tap == 8) {
#pragma unroll
        for (int j = 0; j < NCOW; ++j) { ah[j] = nah[j]; al[j] = nal[j]; }
      }
    });
  };
  long long t0 = __builtin_amdgcn_s_memtime();
  // 6 k-steps per iteration (k-step index cycles 0,1,2; with PRE consecutive k-steps alternate the buffer parity)
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, PRE ? 1 : 0>;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    kstep(I0{}, 0, 1, it == 0);
    kstep(I1{}, 1, 2, false);
    kstep(I0{}, 2, 0, false);
    kstep(I1{}, 0, 1, false);
    kstep(I0{}, 1, 2, false);
    kstep(I1{}, 2, 0, false);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NCOW; ++j) s += acc[t][j][0] + acc[t][j][1] + acc[t][j][2] + acc[t][j][3];
  out[(size_t)blockIdx.x * NTH + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NW, int NCOW, int ORDER, int READS, int PRE, int WGCU>
void run(const char *name) {
  constexpr int wg_per_cu = WGCU;
  const int grid = 256 * wg_per_cu, iters = 200;
  float *out; long long *cyc;
  hipMalloc(&out, (size_t)grid * NW * 64 * 4); hipMalloc(&cyc, grid * 8);
  const size_t lds = 2 * XPL + 2 * YPL;
  auto kern = kstep_kernel<NW, NCOW, ORDER, READS, PRE, NW / 4 * WGCU>;
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(grid);
  hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  std::sort(c.begin(), c.end());
  const double ksteps = 6.0 * iters, mf = 27.0 * NCOW;
  printf("%-44s wg/cu %d: %7.0f cycles/k-step (median), ideal %5.0f -> %4.1f %% of the MFMA rate; wall %.2f ms, %.0f TFLOP/s chip (MFMA work)\n", name, wg_per_cu,
         c[grid / 2] / ksteps, mf * 16, 100.0 * mf * 16 / (c[grid / 2] / ksteps), ms,
         ksteps * mf * NW * grid * 16384.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<4, 4, 0, 0, 0, 1>("4 waves 4co, no LDS reads");
  run<4, 4, 0, 0, 0, 2>("4 waves 4co, no LDS reads");
  run<4, 4, 0, 1, 0, 1>("4 waves 4co, round-1 order");
  run<4, 4, 0, 1, 0, 2>("4 waves 4co, round-1 order");
  run<4, 4, 1, 1, 0, 1>("4 waves 4co, parity order");
  run<4, 4, 1, 1, 0, 2>("4 waves 4co, parity order");
  run<4, 4, 1, 1, 1, 1>("4 waves 4co, parity order, prefetch");
  run<4, 4, 1, 1, 1, 2>("4 waves 4co, parity order, prefetch");
  run<8, 2, 0, 1, 0, 1>("8 waves 2co, round-1 order");
  run<8, 2, 1, 1, 0, 1>("8 waves 2co, parity order");
  run<8, 2, 1, 1, 0, 2>("8 waves 2co, parity order");
  run<8, 2, 1, 1, 1, 1>("8 waves 2co, parity order, prefetch");
  run<8, 4, 1, 1, 0, 1>("8 waves 4co (128 co), parity order");
  run<8, 4, 1, 1, 1, 1>("8 waves 4co (128 co), parity, prefetch");
  return 0;
}
