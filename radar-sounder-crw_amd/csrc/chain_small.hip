// Persistent LDS-resident chain for small node counts (Np = 32 or 64, i.e. N <= 64: every
// reference-default shape).  The two recurrences of the walk
//     forward :  X_{k+1} = P_k X_k                  (X = Lt with P = Gt,  X = R with P = F)
//     backward:  Y_k    += P_k^T Y_{k+1}            (Y = dLt / dR, holding the k-local terms)
// are the only sequential part of the schedule; at N = 63 one product is 0.5 MFLOP, so launching
// them one by one is pure launch latency (~7 us each, 2 x 29 per step).  Here ONE workgroup per
// (batch item, chain) walks all T-3 steps: the running matrix stays in LDS (double buffered),
// P_{k+1} is prefetched into registers while step k's MFMAs (v_mfma_f32_16x16x4_f32, exact fp32)
// run, one (LDS-only) barrier per step.  Results are written to HBM each step because the backward pass and
// the batched cycle products At_k = Lt_k^T R_k need every Lt_k / R_k.
#include "crw_common.h"

namespace crw {
namespace {

template <int NP, bool BWD, int NTH>
__global__ __launch_bounds__(NTH) void chain_small_kernel(const float *__restrict__ Gt, const float *__restrict__ F,
                                                          float *__restrict__ X0, float *__restrict__ X1, int B,
                                                          int K) {
  constexpr int LD = NP + 16;          // (LD mod 32) == 16: conflict-free ds_read_b32 fragment reads (k-major images)
  constexpr int LDA = BWD ? LD : NP + 4;  // forward keeps P row-major (m-major): rows 4 banks apart, 2-way at worst,
                                          // and no transposing 4-byte scatter (16-way bank conflicts) when staging P
  constexpr int NT = NP / 16;          // 16x16 tiles per side
  constexpr int TPW = NT * NT / (NTH / 64);  // tiles per wave: 1 (the chain is sequential and MFMA-issue bound per
                                              // wave, so Np = 64 runs 16 waves with one tile each)
  constexpr int V4 = NP * NP / 4 / NTH;      // float4 per thread per matrix
  static_assert(TPW >= 1 && V4 >= 1, "threads");
  __shared__ __attribute__((aligned(16))) float As[2][NP * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][NP * LD];

  const int b = blockIdx.x >> 1, which = blockIdx.x & 1;
  const float *P = which ? F : Gt;
  float *X = which ? X1 : X0;
  const long M = (long)NP * NP, BM = (long)B * M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto mat = [&](const float *base, int i) { return base + (long)i * BM + (long)b * M; };

  // backward only: the k-local terms Y_out that seed the accumulators, fetched one step ahead like P
  float yreg[TPW][4];
  auto load_Y = [&](int i) {
    const float *y = mat(X, i);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave * TPW + t, r0 = (tile / NT) * 16, c0 = (tile % NT) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) yreg[t][r] = y[(long)(r0 + (lane >> 4) * 4 + r) * NP + c0 + (lane & 15)];
    }
  };
  float4 preg[V4];
  auto load_P = [&](int i) {
    const float *src = mat(P, i);
#pragma unroll
    for (int v = 0; v < V4; ++v) preg[v] = *reinterpret_cast<const float4 *>(src + 4 * (tid + v * NTH));
  };
  auto store_P = [&](float *as) {  // backward: As[k][m] = P^T(m,k) = P[k][m]; forward: As[m][k] = P[m][k] -- both row copies
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int e = 4 * (tid + v * NTH), r = e / NP, c = e % NP;  // P[r][c..c+3]
      *reinterpret_cast<float4 *>(as + r * LDA + c) = preg[v];
    }
  };

  const int nsteps = K - 1;
  if (nsteps < 1) return;
  // first operands
  const int first_in = BWD ? K - 1 : 0;
  {
    const float *src = mat(X, first_in);
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int e = 4 * (tid + v * NTH), r = e / NP, c = e % NP;
      *reinterpret_cast<float4 *>(&Bs[0][r * LD + c]) = *reinterpret_cast<const float4 *>(src + e);
    }
    load_P(BWD ? K - 1 : 1);
    store_P(As[0]);
    if (BWD) load_Y(K - 2);
  }
  __syncthreads();

  int cur = 0;
  for (int st = 0; st < nsteps; ++st) {
    // forward: out index i = st+1 uses P[i];  backward: out index i = K-2-st uses P[i+1]
    const int out = BWD ? K - 2 - st : st + 1;
    const bool more = st + 1 < nsteps;
    if (more) load_P(BWD ? out : out + 1);  // P of the next step: backward P[(out-1)+1], forward P[out+1]

    f32x4 acc[TPW];
    int tr[TPW], tc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave * TPW + t;
      tr[t] = (tile / NT) * 16;
      tc[t] = (tile % NT) * 16;
      if (BWD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = yreg[t][r];
      } else {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (BWD && more) load_Y(out - 1);  // next step's seed (written by earlier kernels, never by this one)
    const float *as = As[cur], *bs = Bs[cur];
#pragma unroll 4
    for (int kk = 0; kk < NP; kk += 4) {
      const int kr = kk + (lane >> 4), c = lane & 15;
      // TPW == 4: the wave's tiles share one row block (tile / NT == wave)
      const float a0 = BWD ? as[kr * LDA + tr[0] + c] : as[(tr[0] + c) * LDA + kr];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const float a = (TPW == 4) ? a0 : (BWD ? as[kr * LDA + tr[t] + c] : as[(tr[t] + c) * LDA + kr]);
        const float bv = bs[kr * LD + tc[t] + c];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[t], 0, 0, 0);
      }
    }
    float *dst = X + (long)out * BM + (long)b * M;
    float *bn = Bs[cur ^ 1];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tr[t] + (lane >> 4) * 4 + r, col = tc[t] + (lane & 15);
        dst[(long)row * NP + col] = acc[t][r];
        bn[row * LD + col] = acc[t][r];
      }
    if (more) store_P(As[cur ^ 1]);
    // LDS-only barrier: __syncthreads() would also wait (vmcnt(0)) for this step's result stores and for the
    // prefetch of the next P, i.e. one HBM round trip per step of a 29-step sequential chain.  Nothing written to
    // global memory in this kernel is read back by another lane.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }
}

template <bool BWD>
int launch(const float *Gt, const float *F, float *X0, float *X1, int B, int K, int n, hipStream_t s) {
  if (B < 1 || K < 1 || (n != 32 && n != 64)) return CRW_EINVAL;
  if (n == 64)
    hipLaunchKernelGGL((chain_small_kernel<64, BWD, 1024>), dim3(2 * B), dim3(1024), 0, s, Gt, F, X0, X1, B, K);
  else
    hipLaunchKernelGGL((chain_small_kernel<32, BWD, 256>), dim3(2 * B), dim3(256), 0, s, Gt, F, X0, X1, B, K);
  return check_launch();
}

}  // namespace

int launch_chain_small_fwd(const float *Gt, const float *F, float *Lt, float *R, int B, int K, int n, hipStream_t s) {
  return launch<false>(Gt, F, Lt, R, B, K, n, s);
}
int launch_chain_small_bwd(const float *Gt, const float *F, float *dLt, float *dR, int B, int K, int n,
                           hipStream_t s) {
  return launch<true>(Gt, F, dLt, dR, B, K, n, s);
}

}  // namespace crw
