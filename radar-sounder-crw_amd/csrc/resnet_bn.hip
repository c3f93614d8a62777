// Element-wise kernels of the Resnet encoder: train-mode BatchNorm (statistics merge, apply, backward), the 3x3/2 max-pool,
// the 1x1 stem (fc0 + bn0 + relu0, reference src/encoder.py:66-74,86-87), weight packing, fp32 <-> bf16 hi/lo planes.
// All HBM-bound, 16-byte accesses, channels-last.  Reductions are two-level with fixed summation order (double precision in the
// merges): results are bitwise reproducible.
//
// Layout conventions (see resnet_gemm.hip): raw convolution outputs Z are fp32 [Ppad][pixels][C]; activations / gradients
// that feed a matrix-core kernel are bf16 planes (hi, lo) of the same shape; rows of the patches P <= p < Ppad are ZERO in
// every plane (the GEMM epilogues then need no row predicate and the padded rows add nothing to any statistic).
// A BatchNorm's coefficients travel as coef[4][C] = (scale = gamma * invstd, shift = beta - mean * scale, mean, invstd).
#include "crw_common.h"
#include <algorithm>
#include <atomic>
#include "resnet.h"

namespace crw {
namespace {

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

struct Oct {  // eight consecutive channels
  float v[8];
};
__device__ inline Oct load8(const float *p) {
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  return Oct{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ inline void store8(float *p, const Oct &o) {
  *reinterpret_cast<float4 *>(p) = float4{o.v[0], o.v[1], o.v[2], o.v[3]};
  *reinterpret_cast<float4 *>(p + 4) = float4{o.v[4], o.v[5], o.v[6], o.v[7]};
}
__device__ inline Oct load8_planes(const uint16_t *hi, const uint16_t *lo) {
  const uint4 h = *reinterpret_cast<const uint4 *>(hi), l = *reinterpret_cast<const uint4 *>(lo);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
  Oct o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o.v[2 * i] = bf2f((uint16_t)(hw[i] & 0xffff)) + bf2f((uint16_t)(lw[i] & 0xffff));
    o.v[2 * i + 1] = bf2f((uint16_t)(hw[i] >> 16)) + bf2f((uint16_t)(lw[i] >> 16));
  }
  return o;
}
__device__ inline void store8_planes(uint16_t *hi, uint16_t *lo, const Oct &o) {
  uint32_t hw[4], lw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint16_t h0 = f2bf(o.v[2 * i]), h1 = f2bf(o.v[2 * i + 1]);
    const uint16_t l0 = f2bf(o.v[2 * i] - bf2f(h0)), l1 = f2bf(o.v[2 * i + 1] - bf2f(h1));
    hw[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    lw[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
  }
  *reinterpret_cast<uint4 *>(hi) = uint4{hw[0], hw[1], hw[2], hw[3]};
  *reinterpret_cast<uint4 *>(lo) = uint4{lw[0], lw[1], lw[2], lw[3]};
}
// sign bits of a hi plane: the activation was written as relu(.), so "> 0" is "hi != 0" (hi = 0 only for |x| < 2^-133)
__device__ inline void load_mask8(const uint16_t *hi, bool (&m)[8]) {
  const uint4 h = *reinterpret_cast<const uint4 *>(hi);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[2 * i] = (hw[i] & 0xffff) != 0;
    m[2 * i + 1] = (hw[i] >> 16) != 0;
  }
}

// ------------------------------------------------------------------------------------------------ two-level column sums
// in [R][W] fp32 -> out [ceil(R / RB)][W] double: block row r2 adds rows [r2*RB, (r2+1)*RB); block = 64 columns x 16 row lanes
// (lane l takes rows l, l+16, ...; the 16 lane sums are added in lane order): fixed order, 16 rows in flight per column
__global__ __launch_bounds__(1024) void rn_rows_reduce_kernel(const float *__restrict__ in, int R, int W, int RB,
                                                              double *__restrict__ out) {
  __shared__ double sh[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int w = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * RB, r1 = min(R, r0 + RB);
  double acc = 0.0;
  if (w < W) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 16) acc += (double)in[(long)r * W + w];
  }
  sh[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && w < W) {
    double tot = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l) tot += sh[l][cl];
    out[(long)blockIdx.y * W + w] = tot;
  }
}

// sum of the R2 rows of part2 [R2][W] at column w, by the 4 row lanes of a 64-column block (fixed order)
__device__ inline double merge_rows(const double *__restrict__ part2, int R2, int W, int w, int rl, double *sh /*[4][64]*/) {
  double acc = 0.0;
  if (w < W)
    for (int r = rl; r < R2; r += 4) acc += part2[(long)r * W + w];
  sh[rl * 64 + (threadIdx.x & 63)] = acc;
  __syncthreads();
  const int c = threadIdx.x & 63;
  const double tot = (sh[c] + sh[64 + c]) + (sh[128 + c] + sh[192 + c]);
  __syncthreads();
  return tot;
}

// BatchNorm batch statistics -> coef (scale, shift, mean, invstd) and the running statistics (nn.BatchNorm2d: momentum on
// the batch mean and the UNBIASED batch variance).  part2 [R2][C][2] = (sum, sum of squares); block = 64 channels x 4 row lanes.
__global__ __launch_bounds__(256) void rn_bn_finalize_kernel(const double *__restrict__ part2, int R2, int C, double count,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ run_mean, float *__restrict__ run_var,
                                                             float momentum, float eps, float *__restrict__ coef) {
  __shared__ double sh[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const double s1 = merge_rows(part2, R2, 2 * C, 2 * c, rl, sh);
  const double s2 = merge_rows(part2, R2, 2 * C, 2 * c + 1, rl, sh);
  if (rl != 0 || c >= C) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double scale = (double)gamma[c] * invstd;
  coef[c] = (float)scale;
  coef[C + c] = (float)((double)beta[c] - mean * scale);
  coef[2 * C + c] = (float)mean;
  coef[3 * C + c] = (float)invstd;
  if (run_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mean);
    run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unbiased);
  }
}

// ------------------------------------------------------------------------------------------------ column sums, merge in the last block
// The same two-level sum in ONE launch: grid (channel blocks, R2 row blocks); a block adds its rows of in [R][NS sums][C] (or
// [R][C][NS] when INTERLEAVED) into part2 [R2][NS][C] (doubles), takes a ticket for its channel block, and the block that draws the
// last ticket merges the R2 rows -- in row order, whichever block it is: the result does not depend on the schedule -- and runs
// the tail (BatchNorm coefficients / parameter gradients).  Saves the 6-8 us finalize launch behind every reduction of a step.
// The tickets (one counter per 64-channel block, RN_TICKET_BLOCKS of them = 128 bytes) belong to the CALLER: a zeroed corner of the
// workspace it passes, one per stream that runs these kernels at the same time; the kernel leaves them zeroed.  No library state.

struct StatsTail {  // forward: sums (s, ss) -> coef (scale, shift, mean, invstd) + running statistics (see rn_bn_finalize_kernel)
  double count;
  const float *gamma, *beta;
  float *run_mean, *run_var;
  float momentum, eps;
  float *coef;
  __device__ void operator()(int c, int C, const double *s) const {
    const double mean = s[0] / count;
    double var = s[1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double scale = (double)gamma[c] * invstd;
    coef[c] = (float)scale;
    coef[C + c] = (float)((double)beta[c] - mean * scale);
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = (float)invstd;
    if (run_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mean);
      run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unbiased);
    }
  }
};
template <int NS>
struct BwdTail {  // backward: sums (g, g xhat [, g xhat_d]) -> dbeta, dgamma [, the shortcut's] and the apply pass's constants
  double count;
  float *dgamma, *dbeta, *dgamma_d, *dbeta_d, *kc;
  __device__ void operator()(int c, int C, const double *s) const {
    dbeta[c] = (float)s[0];
    dgamma[c] = (float)s[1];
    kc[c] = (float)(s[0] / count);
    kc[C + c] = (float)(s[1] / count);
    if (NS == 3) {
      dbeta_d[c] = (float)s[0];
      dgamma_d[c] = (float)s[2];
      kc[2 * C + c] = (float)(s[2] / count);
    }
  }
};

// Hand-off between the blocks of a channel block, in the terms of the HIP memory model: every thread's partial-sum stores
// happen-before the block barrier, the barrier happens-before thread 0's ticket fetch_add, which is a RELEASE at agent scope; the
// block that reads the last ticket from it then executes an ACQUIRE fence at agent scope -- a fence after an atomic read that took
// its value from a release sequence synchronizes with every release in it -- and its barrier orders the other threads' loads behind
// the fence.  On gfx950 the release is one L2 write-back request + wait per BLOCK (the partial sums are written through by their
// sc1 stores, so it finds nothing of ours to write), the acquire one L1 / L2 invalidate in the ONE block per channel block that goes
// on (an acq_rel fetch_add would invalidate in every block and take the other kernels' L2 contents with it: mode 2, measured).
// mode (CRW_RN_TICKET, A/B only): 0 = this; 1 = "relaxed", the round-3 form -- relaxed atomics behind s_waitcnt vmcnt(0): it works on
// this chip because the sc1 stores are acknowledged by the memory side before the ticket is drawn and the sc1 loads bypass the
// non-coherent caches, but it is outside the language's model; 2 = "acqrel" on the fetch_add.
template <int NS, bool INTERLEAVED, class Tail>
__global__ __launch_bounds__(1024) void rn_sums_tail_kernel(const float *__restrict__ in, int R, int C, int RB, double *part2,
                                                            unsigned *tickets, int mode, Tail tail) {
  __shared__ double sh[NS][16][64];
  __shared__ unsigned ticket;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, W = NS * C;
  const int r0 = blockIdx.y * RB, r1 = min(R, r0 + RB);
  double acc[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) acc[n] = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 16)
#pragma unroll
      for (int n = 0; n < NS; ++n) acc[n] += (double)in[(long)r * W + (INTERLEAVED ? NS * c + n : n * C + c)];
  }
#pragma unroll
  for (int n = 0; n < NS; ++n) sh[n][rl][cl] = acc[n];
  __syncthreads();
  if (rl < NS && c < C) {  // wave n adds the 16 row lanes of sum n in lane order
    double tot = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l) tot += sh[rl][l][cl];
    // agent-scope (sc1) store: written through to where every XCD reads it.  (A device-scope __threadfence() here writes the
    // XCD's L2 back once per wave: measured +0.8 ms per step.)
    __hip_atomic_store(&part2[((long)blockIdx.y * NS + rl) * C + c], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (mode == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have been acknowledged ...
  __syncthreads();                                                  // ... for every wave of the block, before its ticket is drawn
  if (threadIdx.x == 0) {
    unsigned t;
    if (mode == 1) t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (mode == 2) t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    else {
      t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gridDim.y - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ticket = t;
  }
  __syncthreads();
  if (ticket != gridDim.y - 1) return;
  const int R2 = gridDim.y;
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    double a = 0.0;
    if (c < C)
      for (int r = rl; r < R2; r += 16) a += __hip_atomic_load(&part2[((long)r * NS + n) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[n][rl][cl] = a;
  }
  __syncthreads();
  if (rl == 0 && c < C) {
    double tot[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      double t = 0.0;
#pragma unroll
      for (int l = 0; l < 16; ++l) t += sh[n][l][cl];
      tot[n] = t;
    }
    tail(c, C, tot);
  }
  if (threadIdx.x == 0) __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zeroed again for the next launch
}

// ------------------------------------------------------------------------------------------------ BatchNorm apply
// y = relu?( z * scale + shift  [+ zd * scale_d + shift_d]  [+ residual planes] ) -> planes; rows of p >= P are written as 0
__global__ __launch_bounds__(256) void rn_bn_apply_kernel(const float *__restrict__ Z, const float *__restrict__ coef,
                                                          const float *__restrict__ Zd, const float *__restrict__ coef_d,
                                                          const uint16_t *__restrict__ res_hi, const uint16_t *__restrict__ res_lo,
                                                          long real_oct, long total_oct, int C, int relu, uint16_t *__restrict__ y_hi,
                                                          uint16_t *__restrict__ y_lo) {
  const int c8 = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (long)gridDim.x * 256) {
    Oct y;
    if (i < real_oct) {
      const int c = (int)(i % c8) * 8;
      const Oct z = load8(Z + i * 8), s = load8(coef + c), t = load8(coef + C + c);
#pragma unroll
      for (int k = 0; k < 8; ++k) y.v[k] = z.v[k] * s.v[k] + t.v[k];
      if (Zd) {
        const Oct zd = load8(Zd + i * 8), sd = load8(coef_d + c), td = load8(coef_d + C + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) y.v[k] += zd.v[k] * sd.v[k] + td.v[k];
      }
      if (res_hi) {
        const Oct r = load8_planes(res_hi + i * 8, res_lo + i * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) y.v[k] += r.v[k];
      }
      if (relu)
#pragma unroll
        for (int k = 0; k < 8; ++k) y.v[k] = fmaxf(y.v[k], 0.f);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) y.v[k] = 0.f;
    }
    store8_planes(y_hi + i * 8, y_lo + i * 8, y);
  }
}

// relu(bn(z)) followed by the 3x3 / stride 2 / padding 1 max-pool (src/encoder.py:190-193,257-260): Z [Ppad][H*W][C] ->
// planes [Ppad][Ho*Wo][C]; amax (may be null) [Ppad][Ho*Wo][C] bytes: which window position ky * 3 + kx holds the maximum --
// the FIRST one in row-major order, ATen's max_pool2d rule -- so that the backward pass routes gradients without re-deriving it
__global__ __launch_bounds__(256) void rn_bn_pool_kernel(const float *__restrict__ Z, const float *__restrict__ coef, int P, long total_oct,
                                                         int H, int W, int Ho, int Wo, int C, uint16_t *__restrict__ y_hi,
                                                         uint16_t *__restrict__ y_lo, uint8_t *__restrict__ amax) {
  const int c8 = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8;
    const long op = i / c8;
    const int o = (int)(op % (Ho * Wo));
    const long p = op / (Ho * Wo);
    Oct y;
    int am[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      y.v[k] = -1.f;  // relu output is >= 0 and every window holds at least one pixel
      am[k] = 0;
    }
    if (p < P) {
      const Oct s = load8(coef + c), t = load8(coef + C + c);
      const int oy = o / Wo, ox = o % Wo;
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = 2 * ox + kx - 1;
          if (ix < 0 || ix >= W) continue;
          const Oct z = load8(Z + ((p * H + iy) * W + ix) * C + c);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float v = fmaxf(z.v[k] * s.v[k] + t.v[k], 0.f);
            if (v > y.v[k]) {
              y.v[k] = v;
              am[k] = ky * 3 + kx;
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) y.v[k] = 0.f;
    }
    store8_planes(y_hi + i * 8, y_lo + i * 8, y);
    if (amax) {
      const uint32_t lo = (uint32_t)am[0] | ((uint32_t)am[1] << 8) | ((uint32_t)am[2] << 16) | ((uint32_t)am[3] << 24);
      const uint32_t hi = (uint32_t)am[4] | ((uint32_t)am[5] << 8) | ((uint32_t)am[6] << 16) | ((uint32_t)am[7] << 24);
      *reinterpret_cast<uint2 *>(amax + i * 8) = uint2{lo, hi};
    }
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm backward
// g = (g1 [+ g2]) where the block's output activation is positive (mask plane); per channel: sum g, sum g * xhat, and for the
// shortcut's BatchNorm (Zd) sum g * xhat_d.  Block = `rows_per_block` rows x all channels; block partials [nblk][NS][C].
template <int NS>
__global__ __launch_bounds__(256) void rn_bn_bwd_reduce_kernel(const float *__restrict__ g1, const float *__restrict__ g2,
                                                               const uint16_t *__restrict__ mask_hi, const float *__restrict__ Z,
                                                               const float *__restrict__ coef, const float *__restrict__ Zd,
                                                               const float *__restrict__ coef_d, long rows, int rows_per_block, int C,
                                                               float *__restrict__ part) {
  extern __shared__ float red[];  // [lanes][NS][C]
  const int c8 = C >> 3, lanes = 256 / c8;
  const int co = threadIdx.x % c8, rl = threadIdx.x / c8, c = co * 8;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[NS][8];
#pragma unroll
  for (int n = 0; n < NS; ++n)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[n][k] = 0.f;
  const Oct mean = load8(coef + 2 * C + c), istd = load8(coef + 3 * C + c);
  Oct mean_d = mean, istd_d = istd;
  if (NS == 3) {
    mean_d = load8(coef_d + 2 * C + c);
    istd_d = load8(coef_d + 3 * C + c);
  }
  if (rl < lanes)
    for (long r = r0 + rl; r < r1; r += lanes) {
      const long e = r * C + c;
      Oct g = load8(g1 + e);
      if (g2) {
        const Oct h = load8(g2 + e);
#pragma unroll
        for (int k = 0; k < 8; ++k) g.v[k] += h.v[k];
      }
      bool m[8];
      load_mask8(mask_hi + e, m);
      const Oct z = load8(Z + e);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float gv = m[k] ? g.v[k] : 0.f;
        acc[0][k] += gv;
        acc[1][k] += gv * ((z.v[k] - mean.v[k]) * istd.v[k]);
      }
      if (NS == 3) {
        const Oct zd = load8(Zd + e);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[2][k] += (m[k] ? g.v[k] : 0.f) * ((zd.v[k] - mean_d.v[k]) * istd_d.v[k]);
      }
    }
  if (rl < lanes)
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[(rl * NS + n) * C + c + k] = acc[n][k];
  __syncthreads();
  for (int j = threadIdx.x; j < NS * C; j += 256) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * NS * C + j];
    part[(long)blockIdx.x * NS * C + j] = s;
  }
}

// part2 [R2][NS][C] -> bwd[NS+?]: dgamma, dbeta (and the shortcut BN's) + the per-channel constants of the apply pass:
//   k[0][c] = sum g / n, k[1][c] = sum g xhat / n, k[2][c] = sum g xhat_d / n
template <int NS>
__global__ __launch_bounds__(256) void rn_bn_bwd_finalize_kernel(const double *__restrict__ part2, int R2, int C, double count,
                                                                 float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                 float *__restrict__ dgamma_d, float *__restrict__ dbeta_d,
                                                                 float *__restrict__ kc) {
  __shared__ double sh[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  double s[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) s[n] = merge_rows(part2, R2, NS * C, n * C + c, rl, sh);
  if (rl != 0 || c >= C) return;
  dbeta[c] = (float)s[0];
  dgamma[c] = (float)s[1];
  kc[c] = (float)(s[0] / count);
  kc[C + c] = (float)(s[1] / count);
  if (NS == 3) {
    dbeta_d[c] = (float)s[0];
    dgamma_d[c] = (float)s[2];
    kc[2 * C + c] = (float)(s[2] / count);
  }
}

// dZ = scale * (g - k0 - xhat * k1) -> planes (and the shortcut's dZd with its own scale / xhat_d / k2); g_out (optional):
// the masked gradient itself in fp32 (the identity shortcut of layer1 carries it to the block input).  Rows p >= P -> 0.
__global__ __launch_bounds__(256) void rn_bn_bwd_apply_kernel(const float *__restrict__ g1, const float *__restrict__ g2,
                                                              const uint16_t *__restrict__ mask_hi, const float *__restrict__ Z,
                                                              const float *__restrict__ coef, const float *__restrict__ Zd,
                                                              const float *__restrict__ coef_d, const float *__restrict__ kc,
                                                              long real_oct, long total_oct, int C, uint16_t *__restrict__ dz_hi,
                                                              uint16_t *__restrict__ dz_lo, uint16_t *__restrict__ dzd_hi,
                                                              uint16_t *__restrict__ dzd_lo, float *__restrict__ g_out) {
  const int c8 = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (long)gridDim.x * 256) {
    Oct dz, dzd, gm;
#pragma unroll
    for (int k = 0; k < 8; ++k) dz.v[k] = dzd.v[k] = gm.v[k] = 0.f;
    if (i < real_oct) {
      const int c = (int)(i % c8) * 8;
      const long e = i * 8;
      Oct g = load8(g1 + e);
      if (g2) {
        const Oct h = load8(g2 + e);
#pragma unroll
        for (int k = 0; k < 8; ++k) g.v[k] += h.v[k];
      }
      bool m[8];
      load_mask8(mask_hi + e, m);
      const Oct z = load8(Z + e), sc = load8(coef + c), mean = load8(coef + 2 * C + c), istd = load8(coef + 3 * C + c);
      const Oct k0 = load8(kc + c), k1 = load8(kc + C + c);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        gm.v[k] = m[k] ? g.v[k] : 0.f;
        dz.v[k] = sc.v[k] * (gm.v[k] - k0.v[k] - (z.v[k] - mean.v[k]) * istd.v[k] * k1.v[k]);
      }
      if (Zd) {
        const Oct zd = load8(Zd + e), sd = load8(coef_d + c), md = load8(coef_d + 2 * C + c), id = load8(coef_d + 3 * C + c);
        const Oct k2 = load8(kc + 2 * C + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) dzd.v[k] = sd.v[k] * (gm.v[k] - k0.v[k] - (zd.v[k] - md.v[k]) * id.v[k] * k2.v[k]);
      }
    }
    store8_planes(dz_hi + i * 8, dz_lo + i * 8, dz);
    if (dzd_hi) store8_planes(dzd_hi + i * 8, dzd_lo + i * 8, dzd);
    if (g_out) store8(g_out + i * 8, gm);
  }
}

// ------------------------------------------------------------------------------------------------ max-pool + bn1 backward
// Gradient of pooled = maxpool3x3/2(relu(bn(z))) with respect to bn(z) at pixel (iy, ix) of patch p, channels c..c+7: the sum
// over the (at most four) windows that contain the pixel of their gradient where the recorded arg-max is this pixel, gated by relu.
__device__ inline Oct pool_grad8(const float *__restrict__ d1, const float *__restrict__ d2, const uint8_t *__restrict__ amax, long p, int iy,
                                 int ix, int c, int C, int Ho, int Wo, const Oct &z, const Oct &scale, const Oct &shift) {
  Oct g;
#pragma unroll
  for (int k = 0; k < 8; ++k) g.v[k] = 0.f;
  const int oy0 = iy < 1 ? 0 : iy / 2, oy1 = min(Ho - 1, (iy + 1) / 2);  // windows with 2 oy - 1 <= iy <= 2 oy + 1: one or two rows
  const int ox0 = ix < 1 ? 0 : ix / 2, ox1 = min(Wo - 1, (ix + 1) / 2);
  // the (at most) 2 x 2 candidate windows: all loads issued up front with clamped indices, validity applied afterwards (loads
  // under data-dependent loop bounds are waited for one by one)
  uint2 am[4];
  Oct d[4];
  bool ok[4];
  int code[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int oy = oy0 + (w >> 1), ox = ox0 + (w & 1);
    ok[w] = oy <= oy1 && ox <= ox1;
    const int cy = min(oy, oy1), cx = min(ox, ox1);
    const long e = ((p * Ho + cy) * Wo + cx) * C + c;
    code[w] = (iy - 2 * cy + 1) * 3 + (ix - 2 * cx + 1);
    am[w] = *reinterpret_cast<const uint2 *>(amax + e);
    d[w] = load8(d1 + e);
    if (d2) {
      const Oct h = load8(d2 + e);
#pragma unroll
      for (int k = 0; k < 8; ++k) d[w].v[k] += h.v[k];
    }
  }
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int a = (int)(((k < 4 ? am[w].x : am[w].y) >> (8 * (k & 3))) & 0xff);
      g.v[k] += (ok[w] && a == code[w]) ? d[w].v[k] : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) g.v[k] = (z.v[k] * scale.v[k] + shift.v[k]) > 0.f ? g.v[k] : 0.f;
  return g;
}

// per channel: sum g, sum g * xhat over (patches, pixels); same block shape and partial layout as rn_bn_bwd_reduce_kernel<2>
__global__ __launch_bounds__(256) void rn_pool_bwd_reduce_kernel(const float *__restrict__ d1, const float *__restrict__ d2,
                                                                 const uint8_t *__restrict__ amax, const float *__restrict__ Z,
                                                                 const float *__restrict__ coef, long rows, int rows_per_block, int H, int W,
                                                                 int C, float *__restrict__ part) {
  extern __shared__ float red[];  // [lanes][2][C]
  const int c8 = C >> 3, lanes = 256 / c8;
  const int co = threadIdx.x % c8, rl = threadIdx.x / c8, c = co * 8;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[2][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[0][k] = acc[1][k] = 0.f;
  const Oct scale = load8(coef + c), shift = load8(coef + C + c), mean = load8(coef + 2 * C + c), istd = load8(coef + 3 * C + c);
  for (long r = r0 + rl; r < r1; r += lanes) {
    const long p = r / (H * W);
    const int i = (int)(r % (H * W));
    const Oct z = load8(Z + r * C + c);
    const Oct g = pool_grad8(d1, d2, amax, p, i / W, i % W, c, C, Ho, Wo, z, scale, shift);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[0][k] += g.v[k];
      acc[1][k] += g.v[k] * ((z.v[k] - mean.v[k]) * istd.v[k]);
    }
  }
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(rl * 2 + n) * C + c + k] = acc[n][k];
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * C; j += 256) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * 2 * C + j];
    part[(long)blockIdx.x * 2 * C + j] = s;
  }
}

__global__ __launch_bounds__(256) void rn_pool_bwd_apply_kernel(const float *__restrict__ d1, const float *__restrict__ d2,
                                                                const uint8_t *__restrict__ amax, const float *__restrict__ Z,
                                                                const float *__restrict__ coef, const float *__restrict__ kc, long real_oct,
                                                                long total_oct, int H, int W, int C, uint16_t *__restrict__ dz_hi,
                                                                uint16_t *__restrict__ dz_lo) {
  const int c8 = C >> 3;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (long)gridDim.x * 256) {
    Oct dz;
#pragma unroll
    for (int k = 0; k < 8; ++k) dz.v[k] = 0.f;
    if (i < real_oct) {
      const int c = (int)(i % c8) * 8;
      const long r = i / c8;
      const long p = r / (H * W);
      const int px = (int)(r % (H * W));
      const Oct z = load8(Z + i * 8), scale = load8(coef + c), shift = load8(coef + C + c), mean = load8(coef + 2 * C + c),
                istd = load8(coef + 3 * C + c), k0 = load8(kc + c), k1 = load8(kc + C + c);
      const Oct g = pool_grad8(d1, d2, amax, p, px / W, px % W, c, C, Ho, Wo, z, scale, shift);
#pragma unroll
      for (int k = 0; k < 8; ++k) dz.v[k] = scale.v[k] * (g.v[k] - k0.v[k] - (z.v[k] - mean.v[k]) * istd.v[k] * k1.v[k]);
    }
    store8_planes(dz_hi + i * 8, dz_lo + i * 8, dz);
  }
}

// ------------------------------------------------------------------------------------------------ stem: fc0 (1x1, padding 1) + bn0 + relu0
// moments of the input patches: per block (sum x_i, sum x_i x_j) -> part [nblk][8] (cin <= 2: s0, s1, s00, s01, s11)
__global__ __launch_bounds__(256) void rn_stem_moments_kernel(const float *__restrict__ x, int P, int cin, int hw, float *__restrict__ part) {
  __shared__ float red[4][8];
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const long total = (long)P * hw;  // pixels
  const long per = (total + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * per, i1 = min(total, i0 + per);
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const long p = i / hw, q = i % hw;
    const float a = x[(p * cin) * hw + q];
    s[0] += a;
    s[2] += a * a;
    if (cin == 2) {
      const float b = x[(p * cin + 1) * hw + q];
      s[1] += b;
      s[3] += a * b;
      s[4] += b * b;
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) s[k] = wave_sum(s[k]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 8)
    part[(long)blockIdx.x * 8 + threadIdx.x] =
        threadIdx.x < 5 ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.f;
}

// One thread: batch statistics of y_c = sum_i w[c][i] x~_i + b[c] over the zero-padded (h+2) x (w+2) map, from the input moments.
// stem [32] (saved for the backward): [0..2] scale_c, [3..5] shift_c, [6..8] mean_c, [9..11] invstd_c, [12..17] a[c][i] = scale_c * w[c][i],
// [18..20] d_c = scale_c * b_c + shift_c, [21..25] the moments (s0, s1, s00, s01, s11), [26] n
__global__ void rn_stem_finalize_kernel(const double *__restrict__ part2, int R2, int cin, double n, const float *__restrict__ w0,
                                        const float *__restrict__ b0, const float *__restrict__ gamma, const float *__restrict__ beta,
                                        float *__restrict__ run_mean, float *__restrict__ run_var, float momentum, float eps,
                                        float *__restrict__ stem) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double m[5] = {0, 0, 0, 0, 0};
  for (int r = 0; r < R2; ++r)
    for (int k = 0; k < 5; ++k) m[k] += part2[(long)r * 8 + k];
  const double mu[2] = {m[0] / n, m[1] / n};
  const double cov[2][2] = {{m[2] / n - mu[0] * mu[0], m[3] / n - mu[0] * mu[1]}, {m[3] / n - mu[0] * mu[1], m[4] / n - mu[1] * mu[1]}};
  for (int c = 0; c < 3; ++c) {
    double mean = b0[c], var = 0.0;
    for (int i = 0; i < cin; ++i) {
      mean += (double)w0[c * cin + i] * mu[i];
      for (int j = 0; j < cin; ++j) var += (double)w0[c * cin + i] * (double)w0[c * cin + j] * cov[i][j];
    }
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps), scale = (double)gamma[c] * invstd, shift = (double)beta[c] - mean * scale;
    stem[c] = (float)scale;
    stem[3 + c] = (float)shift;
    stem[6 + c] = (float)mean;
    stem[9 + c] = (float)invstd;
    for (int i = 0; i < 2; ++i) stem[12 + 2 * c + i] = i < cin ? (float)(scale * (double)w0[c * cin + i]) : 0.f;
    stem[18 + c] = (float)(scale * (double)b0[c] + shift);
    if (run_mean) {
      const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
      run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mean);
      run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unbiased);
    }
  }
  for (int k = 0; k < 5; ++k) stem[21 + k] = (float)m[k];
  stem[26] = (float)n;
}

// x [P][cin][h][w] -> zero-padded 4-channel map planes [Ppad][Hm][Wm][4]: relu(bn0(fc0(x))) on the (h+2) x (w+2) map placed at
// offset 3 (the padding of the 7x7 convolution), channel 3 = 0.  Thread = map pixel.
__global__ __launch_bounds__(256) void rn_stem_apply_kernel(const float *__restrict__ x, const float *__restrict__ stem, int P, long total_px,
                                                            int cin, int h, int w, int Hm, int Wm, uint16_t *__restrict__ m_hi,
                                                            uint16_t *__restrict__ m_lo) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_px; i += (long)gridDim.x * 256) {
    const int mx = (int)(i % Wm), my = (int)((i / Wm) % Hm);
    const long p = i / ((long)Wm * Hm);
    float v[3] = {0.f, 0.f, 0.f};
    const int y0 = my - 3, x0 = mx - 3;  // position on the (h+2) x (w+2) map of fc0
    if (p < P && y0 >= 0 && y0 < h + 2 && x0 >= 0 && x0 < w + 2) {
      float xi[2] = {0.f, 0.f};
      if (y0 >= 1 && y0 <= h && x0 >= 1 && x0 <= w)
        for (int k = 0; k < cin; ++k) xi[k] = x[((p * cin + k) * h + (y0 - 1)) * w + (x0 - 1)];
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = fmaxf(stem[12 + 2 * c] * xi[0] + stem[13 + 2 * c] * xi[1] + stem[18 + c], 0.f);
    }
    uint16_t hh[4], ll[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      hh[c] = f2bf(v[c]);
      ll[c] = f2bf(v[c] - bf2f(hh[c]));
    }
    hh[3] = ll[3] = 0;
    *reinterpret_cast<uint2 *>(m_hi + i * 4) = uint2{(uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16)};
    *reinterpret_cast<uint2 *>(m_lo + i * 4) = uint2{(uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16)};
  }
}

// Backward of the stem from dX0 [Ppad][H0][ldx] (columns ix * 3 + c: gradient of the conv1 input on the (h+2) x (w+2) map):
// per block and channel c: S1 = sum g, S2 = sum g xhat_c, S3[i] = sum g x~_i with g = dX0 where relu0 passed.  part [nblk][16]
// (c * 4 + {S1, S2, S3_0, S3_1}).
__global__ __launch_bounds__(256) void rn_stem_bwd_reduce_kernel(const float *__restrict__ dX0, const float *__restrict__ x,
                                                                 const float *__restrict__ stem, const float *__restrict__ w0,
                                                                 const float *__restrict__ b0, int P, int cin, int h, int w, int ldx,
                                                                 float *__restrict__ part) {
  __shared__ float red[4][12];
  const int H0 = h + 2, W0 = w + 2;
  const long total = (long)P * H0 * W0;
  const long per = (total + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * per, i1 = min(total, i0 + per);
  float s[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) s[k] = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const int x0 = (int)(i % W0), y0 = (int)((i / W0) % H0);
    const long p = i / ((long)W0 * H0);
    float xi[2] = {0.f, 0.f};
    if (y0 >= 1 && y0 <= h && x0 >= 1 && x0 <= w)
      for (int k = 0; k < cin; ++k) xi[k] = x[((p * cin + k) * h + (y0 - 1)) * w + (x0 - 1)];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float y = stem[12 + 2 * c] * xi[0] + stem[13 + 2 * c] * xi[1] + stem[18 + c];
      const float g = y > 0.f ? dX0[(p * H0 + y0) * ldx + x0 * 3 + c] : 0.f;
      float pre = b0[c];
      for (int k = 0; k < cin; ++k) pre += w0[c * cin + k] * xi[k];
      const float xh = (pre - stem[6 + c]) * stem[9 + c];
      s[c * 4] += g;
      s[c * 4 + 1] += g * xh;
      s[c * 4 + 2] += g * xi[0];
      s[c * 4 + 3] += g * xi[1];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) s[k] = wave_sum(s[k]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 12; ++k) red[threadIdx.x >> 6][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 16)
    part[(long)blockIdx.x * 16 + threadIdx.x] =
        threadIdx.x < 12 ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.f;
}

// dgamma0 = S2, dbeta0 = S1; with dZ0 = scale (g - S1/n - xhat S2/n):  dw0[c][i] = sum dZ0 x~_i, db0[c] = sum dZ0 (= 0: sum xhat = 0)
__global__ void rn_stem_bwd_finalize_kernel(const double *__restrict__ part2, int R2, int cin, const float *__restrict__ stem,
                                            const float *__restrict__ w0, const float *__restrict__ b0, float *__restrict__ dw0,
                                            float *__restrict__ db0, float *__restrict__ dgamma, float *__restrict__ dbeta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double S[12];
  for (int k = 0; k < 12; ++k) S[k] = 0.0;
  for (int r = 0; r < R2; ++r)
    for (int k = 0; k < 12; ++k) S[k] += part2[(long)r * 16 + k];
  const double n = stem[26];
  const double sx[2] = {stem[21], stem[22]};
  const double sxx[2][2] = {{stem[23], stem[24]}, {stem[24], stem[25]}};
  for (int c = 0; c < 3; ++c) {
    const double S1 = S[c * 4], S2 = S[c * 4 + 1];
    const double scale = stem[c], mean = stem[6 + c], invstd = stem[9 + c];
    dgamma[c] = (float)S2;
    dbeta[c] = (float)S1;
    double sum_xh = 0.0;  // sum over the map of xhat_c (0 up to rounding)
    for (int i = 0; i < cin; ++i) {
      // sum xhat_c x~_i = invstd * (sum_j w[c][j] sum x~_j x~_i + (b_c - mean_c) sum x~_i)
      double sxh = ((double)b0[c] - mean) * sx[i];
      for (int j = 0; j < cin; ++j) sxh += (double)w0[c * cin + j] * sxx[j][i];
      sxh *= invstd;
      dw0[c * cin + i] = (float)(scale * (S[c * 4 + 2 + i] - S1 / n * sx[i] - S2 / n * sxh));
      sum_xh += (double)w0[c * cin + i] * sx[i];
    }
    sum_xh = (sum_xh + ((double)b0[c] - mean) * n) * invstd;
    db0[c] = (float)(-scale * S2 / n * sum_xh);
  }
}

// ------------------------------------------------------------------------------------------------ weight packing, planes
// w [cout][cin][T] fp32 -> forward planes [cout][T][cin], backward planes [cin][T][cout] (hi, lo)
__global__ __launch_bounds__(256) void rn_pack_conv_kernel(const float *__restrict__ w, int cout, int cin, int T, uint16_t *__restrict__ fh,
                                                           uint16_t *__restrict__ fl, uint16_t *__restrict__ bh, uint16_t *__restrict__ bl) {
  const long total = (long)cout * cin * T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    // i indexes the forward plane [co][t][ci] (coalesced stores); the source read is strided
    const int ci = (int)(i % cin), t = (int)((i / cin) % T), co = (int)(i / ((long)cin * T));
    const float v = w[((long)co * cin + ci) * T + t];
    const uint16_t h = f2bf(v), l = f2bf(v - bf2f(h));
    fh[i] = h;
    fl[i] = l;
    const long j = ((long)ci * T + t) * cout + co;
    bh[j] = h;
    bl[j] = l;
  }
}

// every convolution / linear weight of the network in ONE launch (12 launches of the kernel above were 115 us of a 5.5 ms
// step).  Block = 8 output channels x one chunk of input channels of one layer, staged through LDS so that BOTH layouts leave in
// 16-byte pieces: the forward planes [co][t][ci] as 8 consecutive ci, the backward planes [ci][t][co] as the block's 8 co.  (The
// first version -- a block per output channel, one 2-byte store per element and plane -- wrote 318 MB per step for 40 MB of
// planes: every backward-plane store touched its own 32-byte sector; PMC, profiles/r03_pmc_resnet.json.)
constexpr int RN_PACK_CO = 8;
__device__ inline int rn_pack_ci_chunk(int T) { return T <= 16 ? 64 : 8; }  // 8 * chunk * T floats of LDS: at most 32 KB (T <= 64)
__global__ __launch_bounds__(256) void rn_pack_all_kernel(RnPackJobs jobs) {
  extern __shared__ float pv[];  // [8 co][chunk ci][T]
  int j = 0;
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].first_block) ++j;
  const RnPackJob &q = jobs.job[j];
  const int T = q.T, cb = rn_pack_ci_chunk(T), ncc = q.cin / cb;
  const int local = blockIdx.x - q.first_block;
  const int co0 = (local / ncc) * RN_PACK_CO, ci0 = (local % ncc) * cb;
  const int row = cb * T;  // values of one output channel in this chunk: contiguous in the source (ci-major, tap-minor)
  for (int idx = threadIdx.x; idx < RN_PACK_CO * row; idx += 256) {
    const int r = idx / row, e = idx - r * row;
    pv[idx] = q.bcast ? q.w[(long)(co0 + r) * q.cin + ci0 + e / T] * q.scale : q.w[((long)(co0 + r) * q.cin + ci0) * T + e];
  }
  __syncthreads();
  // forward planes: (co, t) rows of q.cin values; this block holds cb of them per row = cb / 8 pieces of 16 bytes
  for (int task = threadIdx.x; task < RN_PACK_CO * T * (cb / 8); task += 256) {
    const int c8 = task % (cb / 8), t = (task / (cb / 8)) % T, r = task / ((cb / 8) * T);
    uint32_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const float v0 = pv[r * row + (c8 * 8 + k) * T + t], v1 = pv[r * row + (c8 * 8 + k + 1) * T + t];
      const uint16_t h0 = f2bf(v0), h1 = f2bf(v1);
      h[k / 2] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      l[k / 2] = (uint32_t)f2bf(v0 - bf2f(h0)) | ((uint32_t)f2bf(v1 - bf2f(h1)) << 16);
    }
    const long f = ((long)(co0 + r) * T + t) * q.cin + ci0 + c8 * 8;
    *reinterpret_cast<uint4 *>(q.fh + f) = uint4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uint4 *>(q.fl + f) = uint4{l[0], l[1], l[2], l[3]};
  }
  // backward planes: (ci, t) rows of q.cout values; this block holds its 8 output channels of each = one piece of 16 bytes
  for (int task = threadIdx.x; task < row; task += 256) {  // task = ci_local * T + t, the source order
    uint32_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const float v0 = pv[k * row + task], v1 = pv[(k + 1) * row + task];
      const uint16_t h0 = f2bf(v0), h1 = f2bf(v1);
      h[k / 2] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      l[k / 2] = (uint32_t)f2bf(v0 - bf2f(h0)) | ((uint32_t)f2bf(v1 - bf2f(h1)) << 16);
    }
    const long b = ((long)ci0 * T + task) * q.cout + co0;
    *reinterpret_cast<uint4 *>(q.bh + b) = uint4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uint4 *>(q.bl + b) = uint4{l[0], l[1], l[2], l[3]};
  }
}

// stem 7x7/2 weights w1 [64][3][7][7]:
//   forward planes [64][256]: k = ky * 32 + kx * 4 + c (zero for ky = 7, kx = 7, c = 3)
//   Toeplitz planes [H0][ncols][ldt] for the backward-data GEMM (ncols = 3 * W0 rounded up to 64): row iy, column n = ix * 3 + c (zero for n >= 3 * W0),
//     k = (oy - oy0(iy)) * (W1 * 64) + ox * 64 + co holds w1[co][c][iy + 3 - 2 oy][ix + 3 - 2 ox] (zero outside the kernel)
__global__ __launch_bounds__(256) void rn_pack_stem_kernel(const float *__restrict__ w1, int H0, int W0, int H1, int W1, int ldt, int ncols,
                                                           uint16_t *__restrict__ fh, uint16_t *__restrict__ fl, uint16_t *__restrict__ th,
                                                           uint16_t *__restrict__ tl) {
  const long nf = 64 * 256, nt = (long)H0 * ncols * ldt;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nf + nt; i += (long)gridDim.x * 256) {
    float v = 0.f;
    if (i < nf) {
      const int k = (int)(i & 255), co = (int)(i >> 8);
      const int ky = k >> 5, kx = (k >> 2) & 7, c = k & 3;
      if (ky < 7 && kx < 7 && c < 3) v = w1[((co * 3 + c) * 7 + ky) * 7 + kx];
      const uint16_t h = f2bf(v);
      fh[i] = h;
      fl[i] = f2bf(v - bf2f(h));
    } else {
      const long j = i - nf;
      const int k = (int)(j % ldt), n = (int)((j / ldt) % ncols), iy = (int)(j / ((long)ldt * ncols));
      int oy0 = (iy + 3 - 7 + 2) / 2;
      if (iy + 3 - 7 + 1 <= 0) oy0 = 0;
      const int co = k & 63, ox = (k >> 6) % W1, oy = oy0 + (k >> 6) / W1;
      const int ix = n / 3, c = n % 3;
      const int ky = iy + 3 - 2 * oy, kx = ix + 3 - 2 * ox;
      if (n < 3 * W0 && oy < H1 && ky >= 0 && ky < 7 && kx >= 0 && kx < 7) v = w1[((co * 3 + c) * 7 + ky) * 7 + kx];
      const uint16_t h = f2bf(v);
      th[j] = h;
      tl[j] = f2bf(v - bf2f(h));
    }
  }
}

// dw [n] = scale * sum_t src [n][T], in tap order (the head's weight gradient behind an average pool over T pixels)
__global__ __launch_bounds__(256) void rn_tapsum_kernel(const float *__restrict__ src, long n, int T, float scale, float *__restrict__ dw) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += src[i * T + t];
    dw[i] = acc * scale;
  }
}

// eval-mode BatchNorm (nn.BatchNorm2d with training = False: normalise by the RUNNING statistics, update nothing)
__global__ __launch_bounds__(256) void rn_bn_coef_eval_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              const float *__restrict__ run_mean, const float *__restrict__ run_var, float eps,
                                                              int C, float *__restrict__ coef) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double invstd = 1.0 / sqrt((double)run_var[c] + (double)eps), scale = (double)gamma[c] * invstd;
  coef[c] = (float)scale;
  coef[C + c] = (float)((double)beta[c] - (double)run_mean[c] * scale);
  coef[2 * C + c] = run_mean[c];
  coef[3 * C + c] = (float)invstd;
}

// the stem record (rn_stem_finalize_kernel's layout) from bn0's running statistics: entries [0..20]; the moments [21..26] are zero
__global__ void rn_stem_eval_kernel(int cin, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, const float *__restrict__ run_mean, const float *__restrict__ run_var,
                                    float eps, float *__restrict__ stem) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int c = 0; c < 3; ++c) {
    const double invstd = 1.0 / sqrt((double)run_var[c] + (double)eps), scale = (double)gamma[c] * invstd;
    const double shift = (double)beta[c] - (double)run_mean[c] * scale;
    stem[c] = (float)scale;
    stem[3 + c] = (float)shift;
    stem[6 + c] = run_mean[c];
    stem[9 + c] = (float)invstd;
    for (int i = 0; i < 2; ++i) stem[12 + 2 * c + i] = i < cin ? (float)(scale * (double)w0[c * cin + i]) : 0.f;
    stem[18 + c] = (float)(scale * (double)b0[c] + shift);
  }
  for (int k = 21; k < 32; ++k) stem[k] = 0.f;
}

// fp32 [rows][C] -> planes [rows_pad][C], zero rows beyond `rows`
__global__ __launch_bounds__(256) void rn_split_kernel(const float *__restrict__ x, long real_oct, long total_oct, uint16_t *__restrict__ hi,
                                                       uint16_t *__restrict__ lo) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (long)gridDim.x * 256) {
    Oct v;
    if (i < real_oct) v = load8(x + i * 8);
    else
#pragma unroll
      for (int k = 0; k < 8; ++k) v.v[k] = 0.f;
    store8_planes(hi + i * 8, lo + i * 8, v);
  }
}

__global__ __launch_bounds__(256) void rn_colsum_finalize_kernel(const double *__restrict__ part2, int R2, int W, float *__restrict__ out) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= W) return;
  double acc = 0.0;
  for (int r = 0; r < R2; ++r) acc += part2[(long)r * W + w];
  out[w] = (float)acc;
}

inline unsigned grid_for(long n, int per_block = 256, long cap = 8192) {
  long b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

// in [R][NS][C] (or [R][C][NS]) fp32 -> sums over R per channel, handed to `tail` by the last block; part2: 64 * NS * C doubles;
// tickets: RN_TICKET_BLOCKS zeroed counters of the caller's (left zeroed)
template <int NS, bool INTERLEAVED, class Tail>
int rn_sums_tail(const float *in, int R, int C, double *part2, unsigned *tickets, const Tail &tail, hipStream_t s) {
  if ((C + 63) / 64 > RN_TICKET_BLOCKS || !tickets) return CRW_EINVAL;
  static const int mode = [] {
    const char *e = getenv("CRW_RN_TICKET");
    return !e ? 0 : (e[0] == 'r' ? 1 : (e[0] == 'a' ? 2 : 0));  // "relaxed" | "acqrel" (A/B); default: release + acquire fence
  }();
  const int want = NS * C >= 256 ? 32 : 64;
  const int RB = (R + want - 1) / want > 0 ? (R + want - 1) / want : 1;
  const int R2 = (R + RB - 1) / RB;
  hipLaunchKernelGGL((rn_sums_tail_kernel<NS, INTERLEAVED, Tail>), dim3((C + 63) / 64, R2), dim3(1024), 0, s, in, R, C, RB, part2, tickets,
                     mode, tail);
  return check_launch();
}

// in [R][W] fp32 -> ws (doubles) [R2][W]; returns R2
int rn_rows_reduce(const float *in, int R, int W, double *ws, hipStream_t s) {
  // 32 block rows for wide inputs; narrow ones (the stem's 8 / 16 columns) get 64 so that more than a handful of blocks run
  const int want = W >= 256 ? 32 : 64;
  const int RB = (R + want - 1) / want > 0 ? (R + want - 1) / want : 1;
  const int R2 = (R + RB - 1) / RB;
  hipLaunchKernelGGL(rn_rows_reduce_kernel, dim3((W + 63) / 64, R2), dim3(1024), 0, s, in, R, W, RB, ws);
  return R2;
}

int rn_zero_tickets(unsigned *tickets, int sets, hipStream_t s) {
  if (hipMemsetAsync(tickets, 0, (size_t)sets * RN_TICKET_BYTES, s) != hipSuccess) {
    g_last_hip_error = (int)hipGetLastError();
    return CRW_EHIP;
  }
  return CRW_OK;
}

int launch_rn_bn_stats(const float *part, int R, int C, double count, const float *gamma, const float *beta, float *run_mean,
                       float *run_var, float momentum, float eps, float *coef, double *ws, unsigned *tickets, hipStream_t s) {
  const StatsTail tail{count, gamma, beta, run_mean, run_var, momentum, eps, coef};
  return rn_sums_tail<2, true>(part, R, C, ws, tickets, tail, s);
}

int launch_rn_bn_apply(const float *Z, const float *coef, const float *Zd, const float *coef_d, const uint16_t *res_hi,
                       const uint16_t *res_lo, int P, int Ppad, int npix, int C, int relu, uint16_t *y_hi, uint16_t *y_lo, hipStream_t s) {
  const long real = (long)P * npix * C / 8, total = (long)Ppad * npix * C / 8;
  hipLaunchKernelGGL(rn_bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, Z, coef, Zd, coef_d, res_hi, res_lo, real, total, C, relu,
                     y_hi, y_lo);
  return check_launch();
}

int launch_rn_bn_pool(const float *Z, const float *coef, int P, int Ppad, int H, int W, int C, uint16_t *y_hi, uint16_t *y_lo,
                      uint8_t *amax, hipStream_t s) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)Ppad * Ho * Wo * C / 8;
  hipLaunchKernelGGL(rn_bn_pool_kernel, dim3(grid_for(total)), dim3(256), 0, s, Z, coef, P, total, H, W, Ho, Wo, C, y_hi, y_lo, amax);
  return check_launch();
}

size_t rn_bn_bwd_ws_bytes(int P, int npix, int C) {
  const long rows = (long)P * npix;
  const long nblk = (rows + 255) / 256 + 1;
  return align_up((size_t)nblk * 3 * C * 4, 256) + (size_t)64 * 3 * C * 8 + (size_t)3 * C * 4 + 256;
}

int launch_rn_bn_bwd(const float *g1, const float *g2, const uint16_t *mask_hi, const float *Z, const float *coef, const float *Zd,
                     const float *coef_d, int P, int Ppad, int npix, int C, uint16_t *dz_hi, uint16_t *dz_lo, uint16_t *dzd_hi,
                     uint16_t *dzd_lo, float *g_out, float *dgamma, float *dbeta, float *dgamma_d, float *dbeta_d, void *ws,
                     unsigned *tickets, hipStream_t s, const float *ext_part, int ext_rows) {
  if (C % 8 || C > 2048 || 256 % (C / 8 > 256 ? 256 : C / 8)) return CRW_EINVAL;
  const long rows = (long)P * npix;
  int rpb = (int)((rows + 2047) / 2048);  // ~2048 blocks, at least 256 rows each (the workspace holds rows / 256 + 1 partials)
  if (rpb < 256) rpb = 256;
  const int nblk = (int)((rows + rpb - 1) / rpb);
  const int NS = Zd ? 3 : 2;
  float *part = (float *)ws;
  double *part2 = (double *)((char *)ws + align_up((size_t)((rows + 255) / 256 + 1) * 3 * C * 4, 256));
  float *kc = (float *)(part2 + (size_t)64 * 3 * C);
  const size_t lds = (size_t)256 * 8 * NS * 4;
  if (C / 8 > 256) return CRW_EINVAL;
  const float *sums = ext_part;  // the sums came out of the epilogue of the product that made g (resnet_gemm.hip): only the merge is left
  int srows = ext_rows;
  if (!ext_part) {
    if (NS == 3)
      hipLaunchKernelGGL(rn_bn_bwd_reduce_kernel<3>, dim3(nblk), dim3(256), lds, s, g1, g2, mask_hi, Z, coef, Zd, coef_d, rows, rpb, C, part);
    else
      hipLaunchKernelGGL(rn_bn_bwd_reduce_kernel<2>, dim3(nblk), dim3(256), lds, s, g1, g2, mask_hi, Z, coef, Zd, coef_d, rows, rpb, C, part);
    sums = part;
    srows = nblk;
  }
  {
    const BwdTail<3> t3{(double)rows, dgamma, dbeta, dgamma_d, dbeta_d, kc};
    const BwdTail<2> t2{(double)rows, dgamma, dbeta, dgamma_d, dbeta_d, kc};
    const int st = NS == 3 ? rn_sums_tail<3, false>(sums, srows, C, part2, tickets, t3, s) : rn_sums_tail<2, false>(sums, srows, C, part2, tickets, t2, s);
    if (st != CRW_OK) return st;
  }
  const long real = rows * C / 8, total = (long)Ppad * npix * C / 8;
  hipLaunchKernelGGL(rn_bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, g1, g2, mask_hi, Z, coef, Zd, coef_d, kc, real, total,
                     C, dz_hi, dz_lo, dzd_hi, dzd_lo, g_out);
  return check_launch();
}

size_t rn_pool_bwd_ws_bytes(int P, int H, int W, int C) { return rn_bn_bwd_ws_bytes(P, H * W, C); }

int launch_rn_pool_bwd(const float *d1, const float *d2, const uint8_t *amax, const float *Z, const float *coef, int P, int Ppad, int H,
                       int W, int C, uint16_t *dz_hi, uint16_t *dz_lo, float *dgamma, float *dbeta, void *ws, unsigned *tickets, hipStream_t s) {
  if (C % 8 || C / 8 > 256 || 256 % (C / 8)) return CRW_EINVAL;
  const long rows = (long)P * H * W;
  int rpb = (int)((rows + 2047) / 2048);
  if (rpb < 256) rpb = 256;
  const int nblk = (int)((rows + rpb - 1) / rpb);
  float *part = (float *)ws;
  double *part2 = (double *)((char *)ws + align_up((size_t)((rows + 255) / 256 + 1) * 3 * C * 4, 256));
  float *kc = (float *)(part2 + (size_t)64 * 3 * C);
  hipLaunchKernelGGL(rn_pool_bwd_reduce_kernel, dim3(nblk), dim3(256), (size_t)256 * 8 * 2 * 4, s, d1, d2, amax, Z, coef, rows, rpb, H, W, C,
                     part);
  {
    const BwdTail<2> t2{(double)rows, dgamma, dbeta, nullptr, nullptr, kc};
    const int st = rn_sums_tail<2, false>(part, nblk, C, part2, tickets, t2, s);
    if (st != CRW_OK) return st;
  }
  const long real = rows * C / 8, total = (long)Ppad * H * W * C / 8;
  hipLaunchKernelGGL(rn_pool_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, d1, d2, amax, Z, coef, kc, real, total, H, W, C, dz_hi,
                     dz_lo);
  return check_launch();
}

constexpr int STEM_BLOCKS = 1024;
size_t rn_stem_ws_bytes() { return (size_t)STEM_BLOCKS * 16 * 4 + (size_t)64 * 16 * 8 + 256; }

// bn0's batch statistics from the moments of the patches -> the stem record (no map is written)
int launch_rn_stem_stats(const float *x, int P, int cin, int h, int w, const float *w0, const float *b0, const float *gamma,
                         const float *beta, float *run_mean, float *run_var, float momentum, float eps, float *stem, void *ws,
                         hipStream_t s) {
  float *part = (float *)ws;
  double *part2 = (double *)((char *)ws + (size_t)STEM_BLOCKS * 16 * 4);
  hipLaunchKernelGGL(rn_stem_moments_kernel, dim3(STEM_BLOCKS), dim3(256), 0, s, x, P, cin, h * w, part);
  const int R2 = rn_rows_reduce(part, STEM_BLOCKS, 8, part2, s);
  hipLaunchKernelGGL(rn_stem_finalize_kernel, dim3(1), dim3(64), 0, s, part2, R2, cin, (double)P * (h + 2) * (w + 2), w0, b0, gamma, beta,
                     run_mean, run_var, momentum, eps, stem);
  return check_launch();
}

int launch_rn_stem_fwd(const float *x, int P, int Ppad, int cin, int h, int w, int Hm, int Wm, const float *w0, const float *b0,
                       const float *gamma, const float *beta, float *run_mean, float *run_var, float momentum, float eps,
                       uint16_t *m_hi, uint16_t *m_lo, float *stem, void *ws, hipStream_t s) {
  CRW_TRY(launch_rn_stem_stats(x, P, cin, h, w, w0, b0, gamma, beta, run_mean, run_var, momentum, eps, stem, ws, s));
  const long total_px = (long)Ppad * Hm * Wm;
  hipLaunchKernelGGL(rn_stem_apply_kernel, dim3(grid_for(total_px)), dim3(256), 0, s, x, stem, P, total_px, cin, h, w, Hm, Wm, m_hi, m_lo);
  return check_launch();
}

// the map planes from a stem record that is already there (eval mode: launch_rn_stem_eval)
int launch_rn_stem_apply(const float *x, int P, int Ppad, int cin, int h, int w, int Hm, int Wm, const float *stem, uint16_t *m_hi,
                         uint16_t *m_lo, hipStream_t s) {
  const long total_px = (long)Ppad * Hm * Wm;
  hipLaunchKernelGGL(rn_stem_apply_kernel, dim3(grid_for(total_px)), dim3(256), 0, s, x, stem, P, total_px, cin, h, w, Hm, Wm, m_hi, m_lo);
  return check_launch();
}

int launch_rn_stem_bwd(const float *dX0, const float *x, const float *stem, const float *w0, const float *b0, int P, int cin, int h, int w,
                       int ldx, float *dw0, float *db0, float *dgamma, float *dbeta, void *ws, hipStream_t s) {
  float *part = (float *)ws;
  double *part2 = (double *)((char *)ws + (size_t)STEM_BLOCKS * 16 * 4);
  hipLaunchKernelGGL(rn_stem_bwd_reduce_kernel, dim3(STEM_BLOCKS), dim3(256), 0, s, dX0, x, stem, w0, b0, P, cin, h, w, ldx, part);
  const int R2 = rn_rows_reduce(part, STEM_BLOCKS, 16, part2, s);
  hipLaunchKernelGGL(rn_stem_bwd_finalize_kernel, dim3(1), dim3(64), 0, s, part2, R2, cin, stem, w0, b0, dw0, db0, dgamma, dbeta);
  return check_launch();
}

// the stem's backward sums as per-wave partials part [rows][16] (rn_stem_bwd_kernel) -> dw0, db0, dgamma0, dbeta0
int launch_rn_stem_bwd_finalize(const float *part, int rows, int cin, const float *stem, const float *w0, const float *b0, float *dw0,
                                float *db0, float *dgamma, float *dbeta, void *ws, hipStream_t s) {
  double *part2 = (double *)ws;
  const int R2 = rn_rows_reduce(part, rows, 16, part2, s);
  hipLaunchKernelGGL(rn_stem_bwd_finalize_kernel, dim3(1), dim3(64), 0, s, part2, R2, cin, stem, w0, b0, dw0, db0, dgamma, dbeta);
  return check_launch();
}

int launch_rn_pack_conv(const float *w, int cout, int cin, int T, uint16_t *fh, uint16_t *fl, uint16_t *bh, uint16_t *bl, hipStream_t s) {
  hipLaunchKernelGGL(rn_pack_conv_kernel, dim3(grid_for((long)cout * cin * T)), dim3(256), 0, s, w, cout, cin, T, fh, fl, bh, bl);
  return check_launch();
}

int launch_rn_pack_all(RnPackJobs &jobs, hipStream_t s) {
  if (jobs.n < 1 || jobs.n > RN_MAX_PACK_JOBS) return CRW_EINVAL;
  int blocks = 0;
  size_t lds = 0;
  for (int i = 0; i < jobs.n; ++i) {
    RnPackJob &q = jobs.job[i];
    const int cb = q.T <= 16 ? 64 : 8;
    if (q.T < 1 || q.T > 64 || q.cout % RN_PACK_CO || q.cin % cb || (((uintptr_t)q.fh | (uintptr_t)q.fl | (uintptr_t)q.bh | (uintptr_t)q.bl) & 15))
      return CRW_EINVAL;
    q.first_block = blocks;
    blocks += (q.cout / RN_PACK_CO) * (q.cin / cb);
    lds = std::max(lds, (size_t)RN_PACK_CO * cb * q.T * sizeof(float));
  }
  hipLaunchKernelGGL(rn_pack_all_kernel, dim3(blocks), dim3(256), lds, s, jobs);
  return check_launch();
}

int rn_stem_cols(int w) { return round_up(3 * (w + 2), 64); }

int launch_rn_pack_stem(const float *w1, int H0, int W0, int H1, int W1, int ldt, int ncols, uint16_t *fh, uint16_t *fl, uint16_t *th,
                        uint16_t *tl, hipStream_t s) {
  hipLaunchKernelGGL(rn_pack_stem_kernel, dim3(grid_for(64 * 256 + (long)H0 * ncols * ldt)), dim3(256), 0, s, w1, H0, W0, H1, W1, ldt, ncols,
                     fh, fl, th, tl);
  return check_launch();
}

int launch_rn_tapsum(const float *src, long n, int T, float scale, float *dw, hipStream_t s) {
  hipLaunchKernelGGL(rn_tapsum_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, n, T, scale, dw);
  return check_launch();
}

int launch_rn_bn_coef_eval(const float *gamma, const float *beta, const float *run_mean, const float *run_var, float eps, int C, float *coef,
                           hipStream_t s) {
  hipLaunchKernelGGL(rn_bn_coef_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, run_mean, run_var, eps, C, coef);
  return check_launch();
}

int launch_rn_stem_eval(int cin, const float *w0, const float *b0, const float *gamma, const float *beta, const float *run_mean,
                        const float *run_var, float eps, float *stem, hipStream_t s) {
  hipLaunchKernelGGL(rn_stem_eval_kernel, dim3(1), dim3(64), 0, s, cin, w0, b0, gamma, beta, run_mean, run_var, eps, stem);
  return check_launch();
}

int launch_rn_split(const float *x, long rows, long rows_pad, int C, uint16_t *hi, uint16_t *lo, hipStream_t s) {
  hipLaunchKernelGGL(rn_split_kernel, dim3(grid_for(rows_pad * C / 8)), dim3(256), 0, s, x, rows * C / 8, rows_pad * C / 8, hi, lo);
  return check_launch();
}

size_t rn_colsum_ws_bytes(int W) { return (size_t)64 * W * 8 + 256; }
int launch_rn_colsum(const float *x, int R, int W, float *out, void *ws, hipStream_t s) {
  const int R2 = rn_rows_reduce(x, R, W, (double *)ws, s);
  hipLaunchKernelGGL(rn_colsum_finalize_kernel, dim3((W + 255) / 256), dim3(256), 0, s, (const double *)ws, R2, W, out);
  return check_launch();
}

}  // namespace crw
