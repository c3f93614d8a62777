// HBM-bound kernels around the chain: dual (row + column) softmax of the affinity logits, the
// cycle-consistency loss rows, their backward, and small fills/copies.
//
// Layout: logits A are dense [B][T-1][N][N] (the tensor returned to the caller); every internal
// matrix (F, Gt, Lt, R, At, dF, dGt, ...) is [nmat][Np][Np] zero padded (crw_common.h), in [j][b] order.
//   F  = row-softmax(A)                         (src/model.py:44, softmax(current, dim=-1) on A_j)
//   Gt = column-softmax(A), same layout as A    (row-softmax of A_j^T, the flipped half of the palindrome, src/model.py:31)
//
// Traffic plan (what matters at large N, where every N x N pass is HBM time):
//   * the four softmax statistics (row max / sum, column max / sum) either arrive with A from the affinity kernel's
//     epilogue (crw_affinity_fwd: no extra pass over A) or are computed here in one pass;
//   * ONE write pass reads A once and emits exactly what the chain GEMM consumes: fp32 F / Gt for the fp32 chain, bf16
//     hi (+ lo) images for the bf16 chains -- no fp32 copies that are only re-read to be converted;
//   * the backward never reads F / Gt: it recomputes them from A and the statistics (A is half the bytes of F + Gt);
//   * a wave owns a row and moves 16 bytes per lane; the column kernels give a thread a column (coalesced rows).
#include "crw_common.h"

namespace crw {
namespace {

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// caller's matrix index of internal matrix `mat` ([j][b] -> [b][j])
__device__ inline long caller_mat(long mat, int B, int Tm1) { return (mat % B) * Tm1 + mat / B; }

// 4 consecutive logits of a dense row (row start is only 4-byte aligned when N % 4 != 0); elements >= N read as -inf
template <bool VEC>
__device__ inline float4 load_a4(const float *__restrict__ row, int c, int N) {
  if (VEC) {  // N % 4 == 0: c + 3 < N whenever c < N
    if (c < N) return *reinterpret_cast<const float4 *>(row + c);
    return float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  }
  float4 v;
  v.x = c < N ? row[c] : -INFINITY;
  v.y = c + 1 < N ? row[c + 1] : -INFINITY;
  v.z = c + 2 < N ? row[c + 2] : -INFINITY;
  v.w = c + 3 < N ? row[c + 3] : -INFINITY;
  return v;
}

__device__ inline void store_img4(uint16_t *__restrict__ hi, uint16_t *__restrict__ lo, long o, float4 v) {
  const uint16_t h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
  *reinterpret_cast<uint2 *>(hi + o) = uint2{(uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16)};
  if (lo) {
    const uint16_t l0 = f2bf(v.x - bf2f(h0)), l1 = f2bf(v.y - bf2f(h1)), l2 = f2bf(v.z - bf2f(h2)), l3 = f2bf(v.w - bf2f(h3));
    *reinterpret_cast<uint2 *>(lo + o) = uint2{(uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16)};
  }
}

// ---- forward statistics (only when the caller did not get them from the affinity kernel) --------------------------
// blocks [0, nrow): one wave per row -> rmax, rsum.  Other blocks: a stripe of 256 columns, one thread per column, the
// rows in one pass with an online (max, sum) pair -> cmax, csum.
template <bool VEC>
__global__ __launch_bounds__(256) void stats_kernel(const float *__restrict__ A, int B, int Tm1, int N, int Np, int nrow,
                                                    float *__restrict__ rmax, float *__restrict__ rsum,
                                                    float *__restrict__ cmax, float *__restrict__ csum) {
  const long mat = blockIdx.y;
  const float *a = A + caller_mat(mat, B, Tm1) * N * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < nrow) {  // block-uniform
    const int row = blockIdx.x * 4 + wave;
    if (row >= N) return;
    const float *r = a + (long)row * N;
    float m = -INFINITY;
    for (int c = 4 * lane; c < N; c += 256) {
      const float4 v = load_a4<VEC>(r, c, N);
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    m = wave_max(m);
    float s = 0.f;
    for (int c = 4 * lane; c < N; c += 256) {
      const float4 v = load_a4<VEC>(r, c, N);
      s += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));  // exp(-inf) = 0 for the masked tail
    }
    s = wave_sum(s);
    if (lane == 0) {
      rmax[mat * Np + row] = m;
      rsum[mat * Np + row] = s;
    }
    return;
  }
  const int col = ((int)blockIdx.x - nrow) * 256 + threadIdx.x;
  if (col >= N) return;
  float m = -INFINITY, s = 0.f;
  for (int r = 0; r < N; ++r) {
    const float v = a[(long)r * N + col];
    if (v > m) {
      s = s * expf(m - v) + 1.f;  // (first row: s = 0 * exp(-inf) + 1)
      m = v;
    } else {
      s += expf(v - m);
    }
  }
  cmax[mat * Np + col] = m;
  csum[mat * Np + col] = s;
}

// statistics delivered by crw_affinity_fwd: dense [4][B][T-1][N] in the caller's matrix order -> internal [4][nmat][Np]
__global__ __launch_bounds__(256) void import_stats_kernel(const float *__restrict__ ext, int B, int Tm1, int N, int Np,
                                                           float *__restrict__ st) {
  const long mat = blockIdx.y, amat = caller_mat(mat, B, Tm1), nmat = (long)B * Tm1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) st[(k * nmat + mat) * Np + i] = ext[(k * nmat + amat) * N + i];
}

// one pass over A: F / Gt as fp32 planes (fp32 chain) and / or bf16 hi (+ lo) images (bf16 chains), padded
template <bool VEC>
__global__ __launch_bounds__(256) void softmax_write_kernel(const float *__restrict__ A, int B, int Tm1, int N, int Np,
                                                            const float *__restrict__ rmax, const float *__restrict__ rsum,
                                                            const float *__restrict__ cmax, const float *__restrict__ csum,
                                                            float *__restrict__ F, float *__restrict__ Gt,
                                                            uint16_t *__restrict__ Fh, uint16_t *__restrict__ Fl,
                                                            uint16_t *__restrict__ Gh, uint16_t *__restrict__ Gl) {
  const long mat = blockIdx.y;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Np) return;
  const float *r = A + caller_mat(mat, B, Tm1) * N * N + (long)row * N;
  const long obase = mat * Np * Np + (long)row * Np;
  const bool live = row < N;
  const float rm = live ? rmax[mat * Np + row] : 0.f, rs = live ? rsum[mat * Np + row] : 1.f;
  for (int c = 4 * lane; c < Np; c += 256) {
    float4 f = float4{0.f, 0.f, 0.f, 0.f}, g = f;
    if (live && c < N) {
      const float4 a = load_a4<VEC>(r, c, N);
      const float4 cm = *reinterpret_cast<const float4 *>(cmax + mat * Np + c);
      const float4 cs = *reinterpret_cast<const float4 *>(csum + mat * Np + c);
      // (columns >= N of a partial quad: a = -inf -> exp = 0; their statistics are never written: guard the division)
      f = float4{expf(a.x - rm) / rs, expf(a.y - rm) / rs, expf(a.z - rm) / rs, expf(a.w - rm) / rs};
      g.x = expf(a.x - cm.x) / cs.x;
      g.y = c + 1 < N ? expf(a.y - cm.y) / cs.y : 0.f;
      g.z = c + 2 < N ? expf(a.z - cm.z) / cs.z : 0.f;
      g.w = c + 3 < N ? expf(a.w - cm.w) / cs.w : 0.f;
    }
    if (F) {
      *reinterpret_cast<float4 *>(F + obase + c) = f;
      *reinterpret_cast<float4 *>(Gt + obase + c) = g;
    }
    if (Fh) {
      store_img4(Fh, Fl, obase + c, f);
      store_img4(Gh, Gl, obase + c, g);
    }
  }
}

// ---- backward: dA = F (dF - <F, dF>_row) + Gt (dGt - <Gt, dGt>_col), F / Gt recomputed from A and the statistics ------
// row dots (one wave per row) and column dots (256-column stripes, the 4 waves split the rows) in one launch
template <bool VEC>
__global__ __launch_bounds__(256) void dots_kernel(const float *__restrict__ A, int B, int Tm1, int N, int Np, int nrow,
                                                   const float *__restrict__ rmax, const float *__restrict__ rsum,
                                                   const float *__restrict__ cmax, const float *__restrict__ csum,
                                                   const float *__restrict__ dF, const float *__restrict__ dGt,
                                                   float *__restrict__ rdot, float *__restrict__ cdot) {
  __shared__ float part[4][64];
  const long mat = blockIdx.y;
  const float *a = A + caller_mat(mat, B, Tm1) * N * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < nrow) {  // block-uniform
    const int row = blockIdx.x * 4 + wave;
    if (row >= N) return;
    const float *r = a + (long)row * N, *d = dF + mat * Np * Np + (long)row * Np;
    const float rm = rmax[mat * Np + row];
    float s = 0.f;
    for (int c = 4 * lane; c < N; c += 256) {
      const float4 v = load_a4<VEC>(r, c, N);
      const float4 dv = *reinterpret_cast<const float4 *>(d + c);  // padded plane: always in range
      s += (expf(v.x - rm) * dv.x + expf(v.y - rm) * dv.y) + (expf(v.z - rm) * dv.z + expf(v.w - rm) * dv.w);
    }
    s = wave_sum(s);
    if (lane == 0) rdot[mat * Np + row] = s / rsum[mat * Np + row];
    return;
  }
  // 64 columns per block, one lane per column, waves take rows wave, wave + 4, ...
  const int col = ((int)blockIdx.x - nrow) * 64 + lane;
  float s = 0.f;
  if (col < N) {
    const float cm = cmax[mat * Np + col];
    const float *d = dGt + mat * Np * Np + col;
    for (int r = wave; r < N; r += 4) s += expf(a[(long)r * N + col] - cm) * d[(long)r * Np];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) cdot[mat * Np + col] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / csum[mat * Np + col];
}

template <bool VEC>
__global__ __launch_bounds__(256) void softmax_bwd_write_kernel(const float *__restrict__ A, int B, int Tm1, int N, int Np,
                                                                const float *__restrict__ rmax, const float *__restrict__ rsum,
                                                                const float *__restrict__ cmax, const float *__restrict__ csum,
                                                                const float *__restrict__ dF, const float *__restrict__ dGt,
                                                                const float *__restrict__ rdot, const float *__restrict__ cdot,
                                                                float *__restrict__ dA) {
  const long mat = blockIdx.y, amat = caller_mat(mat, B, Tm1);
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float *r = A + amat * N * N + (long)row * N;
  float *o = dA + amat * N * N + (long)row * N;
  const long pbase = mat * Np * Np + (long)row * Np;
  const float rm = rmax[mat * Np + row], rs = rsum[mat * Np + row], rd = rdot[mat * Np + row];
  for (int c = 4 * lane; c < N; c += 256) {
    const float4 a = load_a4<VEC>(r, c, N);
    const float4 df = *reinterpret_cast<const float4 *>(dF + pbase + c), dg = *reinterpret_cast<const float4 *>(dGt + pbase + c);
    const float4 cm = *reinterpret_cast<const float4 *>(cmax + mat * Np + c), cs = *reinterpret_cast<const float4 *>(csum + mat * Np + c);
    const float4 cd = *reinterpret_cast<const float4 *>(cdot + mat * Np + c);
    float4 v;
    v.x = expf(a.x - rm) / rs * (df.x - rd) + expf(a.x - cm.x) / cs.x * (dg.x - cd.x);
    v.y = expf(a.y - rm) / rs * (df.y - rd) + expf(a.y - cm.y) / cs.y * (dg.y - cd.y);
    v.z = expf(a.z - rm) / rs * (df.z - rd) + expf(a.z - cm.z) / cs.z * (dg.z - cd.z);
    v.w = expf(a.w - rm) / rs * (df.w - rd) + expf(a.w - cm.w) / cs.w * (dg.w - cd.w);
    if (VEC) {
      *reinterpret_cast<float4 *>(o + c) = v;
    } else {
      o[c] = v.x;
      if (c + 1 < N) o[c + 1] = v.y;
      if (c + 2 < N) o[c + 2] = v.z;
      if (c + 3 < N) o[c + 3] = v.w;
    }
  }
}

// ---- fills / copies -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void identity_kernel(float *__restrict__ R, uint16_t *__restrict__ Rb, int Np,
                                                       int N, float *__restrict__ cdst, const float *__restrict__ csrc) {
  const long mat = blockIdx.y;
  const long per = (long)Np * Np;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < per; idx += (long)gridDim.x * 256) {
    const int i = idx / Np, j = idx % Np;
    const float v = (i == j && i < N) ? 1.f : 0.f;
    if (R) R[mat * per + idx] = v;
    if (Rb) Rb[mat * per + idx] = f2bf(v);
    if (cdst) cdst[mat * per + idx] = csrc[mat * per + idx];  // (the walk's Lt_1 = Gt_0 rides along: one launch fewer)
  }
}

__global__ __launch_bounds__(256) void copy_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                   long dst_bs, long src_bs, long n4) {
  const long b = blockIdx.y;
  const float4 *s = reinterpret_cast<const float4 *>(src + b * src_bs);
  float4 *d = reinterpret_cast<float4 *>(dst + b * dst_bs);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) d[i] = s[i];
}

// ---- loss ---------------------------------------------------------------------------------
// term[d] = logsumexp_c At[d,c] - At[d,d]   (cross_entropy(At^T, I) per sample, src/model.py:45); At is a padded plane
// (zeros beyond N), one wave per row, 16 bytes per lane
__global__ __launch_bounds__(256) void loss_rows_kernel(const float *__restrict__ At, int N, int Np,
                                                        float *__restrict__ lse, float *__restrict__ terms) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  const long mat = blockIdx.y;
  if (row >= Np) return;
  float l = 0.f, t = 0.f;
  if (row < N) {
    const float *a = At + mat * Np * Np + (long)row * Np;
    float m = -INFINITY;
    for (int c = 4 * lane; c < N; c += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(a + c);
      m = fmaxf(m, v.x);
      if (c + 1 < N) m = fmaxf(m, v.y);
      if (c + 2 < N) m = fmaxf(m, v.z);
      if (c + 3 < N) m = fmaxf(m, v.w);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int c = 4 * lane; c < N; c += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(a + c);
      s += expf(v.x - m);
      if (c + 1 < N) s += expf(v.y - m);
      if (c + 2 < N) s += expf(v.z - m);
      if (c + 3 < N) s += expf(v.w - m);
    }
    s = wave_sum(s);
    l = logf(s) + m;
    t = l - a[row];
  }
  if (lane == 0) {
    lse[mat * Np + row] = l;
    terms[mat * Np + row] = t;
  }
}

// deterministic single-block reduction: loss = scale * sum(terms)
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float *__restrict__ terms, long n, float scale,
                                                           float *__restrict__ loss) {
  __shared__ float sh[1024];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) s += terms[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = sh[0] * scale;
}

// dAt[d,c] = gloss * coef * (softmax(At[d,:])[c] - [c == d]) -> fp32 plane (fp32 chain) or bf16 hi (+ lo) images (bf16 chains:
// dAt is only ever a GEMM operand there); one wave per row
__global__ __launch_bounds__(256) void dAt_kernel(const float *__restrict__ At, const float *__restrict__ lse,
                                                  const float *__restrict__ gloss, float coef, int N, int Np,
                                                  float *__restrict__ dAt, uint16_t *__restrict__ dh, uint16_t *__restrict__ dl) {
  const long mat = blockIdx.y;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Np) return;
  const long base = mat * Np * Np + (long)row * Np;
  const float g = gloss[0] * coef;
  const bool live = row < N;
  const float ls = live ? lse[mat * Np + row] : 0.f;
  for (int c = 4 * lane; c < Np; c += 256) {
    float4 v = float4{0.f, 0.f, 0.f, 0.f};
    if (live && c < N) {
      const float4 a = *reinterpret_cast<const float4 *>(At + base + c);
      v.x = g * (expf(a.x - ls) - (c == row ? 1.f : 0.f));
      v.y = c + 1 < N ? g * (expf(a.y - ls) - (c + 1 == row ? 1.f : 0.f)) : 0.f;
      v.z = c + 2 < N ? g * (expf(a.z - ls) - (c + 2 == row ? 1.f : 0.f)) : 0.f;
      v.w = c + 3 < N ? g * (expf(a.w - ls) - (c + 3 == row ? 1.f : 0.f)) : 0.f;
    }
    if (dAt) *reinterpret_cast<float4 *>(dAt + base + c) = v;
    if (dh) store_img4(dh, dl, base + c, v);
  }
}

// At [K][B][Np][Np] -> out [B][K][N][N]
__global__ __launch_bounds__(256) void unpad_At_kernel(const float *__restrict__ At, int K, int B, int N, int Np,
                                                       float *__restrict__ out) {
  const int k = blockIdx.y % K, b = blockIdx.y / K;
  const long per = (long)N * N;
  const float *src = At + ((long)k * B + b) * Np * Np;
  float *dst = out + ((long)b * K + k) * per;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < per; idx += (long)gridDim.x * 256) {
    const int d = idx / N, c = idx % N;
    dst[idx] = src[(long)d * Np + c];
  }
}

inline int ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

// stats_ext: optional dense [4][B][T-1][N] statistics from crw_affinity_fwd (else computed here); stats: internal [4][nmat][Np]
int launch_softmax_fwd(const float *A, const float *stats_ext, int B, int Tm1, int N, int Np, float *F, float *Gt, void *Fh,
                       void *Fl, void *Gh, void *Gl, float *stats, hipStream_t s) {
  const int nmat = B * Tm1;
  float *rmax = stats, *rsum = stats + (long)nmat * Np, *cmax = stats + 2L * nmat * Np, *csum = stats + 3L * nmat * Np;
  const bool vec = (N % 4) == 0 && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  if (stats_ext) {
    hipLaunchKernelGGL(import_stats_kernel, dim3((N + 255) / 256, nmat), dim3(256), 0, s, stats_ext, B, Tm1, N, Np, stats);
  } else {
    const dim3 grid((N + 3) / 4 + (N + 255) / 256, nmat);
    if (vec) hipLaunchKernelGGL(stats_kernel<true>, grid, dim3(256), 0, s, A, B, Tm1, N, Np, (N + 3) / 4, rmax, rsum, cmax, csum);
    else hipLaunchKernelGGL(stats_kernel<false>, grid, dim3(256), 0, s, A, B, Tm1, N, Np, (N + 3) / 4, rmax, rsum, cmax, csum);
  }
  const dim3 grid(Np / 4, nmat);
  if (vec)
    hipLaunchKernelGGL(softmax_write_kernel<true>, grid, dim3(256), 0, s, A, B, Tm1, N, Np, rmax, rsum, cmax, csum, F, Gt,
                       (uint16_t *)Fh, (uint16_t *)Fl, (uint16_t *)Gh, (uint16_t *)Gl);
  else
    hipLaunchKernelGGL(softmax_write_kernel<false>, grid, dim3(256), 0, s, A, B, Tm1, N, Np, rmax, rsum, cmax, csum, F, Gt,
                       (uint16_t *)Fh, (uint16_t *)Fl, (uint16_t *)Gh, (uint16_t *)Gl);
  return check_launch();
}

int launch_stats_dense(const float *A, int nmat, int N, float *stats, hipStream_t s) {
  // stats_kernel with B = nmat, T-1 = 1 maps internal matrix i to caller matrix i, and Np = N makes the rows dense
  float *rmax = stats, *rsum = stats + (long)nmat * N, *cmax = stats + 2L * nmat * N, *csum = stats + 3L * nmat * N;
  const dim3 grid((N + 3) / 4 + (N + 255) / 256, nmat);
  if ((N % 4) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0)
    hipLaunchKernelGGL(stats_kernel<true>, grid, dim3(256), 0, s, A, nmat, 1, N, N, (N + 3) / 4, rmax, rsum, cmax, csum);
  else
    hipLaunchKernelGGL(stats_kernel<false>, grid, dim3(256), 0, s, A, nmat, 1, N, N, (N + 3) / 4, rmax, rsum, cmax, csum);
  return check_launch();
}

int launch_softmax_bwd(const float *A, const float *stats, const float *dF, const float *dGt, int B, int Tm1, int N, int Np,
                       float *dots, float *dA, hipStream_t s) {
  const int nmat = B * Tm1;
  const float *rmax = stats, *rsum = stats + (long)nmat * Np, *cmax = stats + 2L * nmat * Np, *csum = stats + 3L * nmat * Np;
  float *rdot = dots, *cdot = dots + (long)nmat * Np;
  const bool vec = (N % 4) == 0 && (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(dA)) & 15) == 0);
  const dim3 g1((N + 3) / 4 + (N + 63) / 64, nmat), g2((N + 3) / 4, nmat);
  if (vec) {
    hipLaunchKernelGGL(dots_kernel<true>, g1, dim3(256), 0, s, A, B, Tm1, N, Np, (N + 3) / 4, rmax, rsum, cmax, csum, dF, dGt, rdot, cdot);
    hipLaunchKernelGGL(softmax_bwd_write_kernel<true>, g2, dim3(256), 0, s, A, B, Tm1, N, Np, rmax, rsum, cmax, csum, dF, dGt, rdot, cdot, dA);
  } else {
    hipLaunchKernelGGL(dots_kernel<false>, g1, dim3(256), 0, s, A, B, Tm1, N, Np, (N + 3) / 4, rmax, rsum, cmax, csum, dF, dGt, rdot, cdot);
    hipLaunchKernelGGL(softmax_bwd_write_kernel<false>, g2, dim3(256), 0, s, A, B, Tm1, N, Np, rmax, rsum, cmax, csum, dF, dGt, rdot, cdot, dA);
  }
  return check_launch();
}

int launch_identity(float *R, void *Rb, int batch, int Np, int N, hipStream_t s, float *copy_dst, const float *copy_src) {
  hipLaunchKernelGGL(identity_kernel, dim3(ew_blocks((long)Np * Np), batch), dim3(256), 0, s, R, (uint16_t *)Rb,
                     Np, N, copy_dst, copy_src);
  return check_launch();
}

int launch_copy_f32(float *dst, const float *src, long dst_bs, long src_bs, long n_per_batch, int batch,
                    hipStream_t s) {
  if (n_per_batch % 4) return CRW_EINVAL;
  hipLaunchKernelGGL(copy_kernel, dim3(ew_blocks(n_per_batch / 4), batch), dim3(256), 0, s, dst, src, dst_bs,
                     src_bs, n_per_batch / 4);
  return check_launch();
}

int launch_loss_rows(const float *At, int nmat, int N, int Np, float *lse, float *terms, hipStream_t s) {
  hipLaunchKernelGGL(loss_rows_kernel, dim3((Np + 3) / 4, nmat), dim3(256), 0, s, At, N, Np, lse, terms);
  return check_launch();
}

int launch_loss_reduce(const float *terms, long n, float scale, float *loss, hipStream_t s) {
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, terms, n, scale, loss);
  return check_launch();
}

int launch_dAt(const float *At, const float *lse, const float *gloss, float coef, int nmat, int N, int Np,
               float *dAt, void *dh, void *dl, hipStream_t s) {
  hipLaunchKernelGGL(dAt_kernel, dim3(Np / 4, nmat), dim3(256), 0, s, At, lse, gloss, coef, N, Np, dAt, (uint16_t *)dh,
                     (uint16_t *)dl);
  return check_launch();
}

int launch_unpad_At(const float *At, int K, int B, int N, int Np, float *out, hipStream_t s) {
  hipLaunchKernelGGL(unpad_At_kernel, dim3(ew_blocks((long)N * N), K * B), dim3(256), 0, s, At, K, B, N, Np, out);
  return check_launch();
}

}  // namespace crw
