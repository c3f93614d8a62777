// HBM-bound kernels around the chain: dual (row + column) softmax of the affinity logits, the
// cycle-consistency loss rows, their backward, and small fills/copies.
//
// Layout: logits A are dense [nmat][N][N] (the tensor returned to the caller); every internal
// matrix (F, Gt, Lt, R, At, dF, dGt, ...) is [nmat][Np][Np] zero padded (crw_common.h).
//   F  = row-softmax(A)                         (src/model.py:44, softmax(current, dim=-1) on A_j)
//   Gt = column-softmax(A), same layout as A    (row-softmax of A_j^T, the flipped half of the palindrome, src/model.py:31)
#include "crw_common.h"

namespace crw {
namespace {

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }

// ---- forward statistics -------------------------------------------------------------------
// one wave per row: rmax[n], rsum[n]
__device__ inline void row_stats_block(int bx, const float *__restrict__ A, int B, int Tm1, int N, int Np,
                                       float *__restrict__ rmax, float *__restrict__ rsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = bx * 4 + wave;
  const long mat = blockIdx.y, amat = (mat % B) * Tm1 + mat / B;  // internal [j][b]  <->  caller's [b][j]
  if (row >= N) return;
  const float *a = A + amat * N * N + (long)row * N;
  float m = -INFINITY;
  for (int c = lane; c < N; c += 64) m = fmaxf(m, a[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += expf(a[c] - m);
  s = wave_sum(s);
  if (lane == 0) {
    rmax[mat * Np + row] = m;
    rsum[mat * Np + row] = s;
  }
}

// one lane per column (64 columns per block), the 4 waves split the rows: cmax[m], csum[m]
__device__ inline void col_stats_block(int bx, const float *__restrict__ A, int B, int Tm1, int N, int Np,
                                       float *__restrict__ cmax, float *__restrict__ csum) {
  __shared__ float sm[4][64], ss[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = bx * 64 + lane;
  const long mat = blockIdx.y, amat = (mat % B) * Tm1 + mat / B;
  const float *a = A + amat * N * N;
  float m = -INFINITY;
  if (col < N)
    for (int r = wave; r < N; r += 4) m = fmaxf(m, a[(long)r * N + col]);
  sm[wave][lane] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
  float s = 0.f;
  if (col < N)
    for (int r = wave; r < N; r += 4) s += expf(a[(long)r * N + col] - m);
  ss[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) {
    cmax[mat * Np + col] = m;
    csum[mat * Np + col] = (ss[0][lane] + ss[1][lane]) + (ss[2][lane] + ss[3][lane]);
  }
}

// both statistics in one launch: blocks [0, nrow) take rows, the rest columns (block-uniform branch)
__global__ __launch_bounds__(256) void stats_kernel(const float *__restrict__ A, int B, int Tm1, int N, int Np, int nrow,
                                                    float *__restrict__ rmax, float *__restrict__ rsum,
                                                    float *__restrict__ cmax, float *__restrict__ csum) {
  if ((int)blockIdx.x < nrow) row_stats_block(blockIdx.x, A, B, Tm1, N, Np, rmax, rsum);
  else col_stats_block((int)blockIdx.x - nrow, A, B, Tm1, N, Np, cmax, csum);
}

// elementwise: F, Gt padded (+ optional bf16 shadows)
__global__ __launch_bounds__(256) void softmax_write_kernel(const float *__restrict__ A, int B, int Tm1, int N, int Np,
                                                            const float *__restrict__ rmax,
                                                            const float *__restrict__ rsum,
                                                            const float *__restrict__ cmax,
                                                            const float *__restrict__ csum, float *__restrict__ F,
                                                            float *__restrict__ Gt, uint16_t *__restrict__ Fb,
                                                            uint16_t *__restrict__ Gtb) {
  const long mat = blockIdx.y, amat = (mat % B) * Tm1 + mat / B;
  const long per = (long)Np * Np;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < per; idx += (long)gridDim.x * 256) {
    const int n = idx / Np, m = idx % Np;
    float f = 0.f, g = 0.f;
    if (n < N && m < N) {
      const float a = A[amat * N * N + (long)n * N + m];
      f = expf(a - rmax[mat * Np + n]) / rsum[mat * Np + n];
      g = expf(a - cmax[mat * Np + m]) / csum[mat * Np + m];
    }
    F[mat * per + idx] = f;
    Gt[mat * per + idx] = g;
    if (Fb) {
      Fb[mat * per + idx] = f2bf(f);
      Gtb[mat * per + idx] = f2bf(g);
    }
  }
}

// ---- backward statistics (padded operands: pads are zero, no bounds needed) -----------------
__device__ inline void row_dot_block(int bx, const float *__restrict__ X, const float *__restrict__ dX, int Np,
                                     float *__restrict__ rdot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = bx * 4 + wave;
  const long mat = blockIdx.y;
  if (row >= Np) return;
  const long off = mat * Np * Np + (long)row * Np;
  float s = 0.f;
  for (int c = lane; c < Np; c += 64) s += X[off + c] * dX[off + c];
  s = wave_sum(s);
  if (lane == 0) rdot[mat * Np + row] = s;
}

__device__ inline void col_dot_block(int bx, const float *__restrict__ X, const float *__restrict__ dX, int Np,
                                     float *__restrict__ cdot) {
  __shared__ float ss[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = bx * 64 + lane;  // Np is a multiple of 32; guard the last half strip
  const long mat = blockIdx.y;
  float s = 0.f;
  if (col < Np)
    for (int r = wave; r < Np; r += 4) {
      const long o = mat * Np * Np + (long)r * Np + col;
      s += X[o] * dX[o];
    }
  ss[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < Np) cdot[mat * Np + col] = (ss[0][lane] + ss[1][lane]) + (ss[2][lane] + ss[3][lane]);
}

// row dots of (F, dF) and column dots of (Gt, dGt) in one launch
__global__ __launch_bounds__(256) void dots_kernel(const float *__restrict__ F, const float *__restrict__ dF,
                                                   const float *__restrict__ Gt, const float *__restrict__ dGt, int Np,
                                                   int nrow, float *__restrict__ rdot, float *__restrict__ cdot) {
  if ((int)blockIdx.x < nrow) row_dot_block(blockIdx.x, F, dF, Np, rdot);
  else col_dot_block((int)blockIdx.x - nrow, Gt, dGt, Np, cdot);
}

// dA = F (dF - rdot[n]) + Gt (dGt - cdot[m])   -> dense [nmat][N][N]
__global__ __launch_bounds__(256) void softmax_bwd_write_kernel(const float *__restrict__ F,
                                                                const float *__restrict__ Gt,
                                                                const float *__restrict__ dF,
                                                                const float *__restrict__ dGt,
                                                                const float *__restrict__ rdot,
                                                                const float *__restrict__ cdot, int B, int Tm1,
                                                                int N, int Np, float *__restrict__ dA) {
  const long mat = blockIdx.y, amat = (mat % B) * Tm1 + mat / B;
  const long per = (long)N * N;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < per; idx += (long)gridDim.x * 256) {
    const int n = idx / N, m = idx % N;
    const long o = mat * Np * Np + (long)n * Np + m;
    dA[amat * per + idx] = F[o] * (dF[o] - rdot[mat * Np + n]) + Gt[o] * (dGt[o] - cdot[mat * Np + m]);
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
// term[d] = logsumexp_c At[d,c] - At[d,d]   (cross_entropy(At^T, I) per sample, src/model.py:45)
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
    for (int c = lane; c < N; c += 64) m = fmaxf(m, a[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < N; c += 64) s += expf(a[c] - m);
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

// dAt[d,c] = gloss * coef * (softmax(At[d,:])[c] - [c == d])
__global__ __launch_bounds__(256) void dAt_kernel(const float *__restrict__ At, const float *__restrict__ lse,
                                                  const float *__restrict__ gloss, float coef, int N, int Np,
                                                  float *__restrict__ dAt, uint16_t *__restrict__ dAtb) {
  const long mat = blockIdx.y;
  const long per = (long)Np * Np;
  const float g = gloss[0] * coef;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < per; idx += (long)gridDim.x * 256) {
    const int d = idx / Np, c = idx % Np;
    float v = 0.f;
    if (d < N && c < N) v = g * (expf(At[mat * per + idx] - lse[mat * Np + d]) - (c == d ? 1.f : 0.f));
    dAt[mat * per + idx] = v;
    if (dAtb) dAtb[mat * per + idx] = f2bf(v);
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

int launch_softmax_fwd(const float *A, int B, int Tm1, int N, int Np, float *F, float *Gt, void *Fb, void *Gtb,
                       float *stats, hipStream_t s) {
  const int nmat = B * Tm1;
  float *rmax = stats, *rsum = stats + (long)nmat * Np, *cmax = stats + 2L * nmat * Np, *csum = stats + 3L * nmat * Np;
  hipLaunchKernelGGL(stats_kernel, dim3((N + 3) / 4 + (N + 63) / 64, nmat), dim3(256), 0, s, A, B, Tm1, N, Np, (N + 3) / 4,
                     rmax, rsum, cmax, csum);
  hipLaunchKernelGGL(softmax_write_kernel, dim3(ew_blocks((long)Np * Np), nmat), dim3(256), 0, s, A, B, Tm1, N, Np, rmax,
                     rsum, cmax, csum, F, Gt, (uint16_t *)Fb, (uint16_t *)Gtb);
  return check_launch();
}

int launch_softmax_bwd(const float *F, const float *Gt, const float *dF, const float *dGt, int B, int Tm1, int N,
                       int Np, float *stats, float *dA, hipStream_t s) {
  const int nmat = B * Tm1;
  float *rdot = stats, *cdot = stats + (long)nmat * Np;
  hipLaunchKernelGGL(dots_kernel, dim3((Np + 3) / 4 + (Np + 63) / 64, nmat), dim3(256), 0, s, F, dF, Gt, dGt, Np,
                     (Np + 3) / 4, rdot, cdot);
  hipLaunchKernelGGL(softmax_bwd_write_kernel, dim3(ew_blocks((long)N * N), nmat), dim3(256), 0, s, F, Gt, dF, dGt,
                     rdot, cdot, B, Tm1, N, Np, dA);
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
               float *dAt, void *dAtb, hipStream_t s) {
  hipLaunchKernelGGL(dAt_kernel, dim3(ew_blocks((long)Np * Np), nmat), dim3(256), 0, s, At, lse, gloss, coef, N, Np,
                     dAt, (uint16_t *)dAtb);
  return check_launch();
}

int launch_unpad_At(const float *At, int K, int B, int N, int Np, float *out, hipStream_t s) {
  hipLaunchKernelGGL(unpad_At_kernel, dim3(ew_blocks((long)N * N), K * B), dim3(256), 0, s, At, K, B, N, Np, out);
  return check_launch();
}

}  // namespace crw
