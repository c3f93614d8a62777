// Shared declarations for the gfx950 kernels of libcrw_hip.so (internal; the public ABI is
// include/crw_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "crw_hip.h"

namespace crw {

constexpr int WAVE = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

extern thread_local int g_last_hip_error;

// HIP keeps a per-thread "last error" that other users of the runtime in this process (PyTorch,
// MIOpen) may leave set; entry points drop it before their first launch so that check_launch()
// only reports errors of our own launches.
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return CRW_EHIP;
  }
  return CRW_OK;
}

#define CRW_TRY(expr)               \
  do {                              \
    int _st = (expr);               \
    if (_st != CRW_OK) return _st;  \
  } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Padded node count: every internal NxN matrix is [Np][Np], zero padded, so the chain GEMM
// needs no edge handling.  Tile candidates are 32/64/128 (gemm_f32.hip).
inline int padded_nodes(int N) {
  if (N <= 128) return round_up(N, 32);
  if (N <= 1024) return round_up(N, 64);
  return round_up(N, 128);
}

// ---- wave-level reductions (64 lanes) -----------------------------------------------------
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- one product of a grouped, batched, zero-padded square GEMM ---------------------------
// C[b] (+)= op(A[b]) * op(B[b]) [+ op(A2[b]) * op(B2[b])], all [n][n] row-major with ld = n.
// op(X) = X^T when the matching flag is set.
// fp32 kernel: A/B/A2/B2 are float*.  bf16 kernel: they are the bf16 "hi" images and Al/Bl/A2l/B2l
// the "lo" images (split-3 only); Cb/Cbl receive bf16 hi/lo images of the result (optional).
struct GemmProb {
  const void *A, *B, *A2, *B2;
  const void *Al, *Bl, *A2l, *B2l;
  float *C;                  // fp32 result (bf16 kernel: optional)
  void *Cb, *Cbl;
  long sA, sB, sA2, sB2, sC; // batch strides in elements
  int ta, tb, ta2, tb2;
  int beta;                  // 1: accumulate into the existing C
};
constexpr int MAX_GROUP = 6;
struct GemmGroup {
  GemmProb p[MAX_GROUP];
  int nprob;
  int n;     // padded size
  int batch; // grid.y
};

int launch_gemm_group_f32(const GemmGroup &g, hipStream_t s);
int launch_gemm_group_bf16(const GemmGroup &g, int split, hipStream_t s);
int launch_split_bf16(const float *x, long count, void *hi, void *lo, hipStream_t s);

// general (bounds-checked) strided GEMM used by the affinity build and its backward:
// C[b][m][n] = alpha_div ? acc / alpha : acc * alpha, acc = sum_k A(m,k) B(k,n) (+ second pair)
struct EdgeOperand {
  const float *p;
  long sb;     // batch stride
  long rs, cs; // strides of the logical (row, col) of the operand as used in the product
};
struct EdgeGemm {
  EdgeOperand A, B, A2, B2; // A2.p == nullptr -> single product
  float *C;
  long sCb, ldc;
  int M, N, K, K2;
  int batch_inner;          // batch index b = blockIdx.z ; (outer, inner) = (b / inner, b % inner)
  long sA_outer, sB_outer, sA2_outer, sB2_outer, sC_outer;
  // per-inner-index enable masks: product 1 is skipped when inner == skip1_inner, product 2 when
  // inner == skip2_inner (used for the first/last frame in the affinity backward)
  int skip1_inner, skip2_inner;
  float scale;
  int divide;               // 1: C = acc / scale, 0: C = acc * scale
};
int launch_edge_gemm(const EdgeGemm &g, int batch, hipStream_t s);
// affinity build on 128 x 128 fp32-MFMA tiles with the softmax statistics in the epilogue (gemm_f32.hip):
// A[b,t] = ehat[b,t] ehat[b,t+1]^T / tau, stats (optional) dense [4][B][T-1][N]; part: workspace of
// affinity_part_floats(B, T, N) floats (per-tile partial statistics)
size_t affinity_part_floats(int B, int T, int N);
int launch_affinity_tiles(const float *ehat, int B, int T, int N, int C, float tau, float *A, float *stats, float *part,
                          hipStream_t s);
// dehat[b,t] = (dA[b,t] ehat[b,t+1] + dA[b,t-1]^T ehat[b,t-1]) / tau on fp32-MFMA tiles (C a multiple of 16, C <= 256)
int launch_affinity_bwd_tiles(const float *dA, const float *ehat, int B, int T, int N, int C, float tau, float *dehat,
                              hipStream_t s);

// softmax.hip --------------------------------------------------------------------------------
// A / dA are the caller's [B][T-1][N][N]; F, Gt, dF, dGt and the statistics are internal [T-1][B][Np][Np] / [..][Np].
// stats_ext (optional): dense [4][B][T-1][N] row max / row sum / column max / column sum from crw_affinity_fwd.
// Outputs of the forward pass are optional per family: fp32 planes (F, Gt) and / or bf16 images (Fh/Fl, Gh/Gl).
int launch_softmax_fwd(const float *A, const float *stats_ext, int B, int Tm1, int N, int Np, float *F, float *Gt, void *Fh,
                       void *Fl, void *Gh, void *Gl, float *stats /* 4*B*Tm1*Np */, hipStream_t s);
// statistics of dense matrices in the caller's order: stats [4][nmat][N] (the layout crw_affinity_fwd hands out)
int launch_stats_dense(const float *A, int nmat, int N, float *stats, hipStream_t s);
// F / Gt are recomputed from A and the statistics; dots: 2*B*Tm1*Np floats of scratch
int launch_softmax_bwd(const float *A, const float *stats, const float *dF, const float *dGt, int B, int Tm1, int N, int Np,
                       float *dots, float *dA, hipStream_t s);
// persistent small-n chain (chain_small.hip): X_{k+1} = P_k X_k  /  Y_k += P_k^T Y_{k+1}
int launch_chain_small_fwd(const float *Gt, const float *F, float *Lt, float *R, int B, int K, int n, hipStream_t s);
int launch_chain_small_bwd(const float *Gt, const float *F, float *dLt, float *dR, int B, int K, int n,
                           hipStream_t s);
// R (fp32) and / or Rb (bf16) = identity on the first N rows; optionally also copy_dst = copy_src (batch * Np * Np floats)
int launch_identity(float *R, void *Rb, int batch, int Np, int N, hipStream_t s, float *copy_dst = nullptr,
                    const float *copy_src = nullptr);
int launch_copy_f32(float *dst, const float *src, long dst_bs, long src_bs, long n_per_batch, int batch,
                    hipStream_t s);
int launch_loss_rows(const float *At, int nmat, int N, int Np, float *lse, float *terms, hipStream_t s);
int launch_loss_reduce(const float *terms, long n, float scale, float *loss, hipStream_t s);
int launch_dAt(const float *At, const float *lse, const float *gloss, float coef, int nmat, int N, int Np,
               float *dAt, void *dAt_hi, void *dAt_lo, hipStream_t s);
int launch_unpad_At(const float *At, int K, int B, int N, int Np, float *out, hipStream_t s);

}  // namespace crw
