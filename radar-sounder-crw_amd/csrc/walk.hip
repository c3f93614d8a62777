// Training path of the contrastive random walk: C-ABI entry points and the launch schedules.
//
//   crw_affinity_fwd : normalise rows (wave per row) + batched E_t E_{t+1}^T / tau  (src/model.py:22-26)
//   crw_walk_fwd     : dual softmax, then the chain in prefix form (which the
//                      oracle proves equal to src/model.py:35-46):
//                          Lt_1 = Gt_0          Lt_{k+1} = Gt_k Lt_k       (Lt_k = L_k^T)
//                          R_1  = I             R_{k+1}  = F_k  R_k
//                          At_k = Lt_k^T R_k    loss = (1/N) sum_k mean_{b,d}(lse(At_k[d,:]) - At_k[d,d])
//                      one grouped GEMM launch per k carries the three independent products.
//   crw_walk_bwd     : reverse sweep, one grouped launch (four products, two of them fused
//                      two-term accumulations) per k, then the dual-softmax backward.
//   crw_affinity_bwd : dE_t = (dA_t E_{t+1} + dA_{t-1}^T E_{t-1}) / tau, then normalise-backward.
#include "crw_common.h"

namespace crw {
namespace {

constexpr float EPS_NORM = 1e-12f;  // F.normalize default eps

__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ e, long rows, int C,
                                                        float *__restrict__ ehat, float *__restrict__ norm) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *x = e + row * C;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += x[c] * x[c];
  ss = wave_sum(ss);
  const float d = fmaxf(sqrtf(ss), EPS_NORM);
  for (int c = lane; c < C; c += 64) ehat[row * C + c] = x[c] / d;
  if (lane == 0 && norm) norm[row] = d;
}

// de = (dehat - ehat * <ehat, dehat>) / norm
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float *__restrict__ dehat,
                                                            const float *__restrict__ ehat,
                                                            const float *__restrict__ norm, long rows, int C,
                                                            float *__restrict__ de) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += ehat[row * C + c] * dehat[row * C + c];
  dot = wave_sum(dot);
  const float d = norm[row];
  for (int c = lane; c < C; c += 64) de[row * C + c] = (dehat[row * C + c] - ehat[row * C + c] * dot) / d;
}

struct WalkState {
  float *F, *Gt, *Lt, *R, *At, *lse, *terms, *stats;
  size_t bytes;
};
struct WalkScratch {
  float *dAt, *dF, *dGt, *dLt, *dR, *stats;
  size_t bytes;
};

template <typename T>
T *carve(char *&p, size_t count) {
  T *r = reinterpret_cast<T *>(p);
  p += align_up(count * sizeof(T), 256);
  return r;
}

WalkState layout_state(void *base, int B, int T, int N) {
  const size_t Np = padded_nodes(N), K = T - 2, nA = (size_t)B * (T - 1), M = Np * Np;
  char *p = static_cast<char *>(base);
  WalkState s;
  s.F = carve<float>(p, nA * M);
  s.Gt = carve<float>(p, nA * M);
  s.Lt = carve<float>(p, K * B * M);
  s.R = carve<float>(p, K * B * M);
  s.At = carve<float>(p, K * B * M);
  s.lse = carve<float>(p, K * B * Np);
  s.terms = carve<float>(p, K * B * Np);
  s.stats = carve<float>(p, 4 * nA * Np);
  s.bytes = p - static_cast<char *>(base);
  return s;
}

WalkScratch layout_scratch(void *base, int B, int T, int N) {
  const size_t Np = padded_nodes(N), K = T - 2, nA = (size_t)B * (T - 1), M = Np * Np;
  char *p = static_cast<char *>(base);
  WalkScratch s;
  s.dAt = carve<float>(p, K * B * M);
  s.dF = carve<float>(p, nA * M);
  s.dGt = carve<float>(p, nA * M);
  s.dLt = carve<float>(p, 2 * (size_t)B * M);
  s.dR = carve<float>(p, 2 * (size_t)B * M);
  s.stats = carve<float>(p, 2 * nA * Np);
  s.bytes = p - static_cast<char *>(base);
  return s;
}

bool bad_shape(int B, int T, int N) { return B < 1 || T < 2 || N < 1 || N > 16384; }

}  // namespace
}  // namespace crw

using namespace crw;

extern "C" {

int crw_abi_version(void) { return 1; }
const char *crw_build_arch(void) { return "gfx950"; }
int crw_last_hip_error(void) { return g_last_hip_error; }
int crw_padded_nodes(int N) { return N < 1 ? 0 : padded_nodes(N); }

size_t crw_walk_state_bytes(int B, int T, int N) {
  if (bad_shape(B, T, N) || T < 3) return 256;
  return layout_state(nullptr, B, T, N).bytes;
}
size_t crw_walk_scratch_bytes(int B, int T, int N) {
  if (bad_shape(B, T, N) || T < 3) return 256;
  return layout_scratch(nullptr, B, T, N).bytes;
}

int crw_normalize(const float *emb, int rows, int C, float *ehat, float *norm, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!emb || !ehat || rows < 1 || C < 1) return CRW_EINVAL;
  hipLaunchKernelGGL(normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, (long)rows, C,
                     ehat, norm);
  return check_launch();
}

int crw_affinity_fwd(const float *emb, int B, int T, int N, int C, float tau, float *ehat, float *norm, float *A,
                     crw_stream_t stream) {
  crw::clear_stale_error();
  if (!emb || !ehat || !norm || !A || bad_shape(B, T, N) || C < 1 || !(tau > 0.f)) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  CRW_TRY(crw_normalize(emb, B * T * N, C, ehat, norm, stream));
  EdgeGemm g{};
  const long NC = (long)N * C, NN = (long)N * N;
  g.A = EdgeOperand{ehat, NC, C, 1};            // (n, c) of frame j
  g.B = EdgeOperand{ehat + NC, NC, 1, C};       // (c, m) of frame j+1
  g.C = A;
  g.sCb = NN; g.ldc = N;
  g.M = N; g.N = N; g.K = C;
  g.batch_inner = T - 1;
  g.sA_outer = g.sB_outer = (long)T * NC;
  g.sC_outer = (long)(T - 1) * NN;
  g.skip1_inner = g.skip2_inner = -1;
  g.scale = tau; g.divide = 1;
  return launch_edge_gemm(g, B * (T - 1), s);
}

int crw_walk_fwd(const float *A, int B, int T, int N, int chain, void *state, size_t state_bytes, float *At_out,
                 float *loss, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!A || !loss || bad_shape(B, T, N)) return CRW_EINVAL;
  if (chain != CRW_CHAIN_F32) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (T < 3) {  // no cycle: the reference returns 0/N
    if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) return CRW_EHIP;
    return CRW_OK;
  }
  if (!state) return CRW_EINVAL;
  WalkState st = layout_state(state, B, T, N);
  if (state_bytes < st.bytes) return CRW_EWORKSPACE;
  const int Np = padded_nodes(N), K = T - 2, nA = B * (T - 1);
  const long M = (long)Np * Np;

  CRW_TRY(launch_softmax_fwd(A, nA, N, Np, st.F, st.Gt, nullptr, nullptr, st.stats, s));
  CRW_TRY(launch_copy_f32(st.Lt, st.Gt, M, (long)(T - 1) * M, M, B, s));  // Lt_1 = Gt_0
  CRW_TRY(launch_identity(st.R, nullptr, B, Np, N, s));                   // R_1 = I

  for (int k = 1; k <= K; ++k) {
    GemmGroup g{};
    g.n = Np; g.batch = B;
    float *Lt_k = st.Lt + (long)(k - 1) * B * M, *R_k = st.R + (long)(k - 1) * B * M;
    GemmProb &p0 = g.p[g.nprob++];  // At_k = Lt_k^T R_k
    p0.A = Lt_k; p0.ta = 1; p0.sA = M;
    p0.B = R_k; p0.sB = M;
    p0.C = st.At + (long)(k - 1) * B * M; p0.sC = M;
    if (k < K) {
      GemmProb &p1 = g.p[g.nprob++];  // Lt_{k+1} = Gt_k Lt_k
      p1.A = st.Gt + (long)k * M; p1.sA = (long)(T - 1) * M;
      p1.B = Lt_k; p1.sB = M;
      p1.C = Lt_k + (long)B * M; p1.sC = M;
      GemmProb &p2 = g.p[g.nprob++];  // R_{k+1} = F_k R_k
      p2.A = st.F + (long)k * M; p2.sA = (long)(T - 1) * M;
      p2.B = R_k; p2.sB = M;
      p2.C = R_k + (long)B * M; p2.sC = M;
    }
    CRW_TRY(launch_gemm_group_f32(g, s));
  }
  CRW_TRY(launch_loss_rows(st.At, K * B, N, Np, st.lse, st.terms, s));
  CRW_TRY(launch_loss_reduce(st.terms, (long)K * B * Np, 1.0f / ((float)B * (float)N * (float)N), loss, s));
  if (At_out) CRW_TRY(launch_unpad_At(st.At, K, B, N, Np, At_out, s));
  return CRW_OK;
}

int crw_walk_bwd(const float *gloss, int B, int T, int N, int chain, void *state, size_t state_bytes, void *scratch,
                 size_t scratch_bytes, float *dA, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!gloss || !dA || bad_shape(B, T, N)) return CRW_EINVAL;
  if (chain != CRW_CHAIN_F32) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long NN = (long)N * N;
  if (T < 3) {
    if (hipMemsetAsync(dA, 0, sizeof(float) * B * (T - 1) * NN, s) != hipSuccess) return CRW_EHIP;
    return CRW_OK;
  }
  if (!state || !scratch) return CRW_EINVAL;
  WalkState st = layout_state(state, B, T, N);
  WalkScratch sc = layout_scratch(scratch, B, T, N);
  if (state_bytes < st.bytes || scratch_bytes < sc.bytes) return CRW_EWORKSPACE;
  const int Np = padded_nodes(N), K = T - 2, nA = B * (T - 1);
  const long M = (long)Np * Np;

  const float coef = 1.0f / ((float)N * (float)B * (float)N);
  CRW_TRY(launch_dAt(st.At, st.lse, gloss, coef, K * B, N, Np, sc.dAt, nullptr, s));
  if (hipMemsetAsync(sc.dF, 0, sizeof(float) * nA * M, s) != hipSuccess) return CRW_EHIP;
  if (hipMemsetAsync(sc.dGt, 0, sizeof(float) * nA * M, s) != hipSuccess) return CRW_EHIP;

  for (int k = K; k >= 1; --k) {
    GemmGroup g{};
    g.n = Np; g.batch = B;
    const float *Lt_k = st.Lt + (long)(k - 1) * B * M, *R_k = st.R + (long)(k - 1) * B * M;
    const float *dAt_k = sc.dAt + (long)(k - 1) * B * M;
    float *dLt_cur = sc.dLt + (long)(k & 1) * B * M, *dLt_nxt = sc.dLt + (long)((k + 1) & 1) * B * M;
    float *dR_cur = sc.dR + (long)(k & 1) * B * M, *dR_nxt = sc.dR + (long)((k + 1) & 1) * B * M;
    GemmProb &p0 = g.p[g.nprob++];  // dLt_k = R_k dAt_k^T (+ Gt_k^T dLt_{k+1})
    p0.A = R_k; p0.sA = M; p0.B = dAt_k; p0.sB = M; p0.tb = 1;
    p0.C = dLt_cur; p0.sC = M;
    GemmProb &p1 = g.p[g.nprob++];  // dR_k = Lt_k dAt_k (+ F_k^T dR_{k+1})
    p1.A = Lt_k; p1.sA = M; p1.B = dAt_k; p1.sB = M;
    p1.C = dR_cur; p1.sC = M;
    if (k < K) {
      p0.A2 = st.Gt + (long)k * M; p0.sA2 = (long)(T - 1) * M; p0.ta2 = 1; p0.B2 = dLt_nxt; p0.sB2 = M;
      p1.A2 = st.F + (long)k * M; p1.sA2 = (long)(T - 1) * M; p1.ta2 = 1; p1.B2 = dR_nxt; p1.sB2 = M;
      GemmProb &p2 = g.p[g.nprob++];  // dGt_k = dLt_{k+1} Lt_k^T
      p2.A = dLt_nxt; p2.sA = M; p2.B = Lt_k; p2.sB = M; p2.tb = 1;
      p2.C = sc.dGt + (long)k * M; p2.sC = (long)(T - 1) * M;
      GemmProb &p3 = g.p[g.nprob++];  // dF_k = dR_{k+1} R_k^T
      p3.A = dR_nxt; p3.sA = M; p3.B = R_k; p3.sB = M; p3.tb = 1;
      p3.C = sc.dF + (long)k * M; p3.sC = (long)(T - 1) * M;
    }
    CRW_TRY(launch_gemm_group_f32(g, s));
  }
  // Lt_1 = Gt_0  ->  dGt_0 = dLt_1
  CRW_TRY(launch_copy_f32(sc.dGt, sc.dLt + (long)(1 & 1) * B * M, (long)(T - 1) * M, M, M, B, s));
  CRW_TRY(launch_softmax_bwd(st.F, st.Gt, sc.dF, sc.dGt, nA, N, Np, sc.stats, dA, s));
  return CRW_OK;
}

int crw_affinity_bwd(const float *dA, const float *ehat, const float *norm, int B, int T, int N, int C, float tau,
                     float *dehat_ws, float *demb, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!dA || !ehat || !norm || !dehat_ws || !demb || bad_shape(B, T, N) || C < 1 || !(tau > 0.f)) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long NC = (long)N * C, NN = (long)N * N;
  EdgeGemm g{};
  // product 1 (frames t < T-1): dA[b,t] (n, m) x ehat[b,t+1] (m, c)
  g.A = EdgeOperand{dA, NN, N, 1};
  g.B = EdgeOperand{ehat + NC, NC, C, 1};
  g.sA_outer = (long)(T - 1) * NN; g.sB_outer = (long)T * NC;
  g.K = N; g.skip1_inner = T - 1;
  // product 2 (frames t > 0): dA[b,t-1]^T (n, m) = dA[b,t-1][m][n] x ehat[b,t-1] (m, c)
  g.A2 = EdgeOperand{dA - NN, NN, 1, N};
  g.B2 = EdgeOperand{ehat - NC, NC, C, 1};
  g.sA2_outer = (long)(T - 1) * NN; g.sB2_outer = (long)T * NC;
  g.K2 = N; g.skip2_inner = 0;
  g.C = dehat_ws; g.sCb = NC; g.ldc = C; g.sC_outer = (long)T * NC;
  g.M = N; g.N = C;
  g.batch_inner = T;
  g.scale = tau; g.divide = 1;
  CRW_TRY(launch_edge_gemm(g, B * T, s));
  const long rows = (long)B * T * N;
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dehat_ws, ehat, norm,
                     rows, C, demb);
  return check_launch();
}

}  // extern "C"
