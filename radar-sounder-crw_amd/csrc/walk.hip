// Training path of the contrastive random walk: C-ABI entry points and the launch schedules.
//
//   crw_affinity_fwd : normalise rows (wave per row) + batched E_t E_{t+1}^T / tau  (src/model.py:22-26)
//   crw_walk_fwd     : dual softmax, then the chain in prefix form (which the
//                      oracle proves equal to src/model.py:35-46):
//                          Lt_1 = Gt_0          Lt_{k+1} = Gt_k Lt_k       (Lt_k = L_k^T)
//                          R_1  = I             R_{k+1}  = F_k  R_k
//                          At_k = Lt_k^T R_k    loss = (1/N) sum_k mean_{b,d}(lse(At_k[d,:]) - At_k[d,d])
//                      Schedule: the two recurrences are the only sequential part (T-3 grouped launches
//                      of 2 products, or ONE persistent LDS-resident kernel when Np <= 64); all
//                      At_k are then independent and go out as one batched launch (batch (T-2)*B).
//   crw_walk_bwd     : dAt for every k; one batched launch for the k-local terms (R_k dAt_k^T,
//                      Lt_k dAt_k); the reverse recurrences (dLt_k += Gt_k^T dLt_{k+1},
//                      dR_k += F_k^T dR_{k+1}) sequentially (or persistent kernel); one batched launch
//                      for dGt_k = dLt_{k+1} Lt_k^T, dF_k = dR_{k+1} R_k^T; dual-softmax backward.
//   crw_affinity_bwd : dE_t = (dA_t E_{t+1} + dA_{t-1}^T E_{t-1}) / tau, then normalise-backward.
#include "crw_common.h"

namespace crw {
namespace {

constexpr float EPS_NORM = 1e-12f;  // F.normalize default eps

// three equally long slices zeroed by one launch (the gradient slices no chain product writes)
__global__ __launch_bounds__(256) void zero3_kernel(float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, long n4) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
    reinterpret_cast<float4 *>(a)[e] = float4{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<float4 *>(b)[e] = float4{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<float4 *>(c)[e] = float4{0.f, 0.f, 0.f, 0.f};
  }
}

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

// fp32 plane + optional bf16 hi / lo images of one family of [Np][Np] matrices
struct Mat {
  float *f = nullptr;
  uint16_t *h = nullptr, *l = nullptr;
};
struct WalkState {
  Mat F, Gt, Lt, R, At;
  float *lse, *terms, *stats;
  size_t bytes;
};
struct WalkScratch {
  Mat dAt, dF, dGt, dLt, dR;
  float *stats;
  size_t bytes;
};

template <typename T>
T *carve(char *&p, size_t count) {
  T *r = reinterpret_cast<T *>(p);
  p += align_up(count * sizeof(T), 256);
  return r;
}

// which planes a family needs: fp32 chain -> fp32 only; bf16 chains -> per family
Mat carve_mat(char *&p, size_t count, int chain, bool want_f32_in_bf16_mode, bool want_img_in_bf16_mode = true) {
  Mat m;
  if (chain == CRW_CHAIN_F32 || want_f32_in_bf16_mode) m.f = carve<float>(p, count);
  if (chain != CRW_CHAIN_F32 && want_img_in_bf16_mode) {
    m.h = carve<uint16_t>(p, count);
    if (chain == CRW_CHAIN_BF16X3) m.l = carve<uint16_t>(p, count);
  }
  return m;
}

int chain_padded_nodes(int N, int chain) { return chain == CRW_CHAIN_F32 ? padded_nodes(N) : round_up(N, 128); }

WalkState layout_state(void *base, int B, int T, int N, int chain) {
  const size_t Np = chain_padded_nodes(N, chain), K = T - 2, nA = (size_t)B * (T - 1), M = Np * Np;
  char *p = static_cast<char *>(base);
  WalkState s;
  s.F = carve_mat(p, nA * M, chain, false);   // bf16 chains: images only (the softmax backward recomputes F / Gt from A)
  s.Gt = carve_mat(p, nA * M, chain, false);
  s.Lt = carve_mat(p, K * B * M, chain, false);  // bf16 chains: operands only
  s.R = carve_mat(p, K * B * M, chain, false);
  s.At = carve_mat(p, K * B * M, chain, true, false);  // loss needs fp32; never a GEMM operand
  s.lse = carve<float>(p, K * B * Np);
  s.terms = carve<float>(p, K * B * Np);
  s.stats = carve<float>(p, 4 * nA * Np);
  s.bytes = p - static_cast<char *>(base);
  return s;
}

WalkScratch layout_scratch(void *base, int B, int T, int N, int chain) {
  const size_t Np = chain_padded_nodes(N, chain), K = T - 2, nA = (size_t)B * (T - 1), M = Np * Np;
  char *p = static_cast<char *>(base);
  WalkScratch s;
  s.dAt = carve_mat(p, K * B * M, chain, false);  // bf16 chains: only ever a GEMM operand
  s.dF = carve_mat(p, nA * M, chain, true, false);
  s.dGt = carve_mat(p, nA * M, chain, true, false);
  s.dLt = carve_mat(p, K * B * M, chain, true);  // fp32 needed: accumulated in place (beta = 1)
  s.dR = carve_mat(p, K * B * M, chain, true);
  s.stats = carve<float>(p, 2 * nA * Np);
  s.bytes = p - static_cast<char *>(base);
  return s;
}

// (matrix counts ride in grid.y / grid.z of the elementwise and edge launches: B * T <= 65535)
bool bad_shape(int B, int T, int N) { return B < 1 || T < 2 || N < 1 || N > 16384 || (long)B * T > 65535; }
bool bad_chain(int chain) { return chain != CRW_CHAIN_F32 && chain != CRW_CHAIN_BF16 && chain != CRW_CHAIN_BF16X3; }

// operand / result descriptors for either arithmetic
struct Opnd {
  const Mat *m;
  long off;     // element offset of batch item 0
  long stride;  // batch stride
};
void set_a(GemmProb &p, int chain, Opnd o, int t) {
  if (chain == CRW_CHAIN_F32) p.A = o.m->f + o.off;
  else { p.A = o.m->h + o.off; p.Al = o.m->l ? o.m->l + o.off : nullptr; }
  p.sA = o.stride; p.ta = t;
}
void set_b(GemmProb &p, int chain, Opnd o, int t) {
  if (chain == CRW_CHAIN_F32) p.B = o.m->f + o.off;
  else { p.B = o.m->h + o.off; p.Bl = o.m->l ? o.m->l + o.off : nullptr; }
  p.sB = o.stride; p.tb = t;
}
void set_a2(GemmProb &p, int chain, Opnd o, int t) {
  if (chain == CRW_CHAIN_F32) p.A2 = o.m->f + o.off;
  else { p.A2 = o.m->h + o.off; p.A2l = o.m->l ? o.m->l + o.off : nullptr; }
  p.sA2 = o.stride; p.ta2 = t;
}
void set_b2(GemmProb &p, int chain, Opnd o, int t) {
  if (chain == CRW_CHAIN_F32) p.B2 = o.m->f + o.off;
  else { p.B2 = o.m->h + o.off; p.B2l = o.m->l ? o.m->l + o.off : nullptr; }
  p.sB2 = o.stride; p.tb2 = t;
}
void set_c(GemmProb &p, Opnd o) {
  p.C = o.m->f ? o.m->f + o.off : nullptr;
  p.Cb = o.m->h ? o.m->h + o.off : nullptr;
  p.Cbl = o.m->l ? o.m->l + o.off : nullptr;
  p.sC = o.stride;
}
int launch_group_once(const GemmGroup &g, int chain, hipStream_t s) {
  if (chain == CRW_CHAIN_F32) return launch_gemm_group_f32(g, s);
  return launch_gemm_group_bf16(g, chain == CRW_CHAIN_BF16X3 ? 3 : 1, s);
}
// grid.y carries the batch: split very large batches
int launch_group(const GemmGroup &g, int chain, hipStream_t s) {
  constexpr int MAXB = 32768;
  if (g.batch <= MAXB) return launch_group_once(g, chain, s);
  const size_t es = chain == CRW_CHAIN_F32 ? 4 : 2;  // operand element size
  for (int b0 = 0; b0 < g.batch; b0 += MAXB) {
    GemmGroup sub = g;
    sub.batch = g.batch - b0 < MAXB ? g.batch - b0 : MAXB;
    for (int i = 0; i < g.nprob; ++i) {
      GemmProb &q = sub.p[i];
      auto adv = [&](const void *&ptr, long stride) { if (ptr) ptr = (const char *)ptr + (size_t)b0 * stride * es; };
      adv(q.A, q.sA); adv(q.Al, q.sA); adv(q.B, q.sB); adv(q.Bl, q.sB);
      adv(q.A2, q.sA2); adv(q.A2l, q.sA2); adv(q.B2, q.sB2); adv(q.B2l, q.sB2);
      if (q.C) q.C += (long)b0 * q.sC;
      if (q.Cb) q.Cb = (uint16_t *)q.Cb + (long)b0 * q.sC;
      if (q.Cbl) q.Cbl = (uint16_t *)q.Cbl + (long)b0 * q.sC;
    }
    CRW_TRY(launch_group_once(sub, chain, s));
  }
  return CRW_OK;
}

}  // namespace
}  // namespace crw

using namespace crw;

extern "C" {

int crw_abi_version(void) { return CRW_ABI_VERSION; }
const char *crw_build_arch(void) { return "gfx950"; }
int crw_last_hip_error(void) { return g_last_hip_error; }
int crw_padded_nodes(int N, int chain) { return (N < 1 || bad_chain(chain)) ? 0 : chain_padded_nodes(N, chain); }

size_t crw_walk_state_bytes(int B, int T, int N, int chain) {
  if (bad_shape(B, T, N) || bad_chain(chain) || T < 3) return 256;
  return layout_state(nullptr, B, T, N, chain).bytes;
}
size_t crw_walk_scratch_bytes(int B, int T, int N, int chain) {
  if (bad_shape(B, T, N) || bad_chain(chain) || T < 3) return 256;
  return layout_scratch(nullptr, B, T, N, chain).bytes;
}

int crw_normalize(const float *emb, int rows, int C, float *ehat, float *norm, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!emb || !ehat || rows < 1 || C < 1) return CRW_EINVAL;
  hipLaunchKernelGGL(normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, (long)rows, C,
                     ehat, norm);
  return check_launch();
}

size_t crw_affinity_ws_bytes(int B, int T, int N) {
  if (bad_shape(B, T, N)) return 0;
  return affinity_part_floats(B, T, N) * sizeof(float);
}

int crw_affinity_fwd(const float *emb, int B, int T, int N, int C, float tau, float *ehat, float *norm, float *A,
                     float *stats, void *ws, size_t ws_bytes, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!emb || !ehat || !norm || !A || bad_shape(B, T, N) || C < 1 || !(tau > 0.f)) return CRW_EINVAL;
  if (stats && (!ws || ws_bytes < crw_affinity_ws_bytes(B, T, N))) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  CRW_TRY(crw_normalize(emb, B * T * N, C, ehat, norm, stream));
  if (C % 4 == 0)  // 128 x 128 fp32-MFMA tiles, statistics in the epilogue
    return launch_affinity_tiles(ehat, B, T, N, C, tau, A, stats, static_cast<float *>(ws), s);
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
  CRW_TRY(launch_edge_gemm(g, B * (T - 1), s));
  if (stats) CRW_TRY(launch_stats_dense(A, B * (T - 1), N, stats, s));  // odd channel counts: one extra pass over A
  return CRW_OK;
}

int crw_walk_fwd(const float *A, const float *stats_ext, int B, int T, int N, int chain, void *state, size_t state_bytes,
                 float *At_out, float *loss, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!A || !loss || bad_shape(B, T, N) || bad_chain(chain)) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (T < 3) {  // no cycle: the reference returns 0/N
    if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) return CRW_EHIP;
    return CRW_OK;
  }
  if (!state) return CRW_EINVAL;
  WalkState st = layout_state(state, B, T, N, chain);
  if (state_bytes < st.bytes) return CRW_EWORKSPACE;
  const int Np = chain_padded_nodes(N, chain), K = T - 2, nA = B * (T - 1);
  const long M = (long)Np * Np, BM = (long)B * M;  // every family is [k][b][Np][Np]: k stride BM, batch stride M
  const bool small = chain == CRW_CHAIN_F32 && Np <= 64;

  (void)nA;
  CRW_TRY(launch_softmax_fwd(A, stats_ext, B, T - 1, N, Np, st.F.f, st.Gt.f, st.F.h, st.F.l, st.Gt.h, st.Gt.l, st.stats, s));
  if (chain == CRW_CHAIN_F32) {
    CRW_TRY(launch_identity(st.R.f, nullptr, B, Np, N, s, st.Lt.f, st.Gt.f));  // R_1 = I and Lt_1 = Gt_0
  } else {  // the same on the bf16 images (two bf16 per float lane of the copy kernel)
    CRW_TRY(launch_copy_f32((float *)st.Lt.h, (const float *)st.Gt.h, 0, 0, BM / 2, 1, s));
    if (st.Lt.l) CRW_TRY(launch_copy_f32((float *)st.Lt.l, (const float *)st.Gt.l, 0, 0, BM / 2, 1, s));
    CRW_TRY(launch_identity(nullptr, st.R.h, B, Np, N, s));
    if (st.R.l && hipMemsetAsync(st.R.l, 0, sizeof(uint16_t) * BM, s) != hipSuccess) return CRW_EHIP;
  }

  // the sequential part: Lt_{k+1} = Gt_k Lt_k, R_{k+1} = F_k R_k   (array index i holds X_{i+1})
  if (small) {
    CRW_TRY(launch_chain_small_fwd(st.Gt.f, st.F.f, st.Lt.f, st.R.f, B, K, Np, s));
  } else {
    for (int i = 1; i < K; ++i) {
      GemmGroup g{};
      g.n = Np; g.batch = B;
      GemmProb &p1 = g.p[g.nprob++];
      set_a(p1, chain, {&st.Gt, i * BM, M}, 0);
      set_b(p1, chain, {&st.Lt, (i - 1) * BM, M}, 0);
      set_c(p1, {&st.Lt, i * BM, M});
      GemmProb &p2 = g.p[g.nprob++];
      set_a(p2, chain, {&st.F, i * BM, M}, 0);
      set_b(p2, chain, {&st.R, (i - 1) * BM, M}, 0);
      set_c(p2, {&st.R, i * BM, M});
      CRW_TRY(launch_group(g, chain, s));
    }
  }
  {  // every cycle product at once: At_k = Lt_k^T R_k, batch K*B
    GemmGroup g{};
    g.n = Np; g.batch = K * B;
    GemmProb &p0 = g.p[g.nprob++];
    set_a(p0, chain, {&st.Lt, 0, M}, 1);
    set_b(p0, chain, {&st.R, 0, M}, 0);
    set_c(p0, {&st.At, 0, M});
    CRW_TRY(launch_group(g, chain, s));
  }
  CRW_TRY(launch_loss_rows(st.At.f, K * B, N, Np, st.lse, st.terms, s));
  CRW_TRY(launch_loss_reduce(st.terms, (long)K * B * Np, 1.0f / ((float)B * (float)N * (float)N), loss, s));
  if (At_out) CRW_TRY(launch_unpad_At(st.At.f, K, B, N, Np, At_out, s));
  return CRW_OK;
}

int crw_walk_bwd(const float *gloss, const float *A, int B, int T, int N, int chain, void *state, size_t state_bytes,
                 void *scratch, size_t scratch_bytes, float *dA, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!gloss || !A || !dA || bad_shape(B, T, N) || bad_chain(chain)) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long NN = (long)N * N;
  if (T < 3) {
    if (hipMemsetAsync(dA, 0, sizeof(float) * B * (T - 1) * NN, s) != hipSuccess) return CRW_EHIP;
    return CRW_OK;
  }
  if (!state || !scratch) return CRW_EINVAL;
  WalkState st = layout_state(state, B, T, N, chain);
  WalkScratch sc = layout_scratch(scratch, B, T, N, chain);
  if (state_bytes < st.bytes || scratch_bytes < sc.bytes) return CRW_EWORKSPACE;
  const int Np = chain_padded_nodes(N, chain), K = T - 2;
  const long M = (long)Np * Np, BM = (long)B * M;
  const bool small = chain == CRW_CHAIN_F32 && Np <= 64;

  const float coef = 1.0f / ((float)N * (float)B * (float)N);
  CRW_TRY(launch_dAt(st.At.f, st.lse, gloss, coef, K * B, N, Np, sc.dAt.f, sc.dAt.h, sc.dAt.l, s));
  // slices no product writes: dF_0, dF_{T-2}, dGt_{T-2}
  if ((reinterpret_cast<uintptr_t>(sc.dF.f) | reinterpret_cast<uintptr_t>(sc.dGt.f)) & 15) {  // caller's scratch not 16-byte aligned
    if (hipMemsetAsync(sc.dF.f, 0, sizeof(float) * BM, s) != hipSuccess) return CRW_EHIP;
    if (hipMemsetAsync(sc.dF.f + (long)(T - 2) * BM, 0, sizeof(float) * BM, s) != hipSuccess) return CRW_EHIP;
    if (hipMemsetAsync(sc.dGt.f + (long)(T - 2) * BM, 0, sizeof(float) * BM, s) != hipSuccess) return CRW_EHIP;
  } else {  // BM = B * Np^2 with Np a multiple of 32: float4 stores
    const long n4 = BM / 4;
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(zero3_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, sc.dF.f,
                       sc.dF.f + (long)(T - 2) * BM, sc.dGt.f + (long)(T - 2) * BM, n4);
    CRW_TRY(check_launch());
  }

  {  // k-local terms for every k at once: dLt_k = R_k dAt_k^T, dR_k = Lt_k dAt_k
    GemmGroup g{};
    g.n = Np; g.batch = K * B;
    GemmProb &p0 = g.p[g.nprob++];
    set_a(p0, chain, {&st.R, 0, M}, 0);
    set_b(p0, chain, {&sc.dAt, 0, M}, 1);
    set_c(p0, {&sc.dLt, 0, M});
    GemmProb &p1 = g.p[g.nprob++];
    set_a(p1, chain, {&st.Lt, 0, M}, 0);
    set_b(p1, chain, {&sc.dAt, 0, M}, 0);
    set_c(p1, {&sc.dR, 0, M});
    CRW_TRY(launch_group(g, chain, s));
  }
  // the sequential part: dLt_k += Gt_k^T dLt_{k+1}, dR_k += F_k^T dR_{k+1}   (index i holds X_{i+1})
  if (small) {
    CRW_TRY(launch_chain_small_bwd(st.Gt.f, st.F.f, sc.dLt.f, sc.dR.f, B, K, Np, s));
  } else {
    for (int i = K - 2; i >= 0; --i) {
      GemmGroup g{};
      g.n = Np; g.batch = B;
      GemmProb &p0 = g.p[g.nprob++];
      set_a(p0, chain, {&st.Gt, (i + 1) * BM, M}, 1);
      set_b(p0, chain, {&sc.dLt, (i + 1) * BM, M}, 0);
      set_c(p0, {&sc.dLt, i * BM, M});
      p0.beta = 1;
      GemmProb &p1 = g.p[g.nprob++];
      set_a(p1, chain, {&st.F, (i + 1) * BM, M}, 1);
      set_b(p1, chain, {&sc.dR, (i + 1) * BM, M}, 0);
      set_c(p1, {&sc.dR, i * BM, M});
      p1.beta = 1;
      CRW_TRY(launch_group(g, chain, s));
    }
  }
  if (K > 1) {  // dGt_k = dLt_{k+1} Lt_k^T, dF_k = dR_{k+1} R_k^T for k = 1..K-1, batch (K-1)*B
    GemmGroup g{};
    g.n = Np; g.batch = (K - 1) * B;
    GemmProb &p2 = g.p[g.nprob++];
    set_a(p2, chain, {&sc.dLt, BM, M}, 0);
    set_b(p2, chain, {&st.Lt, 0, M}, 1);
    set_c(p2, {&sc.dGt, BM, M});
    GemmProb &p3 = g.p[g.nprob++];
    set_a(p3, chain, {&sc.dR, BM, M}, 0);
    set_b(p3, chain, {&st.R, 0, M}, 1);
    set_c(p3, {&sc.dF, BM, M});
    CRW_TRY(launch_group(g, chain, s));
  }
  // Lt_1 = Gt_0  ->  dGt_0 = dLt_1
  CRW_TRY(launch_copy_f32(sc.dGt.f, sc.dLt.f, 0, 0, BM, 1, s));
  CRW_TRY(launch_softmax_bwd(A, st.stats, sc.dF.f, sc.dGt.f, B, T - 1, N, Np, sc.stats, dA, s));
  return CRW_OK;
}

int crw_affinity_bwd(const float *dA, const float *ehat, const float *norm, int B, int T, int N, int C, float tau,
                     float *dehat_ws, float *demb, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!dA || !ehat || !norm || !dehat_ws || !demb || bad_shape(B, T, N) || C < 1 || !(tau > 0.f)) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long NC = (long)N * C, NN = (long)N * N;
  if (C == 32 || C == 64 || C == 128) {  // fp32-MFMA / bf16-split tiles (any node count)
    CRW_TRY(launch_affinity_bwd_tiles(dA, ehat, B, T, N, C, tau, dehat_ws, s));
    const long rows_ = (long)B * T * N;
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3((unsigned)((rows_ + 3) / 4)), dim3(256), 0, s, dehat_ws, ehat, norm, rows_, C, demb);
    return check_launch();
  }
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
