/*
 * crw_hip.h -- C ABI of libcrw_hip.so: the MI355X (gfx950) implementation of the
 * contrastive-random-walk hot path of jdalcorso/radar-sounder-crw.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; its hot path is the sequence
 * of ATen calls listed below.  Each entry point replaces one such group of calls
 * (file:line under the reference repo):
 *
 *   crw_affinity_fwd ........ F.normalize + einsum('bctn,bctm->btnm')/tau      src/model.py:22-26
 *   crw_walk_fwd ............ palindrome cat + (T-2)^2 x {softmax,bmm} +
 *                             (T-2) x cross_entropy + loss/N                  src/model.py:31-46
 *   crw_walk_bwd ............ autograd backward of the above                  (loss.backward(), scripts/train.py:71)
 *   crw_affinity_bwd ........ autograd backward of normalize+einsum           (same)
 *   crw_labelprop_topk ...... einsum + mask + /temp + truncation + topk +
 *                             softmax                                         src/imported/maskedatt.py:151-175
 *   crw_labelprop_gather .... weighted label sum + argmax, frame by frame     src/imported/labelprop.py:106-114, src/utils.py:152-160
 *   crw_labelprop_propagate . the same, chained prefix + parallel tail          (same lines; context bound of maskedatt.py:165-166)
 *   crw_xent_metric ......... "horizontality" metric                          src/utils.py:117-125
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *   - tensors are dense row-major fp32 with the shapes given per function;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); nothing synchronises;
 *   - no allocation inside: workspaces are sized with the crw_*_bytes() queries and passed in;
 *   - return value: CRW_OK or an error code; nothing is thrown across the ABI;
 *   - Threads: the kernels' entry points keep no state between calls and may be called from any thread (one caller per stream).
 *     Two corners do keep state: crw_rn_train_fwd / crw_rn_train_bwd / crw_rn_eval_fwd order part of their work on one internal
 *     side stream PER DEVICE (created on first use, with its event pool) -- at most one host thread per device may be inside them
 *     at a time; and the crw_rn_timing_* diagnostic (one global record list).  Everything they enqueue on the side stream is joined
 *     back into the caller's stream before they return, on error paths too.
 */
#ifndef CRW_HIP_H
#define CRW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumps when a signature changes or an entry point is added.  The ONE place the number is written: crw_abi_version() returns
 * it, the ctypes binding (crw_hip.ABI_VERSION) parses it from this header, and __graft_entry__.build() / the host tests compare
 * the two. */
#define CRW_ABI_VERSION 7

#define CRW_OK 0
#define CRW_EINVAL 1     /* bad shape / null pointer / unsupported size            */
#define CRW_EWORKSPACE 2 /* workspace too small                                     */
#define CRW_EHIP 3       /* a HIP launch failed (see crw_last_hip_error)            */

/* arithmetic used inside the transition-matrix chain */
#define CRW_CHAIN_F32 0  /* exact fp32 MFMA (v_mfma_f32_16x16x4_f32), parity path   */
#define CRW_CHAIN_BF16 1 /* bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16) */
#define CRW_CHAIN_BF16X3 2 /* hi/lo bf16 operand pairs, 3 MFMAs per product: fp32-grade results */

typedef void *crw_stream_t;

/* library / build info ------------------------------------------------------------------- */
int crw_abi_version(void);          /* CRW_ABI_VERSION of the header the library was built with */
const char *crw_build_arch(void);   /* "gfx950"                                             */
int crw_last_hip_error(void);       /* last hipError_t seen by this thread (0 = none)       */

/* geometry ------------------------------------------------------------------------------- */
/* node count padded to the tile size of the chain GEMM (all internal NxN matrices are stored
 * [Np][Np] with zero padding). */
int crw_padded_nodes(int N, int chain);
/* bytes of the state buffer written by crw_walk_fwd and consumed by crw_walk_bwd */
size_t crw_walk_state_bytes(int B, int T, int N, int chain);
/* bytes of the scratch buffer needed by crw_walk_bwd */
size_t crw_walk_scratch_bytes(int B, int T, int N, int chain);

/* training path --------------------------------------------------------------------------- */
/* emb [B,T,N,C] raw encoder output -> ehat [B,T,N,C] (L2-normalised, eps 1e-12),
 * norm [B,T,N] (= max(||e||, eps)), A [B,T-1,N,N] = ehat_t ehat_{t+1}^T / tau.
 * stats (may be NULL): [4][B][T-1][N] = row max, row sum exp, column max, column sum exp of every A[b,t] -- the
 * statistics of the two softmaxes of src/model.py:44 (F = softmax(A), G = softmax(A^T)), produced in the epilogue of
 * the affinity tiles so that crw_walk_fwd needs no pass over A to find them; needs ws of crw_affinity_ws_bytes. */
size_t crw_affinity_ws_bytes(int B, int T, int N);
int crw_affinity_fwd(const float *emb, int B, int T, int N, int C, float tau,
                     float *ehat, float *norm, float *A, float *stats, void *ws, size_t ws_bytes, crw_stream_t stream);

/* A [B,T-1,N,N] -> loss[1] (= sum_k l_k / N, Appendix A.2 of SURVEY.md).
 * state: crw_walk_state_bytes(B,T,N) bytes, kept by the caller until crw_walk_bwd.
 * At_out: optional [B,T-2,N,N] copy of every per-cycle transition product (may be NULL).
 * chain: CRW_CHAIN_F32 | CRW_CHAIN_BF16 | CRW_CHAIN_BF16X3 (same value for fwd, bwd and the size
 * queries).   T < 3 -> loss = 0. */
int crw_walk_fwd(const float *A, const float *stats /* of crw_affinity_fwd, or NULL: computed here */, int B, int T, int N,
                 int chain, void *state, size_t state_bytes, float *At_out, float *loss, crw_stream_t stream);

/* gloss[1] (device scalar, dL/dloss) + the logits A the forward ran on + state -> dA [B,T-1,N,N]
 * (the two softmaxes are recomputed from A and the statistics kept in `state`: half the bytes of storing them). */
int crw_walk_bwd(const float *gloss, const float *A, int B, int T, int N, int chain,
                 void *state, size_t state_bytes, void *scratch, size_t scratch_bytes,
                 float *dA, crw_stream_t stream);

/* dA [B,T-1,N,N], ehat, norm -> demb [B,T,N,C]; dehat_ws is a [B,T,N,C] scratch. */
int crw_affinity_bwd(const float *dA, const float *ehat, const float *norm, int B, int T, int N, int C,
                     float tau, float *dehat_ws, float *demb, crw_stream_t stream);

/* inference path -------------------------------------------------------------------------- */
/* emb [T,N,C] raw -> ehat [T,N,C] (reuses the normalise kernel of crw_affinity_fwd). */
int crw_normalize(const float *emb, int rows, int C, float *ehat, float *norm, crw_stream_t stream);

/* For every frame n = first_frame..T-1 (first_frame >= 1; 1 = whole radargram, T-1 = the
 * reference's one-frame-at-a-time LabelPropVOS_CRW.predict) and query node q: top-`knn` keys among the context frames
 * (frame 0 + last `cxt_size` frames once n > cxt_size+1, else frames 0..n-1) restricted to the
 * band |m-q| < radius, logits <key,query>/temp, softmax over the knn.
 * W [T-first_frame,knn,N] weights, I [same] int32 indices into the (truncated) key list.
 * Requires 1 <= radius, 1 <= knn <= 64. Slots beyond the number of in-band keys get weight 0
 * (the reference fills them with masked keys whose softmax weight is exactly 0).
 * Scores are exact fp32 products accumulated in fp32: on the fp32 matrix cores when N >= 16, C is 64, 128 or 256 and ehat is 16-byte
 * aligned, else on the vector units -- the two differ in summation order only. */
int crw_labelprop_topk(const float *ehat, int T, int N, int C, int cxt_size, int radius, float temp,
                       int knn, int first_frame, float *W, int32_t *I, crw_stream_t stream);
/* The same on a 2-D node grid: the N nodes of a frame are an (N / grid_w) x grid_w grid in row-major order and the band is the
 * Euclidean disc (i - i')^2 + (j - j')^2 < radius^2 of MaskedAttention (src/imported/maskedatt.py:222-245); grid_w = 1 is
 * crw_labelprop_topk (a radargram's frames are N x 1 columns of patches, the only grid the reference's scripts produce). */
int crw_labelprop_topk_grid(const float *ehat, int T, int N, int C, int cxt_size, int radius, float temp, int knn, int first_frame,
                            int grid_w, float *W, int32_t *I, crw_stream_t stream);

/* seed [N] float class ids of frame 0 (NULL: rows of L for frames < first_frame are already
 * filled by the caller); W,I from crw_labelprop_topk with the same first_frame;
 * L [T*N, M] soft labels (frames >= first_frame written), pred [N,T] float class ids
 * (columns >= first_frame written; column 0 = seed when seed is given). */
int crw_labelprop_gather(const float *seed, const float *W, const int32_t *I, int T, int N, int M, int knn,
                         int first_frame, float *L, float *pred, crw_stream_t stream);
/* The same result for lists that come from crw_labelprop_topk(_grid) with context size `cxt_size` (same first_frame): their
 * indices address the truncated key list, i.e. frame n's are < min(n, cxt_size + 1) * N, and the reference applies them to the
 * untruncated label list (src/imported/labelprop.py:103-107 with src/imported/maskedatt.py:165-166) -- so only frames
 * first_frame .. cxt_size form a chain (one workgroup, all their labels in LDS) and every later frame reads frames 0 .. cxt_size
 * alone (one workgroup per frame, all at once).  Bit-identical L and pred; indices outside the bound are clamped to rows before
 * the frame's own, never out of range.  Falls back to crw_labelprop_gather where the chained frames' labels exceed the LDS. */
int crw_labelprop_propagate(const float *seed, const float *W, const int32_t *I, int T, int N, int M, int knn, int first_frame,
                            int cxt_size, float *L, float *pred, crw_stream_t stream);

/* HOST function (no GPU work, host pointers): the change-point search of `propagate` (src/utils.py:125-132,
 * ruptures.Pelt(model="rbf").fit(signal).predict(pen)) -- PELT with the RBF kernel cost at ruptures' documented defaults
 * min_size 2, jump 5; gamma <= 0: the median heuristic.  signal [n] doubles -> sorted breakpoints bkps[0..count) (segment ends, the
 * last one is n); returns count >= 1, or a negative status.  The arithmetic of pelt.py in the same order (parity with ruptures
 * itself is unpinned: it is not installed here). */
int crw_pelt_rbf(const double *signal, int n, double pen, int min_size, int jump, double gamma, int *bkps, int max_bkps);

/* ehat [T,N,C] -> xent [N,T-1]  (channel-shifted within-frame affinity / 0.1, CE vs identity) */
int crw_xent_metric(const float *ehat, int T, int N, int C, float *xent, crw_stream_t stream);

/* building blocks exported for tests and the roofline bench --------------------------------- */
/* Weight gradient of the CNN encoder's linear head (nn.Linear(128, 128), src/encoder.py:40,55; autograd of
 * src/model.py:20): dw[o][i] = sum_p dy[p][o] x[p][i] as P/128 batched 128x128 fp32-MFMA products whose partial matrices
 * are added in a fixed order (a library GEMM runs this M=N=128, K=P shape on 16 workgroups).  P % 128 == 0. */
size_t crw_linear128_wgrad_ws_bytes(int P);
int crw_linear128_wgrad(const float *dy, const float *x, float *dw, int P, void *ws, size_t ws_bytes, crw_stream_t stream);

/* One Adam step (torch.optim.Adam defaults: no weight decay, no amsgrad -- the reference's optimizer, scripts/train.py:54,69) on a
 * flat fp32 parameter buffer p[n] with gradient g[n] and moment buffers m[n], v[n] (all 16-byte aligned); step = 1, 2, ...
 * Operation for operation the arithmetic of torch's default implementation; the host side keeps the module's parameters as views
 * of p (radar-sounder-crw_amd/optim.py) and the gradients as views of g (dist.FlatGradBucket). */
int crw_adam_step(float *p, const float *g, float *m, float *v, long n, float lr, float beta1, float beta2, float eps, int step,
                  crw_stream_t stream);

/* X [batch,n,n] (n multiple of 32, zero padded): C = op(A) * op(B) (+ C if beta), fp32 MFMA. */
int crw_gemm_f32(const float *A, const float *B, float *C, int n, int batch, int transA, int transB,
                 int beta, crw_stream_t stream);

/* encoder: hand-written 3x3 convolutions of CNN (conv3/conv4/conv5, src/encoder.py:26-35) --------- */
/* Activations are channels-last bf16 planes [P][100][C] ("hi" plane and, for split = 3, a "lo"
 * plane: x = hi + lo to ~1e-5 relative).  10x10 feature maps (16x16 input patches).
 * fp32 conv weight [cout][cin][3][3] -> forward planes and backward-data planes (taps flipped, cin <-> cout),
 * 9*cout*cin bf16 each, stored in MFMA fragment order (opaque to the caller); lo planes may be NULL
 * together (plain bf16). */
int crw_enc_pack_weights(const float *w, int cout, int cin, uint16_t *fwd_hi, uint16_t *fwd_lo,
                         uint16_t *bwd_hi, uint16_t *bwd_lo, crw_stream_t stream);
/* fp32 NCHW [P][C][10][10] (output of pool2) -> planes */
int crw_enc_pack_input(const float *x, int P, int C, uint16_t *x_hi, uint16_t *x_lo, crw_stream_t stream);
/* same for feature maps of any size: fp32 NCHW [P][C][H][W] -> planes [P][H*W][C] */
int crw_enc_pack_input_map(const float *x, int P, int C, int H, int W, uint16_t *x_hi, uint16_t *x_lo,
                           crw_stream_t stream);
/* mode 0: y = relu(conv3x3(x, w) + bias), optional gap[P][cout] = mean over pixels (AdaptiveAvgPool2d(1));
 * mode 1: backward-data, y = conv3x3(x = dY, w = backward planes), zeroed where mask_hi (the forward
 *         activation of the layer below, [P][100][cout]) is 0; bias ignored.
 * (cin, cout) in {(32,64),(64,128),(128,128),(128,64),(64,32)}.  Outputs: planes y_hi/y_lo and/or
 * y_f32 [P][100][cout] (fp32); y_lo may be NULL when only the hi plane of the result is needed.
 * dgap (mode 1 only, may be NULL): fused ReLU + global-average-pool backward -- the input gradient
 * is dgap[P][cin] / 100 where x_hi (then the FORWARD activation plane of that layer) is non-zero;
 * x_lo is ignored. */
int crw_enc_conv3x3(int mode, int split, int P, int cin, int cout, const uint16_t *x_hi, const uint16_t *x_lo,
                    const uint16_t *w_hi, const uint16_t *w_lo, const float *bias, const uint16_t *mask_hi,
                    uint16_t *y_hi, uint16_t *y_lo, float *y_f32, float *gap, const float *dgap,
                    crw_stream_t stream);
/* The 3x3 layers on feature maps of ANY size (patch sizes other than 16x16; src/encoder.py:49-53 is size-agnostic):
 * planes are [P][H][W][C]; a workgroup computes one 10x10 output tile from the 12x12 window around it (zeros outside
 * the map).  mode 0: relu(conv + bias); y_hi/y_lo may be NULL when only the pooled result is wanted; gap_part
 * [P * ceil(H/10) * ceil(W/10)][cout] receives per-tile SUMS over the in-map pixels (the caller adds the tiles and
 * divides by H*W = AdaptiveAvgPool2d(1)).  mode 1: backward-data (w = the backward planes of crw_enc_pack_weights,
 * cin/cout swapped), output zeroed where mask_hi (activation map of the layer below, may be NULL) is 0; y_f32
 * (may be NULL): fp32 copy of the output [P][H][W][cout]. */
int crw_enc_conv3x3_map(int mode, int split, int P, int H, int W, int cin, int cout, const uint16_t *x_hi, const uint16_t *x_lo,
                        const uint16_t *w_hi, const uint16_t *w_lo, const float *bias, const uint16_t *mask_hi,
                        uint16_t *y_hi, uint16_t *y_lo, float *y_f32, float *gap_part, crw_stream_t stream);
/* weight / bias gradient of one 3x3 layer on feature maps of any size (autograd of src/encoder.py:49-53 at other patch
 * sizes): dY [P][H][W][cout], X [P][H][W][cin] planes -> dw [cout][cin][3][3], db [cout].
 * ws: crw_enc_wgrad_ws_bytes(P * ceil(H/10) * ceil(W/10), cin, cout, split). */
int crw_enc_conv3x3_wgrad_map(int split, int P, int H, int W, int cin, int cout, const uint16_t *dy_hi, const uint16_t *dy_lo,
                              const uint16_t *x_hi, const uint16_t *x_lo, float *dw, float *db, void *ws, size_t ws_bytes,
                              crw_stream_t stream);
/* dY planes [P][npix][C] = dgap[P][C] / npix where y_hi != 0 (backward of ReLU + global average pool; npix = 100 for
 * 16x16 patches, H*W of the conv3-5 feature map otherwise) */
int crw_enc_gap_bwd(const float *dgap, const uint16_t *y_hi, int P, int C, int npix, uint16_t *dy_hi, uint16_t *dy_lo,
                    crw_stream_t stream);
/* dw [cout][cin][3][3], db [cout] (fp32, overwritten) from dY planes [P][100][cout] and x planes [P][100][cin].
 * Per-slice partial sums go to `ws` (crw_enc_wgrad_ws_bytes) and are added in a fixed order:
 * no float atomics, bitwise reproducible.  dgap (may be NULL): as in crw_enc_conv3x3, dY = dgap/100
 * gated by dy_hi (= forward activation plane); dy_lo ignored. */
size_t crw_enc_wgrad_ws_bytes(int P, int cin, int cout, int split);
int crw_enc_conv3x3_wgrad(int split, int P, int cin, int cout, const uint16_t *dy_hi, const uint16_t *dy_lo,
                          const uint16_t *x_hi, const uint16_t *x_lo, const float *dgap, float *dw, float *db,
                          void *ws, size_t ws_bytes, crw_stream_t stream);

/* encoder front end: conv1 5x5 -> ReLU -> maxpool 2x2/1 -> conv2 5x5 -> ReLU -> maxpool 2x2/1, fused
 * (src/encoder.py:13-24,46-47; 16x16 patches, cin = 1 or 2 with pos_embed) -------------------------- */
/* conv2 weight [32][8][5][5] fp32 -> forward planes [7][32][32] and backward planes [25][8][32] */
int crw_enc_front_pack(const float *w2, uint16_t *fwd_hi, uint16_t *fwd_lo, uint16_t *bwd_hi, uint16_t *bwd_lo,
                       crw_stream_t stream);
/* x [P][cin][16][16] fp32 -> planes [P][100][32] (input of crw_enc_conv3x3 cin = 32).
 * saved (may be NULL; training): crw_enc_front_saved_bytes(P) bytes that receive, per patch, the pool1 output planes and one
 * code byte per pooling window (arg-max position + ReLU gate); crw_enc_front_bwd then needs no recomputation. */
size_t crw_enc_front_saved_bytes(int P);
int crw_enc_front_fwd(int split, const float *x, int P, int cin, const float *w1, const float *b1,
                      const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, uint16_t *y_hi,
                      uint16_t *y_lo, void *saved, crw_stream_t stream);
/* The same front end on patches of any size h, w >= 7 (forward only, inference): x [P][cin][H][W] -> planes
 * [P][(H-6)*(W-6)][32] that feed crw_enc_conv3x3_map.  Work item = (patch, 10x10 tile of the output map). */
int crw_enc_front_fwd_map(int split, const float *x, int P, int cin, int H, int W, const float *w1, const float *b1,
                          const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, uint16_t *y_hi, uint16_t *y_lo,
                          crw_stream_t stream);
/* backward: dy [P][100][32] fp32 -> dw1 [8][cin][5][5], db1 [8], dw2 [32][8][5][5], db2 [32]; partial sums per patch
 * slice in `ws`, added in a fixed order.  saved: the record crw_enc_front_fwd wrote for these patches, or NULL (the kernel
 * then recomputes conv1 -> pool1 -> conv2 per patch; same results, ~40 % more time). */
size_t crw_enc_front_ws_bytes(int P, int cin);
int crw_enc_front_bwd(int split, const float *x, int P, int cin, const float *w1, const float *b1,
                      const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, const uint16_t *w2b_hi,
                      const uint16_t *w2b_lo, const float *dy, const void *saved, float *dw1, float *db1, float *dw2,
                      float *db2, void *ws, size_t ws_bytes, crw_stream_t stream);

/* The same backward on patches of any size h, w >= 7 (training at patch sizes other than 16x16): x [P][cin][H][W], dy
 * [P][H-6][W-6][32] fp32 = gradient of the planes of crw_enc_front_fwd_map.  Unit of work = (patch, 10x10 tile of that map);
 * ws: crw_enc_front_ws_bytes(P * ceil((H-6)/10) * ceil((W-6)/10), cin). */
int crw_enc_front_bwd_map(int split, const float *x, int P, int cin, int H, int W, const float *w1, const float *b1,
                          const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, const uint16_t *w2b_hi,
                          const uint16_t *w2b_lo, const float *dy, float *dw1, float *db1, float *dw2, float *db2,
                          void *ws, size_t ws_bytes, crw_stream_t stream);

/* Resnet encoder (the reference's default model: src/encoder.py:63-89 stem, :109-155 BasicBlock, :157-272 body), forward AND
 * backward on hand-written kernels ----------------------------------------------------------------------------------------
 * Conventions of this group.  Ppad = crw_rn_padded_patches(P) (P rounded up to 128).  "planes" are channels-last bf16 pairs
 * (hi, lo; x = hi + lo) of shape [Ppad][pixels][C] whose rows P..Ppad-1 are ZERO; raw convolution outputs are fp32
 * [Ppad][pixels][C].  Every convolution runs as a matrix product ACROSS PATCHES (tile = 128 patches x 64/128 channels, one
 * group per output pixel, reduction over the kernel taps that fall inside the map) on v_mfma_f32_16x16x32_bf16 with hi/lo
 * operand pairs (3 products, fp32 accumulate).  Train-mode BatchNorm (nn.BatchNorm2d defaults) is split into: per-tile
 * column statistics from the convolution's epilogue -> crw_rn_bn_stats -> crw_rn_bn_apply / crw_rn_bn_pool; backward
 * crw_rn_bn_bwd / crw_rn_pool_bwd.  A BatchNorm's coefficients travel as coef[4][C] = scale, shift, mean, invstd. */
int crw_rn_padded_patches(int P);
/* conv weight [cout][cin][kh][kw] fp32 -> forward planes [cout][kh*kw][cin] and backward-data planes [cin][kh*kw][cout] */
int crw_rn_pack_conv(const float *w, int cout, int cin, int kh, int kw, uint16_t *fwd_hi, uint16_t *fwd_lo, uint16_t *bwd_hi,
                     uint16_t *bwd_lo, crw_stream_t stream);
/* stem convolution model.conv1 [64][3][7][7] (src/encoder.py:185) for h x w patches: forward planes [64][256] and the Toeplitz
 * planes [(h+2)][crw_rn_stem_cols(w)][crw_rn_stem_toeplitz_ld(w)] of its backward-data product.  crw_rn_stem_cols(w) = 3 * (w + 2)
 * rounded up to 64: the columns (ix * 3 + c) of one row of the stem's input gradient -- any patch width. */
int crw_rn_stem_toeplitz_ld(int w);
int crw_rn_stem_cols(int w);
int crw_rn_pack_stem(const float *w1, int h, int w, uint16_t *fwd_hi, uint16_t *fwd_lo, uint16_t *toep_hi, uint16_t *toep_lo,
                     crw_stream_t stream);
/* mode 0: forward of conv2d(kh x kw, stride, pad, no bias) -- a = input planes [Ppad][Hs*Ws][Cs], b = forward weight planes,
 *         out fp32 [Ppad][Hd*Wd][N] (N = cout);  a linear layer is the 1x1 case on a 1x1 map (bias: optional [N]).
 * mode 1: backward-data -- a = dZ planes on the OUTPUT map [Ppad][Hs*Ws][Cs] (Cs = cout), b = backward planes,
 *         out = gradient on the input map [Ppad][Hd*Wd][N] (N = cin).
 * mode 2: stem forward (7x7, stride 2) -- a = the zero-padded 4-channel map of crw_rn_stem_fwd [Ppad][Hs][Ws][4], b = forward
 *         stem planes, out [Ppad][Hd*Wd][64].
 * mode 3: stem backward-data -- a = dZ planes [Ppad][Hs*Ws][64], b = Toeplitz planes, N = crw_rn_stem_cols(w), out [Ppad][Hd][N]:
 *         row iy holds the gradient of the (h+2) x (w+2) map at (iy, ix), channel c in column ix * 3 + c.
 * part (may be NULL): crw_rn_conv_part_floats(P, Hd*Wd, N) floats of per-tile column sums / sums of squares for crw_rn_bn_stats. */
size_t crw_rn_conv_part_floats(int P, int G, int N);
int crw_rn_conv(int mode, int P, int Hs, int Ws, int Cs, int Hd, int Wd, int N, int kh, int kw, int stride, int pad,
                const uint16_t *a_hi, const uint16_t *a_lo, const uint16_t *b_hi, const uint16_t *b_lo, const float *bias, float *out,
                float *part, crw_stream_t stream);
/* weight gradient dw [cout][cin][kh][kw] (fp32, overwritten) from the input planes x [Ppad][Hin*Win][Cin] and the dZ planes
 * [Ppad][Hout*Wout][Cout]; mode 0 = convolution / linear layer, mode 2 = stem (x = the 4-channel map, Hin x Win = its size,
 * Cin = 4; dw = [64][3][7][7]).  Reduction over patches in slices, partial slabs in ws added in a fixed order. */
size_t crw_rn_wgrad_ws_bytes(int mode, int P, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad);
int crw_rn_wgrad(int mode, int P, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad,
                 const uint16_t *x_hi, const uint16_t *x_lo, const uint16_t *d_hi, const uint16_t *d_lo, float *dw, void *ws,
                 size_t ws_bytes, crw_stream_t stream);
/* batch statistics of a convolution output with G pixels per patch from its `part` -> coef[4][C]; run_mean / run_var (may be
 * NULL together) are updated like nn.BatchNorm2d in train mode (momentum on the batch mean and the unbiased batch variance) */
size_t crw_rn_bn_stats_ws_bytes(int C);
int crw_rn_bn_stats(const float *part, int P, int G, int C, const float *gamma, const float *beta, float *run_mean, float *run_var,
                    float momentum, float eps, float *coef, void *ws, size_t ws_bytes, crw_stream_t stream);
/* the same from `rows` partial rows [rows][C] of (sum, sum of squares) over `count` samples per channel (crw_rn_stem16_fwd's) */
int crw_rn_bn_stats_rows(const float *part, int rows, double count, int C, const float *gamma, const float *beta, float *run_mean,
                         float *run_var, float momentum, float eps, float *coef, void *ws, size_t ws_bytes, crw_stream_t stream);
/* y planes = relu?( Z * scale + shift [+ Zd * scale_d + shift_d] [+ residual planes] )   (src/encoder.py:138-153) */
int crw_rn_bn_apply(const float *Z, const float *coef, const float *Zd, const float *coef_d, const uint16_t *res_hi,
                    const uint16_t *res_lo, int P, int npix, int C, int relu, uint16_t *y_hi, uint16_t *y_lo, crw_stream_t stream);
/* y planes [Ppad][Ho*Wo][C] = maxpool3x3/2/1( relu( Z * scale + shift ) ), Z on an H x W map   (src/encoder.py:257-260);
 * amax (may be NULL; [Ppad][Ho*Wo][C] bytes): position ky * 3 + kx of the first maximum of every window (ATen's rule), for the
 * backward pass */
int crw_rn_bn_pool(const float *Z, const float *coef, int P, int H, int W, int C, uint16_t *y_hi, uint16_t *y_lo, uint8_t *amax,
                   crw_stream_t stream);
/* backward of y = relu(bn(Z) [+ bn_d(Zd)] [+ identity]): g = (g1 [+ g2]) where mask_hi (the hi plane of y) is non-zero;
 * dz planes = gradient of Z, dzd planes = gradient of Zd, g_out (may be NULL) = g in fp32 (the identity shortcut's share),
 * dgamma / dbeta [C] (and the shortcut BatchNorm's). */
size_t crw_rn_bn_bwd_ws_bytes(int P, int npix, int C);
int crw_rn_bn_bwd(const float *g1, const float *g2, const uint16_t *mask_hi, const float *Z, const float *coef, const float *Zd,
                  const float *coef_d, int P, int npix, int C, uint16_t *dz_hi, uint16_t *dz_lo, uint16_t *dzd_hi, uint16_t *dzd_lo,
                  float *g_out, float *dgamma, float *dbeta, float *dgamma_d, float *dbeta_d, void *ws, size_t ws_bytes,
                  crw_stream_t stream);
/* backward of crw_rn_bn_pool: d1 (+ d2, may be NULL) = gradient of the pooled map [Ppad][Ho*Wo][C] fp32, amax = the codes the
 * forward recorded -> dz planes [Ppad][H*W][C], dgamma / dbeta of that BatchNorm */
size_t crw_rn_pool_bwd_ws_bytes(int P, int H, int W, int C);
int crw_rn_pool_bwd(const float *d1, const float *d2, const uint8_t *amax, const float *Z, const float *coef, int P, int H, int W, int C,
                    uint16_t *dz_hi, uint16_t *dz_lo, float *dgamma, float *dbeta, void *ws, size_t ws_bytes, crw_stream_t stream);
/* stem: relu0(bn0(fc0(x))) with fc0 = Conv2d(cin, 3, 1, padding 1) (src/encoder.py:66-74,87): x [P][cin][h][w] -> the map planes
 * [Ppad][Hm][Wm][4] that feed the stem convolution (the (h+2) x (w+2) map at offset (3,3), zero elsewhere, channel 3 = 0);
 * bn0's batch statistics follow from the moments of x.  stem [32] floats = the record crw_rn_stem_bwd needs. */
size_t crw_rn_stem_ws_bytes(void);
int crw_rn_stem_fwd(const float *x, int P, int cin, int h, int w, int Hm, int Wm, const float *w0, const float *b0, const float *gamma,
                    const float *beta, float *run_mean, float *run_var, float momentum, float eps, uint16_t *map_hi, uint16_t *map_lo,
                    float *stem, void *ws, size_t ws_bytes, crw_stream_t stream);
/* dX0 = out of crw_rn_conv mode 3 -> dw0 [3][cin], db0 [3], dgamma0 [3], dbeta0 [3] */
int crw_rn_stem_bwd(const float *dX0, const float *x, const float *stem, const float *w0, const float *b0, int P, int cin, int h, int w,
                    float *dw0, float *db0, float *dgamma, float *dbeta, void *ws, size_t ws_bytes, crw_stream_t stream);
/* The stem for 16 x 16 patches, one PATCH per wave (what crw_rn_train_* use at that size; csrc/resnet_stem.hip): the map
 * relu0(bn0(fc0(x))) is rebuilt from the 1 KB patch inside each kernel (LDS image), the 7 x 7 weights live in LDS in MFMA fragment
 * order, and the backward-data pass never writes the gradient map -- only the sums bn0 / fc0 need leave it.
 *   crw_rn_stem_stats   : bn0's batch statistics (from the moments of x) -> stem[32] record (+ running statistics)
 *   crw_rn_pack_stem16  : model.conv1.weight [64][3][7][7] -> forward / transposed fragment packs (28672 bf16 each)
 *   crw_rn_stem16_fwd   : Z1 [P][81][64] fp32 + crw_rn_stem16_rows() partial rows [64] of (sum, sum of squares) for crw_rn_bn_stats_rows
 *   crw_rn_stem16_wgrad : dz planes [P][81][64] (gradient of Z1) -> dw1 [64][3][7][7]
 *   crw_rn_stem16_bwd   : dz planes -> dw0 [3][cin], db0 [3], dgamma0 [3], dbeta0 [3]
 * ws of the last two: crw_rn_stem16_ws_bytes(); of crw_rn_stem_stats: crw_rn_stem_ws_bytes(). */
int crw_rn_stem_stats(const float *x, int P, int cin, int h, int w, const float *w0, const float *b0, const float *gamma, const float *beta,
                      float *run_mean, float *run_var, float momentum, float eps, float *stem, void *ws, size_t ws_bytes,
                      crw_stream_t stream);
int crw_rn_stem16_rows(void);
int crw_rn_pack_stem16(const float *w1, uint16_t *wf, uint16_t *wt, crw_stream_t stream);
int crw_rn_stem16_fwd(const float *x, int P, int cin, const float *stem, const uint16_t *wf, float *Z1, float *part, crw_stream_t stream);
/* The same forward product for patches of ANY size (what crw_rn_train_fwd / crw_rn_eval_fwd use beyond 16 x 16, e.g. the 32 x 32
 * patches of scripts/test/test_mc1.py:19): a BAND of output rows per wave, the map rows it reads rebuilt from the patch in LDS.
 * Z1 [P][H1 * W1][64] fp32 (H1, W1 = the 7x7/2 output of the (h+2) x (w+2) map), part as crw_rn_stem16_fwd's.
 * crw_rn_stem_band_ok: 1 when the geometry is covered (output rows of at most 96 pixels, band image within the LDS), else 0 --
 * then mode 2 of crw_rn_conv on the crw_rn_stem_fwd map planes is the way. */
int crw_rn_stem_band_ok(int h, int w);
int crw_rn_stem_band_fwd(const float *x, int P, int cin, int h, int w, const float *stem, const uint16_t *wf, float *Z1, float *part,
                         crw_stream_t stream);
size_t crw_rn_stem16_ws_bytes(void);
int crw_rn_stem16_wgrad(const float *x, int P, int cin, const float *stem, const uint16_t *dz_hi, const uint16_t *dz_lo, float *dw, void *ws,
                        size_t ws_bytes, crw_stream_t stream);
int crw_rn_stem16_bwd(const float *x, int P, int cin, const float *stem, const float *w0, const float *b0, const uint16_t *wt,
                      const uint16_t *dz_hi, const uint16_t *dz_lo, float *dw0, float *db0, float *dgamma, float *dbeta, void *ws,
                      size_t ws_bytes, crw_stream_t stream);
/* fp32 [P][C] -> planes [Ppad][C] */
int crw_rn_split(const float *x, int P, int C, uint16_t *hi, uint16_t *lo, crw_stream_t stream);
/* out [C] = column sums of x [rows][C] (bias gradient of the head), fixed order */
size_t crw_rn_colsum_ws_bytes(int C);
int crw_rn_colsum(const float *x, int rows, int C, float *out, void *ws, size_t ws_bytes, crw_stream_t stream);

/* The whole encoder from native code: one call runs every launch of the forward (backward) pass on one workspace -- what the
 * host modules use (resnet_hip.py); the entry points above remain as the building blocks the tests exercise one by one.
 * Patches x [P][cin][h][w] of any size (cin = 1, or 2 with pos_embed); 16 x 16 -- the reference's default, scripts/train.py:24 --
 * takes the patch-per-wave stem kernels, other sizes (32 x 32: scripts/test/test_mc1.py:19) the gathered stem products, and where
 * layer4's map has more than one pixel the global average pool + linear head (src/encoder.py:264-266) run as one product over
 * that map.  prm / grads: the CRW_RN_NPARAM parameter tensors (gradients) in
 * nn.Module.named_parameters() order -- fc0.weight, fc0.bias, bn0.weight, bn0.bias, model.conv1.weight, model.bn1.{weight,bias},
 * model.layer{1..4}.0.{conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias [, downsample.0.weight,
 * downsample.1.weight, downsample.1.bias]}, model.fc.weight, model.fc.bias; run_mean / run_var (NULL together: not updated): the
 * CRW_RN_NBN BatchNorm buffers in module order (bn0, model.bn1, then per layer bn1, bn2 [, downsample.1]).
 * ws: crw_rn_train_ws_bytes; crw_rn_train_bwd needs the workspace exactly as crw_rn_train_fwd left it (activations, statistics,
 * packed weights) and the same x / prm.  out [P][128] fp32. */
#define CRW_RN_NPARAM 42
#define CRW_RN_NBN 13
size_t crw_rn_train_ws_bytes(int P, int cin, int h, int w);
int crw_rn_train_fwd(const float *x, int P, int cin, int h, int w, const float *const *prm, float *const *run_mean,
                     float *const *run_var, float momentum, float eps, float *out, void *ws, size_t ws_bytes, crw_stream_t stream);
int crw_rn_train_bwd(const float *dout, const float *x, int P, int cin, int h, int w, const float *const *prm, float *const *grads,
                     void *ws, size_t ws_bytes, crw_stream_t stream);
/* crw_rn_train_fwd when NO backward pass follows (the reference's test scripts: train-mode BatchNorm under torch.no_grad(),
 * scripts/test/test_mc1.py:40-46 with src/utils.py:108): same out, same running-statistics update; what only crw_rn_train_bwd
 * would read (the stem's map planes and Toeplitz weight packs at patch sizes other than 16 x 16) is not produced -- the workspace
 * must NOT be handed to crw_rn_train_bwd afterwards. */
int crw_rn_train_fwd_nograd(const float *x, int P, int cin, int h, int w, const float *const *prm, float *const *run_mean,
                            float *const *run_var, float momentum, float eps, float *out, void *ws, size_t ws_bytes, crw_stream_t stream);
/* The same forward with every BatchNorm in EVAL mode (nn.Module.eval(): normalise by the running statistics, update nothing) --
 * what the reference's scripts/test/test.py:42 runs before utils.propagate.  ws: crw_rn_train_ws_bytes. */
int crw_rn_eval_fwd(const float *x, int P, int cin, int h, int w, const float *const *prm, const float *const *run_mean,
                    const float *const *run_var, float eps, float *out, void *ws, size_t ws_bytes, crw_stream_t stream);
/* Diagnostic (bench.py's in-step roofline; the one stateful corner of the library, not thread-safe): while enabled, every
 * matrix-core launch of crw_rn_train_fwd / _bwd is bracketed by two HIP events on the launch stream.  After synchronising the
 * stream, crw_rn_timing_read fills up to `max` records in launch order and returns how many were taken; enable(…) resets.
 * kind 0 = crw_rn_conv (g = Hs, Ws, Cs, Hd, Wd, N), kind 1 = crw_rn_wgrad incl. its slab sum (g = Hin, Win, Cin, Hout, Wout, Cout). */
typedef struct {
  int kind, mode, g[6], k, stride, pad;
  float ms;
} crw_rn_timing_rec;
int crw_rn_timing_enable(int on);
int crw_rn_timing_read(crw_rn_timing_rec *out, int max);

/* bf16 matrix-core variant of the chain GEMM.  A, B fp32 [batch,n,n] (n multiple of 128) are first
 * converted into bf16 images inside `ws` (convert != 0; pass 0 to reuse the images of the previous
 * call, e.g. when timing the GEMM alone).  split = 1: plain bf16 operands; split = 3: hi/lo operand
 * pairs, Ah*Bh + Ah*Bl + Al*Bh, fp32-grade result.  C fp32. */
size_t crw_gemm_bf16_ws_bytes(int n, int batch, int split);
int crw_gemm_bf16(const float *A, const float *B, float *C, int n, int batch, int transA, int transB,
                  int beta, int split, void *ws, size_t ws_bytes, int convert, crw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CRW_HIP_H */
