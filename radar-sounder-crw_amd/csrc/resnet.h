// Internal declarations shared by the Resnet encoder kernels (resnet_gemm.hip, resnet_bn.hip) and their C-ABI entry points
// (resnet.hip).  Public interface: the crw_rn_* functions of include/crw_hip.h.
#pragma once
#include "crw_common.h"

namespace crw {

// which gather the "NN" kernel performs (see resnet_gemm.hip)
constexpr int RN_MODE_FWD = 0;       // convolution forward: group = output pixel
constexpr int RN_MODE_BWD = 1;       // backward-data: group = input pixel
constexpr int RN_MODE_STEM_FWD = 2;  // 7x7/2 stem convolution on the zero-padded 4-channel map
constexpr int RN_MODE_STEM_BWD = 3;  // its backward-data against the Toeplitz weight planes: group = map row

struct RnConvArgs {
  const uint16_t *a_hi, *a_lo;  // source planes [Mpad][lda]
  const uint16_t *b_hi, *b_lo;  // weight planes [N][ldb] (k-contiguous)
  float *out;                   // [Mpad][ldc]: group g writes columns [g*N, (g+1)*N)
  float *part;                  // per-tile column sums / sums of squares [mtiles*2][G][N] float2, or null
  const float *bias;            // [N] or null
  int accumulate;               // epilogue adds the tile to what `out` holds (the second gradient arriving at a junction)
  // backward-data only: BatchNorm-backward sums of the layer that consumes this gradient, from the epilogue (all [Mpad][ldc] like
  // `out`): red_mask = hi plane of that layer's output activation (ReLU gate), red_z (+ red_zd: the shortcut's) = its raw
  // convolution output(s), red_coef / red_coefd = their coef[4][N]; red_part [mtiles*2][G][2|3][N] partial sums.  null: off
  const uint16_t *red_mask;
  const float *red_z, *red_coef, *red_zd, *red_coefd;
  float *red_part;
  long b_group_stride;          // RN_MODE_STEM_BWD: elements between the weight planes of consecutive groups
  int lda, ldb, ldc, N, G, mtiles;
  int mode;
  int Hs, Ws, Cs;  // source map (A operand): height, width, channels
  int Hd, Wd;      // destination map: G = Hd * Wd (RN_MODE_STEM_BWD: G = Hd)
  int KH, KW, S, PAD;
};
int launch_rn_conv(const RnConvArgs &a, hipStream_t s);

struct RnWgradArgs {
  const uint16_t *x_hi, *x_lo;  // input-activation planes [Ppad][lda]
  const uint16_t *d_hi, *d_lo;  // dZ planes [Ppad][ldb]
  float *slab;                  // [S][ntv][Mtot][Ntot] partial sums
  int lda, ldb, Mtot, Ntot, taps, ktiles_p /* Ppad / 64 */, S;
  int mode;                     // RN_MODE_FWD (taps of a convolution) or RN_MODE_STEM_FWD
  int Hin, Win, Cin, Hout, Wout, Cout, KH, KW, St, PAD;
  int rshift, rstride;          // r -> (r >> rshift) * rstride + (r & ((1 << rshift) - 1)); no segmentation: rshift = 30
  int maxpair;                  // capacity of the kernel's (input pixel, output pixel) table in LDS: Hout * Wout rounded up to 64
  int ntv;                      // taps that reach the input map for at least one output pixel (the others have a zero gradient) ...
  unsigned char tapv[64];       // ... their indices; the slabs hold only these
  signed char tapinv[64];       // tap -> position in tapv, or -1
};
int rn_wgrad_slices(const RnWgradArgs &a);
int launch_rn_wgrad(const RnWgradArgs &a, float *dw, hipStream_t s);

// resnet.hip: launch geometry from the layer description
int rn_padded(int P);
int rn_make_conv(RnConvArgs &a, int mode, int P, int Hs, int Ws, int Cs, int Hd, int Wd, int N, int kh, int kw, int stride, int pad);
int rn_make_wgrad(RnWgradArgs &a, int mode, int P, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride,
                  int pad);

// resnet_bn.hip ------------------------------------------------------------------------------------------------------------
// The sums-with-tail kernels (rn_sums_tail_kernel) count their blocks on RN_TICKET_BLOCKS counters of the CALLER's: `tickets` = one
// zeroed set (RN_TICKET_BYTES) per stream that may run such a kernel at the same time; the kernels leave them zeroed.
constexpr int RN_TICKET_BLOCKS = 32, RN_TICKET_BYTES = RN_TICKET_BLOCKS * 4;
int rn_zero_tickets(unsigned *tickets, int sets, hipStream_t s);
int launch_rn_bn_stats(const float *part, int R, int C, double count, const float *gamma, const float *beta, float *run_mean,
                       float *run_var, float momentum, float eps, float *coef, double *ws /* 64 * 2C doubles */, unsigned *tickets,
                       hipStream_t s);
int launch_rn_bn_apply(const float *Z, const float *coef, const float *Zd, const float *coef_d, const uint16_t *res_hi,
                       const uint16_t *res_lo, int P, int Ppad, int npix, int C, int relu, uint16_t *y_hi, uint16_t *y_lo, hipStream_t s);
int launch_rn_bn_pool(const float *Z, const float *coef, int P, int Ppad, int H, int W, int C, uint16_t *y_hi, uint16_t *y_lo,
                      uint8_t *amax, hipStream_t s);
size_t rn_bn_bwd_ws_bytes(int P, int npix, int C);
int launch_rn_bn_bwd(const float *g1, const float *g2, const uint16_t *mask_hi, const float *Z, const float *coef, const float *Zd,
                     const float *coef_d, int P, int Ppad, int npix, int C, uint16_t *dz_hi, uint16_t *dz_lo, uint16_t *dzd_hi,
                     uint16_t *dzd_lo, float *g_out, float *dgamma, float *dbeta, float *dgamma_d, float *dbeta_d, void *ws,
                     unsigned *tickets, hipStream_t s, const float *ext_part = nullptr, int ext_rows = 0);
size_t rn_pool_bwd_ws_bytes(int P, int H, int W, int C);
int launch_rn_pool_bwd(const float *d1, const float *d2, const uint8_t *amax, const float *Z, const float *coef, int P, int Ppad, int H,
                       int W, int C, uint16_t *dz_hi, uint16_t *dz_lo, float *dgamma, float *dbeta, void *ws, unsigned *tickets, hipStream_t s);
size_t rn_stem_ws_bytes();
int launch_rn_stem_fwd(const float *x, int P, int Ppad, int cin, int h, int w, int Hm, int Wm, const float *w0, const float *b0,
                       const float *gamma, const float *beta, float *run_mean, float *run_var, float momentum, float eps,
                       uint16_t *m_hi, uint16_t *m_lo, float *stem, void *ws, hipStream_t s);
int launch_rn_stem_apply(const float *x, int P, int Ppad, int cin, int h, int w, int Hm, int Wm, const float *stem, uint16_t *m_hi,
                         uint16_t *m_lo, hipStream_t s);
int launch_rn_stem_stats(const float *x, int P, int cin, int h, int w, const float *w0, const float *b0, const float *gamma,
                         const float *beta, float *run_mean, float *run_var, float momentum, float eps, float *stem, void *ws,
                         hipStream_t s);
int launch_rn_stem_bwd_finalize(const float *part, int rows, int cin, const float *stem, const float *w0, const float *b0, float *dw0,
                                float *db0, float *dgamma, float *dbeta, void *ws /* 64 * 16 doubles */, hipStream_t s);
int launch_rn_stem_bwd(const float *dX0, const float *x, const float *stem, const float *w0, const float *b0, int P, int cin, int h, int w,
                       int ldx, float *dw0, float *db0, float *dgamma, float *dbeta, void *ws, hipStream_t s);
int launch_rn_pack_conv(const float *w, int cout, int cin, int T, uint16_t *fh, uint16_t *fl, uint16_t *bh, uint16_t *bl, hipStream_t s);
constexpr int RN_MAX_PACK_JOBS = 16;
struct RnPackJob {
  const float *w;
  uint16_t *fh, *fl, *bh, *bl;
  int cout, cin, T, first_block;
  // bcast: the source is [cout][cin] and every one of the T taps receives w * scale -- the linear head behind a global average pool
  // over T pixels, run as ONE gathered product over the whole final map (src/encoder.py:264-266)
  int bcast;
  float scale;
};
struct RnPackJobs {
  RnPackJob job[RN_MAX_PACK_JOBS];
  int n;
};
int launch_rn_pack_all(RnPackJobs &jobs, hipStream_t s);
int rn_stem_cols(int w);  // columns of the stem's input-gradient rows: 3 * (w + 2) rounded up to 64
int launch_rn_pack_stem(const float *w1, int H0, int W0, int H1, int W1, int ldt, int ncols, uint16_t *fh, uint16_t *fl, uint16_t *th,
                        uint16_t *tl, hipStream_t s);
// dw [cout][cin] = scale * sum over the T taps of src [cout][cin][T]
int launch_rn_tapsum(const float *src, long n, int T, float scale, float *dw, hipStream_t s);
// eval-mode BatchNorm: coef [4][C] (scale, shift, mean, invstd) from the running statistics; stem record from bn0's
int launch_rn_bn_coef_eval(const float *gamma, const float *beta, const float *run_mean, const float *run_var, float eps, int C, float *coef,
                           hipStream_t s);
int launch_rn_stem_eval(int cin, const float *w0, const float *b0, const float *gamma, const float *beta, const float *run_mean,
                        const float *run_var, float eps, float *stem, hipStream_t s);
int launch_rn_split(const float *x, long rows, long rows_pad, int C, uint16_t *hi, uint16_t *lo, hipStream_t s);
size_t rn_colsum_ws_bytes(int W);
int launch_rn_colsum(const float *x, int R, int W, float *out, void *ws, hipStream_t s);

// resnet_stem.hip: the stem convolution for 16 x 16 patches, a patch per wave -----------------------------------------------
constexpr int RN_STEM_FRAG_ELEMS = 7 * 4 * 2 * 512;  // bf16 elements of one fragment-ordered weight pack (hi and lo interleaved per fragment)
int rn_stem16_blocks();
int launch_rn_pack_stem_frag(const float *w1, uint16_t *wf, uint16_t *wt, hipStream_t s);
int launch_rn_stem16_fwd(const float *x, int P, int cin, const float *stem, const uint16_t *wf, float *Z1, float *part /* [blocks*8][64][2] */,
                         hipStream_t s);
// the same forward product for patches of any size, a band of output rows per wave (false: geometry not covered)
bool rn_stem_band_ok(int h, int w);
int launch_rn_stem_band_fwd(const float *x, int P, int cin, int h, int w, const float *stem, const uint16_t *wf, float *Z1,
                            float *part /* [blocks*8][64][2] */, hipStream_t s);
int launch_rn_stem16_wgrad(const float *x, int P, int cin, const float *stem, const uint16_t *dz_hi, const uint16_t *dz_lo,
                           float *slab /* [blocks*4][224][64] */, hipStream_t s);
int launch_rn_stem16_bwd(const float *x, int P, int cin, const float *stem, const float *w0, const float *b0, const uint16_t *wt,
                         const uint16_t *dz_hi, const uint16_t *dz_lo, float *part /* [blocks*8][16] */, hipStream_t s);
int launch_rn_stem_slab_reduce(const float *slab, int nslab, float *dw, hipStream_t s);

}  // namespace crw
