// C-ABI entry points of the Resnet encoder kernels (include/crw_hip.h, "crw_rn_*"): argument checks and launch geometry.
// Kernels: resnet_gemm.hip (matrix-core convolutions / weight gradients), resnet_bn.hip (BatchNorm, pooling, stem, packing).
#include <algorithm>

#include "resnet.h"

using namespace crw;

namespace {
inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
}  // namespace
namespace crw {
int rn_padded(int P) { return round_up(P, 128); }
}
#define padded crw::rn_padded

extern "C" {

int crw_rn_padded_patches(int P) { return P < 1 ? 0 : padded(P); }

int crw_rn_pack_conv(const float *w, int cout, int cin, int kh, int kw, uint16_t *fwd_hi, uint16_t *fwd_lo, uint16_t *bwd_hi,
                     uint16_t *bwd_lo, crw_stream_t stream) {
  clear_stale_error();
  if (!w || !fwd_hi || !fwd_lo || !bwd_hi || !bwd_lo || cout < 1 || cin < 1 || kh < 1 || kw < 1) return CRW_EINVAL;
  return launch_rn_pack_conv(w, cout, cin, kh * kw, fwd_hi, fwd_lo, bwd_hi, bwd_lo, (hipStream_t)stream);
}

int crw_rn_stem_toeplitz_ld(int w) { return w < 1 ? 0 : 4 * ((w + 1) / 2 + 1) * 64; }  // 4 kernel rows x W1 x 64
int crw_rn_stem_cols(int w) { return w < 1 ? 0 : rn_stem_cols(w); }

int crw_rn_pack_stem(const float *w1, int h, int w, uint16_t *fwd_hi, uint16_t *fwd_lo, uint16_t *toep_hi, uint16_t *toep_lo,
                     crw_stream_t stream) {
  clear_stale_error();
  if (!w1 || !fwd_hi || !fwd_lo || !toep_hi || !toep_lo || h < 1 || w < 1) return CRW_EINVAL;
  const int H0 = h + 2, W0 = w + 2, H1 = (H0 - 1) / 2 + 1, W1 = (W0 - 1) / 2 + 1;
  return launch_rn_pack_stem(w1, H0, W0, H1, W1, 4 * W1 * 64, rn_stem_cols(w), fwd_hi, fwd_lo, toep_hi, toep_lo, (hipStream_t)stream);
}

size_t crw_rn_conv_part_floats(int P, int G, int N) {
  if (P < 1 || G < 1 || N < 1) return 0;
  return (size_t)(padded(P) / 128) * 2 * G * N * 2;
}

}  // extern "C"

namespace crw {
// geometry of one launch of the gathered matrix product (pointers are filled in by the caller)
int rn_make_conv(RnConvArgs &a, int mode, int P, int Hs, int Ws, int Cs, int Hd, int Wd, int N, int kh, int kw, int stride, int pad) {
  if (P < 1 || Hs < 1 || Ws < 1 || Cs < 1 || Hd < 1 || Wd < 1 || N < 64 || N % 64 || kh < 1 || kw < 1 || stride < 1 || pad < 0 ||
      mode < RN_MODE_FWD || mode > RN_MODE_STEM_BWD)
    return CRW_EINVAL;
  a = RnConvArgs{};
  a.mode = mode; a.Hs = Hs; a.Ws = Ws; a.Cs = Cs; a.Hd = Hd; a.Wd = Wd; a.N = N;
  a.KH = kh; a.KW = kw; a.S = stride; a.PAD = pad;
  a.mtiles = padded(P) / 128;
  a.lda = Hs * Ws * Cs;
  if (mode == RN_MODE_FWD || mode == RN_MODE_BWD) {
    if (Cs % 64) return CRW_EINVAL;
    a.G = Hd * Wd;
    a.ldb = kh * kw * Cs;
  } else if (mode == RN_MODE_STEM_FWD) {
    // source = zero-padded 4-channel map; the 32-value segments of the last output pixel must stay inside the map
    if (Cs != 4 || kh != 7 || kw != 7 || (stride * (Hd - 1) + 8) > Hs || (stride * (Wd - 1)) * 4 + 32 > Ws * 4) return CRW_EINVAL;
    a.G = Hd * Wd;
    a.ldb = 256;
  } else {
    if (Cs != 64 || kh != 7 || stride != 2 || pad != 3 || N < 3 * (2 * Ws - 1)) return CRW_EINVAL;  // N = crw_rn_stem_cols(w): a whole map row
    a.G = Hd;
    a.ldb = 4 * Ws * Cs;
    a.b_group_stride = (long)N * a.ldb;
  }
  a.ldc = a.G * N;
  return CRW_OK;
}
}  // namespace crw

extern "C" {

int crw_rn_conv(int mode, int P, int Hs, int Ws, int Cs, int Hd, int Wd, int N, int kh, int kw, int stride, int pad,
                const uint16_t *a_hi, const uint16_t *a_lo, const uint16_t *b_hi, const uint16_t *b_lo, const float *bias, float *out,
                float *part, crw_stream_t stream) {
  clear_stale_error();
  if (!a_hi || !a_lo || !b_hi || !b_lo || !out) return CRW_EINVAL;
  if (!aligned16(a_hi) || !aligned16(a_lo) || !aligned16(b_hi) || !aligned16(b_lo)) return CRW_EINVAL;
  RnConvArgs a;
  CRW_TRY(rn_make_conv(a, mode, P, Hs, Ws, Cs, Hd, Wd, N, kh, kw, stride, pad));
  a.a_hi = a_hi; a.a_lo = a_lo; a.b_hi = b_hi; a.b_lo = b_lo; a.out = out; a.part = part; a.bias = bias;
  return launch_rn_conv(a, (hipStream_t)stream);
}

}  // extern "C"

namespace crw {
int rn_make_wgrad(RnWgradArgs &a, int mode, int P, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride,
                  int pad) {
  if (P < 1 || Hin < 1 || Win < 1 || Cin < 1 || Hout < 1 || Wout < 1 || Cout < 64 || Cout % 64 || kh < 1 || kw < 1 || stride < 1 ||
      pad < 0 || (mode != RN_MODE_FWD && mode != RN_MODE_STEM_FWD))
    return CRW_EINVAL;
  a = RnWgradArgs{};
  a.mode = mode; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.KH = kh; a.KW = kw; a.St = stride; a.PAD = pad;
  a.lda = Hin * Win * Cin;
  a.ldb = Hout * Wout * Cout;
  a.maxpair = std::max(128, round_up(Hout * Wout, 64));
  if (a.maxpair > 4096) return CRW_EINVAL;  // resnet_gemm.hip RN_MAXPAIR_CAP: output maps up to 64 x 64
  a.Ntot = Cout;
  a.ktiles_p = padded(P) / 64;
  if (mode == RN_MODE_FWD) {
    if (Cin % 64) return CRW_EINVAL;
    a.Mtot = Cin;
    a.taps = kh * kw;
    a.rshift = 30;
    a.rstride = 0;
  } else {
    if (Cin != 4 || kh != 7 || kw != 7) return CRW_EINVAL;
    a.Mtot = 256;  // 8 kernel rows x 32 (the 8th row and the 8th column are dropped by the reduce kernel)
    a.taps = 1;
    a.rshift = 5;
    a.rstride = Win * Cin;
  }
  // taps that reach the input map for at least one output pixel
  a.ntv = 0;
  for (int t = 0; t < a.taps && t < 64; ++t) {
    bool any = mode != RN_MODE_FWD;
    if (!any) {
      const int ky = t / kw, kx = t % kw;
      bool anyy = false, anyx = false;
      for (int o = 0; o < Hout; ++o) anyy |= (o * stride + ky - pad >= 0 && o * stride + ky - pad < Hin);
      for (int o = 0; o < Wout; ++o) anyx |= (o * stride + kx - pad >= 0 && o * stride + kx - pad < Win);
      any = anyy && anyx;
    }
    a.tapinv[t] = any ? (signed char)a.ntv : (signed char)-1;
    if (any) a.tapv[a.ntv++] = (unsigned char)t;
  }
  if (a.taps > 64 || a.ntv < 1) return CRW_EINVAL;
  a.S = 1;
  a.S = rn_wgrad_slices(a);
  return CRW_OK;
}
}  // namespace crw

extern "C" {

size_t crw_rn_wgrad_ws_bytes(int mode, int P, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad) {
  RnWgradArgs a;
  if (rn_make_wgrad(a, mode, P, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride, pad) != CRW_OK) return 0;
  return (size_t)a.S * a.ntv * a.Mtot * a.Ntot * 4;
}

int crw_rn_wgrad(int mode, int P, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad,
                 const uint16_t *x_hi, const uint16_t *x_lo, const uint16_t *d_hi, const uint16_t *d_lo, float *dw, void *ws,
                 size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  RnWgradArgs a;
  CRW_TRY(rn_make_wgrad(a, mode, P, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride, pad));
  if (!x_hi || !x_lo || !d_hi || !d_lo || !dw || !ws) return CRW_EINVAL;
  if (ws_bytes < (size_t)a.S * a.ntv * a.Mtot * a.Ntot * 4) return CRW_EWORKSPACE;
  a.x_hi = x_hi; a.x_lo = x_lo; a.d_hi = d_hi; a.d_lo = d_lo;
  a.slab = (float *)ws;
  return launch_rn_wgrad(a, dw, (hipStream_t)stream);
}

// the sums' doubles, then one set of block tickets (zeroed by the entry point before every launch: the library keeps no state)
static size_t stats_doubles(int C) { return (size_t)64 * 2 * C; }
size_t crw_rn_bn_stats_ws_bytes(int C) { return C < 1 ? 0 : stats_doubles(C) * 8 + RN_TICKET_BYTES; }

int crw_rn_bn_stats(const float *part, int P, int G, int C, const float *gamma, const float *beta, float *run_mean, float *run_var,
                    float momentum, float eps, float *coef, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!part || !gamma || !beta || !coef || !ws || P < 1 || G < 1 || C < 1 || (run_mean == nullptr) != (run_var == nullptr))
    return CRW_EINVAL;
  if (ws_bytes < crw_rn_bn_stats_ws_bytes(C)) return CRW_EWORKSPACE;
  unsigned *tk = (unsigned *)((double *)ws + stats_doubles(C));
  CRW_TRY(rn_zero_tickets(tk, 1, (hipStream_t)stream));
  return launch_rn_bn_stats(part, (padded(P) / 128) * 2 * G, C, (double)P * G, gamma, beta, run_mean, run_var, momentum, eps, coef,
                            (double *)ws, tk, (hipStream_t)stream);
}

int crw_rn_bn_stats_rows(const float *part, int rows, double count, int C, const float *gamma, const float *beta, float *run_mean,
                         float *run_var, float momentum, float eps, float *coef, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!part || !gamma || !beta || !coef || !ws || rows < 1 || count < 1.0 || C < 1 || (run_mean == nullptr) != (run_var == nullptr))
    return CRW_EINVAL;
  if (ws_bytes < crw_rn_bn_stats_ws_bytes(C)) return CRW_EWORKSPACE;
  unsigned *tk = (unsigned *)((double *)ws + stats_doubles(C));
  CRW_TRY(rn_zero_tickets(tk, 1, (hipStream_t)stream));
  return launch_rn_bn_stats(part, rows, C, count, gamma, beta, run_mean, run_var, momentum, eps, coef, (double *)ws, tk, (hipStream_t)stream);
}

int crw_rn_bn_apply(const float *Z, const float *coef, const float *Zd, const float *coef_d, const uint16_t *res_hi,
                    const uint16_t *res_lo, int P, int npix, int C, int relu, uint16_t *y_hi, uint16_t *y_lo, crw_stream_t stream) {
  clear_stale_error();
  if (!Z || !coef || !y_hi || !y_lo || P < 1 || npix < 1 || C < 8 || C % 8 || (Zd == nullptr) != (coef_d == nullptr) ||
      (res_hi == nullptr) != (res_lo == nullptr))
    return CRW_EINVAL;
  return launch_rn_bn_apply(Z, coef, Zd, coef_d, res_hi, res_lo, P, padded(P), npix, C, relu, y_hi, y_lo, (hipStream_t)stream);
}

int crw_rn_bn_pool(const float *Z, const float *coef, int P, int H, int W, int C, uint16_t *y_hi, uint16_t *y_lo, uint8_t *amax,
                   crw_stream_t stream) {
  clear_stale_error();
  if (!Z || !coef || !y_hi || !y_lo || P < 1 || H < 1 || W < 1 || C < 8 || C % 8) return CRW_EINVAL;
  return launch_rn_bn_pool(Z, coef, P, padded(P), H, W, C, y_hi, y_lo, amax, (hipStream_t)stream);
}

size_t crw_rn_bn_bwd_ws_bytes(int P, int npix, int C) {
  return (P < 1 || npix < 1 || C < 8) ? 0 : rn_bn_bwd_ws_bytes(P, npix, C) + RN_TICKET_BYTES;
}

int crw_rn_bn_bwd(const float *g1, const float *g2, const uint16_t *mask_hi, const float *Z, const float *coef, const float *Zd,
                  const float *coef_d, int P, int npix, int C, uint16_t *dz_hi, uint16_t *dz_lo, uint16_t *dzd_hi, uint16_t *dzd_lo,
                  float *g_out, float *dgamma, float *dbeta, float *dgamma_d, float *dbeta_d, void *ws, size_t ws_bytes,
                  crw_stream_t stream) {
  clear_stale_error();
  if (!g1 || !mask_hi || !Z || !coef || !dz_hi || !dz_lo || !dgamma || !dbeta || !ws || P < 1 || npix < 1 || C < 64 || C % 64 ||
      C > 2048 || (C & (C - 1)))
    return CRW_EINVAL;
  if ((Zd != nullptr) && (!coef_d || !dzd_hi || !dzd_lo || !dgamma_d || !dbeta_d)) return CRW_EINVAL;
  if (ws_bytes < crw_rn_bn_bwd_ws_bytes(P, npix, C)) return CRW_EWORKSPACE;
  unsigned *tk = (unsigned *)((char *)ws + rn_bn_bwd_ws_bytes(P, npix, C));
  CRW_TRY(rn_zero_tickets(tk, 1, (hipStream_t)stream));
  return launch_rn_bn_bwd(g1, g2, mask_hi, Z, coef, Zd, coef_d, P, padded(P), npix, C, dz_hi, dz_lo, Zd ? dzd_hi : nullptr,
                          Zd ? dzd_lo : nullptr, g_out, dgamma, dbeta, dgamma_d, dbeta_d, ws, tk, (hipStream_t)stream);
}

size_t crw_rn_pool_bwd_ws_bytes(int P, int H, int W, int C) {
  return (P < 1 || H < 1 || W < 1 || C < 64) ? 0 : rn_pool_bwd_ws_bytes(P, H, W, C) + RN_TICKET_BYTES;
}

int crw_rn_pool_bwd(const float *d1, const float *d2, const uint8_t *amax, const float *Z, const float *coef, int P, int H, int W, int C,
                    uint16_t *dz_hi, uint16_t *dz_lo, float *dgamma, float *dbeta, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!d1 || !amax || !Z || !coef || !dz_hi || !dz_lo || !dgamma || !dbeta || !ws || P < 1 || H < 1 || W < 1 || C < 64 || C % 64 ||
      (C & (C - 1)) || C > 2048)
    return CRW_EINVAL;
  if (ws_bytes < crw_rn_pool_bwd_ws_bytes(P, H, W, C)) return CRW_EWORKSPACE;
  unsigned *tk = (unsigned *)((char *)ws + rn_pool_bwd_ws_bytes(P, H, W, C));
  CRW_TRY(rn_zero_tickets(tk, 1, (hipStream_t)stream));
  return launch_rn_pool_bwd(d1, d2, amax, Z, coef, P, padded(P), H, W, C, dz_hi, dz_lo, dgamma, dbeta, ws, tk, (hipStream_t)stream);
}

size_t crw_rn_stem_ws_bytes(void) { return rn_stem_ws_bytes(); }

int crw_rn_stem_fwd(const float *x, int P, int cin, int h, int w, int Hm, int Wm, const float *w0, const float *b0, const float *gamma,
                    const float *beta, float *run_mean, float *run_var, float momentum, float eps, uint16_t *map_hi, uint16_t *map_lo,
                    float *stem, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !w0 || !b0 || !gamma || !beta || !map_hi || !map_lo || !stem || !ws || P < 1 || cin < 1 || cin > 2 || h < 1 || w < 1 ||
      Hm < h + 8 || Wm < w + 8 || (run_mean == nullptr) != (run_var == nullptr))
    return CRW_EINVAL;
  if (ws_bytes < rn_stem_ws_bytes()) return CRW_EWORKSPACE;
  return launch_rn_stem_fwd(x, P, padded(P), cin, h, w, Hm, Wm, w0, b0, gamma, beta, run_mean, run_var, momentum, eps, map_hi, map_lo,
                            stem, ws, (hipStream_t)stream);
}

int crw_rn_stem_bwd(const float *dX0, const float *x, const float *stem, const float *w0, const float *b0, int P, int cin, int h, int w,
                    float *dw0, float *db0, float *dgamma, float *dbeta, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!dX0 || !x || !stem || !w0 || !b0 || !dw0 || !db0 || !dgamma || !dbeta || !ws || P < 1 || cin < 1 || cin > 2 || h < 1 || w < 1)
    return CRW_EINVAL;
  if (ws_bytes < rn_stem_ws_bytes()) return CRW_EWORKSPACE;
  return launch_rn_stem_bwd(dX0, x, stem, w0, b0, P, cin, h, w, rn_stem_cols(w), dw0, db0, dgamma, dbeta, ws, (hipStream_t)stream);
}

/* ---- stem convolution for 16 x 16 patches, a patch per wave (resnet_stem.hip) ---- */
int crw_rn_stem_stats(const float *x, int P, int cin, int h, int w, const float *w0, const float *b0, const float *gamma, const float *beta,
                      float *run_mean, float *run_var, float momentum, float eps, float *stem, void *ws, size_t ws_bytes,
                      crw_stream_t stream) {
  clear_stale_error();
  if (!x || !w0 || !b0 || !gamma || !beta || !stem || !ws || P < 1 || cin < 1 || cin > 2 || h < 1 || w < 1 ||
      (run_mean == nullptr) != (run_var == nullptr))
    return CRW_EINVAL;
  if (ws_bytes < rn_stem_ws_bytes()) return CRW_EWORKSPACE;
  return launch_rn_stem_stats(x, P, cin, h, w, w0, b0, gamma, beta, run_mean, run_var, momentum, eps, stem, ws, (hipStream_t)stream);
}

int crw_rn_stem16_rows(void) { return rn_stem16_blocks() * 8; }

int crw_rn_pack_stem16(const float *w1, uint16_t *wf, uint16_t *wt, crw_stream_t stream) {
  clear_stale_error();
  if (!w1 || !wf || !wt) return CRW_EINVAL;
  return launch_rn_pack_stem_frag(w1, wf, wt, (hipStream_t)stream);
}

int crw_rn_stem16_fwd(const float *x, int P, int cin, const float *stem, const uint16_t *wf, float *Z1, float *part, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !stem || !wf || !Z1 || !part || P < 1 || cin < 1 || cin > 2) return CRW_EINVAL;
  return launch_rn_stem16_fwd(x, P, cin, stem, wf, Z1, part, (hipStream_t)stream);
}

int crw_rn_stem_band_ok(int h, int w) { return rn_stem_band_ok(h, w) ? 1 : 0; }

int crw_rn_stem_band_fwd(const float *x, int P, int cin, int h, int w, const float *stem, const uint16_t *wf, float *Z1, float *part,
                         crw_stream_t stream) {
  clear_stale_error();
  if (!x || !stem || !wf || !Z1 || !part || P < 1 || cin < 1 || cin > 2 || !rn_stem_band_ok(h, w)) return CRW_EINVAL;
  return launch_rn_stem_band_fwd(x, P, cin, h, w, stem, wf, Z1, part, (hipStream_t)stream);
}

size_t crw_rn_stem16_ws_bytes(void) {
  return align_up((size_t)rn_stem16_blocks() * 4 * 224 * 64 * 4, 256) + (size_t)rn_stem16_blocks() * 8 * 16 * 4 + 64 * 16 * 8 + 256;
}

int crw_rn_stem16_wgrad(const float *x, int P, int cin, const float *stem, const uint16_t *dz_hi, const uint16_t *dz_lo, float *dw, void *ws,
                        size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !stem || !dz_hi || !dz_lo || !dw || !ws || P < 1 || cin < 1 || cin > 2) return CRW_EINVAL;
  if (ws_bytes < crw_rn_stem16_ws_bytes()) return CRW_EWORKSPACE;
  CRW_TRY(launch_rn_stem16_wgrad(x, P, cin, stem, dz_hi, dz_lo, (float *)ws, (hipStream_t)stream));
  return launch_rn_stem_slab_reduce((const float *)ws, rn_stem16_blocks() * 4, dw, (hipStream_t)stream);
}

int crw_rn_stem16_bwd(const float *x, int P, int cin, const float *stem, const float *w0, const float *b0, const uint16_t *wt,
                      const uint16_t *dz_hi, const uint16_t *dz_lo, float *dw0, float *db0, float *dgamma, float *dbeta, void *ws,
                      size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !stem || !w0 || !b0 || !wt || !dz_hi || !dz_lo || !dw0 || !db0 || !dgamma || !dbeta || !ws || P < 1 || cin < 1 || cin > 2)
    return CRW_EINVAL;
  if (ws_bytes < crw_rn_stem16_ws_bytes()) return CRW_EWORKSPACE;
  float *part = (float *)ws;
  void *ws2 = (char *)ws + align_up((size_t)rn_stem16_blocks() * 8 * 16 * 4, 256);
  CRW_TRY(launch_rn_stem16_bwd(x, P, cin, stem, w0, b0, wt, dz_hi, dz_lo, part, (hipStream_t)stream));
  return launch_rn_stem_bwd_finalize(part, rn_stem16_blocks() * 8, cin, stem, w0, b0, dw0, db0, dgamma, dbeta, ws2, (hipStream_t)stream);
}

int crw_rn_split(const float *x, int P, int C, uint16_t *hi, uint16_t *lo, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !hi || !lo || P < 1 || C < 8 || C % 8) return CRW_EINVAL;
  return launch_rn_split(x, P, padded(P), C, hi, lo, (hipStream_t)stream);
}

size_t crw_rn_colsum_ws_bytes(int C) { return C < 1 ? 0 : rn_colsum_ws_bytes(C); }

int crw_rn_colsum(const float *x, int rows, int C, float *out, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !out || !ws || rows < 1 || C < 1) return CRW_EINVAL;
  if (ws_bytes < rn_colsum_ws_bytes(C)) return CRW_EWORKSPACE;
  return launch_rn_colsum(x, rows, C, out, ws, (hipStream_t)stream);
}

}  // extern "C"
