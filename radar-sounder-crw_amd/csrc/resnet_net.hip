// The whole Resnet encoder (reference src/encoder.py:63-89,109-155,157-272) as TWO entry points: crw_rn_train_fwd and
// crw_rn_train_bwd run every launch of the forward / backward pass from native code on one caller-provided workspace
// (~230 launches per training step: driven from Python, one ctypes call and a handful of tensor allocations per launch, the
// step was host-bound at ~8.9 ms; the kernels themselves take ~7).  The schedule is the one documented in resnet_hip.py.
#include <vector>

#include "resnet.h"

using namespace crw;

namespace {

constexpr int NPARAM = CRW_RN_NPARAM, FEAT = 128;

struct Blk {
  int cin, cout, stride, hin, win, hout, wout, down;
  int pbase;  // index of conv1.weight in the parameter list
  int bn;     // index of bn1 in the running-statistics lists (bn2 = +1, shortcut = +2)
};

struct Planes {
  uint16_t *hi, *lo;
};
struct Wpack {
  uint16_t *fh, *fl, *bh, *bl;
};

// every buffer of a step at a fixed offset of the workspace (the same function sizes it, with base = nullptr)
struct Plan {
  int P, Ppad, cin, h, w, H0, W0, H1, W1, H2, W2, Hm, Wm, ldt;
  int ncols;        // columns of the stem's input-gradient rows (3 * W0 rounded up to 64)
  int hl, wl, npl;  // layer4's map: the head averages over its npl pixels (src/encoder.py:264-266)
  Blk blk[4];
  char *base;
  size_t off = 0;
  bool ok = true;

  template <typename T>
  T *take(size_t n) {
    off = align_up(off, 256);
    T *p = reinterpret_cast<T *>(base + off);
    off += n * sizeof(T);
    return p;
  }
  Planes planes(size_t n) { return Planes{take<uint16_t>(n), take<uint16_t>(n)}; }
  Wpack wpack(size_t n) { return Wpack{take<uint16_t>(n), take<uint16_t>(n), take<uint16_t>(n), take<uint16_t>(n)}; }

  // forward (kept for the backward)
  float *stem;
  uint16_t *wsf_h, *wsf_l, *wst_h, *wst_l;
  bool stem16;           // 16 x 16 patches: the patch-per-wave stem kernels (resnet_stem.hip), no map / Toeplitz planes in HBM
  bool band;             // other sizes the band kernel covers: the stem's FORWARD product on rn_stem_fwd_band_kernel (a band of output
                         // rows per wave); the backward pass keeps the map / Toeplitz planes
  uint16_t *w16f, *w16t;  // their weight fragment packs
  float *stem_part;
  Planes xmap, A1;
  float *Z1, *coef1;
  uint8_t *amax1;  // arg-max codes of the max-pool
  struct {
    Wpack wa, wb, wd;
    float *Za, *Zb, *Zd, *ca, *cb, *cd;
    Planes Aa, Aout;
  } r[4];
  Wpack wfc;
  float *outp;  // [Ppad][128]
  float *dwfc;  // [128][512][npl]: the head's weight gradient per pixel of the averaged map (npl > 1)
  // scratch
  float *part, *part_d, *redpart;   // part_d / stats_ws_d: the shortcut branch computes beside the main one (side stream)
  double *stats_ws, *stats_ws_d;
  unsigned *tickets;  // two sets of block counters (caller's stream, side stream) for the sums-with-tail kernels; zeroed per pass
  void *stem_ws;
  Planes dO, dzb[4], dza[4], dzd[4], dz1;  // per block: the weight gradients read them on a side stream while the chain moves on
  float *g[2], *gA, *dX0;
  void *wgrad_ws, *bnbwd_ws, *poolbwd_ws, *colsum_ws;
  size_t wgrad_bytes = 0;

  static int outdim(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

  Plan(int P_, int cin_, int h_, int w_, void *ws) : P(P_), cin(cin_), h(h_), w(w_), base((char *)ws) {
    Ppad = rn_padded(P);
    H0 = h + 2; W0 = w + 2;
    H1 = outdim(H0, 7, 2, 3); W1 = outdim(W0, 7, 2, 3);
    H2 = outdim(H1, 3, 2, 1); W2 = outdim(W1, 3, 2, 1);
    Hm = std::max(H0 + 6, 2 * H1 + 6); Wm = std::max(W0 + 6, 2 * W1 + 6);
    ldt = 4 * W1 * 64;
    const int couts[4] = {64, 128, 256, 512}, strides[4] = {1, 2, 2, 2}, pbase[4] = {7, 13, 22, 31}, bnidx[4] = {2, 4, 7, 10};
    int hh = H2, ww = W2, c = 64;
    for (int i = 0; i < 4; ++i) {
      Blk &b = blk[i];
      b.cin = c; b.cout = couts[i]; b.stride = strides[i]; b.hin = hh; b.win = ww;
      b.hout = outdim(hh, 3, b.stride, 1); b.wout = outdim(ww, 3, b.stride, 1);
      b.down = (b.stride != 1 || b.cin != b.cout);
      b.pbase = pbase[i]; b.bn = bnidx[i];
      hh = b.hout; ww = b.wout; c = b.cout;
    }
    // Any patch size: where layer4's map has more than one pixel (32 x 32 patches: 2 x 2) the global average pool + linear head
    // run as ONE gathered product over the whole map -- a "convolution" whose kernel covers the map, every tap holding
    // fc.weight / npl (RnPackJob::bcast) -- so the forward, the backward-data and the weight-gradient kernels serve it unchanged;
    // the stem's backward-data product takes as many 64-column tiles as a map row needs.
    hl = hh; wl = ww; npl = hh * ww;
    ncols = rn_stem_cols(w);
    if (cin < 1 || cin > 2 || h < 1 || w < 1 || npl > 64 || H1 * W1 > 4096) ok = false;
    if (!ok) return;
    const size_t pp = (size_t)Ppad;
    stem = take<float>(32);
    stem16 = (h == 16 && w == 16);
    band = !stem16 && rn_stem_band_ok(h, w);
    if (stem16 || band) {
      w16f = take<uint16_t>(RN_STEM_FRAG_ELEMS);
      w16t = take<uint16_t>(RN_STEM_FRAG_ELEMS);
    }
    if (stem16) {
      stem_part = take<float>((size_t)rn_stem16_blocks() * 8 * 16);
    } else {
      wsf_h = take<uint16_t>(64 * 256); wsf_l = take<uint16_t>(64 * 256);
      wst_h = take<uint16_t>((size_t)H0 * ncols * ldt); wst_l = take<uint16_t>((size_t)H0 * ncols * ldt);
      xmap = planes(pp * Hm * Wm * 4);
    }
    Z1 = take<float>(pp * H1 * W1 * 64);
    coef1 = take<float>(4 * 64);
    A1 = planes(pp * H2 * W2 * 64);
    amax1 = take<uint8_t>(pp * H2 * W2 * 64);
    size_t gmax = pp * H2 * W2 * 64, part_max = std::max(crw_rn_conv_part_floats(P, H1 * W1, 64), (size_t)rn_stem16_blocks() * 8 * 128);
    for (int i = 0; i < 4; ++i) {
      const Blk &b = blk[i];
      const size_t n = pp * b.hout * b.wout * b.cout;
      r[i].wa = wpack((size_t)b.cout * b.cin * 9);
      r[i].wb = wpack((size_t)b.cout * b.cout * 9);
      r[i].wd = b.down ? wpack((size_t)b.cout * b.cin) : Wpack{};
      r[i].Za = take<float>(n); r[i].Zb = take<float>(n); r[i].Zd = b.down ? take<float>(n) : nullptr;
      r[i].ca = take<float>(4 * b.cout); r[i].cb = take<float>(4 * b.cout); r[i].cd = b.down ? take<float>(4 * b.cout) : nullptr;
      r[i].Aa = planes(n); r[i].Aout = planes(n);
      gmax = std::max(gmax, std::max(n, pp * b.hin * b.win * b.cin));
      part_max = std::max(part_max, crw_rn_conv_part_floats(P, b.hout * b.wout, b.cout));
    }
    wfc = wpack((size_t)FEAT * 512 * npl);
    outp = take<float>(pp * FEAT);
    dwfc = npl > 1 ? take<float>((size_t)FEAT * 512 * npl) : nullptr;
    part = take<float>(part_max);
    part_d = take<float>(part_max);
    redpart = take<float>(part_max * 2);  // [rows][3][C] against [rows][C][2]
    stats_ws = take<double>((size_t)64 * 2 * 512);
    stats_ws_d = take<double>((size_t)64 * 2 * 512);
    tickets = take<unsigned>(2 * RN_TICKET_BLOCKS);
    stem_ws = take<char>(rn_stem_ws_bytes());
    dO = planes(pp * FEAT);
    for (int i = 0; i < 4; ++i) {
      const size_t n = pp * blk[i].hout * blk[i].wout * blk[i].cout;
      dzb[i] = planes(n); dza[i] = planes(n); dzd[i] = blk[i].down ? planes(n) : Planes{nullptr, nullptr};
    }
    dz1 = planes(pp * H1 * W1 * 64);
    for (int i = 0; i < 2; ++i) g[i] = take<float>(gmax);
    gA = take<float>(gmax);
    dX0 = stem16 ? nullptr : take<float>(pp * H0 * ncols);
    // weight-gradient slabs: the largest of any layer
    auto need2 = [&](int mode, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int kh, int kw, int s, int pad) {
      RnWgradArgs a;
      if (rn_make_wgrad(a, mode, P, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, s, pad) != CRW_OK) { ok = false; return; }
      wgrad_bytes = std::max(wgrad_bytes, (size_t)a.S * a.ntv * a.Mtot * a.Ntot * 4);
    };
    auto need = [&](int mode, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int k, int s, int pad) {
      need2(mode, Hin, Win, Cin, Hout, Wout, Cout, k, k, s, pad);
    };
    if (stem16) wgrad_bytes = (size_t)rn_stem16_blocks() * 4 * 224 * 64 * 4;
    else need(RN_MODE_STEM_FWD, Hm, Wm, 4, H1, W1, 64, 7, 2, 3);
    need2(RN_MODE_FWD, hl, wl, 512, 1, 1, FEAT, hl, wl, 1, 0);
    size_t bnb = 0;
    for (int i = 0; i < 4; ++i) {
      const Blk &b = blk[i];
      need(RN_MODE_FWD, b.hin, b.win, b.cin, b.hout, b.wout, b.cout, 3, b.stride, 1);
      need(RN_MODE_FWD, b.hout, b.wout, b.cout, b.hout, b.wout, b.cout, 3, 1, 1);
      if (b.down) need(RN_MODE_FWD, b.hin, b.win, b.cin, b.hout, b.wout, b.cout, 1, b.stride, 0);
      bnb = std::max(bnb, rn_bn_bwd_ws_bytes(P, b.hout * b.wout, b.cout));
    }
    wgrad_ws = take<char>(wgrad_bytes);
    bnbwd_ws = take<char>(bnb);
    poolbwd_ws = take<char>(rn_pool_bwd_ws_bytes(P, H1, W1, 64));
    colsum_ws = take<char>(rn_colsum_ws_bytes(FEAT));
    off = align_up(off, 256);
  }
};

// ---- optional in-step timing: two HIP events around every matrix-core launch, on the launch stream ------------------------
struct TimingState {
  bool on = false;
  std::vector<hipEvent_t> pool;
  std::vector<crw_rn_timing_rec> recs;
  size_t used = 0;
} g_tm;

struct Timed {
  hipStream_t s;
  bool live;
  size_t idx;
  Timed(hipStream_t s_, int kind, int mode, int a, int b, int c, int d, int e, int f, int k, int st, int pad) : s(s_), live(g_tm.on) {
    if (!live) return;
    if (g_tm.used + 2 > g_tm.pool.size()) {
      for (int i = 0; i < 64; ++i) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) { live = false; return; }
        g_tm.pool.push_back(ev);
      }
    }
    idx = g_tm.used;
    g_tm.used += 2;
    g_tm.recs.push_back(crw_rn_timing_rec{kind, mode, {a, b, c, d, e, f}, k, st, pad, 0.f});
    (void)hipEventRecord(g_tm.pool[idx], s);
  }
  ~Timed() {
    if (live) (void)hipEventRecord(g_tm.pool[idx + 1], s);
  }
};

// the BatchNorm-backward sums a backward-data product can take in its epilogue (for the layer that consumes its gradient)
struct Red {
  const uint16_t *mask = nullptr;
  const float *z = nullptr, *coef = nullptr, *zd = nullptr, *coefd = nullptr;
  float *part = nullptr;
};

// ---- side stream: the weight gradients leave the serial backward chain ---------------------------------------------------
// The chain bn_bwd -> backward-data product -> bn_bwd -> ... is serial and carries ~50 us-scale merge / finalize launches that
// leave the chip nearly idle; the weight gradients depend only on planes the chain has already produced, so they run on a second
// HIP stream beside it (fork by event after the apply pass that writes their dZ planes, join at the end of the pass).
struct SideStream {
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  bool ok = true;
  hipStream_t get() {
    if (!stream && ok && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) ok = false;
    return stream;
  }
  hipEvent_t event() {
    if (used == ev.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ok = false; return nullptr; }
      ev.push_back(e);
    }
    return ev[used++];
  }
  // everything enqueued on `from` so far happens before what is enqueued on `to` from now on
  bool order(hipStream_t from, hipStream_t to) {
    hipEvent_t e = event();
    return e && hipEventRecord(e, from) == hipSuccess && hipStreamWaitEvent(to, e, 0) == hipSuccess;
  }
};
// One side stream (and event pool) per DEVICE, created on first use while that device is current -- a process that drives several
// GPUs gets a stream on each.  The pools are not locked: ONE host thread per device may be inside crw_rn_train_* / crw_rn_eval_fwd
// at a time (include/crw_hip.h, "Threads").
constexpr int MAX_DEVICES = 64;
SideStream g_sides[MAX_DEVICES];
SideStream &side_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  return g_sides[dev];
}

// k = kh * 256 + kw for a kernel that is not square (the head over an hl x wl map), else the side
int conv(hipStream_t s, int mode, int P, int Hs, int Ws, int Cs, int Hd, int Wd, int N, int k, int stride, int pad, Planes a, const uint16_t *bh,
         const uint16_t *bl, const float *bias, float *out, float *part, bool accumulate = false, const Red &red = Red()) {
  RnConvArgs q;
  const int kh = k >= 256 ? k >> 8 : k, kw = k >= 256 ? k & 255 : k;
  CRW_TRY(rn_make_conv(q, mode, P, Hs, Ws, Cs, Hd, Wd, N, kh, kw, stride, pad));
  q.a_hi = a.hi; q.a_lo = a.lo; q.b_hi = bh; q.b_lo = bl; q.out = out; q.part = part; q.bias = bias;
  q.accumulate = accumulate ? 1 : 0;
  q.red_mask = red.mask; q.red_z = red.z; q.red_coef = red.coef; q.red_zd = red.zd; q.red_coefd = red.coefd; q.red_part = red.part;
  Timed t(s, 0, mode, Hs, Ws, Cs, Hd, Wd, N, k, stride, pad);
  return launch_rn_conv(q, s);
}

int wgrad(hipStream_t s, int mode, int P, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int k, int stride, int pad, Planes x,
          Planes d, float *dw, void *ws) {
  RnWgradArgs q;
  const int kh = k >= 256 ? k >> 8 : k, kw = k >= 256 ? k & 255 : k;
  CRW_TRY(rn_make_wgrad(q, mode, P, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride, pad));
  q.x_hi = x.hi; q.x_lo = x.lo; q.d_hi = d.hi; q.d_lo = d.lo;
  q.slab = (float *)ws;
  Timed t(s, 1, mode, Hin, Win, Cin, Hout, Wout, Cout, k, stride, pad);
  return launch_rn_wgrad(q, dw, s);
}


// Whatever a pass has put on the side stream is joined back into the caller's stream on EVERY exit, error paths included: the caller
// frees / re-uses the workspace and the gradient buffer in stream order of ITS stream.
struct SideJoin {
  SideStream &side;
  hipStream_t s, sw;
  ~SideJoin() {
    if (sw != s && !side.order(sw, s)) (void)hipStreamSynchronize(sw);
  }
};

// The forward pass.  training: BatchNorm on batch statistics (+ running-statistics update when run_mean / run_var are given);
// otherwise (nn.Module.eval(), the reference's scripts/test/test.py:42) BatchNorm on the RUNNING statistics, nothing updated.
// keep: a backward pass may follow (crw_rn_train_bwd reads the workspace); false: what only that pass would read is not produced
// (the stem's 4-channel map planes and Toeplitz weight packs at patch sizes on the band kernel)
int forward_pass(bool training, bool keep, const float *x, int P, int cin, int h, int w, const float *const *prm, float *const *run_mean,
                 float *const *run_var, float momentum, float eps, float *out, void *ws, size_t ws_bytes, hipStream_t s) {
  if (!x || !prm || !out || !ws || P < 1 || (run_mean == nullptr) != (run_var == nullptr) || (!training && !run_mean)) return CRW_EINVAL;
  for (int i = 0; i < NPARAM; ++i)
    if (!prm[i]) return CRW_EINVAL;
  Plan pl(P, cin, h, w, ws);
  if (!pl.ok) return CRW_EINVAL;
  if (ws_bytes < pl.off) return CRW_EWORKSPACE;
  auto rm = [&](int i) { return run_mean ? run_mean[i] : nullptr; };
  auto rv = [&](int i) { return run_var ? run_var[i] : nullptr; };
  // one BatchNorm's coef[4][C]: batch statistics from the producing product's per-tile partials, or the running statistics
  auto bn_coef = [&](const float *part, int rows, int C, double cnt, const float *gamma, const float *beta, int bn, float *coef, double *sws,
                     hipStream_t st) {
    if (training)
      return launch_rn_bn_stats(part, rows, C, cnt, gamma, beta, rm(bn), rv(bn), momentum, eps, coef, sws,
                                pl.tickets + (sws == pl.stats_ws_d ? RN_TICKET_BLOCKS : 0), st);
    return launch_rn_bn_coef_eval(gamma, beta, rm(bn), rv(bn), eps, C, coef, st);
  };
  // side stream: the weight packing runs beside the stem (which does not need it), each shortcut convolution + its statistics
  // beside its block's main branch (CRW_RN_STREAMS=0: everything on the caller's stream)
  static const bool use_side = !(getenv("CRW_RN_STREAMS") && getenv("CRW_RN_STREAMS")[0] == '0');
  SideStream &g_side = side_of_current_device();
  hipStream_t sw = use_side ? g_side.get() : nullptr;
  if (!sw) sw = s;
  g_side.used = 0;
  auto fork = [&]() { return sw == s || g_side.order(s, sw) ? CRW_OK : CRW_EHIP; };
  auto join = [&]() { return sw == s || g_side.order(sw, s) ? CRW_OK : CRW_EHIP; };
  SideJoin guard{g_side, s, sw};
  CRW_TRY(rn_zero_tickets(pl.tickets, 2, s));

  // all convolution / linear weights -> hi / lo planes (forward and backward-data layouts), one launch
  CRW_TRY(fork());
  {
    RnPackJobs jobs{};
    auto add = [&](const float *wsrc, const Wpack &d, int cout, int cin_, int T) {
      jobs.job[jobs.n++] = RnPackJob{wsrc, d.fh, d.fl, d.bh, d.bl, cout, cin_, T, 0};
    };
    for (int i = 0; i < 4; ++i) {
      const Blk &b = pl.blk[i];
      add(prm[b.pbase], pl.r[i].wa, b.cout, b.cin, 9);
      add(prm[b.pbase + 3], pl.r[i].wb, b.cout, b.cout, 9);
      if (b.down) add(prm[b.pbase + 6], pl.r[i].wd, b.cout, b.cin, 1);
    }
    add(prm[40], pl.wfc, FEAT, 512, pl.npl);
    if (pl.npl > 1) {  // the head behind the average pool: every pixel of layer4's map takes fc.weight / npl
      jobs.job[jobs.n - 1].bcast = 1;
      jobs.job[jobs.n - 1].scale = 1.f / (float)pl.npl;
    }
    CRW_TRY(launch_rn_pack_all(jobs, sw));
  }
  // stem: fc0 + bn0 + relu0 -> 4-channel map; 7x7/2 convolution + statistics; bn1 + relu + max-pool
  if (pl.stem16) {
    CRW_TRY(launch_rn_pack_stem_frag(prm[4], pl.w16f, pl.w16t, s));
    if (training) CRW_TRY(launch_rn_stem_stats(x, P, cin, h, w, prm[0], prm[1], prm[2], prm[3], rm(0), rv(0), momentum, eps, pl.stem, pl.stem_ws, s));
    else CRW_TRY(launch_rn_stem_eval(cin, prm[0], prm[1], prm[2], prm[3], rm(0), rv(0), eps, pl.stem, s));
    {
      Timed t(s, 0, RN_MODE_STEM_FWD, 24, 24, 4, pl.H1, pl.W1, 64, 7, 2, 3);
      CRW_TRY(launch_rn_stem16_fwd(x, P, cin, pl.stem, pl.w16f, pl.Z1, pl.part, s));
    }
    CRW_TRY(bn_coef(pl.part, rn_stem16_blocks() * 8, 64, (double)P * pl.H1 * pl.W1, prm[5], prm[6], 1, pl.coef1, pl.stats_ws, s));
  } else {
    // the Toeplitz / row packs of the 7x7 weights: with the band kernel only the backward pass reads them -- beside the stem then
    const bool planes_needed = !pl.band || (training && keep);  // the gathered forward product, or the backward pass, reads them
    if (planes_needed)
      CRW_TRY(launch_rn_pack_stem(prm[4], pl.H0, pl.W0, pl.H1, pl.W1, pl.ldt, pl.ncols, pl.wsf_h, pl.wsf_l, pl.wst_h, pl.wst_l, pl.band ? sw : s));
    if (pl.band) CRW_TRY(launch_rn_pack_stem_frag(prm[4], pl.w16f, pl.w16t, s));
    if (training && planes_needed) {  // (bn0's statistics; the 4-channel map planes are what the backward pass multiplies)
      CRW_TRY(launch_rn_stem_fwd(x, P, pl.Ppad, cin, h, w, pl.Hm, pl.Wm, prm[0], prm[1], prm[2], prm[3], rm(0), rv(0), momentum, eps, pl.xmap.hi,
                                 pl.xmap.lo, pl.stem, pl.stem_ws, s));
    } else if (training) {            // bn0's statistics alone
      CRW_TRY(launch_rn_stem_stats(x, P, cin, h, w, prm[0], prm[1], prm[2], prm[3], rm(0), rv(0), momentum, eps, pl.stem, pl.stem_ws, s));
    } else {
      CRW_TRY(launch_rn_stem_eval(cin, prm[0], prm[1], prm[2], prm[3], rm(0), rv(0), eps, pl.stem, s));
      if (!pl.band) CRW_TRY(launch_rn_stem_apply(x, P, pl.Ppad, cin, h, w, pl.Hm, pl.Wm, pl.stem, pl.xmap.hi, pl.xmap.lo, s));
    }
    if (pl.band) {  // the product itself: a band of output rows per wave, map rebuilt from the patch in LDS (resnet_stem.hip)
      {
        Timed t(s, 0, RN_MODE_STEM_FWD, pl.Hm, pl.Wm, 4, pl.H1, pl.W1, 64, 7, 2, 3);
        CRW_TRY(launch_rn_stem_band_fwd(x, P, cin, h, w, pl.stem, pl.w16f, pl.Z1, pl.part, s));
      }
      CRW_TRY(bn_coef(pl.part, rn_stem16_blocks() * 8, 64, (double)P * pl.H1 * pl.W1, prm[5], prm[6], 1, pl.coef1, pl.stats_ws, s));
    } else {
      CRW_TRY(conv(s, RN_MODE_STEM_FWD, P, pl.Hm, pl.Wm, 4, pl.H1, pl.W1, 64, 7, 2, 3, pl.xmap, pl.wsf_h, pl.wsf_l, nullptr, pl.Z1,
                   training ? pl.part : nullptr));
      CRW_TRY(bn_coef(pl.part, (pl.Ppad / 128) * 2 * pl.H1 * pl.W1, 64, (double)P * pl.H1 * pl.W1, prm[5], prm[6], 1, pl.coef1, pl.stats_ws, s));
    }
  }
  CRW_TRY(launch_rn_bn_pool(pl.Z1, pl.coef1, P, pl.Ppad, pl.H1, pl.W1, 64, pl.A1.hi, pl.A1.lo, pl.amax1, s));

  CRW_TRY(join());  // packed weights ready
  Planes A = pl.A1;
  for (int i = 0; i < 4; ++i) {
    const Blk &b = pl.blk[i];
    auto &r = pl.r[i];
    const int npix = b.hout * b.wout, rows = (pl.Ppad / 128) * 2 * npix;
    const double cnt = (double)P * npix;
    const float *const *q = prm + b.pbase;
    if (b.down) {  // shortcut: 1x1 / stride-2 convolution of the block input + its statistics, beside the main branch
      CRW_TRY(fork());
      CRW_TRY(conv(sw, RN_MODE_FWD, P, b.hin, b.win, b.cin, b.hout, b.wout, b.cout, 1, b.stride, 0, A, r.wd.fh, r.wd.fl, nullptr, r.Zd, training ? pl.part_d : nullptr));
      CRW_TRY(bn_coef(pl.part_d, rows, b.cout, cnt, q[7], q[8], b.bn + 2, r.cd, pl.stats_ws_d, sw));
    }
    float *part = training ? pl.part : nullptr;
    CRW_TRY(conv(s, RN_MODE_FWD, P, b.hin, b.win, b.cin, b.hout, b.wout, b.cout, 3, b.stride, 1, A, r.wa.fh, r.wa.fl, nullptr, r.Za, part));
    CRW_TRY(bn_coef(pl.part, rows, b.cout, cnt, q[1], q[2], b.bn, r.ca, pl.stats_ws, s));
    CRW_TRY(launch_rn_bn_apply(r.Za, r.ca, nullptr, nullptr, nullptr, nullptr, P, pl.Ppad, npix, b.cout, 1, r.Aa.hi, r.Aa.lo, s));
    CRW_TRY(conv(s, RN_MODE_FWD, P, b.hout, b.wout, b.cout, b.hout, b.wout, b.cout, 3, 1, 1, r.Aa, r.wb.fh, r.wb.fl, nullptr, r.Zb, part));
    CRW_TRY(bn_coef(pl.part, rows, b.cout, cnt, q[4], q[5], b.bn + 1, r.cb, pl.stats_ws, s));
    if (b.down) {
      CRW_TRY(join());
      CRW_TRY(launch_rn_bn_apply(r.Zb, r.cb, r.Zd, r.cd, nullptr, nullptr, P, pl.Ppad, npix, b.cout, 1, r.Aout.hi, r.Aout.lo, s));
    } else {
      CRW_TRY(launch_rn_bn_apply(r.Zb, r.cb, nullptr, nullptr, A.hi, A.lo, P, pl.Ppad, npix, b.cout, 1, r.Aout.hi, r.Aout.lo, s));
    }
    A = r.Aout;
  }
  // head: global average pool + linear 512 -> 128 with bias, one product over layer4's whole map (a 1 x 1 map: the plain linear layer)
  CRW_TRY(conv(s, RN_MODE_FWD, P, pl.hl, pl.wl, 512, 1, 1, FEAT, pl.npl > 1 ? pl.hl * 256 + pl.wl : 1, 1, 0, A, pl.wfc.fh, pl.wfc.fl, prm[41],
               pl.outp, nullptr));
  if (hipMemcpyAsync(out, pl.outp, (size_t)P * FEAT * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    g_last_hip_error = (int)hipGetLastError();
    return CRW_EHIP;
  }
  return CRW_OK;
}

}  // namespace

extern "C" {

size_t crw_rn_train_ws_bytes(int P, int cin, int h, int w) {
  if (P < 1) return 0;
  Plan pl(P, cin, h, w, nullptr);
  return pl.ok ? pl.off : 0;
}

int crw_rn_train_fwd(const float *x, int P, int cin, int h, int w, const float *const *prm, float *const *run_mean,
                     float *const *run_var, float momentum, float eps, float *out, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  return forward_pass(true, true, x, P, cin, h, w, prm, run_mean, run_var, momentum, eps, out, ws, ws_bytes, (hipStream_t)stream);
}

int crw_rn_train_fwd_nograd(const float *x, int P, int cin, int h, int w, const float *const *prm, float *const *run_mean,
                            float *const *run_var, float momentum, float eps, float *out, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  return forward_pass(true, false, x, P, cin, h, w, prm, run_mean, run_var, momentum, eps, out, ws, ws_bytes, (hipStream_t)stream);
}

int crw_rn_eval_fwd(const float *x, int P, int cin, int h, int w, const float *const *prm, const float *const *run_mean,
                    const float *const *run_var, float eps, float *out, void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  return forward_pass(false, false, x, P, cin, h, w, prm, const_cast<float *const *>(reinterpret_cast<const float *const *>(run_mean)),
                      const_cast<float *const *>(reinterpret_cast<const float *const *>(run_var)), 0.f, eps, out, ws, ws_bytes,
                      (hipStream_t)stream);
}

int crw_rn_train_bwd(const float *dout, const float *x, int P, int cin, int h, int w, const float *const *prm, float *const *grads,
                     void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!dout || !x || !prm || !grads || !ws || P < 1) return CRW_EINVAL;
  for (int i = 0; i < NPARAM; ++i)
    if (!prm[i] || !grads[i]) return CRW_EINVAL;
  Plan pl(P, cin, h, w, ws);
  if (!pl.ok) return CRW_EINVAL;
  if (ws_bytes < pl.off) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // weight gradients on the side stream (CRW_RN_STREAMS=0: everything on the caller's stream)
  static const bool use_side = !(getenv("CRW_RN_STREAMS") && getenv("CRW_RN_STREAMS")[0] == '0');
  SideStream &g_side = side_of_current_device();
  hipStream_t sw = use_side ? g_side.get() : nullptr;
  if (!sw) sw = s;
  g_side.used = 0;
  auto fork = [&]() { return sw == s || g_side.order(s, sw) ? CRW_OK : CRW_EHIP; };  // side stream sees what the chain has produced
  SideJoin guard{g_side, s, sw};  // error paths included: the caller's stream waits for whatever the side stream still holds
  CRW_TRY(rn_zero_tickets(pl.tickets, 2, s));

  // head (one product over layer4's hl x wl map, see Plan): per-pixel weight gradients, then their mean = fc.weight's gradient
  const int khead = pl.npl > 1 ? pl.hl * 256 + pl.wl : 1;
  CRW_TRY(launch_rn_split(dout, P, pl.Ppad, FEAT, pl.dO.hi, pl.dO.lo, s));
  CRW_TRY(fork());
  CRW_TRY(wgrad(sw, RN_MODE_FWD, P, pl.hl, pl.wl, 512, 1, 1, FEAT, khead, 1, 0, pl.r[3].Aout, pl.dO, pl.npl > 1 ? pl.dwfc : grads[40], pl.wgrad_ws));
  if (pl.npl > 1) CRW_TRY(launch_rn_tapsum(pl.dwfc, (long)FEAT * 512, pl.npl, 1.f / (float)pl.npl, grads[40], sw));
  CRW_TRY(launch_rn_colsum(dout, P, FEAT, grads[41], pl.colsum_ws, s));
  // Gradients meet at every block output (main branch + shortcut): the first product writes, the second ADDS in its epilogue.
  // The product that completes a gradient also takes the BatchNorm-backward sums of the layer it feeds in its epilogue
  // (RnConvArgs::red_*): the tile goes through LDS and comes back row-contiguous, so the extra mask / Z reads are whole lines.
  // The separate reduce pass (10 B per element re-read from HBM) disappears: 5.04 -> 4.8 ms per step.  CRW_RN_FUSE_RED=0 keeps it.
  // (Straight from the accumulator layout -- 64-byte row pieces -- the same fusion measured SLOWER than the separate pass.)
  static const bool fuse_red = !(getenv("CRW_RN_FUSE_RED") && getenv("CRW_RN_FUSE_RED")[0] == '0');
  const int rrows = pl.Ppad / 128 * 2;  // partial rows per group
  float *g = pl.g[0];                   // gradient of the current block's output
  {
    const auto &r3 = pl.r[3];
    CRW_TRY(conv(s, RN_MODE_BWD, P, 1, 1, FEAT, pl.hl, pl.wl, 512, khead, 1, 0, pl.dO, pl.wfc.bh, pl.wfc.bl, nullptr, g, nullptr, false,
                 fuse_red ? Red{r3.Aout.hi, r3.Zb, r3.cb, r3.Zd, r3.cd, pl.redpart} : Red()));
  }
  for (int i = 3; i >= 0; --i) {
    const Blk &b = pl.blk[i];
    auto &r = pl.r[i];
    const int npix = b.hout * b.wout;
    float *const *gq = grads + b.pbase;
    float *gin = g == pl.g[0] ? pl.g[1] : pl.g[0];  // gradient of the block's input (= the previous block's output)
    const Planes Ain = i == 0 ? pl.A1 : pl.r[i - 1].Aout;
    // block output: bn2 (+ the shortcut's BatchNorm); an identity shortcut hands the masked gradient on in `gin`
    CRW_TRY(launch_rn_bn_bwd(g, nullptr, r.Aout.hi, r.Zb, r.cb, r.Zd, r.cd, P, pl.Ppad, npix, b.cout, pl.dzb[i].hi, pl.dzb[i].lo, b.down ? pl.dzd[i].hi : nullptr,
                             b.down ? pl.dzd[i].lo : nullptr, b.down ? nullptr : gin, gq[4], gq[5], b.down ? gq[7] : nullptr,
                             b.down ? gq[8] : nullptr, pl.bnbwd_ws, pl.tickets, s, fuse_red ? pl.redpart : nullptr, rrows * npix));
    CRW_TRY(fork());
    CRW_TRY(wgrad(sw, RN_MODE_FWD, P, b.hout, b.wout, b.cout, b.hout, b.wout, b.cout, 3, 1, 1, r.Aa, pl.dzb[i], gq[3], pl.wgrad_ws));
    if (b.down) CRW_TRY(wgrad(sw, RN_MODE_FWD, P, b.hin, b.win, b.cin, b.hout, b.wout, b.cout, 1, b.stride, 0, Ain, pl.dzd[i], gq[6], pl.wgrad_ws));
    CRW_TRY(conv(s, RN_MODE_BWD, P, b.hout, b.wout, b.cout, b.hout, b.wout, b.cout, 3, 1, 1, pl.dzb[i], r.wb.bh, r.wb.bl, nullptr, pl.gA, nullptr, false,
                 fuse_red ? Red{r.Aa.hi, r.Za, r.ca, nullptr, nullptr, pl.redpart} : Red()));
    CRW_TRY(launch_rn_bn_bwd(pl.gA, nullptr, r.Aa.hi, r.Za, r.ca, nullptr, nullptr, P, pl.Ppad, npix, b.cout, pl.dza[i].hi, pl.dza[i].lo, nullptr, nullptr,
                             nullptr, gq[1], gq[2], nullptr, nullptr, pl.bnbwd_ws, pl.tickets, s, fuse_red ? pl.redpart : nullptr, rrows * npix));
    CRW_TRY(fork());
    CRW_TRY(wgrad(sw, RN_MODE_FWD, P, b.hin, b.win, b.cin, b.hout, b.wout, b.cout, 3, b.stride, 1, Ain, pl.dza[i], gq[0], pl.wgrad_ws));
    if (b.down) {
      // input gradient = main branch (written) + shortcut (added); the sum feeds the previous block's output BatchNorms
      const auto &rp = pl.r[i - 1];  // blocks 1..3 have the shortcut convolution, so i >= 1 here
      CRW_TRY(conv(s, RN_MODE_BWD, P, b.hout, b.wout, b.cout, b.hin, b.win, b.cin, 3, b.stride, 1, pl.dza[i], r.wa.bh, r.wa.bl, nullptr, gin, nullptr));
      CRW_TRY(conv(s, RN_MODE_BWD, P, b.hout, b.wout, b.cout, b.hin, b.win, b.cin, 1, b.stride, 0, pl.dzd[i], r.wd.bh, r.wd.bl, nullptr, gin, nullptr, true,
                   fuse_red ? Red{rp.Aout.hi, rp.Zb, rp.cb, rp.Zd, rp.cd, pl.redpart} : Red()));
    } else {
      // identity shortcut (layer1): `gin` holds the masked gradient of the block output; the main branch adds to it
      CRW_TRY(conv(s, RN_MODE_BWD, P, b.hout, b.wout, b.cout, b.hin, b.win, b.cin, 3, b.stride, 1, pl.dza[i], r.wa.bh, r.wa.bl, nullptr, gin, nullptr, true));
    }
    g = gin;
  }
  float *g1 = g, *g2 = nullptr;
  // max-pool + bn1, stem convolution, stem
  CRW_TRY(launch_rn_pool_bwd(g1, g2, pl.amax1, pl.Z1, pl.coef1, P, pl.Ppad, pl.H1, pl.W1, 64, pl.dz1.hi, pl.dz1.lo, grads[5], grads[6], pl.poolbwd_ws, pl.tickets, s));
  auto join = [&]() { return sw == s || g_side.order(sw, s) ? CRW_OK : CRW_EHIP; };  // the caller's stream waits for the side stream
  if (pl.stem16) {
    CRW_TRY(fork());
    {
      Timed t(sw, 1, RN_MODE_STEM_FWD, 24, 24, 4, pl.H1, pl.W1, 64, 7, 2, 3);
      CRW_TRY(launch_rn_stem16_wgrad(x, P, cin, pl.stem, pl.dz1.hi, pl.dz1.lo, (float *)pl.wgrad_ws, sw));
      CRW_TRY(launch_rn_stem_slab_reduce((const float *)pl.wgrad_ws, rn_stem16_blocks() * 4, grads[4], sw));
    }
    {
      Timed t(s, 0, RN_MODE_STEM_BWD, pl.H1, pl.W1, 64, pl.H0, 1, 64, 7, 2, 3);
      CRW_TRY(launch_rn_stem16_bwd(x, P, cin, pl.stem, prm[0], prm[1], pl.w16t, pl.dz1.hi, pl.dz1.lo, pl.stem_part, s));
    }
    CRW_TRY(launch_rn_stem_bwd_finalize(pl.stem_part, rn_stem16_blocks() * 8, cin, pl.stem, prm[0], prm[1], grads[0], grads[1], grads[2],
                                        grads[3], pl.stem_ws, s));
    return join();
  }
  CRW_TRY(join());
  CRW_TRY(wgrad(s, RN_MODE_STEM_FWD, P, pl.Hm, pl.Wm, 4, pl.H1, pl.W1, 64, 7, 2, 3, pl.xmap, pl.dz1, grads[4], pl.wgrad_ws));
  CRW_TRY(conv(s, RN_MODE_STEM_BWD, P, pl.H1, pl.W1, 64, pl.H0, 1, pl.ncols, 7, 2, 3, pl.dz1, pl.wst_h, pl.wst_l, nullptr, pl.dX0, nullptr));
  CRW_TRY(launch_rn_stem_bwd(pl.dX0, x, pl.stem, prm[0], prm[1], P, cin, h, w, pl.ncols, grads[0], grads[1], grads[2], grads[3], pl.stem_ws, s));
  return CRW_OK;
}

int crw_rn_timing_enable(int on) {
  g_tm.on = on != 0;
  g_tm.used = 0;
  g_tm.recs.clear();
  return CRW_OK;
}

int crw_rn_timing_read(crw_rn_timing_rec *out, int max) {
  const int n = (int)std::min<size_t>(g_tm.recs.size(), max > 0 ? (size_t)max : 0);
  for (int i = 0; i < n; ++i) {
    crw_rn_timing_rec rec = g_tm.recs[i];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_tm.pool[2 * i], g_tm.pool[2 * i + 1]) != hipSuccess) {
      (void)hipGetLastError();
      ms = -1.f;
    }
    rec.ms = ms;
    out[i] = rec;
  }
  return (int)g_tm.recs.size();
}

}  // extern "C"
