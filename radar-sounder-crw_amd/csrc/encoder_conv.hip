// Hand-written conv stack for the 3x3 layers of the CNN encoder (conv3, conv4, conv5 of
// src/encoder.py:26-35 -- 96.7 % of the encoder's flops).
//
// Implicit GEMM on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), fp32 accumulate:
//     out[p, y, x, co] = sum_{dy,dx,ci} in[p, y+dy-1, x+dx-1, ci] * W[co, ci, dy, dx]
//   M = the 100 pixels of one 10x10 patch (7 MFMA row tiles), N = Cout, K = 9 taps x Cin.
//   SPLIT = 3: activations and weights are (hi, lo) bf16 pairs and every product is
//              Ah*Bh + Ah*Bl + Al*Bh -> fp32-grade results (parity path);
//   SPLIT = 1: plain bf16 operands (throughput path).
//
// Layout: activations are channels-last with a one-pixel zero halo, [P][12*12][C] bf16 planes
// (hi and lo), so a tap is a constant pixel offset and no border predication exists.  One
// workgroup = one patch: the patch (all channels, both planes) is loaded into LDS once and serves
// all 9 taps x all output channels; rows are XOR-swizzled by pixel so the 16 pixels of an MFMA row
// tile hit distinct banks.  Waves split the output channels; weight fragments ([tap][co][ci], 16 B
// per lane) stream straight from L2 into registers, prefetched one k-step ahead.  The epilogue
// stages the output tile through the same LDS and writes whole 16-byte chunks.
//
// The same kernel is the backward-data pass: dX = conv(dY, W flipped, ci <-> co), with the ReLU
// mask of the layer below applied in the epilogue instead of bias + ReLU.
//
// conv3x3_wgrad_kernel: dW[co,ci,tap] = sum_{p,pix} dY[p,pix,co] * X[p,pix+tap,ci]: both operands
// are pixel-major in memory, i.e. strided along the reduction dimension, so fragments come from
// LDS through the hardware-transposing ds_read_b64_tr_b16.  One workgroup owns one tap row (3
// taps), walks a slice of the patches accumulating in registers, then adds its partial sums to
// the fp32 gradient with atomics.
#include "crw_common.h"

namespace crw {
namespace {

constexpr int IMG_W = 10, PAD_W = 12, NPIX = 100, NPAD = 144, MT = 7;  // 7 row tiles of 16 = 112 >= 100

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char *lds_cp;

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// byte offset of 16-byte chunk `chunk` of pixel `pp` in an LDS plane with C channels
template <int C>
__device__ inline int px_off(int pp, int chunk) {
  constexpr int NCH = C / 8;                             // chunks per pixel
  constexpr int RPB = (C * 2 >= 256) ? 1 : 256 / (C * 2);  // pixels per 256-byte bank row
  return pp * (C * 2) + 16 * (chunk ^ ((pp / RPB) % NCH));
}

struct ConvArgs {
  const uint16_t *xh, *xl;   // [P][144][CIN] input planes (zero halo)
  const uint16_t *wh, *wl;   // [9][COUT][CIN] weights (already flipped/transposed for backward-data)
  const float *bias;         // [COUT] (MODE 0) or null
  const uint16_t *maskh;     // [P][144][COUT]: output is zeroed where this plane is 0 (MODE 1) or null
  uint16_t *yh, *yl;         // [P][144][COUT] output planes (zero halo written) or null
  float *yf;                 // optional fp32 output [P][100][COUT]
  float *gap;                // optional [P][COUT]: mean over the 100 pixels (MODE 0)
  int P;
};

template <int SPLIT, int CIN, int COUT, int MODE>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
  constexpr int NPL = (SPLIT == 3) ? 2 : 1;
  constexpr int CMAX = CIN > COUT ? CIN : COUT;
  constexpr int PLANE = NPAD * CMAX * 2;  // bytes per LDS plane (input image, later output staging)
  constexpr int WN = (COUT / 16 >= 4) ? 4 : COUT / 16;  // waves across the output channels
  constexpr int WM = 4 / WN;                            // waves across the pixel row tiles
  constexpr int NTW = COUT / 16 / WN;                   // 16-wide column tiles per wave
  constexpr int MTW = (MT + WM - 1) / WM;               // row tiles per wave (tile wm + WM*k)
  constexpr int KCH = CIN / 32;                         // 32-deep k-steps per tap
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;

  // ---- patch -> LDS (swizzled) ---------------------------------------------------------------
  {
    constexpr int NCH = CIN / 8, TOTAL = NPAD * NCH;
    for (int pl = 0; pl < NPL; ++pl) {
      const uint16_t *src = (pl ? a.xl : a.xh) + (long)p * NPAD * CIN;
      char *dst = lds + pl * PLANE;
      for (int c = tid; c < TOTAL; c += 256) {
        const int pix = c / NCH, ch = c % NCH;
        *reinterpret_cast<uint4 *>(dst + px_off<CIN>(pix, ch)) = *reinterpret_cast<const uint4 *>(src + (long)c * 8);
      }
    }
  }
  __syncthreads();

  // row tile mt, row r16 -> interior pixel i -> padded index of the tap-(0,0) source pixel
  const int wm = wave / WN, wn = wave % WN;
  int pp0[MTW];
#pragma unroll
  for (int k = 0; k < MTW; ++k) {
    int i = 16 * (wm + WM * k) + r16;
    if (i >= NPIX) i = 0;  // dummy rows recompute pixel 0 and are dropped in the epilogue
    pp0[k] = (i / IMG_W) * PAD_W + (i % IMG_W);
  }

  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int k = 0; k < MTW; ++k)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int co_w = wn * (16 * NTW);  // first output channel of this wave
  auto load_b = [&](int step, bf8 (&bh)[NTW], bf8 (&bl)[NTW]) {
    const int tap = step / KCH, cc = step % KCH;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const long o = ((long)tap * COUT + co_w + 16 * j + r16) * CIN + 32 * cc + 8 * g;
      bh[j] = *reinterpret_cast<const bf8 *>(a.wh + o);
      if (SPLIT == 3) bl[j] = *reinterpret_cast<const bf8 *>(a.wl + o);
    }
  };

  constexpr int NSTEP = 9 * KCH;
  auto do_step = [&](int step, bf8 (&bhc)[NTW], bf8 (&blc)[NTW], bf8 (&bhn)[NTW], bf8 (&bln)[NTW]) {
    if (step + 1 < NSTEP) load_b(step + 1, bhn, bln);  // weights of the next k-step stream in behind the MFMAs
    const int tap = step / KCH, cc = step % KCH;
    const int toff = (tap / 3) * PAD_W + (tap % 3);
#pragma unroll
    for (int k = 0; k < MTW; ++k) {
      if (wm + WM * k < MT) {  // wave-uniform
        const int pp = pp0[k] + toff;
        const bf8 ah = *reinterpret_cast<const bf8 *>(lds + px_off<CIN>(pp, 4 * cc + g));
        bf8 al;
        if (SPLIT == 3) al = *reinterpret_cast<const bf8 *>(lds + PLANE + px_off<CIN>(pp, 4 * cc + g));
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          if (SPLIT == 3) {
            acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhc[j], acc[k][j], 0, 0, 0);
            acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, blc[j], acc[k][j], 0, 0, 0);
          }
          acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhc[j], acc[k][j], 0, 0, 0);
        }
      }
    }
  };
  // two register sets for the weight fragments, addressed statically (a run-time index would send
  // them to scratch)
  bf8 bh0[NTW], bl0[NTW], bh1[NTW], bl1[NTW];
  load_b(0, bh0, bl0);
  for (int step = 0; step + 1 < NSTEP; step += 2) {
    do_step(step, bh0, bl0, bh1, bl1);
    do_step(step + 1, bh1, bl1, bh0, bl0);
  }
  if (NSTEP & 1) do_step(NSTEP - 1, bh0, bl0, bh1, bl1);

  // ---- epilogue ----------------------------------------------------------------------------------
  // C/D map: acc[k][j][r] = out[pixel 16 (wm + WM k) + 4 g + r][channel co_w + 16 j + r16]
  float gsum[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) gsum[j] = 0.f;
  __syncthreads();  // every wave is done reading the input image: reuse LDS as the output staging
  if (a.yh) {       // zero the staging planes (the halo stays zero)
    constexpr int OUTB = NPAD * COUT * 2;
    for (int pl = 0; pl < NPL; ++pl)
      for (int c = tid; c < OUTB / 16; c += 256) *reinterpret_cast<uint4 *>(lds + pl * PLANE + 16 * c) = uint4{0, 0, 0, 0};
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < MTW; ++k)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int co = co_w + 16 * j + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * (wm + WM * k) + 4 * g + r;
        if (i < NPIX) {
          const int pp = (i / IMG_W + 1) * PAD_W + (i % IMG_W + 1);
          float v = acc[k][j][r];
          if (MODE == 0) {
            if (a.bias) v += a.bias[co];
            v = fmaxf(v, 0.f);
            gsum[j] += v;
          } else if (a.maskh) {
            if ((a.maskh[((long)p * NPAD + pp) * COUT + co] & 0x7fff) == 0) v = 0.f;
          }
          if (a.yf) a.yf[((long)p * NPIX + i) * COUT + co] = v;
          if (a.yh) {
            const uint16_t h = f2bf(v);
            *reinterpret_cast<uint16_t *>(lds + px_off<COUT>(pp, co >> 3) + 2 * (co & 7)) = h;
            if (SPLIT == 3)
              *reinterpret_cast<uint16_t *>(lds + PLANE + px_off<COUT>(pp, co >> 3) + 2 * (co & 7)) = f2bf(v - bf2f(h));
          }
        }
      }
    }
  if (MODE == 0 && WM == 1 && a.gap) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      float s = gsum[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (g == 0) a.gap[(long)p * COUT + co_w + 16 * j + r16] = s * (1.0f / NPIX);
    }
  }
  if (a.yh) {
    __syncthreads();
    constexpr int NCH = COUT / 8, TOTAL = NPAD * NCH;
    for (int pl = 0; pl < NPL; ++pl) {
      uint16_t *dst = (pl ? a.yl : a.yh) + (long)p * NPAD * COUT;
      const char *src = lds + pl * PLANE;
      for (int c = tid; c < TOTAL; c += 256) {
        const int pix = c / NCH, ch = c % NCH;
        *reinterpret_cast<uint4 *>(dst + (long)c * 8) = *reinterpret_cast<const uint4 *>(src + px_off<COUT>(pix, ch));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient.  grid = (3 tap rows, NG patch slices); 512 threads = 8 waves, wave w owns output
// channels [w*COUT/8, +COUT/8) x all CIN x the 3 taps of its row.
struct WgradArgs {
  const uint16_t *dyh, *dyl;  // [P][144][COUT] masked output gradient planes (zero halo)
  const uint16_t *xh, *xl;    // [P][144][CIN] layer input planes (zero halo)
  float *dw;                  // [COUT][CIN][3][3] fp32, accumulated with atomics (pre-zeroed)
  float *db;                  // [COUT] fp32, accumulated with atomics (pre-zeroed)
  int P, patches_per_block;
};

__device__ inline s4v tr_read(uint32_t lds_addr) {
  s4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}

// transposed fragment: 8 consecutive "rows" (pixels, the reduction index) x 16 channels starting at c0,
// from a plane [pixel][C] swizzled with px_off<C>; rowpix(q) gives the pixel of fragment row 8g+q
template <int C, typename F>
__device__ inline bf8 tr_frag(uint32_t plane, int c0, int lane, F rowpix) {
  const int t = lane & 15, q = t >> 2, pq = t & 3;
  const int chunk = (c0 >> 3) + (pq >> 1);
  const int p_lo = rowpix(q), p_hi = rowpix(q + 4);
  const s4v lo = tr_read(plane + px_off<C>(p_lo, chunk) + 8 * (pq & 1));
  const s4v hi = tr_read(plane + px_off<C>(p_hi, chunk) + 8 * (pq & 1));
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8, v);
}

template <int SPLIT, int CIN, int COUT>
__global__ __launch_bounds__(512) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int NPL = (SPLIT == 3) ? 2 : 1;
  constexpr int XPL = NPAD * CIN * 2, YPL = NPAD * COUT * 2;  // bytes per plane
  constexpr int MTW = COUT / 128 > 0 ? COUT / 128 : 1;        // 16-wide co tiles per wave (8 waves)
  constexpr int COW = COUT / 8;                               // co per wave (16 or 8 -> see below)
  static_assert(COUT % 128 == 0 || COUT == 64, "COUT");
  constexpr int NT = CIN / 16;                                // ci tiles
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *xs = lds, *ys = lds + NPL * XPL;

  const int dy = blockIdx.x;  // tap row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  // COUT = 64: only 4 co tiles exist; waves 4..7 take the same tiles as 0..3 but the other half of the ci tiles
  constexpr bool SPLIT_N = (COUT == 64);
  const int co0 = SPLIT_N ? (wave & 3) * 16 : wave * COW;
  constexpr int NTW = SPLIT_N ? NT / 2 : NT;  // ci tiles per wave
  const int nt0 = SPLIT_N ? (wave >> 2) * NTW : 0;

  f32x4 acc[3][MTW][NTW];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbsum = 0.f;

  const int p_begin = blockIdx.y * a.patches_per_block;
  const int p_end = min(a.P, p_begin + a.patches_per_block);
  const uint32_t xs_a = (uint32_t)(uintptr_t)(lds_cp)xs, ys_a = (uint32_t)(uintptr_t)(lds_cp)ys;

  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();  // previous patch fully consumed
    {
      constexpr int NCX = CIN / 8, NCY = COUT / 8;
      for (int pl = 0; pl < NPL; ++pl) {
        const uint16_t *sx = (pl ? a.xl : a.xh) + (long)p * NPAD * CIN;
        for (int c = tid; c < NPAD * NCX; c += 512)
          *reinterpret_cast<uint4 *>(xs + pl * XPL + px_off<CIN>(c / NCX, c % NCX)) =
              *reinterpret_cast<const uint4 *>(sx + (long)c * 8);
        const uint16_t *sy = (pl ? a.dyl : a.dyh) + (long)p * NPAD * COUT;
        for (int c = tid; c < NPAD * NCY; c += 512)
          *reinterpret_cast<uint4 *>(ys + pl * YPL + px_off<COUT>(c / NCY, c % NCY)) =
              *reinterpret_cast<const uint4 *>(sy + (long)c * 8);
      }
    }
    __syncthreads();
    // bias gradient: tap row 0 only; thread t < COUT sums column t over the 144 pixels (halo is zero)
    if (dy == 0 && tid < COUT) {
      float s = 0.f;
      for (int pp = 0; pp < NPAD; ++pp) {
        const int o = px_off<COUT>(pp, tid >> 3) + 2 * (tid & 7);
        s += bf2f(*reinterpret_cast<const uint16_t *>(ys + o));
        if (SPLIT == 3) s += bf2f(*reinterpret_cast<const uint16_t *>(ys + YPL + o));
      }
      dbsum += s;
    }
    // reduction over the padded pixel grid: k = padded pixel index of dY (halo rows are zero, so
    // they add nothing); the matching X pixel is k + (dy-1)*12 + (dx-1), clamped into the plane
    // (a clamped row only ever meets a zero dY row).
    constexpr int KSTEPS = (NPAD + 31) / 32;  // 5 (160 rows, rows >= 144 clamp onto zero halo rows)
#pragma unroll 1
    for (int ks = 0; ks < KSTEPS; ++ks) {
      auto ypix = [&](int q) { const int k = 32 * ks + 8 * g + q; return k < NPAD ? k : 0; };  // pixel 0 is halo (zero)
      bf8 ah[MTW], al[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        ah[i] = tr_frag<COUT>(ys_a, co0 + 16 * i, lane, ypix);
        if (SPLIT == 3) al[i] = tr_frag<COUT>(ys_a + YPL, co0 + 16 * i, lane, ypix);
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int sh = (dy - 1) * PAD_W + (dx - 1);
        auto xpix = [&](int q) {
          int k = 32 * ks + 8 * g + q;
          if (k >= NPAD) k = 0;
          const int s = k + sh;
          return s < 0 ? 0 : (s >= NPAD ? NPAD - 1 : s);
        };
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const bf8 bh = tr_frag<CIN>(xs_a, 16 * (nt0 + j), lane, xpix);
          bf8 bl;
          if (SPLIT == 3) bl = tr_frag<CIN>(xs_a + XPL, 16 * (nt0 + j), lane, xpix);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MTW; ++i) {
            if (SPLIT == 3) {
              acc[dx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[dx][i][j], 0, 0, 0);
              acc[dx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[dx][i][j], 0, 0, 0);
            }
            acc[dx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[dx][i][j], 0, 0, 0);
          }
        }
      }
    }
  }

  // partial sums -> global gradient.  acc[dx][i][j][r] = dW[co0 + 16 i + 4 g + r][ci 16 (nt0+j) + lane&15][dy][dx]
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + 16 * i + 4 * g + r, ci = 16 * (nt0 + j) + (lane & 15);
          atomicAdd(a.dw + (((long)co * CIN + ci) * 3 + dy) * 3 + dx, acc[dx][i][j][r]);
        }
  if (dy == 0 && tid < COUT) atomicAdd(a.db + tid, dbsum);
}

// ---- small helpers --------------------------------------------------------------------------------
// fp32 W[co][ci][3][3] -> forward planes [tap][co][ci] and backward-data planes [8-tap][ci][co]
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ w, int CO, int CI,
                                                           uint16_t *fh, uint16_t *fl, uint16_t *bh, uint16_t *bl) {
  const long n = (long)CO * CI * 9;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int tap = e % 9, ci = (e / 9) % CI, co = e / (9L * CI);
    const float v = w[e];
    const uint16_t h = f2bf(v), l = f2bf(v - bf2f(h));
    const long of = ((long)tap * CO + co) * CI + ci, ob = ((long)(8 - tap) * CI + ci) * CO + co;
    fh[of] = h; bh[ob] = h;
    if (fl) { fl[of] = l; bl[ob] = l; }
  }
}

// fp32 NCHW [P][C][10][10] -> planes [P][144][C] with zero halo
__global__ __launch_bounds__(256) void pack_input_kernel(const float *__restrict__ x, int P, int C, uint16_t *xh,
                                                         uint16_t *xl) {
  const long n = (long)P * NPAD * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = e % C, pp = (e / C) % NPAD;
    const long p = e / ((long)C * NPAD);
    const int yy = pp / PAD_W - 1, xx = pp % PAD_W - 1;
    float v = 0.f;
    if (yy >= 0 && yy < IMG_W && xx >= 0 && xx < IMG_W) v = x[((p * C + c) * IMG_W + yy) * IMG_W + xx];
    const uint16_t h = f2bf(v);
    xh[e] = h;
    if (xl) xl[e] = f2bf(v - bf2f(h));
  }
}

// dY[p][pp][c] = dgap[p][c] / 100 where the forward output was positive (ReLU) and pp is interior
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float *__restrict__ dgap, const uint16_t *__restrict__ yh,
                                                      int P, int C, uint16_t *dh, uint16_t *dl) {
  const long n = (long)P * NPAD * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = e % C;
    const long p = e / ((long)C * NPAD);
    float v = 0.f;
    if ((yh[e] & 0x7fff) != 0) v = dgap[p * C + c] * (1.0f / NPIX);  // halo of yh is zero
    const uint16_t h = f2bf(v);
    dh[e] = h;
    if (dl) dl[e] = f2bf(v - bf2f(h));
  }
}

template <int SPLIT, int CIN, int COUT, int MODE>
int launch_conv(const ConvArgs &a, hipStream_t s) {
  constexpr int CMAX = CIN > COUT ? CIN : COUT;
  const size_t lds = (size_t)(SPLIT == 3 ? 2 : 1) * NPAD * CMAX * 2;
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void *)conv3x3_kernel<SPLIT, CIN, COUT, MODE>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_kernel<SPLIT, CIN, COUT, MODE>), dim3(a.P), dim3(256), lds, s, a);
  return check_launch();
}

template <int SPLIT, int CIN, int COUT>
int launch_wgrad(const WgradArgs &a, hipStream_t s) {
  const size_t lds = (size_t)(SPLIT == 3 ? 2 : 1) * NPAD * (CIN + COUT) * 2;
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void *)conv3x3_wgrad_kernel<SPLIT, CIN, COUT>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  const int nblk = (a.P + a.patches_per_block - 1) / a.patches_per_block;
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<SPLIT, CIN, COUT>), dim3(3, nblk), dim3(512), lds, s, a);
  return check_launch();
}

inline int ew_grid(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace
}  // namespace crw

using namespace crw;

extern "C" {

int crw_enc_pack_weights(const float *w, int cout, int cin, uint16_t *fwd_hi, uint16_t *fwd_lo, uint16_t *bwd_hi,
                         uint16_t *bwd_lo, crw_stream_t stream) {
  clear_stale_error();
  if (!w || !fwd_hi || !bwd_hi || cout < 1 || cin < 1 || (fwd_lo == nullptr) != (bwd_lo == nullptr)) return CRW_EINVAL;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_grid((long)cout * cin * 9)), dim3(256), 0, (hipStream_t)stream, w, cout,
                     cin, fwd_hi, fwd_lo, bwd_hi, bwd_lo);
  return check_launch();
}

int crw_enc_pack_input(const float *x, int P, int C, uint16_t *xh, uint16_t *xl, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !xh || P < 1 || C < 8 || C % 8) return CRW_EINVAL;
  hipLaunchKernelGGL(pack_input_kernel, dim3(ew_grid((long)P * NPAD * C)), dim3(256), 0, (hipStream_t)stream, x, P, C,
                     xh, xl);
  return check_launch();
}

int crw_enc_gap_bwd(const float *dgap, const uint16_t *y_hi, int P, int C, uint16_t *dy_hi, uint16_t *dy_lo,
                    crw_stream_t stream) {
  clear_stale_error();
  if (!dgap || !y_hi || !dy_hi || P < 1 || C < 1) return CRW_EINVAL;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_grid((long)P * NPAD * C)), dim3(256), 0, (hipStream_t)stream, dgap, y_hi, P,
                     C, dy_hi, dy_lo);
  return check_launch();
}

int crw_enc_conv3x3(int mode, int split, int P, int cin, int cout, const uint16_t *x_hi, const uint16_t *x_lo,
                    const uint16_t *w_hi, const uint16_t *w_lo, const float *bias, const uint16_t *mask_hi,
                    uint16_t *y_hi, uint16_t *y_lo, float *y_f32, float *gap, crw_stream_t stream) {
  clear_stale_error();
  if (!x_hi || !w_hi || P < 1 || (mode != 0 && mode != 1) || (split != 1 && split != 3)) return CRW_EINVAL;
  if (split == 3 && (!x_lo || !w_lo || (y_hi && !y_lo))) return CRW_EINVAL;
  if (!y_hi && !y_f32 && !gap) return CRW_EINVAL;
  ConvArgs a{x_hi, x_lo, w_hi, w_lo, bias, mask_hi, y_hi, y_lo, y_f32, gap, P};
  hipStream_t s = (hipStream_t)stream;
#define CRW_CONV_CASE(CI, CO)                                                                      \
  if (cin == CI && cout == CO) {                                                                   \
    if (split == 3) return mode == 0 ? launch_conv<3, CI, CO, 0>(a, s) : launch_conv<3, CI, CO, 1>(a, s); \
    return mode == 0 ? launch_conv<1, CI, CO, 0>(a, s) : launch_conv<1, CI, CO, 1>(a, s);           \
  }
  CRW_CONV_CASE(32, 64)
  CRW_CONV_CASE(64, 128)
  CRW_CONV_CASE(128, 128)
  CRW_CONV_CASE(128, 64)
  CRW_CONV_CASE(64, 32)
#undef CRW_CONV_CASE
  return CRW_EINVAL;
}

int crw_enc_conv3x3_wgrad(int split, int P, int cin, int cout, const uint16_t *dy_hi, const uint16_t *dy_lo,
                          const uint16_t *x_hi, const uint16_t *x_lo, float *dw, float *db, crw_stream_t stream) {
  clear_stale_error();
  if (!dy_hi || !x_hi || !dw || !db || P < 1 || (split != 1 && split != 3)) return CRW_EINVAL;
  if (split == 3 && (!dy_lo || !x_lo)) return CRW_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dw, 0, sizeof(float) * cout * cin * 9, s) != hipSuccess) return CRW_EHIP;
  if (hipMemsetAsync(db, 0, sizeof(float) * cout, s) != hipSuccess) return CRW_EHIP;
  // ~2 blocks per tap row per CU worth of slices
  int ppb = (P + 511) / 512;
  if (ppb < 1) ppb = 1;
  WgradArgs a{dy_hi, dy_lo, x_hi, x_lo, dw, db, P, ppb};
#define CRW_WG_CASE(CI, CO)                                                                  \
  if (cin == CI && cout == CO) return split == 3 ? launch_wgrad<3, CI, CO>(a, s) : launch_wgrad<1, CI, CO>(a, s);
  CRW_WG_CASE(32, 64)
  CRW_WG_CASE(64, 128)
  CRW_WG_CASE(128, 128)
#undef CRW_WG_CASE
  return CRW_EINVAL;
}

}  // extern "C"
