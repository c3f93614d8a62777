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
// Layout: activations live in HBM as compact channels-last bf16 planes [P][100][C] (hi and lo).
// In LDS a patch is a 12x12 image with a one-pixel zero halo (zeroed once, never loaded), so a tap
// is a constant pixel offset and no border predication exists.  One workgroup = one patch: the patch
// (all channels, both planes) is loaded into LDS once and serves all 9 taps x all output channels; pixel
// rows are padded (stride 2C+32 bytes) so the lane groups of a fragment read hit distinct banks.  Waves
// split the output channels; weight fragments (pre-packed in MFMA fragment order, 1 KiB contiguous per
// wave-instruction) stream straight from L2 into rotating register sets 4 k-steps ahead, activation
// fragments are read 3 row tiles ahead of their MFMAs.  The epilogue stages the output tile through the
// same LDS and writes whole 16-byte chunks.
//
// The same kernel is the backward-data pass: dX = conv(dY, W flipped, ci <-> co), with the ReLU
// mask of the layer below applied in the epilogue instead of bias + ReLU; and, as the MAP variant, the
// forward pass on feature maps of any size (10x10 output tiles, window gathered from the map).
//
// conv3x3_wgrad_kernel: dW[co,ci,tap] = sum_{p,pix} dY[p,pix,co] * X[p,pix+tap,ci]: both operands
// are pixel-major in memory, i.e. strided along the reduction dimension, so fragments come from
// LDS through the hardware-transposing ds_read_b64_tr_b16.  One workgroup owns all 9 taps x 64
// output channels x a group of (up to) 64 input channels, walks a slice of the patches accumulating in
// registers (k-steps of 32 pixels streamed across patch boundaries), and writes its partial sums to a
// workspace; a second kernel adds the slices in a fixed order (bitwise reproducible, no float atomics).
#include "crw_common.h"
#include <type_traits>
#include <cstdlib>

namespace crw {
namespace {

constexpr int IMG_W = 10, PAD_W = 12, NPIX = 100, NPAD = 144, MT = 7;  // 7 row tiles of 16 = 112 >= 100

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char *lds_cp;

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// LDS planes: pixel rows are padded by one 16-byte chunk (stride 2C + 16 bytes), so that the 16
// pixels of an MFMA row tile -- or the 8 pixel rows of a transposed read -- start in different
// banks, while every chunk offset stays a compile-time constant (an XOR swizzle costs VALU work
// per fragment read; with padding the reads of one k-step share one address register).
template <int C>
constexpr int row_stride() { return C * 2 + 16; }
template <int C>
constexpr int plane_bytes() { return NPAD * row_stride<C>(); }
template <int C>
__device__ inline int px_off(int pp, int chunk) { return pp * row_stride<C>() + 16 * chunk; }
// The forward / backward-data kernel reads its activation fragments with ds_read_b128, whose lane groups
// ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table) collide 2-way on a 2C+16 stride and are
// conflict-free on 2C+32 for sixteen pixels whose padded positions are distinct mod 8 within each 8-lane set -- which
// consecutive OUTPUT pixels are not (the position jumps at every image-row end): see slot_pixel.  To keep two workgroups per CU at
// C = 128, the lo plane starts 132 rows after the hi plane: its top halo row overlays the hi plane's
// bottom halo row -- both are zeros.
template <int C>
constexpr int crs() { return C * 2 + 32; }
constexpr int PLANE_ROWS = NPAD - PAD_W;  // 132


// global plane [144][C] (one patch) -> LDS plane.  load() only issues the global loads, store()
// writes LDS: callers issue the loads of ALL planes first, so one HBM round trip covers them all
// (load-wait-store per plane, or per 16 bytes, exposes one round trip each).
__device__ inline int interior_pp(int i) { return (i / IMG_W + 1) * PAD_W + (i % IMG_W + 1); }

template <int C, int NTHREADS, int RS = row_stride<C>()>
struct PlaneLoad {
  static constexpr int NCH = C / 8, TOTAL = NPIX * NCH, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
  uint4 v[ITER];
  // Loads are unconditional with a clamped index (a predicated load makes the compiler branch around it,
  // wait for it at once and park the array in scratch memory); only the LDS stores are predicated.
  __device__ inline void load(const uint16_t *__restrict__ src, int tid) {  // src: [100][C] of one patch
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int c = min(tid + i * NTHREADS, TOTAL - 1);
      v[i] = *reinterpret_cast<const uint4 *>(src + (long)c * 8);
    }
  }
  __device__ inline void store(char *dst, int tid) const {  // into the interior of the padded LDS image
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int c = tid + i * NTHREADS;
      if (TOTAL % NTHREADS == 0 || c < TOTAL)
        *reinterpret_cast<uint4 *>(dst + interior_pp(c / NCH) * RS + 16 * (c % NCH)) = v[i];
    }
  }
};

// Pieces of the fused ReLU5 + global-average-pool backward (dY = dgap / 100 where the forward activation is
// non-zero).  A thread always handles the same 8 channels, so it splits dgap/100 into packed hi/lo bf16 pairs
// once per patch (gap_split8) and each pixel chunk only masks the pairs with "activation != 0" (gap_mask8).
// Contraction is off: these loaders and gap_bwd_kernel must round alike (no a*b - c fused in one of them).
__device__ inline void gap_split8(const float *__restrict__ dg8, uint32_t (&gh)[4], uint32_t (&gl)[4]) {
#pragma clang fp contract(off)
  const float4 g0 = *reinterpret_cast<const float4 *>(dg8);
  const float4 g1 = *reinterpret_cast<const float4 *>(dg8 + 4);
  const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float a0 = gv[2 * w] * (1.0f / NPIX), a1 = gv[2 * w + 1] * (1.0f / NPIX);
    const uint16_t h0 = f2bf(a0), h1 = f2bf(a1);
    gh[w] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    gl[w] = (uint32_t)f2bf(a0 - bf2f(h0)) | ((uint32_t)f2bf(a1 - bf2f(h1)) << 16);
  }
}
__device__ inline void gap_mask8(const uint4 &y, const uint32_t (&gh)[4], const uint32_t (&gl)[4], uint4 &oh, uint4 &ol) {
  const uint32_t yw[4] = {y.x, y.y, y.z, y.w};
  uint32_t h[4], l[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t m = ((yw[w] & 0x7fffu) ? 0xffffu : 0u) | ((yw[w] & 0x7fff0000u) ? 0xffff0000u : 0u);
    h[w] = gh[w] & m;
    l[w] = gl[w] & m;
  }
  oh = uint4{h[0], h[1], h[2], h[3]};
  ol = uint4{l[0], l[1], l[2], l[3]};
}

// Fused ReLU5 + global-average-pool backward: the gradient planes of the last conv layer are never
// materialised; dY[i][c] = dgap[c] / 100 where the forward activation y[i][c] (hi plane) is non-zero.
// Loads the activation chunks, builds the hi/lo planes of dY directly in LDS.
template <int C, int NTHREADS, int SPLIT, int RS = row_stride<C>()>
__device__ inline void gap_planes_to_lds(const uint16_t *__restrict__ yh, const float *__restrict__ dgap_row,
                                         char *dst_hi, char *dst_lo, int tid) {
  constexpr int NCH = C / 8, TOTAL = NPIX * NCH, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
  uint4 v[ITER];
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = min(tid + i * NTHREADS, TOTAL - 1);  // clamped, unconditional (see PlaneLoad)
    v[i] = *reinterpret_cast<const uint4 *>(yh + (long)c * 8);
  }
  static_assert(NTHREADS % NCH == 0, "a thread keeps its channel chunk");
  const int ch = tid % NCH;
  uint32_t gh[4], gl[4];
  gap_split8(dgap_row + 8 * ch, gh, gl);
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = tid + i * NTHREADS;
    if (TOTAL % NTHREADS == 0 || c < TOTAL) {
      uint4 oh, ol;
      gap_mask8(v[i], gh, gl, oh, ol);
      const int off = interior_pp(c / NCH) * RS + 16 * ch;
      *reinterpret_cast<uint4 *>(dst_hi + off) = oh;
      if (SPLIT == 3) *reinterpret_cast<uint4 *>(dst_lo + off) = ol;
    }
  }
}

// zero the 44 halo pixels of an LDS plane
template <int C, int NTHREADS, int RS = row_stride<C>()>
__device__ inline void zero_halo(char *plane, int tid) {
  constexpr int NCH = C / 8;
  for (int c = tid; c < 44 * NCH; c += NTHREADS) {
    const int h = c / NCH, ch = c % NCH;
    // halo pixels in order: top row (12), bottom row (12), then left/right of the 10 middle rows
    const int pp = h < 12 ? h : (h < 24 ? 11 * PAD_W + (h - 12) : (1 + (h - 24) / 2) * PAD_W + ((h - 24) & 1) * 11);
    *reinterpret_cast<uint4 *>(plane + pp * RS + 16 * ch) = uint4{0, 0, 0, 0};
  }
}

// Which output pixel an MFMA row computes.  A ds_read_b128 is served in four groups of 16 lanes, and with the 2C+32 row
// stride a group is conflict-free exactly when the source pixels of its two 8-lane sets -- tile rows {0-3, 12-15} and
// {4-11} -- have padded positions (12 y + x) that are distinct mod 8.  Sixteen CONSECUTIVE output pixels break that at every
// image-row end (the position jumps by 3): PMC showed 0.9 extra LDS cycles per cycle of fragment reads, on a kernel whose
// LDS array is busy 96 % of the time.  So the 7 x 16 rows are dealt out by residue instead:
//   tiles 0-4: rows 4-11 = image row 2t, x = 0..7; rows 0-3, 12-15 = image row 2t + 1, x = 0..7 (every residue once);
//   tiles 5-6: the pixels x = 8, 9 (residues {0,1} on even image rows, {4,5} on odd ones) in four sets: three hold one even
//   + one odd row (4 pixels; the other 4 lanes repeat them: same address = broadcast) and the last one holds image rows 6-9
//   (a 2-way conflict: the residues 0, 1, 4, 5 occur 15 times each in 14 sets).
// row = row inside the tile (0..15).  Returns false for a lane that only repeats another lane's pixel.
__device__ inline bool slot_pixel(int tile, int row, int &y, int &x) {
  const bool s2 = row >= 4 && row < 12;
  const int q = s2 ? row - 4 : (row < 4 ? row : row - 8);
  if (tile < 5) {
    y = 2 * tile + (s2 ? 0 : 1);
    x = q;
    return true;
  }
  const int L = 2 * (tile - 5) + (s2 ? 1 : 0);
  y = 2 * L + ((q >> 1) & 1) + (L == 3 ? 2 * (q >> 2) : 0);
  x = 8 + (q & 1);
  return q < 4 || L == 3;
}

template <int N, class F>
__device__ inline void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

struct ConvArgs {
  const uint16_t *xh, *xl;   // [P][100][CIN] input planes
  const uint16_t *wh, *wl;   // weights in fragment order (pack_weights_kernel; flipped/transposed for backward-data)
  const float *bias;         // [COUT] (MODE 0) or null
  const uint16_t *maskh;     // [P][100][COUT]: output is zeroed where this plane is 0 (MODE 1) or null
  uint16_t *yh, *yl;         // [P][100][COUT] output planes or null
  float *yf;                 // optional fp32 output [P][100][COUT]
  float *gap;                // optional [P][COUT]: mean over the 100 pixels (MODE 0)
  const float *dgap;         // MODE 1, optional [P][CIN]: the input gradient is dgap/100 gated by xh (= forward
                             // activation hi plane) instead of being read from xh/xl
  int P;
  long long *stamps;         // diagnostic builds only: [grid][5] s_memtime at phase boundaries (else null)
  // MAP variant only: planes are feature maps [P][mh][mw][C] of any size, a workgroup computes one 10x10 output tile
  int mh, mw, tiles_x, tiles_y;
  // ... the tiles of this launch: tile_list[0..ncls) (a launch takes up to 64 tiles of one class: full ones, or the small edge
  // tiles; a map with more tiles of a class than that goes out in several launches)
  int ncls;
  unsigned short tile_list[64];
};

// One workgroup = one patch, NW waves (4 or 8).  With 8 waves a wave owns one 16-channel output tile
// (COUT = 128), stays under 128 registers and two workgroups put 4 waves on every SIMD: a workgroup's own
// load / epilogue phases are then covered by the other workgroup's MFMAs, and twice as many waves keep
// weight-fragment loads in flight (4-wave workgroups at 2 waves per SIMD measured ~65 % matrix-pipe
// occupancy: each workgroup's phases are latency-bound on their own).
// MAP = true: the same kernel on feature maps of any size (other patch sizes than 16x16, e.g. the 26x26 maps of
// 32x32 patches): workgroup = (patch, 10x10 output tile); the 12x12 input window is gathered from the map with
// zeros outside it, only in-map output pixels are stored, `gap` receives per-tile SUMS (forward only).
// waves per SIMD the register allocation aims at: 4 (two 8-wave workgroups per CU, what the LDS of the 128-channel layers allows);
// 6 for the 8-wave kernels of the 32 <-> 64 channel layers: their 44 KB of LDS admit three workgroups, and with the shorter
// look-aheads below they fit 80 registers without spills (conv3 forward 239 -> 219 us inside the step: these layers spend under
// 40 % of a workgroup's life in the k-loop, so a third workgroup per CU buys more than the look-aheads cost)
template <int CIN, int COUT, int NW, bool MAP>
constexpr int conv_waves_per_simd() {
  return (!MAP && NW == 8 && CIN <= 64 && COUT <= 64) ? 6 : NW / 2;
}
// MTL (MAP only): row tiles of 16 that a tile of this launch needs.  Edge tiles of a map hold tw x th < 100 in-map pixels
// (26 x 26 maps: 10 x 6, 6 x 10, 6 x 6); those with at most 64 go to an MTL = 4 launch whose MFMA rows are dealt to the in-map
// pixels only (row i -> pixel (i / tw, i % tw)): 28 + 20 = 48 row tiles per 26 x 26 map instead of 63 (r02 cfg5: conv5 at 0.134 of
// the roof against 0.20 on 10 x 10 maps).  MTL = MT keeps the 10-wide row order (pixels outside the map are computed and dropped).
template <int SPLIT, int CIN, int COUT, int MODE, int NW, bool MAP = false, int MTL = MT>
__global__ __launch_bounds__(NW * 64, (conv_waves_per_simd<CIN, COUT, NW, MAP>())) void conv3x3_kernel(ConvArgs a) {
  static_assert(MAP || MTL == MT, "MTL is a MAP parameter");
  constexpr bool REMAP = MTL < MT;
  constexpr int NT = NW * 64;
  constexpr int NPL = (SPLIT == 3) ? 2 : 1;
  constexpr int CMAX = CIN > COUT ? CIN : COUT;
  // LDS row strides of the input image / the output staging and byte offset of the lo plane.  MAP windows carry real
  // neighbour data in their halo rows, so the two planes cannot share a halo row: full 144-row planes on the
  // 2C+16 stride (2-way conflicts, still two workgroups per CU)
  constexpr int RSI = MAP ? row_stride<CIN>() : crs<CIN>(), RSO = MAP ? row_stride<COUT>() : crs<COUT>();
  constexpr int PLANE = MAP ? NPAD * row_stride<CMAX>() : PLANE_ROWS * crs<CMAX>();
  constexpr int WN = (COUT / 16 >= NW) ? NW : COUT / 16;  // waves across the output channels
  constexpr int WM = NW / WN;                             // waves across the pixel row tiles
  constexpr int NTW = COUT / 16 / WN;                     // 16-wide column tiles per wave
  constexpr int MTW = (MTL + WM - 1) / WM;                // row tiles per wave and patch (tile wm + WM*k)
  constexpr int KCH = CIN / 32;                           // 32-deep k-steps per tap
  // rows dealt out by bank residue (slot_pixel) where a wave owns all 7 row tiles; with row tiles split over waves
  // (64 / 32 output channels at 8 waves) the wave-uniform tile index costs the short epilogue more than the reads gain
  // (measured, tools/r02_conv_ab.sh: dealing by residue with split row tiles gains 2-3 % on conv4's backward-data -- 36 k-steps
  // per patch -- and nothing on the 9 / 18-step layers)
  constexpr bool PERM = !MAP && (WM == 1 || (MODE == 1 && CIN == 128 && COUT == 64));
  extern __shared__ __attribute__((aligned(16))) char lds[];

  int p0 = blockIdx.x, ty0 = 0, tx0 = 0;  // patch; MAP: origin of this workgroup's output tile in the map
  int tw = IMG_W, npix_live = NPIX, tw_recip = 0, gap_slot = 0;
  if constexpr (MAP) {
    const int tile = a.tile_list[blockIdx.x % a.ncls];
    p0 = blockIdx.x / a.ncls;
    ty0 = (tile / a.tiles_x) * IMG_W;
    tx0 = (tile % a.tiles_x) * IMG_W;
    gap_slot = p0 * (a.tiles_x * a.tiles_y) + tile;
    if constexpr (REMAP) {
      tw = min(IMG_W, a.mw - tx0);
      npix_live = tw * min(IMG_W, a.mh - ty0);  // <= 16 MTL by the launcher's choice of class
      tw_recip = 65536 / tw + 1;                // (i * tw_recip) >> 16 == i / tw for i < 112, tw <= 10
    }
  }
  (void)tw; (void)npix_live; (void)tw_recip; (void)gap_slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;

  auto stamp = [&](int k) {  // phase timestamps, only in `make STAMPS=1` builds (tools/stamp_probe.py)
#ifdef CRW_CONV_STAMPS
    if (a.stamps && tid == 0) a.stamps[(long)blockIdx.x * 5 + k] = (long long)__builtin_amdgcn_s_memtime();
#else
    (void)k;
#endif
  };
  stamp(0);
  // ---- patch -> LDS ------------------------------------------------------------------------------
  if constexpr (MAP) {
    // all 144 pixels of the 12x12 window (halo included) come from the map; outside it they are zero
    constexpr int NCH = CIN / 8, WTOT = NPAD * NCH, WIT = (WTOT + NT - 1) / NT;
    uint4 vh[WIT], vl[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = min(tid + i * NT, WTOT - 1), wp = c / NCH, ch = c % NCH;
      const int my = ty0 + wp / PAD_W - 1, mx = tx0 + wp % PAD_W - 1;
      const bool ok = my >= 0 && my < a.mh && mx >= 0 && mx < a.mw;
      const long off = (((long)p0 * a.mh + min(max(my, 0), a.mh - 1)) * a.mw + min(max(mx, 0), a.mw - 1)) * CIN + 8 * ch;
      vh[i] = *reinterpret_cast<const uint4 *>(a.xh + off);  // unconditional, clamped (see PlaneLoad)
      if (SPLIT == 3) vl[i] = *reinterpret_cast<const uint4 *>(a.xl + off);
      if (!ok) vh[i] = vl[i] = uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int c = tid + i * NT;
      if (WTOT % NT == 0 || c < WTOT) {
        *reinterpret_cast<uint4 *>(lds + (c / NCH) * RSI + 16 * (c % NCH)) = vh[i];
        if (SPLIT == 3) *reinterpret_cast<uint4 *>(lds + PLANE + (c / NCH) * RSI + 16 * (c % NCH)) = vl[i];
      }
    }
  } else {
    const int p = p0;
    char *img = lds;
    zero_halo<CIN, NT, RSI>(img, tid);
    if (SPLIT == 3) zero_halo<CIN, NT, RSI>(img + PLANE, tid);
    if (MODE == 1 && a.dgap) {
      gap_planes_to_lds<CIN, NT, SPLIT, RSI>(a.xh + (long)p * NPIX * CIN, a.dgap + (long)p * CIN, img, img + PLANE, tid);
    } else {
      PlaneLoad<CIN, NT, RSI> lh, ll;
      lh.load(a.xh + (long)p * NPIX * CIN, tid);
      if (SPLIT == 3) ll.load(a.xl + (long)p * NPIX * CIN, tid);
      lh.store(img, tid);
      if (SPLIT == 3) ll.store(img + PLANE, tid);
    }
  }
  __syncthreads();
  stamp(1);

  // row tile mt, row r16 -> interior pixel i -> LDS byte offset of this lane's 16-byte chunk of the tap-(0,0)
  // source pixel (k-chunk g); a k-step adds a wave-uniform offset (tap shift + 64 bytes per 32 channels)
  const int wm = wave / WN, wn = wave % WN;
  int abase[MTW];
#pragma unroll
  for (int k = 0; k < MTW; ++k) {
    if constexpr (REMAP) {
      int i = 16 * (wm + WM * k) + r16;
      if (i >= npix_live) i = 0;  // dummy rows recompute pixel 0 and are dropped in the epilogue
      const int y = (i * tw_recip) >> 16;
      abase[k] = (y * PAD_W + (i - y * tw)) * RSI + 16 * g;
    } else if constexpr (!PERM) {
      int i = 16 * (wm + WM * k) + r16;
      if (i >= NPIX) i = 0;  // dummy rows recompute pixel 0 and are dropped in the epilogue
      abase[k] = ((i / IMG_W) * PAD_W + (i % IMG_W)) * RSI + 16 * g;
    } else {
      int y, x;
      const int tile = wm + WM * k;
      slot_pixel(tile < MT ? tile : 0, r16, y, x);
      abase[k] = (y * PAD_W + x) * RSI + 16 * g;
    }
  }
  auto tile_live = [&](int k) { return WM == 1 || wm + WM * k < MTL; };  // wave-uniform
  auto step_off = [&](int step) {
    const int tap = step / KCH, cc = step % KCH;
    return ((tap / 3) * PAD_W + (tap % 3)) * RSI + 64 * cc;
  };
  auto read_a = [&](int soff, int k, bf8 &h, bf8 &l) {
    const char *p = lds + abase[k] + soff;
    h = *reinterpret_cast<const bf8 *>(p);
    if (SPLIT == 3) l = *reinterpret_cast<const bf8 *>(p + PLANE);
  };

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
      // weights are packed in fragment order: one wave-instruction reads 1 KiB contiguous (8 full
      // cache lines) instead of 16 half lines of a [co][ci] row layout
      const long o = ((((long)tap * KCH + cc) * (COUT / 16) + (co_w / 16 + j)) * 64 + lane) * 8;
      bh[j] = *reinterpret_cast<const bf8 *>(a.wh + o);
      if (SPLIT == 3) bl[j] = *reinterpret_cast<const bf8 *>(a.wl + o);
    }
  };

  constexpr int NSTEP = 9 * KCH;
  constexpr bool OCC6 = conv_waves_per_simd<CIN, COUT, NW, MAP>() == 6;  // 80 registers: shorter look-aheads
  constexpr int AHEAD = OCC6 ? (CIN == 32 ? 1 : 2) : 4;  // weight fragments are requested AHEAD k-steps before use
  // Activation fragments are read from LDS DIST row tiles before the MFMAs that consume them (the compiler,
  // left alone, issues them one tile = 96 cycles ahead: less than the LDS latency, so a workgroup alone in
  // its k-loop kept the matrix pipe only ~55 % busy).  The first DIST tiles of the NEXT k-step are read
  // during the last tiles of this one and carried in nah/nal.
  constexpr int DIST = OCC6 ? (MTW < 2 ? MTW : 2) : (MTW < 3 ? MTW : 3);
  bf8 nah[DIST], nal[DIST];
#pragma unroll
  for (int k = 0; k < DIST; ++k)
    if (tile_live(k)) read_a(step_off(0), k, nah[k], nal[k]);
  auto do_step = [&](auto LAST, int step, bf8 (&bhc)[NTW], bf8 (&blc)[NTW], bf8 (&bhn)[NTW], bf8 (&bln)[NTW]) {
    constexpr bool last = decltype(LAST)::value;
    if (step + AHEAD < NSTEP) load_b(step + AHEAD, bhn, bln);
    const int so = step_off(step), so1 = step_off(step + 1);
    bf8 ah[MTW], al[MTW];
#pragma unroll
    for (int k = 0; k < DIST; ++k) {
      ah[k] = nah[k];
      al[k] = nal[k];
    }
#pragma unroll
    for (int k = 0; k < MTW; ++k) {
      if (k + DIST < MTW) {
        if (tile_live(k + DIST)) read_a(so, k + DIST, ah[k + DIST], al[k + DIST]);
      } else if (!last) {
        if (tile_live(k + DIST - MTW)) read_a(so1, k + DIST - MTW, nah[k + DIST - MTW], nal[k + DIST - MTW]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tile_live(k)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          if (SPLIT == 3) {
            acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[k], bhc[j], acc[k][j], 0, 0, 0);
            acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k], blc[j], acc[k][j], 0, 0, 0);
          }
          acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k], bhc[j], acc[k][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // five register sets for the weight fragments, rotated with static indices (a run-time index would
  // send them to scratch): step s uses set s % 5 and refills set (s + AHEAD) % 5, i.e. AHEAD k-steps of
  // weight loads (16 KiB per wave in bf16x3 at AHEAD = 4) are in flight.  Six sets measured faster per workgroup but
  // push conv5 past 256 registers (one workgroup per CU: slower).
  bf8 bh[5][NTW], bl[5][NTW];
  using std::integral_constant;
  static_for<AHEAD>([&](auto I) { load_b(decltype(I)::value, bh[decltype(I)::value], bl[decltype(I)::value]); });
  static_assert(NSTEP >= 9 && NSTEP % 5 != 0, "NSTEP");  // the last k-step is one of the tail calls below
  constexpr int TAIL = NSTEP % 5;
  int step = 0;
  for (; step + 5 <= NSTEP; step += 5)
    static_for<5>([&](auto I) {
      constexpr int i = decltype(I)::value;
      do_step(integral_constant<bool, false>{}, step + i, bh[i], bl[i], bh[(i + AHEAD) % 5], bl[(i + AHEAD) % 5]);
    });
  static_for<TAIL>([&](auto I) {
    constexpr int i = decltype(I)::value;
    do_step(integral_constant<bool, i == TAIL - 1>{}, step + i, bh[i], bl[i], bh[(i + AHEAD) % 5], bl[(i + AHEAD) % 5]);
  });

  // ---- epilogue ----------------------------------------------------------------------------------
  // C/D map: acc[k][j][r] = out[patch p0][pixel 16 (wm + WM k) + 4 g + r][channel co_w + 16 j + r16]
  float bias_r[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) bias_r[j] = (MODE == 0 && a.bias) ? a.bias[co_w + 16 * j + r16] : 0.f;
  stamp(2);
  __syncthreads();  // every wave is done reading the input image: reuse LDS as the output staging
  stamp(3);
  {
    const int p = p0;
    char *img = lds;
    float gsum[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) gsum[j] = 0.f;
    // per-lane terms of slot_pixel for the accumulator rows 4 g + r (see there): tiles 0-4 / tiles 5-6
    const int slot_pp = g == 0 ? PAD_W : g == 1 ? 0 : g == 2 ? 4 : PAD_W + 4, slot_i = g == 0 ? IMG_W : g == 1 ? 0 : g == 2 ? 4 : IMG_W + 4;
    const int slot_pp5 = 2 * PAD_W * g, slot_i5 = 2 * IMG_W * g;
    (void)slot_pp; (void)slot_i; (void)slot_pp5; (void)slot_i5;
#pragma unroll
    for (int k = 0; k < MTW; ++k)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int co = co_w + 16 * j + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int i = 16 * (wm + WM * k) + 4 * g + r;
          bool valid = i < NPIX;  // also false for the tiles a wave does not own (wm + WM k >= 7)
          int pp = 0;
          if constexpr (REMAP) {  // row i of the tile = in-map pixel (i / tw, i % tw); canonical pixel id y * 10 + x from here on
            valid = i < npix_live && wm + WM * k < MTL;
            const int y = (i * tw_recip) >> 16;
            i = y * IMG_W + (i - y * tw);
            if (valid) pp = interior_pp(i);
          } else if constexpr (!PERM) {
            if (valid) pp = interior_pp(i);
          } else {
            // slot_pixel(tile, 4 g + r) split into a wave-uniform part (tile, r) and the per-lane terms of the lane group g
            const int tile = wm + WM * k;
            if (tile < 5) {
              i = 20 * tile + r + slot_i;
              pp = (2 * tile + 1) * PAD_W + 1 + r + slot_pp;
              valid = true;
            } else {
              i = 40 * (tile - 5) + 10 * (r >> 1) + (r & 1) + 8 + slot_i5;
              pp = 4 * PAD_W * (tile - 5) + PAD_W * (r >> 1) + (r & 1) + 21 + slot_pp5;
              valid = tile < MT && (g < 2 || (g == 2 && tile == 6));
            }
          }
          if (valid) {
            float v = acc[k][j][r];
            if (MODE == 0) {
              v = fmaxf(v + bias_r[j], 0.f);
              if (!MAP || (ty0 + i / IMG_W < a.mh && tx0 + i % IMG_W < a.mw)) gsum[j] += v;
            }
            if (a.yf) {
              if constexpr (MAP) {
                const int oy = ty0 + i / IMG_W, ox = tx0 + i % IMG_W;
                if (oy < a.mh && ox < a.mw) a.yf[(((long)p * a.mh + oy) * a.mw + ox) * COUT + co] = v;
              } else {
                a.yf[((long)p * NPIX + i) * COUT + co] = v;
              }
            }
            if (a.yh) {
              const uint16_t h = f2bf(v);
              *reinterpret_cast<uint16_t *>(img + pp * RSO + 2 * co) = h;
              if (SPLIT == 3 && a.yl) *reinterpret_cast<uint16_t *>(img + PLANE + pp * RSO + 2 * co) = f2bf(v - bf2f(h));
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
        if (g == 0) {
          if constexpr (MAP) a.gap[(long)gap_slot * COUT + co_w + 16 * j + r16] = s;  // per-tile sum; the caller reduces
          else a.gap[(long)p * COUT + co_w + 16 * j + r16] = s * (1.0f / NPIX);
        }
      }
    }
  }
  if (a.yh) {
    __syncthreads();
    // staged tiles -> global in whole 16-byte chunks; the ReLU mask of the layer below (MODE 1) is
    // applied here, 8 channels at a time, from the matching chunk of its activation plane
    constexpr int NCH = COUT / 8, TOTAL = NPIX * NCH, ITER = (TOTAL + NT - 1) / NT;
    const int p = p0;
    const uint16_t *mk = (MODE == 1 && a.maskh) ? a.maskh + (MAP ? 0 : (long)p * NPIX * COUT) : nullptr;
    uint4 mv[ITER];
    if (mk) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int c = min(tid + it * NT, TOTAL - 1);
        if constexpr (MAP) {  // the mask plane is a map too: the chunk of the (clamped) output pixel
          const int oy = min(ty0 + (c / NCH) / IMG_W, a.mh - 1), ox = min(tx0 + (c / NCH) % IMG_W, a.mw - 1);
          mv[it] = *reinterpret_cast<const uint4 *>(mk + (((long)p * a.mh + oy) * a.mw + ox) * COUT + 8 * (c % NCH));
        } else {
          mv[it] = *reinterpret_cast<const uint4 *>(mk + (long)c * 8);
        }
      }
    }
    for (int pl = 0; pl < NPL; ++pl) {
      if (pl && !a.yl) break;  // the lo plane is optional (nobody reads conv5's)
      uint16_t *dst = (pl ? a.yl : a.yh) + (MAP ? 0 : (long)p * NPIX * COUT);
      const char *src = lds + pl * PLANE;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int c = tid + it * NT;
        bool in_map = true;
        long dpix = c / NCH;  // destination pixel index (in the patch plane, or in the whole map tensor)
        if constexpr (MAP) {
          const int oy = ty0 + (c / NCH) / IMG_W, ox = tx0 + (c / NCH) % IMG_W;
          in_map = oy < a.mh && ox < a.mw;
          dpix = ((long)p * a.mh + oy) * a.mw + ox;
        }
        if (c < TOTAL && in_map) {
          uint4 v = *reinterpret_cast<const uint4 *>(src + interior_pp(c / NCH) * RSO + 16 * (c % NCH));
          if (mk) {  // keep a bf16 lane only where the mask lane is non-zero
            const uint32_t m[4] = {mv[it].x, mv[it].y, mv[it].z, mv[it].w};
            uint32_t o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const uint32_t keep = ((m[w] & 0x7fffu) ? 0xffffu : 0u) | ((m[w] & 0x7fff0000u) ? 0xffff0000u : 0u);
              o[w] &= keep;
            }
            v = uint4{o[0], o[1], o[2], o[3]};
          }
          *reinterpret_cast<uint4 *>(dst + dpix * COUT + 8 * (c % NCH)) = v;
        }
      }
    }
  }
  stamp(4);
}

// ------------------------------------------------------------------------------------------------
// weight gradient.  1-D grid over (patch slice, channel group); 4 or 8 waves per workgroup (see the kernel).
struct WgradArgs {
  const uint16_t *dyh, *dyl;  // [P][100][COUT] masked output gradient planes
  const uint16_t *xh, *xl;    // [P][100][CIN] layer input planes
  const float *dgap;          // optional [P][COUT]: dY = dgap/100 gated by dyh (= forward activation hi plane)
  float *dw_part;             // [nslice][9 taps][COUT][CIN] fp32 partial sums (every element written)
  float *db_part;             // [nslice][COUT] fp32 partial sums
  int P, patches_per_block;
  int xcd_map;                // 1: XCD-aware id -> (slice, group) mapping (needs the slice count to be a multiple of 8)
  long long *stamps;          // `make STAMPS=1`: [workgroup][64]: 4 phase sums over the slice, then k-loop start times
  // MAP variant only: planes are feature maps [P][mh][mw][C]; the unit of work is (patch, 10x10 tile), P counts units
  int mh, mw, tiles_x, tiles_y;
};

template <int IMM>
__device__ inline s4v tr_read(uint32_t lds_addr) {
  s4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(IMM) : "memory");
  return v;
}

// transposed fragment: 8 reduction rows (pixels) x the 16 channels starting at channel C0.  a_lo / a_hi
// are this lane's LDS byte addresses of its two groups of 4 fragment rows (already including the lane's
// 8*(pq&1) + 16*(pq>>1) column part); C0 and the plane offset enter as an instruction immediate.
template <int IMM>
__device__ inline bf8 tr_frag(uint32_t a_lo, uint32_t a_hi) {
  const s4v lo = tr_read<IMM>(a_lo);
  const s4v hi = tr_read<IMM>(a_hi);
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8, v);
}

// slice (1-D grid; id -> (slice, channel group) mapping below).
// One workgroup (4 waves) = all 9 taps x WG_NCO output channels x NCI input channels of one patch
// slice; blockIdx.x selects the slice, blockIdx.y the (ci group, co group).
//   * A wave owns NCOW co tiles x ONE ci tile x 9 taps (36 accumulator tiles for conv4/conv5).  The dY
//     fragments do not depend on the tap, the X fragments do, so this split needs 2 NCOW + 18 fragment
//     reads per k-step and wave (26 for bf16x3) where one co tile x four ci tiles needs 74: the kernel
//     is bound by LDS reads, not by the matrix cores.
//   * X fragments are double-buffered in registers across taps (reads of tap t+1 fly during the
//     MFMAs of tap t; counted s_waitcnt since the transposing reads are inline asm).
//   * dY lives in LDS without a halo (only X is shifted by the tap): 100 pixel rows + one zero row for
//     the k padding, which keeps a workgroup at 69 KiB -> two independent workgroups per CU, one
//     loading its next patch while the other computes.
constexpr int WG_NCO = 64, YROWS = NPIX + 1;


// k-steps stream ACROSS patch boundaries: a patch has 100 pixels = 3 k-steps of 32 + 4, and padding every patch to
// 4 k-steps wastes 22 % of the MFMAs and fragment reads.  Instead the last 4 (then 8) pixels of a patch are
// deferred and contracted together with the first 28 (24) pixels of the next one; every third patch flushes
// (4 k-steps, the last one padded): 10 k-steps per 3 patches instead of 12.  The deferred pixels lie in image
// row 9, so all a later k-step needs of the old patch is its padded X rows 9-10 (row 11 is the zero halo) and
// 8 dY rows: they are copied to a tail strip before the next patch overwrites the image.  Layout per plane:
//   X : [tail strip: 24 pixels = old padded rows 9, 10][main 12x12 image]   -- the strip sits right in front of
//       the image so that its "row 11" IS the image's (always zero) top halo row;
//   dY: [100 pixel rows][zero row][8 tail rows = old pixels 92..99].
constexpr int XTAIL = 2 * PAD_W, TAILPIX = 8, YROWS_ALL = YROWS + TAILPIX;

// MAP = true: feature maps of any size.  A unit is (patch, 10x10 output tile): its X operand is the 12x12 window around
// the tile gathered from the map (real neighbours in the halo, zeros outside the map), its dY operand the tile's pixels (zero
// outside the map); every unit is 4 k-steps on its own (no streaming across units: the halo is not zero there).
template <int SPLIT, int CIN, int COUT, int NCI, int NW, bool MAP = false>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int NCO = WG_NCO, NTH = NW * 64;
  constexpr int NPL = (SPLIT == 3) ? 2 : 1;
  constexpr int XS = row_stride<NCI>(), XPL = (XTAIL + NPAD) * XS;  // X plane: tail strip + 12x12 padded image
  constexpr int YS = row_stride<NCO>(), YPL = YROWS_ALL * YS;       // dY plane: compact rows + zero row + tail rows
  constexpr int NT = NCI / 16;                                       // ci tiles of this group
  constexpr int WCI = NT < 4 ? NT : 4, WCO = NW / WCI;              // waves along ci tiles / co tiles
  constexpr int NCOW = NCO / 16 / WCO;                              // co tiles per wave
  static_assert(NT == WCI && COUT % NCO == 0 && CIN % NCI == 0, "tiling");
  constexpr int NGRP_CI = CIN / NCI;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *xs = lds, *ys = lds + NPL * XPL;

  // 1-D grid, workgroup id L -> (slice, channel group).  Workgroup L runs on XCD L % 8 and, for the ids
  // resident at launch, on CU (L / 8) % 32 of that XCD (tools/ubench/dispatch_map.hip).  With
  //   group = (L / 8) % NGRP,  slice = (L / 8 / NGRP) * 8 + L % 8
  // the NGRP groups of a slice sit on neighbouring CUs of ONE XCD and read the slice's patches at the same
  // time: the planes come from HBM once and are served to the other groups by that XCD's L2 (with groups
  // on different XCDs PMC showed every byte fetched from HBM twice), while the two workgroups that share a
  // CU (L and L + 256) belong to different slices.
  constexpr int NGRP = NGRP_CI * (COUT / NCO);
  const int wg_slice = a.xcd_map ? ((int)blockIdx.x / 8 / NGRP) * 8 + (int)blockIdx.x % 8 : (int)blockIdx.x / NGRP;
  const int wg_grp = a.xcd_map ? ((int)blockIdx.x / 8) % NGRP : (int)blockIdx.x % NGRP;
  const int grp_ci = wg_grp % NGRP_CI, grp_co = wg_grp / NGRP_CI;
  const int ci_base = grp_ci * NCI, co_base = grp_co * NCO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int wci = wave % WCI, wco = wave / WCI;
  const int co0w = wco * NCOW * 16;  // first output channel of this wave inside the workgroup's NCO

  f32x4 acc[9][NCOW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NCOW; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient on the matrix cores: dY^T fragments x an all-ones fragment = the pixel sums of 16 output
  // channels in every column; only the waves of ci tile 0 in ci group 0 do it (8 extra MFMAs per k-step)
  f32x4 accb[NCOW];
#pragma unroll
  for (int j = 0; j < NCOW; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // scalar condition (readfirstlane): matrix instructions ignore EXEC, so a block holding them must be branched around
  const bool bias_wave = grp_ci == 0 && __builtin_amdgcn_readfirstlane(wci) == 0;

  const int p_begin = wg_slice * a.patches_per_block;
  const int p_end = min(a.P, p_begin + a.patches_per_block);
  const uint32_t xs_a = (uint32_t)(uintptr_t)(lds_cp)xs, ys_a = (uint32_t)(uintptr_t)(lds_cp)ys;

  // zeroed once: the halo of the X images, the tail strips, and the k-padding row of the dY planes
  for (int pl = 0; pl < NPL; ++pl) {
    zero_halo<NCI, NTH, XS>(xs + pl * XPL + XTAIL * XS, tid);
    for (int c = tid; c < XTAIL * XS / 16; c += NTH) *reinterpret_cast<uint4 *>(xs + pl * XPL + 16 * c) = uint4{0, 0, 0, 0};
    for (int c = tid; c < (1 + TAILPIX) * YS / 16; c += NTH) *reinterpret_cast<uint4 *>(ys + pl * YPL + NPIX * YS + 16 * c) = uint4{0, 0, 0, 0};
  }

  constexpr int XCH = NCI / 8, XTOT = NPIX * XCH, XIT = (XTOT + NTH - 1) / NTH;
  constexpr int YCH = NCO / 8, YTOT = NPIX * YCH, YIT = (YTOT + NTH - 1) / NTH;
  // tail copy: per plane XTAIL pixels x XCH chunks of X and TAILPIX rows x YCH chunks of dY
  constexpr int TXC = XTAIL * XCH, TYC = TAILPIX * YCH, TTOT = NPL * (TXC + TYC), TIT = (TTOT + NTH - 1) / NTH;
#ifdef CRW_CONV_STAMPS
  long long ph_[4] = {0, 0, 0, 0}, prev_ = (long long)__builtin_amdgcn_s_memtime();
#define WG_STAMP(k) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); ph_[k] += n_ - prev_; prev_ = n_; }
#else
#define WG_STAMP(k)
#endif

  // LDS byte offsets (within a plane) of tail-copy chunk c: source in the image / dY rows, destination in the tails
  auto tail_src_dst = [&](int c, int &src, int &dst) {
    const int pl = c / (TXC + TYC), r = c % (TXC + TYC);
    if (r < TXC) {  // X: strip pixel s <- image pixel 9*12 + s
      const int s = r / XCH, ch = r % XCH;
      dst = pl * XPL + s * XS + 16 * ch;
      src = pl * XPL + (XTAIL + 9 * PAD_W + s) * XS + 16 * ch;
    } else {        // dY: tail row j <- pixel row 92 + j   (offsets relative to xs: ys = xs + NPL * XPL)
      const int q = r - TXC, j = q / YCH, ch = q % YCH;
      dst = NPL * XPL + pl * YPL + (YROWS + j) * YS + 16 * ch;
      src = NPL * XPL + pl * YPL + (NPIX - TAILPIX + j) * YS + 16 * ch;
    }
  };

  // one k-step of 32 virtual rows: row v < carry -> deferred pixel 100 - carry + v of the previous patch (tail
  // strips), else pixel v - carry of the current patch if it exists (and `flush` is false), else the zero row
  const int t16 = lane & 15, q4 = t16 >> 2, pq = t16 & 3;
  const uint32_t lane_col = 8 * (pq & 1) + 16 * (pq >> 1);
  auto row_addr = [&](int v, int carry, bool flush, uint32_t &ya, uint32_t &xa) {
    const int i = v - carry;
    int yrow = NPIX, xpix = XTAIL;  // zero dY row; any valid X pixel
    if (v < carry) {
      const int t = NPIX - carry + v;  // 92..99: image row 9, column t - 90
      yrow = YROWS + (t - (NPIX - TAILPIX));
      xpix = t - 90;
    } else if (!flush && i < NPIX) {
      yrow = i;
      xpix = XTAIL + (i / IMG_W) * PAD_W + (i % IMG_W);  // tap (0,0) source = padded (y, x)
    }
    ya = ys_a + yrow * YS + lane_col + 2 * co0w;
    xa = xs_a + xpix * XS + lane_col + 32 * wci;
  };
  auto kstep = [&](int ks, int carry, bool flush) {
    // k-slot order: lane group g holds virtual rows 32 ks + 4 g + q (first read) and + 16 (second read); dY and X
    // use the same order, so the contraction over the 32 rows is unchanged
    uint32_t ya_lo, ya_hi, xa_lo, xa_hi;
    row_addr(32 * ks + 4 * g + q4, carry, flush, ya_lo, xa_lo);
    row_addr(32 * ks + 4 * g + q4 + 16, carry, flush, ya_hi, xa_hi);
    bf8 ah[NCOW], al[NCOW];
    static_for<NCOW>([&](auto JC) {
      constexpr int j = decltype(JC)::value;
      ah[j] = tr_frag<32 * j>(ya_lo, ya_hi);
      if (SPLIT == 3) al[j] = tr_frag<YPL + 32 * j>(ya_lo, ya_hi);
    });
    bf8 bh[2], bl[2];
    auto read_b = [&](auto TC) {
      constexpr int tap = decltype(TC)::value;
      constexpr int XO = ((tap / 3) * PAD_W + (tap % 3)) * XS;  // tap shift in LDS bytes
      bh[tap & 1] = tr_frag<XO>(xa_lo, xa_hi);
      if (SPLIT == 3) bl[tap & 1] = tr_frag<XPL + XO>(xa_lo, xa_hi);
    };
    read_b(std::integral_constant<int, 0>{});
    static_for<9>([&](auto TC) {
      constexpr int tap = decltype(TC)::value;
      if constexpr (tap < 8) {
        read_b(std::integral_constant<int, tap + 1>{});
        // LDS returns in order: all but the reads just issued (2 ds_read per fragment) have landed
        if (SPLIT == 3) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (tap == 0) {
        if (bias_wave) {
          const s8v o = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};  // bf16 1.0
          const bf8 ones = __builtin_bit_cast(bf8, o);
#pragma unroll
          for (int j = 0; j < NCOW; ++j) {
            if (SPLIT == 3) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j], ones, accb[j], 0, 0, 0);
            accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], ones, accb[j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NCOW; ++j) {
        if (SPLIT == 3) {
          acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j], bh[tap & 1], acc[tap][j], 0, 0, 0);
          acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bl[tap & 1], acc[tap][j], 0, 0, 0);
        }
        acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bh[tap & 1], acc[tap][j], 0, 0, 0);
      }
    });
  };

  int carry = 0;  // deferred pixels of the previous patch waiting in the tail strips (0, 4 or 8)
  uint4 tv[TIT];
#pragma unroll
  for (int i = 0; i < TIT; ++i) tv[i] = uint4{0, 0, 0, 0};
  if constexpr (MAP) {
    const int ntile = a.tiles_x * a.tiles_y;
    constexpr int WTOT = NPAD * XCH, WIT = (WTOT + NTH - 1) / NTH;
    for (int u = p_begin; u < p_end; ++u) {
      const int pq = u / ntile, tile = u % ntile;
      const int ty0 = (tile / a.tiles_x) * IMG_W, tx0 = (tile % a.tiles_x) * IMG_W;
      __syncthreads();  // previous unit fully consumed
      // plane by plane (one round trip each; this path is not the headline shape, registers stay low)
      for (int pl = 0; pl < NPL; ++pl) {
        uint4 xv[WIT];
        const uint16_t *src = (pl ? a.xl : a.xh) + ci_base;
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
          const int c = min(tid + i * NTH, WTOT - 1), wp = c / XCH, ch = c % XCH;
          const int my = ty0 + wp / PAD_W - 1, mx = tx0 + wp % PAD_W - 1;
          const bool ok = my >= 0 && my < a.mh && mx >= 0 && mx < a.mw;
          const long off = (((long)pq * a.mh + min(max(my, 0), a.mh - 1)) * a.mw + min(max(mx, 0), a.mw - 1)) * CIN + 8 * ch;
          xv[i] = *reinterpret_cast<const uint4 *>(src + off);  // unconditional, clamped (see PlaneLoad)
          if (!ok) xv[i] = uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
          const int c = tid + i * NTH;
          if (WTOT % NTH == 0 || c < WTOT) *reinterpret_cast<uint4 *>(xs + pl * XPL + (XTAIL + c / XCH) * XS + 16 * (c % XCH)) = xv[i];
        }
        uint4 yv[YIT];
        const uint16_t *ysrc = (pl ? a.dyl : a.dyh) + co_base;
#pragma unroll
        for (int i = 0; i < YIT; ++i) {
          const int c = min(tid + i * NTH, YTOT - 1), pix = c / YCH, ch = c % YCH;
          const int oy = ty0 + pix / IMG_W, ox = tx0 + pix % IMG_W;
          const bool ok = oy < a.mh && ox < a.mw;
          const long off = (((long)pq * a.mh + min(oy, a.mh - 1)) * a.mw + min(ox, a.mw - 1)) * COUT + 8 * ch;
          yv[i] = *reinterpret_cast<const uint4 *>(ysrc + off);
          if (!ok) yv[i] = uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < YIT; ++i) {
          const int c = tid + i * NTH;
          if (YTOT % NTH == 0 || c < YTOT) *reinterpret_cast<uint4 *>(ys + pl * YPL + (c / YCH) * YS + 16 * (c % YCH)) = yv[i];
        }
      }
      __syncthreads();
#pragma unroll 1
      for (int ks = 0; ks < 4; ++ks) kstep(ks, 0, false);  // pixels 0..127: rows >= 100 read the zero dY row
    }
  }
  for (int p = MAP ? p_end : p_begin; p < p_end; ++p) {
    __syncthreads();  // previous patch fully consumed (its tail rows are in tv[])
    WG_STAMP(3)
    {
      // the deferred rows of the previous patch -> tail strips (read into tv[] before the barrier above)
#pragma unroll
      for (int i = 0; i < TIT; ++i) {
        const int c = tid + i * NTH;
        if (TTOT % NTH == 0 || c < TTOT) {
          int src, dst;
          tail_src_dst(c, src, dst);
          *reinterpret_cast<uint4 *>(xs + dst) = tv[i];
        }
      }
      // all global loads of the patch are issued before the first LDS store (one round trip)
      uint4 xv[NPL][XIT], yv[NPL][YIT];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const uint16_t *src = (pl ? a.xl : a.xh) + (long)p * NPIX * CIN + ci_base;
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
          const int c = min(tid + i * NTH, XTOT - 1);  // clamped, unconditional (see PlaneLoad)
          xv[pl][i] = *reinterpret_cast<const uint4 *>(src + (long)(c / XCH) * CIN + 8 * (c % XCH));
        }
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        // (the fused ReLU + GAP backward builds both planes from the activation hi plane: dyl is null then)
        const uint16_t *src = ((pl && !a.dgap) ? a.dyl : a.dyh) + (long)p * NPIX * COUT + co_base;
#pragma unroll
        for (int i = 0; i < YIT; ++i) {
          const int c = min(tid + i * NTH, YTOT - 1);
          yv[pl][i] = *reinterpret_cast<const uint4 *>(src + (long)(c / YCH) * COUT + 8 * (c % YCH));
        }
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
          const int c = tid + i * NTH;
          if (XTOT % NTH == 0 || c < XTOT)
            *reinterpret_cast<uint4 *>(xs + pl * XPL + (XTAIL + interior_pp(c / XCH)) * XS + 16 * (c % XCH)) = xv[pl][i];
        }
      if (a.dgap) {
        // dY[i][c] = dgap[c] / 100 where the forward activation (hi plane, in yv[0]) is non-zero
        static_assert(NTH % YCH == 0, "a thread keeps its channel chunk");
        const int ch = tid % YCH;
        uint32_t gh[4], gl[4];
        gap_split8(a.dgap + (long)p * COUT + co_base + 8 * ch, gh, gl);
#pragma unroll
        for (int i = 0; i < YIT; ++i) {
          const int c = tid + i * NTH;
          if (YTOT % NTH == 0 || c < YTOT) {
            uint4 oh, ol;
            gap_mask8(yv[0][i], gh, gl, oh, ol);
            const int off = (c / YCH) * YS + 16 * ch;
            *reinterpret_cast<uint4 *>(ys + off) = oh;
            if (SPLIT == 3) *reinterpret_cast<uint4 *>(ys + YPL + off) = ol;
          }
        }
      } else {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int i = 0; i < YIT; ++i) {
            const int c = tid + i * NTH;
            if (YTOT % NTH == 0 || c < YTOT) *reinterpret_cast<uint4 *>(ys + pl * YPL + (c / YCH) * YS + 16 * (c % YCH)) = yv[pl][i];
          }
      }
    }
    __syncthreads();
    WG_STAMP(0)
    WG_STAMP(1)
#ifdef CRW_CONV_STAMPS
    if (a.stamps && tid == 0 && p - p_begin < 60)  // absolute k-loop start times of the first patches (phase drift study)
      a.stamps[(long)blockIdx.x * 64 + 4 + (p - p_begin)] = (long long)__builtin_amdgcn_s_memtime();
#endif
    // carry 0: pixels 0..95 now, 4 deferred; carry 4: 4 old + pixels 0..91, 8 deferred; carry 8: 8 old + all 100
    // pixels (4 k-steps, the last one padded with the zero row), nothing deferred
    const int nks = carry == 8 ? 4 : 3;
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) kstep(ks, carry, false);
    carry = carry == 8 ? 0 : carry + 4;
    // this thread's share of the rows a later k-step still needs (image rows 9-10, dY rows 92..99)
#pragma unroll
    for (int i = 0; i < TIT; ++i) {
      const int c = min(tid + i * NTH, TTOT - 1);
      int src, dst;
      tail_src_dst(c, src, dst);
      tv[i] = *reinterpret_cast<const uint4 *>(xs + src);
    }
    WG_STAMP(2)
  }
  if (carry > 0) {  // the slice ended with deferred pixels: one more k-step over the tail strips alone
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TIT; ++i) {
      const int c = tid + i * NTH;
      if (TTOT % NTH == 0 || c < TTOT) {
        int src, dst;
        tail_src_dst(c, src, dst);
        *reinterpret_cast<uint4 *>(xs + dst) = tv[i];
      }
    }
    __syncthreads();
    kstep(0, carry, true);
  }
#ifdef CRW_CONV_STAMPS
  if (a.stamps && tid == 0)
    for (int k = 0; k < 4; ++k) a.stamps[(long)blockIdx.x * 64 + k] = ph_[k];
#endif

  // partial sums of this patch slice -> workspace (plain stores; a second kernel adds the slices
  // in a fixed order, so gradients are bitwise reproducible and no float atomics are needed).
  // acc[tap][j][r] = dW[co_base + co0w + 16 j + 4 g + r][ci_base + 16 wci + lane&15][tap]; the partial slab is
  // laid out [tap][co][ci] so that the 16 lanes of a fragment row store 64 contiguous bytes (a [co][ci][tap]
  // slab makes every 4-byte store its own HBM transaction: PMC WRITE_SIZE was 20x the slab size)
  float *dwp = a.dw_part + (long)wg_slice * COUT * CIN * 9;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int j = 0; j < NCOW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co_base + co0w + 16 * j + 4 * g + r, ci = ci_base + 16 * wci + (lane & 15);
        dwp[((long)tap * COUT + co) * CIN + ci] = acc[tap][j][r];
      }
  // bias partials: column 0 of accb[j] holds sum_pix dY[pix][co_base + co0w + 16 j + 4 g + r]
  if (bias_wave && (lane & 15) == 0) {
#pragma unroll
    for (int j = 0; j < NCOW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) a.db_part[(long)wg_slice * COUT + co_base + co0w + 16 * j + 4 * g + r] = accb[j][r];
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient, second generation ("streamed"): ONE workgroup of 8 waves per CU, two waves per SIMD that both
// compute all the time (tools/ubench/wgrad_kstep.hip: this wave arrangement keeps the matrix pipe as busy as a loop
// without any LDS read, where a lone wave per SIMD loses 10-12 % and the two-workgroup arrangement above loses each
// workgroup's load phase).  Workgroup tile = 9 taps x 128 output x 64 input channels, wave tile = 4 co x 1 ci x 9 taps.
//   * The reduction runs over the pixel STREAM of a patch slice (pixel G = 100 * patch + pixel, contiguous in the dY
//     planes): k-step n takes stream pixels [32 n, 32 n + 32), whatever patches they belong to -- no per-patch
//     padding, no tail strips.
//   * dY lives in LDS one k-step at a time (two ring slots of 32 rows), X as whole 12x12 patch images in two slots
//     (patch q in slot q & 1); a k-step that straddles two patches reads both slots (per-lane slot choice).
//   * Loads are register-staged one k-step ahead: at the top of k-step n every thread issues its global loads for
//     k-step n + 1 (32 dY rows; and, where a new patch begins at k-step n + 2 / n + 1, the hi / lo plane of its X
//     image), computes k-step n, then stores the staged registers to LDS; ONE barrier per k-step.  A slot is only
//     written a full k-step after its last reader (see the schedule in the kernel).
//   * Measured and NOT kept: the same kernel with LDS-DMA staging (global_load_lds two k-steps ahead into a three-slot dY ring
//     and 16-pitch X images, source-side swizzles, counted vmcnt + raw barrier, no staging registers): parity-green, 1123 us
//     against 1073 us for this register-staged form (conv5, P = 16128; in-step 1205 vs 1070) -- 3.6 DMA issues and ~150
//     address VALU per wave and k-step cost more than the staging they replace.  What the staging costs here: CRW_WGRAD_DIAG.
//   * LDS planes on a 2C+32 row stride, fragment rows in stream order: eight consecutive dY rows start in eight different
//     bank octets; the X reads (eight pixels that usually straddle an image-row end) cost 1.6 LDS cycles each instead of the 2.0
//     of the first version (rows dealt by parity on a 2C+16 stride) -- worth 0.5-1 % of the kernel.
constexpr int W2_NCO = 128, W2_NCI = 64, W2_NW = 8, W2_KROWS = 32;
template <int C>
constexpr int w2_stride() { return 2 * C + 32; }  // LDS row stride of the streamed kernel's planes (see the fragment-row order below)

// DIAG (timing-only diagnostic builds of the conv5 shape, CRW_WGRAD_DIAG=1..5; results are wrong): 1 = no barrier in the k-loop,
// 2 = no staging (no global loads, no LDS stores), 3 = both, 4 = global loads but no LDS stores, 5 = LDS stores but no global
// loads -- what the k-loop costs without its pipeline partners
template <int SPLIT, int CIN, int COUT, int DIAG = 0>
__global__ __launch_bounds__(W2_NW * 64, 2) void conv3x3_wgrad2_kernel(WgradArgs a) {
  constexpr int NCO = W2_NCO, NCI = W2_NCI, NTH = W2_NW * 64, NPL = (SPLIT == 3) ? 2 : 1;
  constexpr int XS = w2_stride<NCI>(), XPL = NPAD * XS, XSLOT = NPL * XPL;     // X: 12x12 padded image per plane
  constexpr int YS = w2_stride<NCO>(), YPL = W2_KROWS * YS, YSLOT = NPL * YPL; // dY: 32 rows per plane
  constexpr int NCOW = 4, WCI = 4;                                              // wave tile: 4 co tiles x 1 ci tile
  static_assert(COUT == NCO && CIN % NCI == 0, "tiling");
  constexpr int NGRP = CIN / NCI;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *xs = lds, *ys = lds + 2 * XSLOT;
  char *gs = ys + 2 * YSLOT;  // dgap mode: [2 patch slots][hi, lo][128 channels] bf16 = dgap / 100 of the two resident patches

  // workgroup id -> (slice, ci group): the groups of a slice sit on one XCD (ids L and L + 8), see the kernel above
  const int wg_slice = a.xcd_map ? ((int)blockIdx.x / 8 / NGRP) * 8 + (int)blockIdx.x % 8 : (int)blockIdx.x / NGRP;
  const int grp_ci = a.xcd_map ? ((int)blockIdx.x / 8) % NGRP : (int)blockIdx.x % NGRP;
  const int ci_base = grp_ci * NCI;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int wci = wave % WCI, wco = wave / WCI;
  const int wci_s = __builtin_amdgcn_readfirstlane(wci);  // provably wave-uniform copy for scalar branches
  const int co0w = wco * NCOW * 16;

  const int p_begin = wg_slice * a.patches_per_block;
  const int np = min(a.P, p_begin + a.patches_per_block) - p_begin;  // patches of this slice (>= 1 by construction)
  const int npx = np * NPIX;                                         // pixels of the stream
  const int NK = (npx + W2_KROWS - 1) / W2_KROWS;

  f32x4 acc[9][NCOW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NCOW; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};  // bias gradient of co tile `wci` of this wave's co half (ci group 0 only)
  const bool bias_wg = grp_ci == 0;        // workgroup-uniform

  // zero both X slots once (the halo stays zero: only interior pixels are ever written)
  for (int c = tid; c < 2 * XSLOT / 16; c += NTH) *reinterpret_cast<uint4 *>(xs + 16 * c) = uint4{0, 0, 0, 0};

  // ---- staging: what a thread loads / stores ----------------------------------------------------------------
  // dY of a k-step: 32 rows x 16 chunks (8 channels) = 512 chunks = one per thread and plane
  const int yrow = tid >> 4, ych = tid & 15;
  // X plane of a patch: 100 pixels x 8 chunks = 800 chunks: threads take chunk tid and tid + 512
  constexpr int XCH = NCI / 8, XTOT = NPIX * XCH;
  // Global loads go through buffer descriptors rooted at the slice (scalar base + 32-bit per-thread offset: no 64-bit
  // address registers are kept alive across the k-loop, and a row past the end of the stream reads as zero)
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  const long dy_off = (long)p_begin * NPIX * COUT, x_off = (long)p_begin * NPIX * CIN + ci_base;
  const __amdgpu_buffer_rsrc_t dyh_rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dyh + dy_off), 0, npx * COUT * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyl_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void *)((SPLIT == 3 && !a.dgap ? a.dyl : a.dyh) + dy_off), 0, npx * COUT * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t xh_rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xh + x_off), 0, (npx * CIN - ci_base) * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t xl_rs = __builtin_amdgcn_make_buffer_rsrc((void *)((SPLIT == 3 ? a.xl : a.xh) + x_off), 0,
                                                                          (npx * CIN - ci_base) * 2, 0x00020000);
  const int y_voff = (yrow * COUT + 8 * ych) * 2;
  int x_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = min(tid + i * NTH, XTOT - 1);
    x_voff[i] = ((c / XCH) * CIN + 8 * (c % XCH)) * 2;
  }
  uint4 sy[NPL], sx[2];
  auto load_dy = [&](int n) {  // the 32 rows of k-step n; rows past the stream are out of the descriptor's range: zeros
    const int soff = n * (W2_KROWS * COUT * 2);
    constexpr int AUX = DIAG == 6 ? 2 : 0;  // DIAG 6: non-temporal policy on the staged loads (results stay correct)
    sy[0] = __builtin_bit_cast(uint4, (u4v)__builtin_amdgcn_raw_buffer_load_b128(dyh_rs, y_voff, soff, AUX));
    if (SPLIT == 3 && !a.dgap) sy[NPL - 1] = __builtin_bit_cast(uint4, (u4v)__builtin_amdgcn_raw_buffer_load_b128(dyl_rs, y_voff, soff, AUX));
  };
  auto store_dy = [&](int n) {
    uint4 vh = sy[0], vl = uint4{0, 0, 0, 0};
    if (SPLIT == 3 && !a.dgap) vl = sy[NPL - 1];
    if (a.dgap) {  // fused ReLU + GAP backward: dY = dgap / 100 where the forward activation (sy[0]) is non-zero
      // the split dgap row of the row's patch waits in LDS (stage_gap): no global round trip here.  Rows past the
      // stream loaded zero activations: their dY is zero whatever slot is read.
      const int q = (32 * n + yrow) / NPIX;
      const uint4 h4 = *reinterpret_cast<const uint4 *>(gs + (q & 1) * (4 * COUT) + 16 * ych);
      const uint4 l4 = *reinterpret_cast<const uint4 *>(gs + (q & 1) * (4 * COUT) + 2 * COUT + 16 * ych);
      const uint32_t gh[4] = {h4.x, h4.y, h4.z, h4.w}, gl[4] = {l4.x, l4.y, l4.z, l4.w};
      gap_mask8(sy[0], gh, gl, vh, vl);
    }
    char *dst = ys + (n & 1) * YSLOT + yrow * YS + 16 * ych;
    *reinterpret_cast<uint4 *>(dst) = vh;
    if (SPLIT == 3) *reinterpret_cast<uint4 *>(dst + YPL) = vl;
  };
  auto load_x = [&](int q, int pl) {  // plane pl of patch q of the slice
    const int soff = q * (NPIX * CIN * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      sx[i] = __builtin_bit_cast(uint4, (u4v)__builtin_amdgcn_raw_buffer_load_b128(pl ? xl_rs : xh_rs, x_voff[i], soff, DIAG == 6 ? 2 : 0));
  };
  auto store_x = [&](int q, int pl) {
    char *dst = xs + (q & 1) * XSLOT + pl * XPL;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * NTH;
      if (c < XTOT) *reinterpret_cast<uint4 *>(dst + interior_pp(c / XCH) * XS + 16 * (c % XCH)) = sx[i];
    }
  };
  // dgap mode: split dgap[patch q] / 100 into bf16 hi / lo once per patch (16 threads, 8 channels each) -> LDS strip
  auto stage_gap = [&](int q) {
    if (tid < COUT / 8) {
      uint32_t gh[4], gl[4];
      gap_split8(a.dgap + (long)(p_begin + q) * COUT + 8 * tid, gh, gl);
      *reinterpret_cast<uint4 *>(gs + (q & 1) * (4 * COUT) + 16 * tid) = uint4{gh[0], gh[1], gh[2], gh[3]};
      *reinterpret_cast<uint4 *>(gs + (q & 1) * (4 * COUT) + 2 * COUT + 16 * tid) = uint4{gl[0], gl[1], gl[2], gl[3]};
    }
  };
  // first k-step that touches patch q
  auto first_k = [&](int q) { return (q * NPIX) / W2_KROWS; };
  // patch whose first k-step is n, or -1
  auto patch_starting_at = [&](int n) {
    if (n >= NK) return -1;
    const int q = min((32 * n + 31) / NPIX, np - 1);  // patch of the last stream pixel of k-step n
    return first_k(q) == n ? q : -1;
  };

  // ---- fragment addressing -------------------------------------------------------------------------------------
  const uint32_t xs_a = (uint32_t)(uintptr_t)(lds_cp)xs, ys_a = (uint32_t)(uintptr_t)(lds_cp)ys;
  const int t16 = lane & 15, q4 = t16 >> 2, pq = t16 & 3;
  const uint32_t lane_col = 8 * (pq & 1) + 16 * (pq >> 1);
  // fragment row order: lane group g, fragment row q4 -> k-step row 4 g + q4  (+16 for the second read).  On the 2C+32 row stride
  // eight consecutive dY rows start in eight different bank octets (conflict-free), and the eight X pixels of a lane-group pair,
  // which straddle an image-row end more often than not, cost 1.6 LDS cycles per read on average (bank simulation over the
  // pixel stream); the parity order on the 2C+16 stride of the first version cost 2.0
  const int r_lo = 4 * g + q4, r_hi = r_lo + 16;
  const uint32_t ya_lo0 = ys_a + r_lo * YS + lane_col + 2 * co0w, ya_hi0 = ys_a + r_hi * YS + lane_col + 2 * co0w;
  auto x_addr = [&](int G) {  // tap-(0,0) source pixel of stream pixel G in its patch slot
    const int Gc = min(G, npx - 1);  // rows past the stream multiply zero dY rows: any valid address
    const int q = Gc / NPIX, i = Gc - q * NPIX;
    return xs_a + (q & 1) * XSLOT + ((i / IMG_W) * PAD_W + (i % IMG_W)) * XS + lane_col + 32 * wci;
  };

  auto kstep = [&](int n) {
    const uint32_t ya_lo = ya_lo0 + (n & 1) * YSLOT, ya_hi = ya_hi0 + (n & 1) * YSLOT;
    const uint32_t xa_lo = x_addr(32 * n + r_lo), xa_hi = x_addr(32 * n + r_hi);
    bf8 ah[NCOW], al[NCOW];
    static_for<NCOW>([&](auto JC) {
      constexpr int j = decltype(JC)::value;
      ah[j] = tr_frag<32 * j>(ya_lo, ya_hi);
      if (SPLIT == 3) al[j] = tr_frag<YPL + 32 * j>(ya_lo, ya_hi);
    });
    bf8 bh[2], bl[2];
    auto read_b = [&](auto TC) {
      constexpr int tap = decltype(TC)::value;
      constexpr int XO = ((tap / 3) * PAD_W + (tap % 3)) * XS;
      bh[tap & 1] = tr_frag<XO>(xa_lo, xa_hi);
      if (SPLIT == 3) bl[tap & 1] = tr_frag<XPL + XO>(xa_lo, xa_hi);
    };
    read_b(std::integral_constant<int, 0>{});
    static_for<9>([&](auto TC) {
      constexpr int tap = decltype(TC)::value;
      if constexpr (tap < 8) {
        read_b(std::integral_constant<int, tap + 1>{});
        if (SPLIT == 3) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (tap == 0) {
        if (bias_wg) {  // one co tile per wave (wave-uniform choice, static fragment index)
          const s8v o = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};  // bf16 1.0
          const bf8 ones = __builtin_bit_cast(bf8, o);
          // wci_s is a SCALAR (readfirstlane): with a per-lane condition hipcc predicates the block through EXEC, and the
          // matrix instructions ignore EXEC -- every wave would add all four tiles
          static_for<NCOW>([&](auto JC) {
            constexpr int j = decltype(JC)::value;
            if (wci_s == j) {
              if (SPLIT == 3) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j], ones, accb, 0, 0, 0);
              accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], ones, accb, 0, 0, 0);
            }
          });
        }
      }
#pragma unroll
      for (int j = 0; j < NCOW; ++j) {
        if (SPLIT == 3) {
          acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j], bh[tap & 1], acc[tap][j], 0, 0, 0);
          acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bl[tap & 1], acc[tap][j], 0, 0, 0);
        }
        acc[tap][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bh[tap & 1], acc[tap][j], 0, 0, 0);
      }
    });
  };

  // ---- prologue: patch 0 and k-step 0 --------------------------------------------------------------------------
  load_dy(0);
  load_x(0, 0);
  if (a.dgap) stage_gap(0);
  __syncthreads();  // (the zero fill above is complete; patch 0's dgap strip is visible)
  store_x(0, 0);
  if (SPLIT == 3) {
    load_x(0, 1);
    store_x(0, 1);
  }
  store_dy(0);
  __syncthreads();

  // ---- main loop ---------------------------------------------------------------------------------------------
  // k-step n: issue the loads for k-step n + 1 (dY; X hi plane of the patch that starts at n + 2, X lo plane -- or the only
  // plane -- of the patch that starts at n + 1), compute, store the staged registers, barrier.
  // Slot safety: patch q (first k-step F) is stored at the end of k-steps F - 2 (hi) / F - 1 (lo); the patch q - 2 that
  // occupied the slot was last read in k-step <= F - 3 (its last pixel is 100 (q - 1) - 1, and 100 q / 32 - (100 q - 101) / 32 > 3).
#pragma unroll 1
  for (int n = 0; n < NK; ++n) {
    const bool more = n + 1 < NK;
    const int q_2 = patch_starting_at(n + 2);                       // wave-uniform
    const int q_hi = (SPLIT == 3) ? q_2 : -1;
    const int q_lo = patch_starting_at(n + 1);
    // (a patch starts every 3.125 k-steps: q_hi and q_lo are never both set)
    const int xq = q_lo >= 0 ? q_lo : q_hi, xpl = q_lo >= 0 ? NPL - 1 : 0;
    // the loads are UNCONDITIONAL (clamped): a load under a run-time condition makes hipcc wait for it on the spot (the
    // value is merged with the not-taken path), which would serialise a memory round trip into every k-step
    if (DIAG != 2 && DIAG != 3 && DIAG != 5) {
      load_dy(more ? n + 1 : n);
      load_x(xq >= 0 ? xq : 0, xpl);
    }
    kstep(n);
    if (DIAG == 2 || DIAG == 3) {
      if (DIAG & 1) continue;
      __syncthreads();
      continue;
    }
    if (DIAG == 4) {  // keep the loaded registers alive without storing them
      asm volatile("" ::"v"(sy[0].x), "v"(sy[NPL - 1].y), "v"(sx[0].x), "v"(sx[1].y));
      __syncthreads();
      continue;
    }
    if (more) store_dy(n + 1);
    if (xq >= 0) store_x(xq, xpl);
    // the dgap strip of a patch is written two k-steps before its first pixel: its first reader is the store_dy at the end
    // of the NEXT k-step (behind the barrier); the previous tenant of the slot (patch q - 2) ended >= 3 k-steps before
    if (a.dgap && q_2 >= 0) stage_gap(q_2);
    if (!(DIAG & 1)) __syncthreads();
  }

  float *dwp = a.dw_part + (long)wg_slice * COUT * CIN * 9;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int j = 0; j < NCOW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0w + 16 * j + 4 * g + r, ci = ci_base + 16 * wci + (lane & 15);
        dwp[((long)tap * COUT + co) * CIN + ci] = acc[tap][j][r];
      }
  if (bias_wg && (lane & 15) == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a.db_part[(long)wg_slice * COUT + co0w + 16 * wci + 4 * g + r] = accb[r];
  }
}

// out[e] = sum over slices of part[s][e]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void slice_sum_kernel(const float *__restrict__ part, int nslice, long n,
                                                        float *__restrict__ out) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < nslice; ++k) s += part[(long)k * n + e];
    out[e] = s;
  }
}

// The weight gradient: slabs are [tap][co*ci], the result is the framework layout [co*ci][tap].
// A block owns 64 consecutive slab elements; its 4 waves each add a quarter of the slices (8 independent
// loads in flight per lane), the quarters meet in LDS in a fixed order (deterministic).
// The blocks past nblk_dw add the bias-gradient slices (few outputs, many slices): one wave per output, lanes stride the
// slices, fixed shuffle tree -- one launch for both.
__global__ __launch_bounds__(256) void slice_sum_dw_kernel(const float *__restrict__ part, int nslice, int coci,
                                                           float *__restrict__ out, int nblk_dw,
                                                           const float *__restrict__ bpart, int cout, float *__restrict__ bout) {
  __shared__ float red[4][64];
  const long n = 9L * coci;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  if ((int)blockIdx.x >= nblk_dw) {  // block-uniform
    const int e = ((int)blockIdx.x - nblk_dw) * 4 + q;
    if (e >= cout) return;
    float s = 0.f;
    for (int k = lane; k < nslice; k += 64) s += bpart[(long)k * cout + e];
    s = wave_sum(s);
    if (lane == 0) bout[e] = s;
    return;
  }
  const long e = (long)blockIdx.x * 64 + lane;
  const int per = (nslice + 3) / 4, k0 = q * per, k1 = min(nslice, k0 + per);
  float s = 0.f;
  if (e < n) {
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(k + u) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += part[(long)k * n + e];
  }
  red[q][lane] = s;
  __syncthreads();
  if (q == 0 && e < n) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int tap = e / coci, cc = e % coci;
    out[(long)cc * 9 + tap] = t;
  }
}

// ---- small helpers --------------------------------------------------------------------------------
// fp32 W[co][ci][3][3] -> forward and backward-data (taps flipped, ci <-> co) planes, both in MFMA
// fragment order: [tap][k-chunk of 32][16-channel output tile][lane = 16 (k/8) + channel][8 k]
__device__ inline long frag_index(int tap, int n, int k, int N, int K) {  // n: output channel, k: reduction channel
  return ((((long)tap * (K / 32) + k / 32) * (N / 16) + n / 16) * 64 + ((k % 32) / 8) * 16 + n % 16) * 8 + k % 8;
}
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ w, int CO, int CI,
                                                           uint16_t *fh, uint16_t *fl, uint16_t *bh, uint16_t *bl) {
  const long n = (long)CO * CI * 9;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int tap = e % 9, ci = (e / 9) % CI, co = e / (9L * CI);
    const float v = w[e];
    const uint16_t h = f2bf(v), l = f2bf(v - bf2f(h));
    const long of = frag_index(tap, co, ci, CO, CI), ob = frag_index(8 - tap, ci, co, CI, CO);
    fh[of] = h; bh[ob] = h;
    if (fl) { fl[of] = l; bl[ob] = l; }
  }
}

// fp32 NCHW [P][C][10][10] -> channels-last planes [P][100][C]
__global__ __launch_bounds__(256) void pack_input_kernel(const float *__restrict__ x, int P, int C, int npix, uint16_t *xh,
                                                         uint16_t *xl) {
  const long n = (long)P * npix * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = e % C, i = (e / C) % npix;
    const long p = e / ((long)C * npix);
    const float v = x[(p * C + c) * npix + i];
    const uint16_t h = f2bf(v);
    xh[e] = h;
    if (xl) xl[e] = f2bf(v - bf2f(h));
  }
}

// dY[p][i][c] = dgap[p][c] / npix where the forward output was positive (ReLU); 8 channels (16 bytes of a plane) per thread
// (one element per thread, 2-byte accesses: 3.3 ms for the 26 x 26 x 128 maps of 12288 patches; HBM floor 1.1 ms)
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float *__restrict__ dgap, const uint16_t *__restrict__ yh,
                                                      int P, int C, int npix, uint16_t *dh, uint16_t *dl) {
  const long n8 = (long)P * npix * C / 8;
  const int c8 = C / 8;
  const float inv = 1.0f / (float)npix;  // (npix = 100: the same constant the fused loaders multiply by)
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long)gridDim.x * 256) {
#pragma clang fp contract(off)  // same rounding as the fused loaders
    const int c = (int)(e % c8) * 8;
    const long p = e / ((long)c8 * npix);
    const uint4 y = *reinterpret_cast<const uint4 *>(yh + e * 8);
    const float4 g0 = *reinterpret_cast<const float4 *>(dgap + p * C + c), g1 = *reinterpret_cast<const float4 *>(dgap + p * C + c + 4);
    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const uint32_t yw[4] = {y.x, y.y, y.z, y.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a0 = (yw[w] & 0x7fffu) ? gv[2 * w] * inv : 0.f, a1 = (yw[w] & 0x7fff0000u) ? gv[2 * w + 1] * inv : 0.f;
      const uint16_t h0 = f2bf(a0), h1 = f2bf(a1);
      h[w] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      l[w] = (uint32_t)f2bf(a0 - bf2f(h0)) | ((uint32_t)f2bf(a1 - bf2f(h1)) << 16);
    }
    *reinterpret_cast<uint4 *>(dh + e * 8) = uint4{h[0], h[1], h[2], h[3]};
    if (dl) *reinterpret_cast<uint4 *>(dl + e * 8) = uint4{l[0], l[1], l[2], l[3]};
  }
}

template <int C>
constexpr size_t conv_lds_bytes(int npl) { return (size_t)(npl - 1) * PLANE_ROWS * crs<C>() + (size_t)NPAD * crs<C>(); }

template <int SPLIT, int CIN, int COUT, int MODE, int NW>
int launch_conv_nw(const ConvArgs &a, hipStream_t s) {
  constexpr int CMAX = CIN > COUT ? CIN : COUT;
  size_t lds = conv_lds_bytes<CMAX>(SPLIT == 3 ? 2 : 1);
#ifdef CRW_CONV_STAMPS
  if (const char *e = getenv("CRW_CONV_LDS_PAD")) lds += (size_t)atoi(e);  // diagnostics: force fewer workgroups per CU
#endif
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void *)conv3x3_kernel<SPLIT, CIN, COUT, MODE, NW>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_kernel<SPLIT, CIN, COUT, MODE, NW>), dim3(a.P), dim3(NW * 64), lds, s, a);
  return check_launch();
}

constexpr int MT_SMALL = 4;  // row tiles of the small-edge-tile launch (tiles with at most 64 in-map pixels)

template <int SPLIT, int CIN, int COUT, int MODE, int MTL>
int launch_conv_map_cls(const ConvArgs &a, hipStream_t s) {
  constexpr int CMAX = CIN > COUT ? CIN : COUT, NW = (SPLIT == 3 ? 8 : 4);
  const size_t lds = (size_t)(SPLIT == 3 ? 2 : 1) * NPAD * row_stride<CMAX>();
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void *)conv3x3_kernel<SPLIT, CIN, COUT, MODE, NW, true, MTL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_kernel<SPLIT, CIN, COUT, MODE, NW, true, MTL>), dim3(a.P * a.ncls), dim3(NW * 64), lds, s, a);
  return check_launch();
}

// two classes of launches: the tiles with more than 64 in-map pixels on 7 row tiles, the small edge tiles on 4; a launch carries
// the indices of at most 64 tiles in its kernel arguments, so a map with more tiles of a class (over 80 x 80 pixels, or a long
// strip) takes one launch per 64 of them -- any map size, like the reference's convolutions
template <int SPLIT, int CIN, int COUT, int MODE = 0>
int launch_conv_map(const ConvArgs &a0, hipStream_t s) {
  const int ntile = a0.tiles_x * a0.tiles_y;
  if (ntile > 65535) return CRW_EINVAL;  // 16-bit tile indices: maps up to 2550 x 2550
  constexpr int CHUNK = (int)(sizeof(a0.tile_list) / sizeof(a0.tile_list[0]));
  ConvArgs big = a0, small = a0;
  big.ncls = small.ncls = 0;
  for (int t = 0; t < ntile; ++t) {
    const int th = std::min(IMG_W, a0.mh - (t / a0.tiles_x) * IMG_W), tw = std::min(IMG_W, a0.mw - (t % a0.tiles_x) * IMG_W);
    if (tw * th > 16 * MT_SMALL) {
      big.tile_list[big.ncls++] = (unsigned short)t;
      if (big.ncls == CHUNK) {
        CRW_TRY((launch_conv_map_cls<SPLIT, CIN, COUT, MODE, MT>(big, s)));
        big.ncls = 0;
      }
    } else {
      small.tile_list[small.ncls++] = (unsigned short)t;
      if (small.ncls == CHUNK) {
        CRW_TRY((launch_conv_map_cls<SPLIT, CIN, COUT, MODE, MT_SMALL>(small, s)));
        small.ncls = 0;
      }
    }
  }
  if (big.ncls) CRW_TRY((launch_conv_map_cls<SPLIT, CIN, COUT, MODE, MT>(big, s)));
  if (small.ncls) CRW_TRY((launch_conv_map_cls<SPLIT, CIN, COUT, MODE, MT_SMALL>(small, s)));
  return CRW_OK;
}

template <int SPLIT, int CIN, int COUT, int MODE>
int launch_conv(const ConvArgs &a, hipStream_t s) {
  // measured (tools/probe_conv.py, P = 16128): plain bf16 is 10 % faster with 4 waves than with 8
  static const char *force = getenv("CRW_CONV_NW");  // diagnostics (tools/probe_conv.py): force 4 or 8 waves
  if (force && force[0] == '4') return launch_conv_nw<SPLIT, CIN, COUT, MODE, 4>(a, s);
  if (force && force[0] == '8') return launch_conv_nw<SPLIT, CIN, COUT, MODE, 8>(a, s);
  // in-step A/B (tools/ab_kernels.py, CRW_CONV_NW): hi/lo pairs run best with 8 waves.  (At 32 output channels -- conv3
  // backward-data, where 8 waves leave a wave 2 of the 7 row tiles -- 4 waves were 6-9 % faster until the 8-wave kernel got its
  // third workgroup per CU, see conv_waves_per_simd: 265 -> 259 us.)
  return launch_conv_nw<SPLIT, CIN, COUT, MODE, (SPLIT == 3 ? 8 : 4)>(a, s);
}

// input channels per workgroup (one 16-channel tile per wave)
constexpr int wgrad_nci(int cin) { return cin > 64 ? 64 : cin; }

template <int SPLIT, int CIN, int COUT>
int launch_wgrad(const WgradArgs &a, int nblk, hipStream_t s) {
  // measured (tools/probe_conv.py): hi/lo pairs are faster with 4 waves (144 accumulator registers per wave, 26
  // fragment reads per 108 MFMAs), plain bf16 with 8 (4 waves per SIMD hide the load phase)
  constexpr int NCI = wgrad_nci(CIN), NW = SPLIT == 3 ? 4 : 8;
  const size_t lds = (size_t)(SPLIT == 3 ? 2 : 1) * ((XTAIL + NPAD) * row_stride<NCI>() + YROWS_ALL * row_stride<WG_NCO>());
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void *)conv3x3_wgrad_kernel<SPLIT, CIN, COUT, NCI, NW>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<SPLIT, CIN, COUT, NCI, NW>), dim3(nblk * (CIN / NCI) * (COUT / WG_NCO)),
                     dim3(NW * 64), lds, s, a);
  return check_launch();
}

template <int SPLIT, int CIN, int COUT>
int launch_wgrad_map(const WgradArgs &a, int nblk, hipStream_t s) {
  constexpr int NCI = wgrad_nci(CIN), NW = SPLIT == 3 ? 4 : 8;
  const size_t lds = (size_t)(SPLIT == 3 ? 2 : 1) * ((XTAIL + NPAD) * row_stride<NCI>() + YROWS_ALL * row_stride<WG_NCO>());
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void *)conv3x3_wgrad_kernel<SPLIT, CIN, COUT, NCI, NW, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<SPLIT, CIN, COUT, NCI, NW, true>), dim3(nblk * (CIN / NCI) * (COUT / WG_NCO)),
                     dim3(NW * 64), lds, s, a);
  return check_launch();
}

template <int SPLIT, int CIN, int COUT, int DIAG = 0>
int launch_wgrad2(const WgradArgs &a, int nslice, hipStream_t s) {
  constexpr int NPL = SPLIT == 3 ? 2 : 1;
  const size_t lds = (size_t)2 * NPL * (NPAD * w2_stride<W2_NCI>() + W2_KROWS * w2_stride<W2_NCO>()) + 2 * 4 * W2_NCO;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void *)conv3x3_wgrad2_kernel<SPLIT, CIN, COUT, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_wgrad2_kernel<SPLIT, CIN, COUT, DIAG>), dim3(nslice * (CIN / W2_NCI)), dim3(W2_NW * 64), lds, s, a);
  return check_launch();
}

inline int ew_grid(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace
}  // namespace crw

using namespace crw;

long long *g_conv_stamps = nullptr;  // set by crw_debug_conv_stamps (diagnostics)

extern "C" {

// diagnostics: when non-null, crw_enc_conv3x3 launches record s_memtime at 5 phase boundaries per workgroup
#ifdef CRW_CONV_STAMPS
void crw_debug_conv_stamps(long long *buf) { g_conv_stamps = buf; }
#endif

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
  hipLaunchKernelGGL(pack_input_kernel, dim3(ew_grid((long)P * NPIX * C)), dim3(256), 0, (hipStream_t)stream, x, P, C,
                     NPIX, xh, xl);
  return check_launch();
}

int crw_enc_pack_input_map(const float *x, int P, int C, int H, int W, uint16_t *xh, uint16_t *xl, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !xh || P < 1 || C < 8 || C % 8 || H < 1 || W < 1) return CRW_EINVAL;
  hipLaunchKernelGGL(pack_input_kernel, dim3(ew_grid((long)P * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, P, C,
                     H * W, xh, xl);
  return check_launch();
}

int crw_enc_conv3x3_map(int mode, int split, int P, int H, int W, int cin, int cout, const uint16_t *x_hi, const uint16_t *x_lo,
                        const uint16_t *w_hi, const uint16_t *w_lo, const float *bias, const uint16_t *mask_hi, uint16_t *y_hi,
                        uint16_t *y_lo, float *y_f32, float *gap_part, crw_stream_t stream) {
  clear_stale_error();
  if (!x_hi || !w_hi || P < 1 || H < 1 || W < 1 || (split != 1 && split != 3) || (mode != 0 && mode != 1)) return CRW_EINVAL;
  if (split == 3 && (!w_lo || !x_lo)) return CRW_EINVAL;
  if (!y_hi && !gap_part && !y_f32) return CRW_EINVAL;
  if (mode == 1 && (gap_part || bias)) return CRW_EINVAL;
  const int tx = (W + IMG_W - 1) / IMG_W, ty = (H + IMG_W - 1) / IMG_W;
  if ((long)P * tx * ty > 0x7fffffffL) return CRW_EINVAL;
  ConvArgs a{x_hi, x_lo, w_hi, w_lo, bias, mask_hi, y_hi, y_lo, y_f32, gap_part, nullptr, P, nullptr, H, W, tx, ty, 0, {}};
  hipStream_t s = (hipStream_t)stream;
#define CRW_MAP_CASE(CI, CO, MODE) \
  if (mode == MODE && cin == CI && cout == CO) return split == 3 ? launch_conv_map<3, CI, CO, MODE>(a, s) : launch_conv_map<1, CI, CO, MODE>(a, s);
  CRW_MAP_CASE(32, 64, 0)
  CRW_MAP_CASE(64, 128, 0)
  CRW_MAP_CASE(128, 128, 0)
  CRW_MAP_CASE(128, 128, 1)
  CRW_MAP_CASE(128, 64, 1)
  CRW_MAP_CASE(64, 32, 1)
#undef CRW_MAP_CASE
  return CRW_EINVAL;
}

int crw_enc_gap_bwd(const float *dgap, const uint16_t *y_hi, int P, int C, int npix, uint16_t *dy_hi, uint16_t *dy_lo,
                    crw_stream_t stream) {
  clear_stale_error();
  if (!dgap || !y_hi || !dy_hi || P < 1 || C < 8 || C % 8 || npix < 1) return CRW_EINVAL;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_grid((long)P * npix * C / 8)), dim3(256), 0, (hipStream_t)stream, dgap, y_hi, P,
                     C, npix, dy_hi, dy_lo);
  return check_launch();
}

int crw_enc_conv3x3(int mode, int split, int P, int cin, int cout, const uint16_t *x_hi, const uint16_t *x_lo,
                    const uint16_t *w_hi, const uint16_t *w_lo, const float *bias, const uint16_t *mask_hi,
                    uint16_t *y_hi, uint16_t *y_lo, float *y_f32, float *gap, const float *dgap, crw_stream_t stream) {
  clear_stale_error();
  if (!x_hi || !w_hi || P < 1 || (mode != 0 && mode != 1) || (split != 1 && split != 3)) return CRW_EINVAL;
  if (split == 3 && (!w_lo || (!x_lo && !dgap))) return CRW_EINVAL;
  if (dgap && mode != 1) return CRW_EINVAL;
  if (!y_hi && !y_f32 && !gap) return CRW_EINVAL;
  ConvArgs a{x_hi, x_lo, w_hi, w_lo, bias, mask_hi, y_hi, y_lo, y_f32, gap, dgap, P, g_conv_stamps, 0, 0, 1, 1};
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

static int wgrad_groups(int cin) { return cin / wgrad_nci(cin); }

static int wgrad_slices(int P, int cin, int cout, int split) {
  (void)split;
  const int groups = wgrad_groups(cin) * (cout / WG_NCO);  // workgroups per slice
  int n = 256 * 2 / (groups < 1 ? 1 : groups);             // two workgroups per CU on the 256 CUs of an MI355X
  return n > P ? P : n;
}

size_t crw_enc_wgrad_ws_bytes(int P, int cin, int cout, int split) {
  if (P < 1 || cin < 1 || cout < 1) return 0;
  return (size_t)wgrad_slices(P, cin, cout, split) *
         ((size_t)cout * cin * 9 + (size_t)wgrad_groups(cin) * cout) * sizeof(float);
}

int crw_enc_conv3x3_wgrad(int split, int P, int cin, int cout, const uint16_t *dy_hi, const uint16_t *dy_lo,
                          const uint16_t *x_hi, const uint16_t *x_lo, const float *dgap, float *dw, float *db, void *ws,
                          size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!dy_hi || !x_hi || !dw || !db || !ws || P < 1 || (split != 1 && split != 3)) return CRW_EINVAL;
  if (split == 3 && ((!dy_lo && !dgap) || !x_lo)) return CRW_EINVAL;
  if (ws_bytes < crw_enc_wgrad_ws_bytes(P, cin, cout, split)) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int nslice = wgrad_slices(P, cin, cout, split);
  // the streamed kernel (one 8-wave workgroup per CU) for the layers with 128 output channels; CRW_WGRAD=1 forces the
  // first-generation kernel (diagnostics / A-B)
  static const char *force = getenv("CRW_WGRAD");
  const bool streamed = cout == W2_NCO && cin % W2_NCI == 0 && !(force && force[0] == '1');
  if (streamed) {
    const int n2 = 256 / (cin / W2_NCI);
    nslice = n2 > P ? P : n2;
  }
  int ppb = (P + nslice - 1) / nslice;
  if (streamed) nslice = (P + ppb - 1) / ppb;  // no empty slices: every slab is written
  float *dw_part = static_cast<float *>(ws), *db_part = dw_part + (size_t)nslice * cout * cin * 9;
  WgradArgs a{dy_hi, dy_lo, x_hi, x_lo, dgap, dw_part, db_part, P, ppb, nslice % 8 == 0 ? 1 : 0, g_conv_stamps, 0, 0, 1, 1};
  int st = CRW_EINVAL;
  if (streamed) {
    if (cin == 64) st = split == 3 ? launch_wgrad2<3, 64, 128>(a, nslice, s) : launch_wgrad2<1, 64, 128>(a, nslice, s);
    else if (cin == 128) {
#ifdef CRW_CONV_STAMPS
      static const char *diag = getenv("CRW_WGRAD_DIAG");
      if (diag && split == 3 && diag[0] == '1') return launch_wgrad2<3, 128, 128, 1>(a, nslice, s);
      if (diag && split == 3 && diag[0] == '2') return launch_wgrad2<3, 128, 128, 2>(a, nslice, s);
      if (diag && split == 3 && diag[0] == '3') return launch_wgrad2<3, 128, 128, 3>(a, nslice, s);
      if (diag && split == 3 && diag[0] == '4') return launch_wgrad2<3, 128, 128, 4>(a, nslice, s);
      if (diag && split == 3 && diag[0] == '5') return launch_wgrad2<3, 128, 128, 5>(a, nslice, s);
      if (diag && split == 3 && diag[0] == '6') return launch_wgrad2<3, 128, 128, 6>(a, nslice, s);
#endif
      st = split == 3 ? launch_wgrad2<3, 128, 128>(a, nslice, s) : launch_wgrad2<1, 128, 128>(a, nslice, s);
    }
  } else {
#define CRW_WG_CASE(CI, CO)                                                                  \
  if (cin == CI && cout == CO) st = split == 3 ? launch_wgrad<3, CI, CO>(a, nslice, s) : launch_wgrad<1, CI, CO>(a, nslice, s);
  CRW_WG_CASE(32, 64)
  CRW_WG_CASE(64, 128)
  CRW_WG_CASE(128, 128)
#undef CRW_WG_CASE
  }
  if (st != CRW_OK) return st;
  const long nw = (long)cout * cin * 9;
  const int nblk_dw = (int)((nw + 63) / 64);
  hipLaunchKernelGGL(slice_sum_dw_kernel, dim3((unsigned)(nblk_dw + (cout + 3) / 4)), dim3(256), 0, s, dw_part, nslice, cout * cin,
                     dw, nblk_dw, db_part, cout, db);
  return check_launch();
}


/* weight / bias gradient of one 3x3 layer on feature maps of any size: dY [P][H][W][cout], X [P][H][W][cin] planes.
 * ws: crw_enc_wgrad_ws_bytes(P * ceil(H/10) * ceil(W/10), cin, cout, split). */
int crw_enc_conv3x3_wgrad_map(int split, int P, int H, int W, int cin, int cout, const uint16_t *dy_hi, const uint16_t *dy_lo,
                              const uint16_t *x_hi, const uint16_t *x_lo, float *dw, float *db, void *ws, size_t ws_bytes,
                              crw_stream_t stream) {
  clear_stale_error();
  if (!dy_hi || !x_hi || !dw || !db || !ws || P < 1 || H < 1 || W < 1 || (split != 1 && split != 3)) return CRW_EINVAL;
  if (split == 3 && (!dy_lo || !x_lo)) return CRW_EINVAL;
  const int tx = (W + IMG_W - 1) / IMG_W, ty = (H + IMG_W - 1) / IMG_W;
  const long units_l = (long)P * tx * ty;
  if (units_l > 0x7fffffffL) return CRW_EINVAL;
  const int units = (int)units_l;
  if (ws_bytes < crw_enc_wgrad_ws_bytes(units, cin, cout, split)) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int nslice = wgrad_slices(units, cin, cout, split);
  const int upb = (units + nslice - 1) / nslice;
  nslice = (units + upb - 1) / upb;  // no empty slices: every slab is written
  float *dw_part = static_cast<float *>(ws), *db_part = dw_part + (size_t)nslice * cout * cin * 9;
  WgradArgs a{dy_hi, dy_lo, x_hi, x_lo, nullptr, dw_part, db_part, units, upb, nslice % 8 == 0 ? 1 : 0, nullptr, H, W, tx, ty};
  int st = CRW_EINVAL;
#define CRW_WGM_CASE(CI, CO) \
  if (cin == CI && cout == CO) st = split == 3 ? launch_wgrad_map<3, CI, CO>(a, nslice, s) : launch_wgrad_map<1, CI, CO>(a, nslice, s);
  CRW_WGM_CASE(32, 64)
  CRW_WGM_CASE(64, 128)
  CRW_WGM_CASE(128, 128)
#undef CRW_WGM_CASE
  if (st != CRW_OK) return st;
  const long nw = (long)cout * cin * 9;
  const int nblk_dw = (int)((nw + 63) / 64);
  hipLaunchKernelGGL(slice_sum_dw_kernel, dim3((unsigned)(nblk_dw + (cout + 3) / 4)), dim3(256), 0, s, dw_part, nslice, cout * cin,
                     dw, nblk_dw, db_part, cout, db);
  return check_launch();
}

}  // extern "C"
