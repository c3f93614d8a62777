// The stem convolution of the Resnet encoder for 16x16 patches (reference src/encoder.py:66-74 fc0/bn0/relu0, :185 model.conv1
// 7x7 / stride 2 / padding 3, 3 -> 64 channels), forward, weight gradient and backward-data, one PATCH at a time per wave.
//
// The generic gathered matrix product (resnet_gemm.hip modes 2 / 3) runs this layer across patches too, but its operands there
// are 64-byte row pieces of a 4.6 KB map record (stem forward / weight gradient: 0.09-0.19 of the bf16 roof executed, r03) or a
// Toeplitz expansion that multiplies mostly zeros (backward-data).  Here a wave owns a whole patch:
//   * the 18x18x3 map relu0(bn0(fc0(x))) is built from the 1 KB patch itself into an LDS image (24x24 pixels, zero-padded by 3,
//     4 channels of bf16 hi / lo: 8 bytes per pixel and plane), so the im2col gather is an LDS read: fragment k = (ky, kx, c) with
//     k-step = one kernel row (8 pixels x 4 channels = 32, the 8th pixel and the 4th channel carry zero weights);
//   * forward   Z1[81 x 64]  = col[81 x 224] W[224 x 64]      A = col from the map (ds_read_b128), B = weights resident in LDS
//   * weights   dW[224 x 64] = col^T dZ1[81 x 64]             A = col^T by ds_read_b64_tr_b16 on the map, B = dZ1 image (tr reads)
//   * data      G[81 x 224]  = dZ1 W^T, then col2im + relu0 / bn0 / fc0 backward sums     A = dZ1 fragments straight from HBM
//     (each element belongs to exactly one fragment), B = W^T resident in LDS; G goes through a per-wave LDS strip, one kernel
//     row at a time, and is gathered into the 18x18x3 gradient in registers (ds_add_f32 into an LDS image measured 5x slower)
//     -- the gradient map never reaches HBM: only the 12 sums that bn0 / fc0 need leave the kernel.
// 504 MFMAs per patch and pass (6 x 4 x 7 x 3 | 14 x 4 x 3 x 3 | 6 x 14 x 2 x 3), hi / lo operand pairs as everywhere.
#include <cstdlib>

#include "crw_common.h"
#include "resnet.h"

namespace crw {
namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char *lds_cp;

constexpr int MAPW = 24, MAP_PLANE = MAPW * MAPW * 8;  // bytes of one plane of the map image
constexpr int WFRAG_BYTES = 7 * 4 * 2 * 1024;          // weights in fragment order: 7 k-steps x 4 tiles x 2 planes x 1 KB

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ inline f32x4 mfma(bf8 a, bf8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ inline uint32_t lds_addr_of(const char *p) { return (uint32_t)(uintptr_t)(lds_cp)p; }
__device__ inline s4v tr_read(uint32_t lds_addr) {
  s4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
__device__ inline bf8 join(s4v lo, s4v hi) {
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8, v);
}

struct StemCoef {  // relu0(bn0(fc0(x))) = relu(a[c][0] x0 + a[c][1] x1 + d[c])
  float a[3][2], d[3];
};
__device__ inline StemCoef load_coef(const float *__restrict__ stem) {
  StemCoef k;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    k.a[c][0] = stem[12 + 2 * c];
    k.a[c][1] = stem[13 + 2 * c];
    k.d[c] = stem[18 + c];
  }
  return k;
}
// one map pixel: 3 channels (+ a zero) as bf16 hi and lo words
__device__ inline void stem_pixel(const StemCoef &k, float x0, float x1, uint2 &hi, uint2 &lo) {
  uint16_t h[3], l[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = fmaxf(k.a[c][0] * x0 + k.a[c][1] * x1 + k.d[c], 0.f);
    h[c] = f2bf(v);
    l[c] = f2bf(v - bf2f(h[c]));
  }
  hi = uint2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2]};
  lo = uint2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2]};
}
// everything of the map image that does not depend on the patch: zeros, and the ring where fc0 sees only its bias (x = 0)
__device__ inline void map_init(char *map, const StemCoef &k, int lane, int nlanes) {
  uint2 rh, rl;
  stem_pixel(k, 0.f, 0.f, rh, rl);
  for (int i = lane; i < MAPW * MAPW; i += nlanes) {
    const int y = i / MAPW, x = i % MAPW;
    const bool ring = y >= 3 && y <= 20 && x >= 3 && x <= 20;  // the 18 x 18 map of fc0 (interior rewritten per patch)
    *reinterpret_cast<uint2 *>(map + i * 8) = ring ? rh : uint2{0u, 0u};
    *reinterpret_cast<uint2 *>(map + MAP_PLANE + i * 8) = ring ? rl : uint2{0u, 0u};
  }
}
// NPX consecutive pixels of patch row y starting at column x0 -> map image (interior offset 4 = padding 3 + fc0's padding 1)
template <int CIN, int NPX>
__device__ inline void map_fill(char *map, const StemCoef &k, const float *__restrict__ xp, int y, int x0) {
  float xa[NPX], xb[NPX];
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    xa[i] = xp[y * 16 + x0 + i];
    xb[i] = CIN == 2 ? xp[256 + y * 16 + x0 + i] : 0.f;
  }
  char *dst = map + ((y + 4) * MAPW + x0 + 4) * 8;
#pragma unroll
  for (int i = 0; i < NPX; i += 2) {
    uint2 h0, l0, h1, l1;
    stem_pixel(k, xa[i], xb[i], h0, l0);
    stem_pixel(k, xa[i + 1], xb[i + 1], h1, l1);
    *reinterpret_cast<uint4 *>(dst + i * 8) = uint4{h0.x, h0.y, h1.x, h1.y};
    *reinterpret_cast<uint4 *>(dst + MAP_PLANE + i * 8) = uint4{l0.x, l0.y, l1.x, l1.y};
  }
}

// ================================================================================================ forward
// 8 waves per workgroup, one patch per wave and iteration.  LDS: weights (56 KB) + 8 map images (9 KB each).
template <int CIN>
__global__ __launch_bounds__(512) void rn_stem_fwd_kernel(const float *__restrict__ x, const float *__restrict__ stem,
                                                          const uint16_t *__restrict__ wfrag, int P, float *__restrict__ Z1,
                                                          float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char *wl = lds;
  char *map = lds + WFRAG_BYTES + wave * 2 * MAP_PLANE;
  const StemCoef k = load_coef(stem);
  for (int i = threadIdx.x; i < WFRAG_BYTES / 16; i += 512)
    reinterpret_cast<uint4 *>(wl)[i] = reinterpret_cast<const uint4 *>(wfrag)[i];
  map_init(map, k, lane, 64);
  __syncthreads();

  // byte offset of this lane's fragment rows inside the map: row tile i, pixel o = 16 i + (lane & 15), column group lane >> 4
  int abase[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int o = 16 * i + (lane & 15);
    if (o > 80) o = 80;  // rows 81..95 of the last tile: computed, never stored
    abase[i] = ((2 * (o / 9)) * MAPW + 2 * (o % 9) + 2 * (lane >> 4)) * 8;
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int gw = blockIdx.x * 8 + wave, tw = gridDim.x * 8;
  for (int p = gw; p < P; p += tw) {
    map_fill<CIN, 4>(map, k, x + (long)p * CIN * 256, lane >> 2, (lane & 3) * 4);
    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      bf8 b[4], bl[4], a[6], al[6];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        b[j] = *reinterpret_cast<const bf8 *>(wl + ((ky * 4 + j) * 2 + 0) * 1024 + lane * 16);
        bl[j] = *reinterpret_cast<const bf8 *>(wl + ((ky * 4 + j) * 2 + 1) * 1024 + lane * 16);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        a[i] = *reinterpret_cast<const bf8 *>(map + abase[i] + ky * (MAPW * 8));
        al[i] = *reinterpret_cast<const bf8 *>(map + MAP_PLANE + abase[i] + ky * (MAPW * 8));
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mfma(al[i], b[j], acc[i][j]);
          acc[i][j] = mfma(a[i], bl[j], acc[i][j]);
          acc[i][j] = mfma(a[i], b[j], acc[i][j]);
        }
    }
    float *zp = Z1 + (long)p * 81 * 64;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * i + (lane >> 4) * 4 + r;
        if (o < 81)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[i][j][r];
            zp[o * 64 + 16 * j + (lane & 15)] = v;
            s1[j] += v;
            s2[j] += v * v;
          }
      }
  }
  // BatchNorm statistics of this wave's patches: one partial row per wave
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = s1[j], b = s2[j];
    a += __shfl_xor(a, 16);
    b += __shfl_xor(b, 16);
    a += __shfl_xor(a, 32);
    b += __shfl_xor(b, 32);
    if (lane < 16) reinterpret_cast<float2 *>(part)[(long)gw * 64 + 16 * j + lane] = float2{a, b};
  }
}

// ================================================================================================ forward, any patch size
// The same product for patches of any size (32 x 32 at BASELINE config 5: a 17 x 17 x 64 output map per patch, where the gathered
// product of resnet_gemm.hip stages 5.3 GB of 64-byte row pieces for it: 0.71 ms of a 4.6 ms pass).  A wave owns a BAND of RB output
// rows of a patch (at most BAND_MT row tiles of 16 output pixels); the map rows the band reads -- 2 RB + 5 of them, MW pixels wide --
// are rebuilt from the patch into the wave's own LDS image exactly like above (interior: relu0(bn0(fc0(x))); the ring where fc0
// sees only its bias; zeros for the convolution's own padding), the im2col gather is an LDS read, the weights stay resident in LDS.
// Z1 [P][H1 * W1][64] and the per-wave BatchNorm partial sums leave the kernel as in rn_stem_fwd_kernel.
constexpr int BAND_MT = 6;
struct StemBand {
  int h, w, H1, W1;  // patch size, output map of the 7x7/2 convolution
  int RB, NB;        // output rows per band, bands per patch
  int MW, MR;        // band image: MR = 2 RB + 5 map rows of MW pixels (MW even: every fragment read stays 16-byte aligned)
};

template <int CIN>
__global__ __launch_bounds__(512) void rn_stem_fwd_band_kernel(const float *__restrict__ x, const float *__restrict__ stem,
                                                               const uint16_t *__restrict__ wfrag, int P, StemBand g,
                                                               float *__restrict__ Z1, float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int plane = g.MR * g.MW * 8;
  char *wl = lds;
  char *map = lds + WFRAG_BYTES + wave * 2 * plane;
  const StemCoef k = load_coef(stem);
  for (int i = threadIdx.x; i < WFRAG_BYTES / 16; i += 512)
    reinterpret_cast<uint4 *>(wl)[i] = reinterpret_cast<const uint4 *>(wfrag)[i];
  __syncthreads();
  uint2 ring_h, ring_l;
  stem_pixel(k, 0.f, 0.f, ring_h, ring_l);
  constexpr int FILL_G = 3;
  const int npair = g.MW >> 1, rpi = 64 / npair;               // pixel pairs per image row; image rows a wave fills per pass
  const int rl = lane / npair, cp = lane - rl * npair;
  const bool active = rl < rpi;

  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int gw = blockIdx.x * 8 + wave, tw = gridDim.x * 8;
  // a wave takes whole patches, band after band: every wave the same mix of full and short bands (bands dealt out one by one
  // gave a wave the SAME band of every patch whenever the wave count is a multiple of NB -- the short last band's waves idled)
  for (int p = gw; p < P; p += tw)
  for (int b = 0; b < g.NB; ++b) {  // (wave-uniform)
    const int oy0 = b * g.RB, rb = min(g.RB, g.H1 - oy0), npx = rb * g.W1, nt = (npx + 15) >> 4, nr = 2 * rb + 5;
    // band image: map row 2 oy0 + r, r < nr.  Map pixel (my, mx) is fc0-map pixel (my - 3, mx - 3) = patch pixel (my - 4, mx - 4)
    const float *xp = x + (long)p * CIN * g.h * g.w;
    // a lane owns a pixel PAIR of the image row (column pair cp, 16 bytes per plane) in every rpi-th row: no division in the
    // loop, the patch values of FILL_G rows requested together
    for (int r0 = rl; r0 < nr; r0 += FILL_G * rpi) {
      float xa[FILL_G][2], xb[FILL_G][2];
#pragma unroll
      for (int t = 0; t < FILL_G; ++t) {
        const int iy = 2 * oy0 + r0 + t * rpi - 4;
        const bool rowin = (unsigned)iy < (unsigned)g.h;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ix = 2 * cp + e - 4;
          const bool inside = active && rowin && (unsigned)ix < (unsigned)g.w;
          const int off = inside ? iy * g.w + ix : 0;  // (clamped, unconditional loads)
          xa[t][e] = xp[off];
          xb[t][e] = CIN == 2 ? xp[g.h * g.w + off] : 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < FILL_G; ++t) {
        const int r = r0 + t * rpi, my = 2 * oy0 + r, iy = my - 4;
        if (active && r < nr) {
          uint2 hi[2], lo[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int mx = 2 * cp + e, ix = mx - 4;
            const bool inside = (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
            const bool ring = my >= 3 && my < g.h + 5 && mx >= 3 && mx < g.w + 5;
            stem_pixel(k, xa[t][e], xb[t][e], hi[e], lo[e]);
            if (!inside) {
              hi[e] = ring ? ring_h : uint2{0u, 0u};
              lo[e] = ring ? ring_l : uint2{0u, 0u};
            }
          }
          char *dst = map + (r * g.MW + 2 * cp) * 8;
          *reinterpret_cast<uint4 *>(dst) = uint4{hi[0].x, hi[0].y, hi[1].x, hi[1].y};
          *reinterpret_cast<uint4 *>(dst + plane) = uint4{lo[0].x, lo[0].y, lo[1].x, lo[1].y};
        }
      }
    }
    // this lane's fragment rows: row tile i, band pixel o = 16 i + (lane & 15), column group lane >> 4
    int abase[BAND_MT];
#pragma unroll
    for (int i = 0; i < BAND_MT; ++i) {
      const int o = min(16 * i + (lane & 15), npx - 1);  // rows past the band: computed, never stored
      const int oy = o / g.W1, ox = o - oy * g.W1;
      abase[i] = ((2 * oy) * g.MW + 2 * ox + 2 * (lane >> 4)) * 8;
    }
    f32x4 acc[BAND_MT][4];
#pragma unroll
    for (int i = 0; i < BAND_MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rowb = g.MW * 8;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      bf8 bq[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bq[j] = *reinterpret_cast<const bf8 *>(wl + ((ky * 4 + j) * 2 + 0) * 1024 + lane * 16);
        bl[j] = *reinterpret_cast<const bf8 *>(wl + ((ky * 4 + j) * 2 + 1) * 1024 + lane * 16);
      }
#pragma unroll
      for (int i0 = 0; i0 < BAND_MT; i0 += 3) {  // three row tiles' fragments at a time (all six: past the register file)
        bf8 a[3], al[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {  // (tiles past nt read the band's last pixel: valid addresses)
          a[i] = *reinterpret_cast<const bf8 *>(map + abase[i0 + i] + ky * rowb);
          al[i] = *reinterpret_cast<const bf8 *>(map + plane + abase[i0 + i] + ky * rowb);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i0 + i < nt) {  // scalar branch: the matrix instructions ignore EXEC
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[i0 + i][j] = mfma(al[i], bq[j], acc[i0 + i][j]);
              acc[i0 + i][j] = mfma(a[i], bl[j], acc[i0 + i][j]);
              acc[i0 + i][j] = mfma(a[i], bq[j], acc[i0 + i][j]);
            }
          }
      }
    }
    float *zp = Z1 + ((long)p * g.H1 * g.W1 + (long)oy0 * g.W1) * 64;
#pragma unroll
    for (int i = 0; i < BAND_MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * i + (lane >> 4) * 4 + r;
        if (o < npx)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[i][j][r];
            zp[o * 64 + 16 * j + (lane & 15)] = v;
            s1[j] += v;
            s2[j] += v * v;
          }
      }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = s1[j], b = s2[j];
    a += __shfl_xor(a, 16);
    b += __shfl_xor(b, 16);
    a += __shfl_xor(a, 32);
    b += __shfl_xor(b, 32);
    if (lane < 16) reinterpret_cast<float2 *>(part)[(long)gw * 64 + 16 * j + lane] = float2{a, b};
  }
}

// ================================================================================================ weight gradient
// 8 waves = 4 pairs; a pair shares one patch (map image + dZ1 image) and splits the 14 row tiles of dW^T [224][64].
constexpr int DZ_PLANE = 96 * 128;  // dZ1 image: 96 rows (81 used, the rest zero) x 64 channels of bf16
__device__ inline int dz_sw(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }  // = rc_sw<64> of resnet_gemm.hip

template <int CIN>
__global__ __launch_bounds__(512) void rn_stem_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ stem,
                                                            const uint16_t *__restrict__ dz_hi, const uint16_t *__restrict__ dz_lo, int P,
                                                            float *__restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave >> 1, half = wave & 1;
  const int l2 = half * 64 + lane;  // lane within the pair
  char *map = lds + pair * (2 * MAP_PLANE + 2 * DZ_PLANE);
  char *dzi = map + 2 * MAP_PLANE;
  const StemCoef k = load_coef(stem);
  map_init(map, k, l2, 128);
  for (int i = l2; i < 2 * DZ_PLANE / 16; i += 128) reinterpret_cast<uint4 *>(dzi)[i] = uint4{0u, 0u, 0u, 0u};

  // tr-read geometry: k-row = output pixel o = 32 s + 8 g + q (+ 4), r = 16 consecutive col-entries = 4 map pixels
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  auto pixel = [&](int o) {  // byte offset of output pixel o's window corner in the map (+ this lane's pixel of the 4-pixel group)
    if (o > 80) o = 80;      // the dZ1 rows beyond 80 are zero
    return ((2 * (o / 9)) * MAPW + 2 * (o % 9)) * 8 + 8 * p4;
  };
  f32x4 acc[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int stride = gridDim.x * 4;
  const int iters = (P + stride - 1) / stride;
  for (int it = 0; it < iters; ++it) {
    const int p = blockIdx.x * 4 + pair + it * stride;
    const bool valid = p < P;
    __syncthreads();  // the previous patch's fragments have been read
    if (valid) {
      map_fill<CIN, 2>(map, k, x + (long)p * CIN * 256, l2 >> 3, (l2 & 7) * 2);
      const long src = (long)p * 81 * 64;
      for (int c = l2; c < 81 * 8; c += 128) {  // 16-byte chunks of the 81 x 64 planes
        const int row = c >> 3, ch = c & 7;
        const int dst = row * 128 + ((ch ^ dz_sw(row)) << 4);
        *reinterpret_cast<uint4 *>(dzi + dst) = *reinterpret_cast<const uint4 *>(dz_hi + src + c * 8);
        *reinterpret_cast<uint4 *>(dzi + DZ_PLANE + dst) = *reinterpret_cast<const uint4 *>(dz_lo + src + c * 8);
      }
    }
    __syncthreads();
    if (!valid) continue;
    const uint32_t mbase = lds_addr_of(map), dbase = lds_addr_of(dzi);
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
      const int pix0 = pixel(32 * s + 8 * g + q), pix1 = pixel(32 * s + 8 * g + q + 4);
      bf8 b[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 32 * s + 8 * g + q, chunk = 2 * j + (p4 >> 1);
        const uint32_t o0 = row * 128 + ((chunk ^ dz_sw(row)) << 4) + 8 * (p4 & 1);
        const uint32_t o1 = (row + 4) * 128 + ((chunk ^ dz_sw(row + 4)) << 4) + 8 * (p4 & 1);
        b[j] = join(tr_read(dbase + o0), tr_read(dbase + o1));
        bl[j] = join(tr_read(dbase + DZ_PLANE + o0), tr_read(dbase + DZ_PLANE + o1));
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int mt = 7 * half + i;  // row tile of dW^T: kernel row mt / 2, pixels 4 (mt % 2) .. + 3
        const uint32_t off = (mt >> 1) * (MAPW * 8) + (mt & 1) * 32;
        const bf8 a = join(tr_read(mbase + pix0 + off), tr_read(mbase + pix1 + off));
        const bf8 al = join(tr_read(mbase + MAP_PLANE + pix0 + off), tr_read(mbase + MAP_PLANE + pix1 + off));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mfma(al, b[j], acc[i][j]);
          acc[i][j] = mfma(a, bl[j], acc[i][j]);
          acc[i][j] = mfma(a, b[j], acc[i][j]);
        }
      }
    }
  }
  float *sl = slab + ((long)blockIdx.x * 4 + pair) * 224 * 64;
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sl[(112 * half + 16 * i + (lane >> 4) * 4 + r) * 64 + 16 * j + (lane & 15)] = acc[i][j][r];
}

// ================================================================================================ backward-data + stem sums
// 8 waves, one patch per wave and iteration.  LDS: W^T fragments (56 KB) + a [81][32 (+4 pad)] fp32 strip per wave (one kernel row of G).
// strip rows on a 36-float stride: the four row groups of an accumulator store land on disjoint banks and the col2im reads are at most
// 2-way conflicted (on the natural 32-float stride both were 4-way: PMC LDS conflict share 0.51)
constexpr int GS_LD = 36;
constexpr int GSTRIP = 81 * GS_LD * 4;

template <int CIN>
__global__ __launch_bounds__(512) void rn_stem_bwd_kernel(const float *__restrict__ x, const float *__restrict__ stem,
                                                          const float *__restrict__ w0, const float *__restrict__ b0,
                                                          const uint16_t *__restrict__ wtfrag, const uint16_t *__restrict__ dz_hi,
                                                          const uint16_t *__restrict__ dz_lo, int P, float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char *wl = lds;
  float *gs = reinterpret_cast<float *>(lds + WFRAG_BYTES + wave * GSTRIP);
  for (int i = threadIdx.x; i < WFRAG_BYTES / 16; i += 512)
    reinterpret_cast<uint4 *>(wl)[i] = reinterpret_cast<const uint4 *>(wtfrag)[i];
  __syncthreads();
  const StemCoef k = load_coef(stem);
  float wc[3][2], bc[3], mean[3], istd[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    wc[c][0] = w0[c * CIN];
    wc[c][1] = CIN == 2 ? w0[c * CIN + 1] : 0.f;
    bc[c] = b0[c];
    mean[c] = stem[6 + c];
    istd[c] = stem[9 + c];
  }
  float S[4] = {0.f, 0.f, 0.f, 0.f};
  // lane (< 54) owns column (ix, c) of the 18 x 18 x 3 gradient map; the (at most four) strip entries per output row that reach it
  const int ix = lane / 3, c = lane % 3;
  const bool owner = lane < 54;
  const int oxa = ix > 3 ? (ix - 2) >> 1 : 0, oxb = min(8, (ix + 3) >> 1);
  bool okj[4];
  int gofs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    okj[j] = owner && oxa + j <= oxb;
    gofs[j] = okj[j] ? (oxa + j) * GS_LD + (ix + 3 - 2 * (oxa + j)) * 4 + c : 0;
  }

  const int gw = blockIdx.x * 8 + wave, tw = gridDim.x * 8;
  for (int p = gw; p < P; p += tw) {
    float dxe[9], dxo[9];  // this lane's column of the gradient map: even rows 2 m, odd rows 2 m + 1
#pragma unroll
    for (int n = 0; n < 9; ++n) dxe[n] = dxo[n] = 0.f;
    const long src = (long)p * 81 * 64;
    // two passes of three row tiles (output pixels 0..47, 48..80): the dZ1 fragments of a pass stay in registers for all 7 kernel rows
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int olo = 48 * pass, ohi = pass ? 81 : 48;
      // row tile i, k-step s (32 channels), planes hi / lo -- 16 bytes per lane each; every element belongs to one fragment
      bf8 a[3][2], al[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        int o = olo + 16 * i + (lane & 15);
        if (o > 80) o = 80;  // rows of G beyond 80 are never read
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          a[i][s] = *reinterpret_cast<const bf8 *>(dz_hi + src + o * 64 + 32 * s + 8 * (lane >> 4));
          al[i][s] = *reinterpret_cast<const bf8 *>(dz_lo + src + o * 64 + 32 * s + 8 * (lane >> 4));
        }
      }
#pragma unroll 1
      for (int ky = 0; ky < 7; ++ky) {
        f32x4 acc[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const bf8 b = *reinterpret_cast<const bf8 *>(wl + (((ky * 2 + jj) * 2 + s) * 2 + 0) * 1024 + lane * 16);
            const bf8 bl = *reinterpret_cast<const bf8 *>(wl + (((ky * 2 + jj) * 2 + s) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              acc[i][jj] = mfma(al[i][s], b, acc[i][jj]);
              acc[i][jj] = mfma(a[i][s], bl, acc[i][jj]);
              acc[i][jj] = mfma(a[i][s], b, acc[i][jj]);
            }
          }
        // G of this kernel row (the pass's pixels) -> the wave's strip [o][kx * 4 + c]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = olo + 16 * i + (lane >> 4) * 4 + r;
            if (o < 81) {
              gs[o * GS_LD + (lane & 15)] = acc[i][0][r];
              gs[o * GS_LD + 16 + (lane & 15)] = acc[i][1][r];
            }
          }
        // col2im: output row oy adds to map row iy = 2 oy + ky - 3; a map column receives from at most four output columns
        // ox = oxa + j (kx = ix + 3 - 2 ox in [0, 6]).  All 36 strip reads of the kernel row are issued together (the row sums
        // tmp[oy]), then added to the rows they belong to: the shift between oy and iy is wave-uniform, one static case each.
        float tmp[9];
#pragma unroll
        for (int oy = 0; oy < 9; ++oy) {
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int o = oy * 9 + oxa + j;
            const bool ok = okj[j] && o >= olo && o < ohi;
            const float v = gs[ok ? oy * (9 * GS_LD) + gofs[j] : 0];
            sum += ok ? v : 0.f;
          }
          tmp[oy] = sum;
        }
        // iy = 2 oy + ky - 3: odd ky -> even rows 2 (oy + d), d = (ky - 3) / 2; even ky -> odd rows 2 (oy + d) + 1, d = (ky - 4) / 2
        if (ky & 1) {
          const int d = (ky - 3) / 2;
#pragma unroll
          for (int dd = -1; dd <= 1; ++dd)
            if (d == dd)
#pragma unroll
              for (int oy = 0; oy < 9; ++oy)
                if (oy + dd >= 0 && oy + dd < 9) dxe[oy + dd] += tmp[oy];
        } else {
          const int d = (ky - 4) / 2;
#pragma unroll
          for (int dd = -2; dd <= 1; ++dd)
            if (d == dd)
#pragma unroll
              for (int oy = 0; oy < 9; ++oy)
                if (oy + dd >= 0 && oy + dd < 9) dxo[oy + dd] += tmp[oy];
        }
      }
    }
    // relu0 / bn0 / fc0 backward sums (what rn_stem_bwd_reduce_kernel computes from a materialised gradient map); this lane's
    // channel c only: S[0..3] = sum g, sum g xhat, sum g x0, sum g x1
    if (owner) {
      const float *xp = x + (long)p * CIN * 256;
      const float a0 = c == 0 ? k.a[0][0] : c == 1 ? k.a[1][0] : k.a[2][0], a1 = c == 0 ? k.a[0][1] : c == 1 ? k.a[1][1] : k.a[2][1];
      const float dd = c == 0 ? k.d[0] : c == 1 ? k.d[1] : k.d[2];
      const float w_0 = c == 0 ? wc[0][0] : c == 1 ? wc[1][0] : wc[2][0], w_1 = c == 0 ? wc[0][1] : c == 1 ? wc[1][1] : wc[2][1];
      const float bb = c == 0 ? bc[0] : c == 1 ? bc[1] : bc[2], mm = c == 0 ? mean[0] : c == 1 ? mean[1] : mean[2];
      const float is = c == 0 ? istd[0] : c == 1 ? istd[1] : istd[2];
#pragma unroll
      for (int iy = 0; iy < 18; ++iy) {
        float xi0 = 0.f, xi1 = 0.f;
        if (iy >= 1 && iy <= 16 && ix >= 1 && ix <= 16) {
          xi0 = xp[(iy - 1) * 16 + ix - 1];
          if (CIN == 2) xi1 = xp[256 + (iy - 1) * 16 + ix - 1];
        }
        const float y = a0 * xi0 + a1 * xi1 + dd;
        const float gsel = y > 0.f ? ((iy & 1) ? dxo[iy >> 1] : dxe[iy >> 1]) : 0.f;
        const float xh = (w_0 * xi0 + w_1 * xi1 + bb - mm) * is;
        S[0] += gsel;
        S[1] += gsel * xh;
        S[2] += gsel * xi0;
        S[3] += gsel * xi1;
      }
    }
  }
  // per-wave partial [16]: channel c of lanes c, c+3, ... -> part[c * 4 + {0..3}]
  float T[12];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int j = 0; j < 4; ++j) T[cc * 4 + j] = wave_sum((lane < 54 && lane % 3 == cc) ? S[j] : 0.f);
  if (lane < 16) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) v = lane == i ? T[i] : v;
    part[(long)gw * 16 + lane] = v;
  }
}

// ================================================================================================ weight fragments
// w1 [64][3][7][7] fp32 -> (a) forward fragments [ky][j][plane][lane][8]: column co = 16 j + (lane & 15), k = 8 (lane >> 4) + e of
// kernel row ky: (kx, c) = (2 (lane >> 4) + e / 4, e % 4);  (b) transposed fragments [nt][s][plane][lane][8] for the backward-data
// product: column = col-entry 16 nt + (lane & 15) (ky = nt / 2, kx = (16 (nt % 2) + (lane & 15)) / 4, c = lane & 3), k = output
// channel 32 s + 8 (lane >> 4) + e.  Zero for kx = 7 and c = 3.
__global__ __launch_bounds__(256) void rn_pack_stem_frag_kernel(const float *__restrict__ w1, uint16_t *__restrict__ wf,
                                                                uint16_t *__restrict__ wt) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // one (fragment, lane, element) of one layout: 2 x 28 x 64 x 8
  if (i >= 2 * 28 * 512) return;
  const int which = i / (28 * 512), rem = i % (28 * 512);
  const int f = rem / 512, lane = (rem % 512) / 8, e = rem % 8;
  int co, ky, kx, c;
  if (which == 0) {
    ky = f / 4;
    co = 16 * (f % 4) + (lane & 15);
    kx = 2 * (lane >> 4) + e / 4;
    c = e % 4;
  } else {
    const int nt = f / 2, s = f % 2;
    ky = nt / 2;
    kx = (16 * (nt % 2) + (lane & 15)) / 4;
    c = lane & 3;
    co = 32 * s + 8 * (lane >> 4) + e;
  }
  const float v = (kx < 7 && c < 3) ? w1[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
  const uint16_t h = f2bf(v), l = f2bf(v - bf2f(h));
  uint16_t *dst = which == 0 ? wf : wt;
  dst[(f * 2 + 0) * 512 + lane * 8 + e] = h;
  dst[(f * 2 + 1) * 512 + lane * 8 + e] = l;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    g_last_hip_error = (int)hipGetLastError();
    return CRW_EHIP;
  }
  return CRW_OK;
}

}  // namespace

int rn_stem16_blocks() { return 256; }  // one 8-wave workgroup per CU

int launch_rn_pack_stem_frag(const float *w1, uint16_t *wf, uint16_t *wt, hipStream_t s) {
  hipLaunchKernelGGL(rn_pack_stem_frag_kernel, dim3((2 * 28 * 512 + 255) / 256), dim3(256), 0, s, w1, wf, wt);
  return check_launch();
}

// Z1 [P][81][64] fp32 (rows of patches beyond P are not written), part [blocks * 8][64] float2
int launch_rn_stem16_fwd(const float *x, int P, int cin, const float *stem, const uint16_t *wf, float *Z1, float *part, hipStream_t s) {
  const size_t lds = WFRAG_BYTES + 8 * 2 * MAP_PLANE;
  static bool set1 = false, set2 = false;
  if (cin == 1) {
    if (!set1) { CRW_TRY(set_lds(rn_stem_fwd_kernel<1>, lds)); set1 = true; }
    hipLaunchKernelGGL(rn_stem_fwd_kernel<1>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, wf, P, Z1, part);
  } else {
    if (!set2) { CRW_TRY(set_lds(rn_stem_fwd_kernel<2>, lds)); set2 = true; }
    hipLaunchKernelGGL(rn_stem_fwd_kernel<2>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, wf, P, Z1, part);
  }
  return check_launch();
}

// The band kernel's geometry for h x w patches; false: not covered (the gathered product of resnet_gemm.hip takes the layer)
static bool stem_band_geometry(int h, int w, StemBand &g) {
  g.h = h; g.w = w;
  g.H1 = (h + 2 + 6 - 7) / 2 + 1; g.W1 = (w + 2 + 6 - 7) / 2 + 1;
  if (h < 1 || w < 1 || g.W1 > 16 * BAND_MT || w + 9 > 128) return false;  // (an image row of at most 64 pixel pairs)
  g.RB = (16 * BAND_MT) / g.W1;
  if (g.RB > g.H1) g.RB = g.H1;
  g.NB = (g.H1 + g.RB - 1) / g.RB;
  g.MW = (w + 8 + 1) & ~1;
  g.MR = 2 * g.RB + 5;
  return WFRAG_BYTES + (size_t)8 * 2 * g.MR * g.MW * 8 <= (size_t)160 * 1024;
}
bool rn_stem_band_ok(int h, int w) {
  StemBand g;
  static const bool off = getenv("CRW_RN_STEM_BAND") && getenv("CRW_RN_STEM_BAND")[0] == '0';  // A/B: the gathered product
  return !off && stem_band_geometry(h, w, g);
}
// Z1 [P][H1 * W1][64] fp32, part [blocks * 8][64] float2 -- rn_stem_fwd_kernel's outputs for patches of any size
int launch_rn_stem_band_fwd(const float *x, int P, int cin, int h, int w, const float *stem, const uint16_t *wf, float *Z1, float *part,
                            hipStream_t s) {
  StemBand g;
  if (!stem_band_geometry(h, w, g) || (cin != 1 && cin != 2)) return CRW_EINVAL;
  const size_t lds = WFRAG_BYTES + (size_t)8 * 2 * g.MR * g.MW * 8;
  static size_t set1 = 0, set2 = 0;  // dynamic-LDS limit raised so far, per instantiation
  if (cin == 1) {
    if (set1 < lds) { CRW_TRY(set_lds(rn_stem_fwd_band_kernel<1>, (size_t)160 * 1024)); set1 = (size_t)160 * 1024; }
    hipLaunchKernelGGL(rn_stem_fwd_band_kernel<1>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, wf, P, g, Z1, part);
  } else {
    if (set2 < lds) { CRW_TRY(set_lds(rn_stem_fwd_band_kernel<2>, (size_t)160 * 1024)); set2 = (size_t)160 * 1024; }
    hipLaunchKernelGGL(rn_stem_fwd_band_kernel<2>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, wf, P, g, Z1, part);
  }
  return check_launch();
}

// slab [blocks * 4][224][64] fp32
int launch_rn_stem16_wgrad(const float *x, int P, int cin, const float *stem, const uint16_t *dz_hi, const uint16_t *dz_lo, float *slab,
                           hipStream_t s) {
  const size_t lds = 4 * (2 * MAP_PLANE + 2 * DZ_PLANE);
  static bool set1 = false, set2 = false;
  if (cin == 1) {
    if (!set1) { CRW_TRY(set_lds(rn_stem_wgrad_kernel<1>, lds)); set1 = true; }
    hipLaunchKernelGGL(rn_stem_wgrad_kernel<1>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, dz_hi, dz_lo, P, slab);
  } else {
    if (!set2) { CRW_TRY(set_lds(rn_stem_wgrad_kernel<2>, lds)); set2 = true; }
    hipLaunchKernelGGL(rn_stem_wgrad_kernel<2>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, dz_hi, dz_lo, P, slab);
  }
  return check_launch();
}

// part [blocks * 8][16] floats (c * 4 + {sum g, sum g xhat, sum g x0, sum g x1})
int launch_rn_stem16_bwd(const float *x, int P, int cin, const float *stem, const float *w0, const float *b0, const uint16_t *wt,
                         const uint16_t *dz_hi, const uint16_t *dz_lo, float *part, hipStream_t s) {
  const size_t lds = WFRAG_BYTES + 8 * GSTRIP;
  static bool set1 = false, set2 = false;
  if (cin == 1) {
    if (!set1) { CRW_TRY(set_lds(rn_stem_bwd_kernel<1>, lds)); set1 = true; }
    hipLaunchKernelGGL(rn_stem_bwd_kernel<1>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, w0, b0, wt, dz_hi, dz_lo, P, part);
  } else {
    if (!set2) { CRW_TRY(set_lds(rn_stem_bwd_kernel<2>, lds)); set2 = true; }
    hipLaunchKernelGGL(rn_stem_bwd_kernel<2>, dim3(rn_stem16_blocks()), dim3(512), lds, s, x, stem, w0, b0, wt, dz_hi, dz_lo, P, part);
  }
  return check_launch();
}

}  // namespace crw
