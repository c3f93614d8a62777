// Fused front end of the CNN encoder (src/encoder.py:13-24,46-47):
//     x [cin,16,16] -> conv1 5x5 pad 1 (cin->8) -> ReLU -> maxpool 2x2/1 -> conv2 5x5 pad 1 (8->32)
//       -> ReLU -> maxpool 2x2/1 -> [100][32] channels-last bf16 planes (hi[,lo]) for conv3.
// Persistent workgroups stream patches, the next patch prefetched into registers; everything between the 1 KB input patch
// and the 6.4 KB output planes stays in LDS.  Forward: two 512-thread workgroups per CU (60 KB of LDS each) that fill each
// other's barrier waits (0.27 -> 0.21 ms per 16128 patches against one 1024-thread workgroup); backward: one 1024-thread
// workgroup per CU (98 KB of LDS).
//   conv1 (25*cin MACs per output, 3 % of this stage): fp32 VALU, exact.
//   conv2 (200 MACs per output): implicit GEMM on v_mfma_f32_16x16x32_bf16 with k = (tap, ci): one
//     32-deep k-step = 4 taps x 8 input channels, so an A fragment is ONE 16-byte channels-last read
//     of the pixel shifted by the lane group's tap.  SPLIT = 3 uses hi/lo operand pairs (fp32-grade).
// The backward kernel recomputes this forward per patch (cheaper than storing c1/a1/c2 and the
// pooling indices: 30 KB per patch) and then runs, per patch,
//     pool2/ReLU2 backward -> conv2 weight gradient (MFMA, transposed LDS reads) and bias gradient
//     -> conv2 backward-data (MFMA) -> pool1/ReLU1 backward -> conv1 weight/bias gradient (VALU),
// accumulating the weight gradients in registers over its slice of patches; slices are added in a
// fixed order by slice_sum (deterministic).
#include "crw_common.h"
#include <type_traits>
#include <cstdlib>

namespace crw {
namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char *lds_cp;

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// Workgroup barrier that makes LDS traffic visible but does NOT drain outstanding global loads
// (__syncthreads() waits for vmcnt(0), which would expose the latency of the next patch's prefetch).
// Only for phases whose cross-wave communication goes through LDS.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// geometry
constexpr int XPW = 18;                    // 16x16 input patch, padded by 1
constexpr int C1W = 14, C1N = C1W * C1W;   // conv1 output 14x14
constexpr int A1W = 13, A1PW = 15;         // pool1 output 13x13, padded by 1 -> 15x15
constexpr int C2W = 11, C2N = C2W * C2W;   // conv2 output 11x11 (121)
constexpr int OW = 10, ON = 100;           // pool2 output
constexpr int D2PW = 19;                   // conv2-output gradient padded by 4 (for backward-data)
// Those planes (and the backward weights) are stored as two 16-channel halves of 32-byte rows: with 64-byte rows
// the ds_read_b128 lane groups of an MFMA fragment read (lanes {0-3,12-15,20-27}, ...) collide in the LDS banks;
// with [half][row][32 B] the 16 lanes of a group cover all 64 banks (MI355X_MICROARCH.md, LDS table).
constexpr int D2HALF = D2PW * D2PW * 32;   // bytes per half plane
constexpr int KS2 = 7;                     // conv2 k-steps (28 taps, 25 real)
constexpr int NTH = 1024, NWV = NTH / 64;  // 16 waves per workgroup: the stage is LDS-latency bound, one
                                           // 144 KB workgroup per CU, so latency is hidden by waves, not by workgroups

// What the forward pass keeps per patch for the backward pass when training (crw_enc_front_fwd `saved`), so that the
// backward kernel does not recompute conv1 -> pool1 -> conv2:
//   a1 hi / lo planes  [169 pixels][8 ch] bf16 (2704 B each)   the pool1 output = conv2's input (its weight gradient needs it)
//   k2 codes           [100 outputs][32 ch] 1 byte              pool2: argmax position (2 bits, first maximum in row-major
//   k1 codes           [169 outputs][8 ch]  1 byte (pad to 1360) order = torch's tie rule) | 4 if the maximum is > 0 (ReLU gate)
constexpr int SV_A1 = 169 * 16, SV_K2 = 100 * 32, SV_K1 = 1360;
constexpr int SV_OFF_A1L = SV_A1, SV_OFF_K2 = 2 * SV_A1, SV_OFF_K1 = 2 * SV_A1 + SV_K2, SV_BYTES = 2 * SV_A1 + SV_K2 + SV_K1;  // 9968
static_assert(SV_BYTES % 16 == 0 && SV_A1 % 16 == 0 && SV_K2 % 16 == 0, "16-byte chunks");

struct FrontArgs {
  const float *x;            // [P][cin][16][16]
  const float *w1, *b1;      // [8][cin][5][5], [8]
  const uint16_t *w2h, *w2l; // packed conv2 weights [7][32 co][32 k], k = 8*(tap%4) + ci, tap = 4*s + ..
  const float *b2;           // [32]
  uint16_t *yh, *yl;         // out planes [P][100][32]
  int P, cin;
  char *saved;               // optional [P][SV_BYTES] (training: see above)
};

// ---- shared forward pieces ----------------------------------------------------------------------
// c1r / c2r pixel index.  PAD = true (backward kernel): the maps carry a one-pixel border (never read by a valid
// pooling window, never initialised), so the pooling backward reads a pixel's 3x3 neighbourhood at constant offsets.
template <bool PAD>
__device__ inline int c1_idx(int y, int x) { return PAD ? (y + 1) * (C1W + 2) + x + 1 : y * C1W + x; }
template <bool PAD>
__device__ inline int c2_idx(int y, int x) { return PAD ? (y + 1) * (C2W + 2) + x + 1 : y * C2W + x; }

struct FwdLds {
  float *xs;      // [cin][18][18]
  float *w1;      // [cin][25][8 co] + b1[8]
  float *c1r;     // [196][8]
  char *a1h, *a1l;  // [225][8] bf16 (16 B per pixel)
  char *w2h, *w2l;  // [7][2 k halves][32 co][16 k] bf16 (w2_lds_chunk)
  float *c2r;     // [121][32]
};

// conv1 + bias + ReLU: thread = (output pixel, 4 of the 8 channels): 392 threads.  The LDS weights are laid out
// [ci][tap][8 co] so the four weights of a tap are one 16-byte broadcast read: 2 LDS reads per 4 FMAs.
template <bool PAD>
__device__ inline void conv1_relu(const FwdLds &L, int cin, int tid) {
  if (tid < C1N * 2) {
    const int pix = tid >> 1, c0 = 4 * (tid & 1);
    const int y = pix / C1W, xx = pix % C1W;
    float4 acc = *reinterpret_cast<const float4 *>(L.w1 + 8 * cin * 25 + c0);
    for (int ci = 0; ci < cin; ++ci) {
      const float *xs = L.xs + (ci * XPW + y) * XPW + xx;
      const float *w = L.w1 + ci * 25 * 8 + c0;
#pragma unroll
      for (int t = 0; t < 25; ++t) {
        const float v = xs[(t / 5) * XPW + t % 5];
        const float4 wv = *reinterpret_cast<const float4 *>(w + t * 8);
        acc.x = fmaf(v, wv.x, acc.x);
        acc.y = fmaf(v, wv.y, acc.y);
        acc.z = fmaf(v, wv.z, acc.z);
        acc.w = fmaf(v, wv.w, acc.w);
      }
    }
    *reinterpret_cast<float4 *>(L.c1r + c1_idx<PAD>(y, xx) * 8 + c0) = float4{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
  }
}

// maxpool 2x2/1 of c1r -> a1 planes (interior of the 15x15 padded image)
__device__ inline int argmax4(float a, float b, float c, float d);
template <int SPLIT, bool PAD, int NT = NTH>
__device__ inline void pool1(const FwdLds &L, int tid, char *sv = nullptr) {  // sv: this patch's saved record (or null)
  constexpr int ROW = (PAD ? C1W + 2 : C1W) * 8;
  for (int e = tid; e < A1W * A1W * 8; e += NT) {
    const int c = e & 7, p = e >> 3, y = p / A1W, x = p % A1W;
    const float *s = L.c1r + c1_idx<PAD>(y, x) * 8 + c;
    const float v = fmaxf(fmaxf(s[0], s[8]), fmaxf(s[ROW], s[ROW + 8]));
    const uint16_t h = f2bf(v), l = f2bf(v - bf2f(h));
    const int o = ((y + 1) * A1PW + x + 1) * 16 + 2 * c;
    *reinterpret_cast<uint16_t *>(L.a1h + o) = h;
    if (SPLIT == 3) *reinterpret_cast<uint16_t *>(L.a1l + o) = l;
    if (sv) {
      reinterpret_cast<uint16_t *>(sv)[e] = h;
      if (SPLIT == 3) reinterpret_cast<uint16_t *>(sv + SV_OFF_A1L)[e] = l;
      sv[SV_OFF_K1 + e] = (char)(argmax4(s[0], s[8], s[ROW], s[ROW + 8]) | (v > 0.f ? 4 : 0));
    }
  }
}

// conv2 forward weights in LDS: the packed global planes [7 k-steps][32 co][4 chunks of 8 k] are stored as
// [7][2 k halves][32 co][2 chunks] (32-byte rows): on 64-byte rows the ds_read_b128 lane groups of a B-fragment read
// collide 2-way, on the split layout they cover all 64 banks (same trick as the d2 planes).
__device__ inline int w2_lds_chunk(int e) {  // 16-byte chunk index: global -> LDS
  const int ch = e & 3, co = (e >> 2) & 31, s = e >> 7;
  return ((s * 2 + (ch >> 1)) * 32 + co) * 2 + (ch & 1);
}

// conv2 + bias + ReLU on the matrix cores -> c2r [121][32] fp32.  16 output tiles (8 row x 2 column): one per wave of a
// 1024-thread workgroup; a 512-thread workgroup gives each wave two row tiles of the same column tile, so the weight
// fragments are read once for both (6 LDS reads per 6 MFMAs instead of 8).
template <int SPLIT, bool PAD, int NT = NTH>
__device__ inline void conv2_relu(const FwdLds &L, float bias, int tid) {  // bias = b2[16 (wave & 1) + lane % 16]
  constexpr int NW = NT / 64, U = 16 / NW;
  static_assert(NW % 2 == 0 && 16 % NW == 0, "waves per workgroup");
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  const int j = wave & 1;
  int base[U];  // padded a1 pixel of tap (0,0)
  f32x4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int i0 = 16 * ((wave + NW * u) >> 1) + r16;
    if (i0 >= C2N) i0 = 0;
    base[u] = (i0 / C2W) * A1PW + (i0 % C2W);
    acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s = 0; s < KS2; ++s) {
    int tap = 4 * s + g;
    if (tap > 24) tap = 24;  // k-steps beyond tap 24 carry zero weights
    const int toff = (tap / 5) * A1PW + (tap % 5);
    const int o = (((s * 2 + (g >> 1)) * 32 + 16 * j + r16) * 2 + (g & 1)) * 16;
    const bf8 bh = *reinterpret_cast<const bf8 *>(L.w2h + o);
    bf8 bl;
    if (SPLIT == 3) bl = *reinterpret_cast<const bf8 *>(L.w2l + o);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bf8 ah = *reinterpret_cast<const bf8 *>(L.a1h + (base[u] + toff) * 16);
      if (SPLIT == 3) {
        const bf8 al = *reinterpret_cast<const bf8 *>(L.a1l + (base[u] + toff) * 16);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[u], 0, 0, 0);
      }
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[u], 0, 0, 0);
    }
  }
  const int co = 16 * j + r16;
  const float b = bias;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ((wave + NW * u) >> 1) + 4 * g + r;
      if (i < C2N) L.c2r[c2_idx<PAD>(i / C2W, i % C2W) * 32 + co] = fmaxf(acc[u][r] + b, 0.f);
    }
}

__device__ inline FwdLds carve_fwd(char *&p, int cin, bool pad = false) {  // pad: bordered c1r / c2r (backward kernel)
  FwdLds L;
  auto take = [&](size_t bytes) { char *r = p; p += (bytes + 15) & ~(size_t)15; return r; };
  L.xs = (float *)take(sizeof(float) * cin * XPW * XPW);
  L.w1 = (float *)take(sizeof(float) * (8 * cin * 25 + 8));
  L.c1r = (float *)take(sizeof(float) * (pad ? (C1W + 2) * (C1W + 2) : C1N) * 8);
  L.a1h = take(A1PW * A1PW * 16);
  L.a1l = take(A1PW * A1PW * 16);
  L.w2h = take(KS2 * 32 * 32 * 2);
  L.w2l = take(KS2 * 32 * 32 * 2);
  L.c2r = (float *)take(sizeof(float) * (pad ? (C2W + 2) * (C2W + 2) : C2N) * 32);
  return L;
}

template <int SPLIT, int NT = NTH>
__device__ inline void stage_constants(const FwdLds &L, const FrontArgs &a, int tid) {
  for (int e = tid; e < a.cin * XPW * XPW; e += NT) L.xs[e] = 0.f;  // zero border of the padded patch
  for (int e = tid; e < 8 * a.cin * 25; e += NT) {  // [co][ci][tap] -> [ci][tap][co]
    const int t = e % 25, ci = (e / 25) % a.cin, co = e / (25 * a.cin);
    L.w1[(ci * 25 + t) * 8 + co] = a.w1[e];
  }
  if (tid < 8) L.w1[8 * a.cin * 25 + tid] = a.b1[tid];
  for (int e = tid; e < A1PW * A1PW * 4; e += NT) {  // zero halo (and interior) of the a1 planes
    reinterpret_cast<uint32_t *>(L.a1h)[e] = 0;
    reinterpret_cast<uint32_t *>(L.a1l)[e] = 0;
  }
  for (int e = tid; e < KS2 * 32 * 32 / 8; e += NT) {
    const int d = w2_lds_chunk(e);
    reinterpret_cast<uint4 *>(L.w2h)[d] = reinterpret_cast<const uint4 *>(a.w2h)[e];
    if (SPLIT == 3) reinterpret_cast<uint4 *>(L.w2l)[d] = reinterpret_cast<const uint4 *>(a.w2l)[e];
  }
}

// NT = 512: two workgroups share a CU (60 KB of LDS and <= 128 registers each) and fill each other's barrier waits; at 1024 threads
// the 122 registers of the hi/lo instance admit one workgroup per CU only.
template <int SPLIT, int NT>
__global__ __launch_bounds__(NT) void front_fwd_kernel(FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *p = lds;
  const FwdLds L = carve_fwd(p, a.cin);
  const int tid = threadIdx.x;
  stage_constants<SPLIT, NT>(L, a, tid);
  __syncthreads();
  const float b2r = a.b2[16 * ((tid >> 6) & 1) + (tid & 15)];  // read once: the barriers in the loop are memory clobbers
  // the next patch (cin*256 <= 512 floats, one per thread) is fetched while the current one is processed
  const int nx = a.cin * 256;
  float x_r = blockIdx.x < a.P ? a.x[(long)blockIdx.x * nx + min(tid, nx - 1)] : 0.f;
  for (int pt = blockIdx.x; pt < a.P; pt += gridDim.x) {
    if (tid < nx) L.xs[((tid >> 8) * XPW + ((tid >> 4) & 15) + 1) * XPW + (tid & 15) + 1] = x_r;
    if (pt + (int)gridDim.x < a.P) x_r = a.x[(long)(pt + gridDim.x) * nx + min(tid, nx - 1)];
    lds_barrier();
    conv1_relu<false>(L, a.cin, tid);
    lds_barrier();
    char *sv = a.saved ? a.saved + (long)pt * SV_BYTES : nullptr;
    pool1<SPLIT, false, NT>(L, tid, sv);
    lds_barrier();
    conv2_relu<SPLIT, false, NT>(L, b2r, tid);
    lds_barrier();
    // maxpool 2x2/1 -> output planes [100][32]
    for (int e = tid; e < ON * 32; e += NT) {
      const int c = e & 31, q = e >> 5, y = q / OW, x = q % OW;
      const float *s = L.c2r + (y * C2W + x) * 32 + c;
      const float v = fmaxf(fmaxf(s[0], s[32]), fmaxf(s[C2W * 32], s[C2W * 32 + 32]));
      const uint16_t h = f2bf(v);
      a.yh[(long)pt * ON * 32 + e] = h;
      if (SPLIT == 3) a.yl[(long)pt * ON * 32 + e] = f2bf(v - bf2f(h));
      if (sv) sv[SV_OFF_K2 + e] = (char)(argmax4(s[0], s[32], s[C2W * 32], s[C2W * 32 + 32]) | (v > 0.f ? 4 : 0));
    }
    // the next iteration's first barrier orders these c2r reads before conv2_relu overwrites it
  }
}

// ---- the same front end on patches of any size (inference) -------------------------------------------------------
// A work item = (patch, 10x10 tile of the pool2 output map [H-6][W-6]).  The tile's receptive field is a 20x20
// window of the patch (zeros outside it): conv1 over 16x16 positions, pool1 over the 15x15 a1 window that conv2
// needs -- where an a1 position lies outside the a1 map [H-3][W-3] it holds the zero padding of conv2 instead --
// then conv2 / pool2 exactly as in the 16x16 kernel.  Output: planes [P][(H-6)*(W-6)][32] for crw_enc_conv3x3_map.
constexpr int MXW = 20, MC1W = 16;  // window / conv1 extent of a tile

struct FrontMapArgs {
  FrontArgs f;     // x: [P][cin][H][W]; yh/yl: [P][(H-6)*(W-6)][32]
  int H, W, tiles_x, tiles_y;
};

template <int SPLIT, int NT>
__global__ __launch_bounds__(NT) void front_fwd_map_kernel(FrontMapArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int cin = a.f.cin, tid = threadIdx.x;
  char *p = lds;
  auto take = [&](size_t bytes) { char *r = p; p += (bytes + 15) & ~(size_t)15; return r; };
  FwdLds L;
  L.xs = (float *)take(sizeof(float) * cin * MXW * MXW);
  L.w1 = (float *)take(sizeof(float) * (8 * cin * 25 + 8));
  L.c1r = (float *)take(sizeof(float) * MC1W * MC1W * 8);
  L.a1h = take(A1PW * A1PW * 16);
  L.a1l = take(A1PW * A1PW * 16);
  L.w2h = take(KS2 * 32 * 32 * 2);
  L.w2l = take(KS2 * 32 * 32 * 2);
  L.c2r = (float *)take(sizeof(float) * C2N * 32);
  for (int e = tid; e < 8 * cin * 25; e += NT) {  // [co][ci][tap] -> [ci][tap][co]
    const int t = e % 25, ci = (e / 25) % cin, co = e / (25 * cin);
    L.w1[(ci * 25 + t) * 8 + co] = a.f.w1[e];
  }
  if (tid < 8) L.w1[8 * cin * 25 + tid] = a.f.b1[tid];
  for (int e = tid; e < KS2 * 32 * 32 / 8; e += NT) {
    const int d = w2_lds_chunk(e);
    reinterpret_cast<uint4 *>(L.w2h)[d] = reinterpret_cast<const uint4 *>(a.f.w2h)[e];
    if (SPLIT == 3) reinterpret_cast<uint4 *>(L.w2l)[d] = reinterpret_cast<const uint4 *>(a.f.w2l)[e];
  }
  const float b2r = a.f.b2[16 * ((tid >> 6) & 1) + (tid & 15)];
  const int H = a.H, W = a.W, Ho = H - 6, Wo = W - 6, ntile = a.tiles_x * a.tiles_y;
  const long nwork = (long)a.f.P * ntile;
  __syncthreads();
  for (long wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
    const int pt = (int)(wk / ntile), tile = (int)(wk % ntile);
    const int oy0 = (tile / a.tiles_x) * OW, ox0 = (tile % a.tiles_x) * OW;
    // 20x20 window of the patch: local (r, c) = image (oy0 + r - 2, ox0 + c - 2)
    for (int e = tid; e < cin * MXW * MXW; e += NT) {
      const int ci = e / (MXW * MXW), r = (e / MXW) % MXW, c = e % MXW;
      const int iy = oy0 + r - 2, ix = ox0 + c - 2;
      L.xs[e] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? a.f.x[(((long)pt * cin + ci) * H + iy) * W + ix] : 0.f;
    }
    lds_barrier();
    if (tid < MC1W * MC1W * 2) {  // conv1 + bias + ReLU over the 16x16 window: thread = (position, 4 of 8 channels)
      const int pix = tid >> 1, c0 = 4 * (tid & 1);
      const int y = pix / MC1W, xx = pix % MC1W;
      float4 acc = *reinterpret_cast<const float4 *>(L.w1 + 8 * cin * 25 + c0);
      for (int ci = 0; ci < cin; ++ci) {
        const float *xs = L.xs + (ci * MXW + y) * MXW + xx;
        const float *w = L.w1 + ci * 25 * 8 + c0;
#pragma unroll
        for (int t = 0; t < 25; ++t) {
          const float v = xs[(t / 5) * MXW + t % 5];
          const float4 wv = *reinterpret_cast<const float4 *>(w + t * 8);
          acc.x = fmaf(v, wv.x, acc.x);
          acc.y = fmaf(v, wv.y, acc.y);
          acc.z = fmaf(v, wv.z, acc.z);
          acc.w = fmaf(v, wv.w, acc.w);
        }
      }
      *reinterpret_cast<float4 *>(L.c1r + pix * 8 + c0) = float4{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
    }
    lds_barrier();
    // pool1 over the whole 15x15 a1 window: local (r, c) = a1 map (oy0 + r - 1, ox0 + c - 1); outside the map = conv2's zero padding
    for (int e = tid; e < A1PW * A1PW * 8; e += NT) {
      const int c = e & 7, q = e >> 3, r = q / A1PW, cc = q % A1PW;
      const int ay = oy0 + r - 1, ax = ox0 + cc - 1;
      float v = 0.f;
      if (ay >= 0 && ay < H - 3 && ax >= 0 && ax < W - 3) {
        const float *s1 = L.c1r + (r * MC1W + cc) * 8 + c;
        v = fmaxf(fmaxf(s1[0], s1[8]), fmaxf(s1[MC1W * 8], s1[MC1W * 8 + 8]));
      }
      const uint16_t h = f2bf(v);
      *reinterpret_cast<uint16_t *>(L.a1h + q * 16 + 2 * c) = h;
      if (SPLIT == 3) *reinterpret_cast<uint16_t *>(L.a1l + q * 16 + 2 * c) = f2bf(v - bf2f(h));
    }
    lds_barrier();
    conv2_relu<SPLIT, false, NT>(L, b2r, tid);
    lds_barrier();
    for (int e = tid; e < ON * 32; e += NT) {  // maxpool 2x2/1 -> the in-map part of the output tile
      const int c = e & 31, q = e >> 5, y = q / OW, x = q % OW;
      if (oy0 + y < Ho && ox0 + x < Wo) {
        const float *s2 = L.c2r + (y * C2W + x) * 32 + c;
        const float v = fmaxf(fmaxf(s2[0], s2[32]), fmaxf(s2[C2W * 32], s2[C2W * 32 + 32]));
        const uint16_t h = f2bf(v);
        const long o = (((long)pt * Ho + oy0 + y) * Wo + ox0 + x) * 32 + c;
        a.f.yh[o] = h;
        if (SPLIT == 3) a.f.yl[o] = f2bf(v - bf2f(h));
      }
    }
    // the next iteration's first barrier orders these c2r reads before conv2_relu overwrites it
  }
}

// fp32 conv2 weight [32][8][5][5] -> forward planes [7][32 co][32 k] (k = 8*(tap-4s) + ci, zero beyond tap 24)
//                                 -> backward planes [25 tap][8 ci][32 co]
__global__ __launch_bounds__(256) void pack_w2_kernel(const float *__restrict__ w, uint16_t *fh, uint16_t *fl,
                                                      uint16_t *bh, uint16_t *bl) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < KS2 * 32 * 32; e += gridDim.x * 256) {
    const int k = e & 31, co = (e >> 5) & 31, s = e >> 10;
    const int tap = 4 * s + (k >> 3), ci = k & 7;
    const float v = tap < 25 ? w[(co * 8 + ci) * 25 + tap] : 0.f;
    const uint16_t h = f2bf(v);
    fh[e] = h;
    if (fl) fl[e] = f2bf(v - bf2f(h));
  }
  for (int e = blockIdx.x * 256 + threadIdx.x; e < 25 * 8 * 32; e += gridDim.x * 256) {
    const int co = e & 31, ci = (e >> 5) & 7, tap = e >> 8;
    const float v = w[(co * 8 + ci) * 25 + tap];
    const uint16_t h = f2bf(v);
    bh[e] = h;
    if (bl) bl[e] = f2bf(v - bf2f(h));
  }
}

// ---- backward -------------------------------------------------------------------------------------
struct FrontBwdArgs {
  FrontArgs f;               // forward inputs (yh/yl unused)
  const uint16_t *w2bh, *w2bl;  // conv2 backward planes [25][8][32]
  const float *dy;           // [P][100][32] fp32 gradient of the pool2 output
  float *part;               // [nslice][PART] partial sums: dW2 [32][8][25], db2 [32], dW1 [8][cin][25], db1 [8]
  int patches_per_block;
  long long *stamps;         // `make STAMPS=1` builds: [grid][NPHASE] cycles per phase summed over the slice (else null)
  // tiled variant only (front_bwd_tile_kernel): patches are [P][cin][H][W], dy is the map [P][H-6][W-6][32], the unit of
  // work is (patch, 10x10 tile of that map) and `patches_per_block` counts units
  int H, W, tiles_x, tiles_y;
};
[[maybe_unused]] constexpr int NPHASE = 10;
#ifdef CRW_CONV_STAMPS
#define FRONT_STAMP(k)                                                       \
  if (a.stamps && tid == 0) {                                                \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();          \
    phase_[k] += now_ - prev_;                                               \
    prev_ = now_;                                                            \
  }
#else
#define FRONT_STAMP(k)
#endif

__device__ inline s4v tr_read0(uint32_t lds_addr) {
  s4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
__device__ inline bf8 tr_pair(uint32_t a_lo, uint32_t a_hi) {
  const s4v lo = tr_read0(a_lo), hi = tr_read0(a_hi);
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8, v);
}

// index (0..3) of the first maximum of a 2x2 window in row-major order (torch's max_pool2d tie rule)
__device__ inline int argmax4(float a, float b, float c, float d) {
  int k = 0;
  float m = a;
  if (b > m) { m = b; k = 1; }
  if (c > m) { m = c; k = 2; }
  if (d > m) { m = d; k = 3; }
  return k;
}

// Gradient reaching pixel (y, x) of a W x W map through a 2x2/stride-1 max-pool whose (W-1) x (W-1) output has
// gradient dout: the up to four windows containing the pixel route their gradient to it iff it is the FIRST
// maximum of the window in row-major order (torch's tie rule, = argmax4 above).  The four windows only involve
// the pixel's 3x3 neighbourhood, read once (8 LDS reads instead of 16).  in: bordered map (c1_idx / c2_idx with
// PAD), dout: [W-1][W-1], both channel-interleaved with CS floats per pixel; the ReLU gate (in[pixel] > 0) is applied here.
template <int W, int CS>
__device__ inline float pool_bwd_pixel(const float *__restrict__ in, const float *__restrict__ dout, int y, int x, int c) {
  // `in` is the bordered map [(W+2)][(W+2)][CS]: the 8 neighbours sit at constant offsets from the pixel.  All 13
  // LDS reads (pixel, neighbours, the 4 window gradients at clamped addresses) are unconditional and independent,
  // so they are in flight together; branches around single loads made each of them a separate round trip.
  const float *ctr = in + ((y + 1) * (W + 2) + x + 1) * CS + c;
  float nb[3][3], dv[2][2];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) nb[dy][dx] = ctr[((dy - 1) * (W + 2) + (dx - 1)) * CS];
#pragma unroll
  for (int dyw = 0; dyw < 2; ++dyw)
#pragma unroll
    for (int dxw = 0; dxw < 2; ++dxw) {
      const int wy = min(max(y - dyw, 0), W - 2), wx = min(max(x - dxw, 0), W - 2);
      dv[dyw][dxw] = dout[(wy * (W - 1) + wx) * CS + c];
    }
  const float mine = nb[1][1];
  float gsum = 0.f;
#pragma unroll
  for (int dyw = 0; dyw < 2; ++dyw)
#pragma unroll
    for (int dxw = 0; dxw < 2; ++dxw) {
      const int wy = y - dyw, wx = x - dxw;
      bool win = mine > 0.f && wy >= 0 && wy < W - 1 && wx >= 0 && wx < W - 1;  // ReLU; border values only meet invalid windows
      const int k = dyw * 2 + dxw;  // this pixel's position in the window
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j != k) {
          const float v = nb[1 - dyw + (j >> 1)][1 - dxw + (j & 1)];
          win = win && (j < k ? mine > v : mine >= v);
        }
      gsum += win ? dv[dyw][dxw] : 0.f;
    }
  return gsum;
}

template <int SPLIT>
__global__ __launch_bounds__(NTH) void front_bwd_kernel(FrontBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *p = lds;
  const int cin = a.f.cin;
  const FwdLds L = carve_fwd(p, cin, true);
  auto take = [&](size_t bytes) { char *r = p; p += (bytes + 15) & ~(size_t)15; return r; };
  char *wbh = take(25 * 8 * 32 * 2), *wbl = take(25 * 8 * 32 * 2);
  char *d2h = take(2 * D2HALF), *d2l = take(2 * D2HALF);  // dC2 (masked), padded by 4, [co half][pix][16] bf16
  float *dyb = (float *)take(sizeof(float) * ON * 32);                // dy of this patch; later dA1 [169][8] + dC1 [196][8]
  float *dA1 = dyb, *dC1 = dyb + A1W * A1W * 8;
  static_assert(A1W * A1W * 8 + C1N * 8 <= ON * 32, "alias");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  stage_constants<SPLIT>(L, a.f, tid);
  for (int e = tid; e < 25 * 8 * 32 / 8; e += NTH) {
    // global [tap][8 ci][4 chunks of 8 co] -> LDS [tap][co half][8 ci][2 chunks]
    const int ch = e & 3, ci = (e >> 2) & 7, tap = e >> 5;
    const int d = ((tap * 2 + (ch >> 1)) * 8 + ci) * 2 + (ch & 1);
    reinterpret_cast<uint4 *>(wbh)[d] = reinterpret_cast<const uint4 *>(a.w2bh)[e];
    if (SPLIT == 3) reinterpret_cast<uint4 *>(wbl)[d] = reinterpret_cast<const uint4 *>(a.w2bl)[e];
  }
  for (int e = tid; e < 2 * D2HALF / 4; e += NTH) {  // zero the padded gradient planes once (halo stays zero)
    reinterpret_cast<uint32_t *>(d2h)[e] = 0;
    reinterpret_cast<uint32_t *>(d2l)[e] = 0;
  }
  __syncthreads();

  // conv2 weight-gradient tiles: M = 32 co (2 tiles) x N = 13 tiles of (2 taps x 8 ci) = 26 tiles;
  // wave w owns tile w (and tile w + 16 for w < 10): tile t -> (co tile t & 1, N tile t >> 1)
  f32x4 wacc[2];
  wacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  wacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db2 = 0.f, db1 = 0.f, dw1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  // conv1 weight gradient: outputs; threads (co, ci, ty, row part), each part = cin pixel rows -> 560 threads either way
  const int nout = 8 * cin * 25, nrp = C1W / cin, nrowthr = 8 * cin * 5 * nrp;

  const uint32_t d2h_a = (uint32_t)(uintptr_t)(lds_cp)d2h, d2l_a = (uint32_t)(uintptr_t)(lds_cp)d2l;
  const uint32_t a1h_a = (uint32_t)(uintptr_t)(lds_cp)L.a1h, a1l_a = (uint32_t)(uintptr_t)(lds_cp)L.a1l;

  const int p_begin = blockIdx.x * a.patches_per_block;
  const int p_end = min(a.f.P, p_begin + a.patches_per_block);
#ifdef CRW_CONV_STAMPS
  long long phase_[NPHASE] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prev_ = (long long)__builtin_amdgcn_s_memtime();
#endif
  const float b2r = a.f.b2[16 * (wave & 1) + r16];  // read once: the barriers in the loop are memory clobbers
  // x (cin*256 floats) and dy (3200 floats) of the NEXT patch are fetched into registers while the current
  // one is processed, so no phase waits on HBM
  constexpr int DYIT = (ON * 32 / 4 + NTH - 1) / NTH;  // float4 chunks of dy per thread (800 chunks)
  float4 dy_r[DYIT];
  float x_r = 0.f;                                      // cin*256 <= 512 floats: threads < cin*256 hold one each
  // buffer loads: base in a scalar resource descriptor, a 32-bit per-lane offset and a scalar per-patch offset --
  // no 64-bit per-lane pointers live across the patch loop (they were being spilled and reloaded every
  // iteration with a full vmcnt(0) wait); reads past the end of the tensors return 0
  const __amdgpu_buffer_rsrc_t dy_rs =
      __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, (int)((long)a.f.P * ON * 32 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc((void *)a.f.x, 0, (int)((long)a.f.P * cin * 256 * 4), 0x00020000);
  auto fetch = [&](int pt) {
#pragma unroll
    for (int i = 0; i < DYIT; ++i)
      dy_r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(dy_rs, (tid + i * NTH) * 16, pt * (ON * 32 * 4), 0));
    x_r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, tid * 4, pt * (cin * 256 * 4), 0));
  };
  if (p_begin < p_end) fetch(p_begin);
  for (int pt = p_begin; pt < p_end; ++pt) {
    // ---- recompute the forward of this patch ---------------------------------------------------
    if (tid < cin * 256) L.xs[((tid >> 8) * XPW + ((tid >> 4) & 15) + 1) * XPW + (tid & 15) + 1] = x_r;
#pragma unroll
    for (int i = 0; i < DYIT; ++i)
      if (tid + i * NTH < ON * 32 / 4) reinterpret_cast<float4 *>(dyb)[tid + i * NTH] = dy_r[i];
    if (pt + 1 < p_end) fetch(pt + 1);
    lds_barrier();
    FRONT_STAMP(0)
    conv1_relu<true>(L, cin, tid);
    lds_barrier();
    FRONT_STAMP(1)
    pool1<SPLIT, true>(L, tid);
    lds_barrier();
    FRONT_STAMP(2)
    conv2_relu<SPLIT, true>(L, b2r, tid);
    lds_barrier();
    FRONT_STAMP(3)

    // ---- pool2 + ReLU2 backward: dC2[pix][co] (masked) -> padded bf16 planes, bias gradient ------
    for (int e = tid; e < C2N * 32; e += NTH) {
      const int co = e & 31, pix = e >> 5, y = pix / C2W, x = pix % C2W;
      const float gsum = pool_bwd_pixel<C2W, 32>(L.c2r, dyb, y, x, co);
      db2 += gsum;  // thread t always meets channel t & 31
      const uint16_t h = f2bf(gsum);
      const int o = (co >> 4) * D2HALF + ((y + 4) * D2PW + x + 4) * 32 + 2 * (co & 15);
      *reinterpret_cast<uint16_t *>(d2h + o) = h;
      if (SPLIT == 3) *reinterpret_cast<uint16_t *>(d2l + o) = f2bf(gsum - bf2f(h));
    }
    lds_barrier();
    FRONT_STAMP(4)

    // ---- conv2 weight gradient: dW2[co][ci][tap] += sum_pix dC2[pix][co] * a1pad[pix + tap][ci] -----
    {
      const int t16 = lane & 15, q = t16 >> 2, pq = t16 & 3;
#pragma unroll 1
      for (int ks = 0; ks < 4; ++ks) {  // 121 pixels in 4 k-steps of 32 (rows >= 121 hit a zero halo pixel of dC2)
        const int i_lo = 32 * ks + 8 * g + q, i_hi = i_lo + 4;
        const bool v_lo = i_lo < C2N, v_hi = i_hi < C2N;
        const int y_lo = v_lo ? i_lo / C2W : 0, x_lo = v_lo ? i_lo % C2W : 0;
        const int y_hi = v_hi ? i_hi / C2W : 0, x_hi = v_hi ? i_hi % C2W : 0;
        // A: dC2^T, rows = pixels (padded plane index, or halo pixel 0 for dummy rows), 16 co per tile
        const uint32_t ya_lo = (v_lo ? ((y_lo + 4) * D2PW + x_lo + 4) : 0) * 32 + 8 * (pq & 1) + 16 * (pq >> 1);
        const uint32_t ya_hi = (v_hi ? ((y_hi + 4) * D2PW + x_hi + 4) : 0) * 32 + 8 * (pq & 1) + 16 * (pq >> 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int tile = wave + NWV * u;
          if (tile < 26) {  // wave-uniform
            const int i = tile & 1, nt = tile >> 1;
            int tap = 2 * nt + (pq >> 1);
            if (tap > 24) tap = 24;  // the 26th tap does not exist: recompute tap 24, dropped at the end
            const int toff = (tap / 5) * A1PW + (tap % 5);
            // B: a1 padded plane [pix][8 ci] (16 B rows): fragment columns 0-7 = tap 2nt, 8-15 = tap 2nt+1
            const uint32_t xa_lo = ((y_lo * A1PW + x_lo) + toff) * 16 + 8 * (pq & 1);
            const uint32_t xa_hi = ((y_hi * A1PW + x_hi) + toff) * 16 + 8 * (pq & 1);
            const bf8 ah = tr_pair(d2h_a + ya_lo + D2HALF * i, d2h_a + ya_hi + D2HALF * i);
            const bf8 bh = tr_pair(a1h_a + xa_lo, a1h_a + xa_hi);
            bf8 al, bl;
            if (SPLIT == 3) {
              al = tr_pair(d2l_a + ya_lo + D2HALF * i, d2l_a + ya_hi + D2HALF * i);
              bl = tr_pair(a1l_a + xa_lo, a1l_a + xa_hi);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (SPLIT == 3) {
              wacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, wacc[u], 0, 0, 0);
              wacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, wacc[u], 0, 0, 0);
            }
            wacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, wacc[u], 0, 0, 0);
          }
        }
      }
    }

    FRONT_STAMP(5)
    // ---- conv2 backward-data: dA1[ya][xa][ci] = sum_{tap,co} dC2[ya - ty + 1][xa - tx + 1][co] W2[co][ci][tap] ----
    {
      // Only 8 of the 16 MFMA output columns are real (8 input channels), so for bf16x3 the idle columns carry
      // a correction term: B columns 0-7 = Wh, 8-15 = Wl (lanes r16 >= 8 simply read the lo plane), hence
      //   ah x [Wh | Wl] -> columns 0-7 = ah*Wh, 8-15 = ah*Wl        al x [Wh | ..] -> columns 0-7 = al*Wh
      // two MFMAs and three LDS reads per tap instead of three and four; the halves meet in a lane shift.
      f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f}, dacc1 = dacc;
      // One image row of dA1 (13 pixels) per wave, dealt to the MFMA rows by LDS bank residue: a ds_read_b128 is served in
      // 16-lane groups made of tile rows {0-3, 12-15} and {4-11}, conflict-free on these 32-byte pixel rows iff the pixel
      // positions inside each set are distinct mod 8 -- rows 4-11 take x = 0..7, rows 0-3 and 12 take x = 8..12, rows 13-15
      // repeat x = 12 (same address: broadcast).  Sixteen consecutive pixels of the 13-wide map (11 tiles) broke that at
      // every row end; 13 tiles cost nothing, 5 of the 16 waves were idle in this phase.
      if (wave < A1W) {  // wave-uniform
        const int xl = (r16 >= 4 && r16 < 12) ? r16 - 4 : (r16 < 4 ? 8 + r16 : 12);
        const int base = (wave + 5) * D2PW + xl + 5;  // padded dC2 pixel of tap (0,0); tap shifts by -(ty*19 + tx)
        // B: backward weights, LDS layout [tap][co half][8 ci][16 co]: lane (ci = r16 & 7, k chunk g = co 8g..8g+7)
        const char *wb = ((SPLIT == 3 && r16 >= 8) ? wbl : wbh) + (g >> 1) * 256 + (r16 & 7) * 32 + (g & 1) * 16;
        // taps shift the A pixel by -(ty*19 + tx); rebased so that every tap is a non-negative immediate
        const char *ab = d2h + (g >> 1) * D2HALF + (base - (4 * D2PW + 4)) * 32 + (g & 1) * 16;
        const long lo_a = d2l - d2h;
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
          const int arel = ((4 - tap / 5) * D2PW + (4 - tap % 5)) * 32;
          const bf8 b = *reinterpret_cast<const bf8 *>(wb + tap * 512);
          const bf8 ah = *reinterpret_cast<const bf8 *>(ab + arel);
          if (SPLIT == 3) {
            const bf8 al = *reinterpret_cast<const bf8 *>(ab + lo_a + arel);
            dacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b, dacc1, 0, 0, 0);
          }
          dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b, dacc, 0, 0, 0);
        }
        if (SPLIT == 3) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dacc[r] += __shfl_down(dacc[r], 8, 64) + dacc1[r];  // lanes r16 < 8 are the ones stored
        }
      }
      lds_barrier();  // dyb (aliased by dA1) is no longer read
      if (wave < A1W && r16 < 8)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * g + r;  // tile row -> x as above; rows 13-15 only repeated pixel 12
          const int x = (row >= 4 && row < 12) ? row - 4 : (row < 4 ? 8 + row : 12);
          if (row < 13) dA1[(wave * A1W + x) * 8 + r16] = dacc[r];
        }
    }
    lds_barrier();
    FRONT_STAMP(6)

    // ---- pool1 + ReLU1 backward -> dC1 [196][8]; then the d2 planes' interior is cleared for the next patch ----
    for (int e = tid; e < C1N * 8; e += NTH) {
      const int co = e & 7, pix = e >> 3, y = pix / C1W, x = pix % C1W;
      const float gsum = pool_bwd_pixel<C1W, 8>(L.c1r, dA1, y, x, co);
      dC1[co * C1N + pix] = gsum;  // channel-major: the weight-gradient threads read whole pixel rows
      db1 += gsum;                 // thread t always meets channel t & 7
    }
    lds_barrier();
    FRONT_STAMP(7)

    // ---- conv1 weight gradient (VALU): thread = (co, ci, tap row ty, pixel row y).  It reads its 14 dC1 values and the
    // 18 input values of the shifted row once and forms the 5 taps of that row from registers (32 LDS reads per 70
    // FMAs; one output per thread needed 2 reads per FMA and was LDS-bound).  Lanes differ in y: strides 14 and 18
    // floats are conflict-free.  The 14 row sums of an output meet at the very end of the kernel ----
    if (tid < nrowthr) {
      const int rp = tid % nrp, q = tid / nrp, ty = q % 5, ci = (q / 5) % cin, co = q / (5 * cin);
      for (int y = rp; y < C1W; y += nrp) {
        const float *dr = dC1 + co * C1N + y * C1W, *xr = L.xs + (ci * XPW + y + ty) * XPW;
        float dv[C1W], xv[XPW];
#pragma unroll
        for (int x = 0; x < C1W; ++x) dv[x] = dr[x];
#pragma unroll
        for (int x = 0; x < XPW; ++x) xv[x] = xr[x];
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
          float s1 = 0.f;
#pragma unroll
          for (int x = 0; x < C1W; ++x) s1 = fmaf(dv[x], xv[x + tx], s1);
          dw1[tx] += s1;
        }
      }
    }
    lds_barrier();  // next patch may overwrite xs / dyb
    FRONT_STAMP(8)
  }
#ifdef CRW_CONV_STAMPS
  if (a.stamps && tid == 0)
    for (int k = 0; k < NPHASE; ++k) a.stamps[(long)blockIdx.x * NPHASE + k] = phase_[k];
#endif

  // ---- partial sums of this slice -> workspace ------------------------------------------------------
  const int PART = 32 * 8 * 25 + 32 + 8 * cin * 25 + 8;
  float *out = a.part + (long)blockIdx.x * PART;
  // wacc[u][r] of tile t = wave + 16 u: dW2[co = 16 (t & 1) + 4 g + r][ci = r16 & 7][tap = 2 (t >> 1) + (r16 >> 3)]
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int tile = wave + NWV * u;
    const int tap = 2 * (tile >> 1) + (r16 >> 3);
    if (tile < 26 && tap < 25)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((16 * (tile & 1) + 4 * g + r) * 8 + (r16 & 7)) * 25 + tap] = wacc[u][r];
  }
  // db2: thread t summed channel t & 31 over its pixel subset -> reduce the 32 threads per channel through LDS
  __syncthreads();
  float *red = reinterpret_cast<float *>(lds);
  red[tid] = db2;
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
    for (int k = 0; k < NTH / 32; ++k) s += red[tid + 32 * k];
    out[32 * 8 * 25 + tid] = s;
  }
  __syncthreads();
  if (tid < nrowthr)
#pragma unroll
    for (int tx = 0; tx < 5; ++tx) red[tid * 5 + tx] = dw1[tx];  // [(co, ci, ty)][row part][tx]
  __syncthreads();
  if (tid < nout) {  // output o = ((co * cin + ci) * 5 + ty) * 5 + tx: add its row parts in a fixed order
    const int q = tid / 5, tx = tid % 5;
    float s = 0.f;
    for (int y = 0; y < nrp; ++y) s += red[(q * nrp + y) * 5 + tx];
    out[32 * 8 * 25 + 32 + tid] = s;
  }
  __syncthreads();
  red[tid] = db1;
  __syncthreads();
  if (tid < 8) {
    float s = 0.f;
    for (int k = 0; k < NTH / 8; ++k) s += red[tid + 8 * k];
    out[32 * 8 * 25 + 32 + nout + tid] = s;
  }
}

// ---- backward on patches of any size (training at patch sizes other than 16x16) -----------------------------------------
// The unit of work is (patch, 10x10 tile of the pool2 output map), with the geometry of front_fwd_map_kernel: a 20x20 window
// of the patch (zeros outside it), conv1 over 16x16 positions, the 15x15 a1 window conv2 needs (a position outside the a1 map
// holds conv2's zero padding and passes no gradient), conv2 11x11, pool2 10x10.  Every gradient is linear in dy and every
// pooled output belongs to exactly one tile, so the per-tile contributions to dW1 / dW2 / db simply add up (overlapping
// windows recompute the same forward values, hence the same ReLU gates and arg-max choices).  The phases are those of
// front_bwd_kernel -- whose 16x16 patch is the one-tile case with the window's border ring outside the maps -- with the
// extents 16 / 15 instead of 14 / 13; conv2, pool2 backward and the conv2 weight gradient are shared code.
constexpr int TXW = MXW, TCW = MC1W, TCN = TCW * TCW, TAW = A1PW;  // 20, 16, 256, 15

template <int SPLIT>
__global__ __launch_bounds__(NTH) void front_bwd_tile_kernel(FrontBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *p = lds;
  const int cin = a.f.cin;
  auto take = [&](size_t bytes) { char *r = p; p += (bytes + 15) & ~(size_t)15; return r; };
  FwdLds L;
  L.xs = (float *)take(sizeof(float) * cin * TXW * TXW);
  L.w1 = (float *)take(sizeof(float) * (8 * cin * 25 + 8));
  L.c1r = (float *)take(sizeof(float) * (TCW + 2) * (TCW + 2) * 8);  // bordered: pool_bwd_pixel reads 3x3 neighbourhoods
  L.a1h = take(TAW * TAW * 16);
  L.a1l = take(TAW * TAW * 16);
  L.w2h = take(KS2 * 32 * 32 * 2);
  L.w2l = take(KS2 * 32 * 32 * 2);
  L.c2r = (float *)take(sizeof(float) * (C2W + 2) * (C2W + 2) * 32);
  char *wbh = take(25 * 8 * 32 * 2), *wbl = take(25 * 8 * 32 * 2);
  char *d2h = take(2 * D2HALF), *d2l = take(2 * D2HALF);
  float *dyb = (float *)take(sizeof(float) * ON * 32);
  // dA1 [225][8] aliases dyb and dC1 [8][256] aliases c2r: both are dead after the pool2 backward phase (157 KB of LDS in all)
  float *dA1 = dyb, *dC1 = L.c2r;
  static_assert(TAW * TAW * 8 <= ON * 32 && TCN * 8 <= (C2W + 2) * (C2W + 2) * 32, "aliases");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  for (int e = tid; e < 8 * cin * 25; e += NTH) {  // [co][ci][tap] -> [ci][tap][co]
    const int t = e % 25, ci = (e / 25) % cin, co = e / (25 * cin);
    L.w1[(ci * 25 + t) * 8 + co] = a.f.w1[e];
  }
  if (tid < 8) L.w1[8 * cin * 25 + tid] = a.f.b1[tid];
  for (int e = tid; e < KS2 * 32 * 32 / 8; e += NTH) {
    const int d = w2_lds_chunk(e);
    reinterpret_cast<uint4 *>(L.w2h)[d] = reinterpret_cast<const uint4 *>(a.f.w2h)[e];
    if (SPLIT == 3) reinterpret_cast<uint4 *>(L.w2l)[d] = reinterpret_cast<const uint4 *>(a.f.w2l)[e];
  }
  for (int e = tid; e < 25 * 8 * 32 / 8; e += NTH) {
    const int ch = e & 3, ci = (e >> 2) & 7, tap = e >> 5;
    const int d = ((tap * 2 + (ch >> 1)) * 8 + ci) * 2 + (ch & 1);
    reinterpret_cast<uint4 *>(wbh)[d] = reinterpret_cast<const uint4 *>(a.w2bh)[e];
    if (SPLIT == 3) reinterpret_cast<uint4 *>(wbl)[d] = reinterpret_cast<const uint4 *>(a.w2bl)[e];
  }
  for (int e = tid; e < 2 * D2HALF / 4; e += NTH) {
    reinterpret_cast<uint32_t *>(d2h)[e] = 0;
    reinterpret_cast<uint32_t *>(d2l)[e] = 0;
  }
  __syncthreads();

  f32x4 wacc[2];
  wacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  wacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db2 = 0.f, db1 = 0.f, dw1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int nout = 8 * cin * 25, nrp = TCW / cin, nrowthr = 8 * cin * 5 * nrp;  // 640 threads either way
  const uint32_t d2h_a = (uint32_t)(uintptr_t)(lds_cp)d2h, d2l_a = (uint32_t)(uintptr_t)(lds_cp)d2l;
  const uint32_t a1h_a = (uint32_t)(uintptr_t)(lds_cp)L.a1h, a1l_a = (uint32_t)(uintptr_t)(lds_cp)L.a1l;
  const int H = a.H, W = a.W, Ho = H - 6, Wo = W - 6, ntile = a.tiles_x * a.tiles_y;
  const int u_begin = blockIdx.x * a.patches_per_block;
  const int u_end = min(a.f.P * ntile, u_begin + a.patches_per_block);
  const float b2r = a.f.b2[16 * (wave & 1) + r16];

  for (int u = u_begin; u < u_end; ++u) {
    const int pt = u / ntile, tile = u % ntile;
    const int oy0 = (tile / a.tiles_x) * OW, ox0 = (tile % a.tiles_x) * OW;
    // ---- window of the patch and the tile of dy (zeros outside the patch / the map) ---------------------------
    for (int e = tid; e < cin * TXW * TXW; e += NTH) {
      const int ci = e / (TXW * TXW), r = (e / TXW) % TXW, c = e % TXW;
      const int iy = oy0 + r - 2, ix = ox0 + c - 2;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float v = a.f.x[(((long)pt * cin + ci) * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
      L.xs[e] = ok ? v : 0.f;
    }
    for (int e = tid; e < ON * 32 / 4; e += NTH) {
      const int q = e >> 3, c4 = e & 7, y = q / OW, x = q % OW;
      const bool ok = oy0 + y < Ho && ox0 + x < Wo;
      const float4 v = *reinterpret_cast<const float4 *>(a.dy + (((long)pt * Ho + min(oy0 + y, Ho - 1)) * Wo + min(ox0 + x, Wo - 1)) * 32 + 4 * c4);
      reinterpret_cast<float4 *>(dyb)[e] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
    lds_barrier();
    // ---- recompute the forward of this tile -------------------------------------------------------------------
    if (tid < TCN * 2) {  // conv1 + bias + ReLU over the 16x16 positions: thread = (position, 4 of 8 channels)
      const int pix = tid >> 1, c0 = 4 * (tid & 1);
      const int y = pix / TCW, xx = pix % TCW;
      float4 acc = *reinterpret_cast<const float4 *>(L.w1 + 8 * cin * 25 + c0);
      for (int ci = 0; ci < cin; ++ci) {
        const float *xs = L.xs + (ci * TXW + y) * TXW + xx;
        const float *w = L.w1 + ci * 25 * 8 + c0;
#pragma unroll
        for (int t = 0; t < 25; ++t) {
          const float v = xs[(t / 5) * TXW + t % 5];
          const float4 wv = *reinterpret_cast<const float4 *>(w + t * 8);
          acc.x = fmaf(v, wv.x, acc.x);
          acc.y = fmaf(v, wv.y, acc.y);
          acc.z = fmaf(v, wv.z, acc.z);
          acc.w = fmaf(v, wv.w, acc.w);
        }
      }
      *reinterpret_cast<float4 *>(L.c1r + ((y + 1) * (TCW + 2) + xx + 1) * 8 + c0) =
          float4{fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)};
    }
    lds_barrier();
    // pool1 over the whole 15x15 a1 window: local (r, c) = a1 map (oy0 + r - 1, ox0 + c - 1); outside the map = conv2's zero padding
    for (int e = tid; e < TAW * TAW * 8; e += NTH) {
      const int c = e & 7, q = e >> 3, r = q / TAW, cc = q % TAW;
      const int ay = oy0 + r - 1, ax = ox0 + cc - 1;
      float v = 0.f;
      if (ay >= 0 && ay < H - 3 && ax >= 0 && ax < W - 3) {
        const float *s1 = L.c1r + ((r + 1) * (TCW + 2) + cc + 1) * 8 + c;
        v = fmaxf(fmaxf(s1[0], s1[8]), fmaxf(s1[(TCW + 2) * 8], s1[(TCW + 2) * 8 + 8]));
      }
      const uint16_t h = f2bf(v);
      *reinterpret_cast<uint16_t *>(L.a1h + q * 16 + 2 * c) = h;
      if (SPLIT == 3) *reinterpret_cast<uint16_t *>(L.a1l + q * 16 + 2 * c) = f2bf(v - bf2f(h));
    }
    lds_barrier();
    conv2_relu<SPLIT, true>(L, b2r, tid);
    lds_barrier();

    // ---- pool2 + ReLU2 backward (as in front_bwd_kernel) ------------------------------------------------------
    for (int e = tid; e < C2N * 32; e += NTH) {
      const int co = e & 31, pix = e >> 5, y = pix / C2W, x = pix % C2W;
      const float gsum = pool_bwd_pixel<C2W, 32>(L.c2r, dyb, y, x, co);
      db2 += gsum;
      const uint16_t h = f2bf(gsum);
      const int o = (co >> 4) * D2HALF + ((y + 4) * D2PW + x + 4) * 32 + 2 * (co & 15);
      *reinterpret_cast<uint16_t *>(d2h + o) = h;
      if (SPLIT == 3) *reinterpret_cast<uint16_t *>(d2l + o) = f2bf(gsum - bf2f(h));
    }
    lds_barrier();

    // ---- conv2 weight gradient (as in front_bwd_kernel) -------------------------------------------------------
    {
      const int t16 = lane & 15, q = t16 >> 2, pq = t16 & 3;
#pragma unroll 1
      for (int ks = 0; ks < 4; ++ks) {
        const int i_lo = 32 * ks + 8 * g + q, i_hi = i_lo + 4;
        const bool v_lo = i_lo < C2N, v_hi = i_hi < C2N;
        const int y_lo = v_lo ? i_lo / C2W : 0, x_lo = v_lo ? i_lo % C2W : 0;
        const int y_hi = v_hi ? i_hi / C2W : 0, x_hi = v_hi ? i_hi % C2W : 0;
        const uint32_t ya_lo = (v_lo ? ((y_lo + 4) * D2PW + x_lo + 4) : 0) * 32 + 8 * (pq & 1) + 16 * (pq >> 1);
        const uint32_t ya_hi = (v_hi ? ((y_hi + 4) * D2PW + x_hi + 4) : 0) * 32 + 8 * (pq & 1) + 16 * (pq >> 1);
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          const int tl = wave + NWV * uu;
          if (tl < 26) {
            const int i = tl & 1, nt = tl >> 1;
            int tap = 2 * nt + (pq >> 1);
            if (tap > 24) tap = 24;
            const int toff = (tap / 5) * A1PW + (tap % 5);
            const uint32_t xa_lo = ((y_lo * A1PW + x_lo) + toff) * 16 + 8 * (pq & 1);
            const uint32_t xa_hi = ((y_hi * A1PW + x_hi) + toff) * 16 + 8 * (pq & 1);
            const bf8 ah = tr_pair(d2h_a + ya_lo + D2HALF * i, d2h_a + ya_hi + D2HALF * i);
            const bf8 bh = tr_pair(a1h_a + xa_lo, a1h_a + xa_hi);
            bf8 al, bl;
            if (SPLIT == 3) {
              al = tr_pair(d2l_a + ya_lo + D2HALF * i, d2l_a + ya_hi + D2HALF * i);
              bl = tr_pair(a1l_a + xa_lo, a1l_a + xa_hi);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (SPLIT == 3) {
              wacc[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, wacc[uu], 0, 0, 0);
              wacc[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, wacc[uu], 0, 0, 0);
            }
            wacc[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, wacc[uu], 0, 0, 0);
          }
        }
      }
    }
    // ---- conv2 backward-data: one row of the 15x15 a1 window per wave (15 waves), masked where the window leaves the a1 map ----
    {
      f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f}, dacc1 = dacc;
      if (wave < TAW) {
        const int xl = (r16 >= 4 && r16 < 12) ? r16 - 4 : (r16 < 4 ? 8 + r16 : min(r16, TAW - 1));
        const int base = (wave + 4) * D2PW + xl + 4;  // a1 window position (wave, xl) = padded dC2 pixel of tap (0,0)
        const char *wb = ((SPLIT == 3 && r16 >= 8) ? wbl : wbh) + (g >> 1) * 256 + (r16 & 7) * 32 + (g & 1) * 16;
        const char *ab = d2h + (g >> 1) * D2HALF + (base - (4 * D2PW + 4)) * 32 + (g & 1) * 16;
        const long lo_a = d2l - d2h;
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
          const int arel = ((4 - tap / 5) * D2PW + (4 - tap % 5)) * 32;
          const bf8 b = *reinterpret_cast<const bf8 *>(wb + tap * 512);
          const bf8 ah = *reinterpret_cast<const bf8 *>(ab + arel);
          if (SPLIT == 3) {
            const bf8 al = *reinterpret_cast<const bf8 *>(ab + lo_a + arel);
            dacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b, dacc1, 0, 0, 0);
          }
          dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b, dacc, 0, 0, 0);
        }
        if (SPLIT == 3) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dacc[r] += __shfl_down(dacc[r], 8, 64) + dacc1[r];
        }
        if (r16 < 8) {
          // (opaque to the compiler: it would hoist the four rows' positions and LDS addresses out of the unit loop and, at the
          // 128-register cap of a 1024-thread workgroup, park them in scratch -- reloaded with an exposed latency each)
          int gq = g;
          asm volatile("" : "+v"(gq));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * gq + r;
            const int x = (row >= 4 && row < 12) ? row - 4 : (row < 4 ? 8 + row : min(row, TAW - 1));
            const int ay = oy0 + wave - 1, ax = ox0 + x - 1;
            const bool ok = ay >= 0 && ay < H - 3 && ax >= 0 && ax < W - 3;
            if (row < TAW) dA1[(wave * TAW + x) * 8 + r16] = ok ? dacc[r] : 0.f;
          }
        }
      }
    }
    lds_barrier();

    // ---- pool1 + ReLU1 backward -> dC1 [8][256] ---------------------------------------------------------------
    for (int e = tid; e < TCN * 8; e += NTH) {
      const int co = e & 7, pix = e >> 3, y = pix / TCW, x = pix % TCW;
      const float gsum = pool_bwd_pixel<TCW, 8>(L.c1r, dA1, y, x, co);
      dC1[co * TCN + pix] = gsum;
      db1 += gsum;
    }
    lds_barrier();

    // ---- conv1 weight gradient: thread = (co, ci, tap row ty, pixel-row part) ----------------------------------
    if (tid < nrowthr) {
      const int rp = tid % nrp, q = tid / nrp, ty = q % 5, ci = (q / 5) % cin, co = q / (5 * cin);
      for (int y = rp; y < TCW; y += nrp) {
        const float *dr = dC1 + co * TCN + y * TCW, *xr = L.xs + (ci * TXW + y + ty) * TXW;
        // two halves of the pixel row: 8 + 12 values live at a time instead of 16 + 20 (the whole row at once put this kernel
        // 84 bytes per lane into scratch at its 128-register cap)
#pragma unroll 1
        for (int h = 0; h < TCW; h += TCW / 2) {
          float dv[TCW / 2], xv[TCW / 2 + 4];
#pragma unroll
          for (int x = 0; x < TCW / 2; ++x) dv[x] = dr[h + x];
#pragma unroll
          for (int x = 0; x < TCW / 2 + 4; ++x) xv[x] = xr[h + x];
#pragma unroll
          for (int tx = 0; tx < 5; ++tx) {
            float s1 = 0.f;
#pragma unroll
            for (int x = 0; x < TCW / 2; ++x) s1 = fmaf(dv[x], xv[x + tx], s1);
            dw1[tx] += s1;
          }
        }
      }
    }
    lds_barrier();  // the next unit overwrites xs / dyb
  }

  // ---- partial sums of this slice -> workspace (layout of front_bwd_kernel) ---------------------------
  const int PART = 32 * 8 * 25 + 32 + 8 * cin * 25 + 8;
  float *out = a.part + (long)blockIdx.x * PART;
#pragma unroll
  for (int uu = 0; uu < 2; ++uu) {
    const int tl = wave + NWV * uu;
    const int tap = 2 * (tl >> 1) + (r16 >> 3);
    if (tl < 26 && tap < 25)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((16 * (tl & 1) + 4 * g + r) * 8 + (r16 & 7)) * 25 + tap] = wacc[uu][r];
  }
  __syncthreads();
  float *red = reinterpret_cast<float *>(lds);
  red[tid] = db2;
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
    for (int k = 0; k < NTH / 32; ++k) s += red[tid + 32 * k];
    out[32 * 8 * 25 + tid] = s;
  }
  __syncthreads();
  if (tid < nrowthr)
#pragma unroll
    for (int tx = 0; tx < 5; ++tx) red[tid * 5 + tx] = dw1[tx];
  __syncthreads();
  if (tid < nout) {
    const int q = tid / 5, tx = tid % 5;
    float s = 0.f;
    for (int y = 0; y < nrp; ++y) s += red[(q * nrp + y) * 5 + tx];
    out[32 * 8 * 25 + 32 + tid] = s;
  }
  __syncthreads();
  red[tid] = db1;
  __syncthreads();
  if (tid < 8) {
    float s = 0.f;
    for (int k = 0; k < NTH / 8; ++k) s += red[tid + 8 * k];
    out[32 * 8 * 25 + 32 + nout + tid] = s;
  }
}

// ---- backward without recomputation -------------------------------------------------------------------------------
// The same gradients from what the forward pass saved (pool1 planes + pooling codes, FrontArgs::saved): no conv1 / pool1 /
// conv2 recomputation (7.5 k of the 29 k cycles per patch of front_bwd_kernel) and the two pooling backward passes read
// one code byte per window instead of the window's values (8 LDS reads per pixel instead of 13, no comparisons).
// Phases per patch: registers -> LDS | pool2 backward | conv2 weight gradient + backward-data | pool1 backward | conv1
// weight gradient.  x, dy and the saved record of the NEXT patch are fetched into registers while this one is processed.
template <int SPLIT>
__global__ __launch_bounds__(NTH) void front_bwd_saved_kernel(FrontBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *p = lds;
  const int cin = a.f.cin;
  auto take = [&](size_t bytes) { char *r = p; p += (bytes + 15) & ~(size_t)15; return r; };
  float *xs = (float *)take(sizeof(float) * cin * XPW * XPW);
  char *a1h = take(A1PW * A1PW * 16), *a1l = take(A1PW * A1PW * 16);
  char *wbh = take(25 * 8 * 32 * 2), *wbl = take(25 * 8 * 32 * 2);
  char *d2h = take(2 * D2HALF), *d2l = take(2 * D2HALF);
  float *dyb = (float *)take(sizeof(float) * ON * 32);
  char *k2s = take(SV_K2), *k1s = take(SV_K1);
  float *dA1 = dyb, *dC1 = dyb + A1W * A1W * 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  for (int e = tid; e < cin * XPW * XPW; e += NTH) xs[e] = 0.f;
  for (int e = tid; e < A1PW * A1PW * 4; e += NTH) {
    reinterpret_cast<uint32_t *>(a1h)[e] = 0;
    reinterpret_cast<uint32_t *>(a1l)[e] = 0;
  }
  for (int e = tid; e < 25 * 8 * 32 / 8; e += NTH) {
    const int ch = e & 3, ci = (e >> 2) & 7, tap = e >> 5;
    const int d = ((tap * 2 + (ch >> 1)) * 8 + ci) * 2 + (ch & 1);
    reinterpret_cast<uint4 *>(wbh)[d] = reinterpret_cast<const uint4 *>(a.w2bh)[e];
    if (SPLIT == 3) reinterpret_cast<uint4 *>(wbl)[d] = reinterpret_cast<const uint4 *>(a.w2bl)[e];
  }
  for (int e = tid; e < 2 * D2HALF / 4; e += NTH) {
    reinterpret_cast<uint32_t *>(d2h)[e] = 0;
    reinterpret_cast<uint32_t *>(d2l)[e] = 0;
  }
  __syncthreads();

  f32x4 wacc[2];
  wacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  wacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db2 = 0.f, db1 = 0.f, dw1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int nout = 8 * cin * 25, nrp = C1W / cin, nrowthr = 8 * cin * 5 * nrp;
  const uint32_t d2h_a = (uint32_t)(uintptr_t)(lds_cp)d2h, d2l_a = (uint32_t)(uintptr_t)(lds_cp)d2l;
  const uint32_t a1h_a = (uint32_t)(uintptr_t)(lds_cp)a1h, a1l_a = (uint32_t)(uintptr_t)(lds_cp)a1l;
  const int p_begin = blockIdx.x * a.patches_per_block;
  const int p_end = min(a.f.P, p_begin + a.patches_per_block);

  // prefetch registers: dy (one float4), x (one float), and one 16-byte chunk of the saved record per thread:
  // chunks 0..337 = a1 hi / lo planes, 338..537 = k2 codes, 538..622 = k1 codes
  constexpr int NA1 = 2 * SV_A1 / 16, NK2 = SV_K2 / 16, NK1 = SV_K1 / 16, NSV = NA1 + NK2 + NK1;
  static_assert(NSV <= NTH && ON * 32 / 4 <= NTH, "one chunk per thread");
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  float4 dy_r;
  float x_r = 0.f;
  u4v sv_r;
  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, (int)((long)a.f.P * ON * 32 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.f.x, 0, (int)((long)a.f.P * cin * 256 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t sv_rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.f.saved, 0, (int)((long)a.f.P * SV_BYTES), 0x00020000);
  auto fetch = [&](int pt) {
    dy_r = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(dy_rs, tid * 16, pt * (ON * 32 * 4), 0));
    x_r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, tid * 4, pt * (cin * 256 * 4), 0));
    sv_r = __builtin_amdgcn_raw_buffer_load_b128(sv_rs, min(tid, NSV - 1) * 16, pt * SV_BYTES, 0);
  };
  if (p_begin < p_end) fetch(p_begin);
  for (int pt = p_begin; pt < p_end; ++pt) {
    // ---- registers -> LDS ----------------------------------------------------------------------
    if (tid < cin * 256) xs[((tid >> 8) * XPW + ((tid >> 4) & 15) + 1) * XPW + (tid & 15) + 1] = x_r;
    if (tid < ON * 32 / 4) reinterpret_cast<float4 *>(dyb)[tid] = dy_r;
    if (tid < NA1) {  // a1 plane chunk: pixel (tid % 169) of plane (tid / 169) -> interior of the padded 15x15 image
      const int pl = tid / (A1W * A1W), px = tid % (A1W * A1W);
      *reinterpret_cast<u4v *>((pl ? a1l : a1h) + ((px / A1W + 1) * A1PW + px % A1W + 1) * 16) = sv_r;
    } else if (tid < NA1 + NK2) {
      reinterpret_cast<u4v *>(k2s)[tid - NA1] = sv_r;
    } else if (tid < NSV) {
      reinterpret_cast<u4v *>(k1s)[tid - NA1 - NK2] = sv_r;
    }
    if (pt + 1 < p_end) fetch(pt + 1);
    lds_barrier();

    // ---- pool2 + ReLU2 backward from the codes: dC2[pix][co] -> padded bf16 planes, bias gradient ------
    for (int e = tid; e < C2N * 32; e += NTH) {
      const int co = e & 31, pix = e >> 5, y = pix / C2W, x = pix % C2W;
      float gsum = 0.f;
#pragma unroll
      for (int dyw = 0; dyw < 2; ++dyw)
#pragma unroll
        for (int dxw = 0; dxw < 2; ++dxw) {
          const int wy = y - dyw, wx = x - dxw;
          const bool ok = wy >= 0 && wy < OW && wx >= 0 && wx < OW;
          const int idx = (min(max(wy, 0), OW - 1) * OW + min(max(wx, 0), OW - 1)) * 32 + co;
          const int code = k2s[idx];
          const float d = dyb[idx];
          gsum += (ok && code == (4 | (dyw * 2 + dxw))) ? d : 0.f;  // this pixel is the window's first maximum and it is > 0
        }
      db2 += gsum;
      const uint16_t h = f2bf(gsum);
      const int o = (co >> 4) * D2HALF + ((y + 4) * D2PW + x + 4) * 32 + 2 * (co & 15);
      *reinterpret_cast<uint16_t *>(d2h + o) = h;
      if (SPLIT == 3) *reinterpret_cast<uint16_t *>(d2l + o) = f2bf(gsum - bf2f(h));
    }
    lds_barrier();

    // ---- conv2 weight gradient (as in front_bwd_kernel) ---------------------------------------------
    {
      const int t16 = lane & 15, q = t16 >> 2, pq = t16 & 3;
#pragma unroll 1
      for (int ks = 0; ks < 4; ++ks) {
        const int i_lo = 32 * ks + 8 * g + q, i_hi = i_lo + 4;
        const bool v_lo = i_lo < C2N, v_hi = i_hi < C2N;
        const int y_lo = v_lo ? i_lo / C2W : 0, x_lo = v_lo ? i_lo % C2W : 0;
        const int y_hi = v_hi ? i_hi / C2W : 0, x_hi = v_hi ? i_hi % C2W : 0;
        const uint32_t ya_lo = (v_lo ? ((y_lo + 4) * D2PW + x_lo + 4) : 0) * 32 + 8 * (pq & 1) + 16 * (pq >> 1);
        const uint32_t ya_hi = (v_hi ? ((y_hi + 4) * D2PW + x_hi + 4) : 0) * 32 + 8 * (pq & 1) + 16 * (pq >> 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int tile = wave + NWV * u;
          if (tile < 26) {
            const int i = tile & 1, nt = tile >> 1;
            int tap = 2 * nt + (pq >> 1);
            if (tap > 24) tap = 24;
            const int toff = (tap / 5) * A1PW + (tap % 5);
            const uint32_t xa_lo = ((y_lo * A1PW + x_lo) + toff) * 16 + 8 * (pq & 1);
            const uint32_t xa_hi = ((y_hi * A1PW + x_hi) + toff) * 16 + 8 * (pq & 1);
            const bf8 ah = tr_pair(d2h_a + ya_lo + D2HALF * i, d2h_a + ya_hi + D2HALF * i);
            const bf8 bh = tr_pair(a1h_a + xa_lo, a1h_a + xa_hi);
            bf8 al, bl;
            if (SPLIT == 3) {
              al = tr_pair(d2l_a + ya_lo + D2HALF * i, d2l_a + ya_hi + D2HALF * i);
              bl = tr_pair(a1l_a + xa_lo, a1l_a + xa_hi);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (SPLIT == 3) {
              wacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, wacc[u], 0, 0, 0);
              wacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, wacc[u], 0, 0, 0);
            }
            wacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, wacc[u], 0, 0, 0);
          }
        }
      }
    }
    // ---- conv2 backward-data (as in front_bwd_kernel: one image row of dA1 per wave) ---------------------
    {
      f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f}, dacc1 = dacc;
      if (wave < A1W) {
        const int xl = (r16 >= 4 && r16 < 12) ? r16 - 4 : (r16 < 4 ? 8 + r16 : 12);
        const int base = (wave + 5) * D2PW + xl + 5;
        const char *wb = ((SPLIT == 3 && r16 >= 8) ? wbl : wbh) + (g >> 1) * 256 + (r16 & 7) * 32 + (g & 1) * 16;
        const char *ab = d2h + (g >> 1) * D2HALF + (base - (4 * D2PW + 4)) * 32 + (g & 1) * 16;
        const long lo_a = d2l - d2h;
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
          const int arel = ((4 - tap / 5) * D2PW + (4 - tap % 5)) * 32;
          const bf8 b = *reinterpret_cast<const bf8 *>(wb + tap * 512);
          const bf8 ah = *reinterpret_cast<const bf8 *>(ab + arel);
          if (SPLIT == 3) {
            const bf8 al = *reinterpret_cast<const bf8 *>(ab + lo_a + arel);
            dacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b, dacc1, 0, 0, 0);
          }
          dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b, dacc, 0, 0, 0);
        }
        if (SPLIT == 3) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dacc[r] += __shfl_down(dacc[r], 8, 64) + dacc1[r];
        }
      }
      lds_barrier();  // dyb (aliased by dA1) is no longer read
      if (wave < A1W && r16 < 8)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * g + r;
          const int x = (row >= 4 && row < 12) ? row - 4 : (row < 4 ? 8 + row : 12);
          if (row < 13) dA1[(wave * A1W + x) * 8 + r16] = dacc[r];
        }
    }
    lds_barrier();

    // ---- pool1 + ReLU1 backward from the codes -> dC1 [8][196] ----------------------------------------
    for (int e = tid; e < C1N * 8; e += NTH) {
      const int co = e & 7, pix = e >> 3, y = pix / C1W, x = pix % C1W;
      float gsum = 0.f;
#pragma unroll
      for (int dyw = 0; dyw < 2; ++dyw)
#pragma unroll
        for (int dxw = 0; dxw < 2; ++dxw) {
          const int wy = y - dyw, wx = x - dxw;
          const bool ok = wy >= 0 && wy < A1W && wx >= 0 && wx < A1W;
          const int idx = (min(max(wy, 0), A1W - 1) * A1W + min(max(wx, 0), A1W - 1)) * 8 + co;
          const int code = k1s[idx];
          const float d = dA1[idx];
          gsum += (ok && code == (4 | (dyw * 2 + dxw))) ? d : 0.f;
        }
      dC1[co * C1N + pix] = gsum;
      db1 += gsum;
    }
    lds_barrier();

    // ---- conv1 weight gradient (as in front_bwd_kernel) ----------------------------------------------
    if (tid < nrowthr) {
      const int rp = tid % nrp, q = tid / nrp, ty = q % 5, ci = (q / 5) % cin, co = q / (5 * cin);
      for (int y = rp; y < C1W; y += nrp) {
        const float *dr = dC1 + co * C1N + y * C1W, *xr = xs + (ci * XPW + y + ty) * XPW;
        float dv[C1W], xv[XPW];
#pragma unroll
        for (int x = 0; x < C1W; ++x) dv[x] = dr[x];
#pragma unroll
        for (int x = 0; x < XPW; ++x) xv[x] = xr[x];
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
          float s1 = 0.f;
#pragma unroll
          for (int x = 0; x < C1W; ++x) s1 = fmaf(dv[x], xv[x + tx], s1);
          dw1[tx] += s1;
        }
      }
    }
    lds_barrier();
  }

  // ---- partial sums of this slice -> workspace (layout of front_bwd_kernel) ---------------------------
  const int PART = 32 * 8 * 25 + 32 + 8 * cin * 25 + 8;
  float *out = a.part + (long)blockIdx.x * PART;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int tile = wave + NWV * u;
    const int tap = 2 * (tile >> 1) + (r16 >> 3);
    if (tile < 26 && tap < 25)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((16 * (tile & 1) + 4 * g + r) * 8 + (r16 & 7)) * 25 + tap] = wacc[u][r];
  }
  __syncthreads();
  float *red = reinterpret_cast<float *>(lds);
  red[tid] = db2;
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
    for (int k = 0; k < NTH / 32; ++k) s += red[tid + 32 * k];
    out[32 * 8 * 25 + tid] = s;
  }
  __syncthreads();
  if (tid < nrowthr)
#pragma unroll
    for (int tx = 0; tx < 5; ++tx) red[tid * 5 + tx] = dw1[tx];
  __syncthreads();
  if (tid < nout) {
    const int q = tid / 5, tx = tid % 5;
    float s = 0.f;
    for (int y = 0; y < nrp; ++y) s += red[(q * nrp + y) * 5 + tx];
    out[32 * 8 * 25 + 32 + tid] = s;
  }
  __syncthreads();
  red[tid] = db1;
  __syncthreads();
  if (tid < 8) {
    float s = 0.f;
    for (int k = 0; k < NTH / 8; ++k) s += red[tid + 8 * k];
    out[32 * 8 * 25 + 32 + nout + tid] = s;
  }
}

// out[e] = sum_k part[k * stride + e], e < n: one wave per output, lanes stride the slices, fixed
// shuffle tree (deterministic)
// (the four gradient tensors are consecutive segments of a slice's partial vector: one launch covers them all)
struct FrontSums {
  const float *part;
  int nslice, stride, n;
  int end[4];     // exclusive end of segment i in the partial vector
  float *dst[4];  // its output tensor
};
__global__ __launch_bounds__(256) void front_slice_sum_kernel(FrontSums a) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= a.n) return;
  float s = 0.f;
  for (int k = lane; k < a.nslice; k += 64) s += a.part[(long)k * a.stride + e];
  s = wave_sum(s);
  if (lane == 0) {
    const int seg = e < a.end[0] ? 0 : e < a.end[1] ? 1 : e < a.end[2] ? 2 : 3;
    a.dst[seg][e - (seg ? a.end[seg - 1] : 0)] = s;
  }
}

size_t fwd_lds_bytes(int cin, bool pad = false) {
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t c1 = pad ? (C1W + 2) * (C1W + 2) : C1N, c2 = pad ? (C2W + 2) * (C2W + 2) : C2N;
  return r(4 * cin * XPW * XPW) + r(4 * (8 * cin * 25 + 8)) + r(4 * c1 * 8) + 2 * r(A1PW * A1PW * 16) +
         2 * r(KS2 * 32 * 32 * 2) + r(4 * c2 * 32);
}
size_t bwd_lds_bytes(int cin) {
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  return fwd_lds_bytes(cin, true) + 2 * r(25 * 8 * 32 * 2) + 2 * r(2 * D2HALF) + r(4 * ON * 32);
}
size_t bwd_saved_lds_bytes(int cin) {
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  return r(4 * cin * XPW * XPW) + 2 * r(A1PW * A1PW * 16) + 2 * r(25 * 8 * 32 * 2) + 2 * r(2 * D2HALF) + r(4 * ON * 32) + r(SV_K2) + r(SV_K1);
}
int front_slices(int P) { return P < 256 ? P : 256; }  // one 144 KB workgroup per CU

}  // namespace
}  // namespace crw

using namespace crw;

long long *g_front_stamps = nullptr;  // diagnostics (make STAMPS=1)

extern "C" {

#ifdef CRW_CONV_STAMPS
void crw_debug_front_stamps(long long *buf) { g_front_stamps = buf; }
#endif

// conv2 weight [32][8][5][5] fp32 -> forward planes [7][32][32] and backward planes [25][8][32] (bf16 hi[, lo])
int crw_enc_front_pack(const float *w2, uint16_t *fwd_hi, uint16_t *fwd_lo, uint16_t *bwd_hi, uint16_t *bwd_lo,
                       crw_stream_t stream) {
  clear_stale_error();
  if (!w2 || !fwd_hi || !bwd_hi || (fwd_lo == nullptr) != (bwd_lo == nullptr)) return CRW_EINVAL;
  hipLaunchKernelGGL(pack_w2_kernel, dim3(28), dim3(256), 0, (hipStream_t)stream, w2, fwd_hi, fwd_lo, bwd_hi, bwd_lo);
  return check_launch();
}

// threads per workgroup of the forward kernels: 512 (two workgroups per CU), CRW_FRONT_NT=1024 for the one-workgroup form
static int front_fwd_threads() {
  static const int nt = [] { const char *e = getenv("CRW_FRONT_NT"); return e && atoi(e) == 1024 ? 1024 : 512; }();
  return nt;
}

size_t crw_enc_front_saved_bytes(int P) { return P < 1 ? 0 : (size_t)P * SV_BYTES; }

int crw_enc_front_fwd(int split, const float *x, int P, int cin, const float *w1, const float *b1,
                      const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, uint16_t *y_hi, uint16_t *y_lo,
                      void *saved, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !w1 || !b1 || !w2_hi || !b2 || !y_hi || P < 1 || (cin != 1 && cin != 2) || (split != 1 && split != 3))
    return CRW_EINVAL;
  if (split == 3 && (!w2_lo || !y_lo)) return CRW_EINVAL;
  FrontArgs a{x, w1, b1, w2_hi, w2_lo, b2, y_hi, y_lo, P, cin, static_cast<char *>(saved)};
  const size_t lds = fwd_lds_bytes(cin);
  const int grid = P < 512 ? P : 512;  // two workgroups (60 KB of LDS each) per CU
  if (front_fwd_threads() == 1024) {
    if (split == 3) hipLaunchKernelGGL((front_fwd_kernel<3, 1024>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((front_fwd_kernel<1, 1024>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, a);
  } else {
    if (split == 3) hipLaunchKernelGGL((front_fwd_kernel<3, 512>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((front_fwd_kernel<1, 512>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
  }
  return check_launch();
}

// front end on patches of any size (h, w >= 7), forward only: x [P][cin][H][W] -> planes [P][(H-6)*(W-6)][32]
int crw_enc_front_fwd_map(int split, const float *x, int P, int cin, int H, int W, const float *w1, const float *b1,
                          const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, uint16_t *y_hi, uint16_t *y_lo,
                          crw_stream_t stream) {
  clear_stale_error();
  if (!x || !w1 || !b1 || !w2_hi || !b2 || !y_hi || P < 1 || (cin != 1 && cin != 2) || (split != 1 && split != 3) || H < 7 ||
      W < 7)
    return CRW_EINVAL;
  if (split == 3 && (!w2_lo || !y_lo)) return CRW_EINVAL;
  const int tx = (W - 6 + OW - 1) / OW, ty = (H - 6 + OW - 1) / OW;
  FrontMapArgs a{{x, w1, b1, w2_hi, w2_lo, b2, y_hi, y_lo, P, cin}, H, W, tx, ty};
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t lds = r(4 * cin * MXW * MXW) + r(4 * (8 * cin * 25 + 8)) + r(4 * MC1W * MC1W * 8) + 2 * r(A1PW * A1PW * 16) +
                     2 * r(KS2 * 32 * 32 * 2) + r(4 * C2N * 32);
  const long nwork = (long)P * tx * ty;
  const int grid = nwork < 512 ? (int)nwork : 512;  // two workgroups per CU
  const bool big = front_fwd_threads() == 1024;
  static bool attr[4] = {false, false, false, false};
  const void *fns[4] = {(const void *)front_fwd_map_kernel<1, 512>, (const void *)front_fwd_map_kernel<3, 512>,
                        (const void *)front_fwd_map_kernel<1, 1024>, (const void *)front_fwd_map_kernel<3, 1024>};
  const int which = (big ? 2 : 0) + (split == 3 ? 1 : 0);
  if (!attr[which] && lds > 64 * 1024) {
    if (hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return CRW_EHIP;
    attr[which] = true;
  }
  if (big) {
    if (split == 3) hipLaunchKernelGGL((front_fwd_map_kernel<3, 1024>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((front_fwd_map_kernel<1, 1024>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, a);
  } else {
    if (split == 3) hipLaunchKernelGGL((front_fwd_map_kernel<3, 512>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((front_fwd_map_kernel<1, 512>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
  }
  return check_launch();
}

size_t crw_enc_front_ws_bytes(int P, int cin) {
  if (P < 1 || (cin != 1 && cin != 2)) return 0;
  return (size_t)front_slices(P) * (32 * 8 * 25 + 32 + 8 * cin * 25 + 8) * sizeof(float);
}

// dy [P][100][32] fp32 -> dw2 [32][8][5][5], db2 [32], dw1 [8][cin][5][5], db1 [8]
int crw_enc_front_bwd(int split, const float *x, int P, int cin, const float *w1, const float *b1,
                      const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, const uint16_t *w2b_hi,
                      const uint16_t *w2b_lo, const float *dy, const void *saved, float *dw1, float *db1, float *dw2, float *db2,
                      void *ws, size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !w1 || !b1 || !w2_hi || !b2 || !w2b_hi || !dy || !dw1 || !db1 || !dw2 || !db2 || !ws || P < 1 ||
      (cin != 1 && cin != 2) || (split != 1 && split != 3))
    return CRW_EINVAL;
  if (split == 3 && (!w2_lo || !w2b_lo)) return CRW_EINVAL;
  if (ws_bytes < crw_enc_front_ws_bytes(P, cin)) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int nslice = front_slices(P), ppb = (P + nslice - 1) / nslice;
  FrontBwdArgs a{{x, w1, b1, w2_hi, w2_lo, b2, nullptr, nullptr, P, cin, const_cast<char *>(static_cast<const char *>(saved))},
                 w2b_hi, w2b_lo, dy, (float *)ws, ppb, g_front_stamps};
  static const char *force_rc = getenv("CRW_FRONT_RECOMPUTE");  // diagnostics / A-B: ignore the saved record
  if (saved && !(force_rc && force_rc[0] == '1')) {
    const size_t ldss = bwd_saved_lds_bytes(cin);
    static bool sattr3 = false, sattr1 = false;
    bool &sattr = split == 3 ? sattr3 : sattr1;
    const void *fn = split == 3 ? (const void *)front_bwd_saved_kernel<3> : (const void *)front_bwd_saved_kernel<1>;
    if (!sattr) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRW_EHIP;
      sattr = true;
    }
    if (split == 3) hipLaunchKernelGGL(front_bwd_saved_kernel<3>, dim3(nslice), dim3(NTH), ldss, s, a);
    else hipLaunchKernelGGL(front_bwd_saved_kernel<1>, dim3(nslice), dim3(NTH), ldss, s, a);
    CRW_TRY(check_launch());
    const int n2 = 32 * 8 * 25, n1 = 8 * cin * 25, PART = n2 + 32 + n1 + 8;
    FrontSums fs{(float *)ws, nslice, PART, n2 + 32 + n1 + 8, {n2, n2 + 32, n2 + 32 + n1, n2 + 32 + n1 + 8}, {dw2, db2, dw1, db1}};
    hipLaunchKernelGGL(front_slice_sum_kernel, dim3((fs.n + 3) / 4), dim3(256), 0, s, fs);
    return check_launch();
  }
  const size_t lds = bwd_lds_bytes(cin);
  static bool attr3 = false, attr1 = false;
  if (split == 3) {
    if (!attr3) {
      if (hipFuncSetAttribute((const void *)front_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
          hipSuccess) return CRW_EHIP;
      attr3 = true;
    }
    hipLaunchKernelGGL(front_bwd_kernel<3>, dim3(nslice), dim3(NTH), lds, s, a);
  } else {
    if (!attr1) {
      if (hipFuncSetAttribute((const void *)front_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
          hipSuccess) return CRW_EHIP;
      attr1 = true;
    }
    hipLaunchKernelGGL(front_bwd_kernel<1>, dim3(nslice), dim3(NTH), lds, s, a);
  }
  CRW_TRY(check_launch());
  const int n2 = 32 * 8 * 25, n1 = 8 * cin * 25, PART = n2 + 32 + n1 + 8;
  // the partial layout is [dW2 | db2 | dW1 | db1]; the outputs are four separate tensors
  float *part = (float *)ws;
  FrontSums fs{part, nslice, PART, n2 + 32 + n1 + 8, {n2, n2 + 32, n2 + 32 + n1, n2 + 32 + n1 + 8}, {dw2, db2, dw1, db1}};
  hipLaunchKernelGGL(front_slice_sum_kernel, dim3((fs.n + 3) / 4), dim3(256), 0, s, fs);
  return check_launch();
}


/* backward of the front end on patches of any size (h, w >= 7): x [P][cin][H][W], dy [P][H-6][W-6][32] fp32 (gradient of the
 * planes crw_enc_front_fwd_map produces) -> dw1, db1, dw2, db2.  ws: crw_enc_front_ws_bytes(P * tiles, cin). */
int crw_enc_front_bwd_map(int split, const float *x, int P, int cin, int H, int W, const float *w1, const float *b1,
                          const uint16_t *w2_hi, const uint16_t *w2_lo, const float *b2, const uint16_t *w2b_hi,
                          const uint16_t *w2b_lo, const float *dy, float *dw1, float *db1, float *dw2, float *db2, void *ws,
                          size_t ws_bytes, crw_stream_t stream) {
  clear_stale_error();
  if (!x || !w1 || !b1 || !w2_hi || !b2 || !w2b_hi || !dy || !dw1 || !db1 || !dw2 || !db2 || !ws || P < 1 ||
      (cin != 1 && cin != 2) || (split != 1 && split != 3) || H < 7 || W < 7)
    return CRW_EINVAL;
  if (split == 3 && (!w2_lo || !w2b_lo)) return CRW_EINVAL;
  const int tx = (W - 6 + OW - 1) / OW, ty = (H - 6 + OW - 1) / OW;
  const long units_l = (long)P * tx * ty;
  if (units_l > 0x7fffffffL) return CRW_EINVAL;
  const int units = (int)units_l;
  if (ws_bytes < crw_enc_front_ws_bytes(units, cin)) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int nslice = front_slices(units);
  const int upb = (units + nslice - 1) / nslice;
  nslice = (units + upb - 1) / upb;
  FrontBwdArgs a{{x, w1, b1, w2_hi, w2_lo, b2, nullptr, nullptr, P, cin, nullptr}, w2b_hi, w2b_lo, dy, (float *)ws, upb, nullptr, H, W, tx, ty};
  const void *fn = split == 3 ? (const void *)front_bwd_tile_kernel<3> : (const void *)front_bwd_tile_kernel<1>;
  static bool attr3 = false, attr1 = false;
  bool &attr = split == 3 ? attr3 : attr1;
  if (!attr) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return CRW_EHIP;
    attr = true;
  }
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t lds = r(4 * cin * TXW * TXW) + r(4 * (8 * cin * 25 + 8)) + r(4 * (TCW + 2) * (TCW + 2) * 8) + 2 * r(TAW * TAW * 16) +
                     2 * r(KS2 * 32 * 32 * 2) + r(4 * (C2W + 2) * (C2W + 2) * 32) + 2 * r(25 * 8 * 32 * 2) + 2 * r(2 * D2HALF) +
                     r(4 * ON * 32);
  if (split == 3) hipLaunchKernelGGL(front_bwd_tile_kernel<3>, dim3(nslice), dim3(NTH), lds, s, a);
  else hipLaunchKernelGGL(front_bwd_tile_kernel<1>, dim3(nslice), dim3(NTH), lds, s, a);
  CRW_TRY(check_launch());
  const int n2 = 32 * 8 * 25, n1 = 8 * cin * 25, PART = n2 + 32 + n1 + 8;
  FrontSums fs{(float *)ws, nslice, PART, n2 + 32 + n1 + 8, {n2, n2 + 32, n2 + 32 + n1, n2 + 32 + n1 + 8}, {dw2, db2, dw1, db1}};
  hipLaunchKernelGGL(front_slice_sum_kernel, dim3((fs.n + 3) / 4), dim3(256), 0, s, fs);
  return check_launch();
}

}  // extern "C"
