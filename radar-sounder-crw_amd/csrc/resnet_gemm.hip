// Matrix-core kernels of the Resnet encoder (reference: src/encoder.py:63-89 stem, :109-155 BasicBlock, :157-272 body).
//
// At 16x16 patches the Resnet's feature maps are tiny (18^2 -> 9^2 -> 5^2 -> 5^2 -> 3^2 -> 2^2 -> 1), so every
// convolution is computed as a GEMM ACROSS PATCHES: the M dimension of a tile is 128 patches, one "group" of the launch
// is one output pixel, and the reduction runs over the taps of that pixel that fall inside the map (a 3x3 convolution on
// the 1x1 map of layer4 is its centre tap only; nothing is multiplied by padding zeros).  Activations are channels-last
// bf16 planes [Ppad][pixels][C] (hi and lo: x = hi + lo to ~2^-17, products as Ah*Bh + Ah*Bl + Al*Bh on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation: fp32-grade results), Ppad = patches rounded up to 128 with ZERO rows, so
// that for a fixed output pixel the operand row of patch p is a few contiguous channel vectors of that patch's record:
//
//   rn_conv_kernel   C[p][g][n] = sum_seg sum_c A[p][seg_a(g,seg) + c] * B[n][seg_b(g,seg) + c]         ("NN", k-contiguous operands)
//       forward        g = output pixel, seg = in-range taps, A = input planes, B = weights [cout][tap][cin]
//       backward-data  g = input pixel,  seg = (output pixel, tap) pairs that reach it, A = dZ planes, B = weights [cin][tap][cout]
//       stem forward   7x7/2 on the 3-channel map, stored zero-padded with 4 channels: seg = kernel row, 32 contiguous values
//       stem backward  g = map row, ONE contiguous segment of dZ rows against a Toeplitz expansion of the 7x7 weights
//       the epilogue also emits per-tile column sums and sums of squares: the BatchNorm batch statistics cost no extra pass
//   rn_wgrad_kernel  dW[tap][r][c] = sum_{pairs(tap)} sum_p X[p][pa(pair) + r] * dZ[p][pb(pair) + c]      ("TN", reduction over patches)
//       both operands are patch-major, so their fragments come from ds_read_b64_tr_b16 (hardware transpose);
//       split over patch slices, partial slabs added in a fixed order by rn_wgrad_reduce_kernel (no float atomics).
//
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of lane-linear images with the bank swizzle on the source
// address, counted vmcnt + raw s_barrier -- the staging scheme of gemm_bf16.hip, here with gathered / strided operands.
#include <algorithm>

#include "crw_common.h"
#include "resnet.h"

namespace crw {
namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char *lds_cp;

__device__ inline void glds16(const uint16_t *g, char *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ inline uint32_t lds_addr_of(const char *p) { return (uint32_t)(uintptr_t)(lds_cp)p; }
// inline asm: with the builtin hipcc drains every in-flight LDS-DMA in front of the read (gemm_bf16.hip)
__device__ inline s4v tr_read(uint32_t lds_addr) {
  s4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}

// ---- k-contiguous images: [rows][BK], 2*BK-byte rows, fragments by ds_read_b128 ---------------------------------------
template <int BK>
__device__ inline int kc_sw(int row) { return BK == 64 ? (row & 7) : ((row & 8) ? 3 : 0); }
template <int BK>
__device__ inline int kc_off(int row, int chunk) { return row * (2 * BK) + 16 * (chunk ^ kc_sw<BK>(row)); }
template <int BK>
__device__ inline bf8 kc_frag(const char *img, int rb, int s, int lane) {
  return *reinterpret_cast<const bf8 *>(img + kc_off<BK>(rb + (lane & 15), 4 * s + (lane >> 4)));
}

// ---- r-contiguous images: [BK k-rows][TB], 2*TB-byte rows, fragments by two ds_read_b64_tr_b16 ------------------------
// the chunk swizzle keeps the 32-byte windows that the eight k-rows of a half-wave read on distinct banks
template <int TB>
__device__ inline int rc_sw(int row) {
  return TB == 64 ? ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1) : (((row & 3) << 2) | ((row >> 2) & 3));
}
template <int TB>
__device__ inline int rc_off(int row, int chunk) { return row * (TB * 2) + 16 * (chunk ^ rc_sw<TB>(row)); }
template <int TB>
__device__ inline bf8 rc_frag(const char *img, int rb, int s, int lane) {
  const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
  const int row = 32 * s + 8 * g + q;
  const int chunk = (rb >> 3) + (p >> 1);
  const uint32_t base = lds_addr_of(img);
  const s4v lo = tr_read(base + rc_off<TB>(row, chunk) + 8 * (p & 1));
  const s4v hi = tr_read(base + rc_off<TB>(row + 4, chunk) + 8 * (p & 1));
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8, v);
}

__device__ inline f32x4 mfma(bf8 a, bf8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// bijective block -> work remap: blocks b, b+8, b+16, ... run on one XCD (speed only); every XCD gets a contiguous range of
// the linear work index, so the blocks that share a patch tile (all groups / column tiles of it) meet in one 4 MiB L2
__device__ inline int xcd_linear(int bid, int total) {
  const int q = total >> 3, r = total & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}

// =============================================================================================== NN: conv forward / backward-data
constexpr int RN_MAXSEG = 64;

template <int TN, int BK, int NSTAGE, int TM_ = 128>
struct NNCfg {
  // TM x TN tile, one wave per 64 x TN/2 sub-tile: 4 waves (2 x 2) at 128 rows, 8 waves (4 x 2) at 256
  static constexpr int TM = TM_, WAVES = TM / 32, FM = 4, FN = TN / 32;
  static constexpr int IMG_A = TM * 2 * BK, IMG_B = TN * 2 * BK, STAGE = 2 * IMG_A + 2 * IMG_B;
  static constexpr int PIECES_A = TM * BK / 512, PIECES_B = TN * BK / 512;        // 1 KiB pieces per image
  static constexpr int PA = PIECES_A / WAVES, PB = (PIECES_B + WAVES - 1) / WAVES;  // per wave (the last B round may be partial)
  static constexpr int G = 2 * PA + 2 * PB;                                      // LDS-DMA instructions per wave and k-tile
  static constexpr int TABLE = 2 * RN_MAXSEG * 4 + 16;
  static constexpr size_t LDS = (size_t)NSTAGE * STAGE + TABLE;
  static_assert(PA >= 1 && PIECES_A % WAVES == 0, "A pieces per wave");
  static_assert(PIECES_B % WAVES == 0 || NSTAGE == 2, "a partial B round needs the uncounted vmcnt(0) of the two-stage ring");
};

// the segments of group g: seg_a[] = element offset of the segment inside an A row, seg_b[] = inside a B row
__device__ inline void rn_segments(const RnConvArgs &a, int g, int lane, int *seg_a, int *seg_b, int *hdr) {
  int nseg = 0, seglen = a.Cs;
  if (a.mode == RN_MODE_FWD || a.mode == RN_MODE_BWD) {
    const int T = a.KH * a.KW;
    const int dy = g / a.Wd, dx = g % a.Wd;
    bool valid = false;
    int oa = 0;
    if (lane < T) {
      const int ky = lane / a.KW, kx = lane % a.KW;
      if (a.mode == RN_MODE_FWD) {  // destination = output pixel, source = input map
        const int iy = dy * a.S + ky - a.PAD, ix = dx * a.S + kx - a.PAD;
        valid = iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws;
        oa = (iy * a.Ws + ix) * a.Cs;
      } else {                      // destination = input pixel, source = dZ map (output pixels)
        const int ty = dy + a.PAD - ky, tx = dx + a.PAD - kx;
        valid = ty >= 0 && tx >= 0 && ty % a.S == 0 && tx % a.S == 0 && ty / a.S < a.Hs && tx / a.S < a.Ws;
        oa = ((ty / a.S) * a.Ws + tx / a.S) * a.Cs;
      }
    }
    const unsigned long long m = __ballot(valid);
    nseg = __popcll(m);
    if (valid) {
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      seg_a[pos] = oa;
      seg_b[pos] = lane * a.Cs;
    }
  } else if (a.mode == RN_MODE_STEM_FWD) {  // rows of the 7x7 kernel (+ one all-zero row): 32 contiguous values each
    const int dy = g / a.Wd, dx = g % a.Wd;
    nseg = 8;
    seglen = 32;
    if (lane < 8) {
      seg_a[lane] = ((dy * a.S + lane) * a.Ws + dx * a.S) * a.Cs;
      seg_b[lane] = lane * 32;
    }
  } else {  // RN_MODE_STEM_BWD: destination = map row g, ONE segment = the dZ rows whose kernel reaches it
    int o0 = (g + a.PAD - a.KH + a.S) / a.S;  // ceil((g + PAD - KH + 1) / S) for a non-negative numerator, clamped below
    if (g + a.PAD - a.KH + 1 <= 0) o0 = 0;
    int o1 = (g + a.PAD) / a.S;
    if (o1 > a.Hs - 1) o1 = a.Hs - 1;
    nseg = 1;
    seglen = (o1 - o0 + 1) * a.Ws * a.Cs;
    if (lane == 0) {
      seg_a[0] = o0 * a.Ws * a.Cs;
      seg_b[0] = 0;
    }
  }
  if (lane == 0) {
    hdr[0] = nseg;
    hdr[1] = nseg * seglen;                                   // K
    hdr[2] = nseg == 1 ? 30 : 31 - __builtin_clz(seglen);     // k >> shift = segment
  }
}

// LDS the EPI = 2 epilogue needs: the fp32 tile with TN + 4 floats per row, then the per-thread column sums [TM/64][3][row lanes][TN]
template <int TN, int TM, int NT = TM * 2>
constexpr size_t rn_epi2_lds() {
  const size_t tile = (size_t)TM * (TN + 4) * 4, sums = (size_t)(TM / 64) * 3 * (NT / (TN / 4)) * TN * 4;
  return tile > sums ? tile : sums;
}


// The EPI = 2 epilogue: the tile goes through LDS (the ring is free) and comes back row-contiguous for NT threads -- the waves that
// hold accumulators (has_acc) write them, every wave of the workgroup takes part in the row pass.
template <int TN, int TM, int NT, int FM, int FN>
__device__ inline void rn_epi2_staged(const RnConvArgs &a, char *lds, const f32x4 (&acc)[FM][FN], bool has_acc, int wm, int wn, int lane,
                                      int m0, int n0, int g, int mt) {
  // The tile goes through LDS (the ring is free now) and comes back row-contiguous: float4 stores of whole 128-byte lines and
  // float4 / 8-byte loads of the consuming layer's Z and activation planes.  (Straight from the MFMA layout -- 4-byte
  // accesses, 16 lanes per row segment -- the extra loads cost as much as the separate reduce pass they replace.)
  constexpr int LDT = TN + 4;                   // row stride in floats: the four row groups of a wave hit disjoint banks
  constexpr int TPR = TN / 4, RPP = NT / TPR;   // threads per row (one float4 each), rows per pass
  constexpr int NH = TM / 64;                   // 64-row pieces of the tile: one partial row of sums each
  constexpr int NB = NT == 512 ? 2 : ((64 / RPP) % 4 == 0 ? 4 : 64 / RPP);  // passes whose loads are in flight together (512 threads: 128-register budget)
  static_assert(64 % RPP == 0 && (64 / RPP) % NB == 0, "passes per 64-row piece");
  float *tile = reinterpret_cast<float *>(lds);
  const int nsum = a.red_zd ? 3 : 2;
  __syncthreads();  // every wave has read its last fragments
  if (has_acc) {
    const int gq = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(wm + 16 * i + 4 * gq + r) * LDT + wn + 16 * j + c16] = acc[i][j][r];
  }
  __syncthreads();
  const int c4 = threadIdx.x % TPR, rl = threadIdx.x / TPR;
  const int col = n0 + 4 * c4;
  const float4 mean = *reinterpret_cast<const float4 *>(a.red_coef + 2 * a.N + col);
  const float4 istd = *reinterpret_cast<const float4 *>(a.red_coef + 3 * a.N + col);
  float4 mean_d = float4{0.f, 0.f, 0.f, 0.f}, istd_d = mean_d;
  if (a.red_zd) {
    mean_d = *reinterpret_cast<const float4 *>(a.red_coefd + 2 * a.N + col);
    istd_d = *reinterpret_cast<const float4 *>(a.red_coefd + 3 * a.N + col);
  }
  const long base = (long)m0 * a.ldc + (long)g * a.N + col;
  float sums[NH][3][4];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) sums[h][k][e] = 0.f;
#pragma unroll 1
    for (int p0 = 0; p0 < 64 / RPP; p0 += NB) {
      float4 prev[NB], z[NB], zd[NB];
      uint2 mk[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const long idx = base + (long)(64 * h + (p0 + u) * RPP + rl) * a.ldc;
        prev[u] = a.accumulate ? *reinterpret_cast<const float4 *>(a.out + idx) : float4{0.f, 0.f, 0.f, 0.f};
        mk[u] = *reinterpret_cast<const uint2 *>(a.red_mask + idx);
        z[u] = *reinterpret_cast<const float4 *>(a.red_z + idx);
        zd[u] = a.red_zd ? *reinterpret_cast<const float4 *>(a.red_zd + idx) : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int row = 64 * h + (p0 + u) * RPP + rl;
        float4 v = *reinterpret_cast<const float4 *>(tile + row * LDT + 4 * c4);
        v.x += prev[u].x; v.y += prev[u].y; v.z += prev[u].z; v.w += prev[u].w;
        *reinterpret_cast<float4 *>(a.out + base + (long)row * a.ldc) = v;
        const float gv[4] = {(mk[u].x & 0xffffu) ? v.x : 0.f, (mk[u].x >> 16) ? v.y : 0.f, (mk[u].y & 0xffffu) ? v.z : 0.f,
                             (mk[u].y >> 16) ? v.w : 0.f};
        const float zz[4] = {z[u].x, z[u].y, z[u].z, z[u].w}, zzd[4] = {zd[u].x, zd[u].y, zd[u].z, zd[u].w};
        const float mm[4] = {mean.x, mean.y, mean.z, mean.w}, ii[4] = {istd.x, istd.y, istd.z, istd.w};
        const float mmd[4] = {mean_d.x, mean_d.y, mean_d.z, mean_d.w}, iid[4] = {istd_d.x, istd_d.y, istd_d.z, istd_d.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sums[h][0][e] += gv[e];
          sums[h][1][e] += gv[e] * ((zz[e] - mm[e]) * ii[e]);
          sums[h][2][e] += gv[e] * ((zzd[e] - mmd[e]) * iid[e]);
        }
      }
    }
  }
  __syncthreads();  // the tile has been read
  float *red = tile;  // [NH pieces][3 sums][RPP row lanes][TN]
  static_assert(rn_epi2_lds<TN, TM, NT>() >= (size_t)NH * 3 * RPP * TN * 4 && rn_epi2_lds<TN, TM, NT>() >= (size_t)TM * LDT * 4, "staging space");
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      *reinterpret_cast<float4 *>(red + ((h * 3 + k) * RPP + rl) * TN + 4 * c4) =
          float4{sums[h][k][0], sums[h][k][1], sums[h][k][2], sums[h][k][3]};
  __syncthreads();
  for (int o = threadIdx.x; o < NH * nsum * TN; o += NT) {  // [row][nsum][N]: the layout of rn_bn_bwd_reduce_kernel's partials
    const int c = o % TN, k = (o / TN) % nsum, h = o / (TN * nsum);
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < RPP; ++q) t += red[((h * 3 + k) * RPP + q) * TN + c];
    a.red_part[(((long)(mt * NH + h) * a.G + g) * nsum + k) * a.N + n0 + c] = t;
  }
}

// EPI: what the epilogue does besides storing the fp32 tile -- 0: (+ bias) and the forward statistics partials; 1: adds the tile to
// what `out` holds (second gradient of a junction); 2: that (optionally) plus the BatchNorm-backward sums of the consuming layer.
// Separate kernels: as run-time variants of one kernel they cost every launch its second workgroup per CU (104 -> 256 registers).
template <int TN, int BK, int NSTAGE, int EPI, int TM = 128>
__global__ __launch_bounds__(TM * 2) void rn_conv_kernel(RnConvArgs a) {  // TM = 128: two workgroups per CU (short k-loops: they cover each other)
  using C = NNCfg<TN, BK, NSTAGE, TM>;
  constexpr int NT = TM * 2;  // threads
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int *seg_a = reinterpret_cast<int *>(lds + NSTAGE * C::STAGE);
  int *seg_b = seg_a + RN_MAXSEG;
  int *hdr = seg_b + RN_MAXSEG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = a.N / TN;
  const int lin = xcd_linear(blockIdx.x, gridDim.x);
  const int per_mt = ntiles * a.G;
  const int mt = lin / per_mt, rest = lin % per_mt;
  const int g = rest / ntiles, nt = rest % ntiles;
  const int m0 = mt * C::TM, n0 = nt * TN;

  if (wave == 0) rn_segments(a, g, lane, seg_a, seg_b, hdr);
  __syncthreads();
  const int K = hdr[1], shift = hdr[2];
  const int nkt = K / BK;
  const unsigned kmask = (1u << shift) - 1u;

  const uint16_t *Bh = a.b_hi + (long)g * a.b_group_stride, *Bl = a.b_lo + (long)g * a.b_group_stride;
  // per-lane constants of the staging: which 16-byte chunk of a k-tile this lane fetches (swizzled) and its first row
  constexpr int LPR = BK / 8, RPP = 64 / LPR;  // lanes per row, rows per 1 KiB piece
  const int lrow = lane / LPR;
  const int gchunk = (lane % LPR) ^ kc_sw<BK>(lrow);
  const long a_row0 = (long)(m0 + lrow) * a.lda, b_row0 = (long)(n0 + lrow) * a.ldb;

  auto stage = [&](int t, int buf) {
    char *base = lds + buf * C::STAGE;
    const unsigned k = (unsigned)(t * BK + 8 * gchunk);
    const int sg = (int)(k >> shift), within = (int)(k & kmask);
    const int acol = seg_a[sg] + within, bcol = seg_b[sg] + within;
#pragma unroll
    for (int i = 0; i < C::PA; ++i) {
      const int piece = C::WAVES * i + wave;
      const long off = a_row0 + (long)(RPP * piece) * a.lda + acol;
      glds16(a.a_hi + off, base + piece * 1024);
      glds16(a.a_lo + off, base + C::IMG_A + piece * 1024);
    }
#pragma unroll
    for (int i = 0; i < C::PB; ++i) {
      const int piece = C::WAVES * i + wave;
      if (C::PIECES_B % C::WAVES == 0 || piece < C::PIECES_B) {
        const long off = b_row0 + (long)(RPP * piece) * a.ldb + bcol;
        glds16(Bh + off, base + 2 * C::IMG_A + piece * 1024);
        glds16(Bl + off, base + 2 * C::IMG_A + C::IMG_B + piece * 1024);
      }
    }
  };

  f32x4 acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (TN / 2);

  for (int t = 0; t < NSTAGE - 1 && t < nkt; ++t) stage(t, t);
  for (int t = 0; t < nkt; ++t) {
    if (NSTAGE >= 3 && t + 1 < nkt)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::G) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t landed for every wave; the buffer of tile t-1 is free
    if (t + NSTAGE - 1 < nkt) stage(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
    const char *base = lds + (t % NSTAGE) * C::STAGE;
#pragma unroll
    for (int s = 0; s < BK / 32; ++s) {
      bf8 b[C::FN], bl[C::FN], af[C::FM], al[C::FM];
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        b[j] = kc_frag<BK>(base + 2 * C::IMG_A, wn + 16 * j, s, lane);
        bl[j] = kc_frag<BK>(base + 2 * C::IMG_A + C::IMG_B, wn + 16 * j, s, lane);
      }
#pragma unroll
      for (int i = 0; i < C::FM; ++i) {
        af[i] = kc_frag<BK>(base, wm + 16 * i, s, lane);
        al[i] = kc_frag<BK>(base + C::IMG_A, wm + 16 * i, s, lane);
      }
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j) {
          acc[i][j] = mfma(al[i], b[j], acc[i][j]);
          acc[i][j] = mfma(af[i], bl[j], acc[i][j]);
          acc[i][j] = mfma(af[i], b[j], acc[i][j]);
        }
    }
  }

  // epilogue: fp32 tile (+ bias, or added to what `out` holds: the second gradient of a junction) and per-tile column sums --
  // forward: sum / sum of squares (BatchNorm statistics); backward-data: the BatchNorm backward sums of the layer that CONSUMES
  // this gradient (g = tile where its activation plane is positive: sum g, sum g * xhat [, sum g * xhat of the shortcut's BatchNorm])
  // Three straight-line variants under wave-uniform tests (conditions inside the store loops made every variant wait for each
  // element's load before its store: +35 % on all launches).
  const long crow0 = (long)(m0 + wm + (lane >> 4) * 4) * a.ldc + (long)g * a.N + n0 + wn + (lane & 15);
  const long prow = (long)(mt * (TM / 64) + (wave >> 1)) * a.G + g;
  auto colsum = [&](float v) {  // over the wave's 64 rows: lanes 0..15 end up with the column totals
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
  };
  if constexpr (EPI == 2) {
    rn_epi2_staged<TN, TM, NT, C::FM, C::FN>(a, lds, acc, true, wm, wn, lane, m0, n0, g, mt);
  } else if constexpr (EPI == 1) {
#pragma unroll
    for (int j = 0; j < C::FN; ++j) {
      float prev[C::FM][4];
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) prev[i][r] = a.out[crow0 + (long)(16 * i + r) * a.ldc + 16 * j];
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.out[crow0 + (long)(16 * i + r) * a.ldc + 16 * j] = acc[i][j][r] + prev[i][r];
    }
  } else {
#pragma unroll
    for (int j = 0; j < C::FN; ++j) {
      const int col = n0 + wn + 16 * j + (lane & 15);
      const float bias = a.bias ? a.bias[col] : 0.f;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r] + bias;
          a.out[crow0 + (long)(16 * i + r) * a.ldc + 16 * j] = v;
          s1 += v;
          s2 += v * v;
        }
      if (a.part) {
        s1 = colsum(s1);
        s2 = colsum(s2);
        if (lane < 16) reinterpret_cast<float2 *>(a.part)[prow * a.N + col] = float2{s1, s2};
      }
    }
  }
}

template <int TN, int BK, int NSTAGE, int EPI, int TM = 128>
int launch_conv_epi(const RnConvArgs &a, hipStream_t s) {
  using C = NNCfg<TN, BK, NSTAGE, TM>;
  constexpr size_t staged = rn_epi2_lds<TN, TM>();  // EPI 2 passes the output tile through LDS
  constexpr size_t lds_bytes = (EPI == 2 && staged > C::LDS) ? staged : C::LDS;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)rn_conv_kernel<TN, BK, NSTAGE, EPI, TM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr_set = true;
  }
  const int total = a.mtiles / (TM / 128) * (a.N / TN) * a.G;
  hipLaunchKernelGGL((rn_conv_kernel<TN, BK, NSTAGE, EPI, TM>), dim3(total), dim3(TM * 2), lds_bytes, s, a);
  return check_launch();
}

// ---- the same product with the roles split over the waves of a 512-thread workgroup ----------------------------------------------
// Waves 0-3 run the MFMAs (the 2 x 2 layout of rn_conv_kernel), waves 4-7 only issue the LDS-DMA of the ring, NSTAGE-1 k-tiles
// ahead.  An LDS-DMA piece costs its wave 60-180 issue cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"): 8 pieces per
// k-tile are as long as the k-tile's 48 MFMAs, and in rn_conv_kernel every wave pays both in turn.  One barrier per k-tile as
// before: the loaders arrive when tile t has landed, the consumers when they have read tile t-1.
template <int TN, int BK, int NSTAGE, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void rn_conv_spec_kernel(RnConvArgs a) {  // two workgroups per CU
  using C = NNCfg<TN, BK, NSTAGE>;
  static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int *seg_a = reinterpret_cast<int *>(lds + NSTAGE * C::STAGE);
  int *seg_b = seg_a + RN_MAXSEG;
  int *hdr = seg_b + RN_MAXSEG;
  const int lane = threadIdx.x & 63, wave8 = threadIdx.x >> 6;
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int ntiles = a.N / TN;
  const int lin = xcd_linear(blockIdx.x, gridDim.x);
  const int per_mt = ntiles * a.G;
  const int mt = lin / per_mt, rest = lin % per_mt;
  const int g = rest / ntiles, nt = rest % ntiles;
  const int m0 = mt * C::TM, n0 = nt * TN;

  if (wave8 == 0) rn_segments(a, g, lane, seg_a, seg_b, hdr);
  __syncthreads();
  const int K = hdr[1], shift = hdr[2];
  const int nkt = K / BK;

  if (loader) {
    const unsigned kmask = (1u << shift) - 1u;
    const uint16_t *Bh = a.b_hi + (long)g * a.b_group_stride, *Bl = a.b_lo + (long)g * a.b_group_stride;
    constexpr int LPR = BK / 8, RPP = 64 / LPR;
    const int lrow = lane / LPR;
    const int gchunk = (lane % LPR) ^ kc_sw<BK>(lrow);
    const long a_row0 = (long)(m0 + lrow) * a.lda, b_row0 = (long)(n0 + lrow) * a.ldb;
    auto stage = [&](int t, int buf) {
      char *base = lds + buf * C::STAGE;
      const unsigned k = (unsigned)(t * BK + 8 * gchunk);
      const int sg = (int)(k >> shift), within = (int)(k & kmask);
      const int acol = seg_a[sg] + within, bcol = seg_b[sg] + within;
#pragma unroll
      for (int i = 0; i < C::PA; ++i) {
        const int piece = C::WAVES * i + wave;
        const long off = a_row0 + (long)(RPP * piece) * a.lda + acol;
        glds16(a.a_hi + off, base + piece * 1024);
        glds16(a.a_lo + off, base + C::IMG_A + piece * 1024);
      }
#pragma unroll
      for (int i = 0; i < C::PB; ++i) {
        const int piece = C::WAVES * i + wave;
        const long off = b_row0 + (long)(RPP * piece) * a.ldb + bcol;
        glds16(Bh + off, base + 2 * C::IMG_A + piece * 1024);
        glds16(Bl + off, base + 2 * C::IMG_A + C::IMG_B + piece * 1024);
      }
    };
    for (int t = 0; t < NSTAGE - 1 && t < nkt; ++t) stage(t, t);
    for (int t = 0; t < nkt; ++t) {
      const int ahead = min(NSTAGE - 2, nkt - 1 - t);  // k-tiles issued after tile t
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::G) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::G) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // tile t has landed; the consumers have left the buffer of tile t-1
      if (t + NSTAGE - 1 < nkt) stage(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
    }
    if constexpr (EPI == 2) {  // the loaders take their share of the row pass
      f32x4 none[C::FM][C::FN];
      rn_epi2_staged<TN, C::TM, 512, C::FM, C::FN>(a, lds, none, false, 0, 0, lane, m0, n0, g, mt);
    }
    return;
  }

  f32x4 acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (TN / 2);
  for (int t = 0; t < nkt; ++t) {
    __builtin_amdgcn_s_barrier();
    const char *base = lds + (t % NSTAGE) * C::STAGE;
#pragma unroll
    for (int s = 0; s < BK / 32; ++s) {
      bf8 b[C::FN], bl[C::FN], af[C::FM], al[C::FM];
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        b[j] = kc_frag<BK>(base + 2 * C::IMG_A, wn + 16 * j, s, lane);
        bl[j] = kc_frag<BK>(base + 2 * C::IMG_A + C::IMG_B, wn + 16 * j, s, lane);
      }
#pragma unroll
      for (int i = 0; i < C::FM; ++i) {
        af[i] = kc_frag<BK>(base, wm + 16 * i, s, lane);
        al[i] = kc_frag<BK>(base + C::IMG_A, wm + 16 * i, s, lane);
      }
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j) {
          acc[i][j] = mfma(al[i], b[j], acc[i][j]);
          acc[i][j] = mfma(af[i], bl[j], acc[i][j]);
          acc[i][j] = mfma(af[i], b[j], acc[i][j]);
        }
    }
  }
  if constexpr (EPI == 2) {
    rn_epi2_staged<TN, C::TM, 512, C::FM, C::FN>(a, lds, acc, true, wm, wn, lane, m0, n0, g, mt);
    return;
  }
  const long crow0 = (long)(m0 + wm + (lane >> 4) * 4) * a.ldc + (long)g * a.N + n0 + wn + (lane & 15);
  const long prow = (long)(mt * 2 + (wave >> 1)) * a.G + g;
  if constexpr (EPI == 1) {
#pragma unroll
    for (int j = 0; j < C::FN; ++j) {
      float prev[C::FM][4];
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) prev[i][r] = a.out[crow0 + (long)(16 * i + r) * a.ldc + 16 * j];
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.out[crow0 + (long)(16 * i + r) * a.ldc + 16 * j] = acc[i][j][r] + prev[i][r];
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < C::FN; ++j) {
    const int col = n0 + wn + 16 * j + (lane & 15);
    const float bias = a.bias ? a.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[i][j][r] + bias;
        a.out[crow0 + (long)(16 * i + r) * a.ldc + 16 * j] = v;
        s1 += v;
        s2 += v * v;
      }
    if (a.part) {
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16);
      s2 += __shfl_xor(s2, 32);
      if (lane < 16) reinterpret_cast<float2 *>(a.part)[prow * a.N + col] = float2{s1, s2};
    }
  }
}

template <int TN, int BK, int NSTAGE, int EPI>
int launch_conv_spec_epi(const RnConvArgs &a, hipStream_t s) {
  using C = NNCfg<TN, BK, NSTAGE>;
  constexpr size_t staged = rn_epi2_lds<TN, C::TM, 512>();
  constexpr size_t lds_bytes = (EPI == 2 && staged > C::LDS) ? staged : C::LDS;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)rn_conv_spec_kernel<TN, BK, NSTAGE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr_set = true;
  }
  const int total = a.mtiles * (a.N / TN) * a.G;
  hipLaunchKernelGGL((rn_conv_spec_kernel<TN, BK, NSTAGE, EPI>), dim3(total), dim3(512), lds_bytes, s, a);
  return check_launch();
}

template <int TN, int BK, int NSTAGE>
int launch_conv_spec(const RnConvArgs &a, hipStream_t s) {
  if (a.red_mask) return launch_conv_spec_epi<TN, BK, NSTAGE, 2>(a, s);
  if (a.accumulate) return launch_conv_spec_epi<TN, BK, NSTAGE, 1>(a, s);
  return launch_conv_spec_epi<TN, BK, NSTAGE, 0>(a, s);
}

template <int TN, int BK, int NSTAGE, int TM = 128>
int launch_conv_cfg(const RnConvArgs &a, hipStream_t s) {
  if (a.red_mask) return launch_conv_epi<TN, BK, NSTAGE, 2, TM>(a, s);
  if (a.accumulate) return launch_conv_epi<TN, BK, NSTAGE, 1, TM>(a, s);
  return launch_conv_epi<TN, BK, NSTAGE, 0, TM>(a, s);
}

// =============================================================================================== TN: weight gradients
// the (input pixel, output pixel) table of a tap lives in LDS behind the ring: 2 * maxpair ints, maxpair = the layer's output pixels
// rounded up to 64 (RnWgradArgs::maxpair; 16 x 16 patches: at most 128 -- the stem's 81; 32 x 32: 289; the cap leaves the
// 128 x 128 kernel its two 32 KB stages + 32 KB of table)
constexpr int RN_MAXPAIR_CAP = 4096;

template <int TM, int TN, int BK_>
struct TNCfg {
  static constexpr int BK = BK_, WAVES = 4, FM = TM / 32, FN = TN / 32, NSTAGE = 2;
  static constexpr int IMG_A = TM * 2 * BK, IMG_B = TN * 2 * BK, STAGE = 2 * IMG_A + 2 * IMG_B;
  static constexpr int PA = TM * BK / 512 / WAVES, PB = TN * BK / 512 / WAVES;
  static constexpr size_t RING = (size_t)NSTAGE * STAGE;
  static constexpr size_t lds(int maxpair) { return RING + (size_t)2 * maxpair * 4 + 16; }
};

// pairs of `tap`: pa[] = element offset of the input pixel inside an X row, pb[] = of the output pixel inside a dZ row
__device__ inline void rn_pairs(const RnWgradArgs &a, int tap, int tid, int *pa, int *pb, int *hdr) {
  // one wave (64 lanes), one round per 64 output pixels
  int count = 0;
  const int npix = a.Hout * a.Wout;
  for (int base = 0; base < npix; base += 64) {
    const int o = base + tid;
    bool valid = false;
    int va = 0;
    if (o < npix) {
      const int oy = o / a.Wout, ox = o % a.Wout;
      if (a.mode == RN_MODE_STEM_FWD) {  // every output pixel; the kernel rows are the r-segments of the A operand
        valid = true;
        va = ((oy * a.St) * a.Win + ox * a.St) * a.Cin;
      } else {
        const int ky = tap / a.KW, kx = tap % a.KW;
        const int iy = oy * a.St + ky - a.PAD, ix = ox * a.St + kx - a.PAD;
        valid = iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        va = (iy * a.Win + ix) * a.Cin;
      }
    }
    const unsigned long long m = __ballot(valid);
    if (valid) {
      const int pos = count + __popcll(m & ((1ull << tid) - 1ull));
      pa[pos] = va;
      pb[pos] = o * a.Cout;
    }
    count += __popcll(m);
  }
  if (tid == 0) hdr[0] = count;
}

template <int TM, int TN, int BK_>
__global__ __launch_bounds__(256) void rn_wgrad_kernel(RnWgradArgs a) {
  using C = TNCfg<TM, TN, BK_>;
  constexpr int BK = C::BK;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int *pa = reinterpret_cast<int *>(lds + C::RING);
  int *pb = pa + a.maxpair;
  int *hdr = pb + a.maxpair;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mtiles = a.Mtot / TM, ntiles = a.Ntot / TN;
  // every (tap, tile) of a patch slice reads the same patches: one XCD takes whole slices, so its L2 serves the re-reads
  const int per_slice = a.ntv * mtiles * ntiles;
  const int work = xcd_linear(blockIdx.x, gridDim.x);
  const int slice = work / per_slice, lin = work % per_slice;
  const int tv = lin / (mtiles * ntiles), mt = (lin / ntiles) % mtiles, nt = lin % ntiles;
  const int tap = a.tapv[tv];

  if (wave == 0) rn_pairs(a, tap, lane, pa, pb, hdr);
  __syncthreads();
  const int npairs = hdr[0];
  const long kt_total = (long)npairs * a.ktiles_p * (64 / BK);
  const int kt0 = (int)(kt_total * slice / a.S), kt1 = (int)(kt_total * (slice + 1) / a.S);

  // staging constants: r-contiguous images, LPR lanes per k-row
  constexpr int LPRA = TM / 8, RPPA = 64 / LPRA, LPRB = TN / 8, RPPB = 64 / LPRB;
  const int r0 = mt * TM, c0 = nt * TN;

  auto stage = [&](int kt, int buf) {
    char *base = lds + buf * C::STAGE;
    // k-tiles in PATCH-major order (tile kt = patch chunk kt / npairs, pair kt % npairs): a slice is then a contiguous range of
    // patches with all their pixels, so the workgroups of the 9 taps of a slice -- co-scheduled on one XCD -- re-read the same
    // ~1 MB of planes from its L2.  (Pair-major order read every pixel plane once per tap from HBM: PMC 1.26 GB per launch of
    // layer1's weight gradient against 0.21 GB of tensors, at 6.5 TB/s.)
    const int chunk = kt / npairs, pair = kt - chunk * npairs;
    const int p0 = chunk * BK;
    const int oa = pa[pair], ob = pb[pair];
#pragma unroll
    for (int i = 0; i < C::PA; ++i) {
      const int piece = C::WAVES * i + wave;
      const int row = RPPA * piece + lane / LPRA;
      const int r = r0 + 8 * ((lane % LPRA) ^ rc_sw<TM>(row));
      const int roff = (r >> a.rshift) * a.rstride + (r & ((1 << a.rshift) - 1));  // stem: one kernel row = 32 values of a map row
      const long off = (long)(p0 + row) * a.lda + oa + roff;
      glds16(a.x_hi + off, base + piece * 1024);
      glds16(a.x_lo + off, base + C::IMG_A + piece * 1024);
    }
#pragma unroll
    for (int i = 0; i < C::PB; ++i) {
      const int piece = C::WAVES * i + wave;
      const int row = RPPB * piece + lane / LPRB;
      const int c = c0 + 8 * ((lane % LPRB) ^ rc_sw<TN>(row));
      const long off = (long)(p0 + row) * a.ldb + ob + c;
      glds16(a.d_hi + off, base + 2 * C::IMG_A + piece * 1024);
      glds16(a.d_lo + off, base + 2 * C::IMG_A + C::IMG_B + piece * 1024);
    }
  };

  f32x4 acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);

  if (kt0 < kt1) stage(kt0, 0);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int it = kt - kt0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < kt1) stage(kt + 1, (it + 1) & 1);
    const char *base = lds + (it & 1) * C::STAGE;
#pragma unroll
    for (int s = 0; s < BK / 32; ++s) {
      bf8 b[C::FN], bl[C::FN], af[C::FM], al[C::FM];
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        b[j] = rc_frag<TN>(base + 2 * C::IMG_A, wn + 16 * j, s, lane);
        bl[j] = rc_frag<TN>(base + 2 * C::IMG_A + C::IMG_B, wn + 16 * j, s, lane);
      }
#pragma unroll
      for (int i = 0; i < C::FM; ++i) {
        af[i] = rc_frag<TM>(base, wm + 16 * i, s, lane);
        al[i] = rc_frag<TM>(base + C::IMG_A, wm + 16 * i, s, lane);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // inline-asm reads are invisible to the compiler's bookkeeping
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int j = 0; j < C::FN; ++j) {
          acc[i][j] = mfma(al[i], b[j], acc[i][j]);
          acc[i][j] = mfma(af[i], bl[j], acc[i][j]);
          acc[i][j] = mfma(af[i], b[j], acc[i][j]);
        }
    }
  }

  float *slab = a.slab + ((long)slice * a.ntv + tv) * a.Mtot * a.Ntot;
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + wm + 16 * i + (lane >> 4) * 4 + r, col = c0 + wn + 16 * j + (lane & 15);
        slab[(long)row * a.Ntot + col] = acc[i][j][r];
      }
}

// dw (torch layout) = sum over slices of the slabs; block = 64 outputs x 4 slice lanes (lane l adds slices l, l+4, ... in order,
// then the four lane sums are added pairwise): fixed order.  Taps that never reach the input map get zeros.
//   conv:  dw[co][ci][tap] = sum_s slab[s][tapinv[tap]][ci][co]                (Mtot = cin, Ntot = cout)
//   stem:  dw[co][c][ky][kx] = sum_s slab[s][0][ky * 32 + kx * 4 + c][co]      (Mtot = 256, Ntot = 64, cin = 3, 7x7)
__global__ __launch_bounds__(256) void rn_wgrad_reduce_kernel(RnWgradArgs a, int stem, float *__restrict__ dw) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long mn = (long)a.Mtot * a.Ntot, per_all = (long)a.taps * mn, per = (long)a.ntv * mn;
  const long idx = (long)blockIdx.x * 64 + cl;  // over [taps][Mtot][Ntot]
  const int tap = idx < per_all ? (int)(idx / mn) : 0;
  const int tv = a.tapinv[tap];
  float acc = 0.f;
  if (idx < per_all && tv >= 0) {
    const float *src = a.slab + (long)tv * mn + idx % mn;
#pragma unroll 4
    for (int s = sl; s < a.S; s += 4) acc += src[(long)s * per];
  }
  sh[sl][cl] = acc;
  __syncthreads();
  if (sl != 0 || idx >= per_all) return;
  acc = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
  const int co = (int)(idx % a.Ntot);
  const int r = (int)((idx / a.Ntot) % a.Mtot);
  if (stem) {
    const int ky = r >> 5, kx = (r >> 2) & 7, c = r & 3;
    if (ky < 7 && kx < 7 && c < 3) dw[((co * 3 + c) * 7 + ky) * 7 + kx] = acc;
  } else {
    dw[((long)co * a.Mtot + r) * a.taps + tap] = acc;
  }
}

template <int TM, int TN, int BK_>
int launch_wgrad_cfg(const RnWgradArgs &a, hipStream_t s) {
  using C = TNCfg<TM, TN, BK_>;
  static bool attr_set = false;
  if (!attr_set) {
    // (the deepest ring -- 128 x 128 tiles of 64 patches, the CRW_RN_WBK=64 A/B partner -- leaves 32 KB for the table: 4096 pairs fit
    // the CU's 160 KB exactly once the attribute is clamped to it)
    if (hipFuncSetAttribute((const void *)rn_wgrad_kernel<TM, TN, BK_>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)std::min(C::lds(RN_MAXPAIR_CAP), (size_t)160 * 1024)) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr_set = true;
  }
  if (C::lds(a.maxpair) > (size_t)160 * 1024) return CRW_EINVAL;
  hipLaunchKernelGGL((rn_wgrad_kernel<TM, TN, BK_>), dim3(a.ntv * (a.Mtot / TM) * (a.Ntot / TN) * a.S), dim3(256), C::lds(a.maxpair), s, a);
  return check_launch();
}

}  // namespace

int rn_conv_bk() {
  static int bk = -1;
  if (bk < 0) {
    const char *e = getenv("CRW_RN_BK");  // A/B knob: force the k-tile depth (64: two 64-deep stages; 32: three 32-deep stages)
    bk = e ? atoi(e) : 0;
  }
  return bk;
}

int launch_rn_conv(const RnConvArgs &a, hipStream_t s) {
  if (a.N % 64 || a.mtiles < 1 || a.G < 1 || a.KH * a.KW > RN_MAXSEG) return CRW_EINVAL;
  const bool wide = a.N % 128 == 0;
  // measured in the training step (r03): the k-loops are short (4-18 tiles), so what pays is a second workgroup on the CU that
  // covers the first one's prologue / epilogue: 32-deep k-tiles -- three stages for the 64-column tiles (72 KB of LDS; layer1 and
  // the stem forward 20 % faster than on two 64-deep stages), two stages for the 128-column tiles (64 KB; step 6.9 -> 6.6 ms);
  // the long single segment of the stem's backward-data product prefers two 64-deep stages.  CRW_RN_BK = 64 | 32 | 322 forces one.
  // Default: the split-role kernel (4 MFMA waves + 4 LDS-DMA waves, two 512-thread workgroups per CU, two 32-deep stages): the step
  // measured 1.1-1.4 % faster than on rn_conv_kernel's 4-wave workgroups (4.75 -> 4.69 ms).  Measured SLOWER: deeper rings with one
  // workgroup per CU (CRW_RN_SPEC=3 / 4: 4.77 / 4.82 ms) and 256-patch tiles on 8 waves (CRW_RN_TM=256, rn_conv_kernel<..., TM = 256>:
  // a quarter fewer staged bytes per flop but one workgroup per CU, 4.98 ms; forward 3x3x128 product 87 -> 110 us): what these
  // short-k products need is the second workgroup that covers prologue and epilogue.  CRW_RN_SPEC=0 selects rn_conv_kernel.
  static const int tm = [] { const char *e = getenv("CRW_RN_TM"); return e ? atoi(e) : 128; }();  // 256: 256-patch tiles on 8 waves
  if (tm == 256 && a.mtiles % 2 == 0 && a.mode != RN_MODE_STEM_BWD)
    return wide ? launch_conv_cfg<128, 32, 2, 256>(a, s) : launch_conv_cfg<64, 32, 2, 256>(a, s);
  static const int spec = [] { const char *e = getenv("CRW_RN_SPEC"); return e ? atoi(e) : 2; }();
  if (spec && a.mode != RN_MODE_STEM_BWD) {
    if (spec == 3) return wide ? launch_conv_spec<128, 32, 3>(a, s) : launch_conv_spec<64, 32, 3>(a, s);
    if (spec == 4) return wide ? launch_conv_spec<128, 32, 4>(a, s) : launch_conv_spec<64, 32, 4>(a, s);
    return wide ? launch_conv_spec<128, 32, 2>(a, s) : launch_conv_spec<64, 32, 2>(a, s);
  }
  int bk = rn_conv_bk();
  if (bk != 32 && bk != 64 && bk != 322) bk = a.mode == RN_MODE_STEM_BWD ? 64 : 322;
  if (bk == 322) return wide ? launch_conv_cfg<128, 32, 2>(a, s) : launch_conv_cfg<64, 32, 3>(a, s);
  if (bk == 32) return wide ? launch_conv_cfg<128, 32, 3>(a, s) : launch_conv_cfg<64, 32, 3>(a, s);
  return wide ? launch_conv_cfg<128, 64, 2>(a, s) : launch_conv_cfg<64, 64, 2>(a, s);
}

// dw [64][3][7][7] = sum of `nslab` slabs [224][64] (rn_stem_wgrad_kernel: one per wave pair), fixed order
int launch_rn_stem_slab_reduce(const float *slab, int nslab, float *dw, hipStream_t s) {
  RnWgradArgs a{};
  a.slab = const_cast<float *>(slab);
  a.S = nslab; a.taps = 1; a.ntv = 1; a.Mtot = 224; a.Ntot = 64; a.maxpair = 128;
  a.tapv[0] = 0; a.tapinv[0] = 0;
  hipLaunchKernelGGL(rn_wgrad_reduce_kernel, dim3((224 * 64 + 63) / 64), dim3(256), 0, s, a, 1, dw);
  return check_launch();
}

int rn_wgrad_slices(const RnWgradArgs &a) {
  const int tm = a.Mtot % 128 ? 64 : 128, tn = a.Ntot % 128 ? 64 : 128;
  const int tiles = a.ntv * (a.Mtot / tm) * (a.Ntot / tn);
  int S = (1024 + tiles - 1) / tiles;
  const long kt = (long)a.Hout * a.Wout * a.ktiles_p;  // k-tiles of a tap that sees every output pixel
  if (S > kt / 4) S = (int)(kt / 4);                   // at least four k-tiles per slice
  if (S < 1) S = 1;
  if (S > 256) S = 256;
  return S;
}

int launch_rn_wgrad(const RnWgradArgs &a, float *dw, hipStream_t s) {
  if (a.Mtot % 64 || a.Ntot % 64 || a.Hout * a.Wout > a.maxpair || a.maxpair > RN_MAXPAIR_CAP || a.maxpair % 64 || a.S < 1) return CRW_EINVAL;
  const bool m128 = a.Mtot % 128 == 0, n128 = a.Ntot % 128 == 0;
  static int bk = -1;
  if (bk < 0) {
    // patches per k-tile of the weight gradient: 32 (two 32 KB stages at 128 x 128: two workgroups per CU; step 6.9 -> 6.6 ms
    // against 64-patch tiles with one workgroup per CU); CRW_RN_WBK=64 restores the deeper tiles for A/B runs
    const char *e = getenv("CRW_RN_WBK");
    bk = e ? atoi(e) : 32;
  }
  int st;
  if (bk == 32) {
    if (m128 && n128) st = launch_wgrad_cfg<128, 128, 32>(a, s);
    else if (m128) st = launch_wgrad_cfg<128, 64, 32>(a, s);
    else if (n128) st = launch_wgrad_cfg<64, 128, 32>(a, s);
    else st = launch_wgrad_cfg<64, 64, 32>(a, s);
  } else {
    if (m128 && n128) st = launch_wgrad_cfg<128, 128, 64>(a, s);
    else if (m128) st = launch_wgrad_cfg<128, 64, 64>(a, s);
    else if (n128) st = launch_wgrad_cfg<64, 128, 64>(a, s);
    else st = launch_wgrad_cfg<64, 64, 64>(a, s);
  }
  if (st != CRW_OK) return st;
  const long per = (long)a.taps * a.Mtot * a.Ntot;
  hipLaunchKernelGGL(rn_wgrad_reduce_kernel, dim3((unsigned)((per + 63) / 64)), dim3(256), 0, s, a, a.mode == RN_MODE_STEM_FWD ? 1 : 0, dw);
  return check_launch();
}

}  // namespace crw
