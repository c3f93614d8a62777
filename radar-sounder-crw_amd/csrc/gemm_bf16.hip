// bf16 matrix-core GEMM for the transition-matrix chain (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
//   SPLIT = 1 : operands are bf16 images of the fp32 matrices (probabilities in [0,1]); 2.5 PF roof.
//   SPLIT = 3 : every operand is a (hi, lo) pair of bf16 images, x = hi + lo to ~2^-17 relative;
//               A*B ~= Ah*Bh + Ah*Bl + Al*Bh  (three MFMAs per fragment pair) -> fp32-grade
//               results at a third of the bf16 rate, still ~5x the fp32-MFMA roof.
//
// Geometry: 128x128 output tile per 256-thread workgroup (2x2 waves of 64x64 = 4x4 MFMA tiles),
// BK = 64, zero-padded square operands [n][n] (n multiple of 128) -> no edge handling.
// HBM -> LDS goes through LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): the LDS
// image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and again on
// the read.  Two images, both 16 KiB:
//   KC ("k-contiguous", operand stored [r][k]) : 128 rows x 128 B, chunk' = chunk ^ (row & 7),
//        fragments by ds_read_b128;
//   RC ("r-contiguous", operand stored [k][r]) :  64 rows x 256 B, chunk' = chunk ^ (((row&3)<<2)|((row>>2)&3)),
//        fragments by two ds_read_b64_tr_b16 (hardware transpose), so transposed operands of the
//        chain (Lt^T R, Gt^T dLt, ...) never need a transposed copy in HBM.
// Double-buffered: the DMA of k-tile t+1 is in flight while the MFMAs of tile t run; one barrier
// per k-tile.  Tiles are 128x128 (4 waves) or 256x256 (8 waves, plain bf16 when the grid still fills the chip).
#include "crw_common.h"

namespace crw {
namespace {

// k-tile depth BK: 64, except 32 for 256x256 tiles of hi/lo pairs (halves the 4-image stage to 64 KiB so that two stages
// and 8 waves fit: 1.1 PFLOP/s of MFMA work against 0.84 with 128x128 tiles of 4 waves).  For plain bf16 at 256x256 a
// 4-stage BK = 32 ring measured slower than the 2-stage BK = 64 ring (858 vs 940 TFLOP/s: twice the barriers).
template <int SPLIT, int TB>
constexpr int tile_bk() { return (TB == 256 && SPLIT == 3) ? 32 : 64; }

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char *lds_cp;

// byte offsets inside an image.  KC rows are 2*BK bytes; the chunk swizzle makes the ds_read_b128 lane groups
// ({0-3,12-15,20-27}, ...) of a fragment read conflict-free: BK = 64: chunk ^ (row & 7); BK = 32 (4 chunks per
// row, rows r, r+4, r+8, r+12 share their banks): chunk ^ (row & 8 ? 3 : 0)
template <int BK>
__device__ inline int kc_sw(int row) { return BK == 64 ? (row & 7) : ((row & 8) ? 3 : 0); }
template <int BK>
__device__ inline int kc_off(int row, int chunk) { return row * (2 * BK) + 16 * (chunk ^ kc_sw<BK>(row)); }
__device__ inline int rc_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <int TB>
__device__ inline int rc_off(int row, int chunk) { return row * (TB * 2) + 16 * (chunk ^ rc_sw(row)); }

__device__ inline void glds16(const uint16_t *g, char *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// The transposed LDS read goes through inline asm: given the builtin, hipcc (ROCm 7.2) drains every
// in-flight LDS-DMA (s_waitcnt vmcnt(0)) in front of it, which serialises the ring.  The caller
// issues s_waitcnt lgkmcnt(0) + sched_barrier before the MFMAs that consume the result.
template <int OFF>
__device__ inline s4v tr_read(uint32_t lds_addr) {  // OFF: image offset inside the stage, an instruction immediate, so
  s4v v;                                            // the hi and lo planes of an operand share one address register
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
  return v;
}
__device__ inline uint32_t lds_addr_of(const char *p) { return (uint32_t)(uintptr_t)(lds_cp)p; }

// stage one image of an operand tile with TB rows (or TB columns): TB/8 pieces of 1 KiB
//   KC: [TB rows][BK k]   2*BK-byte rows, 512/BK rows per piece
//   RC: [BK k rows][TB]   2*TB-byte rows, 512/TB rows per piece
template <bool KC, int TB, int WAVES, int BK>
__device__ inline void stage_image(const uint16_t *__restrict__ X, int ld, int r0, int k0, char *img, int wave,
                                   int lane) {
  constexpr int PIECES = TB * BK / 512, PER = PIECES / WAVES;  // 1 KiB pieces of the TB x BK image
  static_assert(PER >= 1 && PIECES % WAVES == 0, "pieces per wave");
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int piece = WAVES * i + wave;
    const uint16_t *src;
    if (KC) {
      constexpr int LPR = BK / 8;  // lanes (16-byte chunks) per row: 8 or 4
      const int row = (64 / LPR) * piece + lane / LPR;
      const int chunk = (lane % LPR) ^ kc_sw<BK>(row);
      src = X + (long)(r0 + row) * ld + k0 + 8 * chunk;
    } else {
      constexpr int LPR = TB / 8;  // lanes (16-byte chunks) per row: 16 or 32
      const int row = (64 / LPR) * piece + lane / LPR;
      const int chunk = (lane % LPR) ^ rc_sw(row);
      src = X + (long)(k0 + row) * ld + r0 + 8 * chunk;
    }
    glds16(src, img + piece * 1024);
  }
}

// The same image through registers: global_load_dwordx4 into PER registers per lane, later ds_write_b128 to the lane-linear
// position the DMA would have written.  The LDS-DMA path of a CU issues ~16 B/clk (MI355X_MICROARCH.md, "LDS-DMA"), the
// vector-memory path to registers 64 B/clk: staging ONE operand this way takes half the bytes off the slower path.
template <bool KC, int TB, int WAVES, int BK>
__device__ inline void load_image_regs(const uint16_t *__restrict__ X, int ld, int r0, int k0, int wave, int lane,
                                       u4v (&r)[TB * BK / 512 / WAVES]) {
  constexpr int PER = TB * BK / 512 / WAVES;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int piece = WAVES * i + wave;
    const uint16_t *src;
    if (KC) {
      constexpr int LPR = BK / 8;
      const int row = (64 / LPR) * piece + lane / LPR;
      const int chunk = (lane % LPR) ^ kc_sw<BK>(row);
      src = X + (long)(r0 + row) * ld + k0 + 8 * chunk;
    } else {
      constexpr int LPR = TB / 8;
      const int row = (64 / LPR) * piece + lane / LPR;
      const int chunk = (lane % LPR) ^ rc_sw(row);
      src = X + (long)(k0 + row) * ld + r0 + 8 * chunk;
    }
    // inline asm: hipcc would wait vmcnt(0) (the younger LDS-DMA included) before the first use of a tracked load; the caller
    // waits with a counted vmcnt instead (vector-memory loads return in order)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(src) : "memory");
  }
}
template <int TB, int WAVES, int BK>
__device__ inline void store_image_regs(char *img, int wave, int lane, const u4v (&r)[TB * BK / 512 / WAVES]) {
  constexpr int PER = TB * BK / 512 / WAVES;
  const uint32_t a = lds_addr_of(img) + wave * 1024 + lane * 16;
#pragma unroll
  for (int i = 0; i < PER; ++i)  // inline asm: a plain LDS store makes hipcc drain the in-flight LDS-DMA first
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(r[i]), "n"(WAVES * 1024 * i) : "memory");
}

// fragment of the 16 rows/cols [rb, rb+16) x k-step s (32 deep) of an image
template <bool KC, int TB, int BK, int OFF>
__device__ inline bf8 read_frag(const char *stage, int rb, int s, int lane) {  // image at stage + OFF
  static_assert(OFF >= 0 && OFF < 65536, "image offset must fit the DS offset field");
  if (KC) {
    const int row = rb + (lane & 15);
    return *reinterpret_cast<const bf8 *>(stage + OFF + kc_off<BK>(row, 4 * s + (lane >> 4)));
  } else {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const int row = 32 * s + 8 * g + q;
    const int chunk = (rb >> 3) + (p >> 1);
    const uint32_t base = lds_addr_of(stage);
    const s4v lo = tr_read<OFF>(base + rc_off<TB>(row, chunk) + 8 * (p & 1));
    const s4v hi = tr_read<OFF>(base + rc_off<TB>(row + 4, chunk) + 8 * (p & 1));
    const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8, v);
  }
}

struct BfOperand {
  const uint16_t *hi, *lo;
};

// Tile configuration: TB x TB output tile, WAVES waves laid out WM x WN, each FM x FN MFMA tiles.
template <int TB_>
struct Cfg;
template <>
struct Cfg<128> { static constexpr int WAVES = 4, WN = 2, FM = 4, FN = 4; };
template <>
struct Cfg<256> { static constexpr int WAVES = 8, WN = 4, FM = 8, FN = 4; };

// Stages of the LDS-DMA ring: two everywhere.  At 128 x 128 / plain bf16 a stage is 32 KiB, which leaves room for TWO
// workgroups per CU; measured at n = 4096 that beats one workgroup with a 4-stage ring (800 vs 605 TFLOP/s).
template <int SPLIT, int TB>
constexpr int ring_stages() {
  return 2;
}
template <int SPLIT, int TB>
constexpr size_t ring_bytes() { return (size_t)ring_stages<SPLIT, TB>() * ((SPLIT == 3) ? 4 : 2) * TB * 2 * tile_bk<SPLIT, TB>(); }

// DIAG (tools/ubench/gemm_ceiling.hip only; the library instantiates DIAG = 0): timing-only variants of this loop that leave parts
// of it out, so that what each part costs can be read off on the chip -- results are meaningless.
//   1 = MFMAs alone (fragments read once, before the loop)      2 = + the LDS fragment reads of every k-step
//   3 = + the barrier (and its waits) per k-tile                 4 = + the LDS-DMA stream: the whole loop (= 0)
//   5 = the LDS-DMA stream alone with its waits (no reads, no MFMAs, no barrier)      6 = DMA stream + barrier
//   8 = NOT a diagnostic: the whole loop with the LDS-DMA requests of half the waves moved behind the first half of the k-tile's
//       MFMAs (CRW_GEMM_STAGGER=1).  The barrier lines all eight waves up, so the two waves of a SIMD issue their 8 requests each
//       (60-180 issue cycles apiece) at the same moment and the matrix pipe waits; staggered, one wave of a SIMD requests while the
//       other multiplies.
//   9 = the same with the second half's requests after the FIRST QUARTER of the k-tile (BK = 64 only; more time to land)
constexpr bool diag_stagger(int d) { return d == 8 || d == 9; }
constexpr bool diag_timing(int d) { return d != 0 && d != 8 && d != 9; }
constexpr bool diag_dma(int d) { return d == 0 || d >= 4; }
constexpr bool diag_barrier(int d) { return d == 0 || d == 3 || d == 4 || d == 6 || d == 8 || d == 9; }
constexpr bool diag_reads(int d) { return d == 0 || (d >= 2 && d <= 4) || d == 8 || d == 9; }
constexpr bool diag_mfma(int d) { return d <= 4 || d == 8 || d == 9; }

template <int SPLIT, int TB, bool AKC, bool BKC, bool REGA, int DIAG = 0>
__device__ inline void mainloop(f32x4 (&acc)[Cfg<TB>::FM][Cfg<TB>::FN], BfOperand A, BfOperand B, int n, int m0,
                                int n0, char *lds, int wave, int lane) {
  using C = Cfg<TB>;
  constexpr int BKB = tile_bk<SPLIT, TB>();
  constexpr int IMG = TB * 2 * BKB;             // bytes per image
  constexpr int NIMG = (SPLIT == 3) ? 4 : 2;    // images per stage: A(hi[,lo]) B(hi[,lo])
  constexpr int NSTAGE = ring_stages<SPLIT, TB>();
  constexpr int G = (TB * BKB / 512 / C::WAVES) * NIMG;  // LDS-DMA instructions per wave per k-tile
  const int wm = (wave / C::WN) * (C::FM * 16), wn = (wave % C::WN) * (C::FN * 16);
  const int nt = n / BKB;
  auto stage = [&](int t, int buf) {
    char *base = lds + buf * NIMG * IMG;
    stage_image<AKC, TB, C::WAVES, BKB>(A.hi, n, m0, t * BKB, base, wave, lane);
    stage_image<BKC, TB, C::WAVES, BKB>(B.hi, n, n0, t * BKB, base + IMG, wave, lane);
    if (SPLIT == 3) {
      stage_image<AKC, TB, C::WAVES, BKB>(A.lo, n, m0, t * BKB, base + 2 * IMG, wave, lane);
      stage_image<BKC, TB, C::WAVES, BKB>(B.lo, n, n0, t * BKB, base + 3 * IMG, wave, lane);
    }
  };
  // REGA (plain bf16, two stages): the A image goes through registers.  Per k-tile: [barrier] A loads of tile t+1 to registers,
  // B DMA of tile t+1, MFMAs of tile t, then vmcnt(G/2) (loads return in order: the A registers have arrived, the B DMA may
  // still be in flight) and the ds_writes of A into the other buffer, which every wave left at the barrier above.
  constexpr int GB = G / NIMG;  // DMA instructions per wave per image
  u4v ra[TB * BKB / 512 / C::WAVES];
  if constexpr (REGA) {
    static_assert(SPLIT == 1 && NSTAGE == 2, "register staging: plain bf16, two stages");
    load_image_regs<AKC, TB, C::WAVES, BKB>(A.hi, n, m0, 0, wave, lane, ra);
    stage_image<BKC, TB, C::WAVES, BKB>(B.hi, n, n0, 0, lds + IMG, wave, lane);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GB) : "memory");
    store_image_regs<TB, C::WAVES, BKB>(lds, wave, lane, ra);
  } else {
    for (int t = 0; t < NSTAGE - 1 && t < nt; ++t) stage(t, t);
  }
  // (DIAG builds without the DMA stream or without fragment reads in the loop: both buffers hold real tiles, fragments of tile 0)
  bf8 dg_b[C::FN], dg_bl[C::FN], dg_a[4], dg_al[4];
  if constexpr (diag_timing(DIAG)) {
    if (nt > 1 && NSTAGE == 2) stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < C::FN; ++j) {
      dg_b[j] = read_frag<BKC, TB, BKB, IMG>(lds, wn + 16 * j, 0, lane);
      dg_bl[j] = dg_b[j];
    }
    for (int i = 0; i < 4; ++i) {
      dg_a[i] = read_frag<AKC, TB, BKB, 0>(lds, wm + 16 * i, 0, lane);
      dg_al[i] = dg_a[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  for (int t = 0; t < nt; ++t) {
    if constexpr (diag_dma(DIAG)) {
      if (t + 2 < nt && NSTAGE >= 4)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
      else if (t + 1 < nt && NSTAGE >= 3)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if constexpr (diag_barrier(DIAG)) __builtin_amdgcn_s_barrier();  // tile t landed for every wave; buffer (t-1) % NSTAGE is free
    if constexpr (REGA) {
      if (t + 1 < nt) {
        load_image_regs<AKC, TB, C::WAVES, BKB>(A.hi, n, m0, (t + 1) * BKB, wave, lane, ra);
        stage_image<BKC, TB, C::WAVES, BKB>(B.hi, n, n0, (t + 1) * BKB, lds + ((t + 1) & 1) * NIMG * IMG + IMG, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (diag_dma(DIAG)) {
      if (t + NSTAGE - 1 < nt && (!diag_stagger(DIAG) || wave < C::WAVES / 2)) stage(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
    }
    const char *base = lds + (t % NSTAGE) * NIMG * IMG;
#pragma unroll
    for (int s = 0; s < BKB / 32; ++s) {
      // B fragments once per k-step; A fragments in groups of 4 row tiles (with 8 row tiles per wave, all 16 hi/lo
      // A fragments at once push the hi/lo kernels past 256 registers)
      constexpr int AG = C::FM > 4 ? 4 : C::FM;
      bf8 b[C::FN], bl[C::FN];
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        if constexpr (diag_reads(DIAG)) {
          b[j] = read_frag<BKC, TB, BKB, IMG>(base, wn + 16 * j, s, lane);
          if constexpr (SPLIT == 3) bl[j] = read_frag<BKC, TB, BKB, 3 * IMG>(base, wn + 16 * j, s, lane);
        } else {
          b[j] = dg_b[j];
          bl[j] = dg_bl[j];
        }
      }
#pragma unroll
      for (int i0 = 0; i0 < C::FM; i0 += AG) {
        bf8 a[AG], al[AG];
#pragma unroll
        for (int i = 0; i < AG; ++i) {
          if constexpr (diag_reads(DIAG)) {
            a[i] = read_frag<AKC, TB, BKB, 0>(base, wm + 16 * (i0 + i), s, lane);
            if constexpr (SPLIT == 3) al[i] = read_frag<AKC, TB, BKB, 2 * IMG>(base, wm + 16 * (i0 + i), s, lane);
          } else {
            a[i] = dg_a[i];
            al[i] = dg_al[i];
          }
        }
        if ((!AKC || !BKC) && diag_reads(DIAG)) {  // inline-asm reads are invisible to the compiler's lgkmcnt bookkeeping
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (diag_mfma(DIAG)) {
#pragma unroll
          for (int i = 0; i < AG; ++i)
#pragma unroll
            for (int j = 0; j < C::FN; ++j) {
              if (SPLIT == 3) {
                acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], b[j], acc[i0 + i][j], 0, 0, 0);
                acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bl[j], acc[i0 + i][j], 0, 0, 0);
              }
              acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i0 + i][j], 0, 0, 0);
            }
          if constexpr (diag_stagger(DIAG)) {
            // the second half of the waves requests tile t + 1 here: after the first k-step (BK = 64) / the first group of row tiles
            // (BK = 32) -- the SIMD's other wave keeps the matrix pipe busy meanwhile
            constexpr bool two_steps = BKB / 32 == 2;
            const bool here = (two_steps && DIAG == 9) ? (s == 0 && i0 == 0) : (two_steps ? (s == 0 && i0 + AG >= C::FM) : (i0 == 0));
            if (here && wave >= C::WAVES / 2 && t + NSTAGE - 1 < nt)
              stage(t + NSTAGE - 1, (t + NSTAGE - 1) % NSTAGE);
          }
        } else if constexpr (diag_reads(DIAG)) {  // (no MFMAs: keep the fragment reads alive)
#pragma unroll
          for (int i = 0; i < AG; ++i) asm volatile("" ::"v"(a[i]), "v"(al[i]));
#pragma unroll
          for (int j = 0; j < C::FN; ++j) asm volatile("" ::"v"(b[j]), "v"(bl[j]));
        }
      }
    }
    if constexpr (REGA) {
      if (t + 1 < nt) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GB) : "memory");
        store_image_regs<TB, C::WAVES, BKB>(lds + ((t + 1) & 1) * NIMG * IMG, wave, lane, ra);
      }
    }
  }
  __syncthreads();  // the second product (or the next use of LDS) may restage buffer 0
}

// ---- 256 x 256, plain bf16: ring of FIVE 32-deep half-tiles, one barrier per PAIR of them ---------------------------------------
// tools/ubench/gemm_ceiling.hip (profiles/r04_gemm_ceiling.log) on the two-stage ring above: MFMAs + fragment reads + barrier take
// 1.46 us per 64-deep k-tile and CU, the LDS-DMA stream ALONE 1.12 us (58 GB/s per CU) -- but the stream WITH the barrier 1.73 us:
// with two 64 KiB stages, tile t+1 can only be requested once every wave has left tile t-1 (the barrier) and must have landed by
// the next barrier, so every k-tile pays the full issue + landing latency of its 64 KiB, and 160 KiB of LDS hold no third stage.
// Here the same 64-deep tile is two half-tiles of 32 KiB in a ring of five: at barrier t the pair (2t, 2t+1) has landed, half 2t+2
// is already in flight (requested one barrier earlier), and the two buffers the barrier frees take halves 2t+3 and 2t+4.  Every
// half gets a whole period (or two) to land, 96 KiB are in flight instead of 64, and the barrier count stays one per 64 deep
// (round 2's 5-stage ring of 32-deep tiles paid a barrier per 32 and lost).  LDS: 5 x 32 KiB = all 160 KiB of the CU.
template <int TB, bool AKC, bool BKC, int DIAG = 0>
__device__ inline void mainloop_ring5(f32x4 (&acc)[Cfg<TB>::FM][Cfg<TB>::FN], BfOperand A, BfOperand B, int n, int m0, int n0, char *lds,
                                      int wave, int lane) {
  using C = Cfg<TB>;
  constexpr int BKB = 32, IMG = TB * 2 * BKB, STG = 2 * IMG, NST = 5;
  constexpr int GH = (TB * BKB / 512 / C::WAVES) * 2;  // LDS-DMA instructions per wave and half-tile
  const int wm = (wave / C::WN) * (C::FM * 16), wn = (wave % C::WN) * (C::FN * 16);
  const int nh = n / BKB;  // half-tiles (even: n is a multiple of 256)
  auto stage = [&](int h) {
    char *base = lds + (h % NST) * STG;
    stage_image<AKC, TB, C::WAVES, BKB>(A.hi, n, m0, h * BKB, base, wave, lane);
    stage_image<BKC, TB, C::WAVES, BKB>(B.hi, n, n0, h * BKB, base + IMG, wave, lane);
  };
  for (int h = 0; h < 3 && h < nh; ++h) stage(h);
  for (int t = 0; 2 * t < nh; ++t) {
    if (2 * t + 2 < nh) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GH) : "memory");  // all but the youngest half (2t+2) have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // ... for every wave; the buffers of halves 2t-2 and 2t-1 are free
    if constexpr (DIAG == 0 || DIAG >= 4) {
      if (2 * t + 3 < nh) stage(2 * t + 3);
      if (2 * t + 4 < nh) stage(2 * t + 4);
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const char *base = lds + ((2 * t + hh) % NST) * STG;
      constexpr int AG = 4;
      bf8 b[C::FN];
#pragma unroll
      for (int j = 0; j < C::FN; ++j) b[j] = read_frag<BKC, TB, BKB, IMG>(base, wn + 16 * j, 0, lane);
#pragma unroll
      for (int i0 = 0; i0 < C::FM; i0 += AG) {
        bf8 a[AG];
#pragma unroll
        for (int i = 0; i < AG; ++i) a[i] = read_frag<AKC, TB, BKB, 0>(base, wm + 16 * (i0 + i), 0, lane);
        if (!AKC || !BKC) {  // inline-asm reads are invisible to the compiler's lgkmcnt bookkeeping
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DIAG <= 4) {
#pragma unroll
          for (int i = 0; i < AG; ++i)
#pragma unroll
            for (int j = 0; j < C::FN; ++j) acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i0 + i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < AG; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
          for (int j = 0; j < C::FN; ++j) asm volatile("" ::"v"(b[j]));
        }
      }
    }
  }
  __syncthreads();  // the second product (or the next use of LDS) may restage the ring
}

__device__ inline uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// One kernel per operand layout (AKC, BKC): a run-time switch over the four main loops costs ~70
// VGPRs and spills the 256x256 configuration.  The launcher groups products by layout.
template <int SPLIT, int TB, bool AKC, bool BKC, bool REGA, int DIAG = 0, bool RING5 = false>
__global__ __launch_bounds__(Cfg<TB>::WAVES * 64) void gemm_pad_bf16_kernel(GemmGroup g) {
  static_assert(!RING5 || (SPLIT == 1 && TB == 256 && !REGA), "the five-half-tile ring: plain bf16 on 256 x 256 tiles");
  using C = Cfg<TB>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const GemmProb &p = g.p[blockIdx.z];
  const int n = g.n;
  const int tiles = n / TB;
  // XCD-aware remap (speed only, bijective): blocks b and b+8 share an XCD and its 4 MiB L2.  The
  // blocks an XCD runs together are made compact 4 x 8 super-tiles, so per k-step they pull
  // 4 A-panels + 8 B-panels through that L2 instead of 1 + 32.
  int tm, tn;
  {
    const int bid = blockIdx.x;
    if ((tiles & 7) == 0) {
      const int nwg = tiles * tiles;
      const int lin = (bid & 7) * (nwg >> 3) + (bid >> 3);  // XCD x owns a contiguous run of lin
      const int st = lin >> 5, local = lin & 31;            // super-tile major order
      const int st_cols = tiles >> 3;
      tm = (st / st_cols) * 4 + (local >> 3);
      tn = (st % st_cols) * 8 + (local & 7);
    } else {
      tm = bid / tiles;
      tn = bid % tiles;
    }
  }
  const int m0 = tm * TB, n0 = tn * TB;
  const long b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  f32x4 acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int prod = 0; prod < 2; ++prod) {
    const uint16_t *Ah = (const uint16_t *)(prod ? p.A2 : p.A);
    if (!Ah) continue;
    const long sA = prod ? p.sA2 : p.sA, sB = prod ? p.sB2 : p.sB;
    BfOperand A{Ah + b * sA, (const uint16_t *)(prod ? p.A2l : p.Al) + b * sA};
    BfOperand B{(const uint16_t *)(prod ? p.B2 : p.B) + b * sB, (const uint16_t *)(prod ? p.B2l : p.Bl) + b * sB};
    if constexpr (RING5) mainloop_ring5<TB, AKC, BKC, DIAG>(acc, A, B, n, m0, n0, lds, wave, lane);
    else mainloop<SPLIT, TB, AKC, BKC, REGA, DIAG>(acc, A, B, n, m0, n0, lds, wave, lane);
  }

  const int wm = (wave / C::WN) * (C::FM * 16), wn = (wave % C::WN) * (C::FN * 16);
  float *Cf = p.C ? p.C + b * p.sC : nullptr;
  uint16_t *Ch = p.Cb ? (uint16_t *)p.Cb + b * p.sC : nullptr;
  uint16_t *Cl = p.Cbl ? (uint16_t *)p.Cbl + b * p.sC : nullptr;
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) {
      const int col = n0 + wn + 16 * j + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + 16 * i + (lane >> 4) * 4 + r;
        const long o = (long)row * n + col;
        float v = acc[i][j][r];
        if (Cf) {
          if (p.beta) v += Cf[o];
          Cf[o] = v;
        }
        if (Ch) {
          const uint16_t h = f2bf(v);
          Ch[o] = h;
          if (Cl) Cl[o] = f2bf(v - bf2f(h));
        }
      }
    }
}

template <int SPLIT, int TB, bool AKC, bool BKC, bool REGA = false, int DIAG = 0, bool RING5 = false>
int launch_one(const GemmGroup &g, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = RING5 ? (size_t)5 * 2 * TB * 2 * 32 : ring_bytes<SPLIT, TB>();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)gemm_pad_bf16_kernel<SPLIT, TB, AKC, BKC, REGA, DIAG, RING5>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr_set = true;
  }
  const int tiles = g.n / TB;
  hipLaunchKernelGGL((gemm_pad_bf16_kernel<SPLIT, TB, AKC, BKC, REGA, DIAG, RING5>), dim3(tiles * tiles, g.batch, g.nprob),
                     dim3(Cfg<TB>::WAVES * 64), lds, s, g);
  return check_launch();
}

// CRW_GEMM_REGA=1: plain-bf16 products stage their A operand through registers (see load_image_regs)
inline bool rega_on() {
  static const bool on = [] { const char *e = getenv("CRW_GEMM_REGA"); return e && e[0] == '1'; }();
  return on;
}

// CRW_GEMM_RING5=1 (opt-in): plain-bf16 256 x 256 products on the ring of five 32-deep half-tiles instead of the two-stage ring of
// 64-deep tiles.  Measured both ways (profiles/r04_gemm_ceiling.log, profiles/r04_gemm_ring5.log): back to back in the stand-alone
// micro-benchmark the five-slot ring is 6 % faster (1151 against 1080 TFLOP/s), through the library from the walk it is 4-9 %
// SLOWER in all four operand layouts and the whole N = 4096 walk takes 48.0 against 46.3 ms -- and the micro-benchmark's
// "LDS-DMA stream + barrier" variant runs at the same 38-44 GB/s per CU with either ring: deeper prefetch does not raise what a
// barrier-synchronised workgroup takes in, so the two-stage ring stays the default.
inline bool ring5_on() {
  static const bool on = [] { const char *e = getenv("CRW_GEMM_RING5"); return e && e[0] == '1'; }();
  return on;
}

// 256 x 256 tiles: the LDS-DMA requests of half the waves (one of each SIMD's pair) are moved behind the first half of the k-tile's
// MFMAs (mainloop, DIAG 8) -- the default since round 4: plain bf16 +4-7 % in all four operand layouts at n = 4096 (988 -> 1031,
// 1003 -> 1054, 946 -> 1018, 968 -> 1032 TFLOP/s), hi/lo pairs +1 %, the whole N = 4096 walk 45.2 -> 43.9 ms
// (profiles/r04_gemm_stagger.log).  CRW_GEMM_STAGGER=0: every wave requests right behind the barrier (rounds 1-3); 2: the second
// half requests after the first QUARTER of the k-tile (equal within noise: profiles/r04_gemm_stagger.log).
inline int stagger_on() {
  static const int on = [] { const char *e = getenv("CRW_GEMM_STAGGER"); return e ? atoi(e) : 1; }();
  return on;
}

template <int SPLIT, int TB>
int launch_layout(const GemmGroup &g, int code, hipStream_t s) {
  if constexpr (TB == 256) {
    if (stagger_on() == 2 && !rega_on()) switch (code) {
        case 3: return launch_one<SPLIT, TB, true, true, false, 9>(g, s);
        case 2: return launch_one<SPLIT, TB, true, false, false, 9>(g, s);
        case 1: return launch_one<SPLIT, TB, false, true, false, 9>(g, s);
        default: return launch_one<SPLIT, TB, false, false, false, 9>(g, s);
      }
    if (stagger_on() == 1 && !rega_on()) switch (code) {
        case 3: return launch_one<SPLIT, TB, true, true, false, 8>(g, s);
        case 2: return launch_one<SPLIT, TB, true, false, false, 8>(g, s);
        case 1: return launch_one<SPLIT, TB, false, true, false, 8>(g, s);
        default: return launch_one<SPLIT, TB, false, false, false, 8>(g, s);
      }
  }
  if constexpr (SPLIT == 1 && TB == 256) {
    if (ring5_on() && !rega_on()) switch (code) {
        case 3: return launch_one<SPLIT, TB, true, true, false, 0, true>(g, s);
        case 2: return launch_one<SPLIT, TB, true, false, false, 0, true>(g, s);
        case 1: return launch_one<SPLIT, TB, false, true, false, 0, true>(g, s);
        default: return launch_one<SPLIT, TB, false, false, false, 0, true>(g, s);
      }
  }
  if constexpr (SPLIT == 1) {
    if (rega_on()) switch (code) {
        case 3: return launch_one<SPLIT, TB, true, true, true>(g, s);
        case 2: return launch_one<SPLIT, TB, true, false, true>(g, s);
        case 1: return launch_one<SPLIT, TB, false, true, true>(g, s);
        default: return launch_one<SPLIT, TB, false, false, true>(g, s);
      }
  }
  switch (code) {
    case 3: return launch_one<SPLIT, TB, true, true>(g, s);
    case 2: return launch_one<SPLIT, TB, true, false>(g, s);
    case 1: return launch_one<SPLIT, TB, false, true>(g, s);
    default: return launch_one<SPLIT, TB, false, false>(g, s);
  }
}

// op(A)(m,k) is k-contiguous unless transposed; op(B)(k,c) is k-contiguous only when transposed
inline int layout_code(int ta, int tb) { return (ta ? 0 : 2) | (tb ? 1 : 0); }

}  // namespace

int launch_gemm_group_bf16(const GemmGroup &g, int split, hipStream_t s) {
  if (g.nprob < 1 || g.nprob > MAX_GROUP || g.n <= 0 || g.n % 128 || g.batch < 1) return CRW_EINVAL;
  // pass 0: first products (plus second products of the same layout); pass 1: second products whose
  // layout differs, accumulated with beta = 1 (needs the fp32 C)
  for (int pass = 0; pass < 2; ++pass)
    for (int code = 0; code < 4; ++code) {
      GemmGroup sub{};
      sub.n = g.n;
      sub.batch = g.batch;
      for (int i = 0; i < g.nprob; ++i) {
        const GemmProb &p = g.p[i];
        const int c1 = layout_code(p.ta, p.tb);
        const bool two = p.A2 != nullptr;
        const int c2 = two ? layout_code(p.ta2, p.tb2) : c1;
        if (pass == 0 && c1 == code) {
          GemmProb q = p;
          if (two && c2 != c1) {  // second product goes to pass 1; keep the bf16 images for the end
            if (!p.C) return CRW_EINVAL;
            q.A2 = q.B2 = q.A2l = q.B2l = nullptr;
            q.Cb = q.Cbl = nullptr;
          }
          sub.p[sub.nprob++] = q;
        } else if (pass == 1 && two && c2 != c1 && c2 == code) {
          GemmProb q = p;
          q.A = p.A2; q.B = p.B2; q.Al = p.A2l; q.Bl = p.B2l;
          q.sA = p.sA2; q.sB = p.sB2; q.ta = p.ta2; q.tb = p.tb2;
          q.A2 = q.B2 = q.A2l = q.B2l = nullptr;
          q.beta = 1;
          sub.p[sub.nprob++] = q;
        }
      }
      if (!sub.nprob) continue;
      // 256 x 256 tiles (half the operand bytes per flop) when they still fill the chip
      const bool big = (g.n % 256 == 0) && ((long)(g.n / 256) * (g.n / 256) * sub.batch * sub.nprob >= 256);
      int st;
      if (split == 3) st = big ? launch_layout<3, 256>(sub, code, s) : launch_layout<3, 128>(sub, code, s);
      else st = big ? launch_layout<1, 256>(sub, code, s) : launch_layout<1, 128>(sub, code, s);
      if (st != CRW_OK) return st;
    }
  return CRW_OK;
}

// fp32 [count] -> bf16 hi (and lo) images
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *__restrict__ x, long count,
                                                         uint16_t *__restrict__ hi, uint16_t *__restrict__ lo) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const float v = x[i];
    const uint16_t h = f2bf(v);
    hi[i] = h;
    if (lo) lo[i] = f2bf(v - bf2f(h));
  }
}

int launch_split_bf16(const float *x, long count, void *hi, void *lo, hipStream_t s) {
  long blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, count, (uint16_t *)hi,
                     (uint16_t *)lo);
  return check_launch();
}

}  // namespace crw

// A, B fp32 [batch][n][n] are converted to bf16 (hi[,lo]) images in `ws` (crw_gemm_bf16_ws_bytes),
// then C = op(A) op(B) (+C) with bf16 MFMA.  split = 1 (plain bf16) or 3 (hi/lo, fp32-grade).
extern "C" size_t crw_gemm_bf16_ws_bytes(int n, int batch, int split) {
  if (n < 1 || batch < 1) return 0;
  return (size_t)n * n * batch * 2 * 2 * (split == 3 ? 2 : 1);
}

extern "C" int crw_gemm_bf16(const float *A, const float *B, float *C, int n, int batch, int transA, int transB,
                             int beta, int split, void *ws, size_t ws_bytes, int convert, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!A || !B || !C || !ws || (split != 1 && split != 3)) return CRW_EINVAL;
  if (n < 128 || n % 128) return CRW_EINVAL;
  if (ws_bytes < crw_gemm_bf16_ws_bytes(n, batch, split)) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const long cnt = (long)n * n * batch;
  uint16_t *Ah = (uint16_t *)ws, *Bh = Ah + cnt, *Al = nullptr, *Bl = nullptr;
  if (split == 3) { Al = Bh + cnt; Bl = Al + cnt; }
  if (convert) {
    CRW_TRY(crw::launch_split_bf16(A, cnt, Ah, Al, s));
    CRW_TRY(crw::launch_split_bf16(B, cnt, Bh, Bl, s));
  }
  crw::GemmGroup g{};
  g.nprob = 1; g.n = n; g.batch = batch;
  crw::GemmProb &p = g.p[0];
  p.A = Ah; p.Al = Al; p.B = Bh; p.Bl = Bl; p.C = C;
  p.sA = p.sB = p.sC = (long)n * n;
  p.ta = transA; p.tb = transB; p.beta = beta;
  return crw::launch_gemm_group_bf16(g, split, s);
}
