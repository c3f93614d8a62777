// fp32 matrix-core GEMMs for the random-walk chain (exact fp32: v_mfma_f32_16x16x4_f32 is a
// k-ordered fmaf chain, so results are reproducible and within fp32 rounding of the reference's
// CPU sgemm).
//
//  * gemm_pad_f32_kernel  -- grouped + batched product of zero-padded square matrices
//                            [n][n], n a multiple of the tile.  No edge handling, 16-byte
//                            global loads, LDS-staged k-major tiles, register prefetch of the
//                            next k-tile.  One launch carries up to MAX_GROUP independent
//                            products (the three products of one walk step / the four of one
//                            backward step), blockIdx.z selects the product.
//  * edge_gemm_kernel     -- bounds-checked, arbitrarily strided product used for the affinity
//                            build (K = C = 128) and its backward.
//
// Fragment maps (gfx950, 16x16x4 f32): A lane l -> A[row l&15][k l>>4], B lane l -> B[k l>>4][col l&15],
// C/D reg r of lane l -> C[row (l>>4)*4 + r][col l&15].
#include "crw_common.h"
#include <cstdlib>

namespace crw {

thread_local int g_last_hip_error = 0;

namespace {

constexpr int BK = 16;

// Stage an R x BK operand tile.  Logical element (r, k) of op(X):
//   kcontig : X[(r0 + r) * n + k0 + k]      (rows of X run along k)
//   !kcontig: X[(k0 + k) * n + r0 + r]      (rows of X run along r)
template <int R>
struct Stager {
  static constexpr int NV4 = R * BK / 4;                 // float4 per tile
  static constexpr int PER = (NV4 + 255) / 256;          // float4 per thread
  float4 v[PER];

  __device__ inline void load(const float *__restrict__ X, int n, int r0, int k0, bool kcontig, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int e = tid + i * 256;
      if (NV4 % 256 == 0 || e < NV4) {
        if (kcontig) {
          int r = e >> 2, kq = e & 3;
          v[i] = *reinterpret_cast<const float4 *>(X + (long)(r0 + r) * n + k0 + 4 * kq);
        } else {
          int k = e / (R / 4), rq = e % (R / 4);
          v[i] = *reinterpret_cast<const float4 *>(X + (long)(k0 + k) * n + r0 + 4 * rq);
        }
      }
    }
  }
  template <int LD>
  __device__ inline void store(float *S, bool kcontig, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int e = tid + i * 256;
      if (NV4 % 256 == 0 || e < NV4) {
        if (kcontig) {
          int r = e >> 2, kq = e & 3;
          S[(4 * kq + 0) * LD + r] = v[i].x;
          S[(4 * kq + 1) * LD + r] = v[i].y;
          S[(4 * kq + 2) * LD + r] = v[i].z;
          S[(4 * kq + 3) * LD + r] = v[i].w;
        } else {
          int k = e / (R / 4), rq = e % (R / 4);
          *reinterpret_cast<float4 *>(S + k * LD + 4 * rq) = v[i];
        }
      }
    }
  }
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_pad_f32_kernel(GemmGroup g) {
  constexpr int LDA = BM + 16, LDB = BN + 16;  // (LD mod 32) == 16: the two k-rows a 32-lane group reads hit disjoint banks
  constexpr int TM = BM / 2 / 16, TN = BN / 2 / 16;
  __shared__ __attribute__((aligned(16))) float lds[BK * LDA + BK * LDB];
  float *As = lds, *Bs = lds + BK * LDA;

  const GemmProb &p = g.p[blockIdx.z];
  const int n = g.n;
  const int tiles_n = n / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const long b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

  const int nk = n / BK;
  const int nprod = p.A2 ? 2 : 1;
  const int nt = nk * nprod;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stager<BM> sa;
  Stager<BN> sb;
  auto issue = [&](int t) {
    const bool second = t >= nk;
    const int k0 = (second ? t - nk : t) * BK;
    const float *A = second ? (const float *)p.A2 + b * p.sA2 : (const float *)p.A + b * p.sA;
    const float *Bp = second ? (const float *)p.B2 + b * p.sB2 : (const float *)p.B + b * p.sB;
    const bool ta = second ? p.ta2 : p.ta, tb = second ? p.tb2 : p.tb;
    sa.load(A, n, m0, k0, !ta, tid);   // op(A)(m,k): k-contiguous unless transposed
    sb.load(Bp, n, n0, k0, tb, tid);   // op(B)(k,c): k-contiguous only when transposed
  };
  auto commit = [&](int t) {
    const bool second = t >= nk;
    const bool ta = second ? p.ta2 : p.ta, tb = second ? p.tb2 : p.tb;
    sa.template store<LDA>(As, !ta, tid);
    sb.template store<LDB>(Bs, tb, tid);
  };

  issue(0);
  for (int t = 0; t < nt; ++t) {
    commit(t);
    __syncthreads();
    if (t + 1 < nt) issue(t + 1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[TM], bv[TN];
      const int kr = kk + (lane >> 4), c = lane & 15;
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA + wm + i * 16 + c];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[kr * LDB + wn + j * 16 + c];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  float *C = p.C + b * p.sC;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
        float v = acc[i][j][r];
        float *dst = C + (long)row * n + col;
        if (p.beta) v += *dst;
        *dst = v;
      }
    }
}

// ------------------------------------------------------------------------------------------
// bounds-checked strided GEMM, 64x64 tile, 4 waves of 32x32.
constexpr int EB = 64, ELD = 65;

__device__ inline void edge_stage(const float *__restrict__ base, long rs, long cs, bool a_side, int R, int K,
                                  int r0, int k0, float *S, int tid) {
  // a_side: logical (row=r, col=k) ; b side: logical (row=k, col=r)
  const long sr = a_side ? rs : cs, sk = a_side ? cs : rs;
  const bool kfast = (sk == 1);
#pragma unroll
  for (int i = 0; i < EB * BK / 256; ++i) {
    int e = tid + i * 256;
    int k, r;
    if (kfast) { k = e % BK; r = e / BK; } else { r = e % EB; k = e / EB; }
    float v = 0.f;
    if (r0 + r < R && k0 + k < K) v = base[(long)(r0 + r) * sr + (long)(k0 + k) * sk];
    S[k * ELD + r] = v;
  }
}

__global__ __launch_bounds__(256) void edge_gemm_kernel(EdgeGemm g) {
  __shared__ float As[BK * ELD], Bs[BK * ELD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * EB, n0 = blockIdx.x * EB;
  const int outer = blockIdx.z / g.batch_inner, inner = blockIdx.z % g.batch_inner;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int prod = 0; prod < 2; ++prod) {
    const EdgeOperand &oa = prod ? g.A2 : g.A;
    const EdgeOperand &ob = prod ? g.B2 : g.B;
    if (oa.p == nullptr) continue;
    if (inner == (prod ? g.skip2_inner : g.skip1_inner)) continue;
    const int K = prod ? g.K2 : g.K;
    const float *A = oa.p + (long)outer * (prod ? g.sA2_outer : g.sA_outer) + (long)inner * oa.sb;
    const float *Bp = ob.p + (long)outer * (prod ? g.sB2_outer : g.sB_outer) + (long)inner * ob.sb;
    for (int k0 = 0; k0 < K; k0 += BK) {
      edge_stage(A, oa.rs, oa.cs, true, g.M, K, m0, k0, As, tid);
      edge_stage(Bp, ob.rs, ob.cs, false, g.N, K, n0, k0, Bs, tid);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < BK; kk += 4) {
        const int kr = kk + (lane >> 4), c = lane & 15;
        float a[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[kr * ELD + wm + i * 16 + c];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = Bs[kr * ELD + wn + j * 16 + c];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  float *C = g.C + (long)outer * g.sC_outer + (long)inner * g.sCb;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r];
          C[(long)row * g.ldc + col] = g.divide ? v / g.scale : v * g.scale;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
// Affinity build and its backward on 128-row fp32-MFMA tiles (exact fp32 like the chain; the bounds-checked
// edge_gemm_kernel above stays the fallback for channel counts that are not a multiple of 4 / node counts that are not).
//
// Bounds-aware operand stager (k-major LDS tile [BK][LD] like gemm_pad_f32_kernel): rows beyond `rmax` are clamped (their
// products land in tile rows / columns that are masked at the store), k beyond `kmax` is zero-filled.  Loads are
// unconditional at clamped addresses, then zeroed by a select (a predicated load makes hipcc wait for it on the spot).
template <int R>
struct BoundStager {
  static constexpr int NV4 = R * BK / 4, PER = (NV4 + 255) / 256;
  float4 v[PER];
  // kcontig : X[(r0 + r) * ld + k0 + k]   (kmax % 4 == 0)         !kcontig: X[(k0 + k) * ld + r0 + r]   (rmax % 4 == 0)
  // AL = false: no alignment or multiple-of-4 assumption (element loads, bounds per element): operands whose leading dimension is
  // not a multiple of 4, e.g. dA at the default 63 nodes
  template <bool AL = true>
  __device__ inline void load(const float *__restrict__ X, long ld, int r0, int k0, bool kcontig, int rmax, int kmax, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = min(tid + i * 256, NV4 - 1);
      if (kcontig) {
        const int r = e >> 2, k = k0 + 4 * (e & 3);
        if constexpr (AL) {
          const float4 t = *reinterpret_cast<const float4 *>(X + (long)min(r0 + r, rmax - 1) * ld + min(k, kmax - 4));
          v[i] = k < kmax ? t : float4{0.f, 0.f, 0.f, 0.f};
        } else {
          const float *row = X + (long)min(r0 + r, rmax - 1) * ld;
          float t[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float x = row[min(k + u, kmax - 1)];  // unconditional, clamped
            t[u] = k + u < kmax ? x : 0.f;
          }
          v[i] = float4{t[0], t[1], t[2], t[3]};
        }
      } else {
        const int k = k0 + e / (R / 4), r = r0 + 4 * (e % (R / 4));
        if constexpr (AL) {
          const float4 t = *reinterpret_cast<const float4 *>(X + (long)min(k, kmax - 1) * ld + min(r, rmax - 4));
          v[i] = (k < kmax && r < rmax) ? t : float4{0.f, 0.f, 0.f, 0.f};
        } else {
          const float *row = X + (long)min(k, kmax - 1) * ld;
          float t[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float x = row[min(r + u, rmax - 1)];
            t[u] = (k < kmax && r + u < rmax) ? x : 0.f;
          }
          v[i] = float4{t[0], t[1], t[2], t[3]};
        }
      }
    }
  }
  template <int LD>
  __device__ inline void store(float *S, bool kcontig, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + i * 256;
      if (NV4 % 256 == 0 || e < NV4) {
        if (kcontig) {
          const int r = e >> 2, kq = e & 3;
          S[(4 * kq + 0) * LD + r] = v[i].x;
          S[(4 * kq + 1) * LD + r] = v[i].y;
          S[(4 * kq + 2) * LD + r] = v[i].z;
          S[(4 * kq + 3) * LD + r] = v[i].w;
        } else {
          const int k = e / (R / 4), rq = e % (R / 4);
          *reinterpret_cast<float4 *>(S + k * LD + 4 * rq) = v[i];
        }
      }
    }
  }
};

// one 128 x 128 x BK MFMA block on k-major LDS tiles; wave (wm, wn) owns TM x TN 16 x 16 tiles
template <int TM, int TN, int LDA, int LDB>
__device__ inline void mfma_block(const float *As, const float *Bs, int wm, int wn, int lane, f32x4 (&acc)[TM][TN]) {
#pragma unroll
  for (int kk = 0; kk < BK; kk += 4) {
    float a[TM], bv[TN];
    const int kr = kk + (lane >> 4), c = lane & 15;
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA + wm + i * 16 + c];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = Bs[kr * LDB + wn + j * 16 + c];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
  }
}

// ---- "bf16 x 6": fp32 products on the bf16 matrix cores -------------------------------------------------------------
// An fp32 value is split into three bf16 terms, x = h + m + l exactly (8 + 8 + 8 significand bits), and a product a * b
// is accumulated as al*bh + ah*bl + am*bm + am*bh + ah*bm + ah*bh in fp32: the dropped terms (ml, lm, ll) are below
// 2^-24 |a b|, i.e. the result is fp32-grade like the 16x16x4 fp32 MFMA, at 6/16 of its matrix-pipe time (the fp32 MFMA
// runs at 1/16 of the bf16 rate).  Used by the affinity build and its backward at large node counts, where those two
// fp32-MFMA kernels were 10 % of the plain-bf16 chain step at N = 4096.
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
__device__ inline uint16_t bf_of(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ inline float f_of(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ inline void split3(float x, uint16_t &h, uint16_t &m, uint16_t &l) {
#pragma clang fp contract(off)
  h = bf_of(x);
  const float r1 = x - f_of(h);
  m = bf_of(r1);
  l = bf_of(r1 - f_of(m));
}
// bf16 operand plane in LDS: [R rows][32 k], 64-byte rows, 16-byte chunks swizzled so that the ds_read_b128 lane groups of
// a fragment are conflict-free (same scheme as the 32-deep k-tiles of gemm_bf16.hip)
constexpr int X6K = 32;
__device__ inline int x6_off(int row, int chunk) { return row * 64 + 16 * (chunk ^ ((row & 8) ? 3 : 0)); }
// four consecutive k (k0 = 4 kq) of one row -> the three planes (plane stride PL bytes)
template <int PL>
__device__ inline void x6_put4(char *planes, int row, int kq, float4 v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint16_t h[4], m[4], l[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) split3(x[u], h[u], m[u], l[u]);
  char *dst = planes + x6_off(row, kq >> 1) + 8 * (kq & 1);
  *reinterpret_cast<uint2 *>(dst) = uint2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
  *reinterpret_cast<uint2 *>(dst + PL) = uint2{(uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16)};
  *reinterpret_cast<uint2 *>(dst + 2 * PL) = uint2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
}
// one 32-deep k-step of a wave's TM x TN block of 16 x 16 tiles from the planes of A (rows wm..) and B (rows wn..)
template <int TM, int TN, int PLA, int PLB>
__device__ inline void x6_block(const char *Ap, const char *Bp, int wm, int wn, int lane, f32x4 (&acc)[TM][TN]) {
  const int r16 = lane & 15, g = lane >> 4;
  bf8v ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const char *p = Ap + x6_off(wm + 16 * i + r16, g);
    ah[i] = *reinterpret_cast<const bf8v *>(p);
    am[i] = *reinterpret_cast<const bf8v *>(p + PLA);
    al[i] = *reinterpret_cast<const bf8v *>(p + 2 * PLA);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const char *p = Bp + x6_off(wn + 16 * j + r16, g);
    bh[j] = *reinterpret_cast<const bf8v *>(p);
    bm[j] = *reinterpret_cast<const bf8v *>(p + PLB);
    bl[j] = *reinterpret_cast<const bf8v *>(p + 2 * PLB);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x4 c = acc[i][j];
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm[j], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh[j], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm[j], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], c, 0, 0, 0);
      acc[i][j] = c;
    }
}
// R rows x 32 k of a k-contiguous fp32 operand X[(r0 + r) * ld + k0 + k]: R * 8 float4, loaded unconditionally at clamped
// addresses (rows >= rmax repeat the last row: their products are masked at the store; k >= kmax reads as zero)
template <int R>
struct X6StagerK {
  static constexpr int PER = R * 8 / 256;
  float4 v[PER];
  __device__ inline void load(const float *__restrict__ X, long ld, int r0, int k0, int rmax, int kmax, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + i * 256, r = e >> 3, k = k0 + 4 * (e & 7);
      const float4 t = *reinterpret_cast<const float4 *>(X + (long)min(r0 + r, rmax - 1) * ld + min(k, kmax - 4));
      v[i] = k < kmax ? t : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  template <int PL>
  __device__ inline void store(char *planes, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + i * 256;
      x6_put4<PL>(planes, e >> 3, e & 7, v[i]);
    }
  }
};
// R rows x 32 k of an r-contiguous operand X[(k0 + k) * ld + r0 + r] (R = 128): one 4 k x 4 r block per thread, transposed in
// registers so that every row receives 4 consecutive k (rmax % 4 == 0; k >= kmax and r >= rmax read as zero)
struct X6StagerR {
  float4 v[4];
  __device__ inline void load(const float *__restrict__ X, long ld, int r0, int k0, int rmax, int kmax, int tid) {
    const int rq = tid & 31, kq = tid >> 5, r = r0 + 4 * rq;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 4 * kq + u;
      const float4 t = *reinterpret_cast<const float4 *>(X + (long)min(k, kmax - 1) * ld + min(r, rmax - 4));
      v[u] = (k < kmax && r < rmax) ? t : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  template <int PL>
  __device__ inline void store(char *planes, int tid) const {
    const int rq = tid & 31, kq = tid >> 5;
    x6_put4<PL>(planes, 4 * rq + 0, kq, float4{v[0].x, v[1].x, v[2].x, v[3].x});
    x6_put4<PL>(planes, 4 * rq + 1, kq, float4{v[0].y, v[1].y, v[2].y, v[3].y});
    x6_put4<PL>(planes, 4 * rq + 2, kq, float4{v[0].z, v[1].z, v[2].z, v[3].z});
    x6_put4<PL>(planes, 4 * rq + 3, kq, float4{v[0].w, v[1].w, v[2].w, v[3].w});
  }
};

// all-lanes max / sum over the 16 lanes of a DPP row (lanes with equal lane >> 4): four rotate-and-combine steps on the VALU
// (a __shfl_xor butterfly is four LDS round trips per value)
template <int CTRL>
__device__ inline float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ inline float row16_max(float x) {
  x = fmaxf(x, dpp_f<0x128>(x));  // row_ror:8
  x = fmaxf(x, dpp_f<0x124>(x));  // row_ror:4
  x = fmaxf(x, dpp_f<0x122>(x));  // row_ror:2
  return fmaxf(x, dpp_f<0x121>(x));  // row_ror:1
}
__device__ inline float row16_sum(float x) {
  x += dpp_f<0x128>(x);
  x += dpp_f<0x124>(x);
  x += dpp_f<0x122>(x);
  return x + dpp_f<0x121>(x);
}

constexpr int AT = 128, ATLD = AT + 4;  // affinity tile and the row stride of its LDS image (16-byte aligned rows; the accumulator writes are conflict-free)

// A[b,t] tile = ehat[b,t][m0..] ehat[b,t+1][n0..]^T / tau, plus (optional) the tile's partial softmax statistics:
// per row the (max, sum exp) over the tile's valid columns -> rpart[mat][tn][row], per column over its rows -> cpart[mat][tm][col]
// TS = rows and columns of the tile: 128, or 64 for matrices that fit one 64 x 64 tile (N <= 64: a quarter of the MFMA and
// statistics work of a 128-tile that is three quarters empty -- the kernel is pure latency there, one workgroup per CU)
template <bool X6, int TS = 128>  // X6: products on the bf16 matrix cores in three-term splits (C % 32 == 0), else fp32 MFMA
__global__ __launch_bounds__(256) void affinity_tile_kernel(const float *__restrict__ ehat, int T, int N, int C, float tau,
                                                            float *__restrict__ A, float *__restrict__ part, int tiles,
                                                            float *__restrict__ direct) {
  constexpr int LD = TS + 16, TW = TS / 32, HW = TS / 2, TSLD = TS + 4;  // tiles per wave and dimension, wave quadrant, image stride
  static_assert(!X6 || TS == 128, "the bf16 form is built for 128-tiles");
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  float *As = lds_f, *Bs = lds_f + BK * LD, *tile = lds_f;  // the output tile reuses the operand space
  const int tm = blockIdx.x / tiles, tn = blockIdx.x % tiles, m0 = tm * TS, n0 = tn * TS;
  const long amat = blockIdx.y;                    // caller order: b * (T - 1) + t
  const long b = amat / (T - 1), t = amat % (T - 1);
  const float *E0 = ehat + (b * T + t) * (long)N * C, *E1 = E0 + (long)N * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * HW, wn = (wave & 1) * HW;
  f32x4 acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (X6) {
    constexpr int PL = AT * 64;  // one bf16 plane of a 128-row operand, 32 deep
    char *Ap = reinterpret_cast<char *>(lds_f), *Bp = Ap + 3 * PL;
    // two register sets: the loads of chunk kt + 2 are issued while chunk kt is multiplied, so a load has two MFMA
    // blocks to land (one block is shorter than a load's round trip; with one set every chunk waited for its data)
    X6StagerK<AT> sa0, sb0, sa1, sb1;
    const int nk = C / X6K;
    sa0.load(E0, C, m0, 0, N, C, tid);
    sb0.load(E1, C, n0, 0, N, C, tid);
    sa1.load(E0, C, m0, min(1, nk - 1) * X6K, N, C, tid);
    sb1.load(E1, C, n0, min(1, nk - 1) * X6K, N, C, tid);
    for (int kt = 0; kt < nk; kt += 2) {
      sa0.store<PL>(Ap, tid);
      sb0.store<PL>(Bp, tid);
      __syncthreads();
      sa0.load(E0, C, m0, min(kt + 2, nk - 1) * X6K, N, C, tid);  // unconditional (clamped): unused past the end
      sb0.load(E1, C, n0, min(kt + 2, nk - 1) * X6K, N, C, tid);
      x6_block<TW, TW, PL, PL>(Ap, Bp, wm, wn, lane, acc);
      __syncthreads();
      if (kt + 1 < nk) {
        sa1.store<PL>(Ap, tid);
        sb1.store<PL>(Bp, tid);
        __syncthreads();
        sa1.load(E0, C, m0, min(kt + 3, nk - 1) * X6K, N, C, tid);
        sb1.load(E1, C, n0, min(kt + 3, nk - 1) * X6K, N, C, tid);
        x6_block<TW, TW, PL, PL>(Ap, Bp, wm, wn, lane, acc);
        __syncthreads();
      }
    }
  } else {
  BoundStager<TS> sa, sb;
  const int nk = (C + BK - 1) / BK;
  sa.load(E0, C, m0, 0, true, N, C, tid);
  sb.load(E1, C, n0, 0, true, N, C, tid);
  for (int kt = 0; kt < nk; ++kt) {
    sa.template store<LD>(As, true, tid);
    sb.template store<LD>(Bs, true, tid);
    __syncthreads();
    if (kt + 1 < nk) {
      sa.load(E0, C, m0, (kt + 1) * BK, true, N, C, tid);
      sb.load(E1, C, n0, (kt + 1) * BK, true, N, C, tid);
    }
    mfma_block<TW, TW, LD, LD>(As, Bs, wm, wn, lane, acc);
    __syncthreads();
  }
  }
  // ---- epilogue: the tile goes to global memory through an LDS image (row-contiguous 16-byte stores), its partial softmax
  // statistics come straight from the accumulators: per wave a row's (max, sum exp) over the wave's 64 columns is an
  // in-register reduction over j plus a 16-lane DPP rotate-reduce, a column's over the wave's 64 rows one over (i, r) plus the 4 lane
  // groups; the two waves that share a row (column) range meet in LDS.  (Walking the LDS image row by row and column by column
  // with one thread each -- the first version -- cost four times the MFMA time of the tile.)
  float v[TW][TW][4];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[i][j][r] = acc[i][j][r] / tau;
  const int g = lane >> 4, r16 = lane & 15;
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(wm + 16 * i + 4 * g + r) * TSLD + wn + 16 * j + r16] = v[i][j][r];
  float *rpart = tile + TS * TSLD, *cpart = rpart + 2 * TS * 2;  // [2 column halves][128 rows][m, s], [2 row halves][128 cols][m, s]
  if (part || direct) {
    bool cok[TW], rok[TW][4];
#pragma unroll
    for (int j = 0; j < TW; ++j) cok[j] = n0 + wn + 16 * j + r16 < N;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) rok[i][r] = m0 + wm + 16 * i + 4 * g + r < N;
    // rows: over this wave's columns
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < TW; ++j) m = cok[j] ? fmaxf(m, v[i][j][r]) : m;
        m = row16_max(m);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TW; ++j) sum += cok[j] ? expf(v[i][j][r] - m) : 0.f;
        sum = row16_sum(sum);
        if (r16 == 0) {
          float *dst = rpart + ((wave & 1) * TS + wm + 16 * i + 4 * g + r) * 2;
          dst[0] = m;
          dst[1] = sum;
        }
      }
    // columns: over this wave's rows
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = rok[i][r] ? fmaxf(m, v[i][j][r]) : m;
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += rok[i][r] ? expf(v[i][j][r] - m) : 0.f;
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      if (g == 0) {
        float *dst = cpart + ((wave >> 1) * TS + wn + 16 * j + r16) * 2;
        dst[0] = m;
        dst[1] = sum;
      }
    }
  }
  __syncthreads();
  float *Ab = A + amat * (long)N * N;
  if ((N & 3) == 0) {  // a wave writes 512 contiguous bytes of two rows per instruction
    for (int e = tid; e < TS * TS / 4; e += 256) {
      const int r = e / (TS / 4), c = 4 * (e % (TS / 4));
      if (m0 + r < N && n0 + c < N)
        *reinterpret_cast<float4 *>(Ab + (long)(m0 + r) * N + n0 + c) = *reinterpret_cast<const float4 *>(tile + r * TSLD + c);
    }
  } else {
    for (int e = tid; e < TS * TS; e += 256) {
      const int r = e / TS, c = e % TS;
      if (m0 + r < N && n0 + c < N) Ab[(long)(m0 + r) * N + n0 + c] = tile[r * TSLD + c];
    }
  }
  if (!part && !direct) return;
  // threads 0..127: one tile row each; threads 128..255: one tile column each: merge the two half-tile partials
  if (tid >= 2 * TS) return;
  const int idx = tid % TS;
  const bool rows = tid < TS;
  const float *pp = (rows ? rpart : cpart) + idx * 2;
  const float ma = pp[0], sa = pp[1], mb = pp[TS * 2], sb = pp[TS * 2 + 1];
  const float m = fmaxf(ma, mb);
  const float sum = (sa > 0.f ? sa * expf(ma - m) : 0.f) + (sb > 0.f ? sb * expf(mb - m) : 0.f);
  const int gi = (rows ? m0 : n0) + idx;
  if (gi < N) {
    const long nmat = gridDim.y;
    if (direct) {  // one tile per matrix (N <= 128): the partial IS the statistic -> dense [4][nmat][N], no merge launch
      direct[((rows ? 0 : 2) * nmat + amat) * N + gi] = m;
      direct[((rows ? 1 : 3) * nmat + amat) * N + gi] = sum;
    } else {       // part: [2 (row / column)][nmat][tiles][N][2]
      float *dst = part + ((((rows ? 0 : nmat) + amat) * tiles + (rows ? tn : tm)) * (long)N + gi) * 2;
      dst[0] = m;
      dst[1] = sum;
    }
  }
}

// merge the per-tile partials in tile order (deterministic): stats dense [4][nmat][N]
__global__ __launch_bounds__(256) void affinity_stats_merge_kernel(const float *__restrict__ part, int nmat, int tiles, int N,
                                                                   float *__restrict__ stats) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over 2 * nmat * N
  if (i >= 2L * nmat * N) return;
  const long which = i / ((long)nmat * N), rem = i % ((long)nmat * N), mat = rem / N, n = rem % N;
  const float *p = part + ((which * nmat + mat) * tiles * (long)N + n) * 2;
  float m = -INFINITY;
  for (int t = 0; t < tiles; ++t) m = fmaxf(m, p[(long)t * N * 2]);
  float s = 0.f;
  for (int t = 0; t < tiles; ++t) s += p[(long)t * N * 2 + 1] * expf(p[(long)t * N * 2] - m);
  stats[((2 * which) * nmat + mat) * N + n] = m;
  stats[((2 * which + 1) * nmat + mat) * N + n] = s;
}

// dehat[b,t] tile (RM rows x C) = (dA[b,t] ehat[b,t+1] + dA[b,t-1]^T ehat[b,t-1]) / tau;  C = 32 * TN; RM = 128, or 64 for matrices of
// at most 64 nodes (half the MFMA work of a half-empty 128-row tile: the launch is pure latency there); AL = false: N % 4 != 0
template <int TN, int RM = 128, bool AL = true>  // TN: 16-wide column tiles per wave
__global__ __launch_bounds__(256) void affinity_bwd_tile_kernel(const float *__restrict__ dA, const float *__restrict__ ehat,
                                                                int T, int N, float tau, float *__restrict__ dehat) {
  constexpr int C = 32 * TN, LDA = RM + 16, LDB = C + 16, TM = RM / 32;
  __shared__ __attribute__((aligned(16))) float lds_f[BK * LDA + BK * LDB];
  float *As = lds_f, *Bs = lds_f + BK * LDA;
  const int m0 = blockIdx.x * RM;
  const long bt = blockIdx.y, b = bt / T, t = bt % T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * (RM / 2), wn = (wave & 1) * (C / 2);
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  BoundStager<RM> sa;
  BoundStager<C> sb;
  const long NN = (long)N * N, NC = (long)N * C;
  const int nk = (N + BK - 1) / BK;
  for (int prod = 0; prod < 2; ++prod) {
    if (prod == 0 ? t == T - 1 : t == 0) continue;  // block-uniform
    // product 0: dA[b,t] (row n, k = m), k-contiguous; product 1: dA[b,t-1]^T (row n, k = m) = dA[b,t-1][m][n], r-contiguous
    const float *X = dA + (b * (T - 1) + (prod == 0 ? t : t - 1)) * NN;
    const float *E = ehat + (b * T + (prod == 0 ? t + 1 : t - 1)) * NC;
    const bool akc = prod == 0;
    sa.template load<AL>(X, N, m0, 0, akc, N, N, tid);
    sb.load(E, C, 0, 0, false, C, N, tid);
    for (int kt = 0; kt < nk; ++kt) {
      sa.template store<LDA>(As, akc, tid);
      sb.template store<LDB>(Bs, false, tid);
      __syncthreads();
      if (kt + 1 < nk) {
        sa.template load<AL>(X, N, m0, (kt + 1) * BK, akc, N, N, tid);
        sb.load(E, C, 0, (kt + 1) * BK, false, C, N, tid);
      }
      mfma_block<TM, TN, LDA, LDB>(As, Bs, wm, wn, lane, acc);
      __syncthreads();
    }
  }
  float *D = dehat + bt * NC;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
        if (row < N) D[(long)row * C + wn + 16 * j + (lane & 15)] = acc[i][j][r] / tau;
      }
}

// The same on the bf16 matrix cores (three-term splits of dA and ehat): C = 128 only (one 128 x 128 output tile per workgroup)
__global__ __launch_bounds__(256) void affinity_bwd_x6_kernel(const float *__restrict__ dA, const float *__restrict__ ehat,
                                                              int T, int N, float tau, float *__restrict__ dehat) {
  constexpr int C = 128, PL = AT * 64;
  __shared__ __attribute__((aligned(16))) char lds_c[6 * PL];
  char *Ap = lds_c, *Bp = lds_c + 3 * PL;
  const int m0 = blockIdx.x * AT;
  const long bt = blockIdx.y, b = bt / T, t = bt % T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  X6StagerK<AT> sak;
  X6StagerR sar, sb;
  const long NN = (long)N * N, NC = (long)N * C;
  const int nk = (N + X6K - 1) / X6K;
  for (int prod = 0; prod < 2; ++prod) {
    if (prod == 0 ? t == T - 1 : t == 0) continue;  // block-uniform
    // product 0: dA[b,t] (row n, k = m), k-contiguous; product 1: dA[b,t-1]^T = dA[b,t-1][m][n], r-contiguous
    const float *X = dA + (b * (T - 1) + (prod == 0 ? t : t - 1)) * NN;
    const float *E = ehat + (b * T + (prod == 0 ? t + 1 : t - 1)) * NC;  // [k = node][c]: r-contiguous, 128 "rows" c
    if (prod == 0) sak.load(X, N, m0, 0, N, N, tid);
    else sar.load(X, N, m0, 0, N, N, tid);
    sb.load(E, C, 0, 0, C, N, tid);
    for (int kt = 0; kt < nk; ++kt) {
      if (prod == 0) sak.store<PL>(Ap, tid);
      else sar.store<PL>(Ap, tid);
      sb.store<PL>(Bp, tid);
      __syncthreads();
      if (kt + 1 < nk) {
        if (prod == 0) sak.load(X, N, m0, (kt + 1) * X6K, N, N, tid);
        else sar.load(X, N, m0, (kt + 1) * X6K, N, N, tid);
        sb.load(E, C, 0, (kt + 1) * X6K, C, N, tid);
      }
      x6_block<4, 4, PL, PL>(Ap, Bp, wm, wn, lane, acc);
      __syncthreads();
    }
  }
  float *D = dehat + bt * NC;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
        if (row < N) D[(long)row * C + wn + 16 * j + (lane & 15)] = acc[i][j][r] / tau;
      }
}

// largest tile (128 / 64 / 32) that divides n and still yields a well-filled grid
int pick_tile(int n, long batch_times_prob) {
  const int cand[3] = {128, 64, 32};
  for (int i = 0; i < 3; ++i) {
    int t = cand[i];
    if (n % t) continue;
    long blocks = (long)(n / t) * (n / t) * batch_times_prob;
    if (blocks >= 512 || t == 32) return t;
  }
  return 32;
}

}  // namespace

int launch_gemm_group_f32(const GemmGroup &g, hipStream_t s) {
  if (g.nprob < 1 || g.nprob > MAX_GROUP || g.n <= 0 || g.n % 32 || g.batch < 1) return CRW_EINVAL;
  const int tile = pick_tile(g.n, (long)g.batch * g.nprob);
  dim3 grid((g.n / tile) * (g.n / tile), g.batch, g.nprob);
  if (tile == 128)
    hipLaunchKernelGGL((gemm_pad_f32_kernel<128, 128>), grid, dim3(256), 0, s, g);
  else if (tile == 64)
    hipLaunchKernelGGL((gemm_pad_f32_kernel<64, 64>), grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_pad_f32_kernel<32, 32>), grid, dim3(256), 0, s, g);
  return check_launch();
}

int launch_edge_gemm(const EdgeGemm &g, int batch, hipStream_t s) {
  if (g.M < 1 || g.N < 1 || batch < 1) return CRW_EINVAL;
  dim3 grid((g.N + EB - 1) / EB, (g.M + EB - 1) / EB, batch);
  hipLaunchKernelGGL(edge_gemm_kernel, grid, dim3(256), 0, s, g);
  return check_launch();
}

// ARITHMETIC OF THE AFFINITY BUILD (read this before calling the default "exact fp32"): from 256 nodes on (C % 32 == 0) the
// affinity products and their backward run on the bf16 matrix cores with every fp32 operand split EXACTLY into three bf16 terms
// and six of the nine cross products accumulated in fp32 ("bf16 x 6"); the three dropped terms are each < 2^-24 of the product,
// i.e. the result is fp32-grade, not bit-identical to an fp32 fma chain: measured against fp64 at N = 300 / 512, tau = 0.01
// (logits up to +-100) both paths sit within 5e-5 absolute (tests/test_hip_parity.py::test_affinity_both_arithmetics_vs_fp64).
// Smaller node counts keep the fp32 MFMA (launch-bound there).  CRW_AFFINITY_F32=1 in the environment forces the fp32 MFMA
// everywhere; it is read on every call, so a caller (or a test) can switch per call.
bool affinity_on_bf16(int N, int C) {
  const char *e = getenv("CRW_AFFINITY_F32");
  const bool force_f32 = e && e[0] && e[0] != '0';
  return !force_f32 && N >= 256 && C % 32 == 0;
}

size_t affinity_part_floats(int B, int T, int N) {
  const size_t tiles = (N + AT - 1) / AT;
  return (size_t)2 * B * (T - 1) * tiles * N * 2;
}

int launch_affinity_tiles(const float *ehat, int B, int T, int N, int C, float tau, float *A, float *stats, float *part,
                          hipStream_t s) {
  if (C % 4 || C < 4 || (stats && !part)) return CRW_EINVAL;
  const int tiles = (N + AT - 1) / AT, nmat = B * (T - 1);
  const size_t lds = sizeof(float) * (size_t)(AT * ATLD + 2 * 2 * AT * 2);  // tile image + the half-tile partials (>= the operand planes)
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void *)affinity_tile_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)affinity_tile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  const bool single = tiles == 1;
  if (N <= 64) {  // one 64 x 64 tile per matrix
    const size_t lds64 = sizeof(float) * (size_t)(64 * 68 + 2 * 2 * 64 * 2);  // tile image + half-tile partials (> the operand tiles)
    hipLaunchKernelGGL((affinity_tile_kernel<false, 64>), dim3(1, nmat), dim3(256), lds64, s, ehat, T, N, C, tau, A, nullptr, 1,
                       stats);
    return check_launch();
  }
  if (affinity_on_bf16(N, C))
    hipLaunchKernelGGL(affinity_tile_kernel<true>, dim3(tiles * tiles, nmat), dim3(256), lds, s, ehat, T, N, C, tau, A,
                       (stats && !single) ? part : nullptr, tiles, (stats && single) ? stats : nullptr);
  else
    hipLaunchKernelGGL(affinity_tile_kernel<false>, dim3(tiles * tiles, nmat), dim3(256), lds, s, ehat, T, N, C, tau, A,
                       (stats && !single) ? part : nullptr, tiles, (stats && single) ? stats : nullptr);
  if (stats && !single) {
    const long n = 2L * nmat * N;
    hipLaunchKernelGGL(affinity_stats_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, nmat, tiles, N, stats);
  }
  return check_launch();
}

int launch_affinity_bwd_tiles(const float *dA, const float *ehat, int B, int T, int N, int C, float tau, float *dehat,
                              hipStream_t s) {
  if (N < 1) return CRW_EINVAL;
  const dim3 grid((N + AT - 1) / AT, B * T);
  if (N <= 64 || (N & 3)) {  // one 64-row tile per matrix / node counts that are not a multiple of 4 (element loads of dA)
#define CRW_AFB(TNV)                                                                                                            \
  if (N <= 64 && !(N & 3)) hipLaunchKernelGGL((affinity_bwd_tile_kernel<TNV, 64, true>), dim3(1, B * T), dim3(256), 0, s, dA, ehat, T, N, tau, dehat); \
  else if (N <= 64) hipLaunchKernelGGL((affinity_bwd_tile_kernel<TNV, 64, false>), dim3(1, B * T), dim3(256), 0, s, dA, ehat, T, N, tau, dehat);       \
  else hipLaunchKernelGGL((affinity_bwd_tile_kernel<TNV, 128, false>), grid, dim3(256), 0, s, dA, ehat, T, N, tau, dehat);
    switch (C) {
      case 32: CRW_AFB(1) break;
      case 64: CRW_AFB(2) break;
      case 128: CRW_AFB(4) break;
      default: return CRW_EINVAL;
    }
#undef CRW_AFB
    return check_launch();
  }
  if (C == 128 && affinity_on_bf16(N, C)) {
    hipLaunchKernelGGL(affinity_bwd_x6_kernel, grid, dim3(256), 0, s, dA, ehat, T, N, tau, dehat);
    return check_launch();
  }
  switch (C) {
    case 32: hipLaunchKernelGGL(affinity_bwd_tile_kernel<1>, grid, dim3(256), 0, s, dA, ehat, T, N, tau, dehat); break;
    case 64: hipLaunchKernelGGL(affinity_bwd_tile_kernel<2>, grid, dim3(256), 0, s, dA, ehat, T, N, tau, dehat); break;
    case 128: hipLaunchKernelGGL(affinity_bwd_tile_kernel<4>, grid, dim3(256), 0, s, dA, ehat, T, N, tau, dehat); break;
    default: return CRW_EINVAL;
  }
  return check_launch();
}

// out[e] = sum_b part[b][e], fixed order (deterministic)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float *__restrict__ part, int nb, int n, float *__restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= nb; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long)(b + u) * n + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < nb; ++b) s += part[(long)b * n + e];
  out[e] = s;
}

}  // namespace crw

// Weight gradient of the CNN encoder's linear head (nn.Linear(128, 128), src/encoder.py:40,55):
//   dW[o][i] = sum_p dy[p][o] x[p][i]      (M = N = 128, K = P)
// A library GEMM sees 16 output tiles and a K of P (hipBLASLt: 16 workgroups, ~100 us at P = 16128); here the sum over p
// is split into P/128 batched 128 x 128 fp32-MFMA products dy_b^T x_b (every CU busy) whose partial matrices are added
// in a fixed order.  P must be a multiple of 128; ws holds P/128 * 128 * 128 floats.
extern "C" size_t crw_linear128_wgrad_ws_bytes(int P) { return P < 128 ? 0 : (size_t)(P / 128) * 128 * 128 * sizeof(float); }

extern "C" int crw_linear128_wgrad(const float *dy, const float *x, float *dw, int P, void *ws, size_t ws_bytes,
                                   crw_stream_t stream) {
  crw::clear_stale_error();
  if (!dy || !x || !dw || !ws || P < 128 || P % 128) return CRW_EINVAL;
  if (ws_bytes < crw_linear128_wgrad_ws_bytes(P)) return CRW_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  crw::GemmGroup g{};
  g.nprob = 1;
  g.n = 128;
  g.batch = P / 128;
  crw::GemmProb &p = g.p[0];
  p.A = dy; p.B = x; p.C = static_cast<float *>(ws);
  p.sA = p.sB = p.sC = 128L * 128;
  p.ta = 1; p.tb = 0; p.beta = 0;
  CRW_TRY(crw::launch_gemm_group_f32(g, s));
  hipLaunchKernelGGL(crw::batch_sum_kernel, dim3(128 * 128 / 256), dim3(256), 0, s, static_cast<const float *>(ws), P / 128,
                     128 * 128, dw);
  return crw::check_launch();
}

extern "C" int crw_gemm_f32(const float *A, const float *B, float *C, int n, int batch, int transA, int transB,
                            int beta, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!A || !B || !C) return CRW_EINVAL;
  crw::GemmGroup g{};
  g.nprob = 1;
  g.n = n;
  g.batch = batch;
  crw::GemmProb &p = g.p[0];
  p.A = A; p.B = B; p.C = C;
  p.sA = p.sB = p.sC = (long)n * n;
  p.ta = transA; p.tb = transB; p.beta = beta;
  return crw::launch_gemm_group_f32(g, (hipStream_t)stream);
}
