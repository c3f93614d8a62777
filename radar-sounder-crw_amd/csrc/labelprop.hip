// Inference path: k-NN label propagation along the radargram (src/utils.py:134-161,
// src/imported/labelprop.py:67-115, src/imported/maskedatt.py:151-175) and the
// "horizontality" metric (src/utils.py:117-125).
//
// Restructured for the GPU: the reference recomputes, for every frame n, the affinity of frame n
// against the whole growing context and only then gathers labels.  Features never change, so
//   (1) crw_labelprop_topk computes the top-k weights/indices of ALL frames in one launch
//       (one workgroup per (query node, frame); only in-band keys |m-q| < radius are scored --
//       out-of-band keys carry logit -1e10/temp whose softmax weight is exactly 0 in fp32);
//   (2) crw_labelprop_gather is the only sequential part: a single workgroup walks the frames and
//       does the weighted label sums + argmax.
// Index quirk kept on purpose: indices address the truncated key list
// [frame 0, last cxt frames] but are applied to the untruncated label list.
#include "crw_common.h"
#include <cstdlib>

namespace crw {
namespace {

constexpr int MAX_KNN = 64;

// Nodes of a frame form a gh x gw grid (node = i * gw + j; a radargram's patch column: gw = 1).  A key (i', j') is in band for the
// query (i, j) when (i - i')^2 + (j - j')^2 < radius^2 (MaskedAttention.make, src/imported/maskedatt.py:232-245); candidates are the
// keys of the clipped bounding box of that disc, in node order, and box keys outside the disc never score (the reference gives them
// logit -1e10 / temp: softmax weight exactly 0).
__global__ __launch_bounds__(256) void labelprop_topk_kernel(const float *__restrict__ ehat, int T, int N, int C,
                                                             int cxt, int radius, float temp, int knn,
                                                             int first_frame, int gw, float *__restrict__ W,
                                                             int32_t *__restrict__ I) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int q = blockIdx.x, n = blockIdx.y + first_frame;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool trunc = n > cxt + 1;
  const int nf = trunc ? cxt + 1 : n;
  const int gh = N / gw, qi = q / gw, qj = q - qi * gw;
  const int lo = max(0, qi - radius + 1), hi = min(gh - 1, qi + radius - 1);        // box rows
  const int lj = max(0, qj - radius + 1), hj = min(gw - 1, qj + radius - 1);        // box columns
  const int bwj = hj - lj + 1;
  const int bw = (hi - lo + 1) * bwj;                                               // keys of the box per context frame
  const int ncand = nf * bw;

  float *qv = smem;                       // [C] (padded to a multiple of 4)
  float *val = smem + ((C + 3) & ~3);     // [ncand]
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ float sel_v[MAX_KNN];
  __shared__ int sel_i[MAX_KNN];

  for (int c = tid; c < C; c += 256) qv[c] = ehat[((long)n * N + q) * C + c];
  __syncthreads();

  // 16 lanes per candidate
  const int sub = tid & 15, grp = tid >> 4;  // 16 groups per block
  for (int cand = grp; cand < ncand; cand += 16) {
    const int p = cand / bw, r = cand - p * bw;
    const int mi = lo + r / bwj, mj = lj + r % bwj, m = mi * gw + mj;
    const bool inside = (mi - qi) * (mi - qi) + (mj - qj) * (mj - qj) < radius * radius;  // (always, on an N x 1 grid)
    const int frame = trunc ? (p == 0 ? 0 : n - cxt + (p - 1)) : p;
    const float *key = ehat + ((long)frame * N + m) * C;
    float d = 0.f;
    if ((C & 3) == 0) {
      for (int c = 4 * sub; c < C; c += 64) {
        const float4 kv = *reinterpret_cast<const float4 *>(key + c);
        const float4 qq = *reinterpret_cast<const float4 *>(qv + c);
        d += kv.x * qq.x + kv.y * qq.y + kv.z * qq.z + kv.w * qq.w;
      }
    } else {
      for (int c = sub; c < C; c += 16) d += key[c] * qv[c];
    }
    d += __shfl_xor(d, 8);
    d += __shfl_xor(d, 4);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 1);
    if (sub == 0) val[cand] = inside ? d / temp : -INFINITY;
  }
  __syncthreads();

  // knn rounds of block-wide arg-max (ties -> lowest candidate index)
  for (int j = 0; j < knn; ++j) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = tid; c < ncand; c += 256) {
      const float v = val[c];
      if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
      sel_v[j] = bv;
      sel_i[j] = bi;
      if (bi != 0x7fffffff && bv != -INFINITY) val[bi] = -INFINITY;
    }
    __syncthreads();
  }

  if (tid < knn) {
    const float vmax = sel_v[0];  // in-band key m == q always exists (radius >= 1)
    float ssum = 0.f;
    for (int j = 0; j < knn; ++j) ssum += (sel_v[j] == -INFINITY) ? 0.f : expf(sel_v[j] - vmax);
    const float v = sel_v[tid];
    float w = 0.f;
    int idx = 0;
    if (v != -INFINITY) {
      w = expf(v - vmax) / ssum;
      const int c = sel_i[tid], r = c % bw;
      idx = (c / bw) * N + (lo + r / bwj) * gw + lj + r % bwj;
    }
    const long o = ((long)(n - first_frame) * knn + tid) * N + q;
    W[o] = w;
    I[o] = idx;
  }
}

// ---- the same lists on an N x 1 grid (a radargram's patch column), scores on the fp32 matrix cores ------------------------------
// The kernel above spends its time in vector arithmetic: 16 lanes per (query, key) pair, ~30 instructions for 128 multiply-adds.
// Here a workgroup owns 16 consecutive query nodes of a frame; S = Q K^T for 16 keys at a time is 32 v_mfma_f32_16x16x4_f32 (exact
// fp32 products, fp32 accumulation -- the summation order differs from the vector kernel's, as that one's differs from the
// reference's BLAS), a wave per (context frame, 16-node key tile), the next tile's rows requested before this one's MFMAs.  In-band
// scores go to LDS as val[query][candidate] in the reference's candidate order; selection: a wave per PAIR of queries with their candidates in
// registers (TK_NV per lane and query) -- no workgroup barrier inside the knn rounds --, same rule: highest value, then lowest candidate index.
constexpr int TK_Q = 16, TK_NV = 32, TK_NT = 512, TK_NW = TK_NT / 64;  // eight waves: two per SIMD cover each other's round trips
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// NCH = 2: the context frames in two halves, one after the other through HALF the score buffer (two workgroups per CU: one's
// scoring -- matrix cores, loads, LDS scatter -- runs beside the other's selection -- vector compares), the first half's k best
// carried into the second selection as extra candidates: top-k(A u B) = top-k(top-k(A) u B), and the carried candidates have the
// lower indices, so "highest value, then lowest candidate index" picks the same list.  pf = context frames per chunk.
template <int CSTEPS, int NCH>  // C / 16: float4 per lane and row
__global__ __launch_bounds__(TK_NT, 2 * NCH) void labelprop_topk_mfma_kernel(const float *__restrict__ ehat, int T, int N, int cxt, int radius, float temp,
                                                                  int knn, int first_frame, int maxcand, int pf, float *__restrict__ W,
                                                                  int32_t *__restrict__ I) {
  constexpr int C = 16 * CSTEPS, NV = TK_NV / NCH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *val = smem;  // [TK_Q][maxcand] (maxcand: candidates of ONE chunk)
  __shared__ float sel_v[TK_Q][MAX_KNN];
  __shared__ int sel_i[TK_Q][MAX_KNN];
  const int q0 = blockIdx.x * TK_Q, nq = min(TK_Q, N - q0), n = blockIdx.y + first_frame;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const bool trunc = n > cxt + 1;
  const int nf = trunc ? cxt + 1 : n;

  const int ulo = max(0, q0 - radius + 1), uhi = min(N - 1, q0 + nq - 1 + radius - 1);
  const int kt_lo = ulo >> 4, nkt = (uhi >> 4) - kt_lo + 1;  // item = (context frame p, key tile kt)
  int lo_r[4], bw_r[4];  // bands of the four queries whose scores this lane receives (rows 4 g + r of the tile)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r, lo = max(0, q - radius + 1), hi = min(N - 1, q + radius - 1);
    lo_r[r] = lo;
    bw_r[r] = q < N ? hi - lo + 1 : 0;
  }
  // selection: wave w takes the queries w and w + 8 TOGETHER -- two independent chains of compares / cross-lane exchanges in one
  // instruction stream
  const int u0 = wave, u1 = wave + TK_NW;
  int lo2[2], bw2[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q = q0 + (h ? u1 : u0), lo = max(0, q - radius + 1), hi = min(N - 1, q + radius - 1);
    lo2[h] = lo;
    bw2[h] = hi - lo + 1;
  }

  for (int ch = 0; ch < NCH; ++ch) {
    const int p0 = ch * pf, np = min(nf - p0, NCH == 1 ? nf : pf);  // this chunk's context frames [p0, p0 + np)
    if (np <= 0) break;                                            // (block-uniform)
    const int nitems = np * nkt;
    // A operand: query row q0 + r16 (rows beyond the column repeat the last node: their scores are never stored); lane group g holds
    // elements 16 j + 4 g .. + 3 of every 16 -- the k order of the MFMA steps, the same for both operands (per chunk: not live
    // through the selection)
    float4 a[CSTEPS];
    {
      const float *qp = ehat + ((long)n * N + min(q0 + r16, N - 1)) * C + 4 * g;
#pragma unroll
      for (int j = 0; j < CSTEPS; ++j) a[j] = *reinterpret_cast<const float4 *>(qp + 16 * j);
    }
    auto fetch = [&](int item, float4 (&b)[CSTEPS]) {
      const int p = p0 + item / nkt, kt = kt_lo + item % nkt;
      const int frame = trunc ? (p == 0 ? 0 : n - cxt + (p - 1)) : p;
      const float *kp = ehat + ((long)frame * N + min(16 * kt + r16, N - 1)) * C + 4 * g;
#pragma unroll
      for (int j = 0; j < CSTEPS; ++j) b[j] = *reinterpret_cast<const float4 *>(kp + 16 * j);
    };
    // NCH = 1: the next item's key rows are requested before this one's MFMAs (two waves per SIMD); NCH = 2: four waves per SIMD
    // cover the round trip, and the second register set would not fit 128 registers
    float4 b[CSTEPS], bn[NCH == 1 ? CSTEPS : 1];
    if (NCH == 1 && wave < nitems) fetch(wave, b);
    for (int item = wave; item < nitems; item += TK_NW) {
      if constexpr (NCH == 1) {
        if (item + TK_NW < nitems) fetch(item + TK_NW, bn);
      } else {
        fetch(item, b);
      }
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // even / odd steps: two independent chains
#pragma unroll
      for (int j = 0; j < CSTEPS; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b[j].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b[j].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b[j].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b[j].w, acc1, 0, 0, 0);
      }
      const int pl = item / nkt, m = 16 * (kt_lo + item - pl * nkt) + r16;  // pl: frame slot inside the chunk
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = m - lo_r[r];
        if (m < N && rr >= 0 && rr < bw_r[r]) val[(4 * g + r) * maxcand + pl * bw_r[r] + rr] = (acc0[r] + acc1[r]) / temp;
      }
      if constexpr (NCH == 1) {
#pragma unroll
        for (int j = 0; j < CSTEPS; ++j) b[j] = bn[j];
      }
    }
    __syncthreads();

    {
      int nc2[2], base2[2];
      float v[2][NV], ve[2];
      int ie[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int u = h ? u1 : u0;
        nc2[h] = u < nq ? np * bw2[h] : 0;
        base2[h] = p0 * bw2[h];  // candidate index of the chunk's first candidate
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int c = lane + 64 * i;
          v[h][i] = c < nc2[h] ? val[u * maxcand + c] : -INFINITY;
        }
        // the k best of the chunks before (lane j holds the j-th): lower candidate indices than anything in this chunk
        const bool carried = ch > 0 && u < nq && lane < knn;
        ve[h] = carried ? sel_v[u][lane] : -INFINITY;
        ie[h] = carried ? sel_i[u][lane] : 0x7fffffff;
        if (ve[h] == -INFINITY) ie[h] = 0x7fffffff;
      }
      for (int j = 0; j < knn; ++j) {
        float bv[2];
        int bi[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float x = ve[h];
          int bs = -1;
#pragma unroll
          for (int i = 0; i < NV; ++i)
            if (v[h][i] > x) { x = v[h][i]; bs = i; }  // ties inside a lane: the carried one, then the lowest slot = the lowest index
          bv[h] = x;
          bi[h] = x == -INFINITY ? 0x7fffffff : (bs < 0 ? ie[h] : base2[h] + lane + 64 * bs);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float ov = __shfl_xor(bv[h], o);
            const int oi = __shfl_xor(bi[h], o);
            if (ov > bv[h] || (ov == bv[h] && oi < bi[h])) { bv[h] = ov; bi[h] = oi; }
          }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (lane == 0 && (h ? u1 : u0) < nq) {
            sel_v[h ? u1 : u0][j] = bv[h];
            sel_i[h ? u1 : u0][j] = bi[h];
          }
          if (bi[h] != 0x7fffffff) {
            if (ie[h] == bi[h]) {
              ve[h] = -INFINITY;
              ie[h] = 0x7fffffff;
            } else if (bi[h] >= base2[h] && ((bi[h] - base2[h]) & 63) == lane) {
              const int slot = (bi[h] - base2[h]) >> 6;
#pragma unroll
              for (int i = 0; i < NV; ++i)
                if (i == slot) v[h][i] = -INFINITY;
            }
          }
        }
      }
    }
    if (NCH > 1) __syncthreads();  // the next chunk's scores overwrite val; lane 0's sel stores are in LDS for the carried reads
  }
  {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // lane 0's stores are in LDS before the wave reads them back
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int u = h ? u1 : u0;
      if (u < nq && lane < knn) {
        const float vmax = sel_v[u][0];  // the query's own node is always in its band (radius >= 1)
        float ssum = 0.f;
        for (int j = 0; j < knn; ++j) ssum += (sel_v[u][j] == -INFINITY) ? 0.f : expf(sel_v[u][j] - vmax);
        const float x = sel_v[u][lane];
        float w = 0.f;
        int idx = 0;
        if (x != -INFINITY) {
          w = expf(x - vmax) / ssum;
          const int c = sel_i[u][lane];
          idx = (c / bw2[h]) * N + lo2[h] + c % bw2[h];
        }
        const long o = ((long)(n - first_frame) * knn + lane) * N + q0 + u;
        W[o] = w;
        I[o] = idx;
      }
    }
  }
}

template <int CSTEPS, int NCH>
int launch_topk_mfma_n(const float *ehat, int T, int N, int cxt, int radius, float temp, int knn, int first_frame, int chunkcand, int pf,
                       float *W, int32_t *I, hipStream_t s) {
  const size_t lds = (size_t)TK_Q * chunkcand * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void *)labelprop_topk_mfma_kernel<CSTEPS, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
        hipSuccess) {
      g_last_hip_error = (int)hipGetLastError();
      return CRW_EHIP;
    }
    attr = true;
  }
  hipLaunchKernelGGL((labelprop_topk_mfma_kernel<CSTEPS, NCH>), dim3((N + TK_Q - 1) / TK_Q, T - first_frame), dim3(TK_NT), lds, s, ehat, T, N,
                     cxt, radius, temp, knn, first_frame, chunkcand, pf, W, I);
  return check_launch();
}

// max_nf context frames of max_bi in-band keys: in one piece, or -- when the scores of a 16-query tile fill more than half a CU's LDS --
// in two halves, so that two workgroups share a CU (CRW_LABELPROP_TOPK_CHUNKS=1: always one piece, A/B)
template <int CSTEPS>
int launch_topk_mfma(const float *ehat, int T, int N, int cxt, int radius, float temp, int knn, int first_frame, int max_nf, int max_bi, float *W,
                     int32_t *I, hipStream_t s) {
  static const bool one = getenv("CRW_LABELPROP_TOPK_CHUNKS") && getenv("CRW_LABELPROP_TOPK_CHUNKS")[0] == '1';
  const long maxcand = (long)max_nf * max_bi;
  const int pf = (max_nf + 1) / 2;
  if (!one && CSTEPS <= 8 /* 256 channels: past 128 registers */ && (size_t)TK_Q * maxcand * 4 > 76 * 1024 && (long)pf * max_bi <= 64L * (TK_NV / 2))
    return launch_topk_mfma_n<CSTEPS, 2>(ehat, T, N, cxt, radius, temp, knn, first_frame, pf * max_bi, pf, W, I, s);
  return launch_topk_mfma_n<CSTEPS, 1>(ehat, T, N, cxt, radius, temp, knn, first_frame, (int)maxcand, max_nf, W, I, s);
}

__device__ inline float ld_l2(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void st_l2(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// single workgroup; frame n reads soft labels written for earlier frames by other waves of this
// same workgroup: stores/loads are agent-scope (L2) and separated by vmcnt(0) + barrier.
__global__ __launch_bounds__(1024) void labelprop_gather_kernel(const float *__restrict__ seed,
                                                                const float *__restrict__ W,
                                                                const int32_t *__restrict__ I, int T, int N, int M,
                                                                int knn, int first_frame, float *L,
                                                                float *__restrict__ pred) {
  const int tid = threadIdx.x, nt = blockDim.x;
  if (seed) {
    for (int it = tid; it < N * M; it += nt) {
      const int q = it / M, c = it % M;
      st_l2(L + it, seed[q] == (float)c ? 1.f : 0.f);
    }
    for (int q = tid; q < N; q += nt) pred[(long)q * T] = seed[q];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int n = first_frame; n < T; ++n) {
    const float *Wn = W + (long)(n - first_frame) * knn * N;
    const int32_t *In = I + (long)(n - first_frame) * knn * N;
    for (int it = tid; it < N * M; it += nt) {
      const int q = it / M, c = it % M;
      float p = 0.f;
      for (int j = 0; j < knn; ++j) p += ld_l2(L + (long)In[j * N + q] * M + c) * Wn[j * N + q];
      st_l2(L + ((long)n * N + q) * M + c, p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = tid; q < N; q += nt) {
      const float *row = L + ((long)n * N + q) * M;
      float bv = ld_l2(row);
      int bi = 0;
      for (int c = 1; c < M; ++c) {
        const float v = ld_l2(row + c);
        if (v > bv) { bv = v; bi = c; }
      }
      pred[(long)q * T + n] = (float)bi;
    }
  }
}

// The same propagation with the soft labels of the most recent R frames (and of frame 0, the long-term frame every step
// reads) kept in LDS, the step's (weight, index) lists prefetched one frame ahead into a double-buffered LDS copy, and the
// arg-max read back from LDS: a frame costs LDS traffic and ONE barrier where the kernel above pays three L2 round trips
// (gather, store acknowledge, arg-max reload) -- 10 us per frame at [T, N] = [256, 48].  Labels older than R frames are read
// from L in global memory (written there by every frame as before; they are long since visible).  Same operations in the
// same order: bit-identical results.
constexpr int GATHER_NT = 256, GATHER_PF = 8;  // threads; prefetch registers per thread (knn * N <= 2048)
constexpr int GATHER_CH = 20;                  // neighbours whose LDS reads are in flight together (KNN = 20: one batch per output; 4: 0.64, 10: 0.61, 20: 0.58 ms per cfg5 pass)
__global__ __launch_bounds__(GATHER_NT) void labelprop_gather_lds_kernel(const float *__restrict__ seed,
                                                                         const float *__restrict__ W,
                                                                         const int32_t *__restrict__ I, int T, int N, int M,
                                                                         int knn, int first_frame, float *L,
                                                                         float *__restrict__ pred, int R) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, NM = N * M, KN = knn * N;
  float *l0 = sm, *ring = l0 + NM, *wbuf = ring + (long)R * NM;  // [NM], [R][NM], [2][KN]
  int *ibuf = reinterpret_cast<int *>(wbuf + 2 * KN);              // [2][KN]
  for (int it = tid; it < NM; it += GATHER_NT) {
    float v;
    if (seed) {
      v = seed[it / M] == (float)(it % M) ? 1.f : 0.f;
      st_l2(L + it, v);
    } else {
      v = ld_l2(L + it);
    }
    l0[it] = v;
  }
  if (seed)
    for (int q = tid; q < N; q += GATHER_NT) pred[(long)q * T] = seed[q];
  for (int e = tid; e < KN; e += GATHER_NT) {
    wbuf[(first_frame & 1) * KN + e] = W[e];
    ibuf[(first_frame & 1) * KN + e] = I[e];
  }
  __syncthreads();
  const int RNM = R * NM;
  for (int n = first_frame; n < T; ++n) {
    const int cur = n & 1;
    // next frame's lists: global -> registers now, -> LDS after this frame's arithmetic (unconditional, clamped loads)
    const long nxt = (long)(min(n + 1, T - 1) - first_frame) * KN;
    float pw[GATHER_PF];
    int pi[GATHER_PF];
#pragma unroll
    for (int u = 0; u < GATHER_PF; ++u) {
      const int e = min(tid + u * GATHER_NT, KN - 1);
      pw[u] = W[nxt + e];
      pi[u] = I[nxt + e];
    }
    // frames [lo_f, n - 1] live in the ring; a frame f sits at flat offset (f % R) * NM
    const int lo_f = max(first_frame, n - R + 1);
    const int lo_idx = lo_f * N;                       // first label-list row that is in the ring
    const int epoch_base = (n / R - 1) * RNM;          // flat offset of the older of the (at most) two ring epochs in view
    const float *wn = wbuf + cur * KN;
    const int *in = ibuf + cur * KN;
    float *slot = ring + (n % R) * NM;
    for (int it = tid; it < NM; it += GATHER_NT) {
      const int q = it / M, c = it % M;
      float p = 0.f;
      // GATHER_CH neighbours at a time: their (index, weight) reads and then their label reads are independent LDS round trips
      // (one neighbour after the other is two dependent round trips each: 3 of the 4 us a frame took); the sum stays in j order
      for (int j0 = 0; j0 < knn; j0 += GATHER_CH) {
        int idx[GATHER_CH];
        float w[GATHER_CH], v[GATHER_CH];
#pragma unroll
        for (int u = 0; u < GATHER_CH; ++u) {
          const int jj = min(j0 + u, knn - 1);
          idx[u] = in[jj * N + q];
          w[u] = wn[jj * N + q];
        }
#pragma unroll
        for (int u = 0; u < GATHER_CH; ++u) {
          // frame 0 sits in l0 = sm[0, NM), the ring behind it; a label older than the ring comes from global memory
          int off = idx[u] * M + c - epoch_base;
          if (off >= RNM) off -= RNM;
          const bool first = idx[u] < N, old = !first && idx[u] < lo_idx;
          const int a = first ? idx[u] * M + c : (old ? 0 : NM + off);
          v[u] = sm[a];
          if (old) v[u] = ld_l2(L + (long)idx[u] * M + c);
        }
#pragma unroll
        for (int u = 0; u < GATHER_CH; ++u)
          if (j0 + u < knn) p += v[u] * w[u];
      }
      st_l2(L + ((long)n * N + q) * M + c, p);
      slot[it] = p;
    }
#pragma unroll
    for (int u = 0; u < GATHER_PF; ++u) {
      const int e = tid + u * GATHER_NT;
      if (e < KN) {
        wbuf[(cur ^ 1) * KN + e] = pw[u];
        ibuf[(cur ^ 1) * KN + e] = pi[u];
      }
    }
    __syncthreads();
    for (int q = tid; q < N; q += GATHER_NT) {
      const float *row = slot + q * M;
      float bv = row[0];
      int bi = 0;
      for (int c = 1; c < M; ++c) {
        const float v = row[c];
        if (v > bv) { bv = v; bi = c; }
      }
      pred[(long)q * T + n] = (float)bi;
    }
  }
}

// ---- propagation when the lists' context bound is known (crw_labelprop_propagate) -------------------------------------------------
// Index quirk Q7 read the other way round: crw_labelprop_topk's indices address the TRUNCATED key list [frame 0, last cxt frames],
// i.e. they are < min(n, cxt + 1) * N, and the reference applies them to the UNTRUNCATED label list (src/imported/labelprop.py:
// 103-107 with src/imported/maskedatt.py:165-166) -- so frame n reads the soft labels of frames 0 .. min(n - 1, cxt) and nothing
// else.  Only the first cxt frames form a chain (frame n needs frame n - 1); every frame n > cxt depends on frames 0 .. cxt alone.
//   prefix kernel: ONE workgroup walks frames first_frame .. cxt with ALL their labels in LDS (direct addressing, no ring).  Roles by
//     wave: compute waves own one output (query, class) per lane and do nothing but the dependent chain -- label reads (addresses
//     and weights already in registers), the sum in neighbour order, the LDS + global store, then the NEXT frame's (index, weight)
//     reads from an LDS copy while the barrier gathers; loader waves, one frame each in rotation, keep the lists of the next frames
//     in flight from global memory and copy them into a double-buffered LDS slot two frames ahead; one more wave takes the
//     arg-max / pred store of the previous frame.  The per-frame barrier waits for LDS traffic only (lds_barrier): global loads and
//     stores stay in flight across frames, where labelprop_gather_lds_kernel's __syncthreads() drained them every frame (2.4 us per frame at cfg5).
//   tail kernel: a workgroup per frame n > cxt, all at once, labels gathered from L in global memory (complete since the prefix).
// Same operations in the same order per output as the kernels above: bit-identical L and pred.
constexpr int PX_PF = 16, PX_NT = 512;  // list entries per loader lane and frame (a frame's lists: knn * N <= 64 * PX_PF); threads at most
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KP>
__global__ __launch_bounds__(PX_NT) void labelprop_prefix_kernel(const float *__restrict__ seed, const float *__restrict__ W,
                                                                   const int32_t *__restrict__ I, int T, int N, int M, int knn,
                                                                   int first_frame, int last_frame, float *L,
                                                                   float *__restrict__ pred, int ncw) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, NM = N * M, KN = knn * N;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave-uniform roles: compute waves [0, ncw), D list loaders [ncw, nwaves - 1), ONE arg-max wave (the last).
  // Loader k owns the frames f with (f - first_frame) % D == k: in the phase of frame f it copies the lists of frame f + 2 from its
  // registers into LDS and requests those of frame f + 2 + D -- a wave has ONE request in flight and touches it D phases later, so
  // the s_waitcnt vmcnt(0) hipcc puts in front of the copy meets loads that are long back.  (Measured the other ways: the arg-max's
  // pred stores in a loader wave -- two event types on one counter, hipcc waits for everything: 1.36 us per frame; several register
  // sets in rotation in one wave -- hipcc's counted waits across the loop's back edge come out as vmcnt(0..14): the same.)
  const int nwaves = (int)blockDim.x >> 6, D = nwaves - 1 - ncw;
  const bool compute = wave < ncw, loader = !compute && wave < nwaves - 1, amax = wave == nwaves - 1;
  const int lk = wave - ncw, at = tid - (nwaves - 1) * 64;
  constexpr int KNP = 64 * PX_PF;                           // entries of a list slot in LDS
  float *lab = sm;                                          // [(last_frame + 1) * NM] soft labels, frame-major like L
  // [2][KNP] (index, weight bits) pairs of frames f, f + 1 (slot = frame & 1), 8-byte aligned: one ds_read_b64 per neighbour
  int2 *pl = reinterpret_cast<int2 *>(lab + (((long)(last_frame + 1) * NM + 1) & ~1L));

  // labels of the frames before first_frame: frame 0 from the seed when given, the rest from L (filled by the caller)
  for (int it = tid; it < first_frame * NM; it += (int)blockDim.x) {
    float v;
    if (seed && it < NM) {
      v = seed[it / M] == (float)(it % M) ? 1.f : 0.f;
      L[it] = v;
    } else {
      v = L[it];
    }
    lab[it] = v;
  }
  if (seed)
    for (int q = tid; q < N; q += (int)blockDim.x) pred[(long)q * T] = seed[q];

  float pw[PX_PF];
  int pi[PX_PF];
  auto svc_load = [&](int f) {  // (clamped, unconditional: a frame past the last one re-reads the last one's lists, never used)
    const long o = (long)(min(f, last_frame) - first_frame) * KN;
#pragma unroll
    for (int u = 0; u < PX_PF; ++u) {
      const int e = min(lane + u * 64, KN - 1);
      pw[u] = W[o + e];
      pi[u] = I[o + e];
    }
  };
  auto svc_store = [&](int f) {  // unconditional too: a slot holds 64 * PX_PF >= KN entries, the surplus repeats the last entry
#pragma unroll
    for (int u = 0; u < PX_PF; ++u) pl[(f & 1) * KNP + lane + u * 64] = int2{pi[u], __float_as_int(pw[u])};
  };
  if (loader) {  // lists of the first two frames -> LDS; then every loader requests its first frame
    if (lk == 0) {
      svc_load(first_frame);
      svc_store(first_frame);
    }
    if (lk == (D > 1 ? 1 : 0)) {
      svc_load(first_frame + 1);
      svc_store(first_frame + 1);
    }
    svc_load(first_frame + 2 + lk);
  }
  lds_barrier();

  // compute lane: output `tid` = (q, c); LDS offsets and weights of its neighbours for the coming frame in registers
  // (the surplus lanes of the last compute wave repeat output NM - 1: same reads, same sum, same stores -- no branch around the
  // chain: under `if (it < NM)` hipcc sinks the label reads into the branch and waits for them one by one, 24 LDS round trips)
  const int it = min(tid, NM - 1), q = it / M, c = it % M;
  int a[KP];
  float w[KP];
  auto fetch_lists = [&](int f, int (&ix)[KP], float (&ww)[KP]) {  // frame f's lists: LDS -> registers (no dependence on labels)
    const int2 *ln = pl + (f & 1) * KNP;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      const int2 e = ln[min(j, knn - 1) * N + q];
      ix[j] = e.x;
      ww[j] = __int_as_float(e.y);
    }
  };
  auto to_offsets = [&](int f, int (&ix)[KP], float (&ww)[KP]) {
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      ix[j] = min(max(ix[j], 0), f * N - 1) * M + c;  // in range whatever the lists hold (topk: < min(f, cxt + 1) * N)
      if (j >= knn) ww[j] = 0.f;                      // padding slot: p + v * 0 = p
    }
  };
  if (compute) {
    fetch_lists(first_frame, a, w);
    to_offsets(first_frame, a, w);
  }

  int turn = 0;  // (f - first_frame) % D
  for (int f = first_frame; f <= last_frame; ++f) {
    if (compute) {
      float v[KP];
#pragma unroll
      for (int j = 0; j < KP; ++j) v[j] = lab[a[j]];
      int na[KP];
      float nw[KP];
      fetch_lists(f + 1, na, nw);  // slot (f + 1) & 1: written before the previous barrier (past last_frame: unused)
      float p = 0.f;
#pragma unroll
      for (int j = 0; j < KP; ++j) p += v[j] * w[j];
      lab[(long)f * NM + it] = p;
      L[(long)f * NM + it] = p;
      to_offsets(f + 1, na, nw);
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        a[j] = na[j];
        w[j] = nw[j];
      }
    } else if (loader) {
      if (turn == lk) {
        svc_store(f + 2);  // slot f & 1: last read (frame f's lists) before the previous barrier
        svc_load(f + 2 + D);
      }
    } else if (f > first_frame) {  // arg-max of the previous frame (first maximum wins, like torch.argmax on distinct values)
      for (int qq = at; qq < N; qq += 64) {
        const float *row = lab + ((long)(f - 1) * N + qq) * M;
        float bv = row[0];
        int bi = 0;
        for (int cc = 1; cc < M; ++cc) {
          const float x = row[cc];
          if (x > bv) { bv = x; bi = cc; }
        }
        pred[(long)qq * T + f - 1] = (float)bi;
      }
    }
    turn = turn + 1 == D ? 0 : turn + 1;
    lds_barrier();
  }
  if (amax) {
    for (int qq = at; qq < N; qq += 64) {
      const float *row = lab + ((long)last_frame * N + qq) * M;
      float bv = row[0];
      int bi = 0;
      for (int cc = 1; cc < M; ++cc) {
        const float x = row[cc];
        if (x > bv) { bv = x; bi = cc; }
      }
      pred[(long)qq * T + last_frame] = (float)bi;
    }
  }
}

// frames t0 .. T-1, none of which reads a label this launch writes: a workgroup per frame, neighbours' labels from L (global memory)
constexpr int TAIL_NT = 256, TAIL_CH = 16;
__global__ __launch_bounds__(TAIL_NT) void labelprop_tail_kernel(const float *__restrict__ W, const int32_t *__restrict__ I, int T, int N,
                                                                 int M, int knn, int first_frame, int t0, float *L,
                                                                 float *__restrict__ pred) {
  extern __shared__ __attribute__((aligned(16))) float slot[];  // [N * M]
  const int tid = threadIdx.x, NM = N * M, n = t0 + blockIdx.x;
  const float *Wn = W + (long)(n - first_frame) * knn * N;
  const int32_t *In = I + (long)(n - first_frame) * knn * N;
  const int row_lim = n * N - 1;
  for (int it = tid; it < NM; it += TAIL_NT) {
    const int q = it / M, c = it % M;
    float p = 0.f;
    for (int j0 = 0; j0 < knn; j0 += TAIL_CH) {
      int idx[TAIL_CH];
      float w[TAIL_CH], v[TAIL_CH];
#pragma unroll
      for (int u = 0; u < TAIL_CH; ++u) {
        const int jj = min(j0 + u, knn - 1);
        idx[u] = min(max(In[jj * N + q], 0), row_lim);
        w[u] = Wn[jj * N + q];
      }
#pragma unroll
      for (int u = 0; u < TAIL_CH; ++u) v[u] = L[(long)idx[u] * M + c];
#pragma unroll
      for (int u = 0; u < TAIL_CH; ++u)
        if (j0 + u < knn) p += v[u] * w[u];
    }
    L[((long)n * N + q) * M + c] = p;
    slot[it] = p;
  }
  __syncthreads();
  for (int q = tid; q < N; q += TAIL_NT) {
    const float *row = slot + q * M;
    float bv = row[0];
    int bi = 0;
    for (int c = 1; c < M; ++c) {
      const float x = row[c];
      if (x > bv) { bv = x; bi = c; }
    }
    pred[(long)q * T + n] = (float)bi;
  }
}

// xent[a, i] = logsumexp_c A_i[c, a] - A_i[a, a],  A_i[c, a] = <ehat[i,c,0:C-1], ehat[i,a,1:C]> / 0.1
__global__ __launch_bounds__(64) void xent_metric_kernel(const float *__restrict__ ehat, int T, int N, int C,
                                                         float *__restrict__ xent) {
  extern __shared__ float col[];  // [N]
  const int a = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
  const float *ea = ehat + ((long)i * N + a) * C + 1;
  float m = -INFINITY;
  for (int c = 0; c < N; ++c) {
    const float *ec = ehat + ((long)i * N + c) * C;
    float d = 0.f;
    for (int ch = lane; ch < C - 1; ch += 64) d += ec[ch] * ea[ch];
    d = wave_sum(d) / 0.1f;
    if (lane == 0) col[c] = d;
    m = fmaxf(m, d);
  }
  __syncthreads();
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += expf(col[c] - m);
  s = wave_sum(s);
  if (lane == 0) xent[(long)a * (T - 1) + i] = (logf(s) + m) - col[a];
}

// The same metric with a workgroup per FRAME: the frame's features sit in LDS twice (as they are, and shifted by one channel, so both
// operands of a pair are read as aligned float4), every thread takes N * N / 256 (c, a) pairs, then a thread per column does the
// log-sum-exp.  12240 one-wave workgroups with 48 serial wave reductions each took 92 us at cfg5 -- on the way to the host's
// change-point search, which is the tail of that step.
constexpr int XENT_NT = 256;
__global__ __launch_bounds__(XENT_NT) void xent_metric_frame_kernel(const float *__restrict__ ehat, int T, int N, int C, float *__restrict__ xent) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LD = ((C + 3) & ~3) + 4;  // row stride: a multiple of 4 floats, consecutive rows 4 banks apart
  float *e0 = sm, *e1 = e0 + N * LD, *A = e1 + N * LD;  // e1[a][k] = e[a][k + 1]; A[c][a]
  const int i = blockIdx.x, tid = threadIdx.x;
  const float *src = ehat + (long)i * N * C;
  for (int it = tid; it < N * LD; it += XENT_NT) {
    const int r = it / LD, k = it - r * LD;
    e0[it] = k < C - 1 ? src[r * C + k] : 0.f;       // channels 0 .. C-2, zero beyond
    e1[it] = k < C - 1 ? src[r * C + k + 1] : 0.f;   // channels 1 .. C-1
  }
  __syncthreads();
  const int K4 = (C - 1 + 3) >> 2;
  for (int p = tid; p < N * N; p += XENT_NT) {
    const int c = p / N, a = p - c * N;
    const float4 *x = reinterpret_cast<const float4 *>(e0 + c * LD), *y = reinterpret_cast<const float4 *>(e1 + a * LD);
    float d = 0.f;
    for (int k = 0; k < K4; ++k) {
      const float4 u = x[k], v = y[k];
      d += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
    A[c * N + a] = d / 0.1f;
  }
  __syncthreads();
  for (int a = tid; a < N; a += XENT_NT) {
    float m = -INFINITY;
    for (int c = 0; c < N; ++c) m = fmaxf(m, A[c * N + a]);
    float sum = 0.f;
    for (int c = 0; c < N; ++c) sum += expf(A[c * N + a] - m);
    xent[(long)a * (T - 1) + i] = (logf(sum) + m) - A[a * N + a];
  }
}

}  // namespace
}  // namespace crw

using namespace crw;

extern "C" {

int crw_labelprop_topk_grid(const float *ehat, int T, int N, int C, int cxt_size, int radius, float temp, int knn, int first_frame,
                            int grid_w, float *W, int32_t *I, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!ehat || !W || !I || T < 2 || N < 1 || C < 1 || cxt_size < 1 || radius < 1 || knn < 1 || knn > MAX_KNN ||
      !(temp > 0.f) || first_frame < 1 || first_frame >= T || grid_w < 1 || N % grid_w)
    return CRW_EINVAL;
  const long gh = N / grid_w;
  const long max_nf = (long)(cxt_size + 1 < T - 1 ? cxt_size + 1 : T - 1);
  const long max_bi = (2L * radius - 1 < gh) ? 2L * radius - 1 : gh, max_bj = (2L * radius - 1 < grid_w) ? 2L * radius - 1 : grid_w;
  const size_t lds = (((size_t)C + 3) & ~(size_t)3) * 4 + (size_t)(max_nf * max_bi * max_bj) * 4;
  // a radargram's patch column with at least one full tile of queries: scores on the fp32 matrix cores (labelprop_topk_mfma_kernel).
  // CRW_LABELPROP_TOPK_VALU=1 keeps the vector kernel (A/B)
  static const bool valu = getenv("CRW_LABELPROP_TOPK_VALU") && getenv("CRW_LABELPROP_TOPK_VALU")[0] == '1';
  const long maxcand = max_nf * max_bi;
  if (grid_w == 1 && !valu && N >= TK_Q && (C == 64 || C == 128 || C == 256) && maxcand <= 64L * TK_NV &&
      (size_t)TK_Q * maxcand * 4 <= 150 * 1024 && (((uintptr_t)ehat) & 15) == 0) {
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) return launch_topk_mfma<4>(ehat, T, N, cxt_size, radius, temp, knn, first_frame, (int)max_nf, (int)max_bi, W, I, s);
    if (C == 128) return launch_topk_mfma<8>(ehat, T, N, cxt_size, radius, temp, knn, first_frame, (int)max_nf, (int)max_bi, W, I, s);
    return launch_topk_mfma<16>(ehat, T, N, cxt_size, radius, temp, knn, first_frame, (int)max_nf, (int)max_bi, W, I, s);
  }
  if (lds > 60 * 1024) return CRW_EINVAL;
  hipLaunchKernelGGL(labelprop_topk_kernel, dim3(N, T - first_frame), dim3(256), lds, (hipStream_t)stream, ehat, T,
                     N, C, cxt_size, radius, temp, knn, first_frame, grid_w, W, I);
  return check_launch();
}

int crw_labelprop_topk(const float *ehat, int T, int N, int C, int cxt_size, int radius, float temp, int knn,
                       int first_frame, float *W, int32_t *I, crw_stream_t stream) {
  return crw_labelprop_topk_grid(ehat, T, N, C, cxt_size, radius, temp, knn, first_frame, 1, W, I, stream);
}

int crw_labelprop_gather(const float *seed, const float *W, const int32_t *I, int T, int N, int M, int knn,
                         int first_frame, float *L, float *pred, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!W || !I || !L || !pred || T < 2 || N < 1 || M < 1 || knn < 1 || first_frame < 1 || first_frame >= T ||
      (!seed && first_frame < 1))
    return CRW_EINVAL;
  // LDS-resident form when the lists of a frame fit the prefetch registers and at least a few frames fit the ring
  const long NM = (long)N * M, KN = (long)knn * N;
  const long budget = 150 * 1024 - 16 * KN;  // bytes left for frame 0 + the ring after the two (weight, index) list copies
  long R = budget > 0 ? budget / (4 * NM) - 1 : 0;
  if (R > T) R = T;
  static const char *force_global = getenv("CRW_LABELPROP_GATHER_GLOBAL");  // diagnostics: the L2-round-trip kernel
  if (KN <= (long)GATHER_NT * GATHER_PF && R >= 4 && (long)T * N < (1L << 30) / M && !force_global) {
    const size_t lds = (size_t)(4 * (NM * (R + 1) + 4 * KN));
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void *)labelprop_gather_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
          hipSuccess) {
        g_last_hip_error = (int)hipGetLastError();
        return CRW_EHIP;
      }
      attr = true;
    }
    hipLaunchKernelGGL(labelprop_gather_lds_kernel, dim3(1), dim3(GATHER_NT), lds, (hipStream_t)stream, seed, W, I, T, N, M,
                       knn, first_frame, L, pred, (int)R);
    return check_launch();
  }
  hipLaunchKernelGGL(labelprop_gather_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, seed, W, I, T, N, M, knn,
                     first_frame, L, pred);
  return check_launch();
}

int crw_labelprop_propagate(const float *seed, const float *W, const int32_t *I, int T, int N, int M, int knn, int first_frame,
                            int cxt_size, float *L, float *pred, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!W || !I || !L || !pred || T < 2 || N < 1 || M < 1 || knn < 1 || first_frame < 1 || first_frame >= T || cxt_size < 1)
    return CRW_EINVAL;
  const long NM = (long)N * M, KN = (long)knn * N;
  // frames from t0 on read no label of this call: one frame alone, or every frame beyond the context bound
  const int t0 = (T - first_frame == 1) ? first_frame : (first_frame > cxt_size + 1 ? first_frame : (cxt_size + 1 < T ? cxt_size + 1 : T));
  const int last = t0 - 1;  // the chained frames first_frame .. last (none when last < first_frame)
  static const bool seq = getenv("CRW_LABELPROP_GATHER_SEQ") && getenv("CRW_LABELPROP_GATHER_SEQ")[0] == '1';  // A/B: one-workgroup walk
  // chained frames: compute waves + 1..4 list loaders (fewer when the outputs need more compute waves) + the arg-max wave
  const int ncw = (int)((NM + 63) / 64), nld = PX_NT / 64 - 1 - ncw < 4 ? PX_NT / 64 - 1 - ncw : 4;
  const size_t px_lds = (size_t)((((long)(last + 1) * NM + 1) & ~1L) * 4 + 16L * 64 * PX_PF);
  const bool px_ok = last < first_frame || (knn <= 24 /* 32 neighbours in registers spill */ && nld >= 1 && KN <= 64 * PX_PF &&
                                            px_lds <= 150 * 1024);
  if (seq || !px_ok || NM * 4 > 60 * 1024 || (long)T * N >= (1L << 30) / M)
    return crw_labelprop_gather(seed, W, I, T, N, M, knn, first_frame, L, pred, stream);
  hipStream_t s = (hipStream_t)stream;
  if (last >= first_frame) {
    static bool attr[3] = {false, false, false};  // dynamic-LDS limit raised, per instantiation
    auto launch = [&](auto kern, int which) -> int {
      if (!attr[which]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
          g_last_hip_error = (int)hipGetLastError();
          return CRW_EHIP;
        }
        attr[which] = true;
      }
      hipLaunchKernelGGL(kern, dim3(1), dim3((ncw + nld + 1) * 64), px_lds, s, seed, W, I, T, N, M, knn, first_frame, last, L, pred, ncw);
      return check_launch();
    };
    int rc;
    if (knn <= 8) rc = launch(labelprop_prefix_kernel<8>, 0);
    else if (knn <= 16) rc = launch(labelprop_prefix_kernel<16>, 1);
    else rc = launch(labelprop_prefix_kernel<24>, 2);
    if (rc != CRW_OK) return rc;
  } else if (seed) {  // a seed with a later first frame and no chained frame: the general kernel initialises frame 0
    return crw_labelprop_gather(seed, W, I, T, N, M, knn, first_frame, L, pred, stream);
  }
  if (t0 < T) {
    hipLaunchKernelGGL(labelprop_tail_kernel, dim3(T - t0), dim3(TAIL_NT), (size_t)NM * 4, s, W, I, T, N, M, knn, first_frame, t0, L, pred);
    return check_launch();
  }
  return CRW_OK;
}

int crw_xent_metric(const float *ehat, int T, int N, int C, float *xent, crw_stream_t stream) {
  crw::clear_stale_error();
  if (!ehat || !xent || T < 2 || N < 1 || C < 2 || (size_t)N * 4 > 60 * 1024) return CRW_EINVAL;
  const size_t ld = (((size_t)C + 3) & ~(size_t)3) + 4, frame_lds = (2 * (size_t)N * ld + (size_t)N * N) * 4;
  if (frame_lds <= 64 * 1024) {  // a frame's features fit LDS twice: a workgroup per frame
    hipLaunchKernelGGL(xent_metric_frame_kernel, dim3(T - 1), dim3(XENT_NT), frame_lds, (hipStream_t)stream, ehat, T, N, C, xent);
    return check_launch();
  }
  hipLaunchKernelGGL(xent_metric_kernel, dim3(N, T - 1), dim3(64), (size_t)N * 4, (hipStream_t)stream, ehat, T, N, C,
                     xent);
  return check_launch();
}

}  // extern "C"
