// Host-side change-point search of `utils.propagate` (src/utils.py:125-132: ruptures.Pelt(model="rbf").fit(d).predict(pen)):
// the arithmetic of pelt.py (the restatement of the published algorithms -- PELT, Killick et al. 2012; kernel segment cost,
// Arlot et al. 2019; ruptures' documented defaults) in the same order of operations, as plain C++ on doubles.  No GPU work: it
// runs on the host while the GPU propagates labels, and at BASELINE config 5 the numpy form (0.9 ms) had become the tail of the
// step once the label propagation itself took 0.45 ms.  PARITY UNPINNED like pelt.py (no ruptures build here); the tests hold it to
// pelt.py's breakpoints on random and structured signals.
//
// Same sums in the same order as pelt.py (column sums of the Gram matrix row after row, then running sums along the grid rows),
// compiled without floating-point contraction; exp() is libm's where numpy uses its own vector routine (both within an ulp), so a
// cost can differ in the last bit -- a breakpoint only where two segmentations tie to 1e-16.
#include "crw_hip.h"
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#pragma STDC FP_CONTRACT OFF

static int pelt_rbf_impl(const double *signal, int n, double pen, int min_size, int jump, double gamma, int *bkps, int max_bkps) {
  // (the Gram matrix is n x n doubles, like pelt.py's: 2 GiB at the bound)
  if (!signal || !bkps || n < 1 || n > 16384 || min_size < 1 || jump < 1 || max_bkps < 1) return -CRW_EINVAL;
  const size_t N = (size_t)n;
  // Gram matrix K_ij = exp(-clip(gamma (x_i - x_j)^2, 1e-2, 1e2)), K_ii = 1; gamma <= 0: 1 / median of the pairwise squared
  // distances (mean of the two middle elements), 1 if that median is 0
  std::vector<double> d2(N * N);
  for (size_t i = 0; i < N; ++i)
    for (size_t j = 0; j < N; ++j) {
      const double d = signal[i] - signal[j];
      d2[i * N + j] = d * d;
    }
  if (!(gamma > 0)) {
    std::vector<double> v;
    v.reserve(N * (N - 1) / 2);
    for (size_t i = 0; i < N; ++i)
      for (size_t j = i + 1; j < N; ++j) v.push_back(d2[i * N + j]);
    double med = 0.0;
    const size_t m = v.size();
    if (m) {
      std::nth_element(v.begin(), v.begin() + m / 2, v.end());
      med = v[m / 2];
      if (m % 2 == 0) med = (*std::max_element(v.begin(), v.begin() + m / 2) + med) / 2.0;
    }
    gamma = med != 0 ? 1.0 / med : 1.0;
  }
  std::vector<double> &gram = d2;
  for (size_t i = 0; i < N; ++i) {
    for (size_t j = i + 1; j < N; ++j) {
      double t = d2[i * N + j] * gamma;
      t = t < 1e-2 ? 1e-2 : (t > 1e2 ? 1e2 : t);
      const double e = std::exp(-t);
      gram[i * N + j] = e;
      gram[j * N + i] = e;  // (x_i - x_j)^2 == (x_j - x_i)^2 bit for bit
    }
    gram[i * N + i] = 1.0;
  }
  // admissible end points and the grid of positions the costs are needed on
  std::vector<int> ends;
  for (int k = 0; k < n; k += jump)
    if (k >= min_size) ends.push_back(k);
  ends.push_back(n);
  auto start_of = [&](int b) { return (int)std::floor((double)(b - min_size) / jump) * jump; };
  std::vector<int> grid(ends);
  for (int b : ends)
    if (start_of(b) >= 0) grid.push_back(start_of(b));
  grid.push_back(0);
  std::sort(grid.begin(), grid.end());
  grid.erase(std::unique(grid.begin(), grid.end()), grid.end());
  const int m = (int)grid.size();
  auto pos = [&](int v) { return (int)(std::lower_bound(grid.begin(), grid.end(), v) - grid.begin()); };
  // S on the grid: S[a][b] = sum of gram[:grid[a], :grid[b]] -- c0 = running column sums row after row, then running sums
  // along each grid row (the order of np.cumsum(axis=0) followed by np.cumsum(axis=1) on the selected rows)
  std::vector<double> Sg((size_t)m * m, 0.0), c0(N, 0.0), run(N);
  {
    int gi = 0;
    while (gi < m && grid[gi] == 0) ++gi;
    for (size_t i = 0; i < N && gi < m; ++i) {
      for (size_t j = 0; j < N; ++j) c0[j] = i ? c0[j] + gram[i * N + j] : gram[j];
      if ((int)i + 1 == grid[gi]) {
        double acc = 0.0;
        for (size_t j = 0; j < N; ++j) {
          acc = j ? acc + c0[j] : c0[0];
          run[j] = acc;
        }
        for (int b = 0; b < m; ++b)
          if (grid[b] > 0) Sg[(size_t)gi * m + b] = run[grid[b] - 1];
        ++gi;
      }
    }
  }
  std::vector<double> dg(m);
  {
    // np.cumsum of the diagonal (all ones, but summed like pelt.py does)
    std::vector<double> dfull(N + 1, 0.0);
    for (size_t i = 0; i < N; ++i) dfull[i + 1] = i ? dfull[i] + gram[i * N + i] : gram[0];
    for (int a = 0; a < m; ++a) dg[a] = dfull[grid[a]];
  }
  auto cost = [&](int a, int b) {  // segment [grid[a], grid[b])
    const double block = ((Sg[(size_t)b * m + b] - Sg[(size_t)a * m + b]) - Sg[(size_t)b * m + a]) + Sg[(size_t)a * m + a];
    return (dg[b] - dg[a]) - block / (double)(grid[b] - grid[a]);
  };
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> total(m, inf);
  std::vector<int> prev(m, -1), admissible, keep;
  std::vector<std::pair<double, int>> cand;
  total[0] = 0.0;
  for (int bkp : ends) {
    const int b = pos(bkp), start = start_of(bkp);
    if (start >= 0) admissible.push_back(pos(start));
    double best = inf;
    int best_t = -1;
    cand.clear();
    for (int t : admissible) {
      if (total[t] == inf || bkp - grid[t] < min_size) continue;
      const double c = (total[t] + cost(t, b)) + pen;
      cand.emplace_back(c, t);
      if (c < best) {  // the first minimum
        best = c;
        best_t = t;
      }
    }
    if (cand.empty()) continue;
    total[b] = best;
    prev[b] = best_t;
    keep.clear();
    for (auto &ct : cand)
      if (ct.first <= best + pen) keep.push_back(ct.second);  // PELT pruning
    admissible.swap(keep);
  }
  int cnt = 0;
  if (total[pos(n)] == inf) {
    bkps[cnt++] = n;
    return cnt;
  }
  std::vector<int> out;
  for (int t = pos(n); t > 0; t = prev[t]) out.push_back(grid[t]);
  std::sort(out.begin(), out.end());
  if ((int)out.size() > max_bkps) return -CRW_EINVAL;
  for (int v : out) bkps[cnt++] = v;
  return cnt;
}

extern "C" int crw_pelt_rbf(const double *signal, int n, double pen, int min_size, int jump, double gamma, int *bkps, int max_bkps) {
  try {  // nothing is thrown across the ABI (the vectors above allocate)
    return pelt_rbf_impl(signal, n, pen, min_size, jump, gamma, bkps, max_bkps);
  } catch (...) {
    return -CRW_EWORKSPACE;
  }
}
