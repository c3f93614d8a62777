"""Penalised change-point detection used by ``utils.propagate`` for its ``change_idx`` output
(src/utils.py:125-132 calls ``ruptures.Pelt(model="rbf").fit(d).predict(pen=5)``).

``ruptures`` is an un-vendored, un-pinned dependency of the reference (pyproject.toml:18) and is not
installed here, so this is a restatement of the PUBLISHED algorithms it implements, with its
documented defaults (``min_size=2``, ``jump=5``, median-heuristic bandwidth):
  * PELT -- Killick, Fearnhead & Eckley, "Optimal detection of changepoints with a linear
    computational cost", JASA 2012: dynamic programme over admissible last-change positions with pruning;
  * kernel (RBF) segment cost -- Arlot, Celisse & Harchaoui, "A kernel multiple change-point algorithm
    via model selection", JMLR 2019:  c(a,b) = sum_i K_ii - (1/(b-a)) sum_ij K_ij over [a,b),
    K_ij = exp(-gamma |x_i - x_j|^2), gamma = 1 / median pairwise squared distance, gamma*d^2 clipped to
    [1e-2, 1e2].
PARITY UNPINNED: no ruptures build or golden vector is available to check the breakpoints against; when
``ruptures`` is importable ``utils.change_point`` uses it instead of this module.  Host-side numpy on a
signal of T-2 samples -- not part of the GPU hot path.
"""
import math

import numpy as np


_TRIU = {}


def rbf_gram(signal, gamma=None):
    x = np.asarray(signal, dtype=np.float64).reshape(len(signal), -1)
    n = len(x)
    if x.shape[1] == 1:
        d2 = x - x.T
        d2 *= d2
    else:
        d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    if gamma is None:
        iu = _TRIU.get(n)
        if iu is None:
            iu = _TRIU[n] = np.triu_indices(n, 1)
        v = np.sort(d2[iu])  # median of the pairwise squared distances (np.median's value: mean of the two middle elements)
        m = len(v)
        med = 0.0 if m == 0 else (v[m // 2] if m % 2 else (v[m // 2 - 1] + v[m // 2]) / 2.0)
        gamma = 1.0 / med if med != 0 else 1.0
    d2 *= gamma
    np.clip(d2, 1e-2, 1e2, out=d2)
    np.negative(d2, out=d2)
    g = np.exp(d2, out=d2)
    np.fill_diagonal(g, 1.0)
    return g


def pelt_rbf(signal, pen, min_size=2, jump=5, gamma=None):
    """-> sorted breakpoints (segment ends, the last one is len(signal)).

    Every admissible change position and every end point is a multiple of `jump` (or n), so the segment costs are evaluated on that
    grid only: block sums from 2-D prefix sums S[i, j] = sum of gram[:i, :j] taken at the grid rows / columns (row by row the same
    sequential additions as a full cumulative sum), the cost matrix of all (start, end) grid pairs in a handful of array
    expressions, and the dynamic programme itself on plain floats -- the host runs this while the GPU propagates labels, and at
    BASELINE config 5 it was the tail of the step."""
    n = len(signal)
    gram = rbf_gram(signal, gamma)
    ends = [k for k in range(0, n, jump) if k >= min_size] + [n]
    grid = sorted(set(ends) | {v for v in (int(math.floor((b - min_size) / jump)) * jump for b in ends) if v >= 0} | {0})
    pos = {v: i for i, v in enumerate(grid)}
    g = np.asarray(grid)
    m = len(grid)
    # S on the grid: S[0, :] = S[:, 0] = 0; row i >= 1 of the full table is the running sum along the columns of c0[i - 1]
    Sg = np.zeros((m, m))
    if m > 1 and n > 0:
        c0 = np.cumsum(gram, axis=0)
        nz = g > 0
        rows = np.cumsum(c0[g[nz] - 1], axis=1)
        Sg[np.ix_(nz, nz)] = rows[:, g[nz] - 1]
    dfull = np.concatenate([[0.0], np.cumsum(np.diagonal(gram))])
    dg = dfull[g]
    Sd = np.diagonal(Sg)
    with np.errstate(divide="ignore", invalid="ignore"):
        # cost[a, b] of the segment [grid[a], grid[b]): (diag[b] - diag[a]) - (S[b,b] - S[a,b] - S[b,a] + S[a,a]) / (b - a)
        block = Sd[None, :] - Sg - Sg.T + Sd[:, None]
        cost = ((dg[None, :] - dg[:, None]) - block / (g[None, :] - g[:, None])).tolist()
    # total[i] = penalised cost of the best segmentation of signal[:grid[i]], prev[i] = its last change position (grid index)
    inf = float("inf")
    total = [inf] * m
    prev = [-1] * m
    total[0] = 0.0
    admissible = []
    for bkp in ends:
        b = pos[bkp]
        start = int(math.floor((bkp - min_size) / jump)) * jump
        if start >= 0:  # (a negative position is never the end of a segmentation)
            admissible.append(pos[start])
        best, best_t, cand = inf, -1, []
        for t in admissible:
            if total[t] == inf or bkp - grid[t] < min_size:
                continue
            c = total[t] + cost[t][b] + pen
            cand.append((c, t))
            if c < best:  # the first minimum, as min() over the candidates in order
                best, best_t = c, t
        if not cand:
            continue
        total[b] = best
        prev[b] = best_t
        admissible = [t for c, t in cand if c <= best + pen]  # PELT pruning
    if total[pos[n]] == inf:
        return [n]
    bkps = []
    t = pos[n]
    while t > 0:
        bkps.append(grid[t])
        t = prev[t]
    return sorted(bkps)
