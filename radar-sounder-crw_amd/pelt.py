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
    """-> sorted breakpoints (segment ends, the last one is len(signal))."""
    n = len(signal)
    gram = rbf_gram(signal, gamma)
    # prefix sums: S[i, j] = sum of gram[:i, :j]  ->  block sums in O(1)
    S = np.zeros((n + 1, n + 1))
    np.cumsum(np.cumsum(gram, axis=0), axis=1, out=S[1:, 1:])
    diag = np.concatenate([[0.0], np.cumsum(np.diagonal(gram))])

    def cost(a, b):
        block = S[b, b] - S[a, b] - S[b, a] + S[a, a]
        return (diag[b] - diag[a]) - block / (b - a)

    # total[t] = penalised cost of the best segmentation of signal[:t], prev[t] = its last change position (-1: none yet).  The
    # admissible last-change positions of one end point are costed in ONE vectorised expression (same arithmetic per element as
    # cost() above; np.argmin takes the first minimum, as min() over the list did)
    total = np.full(n + 1, np.inf)
    prev = np.full(n + 1, -1, dtype=np.int64)
    total[0] = 0.0
    Sd = np.diagonal(S)
    admissible = np.zeros(0, dtype=np.int64)
    ends = [k for k in range(0, n, jump) if k >= min_size] + [n]
    for bkp in ends:
        admissible = np.append(admissible, int(math.floor((bkp - min_size) / jump)) * jump)
        t = admissible[np.isfinite(total[admissible]) & (bkp - admissible >= min_size)]
        if t.size == 0:
            continue
        block = S[bkp, bkp] - S[t, bkp] - S[bkp, t] + Sd[t]
        cand = total[t] + ((diag[bkp] - diag[t]) - block / (bkp - t)) + pen
        k = int(np.argmin(cand))
        total[bkp] = cand[k]
        prev[bkp] = t[k]
        admissible = t[cand <= cand[k] + pen]  # PELT pruning
    if not np.isfinite(total[n]):
        return [n]
    bkps = []
    t = n
    while t > 0:
        bkps.append(int(t))
        t = int(prev[t])
    return sorted(bkps)
