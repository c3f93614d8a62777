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


def rbf_gram(signal, gamma=None):
    x = np.asarray(signal, dtype=np.float64).reshape(len(signal), -1)
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    iu = np.triu_indices(len(x), 1)
    if gamma is None:
        med = np.median(d2[iu]) if len(iu[0]) else 0.0
        gamma = 1.0 / med if med != 0 else 1.0
    k = np.clip(d2 * gamma, 1e-2, 1e2)
    g = np.exp(-k)
    np.fill_diagonal(g, 1.0)
    return g


def pelt_rbf(signal, pen, min_size=2, jump=5, gamma=None):
    """-> sorted breakpoints (segment ends, the last one is len(signal))."""
    n = len(signal)
    gram = rbf_gram(signal, gamma)
    # prefix sums: S[i, j] = sum of gram[:i, :j]  ->  block sums in O(1)
    S = np.zeros((n + 1, n + 1))
    S[1:, 1:] = gram.cumsum(0).cumsum(1)
    diag = np.concatenate([[0.0], np.cumsum(np.diagonal(gram))])

    def cost(a, b):
        block = S[b, b] - S[a, b] - S[b, a] + S[a, a]
        return (diag[b] - diag[a]) - block / (b - a)

    # partitions[t] = (total penalised cost, breakpoints) of the best segmentation of signal[:t]
    partitions = {0: (0.0, ())}
    admissible = []
    ends = [k for k in range(0, n, jump) if k >= min_size] + [n]
    for bkp in ends:
        admissible.append(int(math.floor((bkp - min_size) / jump)) * jump)
        cand = []
        for t in admissible:
            if t not in partitions or bkp - t < min_size:
                continue
            total, bk = partitions[t]
            cand.append((total + cost(t, bkp) + pen, bk + (bkp,), t))
        if not cand:
            continue
        best = min(cand, key=lambda c: c[0])
        partitions[bkp] = (best[0], best[1])
        admissible = [t for total, _, t in cand if total <= best[0] + pen]  # PELT pruning
    if n not in partitions:
        return [n]
    return sorted(partitions[n][1])
