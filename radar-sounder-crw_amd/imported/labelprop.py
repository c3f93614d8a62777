"""k-NN label propagation (``LabelPropVOS_CRW``) -- same ``cfg`` keys and ``predict`` signature as
the reference (src/imported/labelprop.py:42-115, which in turn follows videowalk, Jabri et al.
2020), computed by the HIP kernels ``crw_labelprop_topk`` / ``crw_labelprop_gather``.

Two entry points:
  * ``predict(feats, masks, curr_feat)`` -- the reference's frame-at-a-time interface;
  * ``propagate_all(feats, seed, nclasses)`` -- whole radargram in two launches (what
    ``utils.propagate`` uses): affinities/top-k of every frame at once, then one sequential
    gather kernel.  Both produce identical label maps.
"""
import torch

import crw_hip

MASK_NEG = -1e10


class LabelPropVOS_CRW(object):
    def __init__(self, cfg):
        self.cxt_size = cfg['CXT_SIZE']
        self.radius = cfg['RADIUS']
        self.temperature = cfg['TEMP']
        self.topk = cfg['KNN']
        self.mask = None
        self.mask_hw = None

    # context bookkeeping of the videowalk interface (kept for API compatibility)
    def context_long(self, t0, t):
        return [t0]

    def context_short(self, t0, t):
        return [max(tt, t0) for tt in range(t - self.cxt_size, t)]

    def context_index(self, t0, t):
        return self.context_long(t0, t) + self.context_short(t0, t)

    def _band(self, h, w, dev):
        """additive locality mask [1, h*w, h*w]: 0 inside the radius, -1e10 outside.  Kept as an
        attribute like the reference; the kernels apply the same band arithmetically."""
        if self.mask is None or self.mask_hw != (h, w):
            i = torch.arange(h, device=dev).repeat_interleave(w).float()
            j = torch.arange(w, device=dev).repeat(h).float()
            d = ((i[:, None] - i[None, :]) ** 2 + (j[:, None] - j[None, :]) ** 2).sqrt()
            self.mask = torch.where(d < self.radius, 0.0, MASK_NEG)[None]
            self.mask_hw = (h, w)
        return self.mask

    def _check_grid(self, h, w):
        if self.topk > h * w:
            raise RuntimeError(f"KNN={self.topk} exceeds the number of nodes per frame ({h * w}); "
                               "torch.topk in the reference raises for the first frame as well")

    def predict(self, feats, masks, curr_feat, ref_index=None, t=None):
        """feats: list of n [1,C,h,w] context features; masks: list of n [1,M,h,w] soft labels;
        curr_feat [1,C,h,w]  ->  soft labels of the current frame [1,M,h,w].  A radargram's frames are columns of patches
        (w = 1, what `utils.propagate` passes); any h x w grid is taken like the reference's (nodes in row-major order, the band
        the Euclidean disc of `MaskedAttention`, src/imported/maskedatt.py:222-245)."""
        h, w = curr_feat.shape[-2:]
        self._check_grid(h, w)
        self._band(h, w, curr_feat.device)
        n, N = len(feats), h * w
        E = torch.cat(list(feats) + [curr_feat], 0).flatten(2).permute(0, 2, 1).contiguous().float()  # [n+1, N, C]
        M = masks[0].shape[1]
        L = torch.empty((n + 1) * N, M, device=E.device, dtype=torch.float32)
        L[:n * N] = torch.cat(list(masks), 0).flatten(2).permute(0, 2, 1).reshape(n * N, M)
        Wt, It = crw_hip.labelprop_topk(E, self.cxt_size, self.radius, self.temperature, self.topk, first_frame=n, grid_w=w)
        crw_hip.labelprop_gather(None, Wt, It, n + 1, N, M, first_frame=n, L=L, cxt_size=self.cxt_size)
        return L[n * N:].reshape(N, M).t().reshape(1, M, h, w)

    def propagate_all(self, feats, seed, nclasses, grid_w=1):
        """feats [T,N,C] (normalised features), seed [N] float class ids of frame 0
        -> (pred [N,T] float class ids, L [T*N, M] soft labels).  grid_w: the N nodes are an (N / grid_w) x grid_w grid."""
        T, N, C = feats.shape
        self._check_grid(N // grid_w, grid_w)
        self._band(N // grid_w, grid_w, feats.device)
        Wt, It = crw_hip.labelprop_topk(feats, self.cxt_size, self.radius, self.temperature, self.topk, first_frame=1, grid_w=grid_w)
        L, pred = crw_hip.labelprop_gather(seed.float().contiguous(), Wt, It, T, N, nclasses, first_frame=1, cxt_size=self.cxt_size)
        return pred, L
