"""``CRW`` -- the contrastive-random-walk model, same constructor / forward surface as the
reference (src/model.py:6-46) with the hot path running in hand-written HIP kernels.

    loss, A = CRW(encoder, tau, pos_embed)(seq)          # seq float32 [B, T, N, h, w]

forward:  patches -> encoder (PyTorch-ROCm) -> [B,T,N,C] features
          -> crw_affinity_fwd  (L2 normalise + E_t E_{t+1}^T / tau)          src/model.py:22-26
          -> crw_walk_fwd      (dual softmax + transition chain + CE loss)   src/model.py:31-46
backward: crw_walk_bwd -> crw_affinity_bwd -> encoder autograd.

There is no CPU / eager fallback: tensors must live on an MI355X and libcrw_hip.so must be built.
"""
import torch
import torch.nn as nn

import crw_hip
from utils import pos_embed


class _Affinity(torch.autograd.Function):
    """-> (A, stats): the logits and, from the same kernel's epilogue, the statistics of their two softmaxes (not differentiable:
    they only save `walk_loss` a pass over A)."""

    @staticmethod
    def forward(ctx, emb, tau, want_stats=True):
        A, ehat, norm, stats = crw_hip.affinity_fwd(emb.contiguous().float(), tau, want_stats=want_stats)
        ctx.save_for_backward(ehat, norm)
        ctx.tau = tau
        if stats is None:
            return A, None
        ctx.mark_non_differentiable(stats)
        return A, stats

    @staticmethod
    def backward(ctx, dA, _dstats=None):
        ehat, norm = ctx.saved_tensors
        return crw_hip.affinity_bwd(dA.contiguous(), ehat, norm, ctx.tau), None, None


class _WalkLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, chain, stats):
        A = A.contiguous()
        loss, state, _ = crw_hip.walk_fwd(A, chain, stats=stats)
        ctx.save_for_backward(A, state)  # freed by autograd after backward (kept under retain_graph)
        ctx.chain = chain
        return loss

    @staticmethod
    def backward(ctx, gloss):
        A, state = ctx.saved_tensors
        return crw_hip.walk_bwd(gloss, A, state, ctx.chain), None, None


def affinity(emb, tau):
    """emb [B,T,N,C] (raw encoder output) -> logits A [B,T-1,N,N]; differentiable."""
    return _Affinity.apply(emb, float(tau), False)[0]  # no statistics: neither their epilogue work nor their workspace


def affinity_with_stats(emb, tau):
    """-> (A, stats [4,B,T-1,N]): pass both to `walk_loss` and the walk skips its statistics pass over A."""
    return _Affinity.apply(emb, float(tau))


def walk_loss(A, stats_or_chain=None, stats=None, chain=None):
    """A [B,T-1,N,N] -> cycle-consistency loss (0-d); differentiable.  Accepts walk_loss(A), walk_loss(A, chain),
    walk_loss(A, chain, stats) and walk_loss(*affinity_with_stats(emb, tau)) (= walk_loss(A, stats))."""
    if torch.is_tensor(stats_or_chain):
        stats, stats_or_chain = stats_or_chain, None
    if chain is None:
        chain = crw_hip.CHAIN_F32 if stats_or_chain is None else stats_or_chain
    return _WalkLoss.apply(A, chain, stats)


class CRW(nn.Module):
    def __init__(self, encoder, tau, pos_embed, only_a=False):
        super().__init__()
        self.encoder = encoder
        self.tau = tau
        self.pos_embed = pos_embed
        self.only_a = only_a
        self.chain = crw_hip.CHAIN_F32  # CRW_CHAIN_F32 (parity) | CRW_CHAIN_BF16

    def forward(self, seq):
        B, T, N, H, W = seq.shape
        x = seq.reshape(-1, H, W).unsqueeze(1)  # [B*T*N, 1, H, W]
        if self.pos_embed:
            x = pos_embed(x)
        emb = self.encoder(x).reshape(B, T, N, -1)
        A, stats = affinity_with_stats(emb, self.tau)
        if self.only_a:
            return A
        return walk_loss(A, self.chain, stats), A

    def forward_columns(self, cols, length, stride=1):
        """Shared-encoder training step over OVERLAPPING items (opt-in, not in
        the reference).  ``cols`` float32 [B, Tc, N, h, w] are consecutive patch-columns of B radargrams
        (``RGDataset.columns``); the items are the windows of ``length`` columns starting every ``stride``
        columns.  Every patch-column is encoded ONCE and every column-to-column affinity is built once; only
        the walks run per item.  Returns ``(loss, A_all[B, Tc-1, N, N])`` where ``loss`` (and, through
        autograd, the encoder gradient) equals what ``forward`` gives on the batch of all those items --
        the reference semantics for that batch, with up to ``length`` times less encoder work."""
        B, Tc, N, H, W = cols.shape
        if not 3 <= length <= Tc or stride < 1:
            raise ValueError(f"need 3 <= length <= {Tc} columns and stride >= 1")
        x = cols.reshape(-1, H, W).unsqueeze(1)
        if self.pos_embed:
            x = pos_embed(x)
        emb = self.encoder(x).reshape(B, Tc, N, -1)
        A_all = affinity(emb, self.tau)                           # [B, Tc-1, N, N]
        win = A_all.unfold(1, length - 1, stride)                 # [B, S, N, N, length-1] (view)
        A_items = win.permute(0, 1, 4, 2, 3).reshape(-1, length - 1, N, N)  # materialised per-item logits
        if self.only_a:
            return A_all
        return walk_loss(A_items, self.chain), A_all
