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
    @staticmethod
    def forward(ctx, emb, tau):
        A, ehat, norm = crw_hip.affinity_fwd(emb.contiguous().float(), tau)
        ctx.save_for_backward(ehat, norm)
        ctx.tau = tau
        return A

    @staticmethod
    def backward(ctx, dA):
        ehat, norm = ctx.saved_tensors
        return crw_hip.affinity_bwd(dA.contiguous(), ehat, norm, ctx.tau), None


class _WalkLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, chain):
        B, Tm1, N, _ = A.shape
        loss, state, _ = crw_hip.walk_fwd(A.contiguous(), chain)
        ctx.state, ctx.dims, ctx.chain = state, (B, Tm1 + 1, N), chain
        return loss

    @staticmethod
    def backward(ctx, gloss):
        B, T, N = ctx.dims
        dA = crw_hip.walk_bwd(gloss, ctx.state, B, T, N, ctx.chain)
        ctx.state = None
        return dA, None


def affinity(emb, tau):
    """emb [B,T,N,C] (raw encoder output) -> logits A [B,T-1,N,N]; differentiable."""
    return _Affinity.apply(emb, float(tau))


def walk_loss(A, chain=crw_hip.CHAIN_F32):
    """A [B,T-1,N,N] -> cycle-consistency loss (0-d); differentiable."""
    return _WalkLoss.apply(A, chain)


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
        A = affinity(emb, self.tau)
        if self.only_a:
            return A
        return walk_loss(A, self.chain), A
