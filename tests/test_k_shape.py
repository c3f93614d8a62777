"""GPU parity at the kernel-only stress shape K: n = 4096 nodes -- the shape bench.py's
`roofline_chain_n4096*` entries and `--workload chain` quote.  No CPU oracle finishes 4096^3 products in seconds, so the
references are evaluated in float64 ON THE GPU: torch.matmul for the chain GEMMs, and the oracle's own differentiable walk
(`oracle.walk_loss_torch`, the restatement of src/model.py:22-46 pinned by the reference goldens) run on fp64 CUDA tensors for
the walk.  What only this size exercises: the 16x16-tile XCD super-tile remap of the 256x256 bf16 kernels, 64 MB matrices
(offsets beyond 2^24 elements; the batch-3 products reach 2^25.6), the state layout of a walk whose every matrix is 64 MB."""
import numpy as np
import pytest
import torch

from oracle import crw_oracle as orc

pytestmark = pytest.mark.gpu

N4K = 4096


@pytest.fixture(scope="module")
def hip():
    import crw_hip
    crw_hip.lib()
    assert torch.cuda.is_available()
    return crw_hip


def _operands(batch, n, seed, probability_like):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if probability_like:  # what the chain multiplies: non-negative, row-scaled, plus a signed part
        A = torch.rand(batch, n, n, generator=g, device="cuda") + 0.1 * torch.randn(batch, n, n, generator=g, device="cuda")
        B = torch.rand(batch, n, n, generator=g, device="cuda") * torch.linspace(0.5, 1.5, n, device="cuda")[None, None, :]
    else:
        A = torch.randn(batch, n, n, generator=g, device="cuda")
        B = torch.randn(batch, n, n, generator=g, device="cuda")
    C0 = torch.randn(batch, n, n, generator=g, device="cuda")
    return A, B, C0


def _op(X, t):
    return X.transpose(1, 2) if t else X


@pytest.mark.parametrize("batch", [1, 3])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_f32_n4096_matches_fp64(hip, batch, ta, tb):
    """`gemm_pad_f32_kernel<128,128>` at the bench shape against torch.float64 matmul on the box."""
    A, B, C0 = _operands(batch, N4K, 4096 + 7 * ta + 3 * tb + batch, False)
    ref = _op(A, ta).double() @ _op(B, tb).double()
    out = hip.gemm_f32(A, B, transA=ta, transB=tb)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5 * N4K ** 0.5)
    acc = hip.gemm_f32(A, B, C0.clone(), transA=ta, transB=tb, beta=True)
    torch.testing.assert_close(acc.double(), ref + C0.double(), rtol=1e-5, atol=1e-5 * N4K ** 0.5)


@pytest.mark.parametrize("batch", [1, 3])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("split", [1, 3])
def test_gemm_bf16_n4096_matches_fp64(hip, batch, ta, tb, split):
    """`gemm_pad_bf16_kernel<1|3>` (256x256 tiles, 16x16 tiles per matrix -> the XCD super-tile block map) at the bench
    shape.  split 1: exact products of the bf16-rounded operands with fp32 accumulation; split 3 (hi/lo pairs): fp32-grade
    against the unrounded operands."""
    A, B, C0 = _operands(batch, N4K, 4096 + 7 * ta + 3 * tb + split + 10 * batch, True)
    out, ws = hip.gemm_bf16(A, B, transA=ta, transB=tb, split=split)
    if split == 1:
        ref = _op(A.bfloat16().double(), ta) @ _op(B.bfloat16().double(), tb)
        torch.testing.assert_close(out.double(), ref, rtol=2e-5, atol=1e-4)
    else:
        ref = _op(A.double(), ta) @ _op(B.double(), tb)
        torch.testing.assert_close(out.double(), ref, rtol=3e-5, atol=3e-5 * N4K ** 0.5)
    del ref
    acc, _ = hip.gemm_bf16(A, B, C0.clone(), transA=ta, transB=tb, beta=True, split=split, ws=ws, convert=False)
    torch.testing.assert_close(acc, out + C0, rtol=1e-6, atol=1e-3)


def _walk_reference_fp64(emb, tau):
    e64 = emb.detach().double().requires_grad_(True)
    Ats = []
    loss, A = orc.walk_loss_torch(e64, tau, At_out=Ats)
    loss.backward()
    return loss.item(), A.detach(), torch.stack(Ats, 1), e64.grad


# tolerances per chain arithmetic: (At rtol, At atol as a fraction of max|At|, loss, gradient rtol, gradient atol as a fraction of
# max|gradient|).  Measured on the box (T = 6): f32 8.5e-5 of the At bound, gradient 1.9e-5 of its maximum; bf16x3 3.7e-5 / 1.7e-5;
# bf16 1.3e-2 / 2.7e-3; |loss - fp64| 2e-10 / 7e-10 / 8e-7
_TOL = {0: (1e-4, 1e-6, 1e-4, 2e-3, 1e-3),      # CRW_CHAIN_F32: the parity bar
        2: (2e-4, 2e-6, 1e-4, 2e-3, 1e-3),      # CRW_CHAIN_BF16X3: hi/lo pairs, fp32-grade
        1: (3e-2, 2e-3, 1e-4, 5e-2, 1e-2)}      # CRW_CHAIN_BF16: plain bf16 probabilities, the throughput mode


@pytest.mark.parametrize("T,tau", [(4, 0.05), (6, 0.05), (5, 0.3)])
@pytest.mark.parametrize("chain", [0, 2, 1])
def test_walk_n4096_matches_fp64_oracle(hip, chain, T, tau):
    """One walk at [B,T,N,C] = [1,T,4096,128] (bench.py --workload chain quotes T = 32 of the same, tau 0.05) in every chain
    arithmetic: logits, every cycle product At_k, loss and dLoss/dEmb against the oracle's formulas evaluated in float64 on the
    GPU.  tau = 0.05 gives peaked transition rows (a near-permutation walk), tau = 0.3 flat ones (thousands of comparable terms
    per dot product)."""
    import model as crw_model
    C = 128
    g = torch.Generator().manual_seed(11 + T)
    base = torch.randn(1, 1, N4K, C, generator=g)
    emb = (base + 0.5 * torch.randn(1, T, N4K, C, generator=g)).cuda().requires_grad_(True)
    loss_ref, A_ref, At_ref, demb_ref = _walk_reference_fp64(emb, tau)

    A, stats = crw_model.affinity_with_stats(emb, tau)
    assert (A.detach().double() - A_ref).abs().max().item() <= 1e-4  # logits up to 1 / tau = 20
    loss, _, At = hip.walk_fwd(A.detach().contiguous(), chain=chain, want_At=True, stats=stats)
    rtol, afrac, ltol, grtol, gafrac = _TOL[chain]
    scale = At_ref.abs().amax(dim=(-1, -2), keepdim=True)
    err = (At.double() - At_ref).abs()
    worst = (err - rtol * At_ref.abs() - afrac * scale).max().item()
    assert worst <= 0, (chain, "At", worst, err.max().item(), scale.flatten().tolist())
    assert abs(loss.item() - loss_ref) <= ltol, (loss.item(), loss_ref)
    crw_model.walk_loss(A, chain, stats).backward()
    gerr = (emb.grad.double() - demb_ref).abs()
    gscale = demb_ref.abs().max()
    worst = (gerr - grtol * demb_ref.abs() - gafrac * gscale).max().item()
    assert worst <= 0, (chain, "demb", worst, gerr.max().item(), gscale.item())
    print(f"chain {chain} T {T} tau {tau}: max|dAt| {err.max().item():.3e} (max At {scale.max().item():.3e}), max rel dAt "
          f"{(err / (At_ref.abs() + afrac * scale)).max().item():.3e}, dloss {abs(loss.item() - loss_ref):.3e}, max|ddemb|/max|demb| "
          f"{(gerr.max() / gscale).item():.3e}")
    # the walk that computes its own statistics (stats kernel) takes the same state layout
    loss2, _, _ = hip.walk_fwd(A.detach().contiguous(), chain=chain)
    assert abs(loss2.item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
