"""GPU parity tests: the HIP path (through the C ABI / ctypes) against the golden vectors of the
reference and against the CPU oracle on seeded inputs.  Tolerance: north_star asks for 1e-4 fp32
on the loss and propagated label map; label maps are compared exactly."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import crw_oracle as orc

pytestmark = pytest.mark.gpu

WALK_CASES = ["walk_cfg1_B2T8N7", "walk_odd_B1T4N5", "walk_onecycle_B3T3N6", "walk_cfg2_B1T16N63",
              "walk_N70_B2T6", "walk_cfg3_B1T32N63", "walk_noise_B2T8N7_tau0p1"]


@pytest.fixture(scope="module")
def hip():
    import crw_hip
    crw_hip.lib()
    assert torch.cuda.is_available()
    return crw_hip


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("n,batch", [(32, 3), (64, 2), (96, 1), (128, 2), (192, 1), (256, 9), (512, 2), (1152, 1)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_f32_matches_torch(hip, n, batch, ta, tb):
    g = torch.Generator().manual_seed(n + 7 * ta + 3 * tb)
    A = torch.randn(batch, n, n, generator=g).cuda()
    B = torch.randn(batch, n, n, generator=g).cuda()
    C0 = torch.randn(batch, n, n, generator=g).cuda()
    ref = (A.transpose(1, 2) if ta else A).double() @ (B.transpose(1, 2) if tb else B).double()
    out = hip.gemm_f32(A, B, transA=ta, transB=tb)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5 * n ** 0.5)
    acc = hip.gemm_f32(A, B, C0.clone(), transA=ta, transB=tb, beta=True)
    torch.testing.assert_close(acc.double(), ref + C0.double(), rtol=1e-5, atol=1e-5 * n ** 0.5)


# (1024, 16): (n/256)^2 * batch = 256 -> the 256x256-tile kernels (8 waves; BK = 32 for hi/lo pairs) in every operand layout;
# (512, 64) the same kernels with 2x2 tiles per matrix (the non-super-tile block map: tiles % 8 != 0)
@pytest.mark.parametrize("n,batch", [(128, 2), (256, 1), (384, 2), (1024, 16), (512, 64)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("split", [1, 3])
def test_gemm_bf16_matches_torch(hip, n, batch, ta, tb, split):
    g = torch.Generator().manual_seed(n + 7 * ta + 3 * tb + split)
    # asymmetric, non-negative-biased operands (probability-like) + a signed part
    A = (torch.rand(batch, n, n, generator=g) + 0.1 * torch.randn(batch, n, n, generator=g)).cuda()
    B = (torch.rand(batch, n, n, generator=g) * torch.linspace(0.5, 1.5, n)[None, None, :]).cuda()
    opA = lambda X: X.transpose(1, 2) if ta else X
    opB = lambda X: X.transpose(1, 2) if tb else X
    out, ws = hip.gemm_bf16(A, B, transA=ta, transB=tb, split=split)
    if split == 1:  # exact products of the bf16-rounded operands, fp32 accumulation
        ref = opA(A.bfloat16().double()) @ opB(B.bfloat16().double())
        torch.testing.assert_close(out.double(), ref, rtol=2e-5, atol=1e-4)
    else:
        ref = opA(A.double()) @ opB(B.double())
        torch.testing.assert_close(out.double(), ref, rtol=3e-5, atol=3e-5 * n ** 0.5)
    C0 = torch.randn(batch, n, n, generator=g).cuda()
    acc, _ = hip.gemm_bf16(A, B, C0.clone(), transA=ta, transB=tb, beta=True, split=split, ws=ws, convert=False)
    torch.testing.assert_close(acc, out + C0, rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("name", WALK_CASES)
def test_training_path_matches_reference(hip, name):
    import model as crw_model
    g = load_golden(name)
    tau = float(g["tau"])
    emb = dev(g["emb"]).requires_grad_(True)
    A = crw_model.affinity(emb, tau)
    np.testing.assert_allclose(A.detach().cpu().numpy(), g["A"], rtol=1e-4, atol=1e-4 / tau * 1e-2)
    loss, state, At = hip.walk_fwd(A.detach().contiguous(), want_At=True)
    np.testing.assert_allclose(At.cpu().numpy(), g["At"], rtol=1e-4, atol=1e-6)
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    loss2 = crw_model.walk_loss(A)
    assert loss2.item() == loss.item()  # deterministic
    loss2.backward()
    scale = np.abs(g["demb"]).max()
    np.testing.assert_allclose(emb.grad.cpu().numpy(), g["demb"], rtol=2e-3, atol=2e-4 * scale)


@pytest.mark.parametrize("name", ["walk_cfg1_B2T8N7", "walk_cfg2_B1T16N63", "walk_N70_B2T6", "walk_noise_B2T8N7_tau0p1",
                                  "walk_cfg3_B1T32N63"])
@pytest.mark.parametrize("chain", [1, 2])
def test_bf16_chain_modes_match_reference(hip, name, chain):
    """CRW_CHAIN_BF16X3 (hi/lo operand pairs) keeps the fp32 parity bar; CRW_CHAIN_BF16 (plain bf16
    probabilities) is the throughput mode and is held to a looser, stated tolerance."""
    import model as crw_model
    g = load_golden(name)
    tau = float(g["tau"])
    emb = dev(g["emb"]).requires_grad_(True)
    A = crw_model.affinity(emb, tau)
    loss, state, At = hip.walk_fwd(A.detach().contiguous(), chain=chain, want_At=True)
    if chain == 2:
        np.testing.assert_allclose(At.cpu().numpy(), g["At"], rtol=2e-4, atol=2e-6)
        assert abs(loss.item() - float(g["loss"])) <= 1e-4
    else:
        np.testing.assert_allclose(At.cpu().numpy(), g["At"], rtol=3e-2, atol=2e-3)
        assert abs(loss.item() - float(g["loss"])) <= 5e-3
    crw_model.walk_loss(A, chain).backward()
    scale = np.abs(g["demb"]).max()
    tol = 2e-3 if chain == 2 else 6e-2
    np.testing.assert_allclose(emb.grad.cpu().numpy(), g["demb"], rtol=10 * tol, atol=tol * scale)


@pytest.mark.parametrize("chain", [1, 2])
def test_bf16_chain_modes_on_256_tiles_match_oracle(hip, chain):
    """B = 8, T = 10, N = 500 -> Np = 512 and K*B = 64: the batched cycle products ((512/256)^2 * 64 = 256 workgroups) and
    the k-local backward group (512) dispatch the 256x256-tile bf16 kernels (`gemm_bf16.hip: launch_gemm_group_bf16`), the
    sequential recurrences (batch 8) stay on 128x128 tiles -- both against the fp64 oracle with the tolerances of the
    golden-vector cases above."""
    import model as crw_model
    B, T, N, C, tau = 8, 10, 500, 32, 0.07
    g = torch.Generator().manual_seed(500 + chain)
    base = torch.randn(1, 1, N, C, generator=g)
    emb_cpu = (base + 0.5 * torch.randn(B, T, N, C, generator=g)).float()
    o = orc.crw_from_features(emb_cpu.numpy(), tau, np.float64)
    emb = emb_cpu.cuda().requires_grad_(True)
    A = crw_model.affinity(emb, tau)
    loss, _, At = hip.walk_fwd(A.detach().contiguous(), chain=chain, want_At=True)
    if chain == 2:
        np.testing.assert_allclose(At.cpu().numpy(), o["At"], rtol=2e-4, atol=2e-6)
        assert abs(loss.item() - float(o["loss"])) <= 1e-4
    else:
        np.testing.assert_allclose(At.cpu().numpy(), o["At"], rtol=3e-2, atol=2e-3)
        assert abs(loss.item() - float(o["loss"])) <= 5e-3
    crw_model.walk_loss(A, chain).backward()
    scale = np.abs(o["demb"]).max()
    tol = 2e-3 if chain == 2 else 6e-2
    np.testing.assert_allclose(emb.grad.cpu().numpy(), o["demb"], rtol=10 * tol, atol=tol * scale)


@pytest.mark.parametrize("B,T,N,C,tau", [(2, 4, 63, 128, 0.01), (1, 3, 130, 64, 0.05), (2, 3, 257, 32, 0.1), (1, 5, 512, 128, 0.02),
                                         (1, 3, 40, 20, 0.07), (1, 3, 300, 128, 0.01), (2, 3, 260, 64, 0.05), (1, 260, 300, 128, 0.05)])
def test_affinity_tiles_and_fused_statistics(hip, B, T, N, C, tau):
    """crw_affinity_fwd on the 128-row tiles -- fp32 MFMA below 256 nodes, three-term bf16 splits on the bf16 matrix cores from
    256 nodes on (C % 32 == 0): logits against fp64, the softmax statistics of its epilogue (per-tile partials merged in tile
    order) against a direct computation, and the walk fed with them against the walk that computes its own (both softmax
    paths: imported statistics / stats kernel).  Also the tiled affinity backward (N % 4 == 0: fp32 MFMA at C in {32, 64}, or
    below 256 nodes; bf16 splits at C = 128 from 256 nodes on, here with a ragged last k-chunk at N = 300 and with more
    workgroups than the chip holds at once: T = 260) and the
    bounds-checked fallback against fp64."""
    g = torch.Generator().manual_seed(N + C)
    emb = (torch.randn(1, 1, N, C, generator=g) + 0.6 * torch.randn(B, T, N, C, generator=g)).float()
    A, ehat, norm, stats = hip.affinity_fwd(emb.cuda(), tau)
    eh = emb.double() / emb.double().norm(dim=-1, keepdim=True).clamp_min(1e-12)
    A_ref = torch.einsum("btnc,btmc->btnm", eh[:, :-1], eh[:, 1:]) / tau
    torch.testing.assert_close(A.cpu().double(), A_ref, rtol=1e-5, atol=2e-5 / tau * 1e-1)
    Ad = A.double()
    rmax, cmax = Ad.max(-1).values, Ad.max(-2).values
    rsum, csum = (Ad - rmax[..., None]).exp().sum(-1), (Ad - cmax[..., None, :]).exp().sum(-2)
    for got, want in zip(stats, (rmax, rsum, cmax, csum)):
        torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-6)
    if T <= 16:  # (the T = 260 case is there for the backward's grid; a 258-product chain amplifies the statistics' last bits)
        l0, _, At0 = hip.walk_fwd(A, want_At=True)
        l1, _, At1 = hip.walk_fwd(A, want_At=True, stats=stats)
        assert abs(l0.item() - l1.item()) <= 1e-6 * max(1.0, abs(l0.item()))
        torch.testing.assert_close(At0, At1, rtol=1e-5, atol=1e-7)
    dA = torch.randn(B, T - 1, N, N, generator=g).float()
    demb = hip.affinity_bwd(dA.cuda(), ehat, norm, tau)
    deh = torch.zeros_like(eh)
    deh[:, :-1] += torch.einsum("btnm,btmc->btnc", dA.double(), eh[:, 1:]) / tau
    deh[:, 1:] += torch.einsum("btnm,btnc->btmc", dA.double(), eh[:, :-1]) / tau
    nrm = emb.double().norm(dim=-1, keepdim=True).clamp_min(1e-12)
    want = (deh - eh * (eh * deh).sum(-1, keepdim=True)) / nrm
    torch.testing.assert_close(demb.cpu().double(), want, rtol=1e-4, atol=1e-4 * want.abs().max().item())


@pytest.mark.parametrize("N,C,tau", [(300, 128, 0.01), (512, 128, 0.01), (256, 64, 0.05)])
def test_affinity_both_arithmetics_vs_fp64(hip, N, C, tau, monkeypatch):
    """From 256 nodes on the affinity build runs on three-term bf16 splits (csrc/gemm_f32.hip `affinity_on_bf16`) unless
    CRW_AFFINITY_F32=1 forces the fp32 MFMA: both paths against fp64 at tau = 0.01, where 1 / tau amplifies the error of the
    logits that feed the softmaxes -- logits within 2e-5, the walk's loss within 1e-6 of each other."""
    g = torch.Generator().manual_seed(N)
    emb = (torch.randn(1, 1, N, C, generator=g) + 0.6 * torch.randn(1, 4, N, C, generator=g)).float().cuda()
    eh = emb.double() / emb.double().norm(dim=-1, keepdim=True).clamp_min(1e-12)
    A_ref = (torch.einsum("btnc,btmc->btnm", eh[:, :-1], eh[:, 1:]) / tau)
    dA = torch.randn(A_ref.shape, generator=g).float().cuda()
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CRW_AFFINITY_F32", mode)
        A, ehat, norm, stats = hip.affinity_fwd(emb, tau)
        err = (A.double() - A_ref).abs().max().item()
        assert err <= 1e-4, (mode, err)  # logits up to 1 / tau = 100: 1e-6 of that
        loss, _, _ = hip.walk_fwd(A, stats=stats)
        out[mode] = (A, loss.item(), hip.affinity_bwd(dA, ehat, norm, tau), err)
    assert not torch.equal(out["0"][0], out["1"][0]), "the two arithmetics should differ in the last bits (is the switch live?)"
    assert abs(out["0"][1] - out["1"][1]) <= 1e-6 * max(1.0, abs(out["1"][1]))
    torch.testing.assert_close(out["0"][2], out["1"][2], rtol=1e-4, atol=1e-4 * out["1"][2].abs().max().item())


def test_no_cycle_T2(hip):
    import model as crw_model
    g = load_golden("walk_T2_nocycle")
    emb = dev(g["emb"]).requires_grad_(True)
    A = crw_model.affinity(emb, float(g["tau"]))
    np.testing.assert_allclose(A.detach().cpu().numpy(), g["A"], rtol=1e-4, atol=1e-4)
    loss = crw_model.walk_loss(A)
    assert loss.item() == 0.0
    loss.backward()
    assert torch.count_nonzero(emb.grad).item() == 0


@pytest.mark.parametrize("B,T,N,C,tau", [(2, 5, 100, 64, 0.05), (1, 4, 130, 32, 0.1), (1, 3, 200, 128, 0.02),
                                         (3, 6, 33, 20, 0.07), (1, 5, 257, 16, 0.2), (1, 4, 1100, 16, 0.3),
                                         (5, 9, 32, 8, 0.1), (2, 4, 64, 12, 0.05)])
def test_training_path_matches_oracle_seeded(hip, B, T, N, C, tau):
    import model as crw_model
    g = torch.Generator().manual_seed(B * 1000 + N)
    base = torch.randn(1, 1, N, C, generator=g)
    emb_cpu = (base + 0.4 * torch.randn(B, T, N, C, generator=g)).float()
    o = orc.crw_from_features(emb_cpu.numpy(), tau, np.float64)
    emb = emb_cpu.cuda().requires_grad_(True)
    A = crw_model.affinity(emb, tau)
    np.testing.assert_allclose(A.detach().cpu().numpy(), o["A"], rtol=1e-4, atol=1e-4)
    loss, _, At = hip.walk_fwd(A.detach().contiguous(), want_At=True)
    np.testing.assert_allclose(At.cpu().numpy(), o["At"], rtol=1e-4, atol=1e-6)
    assert abs(loss.item() - float(o["loss"])) <= 1e-4
    crw_model.walk_loss(A).backward()
    # gradients here are ~1e-7 (near-uniform walk): fp32 cancellation vs the fp64 oracle
    np.testing.assert_allclose(emb.grad.cpu().numpy(), o["demb"], rtol=2e-3, atol=max(1e-3 * np.abs(o["demb"]).max(), 1e-12))


def test_walk_backward_dA_matches_oracle(hip):
    g = torch.Generator().manual_seed(5)
    A_cpu = (torch.randn(2, 6, 40, 40, generator=g) * 3).float()
    dA_ref = orc.walk_backward(A_cpu.double().numpy(), gloss=0.7)
    A = A_cpu.cuda()
    loss, state, _ = hip.walk_fwd(A)
    dA = hip.walk_bwd(torch.tensor(0.7).cuda(), A, state)
    np.testing.assert_allclose(dA.cpu().numpy(), dA_ref, rtol=1e-3, atol=1e-4 * np.abs(dA_ref).max())
    # A_{T-2} never enters the loss
    assert torch.count_nonzero(dA[:, -1]).item() == 0


@pytest.mark.parametrize("name,wname", [("cnn_cfg1_B2T8N7", "cnn_weights_seed11"),
                                        ("cnn_posembed_B1T4N3", "cnn_weights_posembed_seed21")])
@pytest.mark.parametrize("convs", ["bf16x3", None])
def test_full_model_matches_reference(hip, name, wname, convs):
    """convs = "bf16x3": conv3-5 fwd+bwd on the hand-written HIP kernels (default product path);
    None: the same model on PyTorch-ROCm convolutions."""
    import model as crw_model
    import encoder as crw_encoder
    g, w = load_golden(name), load_golden(wname)
    torch.backends.cudnn.allow_tf32 = False
    enc = crw_encoder.CNN(bool(g["pos_embed"]))
    enc.hip_convs = convs
    enc.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    net = crw_model.CRW(enc, float(g["tau"]), bool(g["pos_embed"])).cuda()
    loss, A = net(dev(g["seq"]))
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    np.testing.assert_allclose(A.detach().cpu().numpy(), g["A"], rtol=1e-3, atol=5e-3)
    loss.backward()
    for k, p in enc.named_parameters():
        ref = g["grad." + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-2, atol=2e-3 * np.abs(ref).max())
    assert crw_model.CRW(enc, 0.01, bool(g["pos_embed"]), only_a=True).cuda()(dev(g["seq"])).shape == A.shape


def test_training_trajectory_matches_oracle(hip):
    """Six Adam steps (the reference's optimizer and learning rate, scripts/train.py:55,26) of the product path -- HIP
    encoder + walk, flat gradient bucket, fused Adam exactly as train.py / bench.py run them -- against the same six
    steps of the fp64 oracle from the same golden weights and batch.  Trajectories diverge chaotically (the oracle in
    fp32 is 3e-6 / 5e-5 off its fp64 self at steps 5 / 6), hence the widening tolerance; the weight UPDATES must agree
    to a few per cent in L2 (Adam's m/sqrt(v) amplifies noise on near-zero gradients)."""
    import model as crw_model
    import encoder as crw_encoder
    import dist as crw_dist
    g, w = load_golden("cnn_cfg1_B2T8N7"), load_golden("cnn_weights_seed11")
    steps, lr, tau = 6, 1e-3, float(g["tau"])
    sd = {k: torch.tensor(v).double().requires_grad_(True) for k, v in w.items()}
    opt64 = torch.optim.Adam(list(sd.values()), lr=lr)
    seq64 = torch.tensor(g["seq"]).double()
    want = []
    for _ in range(steps):
        opt64.zero_grad()
        loss, _, _ = orc.crw_forward_torch(seq64, sd, tau)
        loss.backward()
        opt64.step()
        want.append(loss.item())

    enc = crw_encoder.CNN(False)
    enc.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    net = crw_model.CRW(enc, tau, False).cuda()
    net.train(True)
    import optim as crw_optim
    bucket = crw_dist.FlatGradBucket(net.parameters(), lazy=True)
    opt = crw_optim.FlatAdam(bucket, lr=lr)  # what train.py / bench.py use (crw_adam_step on the flat parameter buffer)
    seq = dev(g["seq"])
    got = []
    for _ in range(steps):
        bucket.zero()
        loss, _ = net(seq)
        loss.backward()
        bucket.all_reduce_mean()
        opt.step()
        got.append(loss.item())
    assert want[-1] < 0.8 * want[0]  # the steps do train
    for i, (a, b) in enumerate(zip(got, want)):
        assert abs(a - b) <= (1e-4 if i < 4 else 1e-3) * abs(b), (i, got, want)
    for k, p_ in enc.named_parameters():
        upd_ref = sd[k].detach() - torch.tensor(w[k]).double()
        upd = p_.detach().cpu().double() - torch.tensor(w[k]).double()
        assert (upd - upd_ref).norm() <= 5e-2 * upd_ref.norm(), (k, float((upd - upd_ref).norm() / upd_ref.norm()))


@pytest.mark.parametrize("hw", [(32, 32), (20, 27), (16, 16), (9, 40)])
@pytest.mark.parametrize("split", [3, 1])
def test_encoder_inference_trunk_any_patch_size(hip, hw, split):
    """CNN inference at patch sizes other than 16x16 (BASELINE config 5 uses 32x32): conv3-5 + ReLU + global
    average pool on the tiled HIP kernels (10x10 output tiles incl. partial tiles) against the same network in
    fp64 on the CPU; also the map conv kernel itself (planes) on an odd-sized map."""
    import encoder as crw_encoder
    torch.manual_seed(7)
    enc = crw_encoder.CNN(False)
    enc.hip_convs = "bf16x3" if split == 3 else "bf16"
    x = torch.randn(6, 1, *hw)
    enc64 = crw_encoder.CNN(False).double()
    enc64.load_state_dict({k: v.double() for k, v in enc.state_dict().items()})
    enc64.hip_convs = None
    with torch.no_grad():
        want = enc64(x.double())
        got = enc.cuda()(x.cuda())
    tol = dict(rtol=2e-4, atol=2e-5) if split == 3 else dict(rtol=5e-2, atol=5e-3)
    torch.testing.assert_close(got.cpu().double(), want, **tol)
    if hw == (20, 27) and split == 3:  # planes of one layer on a 14x21 map (2 x 3 tiles, partial in both directions)
        import torch.nn.functional as TF
        g = torch.Generator().manual_seed(3)
        xm = torch.randn(3, 32, 14, 21, generator=g)
        w = torch.randn(64, 32, 3, 3, generator=g) * 0.08
        b = torch.randn(64, generator=g) * 0.1
        fh, fl, _, _ = hip.enc_pack_weights(w.cuda(), 3)
        xh, xl = hip.enc_pack_input_map(xm.cuda(), 3)
        yh, yl, gap = hip.enc_conv3x3_map(3, xh, xl, fh, fl, 64, 14, 21, bias=b.cuda(), gap=False)  # (mode 0)
        xq = (xh.float() + xl.float()).cpu().double().reshape(3, 14, 21, 32).permute(0, 3, 1, 2)
        y_ref = TF.relu(TF.conv2d(xq, w.double(), b.double(), padding=1)).permute(0, 2, 3, 1).reshape(3, 14 * 21, 64)
        torch.testing.assert_close((yh.float() + yl.float()).cpu().double(), y_ref, rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("hw", [(96, 96), (12, 706), (100, 73)])
def test_map_convolutions_beyond_64_tiles(hip, hw):
    """Feature maps with more than 64 tiles of a class (a map launch names at most 64 tiles in its kernel arguments; larger maps
    go out in chunks): 96x96 patches = 81 full tiles, a 12x706 strip = 70 small edge tiles (6x10 pixels), 100x73 = 10x7 tiles with
    7 + 10 - 1 partial ones on the edges -- inference and a training step (forward, backward-data, weight gradients, front end) of
    the whole CNN against the same network in fp64; the reference's convolutions take any patch size (src/encoder.py:13-32)."""
    import warnings
    import encoder as crw_encoder
    torch.manual_seed(7)
    enc = crw_encoder.CNN(False)
    enc64 = crw_encoder.CNN(False).double()
    enc64.load_state_dict({k: v.double() for k, v in enc.state_dict().items()})
    enc64.hip_convs = None
    g = torch.Generator().manual_seed(hw[0])
    x = torch.randn(2, 1, *hw, generator=g)
    gy = torch.randn(2, 128, generator=g)
    want = enc64(x.double())
    want.backward(gy.double())
    enc = enc.cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the PyTorch-op fallback would warn: it must not be taken
        with torch.no_grad():
            got_inf = enc(x.cuda())
        got = enc(x.cuda())
        got.backward(gy.cuda())
    torch.testing.assert_close(got_inf.cpu().double(), want.detach(), rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(got.detach().cpu().double(), want.detach(), rtol=2e-4, atol=2e-5)
    for (k, p_), (_, q) in zip(enc.named_parameters(), enc64.named_parameters()):
        ref = q.grad
        torch.testing.assert_close(p_.grad.cpu().double(), ref, rtol=2e-2, atol=2e-3 * ref.abs().max().item(), msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("hw,ov", [((32, 32), (24, 0)), ((20, 27), (10, 0))])
def test_training_at_other_patch_sizes_vs_oracle(hip, hw, ov):
    """Training step at patch sizes other than 16x16 (BASELINE config 5 is a 32x32-patch model, overlap (24,0),
    scripts/test/test_mc1.py:19-26): the whole conv trunk forward and backward on the tiled HIP kernels (`_HipMapEncoder`:
    26x26 and 14x21 feature maps = 3x3 / 2x3 tiles with partial tiles in both directions; the front end's tiled backward
    `crw_enc_front_bwd_map` included), HIP affinity + walk -- loss and every parameter gradient against the fp64 oracle,
    tolerances of test_full_model_matches_reference."""
    import warnings
    import model as crw_model
    import encoder as crw_encoder
    import dataset as crw_dataset
    T = 4
    h, w = hw
    ds = crw_dataset.RGDataset.synthetic(ov[0] + 5 * (h - ov[0]), 2 * T * w, T, hw, ov, seed=13)
    seq = torch.stack([ds[0], ds[T]])                          # [2, T, 5, h, w]
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    sd = {k: v.detach().clone().double().requires_grad_(True) for k, v in enc.state_dict().items()}
    loss_ref, _, _ = orc.crw_forward_torch(seq.double(), sd, 0.05)
    loss_ref.backward()
    net = crw_model.CRW(enc, 0.05, False).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                        # the PyTorch-op fallback would warn: it must not be taken
        loss, A = net(seq.cuda())
        loss.backward()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    for k, p_ in enc.named_parameters():
        ref = sd[k].grad.float().numpy()
        np.testing.assert_allclose(p_.grad.cpu().numpy(), ref, rtol=2e-2, atol=2e-3 * np.abs(ref).max(), err_msg=k)


@pytest.mark.parametrize("split", [3, 1])
def test_map_backward_kernels_match_torch(hip, split):
    """backward-data (with and without ReLU mask, fp32 copy) and weight / bias gradient of one 3x3 layer on a 14x21 map
    (2 x 3 tiles, partial in both directions) against fp64 torch, and the generic ReLU + average-pool backward."""
    import torch.nn.functional as TF
    P, cin, cout, H, W = 3, 64, 128, 14, 21
    g = torch.Generator().manual_seed(40 + split)
    hl = lambda t: (t.bfloat16(), (t - t.bfloat16().float()).bfloat16() if split == 3 else None)
    val = lambda h_, l_: (h_.float() + (l_.float() if l_ is not None else 0)).cpu().double()
    x = (torch.randn(P, H * W, cin, generator=g) * 0.5).cuda()
    dy = (torch.randn(P, H * W, cout, generator=g) * 0.5).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    fh, fl, bh, bl = hip.enc_pack_weights(w, split)
    (xh, xl), (dh, dl) = hl(x), hl(dy)
    nchw = lambda t, c: t.reshape(P, H, W, c).permute(0, 3, 1, 2)
    xq, dq = nchw(val(xh, xl), cin), nchw(val(dh, dl), cout)
    wq = w.bfloat16().float().cpu().double() if split == 1 else w.cpu().double()
    tol = dict(rtol=5e-5, atol=5e-5) if split == 3 else dict(rtol=3e-2, atol=3e-2)
    dx_ref = TF.conv_transpose2d(dq, wq, padding=1)                       # [P, cin, H, W]
    mask = (torch.rand(P, H * W, cin, generator=g) > 0.4).float().cuda().bfloat16()
    mh_, ml_, mf = hip.enc_conv3x3_map(split, dh, dl, bh, bl, cin, H, W, mode=1, mask=mask, f32=True)
    full = dx_ref.permute(0, 2, 3, 1).reshape(P, H * W, cin)
    torch.testing.assert_close(mf.cpu().double(), full, **tol)                      # the fp32 copy is the unmasked gradient
    torch.testing.assert_close(val(mh_, ml_), full * mask.float().cpu().double(), **tol)   # the planes carry the ReLU mask
    _, _, uf = hip.enc_conv3x3_map(split, dh, dl, bh, bl, cin, H, W, mode=1, planes=False, f32=True)
    torch.testing.assert_close(uf.cpu().double(), dx_ref.permute(0, 2, 3, 1).reshape(P, H * W, cin), **tol)
    dw, db = hip.enc_wgrad_map(split, dh, dl, xh, xl, H, W)
    dw_ref = torch.nn.grad.conv2d_weight(xq, w.shape, dq, padding=1)
    wtol = dict(rtol=1e-4, atol=1e-4 * dw_ref.abs().max().item()) if split == 3 else dict(rtol=5e-2, atol=2e-2 * dw_ref.abs().max().item())
    torch.testing.assert_close(dw.cpu().double(), dw_ref, **wtol)
    torch.testing.assert_close(db.cpu().double(), dq.sum((0, 2, 3)), **wtol)
    dgap = torch.randn(P, cin, generator=g).cuda()
    gh, gl = hip.enc_gap_bwd(dgap, xh, split)
    want_g = (dgap.cpu().double()[:, None, :] / (H * W)) * (val(xh, None) != 0)
    torch.testing.assert_close(val(gh, gl), want_g, rtol=1e-5 if split == 3 else 1e-2, atol=1e-7 if split == 3 else 1e-4)


@pytest.mark.parametrize("convs", ["bf16x3", "mixed", "bf16"])
def test_full_model_at_baseline_shape_vs_oracle(hip, convs):
    """One item of BASELINE configs[2] ([T,N] = [32,63], 16x16 patches, 2016 patches through the whole HIP conv
    trunk + affinity + walk) against the CPU oracle.  Forward: loss within 1e-4 relative (the north_star tolerance)
    in EVERY conv arithmetic ("mixed" runs the default's forward: the same loss and logits bit for bit).  Backward: "bf16x3" (hi/lo
    pairs, the default) every parameter gradient within 2 % of its scale and > 0.9999 cosine; "mixed" (opt-in: the backward kernels
    on the hi planes alone) within 2 % and > 0.9995; "bf16" (plain bf16 operands both ways, opt-in) within 5 % and > 0.999."""
    import model as crw_model
    import encoder as crw_encoder
    import dataset as crw_dataset
    from oracle import crw_oracle as orc
    ds = crw_dataset.RGDataset.synthetic(512, 1024, 32, (16, 16), (8, 0), seed=11)
    item = ds[1][None].contiguous()  # [1, 32, 63, 16, 16]
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    enc.hip_convs = convs
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    loss_ref, _, _ = orc.crw_forward_torch(item, sd, 0.01)
    loss_ref.backward()
    net = crw_model.CRW(enc, 0.01, False).cuda()
    loss, A = net(item.cuda())
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    loss.backward()
    cos_min, rel_max = {"bf16x3": (0.9999, 2e-2), "mixed": (0.9995, 2e-2), "bf16": (0.999, 5e-2)}[convs]
    worst = (1.0, 0.0)
    for k, p in enc.named_parameters():
        r = sd[k].grad.double().flatten()
        gq = p.grad.cpu().double().flatten()
        cos = torch.dot(r, gq) / (r.norm() * gq.norm() + 1e-300)
        worst = (min(worst[0], cos.item()), max(worst[1], ((gq - r).abs().max() / r.abs().max()).item()))
        assert cos > cos_min, (k, cos.item())
        assert (gq - r).abs().max() <= rel_max * r.abs().max(), k
    print(f"convs = {convs}: loss error {abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()):.2e} relative; worst gradient cosine "
          f"{worst[0]:.6f}, worst max-entry error {worst[1]:.2e} of the gradient's scale")
    if convs == "mixed":  # the forward is the default's
        enc.hip_convs = "bf16x3"
        loss2, A2 = net(item.cuda())
        assert loss2.item() == loss.item() and torch.equal(A2, A)


@pytest.mark.parametrize("stride", [1, 2])
def test_shared_column_encoding_matches_itemwise_and_oracle(hip, stride):
    """CRW.forward_columns (every patch-column encoded once, windows of the shared
    affinity walked per item) == CRW.forward on the batch of overlapping items == the CPU oracle on
    that batch (loss and encoder gradients)."""
    import model as crw_model
    import encoder as crw_encoder
    import dataset as crw_dataset
    from oracle import crw_oracle as orc
    T = 5
    ds = crw_dataset.RGDataset.synthetic(64, 192, T, (16, 16), (8, 0), seed=5)  # 8 items of [5, 7, 16, 16]
    cols = ds.columns()                                                         # [12, 7, 16, 16]
    items = torch.stack([ds[i] for i in range(0, len(ds), stride)])
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    loss_ref, _, _ = orc.crw_forward_torch(items, sd, 0.05)
    loss_ref.backward()
    net = crw_model.CRW(enc, 0.05, False).cuda()
    grads = {}
    for mode in ("items", "columns"):
        net.zero_grad()
        if mode == "items":
            loss, A = net(items.cuda())
        else:
            loss, A = net.forward_columns(cols[None].cuda(), T, stride)
            assert A.shape == (1, cols.shape[0] - 1, 7, 7)
        assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()), mode
        loss.backward()
        grads[mode] = {k: p.grad.clone() for k, p in enc.named_parameters()}
    for k, ref in sd.items():
        r = ref.grad.numpy()
        np.testing.assert_allclose(grads["columns"][k].cpu().numpy(), r, rtol=2e-2, atol=2e-3 * np.abs(r).max())
        np.testing.assert_allclose(grads["columns"][k].cpu().numpy(), grads["items"][k].cpu().numpy(),
                                   rtol=1e-3, atol=1e-4 * np.abs(r).max())
    with pytest.raises(ValueError):
        net.forward_columns(cols[None].cuda(), 2)


def _to_planes_ref(t):
    """fp32 NCHW [P,C,10,10] -> fp32 channels-last [P,100,C] (what the bf16 planes represent)."""
    P, C = t.shape[:2]
    return t.permute(0, 2, 3, 1).reshape(P, 100, C).contiguous()


def _planes_value(h, l):
    v = h.float()
    return v + l.float() if l is not None else v


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("split", [3, 1])
def test_encoder_conv_kernels_match_torch(hip, cin, cout, split):
    """forward (bias + ReLU [+ GAP]), backward-data (with ReLU mask) and weight/bias gradient of one
    3x3 layer against fp64 torch-CPU convolutions."""
    import torch.nn.functional as TF
    P = 5
    g = torch.Generator().manual_seed(cin + cout + split)
    x = torch.randn(P, cin, 10, 10, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    # hi + lo carries 16 mantissa bits: ~1e-5 relative (fp32-grade for the 1e-4 parity bar)
    tol = dict(rtol=5e-5, atol=5e-5) if split == 3 else dict(rtol=3e-2, atol=3e-2)
    fh, fl, bh, bl = hip.enc_pack_weights(w.cuda(), split)
    if cin == 32:
        xh, xl = hip.enc_pack_input(x.cuda(), split)
        torch.testing.assert_close(_planes_value(xh, xl).cpu(), _to_planes_ref(x), rtol=2e-5 if split == 3 else 1e-2,
                                   atol=2e-5 if split == 3 else 1e-2)
    else:  # other layers get planes from a previous conv: build them via a fake 'previous activation'
        xp = _to_planes_ref(x).cuda()
        xh = xp.bfloat16()
        xl = (xp - xh.float()).bfloat16() if split == 3 else None
    xq = _planes_value(xh, xl).cpu().double().reshape(P, 10, 10, cin).permute(0, 3, 1, 2)
    wq = w.cuda().bfloat16().float().cpu().double() if split == 1 else w.double()
    y_ref = TF.relu(TF.conv2d(xq, wq, b.double(), padding=1))
    yh, yl, yf, gap = hip.enc_conv3x3(0, split, xh, xl, fh, fl, cout, bias=b.cuda(), f32=True, gap=(cout == 128))
    torch.testing.assert_close(yf.cpu().double(), y_ref.permute(0, 2, 3, 1).reshape(P, 100, cout), **tol)
    torch.testing.assert_close(_planes_value(yh, yl).cpu().double(), _to_planes_ref(y_ref), **tol)
    if cout == 128:
        torch.testing.assert_close(gap.cpu().double(), y_ref.mean((2, 3)), **tol)
    # backward-data with the ReLU mask of the layer below (= sign of x here) and without
    dy = torch.randn(P, cout, 10, 10, generator=g)
    dyp = _to_planes_ref(dy).cuda()
    dyh = dyp.bfloat16()
    dyl = (dyp - dyh.float()).bfloat16() if split == 3 else None
    dyq = _planes_value(dyh, dyl).cpu().double().reshape(P, 10, 10, cout).permute(0, 3, 1, 2)
    dx_ref = TF.conv_transpose2d(dyq, wq, padding=1)
    maskp = _to_planes_ref(TF.relu(x)).cuda().bfloat16()
    _, _, dxf, _ = hip.enc_conv3x3(1, split, dyh, dyl, bh, bl, cin, planes=False, f32=True)
    torch.testing.assert_close(dxf.cpu().double(), dx_ref.permute(0, 2, 3, 1).reshape(P, 100, cin), **tol)
    dmh, dml, _, _ = hip.enc_conv3x3(1, split, dyh, dyl, bh, bl, cin, mask=maskp)
    torch.testing.assert_close(_planes_value(dmh, dml).cpu().double(), _to_planes_ref(dx_ref * (x > 0)), **tol)
    if cout == 128:  # fused ReLU + GAP backward: dY = dgap/100 gated by the forward activation plane
        dgap = torch.randn(P, cout, generator=g).cuda()
        gh, gl = hip.enc_gap_bwd(dgap, yh, split)
        a1 = hip.enc_conv3x3(1, split, gh, gl, bh, bl, cin, mask=maskp)
        a2 = hip.enc_conv3x3(1, split, yh, None, bh, bl, cin, mask=maskp, dgap=dgap)
        assert torch.equal(a1[0], a2[0]) and (split == 1 or torch.equal(a1[1], a2[1]))
        w1 = hip.enc_wgrad(split, gh, gl, xh, xl)
        w2 = hip.enc_wgrad(split, yh, None, xh, xl, dgap=dgap)
        assert torch.equal(w1[0], w2[0]) and torch.equal(w1[1], w2[1])
    # weight / bias gradient
    dw, db = hip.enc_wgrad(split, dyh, dyl, xh, xl)
    dw_ref = torch.nn.grad.conv2d_weight(xq, w.shape, dyq, padding=1)
    wtol = dict(rtol=1e-4, atol=1e-4 * dw_ref.abs().max().item()) if split == 3 else \
        dict(rtol=5e-2, atol=2e-2 * dw_ref.abs().max().item())
    torch.testing.assert_close(dw.cpu().double(), dw_ref, **wtol)
    torch.testing.assert_close(db.cpu().double(), dyq.sum((0, 2, 3)), **wtol)


@pytest.mark.parametrize("cin,cout", [(128, 128), (64, 128), (32, 64)])
def test_encoder_kernels_full_size_by_replication(hip, cin, cout):
    """BASELINE workload size (P = 16128 patches, where the weight gradient runs 128-512 patch slices with the
    XCD-aware workgroup mapping) checked through a size-independent property: the batch is 8 distinct patches
    replicated 2016 times, so every replica's conv output must equal the 8-patch run bit for bit and the weight /
    bias gradients must be 2016x the 8-patch gradients; also the front-end kernels at that size."""
    P0, reps, split = 8, 2016, 3
    g = torch.Generator().manual_seed(cin)
    xp = (torch.randn(P0, 100, cin, generator=g) * 0.5).cuda()
    dyp = (torch.randn(P0, 100, cout, generator=g) * 0.5).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    fh, fl, bh, bl = hip.enc_pack_weights(w, split)
    hl = lambda t: (t.bfloat16(), (t - t.bfloat16().float()).bfloat16())
    (xh, xl), (dh, dl) = hl(xp), hl(dyp)
    rep = lambda t: t.repeat(reps, 1, 1).contiguous()
    y_s = hip.enc_conv3x3(0, split, xh, xl, fh, fl, cout, bias=b)
    y_f = hip.enc_conv3x3(0, split, rep(xh), rep(xl), fh, fl, cout, bias=b)
    d_s = hip.enc_conv3x3(1, split, dh, dl, bh, bl, cin, mask=xh)
    d_f = hip.enc_conv3x3(1, split, rep(dh), rep(dl), bh, bl, cin, mask=rep(xh))
    for small, full in ((y_s, y_f), (d_s, d_f)):
        for pl in (0, 1):
            v = full[pl].view(reps, P0, 100, -1)
            assert torch.equal(v[0], small[pl]) and torch.equal(v[reps - 1], small[pl]) and torch.equal(v[reps // 2], small[pl])
    dw_s, db_s = hip.enc_wgrad(split, dh, dl, xh, xl)
    dw_f, db_f = hip.enc_wgrad(split, rep(dh), rep(dl), rep(xh), rep(xl))
    torch.testing.assert_close(dw_f, dw_s * reps, rtol=2e-4, atol=2e-4 * (dw_s.abs().max().item() * reps))
    torch.testing.assert_close(db_f, db_s * reps, rtol=2e-4, atol=2e-4 * (db_s.abs().max().item() * reps))
    dw_again, db_again = hip.enc_wgrad(split, rep(dh), rep(dl), rep(xh), rep(xl))
    assert torch.equal(dw_f, dw_again) and torch.equal(db_f, db_again)  # ordered slab sums: bitwise reproducible
    if cin == 32:  # the fused front end at the same size
        x0 = torch.randn(P0, 1, 16, 16, generator=g).cuda()
        w1 = (torch.randn(8, 1, 5, 5, generator=g) * 0.2).cuda()
        b1 = (torch.randn(8, generator=g) * 0.1).cuda()
        w2 = (torch.randn(32, 8, 5, 5, generator=g) * 0.07).cuda()
        b2 = (torch.randn(32, generator=g) * 0.1).cuda()
        w2p = hip.enc_front_pack(w2, split)
        o_s = hip.enc_front_fwd(split, x0, w1, b1, w2p[:2], b2)
        o_f = hip.enc_front_fwd(split, x0.repeat(reps, 1, 1, 1).contiguous(), w1, b1, w2p[:2], b2)
        for pl in (0, 1):
            v = o_f[pl].view(reps, P0, 100, 32)
            assert torch.equal(v[0], o_s[pl]) and torch.equal(v[reps - 1], o_s[pl])
        dy0 = torch.randn(P0, 100, 32, generator=g).cuda()
        g_s = hip.enc_front_bwd(split, x0, w1, b1, w2p[:2], b2, w2p[2:], dy0)
        xr = x0.repeat(reps, 1, 1, 1).contiguous()
        sv_f = hip.enc_front_fwd(split, xr, w1, b1, w2p[:2], b2, save=True)[2]
        g_f = hip.enc_front_bwd(split, xr, w1, b1, w2p[:2], b2, w2p[2:], rep(dy0), saved=sv_f)  # the training path's kernel
        g_r = hip.enc_front_bwd(split, xr, w1, b1, w2p[:2], b2, w2p[2:], rep(dy0))              # recomputing kernel
        for a_, b_ in zip(g_f, g_r):
            assert torch.equal(a_, b_)
        for a_, b_ in zip(g_f, g_s):
            torch.testing.assert_close(a_, b_ * reps, rtol=2e-4, atol=2e-4 * (b_.abs().max().item() * reps))


@pytest.mark.parametrize("reps", [25, 37, 129])
@pytest.mark.parametrize("split", [3, 1])
def test_wgrad_patch_slice_lengths(hip, reps, split):
    """Weight gradient with 2, 3 and 9 patches per workgroup slice (P = 200, 296, 1032 against 128 slices): the
    k-steps that stream across patch boundaries end with 8, 0 or 0 deferred pixels, i.e. with and without the final
    flush step.  Checked by replication: gradients must be `reps` times those of the 8 distinct patches."""
    P0, cin, cout = 8, 128, 128
    g = torch.Generator().manual_seed(reps)
    xp = (torch.randn(P0, 100, cin, generator=g) * 0.5).cuda()
    dyp = (torch.randn(P0, 100, cout, generator=g) * 0.5).cuda()
    hl = lambda t: (t.bfloat16(), (t - t.bfloat16().float()).bfloat16() if split == 3 else None)
    (xh, xl), (dh, dl) = hl(xp), hl(dyp)
    rep = lambda t: None if t is None else t.repeat(reps, 1, 1).contiguous()
    dw_s, db_s = hip.enc_wgrad(split, dh, dl, xh, xl)
    dw_f, db_f = hip.enc_wgrad(split, rep(dh), rep(dl), rep(xh), rep(xl))
    torch.testing.assert_close(dw_f, dw_s * reps, rtol=2e-4, atol=2e-4 * (dw_s.abs().max().item() * reps))
    torch.testing.assert_close(db_f, db_s * reps, rtol=2e-4, atol=2e-4 * (db_s.abs().max().item() * reps))
    # and against fp64 on the 8 distinct patches
    xq = ((xh.float() + (xl.float() if xl is not None else 0)).cpu().double()).reshape(P0, 10, 10, cin).permute(0, 3, 1, 2)
    dq = ((dh.float() + (dl.float() if dl is not None else 0)).cpu().double()).reshape(P0, 10, 10, cout).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xq, (cout, cin, 3, 3), dq, padding=1)
    torch.testing.assert_close(dw_s.cpu().double(), ref, rtol=1e-4, atol=1e-4 * ref.abs().max().item())


@pytest.mark.parametrize("cin", [1, 2])
@pytest.mark.parametrize("split", [3, 1])
def test_encoder_front_kernels_match_torch(hip, cin, split):
    """fused conv1-ReLU-pool-conv2-ReLU-pool forward and its backward against fp64 torch autograd."""
    import torch.nn.functional as TF
    P = 7
    g = torch.Generator().manual_seed(10 * cin + split)
    x = torch.randn(P, cin, 16, 16, generator=g)
    w1 = (torch.randn(8, cin, 5, 5, generator=g) * 0.2).double().requires_grad_(True)
    b1 = (torch.randn(8, generator=g) * 0.1).double().requires_grad_(True)
    w2 = (torch.randn(32, 8, 5, 5, generator=g) * 0.07).double().requires_grad_(True)
    b2 = (torch.randn(32, generator=g) * 0.1).double().requires_grad_(True)
    a1 = TF.max_pool2d(TF.relu(TF.conv2d(x.double(), w1, b1, padding=1)), 2, 1)
    y = TF.max_pool2d(TF.relu(TF.conv2d(a1, w2, b2, padding=1)), 2, 1)  # [P,32,10,10]
    dy = torch.randn(P, 100, 32, generator=g)
    y.backward(dy.double().reshape(P, 10, 10, 32).permute(0, 3, 1, 2))
    c = lambda t: t.detach().float().cuda()
    w2f = hip.enc_front_pack(c(w2), split)
    yh, yl = hip.enc_front_fwd(split, x.cuda(), c(w1), c(b1), w2f[:2], c(b2))
    yh2, yl2, saved = hip.enc_front_fwd(split, x.cuda(), c(w1), c(b1), w2f[:2], c(b2), save=True)
    assert torch.equal(yh, yh2) and (yl is None or torch.equal(yl, yl2))
    tol = dict(rtol=1e-4, atol=1e-4) if split == 3 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(_planes_value(yh, yl).cpu().double(), _to_planes_ref(y.detach()), **tol)
    dw1, db1, dw2, db2 = hip.enc_front_bwd(split, x.cuda(), c(w1), c(b1), w2f[:2], c(b2), w2f[2:], dy.cuda())
    # the backward from the forward's saved record (pool1 planes + pooling codes, no recomputation): the same arithmetic on
    # the same values, so the same bits
    sv = hip.enc_front_bwd(split, x.cuda(), c(w1), c(b1), w2f[:2], c(b2), w2f[2:], dy.cuda(), saved=saved)
    for a_, b_ in zip(sv, (dw1, db1, dw2, db2)):
        assert torch.equal(a_, b_)
    for got, ref in ((dw2, w2.grad), (db2, b2.grad), (dw1, w1.grad), (db1, b1.grad)):
        # plain bf16: rounding conv2's inputs can flip a max-pool arg-max, which re-routes a gradient
        t = dict(rtol=2e-3, atol=2e-3 * ref.abs().max().item()) if split == 3 else \
            dict(rtol=2e-1, atol=1.5e-1 * ref.abs().max().item())
        torch.testing.assert_close(got.cpu().double(), ref, **t)


@pytest.mark.parametrize("cin,hw", [(1, (20, 27)), (2, (32, 32)), (1, (16, 16)), (1, (9, 12))])
@pytest.mark.parametrize("split", [3, 1])
def test_front_backward_on_tiles_matches_torch(hip, cin, hw, split):
    """conv1-ReLU-pool-conv2-ReLU-pool backward on patches of any size (`crw_enc_front_bwd_map`: units of 10x10 output tiles,
    partial tiles, a one-tile 16x16 case and a map smaller than a tile) against fp64 torch autograd."""
    import torch.nn.functional as TF
    P = 5
    h, w = hw
    g = torch.Generator().manual_seed(100 * cin + h + split)
    x = torch.randn(P, cin, h, w, generator=g)
    w1 = (torch.randn(8, cin, 5, 5, generator=g) * 0.2).double().requires_grad_(True)
    b1 = (torch.randn(8, generator=g) * 0.1).double().requires_grad_(True)
    w2 = (torch.randn(32, 8, 5, 5, generator=g) * 0.07).double().requires_grad_(True)
    b2 = (torch.randn(32, generator=g) * 0.1).double().requires_grad_(True)
    a1 = TF.max_pool2d(TF.relu(TF.conv2d(x.double(), w1, b1, padding=1)), 2, 1)
    y = TF.max_pool2d(TF.relu(TF.conv2d(a1, w2, b2, padding=1)), 2, 1)  # [P,32,h-6,w-6]
    H, W = h - 6, w - 6
    dy = torch.randn(P, H * W, 32, generator=g)
    y.backward(dy.double().reshape(P, H, W, 32).permute(0, 3, 1, 2))
    c = lambda t: t.detach().float().cuda()
    w2f = hip.enc_front_pack(c(w2), split)
    yh, yl = hip.enc_front_fwd_map(split, x.cuda(), c(w1), c(b1), w2f[:2], c(b2))
    tol = dict(rtol=1e-4, atol=1e-4) if split == 3 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(_planes_value(yh, yl).cpu().double(), y.detach().permute(0, 2, 3, 1).reshape(P, H * W, 32), **tol)
    dw1, db1, dw2, db2 = hip.enc_front_bwd_map(split, x.cuda(), c(w1), c(b1), w2f[:2], c(b2), w2f[2:], dy.cuda())
    for got, ref in ((dw2, w2.grad), (db2, b2.grad), (dw1, w1.grad), (db1, b1.grad)):
        t = dict(rtol=2e-3, atol=2e-3 * ref.abs().max().item()) if split == 3 else \
            dict(rtol=2e-1, atol=1.5e-1 * ref.abs().max().item())
        torch.testing.assert_close(got.cpu().double(), ref, **t)


LP_CASES = ["labelprop_trunc_T14N10", "labelprop_full_T40N48", "labelprop_last_T20N24", "labelprop_mc1_T100N12"]


class _Flatten(torch.nn.Module):
    def forward(self, x):
        return x.flatten(1)


@pytest.mark.parametrize("name", LP_CASES)
def test_propagate_matches_reference(hip, name):
    import utils as crw_utils
    from imported.labelprop import LabelPropVOS_CRW
    g = load_golden(name)
    T, N, C = g["emb"].shape
    cfg = dict(CXT_SIZE=int(g["cxt_size"]), RADIUS=int(g["radius"]), TEMP=float(g["temp"]), KNN=int(g["knn"]))
    seq = dev(g["emb"]).reshape(T, N, C // 4, 4)
    pred, xent, change_idx = crw_utils.propagate(seq, dev(g["seg_ref"]), _Flatten(), LabelPropVOS_CRW(cfg),
                                                 int(g["nclasses"]), False, bool(g["use_last"]))
    assert pred.shape == (N, T) and xent.shape == (N, T - 1) and not xent.is_cuda
    assert np.array_equal(pred.cpu().numpy(), g["pred"]), f"{(pred.cpu().numpy() != g['pred']).sum()} labels differ"
    np.testing.assert_allclose(xent.numpy(), g["xent"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["labelprop_trunc_T14N10", "labelprop_last_T20N24"])
def test_propagate_with_a_foreign_label_propagation_object(hip, name):
    """`utils.propagate` with an `lp` that only offers the reference's frame-by-frame `predict(feats, masks, curr_feat)` protocol
    (src/imported/labelprop.py:67; no `propagate_all`): the fallback branch that stacks features / masks like src/utils.py:148-160
    must give the reference's label map too."""
    import utils as crw_utils
    from imported.labelprop import LabelPropVOS_CRW
    g = load_golden(name)
    T, N, C = g["emb"].shape
    cfg = dict(CXT_SIZE=int(g["cxt_size"]), RADIUS=int(g["radius"]), TEMP=float(g["temp"]), KNN=int(g["knn"]))

    class PredictOnly:  # the reference's interface and nothing more
        def __init__(self, inner):
            self.inner, self.calls = inner, 0

        def predict(self, feats, masks, curr_feat):
            self.calls += 1
            assert len(feats) == len(masks) == self.calls and curr_feat.shape == (1, C, N, 1) and masks[0].shape[2:] == (N, 1)
            return self.inner.predict(feats, masks, curr_feat)

    lp = PredictOnly(LabelPropVOS_CRW(cfg))
    seq = dev(g["emb"]).reshape(T, N, C // 4, 4)
    pred, xent, _ = crw_utils.propagate(seq, dev(g["seg_ref"]), _Flatten(), lp, int(g["nclasses"]), False, bool(g["use_last"]))
    assert lp.calls == T - 1
    assert np.array_equal(pred.cpu().numpy(), g["pred"]), f"{(pred.cpu().numpy() != g['pred']).sum()} labels differ"
    np.testing.assert_allclose(xent.numpy(), g["xent"], rtol=1e-4, atol=1e-4)


def test_predict_frame_by_frame_equals_batched(hip):
    from imported.labelprop import LabelPropVOS_CRW
    g = load_golden("labelprop_trunc_T14N10")
    T, N, C = g["emb"].shape
    M = int(g["nclasses"])
    cfg = dict(CXT_SIZE=int(g["cxt_size"]), RADIUS=int(g["radius"]), TEMP=float(g["temp"]), KNN=int(g["knn"]))
    feats = hip.normalize(dev(g["emb"]))
    seed = dev(orc.seed_labels(g["seg_ref"], N))
    pred_all, L = LabelPropVOS_CRW(cfg).propagate_all(feats, seed, M)
    lp = LabelPropVOS_CRW(cfg)
    as_feat = lambda n: feats[n].t()[None, :, :, None]
    fl = [as_feat(0)]
    ml = [(seed[None, :] == torch.arange(M).cuda()[:, None]).float()[None, :, :, None]]
    for n in range(1, T):
        m = lp.predict(fl, ml, as_feat(n))
        assert m.shape == (1, M, N, 1)
        torch.testing.assert_close(m[0, :, :, 0].t(), L[n * N:(n + 1) * N], rtol=0, atol=0)
        fl.append(as_feat(n))
        ml.append(m)
    assert np.array_equal(pred_all.cpu().numpy(), g["pred"])
    assert lp.mask.shape == (1, N, N) and lp.mask_hw == (N, 1)


@pytest.mark.parametrize("name", ["labelprop_grid_T10_6x5", "labelprop_grid_T8_4x9"])
def test_labelprop_on_2d_grids_matches_reference(hip, name):
    """2-D node grids (the reference's `predict` takes any h x w; its mask is the Euclidean disc of src/imported/maskedatt.py:222-245):
    frame-by-frame `predict` on [1,C,h,w] features and the batched `propagate_all(grid_w=w)` give the label maps the reference
    produced; and random grids (radius larger than the grid, radius 1, non-square) against the oracle."""
    from imported.labelprop import LabelPropVOS_CRW
    g = load_golden(name)
    h, w = (int(v) for v in g["grid"])
    T, N, C = g["emb"].shape
    M = int(g["nclasses"])
    cfg = dict(CXT_SIZE=int(g["cxt_size"]), RADIUS=int(g["radius"]), TEMP=float(g["temp"]), KNN=int(g["knn"]))
    feats = hip.normalize(dev(g["emb"]))
    seed = dev(g["seed_labels"])
    pred_all, L = LabelPropVOS_CRW(cfg).propagate_all(feats, seed, M, grid_w=w)
    assert np.array_equal(pred_all.cpu().numpy(), g["pred"])
    lp = LabelPropVOS_CRW(cfg)
    as_feat = lambda n: feats[n].t().reshape(1, C, h, w)
    fl = [as_feat(0)]
    ml = [(seed[None, :] == torch.arange(M).cuda()[:, None]).float().reshape(1, M, h, w)]
    for n in range(1, T):
        m = lp.predict(fl, ml, as_feat(n))
        assert m.shape == (1, M, h, w)
        torch.testing.assert_close(m.reshape(M, N).t(), L[n * N:(n + 1) * N], rtol=0, atol=0)
        assert np.array_equal(m.argmax(1).flatten().cpu().numpy(), g["pred"][:, n])
        fl.append(as_feat(n))
        ml.append(m)
    assert lp.mask.shape == (1, N, N) and lp.mask_hw == (h, w)
    assert np.array_equal(lp.mask[0].cpu().numpy(), orc.band_bias(N, cfg["RADIUS"], grid_w=w))
    for (T2, h2, w2, C2, M2, cxt, radius, knn) in ((7, 5, 7, 12, 3, 2, 9, 6), (6, 4, 4, 8, 2, 10, 1, 3), (9, 3, 11, 20, 4, 3, 3, 8)):
        gen = torch.Generator().manual_seed(h2 * 10 + w2)
        emb = (torch.randn(1, h2 * w2, C2, generator=gen) + 0.4 * torch.randn(T2, h2 * w2, C2, generator=gen)).float()
        sd = (torch.arange(h2 * w2) // w2 * M2 // h2).float()
        ref = orc.labelprop(emb.numpy(), sd.numpy(), M2, cxt, radius, 0.1, knn, grid_w=w2)
        got, _ = LabelPropVOS_CRW(dict(CXT_SIZE=cxt, RADIUS=radius, TEMP=0.1, KNN=knn)).propagate_all(hip.normalize(emb.cuda()), sd.cuda(), M2, grid_w=w2)
        assert np.array_equal(got.cpu().numpy(), ref), (h2, w2, radius)


@pytest.mark.parametrize("T,N,C,M,cxt,radius,knn,temp", [(10, 12, 16, 3, 3, 2, 5, 0.1),    # knn > in-band keys of the first frames
                                                         (9, 7, 10, 2, 2, 1, 3, 0.05),    # radius 1: only the same node is in band
                                                         (30, 20, 24, 4, 5, 30, 20, 0.2),  # radius > N: no mask at all, knn = N
                                                         (6, 5, 7, 3, 100, 2, 2, 0.1)])   # C not a multiple of 4 (scalar dot path)
def test_labelprop_edge_cases_match_oracle(hip, T, N, C, M, cxt, radius, knn, temp):
    from imported.labelprop import LabelPropVOS_CRW
    g = torch.Generator().manual_seed(T * 100 + N)
    proto = torch.randn(N + 8, C, generator=g)
    t = torch.arange(T).float()
    depth = torch.arange(N).float()[None] + 1.5 * torch.sin(2 * np.pi * t / 7)[:, None] + 3
    lo = depth.floor().long()
    fr = (depth - lo.float()).unsqueeze(-1)
    emb = (proto[lo] * (1 - fr) + proto[lo + 1] * fr + 0.3 * torch.randn(T, N, C, generator=g)).float()
    seed = (torch.arange(N) * M // N).float()
    ref = orc.labelprop(emb.numpy(), seed.numpy(), M, cxt, radius, temp, knn)
    feats = hip.normalize(emb.cuda())
    pred, _ = LabelPropVOS_CRW(dict(CXT_SIZE=cxt, RADIUS=radius, TEMP=temp, KNN=knn)).propagate_all(feats, seed.cuda(), M)
    assert np.array_equal(pred.cpu().numpy(), ref), f"{(pred.cpu().numpy() != ref).sum()} of {ref.size} labels differ"


@pytest.mark.parametrize("T,N,C,cxt,radius,knn,first", [
    (256, 48, 128, 80, 10, 20, 1),  # BASELINE config 5
    (40, 21, 64, 6, 4, 9, 1),       # node count not a multiple of the 16-query tile; truncated context
    (12, 16, 128, 20, 30, 64, 1),   # radius larger than the column, more neighbours asked for than there are candidates
    (30, 37, 256, 5, 1, 3, 4),      # radius 1 (the query's own node only), 256 channels, later first frame
    (70, 100, 128, 9, 17, 12, 1)])  # three key tiles per query tile and more
def test_labelprop_topk_on_matrix_cores_agrees_with_vector_kernel(hip, T, N, C, cxt, radius, knn, first):
    """A radargram's column (N x 1 grid, N >= 16, C = 64 / 128 / 256) scores on the fp32 matrix cores
    (csrc/labelprop.hip labelprop_topk_mfma_kernel); a 1 x N grid is the same problem -- same candidates in the same order -- and runs
    the vector kernel.  Exact fp32 products in both, different summation order: the lists must agree except where two candidates
    score within rounding of each other (then they carry the same weight to 1e-6), and the scores against fp64."""
    g = torch.Generator().manual_seed(T + N)
    feats = hip.normalize((torch.randn(1, N, C, generator=g) + 0.5 * torch.randn(T, N, C, generator=g)).float().cuda())
    Wm, Im = hip.labelprop_topk(feats, cxt, radius, 0.1, knn, first_frame=first, grid_w=1)
    Wv, Iv = hip.labelprop_topk(feats, cxt, radius, 0.1, knn, first_frame=first, grid_w=N)
    torch.testing.assert_close(Wm, Wv, rtol=2e-5, atol=1e-6)
    differ = Im != Iv
    assert differ.float().mean().item() < 1e-3, f"{differ.sum().item()} of {differ.numel()} neighbours differ"
    # a swapped pair holds two near-equal scores: the weights at those positions agree (checked above); every list is a set of
    # distinct candidates with weights in descending order
    assert (Wm[:, :-1] >= Wm[:, 1:] - 1e-7).all()
    ws = Wm.sum(1)
    torch.testing.assert_close(ws, torch.ones_like(ws), rtol=1e-5, atol=1e-5)
    # exact ties: every odd frame repeats the frame before it, so each key has a twin with the SAME score bit for bit inside either
    # kernel -- the rule (highest value, then lowest candidate index) must then pick the same twin in both
    feats2 = feats.clone()
    feats2[1::2] = feats2[0:T - 1:2][: feats2[1::2].shape[0]]
    Wm2, Im2 = hip.labelprop_topk(feats2, cxt, radius, 0.1, knn, first_frame=first, grid_w=1)
    Wv2, Iv2 = hip.labelprop_topk(feats2, cxt, radius, 0.1, knn, first_frame=first, grid_w=N)
    torch.testing.assert_close(Wm2, Wv2, rtol=2e-5, atol=1e-6)
    assert (Im2 != Iv2).float().mean().item() < 2e-3


@pytest.mark.parametrize("T,N,C", [(256, 48, 128), (5, 3, 6), (9, 70, 33), (4, 130, 128)])  # the last: a frame too large for LDS twice
def test_xent_metric_matches_oracle(hip, T, N, C):
    """The 'horizontality' metric (src/utils.py:117-125; quirk Q8: channel-shifted features) at BASELINE config 5's shape and at
    shapes that take the workgroup-per-frame kernel's edges (C - 1 not a multiple of 4) and the one-wave-per-pair fallback."""
    g = torch.Generator().manual_seed(T * N + C)
    emb = (torch.randn(1, N, C, generator=g) + 0.5 * torch.randn(T, N, C, generator=g)).float()
    ref = orc.xent_metric(emb.numpy().astype(np.float64), np.float64)
    got = hip.xent_metric(hip.normalize(emb.cuda())).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


def test_labelprop_matches_oracle_mcords_shape(hip):
    """BASELINE config 5 shape [T,N] = [256,48] with truncation (CXT_SIZE=80)."""
    from imported.labelprop import LabelPropVOS_CRW
    g = torch.Generator().manual_seed(77)
    T, N, C, M = 256, 48, 128, 4
    proto = torch.randn(N + 16, C, generator=g)
    t = torch.arange(T).float()
    depth = torch.arange(N).float()[None] + 3 * torch.sin(2 * np.pi * t / 40)[:, None] + 6
    lo = depth.floor().long()
    fr = (depth - lo.float()).unsqueeze(-1)
    emb = (proto[lo] * (1 - fr) + proto[lo + 1] * fr + 0.3 * torch.randn(T, N, C, generator=g)).float()
    seed = (torch.arange(N) * M // N).float()
    ref = orc.labelprop(emb.numpy(), seed.numpy(), M, 80, 10, 0.1, 20)
    feats = hip.normalize(emb.cuda())
    pred, _ = LabelPropVOS_CRW(dict(CXT_SIZE=80, RADIUS=10, TEMP=0.1, KNN=20)).propagate_all(feats, seed.cuda(), M)
    mism = (pred.cpu().numpy() != ref).sum()
    assert mism == 0, f"{mism} of {ref.size} labels differ"


@pytest.mark.parametrize("T,N,C,M,cxt,radius,knn,grid_w", [
    (256, 48, 128, 4, 80, 10, 20, 1),   # BASELINE config 5: 80 chained frames in one workgroup + 175 frames at once
    (14, 10, 16, 3, 3, 4, 5, 1),        # short chain, long tail
    (9, 12, 16, 3, 20, 3, 4, 1),        # T <= cxt + 1: every frame chained, no tail
    (2, 10, 16, 3, 3, 4, 5, 1),         # two frames: one propagated frame (the seed path of the one-workgroup kernel)
    (3, 9, 8, 2, 1, 2, 3, 1),           # three frames, cxt = 1: one chained frame + one tail frame
    (40, 63, 64, 6, 7, 5, 10, 1),       # 378 outputs per frame: six compute waves
    (30, 24, 32, 2, 1, 6, 24, 1),       # cxt = 1: frame 1 alone is chained; 24 neighbours (the widest register instance)
    (12, 30, 16, 3, 4, 3, 7, 5),        # 6 x 5 node grid
    (20, 120, 32, 5, 3, 4, 6, 1),       # 600 outputs per frame: beyond the compute waves -> the one-workgroup walk
    (16, 40, 32, 3, 5, 40, 30, 1)])     # 30 neighbours: beyond the register instances -> the one-workgroup walk
def test_labelprop_propagate_equals_the_one_workgroup_walk(hip, T, N, C, M, cxt, radius, knn, grid_w):
    """crw_labelprop_propagate (chained frames first_frame..cxt in one workgroup with their labels in LDS, every later frame at once:
    the truncated lists' indices are < min(n, cxt + 1) * N, src/imported/maskedatt.py:165-166 applied as in
    src/imported/labelprop.py:103-107) against crw_labelprop_gather (one workgroup, frame after frame, any lists): soft labels and
    label map bit for bit -- for the whole radargram, for a later first frame with the earlier labels given, and one frame at a time."""
    g = torch.Generator().manual_seed(T * 7 + N)
    feats = hip.normalize((torch.randn(1, N, C, generator=g) + 0.5 * torch.randn(T, N, C, generator=g)).float().cuda())
    seed = (torch.arange(N) * M // N).float().cuda()
    Wt, It = hip.labelprop_topk(feats, cxt, radius, 0.1, knn, first_frame=1, grid_w=grid_w)
    for n in range(1, T):  # the bound the new entry point relies on
        assert int(It[n - 1].max()) < min(n, cxt + 1) * N
    L0, p0 = hip.labelprop_gather(seed, Wt, It, T, N, M, first_frame=1)
    L1, p1 = hip.labelprop_gather(seed, Wt, It, T, N, M, first_frame=1, cxt_size=cxt)
    assert torch.equal(L0, L1) and torch.equal(p0, p1)
    for first in sorted({2, max(2, cxt), min(T - 1, cxt + 1), min(T - 1, cxt + 3), T - 1}):
        if first >= T:
            continue
        Wf, If = hip.labelprop_topk(feats, cxt, radius, 0.1, knn, first_frame=first, grid_w=grid_w)
        assert torch.equal(Wf, Wt[first - 1:]) and torch.equal(If, It[first - 1:])
        La, Lb = L0.clone(), L0.clone()
        La[first * N:] = -7.0
        Lb[first * N:] = -7.0
        pa, pb = torch.full((N, T), -1.0).cuda(), torch.full((N, T), -1.0).cuda()
        hip.labelprop_gather(None, Wf, If, T, N, M, first_frame=first, L=La, pred=pa)
        hip.labelprop_gather(None, Wf, If, T, N, M, first_frame=first, L=Lb, pred=pb, cxt_size=cxt)
        assert torch.equal(La, L0) and torch.equal(Lb, L0), first
        assert torch.equal(pa, pb) and torch.equal(pb[:, first:], p0[:, first:]) and (pb[:, :first] == -1).all(), first
    # one frame per call (LabelPropVOS_CRW.predict)
    Ls = L0.clone()
    Ls[N:] = 0
    for n in range(1, T):
        Wn, In = hip.labelprop_topk(feats[:n + 1], cxt, radius, 0.1, knn, first_frame=n, grid_w=grid_w)
        hip.labelprop_gather(None, Wn, In, n + 1, N, M, first_frame=n, L=Ls[:(n + 1) * N], cxt_size=cxt)
    assert torch.equal(Ls, L0)


def test_bidirectional_segmentation_pipeline_matches_oracle(hip):
    """forward pass + reversed pass (use_last) + class-2 merge over two radargram items, checked
    against the same pipeline restated with the CPU oracle (scripts/test/test_all.py:91-159)."""
    import dataset as crw_dataset
    import inference as crw_inference
    from imported.labelprop import LabelPropVOS_CRW
    T, h, w, oh, M = 6, 8, 8, 4, 3
    rg = crw_dataset.synthetic_radargram(44, 2 * T * w, seed=5)
    ds = crw_dataset.RGDataset.from_tensor(rg, T, (h, w), (oh, 0))
    N = ds[0].shape[1]
    rows = N * (h - oh) + oh
    seg = torch.zeros(rows, rg.shape[1])
    for r in range(rows):
        seg[r] = float(min(M - 1, r * M // rows))
    seg[rows // 2:, T * w:] = 2.0  # second radargram: bedrock lower half, so the reverse merge has work to do
    enc = _Flatten()
    cfg = dict(CXT_SIZE=3, RADIUS=3, TEMP=0.1, KNN=4)
    out = crw_inference.segment(ds, seg, enc, LabelPropVOS_CRW(cfg), M, T, (h, w), (oh, 0), use_last=True, dataset_id=0)
    rg_len = T * w

    def oracle_pass(item, seg_ref, reverse):
        emb = item.reshape(T, N, -1).numpy()
        if reverse:
            emb = emb[::-1].copy()
        seed = orc.seed_labels(seg_ref.numpy(), N)
        pred = orc.labelprop(emb, seed, M, cfg["CXT_SIZE"], cfg["RADIUS"], cfg["TEMP"], cfg["KNN"])
        ri = np.floor(np.arange(rows) * (N / rows)).astype(int)
        ci = np.floor(np.arange(rg_len) * (T / rg_len)).astype(int)
        return pred[ri][:, ci]

    fwd, rev = [], []
    seg_rev = torch.flip(seg.unfold(1, rg_len, rg_len), (-1,)).reshape(rows, -1)
    for t, i in enumerate(range(0, len(ds), T)):
        fwd.append(oracle_pass(ds[i], seg[:rows, rg_len * t:rg_len * t + w], False))
        rev.append(oracle_pass(ds[i], seg_rev[:rows, rg_len * t:rg_len * t + w], True))
    fwd = np.concatenate(fwd, 1)
    rev = np.concatenate([r[:, ::-1] for r in rev], 1)
    ref = fwd.copy()
    ref[rev == 2] = 2
    got = out["pred"].cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), f"{(got != ref).sum()} of {ref.size} pixels differ"
    assert (rev == 2).any() and (fwd != ref).any()  # the merge really changed something


@pytest.mark.parametrize("name", ["segment_ds0_reverse", "segment_ds1_reverse", "segment_ds3_reverse",
                                  "segment_ds0_correction", "segment_ds3_correction_reverse"])
def test_segment_pipeline_matches_reference_main(hip, name):
    """SURVEY section 8 row f2 pinned to the reference itself: `inference.segment` on the HIP path against the int8 map the
    reference's scripts/test/test_all.py main(args) saves (forward + correction, test_all.py:91-128) and the map its
    report is computed on (reverse pass + per-dataset merge, test_all.py:132-159).  Exact."""
    import utils as crw_utils
    from test_host import run_segment_golden
    g = load_golden(name)
    out = run_segment_golden(g, crw_utils.propagate, "cuda")
    fwd = out["forward"].cpu().numpy().astype(np.int8)
    fin = out["pred"].cpu().numpy().astype(np.int8)
    assert np.array_equal(fwd, g["saved_map"]), f"{(fwd != g['saved_map']).sum()} of {fwd.size} pixels differ (forward map)"
    assert np.array_equal(fin, g["final_map"]), f"{(fin != g['final_map']).sum()} of {fin.size} pixels differ (final map)"


def test_resnet_training_step_matches_reference(hip):
    """the reference's DEFAULT encoder (`Resnet`, BatchNorm in train mode) through CRW.forward +
    backward on the GPU (PyTorch-ROCm convolutions / BatchNorm + the HIP affinity and walk) against the reference's own
    CPU run (fixture resnet_train_*): loss within 1e-4, logits, selected parameter gradients, every gradient's norm,
    BatchNorm running statistics."""
    import model as crw_model
    import encoder as crw_encoder
    g = load_golden("resnet_train_B2T4N5")
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.manual_seed(int(g["seed"]))
    enc = crw_encoder.Resnet(False)
    net = crw_model.CRW(enc, float(g["tau"]), False).cuda()
    net.train(True)
    loss, A = net(dev(g["seq"]))
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    np.testing.assert_allclose(A.detach().cpu().numpy(), g["A"], rtol=1e-3, atol=5e-3)
    loss.backward()
    np.testing.assert_allclose(enc.bn0.running_mean.cpu().numpy(), g["bn0.running_mean"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(enc.bn0.running_var.cpu().numpy(), g["bn0.running_var"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(enc.model.bn1.running_mean.cpu().numpy(), g["model.bn1.running_mean"], rtol=1e-3, atol=1e-5)
    names = [k for k, _ in enc.named_parameters()]
    assert names == list(g["grad_names"])
    for (k, p_), ref_norm in zip(enc.named_parameters(), g["grad_norms"]):
        got = float(p_.grad.double().norm())
        # (fc0.bias feeds a BatchNorm: its true gradient is zero, both sides hold rounding noise ~1e-6)
        assert abs(got - ref_norm) <= 2e-2 * ref_norm + 1e-4, (k, got, ref_norm)
        if "grad." + k in g:
            ref = g["grad." + k]
            np.testing.assert_allclose(p_.grad.cpu().numpy(), ref, rtol=2e-2, atol=2e-3 * np.abs(ref).max())


def test_flat_gradient_all_reduce_on_rccl(hip, tmp_path):
    """The `nccl` (= RCCL) branch of `FlatGradBucket.all_reduce_mean` -- `all_reduce(op=AVG)` on the flat fp32 gradient,
    dist.py -- on a one-rank process group after a HIP backward: the collective must run on the stream the gradients
    were produced on and leave them unchanged (mean over one rank)."""
    import torch.distributed as tdist
    import model as crw_model
    import encoder as crw_encoder
    import dist as crw_dist
    g, w = load_golden("cnn_cfg1_B2T8N7"), load_golden("cnn_weights_seed11")
    enc = crw_encoder.CNN(False)
    enc.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    net = crw_model.CRW(enc, float(g["tau"]), False).cuda()
    assert not tdist.is_initialized()
    tdist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1)
    try:
        bucket = crw_dist.FlatGradBucket(net.parameters(), lazy=True)
        bucket.zero()
        loss, _ = net(dev(g["seq"]))
        loss.backward()
        before = torch.cat([p.grad.reshape(-1) for p in bucket.params]).clone()
        flat = bucket.all_reduce_mean()
        torch.cuda.synchronize()
        assert crw_dist.LAST_COLLECTIVE == "nccl:avg"
        assert torch.equal(flat, before)
        for k, p_ in enc.named_parameters():
            ref = g["grad." + k]
            assert p_.grad.data_ptr() >= flat.data_ptr() and p_.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
            np.testing.assert_allclose(p_.grad.cpu().numpy(), ref, rtol=2e-2, atol=2e-3 * np.abs(ref).max())
        t = torch.full((4,), 3.0, device="cuda")
        tdist.all_reduce(t, op=tdist.ReduceOp.SUM)   # the loss all-reduce of train.py
        assert t.sum().item() == 12.0
    finally:
        tdist.destroy_process_group()


def test_flat_adam_matches_torch_adam(hip):
    """optim.FlatAdam (one crw_adam_step launch over the flat parameter / gradient buffers) against torch.optim.Adam with its
    defaults -- the reference's optimizer (scripts/train.py:54,69) -- over 6 steps of random gradients: same parameters to fp32
    rounding, the module's parameters stay views of the flat buffer, and a second module driven by torch's Adam ends up equal."""
    import dist as crw_dist
    import optim as crw_optim
    torch.manual_seed(5)
    a = torch.nn.Sequential(torch.nn.Conv2d(1, 3, 3), torch.nn.Linear(7, 5), torch.nn.Linear(5, 130)).cuda()
    b = torch.nn.Sequential(torch.nn.Conv2d(1, 3, 3), torch.nn.Linear(7, 5), torch.nn.Linear(5, 130)).cuda()
    b.load_state_dict(a.state_dict())
    bucket = crw_dist.FlatGradBucket(a.parameters(), lazy=False)
    fa = crw_optim.FlatAdam(bucket, lr=1e-2)
    ta = torch.optim.Adam(b.parameters(), lr=1e-2)
    g = torch.Generator(device="cuda").manual_seed(9)
    for step in range(6):
        scale = 10.0 ** (step - 3)  # gradients over six orders of magnitude
        for pa, pb in zip(a.parameters(), b.parameters()):
            gr = torch.randn(pa.shape, device="cuda", generator=g) * scale
            pa.grad.copy_(gr)
            pb.grad = gr.clone()
        fa.step()
        ta.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            torch.testing.assert_close(pa.data, pb.data, rtol=2e-6, atol=1e-8)  # an update is ~lr = 1e-2: a few of its ulps
    o = 0
    for pa in a.parameters():  # still views of the flat buffer, in order
        assert pa.data.data_ptr() == fa.flat.data_ptr() + 4 * o
        o += pa.numel()
    assert set(a.state_dict()) == set(b.state_dict())


def test_bench_two_rank_control_flow_rehearsal(hip):
    """bench.py's N > 1 path (launcher env, sharded radargrams, flat-gradient exchange every step, barrier + max-over-ranks timing,
    one JSON line from rank 0) run as TWO ranks -- on this box's one GPU over gloo (CRW_DIST_REHEARSAL: RCCL refuses two ranks on
    one device; the RCCL collective itself is test_flat_gradient_all_reduce_on_rccl).  Child processes, 2 ranks on the card."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CRW_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--no-events"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"].startswith("dp2") and "cpu_baseline" not in d
    # whole-job aggregate: both ranks' columns over the slower rank's time
    assert abs(d["value"] - 2 * d["config"]["columns_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert np.isfinite(d["config"]["loss"])


def test_train_entrypoint_two_steps_vs_oracle(hip, tmp_path):
    """`scripts/train.py main(args)` (the entrypoint north_star keeps; reference scripts/train.py:39-93) for two steps at
    BASELINE config 2 (512x1024 radargram, T = 16, 16x16 patches, overlap (8,0), batch 8, tau 0.01, Adam 1e-3): the two step
    losses against the oracle run on the very same two batches with the same seeded weights and Adam, and the saved
    checkpoint's keys / values."""
    import importlib.util
    from conftest import PKG
    import dataset as crw_dataset
    import encoder as crw_encoder
    spec = importlib.util.spec_from_file_location("crw_train_entry", os.path.join(PKG, "scripts", "train.py"))
    train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train)
    ckpt = str(tmp_path / "enc.pt")
    args = train.get_args_parser().parse_args(["--model", "0", "--synthetic", "512", "1024", "--seq_length", "16",
                                               "--steps", "2", "--epochs", "1", "--save", ckpt])
    args.overlap = tuple(args.overlap)
    step_losses = []
    torch.manual_seed(11)                       # what the module does at import (scripts/train.py:15)
    epoch_means = train.main(args, on_step=lambda i, l: step_losses.append(l))
    got = [float(l) for l in step_losses]
    assert len(got) == 2 and abs(np.mean(got) - epoch_means[0]) < 1e-6
    # the same two batches, weights and optimizer on the oracle
    ds = crw_dataset.RGDataset.synthetic(512, 1024, 16, (16, 16), (8, 0))
    assert len(ds) == 49 and tuple(ds[0].shape) == (16, 63, 16, 16)
    order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(11)).tolist()
    torch.manual_seed(11)
    ref_enc = crw_encoder.CNN(False)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in ref_enc.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    want = []
    for step in range(2):
        batch = torch.stack([ds[i] for i in order[8 * step:8 * step + 8]])
        opt.zero_grad()
        loss, _, _ = orc.crw_forward_torch(batch, sd, 0.01)
        loss.backward()
        opt.step()
        want.append(loss.item())
    assert abs(got[0] - want[0]) <= 1e-4 * abs(want[0]), (got, want)
    assert abs(got[1] - want[1]) <= 1e-3 * abs(want[1]), (got, want)   # after one Adam step of fp32 vs fp32-grade gradients
    saved = torch.load(ckpt)
    assert list(saved.keys()) == list(sd.keys())          # encoder.state_dict() (scripts/train.py:92)
    for k in sd:
        upd_ref = sd[k].detach() - ref_enc.state_dict()[k]
        upd = saved[k].cpu() - ref_enc.state_dict()[k]
        assert (upd - upd_ref).norm() <= 0.1 * upd_ref.norm() + 1e-9, k


def _cnn_labelprop_audit(hip, H, W, T, cfg, M=4, seed=9, train_steps=0):
    """Label propagation with the real CNN encoder at 32x32 patches, overlap (24,0) (BASELINE config 5 geometry): the
    HIP pipeline (tiled conv trunk -> normalise -> top-k -> gather) against (a) the fp32 oracle fed with the SAME
    normalised features, free-running, and (b) the fp64 teacher-forced tie audit (oracle.labelprop_tie_audit):
    every frame is recomputed in fp64 from the device's own context labels, and every label on which the device and
    the fp64 argmax disagree must be a near-tie (class-probability margin or top-k boundary logit gap < 1e-5).
    Integer outputs are exact up to such ties; the count of ties is printed, the count of non-ties must be 0."""
    import dataset as crw_dataset
    import encoder as crw_encoder
    import utils as crw_utils
    from imported.labelprop import LabelPropVOS_CRW
    ds = crw_dataset.RGDataset.synthetic(H, W, T, (32, 32), (24, 0), seed=seed)
    seq = ds[0]                                  # [T, N, 32, 32]
    N = seq.shape[1]
    rows = N * 8 + 24
    seg = (torch.arange(rows)[:, None] * M // rows).float().repeat(1, 32)
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False).cuda()
    if train_steps:  # a TRAINED encoder: the cycle loss pulls the nodes' features apart (affinities no longer near-uniform)
        import model as crw_model
        import dist as crw_dist
        import optim as crw_optim
        net = crw_model.CRW(enc, 0.01, False).cuda()
        net.train(True)
        bucket = crw_dist.FlatGradBucket(net.parameters(), lazy=True)
        opt = crw_optim.FlatAdam(bucket, lr=1e-3)
        items = crw_dataset.RGDataset.synthetic(H, 2048, 16, (32, 32), (24, 0), seed=seed + 1)
        batch = torch.stack([items[i] for i in range(0, len(items), max(1, len(items) // 4))][:4]).cuda()
        for _ in range(train_steps):
            bucket.zero()
            loss, _ = net(batch)
            loss.backward()
            bucket.all_reduce_mean()
            opt.step()
    enc.eval()
    lp = LabelPropVOS_CRW(cfg)
    pred, xent, _ = crw_utils.propagate(seq.cuda(), seg.cuda(), enc, lp, M, False, False)
    assert pred.shape == (N, T)
    with torch.no_grad():
        emb = enc(seq.cuda().reshape(-1, 32, 32).unsqueeze(1)).reshape(T, N, -1).float().contiguous()
    feats = hip.normalize(emb)
    seed_lab = crw_utils.seed_labels(seg.cuda(), N)
    pred2, L = LabelPropVOS_CRW(cfg).propagate_all(feats, seed_lab, M)
    assert torch.equal(pred, pred2)              # utils.propagate == the two kernels on the same features
    # encoder features vs the same network in fp64 on the CPU (what feeds the label propagation)
    enc64 = crw_encoder.CNN(False).double()
    enc64.load_state_dict({k: v.double().cpu() for k, v in enc.state_dict().items()})
    enc64.hip_convs = None
    with torch.no_grad():
        emb64 = enc64(seq[:4].reshape(-1, 32, 32).unsqueeze(1).double()).reshape(4, N, -1)
    torch.testing.assert_close(emb[:4].cpu().double(), emb64, rtol=2e-4, atol=2e-5)
    fe = feats.cpu().numpy()
    audit = orc.labelprop_tie_audit(fe, L.cpu().numpy(), pred.cpu().numpy(), cfg["CXT_SIZE"], cfg["RADIUS"],
                                    cfg["TEMP"], cfg["KNN"], eps=1e-5)
    ref = orc.labelprop(fe, seed_lab.cpu().numpy(), M, cfg["CXT_SIZE"], cfg["RADIUS"], cfg["TEMP"], cfg["KNN"])
    free = int((pred.cpu().numpy() != ref).sum())
    print(f"[{T},{N}] free-running mismatches vs fp32 oracle: {free} of {ref.size}; teacher-forced fp64 audit: {audit}")
    assert audit["not_ties"] == 0, audit
    assert audit["max_soft_err"] <= 1e-4, audit
    # the free-running maps can only differ downstream of a tie: no tie at all -> identical maps
    if audit["step_mismatches"] == 0 and audit["boundary_ties"] == 0:
        assert free == 0
    return free, audit


def test_propagate_with_cnn_at_32x32_patches_vs_oracle(hip):
    """BASELINE config 5 in small: [T,N] = [12,14]."""
    _cnn_labelprop_audit(hip, 130, 32 * 12, 12, dict(CXT_SIZE=5, RADIUS=4, TEMP=0.1, KNN=5))


def test_propagate_with_cnn_at_mcords_size_vs_oracle(hip):
    """BASELINE config 5 at full size: 410x8192 radargram -> [T,N] = [256,48], CXT_SIZE 80 (truncation quirk Q7),
    RADIUS 10, TEMP 0.1, KNN 20 (scripts/test/test_all.py:27-30), random-init CNN (near-uniform affinities = the
    hardest case for ties)."""
    _cnn_labelprop_audit(hip, 410, 8192, 256, dict(CXT_SIZE=80, RADIUS=10, TEMP=0.1, KNN=20), seed=11)


def test_propagate_with_trained_cnn_at_mcords_size_is_exact(hip):
    """BASELINE config 5 at full size again, with an encoder that has TRAINED (Adam steps on the cycle loss: the nodes' features
    separate -- 300 steps: 38 boundary ties left of the 10 271 of the random-init encoder --, the top-k boundaries and class margins move away from the 1e-7 rounding level): the free-running HIP label map
    must equal the oracle's outright -- no appeal to ties."""
    free, audit = _cnn_labelprop_audit(hip, 410, 8192, 256, dict(CXT_SIZE=80, RADIUS=10, TEMP=0.1, KNN=20), seed=11, train_steps=int(os.environ.get('CRW_TEST_TRAIN_STEPS', '300')))
    assert free == 0, (free, audit)


def test_shared_column_encoding_two_radargrams(hip):
    """forward_columns with B = 2 radargrams and stride 3: loss == item-wise loss on the same 2 x 3 windows."""
    import model as crw_model
    import encoder as crw_encoder
    import dataset as crw_dataset
    T = 6
    cols, items = [], []
    for seed in (21, 22):
        ds = crw_dataset.RGDataset.synthetic(64, 16 * 13, T, (16, 16), (8, 0), seed=seed)  # 13 columns -> windows at 0, 3, 6
        cols.append(ds.columns())
        items += [ds[i] for i in range(0, len(ds), 3)]
    torch.manual_seed(3)
    net = crw_model.CRW(crw_encoder.CNN(False), 0.05, False).cuda()
    l_cols, A = net.forward_columns(torch.stack(cols).cuda(), T, 3)
    l_items, _ = net(torch.stack(items).cuda())
    assert A.shape[:2] == (2, 12) and len(items) == 6
    assert abs(l_cols.item() - l_items.item()) <= 1e-6 * abs(l_items.item())


def test_linear_head_kernel_and_model_path(hip):
    """crw_linear128_wgrad (weight gradient of the 128 -> 128 head, split over the patches) against dy^T x in fp64, and the
    model path that uses it (patch count a multiple of 128) against the same model on PyTorch's linear."""
    import torch.nn.functional as TF
    import model as crw_model
    import encoder as crw_encoder
    g = torch.Generator().manual_seed(4)
    x = torch.randn(384, 128, generator=g).cuda()
    w = (torch.randn(128, 128, generator=g) * 0.1).cuda()
    dy = torch.randn(384, 128, generator=g).cuda()
    ref = (dy.double().t() @ x.double()).float()
    got = hip.linear128_wgrad(dy, x)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-4)
    assert torch.equal(got, hip.linear128_wgrad(dy, x))  # fixed summation order
    with pytest.raises(RuntimeError):
        hip.linear128_wgrad(dy[:100], x[:100])
    seq = torch.randn(1, 8, 16, 16, 16, generator=g).cuda()  # 128 patches
    torch.manual_seed(2)
    enc = crw_encoder.CNN(False)
    net = crw_model.CRW(enc, 0.05, False).cuda()
    res = []
    for use_hip_head in (True, False):
        net.zero_grad()
        if not use_hip_head:
            enc._head = enc.fc
        loss, _ = net(seq)
        loss.backward()
        res.append((loss.item(), [p.grad.clone() for p in enc.parameters()]))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0])
    for a_, b_ in zip(res[0][1], res[1][1]):
        torch.testing.assert_close(a_, b_, rtol=1e-3, atol=1e-5 * b_.abs().max().item() + 1e-9)


def test_no_cpu_fallback(hip):
    import model as crw_model
    with pytest.raises(RuntimeError):
        crw_model.affinity(torch.randn(1, 3, 4, 8), 0.1)
