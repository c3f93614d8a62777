"""CPU-side tests: host logic, class surface, C-ABI library loads and exports what the header declares."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG, load_golden
from oracle import crw_oracle as orc


def test_library_exports_every_declared_symbol():
    import crw_hip
    if not os.path.exists(crw_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = ctypes.CDLL(crw_hip.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "crw_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(crw_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in crw_hip.h but not exported"
    assert declared == set(crw_hip.SIGNATURES), declared ^ set(crw_hip.SIGNATURES)
    lib = crw_hip.lib()
    assert lib.crw_abi_version() == crw_hip.ABI_VERSION and lib.crw_build_arch() == b"gfx950"
    # pure host-side geometry queries
    assert [lib.crw_padded_nodes(n, 0) for n in (1, 7, 63, 64, 65, 128, 129, 497, 1024, 1025, 4096)] == \
        [32, 32, 64, 64, 96, 128, 192, 512, 1024, 1152, 4096]
    assert [lib.crw_padded_nodes(n, 1) for n in (7, 63, 128, 129, 4096)] == [128, 128, 128, 256, 4096]
    assert lib.crw_walk_state_bytes(1, 2, 8, 0) == 256
    s1, s2 = lib.crw_walk_state_bytes(1, 32, 63, 0), lib.crw_walk_state_bytes(2, 32, 63, 0)
    assert 0 < s1 < s2 <= 2 * s1
    assert lib.crw_walk_state_bytes(1, 8, 200, 2) > lib.crw_walk_state_bytes(1, 8, 200, 1) > 0


def test_build_hook_with_and_without_a_prebuilt_library():
    """`__graft_entry__.build()` is the driver's "does it build" check: it must succeed on a tree that already holds the
    library and on one that does not (fresh clone: *.so is git-ignored), and the ABI number it checks is the header's."""
    import shutil
    import __graft_entry__
    import crw_hip
    header = open(os.path.join(ROOT, "include", "crw_hip.h")).read()
    assert int(re.search(r"^#define\s+CRW_ABI_VERSION\s+(\d+)", header, re.M).group(1)) == crw_hip.ABI_VERSION
    __graft_entry__.build()                       # library present (or built now)
    assert os.path.exists(crw_hip.LIB_PATH)
    aside = crw_hip.LIB_PATH + ".aside"
    shutil.move(crw_hip.LIB_PATH, aside)
    try:
        __graft_entry__.build()                   # library absent: make links / compiles it again
        assert os.path.exists(crw_hip.LIB_PATH)
        assert ctypes.CDLL(crw_hip.LIB_PATH).crw_abi_version() == crw_hip.ABI_VERSION
    finally:
        if not os.path.exists(crw_hip.LIB_PATH):
            shutil.move(aside, crw_hip.LIB_PATH)
        elif os.path.exists(aside):
            os.remove(aside)


def test_cpu_tensors_are_refused():
    import crw_hip
    import model as crw_model
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        crw_model.affinity(torch.randn(1, 3, 4, 8), 0.1)
    with pytest.raises(RuntimeError):
        crw_hip.normalize(torch.randn(4, 8))


def test_cnn_surface_and_seeded_init_match_reference():
    import encoder as crw_encoder
    w = load_golden("cnn_weights_seed11")
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    sd = enc.state_dict()
    assert list(sd.keys()) == list(w.keys())
    assert sum(p.numel() for p in enc.parameters()) == 263088
    for k in w:
        assert np.array_equal(sd[k].numpy(), w[k]), k
    wp = load_golden("cnn_weights_posembed_seed21")
    torch.manual_seed(21)
    encp = crw_encoder.CNN(True)
    for k in wp:
        assert np.array_equal(encp.state_dict()[k].numpy(), wp[k]), k
    g = load_golden("cnn_cfg1_B2T8N7")
    B, T, N, h, w_ = g["seq"].shape
    with torch.no_grad():
        out = enc(torch.tensor(g["seq"]).reshape(-1, 1, h, w_))
    np.testing.assert_allclose(out.reshape(B, T, N, -1).numpy(), g["emb"], rtol=1e-4, atol=1e-5)


def test_resnet_surface_matches_reference():
    import encoder as crw_encoder
    g = load_golden("resnet_seed11")
    torch.manual_seed(11)
    net = crw_encoder.Resnet(False)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert sum(p.numel() for p in net.parameters()) == 4971468
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    np.testing.assert_allclose(sums, g["sums"], rtol=1e-6, atol=1e-6)
    net.eval()
    with torch.no_grad():
        out = net(torch.tensor(g["x"]))
    np.testing.assert_allclose(out.numpy(), g["y_eval"], rtol=1e-4, atol=1e-5)


def test_pos_embed_and_ndiag():
    import utils as crw_utils
    x = torch.randn(5, 1, 6, 4)
    y = crw_utils.pos_embed(x)
    assert y.shape == (5, 2, 6, 4)
    assert torch.equal(y[:, 1], x[:, 0])
    assert torch.allclose(y[0, 0, :, 0], torch.arange(6) / 6 - 0.5)
    assert torch.equal(y, orc.pos_embed(x))
    assert torch.equal(crw_utils.ndiag_matrix(5, 1), torch.eye(5))
    m = crw_utils.ndiag_matrix(5, 3)
    assert torch.allclose(m.sum(1), torch.ones(5)) and m[0, 1] > 0 and m[0, 2] == 0
    seg = torch.tensor(load_golden("labelprop_trunc_T14N10")["seg_ref"])
    assert np.array_equal(crw_utils.seed_labels(seg, 10).numpy(), orc.seed_labels(seg.numpy(), 10))


@pytest.mark.parametrize("name", ["dataset_64x256", "dataset_50x200_ow"])
def test_dataset_matches_reference(name):
    import dataset as crw_dataset
    g = load_golden(name)
    dim, ov, L = tuple(int(x) for x in g["dim"]), tuple(int(x) for x in g["overlap"]), int(g["length"])
    ds = crw_dataset.RGDataset.from_tensor(torch.tensor(g["rg"]), L, dim, ov)
    assert len(ds) == int(g["n_items"])
    for i, idx in enumerate(g["picks"]):
        item = ds[int(idx)]
        assert item.dtype == torch.float32 and np.array_equal(item.numpy(), g["items"][i])
    short = ds.get_smaller_item(0, 2)
    assert short.shape[0] == 2 and ds.pxw == 2 * dim[1] - ov[1]


def test_create_dataset_synthetic_and_model_factory():
    import utils as crw_utils
    import encoder as crw_encoder
    full = crw_utils.create_dataset(0, 8, (16, 16), (8, 0), full=True, synthetic=(64, 256))
    sub = crw_utils.create_dataset(0, 8, (16, 16), (8, 0), full=False, synthetic=(64, 256))
    assert len(full) == 9 and len(sub) == 2 and full[0].shape == (8, 7, 16, 16)
    assert isinstance(crw_utils.create_model(0, False), crw_encoder.CNN)
    assert isinstance(crw_utils.create_model(1, True), crw_encoder.Resnet)


def test_crw_constructor_surface():
    import model as crw_model
    enc = torch.nn.Identity()
    m = crw_model.CRW(enc, 0.01, False)
    assert (m.encoder, m.tau, m.pos_embed, m.only_a) == (enc, 0.01, False, False)
    assert list(m.parameters()) == []
    assert crw_model.CRW(enc, 0.1, True, only_a=True).only_a


def test_labelprop_config_surface():
    from imported.labelprop import LabelPropVOS_CRW
    lp = LabelPropVOS_CRW(dict(CXT_SIZE=4, RADIUS=3, TEMP=0.1, KNN=5))
    assert (lp.cxt_size, lp.radius, lp.temperature, lp.topk, lp.mask, lp.mask_hw) == (4, 3, 0.1, 5, None, None)
    assert lp.context_index(0, 6) == [0, 2, 3, 4, 5]
    m = lp._band(6, 1, torch.device("cpu"))
    assert np.array_equal(m[0].numpy(), orc.band_bias(6, 3))


def test_dataset_columns_are_the_items():
    """RGDataset.columns()[i : i + length] == item i (the shared-encoder feed)."""
    import dataset as crw_dataset
    for overlap in ((8, 0), (8, 4)):
        ds = crw_dataset.RGDataset.synthetic(64, 200, 5, (16, 16), overlap, seed=3)
        cols = ds.columns()
        assert cols.shape[0] == len(ds) + 4 and cols.shape[1:] == ds[0].shape[1:]
        for i in range(len(ds)):
            assert torch.equal(cols[i:i + 5], ds[i])
        assert torch.equal(ds.columns(2, 6), cols[2:8])
        with pytest.raises(IndexError):
            ds.columns(3, cols.shape[0])


def test_pelt_rbf_restatement():
    """pelt.py (restated PELT + RBF kernel cost; parity with `ruptures` is unpinned): a clean mean shift is found
    at a multiple of `jump`, a flat signal has no interior breakpoint, and utils.change_point applies the
    reference's `result[-2] + 5` / None-on-failure convention (src/utils.py:125-132)."""
    import pelt
    import utils as crw_utils
    rng = np.random.default_rng(0)
    sig = np.concatenate([rng.normal(0, 0.1, 40), rng.normal(3, 0.1, 35)])
    bk = pelt.pelt_rbf(sig, pen=5)
    assert bk[-1] == len(sig) and 40 in bk
    assert pelt.pelt_rbf(np.zeros(50), pen=5) == [50]
    g = pelt.rbf_gram(sig)
    assert np.allclose(np.diagonal(g), 1.0) and np.allclose(g, g.T) and g.min() >= np.exp(-100.0)
    # xent [N, T-1] whose column-to-column change jumps at column 40: change_idx = breakpoint + 5
    xent = torch.zeros(4, 77)
    xent[:, 41:] = torch.arange(36).float().repeat(4, 1) % 2 * 3.0
    ci = crw_utils.change_point(xent)
    assert ci is not None and ci >= 5
    assert crw_utils.change_point(torch.zeros(4, 30)) is None


def test_native_pelt_equals_the_numpy_restatement():
    """`crw_pelt_rbf` (csrc/pelt.cpp, a HOST function of the library: what `utils.change_point` calls) repeats pelt.py's sums in
    pelt.py's order: the same breakpoints on random, shifted and tie-ridden signals of every length the dataset can produce,
    with the median heuristic and with a given bandwidth; degenerate inputs behave alike."""
    import pelt
    import crw_hip
    rng = np.random.default_rng(3)
    for trial in range(600):
        n = int(rng.integers(3, 300))
        kind = trial % 4
        if kind == 0:
            x = rng.random(n)
        elif kind == 1:
            x = np.abs(rng.standard_normal(n)) * rng.random()
        elif kind == 2:
            x = rng.standard_normal(n) * 0.2
            x[int(rng.integers(1, n)):] += rng.random() * 3
        else:
            x = np.round(rng.random(n) * 4) / 4  # repeated values: zero distances, exact ties between segmentations
        for pen, gamma in ((5, None), (1.0, None), (5, 2.5)):
            assert crw_hip.pelt_rbf(x, pen=pen, gamma=gamma) == pelt.pelt_rbf(x, pen=pen, gamma=gamma), (trial, n, pen, gamma)
    for x in (np.zeros(50), [1.0], [1.0, 2.0], [1.0, 2.0, 3.0], np.arange(11.0)):
        assert crw_hip.pelt_rbf(x, pen=5) == pelt.pelt_rbf(x, pen=5)
    with pytest.raises(crw_hip.CrwError):
        crw_hip.pelt_rbf([], pen=5)


def test_pelt_pruning_keeps_the_exact_optimum():
    """PELT's defining property (Killick et al. 2012): with pruning it still returns the exact minimiser of
    sum of segment costs + pen * (number of segments) over the admissible grid (breakpoints at multiples of `jump`, segments of at
    least `min_size`).  Checked against the unpruned dynamic programme on random multi-regime signals and several penalties."""
    import pelt
    rng = np.random.default_rng(7)

    def optimum(sig, pen, min_size=2, jump=5):
        n = len(sig)
        gram = pelt.rbf_gram(sig)

        def cost(a, b):
            blk = gram[a:b, a:b]
            return np.trace(blk) - blk.sum() / (b - a)
        ends = [k for k in range(0, n, jump) if k >= min_size] + [n]
        best = {0: (0.0, ())}
        for e in ends:
            cands = [(best[t][0] + cost(t, e) + pen, best[t][1] + (e,)) for t in best if e - t >= min_size and t < e]
            if cands:
                best[e] = min(cands, key=lambda c: c[0])
        return best[n]

    def total(sig, bks, pen):
        gram = pelt.rbf_gram(sig)
        c, a = 0.0, 0
        for b in bks:
            blk = gram[a:b, a:b]
            c += np.trace(blk) - blk.sum() / (b - a) + pen
            a = b
        return c

    for trial in range(6):
        levels = rng.normal(0, 2.0, size=rng.integers(1, 5))
        sig = np.concatenate([rng.normal(m, 0.3, size=rng.integers(8, 40)) for m in levels])
        for pen in (0.5, 5.0, 20.0):
            got = pelt.pelt_rbf(sig, pen=pen)
            want_cost, want_bks = optimum(sig, pen)
            assert got[-1] == len(sig) and all(b % 5 == 0 for b in got[:-1])
            assert abs(total(sig, got, pen) - want_cost) <= 1e-9 * max(1.0, abs(want_cost)), (trial, pen, got, want_bks)


SEGMENT_CASES = ["segment_ds0_reverse", "segment_ds1_reverse", "segment_ds3_reverse", "segment_ds0_correction",
                 "segment_ds3_correction_reverse"]


def run_segment_golden(g, propagate_fn, device):
    """Drive ``inference.segment`` on the inputs of a ``segment_*`` fixture (written by the reference's own
    scripts/test/test_all.py main(args)); ``propagate_fn`` stands in for ``utils.propagate`` (the CPU tests put the
    oracle there, the GPU tests the real HIP path) and is wrapped to force the fixture's change indices, exactly
    like the generator wrapped the reference's ``propagate``."""
    import dataset as crw_dataset
    import inference as crw_inference
    from imported.labelprop import LabelPropVOS_CRW
    T, patch, overlap = int(g["T"]), tuple(int(v) for v in g["patch"]), tuple(int(v) for v in g["overlap"])
    ds = crw_dataset.RGDataset.from_tensor(torch.tensor(g["rg"]), T, patch, overlap)
    N = ds[0].shape[1]
    seg = torch.tensor(g["seg"])[:N * patch[0]]           # get_reference(h = N*H) (test_all.py:60)
    forced = [None if f < 0 else int(f) for f in g["forced_change"]]
    calls = {"n": 0}

    def propagate(seq, seg_ref, model, lp, ncls, do_pos_embed, use_last):
        pred, xent, change = propagate_fn(seq, seg_ref, model, lp, ncls, do_pos_embed, use_last)
        i = calls["n"]
        calls["n"] += 1
        if i < len(forced):
            change = forced[i]
        return pred, xent, change

    class Flatten(torch.nn.Module):
        def forward(self, x):
            return x.flatten(1)

    cfg = dict(CXT_SIZE=int(g["cxt_size"]), RADIUS=int(g["radius"]), TEMP=float(g["temp"]), KNN=int(g["knn"]))
    orig = crw_inference.propagate
    crw_inference.propagate = propagate
    try:
        out = crw_inference.segment(ds, seg, Flatten(), LabelPropVOS_CRW(cfg), int(g["nclasses"]), T, patch, overlap,
                                    correction=bool(g["correction"]), use_last=bool(g["use_last"]),
                                    dataset_id=int(g["dataset_id"]), device=device)
    finally:
        crw_inference.propagate = orig
    return out


@pytest.mark.parametrize("name", SEGMENT_CASES)
def test_segment_driver_matches_reference_main(name):
    """Host logic of the whole-radargram pipeline (forward, correction with the reference's `get_smaller_item`
    semantics, reverse pass, per-dataset merge rules) against maps produced by the reference's scripts/test/test_all.py
    main(args); label propagation itself is the CPU oracle here (the GPU twin of this test runs the HIP kernels)."""
    g = load_golden(name)

    def oracle_propagate(seq, seg_ref, model, lp, ncls, do_pos_embed, use_last):
        T, N = seq.shape[:2]
        emb = model(seq.reshape(T * N, 1, *seq.shape[2:])).reshape(T, N, -1).numpy()
        if use_last:
            emb = emb[::-1].copy()
        pred = orc.labelprop(emb, orc.seed_labels(seg_ref.numpy(), N), ncls, lp.cxt_size, lp.radius, lp.temperature, lp.topk)
        return torch.tensor(pred), torch.tensor(orc.xent_metric(emb)), None

    out = run_segment_golden(g, oracle_propagate, "cpu")
    assert np.array_equal(out["forward"].numpy().astype(np.int8), g["saved_map"])
    assert np.array_equal(out["pred"].numpy().astype(np.int8), g["final_map"])
    if bool(g["correction"]):
        assert any(f >= 0 for f in g["forced_change"])


@pytest.mark.parametrize("status,raises", [(1, False), (2, True), (3, True)])
def test_correction_step_error_policy(status, raises):
    """A correction that fails on its DATA is skipped like the reference's bare `except` does (scripts/test/test_all.py:113-122) --
    that includes CRW_EINVAL from the C ABI (a degenerate correction window) and ordinary Python errors; a failure of the HIP path
    itself (CRW_EWORKSPACE / CRW_EHIP, typed `crw_hip.CrwError`) aborts the segmentation."""
    import crw_hip
    g = load_golden("segment_ds0_correction")
    calls = {"n": 0}
    n_items = sum(1 for _ in g["forced_change"])

    def propagate(seq, seg_ref, model, lp, ncls, do_pos_embed, use_last):
        T, N = seq.shape[:2]
        calls["n"] += 1
        if calls["n"] > n_items:  # the correction calls come after one forward call per item
            raise crw_hip.CrwError("crw_labelprop_topk", status, 0)
        return torch.zeros(N, T), torch.zeros(N, T - 1), None

    if raises:
        with pytest.raises(crw_hip.CrwError) as e:
            run_segment_golden(g, propagate, "cpu")
        assert e.value.status == status and e.value.device_failure
    else:
        out = run_segment_golden(g, propagate, "cpu")
        assert calls["n"] > n_items and not out["forward"].any()  # the correction ran, failed on its data and was skipped

    def broken(seq, seg_ref, model, lp, ncls, do_pos_embed, use_last):
        calls["n"] += 1
        if calls["n"] > n_items:
            raise IndexError("a data error inside the correction")
        return torch.zeros(seq.shape[1], seq.shape[0]), torch.zeros(seq.shape[1], seq.shape[0] - 1), None

    calls["n"] = 0
    run_segment_golden(g, broken, "cpu")


RESNET_TRAIN_CASES = ["resnet_train_B2T4N5", "resnet_train_32x32_B2T4N5", "resnet_train_20x27_B2T4N5"]


@pytest.mark.parametrize("name", RESNET_TRAIN_CASES)
def test_resnet_train_mode_matches_reference_on_cpu(name):
    """`Resnet` (the reference's default encoder) in train mode: features, BatchNorm running statistics and -- through the
    oracle's walk -- loss and parameter gradients against the reference's CRW forward/backward (fixture resnet_train_*: 16x16
    patches; 32x32 with overlap 24 -- scripts/test/test_mc1.py:19,21, layer4's map 2x2 --; 20x27); then, where the fixture has it,
    the encoder switched to eval mode after the step (scripts/test/test.py:42) against the reference's eval features."""
    import encoder as crw_encoder
    g = load_golden(name)
    torch.manual_seed(int(g["seed"]))
    enc = crw_encoder.Resnet(False)
    enc.train(True)
    seq = torch.tensor(g["seq"])
    B, T, N, h, w = seq.shape
    emb = enc(seq.reshape(-1, h, w).unsqueeze(1))
    np.testing.assert_allclose(emb.detach().numpy(), g["emb"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(enc.bn0.running_mean.numpy(), g["bn0.running_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(enc.bn0.running_var.numpy(), g["bn0.running_var"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(enc.model.bn1.running_mean.numpy(), g["model.bn1.running_mean"], rtol=1e-4, atol=1e-6)
    loss, A = orc.walk_loss_torch(emb.reshape(B, T, N, -1), float(g["tau"]))
    assert abs(loss.item() - float(g["loss"])) <= 1e-5
    np.testing.assert_allclose(A.detach().numpy(), g["A"], rtol=1e-3, atol=1e-3)
    loss.backward()
    names = [k for k, _ in enc.named_parameters()]
    assert names == list(g["grad_names"])
    for k, p in enc.named_parameters():
        if "grad." + k in g:
            ref = g["grad." + k]
            np.testing.assert_allclose(p.grad.numpy(), ref, rtol=5e-3, atol=5e-4 * np.abs(ref).max())
    if "emb_eval" in g:
        enc.train(False)
        with torch.no_grad():
            emb_eval = enc(seq.reshape(-1, h, w).unsqueeze(1))
        np.testing.assert_allclose(emb_eval.numpy(), g["emb_eval"], rtol=1e-4, atol=1e-5)


def test_resnet_hip_coverage_rule():
    """`resnet_hip.supported` (what may take the HIP kernels) as a pure function of module and input: checked with a stand-in for
    a device tensor, no GPU involved."""
    import encoder as crw_encoder
    import resnet_hip

    class X:  # the attributes `supported` reads
        is_cuda, dtype = True, torch.float32

        def __init__(self, *shape):
            self.shape = torch.Size(shape)

        def dim(self):
            return len(self.shape)

    net = crw_encoder.Resnet(False)
    ok = lambda x, n=net: resnet_hip.supported(x, n)
    assert resnet_hip.final_map(16, 16) == (1, 1) and resnet_hip.final_map(32, 32) == (2, 2) and resnet_hip.final_map(20, 27) == (1, 1)
    assert resnet_hip.final_map(64, 64) == (3, 3)
    net.train(True)
    assert ok(X(4, 1, 16, 16)) and ok(X(4, 1, 32, 32)) and ok(X(4, 1, 20, 27)) and ok(X(2, 1, 96, 40))
    assert not ok(X(4, 2, 16, 16))              # fc0 was built for one channel: PyTorch's shape error, not an out-of-bounds read
    assert resnet_hip.supported(X(4, 2, 16, 16), crw_encoder.Resnet(True))
    x64 = X(4, 1, 16, 16)
    x64.dtype = torch.float64
    assert not ok(x64)
    net.train(False)                             # eval mode: the forward only
    with torch.no_grad():
        assert ok(X(4, 1, 16, 16)) and ok(X(4, 1, 32, 32))
    assert not ok(X(4, 1, 16, 16))               # (grad enabled)
    net.train(True)
    net.model.layer3[0].bn2.eval()               # one frozen BatchNorm: the native pass has one mode for all of them
    assert not ok(X(4, 1, 16, 16))
    net.train(True)
    net.model.layer2[0].bn1.eps = 1e-3           # ... and one eps / momentum
    assert not ok(X(4, 1, 16, 16))
    net.model.layer2[0].bn1.eps = net.bn0.eps
    net.bn0.momentum = None                      # cumulative moving average: not implemented
    assert not ok(X(4, 1, 16, 16))
