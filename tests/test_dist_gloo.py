"""world_size-2 `gloo` test of the data-parallel plumbing (dist.py): sharding of independent
sequences + one flat-gradient all-reduce (mean) reproduces the single-process gradient.
The loss here is the oracle's torch-CPU walk (the HIP path cannot run without a GPU); what is
under test is the sharding/bucket/collective logic that bench.py and train.py use on RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _grads(enc, seq, tau):
    from oracle import crw_oracle as orc
    sd = dict(enc.named_parameters())
    loss, _, _ = orc.crw_forward_torch(seq, sd, tau)
    loss.backward()
    return loss.detach()


def _worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import dist as crw_dist
    import encoder as crw_encoder
    r, w, _ = crw_dist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    bucket = crw_dist.FlatGradBucket(enc.parameters())
    g = torch.Generator().manual_seed(5)
    seq = torch.randn(4, 4, 5, 16, 16, generator=g)
    mine = crw_dist.shard_indices(4, rank, world)
    bucket.zero()
    loss = _grads(enc, seq[mine], 0.05)
    flat = bucket.all_reduce_mean().clone()
    dist.all_reduce(loss)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), flat=flat.numpy(), loss=(loss / world).numpy())
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    got = np.load(tmp_path / "dp.npz")
    import dist as crw_dist
    import encoder as crw_encoder
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    bucket = crw_dist.FlatGradBucket(enc.parameters())
    g = torch.Generator().manual_seed(5)
    seq = torch.randn(4, 4, 5, 16, 16, generator=g)
    loss = _grads(enc, seq, 0.05)
    np.testing.assert_allclose(got["loss"], loss.numpy(), rtol=1e-5)
    ref = bucket.flat.numpy()
    np.testing.assert_allclose(got["flat"], ref, rtol=1e-3, atol=1e-5 * np.abs(ref).max())


def test_shard_indices_and_bucket_views():
    import dist as crw_dist
    assert crw_dist.shard_indices(8, 1, 4) == [1, 5]
    assert crw_dist.shard_indices(9, 0, 2) == [0, 2, 4, 6]
    lin = torch.nn.Linear(3, 2)
    b = crw_dist.FlatGradBucket(lin.parameters())
    assert b.flat.numel() == 8
    lin(torch.ones(1, 3)).sum().backward()
    assert torch.equal(b.flat[:6].view(2, 3), lin.weight.grad) and b.flat[6:].tolist() == [1.0, 1.0]
    assert lin.weight.grad.data_ptr() == b.flat.data_ptr()
    b.zero()
    assert lin.weight.grad.abs().sum() == 0


def test_lazy_bucket_gathers_fresh_gradients():
    """lazy=True: zero() drops the gradients, backward produces fresh tensors (no accumulation kernels),
    all_reduce_mean() gathers them into the flat buffer and re-binds p.grad to its views; a second step starts clean."""
    import dist as crw_dist
    lin = torch.nn.Linear(3, 2)
    b = crw_dist.FlatGradBucket(lin.parameters(), lazy=True)
    for scale in (1.0, 3.0):
        b.zero()
        assert lin.weight.grad is None and lin.bias.grad is None
        (lin(torch.ones(1, 3)).sum() * scale).backward()
        assert lin.weight.grad.data_ptr() != b.flat.data_ptr()
        flat = b.all_reduce_mean()
        assert flat[:6].tolist() == [scale] * 6 and flat[6:].tolist() == [scale] * 2
        assert lin.weight.grad.data_ptr() == b.flat.data_ptr() and torch.equal(lin.weight.grad, flat[:6].view(2, 3))
    b.zero()
    lin.weight.sum().backward()  # bias gets no gradient this time
    flat = b.all_reduce_mean()
    assert flat[:6].tolist() == [1.0] * 6 and flat[6:].tolist() == [0.0, 0.0]
