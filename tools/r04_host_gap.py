"""Is the training step launch-bound?  Host time to ENQUEUE a step (no synchronisation) against the GPU's time per step, for the CNN
and the Resnet encoder at the bench workload.  usage: python tools/r04_host_gap.py [model=1] [steps=50]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
import crw_hip  # noqa: E402
import dist as crw_dist  # noqa: E402
import model as crw_model  # noqa: E402
import optim as crw_optim  # noqa: E402
import utils as crw_utils  # noqa: E402

model_id = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
crw_hip.lib()
torch.manual_seed(11)
enc = crw_utils.create_model(model_id, False)
net = crw_model.CRW(enc, bench.TAU, False).cuda().train(True)
bucket = crw_dist.FlatGradBucket(net.parameters(), lazy=True)
opt = crw_optim.FlatAdam(bucket, lr=1e-3)
seq = bench.make_batch(0, torch.device("cuda"))


def step(mark=None):
    bucket.zero()
    loss, _ = net(seq)
    if mark is not None:
        mark.append(time.perf_counter())
    loss.backward()
    bucket.all_reduce_mean()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
fw = []
for _ in range(steps):
    ts = time.perf_counter()
    m = []
    step(m)
    fw.append((m[0] - ts, time.perf_counter() - m[0]))
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"model {model_id}: host enqueue {t_enq / steps * 1e3:.3f} ms/step (forward {sum(f for f, _ in fw) / steps * 1e3:.3f}, backward + optimizer "
      f"{sum(b for _, b in fw) / steps * 1e3:.3f}); enqueue + drain {t_all / steps * 1e3:.3f} ms/step")
# one step at a time: the GPU's own time for a step whose launches are all queued behind a long dummy kernel
big = torch.empty(1 << 28, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(10):
    torch.cuda.synchronize()
    for _ in range(6):
        big.mul_(1.0)  # ~1 GB each: the queue fills while these run
    e0.record()
    step()
    e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print(f"model {model_id}: GPU time of a step queued behind other work (no host gaps possible): {tot / 10:.3f} ms")
