"""Where the HOST spends its time while enqueueing a training step (no synchronisation inside the profiled region): cProfile over
`steps` steps of the bench workload.  usage: python tools/r04_host_profile.py [model=1] [steps=40]"""
import cProfile
import os
import pstats
import sys

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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
crw_hip.lib()
torch.manual_seed(11)
enc = crw_utils.create_model(model_id, False)
net = crw_model.CRW(enc, bench.TAU, False).cuda().train(True)
bucket = crw_dist.FlatGradBucket(net.parameters(), lazy=True)
opt = crw_optim.FlatAdam(bucket, lr=1e-3)
seq = bench.make_batch(0, torch.device("cuda"))


def step():
    bucket.zero()
    loss, _ = net(seq)
    loss.backward()
    bucket.all_reduce_mean()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
