#!/usr/bin/env python3
"""Benchmark of the CRW training step (BASELINE.json metric: radargram columns/sec, CRW fwd+bwd,
512x4096 sequences).

  python bench.py [--gpus N --steps K --warmup W]        (N > 1: launched by torch.distributed.run)

A "step" = one synthetic 512x4096 radargram per GPU = 8 non-overlapping items [T=32, N=63, 16x16]
(patch 16x16, overlap (8,0), reference defaults) -> encoder fwd, normalise+affinity, dual softmax,
transition chain, loss, full backward to the encoder weight gradients, ONE all-reduce of the flat
gradient (N > 1) and the Adam update.  Inputs are resident in HBM before the timed region.
Rank 0 prints one JSON line.  Weak scaling: each rank processes its own radargram.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "radar-sounder-crw_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s (about 6 TB/s is what streaming kernels reach)
PEAK_HBM_GBS = 8000.0

H_RG, W_RG, T_SEQ, PATCH, OVERLAP, TAU = 512, 4096, 32, (16, 16), (8, 0), 0.01


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", type=int, default=0, help="0 = CNN (BASELINE 'tiny CNN encoder'), 1 = Resnet")
    ap.add_argument("--convs", default="bf16x3", choices=["bf16x3", "mixed", "bf16", "torch"],
                    help="conv2-5 of the CNN: bf16x3 = HIP kernels, hi/lo bf16 operand pairs (fp32-grade, default); mixed = that forward, "
                         "backward on plain bf16 operands; bf16 = HIP kernels, plain bf16 operands; torch = PyTorch-ROCm (MIOpen) fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the per-kernel roofline probes")
    ap.add_argument("--repeats", type=int, default=0,
                    help="the timed region of --steps steps is run this many times; value / ms_per_step = the MEDIAN repeat "
                         "(min / max on the line); every repeat is bracketed by barrier + synchronize like a single one.  0 (default): "
                         "as many repeats as keep the GPU busy for about --gpu-seconds (at least 5), so that an outside sampler of GPU "
                         "activity sees the run")
    ap.add_argument("--gpu-seconds", type=float, default=10.0, help="target duration of all timed repeats together when --repeats is 0")
    ap.add_argument("--no-events", action="store_true",
                    help="do not bracket the conv kernels of the timed steps with HIP events (A/B of their overhead)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline's all-threads leg (a 1-thread step follows)")
    ap.add_argument("--workload", default="radargram", choices=["radargram", "chain", "labelprop", "shared", "dense", "train32"],
                    help="radargram: the BASELINE metric (default); chain: kernel-only stress shape K of SURVEY "
                         "8(d): affinity + walk fwd+bwd on unit-norm random features, no encoder; labelprop: BASELINE "
                         "config 5, MCoRDS-shaped 410x8192 radargram, user-seed label propagation (utils.propagate); "
                         "shared: all 225 overlapping items of the radargram per step, patch-columns encoded once "
                         "(SURVEY 8 f1); dense: shape family D of SURVEY 8(d) = the radargram workload at --overlap 15 0 "
                         "(N = 497 nodes per column)")
    ap.add_argument("--train-steps", type=int, default=0, help="labelprop workload: Adam steps on the cycle loss before the timed passes")
    ap.add_argument("--nodes", type=int, default=4096, help="N of the chain workload")
    ap.add_argument("--walk", type=int, default=32, help="T (frames) of the chain workload")
    ap.add_argument("--modes", default="f32,bf16x3,bf16", help="chain arithmetics the chain workload runs (profiling one at a time)")
    return ap.parse_args()


def make_batch(rank, device):
    import dataset as crw_dataset
    ds = crw_dataset.RGDataset.synthetic(H_RG, W_RG, T_SEQ, PATCH, OVERLAP, seed=11 + rank)
    items = [ds[i] for i in range(0, len(ds), T_SEQ)]  # non-overlapping items of one radargram
    return torch.stack(items).contiguous().to(device)  # [8, 32, 63, 16, 16] ([8, 32, 497, 16, 16] for "dense")


def chain_probe(n, batch, nprob, iters=50):
    """Average duration of one grouped chain-GEMM launch (crw_gemm_f32 = the kernel of
    crw_walk_fwd/bwd) at padded size n, measured with HIP events on the launch stream."""
    import crw_hip
    A = torch.rand(batch * nprob, n, n, device="cuda")
    Bm = torch.rand(batch * nprob, n, n, device="cuda")
    C = torch.empty_like(A)
    for _ in range(3):
        crw_hip.gemm_f32(A, Bm, C)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        crw_hip.gemm_f32(A, Bm, C)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * n ** 3 * batch * nprob
    return ms, flops


def walk_probe(B, T, N, iters=20):
    """Affinity + walk fwd+bwd alone (features resident), fp32 chain, at the radargram workload's shape."""
    import crw_hip
    import model as crw_model
    g = torch.Generator().manual_seed(3)
    emb0 = torch.randn(B, T, N, 128, generator=g).cuda()

    def step():
        emb = emb0.clone().requires_grad_(True)
        crw_model.walk_loss(*crw_model.affinity_with_stats(emb, TAU)).backward()

    for _ in range(3):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    Np = crw_hip.padded_nodes(N)
    return ms, 2.0 * Np ** 3 * (9 * (T - 3) + 3) * B


def chain_probe_bf16(n, batch, split, iters=10):
    """Same probe for the bf16 matrix-core chain GEMM (operands already converted: GEMM time only)."""
    import crw_hip
    A = torch.rand(batch, n, n, device="cuda")
    Bm = torch.rand(batch, n, n, device="cuda")
    C, ws = crw_hip.gemm_bf16(A, Bm, split=split)
    for _ in range(2):
        crw_hip.gemm_bf16(A, Bm, C, split=split, ws=ws, convert=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        crw_hip.gemm_bf16(A, Bm, C, split=split, ws=ws, convert=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * n ** 3 * batch * (3 if split == 3 else 1)


def _rn_pairs(Hin, Win, Hout, Wout, k, stride, pad):
    """number of (output pixel, tap) pairs of a k x k convolution whose tap falls inside the Hin x Win input map"""
    n = 0
    kh, kw = (k >> 8, k & 255) if k >= 256 else (k, k)  # (the head over an hl x wl map records its kernel as hl * 256 + wl)
    for oy in range(Hout):
        for ox in range(Wout):
            for ky in range(kh):
                for kx in range(kw):
                    iy, ix = oy * stride + ky - pad, ox * stride + kx - pad
                    n += 0 <= iy < Hin and 0 <= ix < Win
    return n


def resnet_event_kernels(ev, P, steps):
    """roofline lines of the Resnet matrix-core kernels from the HIP events the native pass records around each of their launches
    in the timed steps (crw_rn_timing_*; ev: geometry key -> list of ms).
    Algorithmic flops of a convolution pass = 2 * P * pairs * cin * cout with pairs = the (output pixel, tap) pairs whose tap lies
    inside the input map (multiplications by padding zeros are not counted: a 3x3 convolution on layer4's 1x1 map is 1 tap)."""
    import crw_hip
    out = []
    for key, pairs in ev.items():
        kind, mode = key[0], key[1]
        if kind == "rn_conv":
            _, _, Hs, Ws, Cs, Hd, Wd, N, k, stride, pad = key
            # algorithmic HBM bytes: the source planes once (hi + lo = 4 B per element) + the fp32 destination once (the fused
            # epilogues of the backward-data products read more -- mask, Z, the other branch's gradient: not counted)
            nbytes = 4.0 * P * (Hs * Ws * Cs + Hd * Wd * N)
            if mode == crw_hip.RN_FWD:
                alg, ex = 2.0 * P * _rn_pairs(Hs, Ws, Hd, Wd, k, stride, pad) * Cs * N, 1.0
                name = f"rn_conv_spec_kernel fwd {Hs}x{Ws}x{Cs} -> {Hd}x{Wd}x{N} k{k}/s{stride}"
            elif mode == crw_hip.RN_BWD:
                alg, ex = 2.0 * P * _rn_pairs(Hd, Wd, Hs, Ws, k, stride, pad) * Cs * N, 1.0
                name = f"rn_conv_spec_kernel bwd-data (+ BatchNorm-backward sums) {Hs}x{Ws}x{Cs} -> {Hd}x{Wd}x{N} k{k}/s{stride}"
            elif mode == crw_hip.RN_STEM_FWD:
                alg, ex = 2.0 * P * Hd * Wd * 147 * 64, 256.0 / 147.0   # K = 8 kernel rows x 32 for the 7 x 7 x 3 = 147 taps
                name = "rn_stem_fwd_kernel 7x7/2 3 -> 64 (patch per wave; rn_conv_spec_kernel at other patch sizes)"
            else:
                alg = 2.0 * P * 81 * 147 * 64
                # executed: 18 map rows x 64 columns x the dZ rows each one sees (Toeplitz planes: zeros are multiplied)
                rows = sum(min(8, (iy + 3) // 2) - max(0, (iy - 2) // 2 if iy > 3 else 0) + 1 for iy in range(Hd))
                ex = (2.0 * P * 64 * rows * 9 * 64) / alg
                name = "rn_stem_bwd_kernel 64 -> 3 (patch per wave; rn_conv_kernel on Toeplitz planes at other patch sizes)"
        else:
            _, _, Hin, Win, Cin, Hout, Wout, Cout, k, stride, pad = key
            nbytes = 4.0 * P * (Hin * Win * Cin + Hout * Wout * Cout)  # both operands' planes once
            if mode == crw_hip.RN_FWD:
                alg, ex = 2.0 * P * _rn_pairs(Hin, Win, Hout, Wout, k, stride, pad) * Cin * Cout, 1.0
                name = f"rn_wgrad_kernel {Hin}x{Win}x{Cin} -> {Hout}x{Wout}x{Cout} k{k}/s{stride} (+ slab sum)"
            else:
                alg, ex = 2.0 * P * Hout * Wout * 147 * 64, 256.0 / 147.0
                name = "rn_stem_wgrad_kernel 7x7/2 (+ slab sum) (patch per wave pair; rn_wgrad_kernel at other patch sizes)"
        kms = sum(pairs) / len(pairs)  # ms of every timed launch (crw_rn_timing_read)
        mfma_frac = alg / (kms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"]
        hbm_gbps = nbytes / (kms * 1e-3) / 1e9
        hbm_frac = hbm_gbps / HBM_PEAK_GBPS
        e = {"kernel": name, "_key": key, "traffic": None, "launch_us": kms * 1e3, "launches_per_step": len(pairs) // steps, "timed_launches": len(pairs),
             "algorithmic_flops_per_launch": alg, "algorithmic_bytes_per_launch": nbytes,
             "mfma_frac": mfma_frac, "mfma_tflops": alg / (kms * 1e-3) / 1e12,
             "mfma_executed_tflops": alg * 3 * ex / (kms * 1e-3) / 1e12,
             "mfma_executed_frac": alg * 3 * ex / (kms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
             "hbm_frac": hbm_frac, "hbm_GBps": hbm_gbps,
             "note": "algorithmic (fp32-equivalent, in-map taps only) flops and HBM bytes / mean HIP-event time of this kernel's launches "
                     "INSIDE the timed steps (the side stream is active: a launch that shares the chip with another one takes longer than "
                     "alone); every product is 3 bf16 MFMAs on hi/lo operand pairs; `bound` = the roof the kernel is closer to"}
        if hbm_frac > mfma_frac:  # the small-K products (1x1 shortcuts, layer1) move more bytes per flop than the matrix roof's ridge
            e.update({"bound": "hbm", "achieved": hbm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_frac})
        else:
            e.update({"bound": "mfma", "achieved": alg / (kms * 1e-3) / 1e12, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "frac": mfma_frac})
        out.append(e)
    out.sort(key=lambda k_: -k_["launch_us"] * k_["launches_per_step"])
    # HBM traffic per launch from the committed in-step PMC passes over this same command (rocprofv3 cannot run inside bench):
    # profiles/r04_pmc_resnet.json, "by_grid" = per (kernel instance, grid size) -- the grid tells the layers of one instance apart
    # (gathered product: blocks = patch tiles x column tiles x groups, 512 threads each).  Constants are attached only while the
    # kernel's live time is compatible with the one recorded with them (see below).
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_resnet.json")))["by_grid"]
        Ppad = (P + 127) // 128 * 128
        for k in out:  # weight gradients: the instance is named by the tile, its layers told apart by their time
            key = k.get("_key")
            if not key or key[0] != "rn_wgrad" or key[1] != crw_hip.RN_FWD:
                continue
            _, _, Hin, Win, Cin, Hout, Wout, Cout, kk, stride, pad = key
            name = f"crw::rn_wgrad_kernel<{128 if Cin % 128 == 0 else 64}, {128 if Cout % 128 == 0 else 64}, 32>"
            red = pmc.get("crw::rn_wgrad_reduce_kernel", {})
            red_us = min((v["us_per_launch"] for v in red.values()), default=12.0)  # the slab sum inside the event bracket (smallest: a lower bound)
            kh, kw = (kk >> 8, kk & 255) if kk >= 256 else (kk, kk)
            slabs = crw_hip.lib().crw_rn_wgrad_ws_bytes(key[1], P, Hin, Win, Cin, Hout, Wout, Cout, kh, kw, stride, pad) // (Cin * Cout * 4)
            tm, tn = (128 if Cin % 128 == 0 else 64), (128 if Cout % 128 == 0 else 64)
            grid = str(slabs * (Cin // tm) * (Cout // tn) * 256)  # slabs = patch slices x live taps; one 256-thread block per slab tile
            rec = pmc.get(name, {}).get(grid)
            if rec is None:
                continue
            if 0.9 * (rec["us_per_launch"] + red_us) <= k["launch_us"] <= 2.2 * (rec["us_per_launch"] + red_us):
                k["traffic"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
                k["traffic_note"] = (f"HBM bytes per launch of the weight-gradient kernel inside the step (its slab sum not included), committed PMC "
                                     f"passes profiles/r04_pmc_resnet.json ({name}, grid {grid}, {rec['us_per_launch']} us): reads 2*1024*FETCH_SIZE = "
                                     f"{rec['hbm_read_bytes_per_launch'] / 1e6:.1f} MB (gfx950 correction), writes 1024*WRITE_SIZE = "
                                     f"{rec['hbm_write_bytes_per_launch'] / 1e6:.1f} MB (the slabs); algorithmic: "
                                     f"{k['algorithmic_bytes_per_launch'] / 1e6:.1f} MB (both operands' planes once)")
        for k in out:
            key = k.get("_key")
            if not key or key[0] != "rn_conv" or key[1] not in (crw_hip.RN_FWD, crw_hip.RN_BWD):
                continue
            _, _, Hs, Ws, Cs, Hd, Wd, N, kk, stride, pad = key
            tn = 128 if N % 128 == 0 else 64
            grid = str((Ppad // 128) * (N // tn) * Hd * Wd * 512)
            cands = [(abs(v[grid]["us_per_launch"] - k["launch_us"]), name, v[grid]) for name, v in pmc.items()
                     if name.startswith(f"crw::rn_conv_spec_kernel<{tn},") and grid in v]
            if not cands:
                continue
            d, name, rec = min(cands)
            # (counter collection serialises the dispatches: the time recorded beside the counters is the kernel ALONE, the live event
            # time includes whatever the side stream runs beside it -- anything from 0.9x to 2.2x of the recorded time is the same kernel)
            if 0.9 * rec["us_per_launch"] <= k["launch_us"] <= 2.2 * rec["us_per_launch"]:
                k["traffic"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
                k["traffic_note"] = (f"HBM bytes per launch inside the step, committed PMC passes profiles/r04_pmc_resnet.json ({name}, grid {grid}: "
                                     f"reads 2*1024*FETCH_SIZE = {rec['hbm_read_bytes_per_launch'] / 1e6:.1f} MB (gfx950 correction), writes "
                                     f"1024*WRITE_SIZE = {rec['hbm_write_bytes_per_launch'] / 1e6:.1f} MB, separate passes); algorithmic: "
                                     f"{k['algorithmic_bytes_per_launch'] / 1e6:.1f} MB (source planes + fp32 destination once; the fused epilogues "
                                     "of the backward-data products read the consuming layer's mask and Z planes on top of that)")
            else:
                k["traffic_note"] = (f"committed PMC constants NOT attached: live {k['launch_us']:.1f} us vs {rec['us_per_launch']:.1f} us recorded "
                                     f"with them ({name}, grid {grid})")
    except Exception:
        pass
    for k in out:
        k.pop("_key", None)
    return out


def host_cpu():
    """CPU model, physical cores and hardware threads of this box (from /proc/cpuinfo), and the share this process may run on"""
    model, cores, threads = "unknown", set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                threads += 1
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:  # blank line = end of one processor's record
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"model": model, "physical_cores": len(cores) or None, "hw_threads": threads or (os.cpu_count() or 1), "threads_available_to_this_job": avail}


def cpu_baseline(budget_s):
    """The oracle (pure torch-CPU restatement, validated against the reference) timed on this
    box's host cores on a bounded sample of the same workload: ONE item [1,32,63,16,16] per step."""
    import numpy as np
    from oracle import crw_oracle as orc
    import encoder as crw_encoder
    import dataset as crw_dataset
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    ds = crw_dataset.RGDataset.synthetic(H_RG, W_RG, T_SEQ, PATCH, OVERLAP, seed=11)
    seq = ds[0][None].contiguous()

    def step():
        for v in sd.values():
            v.grad = None
        loss, _, _ = orc.crw_forward_torch(seq, sd, TAU)
        loss.backward()

    # the box exposes more hardware threads than this job's CPU share: time the step at a few
    # thread counts and report the fastest (cores = threads actually used for that figure)
    cpu = host_cpu()
    avail = cpu["threads_available_to_this_job"]
    candidates = sorted({min(avail, c) for c in (16, 32)})  # all hw threads oversubscribe this job's CPU share
    best = None
    for threads in candidates:
        torch.set_num_threads(threads)
        step()  # warm-up
        times = []
        t_end = time.time() + budget_s / len(candidates)
        while len(times) < 2 or (time.time() < t_end and len(times) < 30):
            t0 = time.time()
            step()
            times.append(time.time() - t0)
        med = float(np.median(times))
        if best is None or med < best[0]:
            best = (med, threads, len(times))
    med, threads, n = best
    cols = T_SEQ * PATCH[1]
    torch.set_num_threads(1)  # the 1-thread figure beside the all-cores one (one step: it takes several seconds)
    t0 = time.time()
    step()
    one = time.time() - t0
    torch.set_num_threads(threads)
    return {"value": cols / med, "unit": "radargram columns/s", "cores": threads, "kind": "port",
            "cpu_model": cpu["model"], "physical_cores": cpu["physical_cores"], "hw_threads": cpu["hw_threads"],
            "threads_available_to_this_job": avail,
            "one_thread": {"value": cols / one, "unit": "radargram columns/s", "cores": 1, "sample": f"the same step once at 1 thread ({one:.2f} s)"},
            "sample": f"1 item [1,{T_SEQ},63,16,16] fwd+bwd per step, median of {n} steps ({med:.3f} s/step) at "
                      f"{threads} threads (fastest of {candidates}; {avail} hw threads visible), "
                      f"torch {torch.__version__} CPU ops"}


def bench_chain(args):
    """Shape family K: features [1, T, N, 128] straight into affinity + walk (fwd+bwd) for each chain
    arithmetic; flops counted = MFMA products actually executed by the prefix-form chain."""
    import crw_hip
    import model as crw_model
    T, N, C = args.walk, args.nodes, 128
    g = torch.Generator().manual_seed(11)
    base = torch.randn(1, 1, N, C, generator=g)
    emb0 = (base + 0.5 * torch.randn(1, T, N, C, generator=g)).cuda()
    nprod = 9 * (T - 3) + 3
    res, ref_loss = {}, None
    for name, chain, mult in (("f32", crw_hip.CHAIN_F32, 1), ("bf16x3", crw_hip.CHAIN_BF16X3, 3),
                              ("bf16", crw_hip.CHAIN_BF16, 1)):
        if name not in args.modes.split(","):
            continue
        Np = crw_hip.padded_nodes(N, chain)

        def step():
            emb = emb0.clone().requires_grad_(True)
            A, stats = crw_model.affinity_with_stats(emb, 0.05)  # what CRW.forward does
            loss = crw_model.walk_loss(A, chain, stats)
            loss.backward()
            return loss

        for _ in range(max(1, args.warmup)):
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        if ref_loss is None:
            ref_loss = loss.item()
        flops = 2.0 * Np ** 3 * nprod * mult
        peak = PEAK_TFLOPS["f32"] if chain == crw_hip.CHAIN_F32 else PEAK_TFLOPS["bf16"]
        res[name] = {"ms_per_step": dt * 1e3, "achieved": flops / dt / 1e12, "peak": peak, "unit": "TFLOP/s",
                     "frac": flops / dt / 1e12 / peak, "bound": "mfma", "loss": loss.item(),
                     "loss_minus_f32": loss.item() - ref_loss, "products": nprod, "n_padded": Np,
                     "note": "whole affinity+softmax+chain+loss fwd+bwd wall time; flops = chain MFMA flops executed"}
    print(json.dumps({"metric": "chain fwd+bwd at kernel-only stress shape K", "workload": f"[B,T,N,C]=[1,{T},{N},{C}]",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "data": "synthetic", "modes": res}),
          flush=True)


def bench_shared(args):
    """ALL overlapping items of one 512x4096 radargram (225 windows of 32
    patch-columns) in one step, every patch-column encoded once (CRW.forward_columns), vs the same items
    batched the reference way (each item re-encoded; measured on a batch of 8 and scaled)."""
    import dataset as crw_dataset
    import encoder as crw_encoder
    import model as crw_model
    ds = crw_dataset.RGDataset.synthetic(H_RG, W_RG, T_SEQ, PATCH, OVERLAP, seed=11)
    cols = ds.columns()[None].cuda()
    items8 = torch.stack([ds[i] for i in range(8)]).cuda()
    torch.manual_seed(11)
    enc = crw_encoder.CNN(False)
    net = crw_model.CRW(enc, TAU, False).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)

    def timed(fn):
        for _ in range(max(1, args.warmup)):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, loss.item()

    def step_shared():
        opt.zero_grad(set_to_none=True)
        loss, _ = net.forward_columns(cols, T_SEQ)
        loss.backward()
        opt.step()
        return loss

    def step_items():
        opt.zero_grad(set_to_none=True)
        loss, _ = net(items8)
        loss.backward()
        opt.step()
        return loss

    dt_s, loss_s = timed(step_shared)
    dt_i, _ = timed(step_items)
    S = len(ds)
    print(json.dumps({"metric": "overlapping items/sec (CRW fwd+bwd+Adam, shared column encoding)", "value": S / dt_s,
                      "unit": "items/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_s * 1e3,
                      "higher_is_better": True, "data": "synthetic", "dtype": "f32 (bf16x3 conv trunk)",
                      "config": {"workload": f"{H_RG}x{W_RG} radargram, all {S} overlapping items [T={T_SEQ},N=63,16x16] per step; "
                                             f"{cols.shape[1]} patch-columns encoded once", "loss": loss_s},
                      "itemwise": {"ms_per_8_items": dt_i * 1e3, "items_per_s": 8 / dt_i,
                                   "what": "same model, CRW.forward on 8 overlapping items (each item re-encoded)"},
                      "speedup_vs_itemwise": (S / dt_s) / (8 / dt_i)}), flush=True)


def bench_labelprop(args):
    """BASELINE config 5: 410x8192 radargram, 32x32 patches, overlap (24,0) -> [T,N] = [256,48]; labels of the
    first patch column propagated along-track (CXT_SIZE 80 -> exercises the truncation quirk, RADIUS 10,
    TEMP 0.1, KNN 20, test_all.py defaults).  Metric: label-map columns per second; CPU leg = the oracle."""
    import numpy as np
    import dataset as crw_dataset
    import encoder as crw_encoder
    import utils as crw_utils
    from imported.labelprop import LabelPropVOS_CRW
    from oracle import crw_oracle as orc
    H, W, T, M = 410, 8192, 256, 4
    ds = crw_dataset.RGDataset.synthetic(H, W, T, (32, 32), (24, 0), seed=11)
    seq = ds[0].cuda()
    N = seq.shape[1]
    rows = N * 8 + 24
    seg = (torch.arange(rows)[:, None] * M // rows).float().repeat(1, 32).cuda()
    import crw_hip
    torch.manual_seed(11)
    if args.model == 1:
        # the reference's own cfg5 encoder: create_model(id = 1) = Resnet on 32 x 32 patches, never .eval()'d (scripts/test/
        # test_mc1.py:19-21,40-46): train-mode BatchNorm on the statistics of the radargram's 12 288 patches, hand-written kernels
        enc = crw_encoder.Resnet(False).cuda().train(True)
    else:
        enc = crw_encoder.CNN(False).cuda().eval()
    cfg = dict(CXT_SIZE=80, RADIUS=10, TEMP=0.1, KNN=20)
    if args.train_steps:  # Adam steps on the cycle loss first: a trained encoder separates the nodes' features (no near-ties left)
        import model as crw_model
        import dist as crw_dist
        import optim as crw_optim
        net = crw_model.CRW(enc, 0.01, False).cuda().train(True)
        bucket = crw_dist.FlatGradBucket(net.parameters(), lazy=True)
        opt = crw_optim.FlatAdam(bucket, lr=1e-3)
        items = crw_dataset.RGDataset.synthetic(H, 2048, 16, (32, 32), (24, 0), seed=12)
        batch = torch.stack([items[i] for i in range(0, len(items), max(1, len(items) // 4))][:4]).cuda()
        for _ in range(args.train_steps):
            bucket.zero()
            tl, _ = net(batch)
            tl.backward()
            bucket.all_reduce_mean()
            opt.step()
        if args.model == 0:
            enc.eval()

    lp = LabelPropVOS_CRW(cfg)  # once per run, like the reference's scripts (scripts/test/test_all.py:69, test_mc1.py:83)

    def run():
        return crw_utils.propagate(seq, seg, enc, lp, M, False, False)

    for _ in range(max(1, args.warmup)):
        pred, xent, _ = run()
    torch.cuda.synchronize()
    if not args.no_events:
        crw_hip.KERNEL_EVENTS = {}  # HIP events around every map-convolution call of the timed passes, on the launch stream
        if args.model == 1:
            crw_hip.rn_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pred, xent, _ = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    kernels = []
    if args.model == 1 and not args.no_events:
        ev = {}
        for rec in crw_hip.rn_timing_read():
            key = ("rn_conv" if rec.kind == 0 else "rn_wgrad", rec.mode, *list(rec.g), rec.k, rec.stride, rec.pad)
            ev.setdefault(key, []).append(rec.ms)
        crw_hip.rn_timing(False)
        crw_hip.KERNEL_EVENTS = None
        kernels = resnet_event_kernels(ev, T * N, args.steps)
    if crw_hip.KERNEL_EVENTS:
        ev, crw_hip.KERNEL_EVENTS = crw_hip.KERNEL_EVENTS, None
        P, mh, mw = T * N, 26, 26  # 32 x 32 patches -> 26 x 26 maps after the front end
        for (kind, cin, cout), pairs in ev.items():
            if kind != "fwd_map":
                continue
            kms = sum(e0.elapsed_time(e1) for e0, e1 in pairs) / len(pairs)
            alg = 2.0 * P * mh * mw * cin * cout * 9
            # executed: 4 full 10 x 10 tiles on 7 row tiles + 5 edge tiles (10 x 6, 6 x 10, 6 x 6) on 4 row tiles of 16 pixels, 3 MFMAs per product
            ex = 2.0 * P * (4 * 7 + 5 * 4) * 16 * cin * cout * 9 * 3
            kernels.append({"kernel": f"conv3x3_kernel<3,{cin},{cout},0,8,MAP> (+bias+ReLU), two launches: full tiles / small edge tiles",
                            "bound": "mfma", "achieved": alg / (kms * 1e-3) / 1e12, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                            "frac": alg / (kms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
                            "mfma_executed_frac": ex / (kms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"], "traffic": None,
                            "launch_us": kms * 1e3, "launches_per_step": len(pairs) // args.steps, "timed_launches": len(pairs),
                            "algorithmic_flops_per_launch": alg,
                            "note": "achieved = algorithmic fp32-equivalent flops (2 * P * 26 * 26 * cin * cout * 9, P = 12288 patches) / mean "
                                    "HIP-event time of the layer's launches inside the timed passes"})
        kernels.sort(key=lambda k: -k["launch_us"])
    # label propagation alone (features resident), the part the HIP kernels own
    with torch.no_grad():
        feats = crw_hip_normalize(enc, seq, T, N)  # (the Resnet in train mode: the same batch -> the same batch statistics and features)
    lp = LabelPropVOS_CRW(cfg)
    seed = crw_utils.seed_labels(seg, N)
    for _ in range(2):
        lp.propagate_all(feats, seed, M)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        p2, _ = lp.propagate_all(feats, seed, M)
    torch.cuda.synchronize()
    dlp = (time.perf_counter() - t1) / args.steps
    # CPU oracle on the same features (bounded: one pass)
    emb = feats.cpu().numpy()
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    tc = time.time()
    ref = orc.labelprop(emb, seed.cpu().numpy(), M, cfg["CXT_SIZE"], cfg["RADIUS"], cfg["TEMP"], cfg["KNN"])
    dcpu = time.time() - tc
    match = float((p2.cpu().numpy() == ref).mean())
    # free-running maps may differ downstream of fp32 near-ties; the teacher-forced fp64 audit says whether any differing
    # label is NOT a near-tie (oracle.labelprop_tie_audit; must be 0)
    _, Ldev = lp.propagate_all(feats, seed, M)
    audit = orc.labelprop_tie_audit(emb, Ldev.cpu().numpy(), p2.cpu().numpy(), cfg["CXT_SIZE"], cfg["RADIUS"], cfg["TEMP"],
                                    cfg["KNN"], eps=1e-5)
    print(json.dumps({"metric": "label-map columns/sec (user-seed label propagation)", "value": W / dt, "unit": "radargram columns/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
                      "data": "synthetic", "dtype": "f32",
                      "config": {"workload": f"{H}x{W} radargram, 32x32 patches overlap (24,0) -> [T,N]=[{T},{N}], {cfg}, "
                                             + ("Resnet encoder (the reference's create_model(id = 1), train-mode BatchNorm as in scripts/test/test_mc1.py; whole "
                                                "forward on the hand-written kernels, crw_rn_train_fwd)" if args.model == 1 else
                                                "CNN encoder (whole conv trunk on the tiled HIP kernels, FC head on PyTorch-ROCm)")
                                             + " + normalise + xent + top-k + gather",
                                 "encoder_weights": (f"trained: {args.train_steps} Adam steps on the cycle loss from the seeded initialisation (label map must "
                                                     "equal the oracle's outright)" if args.train_steps else
                                                     "random initialisation (seed 11): near-uniform affinities, so the free-running label map may differ "
                                                     "from the oracle's downstream of floating-point near-ties -- `label_mismatches_not_ties` (teacher-forced "
                                                     "fp64 audit) must be 0; --train-steps 300 gives the exact-match case")},
                      "labelprop_only": {"ms": dlp * 1e3, "columns_per_s": W / dlp,
                                         "what": "crw_labelprop_topk + crw_labelprop_gather, features resident"},
                      "cpu_baseline": {"value": W / dcpu, "unit": "radargram columns/s", "kind": "port", "cores": torch.get_num_threads(),
                                       "sample": f"oracle labelprop (numpy) on the same features, one pass, {dcpu:.2f} s"},
                      "label_agreement_with_oracle": match, "label_mismatches_not_ties": audit["not_ties"],
                      "tie_audit": audit, **({"roofline": kernels[0], "roofline_kernels": kernels} if kernels else {})}), flush=True)


def crw_hip_normalize(enc, seq, T, N):
    import crw_hip
    H, W = seq.shape[-2:]
    emb = enc(seq.reshape(-1, H, W).unsqueeze(1)).reshape(T, N, -1).float().contiguous()
    return crw_hip.normalize(emb)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.workload in ("radargram", "dense"):
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as CHILD processes (nothing here has
        # touched the GPU yet) and leave with their exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and int(os.environ.get("RANK", "0")) == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: reporting n_gpus = WORLD_SIZE",
              file=sys.stderr)
    if args.workload == "labelprop":
        import crw_hip
        crw_hip.lib()
        return bench_labelprop(args)
    if args.workload == "chain":
        import crw_hip
        crw_hip.lib()
        return bench_chain(args)
    if args.workload == "shared":
        import crw_hip
        crw_hip.lib()
        return bench_shared(args)
    global OVERLAP, H_RG, W_RG, PATCH
    if args.workload == "dense":  # shape family D: same radargram, vertical patch stride 1 -> N = 497
        OVERLAP = (15, 0)
        args.no_probe = args.no_cpu_baseline = True
    if args.workload == "train32":  # BASELINE config 5's geometry in TRAINING: 410 x 8192 radargram, 32 x 32 patches, overlap (24, 0)
        H_RG, W_RG, PATCH, OVERLAP = 410, 8192, (32, 32), (24, 0)  # -> 8 items [T=32, N=48, 32x32]: the tiled ("map") kernels, fwd + bwd
        args.no_probe = True
    import dist as crw_dist
    rank, world, local = crw_dist.init_from_env("nccl")
    if world > 1:  # the stand-alone probes (chain at n = 4096, walk alone) are single-GPU figures; the per-kernel
        args.no_probe = True  # roofline of the timed steps (HIP events) is reported at every N
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs an MI355X"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import crw_hip
    import model as crw_model
    import utils as crw_utils
    crw_hip.lib()

    torch.manual_seed(11)
    enc = crw_utils.create_model(args.model, False)
    if args.model == 0:
        enc.hip_convs = None if args.convs == "torch" else args.convs
    elif args.convs == "torch":
        enc.hip_convs = None  # Resnet on PyTorch-ROCm / MIOpen (the round-2 path), for comparison
    net = crw_model.CRW(enc, TAU, False).to(device)
    net.train(True)
    bucket = crw_dist.FlatGradBucket(net.parameters(), lazy=True)
    import optim as crw_optim
    opt = crw_optim.FlatAdam(bucket, lr=1e-3)  # torch.optim.Adam's update, one launch over the flat parameter buffer
    seq = make_batch(rank, device)
    B, T, N = seq.shape[:3]
    cols_per_step = B * (T * (PATCH[1] - OVERLAP[1]) + OVERLAP[1])

    def step():
        bucket.zero()
        loss, _ = net(seq)
        loss.backward()
        bucket.all_reduce_mean()
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if args.repeats <= 0:
        # size the run: one more untimed pass of `steps` steps tells how long a repeat takes; every rank must run the same number
        # of repeats, so the slowest rank's estimate decides
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        est = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(est, op=torch.distributed.ReduceOp.MAX)
        args.repeats = int(min(500, max(5, round(args.gpu_seconds / max(est.item(), 1e-4)))))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    if rank == 0 and not args.no_events:
        crw_hip.KERNEL_EVENTS = {}  # HIP events around every conv launch of the timed steps, on the launch stream
        if args.model == 1 and getattr(enc, "hip_convs", None):
            crw_hip.rn_timing(True)  # the Resnet pass is driven from native code: it records its own events (crw_rn_timing_*)
    rep_elapsed = []
    ev_repeats = min(max(1, args.repeats), 5)  # the per-kernel HIP events cover the first 5 repeats (tens of thousands of events otherwise)
    ev_stash, rn_records = None, None
    for rep in range(max(1, args.repeats)):  # each repeat: EXACTLY --steps steps between barrier + synchronize, MAX over ranks
        if rep == ev_repeats and rank == 0:
            ev_stash, crw_hip.KERNEL_EVENTS = crw_hip.KERNEL_EVENTS, None
            if args.model == 1 and not args.no_events and getattr(enc, "hip_convs", None):
                rn_records = crw_hip.rn_timing_read()
                crw_hip.rn_timing(False)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = t.item()
        rep_elapsed.append(el)
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]  # the median repeat is the one reported
    final_loss = loss.item()
    if rank == 0 and ev_stash is None:
        ev_stash, crw_hip.KERNEL_EVENTS = crw_hip.KERNEL_EVENTS, None
        if args.model == 1 and not args.no_events and getattr(enc, "hip_convs", None):
            rn_records = crw_hip.rn_timing_read()
            crw_hip.rn_timing(False)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "radargram columns/sec (CRW fwd+bwd)", "value": cols_per_step * world / (elapsed / args.steps),
            "unit": "radargram columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "repeats": {"n": len(rep_elapsed), "what": "value / ms_per_step are the median of n timed regions of `steps` steps each "
                                                       "(n sized to keep the GPU busy for about --gpu-seconds unless --repeats is given)",
                        "gpu_busy_s": sum(rep_elapsed),
                        "ms_per_step_min": min(rep_elapsed) / args.steps * 1e3, "ms_per_step_max": max(rep_elapsed) / args.steps * 1e3,
                        "ms_per_step_all": [round(e / args.steps * 1e3, 4) for e in rep_elapsed]},
            "dtype": {"bf16x3": "f32 (conv2-5 multiply on the bf16 matrix cores with hi/lo operand pairs, fp32 accumulate: fp32-grade; everything else fp32)",
                      "bf16": "bf16 (conv2-5 operands, fp32 accumulate), f32 elsewhere", "torch": "f32",
                      "mixed": "f32 forward (hi/lo operand pairs), bf16 operands / fp32 accumulate in the conv backward kernels, f32 elsewhere",
                      "resnet": "f32 (every convolution and the linear head multiply on the bf16 matrix cores with hi/lo operand pairs, fp32 accumulate: fp32-grade; "
                                "BatchNorm statistics merged in fp64; everything else fp32)"}[
                          args.convs if args.model == 0 else ("resnet" if getattr(enc, "hip_convs", None) else "torch")],
            "data": "synthetic",
            "config": {"workload": f"one synthetic {H_RG}x{W_RG} radargram per GPU per step = {B} items "
                                   f"[T={T},N={N},{PATCH[0]}x{PATCH[1]}] (patch {PATCH[0]}x{PATCH[1]}, overlap {OVERLAP}), tau={TAU}, "
                                   f"{'CNN' if args.model == 0 else 'Resnet'} encoder, fwd+bwd+all-reduce+Adam",
                       "columns_per_step_per_gpu": cols_per_step, "parallelism": f"dp{world} (independent sequences)",
                       "chain": "fp32 MFMA 16x16x4, prefix form",
                       "encoder_convs": args.convs if (args.model == 0 or args.convs == "torch") else "bf16x3 (resnet_hip)", "loss": final_loss},
        }
        kernels = []
        if rn_records is not None:
            ev = {}
            for rec in rn_records:
                key = ("rn_conv" if rec.kind == 0 else "rn_wgrad", rec.mode, *list(rec.g), rec.k, rec.stride, rec.pad)
                ev.setdefault(key, []).append(rec.ms)
            kernels = resnet_event_kernels(ev, B * T * N, args.steps * ev_repeats)
            ev_stash = None
        if ev_stash:
            # per-kernel durations measured live over the timed region (HIP events on the launch stream; first `ev_repeats` repeats)
            ev = ev_stash
            split = 3 if args.convs == "bf16x3" else 1  # ("mixed": the backward kernels' executed-MFMA figures are then 3x too high)
            P = B * T * N
            mpx = (PATCH[0] - 6) * (PATCH[1] - 6)  # pixels of the conv3-5 feature map (100 for 16 x 16 patches)
            names = {"fwd_map": "conv3x3_kernel<MAP> fwd (bias+ReLU), full + small-edge tile launches",
                     "bwd_map": "conv3x3_kernel<MAP> bwd-data (+ReLU mask)", "wgrad_map": "conv3x3_wgrad_kernel<MAP> (+ slice sum)",
                     "fwd": "conv3x3_kernel fwd (bias+ReLU)", "bwd": "conv3x3_kernel bwd-data (+ReLU mask)",
                     "front_fwd": "front_fwd_kernel (conv1-pool-conv2-pool, saves pool1 planes + pooling codes)",
                     "front_bwd": "front_bwd_saved_kernel (+ slice sum)",
                     "wgrad": "conv3x3_wgrad2_kernel / conv3x3_wgrad_kernel (+ slice sum)"}
            for (kind, cin, cout), pairs in sorted(ev.items(), key=lambda kv: (-kv[0][1] * kv[0][2], kv[0][0])):
                kms = sum(e0.elapsed_time(e1) for e0, e1 in pairs) / len(pairs)
                alg = 2.0 * P * (mpx if kind.endswith("_map") else 100) * cin * cout * 9  # algorithmic flops of one pass over one layer
                if kind.startswith("front"):  # conv1 (196 outputs x 8 ch x 25 cin taps) + conv2 (121 x 32 x 200), x2 for the backward
                    alg = 2.0 * P * (196 * 8 * 25 * cin + 121 * 32 * 200) * (2 if kind == "front_bwd" else 1)
                # MFMA padding of the pixel dim: forward / backward-data run 7 row tiles of 16 per patch; the streamed weight gradient
                # (128 output channels) takes its k-steps of 32 pixels over the whole pixel stream of a slice, the first-generation
                # kernel (conv3's 64 output channels) 10 k-steps per 3 patches
                pad = (1.0 if cout == 128 else (10 * 32 / 3) / 100.0) if kind == "wgrad" else 112.0 / 100.0
                if kind.endswith("_map"):  # 26 x 26 maps: 4 full tiles x 7 + 5 edge tiles x 4 row tiles of 16 (weight gradient: 9 units x 4 k-steps of 32)
                    pad = ((4 * 7 + 5 * 4) * 16 if kind != "wgrad_map" else 9 * 128) / float(mpx) if mpx == 676 else 1.0
                if kind.startswith("front"):
                    pad = 1.0
                kernels.append({"kernel": f"{names[kind]} cin={cin} cout={cout}", "bound": "mfma",
                                "achieved": alg / (kms * 1e-3) / 1e12, "peak": PEAK_TFLOPS["bf16"],
                                "unit": "TFLOP/s", "frac": alg / (kms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
                                "mfma_executed_tflops": alg * split * pad / (kms * 1e-3) / 1e12,
                                "mfma_executed_frac": alg * split * pad / (kms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
                                "traffic": None, "launch_us": kms * 1e3, "launches_per_step": len(pairs) // (args.steps * ev_repeats),
                                "timed_launches": len(pairs),
                                "note": "achieved = algorithmic (fp32-equivalent) flops / mean HIP-event time of this kernel's "
                                        "launches INSIDE the timed steps (events recorded on the launch stream); "
                                        f"split={split}: each product is {split} bf16 MFMAs"})
        if not args.no_probe:
            Np = crw_hip.padded_nodes(N)
            pms, pfl = chain_probe(Np, B, 3)
            kernels.append({"kernel": "gemm_pad_f32_kernel<32,32> (batched cycle products, one launch)", "bound": "mfma",
                            "achieved": pfl / (pms * 1e-3) / 1e12, "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                            "frac": pfl / (pms * 1e-3) / 1e12 / PEAK_TFLOPS["f32"], "traffic": None,
                            "launch_us": pms * 1e3, "shape": f"n={Np} batch={B}x3",
                            "note": "launch-latency bound at the reference-default node count"})
        # HBM traffic per launch from the committed PMC passes over this same command (rocprofv3 cannot run inside bench).  The
        # constants are keyed by kernel NAME, so they would go stale silently when a kernel changes and keeps its name: they are
        # attached only where the live in-step event time of the kernel is within 10 % of the duration recorded beside them.
        def pmc_fresh(k, rec):
            stored = rec.get("duration_us")
            if not stored:
                return False
            ok = abs(k["launch_us"] - stored) <= 0.10 * stored  # (boxes of one round differ by +-3-4 %; a changed kernel moves more or not at all)
            if not ok:
                k["traffic"] = None
                k["traffic_note"] = (f"committed PMC constants NOT attached: live in-step launch time {k['launch_us']:.1f} us differs by more "
                                     f"than 10 % from the {stored} us recorded with them (profiles/r04_pmc_in_step.json) -- re-run "
                                     "tools/r04_pmc_cnn.sh + tools/pmc_report.py")
            return ok
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_in_step_traffic.json")))["kernels"]
            dur = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_in_step.json")))["kernels"]
            for k in kernels:
                m = re.search(r"cin=(\d+) cout=(\d+)", k["kernel"])
                if not m or args.convs != "bf16x3" or B * T * N != 16128:  # the passes ran at the default workload
                    continue
                cin, cout = int(m.group(1)), int(m.group(2))
                key = ((f"conv3x3_wgrad2_kernel<3, {cin}, {cout}, 0>" if cout == 128 else
                        f"conv3x3_wgrad_kernel<3, {cin}, {cout}, {min(cin, 64)}, 4, false>") if "wgrad" in k["kernel"] else
                       f"conv3x3_kernel<3, {cout}, {cin}, 1, 8, false, 7>" if "bwd-data" in k["kernel"] else
                       f"conv3x3_kernel<3, {cin}, {cout}, 0, 8, false, 7>")
                if key in pmc and pmc_fresh(k, dur.get(key, {})):
                    k["traffic"] = pmc[key]["hbm_bytes"]
                    k["traffic_note"] = ("HBM bytes per launch INSIDE the step, committed PMC passes profiles/r04_pmc_in_step_traffic.json: "
                                         "reads = 2*1024*FETCH_SIZE (gfx950 correction) + writes = 1024*WRITE_SIZE; read/algorithmic = "
                                         f"{pmc[key]['read_over_algorithmic']}"
                                         + (f", write/algorithmic = {pmc[key]['write_over_algorithmic']}"
                                            if "write_over_algorithmic" in pmc[key] else ""))
            util = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_in_step.json")))["kernels"]
            for k in kernels:
                m = re.search(r"cin=(\d+) cout=(\d+)", k["kernel"])
                if not m or args.convs != "bf16x3" or B * T * N != 16128:
                    continue
                cin, cout = int(m.group(1)), int(m.group(2))
                key = ((f"conv3x3_wgrad2_kernel<3, {cin}, {cout}, 0>" if cout == 128 else
                        f"conv3x3_wgrad_kernel<3, {cin}, {cout}, {min(cin, 64)}, 4, false>") if "wgrad" in k["kernel"] else
                       f"conv3x3_kernel<3, {cout}, {cin}, 1, 8, false, 7>" if "bwd-data" in k["kernel"] else
                       f"conv3x3_kernel<3, {cin}, {cout}, 0, 8, false, 7>")
                if key in util and pmc_fresh(k, util[key]):
                    k["pmc_mfma_pipe_occupancy"] = util[key]["mfma_pipe_occupancy"]
                    k["pmc_effective_clock_GHz"] = util[key]["effective_clock_GHz"]
                    k["pmc_note"] = ("committed PMC pass over this same command (profiles/r04_pmc_in_step.json): "
                                     "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8) and the clock the chip held "
                                     "under this kernel inside the step (nominal 2.4 GHz)")
        except Exception:
            pass
        if kernels:
            # the dominant kernel of the timed step = the hand-written kernel with the largest time per step
            out["roofline"] = max((k for k in kernels if "launches_per_step" in k),
                                  key=lambda k: k["launch_us"] * k["launches_per_step"], default=kernels[-1])
            out["roofline_kernels"] = kernels
        if not args.no_probe:
            wms, wfl = walk_probe(B, T, N)
            out["walk_at_workload_shape"] = {"what": "affinity + dual softmax + chain + loss, fwd+bwd, features resident",
                                             "ms": wms, "chain_tflops": wfl / (wms * 1e-3) / 1e12,
                                             "share_of_step": wms / ms}
            kms, kfl = chain_probe(4096, 1, 3, iters=5)
            out["roofline_chain_n4096"] = {"kernel": "gemm_pad_f32_kernel<128,128>", "bound": "mfma",
                                           "achieved": kfl / (kms * 1e-3) / 1e12, "peak": PEAK_TFLOPS["f32"],
                                           "unit": "TFLOP/s", "frac": kfl / (kms * 1e-3) / 1e12 / PEAK_TFLOPS["f32"],
                                           "traffic": None, "launch_us": kms * 1e3, "shape": "n=4096 batch=1x3"}
            for split, key in ((1, "roofline_chain_n4096_bf16"), (3, "roofline_chain_n4096_bf16x3")):
                bms, bfl = chain_probe_bf16(4096, 3, split)
                out[key] = {"kernel": f"gemm_pad_bf16_kernel<{split}>", "bound": "mfma",
                            "achieved": bfl / (bms * 1e-3) / 1e12, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                            "frac": bfl / (bms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"], "traffic": None,
                            "launch_us": bms * 1e3, "shape": "n=4096 batch=3",
                            "note": "MFMA flops executed" + (" (3 bf16 MFMAs per product: useful flops = 1/3)"
                                                             if split == 3 else "")}
        if world == 1 and args.workload == "dense":
            # the same step with the opt-in hi/lo-pair chain arithmetic (CRW.chain = CRW_CHAIN_BF16X3: fp32-grade products at a third
            # of the bf16 rate instead of fp32 MFMA): at N = 497 the chain GEMMs are 11 % of the default step
            net.chain = crw_hip.CHAIN_BF16X3
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                l3 = step()
            torch.cuda.synchronize()
            ms3 = (time.perf_counter() - t1) / args.steps * 1e3
            out["option_chain_bf16x3"] = {"ms_per_step": ms3, "value": cols_per_step / (ms3 * 1e-3), "loss_after_these_steps": l3.item(),
                                          "note": "same workload, training continued with CRW.chain = CRW_CHAIN_BF16X3 (opt-in; the "
                                                  "default stays exact fp32)"}
        if world == 1 and args.model == 0 and args.convs == "bf16x3" and args.workload == "radargram" and not args.no_probe:
            # the same step in the two opt-in conv arithmetics (never `value`): BASELINE configs[2] words its shape "bf16 MFMA"
            out["options"] = {}
            for mode, what in (("mixed", "forward as the default (hi/lo pairs: the same loss and logits bit for bit), backward kernels on plain bf16 "
                                         "operands with fp32 accumulation; gradients within 2 % / cosine > 0.9995 of the fp32 oracle's "
                                         "(tests/test_hip_parity.py::test_full_model_at_baseline_shape_vs_oracle)"),
                               ("bf16", "plain bf16 operands, fp32 accumulation, forward and backward: loss within 1e-4 relative of the oracle's "
                                        "at this shape, logits NOT held to the 1e-4 bar (bf16 features at tau = 0.01), gradients within 5 %")):
                enc.hip_convs = mode
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    lo = step()
                torch.cuda.synchronize()
                mso = (time.perf_counter() - t1) / args.steps * 1e3
                out["options"][f"convs_{mode}"] = {"ms_per_step": mso, "value": cols_per_step / (mso * 1e-3), "unit": "radargram columns/s",
                                                   "what": what, "note": "opt-in (CNN.hip_convs / --convs), training continued from the timed steps"}
            enc.hip_convs = args.convs
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
