"""The kernel variants behind environment switches (read once per process, so each runs in a child process): the non-default
schedules and kernels that DESIGN.md quotes A/B numbers for must stay correct, not only the defaults the rest of the suite runs."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

RESNET = "tests/test_resnet_hip.py"
PARITY = "tests/test_hip_parity.py"
VARIANTS = [
    # Resnet products on the all-in-one 4-wave workgroups instead of the split-role ones (csrc/resnet_gemm.hip: launch_rn_conv)
    ("CRW_RN_SPEC", "0", RESNET, "native_and_stepwise or training_step or conv_forward_backward"),
    # 256-patch tiles on 8 waves, one workgroup per CU (measured slower; kept as the A/B partner DESIGN quotes)
    ("CRW_RN_TM", "256", RESNET, "native_and_stepwise or training_step or conv_forward_backward"),
    # BatchNorm-backward sums as a pass of their own instead of the products' LDS-staged epilogue (csrc/resnet_net.hip)
    ("CRW_RN_FUSE_RED", "0", RESNET, "native_and_stepwise or training_step or matches_pytorch_modules"),
    # everything on the caller's stream (no side stream)
    ("CRW_RN_STREAMS", "0", RESNET, "native_and_stepwise or training_step"),
    # forward front end on one 1024-thread workgroup per CU (csrc/encoder_front.hip)
    ("CRW_FRONT_NT", "1024", PARITY, "encoder_front_kernels or encoder_inference_trunk"),
    # bf16 chain GEMM with its A operand staged through registers (csrc/gemm_bf16.hip)
    ("CRW_GEMM_REGA", "1", PARITY, "gemm_bf16"),
    # plain-bf16 256 x 256 chain GEMM on the ring of five 32-deep half-tiles (opt-in, csrc/gemm_bf16.hip mainloop_ring5), incl. n = 4096
    ("CRW_GEMM_RING5", "1", PARITY + " tests/test_k_shape.py", "gemm_bf16 or bf16_chain_modes_on_256 or (walk_n4096 and 1-4)"),
    # 256 x 256 chain GEMM with every wave requesting its LDS-DMA right behind the barrier (rounds 1-3; the default staggers them), and
    # with the second half of the waves requesting after a quarter of the k-tile
    ("CRW_GEMM_STAGGER", "0", PARITY + " tests/test_k_shape.py", "gemm_bf16 or bf16_chain_modes_on_256 or (walk_n4096 and 1-4)"),
    ("CRW_GEMM_STAGGER", "2", PARITY, "gemm_bf16"),
    # the round-3 hand-off of the sums-with-tail kernels (relaxed atomics behind s_waitcnt) and the acq_rel form: A/B partners of the
    # default release + acquire fence (csrc/resnet_bn.hip rn_sums_tail_kernel)
    ("CRW_RN_TICKET", "relaxed", RESNET, "native_and_stepwise or training_step or reproducible"),
    ("CRW_RN_TICKET", "acqrel", RESNET, "native_and_stepwise or reproducible"),
    # the stem's forward product at other patch sizes than 16 x 16 on the gathered product of resnet_gemm.hip where the default is the
    # band-per-wave kernel (csrc/resnet_stem.hip rn_stem_fwd_band_kernel)
    ("CRW_RN_STEM_BAND", "0", RESNET, "matches_pytorch_modules or inference_through_propagate or training_step_matches_reference"),
    # matrix-core top-k with a query tile's scores in one piece (one workgroup per CU at config 5) where the default takes the context
    # frames in two halves through half the LDS (two workgroups per CU; csrc/labelprop.hip labelprop_topk_mfma_kernel<.., NCH>)
    ("CRW_LABELPROP_TOPK_CHUNKS", "1", PARITY, "labelprop_topk_on_matrix_cores or labelprop_matches_oracle_mcords_shape or propagate_matches_reference"),
    # label-propagation top-k on the vector kernel where the default scores on the fp32 matrix cores (csrc/labelprop.hip)
    ("CRW_LABELPROP_TOPK_VALU", "1", PARITY, "labelprop_matches_oracle_mcords_shape or labelprop_edge_cases or propagate_matches_reference"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,value,path,expr", VARIANTS, ids=[f"{v[0]}={v[1]}" for v in VARIANTS])
def test_kernel_variant_behind_env_switch(name, value, path, expr):
    env = dict(os.environ, **{name: value})
    r = subprocess.run([sys.executable, "-m", "pytest", *[os.path.join(ROOT, p_) for p_ in path.split()], "-x", "-q", "-m", "gpu", "-k", expr,
                        "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
