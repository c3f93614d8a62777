"""ctypes binding of libcrw_hip.so (C ABI: include/crw_hip.h) for the Python host code.

PyTorch-ROCm is plumbing here: it owns device memory and streams; every product computation of
the random walk goes through the HIP library.  There is NO fallback: if the library is missing or
a tensor is not on an MI355X device the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRW_HIP_LIB") or os.path.join(_HERE, "libcrw_hip.so")  # CRW_HIP_LIB: A/B of two builds (tools/)

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "crw_hip.h")


def _header_abi_version():
    """CRW_ABI_VERSION as include/crw_hip.h defines it -- the number is written nowhere else."""
    import re
    m = re.search(r"^#define\s+CRW_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read(), re.M)
    if not m:
        raise RuntimeError(f"no CRW_ABI_VERSION in {HEADER_PATH}")
    return int(m.group(1))


ABI_VERSION = _header_abi_version()
CRW_OK, CRW_EINVAL, CRW_EWORKSPACE, CRW_EHIP = 0, 1, 2, 3
CHAIN_F32, CHAIN_BF16, CHAIN_BF16X3 = 0, 1, 2
_ERR = {1: "CRW_EINVAL (bad shape / null pointer / unsupported size)",
        2: "CRW_EWORKSPACE (workspace too small)", 3: "CRW_EHIP (HIP launch failed)"}

_c_int, _c_f, _c_sz, _p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/crw_hip.h one to one
SIGNATURES = {
    "crw_abi_version": (_c_int, []),
    "crw_build_arch": (ctypes.c_char_p, []),
    "crw_last_hip_error": (_c_int, []),
    "crw_padded_nodes": (_c_int, [_c_int, _c_int]),
    "crw_walk_state_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "crw_walk_scratch_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "crw_affinity_ws_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "crw_affinity_fwd": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _c_f, _p, _p, _p, _p, _p, _c_sz, _p]),
    "crw_walk_fwd": (_c_int, [_p, _p, _c_int, _c_int, _c_int, _c_int, _p, _c_sz, _p, _p, _p]),
    "crw_walk_bwd": (_c_int, [_p, _p, _c_int, _c_int, _c_int, _c_int, _p, _c_sz, _p, _c_sz, _p, _p]),
    "crw_affinity_bwd": (_c_int, [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_f, _p, _p, _p]),
    "crw_normalize": (_c_int, [_p, _c_int, _c_int, _p, _p, _p]),
    "crw_labelprop_topk": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_int, _c_int, _p, _p, _p]),
    "crw_labelprop_topk_grid": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_int, _c_int, _c_int, _p, _p, _p]),
    "crw_labelprop_gather": (_c_int, [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
    "crw_pelt_rbf": (_c_int, [_p, _c_int, ctypes.c_double, _c_int, _c_int, ctypes.c_double, _p, _c_int]),
    "crw_labelprop_propagate": (_c_int, [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
    "crw_xent_metric": (_c_int, [_p, _c_int, _c_int, _c_int, _p, _p]),
    "crw_linear128_wgrad_ws_bytes": (_c_sz, [_c_int]),
    "crw_linear128_wgrad": (_c_int, [_p, _p, _p, _c_int, _p, _c_sz, _p]),
    "crw_adam_step": (_c_int, [_p, _p, _p, _p, ctypes.c_long, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                               _c_int, _p]),
    "crw_gemm_f32": (_c_int, [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p]),
    "crw_enc_pack_weights": (_c_int, [_p, _c_int, _c_int, _p, _p, _p, _p, _p]),
    "crw_enc_pack_input": (_c_int, [_p, _c_int, _c_int, _p, _p, _p]),
    "crw_enc_pack_input_map": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
    "crw_enc_conv3x3_map": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "crw_enc_conv3x3_wgrad_map": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _c_sz, _p]),
    "crw_enc_conv3x3": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "crw_enc_gap_bwd": (_c_int, [_p, _p, _c_int, _c_int, _c_int, _p, _p, _p]),
    "crw_enc_wgrad_ws_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "crw_enc_conv3x3_wgrad": (_c_int, [_c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _c_sz, _p]),
    "crw_enc_front_pack": (_c_int, [_p, _p, _p, _p, _p, _p]),
    "crw_enc_front_saved_bytes": (_c_sz, [_c_int]),
    "crw_enc_front_fwd": (_c_int, [_c_int, _p, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "crw_enc_front_fwd_map": (_c_int, [_c_int, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    "crw_enc_front_ws_bytes": (_c_sz, [_c_int, _c_int]),
    "crw_enc_front_bwd": (_c_int, [_c_int, _p, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                   _c_sz, _p]),
    "crw_enc_front_bwd_map": (_c_int, [_c_int, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                       _c_sz, _p]),
    # Resnet encoder kernels
    "crw_rn_padded_patches": (_c_int, [_c_int]),
    "crw_rn_pack_conv": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p]),
    "crw_rn_stem_toeplitz_ld": (_c_int, [_c_int]),
    "crw_rn_pack_stem": (_c_int, [_p, _c_int, _c_int, _p, _p, _p, _p, _p]),
    "crw_rn_conv_part_floats": (_c_sz, [_c_int, _c_int, _c_int]),
    "crw_rn_conv": (_c_int, [_c_int] * 12 + [_p] * 8),
    "crw_rn_wgrad_ws_bytes": (_c_sz, [_c_int] * 12),
    "crw_rn_wgrad": (_c_int, [_c_int] * 12 + [_p] * 6 + [_c_sz, _p]),
    "crw_rn_bn_stats_ws_bytes": (_c_sz, [_c_int]),
    "crw_rn_bn_stats": (_c_int, [_p, _c_int, _c_int, _c_int, _p, _p, _p, _p, _c_f, _c_f, _p, _p, _c_sz, _p]),
    "crw_rn_bn_apply": (_c_int, [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
    "crw_rn_bn_pool": (_c_int, [_p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p]),
    "crw_rn_bn_bwd_ws_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "crw_rn_bn_bwd": (_c_int, [_p] * 7 + [_c_int] * 3 + [_p] * 10 + [_c_sz, _p]),
    "crw_rn_pool_bwd_ws_bytes": (_c_sz, [_c_int] * 4),
    "crw_rn_pool_bwd": (_c_int, [_p] * 5 + [_c_int] * 4 + [_p] * 5 + [_c_sz, _p]),
    "crw_rn_stem_ws_bytes": (_c_sz, []),
    "crw_rn_stem_fwd": (_c_int, [_p] + [_c_int] * 6 + [_p] * 6 + [_c_f, _c_f] + [_p] * 4 + [_c_sz, _p]),
    "crw_rn_stem_bwd": (_c_int, [_p] * 5 + [_c_int] * 4 + [_p] * 5 + [_c_sz, _p]),
    "crw_rn_split": (_c_int, [_p, _c_int, _c_int, _p, _p, _p]),
    "crw_rn_colsum_ws_bytes": (_c_sz, [_c_int]),
    "crw_rn_colsum": (_c_int, [_p, _c_int, _c_int, _p, _p, _c_sz, _p]),
    "crw_rn_bn_stats_rows": (_c_int, [_p, _c_int, ctypes.c_double, _c_int, _p, _p, _p, _p, _c_f, _c_f, _p, _p, _c_sz, _p]),
    "crw_rn_stem_stats": (_c_int, [_p] + [_c_int] * 4 + [_p] * 6 + [_c_f, _c_f, _p, _p, _c_sz, _p]),
    "crw_rn_stem16_rows": (_c_int, []),
    "crw_rn_pack_stem16": (_c_int, [_p, _p, _p, _p]),
    "crw_rn_stem16_fwd": (_c_int, [_p, _c_int, _c_int, _p, _p, _p, _p, _p]),
    "crw_rn_stem_band_ok": (_c_int, [_c_int, _c_int]),
    "crw_rn_stem_band_fwd": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p]),
    "crw_rn_stem16_ws_bytes": (_c_sz, []),
    "crw_rn_stem16_wgrad": (_c_int, [_p, _c_int, _c_int, _p, _p, _p, _p, _p, _c_sz, _p]),
    "crw_rn_stem16_bwd": (_c_int, [_p, _c_int, _c_int] + [_p] * 11 + [_c_sz, _p]),
    "crw_rn_train_ws_bytes": (_c_sz, [_c_int] * 4),
    "crw_rn_train_fwd": (_c_int, [_p] + [_c_int] * 4 + [_p, _p, _p, _c_f, _c_f, _p, _p, _c_sz, _p]),
    "crw_rn_train_fwd_nograd": (_c_int, [_p] + [_c_int] * 4 + [_p, _p, _p, _c_f, _c_f, _p, _p, _c_sz, _p]),
    "crw_rn_train_bwd": (_c_int, [_p, _p] + [_c_int] * 4 + [_p, _p, _p, _c_sz, _p]),
    "crw_rn_eval_fwd": (_c_int, [_p] + [_c_int] * 4 + [_p, _p, _p, _c_f, _p, _p, _c_sz, _p]),
    "crw_rn_stem_cols": (_c_int, [_c_int]),
    "crw_rn_timing_enable": (_c_int, [_c_int]),
    "crw_rn_timing_read": (_c_int, [_p, _c_int]),
    "crw_gemm_bf16_ws_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "crw_gemm_bf16": (_c_int, [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _c_sz, _c_int, _p]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it has not been built: `python -c 'import
    __graft_entry__ as g; g.build()'` or `make -C radar-sounder-crw_amd/csrc`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built -- the CRW hot path has no CPU/PyTorch fallback; "
                               "run `make -C radar-sounder-crw_amd/csrc` (hipcc --offload-arch=gfx950)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.crw_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} was built for ABI {handle.crw_abi_version()}, include/crw_hip.h says {ABI_VERSION}: "
                               "stale library -- rebuild with `make -C radar-sounder-crw_amd/csrc`")
        _lib = handle
    return _lib


class CrwError(RuntimeError):
    """A non-zero status from the C ABI.  `status` is the CRW_* code: CRW_EINVAL is the caller's data (shape / size / null
    pointer), CRW_EWORKSPACE and CRW_EHIP are failures of the HIP path itself (`hip_error` = the runtime's last error code)."""

    def __init__(self, what, status, hip_error):
        super().__init__(f"{what} failed: {_ERR.get(status, status)} (hipError {hip_error})")
        self.what, self.status, self.hip_error = what, status, hip_error

    @property
    def device_failure(self):
        return self.status != CRW_EINVAL


def _check(status, what):
    if status != CRW_OK:
        raise CrwError(what, status, lib().crw_last_hip_error())


def _dev(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on an MI355X device (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous {dtype}")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# bench.py sets this to {} for its timed region: every conv / weight-gradient launch is then bracketed by two HIP events
# recorded on the stream the kernel is launched on, {(kind, layer_cin, layer_cout): [(start, end), ...]}.  None = off.
KERNEL_EVENTS = None


def _ev_begin():
    if KERNEL_EVENTS is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _ev_end(e0, key):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        KERNEL_EVENTS.setdefault(key, []).append((e0, e1))


def padded_nodes(N, chain=CHAIN_F32):
    return lib().crw_padded_nodes(N, chain)


# ------------------------------------------------------------------------------ training path
def affinity_fwd(emb, tau, want_stats=True):
    """emb [B,T,N,C] -> (A [B,T-1,N,N], ehat, norm, stats [4,B,T-1,N] | None).  stats = row max / row sum exp / column max /
    column sum exp of every A[b,t], from the epilogue of the affinity tiles (walk_fwd then needs no pass over A for them)."""
    B, T, N, C = emb.shape
    ehat = torch.empty_like(emb)
    norm = torch.empty(B, T, N, device=emb.device, dtype=torch.float32)
    A = torch.empty(B, T - 1, N, N, device=emb.device, dtype=torch.float32)
    stats = ws = None
    nbytes = 0
    if want_stats and T > 1:
        stats = torch.empty(4, B, T - 1, N, device=emb.device, dtype=torch.float32)
        nbytes = lib().crw_affinity_ws_bytes(B, T, N)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=emb.device)
    _check(lib().crw_affinity_fwd(_dev(emb, "emb"), B, T, N, C, float(tau), _dev(ehat, "ehat"), _dev(norm, "norm"),
                                  _dev(A, "A"), _dev(stats, "stats") if stats is not None else None,
                                  ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, nbytes, _stream()), "crw_affinity_fwd")
    return A, ehat, norm, stats


def affinity_bwd(dA, ehat, norm, tau):
    B, T, N, C = ehat.shape
    ws = torch.empty_like(ehat)
    demb = torch.empty_like(ehat)
    _check(lib().crw_affinity_bwd(_dev(dA, "dA"), _dev(ehat, "ehat"), _dev(norm, "norm"), B, T, N, C, float(tau),
                                  _dev(ws, "ws"), _dev(demb, "demb"), _stream()), "crw_affinity_bwd")
    return demb


def walk_fwd(A, chain=CHAIN_F32, want_At=False, stats=None):
    """A [B,T-1,N,N] (+ optional stats of affinity_fwd) -> (loss 0-d, state buffer, At [B,T-2,N,N] or None)."""
    B, Tm1, N, _ = A.shape
    T = Tm1 + 1
    nbytes = lib().crw_walk_state_bytes(B, T, N, chain)
    state = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
    loss = torch.empty((), dtype=torch.float32, device=A.device)
    At = torch.empty(B, max(T - 2, 0), N, N, device=A.device, dtype=torch.float32) if want_At else None
    if stats is not None and tuple(stats.shape) != (4, B, Tm1, N):
        raise RuntimeError(f"stats must be [4, {B}, {Tm1}, {N}] (got {tuple(stats.shape)})")
    _check(lib().crw_walk_fwd(_dev(A, "A"), _dev(stats, "stats") if stats is not None else None, B, T, N, chain,
                              ctypes.c_void_p(state.data_ptr()), nbytes,
                              _dev(At, "At") if (want_At and T > 2) else None, _dev(loss, "loss"), _stream()),
           "crw_walk_fwd")
    return loss, state, At


def walk_bwd(gloss, A, state, chain=CHAIN_F32):
    """dLoss/dA for the logits A the forward ran on (the softmaxes are recomputed from A and the statistics in `state`)."""
    B, Tm1, N, _ = A.shape
    T = Tm1 + 1
    nbytes = lib().crw_walk_scratch_bytes(B, T, N, chain)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=state.device)
    dA = torch.empty(B, T - 1, N, N, device=state.device, dtype=torch.float32)
    g = gloss.reshape(1).to(torch.float32).contiguous()
    _check(lib().crw_walk_bwd(_dev(g, "gloss"), _dev(A, "A"), B, T, N, chain, ctypes.c_void_p(state.data_ptr()), state.numel(),
                              ctypes.c_void_p(scratch.data_ptr()), nbytes, _dev(dA, "dA"), _stream()), "crw_walk_bwd")
    return dA


# ------------------------------------------------------------------------------ inference path
def normalize(emb):
    rows, C = emb.numel() // emb.shape[-1], emb.shape[-1]
    ehat = torch.empty_like(emb)
    _check(lib().crw_normalize(_dev(emb, "emb"), rows, C, _dev(ehat, "ehat"), None, _stream()), "crw_normalize")
    return ehat


def labelprop_topk(ehat, cxt_size, radius, temp, knn, first_frame=1, grid_w=1):
    """grid_w: the N nodes of a frame form an (N / grid_w) x grid_w grid (1 = a radargram's column of patches)"""
    T, N, C = ehat.shape
    W = torch.empty(T - first_frame, knn, N, device=ehat.device, dtype=torch.float32)
    I = torch.empty(T - first_frame, knn, N, device=ehat.device, dtype=torch.int32)
    _check(lib().crw_labelprop_topk_grid(_dev(ehat, "ehat"), T, N, C, int(cxt_size), int(radius), float(temp), int(knn),
                                         int(first_frame), int(grid_w), _dev(W, "W"), _dev(I, "I", torch.int32), _stream()),
           "crw_labelprop_topk_grid")
    return W, I


def labelprop_gather(seed, W, I, T, N, M, first_frame=1, L=None, pred=None, cxt_size=None):
    """cxt_size: the context size the lists were made with by `labelprop_topk` (same first_frame) -> crw_labelprop_propagate
    (chained frames in one workgroup, the frames beyond the context bound all at once); None: any lists, one workgroup."""
    knn = W.shape[1]
    if L is None:
        L = torch.empty(T * N, M, device=W.device, dtype=torch.float32)
    if pred is None:
        pred = torch.zeros(N, T, device=W.device, dtype=torch.float32)
    if cxt_size is not None:
        _check(lib().crw_labelprop_propagate(_dev(seed, "seed") if seed is not None else None, _dev(W, "W"),
                                             _dev(I, "I", torch.int32), T, N, M, knn, int(first_frame), int(cxt_size),
                                             _dev(L, "L"), _dev(pred, "pred"), _stream()), "crw_labelprop_propagate")
        return L, pred
    _check(lib().crw_labelprop_gather(_dev(seed, "seed") if seed is not None else None, _dev(W, "W"),
                                      _dev(I, "I", torch.int32), T, N, M, knn, int(first_frame), _dev(L, "L"),
                                      _dev(pred, "pred"), _stream()), "crw_labelprop_gather")
    return L, pred


def pelt_rbf(signal, pen, min_size=2, jump=5, gamma=None):
    """HOST function of the library (no GPU): PELT with the RBF kernel cost on a 1-D signal -> sorted breakpoints, the last one
    len(signal) -- pelt.pelt_rbf's arithmetic in C++ (csrc/pelt.cpp)."""
    import numpy as np
    x = np.ascontiguousarray(np.asarray(signal, dtype=np.float64).reshape(-1))
    n = len(x)
    out = np.empty(max(n, 1) + 1, dtype=np.int32)
    cnt = lib().crw_pelt_rbf(ctypes.c_void_p(x.ctypes.data), n, float(pen), int(min_size), int(jump),
                             -1.0 if gamma is None else float(gamma), ctypes.c_void_p(out.ctypes.data), len(out))
    if cnt < 1:
        raise CrwError("crw_pelt_rbf", -cnt, 0)
    return [int(v) for v in out[:cnt]]


def xent_metric(ehat):
    T, N, C = ehat.shape
    out = torch.empty(N, T - 1, device=ehat.device, dtype=torch.float32)
    _check(lib().crw_xent_metric(_dev(ehat, "ehat"), T, N, C, _dev(out, "xent"), _stream()), "crw_xent_metric")
    return out


def gemm_f32(A, B, C=None, transA=False, transB=False, beta=False):
    batch, n, _ = A.shape
    if C is None:
        C = torch.empty_like(A)
    _check(lib().crw_gemm_f32(_dev(A, "A"), _dev(B, "B"), _dev(C, "C"), n, batch, int(transA), int(transB), int(beta),
                              _stream()), "crw_gemm_f32")
    return C


def gemm_bf16(A, B, C=None, transA=False, transB=False, beta=False, split=1, ws=None, convert=True):
    """bf16 matrix-core product of fp32 [batch,n,n] operands (n % 128 == 0); returns (C, ws)."""
    batch, n, _ = A.shape
    if C is None:
        C = torch.empty_like(A)
    nbytes = lib().crw_gemm_bf16_ws_bytes(n, batch, split)
    if ws is None:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
    _check(lib().crw_gemm_bf16(_dev(A, "A"), _dev(B, "B"), _dev(C, "C"), n, batch, int(transA), int(transB),
                               int(beta), int(split), ctypes.c_void_p(ws.data_ptr()), ws.numel(), int(convert),
                               _stream()), "crw_gemm_bf16")
    return C, ws


# ------------------------------------------------------------------------------ encoder conv stack
_BF = torch.bfloat16


def _bf(t, name):
    return _dev(t, name, _BF) if t is not None else None


def enc_pack_weights(w, split):
    """fp32 conv weight [cout,cin,3,3] -> (fwd_hi, fwd_lo, bwd_hi, bwd_lo) bf16 planes (lo = None for split 1)."""
    cout, cin = w.shape[:2]
    mk = lambda a, b: torch.empty(9, a, b, dtype=_BF, device=w.device)
    fh, bh = mk(cout, cin), mk(cin, cout)
    fl, bl = (mk(cout, cin), mk(cin, cout)) if split == 3 else (None, None)
    _check(lib().crw_enc_pack_weights(_dev(w.contiguous(), "w"), cout, cin, _bf(fh, "fh"), _bf(fl, "fl"), _bf(bh, "bh"),
                                      _bf(bl, "bl"), _stream()), "crw_enc_pack_weights")
    return fh, fl, bh, bl


def enc_pack_input(x, split):
    """fp32 NCHW [P,C,10,10] -> channels-last planes [P,100,C]."""
    P, C, H, W = x.shape
    if (H, W) != (10, 10):
        raise RuntimeError(f"the HIP conv stack handles 10x10 feature maps (16x16 patches), got {H}x{W}")
    xh = torch.empty(P, 100, C, dtype=_BF, device=x.device)
    xl = torch.empty_like(xh) if split == 3 else None
    _check(lib().crw_enc_pack_input(_dev(x.contiguous(), "x"), P, C, _bf(xh, "xh"), _bf(xl, "xl"), _stream()),
           "crw_enc_pack_input")
    return xh, xl


def enc_conv3x3(mode, split, xh, xl, wh, wl, cout, bias=None, mask=None, planes=True, f32=False, gap=False,
                dgap=None, lo_plane=True):
    """mode 0: relu(conv + bias) ; mode 1: backward-data with optional ReLU mask.  -> (yh, yl, yf, gap).
    dgap (mode 1): the input gradient is dgap/100 gated by xh (forward activation plane), xl ignored."""
    P, _, cin = xh.shape
    dev = xh.device
    yh = torch.empty(P, 100, cout, dtype=_BF, device=dev) if planes else None
    yl = torch.empty_like(yh) if (planes and split == 3 and lo_plane) else None
    yf = torch.empty(P, 100, cout, dtype=torch.float32, device=dev) if f32 else None
    gp = torch.empty(P, cout, dtype=torch.float32, device=dev) if gap else None
    ev = _ev_begin()
    _check(lib().crw_enc_conv3x3(mode, split, P, cin, cout, _bf(xh, "xh"), _bf(xl, "xl"), _bf(wh, "wh"), _bf(wl, "wl"),
                                 _dev(bias, "bias") if bias is not None else None, _bf(mask, "mask"), _bf(yh, "yh"),
                                 _bf(yl, "yl"), _dev(yf, "yf") if f32 else None, _dev(gp, "gap") if gap else None,
                                 _dev(dgap, "dgap") if dgap is not None else None, _stream()), "crw_enc_conv3x3")
    _ev_end(ev, ("fwd", cin, cout) if mode == 0 else ("bwd", cout, cin))
    return yh, yl, yf, gp


def linear128_wgrad(dy, x):
    """dw [128,128] = dy.T @ x for dy, x [P,128] with P % 128 == 0 (split over P, deterministic)."""
    P = x.shape[0]
    dw = torch.empty(128, 128, dtype=torch.float32, device=x.device)
    nbytes = lib().crw_linear128_wgrad_ws_bytes(P)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _check(lib().crw_linear128_wgrad(_dev(dy.contiguous(), "dy"), _dev(x.contiguous(), "x"), _dev(dw, "dw"), P,
                                     ctypes.c_void_p(ws.data_ptr()), nbytes, _stream()), "crw_linear128_wgrad")
    return dw


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One Adam step in place on the flat fp32 buffers p (parameters), m, v with the flat gradient g (torch.optim.Adam defaults)."""
    _check(lib().crw_adam_step(_dev(p, "p"), _dev(g, "g"), _dev(m, "m"), _dev(v, "v"), p.numel(), float(lr), float(beta1),
                               float(beta2), float(eps), int(step), _stream()), "crw_adam_step")


def enc_pack_input_map(x, split):
    """fp32 NCHW [P,C,H,W] -> bf16 planes [P, H*W, C] (hi, lo | None)."""
    P, C, H, W = x.shape
    xh = torch.empty(P, H * W, C, dtype=_BF, device=x.device)
    xl = torch.empty_like(xh) if split == 3 else None
    _check(lib().crw_enc_pack_input_map(_dev(x.contiguous(), "x"), P, C, H, W, _bf(xh, "xh"), _bf(xl, "xl"), _stream()),
           "crw_enc_pack_input_map")
    return xh, xl


def enc_conv3x3_map(split, xh, xl, wh, wl, cout, H, W, bias=None, planes=True, gap=False, lo_plane=True, mode=0, mask=None,
                    f32=False):
    """mode 0: relu(conv3x3 + bias) on feature maps [P, H*W, cin] of any size -> (yh, yl, gap [P, cout] mean | None);
    mode 1: backward-data with the backward weight planes, optional ReLU mask map and fp32 copy -> (yh, yl, yf)."""
    P, hw, cin = xh.shape
    assert hw == H * W
    dev = xh.device
    yh = torch.empty(P, hw, cout, dtype=_BF, device=dev) if planes else None
    yl = torch.empty_like(yh) if (planes and split == 3 and lo_plane) else None
    yf = torch.empty(P, hw, cout, dtype=torch.float32, device=dev) if f32 else None
    ntile = ((H + 9) // 10) * ((W + 9) // 10)
    gp = torch.empty(P, ntile, cout, dtype=torch.float32, device=dev) if gap else None
    ev = _ev_begin()
    _check(lib().crw_enc_conv3x3_map(mode, split, P, H, W, cin, cout, _bf(xh, "xh"), _bf(xl, "xl"), _bf(wh, "wh"), _bf(wl, "wl"),
                                     _dev(bias, "bias") if bias is not None else None, _bf(mask, "mask"), _bf(yh, "yh"),
                                     _bf(yl, "yl"), _dev(yf, "yf") if f32 else None, _dev(gp, "gap") if gap else None, _stream()),
           "crw_enc_conv3x3_map")
    _ev_end(ev, ("fwd_map", cin, cout) if mode == 0 else ("bwd_map", cout, cin))
    if mode == 1:
        return yh, yl, yf
    return yh, yl, (gp.sum(1) / float(hw) if gap else None)


def enc_wgrad_map(split, dyh, dyl, xh, xl, H, W):
    """weight / bias gradient of a 3x3 layer on feature maps [P, H*W, C] -> (dw [cout,cin,3,3], db [cout])."""
    P, hw, cout = dyh.shape
    cin = xh.shape[2]
    assert hw == H * W
    dw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=xh.device)
    db = torch.empty(cout, dtype=torch.float32, device=xh.device)
    units = P * ((H + 9) // 10) * ((W + 9) // 10)
    nbytes = lib().crw_enc_wgrad_ws_bytes(units, cin, cout, split)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xh.device)
    ev = _ev_begin()
    _check(lib().crw_enc_conv3x3_wgrad_map(split, P, H, W, cin, cout, _bf(dyh, "dyh"), _bf(dyl, "dyl"), _bf(xh, "xh"),
                                           _bf(xl, "xl"), _dev(dw, "dw"), _dev(db, "db"), ctypes.c_void_p(ws.data_ptr()), nbytes,
                                           _stream()), "crw_enc_conv3x3_wgrad_map")
    _ev_end(ev, ("wgrad_map", cin, cout))
    return dw, db


def enc_gap_bwd(dgap, yh, split):
    """dY = dgap / npix where yh != 0, for planes [P, npix, C] of any map size."""
    P, npix, C = yh.shape
    dh = torch.empty_like(yh)
    dl = torch.empty_like(yh) if split == 3 else None
    _check(lib().crw_enc_gap_bwd(_dev(dgap.contiguous(), "dgap"), _bf(yh, "yh"), P, C, npix, _bf(dh, "dh"), _bf(dl, "dl"),
                                 _stream()), "crw_enc_gap_bwd")
    return dh, dl


def enc_wgrad(split, dyh, dyl, xh, xl, dgap=None):
    """dgap: dY = dgap/100 gated by dyh (forward activation plane) -- fused ReLU + GAP backward."""
    P, _, cout = dyh.shape
    cin = xh.shape[2]
    dw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=xh.device)
    db = torch.empty(cout, dtype=torch.float32, device=xh.device)
    nbytes = lib().crw_enc_wgrad_ws_bytes(P, cin, cout, split)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xh.device)
    ev = _ev_begin()
    _check(lib().crw_enc_conv3x3_wgrad(split, P, cin, cout, _bf(dyh, "dyh"), _bf(dyl, "dyl"), _bf(xh, "xh"),
                                       _bf(xl, "xl"), _dev(dgap, "dgap") if dgap is not None else None,
                                       _dev(dw, "dw"), _dev(db, "db"), ctypes.c_void_p(ws.data_ptr()),
                                       nbytes, _stream()), "crw_enc_conv3x3_wgrad")
    _ev_end(ev, ("wgrad", cin, cout))
    return dw, db


def enc_front_pack(w2, split):
    """conv2 weight [32,8,5,5] -> (fwd_hi, fwd_lo, bwd_hi, bwd_lo) bf16 planes."""
    fh = torch.empty(7, 32, 32, dtype=_BF, device=w2.device)
    bh = torch.empty(25, 8, 32, dtype=_BF, device=w2.device)
    fl, bl = (torch.empty_like(fh), torch.empty_like(bh)) if split == 3 else (None, None)
    _check(lib().crw_enc_front_pack(_dev(w2.contiguous(), "w2"), _bf(fh, "fh"), _bf(fl, "fl"), _bf(bh, "bh"),
                                    _bf(bl, "bl"), _stream()), "crw_enc_front_pack")
    return fh, fl, bh, bl


def enc_front_fwd(split, x, w1, b1, w2f, b2, save=False):
    """x [P,cin,16,16] -> planes [P,100,32] (conv1-ReLU-pool-conv2-ReLU-pool).  save=True (training): also returns the
    record (pool1 planes + pooling codes) that lets enc_front_bwd skip the recomputation: (yh, yl, saved)."""
    P, cin = x.shape[:2]
    yh = torch.empty(P, 100, 32, dtype=_BF, device=x.device)
    yl = torch.empty_like(yh) if split == 3 else None
    saved = torch.empty(lib().crw_enc_front_saved_bytes(P), dtype=torch.uint8, device=x.device) if save else None
    ev = _ev_begin()
    _check(lib().crw_enc_front_fwd(split, _dev(x, "x"), P, cin, _dev(w1.contiguous(), "w1"), _dev(b1, "b1"),
                                   _bf(w2f[0], "w2h"), _bf(w2f[1], "w2l"), _dev(b2, "b2"), _bf(yh, "yh"), _bf(yl, "yl"),
                                   ctypes.c_void_p(saved.data_ptr()) if save else None, _stream()), "crw_enc_front_fwd")
    _ev_end(ev, ("front_fwd", cin, 32))
    return (yh, yl, saved) if save else (yh, yl)


def enc_front_fwd_map(split, x, w1, b1, w2f, b2):
    """front end on patches of any size: x [P,cin,H,W] -> planes [P, (H-6)*(W-6), 32] (hi, lo | None)."""
    P, cin, H, W = x.shape
    yh = torch.empty(P, (H - 6) * (W - 6), 32, dtype=_BF, device=x.device)
    yl = torch.empty_like(yh) if split == 3 else None
    _check(lib().crw_enc_front_fwd_map(split, _dev(x.contiguous(), "x"), P, cin, H, W, _dev(w1.contiguous(), "w1"),
                                       _dev(b1, "b1"), _bf(w2f[0], "w2h"), _bf(w2f[1], "w2l"), _dev(b2, "b2"),
                                       _bf(yh, "yh"), _bf(yl, "yl"), _stream()), "crw_enc_front_fwd_map")
    return yh, yl


def enc_front_bwd(split, x, w1, b1, w2f, b2, w2b, dy, saved=None):
    """-> (dw1, db1, dw2, db2).  saved: the record of enc_front_fwd(save=True) for the same patches (no recomputation)."""
    P, cin = x.shape[:2]
    dev = x.device
    dw1 = torch.empty(8, cin, 5, 5, device=dev)
    db1 = torch.empty(8, device=dev)
    dw2 = torch.empty(32, 8, 5, 5, device=dev)
    db2 = torch.empty(32, device=dev)
    nbytes = lib().crw_enc_front_ws_bytes(P, cin)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if saved is not None and saved.numel() < lib().crw_enc_front_saved_bytes(P):
        raise RuntimeError("saved record too small for this patch count")
    ev = _ev_begin()
    _check(lib().crw_enc_front_bwd(split, _dev(x, "x"), P, cin, _dev(w1.contiguous(), "w1"), _dev(b1, "b1"),
                                   _bf(w2f[0], "w2h"), _bf(w2f[1], "w2l"), _dev(b2, "b2"), _bf(w2b[0], "w2bh"),
                                   _bf(w2b[1], "w2bl"), _dev(dy.contiguous(), "dy"),
                                   ctypes.c_void_p(saved.data_ptr()) if saved is not None else None, _dev(dw1, "dw1"),
                                   _dev(db1, "db1"), _dev(dw2, "dw2"), _dev(db2, "db2"), ctypes.c_void_p(ws.data_ptr()), nbytes,
                                   _stream()), "crw_enc_front_bwd")
    _ev_end(ev, ("front_bwd", cin, 32))
    return dw1, db1, dw2, db2


def enc_front_bwd_map(split, x, w1, b1, w2f, b2, w2b, dy):
    """front-end backward on patches of any size: x [P,cin,H,W], dy [P,(H-6)*(W-6),32] fp32 -> (dw1, db1, dw2, db2)."""
    P, cin, H, W = x.shape
    dev = x.device
    dw1 = torch.empty(8, cin, 5, 5, device=dev)
    db1 = torch.empty(8, device=dev)
    dw2 = torch.empty(32, 8, 5, 5, device=dev)
    db2 = torch.empty(32, device=dev)
    units = P * ((H - 6 + 9) // 10) * ((W - 6 + 9) // 10)
    nbytes = lib().crw_enc_front_ws_bytes(units, cin)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _check(lib().crw_enc_front_bwd_map(split, _dev(x.contiguous(), "x"), P, cin, H, W, _dev(w1.contiguous(), "w1"), _dev(b1, "b1"),
                                       _bf(w2f[0], "w2h"), _bf(w2f[1], "w2l"), _dev(b2, "b2"), _bf(w2b[0], "w2bh"),
                                       _bf(w2b[1], "w2bl"), _dev(dy.contiguous(), "dy"), _dev(dw1, "dw1"), _dev(db1, "db1"),
                                       _dev(dw2, "dw2"), _dev(db2, "db2"), ctypes.c_void_p(ws.data_ptr()), nbytes, _stream()),
           "crw_enc_front_bwd_map")
    return dw1, db1, dw2, db2


# ------------------------------------------------------------------------------ Resnet encoder kernels (crw_rn_*)
RN_FWD, RN_BWD, RN_STEM_FWD, RN_STEM_BWD = 0, 1, 2, 3


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


def rn_padded(P):
    return lib().crw_rn_padded_patches(int(P))


def rn_pack_conv(w):
    """conv / linear weight [cout,cin,kh,kw] (or [cout,cin]) fp32 -> (fwd_hi, fwd_lo, bwd_hi, bwd_lo) bf16 planes."""
    w = w.detach()
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    mk = lambda: torch.empty(cout * cin * kh * kw, dtype=_BF, device=w.device)
    fh, fl, bh, bl = mk(), mk(), mk(), mk()
    _check(lib().crw_rn_pack_conv(_dev(w.contiguous(), "w"), cout, cin, kh, kw, _bf(fh, "fh"), _bf(fl, "fl"), _bf(bh, "bh"),
                                  _bf(bl, "bl"), _stream()), "crw_rn_pack_conv")
    return fh, fl, bh, bl


def rn_stem_cols(w):
    """columns of one row of the stem's input gradient: 3 * (w + 2) rounded up to 64"""
    return lib().crw_rn_stem_cols(int(w))


def rn_pack_stem(w1, h, w):
    """model.conv1 weight [64,3,7,7] -> (fwd_hi, fwd_lo [64*256], toeplitz_hi, toeplitz_lo [(h+2)*rn_stem_cols(w)*ld])."""
    ld = lib().crw_rn_stem_toeplitz_ld(w)
    fh = torch.empty(64 * 256, dtype=_BF, device=w1.device)
    fl = torch.empty_like(fh)
    th = torch.empty((h + 2) * rn_stem_cols(w) * ld, dtype=_BF, device=w1.device)
    tl = torch.empty_like(th)
    _check(lib().crw_rn_pack_stem(_dev(w1.detach().contiguous(), "w1"), h, w, _bf(fh, "fh"), _bf(fl, "fl"), _bf(th, "th"),
                                  _bf(tl, "tl"), _stream()), "crw_rn_pack_stem")
    return fh, fl, th, tl


def rn_conv(mode, P, src, dst, N, k, stride, pad, a, b, bias=None, stats=False):
    """a = (hi, lo) planes on the source map src = (Hs, Ws, Cs); b = (hi, lo) weight planes; dst = (Hd, Wd).
    -> (out fp32 [Ppad, G*N], part | None)."""
    Hs, Ws, Cs = src
    Hd, Wd = dst
    G = Hd if mode == RN_STEM_BWD else Hd * Wd
    Ppad = rn_padded(P)
    dev = a[0].device
    out = torch.empty(Ppad, G * N, dtype=torch.float32, device=dev)
    part = torch.empty(lib().crw_rn_conv_part_floats(P, G, N), dtype=torch.float32, device=dev) if stats else None
    ev = _ev_begin()
    _check(lib().crw_rn_conv(mode, P, Hs, Ws, Cs, Hd, Wd, N, k[0], k[1], stride, pad, _bf(a[0], "a_hi"), _bf(a[1], "a_lo"),
                             _bf(b[0], "b_hi"), _bf(b[1], "b_lo"), _dev(bias, "bias") if bias is not None else None,
                             _dev(out, "out"), _ptr(part), _stream()), "crw_rn_conv")
    _ev_end(ev, ("rn_conv", mode, Hs, Ws, Cs, Hd, Wd, N, k[0], stride, pad))
    return out, part


def rn_wgrad(mode, P, xin, xout, k, stride, pad, x, d):
    """x = (hi, lo) input planes on xin = (Hin, Win, Cin), d = (hi, lo) dZ planes on xout = (Hout, Wout, Cout)
    -> dw [cout, cin, kh, kw] ([64,3,7,7] for the stem)."""
    Hin, Win, Cin = xin
    Hout, Wout, Cout = xout
    dev = x[0].device
    geo = (mode, P, Hin, Win, Cin, Hout, Wout, Cout, k[0], k[1], stride, pad)
    nbytes = lib().crw_rn_wgrad_ws_bytes(*geo)
    if nbytes == 0:
        raise RuntimeError(f"crw_rn_wgrad: unsupported geometry {geo}")
    ws = _ws(nbytes, dev)
    dw = torch.empty((64, 3, 7, 7) if mode == RN_STEM_FWD else (Cout, Cin, k[0], k[1]), dtype=torch.float32, device=dev)
    ev = _ev_begin()
    _check(lib().crw_rn_wgrad(*geo, _bf(x[0], "x_hi"), _bf(x[1], "x_lo"), _bf(d[0], "d_hi"), _bf(d[1], "d_lo"), _dev(dw, "dw"),
                              _ptr(ws), nbytes, _stream()), "crw_rn_wgrad")
    _ev_end(ev, ("rn_wgrad", mode, Hin, Win, Cin, Hout, Wout, Cout, k[0], stride, pad))
    return dw


def rn_bn_stats(part, P, G, bn, momentum, update_running=True):
    """per-tile statistics of a convolution output -> coef [4, C]; updates bn.running_mean / running_var in place."""
    C = bn.weight.numel()
    dev = part.device
    coef = torch.empty(4, C, dtype=torch.float32, device=dev)
    nbytes = lib().crw_rn_bn_stats_ws_bytes(C)
    ws = _ws(nbytes, dev)
    rm = bn.running_mean if (update_running and bn.running_mean is not None) else None
    rv = bn.running_var if rm is not None else None
    _check(lib().crw_rn_bn_stats(_dev(part, "part"), P, G, C, _dev(bn.weight.detach(), "gamma"), _dev(bn.bias.detach(), "beta"),
                                 _ptr(rm), _ptr(rv), float(momentum), float(bn.eps), _dev(coef, "coef"), _ptr(ws), nbytes,
                                 _stream()), "crw_rn_bn_stats")
    return coef


def rn_bn_apply(Z, coef, P, npix, C, Zd=None, coef_d=None, res=None, relu=True):
    Ppad = rn_padded(P)
    yh = torch.empty(Ppad, npix * C, dtype=_BF, device=Z.device)
    yl = torch.empty_like(yh)
    _check(lib().crw_rn_bn_apply(_dev(Z, "Z"), _dev(coef, "coef"), _ptr(Zd), _ptr(coef_d), _ptr(res[0]) if res else None,
                                 _ptr(res[1]) if res else None, P, npix, C, int(relu), _bf(yh, "yh"), _bf(yl, "yl"), _stream()),
           "crw_rn_bn_apply")
    return yh, yl


def rn_bn_pool(Z, coef, P, H, W, C):
    Ppad = rn_padded(P)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    yh = torch.empty(Ppad, Ho * Wo * C, dtype=_BF, device=Z.device)
    yl = torch.empty_like(yh)
    amax = torch.empty(Ppad, Ho * Wo * C, dtype=torch.uint8, device=Z.device)
    _check(lib().crw_rn_bn_pool(_dev(Z, "Z"), _dev(coef, "coef"), P, H, W, C, _bf(yh, "yh"), _bf(yl, "yl"), _ptr(amax), _stream()),
           "crw_rn_bn_pool")
    return (yh, yl), amax


def rn_bn_bwd(g1, g2, mask_hi, Z, coef, P, npix, C, Zd=None, coef_d=None, want_g=False):
    """-> (dz (hi, lo), dzd (hi, lo) | None, g fp32 | None, dgamma, dbeta, dgamma_d | None, dbeta_d | None)"""
    Ppad = rn_padded(P)
    dev = Z.device
    mk = lambda: torch.empty(Ppad, npix * C, dtype=_BF, device=dev)
    dzh, dzl = mk(), mk()
    dzdh, dzdl = (mk(), mk()) if Zd is not None else (None, None)
    gout = torch.empty(Ppad, npix * C, dtype=torch.float32, device=dev) if want_g else None
    vec = lambda: torch.empty(C, dtype=torch.float32, device=dev)
    dg, db = vec(), vec()
    dgd, dbd = (vec(), vec()) if Zd is not None else (None, None)
    nbytes = lib().crw_rn_bn_bwd_ws_bytes(P, npix, C)
    ws = _ws(nbytes, dev)
    _check(lib().crw_rn_bn_bwd(_dev(g1, "g1"), _ptr(g2), _bf(mask_hi, "mask"), _dev(Z, "Z"), _dev(coef, "coef"), _ptr(Zd),
                               _ptr(coef_d), P, npix, C, _bf(dzh, "dzh"), _bf(dzl, "dzl"), _ptr(dzdh), _ptr(dzdl), _ptr(gout),
                               _ptr(dg), _ptr(db), _ptr(dgd), _ptr(dbd), _ptr(ws), nbytes, _stream()), "crw_rn_bn_bwd")
    return (dzh, dzl), ((dzdh, dzdl) if Zd is not None else None), gout, dg, db, dgd, dbd


def rn_pool_bwd(d1, d2, amax, Z, coef, P, H, W, C):
    Ppad = rn_padded(P)
    dev = Z.device
    dzh = torch.empty(Ppad, H * W * C, dtype=_BF, device=dev)
    dzl = torch.empty_like(dzh)
    dg = torch.empty(C, dtype=torch.float32, device=dev)
    db = torch.empty_like(dg)
    nbytes = lib().crw_rn_pool_bwd_ws_bytes(P, H, W, C)
    ws = _ws(nbytes, dev)
    _check(lib().crw_rn_pool_bwd(_dev(d1, "d1"), _ptr(d2), _ptr(amax), _dev(Z, "Z"), _dev(coef, "coef"), P, H, W, C, _bf(dzh, "dzh"),
                                 _bf(dzl, "dzl"), _ptr(dg), _ptr(db), _ptr(ws), nbytes, _stream()), "crw_rn_pool_bwd")
    return (dzh, dzl), dg, db


def rn_stem_fwd(x, fc0, bn0, Hm, Wm, momentum, update_running=True):
    """x [P,cin,h,w] -> (map (hi, lo) [Ppad, Hm*Wm*4], stem record [32])."""
    P, cin, h, w = x.shape
    Ppad = rn_padded(P)
    dev = x.device
    mh = torch.empty(Ppad, Hm * Wm * 4, dtype=_BF, device=dev)
    ml = torch.empty_like(mh)
    stem = torch.empty(32, dtype=torch.float32, device=dev)
    nbytes = lib().crw_rn_stem_ws_bytes()
    ws = _ws(nbytes, dev)
    rm = bn0.running_mean if (update_running and bn0.running_mean is not None) else None
    rv = bn0.running_var if rm is not None else None
    _check(lib().crw_rn_stem_fwd(_dev(x, "x"), P, cin, h, w, Hm, Wm, _dev(fc0.weight.detach().reshape(3, cin).contiguous(), "w0"),
                                 _dev(fc0.bias.detach(), "b0"), _dev(bn0.weight.detach(), "gamma"), _dev(bn0.bias.detach(), "beta"),
                                 _ptr(rm), _ptr(rv), float(momentum), float(bn0.eps), _bf(mh, "mh"), _bf(ml, "ml"), _dev(stem, "stem"),
                                 _ptr(ws), nbytes, _stream()), "crw_rn_stem_fwd")
    return (mh, ml), stem


def rn_stem_bwd(dX0, x, stem, w0, b0):
    P, cin, h, w = x.shape
    dev = x.device
    dw0 = torch.empty(3, cin, 1, 1, dtype=torch.float32, device=dev)
    db0, dg, db = (torch.empty(3, dtype=torch.float32, device=dev) for _ in range(3))
    nbytes = lib().crw_rn_stem_ws_bytes()
    ws = _ws(nbytes, dev)
    _check(lib().crw_rn_stem_bwd(_dev(dX0, "dX0"), _dev(x, "x"), _dev(stem, "stem"), _dev(w0.reshape(3, cin).contiguous(), "w0"),
                                 _dev(b0, "b0"), P, cin, h, w, _dev(dw0, "dw0"), _dev(db0, "db0"), _dev(dg, "dg"), _dev(db, "db"),
                                 _ptr(ws), nbytes, _stream()), "crw_rn_stem_bwd")
    return dw0, db0, dg, db


def rn_stem_stats(x, fc0, bn0, momentum):
    """bn0's batch statistics from the moments of the patches -> stem record [32] (updates bn0's running statistics)"""
    P, cin, h, w = x.shape
    stem = torch.empty(32, dtype=torch.float32, device=x.device)
    nbytes = lib().crw_rn_stem_ws_bytes()
    ws = _ws(nbytes, x.device)
    _check(lib().crw_rn_stem_stats(_dev(x, "x"), P, cin, h, w, _dev(fc0.weight.detach().reshape(3, cin).contiguous(), "w0"),
                                   _dev(fc0.bias.detach(), "b0"), _dev(bn0.weight.detach(), "gamma"), _dev(bn0.bias.detach(), "beta"),
                                   _ptr(bn0.running_mean), _ptr(bn0.running_var), float(momentum), float(bn0.eps), _dev(stem, "stem"),
                                   _ptr(ws), nbytes, _stream()), "crw_rn_stem_stats")
    return stem


def rn_pack_stem16(w1):
    wf = torch.empty(28672, dtype=_BF, device=w1.device)
    wt = torch.empty_like(wf)
    _check(lib().crw_rn_pack_stem16(_dev(w1.detach().contiguous(), "w1"), _bf(wf, "wf"), _bf(wt, "wt"), _stream()), "crw_rn_pack_stem16")
    return wf, wt


def rn_stem16_fwd(x, stem, wf):
    """-> (Z1 [Ppad, 81*64] fp32 (rows of the padding patches zero), part [rows, 64, 2])"""
    P, cin = x.shape[:2]
    Z1 = torch.zeros(rn_padded(P), 81 * 64, dtype=torch.float32, device=x.device)
    part = torch.empty(lib().crw_rn_stem16_rows(), 64, 2, dtype=torch.float32, device=x.device)
    _check(lib().crw_rn_stem16_fwd(_dev(x, "x"), P, cin, _dev(stem, "stem"), _bf(wf, "wf"), _dev(Z1, "Z1"), _dev(part, "part"),
                                   _stream()), "crw_rn_stem16_fwd")
    return Z1, part


def rn_stem_band_ok(h, w):
    return bool(lib().crw_rn_stem_band_ok(int(h), int(w)))


def rn_stem_band_fwd(x, stem, wf, H1, W1):
    """the stem's forward product for patches of any size -> (Z1 [Ppad, H1*W1*64] fp32, part [rows, 64, 2])"""
    P, cin, h, w = x.shape
    Z1 = torch.zeros(rn_padded(P), H1 * W1 * 64, dtype=torch.float32, device=x.device)
    part = torch.empty(lib().crw_rn_stem16_rows(), 64, 2, dtype=torch.float32, device=x.device)
    _check(lib().crw_rn_stem_band_fwd(_dev(x, "x"), P, cin, h, w, _dev(stem, "stem"), _bf(wf, "wf"), _dev(Z1, "Z1"), _dev(part, "part"),
                                      _stream()), "crw_rn_stem_band_fwd")
    return Z1, part


def rn_bn_stats_rows(part, count, bn, momentum):
    rows, C = part.shape[0], bn.weight.numel()
    coef = torch.empty(4, C, dtype=torch.float32, device=part.device)
    nbytes = lib().crw_rn_bn_stats_ws_bytes(C)
    ws = _ws(nbytes, part.device)
    _check(lib().crw_rn_bn_stats_rows(_dev(part, "part"), rows, float(count), C, _dev(bn.weight.detach(), "gamma"),
                                      _dev(bn.bias.detach(), "beta"), _ptr(bn.running_mean), _ptr(bn.running_var), float(momentum),
                                      float(bn.eps), _dev(coef, "coef"), _ptr(ws), nbytes, _stream()), "crw_rn_bn_stats_rows")
    return coef


def rn_stem16_wgrad(x, stem, dz):
    P, cin = x.shape[:2]
    dw = torch.empty(64, 3, 7, 7, dtype=torch.float32, device=x.device)
    nbytes = lib().crw_rn_stem16_ws_bytes()
    ws = _ws(nbytes, x.device)
    _check(lib().crw_rn_stem16_wgrad(_dev(x, "x"), P, cin, _dev(stem, "stem"), _bf(dz[0], "dz_hi"), _bf(dz[1], "dz_lo"), _dev(dw, "dw"),
                                     _ptr(ws), nbytes, _stream()), "crw_rn_stem16_wgrad")
    return dw


def rn_stem16_bwd(x, stem, w0, b0, wt, dz):
    P, cin = x.shape[:2]
    dev = x.device
    dw0 = torch.empty(3, cin, 1, 1, dtype=torch.float32, device=dev)
    db0, dg, db = (torch.empty(3, dtype=torch.float32, device=dev) for _ in range(3))
    nbytes = lib().crw_rn_stem16_ws_bytes()
    ws = _ws(nbytes, dev)
    _check(lib().crw_rn_stem16_bwd(_dev(x, "x"), P, cin, _dev(stem, "stem"), _dev(w0.reshape(3, cin).contiguous(), "w0"), _dev(b0, "b0"),
                                   _bf(wt, "wt"), _bf(dz[0], "dz_hi"), _bf(dz[1], "dz_lo"), _dev(dw0, "dw0"), _dev(db0, "db0"),
                                   _dev(dg, "dg"), _dev(db, "db"), _ptr(ws), nbytes, _stream()), "crw_rn_stem16_bwd")
    return dw0, db0, dg, db


def rn_split(x, P, C):
    Ppad = rn_padded(P)
    hi = torch.empty(Ppad, C, dtype=_BF, device=x.device)
    lo = torch.empty_like(hi)
    _check(lib().crw_rn_split(_dev(x, "x"), P, C, _bf(hi, "hi"), _bf(lo, "lo"), _stream()), "crw_rn_split")
    return hi, lo


def rn_colsum(x):
    rows, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    nbytes = lib().crw_rn_colsum_ws_bytes(C)
    ws = _ws(nbytes, x.device)
    _check(lib().crw_rn_colsum(_dev(x, "x"), rows, C, _dev(out, "out"), _ptr(ws), nbytes, _stream()), "crw_rn_colsum")
    return out


# ---- the whole encoder from native code (crw_rn_train_fwd / _bwd) ------------------------------------------------------------
RN_NPARAM, RN_NBN = 42, 13


class RnTimingRec(ctypes.Structure):
    _fields_ = [("kind", _c_int), ("mode", _c_int), ("g", _c_int * 6), ("k", _c_int), ("stride", _c_int), ("pad", _c_int),
                ("ms", _c_f)]


def _ptr_array(tensors, n):
    if len(tensors) != n:
        raise RuntimeError(f"expected {n} tensors, got {len(tensors)}")
    for t in tensors:
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("Resnet parameters / buffers must be contiguous float32 tensors on the MI355X")
    return (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])


def rn_train_fwd(x, params, run_mean, run_var, momentum, eps, keep=True):
    """x [P,cin,h,w]; params: the 42 parameter tensors in named_parameters() order; run_mean / run_var: the 13 BatchNorm buffer
    pairs in module order (updated in place) -> (out [P,128], workspace kept for rn_train_bwd).  keep=False: no backward pass
    follows (crw_rn_train_fwd_nograd) -> (out, None)."""
    P, cin, h, w = x.shape
    nbytes = lib().crw_rn_train_ws_bytes(P, cin, h, w)
    if nbytes == 0:
        raise RuntimeError(f"crw_rn_train_fwd: unsupported input {tuple(x.shape)}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty(P, 128, dtype=torch.float32, device=x.device)
    fn = lib().crw_rn_train_fwd if keep else lib().crw_rn_train_fwd_nograd
    _check(fn(_dev(x, "x"), P, cin, h, w, _ptr_array(params, RN_NPARAM), _ptr_array(run_mean, RN_NBN),
              _ptr_array(run_var, RN_NBN), float(momentum), float(eps), _dev(out, "out"), _ptr(ws), nbytes,
              _stream()), "crw_rn_train_fwd" if keep else "crw_rn_train_fwd_nograd")
    return out, (ws if keep else None)


def rn_eval_fwd(x, params, run_mean, run_var, eps):
    """the forward with every BatchNorm on its running statistics (module.eval()); nothing is updated -> out [P,128]"""
    P, cin, h, w = x.shape
    nbytes = lib().crw_rn_train_ws_bytes(P, cin, h, w)
    if nbytes == 0:
        raise RuntimeError(f"crw_rn_eval_fwd: unsupported input {tuple(x.shape)}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty(P, 128, dtype=torch.float32, device=x.device)
    _check(lib().crw_rn_eval_fwd(_dev(x, "x"), P, cin, h, w, _ptr_array(params, RN_NPARAM), _ptr_array(run_mean, RN_NBN),
                                 _ptr_array(run_var, RN_NBN), float(eps), _dev(out, "out"), _ptr(ws), nbytes, _stream()),
           "crw_rn_eval_fwd")
    return out


def rn_grad_views(params):
    """the 42 gradient tensors of crw_rn_train_bwd as views of one flat buffer + the two pointer arrays of the call.  Built at
    FORWARD time by the autograd function: the host is ahead of the GPU there, while at the start of the backward pass the GPU has
    just run the walk's few small kernels and would wait ~50 us for these 42 views."""
    sizes = [p.numel() for p in params]
    starts, tot = [], 0
    for n in sizes:
        starts.append(tot)
        tot += (n + 3) // 4 * 4  # 16-byte aligned views
    flat = torch.empty(tot, dtype=torch.float32, device=params[0].device)
    grads = [flat[o:o + n].view(p.shape) for o, n, p in zip(starts, sizes, params)]
    return grads, _ptr_array(params, RN_NPARAM), _ptr_array(grads, RN_NPARAM)


def rn_train_bwd(dout, x, params, ws, prepared=None):
    """-> the 42 gradients (views of one flat buffer), in the order of `params`; prepared: rn_grad_views(params) made earlier"""
    P, cin, h, w = x.shape
    grads, pp, gp = prepared if prepared is not None else rn_grad_views(params)
    _check(lib().crw_rn_train_bwd(_dev(dout, "dout"), _dev(x, "x"), P, cin, h, w, pp, gp, _ptr(ws), ws.numel(), _stream()),
           "crw_rn_train_bwd")
    return grads


def rn_timing(enable):
    _check(lib().crw_rn_timing_enable(int(enable)), "crw_rn_timing_enable")


def rn_timing_read(max_records=100000):
    buf = (RnTimingRec * max_records)()
    n = lib().crw_rn_timing_read(buf, max_records)
    return [buf[i] for i in range(min(n, max_records))]
