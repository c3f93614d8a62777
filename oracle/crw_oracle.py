"""CPU oracle for the contrastive-random-walk hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (numpy + torch-CPU) of the algorithm that
jdalcorso/radar-sounder-crw runs on its hot path.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product
path (``radar-sounder-crw_amd/``) never imports it and fails loudly when the HIP library is missing.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py`` against
fixtures in ``tests/golden/`` that were produced by importing and running the reference itself on
CPU in the build container (``tests/golden/make_golden.py``; the reference repo has no tests or
golden vectors of its own).

Reference lines each function follows (paths under the reference repo):
  l2_normalize ............ src/model.py:22            (F.normalize, eps 1e-12)
  affinity ................ src/model.py:26
  walk_reference_form ..... src/model.py:31-46         (literal palindrome / O(T^2) chain)
  walk_prefix_form ........ same value, O(T) products  
  walk_backward ........... analytic gradient of the above (checked against reference autograd)
  cnn_forward ............. src/encoder.py:13-57
  pos_embed ............... src/utils.py:76-90
  crw_forward ............. src/model.py:15-46
  band_bias ............... src/imported/maskedatt.py:222-245 + labelprop.py:89-96 (w == 1)
  labelprop_weights ....... src/imported/maskedatt.py:151-175
  labelprop ............... src/utils.py:134-161 + src/imported/labelprop.py:67-115
  labelprop_tie_audit ..... same lines, fp64, teacher-forced per frame (near-tie proof for label maps)
  seed_labels ............. src/utils.py:139-147       (NEAREST resize to (N,1))
  xent_metric ............. src/utils.py:117-125
  unfold_item ............. src/dataset.py:19-39
"""
import numpy as np

EPS_NORM = 1e-12


# --------------------------------------------------------------------------------------------
# training forward
# --------------------------------------------------------------------------------------------
def l2_normalize(e, dtype=np.float64):
    e = np.asarray(e, dtype=dtype)
    nrm = np.sqrt((e * e).sum(-1, keepdims=True))
    return e / np.maximum(nrm, EPS_NORM)


def affinity(ehat, tau):
    """A[b,j,n,m] = <ehat[b,j,n], ehat[b,j+1,m]> / tau ; ehat [B,T,N,C] -> [B,T-1,N,N]."""
    return np.einsum("btnc,btmc->btnm", ehat[:, :-1], ehat[:, 1:]) / ehat.dtype.type(tau)


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def _cycle_loss_terms(At):
    """-(1/(B*N)) sum_{b,d} log softmax(At[b,d,:])[d]   (cross_entropy with class dim 1 on At^T)."""
    B, N, _ = At.shape
    m = At.max(-1, keepdims=True)
    lse = np.log(np.exp(At - m).sum(-1)) + m[..., 0]
    diag = np.einsum("bdd->bd", At)
    return -(diag - lse).sum() / (B * N)


def walk_reference_form(A):
    """Literal restatement: palindrome AA, for each k a fresh chain of 2k-1 left-multiplications."""
    B, Tm1, N, _ = A.shape
    T = Tm1 + 1
    AA = np.concatenate([A, np.flip(A, 1).transpose(0, 1, 3, 2)], 1)  # [B, 2T-2, N, N]
    loss = A.dtype.type(0)
    Ats = []
    for k in range(1, T - 1):
        At = np.broadcast_to(np.eye(N, dtype=A.dtype), (B, N, N)).copy()
        AA_this = np.concatenate([AA[:, :k], AA[:, -k:]], 1) if k > 0 else AA[:, :0]
        for t in range(1, 2 * k):
            At = _softmax(AA_this[:, t], -1) @ At
        Ats.append(At)
        loss = loss + _cycle_loss_terms(At)
    At_all = np.stack(Ats, 1) if Ats else np.zeros((B, 0, N, N), A.dtype)
    return loss / N, At_all


def softmax_pair(A):
    """F_j = row-softmax(A_j);  Gt_j = column-softmax(A_j) kept in A's layout (G_j = Gt_j^T)."""
    return _softmax(A, -1), _softmax(A, -2)


def walk_prefix_form(A, return_state=False):
    """Lt_1 = Gt_0, Lt_{k+1} = Gt_k Lt_k ; R_1 = I, R_{k+1} = F_k R_k ; At_k = Lt_k^T R_k."""
    B, Tm1, N, _ = A.shape
    T = Tm1 + 1
    F, Gt = softmax_pair(A)
    Lt, R, Ats = [], [], []
    loss = A.dtype.type(0)
    for k in range(1, T - 1):
        if k == 1:
            Lt.append(Gt[:, 0].copy())
            R.append(np.broadcast_to(np.eye(N, dtype=A.dtype), (B, N, N)).copy())
        else:
            Lt.append(Gt[:, k - 1] @ Lt[-1])
            R.append(F[:, k - 1] @ R[-1])
        At = Lt[-1].transpose(0, 2, 1) @ R[-1]
        Ats.append(At)
        loss = loss + _cycle_loss_terms(At)
    At_all = np.stack(Ats, 1) if Ats else np.zeros((B, 0, N, N), A.dtype)
    if return_state:
        return loss / N, At_all, dict(F=F, Gt=Gt, Lt=Lt, R=R)
    return loss / N, At_all


def walk_backward(A, gloss=1.0):
    """dLoss/dA by hand (reverse of the prefix recurrences, Appendix A.3), [B,T-1,N,N]."""
    B, Tm1, N, _ = A.shape
    T = Tm1 + 1
    _, At_all, st = walk_prefix_form(A, return_state=True)
    F, Gt, Lt, R = st["F"], st["Gt"], st["Lt"], st["R"]
    dF = np.zeros_like(A)
    dGt = np.zeros_like(A)
    K = T - 2
    if K < 1:
        return np.zeros_like(A)
    coef = gloss / (N * B * N)
    dLt_next = dR_next = None
    for k in range(K, 0, -1):  # k = K..1 ; lists are 0-based (index k-1)
        At = At_all[:, k - 1]
        dAt = coef * (_softmax(At, -1) - np.eye(N, dtype=A.dtype))
        dLt = R[k - 1] @ dAt.transpose(0, 2, 1)
        dR = Lt[k - 1] @ dAt
        if k < K:
            dLt = dLt + Gt[:, k].transpose(0, 2, 1) @ dLt_next
            dGt[:, k] = dLt_next @ Lt[k - 1].transpose(0, 2, 1)
            dR = dR + F[:, k].transpose(0, 2, 1) @ dR_next
            dF[:, k] = dR_next @ R[k - 1].transpose(0, 2, 1)
        dLt_next, dR_next = dLt, dR
    dGt[:, 0] = dLt_next
    dA = F * (dF - (dF * F).sum(-1, keepdims=True)) + Gt * (dGt - (dGt * Gt).sum(-2, keepdims=True))
    return dA


def affinity_backward(dA, e, tau):
    """dLoss/d(raw features) through affinity + L2 normalisation."""
    e = np.asarray(e, dtype=dA.dtype)
    nrm = np.maximum(np.sqrt((e * e).sum(-1, keepdims=True)), EPS_NORM)
    eh = e / nrm
    deh = np.zeros_like(eh)
    deh[:, :-1] += np.einsum("btnm,btmc->btnc", dA, eh[:, 1:]) / tau
    deh[:, 1:] += np.einsum("btnm,btnc->btmc", dA, eh[:, :-1]) / tau
    return (deh - eh * (eh * deh).sum(-1, keepdims=True)) / nrm


def crw_from_features(emb, tau, dtype=np.float64):
    """emb [B,T,N,C] raw encoder outputs -> dict(loss, A, At, demb)."""
    eh = l2_normalize(emb, dtype)
    A = affinity(eh, tau)
    loss, At = walk_prefix_form(A)
    dA = walk_backward(A)
    demb = affinity_backward(dA, np.asarray(emb, dtype), tau)
    return dict(loss=loss, A=A, At=At, dA=dA, demb=demb)


# --------------------------------------------------------------------------------------------
# encoder (torch CPU; the oracle may use torch ops, it is never the thing measured on the GPU)
# --------------------------------------------------------------------------------------------
def pos_embed(x):
    """x [P,1,h,w] torch -> [P,2,h,w]; prepended channel pe[r,:] = r/h - 0.5."""
    import torch
    P, _, h, w = x.shape
    pe = (torch.arange(h, dtype=torch.float32) / h - 0.5).view(1, 1, h, 1).expand(P, 1, h, w)
    return torch.cat([pe.to(x.dtype), x], 1)


def cnn_forward(x, sd):
    """x [P,cin,h,w] torch, sd: dict name -> torch tensor (conv{1..5}.{weight,bias}, fc.*)."""
    import torch.nn.functional as TF
    x = TF.max_pool2d(TF.relu(TF.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1)), 2, 1)
    x = TF.max_pool2d(TF.relu(TF.conv2d(x, sd["conv2.weight"], sd["conv2.bias"], padding=1)), 2, 1)
    x = TF.relu(TF.conv2d(x, sd["conv3.weight"], sd["conv3.bias"], padding=1))
    x = TF.relu(TF.conv2d(x, sd["conv4.weight"], sd["conv4.bias"], padding=1))
    x = TF.relu(TF.conv2d(x, sd["conv5.weight"], sd["conv5.bias"], padding=1))
    x = x.mean((2, 3))
    return TF.linear(x, sd["fc.weight"], sd["fc.bias"])


def crw_forward_torch(seq, sd, tau, use_pos_embed=False, dtype=None):
    """Whole training forward in torch-CPU with autograd (used for weight-gradient parity and as
    the timed CPU baseline): seq [B,T,N,h,w] -> (loss, A, emb)."""
    import torch
    B, T, N, h, w = seq.shape
    x = seq.reshape(-1, h, w).unsqueeze(1)
    if use_pos_embed:
        x = pos_embed(x)
    emb = cnn_forward(x, sd).reshape(B, T, N, -1)
    loss, A = walk_loss_torch(emb, tau)
    return loss, A, emb


def walk_loss_torch(emb, tau, At_out=None):
    """Prefix-form walk in torch (differentiable, any device / dtype); emb [B,T,N,C] raw -> (loss, A).  `At_out`: a list that
    receives every cycle product At_k (detached), k = 1..T-2."""
    import torch
    B, T, N, C = emb.shape
    eh = emb / emb.norm(dim=-1, keepdim=True).clamp_min(EPS_NORM)
    A = torch.einsum("btnc,btmc->btnm", eh[:, :-1], eh[:, 1:]) / tau
    if T < 3:
        return emb.new_zeros(()), A
    F = torch.softmax(A, -1)
    Gt = torch.softmax(A, -2)
    eye = torch.eye(N, dtype=emb.dtype, device=emb.device)
    loss = emb.new_zeros(())
    Lt = R = None
    for k in range(1, T - 1):
        if k == 1:
            Lt, R = Gt[:, 0], eye.expand(B, N, N)
        else:
            Lt, R = Gt[:, k - 1] @ Lt, F[:, k - 1] @ R
        At = Lt.transpose(1, 2) @ R
        if At_out is not None:
            At_out.append(At.detach())
        lse = torch.logsumexp(At, -1)
        loss = loss - (torch.diagonal(At, dim1=1, dim2=2) - lse).sum() / (B * N)
    return loss / N, A


# --------------------------------------------------------------------------------------------
# inference: label propagation
# --------------------------------------------------------------------------------------------
MASK_NEG = -1e10


def band_bias(N, radius, dtype=np.float32, grid_w=1):
    """bias[m,q] = 0 if the nodes m and q are closer than `radius` else -1e10.  The N nodes form an (N / grid_w) x grid_w grid in
    row-major order and the distance is Euclidean (MaskedAttention.make, src/imported/maskedatt.py:232-245); a radargram's patch
    grid is N x 1 (grid_w = 1), where the mask degenerates to the band |m-q| < radius."""
    i, j = np.arange(N) // grid_w, np.arange(N) % grid_w
    d = np.sqrt(((i[:, None] - i[None, :]) ** 2 + (j[:, None] - j[None, :]) ** 2).astype(np.float32))
    return np.where(d < radius, 0.0, MASK_NEG).astype(dtype)


def seed_labels(seg_ref, N):
    """NEAREST resize of seg_ref [rows, w] to (N,1): row floor(i*rows/N), column 0."""
    seg_ref = np.asarray(seg_ref)
    rows = seg_ref.shape[0]
    idx = np.floor(np.arange(N) * (rows / N)).astype(np.int64)
    idx = np.minimum(idx, rows - 1)
    return seg_ref[idx, 0].astype(np.float32)


def labelprop_weights(ehat, n, cxt_size, radius, temp, knn, dtype=np.float32, grid_w=1):
    """Top-k neighbour weights/indices of frame n against frames 0..n-1.
    Returns (W [knn,N], I [knn,N]) with I addressing the (possibly truncated) key list."""
    T, N, C = ehat.shape
    keys = ehat[:n].reshape(n * N, C).astype(dtype)
    q = ehat[n].astype(dtype)
    S = (keys @ q.T).reshape(n, N, N) + band_bias(N, radius, dtype, grid_w)[None]
    S = S.reshape(n * N, N) / dtype(temp)
    if S.shape[0] > (cxt_size + 1) * N:
        S = np.concatenate([S[:N], S[-N * cxt_size:]], 0)
    # top-k along keys, descending (ties: any order -- masked keys carry weight exactly 0)
    I = np.argsort(-S, axis=0, kind="stable")[:knn]
    Wl = np.take_along_axis(S, I, 0)
    Wl = np.exp(Wl - Wl.max(0, keepdims=True))
    W = Wl / Wl.sum(0, keepdims=True)
    return W.astype(dtype), I


def labelprop(emb, seed, nclasses, cxt_size, radius, temp, knn, dtype=np.float32, grid_w=1):
    """emb [T,N,C] raw features (already flipped by the caller if use_last), seed [N] float labels
    of frame 0 -> pred [N,T] float labels.  Indices returned for the truncated key list are used
    against the *untruncated* label list (quirk Q7).  grid_w: see band_bias."""
    T, N, C = emb.shape
    ehat = l2_normalize(emb, dtype).astype(dtype)
    L = np.zeros((T * N, nclasses), dtype)
    L[np.arange(N), :] = (seed[:, None] == np.arange(nclasses)[None, :]).astype(dtype)
    pred = np.zeros((N, T), np.float32)
    pred[:, 0] = seed
    for n in range(1, T):
        W, I = labelprop_weights(ehat, n, cxt_size, radius, temp, knn, dtype, grid_w)
        p = (L[I] * W[..., None]).sum(0)  # [N, M]
        L[n * N:(n + 1) * N] = p
        pred[:, n] = p.argmax(-1)
    return pred


def labelprop_tie_audit(ehat, L_dev, pred_dev, cxt_size, radius, temp, knn, eps=1e-5):
    """Frame-by-frame (teacher-forced) fp64 audit of a propagated label map -- how the tests prove that every label
    on which a device run and the fp32 oracle disagree is a floating-point near-tie, not a defect.

    Label propagation is argmax / top-k over floating-point scores, so two correct fp32 implementations (different
    summation orders) can legitimately pick different labels where two class probabilities or the k-th and (k+1)-th key
    logits coincide to rounding, and one such flip then cascades through every later frame that uses the flipped
    labels as context.  This audit removes the cascade: for every frame n it recomputes, in fp64 and with the DEVICE's
    own soft labels of frames < n as context (``L_dev`` [T*N, M]; same index quirk Q7 as ``labelprop``), the scores of
    frame n from the device's normalised features ``ehat`` [T,N,C], and compares argmax with ``pred_dev`` [N,T].

    Returns dict(step_mismatches, not_ties, worst_margin, max_soft_err): a step mismatch is a *tie* when the fp64
    top-2 class-probability margin is < eps or the top-k boundary logit gap (k-th minus (k+1)-th key logit of that
    query) is < eps; ``not_ties`` must be 0.  ``max_soft_err`` = max |L_dev - fp64 soft labels| over the queries
    without a boundary tie (the gathered weights themselves)."""
    T, N, C = ehat.shape
    eh = np.asarray(ehat, np.float64)
    L = np.asarray(L_dev, np.float64)
    M = L.shape[1]
    bias = band_bias(N, radius, np.float64)
    out = dict(step_mismatches=0, not_ties=0, worst_margin=0.0, max_soft_err=0.0, boundary_ties=0)
    for n in range(1, T):
        S = (eh[:n].reshape(n * N, C) @ eh[n].T).reshape(n, N, N) + bias[None]
        S = S.reshape(n * N, N) / temp
        if S.shape[0] > (cxt_size + 1) * N:
            S = np.concatenate([S[:N], S[-N * cxt_size:]], 0)
        order = np.argsort(-S, axis=0, kind="stable")
        I = order[:knn]
        top = np.take_along_axis(S, I, 0)
        if S.shape[0] > knn:
            nxt = np.take_along_axis(S, order[knn:knn + 1], 0)[0]
            gap = top[-1] - nxt                      # masked keys sit at -1e10/temp: a huge gap, never a tie
        else:
            gap = np.full(N, np.inf)
        # ties INSIDE the selected set do not matter (same keys, same weights); only the boundary does
        Wl = np.exp(top - top.max(0, keepdims=True))
        W = Wl / Wl.sum(0, keepdims=True)
        p = (L[I] * W[..., None]).sum(0)             # [N, M]
        ps = np.sort(p, -1)
        margin = ps[:, -1] - ps[:, -2] if M > 1 else np.full(N, np.inf)
        lab = p.argmax(-1)
        dev = np.asarray(pred_dev[:, n]).astype(np.int64)
        boundary_tie = gap < eps
        out["boundary_ties"] += int(boundary_tie.sum())
        clean = ~boundary_tie
        if clean.any():
            out["max_soft_err"] = max(out["max_soft_err"], float(np.abs(L[n * N:(n + 1) * N][clean] - p[clean]).max()))
        bad = lab != dev
        # a device label different from the fp64 argmax is fine when the device's class is within eps of the best one
        dev_short = ps[:, -1] - p[np.arange(N), dev]
        out["step_mismatches"] += int(bad.sum())
        nt = bad & ~boundary_tie & ~(dev_short < eps)
        out["not_ties"] += int(nt.sum())
        if bad.any():
            out["worst_margin"] = max(out["worst_margin"], float(np.where(boundary_tie, 0.0, dev_short)[bad].max()))
    return out


def xent_metric(emb, dtype=np.float32):
    """'Horizontality' metric: within-frame affinity on channel-shifted features (quirk Q8),
    temperature 0.1, CE of A_i^T against the identity -> xent [N, T-1]."""
    T, N, C = emb.shape
    eh = l2_normalize(emb, dtype).astype(dtype)
    A = np.einsum("tnc,tmc->tnm", eh[:, :, :-1], eh[:, :, 1:]) / dtype(0.1)
    out = np.zeros((N, T - 1), np.float32)
    for i in range(T - 1):
        X = A[i].T  # input [N(batch), N(class)]
        m = X.max(-1, keepdims=True)
        lse = np.log(np.exp(X - m).sum(-1)) + m[:, 0]
        out[:, i] = -(np.diag(X) - lse)
    return out


# --------------------------------------------------------------------------------------------
# dataset
# --------------------------------------------------------------------------------------------
def dataset_geometry(H, W, length, dim, overlap):
    h, w = dim
    oh, ow = overlap
    nh = (H - oh) // (h - oh)
    pxw = length * w - ow * (length - 1)
    nw = (W - (length * (w - ow) + ow)) // (w - ow) + 1
    pxh = nh * h - oh * (nh - 1)
    return nh, nw, pxh, pxw


def unfold_item(rg, index, length, dim, overlap):
    h, w = dim
    oh, ow = overlap
    nh, nw, pxh, pxw = dataset_geometry(rg.shape[0], rg.shape[1], length, dim, overlap)
    out = np.empty((length, nh, h, w), np.float32)
    c_item = (w - ow) * index
    for t in range(length):
        for n in range(nh):
            out[t, n] = rg[n * (h - oh):n * (h - oh) + h, c_item + t * (w - ow):c_item + t * (w - ow) + w]
    return out
