"""Factories and inference glue with the reference's names and signatures (src/utils.py):
``create_model``, ``create_dataset``, ``get_reference``, ``pos_embed``, ``propagate``,
``ndiag_matrix``.  Device-agnostic where the reference hard-codes ``'cuda'``.
"""
import os

import torch
import torch.nn.functional as TF
from torch.utils.data import Subset

import crw_hip
from encoder import CNN, Resnet
from dataset import RGDataset, trim_miguel, synthetic_radargram

# dataset id -> (radargram path, reference-segmentation path, number of classes); the reference
# hard-codes these (src/utils.py:30-38, 57-70).  CRW_DATA_ROOT re-roots them.
_DATA = {
    0: ('/data/MCoRDS1_2010_DC8/RG2_MCoRDS1_2010_DC8.pt', '/data/MCoRDS1_2010_DC8/SG2_MCoRDS1_2010_DC8.pt', 4),
    1: ('/datasets/MCORDS1_Miguel/rg2.pt', '/datasets/MCORDS1_Miguel/seg3.pt', 6),
    2: (None, '/data/MCoRDS1_2010_DC8/SG3_MCoRDS1_2010_DC8.pt', 4),
    3: ('/datasets/SHARAD/sharad_north_rg.pt', '/datasets/SHARAD/sharad_north_sg5.pt', 5),
}


def _rooted(path):
    root = os.environ.get('CRW_DATA_ROOT')
    return os.path.join(root, path.lstrip('/')) if root else path


def create_model(id, pos_embed):
    if id == 0:
        return CNN(pos_embed)
    if id == 1:
        return Resnet(pos_embed)
    raise ValueError(f'unknown model id {id} (0 = CNN, 1 = Resnet)')


def create_dataset(id, length, dim, overlap, full=False, flip=False, data_path=None, synthetic=None):
    """id 0/1/3 = MCoRDS1 / MCoRDS3 / SHARAD at the reference's paths.  Extensions: ``data_path``
    (any H x W ``.pt`` radargram) and ``synthetic=(H, W)`` (seeded generator, no file)."""
    if synthetic is not None:
        ds = RGDataset.synthetic(synthetic[0], synthetic[1], length, dim, overlap)
    else:
        path = data_path if data_path is not None else _rooted(_DATA[id][0])
        ds = RGDataset(filepath=path, length=length, dim=dim, overlap=overlap, flip=flip)
    if full:
        return ds
    print('Non-Overlapping dataset! Number of items is the above divided by the length of the sequence...')
    return Subset(ds, range(0, len(ds), length))


def get_reference(id, h, w, flip=False, length=None, dim=None, overlap=None, seg_path=None):
    """-> (nclasses, reference segmentation [h, w or full width]) for dataset ``id``."""
    _, path, nclasses = _DATA[id]
    data = torch.load(seg_path if seg_path is not None else _rooted(path))
    if id == 1:
        data = trim_miguel(data, length, dim)
    data = data[:h, :] if w == 0 else data[:h, :w]
    return (nclasses, torch.flip(data, (1,))) if flip else (nclasses, data)


def pos_embed(seq):
    """[P,1,H,W] -> [P,2,H,W]: prepends a channel holding the row position r/H - 0.5."""
    P, _, H, W = seq.shape
    pe = (torch.arange(H, device=seq.device, dtype=torch.float32) / H - 0.5).view(1, 1, H, 1)
    return torch.cat([pe.expand(P, 1, H, W).to(seq.dtype), seq], dim=1)


def seed_labels(seg_ref, N):
    """Nearest-neighbour resize of the reference segmentation [rows, w] to one label per node
    (torchvision Resize((N,1), NEAREST) in the reference, src/utils.py:139-141)."""
    return TF.interpolate(seg_ref[None, None].float(), size=(N, 1), mode='nearest')[0, 0, :, 0]


def column_diffs_async(xent):
    """|xent[:, i] - xent[:, i+1]| summed over the nodes (src/utils.py:125), copied to pinned host memory WITHOUT
    blocking: -> (host tensor, event).  Lets the caller queue more GPU work before the host needs the values."""
    d = (xent[:, :-1] - xent[:, 1:]).abs().sum(0)
    host = torch.empty(d.shape, dtype=d.dtype, pin_memory=True)
    host.copy_(d, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev


def change_point(xent, diffs=None):
    """PELT(rbf, pen=5) on the column-to-column change of the metric, then `result[-2] + 5` clamped at 0
    (src/utils.py:125-132).  Uses `ruptures` when it is importable, else the restatement of its published
    algorithm in pelt.py (parity unpinned, see there); like the reference, any failure -- e.g. fewer than
    two breakpoints -- yields None.  (`crw_hip.pelt_rbf` is pelt.py's arithmetic as a host function of the library: the numpy
    form took 0.9 ms per radargram and had become the tail of the config-5 step.)  diffs: the (host, event) pair of `column_diffs_async` if already requested."""
    if diffs is not None:
        diffs[1].synchronize()
        diffs = diffs[0].numpy()
    else:
        diffs = (xent[:, :-1] - xent[:, 1:]).abs().sum(0).cpu().numpy()
    try:
        try:
            import ruptures as rpt
            result = rpt.Pelt(model="rbf").fit(diffs).predict(pen=5)
        except ImportError:  # pelt.py's restatement, as the library's host function (csrc/pelt.cpp: same sums, plain C++)
            result = crw_hip.pelt_rbf(diffs, pen=5)
        return max(0, int(result[-2] + 5))
    except Exception:
        return None


@torch.no_grad()
def propagate(seq, seg_ref, model, lp, nclasses, do_pos_embed, use_last):
    """seq [T,N,h,w]; seg_ref [rows, w] class ids of the first (or last) frame; model: encoder;
    lp: LabelPropVOS_CRW  ->  (labels [N,T] float, xent [N,T-1] (CPU), change_idx | None)."""
    T, N, H, W = seq.shape
    if use_last:
        seq = torch.flip(seq, (0,))
    x = seq.reshape(-1, H, W).unsqueeze(1)
    if do_pos_embed:
        x = pos_embed(x)
    emb = model(x).reshape(T, N, -1).float().contiguous()
    feats = crw_hip.normalize(emb)
    seed = seed_labels(seg_ref.to(feats.device), N)
    if T == 1:  # a one-frame item (the correction step of test_all.py can ask for it): nothing to propagate, like the reference
        return seed[:, None].clone(), torch.zeros(N, 0), None
    xent = crw_hip.xent_metric(feats)
    diffs = column_diffs_async(xent) if T > 2 else None  # on its way to the host before the label propagation is queued
    if hasattr(lp, 'propagate_all'):
        pred, _ = lp.propagate_all(feats, seed, nclasses)
    else:  # foreign label-propagation object: reference's frame-by-frame protocol
        pred = torch.zeros(N, T, device=feats.device)
        pred[:, 0] = seed
        mask = (seed[None, :] == torch.arange(nclasses, device=feats.device)[:, None]).float()[None, :, :, None]
        as_feat = lambda n: feats[n].t()[None, :, :, None]
        fl, ml = [as_feat(0)], [mask]
        for n in range(1, T):
            mask = lp.predict(feats=fl, masks=ml, curr_feat=as_feat(n))
            fl.append(as_feat(n))
            ml.append(mask)
            pred[:, n] = mask.argmax(1).squeeze()
    # the change point is host work (PELT on T-2 samples): it runs while the GPU propagates the labels queued above
    change_idx = change_point(xent, diffs)
    return pred, xent.cpu(), change_idx


def ndiag_matrix(size, n=1):
    """Row-normalised band matrix (n <= 2: identity, 3: tri-diagonal, ...)."""
    i = torch.arange(size)
    m = ((i[:, None] - i[None, :]).abs() <= max(n - 2, 0)).float()
    return m / m.sum(dim=1, keepdim=True)
