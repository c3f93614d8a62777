"""Whole-radargram inference on top of ``utils.propagate`` -- the driver loop of the reference's
evaluation script (scripts/test/test_all.py:71-159) without its plotting / report / private-data parts:

  * forward pass: every non-overlapping item is seeded with the reference segmentation of its first
    patch column and propagated along-track (test_all.py:91-100);
  * optional correction: where ``propagate`` reports a change point, a shortened item is re-seeded
    and its map replaces the tail of the radargram's map (test_all.py:103-122);
  * optional reverse pass: the items propagated from their LAST column (``use_last``) and merged
    into the forward map by a class rule (test_all.py:132-159).

The label maps are the reference's, operation for operation (pinned by ``tests/golden/segment_*.npz``,
which ``tests/golden/make_golden.py`` produces by running the reference's own ``main(args)``), which
includes three things one might not expect:
  * the correction step cuts its shortened item with ``dataset.get_smaller_item(index, small_length)``
    (src/dataset.py:41-47): the FIRST ``small_length`` patch-columns of the item -- not its tail -- while the
    seed labels and the overwritten map columns are the tail's (test_all.py:112-119);
  * ``get_smaller_item`` permanently shortens the dataset's item length, so a reverse pass that follows a
    correction runs on items of the last corrected length (their maps are stretched back to ``rg_len``);
  * the forward pass seeds from ``seg[:rg_h]`` (test_all.py:94), the correction and reverse passes from every
    row of ``seg`` (test_all.py:115,141);
  * any error inside a correction is swallowed (bare ``except``, test_all.py:121-122).
Label maps are upsampled to pixels with nearest-neighbour interpolation like the reference
(``transforms.Resize(NEAREST)``).  Everything heavy runs in ``propagate`` (encoder + HIP kernels).
"""
import torch
import torch.nn.functional as TF

import crw_hip
from utils import propagate


def _upsample(pred, rows, cols):
    """[N, T] node labels -> [rows, cols] pixel labels (nearest)."""
    return TF.interpolate(pred[None, None].float(), size=(rows, cols), mode='nearest')[0, 0]


def merge_reverse(final_pred, pred_rev, dataset_id):
    """Class-specific merge of the reversed pass into the forward map (test_all.py:146-159):
    class 2 (bedrock) of the reversed pass wins, with per-dataset restrictions."""
    rows = pred_rev.shape[0]
    mask = pred_rev.flatten() == 2
    if dataset_id == 1:
        mask = torch.logical_and(mask, final_pred.flatten() != 3)
        under_ice = torch.all(pred_rev != 4, dim=0)[None].repeat(rows, 1).flatten()
        mask = torch.logical_and(mask, under_ice)
    elif dataset_id == 3:
        mask = mask.clone()
        mask[:mask.numel() // 2] = False
    elif dataset_id != 0:
        raise ValueError(f'no merge rule for dataset id {dataset_id} (the reference defines 0, 1 and 3)')
    out = final_pred.flatten().clone()
    out[mask] = 2
    return out.view_as(final_pred)


@torch.no_grad()
def segment(dataset, seg, encoder, lp, nclasses, seq_length, patch_size, overlap, pos_embed=False,
            correction=False, use_last=False, dataset_id=0, device='cuda'):
    """dataset: RGDataset (full, overlapping items); seg: reference segmentation [rows, W_rg].
    -> dict(pred [rows, n_rg * rg_len] float labels after the optional reverse merge,
            forward: the forward (+ corrected) map the reference saves as int8 (test_all.py:128),
            xent list, change_idx list)."""
    T, (H, W), (oh, ow) = seq_length, patch_size, overlap
    N = dataset[0].shape[1]
    rg_len = T * (W - ow) + ow
    rg_h = N * (H - oh) + oh
    idx = list(range(0, len(dataset), T))
    n_rg = min(len(idx), seg.shape[-1] // rg_len)
    idx = idx[:n_rg]
    seg = seg[:, :n_rg * rg_len].to(device)
    rows = seg.shape[0]

    maps, xents, changes = [], [], []
    for t, i in enumerate(idx):
        seq = dataset[i].to(device)
        seg_ref = seg[:rg_h, rg_len * t:rg_len * t + W]
        pred, xent, change = propagate(seq, seg_ref, encoder, lp, nclasses, pos_embed, use_last=False)
        maps.append(_upsample(pred, rows, rg_len))
        xents.append(xent)
        changes.append(change)

    if correction:
        for t, change in enumerate(changes):
            if change is None:
                continue
            small = T - change
            px = small * (W - ow)
            try:  # like the reference, a correction that fails on its DATA (shape / index errors) is skipped silently ...
                seq = dataset.get_smaller_item(idx[t], small).to(device)  # first `small` columns; shortens the dataset
                seg_ref = seg[:, rg_len * t + rg_len - px:rg_len * t + rg_len - px + W]
                pred, _, _ = propagate(seq, seg_ref, encoder, lp, nclasses, pos_embed, use_last=False)
                maps[t][:, rg_len - px:] = _upsample(pred, rows, px)
            except crw_hip.CrwError as e:
                # ... but a failure of the HIP path itself (CRW_EHIP: launch failure / GPU fault, CRW_EWORKSPACE) is not a data
                # problem: the reference's bare `except` would hide a poisoned device context behind an uncorrected map.
                # CRW_EINVAL (a degenerate correction window: bad shape / unsupported size) IS the data and is skipped
                if e.device_failure:
                    raise
            except torch.AcceleratorError:  # the device runtime's own errors (hipError* raised by PyTorch)
                raise
            except Exception:
                pass

    forward = torch.cat(maps, dim=1)
    final = forward
    if use_last:
        rev_maps = []
        seg_rev = torch.flip(seg.unfold(1, rg_len, rg_len), (-1,)).reshape(rows, -1)
        for t, i in enumerate(idx):
            seq = dataset[i].to(device)
            seg_ref = seg_rev[:, rg_len * t:rg_len * t + W]
            pred, _, _ = propagate(seq, seg_ref, encoder, lp, nclasses, pos_embed, use_last=True)
            rev_maps.append(_upsample(pred, rows, rg_len))
        rev = torch.cat(rev_maps, dim=1).unfold(1, rg_len, rg_len)
        rev = torch.flip(rev, (-1,)).reshape(rows, -1)
        final = merge_reverse(forward, rev, dataset_id)
    return dict(pred=final, forward=forward, xent=xents, change_idx=changes)
