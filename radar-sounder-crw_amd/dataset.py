"""``RGDataset`` -- cuts a radargram (2-D float tensor H x W, a ``.pt`` file or a tensor) into
items of ``length`` consecutive patch-columns: item ``i`` is a [T, N, h, w] view starting at
column ``(w-ow)*i``.  Same constructor, ``__len__``, ``__getitem__`` and ``get_smaller_item`` as
the reference (src/dataset.py:5-47), plus ``from_tensor`` / ``synthetic`` sources because the
reference's data paths are private.
"""
import math

import torch
from torch.utils.data import Dataset

MIGUEL_SPLITS = (9984, 6656, 9984, 20000, 16640, 32864, 8992)


def trim_miguel(T, length, dim):
    """Keep, from each of the 7 concatenated MCoRDS3 radargrams, the largest prefix whose width
    is a multiple of one item (src/dataset.py:66-80)."""
    item_w = dim[1] * length
    out, start = [], 0
    for L in MIGUEL_SPLITS:
        keep = (L // item_w) * item_w
        out.append(T[:, start:start + keep])
        start += L
    return torch.cat(out, dim=1)


def synthetic_radargram(H, W, seed=11, layered=True):
    """Deterministic synthetic radargram: N(0,1) speckle plus (optionally) smooth sub-horizontal
    layering so that features are not pure noise."""
    g = torch.Generator().manual_seed(seed)
    rg = torch.randn(H, W, generator=g)
    if layered:
        r = torch.arange(H).view(H, 1).float()
        c = torch.arange(W).view(1, W).float()
        rg = rg * 0.6 + torch.sin(2 * math.pi * r / 64 + 0.002 * c)
    return rg.float()


class RGDataset(Dataset):
    def __init__(self, filepath='/data/MCoRDS1_2010_DC8/RG2_MCoRDS1_2010_DC8.pt', length=10, dim=(24, 24),
                 overlap=(0, 0), flip=False, tensor=None):
        self.filepath = filepath
        self.l = length
        self.T = tensor if tensor is not None else torch.load(filepath)
        if tensor is None and str(filepath).endswith('rg2.pt'):
            self.T = trim_miguel(self.T, length, dim)
            print('Trimmed Dataset to match radargram sizes!')
        if flip:
            self.T = torch.flip(self.T, dims=(1,))
        H, W = self.T.shape
        self.h, self.w = dim
        self.oh, self.ow = overlap
        sh, sw = self.h - self.oh, self.w - self.ow
        self.nh = (H - self.oh) // sh                       # nodes (vertical patches) per frame
        self.pxw = length * self.w - self.ow * (length - 1)  # radargram columns spanned by one item
        self.pxh = self.nh * self.h - self.oh * (self.nh - 1)
        self.nw = (W - self.pxw) // sw + 1                  # number of items
        print('Total items:', self.nw, 'Length of item in pixels:', self.pxw)

    @classmethod
    def from_tensor(cls, tensor, length, dim, overlap, flip=False):
        return cls(filepath='<tensor>', length=length, dim=dim, overlap=overlap, flip=flip, tensor=tensor)

    @classmethod
    def synthetic(cls, H, W, length, dim, overlap, seed=11):
        return cls.from_tensor(synthetic_radargram(H, W, seed), length, dim, overlap)

    def __len__(self):
        return self.nw

    def _cut(self, index, pxw):
        c0 = (self.w - self.ow) * index
        block = self.T[:self.pxh, c0:c0 + pxw]
        patches = block.unfold(0, self.h, self.h - self.oh).unfold(1, self.w, self.w - self.ow)  # [N, T, h, w]
        return patches.permute(1, 0, 2, 3).float()

    def __getitem__(self, index):
        return self._cut(index, self.pxw)

    def columns(self, first=0, count=None):
        """Patch-columns ``first .. first+count-1`` of the radargram as one [Tc, N, h, w] tensor (default: all
        ``len(self) + length - 1`` of them).  Item ``i`` of the dataset is ``columns()[i : i + length]``, so
        overlapping items can share one encoder pass (``CRW.forward_columns``, SURVEY.md section 8 row f1;
        the reference re-encodes every item, src/dataset.py:19-39 + scripts/train.py:66-67)."""
        total = len(self) + self.l - 1
        count = total - first if count is None else count
        if first < 0 or count < 1 or first + count > total:
            raise IndexError(f'columns [{first}, {first + count}) outside [0, {total})')
        c0 = (self.w - self.ow) * first
        pxw = count * self.w - self.ow * (count - 1)
        block = self.T[:self.pxh, c0:c0 + pxw]
        patches = block.unfold(0, self.h, self.h - self.oh).unfold(1, self.w, self.w - self.ow)  # [N, Tc, h, w]
        return patches.permute(1, 0, 2, 3).float()

    def get_smaller_item(self, index, small_length):
        # like the reference this permanently shortens the item length of the dataset
        self.small_pxw = self.pxw = small_length * self.w - self.ow * (small_length - 1)
        return self._cut(index, self.pxw)
