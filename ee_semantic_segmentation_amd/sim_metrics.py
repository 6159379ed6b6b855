"""Similarity gates between the label maps of two exits (reference ``sim_metrics.py``: MSE :41-58, NMI :60-76,
VI :78-99, Seg_comp :101-120), computed on the device from the per-image C x C contingency table that
``eeseg_argmax_pair_hist`` builds while upsampling both exits - no label map or probability tensor is
materialised or copied to the host (the reference moves both maps to numpy and calls scikit-image per image).

scikit-image is not installed in the build image and the reference holds no fixture for these functions, so
their parity is *unpinned*: the formulas are the published scikit-image ones (see oracle/sim_ref.py).  SSIM
(:15-37) is a windowed statistic of the label IMAGE, not a function of the table: both exits' label maps are
produced by the fused upsample+argmax kernel and ``eeseg_ssim_labels`` reduces them on the device.

Every metric takes ``(a, b)`` = two ``[1,C,H,W]`` score tensors, two ``[H,W]`` label maps, or an ``ExitLogits``
plus two exit indices via ``pair_table`` and returns a python float, like the reference.
"""
import math

import torch

from . import engine as E
from . import kernels as K
from .from_deepv3_new import ExitLogits


def _as_lowres(t):
    """[1,C,H,W] scores -> ([1,H,W,CPAD] fp32 'low-res' logits at full size, C)."""
    _, C, H, W = t.shape
    lr = torch.zeros((1, H, W, E.CPAD), dtype=torch.float32, device=t.device)
    lr[..., :C] = t.detach().permute(0, 2, 3, 1)
    return lr, C, (H, W)


def pair_table(a, b, exit_a=None, exit_b=None, num_classes=None):
    """float64 [C,C] contingency table of one image: rows = labels of `a`, columns = labels of `b`."""
    if isinstance(a, ExitLogits):
        assert a.lowres[exit_a].shape[0] == 1, "one image at a time (as the reference)"
        hist = K.argmax_pair_hist(a.lowres[exit_a].detach().contiguous(), a.lowres[exit_b].detach().contiguous(),
                                  a.num_classes, *a.size)
        return hist[0].double()
    if a.dim() == 4:
        la, C, size = _as_lowres(a)
        lb, _, _ = _as_lowres(b)
        return K.argmax_pair_hist(la, lb, C, *size)[0].double()
    # label maps
    a, b = a.reshape(-1).long(), b.reshape(-1).long()
    n = num_classes or int(max(a.max().item(), b.max().item())) + 1
    return torch.bincount(a * n + b, minlength=n * n).view(n, n).double()


def _entropy(p):
    p = p.reshape(-1)
    p = p / p.sum()
    nz = p[p > 0]
    return float(-(nz * nz.log()).sum().item())


def mse_from_table(t):
    n = t.shape[0]
    idx = torch.arange(n, dtype=torch.float64, device=t.device)
    d2 = (idx[:, None] - idx[None, :]) ** 2
    return float(((t * d2).sum() / t.sum()).item())


def nmi_from_table(t):
    h01 = _entropy(t)
    if h01 == 0.0:
        return float("nan")
    return (_entropy(t.sum(dim=1)) + _entropy(t.sum(dim=0))) / h01


def vi_from_table(t, ignore=()):
    """[H(B|A), H(A|B)] in bits; pixels whose label in `a` is ignored get weight 0 (scikit-image semantics)."""
    t = t.clone()
    for lab in ignore:
        if 0 <= lab < t.shape[0]:
            t[lab] = 0
    tot = t.sum()
    if tot <= 0:
        return 0.0, 0.0
    pxy = t / tot
    px, py = pxy.sum(dim=1), pxy.sum(dim=0)

    def xlogx(x):
        return torch.where(x > 0, x * torch.log2(torch.where(x > 0, x, torch.ones_like(x))), torch.zeros_like(x))

    inv = lambda v: torch.where(v > 0, 1.0 / torch.where(v > 0, v, torch.ones_like(v)), torch.zeros_like(v))
    hygx = -(px * xlogx(pxy * inv(px)[:, None]).sum(dim=1)).sum()
    hxgy = -(xlogx(pxy * inv(py)[None, :]).sum(dim=0) * py).sum()
    return float(hygx.item()), float(hxgy.item())


def label_map(t, exit_index=None):
    """[N,H,W] int64 argmax label maps of an ExitLogits exit / a [N,C,H,W] score tensor / label maps."""
    if isinstance(t, ExitLogits):
        return K.argmax_confusion(t.lowres[exit_index].detach().contiguous(), t.num_classes, None, *t.size,
                                  want_pred=True)[1]
    if t.dim() == 4:
        lr, C, size = _as_lowres_batch(t)
        return K.argmax_confusion(lr, C, None, *size, want_pred=True)[1]
    t = t.squeeze()
    return (t if t.dim() == 3 else t.unsqueeze(0)).long().contiguous()


def _as_lowres_batch(t):
    N, C, H, W = t.shape
    lr = torch.zeros((N, H, W, E.CPAD), dtype=torch.float32, device=t.device)
    lr[..., :C] = t.detach().permute(0, 2, 3, 1)
    return lr, C, (H, W)


class SSIM:
    """sim_metrics.py:15-37: skimage's structural_similarity (defaults) of the two argmax label maps, with the
    caller's `data_range`.  Returns a python float for one image (as the reference), a list for a batch."""

    def __init__(self, data_range):
        self.dr = data_range

    def device_value(self, a, b, exit_a=None, exit_b=None):
        return K.ssim_labels(label_map(a, exit_a), label_map(b if b is not None else a, exit_b), self.dr)

    def __call__(self, a, b, exit_a=None, exit_b=None):
        v = self.device_value(a, b, exit_a, exit_b).cpu()
        return float(v[0]) if v.numel() == 1 else v.tolist()


def MSE(a, b, **kw):
    return mse_from_table(pair_table(a, b, **kw))


def NMI(a, b, **kw):
    return nmi_from_table(pair_table(a, b, **kw))


class VI:
    def __init__(self, ignore=()):
        self.ignore = tuple(ignore)

    def __call__(self, a, b, **kw):
        return math.fsum(vi_from_table(pair_table(a, b, **kw), self.ignore))


class Seg_comp(VI):
    def __init__(self, x_y=True, ignore=()):
        super().__init__(ignore=ignore)
        self.x_y = x_y

    def __call__(self, a, b, **kw):
        return vi_from_table(pair_table(a, b, **kw), self.ignore)[int(self.x_y)]
