"""``BrXEntropyLoss`` - the reference's multi-exit cross entropy
(my_pixelwise_xentropy.py:19-46) on the fused HIP upsample+CE kernels.

Accepts the reference's ``y_pred [E,B,C,H,W]`` tensor or this package's
``ExitLogits`` (low-resolution logits; the bilinear upsample is fused into the
loss kernel so the stacked tensor is never materialised).  Fixes B-6
(``update_n``) and B-7 (squeeze only the channel dim of the targets).
"""
import torch
from torch import nn

from . import engine as E
from . import kernels as K
from .from_deepv3_new import ExitLogits


CHECK_LABELS = __import__("os").environ.get("EESEG_CHECK_LABELS") == "1"


def _as_lowres(y, C):
    """[B,C,H,W] full-resolution logits -> NHWC [B,H,W,32] fp32 (identity upsample)."""
    B, Cc, H, W = y.shape
    lr = torch.zeros((B, H, W, E.CPAD), dtype=torch.float32, device=y.device)
    lr[..., :Cc] = y.permute(0, 2, 3, 1)
    return lr


class _FusedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, C, H, W, ignore_index, mean, comm, *lrs):
        dev = lrs[0].device
        accum = torch.zeros((len(lrs), 2), dtype=torch.float64, device=dev)
        lrs = [lr.contiguous() for lr in lrs]
        for e, lr in enumerate(lrs):
            K.upsample_ce_fwd(lr, C, target, H, W, ignore_index, accum[e])
        if mean:
            if comm is not None and comm.dp_active():
                # global valid-pixel count so that the DP average of the per-rank losses is
                # the single-device loss of the whole batch; same group as the
                # gradient average and SyncBN (engine.Config.group)
                accum[:, 1] = E.global_mean_normaliser(comm, accum[:, 1])
        else:
            accum[:, 1] = 1.0
        ctx.save_for_backward(target, accum, *lrs)
        ctx.meta = (C, H, W, ignore_index)
        return (accum[:, 0] / accum[:, 1]).float()

    @staticmethod
    def backward(ctx, dl):
        target, accum, *lrs = ctx.saved_tensors
        C, H, W, ignore_index = ctx.meta
        dl = dl.contiguous().float()
        grads = []
        for e, lr in enumerate(lrs):
            dlr = torch.zeros_like(lr)
            K.upsample_ce_bwd(lr, C, target, H, W, ignore_index, accum[e], 1.0, dlr, gscale_dev=dl[e:e + 1])
            grads.append(dlr)
        return (None, None, None, None, None, None, None, *grads)


def _default_comm(group=None):
    c = E.Config()
    c.group = group
    return c


def fused_cross_entropy(lowres, target, num_classes, size, ignore_index=-100, reduction="mean", comm=None):
    """Per-exit CrossEntropyLoss(reduction, ignore_index) of bilinearly upsampled
    low-res logits.  Returns a [E] fp32 tensor.  `comm`: the engine.Config whose process group the
    valid-pixel count is summed over (None = the default group)."""
    if comm is None:
        comm = _default_comm()
    if reduction not in ("mean", "sum"):
        raise ValueError("fused cross entropy supports reduction 'mean' or 'sum'")
    if target.dim() > 3:
        target = target.squeeze(1)                       # B-7: only the channel dim
    target = target.contiguous()
    if target.dtype != torch.int64:
        target = target.long()
    if CHECK_LABELS:
        # torch's CrossEntropyLoss asserts on a label outside [0, C) that is not ignore_index; the fused kernel skips
        # such pixels.  The check costs a device->host sync, so it is a debugging switch (EESEG_CHECK_LABELS=1): it
        # catches e.g. VOC's 255 border label meeting the default ignore_index = -100.
        bad = ((target < 0) | (target >= num_classes)) & (target != ignore_index)
        if bool(bad.any()):
            raise ValueError(f"{int(bad.sum())} target values outside [0, {num_classes}) that are not ignore_index="
                             f"{ignore_index} (first: {int(target[bad][0])})")
    H, W = size
    return _FusedCE.apply(target, num_classes, H, W, int(ignore_index), reduction == "mean", comm, *lowres)


class _cross_entropy(nn.Module):
    def __init__(self, reduction="mean", ignore_index=-100, group=None):
        super().__init__()
        self.reduction, self.ignore_index = reduction, ignore_index
        self.group = group           # process group of the valid-pixel count when y_pred is a plain tensor

    def _exit_losses(self, y_pred, targets, n):
        if isinstance(y_pred, ExitLogits):
            comm = y_pred.cfg if y_pred.cfg is not None else _default_comm(self.group)
            return fused_cross_entropy(y_pred.lowres[:n], targets, y_pred.num_classes, y_pred.size,
                                       self.ignore_index, self.reduction, comm)
        ys = [y_pred] if y_pred.dim() == 4 else [y_pred[i] for i in range(n)]
        C = ys[0].shape[1]
        return fused_cross_entropy([_as_lowres(y, C) for y in ys], targets, C, ys[0].shape[-2:], self.ignore_index,
                                   self.reduction, _default_comm(self.group))

    def _compute_loss(self, y_pred, targets):
        return self._exit_losses(y_pred, targets, 1)[0]

    def forward(self, y_pred, targets):
        return self._compute_loss(y_pred, targets)


class BrXEntropyLoss(_cross_entropy):
    def __init__(self, reduction="mean", ignore_index=-100, b_reduction="mean", n_exits=0, weights=None, group=None):
        super().__init__(reduction, ignore_index, group)
        self.b_reduction = b_reduction
        self.n_exits = n_exits
        if weights and len(weights) == n_exits:
            self.weights = torch.tensor(weights, dtype=torch.float32)
        else:
            self.weights = weights

    def update_n(self, n):                                # B-6 (deepv3_funcs.py:231 calls it)
        self.n_exits = n + 1

    def forward(self, y_pred, targets):
        if not self.n_exits:
            return self._compute_loss(y_pred, targets)
        n_have = len(y_pred) if isinstance(y_pred, ExitLogits) else y_pred.shape[0]
        assert self.n_exits <= n_have
        losses = self._exit_losses(y_pred, targets, self.n_exits)
        if self.weights is not None:
            losses = losses * self.weights.to(losses.device)
        if self.b_reduction == "sum":
            return losses.sum()
        if self.b_reduction == "mean":
            return losses.mean()
        return losses
