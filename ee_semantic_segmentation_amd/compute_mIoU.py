"""``mIoU`` - the reference's streaming TP/FP/FN accumulator (compute_mIoU.py:7-36 +
seg_metrics.py:13-28) on the fused HIP upsample+argmax+confusion kernel.

Quirks reproduced: void pixels (label >= C) count as FP of the predicted class
(B-9); a class absent from prediction and target yields NaN (B-8) unless
``nan_safe=True``; the accumulator is fp32 and is updated with per-call exact
integer counts exactly like the reference's ``accumulator[k] += TP.sum(0)``.
"""
import torch

from . import engine as E
from . import kernels as K
from .from_deepv3_new import ExitLogits


def confusion_counts(y_pred, targets, exit_index=None):
    """int32 [3,C] (TP, FP, FN) of one exit.  y_pred: [B,C,H,W] logits tensor, or an
    ExitLogits together with exit_index."""
    if targets.dim() > 3:
        targets = targets.squeeze(1)
    targets = targets.contiguous().long()
    if isinstance(y_pred, ExitLogits):
        lr, C, (H, W) = y_pred.lowres[exit_index].detach().contiguous(), y_pred.num_classes, y_pred.size
    else:
        B, C, H, W = y_pred.shape
        lr = torch.zeros((B, H, W, E.CPAD), dtype=torch.float32, device=y_pred.device)
        lr[..., :C] = y_pred.detach().permute(0, 2, 3, 1)
    counts, _ = K.argmax_confusion(lr, C, targets, H, W)
    return counts


class mIoU:
    def __init__(self, n_classes, device="cpu", nan_safe=False):
        self.C = n_classes
        self.accumulator = torch.zeros((3, n_classes))
        self.nan_safe = nan_safe

    def __call__(self, y_pred, targets, exit_index=None):
        return self.forward(y_pred, targets, exit_index)

    def forward(self, y_pred, targets, exit_index=None):
        counts = confusion_counts(y_pred, targets, exit_index)
        assert counts.shape[1] == self.accumulator.shape[1]
        if counts.device != self.accumulator.device:
            self.accumulator = self.accumulator.to(counts.device)
        self.accumulator += counts.to(torch.float32)

    def compute(self):
        den = self.accumulator.sum(dim=0)
        cIoU = torch.div(self.accumulator[0], den)
        if self.nan_safe:
            cIoU = torch.nan_to_num(cIoU, nan=1.0)
        return (cIoU.sum() / self.C).cpu()


class img_mIoU:
    """Per-image mean IoU over the classes present in the target (compute_mIoU.py:38-63), one image per call, on
    the fused upsample+argmax+confusion kernel: IoU_i = TP_i / (TP_i + FP_i + FN_i) for every class with
    TP_i + FN_i > 0; when the image has void pixels the void label is one more "present class" with IoU 0
    (it is in `unique(target)` and never predicted), exactly as the reference counts it."""

    def __init__(self):
        self.accumulator = [0.0, 0]

    def __call__(self, y_pred, target, exit_index=None):
        return self.forward(y_pred, target, exit_index)

    def forward(self, y_pred, target, exit_index=None):
        if not isinstance(y_pred, ExitLogits) and y_pred.dim() == 3:
            y_pred = y_pred.unsqueeze(0)
        t = target.reshape(1, *target.shape[-2:])
        counts = confusion_counts(y_pred, t, exit_index).to(torch.float32)
        tp, fp, fn = counts[0], counts[1], counts[2]
        present = (tp + fn) > 0
        iou = torch.where(present, tp / (tp + fp + fn), torch.zeros_like(tp))
        n_void = t.numel() - int((tp + fn).sum().item())
        n_classes = int(present.sum().item()) + (1 if n_void > 0 else 0)
        self.accumulator[0] += (iou.sum() / n_classes).item()
        self.accumulator[1] += 1

    def compute(self):
        if self.accumulator[1] <= 0:
            return float("nan")
        return self.accumulator[0] / self.accumulator[1]
