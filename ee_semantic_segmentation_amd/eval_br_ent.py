"""Entropy-gated early-exit evaluation (eval_br_ent.py:19-84) with the gate metric
computed by the fused on-device kernel (upsample + softmax + normalised entropy +
optional s x s block pooling + mean): no D2H copy of the probabilities, no scipy."""
import torch

from . import engine as E
from . import kernels as K
from .compute_mIoU import mIoU
from .eval_mIoU import _forward_fused
from .from_deepv3_new import ExitLogits


class img_norm_entropy:
    """eval_br_ent.py:19-36.  The reference takes softmax probabilities [C,H,W] on the
    CPU; here ``__call__`` takes LOGITS: an ExitLogits + exit index (fused path) or a
    [C,H,W] / [B,C,H,W] logits tensor, and returns the gate value(s) as a device tensor."""

    def __init__(self, n_classes, pool_min=False, s=1):
        self.pool = s != 1
        self.pool_min = pool_min
        self.size = (s, s)
        self.C = n_classes

    def device_value(self, logits, exit_index=None, tau=0.0):
        mode = (2 if self.pool_min else 1) if self.pool else 0
        if isinstance(logits, ExitLogits):
            lr, (H, W) = logits.lowres[exit_index].detach().contiguous(), logits.size
        else:
            if logits.dim() == 3:
                logits = logits.unsqueeze(0)
            B, C, H, W = logits.shape
            lr = torch.zeros((B, H, W, E.CPAD), dtype=torch.float32, device=logits.device)
            lr[..., :C] = logits.detach().permute(0, 2, 3, 1)
        return K.entropy_gate(lr, self.C, H, W, tau, mode, self.size[0])

    def __call__(self, logits, exit_index=None):
        ent, _ = self.device_value(logits, exit_index)
        return ent if ent.numel() > 1 else ent[0]


def br_evaluator(net, n_exits, n_classes, test_loader, device, tau, metric="ent", size=1, ignore=(), skip=0):
    accumulator = [mIoU(n_classes=n_classes, device=device) for _ in range(n_exits + 1)]
    out_count = [0 for _ in range(n_exits + 1)]
    if metric.lower() == "max":
        l = img_norm_entropy(n_classes, s=size)
    elif metric.lower() == "min":
        l = img_norm_entropy(n_classes, s=size, pool_min=True)
    else:
        l = img_norm_entropy(n_classes)
    n_branches = n_exits - 1
    with torch.no_grad():
        for X, y in test_loader:
            X, y = X.to(device, non_blocking=True), y.to(device, non_blocking=True)
            y_pred = _forward_fused(net, X)
            fused = isinstance(y_pred, ExitLogits)
            # all gate decisions of the batch are taken on the device; one small D2H at the end
            flags = [l.device_value(y_pred if fused else y_pred[i], i, tau)[1] for i in range(skip, n_branches)]
            flags = torch.stack(flags, 0).cpu() if flags else None      # [branches, B]
            B = X.shape[0]
            for b in range(B):
                left = False
                yb = y[b:b + 1]
                for j, i in enumerate(range(skip, n_branches)):
                    if flags[j, b]:
                        pb = _one_image(y_pred, i, b)
                        accumulator[i](pb, yb, 0)
                        accumulator[-1](pb, yb, 0)
                        out_count[i] += 1
                        left = True
                        break
                if not left:
                    pb = _one_image(y_pred, len(y_pred) - 1 if fused else -1, b)
                    accumulator[-2](pb, yb, 0)
                    accumulator[-1](pb, yb, 0)
                    out_count[-2] += 1
                out_count[-1] += 1
    res = dict()
    for i in range(n_branches):
        res[f"b{i + 1}_mIoU"] = accumulator[i].compute().item()
        res[f"b{i + 1}_count"] = out_count[i]
    res["mIoU_out"] = accumulator[-2].compute().item()
    res["count_out"] = out_count[-2]
    res["mIoU_gl"] = accumulator[-1].compute().item()
    res["out_gl"] = out_count[-1]
    res["t"] = tau
    res["pool"] = metric
    res["pool_size"] = size
    return res


def _one_image(y_pred, exit_index, b):
    if isinstance(y_pred, ExitLogits):
        return ExitLogits([y_pred.lowres[exit_index][b:b + 1]], y_pred.num_classes, y_pred.size)
    return y_pred[exit_index][b:b + 1]
