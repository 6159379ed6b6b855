"""Similarity-gated early-exit evaluation (eval_br_sim.py:16-67): an image leaves at branch i when the label
maps of exits i-1 and i are similar enough (MSE / VI / H(X|Y) / H(Y|X) below tau, NMI above tau).  The gate
value comes from the per-image contingency table built on the device by ``eeseg_argmax_pair_hist`` for the
whole batch at once; the reference moves both full-resolution score tensors to the CPU per image and branch.
Result keys as the reference: ``b{i}_mIoU``, ``b{i}_count``, ``mIoU_out``, ``count_out``, ``mIoU_gl``,
``out_gl``, ``t``.  ``metric='ssim'`` (eval_br_sim.py:20-21: ``M.SSIM(n_classes - 1)``) compares the two label maps
with the windowed ``eeseg_ssim_labels`` kernel instead of the table."""
import math

import torch

from . import kernels as K
from . import sim_metrics as M
from .compute_mIoU import mIoU
from .eval_br_ent import _one_image
from .eval_mIoU import _forward_fused
from .from_deepv3_new import ExitLogits


def gate_function(metric, ignore=()):
    """-> (f(table) -> float, larger_is_similar) for a metric name of eval_br_sim.py:20-31."""
    m = metric.lower()
    if m == "ssim":
        raise ValueError("SSIM is not a function of the contingency table: br_evaluator handles it on the label maps")
    if m == "nmi":
        return M.nmi_from_table, True
    if m == "vi":
        return (lambda t: math.fsum(M.vi_from_table(t, ignore))), False
    if m == "h_xy":
        return (lambda t: M.vi_from_table(t, ignore)[1]), False
    if m == "h_yx":
        return (lambda t: M.vi_from_table(t, ignore)[0]), False
    return M.mse_from_table, False


def br_evaluator(net, n_exits, n_classes, test_loader, device, metric, tau, ignore=(), skip=0):
    accumulator = [mIoU(n_classes=n_classes, device=device) for _ in range(n_exits + 1)]
    out_count = [0 for _ in range(n_exits + 1)]
    ssim = M.SSIM(n_classes - 1) if metric.lower() == "ssim" else None          # eval_br_sim.py:20-21
    f, larger = (None, True) if ssim is not None else gate_function(metric, ignore)
    n_branches = n_exits - 1
    with torch.no_grad():
        for X, y in test_loader:
            X, y = X.to(device, non_blocking=True), y.to(device, non_blocking=True)
            y_pred = _forward_fused(net, X)
            fused = isinstance(y_pred, ExitLogits)
            B = X.shape[0]
            # contingency tables of every gated pair for the whole batch: one launch per pair, one D2H in all
            pairs = list(range(1 + skip, n_branches))
            tables = {}
            for i in pairs:
                if ssim is not None:      # [B] SSIM values of the pair's label maps, computed on the device
                    tables[i] = ssim.device_value(y_pred if fused else y_pred[i - 1], y_pred if fused else y_pred[i],
                                                  i - 1 if fused else None, i if fused else None)
                elif fused:
                    tables[i] = K.argmax_pair_hist(y_pred.lowres[i - 1].contiguous(), y_pred.lowres[i].contiguous(),
                                                   n_classes, *y_pred.size)
                else:
                    tables[i] = torch.stack([M.pair_table(y_pred[i - 1][b:b + 1], y_pred[i][b:b + 1]) for b in range(B)])
            tables = {i: t.double().cpu() for i, t in tables.items()}
            for b in range(B):
                left = False
                yb = y[b:b + 1]
                for i in pairs:
                    t = float(tables[i][b]) if ssim is not None else f(tables[i][b])
                    if (t > tau) if larger else (t < tau):
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
    return res
