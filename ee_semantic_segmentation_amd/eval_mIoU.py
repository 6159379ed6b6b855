"""``mIoU_evaluator`` (eval_mIoU.py:15-40): per-exit mIoU of a network over a loader.

Under data parallelism every rank scores its shard of the loader and the ``[E,3,C]`` (TP, FP, FN) counters
are summed over the ranks ONCE per evaluation, so every rank reports the mIoU of the whole set."""
import torch

from .compute_mIoU import mIoU
from .from_deepv3_new import ExitLogits


def _forward_fused(net, X):
    """Run the network and return its exits without materialising the stack."""
    if hasattr(net, "forward_lowres"):
        return ExitLogits(net.forward_lowres(X), net.num_classes, X.shape[-2:])
    return net(X)


def reduce_counters(net, accumulator):
    """Sum the per-exit accumulators over the data-parallel ranks (one collective; fp64 on the wire so integer counts
    stay exact far beyond fp32's 2^24).  No-op without a data-parallel transport."""
    cfg = getattr(net, "cfg", None)
    if cfg is None or not cfg.dp_active() or cfg.dp_world() == 1:
        return
    dev = next(net.parameters()).device
    stacked = torch.stack([a.accumulator.to(dev) for a in accumulator]).to(torch.float64).contiguous()
    cfg.all_reduce(stacked)
    for a, t in zip(accumulator, stacked):
        a.accumulator = t.to(torch.float32)


def mIoU_evaluator(net, n_exits, n_classes, test_loader, device, nan_safe=False, reduce=True):
    accumulator = [mIoU(n_classes=n_classes, device=device, nan_safe=nan_safe) for _ in range(n_exits)]
    n_branches = n_exits - 1
    with torch.no_grad():
        for X, y in test_loader:
            X, y = X.to(device, non_blocking=True), y.to(device, non_blocking=True)
            y_pred = _forward_fused(net, X)
            fused = isinstance(y_pred, ExitLogits)
            for i in range(n_branches):
                accumulator[i](y_pred if fused else y_pred[i], y, i)
            last = len(y_pred) - 1 if fused else -1
            accumulator[-1](y_pred if fused else (y_pred[-1] if n_branches else y_pred), y, last)
    if reduce:
        reduce_counters(net, accumulator)
    res = dict()
    for i in range(n_branches):
        res[f"b{i + 1}_mIoU"] = accumulator[i].compute().item()
    res["mIoU"] = accumulator[-1].compute().item()
    return res
