"""Oracle: multi-exit cross-entropy and raw-logit Lovasz losses (CPU, torch fp32).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Pinned by
``tests/golden/losses_*.npz`` (generated from the importable reference files).
"""
import torch
from torch.nn import functional as F


def br_xentropy(y_pred, targets, ignore_index=-100, b_reduction="mean", n_exits=0, weights=None):
    """my_pixelwise_xentropy.py:30-46 (+ _compute_loss :11-14).

    y_pred [E,B,C,H,W]; targets [B,1,H,W] or [B,H,W] int64.  Per exit:
    CrossEntropyLoss(mean over non-ignored pixels); then sum/mean/none over
    exits.  B-7 (``targets.squeeze()`` dropping a batch dim of 1) is fixed by
    squeezing dim 1 only.
    """
    if targets.dim() > 3:
        targets = targets.squeeze(1)
    if not n_exits:
        return F.cross_entropy(y_pred, targets, ignore_index=ignore_index)
    assert n_exits <= y_pred.shape[0]
    losses = torch.stack([F.cross_entropy(y_pred[i], targets, ignore_index=ignore_index)
                          for i in range(n_exits)])
    if weights is not None:
        losses = losses * torch.as_tensor(weights, dtype=losses.dtype)
    if b_reduction == "sum":
        return losses.sum()
    if b_reduction == "mean":
        return losses.mean()
    return losses


def lovasz_grad(gt_sorted):
    """lovaszsoftmax.py:19-31."""
    p = gt_sorted.numel()
    gts = gt_sorted.sum()
    inter = gts - gt_sorted.cumsum(0)
    union = gts + (1.0 - gt_sorted).cumsum(0)
    jac = 1.0 - inter / union
    if p > 1:
        jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
    return jac


def lovasz_softmax_flat(probas, labels, classes="present"):
    """lovaszsoftmax.py:172-200 on [P,C] scores / [P] labels."""
    if probas.numel() == 0:
        return probas.sum() * 0.0
    C = probas.shape[1]
    losses = []
    cls = list(range(C)) if classes in ("all", "present") else classes
    for c in cls:
        fg = (labels == c).float()
        if classes == "present" and fg.sum() == 0:
            continue
        err = (fg - probas[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        losses.append(torch.dot(err_sorted, lovasz_grad(fg[perm])))
    if not losses:
        return probas.sum() * 0.0
    return torch.stack(losses).mean()


def lovasz_softmax(scores, labels, classes="present", per_image=False, ignore=None):
    """lovaszsoftmax.py:154-169 + flatten_probas :203-219.  ``scores`` is what
    the caller passes - the reference passes RAW LOGITS (F6 / B-4)."""
    def flat(s, l):
        C = s.shape[1]
        s = s.permute(0, 2, 3, 1).reshape(-1, C)
        l = l.reshape(-1)
        if ignore is None:
            return s, l
        valid = l != ignore
        return s[valid], l[valid]

    if per_image:
        vals = [lovasz_softmax_flat(*flat(s.unsqueeze(0), l.unsqueeze(0)), classes=classes)
                for s, l in zip(scores, labels)]
        return torch.stack(vals).mean()
    return lovasz_softmax_flat(*flat(scores, labels), classes=classes)


def br_lovasz(y_pred, targets, classes="present", per_image=False, ignore=None, n_branches=0,
              prev_out=False):
    """branchy_seg_losses.py:133-159: sum over exits of lovasz on raw logits;
    with ``prev_out`` a linspace(0,1,n+1)[1:] weighted sum."""
    n = n_branches + 1
    if targets.dim() > 3:
        targets = targets.squeeze(1)
    losses = torch.stack([lovasz_softmax(y_pred[i], targets, classes, per_image, ignore)
                          for i in range(n)])
    if prev_out:
        w = torch.linspace(0, 1, n + 1)[1:]
        return torch.dot(w, losses)
    return losses.sum()
