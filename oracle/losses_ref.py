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


# ---------------------------------------------------------------------------------------------------------
# Region / focal losses (branchy_seg_losses.py:9-131), restated with plain torch ops so autograd supplies the
# gradients; pinned by tests/golden/region_losses.npz (values + gradients from the reference classes).
# ---------------------------------------------------------------------------------------------------------
def _br_reduce(per_exit, reduction, weights):
    """BrSegLoss.forward (:24-38): stack the exits, reduce all other dims, dot with the exit weights."""
    import torch
    losses = torch.cat([l.unsqueeze(0) for l in per_exit])
    dim = list(range(1, losses.dim()))
    if reduction == "mean":
        losses = losses.mean(dim=dim)
    elif reduction == "sum":
        losses = losses.sum(dim=dim)
    else:
        return losses
    w = torch.ones(len(per_exit)) if weights is None else torch.as_tensor(weights, dtype=torch.float32)
    return torch.dot(w, losses)


def _probs_onehot(y, t, drop_void=False):
    import torch
    import torch.nn.functional as F
    N, C = y.shape[:2]
    probs = F.softmax(y, 1).view(N, C, -1)
    t = t.view(N, -1).to(torch.int64)
    if drop_void:                        # JaccardLoss (:58-66): one-hot wide enough for the void index, then cut
        width = max(int(t.max()) + 1, C)
        oh = F.one_hot(t, num_classes=width).transpose(1, 2)[:, :C, :]
    else:
        oh = F.one_hot(t, num_classes=C).transpose(1, 2)
    return probs, oh


def br_dice(y_pred, targets, n_exits, smooth=1e-6, reduction="mean", weights=None):
    """DiceLoss (:40-48): per image 1 - (2 sum(p*t) + s) / (sum(p + t) + s)."""
    per = []
    for i in range(n_exits):
        p, oh = _probs_onehot(y_pred[i], targets)
        per.append(1 - (2 * (p * oh).sum(dim=(1, 2)) + smooth) / ((p + oh).sum(dim=(1, 2)) + smooth))
    return _br_reduce(per, reduction, weights)


def br_jaccard(y_pred, targets, n_exits, smooth=1e-6, reduction="mean", downgrad_bg=1.0):
    """JaccardLoss (:50-78): per image and class 1 - (I + s)/(U + s); class 0 scaled by downgrad_bg; with
    downgrad_bg == 0 the classes are summed instead."""
    import torch
    downgrad_bg = downgrad_bg if 0 <= downgrad_bg <= 1.0 else 1.0
    per = []
    for i in range(n_exits):
        p, oh = _probs_onehot(y_pred[i], targets, drop_void=True)
        inter = (p * oh).sum(dim=-1)
        union = (p + oh).sum(dim=-1) - inter
        iou = (inter + smooth) / (union + smooth)
        if downgrad_bg:
            loss = 1 - iou
            scale = torch.ones_like(loss)
            scale[:, 0] = downgrad_bg
            per.append(loss * scale)
        else:
            per.append((1 - iou).sum(dim=-1))
    return _br_reduce(per, reduction, None)


def br_tversky(y_pred, targets, n_exits, smooth=1e-6, alpha=.5, beta=.5, gamma=None, reduction="mean", weights=None):
    """TverskyLoss / FocalTverskyLoss (:80-111): TP/FP/FN of the ARGMAX one-hot (no gradient to the scores)."""
    import torch
    import torch.nn.functional as F
    per = []
    for i in range(n_exits):
        y = y_pred[i]
        N, C = y.shape[:2]
        pred = F.one_hot(torch.argmax(F.softmax(y, 1).view(N, C, -1), dim=1), num_classes=C).transpose(1, 2)
        oh = F.one_hot(targets.view(N, -1).to(torch.int64), num_classes=C).transpose(1, 2)
        TP = (pred * oh).sum(dim=-1)
        FP = (pred * (1 - oh)).sum(dim=-1)
        FN = ((1 - pred) * oh).sum(dim=-1)
        l = 1 - (TP + smooth) / (TP + alpha * FP + beta * FN + smooth)
        per.append(l if gamma is None else l ** gamma)
    return _br_reduce(per, reduction, weights)


def br_focal(y_pred, targets, n_exits, alpha=None, gamma=2, reduction="mean", weights=None, faithful_alpha=True):
    """FocalLoss (:113-131): per pixel -(1 - p_t)^gamma log p_t (* alpha_t).

    Reference quirk (:126-129): `alpha[targets]` keeps the targets' channel axis ([B,1,H,W]) while the loss map is
    [B,H,W], so their product BROADCASTS to [B,B,H,W] - every image's loss is multiplied by every image's alpha map.
    `faithful_alpha=True` reproduces that (it is what the golden vectors contain); False weights each pixel by the
    alpha of its own label, which is the same thing for batch size 1 and is what the HIP path implements."""
    import torch
    import torch.nn.functional as F
    per = []
    for i in range(n_exits):
        logp = F.log_softmax(y_pred[i], dim=1)
        t = targets.to(torch.int64)
        lp = logp.gather(1, t).squeeze(1)
        l = -((1 - torch.exp(lp)) ** gamma) * lp
        if alpha is not None:
            a = torch.as_tensor(alpha)
            l = l * (a[t] if faithful_alpha else a[t.squeeze(1)])
        per.append(l)
    return _br_reduce(per, reduction, weights)
