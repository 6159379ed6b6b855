"""The reference's multi-exit segmentation losses (branchy_seg_losses.py) on HIP kernels.

``LovaszSoftmax`` (:133-159) runs on the Lovasz kernels (sort + scan on device).  ``DiceLoss`` (:40-48),
``JaccardLoss`` (:50-78), ``TverskyLoss`` / ``FocalTverskyLoss`` (:80-111) and ``FocalLoss`` (:113-131) share one
fused pass per exit (``eeseg_class_sums_fwd/bwd``): softmax of the bilinearly upsampled logits is reduced on the
fly to per-image, per-class sums (and the focal sum), the closed forms on those few numbers are ordinary torch ops
(autograd differentiates them), and the backward kernel turns dL/dS, dL/dI, dL/dF into low-res logit gradients.

Notes on LovaszSoftmax:

Faithful quirks: the exits' RAW LOGITS are fed to the Lovasz extension (SURVEY F6 /
B-4: the reference never applies a softmax), per_image=False ranks all pixels of the
batch jointly, the per-exit losses are summed (or linspace-weighted with ``prev_out``).
classes = 'present' (what main_bradeepv3.py:121 uses), 'all' or a list, per_image False / True: all on
the GPU path (lovaszsoftmax.py:154-169,185-188).
"""
import torch
from torch import nn

from . import kernels as K
from .from_deepv3_new import ExitLogits


class _LovaszFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, target, ignore, classes):
        scores = scores.contiguous()
        need = scores.requires_grad or torch.is_grad_enabled()
        # gradient is produced in the same pass as the loss (it is the sorted Jaccard
        # increment scattered back), so compute it now with unit scale
        loss, ds = K.lovasz(scores.detach(), target, ignore, want_grad=need, classes=classes)
        ctx.save_for_backward(ds) if ds is not None else None
        ctx.has = ds is not None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        if not ctx.has:
            return None, None, None, None
        (ds,) = ctx.saved_tensors
        return ds * g, None, None, None


class _LovaszShareFn(torch.autograd.Function):
    """This rank's share of the whole batch's Lovasz loss under class-sharded data parallelism: sum over ITS classes of
    loss_c, divided by the number of classes in the mean (`norm`, a device int32).  `scores` holds one plane per owned
    class, `class_ids` says which label class each plane ranks, labels of every other class are valid background."""

    @staticmethod
    def forward(ctx, scores, target, ignore, class_ids, n_label, present, norm):
        scores = scores.contiguous()
        need = scores.requires_grad or torch.is_grad_enabled()
        loss, ds = K.lovasz(scores.detach(), target, ignore, want_grad=need, classes="present" if present else "all",
                            n_label_classes=n_label, class_ids=class_ids, norm_classes_dev=norm)
        ctx.save_for_backward(ds) if ds is not None else None
        ctx.has = ds is not None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        if not ctx.has:
            return (None,) * 7
        (ds,) = ctx.saved_tensors
        return (ds * g,) + (None,) * 6


class _GatherRowsSummedBack(torch.autograd.Function):
    """[n, ...] on every rank -> [world*n, ...] (rank-major).  Backward for losses of which every rank evaluates a SHARE on the
    whole batch (class-sharded Lovasz): the gradient of this rank's rows is the SUM over the ranks of their gradients of those
    rows (reduce-scatter), times world because the data-parallel reducer averages the parameter gradients over the ranks."""

    @staticmethod
    def forward(ctx, t, cfg):
        ctx.cfg, ctx.n = cfg, t.shape[0]
        return cfg.all_gather(t.contiguous()).flatten(0, 1)

    @staticmethod
    def backward(ctx, g):
        cfg, n = ctx.cfg, ctx.n
        world = cfg.dp_world()
        mine = cfg.reduce_scatter(g.contiguous().view(world, n, *g.shape[1:]))
        return mine * float(world), None


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
    """lovaszsoftmax.py:154-169.  classes: 'present', 'all' or a list of class indices (:185-188).  per_image=True
    (:165-166): every image is ranked alone and the per-image losses are averaged - one segmented sort per image (the
    sort segments are (image, class) pairs instead of classes)."""
    if labels.dim() > 3:
        labels = labels.squeeze(1)
    labels = labels.contiguous().long()
    probas = probas.float()
    if not isinstance(classes, str):
        classes = tuple(int(c) for c in classes)
    if not per_image:
        return _LovaszFn.apply(probas, labels, ignore, classes)
    per = [_LovaszFn.apply(probas[n:n + 1], labels[n:n + 1], ignore, classes) for n in range(probas.shape[0])]
    return torch.stack(per).sum() / len(per)            # lovaszsoftmax.py:233-251 `mean`


class LovaszSoftmax(nn.Module):
    """Data-parallel runs: ``per_image=False`` ranks all pixels of the batch JOINTLY (branchy_seg_losses.py:134,154;
    lovaszsoftmax.py:172-200), so the loss of a sharded batch is not a function of per-shard losses.  The default
    (`per_shard=False`) is exact AND non-redundant (round 4): the loss is a mean over classes of per-class losses that are
    independent given all pixels, so the ranks share the CLASSES - every rank all-gathers the exits' low-resolution logits
    and the labels (a few MB; 10x less than moving the full-resolution (error, flag) pairs of a key-range partition would),
    upsamples and ranks only classes r, r + world, ... over the whole batch (sort keys per rank = P_global * C / world =
    its own pixels' worth), the scalar shares are summed, and the low-resolution logit gradients go back by ONE
    reduce-scatter per exit (each rank receives the summed gradient of its own images).  `class_sharded=False` keeps the
    round-2/3 form (every rank ranks every class of the whole batch: world times the sort work).  `per_shard=True` ranks
    each rank's pixels alone (cheaper, a documented deviation).  `comm`: engine.Config whose group is used when the
    prediction is a plain tensor (an ExitLogits carries its network's)."""

    def __init__(self, classes="present", per_image=False, ignore=None, n_branches=0, prev_out=False, per_shard=False,
                 comm=None, class_sharded=True):
        super().__init__()
        self.classes, self.per_image, self.ignore = classes, per_image, ignore
        self.class_sharded = class_sharded
        self.last_sort_keys = None       # elements this rank ranked in the last forward call, per exit (tests / INTEGRATION)
        self.n = n_branches + 1
        self.prev_out = prev_out
        self.per_shard = per_shard
        self.comm = comm
        self._set_weights()

    def _set_weights(self):
        self.weights = torch.linspace(0, 1, self.n + 1)[1:] if self.prev_out else None

    def update_n(self, n):
        self.n = n + 1
        self._set_weights()

    def forward(self, y_pred, targets):
        losses = []
        comm = getattr(y_pred, "cfg", None) or self.comm
        # per_image=True ranks every image alone: the mean over a rank's images, averaged over equal shards by the
        # data-parallel reducer, IS the whole batch's loss - nothing to gather
        gather = comm is not None and comm.dp_active() and not self.per_shard and not self.per_image
        if gather:
            t = targets.squeeze(1) if targets.dim() > 3 else targets
            targets = comm.all_gather(t.contiguous()).flatten(0, 1)
        if gather and self.class_sharded and isinstance(y_pred, ExitLogits):
            return self._forward_class_sharded(y_pred, targets.contiguous().long(), comm)
        for i in range(self.n):
            if gather and isinstance(y_pred, ExitLogits):
                from .from_deepv3_new import upsample_logits
                lr = E.AllGatherRows.apply(y_pred.lowres[i], comm)
                yi = upsample_logits(lr, y_pred.num_classes, y_pred.size)
            elif gather:
                yi = E.AllGatherRows.apply(y_pred[i], comm)
            else:
                yi = y_pred[i]            # ExitLogits materialises one exit; a tensor indexes
            losses.append(lovasz_softmax(yi, targets, self.classes, self.per_image, self.ignore).unsqueeze(0))
        losses = torch.cat(losses)
        if self.prev_out:
            return torch.dot(self.weights.to(losses.device), losses).sum()
        return losses.sum()


def _forward_class_sharded(self, y_pred, targets, comm):
    """LovaszSoftmax.forward for a data-parallel run, exact and non-redundant (class docstring).  `targets`: the gathered
    labels of the whole batch [N_global, H, W]."""
    from .from_deepv3_new import upsample_logits
    world, rank = comm.dp_world(), comm.dp_rank()
    C_ = y_pred.num_classes
    dev = targets.device
    cand = list(range(C_)) if isinstance(self.classes, str) else [int(c) for c in self.classes]
    own = cand[rank::world]                        # round-robin over the classes of the mean
    present = self.classes == "present"
    if present:                                     # number of classes in the mean = classes present in the WHOLE batch
        norm = torch.count_nonzero(K.label_hist(targets, C_, self.ignore)).to(torch.int32).reshape(1)
    else:
        norm = torch.full((1,), len(cand), dtype=torch.int32, device=dev)
    idx = torch.tensor(own, dtype=torch.long, device=dev)
    shares, keys = [], []
    for i in range(self.n):
        lr = _GatherRowsSummedBack.apply(y_pred.lowres[i], comm)          # [N_global, h, w, CPAD], every class
        if own:
            planes = torch.zeros(lr.shape[:-1] + (lr.shape[-1],), dtype=lr.dtype, device=dev).index_copy(
                -1, torch.arange(len(own), device=dev), lr.index_select(-1, idx))      # owned classes in the leading planes
            yi = upsample_logits(planes, len(own), y_pred.size)          # [N_global, C_own, H, W]
            shares.append(_LovaszShareFn.apply(yi, targets, self.ignore, own, C_, present, norm).unsqueeze(0))
            keys.append(int(yi.shape[0] * yi.shape[1] * yi.shape[2] * yi.shape[3]))
        else:                                       # more ranks than classes: nothing to rank, but every rank joins the collectives
            shares.append((lr.sum() * 0.0).unsqueeze(0))
            keys.append(0)
    self.last_sort_keys = keys
    shares = torch.cat(shares)
    total = shares.detach().clone()
    comm.all_reduce(total)                          # the value every rank reports: the whole batch's loss of every exit
    losses = shares + (total - shares.detach()) / float(world) * 0.0 + (total - shares.detach())
    # (value = total; gradient = this rank's share's)  NB: the data-parallel reducer AVERAGES rank gradients and
    # _GatherRowsSummedBack multiplies by world, so the parameter gradients are those of `total`
    if self.prev_out:
        return torch.dot(self.weights.to(losses.device), losses).sum()
    return losses.sum()


LovaszSoftmax._forward_class_sharded = _forward_class_sharded


# --------------------------------------------------------------------------------------------------------------
# region / focal losses (branchy_seg_losses.py:9-131)
# --------------------------------------------------------------------------------------------------------------
from . import engine as E  # noqa: E402


class _ClassSums(torch.autograd.Function):
    """(S, I, T, void, F) of one exit from its low-res logits; differentiable in S, I, F."""

    @staticmethod
    def forward(ctx, lr, target, C, H, W, gamma, alpha, alpha_batch_sum=False):
        lr = lr.contiguous()
        sums, extra = K.class_sums_fwd(lr.detach(), C, target, H, W, gamma, alpha, alpha_batch_sum)
        ctx.save_for_backward(lr, target, alpha if alpha is not None else torch.empty(0, device=lr.device))
        ctx.meta = (C, H, W, gamma, alpha is not None, alpha_batch_sum)
        S, I, T = sums[:, 0, :C].float(), sums[:, 1, :C].float(), sums[:, 2, :C].float()
        ctx.mark_non_differentiable(T)
        void, F = extra[:, 0].float(), extra[:, 1].float()
        ctx.mark_non_differentiable(void)
        return S, I, T, void, F

    @staticmethod
    def backward(ctx, gS, gI, _gT, _gV, gF):
        lr, target, alpha = ctx.saved_tensors
        C, H, W, gamma, has_alpha, batch_sum = ctx.meta
        N = lr.shape[0]

        def pad(g):
            if g is None:
                return None
            out = torch.zeros((N, 32), dtype=torch.float32, device=lr.device)
            out[:, :C] = g
            return out

        gf = None
        if gF is not None and gamma >= 0:
            # the same dL/dF for every image only if it was reduced by a plain sum/mean: pass the per-image factor
            gf = gF.float().contiguous()
        dlr = torch.zeros_like(lr)
        if gf is None or N == 1 or bool((gf == gf[0]).all()):
            K.class_sums_bwd(lr, C, target, H, W, pad(gS), pad(gI), None if gf is None else gf[:1].contiguous(), dlr, gamma,
                             alpha if has_alpha else None, batch_sum)
        elif batch_sum:
            raise NotImplementedError("faithful FocalLoss alpha couples the images of a batch: reduce with 'mean' / 'sum'")
        else:                       # image-dependent focal weights: one launch per image
            K.class_sums_bwd(lr, C, target, H, W, pad(gS), pad(gI), None, dlr, gamma, alpha if has_alpha else None)
            for n in range(N):
                K.class_sums_bwd(lr[n:n + 1], C, target[n:n + 1], H, W, None, None, gf[n:n + 1].contiguous(), dlr[n:n + 1],
                                 gamma, alpha if has_alpha else None)
        return dlr, None, None, None, None, None, None, None


def _exit_lowres(y_pred, i):
    """-> (low-res logits [N,h,w,32] (autograd-connected), C, (H, W)) of exit i."""
    if isinstance(y_pred, ExitLogits):
        return y_pred.lowres[i], y_pred.num_classes, y_pred.size
    y = y_pred[i]
    N, C, H, W = y.shape
    lr = torch.zeros((N, H, W, E.CPAD), dtype=torch.float32, device=y.device)
    lr[..., :C] = y.permute(0, 2, 3, 1)
    return lr, C, (H, W)


def _targets(targets):
    if targets.dim() > 3:
        targets = targets.squeeze(1)
    return targets.contiguous().long()


class BrSegLoss(nn.Module):
    """branchy_seg_losses.py:9-38: per-exit `_compute_loss`, reduced over everything but the exit axis
    ('mean' / 'sum' / anything else = no reduction), then dotted with the per-exit weights."""

    def __init__(self, smooth=1e-6, reduction="mean", n_branches=0, weights=None):
        super().__init__()
        self.smooth, self.reduction = smooth, reduction
        self.n = n_branches + 1
        if weights and len(weights) == n_branches + 1:
            self.weights = torch.tensor(weights, dtype=torch.float32, requires_grad=True)
        else:
            self.weights = torch.ones(self.n, requires_grad=True)

    def update_n(self, n):
        self.n = n + 1

    def _sums(self, y_pred, i, targets, gamma=-1.0, alpha=None, allow_void=False, alpha_batch_sum=False):
        lr, C, (H, W) = _exit_lowres(y_pred, i)
        S, I, T, void, F = _ClassSums.apply(lr, targets, C, H, W, gamma, alpha, alpha_batch_sum)
        if not allow_void and float(void.sum()) > 0:
            # the reference one-hot encodes / gathers with num_classes = C and fails the same way on a void label
            raise RuntimeError("Class values must be smaller than num_classes.")
        return S, I, T, F, C, H * W

    def forward(self, y_pred, targets):
        targets = _targets(targets)
        losses = torch.cat([self._compute_loss(y_pred, i, targets).unsqueeze(0) for i in range(self.n)])
        dim = list(range(1, losses.dim()))
        if self.reduction == "mean":
            losses = losses.mean(dim=dim)
        elif self.reduction == "sum":
            losses = losses.sum(dim=dim)
        else:
            return losses
        return torch.dot(self.weights.to(device=losses.device), losses)


class DiceLoss(BrSegLoss):
    def _compute_loss(self, y_pred, i, targets):           # :40-48 -> [N]
        S, I, T, _, _, _ = self._sums(y_pred, i, targets)
        num = 2 * I.sum(dim=1) + self.smooth
        den = (S + T).sum(dim=1) + self.smooth
        return 1 - num / den


class JaccardLoss(BrSegLoss):
    def __init__(self, smooth=1e-6, reduction="mean", n_branches=0, downgrad_bg=1.):
        super().__init__(smooth, reduction, n_branches)
        self.downgrad_bg = downgrad_bg if 0 <= downgrad_bg <= 1. else 1.

    def _compute_loss(self, y_pred, i, targets):           # :55-78 -> [N,C] (or [N] when downgrad_bg == 0)
        S, I, T, _, _, _ = self._sums(y_pred, i, targets, allow_void=True)    # the void one-hot column is dropped
        union = (S + T) - I
        IoU = (I + self.smooth) / (union + self.smooth)
        if self.downgrad_bg:
            loss = 1 - IoU
            scale = torch.ones_like(loss)
            scale[:, 0] = self.downgrad_bg
            return loss * scale
        return (1 - IoU).sum(dim=-1)


class TverskyLoss(BrSegLoss):
    """:80-100.  Built on the ARGMAX of the probabilities, so it is piecewise constant: like the reference it
    yields no gradient for the network (only for the per-exit weights)."""

    def __init__(self, smooth=1e-6, alpha=.5, beta=.5, reduction="mean", n_branches=1, weights=None):
        super().__init__(smooth, reduction, n_branches, weights)
        self.alpha, self.beta = alpha, beta

    def _forward_imp(self, y_pred, i, targets):
        lr, C, (H, W) = _exit_lowres(y_pred, i)
        lr = lr.detach().contiguous()
        if int(((targets < 0) | (targets >= C)).sum()) > 0:
            raise RuntimeError("Class values must be smaller than num_classes.")
        rows = []
        for n in range(lr.shape[0]):                       # per-image TP / FP / FN on the fused argmax kernel
            cnt, _ = K.argmax_confusion(lr[n:n + 1], C, targets[n:n + 1], H, W)
            rows.append(cnt.float())
        cnt = torch.stack(rows)                            # [N,3,C]
        TP, FP, FN = cnt[:, 0], cnt[:, 1], cnt[:, 2]
        return 1 - (TP + self.smooth) / (TP + self.alpha * FP + self.beta * FN + self.smooth)

    def _compute_loss(self, y_pred, i, targets):
        return self._forward_imp(y_pred, i, targets)


class FocalTverskyLoss(TverskyLoss):
    def __init__(self, smooth=1e-6, alpha=.5, beta=.5, gamma=1., reduction="mean", n_branches=1, weights=None):
        super().__init__(smooth, alpha, beta, reduction, n_branches, weights)
        self.gamma = gamma

    def _compute_loss(self, y_pred, i, targets):
        return self._forward_imp(y_pred, i, targets) ** self.gamma


class _FocalMap(torch.autograd.Function):
    """The unreduced focal map of one exit from its low-res logits (eeseg_focal_map_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, lr, target, C, H, W, gamma, alpha, alpha_mode):
        lr = lr.contiguous()
        out, void = K.focal_map_fwd(lr.detach(), C, target, H, W, gamma, alpha, alpha_mode)
        if int(void.item()) > 0:
            raise RuntimeError("Class values must be smaller than num_classes.")      # as the reference's gather
        ctx.save_for_backward(lr, target, alpha if alpha is not None else torch.empty(0, device=lr.device))
        ctx.meta = (C, H, W, gamma, alpha_mode)
        return out

    @staticmethod
    def backward(ctx, dmap):
        lr, target, alpha = ctx.saved_tensors
        C, H, W, gamma, alpha_mode = ctx.meta
        dlr = torch.zeros_like(lr)
        K.focal_map_bwd(lr, C, target, H, W, gamma, dmap.float().contiguous(), dlr, alpha if alpha_mode else None, alpha_mode)
        return dlr, None, None, None, None, None, None, None


class FocalLoss(BrSegLoss):
    """:113-131.  The reference returns the per-pixel map [N,H,W] and reduces it afterwards; for reduction 'mean' / 'sum' the
    fused kernel reduces on the fly (the map is never materialised), any other reduction returns the stacked maps
    [E,N,H,W] (with alpha: [E,N,N,H,W] when faithful_alpha, as the reference's broadcast gives) from eeseg_focal_map_fwd.

    ``alpha``: the reference multiplies the [B,H,W] loss map by ``alpha[targets]`` of shape [B,1,H,W], which broadcasts
    to [B,B,H,W] - every image's loss is weighted by every image's alpha map (:126-129).  ``faithful_alpha=True``
    (default: what a drop-in must return, golden-pinned for every batch size) reproduces exactly that: summed over the
    extra axis it is a per-pixel weight sum_i alpha[t_i(h,w)], evaluated inside the fused kernels
    (eeseg_class_sums_* alpha_batch_sum).  ``faithful_alpha=False`` weights each pixel by the alpha of its own label -
    the evident intent; the two coincide for batch size 1."""

    def __init__(self, alpha=None, gamma=2, smooth=1e-6, reduction="mean", n_branches=1, weights=None,
                 faithful_alpha=True):
        super().__init__(smooth, reduction, n_branches, weights)
        self.alpha = None if alpha is None else torch.as_tensor(alpha, dtype=torch.float32)
        self.gamma = gamma
        self.faithful_alpha = faithful_alpha

    def _compute_loss(self, y_pred, i, targets):
        lr = _exit_lowres(y_pred, i)[0]
        alpha = None if self.alpha is None else self.alpha.to(lr.device).contiguous()
        if self.reduction not in ("mean", "sum"):               # the unreduced map (:132 `return loss`)
            lr, C, (H, W) = _exit_lowres(y_pred, i)
            mode = 0 if alpha is None else (2 if self.faithful_alpha else 1)
            return _FocalMap.apply(lr, targets, C, H, W, float(self.gamma), alpha, mode)
        batch_sum = alpha is not None and self.faithful_alpha and lr.shape[0] > 1
        _, _, _, F, _, hw = self._sums(y_pred, i, targets, gamma=float(self.gamma), alpha=alpha, alpha_batch_sum=batch_sum)
        # [N] per-image sums; 'mean' over [N,H,W] = sum / (N*H*W): scale so that BrSegLoss.forward's mean over dim 1 fits
        # (the faithful alpha form averages over [N,N,H,W]: one more factor N)
        if self.reduction == "mean":
            return F / (hw * lr.shape[0]) if batch_sum else F / hw
        return F
