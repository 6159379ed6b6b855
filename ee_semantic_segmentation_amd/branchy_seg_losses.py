"""``LovaszSoftmax`` - the reference's multi-exit Lovasz wrapper
(branchy_seg_losses.py:133-159) on the HIP Lovasz kernels (sort + scan on device).

Faithful quirks: the exits' RAW LOGITS are fed to the Lovasz extension (SURVEY F6 /
B-4: the reference never applies a softmax), per_image=False ranks all pixels of the
batch jointly, the per-exit losses are summed (or linspace-weighted with ``prev_out``).
Only classes='present' and per_image=False (what main_bradeepv3.py:121 uses) run on
the GPU path.
"""
import torch
from torch import nn

from . import kernels as K
from .from_deepv3_new import ExitLogits


class _LovaszFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, target, ignore):
        scores = scores.contiguous()
        need = scores.requires_grad or torch.is_grad_enabled()
        # gradient is produced in the same pass as the loss (it is the sorted Jaccard
        # increment scattered back), so compute it now with unit scale
        loss, ds = K.lovasz(scores.detach(), target, ignore, want_grad=need)
        ctx.save_for_backward(ds) if ds is not None else None
        ctx.has = ds is not None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        if not ctx.has:
            return None, None, None
        (ds,) = ctx.saved_tensors
        return ds * g, None, None


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
    """lovaszsoftmax.py:154-169 restricted to the configuration the reference trains with."""
    if classes != "present" or per_image:
        raise NotImplementedError("the HIP Lovasz path implements classes='present', per_image=False")
    if labels.dim() > 3:
        labels = labels.squeeze(1)
    labels = labels.contiguous().long()
    return _LovaszFn.apply(probas.float(), labels, ignore)


class LovaszSoftmax(nn.Module):
    def __init__(self, classes="present", per_image=False, ignore=None, n_branches=0, prev_out=False):
        super().__init__()
        self.classes, self.per_image, self.ignore = classes, per_image, ignore
        self.n = n_branches + 1
        self.prev_out = prev_out
        self._set_weights()

    def _set_weights(self):
        self.weights = torch.linspace(0, 1, self.n + 1)[1:] if self.prev_out else None

    def update_n(self, n):
        self.n = n + 1
        self._set_weights()

    def forward(self, y_pred, targets):
        losses = []
        for i in range(self.n):
            yi = y_pred[i]            # ExitLogits materialises one exit; a tensor indexes
            losses.append(lovasz_softmax(yi, targets, self.classes, self.per_image, self.ignore).unsqueeze(0))
        losses = torch.cat(losses)
        if self.prev_out:
            return torch.dot(self.weights.to(losses.device), losses).sum()
        return losses.sum()
