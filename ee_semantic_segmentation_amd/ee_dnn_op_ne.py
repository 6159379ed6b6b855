"""Progressive per-image early-exit inference with the entropy gate
(ee_dnn_op_ne.py:40-108).  Same ``__call__`` dict contract (``exit``, ``exit_flops``,
``edge_flops``, ``n``, ``last``, ``last_flops``); FLOPs come from the analytic conv-MAC
counter instead of re-tracing pthflops on every call, and the gate is the fused device
kernel (one 4-byte D2H per gated branch instead of a [C,H,W] probability map).
Like the reference this is a FLOP-accounting simulator: the backbone is always finished
(ee_dnn_op_ne.py:91-101); ``stop_at_exit=True`` really stops."""
import torch

from . import kernels as K
from .eval_br_ent import img_norm_entropy  # noqa: F401  (the reference imports it from here)
from .from_deepv3_new import _conv_out, block_macs, conv_macs, head_macs
from .nn_modules import Bottleneck, Conv2d, MaxPool2d


def section_flops(section, h, w):
    """2 x conv MACs of one backbone section at feature size (h, w) -> (flops, h_out, w_out)."""
    tot = 0
    for m in section:
        if isinstance(m, Conv2d):
            c, h, w = conv_macs(m, h, w)
            tot += c
        elif isinstance(m, MaxPool2d):
            h, w = _conv_out(h, 3, 2, 1, 1), _conv_out(w, 3, 2, 1, 1)
        elif isinstance(m, Bottleneck):
            c, h, w = block_macs(m, h, w)
            tot += c
    return 2 * tot, h, w


class eval_ee_deeplabv3:
    def __init__(self, ee_model, metric, th, less_than=True, ignore=(), device=torch.device("cuda"), stop_at_exit=False):
        self.model = ee_model
        self.n = ee_model.n_branches
        self.ignore = list(ignore)
        self.metric = metric                    # an img_norm_entropy instance
        self.less_than = less_than
        self.threshold = th
        self.device = device
        self.stop_at_exit = stop_at_exit
        self.last_br = max([i for i in range(self.n) if i not in self.ignore], default=-1)

    @torch.no_grad()
    def __call__(self, X):
        output = dict()
        H, W = X.shape[-2:]
        C = self.model.num_classes
        main_flops, branch_flops = [], []
        left = False
        x = X.unsqueeze(0).to(self.device)
        h, w = H, W
        for i in range(self.n):
            f, h, w = section_flops(self.model.base_model[i], h, w)
            main_flops.append(f)
            x = self.model.base_model[i](x)
            if i not in self.ignore and not left:
                lr = self.model.branches[i](x)
                branch_flops.append(2 * head_macs(self.model.branches[i], h, w))
                mode = (2 if self.metric.pool_min else 1) if self.metric.pool else 0
                ent, _ = K.entropy_gate(lr.contiguous(), C, H, W, self.threshold, mode, self.metric.size[0])
                t = float(ent[0].item())
                if (t < self.threshold) == bool(self.less_than):
                    _, pred = K.argmax_confusion(lr.contiguous(), C, None, H, W, want_pred=True)
                    output["exit"] = pred[0].cpu()
                    output["exit_flops"] = sum(branch_flops) + sum(main_flops)
                    output["edge_flops"] = output["exit_flops"]
                    output["n"] = i + 1
                    left = True
                    if self.stop_at_exit:
                        return output
            if not left and i == self.last_br:
                output["edge_flops"] = sum(branch_flops) + sum(main_flops)
        f, h, w = section_flops(self.model.base_model[-1], h, w)
        main_flops.append(f)
        x = self.model.base_model[-1](x)
        main_flops.append(2 * head_macs(self.model.classifier, h, w))
        lr = self.model.classifier(x)
        _, pred = K.argmax_confusion(lr.contiguous(), C, None, H, W, want_pred=True)
        output["last"] = pred[0].cpu()
        output["last_flops"] = sum(branch_flops) + sum(main_flops)
        if not left:
            output["exit"] = output["last"]
            output["exit_flops"] = output["last_flops"]
            output["n"] = self.n + 1
        return output
