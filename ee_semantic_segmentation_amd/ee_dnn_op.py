"""Progressive per-image early-exit inference with the SIMILARITY gate (ee_dnn_op.py:40-118): the first
non-ignored branch only provides the reference label map; every later one exits when
``metric(previous map, its map)`` passes the threshold.  Same ``__call__`` dict contract as the reference
(``exit``, ``exit_flops``, ``exit_flops_2``, ``edge_flops``, ``edge_flops_2``, ``n``, ``last``, ``last_flops``,
``last_flops_2``; the ``_2`` variants leave out the first evaluated branch, ee_dnn_op.py:88-90).  FLOPs come from
the analytic conv-MAC counter; the two label maps are compared through the on-device contingency table of
``eeseg_argmax_pair_hist`` (low-res logits of both exits in, one C x C int table out).  ``metric`` is a function
of that table: use ``eval_br_sim.gate_function(name, ignore)[0]`` or any ``sim_metrics.*_from_table``; a
``sim_metrics.SSIM`` instance is evaluated on the two label maps (``eeseg_ssim_labels``).
``stop_at_exit=True`` really stops after the exit; by default the backbone is finished and ``last``
is reported, like the reference."""
import torch

from . import kernels as K
from .ee_dnn_op_ne import section_flops
from .from_deepv3_new import head_macs
from .sim_metrics import SSIM


class eval_ee_deeplabv3:
    def __init__(self, ee_model, metric, th, less_than=True, ignore=(), device=torch.device("cuda"), stop_at_exit=False):
        self.model = ee_model
        self.n = ee_model.n_branches
        self.ignore = list(ignore)
        self.metric = metric
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
        ref_lr = None
        left = False
        x = X.unsqueeze(0).to(self.device)
        h, w = H, W
        for i in range(self.n):
            f, h, w = section_flops(self.model.base_model[i], h, w)
            main_flops.append(f)
            x = self.model.base_model[i](x)
            if i not in self.ignore and not left:
                lr = self.model.branches[i](x).contiguous()
                branch_flops.append(2 * head_macs(self.model.branches[i], h, w))
                similar = False
                if ref_lr is not None and isinstance(self.metric, SSIM):
                    maps = [K.argmax_confusion(l, C, None, H, W, want_pred=True)[1] for l in (ref_lr, lr)]
                    t = float(K.ssim_labels(maps[0], maps[1], self.metric.dr)[0])
                    similar = (t < self.threshold) if self.less_than else (t > self.threshold)
                elif ref_lr is not None:
                    t = self.metric(K.argmax_pair_hist(ref_lr, lr, C, H, W)[0].double())
                    similar = (t < self.threshold) if self.less_than else (t > self.threshold)
                if similar:
                    _, pred = K.argmax_confusion(lr, C, None, H, W, want_pred=True)
                    output["exit"] = pred[0].cpu()
                    output["exit_flops"] = sum(branch_flops) + sum(main_flops)
                    output["exit_flops_2"] = sum(branch_flops[1:]) + sum(main_flops)
                    output["edge_flops"] = output["exit_flops"]
                    output["edge_flops_2"] = output["exit_flops_2"]
                    output["n"] = i + 1
                    left = True
                    if self.stop_at_exit:
                        return output
                else:
                    ref_lr = lr
            if not left and i == self.last_br:
                output["edge_flops"] = sum(branch_flops) + sum(main_flops)
                output["edge_flops_2"] = sum(branch_flops[1:]) + sum(main_flops)
        f, h, w = section_flops(self.model.base_model[-1], h, w)
        main_flops.append(f)
        x = self.model.base_model[-1](x)
        main_flops.append(2 * head_macs(self.model.classifier, h, w))
        lr = self.model.classifier(x).contiguous()
        _, pred = K.argmax_confusion(lr, C, None, H, W, want_pred=True)
        output["last"] = pred[0].cpu()
        output["last_flops"] = sum(branch_flops) + sum(main_flops)
        output["last_flops_2"] = sum(branch_flops[1:]) + sum(main_flops)
        if not left:
            output["exit"] = output["last"]
            output["exit_flops"] = output["last_flops"]
            output["exit_flops_2"] = output["last_flops_2"]
            output["n"] = self.n + 1
        return output
