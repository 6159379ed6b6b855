"""Fused multi-tensor SGD (torch.optim.SGD semantics, deepv3_funcs.py:99) - one HIP
launch per parameter group instead of ~5 elementwise launches per tensor."""
import ctypes as C

import torch

from . import _lib, engine


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, grad_scale=1.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.grad_scale = grad_scale

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if not ps[0].is_cuda:
                raise _lib.EesegError("the fused SGD step needs device parameters (no CPU fallback)")
            first = False
            rows, sizes = [], []
            for p in ps:
                st = self.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.preserve_format)
                    first = True
                g = p.grad
                if g.stride() != p.stride():          # kernels index raw memory: layouts must agree
                    g = g.contiguous(memory_format=torch.preserve_format)
                    if g.stride() != p.stride():
                        g = torch.empty_like(p).copy_(p.grad)
                    p.grad = g
                if g.dtype != torch.float32 or p.dtype != torch.float32:
                    raise _lib.EesegError("fused SGD expects fp32 master parameters and gradients")
                rows.append([p.data_ptr(), g.data_ptr(), st["momentum_buffer"].data_ptr()])
                sizes.append(p.numel())
            table = torch.tensor(rows, dtype=torch.int64).to(dev, non_blocking=True)
            sz = torch.tensor(sizes, dtype=torch.int64).to(dev, non_blocking=True)
            lrs = torch.full((len(ps),), float(group["lr"]), dtype=torch.float32, device=dev)
            rc = _lib.lib().eeseg_sgd_step(C.c_void_p(table.data_ptr()), C.c_void_p(sz.data_ptr()),
                                           C.c_void_p(lrs.data_ptr()), len(ps), float(group["momentum"]),
                                           float(group["weight_decay"]), float(self.grad_scale), int(first),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, "eeseg_sgd_step")
            # keep the tables alive until the launch has consumed them
            self._keep = getattr(self, "_keep", [])[-8:] + [(table, sz, lrs)]
        engine.bump_weights_epoch()
        return loss
