"""Fused multi-tensor SGD (torch.optim.SGD semantics, deepv3_funcs.py:99) - one HIP
launch per parameter group instead of ~5 elementwise launches per tensor."""
import ctypes as C

import torch

from . import _lib, engine


def _same_layout(g, p):
    """True when g and p place element [i,j,...] at the same offset (strides of size-1 dims are irrelevant)."""
    return g.shape == p.shape and all(a == b for a, b, n in zip(g.stride(), p.stride(), p.shape) if n > 1)


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, grad_scale=1.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.grad_scale = grad_scale
        self._tables = {}          # group index -> (key, table, sizes, lrs, lr value)
        self.static_grads = False  # set by GradArena users: zero_grad must keep the .grad views

    def zero_grad(self, set_to_none=True):
        if self.static_grads:
            return                 # arena gradients are overwritten by the next backward
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
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
                if not _same_layout(g, p):                          # kernels index raw memory
                    if self.static_grads:
                        # arena / HIP-graph mode: the backward kernels keep writing the arena, so a private copy would
                        # freeze this gradient at its first value (and fall out of the data-parallel buckets)
                        raise _lib.EesegError(f"gradient of a {tuple(p.shape)} parameter is laid out differently from "
                                              "the parameter; static (arena) gradients cannot be re-bound")
                    g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                    g2.copy_(g)
                    p.grad = g = g2
                if g.dtype != torch.float32 or p.dtype != torch.float32:
                    raise _lib.EesegError("fused SGD expects fp32 master parameters and gradients")
                rows.append((p.data_ptr(), g.data_ptr(), st["momentum_buffer"].data_ptr()))
                sizes.append(p.numel())
            key = tuple(rows)
            cached = self._tables.get(gi)
            if cached is None or cached[0] != key:
                table = torch.tensor(rows, dtype=torch.int64).to(dev)
                sz = torch.tensor(sizes, dtype=torch.int64).to(dev)
                lrs = torch.full((len(ps),), float(group["lr"]), dtype=torch.float32, device=dev)
                cached = [key, table, sz, lrs, float(group["lr"])]
                self._tables[gi] = cached
            elif cached[4] != float(group["lr"]):
                cached[3].fill_(float(group["lr"]))
                cached[4] = float(group["lr"])
            _, table, sz, lrs, _ = cached
            rc = _lib.lib().eeseg_sgd_step(C.c_void_p(table.data_ptr()), C.c_void_p(sz.data_ptr()),
                                           C.c_void_p(lrs.data_ptr()), len(ps), float(group["momentum"]),
                                           float(group["weight_decay"]), float(self.grad_scale), int(first),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, "eeseg_sgd_step")
        engine.bump_weights_epoch()
        return loss

    def sync_lr(self):
        """Push changed learning rates to the device tables (call after scheduler.step()
        when the step itself is replayed from a captured graph)."""
        for gi, group in enumerate(self.param_groups):
            cached = self._tables.get(gi)
            if cached is not None and cached[4] != float(group["lr"]):
                cached[3].fill_(float(group["lr"]))
                cached[4] = float(group["lr"])
