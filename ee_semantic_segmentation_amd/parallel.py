"""Data-parallel training over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The reference has no parallelism at all (nn.DataParallel is commented
out at train_funcs.py:72-74), so this is new: gradients are all-reduced in
buckets that fill in reverse-layer order while backward is still running
(classifier / deep heads first), each bucket flattened by one kernel and reduced
asynchronously on RCCL's stream; ``finish()`` makes the compute stream wait before
the optimizer step.  xGMI is point-to-point, so buckets are large (default 64 MiB)
- few, big collectives keep every link busy instead of paying per-call latency.
BatchNorm statistics (engine.Config.sync_bn) and the CE valid-pixel count are
all-reduced inside the respective layers.
"""
import torch
import torch.distributed as dist


def _flat_view(t):
    """1-D view of a dense tensor in its PHYSICAL element order (no copy)."""
    if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)
    return t.reshape(-1)


class GradReducer:
    def __init__(self, module, bucket_bytes=64 << 20, group=None, average=True):
        self.module, self.group, self.average = module, group, average
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        # backward produces gradients roughly in reverse registration order
        self.buckets, cur, size = [], [], 0
        for p in reversed(params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._where = {}
        for b, ps in enumerate(self.buckets):
            for p in ps:
                self._where[p] = b
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params] if self.world > 1 else []
        self.reset()

    def reset(self):
        self._pending = [len(ps) for ps in self.buckets]
        self._works = []

    def _on_grad(self, p):
        b = self._where[p]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        ps = [p for p in self.buckets[b] if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([_flat_view(p.grad) for p in ps])
        op = dist.ReduceOp.AVG if (self.average and dist.get_backend(self.group) == "nccl") else dist.ReduceOp.SUM
        work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self._works.append((work, flat, ps, op))

    def finish(self):
        """Wait for every bucket, then point each .grad at its reduced slice."""
        if self.world == 1:
            return
        for b, n in enumerate(self._pending):      # parameters that never received a gradient
            if n > 0:
                self._launch(b)
        for work, flat, ps, op in self._works:
            work.wait()
            if self.average and op == dist.ReduceOp.SUM:
                flat.div_(self.world)
            off = 0
            for p in ps:
                n = p.numel()
                seg = flat[off:off + n]
                if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
                    co, ci, r, s = p.shape
                    p.grad = seg.view(co, r, s, ci).permute(0, 3, 1, 2)
                else:
                    p.grad = seg.view(p.shape)
                off += n
        self.reset()


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s weights and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            flat = _flat_view(t.data)
            dist.broadcast(flat, src, group=group)
