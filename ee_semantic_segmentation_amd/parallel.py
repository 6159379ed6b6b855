"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI
called through libeeseg's C ABI (``comm.DataParallelComm``); ``torch.distributed`` does the
rendezvous (any backend - gloo is enough) and the CPU tests.

The reference has no parallelism at all (nn.DataParallel is commented
out at train_funcs.py:72-74), so this is new: gradients are all-reduced in
buckets that fill in reverse-layer order while backward is still running
(classifier / deep heads first); a bucket is a plain slice of the gradient arena,
reduced on the package's own lane stream beside the rest of backward; ``finish()``
makes the compute stream wait before the optimizer step.  xGMI is point-to-point, so
buckets are large (default 32 MiB, SURVEY section 5: 25-50 MB; unmeasured with more than one rank) - few, big
collectives keep every link busy instead of paying per-call latency, while the LAST bucket (it ends with the stem
and is launched when backward ends, so its all-reduce is exposed) is capped at ArenaReducer.TAIL_BYTES (4 MiB): the
bucket before it is closed early.  BatchNorm statistics (engine.Config.sync_bn) and the CE
valid-pixel count are all-reduced inside the respective layers.  No c10d ``Work`` is
ever created for a device tensor: comm.py says why.
"""
import os

import torch
import torch.distributed as dist

from . import engine


def _flat_view(t):
    """1-D view of a dense tensor in its PHYSICAL element order (no copy)."""
    if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)
    return t.reshape(-1)


class GradReducer:
    """Autograd-hook bucket reducer over torch.distributed for HOST tensors (gloo): the CPU rehearsal of the bucket
    bookkeeping.  The HIP path uses ArenaReducer (RCCL through libeeseg); device parameters are refused."""

    def __init__(self, module, bucket_bytes=64 << 20, group=None, average=True):
        if any(p.is_cuda for p in module.parameters()):
            raise RuntimeError("GradReducer is the CPU/gloo rehearsal; on the GPU use net.enable_grad_arena() + ArenaReducer")
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
    """Make every rank start from rank `src`'s weights and buffers.  Device tensors travel over the network's RCCL
    communicator (engine.Config.comm), host tensors over torch.distributed."""
    cfg = getattr(module, "cfg", None)
    comm = getattr(cfg, "comm", None)
    if comm is None and (not dist.is_initialized() or dist.get_world_size(group) == 1):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            flat = _flat_view(t.data)
            if comm is not None and flat.is_cuda:
                if comm.world > 1:
                    comm.stat.broadcast(flat, src)
            elif flat.is_cuda and getattr(cfg, "collective", None) is not None:
                host = flat.cpu()                  # test transport (gloo): staged through the host, once, at start-up
                dist.broadcast(host, src, group=group)
                flat.copy_(host)
            elif flat.is_cuda:
                raise RuntimeError("device parameters need parallel.init_data_parallel(net) before broadcast_parameters")
            else:
                dist.broadcast(flat, src, group=group)


def init_data_parallel(net, group=None, sync_bn=True, broadcast=True, transport=None):
    """Attach the data-parallel machinery to `net` (one call per process, after torch.distributed.init_process_group and
    torch.cuda.set_device): the RCCL communicators (rendezvous over `group`), SyncBN so that BatchNorm sees the global
    batch like the reference's single device does (main_bradeepv3.py:119 trains at batch 32), and identical start
    weights.  -> the DataParallelComm, or None when there is nothing to do (no process group / one rank).

    `transport=(collective, gatherer)`: test hook - callables (tensor, group) used instead of the RCCL communicators
    (engine.Config.collective / .gatherer; the world-2 tests on ONE GPU stage device tensors through gloo, where two
    ranks cannot share an RCCL communicator)."""
    from . import comm as C_
    if not dist.is_initialized():
        return None
    world = dist.get_world_size(group)
    if world == 1 and not C_.forced():
        return None
    cfg = net.cfg
    cfg.group = group
    if transport is not None:
        cfg.collective, cfg.gatherer = transport
    elif cfg.comm is None:
        # Communicator creation is COLLECTIVE (ncclCommInitRank blocks until every rank has joined): a rank that cannot even
        # bind librccl must not leave the others waiting forever.  So the non-collective part is checked on every rank first
        # and the verdict is agreed on over the rendezvous group (host tensors: any backend that moves them, i.e. gloo).
        ok, why = 1, ""
        try:
            C_.rccl_version()
        except Exception as exc:                   # noqa: BLE001 - whatever the loader says is the message
            ok, why = 0, repr(exc)
        if world > 1:
            # over a HOST-ONLY channel whatever the caller's backend (comm.host_group: the group itself for gloo, a gloo
            # subgroup of it for an NCCL rendezvous group) - an NCCL group would move the flag through device tensors
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=C_.host_group(group))
            if int(flag.item()) == 0:
                raise RuntimeError("RCCL cannot be bound on at least one rank of the job (this rank: "
                                   + (why or "ok") + "); no communicator was created on any rank")
        elif not ok:
            raise RuntimeError("RCCL cannot be bound: " + why)
        cfg.comm = C_.DataParallelComm(group, next(net.parameters()).device)
    cfg.sync_bn = bool(sync_bn)
    if broadcast:
        broadcast_parameters(net, 0, group)
    return cfg.comm


def dp_info(net=None):
    """(world, rank) of the data-parallel job this process belongs to; (1, 0) without one."""
    cfg = getattr(net, "cfg", None)
    if cfg is not None and (cfg.comm is not None or cfg.collective is not None):
        return cfg.dp_world(), cfg.dp_rank()
    if dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


class ShardSampler(torch.utils.data.Sampler):
    """Index stream of ONE rank for a DataLoader(batch_size = global_batch // world) (SURVEY 8e "DistributedSampler-style
    seeding"): one permutation of the data set per (seed, epoch), identical on every rank; global batch k is
    perm[k*B:(k+1)*B] - exactly what a single process with batch B would train on - and rank r takes the r-th
    contiguous b-sample slice of it.  With more than one rank the ragged last global batch is dropped: SyncBN's
    count * world, the averaging reducer and the exact-Lovasz all-gather assume equal shards on every rank."""

    def __init__(self, n, global_batch, world=1, rank=0, seed=0, shuffle=True):
        if global_batch % world:
            raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
        self.n, self.B, self.world, self.rank, self.seed, self.shuffle = n, global_batch, world, rank, seed, shuffle
        self.b = global_batch // world
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _order(self):
        if not self.shuffle:
            return list(range(self.n))
        g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
        return torch.randperm(self.n, generator=g).tolist()

    def __iter__(self):
        order = self._order()
        full = self.n // self.B
        for k in range(full):
            lo = k * self.B + self.rank * self.b
            yield from order[lo:lo + self.b]
        if self.world == 1:                        # single process: keep the reference's drop_last=False tail
            yield from order[full * self.B:]

    def __len__(self):
        full = self.n // self.B
        return full * self.b + (self.n - full * self.B if self.world == 1 else 0)


def eval_shard(dataset, world, rank):
    """Evaluation under data parallelism: rank r scores samples r, r + world, ... and the per-exit counters are summed
    over the ranks once per evaluation (eval_mIoU.mIoU_evaluator); shards may be ragged."""
    if world == 1:
        return dataset
    return torch.utils.data.Subset(dataset, list(range(rank, len(dataset), world)))


class ArenaReducer:
    """Gradient all-reduce over slices of a GradArena (engine.GradArena).

    The arena is laid out in backward completion order, so bucket k is simply
    ``flat[b_k : b_{k+1}]``; it is launched (async, on RCCL's stream) as soon as the
    last unit it covers has finished its backward, overlapping with the rest of
    backward.  No flatten / unflatten copies."""

    def __init__(self, net, bucket_bytes=32 << 20, group=None, average=True, reserve_cus=None):
        self.cfg, self.group, self.average = net.cfg, group, average
        # CUs left to the RCCL kernels while buckets are in flight: the 256-tile conv kernels run one block per CU and
        # size their rounds / K splits for the CUs they can get, so a launch planned for 256 CUs needs a second round
        # when a collective holds a few of them.  Measured on one GPU with a stand-in for the collective's resident workgroups
        # (scripts/contention_probe.py): at 4 images per GPU planning for 224 CUs costs nothing and gives
        # back half of what the held CUs cost, at 8 images per GPU it costs 2 % and wins 1.3 % - so the default is 32 from
        # 8 ranks on (the metric's 4-image shards) and 0 (the whole chip) below.  EESEG_RCCL_RESERVE_CUS overrides.
        self._reserve_arg = reserve_cus
        self.reserve_cus = 0
        self._reserved = False
        self.arena = net.cfg.arena
        if self.arena is None:
            raise RuntimeError("call net.enable_grad_arena() first")
        self.comm = getattr(net.cfg, "comm", None)
        if self.comm is not None:
            self.world = self.comm.world
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # EESEG_FORCE_ALLREDUCE=1: issue the collectives even in a 1-rank group (lets a single
        # GPU exercise the RCCL-inside-HIP-graph path the multi-GPU bench relies on)
        forced = os.environ.get("EESEG_FORCE_ALLREDUCE") == "1"
        self.active = self.world > 1 or (forced and (self.comm is not None or dist.is_initialized()))
        env = os.environ.get("EESEG_RCCL_RESERVE_CUS")
        if self._reserve_arg is not None:
            self.reserve_cus = int(self._reserve_arg)
        elif env is not None:
            self.reserve_cus = int(env)
        else:
            self.reserve_cus = 32 if self.world >= 8 else 0
        if not 0 <= self.reserve_cus <= 192:
            raise ValueError(f"reserve_cus = {self.reserve_cus}: 0 .. 192")
        self.buckets = self.plan_buckets(self.arena.unit_ranges, bucket_bytes)      # (first unit, last unit, start, end)
        self._done = set()
        self._next = 0
        self._host_div = []
        self.launched = 0            # buckets launched so far (host-side count: eager steps and captures)
        if self.active:
            self.cfg.on_unit_done = self._unit_done

    TAIL_BYTES = 4 << 20         # the last bucket's all-reduce is exposed (nothing of backward is left to hide it): keep it small

    @classmethod
    def plan_buckets(cls, unit_ranges, bucket_bytes):
        """[(first unit, last unit, start, end)] over the arena's units (backward completion order): a bucket closes when
        it holds >= bucket_bytes; the final bucket is then cut so that what is launched after the LAST unit (the stem) is at
        most TAIL_BYTES - the units before the cut leave with the bucket before, or as a bucket of their own."""
        buckets = []
        u0, start = 0, 0
        last = len(unit_ranges) - 1
        for u, (a, b) in enumerate(unit_ranges):
            if (b - start) * 4 >= bucket_bytes or u == last:
                buckets.append((u0, u, start, b))
                u0, start = u + 1, b
        if buckets:
            f0, f1, fa, fb = buckets[-1]
            if (fb - fa) * 4 > cls.TAIL_BYTES and f1 > f0:
                cut = f1                                   # first unit of the tail: as early as the cap allows
                while cut - 1 > f0 and (fb - unit_ranges[cut - 1][0]) * 4 <= cls.TAIL_BYTES:
                    cut -= 1
                if (fb - unit_ranges[cut][0]) * 4 <= cls.TAIL_BYTES or cut == f1:
                    mid = unit_ranges[cut][0]
                    buckets[-1:] = [(f0, cut - 1, fa, mid), (cut, f1, mid, fb)]
        return buckets

    def _unit_done(self, uid):
        self._done.add(uid)
        while self._next < len(self.buckets):
            u0, u1, a, b = self.buckets[self._next]
            if not all(u in self._done for u in range(u0, u1 + 1)):
                break
            self._launch(a, b)
            self._next += 1

    def _plan_cus(self, cus):
        from ._lib import lib
        lib().eeseg_set_option(8, cus)                    # EESEG_OPT_CONV_CUS
        lib().eeseg_set_wgrad_big_grid(cus, 8)

    def _launch(self, a, b):
        if self.reserve_cus > 0 and not self._reserved:
            self._plan_cus(256 - self.reserve_cus)
            self._reserved = True
        seg = self.arena.flat[a:b]
        self.launched += 1
        hook = getattr(self.cfg, "collective", None)
        if hook is not None:                        # test hook (engine.Config.collective): synchronous sum
            hook(seg, self.group)
            self._host_div.append(seg)
        elif self.comm is not None:
            # the bucket's units are complete on the compute stream: fork the lane there, reduce on the lane
            from .comm import AVG, SUM
            lane = self.comm.lane_g
            lane.fork()
            self.comm.grad.all_reduce(seg, AVG if self.average else SUM, stream=lane.stream)
        elif seg.is_cuda:
            raise RuntimeError("device gradients need parallel.init_data_parallel(net) (RCCL through libeeseg)")
        else:                                       # host arena (gloo rehearsal of the bucket schedule)
            dist.all_reduce(seg, group=self.group)
            self._host_div.append(seg)

    def finish(self):
        if not self.active:
            return
        while self._next < len(self.buckets):       # units that produced no gradient this step
            _, _, a, b = self.buckets[self._next]
            self._launch(a, b)
            self._next += 1
        if self.comm is not None:
            self.comm.lane_g.join()                 # the optimizer step waits for the last bucket
        for seg in self._host_div:
            if self.average:
                seg.div_(self.world)
        if self._reserved:
            self._plan_cus(256)
            self._reserved = False
        self.reset()

    def reset(self):
        self._done.clear()
        self._next = 0
        self._host_div = []

    def abort_step(self):
        """After a step that did not reach finish() (a failed graph capture): forget its buckets and give the conv launch
        plans the whole chip back."""
        if self._reserved:
            self._plan_cus(256)
            self._reserved = False
        self.reset()


class GraphedTrainStep:
    """One training step (forward, loss, backward, gradient all-reduce, SGD step) captured
    into a HIP graph and replayed: ~1000 kernel launches per step leave the Python
    critical path.  Needs the gradient arena (static gradient addresses) and this
    package's SGD.  The first `warmup` calls run eagerly (allocator warm-up, momentum
    buffer initialisation), the next call captures, later calls replay."""

    def __init__(self, net, criterion, optimizer, reducer=None, warmup=2, use_graph=True):
        self.net, self.criterion, self.opt, self.reducer = net, criterion, optimizer, reducer
        self.warmup, self.use_graph = warmup, use_graph
        self.calls = 0
        self.graph = None
        self.X = self.y = self.loss = None
        if net.cfg.arena is not None:
            optimizer.static_grads = True
        self._bns = [m for m in net.modules() if type(m).__name__ == "BatchNorm2d"]

    def _eager(self, X, y):
        arena = self.net.cfg.arena
        if arena is not None and not self.net.cfg.accumulate:
            engine.pack_all(self.net, self.net.cfg.compute_dtype)      # one launch instead of one per conv
            arena.prezeroed = True                                     # one memset instead of one per wgrad
            arena.flat.zero_()
        out = self.net(X)
        loss = self.criterion(out, y)
        self.opt.zero_grad(set_to_none=True)
        loss.mean().backward()
        self.net.cfg.run_deferred()
        self.net.cfg.join_side()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        self.net.cfg.end_step()
        if arena is not None:
            arena.prezeroed = False
        return loss.detach()

    def __call__(self, X, y):
        self.calls += 1
        if not self.use_graph or self.calls <= self.warmup:
            return self._eager(X, y)
        if self.graph is None:
            if self.net.cfg.arena is None:
                raise RuntimeError("graph capture needs net.enable_grad_arena()")
            self.X, self.y = X.clone(), y.clone()
            before = [b._pending_batches for b in self._bns]
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                # thread_local: other threads of the host process (a c10d watchdog when the caller's rendezvous group is
                # NCCL, data-loader pin threads) may call HIP APIs that are illegal during a GLOBAL-mode capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self.loss = self._eager(self.X, self.y)
            except Exception as exc:                # e.g. a collective that cannot be captured
                import warnings
                torch.cuda.synchronize()
                if self.reducer is not None:
                    self.reducer.abort_step()       # bucket bookkeeping and the reduced CU plan of the aborted capture
                if self.reducer is not None and self.reducer.active and self.reducer.world > 1:
                    # A per-rank fallback would leave THIS rank eager while the others replay graphs - and the collectives
                    # enqueued inside the aborted capture never ran here, so the ranks' collective sequences diverge: a hang
                    # without an error, or wrong sums.  In a multi-rank job a failed capture is fatal.
                    raise RuntimeError(f"HIP-graph capture of the data-parallel training step failed on rank "
                                       f"{self.net.cfg.dp_rank()} ({exc!r}); refusing a per-rank eager fallback (the ranks' "
                                       "collective sequences would diverge) - rerun with use_graph=False on every rank") from exc
                warnings.warn(f"HIP-graph capture of the training step failed ({exc!r}); running eagerly")
                # a closure held back by conv_bn_bwd during the aborted capture points into the graph's private pool
                self.net.cfg.reset_transients()
                for b, n in zip(self._bns, before):
                    b._pending_batches = n
                self.use_graph = False
                return self._eager(X, y)
            self.graph = graph
            self._active = [b for b, n in zip(self._bns, before) if b._pending_batches != n]
            for b, n in zip(self._bns, before):
                b._pending_batches = n              # capture itself does not run the kernels
        if X.data_ptr() != self.X.data_ptr():
            self.X.copy_(X, non_blocking=True)
        if y.data_ptr() != self.y.data_ptr():
            self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        for b in self._active:
            b._pending_batches += 1
        engine.bump_weights_epoch()      # eager code after a replay must re-pack the updated weights
        return self.loss
