"""Explicit forward / backward of the DeepLabV3 building blocks over libeeseg.

Nothing here is traced or compiled: each function issues the HIP kernels of one
fused layer group on the current stream and returns the tensors backward needs.
The autograd Functions in ``nn_modules`` wrap whole blocks (stem, bottleneck,
head) so gradient accumulation at residual / ASPP joins also happens inside
libeeseg kernels (data-gradient calls accumulate in place through the conv
epilogue's residual input).

Reference ops restated: torchvision Bottleneck / ASPP /
DeepLabHead reached from from_deepv3_new.py:146-151.
"""
import torch
import torch.distributed as dist

from . import kernels as K

CPAD = 32           # classifier output channels are padded to one 128-byte row
_WEIGHTS_EPOCH = [0]


def bump_weights_epoch():
    """Called by optimizers that update parameters through raw pointers."""
    _WEIGHTS_EPOCH[0] += 1


class Config:
    """Run-time switches shared by every module of one network."""

    def __init__(self):
        self.compute_dtype = torch.float32   # torch.bfloat16 = throughput mode
        self.sync_bn = False                 # all-reduce BN statistics over the process group
        self.group = None
        self.collective = None               # test hook: callable(tensor, group) used instead of dist.all_reduce
        self.gatherer = None                 # test hook: callable(tensor, group) -> [world, *tensor.shape]
        self.dropout_seed = 0x5EED
        self._drop_calls = 0
        self.arena = None                    # GradArena: parameter gradients written in place
        self.accumulate = False              # arena mode: add to the stored gradients instead of overwriting
        self.step_counter = None             # device int64[1]; lets a captured graph draw fresh dropout masks
        self.on_unit_done = None             # callable(unit_id): gradient bucket scheduling (GradReducer)
        self.merge_aspp_dgrad = __import__("os").environ.get("EESEG_MERGE_ASPP_DGRAD", "1") != "0"   # one data-gradient launch per ASPP head
        self.fuse_block_residual = __import__("os").environ.get("EESEG_FUSE_BLOCK_RESIDUAL", "1") != "0"   # identity blocks: dout * mask is added by conv1's data-gradient, not written by BN backward
        self.overlap_wgrad = False           # opt-in (measured +-0 with the 256-tile kernels): weight-gradient on a side stream, concurrent with the data-gradient:
        self._side = None                    # the two kernels fill each other's partially filled last block round
        self._side_busy = False
        self._side_keep = []                 # tensors the side stream still reads (freed after the join)
        # SyncBN backward: the weight gradient of layer L is held back and issued right after layer L-1's statistics
        # all-reduce has been launched on the side lane (comm.DataParallelComm.lane_s), so it runs while that latency-bound
        # collective is in flight; in the captured graph the two are parallel branches.  Same mechanism as the gradient
        # buckets (RCCL through the C ABI on a package-owned stream).  OPT-IN (EESEG_DEFER_WGRAD=1): the only measurement that
        # exists - the 1-rank rehearsal, where a collective has no latency to hide - has it 0.4 ms slower (27.10 vs 26.67 ms at
        # 4 images), and no run with more than one rank has executed it; default = collective on the compute stream, weight
        # gradient in place.
        self.defer_wgrad = __import__("os").environ.get("EESEG_DEFER_WGRAD", "0") == "1"
        # round 4: the weight gradients of a unit (the three convs of a bottleneck block) are queued and issued as ONE launch where the
        # library can put them side by side (eeseg_conv_wgrad_group: per-GPU shards of a few images; bigger batches are issued one by
        # one, just later) - arena mode only (the kernel writes the gradient in place, nothing to hand back to autograd)
        self.group_wgrad = __import__("os").environ.get("EESEG_GROUP_WGRAD", "1") == "1"
        # every ReLU layer's forward can leave the 1-bit mask (1/16 of a tensor pass) so that its backward reads it instead of recomputing
        # the sign from the conv output and (scale, shift)
        # (EESEG_BN_MASK_ALL: 0 = residual layers only, the round-2 form; 1 = always; 2 = default: where the backward is two passes)
        self.bn_mask_all = int(__import__("os").environ.get("EESEG_BN_MASK_ALL", "2"))
        self._wgrad_queue = []
        self._deferred = None
        self.comm = None                     # comm.DataParallelComm: RCCL through libeeseg (parallel.init_data_parallel)

    def world(self):
        """Number of ranks BatchNorm statistics are reduced over (1 = local BN).  With
        EESEG_FORCE_ALLREDUCE=1 a 1-rank group still issues the collectives (rehearsal on one GPU)."""
        if not self.sync_bn:
            return 1
        return self.dp_world()

    def sync_active(self):
        return self.sync_bn and self.dp_active()

    def dp_world(self):
        """Ranks of the data-parallel group (independent of sync_bn): the CE valid-pixel count and the exact
        Lovasz mode are always global over it."""
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def dp_active(self):
        import os
        forced = os.environ.get("EESEG_FORCE_ALLREDUCE") == "1"
        if self.comm is not None:
            return self.comm.world > 1 or forced
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or forced)

    def dp_rank(self):
        if self.comm is not None:
            return self.comm.rank
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def _no_transport(self, t):
        from ._lib import EesegError
        return EesegError(
            "a data-parallel collective on a device tensor needs the RCCL communicator of the HIP path: call "
            "parallel.init_data_parallel(net) after torch.distributed.init_process_group (any backend).  "
            "torch.distributed's own NCCL process group is deliberately not used for the data path (DESIGN.md section 7)")

    def all_gather(self, t):
        """-> [world, *t.shape]: `t` of every rank of the data-parallel group, in rank order."""
        if self.gatherer is not None:
            return self.gatherer(t, self.group)
        if self.comm is not None:
            return self.comm.stat_all_gather(t.contiguous())
        if t.is_cuda:
            raise self._no_transport(t)
        out = torch.empty((self.dp_world(),) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def reduce_scatter(self, t):
        """t [world, ...] on every rank -> this rank's slice of the sum over the ranks.  Test hook / host tensors: an
        all-reduce followed by the slice (same result, world times the traffic)."""
        if self.comm is not None and self.collective is None:
            return self.comm.stat_reduce_scatter(t.contiguous())
        full = self.all_reduce(t.contiguous().clone())
        return full[self.dp_rank()].clone()

    def all_reduce(self, t):
        """Sum `t` over the data-parallel group in place, on the compute stream.  Transports: the RCCL communicator
        (`comm`, the product), the `collective` test hook (callable(t, group): the world-2 parity tests stage device
        tensors through gloo), or torch.distributed itself for HOST tensors (gloo; CPU tests of the host algebra)."""
        if self.collective is not None:
            self.collective(t, self.group)
        elif self.comm is not None:
            self.comm.stat_all_reduce(t)            # two lanes: on the compute stream; single lane: fork -> lane -> join
        elif t.is_cuda:
            raise self._no_transport(t)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def all_reduce_begin(self, t):
        """Launch the in-place sum of `t` on the side lane; -> token for all_reduce_end.  Kernels issued on the compute
        stream between the two calls run beside the collective (parallel branches of a captured graph)."""
        if self.collective is not None or self.comm is None:
            self.all_reduce(t)
            return None
        lane = self.comm.lane_s
        lane.fork()
        self.comm.stat.all_reduce(t, stream=lane.stream)
        return lane

    @staticmethod
    def all_reduce_end(lane):
        if lane is not None:
            lane.join()                      # stream-level: the compute stream waits for the collective

    def reset_transients(self):
        """Drop everything a failed / abandoned step may have left behind (held-back weight gradient, side-stream state)."""
        self._deferred = None
        self._wgrad_queue = []
        self._side_busy = False
        self._side_keep = []
        if self.comm is not None:
            self.comm.lane_g.busy = self.comm.lane_s.busy = False

    def run_deferred(self):
        """Issue the weight gradient(s) that conv_bn_bwd held back (no-op when there is none)."""
        fn, self._deferred = self._deferred, None
        if fn is not None:
            fn()
        self.flush_wgrads()

    def queue_wgrad(self, item):
        """item = (x, dy, R, S, stride, pad, dil, out, accumulate) of kernels.conv_wgrad_group; the queue keeps x and dy alive."""
        self._wgrad_queue.append(item)
        if len(self._wgrad_queue) == 4:
            self.flush_wgrads()

    def flush_wgrads(self):
        q, self._wgrad_queue = self._wgrad_queue, []
        if q:
            K.conv_wgrad_group(q)

    def next_seed(self):
        self._drop_calls += 1
        return (self.dropout_seed * 0x9E3779B97F4A7C15 + self._drop_calls) & (2 ** 63 - 1)

    def step_dev(self, device):
        if self.step_counter is None or self.step_counter.device != device:
            self.step_counter = torch.zeros(1, dtype=torch.int64, device=device)
        return self.step_counter

    def end_step(self):
        """Advance the device-side step counter (captured into training graphs)."""
        if self.step_counter is not None:
            self.step_counter.add_(1)

    def side_stream(self, device):
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def join_side(self):
        """Make the current stream wait for the weight-gradient kernels issued on the side stream."""
        if self._side_busy:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            self._side_busy = False
            self._side_keep.clear()

    def gview(self, param):
        return None if self.arena is None else self.arena.kernel_view.get(param)

    def unit_done(self, module):
        self.run_deferred()                  # the unit's last weight gradient
        if self.overlap_wgrad == 2 and self.on_unit_done is None:
            return                           # nobody consumes the gradients before the optimizer step: join there
        self.join_side()                     # the unit's weight gradients are complete from here on
        if self.on_unit_done is not None:
            uid = module.__dict__.get("_eeseg_unit")
            if uid is not None:
                self.on_unit_done(uid)


class GradArena:
    """One flat fp32 buffer holding every parameter gradient of a network.

    ``p.grad`` of each parameter is a view into it and the backward kernels write
    there directly (conv weight gradients in KRSC = the channels_last parameter
    layout, BatchNorm (dbeta, dgamma) as one [2,C] pair, the class-padded classifier
    rows).  Units (stem, bottlenecks, heads) are laid out in REVERSE forward order, so
    the arena fills front to back during backward and data-parallel buckets are plain
    slices of it: no flatten / unflatten copies, static addresses for HIP-graph capture.
    """

    def __init__(self, net):
        units = []
        for i, sec in enumerate(net.base_model):
            mods = list(sec)
            j = 0
            if mods and type(mods[0]).__name__ == "Conv2d":
                units.append(("stem", sec, [(mods[0].weight, "stem"), (mods[1], "bn")]))
                j = 4
            for m in mods[j:]:
                units.append(("block", m, self._block_entries(m)))
            head = net.branches[i] if i < len(net.branches) else net.classifier
            units.append(("head", head, self._head_entries(head)))
        units.reverse()
        dev = next(net.parameters()).device
        off = 0
        plan = []
        self.unit_ranges = []
        for uid, (kind, mod, entries) in enumerate(units):
            start = off
            mod.__dict__["_eeseg_unit"] = uid
            for obj, what in entries:
                if what == "bn":
                    n = 2 * obj.weight.numel()
                elif what == "cls_w":
                    n = CPAD * obj.shape[1]
                elif what == "cls_b":
                    n = CPAD
                else:
                    n = obj.numel()
                n_pad = (n + 3) // 4 * 4
                plan.append((obj, what, off, n))
                off += n_pad
            self.unit_ranges.append((start, off))
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.prezeroed = False         # True: the step zeroes `flat` once, weight gradients accumulate into it
        self.kernel_view = {}          # param (or bn module) -> tensor the kernels write
        for obj, what, o, n in plan:
            seg = self.flat[o:o + n]
            if what == "bn":
                c = obj.weight.numel()
                pair = seg.view(2, c)
                self.kernel_view[obj] = pair
                obj.bias.grad = pair[0]
                obj.weight.grad = pair[1]
            elif what == "cls_w":
                co, ci = obj.shape[0], obj.shape[1]
                self.kernel_view[obj] = seg.view(CPAD, 1, 1, ci)
                obj.grad = self._grad_like(seg, obj)
            elif what == "cls_b":
                self.kernel_view[obj] = seg
                obj.grad = seg[:obj.numel()]
            elif what == "vec":
                self.kernel_view[obj] = seg
                obj.grad = seg
            else:                      # conv weight (incl. stem), stored channels_last = KRSC
                co, ci, r, s_ = obj.shape
                if not (obj.is_contiguous(memory_format=torch.channels_last) or (r == 1 and s_ == 1)):
                    raise RuntimeError("GradArena needs channels_last conv weights")
                kv = seg.view(co, r, s_, ci)
                self.kernel_view[obj] = kv
                obj.grad = self._grad_like(seg, obj)

    @staticmethod
    def _grad_like(seg, p):
        """View of the arena segment with EXACTLY the parameter's shape and strides.  torch leaves the strides of
        size-1 dims arbitrary (a [Co,Ci,1,1] channels_last weight keeps (Ci,1,1,1)), so a permuted KRSC view can differ
        from the parameter in those dims although the bytes are laid out identically; the fused SGD step indexes raw
        memory and must see the arena itself, never a copy."""
        if not p.permute(0, 2, 3, 1).is_contiguous():
            raise RuntimeError("GradArena needs conv weights whose physical order is KRSC (channels_last)")
        return torch.as_strided(seg, p.shape, p.stride())

    @staticmethod
    def _block_entries(b):
        e = [(b.conv1.weight, "w"), (b.bn1, "bn"), (b.conv2.weight, "w"), (b.bn2, "bn"), (b.conv3.weight, "w"),
             (b.bn3, "bn")]
        if b.downsample is not None:
            e += [(b.downsample[0].weight, "w"), (b.downsample[1], "bn")]
        return e

    @staticmethod
    def _head_entries(h):
        e = []
        aspp = h.aspp
        for seq in aspp.convs:
            conv, bn = (seq[1], seq[2]) if type(seq).__name__ == "ASPPPooling" else (seq[0], seq[1])
            e += [(conv.weight, "w"), (bn, "bn")]
        e += [(aspp.project[0].weight, "w"), (aspp.project[1], "bn"), (h.conv3.weight, "w"), (h.bn3, "bn"),
              (h.cls.weight, "cls_w"), (h.cls.bias, "cls_b")]
        if h.pre is not None:          # my_branch(bottleneck=...): finishes last in backward
            e += [(h.pre.weight, "w"), (h.pre.bias, "vec")]
        return e


def pack_all(net, dtype):
    """Pack every conv weight of `net` for the compute dtype in ONE launch (persistent packed
    buffers, static device descriptor table) and mark the per-module caches current."""
    import numpy as np
    cfg = net.cfg
    convs = []
    for mod in net.modules():
        if type(mod).__name__ == "Conv2d" and mod.weight.shape[1] != 3:        # the stem goes through packed_stem
            is_cls = mod.bias is not None and mod.__dict__.get("_eeseg_role") != "pre"
            convs.append((mod, torch.float32 if is_cls else dtype, CPAD if is_cls else None))
    groups = {}
    for mod, dt, cp in convs:
        groups.setdefault(dt, []).append((mod, cp))
    st = cfg.__dict__.setdefault("_pack_tables", {})
    for dt, items in groups.items():
        key = (dt, tuple(m.weight.data_ptr() for m, _ in items))
        tab = st.get(dt)
        if tab is None or tab[0] != key:
            rec = np.zeros((len(items), 6), dtype=np.int64)
            bufs = []
            for i, (m, cp) in enumerate(items):
                w = m.weight
                co, ci, r, s_ = w.shape
                cpad = cp or co
                krsc = 1 if (w.is_contiguous(memory_format=torch.channels_last) or (r == 1 and s_ == 1)) else 0
                wf = torch.empty((cpad, r, s_, ci), dtype=dt, device=w.device)
                wb = torch.empty((ci, r, s_, cpad), dtype=dt, device=w.device)
                bufs.append((wf, wb, cp))
                rec[i, 0], rec[i, 1], rec[i, 2] = w.data_ptr(), wf.data_ptr(), wb.data_ptr()
                rec[i, 3] = co | (cpad << 32)
                rec[i, 4] = ci | ((r * s_) << 32)
                rec[i, 5] = krsc
            tab = (key, torch.from_numpy(rec).to(items[0][0].weight.device), bufs)
            st[dt] = tab
        K.pack_weight_multi(tab[1], len(items), dt)
        for (m, cp), (wf, wb, _) in zip(items, tab[2]):
            w = m.weight
            m.__dict__["_eeseg_pack"] = ((w._version, _WEIGHTS_EPOCH[0], dt, w.data_ptr(), cp), wf, wb)


def packed(conv, dtype, cout_pad=None):
    """(w_fwd KRSC, w_bwd CRSK) of a conv module in `dtype`, cached per weight version."""
    w = conv.weight
    key = (w._version, _WEIGHTS_EPOCH[0], dtype, w.data_ptr(), cout_pad)
    cache = conv.__dict__.get("_eeseg_pack")
    if cache is None or cache[0] != key:
        wf, wb = K.pack_weight(w.detach(), dtype, cout_pad)
        cache = (key, wf, wb)
        conv.__dict__["_eeseg_pack"] = cache
    return cache[1], cache[2]


def _geom(conv):
    return conv.stride[0], conv.padding[0], conv.dilation[0]


class AllGatherRows(torch.autograd.Function):
    """[n, ...] on every rank -> [world*n, ...] (rank-major) on every rank.  For losses that every rank evaluates
    on the WHOLE batch (exact Lovasz): each rank then holds the full gradient, so backward just takes this rank's
    rows, times world because the data-parallel reducer averages the parameter gradients over the ranks."""

    @staticmethod
    def forward(ctx, t, cfg):
        # equal shards on every rank (parallel.ShardSampler drops the ragged last batch): the collective needs equal
        # sizes, and backward's "x world" is exact only under the AVERAGING gradient reducer
        ctx.cfg, ctx.n = cfg, t.shape[0]
        return cfg.all_gather(t.contiguous()).flatten(0, 1)

    @staticmethod
    def backward(ctx, g):
        cfg, n = ctx.cfg, ctx.n
        r = cfg.dp_rank()
        return g[r * n:(r + 1) * n] * float(cfg.dp_world()), None


def _allreduce(cfg, t):
    if cfg.sync_active():
        cfg.all_reduce(t)
    return t


def sync_bn_sums(cfg, sums, count):
    """SyncBN forward: `sums` [2,C] = this rank's (sum x, sum x^2) over `count` samples per channel ->
    the group's sums (in place) and the group's sample count.  Every rank holds the same number of samples
    (equal shards), so the count is count * world."""
    cfg.all_reduce(sums)
    return sums, count * cfg.world()


def global_mean_normaliser(comm, cnt):
    """Per-exit valid-pixel counts `cnt` [E] of this rank's shard -> the divisor that makes the AVERAGE over
    ranks of (rank loss sum / divisor) equal to the whole batch's mean loss: (sum of the counts over the group)
    / world.  Gradients are averaged over ranks the same way (parallel.ArenaReducer)."""
    tot = cnt.clone()
    comm.all_reduce(tot)
    return tot / comm.dp_world()


# ------------------------------------------------------- conv + BN (+ReLU) ----
def conv_bn_fwd(cfg, x, conv, bn, relu, residual=None, out=None, x_is_col=False, frozen=False):
    """Train-mode conv -> BN(batch stats) -> (+residual) -> ReLU.  Returns (y, state).
    `frozen`: differentiable EVAL-mode BatchNorm (running statistics, not updated): what torch does when a
    network in .eval() is back-propagated (fine-tuning with frozen statistics)."""
    if x_is_col:      # stem GEMM: x is the im2col matrix, weight is the padded [Cout,1,1,Kpad] matrix
        wf = packed_stem(conv, x.dtype, x.shape[-1])
        c, part = K.conv_fwd(x, wf, want_stats=not frozen)
    else:
        wf, _ = packed(conv, x.dtype)
        s, p, d = _geom(conv)
        c, part = K.conv_fwd(x, wf, s, p, d, want_stats=not frozen)
    count = c.numel() // c.shape[-1]
    mom = bn.momentum if bn.momentum is not None else 0.1
    if frozen:
        ss = K.bn_eval_scale_shift(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        mi = torch.stack([bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)])
        if residual is not None and relu:
            y, mask = K.bn_apply(c, ss, residual=residual, relu=True, out=out, want_mask=True)
            return y, (x, c, mask, mi, count, relu, ss, True)
        return K.bn_apply(c, ss, residual=residual, relu=relu, out=out), (x, c, None, mi, count, relu, ss, True)
    if cfg.sync_active():
        # SyncBN: partial sums -> ONE collective -> finalize INSIDE the apply pass (eeseg_bn_finalize_apply): no separate
        # finalize launch between the collective and the pass that waits for it
        sums, count = sync_bn_sums(cfg, K.reduce_partials(part), count)
        bn._pending_batches += 1
        want_mask = residual is not None and relu
        y, mask, mi, ss = K.bn_finalize_apply(c, sums, count, bn.weight, bn.bias, bn.eps, mom, bn.running_mean, bn.running_var,
                                              residual=residual, relu=relu, out=out, want_mask=want_mask)
        return y, (x, c, mask, mi, count, relu, ss)
    elif K.bn_fwd_fused_ok(c, part):
        # small tensors (per-GPU shards): partial reduction, finalize and apply in ONE launch - every block reduces the
        # partial sums of its own channels itself (eeseg_bn_fwd_fused; bit-identical with the two launches below)
        bn._pending_batches += 1
        y, mask, mi, ss = K.bn_fwd_fused(c, part, count, bn.weight, bn.bias, bn.eps, mom, bn.running_mean, bn.running_var,
                                         residual=residual, relu=relu, out=out, want_mask=residual is not None and relu)
        return y, (x, c, mask, mi, count, relu, ss)
    elif part.shape[0] > 2048:      # very many tiles (stem): two-level reduction, then finalize
        mi, ss = K.bn_finalize(K.reduce_partials(part), count, bn.weight, bn.bias, bn.eps, mom, bn.running_mean,
                               bn.running_var)
    else:       # local BatchNorm: partial reduction and finalize fused in one launch
        mi, ss = K.bn_reduce_finalize(part, count, bn.weight, bn.bias, bn.eps, mom, bn.running_mean, bn.running_var)
    bn._pending_batches += 1
    # backward recomputes the ReLU mask from c*scale+shift when there is no residual input (saves streaming y
    # again); with a residual the forward leaves a byte mask (1 bit per element) as the mask source
    # ... and, round 4, for every ReLU layer whose backward will be the two-pass form (tensors too large for the one-launch
    # BatchNorm backward): reading 1 bit per element beats recomputing the sign in both passes (B=32: -0.5 ms per step)
    if relu and (residual is not None or cfg.bn_mask_all == 1 or (cfg.bn_mask_all == 2 and c.is_cuda and not K.bn_bwd_coop_ok(c))):
        y, mask = K.bn_apply(c, ss, residual=residual, relu=True, out=out, want_mask=True)
        return y, (x, c, mask, mi, count, relu, ss)
    y = K.bn_apply(c, ss, residual=residual, relu=relu, out=out)
    return y, (x, c, None, mi, count, relu, ss)


def conv_bn_fwd_group(cfg, items, frozen=False):
    """conv_bn_fwd for several INDEPENDENT layers: items = [(x, conv, bn, relu, out)] -> [(y, state)].  Under SyncBN the
    layers' (sum x, sum x^2) pairs travel in ONE all-reduce instead of one each ("coalesce where the dependency
    graph allows - the 5 ASPP branches share one call"): all convs first, their partial sums reduced into slices of one
    flat buffer, one collective, then finalize + apply per layer.  Without SyncBN it is the plain loop."""
    if frozen or not cfg.sync_active() or len(items) < 2:
        return [conv_bn_fwd(cfg, x, conv, bn, relu, out=out, frozen=frozen) for x, conv, bn, relu, out in items]
    widths = [conv.weight.shape[0] for _, conv, _, _, _ in items]
    flat = torch.empty(2 * sum(widths), dtype=torch.float32, device=items[0][0].device)
    pend, off = [], 0
    for (x, conv, bn, relu, out), cw in zip(items, widths):
        wf, _ = packed(conv, x.dtype)
        s, p, d = _geom(conv)
        c, part = K.conv_fwd(x, wf, s, p, d, want_stats=True)
        sums = flat[off:off + 2 * cw].view(2, cw)
        K.reduce_partials(part, out=sums)
        pend.append((c, sums, c.numel() // c.shape[-1]))
        off += 2 * cw
    cfg.all_reduce(flat)
    res = []
    for (x, conv, bn, relu, out), (c, sums, count) in zip(items, pend):
        mom = bn.momentum if bn.momentum is not None else 0.1
        bn._pending_batches += 1
        y, _, mi, ss = K.bn_finalize_apply(c, sums, count * cfg.world(), bn.weight, bn.bias, bn.eps, mom, bn.running_mean,
                                           bn.running_var, relu=relu, out=out)
        res.append((y, (x, c, None, mi, count * cfg.world(), relu, ss)))
    return res


def bn_bwd_local_sums(cfg, st, dy, bn, copy):
    """First half of conv_bn_bwd for a train-mode layer under SyncBN: this rank's (sum g, sum g x_hat) - written into the
    gradient arena as (dbeta, dgamma) like conv_bn_bwd does - and, by the same launch, into `copy` ([2,C] slice of the buffer
    the collective reduces)."""
    x, c, y, mi, count, relu, ss = st[:7]
    pair = cfg.gview(bn)
    if pair is not None and cfg.accumulate:
        sums = K.bn_bwd_reduce(dy, y if relu else None, c, mi, relu, scale_shift=ss, copy=copy)
        pair.add_(sums)
    else:
        sums = K.bn_bwd_reduce(dy, y if relu else None, c, mi, relu, out=pair, scale_shift=ss, copy=copy)
    return sums


def conv_bn_bwd(cfg, st, dy, conv, bn, need_dx=True, dx_accum=None, want_dres=False, x_is_col=False, dx_add=None,
                dc_out=None, synced_sums=None, local_sums=None):
    """Backward of conv_bn_fwd.  Returns (dx, dres, dW(param layout view), dgamma, dbeta); the three
    parameter gradients are None in arena mode (written in place).  dx_add=(t, mask): the data-gradient adds
    t * mask (bit mask) in its epilogue - the masked block gradient of a bottleneck, never materialised."""
    x, c, y, mi, count, relu, ss = st[:7]
    frozen = len(st) > 7 and st[7]
    pair = cfg.gview(bn)
    dc = None
    if (synced_sums is None and not frozen and not cfg.sync_active() and not (pair is not None and cfg.accumulate) and
            K.bn_bwd_coop_ok(c)):
        # local BatchNorm on a tensor that fits the chip's registers (the 4-8 image shards of a data-parallel run): reduce,
        # grid barrier and apply in ONE launch, dy and the conv output read once (eeseg_bn_bwd_coop)
        dc, dres, sums = K.bn_bwd_coop(dy, y if relu else None, c, mi, bn.weight, count, relu, out=pair,
                                       want_dres=want_dres, dx=dc_out, scale_shift=ss)
    elif synced_sums is not None:
        # the caller reduced this layer's sums together with its siblings' (head_bwd: one collective for the ASPP branches):
        # `local_sums` = this rank's (already in the arena), `synced_sums` = the group's
        sums = local_sums
    elif pair is not None and cfg.accumulate:
        sums = K.bn_bwd_reduce(dy, y if relu else None, c, mi, relu, scale_shift=ss)
        pair.add_(sums)
    elif pair is not None and cfg.sync_active() and not frozen:
        # SyncBN + arena: the reduction writes the local sums (dbeta, dgamma) into the arena AND a second copy for the collective
        sync_copy = torch.empty_like(pair)
        K.bn_bwd_reduce(dy, y if relu else None, c, mi, relu, out=pair, scale_shift=ss, copy=sync_copy)
        sums = pair
    else:
        sums = K.bn_bwd_reduce(dy, y if relu else None, c, mi, relu, out=pair, scale_shift=ss)
    dbeta, dgamma = (None, None) if pair is not None else (sums[0], sums[1])
    if dc is not None:
        pass                                 # the one-launch form has applied the sums already
    elif synced_sums is not None:
        if pair is None:
            dbeta, dgamma = dbeta.clone(), dgamma.clone()
        sums = synced_sums
    elif frozen:
        # statistics are constants: dc = g * gamma * invstd, i.e. the train-mode formula without its two mean terms
        sums = torch.zeros_like(sums)
    elif cfg.sync_active():
        if pair is None:
            dbeta, dgamma = dbeta.clone(), dgamma.clone()  # parameter grads stay local (DP averages them)
        if pair is not None and not cfg.accumulate:
            sums = sync_copy                 # (accumulate mode: `sums` is a private tensor already)
        if cfg.defer_wgrad:
            work = cfg.all_reduce_begin(sums)
            cfg.run_deferred()               # the layer above's weight gradient fills the collective's latency
            cfg.all_reduce_end(work)
        else:
            _allreduce(cfg, sums)
    if dc is None:
        dc, dres = K.bn_bwd_apply(dy, y if relu else None, c, mi, bn.weight, sums, count, relu, want_dres=want_dres,
                                  scale_shift=ss, dx=dc_out)    # dc_out: the caller's buffer for the conv-output gradient
    gv = cfg.gview(conv.weight)
    if x_is_col:
        dw = K.conv_wgrad(x, dc, 1, 1)
        cout, cin, r, s_ = conv.weight.shape
        dwk = dw.view(cout, -1)[:, :r * s_ * cin]
        if gv is not None:
            g2 = gv.view(cout, r * s_ * cin)
            if cfg.accumulate:
                g2.add_(dwk)
            else:
                g2.copy_(dwk)
            return None, dres, None, dgamma, dbeta
        return None, dres, dwk.reshape(cout, r, s_, cin).permute(0, 3, 1, 2), dgamma, dbeta
    s, p, d = _geom(conv)
    R, S = conv.weight.shape[2], conv.weight.shape[3]

    def wgrad():
        if gv is not None:
            K.conv_wgrad(x, dc, R, S, s, p, d, out=gv, accumulate=cfg.accumulate or cfg.arena.prezeroed)
            return None
        return K.conv_wgrad(x, dc, R, S, s, p, d).permute(0, 3, 1, 2)

    dx = None
    if need_dx and cfg.overlap_wgrad == 2 and dc.is_cuda:
        # data-gradient first on the main stream, THEN the weight-gradient on the side stream: it runs beside the
        # BatchNorm-backward passes of the layer below (HBM-bound, few registers, small LDS: their blocks fit on a CU next
        # to a one-block-per-CU MFMA kernel) instead of competing with the data-gradient for the CUs
        _, wb = packed(conv, dc.dtype)
        dx = K.conv_dgrad(dc, wb, (x.shape[1], x.shape[2]), s, p, d, accumulate_into=dx_accum, add=dx_add)
        cur = torch.cuda.current_stream(dc.device)
        side = cfg.side_stream(dc.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            dwp = wgrad()
        cfg._side_busy = True
        cfg._side_keep.append((x, dc, dwp))
    elif need_dx and cfg.overlap_wgrad and dc.is_cuda:
        # fork: wgrad goes to the side stream and runs concurrently with the data-gradient and the
        # following BatchNorm-backward kernels of this unit; Config.unit_done() joins.  The tensors it
        # reads are kept alive until then so the allocator cannot hand their memory out early.
        cur = torch.cuda.current_stream(dc.device)
        side = cfg.side_stream(dc.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            dwp = wgrad()
        cfg._side_busy = True
        cfg._side_keep.append((x, dc, dwp))
        _, wb = packed(conv, dc.dtype)
        dx = K.conv_dgrad(dc, wb, (x.shape[1], x.shape[2]), s, p, d, accumulate_into=dx_accum, add=dx_add)
    elif gv is not None and cfg.defer_wgrad and cfg.sync_active() and not frozen:
        # arena mode under SyncBN: data-gradient now, weight gradient (written in place, nothing to return) when the
        # next layer's all-reduce is in flight - at the latest when the unit ends (Config.unit_done)
        cfg.run_deferred()
        if need_dx:
            _, wb = packed(conv, dc.dtype)
            dx = K.conv_dgrad(dc, wb, (x.shape[1], x.shape[2]), s, p, d, accumulate_into=dx_accum, add=dx_add)
        cfg._deferred = wgrad                # the closure keeps x and dc alive
        dwp = None
    else:
        if gv is not None and cfg.group_wgrad and dc.is_cuda:
            cfg.queue_wgrad((x, dc, R, S, s, p, d, gv, cfg.accumulate or cfg.arena.prezeroed))     # issued with the unit's others
            dwp = None
        else:
            dwp = wgrad()
        if need_dx:
            _, wb = packed(conv, dc.dtype)
            dx = K.conv_dgrad(dc, wb, (x.shape[1], x.shape[2]), s, p, d, accumulate_into=dx_accum, add=dx_add)
    return dx, dres, dwp, dgamma, dbeta


def conv_bn_eval(cfg, x, conv, bn, relu, residual=None, out=None, x_is_col=False):
    """Eval-mode conv with BN folded into the conv epilogue (one kernel)."""
    ss = K.bn_eval_scale_shift(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    if x_is_col:
        wf = packed_stem(conv, x.dtype, x.shape[-1])
        y, _ = K.conv_fwd(x, wf, scale=ss[0], shift=ss[1], residual=residual, relu=relu, out=out)
    else:
        wf, _ = packed(conv, x.dtype)
        s, p, d = _geom(conv)
        y, _ = K.conv_fwd(x, wf, s, p, d, scale=ss[0], shift=ss[1], residual=residual, relu=relu, out=out)
    return y


def packed_stem(conv, dtype, kpad):
    w = conv.weight
    key = (w._version, _WEIGHTS_EPOCH[0], dtype, w.data_ptr(), kpad, "stem")
    cache = conv.__dict__.get("_eeseg_pack_stem")
    if cache is None or cache[0] != key:
        cout, cin, r, s = w.shape
        wk = w.detach().permute(0, 2, 3, 1).reshape(cout, r * s * cin)    # KRSC rows (view when channels_last)
        if not wk.is_contiguous():
            wk = wk.contiguous()
        wm = K.pack_matrix(wk, cout, kpad, dtype).view(cout, 1, 1, kpad)
        cache = (key, wm)
        conv.__dict__["_eeseg_pack_stem"] = cache
    return cache[1]


STEM_KPAD = 192     # 7*7*3 = 147 -> multiple of 64


# ------------------------------------------------------------------- stem ----
def stem_fwd(cfg, img, conv, bn, train, frozen=False):
    """img [N,3,H,W] fp32 NCHW -> [N,H/4,W/4,64] NHWC in compute dtype."""
    if not img.is_contiguous():
        img = img.contiguous()
    if img.dtype != torch.float32:
        img = img.float()
    r = conv.weight.shape[2]
    col = K.im2col_nchw(img, r, r, conv.stride[0], conv.padding[0], STEM_KPAD, cfg.compute_dtype)
    if train or frozen:
        y, st = conv_bn_fwd(cfg, col, conv, bn, True, x_is_col=True, frozen=frozen)
    else:
        y, st = conv_bn_eval(cfg, col, conv, bn, True, x_is_col=True), None
    p = K.maxpool3x3s2(y)
    return p, (st, y, p)        # p is layer1's saved input anyway: keeping the reference costs nothing


def stem_bwd(cfg, state, dp, conv, bn):
    st, y, p = state
    dy = K.maxpool3x3s2_bwd(y, dp, p)
    _, _, dw, dg, db = conv_bn_bwd(cfg, st, dy, conv, bn, need_dx=False, x_is_col=True)
    return dw, dg, db


# ------------------------------------------------------------- bottleneck ----
def bottleneck_fwd(cfg, x, blk, train, frozen=False):
    ds = blk.downsample
    if not train and not frozen:
        y1 = conv_bn_eval(cfg, x, blk.conv1, blk.bn1, True)
        y2 = conv_bn_eval(cfg, y1, blk.conv2, blk.bn2, True)
        idn = conv_bn_eval(cfg, x, ds[0], ds[1], False) if ds is not None else x
        return conv_bn_eval(cfg, y2, blk.conv3, blk.bn3, True, residual=idn), None
    y1, s1 = conv_bn_fwd(cfg, x, blk.conv1, blk.bn1, True, frozen=frozen)
    y2, s2 = conv_bn_fwd(cfg, y1, blk.conv2, blk.bn2, True, frozen=frozen)
    sd = None
    if ds is not None:
        idn, sd = conv_bn_fwd(cfg, x, ds[0], ds[1], False, frozen=frozen)
    else:
        idn = x
    out, s3 = conv_bn_fwd(cfg, y2, blk.conv3, blk.bn3, True, residual=idn, frozen=frozen)
    return out, (s1, s2, s3, sd)


def bottleneck_bwd(cfg, state, dout, blk):
    """Returns (dx, [grads in blk.param_list() order])."""
    s1, s2, s3, sd = state
    # identity blocks: the block gradient that flows past the three convs is dout * (ReLU mask of the block output).
    # The data-gradient of conv1 adds it in its epilogue from dout and the 1-bit mask bn_apply left, so BatchNorm
    # backward writes one tensor less (bf16 layers on the 256-tile / pointwise kernels; cfg.fuse_block_residual)
    mask3 = s3[2]
    fuse = (sd is None and cfg.fuse_block_residual and mask3 is not None and mask3.dtype == torch.uint8 and
            dout.is_cuda and K.masked_residual_ok(dout.dtype, dout.shape[-1]) and blk.conv1.stride[0] == 1)
    dy2, dres, dw3, dg3, db3 = conv_bn_bwd(cfg, s3, dout, blk.conv3, blk.bn3, want_dres=not fuse)
    dy1, _, dw2, dg2, db2 = conv_bn_bwd(cfg, s2, dy2, blk.conv2, blk.bn2)
    grads = [None] * 9
    if sd is not None:
        ds = blk.downsample
        dxd, _, dwd, dgd, dbd = conv_bn_bwd(cfg, sd, dres, ds[0], ds[1])
        dx, _, dw1, dg1, db1 = conv_bn_bwd(cfg, s1, dy1, blk.conv1, blk.bn1, dx_accum=dxd)
        extra = [dwd, dgd, dbd]
    elif fuse:
        dx, _, dw1, dg1, db1 = conv_bn_bwd(cfg, s1, dy1, blk.conv1, blk.bn1, dx_add=(dout, mask3))
        extra = []
    else:
        dx, _, dw1, dg1, db1 = conv_bn_bwd(cfg, s1, dy1, blk.conv1, blk.bn1, dx_accum=dres)
        extra = []
    grads = [dw1, dg1, db1, dw2, dg2, db2, dw3, dg3, db3] + extra
    cfg.unit_done(blk)
    return dx, grads


# ------------------------------------------------------------------- head ----
def conv_bias_fwd(cfg, x, conv):
    """1x1 conv + bias, no BN / activation (my_branch's bottleneck conv, from_deepv3_new.py:24)."""
    wf, _ = packed(conv, x.dtype)
    s, p, d = _geom(conv)
    return K.conv_fwd(x, wf, s, p, d, shift=conv.bias.detach())[0]


def conv_bias_bwd(cfg, x, dy, conv, dx_accum=None):
    """-> (dx, dw, dbias); dw/dbias are None when they were written into the gradient arena."""
    gvw, gvb = cfg.gview(conv.weight), cfg.gview(conv.bias)
    r = conv.weight.shape[2]
    s, p, d = _geom(conv)
    if gvw is not None:
        if cfg.accumulate:
            gvb.add_(K.colsum(dy))
        else:
            K.colsum(dy, out=gvb)
        K.conv_wgrad(x, dy, r, r, s, p, d, out=gvw, accumulate=cfg.accumulate or cfg.arena.prezeroed)
        dw = db = None
    else:
        db = K.colsum(dy)
        dw = K.conv_wgrad(x, dy, r, r, s, p, d).permute(0, 3, 1, 2)
    _, wb = packed(conv, x.dtype)
    dx = K.conv_dgrad(dy, wb, x.shape[1:3], s, p, d, accumulate_into=dx_accum)
    return dx, dw, db


def head_fwd(cfg, x, head, train, frozen=False):
    """DeepLabHead on NHWC features -> low-res logits [N,h,w,CPAD] fp32.  `frozen`: differentiable eval mode
    (running statistics, no dropout), state saved for head_bwd."""
    keep = train or frozen
    aspp = head.aspp
    x_in = x
    if head.pre is not None:
        x = conv_bias_fwd(cfg, x_in, head.pre)
    N, h, w, cin = x.shape
    nb = len(aspp.convs)                      # 1x1 + atrous convs + pooling
    mid = aspp.project[0].weight.shape[0]
    cat = torch.empty((N, h, w, nb * mid), dtype=x.dtype, device=x.device)
    states = []
    # image pooling branch: GAP -> 1x1 conv -> BN -> ReLU -> broadcast
    pool = aspp.convs[nb - 1]
    g = K.sum_hw(x, 1.0 / (h * w)).view(N, 1, 1, cin)
    if keep:
        # the five branches are independent: under SyncBN their statistics share ONE collective (conv_bn_fwd_group)
        items = [(x, aspp.convs[i][0], aspp.convs[i][1], True, cat[..., i * mid:(i + 1) * mid]) for i in range(nb - 1)]
        items.append((g, pool[1], pool[2], True, None))
        res = conv_bn_fwd_group(cfg, items, frozen=frozen)
        states = [st for _, st in res]
        pv = res[-1][0]
    else:
        for i in range(nb - 1):
            seq = aspp.convs[i]
            conv_bn_eval(cfg, x, seq[0], seq[1], True, out=cat[..., i * mid:(i + 1) * mid])
        pv = conv_bn_eval(cfg, g, pool[1], pool[2], True)
    K.broadcast_hw(pv.view(N, mid), cat[..., (nb - 1) * mid:])
    proj = aspp.project
    seed = None
    if keep:
        pr, stj = conv_bn_fwd(cfg, cat, proj[0], proj[1], True, frozen=frozen)
        pdrop = 0.0 if frozen else proj[3].p
        if pdrop > 0:
            seed = cfg.next_seed()
            # data parallel: rank r draws the r-th slice of the whole batch's mask (eeseg_dropout index_offset)
            pr_d = K.dropout(pr, pdrop, seed, cfg.step_dev(pr.device), cfg.dp_rank() * pr.numel() if cfg.dp_active() else 0)
        else:
            pr_d = pr
        q, stq = conv_bn_fwd(cfg, pr_d, head.conv3, head.bn3, True, frozen=frozen)
    else:
        pr_d = conv_bn_eval(cfg, cat, proj[0], proj[1], True)
        q = conv_bn_eval(cfg, pr_d, head.conv3, head.bn3, True)
        stj = stq = None
    # classifier 1x1 (+bias) always in fp32, Cout padded to CPAD
    cls = head.cls
    q32 = q if q.dtype == torch.float32 else K.cast(q, torch.float32)
    wf, _ = packed(cls, torch.float32, CPAD)
    bias = cls.__dict__.get("_eeseg_bias_pad")
    bkey = (cls.bias._version, _WEIGHTS_EPOCH[0], cls.bias.data_ptr())
    if bias is None or bias[0] != bkey:
        bp = torch.zeros(CPAD, dtype=torch.float32, device=x.device)
        bp[:cls.bias.numel()] = cls.bias.detach()
        bias = (bkey, bp)
        cls.__dict__["_eeseg_bias_pad"] = bias
    logits, _ = K.conv_fwd(q32, wf, shift=bias[1])
    state = (x, cat, states, stj, seed, stq, q32, x_in) if keep else None
    return logits, state


def head_bwd(cfg, state, dlogits, head, dx_init=None):
    """Returns (dx, grads in head.param_list() order).  `dx_init`: a gradient of the head's input that already
    exists (the next backbone section's, nn_modules._HeadForkFn); the head's gradient is added into it in place."""
    x, cat, states, stj, seed, stq, q32, x_in = state
    aspp = head.aspp
    nb = len(aspp.convs)
    mid = aspp.project[0].weight.shape[0]
    N, h, w, cin = x.shape
    cls = head.cls
    ncls = cls.weight.shape[0]
    # classifier
    gvb, gvw = cfg.gview(cls.bias), cfg.gview(cls.weight)
    if gvw is not None:
        if cfg.accumulate:
            gvb.add_(K.colsum(dlogits))
        else:
            K.colsum(dlogits, out=gvb)
        K.conv_wgrad(q32, dlogits, 1, 1, out=gvw, accumulate=cfg.accumulate or cfg.arena.prezeroed)
        dbias = dwc = None
    else:
        dbias = K.colsum(dlogits)[:ncls]
        dwc = K.conv_wgrad(q32, dlogits, 1, 1)[:ncls].permute(0, 3, 1, 2)
    _, wb = packed(cls, torch.float32, CPAD)
    dq = K.conv_dgrad(dlogits, wb, (h, w))
    if cfg.compute_dtype != torch.float32:
        dq = K.cast(dq, cfg.compute_dtype)
    # 3x3 conv + BN + ReLU
    dpr_d, _, dw3, dg3, db3 = conv_bn_bwd(cfg, stq, dq, head.conv3, head.bn3)
    proj = aspp.project
    dpr = K.dropout(dpr_d, proj[3].p, seed, cfg.step_dev(dpr_d.device),
                    cfg.dp_rank() * dpr_d.numel() if cfg.dp_active() else 0) if seed is not None else dpr_d
    dcat, _, dwj, dgj, dbj = conv_bn_bwd(cfg, stj, dpr, proj[0], proj[1])
    grads_convs = []
    dx = dx_init if head.pre is None else None
    convs = [aspp.convs[i][0] for i in range(nb - 1)]
    ntaps = sum(cv.kernel_size[0] * cv.kernel_size[1] for cv in convs)
    merge = (cfg.merge_aspp_dgrad and dcat.is_cuda and all(cv.stride[0] == 1 for cv in convs) and
             K.multi_dgrad_ok(dcat.dtype, cin, mid, ntaps))
    # SyncBN: the five branches' backward sums share ONE collective (as their forward statistics do): local sums of every
    # branch first, one all-reduce of the concatenation, then each branch continues with its slice
    pool = aspp.convs[nb - 1]
    dpv = K.sum_hw(dcat[..., (nb - 1) * mid:]).view(N, 1, 1, mid)
    synced = [None] * nb
    local = [None] * nb
    if cfg.sync_active() and not (len(states[0]) > 7 and states[0][7]):
        bns = [aspp.convs[i][1] for i in range(nb - 1)] + [pool[2]]
        dys = [dcat[..., i * mid:(i + 1) * mid] for i in range(nb - 1)] + [dpv]
        flat = torch.empty((nb, 2, mid), dtype=torch.float32, device=dcat.device)
        for i in range(nb):
            local[i] = bn_bwd_local_sums(cfg, states[i], dys[i], bns[i], flat[i])
        cfg.all_reduce(flat)
        synced = [flat[i] for i in range(nb)]
    if merge:
        # ONE data-gradient launch for the 1x1 + atrous branches (generalised taps): dx is written once instead of being
        # read-modify-written by every branch, and a 256 x 256 output tile pays one epilogue for all 28 taps
        dcc = torch.empty((nb - 1, N, h, w, mid), dtype=dcat.dtype, device=dcat.device)
        for i in range(nb - 1):
            seq = aspp.convs[i]
            sl = dcat[..., i * mid:(i + 1) * mid]
            _, _, dwi, dgi, dbi = conv_bn_bwd(cfg, states[i], sl, seq[0], seq[1], need_dx=False, dc_out=dcc[i],
                                              synced_sums=synced[i], local_sums=local[i])
            grads_convs += [dwi, dgi, dbi]
        wcat = K.concat_tap_weights([packed(cv, dcat.dtype)[1] for cv in convs])
        dx = K.conv_dgrad_multi(dcc, wcat, [(cv.kernel_size[0], cv.padding[0], cv.dilation[0]) for cv in convs],
                                accumulate_into=dx)
    else:
        for i in range(nb - 1):
            seq = aspp.convs[i]
            sl = dcat[..., i * mid:(i + 1) * mid]
            dx, _, dwi, dgi, dbi = conv_bn_bwd(cfg, states[i], sl, seq[0], seq[1], dx_accum=dx,
                                               synced_sums=synced[i], local_sums=local[i])
            grads_convs += [dwi, dgi, dbi]
    dg_, _, dwp, dgp, dbp = conv_bn_bwd(cfg, states[nb - 1], dpv, pool[1], pool[2], synced_sums=synced[nb - 1],
                                        local_sums=local[nb - 1])
    K.broadcast_hw(dg_.view(N, cin), dx, scale=1.0 / (h * w), accumulate=True)
    grads_convs += [dwp, dgp, dbp]
    grads = grads_convs + [dwj, dgj, dbj, dw3, dg3, db3, dwc, dbias]
    if head.pre is not None:
        dx, dwpre, dbpre = conv_bias_bwd(cfg, x_in, dx, head.pre, dx_accum=dx_init)
        grads += [dwpre, dbpre]
    cfg.unit_done(head)
    return dx, grads
