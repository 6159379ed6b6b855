"""Data-parallel collectives of the HIP path: RCCL through libeeseg's C ABI (``eeseg_comm_*``), enqueued on streams
this package owns.

The reference has no parallelism (``nn.DataParallel`` commented out at train_funcs.py:72-74); the exchange
steps here are the build's own: gradient buckets, SyncBN statistics, the global CE valid-pixel count, the
exact-Lovasz all-gather and the per-exit mIoU counters.

``torch.distributed`` is used for the RENDEZVOUS only (the 128-byte RCCL id travels over the process group the caller
initialised - gloo is enough) and for host-side bookkeeping.  The data path never creates a c10d ``Work``: c10d records
each collective's end event on its own internal stream and polls it from a watchdog thread, and on HIP an event is
"captured" as soon as the stream it was last recorded on is capturing - so a finished warm-up collective aborts the
process when that internal stream is forked into the HIP-graph capture of the training step (round 2's
hipErrorCapturedEvent; ``scripts/captured_event_repro.hip``).  Here a collective is ONE RCCL
kernel on the stream given, eager or captured alike, and the fork / join edges are this package's own events.

Two communicators per process group (``DataParallelComm``):
  * ``grad``  - the large gradient buckets, on the lane ``lane_g`` (overlaps the rest of backward);
  * ``stat``  - the small latency-bound collectives ([2,C] SyncBN pairs, [E] counts, logits all-gather), on the
                compute stream or on ``lane_s`` (deferred SyncBN backward).
A single communicator would serialise a [2,C] reduce behind a 32-MiB bucket.

Two lanes (default) vs ONE lane (``EESEG_DP_SINGLE_LANE=1`` / ``bench.py --dp-lanes 1``).  With two lanes the kernels of the two
communicators are enqueued in the same program order on every rank, but which of two READY RCCL kernels the GPU starts first
is the hardware scheduler's choice, per rank; that is harmless as long as both kernels fit on the chip together (RCCL's
workgroups are small; the one-block-per-CU conv kernels they may have to wait for never wait for RCCL), and it is what the
overlap is built on - but no run with more than one rank has executed it yet (no multi-GPU node was available to this build).
The single-lane mode is the conservative fallback for a first multi-GPU run that misbehaves: EVERY collective of both
communicators - buckets, SyncBN forward and backward, counts, gathers - is issued on ONE lane stream in program order (fork
from the compute stream, the collective, join), so the execution order of all collectives is the same on every rank by stream
order; the price is that a [2,C] reduce queues behind a bucket that is still in flight.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from ._lib import EesegError, check, lib

_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 2, torch.int32: 3, torch.int64: 4}
SUM, AVG, MAX = 0, 1, 2


def rccl_version():
    v = C.c_int(0)
    check(lib().eeseg_comm_available(C.byref(v)), "comm_available")
    return v.value


_HOST_GROUPS = {}


def host_group(group=None):
    """A HOST-ONLY channel over the ranks of `group` for the rendezvous (the 128-byte RCCL id, the "can every rank bind
    librccl" agreement): `group` itself when its backend moves host tensors (gloo), else a gloo subgroup created once per
    group (collective: every rank of `group` calls this at the same point).  With an NCCL rendezvous group the id would
    otherwise travel through device tensors and c10d Works - exactly what the data path avoids."""
    if dist.get_world_size(group) == 1 or dist.get_backend(group) != "nccl":
        return group
    key = id(group) if group is not None else None
    if key not in _HOST_GROUPS:
        ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
        _HOST_GROUPS[key] = dist.new_group(ranks=ranks, backend="gloo")
    return _HOST_GROUPS[key]


class Communicator:
    """One RCCL communicator over the ranks of a torch.distributed group (rendezvous only)."""

    def __init__(self, group=None, device=None):
        if not dist.is_initialized():
            raise EesegError("Communicator needs torch.distributed for the rendezvous (any backend; gloo is enough)")
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.calls = 0                             # collectives enqueued so far (bench.py: collectives per step)
        ident = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            check(lib().eeseg_comm_unique_id(buf), "comm_unique_id")
            ident[0] = bytes(buf.raw)
        if self.world > 1:
            hg = host_group(group)                 # never through device tensors, whatever the caller's backend
            src = dist.get_global_rank(hg, 0) if hg is not None else 0
            dist.broadcast_object_list(ident, src=src, group=hg)
        handle = C.c_void_p(0)
        with torch.cuda.device(self.device):
            check(lib().eeseg_comm_create(ident[0], self.world, self.rank, C.byref(handle)), "comm_create")
        self._h = handle

    def _stream(self, stream):
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        return C.c_void_p(s.cuda_stream)

    def _ok(self, t):
        if not t.is_cuda or not t.is_contiguous():
            raise EesegError("collectives take contiguous device tensors")
        if self._h is None:
            raise EesegError("communicator already closed")

    def all_reduce(self, t, op=SUM, stream=None):
        """In-place reduction of `t` over the ranks, enqueued on `stream` (default: the current stream)."""
        self._ok(t)
        self.calls += 1
        check(lib().eeseg_comm_all_reduce(self._h, C.c_void_p(t.data_ptr()), t.numel(), _DTYPES[t.dtype], op,
                                          self._stream(stream)), "comm_all_reduce")
        return t

    def all_gather(self, t, stream=None):
        """-> [world, *t.shape] in rank order."""
        self._ok(t)
        self.calls += 1
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        check(lib().eeseg_comm_all_gather(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()),
                                          t.numel() * t.element_size(), self._stream(stream)), "comm_all_gather")
        return out

    def reduce_scatter(self, t, op=SUM, stream=None):
        """t [world, ...] on every rank -> this rank's slice [...] of the element-wise reduction over the ranks."""
        self._ok(t)
        assert t.shape[0] == self.world
        self.calls += 1
        out = torch.empty(t.shape[1:], dtype=t.dtype, device=t.device)
        check(lib().eeseg_comm_reduce_scatter(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), out.numel(),
                                              _DTYPES[t.dtype], op, self._stream(stream)), "comm_reduce_scatter")
        return out

    def broadcast(self, t, root=0, stream=None):
        self._ok(t)
        check(lib().eeseg_comm_broadcast(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), root,
                                         self._stream(stream)), "comm_broadcast")
        return t

    def check(self):
        check(lib().eeseg_comm_check(self._h), "comm_check")

    def close(self):
        """Collective.  The device must be idle and every HIP graph that holds this communicator's kernels gone."""
        h, self._h = self._h, None
        if h is not None:
            torch.cuda.synchronize(self.device)
            check(lib().eeseg_comm_destroy(h), "comm_destroy")


class Lane:
    """A stream of this package beside the compute stream, with explicit fork / join edges (plain events: inside a
    HIP-graph capture they become the graph's dependencies)."""

    PROFILE = None      # a list while bench.py measures: every join of a busy lane appends (lane name, event before, event after)
                        # recorded on the COMPUTE stream = how long the compute stream stood waiting for the lane (eager steps only)

    def __init__(self, device, name="lane"):
        self.stream = torch.cuda.Stream(device=device)
        self.device = device
        self.name = name
        self.busy = False

    def fork(self):
        """Work enqueued on the lane from now on starts after everything the compute stream holds so far."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self.busy = True

    def join(self):
        """The compute stream waits for everything on the lane."""
        if self.busy:
            cur = torch.cuda.current_stream(self.device)
            if Lane.PROFILE is not None and not torch.cuda.is_current_stream_capturing():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.stream)
                e1.record(cur)
                Lane.PROFILE.append((self.name, e0, e1))
            else:
                cur.wait_stream(self.stream)
            self.busy = False


class DataParallelComm:
    """The collectives a data-parallel replica of the network issues (module docstring).  ``engine.Config.comm``."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.stat = Communicator(group, device)
        self.grad = Communicator(group, device)
        self.world, self.rank, self.device = self.stat.world, self.stat.rank, self.stat.device
        # EESEG_DP_SINGLE_LANE=1: one lane for both communicators, every collective on it in program order (module docstring)
        self.single_lane = os.environ.get("EESEG_DP_SINGLE_LANE", "0") == "1"
        self.lane_g = Lane(self.device, "grad")
        self.lane_s = self.lane_g if self.single_lane else Lane(self.device, "stat")

    def stat_all_reduce(self, t):
        """A small collective the compute stream needs at once.  Two lanes: on the compute stream itself.  Single lane: on THE
        lane, fork -> collective -> join, behind whatever bucket is in flight there."""
        if self.single_lane:
            self.lane_g.fork()
            self.stat.all_reduce(t, stream=self.lane_g.stream)
            self.lane_g.busy = True
            self.lane_g.join()
        else:
            self.stat.all_reduce(t)
        return t

    def stat_reduce_scatter(self, t):
        if self.single_lane:
            self.lane_g.fork()
            out = self.stat.reduce_scatter(t, stream=self.lane_g.stream)
            self.lane_g.busy = True
            self.lane_g.join()
            return out
        return self.stat.reduce_scatter(t)

    def stat_all_gather(self, t):
        if self.single_lane:
            self.lane_g.fork()
            out = self.stat.all_gather(t, stream=self.lane_g.stream)
            self.lane_g.busy = True
            self.lane_g.join()
            return out
        return self.stat.all_gather(t)

    def close(self):
        self.lane_g.join()
        self.lane_s.join()
        self.grad.close()
        self.stat.close()


def forced():
    """EESEG_FORCE_ALLREDUCE=1: a 1-rank group still issues every collective (rehearsal on one GPU)."""
    return os.environ.get("EESEG_FORCE_ALLREDUCE") == "1"
