"""world_size-2 gloo tests (CPU) of the data-parallel pieces that do not need a GPU:
the bucketed gradient reducer, parameter broadcast and channels_last flat views."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ee_semantic_segmentation_amd.parallel import GradReducer, broadcast_parameters
        torch.manual_seed(100 + rank)                         # different init per rank on purpose
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                                  torch.nn.Conv2d(8, 4, 1), torch.nn.Flatten(), torch.nn.Linear(4 * 36, 5))
        net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
        broadcast_parameters(net)
        w0 = [p.detach().clone() for p in net.parameters()]
        red = GradReducer(net, bucket_bytes=256)              # tiny buckets -> several collectives
        assert len(red.buckets) > 1
        g = torch.Generator().manual_seed(7)
        X = torch.randn(4, 3, 6, 6, generator=g)
        y = torch.randint(0, 5, (4,), generator=g)
        xs, ys = X[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]
        for _ in range(2):                                    # twice: reducer state resets correctly
            net.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(net(xs), ys).backward()
            red.finish()
        grads = [p.grad.detach().clone() for p in net.parameters()]
        assert net[0].weight.grad.stride() == net[0].weight.stride()
        # single-process reference on the full batch (mean of the two half-batch means)
        ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                                  torch.nn.Conv2d(8, 4, 1), torch.nn.Flatten(), torch.nn.Linear(4 * 36, 5))
        for p, w in zip(ref.parameters(), w0):
            p.data.copy_(w)
        torch.nn.functional.cross_entropy(ref(X), y).backward()
        err = max((a - b.grad).abs().max().item() for a, b in zip(grads, ref.parameters()))
        q.put((rank, err, [w.sum().item() for w in w0]))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][2] == res[1][2]                 # broadcast made the weights identical
    for _, err, _ in res:
        assert err < 1e-6, err


def _arena_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from ee_semantic_segmentation_amd._lib import lib
        from ee_semantic_segmentation_amd.parallel import ArenaReducer
        # a stand-in for engine.GradArena: one flat fp32 buffer, units in backward-completion order
        ranges = [(0, 100), (100, 300), (300, 1000), (1000, 1024)]
        g = torch.Generator().manual_seed(50 + rank)
        flat = torch.randn(1024, generator=g)
        mine = flat.clone()
        cfg = SimpleNamespace(arena=SimpleNamespace(flat=flat, unit_ranges=ranges), on_unit_done=None)
        red = ArenaReducer(SimpleNamespace(cfg=cfg), bucket_bytes=800, reserve_cus=16)      # buckets: [0,300) [300,1000) [1000,1024)
        assert red.active and len(red.buckets) == 3 and cfg.on_unit_done is not None
        seen = []
        for step in range(2):                      # twice: the reducer resets its state in finish()
            if step:
                flat.copy_(mine)
            assert lib().eeseg_get_option(8) == 256
            cfg.on_unit_done(0)
            assert red._next == 0                  # bucket 0 needs units 0 and 1
            cfg.on_unit_done(1)
            assert red._next == 1
            seen.append(lib().eeseg_get_option(8))  # the conv launch plans now leave 16 CUs to the collectives
            cfg.on_unit_done(2)
            red.finish()                           # launches the last bucket (unit 3 never reported), joins, restores the plan
            assert lib().eeseg_get_option(8) == 256
        q.put((rank, seen, mine.tolist(), flat.tolist()))
    finally:
        dist.destroy_process_group()


def test_arena_reducer_averages_buckets_and_switches_the_cu_plan():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_arena_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (torch.tensor(res[0][2]) + torch.tensor(res[1][2])) / 2
    for _, seen, _, got in res:
        assert seen == [240, 240]
        assert torch.allclose(torch.tensor(got), want, atol=1e-6)
