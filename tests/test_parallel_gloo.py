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


def _algebra_worker(rank, world, port, q):
    """SyncBN statistics and the global CE normaliser through the SAME host functions the HIP path calls
    (engine.sync_bn_sums / engine.global_mean_normaliser / Config.all_reduce), on CPU tensors over gloo."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ee_semantic_segmentation_amd import engine as E
        sub = dist.new_group([0, 1])                     # a sub-group object, not the default group
        cfg = E.Config()
        cfg.sync_bn, cfg.group = True, sub
        assert cfg.dp_active() and cfg.sync_active() and cfg.world() == world and cfg.dp_world() == world
        g = torch.Generator().manual_seed(9)
        x = torch.randn(6, 5, 7, 7, generator=g) * 2 + 0.5         # whole batch [B,C,H,W]; this rank's shard: 3 images
        xs = x[rank * 3:(rank + 1) * 3].double()
        sums = torch.stack([xs.sum((0, 2, 3)), (xs * xs).sum((0, 2, 3))])
        sums, count = E.sync_bn_sums(cfg, sums, xs.numel() // 5)
        mean = sums[0] / count
        var = sums[1] / count - mean * mean
        ref_mean, ref_var = x.double().mean((0, 2, 3)), x.double().var((0, 2, 3), unbiased=False)
        # CE: per-exit (loss sum, valid count) of the shard -> normaliser; the DP average of the rank losses must be
        # the whole-batch mean loss (oracle = torch cross_entropy with ignore_index on the whole batch)
        C = 4
        logits = torch.randn(2, 6, C, 9, 9, generator=g)                       # [E,B,C,H,W]
        tgt = torch.randint(0, C + 1, (6, 9, 9), generator=g)
        tgt[:3][torch.rand(3, 9, 9, generator=g) < 0.5] = C                 # unequal void share per shard
        sl = slice(rank * 3, (rank + 1) * 3)
        loss_sum = torch.stack([torch.nn.functional.cross_entropy(logits[e, sl], tgt[sl], ignore_index=C,
                                                                  reduction="sum") for e in range(2)]).double()
        cnt = torch.full((2,), float((tgt[sl] != C).sum()), dtype=torch.float64)
        mine = loss_sum / E.global_mean_normaliser(cfg, cnt)
        avg = mine.clone()
        dist.all_reduce(avg, group=sub)
        avg /= world
        want = torch.stack([torch.nn.functional.cross_entropy(logits[e], tgt, ignore_index=C) for e in range(2)])
        q.put((rank, (mean - ref_mean).abs().max().item(), (var - ref_var).abs().max().item(),
               (avg - want.double()).abs().max().item(), count))
    finally:
        dist.destroy_process_group()


def test_syncbn_statistics_and_ce_normaliser_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_algebra_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, emean, evar, eloss, count in res:
        assert count == 6 * 49
        assert emean < 1e-12 and evar < 1e-11 and eloss < 1e-6, (emean, evar, eloss)


def test_grad_arena_views_have_the_parameter_layout():
    """every p.grad must be a view INTO the arena with exactly p's strides, or the fused SGD step
    re-binds it to a private copy that the backward kernels never refresh (torch leaves the strides of a
    [Co,Ci,1,1] channels_last weight at (Ci,1,1,1), unlike a permuted KRSC view)."""
    from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
    from ee_semantic_segmentation_amd.optim import _same_layout
    net = branchyDeepv3(None, "deeplabv3_resnet50", 1, 65, count_branches=False,
                        branch_params=dict(atrous_rates=[2, 4], nout_channels=128, bottleneck=64))
    arena = net.enable_grad_arena()
    lo, hi = arena.flat.data_ptr(), arena.flat.data_ptr() + arena.flat.numel() * 4
    n1x1 = 0
    for name, p in net.named_parameters():
        assert p.grad is not None and lo <= p.grad.data_ptr() < hi, name
        assert _same_layout(p.grad, p), (name, p.grad.stride(), p.stride())
        if p.dim() == 4 and p.shape[2] == 1:
            n1x1 += 1
            kv = arena.kernel_view[p]
            kv.copy_(torch.arange(kv.numel(), dtype=torch.float32).view(kv.shape))
            assert torch.equal(p.grad, kv[:p.shape[0]].permute(0, 3, 1, 2)), name     # same bytes, same element order
    assert n1x1 >= 40


def test_shard_sampler_partitions_every_global_batch():
    """parallel.ShardSampler (SURVEY 8e "DistributedSampler-style seeding"): for every epoch the ranks' index streams are the
    rank-major slices of the SAME global batches one process would train on; the ragged last batch is dropped when N > 1 and
    kept (the reference's drop_last=False) for one process; epochs reshuffle."""
    from ee_semantic_segmentation_amd.parallel import ShardSampler, eval_shard
    n, B = 23, 8
    for world in (2, 4):
        b = B // world
        for epoch in (0, 1, 5):
            one = ShardSampler(n, B, 1, 0, seed=3)
            one.set_epoch(epoch)
            whole = list(one)
            assert sorted(whole) == list(range(n)) and len(one) == n
            ranks = []
            for r in range(world):
                s = ShardSampler(n, B, world, r, seed=3)
                s.set_epoch(epoch)
                ranks.append(list(s))
                assert len(ranks[-1]) == len(s) == (n // B) * b
            for k in range(n // B):                       # global batch k = concatenation of the ranks' k-th local batches
                got = [i for r in range(world) for i in ranks[r][k * b:(k + 1) * b]]
                assert got == whole[k * B:(k + 1) * B], (world, epoch, k)
        a, c = ShardSampler(n, B, 1, 0, seed=3), ShardSampler(n, B, 1, 0, seed=3)
        a.set_epoch(1); c.set_epoch(2)
        assert list(a) != list(c)
    with pytest.raises(ValueError):
        ShardSampler(n, 6, 4, 0)
    ds = list(range(11))
    parts = [list(eval_shard(ds, 3, r)) for r in range(3)]
    assert sorted(i for p in parts for i in p) == ds and [len(p) for p in parts] == [4, 4, 3]
    assert eval_shard(ds, 1, 0) is ds


def _counter_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from ee_semantic_segmentation_amd import engine as E
        from ee_semantic_segmentation_amd.eval_mIoU import reduce_counters
        from ee_semantic_segmentation_amd.parallel import dp_info
        cfg = E.Config()
        cfg.collective = lambda t, group: dist.all_reduce(t, group=group)      # a transport is attached (as init_data_parallel does)
        net = SimpleNamespace(cfg=cfg, parameters=lambda: iter([torch.zeros(1)]))
        assert dp_info(net) == (world, rank)
        g = torch.Generator().manual_seed(70 + rank)
        accs = [SimpleNamespace(accumulator=torch.randint(0, 1 << 22, (3, 5), generator=g).float()) for _ in range(2)]
        mine = [a.accumulator.clone() for a in accs]
        reduce_counters(net, accs)
        q.put((rank, [m.tolist() for m in mine], [a.accumulator.tolist() for a in accs]))
    finally:
        dist.destroy_process_group()


def test_miou_counters_are_summed_over_the_ranks_once_per_evaluation():
    """eval_mIoU.reduce_counters ("all-reduce [E,3,C] fp32 counters once per eval"): every rank ends with the
    exact integer sum of the ranks' per-exit (TP, FP, FN) accumulators (fp64 on the wire), world 2 over gloo."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_counter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (torch.tensor(res[0][1], dtype=torch.float64) + torch.tensor(res[1][1], dtype=torch.float64))
    for _, _, got in res:
        assert torch.equal(torch.tensor(got, dtype=torch.float64), want)


def test_bucket_plan_caps_the_exposed_tail_bucket():
    """ArenaReducer.plan_buckets: buckets close at >= bucket_bytes; the LAST bucket (launched when backward ends, its
    all-reduce exposed) is cut at a unit boundary so that at most TAIL_BYTES remain behind the cut (the docstring
    promised a small tail, the loop left up to bucket_bytes)."""
    from ee_semantic_segmentation_amd.parallel import ArenaReducer
    M = 1 << 18                                    # floats per MiB
    sizes = [20, 20, 1, 6, 1, 1, 1]                # MiB per unit, backward completion order (the stem is last and small)
    ranges, off = [], 0
    for s_ in sizes:
        ranges.append((off, off + s_ * M))
        off += s_ * M
    plan = ArenaReducer.plan_buckets(ranges, 32 << 20)
    # [20, 20] closes at 40 MiB; the rest (10 MiB) would be ONE exposed bucket -> cut so that <= 4 MiB stay behind
    assert plan[0] == (0, 1, 0, 40 * M)
    assert plan[1][:2] == (2, 3) and plan[2][:2] == (4, 6)
    assert (plan[2][3] - plan[2][2]) * 4 == 3 << 20 and plan[2][3] == off
    # buckets tile the arena, in order
    assert plan[0][2] == 0 and all(a[3] == b[2] for a, b in zip(plan, plan[1:]))
    # a last unit that alone exceeds the cap cannot be cut: it stays whole, the units before it leave earlier
    plan2 = ArenaReducer.plan_buckets([(0, 10 * M), (10 * M, 12 * M), (12 * M, 22 * M)], 64 << 20)
    assert [p[:2] for p in plan2] == [(0, 1), (2, 2)]
    # one small bucket: untouched
    assert ArenaReducer.plan_buckets([(0, 100), (100, 300)], 1 << 20) == [(0, 1, 0, 300)]
