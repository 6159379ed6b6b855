#!/usr/bin/env python3
"""Headline benchmark: images/sec of one data-parallel training step (forward +
backward + gradient all-reduce + SGD step) of the early-exit DeepLabV3 on synthetic
513x513 batches.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload = the configuration BASELINE.json's metric is quoted on ("images/sec at
513x513 B=32 fwd+bwd, 1/2/4/8 MI355X" = configs[2]): DeepLabV3-ResNet101, 3 exits,
19 classes, 513x513, GLOBAL batch 32 (the reference trains at batch 32,
main_bradeepv3.py:119) -> 32/N images per GPU ("scaling": "strong"), bf16 MFMA compute
with fp32 master weights / statistics / loss, SyncBN over the N ranks (so BatchNorm sees
the reference's batch-32 statistics), gradients all-reduced over RCCL.
Rank 0 prints ONE JSON line (contract in the task statement) with
  * `roofline`: live HIP-event timing of the dominant conv kernel family vs the dense bf16
    MFMA peak, plus the 3x3-stack aggregate the north-star target names;
  * `cpu_baseline` (N=1): the oracle (torch CPU fp32 restatement of the reference path) timed
    on a bounded sample of the same architecture on this host's cores;
  * `secondary` (N=1): BASELINE configs[1] (R50, 2 exits, 21 classes, B=16, bf16), the fp32 parity mode of the
    headline workload (the mode the 1e-3 logit bar is held in), the per-GPU shards of the 4- and 8-GPU runs on this
    one GPU, BASELINE configs[4] (Lovasz, 769 x 769, 8 images) and configs[3] (inference: images/sec + exit histogram).
"""
import argparse
import gc
import json
import os
import sys
import time

# the hosts of this pool support dmabuf IPC only: without this RCCL's cross-process buffer sharing fails with
# `hipIpcGetMemHandle: invalid argument` (it is exported on the boxes already; kept here so a hand-built env works too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3        # v_mfma_f32_32x32x2_f32
PEAK_HBM_GBS = 8000.0


def synth_batch(B, C, H, W, seed, device):
    """randn image, piecewise-constant labels (32x32 blocks), ~5% void = C."""
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(B, 3, H, W, generator=g)
    blocks = torch.randint(0, C, (B, 1, (H + 31) // 32, (W + 31) // 32), generator=g).float()
    y = torch.nn.functional.interpolate(blocks, size=(H, W), mode="nearest").long()
    y[torch.rand(B, 1, H, W, generator=g) < 0.05] = C
    return X.to(device), y.to(device)


def cpu_baseline(arch, n_branches, C, img, B, seed):
    """A few timed fwd+bwd+SGD steps of the ORACLE (torch CPU fp32, reference-shaped loss
    path: unfused upsample -> stacked tensor -> per-exit CE) on this host's cores."""
    from oracle.deeplab_ref import branchyDeepv3 as Ref
    from oracle import losses_ref
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))                 # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    ref = Ref(f"deeplabv3_{arch}", n_branches, img, count_branches=False, num_classes=C).train()
    opt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    X, y = synth_batch(B, C, img, img, seed, "cpu")

    def step(x, t):
        out = ref(x)
        l = losses_ref.br_xentropy(out, t, ignore_index=C, b_reduction="sum", n_exits=n_branches + 1)
        opt.zero_grad()
        l.mean().backward()
        opt.step()

    xs, ts = synth_batch(2, C, 65, 65, seed, "cpu")
    step(xs, ts)                                   # page in kernels / allocator
    n = 3                                          # ~15-25 s of CPU work on the GPU box's 16-core share
    t0 = time.perf_counter()
    for _ in range(n):
        step(X, y)
    dt = time.perf_counter() - t0
    return {"value": n * B / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} timed fwd+bwd+SGD steps of the torch-CPU fp32 oracle, {arch} {n_branches + 1} exits, "
                      f"{C} classes, {img}x{img}, B={B} ({dt:.1f} s)"}


def _rccl_version():
    from ee_semantic_segmentation_amd.comm import rccl_version
    return rccl_version()


def _staged_transport(world):
    """REHEARSAL ONLY (--dp-transport gloo): the collectives of an N-rank step staged through host memory over gloo, so the
    multi-process plumbing of this file (rendezvous, shards, barriers, rank-0 line, teardown) can be run with N ranks on
    ONE GPU, where ranks cannot share an RCCL communicator.  Never a measurement: the step cannot be graph-captured."""
    calls = [0]

    def reduce_(t, group):
        calls[0] += 1
        h = t.detach().cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)

    reduce_.calls = calls

    def gather_(t, group):
        h = t.detach().cpu().contiguous()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return torch.stack(parts).to(t.device)
    return reduce_, gather_


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


class Run:
    """One workload: network + loss + optimizer + (graphed) step on this rank's shard."""

    def __init__(self, arch, branches, classes, img, batch, dtype, loss, sync_bn, world, rank, dev, args):
        from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
        from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
        from ee_semantic_segmentation_amd.optim import SGD
        from ee_semantic_segmentation_amd.parallel import ArenaReducer, GraphedTrainStep, init_data_parallel
        self.arch, self.C, self.img, self.B, self.dtype, self.loss_name = arch, classes, img, batch, dtype, loss
        self.world = world
        torch.manual_seed(0)
        net = branchyDeepv3(None, f"deeplabv3_{arch}", branches, img, count_branches=False, num_classes=classes,
                            compute_dtype=torch.bfloat16 if dtype == "bf16" else torch.float32,
                            fused_outputs=True).to(dev)
        # N > 1: RCCL communicators through libeeseg (rendezvous over the gloo group), SyncBN, rank 0's weights everywhere.
        # EESEG_FORCE_ALLREDUCE=1 (with RANK/WORLD_SIZE=1 set): a 1-rank communicator still issues every collective -
        # rehearses the N > 1 graph (SyncBN all-reduces, CE count, arena buckets) on one GPU
        net.cfg.overlap_wgrad = args.overlap_wgrad
        init_data_parallel(net, sync_bn=sync_bn, transport=_staged_transport(world) if args.dp_transport == "gloo" else None)
        self.E = net.n_branches + 1
        if loss == "lovasz":
            from ee_semantic_segmentation_amd import branchy_seg_losses as BSL
            crit = BSL.LovaszSoftmax(ignore=classes, n_branches=self.E - 1)
        else:
            crit = BrXEntropyLoss(ignore_index=classes, b_reduction="sum", n_exits=self.E)
        lr = 0.01                                                  # param groups as deepv3_funcs.py:74-99
        opt = SGD([{"params": net.base_model.parameters(), "lr": lr},
                   {"params": net.branches.parameters(), "lr": lr},
                   {"params": net.classifier.parameters(), "lr": lr * 1.1}], lr=lr, momentum=0.9, weight_decay=5e-4)
        net.enable_grad_arena()
        reducer = ArenaReducer(net, reserve_cus=args.reserve_cus)
        self.X, self.y = synth_batch(batch, classes, img, img, 1234 + rank, dev)
        net.train()
        # warm-up: 2 eager steps (allocator, momentum buffers), then the step is captured into a
        # HIP graph and every later call is a replay
        self.reserve_cus = reducer.reserve_cus if reducer.active else 0
        self.reducer = reducer
        self.runner = GraphedTrainStep(net, crit, opt, reducer, warmup=2,
                                       use_graph=not args.no_graph and net.cfg.collective is None)
        self.net = net

    def step(self):
        return self.runner(self.X, self.y)

    def collective_counts(self):
        """(small 'stat' collectives, gradient buckets) enqueued so far by this process (host-side counters: they advance in
        eager steps and during a capture, not per graph replay)."""
        cfg = self.net.cfg
        grad = self.reducer.launched
        if cfg.comm is not None:
            return cfg.comm.stat.calls, cfg.comm.grad.calls
        if cfg.collective is not None and hasattr(cfg.collective, "calls"):
            return cfg.collective.calls[0] - grad, grad
        return 0, 0

    def workload(self):
        loss = "CE" if self.loss_name == "ce" else "Lovasz"
        return (f"{self.dtype} DeepLabV3-{self.arch} E={self.E} {self.img}x{self.img} C={self.C} B={self.world * self.B}"
                f" ({self.B}/GPU) {loss} SGD")

    def flop_per_image(self):
        return 3 * 2.0 * self.net.macs(self.img)                  # fwd + bwd = 3 x fwd, FLOP = 2 MAC


def timed(run, steps, warmup, world, rank, no_graph):
    """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize fences."""
    # the graph path needs 4 untimed steps (2 eager, the capture, a first replay): --warmup below that is raised, and
    # the number actually run is reported as config.warmup_steps_run
    warmup_run = max(warmup, 0 if no_graph else 4)
    for i in range(warmup_run):
        c0 = run.collective_counts()
        l = run.step()
        if i == 0:                                  # the first step is always eager: its host-side counts are one step's
            c1 = run.collective_counts()
            run.stat_collectives_per_step, run.grad_collectives_per_step = c1[0] - c0[0], c1[1] - c0[1]
        if rank == 0:
            torch.cuda.synchronize()
            log(f"warmup {i} done, loss {float(l.item()):.4f}")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # per-step HIP events on the compute stream (median reported beside the contract's wall-clock mean)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        evs[i].record()
        last = run.step()
    evs[steps].record()
    fence()
    dt = time.perf_counter() - t0
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    run.step_ms_median = per[len(per) // 2] if steps % 2 else 0.5 * (per[steps // 2 - 1] + per[steps // 2])
    run.step_ms_min, run.step_ms_max = per[0], per[-1]
    return dt, float(last.item()), warmup_run


def kernel_events(run, nsteps):
    """Per-kernel HIP-event timing: graph replays cannot be bracketed kernel by kernel, so the same
    step runs eagerly (identical kernels / shapes / launch plans) right after the timed region."""
    from ee_semantic_segmentation_amd import kernels as K
    from ee_semantic_segmentation_amd.comm import Lane
    run.net.cfg.overlap_wgrad = 0              # per-kernel events need a serial timeline
    K.PROFILE = []
    Lane.PROFILE = []
    for _ in range(nsteps):
        run.runner._eager(run.X, run.y)
    torch.cuda.synchronize()
    prof, K.PROFILE = K.PROFILE, None
    joins, Lane.PROFILE = Lane.PROFILE, None
    # how long the compute stream stood at a lane join, per eager step (ms): the EXPOSED part of the collectives
    waits = {}
    for name, e0, e1 in joins:
        waits[name] = waits.get(name, 0.0) + e0.elapsed_time(e1)
    run.exposed_ms = {k: v / max(nsteps, 1) for k, v in waits.items()}
    run.join_count = {k: sum(1 for n, _, _ in joins if n == k) / max(nsteps, 1) for k in waits}
    return prof


def _coop_timeouts():
    from ee_semantic_segmentation_amd import kernels as K
    n = K.coop_timeouts()
    if n:
        sys.stderr.write(f"[bench] WARNING: {n} cooperative-kernel barrier state(s) report a launch whose grid was not co-resident: "
                         "the gradients of this run are not to be trusted (lower EESEG_OPT_CONV_CUS / --reserve-cus)\n")
    return n


def roofline(prof, prof_steps, dtype, workload_key):
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    fam, k3, shapes = {}, [0.0, 0.0, 0], {}
    for name, flops, e0, e1, nbytes, tag in prof:
        sec = e0.elapsed_time(e1) * 1e-3
        sh = shapes.setdefault(tag, [0.0, 0.0, 0, 0.0])
        sh[0] += flops
        sh[1] += sec
        sh[2] += 1
        sh[3] += nbytes
        f = fam.setdefault(name, [0.0, 0.0, 0, 0.0])
        f[0] += flops
        f[1] += sec
        f[2] += 1
        f[3] += nbytes
        if tag.startswith("3x3"):
            k3[0] += flops
            k3[1] += sec
            k3[2] += 1
    name, (fl, sec, cnt, nb) = max(fam.items(), key=lambda kv: kv[1][1])
    traffic, tsrc = None, None          # HBM bytes per launch from rocprofv3 PMC passes (profiles/, offline)
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("workload_key") == workload_key:
            traffic = t.get("hbm_bytes_per_launch", {}).get(name)
            tsrc = "profiles/pmc_traffic.json (offline rocprofv3 --pmc passes of this workload; not collected in this run)"
    return {"kernel": name, "bound": "mfma", "achieved": fl / sec / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": fl / sec / 1e12 / peak, "traffic": traffic, "traffic_source": tsrc,
            "algorithmic_bytes_per_launch": nb / cnt, "flops_per_launch": fl / cnt,
            "launches": cnt, "avg_launch_us": sec / cnt * 1e6,
            "measured": f"HIP events around every launch, {prof_steps} eager steps after the timed region",
            "stack_3x3": {"what": "all 3x3 / atrous conv calls (forward, data-gradient, weight-gradient), every kernel family",
                          "tflops": k3[0] / max(k3[1], 1e-12) / 1e12, "frac": k3[0] / max(k3[1], 1e-12) / 1e12 / peak,
                          "ms_per_step": k3[1] / prof_steps * 1e3, "launches_per_step": k3[2] / prof_steps},
            "families": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] / prof_steps * 1e3,
                             "launches_per_step": v[2] / prof_steps} for k, v in fam.items()},
            # per layer shape: measured time against that layer's own roofline max(flops / MFMA peak, algorithmic
            # bytes / HBM peak) - most 1x1 layers sit below the bf16 ridge and are HBM-bound
            "by_shape": {k: {"calls_per_step": v[2] / prof_steps, "us_per_call": v[1] / v[2] * 1e6,
                             "ms_per_step": v[1] / prof_steps * 1e3, "tflops": v[0] / v[1] / 1e12,
                             "roofline_us": max(v[0] / v[2] / (peak * 1e12), v[3] / v[2] / (PEAK_HBM_GBS * 1e9)) * 1e6,
                             "bound": "mfma" if v[0] / (peak * 1e12) > v[3] / (PEAK_HBM_GBS * 1e9) else "hbm",
                             "frac_of_roofline": max(v[0] / (peak * 1e12), v[3] / (PEAK_HBM_GBS * 1e9)) / v[1]}
                         for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:24]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--branches", type=int, default=2)
    ap.add_argument("--img", type=int, default=513)
    ap.add_argument("--global-batch", type=int, default=32, help="images per step over ALL ranks (strong scaling)")
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="override: fixed images per GPU (weak scaling; other BASELINE shapes)")
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--loss", default="ce", choices=["ce", "lovasz"],
                    help="ce = BrXEntropyLoss (the headline workload); lovasz = BSL.LovaszSoftmax (BASELINE configs[4] shape)")
    ap.add_argument("--no-sync-bn", action="store_true", help="N>1: local-batch BatchNorm instead of SyncBN")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="N>1: CUs the conv launch plans leave to RCCL while gradient buckets are in flight (default: "
                         "EESEG_RCCL_RESERVE_CUS, else 32 from 8 ranks on and 0 below - parallel.ArenaReducer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run every step eagerly (no HIP-graph replay)")
    ap.add_argument("--roofline-steps", type=int, default=2)
    ap.add_argument("--wgrad-big-grid", type=str, default=None, metavar="BLOCKS,ROUNDS",
                    help="eeseg_set_wgrad_big_grid: concurrent blocks / max rounds of the 256x256 wgrad kernel")
    ap.add_argument("--wgrad-big-min-ktiles", type=int, default=None)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B switch: eeseg_set_option(KEY, VALUE) (include/eeseg.h EESEG_OPT_*); repeatable")
    ap.add_argument("--ew-grid-cap", type=int, default=None)
    ap.add_argument("--wgrad-blocks", type=int, default=None)
    ap.add_argument("--dp-transport", default="rccl", choices=["rccl", "gloo"],
                    help="rccl (default): RCCL through libeeseg.  gloo: REHEARSAL of the N-rank plumbing on one GPU - "
                         "collectives staged through the host, no graph capture; the line is marked, never a measurement")
    ap.add_argument("--dp-lanes", type=int, default=2, choices=[1, 2],
                    help="N>1: 2 (default) = gradient buckets and SyncBN collectives on lanes / communicators of their own; "
                         "1 = EVERY collective on one lane in program order (EESEG_DP_SINGLE_LANE=1: the conservative fallback, "
                         "comm.py)")
    ap.add_argument("--overlap-wgrad", type=int, default=0, nargs="?", const=1,
                    help="weight-gradient kernels on a side stream: 1 = beside the data-gradient, 2 = after it, beside the "
                         "BatchNorm backward of the layer below (per-kernel timings then overlap)")
    args = ap.parse_args()

    # RCCL / HIP print banners on stdout: keep fd 1 for the single JSON line only
    json_fd = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.dp_lanes == 1:
        os.environ["EESEG_DP_SINGLE_LANE"] = "1"
    if os.environ.get("EESEG_REHEARSAL_ONE_GPU") == "1":          # N ranks on the only GPU of a test box (--dp-transport gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # torch.distributed is the RENDEZVOUS only (the RCCL id, the timing maximum, barriers): gloo.  The data path is RCCL
        # over xGMI called through libeeseg on streams this package owns (ee_semantic_segmentation_amd/comm.py)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"

    from ee_semantic_segmentation_amd._lib import lib as _eelib
    if args.wgrad_big_min_ktiles is not None:
        _eelib().eeseg_set_wgrad_big_min_ktiles(args.wgrad_big_min_ktiles)
    if args.wgrad_big_grid:
        b, r = [int(v) for v in args.wgrad_big_grid.split(",")]
        if _eelib().eeseg_set_wgrad_big_grid(b, r) != 0:
            raise SystemExit("bad --wgrad-big-grid")
    for kv in args.opt:
        k, v = kv.split("=")
        if _eelib().eeseg_set_option(int(k), int(v)) != 0:
            raise SystemExit(f"bad --opt {kv}")
    if args.ew_grid_cap is not None:
        _eelib().eeseg_set_ew_grid_cap(args.ew_grid_cap)
    if args.wgrad_blocks is not None:
        _eelib().eeseg_set_wgrad_target_blocks(args.wgrad_blocks)

    if args.batch_per_gpu is not None:
        B, scaling = args.batch_per_gpu, "weak"
    else:
        if args.global_batch % world:
            raise SystemExit(f"global batch {args.global_batch} is not divisible by {world} ranks")
        B, scaling = args.global_batch // world, "strong"
    run = Run(args.arch, args.branches, args.classes, args.img, B, args.dtype, args.loss, not args.no_sync_bn, world,
              rank, dev, args)
    log(f"model ready: {run.workload()}; warmup {args.warmup}")
    dt, loss_val, warmup_run = timed(run, args.steps, args.warmup, world, rank, args.no_graph)
    log(f"timed {args.steps} steps in {dt:.3f} s")
    prof, prof_steps = None, 0
    if not args.no_kernel_events:              # every rank runs them (the eager steps contain collectives)
        prof_steps = args.roofline_steps
        prof = kernel_events(run, prof_steps)
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, dt)          # host objects over the gloo rendezvous group
        rank_ms = sorted(v / args.steps * 1e3 for v in every)
        dt = max(every)

    line = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        flop_img = run.flop_per_image()
        roof = None
        if prof:
            roof = roofline(prof, prof_steps, args.dtype, f"{args.arch}-{run.E}-{args.img}-{B}-{args.dtype}")
            roof["whole_step_tflops"] = value / world * flop_img / 1e12
        line = {"metric": "images/sec at 513x513 B=32 fwd+bwd (+ gradient all-reduce + SGD step), early-exit DeepLabV3 training",
                "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": run.workload(), "global_batch": world * B, "batch_per_gpu": B,
                           "parallelism": f"dp{world}", "sync_bn": bool(run.net.cfg.sync_bn),
                           "hip_graph": bool(run.runner.graph is not None), "warmup_steps_run": warmup_run,
                           "step_ms_hip_events": {"median": run.step_ms_median, "min": run.step_ms_min,
                                                  "max": run.step_ms_max, "rank": 0},
                           "collectives": ("RCCL %d via libeeseg, 2 communicators, package-owned lanes" % _rccl_version())
                           if run.net.cfg.comm is not None else
                           ("REHEARSAL: staged through the host over gloo - not a measurement" if run.net.cfg.collective else None),
                           "reserve_cus": run.reserve_cus,
                           # what the first multi-GPU record is read with: bucket plan, collectives per
                           # step, how long the compute stream stood at the lane joins (eager steps after the timed region),
                           # the per-rank step spread
                           "dp_lanes": (1 if getattr(run.net.cfg.comm, "single_lane", False) else 2) if run.net.cfg.comm is not None else None,
                           "defer_wgrad": bool(run.net.cfg.defer_wgrad),
                           # the cooperative kernels (one-launch BatchNorm backward, weight-gradient combine / groups) need their whole
                           # grid resident; a launch that found it was not gives up with WRONG results and sets a flag: must be 0
                           "coop_timeouts": _coop_timeouts(),
                           "group_wgrad": bool(getattr(run.net.cfg, "group_wgrad", False)),
                           "n_buckets": len(run.reducer.buckets) if run.reducer.active else 0,
                           "bucket_mib": [round((b - a) * 4 / 2 ** 20, 1) for _, _, a, b in run.reducer.buckets] if run.reducer.active else [],
                           "syncbn_collectives_per_step": getattr(run, "stat_collectives_per_step", None),
                           "grad_collectives_per_step": getattr(run, "grad_collectives_per_step", None),
                           "exposed_allreduce_ms": getattr(run, "exposed_ms", None),
                           "lane_joins_per_step": getattr(run, "join_count", None),
                           "step_ms_over_ranks": {"min": rank_ms[0], "median": rank_ms[len(rank_ms) // 2], "max": rank_ms[-1]},
                           "flop_per_image": flop_img, "loss_last_step": loss_val,
                           "splits": list(run.net.split_names)},
                "roofline": roof}
    headline = (args.arch, args.branches, args.classes, args.img)
    dp_comm = run.net.cfg.comm
    run.runner.graph = None                       # the captured RCCL kernels go before their communicators
    del run, prof
    gc.collect()
    torch.cuda.empty_cache()
    if dp_comm is not None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dp_comm.close()

    if rank == 0 and world == 1 and not args.no_secondary:
        sec = {}
        try:
            # BASELINE configs[1]: R50, 2 exits, 21 classes, B=16 on one GPU (round 1's bench line)
            r2 = Run("resnet50", 1, 21, 513, 16, "bf16", "ce", False, 1, 0, dev, args)
            d2, l2, _ = timed(r2, args.steps, args.warmup, 1, 1, args.no_graph)
            sec["configs1_r50_e2_b16_bf16"] = {"workload": r2.workload(), "value": 16 * args.steps / d2,
                                               "unit": "images/sec", "ms_per_step": d2 / args.steps * 1e3,
                                               "whole_step_tflops": 16 * args.steps / d2 * r2.flop_per_image() / 1e12,
                                               "loss_last_step": l2}
            del r2
            gc.collect()
            torch.cuda.empty_cache()
            # the headline workload itself in the fp32 parity mode (exact-fp32 MFMA: the mode the 1e-3 logit / exact-mask
            # bar is held in) - same batch, same step count, same HIP-graph replay as the headline line
            r3 = Run(headline[0], headline[1], headline[2], headline[3], B, "f32", "ce", False, 1, 0, dev, args)
            d3, l3, _ = timed(r3, args.steps, args.warmup, 1, 1, args.no_graph)
            sec["f32_parity_mode"] = {"workload": r3.workload(), "value": B * args.steps / d3, "unit": "images/sec",
                                      "ms_per_step": d3 / args.steps * 1e3, "steps": args.steps,
                                      "hip_graph": bool(r3.runner.graph is not None),
                                      "step_ms_hip_events_median": r3.step_ms_median,
                                      "whole_step_tflops": B * args.steps / d3 * r3.flop_per_image() / 1e12,
                                      "peak_tflops": PEAK_F32_TFLOPS,
                                      "frac_of_f32_matrix_peak": B * args.steps / d3 * r3.flop_per_image() / 1e12 / PEAK_F32_TFLOPS,
                                      "loss_last_step": l3}
            del r3
            gc.collect()
            torch.cuda.empty_cache()
            # the per-GPU shards of the N-GPU strong-scaling runs (global B=32) on THIS one GPU, no collectives: what the
            # kernels alone allow at N = 2, 4 and 8 (the >= 6x target is decided by the small-M kernels first)
            for bs in (16, 8, 4):
                r4 = Run(headline[0], headline[1], headline[2], headline[3], bs, "bf16", "ce", False, 1, 0, dev, args)
                d4, l4, _ = timed(r4, max(args.steps, 20), args.warmup, 1, 1, args.no_graph)
                n4 = max(args.steps, 20)
                sec[f"shard_b{bs}"] = {"workload": r4.workload(), "value": bs * n4 / d4, "unit": "images/sec",
                                       "ms_per_step": d4 / n4 * 1e3, "steps": n4,
                                       "step_ms_hip_events_median": r4.step_ms_median,
                                       "ranks_at_global_b32": 32 // bs,
                                       "kernel_only_speedup_at_that_many_gpus": (32 // bs) * (bs * n4 / d4) / line["value"]}
                del r4
                gc.collect()
                torch.cuda.empty_cache()
            # BASELINE configs[4] shape on this GPU: R101, 3 exits, 19 classes, 769 x 769, 8 images, raw-logit Lovasz (three
            # segmented radix sorts per step inside the captured graph), SURVEY 8d
            r5 = Run("resnet101", 2, 19, 769, 8, "bf16", "lovasz", False, 1, 0, dev, args)
            d5, l5, _ = timed(r5, args.steps, args.warmup, 1, 1, args.no_graph)
            sec["configs4_lovasz_b8_769"] = {"workload": r5.workload(), "value": 8 * args.steps / d5, "unit": "images/sec",
                                             "ms_per_step": d5 / args.steps * 1e3, "steps": args.steps,
                                             "hip_graph": bool(r5.runner.graph is not None),
                                             "whole_step_tflops": 8 * args.steps / d5 * r5.flop_per_image() / 1e12,
                                             "loss_last_step": l5}
            del r5
            gc.collect()
            torch.cuda.empty_cache()
            # BASELINE configs[3]: inference-only, R101, 4 exits, 1024 x 2048, entropy gates on the device: all exits and the
            # batched truly-progressive form with its exit histogram ("Inference config C4 reports images/sec and
            # exit histogram"); B = 1 and B = 8
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import infer_bench
            sec["configs3_infer"] = infer_bench.run(quick=True)
            gc.collect()
            torch.cuda.empty_cache()
        except Exception as exc:          # a secondary figure must never cost the headline line
            sec["error"] = repr(exc)
        line["secondary"] = sec
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle, torch CPU fp32) ...")
            line["cpu_baseline"] = cpu_baseline(headline[0], headline[1], headline[2], headline[3], 2, 1234)
            log("cpu baseline done")
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
