#!/usr/bin/env python3
"""Headline benchmark: images/sec of one data-parallel training step (forward +
backward + SGD step) of the early-exit DeepLabV3 on synthetic 513x513 batches.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload at N=1 = BASELINE.json configs[1]: DeepLabV3-ResNet50, 2 exits, 513x513,
B=16, 21 classes, bf16 MFMA compute (fp32 master weights / statistics / loss).
N>1 keeps 16 images per GPU (weak scaling); gradients are all-reduced over RCCL.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`
(live HIP-event timing of the dominant conv kernel family vs the dense bf16 MFMA
peak) and, at N=1, `cpu_baseline` (the oracle = torch CPU fp32 restatement of the
reference path, timed on a bounded sample on this host's cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md
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
    n = 4                                          # ~10 s of CPU work on the GPU box's 16-core share
    t0 = time.perf_counter()
    for _ in range(n):
        step(X, y)
    dt = time.perf_counter() - t0
    return {"value": n * B / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} timed fwd+bwd+SGD steps of the torch-CPU fp32 oracle, {arch} {n_branches + 1} exits, "
                      f"{img}x{img}, B={B} ({dt:.1f} s)"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--branches", type=int, default=1)
    ap.add_argument("--img", type=int, default=513)
    ap.add_argument("--batch-per-gpu", type=int, default=16)
    ap.add_argument("--classes", type=int, default=21)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--loss", default="ce", choices=["ce", "lovasz"],
                    help="ce = BrXEntropyLoss (the headline workload); lovasz = BSL.LovaszSoftmax (BASELINE configs[4] shape)")
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="N>1: CUs the conv launch plans leave to RCCL while gradient buckets are in flight (default: "
                         "EESEG_RCCL_RESERVE_CUS or 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run every step eagerly (no HIP-graph replay)")
    ap.add_argument("--roofline-steps", type=int, default=2)
    ap.add_argument("--conv-tap-inner", type=int, default=None, help="override EESEG_OPT_CONV_TAP_INNER (0|1)")
    ap.add_argument("--wgrad-big-grid", type=str, default=None, metavar="BLOCKS,ROUNDS",
                    help="eeseg_set_wgrad_big_grid: concurrent blocks / max rounds of the 256x256 wgrad kernel")
    ap.add_argument("--wgrad-big-min-ktiles", type=int, default=None)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B switch: eeseg_set_option(KEY, VALUE) (include/eeseg.h EESEG_OPT_*); repeatable")
    ap.add_argument("--conv-pipe", type=int, default=None, help="override EESEG_OPT_CONV_PIPE (1|2)")
    ap.add_argument("--ew-grid-cap", type=int, default=None)
    ap.add_argument("--wgrad-blocks", type=int, default=None)
    ap.add_argument("--conv-narrow-max", type=int, default=None)
    ap.add_argument("--conv-auto-narrow", type=int, default=None)
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"

    from ee_semantic_segmentation_amd import kernels as K
    from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
    from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
    from ee_semantic_segmentation_amd.optim import SGD
    from ee_semantic_segmentation_amd.parallel import ArenaReducer, GraphedTrainStep, broadcast_parameters

    from ee_semantic_segmentation_amd._lib import lib as _eelib
    if args.conv_tap_inner is not None:
        _eelib().eeseg_set_option(2, args.conv_tap_inner)
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
    if args.conv_pipe is not None:
        _eelib().eeseg_set_option(1, args.conv_pipe)
    if args.ew_grid_cap is not None:
        _eelib().eeseg_set_ew_grid_cap(args.ew_grid_cap)
    if args.conv_auto_narrow is not None:
        _eelib().eeseg_set_option(4, args.conv_auto_narrow)
    if args.conv_narrow_max is not None:
        _eelib().eeseg_set_option(3, args.conv_narrow_max)
    if args.wgrad_blocks is not None:
        _eelib().eeseg_set_wgrad_target_blocks(args.wgrad_blocks)
    C, img, B = args.classes, args.img, args.batch_per_gpu
    torch.manual_seed(0)
    net = branchyDeepv3(None, f"deeplabv3_{args.arch}", args.branches, img, count_branches=False, num_classes=C,
                        compute_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32,
                        fused_outputs=True).to(dev)
    net.cfg.sync_bn = args.sync_bn and world > 1
    net.cfg.overlap_wgrad = args.overlap_wgrad
    broadcast_parameters(net)
    E = net.n_branches + 1
    if args.loss == "lovasz":
        from ee_semantic_segmentation_amd import branchy_seg_losses as BSL
        crit = BSL.LovaszSoftmax(ignore=C, n_branches=E - 1)
    else:
        crit = BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=E)
    lr = 0.01                                                  # param groups as deepv3_funcs.py:74-99
    opt = SGD([{"params": net.base_model.parameters(), "lr": lr},
               {"params": net.branches.parameters(), "lr": lr},
               {"params": net.classifier.parameters(), "lr": lr * 1.1}], lr=lr, momentum=0.9, weight_decay=5e-4)
    net.enable_grad_arena()
    reducer = ArenaReducer(net, reserve_cus=args.reserve_cus)
    X, y = synth_batch(B, C, img, img, 1234 + rank, dev)
    net.train()
    # warm-up: 2 eager steps (allocator, momentum buffers), then the step is captured into a
    # HIP graph and every later call is a replay
    runner = GraphedTrainStep(net, crit, opt, reducer, warmup=2, use_graph=not args.no_graph)

    def step():
        return runner(X, y)

    log(f"model ready: {args.arch} E={E} {img}x{img} B={B}/GPU {args.dtype}; warmup {args.warmup}")
    # the graph path needs 4 untimed steps (2 eager, the capture, a first replay): --warmup below that is raised, and the
    # number actually run is reported as config.warmup_steps_run
    warmup_run = max(args.warmup, 0 if args.no_graph else 4)
    for i in range(warmup_run):
        l = step()
        if rank == 0:
            torch.cuda.synchronize()
            log(f"warmup {i} done, loss {float(l.item()):.4f}")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    log(f"timed {args.steps} steps in {dt:.3f} s")
    loss_val = float(last.item())
    # per-kernel HIP-event timing for the roofline: graph replays cannot be bracketed kernel by
    # kernel, so the same step runs eagerly (identical kernels / shapes) right after the timed region
    prof = None
    prof_steps = 0
    if not args.no_kernel_events and rank == 0 or (not args.no_kernel_events and world > 1):
        net.cfg.overlap_wgrad = 0              # per-kernel events need a serial timeline
        K.PROFILE = []
        for _ in range(args.roofline_steps):
            runner._eager(X, y)
            prof_steps += 1
        torch.cuda.synchronize()
        prof, K.PROFILE = K.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        flop_img = 3 * 2.0 * net.macs(img)                    # fwd + bwd = 3 x fwd, FLOP = 2 MAC
        roof = None
        if prof:
            fam = {}
            for name, flops, e0, e1, nbytes in prof:
                f = fam.setdefault(name, [0.0, 0.0, 0, 0.0])
                f[0] += flops
                f[1] += e0.elapsed_time(e1) * 1e-3
                f[2] += 1
                f[3] += nbytes
            name, (fl, sec, cnt, nb) = max(fam.items(), key=lambda kv: kv[1][1])
            traffic = None          # HBM bytes per launch from rocprofv3 PMC passes (profiles/, offline)
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                t = json.load(open(tpath))
                if t.get("workload_key") == f"{args.arch}-{E}-{img}-{B}-{args.dtype}":
                    traffic = t.get("hbm_bytes_per_launch", {}).get(name)
            roof = {"kernel": name, "bound": "mfma", "achieved": fl / sec / 1e12, "peak": PEAK_BF16_TFLOPS
                    if args.dtype == "bf16" else 157.3, "unit": "TFLOP/s", "frac": fl / sec / 1e12 /
                    (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3), "traffic": traffic,
                    "algorithmic_bytes_per_launch": nb / cnt, "flops_per_launch": fl / cnt,
                    "launches": cnt, "avg_launch_us": sec / cnt * 1e6,
                    "measured": f"HIP events around every launch, {prof_steps} eager steps after the timed region",
                    "families": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] / prof_steps * 1e3,
                                     "launches_per_step": v[2] / prof_steps} for k, v in fam.items()},
                    "whole_step_tflops": value / world * flop_img / 1e12}
        line = {"metric": "images/sec at 513x513 fwd+bwd+SGD step (early-exit DeepLabV3 training)",
                "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"DeepLabV3-{args.arch} {E} exits, {img}x{img}, {C} classes, B={B}/GPU, "
                                       f"{'per-exit CE (sum)' if args.loss == 'ce' else 'raw-logit Lovasz (sum over exits)'}, SGD momentum 0.9 wd 5e-4",
                           "global_batch": world * B, "parallelism": f"dp{world}", "sync_bn": bool(net.cfg.sync_bn),
                           "hip_graph": bool(runner.graph is not None), "warmup_steps_run": warmup_run,
                           "flop_per_image": flop_img, "loss_last_step": loss_val},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle, torch CPU fp32) ...")
            line["cpu_baseline"] = cpu_baseline(args.arch, args.branches, C, img, 2, 1234)
            log("cpu baseline done")
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
