"""One-launch BatchNorm backward (eeseg_bn_bwd_coop) against the two-launch form, per shard shape (HIP events, graph-free).
    python scripts/bn_coop_bench.py [images]"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ee_semantic_segmentation_amd import kernels as K  # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def timed(fn, n=20):
    """us per call of `fn`, replayed from a HIP graph of n calls (eager launches are host-bound below ~17 us per call)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


for rows, Cc, mode in [(B * 65 * 65, 256, 2), (B * 65 * 65, 1024, 3), (B * 65 * 65, 512, 2), (B * 65 * 65, 2048, 3),
                       (B * 129 * 129, 64, 2), (B * 129 * 129, 256, 3), (B * 257 * 257, 64, 2)]:
    x = torch.randn(rows, Cc, device=DEV).to(torch.bfloat16)
    dy = torch.randn(rows, Cc, device=DEV).to(torch.bfloat16)
    ga = torch.rand(Cc, device=DEV) + 0.5
    mi, ss = K.bn_finalize(K.channel_stats(x), rows, ga, torch.zeros(Cc, device=DEV), 1e-5, 0.1,
                           torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV))
    y, mask = K.bn_apply(x, ss, residual=x, relu=True, want_mask=True)
    ysrc, kw = (mask, {}) if mode == 3 else (None, {"scale_shift": ss})
    dx = torch.empty_like(x)
    sums = torch.empty(2, Cc, device=DEV)

    def two():
        K.bn_bwd_reduce(dy, ysrc, x, mi, True, out=sums, **kw)
        K.bn_bwd_apply(dy, ysrc, x, mi, ga, sums, rows, True, dx=dx, **kw)

    def one():
        K.bn_bwd_coop(dy, ysrc, x, mi, ga, rows, True, out=sums, dx=dx, **kw)

    t2 = timed(two)
    ok = K.bn_bwd_coop_ok(x)
    t1 = timed(one) if ok else float("nan")
    mb = rows * Cc * 2 / 1e6
    if os.environ.get("EESEG_COOP_STAMPS") and ok:
        ws = K.workspace(0, x.device)
        torch.cuda.synchronize()
        st = ws[256 * 2 * 64 * 4:256 * 2 * 64 * 4 + 128].view(torch.int64).cpu().tolist()
        for blk in (0, 1):
            t = st[blk * 8:blk * 8 + 7]
            print("   block", "first" if blk == 0 else "last ", "stamps (us since start):", [round((v - st[0]) * 0.01, 2) for v in t])
    print(f"rows {rows:7d} C {Cc:5d} mask {mode}: two launches {t2:7.1f} us, one launch {t1:7.1f} us "
          f"({mb:.1f} MB per tensor; 3 passes at 6 TB/s = {3 * mb / 6:.1f} us)", flush=True)
print("timeouts:", K.coop_timeouts())
