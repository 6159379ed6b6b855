"""Streaming rates of the BatchNorm kernels at the metric's tensor sizes (graph replay): bytes every kernel must move / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib


def timed(fn, n=10):
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


for kv in os.environ.get("EESEG_PROBE_OPTS", "").split():
    k, v = kv.split("=")
    if k == "cap":
        assert lib().eeseg_set_ew_grid_cap(int(v)) == 0
    else:
        assert lib().eeseg_set_option(int(k), int(v)) == 0, kv
print("options:", os.environ.get("EESEG_PROBE_OPTS", "(defaults)"))
rows = 32 * 65 * 65
for C in (256, 1024):
    x = torch.randn(rows, C, device="cuda").bfloat16()
    dy = torch.randn(rows, C, device="cuda").bfloat16()
    res = torch.randn(rows, C, device="cuda").bfloat16()
    ga = torch.ones(C, device="cuda")
    part = K.channel_stats(x)
    mi, ss = K.bn_finalize(part, rows, ga, torch.zeros(C, device="cuda"), 1e-5, 0.1, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"))
    y, mask = K.bn_apply(x, ss, residual=res, relu=True, want_mask=True)
    y2 = torch.empty_like(x)
    sums = torch.empty(2, C, device="cuda")
    dx = torch.empty_like(x)
    mb = rows * C * 2 / 1e6
    cases = [
        ("channel_stats (1 read)", lambda: K.channel_stats(x), 1.0),
        ("bn_apply relu (1 read, 1 write)", lambda: K.bn_apply(x, ss, relu=True, out=y2), 2.0),
        ("bn_apply + residual + mask (2 reads, 1 write)", lambda: K.bn_apply(x, ss, residual=res, relu=True, want_mask=True), 3.0 + 1 / 16),
        ("bn_bwd_reduce, relu via scale_shift (2 reads)", lambda: K.bn_bwd_reduce(dy, None, x, mi, True, out=sums, scale_shift=ss), 2.0),
        ("bn_bwd_reduce, 1-bit mask (2 reads)", lambda: K.bn_bwd_reduce(dy, mask, x, mi, True, out=sums), 2.0 + 1 / 16),
        ("bn_bwd_apply, relu via scale_shift (2 reads, 1 write)", lambda: K.bn_bwd_apply(dy, None, x, mi, ga, sums, rows, True, dx=dx, scale_shift=ss), 3.0),
        ("bn_bwd_apply, 1-bit mask (2 reads, 1 write)", lambda: K.bn_bwd_apply(dy, mask, x, mi, ga, sums, rows, True, dx=dx), 3.0 + 1 / 16),
    ]
    for name, fn, passes in cases:
        try:
            t = timed(fn)
        except Exception as e:      # a signature this script guessed wrong must not hide the other rows
            print(f"C {C:5d} {name:55s}: {type(e).__name__}: {str(e)[:80]}")
            continue
        print(f"C {C:5d} {name:55s}: {t:7.1f} us  {passes * mb / t / 1e3 * 1e3:5.2f} TB/s", flush=True)
