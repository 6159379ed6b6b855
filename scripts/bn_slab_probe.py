"""Does BatchNorm backward run faster channel slab by channel slab?  Statistics are per channel, so reduce + apply of a
256..512-channel slab touch 70-140 MB and the apply pass may find dy / c in the 256-MB Infinity Cache.
usage: python scripts/bn_slab_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (N, H, W, C) in [(16, 65, 65, 2048), (16, 65, 65, 1024), (16, 129, 129, 256), (16, 65, 65, 512)]:
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    dy = torch.randn(N, H, W, C, device="cuda").bfloat16()
    gamma = torch.rand(C, device="cuda") + 0.5
    cnt = N * H * W
    sums = K.channel_stats(x)
    mi, ss = K.bn_finalize(sums, cnt, gamma, torch.zeros(C, device="cuda"), 1e-5, 0.1, None, None)
    dx = torch.empty_like(x)

    def full():
        bs = K.bn_bwd_reduce(dy, None, x, mi, True, scale_shift=ss)
        K.bn_bwd_apply(dy, None, x, mi, gamma, bs, cnt, True, dx=dx, scale_shift=ss)

    res = {"full": min(timeit(full) for _ in range(3))}
    for cs in (256, 512, 1024):
        if cs >= C:
            continue
        parts = []
        for s in range(0, C, cs):
            sl = slice(s, s + cs)
            parts.append((dy[..., sl], x[..., sl], mi[:, sl].contiguous(), ss[:, sl].contiguous(), gamma[sl].contiguous(), dx[..., sl]))

        def slabs():
            for d_, x_, mi_, ss_, g_, dx_ in parts:
                bs = K.bn_bwd_reduce(d_, None, x_, mi_, True, scale_shift=ss_)
                K.bn_bwd_apply(d_, None, x_, mi_, g_, bs, cnt, True, dx=dx_, scale_shift=ss_)

        res[f"slabs of {cs}"] = min(timeit(slabs) for _ in range(3))
    mb = x.numel() * 2 / 1e6
    print(f"[{N},{H},{W},{C}] ({mb:.0f} MB per tensor): " + "  ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)
