"""Would ONE launch for the three weight gradients of a bottleneck block pay at the per-GPU shard (4 x 65 x 65)?  A single weight
gradient on 256 / 128 / 64 / 32 concurrent blocks (eeseg_set_wgrad_big_grid): if a problem on a quarter of the chip takes about the
time it takes on the whole chip, four of them side by side in one launch cost one launch.  Graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2:
    lib().eeseg_set_wgrad_group(int(sys.argv[2]))


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


H = W = 65
for cin, cout, k, d in ((1024, 256, 1, 1), (256, 256, 3, 2), (256, 1024, 1, 1), (512, 512, 3, 4), (2048, 512, 1, 1)):
    x = torch.randn(B, H, W, cin, device="cuda").bfloat16()
    dy = torch.randn(B, H, W, cout, device="cuda").bfloat16()
    out = torch.empty(cout, k, k, cin, device="cuda")
    row = []
    for blocks in (256, 128, 64, 32):
        lib().eeseg_set_wgrad_big_grid(blocks, 8)
        t = timed(lambda: K.conv_wgrad(x, dy, k, k, 1, d * (k // 2), d, out=out))
        row.append(f"{blocks:3d} blocks {t:6.1f} us (kernel {lib().eeseg_last_kernel(1)})")
    lib().eeseg_set_wgrad_big_grid(256, 8)
    gf = 2.0 * B * H * W * cin * cout * k * k / 1e9
    print(f"B {B} {k}x{k} {cin}->{cout}: {gf:6.1f} GFLOP | " + " | ".join(row), flush=True)

# the three weight gradients of a layer-3 bottleneck block: one by one vs. one launch (eeseg_conv_wgrad_group)
items = []
for cin, cout, k, d in ((1024, 256, 1, 1), (256, 256, 3, 2), (256, 1024, 1, 1)):
    x = torch.randn(B, H, W, cin, device="cuda").bfloat16()
    dy = torch.randn(B, H, W, cout, device="cuda").bfloat16()
    items.append((x, dy, k, k, 1, d * (k // 2), d, torch.zeros(cout, k, k, cin, device="cuda"), True))
t1 = timed(lambda: [K.conv_wgrad(*it[:7], out=it[7], accumulate=True) for it in items])
t3 = timed(lambda: K.conv_wgrad_group(items))
print(f"B {B} bottleneck block, three weight gradients: one by one {t1:6.1f} us, one launch {t3:6.1f} us (grouped problems: {lib().eeseg_last_kernel(3)})")
