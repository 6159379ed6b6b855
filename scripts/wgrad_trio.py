"""The three weight gradients of a layer-3 bottleneck (1x1 1024->256, 3x3 256->256, 1x1 256->1024 at B x 65 x 65): one after the other on
the whole chip (64 / 28 / 64 K splits each = 3 x 64 MB of fp32 atomics) against side by side on three streams with grids sized by their
work (60 / 135 / 60 blocks: 15 splits each, 64 MB of atomics in all).  usage: python scripts/wgrad_trio.py [B]

Measured (B=32): serial 402-420 us per block, side by side 310-335 us for every split of <= 254 blocks (255 blocks: 484-495 us, the one
workgroup that does not fit waits for a whole round).  Wired into the training step (the three gradients of a block collected and launched
together at its end) the STEP got 1 ms slower (102.3 -> 103.3-103.7 ms): held back to the end of the block, the gradients read their operands
cold, and the step pays three stream forks / joins per block.  Not adopted; kept as the record of the experiment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
def rnd(*s): return (torch.randn(*s, device=dev) * 0.1).bfloat16()
x1, dy1 = rnd(B, 65, 65, 1024), rnd(B, 65, 65, 256)
x2, dy2 = rnd(B, 65, 65, 256), rnd(B, 65, 65, 256)
x3, dy3 = rnd(B, 65, 65, 256), rnd(B, 65, 65, 1024)
o1 = torch.zeros(256, 1, 1, 1024, device=dev); o2 = torch.zeros(256, 3, 3, 256, device=dev); o3 = torch.zeros(1024, 1, 1, 256, device=dev)
calls = [(lambda: K.conv_wgrad(x1, dy1, 1, 1, 1, 0, 1, out=o1, accumulate=True), 60),
         (lambda: K.conv_wgrad(x2, dy2, 3, 3, 1, 2, 2, out=o2, accumulate=True), 135),
         (lambda: K.conv_wgrad(x3, dy3, 1, 1, 1, 0, 1, out=o3, accumulate=True), 60)]
streams = [torch.cuda.Stream() for _ in range(3)]

def serial():
    lib().eeseg_set_wgrad_big_grid(256, 8)
    for fn, _ in calls:
        fn()

def side_by_side(grids):
    cur = torch.cuda.current_stream()
    for (fn, _), g, st in zip(calls, grids, streams):
        st.wait_stream(cur)
        lib().eeseg_set_wgrad_big_grid(g, 8)
        with torch.cuda.stream(st):
            fn()
    for st in streams:
        cur.wait_stream(st)
    lib().eeseg_set_wgrad_big_grid(256, 8)

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best

lib().eeseg_set_wgrad_big(2)
print(f"B={B}: serial, whole chip each: {timeit(serial):7.1f} us per bottleneck")
for grids in ((60, 135, 60), (64, 126, 64), (60, 126, 60), (56, 135, 56), (60, 135, 56), (64, 117, 64), (52, 144, 52), (64, 126, 60), (56, 126, 56)):
    print(f"      three streams, grids {grids}: {timeit(lambda: side_by_side(grids)):7.1f} us")
lib().eeseg_set_wgrad_big(1)
