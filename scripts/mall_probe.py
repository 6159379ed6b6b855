"""Does re-reading a tensor that was just read come cheaper when it fits the 256-MB Infinity Cache?  Column statistics (one streamed read)
of [135200, C] bf16, the same tensor again and again (graph replay), for tensors of 17 MB .. 554 MB; and bn_bwd_stage1 + bn_bwd_apply
back to back (the two passes of BatchNorm backward) per size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from pws_bench import timed  # noqa (prints its table once)

rows = 32 * 65 * 65
for C in (64, 128, 256, 512, 1024, 2048):
    x = torch.randn(rows, C, device="cuda").bfloat16()
    mb = rows * C * 2 / 1e6
    t = timed(lambda: K.channel_stats(x))
    dy = torch.randn(rows, C, device="cuda").bfloat16()
    ga = torch.ones(C, device="cuda")
    mi, ss = K.bn_finalize(K.channel_stats(x), rows, ga, torch.zeros(C, device="cuda"), 1e-5, 0.1, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"))
    sums = torch.empty(2, C, device="cuda")
    dx = torch.empty_like(x)

    def two():
        K.bn_bwd_reduce(dy, None, x, mi, False, out=sums)
        K.bn_bwd_apply(dy, None, x, mi, ga, sums, rows, False, dx=dx)
    t2 = timed(two)
    print(f"C {C:5d} ({mb:6.1f} MB): stats re-read {t:7.1f} us = {mb / t / 1e3 * 1e3:5.2f} TB/s | BN backward (2 + 3 passes over {mb:.0f} MB) {t2:7.1f} us = {5 * mb / t2 / 1e3 * 1e3:5.2f} TB/s", flush=True)
