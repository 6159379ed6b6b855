"""Does the weight-stationary pointwise layer's time depend on HOW its output rows lie in memory?  Same 256 -> 1024 layer, output written
contiguously (2 KiB rows, each 512-B quarter by another block), into a slice of a 4096-wide buffer (8 KiB pitch), and the 256 -> 256 layer
with a residual (whole 512-B rows per block)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pws_bench import timed  # noqa: E402  (runs the bench table once on import - fine)

B, H, W = 32, 65, 65
M = B * H * W
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, 256, generator=g).to("cuda").bfloat16()
wf, _ = K.pack_weight((torch.randn(1024, 256, 1, 1, generator=g) / 16).to("cuda"), torch.bfloat16)
wide = torch.empty(B, H, W, 4096, device="cuda", dtype=torch.bfloat16)
y = torch.empty(B, H, W, 1024, device="cuda", dtype=torch.bfloat16)
_, wb = K.pack_weight((torch.randn(256, 256, 1, 1, generator=g) / 16).to("cuda"), torch.bfloat16)
acc = torch.randn(B, H, W, 256, generator=g).to("cuda").bfloat16()
for mode in (1, 2):
    lib().eeseg_set_option(14, mode)
    t0 = timed(lambda: K.conv_fwd(x, wf, out=y))
    t1 = timed(lambda: K.conv_fwd(x, wf, out=wide[..., 1024:2048]))
    t2 = timed(lambda: K.conv_dgrad(x, wb, (H, W), accumulate_into=acc))
    k2 = lib().eeseg_last_kernel(0)
    print(f"option 14 = {mode}: contiguous [M,1024] {t0:6.1f} us ({(M * 1280 * 2) / t0 / 1e6:.2f} TB/s) | slice of [M,4096] {t1:6.1f} us | "
          f"256->256 + residual (kernel {k2}) {t2:6.1f} us ({(M * 768 * 2) / t2 / 1e6:.2f} TB/s)", flush=True)
lib().eeseg_set_option(14, 1)
