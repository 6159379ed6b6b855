"""Run one conv shape a few times (for rocprofv3 --pmc).  args: H W Cin Cout k s p d B mode(fwd|dgrad|wgrad) iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
H, W, Cin, Cout, k, s, p, d, B = [int(v) for v in sys.argv[1:10]]
mode = sys.argv[10] if len(sys.argv) > 10 else "fwd"
iters = int(sys.argv[11]) if len(sys.argv) > 11 else 3
dtype = torch.bfloat16
if os.environ.get("EESEG_CONV_PIPE"):
    from ee_semantic_segmentation_amd._lib import lib
    lib().eeseg_set_option(1, int(os.environ["EESEG_CONV_PIPE"]))
    if os.environ.get("EESEG_CONV_TAIL_MIN"):
        lib().eeseg_set_option(5, int(os.environ["EESEG_CONV_TAIL_MIN"]))
if os.environ.get("EESEG_OPTS"):            # "key=value,key=value" -> eeseg_set_option (A/B of kernel forms under the counters)
    from ee_semantic_segmentation_amd._lib import lib
    for kv in os.environ["EESEG_OPTS"].split(","):
        a, b = kv.split("=")
        lib().eeseg_set_option(int(a), int(b))
x = torch.randn(B, H, W, Cin, device="cuda").to(dtype)
wt = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
wf, wb = K.pack_weight(wt, dtype)
y, _ = K.conv_fwd(x, wf, s, p, d, want_stats=True)
gy = torch.randn_like(y)
for _ in range(iters):
    if mode == "fwd":
        K.conv_fwd(x, wf, s, p, d, want_stats=True)
    elif mode == "dgrad":
        K.conv_dgrad(gy, wb, (H, W), s, p, d)
    else:
        K.conv_wgrad(x, gy, k, k, s, p, d)
torch.cuda.synchronize()
