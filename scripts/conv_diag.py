"""diagnostic build only (-DEESEG_PW_DIAG): time the 3x3 256->256 small-M conv with parts of its K loop removed"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ee_semantic_segmentation_amd import kernels as K
B = 4
x = torch.randn(B, 65, 65, 256, device="cuda").to(torch.bfloat16)
w = torch.randn(256, 256, 3, 3, device="cuda") * 0.05
wf, wb = K.pack_weight(w, torch.bfloat16)
def timed(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for diag, what in [(0, "full"), (1, "no MFMA"), (2, "no DMA"), (4, "no frag reads"), (3, "no MFMA, no DMA"), (5, "no MFMA, no reads"), (6, "no DMA, no reads"), (7, "barriers only")]:
    os.environ["EESEG_PW_DIAG"] = str(diag)
    print(f"diag {diag} ({what}): {timed(lambda: K.conv_fwd(x, wf, 1, 2, 2, want_stats=True)):.1f} us", flush=True)
