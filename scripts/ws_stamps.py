"""Diagnostic: wall-clock stamps of conv_pws_kernel's second tile per block (build with EESEG_EXTRA_FLAGS=-DEESEG_PW_STAMPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ee_semantic_segmentation_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
res = len(sys.argv) > 2
x = torch.randn(B, 65, 65, 256, device="cuda").bfloat16()
wf, _ = K.pack_weight(torch.randn(1024, 256, 1, 1, device="cuda") * 0.05, torch.bfloat16)
r = torch.randn(B, 65, 65, 1024, device="cuda").bfloat16() if res else None
for _ in range(3):
    K.conv_fwd(x, wf, want_stats=not res, residual=r)
ws = K._conv_ws(x.device); ws.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); K.conv_fwd(x, wf, want_stats=not res, residual=r); e1.record(); torch.cuda.synchronize()
st = ws.view(torch.int64)[:512 * 16].view(512, 16).cpu().numpy().astype(np.float64) * 0.01
t0 = st[:, 0].min()
print(f"256->1024 B={B} res={res}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, span {st[:,15].max()-t0:.1f} us; block lifetime median {np.median(st[:,15]-st[:,0]):.1f}")
print(f"  W load               median {np.median(st[:,1]-st[:,0]):6.2f}")
names = ["wait X + barrier", "64 MFMAs (2 passes)", "barrier + stage", "read back, residual, DMA issue, stores", "stats"]
for h in range(2):
    for i, n in enumerate(names):
        d = st[:, 3 + h * 6 + i] - st[:, 2 + h * 6 + i]
        print(f"  half {h} {n:40s} median {np.median(d):6.2f} p90 {np.percentile(d, 90):6.2f}")
print(f"  tile total median {np.median(st[:,13]-st[:,2]):.2f}")
