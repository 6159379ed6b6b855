"""Diagnostic: per-block wall-clock stamps of the small-M conv_pw_kernel (build with EESEG_EXTRA_FLAGS=-DEESEG_PW_STAMPS).
args: Cin Cout k dil [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ee_semantic_segmentation_amd import kernels as K
Cin, Cout, k, d = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
p = d * (k // 2)
x = torch.randn(B, 65, 65, Cin, device="cuda").bfloat16()
wf, _ = K.pack_weight(torch.randn(Cout, Cin, k, k, device="cuda") * 0.05, torch.bfloat16)
for _ in range(3):
    _, part = K.conv_fwd(x, wf, 1, p, d, want_stats=True)
torch.cuda.synchronize()
ws = K._conv_ws(x.device)
ws.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); K.conv_fwd(x, wf, 1, p, d, want_stats=True); e1.record()
torch.cuda.synchronize()
nblk = part.shape[0] * (Cout // 256)
st = ws.view(torch.int64)[:nblk * 8].view(nblk, 8).cpu().numpy().astype(np.float64) * 0.01      # us
t0 = st[:, 0].min()
names = ["setup+issue", "main loop", "drain+sync", "stage", "readback+stores", "stats"]
print(f"{k}x{k} {Cin}->{Cout} d{d} B={B}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {nblk} blocks, first start -> last end {st[:,6].max()-t0:.1f} us, start spread {st[:,0].max()-t0:.1f}")
for i, n in enumerate(names):
    dd = st[:, i + 1] - st[:, i]
    print(f"  {n:18s} median {np.median(dd):6.2f} us  p90 {np.percentile(dd, 90):6.2f}")
tot = st[:, 6] - st[:, 0]
print(f"  block lifetime     median {np.median(tot):6.2f} us  max {tot.max():.2f}")
