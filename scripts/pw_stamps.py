"""Diagnostic: per-block wall-clock stamps of conv_pw_kernel (build with EESEG_EXTRA_FLAGS=-DEESEG_PW_STAMPS).
args: Cin Cout [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ee_semantic_segmentation_amd import kernels as K
Cin, Cout = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
res = len(sys.argv) > 4
x = torch.randn(B, 65, 65, Cin, device="cuda").bfloat16()
wf, _ = K.pack_weight(torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05, torch.bfloat16)
r = torch.randn(B, 65, 65, Cout, device="cuda").bfloat16() if res else None
for _ in range(3):
    K.conv_fwd(x, wf, want_stats=not res, residual=r)
torch.cuda.synchronize()
ws = K._conv_ws(x.device)
ws.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); K.conv_fwd(x, wf, want_stats=not res, residual=r); e1.record()
torch.cuda.synchronize()
nblk = ((B * 65 * 65 + 127) // 128) * (Cout // 256)
st = ws.view(torch.int64)[:nblk * 8].view(nblk, 8).cpu().numpy().astype(np.float64) * 0.01      # us
t0 = st[:, 0].min()
names = ["setup+issue", "main loop", "drain+sync", "stage", "readback+stores", "stats"]
print(f"{Cin}->{Cout} B={B} res={res}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {nblk} blocks, span {st[:,6].max()-t0:.1f} us")
for i, n in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {n:18s} median {np.median(d):6.2f} us  p90 {np.percentile(d, 90):6.2f}")
tot = st[:, 6] - st[:, 0]
print(f"  block lifetime     median {np.median(tot):6.2f} us; blocks alive at once (sum lifetime / span): {tot.sum() / (st[:,6].max()-t0):.0f}")
