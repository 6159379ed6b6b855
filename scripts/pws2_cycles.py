"""diagnostic (EESEG_EXTRA_FLAGS="-DEESEG_PW_STAMPS -DEESEG_PW_CYCLES"): where the rounds of conv_pws2_kernel spend their time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 65, 65, 256, device="cuda").bfloat16()
wf, _ = K.pack_weight(torch.randn(1024, 256, 1, 1, device="cuda") / 16, torch.bfloat16)
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib().eeseg_set_option(14, MODE)
for _ in range(3):
    K.conv_fwd(x, wf, want_stats=True)
torch.cuda.synchronize()
ws = K._conv_ws(x.device)
st = ws.view(torch.int64)[8192:8192 + 7 * 2 * 8].view(7, 2, 8).cpu().numpy().astype(np.int64)
for b in range(7):
    m, o = st[b, 0], st[b, 1]
    n = max(int(m[4]), 1)
    print(f"block {37 * b:3d} ({n} sub-tiles): MFMA wave per round: vmcnt wait {m[1] / n:5.0f}  barrier {m[2] / n:5.0f}  DMA issue {m[3] / n:5.0f}  reads+MFMA {m[5] / n:5.0f}  pack+stage {m[6] / n:5.0f}  loop {m[0] / n:4.0f}"
          f"  | output wave: barrier {o[1] / n:6.0f}  staging read {o[2] / n:6.0f}  relu+stores {o[3] / n:6.0f}  stats {o[0] / n:6.0f}   (shader cycles)")
