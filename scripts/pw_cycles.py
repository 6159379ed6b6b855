"""diagnostic (-DEESEG_PW_STAMPS -DEESEG_PW_CYCLES): shader-clock stamps inside the small-M K loop, block 0, K tiles 16..19"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ee_semantic_segmentation_amd import kernels as K
x = torch.randn(4, 65, 65, 256, device="cuda").bfloat16()
wf, _ = K.pack_weight(torch.randn(256, 256, 3, 3, device="cuda") * 0.05, torch.bfloat16)
for _ in range(3):
    K.conv_fwd(x, wf, 1, 2, 2, want_stats=True)
torch.cuda.synchronize()
ws = K._conv_ws(x.device)
st = ws.view(torch.int64)[8192:8192 + 4 * 4 * 8].view(4, 4, 8).cpu().numpy().astype(np.int64)
for t in range(4):
    for w in range(4):
        r = st[t, w]
        print(f"tile {16 + t} wave {w}: vmcnt wait {r[1]-r[0]:5d}  lgkm wait {r[2]-r[1]:5d}  barrier {r[3]-r[2]:5d}  mfma block {r[4]-r[3]:5d}"
              + (f"  | tile period {st[t+1, w, 0]-r[0]:5d}" if t < 3 else ""))
