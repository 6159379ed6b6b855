"""Where does a 256x256 conv tile spend its cycles?  (debug stamps; GPU)  args: H W Cin Cout k s p d B"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
H, W, Cin, Cout, k, s, p, d, B = [int(v) for v in sys.argv[1:10]]
x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
wt = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
wf, wb = K.pack_weight(wt, torch.bfloat16)
lib().eeseg_set_option(5, 0)      # no K-split tail: one launch
for _ in range(3):
    K.conv_fwd(x, wf, s, p, d, want_stats=True)
st = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
f = lib().eeseg_debug_set_stamps
f.argtypes = [C.c_void_p]; f.restype = C.c_int
assert f(st.data_ptr()) == 0
K.conv_fwd(x, wf, s, p, d, want_stats=True)
torch.cuda.synchronize()
f(None)
t = st.view(-1, 8).cpu()
t = t[t[:, 0] > 0].double()
names = ["prologue(index math, tap mask)", "pipeline fill (2 K tiles)", "main loop", "acc->LDS staging", "read back + stores", "total to stats"]
d = [(t[:, i + 1] - t[:, i]) for i in range(5)] + [t[:, 5] - t[:, 0]]
print(f"{t.shape[0]} blocks; cycles (median / mean):")
for n, v in zip(names, d):
    print(f"  {n:34s} {v.median().item():9.0f} {v.mean().item():9.0f}")
t0 = t[:, 0].min()
order = t[:, 0].argsort()
starts = (t[order, 0] - t0)
print("block start offsets (cycles) percentiles:", [int(starts[int(q * (len(starts) - 1))].item()) for q in (0, 0.24, 0.26, 0.5, 0.75, 1.0)])
print("kernel span (cycles):", int((t[:, 5].max() - t0).item()))
