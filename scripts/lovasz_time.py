"""Time the Lovasz loss kernels at a training shape.  args: N C H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
N, C, H, W = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (16, 21, 513, 513)
s = torch.randn(N, C, H, W, device="cuda")
t = torch.randint(0, C + 1, (N, H, W), device="cuda")
for _ in range(2):
    loss, ds = K.lovasz(s, t, C, want_grad=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    loss, ds = K.lovasz(s, t, C, want_grad=True)
e1.record(); torch.cuda.synchronize()
print(f"lovasz fwd+grad {N}x{C}x{H}x{W}: {e0.elapsed_time(e1) / 5:.2f} ms, loss {loss.item():.6f}")
