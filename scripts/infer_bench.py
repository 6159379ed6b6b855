"""Inference-only early-exit benchmark (BASELINE.json configs[3] shape): R101, 4 exits, 1024x2048, B=1, bf16,
fused entropy gate on every branch + argmax of the chosen exit; all exits computed (like eval_br_ent.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
C = 19
torch.manual_seed(0)
net = branchyDeepv3(None, "deeplabv3_resnet101", 3, 1024, count_branches=False, num_classes=C,
                    compute_dtype=torch.bfloat16).cuda().eval()
X = torch.randn(1, 3, H, W, device="cuda")
def step():
    with torch.no_grad():
        lrs = net.forward_lowres(X)
        flags = [K.entropy_gate(lr, C, H, W, 0.5)[1] for lr in lrs[:-1]]
        preds = [K.argmax_confusion(lr, C, None, H, W, want_pred=True)[1] for lr in lrs]
    return flags, preds
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
macs = net.macs(H, W)
print(f"R101 4 exits {H}x{W} B=1 bf16 eval (all exits + 3 gates + 4 argmax): {dt*1e3:.2f} ms/img, {1/dt:.1f} img/s, "
      f"{2*macs/dt/1e12:.0f} TFLOP/s algorithmic ({macs/1e9:.1f} GMAC/img), splits {net.split_names}")
