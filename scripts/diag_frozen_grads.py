"""Whole-network gradients on frozen (calibrated) BatchNorm statistics: HIP fp32 vs the CPU oracle, per parameter:
max-abs relative error, relative L2 error and cosine (a ReLU whose pre-activation is within rounding of zero may take
a different side in the two implementations: one such pixel shifts every gradient upstream of it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from test_model_gpu import _inputs, _pair, _rel
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
from oracle import losses_ref
img = int(sys.argv[1]) if len(sys.argv) > 1 else 97
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = sys.argv[3] if len(sys.argv) > 3 else "eval"
C, n = 21, 2
net, ref = _pair("deeplabv3_resnet50", n, img)
X, y = _inputs(B, C, img, img)
ref.train()
for m in ref.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.momentum = 1.0
with torch.no_grad():
    ref(X)
net.load_state_dict(ref.state_dict())
if mode == "eval":
    ref.eval(); net.eval()
else:
    net.train()
out_ref = ref(X)
losses_ref.br_xentropy(out_ref, y, ignore_index=C, b_reduction="sum", n_exits=n + 1).mean().backward()
out = net(X.cuda())
print(mode, img, B, "logit err", (out.detach().cpu() - out_ref.detach()).abs().max().item(), "max |logit|", out_ref.abs().max().item())
BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=n + 1)(out, y.cuda()).mean().backward()
rp = dict(ref.named_parameters())
mx, l2, cs = [], [], []
for k, p in net.named_parameters():
    a, b = p.grad.detach().double().cpu().reshape(-1), rp[k].grad.double().reshape(-1)
    mx.append(((a - b).abs().max() / b.abs().max()).item())
    l2.append(((a - b).norm() / b.norm()).item())
    cs.append(float(a @ b / (a.norm() * b.norm())))
for name, v in (("max-rel", mx), ("rel-L2", l2), ("1-cos", [1 - c for c in cs])):
    v = np.sort(np.array(v))
    print("%-8s median %.2e  p90 %.2e  max %.2e" % (name, v[len(v) // 2], v[int(0.9 * len(v))], v[-1]))
