"""Per-tensor gradient errors of my_branch(bottleneck=b) against the oracle module.  usage: diag_mybranch.py [b]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from ee_semantic_segmentation_amd import engine as E
from ee_semantic_segmentation_amd.from_deepv3_new import my_branch
from oracle.deeplab_ref import my_branch as RM

b = int(sys.argv[1]) if len(sys.argv) > 1 else 100
DEV = torch.device("cuda")
def rel(a, c):
    a, c = a.detach().float().cpu(), c.detach().float().cpu()
    return (a - c).abs().max().item() / (c.abs().max().item() + 1e-12)
cfg = E.Config()
g = torch.Generator().manual_seed(4)
torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
params = dict(atrous_rates=[2, 4], nout_channels=128, bottleneck=b)
rh = RM(256, 21, **params).train()
for m in rh.modules():
    if isinstance(m, nn.Dropout):
        m.p = 0.0
head = my_branch(256, 21, cfg=cfg, **params)
head.load_state_dict(rh.state_dict())
head.aspp.project[3].p = 0.0
head = head.to(DEV).train()
x = torch.randn(4, 256, 21, 19, generator=g).requires_grad_(True)
gy = torch.randn(4, 21, 21, 19, generator=g)
feats = {}
def _keep(m, i, o):
    o.retain_grad()
    feats["pre"] = o
rh[0].register_forward_hook(_keep)
yr = rh(x)
yr.backward(gy)
xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
lo = head(xd)
gpad = torch.zeros(4, 21, 19, 32)
gpad[..., :21] = gy.permute(0, 2, 3, 1)
lo.backward(gpad.to(DEV))
print("fwd", rel(lo[..., :21].permute(0, 3, 1, 2), yr), "dx", rel(xd.grad.permute(0, 3, 1, 2), x.grad))
rp = dict(rh.named_parameters())
for k, p in head.named_parameters():
    gg = p.grad.cpu()
    true = tuple(slice(0, n) for n in rp[k].shape)
    rest = gg.clone(); rest[true] = 0
    if rel(gg[true], rp[k].grad) > 1e-4:
        print(f"{k:28s} {tuple(gg.shape)} rel {rel(gg[true], rp[k].grad):.2e} pad max {float(rest.abs().max()):.2e}")
# the pre conv's data gradient alone, from the oracle's d(pre output)
from ee_semantic_segmentation_amd import kernels as K
dpre = feats["pre"].grad.permute(0, 2, 3, 1).contiguous()
dpad = torch.zeros(4, 21, 19, head.pre.cout_stored); dpad[..., :b] = dpre
_, wb = E.packed(head.pre, torch.float32)
dx = K.conv_dgrad(dpad.to(DEV), wb, (21, 19), 1, 0, 1)
print("pre dgrad alone", rel(dx.permute(0, 3, 1, 2), x.grad))
