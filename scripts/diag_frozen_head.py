import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch import nn
from test_model_gpu import _rel
from ee_semantic_segmentation_amd import engine as E
from ee_semantic_segmentation_amd.nn_modules import DeepLabHead
from oracle.deeplab_ref import DeepLabHead as RH
cfg = E.Config()
g = torch.Generator().manual_seed(3)
torch.manual_seed(2)
rh = RH(256, 21)
for m in rh.modules():
    if isinstance(m, nn.Dropout): m.p = 0.0
    if isinstance(m, nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
head = DeepLabHead(256, 21, cfg=cfg)
head.load_state_dict(rh.state_dict())
head[0].project[3].p = 0.0
head = head.cuda()
rh.eval(); head.eval()
grads = {}
def hook(name):
    def f(mod, gin, gout):
        grads[name] = gout[0].detach().clone()
    return f
rh[0].project.register_full_backward_hook(hook("project_out"))
rh[0].project[1].register_full_backward_hook(hook("project_bn_out"))
rh[0].project[0].register_full_backward_hook(hook("project_conv_out"))
for i in range(5):
    rh[0].convs[i].register_full_backward_hook(hook(f"branch{i}_out"))
rh[2].register_full_backward_hook(hook("bn3_out"))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0):       # advance the generator like diag_frozen_blocks.py
    torch.randn(4, 64, 33, 31, generator=g); torch.randn(4, 256, 33, 31, generator=g)
    torch.randn(4, 256, 21, 19, generator=g); torch.randn(4, 21, 21, 19, generator=g)
torch.randn(4, 64, 33, 31, generator=g); torch.randn(4, 256, 33, 31, generator=g)
x = torch.randn(4, 256, 21, 19, generator=g).requires_grad_(True)
gy = torch.randn(4, 21, 21, 19, generator=g)
yr = rh(x); yr.backward(gy)
rec = []
orig = E.conv_bn_bwd
def spy(cfg_, st, dy, conv, bn, **kw):
    out = orig(cfg_, st, dy, conv, bn, **kw)
    rec.append((tuple(conv.weight.shape), dy.detach().clone(), None if out[0] is None else out[0].detach().clone(), st[1].detach().clone()))
    return out
E.conv_bn_bwd = spy
xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
lo = head(xd)
gpad = torch.zeros(4, 21, 19, 32); gpad[..., :21] = gy.permute(0, 2, 3, 1)
lo.backward(gpad.cuda())
names = ["bn3_out", "project_bn_out", "branch0_out", "branch1_out", "branch2_out", "branch3_out", "branch4_out"]
for (shape, dy, dx, c), n in zip(rec, names):
    ref = grads[n]
    got = dy.permute(0, 3, 1, 2) if dy.dim() == 4 else dy
    if ref.shape != got.shape:
        print(n, shape, "shapes", tuple(ref.shape), tuple(got.shape)); continue
    print(n, shape, "dy rel err %.2e" % _rel(got, ref))
# project conv output grad (dc of project) vs oracle
print("project dc:", _rel(rec[1][2].permute(0, 3, 1, 2)[:, :256] if False else rec[1][1].permute(0,3,1,2), grads["project_bn_out"]))
rp = dict(rh.named_parameters())
print("dx", _rel(xd.grad.permute(0, 3, 1, 2), x.grad))
for k, p in head.named_parameters():
    print("   ", k, "%.2e" % _rel(p.grad, rp[k].grad))
dpr = rec[0][2].permute(0, 3, 1, 2).cpu()
ref = grads["project_out"]
err = (dpr - ref).abs()
print("dpr (conv3 dgrad out) vs oracle: rel max", (err.max() / ref.abs().max()).item(), "n bad", int((err > 1e-4 * ref.abs().max()).sum()), "of", err.numel())
bad = (err > 1e-4 * ref.abs().max()).nonzero()
print(bad[:10].tolist())
# project pre-activation from HIP state c vs oracle
c_proj = rec[1][3].permute(0, 3, 1, 2).cpu()
with torch.no_grad():
    cat_ref = torch.cat([m(x) for m in rh[0].convs], dim=1)
    c_ref = rh[0].project[0](cat_ref)
    z_ref = rh[0].project[1](c_ref)
print("project conv out c: rel", _rel(c_proj, c_ref), " near-zero preacts (|z|<1e-5):", int((z_ref.abs() < 1e-5).sum()))
dcat = rec[1][2].permute(0, 3, 1, 2).cpu()
e2 = (dcat[:, :256] - grads["branch0_out"]).abs()
b2 = (e2 > 1e-3 * grads["branch0_out"].abs().max()).nonzero()
print("branch0 dy bad count", b2.shape[0], "pixels:", sorted(set((int(a), int(c), int(d)) for a, _, c, d in b2.tolist()))[:12])
