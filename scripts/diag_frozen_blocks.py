"""Frozen-statistics backward, block level + run-to-run determinism of whole-network gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch import nn
from test_model_gpu import _inputs, _pair, _rel
from ee_semantic_segmentation_amd import engine as E
from ee_semantic_segmentation_amd.nn_modules import BatchNorm2d, Bottleneck, Conv2d, DeepLabHead
from oracle.deeplab_ref import Bottleneck as RB, DeepLabHead as RH
DEV = "cuda"
cfg = E.Config()
g = torch.Generator().manual_seed(3)
for mode in sys.argv[1].split(","):
    torch.manual_seed(1)
    rb = RB(64, 64, 1, nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256)), 2)
    for m in rb.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    blk = Bottleneck(64, 64, 1, nn.Sequential(Conv2d(64, 256, 1), BatchNorm2d(256)), 2, cfg=cfg)
    blk.load_state_dict(rb.state_dict())
    blk = blk.to(DEV)
    (rb.train(), blk.train()) if mode == "train" else (rb.eval(), blk.eval())
    x = torch.randn(4, 64, 33, 31, generator=g).requires_grad_(True)
    gy = torch.randn(4, 256, 33, 31, generator=g)
    yr = rb(x); yr.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    out = blk(xd)
    out.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    print(mode, "bottleneck fwd", _rel(out.permute(0, 3, 1, 2), yr), "dx", _rel(xd.grad.permute(0, 3, 1, 2), x.grad))
    rp = dict(rb.named_parameters())
    for k, p in blk.named_parameters():
        print("   ", k, "%.2e" % _rel(p.grad, rp[k].grad))
    torch.manual_seed(2)
    rh = RH(256, 21)
    for m in rh.modules():
        if isinstance(m, nn.Dropout): m.p = 0.0
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    head = DeepLabHead(256, 21, cfg=cfg)
    head.load_state_dict(rh.state_dict())
    head[0].project[3].p = 0.0
    head = head.to(DEV)
    (rh.train(), head.train()) if mode == "train" else (rh.eval(), head.eval())
    x = torch.randn(4, 256, 21, 19, generator=g).requires_grad_(True)
    gy = torch.randn(4, 21, 21, 19, generator=g)
    yr = rh(x); yr.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    lo = head(xd)
    gpad = torch.zeros(4, 21, 19, 32); gpad[..., :21] = gy.permute(0, 2, 3, 1)
    lo.backward(gpad.to(DEV))
    print(mode, "head fwd", _rel(lo[..., :21].permute(0, 3, 1, 2), yr), "dx", _rel(xd.grad.permute(0, 3, 1, 2), x.grad))
    rp = dict(rh.named_parameters())
    for k, p in head.named_parameters():
        print("   ", k, "%.2e" % _rel(p.grad, rp[k].grad))

# determinism: the same whole-network gradient twice
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
for mode in ():
    net, ref = _pair("deeplabv3_resnet50", 2, 97)
    X, y = _inputs(2, 21, 97, 97)
    net.train() if mode == "train" else net.eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    gs = []
    for rep in range(2):
        net.load_state_dict(sd)
        net.zero_grad(set_to_none=True)
        out = net(X.cuda())
        BrXEntropyLoss(ignore_index=21, b_reduction="sum", n_exits=3)(out, y.cuda()).mean().backward()
        gs.append({k: p.grad.clone() for k, p in net.named_parameters()})
    rows = sorted(((_rel(gs[1][k], gs[0][k]), k) for k in gs[0]), reverse=True)
    print(mode, "run-to-run gradient differences (top 6):", [("%.2e" % a, b) for a, b in rows[:6]])
