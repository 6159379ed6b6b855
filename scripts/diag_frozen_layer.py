import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch import nn
from test_model_gpu import _rel
from ee_semantic_segmentation_amd import engine as E
from ee_semantic_segmentation_amd.nn_modules import BatchNorm2d, Conv2d
cfg = E.Config()
g = torch.Generator().manual_seed(3)
for (cin, cout, k, d, frozen) in [(1280, 256, 1, 1, True), (256, 256, 3, 1, True), (256, 256, 3, 12, True), (1280, 256, 1, 1, False)]:
    torch.manual_seed(1)
    rc, rb = nn.Conv2d(cin, cout, k, padding=d * (k // 2), dilation=d, bias=False), nn.BatchNorm2d(cout)
    rb.running_mean.normal_(0, 0.3); rb.running_var.uniform_(0.5, 2.0); rb.weight.data.uniform_(0.5, 1.5); rb.bias.data.normal_(0, 0.2)
    conv, bn = Conv2d(cin, cout, k, padding=d * (k // 2), dilation=d), BatchNorm2d(cout)
    conv.load_state_dict(rc.state_dict()); bn.load_state_dict(rb.state_dict())
    conv, bn = conv.cuda(), bn.cuda()
    (rb.eval() if frozen else rb.train())
    x = torch.randn(4, cin, 21, 19, generator=g).requires_grad_(True)
    gy = torch.randn(4, cout, 21, 19, generator=g)
    yr = torch.relu(rb(rc(x))); yr.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    y, st = E.conv_bn_fwd(cfg, xd, conv, bn, True, frozen=frozen)
    dx, _, dw, dgm, dbt = E.conv_bn_bwd(cfg, st, gy.permute(0, 2, 3, 1).contiguous().cuda(), conv, bn)
    print((cin, cout, k, d, frozen), "fwd %.1e dx %.1e dw %.1e dgamma %.1e dbeta %.1e" % (
        _rel(y.permute(0, 3, 1, 2), yr), _rel(dx.permute(0, 3, 1, 2), x.grad), _rel(dw, rc.weight.grad),
        _rel(dgm, rb.weight.grad), _rel(dbt, rb.bias.grad)))
