import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from test_model_gpu import _pair, _inputs
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
from ee_semantic_segmentation_amd.optim import SGD
from ee_semantic_segmentation_amd.parallel import GraphedTrainStep
C, B, img, steps = 21, 8, 129, 6
X, y = _inputs(B, C, img, img); Xd, yd = X.cuda(), y.cuda()
crit = BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=2)
for mode in ("eager", "arena-eager", "graph-w2", "graph-w4"):
    net, _ = _pair("deeplabv3_resnet50", 1, img); net.train(); net.fused_outputs = True
    opt = SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    losses = []
    if mode == "eager":
        for _ in range(steps):
            l = crit(net(Xd), yd); opt.zero_grad(); l.mean().backward(); opt.step(); losses.append(l.item())
    else:
        net.enable_grad_arena()
        runner = GraphedTrainStep(net, crit, opt, warmup={"arena-eager": 99, "graph-w2": 2, "graph-w4": 4}[mode])
        for _ in range(steps):
            losses.append(float(runner(Xd, yd).item()))
    print(mode, [round(v, 5) for v in losses])
