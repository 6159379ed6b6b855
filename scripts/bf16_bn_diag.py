"""with the CONSTANT-learning-rate protocol that once gave final-exit mIoU bf16 0.809 vs fp32 0.917
(gpurun_out/r3/gpu_all2.log), compare the BatchNorm running statistics of the bf16 network with the fp32 network's, layer
by layer, against the fp32-vs-fp32 (other summation order) yardstick, and score the final exit in eval() AND in train() mode.
    python scripts/bf16_bn_diag.py [steps] [seeds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_configs_gpu import _inputs  # noqa: E402
from ee_semantic_segmentation_amd._lib import lib  # noqa: E402
from ee_semantic_segmentation_amd.eval_mIoU import mIoU_evaluator  # noqa: E402
from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3  # noqa: E402
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss  # noqa: E402
from ee_semantic_segmentation_amd.optim import SGD  # noqa: E402

DEV = "cuda"
K_STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 160
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
C, B, img = 19, 16, 129


def train(dt, colreduce, seed):
    torch.manual_seed(seed)
    lib().eeseg_set_option(11, colreduce)
    try:
        net = branchyDeepv3(None, "deeplabv3_resnet50", 1, img, count_branches=False, num_classes=C, compute_dtype=dt,
                            fused_outputs=True).to(DEV).train()
        for m in net.modules():
            if type(m).__name__ == "Dropout":
                m.p = 0.0
        crit = BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=2)
        opt = SGD(net.parameters(), lr=0.02, momentum=0.9, weight_decay=5e-4)
        for k in range(K_STEPS):
            l = crit(net(Xd), yd)
            opt.zero_grad()
            l.mean().backward()
            opt.step()
    finally:
        lib().eeseg_set_option(11, 512)
    stats = {n: (m.running_mean.detach().float().cpu().clone(), m.running_var.detach().float().cpu().clone())
             for n, m in net.named_modules() if type(m).__name__ == "BatchNorm2d"}
    net.eval()
    m_eval = mIoU_evaluator(net, 2, C, [(X, y)], DEV, nan_safe=True)
    net.train()                                   # batch statistics of these very images (the running statistics move: read above)
    mt = mIoU_evaluator(net, 2, C, [(X, y)], DEV, nan_safe=True)
    return float(l.item()), stats, m_eval, mt


for seed in range(SEEDS):
    X, y = _inputs(B, C, img, img, seed=77 + seed, block=33)
    Xd, yd = X.to(DEV), y.to(DEV)
    runs = {name: train(dt, cr, seed) for name, dt, cr in (("f32", torch.float32, 512), ("f32o", torch.float32, 0), ("bf16", torch.bfloat16, 512))}
    print(f"== seed {seed}, {K_STEPS} constant-LR steps: final loss " + " / ".join(f"{n} {r[0]:.4f}" for n, r in runs.items()))
    for n, r in runs.items():
        print(f"   {n:5s} eval() mIoU {r[2]}   train() mIoU {r[3]}")
    ref = runs["f32"][1]
    print("   layer                                  |rm16-rm32|/|rm32|  yardstick   |rv16-rv32|/|rv32|  yardstick")
    worst = (0.0, "")
    for name in ref:
        def rel(a, b):
            return float((a - b).norm() / (b.norm() + 1e-12))
        dm16, dmo = rel(runs["bf16"][1][name][0], ref[name][0]), rel(runs["f32o"][1][name][0], ref[name][0])
        dv16, dvo = rel(runs["bf16"][1][name][1], ref[name][1]), rel(runs["f32o"][1][name][1], ref[name][1])
        ratio = max(dm16 / (dmo + 1e-3), dv16 / (dvo + 1e-3))
        if ratio > worst[0]:
            worst = (ratio, name)
        print(f"   {name:40s} {dm16:10.4f} {dmo:10.4f}      {dv16:10.4f} {dvo:10.4f}")
    print("   worst bf16 / (yardstick + 1e-3) ratio:", worst, flush=True)
