"""Which weight-gradient groups of one training step share a launch?  (eager step, kernels.conv_wgrad_group wrapped)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = branchyDeepv3(None, "deeplabv3_resnet101", 2, 513, count_branches=False, num_classes=19).to("cuda").train()
net.cfg.compute_dtype = torch.bfloat16
net.enable_grad_arena()
seen = collections.Counter()
orig = K.conv_wgrad_group


def wrapped(items):
    orig(items)
    n = lib().eeseg_last_kernel(3)
    key = " + ".join(f"{it[2]}x{it[3]} {it[0].shape[-1]}->{it[1].shape[-1]} @{it[1].shape[1]}" for it in items)
    seen[(key, n)] += 1


K.conv_wgrad_group = wrapped
import ee_semantic_segmentation_amd.engine as E
E.K.conv_wgrad_group = wrapped
x = torch.randn(B, 3, 513, 513, device="cuda")
y = torch.randint(0, 19, (B, 1, 513, 513), device="cuda")
crit = BrXEntropyLoss(ignore_index=19, b_reduction="sum", n_exits=3)
loss = crit(net(x), y)
loss.mean().backward()
torch.cuda.synchronize()
for (key, n), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d} x  grouped {n}  [{key}]")
