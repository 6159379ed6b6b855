"""Which torch ops of one training step launch device copies / fills?  (eager step under torch.profiler, shapes + python stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
from ee_semantic_segmentation_amd.optim import SGD
from ee_semantic_segmentation_amd.parallel import ArenaReducer, GraphedTrainStep
B = 4
net = branchyDeepv3(None, "deeplabv3_resnet101", 2, 513, count_branches=False, num_classes=19, compute_dtype=torch.bfloat16,
                    fused_outputs=True).to("cuda").train()
net.enable_grad_arena()
opt = SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
runner = GraphedTrainStep(net, BrXEntropyLoss(ignore_index=19, b_reduction="sum", n_exits=3), opt, ArenaReducer(net), warmup=100)
x = torch.randn(B, 3, 513, 513, device="cuda")
y = torch.randint(0, 19, (B, 1, 513, 513), device="cuda")
for _ in range(2):
    runner(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    runner(x, y)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::add_", "aten::add", "aten::mul", "aten::sum",
                  "aten::cat", "aten::stack", "aten::_to_copy", "aten::zeros", "aten::zeros_like"):
        st = [s for s in (e.stack or []) if "ee_semantic_segmentation_amd" in s or "bench" in s]
        cnt[(e.name, str(e.input_shapes)[:70], st[0][-70:] if st else "?")] += 1
for (n, sh, st), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d} {n:18s} {sh:70s} {st}")
