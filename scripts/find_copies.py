"""Which torch ops put device-to-device copies / fills into the training step?  (kernel traces show ~58 __amd_rocclr_copyBuffer
and ~8 fill launches per step.)  Eager step under torch.profiler with stacks; prints the call sites."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import Run, synth_batch
import argparse
args = argparse.Namespace(overlap_wgrad=0, reserve_cus=None, no_graph=True, dp_transport="rccl")
run = Run("resnet101", 2, 19, 257, 4, "bf16", "ce", False, 1, 0, torch.device("cuda", 0), args)
for _ in range(3):
    run.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run.step()
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat", "aten::add_", "aten::mul", "aten::div", "aten::sum", "aten::mean", "aten::stack"):
        st = [s for s in (ev.stack or []) if "ee_semantic_segmentation_amd" in s or "bench.py" in s]
        sites[(ev.name, st[0] if st else "?")] += 1
for (name, site), n in sites.most_common(40):
    print(n, name, site)
