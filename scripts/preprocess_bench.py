"""Throughput of the device input pipeline (DevicePreprocess) on VOC-sized images vs the Pillow chain on one core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ee_semantic_segmentation_amd.get_seg_datasets import DevicePreprocess
from oracle.preprocess_ref import image_chain, target_chain
rng = np.random.default_rng(0)
B, dim = 64, 513
imgs = [torch.from_numpy(rng.integers(0, 256, (375, 500, 3), dtype=np.uint8)).pin_memory() for _ in range(B)]
lbls = [torch.from_numpy(rng.integers(0, 22, (375, 500), dtype=np.uint8)).pin_memory() for _ in range(B)]
pre = DevicePreprocess(dim)
pre.batch(imgs, lbls); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    pre.batch(imgs, lbls)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
dimgs = [i.cuda() for i in imgs]; dl = [l.cuda() for l in lbls]
pre.batch(dimgs, dl); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    pre.batch(dimgs, dl)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for i in range(8):
    image_chain(imgs[i].numpy(), dim); target_chain(lbls[i].numpy(), dim)
dc = (time.perf_counter() - t0) / 8
print(f"device chain, uint8 on host (pinned) : {B / dt:8.0f} img/s  ({dt / B * 1e6:.0f} us/img incl. H2D)")
print(f"device chain, uint8 already on device: {B / dt2:8.0f} img/s  ({dt2 / B * 1e6:.0f} us/img)")
print(f"Pillow + torch CPU chain, one core   : {1 / dc:8.0f} img/s  ({dc * 1e3:.2f} ms/img)")
