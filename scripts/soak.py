"""Soak of the benchmarked training step: the metric's network (R101, 3 exits, 19 classes, 513x513, B=32, bf16, gradient
arena + HIP-graph replay, SGD momentum 0.9 / wd 5e-4 / lr 0.01) for N steps over a small fixed set of synthetic batches.
Checks: every loss finite, the loss falls, every parameter gradient still lives in the arena at the end, no parameter
went non-finite; an EAGER twin (no graph) started from the same weights follows the same losses for the first steps.
usage: python scripts/soak.py [steps] [out.json] [images per step = 32]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
out = sys.argv[2] if len(sys.argv) > 2 else None
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
args = argparse.Namespace(overlap_wgrad=0, reserve_cus=None, no_graph=False, dp_transport="rccl")
NB = 4                                                    # distinct batches, cycled


def make(no_graph):
    a = argparse.Namespace(**vars(args))
    a.no_graph = no_graph
    return bench.Run("resnet101", 2, 19, 513, BATCH, "bf16", "ce", False, 1, 0, dev, a)


run = make(False)
batches = [bench.synth_batch(BATCH, 19, 513, 513, 100 + i, dev) for i in range(NB)]
losses = []
t0 = time.time()
for s in range(steps):
    X, y = batches[s % NB]
    l = run.runner(X, y)
    if s % 10 == 0 or s == steps - 1:
        losses.append((s, float(l.item())))
        print(f"step {s:4d} loss {losses[-1][1]:.4f}", flush=True)
torch.cuda.synchronize()
dt = time.time() - t0
arena = run.net.cfg.arena
lo, hi = arena.flat.data_ptr(), arena.flat.data_ptr() + arena.flat.numel() * 4
in_arena = all(lo <= p.grad.data_ptr() < hi for p in run.net.parameters() if p.grad is not None)
finite = all(bool(torch.isfinite(p).all()) for p in run.net.parameters())
first = sum(v for _, v in losses[:3]) / 3
last = sum(v for _, v in losses[-3:]) / 3
graph_used = run.runner.graph is not None
del run
torch.cuda.empty_cache()

# eager twin for the first 12 steps (same seed -> same initial weights, same batches)
g2 = make(False)
e2 = make(True)
lg, le = [], []
for s in range(12):
    X, y = batches[s % NB]
    lg.append(float(g2.runner(X, y).item()))
    le.append(float(e2.runner(X, y).item()))
twin = max(abs(a - b) / abs(b) for a, b in zip(lg, le))
from ee_semantic_segmentation_amd import kernels as K  # noqa: E402
res = {"workload": f"R101 3 exits 19 classes 513x513 B={BATCH} bf16, arena + HIP graph, 4 synthetic batches cycled", "coop_timeouts": K.coop_timeouts(),
       "steps": steps, "seconds": dt, "hip_graph": graph_used, "loss_first3_mean": first, "loss_last3_mean": last,
       "losses_every_10": losses, "all_losses_finite": all(v == v and abs(v) < 1e9 for _, v in losses),
       "grads_in_arena": in_arena, "params_finite": finite,
       "graph_vs_eager_first12_max_rel": twin, "graph_first12": lg, "eager_first12": le}
print(json.dumps({k: v for k, v in res.items() if k not in ("losses_every_10", "graph_first12", "eager_first12")}))
assert res["all_losses_finite"] and in_arena and finite and last < first and twin < 2e-2 and res["coop_timeouts"] == 0, "soak failed"
if out:
    json.dump(res, open(out, "w"), indent=1)
