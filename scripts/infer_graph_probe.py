"""Is the 1024x2048 inference host-bound at B=1?  all-exits evaluation and forward_progressive: eager vs HIP-graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
C, H, W = 19, 1024, 2048
torch.manual_seed(0)
net = branchyDeepv3(None, "deeplabv3_resnet101", 3, 1024, count_branches=False, num_classes=C, compute_dtype=torch.bfloat16).cuda().eval()


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in (1, 8):
    X = torch.randn(B, 3, H, W, device="cuda")

    def all_exits():
        with torch.no_grad():
            lrs = net.forward_lowres(X)
            gates = [K.entropy_gate(lr, C, H, W, 0.5) for lr in lrs[:-1]]
            preds = [K.argmax_confusion(lr, C, None, H, W, want_pred=True)[1] for lr in lrs]
        return gates, preds

    def prog():
        return net.forward_progressive(X, 0.005)

    for name, fn in (("all exits", all_exits), ("progressive tau=0.005", prog)):
        te = timed(fn)
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            tg = timed(g.replay)
            print(f"B={B} {name:24s}: eager {te / B:6.2f} ms/image, graph replay {tg / B:6.2f} ms/image", flush=True)
            del g
        except Exception as e:
            print(f"B={B} {name:24s}: eager {te / B:6.2f} ms/image, capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
            torch.cuda.synchronize()
