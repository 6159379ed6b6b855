"""Inference-only early-exit benchmark (BASELINE.json configs[3]: R101, 4 exits, 1024x2048, entropy-gated, bf16):
  all:          every exit computed + 3 fused gates + 4 fused argmax masks (eval_br_ent.py's evaluation shape);
  progressive:  branchyDeepv3.forward_progressive - exits decided on the device, later sections only for the images
                still in flight (B = 1 and B = 8), exit histogram for thresholds at the quartiles of the gate values.
usage: python scripts/infer_bench.py [H W]          (bench.py's `secondary.configs3_infer` calls run(quick=True))"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(H=1024, W=2048, batches=(1, 8), quick=False):
    """quick: fewer timed iterations and only the median-threshold progressive run (a few seconds: the driver-run bench line)."""
    from ee_semantic_segmentation_amd import kernels as K
    from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
    C = 19
    torch.manual_seed(0)
    net = branchyDeepv3(None, "deeplabv3_resnet101", 3, 1024, count_branches=False, num_classes=C,
                        compute_dtype=torch.bfloat16).cuda().eval()
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if type(m).__name__ == "BatchNorm2d":
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
    macs = net.macs(H, W)
    n_it, n_warm = (5, 2) if quick else (10, 3)

    def timed(fn):
        for _ in range(n_warm):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n_it):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_it, out

    def all_exits(X):
        with torch.no_grad():
            lrs = net.forward_lowres(X)
            gates = [K.entropy_gate(lr, C, H, W, 0.5) for lr in lrs[:-1]]
            preds = [K.argmax_confusion(lr, C, None, H, W, want_pred=True)[1] for lr in lrs]
        return gates, preds

    res = {"workload": f"DeepLabV3-resnet101 4 exits, {H}x{W}, {C} classes, bf16 inference, splits {net.split_names}",
           "gmac_per_image_all_exits": macs / 1e9}
    for B in batches:
        X = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(7)).cuda()
        X *= torch.linspace(0.3, 2.0, B, device="cuda").view(B, 1, 1, 1)
        dt, (gates, _) = timed(lambda: all_exits(X))
        ents = torch.stack([g[0] for g in gates]).cpu().numpy()
        row = {"all_exits_ms_per_image": dt / B * 1e3, "all_exits_img_per_s": B / dt, "all_exits_tflops": 2 * macs * B / dt / 1e12}
        variants = (("q50", 50),) if quick else (("never", None), ("q75", 75), ("q50", 50), ("q25", 25), ("always", 101))
        for name, q in variants:
            tau = -1.0 if q is None else (2.0 if q > 100 else float(np.percentile(ents, q)))
            dtp, out = timed(lambda: net.forward_progressive(X, tau))
            hist = np.bincount(out["exit"].cpu().numpy(), minlength=net.n_branches + 2)[1:].tolist()
            row[f"progressive_tau_{name}"] = {"tau": tau, "ms_per_image": dtp / B * 1e3, "img_per_s": B / dtp, "exit_histogram": hist}
        res[f"B={B}"] = row
        del X
    del net
    return res


if __name__ == "__main__":
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
    print(json.dumps(run(H, W), indent=1))
