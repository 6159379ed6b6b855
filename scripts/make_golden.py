#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the importable subset of the reference.

Run in the build container only (needs /root/reference, which never travels):
    python scripts/make_golden.py
The six importable reference files are imported as-is; their
outputs on seeded inputs are stored together with the inputs.  The fixtures are
data (inputs + expected outputs), never reference source.
"""
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def lovasz_variants():
    """LovaszSoftmax beyond the training default (lovaszsoftmax.py:154-169,185-188 through branchy_seg_losses.py:133-159):
    per_image=True, classes='all', classes=[list]; values + gradients from the reference classes.  Ties: continuous random
    scores make equal errors (and so an order-dependent gradient) a measure-zero event."""
    sys.path.insert(0, REF)
    import branchy_seg_losses as RBSL
    out = {}
    cases = [(30, 2, 3, 5, 12, 10, [0, 2, 4]), (31, 2, 2, 19, 9, 11, [1, 3, 18]), (32, 3, 4, 7, 6, 6, [6])]
    for k, (seed, E, B, C, H, W, lst) in enumerate(cases):
        torch.manual_seed(seed)
        y = torch.randn(E, B, C, H, W)
        t = torch.randint(0, C + 1, [B, 1, H, W])            # label C = void
        if k == 1:
            t[t == 5] = 4                                    # class 5 absent everywhere; others absent in single images
        out[f"y{k}"], out[f"t{k}"], out[f"cls{k}"], out[f"void{k}"] = y.numpy(), t.numpy(), np.array(lst), C
        specs = {"pi_present": dict(classes="present", per_image=True), "all": dict(classes="all", per_image=False),
                 "list": dict(classes=lst, per_image=False), "pi_all": dict(classes="all", per_image=True),
                 "pi_list_prev": dict(classes=lst, per_image=True, prev_out=True)}
        for name, kw in specs.items():
            yy = y.clone().requires_grad_(True)
            l = RBSL.LovaszSoftmax(ignore=C, n_branches=E - 1, **kw)(yy, t)
            l.mean().backward()
            out[f"{name}{k}"] = l.detach().numpy()
            out[f"{name}{k}_grad"] = yy.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "lovasz_variants.npz"), **out)
    print("lovasz variants", {k: float(v) for k, v in out.items() if k[-1].isdigit() and not k.startswith(("y", "t", "cls", "void"))})


def focal_unreduced():
    """FocalLoss with reduction='none' (branchy_seg_losses.py:113-131 through BrSegLoss.forward :24-38): the stacked per-pixel maps
    [E,B,H,W] - with alpha [E,B,B,H,W], the reference's broadcast of the map against alpha[targets] of shape [B,1,H,W] - and the
    gradient of (map * seeded weights).sum() w.r.t. the scores.  Batch sizes 2, 1, 3."""
    sys.path.insert(0, REF)
    import branchy_seg_losses as RBSL
    out = {}
    for k, (seed, E, B, C, H, W, gamma) in enumerate([(50, 2, 2, 21, 9, 11, 2.0), (51, 3, 1, 5, 16, 12, 1.5), (52, 2, 3, 19, 8, 8, 0.0)]):
        torch.manual_seed(seed)
        y = torch.randn(E, B, C, H, W) * 2
        t = torch.randint(0, C, [B, 1, H, W])
        alpha = torch.linspace(0.5, 1.5, C)
        out[f"y{k}"], out[f"t{k}"], out[f"gamma{k}"] = y.numpy(), t.numpy(), np.float64(gamma)
        for name, a in (("plain", None), ("alpha", alpha)):
            yy = y.clone().requires_grad_(True)
            m = RBSL.FocalLoss(alpha=a, gamma=gamma, reduction="none", n_branches=E - 1)(yy, t)
            wgt = torch.randn(m.shape, generator=torch.Generator().manual_seed(seed + 100))
            (m * wgt).sum().backward()
            out[f"{name}{k}"], out[f"{name}{k}_w"], out[f"{name}{k}_grad"] = m.detach().numpy(), wgt.numpy(), yy.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "focal_unreduced.npz"), **out)
    print("focal unreduced", {k: v.shape for k, v in out.items() if k.startswith(("plain", "alpha")) and k[-1].isdigit()})


def main():
    warnings.filterwarnings("ignore")
    if "--only-lovasz-variants" in sys.argv:       # add this fixture without rewriting the others
        return lovasz_variants()
    if "--only-focal-unreduced" in sys.argv:
        return focal_unreduced()
    sys.path.insert(0, REF)
    import my_pixelwise_xentropy as RX
    import branchy_seg_losses as RBSL
    import compute_mIoU as RM
    from scipy.stats import entropy

    os.makedirs(OUT, exist_ok=True)

    # ---- losses: several seeded cases incl. the SURVEY section 4 probe ------
    cases = [  # (seed, E, B, C, H, W, void_label)
        (0, 3, 2, 21, 9, 7, 21),
        (1, 2, 3, 19, 17, 13, 19),
        (2, 4, 2, 5, 33, 20, 5),
        (3, 2, 2, 21, 24, 31, 21),
    ]
    for seed, E, B, C, H, W, void in cases:
        torch.manual_seed(seed)
        y = torch.randn(E, B, C, H, W)
        t = torch.randint(0, C + 1, [B, 1, H, W])
        out = {"y": y.numpy(), "t": t.numpy(), "void": void}
        for red in ("sum", "mean"):
            yy = y.clone().requires_grad_(True)
            l = RX.BrXEntropyLoss(ignore_index=void, b_reduction=red, n_exits=E)(yy, t)
            l.mean().backward()
            out[f"ce_{red}"] = l.detach().numpy()
            out[f"ce_{red}_grad"] = yy.grad.numpy()
        for prev in (False, True):
            yy = y.clone().requires_grad_(True)
            l = RBSL.LovaszSoftmax(classes="present", ignore=void, n_branches=E - 1,
                                   prev_out=prev)(yy, t)
            l.mean().backward()
            out[f"lovasz_prev{int(prev)}"] = l.detach().numpy()
            out[f"lovasz_prev{int(prev)}_grad"] = yy.grad.numpy()
        # per-exit mIoU through the reference accumulator
        mi = []
        for e in range(E):
            m = RM.mIoU(C)
            m(y[e], t)
            mi.append(m.compute().item())
            if e == 0:
                out["acc0"] = m.accumulator.numpy().copy()
        out["miou"] = np.array(mi, dtype=np.float64)
        # entropy gate (eval_br_ent.py:29: scipy entropy base C over axis 0, then mean)
        ents = []
        for e in range(E):
            for b in range(B):
                p = torch.softmax(y[e, b:b + 1] * 3.0, 1).squeeze(0).numpy()
                ents.append(np.mean(entropy(p, base=C, axis=0)))
        out["gate_scale"] = 3.0
        out["entropy"] = np.array(ents, dtype=np.float64).reshape(E, B)
        np.savez_compressed(os.path.join(OUT, f"losses_seed{seed}.npz"), **out)
        print(f"seed {seed}: ce_sum={out['ce_sum']:.9f} lovasz={out['lovasz_prev0']:.9f} "
              f"miou0={mi[0]:.9f}")

    # ---- the reference's own self-check tensors (compute_mIoU.py:66-141) ----
    y_true = np.array([[[[0, 1, 1, 1, 0, 0], [1, 1, 2, 2, 1, 1], [1, 1, 2, 2, 1, 1], [0, 1, 1, 1, 0, 0]]],
                       [[[0, 3, 3, 3, 2, 0], [0, 3, 2, 2, 3, 1], [0, 3, 2, 2, 3, 1], [0, 3, 3, 3, 3, 0]]]],
                      dtype=np.float32)
    y_pred = 100 * np.array([
        [[[1, 0, 0, 0, 1, 1], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 1, 1]],
         [[0, 1, 1, 1, 0, 0], [1, 1, 0, 0, 1, 1], [1, 1, 0, 0, 1, 1], [0, 1, 1, 1, 0, 0]],
         [[0, 0, 0, 0, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 0, 0, 0, 0]],
         [[0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]],
        [[[1, 0, 0, 0, 0, 1], [1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 1]],
         [[0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0, 0]],
         [[0, 0, 0, 0, .5, 0], [0, 0, 1, 1, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 0, 0, 0, 0]],
         [[0, 1, 1, 1, 1.5, 1], [0, 1, 0, 0, 1, 0], [0, 1, 0, 0, 1, 0], [0, 1, 1, 1, 1, 0]]]],
        dtype=np.float32)
    m = RM.mIoU(n_classes=4)
    m(torch.from_numpy(y_pred), torch.from_numpy(y_true))
    val = m.compute().item()
    assert abs(val - 0.9513888955116272) < 1e-12, val
    np.savez_compressed(os.path.join(OUT, "miou_selfcheck.npz"), y_pred=y_pred, y_true=y_true,
                        miou=np.float64(val), acc=m.accumulator.numpy())
    print("selfcheck mIoU", val)

    # ---- per-image mIoU (compute_mIoU.py:38-63): streamed over single images, void pixels present in some ----
    cases = []
    for seed, C, H, W, void_frac in [(10, 21, 19, 23, 0.1), (11, 19, 16, 16, 0.0), (12, 5, 33, 20, 0.3), (13, 21, 8, 8, 0.05)]:
        torch.manual_seed(seed)
        m = RM.img_mIoU()
        ys, ts = [], []
        for _ in range(3):
            y = torch.randn(1, C, H, W)
            t = torch.randint(0, C, [1, H, W])
            t[torch.rand(1, H, W) < void_frac] = C
            m(y, t)
            ys.append(y.numpy()); ts.append(t.numpy())
        cases.append((np.concatenate(ys), np.concatenate(ts), m.compute()))
    np.savez_compressed(os.path.join(OUT, "img_miou.npz"),
                        **{f"y{i}": c[0] for i, c in enumerate(cases)}, **{f"t{i}": c[1] for i, c in enumerate(cases)},
                        expected=np.array([c[2] for c in cases], dtype=np.float64))
    print("img_mIoU", [round(c[2], 9) for c in cases])

    # ---- region / focal losses (branchy_seg_losses.py:40-131): value + gradient w.r.t. the scores ----
    out = {}
    for k, (seed, E, B, C, H, W) in enumerate([(20, 2, 2, 21, 9, 11), (21, 3, 1, 5, 16, 12), (22, 2, 3, 19, 8, 8)]):
        torch.manual_seed(seed)
        y = torch.randn(E, B, C, H, W) * 2
        t_clean = torch.randint(0, C, [B, 1, H, W])
        t_void = t_clean.clone()
        t_void[torch.rand(B, 1, H, W) < 0.15] = C
        out[f"y{k}"], out[f"t{k}"], out[f"tv{k}"] = y.numpy(), t_clean.numpy(), t_void.numpy()
        alpha = torch.linspace(0.5, 1.5, C)
        specs = {
            "dice_mean": (RBSL.DiceLoss(reduction="mean", n_branches=E - 1), t_clean),
            "dice_sum": (RBSL.DiceLoss(reduction="sum", n_branches=E - 1, weights=[0.5 + i for i in range(E)]), t_clean),
            "jaccard_mean": (RBSL.JaccardLoss(reduction="mean", n_branches=E - 1, downgrad_bg=0.3), t_void),
            "jaccard_sum0": (RBSL.JaccardLoss(reduction="sum", n_branches=E - 1, downgrad_bg=0.0), t_void),
            "tversky": (RBSL.TverskyLoss(alpha=.3, beta=.7, reduction="mean", n_branches=E - 1), t_clean),
            "focal_tversky": (RBSL.FocalTverskyLoss(alpha=.3, beta=.7, gamma=1.5, reduction="sum", n_branches=E - 1), t_clean),
            "focal_mean": (RBSL.FocalLoss(gamma=2, reduction="mean", n_branches=E - 1), t_clean),
            "focal_sum_alpha": (RBSL.FocalLoss(alpha=alpha, gamma=1.5, reduction="sum", n_branches=E - 1), t_clean),
        }
        for name, (crit, t) in specs.items():
            yy = y.clone().requires_grad_(True)
            l = crit(yy, t)
            l.backward()
            out[f"{name}{k}"] = l.detach().numpy()
            out[f"{name}{k}_grad"] = np.zeros_like(y.numpy()) if yy.grad is None else yy.grad.numpy()
    out["alpha_lo_hi"] = np.array([0.5, 1.5])
    np.savez_compressed(os.path.join(OUT, "region_losses.npz"), **out)
    print("region losses", {k: float(v) for k, v in out.items() if k.endswith("0") and not k.startswith(("y", "t"))})
    lovasz_variants()


if __name__ == "__main__":
    main()
