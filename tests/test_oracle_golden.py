"""Pins the oracle (oracle/*.py) against golden vectors produced by running the
importable reference files (scripts/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import losses_ref, metrics_ref

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "losses_seed*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_ce_matches_reference(path):
    g = np.load(path)
    y = torch.from_numpy(g["y"]); t = torch.from_numpy(g["t"]); void = int(g["void"])
    E = y.shape[0]
    for red in ("sum", "mean"):
        yy = y.clone().requires_grad_(True)
        l = losses_ref.br_xentropy(yy, t, ignore_index=void, b_reduction=red, n_exits=E)
        l.mean().backward()
        assert abs(l.item() - float(g[f"ce_{red}"])) <= 1e-6 * max(1, abs(l.item()))
        np.testing.assert_allclose(yy.grad.numpy(), g[f"ce_{red}_grad"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_lovasz_matches_reference(path):
    g = np.load(path)
    y = torch.from_numpy(g["y"]); t = torch.from_numpy(g["t"]); void = int(g["void"])
    E = y.shape[0]
    for prev in (False, True):
        yy = y.clone().requires_grad_(True)
        l = losses_ref.br_lovasz(yy, t, ignore=void, n_branches=E - 1, prev_out=prev)
        l.mean().backward()
        assert abs(l.item() - float(g[f"lovasz_prev{int(prev)}"])) <= 2e-6 * max(1, abs(l.item()))
        np.testing.assert_allclose(yy.grad.numpy(), g[f"lovasz_prev{int(prev)}_grad"],
                                   rtol=1e-4, atol=1e-7)


VARIANTS = {"pi_present": dict(classes="present", per_image=True), "all": dict(classes="all", per_image=False),
            "list": dict(classes=None, per_image=False), "pi_all": dict(classes="all", per_image=True),
            "pi_list_prev": dict(classes=None, per_image=True, prev_out=True)}


def test_lovasz_variants_match_reference():
    """per_image=True / classes='all' / classes=[list] (lovaszsoftmax.py:154-169,185-188): oracle vs values and gradients
    the reference classes produced (scripts/make_golden.py lovasz_variants)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lovasz_variants.npz"))
    for k in range(3):
        y, t, void = torch.from_numpy(g[f"y{k}"]), torch.from_numpy(g[f"t{k}"]), int(g[f"void{k}"])
        for name, kw in VARIANTS.items():
            kw = dict(kw)
            if kw["classes"] is None:
                kw["classes"] = [int(c) for c in g[f"cls{k}"]]
            yy = y.clone().requires_grad_(True)
            l = losses_ref.br_lovasz(yy, t, ignore=void, n_branches=y.shape[0] - 1, **kw)
            l.mean().backward()
            want = float(g[f"{name}{k}"])
            assert abs(l.item() - want) <= 2e-6 * max(1, abs(want)), (name, k, l.item(), want)
            np.testing.assert_allclose(yy.grad.numpy(), g[f"{name}{k}_grad"], rtol=1e-4, atol=1e-7, err_msg=f"{name}{k}")


def test_survey_probe_values():
    g = np.load([p for p in CASES if "seed0" in p][0])
    assert abs(float(g["ce_sum"]) - 10.523826599121094) < 1e-6
    assert abs(float(g["lovasz_prev0"]) - 6.174560070037842) < 1e-6
    assert abs(float(g["miou"][0]) - 0.02288685366511345) < 1e-9


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_miou_and_entropy_match_reference(path):
    g = np.load(path)
    y, t = g["y"], g["t"]
    E, B, C = y.shape[:3]
    for e in range(E):
        m = metrics_ref.mIoU(C)
        m(y[e], t)
        if e == 0:
            np.testing.assert_array_equal(m.acc, g["acc0"])
        got, want = float(m.compute()), float(g["miou"][e])
        assert (np.isnan(got) and np.isnan(want)) or abs(got - want) < 1e-7
        for b in range(B):
            p = metrics_ref.softmax_np(y[e, b] * float(g["gate_scale"]), axis=0)
            ent = metrics_ref.img_norm_entropy(p, C)
            assert abs(ent - g["entropy"][e, b]) < 2e-6


def test_miou_selfcheck_vector(golden_dir):
    g = np.load(os.path.join(golden_dir, "miou_selfcheck.npz"))
    m = metrics_ref.mIoU(4)
    m(g["y_pred"], g["y_true"])
    np.testing.assert_array_equal(m.acc, g["acc"])
    assert abs(float(m.compute()) - 0.9513888955116272) < 1e-7


def test_block_reduce_semantics():
    a = np.arange(12, dtype=np.float32).reshape(3, 4) + 1
    mx = metrics_ref.block_reduce(a, (2, 2), np.max)
    mn = metrics_ref.block_reduce(a, (2, 2), np.min)
    np.testing.assert_array_equal(mx, [[6, 8], [10, 12]])
    np.testing.assert_array_equal(mn, [[1, 3], [0, 0]])   # zero padding wins the min


def test_img_miou_oracle_matches_reference_vectors(golden_dir):
    """compute_mIoU.img_mIoU run by scripts/make_golden.py on 4 x 3 seeded images (void pixels in three cases)."""
    from oracle.metrics_ref import img_mIoU
    g = np.load(os.path.join(golden_dir, "img_miou.npz"))
    for i, want in enumerate(g["expected"]):
        m = img_mIoU()
        for y, t in zip(g[f"y{i}"], g[f"t{i}"]):
            m(y[None], t[None])
        assert abs(m.compute() - want) < 1e-6, (i, m.compute(), want)


def _region_cases(g, k):
    import torch
    from oracle import losses_ref as L
    y = torch.from_numpy(g[f"y{k}"])
    E, C = y.shape[0], y.shape[2]
    t, tv = torch.from_numpy(g[f"t{k}"]), torch.from_numpy(g[f"tv{k}"])
    alpha = torch.linspace(0.5, 1.5, C)
    return y, {
        "dice_mean": lambda yy: L.br_dice(yy, t, E, reduction="mean"),
        "dice_sum": lambda yy: L.br_dice(yy, t, E, reduction="sum", weights=[0.5 + i for i in range(E)]),
        "jaccard_mean": lambda yy: L.br_jaccard(yy, tv, E, reduction="mean", downgrad_bg=0.3),
        "jaccard_sum0": lambda yy: L.br_jaccard(yy, tv, E, reduction="sum", downgrad_bg=0.0),
        "tversky": lambda yy: L.br_tversky(yy, t, E, alpha=.3, beta=.7, reduction="mean"),
        "focal_tversky": lambda yy: L.br_tversky(yy, t, E, alpha=.3, beta=.7, gamma=1.5, reduction="sum"),
        "focal_mean": lambda yy: L.br_focal(yy, t, E, gamma=2, reduction="mean"),
        "focal_sum_alpha": lambda yy: L.br_focal(yy, t, E, alpha=alpha, gamma=1.5, reduction="sum"),
    }


@pytest.mark.parametrize("k", [0, 1, 2])
def test_region_and_focal_losses_oracle_matches_reference_vectors(golden_dir, k):
    """Dice / Jaccard / Tversky / FocalTversky / Focal (branchy_seg_losses.py:40-131): values and gradients produced
    by the reference classes (scripts/make_golden.py) vs the torch restatement in oracle/losses_ref.py."""
    g = np.load(os.path.join(golden_dir, "region_losses.npz"))
    y, cases = _region_cases(g, k)
    for name, fn in cases.items():
        yy = y.clone().requires_grad_(True)
        l = fn(yy)
        want = float(g[f"{name}{k}"])
        assert abs(float(l) - want) <= 1e-5 * max(1.0, abs(want)), (name, float(l), want)
        if l.requires_grad:
            l.backward()
            got = yy.grad.numpy()
        else:
            got = np.zeros_like(y.numpy())
        wg = g[f"{name}{k}_grad"]
        assert np.abs(got - wg).max() <= 1e-5 * max(1e-6, np.abs(wg).max()) + 1e-9, name


@pytest.mark.parametrize("k", [0, 1, 2])
def test_unreduced_focal_oracle_matches_reference_vectors(golden_dir, k):
    """FocalLoss(reduction='none') (branchy_seg_losses.py:113-131, :24-38): the stacked maps [E,B,H,W] - [E,B,B,H,W] with alpha,
    the reference's broadcast - and the gradient of (map * weights).sum(), reference classes vs oracle/losses_ref.py."""
    import torch
    from oracle import losses_ref as L
    g = np.load(os.path.join(golden_dir, "focal_unreduced.npz"))
    y, t = torch.from_numpy(g[f"y{k}"]), torch.from_numpy(g[f"t{k}"])
    E, C, gamma = y.shape[0], y.shape[2], float(g[f"gamma{k}"])
    for name, alpha in (("plain", None), ("alpha", torch.linspace(0.5, 1.5, C))):
        yy = y.clone().requires_grad_(True)
        m = L.br_focal(yy, t, E, alpha=alpha, gamma=gamma, reduction="none")
        want = g[f"{name}{k}"]
        assert tuple(m.shape) == want.shape and np.abs(m.detach().numpy() - want).max() <= 1e-5 * np.abs(want).max(), name
        (m * torch.from_numpy(g[f"{name}{k}_w"])).sum().backward()
        wg = g[f"{name}{k}_grad"]
        assert np.abs(yy.grad.numpy() - wg).max() <= 1e-5 * np.abs(wg).max() + 1e-9, name


def test_ssim_oracle_against_direct_window_loops():
    """oracle.sim_ref.ssim (summed-area-table form) vs the definition written out with explicit 7x7 window loops
    (skimage.metrics.structural_similarity defaults; parity unpinned: scikit-image is absent)."""
    import numpy as np
    from oracle import sim_ref as R
    rng = np.random.default_rng(5)
    a = rng.integers(0, 21, (13, 17))
    b = np.where(rng.random((13, 17)) < 0.7, a, rng.integers(0, 21, (13, 17)))
    dr, win = 20.0, 7
    c1, c2 = (0.01 * dr) ** 2, (0.03 * dr) ** 2
    vals = []
    for y in range(13 - win + 1):
        for x in range(17 - win + 1):
            wa, wb = a[y:y + win, x:x + win].astype(np.float64), b[y:y + win, x:x + win].astype(np.float64)
            ux, uy = wa.mean(), wb.mean()
            vx, vy = wa.var(ddof=1), wb.var(ddof=1)
            vxy = ((wa - ux) * (wb - uy)).sum() / (win * win - 1)
            vals.append((2 * ux * uy + c1) * (2 * vxy + c2) / ((ux * ux + uy * uy + c1) * (vx + vy + c2)))
    assert abs(R.ssim(a, b, dr) - float(np.mean(vals))) < 1e-12
    assert abs(R.ssim(a, a, dr) - 1.0) < 1e-15
