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
