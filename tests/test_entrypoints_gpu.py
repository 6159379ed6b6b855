"""GPU tests of the reference's entry-point surface: train+test CLI,
entropy-gated evaluator, progressive per-image inference dict contract."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(n=1, img=65, C=21):
    from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
    torch.manual_seed(0)
    return branchyDeepv3(None, "deeplabv3_resnet50", n, img, count_branches=False, num_classes=C).to(DEV)


@pytest.mark.parametrize("loss", ["ce", "lovasz"])
def test_main_entry_point_trains_and_tests(tmp_path, monkeypatch, loss):
    from ee_semantic_segmentation_amd import main_bradeepv3
    monkeypatch.chdir(tmp_path)
    out = main_bradeepv3.main(loss, ["-t", "resnet50", "-n", "1", "-e", "2", "-N", "t50", "--dim", "65", "--batch", "4",
                                     "--loss", loss])
    assert os.path.exists(out)
    sd = torch.load(out, weights_only=True)
    assert "base_model.0.0.weight" in sd and "branches.0.4.bias" in sd
    rows = open(tmp_path / "mIoU_1_branches_results.csv").read().strip().splitlines()
    assert rows[0].split(",") == ["net_id", "b1_mIoU", "mIoU"] and rows[1].startswith("t50,")
    tr = open(tmp_path / "voc_seg_results" / "t50" / "t50_tr.csv").read().splitlines()
    assert tr[0].split(",") == ["val_mIoU_b1_mIoU", "val_mIoU_mIoU", "lr"] and len(tr) == 3
    # like the reference (deepv3_funcs.py:186-188,259) the final model file replaces the
    # best-checkpoint dict that lived at the same path during training
    assert os.path.samefile(out, tmp_path / "voc_seg_results" / "t50" / "t50.pth")


def test_main_bradeepv3_ce_at_the_baseline_plumbing_shape(tmp_path, monkeypatch):
    """BASELINE.json configs[0]: main_bradeepv3_ce, DeepLabV3-ResNet50, 2 exits, 256x256, B=2, 21 classes (the reference's
    CPU-runnable plumbing case), through the reference's own flags; fp32 = the parity mode."""
    from ee_semantic_segmentation_amd import main_bradeepv3
    monkeypatch.chdir(tmp_path)
    out = main_bradeepv3.main("ce", ["-t", "resnet50", "-n", "1", "-e", "2", "-N", "c1", "--dim", "256", "--batch", "2"])
    sd = torch.load(out, weights_only=True)
    assert sd["branches.0.4.weight"].shape[0] == 21 and all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    rows = open(tmp_path / "mIoU_1_branches_results.csv").read().strip().splitlines()
    vals = [float(v) for v in rows[1].split(",")[1:]]
    assert len(vals) == 2 and all(0.0 <= v <= 1.0 or v != v for v in vals)      # NaN allowed: reference quirk B-8


def test_br_evaluator_and_progressive_inference():
    from ee_semantic_segmentation_amd.eval_br_ent import br_evaluator, img_norm_entropy
    from ee_semantic_segmentation_amd.ee_dnn_op_ne import eval_ee_deeplabv3
    from ee_semantic_segmentation_amd.get_seg_datasets import SyntheticSeg
    C, img = 21, 65
    net = _net(2, img, C).eval()
    ds = SyntheticSeg(4, img, C, seed=3)
    loader = torch.utils.data.DataLoader(ds, batch_size=1)
    for metric, size in (("ent", 1), ("max", 4), ("min", 4)):
        lo = br_evaluator(net, 3, C, loader, DEV, tau=0.0, metric=metric, size=size)     # nothing exits early
        hi = br_evaluator(net, 3, C, loader, DEV, tau=2.0, metric=metric, size=size)     # everything exits at b1
        assert lo["count_out"] == 4 and lo["b1_count"] == 0 and lo["out_gl"] == 4
        assert hi["b1_count"] == 4 and hi["count_out"] == 0
        assert set(lo) == {"b1_mIoU", "b1_count", "b2_mIoU", "b2_count", "mIoU_out", "count_out", "mIoU_gl", "out_gl",
                           "t", "pool", "pool_size"}
    X, y = ds[0]
    gate = img_norm_entropy(C)
    never = eval_ee_deeplabv3(net, gate, 0.0, device=DEV)(X)
    always = eval_ee_deeplabv3(net, gate, 2.0, device=DEV)(X)
    assert never["n"] == 3 and torch.equal(never["exit"], never["last"]) and never["exit_flops"] == never["last_flops"]
    assert always["n"] == 1 and always["exit"].shape == (img, img) and always["exit_flops"] < always["last_flops"]
    assert always["edge_flops"] == always["exit_flops"]
    stop = eval_ee_deeplabv3(net, gate, 2.0, device=DEV, stop_at_exit=True)(X)
    assert "last" not in stop and torch.equal(stop["exit"], always["exit"])
    # the final mask equals the argmax of the network's own stacked output
    with torch.no_grad():
        full = net(X.unsqueeze(0).to(DEV))
    assert torch.equal(never["last"], full[-1, 0].argmax(0).cpu())


def test_similarity_gated_evaluator_and_progressive_inference():
    """eval_br_sim.br_evaluator + ee_dnn_op.eval_ee_deeplabv3 (similarity gate between consecutive exits): dict
    contracts, the two limits of the threshold, and the gate values against the numpy oracle on the same maps."""
    from ee_semantic_segmentation_amd.eval_br_sim import br_evaluator, gate_function
    from ee_semantic_segmentation_amd.ee_dnn_op import eval_ee_deeplabv3
    from ee_semantic_segmentation_amd.get_seg_datasets import SyntheticSeg
    from oracle import sim_ref as R
    C, img = 21, 65
    net = _net(3, img, C).eval()                 # 4 exits: gates compare (b1,b2) and (b2,b3)
    ds = SyntheticSeg(4, img, C, seed=3)
    loader = torch.utils.data.DataLoader(ds, batch_size=2)
    keys = {"b1_mIoU", "b1_count", "b2_mIoU", "b2_count", "b3_mIoU", "b3_count", "mIoU_out", "count_out", "mIoU_gl",
            "out_gl", "t"}
    for metric, never, always in (("mse", -1.0, 1e9), ("vi", -1.0, 1e9), ("h_xy", -1.0, 1e9), ("h_yx", -1.0, 1e9),
                                  ("nmi", 1e9, -1.0), ("ssim", 1e9, -2.0)):
        lo = br_evaluator(net, 4, C, loader, DEV, metric, never, ignore=(0,))
        hi = br_evaluator(net, 4, C, loader, DEV, metric, always, ignore=(0,))
        assert set(lo) == keys
        assert lo["count_out"] == 4 and lo["b1_count"] == lo["b2_count"] == 0 and lo["out_gl"] == 4
        # the first gated pair compares exits 0 and 1 and the image leaves AT exit 1: out_count[1] -> "b2_count"
        # (eval_br_sim.py:41-48,61-63); exit 0 can never be left at, it only provides the first map
        assert hi["b2_count"] == 4 and hi["b1_count"] == 0 and hi["count_out"] == 0
    with pytest.raises(ValueError):              # SSIM works on the label maps, not on the contingency table
        gate_function("ssim")
    # gate values vs the oracle on the materialised label maps
    X, _ = ds[1]
    with torch.no_grad():
        full = net(X.unsqueeze(0).to(DEV)).cpu().numpy()          # [E,1,C,H,W]
    from ee_semantic_segmentation_amd import kernels as K
    net.fused_outputs = True
    with torch.no_grad():
        el = net(X.unsqueeze(0).to(DEV))
    net.fused_outputs = False
    t01 = K.argmax_pair_hist(el.lowres[0].contiguous(), el.lowres[1].contiguous(), C, img, img)[0].double()
    assert abs(gate_function("mse")[0](t01) - R.mse(full[0], full[1])) < 1e-9
    assert abs(gate_function("vi", (0,))[0](t01) - R.vi(full[0], full[1], (0,))) < 1e-9
    assert abs(gate_function("nmi")[0](t01) - R.nmi(full[0], full[1])) < 1e-9
    # progressive operator
    f_mse = gate_function("mse")[0]
    never = eval_ee_deeplabv3(net, f_mse, -1.0, device=DEV)(X)
    always = eval_ee_deeplabv3(net, f_mse, 1e9, device=DEV)(X)
    assert never["n"] == 4 and torch.equal(never["exit"], never["last"]) and never["exit_flops"] == never["last_flops"]
    assert always["n"] == 2                      # branch 0 is only the reference map; branch 1 is the first that can exit
    assert always["exit_flops_2"] < always["exit_flops"] < always["last_flops"]
    assert torch.equal(always["exit"], torch.from_numpy(full[1, 0].argmax(0)))
    assert set(always) >= {"exit", "exit_flops", "exit_flops_2", "edge_flops", "edge_flops_2", "n", "last", "last_flops",
                           "last_flops_2"}
    stop = eval_ee_deeplabv3(net, f_mse, 1e9, device=DEV, stop_at_exit=True)(X)
    assert "last" not in stop and torch.equal(stop["exit"], always["exit"])
    from ee_semantic_segmentation_amd.sim_metrics import SSIM
    s_never = eval_ee_deeplabv3(net, SSIM(C - 1), 2.0, less_than=False, device=DEV)(X)       # SSIM > 2 never holds
    s_always = eval_ee_deeplabv3(net, SSIM(C - 1), -2.0, less_than=False, device=DEV)(X)
    assert s_never["n"] == 4 and s_always["n"] == 2 and torch.equal(s_always["exit"], always["exit"])
