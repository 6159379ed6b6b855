"""Oracle model anchors: torchvision's published parameter counts and the
survey's branch placement.  CPU only."""
import torch

from oracle.deeplab_ref import branchyDeepv3

AUX_HEAD = 2_365_205   # torchvision FCNHead aux classifier (not part of branchyDeepv3)


def _count(mods):
    return sum(p.numel() for p in mods.parameters())


def test_param_counts_match_torchvision():
    m50 = branchyDeepv3("deeplabv3_resnet50", 1, 256, count_branches=False)
    m101 = branchyDeepv3("deeplabv3_resnet101", 2, 256, count_branches=False)
    assert _count(m50.base_model) + _count(m50.classifier) + AUX_HEAD == 42_004_074
    assert _count(m101.base_model) + _count(m101.classifier) + AUX_HEAD == 60_996_202
    assert _count(m50) == 55_769_706 and _count(m101) == 83_290_495


def test_branch_placement_matches_survey():
    assert branchyDeepv3("deeplabv3_resnet50", 1, 256, False).split_names == ["layer4.0"]
    assert branchyDeepv3("deeplabv3_resnet101", 2, 513, False).split_names == ["layer3.10", "layer4.0"]
    assert branchyDeepv3("deeplabv3_resnet101", 3, 256, False).split_names == \
        ["layer3.7", "layer3.16", "layer4.0"]


def test_forward_contract_and_keys():
    m = branchyDeepv3("deeplabv3_resnet50", 1, 64, False).eval()
    with torch.no_grad():
        y = m(torch.randn(2, 3, 64, 64))
    assert y.shape == (2, 2, 21, 64, 64)
    keys = set(m.state_dict().keys())
    for k in ("base_model.0.0.weight", "base_model.0.1.running_mean", "base_model.0.4.conv1.weight",
              "base_model.0.4.downsample.0.weight", "branches.0.0.convs.4.1.weight",
              "branches.0.0.project.1.weight", "branches.0.4.bias", "classifier.4.weight"):
        assert k in keys, k
