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


def test_my_branch_any_bottleneck_width_keeps_reference_shapes():
    """my_branch(bottleneck=100) (the reference takes any width, from_deepv3_new.py:23-24): the host module stores the
    100-wide tensors zero-padded to 128, but state_dict keys/shapes, load/save round trips, the default
    initialisation bounds and the MAC count are those of the true width."""
    from ee_semantic_segmentation_amd.from_deepv3_new import head_macs, my_branch
    from oracle.deeplab_ref import my_branch as RM
    params = dict(atrous_rates=[2, 4], nout_channels=128, bottleneck=100)
    torch.manual_seed(0)
    ref = RM(256, 21, **params)
    head = my_branch(256, 21, **params)
    sd, rsd = head.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(rsd.keys())
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in rsd.values()]
    pre = head.pre
    assert (pre.out_channels, pre.cout_stored, tuple(pre.weight.shape)) == (100, 128, (128, 256, 1, 1))
    assert float(pre.weight[100:].abs().max()) == 0.0 and float(pre.bias[100:].abs().max()) == 0.0
    assert float(pre.weight[:100].abs().max()) <= 1 / 256 ** 0.5
    a0 = head.aspp.convs[1][0]
    assert tuple(a0.weight.shape) == (128, 128, 3, 3) and float(a0.weight[:, 100:].abs().max()) == 0.0
    assert float(a0.weight[:, :100].abs().max()) <= 1 / (100 * 9) ** 0.5
    head.load_state_dict(rsd)
    for k, v in head.state_dict().items():
        assert torch.equal(v, rsd[k]), k
    assert float(pre.weight[100:].abs().max()) == 0.0 and float(a0.weight[:, 100:].abs().max()) == 0.0
    with torch.no_grad():                                   # an external in-place init ...
        for p in head.parameters():
            p.fill_(0.5)
    head.train()
    head._keep_padding_zero()                               # ... is undone for the padding by the next training forward
    assert float(pre.weight[100:].abs().max()) == 0.0 and float(a0.weight[:, 100:].abs().max()) == 0.0
    assert float(pre.weight[:100].min()) == 0.5
    narrow = my_branch(256, 21, atrous_rates=[2, 4], nout_channels=128, bottleneck=128)
    h = w = 33
    per_ch = (head_macs(narrow, h, w) - head_macs(my_branch(256, 21, atrous_rates=[2, 4], nout_channels=128,
                                                            bottleneck=64), h, w)) // 64
    assert head_macs(head, h, w) == head_macs(narrow, h, w) - 28 * per_ch      # MACs of width 100, not of 128
