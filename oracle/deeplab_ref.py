"""Oracle: pure-torch CPU restatement of the reference's early-exit DeepLabV3.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Follows
``from_deepv3_new.py:56-155`` for the branchy structure/forward and SURVEY.md
Appendix A for the (un-vendored) torchvision layers.  State-dict keys match the
reference layout (Appendix A.4): ``base_model.{s}.{j}.*``, ``branches.{i}.*``,
``classifier.*``.

Parity note: architecture parity is UNPINNED (torchvision absent); anchors are
the published parameter counts, checked in tests/test_oracle_model.py.
"""
import re

import torch
from torch import nn
from torch.nn import functional as F


# --------------------------------------------------------------------------
# torchvision.models.resnet.Bottleneck (Appendix A.1)
# --------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        width = planes
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


def _resnet_backbone(blocks):
    """Ordered (name, module) list == IntermediateLayerGetter children, with
    replace_stride_with_dilation=[False, True, True] (output stride 8)."""
    mods = [("conv1", nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)),
            ("bn1", nn.BatchNorm2d(64)),
            ("relu", nn.ReLU(inplace=True)),
            ("maxpool", nn.MaxPool2d(3, stride=2, padding=1))]
    state = {"inplanes": 64, "dilation": 1}

    def make_layer(name, planes, n, stride, dilate):
        prev_dil = state["dilation"]
        if dilate:
            state["dilation"] *= stride
            stride = 1
        down = None
        if stride != 1 or state["inplanes"] != planes * 4:
            down = nn.Sequential(
                nn.Conv2d(state["inplanes"], planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        out = [(f"{name}.0", Bottleneck(state["inplanes"], planes, stride, down, prev_dil))]
        state["inplanes"] = planes * 4
        for i in range(1, n):
            out.append((f"{name}.{i}", Bottleneck(state["inplanes"], planes, 1, None,
                                                  state["dilation"])))
        return out

    mods += make_layer("layer1", 64, blocks[0], 1, False)
    mods += make_layer("layer2", 128, blocks[1], 2, False)
    mods += make_layer("layer3", 256, blocks[2], 2, True)
    mods += make_layer("layer4", 512, blocks[3], 2, True)
    for _, m in mods:                       # torchvision ResNet init
        for sub in m.modules():
            if isinstance(sub, nn.Conv2d):
                nn.init.kaiming_normal_(sub.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(sub, nn.BatchNorm2d):
                nn.init.constant_(sub.weight, 1)
                nn.init.constant_(sub.bias, 0)
    return mods


# --------------------------------------------------------------------------
# torchvision.models.segmentation.deeplabv3 (Appendix A.2)
# --------------------------------------------------------------------------
class ASPPConv(nn.Sequential):
    def __init__(self, cin, cout, dilation):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())


class ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        size = x.shape[-2:]
        for mod in self:
            x = mod(x)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


class ASPP(nn.Module):
    def __init__(self, cin, atrous_rates=(12, 24, 36), cout=256):
        super().__init__()
        mods = [nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())]
        for r in atrous_rates:
            mods.append(ASPPConv(cin, cout, r))
        mods.append(ASPPPooling(cin, cout))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(len(mods) * cout, cout, 1, bias=False),
                                     nn.BatchNorm2d(cout), nn.ReLU(), nn.Dropout(0.5))

    def forward(self, x):
        return self.project(torch.cat([c(x) for c in self.convs], dim=1))


class DeepLabHead(nn.Sequential):
    def __init__(self, cin, num_classes, atrous_rates=(12, 24, 36), mid=256):
        super().__init__(ASPP(cin, atrous_rates, mid),
                         nn.Conv2d(mid, mid, 3, padding=1, bias=False),
                         nn.BatchNorm2d(mid), nn.ReLU(),
                         nn.Conv2d(mid, num_classes, 1))


# --------------------------------------------------------------------------
# Analytic conv-MAC counter (stands in for pthflops)
# --------------------------------------------------------------------------
def conv_macs(mods, img_dim, cin=3):
    """Cumulative conv MACs after each entry of an ordered module list."""
    h = w = img_dim
    out, tot = [], 0

    def conv_cost(c, h, w):
        ho = (h + 2 * c.padding[0] - c.dilation[0] * (c.kernel_size[0] - 1) - 1) // c.stride[0] + 1
        wo = (w + 2 * c.padding[1] - c.dilation[1] * (c.kernel_size[1] - 1) - 1) // c.stride[1] + 1
        return ho * wo * c.out_channels * c.in_channels * c.kernel_size[0] * c.kernel_size[1], ho, wo

    for m in mods:
        if isinstance(m, nn.Conv2d):
            c, h, w = conv_cost(m, h, w)
            tot += c
        elif isinstance(m, nn.MaxPool2d):
            h = (h + 2 - 3) // 2 + 1
            w = (w + 2 - 3) // 2 + 1
        elif isinstance(m, Bottleneck):
            c1, _, _ = conv_cost(m.conv1, h, w)
            c2, h2, w2 = conv_cost(m.conv2, h, w)
            c3, _, _ = conv_cost(m.conv3, h2, w2)
            cd = conv_cost(m.downsample[0], h, w)[0] if m.downsample is not None else 0
            tot += c1 + c2 + c3 + cd
            h, w = h2, w2
        out.append(tot)
    return out


def head_macs(cin, num_classes, h, w, mid=256, n_atrous=3, bottleneck=None):
    pre = 0
    if bottleneck:
        pre = cin * bottleneck * h * w
        cin = bottleneck
    per_px = cin * mid + n_atrous * 9 * cin * mid + (n_atrous + 2) * mid * mid + 9 * mid * mid + mid * num_classes
    return pre + per_px * h * w + cin * mid  # + pooled 1x1 conv on a 1x1 map


class my_branch(nn.Sequential):
    """Restatement of from_deepv3_new.py:15-39: a head with its own atrous rates / width and, with `bottleneck`,
    a leading 1x1 conv (+bias, no BN, no activation)."""

    def __init__(self, nin_channels, num_classes, atrous_rates, nout_channels, bottleneck=None, **kwargs):
        rest = lambda cin: [ASPP(cin, atrous_rates, nout_channels),
                            nn.Conv2d(nout_channels, nout_channels, 3, padding=1, bias=False),
                            nn.BatchNorm2d(nout_channels), nn.ReLU(), nn.Conv2d(nout_channels, num_classes, 1)]
        if bottleneck:
            super().__init__(nn.Conv2d(nin_channels, bottleneck, 1), *rest(bottleneck))
        else:
            super().__init__(*rest(nin_channels))


class branchyDeepv3(nn.Module):
    """Restatement of from_deepv3_new.py:56-155.

    ``base_type`` contains 'resnet50' or 'resnet101'.  Weights are seeded
    random (no network).  ``split_after`` (list of block names)
    overrides the FLOP-proportional placement (F8).
    """

    def __init__(self, base_type="deeplabv3_resnet101", n=1, img_dim=256, count_branches=True,
                 skip=0, num_classes=21, split_after=None, branch_params=None):
        super().__init__()
        custom = isinstance(branch_params, dict) and all(k in branch_params for k in ("nout_channels", "atrous_rates"))

        def gen_branch(cin):              # from_deepv3_new.py:126-131
            if custom:
                return my_branch(nin_channels=cin, num_classes=num_classes, **branch_params)
            return DeepLabHead(cin, num_classes)

        def branch_cost(cin, fh):
            if custom:
                return head_macs(cin, num_classes, fh, fh, branch_params["nout_channels"],
                                 len(branch_params["atrous_rates"]), branch_params.get("bottleneck"))
            return head_macs(cin, num_classes, fh, fh)

        blocks = (3, 4, 6, 3) if re.search("resnet50", base_type) else (3, 4, 23, 3)
        backbone = _resnet_backbone(blocks)
        self.classifier = DeepLabHead(2048, num_classes)
        self.count_branches = count_branches
        mods = [m for _, m in backbone]
        cum = conv_macs(mods, img_dim)
        tot = cum[-1]
        flop_pos = tot / (n + 1)
        base_model, branches, section, names = [], [], [], []
        extra = 0
        cin = 64
        fh = None
        for (name, m), c in zip(backbone, cum):
            section.append(m)
            if isinstance(m, Bottleneck):
                cin = m.conv3.out_channels
                k = len(branches)
                cost = c + (extra if count_branches else 0)
                if split_after is not None:
                    hit = name in split_after
                else:
                    hit = (n > k) and tot > cost > flop_pos * (k + 1 + skip)
                if hit:
                    base_model.append(nn.Sequential(*section))
                    branches.append(gen_branch(cin))
                    names.append(name)
                    section = []
                    fh = self._feat_hw(name, img_dim)
                    extra += branch_cost(cin, fh)
        base_model.append(nn.Sequential(*section))
        self.base_model = nn.ModuleList(base_model)
        self.branches = nn.ModuleList(branches)
        self.n_branches = len(branches)
        self.split_names = names

    @staticmethod
    def _feat_hw(name, img_dim):
        h = (img_dim + 6 - 7) // 2 + 1
        h = (h + 2 - 3) // 2 + 1
        if not name.startswith("layer1"):
            h = (h + 2 - 3) // 2 + 1
        return h

    def forward(self, X):
        outputs = []
        inp_shape = X.shape[-2:]
        for i in range(self.n_branches):
            X = self.base_model[i](X)
            br = self.branches[i](X)
            br = F.interpolate(br, size=inp_shape, mode="bilinear", align_corners=False)
            outputs.append(br.unsqueeze(0))
        y = self.classifier(self.base_model[-1](X))
        out = F.interpolate(y, size=inp_shape, mode="bilinear", align_corners=False)
        outputs.append(out.unsqueeze(0))
        return torch.cat(outputs)
