"""``branchyDeepv3`` - the reference's early-exit DeepLabV3 module surface
(from_deepv3_new.py:56-155; from_deepv3.py:28-125 is the same forward), on libeeseg.

Same constructor arguments, attributes (``base_model``, ``branches``, ``classifier``,
``n_branches``, ``count_branches``) and ``forward(X[B,3,H,W]) -> Tensor[E,B,C,H,W]``
contract, same ``state_dict`` key layout.  Differences, all
documented in DESIGN.md:
  * no network: weights are seeded-random unless ``base_name`` is an existing
    checkpoint of this class / a state_dict;
  * branch placement uses an analytic conv-MAC counter instead of pthflops, and
    can be pinned with ``split_after=[block names]``;
  * ``num_classes`` is a parameter (B-12), ``base_type`` is honoured (B-1);
  * ``fused_outputs=True`` makes forward return an ``ExitLogits`` (low-resolution
    logits + target size) that this package's losses / evaluators consume without
    materialising the [E,B,C,H,W] stack.
"""
import os
import re

import torch
from torch import nn

from . import engine as E
from . import kernels as K
from .nn_modules import (ASPP, ASPPPooling, BatchNorm2d, Bottleneck, Conv2d, DeepLabHead, MaxPool2d, ReLU,
                         Section)


# ------------------------------------------------------------ FLOP counter ----
def _conv_out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def conv_macs(conv, h, w):
    k, s, p, d = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
    ho, wo = _conv_out(h, k, s, p, d), _conv_out(w, k, s, p, d)
    return ho * wo * conv.out_channels * conv.in_channels * k * k, ho, wo


def block_macs(blk, h, w):
    c1, _, _ = conv_macs(blk.conv1, h, w)
    c2, h2, w2 = conv_macs(blk.conv2, h, w)
    c3, _, _ = conv_macs(blk.conv3, h2, w2)
    cd = conv_macs(blk.downsample[0], h, w)[0] if blk.downsample is not None else 0
    return c1 + c2 + c3 + cd, h2, w2


def head_macs(head, h, w):
    tot = conv_macs(head.pre, h, w)[0] if head.pre is not None else 0
    aspp = head.aspp
    for seq in aspp.convs:
        if isinstance(seq, ASPPPooling):
            tot += seq[1].in_channels * seq[1].out_channels      # pooled 1x1 conv on a 1x1 map
        else:
            tot += conv_macs(seq[0], h, w)[0]
    tot += conv_macs(aspp.project[0], h, w)[0] + conv_macs(head.conv3, h, w)[0] + conv_macs(head.cls, h, w)[0]
    return tot


def model_macs(net, H, W):
    """Forward conv MACs of a branchyDeepv3 at input H x W (all exits computed)."""
    tot, h, w = 0, H, W
    for s, sec in enumerate(net.base_model):
        for m in sec:
            if isinstance(m, Conv2d):
                c, h, w = conv_macs(m, h, w)
                tot += c
            elif isinstance(m, MaxPool2d):
                h, w = _conv_out(h, 3, 2, 1, 1), _conv_out(w, 3, 2, 1, 1)
            elif isinstance(m, Bottleneck):
                c, h, w = block_macs(m, h, w)
                tot += c
        head = net.branches[s] if s < len(net.branches) else net.classifier
        tot += head_macs(head, h, w)
    return tot


# ------------------------------------------------------------- my_branch ------
class my_branch(DeepLabHead):
    """from_deepv3_new.py:15-39: a DeepLabHead with its own atrous rates / width and, with `bottleneck`, a leading
    1x1 conv (+bias, no BN, no activation) that narrows the features first.  Child indices (= state_dict keys) are
    the reference's: [0: bottleneck conv,] ASPP, 3x3 conv, BN, ReLU, 1x1 classifier.  Any bottleneck width is
    accepted, like the reference: a width that is not a multiple of 64 is stored zero-padded (nn_modules.Conv2d),
    state_dict shapes and the MAC count stay the true ones."""

    def __init__(self, nin_channels, num_classes, atrous_rates, nout_channels, bottleneck=None, cfg=None, **kw):
        if not bottleneck:
            super().__init__(nin_channels, num_classes, tuple(atrous_rates), nout_channels, cfg=cfg)
            return
        super().__init__(bottleneck, num_classes, tuple(atrous_rates), nout_channels, cfg=cfg, pad_in_to=64)
        rest = list(self.children())
        pre = Conv2d(nin_channels, bottleneck, 1, bias=True, pad_out_to=64)
        pre.__dict__["_eeseg_role"] = "pre"
        for k in list(self._modules):
            del self._modules[k]
        for i, m in enumerate([pre] + rest):
            self.add_module(str(i), m)
        self._off = 1
        self._padded = [m for m in self.modules() if isinstance(m, Conv2d) and m.channel_padded]

    def _keep_padding_zero(self):
        # a training forward re-asserts the zero padding, so an in-place initialiser applied from outside
        # (net.apply(init_fn)) cannot turn the padding channels into real ones
        if self.training and torch.is_grad_enabled():
            for m in getattr(self, "_padded", ()):
                m.zero_channel_padding_()

    def forward(self, x):
        self._keep_padding_zero()
        return super().forward(x)

    def forward_fork(self, x):
        self._keep_padding_zero()
        return super().forward_fork(x)


# ------------------------------------------------------------ exit logits -----
class ExitLogits:
    """Low-resolution logits of every exit + the size they are upsampled to.

    ``stack()`` materialises the reference's ``[E,B,C,H,W]`` tensor; ``y[i]``
    materialises one exit.  Fused consumers use ``.lowres`` directly."""

    def __init__(self, lowres, num_classes, size, cfg=None):
        self.lowres = list(lowres)          # E x [B,h,w,32] fp32
        self.num_classes = num_classes
        self.size = tuple(size)
        self.cfg = cfg                      # engine.Config of the producing network (process group of the losses)

    def __len__(self):
        return len(self.lowres)

    @property
    def shape(self):
        return torch.Size((len(self.lowres), self.lowres[0].shape[0], self.num_classes) + self.size)

    def __getitem__(self, i):
        return upsample_logits(self.lowres[i], self.num_classes, self.size)

    def stack(self):
        return _StackFn.apply(self.num_classes, self.size[0], self.size[1], *self.lowres)


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lr, C, H, W):
        ctx.meta = (lr.shape[1], lr.shape[2], lr.shape[3])
        return K.upsample_bilinear_nchw(lr.contiguous(), C, H, W)

    @staticmethod
    def backward(ctx, dout):
        h, w, ldc = ctx.meta
        return K.upsample_bilinear_nchw_bwd(dout.contiguous().float(), h, w, ldc), None, None, None


class _StackFn(torch.autograd.Function):
    """Upsample every exit straight into one [E,B,C,H,W] buffer (from_deepv3_new.py:150-155)."""

    @staticmethod
    def forward(ctx, C, H, W, *lrs):
        B = lrs[0].shape[0]
        out = torch.empty((len(lrs), B, C, H, W), dtype=torch.float32, device=lrs[0].device)
        for e, lr in enumerate(lrs):
            K.upsample_bilinear_nchw(lr.contiguous(), C, H, W, out=out[e])
        ctx.meta = [(lr.shape[1], lr.shape[2], lr.shape[3]) for lr in lrs]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous().float()
        grads = [K.upsample_bilinear_nchw_bwd(dout[e], *m) for e, m in enumerate(ctx.meta)]
        return (None, None, None, *grads)


def upsample_logits(lr, num_classes, size):
    """F.interpolate(bilinear, align_corners=False) of low-res NHWC logits -> [B,C,H,W]
    (from_deepv3_new.py:149,152)."""
    if lr.requires_grad and torch.is_grad_enabled():
        return _UpsampleFn.apply(lr, num_classes, size[0], size[1])
    return K.upsample_bilinear_nchw(lr.detach().contiguous(), num_classes, size[0], size[1])


# --------------------------------------------------------------- backbone -----
def _kaiming_fan_out(conv):
    nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")


def _make_backbone(blocks, cfg):
    """[(name, module)] == children of torchvision's dilated ResNet backbone."""
    mods = [("conv1", Conv2d(3, 64, 7, stride=2, padding=3)), ("bn1", BatchNorm2d(64)),
            ("relu", ReLU(inplace=True)), ("maxpool", MaxPool2d(3, 2, 1))]
    st = {"inplanes": 64, "dilation": 1}

    def layer(name, planes, n, stride, dilate):
        prev = st["dilation"]
        if dilate:
            st["dilation"] *= stride
            stride = 1
        down = None
        if stride != 1 or st["inplanes"] != planes * 4:
            down = nn.Sequential(Conv2d(st["inplanes"], planes * 4, 1, stride=stride), BatchNorm2d(planes * 4))
        out = [(f"{name}.0", Bottleneck(st["inplanes"], planes, stride, down, prev, cfg=cfg))]
        st["inplanes"] = planes * 4
        for i in range(1, n):
            out.append((f"{name}.{i}", Bottleneck(st["inplanes"], planes, 1, None, st["dilation"], cfg=cfg)))
        return out

    mods += layer("layer1", 64, blocks[0], 1, False)
    mods += layer("layer2", 128, blocks[1], 2, False)
    mods += layer("layer3", 256, blocks[2], 2, True)
    mods += layer("layer4", 512, blocks[3], 2, True)
    for _, m in mods:                                   # torchvision ResNet init
        for sub in m.modules():
            if isinstance(sub, Conv2d):
                _kaiming_fan_out(sub)
    return mods


def get_base_model(name, model="deeplabv3_resnet101", pretrained=True):
    """from_deepv3_new.py:41-54.  There is no network here: returns a state_dict
    loaded from `name` when that file exists (weights_only load), else None."""
    if name and os.path.exists(name):
        obj = torch.load(name, map_location="cpu", weights_only=True)
        if isinstance(obj, dict) and "model_state_dict" in obj:
            obj = obj["model_state_dict"]
        return obj
    return None


class branchyDeepv3(nn.Module):
    def __init__(self, base_name=None, base_type="deeplabv3_resnet101", n=1, img_dim=256, count_branches=True,
                 skip=0, branch_params=None, num_classes=21, split_after=None, compute_dtype=torch.float32,
                 fused_outputs=False):
        super().__init__()
        cfg = E.Config()
        cfg.compute_dtype = compute_dtype
        self.__dict__["cfg"] = cfg
        self.count_branches = count_branches
        self.num_classes = num_classes
        self.fused_outputs = fused_outputs
        blocks = (3, 4, 6, 3) if re.search("resnet50", base_type) else (3, 4, 23, 3)
        self.base_type = base_type
        backbone = _make_backbone(blocks, cfg)
        self.classifier = DeepLabHead(2048, num_classes, cfg=cfg)

        # cumulative conv MACs after every backbone module at img_dim (stands in for
        # pthflops.count_ops, from_deepv3_new.py:68-69,99-115)
        cum, tot, h, w = [], 0, img_dim, img_dim
        for _, m in backbone:
            if isinstance(m, Conv2d):
                c, h, w = conv_macs(m, h, w)
                tot += c
            elif isinstance(m, MaxPool2d):
                h, w = _conv_out(h, 3, 2, 1, 1), _conv_out(w, 3, 2, 1, 1)
            elif isinstance(m, Bottleneck):
                c, h, w = block_macs(m, h, w)
                tot += c
            cum.append((tot, h, w))
        flop_pos = tot / (n + 1)

        base_model, branches, section, names = [], [], [], []
        extra = 0
        for (name, m), (c, fh, fw) in zip(backbone, cum):
            section.append(m)
            if not isinstance(m, Bottleneck):
                continue
            k = len(branches)
            cost = c + (extra if count_branches else 0)
            if split_after is not None:
                hit = name in split_after
            else:
                hit = (n > k) and tot > cost > flop_pos * (k + 1 + skip)      # from_deepv3_new.py:83
            if hit:
                base_model.append(Section(*section, cfg=cfg))
                cin = m.conv3.out_channels
                branches.append(self._gen_branch(cin, num_classes, branch_params, cfg))
                names.append(name)
                section = []
                extra += head_macs(branches[-1], fh, fw)
        base_model.append(Section(*section, cfg=cfg))
        self.base_model = nn.ModuleList(base_model)
        self.branches = nn.ModuleList(branches)
        self.n_branches = len(branches)
        self.split_names = names

        sd = get_base_model(base_name, base_type) if isinstance(base_name, str) else base_name
        if isinstance(sd, dict):
            self.load_state_dict(sd)

    @staticmethod
    def _gen_branch(cin, num_classes, branch_params, cfg):
        if isinstance(branch_params, dict) and all(k in branch_params for k in ("nout_channels", "atrous_rates")):
            return my_branch(nin_channels=cin, num_classes=num_classes, cfg=cfg, **branch_params)
        return DeepLabHead(cin, num_classes, cfg=cfg)

    # -- configuration ----------------------------------------------------------
    @property
    def cfg(self):
        return self.__dict__["cfg"]

    def set_compute_dtype(self, dtype):
        self.cfg.compute_dtype = dtype
        return self

    def macs(self, H, W=None):
        return model_macs(self, H, W or H)

    def enable_grad_arena(self, accumulate=False):
        """Keep every parameter gradient in one flat buffer that the backward kernels write
        directly (static addresses: needed for HIP-graph capture and zero-copy DP buckets).
        Call after .to(device).  Use this package's SGD (its zero_grad keeps the views)."""
        self.cfg.arena = E.GradArena(self)
        self.cfg.accumulate = accumulate
        return self.cfg.arena

    # -- forward ------------------------------------------------------------------
    def forward_lowres(self, X):
        """Low-res logits of every exit, shallow -> deep (final exit last)."""
        outs = []
        for i in range(self.n_branches):
            X = self.base_model[i](X)
            lo, X = self.branches[i].forward_fork(X)
            outs.append(lo)
        outs.append(self.classifier(self.base_model[-1](X)))
        return outs

    @torch.no_grad()
    def forward_progressive(self, X, tau, pool=0, pool_size=1, less_than=True, ignore=()):
        """Batched, truly progressive early-exit inference (ee_dnn_op_ne.py:51-108 evaluates one image and
        always finishes the backbone).  After every non-ignored branch the fused gate decides per image on the device;
        images that leave get their mask written at their place in the batch, the others are compacted into the leading
        slots and only those slots are computed from there on (eeseg_conv_args.n_active: conv blocks of later slots
        return at once).  Nothing is read back between sections: the host enqueues the whole network once.

        Returns {'pred': [B,H,W] int64, 'exit': [B] int32 (the reference's `n`: branch index + 1, n_branches + 1 = the
        final classifier), 'entropy': [n_branches, B] fp32 gate values by SLOT at the time of the gate} - device tensors."""
        if self.training:
            raise RuntimeError("forward_progressive is an inference path: call .eval() first")
        B, _, H, W = X.shape
        dev, C = X.device, self.num_classes
        n_active = torch.full((1,), B, dtype=torch.int32, device=dev)
        order = torch.arange(B, dtype=torch.int32, device=dev)
        src_slot = torch.zeros(B, dtype=torch.int32, device=dev)
        exit_idx = torch.zeros(B, dtype=torch.int32, device=dev)
        pred = torch.zeros((B, H, W), dtype=torch.int64, device=dev)
        ents = torch.zeros((max(self.n_branches, 1), B), dtype=torch.float32, device=dev)
        x = X
        with K.active_images(n_active):
            for i in range(self.n_branches):
                x = self.base_model[i](x)
                if i in ignore:
                    continue
                lr = self.branches[i](x).contiguous()
                ent, flag = K.entropy_gate(lr, C, H, W, tau, pool, pool_size, n_active=n_active, less_than=less_than)
                ents[i] = ent
                K.argmax_exit(lr, C, H, W, flag, order, n_active, pred)
                K.exit_select(flag, i + 1, n_active, order, src_slot, exit_idx)
                x = K.gather_images(x.contiguous(), src_slot, n_active)
            lr = self.classifier(self.base_model[-1](x)).contiguous()
            K.argmax_exit(lr, C, H, W, None, order, n_active, pred)
            K.exit_select(torch.ones(B, dtype=torch.int32, device=dev), self.n_branches + 1, n_active, order, src_slot,
                          exit_idx)
        return {"pred": pred, "exit": exit_idx, "entropy": ents}

    def forward(self, X):
        size = X.shape[-2:]
        el = ExitLogits(self.forward_lowres(X), self.num_classes, size, self.cfg)
        if self.fused_outputs:
            return el
        return el.stack()
