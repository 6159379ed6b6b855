"""nn.Module surface of the torchvision pieces the reference composes
(SURVEY.md Appendix A), running on libeeseg.  Parameter / buffer names and shapes
match torchvision exactly, so reference checkpoints load key-for-key:
``conv1.weight``, ``bn1.{weight,bias,running_mean,running_var,num_batches_tracked}``,
``downsample.{0,1}.*``, ``convs.{i}.{0,1}.*``, ``project.{0,1}.*`` ...

Activations between these modules are NHWC tensors in the network's compute
dtype; the stem takes the NCHW fp32 image, a head returns low-resolution fp32
logits [N,h,w,32] (classes padded to one 128-byte row).
"""
import torch
from torch import nn

from . import engine as E


# ---------------------------------------------------------------- holders ----
class Conv2d(nn.Module):
    """Parameter holder with torch.nn.Conv2d's attributes; weight is stored
    channels_last (= KRSC, the kernels' layout) so weight gradients need no copy.

    ``pad_to``: the conv kernels move K in 128-byte steps, so channel counts that are not a multiple of 64 are
    STORED zero-padded up to one (``weight``/``bias`` hold the padded tensors, the padding rows/columns are exact
    zeros and stay zero under SGD: their activations and gradients are 0).  ``in_channels``/``out_channels``, the
    default initialisation, the MAC counter and ``state_dict`` (save and load) keep the true sizes, i.e. the
    reference's (my_branch(bottleneck=...) accepts any width, from_deepv3_new.py:23-24)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, bias=False, pad_in_to=1, pad_out_to=1):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride = (k, k), (stride, stride)
        self.padding, self.dilation = (padding, padding), (dilation, dilation)
        self.cin_stored = -(-cin // pad_in_to) * pad_in_to
        self.cout_stored = -(-cout // pad_out_to) * pad_out_to
        w = torch.empty(self.cout_stored, self.cin_stored, k, k).contiguous(memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(self.cout_stored)) if bias else None
        self.reset_parameters()

    @property
    def channel_padded(self):
        return self.cin_stored != self.in_channels or self.cout_stored != self.out_channels

    def reset_parameters(self):   # torch.nn.Conv2d default init (heads keep it: SURVEY F7)
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        with torch.no_grad():
            bound = (6.0 / ((1 + 5) * fan_in)) ** 0.5            # kaiming_uniform_(a=sqrt(5)) on the TRUE fan-in
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-1 / fan_in ** 0.5, 1 / fan_in ** 0.5)
        self.zero_channel_padding_()

    @torch.no_grad()
    def zero_channel_padding_(self):
        """Restore the invariant after anything wrote the stored tensors in place (an external init)."""
        if self.cout_stored != self.out_channels:
            self.weight[self.out_channels:].zero_()
            if self.bias is not None:
                self.bias[self.out_channels:].zero_()
        if self.cin_stored != self.in_channels:
            self.weight[:, self.in_channels:].zero_()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.channel_padded:                                # the reference's shapes
            destination[prefix + "weight"] = destination[prefix + "weight"][:self.out_channels, :self.in_channels]
            if self.bias is not None:
                destination[prefix + "bias"] = destination[prefix + "bias"][:self.out_channels]

    def _load_from_state_dict(self, state_dict, prefix, *args):
        if self.channel_padded:
            for name, true in (("weight", (self.out_channels, self.in_channels)), ("bias", (self.out_channels,))):
                t = state_dict.get(prefix + name)
                if t is not None and tuple(t.shape[:len(true)]) == true:
                    full = torch.zeros(getattr(self, name).shape, dtype=t.dtype, device=t.device)
                    full[tuple(slice(0, n) for n in true)] = t
                    state_dict[prefix + name] = full
        super()._load_from_state_dict(state_dict, prefix, *args)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}, bias={self.bias is not None}"
                + (f", stored as {self.cin_stored}->{self.cout_stored}" if self.channel_padded else ""))

    def forward(self, x):
        raise RuntimeError("Conv2d is fused by its parent block (stem / Bottleneck / DeepLabHead)")


class BatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"

    def forward(self, x):
        raise RuntimeError("BatchNorm2d is fused by its parent block")


class ReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        raise RuntimeError("ReLU is fused by its parent block")


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size=3, stride=2, padding=1):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        raise RuntimeError("MaxPool2d is fused into the stem")


class AdaptiveAvgPool2d(nn.Module):
    def __init__(self, output_size=1):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        raise RuntimeError("AdaptiveAvgPool2d is fused into the ASPP head")


class Dropout(nn.Module):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, x):
        raise RuntimeError("Dropout is fused into the ASPP head")


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ----------------------------------------------------------------- stem ------
class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, w, g, b, conv, bn, cfg, section, frozen=False):
        out, st = E.stem_fwd(cfg, img, conv, bn, not frozen, frozen=frozen)
        ctx.st, ctx.mods, ctx.cfg, ctx.section = st, (conv, bn), cfg, section
        return out

    @staticmethod
    def backward(ctx, dout):
        conv, bn = ctx.mods
        dw, dg, db = E.stem_bwd(ctx.cfg, ctx.st, dout.contiguous(), conv, bn)
        ctx.st = None
        ctx.cfg.unit_done(ctx.section)
        return None, dw, dg, db, None, None, None, None, None


def run_stem(cfg, img, conv, bn, train, section=None):
    if _needs_grad(conv.weight, bn.weight, bn.bias):       # eval mode + autograd = frozen statistics
        return _StemFn.apply(img, conv.weight, bn.weight, bn.bias, conv, bn, cfg, section, not train)
    with torch.no_grad():
        return E.stem_fwd(cfg, img, conv, bn, train)[0]


# ------------------------------------------------------------ bottleneck -----
class _BottleneckFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, cfg, *params):
        out, st = E.bottleneck_fwd(cfg, x, blk, blk.training, frozen=not blk.training)
        ctx.st, ctx.blk, ctx.cfg = st, blk, cfg
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, grads = E.bottleneck_bwd(ctx.cfg, ctx.st, dout.contiguous(), ctx.blk)
        ctx.st = None
        return (dx, None, None, *grads)


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (Appendix A.1)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, cfg=None):
        super().__init__()
        w = planes
        self.conv1 = Conv2d(inplanes, w, 1)
        self.bn1 = BatchNorm2d(w)
        self.conv2 = Conv2d(w, w, 3, stride=stride, padding=dilation, dilation=dilation)
        self.bn2 = BatchNorm2d(w)
        self.conv3 = Conv2d(w, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.__dict__["cfg"] = cfg

    def param_list(self):
        ps = [self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias,
              self.conv3.weight, self.bn3.weight, self.bn3.bias]
        if self.downsample is not None:
            ps += [self.downsample[0].weight, self.downsample[1].weight, self.downsample[1].bias]
        return ps

    def forward(self, x):
        cfg = self.__dict__["cfg"]
        ps = self.param_list()
        if _needs_grad(x, *ps):        # .eval() + autograd: BatchNorm uses (and keeps) its running statistics
            return _BottleneckFn.apply(x, self, cfg, *ps)
        with torch.no_grad():
            return E.bottleneck_fwd(cfg, x, self, self.training)[0]


# ------------------------------------------------------------------ head -----
class ASPPConv(nn.Sequential):
    def __init__(self, cin, cout, dilation, pad_in_to=1):
        super().__init__(Conv2d(cin, cout, 3, padding=dilation, dilation=dilation, pad_in_to=pad_in_to),
                         BatchNorm2d(cout), ReLU())


class ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout, pad_in_to=1):
        super().__init__(AdaptiveAvgPool2d(1), Conv2d(cin, cout, 1, pad_in_to=pad_in_to), BatchNorm2d(cout), ReLU())


class ASPP(nn.Module):
    def __init__(self, cin, atrous_rates=(12, 24, 36), cout=256, pad_in_to=1):
        super().__init__()
        mods = [nn.Sequential(Conv2d(cin, cout, 1, pad_in_to=pad_in_to), BatchNorm2d(cout), ReLU())]
        for r in atrous_rates:
            mods.append(ASPPConv(cin, cout, r, pad_in_to))
        mods.append(ASPPPooling(cin, cout, pad_in_to))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(Conv2d(len(mods) * cout, cout, 1), BatchNorm2d(cout), ReLU(), Dropout(0.5))

    def forward(self, x):
        raise RuntimeError("ASPP is fused by DeepLabHead")


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, head, cfg, *params):
        logits, st = E.head_fwd(cfg, x, head, head.training, frozen=not head.training)
        ctx.st, ctx.head, ctx.cfg = st, head, cfg
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dx, grads = E.head_bwd(ctx.cfg, ctx.st, dlogits.contiguous(), ctx.head)
        ctx.st = None
        return (dx, None, None, *grads)


class _HeadForkFn(torch.autograd.Function):
    """A branch head at a split point: returns (logits, x) where x goes on into the next backbone section.
    Backward therefore receives the next section's gradient of x TOGETHER with dlogits and the head's first
    data-gradient conv adds into it through its epilogue; with two separate consumers autograd would sum the two
    [N,h,w,C] gradients with a torch elementwise kernel (3 tensor passes) inside the training step."""

    @staticmethod
    def forward(ctx, x, head, cfg, *params):
        logits, st = E.head_fwd(cfg, x, head, head.training, frozen=not head.training)
        ctx.st, ctx.head, ctx.cfg = st, head, cfg
        return logits, x.view_as(x)

    @staticmethod
    def backward(ctx, dlogits, dx_next):
        if dx_next is not None:
            dx_next = dx_next.contiguous()
        dx, grads = E.head_bwd(ctx.cfg, ctx.st, dlogits.contiguous(), ctx.head, dx_init=dx_next)
        ctx.st = None
        return (dx, None, None, *grads)


class DeepLabHead(nn.Sequential):
    """torchvision DeepLabHead (Appendix A.2): ASPP -> 3x3 -> BN -> ReLU -> 1x1(+bias).
    Returns low-resolution logits [N,h,w,32] fp32 (first `num_classes` channels valid)."""

    def __init__(self, cin, num_classes, atrous_rates=(12, 24, 36), mid=256, cfg=None, pad_in_to=1):
        if num_classes > E.CPAD:
            raise ValueError(f"at most {E.CPAD} classes are supported by the fused loss kernels")
        super().__init__(ASPP(cin, atrous_rates, mid, pad_in_to), Conv2d(mid, mid, 3, padding=1), BatchNorm2d(mid), ReLU(),
                         Conv2d(mid, num_classes, 1, bias=True))
        self.__dict__["cfg"] = cfg
        self.num_classes = num_classes

    # structural accessors (the engine and the gradient arena go through these, so a head with a leading
    # bottleneck conv - my_branch(bottleneck=...) - only shifts the Sequential indices)
    _off = 0

    @property
    def pre(self):
        return self[0] if self._off else None

    @property
    def aspp(self):
        return self[self._off]

    @property
    def conv3(self):
        return self[self._off + 1]

    @property
    def bn3(self):
        return self[self._off + 2]

    @property
    def cls(self):
        return self[self._off + 4]

    def param_list(self):
        ps = []
        aspp = self.aspp
        for i, seq in enumerate(aspp.convs):
            conv, bn = (seq[1], seq[2]) if isinstance(seq, ASPPPooling) else (seq[0], seq[1])
            ps += [conv.weight, bn.weight, bn.bias]
        ps += [aspp.project[0].weight, aspp.project[1].weight, aspp.project[1].bias,
               self.conv3.weight, self.bn3.weight, self.bn3.bias, self.cls.weight, self.cls.bias]
        if self.pre is not None:
            ps += [self.pre.weight, self.pre.bias]
        return ps

    def forward(self, x):
        cfg = self.__dict__["cfg"]
        ps = self.param_list()
        if _needs_grad(x, *ps):        # .eval() + autograd: frozen statistics, no dropout
            return _HeadFn.apply(x, self, cfg, *ps)
        with torch.no_grad():
            return E.head_fwd(cfg, x, self, self.training)[0]

    def forward_fork(self, x):
        """-> (logits, x to feed the next backbone section).  Same numbers as ``(self(x), x)``; in a training
        step the two gradients of x are then summed inside the head's data-gradient kernel (_HeadForkFn)."""
        cfg = self.__dict__["cfg"]
        ps = self.param_list()
        if _needs_grad(x, *ps) and x.requires_grad:
            return _HeadForkFn.apply(x, self, cfg, *ps)
        return self(x), x


class Section(nn.Sequential):
    """One backbone section (from_deepv3_new.py:84,90).  Section 0 starts with the
    stem modules [conv1, bn1, relu, maxpool] which run as one fused stem."""

    def __init__(self, *mods, cfg=None):
        super().__init__(*mods)
        self.__dict__["cfg"] = cfg

    def forward(self, x):
        cfg = self.__dict__["cfg"]
        mods = list(self)
        i = 0
        if mods and isinstance(mods[0], Conv2d):
            x = run_stem(cfg, x, mods[0], mods[1], self.training and mods[1].training, self)
            i = 4
        for m in mods[i:]:
            x = m(x)
        return x
