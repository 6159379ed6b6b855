"""Tensor-level wrappers over the libeeseg C ABI.

Every function here launches hand-written HIP kernels on torch's current stream
and returns torch tensors that merely own the device memory.  Activations are
NHWC tensors [N,H,W,C] (possibly a channel slice of a wider buffer: the last
dim is contiguous and every other dim is dense over the row stride).
No function falls back to torch arithmetic.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F32, ConvArgs, WgradArgs, check, lib


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise _lib.EesegError(f"unsupported dtype {t.dtype}")


def _tdt(code):
    return torch.bfloat16 if code == BF16 else torch.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EesegError("libeeseg kernels need device tensors (no CPU fallback)")


def rows_ld(t):
    """(rows, C, ld) of an NHWC tensor/slice; checks the layout is row-strided."""
    C_ = t.shape[-1]
    if t.stride(-1) != 1 and C_ != 1:
        raise _lib.EesegError("channel dim must be contiguous")
    rows = t.numel() // C_
    ld, exp = None, None
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] == 1:
            continue
        if ld is None:
            ld = t.stride(d)
            exp = ld * t.shape[d]
        else:
            if t.stride(d) != exp:
                raise _lib.EesegError(
                    f"tensor is not densely row-strided: shape {tuple(t.shape)} stride {t.stride()}")
            exp *= t.shape[d]
    if ld is None:
        ld = C_
    if ld < C_:
        raise _lib.EesegError(f"row stride {ld} < channels {C_}")
    return rows, C_, ld


_ws = {}

# When set to a list, every conv launch appends (kernel family, algorithmic FLOPs,
# start event, end event) - bench.py uses it for the live roofline measurement.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _prof_end(ev, family, flops, nbytes=0.0, tag=""):
    if ev is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    PROFILE.append((family, flops, ev, end, nbytes, tag))


class _EventShare:
    """A start event that attributes only `share` of the elapsed time to its record: the members of ONE launch (a weight-gradient
    group) each get their part of the launch's time, in proportion to their FLOPs."""

    def __init__(self, ev, share):
        self.ev, self.share = ev, share

    def elapsed_time(self, end):
        return self.ev.elapsed_time(end) * self.share


_ws_retired = []     # superseded scratch buffers: a captured HIP graph may still hold their addresses


# Batched progressive inference: device int32[1] = number of leading batch slots still in flight.  While set, every conv
# launch carries it (eeseg_conv_args.n_active) and blocks of later slots return at once.
ACTIVE = None


class active_images:
    def __init__(self, n_active):
        assert n_active is None or (n_active.dtype == torch.int32 and n_active.numel() == 1 and n_active.is_cuda)
        self.t = n_active

    def __enter__(self):
        global ACTIVE
        self.prev, ACTIVE = ACTIVE, self.t
        return self.t

    def __exit__(self, *exc):
        global ACTIVE
        ACTIVE = self.prev


def workspace(nbytes, device):
    """Shared scratch of the reduction / loss / gate kernels.  It only ever grows (geometrically, so the retired
    buffers sum to less than the live one) and a superseded buffer is never freed: the address of the buffer in use at
    capture time is baked into GraphedTrainStep's graph, and a replay must not scribble over memory the caching
    allocator has handed to somebody else in the meantime."""
    key = str(device)
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _ws_retired.append(t)
        t = torch.empty(max(int(nbytes), 1 << 24, 2 * t.numel() if t is not None else 0), dtype=torch.uint8,
                        device=device)
        _ws[key] = t
    return t


def _conv_ws(device):
    """Fixed scratch of the conv kernel (split-K slabs of a partly filled last round of tiles); allocated once
    per device so its address is stable under HIP-graph capture."""
    key = ("conv", str(device))
    t = _ws.get(key)
    if t is None:
        t = torch.empty(int(lib().eeseg_conv_workspace()), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


WGRAD_SLABS = False      # True + eeseg_set_wgrad_big(.. | 4): reproducible K-split combine of the 256x256 wgrad kernel
WGRAD_COOP = __import__("os").environ.get("EESEG_WGRAD_COOP", "1") != "0"      # in-kernel combine of the K splits (round 4; A/B switch)


def _wgrad_ws(device):
    """Fixed scratch of the 256x256 weight-gradient kernel (K-split slabs); one per device and per stream role
    (the weight gradient may run on the side stream while the conv scratch is in use on the main one)."""
    key = ("wgrad", str(device))
    t = _ws.get(key)
    if t is None:
        t = torch.empty(int(lib().eeseg_wgrad_workspace()), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ------------------------------------------------------------------ conv ----
def _conv_call(x, w, y, N, Hin, Win, Cin, Hout, Wout, Cout, R, S, smul, off, tstep, sdiv, ldy,
               scale=None, shift=None, residual=None, ldres=0, stats=None, relu=False, resmask=None):
    a = ConvArgs()
    a.x, a.w, a.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    a.scale = 0 if scale is None else scale.data_ptr()
    a.shift = 0 if shift is None else shift.data_ptr()
    a.residual = 0 if residual is None else residual.data_ptr()
    a.stats = 0 if stats is None else stats.data_ptr()
    a.N, a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.Cout, a.R, a.S = N, Hin, Win, Cin, Hout, Wout, Cout, R, S
    a.smul, a.off_h, a.off_w, a.tstep_h, a.tstep_w, a.sdiv = smul, off, off, tstep, tstep, sdiv
    a.ldy, a.ldres, a.relu, a.dtype = ldy, ldres, int(relu), _dt(x)
    ws = _conv_ws(x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    a.n_active = 0 if (ACTIVE is None or stats is not None) else ACTIVE.data_ptr()
    a.residual_mask = 0 if resmask is None else resmask.data_ptr()
    a.ld_residual_mask = 0 if resmask is None else resmask.shape[-1]
    ev = _prof_begin()
    check(lib().eeseg_conv_igemm(C.byref(a), _stream()), "eeseg_conv_igemm")
    if ev is not None:
        px = N * (Hin * Win if sdiv > 1 else Hout * Wout)      # algorithmic MACs (padding taps included)
        dt = "bf16" if a.dtype == BF16 else "f32"
        fam = {1: f"conv_igemm_kernel<{dt},128>", 2: f"conv_igemm_kernel<{dt},64>",
               3: "conv_big_kernel<bf16,256x256>",     # one call = full rounds (+ K-split tail + fix-up) launches
               4: "conv_pw_kernel", 5: "conv_pws_kernel"}[lib().eeseg_last_kernel(0)]    # the kernel the library chose
        es = 2 if a.dtype == BF16 else 4       # algorithmic bytes: every operand once
        _prof_end(ev, fam, 2.0 * px * Cout * Cin * R * S,
                  float(es) * (N * Hin * Win * Cin + Cout * R * S * Cin + N * Hout * Wout * Cout),
                  f"{R}x{S} {'dgrad' if (sdiv > 1 or tstep < 0) else 'fwd'} {Cin}->{Cout} d{abs(tstep)}")


def conv_fwd(x, w, stride=1, pad=0, dil=1, *, want_stats=False, scale=None, shift=None, residual=None,
             relu=False, out=None):
    """x [N,H,W,Cin]; w packed KRSC [Cout,R,S,Cin] in x.dtype.  Returns (y, partials|None)."""
    _need_cuda(x, w)
    N, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w.shape
    assert Cin2 == Cin and x.is_contiguous() and w.is_contiguous() and w.dtype == x.dtype
    Ho, Wo = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    assert out.shape == (N, Ho, Wo, Cout) and out.dtype == x.dtype
    _, _, ldy = rows_ld(out)
    ldres = 0
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype
        _, _, ldres = rows_ld(residual)
    partials = None
    if want_stats:
        tiles = lib().eeseg_conv_stats_tiles(N, Ho, Wo)
        partials = torch.empty((tiles, 2, Cout), dtype=torch.float32, device=x.device)
    _conv_call(x, w, out, N, H, W, Cin, Ho, Wo, Cout, R, S, stride, -pad, dil, 1, ldy, scale, shift, residual,
               ldres, partials, relu)
    if partials is not None:
        partials = partials[:lib().eeseg_last_kernel(2)]      # one row per pixel tile of the kernel the library chose
    return out, partials


def masked_residual_ok(dtype, cin):
    """Can conv_dgrad(..., add=(t, mask)) run for a data-gradient with `cin` output channels?"""
    return dtype == torch.bfloat16 and cin % 256 == 0 and lib().eeseg_get_option(1) == 3


def conv_dgrad(dy, w_bwd, in_hw, stride=1, pad=0, dil=1, *, accumulate_into=None, add=None):
    """dy [N,Ho,Wo,Cout]; w_bwd packed CRSK [Cin,R,S,Cout].  Returns dx [N,H,W,Cin]
    (added into `accumulate_into` in place when given).  add=(t, mask): dx = dgrad + t * mask with `mask` the 1-bit
    ReLU mask of bn_apply (one byte per 16-byte chunk) - nothing is modified in place."""
    _need_cuda(dy, w_bwd)
    N, Ho, Wo, Cout = dy.shape
    Cin, R, S, Cout2 = w_bwd.shape
    assert Cout2 == Cout and dy.is_contiguous() and w_bwd.is_contiguous() and w_bwd.dtype == dy.dtype
    H, W = in_hw
    if add is not None:
        assert accumulate_into is None and stride == 1
        t, mask = add
        assert t.shape == (N, H, W, Cin) and t.dtype == dy.dtype and mask.dtype == torch.uint8 and mask.is_contiguous()
        assert mask.shape[-1] * 8 == Cin and mask.numel() == N * H * W * (Cin // 8)
        dx = torch.empty((N, H, W, Cin), dtype=dy.dtype, device=dy.device)
        _conv_call(dy, w_bwd, dx, N, Ho, Wo, Cout, H, W, Cin, R, S, 1, pad, -dil, stride, Cin, residual=t,
                   ldres=rows_ld(t)[2], resmask=mask)
        return dx
    if accumulate_into is not None:
        dx = accumulate_into
        assert dx.shape == (N, H, W, Cin) and dx.dtype == dy.dtype
        res = dx
    else:
        dx = torch.empty((N, H, W, Cin), dtype=dy.dtype, device=dy.device)
        res = None
    _, _, ld = rows_ld(dx)
    _conv_call(dy, w_bwd, dx, N, Ho, Wo, Cout, H, W, Cin, R, S, 1, pad, -dil, stride, ld, residual=res,
               ldres=ld if res is not None else 0)
    return dx


def multi_dgrad_ok(dtype, cin, mid, ntaps):
    """Can conv_dgrad_multi sum `ntaps` taps of data-gradients (dy channels = mid) into a cin-wide dx?"""
    return (dtype == torch.bfloat16 and cin % 256 == 0 and mid % 64 == 0 and 0 < ntaps <= 32 and
            lib().eeseg_get_option(1) == 3)


def concat_tap_weights(w_bwds):
    """[w_bwd_b: [Cin,R_b,S_b,Cout] (CRSK packed)] -> [Cin, sum R_b*S_b, Cout]: the taps of every conv side by side
    (strided row copies, one launch per conv)."""
    cin, co = w_bwds[0].shape[0], w_bwds[0].shape[3]
    T = sum(w.shape[1] * w.shape[2] for w in w_bwds)
    out = torch.empty((cin, T, co), dtype=w_bwds[0].dtype, device=w_bwds[0].device)
    es = out.element_size()
    t0 = 0
    for w in w_bwds:
        assert w.is_contiguous() and w.shape[0] == cin and w.shape[3] == co and w.dtype == out.dtype
        tb = w.shape[1] * w.shape[2]
        check(lib().eeseg_copy2d(_p(w), tb * co * es, C.c_void_p(out.data_ptr() + t0 * co * es), T * co * es, cin,
                                 tb * co * es, _stream()), "eeseg_copy2d")
        t0 += tb
    return out


def conv_dgrad_multi(dys, w_cat, geoms, *, accumulate_into=None):
    """dx = sum_b dgrad_b(dys[b]): the data-gradients of several stride-1 convs of ONE input, in one launch.
    dys [nb,N,H,W,Cout] contiguous (all convs: same output size as the input); w_cat = concat_tap_weights of their
    w_bwd; geoms = [(k, pad, dil)] per conv.  Added into `accumulate_into` in place when given."""
    _need_cuda(dys, w_cat)
    nb, N, H, W, Cout = dys.shape
    Cin, T, Cout2 = w_cat.shape
    assert dys.is_contiguous() and w_cat.is_contiguous() and Cout2 == Cout and w_cat.dtype == dys.dtype and len(geoms) == nb
    taps = []
    for b, (k, pad, dil) in enumerate(geoms):
        assert conv_out_size(H, k, 1, pad, dil) == H and conv_out_size(W, k, 1, pad, dil) == W
        off = b * N * H * W * Cout * dys.element_size()
        for r in range(k):
            for s_ in range(k):
                taps.append((pad - r * dil, pad - s_ * dil, off))      # the data-gradient's source pixel of tap (r, s)
    assert len(taps) == T
    # c_int32 wraps silently: the byte offsets (branch index x tensor bytes) must fit BEFORE they are packed
    if any(not (-2 ** 15 < t[0] < 2 ** 15 and -2 ** 15 < t[1] < 2 ** 15 and 0 <= t[2] < 2 ** 31) for t in taps):
        raise _lib.EesegError("conv_dgrad_multi: tap table entry out of range (shifts 16 bit, source offset < 2 GiB)")
    tab = (C.c_int32 * (3 * T))(*[v for t in taps for v in t])
    if accumulate_into is not None:
        dx = accumulate_into
        assert dx.shape == (N, H, W, Cin) and dx.dtype == dys.dtype
        res = dx
    else:
        dx = torch.empty((N, H, W, Cin), dtype=dys.dtype, device=dys.device)
        res = None
    _, _, ld = rows_ld(dx)
    a = ConvArgs()
    a.x, a.w, a.y = dys.data_ptr(), w_cat.data_ptr(), dx.data_ptr()
    a.scale = a.shift = a.stats = 0
    a.residual = 0 if res is None else res.data_ptr()
    a.N, a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.Cout, a.R, a.S = N, H, W, Cout, H, W, Cin, 1, T
    a.smul, a.off_h, a.off_w, a.tstep_h, a.tstep_w, a.sdiv = 1, 0, 0, 1, 1, 1
    a.ldy, a.ldres, a.relu, a.dtype = ld, (ld if res is not None else 0), 0, _dt(dys)
    ws = _conv_ws(dys.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    a.n_active = 0
    a.residual_mask, a.ld_residual_mask = 0, 0
    a.n_taps, a.taps = T, C.cast(tab, C.c_void_p).value
    ev = _prof_begin()
    check(lib().eeseg_conv_igemm(C.byref(a), _stream()), "eeseg_conv_igemm (tap table)")
    if ev is not None:
        _prof_end(ev, "conv_big_kernel<bf16,256x256>", 2.0 * N * H * W * Cin * Cout * T,
                  2.0 * (nb * N * H * W * Cout + Cin * T * Cout + N * H * W * Cin * (2 if res is not None else 1)),
                  f"3x3 dgrad-multi {Cout}x{nb}->{Cin} T{T}")
    return dx


def conv_wgrad(x, dy, R, S, stride=1, pad=0, dil=1, *, out=None, accumulate=False):
    """Returns dw fp32 KRSC [Cout,R,S,Cin]."""
    _need_cuda(x, dy)
    N, H, W, Cin = x.shape
    N2, Ho, Wo, Cout = dy.shape
    assert N2 == N and x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype
    if out is None:
        out = torch.empty((Cout, R, S, Cin), dtype=torch.float32, device=x.device)
        accumulate = False
    assert out.is_contiguous() and out.numel() == Cout * R * S * Cin and out.dtype == torch.float32
    a = WgradArgs()
    a.x, a.dy, a.dw = x.data_ptr(), dy.data_ptr(), out.data_ptr()
    a.N, a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.Cout, a.R, a.S = N, H, W, Cin, Ho, Wo, Cout, R, S
    a.stride, a.pad, a.dil, a.dtype, a.accumulate = stride, pad, dil, _dt(x), int(accumulate)
    if (WGRAD_SLABS or WGRAD_COOP) and a.dtype == BF16 and Cout % 128 == 0 and Cin % 128 == 0:
        ws = _wgrad_ws(x.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        if WGRAD_COOP:
            a.barrier_state = coop_state(x.device, "wgrad").data_ptr()
    ev = _prof_begin()
    check(lib().eeseg_conv_wgrad(C.byref(a), _stream()), "eeseg_conv_wgrad")
    if ev is not None:
        es = 2 if a.dtype == BF16 else 4
        fam = "conv_wgrad_big_kernel" if lib().eeseg_last_kernel(1) == 7 else \
            f"conv_wgrad_kernel<{'bf16' if a.dtype == BF16 else 'f32'}>"
        _prof_end(ev, fam, 2.0 * N * Ho * Wo * Cout * Cin * R * S,
                  float(es) * (N * H * W * Cin + N * Ho * Wo * Cout) + 4.0 * Cout * R * S * Cin,
                  f"{R}x{S} wgrad {Cin}->{Cout} d{dil}")
    return out


def _wgrad_args(a, x, dy, R, S, stride, pad, dil, out, accumulate):
    N, H, W, Cin = x.shape
    N2, Ho, Wo, Cout = dy.shape
    assert N2 == N and x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype
    assert out.is_contiguous() and out.numel() == Cout * R * S * Cin and out.dtype == torch.float32
    a.x, a.dy, a.dw = x.data_ptr(), dy.data_ptr(), out.data_ptr()
    a.N, a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.Cout, a.R, a.S = N, H, W, Cin, Ho, Wo, Cout, R, S
    a.stride, a.pad, a.dil, a.dtype, a.accumulate = stride, pad, dil, _dt(x), int(accumulate)
    if (WGRAD_SLABS or WGRAD_COOP) and a.dtype == BF16 and Cout % 128 == 0 and Cin % 128 == 0:
        ws = _wgrad_ws(x.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        if WGRAD_COOP:
            a.barrier_state = coop_state(x.device, "wgrad").data_ptr()
    return 2.0 * N * Ho * Wo * Cout * Cin * R * S, (2.0 if a.dtype == BF16 else 4.0) * (N * H * W * Cin + N * Ho * Wo * Cout) + 4.0 * Cout * R * S * Cin


def conv_wgrad_group(items):
    """The weight gradients of up to 4 convolutions in ONE launch where the library can (eeseg_conv_wgrad_group: small per-GPU
    shards), else one by one.  items: [(x, dy, R, S, stride, pad, dil, out, accumulate)] - `out` fp32 KRSC, written / added in place."""
    assert 1 <= len(items) <= 4
    arr = (WgradArgs * len(items))()
    fb = []
    for a, it in zip(arr, items):
        _need_cuda(it[0], it[1])
        fb.append(_wgrad_args(a, *it))
    ev = _prof_begin()
    check(lib().eeseg_conv_wgrad_group(arr, len(items), _stream()), "eeseg_conv_wgrad_group")
    if ev is not None:
        # one launch (or a run of single calls): every member gets its share of the elapsed time, by FLOPs, under its own layer tag
        fam = "conv_wgrad_group_kernel" if lib().eeseg_last_kernel(3) else "conv_wgrad (group issued one by one)"
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        total = sum(f for f, _ in fb)
        for (f, b), it in zip(fb, items):
            PROFILE.append((fam, f, _EventShare(ev, f / total), end, b, f"{it[2]}x{it[3]} wgrad {it[0].shape[-1]}->{it[1].shape[-1]} d{it[6]}"))


def pack_weight(w, dtype, cout_pad=None, want_fwd=True, want_bwd=True):
    """w: fp32 parameter [Cout,Cin,R,S] (torch default or channels_last strides).
    Returns (w_fwd [Cout_pad,R,S,Cin], w_bwd [Cin,R,S,Cout_pad]) in `dtype`."""
    _need_cuda(w)
    Cout, Cin, R, S = w.shape
    cp = cout_pad or Cout
    if w.is_contiguous():
        krsc = 1 if (R == 1 and S == 1) else 0
    elif w.is_contiguous(memory_format=torch.channels_last):
        krsc = 1
    else:
        raise _lib.EesegError("weight must be contiguous or channels_last")
    wf = torch.empty((cp, R, S, Cin), dtype=dtype, device=w.device) if want_fwd else None
    wb = torch.empty((Cin, R, S, cp), dtype=dtype, device=w.device) if want_bwd else None
    code = BF16 if dtype == torch.bfloat16 else F32
    check(lib().eeseg_pack_weight(_p(w), _p(wf), _p(wb), Cout, cp, Cin, R, S, krsc, code, _stream()),
          "eeseg_pack_weight")
    return wf, wb


def pack_weight_multi(table, n, dtype):
    code = BF16 if dtype == torch.bfloat16 else F32
    check(lib().eeseg_pack_weight_multi(_p(table), n, code, _stream()), "eeseg_pack_weight_multi")


def pack_matrix(src, rows_pad, cols_pad, dtype):
    _need_cuda(src)
    rows, cols = src.shape
    assert src.stride(1) == 1 and src.dtype == torch.float32
    dst = torch.empty((rows_pad, cols_pad), dtype=dtype, device=src.device)
    code = BF16 if dtype == torch.bfloat16 else F32
    check(lib().eeseg_pack_matrix(_p(src), rows, cols, src.stride(0), _p(dst), rows_pad, cols_pad, code, _stream()),
          "eeseg_pack_matrix")
    return dst


def im2col_nchw(x, R, S, stride, pad, kpad, dtype):
    """x [N,C,H,W] fp32 NCHW -> col [N,Ho,Wo,Kpad] in dtype."""
    _need_cuda(x)
    N, Cc, H, W = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32
    Ho, Wo = conv_out_size(H, R, stride, pad, 1), conv_out_size(W, S, stride, pad, 1)
    col = torch.empty((N, Ho, Wo, kpad), dtype=dtype, device=x.device)
    code = BF16 if dtype == torch.bfloat16 else F32
    check(lib().eeseg_im2col_nchw(_p(x), _p(col), N, Cc, H, W, R, S, stride, pad, Ho, Wo, kpad, code, _stream()),
          "eeseg_im2col_nchw")
    return col


# ------------------------------------------------------------- batchnorm ----
def reduce_partials(partials, out=None):
    """[tiles, ...] fp32 -> [...] summed over tiles (fixed order); `out`: a contiguous fp32 buffer of that shape."""
    tiles = partials.shape[0]
    kc = partials[0].numel()
    if out is None:
        out = torch.empty(partials.shape[1:], dtype=torch.float32, device=partials.device)
    assert out.is_contiguous() and out.numel() == kc and out.dtype == torch.float32
    ws = workspace(32 * kc * 4, partials.device)
    check(lib().eeseg_bn_reduce_partials(_p(partials), tiles, kc, _p(out), _p(ws), ws.numel(), _stream()),
          "eeseg_bn_reduce_partials")
    return out


def bn_finalize(sums, count, gamma, beta, eps, momentum, running_mean, running_var):
    Cc = sums.shape[1]
    mean_invstd = torch.empty((2, Cc), dtype=torch.float32, device=sums.device)
    scale_shift = torch.empty((2, Cc), dtype=torch.float32, device=sums.device)
    check(lib().eeseg_bn_finalize(_p(sums), float(count), _p(gamma), _p(beta), eps, momentum, _p(running_mean),
                                  _p(running_var), _p(mean_invstd), _p(scale_shift), Cc, _stream()),
          "eeseg_bn_finalize")
    return mean_invstd, scale_shift


def bn_reduce_finalize(partials, count, gamma, beta, eps, momentum, running_mean, running_var):
    """partials [tiles,2,C] -> (mean_invstd [2,C], scale_shift [2,C]) in one launch."""
    tiles, _, Cc = partials.shape
    mean_invstd = torch.empty((2, Cc), dtype=torch.float32, device=partials.device)
    scale_shift = torch.empty((2, Cc), dtype=torch.float32, device=partials.device)
    check(lib().eeseg_bn_reduce_finalize(_p(partials), tiles, float(count), _p(gamma), _p(beta), eps, momentum,
                                         _p(running_mean), _p(running_var), _p(mean_invstd), _p(scale_shift), Cc,
                                         _stream()), "eeseg_bn_reduce_finalize")
    return mean_invstd, scale_shift


def bn_eval_scale_shift(gamma, beta, running_mean, running_var, eps):
    Cc = running_mean.numel()
    ss = torch.empty((2, Cc), dtype=torch.float32, device=running_mean.device)
    check(lib().eeseg_bn_eval_scale_shift(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(ss), Cc,
                                          _stream()), "eeseg_bn_eval_scale_shift")
    return ss


def bn_apply(x, scale_shift, *, residual=None, relu=False, out=None, out_dtype=None, want_mask=False):
    """y = act(x*scale + shift (+ residual)).  want_mask (needs relu): also returns the ReLU byte mask
    [rows, C/epc] uint8 the backward reads instead of y (eeseg_bn_apply_relu_mask)."""
    _need_cuda(x)
    rows, Cc, ldx = rows_ld(x)
    od = out_dtype or x.dtype
    if out is None:
        out = torch.empty(x.shape, dtype=od, device=x.device)
    rows2, C2, ldy = rows_ld(out)
    assert rows2 == rows and C2 == Cc
    ldres = 0
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype
        _, _, ldres = rows_ld(residual)
    if want_mask:
        assert relu and out.dtype == x.dtype
        mask = torch.empty((rows, Cc // (16 // x.element_size())), dtype=torch.uint8, device=x.device)
        check(lib().eeseg_bn_apply_relu_mask(_p(x), ldx, _p(scale_shift), _p(residual), ldres, _p(out), ldy, _p(mask),
                                             rows, Cc, _dt(x), _stream()), "eeseg_bn_apply_relu_mask")
        return out, mask
    check(lib().eeseg_bn_apply(_p(x), ldx, _p(scale_shift), _p(residual), ldres, _p(out), ldy, rows, Cc, int(relu),
                               _dt(x), _dt(out), _stream()), "eeseg_bn_apply")
    return out


def bn_finalize_apply(x, sums, count, gamma, beta, eps, momentum, running_mean, running_var, *, residual=None, relu=False,
                      out=None, want_mask=False):
    """bn_finalize + bn_apply in one launch (SyncBN path: `sums` [2,C] come out of the collective).
    -> (y, mask | None, mean_invstd [2,C], scale_shift [2,C])."""
    _need_cuda(x, sums)
    rows, Cc, ldx = rows_ld(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    rows2, C2, ldy = rows_ld(out)
    assert rows2 == rows and C2 == Cc and out.dtype == x.dtype and sums.shape == (2, Cc) and sums.is_contiguous()
    ldres = 0
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype
        _, _, ldres = rows_ld(residual)
    mask = None
    if want_mask:
        assert relu
        mask = torch.empty((rows, Cc // (16 // x.element_size())), dtype=torch.uint8, device=x.device)
    mean_invstd = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    scale_shift = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    check(lib().eeseg_bn_finalize_apply(_p(x), ldx, _p(sums), float(count), _p(gamma), _p(beta), eps, momentum,
                                        _p(running_mean), _p(running_var), _p(mean_invstd), _p(scale_shift), _p(residual),
                                        ldres, _p(out), ldy, _p(mask), rows, Cc, int(relu), _dt(x), _stream()),
          "eeseg_bn_finalize_apply")
    return out, mask, mean_invstd, scale_shift


# One-launch BN forward (eeseg_bn_fwd_fused): OPT-IN.  Measured at 4 images per GPU (same box, graph replay): 22.34 ms per step
# with it against 21.92 ms without - every block reduces the partial sums of its channels itself (45-90 KB out of L2, a serial
# prologue of ~5 us in front of its rows), which costs more than the 5-us bn_reduce_finalize launch it removes (904 -> 791 launches).
FUSED_BN_FWD = __import__("os").environ.get("EESEG_FUSED_BN_FWD", "0") == "1"


def bn_fwd_fused_ok(x, partials):
    rows, Cc, _ = rows_ld(x)
    return FUSED_BN_FWD and x.is_cuda and bool(lib().eeseg_bn_fwd_fused_ok(rows, Cc, partials.shape[0], _dt(x)))


def bn_fwd_fused(x, partials, count, gamma, beta, eps, momentum, running_mean, running_var, *, residual=None, relu=False,
                 out=None, want_mask=False):
    """bn_reduce_finalize + bn_apply in one launch (eeseg_bn_fwd_fused) -> (y, mask | None, mean_invstd, scale_shift)."""
    _need_cuda(x, partials)
    rows, Cc, ldx = rows_ld(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    rows2, C2, ldy = rows_ld(out)
    tiles = partials.shape[0]
    assert rows2 == rows and C2 == Cc and out.dtype == x.dtype and partials.shape == (tiles, 2, Cc) and partials.is_contiguous()
    ldres = 0
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype
        _, _, ldres = rows_ld(residual)
    mask = None
    if want_mask:
        assert relu
        mask = torch.empty((rows, Cc // (16 // x.element_size())), dtype=torch.uint8, device=x.device)
    mean_invstd = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    scale_shift = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    check(lib().eeseg_bn_fwd_fused(_p(x), ldx, _p(partials), tiles, float(count), _p(gamma), _p(beta), eps, momentum,
                                   _p(running_mean), _p(running_var), _p(mean_invstd), _p(scale_shift), _p(residual), ldres,
                                   _p(out), ldy, _p(mask), rows, Cc, int(relu), _dt(x), _stream()), "eeseg_bn_fwd_fused")
    return out, mask, mean_invstd, scale_shift


def channel_stats(x):
    rows, Cc, ldx = rows_ld(x)
    sums = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    wsb = lib().eeseg_colreduce_workspace(rows, Cc)
    ws = workspace(wsb, x.device)
    check(lib().eeseg_channel_stats(_p(x), ldx, rows, Cc, _p(sums), _dt(x), _p(ws), ws.numel(), _stream()),
          "eeseg_channel_stats")
    return sums


def colsum(x, out=None):
    rows, Cc, ldx = rows_ld(x)
    if out is None:
        out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.numel() == Cc and out.dtype == torch.float32
    ws = workspace(lib().eeseg_colreduce_workspace(rows, Cc), x.device)
    check(lib().eeseg_colsum(_p(x), ldx, rows, Cc, _p(out), _dt(x), _p(ws), ws.numel(), _stream()), "eeseg_colsum")
    return out


def _relu_mode(relu, y, scale_shift):
    """0 none / 1 mask from stored y / 2 mask recomputed from x*scale+shift (no y read) / 3 byte mask of bn_apply."""
    if not relu:
        return 0
    if y is not None and y.dtype == torch.uint8:
        return 3
    return 2 if (y is None and scale_shift is not None) else 1


def bn_bwd_reduce(dy, y, x, mean_invstd, relu, out=None, scale_shift=None, copy=None):
    """-> sums [2,C] (written into `out` when given); `copy`: a second contiguous [2,C] fp32 buffer that receives the same sums."""
    rows, Cc, lddy = rows_ld(dy)
    _, _, ldx = rows_ld(x)
    ldy = rows_ld(y)[2] if y is not None else 0
    sums = out if out is not None else torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    assert sums.is_contiguous() and sums.shape == (2, Cc) and sums.dtype == torch.float32
    ws = workspace(lib().eeseg_colreduce_workspace(rows, Cc), x.device)
    assert copy is None or (copy.is_contiguous() and copy.numel() == 2 * Cc and copy.dtype == torch.float32)
    check(lib().eeseg_bn_bwd_reduce(_p(dy), lddy, _p(y), ldy, _p(x), ldx, _p(mean_invstd), _p(scale_shift), rows, Cc,
                                    _relu_mode(relu, y, scale_shift), _p(sums), _p(copy), _dt(x), _p(ws), ws.numel(),
                                    _stream()), "eeseg_bn_bwd_reduce")
    return sums


def bn_bwd_apply(dy, y, x, mean_invstd, gamma, sums, count, relu, *, want_dres=False, dx=None, scale_shift=None):
    rows, Cc, lddy = rows_ld(dy)
    _, _, ldx = rows_ld(x)
    ldy = rows_ld(y)[2] if y is not None else 0
    if dx is None:
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    lddx = rows_ld(dx)[2]
    dres = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_dres else None
    lddres = rows_ld(dres)[2] if dres is not None else 0
    check(lib().eeseg_bn_bwd_apply(_p(dy), lddy, _p(y), ldy, _p(x), ldx, _p(mean_invstd), _p(gamma), _p(sums),
                                   float(count), _p(dx), lddx, _p(dres), lddres, rows, Cc,
                                   _relu_mode(relu, y, scale_shift), _p(scale_shift), _dt(x), _stream()),
          "eeseg_bn_bwd_apply")
    return dx, dres


_coop_state = {}
COOP_BN_BWD = __import__("os").environ.get("EESEG_COOP_BN_BWD", "1") != "0"      # A/B switch of the one-launch BN backward


def coop_state(device, role="bn"):
    """EESEG_BARRIER_WORDS (8224) zeroed int32 per device and role: arrival / departure counters of the in-kernel group
    barriers (the kernels leave them zeroed) + the sticky give-up word (word 8192).  One per (device, kernel family): the
    kernels of one family all run on ONE stream of a step (the compute stream - eager on torch's current stream, or the capture
    stream of GraphedTrainStep, never both at once; the weight gradients possibly on the side stream of overlap_wgrad, hence a
    state of their own); allocated at the first eager call, i.e. before any capture."""
    key = (str(device), role)
    t = _coop_state.get(key)
    if t is None:
        t = torch.zeros(8224, dtype=torch.int32, device=device)
        _coop_state[key] = t
    return t


def coop_timeouts():
    """Number of barrier states whose give-up word is set (a launch found its grid not co-resident).  Tests assert 0."""
    return sum(int(t[8192].item() != 0) for t in _coop_state.values())


def bn_bwd_coop_ok(x):
    rows, Cc, _ = rows_ld(x)
    return COOP_BN_BWD and x.is_cuda and bool(lib().eeseg_bn_bwd_coop_ok(rows, Cc, _dt(x)))


def bn_bwd_coop(dy, y, x, mean_invstd, gamma, count, relu, *, out=None, copy=None, want_dres=False, dx=None,
                scale_shift=None):
    """bn_bwd_reduce + bn_bwd_apply in one launch (eeseg_bn_bwd_coop) -> (dx, dres, sums [2,C])."""
    rows, Cc, lddy = rows_ld(dy)
    _, _, ldx = rows_ld(x)
    ldy = rows_ld(y)[2] if y is not None else 0
    sums = out if out is not None else torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    assert sums.is_contiguous() and sums.shape == (2, Cc) and sums.dtype == torch.float32
    assert copy is None or (copy.is_contiguous() and copy.numel() == 2 * Cc and copy.dtype == torch.float32)
    if dx is None:
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    lddx = rows_ld(dx)[2]
    dres = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_dres else None
    lddres = rows_ld(dres)[2] if dres is not None else 0
    ws = workspace(lib().eeseg_bn_bwd_coop_workspace(), x.device)
    st = coop_state(x.device)
    check(lib().eeseg_bn_bwd_coop(_p(dy), lddy, _p(y), ldy, _p(x), ldx, _p(mean_invstd), _p(gamma), _p(scale_shift),
                                  float(count), _p(sums), _p(copy), _p(dx), lddx, _p(dres), lddres, rows, Cc,
                                  _relu_mode(relu, y, scale_shift), _dt(x), _p(ws), ws.numel(), _p(st), _stream()),
          "eeseg_bn_bwd_coop")
    return dx, dres, sums


def scale_act_bwd(dy, y, scale, relu, *, want_dres=False):
    rows, Cc, lddy = rows_ld(dy)
    ldy = rows_ld(y)[2] if y is not None else 0
    dx = torch.empty(dy.shape, dtype=dy.dtype, device=dy.device)
    dres = torch.empty(dy.shape, dtype=dy.dtype, device=dy.device) if want_dres else None
    check(lib().eeseg_scale_act_bwd(_p(dy), lddy, _p(y), ldy, _p(scale), _p(dx), Cc, _p(dres), Cc if want_dres else 0,
                                    rows, Cc, int(relu), _dt(dy), _stream()), "eeseg_scale_act_bwd")
    return dx, dres


# ------------------------------------------------------- pooling / misc ----
def maxpool3x3s2(x):
    N, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    check(lib().eeseg_maxpool3x3s2(_p(x), _p(y), N, H, W, Cc, Ho, Wo, _dt(x), _stream()), "eeseg_maxpool3x3s2")
    return y


def maxpool3x3s2_bwd(x, dy, y=None):
    """y = maxpool3x3s2(x) when the caller still holds it (4x fewer loads)."""
    N, H, W, Cc = x.shape
    _, Ho, Wo, _ = dy.shape
    dx = torch.empty_like(x)
    assert y is None or (y.shape == dy.shape and y.dtype == x.dtype and y.is_contiguous())
    check(lib().eeseg_maxpool3x3s2_bwd(_p(x), _p(y), _p(dy), _p(dx), N, H, W, Cc, Ho, Wo, _dt(x), _stream()),
          "eeseg_maxpool3x3s2_bwd")
    return dx


def sum_hw(x, scale=1.0):
    """x [N,H,W,C] (slice ok) -> [N,C] = scale * sum over HW."""
    N = x.shape[0]
    rows, Cc, ldx = rows_ld(x)
    y = torch.empty((N, Cc), dtype=x.dtype, device=x.device)
    ws = workspace(16 * N * Cc * 4, x.device)
    check(lib().eeseg_sum_hw(_p(x), ldx, _p(y), N, rows // N, Cc, float(scale), _dt(x), _p(ws), ws.numel(), _stream()),
          "eeseg_sum_hw")
    return y


def broadcast_hw(x, out, scale=1.0, accumulate=False):
    """x [N,C] -> out [N,H,W,C] (slice ok) (+)= scale*x."""
    N = out.shape[0]
    rows, Cc, ldy = rows_ld(out)
    assert x.shape == (N, Cc) and x.is_contiguous() and x.dtype == out.dtype
    check(lib().eeseg_broadcast_hw(_p(x), _p(out), ldy, N, rows // N, Cc, float(scale), int(accumulate), _dt(x),
                                   _stream()), "eeseg_broadcast_hw")
    return out


def dropout(x, p, seed, step_dev=None, index_offset=0):
    assert x.is_contiguous()
    y = torch.empty_like(x)
    check(lib().eeseg_dropout(_p(x), _p(y), x.numel(), float(p), int(seed) & (2 ** 64 - 1), _p(step_dev),
                              int(index_offset), _dt(x), _stream()), "eeseg_dropout")
    return y


def cast(x, dtype):
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().eeseg_cast(_p(x), _dt(x), _p(y), _dt(y), x.numel(), _stream()), "eeseg_cast")
    return y


def add_inplace(y, x):
    assert y.is_contiguous() and x.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype
    check(lib().eeseg_add_inplace(_p(y), _p(x), y.numel(), _dt(y), _stream()), "eeseg_add_inplace")
    return y


# ------------------------------------------------------ upsample / loss ----
def _lr_dims(lr):
    N, h, w, ldc = lr.shape
    assert lr.is_contiguous() and lr.dtype == torch.float32
    return N, h, w, ldc


def upsample_bilinear_nchw(lr, C_, H, W, out=None):
    N, h, w, ldc = _lr_dims(lr)
    if out is None:
        out = torch.empty((N, C_, H, W), dtype=torch.float32, device=lr.device)
    assert out.is_contiguous() and out.shape == (N, C_, H, W)
    check(lib().eeseg_upsample_bilinear_nchw(_p(lr), ldc, _p(out), N, C_, h, w, H, W, _stream()),
          "eeseg_upsample_bilinear_nchw")
    return out


def upsample_bilinear_nchw_bwd(dout, h, w, ldc):
    N, C_, H, W = dout.shape
    assert dout.is_contiguous() and dout.dtype == torch.float32
    dlr = torch.zeros((N, h, w, ldc), dtype=torch.float32, device=dout.device)
    check(lib().eeseg_upsample_bilinear_nchw_bwd(_p(dout), _p(dlr), ldc, N, C_, h, w, H, W, _stream()),
          "eeseg_upsample_bilinear_nchw_bwd")
    return dlr


def upsample_ce_fwd(lr, C_, target, H, W, ignore_index, accum):
    """accum: float64[2] device tensor, (+)= (loss_sum, valid_count)."""
    N, h, w, ldc = _lr_dims(lr)
    assert target.is_contiguous() and target.dtype == torch.int64 and target.numel() == N * H * W
    assert accum.dtype == torch.float64 and accum.numel() == 2
    check(lib().eeseg_upsample_ce_fwd(_p(lr), ldc, _p(target), N, C_, h, w, H, W, int(ignore_index), _p(accum),
                                      _stream()), "eeseg_upsample_ce_fwd")


def upsample_ce_bwd(lr, C_, target, H, W, ignore_index, accum, gscale, dlr, gscale_dev=None):
    N, h, w, ldc = _lr_dims(lr)
    assert dlr.shape == lr.shape and dlr.is_contiguous() and dlr.dtype == torch.float32
    if gscale_dev is not None:
        assert gscale_dev.dtype == torch.float32 and gscale_dev.numel() == 1 and gscale_dev.is_cuda
    check(lib().eeseg_upsample_ce_bwd(_p(lr), ldc, _p(target), N, C_, h, w, H, W, int(ignore_index), _p(accum),
                                      float(gscale), _p(gscale_dev), _p(dlr), _stream()), "eeseg_upsample_ce_bwd")


def argmax_confusion(lr, C_, target, H, W, counts=None, want_pred=False):
    N, h, w, ldc = _lr_dims(lr)
    pred = torch.empty((N, H, W), dtype=torch.int64, device=lr.device) if want_pred else None
    if target is not None:
        assert target.is_contiguous() and target.dtype == torch.int64 and target.numel() == N * H * W
        if counts is None:
            counts = torch.zeros((3, C_), dtype=torch.int32, device=lr.device)
    check(lib().eeseg_argmax_confusion(_p(lr), ldc, _p(target), N, C_, h, w, H, W, _p(counts), _p(pred), _stream()),
          "eeseg_argmax_confusion")
    return counts, pred


def class_sums_fwd(lr, C_, target, H, W, gamma=-1.0, alpha=None, alpha_batch_sum=False):
    """-> (sums [N,3,32] float64: S, I, T per image and class;  extra [N,2] float64: void pixels, focal sum)."""
    N, h, w, ldc = _lr_dims(lr)
    assert target.is_contiguous() and target.dtype == torch.int64 and target.numel() == N * H * W
    sums = torch.zeros((N, 3, 32), dtype=torch.float64, device=lr.device)
    extra = torch.zeros((N, 2), dtype=torch.float64, device=lr.device)
    check(lib().eeseg_class_sums_fwd(_p(lr), ldc, _p(target), N, C_, h, w, H, W, float(gamma), _p(alpha),
                                     int(bool(alpha_batch_sum)), _p(sums), _p(extra), _stream()), "eeseg_class_sums_fwd")
    return sums, extra


def class_sums_bwd(lr, C_, target, H, W, gS, gI, gF, dlr, gamma=-1.0, alpha=None, alpha_batch_sum=False):
    """dlr += backward of class_sums_fwd; gS / gI [N,32] fp32, gF [1] fp32 (device tensors or None)."""
    N, h, w, ldc = _lr_dims(lr)
    for g in (gS, gI):
        assert g is None or (g.shape == (N, 32) and g.dtype == torch.float32 and g.is_contiguous())
    assert dlr.shape == lr.shape and dlr.is_contiguous()
    check(lib().eeseg_class_sums_bwd(_p(lr), ldc, _p(target), N, C_, h, w, H, W, _p(gS), _p(gI), _p(gF), float(gamma),
                                     _p(alpha), int(bool(alpha_batch_sum)), _p(dlr), _stream()), "eeseg_class_sums_bwd")
    return dlr


def focal_map_fwd(lr, C_, target, H, W, gamma, alpha=None, alpha_mode=0):
    """-> (map [N,H,W] fp32, or [N,N,H,W] for alpha_mode 2; void_count int32[1])."""
    N, h, w, ldc = _lr_dims(lr)
    assert target.is_contiguous() and target.dtype == torch.int64 and target.numel() == N * H * W
    out = torch.empty((N, N, H, W) if alpha_mode == 2 else (N, H, W), dtype=torch.float32, device=lr.device)
    void = torch.zeros(1, dtype=torch.int32, device=lr.device)
    check(lib().eeseg_focal_map_fwd(_p(lr), ldc, _p(target), N, C_, h, w, H, W, float(gamma), _p(alpha), int(alpha_mode), _p(out),
                                    _p(void), _stream()), "eeseg_focal_map_fwd")
    return out, void


def focal_map_bwd(lr, C_, target, H, W, gamma, dmap, dlr, alpha=None, alpha_mode=0):
    N, h, w, ldc = _lr_dims(lr)
    assert dmap.is_contiguous() and dmap.dtype == torch.float32 and dmap.numel() == (N * N if alpha_mode == 2 else N) * H * W
    assert dlr.shape == lr.shape and dlr.is_contiguous()
    check(lib().eeseg_focal_map_bwd(_p(lr), ldc, _p(target), N, C_, h, w, H, W, float(gamma), _p(alpha), int(alpha_mode), _p(dmap),
                                    _p(dlr), _stream()), "eeseg_focal_map_bwd")
    return dlr


def argmax_pair_hist(lr_a, lr_b, C_, H, W, hist=None):
    """Per-image contingency table [N,C,C] int32 of the upsampled argmax maps of two exits."""
    N, h, w, ldc = _lr_dims(lr_a)
    assert _lr_dims(lr_b) == (N, h, w, ldc)
    if hist is None:
        hist = torch.zeros((N, C_, C_), dtype=torch.int32, device=lr_a.device)
    check(lib().eeseg_argmax_pair_hist(_p(lr_a), _p(lr_b), ldc, N, C_, h, w, H, W, _p(hist), _stream()),
          "eeseg_argmax_pair_hist")
    return hist


def ssim_labels(a, b, data_range):
    """a, b [N,H,W] int64 label maps -> [N] float64 mean SSIM (7x7 uniform window, skimage defaults)."""
    _need_cuda(a, b)
    assert a.shape == b.shape and a.dim() == 3 and a.dtype == b.dtype == torch.int64
    a, b = a.contiguous(), b.contiguous()
    N, H, W = a.shape
    out = torch.empty((N,), dtype=torch.float64, device=a.device)
    check(lib().eeseg_ssim_labels(_p(a), _p(b), N, H, W, float(data_range), _p(out), _stream()), "eeseg_ssim_labels")
    return out


def entropy_gate(lr, C_, H, W, tau, pool=0, pool_size=1, n_active=None, less_than=True):
    """-> (entropy [N] fp32, flag [N] int32 = (entropy < tau) == less_than); with `n_active` (device int32[1]) only the
    leading slots are evaluated (the others get flag 0)."""
    N, h, w, ldc = _lr_dims(lr)
    ent = torch.empty((N,), dtype=torch.float32, device=lr.device)
    flag = torch.empty((N,), dtype=torch.int32, device=lr.device)
    wsb = lib().eeseg_entropy_gate_workspace(N, H, W)
    ws = workspace(wsb, lr.device)
    check(lib().eeseg_entropy_gate_active(_p(lr), ldc, N, C_, h, w, H, W, pool, pool_size, float(tau), int(bool(less_than)),
                                          _p(n_active), _p(ent), _p(flag), _p(ws), ws.numel(), _stream()),
          "eeseg_entropy_gate_active")
    return ent, flag


def argmax_exit(lr, C_, H, W, flags, order, n_active, pred):
    """pred[order[slot]] = argmax of the upsampled logits of every active slot that leaves (flags None = all active)."""
    N, h, w, ldc = _lr_dims(lr)
    assert pred.dtype == torch.int64 and pred.is_contiguous() and pred.shape[1:] == (H, W) and order.dtype == torch.int32
    check(lib().eeseg_argmax_exit(_p(lr), ldc, N, C_, h, w, H, W, _p(flags), _p(order), _p(n_active), _p(pred), _stream()),
          "eeseg_argmax_exit")


def exit_select(flags, code, n_active, order, src_slot, exit_idx):
    """Leaving slots record `code` in exit_idx[image]; the others move to the front of `order` (in place)."""
    N = flags.numel()
    for t in (flags, n_active, order, src_slot, exit_idx):
        assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous()
    check(lib().eeseg_exit_select(_p(flags), N, int(code), _p(n_active), _p(order), _p(src_slot), _p(exit_idx), _stream()),
          "eeseg_exit_select")


def gather_images(x, src_slot, n_active):
    """x [N, ...] -> new tensor whose slot k < n_active holds image src_slot[k] (later slots: unspecified)."""
    assert x.is_contiguous()
    N = x.shape[0]
    y = torch.empty_like(x)
    check(lib().eeseg_gather_images(_p(x), _p(y), _p(src_slot), _p(n_active), N, x[0].numel() * x.element_size(), _stream()),
          "eeseg_gather_images")
    return y


def label_hist(target, C_, ignore_index):
    """-> int32 [C]: pixels labelled c (labels == ignore_index or outside [0, C) skipped)."""
    _need_cuda(target)
    assert target.is_contiguous() and target.dtype == torch.int64
    counts = torch.empty(C_, dtype=torch.int32, device=target.device)
    check(lib().eeseg_label_hist(_p(target), target.numel(), C_, int(-1 if ignore_index is None else ignore_index), _p(counts),
                                 _stream()), "eeseg_label_hist")
    return counts


def lovasz(scores, target, ignore_index, want_grad=False, gscale=1.0, gscale_dev=None, classes="present", n_label_classes=0,
           class_ids=None, norm_classes_dev=None):
    """scores [N,C,H,W] fp32 contiguous, target [N,H,W] int64.  Returns (loss[1], dscores|None).
    classes: 'present' | 'all' | list of class indices (lovaszsoftmax.py:185-188).
    n_label_classes / class_ids / norm_classes_dev: ranking a subset of the label classes - the class-sharded data-parallel
    form (eeseg.h, eeseg_lovasz): score plane c ranks label class class_ids[c]."""
    _need_cuda(scores, target)
    N, C_, H, W = scores.shape
    assert scores.is_contiguous() and scores.dtype == torch.float32
    assert target.is_contiguous() and target.dtype == torch.int64 and target.numel() == N * H * W
    P = N * H * W
    wsb = lib().eeseg_lovasz_workspace(P, C_)
    if wsb < 0:
        raise _lib.EesegError("lovasz: N*H*W*C must be < 2^31")
    ws = workspace(wsb, scores.device)
    loss = torch.empty(1, dtype=torch.float32, device=scores.device)
    ds = torch.empty_like(scores) if want_grad else None
    if isinstance(classes, str):
        if classes not in ("present", "all"):
            raise _lib.EesegError(f"lovasz: classes={classes!r} (expected 'present', 'all' or a list of class indices)")
        mask, present_only = (1 << C_) - 1, int(classes == "present")
    else:
        mask, present_only = 0, 0
        for c in classes:
            if not 0 <= int(c) < C_:
                raise _lib.EesegError(f"lovasz: class {c} outside [0, {C_})")
            mask |= 1 << int(c)
    ids = None
    if class_ids is not None:
        assert len(class_ids) == C_
        ids = (C.c_int32 * C_)(*[int(c) for c in class_ids])
    check(lib().eeseg_lovasz(_p(scores), _p(target), N, C_, H * W, int(-1 if ignore_index is None else ignore_index),
                             _p(loss), _p(ds), float(gscale), _p(gscale_dev), mask, present_only, int(n_label_classes),
                             ids, _p(norm_classes_dev), _p(ws), ws.numel(), _stream()), "eeseg_lovasz")
    return loss, ds
