"""Per-shape timing of the conv kernels on the distinct conv layers of a network
(GPU).  usage: python scripts/conv_bench.py [arch branches img batch]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3, _conv_out
from ee_semantic_segmentation_amd.nn_modules import Conv2d, Bottleneck, MaxPool2d, ASPPPooling

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
img = int(sys.argv[3]) if len(sys.argv) > 3 else 513
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dtype = torch.bfloat16
net = branchyDeepv3(None, f"deeplabv3_{arch}", nb, img, count_branches=False)
shapes = collections.Counter()

def add(conv, h, w):
    k, s, p, d = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
    shapes[(h, w, conv.in_channels, conv.out_channels, k, s, p, d)] += 1
    return _conv_out(h, k, s, p, d), _conv_out(w, k, s, p, d)

h = w = img
for s_, sec in enumerate(net.base_model):
    for m in sec:
        if isinstance(m, Conv2d):
            h, w = _conv_out(h, 7, 2, 3, 1), _conv_out(w, 7, 2, 3, 1)      # stem handled as GEMM: skip
        elif isinstance(m, MaxPool2d):
            h, w = _conv_out(h, 3, 2, 1, 1), _conv_out(w, 3, 2, 1, 1)
        elif isinstance(m, Bottleneck):
            add(m.conv1, h, w)
            h2, w2 = add(m.conv2, h, w)
            add(m.conv3, h2, w2)
            if m.downsample is not None:
                add(m.downsample[0], h, w)
            h, w = h2, w2
    head = net.branches[s_] if s_ < len(net.branches) else net.classifier
    for seq in head[0].convs:
        if not isinstance(seq, ASPPPooling):
            add(seq[0], h, w)
    add(head[0].project[0], h, w)
    add(head[1], h, w)

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

tot = collections.defaultdict(float)
print(f"{'shape (H,W,Cin,Cout,k,s,p,d)':42s} cnt   GFLOP |  fwd ms   TF |  dgrad ms   TF |  wgrad ms   TF")
for shp, cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4] ** 2):
    H, W, Cin, Cout, k, s, p, d = shp
    x = torch.randn(B, H, W, Cin, device="cuda").to(dtype)
    wt = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.05)
    wf, wb = K.pack_weight(wt, dtype)
    y, _ = K.conv_fwd(x, wf, s, p, d, want_stats=True)
    gy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Cin * k * k
    from ee_semantic_segmentation_amd._lib import lib
    best = {}
    for rnd in range(3):                     # interleaved A/B of two kernels in one process
        for pipe in (0, 3):                  # 0 = 128x128 LDS-DMA (default), 3 = 256x256 deep-pipelined where eligible
            lib().eeseg_set_option(1, pipe)
            t1 = timeit(lambda: K.conv_fwd(x, wf, s, p, d, want_stats=True))
            t2 = timeit(lambda: K.conv_dgrad(gy, wb, (H, W), s, p, d))
            best[pipe] = (min(best.get(pipe, (9, 9))[0], t1), min(best.get(pipe, (9, 9))[1], t2))
    lib().eeseg_set_option(1, 3); lib().eeseg_set_option(2, 1)        # 256-tile kernel, taps-inner K order
    ti = (min(timeit(lambda: K.conv_fwd(x, wf, s, p, d, want_stats=True)) for _ in range(2)),
          min(timeit(lambda: K.conv_dgrad(gy, wb, (H, W), s, p, d)) for _ in range(2)))
    lib().eeseg_set_option(2, 0)
    tot["fi"] += ti[0] * cnt; tot["di"] += ti[1] * cnt
    lib().eeseg_set_option(1, 0)
    tf, td = best[0]
    tot["f2"] += best[3][0] * cnt; tot["d2"] += best[3][1] * cnt
    lib().eeseg_set_wgrad_big(0)
    tw = timeit(lambda: K.conv_wgrad(x, gy, k, k, s, p, d))
    lib().eeseg_set_wgrad_big(1)
    twb = timeit(lambda: K.conv_wgrad(x, gy, k, k, s, p, d))
    lib().eeseg_set_wgrad_big(5); K.WGRAD_SLABS = True          # same kernel, K splits combined through slabs
    tws = min(timeit(lambda: K.conv_wgrad(x, gy, k, k, s, p, d)) for _ in range(2))
    lib().eeseg_set_wgrad_big(1); K.WGRAD_SLABS = False
    twb = min(twb, timeit(lambda: K.conv_wgrad(x, gy, k, k, s, p, d)))
    tot["ws"] += tws * cnt
    tot["w2"] += twb * cnt
    tot["f"] += tf * cnt; tot["d"] += td * cnt; tot["w"] += tw * cnt; tot["fl"] += fl * cnt
    print(f"{str(shp):42s} {cnt:3d} {fl/1e9:7.1f} | {tf*1e3:7.3f} {fl/tf/1e12:5.0f} | {td*1e3:8.3f} {fl/td/1e12:5.0f} | "
          f"{tw*1e3:8.3f} {fl/tw/1e12:5.0f} || big: fwd {best[3][0]*1e3:7.3f} {fl/best[3][0]/1e12:5.0f}  dgrad {best[3][1]*1e3:7.3f} {fl/best[3][1]/1e12:5.0f}  wgrad {twb*1e3:7.3f} {fl/twb/1e12:5.0f} slabs {tws*1e3:7.3f} | taps-inner fwd {ti[0]*1e3:7.3f} dgrad {ti[1]*1e3:7.3f}", flush=True)
print(f"256x256 kernel (PIPE=3) totals: fwd {tot['f2']*1e3:.2f} ms dgrad {tot['d2']*1e3:.2f} ms wgrad {tot['w2']*1e3:.2f} ms (slabs {tot['ws']*1e3:.2f})  (128-tile kernels below); taps-inner: fwd {tot['fi']*1e3:.2f} dgrad {tot['di']*1e3:.2f}")
print(f"TOTAL per step: fwd {tot['f']*1e3:.2f} ms ({tot['fl']/tot['f']/1e12:.0f} TF)  dgrad {tot['d']*1e3:.2f} ms "
      f"({tot['fl']/tot['d']/1e12:.0f} TF)  wgrad {tot['w']*1e3:.2f} ms ({tot['fl']/tot['w']/1e12:.0f} TF)")
