"""Does block-count quantisation limit the conv kernels?  Time one shape at several batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (H, W, Cin, Cout, k, s, p, d) in [(65, 65, 2048, 256, 3, 1, 12, 12), (65, 65, 2048, 256, 1, 1, 0, 1), (65, 65, 256, 256, 3, 1, 2, 2), (65, 65, 128, 128, 3, 1, 1, 1)]:
    wt = torch.randn(Cout, Cin, k, k, device="cuda") * 0.02
    wf, wb = K.pack_weight(wt, torch.bfloat16)
    for B in (14, 15, 16, 17, 23, 24, 31, 32):
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        M = B * H * W
        nblk = ((M + 127) // 128) * ((Cout + 127) // 128)
        t = timeit(lambda: K.conv_fwd(x, wf, s, p, d, want_stats=True))
        fl = 2.0 * M * Cout * Cin * k * k
        print(f"{(H,W,Cin,Cout,k,d)} B={B:2d} blocks={nblk:5d} rounds@512={nblk/512:5.2f} t={t:7.3f} ms  {fl/t/1e9:6.0f} TF  t/img={t/B*1e3:6.1f} us")
