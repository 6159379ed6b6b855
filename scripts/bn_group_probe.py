"""BatchNorm backward of a conv3-sized tensor (B x 65 x 65 x C, bf16): two passes over the whole tensor (reduce, apply) vs
the same two passes per CHANNEL GROUP back to back (BN is per channel: a group's reduce + apply touch 2 x rows x group
bytes, which a 256-MB Infinity Cache can keep between the two).  usage: python scripts/bn_group_probe.py [B] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda")
M = B * 65 * 65
dy = (torch.randn(M, C, device=dev) * 0.1).bfloat16()
x = (torch.randn(M, C, device=dev)).bfloat16()
res = (torch.randn(M, C, device=dev)).bfloat16()
mi = torch.stack([torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
ss = torch.stack([torch.ones(C, device=dev), torch.zeros(C, device=dev)]).contiguous()
gamma = torch.ones(C, device=dev)
y, mask = K.bn_apply(x, ss, residual=res, relu=True, want_mask=True)
dx = torch.empty_like(x)
filler = torch.empty(512 << 20, dtype=torch.uint8, device=dev)     # evicts the caches between timed runs

def full():
    sums = K.bn_bwd_reduce(dy, mask, x, mi, True, scale_shift=ss)
    K.bn_bwd_apply(dy, mask, x, mi, gamma, sums, M, True, dx=dx, scale_shift=ss)

def grouped(G):
    cg = C // G
    mis = [mi[:, g * cg:(g + 1) * cg].contiguous() for g in range(G)]
    sss = [ss[:, g * cg:(g + 1) * cg].contiguous() for g in range(G)]
    def run():
        for g in range(G):
            sl = slice(g * cg, (g + 1) * cg)
            ms = mask[:, g * cg // 8:(g + 1) * cg // 8]
            sums = K.bn_bwd_reduce(dy[:, sl], ms, x[:, sl], mis[g], True, scale_shift=sss[g])
            K.bn_bwd_apply(dy[:, sl], ms, x[:, sl], mis[g], gamma[sl], sums, M, True, dx=dx[:, sl], scale_shift=sss[g])
    return run

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        filler.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best

full(); ref = dx.clone()
print(f"B={B} C={C}: tensor {M * C * 2 / 1e6:.0f} MB;  whole: {timeit(full):.1f} us")
for G in (2, 4, 8):
    if C // G < 64:
        continue
    run = grouped(G)
    run()
    ok = torch.equal(dx, ref)
    print(f"  {G} channel groups: {timeit(run):.1f} us   same bits: {ok}")
