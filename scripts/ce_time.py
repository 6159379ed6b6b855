import torch, sys
sys.path.insert(0, ".")
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
N, h, w, H, W, C = 16, 65, 65, 513, 513, 21
lr = torch.zeros(N, h, w, 32, device="cuda"); lr[..., :C] = torch.randn(N, h, w, C, device="cuda") * 3
t = torch.randint(0, C + 1, (N, H, W), device="cuda")
acc = torch.zeros(2, dtype=torch.float64, device="cuda"); dlr = torch.zeros_like(lr)
for o in (1, 0, 1, 0):
    lib().eeseg_set_option(6, o)
    K.upsample_ce_fwd(lr, C, t, H, W, C, acc); K.upsample_ce_bwd(lr, C, t, H, W, C, acc, 1.0, dlr); torch.cuda.synchronize()
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e0.record()
    for _ in range(20): K.upsample_ce_fwd(lr, C, t, H, W, C, acc)
    e1.record()
    for _ in range(20): K.upsample_ce_bwd(lr, C, t, H, W, C, acc, 1.0, dlr)
    e2.record(); torch.cuda.synchronize()
    print("CE_SPAN=%d  fwd %.1f us  bwd %.1f us" % (o, e0.elapsed_time(e1) * 50, e1.elapsed_time(e2) * 50))
