import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd import _lib
L = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libeeseg.so"))
B = 16
for (H, W, Cin, Cout, k) in [(65, 65, 256, 1024, 1), (65, 65, 1024, 256, 1), (65, 65, 512, 2048, 1), (65, 65, 512, 512, 3)]:
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    wt = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wf, wb = K.pack_weight(wt, torch.bfloat16)
    for _ in range(3):
        y, _s = K.conv_fwd(x, wf, 1, k // 2, 1, want_stats=True)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    assert L.eeseg_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    tiles = ((B * H * W + 255) // 256) * (Cout // 256)
    st = buf.reshape(8192, 8)[:tiles].astype(np.int64)
    t0 = st[:, 0].min()
    ph = np.diff(st[:, :7], axis=1) * 0.01            # us
    names = ["setup", "prologue-load", "main loop", "stage", "readback+stores", "store drain"]
    print(f"shape {(H, W, Cin, Cout, k)} tiles {tiles}: launch span {(st[:, 6].max() - t0) * 0.01:.1f} us")
    order = np.argsort(st[:, 0])
    for nm, col in zip(names, ph.T):
        print(f"   {nm:18s} mean {col.mean():6.2f}  p10 {np.percentile(col, 10):6.2f}  p90 {np.percentile(col, 90):6.2f} us")
    tot = (st[:, 6] - st[:, 0]) * 0.01
    print(f"   {'block total':18s} mean {tot.mean():6.2f}; start times (us) of blocks by order: "
          + " ".join(f"{(st[order[i], 0] - t0) * 0.01:.1f}" for i in range(0, tiles, max(1, tiles // 16))))
    # per-CU rounds: gap between the end of a block and the start of the next block on the same CU
    cu = st[:, 7]
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(st[idx, 0])]
        for a, b in zip(idx[:-1], idx[1:]):
            gaps.append((st[b, 0] - st[a, 6]) * 0.01)
    if gaps:
        print(f"   distinct hw ids {len(np.unique(cu))}; gap between consecutive blocks on one hw id: mean {np.mean(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} us")
