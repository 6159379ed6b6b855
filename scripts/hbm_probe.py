"""What this box's HBM gives to plain streaming kernels of the pointwise layers' size (graph-replayed torch ops): write only, copy, read only."""
import torch
M, C = 32 * 65 * 65, 1024
y = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
z = torch.randn(M, C, device="cuda").bfloat16()
x = torch.randn(M, 256, device="cuda").bfloat16()
s = torch.zeros(1, device="cuda")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


mb = M * C * 2 / 1e6
for name, fn, traffic in (("fill 277 MB", lambda: y.fill_(1.0), mb), ("copy 277 -> 277 MB", lambda: y.copy_(z), 2 * mb),
                          ("relu out-of-place", lambda: torch.relu(z, out=y) if False else torch.clamp_min(z, 0, out=y), 2 * mb),
                          ("read 277 MB (sum)", lambda: torch.sum(z, dtype=torch.float32), mb)):
    t = timed(fn)
    print(f"{name:22s}: {t:7.1f} us  {traffic / t * 1e3 / 1e3:5.2f} TB/s", flush=True)
