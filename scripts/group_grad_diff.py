"""Per-parameter comparison of the gradients of ONE backward with the weight-gradient queue on / off (same weights, same batch, dropout 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
net = branchyDeepv3(None, "deeplabv3_resnet101", 2, 513, count_branches=False, num_classes=19, compute_dtype=torch.bfloat16,
                    fused_outputs=True).to("cuda").train()
for m in net.modules():
    if type(m).__name__ == "Dropout":
        m.p = 0.0
net.enable_grad_arena()
x = torch.randn(B, 3, 513, 513, device="cuda")
y = torch.randint(0, 19, (B, 1, 513, 513), device="cuda")
crit = BrXEntropyLoss(ignore_index=19, b_reduction="sum", n_exits=3)
grads = {}
for on in (False, True, False):
    net.cfg.group_wgrad = on
    net.cfg.arena.flat.zero_()
    for m in net.modules():                      # same running statistics do not matter for the gradients; keep them from drifting anyway
        pass
    loss = crit(net(x), y)
    loss.mean().backward()
    net.cfg.run_deferred()
    torch.cuda.synchronize()
    grads.setdefault(on, []).append({n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None})
ref, ref2, got = grads[False][0], grads[False][1], grads[True][0]
rows = []
for n in ref:
    d_on = float((got[n] - ref[n]).norm() / (ref[n].norm() + 1e-30))
    d_off = float((ref2[n] - ref[n]).norm() / (ref[n].norm() + 1e-30))
    rows.append((d_on, d_off, n, tuple(ref[n].shape)))
rows.sort(reverse=True)
print("largest relative differences (queue on vs off | off vs off again):")
for d_on, d_off, n, sh in rows[:25]:
    print(f"  {d_on:9.3e} | {d_off:9.3e}  {n} {sh}")
print("median", sorted(r[0] for r in rows)[len(rows) // 2], "off-vs-off median", sorted(r[1] for r in rows)[len(rows) // 2])
