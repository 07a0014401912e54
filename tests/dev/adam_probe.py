import torch
torch.manual_seed(0)
ps = [torch.randn(1000, device="cuda") * s for s in (1.0, 1e-3, 1e-6)]
gs = [torch.randn(1000, device="cuda") * s for s in (1.0, 1e-4, 1e-8)]
out = {}
for fused in (False, True):
    q = [p.clone().requires_grad_(True) for p in ps]
    for a, g in zip(q, gs):
        a.grad = g.clone()
    opt = torch.optim.Adam(q, lr=1e-4, betas=[0.5, 0.9], weight_decay=0.0, fused=fused)
    opt.step()
    out[fused] = [a.detach().clone() for a in q]
for i, (a, b) in enumerate(zip(out[False], out[True])):
    d = (a - b).abs().max().item()
    print(i, "max |foreach - fused| =", d, " lr = 1e-4;  max |delta| foreach", (a - ps[i]).abs().max().item(), "fused", (b - ps[i]).abs().max().item())
