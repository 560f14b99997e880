import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import Conv, HipOps
hip = HipOps("bf16")
g = torch.Generator().manual_seed(21)
N, H, W, n, F = 2, 32, 32, 5, 128
slab = torch.randn(N, H, W, n * F, generator=g).to(torch.bfloat16).cuda()
us = torch.randn(N, H, W, n * F, generator=g).to(torch.bfloat16).cuda()
cvs = [Conv(N, H, W, (k + 1) * F, F) for k in range(n)]
dws = [torch.zeros(F * 9 * (k + 1) * F).cuda() for k in range(n)]; dbs = [torch.zeros(F).cuda() for _ in range(n)]
dws1 = [torch.zeros(F * 9 * (k + 1) * F).cuda() for k in range(n)]; dbs1 = [torch.zeros(F).cuda() for _ in range(n)]
hip.conv_wgrad_dense(cvs, slab, us, dws, dbs)
for k in range(n):
    hip.conv_wgrad(cvs[k], slab[..., :(k + 1) * F], us[..., k * F:(k + 1) * F], dws1[k], db=dbs1[k])
    a, b = dws[k].view(F, 9, (k + 1) * F), dws1[k].view(F, 9, (k + 1) * F)
    d = (a - b).abs()
    print("conv", k + 1, "max err", d.max().item(), "ref max", b.abs().max().item(), "db err", (dbs[k] - dbs1[k]).abs().max().item())
    print("   by input tile:", [round(d[:, :, 128 * t:128 * t + 128].max().item(), 2) for t in range(k + 1)], " by tap:", [round(d[:, t].max().item(), 1) for t in range(9)])
    print("   nonzero frac by input tile:", [round((a[:, :, 128 * t:128 * t + 128] != 0).float().mean().item(), 3) for t in range(k + 1)])
