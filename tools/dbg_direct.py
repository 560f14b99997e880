import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from downgan_amd.ops import Conv, HipOps
from oracle.emu_ops import EmuOps
hip, emu = HipOps("bf16"), EmuOps("bf16")
g = torch.Generator().manual_seed(12)
N, H, W, co = 1, 32, 32, 128
cv = Conv(N, H, W, 16, co, 1, False, cin_real=2)
x = torch.zeros(N, H, W, 16, dtype=torch.bfloat16); x[..., :2] = torch.randn(N, H, W, 2, generator=g).to(torch.bfloat16)
w = torch.zeros(co, 9, 16, dtype=torch.bfloat16); w[..., :2] = (torch.randn(co, 9, 2, generator=g) * 0.3).to(torch.bfloat16)
w = w.reshape(-1)
for name, b in (("nobias", None), ("bias", torch.randn(co, generator=g))):
    y_ref = torch.zeros(N, H, W, co, dtype=torch.bfloat16)
    y = torch.ones(N, H, W, co, dtype=torch.bfloat16).cuda()
    emu.conv_fwd(cv, x, w, y_ref, bias=b)
    hip.conv_fwd(cv, x.cuda(), w.cuda(), y, bias=None if b is None else b.cuda())
    d = (y.cpu().float() - y_ref.float()).abs()[0]
    print(name, "max err", d.max().item(), "ref max", y_ref.float().abs().max().item())
    print(" err by channel block of 16:", [round(d[..., 16 * i:16 * i + 16].max().item(), 3) for i in range(8)])
    print(" err by row:", [round(d[r].max().item(), 2) for r in range(0, 32, 4)])
    print(" err by col:", [round(d[:, c].max().item(), 2) for c in range(0, 32, 2)])
    print(" sample y", y[0, 5, 5, :4].float().tolist(), "ref", y_ref[0, 5, 5, :4].float().tolist())
