#!/usr/bin/env python3
"""bf16 wide weight-gradient kernel (wg3w) against the fp8 one (dg_conv3x3_wgrad_f8) on the critic's stride-1 layers (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import Conv, HipOps

LAYERS = [("C.l2 128->256 @512", 16, 512, 128, 256, 1), ("C.l4 256->512 @256", 32, 256, 256, 512, 1), ("C.l6 512->1024 @128", 32, 128, 512, 1024, 1),
          ("C.l1 128->128 s2 @1024", 8, 1024, 128, 128, 2), ("C.l3 256->256 s2 @512", 16, 512, 256, 256, 2), ("C.l5 512->512 s2 @256", 32, 256, 512, 512, 2),
          ("C.l7 1024->1024 s2 @128", 32, 128, 1024, 1024, 2), ("G 128->128 @128", 32, 128, 128, 128, 1), ("G 640->128 @128", 32, 128, 640, 128, 1)]
o = HipOps("bf16")
g = torch.Generator().manual_seed(0)
for name, N, H, ci, co, st in LAYERS:
    cv = Conv(N, H, H, ci, co, st)
    x = torch.randn(N, H, H, ci, generator=g).to(torch.bfloat16).cuda()
    dy = torch.randn(N, H // st, H // st, co, generator=g).to(torch.bfloat16).cuda()
    xq = (x.float().clamp(-448, 448)).to(torch.float8_e4m3fn).view(torch.uint8)
    dq = (dy.float().clamp(-448, 448)).to(torch.float8_e4m3fn).view(torch.uint8)
    ex = torch.full((ci // 32,), 127, dtype=torch.uint8).cuda(); ey = torch.full((co // 32,), 127, dtype=torch.uint8).cuda()
    dw = o.zeros(co * 9 * ci, dtype=torch.float32)
    res = []
    for tag, fn in (("bf16", lambda: o.conv_wgrad(cv, x, dy, dw)), ("fp8", lambda: o.conv_wgrad_f8(cv, xq, ex, dq, ey, dw))):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        res.append(f"{tag} {ms:7.3f} ms {o.conv_flops(cv) / ms / 1e9:7.1f} TF/s")
    print(f"{name:24s} " + " | ".join(res), flush=True)
