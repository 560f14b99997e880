#!/usr/bin/env python3
"""critic features.0 forward (2 real channels -> 128 at 1024^2, im2col kernel): padded [N,H,W,16] vs compact [N,H,W,2] input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import Conv, HipOps

o = HipOps("bf16")
N, H = 8, 1024
cv = Conv(N, H, H, 16, 128, 1, False, cin_real=2, net="C")
g = torch.Generator().manual_seed(0)
x = torch.zeros(N, H, H, 16, dtype=torch.bfloat16)
x[..., :2] = torch.randn(N, H, H, 2, generator=g).to(torch.bfloat16)
x = x.cuda()
x2 = x[..., :2].contiguous()
w = (torch.randn(128 * 9 * 16, generator=g) * 0.05).to(torch.bfloat16).cuda()
b = torch.randn(128).cuda()
ob = torch.zeros(o.bits_shape((N, H, H, 128)), dtype=torch.int16).cuda()
ys = []
for name, inp in (("padded", x), ("compact", x2)):
    y = o.zeros(N, H, H, 128)
    fn = (lambda: o.conv_fwd(cv, inp, w, y, bias=b, act=0.2)) if os.environ.get("L0_NOBITS") else (lambda: o.conv_fwd(cv, inp, w, y, bias=b, act=0.2, out_bits=ob))
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{name:8s} {ms:7.3f} ms  store {N * H * H * 128 * 2 / ms / 1e9:6.2f} TB/s")
    ys.append(y.clone())
print("identical:", torch.equal(ys[0], ys[1]))
