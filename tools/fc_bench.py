#!/usr/bin/env python3
"""Micro-benchmark of the critic's FC1 kernels at cfg2 size (K = 1024 x 64 x 64, O = 100 -> 112, B = 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import HipOps

o = HipOps("bf16")
B, K, O = 32, 1024 * 64 * 64, 112
g = torch.Generator().manual_seed(0)
x = torch.randn(B, K, generator=g).to(o.tdtype).cuda()
w = (torch.randn(O, K, generator=g) * 0.01).to(o.tdtype).cuda()
y = torch.zeros(B, 128, device="cuda")
dy = torch.randn(B, 128, generator=g).cuda()
dx = torch.empty(B, K, dtype=o.tdtype, device="cuda")


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


wb = O * K * 2 / 1e9
t = timeit(lambda: o.linear_fwd(x, w, y))
print(f"linear_fwd  {t:.3f} ms  ({wb / t:.2f} TB/s of weights)")
t = timeit(lambda: o.linear_dx(dy, w, dx, mask=x, mask_slope=0.2))
print(f"linear_dx   {t:.3f} ms  ({wb / t:.2f} TB/s of weights)")
