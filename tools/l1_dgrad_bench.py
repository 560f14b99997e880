#!/usr/bin/env python3
"""Data gradient of the critic's features.2 (128 -> 128 channels, stride 2, 1024^2 input; 4 output-parity classes = 4 launches)
at batch 8, for a per-launch view under `rocprofv3 --kernel-trace`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import Conv, HipOps

o = HipOps("bf16")
N, H = int(os.environ.get("L1_N", "8")), 1024
cv = Conv(N, H, H, 128, 128, 2, False, net="C")
g = torch.Generator().manual_seed(0)
dy = torch.randn(N, H // 2, H // 2, 128, generator=g).to(torch.bfloat16).cuda()
wd = (torch.randn(128 * 9 * 128, generator=g) * 0.05).to(torch.bfloat16).cuda()
dx = o.zeros(N, H, H, 128)
bits = torch.randint(-32768, 32767, o.bits_shape((N, H, H, 128)), dtype=torch.int16).cuda()
fn = lambda: o.conv_dgrad(cv, dy, wd, dx, mask_bits=bits, mask_slope=0.2)
fn(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    fn()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print(f"dgrad 128>128@1024s2 N={N}: {ms:.3f} ms  {2 * N * (H // 2) ** 2 * 9 * 128 * 128 / ms / 1e9:.0f} TFLOP/s  write {N * H * H * 256 / ms / 1e9:.2f} TB/s")
