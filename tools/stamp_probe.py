#!/usr/bin/env python3
"""Diagnostic: per-segment cycle sums of the halo kernel's tap-step.

    make -C downgan_amd/csrc stamp && DG_LIB_OVERRIDE=downgan_amd/csrc/libdowngan_hip_stamp.so python tools/stamp_probe.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd import _lib
from downgan_amd.ops import Conv, HipOps

o = HipOps("bf16")
g = torch.Generator().manual_seed(0)
for name, N, H, ci, co in [("G.b5 640->128@128", 16, 128, 640, 128), ("C.l6 512->1024@128", 16, 128, 512, 1024), ("G.b1 128->128@128", 16, 128, 128, 128)]:
    cv = Conv(N, H, H, ci, co)
    x = torch.randn(N, H, H, ci, generator=g).to(o.tdtype).cuda()
    w = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(o.tdtype).cuda()
    y = o.zeros(*o.out_shape(cv))
    for _ in range(3):
        o.conv_fwd(cv, x, w, y, act=0.2)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 512)()
    o.lib.dg_debug_stamps.argtypes = [C.c_void_p]
    assert o.lib.dg_debug_stamps(buf) == 0
    print(name)
    for b in range(2):
        for wv in range(8):
            v = [buf[(b * 8 + wv) * 8 + k] for k in range(8)]
            n = max(v[4], 1)
            print(f"  blk {b} wave {wv}: steps {v[4]:3d}  per-step cycles: first half (mma0,1 issued + reads 2,3 issued) {v[0]/n:7.0f}  barrier {v[1]/n:7.0f}  second half {v[2]/n:6.0f}"
                  f"  sum {sum(v[:4])/n:7.0f} | loop {v[5]:8d}  epilogue {v[6]:7d}")
