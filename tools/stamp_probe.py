#!/usr/bin/env python3
"""Diagnostic: per-segment cycle sums of the four-wave halo kernel's tap-step (gg_halo4w_kernel, DG_STAMP build).

    make -C downgan_amd/csrc stamp && DG_LIB_OVERRIDE=downgan_amd/csrc/libdowngan_hip_stamp.so python tools/stamp_probe.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd import _lib
from downgan_amd.ops import Conv, HipOps

o = HipOps("bf16")
g = torch.Generator().manual_seed(0)
CASES = [("G.b5 640->128@128", 64, 128, 640, 128, 1, "fwd"), ("C.l6 512->1024@128", 32, 128, 512, 1024, 1, "fwd"),
         ("G.b1 128->128@128", 64, 128, 128, 128, 1, "fwd"), ("C.l1 128->128 s2 @1024 dgrad, activation mask (merged classes)", 8, 1024, 128, 128, 2, "dgrad"),
         ("C.l1 dgrad, bit mask", 8, 1024, 128, 128, 2, "dgrad_bits"), ("C.l1 dgrad, no mask", 8, 1024, 128, 128, 2, "dgrad_nomask")]
for name, N, H, ci, co, st, op in CASES:
    cv = Conv(N, H, H, ci, co, st)
    x = torch.randn(N, H, H, ci, generator=g).to(o.tdtype).cuda()
    w = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(o.tdtype).cuda()
    y = o.zeros(*o.out_shape(cv))
    if op.startswith("dgrad"):
        dy = torch.randn(*o.out_shape(cv), generator=g).to(o.tdtype).cuda()
        dx = o.zeros(N, H, H, ci)
        bits = torch.randint(-32768, 32767, o.bits_shape(dx.shape), dtype=torch.int16, device="cuda")
        kw = dict(mask=x, mask_slope=0.2) if op == "dgrad" else dict(mask_bits=bits, mask_slope=0.2) if op == "dgrad_bits" else {}
        for _ in range(3):
            o.conv_dgrad(cv, dy, w, dx, **kw)
    else:
        for _ in range(3):
            o.conv_fwd(cv, x, w, y, act=0.2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if op.startswith("dgrad"):
        o.conv_dgrad(cv, dy, w, dx, **kw)
    else:
        o.conv_fwd(cv, x, w, y, act=0.2)
    e1.record(); torch.cuda.synchronize()
    name = f"{name}  [{e0.elapsed_time(e1):.3f} ms, {o.conv_flops(cv) / e0.elapsed_time(e1) / 1e9:.0f} TFLOP/s, DG_ABL={os.environ.get('DG_ABL', '0')}]"
    buf = (C.c_ulonglong * 512)()
    o.lib.dg_debug_stamps.argtypes = [C.c_void_p]
    assert o.lib.dg_debug_stamps(buf) == 0
    print(name)
    for b in range(2):
        for wv in range(4):
            v = [buf[(b * 8 + wv) * 12 + k] for k in range(12)]
            n = max(v[4], 1)
            print(f"  blk {b} wave {wv}: steps {v[4]:3d}  per-step cycles: mma(k0)+reads k1 {v[0]/n:6.0f}  mma(k1) {v[1]/n:6.0f}  barrier {v[2]/n:6.0f}  reads k0'+DMA {v[3]/n:6.0f}"
                  f"  sum {sum(v[:4])/n:6.0f} | prologue {v[7]:7d}  loop {v[5]:8d}  epilogue {v[6]:7d} = entry {v[8]} + setup/mask loads {v[9]} + half 0 {v[10]} + half 1 {v[11]}")
