#!/usr/bin/env python3
"""Diagnostic (DG_STAMP build): per-CU residency of the halo conv kernel's workgroups -- every workgroup of the last launch logs
{HW_ID, XCC_ID, start, end}; this prints how many workgroups a CU holds on average, the spread of their lifetimes and the idle share.

    make -C downgan_amd/csrc stamp && DG_LIB_OVERRIDE=$PWD/downgan_amd/csrc/libdowngan_hip_stamp.so python tools/wg_residency.py
"""
import ctypes as C, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import Conv, HipOps

o = HipOps("bf16")
g = torch.Generator().manual_seed(0)
CASES = [("G.b5 640->128@128 N=64", 64, 128, 640, 128, 1), ("C.l6 512->1024@128 N=32", 32, 128, 512, 1024, 1),
         ("C.l2 128->256@512 N=8", 8, 512, 128, 256, 1), ("G.b1 128->128@128 N=64", 64, 128, 128, 128, 1)]
for name, N, H, ci, co, st in CASES:
    cv = Conv(N, H, H, ci, co, st)
    x = torch.randn(N, H, H, ci, generator=g).to(o.tdtype).cuda()
    w = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(o.tdtype).cuda()
    y = o.zeros(*o.out_shape(cv))
    for _ in range(3):
        o.conv_fwd(cv, x, w, y, act=0.2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o.conv_fwd(cv, x, w, y, act=0.2); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nwg = (H // 16) * (H // 16) * N * ((co + 127) // 128)
    n = min(nwg, 16384)
    buf = (C.c_ulonglong * (3 * n))()
    o.lib.dg_debug_wglog.argtypes = [C.c_void_p, C.c_int]
    assert o.lib.dg_debug_wglog(buf, n) == 0
    per_cu = defaultdict(list)
    for i in range(n):
        hw, t0, t1 = buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]
        key = ((hw >> 32) & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
        per_cu[key].append((t0, t1))
    life = sorted(t1 - t0 for v in per_cu.values() for t0, t1 in v)
    by_xcc = defaultdict(list)
    for k, v in per_cu.items():
        by_xcc[k[0]].extend(v)
    conc, idle, two = [], [], []
    for k, v in per_cu.items():
        lo, hi = min(t for t, _ in v), max(t for _, t in v)            # the CU's own span (counters are not comparable across XCCs)
        span = hi - lo
        ev = sorted([(t0, 1) for t0, _ in v] + [(t1, -1) for _, t1 in v])
        cur, last, t_by = 0, lo, defaultdict(int)
        for t, d in ev:
            t_by[cur] += t - last
            last, cur = t, cur + d
        t_by[0] += hi - last
        conc.append(sum(t1 - t0 for t0, t1 in v) / span)
        idle.append(t_by[0] / span)
        two.append(sum(v_ for k_, v_ in t_by.items() if k_ >= 2) / span)
    spans = {k: max(t for _, t in v) - min(t for t, _ in v) for k, v in per_cu.items()}
    print("   ids seen: xcc", sorted({k[0] for k in per_cu}), "se", sorted({k[1] for k in per_cu}), "sh", sorted({k[2] for k in per_cu}), "cu", sorted({k[3] for k in per_cu}))
    print(f"{name}: {ms:.3f} ms, {n} of {nwg} workgroups on {len(per_cu)} CUs; workgroups per CU {min(len(v) for v in per_cu.values())}..{max(len(v) for v in per_cu.values())}")
    print(f"   lifetime cycles: min {life[0]} median {life[len(life)//2]} p90 {life[int(len(life)*0.9)]} max {life[-1]};  span per CU (cycles): {sorted(spans.values())[0]}..{sorted(spans.values())[-1]}"
          f"  -> {max(spans.values()) / ms / 1e6:.2f} GHz if the counter is the shader clock")
    print(f"   mean concurrent workgroups per CU {sum(conc)/len(conc):.2f} (min {min(conc):.2f}, max {max(conc):.2f}); share of the span with >=2 resident {sum(two)/len(two):.2f}, with none {sum(idle)/len(idle):.3f}")
