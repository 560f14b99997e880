#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernels at cfg2 channel widths (HIP events, TFLOP/s)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import Conv, HipOps

LAYERS = [  # name, N, H, Cin, Cout, stride, ps
    ("G.b1 128->128 @128", 16, 128, 128, 128, 1, False),
    ("G.b3 384->128 @128", 16, 128, 384, 128, 1, False),
    ("G.b5 640->128 @128", 16, 128, 640, 128, 1, False),
    ("G.up 128->512 @256 ps", 8, 256, 128, 512, 1, True),
    ("G.c30 128->128 @1024", 2, 1024, 128, 128, 1, False),
    ("C.l1 128->128 s2 @1024", 2, 1024, 128, 128, 2, False),
    ("C.l2 128->256 @512", 4, 512, 128, 256, 1, False),
    ("C.l3 256->256 s2 @512", 4, 512, 256, 256, 2, False),
    ("C.l4 256->512 @256", 8, 256, 256, 512, 1, False),
    ("C.l5 512->512 s2 @256", 8, 256, 512, 512, 2, False),
    ("C.l6 512->1024 @128", 16, 128, 512, 1024, 1, False),
    ("C.l7 1024->1024 s2 @128", 16, 128, 1024, 1024, 2, False),
    ("C.l0 16->128 @1024", 2, 1024, 16, 128, 1, False),
    ("G.c32 128->16 @1024", 2, 1024, 128, 16, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ops", default="fwd,dgrad,wgrad")
    ap.add_argument("--layers", default="", help="comma-separated substrings of layer names to run")
    ap.add_argument("--f8", action="store_true", help="MXFP8 kernels (operands quantised beforehand; implies --net C, bf16, wide layers only)")
    ap.add_argument("--net", default="", help='layer tag ("C" makes the wide layers eligible for the fp8 kernels with --dtype fp8)')
    ap.add_argument("--nscale", type=int, default=1, help="multiply every layer's batch (the table uses small batches)")
    args = ap.parse_args()
    o = HipOps(args.dtype, f8_critic=args.f8)
    if args.f8:
        args.net = "C"
    g = torch.Generator().manual_seed(0)
    for name, N, H, ci, co, st, ps in LAYERS:
        N *= args.nscale
        if args.layers and not any(k in name for k in args.layers.split(',')):
            continue
        cv = Conv(N, H, H, ci, co, st, ps, cin_real=(2 if ci == 16 and st == 1 else 0), net=args.net)
        x = torch.randn(N, H, H, ci, generator=g).to(o.tdtype).cuda()
        w = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(o.tdtype).cuda()
        y = o.zeros(*o.out_shape(cv))
        dy = torch.randn(*o.out_shape(cv), generator=g).to(o.tdtype).cuda()
        dx = o.zeros(N, H, H, ci)
        dw = o.zeros(co * 9 * ci, dtype=torch.float32)
        fl = o.conv_flops(cv)
        res = []
        q = {}
        if args.f8 and o.f8_eligible(cv, "fwd"):
            q["xq"], q["wq"] = o.quant_mxfp8(x), o.quant_mxfp8(w.view(co * 9, ci))
        if args.f8 and o.f8_eligible(cv, "dgrad") and not ps:
            q["dyq"], q["wdq"] = o.quant_mxfp8(dy), o.quant_mxfp8(w.view(ci * 9, co))
        for op in args.ops.split(","):
            fn = {"fwd": lambda: o.conv_fwd(cv, x, w, y, xq=q.get("xq"), wq=q.get("wq"), act=0.2),
                  "dgrad": lambda: o.conv_dgrad(cv, dy, w, dx, xq=q.get("dyq"), wq=q.get("wdq")),
                  "wgrad": lambda: o.conv_wgrad(cv, x, dy, dw)}[op]
            if op == "dgrad" and ci % 16:
                continue
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                fn()
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / args.iters
            res.append(f"{op} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF/s")
        print(f"{name:28s} " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
