#!/usr/bin/env python3
"""Do an HBM-bound and an MFMA-bound launch of the train step overlap when they are issued on two HIP streams?

Pairs (A = HBM-/store-bound megapixel layer of the critic, B = MFMA-bound conv), batch as in configs[1] (per-GPU 32):
serial = A then B on one stream, `iters` times; overlapped = A on stream 1 and B on stream 2, same counts, joined at the end.
Prints ms per (A + B) pair for both and the ratio; also A ‖ A and B ‖ B as controls (nothing to gain there)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd.ops import Conv, HipOps


def build(o, g, N, H, ci, co, st, op, cin_real=0):
    cv = Conv(N, H, H, ci, co, st, False, cin_real=cin_real)
    x = torch.randn(N, H, H, ci, device='cuda', dtype=o.tdtype)
    w = (torch.randn(co * 9 * ci, device='cuda') * 0.05).to(o.tdtype)
    if op == "fwd":
        y = o.zeros(*o.out_shape(cv))
        return lambda: o.conv_fwd(cv, x, w, y, act=0.2)
    dy = torch.randn(*o.out_shape(cv), device='cuda', dtype=o.tdtype)
    if op == "dgrad":
        dx = o.zeros(N, H, H, ci)
        return lambda: o.conv_dgrad(cv, dy, w, dx)
    dw = o.zeros(co * 9 * ci, dtype=torch.float32)
    return lambda: o.conv_wgrad(cv, x, dy, dw)


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    o = HipOps("bf16")
    g = torch.Generator().manual_seed(0)
    B = args.batch
    ops = {
        "C.l0 fwd 16>128@1024 (store-bound)": build(o, g, B, 1024, 16, 128, 1, "fwd", cin_real=2),
        "C.l1 fwd 128>128@1024s2 (HBM)": build(o, g, B, 1024, 128, 128, 2, "fwd"),
        "C.l1 dgrad 128>128@1024s2 (HBM)": build(o, g, B, 1024, 128, 128, 2, "dgrad"),
        "C.l1 wgrad 128>128@1024s2 (HBM)": build(o, g, B, 1024, 128, 128, 2, "wgrad"),
        "G.b5 fwd 640>128@128 (MFMA)": build(o, g, B, 128, 640, 128, 1, "fwd"),
        "C.l6 fwd 512>1024@128 (MFMA)": build(o, g, B, 128, 512, 1024, 1, "fwd"),
    }
    names = list(ops)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    solo = {}
    for n in names:
        solo[n] = timed(ops[n], args.iters)
        print(f"solo  {n:40s} {solo[n]:8.3f} ms", flush=True)

    def pair(a, b, reps_a, reps_b):
        """reps_a launches of a on s1 and reps_b of b on s2, overlapped; returns wall ms."""
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        st.record(cur)
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            for _ in range(reps_a):
                ops[a]()
        with torch.cuda.stream(s2):
            for _ in range(reps_b):
                ops[b]()
        cur.wait_stream(s1); cur.wait_stream(s2)
        en.record(cur); torch.cuda.synchronize()
        return st.elapsed_time(en)

    hbm = names[:4]
    mfma = names[4:]
    for a in hbm:
        for b in mfma:
            # equal TIME on both streams: reps chosen so that both streams carry about the same serial time
            ra = args.iters
            rb = max(1, round(ra * solo[a] / solo[b]))
            serial = ra * solo[a] + rb * solo[b]
            pair(a, b, 1, 1)
            ov = pair(a, b, ra, rb)
            print(f"pair  {a:36s} x{ra} || {b:30s} x{rb}: serial {serial:8.2f} ms  overlapped {ov:8.2f} ms  ratio {ov / serial:5.3f}", flush=True)
    for a in (hbm[0], mfma[0]):
        serial = 2 * args.iters * solo[a]
        ov = pair(a, a, args.iters, args.iters)
        print(f"ctrl  {a:36s} || itself: serial {serial:8.2f} ms  overlapped {ov:8.2f} ms  ratio {ov / serial:5.3f}", flush=True)


if __name__ == "__main__":
    main()
