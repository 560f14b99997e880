"""What the vendor libraries reach on THIS box for the shapes of the hot path (a calibration of the roofline, not part
of the product): hipBLASLt through torch.matmul on the implicit-GEMM shapes of the dominant convolutions, and MIOpen
through torch.nn.functional.conv2d (channels_last, bf16) on the same layers, forward only, with random and with zero
data (the gap is the clock the chip holds under MFMA load).  The native kernel is timed beside them through the C ABI.

    python tools/lib_ceiling.py > gpurun_out/lib_ceiling.json
"""
import json
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")

LAYERS = [  # (tag, B, H, Cin, Cout, stride)
    ("G 640>128@128", 32, 128, 640, 128, 1),
    ("G 128>128@128", 32, 128, 128, 128, 1),
    ("C 128>256@512", 32, 512, 128, 256, 1),
    ("C 512>1024@128", 32, 128, 512, 1024, 1),
    ("C 128>128@1024s2", 32, 1024, 128, 128, 2),
    ("C 1024>1024@128s2", 32, 128, 1024, 1024, 2),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    import os
    skip_miopen = os.environ.get("LIB_CEILING_NO_MIOPEN") == "1"
    dev = torch.device("cuda:0")
    from downgan_amd.ops import Conv, HipOps
    ops = HipOps("bf16")
    out = {"device": torch.cuda.get_device_name(0), "rows": []}
    torch.backends.cudnn.benchmark = False      # the reference leaves torch at its default (MIOpen immediate mode)
    for tag, B, H, Ci, Co, s in LAYERS:
        Ho = H // s
        flops = 2.0 * 9 * Ci * Co * B * Ho * Ho
        row = {"layer": tag, "gflop": flops / 1e9}
        for data in ("random", "zero"):
            g = torch.Generator(device=dev).manual_seed(1)
            mk = (lambda *sh: torch.randn(*sh, device=dev, generator=g).to(torch.bfloat16)) if data == "random" else \
                 (lambda *sh: torch.zeros(*sh, device=dev, dtype=torch.bfloat16))
            # hipBLASLt on the implicit-GEMM shape (M = output pixels, K = 9*Cin, N = Cout)
            M, K = B * Ho * Ho, 9 * Ci
            a, b = mk(M, K), mk(K, Co)
            ms = timeit(lambda: torch.matmul(a, b), 10)
            row[f"hipblaslt_{data}_tflops"] = round(flops / ms / 1e9, 1)
            # the same GEMM in fp8 (e4m3, per-tensor scales) where this torch build offers it: the ceiling of the MXFP8 path
            try:
                a8, b8 = a.to(torch.float8_e4m3fn), b.t().contiguous().to(torch.float8_e4m3fn)
                one = torch.ones((), device=dev, dtype=torch.float32)
                ms = timeit(lambda: torch._scaled_mm(a8, b8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16), 10)
                row[f"hipblaslt_fp8_{data}_tflops"] = round(flops / ms / 1e9, 1)
                del a8, b8
            except Exception as e:  # noqa: BLE001
                row[f"hipblaslt_fp8_{data}_tflops"] = f"unavailable: {type(e).__name__}: {str(e)[:80]}"
            del a, b
            # MIOpen (what the reference's torch.nn.Conv2d runs on this GPU), channels_last bf16
            x = mk(B, Ci, H, H).contiguous(memory_format=torch.channels_last)
            w = (mk(Co, Ci, 3, 3) * 0.05).contiguous(memory_format=torch.channels_last)
            try:
                if skip_miopen:
                    raise RuntimeError("skipped")
                ms = timeit(lambda: F.conv2d(x, w, None, stride=s, padding=1), 10)
                row[f"miopen_{data}_tflops"] = round(flops / ms / 1e9, 1)
            except Exception as e:  # noqa: BLE001
                row[f"miopen_{data}_tflops"] = f"failed: {type(e).__name__}"
            # backward of the same layer through MIOpen (data + weight gradient in one call)
            dy = mk(B, Co, Ho, Ho).contiguous(memory_format=torch.channels_last)
            try:
                if skip_miopen:
                    raise RuntimeError("skipped")
                ms = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                                        [True, True, False]), 5)
                row[f"miopen_bwd_{data}_tflops"] = round(2 * flops / ms / 1e9, 1)
            except Exception as e:  # noqa: BLE001
                row[f"miopen_bwd_{data}_tflops"] = f"failed: {type(e).__name__}"
            # the native kernels on the same layer (NHWC, weights [Cout][9][Cin]), through the C ABI
            cv = Conv(B, H, H, Ci, Co, s, False)
            xn = x.permute(0, 2, 3, 1).contiguous()
            wn = w.permute(0, 2, 3, 1).contiguous().reshape(-1)
            yn = ops.zeros(*ops.out_shape(cv))
            dyn = dy.permute(0, 2, 3, 1).contiguous()
            dxn = ops.zeros(B, H, H, Ci)
            dwn = ops.zeros(Co * 9 * Ci, dtype=torch.float32)
            ms = timeit(lambda: ops.conv_fwd(cv, xn, wn, yn, act=0.2), 10)
            row[f"native_{data}_tflops"] = round(flops / ms / 1e9, 1)
            ms = timeit(lambda: (ops.conv_dgrad(cv, dyn, wn, dxn), ops.conv_wgrad(cv, xn, dyn, dwn)), 5)
            row[f"native_bwd_{data}_tflops"] = round(2 * flops / ms / 1e9, 1)
            del x, w, dy, xn, wn, yn, dyn, dxn, dwn
            torch.cuda.empty_cache()
        out["rows"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"{time.time() - t0:.0f} s", file=sys.stderr)
