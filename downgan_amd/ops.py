"""Tensor-level wrappers over the C ABI (include/downgan_hip.h) — the only compute backend.

Every op takes torch tensors that live on the GPU (torch is used for device memory and streams
only), checks shapes/layouts on the host, and launches the HIP kernel on torch's current stream.
Activations are NHWC tensors ``[N, H, W, C]`` (or channel-slice views of a wider slab) with unit
channel stride; ``ld`` is the pixel stride in elements.

There is deliberately no CPU implementation here.  ``oracle/emu_ops.py`` (test infrastructure)
implements the same op contracts with torch-CPU so that the host-side engine logic can be tested
without a GPU; the product never imports it.
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import os
import torch

from . import _lib
from ._lib import ConvGeom, Epilogue, check

TORCH_DTYPE = {"f32": torch.float32, "bf16": torch.bfloat16}
DG_DTYPE = {"f32": _lib.DG_F32, "bf16": _lib.DG_BF16}


@dataclasses.dataclass(frozen=True)
class Conv:
    """One reference nn.Conv2d(Cin, Cout, 3, stride, 1) layer in native (padded) terms."""
    N: int
    H: int
    W: int
    Cin: int          # padded
    Cout: int         # padded
    stride: int = 1
    pixel_shuffle: bool = False
    cin_real: int = 0   # real (unpadded) input channels when that is <= 2, else 0
    net: str = ""       # "C" for critic layers: bench.py reports the critic conv stack's MFMA utilisation separately
    cin_alg: int = 0    # real channel counts of the reference layer (0 = same as padded): ALGORITHMIC flops / bytes only,
    cout_alg: int = 0   # zero padding is not counted as work

    @property
    def Ho(self):
        return self.H // self.stride

    @property
    def Wo(self):
        return self.W // self.stride


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def pix_layout(t):
    """(ld, rows) of an NHWC tensor/view whose pixels are laid out densely with stride ld."""
    assert t.dim() == 4 and t.stride(3) == 1, (t.shape, t.stride())
    ld = t.stride(2)
    n, h, w, _ = t.shape
    assert (w == 1 or t.stride(2) == ld) and (h == 1 or t.stride(1) == w * ld) and (n == 1 or t.stride(0) == h * w * ld), \
        (t.shape, t.stride())
    return ld, n * h * w


# The deterministic-mode workspace is ONE per process, like the library's switch (csrc/debug.hip): every HipOps(deterministic=True)
# holds a reference; the buffer is registered once and deregistered when the last holder closes -- a process normally has several
# op objects alive (the losses' cache, the network modules, one per engine re-bind), and collecting an old one must not switch the
# mode off under the live engine.
_DET = {"ws": None, "refs": 0}


def _det_acquire(lib, device, nbytes):
    if _DET["ws"] is None or _DET["ws"].numel() < nbytes or _DET["ws"].device != torch.device(device):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)          # nothing in flight may still be writing partials into the buffer being replaced
        check(lib.dg_set_deterministic_workspace(C.c_void_p(ws.data_ptr()), ws.numel()), "dg_set_deterministic_workspace")
        _DET["ws"] = ws
    _DET["refs"] += 1


def _det_release(lib):
    _DET["refs"] = max(0, _DET["refs"] - 1)
    if _DET["refs"] == 0 and _DET["ws"] is not None:
        lib.dg_set_deterministic_workspace(C.c_void_p(0), 0)
        _DET["ws"] = None


class HipOps:
    """The HIP backend.  ``dtype`` is the storage/compute precision of activations and weight packs."""

    name = "hip"

    def __init__(self, dtype="bf16", device="cuda:0", f8_critic=False, f8_generator=False, deterministic=None, det_workspace_mb=512):
        """``deterministic`` (default: the environment variable DG_DETERMINISTIC): run-to-run bit-identical results.  The split-K
        weight gradients and the small reductions accumulate with fp32 atomics whose order varies between runs; in this mode they
        write their partials into a workspace this object owns (``det_workspace_mb``) and add them in a fixed order
        (csrc/debug.hip, dg_set_deterministic_workspace).  The switch is process-wide in the library: the workspace is shared and
        reference-counted over all HipOps created with ``deterministic=True``; the mode ends when the last of them is closed /
        collected.  ``ops.deterministic`` reads the library's state.

        ``f8_critic``: MXFP8 conv path (BASELINE configs[4]) -- forward and data-gradient convs of critic layers whose
        reduction channels are a multiple of 128 run on the block-scaled fp8 MFMA: their bf16 source tensor and weight pack are
        quantised on the fly (``dg_quant_mxfp8``) unless the caller passes producer-written forms, accumulation is fp32, outputs /
        masks / weight gradients stay bf16.  ``f8_generator`` (implies ``f8_critic``): also the FORWARD of the generator's
        dense-block trunk convs (generator.py:24-41; layers tagged net="G")."""
        assert dtype in TORCH_DTYPE
        if not torch.cuda.is_available():
            raise RuntimeError("downgan_amd.ops.HipOps needs a ROCm GPU (no CPU fallback exists)")
        self.dtype = dtype
        self.tdtype = TORCH_DTYPE[dtype]
        self.dg = DG_DTYPE[dtype]
        self.device = torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.lib = _lib.lib()
        self.prof = None      # optional list of (tag, flops, bytes, start_event, end_event): bench.py's live kernel timing
        self.f8 = bool(f8_critic or f8_generator)
        self.f8_generator = bool(f8_generator)
        self.f8_wgrad = os.environ.get("DG_NO_F8_WGRAD") is None     # fp8 weight gradients of the eligible critic layers (f8 mode only)
        self.f8_gbwd = os.environ.get("DG_NO_F8_GBWD") is None       # f8_generator: also the dense blocks' data gradients on the MXFP8 kernel
        self.f8_gwgrad = os.environ.get("DG_NO_F8_GWGRAD") is None   # ... and their weight gradients on the fp8 kernel (uniform-scale slab copies)
        self.f8_gtail = os.environ.get("DG_NO_F8_GTAIL") is None     # f8_generator: also the forward of the up-sampling tail (upsampling.*, conv3.0)
        self.f8_l0u = os.environ.get("DG_NO_F8_L0U") is None         # fp8 mode: the critic's first layer writes the uniform-scale copy of its output alone
        assert not self.f8 or dtype == "bf16", "the fp8 conv path quantises bf16 tensors"
        self._f8_scratch = {}
        if deterministic is None:
            deterministic = os.environ.get("DG_DETERMINISTIC", "0") not in ("", "0")
        self._det_held = False
        if deterministic:
            _det_acquire(self.lib, self.device, int(det_workspace_mb) << 20)
            self._det_held = True
        self._finite_counts = None

    @property
    def deterministic(self):
        """Whether the LIBRARY is in deterministic mode: the switch is process-wide, so an op object created without the flag
        is deterministic too while any other holds the workspace."""
        return bool(self.lib.dg_deterministic())

    def close(self):
        """Drop this object's reference to the deterministic-mode workspace (no-op otherwise); the library leaves the mode when
        the last holder has gone."""
        if self._det_held:
            self._det_held = False
            _det_release(self.lib)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count_nonfinite(self, bufs):
        """bufs: [(name, tensor)] (<= 8; fp32 or the compute dtype, contiguous storage ranges) -> {name: number of NaN / Inf elements},
        ONE launch and one read-back (dg_count_nonfinite): the per-iteration stand-in for set_detect_anomaly (wasserstein.py:13)."""
        assert 1 <= len(bufs) <= _lib.FINITE_MAX
        f = _lib.FiniteBufs(nbuf=len(bufs))
        for i, (name, t) in enumerate(bufs):
            assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16), (name, t.dtype)
            f.ptr[i], f.n[i], f.dtype[i] = t.data_ptr(), t.numel(), (_lib.DG_F32 if t.dtype == torch.float32 else _lib.DG_BF16)
        if self._finite_counts is None:
            self._finite_counts = torch.zeros(_lib.FINITE_MAX, dtype=torch.int32, device=self.device)
        check(self.lib.dg_count_nonfinite(C.byref(f), _ptr(self._finite_counts), self._stream()), "dg_count_nonfinite")
        counts = self._finite_counts.cpu().tolist()
        return {name: int(counts[i]) for i, (name, _) in enumerate(bufs)}

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        # the C ABI launches on the CURRENT HIP device (function attributes are per device): keep it equal to this backend's
        if torch.cuda.current_device() != self._dev_index:
            torch.cuda.set_device(self.device)
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    per_layer = False     # bench.py --per-layer: tag conv launches with their layer geometry

    def _timed(self, tag, flops, fn, nbytes=0.0, net="", geom=""):
        """Run one launch; when profiling is on, bracket it with HIP events on the launch stream."""
        if self.prof is None:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn()
        e.record()
        if self.per_layer and geom:
            tag = f"{tag}:{geom}"
        if net:
            tag = f"{tag}:{net}"
        if tag.startswith(("conv_fwd", "conv_dgrad")):
            tag = f"{tag}:k{self.lib.dg_last_conv_kernels()}"   # which kernel variant(s) served the call
        self.prof.append((tag, flops, nbytes, s, e))
        return rc

    @staticmethod
    def _geom_tag(cv):
        return f"{cv.Cin}>{cv.Cout}@{cv.H}" + ("s2" if cv.stride == 2 else "") + ("ps" if cv.pixel_shuffle else "")

    @staticmethod
    def conv_flops(cv):
        """algorithmic flops of one conv pass: 2*9*Cin*Cout*Ho*Wo per image with the REFERENCE layer's
        channel counts -- the zero-padded channels the kernels also multiply are not work."""
        return 2.0 * 9 * (cv.cin_alg or cv.Cin) * (cv.cout_alg or cv.Cout) * cv.Ho * cv.Wo * cv.N

    def conv_bytes(self, cv, ep=None):
        """algorithmic HBM bytes of one conv pass: input + output + weights once each, plus the output-sized
        epilogue operands (residuals, activation mask, accumulate) that are passed."""
        es = 2 if self.dtype == "bf16" else 4
        out = cv.N * cv.Ho * cv.Wo * cv.Cout
        extra = 0 if ep is None else sum(1 for k in ("r1", "r2") if ep.get(k) is not None) + int(bool(ep.get("accumulate")))
        if ep is not None and ep.get("mask") is not None:      # only the channels from mask_c0 up are read
            extra += 1.0 - float(ep.get("mask_c0", 0)) / cv.Cin if "mask_c0" in ep else 1
        return float(es * (cv.N * cv.H * cv.W * cv.Cin + out * (1 + extra) + 9 * cv.Cin * cv.Cout))

    def _act(self, t):
        assert t.dtype == self.tdtype and t.is_cuda, (t.dtype, t.device)
        return t

    def _geom(self, cv: Conv, ldx, ldy):
        return ConvGeom(dtype=self.dg, N=cv.N, H=cv.H, W=cv.W, Cin=cv.Cin, Cout=cv.Cout, stride=cv.stride,
                        cin_real=cv.cin_real, pixel_shuffle=int(cv.pixel_shuffle), ldx=ldx, ldy=ldy)

    @staticmethod
    def bits_shape(shape):
        """Shape of the 1-bit mask tensor (int16) of an NHWC activation tensor: [N, H, W, C/64, 4]."""
        N, H, W, Cc = shape
        assert Cc % 64 == 0 and Cc >= 128, Cc
        return (N, H, W, Cc // 64, 4)

    def _epilogue(self, out, bias=None, act=None, r1=None, s1=1.0, r2=None, s2=1.0, mask=None, mask_slope=1.0,
                  accumulate=False, mask_bits=None, out_bits=None, mask_c0=0, mask_last=False, out_q=None, out_u=None, skip_y=False,
                  out_amax=None):
        ep = Epilogue()
        ep.bias = bias.data_ptr() if bias is not None else None
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.numel() >= out.shape[-1]
        ep.has_act = int(act is not None)
        ep.act_slope = float(act) if act is not None else 1.0
        for name, t, s in (("r1", r1, s1), ("r2", r2, s2)):
            if t is not None:
                self._act(t)
                assert t.shape == out.shape, (name, t.shape, out.shape)
                setattr(ep, name, t.data_ptr())
                setattr(ep, "ld" + name, pix_layout(t)[0])
            setattr(ep, "s" + name[1], float(s))
        if mask is not None:
            self._act(mask)
            assert mask.shape == out.shape, (mask.shape, out.shape)
            ep.mask = mask.data_ptr()
            ep.ldmask = pix_layout(mask)[0]
        ep.mask_slope = float(mask_slope)
        assert mask_c0 % 16 == 0 and (mask is not None or (mask_c0 == 0 and not mask_last))
        ep.mask_c0 = int(mask_c0)
        ep.mask_last = int(bool(mask_last))
        ep.accumulate = int(accumulate)
        for name, t in (("mask_bits", mask_bits), ("out_bits", out_bits)):
            if t is not None:
                assert t.dtype == torch.int16 and t.is_cuda and t.is_contiguous() and tuple(t.shape) == self.bits_shape(out.shape), (name, t.shape, out.shape)
                setattr(ep, name, t.data_ptr())
        assert not (mask is not None and mask_bits is not None)
        if out_q is not None:       # (q, scales): MXFP8 copy of the stored output -- views with `out`'s shape and strides
            q, qs = out_q             # (a channel slice of a slab's fp8 form when `out` is a slice of the slab)
            Cc = out.shape[-1]
            assert out.dtype == torch.bfloat16 and Cc % 64 == 0 and Cc >= 128
            assert q.dtype == torch.uint8 and qs.dtype == torch.uint8 and q.shape == out.shape and q.stride() == out.stride(), (q.shape, q.stride(), out.stride())
            assert tuple(qs.shape) == tuple(out.shape[:-1]) + (Cc // 32,) and qs.stride(-1) == 1
            assert all(qs.stride(i) * out.stride(2) == out.stride(i) * qs.stride(2) for i in range(3)), (qs.stride(), out.stride())
            ep.out_q, ep.out_qs, ep.ldqs = q.data_ptr(), qs.data_ptr(), qs.stride(2)
        if out_u is not None:       # (u8, exps): the uniform-scale E4M3 copy for the fp8 weight gradient (one exponent per 32-channel block)
            u, ue = out_u           # (without out_q: first-layer launches only -- the library refuses the others)
            assert u.dtype == torch.uint8 and u.shape == out.shape and u.stride() == out.stride()
            assert ue.dtype == torch.uint8 and ue.is_contiguous() and ue.numel() == out.shape[-1] // 32
            ep.out_u, ep.out_ue = u.data_ptr(), ue.data_ptr()
        if skip_y:                  # the bf16 output is not stored: only its fp8 copies / mask bits (nobody reads the tensor itself)
            assert (out_q is not None or out_u is not None) and not accumulate
            ep.skip_y = 1
        if out_amax is not None:    # census of the largest stored magnitude per 32-channel block (uint32 bit patterns; first-layer launches)
            assert out_amax.dtype == torch.int32 and out_amax.is_cuda and out_amax.is_contiguous() and out_amax.numel() == out.shape[-1] // 32
            ep.out_amax = out_amax.data_ptr()
        return ep

    # ------------------------------------------------------------------ conv family
    def out_shape(self, cv: Conv):
        if cv.pixel_shuffle:
            return (cv.N, 2 * cv.Ho, 2 * cv.Wo, cv.Cout // 4)
        return (cv.N, cv.Ho, cv.Wo, cv.Cout)

    def f8_eligible(self, cv: Conv, kind):
        """Does this layer's forward ("fwd") / data gradient ("dgrad") run on the MXFP8 kernel in f8 mode, its weight gradient
        ("wgrad") on the fp8 kernel with uniform-scale operands (dg_conv3x3_wgrad_f8: channel counts that are multiples of 128,
        output rows of a multiple of 64 pixels)?"""
        if kind == "wgrad":
            return (self.f8 and self.f8_wgrad and cv.net == "C" and not cv.pixel_shuffle and cv.Cin % 128 == 0
                    and cv.Cout % 128 == 0 and cv.Wo % 64 == 0)
        cred, nout = (cv.Cin, cv.Cout) if kind == "fwd" else (cv.Cout, cv.Cin)
        # generator layers (net "G": the dense-block trunk): forward always in f8_generator mode, data gradients when f8_gbwd is on;
        # net "T": the up-sampling tail's forward (pixel-shuffled outputs included) when f8_gtail is on
        nets = ("C", "G") if (self.f8_generator and (kind == "fwd" or self.f8_gbwd)) else ("C",)
        if self.f8_generator and self.f8_gtail and kind == "fwd" and cv.net == "T":
            return cred % 128 == 0 and nout > 64 and (not cv.pixel_shuffle or (cv.Cout // 4) % 32 == 0)
        return self.f8 and cv.net in nets and cred % 128 == 0 and nout > 64 and not cv.pixel_shuffle

    def conv_fwd(self, cv: Conv, x, w_fwd, y, xq=None, wq=None, **ep):
        """``xq`` / ``wq`` (f8 mode, optional): (q, scales) MXFP8 forms of ``x`` / ``w_fwd`` that a producer already wrote
        (``out_q=`` of the previous layer's launch, ``quant_mxfp8`` of the weight pack after an optimizer step); without them
        the operands are quantised on the fly."""
        self._act(x); self._act(y); self._act(w_fwd)
        # layers with <= 2 real input channels (im2col kernel) also take the COMPACT form [N, H, W, cin_real] of their input
        assert tuple(x.shape) == (cv.N, cv.H, cv.W, cv.Cin) or (cv.cin_real and tuple(x.shape) == (cv.N, cv.H, cv.W, 2) and cv.stride == 1), (x.shape, cv)
        assert tuple(y.shape) == self.out_shape(cv), (y.shape, cv)
        assert w_fwd.numel() == cv.Cout * 9 * cv.Cin and w_fwd.is_contiguous()
        g = self._geom(cv, pix_layout(x)[0], pix_layout(y)[0])
        e = self._epilogue(y, **ep)
        if self.f8_eligible(cv, "fwd"):
            check(self._timed("conv_fwd", self.conv_flops(cv), lambda: self._conv_f8(self.lib.dg_conv3x3_fwd_f8, g, e, x, w_fwd, cv.Cout, cv.Cin, y, xq, wq),
                              self.conv_bytes(cv, ep), cv.net, self._geom_tag(cv)), "dg_conv3x3_fwd_f8")
            return
        check(self._timed("conv_fwd", self.conv_flops(cv), lambda: self.lib.dg_conv3x3_fwd(
            C.byref(g), C.byref(e), _ptr(x), _ptr(w_fwd), _ptr(y), self._stream()), self.conv_bytes(cv, ep), cv.net, self._geom_tag(cv)), "dg_conv3x3_fwd")

    def conv_dgrad(self, cv: Conv, dy, w_dgrad, dx, xq=None, wq=None, **ep):
        self._act(dy); self._act(dx); self._act(w_dgrad)
        assert tuple(dx.shape) == (cv.N, cv.H, cv.W, cv.Cin), (dx.shape, cv)
        assert tuple(dy.shape) == self.out_shape(cv), (dy.shape, cv)
        assert w_dgrad.numel() == cv.Cout * 9 * cv.Cin and w_dgrad.is_contiguous()
        g = self._geom(cv, pix_layout(dx)[0], pix_layout(dy)[0])
        e = self._epilogue(dx, **ep)
        if self.f8_eligible(cv, "dgrad"):
            check(self._timed("conv_dgrad", self.conv_flops(cv), lambda: self._conv_f8(self.lib.dg_conv3x3_dgrad_f8, g, e, dy, w_dgrad, cv.Cin, cv.Cout, dx, xq, wq),
                              self.conv_bytes(cv, ep), cv.net, self._geom_tag(cv)), "dg_conv3x3_dgrad_f8")
            return
        check(self._timed("conv_dgrad", self.conv_flops(cv), lambda: self.lib.dg_conv3x3_dgrad(
            C.byref(g), C.byref(e), _ptr(dy), _ptr(w_dgrad), _ptr(dx), self._stream()), self.conv_bytes(cv, ep), cv.net, self._geom_tag(cv)), "dg_conv3x3_dgrad")

    # ------------------------------------------------------------------ MXFP8 conv path
    def quant_mxfp8(self, src, q=None, scales=None):
        """src: [..., C] (bf16 or fp32, unit channel stride; 4-D NHWC with dense pixels, possibly a channel slice of a slab, or
        2-D rows) -> (q uint8 of src's shape, scales uint8 [..., C/32]): MXFP8, blocks of 32 consecutive channels
        (csrc/quant.hip).  ``q`` / ``scales`` may be views of wider tensors (a slab's fp8 form)."""
        Cc = src.shape[-1]
        assert Cc % 128 == 0 and src.stride(-1) == 1 and src.is_cuda and src.dtype in (torch.float32, torch.bfloat16)
        if src.dim() == 4:
            ld, rows = pix_layout(src)
        else:
            assert src.dim() == 2
            ld, rows = src.stride(0), src.shape[0]
        if q is None:
            q = torch.empty(src.shape, dtype=torch.uint8, device=self.device)
            scales = torch.empty(tuple(src.shape[:-1]) + (Cc // 32,), dtype=torch.uint8, device=self.device)
        assert q.dtype == torch.uint8 and scales.dtype == torch.uint8 and q.shape == src.shape and q.stride(-1) == 1 and scales.stride(-1) == 1
        assert tuple(scales.shape) == tuple(src.shape[:-1]) + (Cc // 32,)
        ldq, ldqs = q.stride(-2), scales.stride(-2)
        if src.dim() == 4:
            assert pix_layout(q)[1] == rows and all(scales.stride(i) * ldq == q.stride(i) * ldqs for i in range(3))
        sdt = _lib.DG_F32 if src.dtype == torch.float32 else _lib.DG_BF16
        check(self.lib.dg_quant_mxfp8(sdt, _ptr(src), rows, ld, Cc, _ptr(q), ldq, _ptr(scales), ldqs, self._stream()), "dg_quant_mxfp8")
        return q, scales

    def block_exp_max(self, scales, out, margin=1):
        """out[b] (uint8 [C / 32]) = min(254, max over pixels of the MXFP8 scale bytes of block b + margin): the exponents the
        NEXT pass's uniform-scale copy of this tensor is written with (dg_block_exp_max)."""
        self.block_exp_max_batch([(scales, out)], margin)

    def block_exp_max_batch(self, pairs, margin=1):
        """The same for a list of (scales, out) pairs in one launch per 16 tensors (dg_block_exp_max_batch)."""
        if self._exp_scratch is None:
            self._exp_scratch = torch.zeros(64 * _lib.EXP_BATCH_MAX, dtype=torch.int32, device=self.device)
        for k in range(0, len(pairs), _lib.EXP_BATCH_MAX):
            b = _lib.ExpBatch()
            chunk = pairs[k:k + _lib.EXP_BATCH_MAX]
            b.n = len(chunk)
            nbytes = 0
            for i, (scales, out) in enumerate(chunk):
                nb = scales.shape[-1]
                assert scales.dtype == torch.uint8 and out.dtype == torch.uint8 and out.numel() == nb and scales.stride(-1) == 1
                assert scales.dim() == 2 or all(scales.stride(d) == scales.stride(d + 1) * scales.shape[d + 1] for d in range(scales.dim() - 2)), scales.stride()
                b.scales[i], b.rows[i], b.ld[i], b.nblocks[i], b.out[i] = scales.data_ptr(), scales.numel() // nb, scales.stride(-2), nb, out.data_ptr()
                nbytes += scales.numel()
            check(self._timed("ew_exp_max", 0.0, lambda: self.lib.dg_block_exp_max_batch(
                C.byref(b), int(margin), _ptr(self._exp_scratch), self._stream()), float(nbytes), "C"), "dg_block_exp_max_batch")

    _exp_scratch = None

    def exp_from_amax(self, amax, out, margin=1):
        """out[b] (uint8) = the exponent of the uniform-scale copy of the NEXT pass from the census ``amax`` (int32 [C / 32]: bit patterns
        of the largest stored magnitudes, dg_epilogue.out_amax) -- what block_exp_max derives from MXFP8 scale bytes; clears the census."""
        assert amax.dtype == torch.int32 and out.dtype == torch.uint8 and amax.numel() == out.numel() and amax.is_contiguous() and out.is_contiguous()
        check(self.lib.dg_exp_from_amax(_ptr(amax), amax.numel(), int(margin), _ptr(out), self._stream()), "dg_exp_from_amax")

    def quant_uniform(self, src, q, exps):
        """q = the uniform-scale E4M3 form of src [..., C] (one exponent per 32-channel block, ``exps`` uint8 [C / 32]): dg_quant_uniform."""
        Cc = src.shape[-1]
        assert q.dtype == torch.uint8 and q.shape == src.shape and exps.dtype == torch.uint8 and exps.numel() == Cc // 32
        if src.dim() == 4:
            ld, rows = pix_layout(src)
            ldq = pix_layout(q)[0]
        else:
            assert src.dim() == 2
            ld, rows, ldq = src.stride(0), src.shape[0], q.stride(0)
        sdt = _lib.DG_F32 if src.dtype == torch.float32 else _lib.DG_BF16
        check(self.lib.dg_quant_uniform(sdt, _ptr(src), rows, ld, Cc, _ptr(exps), _ptr(q), ldq, self._stream()), "dg_quant_uniform")

    def _f8_buf(self, key, shape):
        n = 1
        for d in shape:
            n *= d
        b = self._f8_scratch.get(key)
        if b is None or b.numel() < n:
            b = self._f8_scratch[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        return b[:n].view(shape)

    def _conv_f8(self, fn, g, e, src, w, nout, cred, dst, xq=None, wq=None):
        """the fp8 conv (forward or data gradient); operands that come without an MXFP8 form are quantised into scratch."""
        if xq is None:
            xq = self.quant_mxfp8(src, self._f8_buf("xq", tuple(src.shape)), self._f8_buf("xs", tuple(src.shape[:-1]) + (cred // 32,)))
        if wq is None:
            wq = self.quant_mxfp8(w.view(nout * 9, cred), self._f8_buf("wq", (nout * 9, cred)), self._f8_buf("ws", (nout * 9, cred // 32)))
        (xqq, xs), (wqq, ws) = xq, wq
        assert xqq.shape == src.shape and xqq.stride(-1) == 1, (xqq.shape, src.shape)
        if xs.dim() == 1:           # a UNIFORM-scale source (dg_epilogue.out_u): one row of block exponents for every pixel
            assert xs.numel() == cred // 32 and xs.is_contiguous() and xs.dtype == torch.uint8
            ldxs = -1
        else:
            assert tuple(xs.shape) == tuple(src.shape[:-1]) + (cred // 32,), (xs.shape, src.shape)
            assert all(xs.stride(i) * xqq.stride(2) == xqq.stride(i) * xs.stride(2) for i in range(3))
            ldxs = xs.stride(2)
        assert pix_layout(xqq)[1] == pix_layout(src)[1]
        assert wqq.numel() == nout * 9 * cred and wqq.is_contiguous() and ws.numel() == nout * 9 * (cred // 32) and ws.is_contiguous()
        q = _lib.F8Operands(xq=xqq.data_ptr(), xs=xs.data_ptr(), ldxq=xqq.stride(2), ldxs=ldxs, wq=wqq.data_ptr(), ws=ws.data_ptr())
        return fn(C.byref(g), C.byref(e), C.byref(q), _ptr(dst), self._stream())

    def conv_wgrad(self, cv: Conv, x, dy, dw, db=None):
        """dw += weight gradient; db (optional, fp32 [Cout]) += bias gradient = column sums of dy."""
        self._act(x); self._act(dy)
        assert dw.dtype == torch.float32 and dw.numel() == cv.Cout * 9 * cv.Cin and dw.is_contiguous()
        assert db is None or (db.dtype == torch.float32 and db.numel() >= cv.Cout and not cv.pixel_shuffle)
        # layers with <= 2 real input channels (im2col kernel) also take the COMPACT form [N, H, W, 2] of their input
        assert tuple(x.shape) == (cv.N, cv.H, cv.W, cv.Cin) or (cv.cin_real and tuple(x.shape) == (cv.N, cv.H, cv.W, 2) and cv.stride == 1), (x.shape, cv)
        assert tuple(dy.shape) == self.out_shape(cv), (dy.shape, cv)
        g = self._geom(cv, pix_layout(x)[0], pix_layout(dy)[0])
        check(self._timed("conv_wgrad", self.conv_flops(cv), lambda: self.lib.dg_conv3x3_wgrad(
            C.byref(g), _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), self._stream()), self.conv_bytes(cv), cv.net, self._geom_tag(cv)), "dg_conv3x3_wgrad")

    def conv_wgrad_f8(self, cv: Conv, xq, ex, dyq, ey, dw):
        """dw (flat fp32, accumulated into) += weight gradient from E4M3 operands with per-32-channel-block exponents (uint8 tensors
        ``ex`` [Cin / 32], ``ey`` [Cout / 32]; x = xq * 2^(ex - 127)): dg_conv3x3_wgrad_f8.  Raises on shapes the kernel does not
        take (channel counts that are not multiples of 128, output rows that are not multiples of 64 pixels)."""
        assert xq.dtype == torch.uint8 and dyq.dtype == torch.uint8 and ex.dtype == torch.uint8 and ey.dtype == torch.uint8
        assert dw.dtype == torch.float32 and dw.numel() == cv.Cout * 9 * cv.Cin and ex.numel() == cv.Cin // 32 and ey.numel() == cv.Cout // 32
        g = self._geom(cv, pix_layout(xq)[0], pix_layout(dyq)[0])
        check(self._timed("conv_wgrad_f8", self.conv_flops(cv), lambda: self.lib.dg_conv3x3_wgrad_f8(
            C.byref(g), _ptr(xq), _ptr(ex), _ptr(dyq), _ptr(ey), _ptr(dw), self._stream()),
            float(xq.numel() + dyq.numel()), cv.net, self._geom_tag(cv)), "dg_conv3x3_wgrad_f8")

    def conv_wgrad_dense(self, cvs, slab, us, dws, dbs):
        """Weight / bias gradients of all convs of a dense block: conv k (``cvs[k-1]``: k*F -> F channels) reads ``slab[..., :k*F]``,
        its adjoint is ``us[..., (k-1)*F:k*F]``; ``dws`` / ``dbs`` = their flat fp32 gradients.  One launch when the wide kernel
        takes the shape (bf16, F = 128, rows of a multiple of 32 pixels), else one launch per conv."""
        n, F_ = len(cvs), cvs[0].Cout
        cv0 = cvs[0]
        if not (self.dtype == "bf16" and F_ == 128 and cv0.W % 32 == 0 and all(c.Cin == (k + 1) * F_ and c.Cout == F_ and c.stride == 1 and not c.pixel_shuffle
                                                                              for k, c in enumerate(cvs))) or os.environ.get("DG_WG_NODENSE"):
            for k, c in enumerate(cvs):
                self.conv_wgrad(c, slab[..., :(k + 1) * F_], us[..., k * F_:(k + 1) * F_], dws[k], db=dbs[k] if dbs is not None else None)
            return
        self._act(slab); self._act(us)
        assert tuple(slab.shape) == (cv0.N, cv0.H, cv0.W, n * F_) and tuple(us.shape) == tuple(slab.shape)
        for k in range(n):
            assert dws[k].dtype == torch.float32 and dws[k].numel() == F_ * 9 * (k + 1) * F_ and dws[k].is_contiguous()
            assert dbs is None or (dbs[k].dtype == torch.float32 and dbs[k].numel() >= F_)
        cvv = Conv(cv0.N, cv0.H, cv0.W, n * F_, n * F_, net=cv0.net)
        g = self._geom(cvv, pix_layout(slab)[0], pix_layout(us)[0])
        pw = (C.c_void_p * n)(*[t.data_ptr() for t in dws])
        pb = (C.c_void_p * n)(*[t.data_ptr() for t in dbs]) if dbs is not None else None
        flops = sum(self.conv_flops(c) for c in cvs)
        nbytes = sum(self.conv_bytes(c) for c in cvs)
        check(self._timed("conv_wgrad", flops, lambda: self.lib.dg_conv3x3_wgrad_dense(
            C.byref(g), n, _ptr(slab), _ptr(us), pw, pb, self._stream()), nbytes, cv0.net, f"dense{n}x{F_}@{cv0.H}"), "dg_conv3x3_wgrad_dense")

    def conv_wgrad_dense_f8(self, cvs, slab_u, ex, us_u, eu, dws):
        """conv_wgrad_dense on the fp8 kernel (dg_conv3x3_wgrad_dense_f8): ``slab_u`` / ``us_u`` = uniform-scale E4M3 copies of the
        block's activation / adjoint slabs [N, H, W, n*F] with block exponents ``ex`` / ``eu`` (uint8 [n*F / 32]); weight gradients
        only (bias gradients: ``colsum`` of the adjoint slices)."""
        n, F_ = len(cvs), cvs[0].Cout
        cv0 = cvs[0]
        assert F_ == 128 and cv0.W % 64 == 0 and all(c.Cin == (k + 1) * F_ and c.Cout == F_ and c.stride == 1 and not c.pixel_shuffle for k, c in enumerate(cvs))
        assert slab_u.dtype == torch.uint8 and us_u.dtype == torch.uint8 and tuple(slab_u.shape) == (cv0.N, cv0.H, cv0.W, n * F_) == tuple(us_u.shape)
        assert ex.dtype == torch.uint8 and eu.dtype == torch.uint8 and ex.numel() == n * F_ // 32 == eu.numel() and ex.is_contiguous() and eu.is_contiguous()
        for k in range(n):
            assert dws[k].dtype == torch.float32 and dws[k].numel() == F_ * 9 * (k + 1) * F_ and dws[k].is_contiguous()
        cvv = Conv(cv0.N, cv0.H, cv0.W, n * F_, n * F_, net=cv0.net)
        g = self._geom(cvv, pix_layout(slab_u)[0], pix_layout(us_u)[0])
        pw = (C.c_void_p * n)(*[t.data_ptr() for t in dws])
        flops = sum(self.conv_flops(c) for c in cvs)
        check(self._timed("conv_wgrad_f8", flops, lambda: self.lib.dg_conv3x3_wgrad_dense_f8(
            C.byref(g), n, _ptr(slab_u), _ptr(ex), _ptr(us_u), _ptr(eu), pw, self._stream()),
            float(slab_u.numel() + us_u.numel()), cv0.net, f"dense{n}x{F_}@{cv0.H}"), "dg_conv3x3_wgrad_dense_f8")

    def colsum(self, dy, db):
        """db[c] += sum over all pixels/rows of dy[..., c]; dy is NHWC (any dtype of {fp32, compute}) or 2-D."""
        assert db.dtype == torch.float32
        dg = _lib.DG_F32 if dy.dtype == torch.float32 else self.dg
        if dy.dim() == 2:
            rows, ld, Cc = dy.shape[0], dy.stride(0), dy.shape[1]
        else:
            ld, rows = pix_layout(dy)
            Cc = dy.shape[-1]
        assert db.numel() >= Cc
        check(self.lib.dg_colsum(dg, _ptr(dy), rows, ld, 1, ld, Cc, _ptr(db), self._stream()), "dg_colsum")

    def colsum_multi(self, dy, dbs):
        """dbs[k][c] += sum over pixels of dy[..., k * (C / n) + c] for the n equal channel segments of dy, one pass (dg_colsum_multi)."""
        self._act(dy)
        n, Cc = len(dbs), dy.shape[-1]
        assert Cc % n == 0 and all(d.dtype == torch.float32 and d.numel() >= Cc // n for d in dbs)
        ld, rows = pix_layout(dy)
        pb = (C.c_void_p * n)(*[t.data_ptr() for t in dbs])
        check(self._timed("ew_colsum", 0.0, lambda: self.lib.dg_colsum_multi(self.dg, _ptr(dy), rows, ld, Cc, n, pb, self._stream()),
                          float(dy.numel() * dy.element_size()), "G"), "dg_colsum_multi")

    def colsum_ps(self, dy, db):
        """Bias gradient of a pixel-shuffle conv: dy is stored shuffled [N,2H,2W,F]; db has 4F entries
        ordered (2i+j)*F + c (the native packing of the conv's output channels)."""
        self._act(dy)
        ld, _ = pix_layout(dy)
        n, h2, w2, f = dy.shape
        for i in range(2):
            for j in range(2):
                sub = dy[:, i::2, j::2, :]
                check(self.lib.dg_colsum(self.dg, _ptr(sub), n * (h2 // 2), 2 * w2 * ld, w2 // 2, 2 * ld, f,
                                         _ptr(db[(2 * i + j) * f:]), self._stream()), "dg_colsum")

    def repack(self, master, dst, cout, cin, kind):
        assert master.dtype == torch.float32 and master.numel() == cout * 9 * cin
        assert dst.dtype == self.tdtype and dst.numel() == cout * 9 * cin
        check(self.lib.dg_repack_conv_weights(self.dg, kind, _ptr(master), _ptr(dst), cout, cin, self._stream()), "dg_repack_conv_weights")

    def repack_dense(self, masters, dst, F_):
        """masters[k-1]: fp32 weight [F, 9, k*F] of conv k of a dense block -> dst: the stacked data-gradient packs of its slab
        slices (dg_repack_dense_dgrad), concatenated."""
        n = len(masters)
        for k, m in enumerate(masters):
            assert m.dtype == torch.float32 and m.is_contiguous() and m.numel() == F_ * 9 * (k + 1) * F_
        assert dst.dtype == self.tdtype and dst.is_contiguous() and dst.numel() == 9 * F_ * F_ * n * (n + 1) // 2
        pm = (C.c_void_p * n)(*[m.data_ptr() for m in masters])
        check(self.lib.dg_repack_dense_dgrad(self.dg, pm, n, F_, _ptr(dst), self._stream()), "dg_repack_dense_dgrad")

    def wgrad_unswap(self, tmp, dw, cout, cin):
        """dw[co][t][ci] += tmp[ci][8 - t][co] (both fp32 flat): see dg_wgrad_unswap."""
        assert tmp.dtype == torch.float32 and dw.dtype == torch.float32 and tmp.numel() == dw.numel() == cout * 9 * cin
        check(self.lib.dg_wgrad_unswap(_ptr(tmp), _ptr(dw), cout, cin, self._stream()), "dg_wgrad_unswap")

    # ------------------------------------------------------------------ linear family
    def linear_fwd(self, x, w, y, o_real=0, net=""):
        """y[B,ldy] (fp32, pre-zeroed) += x[B,K] @ w[O,K]^T"""
        self._act(x); self._act(w)
        B, K = x.shape
        O = w.shape[0]
        assert w.shape[1] == K and y.dtype == torch.float32 and y.shape[0] == B and y.shape[1] >= O
        es = x.element_size()
        check(self._timed("lin_fwd", 2.0 * B * (o_real or O) * K, lambda: self.lib.dg_linear_fwd(
            self.dg, _ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(y), y.stride(0), B, O, K, self._stream()),
            float(es * (B * K + O * K)), net), "dg_linear_fwd")

    def linear_dx(self, dy, w, dx, mask=None, mask_slope=1.0, o_real=0, net=""):
        """dx[B,K] = (dy[B,O] @ w[O,K]) * leaky'(mask)"""
        self._act(w)
        B, K = dx.shape
        O = w.shape[0]
        assert dy.dtype == torch.float32 and dy.shape[0] == B and dy.shape[1] >= O and w.shape[1] == K
        out_dg = _lib.DG_F32 if dx.dtype == torch.float32 else self.dg
        if mask is not None:
            self._act(mask)
            assert mask.shape == dx.shape
        es = w.element_size()
        check(self._timed("lin_dx", 2.0 * B * (o_real or O) * K, lambda: self.lib.dg_linear_dx(
            self.dg, out_dg, _ptr(dy), dy.stride(0), _ptr(w), w.stride(0), _ptr(dx), dx.stride(0),
            _ptr(mask), mask.stride(0) if mask is not None else 0, float(mask_slope), B, O, K, self._stream()),
            float(es * O * K + dx.element_size() * B * K * (2 if mask is not None else 1)), net), "dg_linear_dx")

    def linear_dw(self, dy, x, dw, o_real=0, net=""):
        """dw[O,K] (fp32) += dy[B,O]^T @ x[B,K]"""
        self._act(x)
        B, K = x.shape
        O = dw.shape[0]
        assert dy.dtype == torch.float32 and dy.shape[0] == B and dy.shape[1] >= O and dw.dtype == torch.float32 and dw.shape[1] == K
        check(self._timed("lin_dw", 2.0 * B * (o_real or O) * K, lambda: self.lib.dg_linear_dw(
            self.dg, _ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dw), dw.stride(0), B, O, K, self._stream()),
            float(x.element_size() * B * K + 8.0 * O * K), net), "dg_linear_dw")

    def linear_dw_wide(self, dy, x, dw, accumulate=True, o_real=0, net=""):
        """dw[O,K] (fp32) (+)= dy[B,O]^T @ x[B,K] for B <= 128 concatenated rows, O <= 112: one sweep over dw."""
        self._act(x)
        B, K = x.shape
        O = dw.shape[0]
        assert dy.dtype == torch.float32 and dy.shape[0] == B and dy.shape[1] >= O and dw.dtype == torch.float32 and dw.shape[1] == K
        assert B <= 128 and O <= 112 and x.stride(1) == 1 and dw.stride(1) == 1 and dy.stride(1) == 1
        check(self._timed("lin_dw", 2.0 * B * (o_real or O) * K, lambda: self.lib.dg_linear_dw_wide(
            self.dg, _ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dw), dw.stride(0), B, O, K, int(bool(accumulate)),
            self._stream()), float(x.element_size() * B * K + (8.0 if accumulate else 4.0) * O * K), net), "dg_linear_dw_wide")

    def bias_act(self, inp, bias, out, act=None, mask=None, mask_slope=1.0):
        rows, Cc = out.shape
        assert inp.dtype == torch.float32 and inp.shape[0] == rows and inp.shape[1] >= Cc
        out_dg = _lib.DG_F32 if out.dtype == torch.float32 else self.dg
        if mask is not None:
            assert mask.dtype == out.dtype
        check(self._timed("ew_bias_act", 0.0, lambda: self.lib.dg_bias_act(
            out_dg, _ptr(inp), inp.stride(0), _ptr(bias), _ptr(out), out.stride(0), rows, Cc, int(act is not None), float(act or 1.0),
            _ptr(mask), mask.stride(0) if mask is not None else 0, float(mask_slope), self._stream()), 0.0, "C"), "dg_bias_act")

    # ------------------------------------------------------------------ elementwise / reductions
    def mask_mul(self, u, y, slope):
        self._act(u); self._act(y)
        assert u.shape == y.shape
        ldu, rows = pix_layout(u)
        check(self.lib.dg_mask_mul(self.dg, _ptr(u), ldu, _ptr(y), pix_layout(y)[0], rows, u.shape[-1], float(slope), self._stream()), "dg_mask_mul")

    def axpby(self, out, x, a=1.0, y=None, b=0.0):
        self._act(out); self._act(x)
        assert out.shape == x.shape and (y is None or y.shape == x.shape)
        ldo, rows = pix_layout(out)
        check(self.lib.dg_axpby(self.dg, _ptr(out), ldo, _ptr(x), pix_layout(x)[0], float(a), _ptr(y),
                                pix_layout(y)[0] if y is not None else 0, float(b), rows, x.shape[-1], self._stream()), "dg_axpby")

    def gp_interp(self, real, fake, alpha, xhat, real_c=None, fake_c=None):
        """xhat = alpha * real + (1 - alpha) * fake per sample.  With a COMPACT ``xhat`` [B, H, W, 2] (fields stored wider with
        two real channels) the compact forms of the inputs can be written in the same pass (``real_c`` / ``fake_c``)."""
        assert alpha.dtype == torch.float32 and alpha.numel() == real.shape[0] and real.shape == fake.shape
        for t in (real, fake, xhat):
            self._act(t); assert t.is_contiguous()
        if xhat.shape[-1] == 2 and real.shape[-1] > 2:
            assert tuple(xhat.shape[:-1]) == tuple(real.shape[:-1])
            for t in (real_c, fake_c):
                assert t is None or (t.shape == xhat.shape and t.is_contiguous() and t.dtype == xhat.dtype)
            n_out = 1 + (real_c is not None) + (fake_c is not None)
            check(self._timed("ew_gp_interp", 0.0, lambda: self.lib.dg_gp_interp_c2(
                self.dg, _ptr(real), _ptr(fake), real.shape[-1], _ptr(alpha), _ptr(xhat), _ptr(real_c), _ptr(fake_c), real.shape[0],
                real[0].numel() // real.shape[-1], self._stream()),
                (2.0 * real.numel() + n_out * xhat.numel()) * real.element_size(), "C"), "dg_gp_interp_c2")
            return
        assert xhat.shape == real.shape and real_c is None and fake_c is None
        check(self._timed("ew_gp_interp", 0.0, lambda: self.lib.dg_gp_interp(
            self.dg, _ptr(real), _ptr(fake), _ptr(alpha), _ptr(xhat), real.shape[0], real[0].numel(), self._stream()),
            3.0 * real.numel() * real.element_size(), "C"), "dg_gp_interp")

    def sumsq_rows(self, g, ss):
        self._act(g); assert g.is_contiguous() and ss.dtype == torch.float32
        check(self._timed("ew_sumsq_rows", 0.0, lambda: self.lib.dg_sumsq_rows(
            self.dg, _ptr(g), g.shape[0], g[0].numel(), _ptr(ss), self._stream()), 1.0 * g.numel() * g.element_size(), "C"), "dg_sumsq_rows")

    def gp_finish(self, ss, B, B_global, gp_lambda, weight, coef, scalar_out):
        check(self.lib.dg_gp_finish(_ptr(ss), B, B_global, float(gp_lambda), float(weight), _ptr(coef), _ptr(scalar_out), self._stream()), "dg_gp_finish")

    def scale_rows(self, g, coef, out):
        """out[b] = coef[b] * g[b]; a COMPACT ``out`` [B, H, W, 2] takes the two real channels of a wider ``g``."""
        self._act(g); self._act(out); assert g.is_contiguous() and out.is_contiguous()
        if out.shape[-1] == 2 and g.shape[-1] > 2:
            assert tuple(out.shape[:-1]) == tuple(g.shape[:-1])
            check(self._timed("ew_scale_rows", 0.0, lambda: self.lib.dg_scale_rows_c2(
                self.dg, _ptr(g), g.shape[-1], _ptr(coef), _ptr(out), g.shape[0], g[0].numel() // g.shape[-1], self._stream()),
                (g.numel() + out.numel()) * g.element_size(), "C"), "dg_scale_rows_c2")
            return
        assert out.shape == g.shape
        check(self._timed("ew_scale_rows", 0.0, lambda: self.lib.dg_scale_rows(
            self.dg, _ptr(g), _ptr(coef), _ptr(out), g.shape[0], g[0].numel(), self._stream()), 2.0 * g.numel() * g.element_size(), "C"), "dg_scale_rows")

    def l1(self, a, b, acc, grad=None, grad_scale=0.0, addend=None):
        self._act(a); self._act(b)
        assert a.shape == b.shape and acc.dtype == torch.float32
        lda, rows = pix_layout(a)
        check(self.lib.dg_l1(self.dg, _ptr(a), lda, _ptr(b), pix_layout(b)[0], rows, a.shape[-1], _ptr(acc), _ptr(grad),
                             pix_layout(grad)[0] if grad is not None else 0, float(grad_scale), _ptr(addend),
                             pix_layout(addend)[0] if addend is not None else 0, self._stream()), "dg_l1")

    def sqdiff(self, a, b, acc):
        self._act(a); self._act(b)
        assert a.shape == b.shape and acc.dtype == torch.float32
        lda, rows = pix_layout(a)
        check(self.lib.dg_sqdiff(self.dg, _ptr(a), lda, _ptr(b), pix_layout(b)[0], rows, a.shape[-1], _ptr(acc), self._stream()), "dg_sqdiff")

    # ------------------------------------------------------------------ MS-SSIM pieces (metrics pass)
    def minmax(self, x, c_real, partial, minmax):
        """minmax[c] = (min, max) of channel c < c_real over every pixel of the NHWC tensor x."""
        self._act(x)
        assert partial.dtype == torch.float32 and partial.numel() >= _lib.MINMAX_PARTS * c_real * 2 and minmax.numel() >= 2 * c_real
        ld, rows = pix_layout(x)
        check(self.lib.dg_minmax_partial(self.dg, _ptr(x), rows, ld, c_real, _ptr(partial), self._stream()), "dg_minmax_partial")
        check(self.lib.dg_minmax_finish(_ptr(partial), c_real, _ptr(minmax), self._stream()), "dg_minmax_finish")

    def normalise_planar(self, x, c_real, minmax, out):
        """out[n, c, h, w] (fp32) = (x[n, h, w, c] - min_c) / (max_c - min_c)"""
        self._act(x)
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
        assert out.dtype == torch.float32 and tuple(out.shape) == (N, c_real, H, W) and out.is_contiguous()
        check(self.lib.dg_normalise_planar(self.dg, _ptr(x), N, H, W, pix_layout(x)[0], c_real, _ptr(minmax), _ptr(out),
                                           self._stream()), "dg_normalise_planar")

    def ssim_level(self, X, Y, params, sums):
        assert X.dtype == torch.float32 and X.shape == Y.shape and X.is_contiguous() and Y.is_contiguous()
        planes, H, W = X.shape[0] * X.shape[1], X.shape[2], X.shape[3]
        assert sums.dtype == torch.float32 and sums.numel() == 2 * planes
        check(self.lib.dg_ssim_level(_ptr(X), _ptr(Y), planes, H, W, C.byref(params), _ptr(sums), self._stream()), "dg_ssim_level")

    def avgpool2(self, inp, out):
        assert inp.dtype == torch.float32 and out.dtype == torch.float32 and inp.is_contiguous() and out.is_contiguous()
        planes, H, W = inp.shape[0] * inp.shape[1], inp.shape[2], inp.shape[3]
        check(self.lib.dg_avgpool2(_ptr(inp), _ptr(out), planes, H, W, self._stream()), "dg_avgpool2")

    def msssim_finish(self, sums, levels, planes, combine, out):
        assert sums.numel() == levels * planes * 2 and out.dtype == torch.float32
        check(self.lib.dg_msssim_finish(_ptr(sums), levels, planes, C.byref(combine), _ptr(out), self._stream()), "dg_msssim_finish")

    # ------------------------------------------------------------------ frequency separation
    def lowpass5(self, x, low=None, high=None):
        """low = 5x5 box mean with replicated borders, high = x - low (either may be None); NHWC, all padded channels."""
        self._act(x)
        N, H, W, Cc = x.shape
        for t in (low, high):
            assert t is None or (self._act(t) is t and t.shape == x.shape)
        check(self.lib.dg_lowpass5(self.dg, _ptr(x), pix_layout(x)[0], N, H, W, Cc, _ptr(low), pix_layout(low)[0] if low is not None else 0,
                                   _ptr(high), pix_layout(high)[0] if high is not None else 0, self._stream()), "dg_lowpass5")

    def lowpass5_adjoint(self, g, out):
        self._act(g); self._act(out)
        assert g.shape == out.shape and g.data_ptr() != out.data_ptr()
        N, H, W, Cc = g.shape
        check(self.lib.dg_lowpass5_adjoint(self.dg, _ptr(g), pix_layout(g)[0], N, H, W, Cc, _ptr(out), pix_layout(out)[0], self._stream()),
              "dg_lowpass5_adjoint")

    def gather_samples(self, store, idx, dst):
        """dst[b] (NHWC, padded channels) = store[idx[b]] ([n, H, W, c_real], compute dtype)."""
        self._act(dst)
        assert store.dtype == self.tdtype and store.is_contiguous() and store.dim() == 4 and dst.is_contiguous()
        assert idx.dtype == torch.int64 and idx.is_cuda and idx.numel() == dst.shape[0]
        assert tuple(store.shape[1:3]) == tuple(dst.shape[1:3]) and store.shape[3] <= dst.shape[3]
        check(self.lib.dg_gather_samples(self.dg, _ptr(store), store.shape[1] * store.shape[2], store.shape[3], _ptr(idx), idx.numel(),
                                         _ptr(dst), dst.shape[3], self._stream()), "dg_gather_samples")

    def moments(self, x, acc):
        """acc[3] (float64, pre-zeroed) += {sum, sum of squares, count} of the non-NaN elements of the fp32 tensor x."""
        assert x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and acc.dtype == torch.float64 and acc.numel() == 3
        check(self.lib.dg_moments(_ptr(x), x.numel(), _ptr(acc), self._stream()), "dg_moments")

    def stage_fields(self, planes, mean, inv_std, dst):
        """dst[p, k] = (planes[k].flat[p] - mean[k]) * inv_std[k]; planes: c fp32 tensors of equal numel, dst: [..., c] store."""
        c = len(planes)
        assert 1 <= c <= _lib.MAX_FIELDS and dst.shape[-1] == c and dst.is_contiguous() and dst.dtype == self.tdtype
        f = _lib.FieldPlanes(c=c)
        for k, t in enumerate(planes):
            assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous() and t.numel() * c == dst.numel(), (k, t.shape, dst.shape)
            f.plane[k], f.mean[k], f.inv_std[k] = t.data_ptr(), float(mean[k]), float(inv_std[k])
        check(self.lib.dg_stage_fields(self.dg, C.byref(f), planes[0].numel(), _ptr(dst), self._stream()), "dg_stage_fields")

    def div_vort_sums(self, hr, fake, sums):
        """sums[10] (float64, pre-zeroed) += moments of the divergence / vorticity of hr and fake (NHWC, channels 0, 1)."""
        self._act(hr); self._act(fake)
        assert hr.shape == fake.shape and sums.dtype == torch.float64 and sums.numel() == 10
        N, H, W = hr.shape[0], hr.shape[1], hr.shape[2]
        check(self.lib.dg_div_vort_sums(self.dg, _ptr(hr), pix_layout(hr)[0], _ptr(fake), pix_layout(fake)[0], N, H, W, _ptr(sums),
                                        self._stream()), "dg_div_vort_sums")

    def sum_strided(self, inp, n, stride, scale, out):
        assert inp.dtype == torch.float32 and out.dtype == torch.float32
        check(self.lib.dg_sum_strided(_ptr(inp), n, stride, float(scale), _ptr(out), self._stream()), "dg_sum_strided")

    def fill_col(self, buf, col, value):
        assert buf.dtype == torch.float32 and buf.dim() == 2
        check(self.lib.dg_fill_col(_ptr(buf), buf.shape[0], buf.stride(0), col, float(value), self._stream()), "dg_fill_col")

    def adam(self, p, g, m, v, shadow, lr, beta1, beta2, eps, step, grad_scale=1.0):
        for t in (p, g, m, v):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
        if shadow is not None:
            assert shadow.dtype == torch.bfloat16 and shadow.numel() == p.numel()
        check(self.lib.dg_adam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(shadow), p.numel(), lr, beta1, beta2, eps, step,
                               float(grad_scale), self._stream()), "dg_adam")

    def nchw_to_nhwc(self, src, dst):
        n, c, h, w = src.shape
        assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape[:3]) == (n, h, w)
        self._act(dst)
        check(self.lib.dg_nchw_to_nhwc(self.dg, _ptr(src), _ptr(dst), n, c, h, w, dst.shape[3], self._stream()), "dg_nchw_to_nhwc")

    def nhwc_to_nchw(self, src, dst):
        n, c, h, w = dst.shape
        self._act(src)
        assert dst.dtype == torch.float32 and dst.is_contiguous()
        check(self.lib.dg_nhwc_to_nchw(self.dg, _ptr(src), pix_layout(src)[0], _ptr(dst), n, c, h, w, self._stream()), "dg_nhwc_to_nchw")

    def cast(self, src, dst):
        assert src.dtype == torch.float32 and dst.dtype == self.tdtype and src.numel() == dst.numel()
        check(self.lib.dg_cast(self.dg, _ptr(src), _ptr(dst), src.numel(), self._stream()), "dg_cast")

    # ------------------------------------------------------------------ plumbing (device memory only)
    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.tdtype, device=self.device)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.tdtype, device=self.device)
