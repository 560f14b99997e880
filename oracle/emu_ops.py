"""TEST INFRASTRUCTURE — torch-CPU emulation of the op contracts of ``downgan_amd.ops.HipOps``.

Purpose: run the host-side engine (buffer plumbing, slab offsets, hand-derived backward and
double-backward chains) on a machine without a GPU and compare it with ``oracle/ref_step.py``.
The conv ops go through the REAL host planner of libdowngan_hip.so (``dg_conv3x3_plan``, which needs
no GPU) and then evaluate the planned gather-GEMM descriptor with plain torch indexing, so the
planner's tap/parity-class logic is covered by the CPU test-suite too.

Never imported by the product package.
"""
from __future__ import annotations

import ctypes as C

import torch

from downgan_amd import _lib
from downgan_amd.ops import Conv, pix_layout

TORCH_DTYPE = {"f32": torch.float32, "bf16": torch.bfloat16}
DG_DTYPE = {"f32": _lib.DG_F32, "bf16": _lib.DG_BF16}


def _lgrad(y, slope):
    return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))


class EmuOps:
    name = "emu"

    def __init__(self, dtype="f32", device="cpu", f8_critic=False, f8_generator=False):
        self.f8 = bool(f8_critic or f8_generator)
        self.f8_generator = bool(f8_generator)
        self.dtype = dtype
        self.tdtype = TORCH_DTYPE[dtype]
        self.dg = DG_DTYPE[dtype]
        self.device = torch.device(device)
        self.lib = _lib.lib()

    deterministic = True      # torch-CPU reductions: nothing to switch

    def close(self):
        pass

    def count_nonfinite(self, bufs):
        return {name: int((~torch.isfinite(t.float())).sum()) for name, t in bufs}

    # ------------------------------------------------------------------ conv family
    def out_shape(self, cv: Conv):
        if cv.pixel_shuffle:
            return (cv.N, 2 * cv.Ho, 2 * cv.Wo, cv.Cout // 4)
        return (cv.N, cv.Ho, cv.Wo, cv.Cout)

    def _plan(self, cv: Conv, kind, ldx, ldy):
        g = _lib.ConvGeom(dtype=self.dg, N=cv.N, H=cv.H, W=cv.W, Cin=cv.Cin, Cout=cv.Cout, stride=cv.stride,
                          cin_real=cv.cin_real, pixel_shuffle=int(cv.pixel_shuffle), ldx=ldx, ldy=ldy)
        d = (_lib.GGDesc * 4)()
        n = self.lib.dg_conv3x3_plan(C.byref(g), kind, d)
        assert n > 0, n
        return [d[i] for i in range(n)]

    @staticmethod
    def bits_shape(shape):
        N, H, W, Cc = shape
        assert Cc % 64 == 0 and Cc >= 128, Cc
        return (N, H, W, Cc // 64, 4)

    @staticmethod
    def _bit_index(Cc):
        """channel c -> (block, word, bit): c = 64*block + 16*word + bit, i.e. a plain little-endian bit string over the
        channels of a pixel (include/downgan_hip.h, dg_epilogue)."""
        c = torch.arange(Cc)
        return c // 64, (c % 64) // 16, c % 16

    def _pack_bits(self, pos, bits):
        blk, g, b = self._bit_index(pos.shape[-1])
        words = torch.zeros(*bits.shape, dtype=torch.int32)
        for c in range(pos.shape[-1]):
            words[..., blk[c], g[c]] |= (pos[..., c].to(torch.int32) << int(b[c]))
        bits.copy_(words.to(torch.int16))          # wraps bit 15 into the sign

    def _unpack_bits(self, bits, Cc):
        blk, g, b = self._bit_index(Cc)
        w = bits.to(torch.int32) & 0xffff
        return torch.stack([(w[..., blk[c], g[c]] >> int(b[c])) & 1 for c in range(Cc)], dim=-1).bool()

    def _gather_gemm(self, d, x, w, y, bias=None, act=None, r1=None, s1=1.0, r2=None, s2=1.0, mask=None,
                     mask_slope=1.0, accumulate=False, mask_bits=None, out_bits=None, mask_c0=0, mask_last=False, out_q=None, out_u=None, skip_y=False,
                     out_amax=None):
        if skip_y:      # dg_epilogue.skip_y: the fp8 copies / mask bits are formed from the values the launch WOULD store; y keeps its content
            assert (out_q is not None or out_u is not None) and not accumulate
            y = y.clone()
        N = d.N
        xs = x.float()
        if d.src_ps:
            cps = d.Cred // 4
            assert tuple(xs.shape) == (N, 2 * d.Hs, 2 * d.Ws, cps)
            xs = xs.view(N, d.Hs, 2, d.Ws, 2, cps).permute(0, 1, 3, 2, 4, 5).reshape(N, d.Hs, d.Ws, 4 * cps)
        assert tuple(xs.shape) == (N, d.Hs, d.Ws, d.Cred), (xs.shape, d.Hs, d.Ws, d.Cred)
        Wp = w.float().view(d.Nout, 9, d.Cred)
        assert d.ldw == 9 * d.Cred
        gy = torch.arange(d.Hg)
        gx = torch.arange(d.Wg)
        acc = torch.zeros(N, d.Hg, d.Wg, d.Nout)
        for t in range(d.ntaps):
            sy = gy * d.sy_mul + d.tap_dy[t]
            sx = gx * d.sx_mul + d.tap_dx[t]
            vy = (sy >= 0) & (sy < d.Hs)
            vx = (sx >= 0) & (sx < d.Ws)
            patch = xs[:, sy.clamp(0, d.Hs - 1)][:, :, sx.clamp(0, d.Ws - 1)]
            patch = patch * (vy[:, None] & vx[None, :])[None, :, :, None]
            acc += torch.einsum("nhwc,oc->nhwo", patch, Wp[:, d.tap_w[t], :])
        if d.dst_ps:
            cps = d.Nout // 4
            acc = acc.view(N, d.Hg, d.Wg, 2, 2, cps).permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * d.Hg, 2 * d.Wg, cps)
            sl = (slice(None), slice(None), slice(None))
            bias_v = None
            if bias is not None:  # bias is in packed (q*cps + c) order -> broadcast per sub-position
                bias_v = bias[:d.Nout].view(2, 2, cps)[None, None, :, None, :, :].expand(N, d.Hg, 2, d.Wg, 2, cps).reshape(N, 2 * d.Hg, 2 * d.Wg, cps)
        else:
            sl = (slice(None), slice(d.dy_off, None, d.dy_mul), slice(d.dx_off, None, d.dx_mul))
            bias_v = bias[:d.Nout] if bias is not None else None
        ysub = y[sl]
        assert tuple(ysub.shape) == tuple(acc.shape), (ysub.shape, acc.shape)
        v = acc
        if bias_v is not None:
            v = v + bias_v
        if act is not None:
            v = torch.where(v > 0, v, v * act)
        if r1 is not None:
            v = v * s1 + r1[sl].float()
        if r2 is not None:
            v = v * s2 + r2[sl].float()
        def apply_mask(v):       # channels >= mask_c0 only (include/downgan_hip.h, dg_epilogue)
            f = _lgrad(mask[sl].float(), mask_slope)
            f = torch.where(torch.arange(v.shape[-1]) >= mask_c0, f, torch.ones(()))
            return v * f
        if mask is not None and not mask_last:
            v = apply_mask(v)
        if mask_bits is not None:
            assert not d.dst_ps
            pos = self._unpack_bits(mask_bits[sl], y.shape[-1])
            v = v * torch.where(pos[..., :v.shape[-1]], torch.ones(()), torch.full((), float(mask_slope)))
        if accumulate:
            v = v + ysub.float()
        if mask is not None and mask_last:
            v = apply_mask(v)
        ysub.copy_(v.to(y.dtype))
        if out_bits is not None:
            assert not d.dst_ps and d.dy_mul == 1 and d.dx_mul == 1
            full = torch.zeros(*y.shape, dtype=torch.bool)
            full[..., :v.shape[-1]] = y[..., :v.shape[-1]].float() > 0
            self._pack_bits(full, out_bits)
        if out_q is not None:      # MXFP8 copy of the stored (rounded) output; a strided destination is complete after its last class
            q, s, _ = self.mx_quant(y)
            out_q[0].copy_(q)
            out_q[1].copy_(s)
        if out_u is not None:      # uniform-scale E4M3 copy (one exponent per 32-channel block of the whole tensor): dg_epilogue.out_u
            assert out_q is not None or d.Cred <= 16      # (alone: first-layer launches)
            out_u[0].copy_(self.uq_quant(y, out_u[1])[0])
        if out_amax is not None:   # dg_epilogue.out_amax: running maximum of the stored magnitudes' bit patterns per 32-channel block
            Cc = y.shape[-1]
            m = y.float().abs().reshape(-1, Cc // 32, 32).amax(dim=(0, 2)).contiguous().view(torch.int32)
            out_amax.copy_(torch.maximum(out_amax, m))

    # ---- uniform-scale fp8 (csrc/gg_common.h epi64_pixel f_u; csrc/wgrad.hip wg3w_f8_kernel): E4M3 elements, ONE E8M0 exponent
    # per 32-channel block for the whole tensor -- the operand format of the fp8 weight gradient, whose contraction runs over pixels
    @staticmethod
    def uq_quant(x, exps):
        """x [..., C], exps uint8 [C/32] -> (q uint8 [..., C], dequantised fp32 [..., C]); a block holding a NaN / Inf is poisoned
        (32 x 0x7F) like the MXFP8 copy."""
        Cc = x.shape[-1]
        v = x.float().reshape(-1, Cc // 32, 32)
        scale = torch.ldexp(torch.ones(Cc // 32), exps.reshape(-1).int() - 127).view(1, -1, 1)
        q8 = (v / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        q8 = torch.where(torch.isfinite(v.abs().amax(-1))[..., None], q8, torch.full_like(q8, 0x7F))
        deq = q8.view(torch.float8_e4m3fn).float() * scale
        return q8.reshape(x.shape), deq.reshape(x.shape)

    @staticmethod
    def uq_dequant(q, exps):
        Cc = q.shape[-1]
        scale = torch.ldexp(torch.ones(Cc // 32), exps.reshape(-1).int() - 127).view(1, -1, 1)
        return (q.contiguous().view(torch.float8_e4m3fn).float().reshape(-1, Cc // 32, 32) * scale).reshape(q.shape)

    def quant_uniform(self, src, q, exps):
        q.copy_(self.uq_quant(src, exps)[0])

    def block_exp_max_batch(self, pairs, margin=1):
        for scales, out in pairs:
            self.block_exp_max(scales, out, margin)

    def block_exp_max(self, scales, out, margin=1):
        nb = scales.shape[-1]
        v = (scales.reshape(-1, nb).to(torch.int32).amax(0) + int(margin)).clamp(max=254)
        out.copy_(self._exp_decay(v, out).to(torch.uint8))

    @staticmethod
    def _exp_decay(v, old):
        """An exponent falls by at most ONE per update (csrc/quant.hip block_exp_finish_kernel): passes over one buffer alternate between
        inputs of different size, and exponents taken from the smaller one would saturate the larger one's copy."""
        o = old.reshape(-1).to(torch.int32)
        return torch.where((o > 0) & (v + 1 < o), o - 1, v)

    def conv_wgrad_f8(self, cv, xq, ex, dyq, ey, dw):
        assert cv.Cin % 128 == 0 and cv.Cout % 128 == 0 and cv.Wo % 64 == 0 and not cv.pixel_shuffle
        self.conv_wgrad(cv, self.uq_dequant(xq, ex), self.uq_dequant(dyq, ey), dw)

    def conv_wgrad_dense_f8(self, cvs, slab_u, ex, us_u, eu, dws):
        """dg_conv3x3_wgrad_dense_f8: the dense block's five weight gradients from the uniform-scale copies of its slabs."""
        F_ = cvs[0].Cout
        assert cvs[0].W % 64 == 0 and F_ == 128
        x, u = self.uq_dequant(slab_u, ex), self.uq_dequant(us_u, eu)
        for k, c in enumerate(cvs):
            self.conv_wgrad(c, x[..., :(k + 1) * F_], u[..., k * F_:(k + 1) * F_], dws[k])

    # ---- MXFP8 (csrc/quant.hip): OCP E4M3 elements, one E8M0 scale per block of 32 consecutive channels (OCP MX layout);
    # scale = 2^(floor(log2 amax) - 8)
    @staticmethod
    def mx_quant(x):
        """x [..., C] -> (q uint8 [..., C] E4M3 bit patterns, scales uint8 [..., C/32], dequantised fp32 [..., C])."""
        Cc = x.shape[-1]
        assert Cc % 128 == 0
        v = x.float().reshape(-1, Cc // 32, 32)
        amax = v.abs().amax(-1)
        e = (((amax.contiguous().view(torch.int32) >> 23) & 0xff) - 8).clamp(min=0)
        scale = torch.ldexp(torch.ones_like(amax), e - 127)
        q8 = (v / scale[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        # a block that holds a NaN / Inf is written as 32 x NaN (0x7F) -- csrc/dg_internal.h mx_poison
        q8 = torch.where(torch.isfinite(amax)[..., None], q8, torch.full_like(q8, 0x7F))
        deq = q8.view(torch.float8_e4m3fn).float() * scale[..., None]
        return (q8.reshape(x.shape), e.to(torch.uint8).reshape(tuple(x.shape[:-1]) + (Cc // 32,)), deq.reshape(x.shape))

    def quant_mxfp8(self, src, q=None, scales=None):
        qq, ss, _ = self.mx_quant(src)
        if q is None:
            return qq, ss
        q.copy_(qq.view(q.shape))
        scales.copy_(ss.view(scales.shape))
        return q, scales

    f8_wgrad = True
    f8_gbwd = True
    f8_gwgrad = True
    f8_gtail = True
    f8_l0u = True

    def f8_eligible(self, cv, kind):
        if kind == "wgrad":
            return (self.f8 and self.f8_wgrad and cv.net == "C" and not cv.pixel_shuffle and cv.Cin % 128 == 0
                    and cv.Cout % 128 == 0 and cv.Wo % 64 == 0)
        cred, nout = (cv.Cin, cv.Cout) if kind == "fwd" else (cv.Cout, cv.Cin)
        # generator layers (net "G": the dense-block trunk): forward always in f8_generator mode, data gradients when f8_gbwd is on
        nets = ("C", "G") if (self.f8_generator and (kind == "fwd" or self.f8_gbwd)) else ("C",)
        # net "T": the forward of the generator's up-sampling tail (pixel-shuffled outputs included) when f8_gtail is on
        if self.f8_generator and self.f8_gtail and kind == "fwd" and cv.net == "T":
            return cred % 128 == 0 and nout > 64 and (not cv.pixel_shuffle or (cv.Cout // 4) % 32 == 0)
        return self.f8 and cv.net in nets and cred % 128 == 0 and nout > 64 and not cv.pixel_shuffle

    @staticmethod
    def mx_dequant(q, s):
        """(E4M3 bytes [..., C], E8M0 scale bytes [..., C/32]; any strides) -> fp32 values [..., C]."""
        Cc = q.shape[-1]
        v = q.contiguous().view(torch.float8_e4m3fn).float().reshape(-1, Cc // 32, 32)
        sc = torch.ldexp(torch.ones(s.numel()), s.contiguous().reshape(-1).int() - 127).reshape(-1, Cc // 32, 1)
        return (v * sc).reshape(q.shape)

    def _f8_operand(self, t, pre):
        """dequantised fp32 value of an operand: from its producer-written MXFP8 form if given (a 1-D scale tensor = the block
        exponents of a UNIFORM-scale form, valid for every pixel), else quantised here."""
        if pre is not None and pre[1].dim() == 1:
            return self.uq_dequant(pre[0], pre[1]).reshape(t.shape)
        return self.mx_dequant(*pre).reshape(t.shape) if pre is not None else self.mx_quant(t)[2]

    def exp_from_amax(self, amax, out, margin=1):
        e = ((((amax >> 23) & 0xff) - 8).clamp(min=0) + int(margin)).clamp(max=254)
        out.copy_(self._exp_decay(e, out).to(torch.uint8))
        amax.zero_()

    @staticmethod
    def _widen(cv, x):
        """COMPACT [N, H, W, 2] input of a layer with <= 2 real input channels -> the padded form the restatement walks."""
        if cv.cin_real and x.shape[-1] == 2 and cv.Cin > 2:
            return torch.nn.functional.pad(x, (0, cv.Cin - 2))
        return x

    def conv_fwd(self, cv: Conv, x, w_fwd, y, xq=None, wq=None, **ep):
        x = self._widen(cv, x)
        assert tuple(x.shape) == (cv.N, cv.H, cv.W, cv.Cin) and tuple(y.shape) == self.out_shape(cv)
        (d,) = self._plan(cv, 0, pix_layout(x)[0], pix_layout(y)[0])
        if self.f8_eligible(cv, "fwd"):       # fp8 operands, exact products, fp32 accumulation
            x = self._f8_operand(x, xq)
            w_fwd = self._f8_operand(w_fwd.view(cv.Cout * 9, cv.Cin), wq)
        self._gather_gemm(d, x, w_fwd, y, **ep)

    def conv_dgrad(self, cv: Conv, dy, w_dgrad, dx, xq=None, wq=None, **ep):
        assert tuple(dx.shape) == (cv.N, cv.H, cv.W, cv.Cin) and tuple(dy.shape) == self.out_shape(cv)
        if self.f8_eligible(cv, "dgrad"):
            dy = self._f8_operand(dy, xq)
            w_dgrad = self._f8_operand(w_dgrad.view(cv.Cin * 9, cv.Cout), wq)
        if ep.get("skip_y"):      # the parity classes of a stride-2 layer complete ONE would-be tensor: its copies come from all of them
            dx, ep = dx.clone(), dict(ep, skip_y=False)
        for d in self._plan(cv, 1, pix_layout(dx)[0], pix_layout(dy)[0]):
            self._gather_gemm(d, dy, w_dgrad, dx, **ep)

    def conv_wgrad(self, cv: Conv, x, dy, dw, db=None):
        x = self._widen(cv, x)
        if db is not None:
            self.colsum(dy, db)
        u = dy.float()
        if cv.pixel_shuffle:
            cps = cv.Cout // 4
            u = u.view(cv.N, cv.Ho, 2, cv.Wo, 2, cps).permute(0, 1, 3, 2, 4, 5).reshape(cv.N, cv.Ho, cv.Wo, cv.Cout)
        xp = torch.nn.functional.pad(x.float(), (0, 0, 1, 1, 1, 1))
        g = dw.view(cv.Cout, 9, cv.Cin)
        st = cv.stride
        for r in range(3):
            for s in range(3):
                patch = xp[:, r:r + st * cv.Ho:st, s:s + st * cv.Wo:st, :]
                g[:, r * 3 + s, :] += torch.einsum("nhwo,nhwc->oc", u, patch)

    def conv_wgrad_dense(self, cvs, slab, us, dws, dbs):
        F_ = cvs[0].Cout
        for k, c in enumerate(cvs):
            self.conv_wgrad(c, slab[..., :(k + 1) * F_], us[..., k * F_:(k + 1) * F_], dws[k], db=dbs[k])

    def colsum(self, dy, db):
        Cc = dy.shape[-1]
        db[:Cc] += dy.float().reshape(-1, Cc).sum(0)

    def colsum_multi(self, dy, dbs):
        seg = dy.shape[-1] // len(dbs)
        for k, db in enumerate(dbs):
            self.colsum(dy[..., k * seg:(k + 1) * seg], db)

    def colsum_ps(self, dy, db):
        f = dy.shape[-1]
        for i in range(2):
            for j in range(2):
                db[(2 * i + j) * f:(2 * i + j + 1) * f] += dy[:, i::2, j::2, :].float().reshape(-1, f).sum(0)

    def repack(self, master, dst, cout, cin, kind):
        m = master.view(cout, 9, cin)
        if kind == 0:
            dst.copy_(m.reshape(-1).to(dst.dtype))
        elif kind == 1:
            dst.copy_(m.permute(2, 1, 0).reshape(-1).to(dst.dtype))
        else:                                    # kind 2: data-gradient pack with mirrored taps
            dst.copy_(m.permute(2, 1, 0).flip(1).reshape(-1).to(dst.dtype))

    def repack_dense(self, masters, dst, F_):
        n, off = len(masters), 0
        for j in range(n):
            wv = torch.cat([masters[k - 1].view(F_, 9, k * F_)[:, :, j * F_:(j + 1) * F_] for k in range(j + 1, n + 1)], 0).contiguous()
            size = wv.numel()
            self.repack(wv.reshape(-1), dst[off:off + size], (n - j) * F_, F_, 1)
            off += size

    def wgrad_unswap(self, tmp, dw, cout, cin):
        dw.view(cout, 9, cin).add_(tmp.view(cin, 9, cout).flip(1).permute(2, 1, 0))

    # ------------------------------------------------------------------ linear family
    def linear_fwd(self, x, w, y, o_real=0, net=""):
        O = w.shape[0]
        y[:, :O] += x.float() @ w.float().t()

    def linear_dx(self, dy, w, dx, mask=None, mask_slope=1.0, o_real=0, net=""):
        O = w.shape[0]
        v = dy[:, :O] @ w.float()
        if mask is not None:
            v = v * _lgrad(mask.float(), mask_slope)
        dx.copy_(v.to(dx.dtype))

    def linear_dw(self, dy, x, dw, o_real=0, net=""):
        O = dw.shape[0]
        dw += dy[:, :O].t() @ x.float()

    def linear_dw_wide(self, dy, x, dw, accumulate=True, o_real=0, net=""):
        O = dw.shape[0]
        r = dy[:, :O].t() @ x.float()
        if accumulate:
            dw += r
        else:
            dw.copy_(r)

    def bias_act(self, inp, bias, out, act=None, mask=None, mask_slope=1.0):
        Cc = out.shape[1]
        v = inp[:, :Cc].clone()
        if bias is not None:
            v = v + bias[:Cc]
        if act is not None:
            v = torch.where(v > 0, v, v * act)
        if mask is not None:
            v = v * _lgrad(mask.float(), mask_slope)
        out.copy_(v.to(out.dtype))

    # ------------------------------------------------------------------ elementwise
    def mask_mul(self, u, y, slope):
        u.copy_((u.float() * _lgrad(y.float(), slope)).to(u.dtype))

    def axpby(self, out, x, a=1.0, y=None, b=0.0):
        v = a * x.float()
        if y is not None:
            v = v + b * y.float()
        out.copy_(v.to(out.dtype))

    def gp_interp(self, real, fake, alpha, xhat, real_c=None, fake_c=None):
        a = alpha.view(-1, 1, 1, 1)
        c = xhat.shape[-1]                  # compact outputs take the leading channels
        xhat.copy_((a * real[..., :c].float() + (1 - a) * fake[..., :c].float()).to(xhat.dtype))
        if real_c is not None:
            real_c.copy_(real[..., :c])
        if fake_c is not None:
            fake_c.copy_(fake[..., :c])

    def sumsq_rows(self, g, ss):
        ss += (g.float() ** 2).reshape(g.shape[0], -1).sum(1)

    def gp_finish(self, ss, B, B_global, gp_lambda, weight, coef, scalar_out):
        n = torch.sqrt(ss[:B] + 1e-12)
        d = n - 1
        coef[:B] = weight * gp_lambda * (2.0 / B_global) * d / n
        scalar_out[0] = gp_lambda * (d * d).sum() / B_global

    def scale_rows(self, g, coef, out):
        out.copy_((g[..., :out.shape[-1]].float() * coef[:g.shape[0]].view(-1, 1, 1, 1)).to(out.dtype))

    def l1(self, a, b, acc, grad=None, grad_scale=0.0, addend=None):
        d = a.float() - b.float()
        acc[0] += d.abs().sum()
        if grad is not None:
            g = torch.sign(d) * grad_scale
            if addend is not None:
                g = g + addend.float()
            grad.copy_(g.to(grad.dtype))

    def sqdiff(self, a, b, acc):
        acc[0] += ((a.float() - b.float()) ** 2).sum()

    # MS-SSIM pieces: plain torch restatements of the kernel contracts of csrc/metrics.hip
    def minmax(self, x, c_real, partial, minmax):
        v = x[..., :c_real].float().reshape(-1, c_real)
        mm = minmax.view(c_real, 2)
        mm[:, 0] = v.amin(0)
        mm[:, 1] = v.amax(0)

    def normalise_planar(self, x, c_real, minmax, out):
        mm = minmax.view(c_real, 2)
        v = x[..., :c_real].float().permute(0, 3, 1, 2)
        out.copy_((v - mm[:, 0].view(1, -1, 1, 1)) / (mm[:, 1] - mm[:, 0]).view(1, -1, 1, 1))

    def ssim_level(self, X, Y, params, sums):
        win = torch.tensor([params.g[i] for i in range(params.win)], dtype=torch.float32)

        def gf(t):
            Cn = t.shape[1]
            w = win.view(1, 1, -1, 1).repeat(Cn, 1, 1, 1)
            t = torch.nn.functional.conv2d(t, w, groups=Cn)
            return torch.nn.functional.conv2d(t, w.transpose(2, 3), groups=Cn)
        mu1, mu2 = gf(X), gf(Y)
        s1, s2, s12 = gf(X * X) - mu1 * mu1, gf(Y * Y) - mu2 * mu2, gf(X * Y) - mu1 * mu2
        cs = (2 * s12 + params.C2) / (s1 + s2 + params.C2)
        ss = ((2 * mu1 * mu2 + params.C1) / (mu1 * mu1 + mu2 * mu2 + params.C1)) * cs
        sm = sums.view(-1, 2)
        sm[:, 0] += ss.flatten(2).sum(-1).flatten()
        sm[:, 1] += cs.flatten(2).sum(-1).flatten()

    def avgpool2(self, inp, out):
        out.copy_(torch.nn.functional.avg_pool2d(inp, 2, padding=[inp.shape[2] % 2, inp.shape[3] % 2]))

    def msssim_finish(self, sums, levels, planes, combine, out):
        sm = sums.view(levels, planes, 2)
        v = torch.ones(planes)
        for l in range(levels):
            term = torch.relu((sm[l, :, 0] if l == levels - 1 else sm[l, :, 1]) * combine.inv_count[l])
            v = v * term ** combine.weight[l]
        out[0] = v.sum()

    # frequency separation (csrc/freqsep.hip contracts)
    @staticmethod
    def _lp(v):
        return torch.nn.functional.avg_pool2d(torch.nn.functional.pad(v, (2, 2, 2, 2), mode="replicate"), 5, stride=1)

    def lowpass5(self, x, low=None, high=None):
        v = x.float().permute(0, 3, 1, 2)
        lo = self._lp(v)
        if low is not None:
            low.copy_(lo.permute(0, 2, 3, 1).to(low.dtype))
        if high is not None:
            high.copy_((v - lo).permute(0, 2, 3, 1).to(high.dtype))

    def lowpass5_adjoint(self, g, out):
        v = g.float().permute(0, 3, 1, 2).clone().requires_grad_(False)
        probe = torch.zeros_like(v, requires_grad=True)
        (adj,) = torch.autograd.grad(self._lp(probe), probe, grad_outputs=v)
        out.copy_(adj.permute(0, 2, 3, 1).to(out.dtype))

    def sum_strided(self, inp, n, stride, scale, out):
        out[0] = inp.reshape(-1)[:n * stride:stride].sum() * scale

    def fill_col(self, buf, col, value):
        buf[:, col] = value

    def adam(self, p, g, m, v, shadow, lr, beta1, beta2, eps, step, grad_scale=1.0):
        gg = g * grad_scale
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        denom = v.sqrt() / (bc2 ** 0.5) + eps
        p.addcdiv_(m, denom, value=-lr / bc1)
        if shadow is not None:
            shadow.copy_(p.to(shadow.dtype))

    def gather_samples(self, store, idx, dst):
        dst.zero_()
        dst[..., :store.shape[3]] = store[idx.long()].to(dst.dtype)

    def moments(self, x, acc):
        v = x.double().flatten()
        v = v[~torch.isnan(v)]
        acc += torch.stack([v.sum(), (v * v).sum(), torch.tensor(float(v.numel()), dtype=torch.float64)])

    def stage_fields(self, planes, mean, inv_std, dst):
        for k, t in enumerate(planes):
            dst.view(-1, len(planes))[:, k] = ((t.flatten() - torch.tensor(mean[k], dtype=torch.float32))
                                               * torch.tensor(inv_std[k], dtype=torch.float32)).to(dst.dtype)

    def nchw_to_nhwc(self, src, dst):
        dst.zero_()
        dst[..., :src.shape[1]] = src.permute(0, 2, 3, 1).to(dst.dtype)

    def nhwc_to_nchw(self, src, dst):
        dst.copy_(src[..., :dst.shape[1]].float().permute(0, 3, 1, 2))

    def cast(self, src, dst):
        dst.copy_(src.view(dst.shape).to(dst.dtype))

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.tdtype)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.tdtype)
