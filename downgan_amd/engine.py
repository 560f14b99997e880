"""The native train-step engine: generator, critic, gradient penalty and Adam as explicit kernel chains.

No autograd graph is built.  Forward, backward and the gradient-penalty double backward are written
out as sequences of ops of ``downgan_amd.ops`` (HIP kernels through the C ABI); every buffer is
pre-allocated once ("caller owns memory").  The maths follows the reference:

* Generator  — reference DoWnGAN/networks/generator.py:14-90.  Each DenseResidualBlock keeps one
  NHWC slab of 5F channels [x | b1 | b2 | b3 | b4]; conv k reads the first k*F channels and writes
  its output at channel offset k*F, which removes every ``torch.cat`` (generator.py:38-40).  The
  residuals ``out*0.2 + x`` (:41, :53) and the trunk skip (:87) are conv epilogues; PixelShuffle
  (:73) is folded into the store addresses of the up-sampling convs.
* Critic     — reference DoWnGAN/networks/critic.py:20-106; LeakyReLU(0.2) fused into the convs,
  NCHW flatten (:103) absorbed by a one-time column permutation of FC1.
* Critic step / GP / generator step — reference DoWnGAN/GAN/wasserstein.py:27-117, see
  ``TrainEngine``.

LeakyReLU'(z) is recovered from the saved activation (sign(phi(z)) == sign(z)); the critic at >= 128-channel widths keeps
1-bit masks beside its activations instead (16x fewer bytes in the HBM-bound data-gradient / tangent epilogues).

Precision modes (``ops.dtype`` / ``ops.f8``): "f32" exact-fp32 MFMA (parity), "bf16" (throughput; fp32 masters and Adam), and
bf16 with the MXFP8 conv path (BASELINE configs[4]): the critic's wide convs forward + data gradient, optionally the forward
of the generator's dense-block trunk, read fp8 forms that the producing launch's epilogue wrote.
"""
from __future__ import annotations

import dataclasses
import math
import os

import torch

from . import layout
from .ops import Conv

G_SLOPE = 0.01   # nn.LeakyReLU() default, generator.py:26,72,79
C_SLOPE = 0.2    # critic.py:24...
RES_SCALE = 0.2  # generator.py:19,45
FC_HID = 100     # critic.py:96
FC_HID_P = 112   # rows of the packed FC1 weight (7 MFMA fragments)
FC_HID_LD = 128  # row stride of the hidden activations / K of FC2
FC_OUT_P = 16


@dataclasses.dataclass
class HyperParams:
    """reference DoWnGAN/config/hyperparams.py:16-22 and GAN/stage.py:63-64."""
    gp_lambda: float = 10.0
    critic_iterations: int = 5
    batch_size: int = 32
    gamma: float = 0.01
    content_lambda: float = 5.0
    lr: float = 0.00025
    beta1: float = 0.9
    beta2: float = 0.99
    eps: float = 1e-8


class ParamStore:
    """All parameters of one network in ONE flat fp32 buffer (+ grads, Adam moments, compute-precision
    shadow and data-gradient weight packs), so Adam and the RCCL all-reduce are single flat passes."""

    def __init__(self, ops):
        self.ops = ops
        self.entries = {}      # name -> (offset, numel, shape)
        self.size = 0
        self.conv_dgrad = {}   # name -> (cout_p, cin_p)
        self.after_refresh = []   # callbacks run whenever the compute-precision packs were re-derived (fp8 weight forms)

    def add(self, name, shape):
        n = int(math.prod(shape))
        assert n % 4 == 0, (name, shape)
        self.entries[name] = (self.size, n, tuple(shape))
        self.size += n

    def finalize(self):
        o = self.ops
        self.p = o.zeros(self.size, dtype=torch.float32)
        self.g = o.zeros(self.size, dtype=torch.float32)
        self.m = o.zeros(self.size, dtype=torch.float32)
        self.v = o.zeros(self.size, dtype=torch.float32)
        self.shadow = self.p if o.tdtype == torch.float32 else o.zeros(self.size)
        self.dgrad_pack = {k: o.zeros(co * 9 * ci) for k, (co, ci) in self.conv_dgrad.items()}
        self.t = 0

    # A parameter update whose gradient all-reduce is still in flight on RCCL's stream can be parked here; every accessor
    # below completes it first, so whatever needs the parameters (or the gradient buffer) next waits for it, and anything
    # that does not (the generator's forward after a critic update) overlaps with the collective.
    _pending = None

    def defer(self, fn):
        self.sync()
        self._pending = fn

    def sync(self):
        fn, self._pending = self._pending, None
        if fn is not None:
            fn()

    def view(self, buf, name):
        if self._pending is not None:
            self.sync()
        off, n, shape = self.entries[name]
        return buf[off:off + n].view(shape)

    def master(self, name):
        return self.view(self.p, name)

    def grad(self, name):
        return self.view(self.g, name)

    def w(self, name):
        """compute-precision forward pack (flat)"""
        if self._pending is not None:
            self.sync()
        off, n, _ = self.entries[name]
        return self.shadow[off:off + n]

    def w2d(self, name):
        return self.view(self.shadow, name)

    def wd(self, name):
        if self._pending is not None:
            self.sync()
        return self.dgrad_pack[name]

    def refresh(self, shadow_done=False):
        """Re-derive the compute-precision packs from the fp32 master (after load / Adam)."""
        o = self.ops
        if o.tdtype != torch.float32 and not shadow_done:
            o.cast(self.p, self.shadow)
        for name, (co, ci) in self.conv_dgrad.items():
            o.repack(self.master(name).reshape(-1), self.dgrad_pack[name], co, ci, 1)
        for fn in self.after_refresh:
            fn()

    def zero_grad(self, skip=None):
        """``skip``: an entry whose gradient the caller WRITES (not accumulates) later in the iteration."""
        self.sync()
        if skip is None:
            self.g.zero_()
        else:
            off, n, _ = self.entries[skip]
            self.g[:off].zero_()
            self.g[off + n:].zero_()

    def adam_step(self, hp: HyperParams, grad_scale=1.0):
        self.t += 1
        o = self.ops
        o.adam(self.p, self.g, self.m, self.v, None if o.tdtype == torch.float32 else self.shadow,
               hp.lr, hp.beta1, hp.beta2, hp.eps, self.t, grad_scale)
        self.refresh(shadow_done=True)

    def load_host(self, packed: dict):
        """packed: name -> CPU fp32 tensor of the entry's shape."""
        self.sync()
        host = torch.zeros(self.size, dtype=torch.float32)
        for name, t in packed.items():
            off, n, shape = self.entries[name]
            assert tuple(t.shape) == shape, (name, t.shape, shape)
            host[off:off + n] = t.reshape(-1)
        self.p.copy_(host)
        self.refresh()

    def export_state(self):
        """(p, m, v, t): the flat master parameters and Adam state (device tensors, by reference) for ``import_state`` of another
        store of the same network -- the layout of the flat buffer does not depend on the batch size."""
        self.sync()
        return self.p, self.m, self.v, self.t

    def import_state(self, state):
        p, m, v, t = state
        assert p.numel() == self.size, (p.numel(), self.size)
        self.sync()
        self.p.copy_(p); self.m.copy_(m); self.v.copy_(v)
        self.t = int(t)
        self.refresh()

    def to_host(self, buf=None):
        self.sync()
        h = (self.p if buf is None else buf).detach().cpu()
        return {name: h[off:off + n].view(shape).clone() for name, (off, n, shape) in self.entries.items()}


# =============================================================================================== critic
class NativeCritic:
    """reference DoWnGAN/networks/critic.py:9-106 (Critic(coarse_dim, fine_dim, nc))."""

    STRIDES = (1, 2, 1, 2, 1, 2, 1, 2)

    def __init__(self, ops, coarse_dim, fine_dim, nc, batch, stacked=False):
        assert fine_dim % 16 == 0, "critic.py:95 needs fine_dim divisible by 16"
        self.ops, self.B = ops, batch
        self.cd, self.fine, self.nc = coarse_dim, fine_dim, nc
        cd = coarse_dim
        self.c_real = [nc, cd, cd, 2 * cd, 2 * cd, 4 * cd, 4 * cd, 8 * cd, 8 * cd]
        self.c_pad = [layout.pad16(c) for c in self.c_real]
        self.convs = []
        h = fine_dim
        for l, st in enumerate(self.STRIDES):
            self.convs.append(Conv(batch, h, h, self.c_pad[l], self.c_pad[l + 1], st, False,
                                   cin_real=(nc if l == 0 and nc <= 2 else 0), net="C",
                                   cin_alg=self.c_real[l], cout_alg=self.c_real[l + 1]))
            h //= st
        self.hf = h
        self.fc_k = h * h * self.c_pad[8]
        P = self.P = ParamStore(ops)
        for l, cv in enumerate(self.convs):
            P.add(f"features.{2 * l}.weight", (cv.Cout, 9, cv.Cin))
            P.conv_dgrad[f"features.{2 * l}.weight"] = (cv.Cout, cv.Cin)
        P.add("features.0.bias", (self.c_pad[1],))
        P.add("classifier.0.weight", (FC_HID_P, self.fc_k))
        P.add("classifier.0.bias", (FC_HID_LD,))
        P.add("classifier.2.weight", (FC_OUT_P, FC_HID_LD))
        P.add("classifier.2.bias", (FC_OUT_P,))
        P.finalize()
        # activations / adjoints / tangents
        o = ops
        # 1-bit LeakyReLU' masks beside the activations (dg_epilogue.mask_bits / out_bits): the data-gradient and
        # tangent-forward epilogues are HBM-bound and the mask would otherwise be a full re-read of the activation.
        # Needs 64-channel wave tiles, i.e. every conv width >= 128 (cfg2: yes; the 16-filter configs keep plain masks).
        use_bits = os.environ.get("DG_NO_MASK_BITS") is None and all(c >= 128 and c % 64 == 0 for c in self.c_pad[1:])
        self.f8 = bool(getattr(o, "f8", False))
        # STACKED passes (critic_iteration_stacked): the real, the generated and the interpolated batch of a critic iteration go
        # through the layers as ONE batch of 3B -- a third of the launches, each three times as long (the conv kernels gain 4-9 %
        # forward, 1-3 % backward from the longer grids) -- so every per-sample buffer holds 3B samples; the single-pass methods
        # below work on the first B.  Needs the bit masks (the tangent pass reads a row range of them), the fused FC1 gradient and,
        # for the memory (3 x 40 GB of activations + adjoints at cfg2), no fp8 copies.
        self.stacked = bool(stacked) and use_bits and not self.f8 and 3 * batch <= 128 and os.environ.get("DG_NO_FC1_FUSED") is None
        # stacked = k > 1 (an int): only layers k - 1 ... 7 run as one batch of 3B; layers below it -- the ones that read or write the
        # first layer's 25-GB-at-3B tensors, which LOSE 5-8 % when stacked -- keep one launch per pass into row ranges of the same buffers
        self.stack_from = (int(stacked) - 1 if (stacked is not True and int(stacked) > 1) else 0) if self.stacked else 0
        cap = 3 * batch if self.stacked else batch
        self.cap = cap
        full = lambda cv: (cap,) + tuple(o.out_shape(cv))[1:]
        self._acts_all = [o.zeros(*full(cv)) for cv in self.convs[:7]] + [None]
        self._us_all = [o.zeros(*full(cv)) for cv in self.convs]
        self._bits_all = [o.zeros(*o.bits_shape(full(cv)), dtype=torch.int16) for cv in self.convs] if use_bits else None
        self.acts = [a[:batch] if a is not None else o.zeros(*o.out_shape(self.convs[7])) for a in self._acts_all]
        self.us = [u[:batch] for u in self._us_all]
        self.act_bits = [b[:batch] for b in self._bits_all] if use_bits else None
        self._h1pre_all = o.zeros(cap, FC_HID_LD, dtype=torch.float32)
        self._h1_all = o.zeros(cap, FC_HID_LD)
        self._outpre_all = o.zeros(cap, FC_OUT_P, dtype=torch.float32)
        self._out_all = o.zeros(cap, FC_OUT_P, dtype=torch.float32)
        self._dout_all = o.zeros(cap, FC_OUT_P, dtype=torch.float32)
        self.h1pre, self.h1, self.outpre = self._h1pre_all[:batch], self._h1_all[:batch], self._outpre_all[:batch]
        self.out, self.dout = self._out_all[:batch], self._dout_all[:batch]
        self.uh1 = o.zeros(batch, FC_HID_LD, dtype=torch.float32)
        self._tan = None
        # FC1's weight gradient (1.9 GB fp32 at cfg2) is formed ONCE per critic iteration from the rows of all three passes
        # (dg_linear_dw_wide) instead of being read-modify-written by each: slots 0-2 of these buffers hold FC1's input rows
        # (real activation, fake activation, penalty tangent) and the matching adjoint rows; slot 3 = passes without one.
        self.fc1_fused = 3 * batch <= 128 and os.environ.get("DG_NO_FC1_FUSED") is None
        if self.fc1_fused:
            self._y7_all = o.zeros(4 * batch, *self.acts[7].shape[1:])
            self._uh1_all = o.zeros(3 * batch, FC_HID_LD, dtype=torch.float32)
            self.acts[7] = self._y7_all[3 * batch:]
            if self.stacked:
                self._acts_all[7] = self._y7_all[:3 * batch]     # a stacked forward leaves FC1's rows in slots 0-2 directly
        # fp8 mode (BASELINE configs[4]): MXFP8 forms of every tensor an fp8 conv reads -- activations act[l-1] (forward of
        # layer l), adjoints us[l] (data gradient of layer l), the penalty's tangents, both weight packs -- written by the
        # PRODUCING launch's epilogue (dg_epilogue.out_q) or, for the weights, once per optimizer step; None = not needed
        qbuf = lambda t: (o.zeros(*t.shape, dtype=torch.uint8), o.zeros(*t.shape[:-1], t.shape[-1] // 32, dtype=torch.uint8))
        self.actq = [None] * 8
        self.usq = [None] * 8
        self.wq_f, self.wq_d = [None] * 8, [None] * 8
        self._tanq = None
        if self.f8:
            for l, cv in enumerate(self.convs):
                if l + 1 < 8 and o.f8_eligible(self.convs[l + 1], "fwd"):
                    self.actq[l] = qbuf(self.acts[l])
                if o.f8_eligible(cv, "dgrad"):
                    self.usq[l] = qbuf(self.us[l])
                    self.wq_d[l] = (o.zeros(cv.Cin * 9, cv.Cout, dtype=torch.uint8), o.zeros(cv.Cin * 9, cv.Cout // 32, dtype=torch.uint8))
                if o.f8_eligible(cv, "fwd"):
                    self.wq_f[l] = (o.zeros(cv.Cout * 9, cv.Cin, dtype=torch.uint8), o.zeros(cv.Cout * 9, cv.Cin // 32, dtype=torch.uint8))
            P.after_refresh.append(self._requantise_weights)
        # fp8 WEIGHT GRADIENTS (dg_conv3x3_wgrad_f8): the contraction runs over pixels, so the operands are second fp8 copies whose
        # scale does not vary along pixels -- one exponent per 32-channel block of the whole tensor, taken from the MXFP8 scale bytes
        # of the PREVIOUS pass over the same buffer (delayed scaling: ops.block_exp_max after the pass that consumed the copy).  The
        # producing epilogue writes the copy (dg_epilogue.out_u) beside the MXFP8 one: activation act[l-1] and adjoint us[l] of every
        # eligible layer l, and the penalty's tangents.  Until a role's exponents exist (first pass) its layers use the bf16 kernel.
        ubuf = lambda t: (o.zeros(*t.shape, dtype=torch.uint8), o.zeros(t.shape[-1] // 32, dtype=torch.uint8))
        # (layer 1 reads the first layer's megapixel activation: its im2col kernel writes the copies too -- and, once the weight gradient
        # is the fp8 kernel, no longer the 8.6-GB bf16 tensor itself: -7 ms per step at configs[1])
        self._l0_f8_out = bool(self.convs[0].cin_real) and self.convs[0].W % 16 == 0 and self.convs[0].Cout % 128 == 0
        self.wg8 = [self.f8 and l > 0 and (l > 1 or self._l0_f8_out) and bool(o.f8_eligible(cv, "wgrad")) and self.actq[l - 1] is not None
                    and self.usq[l] is not None for l, cv in enumerate(self.convs)]
        self.actu = [ubuf(self.acts[l]) if l + 1 < 8 and self.wg8[l + 1] else None for l in range(8)]
        self.usu = [ubuf(self.us[l]) if self.wg8[l] else None for l in range(8)]
        # the adjoints of the loss passes (d out = +-1 / B) and of the penalty pass (d out = 1) differ by the batch size: one exponent
        # set per kind of pass over the same byte buffers ("loss": usu[l][1], "gp": us_exp_gp[l])
        self.us_exp_gp = [o.zeros(self.us[l].shape[-1] // 32, dtype=torch.uint8) if self.wg8[l] else None for l in range(8)]
        self._us_gp_exp_ok = False
        self.tan_exp = [o.zeros(self.acts[l].shape[-1] // 32, dtype=torch.uint8) if l + 1 < 8 and self.wg8[l + 1] else None for l in range(8)]
        self._tanu = None
        self.skip_dead = os.environ.get("DG_F8_KEEP_BF16") is None      # do not store bf16 tensors that only fp8 readers follow (dg_epilogue.skip_y)
        # The first layer's 8.6-GB output (store-bound launch) and its tangent: once exponents exist they are written as the
        # uniform-scale copy ALONE -- layer 1's MXFP8 conv reads that copy with the exponents as one scale row for every pixel, its
        # fp8 weight gradient reads it anyway -- and the launch keeps the census of magnitudes the next exponents come from itself
        # (dg_epilogue.out_amax, ops.exp_from_amax): half the bytes of those launches.
        self.l0u = (self.f8 and bool(getattr(o, "f8_l0u", False)) and self._l0_f8_out and self.wg8[1] and self.skip_dead and self.act_bits is not None
                    and self.actu[0] is not None and self.tan_exp[0] is not None)
        nb0 = self.acts[0].shape[-1] // 32
        self._a0_amax = o.zeros(nb0, dtype=torch.int32) if self.l0u else None
        self._t0_amax = o.zeros(nb0, dtype=torch.int32) if self.l0u else None
        self._act_exp_ok = self._us_exp_ok = self._tan_exp_ok = False     # exponents of the role initialised by an earlier pass
        self._u_act_live = False                                          # this pass's forward wrote valid uniform-scale activations

    def _requantise_weights(self):
        o, P = self.ops, self.P
        for l, cv in enumerate(self.convs):
            name = f"features.{2 * l}.weight"
            off, n, _ = P.entries[name]
            if self.wq_f[l] is not None:
                o.quant_mxfp8(P.shadow[off:off + n].view(cv.Cout * 9, cv.Cin), *self.wq_f[l])
            if self.wq_d[l] is not None:
                o.quant_mxfp8(P.dgrad_pack[name].view(cv.Cin * 9, cv.Cout), *self.wq_d[l])

    # ---- state_dict interchange (reference key names / OIHW) -------------------------------------
    def load_state_dict(self, sd):
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).cpu()
        packed = {}
        for l, cv in enumerate(self.convs):
            packed[f"features.{2 * l}.weight"] = layout.pack_conv_weight(t(sd[f"features.{2 * l}.weight"]), cv.Cout, cv.Cin)
        packed["features.0.bias"] = layout.pack_bias(t(sd["features.0.bias"]), self.c_pad[1])
        packed["classifier.0.weight"] = layout.pack_fc1_weight(t(sd["classifier.0.weight"]), self.c_real[8], self.c_pad[8],
                                                              self.hf, self.hf, FC_HID_P)
        b1 = torch.zeros(FC_HID_LD); b1[:FC_HID] = t(sd["classifier.0.bias"])
        packed["classifier.0.bias"] = b1
        w2 = torch.zeros(FC_OUT_P, FC_HID_LD); w2[0, :FC_HID] = t(sd["classifier.2.weight"])[0]
        packed["classifier.2.weight"] = w2
        b2 = torch.zeros(FC_OUT_P); b2[0] = t(sd["classifier.2.bias"])[0]
        packed["classifier.2.bias"] = b2
        self.P.load_host(packed)

    def unpack(self, host):
        """packed host dict (params or grads) -> reference-shaped dict."""
        sd = {}
        for l in range(8):
            sd[f"features.{2 * l}.weight"] = layout.unpack_conv_weight(host[f"features.{2 * l}.weight"], self.c_real[l + 1], self.c_real[l])
        sd["features.0.bias"] = layout.unpack_bias(host["features.0.bias"], self.c_real[1])
        sd["classifier.0.weight"] = layout.unpack_fc1_weight(host["classifier.0.weight"], FC_HID, self.c_real[8], self.c_pad[8], self.hf, self.hf)
        sd["classifier.0.bias"] = host["classifier.0.bias"][:FC_HID].clone()
        sd["classifier.2.weight"] = host["classifier.2.weight"][:1, :FC_HID].clone()
        sd["classifier.2.bias"] = host["classifier.2.bias"][:1].clone()
        return sd

    def state_dict(self):
        return self.unpack(self.P.to_host())

    def grad_dict(self):
        return self.unpack(self.P.to_host(self.P.g))

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x, fc1_slot=None, for_wgrad=False):
        """critic.py:101-106.  x: NHWC [B, fine, fine, c_pad[0]].  Returns out[:, 0] (fp32 view).
        ``fc1_slot``: keep FC1's input rows of this pass for the iteration's single FC1 weight-gradient sweep.
        ``for_wgrad`` (fp8 mode): a backward with weight gradients follows -- also write the uniform-scale fp8 activations."""
        o, P = self.ops, self.P
        if self.fc1_fused:
            s = 3 if fc1_slot is None else fc1_slot
            self.acts[7] = self._y7_all[s * self.B:(s + 1) * self.B]
        cur = x
        want_u = self.f8 and for_wgrad and self._act_exp_ok
        self._u_act_live = want_u
        l0u = self.l0u and self._act_exp_ok           # the first layer's output: uniform-scale copy alone (every pass, once exponents exist)
        for l, cv in enumerate(self.convs):
            f8kw = dict(xq=self.actq[l - 1] if l else None, wq=self.wq_f[l], out_q=self.actq[l]) if self.f8 else {}
            if want_u and self.actu[l] is not None:
                f8kw["out_u"] = self.actu[l]
            if l0u and l == 0:
                f8kw = dict(out_u=self.actu[0], out_amax=self._a0_amax, skip_y=True)
            elif l0u and l == 1:
                f8kw["xq"] = self.actu[0]             # (bytes, block exponents): the exponents are the scale row of every pixel
            elif self.l0u and l == 0:                 # bootstrap pass: MXFP8 copy as before, and the census starts
                f8kw["out_amax"] = self._a0_amax
            # the bf16 activation itself is dead when the next conv reads the MXFP8 copy, the masks are bits and the only other
            # reader -- layer l + 1's weight gradient -- is the fp8 kernel (or does not run in this pass): not stored then
            if self.f8 and self.skip_dead and l <= 6 and self.actq[l] is not None and self.act_bits is not None \
                    and (o.f8_eligible(cv, "fwd") or (l == 0 and self._l0_f8_out)) \
                    and (not for_wgrad or (want_u and self.wg8[l + 1] and self._us_exp_ok)) and not (l0u and l == 0):
                f8kw["skip_y"] = True
            o.conv_fwd(cv, cur, P.w(f"features.{2 * l}.weight"), self.acts[l],
                       bias=P.master("features.0.bias") if l == 0 else None, act=C_SLOPE,
                       out_bits=self.act_bits[l] if self.act_bits else None, **f8kw)
            cur = self.acts[l]
        y7 = self.acts[7].view(self.B, self.fc_k)
        self.h1pre.zero_()
        o.linear_fwd(y7, P.w2d("classifier.0.weight"), self.h1pre, o_real=FC_HID, net="C")
        o.bias_act(self.h1pre, P.master("classifier.0.bias"), self.h1, act=C_SLOPE)
        self.outpre.zero_()
        o.linear_fwd(self.h1, P.w2d("classifier.2.weight"), self.outpre, o_real=1, net="C")
        o.bias_act(self.outpre, P.master("classifier.2.bias"), self.out)
        return self.out

    # ---- adjoint chain (backward of a forward just run on x) ---------------------------------------
    def backward(self, x, dout_value, wgrad=True, dx=None, fc1_slot=None, u_role="loss"):
        """d(out_b)/d(.) * dout_value for every sample.  wgrad: accumulate parameter gradients
        (autograd backward of wasserstein.py:52); dx: if given, receives the input gradient
        (wasserstein.py:100-106 / :80).  ``fc1_slot``: FC1's adjoint rows go to that slot and its weight gradient is left
        to ``fc1_flush`` (the forward must have used the same slot)."""
        o, P = self.ops, self.P
        uh1 = self.uh1 if fc1_slot is None else self._uh1_all[fc1_slot * self.B:(fc1_slot + 1) * self.B]
        self.dout.zero_()
        o.fill_col(self.dout, 0, dout_value)
        y7 = self.acts[7].view(self.B, self.fc_k)
        if wgrad:
            o.linear_dw(self.dout, self.h1, P.grad("classifier.2.weight"), o_real=1, net="C")
            o.colsum(self.dout, P.grad("classifier.2.bias"))
        o.linear_dx(self.dout, P.w2d("classifier.2.weight"), uh1, mask=self.h1, mask_slope=C_SLOPE, o_real=1, net="C")
        if wgrad:
            if fc1_slot is None:
                o.linear_dw(uh1[:, :FC_HID_P], y7, P.grad("classifier.0.weight"), o_real=FC_HID, net="C")
            o.colsum(uh1, P.grad("classifier.0.bias"))
        o.linear_dx(uh1[:, :FC_HID_P], P.w2d("classifier.0.weight"), self.us[7].view(self.B, self.fc_k),
                    mask=y7, mask_slope=C_SLOPE, o_real=FC_HID, net="C")
        # fp8 weight gradients: which exponent set this pass's adjoint copies are written with (None: nobody reads them)
        us_u = self.f8 and ((u_role == "loss" and wgrad and self._us_exp_ok) or (u_role == "gp" and self._us_gp_exp_ok))
        uexp = (lambda l: self.usu[l][1] if u_role == "loss" else self.us_exp_gp[l])
        if self.usq[7] is not None:       # the only adjoint that does not come out of a conv epilogue
            o.quant_mxfp8(self.us[7], *self.usq[7])
            if self.usu[7] is not None and us_u:
                o.quant_uniform(self.us[7], self.usu[7][0], uexp(7))
        for l in range(7, -1, -1):
            cv = self.convs[l]
            name = f"features.{2 * l}.weight"
            xin = self.acts[l - 1] if l > 0 else x
            if wgrad:   # features.0 also carries the only conv bias of the critic (critic.py:21-23)
                if self.f8 and self.wg8[l] and us_u and self._u_act_live:
                    o.conv_wgrad_f8(cv, self.actu[l - 1][0], self.actu[l - 1][1], self.usu[l][0], uexp(l), P.grad(name).reshape(-1))
                else:
                    o.conv_wgrad(cv, xin, self.us[l], P.grad(name).reshape(-1), db=P.grad("features.0.bias") if l == 0 else None)
            if l > 0:
                f8kw = dict(xq=self.usq[l], wq=self.wq_d[l], out_q=self.usq[l - 1]) if self.f8 else {}
                if us_u and self.usu[l - 1] is not None:
                    f8kw["out_u"] = (self.usu[l - 1][0], uexp(l - 1))
                # bf16 adjoint us[l-1]: read only by layer l-1's weight gradient (the next data gradient reads the MXFP8 copy) -- dead when
                # that is the fp8 kernel, or no weight gradient of this pass's adjoints runs at all (generator iteration)
                if self.f8 and self.skip_dead and l - 1 >= 1 and self.usq[l - 1] is not None and self.act_bits and o.f8_eligible(cv, "dgrad"):
                    if u_role == "loss":
                        dead = (not wgrad) or (self.wg8[l - 1] and us_u and self._u_act_live)
                    else:       # penalty pass: the tangent weight gradients read the adjoints
                        dead = self.wg8[l - 1] and us_u and self._tan_exp_ok
                    if dead:
                        f8kw["skip_y"] = True
                if self.act_bits:
                    o.conv_dgrad(cv, self.us[l], P.wd(name), self.us[l - 1], mask_bits=self.act_bits[l - 1], mask_slope=C_SLOPE, **f8kw)
                else:
                    o.conv_dgrad(cv, self.us[l], P.wd(name), self.us[l - 1], mask=self.acts[l - 1], mask_slope=C_SLOPE, **f8kw)
            elif dx is not None:
                o.conv_dgrad(cv, self.us[0], P.wd(name), dx)
        self._us_live = us_u
        if self.f8 and u_role == "loss" and wgrad:     # (the penalty pass updates after its tangent weight gradients have read the copies)
            self._update_exponents("loss")

    def _update_exponents(self, role):
        """After a pass's weight gradients have consumed the uniform-scale copies: the exponents the NEXT pass of the same kind writes
        them with = the largest MXFP8 block exponent this pass's tensors reached (+1: values may grow from pass to pass; saturation
        at +-448 beyond that, like the MX rule itself).  One launch for all tensors of the pass."""
        pairs = []
        for l in range(8):
            if role == "loss" and self.actu[l] is not None:
                if l == 0 and self.l0u:       # the first layer's launches kept the census themselves (no MXFP8 scale bytes in that mode)
                    self.ops.exp_from_amax(self._a0_amax, self.actu[0][1])
                    continue
                pairs.append((self.actq[l][1], self.actu[l][1]))
            if self.usu[l] is not None:
                pairs.append((self.usq[l][1], self.usu[l][1] if role == "loss" else self.us_exp_gp[l]))
        if pairs:
            self.ops.block_exp_max_batch(pairs)
            if role == "loss":
                self._act_exp_ok = self._us_exp_ok = True
            else:
                self._us_gp_exp_ok = True

    # ---- gradient penalty: forward, adjoint, norm, tangent forward + weight gradients -------------
    def fc1_flush(self, accumulate=False):
        """The FC1 weight gradient of the passes parked in slots 0-2 (real, fake, penalty tangent): one sweep."""
        n = 3 * self.B
        self.ops.linear_dw_wide(self._uh1_all[:, :FC_HID_P], self._y7_all[:n].view(n, self.fc_k),
                                self.P.grad("classifier.0.weight"), accumulate=accumulate, o_real=FC_HID, net="C")

    def gp_pass(self, xhat, g_buf, v_buf, ss, coef, gp_scalar, hp: HyperParams, b_global, fc1_slot=None):
        """wasserstein.py:87-117 and its contribution to critic_loss.backward (:52).

        g = d sum_b C(xhat_b) / d xhat (adjoint chain with grad_outputs = 1).  With piecewise-linear
        activations the penalty's parameter gradient is  sum_l wgrad(tangent_{l-1}, adjoint_l)  where
        the tangent is v0 = dGP/dg pushed forward through the same masked linear maps."""
        o, P = self.ops, self.P
        self.forward(xhat)
        self.backward(xhat, 1.0, wgrad=False, dx=g_buf, fc1_slot=fc1_slot, u_role="gp")
        self.gp_tangent(g_buf, v_buf, ss, coef, gp_scalar, hp, b_global, 0, fc1_slot)
        if self.f8:
            self._update_exponents("gp")

    def gp_tangent(self, g_buf, v_buf, ss, coef, gp_scalar, hp: HyperParams, b_global, r0, fc1_slot):
        """Second half of the penalty (after g = dC/dx-hat is in ``g_buf``): norm, v0 = dGP/dg, tangent forward and the
        penalty's weight gradients.  ``r0``: first sample of the x-hat pass in the per-sample buffers (0, or 2B when stacked)."""
        o, P, B = self.ops, self.P, self.B
        rows = slice(r0, r0 + B)
        us = [u[rows] for u in self._us_all]
        bits = [b[rows] for b in self._bits_all] if self._bits_all is not None else None
        acts = self.acts if r0 == 0 else [a[rows] for a in self._acts_all]      # (masks of the 16-filter configs; r0 = 0 there)
        h1 = self._h1_all[rows]
        uh1 = self.uh1 if fc1_slot is None else self._uh1_all[fc1_slot * B:(fc1_slot + 1) * B]
        ss.zero_()
        o.sumsq_rows(g_buf, ss)
        o.gp_finish(ss, B, b_global, hp.gp_lambda, hp.gp_lambda, coef, gp_scalar)
        o.scale_rows(g_buf, coef, v_buf)
        if self._tan is None:
            big = max(a.numel() for a in self.acts)
            self._tan = [o.zeros(big), o.zeros(big)]
            self._th1pre = o.zeros(B, FC_HID_LD, dtype=torch.float32)
            self._th1 = o.zeros(B, FC_HID_LD)
            self._ones = o.zeros(B, FC_OUT_P, dtype=torch.float32)
            o.fill_col(self._ones, 0, 1.0)
            if self.f8:
                self._tanq = [(o.zeros(big, dtype=torch.uint8), o.zeros(big // 32, dtype=torch.uint8)) for _ in range(2)]
                if any(e is not None for e in self.tan_exp):
                    self._tanu = [o.zeros(big, dtype=torch.uint8) for _ in range(2)]
        t, tq, tu = v_buf, None, None          # the tangent t_{l-1}, its MXFP8 form, its uniform-scale form (bytes, exponents)
        t0_census = False                      # l0u: t_0 exists as its uniform-scale copy alone (tq = that copy, the exponents its scale row)
        for l, cv in enumerate(self.convs):
            name = f"features.{2 * l}.weight"
            if self.f8 and self.wg8[l] and tu is not None and getattr(self, "_us_live", False) and r0 == 0:
                o.conv_wgrad_f8(cv, tu[0], tu[1], self.usu[l][0], self.us_exp_gp[l], P.grad(name).reshape(-1))
            else:
                o.conv_wgrad(cv, t, us[l], P.grad(name).reshape(-1))
            if self.f8 and l > 0 and self.tan_exp[l - 1] is not None and tq is not None and not (l == 1 and t0_census):
                o.block_exp_max(tq[1], self.tan_exp[l - 1])     # t_{l-1}'s copy has been consumed: exponents for the next tangent pass
            tn = self._tan[l & 1][:self.acts[l].numel()].view(self.acts[l].shape)
            if l == 7 and fc1_slot is not None:           # FC1's tangent input rows stay for fc1_flush
                tn = self._y7_all[fc1_slot * B:(fc1_slot + 1) * B]
            f8kw, tqn = {}, None
            if self.f8:
                if self.actq[l] is not None:          # the next layer's tangent forward is an fp8 conv
                    n, sh = self.acts[l].numel(), self.acts[l].shape
                    tqn = (self._tanq[l & 1][0][:n].view(sh), self._tanq[l & 1][1][:n // 32].view(*sh[:-1], sh[-1] // 32))
                f8kw = dict(xq=tq, wq=self.wq_f[l], out_q=tqn)
            tun = None
            if self.f8 and self.tan_exp[l] is not None and tqn is not None and self._tan_exp_ok:
                tun = (self._tanu[l & 1][:self.acts[l].numel()].view(self.acts[l].shape), self.tan_exp[l])
                f8kw["out_u"] = tun
                if self.skip_dead and bits and l < 7 and self.wg8[l + 1] and getattr(self, "_us_live", False) \
                        and (o.f8_eligible(cv, "fwd") or (l == 0 and self._l0_f8_out)):
                    f8kw["skip_y"] = True        # the bf16 tangent: read by the next conv (MXFP8 copy) and layer l + 1's fp8 weight gradient only
            if self.l0u and l == 0 and tun is not None and f8kw.get("skip_y") and r0 == 0:
                f8kw = dict(out_u=tun, out_amax=self._t0_amax, skip_y=True)
                tqn, t0_census = tun, True
            if bits:
                o.conv_fwd(cv, t, P.w(name), tn, mask_bits=bits[l], mask_slope=C_SLOPE, **f8kw)
            else:
                o.conv_fwd(cv, t, P.w(name), tn, mask=acts[l], mask_slope=C_SLOPE, **f8kw)
            if l == 1 and t0_census:          # layer 1's launches have read t_0's copy with the old exponents: the next pass's from the census
                o.exp_from_amax(self._t0_amax, self.tan_exp[0])
            t, tq, tu = tn, tqn, tun
        if self.f8 and any(e is not None for e in self.tan_exp):
            self._tan_exp_ok = True
        t7 = t.view(B, self.fc_k)
        if fc1_slot is None:
            o.linear_dw(uh1[:, :FC_HID_P], t7, P.grad("classifier.0.weight"), o_real=FC_HID, net="C")
        self._th1pre.zero_()
        o.linear_fwd(t7, P.w2d("classifier.0.weight"), self._th1pre, o_real=FC_HID, net="C")
        o.bias_act(self._th1pre, None, self._th1, mask=h1, mask_slope=C_SLOPE)
        o.linear_dw(self._ones, self._th1, P.grad("classifier.2.weight"), o_real=1, net="C")

    # ---- stacked passes: real | fake | x-hat as one batch of 3B (see __init__) ------------------------------------------
    def _cvn(self, n):
        if n not in self._cv_cache:
            self._cv_cache[n] = [dataclasses.replace(cv, N=n) for cv in self.convs]
        return self._cv_cache[n]

    _cv_cache = None

    def forward_stacked(self, x3):
        """critic.py:101-106 on x3 = [real | fake | x-hat] (3B samples, compact or padded NHWC).  Returns out[3B, .] (fp32)."""
        o, P, B = self.ops, self.P, self.B
        if self._cv_cache is None:
            self._cv_cache = {}
        cur = x3
        for l, cv in enumerate(self._cvn(3 * B)):
            w, bias = P.w(f"features.{2 * l}.weight"), P.master("features.0.bias") if l == 0 else None
            if l < self.stack_from:           # one launch per pass (real | fake | x-hat) on row ranges of the stacked buffers
                for p in range(3):
                    r = slice(p * B, (p + 1) * B)
                    o.conv_fwd(self.convs[l], cur[r], w, self._acts_all[l][r], bias=bias, act=C_SLOPE, out_bits=self._bits_all[l][r])
            else:
                o.conv_fwd(cv, cur, w, self._acts_all[l], bias=bias, act=C_SLOPE, out_bits=self._bits_all[l])
            cur = self._acts_all[l]
        y7 = self._acts_all[7].view(3 * B, self.fc_k)
        self._h1pre_all.zero_()
        for p in range(3):                                # the split-K FC1 kernel takes up to 32 rows per call
            rows = slice(p * B, (p + 1) * B)
            o.linear_fwd(y7[rows], P.w2d("classifier.0.weight"), self._h1pre_all[rows], o_real=FC_HID, net="C")
        o.bias_act(self._h1pre_all, P.master("classifier.0.bias"), self._h1_all, act=C_SLOPE)
        self._outpre_all.zero_()
        for p in range(3):
            rows = slice(p * B, (p + 1) * B)
            o.linear_fwd(self._h1_all[rows], P.w2d("classifier.2.weight"), self._outpre_all[rows], o_real=1, net="C")
        o.bias_act(self._outpre_all, P.master("classifier.2.bias"), self._out_all)
        return self._out_all

    def backward_stacked(self, x3, dout_values, dx):
        """Adjoint chain of forward_stacked: d out_b = dout_values[pass of b]; parameter gradients from the first TWO passes
        (wasserstein.py:52: the real and the generated batch; FC1's is left to fc1_flush), ``dx`` = input gradient of the THIRD
        (wasserstein.py:100-106: g = dC/dx-hat, whose adjoints stay in rows [2B, 3B) for the tangent pass)."""
        o, P, B = self.ops, self.P, self.B
        n3, w2 = 3 * B, slice(0, 2 * B)
        dout, uh1, h1 = self._dout_all, self._uh1_all, self._h1_all
        dout.zero_()
        for p in range(3):
            o.fill_col(dout[p * B:(p + 1) * B], 0, dout_values[p])
        y7 = self._acts_all[7].view(n3, self.fc_k)
        o.linear_dw(dout[w2], h1[w2], P.grad("classifier.2.weight"), o_real=1, net="C")
        o.colsum(dout[w2], P.grad("classifier.2.bias"))
        o.linear_dx(dout, P.w2d("classifier.2.weight"), uh1, mask=h1, mask_slope=C_SLOPE, o_real=1, net="C")
        o.colsum(uh1[w2], P.grad("classifier.0.bias"))
        o.linear_dx(uh1[:, :FC_HID_P], P.w2d("classifier.0.weight"), self._us_all[7].view(n3, self.fc_k),
                    mask=y7, mask_slope=C_SLOPE, o_real=FC_HID, net="C")
        cv3, cv2, cv1 = self._cvn(n3), self._cvn(2 * B), self.convs
        for l in range(7, -1, -1):
            name = f"features.{2 * l}.weight"
            xin = self._acts_all[l - 1] if l > 0 else x3
            db = P.grad("features.0.bias") if l == 0 else None
            per_pass = l < self.stack_from
            if per_pass:
                for p in range(2):
                    r = slice(p * B, (p + 1) * B)
                    o.conv_wgrad(cv1[l], xin[r], self._us_all[l][r], P.grad(name).reshape(-1), db=db)
            else:
                o.conv_wgrad(cv2[l], xin[w2], self._us_all[l][w2], P.grad(name).reshape(-1), db=db)
            if l > 0 and per_pass:
                for p in range(3):
                    r = slice(p * B, (p + 1) * B)
                    o.conv_dgrad(cv1[l], self._us_all[l][r], P.wd(name), self._us_all[l - 1][r], mask_bits=self._bits_all[l - 1][r], mask_slope=C_SLOPE)
            elif l > 0:
                o.conv_dgrad(cv3[l], self._us_all[l], P.wd(name), self._us_all[l - 1], mask_bits=self._bits_all[l - 1], mask_slope=C_SLOPE)
            else:
                o.conv_dgrad(cv1[0], self._us_all[0][2 * B:], P.wd(name), dx)


# =============================================================================================== generator
class NativeGenerator:
    """reference DoWnGAN/networks/generator.py:56-90
    (Generator(filters, fine_dims, channels, n_predictands=2, num_res_blocks=16, num_upsample=3))."""

    def __init__(self, ops, filters, channels, batch, coarse_side, n_predictands=2, num_res_blocks=16, num_upsample=3):
        assert filters % 16 == 0, "native path needs filters to be a multiple of 16"
        self.ops, self.B, self.S = ops, batch, coarse_side
        self.F, self.cin, self.npred = filters, channels, n_predictands
        self.cin_p, self.np_p = layout.pad16(channels), layout.pad16(n_predictands)
        self.nrb, self.nup = num_res_blocks, num_upsample
        F_, S, B = filters, coarse_side, batch
        self.cv_conv1 = Conv(B, S, S, self.cin_p, F_, cin_real=(channels if channels <= 2 else 0), cin_alg=channels)
        self.cv_b = [Conv(B, S, S, k * F_, F_, net="G") for k in range(1, 6)]      # net="G": fp8-eligible in f8_generator mode
        self.cv_conv2 = Conv(B, S, S, F_, F_, net="G")
        self.cv_up = [Conv(B, S << u, S << u, F_, 4 * F_, 1, True, net="T") for u in range(num_upsample)]   # net="T": fp8-eligible forward (f8_gtail)
        hs = S << num_upsample
        self.cv_c30 = Conv(B, hs, hs, F_, F_, net="T")
        self.cv_c32 = Conv(B, hs, hs, F_, self.np_p, cout_alg=n_predictands)
        # conv3.2 has <= 2 real OUTPUT channels: its backward is HBM-bound and runs on the first-layer (im2col) kernels with the
        # operand roles swapped -- data gradient = forward conv of dfake with the mirrored-tap pack, weight gradient =
        # wgrad(x := dfake, dy := c30) folded back by wgrad_unswap (dg_repack_conv_weights kind 2 / dg_wgrad_unswap)
        self.cv_c32_bwd = Conv(B, hs, hs, self.np_p, F_, cin_real=n_predictands, cin_alg=n_predictands) if n_predictands <= 2 else None
        P = self.P = ParamStore(ops)

        def addconv(name, cv, dgrad=True):
            P.add(name + ".weight", (cv.Cout, 9, cv.Cin))
            P.add(name + ".bias", (cv.Cout,))
            if dgrad:
                P.conv_dgrad[name + ".weight"] = (cv.Cout, cv.Cin)
        addconv("conv1", self.cv_conv1, dgrad=False)
        for i in range(num_res_blocks):
            for j in range(3):
                for k in range(1, 6):       # the dense blocks' data gradients run on the stacked packs below, not per conv
                    addconv(f"res_blocks.{i}.dense_blocks.{j}.b{k}.0", self.cv_b[k - 1], dgrad=False)
        addconv("conv2", self.cv_conv2)
        for u in range(num_upsample):
            addconv(f"upsampling.{3 * u}", self.cv_up[u])
        addconv("conv3.0", self.cv_c30)
        addconv("conv3.2", self.cv_c32)
        P.finalize()
        o = ops
        self.ndrb = 3 * num_res_blocks
        self.out1 = o.zeros(B, S, S, F_)
        self.trunk = o.zeros(B, S, S, F_)
        self.ups = [o.zeros(B, S << (u + 1), S << (u + 1), F_) for u in range(num_upsample)]
        self.c30 = o.zeros(B, hs, hs, F_)
        self.fake = o.zeros(B, hs, hs, self.np_p)
        self._ring = [o.zeros(B, S, S, 5 * F_) for _ in range(4)]
        self._saved = None
        self._bwd = None
        # fp8 mode (HipOps(f8_generator=True)): the FORWARD of the dense-block trunk (240 convs + conv2, 84 % of the generator's
        # flops) runs on the MXFP8 kernel.  Every slab gets an fp8 form [B,S,S,5F] + scales [B,S,S,5F/32] (four rotating ones: the
        # backward never reads them); conv k's epilogue writes its slice of it, the next conv reads the first k+1 slices.
        self.f8 = bool(getattr(o, "f8_generator", False)) and all(o.f8_eligible(cv, "fwd") for cv in self.cv_b)
        self._qring = [(o.zeros(B, S, S, 5 * F_, dtype=torch.uint8), o.zeros(B, S, S, 5 * F_ // 32, dtype=torch.uint8)) for _ in range(4)] if self.f8 else None
        # ... and the FORWARD of the up-sampling tail (upsampling.*, conv3.0: generator.py:69-81,88-89): every tail tensor gets an MXFP8
        # form written by its producer's epilogue (conv2: residual + copy; up-sampling convs: the copy in SHUFFLED pixel order); a
        # forward nobody differentiates (critic iterations) does not store the bf16 up-sampled tensors at all.  conv3.2 (2 real
        # output channels, HBM-bound) and the tail's backward stay bf16.
        self.f8_tail = self.f8 and all(o.f8_eligible(cv, "fwd") for cv in self.cv_up + [self.cv_c30])
        self._tq = None
        self._w_c32_bwd = o.zeros(F_ * 9 * self.np_p) if self.cv_c32_bwd is not None else None
        self._dw_c32_tmp = None
        if self.cv_c32_bwd is not None:
            P.after_refresh.append(lambda: o.repack(P.master("conv3.2.weight").reshape(-1), self._w_c32_bwd, self.np_p, F_, 2))
        self._wq = {}
        self._qlast = None
        # Dense-block backward (generator.py:14-41).  Conv k of a block reads slab channels [0, kF), so autograd hands the block
        # five data gradients with 128 reduction channels and k*128 output channels each, the later ones accumulating into the
        # earlier ones' output: 18 tap-steps per 64-KB read-modify-write epilogue (930-1090 TFLOP/s against the forward's 1250-1290).
        # The same sums GATHERED by slab slice: the adjoint of slice j is ONE data gradient over the stacked adjoints of convs
        # j+1..5 -- a "virtual" conv of F input and (5-j)F output channels whose weight rows are the slice-j columns of those
        # convs -- i.e. exactly the forward's shapes (reduction 128..640 channels, 128 outputs, one plain epilogue per tile).
        self.cv_v = [Conv(B, S, S, F_, (5 - j) * F_, net="G") for j in range(5)]
        self._vsize = [(5 - j) * F_ * 9 * F_ for j in range(5)]
        self._vpack = [o.zeros(sum(self._vsize)) for _ in range(self.ndrb)]
        # fp8 mode: the dense blocks' DATA GRADIENTS on the MXFP8 kernel too (their reduction runs over the stacked adjoint channels:
        # the adjoint slab gets an MXFP8 form written by the producing epilogues, the virtual packs are quantised after every step)
        self.f8_bwd = self.f8 and bool(getattr(o, "f8_gbwd", False)) and all(o.f8_eligible(cv, "dgrad") for cv in self.cv_v)
        # ... and their WEIGHT GRADIENTS on the fp8 kernel (dg_conv3x3_wgrad_dense_f8; critic: NativeCritic.wg8): uniform-scale E4M3 copies
        # of every saved activation slab and of the two rotating adjoint slabs, written by the producing epilogues (out_u) with the
        # block exponents the PREVIOUS generator iteration's tensors reached (delayed scaling: `_ex` / `_eu` [block][5F / 32], refreshed
        # at the end of backward from the `_new` sets collected during the pass); until a first iteration has produced exponents the
        # bf16 kernel runs.  Bias gradients: column sums of the bf16 adjoint slices.
        self.f8_wg = (self.f8_bwd and bool(getattr(o, "f8_wgrad", False)) and bool(getattr(o, "f8_gwgrad", False)) and F_ == 128 and S % 64 == 0)
        self._slab_u = self._us_u = self._ex = self._eu = self._ex_new = self._eu_new = None
        self._u_ok = self._u_live = False
        P.after_refresh.append(self._rebuild_vpacks)
        if self.f8:
            P.after_refresh.append(self._requantise_weights)

    def _f8_names(self):
        for i in range(self.nrb):
            for j in range(3):
                for k in range(1, 6):
                    yield f"res_blocks.{i}.dense_blocks.{j}.b{k}.0", self.cv_b[k - 1]
        yield "conv2", self.cv_conv2
        if self.f8_tail:
            for u in range(self.nup):
                yield f"upsampling.{3 * u}", self.cv_up[u]
            yield "conv3.0", self.cv_c30

    def _requantise_weights(self):
        o, P = self.ops, self.P
        for name, cv in self._f8_names():
            if name not in self._wq:
                self._wq[name] = (o.zeros(cv.Cout * 9, cv.Cin, dtype=torch.uint8), o.zeros(cv.Cout * 9, cv.Cin // 32, dtype=torch.uint8))
            off, n, _ = P.entries[name + ".weight"]
            o.quant_mxfp8(P.shadow[off:off + n].view(cv.Cout * 9, cv.Cin), *self._wq[name])

    # ---- state_dict interchange ------------------------------------------------------------------
    def _conv_names(self):
        yield "conv1", self.cv_conv1, self.cin, self.F
        for i in range(self.nrb):
            for j in range(3):
                for k in range(1, 6):
                    yield f"res_blocks.{i}.dense_blocks.{j}.b{k}.0", self.cv_b[k - 1], k * self.F, self.F
        yield "conv2", self.cv_conv2, self.F, self.F
        for u in range(self.nup):
            yield f"upsampling.{3 * u}", self.cv_up[u], self.F, 4 * self.F
        yield "conv3.0", self.cv_c30, self.F, self.F
        yield "conv3.2", self.cv_c32, self.F, self.npred

    def load_state_dict(self, sd):
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).cpu()
        packed = {}
        for name, cv, ci, co in self._conv_names():
            packed[name + ".weight"] = layout.pack_conv_weight(t(sd[name + ".weight"]), cv.Cout, cv.Cin, cv.pixel_shuffle)
            packed[name + ".bias"] = layout.pack_bias(t(sd[name + ".bias"]), cv.Cout, cv.pixel_shuffle)
        self.P.load_host(packed)

    def unpack(self, host):
        sd = {}
        for name, cv, ci, co in self._conv_names():
            sd[name + ".weight"] = layout.unpack_conv_weight(host[name + ".weight"], co, ci, cv.pixel_shuffle)
            sd[name + ".bias"] = layout.unpack_bias(host[name + ".bias"], co, cv.pixel_shuffle)
        return sd

    def state_dict(self):
        return self.unpack(self.P.to_host())

    def grad_dict(self):
        return self.unpack(self.P.to_host(self.P.g))

    def vpack(self, d, j):
        """data-gradient pack of dense block d's virtual conv for slab slice j (see __init__)"""
        off = sum(self._vsize[:j])
        return self._vpack[d][off:off + self._vsize[j]]

    def _rebuild_vpacks(self):
        """after every optimizer step / load: rows (k-j-1)F.. of slice j's virtual weight = columns [jF, (j+1)F) of conv k."""
        o, P = self.ops, self.P
        for d in range(self.ndrb):
            pre = f"res_blocks.{d // 3}.dense_blocks.{d % 3}.b"
            o.repack_dense([P.view(P.p, f"{pre}{k}.0.weight").reshape(-1) for k in range(1, 6)], self._vpack[d], self.F)
        if self.f8_bwd:          # MXFP8 forms of the virtual data-gradient packs [F * 9][(5 - j) F] (reduction = the stacked adjoint channels)
            F_ = self.F
            if self._vq is None:
                self._vq = [[(o.zeros(F_ * 9, (5 - j) * F_, dtype=torch.uint8), o.zeros(F_ * 9, (5 - j) * F_ // 32, dtype=torch.uint8)) for j in range(5)]
                            for _ in range(self.ndrb)]
            for d in range(self.ndrb):
                for j in range(5):
                    o.quant_mxfp8(self.vpack(d, j).view(F_ * 9, (5 - j) * F_), *self._vq[d][j])

    _vq = None

    # ---- forward -----------------------------------------------------------------------------------
    def _slab(self, d, save):
        if save:
            if self._saved is None:
                o = self.ops
                self._saved = [o.zeros(self.B, self.S, self.S, 5 * self.F) for _ in range(self.ndrb)]
                self._saved.append(o.zeros(self.B, self.S, self.S, self.F))
            return self._saved[d]
        return self._ring[d % 4]

    def forward(self, x, save=False):
        """generator.py:83-90.  x: NHWC [B,S,S,cin_p].  save=True keeps every dense-block slab for backward."""
        o, P, F_ = self.ops, self.P, self.F
        W = lambda n: P.w(n + ".weight")
        Bz = lambda n: P.master(n + ".bias")
        o.conv_fwd(self.cv_conv1, x, W("conv1"), self.out1, bias=Bz("conv1"))
        o.axpby(self._slab(0, save)[..., :F_], self.out1)
        f8 = self.f8
        def qs(d, c0, c1):
            """fp8 form (bytes, scales) of channels [c0, c1) of slab d, with that slab's strides"""
            if save and d == self.ndrb:           # the saved trunk output is an F-wide tensor of its own, not a 5F slab
                if self._qlast is None:
                    self._qlast = (o.zeros(self.B, self.S, self.S, F_, dtype=torch.uint8), o.zeros(self.B, self.S, self.S, F_ // 32, dtype=torch.uint8))
                return self._qlast
            return self._qring[d % 4][0][..., c0:c1], self._qring[d % 4][1][..., c0 // 32:c1 // 32]
        if f8:
            o.quant_mxfp8(self._slab(0, save)[..., :F_], *qs(0, 0, F_))
        # fp8 weight gradients: this (saved) pass writes the uniform-scale copies of its slabs when exponents exist, and collects the
        # exponents for the next generator iteration either way
        wg8 = self.f8_wg and save
        if wg8 and self._slab_u is None:
            nb = 5 * F_ // 32
            self._slab_u = [o.zeros(self.B, self.S, self.S, 5 * F_, dtype=torch.uint8) for _ in range(self.ndrb)]
            self._us_u = [o.zeros(self.B, self.S, self.S, 5 * F_, dtype=torch.uint8) for _ in range(2)]
            self._ex, self._eu, self._ex_new, self._eu_new = (o.zeros(self.ndrb, nb, dtype=torch.uint8) for _ in range(4))
        u_live = wg8 and self._u_ok
        if save:
            self._u_live = u_live
        def su(d, c0, c1):
            """out_u operand for channels [c0, c1) of saved slab d: (bytes, this pass's exponents of those blocks)"""
            return self._slab_u[d][..., c0:c1], self._ex[d][c0 // 32:c1 // 32]
        if u_live:
            o.quant_uniform(self._slab(0, save)[..., :F_], *su(0, 0, F_))
        for i in range(self.nrb):
            rrdb_in = self._slab(3 * i, save)[..., :F_]
            for j in range(3):
                d = 3 * i + j
                slab, nxt = self._slab(d, save), self._slab(d + 1, save)
                pre = f"res_blocks.{i}.dense_blocks.{j}.b"
                for k in range(1, 5):
                    kw = dict(xq=qs(d, 0, k * F_), wq=self._wq[f"{pre}{k}.0"], out_q=qs(d, k * F_, (k + 1) * F_)) if f8 else {}
                    if u_live:
                        kw["out_u"] = su(d, k * F_, (k + 1) * F_)
                    o.conv_fwd(self.cv_b[k - 1], slab[..., :k * F_], W(f"{pre}{k}.0"), slab[..., k * F_:(k + 1) * F_],
                               bias=Bz(f"{pre}{k}.0"), act=G_SLOPE, **kw)
                ep = dict(bias=Bz(f"{pre}5.0"), r1=slab[..., :F_], s1=RES_SCALE)          # generator.py:41
                if j == 2:
                    ep.update(r2=rrdb_in, s2=RES_SCALE)                                  # generator.py:53
                if f8:
                    ep.update(xq=qs(d, 0, 5 * F_), wq=self._wq[f"{pre}5.0"], out_q=qs(d + 1, 0, F_))
                    if u_live and d + 1 < self.ndrb:
                        ep["out_u"] = su(d + 1, 0, F_)
                o.conv_fwd(self.cv_b[4], slab, W(f"{pre}5.0"), nxt[..., :F_], **ep)
            if wg8:      # the RRDB's three slabs are complete in the fp8 ring (four entries: the conv above wrote slice 0 of a fourth)
                o.block_exp_max_batch([(self._qring[d % 4][1], self._ex_new[d]) for d in range(3 * i, 3 * i + 3)])
        kw = dict(xq=qs(self.ndrb, 0, F_), wq=self._wq["conv2"]) if f8 else {}
        f8t = self.f8_tail
        if f8t:
            if self._tq is None:
                qb = lambda t: (o.zeros(*t.shape, dtype=torch.uint8), o.zeros(*t.shape[:-1], F_ // 32, dtype=torch.uint8))
                self._tq = [qb(self.trunk)] + [qb(t) for t in self.ups]
            kw["out_q"] = self._tq[0]
        o.conv_fwd(self.cv_conv2, self._slab(self.ndrb, save)[..., :F_], W("conv2"), self.trunk, bias=Bz("conv2"),
                   r1=self.out1, s1=1.0, **kw)                                           # generator.py:86-87
        cur = self.trunk
        for u in range(self.nup):
            name = f"upsampling.{3 * u}"
            # (bf16 up-sampled tensor: read by the backward only -- masks, weight-gradient inputs)
            kw = dict(xq=self._tq[u], wq=self._wq[name], out_q=self._tq[u + 1], skip_y=not save) if f8t else {}
            o.conv_fwd(self.cv_up[u], cur, W(name), self.ups[u], bias=Bz(name), act=G_SLOPE, **kw)
            cur = self.ups[u]
        kw = dict(xq=self._tq[self.nup], wq=self._wq["conv3.0"]) if f8t else {}
        o.conv_fwd(self.cv_c30, cur, W("conv3.0"), self.c30, bias=Bz("conv3.0"), act=G_SLOPE, **kw)
        o.conv_fwd(self.cv_c32, self.c30, W("conv3.2"), self.fake, bias=Bz("conv3.2"))
        return self.fake

    # ---- backward (after forward(save=True)) ---------------------------------------------------------
    def backward(self, x, dfake):
        """Parameter gradients of the generator for d loss / d fake = dfake (autograd backward of
        wasserstein.py:80 through generator.py:83-90)."""
        o, P, F_, B, S = self.ops, self.P, self.F, self.B, self.S
        if self._bwd is None:
            hs = S << self.nup
            self._bwd = dict(
                d_c30=o.zeros(B, hs, hs, F_), d_ups=[o.zeros(B, S << (u + 1), S << (u + 1), F_) for u in range(self.nup)],
                d_trunk=o.zeros(B, S, S, F_), gy=[o.zeros(B, S, S, F_), o.zeros(B, S, S, F_)],
                gx=o.zeros(B, S, S, F_), us=[o.zeros(B, S, S, 5 * F_), o.zeros(B, S, S, 5 * F_)])
        bw = self._bwd
        f8b = self.f8_bwd
        if f8b and "usq" not in bw:
            bw["usq"] = [(o.zeros(B, S, S, 5 * F_, dtype=torch.uint8), o.zeros(B, S, S, 5 * F_ // 32, dtype=torch.uint8)) for _ in range(2)]
        usq = lambda di, c0, c1: (bw["usq"][di][0][..., c0:c1], bw["usq"][di][1][..., c0 // 32:c1 // 32])
        # fp8 weight gradients: the forward wrote valid uniform-scale slab copies -> this pass writes the adjoint slabs' (block d's with
        # block d's exponents: the two byte buffers rotate, the exponent sets do not) and runs the fp8 kernel
        u_live = self.f8_wg and self._u_live
        uu = lambda d, c0, c1: (self._us_u[d & 1][..., c0:c1], self._eu[d][c0 // 32:c1 // 32])
        W = lambda n: P.w(n + ".weight")
        WD = lambda n: P.wd(n + ".weight")
        GW = lambda n: P.grad(n + ".weight").reshape(-1)
        GB = lambda n: P.grad(n + ".bias")
        # conv3.2 / conv3.0
        if self.cv_c32_bwd is not None:
            if self._dw_c32_tmp is None:
                self._dw_c32_tmp = o.zeros(F_ * 9 * self.np_p, dtype=torch.float32)
            self._dw_c32_tmp.zero_()
            o.conv_wgrad(self.cv_c32_bwd, dfake, self.c30, self._dw_c32_tmp)          # roles swapped: [ci][8 - t][co]
            o.wgrad_unswap(self._dw_c32_tmp, GW("conv3.2"), self.np_p, F_)
            o.colsum(dfake, GB("conv3.2"))
            o.conv_fwd(self.cv_c32_bwd, dfake, self._w_c32_bwd, bw["d_c30"], mask=self.c30, mask_slope=G_SLOPE)
        else:
            o.conv_wgrad(self.cv_c32, self.c30, dfake, GW("conv3.2"), db=GB("conv3.2"))
            o.conv_dgrad(self.cv_c32, dfake, WD("conv3.2"), bw["d_c30"], mask=self.c30, mask_slope=G_SLOPE)
        top = self.ups[-1] if self.nup else self.trunk
        o.conv_wgrad(self.cv_c30, top, bw["d_c30"], GW("conv3.0"), db=GB("conv3.0"))
        dcur = bw["d_ups"][-1] if self.nup else bw["d_trunk"]
        if self.nup:
            o.conv_dgrad(self.cv_c30, bw["d_c30"], WD("conv3.0"), dcur, mask=self.ups[-1], mask_slope=G_SLOPE)
        else:
            o.conv_dgrad(self.cv_c30, bw["d_c30"], WD("conv3.0"), dcur)
        # up-sampling convs (activation before the shuffle: the mask commutes with the permutation)
        for u in range(self.nup - 1, -1, -1):
            name = f"upsampling.{3 * u}"
            xin = self.ups[u - 1] if u > 0 else self.trunk
            o.conv_wgrad(self.cv_up[u], xin, dcur, GW(name)); o.colsum_ps(dcur, GB(name))
            if u > 0:
                o.conv_dgrad(self.cv_up[u], dcur, WD(name), bw["d_ups"][u - 1], mask=self.ups[u - 1], mask_slope=G_SLOPE)
                dcur = bw["d_ups"][u - 1]
            else:
                o.conv_dgrad(self.cv_up[u], dcur, WD(name), bw["d_trunk"])
                dcur = bw["d_trunk"]
        d_trunk = dcur                                   # = d out1 (skip) = d out2
        # conv2
        o.conv_wgrad(self.cv_conv2, self._saved[self.ndrb][..., :F_], d_trunk, GW("conv2"), db=GB("conv2"))
        gy = bw["gy"][0]
        o.conv_dgrad(self.cv_conv2, d_trunk, WD("conv2"), gy)
        gyi = 0
        for i in range(self.nrb - 1, -1, -1):
            # RRDB i: y = 0.2*o_{3i+2} + x_rrdb, so d o_{3i+2} = 0.2 * d y.  What travels from block to block is u5 = 0.2 * d o (the
            # adjoint of conv 5's output, o = 0.2*b5 + x), written by the PREVIOUS block's last data gradient straight into this
            # block's adjoint slab: no separate scaling pass per block.
            o.axpby(bw["us"][(3 * i + 2) & 1][..., 4 * F_:], gy, RES_SCALE * RES_SCALE)
            if f8b:        # (the one adjoint slice of an RRDB that does not come out of a conv epilogue)
                o.quant_mxfp8(bw["us"][(3 * i + 2) & 1][..., 4 * F_:], *usq((3 * i + 2) & 1, 4 * F_, 5 * F_))
                if u_live:
                    o.quant_uniform(bw["us"][(3 * i + 2) & 1][..., 4 * F_:], *uu(3 * i + 2, 4 * F_, 5 * F_))
            for j in (2, 1, 0):
                d = 3 * i + j
                # adjoint slab: channels [(k-1)F, kF) = u_k, the adjoint of conv k's output (k = 1..5)
                slab, us = self._saved[d], bw["us"][d & 1]
                pre = f"res_blocks.{i}.dense_blocks.{j}.b"
                for k in range(4, 0, -1):
                    # u_k = LeakyReLU'(b_k) * sum_{m > k} W_m[:, slice k]^T (*) u_m: one data gradient over u_{k+1..5}
                    uk = us[..., (k - 1) * F_:k * F_]
                    kw = dict(xq=usq(d & 1, k * F_, 5 * F_), wq=self._vq[d][k], out_q=usq(d & 1, (k - 1) * F_, k * F_)) if f8b else {}
                    if u_live:
                        kw["out_u"] = uu(d, (k - 1) * F_, k * F_)
                    o.conv_dgrad(self.cv_v[k], us[..., k * F_:], self.vpack(d, k), uk, mask=slab[..., k * F_:(k + 1) * F_], mask_slope=G_SLOPE, **kw)
                # the block's five weight / bias gradients, wgrad(slab[:kF], u_k): one launch over the 15 (u tile, slab tile) pairs
                if u_live:
                    o.conv_wgrad_dense_f8(self.cv_b, self._slab_u[d], self._ex[d], self._us_u[d & 1], self._eu[d], [GW(f"{pre}{k}.0") for k in range(1, 6)])
                    o.colsum_multi(us, [GB(f"{pre}{k}.0") for k in range(1, 6)])
                else:
                    o.conv_wgrad_dense(self.cv_b, slab, us, [GW(f"{pre}{k}.0") for k in range(1, 6)], [GB(f"{pre}{k}.0") for k in range(1, 6)])
                if self.f8_wg:     # the adjoint slab is complete: exponents for the next generator iteration's copies of it
                    o.block_exp_max(bw["usq"][d & 1][1], self._eu_new[d])
                # d x_drb = sum_m W_m[:, slice 0]^T (*) u_m + d o (identity path) = d o of the previous dense block; stored as that
                # block's u5 = 0.2 * (sum + d o) = 0.2 * sum + this block's u5
                nxt = bw["us"][(d - 1) & 1][..., 4 * F_:] if j > 0 else bw["gx"]
                kw = {}
                if f8b:
                    kw = dict(xq=usq(d & 1, 0, 5 * F_), wq=self._vq[d][0])
                    if j > 0:
                        kw["out_q"] = usq((d - 1) & 1, 4 * F_, 5 * F_)
                        if u_live:
                            kw["out_u"] = uu(d - 1, 4 * F_, 5 * F_)
                o.conv_dgrad(self.cv_v[0], us, self.vpack(d, 0), nxt, r1=us[..., 4 * F_:], s1=RES_SCALE, **kw)
            gyn = bw["gy"][gyi ^ 1]
            o.axpby(gyn, bw["gx"], 1.0 / RES_SCALE, gy, 1.0)   # d x_rrdb = d x_drb(3i) + d y (identity path); gx holds 0.2 * d x_drb
            gy, gyi = gyn, gyi ^ 1
        if self.f8_wg and self._ex_new is not None:      # every copy written with the old exponents has been consumed
            self._ex.copy_(self._ex_new); self._eu.copy_(self._eu_new)
            self._u_ok = True
            self._u_live = False
        # conv1: d out1 = trunk-path gradient + skip gradient
        o.axpby(gy, gy, 1.0, d_trunk, 1.0)
        o.conv_wgrad(self.cv_conv1, x, gy, GW("conv1"), db=GB("conv1"))


# =============================================================================================== train step
class TrainEngine:
    """One WGAN-GP train step = WassersteinGAN._critic_train_iteration (+ _generator_train_iteration
    every ``critic_iterations`` steps), reference DoWnGAN/GAN/wasserstein.py:27-83,131-147.

    ``dist`` (optional) is an object with ``world_size``, ``allreduce_sum_(tensor)`` and
    ``reduce_scalars(dict)``: every rank holds B samples of the global batch B*world; per-sample terms
    are normalised by the global batch so the flat gradient buffers are simply summed (plain data
    parallelism).
    """

    SCALARS = ("c_real_mean", "c_fake_mean", "gp_ret", "g_c_fake_mean", "l1_sum")

    def __init__(self, ops, coarse_side, filters, channels, batch, hp: HyperParams = None, n_predictands=2,
                 num_res_blocks=16, num_upsample=3, dist=None, stacked=None, check_finite=False):
        self.ops, self.hp = ops, hp or HyperParams()
        # debug mode: after every iteration ONE fused census of NaN / Inf over the scalars, the flat gradient buffer and the
        # generated batch; raises naming the first offending buffer.  The reference switches torch's anomaly detection on globally
        # (wasserstein.py:13: a NaN check behind every autograd op, 1.7x slower on the CPU); off by default here.
        self.check_finite = bool(check_finite)
        self.B, self.S = batch, coarse_side
        self.dist = dist
        self.world = dist.world_size if dist is not None else 1
        fine = coarse_side << num_upsample
        self.G = NativeGenerator(ops, filters, channels, batch, coarse_side, n_predictands, num_res_blocks, num_upsample)
        # compact 2-channel fields (below) and, on top of them, the three critic passes of an iteration stacked into one batch
        compact2 = n_predictands <= 2 and fine % 32 == 0 and not any(
            os.environ.get(k) is not None for k in ("DG_NO_COMPACT2", "DG_WG_NOIM2COL", "DG_GG_NOIM2COL"))
        # OPT-IN (stacked=True / DG_STACKED=1): measured +0.4-0.6 % per step at cfg2 for +77 GiB of HBM, and the
        # data-parallel overlap of the generator's gradient exchange with the real-batch pass goes away
        if stacked is None:                 # DG_STACKED=1: every layer; DG_STACKED=k > 1: layers k - 1 .. 7 only (NativeCritic.stack_from)
            env = os.environ.get("DG_STACKED")
            stacked = (int(env) if env.isdigit() and int(env) > 1 else True) if env is not None else False
        stacked = stacked if (stacked and compact2 and self._fits_stacked(ops, filters, batch, coarse_side, fine, num_res_blocks)) else False
        self.C = NativeCritic(ops, filters, fine, n_predictands, batch, stacked=stacked)
        assert self.C.c_pad[0] == self.G.np_p
        o = ops
        # Fields of the fine grid are stored np_p (16) channels wide like every activation.  With <= 2 predictands the critic's
        # first layer reads two real channels and its kernels are gather-bound on that layout (csrc/conv_small.hip:
        # gg_im2col_direct_kernel), so everything it reads also exists in COMPACT form [B, fine, fine, 2], written where the bytes
        # are produced anyway: the interpolate x-hat and the penalty's scaled gradient are compact only, and the interpolate pass
        # also leaves compact copies of its two inputs (the real and the generated batch) for their own critic passes.
        self.fine_shape = (batch, fine, fine, self.G.np_p)
        # (the weight-gradient kernel that takes the compact form needs rows of a multiple of 32 pixels)
        self.compact2 = compact2 and bool(self.C.convs[0].cin_real)
        self.stacked = self.C.stacked and self.compact2
        cshape = (batch, fine, fine, 2) if self.compact2 else self.fine_shape
        self.gbuf = o.zeros(*self.fine_shape)
        self.vbuf = o.zeros(*cshape)
        if self.compact2:      # one buffer [real | fake | x-hat]: the stacked critic pass reads it as a batch of 3B
            self.x3 = o.zeros(3 * batch, fine, fine, 2)
            self.real_c, self.fake_c, self.xhat = self.x3[:batch], self.x3[batch:2 * batch], self.x3[2 * batch:]
        else:
            self.x3 = self.real_c = self.fake_c = None
            self.xhat = o.zeros(*cshape)
        self.dfake = None
        self.ss = o.zeros(batch, dtype=torch.float32)
        self.coef = o.zeros(batch, dtype=torch.float32)
        self.scal = o.zeros(8, dtype=torch.float32)
        self.alpha_dev = o.zeros(batch, dtype=torch.float32)
        self.num_steps = 0
        self.n_real_elems = batch * n_predictands * fine * fine
        self.adam_hp = {}          # id(ParamStore) -> HyperParams whose lr / betas / eps differ from ``hp`` (trainer mirror)

    def _sc(self, name):
        i = self.SCALARS.index(name)
        return self.scal[i:i + 1]

    @staticmethod
    def _fits_stacked(ops, filters, batch, coarse_side, fine, nrb):
        """Is there HBM for the stacked critic passes (3B samples of activations + adjoints) beside everything else?  A rough
        upper estimate of the whole engine's footprint against the device (cfg2: ~225 of 288 GB)."""
        if not torch.cuda.is_available() or getattr(ops, "device", None) is None or torch.device(ops.device).type != "cuda":
            return True                                    # CPU emulation (tests): no budget to respect
        es = 4 if ops.tdtype == torch.float32 else 2
        cd, h, per_c = filters, fine, 0
        for l, st in enumerate(NativeCritic.STRIDES):
            h //= st
            per_c += (cd << (l // 2)) * h * h              # channels cd, cd, 2cd, 2cd, 4cd, ... at the layer's output size
        critic = 2 * 3 * batch * per_c * es * 1.07 + 2 * batch * cd * fine * fine * es + 24 * (cd * 8 * (fine // 16) ** 2 * 112 + 5e7 * (cd / 128.0) ** 2)
        S, F_ = coarse_side, filters
        gen = batch * es * ((3 * nrb + 1 + 4 + 2) * 5 * F_ * S * S + 6 * F_ * S * S + 5 * F_ * fine * fine + 2 * F_ * (fine * fine) // 3) + 24 * 1.1e8 * (F_ / 128.0) ** 2
        total = torch.cuda.get_device_properties(ops.device).total_memory
        return critic + gen < 0.88 * total

    def _allreduce_and_step(self, P, defer=False):
        """Sum the flat gradient buffer over the ranks, then Adam.  Local gradients are already normalised by the GLOBAL
        batch, so ranks are summed, not averaged.  With ``defer`` the bucketed all-reduces are only ENQUEUED (RCCL's own
        stream) and the wait + Adam run when the parameters are next touched: after a critic update that is after the
        following generator forward (next step's ``G(x)`` or this step's generator iteration), which hides the exchange."""
        if self.dist is not None and self.world > 1:
            works = self.dist.allreduce_sum_begin(P.g)

            def finish():
                self.dist.allreduce_finish(works)
                P.adam_step(self.adam_hp.get(id(P), self.hp), 1.0)
            if defer:
                P.defer(finish)
            else:
                finish()
        else:
            P.adam_step(self.adam_hp.get(id(P), self.hp), 1.0)

    def _assert_finite(self, where, P, extra=()):
        """check_finite mode: raise FloatingPointError naming the first buffer of this iteration that holds a NaN / Inf."""
        if not self.check_finite:
            return
        bufs = [("loss scalars", self.scal), (f"{where} gradients (flat buffer)", P.g), ("generated batch G(coarse)", self.G.fake)] + list(extra)
        counts = self.ops.count_nonfinite(bufs)
        bad = [(n, c) for n, c in counts.items() if c]
        # data parallel: the ranks decide TOGETHER (one small MAX all-reduce) -- a NaN seen by one rank only (its own samples) must
        # not leave the others blocked in the gradient all-reduce behind a rank that has already raised
        if self.dist is not None and self.world > 1:
            any_bad = self.dist.any_rank(bool(bad))
            if any_bad and not bad:
                raise FloatingPointError(f"check_finite: step {self.num_steps}, {where} iteration: non-finite values on another rank "
                                         f"(this rank, {getattr(self.dist, 'rank', '?')}, is clean)")
        if bad:
            detail = ""
            if bad[0][0].endswith("(flat buffer)"):      # name the parameters (host side, only on failure)
                g = P.g.detach().float().cpu()
                names = [k for k, (off, n, _) in P.entries.items() if not bool(torch.isfinite(g[off:off + n]).all())]
                detail = f"; parameters: {names[:6]}{' ...' if len(names) > 6 else ''}"
            raise FloatingPointError(f"check_finite: step {self.num_steps}, {where} iteration: {bad[0][1]} non-finite values in '{bad[0][0]}'"
                                     f"{detail} (all counts: {counts})")

    def critic_iteration(self, coarse, fine, alpha, apply_update=True, save_g=False):
        """wasserstein.py:27-55.  coarse/fine: native NHWC tensors; alpha: fp32 [B] on the device
        (replaces torch.rand at :91).  The generator runs forward-only: its backward in the reference's
        critic step is dead work (G grads are zeroed at :65 before any use).

        ``save_g``: keep the generator's dense-block slabs of this forward, so that a generator iteration that follows on
        the SAME batch with the SAME generator parameters (wasserstein.py:134-137: every ``critic_iterations``-th step) can
        skip its own, bit-identical ``G(coarse)`` (``generator_iteration(reuse_fake=True)``).
        While the generator's own update is still in flight (data parallel: its gradient all-reduce, enqueued at the end of
        the previous generator iteration), the real-sample pass -- which needs no generator -- runs first and hides it."""
        o, hp, C, B = self.ops, self.hp, self.C, self.B
        bg = B * self.world
        if self.stacked:
            return self._critic_iteration_stacked(coarse, fine, alpha, apply_update, save_g)
        real_first = self.G.P._pending is not None
        xr = fine
        if not real_first:
            fake = self.G.forward(coarse, save=save_g)            # :35
            if self.compact2:                                     # :94, hoisted: it also writes the compact real / fake batches
                o.gp_interp(fine, fake, alpha, self.xhat, self.real_c, self.fake_c)
                xr = self.real_c
        s0, s1, s2 = (0, 1, 2) if C.fc1_fused else (None, None, None)
        C.P.zero_grad(skip="classifier.0.weight" if C.fc1_fused else None)    # :43 (fc1_flush WRITES that gradient)
        out = C.forward(xr, s0, for_wgrad=True)                   # :37
        o.sum_strided(out, B, out.stride(0), 1.0 / B, self._sc("c_real_mean"))
        C.backward(xr, -1.0 / bg, fc1_slot=s0)                    # d(-mean c_real)
        if real_first:
            fake = self.G.forward(coarse, save=save_g)            # :35 (first use of G's parameters completes their update)
        if real_first or not self.compact2:
            o.gp_interp(fine, fake, alpha, self.xhat, None, self.fake_c)      # :94
        xk = self.fake_c if self.compact2 else fake
        out = C.forward(xk, s1, for_wgrad=True)                   # :38
        o.sum_strided(out, B, out.stride(0), 1.0 / B, self._sc("c_fake_mean"))
        C.backward(xk, 1.0 / bg, fc1_slot=s1)                     # d(+mean c_fake)
        C.gp_pass(self.xhat, self.gbuf, self.vbuf, self.ss, self.coef, self._sc("gp_ret"), hp, bg, fc1_slot=s2)   # :40,:87-117
        if C.fc1_fused:
            C.fc1_flush()
        self._assert_finite("critic", C.P, [("dC/dx-hat", self.gbuf)])
        if apply_update:
            self._allreduce_and_step(C.P, defer=True)             # :52-55 (overlaps with the next generator forward)

    def _critic_iteration_stacked(self, coarse, fine, alpha, apply_update, save_g):
        """The same iteration with the real, generated and interpolated batch going through the critic as ONE batch of 3B
        (NativeCritic.forward_stacked / backward_stacked): wasserstein.py:37, :38 and :97 are one forward; the backward of :52
        through the first two and the autograd.grad of :100-106 through the third are one adjoint chain with per-pass d out."""
        fake = self.G.forward(coarse, save=save_g)                                    # :35
        self.ops.gp_interp(fine, fake, alpha, self.xhat, self.real_c, self.fake_c)    # :94 (+ the compact real / fake batches)
        self._critic_passes_stacked(apply_update)

    def _critic_passes_stacked(self, apply_update):
        o, hp, C, B = self.ops, self.hp, self.C, self.B
        bg = B * self.world
        C.P.zero_grad(skip="classifier.0.weight")                                     # :43 (fc1_flush WRITES that gradient)
        out = C.forward_stacked(self.x3)                                              # :37, :38, :97
        o.sum_strided(out[:B], B, out.stride(0), 1.0 / B, self._sc("c_real_mean"))
        o.sum_strided(out[B:2 * B], B, out.stride(0), 1.0 / B, self._sc("c_fake_mean"))
        C.backward_stacked(self.x3, (-1.0 / bg, 1.0 / bg, 1.0), self.gbuf)            # d(-mean c_real + mean c_fake), dC/dx-hat
        C.gp_tangent(self.gbuf, self.vbuf, self.ss, self.coef, self._sc("gp_ret"), hp, bg, 2 * B, 2)   # :40, :110-117
        C.fc1_flush()
        self._assert_finite("critic", C.P, [("dC/dx-hat", self.gbuf)])
        if apply_update:
            self._allreduce_and_step(C.P, defer=True)                                 # :52-55

    def generator_iteration(self, coarse, fine, apply_update=True, reuse_fake=False):
        """wasserstein.py:58-83: g_loss = -mean(C(G(x)))*gamma + content_lambda*L1(G(x), y).
        ``reuse_fake``: the preceding ``critic_iteration(save_g=True)`` already ran G(coarse) on this batch with these
        parameters (only the critic was updated in between): its output and saved slabs are used as they are."""
        o, hp, C, G, B = self.ops, self.hp, self.C, self.G, self.B
        bg = B * self.world
        if self.dfake is None:
            self.dfake = o.zeros(*self.G.fake.shape)
        G.P.zero_grad()                                           # :65
        fake = G.fake if reuse_fake else G.forward(coarse, save=True)   # :67
        out = C.forward(self.fake_c if reuse_fake and self.compact2 else fake)    # :68 (the critic iteration's compact copy)
        o.sum_strided(out, B, out.stride(0), 1.0 / B, self._sc("g_c_fake_mean"))
        C.backward(fake, -hp.gamma / bg, wgrad=False, dx=self.gbuf)            # d(-gamma*mean c_fake)/d fake
        self._sc("l1_sum").zero_()
        o.l1(fake, fine, self._sc("l1_sum"), grad=self.dfake, grad_scale=hp.content_lambda / (self.n_real_elems * self.world),
             addend=self.gbuf)                                    # :78 + losses.py:51-53
        G.backward(coarse, self.dfake)                            # :80
        self._assert_finite("generator", G.P, [("d loss / d fake", self.dfake)])
        if apply_update:
            self._allreduce_and_step(G.P, defer=True)             # :83 (overlaps with the next critic iteration's real pass)

    def metrics_pass(self, coarse, fine, n_valid=None):
        """Per-batch evaluation metrics of the reference's training loop (mlflow_tools/mlflow_epoch.py:53-63 called at
        wasserstein.py:140): MAE = L1(real, G(x)) (losses.py:40-55), MSE (losses.py:58-70), Wass = mean C(real) -
        mean C(G(x)) (losses.py:8-9), MSSSIM = MS-SSIM of the batch-min-max-normalised fields (losses.py:12-38; msssim.py).
        MSSSIM is None when the tile is too small for 5 scales (pytorch_msssim asserts side > 96).
        ``n_valid`` < B: only the first n_valid samples of the (padded) batch count -- a smaller test batch evaluated on the
        training engine's buffers instead of re-binding the whole engine to its size; every sample is independent on this path
        (no batch norm), so the padded rows change nothing in the first n_valid."""
        o, C, B = self.ops, self.C, self.B
        n = B if n_valid is None else int(n_valid)
        assert 1 <= n <= B
        fake = self.G.forward(coarse, save=False)
        m = self.scal[5:8]
        m.zero_()
        o.l1(fine[:n], fake[:n], m[0:1])
        o.sqdiff(fine[:n], fake[:n], m[1:2])
        out = C.forward(fine)
        o.sum_strided(out, n, out.stride(0), 1.0 / n, self._sc("c_real_mean"))
        out = C.forward(fake)
        o.sum_strided(out, n, out.stride(0), 1.0 / n, self._sc("c_fake_mean"))
        s = self.scal.detach().cpu().tolist()
        d = {"l1_sum": s[5], "sq_sum": s[6], "c_real_mean": s[0], "c_fake_mean": s[1]}
        if self.dist is not None and self.world > 1:
            d = self.dist.reduce_scalars(d, mean=("c_real_mean", "c_fake_mean"), total=("l1_sum", "sq_sum"))
        cnt = (self.n_real_elems // B) * n * self.world
        msssim = None
        H, W = fine.shape[1], fine.shape[2]
        if min(H, W) > 96:
            if self._msssim is None:
                self._msssim = {}
            if n not in self._msssim:
                from .msssim import MsSsim
                self._msssim[n] = MsSsim(o, n, H, W, c_real=self.G.npred)
            msssim = self._msssim[n](fine[:n], fake[:n], dist=self.dist, world=self.world)
        return {"MAE": d["l1_sum"] / cnt, "MSE": d["sq_sum"] / cnt, "Wass": d["c_real_mean"] - d["c_fake_mean"], "MSSSIM": msssim}

    _msssim = None

    # ---- HIP graphs: the ~500 launches of an iteration are captured once and replayed (launch-bound small tiles)
    def enable_graphs(self, coarse, fine):
        """Capture the critic and generator iterations (everything up to, not including, the all-reduce + Adam, whose
        bias correction depends on the host step count) into two HIP graphs.  ``coarse`` / ``fine`` give the shapes;
        later steps copy their inputs into the captured static buffers.  The reference's small native problem size
        (2ch 16x16 -> 128x128, hyperparams.py:18 / config.py:112) is launch-bound without this."""
        o = self.ops
        assert o.prof is None, "kernel timing hooks record events; disable them before capturing"
        assert not self.check_finite, "check_finite reads a counter back every iteration: not capturable"
        self._g_coarse, self._g_fine = coarse.clone(), fine.clone()
        side = torch.cuda.Stream(device=o.device)
        side.wait_stream(torch.cuda.current_stream(o.device))
        with torch.cuda.stream(side):                       # eager warm-up: lazy workspaces, function attributes
            self.critic_iteration(self._g_coarse, self._g_fine, self.alpha_dev, apply_update=False, save_g=True)
            self.generator_iteration(self._g_coarse, self._g_fine, apply_update=False, reuse_fake=True)
        torch.cuda.current_stream(o.device).wait_stream(side)
        torch.cuda.synchronize(o.device)
        self._graph_c, self._graph_cs, self._graph_g = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph_c):
            self.critic_iteration(self._g_coarse, self._g_fine, self.alpha_dev, apply_update=False)
        with torch.cuda.graph(self._graph_cs):                 # generator steps: the critic iteration keeps G's slabs ...
            self.critic_iteration(self._g_coarse, self._g_fine, self.alpha_dev, apply_update=False, save_g=True)
        with torch.cuda.graph(self._graph_g):                  # ... and the generator iteration starts from them
            self.generator_iteration(self._g_coarse, self._g_fine, apply_update=False, reuse_fake=True)
        self.C.P.zero_grad(); self.G.P.zero_grad()
        self.graphs = True

    graphs = False

    def train_step(self, coarse, fine, alpha):
        """loop body of _train_epoch, wasserstein.py:131-147 (metrics pass :140 excluded)."""
        ran_g = self.num_steps % self.hp.critic_iterations == 0   # :136
        if self.graphs:
            if coarse is not self._g_coarse:
                self._g_coarse.copy_(coarse)
            if fine is not self._g_fine:
                self._g_fine.copy_(fine)
            self.alpha_dev.copy_(alpha)
            (self._graph_cs if ran_g else self._graph_c).replay()
            self._allreduce_and_step(self.C.P)
            if ran_g:
                self._graph_g.replay()
                self._allreduce_and_step(self.G.P)
        else:
            # generator steps: ONE G(coarse) serves both iterations (same batch, same generator parameters)
            self.critic_iteration(coarse, fine, alpha, save_g=ran_g)
            if ran_g:
                self.generator_iteration(coarse, fine, reuse_fake=True)
        self.num_steps += 1
        return ran_g

    def read_scalars(self, ran_g=False):
        """Synchronises and returns the scalars the reference computes and drops (:46-50, :74-78)."""
        s = self.scal.detach().cpu().tolist()
        hp = self.hp
        d = dict(zip(self.SCALARS, s))
        if self.dist is not None and self.world > 1:
            d = self.dist.reduce_scalars(d, mean=("c_real_mean", "c_fake_mean", "g_c_fake_mean"), total=("gp_ret", "l1_sum"))
        out = {"c_real_mean": d["c_real_mean"], "c_fake_mean": d["c_fake_mean"], "gp_ret": d["gp_ret"]}
        out["gradient_penalty"] = hp.gp_lambda * d["gp_ret"]
        out["critic_loss"] = d["c_fake_mean"] - d["c_real_mean"] + out["gradient_penalty"]
        out["w_estimate"] = d["c_real_mean"] - d["c_fake_mean"]
        if ran_g:
            out["g_c_fake_mean"] = d["g_c_fake_mean"]
            out["content_loss"] = d["l1_sum"] / (self.n_real_elems * self.world)
            out["g_loss"] = -d["g_c_fake_mean"] * hp.gamma + hp.content_lambda * out["content_loss"]
        return out


class TrainEngineFS(TrainEngine):
    """Frequency-separation variant (DoWnGAN/GAN/wasserstein_fs.py:28-92, hyperparams.py:31-35):
    ``low(x)`` = 5x5 box mean with replicated borders, ``high = x - low``.  The critic (incl. the gradient penalty) sees the
    high-pass parts, the content loss compares the low-pass parts.  Everything else is the parent's hot path."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        o = self.ops
        shape = self.fine_shape
        self.real_high, self.fake_high = o.zeros(*shape), o.zeros(*shape)
        self.real_low = self.fake_low = self.tbuf = None

    def critic_iteration(self, coarse, fine, alpha, apply_update=True, save_g=False):
        """wasserstein_fs.py:28-60."""
        o, hp, C, B = self.ops, self.hp, self.C, self.B
        bg = B * self.world
        fake = self.G.forward(coarse, save=save_g)                # :36
        o.lowpass5(fake, high=self.fake_high)                     # :37,40
        o.lowpass5(fine, high=self.real_high)                     # :38,41
        o.gp_interp(self.real_high, self.fake_high, alpha, self.xhat, self.real_c, self.fake_c)   # :46 -> _gp(real_high, fake_high)
        if self.stacked:
            return self._critic_passes_stacked(apply_update)
        xr, xk = (self.real_c, self.fake_c) if self.compact2 else (self.real_high, self.fake_high)
        s0, s1, s2 = (0, 1, 2) if C.fc1_fused else (None, None, None)
        C.P.zero_grad(skip="classifier.0.weight" if C.fc1_fused else None)    # :49 (fc1_flush WRITES that gradient)
        out = C.forward(xr, s0, for_wgrad=True)                   # :43
        o.sum_strided(out, B, out.stride(0), 1.0 / B, self._sc("c_real_mean"))
        C.backward(xr, -1.0 / bg, fc1_slot=s0)
        out = C.forward(xk, s1, for_wgrad=True)                   # :44
        o.sum_strided(out, B, out.stride(0), 1.0 / B, self._sc("c_fake_mean"))
        C.backward(xk, 1.0 / bg, fc1_slot=s1)
        C.gp_pass(self.xhat, self.gbuf, self.vbuf, self.ss, self.coef, self._sc("gp_ret"), hp, bg, fc1_slot=s2)
        if C.fc1_fused:
            C.fc1_flush()
        self._assert_finite("critic", C.P, [("dC/dx-hat", self.gbuf)])
        if apply_update:
            self._allreduce_and_step(C.P, defer=True)             # :57-60

    def generator_iteration(self, coarse, fine, apply_update=True, reuse_fake=False):
        """wasserstein_fs.py:63-92: g_loss = -gamma*mean C(fake_high) + content_lambda*L1(fake_low, real_low); the gradient
        reaches ``fake`` through both branches: d fake = d_high + low^T(g_L1 - d_high)."""
        o, hp, C, G, B = self.ops, self.hp, self.C, self.G, self.B
        bg = B * self.world
        if self.dfake is None:
            self.dfake = o.zeros(*self.G.fake.shape)
        if self.tbuf is None:
            self.real_low, self.fake_low, self.tbuf = (o.zeros(*self.fine_shape) for _ in range(3))
        G.P.zero_grad()                                           # :70
        fake = G.fake if reuse_fake else G.forward(coarse, save=True)   # :72
        o.lowpass5(fake, low=self.fake_low, high=self.fake_high)  # :73,76
        o.lowpass5(fine, low=self.real_low)                       # :74
        out = C.forward(self.fake_high)                           # :79
        o.sum_strided(out, B, out.stride(0), 1.0 / B, self._sc("g_c_fake_mean"))
        C.backward(self.fake_high, -hp.gamma / bg, wgrad=False, dx=self.gbuf)  # d(-gamma*mean c_fake)/d fake_high
        self._sc("l1_sum").zero_()
        o.l1(self.fake_low, self.real_low, self._sc("l1_sum"), grad=self.tbuf,
             grad_scale=hp.content_lambda / (self.n_real_elems * self.world))  # :86
        o.axpby(self.tbuf, self.tbuf, 1.0, self.gbuf, -1.0)        # g_L1 - d_high
        o.lowpass5_adjoint(self.tbuf, self.dfake)
        o.axpby(self.dfake, self.dfake, 1.0, self.gbuf, 1.0)       # + d_high
        G.backward(coarse, self.dfake)                            # :88
        self._assert_finite("generator", G.P, [("d loss / d fake", self.dfake)])
        if apply_update:
            self._allreduce_and_step(G.P, defer=True)             # :91
