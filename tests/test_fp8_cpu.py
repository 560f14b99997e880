"""MXFP8 emulation (oracle/emu_ops.py::mx_quant -- the CPU statement of csrc/quant.hip's format): scale rule, block
layout, saturation and error bound; and the engine's fp8 mode on emulated ops (host plumbing: which layers quantise)."""
import torch

from oracle.emu_ops import EmuOps


def test_mx_quant_format():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 256, generator=g) * torch.logspace(-5, 2, 256)
    x[0, :128] = 0.0                                     # an all-zero group
    x[1, 3] = 1e4                                        # a dominant element
    q, s, d = EmuOps.mx_quant(x)
    assert q.shape == (7, 256) and s.shape == (7, 8) and d.shape == x.shape
    for r in range(7):
        for grp in range(2):
            for gg in range(4):
                idx = [grp * 128 + 32 * gg + e for e in range(32)]                  # 32 consecutive channels
                blk = x[r, idx]
                amax = float(blk.abs().max())
                e = int(s[r, grp * 4 + gg])
                if amax == 0:
                    assert e == 0 and float(d[r, idx].abs().max()) == 0
                    continue
                import math
                assert e == max(0, math.floor(math.log2(amax)) - 8 + 127)          # OCP MX rule: 2^(floor(log2 amax) - emax)
                scale = 2.0 ** (e - 127)
                assert float((blk / scale).abs().max()) < 512                        # fits E4M3 after saturation at 448
                # E4M3 has 3 mantissa bits: relative error <= 2^-4 for normal values, absolute <= scale * 2^-10 below them
                err = (d[r, idx] - blk).abs()
                assert bool((err <= blk.abs() * 2.0 ** -4 + scale * 2.0 ** -10 + (blk.abs() > 448 * scale) * blk.abs()).all())
    assert float(d[1, 3]) == 448 * 2.0 ** (int(s[1, 0]) - 127) or abs(float(d[1, 3]) - 1e4) <= 1e4 * 2.0 ** -4


def test_mx_quant_poisons_nonfinite_blocks():
    x = torch.randn(4, 128)
    x[1, 40] = float("nan"); x[2, 100] = float("inf")
    q, s, deq = EmuOps.mx_quant(x)
    assert bool((q[1, 32:64] == 0x7F).all()) and bool((q[2, 96:128] == 0x7F).all()) and int((q == 0x7F).sum()) == 64
    assert bool(torch.isnan(deq[1, 32:64]).all()) and bool(torch.isfinite(deq[0]).all()) and bool(torch.isfinite(deq[1, :32]).all())


def test_engine_fp8_mode_only_touches_wide_critic_convs():
    """F = 128 critic on a 64x64 tile, emulated: fp8 mode changes the critic's scalars a little, and not at all when no
    layer is eligible (the generator never is)."""
    from downgan_amd import synthetic
    from downgan_amd.engine import HyperParams, TrainEngine
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(4)
    B, S, F_, cin, nrb = 1, 8, 128, 2, 1
    out = {}
    for f8 in (False, True):
        ops = EmuOps("f32", f8_critic=f8)
        eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
        eng.G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb))
        eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
        coarse, fine = synthetic.tiles(B, cin, S)
        xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.float32)
        xf = nchw_to_nhwc_padded(torch.from_numpy(fine), 16, torch.float32)
        eng.critic_iteration(xc, xf, torch.from_numpy(synthetic.alpha(B, 0)), apply_update=False)
        out[f8] = (eng.read_scalars(), eng.G.fake.clone())
    assert torch.equal(out[False][1], out[True][1])                                  # generator untouched
    a, b = out[False][0], out[True][0]
    assert a["c_real_mean"] != b["c_real_mean"]                                      # the critic did go through fp8 ...
    assert abs(a["c_real_mean"] - b["c_real_mean"]) < 0.05 * max(abs(a["c_real_mean"]), 0.05)   # ... and stayed close
    assert abs(a["gp_ret"] - b["gp_ret"]) < 1e-2 * a["gp_ret"]


def test_generator_fp8_trunk_on_emulated_ops():
    """f8_generator: the dense-block trunk's forward reads / writes the slabs' fp8 forms (slices of a [B,S,S,5F] byte tensor with
    strided scale rows); result close to the plain forward, first conv and tail untouched."""
    from downgan_amd import synthetic
    from downgan_amd.engine import NativeGenerator
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(4)
    B, S, F_, cin, nrb = 1, 8, 128, 2, 1
    coarse, _ = synthetic.tiles(B, cin, S)
    xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.float32)
    outs = {}
    for f8 in (False, True):
        G = NativeGenerator(EmuOps("f32", f8_generator=f8), F_, cin, B, S, num_res_blocks=nrb)
        assert G.f8 == f8
        G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb))
        outs[f8] = (G.forward(xc, save=f8).clone(), G.out1.clone(), G.trunk.clone())
    assert torch.equal(outs[False][1], outs[True][1])                         # conv1 is not an fp8 layer
    d = (outs[True][2] - outs[False][2]).norm() / outs[False][2].norm()
    assert 0 < float(d) < 0.1, float(d)                                        # trunk went through fp8 and stayed close
    assert float((outs[True][0] - outs[False][0]).norm() / outs[False][0].norm()) < 0.1


def test_generator_fp8_tail_forward_on_emulated_ops():
    """f8_gtail: the up-sampling convs and conv3.0 (generator.py:69-81,88-89) read MXFP8 forms written by their producers -- conv2's
    epilogue and the up-sampling convs' own epilogues, in shuffled pixel order -- ; a forward that is not saved leaves the bf16
    up-sampled tensors unwritten; the result stays close to the bf16 tail's."""
    from downgan_amd import synthetic
    from downgan_amd.engine import NativeGenerator
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(4)
    B, S, F_, cin, nrb, nup = 1, 8, 128, 2, 1, 2
    coarse, _ = synthetic.tiles(B, cin, S)
    xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.float32)
    outs = {}
    for tail in (False, True):
        ops = EmuOps("f32", f8_generator=True)
        ops.f8_gtail = tail
        seen = []
        real = ops.conv_fwd
        ops.conv_fwd = lambda cv, x, w, y, _r=real, _s=seen, **kw: (_s.append((cv.net, cv.pixel_shuffle, kw.get("xq") is not None, kw.get("out_q") is not None,
                                                                               bool(kw.get("skip_y")))), _r(cv, x, w, y, **kw))[1]
        G = NativeGenerator(ops, F_, cin, B, S, num_res_blocks=nrb, num_upsample=nup)
        assert G.f8 and G.f8_tail == tail
        G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb, num_upsample=nup))
        fake = G.forward(xc, save=False).clone()
        tcalls = [c for c in seen if c[0] == "T"]
        assert tcalls == ([("T", True, True, True, True)] * nup + [("T", False, True, False, False)] if tail else [("T", True, False, False, False)] * nup + [("T", False, False, False, False)])
        ups_unsaved = [float(t.abs().max()) for t in G.ups]
        fake_saved = G.forward(xc, save=True).clone()
        assert torch.equal(fake, fake_saved)                                       # the bf16 tensors are by-products only
        assert all(float(t.abs().max()) > 0 for t in G.ups)
        assert all(v == 0.0 for v in ups_unsaved) == tail                          # not stored by the unsaved forward in this mode
        if tail:                                                                   # the shuffled copy is the quantised stored tensor
            for u in range(nup):
                q, sc, _ = EmuOps.mx_quant(G.ups[u])
                assert torch.equal(G._tq[u + 1][0], q) and torch.equal(G._tq[u + 1][1], sc)
        outs[tail] = fake
    d = float((outs[True] - outs[False])[..., :2].norm() / outs[False][..., :2].norm())
    assert 0 < d < 0.08, d


def test_generator_fp8_data_gradients_on_emulated_ops():
    """f8_generator with f8_gbwd: the dense blocks' data gradients (autograd of generator.py:24-41) read the adjoint slab's MXFP8
    form -- slices written by the producing epilogues (activation mask + copy, residual + copy) and one quantiser call per RRDB
    -- and the quantised virtual packs; parameter gradients stay close to the ones of the bf16 data gradients, and the layers
    outside the trunk's backward chain (up-sampling tail, conv3.*) are bit-identical."""
    from downgan_amd import synthetic
    from downgan_amd.engine import NativeGenerator
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(4)
    B, S, F_, cin, nrb = 1, 8, 128, 2, 1
    coarse, _ = synthetic.tiles(B, cin, S)
    xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.float32)
    grads, calls = {}, {}
    for gb in (False, True):
        ops = EmuOps("f32", f8_generator=True)
        ops.f8_gbwd = gb
        seen = calls[gb] = []
        real = ops.conv_dgrad
        ops.conv_dgrad = lambda cv, dy, w, dx, _r=real, _s=seen, **kw: (_s.append((cv.net, cv.Cout, kw.get("xq") is not None, kw.get("out_q") is not None)),
                                                                         _r(cv, dy, w, dx, **kw))[1]
        G = NativeGenerator(ops, F_, cin, B, S, num_res_blocks=nrb, num_upsample=1)
        assert G.f8 and G.f8_bwd == gb
        G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb, num_upsample=1))
        fake = G.forward(xc, save=True)
        dfake = torch.randn(fake.shape, generator=torch.Generator().manual_seed(3)) * (torch.arange(fake.shape[-1]) < 2)
        G.P.zero_grad()
        G.backward(xc, dfake)
        grads[gb] = G.grad_dict()
    # 3 dense blocks x 5 virtual convs read producer-written forms; all but the RRDB's first block's last one write the next form
    trunk = [c for c in calls[True] if c[0] == "G" and c[2]]
    assert len(trunk) == 15 and sum(c[3] for c in trunk) == 14, trunk
    assert not any(c[2] or c[3] for c in calls[False])
    for name in ("conv3.2.weight", "conv3.0.weight", "upsampling.0.weight", "conv2.weight"):
        assert torch.equal(grads[False][name], grads[True][name]), name               # upstream of the dense blocks' backward
    worst = 0.0
    for name, ga in grads[False].items():
        gb_ = grads[True][name]
        if name.startswith(("res_blocks", "conv1")):
            d = float((ga - gb_).norm() / ga.norm())
            assert 0 < d < 0.12, (name, d)                                             # went through fp8 and stayed close
            worst = max(worst, d)
    assert worst > 1e-3


def test_generator_fp8_weight_gradients_after_the_first_iteration():
    """f8_gwgrad: the dense blocks' weight gradients (autograd of generator.py:24-41) from uniform-scale copies of the activation and
    adjoint slabs.  The first generator iteration has no exponents yet: bf16 kernel, gradients identical to the mode without; from
    the second on dg_conv3x3_wgrad_dense_f8 runs once per dense block, bias gradients come from column sums of the bf16 adjoint
    (identical), weight gradients stay close."""
    from downgan_amd import synthetic
    from downgan_amd.engine import NativeGenerator
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(4)
    B, S, F_, cin, nrb = 1, 64, 128, 2, 1           # rows of 64 pixels: the fp8 kernel's K step
    coarse, _ = synthetic.tiles(B, cin, S)
    xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.float32)
    res = {}
    for wg in (False, True):
        ops = EmuOps("f32", f8_generator=True)
        ops.f8_wgrad = True
        ops.f8_gwgrad = wg
        calls = []
        real = ops.conv_wgrad_dense_f8
        ops.conv_wgrad_dense_f8 = lambda *a, _r=real: (calls.append(1), _r(*a))[1]
        G = NativeGenerator(ops, F_, cin, B, S, num_res_blocks=nrb, num_upsample=0)
        assert G.f8_bwd and G.f8_wg == wg
        G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb, num_upsample=0))
        per_iter = []
        for it in range(2):
            fake = G.forward(xc, save=True)
            dfake = torch.randn(fake.shape, generator=torch.Generator().manual_seed(3 + it)) * (torch.arange(fake.shape[-1]) < 2)
            G.P.zero_grad()
            n0 = len(calls)
            G.backward(xc, dfake)
            per_iter.append((len(calls) - n0, G.grad_dict()))
        res[wg] = per_iter
    assert [n for n, _ in res[False]] == [0, 0] and [n for n, _ in res[True]] == [0, 3]
    for name, g0 in res[False][0][1].items():
        assert torch.equal(g0, res[True][0][1][name]), name                          # first iteration: no exponents yet
    worst = 0.0
    for name, ga in res[False][1][1].items():
        gb = res[True][1][1][name]
        if name.startswith("res_blocks") and name.endswith("weight"):
            d = float((ga - gb).norm() / ga.norm())
            assert 0 < d < 0.08, (name, d)
            worst = max(worst, d)
        else:                                                                        # biases (column sums of the bf16 adjoint), other layers
            assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-7), name
    assert worst > 1e-3


def test_uniform_scale_copy_and_block_exponents():
    """The operand format of the fp8 weight gradient (oracle/emu_ops.py::uq_quant, csrc/gg_common.h epi64_pixel f_u): E4M3 with one
    exponent per 32-channel block of the whole tensor; block_exp_max = the largest MXFP8 block exponent + margin."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 5, 7, 256, generator=g) * torch.logspace(-3, 1, 256)
    _, s, _ = EmuOps.mx_quant(x)
    exps = torch.zeros(8, dtype=torch.uint8)
    EmuOps("f32").block_exp_max(s, exps, margin=1)
    assert torch.equal(exps, (s.reshape(-1, 8).int().amax(0) + 1).to(torch.uint8))
    q, d = EmuOps.uq_quant(x, exps)
    scale = torch.ldexp(torch.ones(8), exps.int() - 127).repeat_interleave(32)
    assert float((x.abs() / scale).max()) < 256                          # margin 1: the largest element sits below 2^8
    err = (d - x).abs()
    assert bool((err <= x.abs() * 2.0 ** -4 + scale * 2.0 ** -10).all())   # E4M3: 3 mantissa bits, subnormals down to 2^-9
    assert torch.equal(EmuOps.uq_dequant(q, exps), d)
    x[1, 2, 3, 40] = float("inf")
    q2, _ = EmuOps.uq_quant(x, exps)
    assert bool((q2[1, 2, 3, 32:64] == 0x7F).all()) and int((q2 == 0x7F).sum()) >= 32


def test_engine_fp8_weight_gradients_after_the_first_pass():
    """fp8 mode, emulated: the eligible critic layers (128 -> 128 stride 2 at 128x128 -> 64x64, 128 -> 256 at 64x64: output rows of
    64 pixels) take their weight gradients from the uniform-scale fp8 copies once the exponents of a role exist -- the real pass of the first iteration still uses the bf16
    kernel, the fake pass and the penalty's tangent pass of the SECOND iteration use dg_conv3x3_wgrad_f8 -- and the gradients stay
    close to the ones of the fp8 mode with bf16 weight gradients."""
    from downgan_amd import synthetic
    from downgan_amd.engine import HyperParams, TrainEngine
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(4)
    B, S, F_, cin, nrb = 1, 16, 128, 2, 1
    res = {}
    for wg8 in (False, True):
        ops = EmuOps("f32", f8_critic=True)
        ops.f8_wgrad = wg8
        ops.f8_l0u = False            # (the first layer's uniform-scale-only output changes the FORWARD: its own test below)
        calls = []
        real = ops.conv_wgrad_f8
        ops.conv_wgrad_f8 = lambda cv, *a, _r=real: (calls.append((cv.Cin, cv.Cout, cv.H)), _r(cv, *a))[1]
        eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
        eng.G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb))
        eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
        assert eng.C.wg8 == ([False, True, True, False, False, False, False, False] if wg8 else [False] * 8)
        coarse, fine = synthetic.tiles(B, cin, S)
        xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.float32)
        xf = nchw_to_nhwc_padded(torch.from_numpy(fine), 16, torch.float32)
        per_iter = []
        for it in range(2):
            n0 = len(calls)
            eng.critic_iteration(xc, xf, torch.from_numpy(synthetic.alpha(B, it)), apply_update=False)
            per_iter.append(len(calls) - n0)
        res[wg8] = (eng.C.P.g.clone(), per_iter, eng.read_scalars())
    assert res[False][1] == [0, 0]
    # iteration 0: real pass bf16 (no exponents yet), fake pass fp8, tangent pass bf16 (its role has no exponents yet); iteration 1: all three
    assert res[True][1] == [2, 6], res[True][1]
    ga, gb = res[False][0], res[True][0]
    assert 0 < float((ga - gb).norm()) < 0.05 * float(ga.norm())
    for k in ("c_real_mean", "c_fake_mean", "gp_ret"):
        assert res[False][2][k] == res[True][2][k], k                       # the forward and the penalty's value do not depend on it


def test_first_layer_uniform_scale_output_alone():
    """f8_l0u: once exponents exist, the critic's first layer (critic.py:21-24, 2 -> 128 channels: an 8.6-GB store-bound launch at
    configs[1]) writes ONLY the uniform-scale copy of its output -- no MXFP8 copy, no bf16 tensor -- with the census of magnitudes the
    next exponents come from (dg_epilogue.out_amax); layer 1's MXFP8 conv reads that copy with the block exponents as the scale row
    of every pixel, its fp8 weight gradient reads it as before; the penalty's first tangent likewise.  Gradient quality against the
    fp32 engine equals the MXFP8 mode's (cosine per layer within 0.03), which also needs the exponents not to fall to the level of
    the smaller of two alternating inputs (real / generated samples: they fall by at most one per update)."""
    from downgan_amd import synthetic
    from downgan_amd.engine import HyperParams, TrainEngine
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(6)
    B, S, F_, cin, nrb = 1, 16, 128, 2, 1
    grads, seen = {}, {}
    for mode in ("f32", "mx", "l0u"):
        ops = EmuOps("f32", f8_critic=mode != "f32")
        ops.f8_l0u = mode == "l0u"
        log = seen[mode] = []
        real = ops.conv_fwd
        def spy(cv, x, w, y, _r=real, _l=log, **kw):
            xq = kw.get("xq")
            _l.append((cv.Cin if cv.net == "C" else -1, cv.stride, kw.get("out_q") is not None, kw.get("out_u") is not None, bool(kw.get("skip_y")), kw.get("out_amax") is not None,
                       xq is not None and xq[1].dim() == 1))
            return _r(cv, x, w, y, **kw)
        ops.conv_fwd = spy
        eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
        eng.G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb))
        eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
        assert eng.C.l0u == (mode == "l0u")
        coarse, fine = synthetic.tiles(B, cin, S)
        xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.float32)
        xf = nchw_to_nhwc_padded(torch.from_numpy(fine), 16, torch.float32)
        for it in range(2):
            eng.critic_iteration(xc, xf, torch.from_numpy(synthetic.alpha(B, it)), apply_update=False)
        grads[mode] = (eng.C.P.g.clone(), dict(eng.C.P.entries))
    first = [c for c in seen["l0u"] if c[0] == 16 and c[1] == 1]                # first-layer launches of the critic (forward and tangent)
    second = [c for c in seen["l0u"] if c[0] == 128 and c[1] == 2]              # layer 1 (128 -> 128, stride 2)
    assert first[0][2] and first[0][5] and not first[0][3]                      # bootstrap pass: MXFP8 copy + census
    alone = [c for c in first if c[3] and not c[2]]
    assert len(alone) >= 5 and all(c[4] and c[5] for c in alone)                # ... then the uniform-scale copy alone, bf16 tensor skipped
    assert sum(c[6] for c in second) == len(alone)                              # each of them read by layer 1 with the exponent row
    assert not any(c[6] for c in seen["mx"])
    g32, ent = grads["f32"]
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
    for name, (off, n, _) in ent.items():
        if name.startswith("features") and name.endswith("weight"):
            f = g32[off:off + n]
            c_mx, c_u = cos(grads["mx"][0][off:off + n], f), cos(grads["l0u"][0][off:off + n], f)
            assert c_u > c_mx - 0.03 and c_u > 0.8, (name, c_mx, c_u)
