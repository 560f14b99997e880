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
