"""MXFP8 conv path (BASELINE configs[4]) through the C ABI: dg_quant_mxfp8 bit-exact against the emulation, the fp8
forward / stride-2 forward / data-gradient convs of the critic's wide layers (critic.py:25-88) against fp32 convolutions of
the SAME dequantised operands (products of E4M3 values are exact in fp32, so only accumulation order and the bf16 rounding
of the result differ), and one critic + generator iteration of the engine in fp8 mode against the emulated engine."""
import pytest
import torch

from downgan_amd.ops import Conv, HipOps
from oracle.emu_ops import EmuOps

pytestmark = pytest.mark.gpu


def close(a, b, what, tol=1.6e-2):
    a, b = a.float().cpu(), b.float().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.3e}) at {(a - b).abs().argmax().item()}"


@pytest.mark.parametrize("src_dtype", [torch.bfloat16, torch.float32])
def test_quantiser_matches_emulation_bit_for_bit(src_dtype):
    hip = HipOps("bf16")
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(3, 9, 13, 384, generator=g) * torch.logspace(-7, 3, 384)).to(src_dtype)
    x[0, 0, 0, :128] = 0
    x[1, 2, 3, 200] = 3e4
    x[2, :, :, 256:] *= 1e-30                                   # deep underflow: scale byte clamps at 0
    q_ref, s_ref, _ = EmuOps.mx_quant(x)
    q, s = hip.quant_mxfp8(x.cuda())
    assert torch.equal(s.cpu(), s_ref), (s.cpu().int() - s_ref.int()).abs().max()
    same = q.cpu() == q_ref
    zeros = (q.cpu() & 0x7f) == 0                               # +0 / -0 are the same value
    assert bool((same | (zeros & ((q_ref & 0x7f) == 0))).all()), int((~same).sum())
    # a channel-slice view with a wider pixel stride
    slab = torch.randn(2, 4, 4, 640, generator=g).to(src_dtype).cuda()
    q2, s2 = hip.quant_mxfp8(slab[..., 128:384])
    q2r, s2r, _ = EmuOps.mx_quant(slab[..., 128:384].cpu())
    assert torch.equal(q2.cpu(), q2r) and torch.equal(s2.cpu(), s2r)


def test_quantiser_keeps_nan_and_inf_visible():
    """A NaN or an Inf must not be clamped into a finite fp8 value (every later fp8 layer reads only the quantised copy): the
    whole 32-channel block becomes E4M3 NaN -- stand-alone quantiser and fused conv epilogue alike, bit-equal to the emulation --
    and an fp8 conv fed such a block returns NaN for the pixels that read it."""
    hip = HipOps("bf16", f8_critic=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 8, 256, generator=g).to(torch.bfloat16)
    x[0, 1, 2, 37] = float("nan"); x[1, 3, 3, 200] = float("inf"); x[1, 7, 0, 64] = float("-inf")
    q_ref, s_ref, deq = EmuOps.mx_quant(x)
    q, s = hip.quant_mxfp8(x.cuda())
    assert torch.equal(q.cpu(), q_ref) and torch.equal(s.cpu(), s_ref)
    for (n, h, w, c) in ((0, 1, 2, 37), (1, 3, 3, 200), (1, 7, 0, 64)):
        blk = q.cpu()[n, h, w, 32 * (c // 32):32 * (c // 32) + 32]
        assert bool((blk == 0x7F).all()) and bool(torch.isnan(deq[n, h, w, c]))
    assert int((q.cpu() == 0x7F).sum()) == 3 * 32                      # nothing else was touched
    # fused epilogue: a conv whose OUTPUT has a NaN (NaN bias -> every pixel of that channel) writes NaN blocks in its fp8 copy
    cv = Conv(1, 16, 16, 128, 128, 1, False)
    xin = torch.randn(1, 16, 16, 128, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(128 * 9 * 128, generator=g) * 0.05).to(torch.bfloat16).cuda()
    b = torch.zeros(128); b[70] = float("nan")
    y = torch.zeros(1, 16, 16, 128, dtype=torch.bfloat16).cuda()
    oq = (torch.zeros(1, 16, 16, 128, dtype=torch.uint8).cuda(), torch.zeros(1, 16, 16, 4, dtype=torch.uint8).cuda())
    hip.conv_fwd(cv, xin, w, y, bias=b.cuda(), out_q=oq)
    assert bool(torch.isnan(y[..., 70].float()).all())
    assert bool((oq[0][..., 64:96] == 0x7F).all()) and not bool((oq[0][..., :64] == 0x7F).all())
    q2, s2 = hip.quant_mxfp8(y)
    assert torch.equal(oq[0], q2) and torch.equal(oq[1], s2)
    # and the next fp8 layer sees it: NaN in, NaN out
    cvn = Conv(1, 16, 16, 128, 128, 1, False, net="C")
    out = torch.zeros(1, 16, 16, 128, dtype=torch.bfloat16).cuda()
    hip.conv_fwd(cvn, y, w, out, xq=oq)
    assert bool(torch.isnan(out.float()).all())


F8_CONVS = [
    # N, H, W, Cin, Cout, stride  -- the critic's wide layers at cfg2 widths (critic.py:25-88), small grids; ragged tiles
    (1, 32, 32, 128, 128, 2),
    (1, 32, 32, 128, 256, 1),
    (2, 48, 80, 256, 256, 2),
    (1, 32, 32, 256, 512, 1),
    (1, 32, 32, 512, 512, 2),
    (1, 16, 16, 512, 1024, 1),
    (1, 32, 32, 1024, 1024, 2),
    (2, 24, 40, 128, 192, 1),      # partial output-channel tile
]


@pytest.mark.parametrize("cfg", F8_CONVS)
def test_fp8_conv_fwd_and_dgrad(cfg):
    N, H, W, ci, co, st = cfg
    hip, emu = HipOps("bf16", f8_critic=True), EmuOps("bf16", f8_critic=True)
    g = torch.Generator().manual_seed(21)
    cv = Conv(N, H, W, ci, co, st, False, net="C")
    x = (torch.randn(N, H, W, ci, generator=g) * torch.logspace(-2, 1, ci)).to(torch.bfloat16)
    w = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(torch.bfloat16)
    osh = emu.out_shape(cv)
    # forward: LeakyReLU + out_bits (critic forward) / 1-bit mask (penalty tangent forward)
    bits_shape = emu.bits_shape(osh) if co % 64 == 0 else None
    y_ref = torch.zeros(osh, dtype=torch.bfloat16)
    ob_ref = torch.zeros(bits_shape, dtype=torch.int16) if bits_shape else None
    emu.conv_fwd(cv, x, w, y_ref, act=0.2, out_bits=ob_ref)
    y = torch.zeros(osh, dtype=torch.bfloat16).cuda()
    ob = torch.zeros(bits_shape, dtype=torch.int16).cuda() if bits_shape else None
    hip.conv_fwd(cv, x.cuda(), w.cuda(), y, act=0.2, out_bits=ob)
    assert hip.lib.dg_last_conv_kernels() == 32                    # the fp8 kernel served the call
    close(y, y_ref, f"fp8 fwd {cfg}")
    if bits_shape:
        y2_ref, y2 = torch.zeros(osh, dtype=torch.bfloat16), torch.zeros(osh, dtype=torch.bfloat16).cuda()
        emu.conv_fwd(cv, x, w, y2_ref, mask_bits=ob_ref, mask_slope=0.2)
        hip.conv_fwd(cv, x.cuda(), w.cuda(), y2, mask_bits=ob_ref.cuda(), mask_slope=0.2)
        close(y2, y2_ref, f"fp8 fwd mask_bits {cfg}")
    # data gradient (tiny adjoints, as in the train step: the block scales carry the range)
    if ci % 64 == 0 and ci > 64:
        dy = (torch.randn(osh, generator=g) * 1e-4 * torch.logspace(-1, 1, co)).to(torch.bfloat16)
        wd = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(torch.bfloat16)
        mb = (torch.randint(0, 1 << 15, emu.bits_shape((N, H, W, ci)), generator=g)).to(torch.int16)
        dx_ref = torch.zeros(N, H, W, ci, dtype=torch.bfloat16)
        emu.conv_dgrad(cv, dy, wd, dx_ref, mask_bits=mb, mask_slope=0.2)
        dx = torch.zeros(N, H, W, ci, dtype=torch.bfloat16).cuda()
        hip.conv_dgrad(cv, dy.cuda(), wd.cuda(), dx, mask_bits=mb.cuda(), mask_slope=0.2)
        a, b = dx.float().cpu(), dx_ref.float()
        assert float((a - b).abs().max()) <= 1.6e-2 * float(b.abs().max()), (cfg, float((a - b).abs().max()), float(b.abs().max()))


def test_fp8_engine_iteration_matches_emulated_engine():
    """One critic + one generator iteration with the critic's wide convs in fp8, HIP against the emulated engine (bf16
    storage on both sides); F = 128 so that every critic layer but the first is eligible."""
    from downgan_amd import synthetic
    from downgan_amd.engine import HyperParams, TrainEngine
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(8)
    B, S, F_, cin, nrb = 1, 16, 128, 2, 1
    pg, pc = synthetic.generator_params(F_, cin, 2, nrb), synthetic.critic_params(F_, 8 * S, 2)
    coarse, fine = synthetic.tiles(B, cin, S)
    alpha = torch.from_numpy(synthetic.alpha(B, 0))
    res = {}
    for name in ("emu", "hip"):
        ops = EmuOps("bf16", f8_critic=True) if name == "emu" else HipOps("bf16", f8_critic=True)
        eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
        eng.G.load_state_dict(pg); eng.C.load_state_dict(pc)
        if name == "emu":
            xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.bfloat16)
            xf = nchw_to_nhwc_padded(torch.from_numpy(fine), 16, torch.bfloat16)
            al = alpha
        else:
            xc = ops.zeros(B, S, S, 16); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
            xf = ops.zeros(B, 8 * S, 8 * S, 16); ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
            al = alpha.cuda()
        eng.critic_iteration(xc, xf, al, apply_update=False, save_g=True)
        eng.generator_iteration(xc, xf, apply_update=False, reuse_fake=True)
        res[name] = eng.read_scalars(True)
    for k in ("c_real_mean", "c_fake_mean", "gp_ret", "critic_loss", "g_loss", "content_loss"):
        a, b = res["hip"][k], res["emu"][k]
        assert abs(a - b) <= 2e-2 * max(abs(b), 0.05), (k, a, b)


@pytest.mark.parametrize("case", ["halo_bf16", "im2col_first_layer", "fp8_producer", "s2_dgrad_classes", "general_epilogue"])
def test_fused_mxfp8_output_equals_quantised_store(case):
    """dg_epilogue.out_q / out_qs: the MXFP8 copy a conv epilogue writes beside its bf16 output is bit-identical to
    dg_quant_mxfp8 of that output (so the consumer's results do not depend on who quantised)."""
    g = torch.Generator().manual_seed(31)
    f8 = case == "fp8_producer"
    hip = HipOps("bf16", f8_critic=True)
    if case == "im2col_first_layer":          # critic features.0: 2 real input channels, 128 outputs (critic.py:21-24)
        cv = Conv(2, 48, 80, 16, 128, 1, False, cin_real=2, net="C")
    elif case == "s2_dgrad_classes":
        cv = Conv(1, 64, 96, 128, 256, 2, False, net="C")
    else:
        cv = Conv(2, 40, 56, 128, 256, 1, False, net="C" if f8 else "")
    def qpair(shape):
        return torch.zeros(shape, dtype=torch.uint8).cuda(), torch.zeros(tuple(shape[:-1]) + (shape[-1] // 32,), dtype=torch.uint8).cuda()
    if case == "s2_dgrad_classes":
        dy = (torch.randn(hip.out_shape(cv), generator=g) * 1e-3).to(torch.bfloat16).cuda()
        wd = (torch.randn(cv.Cout * 9 * cv.Cin, generator=g) * 0.05).to(torch.bfloat16).cuda()
        mb = torch.randint(0, 1 << 15, hip.bits_shape((cv.N, cv.H, cv.W, cv.Cin)), generator=g).to(torch.int16).cuda()
        y = torch.zeros(cv.N, cv.H, cv.W, cv.Cin, dtype=torch.bfloat16).cuda()
        oq = qpair(y.shape)
        hip.conv_dgrad(cv, dy, wd, y, mask_bits=mb, mask_slope=0.2, out_q=oq)
    else:
        x = torch.randn(cv.N, cv.H, cv.W, cv.Cin, generator=g).to(torch.bfloat16)
        if cv.cin_real:
            x[..., cv.cin_real:] = 0
        w = (torch.randn(cv.Cout * 9 * cv.Cin, generator=g) * 0.05).to(torch.bfloat16).cuda()
        y = torch.zeros(hip.out_shape(cv), dtype=torch.bfloat16).cuda()
        oq = qpair(y.shape)
        ob = torch.zeros(hip.bits_shape(y.shape), dtype=torch.int16).cuda()
        if case == "general_epilogue":     # residual + activation: not one of the straight-line epilogue instances
            r1 = torch.randn(y.shape, generator=g).to(torch.bfloat16).cuda()
            hip.conv_fwd(cv, x.cuda(), w, y, act=0.2, r1=r1, s1=0.5, out_q=oq)
        else:
            b = torch.randn(cv.Cout, generator=g).cuda() if cv.cin_real else None
            hip.conv_fwd(cv, x.cuda(), w, y, bias=b, act=0.2, out_bits=ob, out_q=oq)
            assert hip.lib.dg_last_conv_kernels() == (32 if f8 else 16 if cv.cin_real else 8)
    q_ref, s_ref = hip.quant_mxfp8(y)
    assert float(y.float().abs().max()) > 0
    assert torch.equal(oq[1], s_ref), int((oq[1] != s_ref).sum())
    assert torch.equal(oq[0], q_ref), int((oq[0] != q_ref).sum())


def test_fp8_generator_trunk_matches_emulation():
    """HipOps(f8_generator=True): the generator's dense-block trunk forward on the fp8 kernel with slab-sliced fp8 forms
    (strided scale rows, epilogue-written slices) against the emulated engine; backward still runs from the bf16 slabs."""
    from downgan_amd import synthetic
    from downgan_amd.engine import NativeGenerator
    from downgan_amd.layout import nchw_to_nhwc_padded
    torch.set_num_threads(8)
    B, S, F_, cin, nrb = 2, 16, 128, 2, 2
    pg = synthetic.generator_params(F_, cin, 2, nrb)
    coarse, _ = synthetic.tiles(B, cin, S)
    emu = NativeGenerator(EmuOps("bf16", f8_generator=True), F_, cin, B, S, num_res_blocks=nrb)
    emu.load_state_dict(pg)
    ref = emu.forward(nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.bfloat16), save=True).float()
    ops = HipOps("bf16", f8_generator=True)
    G = NativeGenerator(ops, F_, cin, B, S, num_res_blocks=nrb)
    assert G.f8
    G.load_state_dict(pg)
    xc = ops.zeros(B, S, S, 16); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
    for save in (False, True):
        got = G.forward(xc, save=save).float().cpu()
        err = float((got - ref)[..., :2].abs().max()) / max(1e-6, float(ref[..., :2].abs().max()))
        assert err < 3e-2, (save, err)
    close(G.trunk, emu.trunk, "fp8 trunk", tol=3e-2)
    dfake = torch.randn(G.fake.shape).to(torch.bfloat16).cuda()
    G.P.zero_grad()
    G.backward(xc, dfake)                                                       # bf16 backward from the saved slabs
    assert float(G.P.g.abs().sum()) > 0 and bool(torch.isfinite(G.P.g).all())


def test_mirrors_accept_fp8_dtype():
    """Generator / Critic / WassersteinGAN mirrors with dtype="fp8": the drop-in API reaches the MXFP8 path."""
    import math
    from downgan_amd import synthetic
    from downgan_amd.GAN.wasserstein import WassersteinGAN
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    B, S, F_ = 2, 16, 128
    G = Generator(F_, 8 * S, 2, 2, num_res_blocks=1, dtype="fp8")
    C = Critic(F_, 8 * S, 2, dtype="fp8")
    tr = WassersteinGAN(G, C)
    coarse, fine = synthetic.tiles(B, 2, S)
    out = tr._critic_train_iteration(torch.from_numpy(coarse), torch.from_numpy(fine), alpha=synthetic.alpha(B, 0), _keep_g=True)
    out.update(tr._generator_train_iteration(torch.from_numpy(coarse), torch.from_numpy(fine), _reuse_g=True))
    assert tr._engine.C.f8 and tr._engine.G.f8
    assert all(math.isfinite(v) for v in out.values()), out
