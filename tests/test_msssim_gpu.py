"""MS-SSIM kernels (csrc/metrics.hip) through the C ABI against oracle/msssim.py on the same seeded inputs.
Tolerances: min/max exact; normalised planes 1e-6 abs (one fp32 division); per-scale means and the final index 2e-5 abs
(fp32 sums of up to 1e6 map values in a different order from the oracle's).  bf16 mode feeds the oracle the same
bf16-rounded fields: only the input rounding differs from fp32 mode."""
import math

import pytest
import torch

from oracle import msssim as om

pytestmark = pytest.mark.gpu


def fields(N, C, H, W, seed=0, noise=0.3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    x = torch.nn.functional.avg_pool2d(x, 5, stride=1, padding=2) * 3.0
    y = x + noise * torch.randn(N, C, H, W, generator=g)
    return x, y


def native(dtype):
    from downgan_amd.ops import HipOps
    return HipOps(dtype)


def to_nhwc(o, t, cpad=16):
    N, C, H, W = t.shape
    out = o.zeros(N, H, W, cpad)
    o.nchw_to_nhwc(t.cuda().contiguous(), out)
    return out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 2, 128, 128), (1, 2, 200, 136), (2, 3, 101, 99), (3, 2, 256, 256)])
def test_msssim_matches_oracle(dtype, shape):
    from downgan_amd.msssim import MsSsim
    N, C, H, W = shape
    x, y = fields(N, C, H, W, seed=11)
    o = native(dtype)
    xn, yn = to_nhwc(o, x), to_nhwc(o, y)
    ms = MsSsim(o, N, H, W, c_real=C)
    got = ms(xn, yn)
    xr, yr = xn[..., :C].float().permute(0, 3, 1, 2).cpu(), yn[..., :C].float().permute(0, 3, 1, 2).cpu()   # what the kernels saw
    ref = om.ssim_loss(xr, yr)
    assert math.isfinite(got) and abs(got - ref) < 2e-5, (got, ref)
    # pieces
    mm = ms.minmax.cpu().view(2, C, 2)
    assert torch.equal(mm[0, :, 0], xr.amin((0, 2, 3))) and torch.equal(mm[0, :, 1], xr.amax((0, 2, 3)))
    assert torch.equal(mm[1, :, 0], yr.amin((0, 2, 3))) and torch.equal(mm[1, :, 1], yr.amax((0, 2, 3)))
    Xo, Yo = om.minmax_normalise(xr), om.minmax_normalise(yr)
    assert float((ms.X[0].cpu() - Xo).abs().max()) < 1e-6
    win = om.gauss_1d(7, 1.5)
    sums = ms.sums.cpu()
    for l in range(5):
        s_o, c_o = om.ssim_and_cs(Xo, Yo, 1.0, win)
        h, w = ms.sizes[l]
        cnt = (h - 6) * (w - 6)
        assert float((sums[l, :, 0] / cnt - s_o.flatten()).abs().max()) < 2e-5, l
        assert float((sums[l, :, 1] / cnt - c_o.flatten()).abs().max()) < 2e-5, l
        if l < 4:
            pad = [Xo.shape[2] % 2, Xo.shape[3] % 2]
            Xo = torch.nn.functional.avg_pool2d(Xo, 2, padding=pad)
            Yo = torch.nn.functional.avg_pool2d(Yo, 2, padding=pad)
            assert float((ms.X[l + 1].cpu() - Xo).abs().max()) < 1e-6


def test_identical_fields_give_one_and_size_independent_properties():
    """At the full 1024x1024 tile of BASELINE configs[1] (batch 2): MS-SSIM(x, x) = 1, symmetry, monotone in the noise."""
    from downgan_amd.msssim import MsSsim
    o = native("bf16")
    x, y = fields(2, 2, 1024, 1024, seed=2, noise=0.2)
    _, y2 = fields(2, 2, 1024, 1024, seed=2, noise=0.8)
    xn, yn, y2n = to_nhwc(o, x), to_nhwc(o, y), to_nhwc(o, y2)
    ms = MsSsim(o, 2, 1024, 1024)
    same = ms(xn, xn)
    a, b, c = ms(xn, yn), ms(yn, xn), ms(xn, y2n)
    assert abs(same - 1.0) < 1e-5
    assert abs(a - b) < 1e-5 and c < a < 1.0


def test_reference_named_metric_functions():
    """downgan_amd.GAN.losses mirrors DoWnGAN/GAN/losses.py: NCHW tensors in, 0-dim tensor out, consumed the way the
    reference's metrics pass does (mlflow_epoch.py:58-61: ``.detach().cpu().item()``)."""
    from downgan_amd.GAN import losses as L
    x, y = fields(2, 2, 128, 128, seed=4)
    val = lambda t: t.detach().cpu().item()
    assert torch.is_tensor(L.content_loss(x, y)) and L.content_loss(x, y).dim() == 0
    assert abs(val(L.content_loss(x, y, "cuda:0")) - float(torch.nn.functional.l1_loss(x, y))) < 1e-6
    assert abs(val(L.content_MSELoss(x, y)) - float(torch.nn.functional.mse_loss(x, y))) < 1e-6
    assert abs(val(L.SSIM_Loss(x, y)) - om.ssim_loss(x, y)) < 2e-5
    assert abs(val(L.wass_loss(torch.tensor(0.5), torch.tensor(0.25), "cuda:0")) - 0.25) < 1e-7
    assert set(L.metrics_to_calculate) == {"MAE", "MSE", "MSSSIM", "Wass"}


def test_divergence_and_vorticity_known_answers_and_oracle():
    """losses.py:119-193 through the C ABI: the reference's own known answers (test_losses.py:75-116) and the oracle on
    random fields (relative 1e-5: the kernel accumulates the moments in double, the reference in fp32)."""
    import numpy as np
    from downgan_amd.GAN import losses as L
    from oracle import physics
    hr, fake = physics.reference_test_fixture()
    assert np.isclose(L.divergence_loss(hr, fake).item(), physics.KNOWN["divergence"], atol=physics.KNOWN["atol"])
    assert np.isclose(L.vorticity_loss(hr, fake).item(), physics.KNOWN["vorticity"], atol=physics.KNOWN["atol"])
    assert abs(L.divergence_loss(hr, fake).item() - physics.divergence_loss(hr, fake)) < 1e-6
    x, y = fields(3, 2, 97, 130, seed=8, noise=0.5)
    for nat, orc in ((L.divergence_loss, physics.divergence_loss), (L.vorticity_loss, physics.vorticity_loss)):
        a, b = nat(x, y).item(), orc(x, y)
        assert abs(a - b) < 1e-5 * abs(b) + 1e-7, (a, b)
    assert abs(L.divergence_loss(x, x).item()) < 1e-9


def test_cabi_rejects_bad_arguments():
    import ctypes as C
    from downgan_amd import _lib
    lib = _lib.lib()
    p = _lib.SsimParams(win=13)
    assert lib.dg_ssim_level(None, None, 1, 32, 32, C.byref(p), None, None) == -1
    assert lib.dg_minmax_partial(_lib.DG_F32, None, 10, 16, 2, None, None) == -1
    assert lib.dg_avgpool2(None, None, 1, 1, 1, None) == -1
