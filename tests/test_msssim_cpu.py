"""MS-SSIM of the metrics pass, CPU side: the restatement in oracle/msssim.py is checked against an
independent direct evaluation of the published formulas and against the properties of the index; the host-side buffer
logic (downgan_amd/msssim.py) runs on the torch-CPU op emulation and must reproduce the oracle.

Parity status of this metric: UNPINNED (third-party pytorch_msssim, absent from the reference tree and the container)."""
import math

import numpy as np
import pytest
import torch

from downgan_amd.msssim import MsSsim, level_sizes
from downgan_amd.layout import nchw_to_nhwc_padded
from oracle import msssim as om
from oracle.emu_ops import EmuOps


def fields(N, C, H, W, seed=0, noise=0.3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    x = torch.nn.functional.avg_pool2d(x, 5, stride=1, padding=2) * 3.0          # spatially correlated, like real fields
    y = x + noise * torch.randn(N, C, H, W, generator=g)
    return x, y


def test_gaussian_window_is_normalised_and_symmetric():
    g = om.gauss_1d(7, 1.5)
    assert abs(float(g.sum()) - 1.0) < 1e-6 and torch.allclose(g, g.flip(0))
    assert float(g[3]) == float(g.max())


def test_single_scale_ssim_matches_direct_formula():
    """Direct double-loop evaluation (float64 numpy) of the SSIM / CS maps of Wang et al. with a 7x7 separable Gaussian."""
    x, y = fields(1, 1, 20, 17, seed=3)
    x, y = om.minmax_normalise(x), om.minmax_normalise(y)
    g = om.gauss_1d(7, 1.5).double().numpy()
    w2 = np.outer(g, g)
    X, Y = x[0, 0].double().numpy(), y[0, 0].double().numpy()
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ss, cs = [], []
    for i in range(20 - 6):
        for j in range(17 - 6):
            a, b = X[i:i + 7, j:j + 7], Y[i:i + 7, j:j + 7]
            mu1, mu2 = (w2 * a).sum(), (w2 * b).sum()
            s1, s2, s12 = (w2 * a * a).sum() - mu1 ** 2, (w2 * b * b).sum() - mu2 ** 2, (w2 * a * b).sum() - mu1 * mu2
            c = (2 * s12 + C2) / (s1 + s2 + C2)
            cs.append(c)
            ss.append((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1) * c)
    s_o, c_o = om.ssim_and_cs(x, y, 1.0, om.gauss_1d(7, 1.5))
    assert abs(float(s_o) - np.mean(ss)) < 2e-5 and abs(float(c_o) - np.mean(cs)) < 2e-5


def test_ms_ssim_properties():
    x, y = fields(2, 2, 128, 128, seed=1)
    xn, yn = om.minmax_normalise(x), om.minmax_normalise(y)
    assert abs(float(om.ms_ssim(xn, xn)) - 1.0) < 1e-6                       # identical images
    a, b = float(om.ms_ssim(xn, yn)), float(om.ms_ssim(yn, xn))
    assert abs(a - b) < 1e-6 and 0.0 < a < 1.0                                # symmetric, bounded
    y2 = x + 1.0 * torch.randn(x.shape, generator=torch.Generator().manual_seed(9))
    assert float(om.ms_ssim(xn, om.minmax_normalise(y2))) < a                 # more noise, lower index
    with pytest.raises(AssertionError):
        om.ms_ssim(xn[..., :96, :96], yn[..., :96, :96])                      # the package's minimum-size assertion
    per = om.ms_ssim(xn, yn, per_plane=True)
    assert per.shape == (2, 2) and abs(float(per.mean()) - a) < 1e-7


def test_minmax_normalise_is_per_channel_over_the_batch():
    x, _ = fields(3, 2, 16, 16, seed=5)
    n = om.minmax_normalise(x)
    for c in range(2):
        assert float(n[:, c].min()) == 0.0 and float(n[:, c].max()) == 1.0    # the reference's own assertions (losses.py:31-34)
    assert float(n[0].max()) <= 1.0


def test_level_sizes_follow_avg_pool_padding():
    for H, W in [(128, 128), (200, 136), (101, 99), (1024, 1024)]:
        t = torch.zeros(1, 1, H, W)
        sizes = level_sizes(H, W, 5)
        for l in range(4):
            assert tuple(t.shape[2:]) == sizes[l]
            t = torch.nn.functional.avg_pool2d(t, 2, padding=[t.shape[2] % 2, t.shape[3] % 2])
        assert tuple(t.shape[2:]) == sizes[4]


@pytest.mark.parametrize("shape", [(2, 2, 128, 128), (1, 2, 200, 136), (2, 3, 101, 99)])
def test_host_logic_on_emulated_ops_matches_oracle(shape):
    N, C, H, W = shape
    x, y = fields(N, C, H, W, seed=7)
    ops = EmuOps("f32")
    xn, yn = nchw_to_nhwc_padded(x, 16, ops.tdtype), nchw_to_nhwc_padded(y, 16, ops.tdtype)
    got = MsSsim(ops, N, H, W, c_real=C)(xn, yn)
    ref = om.ssim_loss(x, y)
    assert math.isfinite(got) and abs(got - ref) < 1e-5, (got, ref)
