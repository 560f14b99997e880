"""TEST INFRASTRUCTURE — CPU restatement of the reference's finite-difference physics metrics.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.

Parity status: PINNED by the reference's own known-answer tests (DoWnGAN/GAN/tests/test_losses.py:75-116: Gaussian-bump
fixture, divergence 0.0018 and vorticity 0.00144 within 1e-4) — see tests/test_physics_cpu.py.
"""
from __future__ import annotations

import torch


def _dudy_dvdx(t):
    """losses.py:138-139 / :177-178: forward differences of channel 0 along H and of channel 1 along W on the [1:, 1:] window
    (not divided by the grid spacing: regular grids)."""
    dudy = t[:, 0, 1:, 1:] - t[:, 0, :-1, 1:]
    dvdx = t[:, 1, 1:, 1:] - t[:, 1, 1:, :-1]
    return dudy, dvdx


def _std_normalised_mse(a, b):
    """losses.py:148-156: both fields divided by their own (unbiased, all-element) standard deviation, then nn.MSELoss."""
    a = a / torch.std(a)
    b = b / torch.std(b)
    return torch.nn.functional.mse_loss(a, b).item()


def divergence_loss(hr, fake):
    """DoWnGAN/GAN/losses.py:119-156."""
    ur, vr = _dudy_dvdx(hr)
    uf, vf = _dudy_dvdx(fake)
    return _std_normalised_mse(ur + vr, uf + vf)


def vorticity_loss(hr, fake):
    """DoWnGAN/GAN/losses.py:158-193."""
    ur, vr = _dudy_dvdx(hr)
    uf, vf = _dudy_dvdx(fake)
    return _std_normalised_mse(vr - ur, vf - uf)


def reference_test_fixture():
    """The 'grads' fixture of the reference's own test (test_losses.py:21-37): 64 x 2 x 10 x 12, both channels of every sample a
    Gaussian bump exp(-(x^2+y^2)) (real) / exp(-(x^4+y^4)) (fake) on the integer grid [-5,5) x [-6,6)."""
    xx, yy = torch.meshgrid(torch.arange(-5, 5), torch.arange(-6, 6), indexing="ij")
    zr = torch.exp(-(xx ** 2 + yy ** 2).float())
    zf = torch.exp(-(xx ** 4 + yy ** 4).float())
    hr = zr.expand(64, 2, 10, 12).contiguous()
    fake = zf.expand(64, 2, 10, 12).contiguous()
    return hr, fake


KNOWN = {"divergence": 0.0018, "vorticity": 0.00144, "atol": 1e-4}     # test_losses.py:91-94, 113-116
