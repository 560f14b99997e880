"""TEST INFRASTRUCTURE — CPU restatement of the MS-SSIM metric of the reference's per-step metrics pass.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.

Parity status: **UNPINNED**.  The reference computes this metric with the third-party package ``pytorch_msssim``
(``DoWnGAN/GAN/losses.py:5,37-38``; listed without a version in ``requirements.txt:17``).  The package is not
installed in the build container and not vendored in the reference, and the reference has no test or fixture for
it, so what follows restates the package's PUBLISHED algorithm (``pytorch_msssim/ssim.py`` of the
1.0.0 release: ``_fspecial_gauss_1d``, ``gaussian_filter``, ``_ssim``, ``ms_ssim``, ``MS_SSIM.forward``) and is anchored on the
reference's call site: ``MS_SSIM(win_size=7, data_range=1, channel=2)`` applied to per-channel min-max normalised
tensors (``losses.py:12-38``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)   # pytorch_msssim default weights (Wang et al. 2003)


def gauss_1d(size: int, sigma: float) -> torch.Tensor:
    """``_fspecial_gauss_1d``: normalised 1-D Gaussian, float32."""
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """``gaussian_filter``: separable, per-channel (groups=C), no padding ('valid'), first along H then along W."""
    C = x.shape[1]
    out = x
    w = win.view(1, 1, 1, -1).repeat(C, 1, 1, 1)
    for i, s in enumerate(x.shape[2:]):
        if s >= win.numel():
            out = F.conv2d(out, w.transpose(2 + i, -1), stride=1, padding=0, groups=C)
    return out


def ssim_and_cs(X, Y, data_range, win, K=(0.01, 0.03)):
    """``_ssim``: per-(image, channel) spatial means of the SSIM map and of the contrast-structure map."""
    K1, K2 = K
    C1 = (K1 * data_range) ** 2
    C2 = (K2 * data_range) ** 2
    mu1 = gaussian_filter(X, win)
    mu2 = gaussian_filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    sigma1_sq = gaussian_filter(X * X, win) - mu1_sq
    sigma2_sq = gaussian_filter(Y * Y, win) - mu2_sq
    sigma12 = gaussian_filter(X * Y, win) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1), torch.flatten(cs_map, 2).mean(-1)


def ms_ssim(X, Y, data_range=1.0, win_size=7, win_sigma=1.5, weights=MS_WEIGHTS, K=(0.01, 0.03), per_plane=False):
    """``ms_ssim`` with ``size_average=True``: 5 scales, 2x2 average pooling (padding = size % 2) between them,
    relu on the per-scale terms, weighted geometric combination, mean over (image, channel)."""
    assert X.shape == Y.shape and X.dim() == 4
    assert min(X.shape[-2:]) > (win_size - 1) * (2 ** 4), "image too small for 5 scales"
    win = gauss_1d(win_size, win_sigma)
    wts = torch.tensor(weights, dtype=X.dtype)
    mcs = []
    ssim_pc = None
    for i in range(len(weights)):
        ssim_pc, cs = ssim_and_cs(X, Y, data_range, win, K)
        if i < len(weights) - 1:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    ssim_pc = torch.relu(ssim_pc)
    stack = torch.stack(mcs + [ssim_pc], dim=0)                 # (level, image, channel)
    val = torch.prod(stack ** wts.view(-1, 1, 1), dim=0)
    return val if per_plane else val.mean()


def minmax_normalise(x: torch.Tensor) -> torch.Tensor:
    """DoWnGAN/GAN/losses.py:15-21 (and :23-29): every channel scaled to [0, 1] with its min / max over the WHOLE
    batch.  The reference does this in place on its arguments; this returns a new tensor."""
    mn = x.amin(dim=(0, 2, 3), keepdim=True)
    mx = x.amax(dim=(0, 2, 3), keepdim=True)
    return (x - mn) / (mx - mn)


def ssim_loss(x: torch.Tensor, y: torch.Tensor) -> float:
    """DoWnGAN/GAN/losses.py:12-38 ``SSIM_Loss`` (despite the name it returns the MS-SSIM value, higher = more similar)."""
    return float(ms_ssim(minmax_normalise(x.float()), minmax_normalise(y.float()), data_range=1.0, win_size=7))
