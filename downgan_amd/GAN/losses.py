"""Same-named counterparts of the metric / loss functions of the reference (DoWnGAN/GAN/losses.py) for the four that the
training loop uses (hyperparams.py:38-43 ``metrics_to_calculate``): NCHW tensors in, 0-dim fp32 tensor out (the reference's
caller does ``.detach().cpu().item()`` on every return, mlflow_epoch.py:58-61), all arithmetic in the HIP kernels (no torch
fallback; raises without the native library or a GPU).

  content_loss(hr, fake, device)      losses.py:40-55   nn.L1Loss()            -> dg_l1
  content_MSELoss(hr, fake, device)   losses.py:58-70   nn.MSELoss()           -> dg_sqdiff
  SSIM_Loss(x, y, device)             losses.py:12-38   MS-SSIM of the batch-min-max-normalised fields (pytorch_msssim,
                                                        win 7, data_range 1)   -> csrc/metrics.hip via msssim.MsSsim
  wass_loss(real, fake, device)       losses.py:8-9     real - fake
  divergence_loss / vorticity_loss    losses.py:119-193 std-normalised MSE of finite-difference fields -> dg_div_vort_sums

Unlike the reference's SSIM_Loss this one does NOT normalise its arguments in place (the reference's mutation is a side
effect that nothing downstream reads: the metrics pass is the last use of the batch, wasserstein.py:138-146).
"""
from __future__ import annotations

import torch

from .. import layout
from ..msssim import MsSsim
from .. import backend

_ops = {}
_ms = {}


def _o(device=None):
    key = str(device or "cuda:0")
    if key not in _ops:
        _ops[key] = backend.make_ops("f32", key if key.startswith("cuda") else "cuda:0")
    return _ops[key]


def _native(o, t):
    N, C, H, W = t.shape
    out = o.zeros(N, H, W, layout.pad16(C))
    o.nchw_to_nhwc(t.to(o.device, torch.float32).contiguous(), out)
    return out


def _scalar(o, v):
    """0-dim fp32 tensor on the compute device, like the reference's loss modules return."""
    return torch.tensor(float(v), dtype=torch.float32, device=o.device)


def wass_loss(real, fake, device=None):
    return real - fake


def content_loss(hr, fake, device=None):
    o = _o(device)
    acc = o.zeros(1, dtype=torch.float32)
    o.l1(_native(o, hr), _native(o, fake), acc)
    return _scalar(o, float(acc.item()) / hr.numel())


def content_MSELoss(hr, fake, device=None):
    o = _o(device)
    acc = o.zeros(1, dtype=torch.float32)
    o.sqdiff(_native(o, hr), _native(o, fake), acc)
    return _scalar(o, float(acc.item()) / hr.numel())


def SSIM_Loss(x, y, device=None, reduction="mean", window_size=11):
    """``reduction`` / ``window_size`` are accepted and ignored exactly like in the reference (it hard-codes win_size=7)."""
    o = _o(device)
    N, C, H, W = x.shape
    key = (str(o.device), N, C, H, W)
    if key not in _ms:
        _ms[key] = MsSsim(o, N, H, W, c_real=C)
    return _scalar(o, _ms[key](_native(o, x), _native(o, y)))


def _std_normalised_mse(m, n):
    """MSE(r / std(r), f / std(f)) from {sum r, sum r^2, sum f, sum f^2, sum r f} (float64), torch.std = unbiased."""
    sr, srr, sf, sff, srf = m
    var_r = (srr - sr * sr / n) / (n - 1)
    var_f = (sff - sf * sf / n) / (n - 1)
    return (srr / var_r - 2.0 * srf / (var_r * var_f) ** 0.5 + sff / var_f) / n


def _div_vort(hr, fake, device):
    o = _o(device)
    sums = torch.zeros(10, dtype=torch.float64, device=o.device)
    o.div_vort_sums(_native(o, hr), _native(o, fake), sums)
    N, _, H, W = hr.shape
    return sums.cpu().tolist(), N * (H - 1) * (W - 1)


def divergence_loss(hr, fake, device=None):
    """losses.py:119-156 (u = channel 0 differenced along H, v = channel 1 along W)."""
    m, n = _div_vort(hr, fake, device)
    return _scalar(_o(device), _std_normalised_mse(m[:5], n))


def vorticity_loss(hr, fake, device=None):
    """losses.py:158-193."""
    m, n = _div_vort(hr, fake, device)
    return _scalar(_o(device), _std_normalised_mse(m[5:], n))


metrics_to_calculate = {"MAE": content_loss, "MSE": content_MSELoss, "MSSSIM": SSIM_Loss, "Wass": wass_loss}   # hyperparams.py:38-43
