"""Native MS-SSIM of the reference's per-step metrics pass.

Mirrors ``SSIM_Loss`` (DoWnGAN/GAN/losses.py:12-38): every channel of ``x`` (real) and ``y`` (fake) is min-max normalised over
the whole batch, then ``pytorch_msssim.MS_SSIM(win_size=7, data_range=1, channel=2)`` is evaluated (5 scales, Gaussian sigma 1.5,
K = (0.01, 0.03), 2x2 average pooling between scales, mean over image x channel).  The third-party package is unpinned in
the reference (requirements.txt:17); the kernels follow its published algorithm (see ``oracle/msssim.py``).
All arithmetic runs in the HIP kernels of ``csrc/metrics.hip``; this class only owns the buffers.
"""
from __future__ import annotations

import torch

from . import _lib

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def level_sizes(H, W, levels):
    """Sizes after avg_pool2d(kernel 2, padding = size % 2) between scales."""
    out = [(H, W)]
    for _ in range(levels - 1):
        H = (H + 2 * (H % 2) - 2) // 2 + 1
        W = (W + 2 * (W % 2) - 2) // 2 + 1
        out.append((H, W))
    return out


class MsSsim:
    def __init__(self, ops, N, H, W, c_real=2, win_size=7, win_sigma=1.5, K=(0.01, 0.03), data_range=1.0, weights=WEIGHTS):
        assert min(H, W) > (win_size - 1) * 2 ** 4, "pytorch_msssim asserts smaller_side > (win_size - 1) * 2**4"
        assert win_size % 2 == 1 and win_size <= 11 and c_real <= 8
        self.ops, self.N, self.c, self.levels = ops, N, c_real, len(weights)
        self.sizes = level_sizes(H, W, self.levels)
        coords = torch.arange(win_size, dtype=torch.float32) - win_size // 2   # _fspecial_gauss_1d, float32 like the package
        g = torch.exp(-(coords ** 2) / (2 * win_sigma ** 2))
        g = g / g.sum()
        self.params = _lib.SsimParams(win=win_size, C1=(K[0] * data_range) ** 2, C2=(K[1] * data_range) ** 2)
        for i, v in enumerate(g.tolist()):
            self.params.g[i] = v
        self.combine = _lib.MsssimCombine()
        for l, (h, w) in enumerate(self.sizes):
            self.combine.inv_count[l] = 1.0 / ((h - win_size + 1) * (w - win_size + 1))
            self.combine.weight[l] = weights[l]
        f32 = torch.float32
        self.X = [ops.empty(N, c_real, h, w, dtype=f32) for h, w in self.sizes]
        self.Y = [ops.empty(N, c_real, h, w, dtype=f32) for h, w in self.sizes]
        self.partial = ops.empty(_lib.MINMAX_PARTS * c_real * 2, dtype=f32)
        self.minmax = ops.empty(2, 2 * c_real, dtype=f32)       # row 0: x, row 1: y
        self.sums = ops.zeros(self.levels, N * c_real, 2, dtype=f32)
        self.out = ops.zeros(1, dtype=f32)

    def __call__(self, x, y, dist=None, world=1):
        """x, y: NHWC activation tensors whose first ``c_real`` channels are the fields.  Returns the MS-SSIM value (float).
        With ``dist`` the min/max and the final mean are taken over the global batch."""
        o, c = self.ops, self.c
        o.minmax(x, c, self.partial, self.minmax[0])
        o.minmax(y, c, self.partial, self.minmax[1])
        if dist is not None and world > 1:
            dist.minmax_(self.minmax, c)
        o.normalise_planar(x, c, self.minmax[0], self.X[0])
        o.normalise_planar(y, c, self.minmax[1], self.Y[0])
        self.sums.zero_()
        for l in range(self.levels):
            o.ssim_level(self.X[l], self.Y[l], self.params, self.sums[l])
            if l + 1 < self.levels:
                o.avgpool2(self.X[l], self.X[l + 1])
                o.avgpool2(self.Y[l], self.Y[l + 1])
        o.msssim_finish(self.sums, self.levels, self.N * c, self.combine, self.out)
        total = float(self.out.item())
        if dist is not None and world > 1:
            total = dist.reduce_scalars({"s": total}, total=("s",))["s"]
        return total / (self.N * c * world)
