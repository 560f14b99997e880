"""Layout interchange between the reference's tensors and the native packed layouts (host side).

Reference layouts (the checkpoint contract): conv weights OIHW fp32, Linear [out,in]
with the critic's FC1 columns in NCHW-flatten order (reference DoWnGAN/networks/critic.py:103),
activations NCHW fp32.  Native layouts: conv master weights forward-packed and zero-padded as
[CoutP][3*3][CinP] fp32, pixel-shuffle layers with output rows reordered to (2i+j)*F + c
(torch.nn.PixelShuffle maps channel c*4 + 2i + j to sub-pixel (i, j), generator.py:73), FC1
columns in NHWC-flatten order.  These run once at load / export time on the CPU (plumbing).
"""
from __future__ import annotations

import torch


def pad16(c):
    return (c + 15) // 16 * 16


def pack_conv_weight(w_oihw: torch.Tensor, cout_p: int, cin_p: int, pixel_shuffle=False) -> torch.Tensor:
    """OIHW fp32 -> [cout_p, 9, cin_p] fp32 (zero padded)."""
    co, ci, kh, kw = w_oihw.shape
    assert kh == 3 and kw == 3 and co <= cout_p and ci <= cin_p
    w = w_oihw.permute(0, 2, 3, 1).reshape(co, 9, ci)
    if pixel_shuffle:
        assert co % 4 == 0 and co == cout_p
        f = co // 4
        w = w.view(f, 4, 9, ci).permute(1, 0, 2, 3).reshape(co, 9, ci)   # row q*f + c  <-  channel c*4 + q
    out = torch.zeros(cout_p, 9, cin_p, dtype=torch.float32)
    out[:co, :, :ci] = w
    return out


def unpack_conv_weight(packed: torch.Tensor, co: int, ci: int, pixel_shuffle=False) -> torch.Tensor:
    """Inverse of pack_conv_weight -> OIHW fp32."""
    w = packed[:co, :, :ci]
    if pixel_shuffle:
        f = co // 4
        w = w.reshape(4, f, 9, ci).permute(1, 0, 2, 3).reshape(co, 9, ci)
    return w.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()


def pack_bias(b: torch.Tensor, cout_p: int, pixel_shuffle=False) -> torch.Tensor:
    co = b.numel()
    if pixel_shuffle:
        f = co // 4
        b = b.view(f, 4).t().reshape(co)
    out = torch.zeros(cout_p, dtype=torch.float32)
    out[:co] = b
    return out


def unpack_bias(packed: torch.Tensor, co: int, pixel_shuffle=False) -> torch.Tensor:
    b = packed[:co]
    if pixel_shuffle:
        f = co // 4
        b = b.reshape(4, f).t().reshape(co)
    return b.contiguous()


def pack_fc1_weight(w: torch.Tensor, c: int, c_p: int, hf: int, wf: int, out_p: int) -> torch.Tensor:
    """[out, c*hf*wf] (NCHW flatten) -> [out_p, hf*wf*c_p] (NHWC flatten, zero padded)."""
    o = w.shape[0]
    w4 = w.view(o, c, hf, wf).permute(0, 2, 3, 1)
    out = torch.zeros(out_p, hf, wf, c_p, dtype=torch.float32)
    out[:o, :, :, :c] = w4
    return out.view(out_p, hf * wf * c_p)


def unpack_fc1_weight(packed: torch.Tensor, o: int, c: int, c_p: int, hf: int, wf: int) -> torch.Tensor:
    w4 = packed.view(-1, hf, wf, c_p)[:o, :, :, :c]
    return w4.permute(0, 3, 1, 2).reshape(o, c * hf * wf).contiguous()


def nchw_to_nhwc_padded(x: torch.Tensor, c_p: int, dtype) -> torch.Tensor:
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, c_p, dtype=dtype)
    out[..., :c] = x.permute(0, 2, 3, 1).to(dtype)
    return out
