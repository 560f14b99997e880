"""CPU checks of the host planner (dg_conv3x3_plan in libdowngan_hip.so) and of the emulated op
contracts (oracle/emu_ops.py) against torch's own conv / autograd.  No GPU needed."""
import pytest
import torch
import torch.nn.functional as F

from downgan_amd import layout
from downgan_amd.ops import Conv
from oracle.emu_ops import EmuOps


def _rand_conv(co, ci, stride, ps, N=2, H=8, W=6, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) * 0.2
    b = torch.randn(co, generator=g)
    return x, w, b


@pytest.mark.parametrize("co,ci,stride,ps", [(16, 16, 1, False), (32, 8, 2, False), (24, 2, 1, False), (64, 16, 1, True), (2, 16, 1, False)])
def test_conv_fwd_dgrad_wgrad_match_torch(co, ci, stride, ps):
    ops = EmuOps("f32")
    x, w, b = _rand_conv(co, ci, stride, ps)
    N, _, H, W = x.shape
    cip, cop = layout.pad16(ci), layout.pad16(co)
    cv = Conv(N, H, W, cip, cop, stride, ps)
    xn = layout.nchw_to_nhwc_padded(x, cip, torch.float32)
    wm = layout.pack_conv_weight(w, cop, cip, ps)
    bm = layout.pack_bias(b, cop, ps)
    y = torch.zeros(ops.out_shape(cv))
    ops.conv_fwd(cv, xn, wm.reshape(-1), y, bias=bm, act=0.2)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.leaky_relu(F.conv2d(xr, wr, b, stride=stride, padding=1), 0.2)
    if ps:
        ref = F.pixel_shuffle(ref, 2)
    cr = ref.shape[1]
    assert torch.allclose(y[..., :cr], ref.permute(0, 2, 3, 1), atol=1e-5)
    assert (y[..., cr:] == 0).all()
    # backward: dy random, with LeakyReLU' taken from the saved activation (mask)
    dy = torch.randn(ref.shape)
    (dxr, dwr) = torch.autograd.grad(ref, [xr, wr], dy)
    dyn = layout.nchw_to_nhwc_padded(dy, y.shape[-1], torch.float32)
    u = dyn.clone()
    ops.mask_mul(u, y, 0.2)
    wd = torch.zeros(cop * 9 * cip)
    ops.repack(wm.reshape(-1), wd, cop, cip, 1)
    dx = torch.full((N, H, W, cip), 7.0)
    ops.conv_dgrad(cv, u, wd, dx)
    assert torch.allclose(dx[..., :ci], dxr.permute(0, 2, 3, 1), atol=1e-4)
    dw = torch.zeros(cop * 9 * cip)
    ops.conv_wgrad(cv, xn, u, dw)
    assert torch.allclose(layout.unpack_conv_weight(dw.view(cop, 9, cip), co, ci, ps), dwr, atol=1e-4)
    db = torch.zeros(cop)
    (ops.colsum_ps if ps else ops.colsum)(u, db)
    bref = torch.autograd.grad(F.conv2d(x, w, b.clone().requires_grad_(True), stride=stride, padding=1).sum(), [])if False else None
    pre = F.conv2d(x, w, b, stride=stride, padding=1)
    dpre = dy if not ps else F.pixel_unshuffle(dy, 2)
    dpre = dpre * torch.where(pre > 0, 1.0, 0.2)
    assert torch.allclose(layout.unpack_bias(db, co, ps), dpre.sum((0, 2, 3)), atol=1e-4)


def test_pack_roundtrip():
    w = torch.randn(64, 10, 3, 3)
    for ps in (False, True):
        p = layout.pack_conv_weight(w, 64, 16, ps)
        assert torch.equal(layout.unpack_conv_weight(p, 64, 10, ps), w)
    b = torch.randn(64)
    assert torch.equal(layout.unpack_bias(layout.pack_bias(b, 64, True), 64, True), b)
    fc = torch.randn(100, 8 * 4 * 4)
    p = layout.pack_fc1_weight(fc, 8, 16, 4, 4, 112)
    assert torch.equal(layout.unpack_fc1_weight(p, 100, 8, 16, 4, 4), fc)
