"""GPU parity of every HIP kernel (through the C ABI) against the CPU oracle ops (oracle/emu_ops.py,
itself pinned to torch conv/autograd in tests/test_emu_ops_cpu.py).  fp32 kernels must agree to fp32
re-association noise; bf16 kernels are compared with the oracle evaluated on the same bf16-rounded
inputs (tolerance = bf16 output rounding + fp32 accumulation order)."""
import pytest
import torch

from downgan_amd.ops import Conv, HipOps
from oracle.emu_ops import EmuOps

pytestmark = pytest.mark.gpu

TOL = {"f32": dict(rtol=2e-5, atol=2e-5), "bf16": dict(rtol=1.6e-2, atol=1.6e-2)}


def pair(dtype):
    return HipOps(dtype), EmuOps(dtype)


def rnd(shape, dtype, gen, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(dtype)


def close(a, b, dtype, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    tol = TOL[dtype]
    scale = max(1.0, float(b.abs().max()))
    err = (a - b).abs().max().item()
    assert err <= tol["atol"] * scale, f"{what}: max err {err:.3e} (scale {scale:.3e}); worst idx {(a - b).abs().argmax().item()}"


CONVS = [
    # N, H, W, Cin, Cout, stride, ps
    (2, 16, 16, 16, 16, 1, False),
    (2, 16, 16, 16, 16, 2, False),
    (1, 12, 20, 32, 64, 1, False),
    (1, 12, 20, 64, 32, 2, False),
    (2, 8, 8, 128, 128, 1, False),
    (1, 16, 16, 128, 256, 2, False),
    (1, 8, 8, 256, 128, 1, False),
    (2, 16, 16, 16, 64, 1, True),
    (1, 8, 8, 64, 256, 1, True),
    (1, 20, 12, 80, 48, 1, False),
    (3, 6, 10, 16, 128, 1, False),
    (1, 32, 32, 128, 16, 1, False),
    (1, 64, 64, 64, 128, 2, False),     # Wo = 32: row-aligned K-steps in wgrad, halo tiles in dgrad
    (1, 32, 32, 64, 256, 1, True),
    (2, 32, 64, 128, 128, 1, False),
    (2, 48, 80, 256, 128, 2, False),    # stride-2 forward on the halo kernel: 4 parity planes x 2 channel blocks, ragged 24x40 grid
    (1, 32, 32, 128, 192, 2, False),    # ... with a partial channel tile
    (1, 32, 48, 128, 256, 1, False),    # one reduction block, 2 output-channel tiles: channel tiles walked inside the workgroup
    (1, 32, 32, 128, 512, 1, True),     # ... 4 tiles with the pixel-shuffle store (the generator's up-sampling convs)
    (2, 24, 40, 128, 16, 1, False),     # <= 16 output channels: small-N halo kernel (forward), ragged tiles
    (1, 32, 48, 16, 192, 1, False),     # ... as a data gradient (16 input channels, 3 reduction blocks)
    (1, 32, 32, 16, 64, 2, False),      # ... and its stride-2 parity classes
    (1, 32, 32, 256, 128, 1, False),    # >= 256 input channels, Wo % 32 == 0: row-of-taps weight-gradient kernel
    (1, 16, 64, 320, 192, 1, False),    # ... ragged input- and output-channel tiles
    (1, 32, 32, 256, 256, 1, True),     # ... with a pixel-shuffled adjoint
    (2, 64, 48, 128, 128, 2, False),    # stride 2 with a single reduction-channel block (the critic's features.2 shape class)
    (1, 32, 48, 192, 64, 2, False),     # ... and a partial channel tile
    (1, 64, 64, 256, 128, 2, False),    # stride 2, Wo % 32 == 0, >= 256 input channels: wide DMA-staged weight-gradient kernel
    (2, 64, 128, 320, 192, 2, False),   # ... ragged channel tiles, two images, Wo = 64
    (1, 32, 64, 384, 256, 1, False),    # stride-1 wide weight-gradient kernel: 3 input-channel tiles, 2 adjoint tiles
    # the critic's widest layers at BASELINE configs[1] widths (critic.py:52-88 with cd = 128): 4 / 8 channel tiles on both sides
    (1, 64, 64, 512, 512, 2, False),    # features.10
    (1, 32, 32, 512, 1024, 1, False),   # features.12
    (1, 64, 64, 1024, 1024, 2, False),  # features.14
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd(dtype, cfg):
    N, H, W, ci, co, st, ps = cfg
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(1)
    cv = Conv(N, H, W, ci, co, st, ps)
    x = rnd((N, H, W, ci), emu.tdtype, g)
    w = rnd((co * 9 * ci,), emu.tdtype, g, 0.1)
    b = torch.randn(co, generator=g)
    osh = emu.out_shape(cv)
    r1 = rnd(osh, emu.tdtype, g)
    r2 = rnd(osh, emu.tdtype, g)
    for variant in range(4):
        ep = [dict(bias=b, act=0.2), dict(bias=b, r1=r1, s1=0.2, r2=r2, s2=0.2), dict(mask=r1, mask_slope=0.2), dict(act=0.01, accumulate=True)][variant]
        y_ref = rnd(osh, emu.tdtype, torch.Generator().manual_seed(5))
        y = y_ref.clone().cuda()
        emu.conv_fwd(cv, x, w, y_ref, **ep)
        epg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ep.items()}
        hip.conv_fwd(cv, x.cuda(), w.cuda(), y, **epg)
        close(y, y_ref, dtype, f"fwd {cfg} variant {variant}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_dgrad(dtype, cfg):
    N, H, W, ci, co, st, ps = cfg
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(2)
    cv = Conv(N, H, W, ci, co, st, ps)
    dy = rnd(emu.out_shape(cv), emu.tdtype, g)
    wd = rnd((co * 9 * ci,), emu.tdtype, g, 0.1)
    mask = rnd((N, H, W, ci), emu.tdtype, g)
    # mask_c0 / mask_last: the mask for the upper channels only, applied after the accumulate (dense-block data gradients)
    c_half = (ci // 2) // 16 * 16
    eps = [dict(), dict(mask=mask, mask_slope=0.2), dict(accumulate=True),
           dict(accumulate=True, mask=mask, mask_slope=0.2, mask_c0=c_half, mask_last=True),
           dict(mask=mask, mask_slope=0.2, mask_c0=c_half)]
    if ci >= 256:
        eps.append(dict(accumulate=True, mask=mask, mask_slope=0.2, mask_c0=ci - 128, mask_last=True))
    for ep in eps:
        dx_ref = rnd((N, H, W, ci), emu.tdtype, torch.Generator().manual_seed(6))
        dx = dx_ref.clone().cuda()
        emu.conv_dgrad(cv, dy, wd, dx_ref, **ep)
        hip.conv_dgrad(cv, dy.cuda(), wd.cuda(), dx, **{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ep.items()})
        close(dx, dx_ref, dtype, f"dgrad {cfg} {list(ep)}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_wgrad(dtype, cfg):
    N, H, W, ci, co, st, ps = cfg
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(3)
    cv = Conv(N, H, W, ci, co, st, ps)
    x = rnd((N, H, W, ci), emu.tdtype, g)
    dy = rnd(emu.out_shape(cv), emu.tdtype, g)
    dw_ref = torch.randn(co * 9 * ci, generator=g)
    dw = dw_ref.clone().cuda()
    # bias gradient (accumulated into db like dw): a separate column-sum pass, or fused into the wide weight-gradient kernel
    db_ref = None if ps else torch.randn(co, generator=g)
    db = None if ps else db_ref.clone().cuda()
    emu.conv_wgrad(cv, x, dy, dw_ref, db=db_ref)
    hip.conv_wgrad(cv, x.cuda(), dy.cuda(), dw, db=db)
    a, b = dw.cpu(), dw_ref
    scale = float(b.abs().max())
    tol = 1e-5 if dtype == "f32" else 1e-4   # inputs are identical bf16 values; accumulation is fp32 in both
    assert (a - b).abs().max().item() <= tol * scale * 8, (cfg, (a - b).abs().max().item(), scale)
    if not ps:
        assert (db.cpu() - db_ref).abs().max().item() <= tol * 8 * float(db_ref.abs().max()), cfg


@pytest.mark.parametrize("N,H,W,n", [(2, 32, 32, 5), (1, 16, 64, 3), (1, 8, 32, 1)])
def test_conv_wgrad_dense_block(N, H, W, n):
    """dg_conv3x3_wgrad_dense: the weight / bias gradients of all convs of a dense block (conv k: k*128 -> 128 channels on a shared
    slab, adjoints stacked in a second slab) in one launch == one dg_conv3x3_wgrad per conv == the oracle."""
    hip, emu = pair("bf16")
    g = torch.Generator().manual_seed(21)
    F = 128
    slab = rnd((N, H, W, n * F), emu.tdtype, g)
    us = rnd((N, H, W, n * F), emu.tdtype, g)
    cvs = [Conv(N, H, W, (k + 1) * F, F) for k in range(n)]
    dws_ref = [torch.randn(F * 9 * (k + 1) * F, generator=g) for k in range(n)]
    dbs_ref = [torch.randn(F, generator=g) for _ in range(n)]
    dws = [t.clone().cuda() for t in dws_ref]; dbs = [t.clone().cuda() for t in dbs_ref]
    dws1 = [t.clone().cuda() for t in dws_ref]; dbs1 = [t.clone().cuda() for t in dbs_ref]
    emu.conv_wgrad_dense(cvs, slab, us, dws_ref, dbs_ref)
    hip.conv_wgrad_dense(cvs, slab.cuda(), us.cuda(), dws, dbs)
    assert hip.lib.dg_last_conv_kernels is not None
    for k in range(n):
        hip.conv_wgrad(cvs[k], slab.cuda()[..., :(k + 1) * F], us.cuda()[..., k * F:(k + 1) * F], dws1[k], db=dbs1[k])
        scale = float(dws_ref[k].abs().max())
        assert (dws[k].cpu() - dws_ref[k]).abs().max().item() <= 8e-4 * scale, k
        assert (dws[k] - dws1[k]).abs().max().item() <= 1e-4 * scale, k
        assert (dbs[k].cpu() - dbs_ref[k]).abs().max().item() <= 8e-4 * float(dbs_ref[k].abs().max()), k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("F,n", [(16, 5), (128, 5), (32, 3)])
def test_repack_dense_dgrad(dtype, F, n):
    """dg_repack_dense_dgrad: the stacked data-gradient packs of a dense block in one launch == gather + dg_repack_conv_weights
    per slab slice (the emulation's definition), bit for bit."""
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(41)
    masters = [torch.randn(F * 9 * (k + 1) * F, generator=g) for k in range(n)]
    total = 9 * F * F * n * (n + 1) // 2
    ref = torch.zeros(total, dtype=emu.tdtype)
    emu.repack_dense(masters, ref, F)
    dst = torch.zeros(total, dtype=emu.tdtype).cuda()
    hip.repack_dense([m.cuda() for m in masters], dst, F)
    assert torch.equal(dst.cpu(), ref)


def test_conv_slab_views_f32():
    """channel-slice views of a wider slab as input and output (dense-block layout)."""
    hip, emu = pair("f32")
    g = torch.Generator().manual_seed(4)
    slab = torch.randn(2, 8, 8, 80, generator=g)
    w = torch.randn(16 * 9 * 32, generator=g) * 0.1
    cv = Conv(2, 8, 8, 32, 16)
    ref = slab.clone()
    emu.conv_fwd(cv, ref[..., :32], w, ref[..., 32:48], act=0.01)
    dev = slab.clone().cuda()
    hip.conv_fwd(cv, dev[..., :32], w.cuda(), dev[..., 32:48], act=0.01)
    close(dev, ref, "f32", "slab")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("B,K,O", [(4, 8192, 112), (2, 512, 112), (32, 4096, 112), (4, 128, 16), (33, 1024, 16),
                                   (2, 4194304, 112)])    # FC1 of BASELINE configs[1]: 1024 ch x 64 x 64 columns (critic.py:95)
def test_linear(dtype, B, K, O):
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(7)
    x = rnd((B, K), emu.tdtype, g)
    w = rnd((O, K), emu.tdtype, g, 0.05)
    y_ref = torch.zeros(B, 128)
    emu.linear_fwd(x, w, y_ref)
    y = torch.zeros(B, 128).cuda()
    hip.linear_fwd(x.cuda(), w.cuda(), y)
    close(y, y_ref, "f32" if dtype == "f32" else "bf16", "linear_fwd")
    dy = torch.randn(B, 128, generator=g)
    mask = rnd((B, K), emu.tdtype, g)
    for out_dt in ([torch.float32] if dtype == "f32" else [torch.float32, torch.bfloat16]):
        dx_ref = torch.zeros(B, K, dtype=out_dt)
        emu.linear_dx(dy, w, dx_ref, mask=mask, mask_slope=0.2)
        dx = torch.zeros(B, K, dtype=out_dt).cuda()
        hip.linear_dx(dy.cuda(), w.cuda(), dx, mask=mask.cuda(), mask_slope=0.2)
        close(dx, dx_ref, dtype, "linear_dx")
    if B <= 64:
        dw_ref = torch.randn(O, K, generator=g)
        dw = dw_ref.clone().cuda()
        emu.linear_dw(dy, x, dw_ref)
        hip.linear_dw(dy.cuda(), x.cuda(), dw)
        close(dw, dw_ref, "f32", "linear_dw")
    if B <= 128 and O <= 112:           # the single-sweep form: accumulate and write modes, padded row strides
        for acc in (True, False):
            dw_ref = torch.randn(O, K, generator=g)
            dw = dw_ref.clone().cuda()
            emu.linear_dw_wide(dy, x, dw_ref, accumulate=acc)
            hip.linear_dw_wide(dy.cuda(), x.cuda(), dw, accumulate=acc)
            close(dw, dw_ref, "f32", f"linear_dw_wide acc={acc}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("B,K,O", [(40, 4352, 100), (16, 4096, 112), (96, 8192, 7)])
def test_linear_dx_wide_kernel_shapes(dtype, B, K, O):
    """the 32-rows-per-pass input-gradient kernel (K >= 4096, B >= 16): a ragged last row group, an odd number of weight rows,
    K not a multiple of the 1024-column workgroup span, with and without the LeakyReLU' mask."""
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(9)
    w = rnd((O, K), emu.tdtype, g, 0.1)
    dy = torch.zeros(B, 128); dy[:, :O] = torch.randn(B, O, generator=g)
    mask = rnd((B, K), emu.tdtype, g)
    for m in (None, mask):
        ref = torch.zeros(B, K, dtype=emu.tdtype)
        emu.linear_dx(dy[:, :O], w, ref, mask=m, mask_slope=0.2)
        dx = torch.full((B, K), 3.0, dtype=emu.tdtype).cuda()
        hip.linear_dx(dy.cuda()[:, :O], w.cuda(), dx, mask=None if m is None else m.cuda(), mask_slope=0.2)
        close(dx, ref, dtype, f"linear_dx wide B={B} K={K} O={O} mask={m is not None}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_linear_dw_wide_three_passes(dtype):
    """96 concatenated rows (three passes of batch 32) x 100 real outputs in a 112-row gradient, K with a ragged last
    256-chunk: equals three accumulating dg_linear_dw calls (what the engine did per pass)."""
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(77)
    B, O, K = 96, 112, 256 * 37 + 136
    x = rnd((B, K), emu.tdtype, g)
    dy = torch.zeros(B, 128)
    dy[:, :100] = torch.randn(B, 100, generator=g)
    ref = torch.zeros(O, K)
    for p in range(3):
        emu.linear_dw(dy[32 * p:32 * p + 32], x[32 * p:32 * p + 32], ref)
    dw = torch.full((O, K), float("nan")).cuda()          # write mode must not read the destination
    hip.linear_dw_wide(dy.cuda()[:, :O], x.cuda(), dw, accumulate=False)
    close(dw, ref, "f32", "linear_dw_wide 3x32")
    three = torch.zeros(O, K).cuda()
    for p in range(3):
        hip.linear_dw(dy.cuda()[32 * p:32 * p + 32], x.cuda()[32 * p:32 * p + 32], three)
    close(dw, three, "f32", "linear_dw_wide vs 3 x linear_dw")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_elementwise_and_reductions(dtype):
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(8)
    dt = emu.tdtype
    B, H, W, Cc = 3, 24, 40, 16
    a = rnd((B, H, W, Cc), dt, g); b = rnd((B, H, W, Cc), dt, g)
    slab = rnd((B, H, W, 48), dt, g)
    # mask_mul on a slab slice
    ref = slab.clone(); emu.mask_mul(ref[..., 16:32], a, 0.01)
    dev = slab.clone().cuda(); hip.mask_mul(dev[..., 16:32], a.cuda(), 0.01)
    close(dev, ref, dtype, "mask_mul")
    # axpby (incl. in-place and copy)
    ref = slab.clone(); emu.axpby(ref[..., :16], a, 0.2, b, 1.0)
    dev = slab.clone().cuda(); hip.axpby(dev[..., :16], a.cuda(), 0.2, b.cuda(), 1.0)
    close(dev, ref, dtype, "axpby")
    ref = torch.zeros_like(a); emu.axpby(ref, a, 0.2)
    dev = torch.zeros_like(a).cuda(); hip.axpby(dev, a.cuda(), 0.2)
    close(dev, ref, dtype, "axpby copy")
    # gp interp / sumsq / finish / scale
    alpha = torch.rand(B, generator=g)
    ref = torch.zeros_like(a); emu.gp_interp(a, b, alpha, ref)
    dev = torch.zeros_like(a).cuda(); hip.gp_interp(a.cuda(), b.cuda(), alpha.cuda(), dev)
    close(dev, ref, dtype, "gp_interp")
    small = (a.float() * 1e-3).to(dt)
    ss_ref = torch.zeros(B); emu.sumsq_rows(small, ss_ref)
    ss = torch.zeros(B).cuda(); hip.sumsq_rows(small.cuda(), ss)
    assert torch.allclose(ss.cpu(), ss_ref, rtol=1e-5)
    coef_ref, sc_ref = torch.zeros(B), torch.zeros(1)
    emu.gp_finish(ss_ref, B, 2 * B, 10.0, 10.0, coef_ref, sc_ref)
    coef, sc = torch.zeros(B).cuda(), torch.zeros(1).cuda()
    hip.gp_finish(ss, B, 2 * B, 10.0, 10.0, coef, sc)
    assert torch.allclose(coef.cpu(), coef_ref, rtol=1e-5) and torch.allclose(sc.cpu(), sc_ref, rtol=1e-5)
    ref = torch.zeros_like(a); emu.scale_rows(a, coef_ref, ref)
    dev = torch.zeros_like(a).cuda(); hip.scale_rows(a.cuda(), coef, dev)
    close(dev, ref, dtype, "scale_rows")
    # L1 with gradient and addend
    acc_ref, gr_ref = torch.zeros(1), torch.zeros_like(a)
    emu.l1(a, b, acc_ref, grad=gr_ref, grad_scale=0.37, addend=slab[..., :16])
    acc, gr = torch.zeros(1).cuda(), torch.zeros_like(a).cuda()
    hip.l1(a.cuda(), b.cuda(), acc, grad=gr, grad_scale=0.37, addend=slab.cuda()[..., :16])
    assert torch.allclose(acc.cpu(), acc_ref, rtol=1e-5)
    close(gr, gr_ref, dtype, "l1 grad")
    # colsum: NHWC, 2-D fp32, pixel-shuffled
    db_ref = torch.randn(Cc, generator=g); db = db_ref.clone().cuda()
    emu.colsum(a, db_ref); hip.colsum(a.cuda(), db)
    assert torch.allclose(db.cpu(), db_ref, rtol=1e-4, atol=1e-3)
    d2 = torch.randn(5, 128, generator=g)
    db_ref = torch.zeros(128); db = torch.zeros(128).cuda()
    emu.colsum(d2, db_ref); hip.colsum(d2.cuda(), db)
    assert torch.allclose(db.cpu(), db_ref, rtol=1e-5, atol=1e-5)
    db_ref = torch.zeros(4 * Cc); db = torch.zeros(4 * Cc).cuda()
    emu.colsum_ps(a, db_ref); hip.colsum_ps(a.cuda(), db)
    assert torch.allclose(db.cpu(), db_ref, rtol=1e-4, atol=1e-3)
    # head helpers
    inp = torch.randn(5, 128, generator=g); bias = torch.randn(128, generator=g)
    for out_dt in (torch.float32, dt):
        msk = rnd((5, 112), out_dt, g)
        ref = torch.zeros(5, 112, dtype=out_dt); emu.bias_act(inp, bias, ref, act=0.2)
        dev = torch.zeros(5, 112, dtype=out_dt).cuda(); hip.bias_act(inp.cuda(), bias.cuda(), dev, act=0.2)
        close(dev, ref, dtype, "bias_act")
        ref = torch.zeros(5, 112, dtype=out_dt); emu.bias_act(inp, None, ref, mask=msk, mask_slope=0.2)
        dev = torch.zeros(5, 112, dtype=out_dt).cuda(); hip.bias_act(inp.cuda(), None, dev, mask=msk.cuda(), mask_slope=0.2)
        close(dev, ref, dtype, "bias_act mask")
    out_ref, out = torch.zeros(1), torch.zeros(1).cuda()
    emu.sum_strided(inp, 5, 128, 0.2, out_ref); hip.sum_strided(inp.cuda(), 5, 128, 0.2, out)
    assert torch.allclose(out.cpu(), out_ref, rtol=1e-5)
    buf = torch.zeros(5, 16).cuda(); hip.fill_col(buf, 0, -0.25)
    assert (buf[:, 0] == -0.25).all() and (buf[:, 1:] == 0).all()


def test_adam_and_layout():
    hip, emu = pair("bf16")
    g = torch.Generator().manual_seed(9)
    n = 4096 + 64
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-2
    m, v = torch.zeros(n), torch.zeros(n)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    pd, gd, md, vd = p.cuda(), gr.cuda(), m.cuda(), v.cuda()
    sh = torch.zeros(n, dtype=torch.bfloat16).cuda()
    for step in (1, 2, 3):
        emu.adam(pr, gr, mr, vr, None, 2.5e-4, 0.9, 0.99, 1e-8, step)
        hip.adam(pd, gd, md, vd, sh, 2.5e-4, 0.9, 0.99, 1e-8, step)
    assert torch.allclose(pd.cpu(), pr, rtol=1e-6, atol=1e-7)
    assert torch.allclose(vd.cpu(), vr, rtol=1e-5, atol=1e-12)
    assert torch.equal(sh.cpu(), pd.cpu().to(torch.bfloat16))
    x = torch.randn(2, 6, 10, 12, generator=g)
    for dt in ("f32", "bf16"):
        h, e = pair(dt)
        ref = torch.zeros(2, 10, 12, 16, dtype=e.tdtype); e.nchw_to_nhwc(x, ref)
        dev = torch.ones(2, 10, 12, 16, dtype=e.tdtype).cuda(); h.nchw_to_nhwc(x.cuda(), dev)
        assert torch.equal(dev.cpu(), ref)
        back = torch.zeros(2, 6, 10, 12).cuda(); h.nhwc_to_nchw(dev, back)
        assert torch.equal(back.cpu(), ref[..., :6].float().permute(0, 3, 1, 2))
        m_ = torch.randn(32 * 9 * 16, generator=g)
        for kind in (0, 1):
            ref = torch.zeros(32 * 9 * 16, dtype=e.tdtype); e.repack(m_, ref, 32, 16, kind)
            dev = torch.zeros(32 * 9 * 16, dtype=e.tdtype).cuda(); h.repack(m_.cuda(), dev, 32, 16, kind)
            assert torch.equal(dev.cpu(), ref)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,co,cin_real", [(2, 32, 32, 128, 2), (1, 64, 32, 64, 2), (1, 16, 64, 128, 1), (1, 8, 12, 32, 2)])
def test_conv_wgrad_small_cin_with_bias(dtype, N, H, W, co, cin_real):
    """layers with <= 2 real input channels (critic features.0): im2col weight-gradient + fused bias gradient."""
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(11)
    cv = Conv(N, H, W, 16, co, 1, False, cin_real=cin_real)
    x = torch.zeros(N, H, W, 16, dtype=emu.tdtype)
    x[..., :cin_real] = rnd((N, H, W, cin_real), emu.tdtype, g)
    dy = rnd(emu.out_shape(cv), emu.tdtype, g)
    dw_ref = torch.randn(co * 9 * 16, generator=g); db_ref = torch.randn(co, generator=g)
    dw, db = dw_ref.clone().cuda(), db_ref.clone().cuda()
    emu.conv_wgrad(cv, x, dy, dw_ref, db=db_ref)
    hip.conv_wgrad(cv, x.cuda(), dy.cuda(), dw, db=db)
    tol = 1e-5 if dtype == "f32" else 1e-4
    assert (dw.cpu() - dw_ref).abs().max().item() <= tol * 8 * float(dw_ref.abs().max())
    assert (db.cpu() - db_ref).abs().max().item() <= tol * 8 * float(db_ref.abs().max())
    if W % 32 == 0:        # the same gradient from the COMPACT [N, H, W, 2] form of x (what the train engine feeds the first layer)
        dw2, db2 = torch.zeros(co * 9 * 16).cuda(), torch.zeros(co).cuda()
        dw1, db1 = torch.zeros(co * 9 * 16).cuda(), torch.zeros(co).cuda()
        hip.conv_wgrad(cv, x.cuda(), dy.cuda(), dw1, db=db1)
        hip.conv_wgrad(cv, x[..., :2].contiguous().cuda(), dy.cuda(), dw2, db=db2)
        scale = float(dw1.abs().max())
        assert (dw1 - dw2).abs().max().item() <= 1e-5 * scale and (db1 - db2).abs().max().item() <= 1e-5 * float(db1.abs().max())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_compact_two_channel_fields(dtype):
    """dg_gp_interp_c2 / dg_scale_rows_c2: the interpolate, compact copies of its inputs and the penalty's scaled gradient as
    [B, H, W, 2] from fields stored 16 channels wide; and the first-layer forward from either layout gives identical bits."""
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(31)
    B, H, W = 3, 32, 64
    real = torch.zeros(B, H, W, 16, dtype=emu.tdtype); real[..., :2] = rnd((B, H, W, 2), emu.tdtype, g)
    fake = torch.zeros(B, H, W, 16, dtype=emu.tdtype); fake[..., :2] = rnd((B, H, W, 2), emu.tdtype, g)
    alpha = torch.rand(B, generator=g)
    full = torch.zeros_like(real); emu.gp_interp(real, fake, alpha, full)
    xc, rc, fc = (torch.full((B, H, W, 2), 7.0, dtype=emu.tdtype).cuda() for _ in range(3))
    hip.gp_interp(real.cuda(), fake.cuda(), alpha.cuda(), xc, rc, fc)
    assert torch.equal(rc.cpu(), real[..., :2]) and torch.equal(fc.cpu(), fake[..., :2])
    close(xc, full[..., :2], dtype, "gp_interp_c2")
    dev_full = torch.zeros_like(real).cuda(); hip.gp_interp(real.cuda(), fake.cuda(), alpha.cuda(), dev_full)
    assert torch.equal(xc, dev_full[..., :2]), "compact and padded interpolates must round identically"
    xc2 = torch.zeros(B, H, W, 2, dtype=emu.tdtype).cuda()
    hip.gp_interp(real.cuda(), fake.cuda(), alpha.cuda(), xc2)          # without the copies
    assert torch.equal(xc2, xc)
    coef = torch.randn(B, generator=g)
    ref = torch.zeros_like(real); emu.scale_rows(real, coef, ref)
    vc = torch.zeros(B, H, W, 2, dtype=emu.tdtype).cuda()
    hip.scale_rows(real.cuda(), coef.cuda(), vc)
    close(vc, ref[..., :2], dtype, "scale_rows_c2")
    # emulator: same compact semantics (the CPU engine tests run the compact path through it)
    e_xc, e_rc = torch.zeros(B, H, W, 2, dtype=emu.tdtype), torch.zeros(B, H, W, 2, dtype=emu.tdtype)
    emu.gp_interp(real, fake, alpha, e_xc, e_rc, None)
    assert torch.equal(e_xc, full[..., :2]) and torch.equal(e_rc, real[..., :2])
    cv = Conv(B, H, W, 16, 128, 1, False, cin_real=2)
    w = torch.zeros(128, 9, 16, dtype=emu.tdtype); w[..., :2] = rnd((128, 9, 2), emu.tdtype, g, 0.3)
    b = torch.randn(128, generator=g)
    y1, y2 = hip.zeros(*hip.out_shape(cv)), hip.zeros(*hip.out_shape(cv))
    hip.conv_fwd(cv, real.cuda(), w.reshape(-1).cuda(), y1, bias=b.cuda(), act=0.2)
    hip.conv_fwd(cv, rc, w.reshape(-1).cuda(), y2, bias=b.cuda(), act=0.2)
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,co,cin_real", [(2, 32, 32, 128, 2), (1, 20, 12, 64, 2), (3, 16, 16, 16, 1), (1, 40, 24, 256, 2),
                                               (1, 24, 32, 128, 2),      # barrier-free kernel, linear group order (height not a multiple of 16)
                                               (2, 48, 64, 256, 1)])     # ... tiled order, two channel tiles, one real input channel
def test_conv_fwd_small_cin(dtype, N, H, W, co, cin_real):
    """stride-1 forward with <= 2 real input channels: im2col kernel (bias+act, mask, residual epilogues)."""
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(12)
    cv = Conv(N, H, W, 16, co, 1, False, cin_real=cin_real)
    x = torch.zeros(N, H, W, 16, dtype=emu.tdtype)
    x[..., :cin_real] = rnd((N, H, W, cin_real), emu.tdtype, g)
    w = torch.zeros(co, 9, 16, dtype=emu.tdtype)
    w[..., :cin_real] = rnd((co, 9, cin_real), emu.tdtype, g, 0.3)
    w = w.reshape(-1)
    b = torch.randn(co, generator=g)
    msk = rnd(emu.out_shape(cv), emu.tdtype, g)
    for ep in (dict(bias=b, act=0.2), dict(mask=msk, mask_slope=0.2), dict(bias=b, r1=msk, s1=0.5)):
        y_ref = torch.zeros(emu.out_shape(cv), dtype=emu.tdtype)
        y = torch.ones(emu.out_shape(cv), dtype=emu.tdtype).cuda()
        emu.conv_fwd(cv, x, w, y_ref, **ep)
        hip.conv_fwd(cv, x.cuda(), w.cuda(), y, **{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ep.items()})
        close(y, y_ref, dtype, f"im2col fwd {list(ep)}")


# ------------------------------------------------------------------ 1-bit LeakyReLU' masks (dg_epilogue.mask_bits / out_bits)
BITS = [
    # N, H, W, Cin, Cout, stride, cin_real
    (1, 32, 32, 128, 128, 1, 0),       # halo kernel
    (2, 24, 40, 128, 256, 1, 0),       # ragged tiles, two channel tiles
    (1, 32, 48, 128, 192, 2, 0),       # stride 2: forward on the halo kernel; data gradient = 3 halo classes + 1 per-tap class
    (1, 64, 32, 256, 128, 2, 0),
    (2, 32, 32, 16, 128, 1, 2),        # <= 2 real input channels: im2col forward
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", BITS)
def test_mask_bits_forward_and_dgrad(dtype, cfg):
    """out_bits of a forward launch == (stored activation > 0) in the documented packing; a data gradient masked by the
    bits is bit-identical to the same launch masked by the activation tensor itself."""
    N, H, W, ci, co, st, cr = cfg
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(5)
    cv = Conv(N, H, W, ci, co, st, False, cin_real=cr)
    x = rnd((N, H, W, ci), emu.tdtype, g)
    if cr:
        x[..., cr:] = 0
    w = rnd((co * 9 * ci,), emu.tdtype, g, 0.1)
    b = torch.randn(co, generator=g)
    y = hip.zeros(*hip.out_shape(cv))
    bits = hip.zeros(*hip.bits_shape(y.shape), dtype=torch.int16)
    hip.conv_fwd(cv, x.cuda(), w.cuda(), y, bias=b.cuda(), act=0.2, out_bits=bits)
    want = torch.zeros(*bits.shape, dtype=torch.int16)
    emu._pack_bits(y.float().cpu() > 0, want)
    assert torch.equal(bits.cpu(), want)
    # the same forward as a masked launch (the penalty's tangent forward): bits vs the activation as the mask
    t = rnd((N, H, W, ci), emu.tdtype, g)
    if cr:
        t[..., cr:] = 0
    o1, o2 = hip.zeros(*y.shape), hip.zeros(*y.shape)
    hip.conv_fwd(cv, t.cuda(), w.cuda(), o1, mask=y, mask_slope=0.2)
    hip.conv_fwd(cv, t.cuda(), w.cuda(), o2, mask_bits=bits, mask_slope=0.2)
    assert torch.equal(o1, o2)
    # data gradient of a layer whose INPUT activation has `ci` channels: mask over dx
    if ci >= 128:
        act_in = rnd((N, H, W, ci), emu.tdtype, g).cuda()
        bits_in = hip.zeros(*hip.bits_shape(act_in.shape), dtype=torch.int16)
        packed = torch.zeros(*bits_in.shape, dtype=torch.int16)
        emu._pack_bits(act_in.float().cpu() > 0, packed)
        bits_in.copy_(packed)
        dy = rnd(tuple(y.shape), emu.tdtype, g).cuda()
        wd = rnd((co * 9 * ci,), emu.tdtype, g, 0.1).cuda()
        d1, d2 = hip.zeros(N, H, W, ci), hip.zeros(N, H, W, ci)
        hip.conv_dgrad(cv, dy, wd, d1, mask=act_in, mask_slope=0.2)
        hip.conv_dgrad(cv, dy, wd, d2, mask_bits=bits_in, mask_slope=0.2)
        assert torch.equal(d1, d2)


def test_mask_bits_rejected_for_narrow_layers():
    hip, emu = pair("bf16")
    cv = Conv(1, 16, 16, 64, 64, 1, False)
    y = hip.zeros(1, 16, 16, 64)
    with pytest.raises((AssertionError, RuntimeError)):
        hip.conv_fwd(cv, hip.zeros(1, 16, 16, 64), hip.zeros(64 * 9 * 64), y, act=0.2,
                     out_bits=torch.zeros(1, 16, 16, 1, 4, dtype=torch.int16, device="cuda"))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_small_cout_backward_on_first_layer_kernels(dtype):
    """A stride-1 layer with <= 2 real OUTPUT channels (generator conv3.2, generator.py:80): its data gradient as the forward conv
    of dy with the mirrored-tap pack (dg_repack_conv_weights kind 2 -> im2col forward kernel, with the LeakyReLU' mask epilogue),
    its weight gradient as the swapped-role launch folded back by dg_wgrad_unswap -- both against the ordinary paths' oracle."""
    hip, emu = pair(dtype)
    g = torch.Generator().manual_seed(9)
    N, H, W, F_, npad, nreal = 2, 32, 64, 128, 16, 2
    cv = Conv(N, H, W, F_, npad, 1, False)                     # the layer as it is: 128 -> 16 (2 real)
    cvb = Conv(N, H, W, npad, F_, 1, False, cin_real=nreal)    # its backward as a first-layer-shaped conv
    master = torch.zeros(npad, 9, F_)
    master[:nreal] = torch.randn(nreal, 9, F_, generator=g) * 0.1
    dy = torch.zeros(N, H, W, npad)
    dy[..., :nreal] = torch.randn(N, H, W, nreal, generator=g)
    dy = dy.to(emu.tdtype)
    x = rnd((N, H, W, F_), emu.tdtype, g)
    mask = rnd((N, H, W, F_), emu.tdtype, g)
    # reference: ordinary data gradient / weight gradient through the oracle ops
    wd = torch.zeros(npad * 9 * F_, dtype=emu.tdtype)
    emu.repack(master.reshape(-1), wd, npad, F_, 1)
    dx_ref = torch.zeros(N, H, W, F_, dtype=emu.tdtype)
    emu.conv_dgrad(cv, dy, wd, dx_ref, mask=mask, mask_slope=0.01)
    dw_ref = torch.zeros(npad * 9 * F_)
    emu.conv_wgrad(cv, x, dy, dw_ref)
    # native: mirrored pack + forward kernel; swapped wgrad + unswap
    wm = torch.zeros(npad * 9 * F_, dtype=emu.tdtype).cuda()
    hip.repack(master.reshape(-1).cuda(), wm, npad, F_, 2)
    dx = torch.zeros(N, H, W, F_, dtype=emu.tdtype).cuda()
    hip.conv_fwd(cvb, dy.cuda(), wm, dx, mask=mask.cuda(), mask_slope=0.01)
    assert hip.lib.dg_last_conv_kernels() == 16                 # im2col kernel
    close(dx, dx_ref, dtype, "conv3.2 dgrad as forward")
    tmp = torch.zeros(npad * 9 * F_).cuda()
    hip.conv_wgrad(cvb, dy.cuda(), x.cuda(), tmp)
    dw = torch.zeros(npad * 9 * F_).cuda()
    hip.wgrad_unswap(tmp, dw, npad, F_)
    a, b = dw.cpu(), dw_ref
    tol = 1e-5 if dtype == "f32" else 1e-4
    assert (a - b).abs().max().item() <= tol * 8 * float(b.abs().max()), (a - b).abs().max().item()
