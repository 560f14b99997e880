"""The two debugging aids on the GPU (csrc/debug.hip): deterministic reductions (HipOps(deterministic=True) / DG_DETERMINISTIC=1)
and the per-iteration NaN / Inf census (TrainEngine(check_finite=True), the stand-in for the reference's
torch.autograd.set_detect_anomaly(True), DoWnGAN/GAN/wasserstein.py:13)."""
import pytest
import torch

from downgan_amd import synthetic
from downgan_amd.engine import HyperParams, TrainEngine
from downgan_amd.ops import Conv, HipOps

pytestmark = pytest.mark.gpu


def make(B, S, F_, cin, nrb, dtype, **okw):
    ops = HipOps(dtype, **okw)
    ekw = {}
    eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb, **ekw)
    eng.G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb))
    eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
    coarse, fine = synthetic.tiles(B, cin, S)
    xc = ops.zeros(B, S, S, eng.G.cin_p); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
    xf = ops.zeros(B, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
    return ops, eng, xc, xf


def run_steps(cfg, nsteps, graphs=False, **okw):
    B = cfg[0]
    ops, eng, xc, xf = make(*cfg, **okw)
    if graphs:
        eng.enable_graphs(xc, xf)
    out = []
    for step in range(nsteps):
        ran_g = eng.train_step(xc, xf, torch.from_numpy(synthetic.alpha(B, step)).cuda())
        out.append(eng.read_scalars(ran_g))
    state = (eng.C.P.p.clone(), eng.G.P.p.clone(), eng.C.P.g.clone(), eng.G.P.g.clone())
    ops.close()
    return out, state


@pytest.mark.parametrize("cfg", [(4, 16, 16, 2, 16, "f32"),        # BASELINE configs[0] shapes: per-tap / narrow kernels, fp32
                                 (2, 8, 128, 2, 1, "bf16"),         # 128 filters: wide row-of-taps kernel, dense-block launch, FC1 sweep
                                 (2, 16, 32, 6, 2, "bf16")])
def test_deterministic_mode_is_bit_identical_between_runs(cfg):
    """Six train steps (two generator updates), twice: every scalar of every step, the final parameters and the last gradients of
    both networks are BIT-identical -- the split-K weight gradients and the small reductions add their partials in a fixed order."""
    a, sa = run_steps(cfg, 6, deterministic=True)
    b, sb = run_steps(cfg, 6, deterministic=True)
    assert a == b
    for x, y in zip(sa, sb):
        assert torch.equal(x, y)


def test_deterministic_mode_agrees_with_the_atomic_mode():
    """Same numbers up to fp32 re-association (it is the same arithmetic in another summation order), also with a workspace too
    small for the planned split counts (the launches then run with fewer splits).  The sharp check is the FIRST step: every
    gradient of both networks within 1e-5 relative (observed 2e-8 ... 1e-7).  After an Adam update the runs separate like any two
    runs of the atomic mode do -- the update moves noise-level entries by +-lr on the sign of a rounding error -- so the second
    step is only bounded loosely (observed on MI355X: 2e-8 ... 7e-3 depending on the build's rounding, the runs being
    reproducible bit for bit)."""
    cfg = (2, 8, 128, 2, 1, "bf16")

    def grads_per_step(**okw):
        ops, eng, xc, xf = make(*cfg, **okw)
        out = []
        for step in range(2):
            ran_g = eng.train_step(xc, xf, torch.from_numpy(synthetic.alpha(cfg[0], step)).cuda())
            eng.C.P.sync(); eng.G.P.sync()
            out.append((eng.read_scalars(ran_g), eng.C.P.g.clone(), eng.G.P.g.clone()))
        ops.close()
        return out
    ref = grads_per_step()
    for mb in (512, 2):
        det = grads_per_step(deterministic=True, det_workspace_mb=mb)
        for step, (r, d) in enumerate(zip(ref, det)):
            for k in r[0]:
                assert abs(r[0][k] - d[0][k]) <= 2e-4 * max(abs(r[0][k]), 1e-2), (mb, step, k, r[0][k], d[0][k])
            for x, y in zip(r[1:], d[1:]):        # gradients of C and G
                tol = 1e-5 if step == 0 else 3e-2
                assert float((x - y).norm()) <= tol * float(x.norm()) + 1e-12, (mb, step, float((x - y).norm() / x.norm()))


def test_hip_graph_replay_equals_eager_exactly_when_deterministic():
    """In deterministic mode the captured iterations reproduce the eager launches bit for bit (cfg1, 6 steps, fp32); without it
    the two differ by the order of the fp32 atomics (tests/test_step_gpu.py::test_hip_graph_replay_equals_eager)."""
    cfg = (4, 16, 16, 2, 16, "f32")
    eager, se = run_steps(cfg, 6, deterministic=True)
    replay, sp = run_steps(cfg, 6, graphs=True, deterministic=True)
    assert eager == replay
    assert torch.equal(se[0], sp[0]) and torch.equal(se[1], sp[1])


def test_deterministic_wgrad_full_size_layer_matches_and_repeats():
    """A launch at BASELINE configs[1] size (critic features.4: 128 -> 256 at 512^2, batch 4) on the wide kernel: two deterministic
    runs are bit-identical and agree with the atomic mode to fp32 re-association."""
    cv = Conv(4, 512, 512, 128, 256)
    res = {}
    for det in (False, True, True):
        ops = HipOps("bf16", deterministic=det)
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(cv.N, cv.H, cv.W, cv.Cin, device="cuda", generator=g).to(torch.bfloat16)
        dy = (torch.randn(cv.N, cv.H, cv.W, cv.Cout, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
        dw = ops.zeros(cv.Cout * 9 * cv.Cin, dtype=torch.float32)
        db = ops.zeros(cv.Cout, dtype=torch.float32)
        ops.conv_wgrad(cv, x, dy, dw, db=db)
        ops.conv_wgrad(cv, x, dy, dw, db=db)         # accumulates: the second launch adds onto the first
        res.setdefault(det, []).append((dw.clone(), db.clone()))
        ops.close()
    (w0, b0), = res[False]
    (w1, b1), (w2, b2) = res[True]
    assert torch.equal(w1, w2) and torch.equal(b1, b2)
    assert float((w0 - w1).norm()) <= 1e-5 * float(w0.norm()) and float((b0 - b1).norm()) <= 1e-5 * float(b0.norm())


def test_count_nonfinite_kernel():
    ops = HipOps("bf16")
    a = torch.randn(1 << 20, device="cuda")
    b = torch.randn(3, 1000, 7, device="cuda").to(torch.bfloat16)
    c = torch.zeros(5, device="cuda")
    assert ops.count_nonfinite([("a", a), ("b", b), ("c", c)]) == {"a": 0, "b": 0, "c": 0}
    a[12345] = float("nan"); a[-1] = float("-inf"); a[0] = float("inf")
    b[2, 999, 6] = float("nan"); b[0, 0, 0] = float("inf")
    assert ops.count_nonfinite([("a", a), ("b", b), ("c", c)]) == {"a": 3, "b": 2, "c": 0}
    # largest finite values and denormals are finite
    d = torch.tensor([3.4028234e38, -3.4028234e38, 1e-45, 0.0], device="cuda")
    assert ops.count_nonfinite([("d", d), ("db", d.to(torch.bfloat16)[1:])]) == {"d": 0, "db": 1}     # bf16 rounds FLT_MAX to inf


def test_check_finite_raises_on_the_gpu_path():
    ops = HipOps("f32")
    B, S, F_, cin, nrb = 2, 16, 16, 2, 1
    eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb, check_finite=True)
    eng.G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb))
    pc = synthetic.critic_params(F_, 8 * S, 2)
    eng.C.load_state_dict(pc)
    coarse, fine = synthetic.tiles(B, cin, S)
    xc = ops.zeros(B, S, S, eng.G.cin_p); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
    xf = ops.zeros(B, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
    alpha = torch.from_numpy(synthetic.alpha(B, 0)).cuda()
    assert eng.train_step(xc, xf, alpha)                      # clean step passes
    bad = {k: v.copy() for k, v in pc.items()}
    bad["features.6.weight"][3, 2, 1, 1] = float("nan")
    eng.C.load_state_dict(bad)
    with pytest.raises(FloatingPointError, match="critic iteration"):
        eng.critic_iteration(xc, xf, alpha, apply_update=False)


def test_deterministic_workspace_is_shared_and_reference_counted():
    """The mode is process-wide in the library; a process holds several op objects (the losses' cache, the modules, one per engine
    re-bind).  Collecting an EARLIER deterministic object must not switch the mode off under a live one (round-3 advice): the
    workspace is one per process, released by its last holder, and ``ops.deterministic`` reads the library's state."""
    from downgan_amd import ops as ops_mod
    plain = HipOps("bf16", deterministic=False)
    assert not plain.deterministic
    a = HipOps("bf16", deterministic=True, det_workspace_mb=4)
    ws = ops_mod._DET["ws"].data_ptr()
    b = HipOps("bf16", deterministic=True, det_workspace_mb=4)
    assert ops_mod._DET["ws"].data_ptr() == ws and ops_mod._DET["refs"] >= 2
    assert a.deterministic and b.deterministic and plain.deterministic         # the library's switch, whoever asks
    del a                                                                       # the earlier object goes first
    assert b.deterministic and ops_mod._DET["ws"] is not None
    b.close(); b.close()                                                        # idempotent
    assert not b.deterministic and not plain.deterministic
