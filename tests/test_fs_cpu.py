"""Frequency-separation variant, host side on the torch-CPU op emulation against the oracle's
restatement of wasserstein_fs.py.  Parity status of this variant: UNPINNED (the reference module is not importable)."""
import pytest
import torch

from downgan_amd import synthetic
from downgan_amd.engine import HyperParams, TrainEngineFS
from downgan_amd.layout import nchw_to_nhwc_padded
from oracle import ref_step
from oracle.emu_ops import EmuOps


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-30)


def make(B, S, F_, cin, nrb, dtype_o=torch.float32):
    ops = EmuOps("f32")
    eng = TrainEngineFS(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
    pg = synthetic.generator_params(F_, cin, 2, nrb)
    pc = synthetic.critic_params(F_, 8 * S, 2)
    eng.G.load_state_dict(pg); eng.C.load_state_dict(pc)
    coarse, fine = synthetic.tiles(B, cin, S, mask_channel=(2 if cin > 2 else None))
    tc, tf = torch.from_numpy(coarse), torch.from_numpy(fine)
    xc, xf = nchw_to_nhwc_padded(tc, eng.G.cin_p, ops.tdtype), nchw_to_nhwc_padded(tf, eng.G.np_p, ops.tdtype)
    orc = ref_step.OracleTrainerFS({k: torch.from_numpy(v).to(dtype_o) for k, v in pg.items()},
                                   {k: torch.from_numpy(v).to(dtype_o) for k, v in pc.items()},
                                   ref_step.HP(batch_size=B), num_res_blocks=nrb)
    return eng, orc, tc, tf, xc, xf


def test_lowpass_is_a_replicated_box_mean_and_adjoint_identity():
    x = torch.arange(2 * 1 * 7 * 9.0).view(2, 1, 7, 9)
    lo = ref_step.lowpass(x)
    assert lo.shape == x.shape
    assert abs(float(lo[0, 0, 3, 4]) - float(x[0, 0, 1:6, 2:7].mean())) < 1e-5            # interior: plain 5x5 mean
    corner = torch.stack([x[0, 0, min(max(i, 0), 6), min(max(j, 0), 8)] for i in range(-2, 3) for j in range(-2, 3)]).mean()
    assert abs(float(lo[0, 0, 0, 0]) - float(corner)) < 1e-5                              # border: replicated samples
    ops = EmuOps("f32")
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(2, 7, 9, 8, generator=g), torch.randn(2, 7, 9, 8, generator=g)
    la, ltb = torch.empty_like(a), torch.empty_like(b)
    ops.lowpass5(a, low=la)
    ops.lowpass5_adjoint(b, ltb)
    assert rel(float((la * b).sum()), float((a * ltb).sum())) < 1e-5                     # <L a, b> == <a, L^T b>


@pytest.mark.parametrize("cfg", [(2, 16, 16, 2, 1), (2, 16, 16, 6, 2)])
def test_fs_iterations_match_oracle(cfg):
    B, S, F_, cin, nrb = cfg
    eng, orc, tc, tf, xc, xf = make(B, S, F_, cin, nrb)
    _, o64, *_ = make(B, S, F_, cin, nrb, torch.float64)
    alpha = torch.from_numpy(synthetic.alpha(B, 0))
    ref, cg = orc.critic_iteration(tc, tf, alpha, apply_update=False)
    _, cg64 = o64.critic_iteration(tc.double(), tf.double(), alpha.double(), apply_update=False)
    eng.critic_iteration(xc, xf, alpha, apply_update=False)
    got = eng.read_scalars()
    for k in ("c_real_mean", "c_fake_mean", "gp_ret", "critic_loss"):
        assert rel(got[k], ref[k]) < 1e-4, (k, got[k], ref[k])
    gd = eng.C.grad_dict()
    for k, g in cg64.items():
        err = (gd[k].double() - g).norm() / (g.norm() + 1e-20)
        ref_err = (cg[k].double() - g).norm() / (g.norm() + 1e-20)
        assert err < 1e-5 + 5 * ref_err, (k, float(err), float(ref_err))
    refg, gg32 = orc.generator_iteration(tc, tf, apply_update=False)
    _, gg = o64.generator_iteration(tc.double(), tf.double(), apply_update=False)
    eng.generator_iteration(xc, xf, apply_update=False)
    got = eng.read_scalars(True)
    for k in ("g_loss", "content_loss", "g_c_fake_mean"):
        assert rel(got[k], refg[k]) < 1e-4, (k, got[k], refg[k])
    gd = eng.G.grad_dict()
    for k, g in gg.items():
        err = (gd[k].double() - g).norm() / (g.norm() + 1e-20)
        ref_err = (gg32[k].double() - g).norm() / (g.norm() + 1e-20)
        assert err < 1e-5 + 5 * ref_err, (k, float(err), float(ref_err))


def test_fs_differs_from_the_plain_step():
    """Guards against the variant silently running the parent's iteration."""
    eng, orc, tc, tf, xc, xf = make(2, 16, 16, 2, 1)
    plain = ref_step.OracleTrainer({k: v.detach() for k, v in orc.PG.items()}, {k: v.detach() for k, v in orc.PC.items()},
                                   ref_step.HP(batch_size=2), num_res_blocks=1)
    alpha = torch.from_numpy(synthetic.alpha(2, 0))
    a, _ = orc.critic_iteration(tc, tf, alpha, apply_update=False)
    b, _ = plain.critic_iteration(tc, tf, alpha, apply_update=False)
    assert rel(a["c_real_mean"], b["c_real_mean"]) > 1e-3      # C(real - low(real)) vs C(real)
