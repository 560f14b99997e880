"""Frequency-separation kernels and iterations on the GPU through the C ABI, against the oracle (fp32: kernels within
1e-6 of torch's replicate-pad + avg_pool2d; steps within 1e-4 relative of the oracle's restatement of wasserstein_fs.py)."""
import pytest
import torch

from downgan_amd import synthetic
from oracle import ref_step

pytestmark = pytest.mark.gpu


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-30)


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-6), ("bf16", 2e-2)])
@pytest.mark.parametrize("shape", [(2, 7, 9), (1, 64, 48), (2, 128, 128), (1, 3, 2)])
def test_lowpass_and_adjoint_match_oracle(dtype, tol, shape):
    from downgan_amd.ops import HipOps
    o = HipOps(dtype)
    N, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.zeros(N, H, W, 16); x[..., :2] = torch.randn(N, H, W, 2, generator=g)
    u = torch.zeros(N, H, W, 16); u[..., :2] = torch.randn(N, H, W, 2, generator=g)
    xd, ud = x.to(o.tdtype).cuda(), u.to(o.tdtype).cuda()
    low, high, adj = (o.zeros(N, H, W, 16) for _ in range(3))
    o.lowpass5(xd, low=low, high=high)
    o.lowpass5_adjoint(ud, adj)
    xr = xd.float().cpu().permute(0, 3, 1, 2)
    lo = ref_step.lowpass(xr)
    assert float((low.float().cpu().permute(0, 3, 1, 2) - lo).abs().max()) < tol
    assert float((high.float().cpu().permute(0, 3, 1, 2) - (xr - lo)).abs().max()) < tol
    ur = ud.float().cpu().permute(0, 3, 1, 2)
    probe = torch.zeros_like(ur, requires_grad=True)
    (want,) = torch.autograd.grad(ref_step.lowpass(probe), probe, grad_outputs=ur)
    assert float((adj.float().cpu().permute(0, 3, 1, 2) - want).abs().max()) < tol
    assert float(low[..., 2:].abs().max()) == 0.0                                       # padding channels stay zero
    # size-independent property: <L x, u> == <x, L^T u>
    a, b = float((low.float() * ud.float()).sum()), float((xd.float() * adj.float()).sum())
    assert abs(a - b) < (1e-4 if dtype == "f32" else 3e-2) * max(abs(a), 1.0)


@pytest.mark.parametrize("cfg", [(2, 16, 16, 2, 2), (4, 16, 16, 2, 16)])
def test_fs_steps_match_oracle(cfg):
    from downgan_amd.engine import HyperParams, TrainEngineFS
    from downgan_amd.ops import HipOps
    B, S, F_, cin, nrb = cfg
    ops = HipOps("f32")
    eng = TrainEngineFS(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
    pg, pc = synthetic.generator_params(F_, cin, 2, nrb), synthetic.critic_params(F_, 8 * S, 2)
    eng.G.load_state_dict(pg); eng.C.load_state_dict(pc)
    coarse, fine = synthetic.tiles(B, cin, S)
    tc, tf = torch.from_numpy(coarse), torch.from_numpy(fine)
    xc = ops.zeros(B, S, S, eng.G.cin_p); ops.nchw_to_nhwc(tc.cuda(), xc)
    xf = ops.zeros(B, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(tf.cuda(), xf)
    orc = ref_step.OracleTrainerFS({k: torch.from_numpy(v) for k, v in pg.items()}, {k: torch.from_numpy(v) for k, v in pc.items()},
                                   ref_step.HP(batch_size=B), num_res_blocks=nrb)
    for step in range(3):
        alpha = torch.from_numpy(synthetic.alpha(B, step))
        ref = orc.train_step(tc, tf, alpha)
        ran_g = eng.train_step(xc, xf, alpha.cuda())
        got = eng.read_scalars(ran_g)
        for k in ("c_real_mean", "c_fake_mean", "gp_ret", "critic_loss") + (("g_loss", "content_loss") if ran_g else ()):
            assert rel(got[k], ref[k]) < 1e-4, (step, k, got[k], ref[k])


def test_fs_trainer_mirror_runs_bf16_full_tile():
    """WassersteinGANFS on a 2-sample 128->1024 tile in bf16: finite losses, high-pass critic inputs (smoke at full size)."""
    from downgan_amd.GAN.wasserstein_fs import WassersteinGANFS
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    G, C = Generator(128, 1024, 2, 2, num_res_blocks=1), Critic(128, 1024, 2)
    tr = WassersteinGANFS(G, C)
    coarse, fine = synthetic.tiles(2, 2, 128)
    out = tr._critic_train_iteration(torch.from_numpy(coarse), torch.from_numpy(fine), alpha=synthetic.alpha(2, 0))
    out.update(tr._generator_train_iteration(torch.from_numpy(coarse), torch.from_numpy(fine)))
    assert all(v == v and abs(v) < 1e6 for v in out.values()), out
    e = tr._engine
    assert abs(float(e.real_high[..., :2].float().mean())) < 1e-2          # a high-pass field has (almost) no mean
