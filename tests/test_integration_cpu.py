"""Executes the drop-in snippet of INTEGRATION.md (section 1) VERBATIM: the reference's own call pattern of
DoWnGAN/GAN/stage.py:48-72, train.py:15-31 and mlflow_tools/mlflow_epoch.py:53-69 with only the import lines changed.

On the CPU the mirrors' compute backend (downgan_amd.backend.make_ops -> HipOps, which needs a GPU) is patched to the
op-contract emulation of oracle/emu_ops.py, so what is tested here is the mirrors' HOST logic and API surface
(.to / .parameters / torch.optim.Adam construction / DataLoader batches / tensor-returning metrics); the same snippet runs
against the HIP kernels in tests/test_integration_gpu.py."""
import os
import re
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- integration-snippet-begin -->\s*```python\n(.*?)```\s*<!-- integration-snippet-end -->", text, re.S)
    assert m, "INTEGRATION.md lost its integration snippet markers"
    return m.group(1)


def run_snippet(device, n=4, S=16, cin=2, batch=2, seed=3):
    import downgan_amd.config.hyperparams as hp
    from downgan_amd import synthetic
    old = (hp.batch_size, hp.epochs)
    hp.batch_size, hp.epochs = batch, 1
    try:
        coarse, fine = synthetic.tiles(2 * n, cin, S, seed=seed)
        ns = dict(coarse_train=torch.from_numpy(coarse[:n]), fine_train=torch.from_numpy(fine[:n]),
                  coarse_test=torch.from_numpy(coarse[n:]), fine_test=torch.from_numpy(fine[n:]),
                  config=types.SimpleNamespace(device=torch.device(device)))
        exec(compile(snippet(), "INTEGRATION.md#snippet", "exec"), ns)
        return ns
    finally:
        hp.batch_size, hp.epochs = old


def check(ns, batch=2, n=4):
    import math
    hist = ns["history"]
    assert len(hist) == 1 and len(hist[0]) == n // batch                      # one epoch, full batches
    step0, step1 = hist[0][0], hist[0][1]
    assert {"critic_loss", "gp_ret", "g_loss", "content_loss"} <= set(step0) and "g_loss" not in step1   # generator on step % 5 == 0
    assert all(math.isfinite(v) for v in step0.values())
    assert set(ns["d"]) == {"MAE", "MSE", "MSSSIM", "Wass"} and all(len(v) == 1 and math.isfinite(v[0]) for v in ns["d"].values())
    # the epoch loop's own metrics (wasserstein.py:138-170): train means over both batches, test means over the test loader
    (ep,) = ns["trainer"].metrics_log
    assert ep["epoch"] == 0 and set(ep["train"]) == {"MAE", "MSE", "MSSSIM", "Wass"} == set(ep["test"])
    assert all(math.isfinite(v) for v in list(ep["train"].values()) + list(ep["test"].values()))
    assert ns["fake"].shape == ns["real"].shape and ns["fake"].dtype == torch.float32
    # Adam as configured through torch.optim.Adam was adopted by the native trainer, and it did move the parameters
    e = ns["trainer"]._engine
    assert e.adam_hp[id(e.G.P)].lr == ns["G_optimizer"].param_groups[0]["lr"] and e.adam_hp[id(e.C.P)].beta2 == 0.99
    from downgan_amd import synthetic
    init = synthetic.generator_params(ns["coarse_dim_n"], ns["n_covariates"], ns["n_predictands"], 16)
    assert any(not torch.equal(ns["state"][k], torch.from_numpy(v)) for k, v in init.items())
    assert set(ns["state"]) == set(init) and all(tuple(ns["state"][k].shape) == v.shape for k, v in init.items())
    names = [k for k, _ in ns["critic"].named_parameters()]
    assert names[0] == "features.0.weight" and "classifier.2.bias" in names


def test_integration_snippet_runs_verbatim_on_emulated_ops(monkeypatch):
    from downgan_amd import backend
    from downgan_amd.GAN import losses
    from oracle.emu_ops import EmuOps
    monkeypatch.setattr(backend, "make_ops", lambda dtype, device: EmuOps("f32"))
    monkeypatch.setattr(losses, "_ops", {})
    monkeypatch.setattr(losses, "_ms", {})
    torch.set_num_threads(4)
    check(run_snippet("cpu"))


def test_module_surface_without_a_gpu():
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    G = Generator(16, 128, 2, 2, num_res_blocks=1)
    assert G.to(torch.device("cuda:0")) is G and G.eval() is G and G.train() is G and G.cuda() is G
    ps = list(G.parameters())
    assert len(ps) == len(G.state_dict()) and all(isinstance(p, torch.nn.Parameter) for p in ps)
    opt = torch.optim.Adam(Critic(16, 128, 2).parameters(), 1e-3, betas=(0.5, 0.9))
    from downgan_amd.GAN.wasserstein import _adam_config
    from downgan_amd.engine import HyperParams
    cfg = _adam_config(opt, HyperParams())
    assert (cfg.lr, cfg.beta1, cfg.beta2, cfg.eps) == (1e-3, 0.5, 0.9, 1e-8)
    with pytest.raises(NotImplementedError):
        _adam_config(torch.optim.Adam(G.parameters(), 1e-3, weight_decay=0.1), HyperParams())
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):          # no CPU fallback: the forward needs the HIP backend
            G(torch.zeros(1, 2, 16, 16))


def test_epoch_loop_checkpoints_both_networks(monkeypatch, tmp_path):
    """wasserstein.py:178 / mlflow_epoch.py:65-69 without mlflow: <dir>/<Net>/<Net>_<epoch>/state_dict.pth per epoch, in the
    reference's format (loads back into a fresh mirror)."""
    from downgan_amd import backend, synthetic
    from downgan_amd.GAN import losses
    from downgan_amd.GAN.dataloader import NetCDFSR
    from downgan_amd.GAN.wasserstein import WassersteinGAN
    from downgan_amd.checkpoint import load_state_dict
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    from oracle.emu_ops import EmuOps
    import downgan_amd.config.hyperparams as hp
    monkeypatch.setattr(backend, "make_ops", lambda dtype, device: EmuOps("f32"))
    monkeypatch.setattr(losses, "_ops", {})
    monkeypatch.setattr(hp, "batch_size", 2)
    torch.set_num_threads(4)
    coarse, fine = synthetic.tiles(2, 2, 16, seed=5)
    G, C = Generator(16, 128, 2, 2, num_res_blocks=1), Critic(16, 128, 2)
    tr = WassersteinGAN(G, C)
    tr.log_metrics = False
    tr.checkpoint_dir = str(tmp_path)
    dl = torch.utils.data.DataLoader(NetCDFSR(torch.from_numpy(coarse), torch.from_numpy(fine)), batch_size=2)
    tr.train(dl, None, epochs=2)
    assert [m["epoch"] for m in tr.metrics_log] == [0, 1] and "train" not in tr.metrics_log[0]
    for name, net, fresh in (("Critic", C, Critic(16, 128, 2)), ("Generator", G, Generator(16, 128, 2, 2, num_res_blocks=1))):
        path = tmp_path / name / f"{name}_1" / "state_dict.pth"
        assert path.exists() and (tmp_path / name / f"{name}_0" / "state_dict.pth").exists()
        load_state_dict(fresh, str(path))
        a, b = net.state_dict(), fresh.state_dict()
        assert all(torch.equal(a[k], b[k]) for k in a)


def test_ragged_last_batch_rebinds_the_engine(monkeypatch):
    """A DataLoader's ragged last batch (the reference's own _gp cannot take it: wasserstein.py:110 reshapes with hp.batch_size):
    the mirror re-creates its shape-bound engine for the new batch size and carries the parameters AND both Adam states over, so
    the two steps equal two steps of ONE continuously running optimizer (stage.py:63-64) -- checked against the oracle run the same
    way, and told apart from an optimizer whose moments restart at the re-bind."""
    from downgan_amd import backend, synthetic
    from downgan_amd.GAN import losses
    from downgan_amd.GAN.dataloader import NetCDFSR
    from downgan_amd.GAN.wasserstein import WassersteinGAN
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    from oracle import ref_step
    from oracle.emu_ops import EmuOps
    import downgan_amd.config.hyperparams as hp
    monkeypatch.setattr(backend, "make_ops", lambda dtype, device: EmuOps("f32"))
    monkeypatch.setattr(losses, "_ops", {})
    monkeypatch.setattr(hp, "batch_size", 2)
    torch.set_num_threads(4)
    coarse, fine = synthetic.tiles(3, 2, 16, seed=6)
    tc, tf = torch.from_numpy(coarse), torch.from_numpy(fine)
    G, C = Generator(16, 128, 2, 2, num_res_blocks=1), Critic(16, 128, 2)
    sd0 = C.state_dict()
    tr = WassersteinGAN(G, C)
    tr.log_metrics = False
    alphas = [torch.from_numpy(synthetic.alpha(2, 0)), torch.from_numpy(synthetic.alpha(1, 1))]
    it = iter(alphas)
    monkeypatch.setattr(tr, "_alpha", lambda e, a: next(it))
    dl = torch.utils.data.DataLoader(NetCDFSR(tc, tf), batch_size=2)
    (log,) = tr.train(dl, None, epochs=1)
    assert len(log) == 2 and tr._engine.B == 1 and tr.num_steps == 2
    assert tr._engine.C.P.t == 2 and tr._engine.G.P.t == 1            # the step counts travelled with the moments
    import math
    assert math.isfinite(log[0]["critic_loss"]) and math.isfinite(log[1]["critic_loss"])
    sd1 = C.state_dict()
    moved = max(float((sd1[k] - sd0[k]).abs().max()) for k in sd0)
    assert 1.5 * hp.lr < moved <= 2.0 * hp.lr * 1.01, moved        # Adam moves an entry by at most lr per step: > 1.5 lr needs both steps

    def oracle(reset):
        pg = {k: torch.from_numpy(v) for k, v in synthetic.generator_params(16, 2, 2, 1).items()}
        pc = {k: torch.from_numpy(v) for k, v in synthetic.critic_params(16, 128, 2).items()}
        orc = ref_step.OracleTrainer(pg, pc, ref_step.HP(batch_size=2), num_res_blocks=1)
        r0 = orc.train_step(tc[:2], tf[:2], alphas[0])
        orc.hp.batch_size = 1                                      # wasserstein.py:110 reads the global batch size
        if reset:
            orc.C_opt = ref_step.Adam(orc.PC, orc.hp)
        r1 = orc.train_step(tc[2:], tf[2:], alphas[1])
        return r0, r1, {k: v.detach() for k, v in orc.PC.items()}
    r0, r1, cont = oracle(False)
    _, _, restarted = oracle(True)
    for k in ("critic_loss", "c_real_mean", "gp_ret"):
        assert abs(log[0][k] - r0[k]) <= 1e-4 * max(abs(r0[k]), 1e-3) and abs(log[1][k] - r1[k]) <= 1e-4 * max(abs(r1[k]), 1e-3), (k, log[1][k], r1[k])
    d_cont = sum(float((sd1[k] - cont[k]).abs().sum()) for k in cont)
    d_restart = sum(float((sd1[k] - restarted[k]).abs().sum()) for k in cont)
    assert d_cont < 0.1 * d_restart, (d_cont, d_restart)


def test_test_loader_with_another_batch_size_is_evaluated(monkeypatch):
    """The epoch's test pass (wasserstein.py:157-170) evaluates EVERY test batch, also when the test loader's batch size differs
    from the train loader's or its last batch is ragged -- none is dropped silently; an empty test loader raises."""
    from downgan_amd import backend, synthetic
    from downgan_amd.GAN import losses
    from downgan_amd.GAN.dataloader import NetCDFSR
    from downgan_amd.GAN.wasserstein import WassersteinGAN
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    from oracle.emu_ops import EmuOps
    import downgan_amd.config.hyperparams as hp
    monkeypatch.setattr(backend, "make_ops", lambda dtype, device: EmuOps("f32"))
    monkeypatch.setattr(losses, "_ops", {})
    monkeypatch.setattr(hp, "batch_size", 2)
    torch.set_num_threads(4)
    coarse, fine = synthetic.tiles(5, 2, 16, seed=8)
    tc, tf = torch.from_numpy(coarse), torch.from_numpy(fine)
    tr = WassersteinGAN(Generator(16, 128, 2, 2, num_res_blocks=1), Critic(16, 128, 2))
    train = torch.utils.data.DataLoader(NetCDFSR(tc[:2], tf[:2]), batch_size=2)
    test = torch.utils.data.DataLoader(NetCDFSR(tc[2:], tf[2:]), batch_size=2)     # batches of 2 and 1
    tr.train(train, test, epochs=1)
    (ep,) = tr.metrics_log
    assert ep["test_batches"] == 2 and set(ep["test"]) >= {"MAE", "MSE", "Wass"}
    # the ragged test batch (1 sample) ran PADDED on the training engine's buffers: no re-bind, the optimizer state untouched
    assert tr._engine.B == 2 and tr._engine.C.P.t == 1
    padded = tr.gen_batch_and_log_metrics(tc[4:], tf[4:])
    assert tr._engine.B == 2
    tr._eng(tc[4:], tf[4:])                                                        # the same batch on an engine of its own size
    assert tr._engine.B == 1 and tr._engine.C.P.t == 1
    own = tr.gen_batch_and_log_metrics(tc[4:], tf[4:])
    for k in ("MAE", "MSE", "Wass", "MSSSIM"):
        assert own[k] is not None and abs(padded[k] - own[k]) <= 1e-5 * max(abs(own[k]), 1e-3), (k, padded[k], own[k])
    # a LARGER batch still re-binds (its MS-SSIM normalisation spans the whole batch)
    tr.gen_batch_and_log_metrics(tc[:3], tf[:3])
    assert tr._engine.B == 3 and tr._engine.C.P.t == 1
    with pytest.raises(ValueError):
        tr.train(train, torch.utils.data.DataLoader(NetCDFSR(tc[:0], tf[:0]), batch_size=2), epochs=1)


def test_failed_rebind_restores_the_running_engine(monkeypatch):
    """If the engine for a new batch size cannot be built (out of memory at 142 GiB, an unsupported shape), the parameters and Adam
    state -- which live only in the carried device buffers at that moment -- go back into an engine of the size that was running:
    state_dict() / checkpoints keep seeing the TRAINED weights (round-3 advice), and the error reaches the caller."""
    from downgan_amd import backend, synthetic
    from downgan_amd.GAN import losses
    from downgan_amd.GAN.wasserstein import WassersteinGAN
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    from oracle.emu_ops import EmuOps
    import downgan_amd.config.hyperparams as hp
    monkeypatch.setattr(backend, "make_ops", lambda dtype, device: EmuOps("f32"))
    monkeypatch.setattr(losses, "_ops", {})
    monkeypatch.setattr(hp, "batch_size", 2)
    torch.set_num_threads(4)
    coarse, fine = synthetic.tiles(5, 2, 16, seed=9)
    tc, tf = torch.from_numpy(coarse), torch.from_numpy(fine)
    G, C = Generator(16, 128, 2, 2, num_res_blocks=1), Critic(16, 128, 2)
    sd_init = C.state_dict()
    tr = WassersteinGAN(G, C)
    tr._critic_train_iteration(tc[:2], tf[:2], alpha=torch.from_numpy(synthetic.alpha(2, 0)))
    trained = C.state_dict()
    assert any(not torch.equal(trained[k], sd_init[k]) for k in trained)
    real_build = tr._build

    def failing(B, cin, S, carry):
        if B == 3:
            raise MemoryError("no room for the batch-3 engine")
        return real_build(B, cin, S, carry)
    monkeypatch.setattr(tr, "_build", failing)
    with pytest.raises(MemoryError):
        tr._critic_train_iteration(tc[:3], tf[:3], alpha=torch.from_numpy(synthetic.alpha(3, 1)))
    assert tr._engine is not None and tr._engine.B == 2 and tr._engine.C.P.t == 1
    after = C.state_dict()
    assert all(torch.equal(after[k], trained[k]) for k in trained)                 # not the initial weights
    tr._critic_train_iteration(tc[:2], tf[:2], alpha=torch.from_numpy(synthetic.alpha(2, 1)))     # and training goes on
    assert tr._engine.C.P.t == 2
