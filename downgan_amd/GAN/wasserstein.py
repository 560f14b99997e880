"""Drop-in mirror of the reference trainer (reference DoWnGAN/GAN/wasserstein.py:16-189).

``WassersteinGAN(G, C, G_optimizer, C_optimizer)`` with the same method names and argument meaning.
Differences, all additive: the iteration methods RETURN the scalars the reference computes and drops
(:46-50, :74-78); ``alpha`` can be injected into ``_critic_train_iteration`` / ``_gp`` (the reference
draws it from the device RNG at :91); Adam is the fused native kernel; when ``torch.optim.Adam`` objects
built the reference's way (stage.py:63-64 over ``G.parameters()`` / ``C.parameters()``) are passed, their lr / betas /
eps are adopted, otherwise ``config.hyperparams`` applies (the arguments may be None).  The epoch loop keeps the reference's per-batch metrics pass, test-set pass, epoch means and per-epoch checkpoints
(:138-179) with mlflow / plotting replaced by return values, ``metrics_log`` and plain files.
"""
from __future__ import annotations

import torch

from ..config import hyperparams as hp
from ..engine import TrainEngine
from .. import backend


def _adam_config(opt, base):
    """Hyper-parameters of a ``torch.optim.Adam`` built the reference's way (stage.py:63-64) -> engine HyperParams.  The
    optimizer object itself never steps: Adam runs as the fused native kernel over the flat parameter buffers."""
    if opt is None or not hasattr(opt, "param_groups"):
        return base
    import dataclasses
    g = opt.param_groups[0]
    if type(opt).__name__ != "Adam" or g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
        raise NotImplementedError("the native trainer implements plain Adam (no weight decay / amsgrad), as the reference configures it")
    return dataclasses.replace(base, lr=float(g["lr"]), beta1=float(g["betas"][0]), beta2=float(g["betas"][1]), eps=float(g["eps"]))


class WassersteinGAN:
    def __init__(self, G, C, G_optimizer=None, C_optimizer=None, dist=None) -> None:
        self.G, self.C = G, C
        self.G_optimizer, self.C_optimizer = G_optimizer, C_optimizer
        self.num_steps = 0
        self.dist = dist
        self._engine = None
        self._stage = None
        self.last = {}
        self.metrics_log = []

    check_finite = False         # debug: NaN / Inf census after every iteration (TrainEngine(check_finite=True); the reference
                                 # runs with torch.autograd.set_detect_anomaly(True), wasserstein.py:13)

    def _eng(self, coarse, fine):
        """The shape-bound native engine for this batch.  A batch of another size (a DataLoader's ragged last batch, a test loader
        with its own batch size) re-creates it; the master parameters AND both Adam states (m, v, step count) are carried over on
        the device, so training continues exactly as with one optimizer (stage.py:63-64 builds them once).  The old engine's
        buffers are released BEFORE the new ones are allocated (cfg2: 142 GiB each)."""
        B, cin, S, _ = coarse.shape
        if self._engine is None or (self._engine.B, self._engine.S) != (B, S):
            assert self.G.dtype == self.C.dtype
            carry = old_shape = None
            if self._engine is not None and self._engine.S == S:
                old = self._engine
                old_shape = (old.B, tuple(self._stage[1].shape[1:3]))
                carry = (old.G.P.export_state(), old.C.P.export_state())      # flat fp32 buffers only (8.8 GB at cfg2)
                self.G._bound = self.C._bound = None                          # (no host round trip: the state moves on the device)
                self._engine = self._stage = old = None
                import gc
                gc.collect()
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()
            try:
                e = self._build(B, cin, S, carry)
            except Exception:
                # The live parameters and Adam state exist only in ``carry`` now (the old engine was released to make room).  Put
                # them back into an engine of the size that was running, so that state_dict() / checkpoints / the next batch see
                # the trained weights and not the networks' stale host copies; then let the caller see the failure.
                if carry is not None and old_shape is not None:
                    import gc
                    gc.collect()
                    if torch.cuda.is_available():
                        torch.cuda.empty_cache()
                    self._install(self._build(old_shape[0], cin, S, carry), old_shape[1])
                raise
            self._install(e, (fine.shape[2], fine.shape[3]))
        return self._engine

    def _build(self, B, cin, S, carry):
        ops = backend.make_ops(self.G.dtype, self.G.device)
        e = TrainEngine(ops, S, self.G.filters, cin, B, hp.as_engine_hp(B), self.G.n_predictands,
                        self.G.num_res_blocks, self.G.num_upsample, dist=self.dist, check_finite=self.check_finite)
        if carry is not None:
            e.G.P.import_state(carry[0]); e.C.P.import_state(carry[1])
        return e, carry is not None

    def _install(self, built, fine_hw):
        e, carried = built
        self.G.bind(e.G, load=not carried)
        self.C.bind(e.C, load=not carried)
        self._adopt_optimizers(e)
        e.num_steps = self.num_steps
        self._engine = e
        self._stage = (e.ops.zeros(e.B, e.S, e.S, e.G.cin_p), e.ops.zeros(e.B, fine_hw[0], fine_hw[1], e.G.np_p))

    def _adopt_optimizers(self, e):
        e.adam_hp[id(e.G.P)] = _adam_config(self.G_optimizer, e.hp)
        e.adam_hp[id(e.C.P)] = _adam_config(self.C_optimizer, e.hp)

    def _to_native(self, e, coarse, fine):
        o = e.ops
        if hasattr(coarse, "nhwc"):            # dataloader.NativeBatch: already in native layout (ResidentLoader)
            return coarse.nhwc, fine.nhwc
        xc, xf = self._stage
        o.nchw_to_nhwc(coarse.to(device=o.device, dtype=torch.float32).contiguous(), xc)
        o.nchw_to_nhwc(fine.to(device=o.device, dtype=torch.float32).contiguous(), xf)
        return xc, xf

    def _alpha(self, e, alpha):
        if alpha is None:
            return torch.rand(e.B, device=e.ops.device, dtype=torch.float32)     # wasserstein.py:91
        return torch.as_tensor(alpha, dtype=torch.float32).reshape(-1).to(e.ops.device)

    def _critic_train_iteration(self, coarse, fine, alpha=None, _keep_g=False):
        """``_keep_g`` (set by ``_train_epoch`` on generator steps): keep G(coarse) of this iteration for the generator
        iteration that follows on the same batch, which then skips its own identical forward."""
        e = self._eng(coarse, fine)
        xc, xf = self._to_native(e, coarse, fine)
        e.critic_iteration(xc, xf, self._alpha(e, alpha), save_g=_keep_g)
        self.last = e.read_scalars(False)
        return self.last

    def _generator_train_iteration(self, coarse, fine, _reuse_g=False):
        e = self._eng(coarse, fine)
        xc, xf = self._to_native(e, coarse, fine)
        e.generator_iteration(xc, xf, reuse_fake=_reuse_g)
        out = e.read_scalars(True)
        self.last.update({k: out[k] for k in ("g_loss", "content_loss", "g_c_fake_mean")})
        return {k: out[k] for k in ("g_loss", "content_loss", "g_c_fake_mean")}

    def _gp(self, real, fake, critic=None, alpha=None):
        """Value of gp_lambda * mean((||grad||-1)^2) (wasserstein.py:87-117) for NCHW real / fake."""
        B = real.shape[0]
        e = self._engine
        assert e is not None and e.B == B, "call a train iteration first (buffers are shape-bound)"
        o = e.ops
        xr = o.zeros(*e.fine_shape); o.nchw_to_nhwc(real.to(o.device, torch.float32).contiguous(), xr)
        xk = o.zeros(*e.fine_shape); o.nchw_to_nhwc(fake.to(o.device, torch.float32).contiguous(), xk)
        o.gp_interp(xr, xk, self._alpha(e, alpha), e.xhat)
        e.C.forward(e.xhat)
        e.C.backward(e.xhat, 1.0, wgrad=False, dx=e.gbuf)
        e.ss.zero_()
        o.sumsq_rows(e.gbuf, e.ss)
        o.gp_finish(e.ss, B, B * e.world, e.hp.gp_lambda, 0.0, e.coef, e._sc("gp_ret"))
        return float(e._sc("gp_ret").item())

    def gen_batch_and_log_metrics(self, coarse, fine):
        """Native version of mlflow_tools/mlflow_epoch.py:53-63 (the per-step metrics pass, wasserstein.py:140):
        returns {"MAE", "MSE", "Wass", "MSSSIM"} (MSSSIM None for tiles too small for 5 scales)."""
        e, n = self._engine, coarse.shape[0]
        if (e is not None and n < e.B and coarse.shape[2] == e.S and not hasattr(coarse, "nhwc")
                and (e.dist is None or e.world == 1)):
            # a smaller batch (a test loader's own batch size, a ragged last batch) on the TRAINING engine's buffers, padded:
            # re-binding would tear down and re-allocate every training buffer (142 GiB at configs[1]) for one G forward and two
            # critic forwards, and once more when the next training batch arrives.  Larger batches still re-bind (the MS-SSIM
            # normalisation runs over the whole batch, losses.py:15-29, so they cannot be split).
            o = e.ops
            xc, xf = self._stage
            o.nchw_to_nhwc(coarse.to(device=o.device, dtype=torch.float32).contiguous(), xc[:n])
            o.nchw_to_nhwc(fine.to(device=o.device, dtype=torch.float32).contiguous(), xf[:n])
            return e.metrics_pass(xc, xf, n_valid=n)
        e = self._eng(coarse, fine)
        xc, xf = self._to_native(e, coarse, fine)
        return e.metrics_pass(xc, xf)

    # what the reference's epoch loop does beside the two iterations (wasserstein.py:138-179), switchable because it costs one
    # extra G forward + two critic forwards per batch: the per-batch metrics pass on the train set, the same pass over the test
    # set at the epoch's end, per-epoch means (post_epoch_metric_mean), and the per-epoch checkpoint of both networks
    log_metrics = True
    checkpoint_dir = None        # e.g. "artifacts": <dir>/Critic/Critic_<epoch>/state_dict.pth (mlflow_epoch.py:65-69 without mlflow)

    @staticmethod
    def _metric_means(rows):
        keys = [k for k in ("MAE", "MSE", "MSSSIM", "Wass") if rows and rows[0].get(k) is not None]
        return {k: sum(r[k] for r in rows) / len(rows) for k in keys}

    def _train_epoch(self, dataloader, testdataloader=None, epoch=0):
        """wasserstein.py:120-179: every batch = critic iteration, generator iteration when num_steps % critic_iterations == 0
        (same batch), metrics pass (:140-146); then the epoch means of the train metrics, the metrics over the test loader
        (:157-170) and the checkpoint (:178).  Plotting (gen_grid_images) and mlflow are out of scope; the per-step scalars are
        returned and the epoch summary is appended to ``self.metrics_log``."""
        log, train_metrics, test_metrics = [], [], []
        for data in dataloader:
            coarse, fine = data[0], data[1]
            gen_step = self.num_steps % hp.critic_iterations == 0                 # :136
            out = dict(self._critic_train_iteration(coarse, fine, _keep_g=gen_step))
            if gen_step:
                out.update(self._generator_train_iteration(coarse, fine, _reuse_g=True))
            if self.log_metrics:
                train_metrics.append(self.gen_batch_and_log_metrics(coarse, fine))   # :140-146
            self.num_steps += 1
            if self._engine is not None:
                self._engine.num_steps = self.num_steps
            log.append(out)
        summary = {"epoch": epoch}
        if self.log_metrics:
            summary["train"] = self._metric_means(train_metrics)                     # :150
            if testdataloader is not None:
                # :157-168.  EVERY test batch is evaluated, whatever its size (the engine re-binds, state carried over); the
                # reference's epoch mean is the mean over batches (post_epoch_metric_mean), a ragged batch counting as one
                for data in testdataloader:
                    test_metrics.append(self.gen_batch_and_log_metrics(data[0], data[1]))
                if not test_metrics:
                    raise ValueError("the test loader yielded no batch: no test metrics for this epoch (wasserstein.py:157-170)")
                summary["test"] = self._metric_means(test_metrics)                   # :170
                summary["test_batches"] = len(test_metrics)
        if self.checkpoint_dir is not None:
            from ..checkpoint import log_network_models
            summary["checkpoints"] = log_network_models(self.C, self.G, epoch, self.checkpoint_dir)   # :178
        self.metrics_log.append(summary)
        return log

    def train(self, dataloader, testdataloader=None, epochs=None):
        """wasserstein.py:181-189."""
        self.num_steps = 0
        self.metrics_log = []
        history = []
        for epoch in range(hp.epochs if epochs is None else epochs):
            history.append(self._train_epoch(dataloader, testdataloader, epoch))
        return history
