"""Drop-in mirror of the reference's frequency-separation trainer (reference DoWnGAN/GAN/wasserstein_fs.py:15-198).

``WassersteinGANFS(G, C, G_optimizer, C_optimizer)``: same interface as ``WassersteinGAN``; the critic is trained on the
high-pass parts ``x - low(x)`` of the real and generated fields and the content loss on the low-pass parts, with
``low = AvgPool2d(5, 1, 0) o ReplicationPad2d(2)`` (hyperparams.py:31-35).  The reference module itself is not importable
(broken imports, wasserstein_fs.py:2-10) and nothing reads ``hp.freq_sep``; this follows the text of its iteration methods.
"""
from __future__ import annotations

from ..config import hyperparams as hp
from ..engine import TrainEngineFS
from .. import backend
from .wasserstein import WassersteinGAN


class WassersteinGANFS(WassersteinGAN):
    def _eng(self, coarse, fine):
        B, cin, S, _ = coarse.shape
        if self._engine is None or (self._engine.B, self._engine.S) != (B, S):
            assert self.G.dtype == self.C.dtype
            ops = backend.make_ops(self.G.dtype, self.G.device)
            e = TrainEngineFS(ops, S, self.G.filters, cin, B, hp.as_engine_hp(B), self.G.n_predictands,
                              self.G.num_res_blocks, self.G.num_upsample, dist=self.dist)
            self.G.bind(e.G)
            self.C.bind(e.C)
            self._adopt_optimizers(e)
            e.num_steps = self.num_steps
            self._engine = e
            self._stage = (ops.zeros(B, S, S, e.G.cin_p), ops.zeros(B, fine.shape[2], fine.shape[3], e.G.np_p))
        return self._engine
