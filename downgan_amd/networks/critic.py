"""Drop-in mirror of the reference Critic (reference DoWnGAN/networks/critic.py:9-106):
``Critic(coarse_dim, fine_dim, nc)``, ``forward(NCHW fp32) -> [B, 1]``, reference state_dict keys."""
from __future__ import annotations

import torch

from .. import synthetic
from ..engine import NativeCritic
from .. import backend
from ._module import NativeModule


class Critic(NativeModule):
    def __init__(self, coarse_dim, fine_dim, nc, dtype="bf16", device="cuda:0"):
        self.coarse_dim, self.fine_dim, self.nc = coarse_dim, fine_dim, nc
        self.dtype, self.device = dtype, device
        self._sd = {k: torch.from_numpy(v) for k, v in synthetic.critic_params(coarse_dim, fine_dim, nc).items()}
        self._native = {}
        self._bound = None

    def load_state_dict(self, sd):
        missing = set(self._sd) - set(sd)
        if missing:
            raise KeyError(f"missing keys: {sorted(missing)}")
        self._sd = {k: torch.as_tensor(sd[k], dtype=torch.float32).detach().cpu().clone() for k in self._sd}
        for n in list(self._native.values()) + ([self._bound] if self._bound else []):
            n.load_state_dict(self._sd)

    def state_dict(self):
        if self._bound is not None:
            self._sd = self._bound.state_dict()
        return {k: v.clone() for k, v in self._sd.items()}

    def bind(self, native: NativeCritic, load=True):
        """Hand the live parameters to a TrainEngine's network; ``load=False``: the engine already holds them (state carried over
        from the engine it replaces)."""
        if load:
            native.load_state_dict(self.state_dict())
        self._bound = native

    def _get(self, B):
        if self._bound is not None and self._bound.B == B:
            return self._bound
        if B not in self._native:
            n = NativeCritic(backend.make_ops(self.dtype, self.device), self.coarse_dim, self.fine_dim, self.nc, B)
            n.load_state_dict(self.state_dict())
            self._native[B] = n
        return self._native[B]

    def forward(self, x):
        assert x.dim() == 4 and x.shape[1] == self.nc and x.shape[2] == x.shape[3] == self.fine_dim, x.shape
        B = x.shape[0]
        n = self._get(B)
        o = n.ops
        xn = o.zeros(B, self.fine_dim, self.fine_dim, n.c_pad[0])
        o.nchw_to_nhwc(x.to(device=o.device, dtype=torch.float32).contiguous(), xn)
        return n.forward(xn)[:, :1].clone()

    __call__ = forward
