"""Drop-in mirror of the reference Generator (reference DoWnGAN/networks/generator.py:56-90).

Same constructor arguments and ``forward`` semantics (NCHW fp32 in, NCHW fp32 out), same
``state_dict`` keys/shapes (OIHW), but the forward runs the native NHWC HIP kernels.  Native buffers
depend on the batch/tile shape, so the native network is instantiated on first use per shape.
"""
from __future__ import annotations

import torch

from .. import synthetic
from ..engine import NativeGenerator
from .. import backend
from ._module import NativeModule


class Generator(NativeModule):
    def __init__(self, filters, fine_dims, channels, n_predictands=2, num_res_blocks=16, num_upsample=3,
                 dtype="bf16", device="cuda:0"):
        self.filters, self.fine_dims, self.channels = filters, fine_dims, channels   # fine_dims unused, as in the reference
        self.n_predictands, self.num_res_blocks, self.num_upsample = n_predictands, num_res_blocks, num_upsample
        self.dtype, self.device = dtype, device
        self._sd = {k: torch.from_numpy(v) for k, v in
                    synthetic.generator_params(filters, channels, n_predictands, num_res_blocks, num_upsample).items()}
        self._native = {}
        self._bound = None      # NativeGenerator owned by a TrainEngine (then it holds the live parameters)

    # -- parameters ---------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        missing = set(self._sd) - set(sd)
        if missing:
            raise KeyError(f"missing keys: {sorted(missing)[:4]}...")
        self._sd = {k: torch.as_tensor(sd[k], dtype=torch.float32).detach().cpu().clone() for k in self._sd}
        for n in list(self._native.values()) + ([self._bound] if self._bound else []):
            n.load_state_dict(self._sd)

    def state_dict(self):
        if self._bound is not None:
            self._sd = self._bound.state_dict()
        return {k: v.clone() for k, v in self._sd.items()}

    def bind(self, native: NativeGenerator, load=True):
        """Hand the live parameters to a TrainEngine's network; ``load=False``: the engine already holds them (state carried over
        from the engine it replaces)."""
        if load:
            native.load_state_dict(self.state_dict())
        self._bound = native

    def _get(self, B, S):
        if self._bound is not None and (self._bound.B, self._bound.S) == (B, S):
            return self._bound
        key = (B, S)
        if key not in self._native:
            ops = backend.make_ops(self.dtype, self.device)
            n = NativeGenerator(ops, self.filters, self.channels, B, S, self.n_predictands, self.num_res_blocks, self.num_upsample)
            n.load_state_dict(self.state_dict())
            self._native[key] = n
        return self._native[key]

    def forward(self, x):
        assert x.dim() == 4 and x.shape[1] == self.channels and x.shape[2] == x.shape[3], x.shape
        B, _, S, _ = x.shape
        n = self._get(B, S)
        o = n.ops
        xn = o.zeros(B, S, S, n.cin_p)
        o.nchw_to_nhwc(x.to(device=o.device, dtype=torch.float32).contiguous(), xn)
        fake = n.forward(xn, save=False)
        out = o.zeros(B, self.n_predictands, S << self.num_upsample, S << self.num_upsample, dtype=torch.float32)
        o.nhwc_to_nchw(fake, out)
        return out

    __call__ = forward

    def generate(self, coarse, chunk_size=None):
        """Chunked inference over a long series of coarse fields, like the reference's generation script
        (reference DoWnGAN/helpers/gen_fake_ds.py:147-162: ``G(chunk)`` per chunk, concatenated on the host)."""
        n = coarse.shape[0]
        chunk_size = chunk_size or n
        outs = []
        for i in range(0, n, chunk_size):
            part = coarse[i:i + chunk_size]
            if part.shape[0] < chunk_size and i > 0:      # pad the ragged tail so one native instance serves all chunks
                pad = torch.zeros(chunk_size - part.shape[0], *part.shape[1:], dtype=part.dtype, device=part.device)
                outs.append(self.forward(torch.cat([part, pad], 0))[:part.shape[0]].cpu())
            else:
                outs.append(self.forward(part).cpu())
        return torch.cat(outs, 0)
