"""The slice of ``torch.nn.Module``'s surface that the reference's callers use on its networks
(reference DoWnGAN/GAN/stage.py:59-64: ``.to(config.device)``, ``.parameters()`` handed to ``torch.optim.Adam``;
mlflow_tools/mlflow_epoch.py:53-69: ``__call__``, ``state_dict()``), provided by the native mirrors."""
from __future__ import annotations

import torch


class NativeModule:
    training = True

    # -- device -------------------------------------------------------------------------------------
    def to(self, device=None, *args, **kwargs):
        """Re-bind the network to ``device`` (``torch.device`` / str / index) and return self, like ``nn.Module.to``.
        Native instances of another device are dropped; parameters live in ``state_dict`` form and follow.  A dtype
        argument is ignored: precision is the ``dtype=`` constructor argument ("bf16" / "f32")."""
        if device is None or isinstance(device, torch.dtype):
            return self
        dev = torch.device(f"cuda:{device}") if isinstance(device, int) else torch.device(device)
        name = str(dev) if dev.index is not None or dev.type != "cuda" else "cuda:0"
        if name != self.device:
            self._sd = self.state_dict()
            self._native, self._bound = {}, None
            self.device = name
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else 0))

    # -- parameters ---------------------------------------------------------------------------------
    def named_parameters(self, prefix="", recurse=True):
        """(name, Parameter) pairs with the reference's names / OIHW shapes.  They are host-side views of the state_dict for
        API compatibility -- ``torch.optim.Adam(G.parameters(), lr, betas=...)`` (stage.py:63-64) constructs and its
        hyper-parameters are read by the trainer -- while the live parameters, their gradients and Adam itself stay in the
        flat native buffers (engine.ParamStore)."""
        for k, v in self.state_dict().items():
            yield (prefix + ("." if prefix else "") + k, torch.nn.Parameter(v, requires_grad=False))

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def train(self, mode=True):
        self.training = bool(mode)        # no dropout / batch-norm anywhere on the path: mode changes nothing
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self, set_to_none=True):
        for n in list(self._native.values()) + ([self._bound] if self._bound is not None else []):
            n.P.zero_grad()

    def requires_grad_(self, requires_grad=True):
        return self
