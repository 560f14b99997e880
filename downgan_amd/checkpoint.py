"""Checkpoint interchange with the reference.

The reference checkpoints both networks every epoch with ``mlflow.pytorch.log_state_dict(net.state_dict(), "<Net>/<Net>_<epoch>")``
(DoWnGAN/mlflow_tools/mlflow_epoch.py:65-69), i.e. a ``torch.save``d dict of OIHW fp32 tensors under the reference's parameter
names, stored as ``<artifact dir>/state_dict.pth``.  The native networks read and write exactly that: the NHWC / tap-major
repacking (and the NCHW->NHWC column permutation of the critic's first Linear) happens inside ``load_state_dict`` /
``state_dict`` of the native nets, so a file written here loads into the reference's ``Generator`` / ``Critic`` and vice versa.
"""
from __future__ import annotations

import os

import torch

STATE_DICT_FILE = "state_dict.pth"      # mlflow.pytorch's file name for log_state_dict artifacts


def save_state_dict(net, path):
    """Write ``net.state_dict()`` (reference names, OIHW fp32, CPU) to ``path`` (a file, or a directory -> state_dict.pth)."""
    if os.path.isdir(path) or path.endswith(os.sep):
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, STATE_DICT_FILE)
    sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in net.state_dict().items()}
    torch.save(sd, path)
    return path


def load_state_dict(net, path):
    """Load a reference-format checkpoint (file, or directory holding state_dict.pth) into a native or mirror network."""
    if os.path.isdir(path):
        path = os.path.join(path, STATE_DICT_FILE)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict):
        raise TypeError(f"{path} does not hold a state_dict")
    net.load_state_dict(sd)
    return net


def log_network_models(C, G, epoch, root):
    """Local-directory counterpart of mlflow_epoch.py:65-69: ``<root>/Critic/Critic_<epoch>/state_dict.pth`` and the same
    for the generator (the pickled whole-module artifact of ``log_model`` has no native counterpart)."""
    out = []
    for name, net in (("Critic", C), ("Generator", G)):
        d = os.path.join(root, name, f"{name}_{epoch}")
        os.makedirs(d, exist_ok=True)
        out.append(save_state_dict(net, os.path.join(d, STATE_DICT_FILE)))
    return out
