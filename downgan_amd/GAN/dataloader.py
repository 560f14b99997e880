"""Data feed of the training loop.

``NetCDFSR`` mirrors the reference's dataset (DoWnGAN/GAN/dataloader.py:6-33: two pre-loaded tensors, ``__getitem__`` returns
``(coarse[idx], fine[idx])``) and works with ``torch.utils.data.DataLoader`` exactly like it.

``ResidentLoader`` is the MI355X-native replacement of ``DataLoader(NetCDFSR(...), batch_size, shuffle=True)``
(stage.py:69-72).  The reference already keeps the whole train set on the GPU (stage.py:28-31); here it is staged ONCE into
HBM as ``[n, H, W, c]`` in the compute dtype (real channels only: 4 MB per 2 x 1024 x 1024 bf16 sample, so tens of thousands
of samples fit in 288 GB) and every minibatch is formed by one ``dg_gather_samples`` launch straight into the native padded
NHWC activation buffers -- no per-step NCHW->NHWC conversion, no host round trip.  Sharded runs use one permutation per
epoch (same seed on every rank); rank r takes samples ``[r*B, (r+1)*B)`` of every global batch.  Only full batches are
produced: the reference's ``_gp`` reshapes with the global ``hp.batch_size`` (wasserstein.py:110) and cannot take the ragged
last batch of its own DataLoader.
"""
from __future__ import annotations

import torch
from torch.utils.data import Dataset

from .. import backend, layout
from ..ops import HipOps


class NetCDFSR(Dataset):
    """Data loader from torch.Tensors (reference dataloader.py:6-33; ``device`` is accepted and unused there too)."""

    def __init__(self, coarse: torch.Tensor, fine: torch.Tensor, device=None):
        self.fine = fine
        self.coarse = coarse

    def __len__(self):
        return self.fine.size(0)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return self.coarse[idx, ...], self.fine[idx, ...]


class NativeBatch:
    """A minibatch already in native layout: ``nhwc`` is the padded NHWC activation tensor, ``shape`` its NCHW meaning."""

    def __init__(self, nhwc, channels):
        self.nhwc, self.channels = nhwc, channels

    @property
    def shape(self):
        B, H, W, _ = self.nhwc.shape
        return (B, self.channels, H, W)


def epoch_indices(n, batch, world, epoch, seed, shuffle=True):
    """Global sample order of one epoch, cut to full global batches: LongTensor [n_batches, world, batch]."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n, generator=g) if shuffle else torch.arange(n)
    nb = n // (batch * world)
    return perm[:nb * batch * world].view(nb, world, batch)


class ResidentLoader:
    def __init__(self, dataset: NetCDFSR, batch_size, shuffle=True, dtype="bf16", device="cuda:0", rank=0, world=1, seed=0,
                 ops: HipOps = None, stage_chunk=64, _stores=None):
        self.ops = ops or backend.make_ops(dtype, device)
        o = self.ops
        self.batch, self.shuffle, self.rank, self.world, self.seed = batch_size, shuffle, rank, world, seed
        if _stores is None:
            self.store_c = self._stage(dataset.coarse, stage_chunk)
            self.store_f = self._stage(dataset.fine, stage_chunk)
        else:
            self.store_c, self.store_f = _stores
        self.n = self.store_f.shape[0]
        assert self.n >= batch_size * world and self.store_c.shape[0] == self.n, "dataset smaller than one global batch"
        self.cc, self.cf = self.store_c.shape[3], self.store_f.shape[3]
        B = batch_size
        self.xc = o.zeros(B, self.store_c.shape[1], self.store_c.shape[2], layout.pad16(self.cc))
        self.xf = o.zeros(B, self.store_f.shape[1], self.store_f.shape[2], layout.pad16(self.cf))
        self.epoch = 0
        self.stats = None

    @classmethod
    def from_fields(cls, coarse_fields: dict, fine_fields: dict, coarse_order, fine_order, batch_size, stats=None, **kw):
        """Raw per-variable records (``{name: [time, lat, lon] fp32}``, e.g. ERA-Interim covariates and WRF predictands) ->
        standardised resident stores (GAN/preprocess.py: gen_experiment_datasets.py:195-233 + stage.py:28-31 in two GPU
        passes) -> loader.  ``stats`` ({"coarse": ..., "fine": ...}): statistics to standardise with instead of this record's own.
        The reference standardises each variable over its WHOLE record before the train / test split
        (gen_experiment_datasets.py:241-250): to reproduce it pass ``preprocess.whole_record_stats(ops, train, test)`` for BOTH
        splits (statistics of the train split alone are a different, leakage-free convention)."""
        from . import preprocess
        ops = kw.pop("ops", None) or backend.make_ops(kw.get("dtype", "bf16"), kw.get("device", "cuda:0"))
        chunk = kw.get("stage_chunk", 64)
        sc, st_c = preprocess.stage_standardized(ops, coarse_fields, list(coarse_order), (stats or {}).get("coarse"), chunk=chunk)
        sf, st_f = preprocess.stage_standardized(ops, fine_fields, list(fine_order), (stats or {}).get("fine"), chunk=chunk)
        self = cls(None, batch_size, ops=ops, _stores=(sc, sf), **kw)
        self.stats = {"coarse": st_c, "fine": st_f}
        return self

    def _stage(self, t, chunk):
        """NCHW fp32 (host or device) -> resident [n, H, W, c] in the compute dtype, a bounded chunk at a time."""
        o = self.ops
        n, c, H, W = t.shape
        store = torch.empty(n, H, W, c, dtype=o.tdtype, device=o.device)
        for i in range(0, n, chunk):
            store[i:i + chunk] = t[i:i + chunk].to(o.device, torch.float32).permute(0, 2, 3, 1).to(o.tdtype)
        return store

    def __len__(self):
        return self.n // (self.batch * self.world)

    def __iter__(self):
        o = self.ops
        order = epoch_indices(self.n, self.batch, self.world, self.epoch, self.seed, self.shuffle)
        self.epoch += 1
        for gb in order:
            idx = gb[self.rank].to(o.device)
            o.gather_samples(self.store_c, idx, self.xc)
            o.gather_samples(self.store_f, idx, self.xf)
            yield NativeBatch(self.xc, self.cc), NativeBatch(self.xf, self.cf)
