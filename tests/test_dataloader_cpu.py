"""Data feed host logic: dataset mirror and the epoch order / rank partition of ResidentLoader."""
import torch
from torch.utils.data import DataLoader

from downgan_amd.GAN.dataloader import NetCDFSR, epoch_indices


def test_netcdfsr_matches_reference_semantics():
    coarse, fine = torch.arange(5 * 2 * 4 * 4.0).view(5, 2, 4, 4), torch.arange(5 * 2 * 8 * 8.0).view(5, 2, 8, 8)
    ds = NetCDFSR(coarse, fine, device=torch.device("cpu"))
    assert len(ds) == 5
    c, f = ds[torch.tensor(3)]                                # tensor index -> .tolist() (dataloader.py:28-29)
    assert torch.equal(c, coarse[3]) and torch.equal(f, fine[3])
    batches = list(DataLoader(ds, batch_size=2, shuffle=False))
    assert len(batches) == 3 and torch.equal(batches[1][0], coarse[2:4]) and torch.equal(batches[2][1], fine[4:5])


def test_epoch_order_is_a_partition_shared_by_all_ranks():
    n, B, world = 37, 4, 2
    a = epoch_indices(n, B, world, epoch=0, seed=7)
    assert a.shape == (n // (B * world), world, B)
    flat = a.flatten().tolist()
    assert len(set(flat)) == len(flat) and all(0 <= i < n for i in flat)        # every sample at most once, ragged tail dropped
    assert torch.equal(a, epoch_indices(n, B, world, epoch=0, seed=7))          # same on every rank
    assert not torch.equal(a, epoch_indices(n, B, world, epoch=1, seed=7))      # reshuffled per epoch
    assert torch.equal(epoch_indices(8, 2, 1, 0, 0, shuffle=False).flatten(), torch.arange(8))
