"""Dataset preprocessing + staging of the data feed, native.

Reference: ``DoWnGAN/helpers/gen_experiment_datasets.py:195-233`` standardises every field over its whole record --
``xr_standardize_array``: ``(da - da.mean(skipna=True)) / da.std(skipna=True)`` (population std) -- except the binary
``land_sea_mask`` (:208-209), concatenates the variables in the configured order (:155-166) and ``DoWnGAN/GAN/stage.py:28-31``
moves the result to the device as ``[time, var, lat, lon]`` fp32.  (The NetCDF / xarray reading itself stays out of scope.)

Here the raw fields (one ``[time, lat, lon]`` fp32 array per variable, host memory) go through the GPU twice, a bounded chunk
of time steps at a time: pass 1 accumulates {sum, sum of squares, count} of the non-NaN values in double (``dg_moments``),
pass 2 writes ``(x - mean) / std`` -- identity for exempt fields -- straight into the HBM-resident ``[n, H, W, c]`` store in
the compute dtype (``dg_stage_fields``) that ``ResidentLoader`` gathers minibatches from.  Standardising BEFORE the bf16
rounding matters: raw surface pressure (~1e5 Pa) has no useful bits left in bf16.
"""
from __future__ import annotations

import torch

EXEMPT = ("land_sea_mask",)       # gen_experiment_datasets.py:208-209: "Binary land mask does not need normalization"


def _chunks(t, chunk):
    for i in range(0, t.shape[0], chunk):
        yield i, torch.as_tensor(t[i:i + chunk], dtype=torch.float32)


def field_stats(ops, fields: dict, exempt=EXEMPT, chunk=64):
    """{name: (mean, std)} over each field's whole record (NaNs skipped, population std); exempt fields -> (0.0, 1.0).

    To reproduce the reference, ``fields`` must hold each variable's WHOLE record: gen_experiment_datasets.py:241-250 standardises
    before the year-based train / test split, so both splits share statistics taken over train + test together
    (``whole_record_stats`` below takes the two splits and does that).  A field without a finite value raises; a constant field
    (std 0) raises too -- the reference would divide by zero and fill it with NaN / Inf."""
    stats = {}
    acc = torch.zeros(3, dtype=torch.float64, device=ops.device)
    for name, t in fields.items():
        if name in exempt:
            stats[name] = (0.0, 1.0)
            continue
        acc.zero_()
        for _, part in _chunks(t, chunk):
            ops.moments(part.to(ops.device).contiguous(), acc)
        s, ss, n = acc.cpu().tolist()
        if n == 0:
            raise ValueError(f"field '{name}' has no finite value: its mean / std are undefined")
        mean = s / n
        var = max(ss / n - mean * mean, 0.0)
        if var == 0.0:
            raise ValueError(f"field '{name}' is constant ({mean}): (x - mean) / std is undefined -- list it in `exempt` (as the "
                             "reference does for the binary land_sea_mask) or drop it")
        stats[name] = (mean, var ** 0.5)
    return stats


def whole_record_stats(ops, train_fields: dict, test_fields: dict, exempt=EXEMPT, chunk=64):
    """Statistics over train + test together, as the reference computes them (it standardises each variable's whole record and
    only then splits by year, gen_experiment_datasets.py:241-250): pass the result as ``stats=`` when staging EITHER split."""
    both = {k: torch.cat([torch.as_tensor(train_fields[k], dtype=torch.float32), torch.as_tensor(test_fields[k], dtype=torch.float32)], 0)
            for k in train_fields}
    return field_stats(ops, both, exempt, chunk)


def stage_standardized(ops, fields: dict, order, stats=None, exempt=EXEMPT, chunk=64):
    """Standardise and stage: returns the resident store ``[n, H, W, c]`` (compute dtype) with the variables in ``order``
    (config.covariate_names_ordered / fine_names_ordered, gen_experiment_datasets.py:155-166) and the statistics used."""
    stats = stats or field_stats(ops, {k: fields[k] for k in order}, exempt, chunk)
    first = fields[order[0]]
    n, H, W = first.shape
    assert all(tuple(fields[k].shape) == (n, H, W) for k in order), "fields of one dataset share [time, lat, lon]"
    store = torch.empty(n, H, W, len(order), dtype=ops.tdtype, device=ops.device)
    mean = [stats[k][0] for k in order]
    inv = [1.0 / stats[k][1] for k in order]
    for i in range(0, n, chunk):
        planes = [torch.as_tensor(fields[k][i:i + chunk], dtype=torch.float32).to(ops.device).contiguous() for k in order]
        ops.stage_fields(planes, mean, inv, store[i:i + chunk])
    return store, stats
