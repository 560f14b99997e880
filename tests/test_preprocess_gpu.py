"""dg_moments / dg_stage_fields through the C ABI against the oracle restatement of the reference's standardisation
(gen_experiment_datasets.py:195-233) and staging (stage.py:28-31)."""
import os
import sys

import numpy as np
import pytest
import torch

from downgan_amd.GAN import preprocess as pp
from downgan_amd.GAN.dataloader import ResidentLoader
from downgan_amd.ops import HipOps
from oracle import preprocess as op
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_preprocess_cpu import ORDER, raw_fields  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_standardise_and_stage_match_oracle(dtype):
    ops = HipOps(dtype)
    f = raw_fields(n=37, H=33, W=47, seed=5)                    # odd sizes: vector tail of the moments kernel, ragged chunks
    stats = pp.field_stats(ops, f, chunk=8)
    for k in ("u10", "surface_pressure", "t2"):
        a = f[k].astype(np.float64)
        assert abs(stats[k][0] - np.nanmean(a)) <= 1e-9 * abs(np.nanmean(a)) + 1e-12, k
        assert abs(stats[k][1] - np.nanstd(a)) <= 1e-7 * np.nanstd(a), k
    store, _ = pp.stage_standardized(ops, f, ORDER, chunk=8)
    ref = torch.from_numpy(op.stage(op.xr_standardize_all(f), ORDER).transpose(0, 2, 3, 1).copy())
    got = store.float().cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(ref))      # NaNs stay NaNs, nothing else becomes one
    tol = 2e-3 if dtype == "f32" else 2e-2                       # fp32: the reference's own fp32 mean of ~1e5 Pa; bf16: storage rounding
    assert torch.allclose(got.nan_to_num(), ref.nan_to_num(), rtol=0, atol=tol)
    assert torch.equal(got[..., 2], ref[..., 2])                # the land-sea mask passes through untouched
    ld = ResidentLoader.from_fields(f, {"u": f["u10"], "v": f["t2"]}, ORDER, ["u", "v"], batch_size=4, shuffle=True, ops=ops, seed=1)
    xc, xf = next(iter(ld))
    assert xc.nhwc.shape == (4, 33, 47, 16) and xf.nhwc.shape == (4, 33, 47, 16) and float(xc.nhwc[..., 4:].float().abs().sum()) == 0.0
