"""Dataset standardisation + staging: oracle restatement vs a float64 evaluation, and the native host
logic (chunked two-pass moments, exemption of the land-sea mask, variable order, ResidentLoader.from_fields) on the emulated
ops.  The kernels themselves are compared with the oracle in tests/test_preprocess_gpu.py."""
import numpy as np
import torch

from oracle import preprocess as op


def raw_fields(n=10, H=12, W=20, seed=0):
    g = np.random.default_rng(seed)
    f = {
        "u10": (g.normal(2.0, 5.0, (n, H, W))).astype(np.float32),
        "surface_pressure": (g.normal(101325.0, 900.0, (n, H, W))).astype(np.float32),      # large offset: needs double moments
        "land_sea_mask": (g.random((n, H, W)) > 0.5).astype(np.float32),
        "t2": (g.normal(280.0, 12.0, (n, H, W))).astype(np.float32),
    }
    f["t2"][3, 4, 5] = np.nan                                                                   # skipna
    return f


ORDER = ["u10", "t2", "land_sea_mask", "surface_pressure"]


def test_oracle_standardisation_is_zscore_with_mask_exempt():
    f = raw_fields()
    z = op.xr_standardize_all(f)
    assert np.array_equal(z["land_sea_mask"], f["land_sea_mask"])
    for k in ("u10", "surface_pressure", "t2"):
        a = f[k].astype(np.float64)
        ref = (a - np.nanmean(a)) / np.nanstd(a)
        assert np.allclose(z[k], ref, rtol=0, atol=2e-3 if k == "surface_pressure" else 2e-5, equal_nan=True)
        assert abs(np.nanmean(z[k])) < 1e-2 and abs(np.nanstd(z[k]) - 1) < 1e-1          # the reference's own sanity asserts (:219-228)
    st = op.stage(z, ORDER)
    assert st.shape == (10, 4, 12, 20) and np.array_equal(st[:, 2], f["land_sea_mask"])


def test_native_host_logic_on_emulated_ops():
    from downgan_amd.GAN import preprocess as pp
    from downgan_amd.GAN.dataloader import ResidentLoader
    from oracle.emu_ops import EmuOps
    ops = EmuOps("f32")
    f = raw_fields()
    stats = pp.field_stats(ops, f, chunk=3)                      # ragged chunks
    assert stats["land_sea_mask"] == (0.0, 1.0)
    for k in ("u10", "surface_pressure", "t2"):
        a = f[k].astype(np.float64)
        assert abs(stats[k][0] - np.nanmean(a)) <= 1e-9 * abs(np.nanmean(a)) + 1e-12
        assert abs(stats[k][1] - np.nanstd(a)) <= 1e-7 * np.nanstd(a)
    store, _ = pp.stage_standardized(ops, f, ORDER, chunk=4)
    ref = op.stage(op.xr_standardize_all(f), ORDER).transpose(0, 2, 3, 1)        # [n, H, W, c]
    got = store.numpy()
    assert got.shape == ref.shape
    assert np.allclose(got, ref, rtol=0, atol=2e-3, equal_nan=True) and np.allclose(got[..., :3], ref[..., :3], atol=2e-5, equal_nan=True)
    fine = {"u": f["u10"], "v": f["t2"]}
    ld = ResidentLoader.from_fields(f, fine, ORDER, ["u", "v"], batch_size=2, shuffle=False, ops=ops)
    xc, xf = next(iter(ld))
    assert xc.shape == (2, 4, 12, 20) and xf.shape == (2, 2, 12, 20)
    assert np.allclose(xc.nhwc[..., :4].numpy(), ref[:2], atol=2e-3, equal_nan=True) and float(xc.nhwc[..., 4:].abs().sum()) == 0.0
    # a test split re-uses the train statistics
    ld2 = ResidentLoader.from_fields(f, fine, ORDER, ["u", "v"], batch_size=2, shuffle=False, ops=ops, stats=ld.stats)
    assert torch.equal(ld2.store_c.nan_to_num(), ld.store_c.nan_to_num())


def test_degenerate_fields_raise_and_whole_record_statistics():
    """A constant non-exempt field or a field without a finite value raises instead of dividing by zero (the reference would write
    NaN / Inf); `whole_record_stats` = statistics over train + test together, the reference's order of operations
    (gen_experiment_datasets.py:241-250: standardise, then split)."""
    import pytest
    from downgan_amd.GAN import preprocess as pp
    from oracle.emu_ops import EmuOps
    ops = EmuOps("f32")
    f = raw_fields()
    const = dict(f, u10=np.full_like(f["u10"], 3.5))
    with pytest.raises(ValueError, match="constant"):
        pp.field_stats(ops, const)
    nan = dict(f, t2=np.full_like(f["t2"], np.nan))
    with pytest.raises(ValueError, match="no finite value"):
        pp.field_stats(ops, nan)
    train = {k: v[:6] for k, v in f.items()}
    test = {k: v[6:] for k, v in f.items()}
    both = pp.whole_record_stats(ops, train, test, chunk=4)
    whole = pp.field_stats(ops, f)
    for k in f:
        assert abs(both[k][0] - whole[k][0]) <= 1e-9 * abs(whole[k][0]) + 1e-12 and abs(both[k][1] - whole[k][1]) <= 1e-9 * whole[k][1] + 1e-12
