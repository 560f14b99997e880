"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's dataset standardisation and staging.

Parity status: UNPINNED.  The reference's ``xr_standardize_array`` / ``xr_standardize_all``
(DoWnGAN/helpers/gen_experiment_datasets.py:195-233) need xarray, which is absent from the reference tree and from this
image, and the one data file the reference's tests hold (DoWnGAN/GAN/tests/coarse_test.nc) is NetCDF-4/HDF5, unreadable here
(no netCDF4 / h5py; scipy reads NetCDF-3 only) -- so no golden could be generated.  What xarray computes is public:
``da.mean(skipna=True)`` / ``da.std(skipna=True)`` dispatch to numpy's nanmean / nanstd (ddof = 0) over ALL dimensions of the
DataArray, and the result is ``(da - mean) / std`` in the array's dtype.
"""
import numpy as np

EXEMPT = ("land_sea_mask",)      # gen_experiment_datasets.py:208-209


def xr_standardize_array(a: np.ndarray) -> np.ndarray:
    """gen_experiment_datasets.py:195-201."""
    mean = np.nanmean(a)
    std = np.nanstd(a)
    return (a - mean) / std


def xr_standardize_all(fields: dict) -> dict:
    """gen_experiment_datasets.py:203-233 (without the prints / sanity asserts)."""
    return {k: (v if k in EXEMPT else xr_standardize_array(v)) for k, v in fields.items()}


def stage(fields: dict, order) -> np.ndarray:
    """concat_data_arrays (:155-166) + stage.py:28-31: ``to_array()`` stacks the variables first, ``transpose(0, 1)`` makes it
    [time, var, lat, lon]."""
    return np.stack([fields[k] for k in order], 0).transpose(1, 0, 2, 3)
