"""The one place where the Python mirrors obtain their compute backend: the HIP kernels behind the C ABI.

There is no other backend in the product (``HipOps`` raises without a GPU or without the built library).  The indirection
exists so that CPU tests of the HOST logic of the mirrors (tests/test_integration_cpu.py) can substitute the op-contract
emulation of ``oracle/emu_ops.py`` by patching ``make_ops``; nothing in ``downgan_amd`` ever does.
"""
from __future__ import annotations

from .ops import HipOps


def make_ops(dtype, device):
    """dtype: "bf16" (default precision of the mirrors), "f32" (exact-fp32 parity mode), "fp8" (bf16 + the MXFP8 conv path for the
    critic's wide convs and the generator trunk's forward, BASELINE configs[4]) or "fp8c" (critic only)."""
    if dtype in ("fp8", "fp8c"):
        return HipOps("bf16", device, f8_critic=True, f8_generator=dtype == "fp8")
    return HipOps(dtype, device)
