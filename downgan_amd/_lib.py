"""ctypes binding of libdowngan_hip.so (the C ABI declared in include/downgan_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C downgan_amd/csrc``.
There is no fallback: if the shared object is missing, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DG_LIB_OVERRIDE") or os.path.join(_HERE, "csrc", "libdowngan_hip.so")   # override: A/B builds in tools/

DG_F32, DG_BF16 = 0, 1
STATUS = {0: "DG_OK", -1: "DG_ERR_BAD_SHAPE", -2: "DG_ERR_BAD_DTYPE", -3: "DG_ERR_BAD_ARG", -4: "DG_ERR_LAUNCH"}


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("has_act", C.c_int), ("act_slope", C.c_float),
                ("r1", C.c_void_p), ("ldr1", C.c_int64), ("s1", C.c_float),
                ("r2", C.c_void_p), ("ldr2", C.c_int64), ("s2", C.c_float),
                ("mask", C.c_void_p), ("ldmask", C.c_int64), ("mask_slope", C.c_float),
                ("accumulate", C.c_int), ("mask_bits", C.c_void_p), ("out_bits", C.c_void_p),
                ("mask_c0", C.c_int), ("mask_last", C.c_int), ("out_q", C.c_void_p), ("out_qs", C.c_void_p), ("ldqs", C.c_int64),
                ("out_u", C.c_void_p), ("out_ue", C.c_void_p), ("skip_y", C.c_int), ("out_amax", C.c_void_p)]


class ConvGeom(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("Cin", C.c_int), ("Cout", C.c_int), ("stride", C.c_int), ("cin_real", C.c_int), ("pixel_shuffle", C.c_int),
                ("ldx", C.c_int64), ("ldy", C.c_int64)]


class GGDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("Hs", C.c_int), ("Ws", C.c_int), ("Cred", C.c_int),
                ("lds", C.c_int64), ("src_ps", C.c_int),
                ("Hg", C.c_int), ("Wg", C.c_int), ("sy_mul", C.c_int), ("sx_mul", C.c_int),
                ("ntaps", C.c_int), ("tap_dy", C.c_int * 9), ("tap_dx", C.c_int * 9), ("tap_w", C.c_int * 9),
                ("Nout", C.c_int), ("ldw", C.c_int64),
                ("Hd", C.c_int), ("Wd", C.c_int), ("ldd", C.c_int64),
                ("dy_mul", C.c_int), ("dx_mul", C.c_int), ("dy_off", C.c_int), ("dx_off", C.c_int),
                ("dst_ps", C.c_int)]


class SsimParams(C.Structure):
    _fields_ = [("win", C.c_int), ("g", C.c_float * 11), ("C1", C.c_float), ("C2", C.c_float)]


class MsssimCombine(C.Structure):
    _fields_ = [("inv_count", C.c_float * 5), ("weight", C.c_float * 5)]


class F8Operands(C.Structure):
    _fields_ = [("xq", C.c_void_p), ("xs", C.c_void_p), ("ldxq", C.c_int64), ("ldxs", C.c_int64), ("wq", C.c_void_p), ("ws", C.c_void_p)]


MAX_FIELDS = 8


class FieldPlanes(C.Structure):
    _fields_ = [("plane", C.c_void_p * MAX_FIELDS), ("mean", C.c_float * MAX_FIELDS), ("inv_std", C.c_float * MAX_FIELDS), ("c", C.c_int)]


EXP_BATCH_MAX = 16


class ExpBatch(C.Structure):
    _fields_ = [("scales", C.c_void_p * EXP_BATCH_MAX), ("rows", C.c_int64 * EXP_BATCH_MAX), ("ld", C.c_int64 * EXP_BATCH_MAX),
                ("nblocks", C.c_int * EXP_BATCH_MAX), ("out", C.c_void_p * EXP_BATCH_MAX), ("n", C.c_int)]


FINITE_MAX = 8


class FiniteBufs(C.Structure):
    _fields_ = [("ptr", C.c_void_p * FINITE_MAX), ("n", C.c_int64 * FINITE_MAX), ("dtype", C.c_int * FINITE_MAX), ("nbuf", C.c_int)]


MINMAX_PARTS = 256
_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_PROTOS = {
    "dg_conv3x3_fwd": [C.POINTER(ConvGeom), C.POINTER(Epilogue), _vp, _vp, _vp, _vp],
    "dg_conv3x3_dgrad": [C.POINTER(ConvGeom), C.POINTER(Epilogue), _vp, _vp, _vp, _vp],
    "dg_conv3x3_wgrad": [C.POINTER(ConvGeom), _vp, _vp, _vp, _vp, _vp],
    "dg_conv3x3_wgrad_dense": [_vp, _i, _vp, _vp, _vp, _vp, _vp],
    "dg_conv3x3_wgrad_f8": [C.POINTER(ConvGeom), _vp, _vp, _vp, _vp, _vp, _vp],
    "dg_exp_from_amax": [_vp, _i, _i, _vp, _vp],
    "dg_conv3x3_wgrad_dense_f8": [C.POINTER(ConvGeom), _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "dg_gather_gemm": [C.POINTER(GGDesc), C.POINTER(Epilogue), _vp, _vp, _vp, _vp],
    "dg_conv3x3_plan": [C.POINTER(ConvGeom), _i, C.POINTER(GGDesc)],
    "dg_last_conv_kernels": [],
    "dg_conv3x3_dgrad_launches": [C.POINTER(ConvGeom)],
    "dg_colsum": [_i, _vp, _i64, _i64, _i64, _i64, _i, _vp, _vp],
    "dg_colsum_multi": [_i, _vp, _i64, _i64, _i, _i, _vp, _vp],
    "dg_repack_conv_weights": [_i, _i, _vp, _vp, _i, _i, _vp],
    "dg_repack_dense_dgrad": [_i, _vp, _i, _i, _vp, _vp],
    "dg_wgrad_unswap": [_vp, _vp, _i, _i, _vp],
    "dg_linear_fwd": [_i, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i64, _vp],
    "dg_linear_dx": [_i, _i, _vp, _i, _vp, _i64, _vp, _i64, _vp, _i64, _f, _i, _i, _i64, _vp],
    "dg_linear_dw": [_i, _vp, _i, _vp, _i64, _vp, _i64, _i, _i, _i64, _vp],
    "dg_linear_dw_wide": [_i, _vp, _i, _vp, _i64, _vp, _i64, _i, _i, _i64, _i, _vp],
    "dg_bias_act": [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _i, _f, _vp],
    "dg_mask_mul": [_i, _vp, _i64, _vp, _i64, _i64, _i, _f, _vp],
    "dg_axpby": [_i, _vp, _i64, _vp, _i64, _f, _vp, _i64, _f, _i64, _i, _vp],
    "dg_gp_interp": [_i, _vp, _vp, _vp, _vp, _i, _i64, _vp],
    "dg_gp_interp_c2": [_i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i64, _vp],
    "dg_scale_rows_c2": [_i, _vp, _i64, _vp, _vp, _i, _i64, _vp],
    "dg_sumsq_rows": [_i, _vp, _i, _i64, _vp, _vp],
    "dg_gp_finish": [_vp, _i, _i, _f, _f, _vp, _vp, _vp],
    "dg_scale_rows": [_i, _vp, _vp, _vp, _i, _i64, _vp],
    "dg_l1": [_i, _vp, _i64, _vp, _i64, _i64, _i, _vp, _vp, _i64, _f, _vp, _i64, _vp],
    "dg_sqdiff": [_i, _vp, _i64, _vp, _i64, _i64, _i, _vp, _vp],
    "dg_sum_strided": [_vp, _i, _i, _f, _vp, _vp],
    "dg_fill_col": [_vp, _i, _i, _i, _f, _vp],
    "dg_adam": [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f, _vp],
    "dg_nchw_to_nhwc": [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "dg_nhwc_to_nchw": [_i, _vp, _i64, _vp, _i, _i, _i, _i, _vp],
    "dg_cast": [_i, _vp, _vp, _i64, _vp],
    "dg_minmax_partial": [_i, _vp, _i64, _i64, _i, _vp, _vp],
    "dg_minmax_finish": [_vp, _i, _vp, _vp],
    "dg_normalise_planar": [_i, _vp, _i, _i, _i, _i64, _i, _vp, _vp, _vp],
    "dg_ssim_level": [_vp, _vp, _i, _i, _i, C.POINTER(SsimParams), _vp, _vp],
    "dg_avgpool2": [_vp, _vp, _i, _i, _i, _vp],
    "dg_msssim_finish": [_vp, _i, _i, C.POINTER(MsssimCombine), _vp, _vp],
    "dg_div_vort_sums": [_i, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp],
    "dg_gather_samples": [_i, _vp, _i64, _i, _vp, _i, _vp, _i, _vp],
    "dg_quant_mxfp8": [_i, _vp, _i64, _i64, _i, _vp, _i64, _vp, _i64, _vp],
    "dg_block_exp_max": [_vp, _i64, _i64, _i, _i, _vp, _vp, _vp],
    "dg_block_exp_max_batch": [C.POINTER(ExpBatch), _i, _vp, _vp],
    "dg_quant_uniform": [_i, _vp, _i64, _i64, _i, _vp, _vp, _i64, _vp],
    "dg_conv3x3_fwd_f8": [C.POINTER(ConvGeom), C.POINTER(Epilogue), C.POINTER(F8Operands), _vp, _vp],
    "dg_conv3x3_dgrad_f8": [C.POINTER(ConvGeom), C.POINTER(Epilogue), C.POINTER(F8Operands), _vp, _vp],
    "dg_moments": [_vp, _i64, _vp, _vp],
    "dg_stage_fields": [_i, C.POINTER(FieldPlanes), _i64, _vp, _vp],
    "dg_lowpass5": [_i, _vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp],
    "dg_lowpass5_adjoint": [_i, _vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp],
    "dg_set_deterministic_workspace": [_vp, _i64],
    "dg_deterministic": [],
    "dg_count_nonfinite": [C.POINTER(FiniteBufs), _vp, _vp],
}
EXPORTS = ["dg_version"] + list(_PROTOS)

_lib = None


def _object_of(handle, symbol="hipGetDeviceCount"):
    """Path of the mapped shared object that ``symbol`` resolves to when looked up through ``handle`` (dlsym searches the object
    and its dependencies, so for a library linked against libamdhip64 this names the HIP runtime its kernels register with)."""
    try:
        addr = C.cast(getattr(handle, symbol), C.c_void_p).value
    except AttributeError:
        return None
    with open("/proc/self/maps") as f:
        for line in f:
            parts = line.split(None, 5)
            if len(parts) < 6:
                continue
            lo, hi = (int(x, 16) for x in parts[0].split("-"))
            if lo <= addr < hi:
                return os.path.realpath(parts[5].strip())
    return None


def hip_runtime_binding(l=None):
    """(runtime libdowngan_hip.so is bound to, runtime torch's HIP libraries are bound to); either may be None when it cannot be
    determined (a torch build without HIP libraries)."""
    import torch
    mine = _object_of(l if l is not None else C.CDLL(LIB_PATH))
    c10 = os.path.join(os.path.dirname(torch.__file__), "lib", "libc10_hip.so")
    theirs = _object_of(C.CDLL(c10)) if os.path.exists(c10) else None
    return mine, theirs


def lib():
    """Load (once) and return the shared library; raise loudly when it has not been built, or when it is bound to another HIP
    runtime than torch's."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C downgan_amd/csrc`. downgan_amd has no CPU or eager fallback.")
        # torch first: the library's kernels must register with the HIP runtime torch brings along (its own libamdhip64), the one
        # whose streams and device pointers the C ABI is handed.  Loaded before torch, the library binds /opt/rocm's copy and every
        # launch on a torch stream fails (seen as DG_ERR_LAUNCH from the first kernel when build() and smoke() share a process).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        # ... and verified: a host process that mapped libdowngan_hip.so (or another libamdhip64 with the same soname) BEFORE torch
        # has bound the library to that runtime for good -- say so here instead of failing every launch with DG_ERR_LAUNCH later
        mine, theirs = hip_runtime_binding(l)
        if mine is not None and theirs is not None and mine != theirs:
            raise RuntimeError(
                f"libdowngan_hip.so is bound to the HIP runtime {mine}, torch to {theirs}: two HIP runtimes in one process. The "
                "library's kernels are registered with the first, while the streams and device pointers it is handed belong to the "
                "second, so every launch would fail (DG_ERR_LAUNCH). Import torch (or downgan_amd, which does) before anything that "
                "loads libdowngan_hip.so or /opt/rocm's libamdhip64 into this process.")
        l.dg_version.restype = C.c_char_p
        l.dg_version.argtypes = []
        for name, args in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = C.c_int
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {STATUS.get(rc, rc)}")
