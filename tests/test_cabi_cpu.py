"""The C-ABI shared library loads without a GPU and exports every symbol include/downgan_hip.h
declares; host-side argument validation returns dg_status codes (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from downgan_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "downgan_hip.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
    assert set(_lib.EXPORTS) == set(syms), set(_lib.EXPORTS) ^ set(syms)
    assert b"gfx950" in lib.dg_version()


def test_planner_rejects_bad_geometry():
    lib = _lib.lib()
    d = (_lib.GGDesc * 4)()
    ok = _lib.ConvGeom(dtype=_lib.DG_BF16, N=1, H=8, W=8, Cin=16, Cout=16, stride=1, cin_real=0, pixel_shuffle=0, ldx=16, ldy=16)
    assert lib.dg_conv3x3_plan(C.byref(ok), 0, d) == 1
    assert lib.dg_conv3x3_plan(C.byref(ok), 1, d) == 1
    bad = [dict(stride=3), dict(Cin=12), dict(Cout=8), dict(dtype=7), dict(H=7, stride=2), dict(pixel_shuffle=1, Cout=32)]
    for kw in bad:
        g = _lib.ConvGeom(dtype=_lib.DG_BF16, N=1, H=8, W=8, Cin=16, Cout=16, stride=1, cin_real=0, pixel_shuffle=0, ldx=16, ldy=16)
        for k, v in kw.items():
            setattr(g, k, v)
        assert lib.dg_conv3x3_plan(C.byref(g), 0, d) < 0, kw
    assert lib.dg_conv3x3_plan(C.byref(ok), 2, d) < 0
    # stride-2 data gradient = 4 parity classes with 1/2/2/4 taps (no zero insertion)
    s2 = _lib.ConvGeom(dtype=_lib.DG_F32, N=1, H=8, W=8, Cin=16, Cout=32, stride=2, pixel_shuffle=0, ldx=16, ldy=32)
    assert lib.dg_conv3x3_plan(C.byref(s2), 1, d) == 4
    assert sorted(d[i].ntaps for i in range(4)) == [1, 2, 2, 4]


def test_null_and_shape_errors_do_not_launch():
    lib = _lib.lib()
    g = _lib.ConvGeom(dtype=_lib.DG_BF16, N=1, H=8, W=8, Cin=16, Cout=16, stride=1, cin_real=0, pixel_shuffle=0, ldx=16, ldy=16)
    assert lib.dg_conv3x3_fwd(C.byref(g), None, None, None, None, None) == -3     # DG_ERR_BAD_ARG
    assert lib.dg_conv3x3_wgrad(C.byref(g), None, None, None, None, None) == -3
    assert lib.dg_linear_fwd(_lib.DG_BF16, None, 0, None, 0, None, 0, 1, 16, 32, None) == -3
    assert lib.dg_adam(None, None, None, None, None, 16, 1e-3, 0.9, 0.99, 1e-8, 1, 1.0, None) == -3
    # debugging aids: argument validation only (no device memory is touched)
    assert lib.dg_deterministic() == 0
    assert lib.dg_set_deterministic_workspace(C.c_void_p(0x1000), 64) == -3          # below the 1 MiB minimum
    assert lib.dg_set_deterministic_workspace(C.c_void_p(0x1008), 1 << 21) == -3     # not 16-byte aligned
    assert lib.dg_set_deterministic_workspace(None, 0) == 0 and lib.dg_deterministic() == 0
    assert lib.dg_count_nonfinite(None, None, None) == -3
    assert lib.dg_count_nonfinite(C.byref(_lib.FiniteBufs(nbuf=9)), C.c_void_p(16), None) == -3


def test_integration_stub_matches_binding():
    """The ctypes stub a reference maintainer would copy from INTEGRATION.md has the same field lists (= ABI layout) as the
    package's own binding of dg_conv_geom / dg_epilogue."""
    import ctypes as C
    import os
    from downgan_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "INTEGRATION.md")).read()
    code = src[src.index("class ConvGeom(C.Structure)"):src.index("# critic.py:25-33")]
    ns = {}
    exec("import ctypes as C\n" + code, ns)
    for name in ("ConvGeom", "Epilogue"):
        mine, theirs = getattr(_lib, name), ns[name]
        assert [f[0] for f in theirs._fields_] == [f[0] for f in mine._fields_], name
        assert C.sizeof(theirs) == C.sizeof(mine), name


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every struct of include/downgan_hip.h (compiled as plain C with gcc) equal the ctypes binding's:
    a silent layout mismatch would hand the kernels garbage."""
    import ctypes as C
    import os
    import shutil
    import subprocess
    from downgan_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [("dg_epilogue", _lib.Epilogue), ("dg_conv_geom", _lib.ConvGeom), ("dg_gg_desc", _lib.GGDesc),
             ("dg_ssim_params", _lib.SsimParams), ("dg_msssim_combine", _lib.MsssimCombine),
             ("dg_f8_operands", _lib.F8Operands), ("dg_field_planes", _lib.FieldPlanes), ("dg_finite_bufs", _lib.FiniteBufs),
             ("dg_exp_batch", _lib.ExpBatch)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{root}/include/downgan_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_product_path_fails_loudly_without_library_or_gpu(tmp_path):
    """No CPU / eager fallback: a missing shared object raises on load, and the op layer refuses to start without a GPU."""
    import subprocess
    import sys
    import torch
    env = dict(os.environ, DG_LIB_OVERRIDE=str(tmp_path / "missing.so"))
    r = subprocess.run([sys.executable, "-c", "from downgan_amd import _lib; _lib.lib()"], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "RuntimeError" in r.stderr and "no CPU or eager fallback" in r.stderr, r.stderr[-400:]
    if not torch.cuda.is_available():
        from downgan_amd.ops import HipOps
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            HipOps("bf16")


def test_oracle_is_only_used_as_the_checker():
    """oracle/ is test infrastructure: the package never imports it; bench.py only inside cpu_baseline(), __graft_entry__ only
    inside smoke()."""
    import ast
    pkg = os.path.join(ROOT, "downgan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)

    def importing_functions(path):
        tree = ast.parse(open(path).read())
        out = set()
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            for n in ast.walk(fn):
                if isinstance(n, (ast.Import, ast.ImportFrom)):
                    names = [a.name for a in n.names] if isinstance(n, ast.Import) else [n.module or ""]
                    if any(x == "oracle" or x.startswith("oracle.") for x in names):
                        out.add(fn.name)
        top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        for n in top:
            names = [a.name for a in n.names] if isinstance(n, ast.Import) else [n.module or ""]
            assert not any(x == "oracle" or x.startswith("oracle.") for x in names), path
        return out
    assert importing_functions(os.path.join(ROOT, "bench.py")) <= {"cpu_baseline"}
    assert importing_functions(os.path.join(ROOT, "__graft_entry__.py")) <= {"smoke"}


def test_hip_runtime_binding_is_checked_in_both_load_orders():
    """The library must be bound to the HIP runtime torch uses (the C ABI is handed torch's streams and device pointers).  torch
    first (what ``_lib.lib()`` does itself): one runtime, loads.  The shared object mapped BEFORE torch (a host process that
    dlopen-ed it, or /opt/rocm's libamdhip64, early): it is bound to /opt/rocm's runtime for good -- ``lib()`` must say so instead
    of leaving every later launch to fail with DG_ERR_LAUNCH (round 3: build() and smoke() in one process)."""
    import subprocess
    import sys
    ok = subprocess.run([sys.executable, "-c",
                         "from downgan_amd import _lib; l = _lib.lib(); m, t = _lib.hip_runtime_binding(l); assert m == t, (m, t); print('bound', m)"],
                        cwd=ROOT, capture_output=True, text=True)
    assert ok.returncode == 0 and "bound" in ok.stdout, ok.stderr[-600:]
    if not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no second HIP runtime on this machine to provoke the mismatch with")
    for first in (f"ctypes.CDLL({_lib.LIB_PATH!r})", "ctypes.CDLL('/opt/rocm/lib/libamdhip64.so', mode=ctypes.RTLD_GLOBAL)"):
        bad = subprocess.run([sys.executable, "-c", f"import ctypes; {first}; from downgan_amd import _lib; _lib.lib()"],
                             cwd=ROOT, capture_output=True, text=True)
        if bad.returncode == 0:       # this torch build resolved to the same runtime object: nothing to complain about
            continue
        assert "RuntimeError" in bad.stderr and "two HIP runtimes in one process" in bad.stderr, (first, bad.stderr[-600:])


def test_merged_stride2_dgrad_respects_the_halo_row_step_limit():
    """A stride-2 data gradient runs as ONE merged launch of the halo kernel (conv_halo.hip, SEG) -- unless a source row is too long
    for that kernel's 24-bit row-step multiplies (Wo * Cout * elemsize >= 8 MiB): then the planner must keep the four per-class
    descriptors, which the row-tiled kernel understands, instead of handing it the merged one (it would write one class only)."""
    lib = _lib.lib()
    geom = lambda **kw: _lib.ConvGeom(**dict(dict(dtype=_lib.DG_BF16, N=1, H=64, W=64, Cin=128, Cout=128, stride=2, cin_real=0,
                                                   pixel_shuffle=0, ldx=128, ldy=128), **kw))
    assert lib.dg_conv3x3_dgrad_launches(C.byref(geom())) == 1
    assert lib.dg_conv3x3_dgrad_launches(C.byref(geom(stride=1))) == 1
    assert lib.dg_conv3x3_dgrad_launches(C.byref(geom(Cin=16, ldx=16))) == 4                         # narrow dx: per-class launches
    assert lib.dg_conv3x3_dgrad_launches(C.byref(geom(W=8190, Cout=1024, ldy=1024))) == 1            # 4095 * 1024 * 2 < 8 MiB
    assert lib.dg_conv3x3_dgrad_launches(C.byref(geom(W=8192, Cout=1024, ldy=1024))) == 4            # 4096 * 1024 * 2 = 8 MiB
    assert lib.dg_conv3x3_dgrad_launches(C.byref(geom(dtype=_lib.DG_F32, W=4096, Cout=1024, ldy=1024))) == 4
