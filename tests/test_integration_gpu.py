"""The INTEGRATION.md drop-in snippet, verbatim, against the HIP kernels (see tests/test_integration_cpu.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_integration_cpu import check, run_snippet

pytestmark = pytest.mark.gpu


def test_integration_snippet_runs_verbatim_on_hip():
    check(run_snippet("cuda:0"))


def test_build_then_smoke_in_one_process():
    """``__graft_entry__.build()`` followed by ``smoke()`` in ONE process (round 3: build() loaded the library before torch, it
    bound /opt/rocm's HIP runtime and smoke()'s first launch returned DG_ERR_LAUNCH).  ``_lib.lib()`` now imports torch first AND
    verifies the binding; this runs the driver's two entry points back to back the way a single-process driver would."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0 and "smoke ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
