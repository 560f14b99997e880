"""The INTEGRATION.md drop-in snippet, verbatim, against the HIP kernels (see tests/test_integration_cpu.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_integration_cpu import check, run_snippet

pytestmark = pytest.mark.gpu


def test_integration_snippet_runs_verbatim_on_hip():
    check(run_snippet("cuda:0"))
