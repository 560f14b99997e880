"""bench.py's host-side bookkeeping (no GPU): one named workload per BASELINE.json config, work accounting."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_one_named_workload_per_baseline_config():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    names = ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]
    assert len(base["configs"]) == len(names)
    for i, n in enumerate(names):
        assert n in bench.WORKLOADS and bench.WORKLOAD_INFO[n][0] == f"BASELINE configs[{i}]"
    assert set(bench.WORKLOAD_INFO) == set(bench.WORKLOADS)
    # configs[2]: batch 256 on 8 GPUs; configs[3]: batch 64, 6 channels, 4 GPUs; configs[4]: batch 128 fp8 on 8 GPUs
    assert bench.WORKLOADS["cfg3"][0] * bench.WORKLOAD_INFO["cfg3"][1] == 256 and bench.WORKLOAD_INFO["cfg3"][2] == "bf16"
    assert bench.WORKLOADS["cfg4"][0] * bench.WORKLOAD_INFO["cfg4"][1] == 64 and bench.WORKLOADS["cfg4"][3] == 6
    assert bench.WORKLOADS["cfg5"][0] * bench.WORKLOAD_INFO["cfg5"][1] == 128 and bench.WORKLOAD_INFO["cfg5"][2] == "fp8"
    assert bench.WORKLOADS["cfg2"][:1] == (32,) and bench.WORKLOAD_INFO["cfg2"][1:] == (1, "bf16")
    for n in ("cfg2", "cfg3", "cfg4", "cfg5"):        # the full 128 -> 1024 tile, filters 128, 16 RRDBs
        assert bench.WORKLOADS[n][1:3] == (128, 128) and bench.WORKLOADS[n][4] == 16


def test_algorithmic_flops_of_the_headline_workload():
    """Gf 4203.8, Cf 778.8 GFLOP per sample at configs[1] shapes with real channel counts."""
    gf, cf = bench.conv_flops_per_sample(128, 128, 2, 2, 16)
    assert abs(gf / 1e9 - 4203.8) < 0.5 and abs(cf / 1e9 - 778.8) < 0.5, (gf, cf)
