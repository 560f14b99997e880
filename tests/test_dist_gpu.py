"""Data-parallel path on the GPU: 2 fresh rank processes (torch.distributed.run) against one process with the whole batch.

* ``test_two_nccl_ranks_equal_single_process``: one rank per GPU, gradient exchange over RCCL (``nccl`` backend) --
  needs >= 2 GPUs and is skipped on the one-GPU box; this is the path bench.py --gpus N runs.
* ``test_two_ranks_on_one_gpu_rehearsal``: the same two processes sharing cuda:0 with the exchange over gloo, so the
  multi-process HIP path (flat-buffer all-reduce, deferred updates, sharded metrics) is exercised on every GPU box.
The rank processes are CHILD processes started before they touch the GPU; nothing re-execs an initialised process.
"""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(backend, dtype, outdir, n=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "dist_worker.py"), backend, outdir, dtype]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [torch.load(os.path.join(outdir, f"r{i}.pt")) for i in range(n)]


def _single(dtype):
    import dist_worker as dw
    eng, ops = dw.build(2, 2, dtype, "cuda:0")
    xc, xf, alphas = dw.data(ops, eng, 0, 2, 2)
    return dw.run(eng, xc, xf, alphas)


def _compare(ranks, ref):
    r0, r1 = ranks
    lr, steps = 2.5e-4, 2
    # (critic update pending, generator update pending) after each step.  Step 0 is a generator step: the critic's update was
    # completed by the generator iteration's C(fake), the generator's is parked behind step 1's real-sample pass; after step 1
    # the critic's update is parked behind the next G(coarse): each exchange overlaps with compute that does not need it
    assert r0["pending"] == [(False, True), (True, False)], r0["pending"]
    for k in ref["C"]:
        assert torch.equal(r0["C"][k], r1["C"][k]), k                                  # replicas stay identical
        assert torch.allclose(r0["C"][k], ref["C"][k], rtol=0, atol=2 * lr * steps), (k, float((r0["C"][k] - ref["C"][k]).abs().max()))
        assert float((r0["C"][k] - ref["C"][k]).norm()) <= 2e-4 * float(ref["C"][k].norm()) + 1e-6, k
    for k in ref["G"]:
        assert torch.equal(r0["G"][k], r1["G"][k]), k
        assert torch.allclose(r0["G"][k], ref["G"][k], rtol=0, atol=0.25 * lr), k
    for rec in ("scal0", "scal"):
        for k, v in ref[rec].items():
            assert abs(r0[rec][k] - v) <= 2e-5 * max(1.0, abs(v)), (rec, k, r0[rec][k], v)
            assert r0[rec][k] == r1[rec][k], (rec, k)
    for k in ("MAE", "MSE", "Wass", "MSSSIM"):
        assert r0["metrics"][k] == r1["metrics"][k], k
        assert abs(r0["metrics"][k] - ref["metrics"][k]) <= 1e-5 * max(1.0, abs(ref["metrics"][k])), k


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL ranks cannot share a device)")
def test_two_nccl_ranks_equal_single_process():
    with tempfile.TemporaryDirectory() as d:
        ranks = _launch("nccl", "f32", d)
    _compare(ranks, _single("f32"))


def test_two_ranks_on_one_gpu_rehearsal():
    with tempfile.TemporaryDirectory() as d:
        ranks = _launch("gloo", "f32", d)
    _compare(ranks, _single("f32"))


def _bench(*args):
    """`python bench.py ...` as typed, as a child process (the parent test process has initialised the GPU: never exec from it)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]               # rank 0 prints ONE JSON line, whatever the rank count
    return json.loads(lines[0])


def test_bench_self_launch_rehearsal_line():
    """bench.py's own rank launcher (`--gpus 2` typed bare: the parent starts two fresh rank processes before touching the GPU),
    here as the one-GPU rehearsal: one JSON line, marked as a rehearsal, both ranks reported, finite losses."""
    import math
    d = _bench("--gpus", "2", "--rehearse-on-one-gpu", "--workload", "mid", "--steps", "5", "--warmup", "1", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    assert d["rehearsal"] is True and "REHEARSAL" in d["config"]["exchange"]
    assert d["steps"] == 5 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert [r["rank"] for r in d["ranks_seen"]] == [0, 1] and all(r["device"] == "cuda:0" for r in d["ranks_seen"])
    assert len({r["pid"] for r in d["ranks_seen"]}) == 2
    assert all(math.isfinite(v) for v in d["losses"].values()) and {"critic_loss", "gp_ret", "g_loss"} <= set(d["losses"])
    assert "fp8" not in d and "cpu_baseline" not in d       # single-GPU extras only
    # every deferred exchange of the timed region was queried for completion before its wait (5 critic + 1 generator updates)
    ov = d["exchange_overlap"]
    assert ov["finishes"] == 6 and 0 <= ov["already_complete"] <= ov["finishes"], ov


def test_bench_default_line_carries_fp8_and_event_free_numbers():
    """The default single-GPU command (small workload here) also reports the fp8 mode (BASELINE configs[4]) and a short region
    without per-launch events, and is not marked as a rehearsal."""
    d = _bench("--workload", "mid", "--steps", "5", "--warmup", "1", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["rehearsal"] is False and d["dtype"] == "bf16" and d["ranks_seen"][0]["device"] == "cuda:0"
    assert d["roofline"]["frac"] > 0 and d["critic_conv_stack"]["mfma_frac"] > 0
    assert d["without_kernel_events"]["steps"] == 5 and d["without_kernel_events"]["ms_per_step"] > 0
    f8 = d["fp8"]
    assert f8["steps"] == 5 and f8["value"] > 0 and f8["roofline"]["peak"] == 5000.0 and f8["roofline"]["launches"] > 0
    import math
    assert all(math.isfinite(v) for v in f8["losses"].values())
