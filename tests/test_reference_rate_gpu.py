"""A second baseline beside bench.py's ``cpu_baseline``: the oracle port (oracle/ref_step.py: the reference's train step
restated over torch.nn.functional, DoWnGAN/GAN/wasserstein.py:27-147) run by PyTorch-ROCm ON THE GPU -- MIOpen
convolutions, autograd with create_graph for the penalty -- which is what the reference itself would execute on an
MI355X.  Timed at BASELINE.json configs[1]'s tile and networks on a small batch (per-sample cost does not depend on the
batch; the autograd graph of batch 32 does not fit), in fp32 (the reference's precision) and under bf16 autocast (the
benchmarked precision of the native path).  Opt-in (DG_TEST_REF_GPU_RATE=1): it is a measurement, not a parity check;
the result goes to gpurun_out/reference_port_on_gpu.json.
"""
import json
import os
import time

import pytest
import torch

from downgan_amd import synthetic
from oracle import ref_step

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, F_, NRB = 128, 128, 16


def _beat(msg):
    """progress line under gpurun_out/ (MIOpen compiles its kernels on a fresh box: the first iteration takes minutes)."""
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "reference_port_on_gpu.progress"), "a") as f:
        f.write(f"{time.strftime('%H:%M:%S')} {msg}\n")


def _heartbeat(stop):
    while not stop.wait(60):
        _beat("still running")


def _time_mode(batch, autocast):
    dev = torch.device("cuda:0")
    pg = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.generator_params(F_, 2, 2, NRB).items()}
    pc = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.critic_params(F_, 8 * S, 2).items()}
    coarse, fine = synthetic.tiles(batch, 2, S)
    tc, tf = torch.from_numpy(coarse).to(dev), torch.from_numpy(fine).to(dev)
    alpha = torch.from_numpy(synthetic.alpha(batch, 0)).to(dev)
    tr = ref_step.OracleTrainer(pg, pc, ref_step.HP(batch_size=batch), num_res_blocks=NRB)

    def run(fn):
        _beat(f"{'bf16' if autocast else 'fp32'} batch {batch}: next iteration")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    run(lambda: tr.critic_iteration(tc, tf, alpha))                      # warm-up (MIOpen solver selection)
    run(lambda: tr.generator_iteration(tc, tf))
    tc_s = [run(lambda: tr.critic_iteration(tc, tf, alpha)) for _ in range(3)]
    tg_s = [run(lambda: tr.generator_iteration(tc, tf)) for _ in range(2)]
    t_c, t_g = min(t for t, _ in tc_s), min(t for t, _ in tg_s)
    scal = tc_s[-1][1][0]
    assert all(v == v and abs(v) != float("inf") for v in scal.values()), scal
    return {"batch": batch, "critic_iteration_s": t_c, "generator_iteration_s": t_g,
            "samples_per_s": batch / (t_c + t_g / 5), "hbm_peak_gib": torch.cuda.max_memory_allocated() / 2 ** 30}


@pytest.mark.skipif(os.environ.get("DG_TEST_REF_GPU_RATE") != "1", reason="measurement, opt-in: DG_TEST_REF_GPU_RATE=1")
def test_reference_port_rate_on_this_gpu():
    out = {"what": "oracle/ref_step.py (the reference's train step over torch.nn.functional) executed by PyTorch-ROCm / MIOpen on "
                   "cuda:0 at configs[1]'s tile (2ch 128x128 -> 1024x1024, F=128, 16 RRDBs); samples/s = batch / (critic + generator/5), "
                   "best of 3 / 2 iterations after a warm-up; the port skips the reference's dead generator backward in the critic iteration",
           "device": torch.cuda.get_device_name(0), "torch": torch.__version__}
    import threading
    stop = threading.Event()
    threading.Thread(target=_heartbeat, args=(stop,), daemon=True).start()
    for name, autocast in (("bf16_autocast", True), ("fp32", False)):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        out[name] = _time_mode(int(os.environ.get("DG_TEST_REF_GPU_BATCH", "2")), autocast)
        print(name, out[name], flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "reference_port_on_gpu.json"), "w") as f:
            json.dump(out, f, indent=1)
    stop.set()
    assert out["fp32"]["samples_per_s"] > 0
