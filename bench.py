#!/usr/bin/env python3
"""bench.py — train-step samples/sec (G+D+GP) of the native WGAN-GP step on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 is launched by torch.distributed.run,
one rank per GPU over RCCL).  A "step" is one minibatch through the critic iteration plus the
generator iteration when step % 5 == 0 (reference DoWnGAN/GAN/wasserstein.py:131-147; the metrics
pass :140 is excluded); the timed region always opens with a generator step, so K timed steps hold ceil(K/5)
generator iterations whatever --steps / --warmup are.  Workload at N=1 = BASELINE.json configs[1]: batch 32 per GPU, 2-channel
128x128 -> 1024x1024 tiles, filters 128, 16 RRDBs, bf16 storage + bf16 MFMA with fp32 accumulate,
fp32 master weights / Adam.  Inputs are synthetic N(0,1) tiles resident in HBM in native NHWC layout
before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / shared device tensors fail with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (batch per GPU, coarse side, filters, channels, RRDBs)
    "cfg2": (32, 128, 128, 2, 16),          # BASELINE.json configs[1] (and [2] at 8 GPUs: global batch 256)
    "cfg4": (16, 128, 128, 6, 16),          # 6-covariate input, global batch 64 on 4 GPUs
    "cfg1": (4, 16, 16, 2, 16),             # the reference's CPU-runnable plumbing case
    "mid": (4, 64, 64, 2, 4),               # small GPU sanity workload
}


def conv_flops_per_sample(S, F_, cin_p, np_p, nrb, nup=3):
    """Forward GFLOP/sample of generator (Gf) and critic (Cf) convs+linears, padded channels."""
    gf = 2 * 9 * cin_p * F_ * S * S
    gf += nrb * 3 * sum(2 * 9 * k * F_ * F_ * S * S for k in range(1, 6))
    gf += 2 * 9 * F_ * F_ * S * S
    for u in range(nup):
        gf += 2 * 9 * F_ * 4 * F_ * (S << u) ** 2
    hs = S << nup
    gf += 2 * 9 * F_ * F_ * hs * hs + 2 * 9 * F_ * np_p * hs * hs
    widths = [np_p, F_, F_, 2 * F_, 2 * F_, 4 * F_, 4 * F_, 8 * F_, 8 * F_]
    strides = (1, 2, 1, 2, 1, 2, 1, 2)
    cf, h = 0, hs
    for l, st in enumerate(strides):
        h //= st
        cf += 2 * 9 * widths[l] * widths[l + 1] * h * h
    cf += 2 * (8 * F_ * h * h) * 100 + 2 * 100
    return gf, cf


def host_cores():
    """CPUs this process may actually use: min(os.cpu_count, affinity mask, cgroup v2 cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(args, wl):
    """The CPU oracle (a port: oracle/ref_step.py, PyTorch-CPU fp32) timed on this box's host cores on a
    BOUNDED sample of the same workload.  The full 128->1024 tile costs ~25 TFLOP per sample-step, minutes
    per sample on host cores (torch's CPU double-backward of conv runs at ~0.07 TFLOP/s), so the sample is
    the workload's own networks (same filters / RRDB count) on a tile 2x smaller per side, batch 1: one
    critic iteration + one generator iteration.  Every conv/linear flop scales with the tile area, so
    samples/s at the full tile = sample rate / area ratio; this is stated in `sample`."""
    import torch
    from downgan_amd import synthetic
    from oracle import ref_step
    B, S, F_, cin, nrb = wl
    shrink = 2 if S >= 64 else 1
    Ss = S // shrink
    cores = host_cores()
    torch.set_num_threads(cores)
    pg = {k: torch.from_numpy(v) for k, v in synthetic.generator_params(F_, cin, 2, nrb).items()}
    pc = {k: torch.from_numpy(v) for k, v in synthetic.critic_params(F_, 8 * Ss, 2).items()}
    tr = ref_step.OracleTrainer(pg, pc, ref_step.HP(batch_size=1), num_res_blocks=nrb)
    coarse, fine = synthetic.tiles(1, cin, Ss)
    coarse, fine = torch.from_numpy(coarse), torch.from_numpy(fine)
    alpha = torch.from_numpy(synthetic.alpha(1, 0))
    t0 = time.perf_counter()
    tr.critic_iteration(coarse, fine, alpha)
    tc = time.perf_counter() - t0
    t0 = time.perf_counter()
    tr.generator_iteration(coarse, fine)
    tg = time.perf_counter() - t0
    area = shrink * shrink
    sample = (f"oracle port, batch 1, {cin}ch {Ss}x{Ss}->{8 * Ss}x{8 * Ss} tile with the workload's networks (F={F_}, {nrb} RRDBs), "
              f"fp32: 1 critic iteration {tc:.1f}s + 1 generator iteration {tg:.1f}s; step time = critic + generator/5; "
              f"scaled by the tile-area ratio {area}x to the {S}->{8 * S} tile (all conv/linear flops scale with area)")
    return {"value": 1.0 / ((tc + tg / 5.0) * area), "unit": "samples/s", "cores": cores, "kind": "port", "sample": sample}


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 read correction).  PMC counters cannot be
    collected from inside the process, so the number is read from profiles/ and is null for any other workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_cfg2_bf16.json")
    if args.workload != "cfg2" or args.dtype != "bf16" or args.batch or not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    for k, v in d["kernels"].items():
        if k.startswith(kernel):
            return round(v["traffic_bytes_per_launch"]), "profiles/pmc_traffic_cfg2_bf16.json (" + d["formula"] + ")"
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="diagnostic: split the `kernels` table by layer geometry (with bytes)")
    ap.add_argument("--graphs", action="store_true", help="replay the iterations as captured HIP graphs (small, launch-bound tiles)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and the gradient exchange goes over gloo")
    args = ap.parse_args()

    import torch
    from downgan_amd import synthetic
    from downgan_amd.dist import Dist
    from downgan_amd.engine import HyperParams, TrainEngine
    from downgan_amd.ops import HipOps

    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dist = Dist("gloo" if args.rehearse_on_one_gpu else "nccl") if world > 1 else None
    rank = dist.rank if dist else 0
    local = 0 if args.rehearse_on_one_gpu else (dist.local_rank if dist else 0)
    torch.cuda.set_device(local)
    B, S, F_, cin, nrb = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    ops = HipOps(args.dtype, f"cuda:{local}")
    eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B * world), num_res_blocks=nrb, dist=dist)
    eng.G.load_state_dict(synthetic.generator_params(F_, cin, 2, nrb))       # seed 0 on every rank
    eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
    coarse, fine = synthetic.tiles(B, cin, S, rank=rank)                      # seed 1234 + rank
    xc = ops.zeros(B, S, S, eng.G.cin_p); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
    xf = ops.zeros(B, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
    del coarse, fine
    nsteps = args.warmup + args.steps
    alphas = [torch.from_numpy(synthetic.alpha(B, s, rank=rank)).cuda() for s in range(nsteps)]

    if args.graphs:
        eng.enable_graphs(xc, xf)
        args.no_kernel_timing = True          # per-launch events cannot be recorded inside a replayed graph
    def complete_updates():
        # the critic's Adam is deferred behind the next generator forward (engine.py); finish it so that exactly the work
        # of the steps issued so far lies before the synchronisation point
        eng.C.P.sync(); eng.G.P.sync()

    for s in range(args.warmup):
        eng.train_step(xc, xf, alphas[s])
    # The generator runs every `critic_iterations`-th step.  Align the step counter so that the timed region OPENS with a
    # generator step: K timed steps then contain ceil(K / critic_iterations) generator iterations for every --steps /
    # --warmup combination -- never fewer than the long-run share (a 3-step region after 2 warm-up steps would otherwise
    # hold none and read 25 % too fast).
    ci = eng.hp.critic_iterations
    eng.num_steps = -(-eng.num_steps // ci) * ci
    gen_steps_timed = -(-args.steps // ci)
    complete_updates()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_kernel_timing:
        ops.prof = []
        ops.per_layer = args.per_layer
    t0 = time.perf_counter()
    for s in range(args.warmup, nsteps):
        eng.train_step(xc, xf, alphas[s])
    complete_updates()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        t = t if args.rehearse_on_one_gpu else t.cuda()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    scal = eng.read_scalars(True)

    roofline = None
    critic_stack = None
    kernels = {}
    if ops.prof:
        agg = {}
        for tag, flops, nbytes, s_ev, e_ev in ops.prof:
            a = agg.setdefault(tag, [0.0, 0.0, 0, 0.0])
            a[0] += flops; a[1] += s_ev.elapsed_time(e_ev) * 1e-3; a[2] += 1; a[3] += nbytes
        for tag, (fl, sec, n, nb_) in agg.items():
            kernels[tag] = {"launches": n, "seconds": round(sec, 4), "tflops": round(fl / sec / 1e12, 2) if sec > 0 else None}
            if args.per_layer:
                kernels[tag]["alg_tbps"] = round(nb_ / sec / 1e12, 2) if sec > 0 else None
        # dominant kernel: gg_halo4w_kernel (conv forward of both strides + data gradients; two instantiations).  Only calls that were
        # served by that kernel alone (tag suffix k8) are counted, so achieved = algorithmic flops of those
        # launches / their summed launch durations, and launches/seconds give the average launch duration.
        halo = [t for t in agg if t.endswith(":k8")]
        fl = sum(agg[t][0] for t in halo)
        sec = sum(agg[t][1] for t in halo)
        nl = sum(agg[t][2] for t in halo)
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        ach = fl / sec / 1e12 if sec > 0 else 0.0
        alg_bytes = sum(agg[t][3] for t in halo) / max(nl, 1)
        traffic, traffic_src = pmc_traffic("gg_halo4w_kernel", args)
        roofline = {"kernel": "gg_halo4w_kernel (implicit-GEMM conv3x3: forward, stride 1 and 2, and data gradients)", "bound": "mfma",
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "alg_bytes_per_launch": round(alg_bytes),
                    "launches": nl, "avg_launch_ms": round(sec / max(nl, 1) * 1e3, 4),
                    "share_of_step": round(sec / elapsed, 3)}
        # critic conv stack (BASELINE target: >= 40 % MFMA utilisation): every conv launch of the critic (forward, data- and
        # weight-gradient of its 8 conv layers, incl. the HBM-bound 2-channel first layer), algorithmic flops / summed durations
        cf = sum(v[0] for t, v in agg.items() if t.startswith("conv") and ":C" in t)
        cs = sum(v[1] for t, v in agg.items() if t.startswith("conv") and ":C" in t)
        critic_stack = {"tflops": round(cf / cs / 1e12, 2) if cs > 0 else None, "mfma_frac": round(cf / cs / 1e12 / peak, 4) if cs > 0 else None,
                        "share_of_step": round(cs / elapsed, 3)}
        ops.prof = None

    if rank == 0:
        gf, cf = conv_flops_per_sample(S, F_, eng.G.cin_p, eng.G.np_p, nrb)
        value = args.steps * B * world / elapsed
        w_step = (1.6 * gf + 10.4 * cf)
        out = {
            "metric": "train-step samples/sec (G+D+GP)", "value": round(value, 4), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: batch {B}/GPU (global {B * world}), {cin}ch {S}x{S}->{8 * S}x{8 * S}, "
                                   f"filters {F_}, {nrb} RRDBs, WGAN-GP critic step every step + generator step every 5th",
                       "parallelism": f"dp{world}", "alg_tflop_per_sample_step": round(w_step / 1e12, 4),
                       "generator_steps_in_timed_region": gen_steps_timed},
            "step_mfma_frac": round(w_step * value / world / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 4),
            "losses": {k: scal[k] for k in ("critic_loss", "gp_ret", "g_loss") if k in scal},
            "roofline": roofline, "critic_conv_stack": critic_stack, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, (B, S, F_, cin, nrb))
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()


if __name__ == "__main__":
    main()
