#!/usr/bin/env python3
"""bench.py — train-step samples/sec (G+D+GP) of the native WGAN-GP step on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  One command per BASELINE config: `--workload cfg1|cfg2|cfg3|cfg4|cfg5`
(= configs[0..4]; cfg3 is quoted on --gpus 8, cfg4 on --gpus 4, cfg5 on --gpus 8 and defaults to --dtype fp8).  For N>1 either the driver starts the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, WORLD_SIZE set) or, typed bare, this
script starts them itself (`self_launch`: N fresh child processes before any GPU call); one rank per GPU over RCCL.  A "step" is one minibatch through the critic iteration plus the
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

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0, "fp8c": 5000.0}   # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (batch per GPU, coarse side, filters, channels, RRDBs)
    "cfg2": (32, 128, 128, 2, 16),          # BASELINE.json configs[1]: batch 32, 1x MI355X, bf16 (the headline)
    "cfg3": (32, 128, 128, 2, 16),          # BASELINE.json configs[2]: batch 256 = 32/GPU x 8 GPUs, RCCL gradient all-reduce
    "cfg4": (16, 128, 128, 6, 16),          # BASELINE.json configs[3]: 6-covariate input, batch 64 = 16/GPU x 4 GPUs
    "cfg5": (16, 128, 128, 2, 16),          # BASELINE.json configs[4]: fp8 conv path, batch 128 = 16/GPU x 8 GPUs
    "cfg1": (4, 16, 16, 2, 16),             # BASELINE.json configs[0]: the reference's CPU-runnable plumbing case
    "mid": (4, 64, 64, 2, 4),               # small GPU sanity workload
}
# one unambiguous command per BASELINE config: the rank count and precision each named workload is quoted on (--gpus / --dtype
# still override; `config.workload` says what actually ran)
WORKLOAD_INFO = {
    "cfg1": ("BASELINE configs[0]", 1, "bf16"), "cfg2": ("BASELINE configs[1]", 1, "bf16"), "cfg3": ("BASELINE configs[2]", 8, "bf16"),
    "cfg4": ("BASELINE configs[3]", 4, "bf16"), "cfg5": ("BASELINE configs[4]", 8, "fp8"), "mid": ("sanity workload, not a BASELINE config", 1, "bf16"),
}


def conv_flops_per_sample(S, F_, cin_p, np_p, nrb, nup=3):
    """Forward flops/sample of generator (Gf) and critic (Cf) convs + linears.  Call it with the REAL channel
    counts of the reference layers (2 / 6 inputs, 2 predictands, 100 FC rows) for the algorithmic figures every fraction is
    quoted on (cfg2: Gf 4203.8, Cf 778.8 GFLOP), or with the padded ones for the flops the kernels actually execute."""
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args, wl):
    """The CPU oracle (a port: oracle/ref_step.py, PyTorch-CPU fp32, all host cores) timed per SURVEY.md 8(d)(ii): the
    workload's OWN tile and networks (cfg2: 2ch 128x128 -> 1024x1024, F=128, 16 RRDBs) at batch 1 -- one warm-up critic
    iteration, then `critic_iterations` critic iterations + 1 generator iteration = the reference's 5-step cycle
    (wasserstein.py:131-147).  samples/s = 5 samples-steps / that time; linear in the batch (B=32 on host cores would be
    ~475 TFLOP per step), stated as such.  On a slow host the timed critic iterations are cut to fit ~2 minutes."""
    import torch
    from downgan_amd import synthetic
    from oracle import ref_step
    B, S, F_, cin, nrb = wl
    cores = host_cores()
    torch.set_num_threads(cores)
    pg = {k: torch.from_numpy(v) for k, v in synthetic.generator_params(F_, cin, 2, nrb).items()}
    pc = {k: torch.from_numpy(v) for k, v in synthetic.critic_params(F_, 8 * S, 2).items()}
    tr = ref_step.OracleTrainer(pg, pc, ref_step.HP(batch_size=1), num_res_blocks=nrb)
    coarse, fine = synthetic.tiles(1, cin, S)
    coarse, fine = torch.from_numpy(coarse), torch.from_numpy(fine)
    t0 = time.perf_counter()
    tr.critic_iteration(coarse, fine, torch.from_numpy(synthetic.alpha(1, 0)))            # warm-up (allocator, thread pool)
    tw = time.perf_counter() - t0
    ci = 5
    n_crit = ci if tw < 30.0 else max(1, min(ci, int(100.0 / tw)))
    t0 = time.perf_counter()
    for i in range(n_crit):
        tr.critic_iteration(coarse, fine, torch.from_numpy(synthetic.alpha(1, 1 + i)))
    tc = (time.perf_counter() - t0) / n_crit
    t0 = time.perf_counter()
    tr.generator_iteration(coarse, fine)
    tg = time.perf_counter() - t0
    sample = (f"oracle port (oracle/ref_step.py, PyTorch-CPU fp32), {cores} threads on '{cpu_model()}', the workload's own tile and networks "
              f"at batch 1 ({cin}ch {S}x{S}->{8 * S}x{8 * S}, F={F_}, {nrb} RRDBs): warm-up critic iteration {tw:.1f}s, then "
              f"{n_crit} critic iterations at {tc:.1f}s each + 1 generator iteration {tg:.1f}s; value = 1 / (critic + generator/5) samples/s "
              f"(the reference's 5-step cycle; per-sample cost is independent of the batch, so B={B} is this rate, not {B}x it)")
    return {"value": 1.0 / (tc + tg / ci), "unit": "samples/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port", "sample": sample}


def lib_sha256():
    import hashlib
    from downgan_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this same command (tools/pmc_summary.py: separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950 read correction).  PMC counters cannot be collected from inside the process, so the
    number comes from profiles/ -- and only when that file was produced with THIS build of the library (it records the
    sha256 of libdowngan_hip.so): after any kernel change the entry is null until the passes are re-run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"pmc_traffic_{args.workload}_{args.dtype}.json")
    if args.batch or not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    if d.get("lib_sha256") != lib_sha256():
        return None, f"profiles/{os.path.basename(path)} is from another build of libdowngan_hip.so (stale): traffic withheld"
    # every instantiation of the kernel (stride 1, stride 2 with 4 / 8 waves, pixel-shuffled source), weighted by its launches:
    # the same set of launches `alg_bytes_per_launch` and `frac` are averaged over
    inst = {k: v for k, v in d["kernels"].items() if k.startswith(kernel + "<")}
    n = sum(v["launches"] for v in inst.values())
    if not n:
        return None, None
    traffic = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in inst.values()) / n
    src = (f"profiles/{os.path.basename(path)} (" + d["formula"] + f"; launch-weighted mean over the {len(inst)} instantiations "
           + ", ".join(f"{k[len(kernel):]} x{v['launches']}" for k, v in inst.items()) + ")")
    return round(traffic), src


def self_launch(n):
    """Start `n` rank processes of this script (one per GPU, RCCL rendezvous on 127.0.0.1) and return their exit code.
    Runs BEFORE anything initialises the GPU in this process; the children are new processes, nothing is re-exec'ed."""
    import socket
    import subprocess
    with socket.socket() as sk:          # (downgan_amd.dist.free_port, inlined: this parent process must not import torch)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def build_engine(args, dtype, world, dist, local, rank, nsteps):
    """Engine + HBM-resident synthetic inputs for `dtype` in {"bf16", "f32", "fp8", "fp8c"}."""
    import torch
    from downgan_amd import synthetic
    from downgan_amd.engine import HyperParams, TrainEngine
    from downgan_amd.ops import HipOps
    B, S, F_, cin, nrb = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    is_f8 = dtype in ("fp8", "fp8c")
    ops = HipOps("bf16" if is_f8 else dtype, f"cuda:{local}", f8_critic=is_f8, f8_generator=dtype == "fp8")
    eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B * world), num_res_blocks=nrb, dist=dist, check_finite=args.check_finite)
    pg, pc = synthetic.generator_params(F_, cin, 2, nrb), synthetic.critic_params(F_, 8 * S, 2)      # seed 0 on every rank
    coarse, fine = synthetic.tiles(B, cin, S, rank=rank)                      # seed 1234 + rank
    if args.zero_data:
        pg, pc = {k: v * 0 for k, v in pg.items()}, {k: v * 0 for k, v in pc.items()}
        coarse, fine = coarse * 0, fine * 0
    eng.G.load_state_dict(pg)
    eng.C.load_state_dict(pc)
    xc = ops.zeros(B, S, S, eng.G.cin_p); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
    xf = ops.zeros(B, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
    del coarse, fine
    alphas = [torch.from_numpy(synthetic.alpha(B, s, rank=rank)).cuda() for s in range(nsteps)]
    return ops, eng, xc, xf, alphas


def timed_steps(eng, ops, xc, xf, alphas, first, steps, dist, rehearse, kernel_timing, per_layer):
    """Time exactly `steps` train steps (alphas[first:first+steps]) between barrier + device sync on both sides; MAX over ranks.
    The step counter is aligned so that the region OPENS with a generator step: K timed steps then contain ceil(K / critic_iterations)
    generator iterations for every --steps / --warmup combination -- never fewer than the long-run share."""
    import torch

    def complete_updates():
        # the Adam updates are deferred behind the next forward (engine.py); finish them so that exactly the work of the steps
        # issued so far lies before the synchronisation point
        eng.C.P.sync(); eng.G.P.sync()
    ci = eng.hp.critic_iterations
    eng.num_steps = -(-eng.num_steps // ci) * ci
    complete_updates()
    if dist:
        dist.track_overlap()          # count the exchanges of the timed region only (the one just completed belongs to the warm-up)
        dist.barrier()
    torch.cuda.synchronize()
    if kernel_timing:
        ops.prof = []
        ops.per_layer = per_layer
    t0 = time.perf_counter()
    for s in range(first, first + steps):
        eng.train_step(xc, xf, alphas[s])
    complete_updates()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        t = t if rehearse else t.cuda()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    prof, ops.prof = ops.prof, None
    return elapsed, prof


def summarise_kernels(prof, elapsed, steps, peak, args, per_layer, with_traffic=True):
    """Live HIP-event timings of the launches in the timed region -> (roofline of the dominant kernel, critic stack, kernels table)."""
    if not prof:
        return None, None, {}
    agg, kernels = {}, {}
    for tag, flops, nbytes, s_ev, e_ev in prof:
        a = agg.setdefault(tag, [0.0, 0.0, 0, 0.0])
        a[0] += flops; a[1] += s_ev.elapsed_time(e_ev) * 1e-3; a[2] += 1; a[3] += nbytes
    for tag, (fl, sec, n, nb_) in agg.items():
        kernels[tag] = {"launches": n, "seconds": round(sec, 4), "tflops": round(fl / sec / 1e12, 2) if sec > 0 else None}
        if per_layer:
            kernels[tag]["alg_tbps"] = round(nb_ / sec / 1e12, 2) if sec > 0 else None
    # dominant kernel: gg_halo4w_kernel (conv forward of both strides + data gradients; all its instantiations).  Only calls that were
    # served by that kernel alone (tag suffix k8) are counted, so achieved = algorithmic flops of those
    # launches / their summed launch durations, and launches/seconds give the average launch duration.
    halo = [t for t in agg if t.endswith(":k8")]
    fl = sum(agg[t][0] for t in halo)
    sec = sum(agg[t][1] for t in halo)
    nl = sum(agg[t][2] for t in halo)
    ach = fl / sec / 1e12 if sec > 0 else 0.0
    alg_bytes = sum(agg[t][3] for t in halo) / max(nl, 1)
    traffic, traffic_src = pmc_traffic("gg_halo4w_kernel", args) if with_traffic else (None, None)
    roofline = {"kernel": "gg_halo4w_kernel (implicit-GEMM conv3x3: forward, stride 1 and 2, and data gradients)", "bound": "mfma",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "alg_bytes_per_launch": round(alg_bytes),
                "launches": nl, "avg_launch_ms": round(sec / max(nl, 1) * 1e3, 4),
                "share_of_step": round(sec / elapsed, 3)}
    # critic stack (BASELINE target: >= 40 % MFMA utilisation; SURVEY 8(d): 10.4*Cf*B / t_critic_kernels): EVERY kernel of
    # the critic -- conv forward / data / weight gradients of its 8 layers incl. the HBM-bound 2-channel first layer, the
    # Linear kernels (lin_*), bias/activation and the penalty's elementwise passes (ew_*) -- algorithmic flops (real
    # channels) / summed launch durations.  `conv_only` is the same without the lin_* / ew_* launches.
    isc = lambda t: ":C" in t
    cf = sum(v[0] for t, v in agg.items() if isc(t))
    cs = sum(v[1] for t, v in agg.items() if isc(t))
    ccf = sum(v[0] for t, v in agg.items() if isc(t) and t.startswith("conv"))
    ccs = sum(v[1] for t, v in agg.items() if isc(t) and t.startswith("conv"))
    critic_stack = {"tflops": round(cf / cs / 1e12, 2) if cs > 0 else None, "mfma_frac": round(cf / cs / 1e12 / peak, 4) if cs > 0 else None,
                    "share_of_step": round(cs / elapsed, 3),
                    "conv_only": {"tflops": round(ccf / ccs / 1e12, 2) if ccs > 0 else None,
                                  "mfma_frac": round(ccf / ccs / 1e12 / peak, 4) if ccs > 0 else None},
                    "alg_tflop_per_step": round(cf / steps / 1e12, 3)}
    f8c = [t for t in agg if t.endswith(":k32")]
    f8w = [t for t in agg if t.startswith("conv_wgrad_f8")]
    if f8c:     # fp8 mode: the MXFP8 conv launches (quantisation of their operands included in the bracketed time) + the fp8 weight gradients
        def rate(tags):
            fl_, sec_, n_ = sum(agg[t][0] for t in tags), sum(agg[t][1] for t in tags), sum(agg[t][2] for t in tags)
            return {"achieved": round(fl_ / sec_ / 1e12, 2) if sec_ > 0 else None, "frac": round(fl_ / sec_ / 1e12 / MFMA_PEAK_TFLOPS["fp8"], 4) if sec_ > 0 else None,
                    "launches": n_, "share_of_step": round(sec_ / elapsed, 3)}
        roofline["fp8_kernel"] = dict({"kernel": "every fp8 MFMA launch of the step: gg_halo4w_f8_kernel (MXFP8 conv forward / data gradient of the critic's wide "
                                                 "layers, operand quantisation included; with --dtype fp8 also the generator's dense-block trunk forward + data gradients and "
                                                 "its up-sampling tail's forward) and wg3w_f8_kernel (uniform-scale E4M3 weight gradients of the critic's layers 1-7 "
                                                 "and, from the second generator iteration on, of the generator's dense blocks)", "bound": "mfma",
                                       "peak": MFMA_PEAK_TFLOPS["fp8"], "unit": "TFLOP/s"}, **rate(f8c + f8w),
                                      conv=rate(f8c), weight_gradient=rate(f8w) if f8w else None)
    return roofline, critic_stack, kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32", "fp8", "fp8c"],
                    help="default: the workload's own (bf16; fp8 for cfg5).  fp8 = BASELINE configs[4]: forward / data-gradient convs of the critic's 128..1024-channel layers, of the generator's "
                         "dense-block trunk and the forward of its up-sampling tail on the MXFP8 MFMA (fp32 accumulate), their weight gradients on the fp8 MFMA "
                         "with uniform-scale operands, everything else as in bf16; "
                         "fp8c = the critic's layers only")
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two short extra measurements of the default line (5 steps without per-launch events; 5 steps of the fp8 mode)")
    ap.add_argument("--per-layer", action="store_true", help="diagnostic: split the `kernels` table by layer geometry (with bytes)")
    ap.add_argument("--graphs", action="store_true", help="replay the iterations as captured HIP graphs (small, launch-bound tiles)")
    ap.add_argument("--check-finite", action="store_true",
                    help="debug: after every iteration one fused isfinite reduction over the scalars, the gradient buffers and the generated "
                         "batch; raises naming the first offending buffer (the stand-in for the reference's set_detect_anomaly, wasserstein.py:13)")
    ap.add_argument("--zero-data", action="store_true",
                    help="diagnostic: all-zero tiles AND weights (same launches, same cycles; shows how much of the distance to the MFMA "
                         "peak is the clock the chip holds on random data -- MI355X_MICROARCH.md, DVFS give-back); never a benchmark result")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and the gradient exchange goes over gloo")
    args = ap.parse_args()
    if args.dtype is None:
        args.dtype = WORKLOAD_INFO[args.workload][2]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this parent process never imports torch and never touches the GPU; it starts N
        # FRESH rank processes (one per GPU) through torch.distributed.run and relays their output -- rank 0 prints the JSON line.
        sys.exit(self_launch(args.gpus))

    import torch
    from downgan_amd.dist import Dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dist = Dist("gloo" if args.rehearse_on_one_gpu else "nccl") if world > 1 else None
    rank = dist.rank if dist else 0
    local = 0 if args.rehearse_on_one_gpu else (dist.local_rank if dist else 0)
    torch.cuda.set_device(local)
    B, S, F_, cin, nrb = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    is_f8 = args.dtype in ("fp8", "fp8c")
    extras = world == 1 and not args.no_extras and not args.graphs and not args.per_layer
    nsteps = args.warmup + args.steps + (5 if extras else 0)
    ops, eng, xc, xf, alphas = build_engine(args, args.dtype, world, dist, local, rank, nsteps)

    # who is where: every rank reports the device it computes on (audit trail of the first real multi-GPU run)
    me = {"rank": rank, "local_rank": dist.local_rank if dist else 0, "device": f"cuda:{local}", "device_name": torch.cuda.get_device_name(local),
          "visible_devices": torch.cuda.device_count(), "pid": os.getpid()}
    ranks_seen = [me]
    if dist:
        ranks_seen = [None] * world
        torch.distributed.all_gather_object(ranks_seen, me)
        # a multi-GPU number needs `world` DISTINCT devices: ranks piled onto one GPU (a launcher that lost LOCAL_RANK, a
        # visibility mask per rank) would still print a line.  Only the explicit one-GPU rehearsal may share a device.
        devs = {r["device"] for r in ranks_seen}
        if not args.rehearse_on_one_gpu and (len(devs) != world or any(r["visible_devices"] < world for r in ranks_seen)):
            if rank == 0:
                print(json.dumps({"error": f"--gpus {world}: the ranks compute on {len(devs)} distinct device(s) {sorted(devs)}; "
                                           "not a multi-GPU run (use --rehearse-on-one-gpu for a one-GPU rehearsal)", "ranks_seen": ranks_seen}), flush=True)
            dist.barrier()
            sys.exit(3)
        dist.track_overlap()

    if args.graphs:
        eng.enable_graphs(xc, xf)
        args.no_kernel_timing = True          # per-launch events cannot be recorded inside a replayed graph
    for s in range(args.warmup):
        eng.train_step(xc, xf, alphas[s])
    gen_steps_timed = -(-args.steps // eng.hp.critic_iterations)
    elapsed, prof = timed_steps(eng, ops, xc, xf, alphas, args.warmup, args.steps, dist, args.rehearse_on_one_gpu,
                                not args.no_kernel_timing, args.per_layer)
    overlap = dict(dist.overlap) if dist else None
    scal = eng.read_scalars(True)
    peak = MFMA_PEAK_TFLOPS["bf16" if is_f8 else args.dtype]      # gg_halo4w_kernel is a bf16 / fp32 kernel in every mode
    roofline, critic_stack, kernels = summarise_kernels(prof, elapsed, args.steps, peak, args, args.per_layer)
    hbm_peak = torch.cuda.max_memory_allocated()
    stacked = bool(eng.stacked)

    # ---- extras of the default line (single GPU): the same engine without the per-launch events, then the fp8 mode
    bare = fp8 = None
    if extras:
        if not args.no_kernel_timing:
            e2, _ = timed_steps(eng, ops, xc, xf, alphas, args.warmup + args.steps, 5, None, False, False, False)
            bare = {"steps": 5, "ms_per_step": round(e2 / 5 * 1e3, 2), "value": round(5 * B / e2, 4), "unit": "samples/s",
                    "what": "5 more steps of the same engine WITHOUT the per-launch HIP events that `kernels` / `roofline` come from "
                            "(`value` is the event-instrumented region, the conservative one)"}
        if args.dtype == "bf16" and args.workload in ("cfg2", "mid") and not args.check_finite:
            # BASELINE configs[4] (fp8 conv path) made visible to the default command: free the bf16 engine, build the fp8 one,
            # 1 warm-up + 5 timed steps (one generator step among them, as in the long run).  Not `value`.
            del eng, ops, xc, xf, alphas
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            ops8, eng8, xc8, xf8, al8 = build_engine(args, "fp8", 1, None, local, 0, 6)
            eng8.train_step(xc8, xf8, al8[0])
            e8, prof8 = timed_steps(eng8, ops8, xc8, xf8, al8, 1, 5, None, False, True, False)
            sc8 = eng8.read_scalars(True)
            r8, cs8, _ = summarise_kernels(prof8, e8, 5, MFMA_PEAK_TFLOPS["bf16"], args, False, with_traffic=False)
            fp8 = {"what": "--dtype fp8 (BASELINE configs[4]: MXFP8 MFMA for the forward + data gradients of the critic's wide convs and of the generator's "
                           "dense-block trunk and for the forward of its up-sampling tail, fp8 MFMA with uniform-scale operands for their weight gradients; "
                           "first layers / conv3.2 / tail backward / Linear / Adam as in bf16), same workload, 1 warm-up + 5 timed steps",
                   "value": round(5 * B / e8, 4), "unit": "samples/s", "steps": 5, "ms_per_step": round(e8 / 5 * 1e3, 2),
                   "vs_bf16": round((5 * B / e8) / (args.steps * B / elapsed), 4),
                   "roofline": (r8 or {}).get("fp8_kernel"), "critic_stack_frac_of_bf16_peak": (cs8 or {}).get("mfma_frac"),
                   "losses": {k: sc8[k] for k in ("critic_loss", "gp_ret", "g_loss") if k in sc8},
                   "hbm_peak_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
            del eng8, ops8, xc8, xf8, al8

    if rank == 0:
        gf, cf = conv_flops_per_sample(S, F_, cin, 2, nrb)                          # real channels: algorithmic
        gfp, cfp = conv_flops_per_sample(S, F_, 16 * ((cin + 15) // 16), 16, nrb)   # zero-padded channels: what the MFMAs execute
        value = args.steps * B * world / elapsed
        w_survey = 1.6 * gf + 10.4 * cf        # SURVEY 8(d) necessary work: critic iteration Gf + 10 Cf, generator iteration 3 Gf + 2 Cf
        w_step = 1.4 * gf + 10.4 * cf          # executed: on generator steps ONE G(coarse) serves both iterations (engine.train_step)
        w_padded = 1.4 * gfp + 10.4 * cfp
        out = {
            "metric": "train-step samples/sec (G+D+GP)", "value": round(value, 4), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "ZEROS (clock diagnostic, not a result)" if args.zero_data else "synthetic",
            "rehearsal": bool(args.rehearse_on_one_gpu),
            "config": {"workload": f"{args.workload} ({WORKLOAD_INFO[args.workload][0]}"
                                   + ("" if (world, args.dtype) == WORKLOAD_INFO[args.workload][1:] or args.workload == "mid" else
                                      f", quoted on {WORKLOAD_INFO[args.workload][1]} GPU(s) in {WORKLOAD_INFO[args.workload][2]}: THIS run is {world} GPU(s) in {args.dtype}")
                                   + f"): batch {B}/GPU (global {B * world}), {cin}ch {S}x{S}->{8 * S}x{8 * S}, "
                                   f"filters {F_}, {nrb} RRDBs, WGAN-GP critic step every step + generator step every 5th",
                       "parallelism": f"dp{world}", "per_gpu_batch": B, "global_batch": B * world,
                       "exchange": (None if world == 1 else "gloo over host memory, all ranks on cuda:0 (REHEARSAL: not a multi-GPU measurement)"
                                    if args.rehearse_on_one_gpu else "RCCL (torch.distributed nccl backend) all-reduce of the flat fp32 gradient buffers"),
                       "alg_tflop_per_sample_step": round(w_step / 1e12, 4),
                       "alg_tflop_per_sample_step_survey": round(w_survey / 1e12, 4),
                       "executed_padded_tflop_per_sample_step": round(w_padded / 1e12, 4),
                       "work": "1.4*Gf + 10.4*Cf per sample-step, real channels (SURVEY 8(d) counts 1.6*Gf: the generator iteration's "
                               "G(coarse) is the critic iteration's, computed once)",
                       "generator_steps_in_timed_region": gen_steps_timed,
                       "critic_passes_stacked": stacked},
            "ranks_seen": ranks_seen,
            "exchange_overlap": (None if overlap is None else dict(overlap, what=(
                "deferred gradient exchanges whose all-reduce had ALREADY completed (non-blocking query of the last bucket) when the "
                "consumer reached its wait: already_complete == finishes means the exchange was fully hidden behind the next forward pass"))),
            "hbm_peak_gib": round(hbm_peak / 2 ** 30, 1),
            "step_mfma_frac": round(w_step * value / world / 1e12 / peak, 4),
            "losses": {k: scal[k] for k in ("critic_loss", "gp_ret", "g_loss") if k in scal},
            "roofline": roofline, "critic_conv_stack": critic_stack, "kernels": kernels,
        }
        if bare is not None:
            out["without_kernel_events"] = bare
        if fp8 is not None:
            out["fp8"] = fp8
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, (B, S, F_, cin, nrb))
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()


if __name__ == "__main__":
    main()
