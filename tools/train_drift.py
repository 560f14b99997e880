"""Does training behave the same in bf16 as in the fp32-parity mode BEYOND the first steps?

    python tools/train_drift.py [--steps 40] [--out profiles/train_drift_learnable.json]

The pure-noise tiles of the benchmark make the WGAN-GP step a chaotic system (profiles/fp8_drift_cfg2.json: bf16 itself swings by
1e4 within six steps), so trajectories of two roundings separate whatever the kernels do.  Here the target is LEARNABLE: the fine
field is the bilinear 8x up-sampling of the two first coarse channels plus a fixed smooth pattern, a new batch every step, so the
content loss (losses.py:51-53, weight 5 in g_loss, wasserstein.py:78) falls as the generator learns.  The same run -- same initial
weights, data and alphas -- in fp32-parity mode, bf16 (the benchmarked precision) and fp8 mode; recorded per step: content_loss,
g_loss (generator steps), critic_loss, gp_ret, w_estimate.  What must agree is the LEARNING CURVE: the content loss of the
generator steps, its fall over the run and its final level.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from downgan_amd import synthetic  # noqa: E402
from downgan_amd.engine import HyperParams, TrainEngine  # noqa: E402
from downgan_amd.ops import HipOps  # noqa: E402


def learnable_batch(B, cin, S, step):
    """coarse ~ N(0,1) smoothed; fine = bilinear x8 of coarse[:, :2] + a fixed pattern (numpy PCG64, torch only for interpolate)."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([99, step])))
    c = rng.standard_normal((B, cin, S, S)).astype(np.float32)
    c = (c + np.roll(c, 1, 2) + np.roll(c, 1, 3) + np.roll(c, (1, 1), (2, 3))) * 0.5          # neighbouring cells correlate
    fine = torch.nn.functional.interpolate(torch.from_numpy(c[:, :2]), scale_factor=8, mode="bilinear", align_corners=False)
    yy, xx = np.meshgrid(np.arange(8 * S, dtype=np.float32), np.arange(8 * S, dtype=np.float32), indexing="ij")
    pattern = 0.25 * np.stack([np.sin(yy / 5.0) * np.cos(xx / 7.0), np.cos(yy / 6.0 + xx / 9.0)])
    return c, (fine + torch.from_numpy(pattern)[None]).numpy().astype(np.float32)


def run(mode, steps, B, S, F_, cin, nrb, deterministic=None, round_init=False):
    """``round_init``: the initial parameters rounded to bf16 ONCE (a perturbation of relative size 2^-9, the size of one bf16
    rounding) -- run in fp32 mode it shows how far a single bf16-sized disturbance moves the trajectory: the yardstick for what
    the bf16 mode, which rounds weights and activations at every step, may differ by."""
    ops = HipOps("f32" if mode == "f32" else "bf16", "cuda:0", f8_critic=mode == "fp8", f8_generator=mode == "fp8", deterministic=deterministic)
    eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
    rnd = (lambda d: {k: torch.from_numpy(v).to(torch.bfloat16).to(torch.float32) for k, v in d.items()}) if round_init else (lambda d: d)
    eng.G.load_state_dict(rnd(synthetic.generator_params(F_, cin, 2, nrb)))
    eng.C.load_state_dict(rnd(synthetic.critic_params(F_, 8 * S, 2)))
    xc, xf = ops.zeros(B, S, S, eng.G.cin_p), ops.zeros(B, 8 * S, 8 * S, eng.G.np_p)
    out = []
    for s in range(steps):
        coarse, fine = learnable_batch(B, cin, S, s)
        ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
        ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
        ran_g = eng.train_step(xc, xf, torch.from_numpy(synthetic.alpha(B, s)).cuda())
        out.append(eng.read_scalars(ran_g))
    del eng
    torch.cuda.empty_cache()
    return out


def summarise(traj):
    cl = [(i, t["content_loss"]) for i, t in enumerate(traj) if "content_loss" in t]
    n = max(1, len(cl) // 4)
    return {"content_loss_generator_steps": cl, "content_first_quarter": float(np.mean([v for _, v in cl[:n]])),
            "content_last_quarter": float(np.mean([v for _, v in cl[-n:]])),
            "critic_loss": [t["critic_loss"] for t in traj], "gp_ret": [t["gp_ret"] for t in traj],
            "w_estimate": [t["w_estimate"] for t in traj]}


def compare(steps=40, B=4, S=32, F_=128, cin=2, nrb=2, modes=("f32", "bf16", "fp8")):
    res = {m: summarise(run(m, steps, B, S, F_, cin, nrb)) for m in modes}
    ref = res["f32"]
    cmpd = {}
    for m in modes[1:]:
        a = np.array([v for _, v in res[m]["content_loss_generator_steps"]])
        b = np.array([v for _, v in ref["content_loss_generator_steps"]])
        cmpd[m] = {"max_rel_content_loss_vs_f32": float(np.max(np.abs(a - b) / b)),
                   "last_quarter_ratio_vs_f32": res[m]["content_last_quarter"] / ref["content_last_quarter"],
                   "fall_f32": ref["content_last_quarter"] / ref["content_first_quarter"],
                   "fall": res[m]["content_last_quarter"] / res[m]["content_first_quarter"]}
    return {"what": "learnable synthetic task (fine = bilinear x8 of the coarse field + a fixed pattern, a new batch every step): the "
                    "same run in fp32-parity mode, bf16 and fp8 mode; content loss at the generator steps (every 5th)",
            "config": {"steps": steps, "batch": B, "coarse": S, "filters": F_, "channels": cin, "rrdbs": nrb}, "compare": cmpd, "runs": res}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--coarse", type=int, default=32)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--rrdbs", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train_drift_learnable.json"))
    a = ap.parse_args()
    res = compare(a.steps, a.batch, a.coarse, a.filters, 2, a.rrdbs)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    for m, r in res["runs"].items():
        print(m, "content loss at generator steps:", [round(v, 4) for _, v in r["content_loss_generator_steps"]])
    print(json.dumps(res["compare"], indent=1))


if __name__ == "__main__":
    main()
