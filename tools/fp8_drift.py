"""Loss drift of the MXFP8 conv path against the bf16 path (BASELINE.json configs[4]: "loss-drift vs bf16 reported").

    python tools/fp8_drift.py [--steps 25] [--batch 4] [--out profiles/fp8_drift_cfg2.json]

Both runs start from the same synthetic weights, see the same tiles and alphas (downgan_amd/synthetic.py) and differ only in
HipOps(f8_critic=...): forward / data-gradient convs of the critic's 128..1024-channel layers in MXFP8 (E4M3 elements, E8M0
scale per 32 channels, fp32 accumulate) instead of bf16.  Per step: critic_loss, gp_ret, the critic means, g_loss, and the
relative difference of each to the bf16 run.  BASELINE configs[1] shapes (2ch 128x128 -> 1024x1024, F = 128, 16 RRDBs).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from downgan_amd import synthetic  # noqa: E402
from downgan_amd.engine import HyperParams, TrainEngine  # noqa: E402
from downgan_amd.ops import HipOps  # noqa: E402


def run(f8, steps, B, S, F_, nrb, fresh_batches):
    """f8: False (bf16), "critic" (the critic's wide convs) or "all" (+ the generator trunk's forward)."""
    ops = HipOps("bf16", "cuda:0", f8_critic=bool(f8), f8_generator=f8 == "all")
    eng = TrainEngine(ops, S, F_, 2, B, HyperParams(batch_size=B), num_res_blocks=nrb)
    eng.G.load_state_dict(synthetic.generator_params(F_, 2, 2, nrb))
    eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
    xc, xf = ops.zeros(B, S, S, 16), ops.zeros(B, 8 * S, 8 * S, 16)
    out = []
    for s in range(steps):
        if s == 0 or fresh_batches:
            coarse, fine = synthetic.tiles(B, 2, S, seed=1234 + (s if fresh_batches else 0))
            ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
            ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
        ran_g = eng.train_step(xc, xf, torch.from_numpy(synthetic.alpha(B, s)).cuda())
        out.append(eng.read_scalars(ran_g))
    del eng
    torch.cuda.empty_cache()
    return out


def first_step_gradients(B, S, F_, nrb):
    """critic parameter gradients of a critic iteration on the initial weights (no update) in fp32-parity, bf16 and fp8 mode:
    relative l2 error and cosine against the fp32 gradient, per parameter -- the sharp, dynamics-free drift.  fp8 mode twice: with
    the bf16 weight-gradient kernels (ops.f8_wgrad = False: only the adjoints carry fp8 error) and with the fp8 weight gradients
    (dg_conv3x3_wgrad_f8) -- taken from the SECOND of two identical iterations, the first one being the pass that initialises the
    per-block exponents (no update in between: the same gradient in exact arithmetic)."""
    grads = {}
    for mode in ("f32", "bf16", "fp8_bf16_wgrad", "fp8"):
        f8 = mode.startswith("fp8")
        ops = HipOps("f32" if mode == "f32" else "bf16", "cuda:0", f8_critic=f8, f8_generator=f8)
        if mode == "fp8_bf16_wgrad":
            ops.f8_wgrad = False
        eng = TrainEngine(ops, S, F_, 2, B, HyperParams(batch_size=B), num_res_blocks=nrb)
        eng.G.load_state_dict(synthetic.generator_params(F_, 2, 2, nrb))
        eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
        coarse, fine = synthetic.tiles(B, 2, S)
        xc, xf = ops.zeros(B, S, S, 16), ops.zeros(B, 8 * S, 8 * S, 16)
        ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
        ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
        for _ in range(2 if mode == "fp8" else 1):
            eng.critic_iteration(xc, xf, torch.from_numpy(synthetic.alpha(B, 0)).cuda(), apply_update=False)
        if mode == "fp8":
            grads["_fp8_layers"] = [bool(w) for w in eng.C.wg8]
        grads[mode] = {k: v.double() for k, v in eng.C.grad_dict().items()}
        del eng
        torch.cuda.empty_cache()
    out = {"_layers_on_the_fp8_weight_gradient_kernel": grads.pop("_fp8_layers")}
    for k, g in grads["f32"].items():
        if float(g.norm()) == 0:
            continue
        out[k] = {m: {"rel_l2": float((grads[m][k] - g).norm() / g.norm()),
                      "cosine": float((grads[m][k] * g).sum() / (grads[m][k].norm() * g.norm() + 1e-300))} for m in ("bf16", "fp8_bf16_wgrad", "fp8")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--coarse", type=int, default=128)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--rrdbs", type=int, default=16)
    ap.add_argument("--fresh-batches", action="store_true", help="a new synthetic batch every step instead of one fixed batch")
    ap.add_argument("--gradients-only", action="store_true", help="only the first-step gradient comparison (2 minutes instead of 10)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fp8_drift_cfg2.json"))
    a = ap.parse_args()
    if a.gradients_only:
        res = {"first_step_critic_gradients_vs_fp32": first_step_gradients(min(a.batch, 2), a.coarse, a.filters, a.rrdbs)}
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
        for k, v in res["first_step_critic_gradients_vs_fp32"].items():
            if k.startswith("_"):
                print(k, v)
                continue
            print(f"{k:24s} bf16 rel {v['bf16']['rel_l2']:.3f} cos {v['bf16']['cosine']:.4f} | fp8, bf16 wgrad rel {v['fp8_bf16_wgrad']['rel_l2']:.3f} "
                  f"cos {v['fp8_bf16_wgrad']['cosine']:.4f} | fp8 rel {v['fp8']['rel_l2']:.3f} cos {v['fp8']['cosine']:.4f}")
        return
    ref = run(False, a.steps, a.batch, a.coarse, a.filters, a.rrdbs, a.fresh_batches)
    f8 = run("all", a.steps, a.batch, a.coarse, a.filters, a.rrdbs, a.fresh_batches)
    f8c = run("critic", a.steps, a.batch, a.coarse, a.filters, a.rrdbs, a.fresh_batches)
    keys = ("critic_loss", "gp_ret", "c_real_mean", "c_fake_mean", "w_estimate", "g_loss", "content_loss")
    steps = []
    for s, (r, q, qc) in enumerate(zip(ref, f8, f8c)):
        rec = {"step": s}
        for k in keys:
            if k in r:
                rec[k] = {"bf16": r[k], "fp8": q[k], "fp8_critic_only": qc[k], "rel": abs(q[k] - r[k]) / max(abs(r[k]), 1e-3),
                          "rel_critic_only": abs(qc[k] - r[k]) / max(abs(r[k]), 1e-3)}
        steps.append(rec)
    res = {"what": "MXFP8 conv path (fp8 = critic wide convs + generator trunk forward, as bench.py --dtype fp8; fp8_critic_only = --dtype fp8c) "
                   "vs bf16, same init / data / alpha; rel = |fp8 - bf16| / max(|bf16|, 1e-3)",
           "config": {"batch": a.batch, "coarse": a.coarse, "filters": a.filters, "rrdbs": a.rrdbs, "steps": a.steps,
                      "fresh_batches": a.fresh_batches},
           "max_rel": {k: max(st[k]["rel"] for st in steps if k in st) for k in keys},
           "rel_at_step0": {k: steps[0][k]["rel"] for k in keys if k in steps[0]},
           "rel_at_step0_critic_only": {k: steps[0][k]["rel_critic_only"] for k in keys if k in steps[0]},
           "first_step_critic_gradients_vs_fp32": first_step_gradients(min(a.batch, 2), a.coarse, a.filters, a.rrdbs),
           "note": "the synthetic fixed-batch problem is an oscillating system in EVERY precision (bf16 itself swings between -4e3 and "
                   "+1e4 within 6 steps): the runs agree while the trajectory is smooth and separate, as any two roundings do, once it "
                   "is not; the first-step gradients are the dynamics-free comparison",
           "steps": steps}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    for st in steps:
        print(st["step"], {k: (round(st[k]["bf16"], 5), round(st[k]["fp8"], 5)) for k in ("critic_loss", "gp_ret", "w_estimate") if k in st})
    print("max rel:", {k: f"{v:.3g}" for k, v in res["max_rel"].items()})
    for k, v in res["first_step_critic_gradients_vs_fp32"].items():
        if k.startswith("_"):
            print(k, v)
            continue
        print(f"{k:24s} bf16 rel {v['bf16']['rel_l2']:.3f} cos {v['bf16']['cosine']:.4f} | fp8, bf16 wgrad rel {v['fp8_bf16_wgrad']['rel_l2']:.3f} "
              f"cos {v['fp8_bf16_wgrad']['cosine']:.4f} | fp8 rel {v['fp8']['rel_l2']:.3f} cos {v['fp8']['cosine']:.4f}")


if __name__ == "__main__":
    main()
