"""GPU parity of the whole native train step (C-ABI HIP path) against the reference goldens and the
CPU oracle.  fp32-parity mode: losses / GP within 1e-4 relative (BASELINE.json north star); bf16 mode:
drift is reported and loosely bounded."""
import json
import os

import pytest
import torch

from downgan_amd import synthetic
from downgan_amd.engine import HyperParams, TrainEngine
from downgan_amd.ops import HipOps
from oracle import ref_step

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make(B, S, F_, cin, nrb, dtype):
    ops = HipOps(dtype)
    eng = TrainEngine(ops, S, F_, cin, B, HyperParams(batch_size=B), num_res_blocks=nrb)
    pg = synthetic.generator_params(F_, cin, 2, nrb)
    pc = synthetic.critic_params(F_, 8 * S, 2)
    eng.G.load_state_dict(pg)
    eng.C.load_state_dict(pc)
    coarse, fine = synthetic.tiles(B, cin, S, mask_channel=(2 if cin > 2 else None))
    tc, tf = torch.from_numpy(coarse), torch.from_numpy(fine)
    xc = ops.zeros(B, S, S, eng.G.cin_p); ops.nchw_to_nhwc(tc.cuda(), xc)
    xf = ops.zeros(B, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(tf.cuda(), xf)
    return eng, pg, pc, tc, tf, xc, xf


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-30)


@pytest.mark.parametrize("name,nsteps", [("cfg1", 6), ("cin6_small", 1), ("f32_s32", 2)])
def test_fp32_steps_match_reference_golden(name, nsteps):
    """identical synthetic inputs/weights/alpha; critic_loss, g_loss, gp (and the means) within 1e-4 relative."""
    with open(os.path.join(GOLD, name + ".json")) as f:
        gold = json.load(f)
    c = gold["config"]
    eng, *_, xc, xf = make(c["B"], c["S"], c["F"], c["cin"], c["num_res_blocks"], "f32")
    for step in range(nsteps):
        alpha = torch.from_numpy(synthetic.alpha(c["B"], step)).cuda()
        ran_g = eng.train_step(xc, xf, alpha)
        got = eng.read_scalars(ran_g)
        rec = gold["steps"][step]
        keys = ["c_real_mean", "c_fake_mean", "gp_ret", "gradient_penalty", "critic_loss"] + (["g_loss", "content_loss", "g_c_fake_mean"] if ran_g else [])
        for k in keys:
            assert rel(got[k], rec[k]) < 1e-4, (name, step, k, got[k], rec[k])
        assert abs(got["w_estimate"] - rec["w_estimate"]) < 1e-4 * max(abs(rec["c_real_mean"]), abs(rec["c_fake_mean"]))
    # Post-Adam parameters.  Adam normalises each gradient entry, so entries whose gradient is fp32 noise in the
    # reference itself (real/fake terms cancel, see tests/test_engine_cpu.py) may move by up to +-lr per step in
    # either run: norms are checked to 5e-5 and sampled entries to a fraction of the total possible movement.
    lr = eng.hp.lr
    sd = eng.C.state_dict()
    for k, s in gold["steps"][nsteps - 1]["C_params_after"].items():
        assert rel(float(sd[k].double().norm()), s["l2"]) < 5e-5, k
        flat = sd[k].double().flatten()
        for i, v in zip(s["idx"], s["val"]):
            assert abs(float(flat[i]) - v) < 0.25 * lr * nsteps + 1e-7, (k, i, float(flat[i]), v)
    if "G_params_after" in gold["steps"][nsteps - 1]:
        sd = eng.G.state_dict()
        for k, s in gold["steps"][nsteps - 1]["G_params_after"].items():
            assert rel(float(sd[k].double().norm()), s["l2"]) < 5e-5, k


@pytest.mark.parametrize("name", ["cfg1", "f32_s32"])
def test_metrics_pass_matches_reference_golden(name):
    """MAE / MSE / Wasserstein estimate of the per-step metrics pass (mlflow_epoch.py:53-63) in fp32-parity mode."""
    with open(os.path.join(GOLD, name + ".json")) as f:
        gold = json.load(f)
    c = gold["config"]
    eng, *_, xc, xf = make(c["B"], c["S"], c["F"], c["cin"], c["num_res_blocks"], "f32")
    m = eng.metrics_pass(xc, xf)
    for k, v in gold["forward"]["metrics"].items():
        assert abs(m[k] - v) <= 1e-4 * max(abs(v), abs(gold["steps"][0]["c_real_mean"])), (k, m[k], v)
    # MS-SSIM (third-party in the reference, parity unpinned): against the oracle's restatement on the oracle's own G(x)
    from oracle import msssim as om
    from oracle import ref_step
    pg, pc, tc, tf = _
    PG = {k: torch.from_numpy(v) for k, v in pg.items()}
    with torch.no_grad():
        ref = om.ssim_loss(tf, ref_step.generator_forward(PG, tc, c["num_res_blocks"]))
    assert m["MSSSIM"] is not None and abs(m["MSSSIM"] - ref) < 1e-4, (m["MSSSIM"], ref)


def test_fp32_gradients_match_float64_oracle():
    B, S, F_, cin, nrb = 2, 16, 16, 6, 2
    eng, pg, pc, tc, tf, xc, xf = make(B, S, F_, cin, nrb, "f32")
    o64 = ref_step.OracleTrainer({k: torch.from_numpy(v).double() for k, v in pg.items()},
                                 {k: torch.from_numpy(v).double() for k, v in pc.items()}, ref_step.HP(batch_size=B), num_res_blocks=nrb)
    o32 = ref_step.OracleTrainer({k: torch.from_numpy(v) for k, v in pg.items()},
                                 {k: torch.from_numpy(v) for k, v in pc.items()}, ref_step.HP(batch_size=B), num_res_blocks=nrb)
    alpha = torch.from_numpy(synthetic.alpha(B, 0))
    _, cg = o64.critic_iteration(tc.double(), tf.double(), alpha.double(), apply_update=False)
    _, cg32 = o32.critic_iteration(tc, tf, alpha, apply_update=False)
    eng.critic_iteration(xc, xf, alpha.cuda(), apply_update=False)
    gd = eng.C.grad_dict()
    for k, g in cg.items():
        err = (gd[k].double() - g).norm() / (g.norm() + 1e-20)
        ref_err = (cg32[k].double() - g).norm() / (g.norm() + 1e-20)
        assert err < 2e-5 + 5 * ref_err, (k, float(err), float(ref_err))
    _, gg = o64.generator_iteration(tc.double(), tf.double(), apply_update=False)
    _, gg32 = o32.generator_iteration(tc, tf, apply_update=False)
    eng.generator_iteration(xc, xf, apply_update=False)
    gd = eng.G.grad_dict()
    for k, g in gg.items():
        err = (gd[k].double() - g).norm() / (g.norm() + 1e-20)
        ref_err = (gg32[k].double() - g).norm() / (g.norm() + 1e-20)
        assert err < 2e-5 + 5 * ref_err, (k, float(err), float(ref_err))


def test_bf16_step_drift_vs_golden():
    """bf16 storage + bf16 MFMA: reported drift, bounded loosely (not the parity gate)."""
    with open(os.path.join(GOLD, "cfg1.json")) as f:
        gold = json.load(f)
    eng, *_, xc, xf = make(4, 16, 16, 2, 16, "bf16")
    drift = {}
    for step in range(2):
        alpha = torch.from_numpy(synthetic.alpha(4, step)).cuda()
        ran_g = eng.train_step(xc, xf, alpha)
        got = eng.read_scalars(ran_g)
        rec = gold["steps"][step]
        for k in ["critic_loss", "gp_ret"] + (["g_loss", "content_loss"] if ran_g else []):
            drift[(step, k)] = rel(got[k], rec[k])
        assert abs(got["c_real_mean"] - rec["c_real_mean"]) < 5e-3
    print("bf16 drift vs fp32 reference:", {f"{s}:{k}": f"{v:.2e}" for (s, k), v in drift.items()})
    assert max(drift.values()) < 2e-2


def test_forward_drop_in_modules():
    """Generator / Critic mirrors accept the reference's NCHW fp32 tensors and state_dicts."""
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    B, S, F_, cin, nrb = 2, 16, 16, 6, 2
    pg = {k: torch.from_numpy(v) for k, v in synthetic.generator_params(F_, cin, 2, nrb).items()}
    pc = {k: torch.from_numpy(v) for k, v in synthetic.critic_params(F_, 8 * S, 2).items()}
    coarse, fine = synthetic.tiles(B, cin, S, mask_channel=2)
    G = Generator(F_, 8 * S, cin, 2, num_res_blocks=nrb, dtype="f32"); G.load_state_dict(pg)
    C = Critic(F_, 8 * S, 2, dtype="f32"); C.load_state_dict(pc)
    with torch.no_grad():
        ref_g = ref_step.generator_forward(pg, torch.from_numpy(coarse), nrb)
        ref_c = ref_step.critic_forward(pc, torch.from_numpy(fine))
    out_g = G(torch.from_numpy(coarse).cuda())
    out_c = C(torch.from_numpy(fine).cuda())
    assert out_g.shape == ref_g.shape and out_c.shape == ref_c.shape
    assert torch.allclose(out_g.cpu(), ref_g, atol=2e-5) and torch.allclose(out_c.cpu(), ref_c, atol=1e-6)
    sd = G.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in pg.items())
    # chunked generation (gen_fake_ds.py:147-162) incl. a ragged tail == one-shot forward
    series = torch.from_numpy(synthetic.tiles(5, cin, S, seed=77)[0])
    whole = G(series.cuda()).cpu()
    assert torch.allclose(G.generate(series, chunk_size=2), whole, atol=1e-6)


def test_first_adam_move_has_the_float64_oracles_sign():
    """Post-Adam parameters ENTRY BY ENTRY (stage.py:63-64): after one train step the share of entries of every critic and generator
    parameter whose move has the float64 oracle's sign, weighted by |g|, is >= 0.999 (Adam's first step is -lr * sign(g); entries
    whose gradient is fp32 rounding noise carry no weight).  Norm checks alone would pass a wrong update of a small slice."""
    B, S, F_, cin, nrb = 2, 16, 16, 2, 2
    eng, pg, pc, tc, tf, xc, xf = make(B, S, F_, cin, nrb, "f32")
    d = torch.float64
    o64 = ref_step.OracleTrainer({k: torch.from_numpy(v).to(d) for k, v in pg.items()}, {k: torch.from_numpy(v).to(d) for k, v in pc.items()},
                                 ref_step.HP(batch_size=B), num_res_blocks=nrb)
    alpha = torch.from_numpy(synthetic.alpha(B, 0))
    _, cg = o64.critic_iteration(tc.to(d), tf.to(d), alpha.to(d))
    _, gg = o64.generator_iteration(tc.to(d), tf.to(d))
    assert eng.train_step(xc, xf, alpha.cuda())
    for before, after, ref_after, grads in ((pc, eng.C.state_dict(), o64.PC, cg), (pg, eng.G.state_dict(), o64.PG, gg)):
        for k, g in grads.items():
            p0 = torch.from_numpy(before[k]).double()
            da, db = torch.sign(after[k].double() - p0), torch.sign(ref_after[k].detach() - p0)
            w = g.abs()
            if float(w.sum()) == 0.0:
                assert float(da.abs().sum()) == 0.0, k          # no gradient, no move
                continue
            share = float(((da == db).double() * w).sum() / w.sum())
            assert share >= 0.999, (k, share)
            moved = float((da != 0).double().mean())
            assert moved > 0.99 or float((db != 0).double().mean()) <= moved + 1e-3, (k, moved)


# (full-tile parity at BASELINE configs[1] shapes -- scalars, every gradient, bf16 vs the oracle -- lives in test_cfg2_parity_gpu.py)


def test_bf16_drift_at_full_tile_vs_fp32_native():
    """bf16 throughput mode vs the fp32-parity mode of the SAME native path at BASELINE configs[1] shapes
    (batch 2): losses over 3 steps (one generator update, three critic updates).  Reported, loosely bounded."""
    res = {}
    for dtype in ("f32", "bf16"):
        eng, *_, xc, xf = make(2, 128, 128, 2, 16, dtype)
        out = []
        for step in range(3):
            alpha = torch.from_numpy(synthetic.alpha(2, step)).cuda()
            ran_g = eng.train_step(xc, xf, alpha)
            out.append(eng.read_scalars(ran_g))
        res[dtype] = out
        del eng
        torch.cuda.empty_cache()
    drift = {}
    for step in range(3):
        for k in ("critic_loss", "gp_ret", "c_real_mean", "c_fake_mean"):
            a, b = res["bf16"][step][k], res["f32"][step][k]
            drift[f"{step}:{k}"] = abs(a - b) / max(abs(b), 1e-3)
    for d in ("gpurun_out",):        # scratch; copied into profiles/ by hand (a test run never rewrites tracked evidence)
        try:
            os.makedirs(os.path.join(os.path.dirname(GOLD), "..", d), exist_ok=True)
            with open(os.path.join(os.path.dirname(GOLD), "..", d, "bf16_vs_f32_native_cfg2_b2.json"), "w") as f:
                json.dump({"what": "bf16 mode vs fp32-parity mode of the native path, 2ch 128->1024, F=128, 16 RRDBs, B=2, 3 train steps "
                                   "(updates applied: step 0 is before any update); |a-b| / max(|b|, 1e-3)", "drift": drift,
                           "fp32": res["f32"], "bf16": res["bf16"]}, f, indent=1)
        except OSError:
            pass
    print("bf16 vs fp32 native, 128->1024 tile, B=2:", {k: f"{v:.2e}" for k, v in drift.items()})
    print("fp32:", [{k: round(v, 5) for k, v in r.items() if k in ("critic_loss", "gp_ret", "c_real_mean", "c_fake_mean")} for r in res["f32"]])
    # bounds = 3 x the drift observed on MI355X (profiles/bf16_vs_f32_native_cfg2_b2.json: step 0 <= 8.5e-5).  After the first
    # Adam update the trajectories separate: that update moves every one of the 5.5e8 parameters by lr * sign(gradient), noise-level
    # gradient entries included, so two builds whose bf16 gradients agree with the oracle equally well (profiles/bf16_drift_cfg2.json,
    # identical to three digits) gave means 5.4e-3 and 2.1e-2 apart at step 1 (critic_loss 2e-4 / 5.8e-4; 2.6e-2 / 3.4e-2 at step 2)
    assert max(v for k, v in drift.items() if k.startswith("0:")) < 2.6e-4          # before any update: pure kernel rounding
    assert max(v for k, v in drift.items() if k.startswith("1:")) < 6.3e-2
    assert all(v < 0.1 for v in drift.values())


def test_bf16_learning_curve_tracks_fp32_on_a_learnable_task():
    """Beyond the first steps: on a LEARNABLE task (fine = bilinear x8 of the coarse field + a fixed pattern, a new batch every
    step; tools/train_drift.py) the content loss at the six generator steps of a 30-step run is the same in bf16 (the benchmarked
    precision) as in the fp32-parity mode to 5e-3 relative (observed <= 2e-4; the 200-step curves are in
    profiles/train_drift_learnable.json).  On the benchmark's pure-noise tiles the step is chaotic and no such comparison exists.

    The critic loss swings over +-600 in these steps and crosses zero (..., 640, -282, -110, -14, 28, ...), so a relative bound alone
    fails near the crossing.  The yardsticks for the absolute part are MEASURED here, not typed in.  (i) The fp32 mode is run three
    times -- twice as it is (the order of its fp32 atomics differs from run to run) and once in deterministic mode (index-ordered
    reductions): their spread is what 1e-7-sized disturbances grow to (recorded on MI355X: 4e-5 at step 2, 1e-2 at step 8, 9e-2 at
    the +640 peak).  (ii) The fp32 mode once more with its INITIAL parameters rounded to bf16: what ONE bf16-sized disturbance grows
    to.  bf16, which rounds weights and activations at every step, may be off by max(2 % of the value, 10 x the fp32 spread at that
    step, 10 x the deviation of the rounded-start fp32 run at that step): one order of magnitude above what a disturbance of the same
    size does to the fp32 run itself -- the response is heavy-tailed (two builds of the bf16 kernels that differ only in the
    order of one fp32 addition in the conv epilogue were 0.28 and 1.09 off at step 8, where the rounded-start run is 0.29 off).
    All five curves go to gpurun_out/learning_curve_yardstick.json."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_drift", os.path.join(root, "tools", "train_drift.py"))
    td = importlib.util.module_from_spec(spec); spec.loader.exec_module(td)
    cfg = dict(steps=30, B=4, S=32, F_=128, cin=2, nrb=2)
    runs = {"f32_a": td.summarise(td.run("f32", **cfg)), "f32_b": td.summarise(td.run("f32", **cfg)),
            "f32_deterministic": td.summarise(td.run("f32", deterministic=True, **cfg)),
            "f32_rounded_start": td.summarise(td.run("f32", round_init=True, **cfg)), "bf16": td.summarise(td.run("bf16", **cfg))}
    a = [v for _, v in runs["f32_a"]["content_loss_generator_steps"]]
    b = [v for _, v in runs["bf16"]["content_loss_generator_steps"]]
    assert len(a) == 6 and all(abs(x - y) <= 5e-3 * x for x, y in zip(a, b)), (a, b)
    n = 10
    f32 = [runs[k]["critic_loss"][:n] for k in ("f32_a", "f32_b", "f32_deterministic")]
    cl16 = runs["bf16"]["critic_loss"][:n]
    spread = [max(c) - min(c) for c in zip(*f32)]
    one_rounding = [abs(x - y) for x, y in zip(f32[0], runs["f32_rounded_start"]["critic_loss"][:n])]
    bound = [max(2e-2 * abs(x), 10 * sp, 10 * dv) for x, sp, dv in zip(f32[0], spread, one_rounding)]
    diff = [abs(x - y) for x, y in zip(f32[0], cl16)]
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "learning_curve_yardstick.json"), "w") as f:
            json.dump({"what": "critic loss of the first 10 steps of the learnable task: fp32 mode twice, fp32 deterministic mode, bf16; "
                               "fp32 with bf16-rounded initial parameters; spread = max - min of the three plain fp32 runs per step; "
                               "bound = max(2e-2 |x|, 10 spread, 10 |f32_rounded_start - f32_a|)",
                       "config": cfg, "critic_loss": {k: v["critic_loss"][:n] for k, v in runs.items()}, "spread_f32": spread,
                       "f32_rounded_start_minus_f32": one_rounding, "bf16_minus_f32": diff, "bound": bound,
                       "content_loss_generator_steps": {k: v["content_loss_generator_steps"] for k, v in runs.items()}}, f, indent=1)
    except OSError:
        pass
    print("learning-curve yardstick: fp32 spread", [f"{s:.2e}" for s in spread], "one rounding", [f"{d:.2e}" for d in one_rounding],
          "bf16 - fp32", [f"{d:.2e}" for d in diff])
    assert all(d <= bd for d, bd in zip(diff, bound)), (diff, bound, spread, one_rounding)


def test_hip_graph_replay_equals_eager():
    """critic / generator iterations captured into HIP graphs reproduce the eager launches (cfg1, 6 steps, fp32:
    only the order of the fp32 atomic accumulations differs between the two runs)."""
    res = {}
    for graphs in (False, True):
        eng, *_, xc, xf = make(4, 16, 16, 2, 16, "f32")
        if graphs:
            eng.enable_graphs(xc, xf)
        out = []
        for step in range(6):
            alpha = torch.from_numpy(synthetic.alpha(4, step)).cuda()
            ran_g = eng.train_step(xc, xf, alpha)
            out.append(eng.read_scalars(ran_g))
        res[graphs] = out
    for a, b in zip(res[False], res[True]):
        for k in a:
            # w_estimate is a cancelling difference of two O(0.1) means: absolute bound for it.  tools/graph_flake.py (20
            # repetitions on MI355X): every scalar but w_estimate agrees to <= 4e-7, the trajectories of two runs separate by
            # the order of the fp32 atomics alone, eager vs eager as much as eager vs replay
            assert rel(a[k], b[k]) < 5e-5 or abs(a[k] - b[k]) < 5e-6, (k, a[k], b[k])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_stacked_critic_passes_equal_separate_passes(dtype):
    """TrainEngine(stacked=True) (opt-in, DESIGN 7: real | fake | x-hat through the critic as one batch of 3B) against the three
    separate passes on the HIP kernels: two train steps with updates at F = 128 (bit masks, compact inputs), 8 -> 64 tiles."""
    res = {}
    for stacked in (True, False):
        ops = HipOps(dtype)
        B, S, F_, nrb = 2, 8, 128, 1
        eng = TrainEngine(ops, S, F_, 2, B, HyperParams(batch_size=B), num_res_blocks=nrb, stacked=stacked)
        assert eng.stacked == stacked and eng.compact2
        eng.G.load_state_dict(synthetic.generator_params(F_, 2, 2, nrb))
        eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
        coarse, fine = synthetic.tiles(B, 2, S)
        xc = ops.zeros(B, S, S, eng.G.cin_p); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
        xf = ops.zeros(B, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(torch.from_numpy(fine).cuda(), xf)
        eng.critic_iteration(xc, xf, torch.from_numpy(synthetic.alpha(B, 0)).cuda(), apply_update=False)
        grads = eng.C.grad_dict()
        out = [eng.read_scalars()]
        for step in range(2):
            ran_g = eng.train_step(xc, xf, torch.from_numpy(synthetic.alpha(B, step)).cuda())
            out.append(eng.read_scalars(ran_g))
        res[stacked] = (grads, out)
    (g1, o1), (g0, o0) = res[True], res[False]
    gtol = 2e-5 if dtype == "f32" else 2e-2        # bf16: the longer grids change the order of the fp32 atomics only, but
    for k in g0:                                   # near-cancelling entries of bf16 products move by ulps of the products
        assert float((g1[k] - g0[k]).norm()) <= gtol * float(g0[k].norm()) + 1e-12, k
    stol = 1e-5 if dtype == "f32" else 2e-2
    for a, b in zip(o1, o0):
        for k in ("c_real_mean", "c_fake_mean", "gp_ret", "critic_loss"):
            assert rel(a[k], b[k]) < stol or abs(a[k] - b[k]) < 1e-5, (k, a[k], b[k])
