"""Pin the CPU oracle (oracle/ref_step.py) to golden vectors captured from the real reference.

Fixtures: tests/golden/{cfg1,cin6_small,f32_s32}.json, written by tests/golden/make_golden.py,
which ran the reference's own Generator / Critic / WassersteinGAN on this repo's synthetic inputs.
"""
import json
import os

import numpy as np
import pytest
import torch

from downgan_amd import synthetic
from oracle import ref_step

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def build(gold):
    c = gold["config"]
    B, S, F_, cin, nrb = c["B"], c["S"], c["F"], c["cin"], c["num_res_blocks"]
    pg = {k: torch.from_numpy(v) for k, v in synthetic.generator_params(F_, cin, 2, nrb).items()}
    pc = {k: torch.from_numpy(v) for k, v in synthetic.critic_params(F_, 8 * S, 2).items()}
    coarse, fine = synthetic.tiles(B, cin, S, mask_channel=(2 if cin > 2 else None))
    hp = ref_step.HP(batch_size=B)
    tr = ref_step.OracleTrainer(pg, pc, hp, num_res_blocks=nrb)
    return tr, torch.from_numpy(coarse), torch.from_numpy(fine), B


def close(a, b, rel=1e-5, abs_=1e-7):
    return abs(a - b) <= rel * max(abs(a), abs(b)) + abs_


def check_summary(t, s, rel=2e-5):
    f = t.detach().double().flatten()
    assert close(float(f.norm()), s["l2"], rel, 1e-9), (float(f.norm()), s["l2"])
    for i, v in zip(s["idx"], s["val"]):
        assert close(float(f[i]), v, 1e-4, 1e-8 + 1e-5 * s["l2"] / max(1.0, f.numel() ** 0.5)), (i, float(f[i]), v)


@pytest.mark.parametrize("name,nsteps", [("cfg1", 6), ("cin6_small", 1), ("f32_s32", 2)])
def test_oracle_matches_reference_golden(name, nsteps):
    torch.set_num_threads(os.cpu_count())
    gold = load(name)
    tr, coarse, fine, B = build(gold)
    with torch.no_grad():
        g = tr.G(coarse)
        check_summary(g, gold["forward"]["G_coarse"])
        c = tr.C(fine).flatten().tolist()
    for a, b in zip(c, gold["forward"]["C_fine"]["values"]):
        assert close(a, b, 1e-5, 1e-7)
    m = tr.metrics(coarse, fine)        # metrics pass before any update (mlflow_epoch.py:53-63)
    for k, v in gold["forward"]["metrics"].items():
        assert close(m[k], v, 2e-5, 2e-7), (k, m[k], v)
    for step in range(nsteps):
        rec = gold["steps"][step]
        alpha = torch.from_numpy(synthetic.alpha(B, step))
        out, cg = tr.critic_iteration(coarse, fine, alpha)
        for k in ["c_real_mean", "c_fake_mean", "gp_ret", "critic_loss", "gradient_penalty"]:
            assert close(out[k], rec[k], 2e-5, 1e-7), (step, k, out[k], rec[k])
        assert abs(out["w_estimate"] - rec["w_estimate"]) < 2e-6
        for k, s in rec["C_grads"].items():
            check_summary(cg[k], s, rel=1e-4)
        for k, s in rec["C_params_after"].items():
            check_summary(tr.PC[k], s, rel=1e-5)
        if step % tr.hp.critic_iterations == 0:
            gout, gg = tr.generator_iteration(coarse, fine)
            for k in ["g_loss", "content_loss", "g_c_fake_mean"]:
                assert close(gout[k], rec[k], 2e-5, 1e-7), (step, k, gout[k], rec[k])
            for k, s in rec["G_grads"].items():
                check_summary(gg[k], s, rel=2e-4)
            for k, s in rec["G_params_after"].items():
                assert close(float(tr.PG[k].detach().double().norm()), s["l2"], 1e-5, 1e-9)
        tr.num_steps += 1


def test_param_specs_match_reference_counts():
    # cfg1 generator 1 695 794 params, critic 1 112 313
    g = sum(int(np.prod(s)) for _, s in synthetic.generator_param_specs(16, 2))
    c = sum(int(np.prod(s)) for _, s in synthetic.critic_param_specs(16, 128, 2))
    assert g == 1695794 and c == 1112313
