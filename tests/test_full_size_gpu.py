"""Size-independent properties at BASELINE.json configs[1] FULL launch sizes (batch 32, 1024x1024 tiles: tensors of up to 8.6 GB,
i.e. byte offsets beyond 2^32 and grids of 131 072 workgroups), where the oracle is out of reach:

* batch replication: a batch of 32 copies of one image must give 32 identical outputs, equal to the batch-1 result
  (catches any addressing error at high offsets / large grids), for forward and data gradient;
* linearity in the batch: the weight gradient of the replicated batch is 32x the batch-1 gradient (fp32 accumulation order
  differs: relative 1e-4);
on the layers with the largest tensors of the step -- the critic's first three convs (critic.py:21-42; im2col kernel, the stride-2
halo kernel, the 256-channel stride-1 layer) and the generator's last 128-channel conv at 1024^2 (generator.py:77) -- in bf16 and,
for the critic's wide layers, on the fp8 kernel."""
import pytest
import torch

from downgan_amd.ops import Conv, HipOps

pytestmark = pytest.mark.gpu
B = 32

LAYERS = [
    # name, H, Cin, Cout, stride, cin_real, net, modes
    ("critic features.0", 1024, 16, 128, 1, 2, "C", ("bf16",)),
    ("critic features.2", 1024, 128, 128, 2, 0, "C", ("bf16", "fp8")),
    ("critic features.4", 512, 128, 256, 1, 0, "C", ("bf16", "fp8")),
    ("critic features.6", 512, 256, 256, 2, 0, "C", ("bf16", "fp8")),      # 8-wave stride-2 tile
    ("generator conv3.0", 1024, 128, 128, 1, 0, "", ("bf16",)),
]


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0].replace(" ", "_") for l in LAYERS])
def test_batch_replication_and_linearity_at_full_size(layer):
    name, H, ci, co, st, cin_real, net, modes = layer
    g = torch.Generator().manual_seed(100 + [l[0] for l in LAYERS].index(name))
    x1 = torch.randn(1, H, H, ci, generator=g).to(torch.bfloat16)
    if cin_real:
        x1[..., cin_real:] = 0
    w = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(torch.bfloat16).cuda()
    dy1 = torch.randn(1, H // st, H // st, co, generator=g).to(torch.bfloat16).cuda()
    x1 = x1.cuda()
    for mode in modes:
        ops = HipOps("bf16", f8_critic=mode == "fp8")
        cv1 = Conv(1, H, H, ci, co, st, False, cin_real=cin_real, net=net)
        cvB = Conv(B, H, H, ci, co, st, False, cin_real=cin_real, net=net)
        # forward
        y1 = ops.zeros(*ops.out_shape(cv1))
        ops.conv_fwd(cv1, x1, w, y1, act=0.2)
        xB = x1.expand(B, H, H, ci).contiguous()
        yB = ops.zeros(*ops.out_shape(cvB))
        ops.conv_fwd(cvB, xB, w, yB, act=0.2)
        assert ops.lib.dg_last_conv_kernels() == (32 if mode == "fp8" else 16 if cin_real else 8)
        assert float(y1.float().abs().max()) > 0
        for b in (0, 1, 15, 30, 31):
            assert torch.equal(yB[b], y1[0]), (name, mode, "fwd", b)
        assert torch.equal(yB, y1.expand_as(yB)), (name, mode, "fwd")
        del yB
        # weight gradient (bf16 kernels in every mode): linear in the batch
        dw1 = torch.zeros(co * 9 * ci, dtype=torch.float32).cuda()
        ops.conv_wgrad(cv1, x1, dy1, dw1)
        dyB = dy1.expand(B, H // st, H // st, co).contiguous()
        dwB = torch.zeros(co * 9 * ci, dtype=torch.float32).cuda()
        ops.conv_wgrad(cvB, xB, dyB, dwB)
        err = float((dwB - B * dw1).abs().max()) / float((B * dw1).abs().max())
        assert err < 1e-4, (name, mode, "wgrad", err)
        del xB
        # data gradient (not for the 2-channel layer: its input gradient is a 16-channel-output launch of another kernel family,
        # covered at its full size by the GP pass of test_cfg2_parity_gpu.py)
        if not cin_real:
            dx1 = ops.zeros(1, H, H, ci)
            ops.conv_dgrad(cv1, dy1, w, dx1)
            dxB = ops.zeros(B, H, H, ci)
            ops.conv_dgrad(cvB, dyB, w, dxB)
            assert float(dx1.float().abs().max()) > 0
            assert torch.equal(dxB, dx1.expand_as(dxB)), (name, mode, "dgrad")
            del dxB
        del dyB
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------------------
# The WHOLE train step at the benchmarked batch (DoWnGAN/config/hyperparams.py:18 batch_size = 32; wasserstein.py:27-83).
# The oracle reaches configs[1] shapes at batch 1 only (tests/test_cfg2_parity_gpu.py); this chains the benchmarked batch to it:
# a batch of N copies of ONE (coarse, fine) sample with one alpha repeated must reproduce the batch-1 step -- every per-sample term
# is normalised by the batch (a power of two: the 1/N scalings are exact), so every scalar, the generated field of every replica and
# EVERY parameter gradient of both networks equal the batch-1 values up to fp32 summation order.  Unlike the per-layer test above
# this runs every launch of the step at its full size -- the pixel-shuffle stores into [N,1024,1024,128], conv3.2 and its swapped
# backward, the penalty's 16->128 data gradient, interpolate / sumsq / scale / l1 over the megapixel fields, the Linears at N rows,
# the dense-block weight gradients, Adam -- at byte offsets beyond 2^32.
def _full_step(dtype, n, pg, pc, tc, tf, alpha1):
    import numpy as np
    from downgan_amd.engine import HyperParams, TrainEngine
    S, F_, NRB = 128, 128, 16
    ops = HipOps(dtype)
    eng = TrainEngine(ops, S, F_, 2, n, HyperParams(batch_size=n), num_res_blocks=NRB)
    eng.G.load_state_dict(pg)
    eng.C.load_state_dict(pc)
    xc1 = ops.zeros(1, S, S, eng.G.cin_p); ops.nchw_to_nhwc(tc.cuda(), xc1)
    xf1 = ops.zeros(1, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(tf.cuda(), xf1)
    xc, xf = xc1.expand(n, -1, -1, -1).contiguous(), xf1.expand(n, -1, -1, -1).contiguous()
    alpha = torch.full((n,), float(alpha1), dtype=torch.float32).cuda()
    eng.critic_iteration(xc, xf, alpha, apply_update=False, save_g=True)
    eng.generator_iteration(xc, xf, apply_update=False, reuse_fake=True)
    res = dict(scal=eng.read_scalars(True), fake=eng.G.fake.clone() if n == 1 else None,
               cg=eng.C.P.g.clone(), gg=eng.G.P.g.clone(), centries=dict(eng.C.P.entries), gentries=dict(eng.G.P.entries))
    if n > 1:
        f0 = eng.G.fake[0]
        res["fake0"] = f0.clone()
        res["replicas_equal"] = all(torch.equal(eng.G.fake[b], f0) for b in sorted({1, n // 2 - 1, n - 1}))
        g0 = eng.gbuf[0].float()                 # d C / d fake of the generator iteration: one value per replica up to the FC's atomics
        res["gbuf_spread"] = max(float((eng.gbuf[b].float() - g0).norm() / (g0.norm() + 1e-30)) for b in (1, n - 1))
    # one whole train step WITH both Adam updates (wasserstein.py:131-147), then the next critic iteration's scalars
    cp0, gp0 = eng.C.P.p.clone(), eng.G.P.p.clone()
    assert eng.train_step(xc, xf, alpha)
    eng.C.P.sync(); eng.G.P.sync()
    res["c_move"], res["g_move"] = torch.sign(eng.C.P.p - cp0).to(torch.int8), torch.sign(eng.G.P.p - gp0).to(torch.int8)
    del cp0, gp0
    assert not eng.train_step(xc, xf, alpha)
    res["scal1"] = eng.read_scalars(False)
    res["hbm_gib"] = torch.cuda.max_memory_allocated() / 2 ** 30
    del eng, xc, xf
    torch.cuda.empty_cache()
    return res


FULL_STEP = [("f32", 16, 1e-5, 1e-4), ("bf16", 32, 1e-3, 1e-3)]       # mode, batch, scalar bound, per-parameter gradient rel-l2 bound


@pytest.mark.parametrize("case", FULL_STEP, ids=[f"{c[0]}_b{c[1]}" for c in FULL_STEP])
def test_whole_train_step_at_benchmarked_batch_equals_batch_one(case):
    """fp32-parity mode at batch 16 (the same 8.6-GB tensors as bf16 at batch 32; batch 32 in fp32 would need 2 x 142 GiB) with tight
    bounds, and bf16 at batch 32 -- the benchmarked precision and batch -- against the batch-1 run of the same mode, which
    tests/test_cfg2_parity_gpu.py holds to the pinned oracle."""
    import json
    import os
    from downgan_amd import synthetic
    dtype, n, s_tol, g_tol = case
    S, F_, NRB = 128, 128, 16
    pg = {k: torch.from_numpy(v) for k, v in synthetic.generator_params(F_, 2, 2, NRB).items()}
    pc = {k: torch.from_numpy(v) for k, v in synthetic.critic_params(F_, 8 * S, 2).items()}
    coarse, fine = synthetic.tiles(1, 2, S)
    tc, tf = torch.from_numpy(coarse), torch.from_numpy(fine)
    a1 = float(synthetic.alpha(1, 0)[0])
    one = _full_step(dtype, 1, pg, pc, tc, tf, a1)
    big = _full_step(dtype, n, pg, pc, tc, tf, a1)
    rel = lambda a, b: abs(a - b) / max(abs(a), abs(b), 1e-30)
    report = {"mode": dtype, "batch": n, "hbm_gib": big["hbm_gib"], "scalars": {}, "gbuf_replica_spread": big["gbuf_spread"]}
    # the generated field: bit-identical in every replica and to the batch-1 field (no atomics in a conv forward)
    assert big["replicas_equal"], "G(coarse) differs between replicas of one sample"
    assert torch.equal(big["fake0"], one["fake"][0]), "G(coarse) at batch N differs from batch 1"
    assert big["gbuf_spread"] < g_tol, big["gbuf_spread"]
    for k, v in one["scal"].items():
        report["scalars"][k] = {"b1": v, f"b{n}": big["scal"][k], "rel": rel(v, big["scal"][k])}
        if k == "w_estimate":      # c_real_mean - c_fake_mean: at initialisation a 1e-5 difference of two 2.5e-2 terms -- on THEIR scale
            assert abs(v - big["scal"][k]) < s_tol * max(abs(one["scal"]["c_real_mean"]), abs(one["scal"]["c_fake_mean"])), (k, v, big["scal"][k])
        else:
            assert rel(v, big["scal"][k]) < s_tol, (k, v, big["scal"][k])
    worst = {}
    for net, key, ent in (("C", "cg", "centries"), ("G", "gg", "gentries")):
        errs = {}
        for name, (off, cnt, _) in one[ent].items():
            a, b = one[key][off:off + cnt], big[key][off:off + cnt]
            den = float(a.norm())
            errs[name] = float((a - b).norm()) / den if den > 0 else float(b.norm())
        w = max(errs, key=errs.get)
        worst[net] = (w, errs[w])
        report[f"grad_rel_l2_max_{net}"] = {"param": w, "err": errs[w], "median": sorted(errs.values())[len(errs) // 2]}
        assert errs[w] < g_tol, (net, w, errs[w])
    # Adam's first move is -lr * sign(g) per entry: the share of entries (weighted by |g|) that move the same way
    for net, key, mv in (("C", "cg", "c_move"), ("G", "gg", "g_move")):
        w = one[key].abs().double()
        same = (one[mv] == big[mv]).double()
        share = float((same * w).sum() / w.sum())
        report[f"adam_first_move_sign_agreement_{net}"] = {"weighted": share, "unweighted": float(same.mean())}
        assert share >= 0.9999, (net, share)
    # the critic iteration after both updates (sign flips of the entries with g ~ 0 move parameters by 2 lr: not bit-equal)
    report["step1_scalars"] = {k: {"b1": v, f"b{n}": big["scal1"][k], "rel": rel(v, big["scal1"][k])} for k, v in one["scal1"].items()}
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", f"full_step_b{n}_{dtype}.json"), "w") as f:
            json.dump(report, f, indent=1)
    except OSError:
        pass
    print("whole step at batch", n, dtype, worst, report["step1_scalars"])
    for k, v in report["step1_scalars"].items():
        if k != "w_estimate":
            assert v["rel"] < 50 * s_tol + 1e-4, (k, v)
