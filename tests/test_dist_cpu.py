"""Data-parallel path on the CPU: 2 ranks over gloo, each with half of the global batch, must produce
the same parameters as one process with the whole batch (gradient all-reduce of the flat buffers,
global-batch normalisation of the per-sample terms, scalar reduction).  The engine runs on the
emulated ops (oracle/emu_ops.py); the real kernels are covered by the -m gpu tests."""
import os
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from downgan_amd import synthetic
from downgan_amd.engine import HyperParams, TrainEngine
from downgan_amd.layout import nchw_to_nhwc_padded

CFG = dict(S=16, F_=16, cin=2, nrb=1)


def _build(B, dist=None, global_b=None):
    from oracle.emu_ops import EmuOps
    ops = EmuOps("f32")
    eng = TrainEngine(ops, CFG["S"], CFG["F_"], CFG["cin"], B, HyperParams(batch_size=global_b or B), num_res_blocks=CFG["nrb"], dist=dist)
    eng.G.load_state_dict(synthetic.generator_params(CFG["F_"], CFG["cin"], 2, CFG["nrb"]))
    eng.C.load_state_dict(synthetic.critic_params(CFG["F_"], 8 * CFG["S"], 2))
    return eng, ops


def _data(lo, hi, gb=2):
    coarse, fine = synthetic.tiles(gb, CFG["cin"], CFG["S"])
    alpha = synthetic.alpha(gb, 0)
    tc, tf = torch.from_numpy(coarse[lo:hi]), torch.from_numpy(fine[lo:hi])
    return nchw_to_nhwc_padded(tc, 16, torch.float32), nchw_to_nhwc_padded(tf, 16, torch.float32), torch.from_numpy(alpha[lo:hi])


def _worker(rank, world, port, outdir, bucket_elems=64 * 1024 * 1024):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from downgan_amd.dist import Dist
    dist = Dist("gloo", bucket_elems=bucket_elems)
    dist.track_overlap()
    eng, _ = _build(1, dist, global_b=world)
    xc, xf, alpha = _data(rank, rank + 1, world)
    metrics = eng.metrics_pass(xc, xf)          # before the update: global-batch min/max (all-reduce MIN/MAX) and means
    ran_g = eng.train_step(xc, xf, alpha)       # step 0: critic + generator update
    assert eng.G.P._pending is not None         # the generator update is parked behind the next critic iteration's real pass
    eng.train_step(xc, xf, alpha)               # step 1: critic only; its deferred all-reduce + Adam complete in state_dict()
    assert eng.C.P._pending is not None         # the critic update is parked behind the (next) generator forward
    scal = eng.read_scalars(ran_g)
    torch.save({"C": eng.C.state_dict(), "G": eng.G.state_dict(), "scal": scal, "metrics": metrics, "overlap": dict(dist.overlap)},
               os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()


@pytest.mark.parametrize("world,bucket_elems", [(2, 64 * 1024 * 1024), (4, 100_000)])      # 4 ranks: the 1.1 M-float buffer in 12 buckets
def test_ranks_equal_single_process(world, bucket_elems):
    torch.set_num_threads(4)
    eng, _ = _build(world)
    xc, xf, alpha = _data(0, world, world)
    ref_metrics = eng.metrics_pass(xc, xf)
    ran_g = eng.train_step(xc, xf, alpha)
    eng.train_step(xc, xf, alpha)
    ref_scal = eng.read_scalars(ran_g)
    ref_c, ref_g = eng.C.state_dict(), eng.G.state_dict()
    with tempfile.TemporaryDirectory() as d:
        from downgan_amd.dist import free_port
        port = free_port()
        mp.spawn(_worker, args=(world, port, d, bucket_elems), nprocs=world, join=True)
        r0 = torch.load(os.path.join(d, "r0.pt"))
        r1 = torch.load(os.path.join(d, f"r{world - 1}.pt"))
    # three exchanges finished (critic, generator, critic), each queried for completion before its wait
    assert r0["overlap"]["finishes"] == 3 and 0 <= r0["overlap"]["already_complete"] <= 3, r0["overlap"]
    for k in ref_c:
        assert torch.equal(r0["C"][k], r1["C"][k]), k                       # replicas stay identical
        # two Adam steps: an entry whose gradient is rounding noise (|g| ~ eps) moves by up to lr per step in either
        # direction, so single entries are bounded by 2*lr*steps and the tensors as a whole by their norm
        assert torch.allclose(r0["C"][k], ref_c[k], rtol=0, atol=2 * 2.5e-4 * 2), (k, float((r0["C"][k] - ref_c[k]).abs().max()))
        assert float((r0["C"][k] - ref_c[k]).norm()) <= 2e-4 * float(ref_c[k].norm()) + 1e-6, (k, float((r0["C"][k] - ref_c[k]).norm()), float(ref_c[k].norm()))
    for k in ref_g:
        assert torch.equal(r0["G"][k], r1["G"][k]), k
        assert torch.allclose(r0["G"][k], ref_g[k], rtol=0, atol=0.25 * 2.5e-4), k
    for k in ("critic_loss", "gp_ret", "g_loss", "content_loss", "c_real_mean"):   # scalars of step 1 (g_loss of step 0)
        assert abs(r0["scal"][k] - ref_scal[k]) <= 2e-5 * max(1.0, abs(ref_scal[k])), (k, r0["scal"][k], ref_scal[k])
    for k in ("MAE", "MSE", "Wass", "MSSSIM"):          # metrics pass of the sharded batch == whole batch in one process
        assert r0["metrics"][k] == r1["metrics"][k], k
        assert abs(r0["metrics"][k] - ref_metrics[k]) <= 1e-5 * max(1.0, abs(ref_metrics[k])), (k, r0["metrics"][k], ref_metrics[k])


def _nan_worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle.emu_ops import EmuOps
    from downgan_amd.dist import Dist
    dist = Dist("gloo")
    eng = TrainEngine(EmuOps("f32"), CFG["S"], CFG["F_"], CFG["cin"], 1, HyperParams(batch_size=world), num_res_blocks=CFG["nrb"], dist=dist,
                      check_finite=True)
    eng.G.load_state_dict(synthetic.generator_params(CFG["F_"], CFG["cin"], 2, CFG["nrb"]))
    eng.C.load_state_dict(synthetic.critic_params(CFG["F_"], 8 * CFG["S"], 2))
    xc, xf, alpha = _data(rank, rank + 1, world)
    if rank == 1:
        xf[0, 3, 5, 0] = float("nan")          # only THIS rank's sample is bad
    msg = "no error"
    try:
        eng.train_step(xc, xf, alpha)
    except FloatingPointError as e:
        msg = str(e)
    with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
        f.write(msg)
    dist.barrier()                              # reachable by both ranks: neither is stuck in the gradient all-reduce


def test_check_finite_raises_on_every_rank_together():
    """``check_finite`` in a data-parallel run: a NaN seen by ONE rank (its own samples) makes EVERY rank raise before the gradient
    all-reduce -- otherwise the clean ranks would sit in that collective until its timeout and bury the diagnostic (round-3 advice)."""
    from downgan_amd.dist import free_port
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_nan_worker, args=(2, free_port(), d), nprocs=2, join=True)
        m0, m1 = (open(os.path.join(d, f"r{r}.txt")).read() for r in range(2))
    assert "non-finite values on another rank" in m0, m0
    assert "non-finite values in" in m1 and "critic iteration" in m1, m1


def test_dist_requires_an_exported_port(monkeypatch):
    from downgan_amd.dist import Dist, free_port
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        Dist("gloo")
    a, b = free_port(), free_port()
    assert 1024 < a < 65536 and 1024 < b < 65536
