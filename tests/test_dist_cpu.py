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
    eng, _ = _build(1, dist, global_b=world)
    xc, xf, alpha = _data(rank, rank + 1, world)
    metrics = eng.metrics_pass(xc, xf)          # before the update: global-batch min/max (all-reduce MIN/MAX) and means
    ran_g = eng.train_step(xc, xf, alpha)       # step 0: critic + generator update
    assert eng.G.P._pending is not None         # the generator update is parked behind the next critic iteration's real pass
    eng.train_step(xc, xf, alpha)               # step 1: critic only; its deferred all-reduce + Adam complete in state_dict()
    assert eng.C.P._pending is not None         # the critic update is parked behind the (next) generator forward
    scal = eng.read_scalars(ran_g)
    torch.save({"C": eng.C.state_dict(), "G": eng.G.state_dict(), "scal": scal, "metrics": metrics}, os.path.join(outdir, f"r{rank}.pt"))
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
        port = 29600 + os.getpid() % 200 + world
        mp.spawn(_worker, args=(world, port, d, bucket_elems), nprocs=world, join=True)
        r0 = torch.load(os.path.join(d, "r0.pt"))
        r1 = torch.load(os.path.join(d, f"r{world - 1}.pt"))
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
