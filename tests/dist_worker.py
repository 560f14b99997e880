"""One data-parallel rank of tests/test_dist_gpu.py (started by torch.distributed.run, never collected by pytest).

usage: dist_worker.py <backend: nccl|gloo> <outdir> <dtype>
  nccl: rank r drives cuda:<LOCAL_RANK> and the gradient exchange is RCCL (needs >= WORLD_SIZE GPUs);
  gloo: every rank drives cuda:0 (a rehearsal of the multi-process path on a one-GPU box; the exchange goes over gloo).
Each rank takes ONE sample of the global batch, runs the sharded metrics pass and two train steps through the HIP path
and writes its parameters + scalars to <outdir>/r<rank>.pt.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from downgan_amd import synthetic  # noqa: E402
from downgan_amd.dist import Dist  # noqa: E402
from downgan_amd.engine import HyperParams, TrainEngine  # noqa: E402
from downgan_amd.ops import HipOps  # noqa: E402

CFG = dict(S=16, F_=16, cin=2, nrb=1)
NSTEPS = 2


def build(B, global_b, dtype, device, dist=None):
    ops = HipOps(dtype, device)
    eng = TrainEngine(ops, CFG["S"], CFG["F_"], CFG["cin"], B, HyperParams(batch_size=global_b), num_res_blocks=CFG["nrb"], dist=dist)
    eng.G.load_state_dict(synthetic.generator_params(CFG["F_"], CFG["cin"], 2, CFG["nrb"]))
    eng.C.load_state_dict(synthetic.critic_params(CFG["F_"], 8 * CFG["S"], 2))
    return eng, ops


def data(ops, eng, lo, hi, global_b):
    coarse, fine = synthetic.tiles(global_b, CFG["cin"], CFG["S"])
    S = CFG["S"]
    xc = ops.zeros(hi - lo, S, S, eng.G.cin_p); ops.nchw_to_nhwc(torch.from_numpy(coarse[lo:hi]).to(ops.device), xc)
    xf = ops.zeros(hi - lo, 8 * S, 8 * S, eng.G.np_p); ops.nchw_to_nhwc(torch.from_numpy(fine[lo:hi]).to(ops.device), xf)
    alphas = [torch.from_numpy(synthetic.alpha(global_b, s)[lo:hi]).to(ops.device) for s in range(NSTEPS)]
    return xc, xf, alphas


def run(eng, xc, xf, alphas):
    out = {"metrics": eng.metrics_pass(xc, xf)}
    pend = []
    for s in range(NSTEPS):
        ran_g = eng.train_step(xc, xf, alphas[s])
        pend.append((eng.C.P._pending is not None, eng.G.P._pending is not None))
        if s == 0:
            out["scal0"] = eng.read_scalars(ran_g)
    out["scal"] = eng.read_scalars(False)
    out["pending"] = pend
    out["C"], out["G"] = eng.C.state_dict(), eng.G.state_dict()
    return out


def main():
    backend, outdir, dtype = sys.argv[1], sys.argv[2], sys.argv[3]
    dist = Dist(backend)
    local = dist.local_rank if backend == "nccl" else 0
    torch.cuda.set_device(local)
    eng, ops = build(1, dist.world_size, dtype, f"cuda:{local}", dist)
    xc, xf, alphas = data(ops, eng, dist.rank, dist.rank + 1, dist.world_size)
    out = run(eng, xc, xf, alphas)
    torch.save(out, os.path.join(outdir, f"r{dist.rank}.pt"))
    dist.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
