"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed with the
``nccl`` backend (= RCCL over xGMI on ROCm).  The reference is single-device
(reference DoWnGAN/config/config.py:25); this is the new component of SURVEY.md §8(e).

Each network keeps ONE flat fp32 gradient buffer, so the exchange is a handful of large in-place
all-reduces (bucketed so that several can be in flight on RCCL's stream while the next bucket is
enqueued) instead of one call per parameter.
"""
from __future__ import annotations

import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this pool's driver

import torch  # noqa: E402
import torch.distributed as td  # noqa: E402


def free_port() -> int:
    """An unused TCP port on 127.0.0.1, for the launcher to export as MASTER_PORT to all of its ranks."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


class Dist:
    def __init__(self, backend=None, bucket_elems=64 * 1024 * 1024):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.bucket_elems = bucket_elems
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        self.backend = backend
        if self.world_size > 1 and not td.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                # every launcher (bench.py, torch.distributed.run, tests) exports the port; ranks started some other way cannot agree
                # on a free one by themselves, and a fixed fallback collides with whatever else runs on the node
                raise RuntimeError("Dist: WORLD_SIZE > 1 but MASTER_PORT is not set; export the rendezvous port every rank shares "
                                   "(downgan_amd.dist.free_port() picks an unused one in the launcher)")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            td.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)

    def allreduce_sum_begin(self, flat: torch.Tensor):
        """Enqueue the bucketed in-place sum of ``flat`` over the ranks (RCCL runs them on its own stream, after the work
        already queued on the current stream) and return the handles; nothing waits yet."""
        works = []
        if self.world_size == 1:
            return works
        n = flat.numel()
        for off in range(0, n, self.bucket_elems):
            works.append(td.all_reduce(flat[off:min(n, off + self.bucket_elems)], op=td.ReduceOp.SUM, async_op=True))
        return works

    overlap = None      # {"finishes": n, "already_complete": k}: set by track_overlap()

    def track_overlap(self):
        """Count, from now on, how many deferred exchanges had ALREADY completed when their ``allreduce_finish`` was reached (a
        non-blocking query of the last bucket's work handle): the first multi-GPU run then shows whether the overlap with the
        following forward pass is real (bench.py reports the counts)."""
        self.overlap = {"finishes": 0, "already_complete": 0}

    def allreduce_finish(self, works):
        """Make the current stream (nccl) / the host (gloo) wait for the enqueued all-reduces."""
        if self.overlap is not None and works:
            self.overlap["finishes"] += 1
            try:
                self.overlap["already_complete"] += int(bool(works[-1].is_completed()))
            except Exception:      # a backend without a non-blocking query
                pass
        for w in works:
            w.wait()

    def any_rank(self, flag: bool) -> bool:
        """Logical OR of a per-rank condition over all ranks (one small MAX all-reduce): so that every rank takes the same
        decision -- e.g. raises together instead of one rank raising and the others hanging in the next collective."""
        if self.world_size == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if self.backend == "nccl":
            t = t.cuda()
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return bool(int(t.item()))

    def allreduce_sum_(self, flat: torch.Tensor):
        self.allreduce_finish(self.allreduce_sum_begin(flat))

    def reduce_scalars(self, d: dict, mean=(), total=()):
        if self.world_size == 1:
            return d
        keys = list(mean) + list(total)
        t = torch.tensor([d[k] for k in keys], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.cuda()
        td.all_reduce(t, op=td.ReduceOp.SUM)
        vals = t.cpu().tolist()
        out = dict(d)
        for k, v in zip(keys, vals):
            out[k] = v / self.world_size if k in mean else v
        return out

    def minmax_(self, mm, c):
        """In-place global min / max of a [rows, 2*c] tensor laid out (min_0, max_0, min_1, max_1, ...) per row: the
        batch-wide channel extrema of the MS-SSIM normalisation (losses.py:15-29) when the batch is sharded over ranks."""
        if self.world_size == 1:
            return
        t = mm.view(-1, c, 2)
        t = t if self.backend == "nccl" else t.cpu()
        lo, hi = t[..., 0].contiguous(), t[..., 1].contiguous()
        td.all_reduce(lo, op=td.ReduceOp.MIN)
        td.all_reduce(hi, op=td.ReduceOp.MAX)
        mm.view(-1, c, 2)[..., 0].copy_(lo)
        mm.view(-1, c, 2)[..., 1].copy_(hi)

    def barrier(self):
        if self.world_size > 1:
            if self.backend == "nccl":       # name the rank's own device: never let the barrier guess one
                td.barrier(device_ids=[torch.cuda.current_device()])
            else:
                td.barrier()
