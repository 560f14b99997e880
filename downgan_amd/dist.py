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
            os.environ.setdefault("MASTER_PORT", "29533")
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

    @staticmethod
    def allreduce_finish(works):
        """Make the current stream (nccl) / the host (gloo) wait for the enqueued all-reduces."""
        for w in works:
            w.wait()

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
