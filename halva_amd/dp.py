"""Data parallelism for the DPA step: one process per GPU, full replicas, ONE gradient exchange per optimizer step.

The reference shards everything with DeepSpeed ZeRO-3 over NCCL (src/json/zero3.json; per-module parameter all-gathers on
every forward / recompute / backward).  On MI355X both 7B models fit each GPU many times over (288 GB), so the only
exchange the path needs is the mean of the trainable gradients: the flat fp32 buffer of halva_amd.dpa.FlatTrainables
(LoRA 319.8 M + projector 21.0 M parameters = 1.36 GB) is all-reduced over RCCL/xGMI in a few large buckets
(xGMI is point-to-point: big buckets keep every link of the ring busy and amortise launch latency).
`HalvaTrainer.compute_loss` bypasses any wrapper forward (reference halva_trainer.py:548,573), so DDP-style hooks could
not be used anyway - the explicit all-reduce is the natural design.
"""
import os

import torch
import torch.distributed as dist

BUCKET_ELEMS = 64 * 1024 * 1024        # 256 MB of fp32 per collective


class DistContext:
    def __init__(self, rank=0, world=1, local_rank=0, group=None):
        self.rank, self.world, self.local_rank, self.group = rank, world, local_rank, group

    @classmethod
    def from_env(cls, backend=None):
        """Read RANK / WORLD_SIZE / LOCAL_RANK (torchrun or the `deepspeed` shim); initialise torch.distributed if needed.
        backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" for the CPU tests."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        return cls(rank, world, local)


def allreduce_mean_(flat, ctx):
    """In-place mean over ranks of a flat gradient buffer, bucketed.  Returns the buffer."""
    if ctx.world == 1:
        return flat
    handles = []
    for lo in range(0, flat.numel(), BUCKET_ELEMS):
        handles.append(dist.all_reduce(flat[lo:lo + BUCKET_ELEMS], op=dist.ReduceOp.SUM, group=ctx.group, async_op=True))
    for h in handles:
        h.wait()
    flat.div_(ctx.world)
    return flat


def mean_scalar(x, ctx):
    if ctx.world == 1:
        return float(x)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, group=ctx.group)
    return float(t) / ctx.world


def shard_batches(n_batches, ctx):
    """Indices of the global micro-batches this rank processes (rank r takes r, r + world, ...)."""
    return list(range(ctx.rank, n_batches, ctx.world))


def barrier(ctx):
    if ctx.world > 1:
        dist.barrier(group=ctx.group)
