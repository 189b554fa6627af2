"""Multi-process CPU tests (gloo, world_size 2) of the data-parallel path: the flat-gradient all-reduce, the per-rank
sharding of micro-batches and the N-rank == 1-rank gradient identity it relies on (pairs are independent; the DPA loss of a
global batch is the mean of the per-rank micro-batch losses when every rank holds the same number of pairs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from halva_amd import dp
    ctx = dp.DistContext.from_env("gloo")
    assert (ctx.rank, ctx.world) == (rank, world)
    # 1. bucketed mean all-reduce of a flat fp32 buffer (bucket boundary exercised by shrinking the bucket)
    dp.BUCKET_ELEMS = 1000
    flat = torch.arange(2500, dtype=torch.float32) * (rank + 1)
    dp.allreduce_mean_(flat, ctx)
    assert torch.allclose(flat, torch.arange(2500, dtype=torch.float32) * 1.5)
    # 2. toy "pairs": loss_b = softplus(w . x_b); global batch of 8 pairs, rank r owns pairs r, r+2, ... (micro-batches of 2)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(8, 16, generator=g)
    w = torch.randn(16, generator=g)
    batches = [[0, 1], [2, 3], [4, 5], [6, 7]]
    mine = dp.shard_batches(len(batches), ctx)
    grad = torch.zeros(16)
    for bi in mine:                                           # gradient accumulation over this rank's micro-batches
        ww = w.clone().requires_grad_(True)
        loss = torch.nn.functional.softplus(X[batches[bi]] @ ww).mean() / len(mine)
        loss.backward()
        grad += ww.grad
    dp.allreduce_mean_(grad, ctx)
    ww = w.clone().requires_grad_(True)
    torch.nn.functional.softplus(X @ ww).mean().backward()
    assert torch.allclose(grad, ww.grad, atol=1e-6)
    assert abs(dp.mean_scalar(float(rank), ctx) - 0.5) < 1e-12
    # 3. the trainer's loader shards the sampler's global batch list disjointly and completely
    from llava.train.halva_trainer import HalvaTrainer

    class DS(torch.utils.data.Dataset):
        modality_lengths = list(range(5, 5 + 24))

        def __len__(self):
            return 24

        def __getitem__(self, i):
            return i
    t = HalvaTrainer.__new__(HalvaTrainer)
    t.dist, t.train_dataset, t.data_collator = ctx, DS(), (lambda x: x)
    t.args = type("A", (), dict(per_device_train_batch_size=2, gradient_accumulation_steps=2, group_by_modality_length=True,
                                 dataloader_num_workers=0, dataloader_drop_last=False, seed=42))()
    torch.manual_seed(7)
    seen = [i for b in t.get_train_dataloader() for i in b]
    gathered = [None, None]
    dist.all_gather_object(gathered, seen)
    if rank == 0:
        assert sorted(gathered[0] + gathered[1]) == list(range(24))
        assert not set(gathered[0]) & set(gathered[1])
        out.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"
