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
    # 2b. GradReducer: buckets cut at "layer" boundaries, handed over top-down as the backward reports layers done; the result is
    #     the plain mean whatever the order/timing of the hand-over, and every bucket but the last can go out early
    full = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    want = torch.arange(1000, dtype=torch.float32) * 1.5
    for min_bucket, n_buckets in ((1, 5), (300, 3), (10 ** 6, 1)):
        g = full.clone()
        red = dp.GradReducer(g, ctx, boundaries=[0, 200, 400, 600, 800], min_bucket=min_bucket)
        assert len(red.buckets) == n_buckets and red.buckets[0][1] == 1000 and red.buckets[-1][0] == 0
        assert sorted(x for b in red.buckets for x in b)[0] == 0 and all(a[0] == b[1] for a, b in zip(red.buckets, red.buckets[1:]))
        red.begin()
        for lo in (800, 600, 400, 200):
            red.ready_from(lo)
        assert red.issued_early == sum(1 for b in red.buckets if b[0] >= 200)
        red.finish()
        assert torch.allclose(g, want)

    class _Flat:
        names = ["model.layers.0.a", "model.layers.0.b", "model.layers.1.a", "model.mm_projector.0.weight"]
        offsets = [0, 10, 30, 60, 100]
        grad = torch.ones(100) * (rank + 1)
    red = dp.GradReducer.for_flat(_Flat, ctx, min_bucket=1)
    # the projector tail [60, 100) lies behind the layers but is final only after the whole backward: never handed over early
    assert red.first == {0: 0, 1: 30} and red.buckets == [(30, 60), (0, 30)] and red.late == (60, 100)
    red.begin()
    red.layer_done(1)
    assert red.issued_early == 1
    _Flat.grad[60:] += 1.0               # the projector gradient arrives at the very end of the backward
    red.finish()
    assert torch.allclose(_Flat.grad[:60], torch.full((60,), 1.5)) and torch.allclose(_Flat.grad[60:], torch.full((40,), 2.5))
    _Flat.grad.fill_(rank + 1.0)
    red.begin()
    red.layer_done(1)
    red.layer_done(0)
    assert red.issued_early == 2
    red.finish()
    assert torch.allclose(_Flat.grad, torch.full((100,), 1.5))
    assert dp.max_scalar(float(rank), ctx) == 1.0
    # 2c. VILA names its projector tensors `mm_projector.layers.N.*` (halva_amd/vila_model.py): they are NOT decoder layers - they
    #     must end up in the late tail, never in a bucket that leaves from inside the backward (round-2 advisor finding)
    class _Vila:
        names = ["llm.base_model.model.model.layers.0.q.lora_A", "llm.base_model.model.model.layers.1.q.lora_A",
                 "mm_projector.layers.2.weight", "mm_projector.layers.4.weight", "mm_projector.layers.1.weight",
                 "mm_projector.layers.1.bias", "mm_projector.layers.2.bias", "mm_projector.layers.4.bias"]
        offsets = [0, 30, 60, 70, 80, 85, 90, 95, 100]
        grad = torch.ones(100) * (rank + 1)
    first, tail = dp.layer_boundaries(_Vila.names, _Vila.offsets[:-1])
    assert first == {0: 0, 1: 30} and tail == 60
    red = dp.GradReducer.for_flat(_Vila, ctx, min_bucket=1)
    assert red.buckets == [(30, 60), (0, 30)] and red.late == (60, 100)
    red.begin()
    red.layer_done(1)
    red.layer_done(0)
    _Vila.grad[60:] += 1.0               # the projector gradient arrives after the last decoder layer's
    red.finish()
    assert torch.allclose(_Vila.grad[:60], torch.full((60,), 1.5)) and torch.allclose(_Vila.grad[60:], torch.full((40,), 2.5))
    for bad in (["model.layers.1.a", "model.layers.0.a"],                            # layers out of order
                ["model.layers.0.a", "model.mm_projector.0.weight", "model.layers.1.a"]):     # a decoder tensor behind the tail
        try:
            dp.layer_boundaries(bad, list(range(0, 10 * len(bad), 10)))
            raise AssertionError("layout %r accepted" % (bad,))
        except ValueError:
            pass
    # 2d. a step that fails on ONE rank after it has handed over some buckets (out of memory inside the last backward): that rank
    #     completes the exchange with drain() - the same collectives the healthy rank issues from its backward / finish() - then both
    #     agree through a MAX all-reduce and repeat the step; the repeated step's mean is exact (nothing paired with a stale bucket)
    redf = dp.GradReducer.for_flat(_Flat, ctx, min_bucket=1)
    for fail_after in (0, 1, 2, "before_begin"):
        _Flat.grad.fill_(rank + 1.0)
        failed = rank == 1
        if not (failed and fail_after == "before_begin"):      # (e.g. out of memory in the first group: begin() never ran)
            redf.begin()
            for k, layer in enumerate((1, 0)):
                if failed and fail_after == k:
                    break
                redf.layer_done(layer)
        if failed:
            redf.drain()
        else:
            redf.finish()
        assert dp.max_scalar(1.0 if failed else 0.0, ctx) == 1.0
        _Flat.grad.fill_(rank + 1.0)      # every rank repeats the step
        redf.begin()
        redf.layer_done(1)
        redf.layer_done(0)
        redf.finish()
        assert torch.allclose(_Flat.grad, torch.full((100,), 1.5)), fail_after
    b = torch.full((5,), float(rank + 7))
    dp.broadcast_(b, ctx)
    assert torch.equal(b, torch.full((5,), 7.0))
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


def _forced_worker(port, out):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HALVA_DP_FORCE="1")
    from halva_amd import dp
    ctx = dp.DistContext.from_env("gloo")
    assert ctx.world == 1 and ctx.active and dist.is_initialized() and dist.get_world_size() == 1
    g = torch.arange(1000, dtype=torch.float32) * 0.37
    want = g.clone()
    red = dp.GradReducer(g, ctx, boundaries=[0, 200, 400, 600, 800], min_bucket=1)
    calls = []
    real = dp._allreduce_sum_async
    dp._allreduce_sum_async = lambda t, c: (calls.append(t.numel()), real(t, c))[1]
    red.begin()
    for lo in (800, 600, 400, 200):
        red.ready_from(lo)
    assert red.issued_early == 4 and len(calls) == 4        # the collectives really went out from "the backward"
    red.finish()
    assert len(calls) == 5 and torch.equal(g, want)         # a one-rank sum / 1: bit-identical
    assert dp.max_scalar(3.0, ctx) == 3.0 and dp.mean_scalar(2.5, ctx) == 2.5
    dp.barrier(ctx)
    dist.destroy_process_group()
    out.put("ok")


def test_forced_one_rank_communicator_issues_the_collectives():
    """HALVA_DP_FORCE=1: world == 1 still initialises a process group and every bucket / scalar / barrier goes through it (the
    hardware twin is tests/test_dp_engine_gpu.py::test_one_rank_rccl_communicator_...); without the flag a world of one issues none."""
    from halva_amd import dp
    ctx = dp.DistContext(0, 1, 0)
    assert not ctx.active and dp.max_scalar(1.0, ctx) == 1.0
    red = dp.GradReducer(torch.ones(10), ctx, boundaries=[0, 5], min_bucket=1)
    red.begin()
    red.ready_from(5)
    assert red.issued_early == 0
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    p = mpctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and q.get(timeout=5) == "ok"


def _worker_n(rank, world, port, accum, out):
    """The same plumbing at an arbitrary world size (round 6: every multi-process test stopped at world = 2, so port handling, the
    sampler's `world x accum` dealing and the bucket hand-over had never run with 8 processes - reference src/hallava_7b.sh:21-22,30,
    llava/train/halva_trainer.py:261-274)."""
    import random
    import time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from halva_amd import dp
    ctx = dp.DistContext.from_env("gloo")
    assert (ctx.rank, ctx.world) == (rank, world) and ctx.active
    mean_rank1 = (world + 1) / 2.0                      # mean over the ranks of (rank + 1)
    # 1. the after-the-fact bucketed mean, with a bucket boundary inside the buffer
    dp.BUCKET_ELEMS = 1000
    flat = torch.arange(2500, dtype=torch.float32) * (rank + 1)
    dp.allreduce_mean_(flat, ctx)
    assert torch.allclose(flat, torch.arange(2500, dtype=torch.float32) * mean_rank1)
    # 2. N ranks x 1 micro-batch == one process accumulating N micro-batches (pairs are independent)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(2 * world, 16, generator=g)
    w = torch.randn(16, generator=g)
    mine = dp.shard_batches(world, ctx)
    assert mine == [rank]
    ww = w.clone().requires_grad_(True)
    torch.nn.functional.softplus(X[2 * rank:2 * rank + 2] @ ww).mean().backward()
    grad = ww.grad.clone()
    dp.allreduce_mean_(grad, ctx)
    ww = w.clone().requires_grad_(True)
    torch.nn.functional.softplus(X @ ww).mean().backward()
    assert torch.allclose(grad, ww.grad, atol=1e-6)
    assert abs(dp.mean_scalar(float(rank), ctx) - (world - 1) / 2.0) < 1e-12 and dp.max_scalar(float(rank), ctx) == world - 1.0
    # 3. GradReducer: every rank reports its layers at its OWN pace and granularity (the ranks' backwards are not in lock step: one
    #    reports layer by layer, another skips hooks and reports several layers at once, a third only at finish()); the collectives
    #    still pair bucket by bucket and the result is the plain mean
    class _Flat:
        names = ["model.layers.%d.w" % i for i in range(8)] + ["model.mm_projector.0.weight"]
        offsets = list(range(0, 900, 100)) + [1000]
        grad = None
    rnd = random.Random(1000 + rank)
    for trial in range(4):
        _Flat.grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        red = dp.GradReducer.for_flat(_Flat, ctx, min_bucket=(1, 150, 250, 10 ** 6)[trial])
        assert red.late == (800, 1000) and red.buckets[-1][0] == 0 and red.buckets[0][1] == 800
        red.begin()
        reported = [i for i in range(7, -1, -1) if rnd.random() < (1.0, 0.6, 0.3, 0.0)[(rank + trial) % 4]]
        for i in reported:
            time.sleep(rnd.random() * 0.01)
            red.layer_done(i)
        _Flat.grad[800:] += 1.0                         # the projector's gradient: the last thing a backward produces
        red.finish()
        want = torch.arange(1000, dtype=torch.float32) * mean_rank1
        want[800:] += 1.0
        assert torch.allclose(_Flat.grad, want), trial
    # 4. one rank (5 of 8; the last one in a smaller world) fails in the middle of its backward: drain(), collective MAX, everybody repeats
    bad = 5 if world > 5 else world - 1
    redf = dp.GradReducer.for_flat(_Flat, ctx, min_bucket=1)
    for fail_after in (0, 3, 8, "before_begin"):
        _Flat.grad = torch.full((1000,), rank + 1.0)
        redf.g = _Flat.grad
        failed = rank == bad
        if not (failed and fail_after == "before_begin"):
            redf.begin()
            for k, layer in enumerate(range(7, -1, -1)):
                if failed and fail_after == k:
                    break
                redf.layer_done(layer)
        if failed:
            redf.drain()
        else:
            redf.finish()
        assert dp.max_scalar(1.0 if failed else 0.0, ctx) == 1.0
        _Flat.grad.fill_(rank + 1.0)
        redf.begin()
        for layer in range(7, -1, -1):
            redf.layer_done(layer)
        redf.finish()
        assert torch.allclose(_Flat.grad, torch.full((1000,), mean_rank1)), fail_after
    b = torch.full((5,), float(rank + 7))
    dp.broadcast_(b, ctx)
    assert torch.equal(b, torch.full((5,), 7.0))
    # 5. the trainer's loader: the sampler groups `world x accum` micro-batches by length (reference halva_trainer.py:261-274) and rank r
    #    takes every world-th batch: disjoint, complete, every rank the same number of micro-batches, a multiple of accum
    from llava.train.halva_trainer import HalvaTrainer
    n_samples = 2 * world * accum * 3                   # three optimizer steps of 2-pair micro-batches

    class DS(torch.utils.data.Dataset):
        modality_lengths = [5 + (7 * i) % 41 for i in range(n_samples)]

        def __len__(self):
            return n_samples

        def __getitem__(self, i):
            return i
    t = HalvaTrainer.__new__(HalvaTrainer)
    t.dist, t.train_dataset, t.data_collator = ctx, DS(), (lambda x: x)
    t.args = type("A", (), dict(per_device_train_batch_size=2, gradient_accumulation_steps=accum, group_by_modality_length=True,
                                 dataloader_num_workers=0, dataloader_drop_last=False, seed=42))()
    torch.manual_seed(7)
    batches = [list(b) for b in t.get_train_dataloader()]
    gathered = [None] * world
    dist.all_gather_object(gathered, batches)
    if rank == 0:
        flat_ids = [i for r in gathered for b in r for i in b]
        assert sorted(flat_ids) == list(range(n_samples)), "dealing is not disjoint + complete"
        assert len({len(r) for r in gathered}) == 1 and len(gathered[0]) == 3 * accum
        assert all(len(b) == 2 for r in gathered for b in r)
        out.put("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,accum", [(8, 4), (4, 1)])
def test_eight_rank_gloo(world, accum):
    """world 8 (x accumulation 4: the recipe's `world x accum` sampler dealing) and world 4 on CPU over gloo: plumbing only - no
    scaling curve can come of it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, accum, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"
