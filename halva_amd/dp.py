"""Data parallelism for the DPA step: one process per GPU, full replicas, ONE gradient exchange per optimizer step.

The reference shards everything with DeepSpeed ZeRO-3 over NCCL (src/json/zero3.json; per-module parameter all-gathers on
every forward / recompute / backward).  On MI355X both 7B models fit each GPU many times over (288 GB), so the only
exchange the path needs is the mean of the trainable gradients: the flat fp32 buffer of halva_amd.dpa.FlatTrainables
(LoRA 319.8 M + projector 21.0 M parameters = 1.36 GB) is all-reduced over RCCL/xGMI in a few large buckets
(xGMI is point-to-point: big buckets keep every link of the ring busy and amortise launch latency).
`HalvaTrainer.compute_loss` bypasses any wrapper forward (reference halva_trainer.py:548,573), so DDP-style hooks could
not be used anyway - the explicit all-reduce is the natural design.

The exchange is overlapped with the tail of the step's LAST backward (`GradReducer`): the flat buffer is laid out layer by
layer, the backward finishes the layers from the top down, and the moment layer i's input gradient exists every kernel that
adds into the segments of layers >= i has been enqueued - so the bucket that ends there is handed to RCCL (which runs on its
own stream behind an event of the compute stream) while the layers below are still being differentiated.  Only the last
bucket (bottom layers + projector) is exposed.
"""
import os
import re

import torch
import torch.distributed as dist

BUCKET_ELEMS = 64 * 1024 * 1024        # 256 MB of fp32 per collective


def local_device_index():
    """GPU of this rank: LOCAL_RANK, or LOCAL_RANK modulo the visible devices with HALVA_SHARE_GPU=1 (several ranks on one GPU:
    the 1-GPU test boxes; implies the gloo backend - RCCL cannot put two ranks on one device)."""
    local = max(0, int(os.environ.get("LOCAL_RANK", "0")))
    if os.environ.get("HALVA_SHARE_GPU") == "1":
        return local % max(1, torch.cuda.device_count())
    return local


class DistContext:
    """rank / world / device of this process.  `active` says whether the collectives are really issued: always with world > 1,
    and with world == 1 when HALVA_DP_FORCE=1 asked for a ONE-RANK communicator (RCCL accepts one): every bucket of the gradient
    exchange, the scalar reductions and the barriers then go through torch.distributed on the one GPU a build box has - the code
    an 8-GPU run executes, with sums over a single rank (bit-identical to not exchanging at all)."""

    def __init__(self, rank=0, world=1, local_rank=0, group=None, active=None):
        self.rank, self.world, self.local_rank, self.group = rank, world, local_rank, group
        self.active = (world > 1) if active is None else bool(active)

    @classmethod
    def from_env(cls, backend=None):
        """Read RANK / WORLD_SIZE / LOCAL_RANK (torchrun, `bench.py --gpus N`'s own spawner or the `deepspeed` shim);
        initialise torch.distributed if needed.  backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" for the CPU tests and
        for several ranks sharing one GPU (HALVA_DIST_BACKEND overrides)."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = local_device_index()
        force = os.environ.get("HALVA_DP_FORCE") == "1"
        if (world > 1 or force) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            backend = os.environ.get("HALVA_DIST_BACKEND") or ("gloo" if os.environ.get("HALVA_SHARE_GPU") == "1" else backend)
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        return cls(rank, world, local, active=world > 1 or (force and dist.is_initialized()))


def _is_gloo(ctx):
    return dist.get_backend(ctx.group) == "gloo"


def _allreduce_sum_async(t, ctx):
    """Sum `t` (a contiguous slice of the flat buffer) over the ranks; returns a finisher to call before `t` is read.
    RCCL: asynchronous on the backend's stream, ordered behind the work already enqueued on the current stream.
    gloo (CPU tests / ranks sharing a GPU): device tensors are staged through the host."""
    if t.is_cuda and _is_gloo(ctx):
        host = t.detach().cpu()
        h = dist.all_reduce(host, op=dist.ReduceOp.SUM, group=ctx.group, async_op=True)

        def fin():
            h.wait()
            t.copy_(host)
        return fin
    h = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=ctx.group, async_op=True)
    return h.wait


def broadcast_(t, ctx, src=0):
    """Every rank ends up with rank `src`'s values of `t` (replica initialisation: what DeepSpeed's engine does for the reference at
    start-up - randomly initialised LoRA factors must not differ between the replicas)."""
    if not ctx.active:
        return t
    if t.is_cuda and _is_gloo(ctx):
        host = t.detach().cpu()
        dist.broadcast(host, src=src, group=ctx.group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=ctx.group)
    return t


def allreduce_mean_(flat, ctx):
    """In-place mean over ranks of a flat gradient buffer, bucketed, after the fact (no overlap).  Returns the buffer."""
    if not ctx.active:
        return flat
    fins = [_allreduce_sum_async(flat[lo:lo + BUCKET_ELEMS], ctx) for lo in range(0, flat.numel(), BUCKET_ELEMS)]
    for f in fins:
        f()
    flat.div_(ctx.world)
    return flat


_DECODER_LAYER = re.compile(r"(?:^|\.)layers\.(\d+)\.")


def _is_projector_name(n):
    return "mm_projector" in n


def layer_boundaries(names, offsets):
    """({decoder layer index: first flat element of that layer's parameters}, first element of the tail behind the layers) from
    FlatTrainables' names/offsets ("...model.layers.<i>....").  The tail is the projector: it sits BEHIND the last layer in the
    buffer but its gradient is the LAST thing a backward produces (below layer 0), so it can only be exchanged at the end.
    Only DECODER layers count: VILA's projector tensors are called `mm_projector.layers.{1,2,4}.*` (halva_amd/vila_model.py) and
    must land in the tail, not in a bucket that is handed to RCCL while their gradient does not exist yet.  The layout is
    validated: layer offsets strictly increasing with the layer index, every projector tensor inside the tail."""
    first, tail = {}, None
    for n, o in zip(names, offsets):
        m = None if _is_projector_name(n) else _DECODER_LAYER.search(n)
        if m:
            if tail is not None:
                raise ValueError("flat buffer layout: decoder tensor %r behind the non-layer tail" % n)
            first.setdefault(int(m.group(1)), int(o))
        elif tail is None:
            tail = int(o)
    idx = sorted(first)
    if any(first[a] >= first[b] for a, b in zip(idx, idx[1:])):
        raise ValueError("flat buffer layout: decoder layers are not laid out in increasing order: %r" % (first,))
    for n, o in zip(names, offsets):
        if _is_projector_name(n) and (tail is None or int(o) < tail):
            raise ValueError("flat buffer layout: projector tensor %r is not inside the late tail" % n)
    return first, tail


class GradReducer:
    """Mean of the flat fp32 gradient over the ranks, issued bucket by bucket WHILE the step's last backward is still running.

    begin()            before that backward (nothing of the buffer is final yet);
    ready_from(lo)     "every element at or beyond `lo` will not change any more" - called from the backward (tensor hook on
                       a decoder layer's input, see LlamaModel.grad_ready_hook) with non-increasing `lo`; complete buckets
                       [lo_b, hi_b) with lo_b >= lo are handed to the backend;
    finish()           after the backward returned: the remaining head of the buffer and the late tail, wait for everything,
                       divide by world.
    Buckets are cut at layer boundaries, at least `min_bucket` elements each, walking down from `late_from` (default: the end
    of the buffer); [late_from, end) - the projector - is one more bucket that only finish() may hand over."""

    def __init__(self, flat_grad, ctx, boundaries=(), min_bucket=None, late_from=None):
        self.g, self.ctx = flat_grad, ctx
        min_bucket = BUCKET_ELEMS // 2 if min_bucket is None else min_bucket
        n = flat_grad.numel()
        top = n if late_from is None else int(late_from)
        cuts = sorted({int(b) for b in boundaries if 0 < int(b) < top}, reverse=True)
        self.buckets = []                      # top-down: [(lo, hi)], hi exclusive
        self.late = (top, n) if top < n else None
        hi = top
        for c in cuts:
            if hi - c >= min_bucket:
                self.buckets.append((c, hi))
                hi = c
        if hi > 0:
            self.buckets.append((0, hi))
        self._next = 0
        self._fins = []
        self._open = False                     # between begin() and finish()
        self.issued_early = 0

    @classmethod
    def for_flat(cls, flat, ctx, min_bucket=None):
        """Reducer over a FlatTrainables' gradient buffer with its layer boundaries as bucket cuts."""
        first, tail = layer_boundaries(flat.names, flat.offsets[:-1])
        r = cls(flat.grad, ctx, first.values(), min_bucket, late_from=tail)
        r.first = first
        return r

    def begin(self):
        """Start a step's exchange.  An exchange left open by an attempt that never reached finish() (an exception in the
        backward, e.g. out of memory) is completed first - see drain(): dropping its handles would let this step's buckets pair
        with the stale ones on the other ranks."""
        if self._open:
            self.drain()
        self._next, self._fins, self.issued_early, self._open = 0, [], 0, True

    def drain(self):
        """Complete the exchange of a step that failed on THIS rank before its finish(): hand over every bucket not handed over
        yet (all of them when the step failed before begin()), in finish()'s order, so that each collective pairs with the one
        the healthy ranks issue from their backward / finish(); wait for all of them; leave the buffer's contents undefined (no
        division: the caller zeroes it and repeats the step).  Afterwards every rank has taken part in exactly the collectives
        of one finish(), i.e. the ranks are in step again and can agree on what to do next (bench.py: MAX of an "out of
        memory" flag).  Must not be called for a step whose finish() has returned."""
        if not self._open:
            self._next, self._fins = 0, []
        if self.ctx.active:
            while self._next < len(self.buckets):
                self._issue(*self.buckets[self._next])
                self._next += 1
            if self.late is not None:
                self._issue(*self.late)
        fins, self._fins = self._fins, []
        for f in fins:
            f()
        self._open = False

    def layer_done(self, i):
        """The backward has produced the gradient of decoder layer i's input: layers >= i are final."""
        lo = getattr(self, "first", {}).get(i)
        if lo is not None:
            self.ready_from(lo)

    def _issue(self, lo, hi):
        self._fins.append(_allreduce_sum_async(self.g[lo:hi], self.ctx))

    def ready_from(self, lo):
        if not self.ctx.active:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= lo:
            self._issue(*self.buckets[self._next])
            self._next += 1
            self.issued_early += 1

    def finish(self):
        if not self.ctx.active:
            self._open = False
            return self.g
        if not self._open:                     # a step without begin() (no reducer handed to the backward): exchange everything now
            self._next, self._fins, self.issued_early = 0, [], 0
        while self._next < len(self.buckets):
            self._issue(*self.buckets[self._next])
            self._next += 1
        if self.late is not None:
            self._issue(*self.late)
        for f in self._fins:
            f()
        self._fins = []
        self._open = False
        self.g.div_(self.ctx.world)            # (a forced one-rank communicator: x / 1 is exact)
        return self.g


def _scalar(x, ctx, op):
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(ctx.group) == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=op, group=ctx.group)
    return float(t)


def mean_scalar(x, ctx):
    return float(x) if not ctx.active else _scalar(x, ctx, dist.ReduceOp.SUM) / ctx.world


def max_scalar(x, ctx):
    return float(x) if not ctx.active else _scalar(x, ctx, dist.ReduceOp.MAX)


def shard_batches(n_batches, ctx):
    """Indices of the global micro-batches this rank processes (rank r takes r, r + world, ...)."""
    return list(range(ctx.rank, n_batches, ctx.world))


def barrier(ctx):
    if ctx.active:
        dist.barrier(group=ctx.group)
