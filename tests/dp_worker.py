"""One data-parallel rank of the REAL DPA engine on a tiny model (helper process of tests/test_dp_engine_gpu.py).

Started once per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, like `bench.py --gpus N` and the
`deepspeed` shim start their ranks.  Every rank builds the same policy / reference pair from the committed fixture, runs the
engine on ITS micro-batch (pairs rank, rank+1 of the fixture batch) with the gradient exchange started inside the last
backward (halva_amd.dp.GradReducer), then one AdamW step; rank 0 stores what the test compares with a 1-process run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def micro_batch(batch, rows):
    return {k: v[rows] for k, v in batch.items()}


def run(out_path):
    from golden_util import load_npz
    from model_util import batch_of, build_product_models
    from halva_amd import dp, dpa, hip
    hip.load()
    ctx = dp.DistContext.from_env()
    dev = torch.device("cuda", ctx.local_rank)
    torch.cuda.set_device(dev)
    z = load_npz("dpa_step_d64_init.npz")
    pol, ref, _ = build_product_models(z, device=dev)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    opt = dpa.AdamWFlat(flat, lr=1e-3, weight_decay=0.0, mm_projector_lr=1e-3)
    eng = dpa.DPAEngine(pol, ref, float(z["alpha"]), pairs_per_group=1, ref_rows_per_group=1)
    reducer = dp.GradReducer.for_flat(flat, ctx, min_bucket=1)          # one bucket per decoder layer + the projector
    full = batch_of(z)
    n = full["input_ids"].shape[0]
    batch = micro_batch(full, [ctx.rank % n, (ctx.rank + 1) % n])      # (pairs rank, rank + 1 of the fixture batch, wrapping: 8 ranks on 4 pairs)
    flat.zero_grad()
    loss = float(eng.loss(batch, backward=True, reducer=reducer))
    early = reducer.issued_early
    reducer.finish()
    grad = flat.grad.detach().cpu().clone()
    opt.step()
    mean_loss = dp.mean_scalar(loss, ctx)
    if ctx.rank == 0:
        torch.save({"grad": grad, "master": flat.master.detach().cpu(), "loss": mean_loss, "issued_early": early,
                    "buckets": reducer.buckets + [reducer.late], "backend": torch.distributed.get_backend(), "world": ctx.world}, out_path)
    dp.barrier(ctx)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
