"""The data-parallel path of the REAL engine on the GPU: N ranks == one rank over the same global batch.

Reference: one process per GPU (src/hallava_7b.sh:30), each with its own micro-batch from the sampler
(llava/train/halva_trainer.py:261-272), gradients averaged over the ranks once per optimizer step.  Here two rank
processes (started exactly like bench.py / bin/deepspeed start theirs) run halva_amd.dpa.DPAEngine on pairs {0,1} and
{1,2} of the reference-generated fixture with the all-reduce issued from inside the last backward; a single process that
runs the same two micro-batches one after the other with gradient accumulation (scale 1/2) must end up with the same flat
fp32 gradient, the same mean loss and the same weights after one AdamW step.

On a 1-GPU box the two ranks share the device and exchange over gloo (host-staged); with >= 2 GPUs the same test also runs
over RCCL, one rank per GPU."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, out, backend, share_gpu):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HALVA_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if share_gpu:
            env["HALVA_SHARE_GPU"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out], env=env))
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=600))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(-9)
    assert codes == [0] * world, codes


def _single_process_reference(world=2):
    from dp_worker import micro_batch
    from golden_util import load_npz
    from model_util import batch_of, build_product_models
    from halva_amd import dpa
    z = load_npz("dpa_step_d64_init.npz")
    pol, ref, _ = build_product_models(z, device="cuda:0")
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    opt = dpa.AdamWFlat(flat, lr=1e-3, weight_decay=0.0, mm_projector_lr=1e-3)
    eng = dpa.DPAEngine(pol, ref, float(z["alpha"]), pairs_per_group=1, ref_rows_per_group=1)
    flat.zero_grad()
    full = batch_of(z)
    n = full["input_ids"].shape[0]
    losses = [float(eng.loss(micro_batch(full, [r % n, (r + 1) % n]), backward=True, scale=1.0 / world)) for r in range(world)]
    grad = flat.grad.detach().cpu().clone()
    opt.step()
    return grad, flat.master.detach().cpu(), sum(losses) / world


def _check(got, want, world=2):
    grad, master, loss = want
    assert got["world"] == world
    assert abs(got["loss"] - loss) < 1e-6
    # every per-micro-batch gradient is produced by the same deterministic kernels; (a + b) / 2 vs a/2 + b/2 in fp32
    scale = float(grad.abs().max())
    assert scale > 0
    assert float((got["grad"] - grad).abs().max()) <= 2e-6 * scale
    assert float((got["master"] - master).abs().max()) <= 1e-6
    # the exchange really was started from inside the backward: every layer bucket except the last one, and the projector
    assert len(got["buckets"]) >= 3 and got["issued_early"] >= len(got["buckets"]) - 1


def test_two_ranks_sharing_one_gpu_match_one_rank(tmp_path):
    out = str(tmp_path / "dp2.pt")
    _launch(2, out, "gloo", share_gpu=True)
    _check(torch.load(out, weights_only=False), _single_process_reference())


def test_eight_ranks_sharing_one_gpu_match_one_rank(tmp_path):
    """PLUMBING at the recipe's rank count (every multi-process test stopped at world = 2, so the first 8-GPU run would have
    been the first 8-rank run of any kind; reference src/hallava_7b.sh:21-22,30).  Eight rank processes of the real engine on one GPU (gloo,
    host-staged buckets), micro-batch r = pairs (r, r + 1) mod 4 of the fixture: flat fp32 gradient, mean loss and the weights after AdamW equal
    one process accumulating the same eight micro-batches (scale 1/8).  No scaling curve can come of it."""
    out = str(tmp_path / "dp8.pt")
    _launch(8, out, "gloo", share_gpu=True)
    _check(torch.load(out, weights_only=False), _single_process_reference(8), world=8)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL)")
def test_two_ranks_over_rccl_match_one_rank(tmp_path):
    out = str(tmp_path / "dp2_rccl.pt")
    _launch(2, out, "nccl", share_gpu=False)
    got = torch.load(out, weights_only=False)
    assert got["backend"] == "nccl"
    _check(got, _single_process_reference())


def test_one_rank_rccl_communicator_runs_the_exchange_and_changes_nothing(tmp_path):
    """HALVA_DP_FORCE=1 + WORLD_SIZE=1 + backend nccl: a ONE-rank RCCL communicator.  The code an 8-GPU run executes
    (dist.all_reduce(async_op=True) on slices of the flat device buffer, issued from the backward hooks on RCCL's stream behind
    the compute stream; wait; divide; the nccl branch of the scalar reductions; the barrier) runs on the one GPU of this box.
    A sum over one rank is the identity: flat gradient, weights after AdamW and loss must equal the reducer-less step's BIT FOR
    BIT, with every bucket but the last issued inside the backward."""
    from golden_util import load_npz
    from model_util import batch_of, build_product_models
    from dp_worker import micro_batch
    from halva_amd import dpa
    out = str(tmp_path / "dp1_rccl.pt")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HALVA_DIST_BACKEND="nccl", HALVA_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("HALVA_SHARE_GPU", None)
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out], env=env, timeout=600).returncode
    assert rc == 0
    got = torch.load(out, weights_only=False)
    assert got["backend"] == "nccl" and got["world"] == 1
    assert len(got["buckets"]) >= 3 and got["issued_early"] >= len(got["buckets"]) - 1
    z = load_npz("dpa_step_d64_init.npz")
    pol, ref, _ = build_product_models(z, device="cuda:0")
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    opt = dpa.AdamWFlat(flat, lr=1e-3, weight_decay=0.0, mm_projector_lr=1e-3)
    eng = dpa.DPAEngine(pol, ref, float(z["alpha"]), pairs_per_group=1, ref_rows_per_group=1)
    flat.zero_grad()
    loss = float(eng.loss(micro_batch(batch_of(z), [0, 1]), backward=True))
    assert torch.equal(got["grad"], flat.grad.detach().cpu())
    opt.step()
    assert torch.equal(got["master"], flat.master.detach().cpu())
    assert got["loss"] == loss


def test_bench_json_line_is_the_last_line_of_stdout_with_an_rccl_communicator():
    """The driver reads ONE JSON line from rank 0.  RCCL prints a version banner to stdout through C stdio when its communicator is created; into a
    pipe that buffer used to be flushed at process exit - BEHIND the JSON line (round 6: profiles/r06_bench_dp1_rccl.json's raw output).  bench.py
    flushes it right after the communicator exists: with a (forced one-rank) RCCL communicator the last line of stdout is the JSON record."""
    import json
    env = dict(os.environ, HALVA_DP_FORCE="1", HALVA_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HALVA_SHARE_GPU", "HALVA_BENCH_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--layers", "2", "--pairs-per-gpu", "2",
                        "--pairs-per-group", "2", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    rec = json.loads(lines[-1])                      # (raises if anything trails the record)
    assert rec["grad_allreduce"]["backend"] == "nccl" and rec["grad_allreduce"]["forced_one_rank_communicator"]
    assert sum(1 for l in lines if l.startswith("{")) == 1


def test_bench_under_the_drivers_launcher_form():
    """The driver's N > 1 command, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N --steps K --warmup W` - the ranks exist already (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher), nothing is
    spawned, rank 0 prints the ONE JSON line.  Two ranks; on a 1-GPU box they share the device over gloo (HALVA_SHARE_GPU / HALVA_DIST_BACKEND -
    what the spawner sets for HALVA_BENCH_SHARE_GPU); with 2 GPUs visible: RCCL.  Every other N > 1 test goes through bench.py's own spawner."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env.update(HALVA_SHARE_GPU="1", HALVA_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--layers", "2", "--pairs-per-gpu", "2", "--pairs-per-group", "2", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_pairs"] == 4 and rec["value"] > 0 and rec["grad_allreduce"]["world"] == 2


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no launcher): the parent starts two fresh rank processes before touching the GPU and rank 0
    prints the JSON line.  Two layers of the 7B geometry; on a 1-GPU box the ranks share the device (HALVA_BENCH_SHARE_GPU)."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["HALVA_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--layers", "2",
                        "--pairs-per-gpu", "2", "--pairs-per-group", "2", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_pairs"] == 4 and rec["value"] > 0
    assert rec["grad_allreduce"]["buckets_issued_inside_backward"] >= 1


def test_bench_eight_ranks_and_a_collective_out_of_memory_fallback(tmp_path):
    """`python bench.py --gpus 8 --global-pairs 8` (1 pair per rank, two layers, the ranks sharing the GPU of a 1-GPU box): one JSON line, n_gpus 8,
    global 8 - the spawner's port handling, `--global-pairs` / 8, eight reducers.  Then 4 pairs per rank with rank 5 capped below what groups of 4
    and of 2 need: all eight ranks must fall back together, twice.  Plumbing only: the ranks share one GPU, the line's value means nothing."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 8:
        env["HALVA_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--global-pairs", "8", "--steps", "1", "--warmup", "1", "--layers", "2",
                        "--pairs-per-group", "1", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["config"]["global_pairs"] == 8 and rec["config"]["pairs_per_gpu"] == 1 and rec["value"] > 0
    assert rec["grad_allreduce"]["world"] == 8 and rec["grad_allreduce"]["buckets_issued_inside_backward"] >= 1
    total = torch.cuda.get_device_properties(0).total_memory
    # the 2-rank test's memory ladder (4 pairs per rank, two layers: groups of 4 pairs 11.2 / 12.2 GiB at the peak, of 2 pairs 8.5 / 9.9, of 1 pair
    # 6.2 / 6.7): rank 5 alone capped at 7.6 GiB -> two collective fall-backs of all eight ranks
    env["HALVA_BENCH_MEM_FRACTION"] = "5:%.5f" % (7.6 * 2 ** 30 / total)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--global-pairs", "32", "--steps", "1", "--warmup", "1", "--layers", "2",
                        "--pairs-per-group", "4", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["config"]["oom_fallbacks_in_warmup"] == 2 and rec["config"]["pairs_per_group"] == 1 and rec["n_gpus"] == 8, (rec["config"], r.stderr[-2000:])


def test_bench_out_of_memory_on_one_rank_is_a_collective_decision(tmp_path):
    """Rank 1 alone gets too small a share of the HBM for the requested group size (HALVA_BENCH_MEM_FRACTION=1:<f>) and runs out of
    memory inside the warm-up step, after rank 0 (and possibly itself) has handed gradient buckets to the backend.  The fall-back must be
    collective: rank 1 completes the exchange (GradReducer.drain), both ranks see the MAX of the flag, both halve their groups and the
    run ends with one valid JSON line - no hang, no mismatched collectives.  Reference: one process per GPU, gradients averaged once per
    optimizer step (src/hallava_7b.sh:30, llava/train/halva_trainer.py:261-272)."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["HALVA_BENCH_SHARE_GPU"] = "1"
    total = torch.cuda.get_device_properties(0).total_memory
    # two layers of the 7B geometry, 4 pairs per rank (measured on MI355X, allocated / reserved GiB at the peak): groups of 4 pairs
    # 11.2 / 12.2, groups of 2 pairs 8.5 / 9.9, groups of 1 pair 6.2 / 6.7.  A 7.6 GiB cap fails the warm-up step itself with groups
    # of 4 and of 2 (a cap between two settings can pass the warm-up and fail a later, slightly larger step - nothing the warm-up
    # fall-back could catch) and leaves groups of 1 real headroom: two collective fall-backs.
    env["HALVA_BENCH_MEM_FRACTION"] = "1:%.5f" % (7.6 * 2 ** 30 / total)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--layers", "2",
                        "--pairs-per-gpu", "4", "--pairs-per-group", "4", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["config"]["oom_fallbacks_in_warmup"] == 2, (rec["config"], r.stderr[-2000:])
    assert rec["config"]["pairs_per_group"] == 1 and rec["n_gpus"] == 2 and rec["value"] > 0
    assert "another rank" in r.stderr or "this rank" in r.stderr


def test_deepspeed_shim_two_ranks_train_like_one_rank_with_accumulation(tmp_path):
    """`deepspeed --num_gpus 2 train_halva.py ...` (the reference's launch line, src/hallava_7b.sh:30, through bin/deepspeed): two rank
    processes, each taking every second batch of the sampler (llava/train/halva_trainer.py:261-272), gradients averaged inside the
    last backward of every step.  One rank with --gradient_accumulation_steps 2 consumes the same batches per optimizer step (the
    reference's sampler is built for world_size * accumulation = 2 either way), so both runs must end with the same adapter."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_util
    paths = e2e_util.build(str(tmp_path))
    runner = os.path.join(str(tmp_path), "run_train.py")
    with open(runner, "w") as f:
        f.write("import sys\nsys.path[:0] = [%r, %r, %r]\n"
                "import e2e_util, pytest\nfrom _pytest.monkeypatch import MonkeyPatch\n"
                "paths = dict(vocab_size=%d, data=%r, ref=%r, images=%r, vision=%r)\n"
                "e2e_util.patch_tokenizer(MonkeyPatch(), paths['vocab_size'], warm_paths=paths)\n"
                "import llava.train.train_halva as TH\nTH.train(sys.argv[1:])\n"
                % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), paths["vocab_size"], paths["data"], paths["ref"],
                   paths["images"], paths["vision"]))

    def argv(out, accum):
        return ("--lora_enable True --lora_r 8 --lora_alpha 16 --mm_projector_lr 0 --loss_alpha 0.4 --model_name_or_path %s --version v1 "
                "--data_path %s --ref_data_path %s --image_folder %s --vision_tower %s --mm_projector_type mlp2x_gelu "
                "--mm_vision_select_layer -2 --mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio pad "
                "--group_by_modality_length True --bf16 True --output_dir %s --num_train_epochs 2 --per_device_train_batch_size 1 "
                "--gradient_accumulation_steps %d --learning_rate 1e-3 --warmup_ratio 0.03 --lr_scheduler_type cosine --logging_steps 1 "
                "--save_strategy no --model_max_length 64" % (paths["ckpt"], paths["data"], paths["ref"], paths["images"], paths["vision"], out,
                                                               accum)).split()
    env = dict(os.environ, HALVA_SHARE_GPU="1" if torch.cuda.device_count() < 2 else "0", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out2, out1 = os.path.join(str(tmp_path), "dp2"), os.path.join(str(tmp_path), "dp1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "deepspeed"), "--num_gpus", "2", runner] + argv(out2, 1), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "deepspeed"), "--num_gpus", "1", runner] + argv(out1, 2), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    s2, s1 = (json.load(open(os.path.join(o, "trainer_state.json"))) for o in (out2, out1))
    assert s2["global_step"] == s1["global_step"] == 6                       # 6 samples / (1 per rank x 2 ranks) x 2 epochs
    l2, l1 = ([x["loss"] for x in st["log_history"] if "loss" in x] for st in (s2, s1))
    assert max(abs(a - b) for a, b in zip(l2, l1)) < 2e-3, (l2, l1)         # mean over ranks == mean over the accumulated micro-batches
    a2, a1 = (torch.load(os.path.join(o, "adapter_model.bin")) for o in (out2, out1))
    assert set(a2) == set(a1)
    for k in a1:
        assert float((a2[k].float() - a1[k].float()).norm()) <= 2e-2 * float(a1[k].float().norm()) + 1e-6, k


def test_deepspeed_shim_eight_ranks_train_like_one_rank_with_accumulation(tmp_path):
    """`deepspeed --num_gpus 8 train_halva.py` with the flag set of src/hallava_7b.sh:30-69 against ONE rank with --gradient_accumulation_steps 8:
    the sampler is built for world x accumulation = 8 either way (llava/train/halva_trainer.py:261-274), so both consume the same 8 micro-batches
    per optimizer step - same step count, losses, adapter.  Eight samples, one per rank and step; the ranks share the GPU of a 1-GPU box (gloo)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_util
    paths = e2e_util.build(str(tmp_path), n_samples=8)
    runner = os.path.join(str(tmp_path), "run_train.py")
    with open(runner, "w") as f:
        f.write("import sys\nsys.path[:0] = [%r, %r, %r]\n"
                "import e2e_util, pytest\nfrom _pytest.monkeypatch import MonkeyPatch\n"
                "paths = dict(vocab_size=%d, data=%r, ref=%r, images=%r, vision=%r)\n"
                "e2e_util.patch_tokenizer(MonkeyPatch(), paths['vocab_size'], warm_paths=paths)\n"
                "import llava.train.train_halva as TH\nTH.train(sys.argv[1:])\n"
                % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), paths["vocab_size"], paths["data"], paths["ref"],
                   paths["images"], paths["vision"]))

    def argv(out, accum):      # every flag of the reference's launch line (src/hallava_7b.sh:30-69) that the run's size allows
        return ("--lora_enable True --lora_r 8 --lora_alpha 16 --mm_projector_lr 0 --deepspeed %s --loss_alpha 0.4 --model_name_or_path %s --version v1 "
                "--data_path %s --ref_data_path %s --image_folder %s --vision_tower %s --mm_projector_type mlp2x_gelu "
                "--mm_vision_select_layer -2 --mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio pad "
                "--group_by_modality_length True --bf16 True --output_dir %s --num_train_epochs 2 --per_device_train_batch_size 1 "
                "--per_device_eval_batch_size 4 --gradient_accumulation_steps %d --evaluation_strategy no --save_strategy no --save_steps 50000 "
                "--save_total_limit 1 --learning_rate 1e-3 --weight_decay 0. --warmup_ratio 0.03 --lr_scheduler_type cosine --logging_steps 1 "
                "--tf32 True --model_max_length 64 --gradient_checkpointing True --dataloader_num_workers 0 --lazy_preprocess True "
                "--report_to none --run_name dp8" % (os.path.join(ROOT, "src", "json", "zero3.json"), paths["ckpt"], paths["data"], paths["ref"],
                                                     paths["images"], paths["vision"], out, accum)).split()
    env = dict(os.environ, HALVA_SHARE_GPU="1" if torch.cuda.device_count() < 8 else "0", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out8, out1 = os.path.join(str(tmp_path), "dp8"), os.path.join(str(tmp_path), "dp1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "deepspeed"), "--num_gpus", "8", runner] + argv(out8, 1), env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "deepspeed"), "--num_gpus", "1", runner] + argv(out1, 8), env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    s8, s1 = (json.load(open(os.path.join(o, "trainer_state.json"))) for o in (out8, out1))
    assert s8["global_step"] == s1["global_step"] == 2                       # 8 samples / (1 per rank x 8 ranks) x 2 epochs
    l8, l1 = ([x["loss"] for x in st["log_history"] if "loss" in x] for st in (s8, s1))
    assert max(abs(a - b) for a, b in zip(l8, l1)) < 2e-3, (l8, l1)         # mean over ranks == mean over the accumulated micro-batches
    a8, a1 = (torch.load(os.path.join(o, "adapter_model.bin")) for o in (out8, out1))
    assert set(a8) == set(a1)
    for k in a1:
        assert float((a8[k].float() - a1[k].float()).norm()) <= 2e-2 * float(a1[k].float().norm()) + 1e-6, k
