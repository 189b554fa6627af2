"""Host logic of HalvaTrainer.train() on the CPU (the engine is replaced by a deterministic stand-in; everything else - sampler,
per-rank batch dealing, accumulation boundaries, step count, cosine schedule, checkpoint files, rotation, resume - is the product
code).  Semantics follow HF Trainer 4.31 as the reference inherits them (reference llava/train/halva_trainer.py:155-156,
train_halva.py:1222-1225)."""
import json
import math
import os
import shutil
import types

import pytest
import torch

from halva_amd import dp, dpa
from llava.train.halva_trainer import HalvaTrainer


class _DS(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n
        self.modality_lengths = [20 + (7 * i) % 13 for i in range(n)]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class _Trainer(HalvaTrainer):
    """The real loop around a fake model: `training_step` adds a gradient that depends on the batch and on the weights."""

    def __init__(self, args, n):
        self.args = args
        self.model = types.SimpleNamespace(config=types.SimpleNamespace(save_pretrained=lambda d: open(os.path.join(d, "config.json"), "w").write("{}")))
        self.train_dataset, self.data_collator = _DS(n), (lambda x: torch.tensor(x))
        self.callbacks, self.optimizer = [], None
        from llava.train.halva_trainer import TrainerState
        self.state = TrainerState()
        self.dist = dp.DistContext()
        self._engine = self._flat = None
        self.seen = []

    def create_optimizer(self):
        torch.manual_seed(3)
        ps = [("model.layers.0.self_attn.q.lora_A_cat", torch.nn.Parameter(torch.randn(4, 4).bfloat16())),
              ("model.layers.1.mlp.d.lora_B.default.weight", torch.nn.Parameter(torch.randn(4, 2).bfloat16())),
              ("model.mm_projector.0.weight", torch.nn.Parameter(torch.randn(3, 3).bfloat16()))]
        self._flat = dpa.FlatTrainables(ps)
        a = self.args
        self.optimizer = dpa.AdamWFlat(self._flat, lr=a.learning_rate, weight_decay=0.0, mm_projector_lr=a.learning_rate)
        return self.optimizer

    def training_step(self, inputs, scale=1.0, reducer=None):
        ids = inputs.float()
        self.seen.append(inputs.tolist())
        w = self._flat.flat.float()
        loss = float((w * w).sum() * 0.01 + ids.sum() * 1e-3)
        self._flat.grad += scale * (0.02 * w + torch.sin(torch.arange(w.numel()) * float(ids.sum())))
        return torch.tensor(loss)

    def _to_device(self, batch):
        return batch

    def _save_adapter(self, folder):
        torch.save({"flat": self._flat.flat.clone()}, os.path.join(folder, "adapter_model.bin"))


def _args(out, **kw):
    d = dict(output_dir=out, per_device_train_batch_size=2, gradient_accumulation_steps=1, group_by_modality_length=True,
             dataloader_num_workers=0, dataloader_drop_last=False, seed=42, num_train_epochs=2, max_steps=-1, learning_rate=1e-2,
             warmup_ratio=0.1, logging_steps=1, save_strategy="steps", save_steps=2, save_total_limit=None)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_step_accounting_follows_hf_4_31(tmp_path):
    # 7 micro-batches per epoch (14 samples / bs 2), accumulation 3: floor(7/3) = 2 steps per epoch -> ceil(2 * 2) = 4 steps; the
    # micro-batch counter runs across epochs (total_batched_samples), so steps close after micro-batches 3, 6 | 9, 12 overall
    t = _Trainer(_args(str(tmp_path / "a"), gradient_accumulation_steps=3, save_strategy="no"), 14)
    st = t.train()
    assert st.global_step == 4 and len(t.seen) == 12
    lrs = [r["learning_rate"] for r in st.log_history]
    want = [1e-2 * dpa.cosine_with_warmup(s, 4, 0.1) for s in range(4)]
    assert lrs == pytest.approx(want)
    # --max_steps wins over epochs and stops inside an epoch; the epoch loop is left as well
    t = _Trainer(_args(str(tmp_path / "b"), max_steps=3, num_train_epochs=5, save_strategy="no"), 14)
    assert t.train().global_step == 3 and len(t.seen) == 3
    # an epoch shorter than one accumulation group still steps once per epoch (HF: steps_in_epoch <= accumulation)
    t = _Trainer(_args(str(tmp_path / "c"), gradient_accumulation_steps=8, save_strategy="no"), 6)
    assert t.train().global_step == 2
    # past the end of the schedule the learning rate stays at its floor instead of rising again
    assert dpa.cosine_with_warmup(9, 4, 0.1) == 0.0


def test_checkpoints_rotation_and_resume(tmp_path):
    out_a = str(tmp_path / "run_a")
    ta = _Trainer(_args(out_a), 10)                        # 5 micro-batches x 2 epochs, accumulation 1 -> 10 steps
    sa = ta.train()
    assert sa.global_step == 10
    assert sorted(os.listdir(out_a)) == ["checkpoint-%d" % s for s in (10, 2, 4, 6, 8)]
    for f in ("adapter_model.bin", "config.json", "halva_state.pt", "trainer_state.json"):
        assert os.path.exists(os.path.join(out_a, "checkpoint-6", f))
    st = torch.load(os.path.join(out_a, "checkpoint-6", "halva_state.pt"), weights_only=False)
    assert (st["global_step"], st["epoch_index"], st["micro_in_epoch"], st["micro_total"]) == (6, 1, 1, 6)
    # interrupted after step 6 (inside epoch 2): resume=True picks the newest checkpoint, replays the sampler of that epoch from the
    # saved RNG state, skips the consumed micro-batch and ends with exactly the uninterrupted run's weights and log
    out_b = str(tmp_path / "run_b")
    os.makedirs(out_b)
    for s in (4, 6):
        shutil.copytree(os.path.join(out_a, "checkpoint-%d" % s), os.path.join(out_b, "checkpoint-%d" % s))
    tb = _Trainer(_args(out_b, save_total_limit=2), 10)
    sb = tb.train(resume_from_checkpoint=True)
    assert sb.global_step == 10
    assert tb.seen == ta.seen[6:]                                            # the same micro-batches in the same order
    assert torch.equal(tb._flat.master, ta._flat.master)
    assert [r for r in sb.log_history if "loss" in r and r["step"] > 6] == [
        dict(r, elapsed_s=q["elapsed_s"]) for r, q in zip([r for r in sa.log_history if r["step"] > 6],
                                                          [r for r in sb.log_history if "loss" in r and r["step"] > 6])]
    assert sorted(os.listdir(out_b)) == ["checkpoint-10", "checkpoint-8"]    # --save_total_limit 2
    # resuming from an explicit folder, at an epoch boundary (checkpoint written after the last micro-batch of epoch 1)
    out_c = str(tmp_path / "run_c")
    tc0 = _Trainer(_args(out_c, save_steps=5), 10)
    tc0.train()
    tc = _Trainer(_args(str(tmp_path / "run_c2"), save_strategy="no"), 10)
    sc = tc.train(resume_from_checkpoint=os.path.join(out_c, "checkpoint-5"))
    assert sc.global_step == 10 and tc.seen == ta.seen[5:] and torch.equal(tc._flat.master, ta._flat.master)
    # nothing to resume from: loud failure, not a silent restart
    with pytest.raises(ValueError):
        _Trainer(_args(str(tmp_path / "empty")), 10).train(resume_from_checkpoint=True)
    # a checkpoint of another world size is refused (the per-rank batch order would differ)
    st["world"] = 4
    torch.save(st, os.path.join(out_a, "checkpoint-6", "halva_state.pt"))
    with pytest.raises(RuntimeError):
        _Trainer(_args(str(tmp_path / "w")), 10).train(resume_from_checkpoint=os.path.join(out_a, "checkpoint-6"))


def test_epoch_checkpoint_inside_an_accumulation_group_resumes_bit_for_bit(tmp_path):
    """14 samples / bs 2 = 7 micro-batches per epoch, accumulation 3, 3 epochs: micro-batches are counted ACROSS epochs (HF 4.31), so
    the epoch-1 checkpoint is written with ONE micro-batch of the next optimizer step already in the fp32 accumulator (7 % 3 = 1),
    the epoch-2 checkpoint with two (14 % 3 = 2).  The checkpoint carries the accumulator (round-2 advisor finding: it did not, and the
    resumed step ran on fewer micro-batches at the full 1/accum scale); a resumed run ends with exactly the uninterrupted run's weights."""
    out_a = str(tmp_path / "a")
    ta = _Trainer(_args(out_a, gradient_accumulation_steps=3, num_train_epochs=3, save_strategy="epoch"), 14)
    sa = ta.train()
    ckpts = sorted(os.listdir(out_a), key=lambda d: int(d.split("-")[1]))
    assert ckpts == ["checkpoint-2", "checkpoint-4"]       # (the run ends inside epoch 3, at its 6th step: no third epoch checkpoint)
    st = torch.load(os.path.join(out_a, "checkpoint-2", "halva_state.pt"), weights_only=False)
    assert (st["global_step"], st["epoch_index"], st["micro_in_epoch"], st["micro_total"]) == (2, 1, 0, 7)
    pend = torch.load(os.path.join(out_a, "checkpoint-2", "halva_pending_grad_rank0.pt"), weights_only=False)
    assert pend["pending_micro"] == 1 and float(pend["grad"].abs().sum()) > 0
    assert torch.load(os.path.join(out_a, "checkpoint-4", "halva_pending_grad_rank0.pt"), weights_only=False)["pending_micro"] == 2
    for k, seen_from in ((2, 7), (4, 14)):
        tb = _Trainer(_args(str(tmp_path / ("b%d" % k)), gradient_accumulation_steps=3, num_train_epochs=3, save_strategy="no"), 14)
        sb = tb.train(resume_from_checkpoint=os.path.join(out_a, "checkpoint-%d" % k))
        assert sb.global_step == sa.global_step == 6
        assert tb.seen == ta.seen[seen_from:]
        assert torch.equal(tb._flat.master, ta._flat.master), k
    # the accumulator file is part of the checkpoint: without it the resume fails loudly instead of stepping on a partial sum
    os.remove(os.path.join(out_a, "checkpoint-2", "halva_pending_grad_rank0.pt"))
    with pytest.raises(FileNotFoundError):
        _Trainer(_args(str(tmp_path / "c"), gradient_accumulation_steps=3, num_train_epochs=3, save_strategy="no"), 14).train(
            resume_from_checkpoint=os.path.join(out_a, "checkpoint-2"))
