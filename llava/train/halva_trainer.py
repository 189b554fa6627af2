"""HalvaTrainer - the reference's trainer surface (llava/train/halva_trainer.py:155-592) on the MI355X-native engine.

What stays: class / method names and signatures, batch-dict keys, loss semantics (every quirk listed in SURVEY.md 8a),
the length-grouped sampler, optimizer parameter groups, LR schedule and the output artefacts.
What changes underneath: no HF Trainer / accelerate / DeepSpeed.  One process per GPU holds full bf16 replicas of policy
and reference model; `training_step` runs the fused group-wise engine (halva_amd/dpa.py); trainable gradients live in
one flat fp32 buffer that is all-reduced over RCCL once per optimizer step (halva_amd/dp.py).
`compute_loss` / `concatenated_forward` / `reference_forward` / `cal_batch_logp` / `accumulate_logps` keep the
reference's tensor-level contracts (full logits included) for callers that use them directly.
"""
import json
import math
import os
import time
from collections import defaultdict
from typing import Any, Dict, List, Optional

import torch
from torch.utils.data import DataLoader, Sampler

from halva_amd import dp, dpa
from halva_amd import kernels as K

IGNORE_INDEX = -100


# ------------------------------------------------------------------------------------------------
# sampler (reference halva_trainer.py:60-152)
# ------------------------------------------------------------------------------------------------
def split_to_even_chunks(indices, lengths, num_chunks):
    """Greedy balance: each index goes to the chunk with the smallest summed length that still has room;
    a megabatch that does not divide evenly is dealt out round-robin instead."""
    if len(indices) % num_chunks != 0:
        return [indices[i::num_chunks] for i in range(num_chunks)]
    room = len(indices) // num_chunks
    buckets = [[] for _ in range(num_chunks)]
    weight = [0] * num_chunks
    for idx in indices:
        k = weight.index(min(weight))
        buckets[k].append(idx)
        weight[k] += lengths[idx]
        if len(buckets[k]) == room:
            weight[k] = float("inf")
    return buckets


def get_length_grouped_indices(lengths, batch_size, world_size, generator=None, merge=True):
    order = torch.randperm(len(lengths), generator=generator).tolist()     # torch RNG: part of the contract
    span = world_size * batch_size
    out = []
    for lo in range(0, len(lengths), span):
        mega = sorted(order[lo:lo + span], key=lambda i: lengths[i], reverse=True)
        for chunk in split_to_even_chunks(mega, lengths, world_size):
            out.extend(chunk)
    return out


def get_modality_length_grouped_indices(lengths, batch_size, world_size, generator=None):
    assert all(l != 0 for l in lengths), "Should not have zero length."
    if all(l > 0 for l in lengths) or all(l < 0 for l in lengths):
        return get_length_grouped_indices(lengths, batch_size, world_size, generator=generator)
    mm = [(i, l) for i, l in enumerate(lengths) if l > 0]
    lang = [(i, -l) for i, l in enumerate(lengths) if l < 0]
    mm_order = [mm[i][0] for i in get_length_grouped_indices([l for _, l in mm], batch_size, world_size, generator=None)]
    lang_order = [lang[i][0] for i in get_length_grouped_indices([l for _, l in lang], batch_size, world_size, generator=None)]
    span = world_size * batch_size
    mm_mega = [mm_order[i:i + span] for i in range(0, len(mm_order), span)]
    lang_mega = [lang_order[i:i + span] for i in range(0, len(lang_order), span)]
    leftovers = mm_mega[-1] + lang_mega[-1]
    full = mm_mega[:-1] + lang_mega[:-1]
    perm = torch.randperm(len(full), generator=generator).tolist()
    full = [full[i] for i in perm]
    if leftovers:
        full.append(sorted(leftovers))
    return [i for mega in full for i in mega]


class LengthGroupedSampler(Sampler):
    """Groups samples of similar length into the same megabatch while keeping randomness."""

    def __init__(self, batch_size: int, world_size: int, lengths: Optional[List[int]] = None, generator=None,
                 group_by_modality: bool = False):
        if lengths is None:
            raise ValueError("Lengths must be provided.")
        self.batch_size, self.world_size, self.lengths = batch_size, world_size, lengths
        self.generator, self.group_by_modality = generator, group_by_modality

    def __len__(self):
        return len(self.lengths)

    def __iter__(self):
        fn = get_modality_length_grouped_indices if self.group_by_modality else get_length_grouped_indices
        return iter(fn(self.lengths, self.batch_size, self.world_size, generator=self.generator))


def disable_dropout_in_model(model: torch.nn.Module) -> None:
    for module in model.modules():
        if isinstance(module, torch.nn.Dropout):
            module.p = 0


# ------------------------------------------------------------------------------------------------
class TrainerState:
    def __init__(self):
        self.global_step = 0
        self.epoch = 0.0
        self.log_history = []


class HalvaTrainer:
    def __init__(self, model=None, ref_model=None, args=None, data_collator=None, label_pad_token_id: int = -100,
                 padding_value: int = 0, is_encoder_decoder: bool = False, loss_alpha: Optional[float] = 0.1,
                 train_dataset=None, eval_dataset=None, tokenizer=None, model_init=None, callbacks=None,
                 optimizers=(None, None), preprocess_logits_for_metrics=None, disable_dropout: bool = True,
                 compute_metrics=None):
        self.model, self.ref_model, self.args = model, ref_model, args
        if ref_model is None:
            raise ValueError("HalvaTrainer needs a frozen ref_model (adapter-disabling is not implemented on this path)")
        if disable_dropout:
            disable_dropout_in_model(model)
            disable_dropout_in_model(ref_model)
        self.loss_alpha = loss_alpha
        self.label_pad_token_id = label_pad_token_id
        self.padding_value = padding_value
        self.is_encoder_decoder = is_encoder_decoder
        self.loss_holder = defaultdict(list)
        self.data_collator, self.train_dataset, self.eval_dataset, self.tokenizer = data_collator, train_dataset, eval_dataset, tokenizer
        self.callbacks = list(callbacks or [])
        self.optimizer, self.lr_scheduler = optimizers
        self.state = TrainerState()
        self.ref_model.eval()
        for p in self.ref_model.parameters():
            p.requires_grad_(False)
        self._engine = None
        self._flat = None
        self.dist = dp.DistContext.from_env()

    # -- reference tensor-level API ---------------------------------------------------------------
    def cal_batch_logp(self, logits: torch.FloatTensor, labels: torch.LongTensor) -> torch.FloatTensor:
        """[S, T, V] logits, [S, T] labels -> [S, T-1] log p(label[t+1] | ..t); IGNORE_INDEX scored as token 0."""
        if logits.shape[:-1] != labels.shape:
            raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape.")
        if not self.is_encoder_decoder:
            labels = labels[:, 1:]
            logits = logits[:, :-1, :]
        tgt = labels.masked_fill(labels == self.label_pad_token_id, 0)
        S, T1, V = logits.shape
        lp = K.token_logp(logits.reshape(S * T1, V).contiguous(), tgt.reshape(-1).to(torch.int32).contiguous())
        return lp.view(S, T1)

    def accumulate_logps(self, logps, signs):
        """Per-phrase sums: column i collects the tokens whose sign equals the (i+1)-th smallest id present anywhere in
        `signs` (batch-global slots; rows lacking a phrase keep 0)."""
        slots = torch.unique(signs, sorted=True)[1:]
        all_valid = torch.zeros_like(signs)
        return K.phrase_sum(logps.float().contiguous(), all_valid, signs.contiguous(), slots.contiguous())

    def concatenated_forward(self, model, inputs):
        ids, neg = inputs["input_ids"], inputs["neg_input_ids"]
        B = ids.shape[0]
        width = max(ids.shape[1], neg.shape[1])
        dev = ids.device

        def stack(pos, negt, fill, dtype):
            out = torch.full((2 * B, width), fill, dtype=dtype, device=dev)
            out[:B, :pos.shape[1]] = pos
            out[B:, :negt.shape[1]] = negt
            return out

        cat_ids = stack(ids, neg, 0, ids.dtype)
        cat_labels = stack(inputs["labels"], inputs["neg_labels"], IGNORE_INDEX, inputs["labels"].dtype)
        cat_mask = stack(inputs["attention_mask"], inputs["neg_attention_mask"], False, torch.bool)
        cat_signs = stack(inputs["pos_signs"], inputs["neg_signs"], 0, inputs["pos_signs"].dtype)
        images = inputs["images"]
        (_, _, mask, _, embeds, labels, signs) = model.prepare_inputs_labels_for_multimodal_signed(
            input_ids=cat_ids, position_ids=None, attention_mask=cat_mask, past_key_values=None, labels=cat_labels,
            images=torch.cat([images, images], dim=0), signs=cat_signs)
        all_logits = model.forward(inputs_embeds=embeds, labels=None, attention_mask=mask).logits.to(torch.float32)
        all_logps = self.cal_batch_logp(all_logits, labels)
        if not self.is_encoder_decoder:
            labels = labels[:, 1:].clone()
            signs = signs[:, 1:].clone()
            all_logits = all_logits[:, :-1, :]
        return all_logps[:B], all_logps[B:], labels, all_logits, signs

    def reference_forward(self, model, inputs):
        (_, _, mask, _, embeds, labels) = model.prepare_inputs_labels_for_multimodal(
            input_ids=inputs["ref_input_ids"], position_ids=None, attention_mask=inputs["ref_attention_mask"],
            past_key_values=None, labels=inputs["ref_labels"], images=inputs["ref_images"])
        logits = model.forward(inputs_embeds=embeds, labels=None, attention_mask=mask).logits.to(torch.float32)
        logps = self.cal_batch_logp(logits, labels)
        if not self.is_encoder_decoder:
            labels = labels[:, 1:].clone()
            logits = logits[:, :-1, :]
        return logps, labels, logits

    def compute_loss(self, model, inputs: Dict[str, Any], return_outputs=False):
        """loss = mean_{b,p} log(1 + exp(neg_acc - pos_acc)) + loss_alpha * sum(KL(ref || policy)) / B.
        Like the reference (halva_trainer.py:548,573) the `model` argument is ignored in favour of self.model."""
        pos_logps, neg_logps, labels, _, signs = self.concatenated_forward(self.model, inputs)
        B = pos_logps.shape[0]
        valid = (labels != IGNORE_INDEX)
        pos_logps = pos_logps * valid[:B].float()
        neg_logps = neg_logps * valid[B:].float()
        signs = signs.masked_fill(signs == IGNORE_INDEX, 0)
        pos_acc = self.accumulate_logps(pos_logps, signs[:B])
        neg_acc = self.accumulate_logps(neg_logps, signs[B:])
        alignment = torch.log(1 + torch.exp(neg_acc - pos_acc)).mean()

        _, _, pol_logits = self.reference_forward(self.model, inputs)
        with torch.no_grad():
            _, ref_labels, ref_logits = self.reference_forward(self.ref_model, inputs)
        w = (ref_labels != IGNORE_INDEX).float().reshape(-1).contiguous()
        V = ref_logits.shape[-1]
        kl = K.kl_rows(pol_logits.reshape(-1, V).contiguous(), ref_logits.reshape(-1, V).contiguous(), w)
        divergence = kl.sum() / ref_logits.shape[0]
        return alignment + self.loss_alpha * divergence

    # -- engine-backed training -------------------------------------------------------------------
    def _setup_engine(self):
        if self._engine is not None:
            return
        a = self.args
        self._flat = dpa.FlatTrainables(dpa.trainable_named_parameters(self.model))
        if self.dist.active:         # replicas start from rank 0's trainable tensors (the LoRA A factors are random)
            dp.broadcast_(self._flat.master, self.dist)
            self._flat.flat.copy_(self._flat.master)
        dpa.bind_model(self._flat, self.model)
        dpa.set_grad_sink(self.model, True)
        self._engine = dpa.DPAEngine(self.model, self.ref_model, self.loss_alpha,
                                     pairs_per_group=int(os.environ.get("HALVA_PAIRS_PER_GROUP", "4")),
                                     ref_rows_per_group=int(os.environ.get("HALVA_REF_ROWS_PER_GROUP", "8")))
        if getattr(a, "gradient_checkpointing", False) and os.environ.get("HALVA_FORCE_CHECKPOINT", "0") == "1":
            self.model.get_model().gradient_checkpointing = True

    def create_optimizer(self):
        self._setup_engine()
        if self.optimizer is None:
            a = self.args
            self.optimizer = dpa.AdamWFlat(self._flat, lr=a.learning_rate, weight_decay=a.weight_decay,
                                           mm_projector_lr=getattr(a, "mm_projector_lr", None),
                                           betas=(getattr(a, "adam_beta1", 0.9), getattr(a, "adam_beta2", 0.999)),
                                           eps=getattr(a, "adam_epsilon", 1e-8))
        return self.optimizer

    def training_step(self, inputs, scale=1.0, reducer=None):
        """One micro-batch: forward + backward group by group; gradients accumulate in the flat fp32 buffer.
        reducer: set on the last micro-batch before an optimizer step - the gradient exchange then starts inside its backward."""
        self._setup_engine()
        return self._engine.loss(inputs, backward=True, scale=scale, reducer=reducer)

    def _get_train_sampler(self):
        a = self.args
        if self.train_dataset is None:
            return None
        if getattr(a, "group_by_modality_length", False):
            # reference quirk kept: world_size := world_size * gradient_accumulation_steps (halva_trainer.py:269)
            return LengthGroupedSampler(a.per_device_train_batch_size * 1, world_size=self.dist.world * a.gradient_accumulation_steps,
                                        lengths=self.train_dataset.modality_lengths, group_by_modality=True)
        g = torch.Generator().manual_seed(getattr(a, "seed", 42))
        return torch.utils.data.RandomSampler(self.train_dataset, generator=g)

    def get_train_dataloader(self):
        a = self.args
        sampler = self._get_train_sampler()
        order = list(iter(sampler))
        bs = a.per_device_train_batch_size
        # accelerate-style batch sharding: global batches of bs, rank r takes batches r, r+world, ...
        batches = [order[i:i + bs] for i in range(0, len(order), bs)]
        if not getattr(a, "dataloader_drop_last", False) and len(batches) % self.dist.world:
            batches += batches[:self.dist.world - len(batches) % self.dist.world]      # wrap around like accelerate even_batches
        mine = batches[self.dist.rank::self.dist.world]
        return DataLoader(self.train_dataset, batch_sampler=mine, collate_fn=self.data_collator,
                          num_workers=getattr(a, "dataloader_num_workers", 0))

    def add_callback(self, cb):
        self.callbacks.append(cb() if isinstance(cb, type) else cb)

    def log(self, rec):
        self.state.log_history.append(rec)
        if self.dist.rank == 0:
            print(json.dumps(rec), flush=True)

    # -- checkpoints (HF Trainer._save_checkpoint / _load_from_checkpoint as the reference inherits them; the adapter-only
    #    branch of reference halva_trainer.py:365-390 for tune_mm_mlp_adapter) ---------------------------------------------
    CKPT_PREFIX = "checkpoint"

    def _checkpoint_dirs(self):
        out = getattr(self.args, "output_dir", None)
        if not out or not os.path.isdir(out):
            return []
        found = []
        for n in os.listdir(out):
            if n.startswith(self.CKPT_PREFIX + "-") and n.split("-", 1)[1].isdigit() and os.path.isdir(os.path.join(out, n)):
                found.append((int(n.split("-", 1)[1]), os.path.join(out, n)))
        return [p for _, p in sorted(found)]

    def _save_checkpoint(self, model=None, trial=None, metrics=None):
        """checkpoint-<global_step>/ : adapter weights in the output format (adapter_model.bin + non_lora_trainables.bin +
        config.json), or only mm_projector.bin when tune_mm_mlp_adapter; plus everything needed to continue bit-for-bit:
        fp32 master weights, AdamW moments, step / epoch / micro-batch position (an epoch's batch order follows from seed + epoch)."""
        a = self.args
        folder = os.path.join(a.output_dir, "%s-%d" % (self.CKPT_PREFIX, self.state.global_step))
        save_error = None
        if self.dist.rank == 0:
            try:
                os.makedirs(folder, exist_ok=True)
                self.model.config.save_pretrained(folder)
                if getattr(a, "tune_mm_mlp_adapter", False):
                    proj, _ = dpa._projector_of(self.model)
                    torch.save({"model.mm_projector." + k: v.detach().cpu() for k, v in proj.state_dict().items()},
                               os.path.join(folder, "mm_projector.bin"))
                else:
                    self._save_adapter(folder)
                torch.save({"master": self._flat.master.detach().cpu(), "names": list(self._flat.names),
                            "optimizer": self.optimizer.state_dict(), "global_step": self.state.global_step,
                            "epoch_index": self._pos["epoch"], "micro_in_epoch": self._pos["micro"],
                            "micro_total": self._pos["total"], "pending_micro": self._pos.get("pending", 0),
                            "epoch_rng_state": self._pos["rng"], "log_history": self.state.log_history,
                            "world": self.dist.world},
                           os.path.join(folder, "halva_state.pt"))
                with open(os.path.join(folder, "trainer_state.json"), "w") as f:
                    json.dump({"global_step": self.state.global_step, "epoch": self.state.epoch, "log_history": self.state.log_history}, f,
                              indent=1)
                limit = getattr(a, "save_total_limit", None)
                if limit is not None and limit > 0:                   # HF _rotate_checkpoints: keep the newest `limit`
                    for old in self._checkpoint_dirs()[:-limit]:
                        import shutil
                        shutil.rmtree(old, ignore_errors=True)
            except Exception as e:      # (the other ranks are waiting at the barrier below: tell them instead of leaving them there)
                save_error = e

        # rank 0's folder (with halva_state.pt) exists before any other rank looks for it - and every rank must SEE it: the per-rank files below
        # are only of use beside rank 0's state, so output_dir has to be on a filesystem all ranks share.  A rank that does not see the folder
        # (node-local output_dir on a multi-node run) used to write its accumulator into a private folder without halva_state.pt, which the
        # resumed run then could not load, or loaded from a stale one (round-4 advice); now every rank fails together, at save time.
        # Round 6: (i) a failure of rank 0's save is a collective decision too - broadcast through the same MAX reduction, every rank
        # raises; (ii) on NFS-like filesystems a rank can briefly miss a file another host has just written (attribute / negative-lookup caches):
        # the existence check is retried for a few seconds, listing the parent directory in between (which refreshes those caches), before the
        # folder is declared invisible.
        if dp.max_scalar(1.0 if save_error is not None else 0.0, self.dist) > 0:
            raise RuntimeError("rank 0 could not write checkpoint %s: %s" % (folder, save_error if save_error is not None else "(see rank 0)"))
        seen = self._wait_until_visible(os.path.join(folder, "halva_state.pt"))
        if dp.max_scalar(0.0 if seen else 1.0, self.dist) > 0:
            raise RuntimeError("checkpoint folder %s written by rank 0 is not visible to every rank (this rank %d: %s): --output_dir must be on "
                               "a filesystem shared by all ranks" % (folder, self.dist.rank, "visible" if seen else "NOT visible"))
        # An end-of-epoch checkpoint can fall INSIDE an accumulation group (micro-batches are counted across epochs, 4.31's
        # total_batched_samples): the fp32 accumulator then holds the group's first micro-batches, which the resumed run must not lose.
        # Every rank's accumulator is its own (nothing has been exchanged yet), so every rank writes its own file.
        # `pending` counts the micro-batches accumulated since the last optimizer step (NOT total % accum: a step forced by an epoch
        # shorter than one group zeroes the accumulator at any count).
        pending = int(self._pos.get("pending", 0))
        if pending:
            torch.save({"grad": self._flat.grad.detach().cpu(), "pending_micro": pending},
                       os.path.join(folder, "halva_pending_grad_rank%d.pt" % self.dist.rank))
        dp.barrier(self.dist)
        return folder

    @staticmethod
    def _wait_until_visible(path, seconds=5.0):
        """os.path.exists with retries: True as soon as `path` shows up, False after `seconds`."""
        import time
        deadline = time.time() + seconds
        while True:
            if os.path.exists(path):
                return True
            if time.time() >= deadline:
                return False
            try:
                os.listdir(os.path.dirname(path) or ".")      # refreshes a network filesystem's directory / negative-lookup cache
            except OSError:
                pass
            time.sleep(0.2)

    def _save_adapter(self, folder):
        """The trained tensors in the run's output naming (llava/train/train_halva.py:save_lora_outputs; VILA overrides)."""
        from llava.train.train_halva import save_lora_outputs
        a = self.args
        save_lora_outputs(self.model, type("A", (), dict(output_dir=folder, lora_bias=getattr(a, "lora_bias", "none"),
                                                         lora_r=getattr(a, "lora_r", 0), lora_alpha=getattr(a, "lora_alpha", 0),
                                                         lora_dropout=getattr(a, "lora_dropout", 0.0)))())

    def _load_from_checkpoint(self, folder):
        st_path = os.path.join(folder, "halva_state.pt")
        if not os.path.exists(st_path):
            raise FileNotFoundError("%s holds no halva_state.pt: cannot resume from it (adapter-only checkpoints carry no "
                                    "optimizer state)" % folder)
        st = torch.load(st_path, map_location="cpu", weights_only=False)
        if list(st["names"]) != list(self._flat.names):
            raise RuntimeError("checkpoint %s was written for a different set of trainable tensors" % folder)
        if st.get("world", self.dist.world) != self.dist.world:
            raise RuntimeError("checkpoint %s was written by %d ranks, this run has %d: the per-rank batch order would differ"
                               % (folder, st["world"], self.dist.world))
        self._flat.master.copy_(st["master"])
        self._flat.sync_compute_copy()
        self.optimizer.load_state_dict(st["optimizer"])
        self.state.global_step = int(st["global_step"])
        self.state.log_history = list(st.get("log_history", []))
        st["_folder"] = folder
        return st

    def _restore_pending_gradient(self, st, accum):
        """The accumulator of a checkpoint written inside an accumulation group (see _save_checkpoint); call after zero_grad()."""
        pending = int(st["pending_micro"]) if "pending_micro" in st else int(st["micro_total"]) % accum
        if pending == 0:
            return
        path = os.path.join(st["_folder"], "halva_pending_grad_rank%d.pt" % self.dist.rank)
        if not os.path.exists(path):
            raise FileNotFoundError("%s was written %d micro-batch(es) into an accumulation group but holds no %s: the resumed "
                                    "optimizer step would miss them" % (st["_folder"], pending, os.path.basename(path)))
        g = torch.load(path, map_location="cpu", weights_only=False)
        if int(g["pending_micro"]) != pending or g["grad"].numel() != self._flat.grad.numel():
            raise RuntimeError("%s does not match the checkpoint's position / trainable tensors" % path)
        self._flat.grad.copy_(g["grad"])

    def _should_save(self, end_of_epoch):
        a = self.args
        strat = str(getattr(a, "save_strategy", "no")).split(".")[-1].lower()
        if strat == "steps" and not end_of_epoch:
            n = getattr(a, "save_steps", 0)
            n = int(n) if n >= 1 else int(math.ceil(n * self._max_steps))       # HF: a ratio of the total when < 1
            return n > 0 and self.state.global_step % n == 0
        return strat == "epoch" and end_of_epoch

    def train(self, resume_from_checkpoint=None):
        """HF Trainer._inner_training_loop as the reference runs it (transformers 4.31): per epoch the sampler's batches are
        dealt to the ranks, every `gradient_accumulation_steps`-th micro-batch (counted ACROSS epochs, 4.31's
        total_batched_samples) closes an optimizer step - a trailing partial group of an epoch is not stepped on its own, its
        gradients stay in the accumulator (unless the whole epoch is shorter than one group) - total steps =
        ceil(epochs * floor(micro / accum)) or --max_steps, cosine schedule with warm-up over that total, checkpoint-<step>/ folders per --save_strategy /
        --save_steps / --save_total_limit, and resume_from_checkpoint (True = newest checkpoint-* under output_dir)."""
        a = self.args
        torch.manual_seed(getattr(a, "seed", 42))
        self.create_optimizer()
        accum = max(1, int(a.gradient_accumulation_steps))
        reducer = dp.GradReducer.for_flat(self._flat, self.dist) if self.dist.active else None
        resume = None
        if resume_from_checkpoint:
            folder = resume_from_checkpoint if isinstance(resume_from_checkpoint, str) else None
            if folder is None:
                have = self._checkpoint_dirs()
                if not have:
                    raise ValueError("No valid checkpoint found in output directory (%s)" % getattr(a, "output_dir", None))
                folder = have[-1]
            resume = self._load_from_checkpoint(folder)
            self.log({"resumed_from": folder, "step": self.state.global_step})
        t0 = time.time()
        n_epochs = int(math.ceil(a.num_train_epochs))
        self._pos = {"epoch": 0, "micro": 0, "total": 0, "rng": None, "pending": 0}
        self._max_steps = None
        done = False
        total_micro = int(resume["micro_total"]) if resume is not None else 0
        since_step = 0          # micro-batches in the accumulator
        if resume is not None:
            since_step = int(resume["pending_micro"]) if "pending_micro" in resume else total_micro % accum
        self._flat.zero_grad()
        if resume is not None:
            self._restore_pending_gradient(resume, accum)
        for epoch in range(n_epochs):
            if resume is not None and epoch < resume["epoch_index"]:
                continue
            # the order of an epoch is a function of (seed, epoch index) alone: the global torch generator the reference's sampler
            # draws from (generator=None, halva_trainer.py:146-150) is re-seeded per epoch, so a resumed run re-derives the same
            # batches without having to replay the random stream of the epochs before it
            torch.manual_seed(getattr(a, "seed", 42) + epoch)
            rng = None
            loader = self.get_train_dataloader()
            n_micro = len(loader)
            steps_per_epoch = max(1, n_micro // accum)
            max_steps = getattr(a, "max_steps", -1)
            total_steps = max_steps if max_steps and max_steps > 0 else int(math.ceil(a.num_train_epochs * steps_per_epoch))
            self._max_steps = total_steps
            skip = 0
            if resume is not None and epoch == resume["epoch_index"]:
                skip = int(resume["micro_in_epoch"])
                if skip >= n_micro:            # the checkpoint was written at the end of that epoch
                    resume = None
                    continue
            resume = None if skip == 0 else resume
            if self.state.global_step >= total_steps:
                break
            running, seen = 0.0, 0
            for i, batch in enumerate(loader):
                if i < skip:
                    continue
                resume = None
                total_micro += 1
                since_step += 1
                steps_now = total_micro % accum == 0 or (n_micro <= accum and (i + 1) == n_micro)
                batch = self._to_device(batch)
                loss = self.training_step(batch, scale=1.0 / accum, reducer=reducer if steps_now else None)
                running, seen = running + float(loss), seen + 1
                if not steps_now:
                    continue
                if reducer is not None:
                    reducer.finish()
                self.optimizer.set_lr_factor(dpa.cosine_with_warmup(self.state.global_step, total_steps,
                                                                    getattr(a, "warmup_ratio", 0.0)))
                self.optimizer.step()
                self._flat.zero_grad()
                since_step = 0
                self.state.global_step += 1
                self.state.epoch = epoch + (i + 1) / n_micro
                self._pos = {"epoch": epoch, "micro": i + 1, "total": total_micro, "rng": rng, "pending": 0}
                if os.environ.get("HALVA_TRAIN_DEBUG"):
                    print("debug step %d epoch %d micro %d ids-sum %d master-sum %.9f grad-state %s" % (
                        self.state.global_step, epoch, i, int(batch["input_ids"].sum()), float(self._flat.master.double().sum()),
                        [float(v["exp_avg"].double().sum()) for v in self.optimizer.opt.state.values()]), flush=True)
                if self.state.global_step % max(1, int(getattr(a, "logging_steps", 1))) == 0:
                    mean = dp.mean_scalar(running / seen, self.dist)
                    self.log({"loss": round(mean, 6), "step": self.state.global_step, "epoch": round(self.state.epoch, 4),
                              "learning_rate": self.optimizer.opt.param_groups[0]["lr"], "elapsed_s": round(time.time() - t0, 1)})
                running, seen = 0.0, 0
                if self._should_save(end_of_epoch=False):
                    self._save_checkpoint()
                if self.state.global_step >= total_steps:
                    done = True
                    break
            if done:
                break
            self._pos = {"epoch": epoch + 1, "micro": 0, "total": total_micro, "rng": None, "pending": since_step}
            if self._should_save(end_of_epoch=True):
                self._save_checkpoint()
        for cb in self.callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end(a, self.state, None, model=self.model)
        return self.state

    def _image_pipeline(self):
        """GPU preprocessing of decoded uint8 images (data_args.gpu_image_pipeline); built from the tower's processor."""
        if getattr(self, "_pipe", None) is None:
            from halva_amd.image_pipeline import GpuImagePipeline
            da = self.train_dataset.data_args
            self._pipe = GpuImagePipeline.from_processor(da.image_processor, da.image_aspect_ratio, device=self.model.device)
        return self._pipe

    @staticmethod
    def _is_raw(v):
        return isinstance(v, (list, tuple)) and len(v) > 0 and all(t.dtype == torch.uint8 and t.ndim == 3 and t.shape[-1] == 3 for t in v) \
            or (isinstance(v, torch.Tensor) and v.dtype == torch.uint8 and v.ndim == 4 and v.shape[-1] == 3)

    def _to_device(self, batch):
        dev = self.model.device
        out = dict(batch)
        for k in ("images", "ref_images"):
            v = out[k]
            if self._is_raw(v):
                out[k] = self._image_pipeline()(list(v))
            else:
                out[k] = (torch.stack(v) if isinstance(v, (list, tuple)) else v).to(dev, torch.bfloat16, non_blocking=True)
        return out          # integer tensors stay on the host: the splice plan is computed there

    def save_state(self):
        a = self.args
        if self.dist.rank == 0 and getattr(a, "output_dir", None):
            os.makedirs(a.output_dir, exist_ok=True)
            with open(os.path.join(a.output_dir, "trainer_state.json"), "w") as f:
                json.dump({"global_step": self.state.global_step, "epoch": self.state.epoch, "log_history": self.state.log_history}, f,
                          indent=1)
