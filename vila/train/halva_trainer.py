"""HalvaTrainer of the VILA twin (reference vila/train/halva_trainer.py:440-852) on the MI355X step engine.

Same loss as the LLaVA trainer (llava/train/halva_trainer.py here); what differs, kept as in the reference:
  * the model does the signed splice inside forward(signs=...) and hands back outputs.labels / outputs.signs (:721-731);
  * images are [B, n, 3, H, W]; the reference forward squeezes dim 1 of ref_images (:761);
  * construction is Trainer(...) followed by custom_setup(model=, ref_model=, ...) (:442-488);
  * the length-grouped sampler is commented out (:513-527): plain seeded RandomSampler;
  * every step prints the two loss terms and appends them to self.loss_holder (:844-848).
"""
from collections import defaultdict
from typing import Any, Dict, Optional

import torch

from halva_amd import kernels as K
from llava.train.halva_trainer import HalvaTrainer as _Base
from llava.train.halva_trainer import (LengthGroupedSampler, TrainerState, disable_dropout_in_model,  # noqa: F401
                                       get_length_grouped_indices, get_modality_length_grouped_indices)
from halva_amd import dp

IGNORE_INDEX = -100


class HalvaTrainer(_Base):
    def __init__(self, model=None, tokenizer=None, args=None, train_dataset=None, eval_dataset=None, data_collator=None,
                 callbacks=None, optimizers=(None, None), **unused):
        self.model, self.ref_model, self.args = model, None, args
        self.data_collator, self.train_dataset, self.eval_dataset, self.tokenizer = data_collator, train_dataset, eval_dataset, tokenizer
        self.callbacks = list(callbacks or [])
        self.optimizer, self.lr_scheduler = optimizers
        self.state = TrainerState()
        self.loss_holder = defaultdict(list)
        self.loss_alpha, self.label_pad_token_id, self.padding_value, self.is_encoder_decoder = 0.1, IGNORE_INDEX, 0, False
        self._engine = self._flat = None
        self.dist = dp.DistContext.from_env()

    def custom_setup(self, model=None, ref_model=None, label_pad_token_id: int = -100, padding_value: int = 0,
                     is_encoder_decoder: bool = False, loss_alpha: Optional[float] = 0.1, disable_dropout: bool = True):
        if ref_model is None:
            raise ValueError("HalvaTrainer needs a frozen ref_model (adapter-disabling is not implemented on this path)")
        self.ref_model = ref_model
        if disable_dropout:
            disable_dropout_in_model(model if model is not None else self.model)
            disable_dropout_in_model(ref_model)
        self.loss_alpha, self.label_pad_token_id = loss_alpha, label_pad_token_id
        self.padding_value, self.is_encoder_decoder = padding_value, is_encoder_decoder
        self.loss_holder = defaultdict(list)
        ref_model.eval()
        for p in ref_model.parameters():
            p.requires_grad_(False)

    # -- reference tensor-level API (full logits; the engine below never materialises them) --------
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

        images = inputs["images"]
        out = model(input_ids=stack(ids, neg, 0, ids.dtype), images=torch.cat([images, images], dim=0),
                    labels=stack(inputs["labels"], inputs["neg_labels"], IGNORE_INDEX, inputs["labels"].dtype),
                    attention_mask=stack(inputs["attention_mask"], inputs["neg_attention_mask"], False, torch.bool),
                    signs=stack(inputs["pos_signs"], inputs["neg_signs"], 0, inputs["pos_signs"].dtype))
        all_logits, labels, signs = out.logits.to(torch.float32), out.labels, out.signs
        all_logps = self.cal_batch_logp(all_logits, labels)
        if not self.is_encoder_decoder:
            labels, signs, all_logits = labels[:, 1:].clone(), signs[:, 1:].clone(), all_logits[:, :-1, :]
        return all_logps[:B], all_logps[B:], labels, all_logits, signs

    def reference_forward(self, model, inputs):
        images = inputs["ref_images"]
        out = model(input_ids=inputs["ref_input_ids"], images=images.squeeze(1), attention_mask=inputs["ref_attention_mask"],
                    labels=inputs["ref_labels"])
        logits, labels = out.logits.to(torch.float32), out.labels
        logps = self.cal_batch_logp(logits, labels)
        if not self.is_encoder_decoder:
            labels, logits = labels[:, 1:].clone(), logits[:, :-1, :]
        return logps, labels, logits

    def compute_loss(self, model, inputs: Dict[str, Any], return_outputs=False):
        """vila/train/halva_trainer.py:783-852; uses the `model` argument (the wrapped model) like the reference (:795,821)."""
        pos_logps, neg_logps, labels, _, signs = self.concatenated_forward(model, inputs)
        B = pos_logps.shape[0]
        valid = (labels != IGNORE_INDEX)
        signs = signs.masked_fill(signs == IGNORE_INDEX, 0)
        pos_acc = self.accumulate_logps(pos_logps * valid[:B].float(), signs[:B])
        neg_acc = self.accumulate_logps(neg_logps * valid[B:].float(), signs[B:])
        contrastive = torch.log(1 + torch.exp(neg_acc - pos_acc)).mean()
        _, _, pol_logits = self.reference_forward(model, inputs)
        with torch.no_grad():
            _, ref_labels, ref_logits = self.reference_forward(self.ref_model, inputs)
        w = (ref_labels != IGNORE_INDEX).float().reshape(-1).contiguous()
        V = ref_logits.shape[-1]
        kl = K.kl_rows(pol_logits.reshape(-1, V).contiguous(), ref_logits.reshape(-1, V).contiguous(), w)
        divergence = kl.sum() / ref_logits.shape[0]
        loss = contrastive + self.loss_alpha * divergence
        self._record(loss, contrastive, divergence)
        return loss

    def _record(self, loss, contrastive, divergence):
        l, c, d = (round(float(x), 7) for x in (loss, contrastive, divergence))
        print(f"[loss: {l} contrastive_loss:  {c}, divergence: {d}]")
        self.loss_holder["contrastive_loss"].append(c)
        self.loss_holder["divergence"].append(d)
        self.loss_holder["loss"].append(l)

    # -- engine-backed training ---------------------------------------------------------------------
    def training_step(self, inputs, scale=1.0, reducer=None):
        self._setup_engine()
        loss = self._engine.loss(inputs, backward=True, scale=scale, reducer=reducer)
        p = self._engine.last_parts
        self._record(loss, p["alignment"], p["divergence"])
        return loss

    def _save_adapter(self, folder):
        """checkpoint-<step>/ in VILA's output naming (llm.base_model.model... + mm_projector...)."""
        from vila.train.train_halva import save_lora_outputs
        a = self.args
        save_lora_outputs(self.model, type("A", (), dict(output_dir=folder, lora_bias=getattr(a, "lora_bias", "none"),
                                                         lora_r=getattr(a, "lora_r", 0), lora_alpha=getattr(a, "lora_alpha", 0),
                                                         lora_dropout=getattr(a, "lora_dropout", 0.0)))())

    def _get_train_sampler(self):
        if self.train_dataset is None:
            return None
        a = self.args
        seed = getattr(a, "data_seed", None)
        g = torch.Generator().manual_seed(seed if seed is not None else getattr(a, "seed", 42))
        return torch.utils.data.RandomSampler(self.train_dataset, generator=g)

    def _to_device(self, batch):
        dev = self.model.device
        out = dict(batch)
        for k in ("images", "ref_images"):
            v = out[k]
            if self._is_raw(v):
                out[k] = self._image_pipeline()(list(v)).unsqueeze(1)          # [B, 1, 3, S, S] like the CPU path
                continue
            if isinstance(v, (list, tuple)):
                v = torch.stack([t if t.ndim == 4 else t[None] for t in v])
            out[k] = v.to(dev, torch.bfloat16, non_blocking=True)
        return out
