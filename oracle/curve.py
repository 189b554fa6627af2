"""Oracle (test infrastructure): the reference's TRAINING CURVE - K optimizer steps of compute_loss + AdamW - restated.

What the reference runs (none of it is in /root/reference itself; it is inherited from the recipe's flags):
  * `--bf16 True` with DeepSpeed's bf16 engine (src/hallava_7b.sh:48, src/json/zero3.json "bf16": {"enabled": "auto"}): fp32 MASTER
    weights and fp32 AdamW moments live in the optimizer; the tensors the forward/backward see are a bf16 copy that is re-rounded
    from the master after every step.  Restated here as fp32 arithmetic with a straight-through bf16 rounding of every TRAINABLE
    tensor (`bf16_params=True`): value = bf16(master), gradient = identity.  `bf16_params=False` keeps every parameter fp32 (plain
    torch.optim.AdamW on fp32 leaves) - a second witness that isolates how much of a curve difference is that parameter rounding.
  * optimizer: torch.optim.AdamW, betas (0.9, 0.999), eps 1e-8, weight decay 0 (`optim="adamw_torch"`, reference
    llava/train/train_halva.py:70; src/hallava_7b.sh:58), two learning-rate groups - LoRA factors at `lr`, mm_projector at
    `mm_projector_lr` (reference llava/train/halva_trainer.py:291-337).
  * loss: oracle.dpa.compute_loss (reference llava/train/halva_trainer.py:534-592).

Pinning: step 0 of either curve is the reference's own `compute_loss` value on the fixture (tests/test_oracle_vs_golden.py and
tests/test_loss_curve_gpu.py assert it); the later steps are this restatement of the inherited optimizer semantics - HF Trainer 4.31 +
DeepSpeed 0.9.5 cannot run offline, so they are NOT pinned to a run of the reference, and the 1e-3 curve claim is a claim against this
oracle.  See oracle/__init__.py for the rules (test infrastructure only).
"""
import torch

from . import dpa as odpa


def ste_bf16(t):
    """Value = bf16(t), gradient = identity: the bf16 compute copy of an fp32 master tensor."""
    return t + (t.detach().bfloat16().float() - t.detach())


def training_curve(base, clip_weights, llama_cfg, clip_cfg, max_len, lora, lora_r, lora_alpha, batch, loss_alpha, steps, lr,
                   mm_projector_lr, bf16_params=False):
    """Losses of `steps` consecutive optimizer steps on ONE batch (the loss is evaluated before each update).
    base / clip_weights / lora: {name: fp32 tensor} as in the dpa_step_* fixtures; batch: the collated batch (numpy arrays)."""
    ref = odpa.TinyLlava(base, llama_cfg, clip_weights, clip_cfg, max_len)
    pol_W = {k: v.clone() for k, v in base.items()}
    proj = [k for k in pol_W if "mm_projector" in k]
    for k in proj:
        pol_W[k].requires_grad_(True)
    lora = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    pol = odpa.TinyLlava(pol_W, llama_cfg, clip_weights, clip_cfg, max_len, lora=lora, lora_scale=float(lora_alpha / lora_r))
    pol.W = pol_W
    opt = torch.optim.AdamW([{"params": list(lora.values()), "lr": lr, "weight_decay": 0.0},
                             {"params": [pol_W[k] for k in proj], "lr": mm_projector_lr, "weight_decay": 0.0}],
                            lr=lr, betas=(0.9, 0.999), eps=1e-8)
    curve = []
    for _ in range(steps):
        opt.zero_grad()
        if bf16_params:
            pol.W = {k: (ste_bf16(v) if v.requires_grad else v) for k, v in pol_W.items()}
            pol.lora = {k: ste_bf16(v) for k, v in lora.items()}
        loss, _ = odpa.compute_loss(pol, ref, batch, loss_alpha)
        loss.backward()
        opt.step()
        curve.append(float(loss))
    return curve
