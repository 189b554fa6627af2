#!/usr/bin/env python3
"""Generate the VILA-path golden vectors under tests/golden/ by IMPORTING THE REFERENCE (CPU, this container only).

Test infrastructure; the fixtures are data (inputs + the reference's own outputs), the reference never travels.
`vila/model/__init__.py` drags in the whole VILA zoo (radio / intern encoders, a transformers fork, flash-attn, s2wrapper),
most of which cannot import under the installed transformers.  The files on the HALVA path can, so they are loaded
one by one through synthetic namespace packages (the package __init__ files are not executed):

  vila/model/multimodal_projector/base_projector.py   DownSampleBlock, MultimodalProjector(mlp_downsample)
  vila/model/multimodal_encoder/siglip/modeling_siglip.py   SiglipVisionModel
  vila/model/multimodal_encoder/vision_encoder.py     VisionTower.forward / feature_select
  vila/model/llava_arch.py                            prepare_inputs_labels_for_multimodal_signed, encode_images
  vila/model/language_model/llava_llama.py            LlavaLlamaModel.forward (signs= path)
  vila/train/halva_trainer.py                         HalvaTrainer.{cal_batch_logp,accumulate_logps,concatenated_forward,
                                                      reference_forward,compute_loss}

Absent third-party modules (s2wrapper, deepspeed, wandb, flash_attn, peft) and the two reference builder modules whose
imports cannot resolve here get name-only stand-ins; nothing under test calls into them.  `self.llm` is the vendored
transformers-4.31 LlamaForCausalLM of the reference (llava/model/language_model/modelling_llama.py, eager attention) -
VILA's own transformers fork differs from it only by the flash-attn varlen call (`seqlens_in_batch`), which has no CPU path.

Usage:  python tests/golden/make_golden_vila.py
"""
import collections
import copy
import importlib
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HALVA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402  (imports the llava side of the reference with its shims; reuses its helpers)

t2n, save_npz = MG.t2n, MG.save_npz


class _Names(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return None


def _stub(name):
    m = _Names(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m


def _ns(name):
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, *name.split("."))]
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    sys.modules[name] = m


def import_vila():
    from transformers import AutoModel
    AutoModel.register = classmethod(lambda cls, *a, **k: None)
    import transformers.modeling_utils as MU
    import transformers.trainer as T
    if not hasattr(MU, "no_init_weights"):
        MU.no_init_weights = None
    if not hasattr(MU, "unwrap_model"):
        MU.unwrap_model = None
    for n in ("ALL_LAYERNORM_LAYERS", "ShardedDDPOption"):
        if not hasattr(T, n):
            setattr(T, n, object())
    for n in ("vila", "vila.model", "vila.model.multimodal_encoder", "vila.model.multimodal_encoder.siglip",
              "vila.model.language_model", "vila.model.multimodal_projector", "vila.train"):
        _ns(n)
    for n in ("peft", "peft.peft_model", "s2wrapper", "deepspeed", "wandb", "vila.model.multimodal_encoder.builder", "vila.model.language_model.builder",
              "vila.model.multimodal_projector.builder"):
        _stub(n)
    mods = {}
    for key, name in (("BP", "vila.model.multimodal_projector.base_projector"),
                      ("SG", "vila.model.multimodal_encoder.siglip.modeling_siglip"),
                      ("SGC", "vila.model.multimodal_encoder.siglip.configuration_siglip"),
                      ("VE", "vila.model.multimodal_encoder.vision_encoder"),
                      ("ARCH", "vila.model.llava_arch"),
                      ("LL", "vila.model.language_model.llava_llama"),
                      ("HT", "vila.train.halva_trainer")):
        for _ in range(20):
            try:
                mods[key] = importlib.import_module(name)
                break
            except ModuleNotFoundError as e:
                if e.name.startswith("vila."):
                    raise
                _stub(e.name)
        else:
            raise RuntimeError("could not import " + name)
    return types.SimpleNamespace(**mods)


V = import_vila()
V431 = MG.V431

TINY = dict(MG.TINY64)
SIG = dict(hidden_size=144, intermediate_size=160, num_hidden_layers=3, num_attention_heads=2, image_size=48, patch_size=14,
           hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, num_channels=3)


def bits(t):
    return t.detach().bfloat16().view(torch.int16).numpy().view(np.uint16)


# ----------------------------------------------------------------------------------------------
# V1: DownSampleBlock + mlp_downsample (base_projector.py:33-54,76-83)
# ----------------------------------------------------------------------------------------------
def gen_downsample():
    g = torch.Generator().manual_seed(5)
    blk = V.BP.DownSampleBlock()
    packs = {}
    for name, n, grid, c in (("odd", 2, 3, 8), ("even", 1, 4, 8), ("siglip", 1, 27, 16)):
        x = torch.randn(n, grid * grid, c, generator=g)
        packs[name + ".x"], packs[name + ".y"] = t2n(x), t2n(blk(x))
    cfg = types.SimpleNamespace(mm_hidden_size=16, hidden_size=32)
    proj = V.BP.MultimodalProjector(V.BP.MultimodalProjectorConfig("mlp_downsample"), cfg)
    with torch.no_grad():
        for n_, p in proj.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2 + (1.0 if n_ == "layers.1.weight" else 0.0))
    x = torch.randn(2, 9, 16, generator=g)
    y = proj(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    packs.update({"proj.x": t2n(x), "proj.y": t2n(y), "proj.gy": t2n(gy)})
    for n_, p in proj.named_parameters():
        packs["proj.w." + n_], packs["proj.g." + n_] = t2n(p), t2n(p.grad)
    save_npz("vila_downsample.npz", **packs)


# ----------------------------------------------------------------------------------------------
# V2: SigLIP tower as VisionTower.forward sees it (vision_encoder.py:23-32,121-140)
# ----------------------------------------------------------------------------------------------
def build_tower(seed, sig=None, select_feature="cls_patch", bf16_round=True):
    sig = sig or SIG
    cfg = V.SGC.SiglipVisionConfig(**sig)
    tower = V.VE.VisionTower("tiny-siglip", types.SimpleNamespace(mm_vision_select_layer=-2,
                                                                 mm_vision_select_feature=select_feature))
    torch.manual_seed(seed)
    tower.vision_tower = V.SG.SiglipVisionModel(cfg)
    with torch.no_grad():
        for n, p in tower.vision_tower.named_parameters():
            p.copy_(torch.randn_like(p) * 0.08)
            if ("layer_norm" in n or "layernorm" in n) and n.endswith("weight"):
                p.add_(1.0)
            if bf16_round:
                p.copy_(p.bfloat16().float())
    tower.vision_tower.requires_grad_(False)
    tower.vision_tower.eval()
    tower.is_loaded = True
    return tower


def tower_state(tower, fn=t2n):
    return {n: fn(p) for n, p in tower.vision_tower.state_dict().items()
            if "position_ids" not in n and ".head." not in n and "post_layernorm" not in n}


def gen_siglip():
    tower = build_tower(61)
    g = torch.Generator().manual_seed(62)
    images = torch.randn(2, 3, SIG["image_size"], SIG["image_size"], generator=g).bfloat16().float()
    feats = tower(images)
    packs = {"images": t2n(images), "features": t2n(feats), "cfg": np.frombuffer(json.dumps(SIG).encode(), dtype=np.uint8)}
    tower.select_feature = "patch"
    packs["features_patch"] = t2n(tower(images))
    for n, p in tower_state(tower).items():
        packs["w." + n] = p
    save_npz("vila_siglip.npz", **packs)


# ----------------------------------------------------------------------------------------------
# the VILA model, assembled from reference classes
# ----------------------------------------------------------------------------------------------
class _Llm(V431.LlamaForCausalLM):
    """reference `self.llm`; VILA's fork adds seqlens_in_batch for flash-attn varlen, dropped here (eager + mask)."""

    def forward(self, *a, seqlens_in_batch=None, **k):
        return super().forward(*a, **k)


class RefVila(torch.nn.Module, V.ARCH.LlavaMetaModel, V.ARCH.LlavaMetaForCausalLM):
    forward = V.LL.LlavaLlamaModel.forward

    def __init__(self, llm, tower, projector, config):
        torch.nn.Module.__init__(self)
        self.llm, self.vision_tower, self.mm_projector, self.config = llm, tower, projector, config

    @property
    def device(self):
        return torch.device("cpu")


def build_vila(seed, tower, max_len=64, padding_side="right", std=0.06, bf16_round=True):
    from transformers import LlamaConfig
    torch.manual_seed(seed)
    lc = LlamaConfig(**TINY)
    lc.rope_scaling, lc.pretraining_tp = None, 1
    lc._attn_implementation = "eager"
    llm = _Llm(lc)
    cfg = types.SimpleNamespace(mm_hidden_size=SIG["hidden_size"], hidden_size=TINY["hidden_size"])
    proj = V.BP.MultimodalProjector(V.BP.MultimodalProjectorConfig("mlp_downsample"), cfg)
    with torch.no_grad():
        for n, p in llm.named_parameters():
            p.copy_(1.0 + 0.1 * torch.randn_like(p) if "norm" in n else torch.randn_like(p) * std)
        for n, p in proj.named_parameters():
            p.copy_(torch.randn_like(p) * 0.1 + (1.0 if n == "layers.1.weight" else 0.0))
        if bf16_round:
            for p in list(llm.parameters()) + list(proj.parameters()):
                p.copy_(p.bfloat16().float())
    llm.config.tokenizer_model_max_length = max_len
    llm.config.tokenizer_padding_side = padding_side
    m = RefVila(llm, tower, proj, types.SimpleNamespace())
    m.eval()
    return m


def trainer_stub(policy, ref, alpha):
    stub = types.SimpleNamespace(model=policy, ref_model=ref, loss_alpha=alpha, label_pad_token_id=-100,
                                 is_encoder_decoder=False, loss_holder=collections.defaultdict(list))
    for n in ("cal_batch_logp", "accumulate_logps", "concatenated_forward", "reference_forward", "compute_loss"):
        setattr(stub, n, types.MethodType(getattr(V.HT.HalvaTrainer, n), stub))
    return stub


# ----------------------------------------------------------------------------------------------
# V3: the multi-image signed splice (llava_arch.py:613-871)
# ----------------------------------------------------------------------------------------------
def gen_splice():
    tower = build_tower(63)
    packs = {}
    g = torch.Generator().manual_seed(64)
    Vv = TINY["vocab_size"]
    rows = [  # (ids with -200 markers, pad) - image counts 1, 2, 0, 1
        [1, 5, 6, -200, 7, 8, 9, 10, 2],
        [1, -200, 11, 12, -200, 13, 2],
        [1, 20, 21, 22, 23, 2],
        [1, 30, -200, 31, 32, 33, 34, 35, 36, 2],
    ]
    L = max(len(r) for r in rows)
    ids = torch.zeros(len(rows), L, dtype=torch.long)
    att = torch.zeros(len(rows), L, dtype=torch.bool)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.tensor(r)
        att[i, :len(r)] = True
    labels = torch.where(att, torch.randint(3, Vv, ids.shape, generator=g), torch.tensor(-100))
    labels[:, :3] = -100
    signs = torch.where(att, torch.randint(0, 3, ids.shape, generator=g), torch.tensor(0))
    images = torch.randn(4, 3, SIG["image_size"], SIG["image_size"], generator=g).bfloat16().float()   # 1+2+0+1 images
    for case, max_len, side in (("right", 64, "right"), ("left", 64, "left"), ("trunc", 9, "right")):
        m = build_vila(65, tower, max_len=max_len, padding_side=side)
        with torch.no_grad():
            out = m.prepare_inputs_labels_for_multimodal_signed(ids.clone(), None, att.clone(), None, labels.clone(), images,
                                                                signs.clone())
        _, pos, mask, _, emb, lab, sg = out
        packs.update({case + ".embeds": t2n(emb), case + ".labels": t2n(lab), case + ".signs": t2n(sg),
                      case + ".mask": t2n(mask), case + ".max_len": np.array(max_len)})
        if case == "right":
            with torch.no_grad():
                out = m.prepare_inputs_labels_for_multimodal(ids.clone(), None, att.clone(), None, labels.clone(),
                                                             [images[:1], images[1:3], images[3:]])   # list input, :646-647
            packs["unsigned.embeds"], packs["unsigned.labels"] = t2n(out[4]), t2n(out[5])
            for n, p in m.llm.state_dict().items():
                if "rotary_emb" not in n:
                    packs["llm." + n] = bits(p)
            for n, p in m.mm_projector.state_dict().items():
                packs["proj." + n] = bits(p)
    for n, p in tower_state(tower, bits).items():
        packs["vis." + n] = p
    packs.update({"ids": t2n(ids), "att": t2n(att), "labels": t2n(labels), "signs": t2n(signs), "images": t2n(images),
                  "vis_cfg": np.frombuffer(json.dumps(SIG).encode(), dtype=np.uint8),
                  "llama_cfg": np.frombuffer(json.dumps(TINY).encode(), dtype=np.uint8)})
    save_npz("vila_splice.npz", **packs)


# ----------------------------------------------------------------------------------------------
# V4: full compute_loss of the VILA trainer (halva_trainer.py:692-852), loss + gradients
# ----------------------------------------------------------------------------------------------
def vila_batch(B, seed, n_img=1):
    """make_golden.make_batch's layout with VILA's image shape [B, n, 3, H, W] (train_halva.py:1078-1085); with
    n_img = 2 every prompt carries two image tokens."""
    batch = MG.make_batch(B, seed, 0, vis=SIG, vocab=TINY["vocab_size"])
    g = torch.Generator().manual_seed(seed + 1000)
    H = SIG["image_size"]
    if n_img > 1:
        def widen(ids, fill, *others):
            # insert (n_img - 1) extra image tokens right after the first one, keeping labels / signs / mask aligned
            pos = [(r == -200).nonzero()[0, 0].item() for r in ids]
            out = []
            for t, f in ((ids, -200),) + tuple(others):
                rows = [torch.cat([r[:p + 1], torch.full((n_img - 1,), f, dtype=r.dtype), r[p + 1:]]) for r, p in zip(t, pos)]
                out.append(torch.stack(rows))
            return out
        for pre in ("", "neg_"):
            sg = "pos_signs" if pre == "" else "neg_signs"
            (batch[pre + "input_ids"], batch[pre + "labels"], batch[pre + "attention_mask"], batch[sg]) = widen(
                batch[pre + "input_ids"], -200, (batch[pre + "labels"], -100), (batch[pre + "attention_mask"], True),
                (batch[sg], 0))
    batch["images"] = torch.randn(B, n_img, 3, H, H, generator=g).bfloat16().float()
    batch["ref_images"] = torch.randn(B, 1, 3, H, H, generator=g).bfloat16().float()
    return batch


def gen_step(name, B, seed, n_img=1, std=0.02, lora_std=0.02, max_len=64, with_grads=True):
    tower = build_tower(71)
    ref = build_vila(400 + seed, tower, max_len=max_len, std=std)
    policy = copy.deepcopy(ref)
    policy.vision_tower = tower
    ref.requires_grad_(False)
    packs = {}
    for n, p in ref.llm.state_dict().items():
        if "rotary_emb" not in n:
            packs["llm." + n] = bits(p)
    for n, p in ref.mm_projector.state_dict().items():
        packs["proj." + n] = bits(p)
    for n, p in tower_state(tower, bits).items():
        packs["vis." + n] = p
    fac = MG.lora_merge(policy.llm, seed + 1, bf16_round=True, std=lora_std)
    for k, v in fac.items():
        packs[k] = bits(torch.from_numpy(v))
    packs["lora_cfg"] = np.array([4, 8.0])
    alpha = 0.4
    batch = vila_batch(B, seed, n_img)
    stub = trainer_stub(policy, ref, alpha)
    policy.zero_grad()
    with torch.no_grad():
        pos_logps, neg_logps, batch_labels, _, batch_signs = stub.concatenated_forward(policy, batch)
    loss = stub.compute_loss(policy, batch)
    if with_grads:
        loss.backward()
    with torch.no_grad():
        mask = (batch_labels != -100)
        half = pos_logps.shape[0]
        sg = batch_signs.masked_fill(batch_signs == -100, 0)
        pa = stub.accumulate_logps(pos_logps * mask[:half].float(), sg[:half])
        na = stub.accumulate_logps(neg_logps * mask[half:].float(), sg[half:])
    for k, v in batch.items():
        packs["batch." + k] = t2n(v)
    packs.update({"out.loss": t2n(loss), "out.alignment": np.array(stub.loss_holder["contrastive_loss"][-1]),
                  "out.divergence": np.array(stub.loss_holder["divergence"][-1]),
                  "out.pos_logps": t2n(pos_logps), "out.neg_logps": t2n(neg_logps), "out.batch_labels": t2n(batch_labels),
                  "out.batch_signs": t2n(batch_signs), "out.pos_acc": t2n(pa), "out.neg_acc": t2n(na),
                  "alpha": np.array(alpha), "max_len": np.array(max_len)})
    if with_grads:
        for n, p in policy.named_parameters():
            if p.grad is not None and ("mm_projector" in n or any(s in n for s in (
                    "layers.0.self_attn.q_proj", "layers.1.mlp.down_proj", "layers.1.self_attn.v_proj",
                    "layers.0.mlp.gate_proj", "layers.1.self_attn.o_proj", "layers.0.self_attn.k_proj"))):
                packs["grad." + n] = t2n(p.grad)
    packs["vis_cfg"] = np.frombuffer(json.dumps(SIG).encode(), dtype=np.uint8)
    packs["llama_cfg"] = np.frombuffer(json.dumps(TINY).encode(), dtype=np.uint8)
    save_npz(name + ".npz", **packs)
    print("   ", name, "loss", float(loss), "align", stub.loss_holder["contrastive_loss"][-1], "div",
          stub.loss_holder["divergence"][-1])


def main():
    torch.set_num_threads(4)
    gen_downsample()
    gen_siglip()
    gen_splice()
    gen_step("vila_step_init", B=3, seed=81)
    gen_step("vila_step_multi", B=2, seed=82, n_img=2, with_grads=False)


if __name__ == "__main__":
    main()
