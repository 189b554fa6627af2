"""VILA's LlavaLlamaModel for the DPA step (MI355X-native), keeping the reference's API surface:

reference vila/model/language_model/llava_llama.py:46-177 (LlavaLlamaModel: `.llm`, `.vision_tower`, `.mm_projector`,
`forward(..., signs=)` returning `outputs.labels / .signs`),
reference vila/model/llava_arch.py:56-253 (LlavaMetaModel: init_vlm / load_pretrained / save_pretrained / getters /
encode_images) and :264-871 (prepare_inputs_labels_for_multimodal[_signed] - the multi-image splice),
reference vila/model/multimodal_projector/base_projector.py:33-97 (DownSampleBlock, MultimodalProjector).

Differences from the LLaVA wrapper (halva_amd/llava_model.py) that matter to the step engine (halva_amd/dpa.py):
images arrive as [B, n, 3, H, W] and are flattened; a row consumes one image per image token and an image-less row
consumes none (vila/model/llava_arch.py:708-718); each image contributes ceil(27/2)^2 = 196 tokens after
mlp_downsample; the language model lives under `.llm` and the projector at the top level (parameter names
`llm.…` / `mm_projector.layers.N.…`).  The compute path is the same set of HIP kernels.
"""
import json
import os
import re
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import kernels as K
from . import splice as SP
from .clip import CLIPVisionConfig, CLIPVisionTower, _read_checkpoint
from .llama import LlamaConfig, LlamaModel, add_lora, hf_llama_state_dict, load_hf_llama_weights
from .siglip import SiglipVisionConfig, SiglipVisionTower

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200


class VilaConfig:
    """LlavaLlamaConfig (vila/model/configuration_llava.py:4-55): a json bag whose llm_cfg / vision_tower_cfg /
    mm_projector_cfg are either dicts (saved checkpoint -> sub-folders of the checkpoint root) or paths."""
    model_type = "llava_llama"

    def __init__(self, **kw):
        self.llm_cfg = self.vision_tower_cfg = self.mm_projector_cfg = None
        self.resume_path = None
        self.hidden_size = self.mm_hidden_size = None
        self.mm_vision_select_layer = -2
        self.mm_vision_select_feature = "cls_patch"
        self.mm_use_im_start_end, self.mm_use_im_patch_token = False, True
        self.mm_projector_lr = None
        self.image_aspect_ratio = None
        self.model_dtype = "torch.bfloat16"
        self._name_or_path = ""
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path, **kw):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.update(kw)
        c = cls(**d)
        c._name_or_path = path
        if c.resume_path is None:
            c.resume_path = path
        return c

    def to_dict(self):
        d = {}
        for k, v in self.__dict__.items():
            if k.startswith("_"):
                continue
            if hasattr(v, "to_dict"):
                v = v.to_dict()
            try:
                json.dumps(v)
            except TypeError:
                continue
            d[k] = v
        d["model_type"] = self.model_type
        return d

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)


def get_model_config(config):
    """vila/model/utils.py:23-53: sub-model locations (dict / config object -> <root>/<key minus _cfg>, str -> itself)."""
    root = config._name_or_path if len(getattr(config, "_name_or_path", "") or "") >= 2 else config.resume_path
    out = []
    for key in ("llm_cfg", "vision_tower_cfg", "mm_projector_cfg"):
        cfg = getattr(config, key, None)
        if isinstance(cfg, str):
            out.append(cfg)
        elif cfg is not None:
            if root is None:
                raise ValueError("Cannot find resume path in config for %s!" % key)
            out.append(os.path.join(root, key[:-4]))
    return out


# ------------------------------------------------------------------------------------------------
class LlamaForCausalLM(nn.Module):
    """The `.llm` of the VILA wrapper: Llama decoder (halva_amd/llama.py) + lm_head, HF parameter names."""

    def __init__(self, config, dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, dtype, device)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False, dtype=dtype, device=device)
        self.lm_head.weight.requires_grad_(False)
        self.pad_token_id = None

    @classmethod
    def from_pretrained(cls, path, dtype=torch.bfloat16, device="cuda", model_max_length=None, **kw):
        if not os.path.isdir(path):
            raise FileNotFoundError("%s is not a local checkpoint directory (no network on this path)" % path)
        cfg = LlamaConfig.from_pretrained(path)
        cfg.model_max_length = model_max_length
        ctx = getattr(cfg, "max_position_embeddings", None)
        if model_max_length is not None and ctx and model_max_length > ctx:        # language_model/builder.py:43-50
            import math
            cfg.rope_scaling = {"type": "linear", "factor": float(math.ceil(model_max_length / ctx))}
        m = cls(cfg, dtype=dtype, device=device)
        load_hf_llama_weights(m, _read_checkpoint(path), strict=True)
        return m

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def enable_input_require_grads(self):
        pass

    def hf_state_dict(self):
        return hf_llama_state_dict(self)

    def save_pretrained(self, path, state_dict=None):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        self.config.save_pretrained(path)
        sd = self.hf_state_dict() if state_dict is None else state_dict
        save_file({k: v.detach().cpu().contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))


# ------------------------------------------------------------------------------------------------
class DownSampleBlock(nn.Module):
    """base_projector.py:33-54 (2x2 token merge of the square grid, odd grids zero padded)."""

    def forward(self, x):
        return K.downsample2x2(x.contiguous())


class MultimodalProjector(nn.Module):
    """base_projector.py:65-97.  Parameter names match the reference (`layers.1.weight` = LayerNorm, `layers.2` /
    `layers.4` = the Linears of mlp_downsample; `layers.0` / `layers.2` for mlpNx_gelu)."""

    def __init__(self, mm_projector_type, config, dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        self.config = SimpleNamespace(mm_projector_type=mm_projector_type, model_type="v2l_projector",
                                      to_dict=lambda: {"mm_projector_type": mm_projector_type, "model_type": "v2l_projector"})
        self.kind = mm_projector_type
        c, h = config.mm_hidden_size, config.hidden_size
        mk = dict(dtype=dtype, device=device)
        if mm_projector_type == "mlp_downsample":
            self.layers = nn.Sequential(DownSampleBlock(), nn.LayerNorm(c * 4, **mk), nn.Linear(c * 4, h, **mk), nn.GELU(),
                                        nn.Linear(h, h, **mk))
        elif re.match(r"^mlp2x_gelu$", mm_projector_type):
            self.layers = nn.Sequential(nn.Linear(c, h, **mk), nn.GELU(), nn.Linear(h, h, **mk))
        else:
            raise ValueError("Unsupported projector type on the MI355X DPA path: %s (src_vila/halva_vila_13b.sh uses "
                             "mlp_downsample)" % mm_projector_type)

    def forward(self, x, *args, **kwargs):
        L = self.layers
        x = x.to(L[-1].weight.dtype)                   # the tower hands features back in the images' dtype
        if self.kind == "mlp_downsample":
            return K.downsample_mlp(x, L[1].weight, L[1].bias, L[1].eps, L[2].weight, L[2].bias, L[4].weight, L[4].bias)
        return K.projector_mlp(x, L[0].weight, L[0].bias, L[2].weight, L[2].bias)

    def tokens_per_image(self, n_patches):
        if self.kind == "mlp_downsample":
            g = int(n_patches ** 0.5)
            return ((g + 1) // 2) ** 2
        return n_patches

    @classmethod
    def from_pretrained(cls, path, config, dtype=torch.bfloat16, device="cuda"):
        with open(os.path.join(path, "config.json")) as f:
            kind = json.load(f)["mm_projector_type"]
        m = cls(kind, config, dtype=dtype, device=device)
        m.load_state_dict(_read_checkpoint(path))
        return m

    def save_pretrained(self, path, state_dict=None):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=2)
        sd = self.state_dict() if state_dict is None else state_dict
        save_file({k: v.detach().cpu().contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))


def build_vision_tower(path_or_name, config, dtype=torch.bfloat16, device="cuda"):
    """vila/model/multimodal_encoder/builder.py:10-53 for the two towers on the HALVA scripts (siglip, clip)."""
    if path_or_name is None:
        return None
    arch = str(path_or_name).lower()
    if os.path.isdir(str(path_or_name)):
        with open(os.path.join(path_or_name, "config.json")) as f:
            d = json.load(f)
        arch = (d.get("architectures") or [d.get("model_type", arch)])[0].lower()
    if getattr(config, "s2", False):
        raise NotImplementedError("S2 multi-scale towers are not on the HALVA path (s2 defaults to False)")
    if "siglip" in arch:
        vt = SiglipVisionTower(path_or_name, args=config, dtype=dtype, device=device)
    elif "clip" in arch:
        vt = CLIPVisionTower(path_or_name, args=config, dtype=dtype, device=device)
    else:
        raise ValueError("Unknown vision tower: %s" % path_or_name)
    config.mm_hidden_size = vt.hidden_size
    return vt


class CausalLMOutput(SimpleNamespace):
    pass


def _cpu(t):
    return t.detach().cpu() if isinstance(t, torch.Tensor) else torch.as_tensor(t)


# ------------------------------------------------------------------------------------------------
class VilaLlavaLlamaModel(nn.Module):
    """reference class name: vila.model.LlavaLlamaModel."""
    config_class = VilaConfig

    def __init__(self, config=None, llm=None, vision_tower=None, mm_projector=None, tokenizer=None, model_max_length=None,
                 dtype=torch.bfloat16, device="cuda", **kwargs):
        super().__init__()
        self.config = config
        self._last_plan = None
        self._use_lora = True
        if llm is None:                                            # init_vlm (llava_arch.py:57-86)
            cfgs = get_model_config(config)
            if len(cfgs) != 3:
                raise ValueError("`llm_cfg` `mm_projector_cfg` `vision_tower_cfg` not found in the config.")
            llm_path, vt_path, proj_path = cfgs
            llm = LlamaForCausalLM.from_pretrained(llm_path, dtype=dtype, device=device, model_max_length=model_max_length)
            config.hidden_size = llm.config.hidden_size
            tokenizer = _load_tokenizer(llm_path, model_max_length)
            vision_tower = build_vision_tower(vt_path, config, dtype=dtype, device=device)
            if config.resume_path and os.path.isdir(str(proj_path)):
                mm_projector = MultimodalProjector.from_pretrained(proj_path, config, dtype=dtype, device=device)
            else:
                mm_projector = MultimodalProjector(proj_path, config, dtype=dtype, device=device)
        self.llm, self.vision_tower, self.mm_projector, self.tokenizer = llm, vision_tower, mm_projector, tokenizer
        self.is_loaded = True
        self.post_config()

    @classmethod
    def from_pretrained(cls, path, *args, config=None, **kwargs):
        return cls.load_pretrained(path, *args, config=config, **kwargs)

    @classmethod
    def load_pretrained(cls, path_or_config, *args, **kwargs):
        kwargs.pop("config", None)
        config = VilaConfig.from_pretrained(path_or_config) if isinstance(path_or_config, str) else path_or_config
        return cls(config, *args, **kwargs)

    def post_config(self):                                         # llava_arch.py:193-201
        if getattr(self.config, "llm_cfg", None) is None:
            self.config.llm_cfg = self.llm.config
        if getattr(self.config, "vision_tower_cfg", None) is None and self.vision_tower is not None:
            self.config.vision_tower_cfg = self.vision_tower.config
        if getattr(self.config, "mm_projector_cfg", None) is None and self.mm_projector is not None:
            self.config.mm_projector_cfg = self.mm_projector.config

    def save_pretrained(self, output_dir, state_dict=None):        # llava_arch.py:131-176
        if getattr(self, "tokenizer", None) is not None and hasattr(self.tokenizer, "save_pretrained"):
            self.tokenizer.save_pretrained(os.path.join(output_dir, "llm"))
        self.llm.save_pretrained(os.path.join(output_dir, "llm"))
        self.config.llm_cfg = self.llm.config
        if self.mm_projector is not None:
            self.mm_projector.save_pretrained(os.path.join(output_dir, "mm_projector"))
            self.config.mm_projector_cfg = self.mm_projector.config
        self.config._name_or_path = output_dir
        self.config.architectures = ["LlavaLlamaModel"]
        self.config.save_pretrained(output_dir)

    # -- reference getters -----------------------------------------------------------------------
    def get_llm(self):
        return self.llm

    def get_lm_head(self):
        return getattr(self.llm, "lm_head", None)

    def get_vision_tower(self):
        return self.vision_tower

    def get_mm_projector(self):
        return self.mm_projector

    def get_input_embeddings(self):
        return self.llm.get_input_embeddings()

    def get_output_embeddings(self):
        return self.llm.get_output_embeddings()

    def freezed_module_patch(self):
        pass

    def encode_images(self, images):                               # llava_arch.py:216-219
        return self.mm_projector(self.vision_tower(images))

    def initialize_vision_tokenizer(self, model_args, tokenizer=None):
        if getattr(model_args, "mm_use_im_start_end", False):
            raise NotImplementedError("mm_use_im_start_end is False on the HALVA path (src_vila/halva_vila_13b.sh:48)")
        # mm_use_im_patch_token False (halva_vila_13b.sh:49): nothing to add (llava_arch.py:560-610 is a no-op then)

    # -- what the step engine needs (halva_amd/dpa.py) -------------------------------------------
    def get_model(self):
        return self.llm.model

    @property
    def lm_head(self):
        return self.llm.lm_head

    @property
    def device(self):
        return self.llm.lm_head.weight.device

    @property
    def dtype(self):
        return self.llm.lm_head.weight.dtype

    def dpa_spec(self):
        lc = self.llm.config
        return SimpleNamespace(n_patch=self.mm_projector.tokens_per_image(self.vision_tower.num_patches),
                               max_len=getattr(lc, "tokenizer_model_max_length", None),
                               padding_side=getattr(lc, "tokenizer_padding_side", "right"), imageless_consumes=False)

    def hidden_states(self, inputs_embeds, attention_mask=None, seq_start=None, seq_len=None, branch=None, rows=None):
        S, T, _ = inputs_embeds.shape
        dev = inputs_embeds.device
        if seq_len is None:
            if attention_mask is None:
                seq_start = torch.zeros(S, dtype=torch.int32)
                seq_len = torch.full((S,), T, dtype=torch.int32)
            elif self._last_plan is not None and self._last_plan.mask.shape == attention_mask.shape:
                seq_start, seq_len = self._last_plan.seq_start, self._last_plan.seq_len
            else:
                seq_start, seq_len = SP.spans_from_mask(_cpu(attention_mask))
        return self.llm.model.run_layers(inputs_embeds.to(torch.bfloat16), seq_start.to(dev), seq_len.to(dev), self._use_lora, branch, rows)

    # -- the splice (llava_arch.py:264-871) -------------------------------------------------------
    def _splice(self, input_ids, attention_mask, labels, signs, images):
        if type(images) is list:
            images = torch.cat([im if im.ndim == 4 else im[None] for im in images], dim=0)
        elif images.ndim == 5:
            images = images.flatten(0, 1)
        feats = self.encode_images(images)
        sp = self.dpa_spec()
        _, used = SP.image_slots(_cpu(input_ids), None if attention_mask is None else _cpu(attention_mask), False)
        if used > feats.shape[0]:
            raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (feats.shape[0], feats.shape[0]))
        plan = SP.plan_splice(_cpu(input_ids), None if attention_mask is None else _cpu(attention_mask),
                              None if labels is None else _cpu(labels), None if signs is None else _cpu(signs),
                              n_patch=feats.shape[1], max_len=sp.max_len, padding_side=sp.padding_side,
                              imageless_consumes=False)
        w = self.llm.model.embed_tokens.weight
        embeds = K.splice_rows(w, feats.to(torch.bfloat16), plan.src, plan.S, plan.T)
        self._last_plan = plan
        return embeds, plan

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images):
        if self.vision_tower is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        embeds, plan = self._splice(input_ids, attention_mask, labels, None, images)
        dev = input_ids.device
        new_mask = None if attention_mask is None else plan.mask.to(dev).to(attention_mask.dtype)
        return (None, position_ids, new_mask, past_key_values, embeds, None if labels is None else plan.labels.to(dev))

    def prepare_inputs_labels_for_multimodal_signed(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                                    images, signs):
        if self.vision_tower is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels, signs
        embeds, plan = self._splice(input_ids, attention_mask, labels, signs, images)
        dev = input_ids.device
        new_mask = None if attention_mask is None else plan.mask.to(dev).to(attention_mask.dtype)
        return (None, position_ids, new_mask, past_key_values, embeds, None if labels is None else plan.labels.to(dev),
                None if signs is None else plan.signs.to(dev))

    # -- forward (llava_llama.py:83-177) ----------------------------------------------------------
    def forward(self, input_ids=None, images=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, signs=None):
        if past_key_values is not None or use_cache:
            raise NotImplementedError("KV-cache decoding is not part of the DPA training path")
        if inputs_embeds is None:
            if signs is not None:
                (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels,
                 signs) = self.prepare_inputs_labels_for_multimodal_signed(input_ids, position_ids, attention_mask,
                                                                           past_key_values, labels, images, signs)
            else:
                (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds,
                 labels) = self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                                     labels, images)
            if inputs_embeds is None:
                inputs_embeds = self.llm.model.embed_tokens(input_ids)
        h = self.hidden_states(inputs_embeds, attention_mask)
        logits = torch.nn.functional.linear(h, self.llm.lm_head.weight).float()
        loss = None
        if labels is not None:                                     # the CE the reference computes and never uses
            tgt = labels[..., 1:].contiguous().view(-1)
            keep = (tgt != IGNORE_INDEX).nonzero().flatten()
            if keep.numel():
                lg = logits[..., :-1, :].reshape(-1, logits.shape[-1])
                loss = -K.token_logp(lg[keep].contiguous(), tgt[keep].int()).mean()
        return CausalLMOutput(loss=loss, logits=logits, past_key_values=None, hidden_states=None, attentions=None,
                              labels=labels, signs=signs)


def _load_tokenizer(llm_path, model_max_length):
    """language_model/builder.py:83-114: slow Llama tokenizer, right padding, legacy=False."""
    try:
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(llm_path, model_max_length=model_max_length, padding_side="right", use_fast=False,
                                             legacy=False)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------
class _BaseOnlyLayer(nn.Module):
    def __init__(self, layer):
        super().__init__()
        self._l = [layer]

    def forward(self, x, info, use_lora=False, own_x=False, rows=None):
        return self._l[0](x, info, False, own_x, rows)


class _FrozenProjectorView(nn.Module):
    def __init__(self, proj):
        super().__init__()
        self._p = [proj]
        self.kind, self.config = proj.kind, proj.config

    def tokens_per_image(self, n):
        return self._p[0].tokens_per_image(n)

    def forward(self, x, *a, **k):
        with torch.no_grad():
            return self._p[0](x)


def build_random_vila(llm_kwargs, vision_kwargs, projector="mlp_downsample", tower="siglip", lora_r=0, lora_alpha=0, seed=0,
                      device="cuda", max_len=4096, std=0.02, share_base_from=None):
    """Random-init VILA of a given geometry (no checkpoints offline).  share_base_from: the policy whose frozen base
    tensors the reference model reuses (reference model == base of the policy, train_halva.py:1313-1319)."""
    dtype = torch.bfloat16
    cfg = VilaConfig(mm_vision_select_layer=-2, mm_vision_select_feature="cls_patch", mm_projector_lr=None)
    cfg.mm_hidden_size = vision_kwargs["hidden_size"]
    cfg.hidden_size = llm_kwargs["hidden_size"]
    g = torch.Generator(device=device).manual_seed(seed)
    if share_base_from is None:
        lc = LlamaConfig(**llm_kwargs)
        llm = LlamaForCausalLM(lc, dtype, device)
        if tower == "siglip":
            vt = SiglipVisionTower("random-siglip", args=cfg, delay_load=True, config=SiglipVisionConfig(**vision_kwargs),
                                   dtype=dtype, device=device)
        else:
            vt = CLIPVisionTower("random-clip", args=cfg, delay_load=True, config=CLIPVisionConfig(**vision_kwargs), dtype=dtype,
                                 device=device)
        vt._alloc()
        proj = MultimodalProjector(projector, cfg, dtype=dtype, device=device)
        with torch.no_grad():
            for mod in (llm, vt, proj):
                for n, p in mod.named_parameters():
                    if p.ndim >= 2:
                        p.normal_(0.0, std, generator=g)
                    elif "ln" in n or "norm" in n or (mod is proj and n.startswith("layers.1.") and projector == "mlp_downsample"):
                        p.zero_() if (n.endswith("_b") or n.endswith("bias")) else p.fill_(1.0)
                    else:
                        p.normal_(0.0, std, generator=g)
            vt.patch_w[:, 3 * vision_kwargs["patch_size"] ** 2:].zero_()
            if tower == "siglip" and vt.head_pad != vt.head_dim:                     # keep the pad lanes exactly zero
                H, D, Dp, d = vt._cfg.num_attention_heads, vt.head_dim, vt.head_pad, vt._cfg.hidden_size
                for L in vt.layers:
                    L.qkv_w.view(3, H, Dp, d)[:, :, D:].zero_()
                    L.qkv_b.view(3, H, Dp)[:, :, D:].zero_()
                    L.out_w.view(d, H, Dp)[:, :, D:].zero_()
        vt.requires_grad_(False)
        vt.is_loaded = True
        m = VilaLlavaLlamaModel(cfg, llm=llm, vision_tower=vt, mm_projector=proj, dtype=dtype, device=device)
    else:
        src = share_base_from
        lc = LlamaConfig(**dict(llm_kwargs, num_hidden_layers=0))
        llm = LlamaForCausalLM.__new__(LlamaForCausalLM)
        nn.Module.__init__(llm)
        llm.config = src.llm.config
        llm.model = LlamaModel(lc, dtype, device)
        llm.model.config = src.llm.config
        llm.model.embed_tokens = src.llm.model.embed_tokens
        llm.model.norm = src.llm.model.norm
        llm.model.layers = nn.ModuleList([_BaseOnlyLayer(l) for l in src.llm.model.layers])
        llm.vocab_size = src.llm.vocab_size
        llm.lm_head = src.llm.lm_head
        llm.pad_token_id = None
        m = VilaLlavaLlamaModel(cfg, llm=llm, vision_tower=src.vision_tower, mm_projector=_FrozenProjectorView(src.mm_projector),
                                dtype=dtype, device=device)
        m._use_lora = False
    m.llm.config.tokenizer_model_max_length = max_len
    m.llm.config.tokenizer_padding_side = "right"
    for p in m.parameters():
        p.requires_grad_(False)
    if lora_r:
        add_lora(m.llm, lora_r, lora_alpha, g)
        for p in m.mm_projector.parameters():
            p.requires_grad_(True)
    return m
