"""LlavaLlamaForCausalLM for the DPA step (MI355X-native), keeping the reference's API surface:

reference llava/model/language_model/llava_llama.py:16-97 (LlavaConfig, LlavaLlamaModel, LlavaLlamaForCausalLM)
reference llava/model/llava_arch.py:13-440 (LlavaMetaModel, LlavaMetaForCausalLM: encode_images,
prepare_inputs_labels_for_multimodal[_signed], initialize_vision_modules / _tokenizer).

The splice is a host-computed index plan + one gather launch (halva_amd/splice.py); logits for the loss are
normally never materialised (halva_amd/dpa.py fuses lm_head with the loss kernels), `forward()` still returns
full fp32 logits for API parity with `LlamaForCausalLM.forward(...).logits.float()` (modelling_llama.py:806).
"""
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import kernels as K
from . import splice as SP
from .clip import CLIPVisionTower, _read_checkpoint, build_vision_projector, build_vision_tower
from .llama import (LlamaConfig, LlamaModel, add_lora, hf_llama_state_dict, load_hf_llama_weights)

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200


class LlavaConfig(LlamaConfig):
    model_type = "llava"


class LlavaMetaModel:
    """reference llava/model/llava_arch.py:13-68"""

    def get_vision_tower(self):
        vt = getattr(self, "vision_tower", None)
        if type(vt) is list:
            vt = vt[0]
        return vt

    def initialize_vision_modules(self, model_args, fsdp=None):
        vision_tower = model_args.vision_tower
        self.config.mm_vision_tower = vision_tower
        if self.get_vision_tower() is None:
            vt = build_vision_tower(model_args, dtype=self._dtype, device=self._device)
            self.vision_tower = [vt] if fsdp else vt
        else:
            vt = self.get_vision_tower()
            vt.load_model()
        self.config.use_mm_proj = True
        self.config.mm_projector_type = getattr(model_args, "mm_projector_type", "linear")
        self.config.mm_hidden_size = vt.hidden_size
        self.config.mm_vision_select_layer = model_args.mm_vision_select_layer
        self.config.mm_vision_select_feature = getattr(model_args, "mm_vision_select_feature", "patch")
        if getattr(self, "mm_projector", None) is None:
            self.mm_projector = build_vision_projector(self.config, dtype=self._dtype, device=self._device)
        else:
            for p in self.mm_projector.parameters():          # "In case it is frozen by LoRA" (llava_arch.py:58-61)
                p.requires_grad = True
        ckpt = getattr(model_args, "pretrain_mm_mlp_adapter", None)
        if ckpt is not None:
            w = torch.load(ckpt, map_location="cpu")
            self.mm_projector.load_state_dict({k.split("mm_projector.")[1]: v for k, v in w.items() if "mm_projector" in k})


class LlavaLlamaModel(LlavaMetaModel, LlamaModel):
    config_class = LlavaConfig

    def __init__(self, config, dtype=torch.bfloat16, device="cuda"):
        LlamaModel.__init__(self, config, dtype, device)
        self._dtype, self._device = dtype, device
        if hasattr(config, "mm_vision_tower"):
            self.vision_tower = build_vision_tower(config, delay_load=True, dtype=dtype, device=device)
            self.mm_projector = build_vision_projector(config, dtype=dtype, device=device)


class CausalLMOutput(SimpleNamespace):
    pass


class LlavaMetaForCausalLM:
    """reference llava/model/llava_arch.py:71-440"""

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def encode_images(self, images):
        feats = self.get_model().get_vision_tower()(images)
        return self.get_model().mm_projector(feats)

    # -- the splice ------------------------------------------------------------------------------
    def _splice(self, input_ids, attention_mask, labels, signs, images, image_features=None, image_map=None):
        model = self.get_model()
        dev = model.embed_tokens.weight.device
        if image_features is None:
            if type(images) is list or images.ndim == 5:
                cat = torch.cat([im for im in images], dim=0)
                image_features = self.encode_images(cat)
                # multi-image samples: features are consumed in order, one [n_patch, d] block per image token
            else:
                image_features = self.encode_images(images)
        n_patch = image_features.shape[1]
        plan = SP.plan_splice(_cpu(input_ids), None if attention_mask is None else _cpu(attention_mask),
                              None if labels is None else _cpu(labels), None if signs is None else _cpu(signs),
                              n_patch=n_patch, max_len=getattr(self.config, "tokenizer_model_max_length", None),
                              padding_side=getattr(self.config, "tokenizer_padding_side", "right"), image_map=image_map)
        embeds = K.splice_rows(model.embed_tokens.weight, image_features.to(torch.bfloat16), plan.src, plan.S, plan.T)
        return embeds, plan

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images):
        vt = self.get_vision_tower()
        if vt is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        embeds, plan = self._splice(input_ids, attention_mask, labels, None, images)
        dev = input_ids.device
        new_labels = None if labels is None else plan.labels.to(dev)
        new_mask = None if attention_mask is None else plan.mask.to(dev).to(attention_mask.dtype)
        self._last_plan = plan
        return None, (None if position_ids is None else position_ids), new_mask, past_key_values, embeds, new_labels

    def prepare_inputs_labels_for_multimodal_signed(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                                    images, signs):
        vt = self.get_vision_tower()
        if vt is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels, signs
        embeds, plan = self._splice(input_ids, attention_mask, labels, signs, images)
        dev = input_ids.device
        new_labels = None if labels is None else plan.labels.to(dev)
        new_signs = None if signs is None else plan.signs.to(dev)
        new_mask = None if attention_mask is None else plan.mask.to(dev).to(attention_mask.dtype)
        self._last_plan = plan
        return None, (None if position_ids is None else position_ids), new_mask, past_key_values, embeds, new_labels, new_signs

    def resize_token_embeddings(self, new_num_tokens):
        """HF PreTrainedModel.resize_token_embeddings for this model's two vocabulary-sized tensors: old rows kept, new rows
        N(0, initializer_range) as HF initialises them (the callers below overwrite them with the mean of the old rows)."""
        emb, head = self.get_input_embeddings(), self.get_output_embeddings()
        old = emb.weight.shape[0]
        if new_num_tokens == old:
            return emb
        std = getattr(self.config, "initializer_range", 0.02)
        for mod in (emb, head):
            w = mod.weight.data
            nw = torch.empty(new_num_tokens, w.shape[1], dtype=w.dtype, device=w.device).normal_(0.0, std)
            n = min(old, new_num_tokens)
            nw[:n] = w[:n]
            mod.weight = nn.Parameter(nw, requires_grad=mod.weight.requires_grad)
        if hasattr(emb, "num_embeddings"):
            emb.num_embeddings = new_num_tokens
        if hasattr(head, "out_features"):
            head.out_features = new_num_tokens
        self.config.vocab_size = new_num_tokens
        return emb

    def initialize_vision_tokenizer(self, model_args, tokenizer):
        """reference llava/model/llava_arch.py:398-440: <im_patch> / <im_start>, <im_end> tokens appended to the vocabulary, the new
        rows of embed_tokens and lm_head set to the mean of the old ones (or taken from --pretrain_mm_mlp_adapter).  The HALVA
        scripts pass both flags False (src/hallava_7b.sh:42-43), in which case this is a no-op, as in the reference.  Training the
        embedding matrix itself (tune_mm_mlp_adapter + mm_use_im_start_end) is not on the DPA path and is refused."""
        from llava.constants import DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN
        if getattr(model_args, "mm_use_im_patch_token", False):
            tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
        if getattr(model_args, "mm_use_im_start_end", False):
            if getattr(model_args, "tune_mm_mlp_adapter", False):
                raise NotImplementedError("tune_mm_mlp_adapter with mm_use_im_start_end trains embed_tokens (llava_arch.py:419-423): "
                                          "the MI355X DPA engine keeps the embedding matrix frozen")
            num_new = tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
            if num_new > 0:
                with torch.no_grad():
                    for w in (self.get_input_embeddings().weight, self.get_output_embeddings().weight):
                        w[-num_new:] = w[:-num_new].float().mean(dim=0, keepdim=True).to(w.dtype)
            pre = getattr(model_args, "pretrain_mm_mlp_adapter", None)
            if pre:
                ew = torch.load(pre, map_location="cpu")["model.embed_tokens.weight"]
                assert num_new == 2
                inp = self.get_input_embeddings().weight
                with torch.no_grad():
                    if inp.shape == ew.shape:
                        inp[-num_new:] = ew[-num_new:].to(inp)
                    elif ew.shape[0] == num_new:
                        inp[-num_new:] = ew.to(inp)
                    else:
                        raise ValueError("Unexpected embed_tokens_weight shape. Pretrained: %s. Current: %s. Numer of new tokens: %d."
                                         % (tuple(ew.shape), tuple(inp.shape), num_new))
        # (mm_use_im_patch_token alone only touches requires_grad flags that are already False here: llava_arch.py:435-440)


def _cpu(t):
    return t.detach().cpu() if isinstance(t, torch.Tensor) else torch.as_tensor(t)


class LlavaLlamaForCausalLM(nn.Module, LlavaMetaForCausalLM):
    config_class = LlavaConfig

    def __init__(self, config, dtype=torch.bfloat16, device="cuda"):
        nn.Module.__init__(self)
        self.config = config
        self.model = LlavaLlamaModel(config, dtype, device)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False, dtype=dtype, device=device)
        self.lm_head.weight.requires_grad_(False)
        self._use_lora = True
        self._last_plan = None

    # -- reference surface -----------------------------------------------------------------------
    def get_model(self):
        return self.model

    @property
    def device(self):
        return self.lm_head.weight.device

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def enable_input_require_grads(self):
        pass                                  # inputs_embeds carries grad through the projector already

    def gradient_checkpointing_enable(self, *a, **k):
        self.model.gradient_checkpointing = True

    @classmethod
    def from_pretrained(cls, path, cache_dir=None, dtype=torch.bfloat16, device="cuda", **kw):
        """Load an HF LLaVA checkpoint directory (config.json + safetensors / .bin shards)."""
        if not os.path.isdir(path):
            raise FileNotFoundError("%s is not a local checkpoint directory (no network on this path; HF hub ids must be "
                                    "downloaded beforehand)" % path)
        cfg = LlavaConfig.from_pretrained(path)
        m = cls(cfg, dtype=dtype, device=device)
        sd = _read_checkpoint(path)
        load_hf_llama_weights(m, sd, strict=True)
        proj = {k.split("mm_projector.")[1]: v for k, v in sd.items() if "mm_projector." in k}
        if proj and getattr(m.model, "mm_projector", None) is not None:
            m.model.mm_projector.load_state_dict(proj)
        return m

    def hf_state_dict(self):
        sd = hf_llama_state_dict(self)
        if getattr(self.model, "mm_projector", None) is not None:
            for k, v in self.model.mm_projector.state_dict().items():
                sd["model.mm_projector." + k] = v
        return sd

    # -- forward ---------------------------------------------------------------------------------
    def hidden_states(self, inputs_embeds, attention_mask=None, seq_start=None, seq_len=None, branch=None, rows=None):
        """Decoder stack + final norm on inputs_embeds [S, T, d].  The key-padding mask must be one contiguous run per
        row (what the splice produces); it is the raw [S, T] bool mask of the flash-attn seam
        (llama_flash_attn_monkey_patch.py:71,98-102)."""
        S, T, _ = inputs_embeds.shape
        dev = inputs_embeds.device
        if seq_len is None:
            if attention_mask is None:
                seq_start = torch.zeros(S, dtype=torch.int32)
                seq_len = torch.full((S,), T, dtype=torch.int32)
            elif self._last_plan is not None and self._last_plan.mask.shape == attention_mask.shape:
                seq_start, seq_len = self._last_plan.seq_start, self._last_plan.seq_len     # no device sync
            else:
                seq_start, seq_len = SP.spans_from_mask(_cpu(attention_mask))
        return self.model.run_layers(inputs_embeds.to(torch.bfloat16), seq_start.to(dev), seq_len.to(dev), self._use_lora, branch, rows)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None):
        if past_key_values is not None or use_cache:
            raise NotImplementedError("KV-cache decoding is not part of the DPA training path")
        if inputs_embeds is None:
            if images is not None:
                (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds,
                 labels) = self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                                     labels, images)
            if inputs_embeds is None:
                inputs_embeds = self.model.embed_tokens(input_ids)
        h = self.hidden_states(inputs_embeds, attention_mask)
        logits = torch.nn.functional.linear(h, self.lm_head.weight).float()
        loss = None
        if labels is not None:
            tgt = labels[..., 1:].contiguous().view(-1)
            keep = (tgt != IGNORE_INDEX).nonzero().flatten()
            lg = logits[..., :-1, :].reshape(-1, logits.shape[-1])
            lp = K.token_logp(lg[keep].contiguous(), tgt[keep].int())
            loss = -lp.mean()
        return CausalLMOutput(loss=loss, logits=logits, past_key_values=None, hidden_states=None, attentions=None)

    __call__ = nn.Module.__call__


def build_random_llava(cfg_kwargs, clip_kwargs, lora_r=0, lora_alpha=0, seed=0, device="cuda", max_len=2048, std=0.02,
                       share_base_from=None):
    """Random-init LLaVA of a given geometry (no checkpoints exist offline): weights N(0, std), norms 1.
    share_base_from: another model whose frozen base tensors are reused (the reference model == base of the policy)."""
    from .clip import CLIPVisionConfig
    cfg = LlavaConfig(**cfg_kwargs)
    cfg.mm_vision_tower = "random-clip"
    cfg.mm_projector_type = "mlp2x_gelu"
    cfg.mm_hidden_size = clip_kwargs["hidden_size"]
    cfg.mm_vision_select_layer = -2
    cfg.mm_vision_select_feature = "patch"
    cfg.tokenizer_model_max_length = max_len
    cfg.tokenizer_padding_side = "right"
    g = torch.Generator(device=device).manual_seed(seed)
    if share_base_from is None:
        m = LlavaLlamaForCausalLM.__new__(LlavaLlamaForCausalLM)
        nn.Module.__init__(m)
        m.config = cfg
        m.model = LlavaLlamaModel.__new__(LlavaLlamaModel)
        LlamaModel.__init__(m.model, cfg, torch.bfloat16, device)
        m.model._dtype, m.model._device = torch.bfloat16, device
        m.model.vision_tower = CLIPVisionTower("random-clip", args=cfg, delay_load=True, config=CLIPVisionConfig(**clip_kwargs),
                                               dtype=torch.bfloat16, device=device)
        m.model.vision_tower._alloc()
        m.model.mm_projector = build_vision_projector(cfg, dtype=torch.bfloat16, device=device)
        m.vocab_size = cfg.vocab_size
        m.lm_head = nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False, dtype=torch.bfloat16, device=device)
        m._use_lora, m._last_plan = True, None
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.ndim >= 2:
                    p.normal_(0.0, std, generator=g)
                elif "ln" in n or "norm" in n:
                    if n.endswith("_b") or n.endswith("bias"):
                        p.zero_()
                    else:
                        p.fill_(1.0)
                else:
                    p.normal_(0.0, std, generator=g)
            vt = m.model.vision_tower
            vt.patch_w[:, 3 * clip_kwargs["patch_size"] ** 2:].zero_()
        m.model.vision_tower.requires_grad_(False)
        m.model.vision_tower.is_loaded = True
    else:
        src = share_base_from
        m = LlavaLlamaForCausalLM.__new__(LlavaLlamaForCausalLM)
        nn.Module.__init__(m)
        m.config = cfg
        m.model = LlavaLlamaModel.__new__(LlavaLlamaModel)
        LlamaModel.__init__(m.model, LlavaConfig(**dict(cfg_kwargs, num_hidden_layers=0)), torch.bfloat16, device)
        m.model.config = cfg
        m.model._dtype, m.model._device = torch.bfloat16, device
        m.model.embed_tokens = src.model.embed_tokens
        m.model.norm = src.model.norm
        m.model.vision_tower = src.model.vision_tower
        m.model.mm_projector = _FrozenProjectorView(src.model.mm_projector)
        m.model.layers = nn.ModuleList([_BaseOnlyLayer(l) for l in src.model.layers])
        m.vocab_size = cfg.vocab_size
        m.lm_head = src.lm_head
        m._use_lora, m._last_plan = False, None
    for p in m.parameters():
        p.requires_grad_(False)
    if lora_r:
        add_lora(m, lora_r, lora_alpha, g)
        for p in m.model.mm_projector.parameters():
            p.requires_grad_(True)
    return m


class _BaseOnlyLayer(nn.Module):
    """A decoder layer that reuses another layer's frozen base tensors and never applies its LoRA factors."""

    def __init__(self, layer):
        super().__init__()
        self._l = [layer]            # not registered: the tensors belong to the policy model

    def forward(self, x, info, use_lora=False, own_x=False, rows=None):
        return self._l[0](x, info, False, own_x, rows)


class _FrozenProjectorView(nn.Module):
    def __init__(self, proj):
        super().__init__()
        self._p = [proj]

    def forward(self, x):
        with torch.no_grad():
            return self._p[0](x)
