"""Frozen SigLIP vision tower of the VILA path (google/siglip-so400m-patch14-384 geometry).

Stand-in for reference vila/model/multimodal_encoder/siglip_encoder.py:12-20 (SiglipVisionTower around the vendored
SiglipVisionModel, vila/model/multimodal_encoder/siglip/modeling_siglip.py:246-449,826-879) as used through
VisionTower.forward / feature_select (vila/model/multimodal_encoder/vision_encoder.py:23-32,121-140):
`hidden_states[select_layer]` of a no-CLS ViT, all tokens kept (`cls_patch` == `patch` here: SigLIP has no class
token, src_vila/halva_vila_13b.sh:42).

Differences from the CLIP tower (halva_amd/clip.py): biased 'valid' patch conv (384 = 27*14 + 6 -> 27x27 tokens), no
class embedding, no pre-LayerNorm, tanh GELU, eps 1e-6 and 72-wide heads.  The attention kernel has 64/128-wide
tiles, so every head runs zero-padded to 128 lanes: the q/k/v projection rows and out_proj columns of the pad
lanes are zero (scores and outputs are unchanged), with the softmax scale kept at 72**-0.5.
"""
import json
import os
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from .clip import CLIPVisionTower, _read_checkpoint


class SiglipVisionConfig:
    _defaults = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                     image_size=384, patch_size=14, num_channels=3, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)

    def __init__(self, **kw):
        for k, v in self._defaults.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d = d.get("vision_config", d)
        return cls(**{k: v for k, v in d.items() if k in cls._defaults})

    def to_dict(self):
        return dict({k: getattr(self, k) for k in self._defaults}, model_type="siglip_vision_model")


def _pad_head_dim(D):
    for w in (64, 128):
        if D <= w:
            return w
    raise ValueError("head_dim %d > 128 is not supported by the attention kernel" % D)


class _SiglipLayer(nn.Module):
    def __init__(self, cfg, Dp, dtype, device):
        super().__init__()
        d, f, H = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
        mk = lambda *s: nn.Parameter(torch.zeros(*s, dtype=dtype, device=device), requires_grad=False)
        self.ln1_w, self.ln1_b, self.ln2_w, self.ln2_b = mk(d), mk(d), mk(d), mk(d)
        self.qkv_w, self.qkv_b = mk(3 * H * Dp, d), mk(3 * H * Dp)     # fused [q;k;v], heads padded to Dp lanes
        self.out_w, self.out_b = mk(d, H * Dp), mk(d)
        self.fc1_w, self.fc1_b, self.fc2_w, self.fc2_b = mk(f, d), mk(f), mk(d, f), mk(d)


class SiglipVisionTower(CLIPVisionTower):
    """Same surface as the reference class: forward(images) -> features of hidden_states[select_layer]."""

    def __init__(self, vision_tower, args=None, delay_load=False, config=None, dtype=torch.bfloat16, device="cuda"):
        if config is None and vision_tower and os.path.isdir(str(vision_tower)):
            config = SiglipVisionConfig.from_pretrained(vision_tower)
        if config is None:
            config = SiglipVisionConfig()
        super().__init__(vision_tower, args=args, delay_load=delay_load, config=config, dtype=dtype, device=device)
        if args is None or not hasattr(args, "mm_vision_select_feature"):
            self.select_feature = "cls_patch"

    def _alloc(self):
        cfg, dtype, device = self._cfg, self._dtype, self._device
        d, p = cfg.hidden_size, cfg.patch_size
        self.head_dim = d // cfg.num_attention_heads
        self.head_pad = _pad_head_dim(self.head_dim)
        self.kp = (3 * p * p + 7) // 8 * 8
        mk = lambda *s: nn.Parameter(torch.zeros(*s, dtype=dtype, device=device), requires_grad=False)
        self.patch_w, self.patch_b = mk(d, self.kp), mk(d)
        self.position_embedding = mk((cfg.image_size // p) ** 2, d)
        self.layers = nn.ModuleList([_SiglipLayer(cfg, self.head_pad, dtype, device) for _ in range(cfg.num_hidden_layers)])

    def _load_processor(self):
        try:
            from transformers import SiglipImageProcessor
            self.image_processor = SiglipImageProcessor.from_pretrained(self.vision_tower_name)
        except Exception:
            self.image_processor = None

    def load_model(self, state_dict=None):
        if not hasattr(self, "layers"):
            self._alloc()
        if state_dict is None and self.vision_tower_name and os.path.isdir(str(self.vision_tower_name)):
            state_dict = _read_checkpoint(self.vision_tower_name)
            self._load_processor()
        if state_dict is not None:
            self.load_hf_state_dict(state_dict)
        self.requires_grad_(False)
        self.is_loaded = True

    def load_hf_state_dict(self, sd):
        """HF / vendored SiglipVisionModel names, with or without the `vision_model.` prefix.  The pooling head and
        post_layernorm are not on the path (`hidden_states[-2]` is taken before them) and are ignored."""
        sd = {re.sub(r"^(vision_tower\.)*(vision_model\.)*", "", k): v for k, v in sd.items()}
        cfg = self._cfg
        d, H, D, Dp = cfg.hidden_size, cfg.num_attention_heads, self.head_dim, self.head_pad
        with torch.no_grad():
            w = sd["embeddings.patch_embedding.weight"]
            self.patch_w.zero_()
            self.patch_w[:, :w[0].numel()].copy_(w.reshape(w.shape[0], -1))
            self.patch_b.copy_(sd["embeddings.patch_embedding.bias"])
            self.position_embedding.copy_(sd["embeddings.position_embedding.weight"])
            for i, L in enumerate(self.layers):
                p = "encoder.layers.%d." % i
                if p + "layer_norm1.weight" not in sd:
                    continue
                L.ln1_w.copy_(sd[p + "layer_norm1.weight"]), L.ln1_b.copy_(sd[p + "layer_norm1.bias"])
                L.ln2_w.copy_(sd[p + "layer_norm2.weight"]), L.ln2_b.copy_(sd[p + "layer_norm2.bias"])
                L.qkv_w.zero_(), L.qkv_b.zero_(), L.out_w.zero_()
                for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    wv = L.qkv_w[j * H * Dp:(j + 1) * H * Dp].view(H, Dp, d)
                    wv[:, :D].copy_(sd[p + "self_attn.%s.weight" % n].view(H, D, d))
                    bv = L.qkv_b[j * H * Dp:(j + 1) * H * Dp].view(H, Dp)
                    bv[:, :D].copy_(sd[p + "self_attn.%s.bias" % n].view(H, D))
                L.out_w.view(d, H, Dp)[:, :, :D].copy_(sd[p + "self_attn.out_proj.weight"].view(d, H, D))
                L.out_b.copy_(sd[p + "self_attn.out_proj.bias"])
                L.fc1_w.copy_(sd[p + "mlp.fc1.weight"]), L.fc1_b.copy_(sd[p + "mlp.fc1.bias"])
                L.fc2_w.copy_(sd[p + "mlp.fc2.weight"]), L.fc2_b.copy_(sd[p + "mlp.fc2.bias"])

    def _features(self, images):
        cfg = self._cfg
        d, H = cfg.hidden_size, cfg.num_attention_heads
        eps = cfg.layer_norm_eps
        act = cfg.hidden_act
        x = images.to(device=self.device, dtype=self.dtype).contiguous()
        x = K.vit_patch_embed(x, self.patch_w, self.patch_b, cfg.patch_size, d) + self.position_embedding[None]
        n_run = cfg.num_hidden_layers + 1 + self.select_layer if self.select_layer < 0 else self.select_layer
        scale = float(self.head_dim) ** -0.5
        for L in list(self.layers)[:n_run]:
            h = K.layernorm(x, L.ln1_w, L.ln1_b, eps)
            qkv = F.linear(h, L.qkv_w, L.qkv_b)
            a = K.sdpa_full(qkv, H, self.head_pad, scale)
            x = x + F.linear(a, L.out_w, L.out_b)
            h = K.layernorm(x, L.ln2_w, L.ln2_b, eps)
            h = F.linear(h, L.fc1_w, L.fc1_b)
            if act == "gelu_pytorch_tanh":
                h = F.gelu(h, approximate="tanh")
            elif act == "gelu":
                h = F.gelu(h)
            elif act == "quick_gelu":
                h = h * torch.sigmoid(1.702 * h)
            else:
                raise ValueError("unsupported SigLIP activation %s" % act)
            x = x + F.linear(h, L.fc2_w, L.fc2_b)
        if self.select_feature in ("patch", "cls_patch"):
            # vision_encoder.py:26-29: "patch" drops token 0, "cls_patch" keeps everything.  SigLIP has no class token,
            # so "patch" really drops the first patch - kept bug-for-bug.
            return x[:, 1:] if self.select_feature == "patch" else x
        raise ValueError("Unexpected select feature: %s" % self.select_feature)


def build_siglip_tower(name, args=None, **kwargs):
    return SiglipVisionTower(name, args=args, **kwargs)
