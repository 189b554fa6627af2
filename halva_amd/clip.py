"""Frozen CLIP ViT vision tower + mm_projector for the DPA step.

Stand-in for reference llava/model/multimodal_encoder/clip_encoder.py (CLIPVisionTower around HF CLIPVisionModel) and
llava/model/multimodal_projector/builder.py (mlp2x_gelu).  The tower runs under no_grad exactly like the reference
(clip_encoder.py:37); only the layers that feed `hidden_states[select_layer]` are executed (HF computes all 24 and
the post-LN and throws them away).  Hand-written kernels: patch-embed (im2col + MFMA GEMM), attention (MFMA,
head_dim 64), LayerNorm (wave-per-row kernel), projector MLP (MFMA GEMM, fused bias+GELU epilogue, fwd + bwd).
The four biased linears / quick_gelu of the frozen blocks go through PyTorch-ROCm (hipBLASLt).
"""
import json
import os
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K


class CLIPVisionConfig:
    _defaults = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                     image_size=336, patch_size=14, num_channels=3, hidden_act="quick_gelu", layer_norm_eps=1e-5)

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
        return {k: getattr(self, k) for k in self._defaults}


class _ClipLayer(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        d, f = cfg.hidden_size, cfg.intermediate_size
        mk = lambda *s: nn.Parameter(torch.zeros(*s, dtype=dtype, device=device), requires_grad=False)
        self.ln1_w, self.ln1_b, self.ln2_w, self.ln2_b = mk(d), mk(d), mk(d), mk(d)
        self.qkv_w, self.qkv_b = mk(3 * d, d), mk(3 * d)          # fused [q;k;v]
        self.out_w, self.out_b = mk(d, d), mk(d)
        self.fc1_w, self.fc1_b, self.fc2_w, self.fc2_b = mk(f, d), mk(f), mk(d, f), mk(d)


class CLIPVisionTower(nn.Module):
    """Same surface as the reference class: forward(images) -> patch features of hidden_states[select_layer],
    .hidden_size, .num_patches, .dtype, .device, .config, .is_loaded, .load_model()."""

    def __init__(self, vision_tower, args=None, delay_load=False, config=None, dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        self.vision_tower_name = vision_tower
        self.select_layer = getattr(args, "mm_vision_select_layer", -2) if args is not None else -2
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch") if args is not None else "patch"
        self.is_loaded = False
        self._cfg = config
        self._dtype, self._device = dtype, device
        self.image_processor = None
        if config is None and vision_tower and os.path.isdir(str(vision_tower)):
            self._cfg = CLIPVisionConfig.from_pretrained(vision_tower)
        if self._cfg is None:
            self._cfg = CLIPVisionConfig()              # openai/clip-vit-large-patch14-336 geometry
        if not delay_load:
            self.load_model()

    # -- construction ----------------------------------------------------------------------------
    def _alloc(self):
        cfg, dtype, device = self._cfg, self._dtype, self._device
        d, p = cfg.hidden_size, cfg.patch_size
        self.kp = (3 * p * p + 7) // 8 * 8
        mk = lambda *s: nn.Parameter(torch.zeros(*s, dtype=dtype, device=device), requires_grad=False)
        self.patch_w = mk(d, self.kp)                              # conv weight flattened, zero padded to a multiple of 8
        self.class_embedding = mk(d)
        self.position_embedding = mk((cfg.image_size // p) ** 2 + 1, d)
        self.pre_ln_w, self.pre_ln_b = mk(d), mk(d)
        self.layers = nn.ModuleList([_ClipLayer(cfg, dtype, device) for _ in range(cfg.num_hidden_layers)])

    def load_model(self, state_dict=None):
        if not hasattr(self, "layers"):
            self._alloc()
        if state_dict is None and self.vision_tower_name and os.path.isdir(str(self.vision_tower_name)):
            state_dict = _read_checkpoint(self.vision_tower_name)
            try:
                from transformers import CLIPImageProcessor
                self.image_processor = CLIPImageProcessor.from_pretrained(self.vision_tower_name)
            except Exception:
                self.image_processor = None
        if state_dict is not None:
            self.load_hf_state_dict(state_dict)
        self.requires_grad_(False)
        self.is_loaded = True

    def load_hf_state_dict(self, sd):
        """HF CLIPVisionModel names, with or without the `vision_model.` prefix."""
        sd = {re.sub(r"^(vision_tower\.)?(vision_model\.)?", "", k): v for k, v in sd.items()}
        cfg = self._cfg
        with torch.no_grad():
            w = sd["embeddings.patch_embedding.weight"]
            self.patch_w.zero_()
            self.patch_w[:, :w[0].numel()].copy_(w.reshape(w.shape[0], -1))
            self.class_embedding.copy_(sd["embeddings.class_embedding"])
            self.position_embedding.copy_(sd["embeddings.position_embedding.weight"])
            self.pre_ln_w.copy_(sd["pre_layrnorm.weight"])
            self.pre_ln_b.copy_(sd["pre_layrnorm.bias"])
            d = cfg.hidden_size
            for i, L in enumerate(self.layers):
                p = "encoder.layers.%d." % i
                if p + "layer_norm1.weight" not in sd:
                    continue
                L.ln1_w.copy_(sd[p + "layer_norm1.weight"]), L.ln1_b.copy_(sd[p + "layer_norm1.bias"])
                L.ln2_w.copy_(sd[p + "layer_norm2.weight"]), L.ln2_b.copy_(sd[p + "layer_norm2.bias"])
                for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    L.qkv_w[j * d:(j + 1) * d].copy_(sd[p + "self_attn.%s.weight" % n])
                    L.qkv_b[j * d:(j + 1) * d].copy_(sd[p + "self_attn.%s.bias" % n])
                L.out_w.copy_(sd[p + "self_attn.out_proj.weight"]), L.out_b.copy_(sd[p + "self_attn.out_proj.bias"])
                L.fc1_w.copy_(sd[p + "mlp.fc1.weight"]), L.fc1_b.copy_(sd[p + "mlp.fc1.bias"])
                L.fc2_w.copy_(sd[p + "mlp.fc2.weight"]), L.fc2_b.copy_(sd[p + "mlp.fc2.bias"])

    # -- forward ---------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:
            return [self._features(im.unsqueeze(0)).to(im.dtype) for im in images]
        return self._features(images).to(images.dtype)

    def _features(self, images):
        cfg = self._cfg
        d, H = cfg.hidden_size, cfg.num_attention_heads
        eps = cfg.layer_norm_eps
        x = images.to(device=self.device, dtype=self.dtype).contiguous()
        N = x.shape[0]
        x = K.clip_patch_embed(x, self.patch_w, cfg.patch_size, d)
        x = torch.cat([self.class_embedding.expand(N, 1, d), x], 1) + self.position_embedding[None]
        x = K.layernorm(x.contiguous(), self.pre_ln_w, self.pre_ln_b, eps)
        n_run = cfg.num_hidden_layers + 1 + self.select_layer if self.select_layer < 0 else self.select_layer
        for L in list(self.layers)[:n_run]:
            h = K.layernorm(x, L.ln1_w, L.ln1_b, eps)
            qkv = F.linear(h, L.qkv_w, L.qkv_b)
            a = K.sdpa_full(qkv, H, d // H)
            x = x + F.linear(a, L.out_w, L.out_b)
            h = K.layernorm(x, L.ln2_w, L.ln2_b, eps)
            h = F.linear(h, L.fc1_w, L.fc1_b)
            h = K.quick_gelu_(h)                      # h * sigmoid(1.702 h), CLIPMLP's quick_gelu, in place
            x = x + F.linear(h, L.fc2_w, L.fc2_b)
        if self.select_feature == "patch":
            return x[:, 1:]
        if self.select_feature == "cls_patch":
            return x
        raise ValueError("Unexpected select feature: %s" % self.select_feature)

    # -- reference surface -----------------------------------------------------------------------
    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.patch_w.dtype if hasattr(self, "patch_w") else self._dtype

    @property
    def device(self):
        return self.patch_w.device if hasattr(self, "patch_w") else torch.device(self._device)

    @property
    def config(self):
        return self._cfg

    @property
    def hidden_size(self):
        return self._cfg.hidden_size

    @property
    def num_patches(self):
        return (self._cfg.image_size // self._cfg.patch_size) ** 2


def _read_checkpoint(path):
    """Read every *.safetensors / pytorch_model*.bin shard under `path` into one CPU state dict."""
    sd = {}
    names = sorted(os.listdir(path))
    st = [n for n in names if n.endswith(".safetensors")]
    if st:
        from safetensors.torch import load_file
        for n in st:
            sd.update(load_file(os.path.join(path, n)))
        return sd
    for n in names:
        if n.endswith(".bin") and n.startswith("pytorch_model"):
            sd.update(torch.load(os.path.join(path, n), map_location="cpu"))
    if not sd:
        raise FileNotFoundError("no weights (*.safetensors / pytorch_model*.bin) under %s" % path)
    return sd


def build_vision_tower(vision_tower_cfg, **kwargs):
    """reference llava/model/multimodal_encoder/builder.py"""
    name = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    return CLIPVisionTower(name, args=vision_tower_cfg, **kwargs)


class Projector(nn.Sequential):
    """The reference's projectors with the reference's parameter names (`mm_projector.0.weight` ...): `mlp<N>x_gelu` =
    nn.Sequential(Linear, (GELU, Linear) x (N-1)) (multimodal_projector/builder.py:39-46); forward runs the fused MFMA GEMMs."""

    def forward(self, x):
        lin = [m for m in self if isinstance(m, nn.Linear)]
        if len(self) != 2 * len(lin) - 1 or any(not isinstance(m, nn.GELU) for m in list(self)[1::2]):
            raise NotImplementedError("Projector: expected Linear (GELU Linear)*")
        x = x.to(lin[0].weight.dtype)                  # the tower hands features back in the images' dtype
        if len(lin) == 2:                              # mlp2x_gelu: the shipped recipe (src/hallava_7b.sh:39)
            return K.projector_mlp(x, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias)
        return K.projector_chain(x, lin)


class LinearProjector(nn.Linear):
    """mm_projector_type 'linear' (multimodal_projector/builder.py:36-37): one nn.Linear (parameter names `mm_projector.weight/.bias`)."""

    def forward(self, x):
        return K.projector_chain(x.to(self.weight.dtype), [self])


class IdentityMap(nn.Module):
    """mm_projector_type 'identity' (multimodal_projector/builder.py:6-16)."""

    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


def build_vision_projector(config, dtype=torch.bfloat16, device="cuda", **kwargs):
    """reference llava/model/multimodal_projector/builder.py:33-51: 'linear', 'mlp<N>x_gelu', 'identity'; anything else raises
    the reference's ValueError."""
    kind = getattr(config, "mm_projector_type", "linear")
    if kind == "linear":
        return LinearProjector(config.mm_hidden_size, config.hidden_size, dtype=dtype, device=device)
    m = re.match(r"^mlp(\d+)x_gelu$", kind)
    if m:
        depth = int(m.group(1))
        mods = [nn.Linear(config.mm_hidden_size, config.hidden_size, dtype=dtype, device=device)]
        for _ in range(1, depth):
            mods += [nn.GELU(), nn.Linear(config.hidden_size, config.hidden_size, dtype=dtype, device=device)]
        return Projector(*mods)
    if kind == "identity":
        return IdentityMap()
    raise ValueError("Unknown projector type: %s" % kind)
