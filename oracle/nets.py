"""Oracle (test infrastructure): the networks on the DPA path restated as plain functional torch-CPU code.

Numerics spec = the vendored transformers-4.31 file the reference ran against
(reference llava/model/language_model/modelling_llama.py) for Llama, and HF CLIPVisionModel semantics as
invoked by reference llava/model/multimodal_encoder/clip_encoder.py for the vision tower.  Works in
any float dtype (fp32 for parity, bf16 for the CPU-baseline timing); rounding points follow the spec.
Weights come in as a flat {name: tensor} dict using HF parameter names.  See oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# Llama pieces
# ----------------------------------------------------------------------------------------------
def rmsnorm(x, w, eps):
    """modelling_llama.py:65-70: fp32 upcast, x * rsqrt(mean(x^2) + eps), cast back, then * weight."""
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(x.dtype)


def rope_tables(head_dim, seq_len, base=10000.0, dtype=torch.float32):
    """modelling_llama.py:79-106: inv_freq fp32, emb = cat(freqs, freqs); cos/sin cast to the compute dtype."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    freqs = torch.outer(torch.arange(seq_len, dtype=torch.float32), inv)
    emb = torch.cat([freqs, freqs], -1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rope_apply(x, cos, sin, position_ids):
    """modelling_llama.py:154-169.  x [B,H,T,D]; cos/sin [Tmax,D]; position_ids [B or 1, T]."""
    c = cos[position_ids].unsqueeze(1)
    s = sin[position_ids].unsqueeze(1)
    half = x.shape[-1] // 2
    rot = torch.cat([-x[..., half:], x[..., :half]], -1)
    return x * c + rot * s


def additive_mask(keep, dtype):
    """modelling_llama.py:24-53,556-577: causal + key-padding, both as finfo.min additive terms (summed)."""
    B, T = keep.shape
    neg = torch.finfo(dtype).min
    causal = torch.full((T, T), neg, dtype=dtype).triu(1)
    pad = torch.zeros(B, 1, 1, T, dtype=dtype).masked_fill(~keep[:, None, None, :], neg)
    return pad + causal[None, None]


def lora_linear(x, w, lora=None, scale=0.0):
    """peft 0.4.0 Linear.forward (dropout forced to 0, reference halva_trainer.py:35-38,184-187):
    F.linear(x, W) + scale * lora_B(lora_A(x)), scale = alpha / r."""
    y = F.linear(x, w)
    if lora is not None:
        a, b = lora
        y = y + F.linear(F.linear(x, a.to(x.dtype)), b.to(x.dtype)) * scale
    return y


def attention_eager(q, k, v, mask):
    """modelling_llama.py:311-328: scores / sqrt(d) + mask, softmax in fp32, cast, @ V."""
    att = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(q.shape[-1])
    att = att + mask
    att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(att, v)


def attention_varlen(q, k, v, keep):
    """What the reference's GPU path computes (llava/train/llama_flash_attn_monkey_patch.py:71-91):
    un-pad by the key-padding mask, causal attention inside each sequence, re-pad with ZEROS."""
    B, H, T, D = q.shape
    out = torch.zeros_like(q)
    for b in range(B):
        idx = keep[b].nonzero().flatten()
        n = idx.numel()
        if n == 0:
            continue
        qq, kk, vv = q[b][:, idx], k[b][:, idx], v[b][:, idx]
        att = torch.matmul(qq, kk.transpose(1, 2)) / math.sqrt(D)
        att = att + torch.full((n, n), float("-inf"), dtype=att.dtype).triu(1)
        att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        out[b][:, idx] = torch.matmul(att, vv)
    return out


def decoder_layer(x, W, pre, keep, cfg, lora=None, scale=0.0, varlen=False):
    """modelling_llama.py:352-420 (LlamaDecoderLayer.forward) with the attention of :243-345."""
    B, T, d = x.shape
    H = cfg["num_attention_heads"]
    D = d // H
    L = (lambda n: (lora[pre + n + ".A"], lora[pre + n + ".B"]) if lora is not None and (pre + n + ".A") in lora else None)
    h = rmsnorm(x, W[pre + "input_layernorm.weight"], cfg["rms_norm_eps"])
    q = lora_linear(h, W[pre + "self_attn.q_proj.weight"], L("self_attn.q_proj"), scale).view(B, T, H, D).transpose(1, 2)
    k = lora_linear(h, W[pre + "self_attn.k_proj.weight"], L("self_attn.k_proj"), scale).view(B, T, H, D).transpose(1, 2)
    v = lora_linear(h, W[pre + "self_attn.v_proj.weight"], L("self_attn.v_proj"), scale).view(B, T, H, D).transpose(1, 2)
    cos, sin = rope_tables(D, T, cfg.get("rope_theta", 10000.0), x.dtype)
    pos = torch.arange(T)[None]
    q, k = rope_apply(q, cos, sin, pos), rope_apply(k, cos, sin, pos)
    if varlen:
        a = attention_varlen(q, k, v, keep)
    else:
        a = attention_eager(q, k, v, additive_mask(keep, x.dtype))
    a = a.transpose(1, 2).reshape(B, T, d)
    x = x + lora_linear(a, W[pre + "self_attn.o_proj.weight"], L("self_attn.o_proj"), scale)
    h = rmsnorm(x, W[pre + "post_attention_layernorm.weight"], cfg["rms_norm_eps"])
    g = lora_linear(h, W[pre + "mlp.gate_proj.weight"], L("mlp.gate_proj"), scale)
    u = lora_linear(h, W[pre + "mlp.up_proj.weight"], L("mlp.up_proj"), scale)
    return x + lora_linear(F.silu(g) * u, W[pre + "mlp.down_proj.weight"], L("mlp.down_proj"), scale)


def llama_logits(embeds, keep, W, cfg, lora=None, scale=0.0, varlen=False, upto_hidden=False):
    """modelling_llama.py:580-705,741-806: decoder stack on inputs_embeds -> final norm -> lm_head -> .float()."""
    x = embeds
    for i in range(cfg["num_hidden_layers"]):
        x = decoder_layer(x, W, "model.layers.%d." % i, keep, cfg, lora, scale, varlen)
    x = rmsnorm(x, W["model.norm.weight"], cfg["rms_norm_eps"])
    if upto_hidden:
        return x
    return F.linear(x, W["lm_head.weight"]).float()


# ----------------------------------------------------------------------------------------------
# CLIP vision tower + projector
# ----------------------------------------------------------------------------------------------
def _clip_key(W, name):
    return W[name] if name in W else W["vision_model." + name]


def clip_features(images, W, cfg, select_layer=-2):
    """HF CLIPVisionModel(output_hidden_states=True).hidden_states[select_layer][:, 1:] as called from
    reference clip_encoder.py:27-49 (select_feature == 'patch'): conv patch-embed (no bias) + class token +
    position embedding -> pre-LN -> encoder layers (LN, MHA with biases, LN, fc1, quick_gelu, fc2).
    hidden_states[k] = input to layer k, so select_layer=-2 stops before the last layer."""
    g = lambda n: _clip_key(W, n).to(images.dtype)
    d, P = cfg["hidden_size"], cfg["patch_size"]
    H = cfg["num_attention_heads"]
    eps = cfg.get("layer_norm_eps", 1e-5)
    N = images.shape[0]
    x = F.conv2d(images, g("embeddings.patch_embedding.weight"), stride=P).flatten(2).transpose(1, 2)
    x = torch.cat([g("embeddings.class_embedding").expand(N, 1, d), x], 1) + g("embeddings.position_embedding.weight")[None]
    x = F.layer_norm(x, (d,), g("pre_layrnorm.weight"), g("pre_layrnorm.bias"), eps)
    n_layers = cfg["num_hidden_layers"]
    stop = n_layers + 1 + select_layer if select_layer < 0 else select_layer   # number of layers to run
    S = x.shape[1]
    for i in range(stop):
        p = "encoder.layers.%d." % i
        h = F.layer_norm(x, (d,), g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), eps)
        q = F.linear(h, g(p + "self_attn.q_proj.weight"), g(p + "self_attn.q_proj.bias")).view(N, S, H, d // H).transpose(1, 2)
        k = F.linear(h, g(p + "self_attn.k_proj.weight"), g(p + "self_attn.k_proj.bias")).view(N, S, H, d // H).transpose(1, 2)
        v = F.linear(h, g(p + "self_attn.v_proj.weight"), g(p + "self_attn.v_proj.bias")).view(N, S, H, d // H).transpose(1, 2)
        att = torch.matmul(q, k.transpose(2, 3)) * (d // H) ** -0.5
        att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        a = torch.matmul(att, v).transpose(1, 2).reshape(N, S, d)
        x = x + F.linear(a, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        h = F.layer_norm(x, (d,), g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"), eps)
        h = F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"))
        h = h * torch.sigmoid(1.702 * h)                       # quick_gelu
        x = x + F.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    return x[:, 1:]


def projector(feats, W, prefix="model.mm_projector."):
    """reference multimodal_projector/builder.py:39-46 (mlp2x_gelu): Linear -> GELU(erf) -> Linear."""
    h = F.linear(feats, W[prefix + "0.weight"], W[prefix + "0.bias"])
    return F.linear(F.gelu(h), W[prefix + "2.weight"], W[prefix + "2.bias"])
