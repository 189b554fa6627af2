"""Oracle (test infrastructure): the VILA twin of the DPA step - reference vila/train/halva_trainer.py:662-852,
vila/model/llava_arch.py:613-871, vila/model/multimodal_projector/base_projector.py:33-54,76-83 and the SigLIP
vision model (vila/model/multimodal_encoder/siglip/modeling_siglip.py:246-449,826-879), restated.

Pure torch-CPU / numpy.  See oracle/__init__.py: nothing outside tests / smoke / bench's cpu_baseline imports this.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import dpa, host, nets

IGNORE_INDEX = host.IGNORE_INDEX


def _key(W, name):
    return W[name] if name in W else W["vision_model." + name]


def siglip_features(images, W, cfg, select_layer=-2, select_feature="cls_patch"):
    """SiglipVisionModel(output_hidden_states=True).hidden_states[select_layer] as taken by
    vision_encoder.py:23-32,121-140: biased 'valid' patch conv + position embedding (no class token, no pre-LN) ->
    encoder layers (LN, MHA with biases and scale head_dim**-0.5, LN, fc1, tanh-GELU, fc2)."""
    g = lambda n: _key(W, n).to(images.dtype)
    d, P, H = cfg["hidden_size"], cfg["patch_size"], cfg["num_attention_heads"]
    eps = cfg.get("layer_norm_eps", 1e-6)
    N = images.shape[0]
    x = F.conv2d(images, g("embeddings.patch_embedding.weight"), g("embeddings.patch_embedding.bias"), stride=P)
    x = x.flatten(2).transpose(1, 2) + g("embeddings.position_embedding.weight")[None]
    n_layers = cfg["num_hidden_layers"]
    stop = n_layers + 1 + select_layer if select_layer < 0 else select_layer
    S, D = x.shape[1], d // H
    for i in range(stop):
        p = "encoder.layers.%d." % i
        h = F.layer_norm(x, (d,), g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), eps)
        q, k, v = (F.linear(h, g(p + "self_attn.%s_proj.weight" % n), g(p + "self_attn.%s_proj.bias" % n))
                   .view(N, S, H, D).transpose(1, 2) for n in "qkv")
        att = torch.matmul(q, k.transpose(2, 3)) * D ** -0.5
        att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        a = torch.matmul(att, v).transpose(1, 2).reshape(N, S, d)
        x = x + F.linear(a, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        h = F.layer_norm(x, (d,), g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"), eps)
        h = F.gelu(F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")), approximate="tanh")
        x = x + F.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    return x[:, 1:] if select_feature == "patch" else x


def downsample(x):
    """DownSampleBlock (base_projector.py:33-54), restated index-wise instead of view/permute:
    out[n, b2*G + a2, (2f + e)*c + ch] = xp[n, 2*a2 + f, 2*b2 + e, ch], xp = the g x g grid zero padded to even size."""
    n, s, c = x.shape
    g = int(s ** 0.5)
    G = (g + 1) // 2
    xp = torch.zeros(n, 2 * G, 2 * G, c, dtype=x.dtype)
    xp[:, :g, :g] = x.reshape(n, g, g, c)
    out = torch.zeros(n, G, G, 4 * c, dtype=x.dtype)
    for f in range(2):
        for e in range(2):
            # token (b2, a2) reads grid row 2*a2+f, column 2*b2+e
            blk = xp[:, f::2, e::2]                        # [n, a2, b2, c]
            out[:, :, :, (2 * f + e) * c:(2 * f + e + 1) * c] = blk.transpose(1, 2)
    return out.reshape(n, G * G, 4 * c)


def projector_downsample(feats, W, prefix="layers."):
    """mlp_downsample (base_projector.py:76-83): DownSampleBlock, LayerNorm(4c), Linear, GELU(erf), Linear."""
    x = downsample(feats)
    x = F.layer_norm(x, (x.shape[-1],), W[prefix + "1.weight"], W[prefix + "1.bias"], 1e-5)
    h = F.linear(x, W[prefix + "2.weight"], W[prefix + "2.bias"])
    return F.linear(F.gelu(h), W[prefix + "4.weight"], W[prefix + "4.bias"])


def kl_to_reference(pol_logits, ref_logits, labels):
    """vila/train/halva_trainer.py:826-841 - the same softmax -> log form as the LLaVA trainer."""
    return dpa.kl_to_reference(pol_logits, ref_logits, labels)


class TinyVila:
    """One VILA model's weights: llm (HF Llama names), SigLIP tower, mm_projector (`layers.N.*`)."""

    def __init__(self, W, cfg, vis_W, vis_cfg, proj_W, max_len, lora=None, lora_scale=0.0, padding_side="right", varlen=False,
                 dtype=torch.float32, select_feature="cls_patch"):
        cv = lambda D: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in D.items()}
        self.W, self.vis_W, self.proj_W = cv(W), cv(vis_W), cv(proj_W)
        self.cfg, self.vis_cfg = cfg, vis_cfg
        self.max_len, self.side = max_len, padding_side
        self.lora, self.scale, self.varlen, self.dtype = lora, lora_scale, varlen, dtype
        self.select_feature = select_feature

    def encode_images(self, images):
        with torch.no_grad():
            f = siglip_features(images.to(self.dtype), self.vis_W, self.vis_cfg, -2, self.select_feature)
        return projector_downsample(f, self.proj_W)

    def spliced_logits(self, ids, mask, labels, signs, images):
        """images: [S, n, 3, H, W] or [M, 3, H, W]; flattened like llava_arch.py:650-653."""
        if images.ndim == 5:
            images = images.flatten(0, 1)
        feats = self.encode_images(images)
        tab = self.W["model.embed_tokens.weight"]
        _, l_np, s_np, m_np = host.splice(np.asarray(ids), np.asarray(mask), np.asarray(labels),
                                          None if signs is None else np.asarray(signs),
                                          np.zeros((feats.shape[0], feats.shape[1], 1), np.float32),
                                          np.zeros((tab.shape[0], 1), np.float32), self.max_len, self.side,
                                          imageless_consumes=False)
        embeds = self._assemble(ids, mask, feats, tab, m_np)
        keep = torch.from_numpy(m_np)
        logits = nets.llama_logits(embeds, keep, self.W, self.cfg, self.lora, self.scale, self.varlen)
        return logits, torch.from_numpy(l_np), (None if s_np is None else torch.from_numpy(s_np)), keep

    def _assemble(self, ids, mask, feats, tab, out_mask):
        B, T = out_mask.shape
        rows, img = [], 0
        for b in range(B):
            cur = torch.as_tensor(np.asarray(ids[b])[np.asarray(mask[b]).astype(bool)])
            pos = (cur == host.IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
            cuts = [-1] + pos + [len(cur)]
            parts = []
            for i in range(len(cuts) - 1):
                parts.append(tab[cur[cuts[i] + 1:cuts[i + 1]]])
                if i < len(pos):
                    parts.append(feats[img])
                    img += 1
            e = torch.cat(parts, 0)[:self.max_len]
            pad = torch.zeros(T - e.shape[0], e.shape[1], dtype=e.dtype)
            rows.append(torch.cat([pad, e], 0) if self.side == "left" else torch.cat([e, pad], 0))
        return torch.stack(rows, 0)


def compute_loss(policy, ref, batch, alpha):
    """vila/train/halva_trainer.py:692-852 on two TinyVila models.  images [B, n, 3, H, W]; ref_images [B, 1, 3, H, W]
    (squeezed, :761)."""
    c_ids, c_lab, c_att, c_sig = host.concat_pos_neg(batch)
    images = torch.as_tensor(np.asarray(batch["images"]))
    logits, labels, signs, _ = policy.spliced_logits(c_ids, c_att, c_lab, c_sig, torch.cat([images, images], 0))
    logps = dpa.cal_batch_logp(logits, labels)
    B = logps.shape[0] // 2
    labels_s, signs_s = labels[:, 1:], signs[:, 1:]
    align, pa, na = dpa.alignment_loss(logps[:B], logps[B:], labels_s, signs_s)
    ref_images = torch.as_tensor(np.asarray(batch["ref_images"]))
    if ref_images.ndim == 5:
        ref_images = ref_images.squeeze(1)
    r_ids, r_att, r_lab = (np.asarray(batch[k]) for k in ("ref_input_ids", "ref_attention_mask", "ref_labels"))
    pol_logits, r_labels, _, _ = policy.spliced_logits(r_ids, r_att, r_lab, None, ref_images)
    with torch.no_grad():
        ref_logits, _, _, _ = ref.spliced_logits(r_ids, r_att, r_lab, None, ref_images)
    div = kl_to_reference(pol_logits[:, :-1], ref_logits[:, :-1], r_labels[:, 1:])
    loss = align + alpha * div
    return loss, dict(alignment=align, divergence=div, pos_logps=logps[:B], neg_logps=logps[B:], pos_acc=pa, neg_acc=na,
                      batch_labels=labels_s, batch_signs=signs_s)
