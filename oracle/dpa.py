"""Oracle (test infrastructure): the DPA loss of reference llava/train/halva_trainer.py:392-592, restated.

Pure torch-CPU / numpy; fp32 unless the caller passes another dtype.  See oracle/__init__.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import host, nets

IGNORE_INDEX = host.IGNORE_INDEX


def cal_batch_logp(logits, labels):
    """halva_trainer.py:392-409: shift by one, IGNORE_INDEX -> token 0, log_softmax, gather."""
    assert logits.shape[:-1] == labels.shape
    tgt = labels[:, 1:].clone()
    tgt[tgt == IGNORE_INDEX] = 0
    return torch.gather(logits[:, :-1].log_softmax(-1), 2, tgt.unsqueeze(2)).squeeze(2)


def accumulate_logps(logps, signs):
    """halva_trainer.py:411-419: slots = sorted unique sign ids over the WHOLE half-batch minus the first
    (assumed 0); column i = sum_t logps * (signs == u[i+1])."""
    u = torch.unique(signs, sorted=True)
    out = torch.zeros(signs.shape[0], len(u) - 1, dtype=logps.dtype)
    for i, s in enumerate(u[1:]):
        out[:, i] = (logps * (signs == s).to(logps.dtype)).sum(-1)
    return out


def alignment_loss(pos_logps, neg_logps, labels, signs):
    """halva_trainer.py:550-568.  labels/signs already shifted ([2B,T-1]), rows 0..B-1 pos, B.. neg."""
    B = pos_logps.shape[0]
    m = (labels != IGNORE_INDEX).to(pos_logps.dtype)
    sg = signs.masked_fill(signs == IGNORE_INDEX, 0)
    pa = accumulate_logps(pos_logps * m[:B], sg[:B])
    na = accumulate_logps(neg_logps * m[B:], sg[B:])
    return torch.log(1 + torch.exp(na - pa)).mean(), pa, na


def kl_to_reference(pol_logits, ref_logits, labels):
    """halva_trainer.py:580-588: softmax -> log (NOT log_softmax), masked by shifted ref labels, SUM / B."""
    m = (labels != IGNORE_INDEX)
    p_ref = F.softmax(ref_logits, -1)
    p_pol = F.softmax(pol_logits, -1)
    div = p_ref * (p_ref.log() - p_pol.log()) * m.unsqueeze(-1)
    return div.sum() / div.shape[0]


class TinyLlava:
    """Holds one model's weights (HF names) and runs encode_images / splice / logits like the reference's
    LlavaLlamaForCausalLM does on this path (llava_arch.py:80-83,85-394; llava_llama.py:42-85)."""

    def __init__(self, W, cfg, clip_W, clip_cfg, max_len, lora=None, lora_scale=0.0, padding_side="right",
                 varlen=False, dtype=torch.float32):
        self.W = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in W.items()}
        self.clip_W = {k: v.to(dtype) for k, v in clip_W.items()}
        self.cfg, self.clip_cfg = cfg, clip_cfg
        self.max_len, self.side = max_len, padding_side
        self.lora, self.scale, self.varlen, self.dtype = lora, lora_scale, varlen, dtype

    def encode_images(self, images):
        with torch.no_grad():
            f = nets.clip_features(images.to(self.dtype), self.clip_W, self.clip_cfg, -2)
        return nets.projector(f, self.W)

    def spliced_logits(self, ids, mask, labels, signs, images):
        feats = self.encode_images(images)
        # index plan from the integer splice (host.splice on arange-valued "embeddings" would also do);
        # here the float rows are assembled with differentiable torch ops following the same plan.
        tab = self.W["model.embed_tokens.weight"]
        e_np, l_np, s_np, m_np = host.splice(np.asarray(ids), np.asarray(mask), np.asarray(labels),
                                             None if signs is None else np.asarray(signs),
                                             np.zeros((feats.shape[0], feats.shape[1], 1), np.float32),
                                             np.zeros((tab.shape[0], 1), np.float32), self.max_len, self.side)
        embeds = self._assemble(ids, mask, feats, tab, m_np)
        keep = torch.from_numpy(m_np)
        logits = nets.llama_logits(embeds, keep, self.W, self.cfg, self.lora, self.scale, self.varlen)
        return logits, torch.from_numpy(l_np), (None if s_np is None else torch.from_numpy(s_np)), keep

    def _assemble(self, ids, mask, feats, tab, out_mask):
        B, T = out_mask.shape
        rows = []
        for b in range(B):
            cur = torch.as_tensor(np.asarray(ids[b])[np.asarray(mask[b]).astype(bool)])
            parts, img_used = [], False
            pos = (cur == host.IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
            cuts = [-1] + pos + [len(cur)]
            for i in range(len(cuts) - 1):
                parts.append(tab[cur[cuts[i] + 1:cuts[i + 1]]])
                if i < len(pos):
                    parts.append(feats[b])       # one image per sample on this path
            e = torch.cat(parts, 0)[:self.max_len]
            pad = torch.zeros(T - e.shape[0], e.shape[1], dtype=e.dtype)
            rows.append(torch.cat([pad, e], 0) if self.side == "left" else torch.cat([e, pad], 0))
        return torch.stack(rows, 0)


def compute_loss(policy, ref, batch, alpha):
    """halva_trainer.py:534-592 on two TinyLlava models.  Returns (loss, parts dict)."""
    c_ids, c_lab, c_att, c_sig = host.concat_pos_neg(batch)
    images = torch.as_tensor(np.asarray(batch["images"]))
    logits, labels, signs, _ = policy.spliced_logits(c_ids, c_att, c_lab, c_sig, torch.cat([images, images], 0))
    logps = cal_batch_logp(logits, labels)
    B = logps.shape[0] // 2
    labels_s, signs_s = labels[:, 1:], signs[:, 1:]
    align, pa, na = alignment_loss(logps[:B], logps[B:], labels_s, signs_s)
    ref_images = torch.as_tensor(np.asarray(batch["ref_images"]))
    r_ids, r_att, r_lab = (np.asarray(batch[k]) for k in ("ref_input_ids", "ref_attention_mask", "ref_labels"))
    pol_logits, r_labels, _, _ = policy.spliced_logits(r_ids, r_att, r_lab, None, ref_images)
    with torch.no_grad():
        ref_logits, _, _, _ = ref.spliced_logits(r_ids, r_att, r_lab, None, ref_images)
    div = kl_to_reference(pol_logits[:, :-1], ref_logits[:, :-1], r_labels[:, 1:])
    loss = align + alpha * div
    return loss, dict(alignment=align, divergence=div, pos_logps=logps[:B], neg_logps=logps[B:], pos_acc=pa, neg_acc=na,
                      batch_labels=labels_s, batch_signs=signs_s, all_logits=logits[:, :-1])
