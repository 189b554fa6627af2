"""Oracle (test infrastructure): integer / index side of the DPA path, restated with plain Python + numpy.

Follows the reference line by line in behaviour (quirks included), not in code.  See oracle/__init__.py.
"""
import math

import numpy as np
import torch

IGNORE_INDEX = -100          # reference llava/constants.py:7
IMAGE_TOKEN_INDEX = -200     # reference llava/constants.py:8
MASK_OPEN, MASK_CLOSE = "<MASK>", "</MASK>"   # reference llava/train/train_halva.py:24-25

# reference llava/conversation.py:252-262 (conv_vicuna_v1, SeparatorStyle.TWO)
V1_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the user's questions.")
V1_ROLES = ("USER", "ASSISTANT")
V1_SEP, V1_SEP2 = " ", "</s>"


def v1_prompt(turns):
    """reference llava/conversation.py:51-60: system + sep, then 'ROLE: msg' + alternating seps."""
    out = V1_SYSTEM + V1_SEP
    for i, msg in enumerate(turns):
        role = V1_ROLES[i % 2]
        out += (role + ": " + msg + (V1_SEP, V1_SEP2)[i % 2]) if msg else role + ":"
    return out


def image_token_ids(prompt, tok):
    """reference llava/mm_utils.py:43-62 (tokenizer_image_token): tokenise around '<image>', keep one BOS."""
    chunks = [tok(c).input_ids for c in prompt.split("<image>")]
    has_bos = bool(chunks and chunks[0] and chunks[0][0] == tok.bos_token_id)
    ids = [chunks[0][0]] if has_bos else []
    off = 1 if has_bos else 0
    for n, c in enumerate(chunks):
        if n:
            ids.extend(([IMAGE_TOKEN_INDEX] * (off + 1))[off:])
        ids.extend(c[off:])
    return ids


def walk_masked(text, tok):
    """reference llava/train/train_halva.py:263-335 (split_string_by_mask_and_tokenize).

    Each un-tagged stretch and each tagged phrase is tokenised on its own; the leading BOS (+ the
    lone word-boundary piece, except for the very first stretch) and the trailing piece are dropped.
    A '.'/','/"'s" that directly follows a close tag is glued to the phrase and gets sign 0.
    """
    ids, signs = [], []
    cursor, tag = 0, 1
    while True:
        a = text.find(MASK_OPEN, cursor)
        if a < 0:
            ids += tok(text[cursor:]).input_ids[2:-1]
            signs += [0] * (len(ids) - len(signs))
            return ids, signs
        b = text.find(MASK_CLOSE, a + len(MASK_OPEN))
        ids += tok(text[cursor:a]).input_ids[(1 if cursor == 0 else 2):-1]
        signs += [0] * (len(ids) - len(signs))
        inner = text[a + len(MASK_OPEN):b]
        after = b + len(MASK_CLOSE)
        if text[after:after + 1] in ".,":        # NB: '' in ".," is True, as in the reference (:296)
            glued = (inner + text[after:after + 1]).replace(" .", ". ").replace(" ,", ", ")
            ids += tok(glued).input_ids[2:-1]
            signs += [tag] * (len(ids) - len(signs) - 1) + [0]
            cursor = after + 1
        elif text[after:after + 2] == "'s":
            glued = (inner + "'s").replace(" 's", "'s ")
            ids += tok(glued).input_ids[2:-1]
            signs += [tag] * (len(ids) - len(signs) - 1) + [0]
            cursor = after + 2
        else:
            ids += tok(inner).input_ids[2:-1]
            signs += [tag] * (len(ids) - len(signs))
            cursor = after
        tag += 1


def masked_image_ids(prompt, tok):
    """reference llava/train/train_halva.py:338-363 (tokenizer_image_token_masked)."""
    parts = prompt.split("<image>")
    assert len(parts) == 2
    assert MASK_OPEN not in parts[0]
    ids = list(tok(parts[0]).input_ids) + [IMAGE_TOKEN_INDEX]
    signs = [0] * len(ids)
    w_ids, w_signs = walk_masked(parts[1], tok)
    return ids + w_ids + [tok.eos_token_id], signs + w_signs + [0]


def _mask_instruction(ids, conversation, tok):
    """reference llava/train/train_halva.py:432-473 / 517-556: labels = ids with BOS and each round's
    instruction part set to IGNORE_INDEX; whole sample ignored on a length mismatch."""
    labels = list(ids)
    total = sum(1 for t in ids if t != tok.pad_token_id)
    sep = V1_SEP + V1_ROLES[1] + ": "
    cur = 1
    labels[:cur] = [IGNORE_INDEX] * cur
    for rou in conversation.split(V1_SEP2):
        if rou == "":
            break
        parts = rou.split(sep)
        if len(parts) != 2:
            break
        round_len = len(image_token_ids(rou, tok))
        instr_len = len(image_token_ids(parts[0] + sep, tok)) - 2
        for t in range(cur, min(cur + instr_len, len(labels))):
            labels[t] = IGNORE_INDEX
        cur += round_len
    for t in range(cur, len(labels)):
        labels[t] = IGNORE_INDEX
    if cur < tok.model_max_length and cur != total:
        labels = [IGNORE_INDEX] * len(labels)
    return labels


def preprocess_v1(question, answer_masked, answer_plain, tok):
    """reference llava/train/train_halva.py:366-479.  Returns dict(input_ids, labels, signs) as lists,
    None when the masked tokenisation differs element-wise from the plain one (:426-430), and raises
    RuntimeError when the two differ in LENGTH (torch broadcasting error at :426 in the reference)."""
    plain = v1_prompt([question, answer_plain])
    masked = v1_prompt([question, answer_masked])
    ref_ids = image_token_ids(plain, tok)
    ids, signs = masked_image_ids(masked, tok)
    if len(ids) != len(ref_ids):
        raise RuntimeError("masked/plain tokenisation length mismatch: %d vs %d" % (len(ids), len(ref_ids)))
    if ids != ref_ids:
        return None
    return dict(input_ids=ids, labels=_mask_instruction(ids, plain, tok), signs=signs)


def preprocess_v1_ref(question, answer, tok):
    """reference llava/train/train_halva.py:481-561 (image case)."""
    conv = v1_prompt([question, answer])
    ids = image_token_ids(conv, tok)
    return dict(input_ids=ids, labels=_mask_instruction(ids, conv, tok))


# ------------------------------------------------------------------------------------------------
# collator - reference llava/train/train_halva.py:896-993
# ------------------------------------------------------------------------------------------------
def _pad_right(seqs, fill, max_len):
    width = max(len(s) for s in seqs)
    out = np.full((len(seqs), width), fill, dtype=np.int64)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = np.asarray(s, dtype=np.int64)
    return out[:, :max_len]


def collate(instances, pad_token_id, model_max_length):
    fills = dict(input_ids=pad_token_id, labels=IGNORE_INDEX, neg_input_ids=pad_token_id, neg_labels=IGNORE_INDEX,
                 pos_signs=0, neg_signs=0, ref_input_ids=pad_token_id, ref_labels=IGNORE_INDEX)
    batch = {k: _pad_right([np.asarray(x[k]) for x in instances], f, model_max_length) for k, f in fills.items()}
    batch["attention_mask"] = batch["input_ids"] != pad_token_id
    batch["neg_attention_mask"] = batch["neg_input_ids"] != pad_token_id
    batch["ref_attention_mask"] = batch["ref_input_ids"] != pad_token_id
    for src, dst in (("image", "images"), ("ref_image", "ref_images")):
        if src in instances[0]:
            batch[dst] = np.stack([np.asarray(x[src]) for x in instances])
    return batch


# ------------------------------------------------------------------------------------------------
# sampler - reference llava/train/halva_trainer.py:60-152
# ------------------------------------------------------------------------------------------------
def split_to_even_chunks(indices, lengths, num_chunks):
    """reference halva_trainer.py:60-79: greedy fill of the currently-shortest chunk; strided split if ragged."""
    if len(indices) % num_chunks:
        return [indices[i::num_chunks] for i in range(num_chunks)]
    cap = len(indices) // num_chunks
    chunks = [[] for _ in range(num_chunks)]
    load = [0.0] * num_chunks
    for i in indices:
        j = load.index(min(load))
        chunks[j].append(i)
        load[j] += lengths[i]
        if len(chunks[j]) == cap:
            load[j] = math.inf
    return chunks


def length_grouped_indices(lengths, batch_size, world_size, generator=None):
    """reference halva_trainer.py:110-118.  torch.randperm is part of the contract (same RNG stream)."""
    perm = torch.randperm(len(lengths), generator=generator).tolist()
    mega = world_size * batch_size
    out = []
    for s in range(0, len(lengths), mega):
        block = sorted(perm[s:s + mega], key=lambda i: lengths[i], reverse=True)
        for chunk in split_to_even_chunks(block, lengths, world_size):
            out.extend(chunk)
    return out


def modality_length_grouped_indices(lengths, batch_size, world_size, generator=None):
    """reference halva_trainer.py:82-107."""
    assert all(l != 0 for l in lengths)
    if all(l > 0 for l in lengths) or all(l < 0 for l in lengths):
        return length_grouped_indices(lengths, batch_size, world_size, generator=generator)
    mm = [(i, l) for i, l in enumerate(lengths) if l > 0]
    lang = [(i, -l) for i, l in enumerate(lengths) if l < 0]
    mm_sh = [mm[i][0] for i in length_grouped_indices([l for _, l in mm], batch_size, world_size, generator=None)]
    lang_sh = [lang[i][0] for i in length_grouped_indices([l for _, l in lang], batch_size, world_size, generator=None)]
    mega = world_size * batch_size
    mm_mb = [mm_sh[i:i + mega] for i in range(0, len(mm_sh), mega)]
    lang_mb = [lang_sh[i:i + mega] for i in range(0, len(lang_sh), mega)]
    tail = mm_mb[-1] + lang_mb[-1]
    mbs = mm_mb[:-1] + lang_mb[:-1]
    order = torch.randperm(len(mbs), generator=generator).tolist()
    mbs = [mbs[i] for i in order]
    if tail:
        mbs.append(sorted(tail))
    return [i for mb in mbs for i in mb]


# ------------------------------------------------------------------------------------------------
# splice - reference llava/model/llava_arch.py:85-226 (unsigned) / :229-394 (signed)
# ------------------------------------------------------------------------------------------------
def splice(input_ids, attention_mask, labels, signs, image_features, embed_tokens, max_len, padding_side="right",
           imageless_consumes=True):
    """Per-sample loop exactly as the reference: drop pads by mask, cut at IMAGE_TOKEN_INDEX, embed text,
    insert the image's feature rows (labels/signs := IGNORE_INDEX there), truncate to max_len, pad with
    ZERO vectors / IGNORE_INDEX.  numpy in, numpy out.  signs may be None (un-signed twin).
    image_features: [n_images, n_patch, d]; one image is consumed per sample, also by image-less samples
    (llava_arch.py:287-294).  imageless_consumes=False is the VILA twin (vila/model/llava_arch.py:708-718:
    "we do not have placeholdr image for text-only data now", the index is not advanced).
    """
    B = input_ids.shape[0]
    d = embed_tokens.shape[1]
    rows, labs, sgs = [], [], []
    img = 0
    for b in range(B):
        keep = attention_mask[b].astype(bool)
        ids = input_ids[b][keep]
        lab = labels[b][keep]
        sg = signs[b][keep] if signs is not None else None
        cuts = [-1] + np.where(ids == IMAGE_TOKEN_INDEX)[0].tolist() + [len(ids)]
        n_img = len(cuts) - 2
        e_parts, l_parts, s_parts = [], [], []
        if n_img == 0:
            e_parts.append(embed_tokens[ids])
            l_parts.append(lab)
            if sg is not None:
                s_parts.append(sg)
            if imageless_consumes:
                img += 1
        else:
            for i in range(n_img + 1):
                sl = slice(cuts[i] + 1, cuts[i + 1])
                e_parts.append(embed_tokens[ids[sl]])
                l_parts.append(lab[sl])
                if sg is not None:
                    s_parts.append(sg[sl])
                if i < n_img:
                    f = image_features[img]
                    img += 1
                    e_parts.append(f)
                    l_parts.append(np.full(f.shape[0], IGNORE_INDEX, dtype=np.int64))
                    if sg is not None:
                        s_parts.append(np.full(f.shape[0], IGNORE_INDEX, dtype=np.int64))
        e = np.concatenate(e_parts, 0)
        l = np.concatenate(l_parts, 0)
        if max_len is not None:
            e, l = e[:max_len], l[:max_len]
        rows.append(e)
        labs.append(l)
        if sg is not None:
            s = np.concatenate(s_parts, 0)
            sgs.append(s[:max_len] if max_len is not None else s)
    T = max(r.shape[0] for r in rows)
    out_e = np.zeros((B, T, d), dtype=embed_tokens.dtype)
    out_l = np.full((B, T), IGNORE_INDEX, dtype=np.int64)
    out_s = np.full((B, T), IGNORE_INDEX, dtype=np.int64) if signs is not None else None
    out_m = np.zeros((B, T), dtype=bool)
    for b in range(B):
        n = rows[b].shape[0]
        sl = slice(T - n, T) if padding_side == "left" else slice(0, n)
        if n:
            out_e[b, sl] = rows[b]
            out_l[b, sl] = labs[b]
            out_m[b, sl] = True
            if out_s is not None:
                out_s[b, sl] = sgs[b]
    return out_e, out_l, out_s, out_m


def concat_pos_neg(batch):
    """reference llava/train/halva_trainer.py:434-447: rows 0..B-1 = pos, B..2B-1 = neg; zero / -100 fill."""
    ids, neg = np.asarray(batch["input_ids"]), np.asarray(batch["neg_input_ids"])
    B = ids.shape[0]
    T0 = max(ids.shape[1], neg.shape[1])
    c_ids = np.zeros((2 * B, T0), dtype=np.int64)
    c_lab = np.full((2 * B, T0), IGNORE_INDEX, dtype=np.int64)
    c_att = np.zeros((2 * B, T0), dtype=bool)
    c_sig = np.zeros((2 * B, T0), dtype=np.int64)
    for off, i, l, a, s in ((0, "input_ids", "labels", "attention_mask", "pos_signs"),
                            (B, "neg_input_ids", "neg_labels", "neg_attention_mask", "neg_signs")):
        w = np.asarray(batch[i]).shape[1]
        c_ids[off:off + B, :w] = np.asarray(batch[i])
        c_lab[off:off + B, :w] = np.asarray(batch[l])
        c_att[off:off + B, :w] = np.asarray(batch[a])
        c_sig[off:off + B, :w] = np.asarray(batch[s])
    return c_ids, c_lab, c_att, c_sig
