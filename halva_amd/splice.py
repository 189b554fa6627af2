"""Host-side index plan for the text/image splice (product code; integer work only).

The reference builds `inputs_embeds` with a per-sample Python loop of small device ops and `.tolist()` syncs
(reference llava/model/llava_arch.py:277-374).  Everything in that loop except the final row copies depends
only on `input_ids` / `attention_mask`, which are host data before the H2D copy - so the plan (which source row
feeds every output row, the spliced labels / signs / attention mask, the per-sequence valid span) is computed
here on the host, and one gather kernel (halva_splice_rows) materialises the embeddings on the GPU.

Semantics reproduced exactly (tests/test_hip_kernels.py::test_splice_rows_against_golden, bit-exact):
  * padding is removed with the attention mask, each IMAGE_TOKEN_INDEX expands to the image's n_patch feature rows
    with labels/signs := IGNORE_INDEX, one image is consumed per image token - and also by an image-less sample
    (llava_arch.py:287-294);
  * sequences are truncated to tokenizer_model_max_length AFTER the splice (:334-339);
  * right (or left) padding with zero vectors / IGNORE_INDEX / False (:341-374).
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200


@dataclass
class SplicePlan:
    src: torch.Tensor                 # int32 [S*T]: >=0 token id, -1 zero pad, <=-2 feature row -(idx)-2
    labels: torch.Tensor              # int64 [S, T]
    signs: Optional[torch.Tensor]     # int64 [S, T] or None
    mask: torch.Tensor                # bool  [S, T]
    seq_start: torch.Tensor           # int32 [S]
    seq_len: torch.Tensor             # int32 [S]
    S: int
    T: int
    n_images: int                     # image slots consumed (running index of the reference)


def image_slots(input_ids, attention_mask, imageless_consumes=True):
    """Running image index of the reference's splice loop: for every row, the list of image slots its image tokens
    read, plus the total number of slots consumed.  LLaVA advances the index for an image-less row too
    (llava_arch.py:287-294); VILA does not (vila/model/llava_arch.py:716-718)."""
    ids = np.asarray(input_ids)
    att = np.ones(ids.shape, dtype=bool) if attention_mask is None else np.asarray(attention_mask).astype(bool)
    out, slot = [], 0
    for b in range(ids.shape[0]):
        n_img = int(((ids[b] == IMAGE_TOKEN_INDEX) & att[b]).sum())
        out.append(list(range(slot, slot + n_img)))
        slot += n_img if n_img else (1 if imageless_consumes else 0)
    return out, slot


def plan_splice(input_ids, attention_mask, labels, signs, n_patch, max_len=None, padding_side="right", image_map=None,
                imageless_consumes=True):
    """input_ids/labels/signs: int64 [S, L] (CPU), attention_mask bool [S, L] or None.  image_map (optional) maps the
    running image-slot index of the reference to a row of the feature tensor (lets pos/neg rows share one encode).
    imageless_consumes: see image_slots()."""
    ids = np.asarray(input_ids)
    S, L = ids.shape
    att = np.ones((S, L), dtype=bool) if attention_mask is None else np.asarray(attention_mask).astype(bool)
    lab = np.full((S, L), IGNORE_INDEX, dtype=np.int64) if labels is None else np.asarray(labels)
    sgn = None if signs is None else np.asarray(signs)
    rows_src, rows_lab, rows_sgn = [], [], []
    slot = 0
    for b in range(S):
        keep = att[b]
        cur = ids[b][keep]
        is_img = cur == IMAGE_TOKEN_INDEX
        n_img = int(is_img.sum())
        reps = np.where(is_img, n_patch, 1)
        src = np.repeat(cur, reps).astype(np.int64)
        l = np.repeat(lab[b][keep], reps)
        s = np.repeat(sgn[b][keep], reps) if sgn is not None else None
        if n_img:
            starts = np.cumsum(reps) - reps                       # output offset of every kept token
            for j, pos in enumerate(np.nonzero(is_img)[0]):
                feat = slot + j if image_map is None else int(image_map[slot + j])
                o = starts[pos]
                src[o:o + n_patch] = -(feat * n_patch + np.arange(n_patch)) - 2
                l[o:o + n_patch] = IGNORE_INDEX
                if s is not None:
                    s[o:o + n_patch] = IGNORE_INDEX
            slot += n_img
        elif imageless_consumes:
            slot += 1                                             # an image-less sample still consumes one image (LLaVA)
        if max_len is not None:
            src, l = src[:max_len], l[:max_len]
            if s is not None:
                s = s[:max_len]
        rows_src.append(src)
        rows_lab.append(l)
        rows_sgn.append(s)
    T = max(len(r) for r in rows_src)
    out_src = np.full((S, T), -1, dtype=np.int32)
    out_lab = np.full((S, T), IGNORE_INDEX, dtype=np.int64)
    out_sgn = np.full((S, T), IGNORE_INDEX, dtype=np.int64) if sgn is not None else None
    out_mask = np.zeros((S, T), dtype=bool)
    start = np.zeros(S, dtype=np.int32)
    length = np.zeros(S, dtype=np.int32)
    for b in range(S):
        n = len(rows_src[b])
        o = T - n if padding_side == "left" else 0
        out_src[b, o:o + n] = rows_src[b]
        out_lab[b, o:o + n] = rows_lab[b]
        if out_sgn is not None:
            out_sgn[b, o:o + n] = rows_sgn[b]
        out_mask[b, o:o + n] = True
        start[b], length[b] = o, n
    return SplicePlan(src=torch.from_numpy(out_src.reshape(-1)), labels=torch.from_numpy(out_lab),
                      signs=None if out_sgn is None else torch.from_numpy(out_sgn), mask=torch.from_numpy(out_mask),
                      seq_start=torch.from_numpy(start), seq_len=torch.from_numpy(length), S=S, T=T, n_images=slot)


def spans_from_mask(mask):
    """[S, T] bool key-padding mask -> (seq_start, seq_len) int32; the valid tokens must be one contiguous run
    (what right/left padding produces; the reference's unpad_input accepts holes, the splice never makes them)."""
    m = np.asarray(mask).astype(bool)
    S, T = m.shape
    length = m.sum(1).astype(np.int32)
    start = np.where(length > 0, m.argmax(1), 0).astype(np.int32)
    idx = np.arange(T)[None]
    if not np.array_equal(m, (idx >= start[:, None]) & (idx < (start + length)[:, None])):
        raise ValueError("attention_mask must mark one contiguous run of valid tokens per sequence")
    return torch.from_numpy(start), torch.from_numpy(length)


# ------------------------------------------------------------------------------------------------
# prefix sharing between the two rows of a pair
# ------------------------------------------------------------------------------------------------
@dataclass
class PackedPairs:
    """g packed rows [prefix | A | pad | B] built from the 2g spliced rows [pos(0..g-1) ; neg(0..g-1)] of a pair group.

    The correct and the hallucinated sequence of a pair start with the same image, prompt and (usually) the same beginning of
    the response; under a causal mask their hidden states are identical up to the first differing input row, so the prefix is
    run ONCE: A = the rest of the correct row, B = the rest of the hallucinated row, B never attends to A
    (halva_sdpa_branch_fwd) and B's RoPE positions continue from the prefix."""
    src: torch.Tensor          # int32 [g * T]: splice source per packed row (same encoding as SplicePlan.src)
    pos: torch.Tensor          # int32 [g * T]: RoPE position of every packed row
    br_a: torch.Tensor         # int32 [g]: prefix length (first row of A)
    br_b: torch.Tensor         # int32 [g]: first row of B (multiple of 64)
    seq_len: torch.Tensor      # int32 [g]
    T: int
    row_of: np.ndarray         # int64 [2g, T_unpacked]: packed flat row index holding the hidden state of (row, position); -1 = none
    rows_packed: int           # sum of packed lengths
    rows_unpacked: int         # sum of the 2g un-packed lengths


def pack_pairs(plan, align=64):
    """plan: SplicePlan of 2g right-padded rows, row i and row g+i forming a pair."""
    S, T = plan.S, plan.T
    g = S // 2
    src = plan.src.numpy().reshape(S, T)
    lens = plan.seq_len.numpy().astype(np.int64)
    if int(plan.seq_start.numpy().max(initial=0)) != 0:
        raise ValueError("pack_pairs needs right-padded rows")
    rows, poss, a_s, b_s, n_s = [], [], [], [], []
    maps = []
    for i in range(g):
        lp, ln = int(lens[i]), int(lens[g + i])
        m = min(lp, ln)
        neq = np.nonzero(src[i, :m] != src[g + i, :m])[0]
        L = int(neq[0]) if len(neq) else m                      # length of the common prefix
        b = (lp + align - 1) // align * align                   # B starts on a tile boundary; rows [lp, b) are padding
        nb = ln - L
        row = np.full(b + nb, -1, dtype=np.int32)
        row[:lp] = src[i, :lp]
        row[b:] = src[g + i, L:ln]
        pos = np.zeros(b + nb, dtype=np.int32)
        pos[:lp] = np.arange(lp)
        pos[b:] = np.arange(L, ln)
        rows.append(row), poss.append(pos), a_s.append(L), b_s.append(b), n_s.append(b + nb)
        mp, mn = np.full(T, -1, dtype=np.int64), np.full(T, -1, dtype=np.int64)
        mp[:lp] = np.arange(lp)
        mn[:L] = np.arange(L)
        mn[L:ln] = b + np.arange(nb)
        maps.append((mp, mn))
    Tp = max(n_s) if n_s else 0
    out_src = np.full((g, Tp), -1, dtype=np.int32)
    out_pos = np.zeros((g, Tp), dtype=np.int32)
    row_of = np.full((S, T), -1, dtype=np.int64)
    for i in range(g):
        out_src[i, :n_s[i]] = rows[i]
        out_pos[i, :n_s[i]] = poss[i]
        mp, mn = maps[i]
        row_of[i] = np.where(mp >= 0, mp + i * Tp, -1)
        row_of[g + i] = np.where(mn >= 0, mn + i * Tp, -1)
    i32 = lambda v: torch.from_numpy(np.asarray(v, dtype=np.int32))
    return PackedPairs(src=torch.from_numpy(out_src.reshape(-1)), pos=torch.from_numpy(out_pos.reshape(-1)), br_a=i32(a_s), br_b=i32(b_s),
                       seq_len=i32(n_s), T=Tp, row_of=row_of, rows_packed=int(sum(n_s)), rows_unpacked=int(lens.sum()))
