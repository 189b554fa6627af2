"""The DPA loss and step engine (product code).

Restructures reference llava/train/halva_trainer.py:421-592 (concatenated_forward / reference_forward / compute_loss)
for one MI355X without changing its arithmetic:

  * the loss decomposes exactly over pairs: alignment = sum_b sum_p softplus(neg_acc[b,p] - pos_acc[b,p]) / (B*P) and
    divergence = sum_b KL_b / B, where the only cross-sample coupling - the batch-global phrase slots (torch.unique over
    the half batch, halva_trainer.py:412) and the divisors - depends on integer inputs alone and is computed on the host
    before any forward.  The engine therefore runs forward + backward per GROUP of pairs (then per group of reference
    rows): activations of one group live at a time, no recomputation (the reference re-runs every layer under gradient
    checkpointing), gradients accumulate in the flat fp32 buffer;
  * pos/neg rows of a pair share one CLIP encode (the reference encodes the same image twice, halva_trainer.py:464);
  * the [rows, 32000] logits never reach HBM as fp32: lm_head is applied chunk-wise to the response rows only (rows whose
    shifted label is IGNORE_INDEX are multiplied by 0 in the reference, halva_trainer.py:556-557) and each bf16 chunk is
    consumed in place by the token-logp / KL kernels, which also emit the gradient w.r.t. the hidden state.
"""
import math
import os
import warnings

from types import SimpleNamespace

import numpy as np
import torch

from . import kernels as K
from . import splice as SP
from .hip import BF16, F32, call, ptr, stream_ptr

IGNORE_INDEX = -100
LOGIT_CHUNK_ROWS = 8192            # 8192 x 32000 bf16 = 0.5 GB of transient logits
# token log-probs from fp32 logits (the GEMM accumulates in fp32 anyway; this only skips the rounding of its output to bf16)
LOGITS_F32 = os.environ.get("HALVA_LOGITS_F32", "0") == "1"
# keep the response rows' bf16 logits from the forward for the backward (rows x vocab x 2 B: 2.9 GB for the 7B step's 45 k response
# rows - HBM is there) instead of recomputing them with one more lm_head GEMM per chunk; "0" = recompute (round 1's behaviour)
KEEP_LOGITS = os.environ.get("HALVA_KEEP_LOGITS", "1") != "0"


class _LmHeadLogp(torch.autograd.Function):
    """log p(target) for rows of hidden states: lm_head (modelling_llama.py:801-806) fused chunk-wise with
    log_softmax + gather (halva_trainer.py:406-407).  Backward turns the chunk's logits - kept from the forward (KEEP_LOGITS), or
    recomputed with one more lm_head GEMM (2 % of a sequence forward) - in place into dlogits, then dh = dlogits @ W."""

    @staticmethod
    def forward(ctx, h, W, target):
        R = h.shape[0]
        V = W.shape[0]
        logp = torch.empty(R, dtype=torch.float32, device=h.device)
        lse = torch.empty(R, dtype=torch.float32, device=h.device)
        st = stream_ptr()
        kept = [] if (KEEP_LOGITS and not LOGITS_F32 and ctx.needs_input_grad[0]) else None
        for c0 in range(0, R, LOGIT_CHUNK_ROWS):
            c1 = min(R, c0 + LOGIT_CHUNK_ROWS)
            if LOGITS_F32:
                logits = torch.mm(h[c0:c1], W.t(), out_dtype=torch.float32)
                call("halva_token_logp_fwd", ptr(logits), F32, V, ptr(target[c0:c1]), ptr(logp[c0:c1]), ptr(lse[c0:c1]), c1 - c0, V, st)
                continue
            logits = torch.mm(h[c0:c1], W.t())
            call("halva_token_logp_fwd", ptr(logits), BF16, V, ptr(target[c0:c1]), ptr(logp[c0:c1]), ptr(lse[c0:c1]), c1 - c0, V, st)
            if kept is not None:
                kept.append(logits)
        ctx.save_for_backward(h, target, lse)
        ctx.W = W
        ctx.kept = kept
        return logp

    @staticmethod
    def backward(ctx, g):
        h, target, lse = ctx.saved_tensors
        W = ctx.W
        V = W.shape[0]
        g = g.contiguous().float()
        dh = torch.empty_like(h)
        st = stream_ptr()
        kept, ctx.kept = ctx.kept, None
        for k, c0 in enumerate(range(0, h.shape[0], LOGIT_CHUNK_ROWS)):
            c1 = min(h.shape[0], c0 + LOGIT_CHUNK_ROWS)
            if LOGITS_F32:      # (diagnostic flag) the forward's lse came from fp32 logits: the softmax of the backward must too
                logits = torch.mm(h[c0:c1], W.t(), out_dtype=torch.float32)
                call("halva_token_logp_bwd", ptr(logits), F32, V, ptr(target[c0:c1]), ptr(lse[c0:c1]), ptr(g[c0:c1]), ptr(logits),
                     c1 - c0, V, st)
                torch.mm(logits.to(torch.bfloat16), W, out=dh[c0:c1])
                continue
            logits = kept[k] if kept is not None else torch.mm(h[c0:c1], W.t())      # (the same bf16 values either way)
            call("halva_token_logp_bwd", ptr(logits), BF16, V, ptr(target[c0:c1]), ptr(lse[c0:c1]), ptr(g[c0:c1]), ptr(logits),
                 c1 - c0, V, st)
            torch.mm(logits, W, out=dh[c0:c1])
            if kept is not None:
                kept[k] = None      # a chunk's 0.5 GB go back as soon as its dh exists
        return dh, None, None


class _LmHeadKL(torch.autograd.Function):
    """sum_rows KL(ref || policy) with both lm_heads fused (halva_trainer.py:583-588 before the /B).  The output is a
    scalar, so d/dh_pol is produced in the forward pass (one launch per chunk yields kl and p_pol - p_ref in place)."""

    @staticmethod
    def forward(ctx, h_pol, h_ref, W_pol, W_ref):
        R = h_pol.shape[0]
        V = W_pol.shape[0]
        kl = torch.empty(R, dtype=torch.float32, device=h_pol.device)
        need = h_pol.requires_grad
        dh = torch.empty_like(h_pol) if need else None
        st = stream_ptr()
        for c0 in range(0, R, LOGIT_CHUNK_ROWS):
            c1 = min(R, c0 + LOGIT_CHUNK_ROWS)
            lp = torch.mm(h_pol[c0:c1], W_pol.t())
            lr = torch.mm(h_ref[c0:c1], W_ref.t())
            call("halva_kl_rows", ptr(lp), ptr(lr), BF16, V, None, ptr(kl[c0:c1]), ptr(lp) if need else None, 1.0, c1 - c0, V, st)
            if need:
                torch.mm(lp, W_pol, out=dh[c0:c1])
        ctx.save_for_backward(dh)
        return kl.sum()

    @staticmethod
    def backward(ctx, g):
        (dh,) = ctx.saved_tensors
        return dh * g.float(), None, None, None          # 0-dim fp32 scale: fp32 op-math, bf16 result (no bf16-rounded scale)


def lm_head_logp(h_rows, lm_weight, target_i32):
    return _LmHeadLogp.apply(h_rows.contiguous(), lm_weight, target_i32)


def lm_head_kl(h_pol_rows, h_ref_rows, w_pol, w_ref):
    return _LmHeadKL.apply(h_pol_rows.contiguous(), h_ref_rows.contiguous(), w_pol, w_ref)


# ------------------------------------------------------------------------------------------------
# host-side batch bookkeeping
# ------------------------------------------------------------------------------------------------
def concat_pos_neg(batch):
    """halva_trainer.py:434-447 on host tensors: rows 0..B-1 = pos, B..2B-1 = neg; ids 0-filled, labels -100, signs 0."""
    ids, neg = _np(batch["input_ids"]), _np(batch["neg_input_ids"])
    B = ids.shape[0]
    W = max(ids.shape[1], neg.shape[1])
    c_ids = np.zeros((2 * B, W), dtype=np.int64)
    c_lab = np.full((2 * B, W), IGNORE_INDEX, dtype=np.int64)
    c_att = np.zeros((2 * B, W), dtype=bool)
    c_sig = np.zeros((2 * B, W), dtype=np.int64)
    for off, (i, l, a, s) in ((0, ("input_ids", "labels", "attention_mask", "pos_signs")),
                              (B, ("neg_input_ids", "neg_labels", "neg_attention_mask", "neg_signs"))):
        w = _np(batch[i]).shape[1]
        c_ids[off:off + B, :w] = _np(batch[i])
        c_lab[off:off + B, :w] = _np(batch[l])
        c_att[off:off + B, :w] = _np(batch[a])
        c_sig[off:off + B, :w] = _np(batch[s])
    return c_ids, c_lab, c_att, c_sig


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def phrase_slots(signs_shifted_half):
    """Slots of accumulate_logps (halva_trainer.py:412-415): sorted unique ids of the half batch minus the first one."""
    s = np.where(signs_shifted_half == IGNORE_INDEX, 0, signs_shifted_half)
    return np.unique(s)[1:].astype(np.int64)


class DPAStepPlan:
    """Everything about one micro-batch that is decidable from its integer tensors (host only).

    Images: the reference feeds `cat([images, images])` and lets the splice loop walk a running image index over the
    flattened list (llava_arch.py:277-330, vila/model/llava_arch.py:650-653,700-745); row r of the concatenated batch
    therefore reads slots that map back (mod n_images) onto the un-duplicated images, which is what lets each image be
    encoded once per pair.  n_images = B * images-per-sample."""

    def __init__(self, batch, n_patch, max_len, padding_side="right", n_images=None, imageless_consumes=True):
        c_ids, c_lab, c_att, c_sig = concat_pos_neg(batch)
        self.B = c_ids.shape[0] // 2
        self.cat = (c_ids, c_lab, c_att, c_sig)
        self.n_patch, self.max_len, self.side = n_patch, max_len, padding_side
        self.consumes = imageless_consumes
        B = self.B
        self.n_images = B if n_images is None else int(n_images)
        self.row_slots, used = SP.image_slots(c_ids, c_att, imageless_consumes)
        if used > 2 * self.n_images:
            raise IndexError("the batch has %d image tokens/slots but only %d images (x2 for pos/neg): index %d is out of "
                             "bounds" % (used, self.n_images, 2 * self.n_images))
        full = SP.plan_splice(c_ids, c_att, c_lab, c_sig, n_patch, max_len, padding_side,
                              image_map=[s % self.n_images for s in range(max(used, 1))],
                              imageless_consumes=imageless_consumes)
        sg = full.signs.numpy()[:, 1:]
        self.pos_slots = phrase_slots(sg[:B])
        self.neg_slots = phrase_slots(sg[B:])
        if len(self.pos_slots) != len(self.neg_slots):
            # the reference fails on `neg_logps_acc - pos_logps_acc` (halva_trainer.py:567) with a broadcast error
            raise RuntimeError("pos/neg halves have different numbers of phrase slots (%d vs %d): "
                               "The size of tensor a must match the size of tensor b" % (len(self.pos_slots), len(self.neg_slots)))
        self.P = len(self.pos_slots)
        self.T_full = full.T
        self.ref = tuple(_np(batch[k]) for k in ("ref_input_ids", "ref_labels", "ref_attention_mask"))
        self.ref_slots, used_ref = SP.image_slots(self.ref[0], self.ref[2], imageless_consumes)
        self.n_ref_images = used_ref

    def _local_map(self, slots_per_row, modulo):
        """(image_map for plan_splice's local running index, sorted unique image ids to encode)."""
        ids = sorted({s % modulo for sl in slots_per_row for s in sl})
        pos = {v: i for i, v in enumerate(ids)}
        local = []
        for sl in slots_per_row:
            if sl:
                local += [pos[s % modulo] for s in sl]
            elif self.consumes:
                local.append(0)
        return local or [0], ids

    def pair_group(self, idx):
        """Splice plan of the 2*len(idx) rows [pos(idx) ; neg(idx)] + the (un-duplicated) images those rows read."""
        c_ids, c_lab, c_att, c_sig = self.cat
        rows = list(idx) + [self.B + b for b in idx]
        local, images = self._local_map([self.row_slots[r] for r in rows], self.n_images)
        gp = SP.plan_splice(c_ids[rows], c_att[rows], c_lab[rows], c_sig[rows], self.n_patch, self.max_len, self.side,
                            image_map=local, imageless_consumes=self.consumes)
        gp.images = images
        return gp

    def ref_group(self, idx):
        ids, lab, att = self.ref
        idx = list(idx)
        local, images = self._local_map([self.ref_slots[r] for r in idx], max(self.n_ref_images, 1))
        gp = SP.plan_splice(ids[idx], att[idx], lab[idx], None, self.n_patch, self.max_len, self.side, image_map=local,
                            imageless_consumes=self.consumes)
        gp.images = images
        return gp


def _kept_rows(labels, row_of=None):
    """Rows (flattened [S, T-1] index into the [S, T] hidden states) whose shifted label is a real token.
    row_of (optional, [S, T]): flat index of the hidden state of (row, position) in another layout (packed pairs)."""
    lab = labels.numpy()
    S, T = lab.shape
    tgt = lab[:, 1:]
    s_idx, t_idx = np.nonzero(tgt != IGNORE_INDEX)
    flat = (s_idx * T + t_idx) if row_of is None else row_of[s_idx, t_idx]
    hid = torch.from_numpy(flat.astype(np.int64))                           # position t predicts label t+1
    dense = torch.from_numpy((s_idx * (T - 1) + t_idx).astype(np.int64))
    target = torch.from_numpy(tgt[s_idx, t_idx].astype(np.int32))
    return hid, dense, target


def _distinct_rows(hid, dev):
    """(uniq, inv) on the device: the distinct hidden rows `hid` names and, when a row is named twice (a packed pair's shared prefix
    row predicts a token of BOTH responses), the map back (rows = h_uniq[inv]); (None, None) for an empty `hid`."""
    if hid.numel() == 0:
        return None, None
    u, inv = np.unique(hid.numpy(), return_inverse=True)
    if u.size == hid.numel():
        u, order = hid.numpy(), None                # already distinct: keep the caller's order, no second gather
    else:
        order = torch.from_numpy(inv.astype(np.int64)).to(dev, non_blocking=True)
    return torch.from_numpy(np.ascontiguousarray(u)).to(dev, non_blocking=True), order


def model_spec(model):
    """What the step planner needs to know about a model wrapper: image tokens per image after the projector, the
    post-splice truncation length and padding side, and whether an image-less row advances the image index.
    LlavaLlamaForCausalLM (llava path) is described here; the VILA wrapper supplies its own `dpa_spec()`."""
    if hasattr(model, "dpa_spec"):
        return model.dpa_spec()
    cfg = model.config
    return SimpleNamespace(n_patch=model.get_vision_tower().num_patches,
                           max_len=getattr(cfg, "tokenizer_model_max_length", None),
                           padding_side=getattr(cfg, "tokenizer_padding_side", "right"), imageless_consumes=True)


class DPAEngine:
    """Forward/backward of the DPA loss on one GPU.  `policy` / `ref_model` are LlavaLlamaForCausalLM instances."""

    def __init__(self, policy, ref_model, loss_alpha, pairs_per_group=4, ref_rows_per_group=8, share_prefix=None):
        self.policy, self.ref_model, self.alpha = policy, ref_model, float(loss_alpha)
        self.pairs_per_group, self.ref_rows_per_group = pairs_per_group, ref_rows_per_group
        # run the common prefix of a pair (image + prompt + identical start of the response) once: the correct and the
        # hallucinated row are packed into one branched sequence (halva_amd/splice.py:pack_pairs).  HALVA_SHARE_PREFIX=0 runs
        # them as two rows like the reference does.
        # share_prefix: True = when it saves rows, "always" = even when the 64-row alignment padding eats the saving (tests)
        self.share_prefix = (os.environ.get("HALVA_SHARE_PREFIX", "1") != "0") if share_prefix is None else share_prefix
        self.last_packing = None
        self.last_layout = None
        self.last_top_rows = {}          # {"pairs" | "ref": (rows the top layer's row-wise half ran on, rows of the pass)} (bench accounting)
        from .gemm_tuning import enable_tuned_gemms
        self.gemm_table = enable_tuned_gemms()      # measured hipBLASLt / rocBLAS kernel choices for the step's large matmuls

    @property
    def spec(self):
        return model_spec(self.policy)

    # -- pieces ------------------------------------------------------------------------------------
    def _images(self, batch, key, idx, dev):
        """Flat images `idx` of batch[key]: [B,3,H,W], [B,n,3,H,W] (VILA, flattened like llava_arch.py:650-653) or a list."""
        im = batch[key]
        if isinstance(im, (list, tuple)):
            flat = [x for t in im for x in (t if t.ndim == 4 else t[None])]
            im = torch.stack([flat[i] for i in idx])
        else:
            if im.ndim == 5:
                im = im.flatten(0, 1)
            im = im[list(idx)]
        return im.to(dev, torch.bfloat16, non_blocking=True)

    def _encode(self, model, batch, key, gp, dev):
        if not gp.images:                       # text-only group: a zero-row feature tensor keeps the gather well-formed
            return torch.zeros(0, self.spec.n_patch, model.lm_head.weight.shape[1], dtype=torch.bfloat16, device=dev)
        return model.encode_images(self._images(batch, key, gp.images, dev))

    def _hidden(self, model, plan, feats, rows=None):
        m = model.get_model()
        dev = m.embed_tokens.weight.device
        embeds = K.splice_rows(m.embed_tokens.weight, feats, plan.src, plan.S, plan.T)
        return model.hidden_states(embeds, None, plan.seq_start, plan.seq_len, rows=rows)

    def pair_group_loss(self, batch, plan, idx):
        """Contribution of the pairs `idx` to the alignment loss (already divided by B*P)."""
        pol = self.policy
        dev = pol.device
        gp = plan.pair_group(idx)
        g = len(idx)
        feats = self._encode(pol, batch, "images", gp, dev)                              # [g, n_patch, d]; grads -> projector
        packed = SP.pack_pairs(gp) if (self.share_prefix and gp.seq_start.numpy().max(initial=0) == 0) else None
        if packed is not None and (packed.rows_packed < packed.rows_unpacked or self.share_prefix == "always"):
            # the two rows of a pair run as ONE packed row [prefix | correct rest | pad | hallucinated rest]
            m = pol.get_model()
            embeds = K.splice_rows(m.embed_tokens.weight, feats, packed.src, g, packed.T)
            branch = tuple(t.to(dev, non_blocking=True) for t in (packed.br_a, packed.br_b, packed.pos))
            hid, dense, target = _kept_rows(gp.labels, packed.row_of)
            uniq, inv = _distinct_rows(hid, dev)
            h = pol.hidden_states(embeds, None, torch.zeros(g, dtype=torch.int32), packed.seq_len, branch=branch, rows=uniq)
            self.last_top_rows["pairs"] = (g * packed.T if uniq is None else int(uniq.numel()), g * packed.T)
            self.last_packing = (packed.rows_packed, packed.rows_unpacked)
            self.last_layout = (packed.T, packed.br_a.tolist(), packed.br_b.tolist(), packed.seq_len.tolist())      # host ints (bench accounting)
        else:
            hid, dense, target = _kept_rows(gp.labels)
            uniq, inv = _distinct_rows(hid, dev)
            h = self._hidden(pol, gp, feats, uniq)
            self.last_top_rows["pairs"] = (2 * g * gp.T if uniq is None else int(uniq.numel()), 2 * g * gp.T)
        T1 = gp.T - 1
        logp_dense = torch.zeros(2 * g * T1, dtype=torch.float32, device=dev)
        if hid.numel():
            rows = h if inv is None else h.index_select(0, inv)      # h: the rows `uniq` only (the top layer computed nothing else)
            lp = lm_head_logp(rows, pol.lm_head.weight, target.to(dev, non_blocking=True))
            logp_dense = logp_dense.index_copy(0, dense.to(dev, non_blocking=True), lp)
        logp_dense = logp_dense.view(2 * g, T1)
        lab = gp.labels[:, 1:].contiguous().to(dev, non_blocking=True)
        sgn = gp.signs[:, 1:].contiguous().to(dev, non_blocking=True)
        if plan.P == 0:
            warnings.warn("no phrase slots in this batch: the reference's alignment loss is mean(empty) = NaN")
            return logp_dense.sum() * float("nan"), logp_dense, gp
        pos_acc = K.phrase_sum(logp_dense[:g].contiguous(), lab[:g].contiguous(), sgn[:g].contiguous(),
                               torch.from_numpy(plan.pos_slots).to(dev))
        neg_acc = K.phrase_sum(logp_dense[g:].contiguous(), lab[g:].contiguous(), sgn[g:].contiguous(),
                               torch.from_numpy(plan.neg_slots).to(dev))
        contrib = torch.log(1 + torch.exp(neg_acc - pos_acc)).sum() / (plan.B * plan.P)        # halva_trainer.py:567-568
        return contrib, (logp_dense, pos_acc, neg_acc), gp

    def ref_group_loss(self, batch, plan, idx):
        """Contribution of reference rows `idx` to loss_alpha * divergence (KL summed over tokens and vocab, / B)."""
        pol, ref = self.policy, self.ref_model
        dev = pol.device
        gp = plan.ref_group(idx)
        hid, _, _ = _kept_rows(gp.labels)                      # (one layout, one label per row: distinct)
        sel = hid.to(dev, non_blocking=True) if hid.numel() else None
        h_pol = self._hidden(pol, gp, self._encode(pol, batch, "ref_images", gp, dev), sel)
        self.last_top_rows["ref"] = (gp.S * gp.T if sel is None else int(sel.numel()), gp.S * gp.T)
        if sel is None:
            return h_pol.sum() * 0.0, gp
        with torch.no_grad():
            h_ref = self._hidden(ref, gp, self._encode(ref, batch, "ref_images", gp, dev), sel)
        kl = lm_head_kl(h_pol, h_ref, pol.lm_head.weight, ref.lm_head.weight)      # [n, d] each: the kept rows, in `hid` order
        return self.alpha * kl / plan.B, gp

    # -- whole micro-batch -------------------------------------------------------------------------
    def make_plan(self, batch):
        sp = self.spec
        im = batch["images"]
        if isinstance(im, (list, tuple)):
            n_images = sum(1 if t.ndim == 3 else t.shape[0] for t in im)
        else:
            n_images = im.shape[0] * (im.shape[1] if im.ndim == 5 else 1)
        return DPAStepPlan(batch, sp.n_patch, sp.max_len, sp.padding_side, n_images, sp.imageless_consumes)

    def _groups(self, n, per):
        return [list(range(i, min(n, i + per))) for i in range(0, n, per)]

    def loss(self, batch, backward=False, scale=1.0, reducer=None):
        """compute_loss of the reference (halva_trainer.py:534-592).  backward=True runs loss.backward() group by group
        (gradients accumulate; activations of one group alive at a time) and returns the detached loss value.
        reducer (halva_amd.dp.GradReducer, with backward=True): this is the LAST micro-batch before the optimizer step - the
        data-parallel gradient exchange is started from inside the last group's backward, layer bucket by layer bucket; the
        caller finishes it with reducer.finish()."""
        plan = self.make_plan(batch)
        if reducer is not None:
            reducer.begin()
        total = None
        parts = {"alignment": None, "divergence": None}
        for idx in self._groups(plan.B, self.pairs_per_group):
            c, _, _ = self.pair_group_loss(batch, plan, idx)
            if backward:
                (c * scale).backward()
                c = c.detach()
            parts["alignment"] = c if parts["alignment"] is None else parts["alignment"] + c
        ref_groups = self._groups(plan.B, self.ref_rows_per_group)
        for gi, idx in enumerate(ref_groups):
            armed = backward and reducer is not None and gi == len(ref_groups) - 1
            lm = _llm_of(self.policy)[0].model
            if armed:
                lm.grad_ready_hook = reducer.layer_done
            try:
                c, _ = self.ref_group_loss(batch, plan, idx)
            finally:
                lm.grad_ready_hook = None
            if backward:
                (c * scale).backward()
                c = c.detach()
            parts["divergence"] = c if parts["divergence"] is None else parts["divergence"] + c
        total = parts["alignment"] + parts["divergence"]
        self.last_parts = {"alignment": parts["alignment"].detach(),
                           "divergence": (parts["divergence"].detach() / self.alpha) if self.alpha else parts["divergence"].detach()}
        return total


# ------------------------------------------------------------------------------------------------
# trainable parameters as one flat buffer: bf16 compute copy, fp32 master, fp32 gradient accumulator
# ------------------------------------------------------------------------------------------------
class FlatTrainables:
    """DeepSpeed-bf16 numerics without DeepSpeed: bf16 parameters used in compute, fp32 master weights and fp32 Adam
    moments (reference --bf16 True + ZeRO), and an fp32 gradient accumulator that the kernels' backward
    adds into directly.  The accumulator is THE buffer all-reduced across ranks (halva_amd/dp.py)."""

    def __init__(self, named_params):
        named_params = [(n, p) for n, p in named_params if p.requires_grad]
        if not named_params:
            raise ValueError("no trainable parameters")
        dev = named_params[0][1].device
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.no_decay = {n for n, p in named_params if getattr(p, "no_decay", "bias" in n)}
        sizes = [p.numel() for p in self.params]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        total = int(self.offsets[-1])
        self.flat = torch.empty(total, dtype=torch.bfloat16, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o, n in zip(self.params, self.offsets[:-1], sizes):
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view_as(p)
            p.main_grad = self.grad[o:o + n].view_as(p)
            p.grad_sink = True
        self.master = self.flat.float()
        self.total = total
        self.post_sync = []          # callbacks run after the bf16 compute copy was refreshed from the fp32 master

    def segment(self, pred):
        """Contiguous [lo, hi) covering the parameters whose name satisfies pred (they must be adjacent)."""
        idx = [i for i, n in enumerate(self.names) if pred(n)]
        if not idx:
            return None
        assert idx == list(range(idx[0], idx[-1] + 1)), "segment parameters must be adjacent in the flat buffer"
        return int(self.offsets[idx[0]]), int(self.offsets[idx[-1] + 1])

    def zero_grad(self):
        self.grad.zero_()

    def sync_compute_copy(self):
        self.flat.copy_(self.master)
        for fn in self.post_sync:
            fn()


def _llm_of(model):
    """(causal LM holding `.model.layers`, name prefix): the LLaVA wrapper IS the LM, VILA keeps it under `.llm`."""
    return (model.get_llm(), "llm.") if hasattr(model, "get_llm") else (model, "")


def _projector_of(model):
    if hasattr(model, "get_mm_projector"):
        return model.get_mm_projector(), "mm_projector."
    return getattr(model.get_model(), "mm_projector", None), "model.mm_projector."


def bind_model(flat, model):
    """After every optimizer step the fused LoRA weights of `model` must pick up the new B factors."""
    from .llama import refresh_lora
    lm, _ = _llm_of(model)
    flat.post_sync.append(lambda: refresh_lora(lm))
    refresh_lora(lm)


def set_grad_sink(model, on=True):
    """Route LoRA gradients into `.main_grad` (True) or return them through autograd (False)."""
    for layer in model.get_model().layers:
        if hasattr(layer, "groups"):
            for _, grp in layer.groups():
                grp.grad_sink = on
    proj, _ = _projector_of(model)
    if proj is not None:
        for p in proj.parameters():
            p.grad_sink = on


def trainable_named_parameters(model):
    """Trainable tensors in flat-buffer order: LoRA factors, then the projector's decayed tensors (Linear weights), then
    its un-decayed ones (biases and LayerNorm weights - HF's get_parameter_names(model, ALL_LAYERNORM_LAYERS) split)."""
    from .llama import lora_named_parameters
    lm, pre = _llm_of(model)
    out = [(pre + n, p) for n, p in lora_named_parameters(lm)]
    proj, ppre = _projector_of(model)
    if proj is not None:
        ln = {id(p) for m in proj.modules() if isinstance(m, torch.nn.LayerNorm) for p in m.parameters()}
        named = [(ppre + n, p) for n, p in proj.named_parameters() if p.requires_grad]
        for n, p in named:
            p.no_decay = ("bias" in n) or (id(p) in ln)
        out += [x for x in named if not x[1].no_decay] + [x for x in named if x[1].no_decay]
    return out


class AdamWFlat:
    """torch.optim.AdamW (reference optim="adamw_torch", llava/train/train_halva.py:70) on the fp32 master slices,
    with the reference's parameter groups: {decay, no-decay} x {projector (lr = mm_projector_lr), rest}
    (halva_trainer.py:291-337).  "decay" = neither a bias nor a LayerNorm weight (the latter only exists in VILA's mlp_downsample)."""

    def __init__(self, flat, lr, weight_decay=0.0, mm_projector_lr=None, betas=(0.9, 0.999), eps=1e-8):
        self.flat = flat
        groups = []
        segs = [("lora", lambda n: "mm_projector" not in n, lr, weight_decay),
                ("proj_w", lambda n: "mm_projector" in n and n not in flat.no_decay,
                 lr if mm_projector_lr is None else mm_projector_lr, weight_decay),
                ("proj_b", lambda n: "mm_projector" in n and n in flat.no_decay,
                 lr if mm_projector_lr is None else mm_projector_lr, 0.0)]
        self._views = []
        for name, pred, g_lr, wd in segs:
            s = flat.segment(pred)
            if s is None:
                continue
            p = torch.nn.Parameter(flat.master[s[0]:s[1]])
            p.grad = flat.grad[s[0]:s[1]]
            groups.append({"params": [p], "lr": g_lr, "weight_decay": wd, "name": name})
        self.opt = torch.optim.AdamW(groups, lr=lr, betas=betas, eps=eps)
        for g in self.opt.param_groups:
            g["initial_lr"] = g["lr"]

    def set_lr_factor(self, f):
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * f

    def step(self):
        self.opt.step()
        self.flat.sync_compute_copy()

    def state_dict(self):
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        """Restore step counts and moments (a checkpoint written by state_dict()); the group learning rates stay the run's."""
        lrs = [(g["lr"], g["initial_lr"]) for g in self.opt.param_groups]
        self.opt.load_state_dict(sd)
        for g, (lr, ilr) in zip(self.opt.param_groups, lrs):
            g["lr"], g["initial_lr"] = lr, ilr


def cosine_with_warmup(step, total_steps, warmup_ratio):
    """HF get_cosine_schedule_with_warmup with num_warmup_steps = ceil(warmup_ratio * total_steps)."""
    warm = math.ceil(warmup_ratio * total_steps)
    if step < warm:
        return float(step) / float(max(1, warm))
    prog = min(1.0, float(step - warm) / float(max(1, total_steps - warm)))      # past the end the rate stays at its floor
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
