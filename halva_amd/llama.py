"""Llama decoder with LoRA for the DPA step - the MI355X-native stand-in for HF LlamaForCausalLM + peft LoRA +
the flash-attn monkey patch of the reference (llava/model/language_model/llava_llama.py:42-85 ->
transformers 4.31 LlamaForCausalLM; numerics spec: llava/model/language_model/modelling_llama.py).

MI355X-first choices:
  * frozen base weights are stored fused ([q;k;v] and [gate;up]) so each decoder layer is 4 large hipBLASLt GEMMs
    (PyTorch-ROCm) + 4 skinny LoRA GEMM pairs; residual adds ride in the GEMM epilogue (addmm, beta = 1);
  * RMSNorm, RoPE, causal attention (MFMA), SwiGLU run as hand-written HIP kernels through the C ABI;
  * LoRA/projector gradients accumulate straight into one flat fp32 buffer (`main_grad`) - the buffer the
    data-parallel all-reduce and AdamW operate on - instead of per-parameter bf16 `.grad` tensors.
No CPU fallback: every forward reaches libhalva_hip.so.
"""
import json
import math
import os

import torch
import torch.nn as nn

from . import kernels as K


class LlamaConfig:
    """Minimal HF-compatible config (reads/writes config.json; unknown keys are kept)."""
    model_type = "llama"
    _defaults = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                     num_attention_heads=32, num_key_value_heads=None, hidden_act="silu", max_position_embeddings=4096,
                     rms_norm_eps=1e-5, rope_theta=10000.0, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                     pretraining_tp=1, use_cache=True, tie_word_embeddings=False)

    def __init__(self, **kw):
        for k, v in self._defaults.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if not k.startswith("_") and _jsonable(v)}
        d["model_type"] = self.model_type
        return d

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, path, **kw):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.update(kw)
        return cls(**d)


def _jsonable(v):
    try:
        json.dumps(v)
        return True
    except TypeError:
        return False


# Keep a transposed copy of every fused frozen weight for the backward (dx = dy W as an NT GEMM).  Costs the frozen weights'
# memory once more (13.5 GB at 7B, 26 GB at 13B - of 288 GB); HALVA_DGRAD_WT=0 turns it off.
DGRAD_TRANSPOSED_COPY = os.environ.get("HALVA_DGRAD_WT", "1") != "0"
WGRAD_KERNEL = os.environ.get("HALVA_WGRAD_KERNEL", "1") != "0"      # LoRA weight gradients through halva_wgrad_accumulate
WGRAD_BATCH = os.environ.get("HALVA_WGRAD_BATCH", "1") != "0"        # ... the factors of a group as ONE launch pair (round 6; 0 = one pair per factor, bitwise the same)
RES_INPLACE = os.environ.get("HALVA_RES_INPLACE", "1") != "0"        # residual adds accumulate onto the block's own buffer (A/B: 0)
# dgrad through the MERGED weight: the transposed copy holds (W + scale * B A)^T, so dx = dy (W + scale B A) comes out of the one
# dgrad GEMM complete and the separate dx += (scale * dy B) A pass over [rows, in] is gone (A/B: 0)
DGRAD_MERGED = os.environ.get("HALVA_DGRAD_MERGED", "1") != "0"
# diagnostic (tools/diag_long_fixture.py): the LoRA update as peft computes it - base product rounded to bf16, low-rank product rounded to
# bf16, then added - instead of riding in the K-concatenated GEMM's fp32 accumulator (forward values only; the backward is unchanged)
LORA_TWO_GEMM = os.environ.get("HALVA_LORA_TWO_GEMM", "0") == "1"
K_ = K      # the kernels module under a name that _LoraGroupFn's local `K` (in_features) does not shadow

LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


class _LoraGroupFn(torch.autograd.Function):
    """y = [residual +] x W^T + scale * concat_g( (x A_g^T) B_g^T )  for G LoRA targets sharing one fused base weight -
    peft 0.4.0 Linear.forward (result += lora_B(lora_A(dropout(x))) * scaling; dropout forced to 0 by the reference,
    llava/train/halva_trainer.py:35-38,184-187) - computed as ONE GEMM over a K-concatenated operand:

        xa = [ x | x A_cat^T ]  (rows x (K + G r));   Wc = [ W | blockdiag(scale * B_g) ]  (N x (K + G r));   y = xa Wc^T

    so the low-rank update rides in the big GEMM instead of a read-modify-write pass over y (measured 0.19-0.38 ms per
    projection at 16k tokens, ~25 % on top of the base GEMMs).  The producer kernel (RMSNorm / attention / SwiGLU) wrote x
    straight into the left K columns of xa.  Backward: dxa = dy Wc gives [dx_base | scale * dy_g B_g] in one GEMM.
    Base weight frozen: gradients for x, A, B only; with `sink` they are added into the fp32 `main_grad` views."""

    @staticmethod
    def forward(ctx, xa, residual, Wc, WcT, A, scale, sink, K, res_inplace, merged, *Bs):
        """res_inplace: `residual` is a buffer of this block's own (the copy kernels._RMSNormFork wrote, or - without autograd - the
        previous block's output): the product is accumulated ONTO it (beta = 1, C == D) and it is returned, instead of copying it
        into a fresh output first (what addmm(out=) does when out is another tensor: a [rows, d] device copy per residual add)."""
        width = xa.shape[-1]
        xa2 = xa.view(-1, width)
        N = Wc.shape[0]
        res_inplace = bool(res_inplace) and residual is not None and residual.is_contiguous()
        if res_inplace:
            out = residual
            ctx.mark_dirty(residual)
        else:
            out = torch.empty(*xa.shape[:-1], N, dtype=xa.dtype, device=xa.device)   # returned as-is (not a view): later
        y = out.view(-1, N)                                                            # in-place kernels (RoPE) may dirty it
        lora = A is not None
        if lora:
            Gr = A.shape[0]
            torch.mm(xa2[:, :K], A.t(), out=xa2[:, K:K + Gr])      # written in place through the row stride (no temporary)
            if K + Gr < width:
                xa2[:, K + Gr:].zero_()          # padding columns (rank not a multiple of 8) must not hold NaN garbage
            lhs, rhs = (xa2[:, :K], Wc[:, :K]) if LORA_TWO_GEMM else (xa2, Wc)
        else:
            lhs, rhs = xa2[:, :K], Wc[:, :K]
        if residual is None:
            torch.mm(lhs, rhs.t(), out=y)
        elif res_inplace:
            y.addmm_(lhs, rhs.t())
        else:
            torch.addmm(residual.reshape(-1, N), lhs, rhs.t(), out=y)
        if lora and LORA_TWO_GEMM:
            y.add_(torch.mm(xa2[:, K:K + Gr], Wc[:, K:K + Gr].t()))
        ctx.save_for_backward(xa)          # the input itself (its right columns were filled above), not the internal view
        ctx.params = (Wc, WcT, A, Bs)      # long-lived parameters: kept as objects so `.main_grad` stays reachable
        ctx.meta = (scale, sink, residual is not None, K, xa.shape, bool(merged) and WcT is not None)
        return out

    @staticmethod
    def backward(ctx, dy):
        (xa,) = ctx.saved_tensors
        Wc, WcT, A, Bs = ctx.params
        scale, sink, has_res, K, xa_shape, merged = ctx.meta
        xa2 = xa.view(-1, xa.shape[-1])
        N = Wc.shape[0]
        dy2 = dy.reshape(-1, N)
        dA = None
        dBs = [None] * len(Bs)
        if A is not None:
            r = A.shape[0] // len(Bs)
            # [rows, K + G r] = [dx through W | scale * dy_g B_g].  With the transposed copy the product runs in hipBLASLt's
            # NT form (both operands K-major), measured 12 % faster than NN at these shapes on MI355X.
            dxa = torch.mm(dy2, WcT.t()) if WcT is not None else torch.mm(dy2, Wc)
            da = dxa[:, K:K + A.shape[0]]
            # Weight gradients of the factors: one side of each product is only r..G r wide and the contraction runs over every token
            # row, a shape the library serves with 32-86 tiles.  With an f32 sink they go through the split-k kernel
            # (halva_wgrad_accumulate: 76 vs 136 us for a [4096 x 128] factor at 27 k rows), straight into main_grad.
            fused = (sink and WGRAD_KERNEL and all(t.main_grad.is_contiguous() and t.main_grad.data_ptr() % 16 == 0 and
                                                   t.main_grad.dtype == torch.float32 for t in (A, *Bs))
                     and K_.wgrad_supported(da, xa2[:, :K]) and K_.wgrad_supported(dy2[:, :Bs[0].shape[0]], xa2[:, K:K + r]))
            batch = [] if (fused and WGRAD_BATCH and len(Bs) <= 3) else None      # (the A factor + up to three B factors: one launch pair)
            if batch is not None:
                batch.append((A.main_grad, da, xa2[:, :K], 1.0))
            elif fused:
                K_.wgrad_accumulate(A.main_grad, da, xa2[:, :K], 1.0)
            else:
                gA = torch.mm(da.t(), xa2[:, :K])
            off = 0
            for g, B in enumerate(Bs):
                n = B.shape[0]
                if batch is not None:
                    batch.append((B.main_grad, dy2[:, off:off + n], xa2[:, K + g * r:K + (g + 1) * r], scale))
                elif fused:
                    K_.wgrad_accumulate(B.main_grad, dy2[:, off:off + n], xa2[:, K + g * r:K + (g + 1) * r], scale)
                else:
                    gB = torch.mm(dy2[:, off:off + n].t(), xa2[:, K + g * r:K + (g + 1) * r])
                    if sink:
                        B.main_grad.add_(gB, alpha=scale)
                    else:
                        dBs[g] = gB * scale
                off += n
            if batch is not None:
                K_.wgrad_accumulate_batch(batch)
            if fused:
                pass
            elif sink:
                A.main_grad.add_(gA)
            else:
                dA = gA
            if not merged:                                  # (merged: WcT's base rows already hold (W + scale B A)^T, LoraGroup.refresh_tail)
                dxa[:, :K].addmm_(da, A)                    # + the LoRA path's contribution to dx, in place
        else:
            dxa = torch.zeros(dy2.shape[0], xa2.shape[1], dtype=dy2.dtype, device=dy2.device)
            dxa[:, :K].copy_(torch.mm(dy2, Wc[:, :K]))
        return (dxa.view(xa_shape), dy if has_res else None, None, None, dA, None, None, None, None, None, *dBs)


class LoraTarget(nn.Module):
    """Holder of one target's LoRA factors, named like peft (`<target>.lora_A.default.weight`)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.lora_A = nn.ModuleDict()
        self.lora_B = nn.ModuleDict()


class _W(nn.Module):
    def __init__(self, t):
        super().__init__()
        self.weight = nn.Parameter(t)


class RMSNormW(nn.Module):
    def __init__(self, d, eps, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d, dtype=dtype, device=device), requires_grad=False)
        self.variance_epsilon = eps

    def forward(self, x, out_width=None):
        return K.rmsnorm(x, self.weight, self.variance_epsilon, out_width)

    def fork(self, x, out_width=None, own_x=False):
        """(norm(x), x', private) for a pre-norm residual block: hand x' to the residual add (kernels._RMSNormFork).  `private` says
        that x' is a buffer the block may accumulate onto in place: the copy the fork kernel wrote (autograd path), or x itself when
        the caller owns it and nothing will ask for it again (`own_x`, no autograd)."""
        grad = torch.is_grad_enabled() and x.requires_grad
        inplace = RES_INPLACE
        if not grad or os.environ.get("HALVA_NORM_FORK", "1") == "0":
            return K.rmsnorm(x, self.weight, self.variance_epsilon, out_width), x, (inplace and own_x and not grad)
        h, xc = K.rmsnorm_fork(x, self.weight, self.variance_epsilon, out_width)
        return h, xc, inplace


class LoraGroup(nn.Module):
    """A fused frozen weight [sum(out_g), in] plus the LoRA factors of its G targets.

    Storage: `weight_cat` is [N, in + G*r]: columns [0, in) hold the frozen base weight (`.weight` is that strided view),
    the tail holds blockdiag(scale * B_g) - refreshed from the trainable B parameters whenever they changed (their tensor
    version counter moves on every optimizer step)."""

    def __init__(self, names, in_features, outs, dtype, device):
        super().__init__()
        self.names, self.outs, self.in_features = tuple(names), tuple(outs), in_features
        # a row stride of exactly 4096 / 8192 elements costs hipBLASLt ~13 % on MI355X (measured: N=12288, K=4096 at 1361 TFLOP/s
        # contiguous vs 1571 through a view with row stride 4480): give LoRA-free weights (the reference model) 64 spare columns
        pad = 64 if in_features % 1024 == 0 else 0
        self.weight_cat = nn.Parameter(torch.zeros(sum(outs), in_features + pad, dtype=dtype, device=device), requires_grad=False)
        for n, o in zip(names, outs):
            setattr(self, n, LoraTarget(in_features, o))
        self.A_cat = None          # [G*r, in]: the A factors of the G targets, one GEMM for all of them
        self.weight_cat_t = None   # [in + G*r, N] copy of weight_cat for the dgrad GEMM (built with the LoRA factors)
        self.scale = 0.0
        self.r = 0
        self.grad_sink = False
        self._tail_versions = None

    @property
    def weight(self):
        return self.weight_cat[:, :self.in_features]

    @property
    def in_width(self):
        """Width of the operand buffer the producer kernel must allocate (in + G*r once LoRA is attached)."""
        return self.weight_cat.shape[1]

    def targets(self):
        return [getattr(self, n) for n in self.names]

    def _Bs(self):
        return [getattr(self, n).lora_B["default"].weight for n in self.names]

    def attach_lora(self, r, alpha, dtype, device, generator=None):
        """peft 0.4.0 init: A ~ kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)), B = 0, scaling = alpha / r."""
        G = len(self.names)
        A_all = torch.empty(G * r, self.in_features, dtype=torch.float32, device=device)
        bound = 1.0 / math.sqrt(self.in_features)
        A_all.uniform_(-bound, bound, generator=generator)
        self.A_cat = nn.Parameter(A_all.to(dtype))
        for n, o in zip(self.names, self.outs):
            getattr(self, n).lora_B["default"] = _W(torch.zeros(o, r, dtype=dtype, device=device))
        self.scale = float(alpha) / float(r)
        self.r = r
        base = self.weight_cat.data
        tail = (G * r + 7) // 8 * 8          # 16-byte rows for the producer kernels (r = 128 needs no padding)
        wc = torch.zeros(base.shape[0], self.in_features + tail, dtype=base.dtype, device=base.device)
        wc[:, :self.in_features].copy_(base[:, :self.in_features])
        self.weight_cat = nn.Parameter(wc, requires_grad=False)
        self.weight_cat_t = None
        self._tail_versions = None

    def build_dgrad_copy(self):
        """Second, transposed copy of the fused weight ([in + G r, N], +1x the frozen weights' memory): dx = dy Wc then runs
        as an NT GEMM.  The LoRA tail rows are refreshed together with weight_cat's tail columns; with DGRAD_MERGED its base rows
        hold the merged weight (W + scale B A)^T (refresh_tail)."""
        self.weight_cat_t = self.weight_cat.data.t().contiguous()
        self._tail_versions = None

    @property
    def dgrad_merged(self):
        return DGRAD_MERGED and self.weight_cat_t is not None and self.A_cat is not None

    def refresh_tail(self):
        """weight_cat[:, in:] = blockdiag(scale * B_g) (bf16; scale = alpha / r is a power of two in the reference recipe)."""
        K, r = self.in_features, self.r
        off = 0
        with torch.no_grad():
            for g, (B, n) in enumerate(zip(self._Bs(), self.outs)):
                sb = B * self.scale
                self.weight_cat[off:off + n, K + g * r:K + (g + 1) * r].copy_(sb)
                if self.weight_cat_t is not None:
                    self.weight_cat_t[K + g * r:K + (g + 1) * r, off:off + n].copy_(sb.t())
                off += n
            if self.dgrad_merged:
                # rows [0, in) of the transposed copy <- (W + scale * B A)^T = W^T + A_cat^T blockdiag(scale B)^T: one [in x G r] x
                # [G r x N] GEMM per group and optimizer step (3 TFLOP and 26 GB of traffic over the whole 7B model, ~8 ms) against
                # a read-modify-write pass over [rows, in] per group and backward (~56 ms per step).  fp32 accumulate, ONE rounding
                # to bf16 - dx differs from the two-GEMM form by bf16 noise; dA, dB are formed from the same operands as before.
                G = len(self.names)
                # (W^T by the tiled transpose kernel, then the product accumulated in place: torch.addmm(W.t(), ..., out=) copied the strided
                # W^T into `out` with the framework's element-wise copy first - 0.45 ms per group, 30 ms per step over the model)
                K_.transpose_into(self.weight_cat_t[:K], self.weight_cat[:, :K])
                self.weight_cat_t[:K].addmm_(self.A_cat.data.t(), self.weight_cat_t[K:K + G * r])
        self._tail_versions = tuple(B._version for B in self._Bs()) + (self.A_cat._version,)

    def lora_state(self):
        """{peft-style name: tensor} for this group's targets."""
        out = {}
        if self.A_cat is None:
            return out
        r = self.r
        for g, n in enumerate(self.names):
            out[n + ".lora_A.default.weight"] = self.A_cat.data[g * r:(g + 1) * r]
            out[n + ".lora_B.default.weight"] = getattr(self, n).lora_B["default"].weight.data
        return out

    def forward(self, xa, residual=None, use_lora=True, res_inplace=False):
        """xa: [.., in_width] operand buffer whose left `in` columns hold the input (right columns: scratch)."""
        if xa.shape[-1] != self.in_width:
            raise ValueError("LoraGroup expects an operand buffer of width %d, got %d" % (self.in_width, xa.shape[-1]))
        if self.A_cat is not None and use_lora:
            Bs = self._Bs()
            if self._tail_versions != tuple(B._version for B in Bs) + (self.A_cat._version,):
                self.refresh_tail()
            return _LoraGroupFn.apply(xa, residual, self.weight_cat, self.weight_cat_t, self.A_cat, self.scale, self.grad_sink,
                                      self.in_features, res_inplace, self.dgrad_merged, *Bs)
        return _LoraGroupFn.apply(xa, residual, self.weight_cat, None, None, 0.0, False, self.in_features, res_inplace, False)


class SeqInfo:
    """Per-forward constants shared by all layers: RoPE tables and the valid span of every padded row."""

    def __init__(self, cos, sin, seq_start, seq_len, branch=None):
        self.cos, self.sin, self.seq_start, self.seq_len = cos, sin, seq_start, seq_len
        self.branch = branch       # (br_a, br_b, pos) of branch-packed rows, or None (halva_amd/kernels.py:_SdpaCausal)


class _TakeRows(torch.autograd.Function):
    """x[.., W] viewed as [rows, W] -> x[idx]; idx DISTINCT, so the backward is a copy into zeros (index_select's own backward is an
    index_add: atomics on bf16)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return x.view(-1, x.shape[-1]).index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        full = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        full.view(-1, full.shape[-1]).index_copy_(0, idx, g)
        return full, None


def _take_rows(x, idx):
    return _TakeRows.apply(x, idx)


# HALVA_TOP_ROWS=0: the top decoder layer runs on every row (A/B switch; results are identical row for row)
TOP_ROWS = os.environ.get("HALVA_TOP_ROWS", "1") != "0"


class DecoderLayer(nn.Module):
    """LlamaDecoderLayer (modelling_llama.py:352-420) with attention per llama_flash_attn_monkey_patch.py:16-93."""

    def __init__(self, cfg, dtype, device):
        super().__init__()
        d, Fd = cfg.hidden_size, cfg.intermediate_size
        self.H = cfg.num_attention_heads
        self.D = d // self.H
        if cfg.num_key_value_heads != cfg.num_attention_heads:
            raise NotImplementedError("grouped-query attention is not on the LLaVA-1.5 path (kv heads == heads)")
        self.qkv = LoraGroup(("q_proj", "k_proj", "v_proj"), d, (d, d, d), dtype, device)
        self.o = LoraGroup(("o_proj",), d, (d,), dtype, device)
        self.gate_up = LoraGroup(("gate_proj", "up_proj"), d, (Fd, Fd), dtype, device)
        self.down = LoraGroup(("down_proj",), Fd, (d,), dtype, device)
        self.input_layernorm = RMSNormW(d, cfg.rms_norm_eps, dtype, device)
        self.post_attention_layernorm = RMSNormW(d, cfg.rms_norm_eps, dtype, device)

    def groups(self):
        return (("self_attn", self.qkv), ("self_attn", self.o), ("mlp", self.gate_up), ("mlp", self.down))

    def forward(self, x, info, use_lora=True, own_x=False, rows=None):
        """own_x: the caller will not look at `x` again (it is the previous layer's output): without autograd the residual adds then
        run in place on it.
        rows: int64 [n] distinct flat row indices into [S * T] (the TOP layer only, LlamaModel.run_layers): everything behind the
        attention is row-wise, so the o projection, the post-attention norm and the MLP run on those rows alone and [n, d] is returned."""
        # every producer kernel writes straight into the (wider) operand buffer of the projection that follows it
        # (norm(x), x') come out of one autograd node so that the residual's gradient is added inside the norm's backward kernel, and
        # x' is a buffer of the block's own: the o / down projections accumulate onto it (no copy of the residual)
        h, x, mine = self.input_layernorm.fork(x, self.qkv.in_width, own_x)
        qkv = self.qkv(h, None, use_lora)
        a = K.attention(qkv, info.cos, info.sin, info.seq_start, info.seq_len, self.H, self.D, self.o.in_width, info.branch)
        if rows is not None:
            a, x, mine = _take_rows(a, rows), _take_rows(x, rows), RES_INPLACE      # (the gathered residual is a buffer of this block's own)
        x = self.o(a, x, use_lora, mine)
        h, x, mine = self.post_attention_layernorm.fork(x, self.gate_up.in_width, True)      # (x is this block's own by now)
        act = K.swiglu(self.gate_up(h, None, use_lora), self.down.in_width)
        return self.down(act, x, use_lora, mine)


class LlamaModel(nn.Module):
    def __init__(self, cfg, dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        self.config = cfg
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size, dtype=dtype, device=device)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList([DecoderLayer(cfg, dtype, device) for _ in range(cfg.num_hidden_layers)])
        self.norm = RMSNormW(cfg.hidden_size, cfg.rms_norm_eps, dtype, device)
        self.gradient_checkpointing = False
        self.grad_ready_hook = None     # callable(layer index): set for ONE forward by DPAEngine; fired from that forward's
        self._rope = {}                 # backward when the gradient of a layer's input exists (halva_amd/dp.py:GradReducer)

    def rope(self, T, device):
        key = (str(device), max(T, 1))
        if key not in self._rope:
            n = max(T, getattr(self.config, "tokenizer_model_max_length", 0) or 0, 16)
            sc = getattr(self.config, "rope_scaling", None) or {}
            if sc and sc.get("type", sc.get("rope_type", "linear")) not in ("linear", "default"):
                raise NotImplementedError("rope_scaling %r (only linear scaling is on the HALVA path)" % (sc,))
            self._rope = {key: K.rope_tables(self.config.hidden_size // self.config.num_attention_heads, n,
                                             getattr(self.config, "rope_theta", 10000.0), device,
                                             float(sc.get("factor", 1.0)) if sc.get("type", sc.get("rope_type")) == "linear" else 1.0)}
            self._rope[(str(device), n)] = self._rope[key]
        return self._rope[key]

    def run_layers(self, x, seq_start, seq_len, use_lora=True, branch=None, rows=None):
        """x [S, T, d] bf16 -> last hidden state after the final RMSNorm (modelling_llama.py:580-705).
        branch = (br_a, br_b, pos): rows are packed [prefix | A | B] sequences (see halva_sdpa_branch_fwd).
        rows: int64 [n] DISTINCT flat indices into [S * T] of the only hidden rows the caller reads (the DPA loss reads the rows in
        front of a label != -100, halva_trainer.py:522-537): the result is then [n, d] in that order, and the top layer does its
        row-wise part (o projection, MLP, norms) for those rows only - the same arithmetic per row, no row of the reference's
        [S, T, d] result that anything reads is dropped."""
        T = x.shape[1]
        cos, sin = self.rope(T, x.device)
        if cos.shape[0] < T:
            self._rope = {}
            cos, sin = self.rope(T, x.device)
        info = SeqInfo(cos, sin, seq_start, seq_len, branch)
        hook = self.grad_ready_hook if torch.is_grad_enabled() else None
        for i, layer in enumerate(self.layers):
            if hook is not None and x.requires_grad:
                # the gradient of layer i's input is the last thing layer i's backward produces: every kernel that adds into
                # the LoRA gradient segments of layers >= i has been enqueued when this fires
                x.register_hook(lambda g, i=i, hook=hook: hook(i))
            if self.gradient_checkpointing and torch.is_grad_enabled() and x.requires_grad:
                x = torch.utils.checkpoint.checkpoint(layer, x, info, use_lora, use_reentrant=False)
            elif rows is not None and TOP_ROWS and i == len(self.layers) - 1:
                x = layer(x, info, use_lora, i > 0, rows)
                rows = None
            else:
                x = layer(x, info, use_lora, i > 0)      # from layer 1 on `x` is the previous layer's own output
        x = self.norm(x)
        return x if rows is None else x.view(-1, x.shape[-1]).index_select(0, rows)


def add_lora(model, r, alpha, generator=None):
    """get_peft_model(model, LoraConfig(r, lora_alpha, target_modules=all Llama linears)) of the reference
    (llava/train/train_halva.py:1085-1101): every base parameter frozen, LoRA factors trainable."""
    for p in model.parameters():
        p.requires_grad_(False)
    for layer in model.model.layers:
        for _, grp in layer.groups():
            grp.attach_lora(r, alpha, grp.weight_cat.dtype, grp.weight_cat.device, generator)
            if DGRAD_TRANSPOSED_COPY:
                grp.build_dgrad_copy()
    return model


def refresh_lora(model):
    """Re-materialise blockdiag(scale * B) in every group's fused weight (call after the B factors were updated)."""
    for layer in model.model.layers:
        if hasattr(layer, "groups"):
            for _, grp in layer.groups():
                if grp.A_cat is not None:
                    grp.refresh_tail()


def lora_named_parameters(model):
    """(name, parameter) of every trainable LoRA tensor, grouped-A first then each B (stable order)."""
    out = []
    for i, layer in enumerate(model.model.layers):
        for sub, grp in layer.groups():
            if grp.A_cat is None:
                continue
            out.append(("model.layers.%d.%s.%s.lora_A_cat" % (i, sub, "+".join(grp.names)), grp.A_cat))
            for n in grp.names:
                out.append(("model.layers.%d.%s.%s.lora_B.default.weight" % (i, sub, n), getattr(grp, n).lora_B["default"].weight))
    return out


# ------------------------------------------------------------------------------------------------
# HF checkpoint <-> fused layout
# ------------------------------------------------------------------------------------------------
def load_hf_llama_weights(model, sd, strict=True):
    """Copy an HF Llama state dict ({model.layers.i.self_attn.q_proj.weight, ...}) into the fused layout."""
    def get(name):
        if name not in sd:
            if strict:
                raise KeyError(name)
            return None
        return sd[name]

    m = model.model
    with torch.no_grad():
        w = get("model.embed_tokens.weight")
        if w is not None:
            m.embed_tokens.weight.copy_(w)
        w = get("model.norm.weight")
        if w is not None:
            m.norm.weight.copy_(w)
        w = get("lm_head.weight")
        if w is not None:
            model.lm_head.weight.copy_(w)
        for i, layer in enumerate(m.layers):
            p = "model.layers.%d." % i
            for sub, grp in layer.groups():
                off = 0
                for n, o in zip(grp.names, grp.outs):
                    w = get(p + sub + "." + n + ".weight")
                    if w is not None:
                        grp.weight[off:off + o].copy_(w)
                    off += o
                if grp.weight_cat_t is not None:          # keep the dgrad copy in step with freshly loaded weights
                    grp.build_dgrad_copy()
            for n in ("input_layernorm", "post_attention_layernorm"):
                w = get(p + n + ".weight")
                if w is not None:
                    getattr(layer, n).weight.copy_(w)


def hf_llama_state_dict(model):
    """Inverse of load_hf_llama_weights (views, no copies)."""
    m = model.model
    out = {"model.embed_tokens.weight": m.embed_tokens.weight.data, "model.norm.weight": m.norm.weight.data,
           "lm_head.weight": model.lm_head.weight.data}
    for i, layer in enumerate(m.layers):
        p = "model.layers.%d." % i
        for sub, grp in layer.groups():
            off = 0
            for n, o in zip(grp.names, grp.outs):
                out[p + sub + "." + n + ".weight"] = grp.weight.data[off:off + o]
                off += o
        out[p + "input_layernorm.weight"] = layer.input_layernorm.weight.data
        out[p + "post_attention_layernorm.weight"] = layer.post_attention_layernorm.weight.data
    return out
