"""torch.autograd wrappers over the C-ABI kernels (halva_amd/hip.py).  Device tensors only; bf16 activations.

Each Function names the reference call it stands in for; numerics follow the vendored transformers-4.31 spec
(reference llava/model/language_model/modelling_llama.py) - see include/halva_hip.h for the per-kernel contract.
"""
import math
import ctypes
import os

import torch

from . import hip
from .hip import BF16, F32, call, ptr, stream_ptr

_DT = {torch.bfloat16: BF16, torch.float32: F32}


def _chk(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise hip.HalvaHipError("%s must live on the GPU (got %s); the DPA path has no CPU fallback" % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


# ------------------------------------------------------------------------------------------------
class _RMSNorm(torch.autograd.Function):
    """LlamaRMSNorm.forward (modelling_llama.py:65-70).  Weight is frozen on the DPA path -> dx only.
    out_width > d: the rows are written into the left d columns of a [.., out_width] buffer (the operand buffer of the
    next LoRA projection, whose right columns that projection fills) and the incoming gradient has that width too."""

    @staticmethod
    def forward(ctx, x, w, eps, out_width):
        _chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w")
        d = x.shape[-1]
        rows = x.numel() // d
        width = out_width or d
        y = torch.empty(*x.shape[:-1], width, dtype=x.dtype, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        call("halva_rmsnorm_fwd_ld", ptr(x), ptr(w), ptr(y), width, ptr(rstd), rows, d, float(eps), stream_ptr())
        ctx.save_for_backward(x, w, rstd)
        ctx.width = width
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, rstd = ctx.saved_tensors
        dy = _chk(dy.contiguous(), torch.bfloat16, "dy")
        dx = torch.empty_like(x)
        d = x.shape[-1]
        call("halva_rmsnorm_bwd_ld", ptr(dy), ctx.width, ptr(x), ptr(w), ptr(rstd), ptr(dx), x.numel() // d, d, stream_ptr())
        return dx, None, None, None


def rmsnorm(x, w, eps, out_width=None):
    return _RMSNorm.apply(x, w, eps, out_width)


class _RMSNormFork(torch.autograd.Function):
    """The residual fork of a decoder layer as ONE node: returns (rmsnorm(x), x').  The second output is what the residual
    connection must consume (modelling_llama.py:395-417); its gradient then arrives here together with the norm's and the
    backward kernel adds it while writing dx - instead of autograd's separate accumulation pass over the hidden state.
    x' is a COPY of x written by the same kernel (halva_rmsnorm_fwd_fork_ld): the projection that closes the block accumulates
    onto it in place (llama._LoraGroupFn, beta = 1 GEMM) instead of first copying the residual into a fresh output buffer."""

    @staticmethod
    def forward(ctx, x, w, eps, out_width):
        _chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w")
        d = x.shape[-1]
        rows = x.numel() // d
        width = out_width or d
        y = torch.empty(*x.shape[:-1], width, dtype=x.dtype, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        xc = torch.empty_like(x)
        call("halva_rmsnorm_fwd_fork_ld", ptr(x), ptr(w), ptr(y), width, ptr(rstd), ptr(xc), rows, d, float(eps), stream_ptr())
        ctx.save_for_backward(x, w, rstd)
        ctx.width = width
        return y, xc

    @staticmethod
    def backward(ctx, dy, dres):
        x, w, rstd = ctx.saved_tensors
        d = x.shape[-1]
        if dy is None:                                  # only the residual path carried a gradient
            return dres, None, None, None
        dy = _chk(dy.contiguous(), torch.bfloat16, "dy")
        if dres is not None:
            dres = _chk(dres.contiguous(), torch.bfloat16, "dres")
        dx = torch.empty_like(x)
        call("halva_rmsnorm_bwd_res_ld", ptr(dy), ctx.width, ptr(x), ptr(w), ptr(rstd), ptr(dres), ptr(dx), x.numel() // d, d,
             stream_ptr())
        return dx, None, None, None


_wgrad_ws = {}


def wgrad_accumulate(C, A, B, alpha=1.0):
    """C [M, N] f32 += alpha * A^T B for bf16 column windows A [rows, M], B [rows, N] of wider row-major buffers (LoRA weight
    gradients, include/halva_hip.h:halva_wgrad_accumulate).  Split over the rows; partials are summed in a fixed order."""
    rows, M = A.shape
    N = B.shape[1]
    assert B.shape[0] == rows and C.shape == (M, N) and C.dtype == torch.float32 and C.is_contiguous()
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.stride(1) == 1 and B.stride(1) == 1
    ws = _wgrad_ws.get(C.device)
    if ws is None:
        ws = _wgrad_ws[C.device] = torch.empty(48 * 2 ** 20, dtype=torch.float32, device=C.device)      # 192 MB: 30+ slabs at these sizes
    call("halva_wgrad_accumulate", ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), M, N, rows, float(alpha), ptr(ws), ws.numel(),
         stream_ptr())


def wgrad_accumulate_batch(items):
    """[(C, A, B, alpha), ...] -> every C [M, N] f32 += alpha * A^T B, as ONE launch pair for the two to four products of a LoRA group
    (include/halva_hip.h:halva_wgrad_accumulate_batch); bitwise the results of wgrad_accumulate called once per item."""
    dev = items[0][0].device
    ws = _wgrad_ws.get(dev)
    if ws is None:
        ws = _wgrad_ws[dev] = torch.empty(48 * 2 ** 20, dtype=torch.float32, device=dev)      # 192 MB: 30+ slabs at these sizes
    arr = (hip.WgradItem * len(items))()
    for it, (C, A, B, alpha) in zip(arr, items):
        rows, M = A.shape
        N = B.shape[1]
        assert B.shape[0] == rows and C.shape == (M, N) and C.dtype == torch.float32 and C.is_contiguous()
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.stride(1) == 1 and B.stride(1) == 1
        it.A, it.lda, it.B, it.ldb, it.C, it.M, it.N, it.rows, it.alpha = ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), M, N, rows, float(alpha)
    call("halva_wgrad_accumulate_batch", len(items), ctypes.cast(arr, ctypes.c_void_p), ptr(ws), ws.numel(), stream_ptr())


def wgrad_supported(A, B):
    """Shapes / alignment halva_wgrad_accumulate takes (16-byte row chunks); anything else stays on the library GEMM."""
    return (A.shape[1] % 8 == 0 and B.shape[1] % 8 == 0 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0 and A.stride(1) == 1
            and B.stride(1) == 1 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0 and A.shape[1] * B.shape[1] <= 16 * 2 ** 20)


def rmsnorm_fork(x, w, eps, out_width=None):
    """(rmsnorm(x), x) with the two gradients summed inside the backward kernel; use the second output for the residual add."""
    return _RMSNormFork.apply(x, w, eps, out_width)


# ------------------------------------------------------------------------------------------------
def rope_tables(head_dim, max_pos, base=10000.0, device="cuda", linear_factor=1.0):
    """cos/sin tables [max_pos, D/2] in bf16 (modelling_llama.py:79-106: computed in fp32, cast to the compute dtype).
    linear_factor: LlamaLinearScalingRotaryEmbedding (t / factor), what vila/model/language_model/builder.py:43-50
    turns on when model_max_length exceeds the checkpoint's max_position_embeddings."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32, device=device) / head_dim))
    freqs = torch.outer(torch.arange(max_pos, dtype=torch.float32, device=device) / float(linear_factor), inv)
    return freqs.cos().to(torch.bfloat16).contiguous(), freqs.sin().to(torch.bfloat16).contiguous()


def _rope_inplace(qkv, cos, sin, T, H, D, inverse, branch=None):
    """branch: optional (br_a, br_b[, ...]) int32 [S] device tensors of branch-packed rows [prefix | A | pad | B]: row t sits at position t, rows of
    B (t >= br_b) at br_a + (t - br_b) - halva_rope_qk_branch, the SAME rule the attention backward's rotating epilogues apply
    (halva_sdpa_branch_bwd_rope).  Round 6: the forward used to rotate with a caller-supplied position table while the backward
    derived the positions from the branch points - a table that disagreed with them gave silently wrong dq / dk.  One source of truth now: the
    branch points, in both directions; the table splice.pack_pairs still returns is what tests/test_pack_pairs_cpu.py holds to that rule."""
    rows = qkv.numel() // (3 * H * D)
    if branch is None:
        call("halva_rope_qk", ptr(qkv), ptr(cos), ptr(sin), ptr(None), rows, T, H, D, cos.shape[0], int(inverse), stream_ptr())
    else:
        call("halva_rope_qk_branch", ptr(qkv), ptr(cos), ptr(sin), ptr(branch[0]), ptr(branch[1]), rows, T, H, D, cos.shape[0], int(inverse), stream_ptr())


class _RopeQK(torch.autograd.Function):
    """apply_rotary_pos_emb (modelling_llama.py:154-169) in place on the q,k thirds of a packed qkv buffer.
    Its transpose is applied by _SdpaCausal.backward on the freshly written dqkv (same launch sequence, no extra
    buffer), so this node's backward is the identity."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, H, D, branch=None):
        _chk(qkv, torch.bfloat16, "qkv")
        _rope_inplace(qkv, cos, sin, qkv.shape[1], H, D, False, branch)
        ctx.mark_dirty(qkv)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        return dqkv, None, None, None, None, None


sdpa_bwd_probe = None      # a list: every causal-SDPA backward launch appends (start event, end event, S, T, H, D, branched)
sdpa_fwd_probe = None      # the same for the forward launches (bench.py)
# dS workspace of the attention backward (include/halva_hip.h:halva_sdpa_branch_bwd_ws): one buffer per device, grown on demand and
# reused by every layer (it carries nothing between calls).  HALVA_SDPA_DS_WS=0 runs the split backward without it.
SDPA_DS_WS = os.environ.get("HALVA_SDPA_DS_WS", "1") != "0"
_sdpa_ws = {}


def _sdpa_workspace(dev, S, T, H, D):
    if not SDPA_DS_WS:
        return None, 0
    need = int(hip.load().halva_sdpa_bwd_ws_bytes(S, T, H, D))
    if need == 0:
        return None, 0
    ws = _sdpa_ws.get(dev)
    if ws is None or ws.numel() < need:
        _sdpa_ws[dev] = ws = torch.empty(need, dtype=torch.uint8, device=dev)
    return ws, need


class _SdpaCausal(torch.autograd.Function):
    """flash_attn_varlen_qkvpacked_func(causal=True) + unpad/pad_input (llama_flash_attn_monkey_patch.py:71-91).
    qkv: [S, T, 3*H*D] bf16 (already rotated).  Returns [S, T, H*D].  If cos/sin are given the backward also applies
    the inverse rotation to dq, dk (see _RopeQK)."""

    @staticmethod
    def forward(ctx, qkv, seq_start, seq_len, H, D, cos, sin, out_width=None, branch=None):
        """branch: optional (br_a, br_b[, pos]) int32 device tensors - packed [prefix | A | B] rows whose B part must not see A
        (halva_sdpa_branch_fwd); their RoPE positions follow from the branch points (_rope_inplace); a third entry is ignored."""
        _chk(qkv, torch.bfloat16, "qkv")
        S, T = qkv.shape[0], qkv.shape[1]
        width = out_width or H * D
        out = torch.empty(S, T, width, dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty(S, H, T, dtype=torch.float32, device=qkv.device)
        br_a, br_b = (branch[0], branch[1]) if branch is not None else (None, None)
        probe = sdpa_fwd_probe
        if probe is not None:          # bench.py: HIP events around the launch, on the stream it goes to
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call("halva_sdpa_branch_fwd", ptr(qkv), ptr(out), width, ptr(lse), ptr(seq_start), ptr(seq_len), ptr(br_a), ptr(br_b), S, T, H,
             D, 0.0, stream_ptr())
        if probe is not None:
            e1.record()
            probe.append((e0, e1, S, T, H, D, branch is not None))
        ctx.branch = branch
        ctx.save_for_backward(qkv, lse, seq_start, seq_len)
        # `out` may be the (wider) operand buffer of the next LoRA projection, which fills its right columns in place; the
        # backward only reads the left H*D columns, so keep a detached alias instead of a version-checked saved tensor
        ctx.out_alias = out.detach()
        ctx.rope = (cos, sin)
        ctx.dims = (S, T, H, D, width)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, lse, seq_start, seq_len = ctx.saved_tensors
        out = ctx.out_alias
        S, T, H, D, width = ctx.dims
        dout = _chk(dout.contiguous(), torch.bfloat16, "dout")
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(S, H, T, dtype=torch.float32, device=qkv.device)
        br_a, br_b = (ctx.branch[0], ctx.branch[1]) if ctx.branch is not None else (None, None)
        probe = sdpa_bwd_probe
        if probe is not None:          # bench.py: HIP events around the launch, on the stream it goes to
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ws, ws_bytes = _sdpa_workspace(qkv.device, S, T, H, D)
        cos, sin = ctx.rope
        # round 5: the inverse rotation of dq / dk rides in the backward kernels' store epilogues (halva_sdpa_branch_bwd_rope; positions follow
        # from the branch points exactly as splice.pack_pairs lays them out); HALVA_ROPE_FUSED_BWD=0 (read by the library): as its own launch
        call("halva_sdpa_branch_bwd_rope", ptr(qkv), ptr(out), width, ptr(dout), dout.shape[-1], ptr(lse), ptr(dqkv), ptr(delta),
             ptr(ws), ws_bytes, ptr(seq_start), ptr(seq_len), ptr(br_a), ptr(br_b), ptr(cos), ptr(sin), 0 if cos is None else cos.shape[0],
             S, T, H, D, 0.0, stream_ptr())
        if probe is not None:
            e1.record()
            probe.append((e0, e1, S, T, H, D, ctx.branch is not None))
        return dqkv, None, None, None, None, None, None, None, None


def attention(qkv, cos, sin, seq_start, seq_len, H, D, out_width=None, branch=None):
    """RoPE (in place) + causal attention on a packed [S, T, 3*H*D] projection output."""
    qkv = _RopeQK.apply(qkv, cos, sin, H, D, branch)      # (positions from the branch points, as in the backward; branch[2], a position table, is not read)
    return _SdpaCausal.apply(qkv, seq_start, seq_len, H, D, cos, sin, out_width, branch)


def sdpa_causal(qkv, seq_start, seq_len, H, D, br_a=None, br_b=None):
    return _SdpaCausal.apply(qkv, seq_start, seq_len, H, D, None, None, None, None if br_a is None else (br_a, br_b, None))


def sdpa_full(qkv, H, D, scale=0.0):
    """Non-causal attention of the frozen CLIP / SigLIP tower (forward only).  qkv [N, S, 3*H*D] -> [N, S, H*D].
    scale 0 = 1/sqrt(D); SigLIP's 72-wide heads run zero-padded to D=128 with scale = 72**-0.5."""
    _chk(qkv, torch.bfloat16, "qkv")
    N, S = qkv.shape[0], qkv.shape[1]
    out = torch.empty(N, S, H * D, dtype=torch.bfloat16, device=qkv.device)
    call("halva_sdpa_full_fwd", ptr(qkv), ptr(out), N, S, H, D, float(scale), stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------
class _SwiGLU(torch.autograd.Function):
    """act_fn(gate) * up (modelling_llama.py:197) on a fused [rows, 2F] gate|up buffer; optional wider output buffer
    (see _RMSNorm)."""

    @staticmethod
    def forward(ctx, gu, out_width):
        _chk(gu, torch.bfloat16, "gu")
        F2 = gu.shape[-1]
        rows = gu.numel() // F2
        width = out_width or F2 // 2
        out = torch.empty(*gu.shape[:-1], width, dtype=torch.bfloat16, device=gu.device)
        call("halva_swiglu_fwd_ld", ptr(gu), ptr(out), width, rows, F2 // 2, stream_ptr())
        ctx.save_for_backward(gu)
        ctx.width = width
        return out

    @staticmethod
    def backward(ctx, dout):
        (gu,) = ctx.saved_tensors
        dout = _chk(dout.contiguous(), torch.bfloat16, "dout")
        dgu = torch.empty_like(gu)
        F2 = gu.shape[-1]
        call("halva_swiglu_bwd_ld", ptr(dout), ctx.width, ptr(gu), ptr(dgu), gu.numel() // F2, F2 // 2, stream_ptr())
        return dgu, None


def swiglu(gu, out_width=None):
    return _SwiGLU.apply(gu, out_width)


# ------------------------------------------------------------------------------------------------
def _splice_bwd_plan(src_host, device):
    """Host-side plan of the splice backward: a feature row feeds one output row per sequence that shows its image (two when the
    rows of a pair run unpacked).  Occurrence k of every feature row, for k = 0, 1, ..: (feature rows, output rows) index pairs with
    UNIQUE targets, so the sum runs in a fixed order (no float atomics - two runs give the same bits) and needs no device->host
    read-back (the index plan `src` is host data, halva_amd/splice.py)."""
    import numpy as np
    src = src_host.numpy().reshape(-1)
    rows = np.nonzero(src <= -2)[0]
    if rows.size == 0:
        return []
    f = -src[rows].astype(np.int64) - 2
    order = np.argsort(f, kind="stable")
    fs, rs = f[order], rows[order]
    start = np.ones(fs.size, dtype=bool)
    start[1:] = fs[1:] != fs[:-1]
    idx = np.arange(fs.size)
    occ = idx - np.maximum.accumulate(np.where(start, idx, 0))
    out = []
    for k in range(int(occ.max()) + 1):
        sel = occ == k
        out.append((torch.from_numpy(fs[sel]).to(device, non_blocking=True), torch.from_numpy(rs[sel]).to(device, non_blocking=True)))
    return out


class _SpliceRows(torch.autograd.Function):
    """The text/image splice of prepare_inputs_labels_for_multimodal[_signed] (llava_arch.py:285-374) as one row
    gather driven by a host-computed index plan.  Gradient flows to the image features (mm_projector) only."""

    @staticmethod
    def forward(ctx, embed, feats, src, S, T, bwd_plan):
        _chk(embed, torch.bfloat16, "embed_tokens"), _chk(feats, torch.bfloat16, "image_features")
        _chk(src, torch.int32, "src")
        d = embed.shape[1]
        out = torch.empty(S, T, d, dtype=torch.bfloat16, device=embed.device)
        call("halva_splice_rows", ptr(embed), ptr(feats), ptr(src), ptr(out), S * T, d, stream_ptr())
        ctx.bwd_plan = bwd_plan
        if bwd_plan is None:
            ctx.save_for_backward(src)
        ctx.feat_shape = feats.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        d = dout.shape[-1]
        plan = ctx.bwd_plan
        if plan is None:      # `src` was handed over as a device tensor: derive the plan from a host copy (one read-back)
            (src,) = ctx.saved_tensors
            plan = _splice_bwd_plan(src.cpu(), dout.device)
        dfeat = torch.zeros(ctx.feat_shape, dtype=torch.float32, device=dout.device).view(-1, d)
        do = dout.reshape(-1, d)
        for fk, rk in plan:
            dfeat[fk] += do[rk].float()
        return None, dfeat.to(torch.bfloat16).view(ctx.feat_shape), None, None, None, None


def splice_rows(embed, feats, src, S, T):
    """`src`: the int32 index plan (halva_amd/splice.py).  Hand it over as the HOST tensor the planner produced: it is uploaded
    here, and the backward's occurrence plan is built on the host beside it (no device->host synchronisation in the backward)."""
    bwd_plan = None
    if not src.is_cuda and embed.is_cuda:
        if feats.requires_grad and torch.is_grad_enabled():
            bwd_plan = _splice_bwd_plan(src, embed.device)
        src = src.to(embed.device, non_blocking=True)
    return _SpliceRows.apply(embed, feats.contiguous(), src, S, T, bwd_plan)


# ------------------------------------------------------------------------------------------------
def gemm(A, B, bias=None, trans_a=False, trans_b=False, epilogue=0, out=None, out_dtype=torch.bfloat16, pre_act=None,
         accumulate=False):
    """C = epi(opA(A) @ opB(B)^T + bias) on the MFMA GEMM kernel.  Shapes: A [M,K] ([K,M] if trans_a), B [N,K] ([K,N])."""
    _chk(A, torch.bfloat16, "A"), _chk(B, torch.bfloat16, "B")
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N = B.shape[1] if trans_b else B.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=A.device)
    call("halva_gemm_bf16", ptr(A), ptr(B), ptr(bias), ptr(out), ptr(pre_act), M, N, K, int(trans_a), int(trans_b),
         int(epilogue), _DT[out.dtype], int(accumulate), stream_ptr())
    return out


class _ProjectorMLP(torch.autograd.Function):
    """mlp2x_gelu projector (multimodal_projector/builder.py:39-46): Linear -> GELU -> Linear with the bias/GELU
    epilogue fused into the first GEMM.  The input (CLIP features) carries no gradient (clip_encoder.py:37)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x2 = _chk(x.reshape(-1, x.shape[-1]).contiguous(), torch.bfloat16, "x")
        h_pre = torch.empty(x2.shape[0], w1.shape[0], dtype=torch.bfloat16, device=x.device)
        g = gemm(x2, w1, b1, epilogue=1, pre_act=h_pre)
        y = gemm(g, w2, b2)
        ctx.save_for_backward(x2, h_pre, g)
        ctx.params = (w1, b1, w2, b2)
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, h_pre, g = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        dy2 = _chk(dy.reshape(-1, dy.shape[-1]).contiguous(), torch.bfloat16, "dy")
        M = dy2.shape[0]
        st = stream_ptr()
        dw2 = gemm(dy2, g, trans_a=True, trans_b=True, out_dtype=torch.float32)          # dW2[n,k] = sum_m dy[m,n] g[m,k]
        db2 = torch.zeros(w2.shape[0], dtype=torch.float32, device=dy.device)
        call("halva_colsum", ptr(dy2), ptr(db2), M, w2.shape[0], st)
        dg = gemm(dy2, w2, trans_b=True)                                                  # dg[m,k] = sum_n dy[m,n] W2[n,k]
        dh = torch.empty_like(dg)
        call("halva_gelu_bwd", ptr(dg), ptr(h_pre), ptr(dh), M, dg.shape[1], st)
        dw1 = gemm(dh, x2, trans_a=True, trans_b=True, out_dtype=torch.float32)
        db1 = torch.zeros(w1.shape[0], dtype=torch.float32, device=dy.device)
        call("halva_colsum", ptr(dh), ptr(db1), M, w1.shape[0], st)
        grads = []
        for prm, gr in ((w1, dw1), (b1, db1), (w2, dw2), (b2, db2)):
            if getattr(prm, "main_grad", None) is not None and getattr(prm, "grad_sink", False):
                prm.main_grad.add_(gr)             # fp32 accumulation straight into the flat DP/optimizer buffer
                grads.append(None)
            else:
                grads.append(gr.to(prm.dtype))
        return (None, *grads)


def projector_mlp(x, w1, b1, w2, b2):
    return _ProjectorMLP.apply(x, w1, b1, w2, b2)


class _ProjectorChain(torch.autograd.Function):
    """`linear` and `mlp<N>x_gelu` projectors of any depth (multimodal_projector/builder.py:33-46): Linear (-> GELU -> Linear)*, every
    Linear one MFMA GEMM with the bias (and, between layers, the GELU) fused into its epilogue.  The input carries no gradient."""

    @staticmethod
    def forward(ctx, x, *wb):
        ws, bs = wb[0::2], wb[1::2]
        h = _chk(x.reshape(-1, x.shape[-1]).contiguous(), torch.bfloat16, "x")
        saved = []
        for i, (w, b) in enumerate(zip(ws, bs)):
            if i + 1 < len(ws):
                pre = torch.empty(h.shape[0], w.shape[0], dtype=torch.bfloat16, device=x.device)
                nxt = gemm(h, w, b, epilogue=1, pre_act=pre)
                saved += [h, pre]
            else:
                nxt = gemm(h, w, b)
                saved.append(h)
            h = nxt
        ctx.save_for_backward(*saved)
        ctx.params = (ws, bs)
        return h.view(*x.shape[:-1], ws[-1].shape[0])

    @staticmethod
    def backward(ctx, dy):
        ws, bs = ctx.params
        saved = list(ctx.saved_tensors)
        d = _chk(dy.reshape(-1, dy.shape[-1]).contiguous(), torch.bfloat16, "dy")
        M, st = d.shape[0], stream_ptr()
        pairs = []
        for i in range(len(ws) - 1, -1, -1):
            inp = saved[2 * i]
            dw = gemm(d, inp, trans_a=True, trans_b=True, out_dtype=torch.float32)
            db = torch.zeros(ws[i].shape[0], dtype=torch.float32, device=dy.device)
            call("halva_colsum", ptr(d), ptr(db), M, ws[i].shape[0], st)
            pairs += [(bs[i], db), (ws[i], dw)]
            if i > 0:
                dg = gemm(d, ws[i], trans_b=True)
                d = torch.empty_like(dg)
                call("halva_gelu_bwd", ptr(dg), ptr(saved[2 * i - 1]), ptr(d), M, dg.shape[1], st)
        grads = _sink_or_return(pairs[::-1])
        return (None, *grads)


def projector_chain(x, linears):
    """linears: the nn.Linear modules of the projector in order (GELU between consecutive ones)."""
    args = []
    for lin in linears:
        args += [lin.weight, lin.bias]
    return _ProjectorChain.apply(x, *args)


def _sink_or_return(pairs):
    """Gradient hand-off shared by the projector Functions: fp32 accumulation straight into `.main_grad` (the flat
    DP / optimizer buffer) when the parameter is bound to one, else a plain autograd gradient."""
    grads = []
    for prm, gr in pairs:
        if getattr(prm, "main_grad", None) is not None and getattr(prm, "grad_sink", False):
            prm.main_grad.add_(gr)
            grads.append(None)
        else:
            grads.append(gr.to(prm.dtype))
    return grads


def layernorm(x, w, b, eps, want_stats=False):
    """nn.LayerNorm forward over the last dim (no autograd: frozen towers).  Returns y or (y, stats[rows, 2])."""
    _chk(x, torch.bfloat16, "x"), _chk(w, torch.bfloat16, "w"), _chk(b, torch.bfloat16, "b")
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if want_stats else None
    call("halva_layernorm_fwd", ptr(x), ptr(w), ptr(b), ptr(y), ptr(stats), rows, d, float(eps), stream_ptr())
    return (y, stats) if want_stats else y


def downsample2x2(x):
    """DownSampleBlock of VILA's mlp_downsample (vila base_projector.py:33-54): [n, g*g, c] -> [n, ceil(g/2)^2, 4c]."""
    _chk(x, torch.bfloat16, "x")
    n, s, c = x.shape
    g = int(s ** 0.5)
    if g * g != s:
        raise ValueError("downsample2x2 needs a square token grid, got %d tokens" % s)
    G = (g + 1) // 2
    out = torch.empty(n, G * G, 4 * c, dtype=x.dtype, device=x.device)
    call("halva_downsample2x2", ptr(x), ptr(out), n, g, c, stream_ptr())
    return out


class _DownsampleMLP(torch.autograd.Function):
    """mlp_downsample projector (vila/model/multimodal_projector/base_projector.py:76-83): DownSampleBlock ->
    LayerNorm(4c) -> Linear -> GELU -> Linear.  The input (SigLIP features) carries no gradient (tower frozen,
    src_vila/halva_vila_13b.sh:44), so the backward stops at the LayerNorm parameters."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, w1, b1, w2, b2):
        xd = downsample2x2(x.contiguous())
        x2 = xd.view(-1, xd.shape[-1])
        xn, stats = layernorm(x2, ln_w, ln_b, eps, want_stats=True)
        h_pre = torch.empty(x2.shape[0], w1.shape[0], dtype=torch.bfloat16, device=x.device)
        g = gemm(xn, w1, b1, epilogue=1, pre_act=h_pre)
        y = gemm(g, w2, b2)
        ctx.save_for_backward(x2, stats, xn, h_pre, g)
        ctx.params = (ln_w, ln_b, w1, b1, w2, b2)
        return y.view(xd.shape[0], xd.shape[1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, stats, xn, h_pre, g = ctx.saved_tensors
        ln_w, ln_b, w1, b1, w2, b2 = ctx.params
        dy2 = _chk(dy.reshape(-1, dy.shape[-1]).contiguous(), torch.bfloat16, "dy")
        M = dy2.shape[0]
        st = stream_ptr()
        dev = dy.device
        dw2 = gemm(dy2, g, trans_a=True, trans_b=True, out_dtype=torch.float32)
        db2 = torch.zeros(w2.shape[0], dtype=torch.float32, device=dev)
        call("halva_colsum", ptr(dy2), ptr(db2), M, w2.shape[0], st)
        dg = gemm(dy2, w2, trans_b=True)
        dh = torch.empty_like(dg)
        call("halva_gelu_bwd", ptr(dg), ptr(h_pre), ptr(dh), M, dg.shape[1], st)
        dw1 = gemm(dh, xn, trans_a=True, trans_b=True, out_dtype=torch.float32)
        db1 = torch.zeros(w1.shape[0], dtype=torch.float32, device=dev)
        call("halva_colsum", ptr(dh), ptr(db1), M, w1.shape[0], st)
        dxn = gemm(dh, w1, trans_b=True)                                                  # [M, 4c]
        d = x2.shape[1]
        dlw = torch.zeros(d, dtype=torch.float32, device=dev)
        dlb = torch.zeros(d, dtype=torch.float32, device=dev)
        call("halva_layernorm_bwd_params", ptr(dxn), ptr(x2), ptr(stats), ptr(dlw), ptr(dlb), M, d, st)
        gr = _sink_or_return(((ln_w, dlw), (ln_b, dlb), (w1, dw1), (b1, db1), (w2, dw2), (b2, db2)))
        return (None, gr[0], gr[1], None, *gr[2:])


def downsample_mlp(x, ln_w, ln_b, eps, w1, b1, w2, b2):
    return _DownsampleMLP.apply(x, ln_w, ln_b, eps, w1, b1, w2, b2)


def vit_patch_embed(images, weight_kp, bias, patch, d):
    """Conv2d(3, d, k=p, s=p, padding='valid', bias) of SiglipVisionEmbeddings as im2col + MFMA GEMM."""
    _chk(images, torch.bfloat16, "images"), _chk(weight_kp, torch.bfloat16, "weight")
    n, hw = images.shape[0], images.shape[-1]
    np_ = (hw // patch) ** 2
    Kp = weight_kp.shape[1]
    ws = torch.empty(n * np_, Kp, dtype=torch.bfloat16, device=images.device)
    out = torch.empty(n, np_, d, dtype=torch.bfloat16, device=images.device)
    call("halva_vit_patch_embed", ptr(images), ptr(weight_kp), ptr(bias), ptr(ws), ptr(out), n, hw, patch, d, Kp, stream_ptr())
    return out


def clip_patch_embed(images, weight_kp, patch, d):
    """Conv2d(3, d, k=p, s=p, bias=False) of HF CLIPVisionEmbeddings as im2col + MFMA GEMM.  images [n,3,hw,hw] bf16."""
    _chk(images, torch.bfloat16, "images"), _chk(weight_kp, torch.bfloat16, "weight")
    n, hw = images.shape[0], images.shape[-1]
    np_ = (hw // patch) ** 2
    Kp = weight_kp.shape[1]
    ws = torch.empty(n * np_, Kp, dtype=torch.bfloat16, device=images.device)
    out = torch.empty(n, np_, d, dtype=torch.bfloat16, device=images.device)
    call("halva_clip_patch_embed", ptr(images), ptr(weight_kp), ptr(ws), ptr(out), n, hw, patch, d, Kp, stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------
class _TokenLogp(torch.autograd.Function):
    """logits.log_softmax(-1).gather(target) (halva_trainer.py:406-407) over rows of a [R, V] matrix."""

    @staticmethod
    def forward(ctx, logits, target):
        _chk(logits, None, "logits"), _chk(target, torch.int32, "target")
        R, V = logits.shape
        logp = torch.empty(R, dtype=torch.float32, device=logits.device)
        lse = torch.empty(R, dtype=torch.float32, device=logits.device)
        call("halva_token_logp_fwd", ptr(logits), _DT[logits.dtype], V, ptr(target), ptr(logp), ptr(lse), R, V, stream_ptr())
        ctx.save_for_backward(logits, target, lse)
        return logp

    @staticmethod
    def backward(ctx, g):
        logits, target, lse = ctx.saved_tensors
        g = _chk(g.contiguous().float(), torch.float32, "g")
        d = torch.empty_like(logits)
        R, V = logits.shape
        call("halva_token_logp_bwd", ptr(logits), _DT[logits.dtype], V, ptr(target), ptr(lse), ptr(g), ptr(d), R, V,
             stream_ptr())
        return d, None


def token_logp(logits2d, target_i32):
    return _TokenLogp.apply(logits2d, target_i32)


class _KLRows(torch.autograd.Function):
    """Per-row KL(ref || policy) (halva_trainer.py:583-586) with the policy gradient produced in the same launch."""

    @staticmethod
    def forward(ctx, pol, ref, w):
        _chk(pol, None, "policy logits"), _chk(ref, pol.dtype, "reference logits")
        R, V = pol.shape
        kl = torch.empty(R, dtype=torch.float32, device=pol.device)
        need = pol.requires_grad
        dpol = torch.empty_like(pol) if need else None
        call("halva_kl_rows", ptr(pol), ptr(ref), _DT[pol.dtype], V, ptr(w), ptr(kl), ptr(dpol), 1.0, R, V, stream_ptr())
        ctx.save_for_backward(dpol)
        return kl

    @staticmethod
    def backward(ctx, g):
        (dpol,) = ctx.saved_tensors
        return torch.mul(dpol, g.float().unsqueeze(1)).to(dpol.dtype), None, None       # scale applied in fp32, rounded once


def kl_rows(pol2d, ref2d, w=None):
    return _KLRows.apply(pol2d, ref2d, w)


class _PhraseSum(torch.autograd.Function):
    """accumulate_logps (halva_trainer.py:411-419) fused with the loss-mask multiply (:556-557)."""

    @staticmethod
    def forward(ctx, logp, labels, signs, slot_ids):
        _chk(logp, torch.float32, "logp"), _chk(labels, torch.int64, "labels"), _chk(signs, torch.int64, "signs")
        B, T1 = logp.shape
        P = slot_ids.numel()
        acc = torch.zeros(B, P, dtype=torch.float32, device=logp.device)
        if P:
            call("halva_phrase_sum_fwd", ptr(logp), ptr(labels), ptr(signs), ptr(slot_ids), P, ptr(acc), B, T1, stream_ptr())
        ctx.save_for_backward(labels, signs, slot_ids)
        return acc

    @staticmethod
    def backward(ctx, dacc):
        labels, signs, slot_ids = ctx.saved_tensors
        B, T1 = labels.shape
        P = slot_ids.numel()
        dlogp = torch.zeros(B, T1, dtype=torch.float32, device=labels.device)
        if P:
            dacc = dacc.contiguous().float()
            call("halva_phrase_sum_bwd", ptr(dacc), ptr(labels), ptr(signs), ptr(slot_ids), P, ptr(dlogp), B, T1, stream_ptr())
        return dlogp, None, None, None


def phrase_sum(logp, labels, signs, slot_ids):
    return _PhraseSum.apply(logp, labels, signs, slot_ids)


def quick_gelu_(h):
    """h <- h * sigmoid(1.702 h) in place (bf16, contiguous): CLIP's activation in one pass."""
    _chk(h, torch.bfloat16, "h")
    assert h.numel() % 8 == 0
    call("halva_quick_gelu", ptr(h), ptr(h), h.numel(), stream_ptr())
    return h


def transpose_into(dst, src):
    """dst[c, r] = src[r, c] for 2-D bf16 tensors whose last dimension is contiguous (row strides free): the tiled transpose kernel
    instead of `dst.copy_(src.t())`, which runs the framework's element-wise strided copy (5x slower at weight-matrix sizes)."""
    _chk_dtype = torch.bfloat16
    assert src.dim() == 2 and dst.dim() == 2 and dst.shape == (src.shape[1], src.shape[0]), (src.shape, dst.shape)
    assert src.dtype == _chk_dtype and dst.dtype == _chk_dtype and src.stride(1) == 1 and dst.stride(1) == 1
    call("halva_transpose_bf16", ptr(src), src.stride(0), ptr(dst), dst.stride(0), src.shape[0], src.shape[1], stream_ptr())
    return dst


def probe_layouts(device="cuda"):
    out = torch.zeros(256 + 1024, dtype=torch.int32, device=device)
    call("halva_probe_layouts", ptr(out), out.numel(), stream_ptr())
    return out.cpu()
