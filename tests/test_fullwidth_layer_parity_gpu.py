"""Parity of the COMPOSED decoder layer at the real widths against the oracle (missing 3 / next 2).

Whole-step parity against the reference's own outputs stops at hidden 256 / 2 layers (tests/test_dpa_step_gpu.py); at d = 4096 / 5120 the
kernels were tested one by one (tests/test_sdpa_bench_shapes_gpu.py) and the step through identities (tests/test_fullsize_*).  Here ONE
`halva_amd.llama.DecoderLayer` - tuned hipBLASLt table on, the K-concatenated LoRA GEMM, the merged-weight dgrad, the RMSNorm fork, RoPE,
sdpa_fwd3 / sdpa_bwd_dkv3 / sdpa_bwd_dq2, SwiGLU, the split-k LoRA weight-gradient kernel's plain twin (autograd .grad path) - runs forward and
backward at the 7B widths (d 4096, 32 x 128 heads, F 11008) and the 13B widths (d 5120, 40 heads, F 13824) with LoRA r = 128, alpha = 256
and B != 0, and is compared with `oracle.nets.decoder_layer` (reference llava/model/language_model/modelling_llama.py:352-420 with the
attention of llava/train/llama_flash_attn_monkey_patch.py:16-93) evaluated in fp32 on the host from the SAME bf16-rounded weights and inputs:

  * plain layout   S = 2 rows of T = 2048, the second one ragged (right padding, 1391 tokens);
  * packed layout  one row [prefix 668 | A 1380 | B 1380] = 3428 tokens with branch points (halva_amd/splice.py:pack_pairs) against the
                   oracle's TWO plain rows [prefix | A] and [prefix | B]: the pair's packing itself is part of what is checked;
  * the top-layer row pruning (`DecoderLayer.forward(rows=)`, LlamaModel.run_layers) on top of the plain layout.

Outputs: y, dx and dA / dB of all seven LoRA targets.  Tolerances (Frobenius norm per tensor): forward 1e-2 relative, gradients 2e-2
relative (what tests/test_sdpa_bench_shapes_gpu.py holds the attention kernels to); the measured values are printed.

Also here: the chunked lm_head -> token log-prob / KL-to-reference path (halva_amd/dpa.py:lm_head_logp / lm_head_kl) at [8192 + 300 rows x 4096]
x 32000 - across a chunk boundary - against oracle.dpa.cal_batch_logp / kl_to_reference (reference llava/train/halva_trainer.py:392-409,580-588).
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dpa as odpa, nets  # noqa: E402

DEV = "cuda"
WIDTHS = {"7b": dict(d=4096, H=32, F=11008), "13b": dict(d=5120, H=40, F=13824)}
R, ALPHA = 128, 256.0
PRE = "model.layers.0."


def bf(t):
    return t.to(torch.bfloat16)


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _build(width, seed):
    """(product layer on the GPU, oracle weight dict, oracle LoRA dict (leaf tensors that require grad), cfg dict)"""
    from halva_amd import gemm_tuning
    from halva_amd.llama import DGRAD_TRANSPOSED_COPY, DecoderLayer, LlamaConfig
    gemm_tuning.enable_tuned_gemms()
    w = WIDTHS[width]
    cfg = LlamaConfig(hidden_size=w["d"], intermediate_size=w["F"], num_attention_heads=w["H"], num_hidden_layers=1)
    layer = DecoderLayer(cfg, torch.bfloat16, DEV)
    g = torch.Generator().manual_seed(seed)
    W, lora = {}, {}
    with torch.no_grad():
        for sub, grp in layer.groups():
            off = 0
            for n, o in zip(grp.names, grp.outs):
                wt = bf(torch.randn(o, grp.in_features, generator=g) * 0.02)
                grp.weight[off:off + o].copy_(wt)
                W[PRE + sub + "." + n + ".weight"] = wt.float()
                off += o
        for n in ("input_layernorm", "post_attention_layernorm"):
            wt = bf(1.0 + 0.1 * torch.randn(w["d"], generator=g))
            getattr(layer, n).weight.copy_(wt)
            W[PRE + n + ".weight"] = wt.float()
        for sub, grp in layer.groups():
            grp.attach_lora(R, ALPHA, torch.bfloat16, DEV)
            for gi, n in enumerate(grp.names):
                a = bf(torch.randn(R, grp.in_features, generator=g) / math.sqrt(grp.in_features))
                b = bf(torch.randn(getattr(grp, n).out_features, R, generator=g) * 0.02)      # B != 0: the LoRA path carries signal
                grp.A_cat[gi * R:(gi + 1) * R].copy_(a)
                getattr(grp, n).lora_B["default"].weight.copy_(b)
                lora[PRE + sub + "." + n + ".A"] = a.float().requires_grad_(True)
                lora[PRE + sub + "." + n + ".B"] = b.float().requires_grad_(True)
            if DGRAD_TRANSPOSED_COPY:
                grp.build_dgrad_copy()
    ocfg = dict(num_attention_heads=w["H"], rms_norm_eps=cfg.rms_norm_eps, rope_theta=10000.0)
    return layer, W, lora, ocfg


def _product(layer, x, dy, lens, branch=None, rows=None):
    """forward + backward of the product layer; returns (y, dx, {oracle-style LoRA name: grad}) on the host in fp32"""
    from halva_amd import kernels as K
    from halva_amd.llama import SeqInfo
    S, T, d = x.shape
    mk = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    cos, sin = K.rope_tables(layer.D, max(T, 16), 10000.0, DEV)
    br = None
    if branch is not None:
        br = (mk(branch[0]), mk(branch[1]), branch[2].to(DEV))
    info = SeqInfo(cos, sin, mk([0] * S), mk(lens), br)
    for p in layer.parameters():
        p.grad = None
    xg = x.to(DEV).clone().requires_grad_(True)
    y = layer(xg, info, True, False, None if rows is None else rows.to(DEV))
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    grads = {}
    for sub, grp in layer.groups():
        for gi, n in enumerate(grp.names):
            grads[PRE + sub + "." + n + ".A"] = grp.A_cat.grad[gi * R:(gi + 1) * R].float().cpu()
            grads[PRE + sub + "." + n + ".B"] = getattr(grp, n).lora_B["default"].weight.grad.float().cpu()
    return y.detach().float().cpu(), xg.grad.float().cpu(), grads


def _threads():
    """one thread per PHYSICAL core: the SMT siblings slow the host GEMMs several-fold (bench.physical_cores)"""
    import bench
    torch.set_num_threads(bench.physical_cores()[0])


def _oracle(W, lora, ocfg, x, dys, keep):
    """oracle.nets.decoder_layer in fp32 on the host (varlen attention = what the reference's GPU path computes): ONE forward, one backward per
    upstream gradient in `dys`; returns y and [(dx, {name: grad})] in that order"""
    xr = x.float().requires_grad_(True)
    y = nets.decoder_layer(xr, W, PRE, keep, ocfg, lora, ALPHA / R, varlen=True)
    outs = []
    for i, dy in enumerate(dys):
        for t in lora.values():
            t.grad = None
        xr.grad = None
        y.backward(dy.float(), retain_graph=i + 1 < len(dys))
        outs.append((xr.grad.clone(), {k: v.grad.clone() for k, v in lora.items()}))
    return y.detach(), outs


def _compare(tag, y, yo, dx, dxo, gr, gro):
    worst = {"fwd": rel_err(y, yo), "dx": rel_err(dx, dxo)}
    for k in sorted(gro):
        worst[k.replace(PRE, "")] = rel_err(gr[k], gro[k])
    print("%s: " % tag + ", ".join("%s %.2e" % kv for kv in worst.items()))
    assert torch.isfinite(y).all() and torch.isfinite(dx).all()
    assert worst["fwd"] < 1e-2, (tag, worst)
    for k, v in worst.items():
        if k != "fwd":
            assert v < 2e-2, (tag, k, worst)


@pytest.mark.parametrize("width", ["7b", "13b"])
def test_decoder_layer_plain_and_ragged_rows_match_the_oracle(width):
    _threads()
    layer, W, lora, ocfg = _build(width, seed=101)
    d = WIDTHS[width]["d"]
    S, T, lens = 2, 2048, [2048, 1391]
    g = torch.Generator().manual_seed(5)
    x = bf(torch.randn(S, T, d, generator=g))
    dy = bf(torch.randn(S, T, d, generator=g))
    keep = torch.zeros(S, T, dtype=torch.bool)
    for s in range(S):
        keep[s, :lens[s]] = True
        x[s, lens[s]:] = 0        # padded rows: embeddings of the pad token in the reference; any finite value - they never reach a valid row
        dy[s, lens[s]:] = 0       # the loss never reads a padded row
    y, dx, gr = _product(layer, x, dy, lens)
    valid = keep.view(-1)
    dys, idx = [dy], None
    if width == "7b":
        # the top layer's row pruning (LlamaModel.run_layers(rows=)): only some valid rows are read by the loss
        idx = torch.nonzero(valid).flatten()
        idx = idx[torch.randperm(idx.numel(), generator=g)[:900]].sort().values
        dyr = bf(torch.randn(idx.numel(), d, generator=g))
        dy_full = torch.zeros(S * T, d, dtype=torch.bfloat16)
        dy_full[idx] = dyr
        dys.append(dy_full.view(S, T, d))
    yo, outs = _oracle(W, lora, ocfg, x, dys, keep)
    dxo, gro = outs[0]
    _compare(width + " plain", y.view(-1, d)[valid], yo.view(-1, d)[valid], dx.view(-1, d)[valid], dxo.view(-1, d)[valid], gr, gro)
    if idx is not None:
        yr, dxr, grr = _product(layer, x, dyr, lens, rows=idx)
        dxo2, gro2 = outs[1]
        _compare(width + " rows=", yr, yo.view(-1, d)[idx], dxr.view(-1, d)[valid], dxo2.view(-1, d)[valid], grr, gro2)


@pytest.mark.parametrize("width", ["7b", "13b"])
def test_decoder_layer_packed_pair_matches_the_oracles_two_rows(width):
    """[prefix 668 | A 1380 | B 1380] with br_a = 668, br_b = 2048 and explicit RoPE positions = the bench's packed row."""
    _threads()
    layer, W, lora, ocfg = _build(width, seed=202)
    d = WIDTHS[width]["d"]
    P, TA = 668, 2048
    T = TA + (TA - P)
    g = torch.Generator().manual_seed(6)
    x = bf(torch.randn(1, T, d, generator=g))
    dy = bf(torch.randn(1, T, d, generator=g))
    pos = torch.cat([torch.arange(TA), torch.arange(P, TA)]).to(torch.int32)
    y, dx, gr = _product(layer, x, dy, [T], branch=([P], [TA], pos))
    # the oracle: two plain causal rows of 2048; the prefix rows of row B carry no upstream gradient of their own (they are row A's)
    xo = torch.stack([x[0, :TA], torch.cat([x[0, :P], x[0, TA:]])])
    dyo = torch.stack([dy[0, :TA], torch.cat([torch.zeros(P, d, dtype=dy.dtype), dy[0, TA:]])])
    keep = torch.ones(2, TA, dtype=torch.bool)
    yo, ((dxo, gro),) = _oracle(W, lora, ocfg, xo, [dyo], keep)
    yo_packed = torch.cat([yo[0], yo[1, P:]])
    dxo_packed = torch.cat([dxo[0, :P] + dxo[1, :P], dxo[0, P:], dxo[1, P:]])
    assert rel_err(yo[1, :P], yo[0, :P]) < 1e-5          # (the oracle's own prefix rows agree: causal)
    _compare(width + " packed", y[0], yo_packed, dx[0], dxo_packed, gr, gro)
    # the branch rows on their own (a wrong branch mask would hide in the whole-tensor norm: 40 % of the rows)
    assert rel_err(y[0, TA:], yo[1, P:]) < 1e-2 and rel_err(dx[0, TA:], dxo[1, P:]) < 2e-2
    assert rel_err(dx[0, :P], dxo_packed[:P]) < 2e-2


def test_lm_head_logp_and_kl_across_a_chunk_boundary_match_the_oracle():
    """halva_amd.dpa.lm_head_logp / lm_head_kl (chunks of 8192 rows) at the 7B head: [8492 x 4096] x 32000."""
    from halva_amd import dpa
    _threads()
    rows, d, V = dpa.LOGIT_CHUNK_ROWS + 300, 4096, 32000
    g = torch.Generator().manual_seed(9)
    h_pol = bf(torch.randn(rows, d, generator=g))
    h_ref = bf(h_pol.float() + 0.3 * torch.randn(rows, d, generator=g))
    Wp = bf(torch.randn(V, d, generator=g) * 0.02)
    tgt = torch.randint(0, V, (rows,), generator=g)
    gl = torch.randn(rows, generator=g)
    # product
    hp = h_pol.to(DEV).requires_grad_(True)
    lp = dpa.lm_head_logp(hp, Wp.to(DEV), tgt.to(torch.int32).to(DEV))
    lp.backward(gl.to(DEV))
    dh_logp = hp.grad.float().cpu()
    hp2 = h_pol.to(DEV).requires_grad_(True)
    kl = dpa.lm_head_kl(hp2, h_ref.to(DEV), Wp.to(DEV), Wp.to(DEV))
    kl.backward()
    torch.cuda.synchronize()
    dh_kl = hp2.grad.float().cpu()
    # oracle: ONE [rows, V] fp32 logits matrix serves both heads ([1, rows + 1, V] with labels shifted by one for cal_batch_logp, which gathers
    # logits[:, :-1] at labels[:, 1:])
    ho = h_pol.float().requires_grad_(True)
    pol_logits = ho @ Wp.float().t()
    labels = torch.cat([torch.tensor([-100]), tgt])[None]
    lo = odpa.cal_batch_logp(torch.cat([pol_logits, torch.zeros(1, V)])[None], labels)[0]
    lo.backward(gl, retain_graph=True)
    d_lp = (lp.detach().float().cpu() - lo.detach()).abs()
    print("lm_head_logp: mean |diff| %.2e max %.2e; dh rel %.2e" % (float(d_lp.mean()), float(d_lp.max()), rel_err(dh_logp, ho.grad)))
    assert float(d_lp.mean()) < 4e-3 and float(d_lp.max()) < 4e-2      # bf16 logits: |logit| 2^-9 per entry
    assert rel_err(dh_logp, ho.grad) < 2e-2
    del lo
    ho.grad = None
    with torch.no_grad():
        ref_logits = (h_ref.float() @ Wp.float().t())[None]
    klo = odpa.kl_to_reference(pol_logits[None], ref_logits, torch.zeros(1, rows, dtype=torch.long))      # every row counts; B = 1
    klo.backward()
    print("lm_head_kl: product %.4f oracle %.4f; dh rel %.2e" % (float(kl.detach()), float(klo.detach()), rel_err(dh_kl, ho.grad)))
    assert abs(float(kl.detach()) - float(klo.detach())) < 1e-2 * abs(float(klo.detach()))
    assert rel_err(dh_kl, ho.grad) < 2e-2
