"""Whole-step parity at FULL WIDTH and DEPTH: one synthetic case (random N(0, 0.02) base, LoRA B != 0) built ONCE on the host, run through the
product's engine on the GPU and through the oracle (oracle/dpa.py: the reference's `compute_loss`, llava/train/halva_trainer.py:534-592, over the
decoder stack of llava/model/language_model/modelling_llama.py:657-672) in fp32 on the host from the same bf16-rounded weights - and through the
oracle's own bf16 realisations (oracle/realise.py), so that the product is judged against a measured floor.  Shared by
tools/fulldepth_parity.py (32 layers, the one-off of VERDICT r05 item 2) and tests/test_fulldepth_parity_gpu.py (8 layers).

Test infrastructure: imports `oracle`; nothing under halva_amd/ imports this file."""
import math
import time

import numpy as np
import torch

LLAMA_7B_WIDTH = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_attention_heads=32, num_key_value_heads=32,
                      rms_norm_eps=1e-5, max_position_embeddings=4096, pad_token_id=0)
CLIP_L = dict(hidden_size=1024, intermediate_size=4096, num_attention_heads=16, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5,
              num_channels=3)
TARGETS = (("self_attn", ("q_proj", "k_proj", "v_proj", "o_proj")), ("mlp", ("gate_proj", "up_proj", "down_proj")))


def bf16_exact(t):
    return t.to(torch.bfloat16).float()


def make_case(layers, resp_len, image=336, clip_layers=24, r=128, alpha=256.0, seed=1234, n_phrases=3, loss_alpha=0.4, width=None):
    """(llama cfg, clip cfg, base weights, clip weights, LoRA factors, batch, max_len): every float weight an fp32 tensor holding a bf16 value.
    One pair in the bench's layout: [BOS, 34 prompt, <image>, 12 question, 5 'ASSISTANT:', R response, EOS]."""
    cfg = dict(width or LLAMA_7B_WIDTH, num_hidden_layers=layers)
    ccfg = dict(CLIP_L, num_hidden_layers=clip_layers, image_size=image)
    d, F, V, dv = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], ccfg["hidden_size"]
    g = torch.Generator().manual_seed(seed)
    rn = lambda *shape, std=0.02: bf16_exact(torch.randn(*shape, generator=g) * std)
    W = {"model.embed_tokens.weight": rn(V, d), "lm_head.weight": rn(V, d), "model.norm.weight": bf16_exact(1.0 + 0.1 * torch.randn(d, generator=g))}
    shapes = {"q_proj": (d, d), "k_proj": (d, d), "v_proj": (d, d), "o_proj": (d, d), "gate_proj": (F, d), "up_proj": (F, d), "down_proj": (d, F)}
    fac = {}
    for i in range(layers):
        pre = "model.layers.%d." % i
        for sub, names in TARGETS:
            for n in names:
                o, k = shapes[n]
                W[pre + sub + "." + n + ".weight"] = rn(o, k)
                fac[pre + sub + "." + n + ".A"] = bf16_exact(torch.randn(r, k, generator=g) / math.sqrt(k))
                fac[pre + sub + "." + n + ".B"] = rn(o, r, std=0.01)      # B != 0: the LoRA path carries signal, KL != 0
        for n in ("input_layernorm", "post_attention_layernorm"):
            W[pre + n + ".weight"] = bf16_exact(1.0 + 0.1 * torch.randn(d, generator=g))
    W["model.mm_projector.0.weight"], W["model.mm_projector.0.bias"] = rn(d, dv), rn(d)
    W["model.mm_projector.2.weight"], W["model.mm_projector.2.bias"] = rn(d, d), rn(d)
    P = ccfg["patch_size"]
    npatch = (image // P) ** 2
    C = {"embeddings.class_embedding": rn(dv), "embeddings.patch_embedding.weight": rn(dv, 3, P, P), "embeddings.position_embedding.weight": rn(npatch + 1, dv),
         "pre_layrnorm.weight": bf16_exact(1.0 + 0.1 * torch.randn(dv, generator=g)), "pre_layrnorm.bias": rn(dv),
         "post_layernorm.weight": bf16_exact(1.0 + 0.1 * torch.randn(dv, generator=g)), "post_layernorm.bias": rn(dv)}
    Fc = ccfg["intermediate_size"]
    for i in range(clip_layers):
        p = "encoder.layers.%d." % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            C[p + "self_attn." + n + ".weight"], C[p + "self_attn." + n + ".bias"] = rn(dv, dv), rn(dv)
        for n in ("layer_norm1", "layer_norm2"):
            C[p + n + ".weight"], C[p + n + ".bias"] = bf16_exact(1.0 + 0.1 * torch.randn(dv, generator=g)), rn(dv)
        C[p + "mlp.fc1.weight"], C[p + "mlp.fc1.bias"] = rn(Fc, dv), rn(Fc)
        C[p + "mlp.fc2.weight"], C[p + "mlp.fc2.bias"] = rn(dv, Fc), rn(dv)
    # the batch
    pre, post = 1 + 34, 12 + 5
    L = pre + 1 + post + resp_len + 1
    off = pre + 1 + post

    def ids():
        x = torch.randint(3, V, (1, L), generator=g)
        x[:, 0], x[:, pre], x[:, -1] = 1, -200, 2
        return x
    pos = ids()
    neg = pos.clone()
    signs = torch.zeros(1, L, dtype=torch.long)
    step = resp_len // (n_phrases + 1)
    for k in range(n_phrases):
        s = off + step * (k + 1)
        signs[:, s:s + 3] = k + 1
        neg[:, s:s + 3] = torch.randint(3, V, (1, 3), generator=g)
    labels, neg_labels = pos.clone(), neg.clone()
    labels[:, :off], neg_labels[:, :off] = -100, -100
    ref = ids()
    ref_labels = ref.clone()
    ref_labels[:, :off] = -100
    ones = torch.ones(1, L, dtype=torch.bool)
    batch = dict(input_ids=pos, labels=labels, attention_mask=ones, neg_input_ids=neg, neg_labels=neg_labels, neg_attention_mask=ones.clone(),
                 pos_signs=signs, neg_signs=signs.clone(), ref_input_ids=ref, ref_labels=ref_labels, ref_attention_mask=ones.clone(),
                 images=bf16_exact(torch.randn(1, 3, image, image, generator=g)), ref_images=bf16_exact(torch.randn(1, 3, image, image, generator=g)))
    max_len = L - 1 + npatch                     # post-splice length: nothing is truncated
    return dict(cfg=cfg, ccfg=ccfg, W=W, C=C, fac=fac, r=r, alpha=alpha, batch=batch, max_len=max_len, loss_alpha=loss_alpha)


def grad_keys(case, layers_probed):
    keys = []
    for i in layers_probed:
        for sub, names in TARGETS:
            for n in names:
                keys.append("model.layers.%d.%s.%s" % (i, sub, n))
    return keys


def run_oracle(case, dtype=torch.float32, realisation="plain", layers_probed=None, threads=None):
    """The reference arithmetic on the host.  Returns dict(loss, alignment, divergence, pos_acc, neg_acc, grads {module: (dA, dB)}, proj grads, seconds)."""
    from oracle import dpa as odpa, realise
    if threads:
        torch.set_num_threads(threads)
    t0 = time.time()
    fac = case["fac"]
    lora = {k: v.clone().to(dtype).requires_grad_(True) for k, v in fac.items()}
    ref = odpa.TinyLlava(case["W"], case["cfg"], case["C"], case["ccfg"], case["max_len"], varlen=True, dtype=dtype)
    pol = odpa.TinyLlava(case["W"], case["cfg"], case["C"], case["ccfg"], case["max_len"], lora=fac, lora_scale=case["alpha"] / case["r"], varlen=True,
                         dtype=dtype)
    proj = {k: v.clone().to(dtype).requires_grad_(True) for k, v in case["W"].items() if "mm_projector" in k}
    pol.W = dict(pol.W)
    pol.W.update(proj)
    pol.lora = lora
    batch = {k: v.numpy() for k, v in case["batch"].items()}
    with realise.realisation(realisation):
        loss, parts = odpa.compute_loss(pol, ref, batch, case["loss_alpha"])
        loss.backward()
    layers_probed = range(case["cfg"]["num_hidden_layers"]) if layers_probed is None else layers_probed
    grads = {m: (lora[m + ".A"].grad.float().clone(), lora[m + ".B"].grad.float().clone()) for m in grad_keys(case, layers_probed)}
    return dict(loss=float(loss.detach()), alignment=float(parts["alignment"].detach()), divergence=float(parts["divergence"].detach()),
                pos_acc=parts["pos_acc"].detach().float().numpy(), neg_acc=parts["neg_acc"].detach().float().numpy(), grads=grads,
                proj={k: v.grad.float().clone() for k, v in proj.items()}, seconds=time.time() - t0)


def run_product(case, pairs_per_group=1, ref_rows_per_group=1, share_prefix="always", layers_probed=None, device="cuda"):
    """The product's engine (tuned GEMM table, prefix sharing and top-row pruning as the bench runs them) on the same case."""
    from halva_amd import dpa, gemm_tuning
    from model_util import build_product_models_from
    gemm_tuning.enable_tuned_gemms()
    t0 = time.time()
    pol, ref, _ = build_product_models_from(case["cfg"], case["ccfg"], case["W"], case["C"], case["fac"], case["r"], case["alpha"], case["max_len"], device)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    eng = dpa.DPAEngine(pol, ref, case["loss_alpha"], pairs_per_group, ref_rows_per_group, share_prefix=share_prefix)
    accs = []
    inner = eng.pair_group_loss

    def spy(batch_, plan, idx):
        out = inner(batch_, plan, idx)
        accs.append((list(idx), out[1][1].detach().float().cpu().numpy(), out[1][2].detach().float().cpu().numpy()))
        return out
    eng.pair_group_loss = spy
    loss = eng.loss(case["batch"], backward=True)
    torch.cuda.synchronize()
    n = case["batch"]["input_ids"].shape[0]
    pa = np.zeros((n,) + accs[0][1].shape[1:], np.float32)
    na = np.zeros_like(pa)
    for idx, a, b in accs:
        pa[idx], na[idx] = a, b
    r = case["r"]
    layers_probed = range(case["cfg"]["num_hidden_layers"]) if layers_probed is None else layers_probed
    grads = {}
    for i in layers_probed:
        for sub, grp in pol.model.layers[i].groups():
            for gi, nme in enumerate(grp.names):
                grads["model.layers.%d.%s.%s" % (i, sub, nme)] = (grp.A_cat.main_grad[gi * r:(gi + 1) * r].float().cpu().clone(),
                                                                 getattr(grp, nme).lora_B["default"].weight.main_grad.float().cpu().clone())
    proj = {}
    for idx in (0, 2):
        for kind in ("weight", "bias"):
            proj["model.mm_projector.%d.%s" % (idx, kind)] = getattr(pol.model.mm_projector[idx], kind).main_grad.float().cpu().clone()
    parts = {k: float(v) for k, v in eng.last_parts.items()}
    out = dict(loss=float(loss), alignment=parts["alignment"], divergence=parts["divergence"], pos_acc=pa, neg_acc=na, grads=grads, proj=proj,
               seconds=time.time() - t0, packing=eng.last_packing is not None)
    del eng, pol, ref, flat
    torch.cuda.empty_cache()
    return out


def compare(got, want):
    """errors of `got` against `want` (the fp32 oracle): loss / alignment / divergence absolute, phrase sums relative, margins absolute,
    gradients relative Frobenius per tensor (max and by name)"""
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    m_got, m_want = got["neg_acc"] - got["pos_acc"], want["neg_acc"] - want["pos_acc"]
    gerr = {}
    for m, (wa, wb) in want["grads"].items():
        ga, gb = got["grads"][m]
        gerr[m + ".A"], gerr[m + ".B"] = rel(ga, wa), rel(gb, wb)
    for k, w in want["proj"].items():
        gerr[k] = rel(got["proj"][k], w)
    nz = want["pos_acc"] != 0
    return dict(loss=abs(got["loss"] - want["loss"]), alignment=abs(got["alignment"] - want["alignment"]), divergence=abs(got["divergence"] - want["divergence"]),
                phrase_rel=float(max((np.abs(got["pos_acc"] - want["pos_acc"]) / np.maximum(np.abs(want["pos_acc"]), 1e-9))[nz].max(),
                                     (np.abs(got["neg_acc"] - want["neg_acc"]) / np.maximum(np.abs(want["neg_acc"]), 1e-9))[nz].max())),
                margin=float(np.abs(m_got - m_want).max()), margin_sign_ok=bool((np.sign(m_got) == np.sign(m_want)).all()),
                grad_max=max(gerr.values()), grad=gerr)
