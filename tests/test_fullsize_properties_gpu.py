"""Size-independent properties of the DPA step at the BASELINE.json geometry (LLaVA-1.5-7B widths: d=4096, 32 heads x 128,
F=11008, vocab 32000, CLIP-L/14@336 -> 576 patches, T=2048 post-splice, LoRA r=128; the layer count is cut to 2 so the test
runs in seconds - every kernel sees its full-size shapes).  The oracle cannot run at this size; these identities do not need it:
  * identical pos / neg responses   -> alignment == log(2) for every phrase slot (softplus(0)), up to bf16 GEMM noise: hipBLASLt's
    stream-K kernels reduce K in a tile-position-dependent order, so two identical rows of one GEMM are not bit-identical;
  * LoRA B == 0 (policy == reference) -> divergence == 0 up to the same noise (the policy's K-concatenated GEMM, K = 4096 + 384,
    sums in another order than the reference's K = 4096 one): the floor is ~2e-5 nat per response token;
  * the loss and the accumulated gradients do not depend on how the batch is cut into groups (the engine's pair-decomposition);
  * padding rows / truncation: a batch padded with extra pad tokens gives the same loss (masks are bit-exact, the varlen
    attention never sees the padding)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(layers=2, lora_b_std=0.01, seed=7):
    import bench
    from halva_amd import dpa
    from halva_amd.llava_model import build_random_llava
    geo = dict(bench.LLAMA_7B, num_hidden_layers=layers)
    clip = dict(bench.CLIP_L_336, num_hidden_layers=2)
    pol = build_random_llava(geo, clip, lora_r=128, lora_alpha=256, seed=seed, device="cuda", max_len=2048)
    if lora_b_std:
        g = torch.Generator(device="cuda").manual_seed(seed + 1)
        with torch.no_grad():
            for layer in pol.model.layers:
                for _, grp in layer.groups():
                    for n in grp.names:
                        getattr(grp, n).lora_B["default"].weight.normal_(0.0, lora_b_std, generator=g)
    ref = build_random_llava(geo, clip, seed=seed, device="cuda", max_len=2048, share_base_from=pol)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    return pol, ref, flat


def _batch(B, seed=11):
    import bench
    return bench.synthetic_batch(B, seed)


def _loss(pol, ref, flat, batch, ppg, rpg, alpha=0.4):
    from halva_amd import dpa
    eng = dpa.DPAEngine(pol, ref, alpha, ppg, rpg)
    flat.zero_grad()
    loss = float(eng.loss(batch, backward=True))
    torch.cuda.synchronize()
    return loss, {k: float(v) for k, v in eng.last_parts.items()}, flat.grad.clone()


def test_identical_pairs_give_log2_and_zero_lora_gives_zero_kl():
    pol, ref, flat = _models(lora_b_std=0.0)
    batch = _batch(2)
    for k in ("input_ids", "labels", "attention_mask"):
        batch["neg_" + k] = batch[k].clone()                 # hallucinated == correct
    loss, parts, grad = _loss(pol, ref, flat, batch, 2, 2)
    n_tok = int((batch["ref_labels"][:, 1:] != -100).sum())
    assert abs(parts["alignment"] - math.log(2.0)) < 2e-3, parts
    assert 0.0 <= parts["divergence"] < 1e-4 * n_tok / 2, (parts, n_tok)      # sum over tokens / B, B = 2
    assert abs(loss - math.log(2.0) - 0.4 * parts["divergence"]) < 2e-3
    assert torch.isfinite(grad).all()


def test_grouping_invariance_at_full_width():
    pol, ref, flat = _models()
    batch = _batch(4)
    l1, p1, g1 = _loss(pol, ref, flat, batch, 4, 4)
    l2, p2, g2 = _loss(pol, ref, flat, batch, 1, 1)
    l3, p3, g3 = _loss(pol, ref, flat, batch, 2, 3)
    for l, p in ((l2, p2), (l3, p3)):
        assert abs(l - l1) < 2e-3 * max(1.0, abs(l1)), (l, l1)          # bf16 GEMM tiling differs with the group's row count
        # (alignment: measured 0.5e-3 .. 2.8e-3 over both forward kernels on this batch - one draw of the GEMMs' bf16 rounding noise each)
        assert abs(p["alignment"] - p1["alignment"]) < 5e-3 and abs(p["divergence"] - p1["divergence"]) < 2e-3 * max(1.0, p1["divergence"])
    for g in (g2, g3):
        assert float((g - g1).norm() / g1.norm()) < 2e-2
    assert math.isfinite(l1) and p1["divergence"] > 0.0


def test_extra_padding_and_truncation_do_not_change_the_loss():
    pol, ref, flat = _models()
    batch = _batch(2)
    l1, p1, _ = _loss(pol, ref, flat, batch, 2, 2)
    padded = dict(batch)
    for ids, lab, att, sg in (("input_ids", "labels", "attention_mask", "pos_signs"),
                              ("neg_input_ids", "neg_labels", "neg_attention_mask", "neg_signs"),
                              ("ref_input_ids", "ref_labels", "ref_attention_mask", None)):
        n = 37
        B = batch[ids].shape[0]
        padded[ids] = torch.cat([batch[ids], torch.zeros(B, n, dtype=batch[ids].dtype)], 1)
        padded[lab] = torch.cat([batch[lab], torch.full((B, n), -100, dtype=batch[lab].dtype)], 1)
        padded[att] = torch.cat([batch[att], torch.zeros(B, n, dtype=torch.bool)], 1)
        if sg:
            padded[sg] = torch.cat([batch[sg], torch.zeros(B, n, dtype=batch[sg].dtype)], 1)
    l2, p2, _ = _loss(pol, ref, flat, padded, 2, 2)
    assert l2 == l1 and p2 == p1            # padding is removed by the mask before the splice: identical work, identical bits
    # a longer response is cut at tokenizer_model_max_length (2048) AFTER the splice, like the reference: the plan shows it
    from halva_amd import dpa
    import bench
    long = bench.synthetic_batch(2, 11, resp_len=1419 + 300)
    plan = dpa.DPAEngine(pol, ref, 0.4).make_plan(long)
    assert plan.T_full == 2048
