"""GPU parity of the whole DPA step (product engine through the C-ABI kernels) against (a) the reference's own
outputs captured in tests/golden/dpa_step_d64.npz (fp32 CPU run of the reference on bf16-exact weights) and (b) the
oracle.  Tolerances: loss / alignment / divergence / phrase log-prob sums within 1e-3 relative-or-absolute combined
bound stated per assert (north_star: "within 1e-3 bf16"); integer masks bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_npz, tensors  # noqa: E402
from model_util import batch_of, build_product_models  # noqa: E402


def _engine(z, pairs_per_group, ref_rows_per_group, share_prefix=None):
    from halva_amd import dpa
    pol, ref, lora = build_product_models(z)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    eng = dpa.DPAEngine(pol, ref, float(z["alpha"]), pairs_per_group, ref_rows_per_group, share_prefix=share_prefix)
    return eng, pol, ref, flat, lora


# Two fixtures of the same head_dim-64 geometry, both produced by the reference itself:
#   dpa_step_d64_init  weights N(0, 0.02) like a real checkpoint/init; running the ORACLE in bf16 instead of fp32 moves its
#                      loss by 4e-4, so the north-star bound (1e-3) is meaningful here;
#   dpa_step_d64       "stress": weights N(0, 0.06), KL 0.55; bf16 alone moves the oracle's loss by 2.7e-3, so the bound is
#                      8e-3 (3x the bf16 noise floor); used for the gradient checks (large, well-conditioned gradients).
#   dpa_step_d128_init head_dim 128 (2 heads x 128, hidden 256) - the headline attention instantiation - N(0, 0.02) weights,
#                      4 pairs with ~120-token responses (T ~ 140: the causal kernel crosses 64-key tile boundaries); 1e-3.
#   dpa_step_d128_long (round 4) the same geometry at MULTI-BLOCK length: responses of 300 / 517 / 806 / 1100 tokens (the last one cut at
#                      tokenizer_model_max_length 1024), correct and hallucinated phrases of different lengths (everything behind the
#                      first phrase is shifted between the two rows), 2-3 phrases per sample: post-splice rows span 4 row blocks of
#                      256 and 8 key blocks of 128, packed rows cross 256-row blocks inside branch B - the block pairing of the
#                      forward / dQ kernels and sdpa_bwd_dkv3's queues run INSIDE a step checked against the reference's own numbers.
#                      Its loss is 4.41 = alignment 4.04 + 0.4 x divergence 0.935, the divergence a SUM over ~2 600 response tokens / 4:
#                      the reference arithmetic re-run in bf16 on the CPU (the oracle, dtype=bf16) is already off by 1.9e-3 / 1.9e-4 / 4.3e-3
#                      (loss / alignment / divergence); the product measures 0.7e-3 / 1.3e-3-1.7e-3 / 2.6e-3 (4e-4 / 2.8e-3 relative).
#                      Bounds 2e-3 / 2.5e-3 / 5e-3: the 1e-3 absolute of the short fixtures is 2.5e-4 relative here.
FIXTURES = {"dpa_step_d64_init": (1e-3, 1e-3, 1e-3), "dpa_step_d64": (8e-3, 8e-3, 8e-3), "dpa_step_d128_init": (1e-3, 1e-3, 1e-3),
            "dpa_step_d128_long": (2e-3, 2.5e-3, 5e-3)}
# Per-phrase log-prob sums (values ~ -10 nat: two-token phrases at vocab 160) are held to 1e-3 RELATIVE on the realistic-init
# fixtures.  The margins neg_acc - pos_acc are differences of two such sums; their absolute error is bounded by the bf16 noise
# floor of the reference's OWN arithmetic: the oracle (CPU restatement, pinned to the reference at 1e-6 in fp32) re-run with bf16
# tensors moves the margins by 3.9e-3 (d64_init) / 1.1e-2 (d128_init) / 2.2e-2 (d64 stress) - a residual stream held in bf16 carries
# 2^-9 relative noise per rounding, whatever executes it.  The product (fp32 accumulation inside every kernel) must not be worse
# than that floor (observed 2.9e-3 / 5.4e-3 / 1.6e-2), and never worse than 1e-3 where the floor is lower.
REL_TOL = {"dpa_step_d64_init": 1e-3, "dpa_step_d128_init": 1e-3, "dpa_step_d64": 2e-3, "dpa_step_d128_long": 1e-3}
# max |margin error| of the oracle run in bf16 (measured in the build container by _bf16_floor below; the CPU test
# tests/test_oracle_vs_golden.py::test_bf16_floor_constants re-measures it and fails if these are more than 2x a live measurement)
MARGIN_FLOOR = {"dpa_step_d64_init": 3.9e-3, "dpa_step_d128_init": 1.13e-2, "dpa_step_d64": 2.2e-2, "dpa_step_d128_long": 5.6e-3}
# Gradients (LoRA factors through the chain rule from the reference's dense dL/dW, projector directly), relative Frobenius error per
# tensor: the same floor argument.  The oracle re-run in bf16 on the CPU is 1.27e-2 / 1.50e-2 / 2.03e-2 away from the reference's fp32
# gradients on the three fixtures (max over the tensors; _bf16_grad_floor below re-measures it, tests/test_oracle_vs_golden.py holds
# the constants to a live measurement); the product is bound by 1.25 x that floor on every fixture (round 2 checked only the stress
# fixture, at a flat 3e-2).
GRAD_FLOOR = {"dpa_step_d64_init": 1.27e-2, "dpa_step_d128_init": 1.50e-2, "dpa_step_d64": 2.03e-2, "dpa_step_d128_long": 1.46e-2}
# The long fixture's phrase sums reach -38 nat over rows of ~1000 tokens and its margins 28 nat: one bf16 realisation of the reference arithmetic
# on the CPU (the floor above) is off by 5.6e-3, the product - another realisation of the same roundings, all four grouping / sharing variants -
# by 0.98e-2-1.23e-2 (either forward kernel), i.e. 3e-4 of the sums involved (which are held to 1e-3 relative).  Bound: 2.5 x the floor for that fixture.
MARGIN_FACTOR = {"dpa_step_d128_long": 2.5}
_floor_cache = {}
_gfloor_cache = {}


def _ref_factor_grads(z, fac, r, alpha):
    """{key: (want dA, want dB)} from the reference's dense weight gradients: dA = s B^T dW, dB = s dW A^T."""
    s = alpha / r
    out = {}
    for k in [k for k in z.files if k.startswith("grad.") and "mm_projector" not in k]:
        mod = k[len("grad."):-len(".weight")]
        dW = torch.from_numpy(z[k])
        A, Bm = fac[mod + ".A"], fac[mod + ".B"]
        out[mod] = (s * Bm.T @ dW, s * dW @ A.T)
    return out


def _bf16_grad_floor(name, z):
    """max relative gradient error (over the LoRA factors and projector tensors the fixture holds) of the reference arithmetic itself
    (oracle) run in bf16 on the CPU, against the reference's fp32 gradients."""
    if name not in _gfloor_cache:
        from golden_util import meta_of
        from oracle import dpa as odpa
        cfg, ccfg = meta_of(z, "llama_cfg"), meta_of(z, "clip_cfg")
        base, clipW = tensors(z, "base."), tensors(z, "clip.")
        r, a = z["lora_cfg"]
        bf = torch.bfloat16
        fac = tensors(z, "lora.")
        lora = {k: v.clone().to(bf).requires_grad_(True) for k, v in fac.items()}
        ref = odpa.TinyLlava(base, cfg, clipW, ccfg, int(z["max_len"]), dtype=bf)
        pol = odpa.TinyLlava(base, cfg, clipW, ccfg, int(z["max_len"]), lora=fac, lora_scale=float(a / r), dtype=bf)
        proj = {k: v.clone().to(bf).requires_grad_(True) for k, v in base.items() if "mm_projector" in k}
        pol.W.update(proj)
        pol.lora = lora
        loss, _ = odpa.compute_loss(pol, ref, {k[len("batch."):]: z[k] for k in z.files if k.startswith("batch.")}, float(z["alpha"]))
        loss.backward()
        errs = []
        for mod, (wa, wb) in _ref_factor_grads(z, fac, float(r), float(a)).items():
            errs.append(float((lora[mod + ".A"].grad.float() - wa).norm() / wa.norm()))
            errs.append(float((lora[mod + ".B"].grad.float() - wb).norm() / wb.norm()))
        for k in [k for k in z.files if k.startswith("grad.") and "mm_projector" in k]:
            want = torch.from_numpy(z[k])
            errs.append(float((proj[k[len("grad."):]].grad.float() - want).norm() / want.norm()))
        _gfloor_cache[name] = max(errs)
    return _gfloor_cache[name]


def _bf16_floor(name, z):
    """max |error| of (pos_acc, neg_acc, margin) when the reference arithmetic itself (oracle) runs in bf16 on the CPU."""
    if name not in _floor_cache:
        from golden_util import meta_of
        from oracle import dpa as odpa
        cfg, ccfg = meta_of(z, "llama_cfg"), meta_of(z, "clip_cfg")
        base, clipW = tensors(z, "base."), tensors(z, "clip.")
        r, a = z["lora_cfg"]
        bf = torch.bfloat16
        ref = odpa.TinyLlava(base, cfg, clipW, ccfg, int(z["max_len"]), dtype=bf)
        pol = odpa.TinyLlava(base, cfg, clipW, ccfg, int(z["max_len"]), lora=tensors(z, "lora."), lora_scale=float(a / r), dtype=bf)
        with torch.no_grad():
            _, parts = odpa.compute_loss(pol, ref, {k[len("batch."):]: z[k] for k in z.files if k.startswith("batch.")}, float(z["alpha"]))
        pa, na = parts["pos_acc"].float().numpy(), parts["neg_acc"].float().numpy()
        _floor_cache[name] = (np.abs(pa - z["out.pos_acc"]).max(), np.abs(na - z["out.neg_acc"]).max(),
                              np.abs((na - pa) - (z["out.neg_acc"] - z["out.pos_acc"])).max())
    return _floor_cache[name]


@pytest.mark.parametrize("fixture", list(FIXTURES))
@pytest.mark.parametrize("ppg,rpg,share", [(8, 8, False), (2, 1, False), (8, 8, "always"), (1, 2, "always")])
def test_step_matches_reference_golden(ppg, rpg, share, fixture):
    """share = "always": the two rows of every pair run as one branched row [prefix | correct | pad | hallucinated] (prefix run
    once) - same reference numbers, same tolerances."""
    tol_loss, tol_align, tol_div = FIXTURES[fixture]
    z = load_npz(fixture + ".npz")
    eng, pol, ref, flat, (r, alpha, fac) = _engine(z, ppg, rpg, share)
    batch = batch_of(z)
    margins = []
    inner = eng.pair_group_loss

    def spy(batch_, plan, idx):            # keep what every pair group hands back: (logp_dense, pos_acc, neg_acc)
        out = inner(batch_, plan, idx)
        margins.append((list(idx), out[1][1].detach().float().cpu().numpy(), out[1][2].detach().float().cpu().numpy()))
        return out
    eng.pair_group_loss = spy
    loss = eng.loss(batch, backward=True)
    torch.cuda.synchronize()
    assert (eng.last_packing is not None) == (share == "always")
    # per-phrase log-prob sums and margins (north_star: "loss and phrase log-prob margins within 1e-3"), against the reference's
    # own pos_acc / neg_acc (halva_trainer.py:562-568), for every pair and phrase slot
    pos_acc = np.zeros_like(z["out.pos_acc"])
    neg_acc = np.zeros_like(z["out.neg_acc"])
    for idx, pa, na in margins:
        pos_acc[idx], neg_acc[idx] = pa, na
    rel = REL_TOL[fixture]
    f_margin = MARGIN_FLOOR[fixture]
    for got_acc, want_acc in ((pos_acc, z["out.pos_acc"]), (neg_acc, z["out.neg_acc"])):
        err = np.abs(got_acc - want_acc)
        assert (err <= rel * np.abs(want_acc) + 1e-6).all(), (fixture, err, want_acc)            # 1e-3 of the phrase log-prob sum
    margin, want_margin = neg_acc - pos_acc, z["out.neg_acc"] - z["out.pos_acc"]
    m_err = np.abs(margin - want_margin).max()
    assert m_err <= max(1e-3, MARGIN_FACTOR.get(fixture, 1.0) * f_margin), (fixture, m_err, f_margin, margin, want_margin)
    assert (np.sign(margin) == np.sign(want_margin)).all()                      # which answer every phrase prefers: unchanged
    print("%s ppg=%s share=%s: max |margin err| %.2e (bf16 floor of the reference arithmetic %.2e), max rel phrase-sum err %.2e"
          % (fixture, ppg, share, m_err, f_margin,
             max((np.abs(pos_acc - z["out.pos_acc"]) / np.maximum(np.abs(z["out.pos_acc"]), 1e-9))[z["out.pos_acc"] != 0].max(),
                 (np.abs(neg_acc - z["out.neg_acc"]) / np.maximum(np.abs(z["out.neg_acc"]), 1e-9))[z["out.neg_acc"] != 0].max())))
    got = float(loss)
    parts = {k: float(v) for k, v in eng.last_parts.items()}
    assert abs(got - float(z["out.loss"])) < tol_loss, (got, float(z["out.loss"]))
    assert abs(parts["alignment"] - float(z["out.alignment"])) < tol_align, (parts, float(z["out.alignment"]))
    assert abs(parts["divergence"] - float(z["out.divergence"])) < tol_div, (parts, float(z["out.divergence"]))
    # gradients on EVERY fixture: LoRA factors via the chain rule from the reference's dense dL/dW, projector directly; bound = the
    # bf16 floor of the reference arithmetic (GRAD_FLOOR) x 1.25
    bound = 1.25 * GRAD_FLOOR[fixture]
    want = _ref_factor_grads(z, fac, float(r), float(alpha))
    checked, worst = 0, 0.0
    for i, layer in enumerate(pol.model.layers):
        for sub, grp in layer.groups():
            for g, n in enumerate(grp.names):
                mod = "model.layers.%d.%s.%s" % (i, sub, n)
                if mod not in want:
                    continue
                gA = grp.A_cat.main_grad[g * r:(g + 1) * r].cpu()
                gB = getattr(grp, n).lora_B["default"].weight.main_grad.cpu()
                refA, refB = want[mod]
                eA, eB = float((gA - refA).norm() / refA.norm()), float((gB - refB).norm() / refB.norm())
                assert eA < bound and eB < bound, (fixture, mod, eA, eB, bound)
                worst = max(worst, eA, eB)
                checked += 1
    assert checked >= 5
    for k in [k for k in z.files if k.startswith("grad.") and "mm_projector" in k]:
        idx = int(k.split("mm_projector.")[1].split(".")[0])
        kind = k.rsplit(".", 1)[1]
        p = getattr(pol.model.mm_projector[idx], kind)
        refg = torch.from_numpy(z[k])
        e = float((p.main_grad.cpu() - refg).norm() / refg.norm())
        assert e < bound, (fixture, k, e, bound)
        worst = max(worst, e)
    print("%s ppg=%s share=%s: max relative gradient error %.2e (bf16 floor of the reference arithmetic %.2e, bound %.2e)"
          % (fixture, ppg, share, worst, GRAD_FLOOR[fixture], bound))


def test_compat_api_matches_golden():
    """The reference-shaped API (HalvaTrainer.concatenated_forward / compute_loss with full logits) on the same fixture."""
    import types
    from llava.train.halva_trainer import HalvaTrainer
    z = load_npz("dpa_step_d64_init.npz")
    from halva_amd import dpa
    pol, ref, _ = build_product_models(z)
    dpa.set_grad_sink(pol, False)
    stub = types.SimpleNamespace(model=pol, ref_model=ref, loss_alpha=float(z["alpha"]), label_pad_token_id=-100,
                                 is_encoder_decoder=False)
    for n in ("cal_batch_logp", "accumulate_logps", "concatenated_forward", "reference_forward", "compute_loss"):
        setattr(stub, n, types.MethodType(getattr(HalvaTrainer, n), stub))
    batch = {k: (v.cuda() if v.dtype.is_floating_point else v.cuda()) for k, v in batch_of(z).items()}
    batch["images"], batch["ref_images"] = batch["images"].bfloat16(), batch["ref_images"].bfloat16()
    pos_logps, neg_logps, labels, logits, signs = stub.concatenated_forward(pol, batch)
    np.testing.assert_array_equal(labels.cpu().numpy(), z["out.batch_labels"])          # token-index masks: bit exact
    np.testing.assert_array_equal(signs.cpu().numpy(), z["out.batch_signs"])
    m = z["out.batch_labels"] != -100
    B = pos_logps.shape[0]
    assert np.abs(pos_logps.detach().cpu().numpy() - z["out.pos_logps"])[m[:B]].max() < 2e-2
    loss = stub.compute_loss(pol, batch)
    assert abs(float(loss) - float(z["out.loss"])) < 1e-3
    loss.backward()
    assert pol.model.mm_projector[0].weight.grad is not None


def test_identity_policy_has_zero_divergence():
    """LoRA B = 0 (fresh adapter) => policy == reference => divergence == 0."""
    from halva_amd import dpa
    from halva_amd.llama import add_lora
    z = load_npz("dpa_step_d64.npz")
    pol, ref, _ = build_product_models(z, lora=False)
    add_lora(pol, 4, 8.0)
    pol._use_lora = True
    for p in pol.model.mm_projector.parameters():
        p.requires_grad_(True)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    eng = dpa.DPAEngine(pol, ref, 0.4, 8, 8)
    eng.loss(batch_of(z), backward=True)
    assert abs(float(eng.last_parts["divergence"])) < 1e-6
    assert float(flat.grad.abs().sum()) > 0


def test_optimizer_step_moves_lora_not_projector():
    from halva_amd import dpa
    z = load_npz("dpa_step_d64.npz")
    eng, pol, ref, flat, _ = _engine(z, 8, 8)
    opt = dpa.AdamWFlat(flat, lr=1e-3, weight_decay=0.0, mm_projector_lr=0.0)
    before = flat.master.clone()
    l0 = float(eng.loss(batch_of(z), backward=True))
    opt.step()
    flat.zero_grad()
    lo, hi = flat.segment(lambda n: "mm_projector" in n and "bias" not in n)
    assert torch.equal(flat.master[lo:hi], before[lo:hi])                 # mm_projector_lr 0 (src/hallava_7b.sh:33)
    assert not torch.equal(flat.master[:lo], before[:lo])
    l1 = float(eng.loss(batch_of(z), backward=False))
    assert l1 < l0


def test_dgrad_layout_switch_gives_the_same_gradients(monkeypatch):
    """HALVA_DGRAD_WT: dx = dy W through the transposed weight copy (NT GEMM, default) or the stored weight (NN GEMM) - the same
    numbers.  HALVA_DGRAD_MERGED (default on, needs the copy): the copy's base rows hold (W + scale B A)^T, so the LoRA path's share
    of dx comes out of the same GEMM instead of a second pass - a different order of bf16 roundings: the flat gradient moves by
    ~6e-3 of its norm, half the ~1.2e-2 that EITHER form is away from the reference's fp32 gradients (tools/diag_dgrad_merged.py:
    max dA / dB error 1.38e-2 / 1.35e-2 merged, 1.38e-2 / 1.33e-2 unmerged on this fixture)."""
    import halva_amd.llama as L
    z = load_npz("dpa_step_d64.npz")
    grads = {}
    for name, (copy, merged) in {"copy+merged": (True, True), "copy": (True, False), "stored": (False, False)}.items():
        monkeypatch.setattr(L, "DGRAD_TRANSPOSED_COPY", copy)
        monkeypatch.setattr(L, "DGRAD_MERGED", merged)
        eng, pol, ref, flat, _ = _engine(z, 8, 8)
        groups = [grp for layer in pol.model.layers for _, grp in layer.groups()]
        assert all((g.weight_cat_t is not None) == copy and g.dgrad_merged == merged for g in groups)
        loss = float(eng.loss(batch_of(z), backward=True))
        grads[name] = (loss, flat.grad.clone())
    assert grads["copy"][0] == grads["stored"][0] == grads["copy+merged"][0]     # the forward does not depend on any of it
    n = grads["stored"][1].norm()
    assert float((grads["copy"][1] - grads["stored"][1]).norm() / n) < 2e-3
    assert float((grads["copy+merged"][1] - grads["stored"][1]).norm() / n) < 1.2e-2
