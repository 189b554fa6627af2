"""Pins the oracle (oracle/) to the reference: every golden vector under tests/golden/ was produced by
importing the reference itself (tests/golden/make_golden.py).  Integer outputs must match bit-exactly;
fp32 outputs to 1e-5 absolute (different summation order only)."""
import numpy as np
import pytest
import torch

from fake_tokenizer import FakeLlamaTokenizer
from golden_util import load_json, load_npz, meta_of, tensors
from oracle import dpa, host, nets


@pytest.fixture(scope="module")
def tok_fix():
    return load_json("tokenize_masks.json")


def test_masked_tokenisation_cases(tok_fix):
    tok = FakeLlamaTokenizer(tok_fix["model_max_length"], vocab=tok_fix["vocab"], frozen=True)
    n_ok = 0
    for c in tok_fix["cases"]:
        q = "<image>\n" + c["question"]
        if c["result"].startswith("raise"):
            with pytest.raises(RuntimeError):
                host.preprocess_v1(q, c["answer_masked"], c["answer"], tok)
            continue
        out = host.preprocess_v1(q, c["answer_masked"], c["answer"], tok)
        if c["result"] == "none":
            assert out is None
            continue
        n_ok += 1
        assert out["input_ids"] == c["input_ids"]
        assert out["signs"] == c["signs"]
        assert out["labels"] == c["labels"]
    assert n_ok >= 12


def test_ref_tokenisation_cases(tok_fix):
    tok = FakeLlamaTokenizer(tok_fix["model_max_length"], vocab=tok_fix["vocab"], frozen=True)
    for c in tok_fix["ref_cases"]:
        out = host.preprocess_v1_ref("<image>\n" + c["question"], c["answer"], tok)
        assert out["input_ids"] == c["input_ids"]
        assert out["labels"] == c["labels"]


def test_span_walker(tok_fix):
    tok = FakeLlamaTokenizer(tok_fix["model_max_length"], vocab=tok_fix["vocab"], frozen=True)
    for w in tok_fix["walk"]:
        ids, signs = host.walk_masked(w["string"], tok)
        assert ids == w["ids"] and signs == w["signs"]


def test_collator():
    z = load_npz("collator.npz")
    for ci, m in enumerate(meta_of(z)):
        inst = []
        for k in range(m["n"]):
            p = "c%d_in%d_" % (ci, k)
            inst.append({key[len(p):]: z[key] for key in z.files if key.startswith(p)})
        out = host.collate(inst, pad_token_id=0, model_max_length=m["max_len"])
        p = "c%d_out_" % ci
        keys = [key[len(p):] for key in z.files if key.startswith(p)]
        assert sorted(keys) == sorted(out.keys())
        for key in keys:
            np.testing.assert_array_equal(out[key], z[p + key], err_msg=key)


def test_sampler():
    fx = load_json("sampler.json")
    for c in fx["cases"]:
        g = torch.Generator().manual_seed(c["seed"])
        torch.manual_seed(c["global_seed"])
        assert host.modality_length_grouped_indices(c["lengths"], c["batch_size"], c["world_size"], g) == c["modality_indices"]
        g = torch.Generator().manual_seed(c["seed"])
        assert host.length_grouped_indices([abs(l) for l in c["lengths"]], c["batch_size"], c["world_size"], g) == c["length_indices"]
    for c in fx["chunks"]:
        assert host.split_to_even_chunks(c["indices"], c["lengths"], c["num_chunks"]) == c["out"]


def test_splice():
    z = load_npz("splice.npz")
    for ci, m in enumerate(meta_of(z)):
        p = "s%d_" % ci
        e, l, s, mk = host.splice(z[p + "ids"], z[p + "mask"], z[p + "labels"], z[p + "signs"], z[p + "features"],
                                  z[p + "embed_tokens"], m["max_len"], m["padding_side"])
        np.testing.assert_array_equal(l, z[p + "out_labels"])
        np.testing.assert_array_equal(s, z[p + "out_signs"])
        np.testing.assert_array_equal(mk, z[p + "out_mask"])
        np.testing.assert_array_equal(e, z[p + "out_embeds"])          # pure copies: exact
        e, l, s, mk = host.splice(z[p + "ref_ids"], z[p + "ref_mask"], z[p + "ref_labels"], None, z[p + "ref_features"],
                                  z[p + "embed_tokens"], m["max_len"], m["padding_side"])
        assert s is None
        np.testing.assert_array_equal(l, z[p + "ref_out_labels"])
        np.testing.assert_array_equal(mk, z[p + "ref_out_mask"])
        np.testing.assert_array_equal(e, z[p + "ref_out_embeds"])


def test_logp_and_phrase_accumulation():
    z = load_npz("loss_small.npz")
    logps = dpa.cal_batch_logp(torch.from_numpy(z["logits"]), torch.from_numpy(z["labels"]))
    np.testing.assert_allclose(logps.numpy(), z["logps"], atol=1e-6)
    acc = dpa.accumulate_logps(torch.from_numpy(z["logps"]), torch.from_numpy(z["signs"]))
    assert acc.shape == z["acc"].shape
    np.testing.assert_allclose(acc.numpy(), z["acc"], atol=1e-6)


def test_clip_tower_and_projector():
    z = load_npz("clip_tower.npz")
    cfg = meta_of(z, "cfg")
    W = tensors(z, "clip.")
    f = nets.clip_features(torch.from_numpy(z["images"]), W, cfg, -2)
    np.testing.assert_allclose(f.numpy(), z["features"], atol=2e-5)
    pw = {"model.mm_projector." + k: v for k, v in tensors(z, "proj.").items()}
    np.testing.assert_allclose(nets.projector(f, pw).numpy(), z["projected"], atol=2e-5)


def test_clip_tower_d64_fixture():
    """The fixture the GPU CLIP-tower test runs on (2 heads x 64, 16 patches + CLS): the oracle agrees with the reference on it."""
    z = load_npz("clip_tower_d64.npz")
    cfg = meta_of(z, "cfg")
    W = tensors(z, "clip.")
    f = nets.clip_features(torch.from_numpy(z["images"]), W, cfg, -2)
    np.testing.assert_allclose(f.numpy(), z["features"], atol=3e-5)
    np.testing.assert_allclose(f.numpy(), z["hidden_m2"][:, 1:], atol=3e-5)          # hidden_states[-2] with the CLS row dropped
    pw = {"model.mm_projector." + k: v for k, v in tensors(z, "proj.").items()}
    np.testing.assert_allclose(nets.projector(f, pw).numpy(), z["projected"], atol=3e-5)


def test_llama_layer_against_vendored_spec():
    z = load_npz("llama_layer.npz")
    cfg = meta_of(z, "llama_cfg")
    W = {"L." + k: v.clone().requires_grad_(True) for k, v in tensors(z, "w.").items()}
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    keep = torch.from_numpy(z["keep"])
    y = nets.decoder_layer(x, W, "L.", keep, cfg)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], atol=2e-5)
    y.backward(torch.from_numpy(z["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["gx"], atol=2e-5)
    for k, v in W.items():
        np.testing.assert_allclose(v.grad.numpy(), z["g." + k[2:]], atol=3e-5, err_msg=k)
    # RoPE tables and application
    cos, sin = nets.rope_tables(16, 20)
    np.testing.assert_allclose(cos.numpy(), z["cos"], atol=1e-6)
    qr = nets.rope_apply(torch.from_numpy(z["q"]), cos, sin, torch.arange(20)[None])
    np.testing.assert_allclose(qr.numpy(), z["q_rope"], atol=1e-6)
    # the un-padded causal form the GPU path uses agrees with the eager additive-mask form on valid rows
    y2 = nets.decoder_layer(x.detach(), {k: v.detach() for k, v in W.items()}, "L.", keep, cfg, varlen=True)
    np.testing.assert_allclose(y2.numpy()[keep.numpy()], z["y"][z["keep"]], atol=2e-5)


def _models_from(z):
    cfg, ccfg = meta_of(z, "llama_cfg"), meta_of(z, "clip_cfg")
    base = tensors(z, "base.")
    clipW = tensors(z, "clip.")
    lora = tensors(z, "lora.") or None
    r, a = z["lora_cfg"]
    max_len = int(z["max_len"])
    ref = dpa.TinyLlava(base, cfg, clipW, ccfg, max_len)
    pol_W = {k: v.clone() for k, v in base.items()}
    for k in pol_W:
        if "mm_projector" in k:
            pol_W[k].requires_grad_(True)
    if lora is not None:
        lora = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    pol = dpa.TinyLlava(pol_W, cfg, clipW, ccfg, max_len, lora=lora, lora_scale=float(a / r))
    pol.W = pol_W          # keep the leaf tensors (dtype cast above made copies)
    return pol, ref, lora


@pytest.mark.parametrize("name", ["dpa_step_a", "dpa_step_trunc", "dpa_step_identity", "dpa_step_d64", "dpa_step_d64_init",
                                  "dpa_step_d128_init", "dpa_step_d128_long"])
def test_compute_loss(name):
    z = load_npz(name + ".npz")
    pol, ref, lora = _models_from(z)
    batch = {k[len("batch."):]: z[k] for k in z.files if k.startswith("batch.")}
    loss, parts = dpa.compute_loss(pol, ref, batch, float(z["alpha"]))
    np.testing.assert_array_equal(parts["batch_labels"].numpy(), z["out.batch_labels"])
    np.testing.assert_array_equal(parts["batch_signs"].numpy(), z["out.batch_signs"])
    if "out.all_logits" in z.files:
        np.testing.assert_allclose(parts["all_logits"].detach().numpy(), z["out.all_logits"], atol=3e-5)
    np.testing.assert_allclose(parts["pos_logps"].detach().numpy(), z["out.pos_logps"], atol=3e-5)
    np.testing.assert_allclose(parts["neg_logps"].detach().numpy(), z["out.neg_logps"], atol=3e-5)
    np.testing.assert_allclose(parts["pos_acc"].detach().numpy(), z["out.pos_acc"], atol=3e-5)
    np.testing.assert_allclose(parts["neg_acc"].detach().numpy(), z["out.neg_acc"], atol=3e-5)
    assert abs(float(parts["alignment"]) - float(z["out.alignment"])) < 1e-5
    assert abs(float(parts["divergence"]) - float(z["out.divergence"])) < 1e-5
    assert abs(float(loss) - float(z["out.loss"])) < 1e-5
    if name == "dpa_step_identity":
        assert float(parts["divergence"]) == 0.0          # LoRA-free policy == ref
    loss.backward()
    # projector grads directly; LoRA grads through the chain rule from the reference's dense dL/dW
    for k in [k for k in z.files if k.startswith("grad.") and "mm_projector" in k]:
        np.testing.assert_allclose(pol.W[k[len("grad."):]].grad.numpy(), z[k], atol=3e-5, err_msg=k)
    if lora is not None:
        r, a = z["lora_cfg"]
        s = float(a / r)
        for k in [k for k in z.files if k.startswith("grad.") and "proj.weight" in k and "mm_projector" not in k]:
            mod = k[len("grad."):-len(".weight")]
            dW = torch.from_numpy(z[k])
            A, Bm = lora[mod + ".A"], lora[mod + ".B"]
            np.testing.assert_allclose(A.grad.numpy(), (s * Bm.detach().T @ dW).numpy(), atol=3e-5, err_msg=k)
            np.testing.assert_allclose(Bm.grad.numpy(), (s * dW @ A.detach().T).numpy(), atol=3e-5, err_msg=k)


@pytest.mark.parametrize("name", ["dpa_step_d64_init", "dpa_step_d128_init", "dpa_step_d64", "dpa_step_d128_long"])
def test_bf16_realisation_table_is_live(name):
    """The GPU step test bounds the product's errors by what the reference arithmetic itself (this oracle) does when it runs in bf16 under the
    realisations of oracle/realise.py (tests/golden/bf16_realisations.json, written by tools/measure_bf16_floors.py --write).  The committed
    table must not be inflated: three of its realisations are re-measured here and every entry must lie within [0.5, 2] x the live value
    (the CPU GEMM library may block a contraction differently with another thread count: another summation order, i.e. one more realisation).
    And the realisations do not change the mathematics: in fp32 they reproduce the reference's numbers like the plain oracle does."""
    import test_dpa_step_gpu as G
    z = load_npz(name + ".npz")
    for real in ("plain", "perm1", "chunk2"):
        live = dict(zip(G.REAL_COLS, G._bf16_realisation(name, z, real)))
        kept = G.REALISED[name][real]
        for col in ("margin", "grad", "pos_acc", "neg_acc"):
            assert 0.5 * live[col] <= kept[col] <= 2.0 * live[col], (name, real, col, kept[col], live[col])
    assert max(G._set(name, "margin")) > 1e-3                                    # bf16 alone already breaks 1e-3 absolute on the margins
    f32 = dict(zip(G.REAL_COLS, G._bf16_realisation(name, z, "perm3_chunk2", dtype=torch.float32)))
    assert f32["loss"] < 2e-5 and f32["margin"] < 2e-4 and f32["grad"] < 1e-3, f32


def test_training_curve_oracle_starts_at_the_reference_loss():
    """oracle/curve.py: step 0 of both curves (fp32 parameters / the recipe's bf16 parameter copy) is the reference's own compute_loss
    value on the fixture when nothing has been rounded yet (the fixture's trainable tensors are bf16-representable), and the curve
    moves.  The later steps restate inherited HF Trainer / DeepSpeed semantics and are not pinned to a reference run (oracle/curve.py)."""
    from oracle import curve
    z = load_npz("dpa_step_d64_init.npz")
    r, a = z["lora_cfg"]
    batch = {k[len("batch."):]: z[k] for k in z.files if k.startswith("batch.")}
    for bf16_params in (False, True):
        c = curve.training_curve(tensors(z, "base."), tensors(z, "clip."), meta_of(z, "llama_cfg"), meta_of(z, "clip_cfg"), int(z["max_len"]),
                                 tensors(z, "lora."), r, a, batch, float(z["alpha"]), 2, 2e-3, 1e-3, bf16_params=bf16_params)
        assert abs(c[0] - float(z["out.loss"])) < 1e-5, (bf16_params, c[0], float(z["out.loss"]))
        assert c[1] < c[0] - 0.05
