"""The VILA-path oracle (oracle/vila.py) and the product's host-side splice plan against the goldens that
tests/golden/make_golden_vila.py produced by running the reference's own VILA code (CPU)."""
import numpy as np
import pytest
import torch

from golden_util import load_npz, meta_of, tensors, to_tensor
from halva_amd import splice as SP
from oracle import vila as OV


def test_downsample_block_bit_exact():
    z = load_npz("vila_downsample.npz")
    for name in ("odd", "even", "siglip"):
        y = OV.downsample(torch.from_numpy(z[name + ".x"]))
        np.testing.assert_array_equal(y.numpy(), z[name + ".y"], err_msg=name)


def test_mlp_downsample_projector_fwd_bwd():
    z = load_npz("vila_downsample.npz")
    W = {k: v.clone().requires_grad_(True) for k, v in tensors(z, "proj.w.").items()}
    y = OV.projector_downsample(torch.from_numpy(z["proj.x"]), W)
    np.testing.assert_allclose(y.detach().numpy(), z["proj.y"], atol=2e-5)
    y.backward(torch.from_numpy(z["proj.gy"]))
    for k, w in W.items():
        np.testing.assert_allclose(w.grad.numpy(), z["proj.g." + k], atol=5e-5, err_msg=k)


def test_siglip_tower():
    z = load_npz("vila_siglip.npz")
    cfg = meta_of(z, "cfg")
    W = tensors(z, "w.")
    images = torch.from_numpy(z["images"])
    f = OV.siglip_features(images, W, cfg, -2, "cls_patch")
    assert f.shape[1] == (cfg["image_size"] // cfg["patch_size"]) ** 2          # 48 px / 14 -> 3x3, 6 px dropped
    np.testing.assert_allclose(f.numpy(), z["features"], atol=3e-5)
    np.testing.assert_allclose(OV.siglip_features(images, W, cfg, -2, "patch").numpy(), z["features_patch"], atol=3e-5)


def _vila_from(z, max_len, side="right", lora=None, scale=0.0):
    return OV.TinyVila(tensors(z, "llm."), meta_of(z, "llama_cfg"), tensors(z, "vis."), meta_of(z, "vis_cfg"),
                       tensors(z, "proj."), max_len, lora=lora, lora_scale=scale, padding_side=side)


@pytest.mark.parametrize("case,side", [("right", "right"), ("left", "left"), ("trunc", "right")])
def test_signed_multi_image_splice(case, side):
    """Rows with 1 / 2 / 0 / 1 image tokens over 4 images: a text-only row consumes no image (vila llava_arch.py:708-718)."""
    z = load_npz("vila_splice.npz")
    max_len = int(z[case + ".max_len"])
    m = _vila_from(z, max_len, side)
    ids, att, labels, signs = z["ids"], z["att"], z["labels"], z["signs"]
    feats = m.encode_images(torch.from_numpy(z["images"]))
    tab = m.W["model.embed_tokens.weight"]
    e, l, s, msk = OV.host.splice(ids, att, labels, signs, feats.numpy(), tab.numpy(), max_len, side, imageless_consumes=False)
    np.testing.assert_array_equal(l, z[case + ".labels"])
    np.testing.assert_array_equal(s, z[case + ".signs"])
    np.testing.assert_array_equal(msk, z[case + ".mask"].astype(bool))
    np.testing.assert_allclose(e, z[case + ".embeds"], atol=3e-5)
    # product host plan: integer outputs bit-exact, and the source-row plan reproduces the embeddings
    plan = SP.plan_splice(ids, att, labels, signs, feats.shape[1], max_len, side, imageless_consumes=False)
    np.testing.assert_array_equal(plan.labels.numpy(), z[case + ".labels"])
    np.testing.assert_array_equal(plan.signs.numpy(), z[case + ".signs"])
    np.testing.assert_array_equal(plan.mask.numpy(), z[case + ".mask"].astype(bool))
    assert plan.n_images == 4
    src = plan.src.numpy()
    flat = feats.reshape(-1, feats.shape[-1]).numpy()
    rows = np.where(src[:, None] >= 0, tab.numpy()[np.maximum(src, 0)],
                    np.where(src[:, None] == -1, 0.0, flat[np.maximum(-src - 2, 0)]))
    np.testing.assert_allclose(rows.reshape(plan.S, plan.T, -1), z[case + ".embeds"], atol=3e-5)
    if case == "right":
        e2, l2, _, _ = OV.host.splice(ids, att, labels, None, feats.numpy(), tab.numpy(), max_len, side, imageless_consumes=False)
        np.testing.assert_array_equal(l2, z["unsigned.labels"])
        np.testing.assert_allclose(e2, z["unsigned.embeds"], atol=3e-5)


def test_image_slots_running_index():
    z = load_npz("vila_splice.npz")
    slots, used = SP.image_slots(z["ids"], z["att"], imageless_consumes=False)
    assert slots == [[0], [1, 2], [], [3]] and used == 4
    slots, used = SP.image_slots(z["ids"], z["att"], imageless_consumes=True)          # LLaVA twin: the text row burns one
    assert slots == [[0], [1, 2], [], [4]] and used == 5


def _step_models(z):
    r, a = z["lora_cfg"]
    max_len = int(z["max_len"])
    ref = _vila_from(z, max_len)
    lora = {k: v.clone().requires_grad_(True) for k, v in tensors(z, "lora.").items()}
    pol = _vila_from(z, max_len, lora=lora, scale=float(a / r))
    pol.proj_W = {k: v.clone().requires_grad_(True) for k, v in pol.proj_W.items()}
    return pol, ref, lora


@pytest.mark.parametrize("name", ["vila_step_init", "vila_step_multi"])
def test_vila_compute_loss(name):
    z = load_npz(name + ".npz")
    pol, ref, lora = _step_models(z)
    batch = {k[len("batch."):]: z[k] for k in z.files if k.startswith("batch.")}
    loss, parts = OV.compute_loss(pol, ref, batch, float(z["alpha"]))
    np.testing.assert_array_equal(parts["batch_labels"].numpy(), z["out.batch_labels"])
    np.testing.assert_array_equal(parts["batch_signs"].numpy(), z["out.batch_signs"])
    for k in ("pos_logps", "neg_logps", "pos_acc", "neg_acc"):
        np.testing.assert_allclose(parts[k].detach().numpy(), z["out." + k], atol=5e-5, err_msg=k)
    assert abs(float(loss) - float(z["out.loss"])) < 1e-5
    assert abs(float(parts["alignment"]) - float(z["out.alignment"])) < 2e-6      # the trainer logs round(x, 7)
    assert abs(float(parts["divergence"]) - float(z["out.divergence"])) < 2e-6
    if not any(k.startswith("grad.") for k in z.files):
        return
    loss.backward()
    for k in [k for k in z.files if k.startswith("grad.mm_projector.")]:
        np.testing.assert_allclose(pol.proj_W[k[len("grad.mm_projector."):]].grad.numpy(), z[k], atol=3e-5, err_msg=k)
    r, a = z["lora_cfg"]
    s = float(a / r)
    for k in [k for k in z.files if k.startswith("grad.llm.")]:
        mod = k[len("grad.llm."):-len(".weight")]
        dW = torch.from_numpy(z[k])
        A, Bm = lora[mod + ".A"], lora[mod + ".B"]
        np.testing.assert_allclose(A.grad.numpy(), (s * Bm.detach().T @ dW).numpy(), atol=3e-5, err_msg=k)
        np.testing.assert_allclose(Bm.grad.numpy(), (s * dW @ A.detach().T).numpy(), atol=3e-5, err_msg=k)
