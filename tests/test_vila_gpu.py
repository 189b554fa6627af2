"""GPU parity of the VILA path through the C-ABI kernels against the reference's own outputs
(tests/golden/vila_*.npz, produced by tests/golden/make_golden_vila.py) and against a plain torch fp32 statement of each
new kernel.  Integer outputs bit-exact; loss within 1e-3 (bf16, north_star), features / gradients within the bf16
bounds stated per assert."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from golden_util import load_npz, meta_of, tensors  # noqa: E402
from model_util import batch_of, build_product_vila  # noqa: E402


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())


@pytest.mark.parametrize("rows,d", [(7, 144), (300, 1152), (33, 4608), (5, 8192)])
def test_layernorm_fwd_and_param_grads(rows, d):
    from halva_amd import kernels as K
    from halva_amd.hip import call, ptr, stream_ptr
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = (torch.randn(rows, d, device="cuda", generator=g) * 2 + 0.5).bfloat16()
    w = (1 + 0.1 * torch.randn(d, device="cuda", generator=g)).bfloat16()
    b = (0.1 * torch.randn(d, device="cuda", generator=g)).bfloat16()
    y, stats = K.layernorm(x, w, b, 1e-6, want_stats=True)
    xf = x.float().requires_grad_(False)
    wf, bf = w.float().requires_grad_(True), b.float().requires_grad_(True)
    yr = F.layer_norm(xf, (d,), wf, bf, 1e-6)
    assert float((y.float() - yr).abs().max()) <= 2 ** -7 * float(yr.abs().max())            # one bf16 ulp of the largest value
    np.testing.assert_allclose(stats[:, 0].cpu().numpy(), xf.mean(1).cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(stats[:, 1].cpu().numpy(), (xf.var(1, unbiased=False) + 1e-6).rsqrt().cpu().numpy(), rtol=1e-5)
    dy = torch.randn(rows, d, device="cuda", generator=g).bfloat16()
    yr.backward(dy.float())
    dw = torch.zeros(d, device="cuda")
    db = torch.zeros(d, device="cuda")
    call("halva_layernorm_bwd_params", ptr(dy), ptr(x), ptr(stats), ptr(dw), ptr(db), rows, d, stream_ptr())
    assert _rel(dw, wf.grad) < 1e-5 and _rel(db, bf.grad) < 1e-5


def test_downsample2x2_bit_exact_against_reference():
    from halva_amd import kernels as K
    z = load_npz("vila_downsample.npz")
    for name in ("odd", "even", "siglip"):
        x = torch.from_numpy(z[name + ".x"]).bfloat16()
        # the block only moves values: run the reference's fp32 output through the same bf16 rounding
        want = torch.from_numpy(z[name + ".y"]).bfloat16()
        got = K.downsample2x2(x.cuda())
        assert torch.equal(got.cpu(), want), name


def test_mlp_downsample_projector_matches_reference():
    from halva_amd import kernels as K
    z = load_npz("vila_downsample.npz")
    W = {k: v.bfloat16().cuda() for k, v in tensors(z, "proj.w.").items()}
    for p in W.values():
        p.requires_grad_(True)
    x = torch.from_numpy(z["proj.x"]).bfloat16().cuda()
    y = K.downsample_mlp(x, W["layers.1.weight"], W["layers.1.bias"], 1e-5, W["layers.2.weight"], W["layers.2.bias"],
                         W["layers.4.weight"], W["layers.4.bias"])
    assert _rel(y, torch.from_numpy(z["proj.y"])) < 2e-2
    y.backward(torch.from_numpy(z["proj.gy"]).bfloat16().cuda())
    for k, w in W.items():
        assert _rel(w.grad, torch.from_numpy(z["proj.g." + k])) < 3e-2, k


@pytest.mark.parametrize("hw,p,d", [(48, 14, 144), (384, 14, 1152), (28, 14, 64)])
def test_vit_patch_embed_valid_conv_with_bias(hw, p, d):
    from halva_amd import kernels as K
    g = torch.Generator(device="cuda").manual_seed(hw)
    img = torch.randn(2, 3, hw, hw, device="cuda", generator=g).bfloat16()
    w = (torch.randn(d, 3, p, p, device="cuda", generator=g) * 0.05).bfloat16()
    b = (torch.randn(d, device="cuda", generator=g) * 0.1).bfloat16()
    kp = (3 * p * p + 7) // 8 * 8
    wk = torch.zeros(d, kp, dtype=torch.bfloat16, device="cuda")
    wk[:, :3 * p * p] = w.reshape(d, -1)
    got = K.vit_patch_embed(img, wk, b, p, d)
    want = F.conv2d(img.float(), w.float(), b.float(), stride=p).flatten(2).transpose(1, 2)
    assert got.shape == want.shape == (2, (hw // p) ** 2, d)
    assert _rel(got, want) < 5e-3


def _tower(z, prefix="w.", cfg_key="cfg", select_feature="cls_patch"):
    from types import SimpleNamespace
    from halva_amd.siglip import SiglipVisionConfig, SiglipVisionTower
    vt = SiglipVisionTower("fixture", args=SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature=select_feature),
                           delay_load=True, config=SiglipVisionConfig(**meta_of(z, cfg_key)), device="cuda")
    vt._alloc()
    vt.load_hf_state_dict(tensors(z, prefix))
    vt.is_loaded = True
    return vt


def test_siglip_tower_matches_reference():
    """head_dim 72 (2 heads x 72 = 144) run zero-padded to 128 lanes; 48 px / patch 14 -> 3x3 tokens, 6 px dropped."""
    z = load_npz("vila_siglip.npz")
    images = torch.from_numpy(z["images"]).cuda()
    vt = _tower(z)
    assert vt.head_dim == 72 and vt.head_pad == 128
    f = vt(images)
    assert f.shape == tuple(z["features"].shape)
    assert _rel(f, torch.from_numpy(z["features"])) < 2e-2
    f2 = _tower(z, select_feature="patch")(images)
    assert _rel(f2, torch.from_numpy(z["features_patch"])) < 2e-2


@pytest.mark.parametrize("case,side", [("right", "right"), ("left", "left"), ("trunc", "right")])
def test_vila_signed_splice_on_gpu(case, side):
    z = load_npz("vila_splice.npz")
    pol, _, _ = build_product_vila(z, lora=False, max_len=int(z[case + ".max_len"]), padding_side=side)
    ids, att = torch.from_numpy(z["ids"]).cuda(), torch.from_numpy(z["att"]).cuda()
    labels, signs = torch.from_numpy(z["labels"]).cuda(), torch.from_numpy(z["signs"]).cuda()
    images = torch.from_numpy(z["images"]).cuda()
    with torch.no_grad():
        out = pol.prepare_inputs_labels_for_multimodal_signed(ids, None, att, None, labels, images, signs)
    assert out[0] is None
    np.testing.assert_array_equal(out[5].cpu().numpy(), z[case + ".labels"])
    np.testing.assert_array_equal(out[6].cpu().numpy(), z[case + ".signs"])
    np.testing.assert_array_equal(out[2].cpu().numpy().astype(bool), z[case + ".mask"].astype(bool))
    want = torch.from_numpy(z[case + ".embeds"])
    assert _rel(out[4], want) < 2e-2
    # text rows are exact copies of bf16 table rows; pad rows exact zeros
    m = torch.from_numpy(z[case + ".mask"].astype(bool))
    assert float(out[4].cpu()[~m].abs().max() if (~m).any() else 0.0) == 0.0
    if case == "right":
        with torch.no_grad():
            o2 = pol.prepare_inputs_labels_for_multimodal(ids, None, att, None, labels, [images[:1], images[1:3], images[3:]])
        np.testing.assert_array_equal(o2[5].cpu().numpy(), z["unsigned.labels"])
        assert _rel(o2[4], torch.from_numpy(z["unsigned.embeds"])) < 2e-2


def _engine(z, ppg, rpg, share=None):
    from halva_amd import dpa
    pol, ref, lora = build_product_vila(z)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    return dpa.DPAEngine(pol, ref, float(z["alpha"]), ppg, rpg, share_prefix=share), pol, ref, flat, lora


@pytest.mark.parametrize("fixture", ["vila_step_init", "vila_step_multi"])
@pytest.mark.parametrize("ppg,rpg,share", [(8, 8, False), (2, 1, False), (8, 8, "always")])
def test_vila_step_matches_reference_golden(fixture, ppg, rpg, share):
    z = load_npz(fixture + ".npz")
    eng, pol, ref, flat, (r, alpha, fac) = _engine(z, ppg, rpg, share)
    batch = batch_of(z)
    loss = eng.loss(batch, backward=True)
    torch.cuda.synchronize()
    parts = {k: float(v) for k, v in eng.last_parts.items()}
    assert abs(float(loss) - float(z["out.loss"])) < 1e-3, (float(loss), float(z["out.loss"]))
    assert abs(parts["alignment"] - float(z["out.alignment"])) < 1e-3
    assert abs(parts["divergence"] - float(z["out.divergence"])) < 1e-3
    if not any(k.startswith("grad.") for k in z.files):
        return
    s = alpha / r
    checked = 0
    for i, layer in enumerate(pol.llm.model.layers):
        for sub, grp in layer.groups():
            for g, n in enumerate(grp.names):
                key = "grad.llm.model.layers.%d.%s.%s.weight" % (i, sub, n)
                if key not in z.files:
                    continue
                dW = torch.from_numpy(z[key])
                A = fac["model.layers.%d.%s.%s.A" % (i, sub, n)]
                Bm = fac["model.layers.%d.%s.%s.B" % (i, sub, n)]
                gA = grp.A_cat.main_grad[g * r:(g + 1) * r].cpu()
                gB = getattr(grp, n).lora_B["default"].weight.main_grad.cpu()
                refA, refB = s * Bm.T @ dW, s * dW @ A.T
                assert _rel(gA, refA) < 4e-2, key
                assert _rel(gB, refB) < 4e-2, key
                checked += 1
    assert checked >= 5
    sd = dict(pol.mm_projector.named_parameters())
    for k in [k for k in z.files if k.startswith("grad.mm_projector.")]:
        p = sd[k[len("grad.mm_projector."):]]
        assert _rel(p.main_grad, torch.from_numpy(z[k])) < 4e-2, k


def test_vila_forward_signs_api():
    """model(input_ids=, images=, labels=, attention_mask=, signs=) -> outputs.logits / .labels / .signs
    (vila llava_llama.py:99-177), and the reference-shaped trainer on top of it."""
    from halva_amd.dpa import concat_pos_neg
    z = load_npz("vila_step_init.npz")
    pol, ref, _ = build_product_vila(z)
    batch = batch_of(z)
    c_ids, c_lab, c_att, c_sig = (torch.from_numpy(a).cuda() for a in concat_pos_neg(batch))
    images = batch["images"].cuda()
    with torch.no_grad():
        out = pol(input_ids=c_ids, images=torch.cat([images, images], 0), labels=c_lab, attention_mask=c_att, signs=c_sig)
    np.testing.assert_array_equal(out.labels[:, 1:].cpu().numpy(), z["out.batch_labels"])
    np.testing.assert_array_equal(out.signs[:, 1:].cpu().numpy(), z["out.batch_signs"])
    assert out.logits.dtype == torch.float32 and out.logits.shape[:2] == out.labels.shape
    lp = out.logits[:, :-1].log_softmax(-1)
    tgt = out.labels[:, 1:].clamp(min=0)
    logps = lp.gather(2, tgt[..., None])[..., 0].cpu()
    B = images.shape[0]
    m = torch.from_numpy(z["out.batch_labels"] != -100)
    want = torch.cat([torch.from_numpy(z["out.pos_logps"]), torch.from_numpy(z["out.neg_logps"])])
    assert float(((logps - want) * m).abs().max()) < 0.1              # per-token bf16 logit noise
    assert abs(float(((logps - want) * m).sum() / m.sum())) < 1e-2    # unbiased
