"""Loss CURVE parity (north_star: "loss curve matching reference within 1e-3"): eight optimizer steps of the product (bf16
compute through the C-ABI kernels, fp32 master weights, flat fp32 gradient buffer, AdamW with the reference's parameter groups)
against the same eight steps of the oracle (fp32 torch-CPU restatement of the reference's compute_loss + torch.optim.AdamW) from
the same reference-generated fixture (weights, LoRA factors, batch).

Two oracle curves:
  * "bf16 parameters": what the reference's run holds (`--bf16 True` + DeepSpeed bf16, src/hallava_7b.sh:48, src/json/zero3.json):
    fp32 master weights and AdamW state, the COMPUTE copy of every trainable tensor rounded to bf16 after each step, arithmetic in
    fp32.  This is the curve the product must follow: every step within 1e-3 (the north-star bound);
  * "fp32 parameters": no rounding anywhere (plain torch.optim.AdamW on fp32 leaves).  Kept as a second witness: at lr 2e-3 (400x
    the recipe's 5e-6, so that eight steps move the loss by 0.5) the bf16 rounding of the updated factors is ~5 % of an update, and
    the two ORACLE curves themselves differ by up to ~1e-3; the product is held to 1e-3 at step 0 and on the curve's mean."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_npz, meta_of, tensors  # noqa: E402
from model_util import batch_of, build_product_models  # noqa: E402

STEPS, LR, PROJ_LR, ALPHA = 8, 2e-3, 1e-3, 0.4


def _oracle_curve(z, bf16_params=False):
    """oracle/curve.py (the restatement of the recipe's optimizer semantics lives with the oracle, with its citations)."""
    from oracle import curve
    r, a = z["lora_cfg"]
    batch = {k[len("batch."):]: z[k] for k in z.files if k.startswith("batch.")}
    return curve.training_curve(tensors(z, "base."), tensors(z, "clip."), meta_of(z, "llama_cfg"), meta_of(z, "clip_cfg"),
                                int(z["max_len"]), tensors(z, "lora."), r, a, batch, ALPHA, STEPS, LR, PROJ_LR, bf16_params=bf16_params)


def _product_curve(z, ppg, rpg):
    from halva_amd import dpa
    pol, ref, _ = build_product_models(z)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
    dpa.bind_model(flat, pol)
    dpa.set_grad_sink(pol, True)
    opt = dpa.AdamWFlat(flat, lr=LR, weight_decay=0.0, mm_projector_lr=PROJ_LR)
    eng = dpa.DPAEngine(pol, ref, ALPHA, ppg, rpg)
    batch = batch_of(z)
    curve = []
    for _ in range(STEPS):
        flat.zero_grad()
        curve.append(float(eng.loss(batch, backward=True)))
        opt.step()
    return curve


@pytest.mark.parametrize("ppg,rpg", [(8, 8), (1, 2)])
def test_loss_curve_matches_oracle(ppg, rpg):
    z = load_npz("dpa_step_d64_init.npz")
    want = _oracle_curve(z, bf16_params=True)
    want32 = _oracle_curve(z, bf16_params=False)
    got = _product_curve(z, ppg, rpg)
    assert abs(want[0] - float(z["out.loss"])) < 1e-5                            # step 0 of the oracle IS the reference's own loss
    assert abs(want32[0] - float(z["out.loss"])) < 1e-5
    assert want[-1] < want[0] - 0.05, want                                       # the curve really moves (lr 2e-3, 8 steps)
    print("oracle bf16-params", [round(x, 5) for x in want])
    print("oracle fp32-params", [round(x, 5) for x in want32])
    print("product           ", [round(x, 5) for x in got])
    d = np.abs(np.array(got) - np.array(want))
    assert d.max() < 1e-3, "product %s vs oracle (bf16 parameter copy) %s: |diff| %s" % (got, want, d)
    d32 = np.abs(np.array(got) - np.array(want32))
    assert d32[0] < 1e-3 and d32.mean() < 7e-4, "product %s vs oracle (fp32 parameters) %s" % (got, want32)
    print("max |product - oracle_bf16| %.2e   max |product - oracle_fp32| %.2e   max |oracle_bf16 - oracle_fp32| %.2e"
          % (d.max(), d32.max(), np.abs(np.array(want) - np.array(want32)).max()))
