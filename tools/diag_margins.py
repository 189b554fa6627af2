"""Diagnostic (GPU): where does the error of the per-phrase log-prob sums come from?  Runs the engine on the reference-generated
fixtures with the lm_head logits of the token-logp path in bf16 (as the reference's --bf16 run produces them) and in fp32."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
from golden_util import load_npz
from model_util import batch_of, build_product_models
from halva_amd import dpa

for fixture in ("dpa_step_d64_init", "dpa_step_d128_init", "dpa_step_d64"):
    z = load_npz(fixture + ".npz")
    for mode in ("bf16", "f32"):
        dpa.LOGITS_F32 = (mode == "f32")
        pol, ref, _ = build_product_models(z)
        flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol))
        dpa.bind_model(flat, pol)
        dpa.set_grad_sink(pol, True)
        eng = dpa.DPAEngine(pol, ref, float(z["alpha"]), 8, 8, share_prefix=False)
        plan = eng.make_plan(batch_of(z))
        c, (lp, pa, na), gp = eng.pair_group_loss(batch_of(z), plan, list(range(plan.B)))
        pa, na = pa.detach().cpu().numpy(), na.detach().cpu().numpy()
        B = plan.B
        m = z["out.batch_labels"] != -100
        lpd = lp.detach().cpu().numpy()
        e_tok = np.abs(np.concatenate([lpd[:B], lpd[B:]])[m] - np.concatenate([z["out.pos_logps"], z["out.neg_logps"]])[m])
        print("%-20s logits %-4s  max|pos_acc err| %.2e  max|neg_acc err| %.2e  max|margin err| %.2e  per-token logp err max %.2e mean %.2e"
              % (fixture, mode, np.abs(pa - z["out.pos_acc"]).max(), np.abs(na - z["out.neg_acc"]).max(),
                 np.abs((na - pa) - (z["out.neg_acc"] - z["out.pos_acc"])).max(), e_tok.max(), e_tok.mean()))
