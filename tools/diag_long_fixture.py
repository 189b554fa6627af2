#!/usr/bin/env python3
"""Diagnostic (GPU, round 5): which rounding carries the product's error on the multi-block step fixture `dpa_step_d128_long`?
VERDICT r04 item 4: the product's phrase margins are 1.0e-2 .. 1.2e-2 from the reference's fp32 numbers, one CPU bf16 realisation of the
reference arithmetic 5.6e-3.  (tools/measure_bf16_floors.py: twelve CPU realisations spread 0.57e-2 .. 1.8e-2, median 1.47e-2 - the single draw
was a lucky one.)  Here the product runs the fixture with ONE rounding switched at a time, each in a child process (the switches are read at
import), and prints signed errors: loss / alignment / divergence, the max and the MEAN signed margin error (a systematic bias shows in the mean).
  python3 tools/diag_long_fixture.py [fixture ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys
ROOT = %r
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
from golden_util import load_npz
from model_util import batch_of, build_product_models
from halva_amd import dpa, kernels as K
fixture, share, fp32_attn = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
if fp32_attn:      # attention in fp32 torch on the GPU (softmax and P V in fp32, ONE rounding of the output), plain rows only
    import math
    def attention(qkv, cos, sin, seq_start, seq_len, H, D, out_width=None, branch=None):
        assert branch is None
        S, T = qkv.shape[:2]
        q, k, v = (qkv.view(S, T, 3, H, D)[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))
        c = torch.cat([cos, cos], -1)[:T].float()[None, None]; s = torch.cat([sin, sin], -1)[:T].float()[None, None]
        rot = lambda x: torch.cat([-x[..., D // 2:], x[..., :D // 2]], -1)
        qb, kb = (q * c + rot(q) * s).to(torch.bfloat16).float(), (k * c + rot(k) * s).to(torch.bfloat16).float()
        out = torch.zeros(S, T, out_width or H * D, dtype=torch.bfloat16, device=qkv.device)
        for i in range(S):
            n = int(seq_len[i]); a = int(seq_start[i])
            att = qb[i, :, a:a + n] @ kb[i, :, a:a + n].transpose(1, 2) / math.sqrt(D)
            att = att + torch.full((n, n), float("-inf"), device=att.device).triu(1)
            o = att.softmax(-1) @ v[i, :, a:a + n]
            out[i, a:a + n, :H * D] = o.permute(1, 0, 2).reshape(n, H * D).to(torch.bfloat16)
        return out
    K.attention = attention
z = load_npz(fixture + ".npz")
pol, ref, _ = build_product_models(z)
flat = dpa.FlatTrainables(dpa.trainable_named_parameters(pol)); dpa.bind_model(flat, pol); dpa.set_grad_sink(pol, True)
eng = dpa.DPAEngine(pol, ref, float(z["alpha"]), 8, 8, share_prefix=(share if share != "False" else False))
margins = []
inner = eng.pair_group_loss
def spy(b, plan, idx):
    out = inner(b, plan, idx)
    margins.append((list(idx), out[1][1].detach().float().cpu().numpy(), out[1][2].detach().float().cpu().numpy()))
    return out
eng.pair_group_loss = spy
with torch.no_grad():
    loss = float(eng.loss(batch_of(z), backward=False))
pa = np.zeros_like(z["out.pos_acc"]); na = np.zeros_like(z["out.neg_acc"])
for idx, a, b in margins:
    pa[idx], na[idx] = a, b
me = (na - pa) - (z["out.neg_acc"] - z["out.pos_acc"])
live = z["out.pos_acc"] != 0
parts = {k: float(v) for k, v in eng.last_parts.items()}
print("RESULT loss %%+.2e align %%+.2e div %%+.2e | margin max %%.2e mean %%+.2e rms %%.2e | pos_acc max %%.2e neg_acc max %%.2e" %% (
    loss - float(z["out.loss"]), parts["alignment"] - float(z["out.alignment"]), parts["divergence"] - float(z["out.divergence"]),
    np.abs(me).max(), me[live].mean(), np.sqrt((me[live] ** 2).mean()), np.abs(pa - z["out.pos_acc"]).max(), np.abs(na - z["out.neg_acc"]).max()))
""" % ROOT
CONFIGS = [("shipped", {}, "0"),
           ("fp32 logits", {"HALVA_LOGITS_F32": "1"}, "0"),
           ("forward kernel of rounds 1-3 (running max)", {"HALVA_SDPA_FWD3": "0"}, "0"),
           ("attention in fp32 torch", {}, "1"),
           ("RMSNorm with the module's two roundings", {"HALVA_RMSNORM_MODULE_ROUNDING": "1"}, "0"),
           ("LoRA as two GEMMs (peft's roundings)", {"HALVA_LORA_TWO_GEMM": "1"}, "0"),
           ("residual adds out of place", {"HALVA_RES_INPLACE": "0"}, "0"),
           ("library-default GEMM kernels", {"HALVA_GEMM_TABLE": "0"}, "0"),
           ("top layer on every row", {"HALVA_TOP_ROWS": "0"}, "0")]
for fixture in sys.argv[1:] or ["dpa_step_d128_long"]:
    for share in ("False", "always"):
        for name, env, fa in CONFIGS:
            if fa == "1" and share != "False":
                continue
            r = subprocess.run([sys.executable, "-c", CHILD, fixture, share, fa], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print("%-20s share=%-6s %-44s %s" % (fixture, share, name, line[-1][7:] if line else "FAILED: " + r.stderr[-400:]), flush=True)
