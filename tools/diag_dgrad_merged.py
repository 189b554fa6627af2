import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); [sys.path.insert(0, p) for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden"))]
import torch
import halva_amd.llama as L
from golden_util import load_npz
from model_util import batch_of
import test_dpa_step_gpu as T
for fx in ("dpa_step_d64.npz", "dpa_step_d64_init.npz", "dpa_step_d128_init.npz"):
    z = load_npz(fx)
    out = {}
    for name, (wt, mg) in {"merged": (True, True), "wt": (True, False), "nn": (False, False)}.items():
        L.DGRAD_TRANSPOSED_COPY, L.DGRAD_MERGED = wt, mg
        eng, pol, ref, flat, _ = T._engine(z, 8, 8)
        loss = float(eng.loss(batch_of(z), backward=True))
        out[name] = flat.grad.clone()
    n = out["nn"].norm()
    print(fx, "merged vs nn %.2e   wt vs nn %.2e   merged vs wt %.2e" % (float((out["merged"] - out["nn"]).norm() / n), float((out["wt"] - out["nn"]).norm() / n), float((out["merged"] - out["wt"]).norm() / n)))

# against the reference's own gradients (stress fixture: dense dL/dW of the reference -> LoRA factor gradients by the chain rule)
from golden_util import tensors
z = load_npz("dpa_step_d64.npz")
fac = tensors(z, "lora.")
r, alpha = int(z["lora_cfg"][0]), float(z["lora_cfg"][1])
s = alpha / r
for name, (wt, mg) in {"merged": (True, True), "two-gemm": (True, False)}.items():
    L.DGRAD_TRANSPOSED_COPY, L.DGRAD_MERGED = wt, mg
    eng, pol, ref, flat, _ = T._engine(z, 8, 8)
    eng.loss(batch_of(z), backward=True)
    errs = []
    for i, layer in enumerate(pol.model.layers):
        for sub, grp in layer.groups():
            for g, n in enumerate(grp.names):
                key = "grad.model.layers.%d.%s.%s.weight" % (i, sub, n)
                if key not in z.files:
                    continue
                dW = torch.from_numpy(z[key])
                A = fac["model.layers.%d.%s.%s.A" % (i, sub, n)]
                Bm = fac["model.layers.%d.%s.%s.B" % (i, sub, n)]
                gA = grp.A_cat.main_grad[g * r:(g + 1) * r].cpu()
                gB = getattr(grp, n).lora_B["default"].weight.main_grad.cpu()
                refA, refB = s * Bm.T @ dW, s * dW @ A.T
                errs.append((float((gA - refA).norm() / refA.norm()), float((gB - refB).norm() / refB.norm()), "%d.%s" % (i, n)))
    print(name, "vs reference gradients: max dA err %.3e  max dB err %.3e  mean %.3e / %.3e" % (max(e[0] for e in errs), max(e[1] for e in errs),
          sum(e[0] for e in errs) / len(errs), sum(e[1] for e in errs) / len(errs)))
    print("   per factor (layer.target: dA, dB):", "  ".join("%s: %.1e, %.1e" % (e[2], e[0], e[1]) for e in errs))
