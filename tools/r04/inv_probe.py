"""prints the grouping / prefix-sharing invariance differences of the full-width tests for both forward kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import test_fullsize_13b_vila_gpu as A
import test_fullsize_properties_gpu as B
for flag in ("0", "1", "0", "1"):
    os.environ["HALVA_SDPA_FWD3"] = flag
    pol, ref, flat, spec = A._llava13b()
    batch = A._batch(pol, spec, 2)
    l1, p1, g1, e1 = A._loss(pol, ref, flat, batch, 2, 2, 0.4, share=False)
    l2, p2, g2, e2 = A._loss(pol, ref, flat, batch, 1, 1, 0.4, share=False)
    l3, p3, g3, e3 = A._loss(pol, ref, flat, batch, 2, 2, 0.4, share="always")
    print("FWD3=%s llava13b: align %.6f | groups of 1: d_align %.2e d_div %.2e d_loss %.2e grad %.2e | packed: d_align %.2e d_div %.2e d_loss %.2e grad %.2e" % (
        flag, p1["alignment"], p2["alignment"] - p1["alignment"], p2["divergence"] - p1["divergence"], l2 - l1, float((g2 - g1).norm() / g1.norm()),
        p3["alignment"] - p1["alignment"], p3["divergence"] - p1["divergence"], l3 - l1, float((g3 - g1).norm() / g1.norm())), flush=True)
    del pol, ref, flat
    torch.cuda.empty_cache()
    pol, ref, flat = B._models()
    batch = B._batch(4)
    l1, p1, g1 = B._loss(pol, ref, flat, batch, 4, 4)
    l2, p2, g2 = B._loss(pol, ref, flat, batch, 1, 1)
    l3, p3, g3 = B._loss(pol, ref, flat, batch, 2, 3)
    print("FWD3=%s 7b width: align %.6f | groups 1/1: d_align %.2e d_div %.2e | groups 2/3: d_align %.2e d_div %.2e" % (
        flag, p1["alignment"], p2["alignment"] - p1["alignment"], p2["divergence"] - p1["divergence"], p3["alignment"] - p1["alignment"], p3["divergence"] - p1["divergence"]), flush=True)
    del pol, ref, flat
    torch.cuda.empty_cache()
