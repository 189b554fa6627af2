"""Parity of the COMPOSED model at 7B width and reduced depth against the oracle (missing 3 / next 2): whole-step parity against the
reference stopped at 2 layers / hidden 256 and the full-width test at ONE layer - nothing compared STACKED full-width layers (the bf16 residual
stream through the depth the reference runs: llava/model/language_model/modelling_llama.py:657-672 under llava/train/halva_trainer.py:534-592).

Here: 8 decoder layers at d 4096 / 32 x 128 heads / F 11008 / vocab 32000, LoRA r = 128 with B != 0, one pair of T = 512 post-splice (224-pixel
images: 256 patches), the product's engine exactly as the bench runs it (tuned GEMM table, prefix sharing, top-row pruning) against
oracle.dpa.compute_loss in fp32 on the host from the same bf16-rounded weights.

What it is held to.  north_star's "loss within 1e-3" is a statement about the reference's fixtures (hidden 256, vocab 160): at the real widths
and a 32 000-entry vocabulary NO bf16 execution of this path has it - the oracle itself, re-run in bf16 on the host, moves the loss by
2e-2 .. 4e-2 on this case (phrase sums of ~-31 nat carry 2^-9 relative noise per rounding; the divergence is a SUM over 200 tokens).  So the
product is judged against that measured floor: tests/golden/fulldepth8_floor.json holds the errors of the oracle's permutation-only bf16
realisations (oracle/realise.py:BOUND_SET; written by tools/fulldepth_parity.py --write-floor on the GPU box's host), one of which is re-measured
here so that the table cannot rot.  Asserted: loss / alignment / divergence / phrase margins / LoRA + projector gradients of layers 0, 3, 7 within
mean + 3 sigma of the realisations' errors (never below the largest), phrase sums within 1e-2 relative, margin signs wherever the margin exceeds
the floor.  The 32-layer one-off of the same code: tools/fulldepth_parity.py -> profiles/r06_fulldepth_parity.log."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import fulldepth_util as U  # noqa: E402

FLOOR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fulldepth8_floor.json")
COLS = ("loss", "alignment", "divergence", "margin", "grad_max")


def _bound(table, col):
    v = np.asarray([r[col] for r in table.values()], dtype=np.float64)
    return float(max(v.max(), v.mean() + 3.0 * v.std(ddof=1)))


def test_eight_full_width_layers_match_the_oracle():
    import bench
    with open(FLOOR) as f:
        floor = json.load(f)
    threads = bench.physical_cores()[0]
    case = U.make_case(**floor["case"])
    probe = (0, 3, 7)
    got = U.run_product(case, layers_probed=probe)
    assert got["packing"]
    want = U.run_oracle(case, torch.float32, "plain", probe, threads)
    # the fp32 oracle of this run is the one the table was measured against
    assert abs(want["loss"] - floor["oracle_fp32"]["loss"]) < 2e-4 * abs(want["loss"]), (want["loss"], floor["oracle_fp32"]["loss"])
    e = U.compare(got, want)
    table = floor["realisations"]
    print("8 x 7B-width layers, T = %d: fp32 oracle loss %.4f = alignment %.4f + %.1f x divergence %.4f (%.0f s), product %.0f s"
          % (case["max_len"], want["loss"], want["alignment"], case["loss_alpha"], want["divergence"], want["seconds"], got["seconds"]))
    print("   product      " + "  ".join("%s %.2e" % (k, e[k]) for k in COLS + ("phrase_rel",)))
    print("   bound        " + "  ".join("%s %.2e" % (k, _bound(table, k)) for k in COLS) + "   (mean + 3 sigma of %d bf16 realisations of the oracle)" % len(table))
    # one realisation re-measured: the committed table is live (another host blocks its GEMMs differently: one more summation order)
    live = U.compare(U.run_oracle(case, torch.bfloat16, "perm1", probe, threads), want)
    print("   perm1 live   " + "  ".join("%s %.2e" % (k, live[k]) for k in COLS) + "   committed " + "  ".join("%.2e" % table["perm1"][k] for k in COLS))
    for k in ("margin", "grad_max", "divergence"):
        assert 0.4 * live[k] <= table["perm1"][k] <= 2.5 * live[k], (k, live[k], table["perm1"][k])
    for k in COLS:
        assert e[k] <= _bound(table, k), (k, e[k], _bound(table, k))
    assert e["phrase_rel"] < 1e-2, e["phrase_rel"]
    m_want = want["neg_acc"] - want["pos_acc"]
    m_got = got["neg_acc"] - got["pos_acc"]
    big = np.abs(m_want) > _bound(table, "margin")
    assert (np.sign(m_got[big]) == np.sign(m_want[big])).all()
