#!/usr/bin/env python3
"""CPU: the reference arithmetic (oracle) re-run in bf16 under every realisation of oracle/realise.py, against the reference's own fp32
outputs held by the step fixtures: |loss|, |alignment|, |divergence| error, max |phrase margin| error, max relative gradient error.
The maxima over the set are the constants of tests/test_dpa_step_gpu.py (MARGIN_FLOOR / GRAD_FLOOR / LONG bounds); DESIGN.md 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
from golden_util import load_npz
import test_dpa_step_gpu as G
from oracle import realise

write = "--write" in sys.argv
names = [a for a in sys.argv[1:] if a != "--write"] or list(G.FIXTURES)
table = {}
for name in names:
    z = load_npz(name + ".npz")
    rows = []
    for r in realise.REALISATIONS:
        m = G._bf16_realisation(name, z, r)
        rows.append(m)
        table.setdefault(name, {})[r] = dict(zip(G.REAL_COLS, [float("%.4g" % v) for v in m]))
        print("%-20s %-14s loss %.2e align %.2e div %.2e | pos_acc %.2e neg_acc %.2e margin %.2e | grad %.2e" % (name, r, *m), flush=True)
    a = np.array(rows)
    print("%-20s %-14s loss %.2e align %.2e div %.2e | pos_acc %.2e neg_acc %.2e margin %.2e | grad %.2e" % (name, "MAX", *a.max(0)))
    print("%-20s %-14s loss %.2e align %.2e div %.2e | pos_acc %.2e neg_acc %.2e margin %.2e | grad %.2e" % (name, "MEDIAN", *np.median(a, 0)))
if write:      # tests/golden/bf16_realisations.json: the committed table (tests/test_oracle_vs_golden.py re-measures part of it)
    import json
    path = os.path.join(ROOT, "tests", "golden", "bf16_realisations.json")
    json.dump({"what": "errors of the reference arithmetic (oracle) run in bf16 on the CPU under the realisations of oracle/realise.py, against the "
                       "reference's fp32 outputs of the step fixtures; written by tools/measure_bf16_floors.py --write", "torch": torch.__version__,
               "threads": torch.get_num_threads(), "fixtures": table}, open(path, "w"), indent=1)
    print("wrote", path)
