#!/usr/bin/env python3
"""Merge TunableOp result files into one table (same validators required; first file wins on duplicate shapes).
usage: merge_gemm_tables.py out.csv in1.csv in2.csv ..."""
import sys
out, ins = sys.argv[1], sys.argv[2:]
validators, rows, seen = None, [], set()
for f in ins:
    v, r = [], []
    for line in open(f):
        (v if line.startswith("Validator,") else r).append(line.rstrip("\n"))
    if validators is None:
        validators = v
    assert v == validators, "validators differ: %s" % f
    for line in r:
        key = tuple(line.split(",")[:2])
        if line and key not in seen:
            seen.add(key)
            rows.append(line)
open(out, "w").write("\n".join(validators + rows) + "\n")
print("%d shapes -> %s" % (len(rows), out))
