#!/bin/bash
# where is the GPU idle inside a bench step?  rocprofv3 --kernel-trace of `bench.py --steps 2 --warmup 2`, then the gaps between consecutive kernels of the LAST step
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gaps
rocprofv3 --kernel-trace -d /tmp/prof_gaps -o g --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/r06_gaps_bench.log 2>&1
f=$(find /tmp/prof_gaps -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $O/r06_step_gaps.log <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last step: from the end of the last-but-one AdamW burst (torch's multi_tensor_apply kernels) to the end of the last one
adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2]]
bursts, prev = [], None
for i in adam:
    if prev is None or rows[i][0] - rows[prev][1] > 50e6: bursts.append([i, i])
    else: bursts[-1][1] = i
    prev = i
assert len(bursts) >= 2, bursts
last = rows[bursts[-2][1] + 1: bursts[-1][1] + 1]
busy = sum(e - s for s, e, _ in last)
span = last[-1][1] - last[0][0]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(last, last[1:]):
    if s1 > e0: gaps.append((s1 - e0, n0[:60], n1[:60]))
print("window: %.1f ms, kernels %d, busy %.1f ms (%.3f), idle %.1f ms in %d gaps" % (span / 1e6, len(last), busy / 1e6, busy / span, (span - busy) / 1e6, len(gaps)))
import collections
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else "<1ms" if g < 1e6 else ">=1ms"] += g
print("idle by gap size (ms):", {k: round(v / 1e6, 2) for k, v in hist.items()})
print("largest gaps:")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f us  after %-60s before %s" % (g / 1e3, a, b))
by = collections.Counter()
for g, a, b in gaps: by[a] += g
print("idle by PRECEDING kernel (ms):")
for a, v in by.most_common(15): print("  %7.2f  %s" % (v / 1e6, a))
PY
cat $O/r06_step_gaps.log
