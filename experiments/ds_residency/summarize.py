#!/usr/bin/env python3
"""summarize.py <kernel_trace.csv> <order.json>: per (shape, eviction) the mean time of sdpa_bwd_dkv3 / sdpa_bwd_dq2 (last 4 of the 6 calls) and dq2's dS rate."""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
dq = [dur(r) for r in rows if "sdpa_bwd_dq2" in r["Kernel_Name"]]
dkv = [dur(r) for r in rows if "sdpa_bwd_dkv3" in r["Kernel_Name"]]
order = json.load(open(sys.argv[2]))
i = 1      # the forward's first backward?  no: run.py calls backward only inside the loops - but the warm-up forward launches none.
i = 0
print("%-16s %6s %10s %10s %12s" % ("shape", "evict", "dkv3 us", "dq2 us", "dq2 dS TB/s"))
for o in order:
    n = o["calls"]
    a, b = dkv[i:i + n][2:], dq[i:i + n][2:]
    i += n
    md, mq = sum(a) / len(a), sum(b) / len(b)
    print("%-16s %6d %10.1f %10.1f %12.2f" % (o["label"], o["evict_mb"], md, mq, o["visible_pairs"] * 2 / mq / 1e6))
assert i == len(dq) == len(dkv), (i, len(dq), len(dkv))
