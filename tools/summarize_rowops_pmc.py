#!/usr/bin/env python3
"""tools/pmc_rowops.sh passes -> HBM rate per row / loss kernel from the counters (json on stdout).
hbm bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on gfx950 FETCH_SIZE counts half the
bytes of wide (16 B per lane) coalesced reads - which is what these kernels issue (MI355X_MICROARCH.md, HBM).  Duration = the kernel's
average in the --kernel-trace --stats pass of the same command.  `algorithmic` = the bytes the kernel has to move at the benchmark's shape
(tools/bench_rowops.py); counters above it mean re-reads, below it cache (MALL) hits that the fabric counters do not see."""
import csv, glob, json, subprocess, sys
root = sys.argv[1]
rows, d, F, R, V = 27424, 4096, 11008, 8192, 32000
ALG = {"rmsnorm_fwd": 2 * rows * d * 2, "rmsnorm_bwd": 3 * rows * d * 2, "swiglu_fwd": 3 * rows * F * 2, "swiglu_bwd": 5 * rows * F * 2,
       "rope_qk": 2 * 8 * (rows // 8) * 2 * 32 * 128 * 2, "token_logp_fwd": R * V * 2, "token_logp_bwd": 2 * R * V * 2, "kl_rows": 3 * R * V * 2}
dur = {}
for f in glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((k for k in ALG if k in r["Name"]), None)
        if k and "anonymous namespace" in r["Name"]:
            t = dur.setdefault(k, [0, 0.0])
            t[0] += int(r["Calls"]); t[1] += float(r["TotalDurationNs"])
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(root + "/" + c + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            k = next((k for k in ALG if k in r["Kernel_Name"]), None)
            if k is None or r["Counter_Name"] != c: continue
            dd = per.setdefault(k, {})
            dd[r["Dispatch_Id"]] = dd.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        for k, dd in per.items():
            cnt.setdefault(k, {})[c] = sum(dd.values()) / len(dd)
import os
commit = os.environ.get("HALVA_COMMIT")      # (the GPU box receives a snapshot without .git: the caller passes the commit)
if not commit:
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        commit = None
out = {"command": "tools/pmc_rowops.sh: rocprofv3 --kernel-trace --stats | --kernel-trace --pmc FETCH_SIZE | --kernel-trace --pmc WRITE_SIZE "
                  "(separate passes) -- python3 tools/bench_rowops.py",
       "shapes": {"rows": rows, "hidden": d, "ffn": F, "loss_rows": R, "vocab": V, "dtype": "bf16"},
       "units": "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch; TB/s = bytes / average duration", "commit": commit, "kernels": {}}
for k in ALG:
    if k not in dur or k not in cnt:
        continue
    us = dur[k][1] / dur[k][0] / 1e3
    f, w = cnt[k].get("FETCH_SIZE", 0.0), cnt[k].get("WRITE_SIZE", 0.0)
    b = (2 * f + w) * 1024
    out["kernels"][k] = {"calls": dur[k][0], "avg_us": round(us, 1), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                         "hbm_bytes_from_counters": int(b), "hbm_tb_s_from_counters": round(b / us / 1e6, 3),
                         "algorithmic_bytes": ALG[k], "algorithmic_tb_s": round(ALG[k] / us / 1e6, 3),
                         "frac_of_8_tb_s_spec": round(ALG[k] / us / 1e6 / 8.0, 3)}
print(json.dumps(out, indent=1))
