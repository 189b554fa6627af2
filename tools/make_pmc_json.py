#!/usr/bin/env python3
"""traffic.json + traffic_step.json of tools/refresh_profiles.sh -> the HBM-traffic record bench.py quotes (profiles/rNN_sdpa_pmc.json).
Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import json, sys
out = sys.argv[1]
def kernels(path):
    j = json.load(open(path))
    res = {}
    for k, v in j.items():
        c = v["raw_per_launch"]
        f, w = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        res[k] = {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    return res
micro, step = kernels(out + "/traffic.json"), kernels(out + "/traffic_step.json")
bwd = lambda d: sum(v["hbm_bytes_per_launch"] for k, v in d.items() if k.startswith("sdpa_bwd"))
import os, time
print(json.dumps({
    "commit": os.environ.get("HALVA_COMMIT", "unrecorded"), "collected": time.strftime("%Y-%m-%d"),
    "command": "tools/refresh_profiles.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 tools/bench_sdpa.py",
    "shape": {"S": 8, "T": 2048, "H": 32, "D": 128},
    "note": "backward = delta + dK/dV (stores dS) + dQ = dS K launches (one C-ABI call); per-launch averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
    "kernels": micro,
    "sdpa_causal_bwd_hbm_bytes_per_launch": bwd(micro),
    "sdpa_causal_fwd_hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] for k, v in micro.items() if k.startswith("sdpa_fwd")),
    "algorithmic_bytes": {"fwd": "qkv 402.7 MB read + out 134.2 MB + lse 2.1 MB written = 539 MB",
                          "bwd": "qkv 402.7 + out 134.2 + dout 134.2 MB read, dqkv 402.7 MB written = 1074 MB, plus dS (visible (query, key) pairs x 2 B, "
                                 "whole 64 x 128 steps: ~1.2 GB) written once and read once = ~3.4 GB"},
    "in_step": {"command": "the same two passes over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline`",
                "kernels": step, "sdpa_causal_bwd_hbm_bytes_per_launch": bwd(step)}}, indent=1))
