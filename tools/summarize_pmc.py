#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv files -> per-kernel, per-launch averages of every counter (json on stdout).
usage: summarize_pmc.py <dir with */*counter_collection.csv> [kernel-name substring ...]"""
import csv, glob, json, sys
root, keys = sys.argv[1], sys.argv[2:] or ["sdpa_fwd", "sdpa_bwd_dq", "sdpa_bwd_dkv", "sdpa_bwd_delta"]
acc = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        k = next((k for k in keys if k in r["Kernel_Name"]), None)
        if k is None: continue
        d = per.setdefault((k, r["Counter_Name"]), {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for (k, c), d in per.items():
        acc.setdefault(k, {})[c] = sum(d.values()) / len(d)
for k, c in acc.items():
    der = {}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        der["cycles_per_launch_per_xcd"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c: der["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
        if "SQ_ACTIVE_INST_LDS" in c: der["lds_inst_active_frac(quad-cycles/CU)"] = c["SQ_ACTIVE_INST_LDS"] * 4 / 256 / cyc
        if "SQ_LDS_BANK_CONFLICT" in c: der["lds_bank_conflict_cycles_per_cu_frac"] = c["SQ_LDS_BANK_CONFLICT"] / 256 / cyc
        if "SQ_LDS_IDX_ACTIVE" in c: der["lds_idx_active_frac"] = c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        der["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in c:      # read requests that leave the L2 for the fabric: 64-byte ones unless counted as 32-byte
        der["l2_to_fabric_read_bytes"] = 64.0 * (c["TCC_EA0_RDREQ_sum"] - c.get("TCC_EA0_RDREQ_32B_sum", 0.0)) + 32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    acc[k] = {"derived": der, "raw_per_launch": c}
print(json.dumps(acc, indent=1))
