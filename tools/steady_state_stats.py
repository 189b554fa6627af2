#!/usr/bin/env python3
"""Kernel statistics of ONE steady-state bench step: the difference of two `rocprofv3 --kernel-trace --stats` runs of the same command
that differ only in --steps (A: --steps n_a, B: --steps n_b > n_a; same --warmup).  Model construction, weight initialisation, the
warm-up step and anything else outside the timed loop appear identically in both and cancel; what is left is (n_b - n_a) timed steps.
usage: steady_state_stats.py A_kernel_stats.csv B_kernel_stats.csv n_a n_b out.csv out.md "title" """
import csv, sys
a_csv, b_csv, n_a, n_b, out_csv, out_md, title = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], sys.argv[7]
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
A, B = load(a_csv), load(b_csv)
n = n_b - n_a
rows = []
for name, (cb, tb) in B.items():
    ca, ta = A.get(name, (0, 0.0))
    dc, dt = cb - ca, tb - ta
    if dc <= 0 or dt <= 0:
        continue
    rows.append((name, dc / n, dt / n))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
with open(out_csv, "w") as f:
    f.write("Name,CallsPerStep,TotalNsPerStep,AverageNs,Percentage\n")
    for name, c, t in rows:
        f.write('"%s",%.2f,%.0f,%.0f,%.3f\n' % (name, c, t, t / c, 100 * t / tot))
def cat(n):
    for key, lab in (("sdpa_bwd_dkv3", "sdpa_bwd_dkv3 (HIP + generated asm)"), ("sdpa_bwd_dkv", "sdpa_bwd_dkv2 (HIP)"), ("sdpa_bwd_dq3", "sdpa_bwd_dq3 (HIP)"), ("sdpa_bwd_dq", "sdpa_bwd_dq2 (HIP)"), ("sdpa_fwd3", "sdpa_fwd3 causal D128 (HIP + generated asm)"), ("sdpa_fwd_kernel<128", "sdpa_fwd causal D128 (HIP)"),
                     ("sdpa_fwd_kernel<64", "sdpa_fwd full D64 CLIP (HIP)"), ("sdpa_bwd_delta", "sdpa_bwd_delta (HIP)")):
        if key in n: return lab
    if "anonymous namespace" in n:
        for k in ("swiglu_bwd", "swiglu_fwd", "rmsnorm_bwd", "rmsnorm_fwd", "rope_qk", "splice_rows", "token_logp_fwd", "token_logp_bwd", "kl_rows",
                  "phrase_sum_fwd", "phrase_sum_bwd", "wgrad_dma", "gemm_kernel", "im2col", "gelu_bwd", "colsum", "splitk_reduce", "clock_probe"):
            if k in n: return k + " (HIP)"
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"): return "hipBLASLt / rocBLAS GEMMs (PyTorch-ROCm)"
    for key, lab in (("multi_tensor_apply", "AdamW (torch foreach kernels)"), ("elementwise_kernel", "torch elementwise (adds, casts, copies, fills)"),
                     ("reduce_kernel", "torch reductions"), ("index", "torch gather / index"), ("CatArray", "torch cat"), ("copyBuffer", "hip copyBuffer")):
        if key in n: return lab
    return "other torch kernels"
agg = {}
for name, c, t in rows:
    a = agg.setdefault(cat(name), [0.0, 0.0]); a[0] += c; a[1] += t
with open(out_md, "w") as f:
    f.write("# %s\n\nOne steady-state step = (run with --steps %d) - (run with --steps %d), per kernel, divided by %d; kernel time %.1f ms per step.\n\n"
            "| kernel group | launches / step | ms / step | %% | avg us |\n|---|---|---|---|---|\n" % (title, n_b, n_a, n, tot / 1e6))
    for c, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
        f.write("| %s | %.0f | %.2f | %.2f | %.1f |\n" % (c, k, t / 1e6, 100 * t / tot, t / k / 1e3))
print(open(out_md).read())
