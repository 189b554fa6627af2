#!/usr/bin/env python3
"""kernel_stats.csv of `rocprofv3 --kernel-trace --stats` -> a grouped markdown summary (used for profiles/)."""
import csv, sys
src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def cat(n):
    for key, lab in (("sdpa_bwd_dkv3", "sdpa_bwd_dkv3 (HIP + generated asm)"), ("sdpa_bwd_dkv", "sdpa_bwd_dkv2 (HIP)"), ("sdpa_bwd_dq3", "sdpa_bwd_dq3 (HIP)"), ("sdpa_bwd_dq", "sdpa_bwd_dq (HIP)"), ("sdpa_fwd3", "sdpa_fwd3 causal D128 (HIP + generated asm)"), ("sdpa_fwd_kernel<128", "sdpa_fwd causal D128 (HIP)"),
                     ("sdpa_fwd_kernel<64", "sdpa_fwd full D64 CLIP (HIP)"), ("sdpa_bwd_delta", "sdpa_bwd_delta (HIP)")):
        if key in n: return lab
    if "anonymous namespace" in n:
        for k in ("swiglu_bwd", "swiglu_fwd", "rmsnorm_bwd", "rmsnorm_fwd", "rope_qk", "splice_rows", "token_logp_fwd", "token_logp_bwd", "kl_rows",
                  "phrase_sum_fwd", "phrase_sum_bwd", "wgrad_dma", "gemm_kernel", "im2col", "gelu_bwd", "colsum", "layernorm_fwd", "layernorm_bwd_params",
                  "downsample2x2", "image_resample_h", "image_resample_v", "splitk_reduce"):
            if k in n: return k + " (HIP)"
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"): return "hipBLASLt GEMMs (PyTorch-ROCm)"
    return "other torch kernels"
agg = {}
for r in rows:
    a = agg.setdefault(cat(r["Name"]), [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
with open(dst, "w") as f:
    f.write("# %s\n\nTotal kernel time %.3f s.\n\n| kernel group | calls | total ms | %% | avg us |\n|---|---|---|---|---|\n" % (title, tot / 1e9))
    for c, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
        f.write("| %s | %d | %.1f | %.2f | %.1f |\n" % (c, n, t / 1e6, 100 * t / tot, t / n / 1e3))
print(open(dst).read())
