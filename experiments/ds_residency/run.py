#!/usr/bin/env python3
"""experiments/ds_residency (round 5): does sdpa_bwd_dq2 run faster when the dS it reads was written JUST BEFORE and still fits the 256 MiB
Infinity Cache?  VERDICT r04 item 1(a): measure before building an interleaved dkv / dq schedule.

One process = one library variant (HALVA_HIP_LIB = libhalva_hip_dsres_{nt,plain}.so, build_variants.sh).  For every shape the C-ABI backward
(delta -> dkv3 (+dS) -> [eviction memset of HALVA_DS_EVICT_MB] -> dq2) is called REPS times with and without the eviction; the kernel times come
from `rocprofv3 --kernel-trace` around this process (summarize.py cuts the trace by the order printed here).
Small shapes: the dS of ONE call fits the cache (plus q/k/v/dO); the large ones are the step's own launches (dS = 2.1 / 4.1 GB: never resident)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from halva_amd import kernels as K
D, dev = 128, "cuda"
REPS = 6
SHAPES = [      # (label, S, T, H, packed)
    ("S2_T1024_H32", 2, 1024, 32, False),      # dS 67 MB, dq2 128 workgroups
    ("S4_T1024_H32", 4, 1024, 32, False),      # dS 134 MB, 256 workgroups
    ("S1_T2048_H16", 1, 2048, 16, False),      # dS 67 MB, 64 workgroups
    ("S1_T2048_H32", 1, 2048, 32, False),      # dS 134 MB, 128 workgroups
    ("S1_T4096_H8", 1, 4096, 8, False),        # dS 134 MB, 64 long-lived workgroups (16 + 1 row blocks each)
    ("S1_T8192_H2", 1, 8192, 2, False),        # dS 134 MB, 32 workgroups
    ("S32_T1024_H32", 32, 1024, 32, False),    # dS 1.07 GB: the non-resident reference for T = 1024
    ("S16_T2048_H32", 16, 2048, 32, False),    # the step's plain launch
    ("S16_T3428_H32p", 16, 3428, 32, True),    # the step's packed launch
]
order = []
for label, S, T, H, packed in SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(S, T, 3 * H * D, generator=g, device=dev).to(torch.bfloat16)
    dout = torch.randn(S, T, H * D, generator=g, device=dev).to(torch.bfloat16)
    ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
    a = torch.full((S,), 668, dtype=torch.int32, device=dev) if packed else None
    b = torch.full((S,), 2048, dtype=torch.int32, device=dev) if packed else None
    q = qkv.clone().requires_grad_(True)
    out = K.sdpa_causal(q, ss, sl, H, D, a, b)
    for evict in (0, 1024):
        os.environ["HALVA_DS_EVICT_MB"] = str(evict)
        for _ in range(REPS):
            q.grad = None
            out.backward(dout, retain_graph=True)
            torch.cuda.synchronize()
        pairs = S * H * (T * (T + 1) // 2 if not packed else 3972906)
        order.append({"label": label, "evict_mb": evict, "calls": REPS, "S": S, "T": T, "H": H, "visible_pairs": pairs})
    del qkv, dout, q, out
    torch.cuda.empty_cache()
json.dump(order, open(os.environ.get("DSRES_ORDER", "dsres_order.json"), "w"))
print("done", len(order))
