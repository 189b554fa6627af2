#!/usr/bin/env python3
"""LoRA weight-gradient launches of one 7B decoder layer (halva_wgrad_accumulate: C[M, N] f32 += A^T B over `rows`), at the step's two row counts:
time, achieved HBM rate against the bytes each launch has to stream once (rows x (M + N) x 2 B) and MFMA rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import kernels as K
dev = "cuda"
d, F, r = 4096, 11008, 128
# (name, M = width of the A window, N = width of the B window): dA_cat[G r, K] = (dy B)^T x ; dB_g[N_g, r] = dy_g^T (x A_g^T)
SHAPES = [("dA qkv", 3 * r, d), ("dA o", r, d), ("dA gate_up", 2 * r, d), ("dA down", r, F),
          ("dB q", d, r), ("dB k", d, r), ("dB v", d, r), ("dB o", d, r), ("dB gate", F, r), ("dB up", F, r), ("dB down", d, r)]
for rows in (int(os.environ.get("ROWS_PACKED", 16 * 3428)), 16 * 2048):
    tot_t = tot_b = 0.0
    for name, M, N in SHAPES:
        A = torch.randn(rows, M, device=dev).to(torch.bfloat16)
        B = torch.randn(rows, N, device=dev).to(torch.bfloat16)
        C = torch.zeros(M, N, dtype=torch.float32, device=dev)
        for _ in range(2):
            K.wgrad_accumulate(C, A, B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            K.wgrad_accumulate(C, A, B)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e-3
        byt = rows * (M + N) * 2.0
        tot_t += t; tot_b += byt
        print("rows %6d %-11s M %5d N %5d: %7.1f us  %5.2f TB/s  %6.1f TFLOP/s" % (rows, name, M, N, t * 1e6, byt / t / 1e12, 2.0 * M * N * rows / t / 1e12))
    print("rows %6d layer total %.1f us, %.2f TB/s of the once-streamed bytes" % (rows, tot_t * 1e6, tot_b / tot_t / 1e12))
