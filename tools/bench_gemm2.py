#!/usr/bin/env python3
"""Base GEMMs of one 7B decoder layer (K-concatenated LoRA widths) at M tokens: fwd (NT), dgrad as stored (NN) and dgrad
with a pre-transposed weight copy (NT); hipBLASLt vs rocBLAS."""
import os, sys, torch
M = int(os.environ.get("M", 32768))
dev = "cuda"
def t(fn, flop, name, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-40s %8.3f ms %8.1f TF/s" % (name, ms, flop / ms / 1e9), flush=True)
    return ms
g = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
print("blas:", torch.backends.cuda.preferred_blas_library())
tot = {"fwd": 0, "dx_nn": 0, "dx_nt": 0}
for name, N, K in (("qkv", 12288, 4096 + 384), ("o", 4096, 4096 + 128), ("gate_up", 22016, 4096 + 256), ("down", 4096, 11008 + 128)):
    x, W, dy = g(M, K), g(N, K), g(M, N)
    Wt = W.t().contiguous()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    fl = 2 * M * N * K
    tot["fwd"] += t(lambda: torch.mm(x, W.t(), out=y), fl, "fwd    %s N=%d K=%d" % (name, N, K))
    tot["dx_nn"] += t(lambda: torch.mm(dy, W, out=dx), fl, "dx NN  %s" % name)
    tot["dx_nt"] += t(lambda: torch.mm(dy, Wt.t(), out=dx), fl, "dx NT  %s (W^T copy)" % name)
print({k: round(v, 3) for k, v in tot.items()})
