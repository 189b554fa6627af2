#!/usr/bin/env python3
"""GEMM shapes of one 7B decoder layer at M tokens through torch.mm (hipBLASLt): achieved TFLOP/s per shape."""
import os, sys, torch
M = int(os.environ.get("M", 16384))
d, F, r = 4096, 11008, 128
dev = "cuda"
def t(fn, flop, name, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-34s %8.3f ms %8.1f TF/s" % (name, ms, flop / ms / 1e9))
    return ms
g = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
tot = 0
for name, N, K in (("qkv", 3 * d, d), ("o", d, d), ("gate_up", 2 * F, d), ("down", d, F)):
    x, W, dy = g(M, K), g(N, K), g(M, N)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tot += t(lambda: torch.mm(x, W.t(), out=y), 2 * M * N * K, "fwd  %s [%d,%d]x[%d,%d]^T" % (name, M, K, N, K))
    dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    tot += t(lambda: torch.mm(dy, W, out=dx), 2 * M * N * K, "dx   %s" % name)
    A, B = g(r, K), g(N, r)
    a = torch.empty(M, r, device=dev, dtype=torch.bfloat16)
    t(lambda: torch.mm(x, A.t(), out=a), 2 * M * r * K, "lora a=x@A^T %s" % name)
    t(lambda: y.addmm_(a, B.t(), alpha=2.0), 2 * M * r * N, "lora y+=a@B^T %s" % name)
    t(lambda: torch.mm(dy, B), 2 * M * r * N, "lora da=dy@B %s" % name)
    t(lambda: torch.mm(dy.t(), a), 2 * M * r * N, "lora dB=dy^T@a %s" % name)
    t(lambda: torch.mm(a.t(), x), 2 * M * r * K, "lora dA=da^T@x %s" % name)
print("base fwd+dx total per layer: %.3f ms" % tot)
