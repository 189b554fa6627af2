"""Achieved HBM rate of the row kernels at the 7B step's shapes (27424 packed rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import kernels as K
rows, d, F = 27424, 4096, 11008
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
x = torch.randn(rows, d, device=dev).to(torch.bfloat16).requires_grad_(True)
w = torch.ones(d, device=dev, dtype=torch.bfloat16)
y = K.rmsnorm(x, w, 1e-5)
g = torch.randn_like(y)
t = timeit(lambda: K.rmsnorm(x.detach(), w, 1e-5)); print("rmsnorm_fwd %.1f us  %.2f TB/s" % (t * 1e6, 2 * rows * d * 2 / t / 1e12))
def bwd():
    x.grad = None
    y.backward(g, retain_graph=True)
t = timeit(bwd); print("rmsnorm_bwd %.1f us  %.2f TB/s (3 streams)" % (t * 1e6, 3 * rows * d * 2 / t / 1e12))
