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
# SwiGLU on the fused gate|up buffer: forward reads 2F, writes F; backward reads 2F + F, writes 2F
gu = torch.randn(rows, 2 * F, device=dev).to(torch.bfloat16).requires_grad_(True)
h = K.swiglu(gu)
gh = torch.randn_like(h)
t = timeit(lambda: K.swiglu(gu.detach())); print("swiglu_fwd  %.1f us  %.2f TB/s" % (t * 1e6, 3 * rows * F * 2 / t / 1e12))
def sbwd():
    gu.grad = None
    h.backward(gh, retain_graph=True)
t = timeit(sbwd); print("swiglu_bwd  %.1f us  %.2f TB/s (incl. autograd overhead)" % (t * 1e6, 5 * rows * F * 2 / t / 1e12))
# RoPE in place on the q,k thirds of a packed qkv buffer: reads and writes 2/3 of it
S_, T_, H_, D_ = 8, rows // 8, 32, 128
qkv = torch.randn(S_, T_, 3 * H_ * D_, device=dev).to(torch.bfloat16)
cos, sin = K.rope_tables(D_, T_, device=dev)
t = timeit(lambda: K._rope_inplace(qkv, cos, sin, T_, H_, D_, False)); print("rope_qk     %.1f us  %.2f TB/s" % (t * 1e6, 2 * S_ * T_ * 2 * H_ * D_ * 2 / t / 1e12))
