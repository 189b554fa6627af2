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
# loss kernels on one lm_head chunk (8192 response rows x 32000 logits, bf16): token_logp reads V*2 B per row (bwd: + writes V*2 B in
# place); kl_rows reads two logit rows, re-reads them from L2/MALL for the gradient and writes one
R_, V_ = 8192, 32000
lg = (torch.randn(R_, V_, device=dev) * 2).to(torch.bfloat16)
lg2 = (torch.randn(R_, V_, device=dev) * 2).to(torch.bfloat16)
tgt = torch.randint(0, V_, (R_,), device=dev, dtype=torch.int32)
from halva_amd.hip import call, ptr, stream_ptr
BF16 = 0
logp = torch.empty(R_, device=dev); lse = torch.empty(R_, device=dev); gg = torch.randn(R_, device=dev); kl = torch.empty(R_, device=dev)
t = timeit(lambda: call("halva_token_logp_fwd", ptr(lg), BF16, V_, ptr(tgt), ptr(logp), ptr(lse), R_, V_, stream_ptr()))
print("token_logp_fwd %.1f us  %.2f TB/s" % (t * 1e6, R_ * V_ * 2 / t / 1e12))
work = lg.clone()
t = timeit(lambda: call("halva_token_logp_bwd", ptr(work), BF16, V_, ptr(tgt), ptr(lse), ptr(gg), ptr(work), R_, V_, stream_ptr()))
print("token_logp_bwd %.1f us  %.2f TB/s (read + write)" % (t * 1e6, 2 * R_ * V_ * 2 / t / 1e12))
dp_ = torch.empty_like(lg)
t = timeit(lambda: call("halva_kl_rows", ptr(lg), ptr(lg2), BF16, V_, None, ptr(kl), ptr(dp_), 1.0, R_, V_, stream_ptr()))
print("kl_rows (+grad) %.1f us  %.2f TB/s (2 reads + 1 write)" % (t * 1e6, 3 * R_ * V_ * 2 / t / 1e12))
