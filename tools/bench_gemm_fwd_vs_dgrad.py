"""Sustained rate of the step's big GEMMs, forward form vs dgrad form, at the bench's row count (the solution table loaded):
forward  y[rows, N]      = xa[rows, K'] Wc[N, K']^T          (torch.mm(xa, Wc.t()))
dgrad    dxa[rows, K']   = dy[rows, N] WcT[K', N]^T          (torch.mm(dy, WcT.t()))            K' = in + G r"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd.gemm_tuning import enable_tuned_gemms
enable_tuned_gemms()
rows = int(os.environ.get("ROWS", 54848))
def sustained(fn, fl, secs=0.6):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    n = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); a.record()
    while time.time() - t0 < secs:
        for _ in range(10): fn()
        n += 10
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    return ms, fl / ms / 1e9
for name, N, Kp in (("qkv", 12288, 4480), ("o", 4096, 4224), ("gate_up", 22016, 4352), ("down", 4096, 11136)):
    xa = torch.randn(rows, Kp, device="cuda").to(torch.bfloat16)
    Wc = torch.randn(N, Kp, device="cuda").to(torch.bfloat16)
    WcT = Wc.t().contiguous()
    dy = torch.randn(rows, N, device="cuda").to(torch.bfloat16)
    y = torch.empty(rows, N, device="cuda", dtype=torch.bfloat16)
    dxa = torch.empty(rows, Kp, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * rows * N * Kp
    f = sustained(lambda: torch.mm(xa, Wc.t(), out=y), fl)
    d = sustained(lambda: torch.mm(dy, WcT.t(), out=dxa), fl)
    d2 = sustained(lambda: torch.mm(dy, Wc, out=dxa), fl)
    print("%-8s forward %.3f ms %5.0f TF/s | dgrad via transposed copy %.3f ms %5.0f TF/s | dgrad via stored weight (NN) %.3f ms %5.0f TF/s" % (name, *f, *d, *d2))
    del xa, Wc, WcT, dy, y, dxa
