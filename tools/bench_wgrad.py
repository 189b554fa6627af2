"""LoRA weight-gradient products at the 7B step's shapes: library GEMM (torch.mm) vs halva_wgrad_accumulate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import kernels as K
rows = int(os.environ.get("ROWS", 54848))      # the 7B step: 16 pairs as one group (27424: groups of 8)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, (M, lda, N, ldb) in {"dB q/k/v/o (dy_g^T xA)": (4096, 12288, 128, 4480), "dB down": (4096, 4096, 128, 11136), "dB gate/up": (11008, 22016, 128, 4352),
                               "dA qkv (da^T x)": (384, 4480, 4096, 4480), "dA gate_up": (256, 4352, 4096, 4352), "dA o": (128, 4224, 4096, 4224),
                               "dA down": (128, 11136, 11008, 11136)}.items():
    abuf = torch.randn(rows, lda, device="cuda").to(torch.bfloat16)
    bbuf = torch.randn(rows, ldb, device="cuda").to(torch.bfloat16)
    A, B = abuf[:, :M], bbuf[:, ldb - N if "dB" in name else 0:][:, :N]
    C = torch.zeros(M, N, device="cuda")
    t_lib = timeit(lambda: C.add_(torch.mm(A.t(), B), alpha=0.5))
    t_new = timeit(lambda: K.wgrad_accumulate(C, A, B, 0.5))
    mb = (A.numel() + B.numel()) * 2 / 1e6
    print("%-26s M %5d N %5d  library mm+add %7.1f us   split-k kernel %7.1f us  (%.0f MB streamed: %.2f TB/s)" % (name, M, N, t_lib, t_new, mb, mb / t_new))
