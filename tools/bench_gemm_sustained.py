"""One large step GEMM (fused q/k/v + LoRA-A forward shape) in bursts vs sustained, same operands vs rotating operands:
separates clock / power effects from cache residency when comparing tuning-table times with in-step times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd.gemm_tuning import enable_tuned_gemms
enable_tuned_gemms()
M, N, K = 27424, 12288, 4480
xs = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(4)]
ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(4)]
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fl = 2.0 * M * N * K
def run(n, rotate):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        torch.mm(xs[i % 4 if rotate else 0], ws[i % 4 if rotate else 0].t(), out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
run(3, False)
for rotate in (False, True):
    time.sleep(2.0)
    print("rotate=%s  burst of 5: %.3f ms (%.0f TF/s)" % (rotate, (t := run(5, rotate)), fl / t / 1e9), end="   ")
    t = run(600, rotate)
    print("600 back to back: %.3f ms (%.0f TF/s)" % (t, fl / t / 1e9), end="   ")
    t = run(50, rotate)
    print("next 50: %.3f ms (%.0f TF/s)" % (t, fl / t / 1e9))
