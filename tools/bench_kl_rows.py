#!/usr/bin/env python3
"""halva_kl_rows at the step's size (16 x 1419 response rows of 32000 bf16 logits, with the gradient): rows held in registers between the two passes
(default) against HALVA_KL_KEEP=0 (the rows read twice).  Algorithmic bytes: 2 rows read + 1 row written = 192 KB per row."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd.hip import call, ptr, stream_ptr
R, V = int(os.environ.get("ROWS", 8192)), 32000
g = torch.Generator(device="cuda").manual_seed(0)
pol = (torch.randn(R, V, generator=g, device="cuda") * 3).to(torch.bfloat16)
ref = (pol.float() + 0.3 * torch.randn(R, V, generator=g, device="cuda")).to(torch.bfloat16)
kl = torch.empty(R, dtype=torch.float32, device="cuda")
dp = torch.empty_like(pol)
for mode in ("1", "0", "1", "0"):
    os.environ["HALVA_KL_KEEP"] = mode
    f = lambda: call("halva_kl_rows", ptr(pol), ptr(ref), 0, V, ptr(None), ptr(kl), ptr(dp), 1.0, R, V, stream_ptr())
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10 * 1e-3
    print("HALVA_KL_KEEP=%s: %7.1f us for %d rows, %.2f TB/s of the 3 algorithmic row passes" % (mode, t * 1e6, R, 3.0 * R * V * 2 / t / 1e12))
