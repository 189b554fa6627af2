#!/usr/bin/env python3
"""SDPA kernels at the two shapes of the 7B step: 16 plain rows of 2048 vs 8 packed rows [668 | 1380 | 1380] (prefix sharing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import kernels as K
H, D, dev = 32, 128, "cuda"
def run(S, T, lens, br_a=None, br_b=None, tag=""):
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(S, T, 3 * H * D, generator=g, device=dev).to(torch.bfloat16)
    dout = torch.randn(S, T, H * D, generator=g, device=dev).to(torch.bfloat16)
    ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.tensor(lens, dtype=torch.int32, device=dev)
    mk = lambda v: None if v is None else torch.tensor(v, dtype=torch.int32, device=dev)
    a, b = mk(br_a), mk(br_b)
    q = qkv.clone().requires_grad_(True)
    out = K.sdpa_causal(q, ss, sl, H, D, a, b); out.backward(dout); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0; n = 10
    for _ in range(n):
        q.grad = None
        e[0].record(); out = K.sdpa_causal(q, ss, sl, H, D, a, b); e[1].record(); out.backward(dout); e[2].record()
        torch.cuda.synchronize(); tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print("%-44s fwd %.3f ms  bwd %.3f ms" % (tag, tf / n, tb / n))
if os.environ.get("BENCH_STEP_SHAPES") == "1":      # the two launches of the 7B step's one-group default: 16 packed pairs, 16 reference rows
    run(16, 3428, [3428] * 16, [668] * 16, [2048] * 16, tag="16 packed rows [668 | 1380 | 1380]")
    run(16, 2048, [2048] * 16, tag="16 plain rows of 2048")
else:
    run(16, 2048, [2048] * 16, tag="16 plain rows of 2048 (8 pairs, two rows each)")
    run(8, 3428, [3428] * 8, [668] * 8, [2048] * 8, tag="8 packed rows [668 | 1380 | 1380]")
    run(8, 3428, [3428] * 8, tag="8 plain causal rows of 3428 (for scale)")
