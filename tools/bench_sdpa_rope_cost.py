#!/usr/bin/env python3
"""What the inverse RoPE inside the attention backward's store epilogues costs a call (round 5): the backward at the 7B step's two launch shapes with and
without the cos / sin tables (halva_sdpa_branch_bwd_rope with / without rotation), alternating, HIP events; and with the rotation as its own launch
(HALVA_ROPE_FUSED_BWD=0 in a second process).  usage: python3 tools/bench_sdpa_rope_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import kernels as K
H, D, dev = 32, 128, "cuda"
def run(S, T, br_a=None, br_b=None, tag=""):
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(S, T, 3 * H * D, generator=g, device=dev).to(torch.bfloat16)
    dout = torch.randn(S, T, H * D, generator=g, device=dev).to(torch.bfloat16)
    ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
    cos, sin = K.rope_tables(D, 4096, device=dev)
    branch = None
    if br_a is not None:
        pos = torch.cat([torch.arange(br_b), br_a + torch.arange(T - br_b)]).to(torch.int32).repeat(S).to(dev)
        branch = (torch.full((S,), br_a, dtype=torch.int32, device=dev), torch.full((S,), br_b, dtype=torch.int32, device=dev), pos)
    q = qkv.clone().requires_grad_(True)
    outs = {k: K._SdpaCausal.apply(q, ss, sl, H, D, c, s_, None, branch) for k, (c, s_) in (("plain", (None, None)), ("rope", (cos, sin)))}
    for k in outs: outs[k].backward(dout, retain_graph=True)
    torch.cuda.synchronize()
    t = {"plain": 0.0, "rope": 0.0}; n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(n):
        for k in ("plain", "rope"):
            q.grad = None
            e0.record(); outs[k].backward(dout, retain_graph=True); e1.record(); torch.cuda.synchronize()
            t[k] += e0.elapsed_time(e1)
    print("%-40s backward without rotation %.3f ms, with %.3f ms  (+%.3f ms, fused=%s)" % (tag, t["plain"] / n, t["rope"] / n, (t["rope"] - t["plain"]) / n, os.environ.get("HALVA_ROPE_FUSED_BWD", "1")))
run(16, 3428, 668, 2048, tag="16 packed rows [668 | 1380 | 1380]")
run(16, 2048, tag="16 plain rows of 2048")
run(8, 2048, tag="8 plain rows of 2048 (bench.py's micro shape)")
