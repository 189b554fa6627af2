#!/usr/bin/env python3
"""Micro-benchmark of the causal SDPA kernels at the 7B per-layer shape (S=8, T=2048, H=32, D=128) + a quick check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import kernels as K

def main():
    S, T, H, D = int(os.environ.get("S", 8)), int(os.environ.get("T", 2048)), 32, 128
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(S, T, 3 * H * D, generator=g, device=dev).to(torch.bfloat16)
    dout = torch.randn(S, T, H * D, generator=g, device=dev).to(torch.bfloat16)
    ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
    q = qkv.clone().requires_grad_(True)
    out = K.sdpa_causal(q, ss, sl, H, D); out.backward(dout); torch.cuda.synchronize()
    # correctness on one (seq, head) slice against fp32 torch
    s, h = S - 1, 5
    x = qkv[s].view(T, 3, H, D)[:, :, h].float()
    qq, kk, vv = x[:, 0].clone().requires_grad_(True), x[:, 1].clone().requires_grad_(True), x[:, 2].clone().requires_grad_(True)
    att = (qq @ kk.T) / D ** 0.5 + torch.full((T, T), float("-inf"), device=dev).triu(1)
    ref = torch.softmax(att, -1) @ vv
    ref.backward(dout[s].view(T, H, D)[:, h].float())
    o = out[s].view(T, H, D)[:, h].float()
    gq = q.grad[s].view(T, 3, H, D)[:, :, h].float()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print("check: out %.2e dq %.2e dk %.2e dv %.2e" % (rel(o, ref), rel(gq[:, 0], qq.grad), rel(gq[:, 1], kk.grad), rel(gq[:, 2], vv.grad)))
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0; n = 20
    for _ in range(n):
        q.grad = None
        e[0].record(); out = K.sdpa_causal(q, ss, sl, H, D); e[1].record(); out.backward(dout); e[2].record()
        torch.cuda.synchronize(); tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    tf, tb = tf / n * 1e-3, tb / n * 1e-3
    flop = 2.0 * T * T * D * H * S
    print("fwd %.3f ms %.1f TF/s | bwd %.3f ms %.1f TF/s (algorithmic)" % (tf * 1e3, flop / tf / 1e12, tb * 1e3, 2.5 * flop / tb / 1e12))

if __name__ == "__main__":
    main()
