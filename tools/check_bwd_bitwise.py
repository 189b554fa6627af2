#!/usr/bin/env python3
"""Bitwise A/B of the causal attention backward between two builds of the library: run once per build (HALVA_HIP_LIB=<variant .so> for the other one) with the
same output file name stem; the second run compares dq / dk / dv with what the first one saved.   usage: python tools/check_bwd_bitwise.py <stem> [save|compare]
Layouts: plain rows, ragged rows, left-padded rows, packed rows with a branch (prefix sharing: wholly and partly hidden strips), with and without the fused
inverse RoPE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import hip, kernels as K
dev = "cuda"
stem, mode = sys.argv[1], sys.argv[2]
H, D = 8, 128
cases = [("plain 4 x 2048", 4, 2048, [0] * 4, [2048] * 4, None, None),
         ("ragged", 3, 1500, [0] * 3, [1500, 777, 1], None, None),
         ("left padded", 2, 1100, [100, 37], [1000, 1063], None, None),
         ("packed [668 | 1380 | 1380]", 3, 3428, [0] * 3, [3428] * 3, [668] * 3, [2048] * 3),
         ("packed, odd branch points", 3, 1900, [0] * 3, [1900, 1811, 1500], [70, 333, 1000], [900, 1001, 1250])]
out = {}
for name, S, T, starts, lens, bra, brb in cases:
    g = torch.Generator(device=dev).manual_seed(7)
    qkv = torch.randn(S, T, 3 * H * D, generator=g, device=dev).to(torch.bfloat16)
    dout = torch.randn(S, T, H * D, generator=g, device=dev).to(torch.bfloat16)
    mk = lambda v: None if v is None else torch.tensor(v, dtype=torch.int32, device=dev)
    ss, sl, a, b = mk(starts), mk(lens), mk(bra), mk(brb)
    for rope in (False, True):
        q = qkv.clone().requires_grad_(True)
        cos, sin = K.rope_tables(D, 4096, device=dev) if rope else (None, None)
        o = K._SdpaCausal.apply(q, ss, sl, H, D, cos, sin, None, None if a is None else (a, b, None))
        o.backward(dout)
        torch.cuda.synchronize()
        out["%s%s" % (name, " +rope" if rope else "")] = q.grad.detach().cpu()
if mode == "save":
    torch.save(out, stem + ".pt")
    print("saved", len(out), "gradients with", os.environ.get("HALVA_HIP_LIB", "the in-tree library"))
else:
    ref = torch.load(stem + ".pt")
    bad = 0
    for k, v in out.items():
        same = torch.equal(v.view(torch.int16), ref[k].view(torch.int16))
        nz = int((v.view(torch.int16) != ref[k].view(torch.int16)).sum())
        print("%-44s %s%s" % (k, "BIT-IDENTICAL" if same else "DIFFERENT", "" if same else " (%d of %d elements, max |diff| %.3e)" % (nz, v.numel(), float((v.float() - ref[k].float()).abs().max()))))
        bad += not same
    sys.exit(1 if bad else 0)
