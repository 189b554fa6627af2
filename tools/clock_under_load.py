#!/usr/bin/env python3
"""Shader clock the chip holds under each kind of kernel of the step, measured IN a kernel (halva_clock_probe: shader cycles per 100 MHz
tick over a 20 us spin, on a side stream) while that kernel runs back to back for ~0.6 s on random data: the library GEMMs of the step
(forward q/k/v and gate/up shapes at the one-group row count), the SDPA forward / backward, SwiGLU, RMSNorm.  Answers "is the GEMM-bound
73 % of the step running at a reduced clock, and by how much" with numbers (MI355X_MICROARCH.md, DVFS give-back items 6 and 7)."""
import json, os, statistics, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import hip, kernels as K
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip.load()

def under_load(name, fn, flop=None, seconds=0.6):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tr = bench.ClockTrace(dev, None)
    tr.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10): fn()
        n += 10
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    s = tr.stop()
    ms = e0.elapsed_time(e1) / n
    rec = {"kernel": name, "ms_per_call": round(ms, 4), "shader_mhz_median": s["shader_mhz_median"], "shader_mhz_p10": s["shader_mhz_p10"],
           "shader_mhz_p90": s["shader_mhz_p90"], "samples": s["samples"], "board_power_w_median": s["board_power_w_median"]}
    if flop:
        rec["tflops"] = round(flop / ms / 1e9, 1)
        rec["frac_of_2500"] = round(flop / ms / 1e9 / 2500, 3)
        if s["shader_mhz_median"]:
            rec["frac_of_clock_adjusted_peak"] = round(flop / ms / 1e9 / (2500 * s["shader_mhz_median"] / 2400.0), 3)
    print(json.dumps(rec), flush=True)
    return rec

out = []
out.append(under_load("idle (probe only)", lambda: None, seconds=0.3))
rows = 54848
for nm, N, Kd in (("library GEMM fwd q/k/v [54848 x 4096] x [12288 x 4096]^T", 12288, 4096),
                  ("library GEMM fwd gate/up [54848 x 4096] x [22016 x 4096]^T", 22016, 4096),
                  ("library GEMM fwd down [54848 x 11008] x [4096 x 11008]^T", 4096, 11008)):
    A = torch.randn(rows, Kd, device=dev).to(torch.bfloat16)
    W = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    C = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)
    out.append(under_load(nm, lambda: torch.mm(A, W.t(), out=C), flop=2.0 * rows * N * Kd))
    del A, W, C
S, T, H, D = 8, 2048, 32, 128
qkv = torch.randn(S, T, 3 * H * D, device=dev).to(torch.bfloat16)
dout = torch.randn(S, T, H * D, device=dev).to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
fl = 2.0 * T * T * D * H * S
out.append(under_load("sdpa_causal_fwd 8x2048x32x128", lambda: K.sdpa_causal(qkv, ss, sl, H, D), flop=fl))
q = qkv.clone().requires_grad_(True)
o = K.sdpa_causal(q, ss, sl, H, D)
def bwd():
    q.grad = None
    o.backward(dout, retain_graph=True)
out.append(under_load("sdpa_causal_bwd 8x2048x32x128 (delta + dK/dV + dQ)", bwd, flop=2.5 * fl))
gu = torch.randn(27424, 2 * 11008, device=dev).to(torch.bfloat16)
out.append(under_load("swiglu_fwd 27424 x 11008", lambda: K.swiglu(gu)))
x = torch.randn(27424, 4096, device=dev).to(torch.bfloat16); w = torch.ones(4096, device=dev, dtype=torch.bfloat16)
out.append(under_load("rmsnorm_fwd 27424 x 4096", lambda: K.rmsnorm(x, w, 1e-5)))
json.dump({"tool": "tools/clock_under_load.py", "commit": os.environ.get("HALVA_COMMIT"), "peak_assumed_tflops_at_2400mhz": 2500, "runs": out},
          open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
