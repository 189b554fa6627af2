import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from halva_amd import hip, kernels as K
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); hip.load()
S, T, H, D = 8, 2048, 32, 128
qkv = torch.randn(S, T, 3 * H * D, device=dev).to(torch.bfloat16)
dout = torch.randn(S, T, H * D, device=dev).to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
q = qkv.clone().requires_grad_(True)
o = K.sdpa_causal(q, ss, sl, H, D)
def bwd():
    q.grad = None
    o.backward(dout, retain_graph=True)
for _ in range(5): bwd()
torch.cuda.synchronize()
tr = bench.ClockTrace(dev, None); tr.start()
t0 = time.perf_counter(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < 1.5:
    for _ in range(10): bwd()
    n += 10; torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
s = tr.stop()
print(os.environ.get("HALVA_SDPA_DKV3", "1"), "bwd ms", round(e0.elapsed_time(e1) / n, 4), "clock median/p10/p90", s["shader_mhz_median"], s["shader_mhz_p10"], s["shader_mhz_p90"], "samples", s["samples"])
