import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from halva_amd import kernels as K
S, T, H, D = 8, 2048, 32, 128
qkv = torch.randn(S, T, 3 * H * D, device="cuda").to(torch.bfloat16)
for _ in range(3): K.sdpa_full(qkv, H, D)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): K.sdpa_full(qkv, H, D)
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20 * 1e-3
print("non-causal fwd %.3f ms  %.0f TF/s (4*T*T*D*H*S flop)" % (t * 1e3, 4.0 * T * T * D * H * S / t / 1e12))
