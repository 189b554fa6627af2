import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from halva_amd import hip, kernels as K
S, T, H, D = 8, 2048, 32, 128
dev = "cuda"
qkv = torch.randn(S, T, 3 * H * D, device=dev).to(torch.bfloat16).requires_grad_(True)
dout = torch.randn(S, T, H * D, device=dev).to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
for _ in range(2):
    qkv.grad = None
    out = K.sdpa_causal(qkv, ss, sl, H, D); out.backward(dout)
torch.cuda.synchronize()
lib = hip.load(); lib.halva_dbg_buffer.restype = ctypes.c_void_p
buf = (ctypes.c_uint64 * 4096)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(lib.halva_dbg_buffer()), 4096 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)
# two-role kernel: waves 0-3 = V side, 4-7 = K side
names = ["tile requests", "block 1", "block 2", "blocks 3-4", "stats + dma wait", "barrier"]
NW = 8
for w in range(NW):
    r = a[w]; nt = int(r[6])
    if nt: print("wave %d tiles(64 rows) %d  " % (w, nt) + "  ".join("%s %.0f" % (n, r[i] / nt) for i, n in enumerate(names)) + "  total/tile %.0f" % (sum(r[:6]) / nt))
