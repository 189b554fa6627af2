import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halva_amd import hip, kernels as K
S, T, H, D = 8, 2048, 32, 128
dev = "cuda"
qkv = torch.randn(S, T, 3 * H * D, device=dev).to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
for _ in range(3): out = K.sdpa_causal(qkv, ss, sl, H, D)
torch.cuda.synchronize()
lib = hip.load(); lib.halva_dbg_buffer.restype = ctypes.c_void_p
ptr = lib.halva_dbg_buffer()
import numpy as np
buf = (ctypes.c_uint64 * 4096)()
hipmemcpy = ctypes.CDLL("libamdhip64.so").hipMemcpy
hipmemcpy(buf, ctypes.c_void_p(ptr), 4096 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)
names = ["top+loads", "S mfma", "softmax", "PV mfma", "store", "barrier"]
nw = 8 if os.environ.get("HALVA_SDPA_NW") == "8" else 4
for w in range(nw):
    r = a[w]; nt = int(r[6])
    if nt == 0: continue
    print("wave %d tiles %d  " % (w, nt) + "  ".join("%s %.0f" % (n, r[i] / nt) for i, n in enumerate(names)) + "  total/tile %.0f" % (sum(r[:6]) / nt))
print("raw", a[:2])
