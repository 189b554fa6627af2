"""Per-section s_memtime stamps of the SDPA forward (needs a -DHALVA_STAMP build of libhalva_hip.so).
usage: HALVA_HIP_LIB=<stamped .so> python tools/stamp_fwd.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
buf = (ctypes.c_uint64 * 4096)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(ptr), 4096 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)
names = os.environ.get("STAMP_NAMES", "loads,S0,S1+softmax0,check,PV0+softmax1+PV1,store+barrier").split(",")
for w in range(8):
    r = a[w]; n = int(r[6])
    if n == 0: continue
    print("wave %d tiles %d  " % (w, n) + "  ".join("%s %.0f" % (nm, r[i] / n) for i, nm in enumerate(names)) + "  total/tile %.0f" % (sum(r[:len(names)]) / n))
