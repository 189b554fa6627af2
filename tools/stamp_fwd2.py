"""Per-phase s_memtime stamps of the phase-pipelined SDPA forward (fwd2).  Needs a -DHALVA_STAMP build:
   hipcc -DHALVA_STAMP ... -o halva_amd/libhalva_hip_stamp.so;  HALVA_HIP_LIB=halva_amd/libhalva_hip_stamp.so python tools/stamp_fwd2.py"""
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
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 16)
names = os.environ.get("STAMP_NAMES", "QK,bar,req+preV,bar,PV,vmwait,bar,preK+bar").split(",")
for w in range(8):
    r = a[w]; n = int(r[8])
    if n == 0: continue
    print("wave %d active tiles %d  " % (w, n) + "  ".join("%s %.0f" % (nm, r[i] / n) for i, nm in enumerate(names)) + "  total/tile %.0f" % (sum(r[:8]) / n))
