"""Whole-row-block anatomy of the SDPA forward (-DHALVA_STAMP build): when, relative to the workgroup's first stamp, each 256-row block's
tile loop starts and ends and its rows are stored.  usage: HALVA_HIP_LIB=<stamped .so> python tools/stamp_fwd_block.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from halva_amd import hip, kernels as K
S, T, H, D = 8, 2048, 32, 128
qkv = torch.randn(S, T, 3 * H * D, device="cuda").to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device="cuda"); sl = torch.full((S,), T, dtype=torch.int32, device="cuda")
for _ in range(3): out = K.sdpa_causal(qkv, ss, sl, H, D)
torch.cuda.synchronize()
lib = hip.load(); lib.halva_dbg_buffer.restype = ctypes.c_void_p
buf = (ctypes.c_uint64 * 4096)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(lib.halva_dbg_buffer()), 4096 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64)[2048:2048 + 8 * 8 * 4].reshape(8, 8, 4).astype(np.int64)
for qb in range(8):
    if a[qb].max() == 0: continue
    t0 = a[qb, :, 0].min()
    print("row block %d (%2d tiles):" % (qb, 4 * qb + 4), "  ".join("w%d entry %5d loop %6d..%6d stored %6d" % (w, *(a[qb, w] - t0)) for w in (0, 7)))
