import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from halva_amd import hip, kernels as K
S, T, H, D = 8, 2048, 32, 128
qkv = torch.randn(S, T, 3 * H * D, device="cuda").to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device="cuda"); sl = torch.full((S,), T, dtype=torch.int32, device="cuda")
for _ in range(3): out = K.sdpa_causal(qkv, ss, sl, H, D)
torch.cuda.synchronize()
lib = hip.load(); lib.halva_dbg_buffer.restype = ctypes.c_void_p
buf = (ctypes.c_uint64 * 4096)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(lib.halva_dbg_buffer()), 4096 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64)[256:256 + 128].reshape(8, 16).astype(np.int64)
t0 = a[:, 0].min()
names = "qk_start qk_end bar1_rel chain_beg chain_end x5 x6 x7 idle2_end bar4_rel".split()
print(" " * 8 + " ".join("%10s" % n for n in names))
for w in range(8): print("wave %d  " % w + " ".join("%10d" % (a[w, k] - t0) for k in range(10)))
