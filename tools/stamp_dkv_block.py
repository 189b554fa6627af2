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
a = np.frombuffer(buf, dtype=np.uint64)[1024:1024 + 16 * 8 * 8].reshape(16, 8, 8).astype(np.int64)
for kb in range(16):
    if a[kb].max() == 0: continue
    t0 = a[kb, :, 0].min()
    print("kb %2d (steps %2d):" % (kb, 32 - 2 * kb), "  ".join("w%d in %5d [scalars+issue %5d, stationary back %5d, tile0 requested %5d] loop %6d..%6d pre_store %6d out %6d" % (w, a[kb, w, 0] - t0, a[kb, w, 5] - t0, a[kb, w, 6] - t0, a[kb, w, 7] - t0, a[kb, w, 4] - t0, a[kb, w, 1] - t0, a[kb, w, 2] - t0, a[kb, w, 3] - t0) for w in (0, 4)))
