"""Block anatomy of sdpa_bwd_dkv3 from a -DHALVA_STAMP build (s_memtime per key block and wave, sequence 0, head 0):
HALVA_HIP_LIB=<stamped build> python tools/stamp_dkv3.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from halva_amd import hip, kernels as K
S, T, H, D = 8, int(os.environ.get("T", 2048)), 32, 128
dev = "cuda"
qkv = torch.randn(S, T, 3 * H * D, device=dev).to(torch.bfloat16).requires_grad_(True)
dout = torch.randn(S, T, H * D, device=dev).to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
for _ in range(200):
    qkv.grad = None
    out = K.sdpa_causal(qkv, ss, sl, H, D); out.backward(dout)
torch.cuda.synchronize()
lib = hip.load(); lib.halva_dbg_buffer.restype = ctypes.c_void_p
buf = (ctypes.c_uint64 * 8192)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(lib.halva_dbg_buffer()), 8192 * 8, 2)
nkb = (T + 127) // 128
a = np.frombuffer(buf, dtype=np.uint64)[1024:1024 + nkb * 12].reshape(nkb, 12).astype(np.int64)
for kb in range(nkb):
    r = a[kb]
    if r.max() == 0: continue
    print("kb %2d: entry->barrier %5d | requests issued %5d | setup + tiles landed %5d | asm block(s): %3d plain + %2d masked steps %7d cyc = %5d/step | "
          "around the asm %5d | stores %6d | total %7d"
          % (kb, r[8] - r[0], r[9] - r[8], r[10] - r[9], r[3], r[5], r[2], r[2] // max(1, r[3] + r[5]), r[6] - r[1] - r[2], r[7] - r[6], r[7] - r[0]))
