"""Where a launch's time goes beyond its workgroups' lives: a -DHALVA_STAMP -DHALVA_STAMP_STRIDE=17 build samples every 17th workgroup of the
2048 (all XCDs, all rounds) with its start (s_memrealtime), life and block index.   HALVA_HIP_LIB=<that build> python tools/stamp_rounds.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from halva_amd import hip, kernels as K
S, T, H, D = int(os.environ.get("S", 8)), 2048, 32, 128
qkv = torch.randn(S, T, 3 * H * D, device="cuda").to(torch.bfloat16).requires_grad_(True)
dout = torch.randn(S, T, H * D, device="cuda").to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device="cuda"); sl = torch.full((S,), T, dtype=torch.int32, device="cuda")
for _ in range(300):
    qkv.grad = None
    out = K.sdpa_causal(qkv, ss, sl, H, D); out.backward(dout)
torch.cuda.synchronize()
lib = hip.load(); lib.halva_dbg_buffer.restype = ctypes.c_void_p
buf = (ctypes.c_uint64 * 8192)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(lib.halva_dbg_buffer()), 8192 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
for region, name in enumerate(("sdpa_fwd", "sdpa_bwd_dkv2", "sdpa_bwd_dq2", "sdpa_bwd_dkv3")):
    w = a[4096 + region * 480: 4096 + region * 480 + 480].reshape(120, 4)
    w = w[w[:, 1] > 0]
    if not len(w): continue
    st = (w[:, 2] - w[:, 2].min()) / 100.0; life = w[:, 1] / 100.0; en = st + life
    print("%s: %d samples, first start -> last end %.1f us, life median %.1f us (sum of 8 lives %.1f)" % (name, len(w), en.max(), np.median(life), 8 * np.median(life)))
    o = np.argsort(st)
    for k in range(0, len(o), 15):
        g = o[k:k + 15]
        print("   starts %6.1f..%6.1f us: life %5.1f..%5.1f (median %5.1f)  blocks %s" % (st[g].min(), st[g].max(), life[g].min(), life[g].max(), np.median(life[g]), " ".join("%d" % b for b in w[g, 3][:6])))
    for x in range(8):
        g = (w[:, 3] % 8) == x
        if g.any(): print("   XCD %d: %2d samples, life median %5.1f, last end %6.1f" % (x, g.sum(), np.median(life[g]), en[g].max()))
