"""sdpa_bwd_dkv3 as persistent workgroups (-DHALVA_STAMP -DHALVA_STAMP_STRIDE=1 build): per workgroup its life, the cycles inside key blocks, the
number of items it drew.   HALVA_HIP_LIB=<that build> python tools/stamp_persistent.py"""
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
w = a[4096 + 3 * 480: 4096 + 3 * 480 + 480].reshape(120, 4); w = w[w[:, 1] > 0]
st = (w[:, 2] - w[:, 2].min()) / 100.0; life = w[:, 1] / 100.0; en = st + life
print("workgroups sampled %d: start %.1f..%.1f us, end %.1f..%.1f us, clock %.0f MHz" % (len(w), st.min(), st.max(), en.min(), en.max(), np.median(100.0 * w[:, 0] / w[:, 1])))
pb = a[6144:6144 + 480].reshape(120, 4); pb = pb[pb[:, 3] > 0]
if not len(pb): sys.exit(0)
n = pb[:, 3].sum()
print("workgroups 0..%d: items %d..%d; per item: before the asm %.0f cycles, in the asm %.0f, behind it (K/V of the next, stores, scheduler) %.0f; life cycles median %d" % (
    len(pb) - 1, pb[:, 3].min(), pb[:, 3].max(), pb[:, 0].sum() / n, pb[:, 1].sum() / n, pb[:, 2].sum() / n, np.median(w[:, 0])))
ks = a[7000:7000 + 240].reshape(120, 2)[:len(pb)]
print("   of the cycles before the asm: round top -> K / V fetch issued %.0f, -> address arithmetic done %.0f, -> scheduler's part done (asm entered) %.0f" % (
    ks[:, 0].sum() / n, ks[:, 1].sum() / n, (pb[:, 0].sum() - ks.sum()) / n))
ex = a[7300:7300 + 360].reshape(120, 3)[:len(pb)]
print("   scheduler's part (resolve, look-up, post) %.0f, then run lengths / addresses up to the asm %.0f;  behind the asm: -> rows stored %.0f, -> barrier passed %.0f, -> next geometry collected %.0f" % (
    ex[:, 0].sum() / n, (pb[:, 0].sum() - ks.sum() - ex[:, 0].sum()) / n, ex[:, 1].sum() / n, ex[:, 2].sum() / n, (pb[:, 2].sum() - ex[:, 1].sum() - ex[:, 2].sum()) / n))
for x in range(8):
    g = pb[x::8]
    print("   XCD %d: items per workgroup %.1f, cycles per item %.0f + %.0f + %.0f" % (x, g[:, 3].mean(), g[:, 0].sum() / g[:, 3].sum(), g[:, 1].sum() / g[:, 3].sum(), g[:, 2].sum() / g[:, 3].sum()))
