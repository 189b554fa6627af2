"""In-kernel clock of every SDPA kernel (-DHALVA_STAMP build): each sampled workgroup brackets its whole life with s_memtime (shader cycles) and
s_memrealtime (100 MHz) - the clock the chip holds IN that kernel (MI355X_MICROARCH.md, DVFS give-back item 6), as opposed to what a co-resident
probe wave sees (halva_clock_probe).   HALVA_HIP_LIB=<stamped build> [HALVA_SDPA_DKV3=0|1] python tools/stamp_clock.py"""
import os, sys, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from halva_amd import hip, kernels as K
S, T, H, D = 8, 2048, 32, 128
dev = "cuda"
qkv = torch.randn(S, T, 3 * H * D, device=dev).to(torch.bfloat16).requires_grad_(True)
dout = torch.randn(S, T, H * D, device=dev).to(torch.bfloat16)
ss = torch.zeros(S, dtype=torch.int32, device=dev); sl = torch.full((S,), T, dtype=torch.int32, device=dev)
import time
t0 = time.time()
while time.time() - t0 < 2.0:      # two seconds back to back, so that the clock has settled
    for _ in range(20):
        qkv.grad = None
        out = K.sdpa_causal(qkv, ss, sl, H, D); out.backward(dout)
    torch.cuda.synchronize()
lib = hip.load(); lib.halva_dbg_buffer.restype = ctypes.c_void_p
buf = (ctypes.c_uint64 * 8192)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(lib.halva_dbg_buffer()), 8192 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
out = {}
for region, name in enumerate(("sdpa_fwd", "sdpa_bwd_dkv2", "sdpa_bwd_dq2", "sdpa_bwd_dkv3")):
    w = a[4096 + region * 480: 4096 + region * 480 + 480].reshape(120, 4)
    w = w[w[:, 1] > 0]
    if not len(w): continue
    mhz = 100.0 * w[:, 0] / w[:, 1]
    out[name] = {"workgroups_sampled": int(len(w)), "cycles_per_workgroup_median": int(np.median(w[:, 0])), "us_per_workgroup_median": round(float(np.median(w[:, 1])) / 100.0, 1),
                 "in_kernel_mhz_median": round(float(np.median(mhz)), 0), "in_kernel_mhz_p10": round(float(np.percentile(mhz, 10)), 0), "in_kernel_mhz_p90": round(float(np.percentile(mhz, 90)), 0)}
    # the samples are the first 120 workgroups of XCD 0 (blockIdx % 8 == 0), 32 CUs, one workgroup per CU for the dK/dV kernels: the k-th start
    # (k >= slots) re-uses the CU the (k - slots)-th end freed -> how long a CU sits between two workgroups
    st = np.sort(w[:, 2]); en = np.sort(w[:, 2] + w[:, 1])
    out[name]["us_per_workgroup_p10_p90"] = [round(float(np.percentile(w[:, 1], 10)) / 100.0, 1), round(float(np.percentile(w[:, 1], 90)) / 100.0, 1)]
    first = int((st < st[0] + 200).sum())          # workgroups that started within 2 us of the first = resident slots on this XCD
    out[name]["resident_slots_xcd0"] = first
    if first < len(st):
        gaps = (st[first:] - en[:len(st) - first]) / 100.0
        out[name]["us_between_workgroups_on_a_slot_median_p90"] = [round(float(np.median(gaps)), 2), round(float(np.percentile(gaps, 90)), 2)]
    print(name, out[name])
if len(sys.argv) > 1:
    json.dump({"tool": "tools/stamp_clock.py", "shape": {"S": S, "T": T, "H": H, "D": D}, "commit": os.environ.get("HALVA_COMMIT"), "kernels": out}, open(sys.argv[1], "w"), indent=1)
