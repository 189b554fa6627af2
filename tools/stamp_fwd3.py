"""Item anatomy of sdpa_fwd3 from a -DHALVA_STAMP build (s_memtime around and inside the generated block, every 9th workgroup):
HALVA_HIP_LIB=halva_amd/libhalva_hip_stamp.so python tools/stamp_fwd3.py [packed]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from halva_amd import hip

packed = len(sys.argv) > 1 and sys.argv[1] == "packed"
S, T, H, D = (8, 3428, 32, 128) if packed else (8, int(os.environ.get("T", 2048)), 32, 128)
dev = "cuda"
qkv = torch.randn(S, T, 3 * H * D, device=dev).to(torch.bfloat16)
out = torch.empty(S, T, H * D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(S, H, T, dtype=torch.float32, device=dev)
ss = torch.zeros(S, dtype=torch.int32, device=dev)
sl = torch.full((S,), T, dtype=torch.int32, device=dev)
ba = torch.full((S,), 668, dtype=torch.int32, device=dev) if packed else None
bb = torch.full((S,), 2048, dtype=torch.int32, device=dev) if packed else None
for _ in range(300):
    hip.call("halva_sdpa_branch_fwd", hip.ptr(qkv), hip.ptr(out), H * D, hip.ptr(lse), hip.ptr(ss), hip.ptr(sl), hip.ptr(ba), hip.ptr(bb), S, T, H, D, 0.0,
             hip.stream_ptr())
torch.cuda.synchronize()
lib = hip.load()
lib.halva_dbg_buffer.restype = ctypes.c_void_p
buf = (ctypes.c_uint64 * 8192)()
ctypes.CDLL("libamdhip64.so").hipMemcpy(buf, ctypes.c_void_p(lib.halva_dbg_buffer()), 8192 * 8, 2)
a = np.frombuffer(buf, dtype=np.uint64)[:30 * 4 * 4 * 16].reshape(30, 4, 4, 16).astype(np.int64)      # [block][the workgroup's first four items][wave]


def d32(x, y):
    return int((int(x) - int(y)) & 0xffffffff)


rows = []
for b in range(30):
    for it in range(4):
        for w in range(4):
            r = a[b, it, w]
            if r[5] == 0:
                continue
            N, qb, blk = int(r[14]) & 0xffff, (int(r[14]) >> 16) & 0xffff, int(r[14]) >> 32
            n = [(int(r[15]) >> (16 * k)) & 0xffff for k in range(4)]
            st = r[6:14]
            rows.append(dict(blk=blk, it=it, w=w, N=N, qb=qb, n=n, pre=int(r[1] - r[0]), asm=int(r[2] - r[1]), post=int(r[5] - r[2]), vote=int(r[3] - r[2]), qld=int(r[4] - r[3]), rows_=int(r[5] - r[4]),
                             req=d32(st[1], st[0]), zero=d32(st[2], st[1]), land=d32(st[3], st[2]), s0=d32(st[4], st[3]), loop=d32(st[5], st[4]),
                             drain=d32(st[6], st[5]), tail=d32(st[7], st[6]), t0=int(r[0]), t3=int(r[5]), raw=[int(x) for x in st]))
print("%4s %2s %1s %3s %2s %-14s | %6s %7s %6s (%5s %5s %5s) | %5s %5s %6s %5s %8s %6s %5s %5s" % ("blk", "it", "w", "N", "qb", "n0/n1/n2/n3", "pre", "asm", "post", "vote", "qload", "rows", "req", "zero", "land", "S0", "loop",
                                                                           "/step", "dr+vo", "tail"))
for r in rows[:64]:
    print("%4d %2d %1d %3d %2d %-14s | %6d %7d %6d (%5d %5d %5d) | %5d %5d %6d %5d %8d %6d %5d %5d" % (r["blk"], r["it"], r["w"], r["N"], r["qb"], "/".join(map(str, r["n"])), r["pre"], r["asm"], r["post"], r["vote"], r["qld"], r["rows_"],
                                                                                 r["req"], r["zero"], r["land"], r["s0"], r["loop"], r["loop"] // max(1, r["N"]), r["drain"], r["tail"]))
rows = [r for r in rows if 0 < r["N"] <= 64 and abs(r["asm"]) < 10 ** 7]
tot = {}
for r in rows:
    for k in ("pre", "asm", "post", "req", "zero", "land", "s0", "loop", "drain", "tail"):
        tot[k] = tot.get(k, 0) + r[k]
    tot["steps"] = tot.get("steps", 0) + r["N"]
    tot["items"] = tot.get("items", 0) + 1
if os.environ.get("STAMP_TAIL") == "1":      # a FWD3_STAMP_TAIL=1 build: stamps 1..3 sit in the tail (row group 0)
    n = len(rows)
    parts = [sum(d32(x, y) for x, y in ((r["raw"][a_], r["raw"][b_]) for r in rows)) / n for a_, b_ in ((1, 6), (2, 1), (3, 2), (7, 3))]
    print("tail (mean over %d wave-items): group 0 Q requests %.0f | group 0 rows + lse %.0f | wait + read-back of group 0's Q %.0f | all of group 1 %.0f" % (n, *parts))
print("per item (mean over %d wave-items): pre %d asm %d post %d | req %d zero %d land %d S0 %d loop %d (%.0f per step) drain+vote %d tail %d"
      % (tot["items"], tot["pre"] / tot["items"], tot["asm"] / tot["items"], tot["post"] / tot["items"], tot["req"] / tot["items"], tot["zero"] / tot["items"],
         tot["land"] / tot["items"], tot["s0"] / tot["items"], tot["loop"] / tot["items"], tot["loop"] / tot["steps"], tot["drain"] / tot["items"], tot["tail"] / tot["items"]))
# workgroup life: entry of item 0 to end of item 1 of wave 0, vs the sum of its parts
life = {}
for r in rows:
    if r["w"] == 0:
        life.setdefault(r["blk"], []).append(r)
for blk, rs in sorted(life.items())[:12]:
    rs.sort(key=lambda x: x["it"])
    print("blk %4d: %s life %d cycles, steps %d" % (blk, " + ".join("item%d(N=%d) %d" % (x["it"], x["N"], x["t3"] - x["t0"]) for x in rs), rs[-1]["t3"] - rs[0]["t0"], sum(x["N"] for x in rs)))
