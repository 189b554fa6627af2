"""Re-rank the library GEMM kernels of the headline step's heavy shapes by their SUSTAINED time (tools/lt_sustained.cpp) and write a candidate table.

usage (GPU box): python tools/tune_gemms_sustained.py [--min-ms 0.6] [--topk 12] [--secs 0.25] [--rows 54848,32768] [--out gpurun_out/sustained]
  reads halva_amd/tuned/gfx950_tunableop.csv, takes every GemmTunableOp_BFloat16 row whose n (token rows) is one of --rows and whose recorded burst time is
  >= --min-ms, runs the harness on it, and writes
    <out>_report.json   per shape: the candidates (library, index, burst ms, sustained ms x 2, kernel name), the table's current pick and its sustained time
    <out>_table.csv     the shipped table with a row replaced wherever a candidate's sustained time (worse of the two rounds) beats the current pick's
                        (better of its two rounds) by more than --margin (default 1 %)
The result is only a CANDIDATE: whether it pays is decided by an alternating A/B of bench.py (HALVA_GEMM_TABLE=<out>_table.csv against the shipped table)."""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: the harness must bind to PyTorch's own hipBLASLt / rocBLAS / HIP runtime)

from halva_amd.gemm_tuning import SHIPPED, table_entries  # noqa: E402


def build():
    so = os.path.join(ROOT, "tools", "_lt_sustained.so")
    src = os.path.join(ROOT, "tools", "lt_sustained.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        subprocess.check_call(["hipcc", "-O2", "-shared", "-fPIC", "-w", "-D__HIP_PLATFORM_AMD__", "-DROCBLAS_BETA_FEATURES_API", "-DROCBLAS_NO_DEPRECATED_WARNINGS", src,
                               "-o", so, "-I/opt/rocm/include", "-L" + tl, "-lhipblaslt", "-lrocblas"])
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-ms", type=float, default=0.6)
    ap.add_argument("--topk", type=int, default=12)
    ap.add_argument("--secs", type=float, default=0.25)
    ap.add_argument("--rows", default="54848,32768")
    ap.add_argument("--margin", type=float, default=0.01)
    ap.add_argument("--only", default="", help="comma-separated shape keys (default: by --rows / --min-ms)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sustained"))
    args = ap.parse_args()
    torch.zeros(1, device="cuda")      # PyTorch's HIP context first
    torch.mm(torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16), torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16))      # and its BLAS libraries loaded
    lib = ctypes.CDLL(build())
    lib.lt_sustained.restype = ctypes.c_int
    lib.lt_sustained.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 6 + [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                 ctypes.c_char_p, ctypes.c_int]
    from halva_amd.gemm_tuning import enable_tuned_gemms
    enable_tuned_gemms()      # torch.mm below = what the step runs today (the table's pick, "Default" included)

    def torch_sustained(ta, tb, m, n, k, lda, ldb, ldc, secs):
        """the same problem through torch.mm (row-major: C^T[n, m] = B^T A^T), back to back for `secs`, twice; ms per launch"""
        u = lambda r, c, ld: (torch.rand(r, ld, device="cuda") * 2 - 1).to(torch.bfloat16)[:, :c]
        a = u(m, k, lda) if ta else u(k, m, lda)           # column-major (k x m, lda) for T = row-major [m, k]
        b = u(n, k, ldb) if not tb else u(k, n, ldb)       # column-major (k x n, ldb) for N = row-major [n, k]
        c = torch.empty(n, ldc, device="cuda", dtype=torch.bfloat16)[:, :m]
        lhs = b if not tb else b.t()                       # [n, k]
        rhs = a.t() if ta else a                           # [k, m]
        out = []
        for _ in range(2):
            t0 = time.time()
            while time.time() - t0 < 0.3 * secs:
                for _ in range(8):
                    torch.mm(lhs, rhs, out=c)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nl = 0
            e0.record()
            t0 = time.time()
            while time.time() - t0 < secs:
                for _ in range(8):
                    torch.mm(lhs, rhs, out=c)
                nl += 8
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / nl)
        return out

    validators, rows = table_entries(SHIPPED)
    want_rows = {int(r) for r in args.rows.split(",")}
    only = {s for s in args.only.split(",") if s}
    report, replaced = {}, {}
    for op, key, sol, ms in rows:
        mt = re.match(r"GemmTunableOp_BFloat16_(\w\w)$", op)
        mk = re.match(r"(\w\w)_(\d+)_(\d+)_(\d+)_ld_(\d+)_(\d+)_(\d+)$", key)
        if not mt or not mk:
            continue
        m, n, k, lda, ldb, ldc = (int(x) for x in mk.groups()[1:])
        if only:
            if key not in only:
                continue
        elif n not in want_rows or ms < args.min_ms:
            continue
        ta, tb = (1 if c == "T" else 0 for c in mt.group(1))
        buf = (ctypes.c_double * (5 * (args.topk + 1)))()
        names = ctypes.create_string_buffer(1 << 16)
        cnt = lib.lt_sustained(ta, tb, m, n, k, lda, ldb, ldc, args.topk, args.secs, 1, buf, names, len(names))
        if cnt <= 0:
            print(key, "harness failed", cnt, flush=True)
            continue
        nm = names.value.decode().split("\n")
        cands = [{"solution": ("Gemm_Hipblaslt_%d" if buf[5 * i] == 0 else "Gemm_Rocblas_%d") % int(buf[5 * i + 1]), "burst_ms": round(buf[5 * i + 2], 4),
                  "sustained_ms": [round(buf[5 * i + 3], 4), round(buf[5 * i + 4], 4)], "kernel": nm[i] if i < len(nm) else ""} for i in range(cnt)]
        cur = next((c for c in cands if c["solution"] == sol), None)
        tms = torch_sustained(ta, tb, m, n, k, lda, ldb, ldc, args.secs)
        flop = 2.0 * m * n * k
        by = sorted(cands, key=lambda c: max(c["sustained_ms"]))
        bestc = by[0]
        rec = {"table_pick": sol, "table_burst_ms": ms, "table_pick_measured": cur, "torch_mm_sustained_ms": [round(x, 4) for x in tms], "best_sustained": bestc, "candidates": cands,
               "pflops_best_sustained": round(flop / max(bestc["sustained_ms"]) / 1e12, 3),
               "pflops_table_pick_sustained": None if cur is None else round(flop / min(cur["sustained_ms"]) / 1e12, 3)}
        report[key] = rec
        # replace when today's choice is beaten by the margin; a "Default" pick (the library's heuristic) cannot be timed by name in the harness: torch.mm times it
        ref = min(min(cur["sustained_ms"]) if cur is not None else 1e30, min(tms))      # what the step has today: the pick in the harness or through torch.mm
        if max(bestc["sustained_ms"]) < (1 - args.margin) * ref and bestc["solution"] != sol:
            replaced[(op, key)] = bestc["solution"]
        print("%-48s table %-22s %s torch.mm %.3f / %.3f | best sustained %-22s %.3f / %.3f ms (burst %.3f) = %.2f PF%s" % (
            key, sol, "%.3f / %.3f ms" % tuple(cur["sustained_ms"]) if cur else "(not among the top %d by burst)" % args.topk, tms[0], tms[1], bestc["solution"], *bestc["sustained_ms"],
            bestc["burst_ms"], flop / max(bestc["sustained_ms"]) / 1e12, "  -> REPLACED" if (op, key) in replaced else ""), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"args": vars(args), "shapes": report, "replaced": {k[1]: v for k, v in replaced.items()}}, open(args.out + "_report.json", "w"), indent=1)
    with open(SHIPPED) as f, open(args.out + "_table.csv", "w") as g:
        for line in f:
            parts = line.rstrip("\n").split(",")
            if len(parts) >= 4 and (parts[0], parts[1]) in replaced:
                parts[2] = replaced[(parts[0], parts[1])]
                rec = report[parts[1]]["best_sustained"]
                parts[3] = "%g" % max(rec["sustained_ms"])
                line = ",".join(parts) + "\n"
            g.write(line)
    print("replaced %d of %d shapes; table: %s_table.csv" % (len(replaced), len(report), args.out))


if __name__ == "__main__":
    main()
