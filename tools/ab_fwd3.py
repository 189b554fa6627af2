"""A/B of the causal forward: sdpa_fwd3 (HALVA_SDPA_FWD3=1, default) against the two-waves-per-SIMD kernel (=0) and an fp32 dense-mask
reference computed on the GPU; correctness on edge-case layouts, then timings at the bench's shapes.  usage: python tools/ab_fwd3.py [quick]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from halva_amd import hip, kernels as K  # noqa: E402

hip.load()
DEV = "cuda:0"


def ref_attn(qkv, starts, lens, br_a, br_b, H, D):
    S, T = qkv.shape[0], qkv.shape[1]
    x = qkv.view(S, T, 3, H, D).float()
    out = torch.zeros(S, T, H, D, device=qkv.device)
    lse = torch.zeros(S, H, T, device=qkv.device)
    for s in range(S):
        L, st = lens[s], starts[s]
        if L == 0:
            continue
        q, k, v = (x[s, st:st + L, i].permute(1, 0, 2) for i in range(3))
        idx = torch.arange(L, device=qkv.device)
        ok = idx[None, :] <= idx[:, None]
        if br_a is not None:
            ok &= ~((idx[:, None] >= br_b[s]) & (idx[None, :] >= br_a[s]) & (idx[None, :] < br_b[s]))
        att = (q @ k.transpose(1, 2)) / math.sqrt(D)
        att = att.masked_fill(~ok[None], float("-inf"))
        lse[s, :, st:st + L] = torch.logsumexp(att, -1)
        out[s, st:st + L] = (att.softmax(-1) @ v).permute(1, 0, 2)
    return out, lse


def run(qkv, starts, lens, br_a, br_b, H, D, flag):
    os.environ["HALVA_SDPA_FWD3"] = flag
    S, T = qkv.shape[0], qkv.shape[1]
    mk = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    out = torch.empty(S, T, H * D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(S, H, T, dtype=torch.float32, device=DEV)
    ss, sl = mk(starts), mk(lens)
    ba, bb = (mk(br_a), mk(br_b)) if br_a is not None else (None, None)
    hip.call("halva_sdpa_branch_fwd", hip.ptr(qkv), hip.ptr(out), H * D, hip.ptr(lse), hip.ptr(ss), hip.ptr(sl), hip.ptr(ba), hip.ptr(bb),
             S, T, H, D, 0.0, hip.stream_ptr())
    torch.cuda.synchronize()
    return out.view(S, T, H, D).float(), lse


def case(name, T, lens, starts, br_a, br_b, H, seed=0, scale=1.0, special=None):
    D = 128
    S = len(lens)
    g = torch.Generator(device=DEV).manual_seed(seed)
    qkv = (torch.randn(S, T, 3 * H * D, generator=g, device=DEV) * scale).to(torch.bfloat16)
    if special is not None:
        qkv = special(qkv.view(S, T, 3, H, D)).reshape(S, T, 3 * H * D).contiguous()
    ref, rlse = ref_attn(qkv, starts, lens, br_a, br_b, H, D)
    res = {}
    for flag in ("1", "0"):
        o, lse = run(qkv, starts, lens, br_a, br_b, H, D, flag)
        fin = bool(torch.isfinite(o).all())
        err = float((o - ref).abs().max()) if fin else float("nan")
        rel = float((o - ref).norm() / ref.norm()) if fin else float("nan")
        m = torch.zeros(S, T, dtype=torch.bool, device=DEV)
        for s in range(S):
            m[s, starts[s]:starts[s] + lens[s]] = True
        pad = float(o[~m].abs().sum())
        lerr = float((lse - rlse).abs().permute(0, 2, 1)[m].max())
        res[flag] = (err, rel, pad, lerr, fin)
        if flag == "1" and (not fin or rel > 1e-2):
            bad = ((o - ref).abs().amax(dim=(2, 3)) > 0.05) | ~torch.isfinite(o).all(dim=3).all(dim=2)
            for s in range(S):
                rows = torch.nonzero(bad[s]).flatten().tolist()
                if rows:
                    print("    seq %d bad rows: %d of %d, first %s last %s" % (s, len(rows), T, rows[:8], rows[-4:]))
    ok = res["1"][4] and res["1"][1] < 1e-2 and res["1"][2] == 0 and res["1"][3] < 2e-2
    print("%-34s fwd3: max %.3e rel %.3e pad %.1e lse %.2e | old: max %.3e rel %.3e lse %.2e  %s"
          % (name, res["1"][0], res["1"][1], res["1"][2], res["1"][3], res["0"][0], res["0"][1], res["0"][3], "ok" if ok else "FAIL"))
    return ok


def timing(name, S, T, H, lens, br_a, br_b, iters=20):
    D = 128
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = torch.randn(S, T, 3 * H * D, generator=g, device=DEV).to(torch.bfloat16)
    mk = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    ss, sl = mk([0] * S), mk(lens)
    ba, bb = (mk(br_a), mk(br_b)) if br_a is not None else (None, None)
    out = torch.empty(S, T, H * D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(S, H, T, dtype=torch.float32, device=DEV)
    pairs = 0
    for s in range(S):
        if br_a is None:
            pairs += lens[s] * (lens[s] + 1) // 2
        else:
            la, lb = min(lens[s], br_b[s]) - br_a[s], max(0, lens[s] - br_b[s])
            n1 = br_a[s] + la
            pairs += n1 * (n1 + 1) // 2 + lb * br_a[s] + lb * (lb + 1) // 2
    flop = 4.0 * D * pairs * H
    line = []
    for flag in ("1", "0", "1", "0"):
        os.environ["HALVA_SDPA_FWD3"] = flag
        def go():
            hip.call("halva_sdpa_branch_fwd", hip.ptr(qkv), hip.ptr(out), H * D, hip.ptr(lse), hip.ptr(ss), hip.ptr(sl), hip.ptr(ba), hip.ptr(bb),
                     S, T, H, D, 0.0, hip.stream_ptr())
        for _ in range(5):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        line.append("%s %.1f us %.0f TF/s (%.3f)" % ("fwd3" if flag == "1" else "old ", ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 2500))
    print("%-24s %s" % (name, " | ".join(line)))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ok = True
    ok &= case("T=64 one tile", 64, [64], [0], None, None, 1)
    ok &= case("T=20 tiny, ragged", 20, [20, 13], [0, 0], None, None, 2)
    ok &= case("T=200 ragged, len 1", 200, [200, 77, 1], [0, 0, 0], None, None, 2)
    ok &= case("T=333 left padded", 333, [333, 300], [0, 33], None, None, 1)
    ok &= case("T=512 aligned", 512, [512, 129], [0, 0], None, None, 2)
    ok &= case("T=1024", 1024, [1024, 1000], [0, 0], None, None, 2)
    ok &= case("branch 700", 700, [700, 650, 300], [0, 0, 0], [100, 257, 512], [384, 448, 512], 2)
    ok &= case("branch 1100 wholly-B blocks", 1100, [1100, 1000], [0, 0], [628, 0], [832, 512], 1)
    ok &= case("branch 520 short B", 520, [520, 513], [0, 0], [130, 511], [512, 512], 1)
    ok &= case("branch 2048 bench geometry", 2048, [2048], [0], [628], [1344], 1)
    ok &= case("packed 3428 bench row", 3428, [3428, 3428], [0, 0], [668, 668], [2048, 2048], 2)
    ok &= case("large scores (x6)", 512, [512], [0], None, None, 2, scale=6.0)

    def grow(x):      # keys whose scores grow by far more than 64 log2 units from tile to tile: forces the repeat against the true maxima
        S, T, _, H, D = x.shape
        u = torch.randn(H, D, device=x.device)
        u = u / u.norm(dim=-1, keepdim=True) * math.sqrt(D)
        c = torch.tensor([0.1, 3.0, 8.0, 20.0], device=x.device).repeat_interleave(64) * (11.3 / math.sqrt(D))
        y = x.float().clone()
        y[0, :, 0] = u[None] + 0.05 * torch.randn(T, H, D, device=x.device)
        y[0, :, 1] = c[:, None, None] * (u[None] + 0.3 * torch.randn(T, H, D, device=x.device))
        return y.to(torch.bfloat16)
    ok &= case("reference outgrown (repeat path)", 256, [256], [0], None, None, 2, special=grow)
    print("ALL OK" if ok else "SOME FAILED")
    if not quick:
        timing("8 x 2048 H32", 8, 2048, 32, [2048] * 8, None, None)
        timing("16 x 2048 H32", 16, 2048, 32, [2048] * 16, None, None)
        timing("16 x 3428 packed H32", 16, 3428, 32, [3428] * 16, [668] * 16, [2048] * 16)
        timing("4 x 4096 H40", 4, 4096, 40, [4096] * 4, None, None)


if __name__ == "__main__":
    main()
