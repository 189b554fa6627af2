import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import subprocess, torch
T = int(os.environ.get("T", 256))
if os.environ.get("CHILD"):
    from halva_amd import kernels as K
    torch.manual_seed(0)
    S, H, D = 1, 1, 128
    qkv = torch.randn(S, T, 3 * H * D, device="cuda").to(torch.bfloat16)
    ss = torch.zeros(S, dtype=torch.int32, device="cuda"); sl = torch.full((S,), T, dtype=torch.int32, device="cuda")
    out = K.sdpa_causal(qkv, ss, sl, H, D)
    torch.save(out.float().cpu(), os.environ["CHILD"])
    sys.exit(0)
outs = {}
for tag, env in (("old", {"HALVA_FWD4": "0"}), ("new", {"HALVA_FWD4": "1", "HALVA_HIP_LIB": os.environ.get("LIB", "")})):
    e = dict(os.environ, CHILD="/tmp/dbg_%s.pt" % tag, **{k: v for k, v in env.items() if v})
    subprocess.run([sys.executable, __file__], env=e, check=True)
    outs[tag] = torch.load("/tmp/dbg_%s.pt" % tag)
d = (outs["new"] - outs["old"]).abs()[0]          # [T, 128]
print("T", T, "max err", float(d.max()))
rows = d.max(dim=1).values
for r0 in range(0, T, 32):
    print("rows %4d..%4d  max %.3e" % (r0, r0 + 31, float(rows[r0:r0 + 32].max())), " cols worst:", [int(c) for c in d[r0:r0+32].max(dim=0).values.topk(3).indices])
