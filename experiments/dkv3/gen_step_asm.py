#!/usr/bin/env python3
"""Generates the hand-placed instruction stream of one dK/dV step (one wave per SIMD, 32 keys per wave, 64 query rows per step) as ONE
inline-asm block that also holds the step loop:  step_asm.inc = a C string literal for `asm volatile(...)` in step_bench.hip (MODE 2).

Every MFMA, LDS read, vector instruction, store and counted s_waitcnt is assigned to an MFMA gap by this script:
  * A operands travel through a ring of 8 four-register slots; the read(s) for MFMA n are issued in gap n - LOOKAHEAD;
  * the vector work of a sub-tile (P = exp2(S sc - lse) in place over S, dZ = P * dP' in place over dP', two bf16 packs) is spread over the
    gaps behind the chain it depends on, at most CAP issue units per gap (a transcendental counts 2);
  * s_waitcnt lgkmcnt(N) values come from a simulation of the in-order LDS queue over two consecutive iterations (steady state);
  * checks: no vector instruction reads an MFMA result earlier than two MFMAs behind the chain's last one, no instruction consumes a
    v_exp result in the very next issue slot, a packed operand is written >= 4 instructions before the MFMA that reads it.
Operands of the asm block: %0-%3 dV^T accumulators, %4-%7 dK^T accumulators ("+a"), %8-%15 K fragments, %16-%23 V fragments ("a"),
%24 row-read lane offset, %25 transposed-read lane offset, %26 statistics lane offset (bytes, "v"), %27 dS base (64-bit "s"),
%28 scale*log2(e) ("s"), %29 number of steps ("s")."""
import sys

import os
# round 4 timing variants: DKV_LA = how many MFMAs ahead an operand is read, DKV_NRING = slots of the operand ring (8: v176-v207; 16: up to v239,
# the address registers then sit behind it), DKV_CAP
LOOKAHEAD, CAP = int(os.environ.get("DKV_LA", "6")), int(os.environ.get("DKV_CAP", "5"))
NRING = int(os.environ.get("DKV_NRING", "8"))
X = [64, 96]; Y = [80, 112]; PB = [128, 144]; ZB = [136, 152]; SL = 160; RING = 176
QRE, QRO, DRE, DRO, QC0, QC1, DC0, DC1, STAT, DSOFF, TMP = [RING + 4 * NRING + i for i in range(11)]
assert LOOKAHEAD < NRING
S_T, S_CNT, S_TMP, S_TOFF = "s90", "s91", "s92", "s93"
Q_LDS, DO_LDS, LSE_LDS, ND_DELTA, NTILE = 0, 65536, 131072, 1024, 4


def vr(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1) if n > 1 else "v%d" % lo


class Ins:
    def __init__(self, text, kind, reads=(), writes=(), lds_defs=None, cost=0):
        self.text, self.kind, self.reads, self.writes, self.lds_defs, self.cost = text, kind, set(reads), set(writes), lds_defs, cost


def regs(lo, n):
    return ["v%d" % i for i in range(lo, lo + n)]


def mfma_list():
    """(index, text builder, A source descriptor, result regs)"""
    out = []
    for sub in (0, 1):
        for ks in range(8):      # S_sub
            out.append(dict(prod="S", sub=sub, ks=ks, a=("row", "q", sub, ks)))
        for ks in range(8):      # dP_sub
            out.append(dict(prod="dP", sub=sub, ks=ks, a=("row", "do", sub, ks)))
    for sub in (0, 1):
        for i in range(8):
            out.append(dict(prod="dV", sub=sub, i=i, a=("col", "do", sub, i)))
        for i in range(8):
            out.append(dict(prod="dK", sub=sub, i=i, a=("col", "q", sub, i)))
    return out


def a_loads(desc, slot):
    """LDS read instruction(s) filling ring slot `slot` with the A operand `desc`."""
    kind, which, sub, j = desc
    base = RING + 4 * slot
    if kind == "row":
        ks = j
        addr = (QRE, QRO)[ks & 1] if which == "q" else (DRE, DRO)[ks & 1]
        off = 8192 * sub + 512 * (ks >> 1)
        return [Ins("ds_read_b128 %s, v%d offset:%d" % (vr(base, 4), addr, off), "lds", reads=["v%d" % addr], writes=regs(base, 4), lds_defs=regs(base, 4))]
    k16, dt = j // 4, j % 4
    c0, c1 = (QC0, QC1) if which == "q" else (DC0, DC1)
    o0 = 2048 * (4 * sub + 2 * k16) + 512 * dt
    return [Ins("ds_read_b64_tr_b16 %s, v%d offset:%d" % (vr(base, 2), c0, o0), "lds", reads=["v%d" % c0], writes=regs(base, 2), lds_defs=regs(base, 2)),
            Ins("ds_read_b64_tr_b16 %s, v%d offset:%d" % (vr(base + 2, 2), c1, o0 + 2048), "lds", reads=["v%d" % c1], writes=regs(base + 2, 2),
                lds_defs=regs(base + 2, 2))]


def mfma_ins(n, m):
    slot = RING + 4 * (n % NRING)
    a = vr(slot, 4)
    if m["prod"] == "S":
        d = vr(X[m["sub"]], 16)
        c = "0" if m["ks"] == 0 else d
        return Ins("v_mfma_f32_32x32x16_bf16 %s, %s, %%%d, %s" % (d, a, 8 + m["ks"], c), "mfma", reads=regs(slot, 4) + (regs(X[m["sub"]], 16) if m["ks"] else []),
                   writes=regs(X[m["sub"]], 16))
    if m["prod"] == "dP":
        d = vr(Y[m["sub"]], 16)
        return Ins("v_mfma_f32_32x32x16_bf16 %s, %s, %%%d, %s" % (d, a, 16 + m["ks"], d), "mfma", reads=regs(slot, 4) + regs(Y[m["sub"]], 16), writes=regs(Y[m["sub"]], 16))
    k16, dt = m["i"] // 4, m["i"] % 4
    if m["prod"] == "dV":
        b = PB[m["sub"]] + 4 * k16
        return Ins("v_mfma_f32_32x32x16_bf16 %%%d, %s, %s, %%%d" % (dt, a, vr(b, 4), dt), "mfma", reads=regs(slot, 4) + regs(b, 4))
    b = ZB[m["sub"]] + 4 * k16
    return Ins("v_mfma_f32_32x32x16_bf16 %%%d, %s, %s, %%%d" % (4 + dt, a, vr(b, 4), 4 + dt), "mfma", reads=regs(slot, 4) + regs(b, 4))


def valu_ops(sub):
    """Ordered vector work of a sub-tile: (stage, Ins).  stage 1 needs the S chain, stage 2 the dP chain."""
    x, y, pb, zb = X[sub], Y[sub], PB[sub], ZB[sub]
    A = lambda r: Ins("v_fma_f32 v%d, v%d, %%28, -v%d" % (x + r, x + r, SL + r), "valu", reads=["v%d" % (x + r), "v%d" % (SL + r)], writes=["v%d" % (x + r)], cost=1)
    B = lambda r: Ins("v_exp_f32_e32 v%d, v%d" % (x + r, x + r), "trans", reads=["v%d" % (x + r)], writes=["v%d" % (x + r)], cost=2)
    C = lambda r: Ins("v_mul_f32_e32 v%d, v%d, v%d" % (y + r, x + r, y + r), "valu", reads=["v%d" % (x + r), "v%d" % (y + r)], writes=["v%d" % (y + r)], cost=1)
    Dp = lambda i: Ins("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (pb + i, x + 2 * i, x + 2 * i + 1), "valu", reads=["v%d" % (x + 2 * i), "v%d" % (x + 2 * i + 1)],
                       writes=["v%d" % (pb + i)], cost=1)
    E = lambda i: Ins("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (zb + i, y + 2 * i, y + 2 * i + 1), "valu", reads=["v%d" % (y + 2 * i), "v%d" % (y + 2 * i + 1)],
                      writes=["v%d" % (zb + i)], cost=1)
    s1 = [A(0), A(1), A(2), A(3)]
    for r in range(12):
        s1 += [B(r), A(r + 4)]
    s1 += [B(12), B(13), B(14), B(15)]
    s2 = []
    for i in range(8):
        s2 += [C(2 * i), C(2 * i + 1), Dp(i)]
        if i >= 1:
            s2.append(E(i - 1))
    s2.append(E(7))
    return [(1, o) for o in s1] + [(2, o) for o in s2]


def build_body():
    M = mfma_list()
    gaps = [[] for _ in range(64)]      # instructions behind MFMA n
    # --- LDS: A operands (for MFMA n at gap n - LOOKAHEAD; the first LOOKAHEAD MFMAs of the NEXT step at the last gaps, after the address update)
    for n in range(64):
        g = n - LOOKAHEAD
        if g >= 0:
            gaps[g] += a_loads(M[n]["a"], n % NRING)
    addr_update_gap = 64 - LOOKAHEAD - 1            # after the reads for MFMA 63
    # --- LDS: statistics.  -delta goes straight into the dP accumulator, lse into SL
    def stat(sub, j, which):
        off = (ND_DELTA if which == "nd" else 0) + 128 * sub + 32 * j
        dst = (Y[sub] if which == "nd" else SL) + 4 * j
        return Ins("ds_read_b128 %s, v%d offset:%d" % (vr(dst, 4), STAT, off), "lds", reads=["v%d" % STAT], writes=regs(dst, 4), lds_defs=regs(dst, 4))
    for j in range(4):
        gaps[0 + j].append(stat(0, j, "nd"))       # before MFMA 8 (first of dP0)
        gaps[4 + j].append(stat(0, j, "lse"))      # before gap 9
        gaps[15 + j].append(stat(1, j, "nd"))      # before MFMA 24; Y1's last reader (pack of dZ1) sits before gap 48 of the previous step
        gaps[19 + j].append(stat(1, j, "lse"))     # SL is free after gap 16 (last fma of sub-tile 0), needed from gap 25
    # --- vector work
    chain_end = {("S", 0): 7, ("dP", 0): 15, ("S", 1): 23, ("dP", 1): 31}
    used = [sum(i.cost for i in g) for g in gaps]
    for sub in (0, 1):
        g = 0
        for stage, ins in valu_ops(sub):
            earliest = chain_end[("S", sub)] + 2 if stage == 1 else max(chain_end[("dP", sub)] + 2, g)
            g = max(g, earliest)
            while used[g] + ins.cost > CAP:
                g += 1
            gaps[g].append(ins)
            used[g] += ins.cost
        assert g < 48, "vector work of a sub-tile ran past its packs' consumers"
    # --- dS out: two 16-byte stores per sub-tile once its packs exist
    def last_gap_writing(regnames):
        lg = -1
        for gi, g in enumerate(gaps):
            for ins in g:
                if ins.writes & set(regnames):
                    lg = max(lg, gi)
        return lg
    for sub in (0, 1):
        g0 = last_gap_writing(regs(ZB[sub], 8)) + 1
        for half in (0, 1):
            gaps[g0 + half].append(Ins("global_store_dwordx4 v%d, %s, %%27 offset:%d nt" % (DSOFF, vr(ZB[sub] + 4 * half, 4), 2048 * sub + 1024 * half), "vmem",
                                       reads=regs(ZB[sub] + 4 * half, 4) + ["v%d" % DSOFF]))
    # --- next step: tile rotation + addresses, then the reads of its first LOOKAHEAD MFMAs
    upd = [Ins("s_add_u32 %s, %s, 1" % (S_T, S_T), "salu"), Ins("s_and_b32 %s, %s, %d" % (S_TMP, S_T, NTILE - 1), "salu"),
           Ins("s_lshl_b32 %s, %s, 14" % (S_TOFF, S_TMP), "salu")]
    upd += addr_setup(loop=True)
    gaps[addr_update_gap] += upd
    for n in range(LOOKAHEAD):
        gaps[64 - LOOKAHEAD + n] += a_loads(M[n]["a"], n % NRING)
    return M, gaps


def addr_setup(loop):
    """address registers of the tile S_TOFF selects (and of step S_T for the statistics / dS slot)"""
    o = []
    o.append(Ins("v_add_u32_e32 v%d, %s, %%24" % (QRE, S_TOFF), "valu", writes=["v%d" % QRE], cost=1))
    o.append(Ins("v_xor_b32_e32 v%d, 32, v%d" % (QRO, QRE), "valu", reads=["v%d" % QRE], writes=["v%d" % QRO], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DRE, DO_LDS, QRE), "valu", reads=["v%d" % QRE], writes=["v%d" % DRE], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DRO, DO_LDS, QRO), "valu", reads=["v%d" % QRO], writes=["v%d" % DRO], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %s, %%25" % (QC0, S_TOFF), "valu", writes=["v%d" % QC0], cost=1))
    o.append(Ins("v_xor_b32_e32 v%d, 32, v%d" % (QC1, QC0), "valu", reads=["v%d" % QC0], writes=["v%d" % QC1], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DC0, DO_LDS, QC0), "valu", reads=["v%d" % QC0], writes=["v%d" % DC0], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DC1, DO_LDS, QC1), "valu", reads=["v%d" % QC1], writes=["v%d" % DC1], cost=1))
    o.append(Ins("s_lshl_b32 %s, %s, 8" % (S_TMP, S_TMP), "salu"))                     # (tile & 3) * 256 bytes of statistics
    o.append(Ins("s_add_u32 %s, %s, %d" % (S_TMP, S_TMP, LSE_LDS), "salu"))
    o.append(Ins("v_add_u32_e32 v%d, %s, %%26" % (STAT, S_TMP), "valu", writes=["v%d" % STAT], cost=1))
    o.append(Ins("s_and_b32 %s, %s, 7" % (S_TMP, S_T), "salu"))
    o.append(Ins("s_lshl_b32 %s, %s, 12" % (S_TMP, S_TMP), "salu"))
    o.append(Ins("v_add_u32_e32 v%d, %s, v%d" % (DSOFF, S_TMP, TMP), "valu", reads=["v%d" % TMP], writes=["v%d" % DSOFF], cost=1))
    return o


def linearize(M, gaps):
    seq = []
    for n in range(64):
        seq.append(mfma_ins(n, M[n]))
        seq[-1].mfma_index = n
        seq += gaps[n]
    return seq


def insert_waits(seq, carried):
    """seq: one iteration; carried: LDS ops outstanding at its top (issued at the end of the previous iteration / the prologue), oldest first.
    Returns (text lines, ops outstanding at the end)."""
    fifo = list(carried)          # each: set of regs it defines
    pending = {}                  # reg -> fifo entry object
    for e in fifo:
        for r in e["defs"]:
            pending[r] = e
    lines = []
    prev = None
    for ins in seq:
        need = [pending[r] for r in (ins.reads | ins.writes) if r in pending]
        if need:
            last = max(fifo.index(e) for e in need)
            cnt = len(fifo) - 1 - last
            assert cnt <= 15, "lgkmcnt field overflow"
            lines.append("s_waitcnt lgkmcnt(%d)" % cnt)
            for e in fifo[:last + 1]:
                for r in e["defs"]:
                    if pending.get(r) is e:
                        del pending[r]
            fifo = fifo[last + 1:]
        if prev is not None and prev.kind == "trans" and ins.kind in ("valu", "trans", "mfma", "vmem") and (prev.writes & ins.reads):
            lines.append("s_nop 0")
        lines.append(ins.text)
        if ins.kind == "lds":
            e = {"defs": set(ins.lds_defs)}
            fifo.append(e)
            for r in e["defs"]:
                pending[r] = e
        assert len(fifo) <= 15, "more than 15 LDS reads in flight"
        if ins.kind != "salu":
            prev = ins
    return lines, fifo


def check(seq):
    pos_mfma = [i for i, s in enumerate(seq) if s.kind == "mfma"]
    # vector reads of an MFMA result: two MFMAs behind the chain's last one
    last_writer = {}
    for i, s in enumerate(seq):
        if s.kind in ("valu", "trans", "vmem"):
            for r in s.reads:
                if r in last_writer and last_writer[r][0] == "mfma":
                    n_between = sum(1 for p in pos_mfma if last_writer[r][1] < p < i)
                    assert n_between >= 2, "%s reads %s too close behind its MFMA chain" % (s.text, r)
        if s.kind == "mfma":
            for r in s.reads:
                if r in last_writer and last_writer[r][0] in ("valu", "trans"):
                    assert i - last_writer[r][1] >= 4, "%s reads %s right behind the vector write" % (s.text, r)
        for r in s.writes:
            last_writer[r] = (s.kind if s.kind != "lds" else "lds", i)


def main():
    M, gaps = build_body()
    seq = linearize(M, gaps)
    check(seq + seq)
    carried = []
    for n in range(LOOKAHEAD):
        for l in a_loads(M[n]["a"], n % NRING):
            carried.append({"defs": set(l.lds_defs)})
    lines1, fifo1 = insert_waits(seq, carried)
    lines2, fifo2 = insert_waits(seq, fifo1)
    assert lines1 == lines2 and [sorted(e["defs"]) for e in fifo1] == [sorted(e["defs"]) for e in fifo2], "loop is not in steady state"
    pro = ["v_mbcnt_lo_u32_b32 v%d, -1, 0" % TMP, "v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (TMP, TMP), "v_lshlrev_b32_e32 v%d, 4, v%d" % (TMP, TMP),
           "s_mov_b32 %s, 0" % S_T, "s_mov_b32 %s, 0" % S_TMP, "s_mov_b32 %s, 0" % S_TOFF, "s_mov_b32 %s, %%29" % S_CNT]
    pro += [i.text for i in addr_setup(loop=False)]
    for n in range(LOOKAHEAD):
        pro += [l.text for l in a_loads(M[n]["a"], n % NRING)]
    body = ["1:"] + lines1 + ["s_sub_u32 %s, %s, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 %s, 0" % S_CNT, "s_cbranch_scc1 1b", "s_waitcnt lgkmcnt(0)"]
    out = sys.argv[1] if len(sys.argv) > 1 else "step_asm.inc"
    with open(out, "w") as f:
        f.write("// generated by gen_step_asm.py - do not edit\n")
        for l in pro + body:
            f.write('"%s\\n\\t"\n' % l)
    nv = sum(1 for s in seq if s.kind in ("valu", "trans"))
    nl = sum(1 for s in seq if s.kind == "lds")
    print("step: 64 MFMAs, %d vector, %d LDS reads, %d lines; busiest gap %d issue units" % (nv, nl, len(lines1), max(sum(i.cost for i in g) for g in gaps)))
    clob = ", ".join('"v%d"' % i for i in range(64, RING + 4 * NRING + 12)) + ', "s90", "s91", "s92", "s93", "scc", "memory"'
    with open(out.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write(clob + "\n")


if __name__ == "__main__":
    main()
