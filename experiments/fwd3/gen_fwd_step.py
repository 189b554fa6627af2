#!/usr/bin/env python3
"""Generates the hand-placed instruction stream of one FORWARD attention step for a one-wave-per-SIMD kernel (next round's candidate for
sdpa_causal_fwd) as ONE inline-asm block that also holds the step loop:  fwd_step_asm.inc, for step_bench.hip (MODE 2).

Orientation (the one sdpa_bwd_dkv3 uses, roles swapped): a wave owns 64 QUERIES (two groups g of 32), their Q fragments (B operands, 64
registers) and their O^T accumulators (2 x 4 x 16 = 128 registers) in the accumulator file; per 64-key step it computes
    S^T_g[key][query] = K Q_g^T      (A = K tile rows from LDS - the same A operand serves both groups - , B = Q fragments;  lane = query,
                                      registers = keys: the row statistics of a query are per-LANE, no cross-lane reduction in the loop)
    P = exp2(S^T sc - m_ref)         in place, m_ref a per-query reference that is NOT updated in this stream (the running maximum is tracked;
                                      a kernel would rescale through a rare out-of-line path when it moves by more than a threshold)
    O^T_g[d][query] += V^T P_g       (A = V^T by transposed reads of the V tile, again shared by both groups, B = P packed to bf16)
= 64 MFMAs per step, 48 LDS reads (32 distinct A operands), ~230 vector instructions.  The S chains of tile t+1 are interleaved with the O
products of tile t (software pipelining over the step loop), so that the vector work of a key half has half a step of MFMA gaps:
    n =  0..15  O(t)   keys  0..31      n = 16..31  S(t+1) keys  0..31  -> its vector work in gaps 33..61
    n = 32..47  O(t)   keys 32..63      n = 48..63  S(t+1) keys 32..63  -> its vector work in gaps 1..29 of the NEXT iteration
Same machinery as experiments/dkv3/gen_step_asm.py: 8-slot A-operand ring filled LOOKAHEAD MFMAs ahead, <= CAP issue units of vector work per
MFMA gap (a transcendental counts 2), every s_waitcnt lgkmcnt(N) from a simulation of the LDS queue over two iterations, hazard checks.
Operands: %0-%3 / %4-%7 O^T accumulators of group 0 / 1 ("+a"), %8-%11 outputs: l and the running maximum of group 0, of group 1 ("=v"),
%12-%19 / %20-%27 Q fragments ("a"), %28 row-read, %29 transposed-read lane offsets ("v"), %30 scale*log2(e), %31 iterations ("s");
fwd_step_dma_asm.inc (tiles streamed through the ring by LDS-DMA) also: %32 / %33 lane offsets of a wave's K / V tile pieces ("v"), %34 wave,
%35 bytes between a wave's pieces (16 rows) ("s"), %36-%39 first K / V tile rows of the wave's first piece (low / high words, "v")."""
import os
import sys

LOOKAHEAD, CAP, CAP_MASKED = 6, int(os.environ.get("FWD_CAP", "5")), 8
# round-4 variants of the row-sum / running-maximum work (timing experiments; results of l / max then differ from MODE 0's)
LSUM = os.environ.get("FWD_LSUM", "pk")          # pk: v_pk_add_f32 (shipped) | add: two v_add_f32 | none
NO_M3 = os.environ.get("FWD_NO_M3", "0") == "1"   # drop the running maximum
EXP_COST = int(os.environ.get("FWD_EXP_COST", "2"))
DMA_M0 = os.environ.get("FWD_DMA_M0", "each")
DMA_REQ = os.environ.get("FWD_DMA_REQ", "1") == "1"
DMA_BAR = os.environ.get("FWD_DMA_BAR", "1") == "1"
DMA_COST = int(os.environ.get("FWD_DMA_COST", "0"))
DMA_GAPS = [int(x) for x in os.environ.get("FWD_DMA_GAPS", "0,1,2,3,4,5,6,7").split(",")]
VARIANT = any(k.startswith("FWD_") for k in os.environ) or any(k in os.environ for k in ("FWD_CAP", "FWD_LSUM", "FWD_NO_M3", "FWD_EXP_COST"))
XS = {(0, 0): 64, (0, 1): 80, (1, 0): 96, (1, 1): 112}      # score tiles [group][key half]: 16 registers each
PB = {(0, 0): 128, (0, 1): 136, (1, 0): 144, (1, 1): 152}   # packed P: 8 registers each
L2, MX, MREF = {0: 160, 1: 164}, {0: 168, 1: 169}, {0: 170, 1: 171}      # l: two pairs per group (alternating), running max, reference
RING = 176
KRE, KRO, VC0, VC1 = 208, 209, 210, 211
S_T, S_CNT, S_TMP, S_TOFFK, S_TOFFV = "s90", "s91", "s92", "s93", "s94"
K_LDS, V_LDS, NTILE = 0, 65536, 4


def vr(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1) if n > 1 else "v%d" % lo


def regs(lo, n):
    return ["v%d" % i for i in range(lo, lo + n)]


class Ins:
    def __init__(self, text, kind, reads=(), writes=(), lds_defs=None, cost=0):
        self.text, self.kind, self.reads, self.writes, self.lds_defs, self.cost = text, kind, set(reads), set(writes), lds_defs, cost


def mfma_list():
    out = []
    for kh in (0, 1):
        for j in range(8):                       # O(t), key half kh: A = V^T(kh, k16 = j / 4, dt = j % 4)
            for g in (0, 1):
                out.append(dict(prod="O", g=g, kh=kh, j=j, a=("col", kh, j)))
        for ks in range(8):                      # S(t+1), key half kh
            for g in (0, 1):
                out.append(dict(prod="S", g=g, kh=kh, ks=ks, a=("row", kh, ks)))
    return out


def a_loads(desc, slot):
    kind, kh, j = desc
    base = RING + 4 * slot
    if kind == "row":
        addr = (KRE, KRO)[j & 1]
        return [Ins("ds_read_b128 %s, v%d offset:%d" % (vr(base, 4), addr, 8192 * kh + 512 * (j >> 1)), "lds", reads=["v%d" % addr], writes=regs(base, 4), lds_defs=regs(base, 4))]
    k16, dt = j // 4, j % 4
    o0 = 2048 * (4 * kh + 2 * k16) + 512 * dt
    return [Ins("ds_read_b64_tr_b16 %s, v%d offset:%d" % (vr(base, 2), VC0, o0), "lds", reads=["v%d" % VC0], writes=regs(base, 2), lds_defs=regs(base, 2)),
            Ins("ds_read_b64_tr_b16 %s, v%d offset:%d" % (vr(base + 2, 2), VC1, o0 + 2048), "lds", reads=["v%d" % VC1], writes=regs(base + 2, 2), lds_defs=regs(base + 2, 2))]


def mfma_ins(n, m):
    slot = RING + 4 * ((n // 2) % 8)
    a = vr(slot, 4)
    if m["prod"] == "S":
        x = XS[(m["g"], m["kh"])]
        d = vr(x, 16)
        return Ins("v_mfma_f32_32x32x16_bf16 %s, %s, %%%d, %s" % (d, a, 12 + 8 * m["g"] + m["ks"], "0" if m["ks"] == 0 else d), "mfma",
                   reads=regs(slot, 4) + (regs(x, 16) if m["ks"] else []), writes=regs(x, 16))
    k16, dt = m["j"] // 4, m["j"] % 4
    b = PB[(m["g"], m["kh"])] + 4 * k16
    acc = 4 * m["g"] + dt
    return Ins("v_mfma_f32_32x32x16_bf16 %%%d, %s, %s, %%%d" % (acc, a, vr(b, 4), acc), "mfma", reads=regs(slot, 4) + regs(b, 4))


def valu_ops(g, kh):
    """ordered vector work of one score tile (16 registers): running maximum, P = exp2(S sc - m_ref) in place, l += P, bf16 packs"""
    x, pb, l2, mx, mref = XS[(g, kh)], PB[(g, kh)], L2[g], MX[g], MREF[g]
    M3 = lambda i: Ins("v_max3_f32 v%d, v%d, v%d, v%d" % (mx, x + 2 * i, x + 2 * i + 1, mx), "valu", reads=["v%d" % (x + 2 * i), "v%d" % (x + 2 * i + 1), "v%d" % mx], writes=["v%d" % mx], cost=1)
    A = lambda r: Ins("v_fma_f32 v%d, v%d, %%30, -v%d" % (x + r, x + r, mref), "valu", reads=["v%d" % (x + r), "v%d" % mref], writes=["v%d" % (x + r)], cost=1)
    B = lambda r: Ins("v_exp_f32_e32 v%d, v%d" % (x + r, x + r), "trans", reads=["v%d" % (x + r)], writes=["v%d" % (x + r)], cost=EXP_COST)
    La = lambda i, j: Ins("v_add_f32_e32 v%d, v%d, v%d" % (l2 + 2 * (i & 1) + j, l2 + 2 * (i & 1) + j, x + 2 * i + j), "valu",
                          reads=["v%d" % (l2 + 2 * (i & 1) + j), "v%d" % (x + 2 * i + j)], writes=["v%d" % (l2 + 2 * (i & 1) + j)], cost=1)
    Ls = lambda i: Ins("v_pk_add_f32 %s, %s, %s" % (vr(l2 + 2 * (i & 1), 2), vr(l2 + 2 * (i & 1), 2), vr(x + 2 * i, 2)), "valu",
                       reads=regs(l2 + 2 * (i & 1), 2) + regs(x + 2 * i, 2), writes=regs(l2 + 2 * (i & 1), 2), cost=1)
    Dp = lambda i: Ins("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (pb + i, x + 2 * i, x + 2 * i + 1), "valu", reads=regs(x + 2 * i, 2), writes=["v%d" % (pb + i)], cost=1)
    o = []
    if MASKED:      # score register r of key half kh holds key 32 kh + (r & 3) + 8 (r >> 2) (+ 4 h, folded into RANGE): visible iff that is < RANGE
        for r in range(16):
            key = 32 * kh + (r & 3) + 8 * (r >> 2)
            # (ONE unit: the pair shares VCC, and the two groups' vector work is interleaved gap by gap)
            o += [Ins("v_cmp_lt_i32_e32 vcc, %d, v%d\\n\\tv_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (key, RANGE[g], x + r, V_NINF, x + r), "valu",
                      reads=["v%d" % RANGE[g], "v%d" % (x + r), "v%d" % V_NINF], writes=["v%d" % (x + r)], cost=2)]
    lsum = lambda i: {"pk": [Ls(i)], "add": [La(i, 0), La(i, 1)], "none": []}[LSUM]
    if not NO_M3:
        o += [M3(i) for i in range(8)]
    o += [A(0), A(1), A(2), A(3)]
    for r in range(12):
        o += [B(r), A(r + 4)]
        if r % 2 == 1 and r >= 3:
            i = (r - 3) // 2
            o += lsum(i) + [Dp(i)]
    o += [B(12), B(13), B(14), B(15)]
    o += lsum(5) + [Dp(5)] + lsum(6) + [Dp(6)] + lsum(7) + [Dp(7)]
    return o


def addr_update(which):
    if which == "v":      # V tile of the O products: the slot of step t (this iteration's S tile = next iteration's O tile)
        return [Ins("s_mov_b32 %s, %s" % (S_TOFFV, S_TOFFK), "salu"),
                Ins("v_add_u32_e32 v%d, %s, %%29" % (VC0, S_TOFFV), "valu", writes=["v%d" % VC0], cost=1),
                Ins("v_add_u32_e32 v%d, %d, v%d" % (VC0, V_LDS, VC0), "valu", reads=["v%d" % VC0], writes=["v%d" % VC0], cost=1),
                Ins("v_xor_b32_e32 v%d, 32, v%d" % (VC1, VC0), "valu", reads=["v%d" % VC0], writes=["v%d" % VC1], cost=1)]
    return [Ins("s_add_u32 %s, %s, 1" % (S_T, S_T), "salu"), Ins("s_and_b32 %s, %s, %d" % (S_TMP, S_T, NTILE - 1), "salu"), Ins("s_lshl_b32 %s, %s, 14" % (S_TOFFK, S_TMP), "salu"),      # (S_T = tile of the next S chains)
            Ins("v_add_u32_e32 v%d, %s, %%28" % (KRE, S_TOFFK), "valu", writes=["v%d" % KRE], cost=1),
            Ins("v_xor_b32_e32 v%d, 32, v%d" % (KRO, KRE), "valu", reads=["v%d" % KRE], writes=["v%d" % KRO], cost=1)]


DMA = False                   # fwd_step_dma_asm.inc: K / V tiles streamed from global memory through the ring of four LDS slots
MASKED = False                # fwd_step_masked_asm.inc: every score tested against the lane's visible-key count (diagonal / tail / branch-edge tiles)
RANGE = {0: 172, 1: 173}      # per group: keys of the CURRENT S tile this lane's query may see, minus 4 h (the lane half's row offset); -= 64 per step
V_NINF = 174
S_SLOT, S_M0SAVE, S_DST = "s95", "s96", "s97"
KP, VP = (98, 99), (100, 101)      # next tile to request (this wave's first piece)


def dma_groups():
    """Iteration t runs O(t-1) on V tile t-1 and S(t) on K tile t.  At its head it requests tile t+2 (K and V: 4 + 4 one-KiB pieces per wave) into
    the slot tile t-2 left at the last barrier (its V half was read by iteration t-1); the K half is needed two iterations later."""
    pre = ["s_add_u32 %s, %s, 2" % (S_DST, S_SLOT), "s_and_b32 %s, %s, 3" % (S_DST, S_DST), "s_lshl_b32 %s, %s, 14" % (S_DST, S_DST),
           "s_lshl_b32 %s, %%34, 10" % S_TMP, "s_add_u32 %s, %s, %s" % (S_DST, S_DST, S_TMP)]
    groups = []
    for ptr, base, voff in ((KP, K_LDS, "%32"), (VP, V_LDS, "%33")):
        for i in range(4):
            m0 = ["s_add_u32 m0, %s, %d" % (S_DST, base + 4096 * i), "s_nop 0"] if (i == 0 or DMA_M0 != "once") else []      # (once: a timing probe, the pieces overwrite each other)
            ld = ["global_load_lds_dwordx4 %s, s[%d:%d]" % (voff, ptr[0], ptr[1])] if DMA_REQ else []
            groups.append((pre if not groups else []) + m0 + ld + ["s_add_u32 s%d, s%d, %%35" % (ptr[0], ptr[0]), "s_addc_u32 s%d, s%d, 0" % (ptr[1], ptr[1])])
    return groups


def build_body():
    M = mfma_list()
    gaps = [[] for _ in range(64)]
    used = [0] * 64
    if DMA:
        for k, grp in enumerate(dma_groups()):
            gaps[DMA_GAPS[k]] += [Ins(t, "raw") for t in grp]
            used[DMA_GAPS[k]] += DMA_COST
        # tile t+1 (its K half is read from gap 10 of the next iteration on) has landed for every wave: only tile t+2's 8 requests may stay in flight
        gaps[57] += [Ins(t, "raw") for t in ((("s_waitcnt vmcnt(8)",) if DMA_REQ else ()) + (("s_barrier",) if DMA_BAR else ()) + ("s_add_u32 %s, %s, 1" % (S_SLOT, S_SLOT), "s_and_b32 %s, %s, 3" % (S_SLOT, S_SLOT)))]
    # A operands: pair m = MFMAs 2m, 2m+1; its read(s) go out in gap 2m - LOOKAHEAD (of the previous iteration for the first pairs)
    for m in range(32):
        g = (2 * m - LOOKAHEAD) % 64
        gaps[g] += a_loads(M[2 * m]["a"], m % 8)
    # address updates: V after the last V read of the iteration (pair 23: MFMA 46, gap 40), K after the last K read (pair 31: gap 56)
    gaps[41] += addr_update("v"); used[41] += 3
    gaps[57] += addr_update("k"); used[57] += 2
    if MASKED:      # the next S tile lies 64 keys further on
        gaps[31] += [Ins("v_subrev_u32_e32 v%d, 64, v%d" % (RANGE[g], RANGE[g]), "valu", reads=["v%d" % RANGE[g]], writes=["v%d" % RANGE[g]], cost=1) for g in (0, 1)]
        used[31] += 2
    # vector work: key half 0 (chains end at MFMA 30 / 31) in gaps 33..61; key half 1 (chains of the PREVIOUS iteration, end 62 / 63) in gaps 1..29
    for kh, first, last in ((1, 1, 30), (0, 33, 62)):
        for grp in (0, 1):
            g = first + grp      # (group 1's chain ends one MFMA later)
            for ins in valu_ops(grp, kh):
                while used[g] + ins.cost > (CAP_MASKED if MASKED else CAP):
                    g += 1
                assert g <= last, "vector work of key half %d does not fit its window" % kh
                gaps[g].append(ins)
                used[g] += ins.cost
    return M, gaps, used


def linearize(M, gaps):
    seq = []
    for n in range(64):
        seq.append(mfma_ins(n, M[n]))
        seq += gaps[n]
    return seq


def insert_waits(seq, carried):
    fifo, pending, lines, prev = list(carried), {}, [], None
    for e in fifo:
        for r in e["defs"]:
            pending[r] = e
    for ins in seq:
        need = [pending[r] for r in (ins.reads | ins.writes) if r in pending]
        if need:
            last = max(fifo.index(e) for e in need)
            cnt = len(fifo) - 1 - last
            assert cnt <= 15
            lines.append("s_waitcnt lgkmcnt(%d)" % cnt)
            for e in fifo[:last + 1]:
                for r in e["defs"]:
                    if pending.get(r) is e:
                        del pending[r]
            fifo = fifo[last + 1:]
        if prev is not None and prev.kind == "trans" and ins.kind in ("valu", "trans", "mfma") and (prev.writes & ins.reads):
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
    last_writer = {}
    for i, s in enumerate(seq):
        if s.kind in ("valu", "trans"):
            for r in s.reads:
                if r in last_writer and last_writer[r][0] == "mfma":
                    assert sum(1 for p in pos_mfma if last_writer[r][1] < p < i) >= 2, "%s reads %s too close behind its MFMA chain" % (s.text, r)
        if s.kind == "mfma":
            for r in s.reads:
                if r in last_writer and last_writer[r][0] in ("valu", "trans"):
                    assert i - last_writer[r][1] >= 4, "%s reads %s right behind the vector write" % (s.text, r)
            for r in s.writes:      # a score tile must have been consumed (packed) before the next S chain overwrites it: checked by order of the packs
                pass
        for r in s.writes:
            last_writer[r] = (s.kind, i)


def main():
    global DMA, MASKED
    for DMA, MASKED in ((False, False), (True, False), (False, True)):
        try:
            emit()
        except AssertionError as e:      # (a timing variant of the plain step need not fit the other two streams)
            if not (DMA or MASKED) or not VARIANT:
                raise
            print("skipped (DMA %s, MASKED %s): %s" % (DMA, MASKED, e))


def emit():
    M, gaps, used = build_body()
    seq = linearize(M, gaps)
    check(seq + seq)
    carried = []
    for m in range(32):
        if 2 * m - LOOKAHEAD < 0:
            for l in a_loads(M[2 * m]["a"], m % 8):
                carried.append({"defs": set(l.lds_defs)})
    lines1, fifo1 = insert_waits(seq, carried)
    lines2, fifo2 = insert_waits(seq, fifo1)
    assert lines1 == lines2, "loop is not in steady state"
    pro = []
    if DMA:      # tiles 0, 1 up front (the caller has filled nothing), then everything has to land; the loop's first requests are for tile 2
        pro += ["s_mov_b32 %s, m0" % S_M0SAVE, "s_mov_b32 %s, 0" % S_SLOT, "v_readfirstlane_b32 s%d, %%36" % KP[0], "v_readfirstlane_b32 s%d, %%37" % KP[1],
                "v_readfirstlane_b32 s%d, %%38" % VP[0], "v_readfirstlane_b32 s%d, %%39" % VP[1], "s_lshl_b32 %s, %%34, 10" % S_TMP, "s_nop 3"]
        for i in range(2):
            for ptr, base, voff in ((KP, K_LDS, "%32"), (VP, V_LDS, "%33")):
                for k in range(4):
                    pro += ["s_add_u32 m0, %s, %d" % (S_TMP, (i << 14) + base + 4096 * k), "s_nop 0", "global_load_lds_dwordx4 %s, s[%d:%d]" % (voff, ptr[0], ptr[1]),
                            "s_add_u32 s%d, s%d, %%35" % (ptr[0], ptr[0]), "s_addc_u32 s%d, s%d, 0" % (ptr[1], ptr[1])]
        pro += ["s_waitcnt vmcnt(0)", "s_barrier"]
    pro += ["s_mov_b32 %s, 0" % S_T, "s_mov_b32 %s, 0" % S_TOFFK, "s_mov_b32 %s, 0" % S_TOFFV, "s_mov_b32 %s, %%31" % S_CNT,
           "v_add_u32_e32 v%d, %s, %%28" % (KRE, S_TOFFK), "v_xor_b32_e32 v%d, 32, v%d" % (KRO, KRE),
           "v_add_u32_e32 v%d, %d, %%29" % (VC0, V_LDS), "v_xor_b32_e32 v%d, 32, v%d" % (VC1, VC0)]
    # P of "tile -1" is zero (the first O products add nothing); the score tiles of key half 1 start at -inf (their vector work runs first)
    for key in PB:
        pro += ["v_mov_b32_e32 v%d, 0" % (PB[key] + i) for i in range(8)]
    for g in (0, 1):
        pro += ["v_mov_b32_e32 v%d, 0xff800000" % (XS[(g, 1)] + r) for r in range(16)]
        pro += ["v_mov_b32_e32 v%d, 0" % (L2[g] + i) for i in range(4)] + ["v_mov_b32_e32 v%d, 0xff800000" % MX[g], "v_mov_b32_e32 v%d, 0" % MREF[g]]
    if MASKED:
        pro += ["v_mov_b32_e32 v%d, %%32" % RANGE[0], "v_mov_b32_e32 v%d, %%33" % RANGE[1], "v_mov_b32_e32 v%d, 0xff800000" % V_NINF]
    for m in range(32):
        if 2 * m - LOOKAHEAD < 0:
            pro += [l.text for l in a_loads(M[2 * m]["a"], m % 8)]
    body = ["1:"] + lines1 + ["s_sub_u32 %s, %s, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 %s, 0" % S_CNT, "s_cbranch_scc1 1b", "s_waitcnt lgkmcnt(0)"]
    if DMA:
        body += ["s_waitcnt vmcnt(0)", "s_mov_b32 m0, %s" % S_M0SAVE]
    epi = []
    for g in (0, 1):
        epi += ["v_pk_add_f32 %s, %s, %s" % (vr(L2[g], 2), vr(L2[g], 2), vr(L2[g] + 2, 2)), "v_add_f32_e32 %%%d, v%d, v%d" % (8 + 2 * g, L2[g], L2[g] + 1),
                "v_mov_b32_e32 %%%d, v%d" % (9 + 2 * g, MX[g])]
    out = "fwd_step_dma_asm.inc" if DMA else ("fwd_step_masked_asm.inc" if MASKED else "fwd_step_asm.inc")
    with open(out, "w") as f:
        f.write("// generated by gen_fwd_step.py - do not edit\n")
        for l in pro + body + epi:
            f.write('"%s\\n\\t"\n' % l)
    nv = sum(1 for s in seq if s.kind in ("valu", "trans"))
    nl = sum(1 for s in seq if s.kind == "lds")
    print("step: 64 MFMAs, %d vector (%d issue units), %d LDS reads, %d lines; busiest gap %d units" % (nv, sum(used), nl, len(lines1), max(used)))
    with open(out.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write(", ".join('"v%d"' % i for i in range(64, 212)) + ', "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "scc", "memory"\n')


if __name__ == "__main__":
    main()
