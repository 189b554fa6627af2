#!/usr/bin/env python3
"""Generates sdpa_fwd3_loop.inc: ONE row block (item) of the causal forward, sdpa_fwd3 (sdpa_fwd3.h), as one inline-asm block.

A wave is alone on its SIMD (512 registers) and owns 64 QUERIES - two groups g of 32: their Q fragments (B operands, a[128:191]) and their
O^T accumulators (a[0:127]) - and per 64-key tile t of the workgroup's K / V ring computes
    S^T_g[key][query] = K Q_g^T      A = K tile rows from LDS (one A operand serves both groups), B = Q fragments; lane = query, registers = keys:
                                      a query's row statistics are per LANE, no cross-lane work inside the loop
    P = exp2(S^T sc - m_ref)         in place; m_ref is a per-query REFERENCE fixed for the whole pass (set in the prologue from the first half
                                      tile's row maximum, or handed in); nothing is tracked in the loop - the row SUMS are judged at the end of
                                      the pass and the item is repeated here, reference raised by 120, while one is not below 2^100 (MAX_REDO)
    O^T_g[d][query] += V^T P_g       A = V^T by transposed LDS reads (shared by both groups again), B = P packed to bf16
= 64 MFMAs per tile.  The stream is the one measured in experiments/fwd3 (2 804 cycles per step; 3 020-3 042 with the tiles streamed by LDS-DMA;
masked variant 3 339), ROTATED by half a step so that ALL vector work of an iteration belongs to ONE tile (its mask state is per iteration):
    n =  0..15  O(t-1) keys 32..63        gaps  1..30: vector work of S(t) keys  0..31 (chain: n = 48..63 of the previous iteration)
    n = 16..31  S(t)   keys 32..63        gaps 33..62: vector work of S(t) keys 32..63
    n = 32..47  O(t)   keys  0..31
    n = 48..63  S(t+1) keys  0..31
K / V tiles: a ring of FOUR slots (K at 0, V at 64 KiB, 16 KiB per tile) fed by LDS-DMA: the item's prologue requests tiles 0..2, iteration t waits
for tile t+1 (ONE counted vmcnt + s_barrier, gap 25) and then requests tile t+3 into the slot tile t-1 left (gaps 32..39); requests past the
item's last tile still go out (rows past the sequence's end of the bounds-checked descriptor: zeros) so that every iteration has the same eight
vector-memory operations.  The tile WALK may
jump once (a row block wholly inside branch B of a packed row skips the tiles of [br.a, br.b)); the sequence's partial last tile needs no case of its
own (rows past the descriptor arrive as zeros).
Iterations come in two bodies: PLAIN (every key of the tile visible to every query of the wave) and MASKED (every score compared with the lane's
visible-key count: causal diagonal, sequence tail, branch edge, wholly hidden tiles); an item is [plain n0][masked n1, rsA][plain n2][masked n3, rsB].
Machinery as gen_dkv3_loop.py: 8-slot A-operand ring filled LOOKAHEAD MFMAs ahead, <= CAP issue units of vector work per MFMA gap, every
s_waitcnt lgkmcnt(N) from a simulation of the in-order LDS queue, hazard checks.

Operands (by NAME; the authoritative list is the asm statement in sdpa_fwd3_call.h): q0-q15 the Q fragments ("+a", fixed a[128:191]: this item's
on entry - possibly still in flight, the block's first wait covers them - the NEXT item's, landed, on exit); the O^T accumulators a[0:127] are the
block's own (clobbered: it zeroes them, normalises and stores the item's rows and lse itself - o_lo / o_hi, lse_lo / lse_hi, ooffc, rows8o, loff0/1,
nt01); rowrel / colrel row-read / transposed-read lane offsets, voff the lane offset of a wave's tile piece ("v"); rsA, rsB0/1 per-lane visible-key
counts (minus 4 h) of the first tile of masked run 1 (the same for both groups: br.a) / of masked run 2; k_lo / k_hi / nrec / soff0 the buffer
descriptor (base, bytes) over the sequence's K rows of this head and the first tile's byte offset (uniform values in VECTOR registers: scalar
operands are scarce), vdlo = v - k in bytes ("s"); nk_* / nnrec / nsoff0 the same for the NEXT item (its first tiles are requested by this item's last
iterations), nq_* / nqrec / nqsoff0/1 / nqvA/B / rows8 / nqg0/1 its Q rows (requested by the tail); sc = scale * log2 e, n01 = n0 | n1 << 16, n23,
nreq = tiles to request | requests before the walk's jump << 16 (0xffff: no jump), jlo the jump in bytes, wave, piece = bytes between a wave's
pieces (16 rows), ctl: bits 0-1 ring slot of tile 0, bit 4 tiles 0..2 were requested by the previous item's block, bits 5-6 tiles of the NEXT item
that this block's last iterations request once its own are all under way, bits 8-9 the first tile's mask state (0 plain, 1 masked run 1, 2 masked
run 2), bit 10 which of the two vote-word sets this item uses, bit 11 the next item's Q rows are gathered by plain loads instead of LDS-DMA ("s")."""

import os
import re
import sys

# Issue units per MFMA gap (one unit ~ 4 cycles; a v_exp_f32 is two).  MI355X_MICROARCH.md prices a gap at 32 cycles = the MFMA's own 8 + 24 of
# other issue, i.e. 6 units - and packed f32 VALU beside MFMAs far above its slot: the row sums as v_pk_add_f32 at CAP 5 ran 2 804 cycles per
# step in experiments/fwd3, as two v_add_f32 at CAP 6 2 339 (round 4; build_variants.sh there).  FWD3_CAP / FWD3_CAP_MASKED / FWD3_LSUM: experiments.
LOOKAHEAD, CAP, CAP_MASKED = 6, int(os.environ.get("FWD3_CAP", "6")), int(os.environ.get("FWD3_CAP_MASKED", "8"))
LSUM = os.environ.get("FWD3_LSUM", "add")
# The running row maximum (one v_max3_f32 per two scores) is NOT tracked: every instruction of the single wave costs an issue slot of ~4 cycles
# whatever its unit, and the check it fed - "has a score outgrown the row's exponent reference" - is made on the row SUMS at the end of the pass
# instead: a partial sum that is not < 2^100 (P overflowed, or is about to) repeats the row block with that row's reference raised by 120
# (log2 units), as often as it takes.  Any reference within ~100 of the row's true maximum gives the same result: P, l and O^T are floating
# point numbers, only their common exponent moves.  FWD3_TRACK_MAX=1 puts the v_max3 back (timing experiments; the value is unused).
TRACK_MAX = os.environ.get("FWD3_TRACK_MAX", "0") == "1"
O_STORE_POLICY = " nt" if os.environ.get("FWD3_O_NT", "0") == "1" else ""      # (experiment, round 5: the output rows written nontemporally - +0.6 %, not kept)
WAIT_AGE = int(os.environ.get("FWD3_WAIT_AGE", "4"))      # 0: one wait per first use
# 2^100, 120.0, repeats at most: 64 x 120 log2 units = scores 5 300 nats above the row's first keys (round 5; 8 before: round-4 advice).  A bound there must
# be: NaN / inf scores overflow on every pass.  Rows beyond it come back as NaN (l = inf) - loudly, not as wrong numbers - unless the launch is followed
# by the running-maximum kernel in repair mode (sdpa.hip: SdpaParams::repair, HALVA_FWD3_REPAIR=1).
REDO_LIMIT, REDO_STEP, MAX_REDO = 0x71800000, 0x42f00000, 64
XS = {(0, 0): 64, (0, 1): 80, (1, 0): 96, (1, 1): 112}      # score tiles [group][key half]: 16 registers each
PB = {(0, 0): 128, (0, 1): 136, (1, 0): 144, (1, 1): 152}   # packed P: 8 registers each
L2, MX, MREF, RANGE = {0: 160, 1: 164}, {0: 168, 1: 169}, {0: 170, 1: 171}, {0: 172, 1: 173}
V_NINF, V_T0, V_T1 = 174, 175, 212
RING = 176
KRE, KRO, VC0, VC1 = 208, 209, 210, 211
V_LAST = 215
S_T, S_CNT, S_TMP, S_TMP2, S_TOFFK, S_TOFFV, S_M0SAVE, S_DST = "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77"
S_RLEFT, S_TOJUMP, S_SEG, S_FLD, S_NPF = "s78", "s79", "s82", "s83", "s85"
S_SOFFK, S_SOFFV, S_SOFF0 = "s86", "s87", "s92"
DESC, DESC0, SRC = (88, 91), (96, 99), (94, 95)      # the K / V buffer descriptor of the walk (four SGPRs, aligned), the item's own (kept for the repeat), a scratch pair
S_REDO, S_PF = "s66", "s67"
S_FIRST, S_LAST = 70, 99
K_LDS, V_LDS, MAIL_LDS = 0, 65536, 131072      # MAIL_LDS = FWD3_MAIL (sdpa_fwd3.h)
WAIT_GAP, V_UPD_GAP = 25, 9
DMA_PRE_GAPS, DMA_GAPS = [26, 27, 28, 29, 30, 31], [32, 33, 34, 35, 36, 37, 38, 39]
MASKED, PHASE = False, "p"


def vr(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1) if n > 1 else "v%d" % lo


def sp(pair):
    return "s[%d:%d]" % pair


def regs(lo, n):
    return ["v%d" % i for i in range(lo, lo + n)]


class Ins:
    def __init__(self, text, kind, reads=(), writes=(), lds_defs=None, cost=0):
        self.text, self.kind, self.reads, self.writes, self.lds_defs, self.cost = text, kind, set(reads), set(writes), lds_defs, cost


def raw(t):
    return Ins(t, "raw")


def mfma_list():
    out = []
    for j in range(8):                       # O(t-1), keys 32..63
        for g in (0, 1):
            out.append(dict(prod="O", g=g, kh=1, j=j, a=("col", 1, j)))
    for ks in range(8):                      # S(t), keys 32..63
        for g in (0, 1):
            out.append(dict(prod="S", g=g, kh=1, ks=ks, a=("row", 1, ks)))
    for j in range(8):                       # O(t), keys 0..31
        for g in (0, 1):
            out.append(dict(prod="O", g=g, kh=0, j=j, a=("col", 0, j)))
    for ks in range(8):                      # S(t+1), keys 0..31
        for g in (0, 1):
            out.append(dict(prod="S", g=g, kh=0, ks=ks, a=("row", 0, ks)))
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
        return Ins("v_mfma_f32_32x32x16_bf16 %s, %s, %%[q%d], %s" % (d, a, 8 * m["g"] + m["ks"], "0" if m["ks"] == 0 else d), "mfma",
                   reads=regs(slot, 4) + (regs(x, 16) if m["ks"] else []), writes=regs(x, 16))
    k16 = m["j"] // 4
    dt = m["j"] % 4
    b = PB[(m["g"], m["kh"])] + 4 * k16
    acc = 4 * m["g"] + dt
    return Ins("v_mfma_f32_32x32x16_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (16 * acc, 16 * acc + 15, a, vr(b, 4), 16 * acc, 16 * acc + 15), "mfma", reads=regs(slot, 4) + regs(b, 4))


def mask_ops(g, kh, rng):
    """score register r of key half kh holds key 32 kh + (r & 3) + 8 (r >> 2) (+ 4 h, folded into the range): visible iff that is < range"""
    x = XS[(g, kh)]
    o = []
    for r in range(16):
        key = 32 * kh + (r & 3) + 8 * (r >> 2)
        o.append(Ins("v_cmp_lt_i32_e32 vcc, %d, v%d\\n\\tv_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (key, rng, x + r, V_NINF, x + r), "valu",
                     reads=["v%d" % rng, "v%d" % (x + r), "v%d" % V_NINF], writes=["v%d" % (x + r)], cost=2))
    return o


def valu_ops(g, kh):
    """ordered vector work of one score tile (16 registers): (mask,) running maximum, P = exp2(S sc - m_ref) in place, l += P, bf16 packs"""
    x, pb, l2, mx, mref = XS[(g, kh)], PB[(g, kh)], L2[g], MX[g], MREF[g]
    M3 = lambda i: Ins("v_max3_f32 v%d, v%d, v%d, v%d" % (mx, x + 2 * i, x + 2 * i + 1, mx), "valu", reads=["v%d" % (x + 2 * i), "v%d" % (x + 2 * i + 1), "v%d" % mx], writes=["v%d" % mx], cost=1)
    A = lambda r: Ins("v_fma_f32 v%d, v%d, %%[sc], -v%d" % (x + r, x + r, mref), "valu", reads=["v%d" % (x + r), "v%d" % mref], writes=["v%d" % (x + r)], cost=1)
    B = lambda r: Ins("v_exp_f32_e32 v%d, v%d" % (x + r, x + r), "trans", reads=["v%d" % (x + r)], writes=["v%d" % (x + r)], cost=2)
    Lp = lambda i: Ins("v_pk_add_f32 %s, %s, %s" % (vr(l2 + 2 * (i & 1), 2), vr(l2 + 2 * (i & 1), 2), vr(x + 2 * i, 2)), "valu",
                       reads=regs(l2 + 2 * (i & 1), 2) + regs(x + 2 * i, 2), writes=regs(l2 + 2 * (i & 1), 2), cost=1)
    La = lambda i, j: Ins("v_add_f32_e32 v%d, v%d, v%d" % (l2 + 2 * (i & 1) + j, l2 + 2 * (i & 1) + j, x + 2 * i + j), "valu",
                          reads=["v%d" % (l2 + 2 * (i & 1) + j), "v%d" % (x + 2 * i + j)], writes=["v%d" % (l2 + 2 * (i & 1) + j)], cost=1)
    Ls = lambda i: [Lp(i)] if LSUM == "pk" else [La(i, 0), La(i, 1)]
    Dp = lambda i: Ins("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (pb + i, x + 2 * i, x + 2 * i + 1), "valu", reads=regs(x + 2 * i, 2), writes=["v%d" % (pb + i)], cost=1)
    o = []
    if MASKED:
        o += mask_ops(g, kh, RANGE[g])
    if TRACK_MAX:
        o += [M3(i) for i in range(8)]
    o += [A(0), A(1), A(2), A(3)]
    for r in range(12):
        o += [B(r), A(r + 4)]
        if r % 2 == 1 and r >= 3:
            i = (r - 3) // 2
            o += Ls(i) + [Dp(i)]
    o += [B(12), B(13), B(14), B(15)]
    o += Ls(5) + [Dp(5)] + Ls(6) + [Dp(6)] + Ls(7) + [Dp(7)]
    return o


def k_addr_from_toff():
    return [Ins("v_add_u32_e32 v%d, %s, %%[rowrel]" % (KRE, S_TOFFK), "valu", writes=["v%d" % KRE], cost=1),
            Ins("v_xor_b32_e32 v%d, 32, v%d" % (KRO, KRE), "valu", reads=["v%d" % KRE], writes=["v%d" % KRO], cost=1)]


def v_addr_from_toff():
    return [Ins("v_add_u32_e32 v%d, %s, %%[colrel]" % (VC0, S_TOFFV), "valu", writes=["v%d" % VC0], cost=1),
            Ins("v_add_u32_e32 v%d, %d, v%d" % (VC0, V_LDS, VC0), "valu", reads=["v%d" % VC0], writes=["v%d" % VC0], cost=1),
            Ins("v_xor_b32_e32 v%d, 32, v%d" % (VC1, VC0), "valu", reads=["v%d" % VC0], writes=["v%d" % VC1], cost=1)]


def addr_update(which):
    if which == "v":      # V tile of the O products: the tile the S chains have been reading (t)
        return [Ins("s_mov_b32 %s, %s" % (S_TOFFV, S_TOFFK), "salu")] + v_addr_from_toff()
    return [Ins("s_add_u32 %s, %s, 1" % (S_T, S_T), "salu"), Ins("s_and_b32 %s, %s, 3" % (S_TMP, S_T), "salu"), Ins("s_lshl_b32 %s, %s, 14" % (S_TOFFK, S_TMP), "salu")] + k_addr_from_toff()


def sq(quad):
    return "s[%d:%d]" % quad


def request_tile(tag, dst_plus):
    """(pre, [8 piece groups], post, out-of-line) - instruction texts - of ONE tile request: the next tile of the walk into ring slot
    (S_T + dst_plus) & 3, by `buffer_load_dwordx4 ... offen lds` through a descriptor that spans the sequence's K / V rows [start, start + len) of
    this head: a row beyond the sequence arrives as ZEROS (the range check covers the scalar offset - experiments/fwd3/oob_probe.hip), so
    neither the partial last tile nor a request past the walk's end needs a case of its own.  Three instructions per 1-KiB piece."""
    # no tile of this item left: the requests go on with the workgroup's NEXT item's first tiles (S_NPF of them; same slot rotation, so that item
    # simply starts on a rotated ring with its first tiles under way)
    pre = ["s_cmp_eq_u32 %s, 0" % S_RLEFT, "s_cbranch_scc1 .Lf3_sw%s_%%=" % tag, ".Lf3_swb%s_%%=:" % tag,
           "s_add_u32 %s, %s, %d" % (S_DST, S_T, dst_plus), "s_and_b32 %s, %s, 3" % (S_DST, S_DST), "s_lshl_b32 %s, %s, 14" % (S_DST, S_DST),
           "s_lshl_b32 %s, %%[wave], 10" % S_TMP2, "s_add_u32 %s, %s, %s" % (S_DST, S_DST, S_TMP2),
           "s_cmp_eq_u32 %s, 0" % S_TOJUMP, "s_cselect_b32 %s, %%[jlo], 0" % S_TMP, "s_add_u32 %s, %s, %s" % (S_SOFFK, S_SOFFK, S_TMP),      # the walk's jump
           "s_sub_u32 %s, %s, 1" % (S_TOJUMP, S_TOJUMP), "s_sub_u32 %s, %s, 1" % (S_RLEFT, S_RLEFT),
           "s_add_u32 %s, %s, %%[vdlo]" % (S_SOFFV, S_SOFFK),                                                                                  # V rows = K rows + (v - k)
           "s_mov_b32 m0, %s" % S_DST]
    groups = []
    for soff, base in ((S_SOFFK, K_LDS), (S_SOFFV, V_LDS)):
        for i in range(4):
            g = (["s_add_u32 m0, %s, %d" % (S_DST, V_LDS), "s_nop 0"] if (base and i == 0) else [])
            g += ["buffer_load_dwordx4 %%[voff], %s, %s offen lds" % (sq(DESC), soff), "s_add_u32 %s, %s, %%[piece]" % (soff, soff), "s_add_u32 m0, m0, 4096"]
            groups.append(g)
    ool = [".Lf3_sw%s_%%=:" % tag, "s_mov_b32 %s, 0x7fffffff" % S_RLEFT, "s_cmp_eq_u32 %s, 0" % S_NPF, "s_cbranch_scc1 .Lf3_swb%s_%%=" % tag,
           "v_readfirstlane_b32 s%d, %%[nk_lo]" % DESC[0], "v_readfirstlane_b32 s%d, %%[nk_hi]" % (DESC[0] + 1), "v_readfirstlane_b32 s%d, %%[nnrec]" % (DESC[0] + 2),
           "v_readfirstlane_b32 %s, %%[nsoff0]" % S_SOFFK, "s_mov_b32 %s, 0" % S_NPF, "s_mov_b32 %s, -1" % S_TOJUMP, "s_nop 3", "s_branch .Lf3_swb%s_%%=" % tag]
    return pre, groups, [], ool


def build_body():
    M = mfma_list()
    gaps = [[] for _ in range(64)]
    used = [0] * 64
    pre, groups, post, ool = request_tile(PHASE, 2)      # (behind gap 25's increment S_T = tile t+1: tile t+3 goes to slot S_T + 2)
    per = (len(pre) + len(DMA_PRE_GAPS) - 1) // len(DMA_PRE_GAPS)
    for k, gp in enumerate(DMA_PRE_GAPS):
        gaps[gp] += [raw(t) for t in pre[k * per:(k + 1) * per]]
    for k, grp in enumerate(groups):
        gaps[DMA_GAPS[k]] += [raw(t) for t in grp]
    gaps[DMA_GAPS[-1]] += [raw(t) for t in post]
    # tile t+1 (its K rows are read from gap 42 on) has landed for every wave: only tile t+2's 8 requests may stay in flight; behind this
    # barrier every wave has issued its last read of tile t-1 (V rows 32..63, n <= 15)
    gaps[WAIT_GAP] += [raw("s_waitcnt vmcnt(8)"), raw("s_barrier")]
    # A operands: pair m = MFMAs 2m, 2m+1; its read(s) go out in gap 2m - LOOKAHEAD (of the previous iteration for the first pairs)
    for m in range(32):
        g = (2 * m - LOOKAHEAD) % 64
        gaps[g] += a_loads(M[2 * m]["a"], m % 8)
    gaps[V_UPD_GAP] += addr_update("v"); used[V_UPD_GAP] += 3
    gaps[WAIT_GAP] += addr_update("k"); used[WAIT_GAP] += 2
    for kh, first, last in ((0, 1, 30), (1, 33, 62)):
        for grp in (0, 1):
            g = first + grp
            for ins in valu_ops(grp, kh):
                while used[g] + ins.cost > (CAP_MASKED if MASKED else CAP):
                    g += 1
                assert g <= last, "vector work of key half %d does not fit its window" % kh
                gaps[g].append(ins)
                used[g] += ins.cost
    if MASKED:      # the next tile lies 64 keys further on
        gaps[63] += [Ins("v_subrev_u32_e32 v%d, 64, v%d" % (RANGE[g], RANGE[g]), "valu", reads=["v%d" % RANGE[g]], writes=["v%d" % RANGE[g]], cost=1) for g in (0, 1)]
        used[63] += 2
    return M, gaps, used, ool


def linearize(M, gaps):
    seq = []
    for n in range(64):
        seq.append(mfma_ins(n, M[n]))
        seq += gaps[n]
    return seq


def insert_waits(seq, carried):
    """s_waitcnt lgkmcnt(N) in front of the first user of an LDS read, N from the in-order queue.  A wait that is due anyway also covers every
    younger read issued at least WAIT_AGE MFMAs ago (long landed: the wait costs the same issue slot and saves the next one - one wait per two
    MFMA pairs instead of one per pair)."""
    # (ages in MFMAs, relative to this stretch's first: what an iteration leaves in flight was issued 64 MFMAs before the same point of the next)
    fifo, pending, lines, prev, now = [dict(e, age=e["age"] - 64 if e.get("age", -99) > 0 else e.get("age", -99)) for e in carried], {}, [], None, 0
    for e in fifo:
        for r in e["defs"]:
            pending[r] = e
    for ins in seq:
        if ins.kind == "mfma":
            now += 1
        need = [pending[r] for r in (ins.reads | ins.writes) if r in pending]
        if need:
            last = max(fifo.index(e) for e in need)
            while WAIT_AGE and last + 1 < len(fifo) and now - fifo[last + 1]["age"] >= WAIT_AGE:
                last += 1
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
            e = {"defs": set(ins.lds_defs), "age": now}
            fifo.append(e)
            for r in e["defs"]:
                pending[r] = e
        assert len(fifo) <= 15, "more than 15 LDS reads in flight"
        if ins.kind not in ("salu", "raw"):
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
        for r in s.writes:
            last_writer[r] = (s.kind, i)


def carried_reads(M):
    c = []
    for m in range(32):
        if 2 * m - LOOKAHEAD < 0:
            for l in a_loads(M[2 * m]["a"], m % 8):
                c.append({"defs": set(l.lds_defs), "age": 2 * m - LOOKAHEAD})      # (issued in gap 64 + 2 m - LOOKAHEAD of the iteration before)
    return c


def variant(masked, phase):
    global MASKED, PHASE
    MASKED, PHASE = masked, phase
    M, gaps, used, ool = build_body()
    seq = linearize(M, gaps)
    check(seq + seq)
    lines1, fifo1 = insert_waits(seq, carried_reads(M))
    lines2, fifo2 = insert_waits(seq, fifo1)
    assert lines1 == lines2 and [sorted(e["defs"]) for e in fifo1] == [sorted(e["defs"]) for e in fifo2], "loop is not in steady state"
    # (the plain and the masked body follow each other: both must end with exactly the next iteration's first operands in flight)
    assert [sorted(e["defs"]) for e in fifo1] == [sorted(e["defs"]) for e in carried_reads(M)], "the iteration's last reads are not the next one's first operands"
    assert [sorted(e["defs"]) for e in fifo1] == [sorted(e["defs"]) for e in carried_reads(M)], "the bodies must leave the same LDS queue"
    nv = sum(1 for s in seq if s.kind in ("valu", "trans"))
    nl = sum(1 for s in seq if s.kind == "lds")
    print("body %s (%s): 64 MFMAs, %d vector (%d issue units, busiest gap %d), %d LDS reads, %d asm lines" % (phase, "masked" if masked else "plain", nv, sum(used), max(used), nl, len(lines1)))
    return M, lines1, ool


def chain_block(M, ns, first_reads_issued, carried):
    """a stretch of the body's MFMAs outside the loop (prologue: S(0) keys 0..31 = n 48..63; drain: O(last) keys 32..63 = n 0..15) with its own
    LOOKAHEAD pipeline of A-operand reads; pairs < first_reads_issued have their reads in flight already (`carried`)"""
    pairs = sorted({n // 2 for n in ns})
    seq = []
    la = LOOKAHEAD // 2
    for k in range(first_reads_issued, min(la, len(pairs))):
        seq += a_loads(M[2 * pairs[k]]["a"], pairs[k] % 8)
    for k, m in enumerate(pairs):
        seq.append(mfma_ins(2 * m, M[2 * m]))
        seq.append(mfma_ins(2 * m + 1, M[2 * m + 1]))
        if k + la < len(pairs):
            seq += a_loads(M[2 * pairs[k + la]]["a"], pairs[k + la] % 8)
    lines, fifo = insert_waits(seq, carried)
    return lines, fifo


QDESC, OB, LB = (88, 91), (96, 97), (98, 99)      # (the walk's descriptor and the item's own are dead behind the vote)
S_QST, S_SOFFQ = "s86", "s87"
T0 = 64                                               # temporaries of the tail: v64.. (the loop's registers are dead)


OSTAGE_LDS = MAIL_LDS + 128     # (FWD3_OSTAGE, sdpa_fwd3.h; 128-byte aligned: the staging addresses are formed with XORs) 4 x 4 KiB: a wave's staging area for its output rows (half the head_dim of a row group at a time)
S_OBJ = (92, 93)


def tail_code():
    """Behind the vote.  Per row group g:
      [the NEXT item's Q rows of the group requested by LDS-DMA into this wave's 8 KiB of the ring slot the item's last tile has left - whole 1-KiB
       pieces through a descriptor over the next sequence's Q rows; rows outside it arrive as zeros]
      [this item's rows of the group: O = O^T / l as bf16, TRANSPOSED through 4 KiB of LDS (two passes of 64 head_dim columns) so that every store
       instruction writes eight whole 128-byte row pieces instead of 16 bytes into each of 32 rows - the address unit takes a lane's address per
       clock, and 16 such stores (plus 16 such loads for Q) of four waves were ~8 000 cycles per item; the lse]
      [the fragments read back from LDS straight into the Q registers]."""
    o = []
    l3, l7off = T0 + 32, T0 + 33
    lane = T0 + 7
    # this wave's 8 KiB of the free slot (S_T - 1) & 3 - K area for waves 0 / 1, V area for waves 2 / 3 -, its 4 KiB output staging area
    o += ["s_sub_u32 %s, %s, 1" % (S_TMP, S_T), "s_and_b32 %s, %s, 3" % (S_TMP, S_TMP), "s_lshl_b32 %s, %s, 14" % (S_QST, S_TMP),
          "s_lshr_b32 %s, %%[wave], 1" % S_TMP, "s_lshl_b32 %s, %s, 16" % (S_TMP, S_TMP), "s_add_u32 %s, %s, %s" % (S_QST, S_QST, S_TMP),
          "s_and_b32 %s, %%[wave], 1" % S_TMP, "s_lshl_b32 %s, %s, 13" % (S_TMP, S_TMP), "s_add_u32 %s, %s, %s" % (S_QST, S_QST, S_TMP),
          "v_readfirstlane_b32 s%d, %%[nq_lo]" % QDESC[0], "v_readfirstlane_b32 s%d, %%[nq_hi]" % (QDESC[0] + 1), "v_readfirstlane_b32 s%d, %%[nqrec]" % (QDESC[0] + 2),
          "s_mov_b32 s%d, 0x00020000" % (QDESC[0] + 3),
          "v_readfirstlane_b32 s%d, %%[o_lo]" % OB[0], "v_readfirstlane_b32 s%d, %%[o_hi]" % OB[1],
          "v_readfirstlane_b32 s%d, %%[lse_lo]" % LB[0], "v_readfirstlane_b32 s%d, %%[lse_hi]" % LB[1],
          "v_add_u32_e32 v%d, %s, %%[rowrel]" % (KRE, S_QST), "v_xor_b32_e32 v%d, 32, v%d" % (KRO, KRE),
          "v_mbcnt_lo_u32_b32 v%d, -1, 0" % lane, "v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (lane, lane), "v_lshrrev_b32_e32 v%d, 3, v%d" % (l3, lane), "s_nop 3"]
    # staging addresses.  Write (lane = row r of the group, half h): row r, chunk c (16 bytes) at r * 128 + ((c ^ ((r >> 1) & 7)) << 4) + 8 h - the
    # lane part is wbase = r * 128 + 8 h and the mask m = ((r >> 1) & 7) << 4; a piece's address = wbase + ((c << 4) ^ m)  (c is a constant).
    # Read (lane l): row 8 j + (l >> 3), chunk l & 7 -> rbase = (l >> 3) * 128 + (((l & 7) ^ (((l >> 3) >> 1) & 7)) << 4), + 1024 j  (8 j is even: the mask of row
    # 8 j + x is that of row x).
    wbase, wmask, rbase, rbase_odd, t = T0 + 34, T0 + 35, T0 + 36, T0 + 37, T0 + 2
    o += ["v_and_b32_e32 v%d, 31, v%d" % (t, lane), "v_lshlrev_b32_e32 v%d, 7, v%d" % (wbase, t), "v_lshrrev_b32_e32 v%d, 1, v%d" % (wmask, t),
          "v_and_b32_e32 v%d, 7, v%d" % (wmask, wmask), "v_lshlrev_b32_e32 v%d, 4, v%d" % (wmask, wmask),
          "v_lshrrev_b32_e32 v%d, 5, v%d" % (t, lane), "v_lshlrev_b32_e32 v%d, 3, v%d" % (t, t), "v_add_u32_e32 v%d, v%d, v%d" % (wbase, wbase, t),
          "s_lshl_b32 %s, %%[wave], 12" % S_TMP, "s_add_u32 %s, %s, %d" % (S_TMP, S_TMP, OSTAGE_LDS), "v_add_u32_e32 v%d, %s, v%d" % (wbase, S_TMP, wbase),
          "v_lshrrev_b32_e32 v%d, 1, v%d" % (t, l3), "v_and_b32_e32 v%d, 7, v%d" % (t, t), "v_and_b32_e32 v%d, 7, v%d" % (rbase, lane), "v_xor_b32_e32 v%d, v%d, v%d" % (rbase, rbase, t),
          "v_lshlrev_b32_e32 v%d, 4, v%d" % (rbase, rbase), "v_lshlrev_b32_e32 v%d, 7, v%d" % (t, l3), "v_add_u32_e32 v%d, v%d, v%d" % (rbase, rbase, t),
          "v_add_u32_e32 v%d, %s, v%d" % (rbase, S_TMP, rbase),
          "v_xor_b32_e32 v%d, 64, v%d" % (rbase_odd, rbase)]      # rows 8 j + x with j odd: ((8 j + x) >> 1) & 7 = (x >> 1) ^ 4
    for g in (0, 1):
        # ---- request the group's 32 rows of the NEXT item: piece c = rows 8 (c / 2) .. + 7, chunks 8 (c % 2) .. + 7 of the half-tile image
        # (ctl bit 11: the next row block begins in front of its sequence - left padding: its Q fragments are gathered row by row with clamped
        # pointers behind the stores instead, as the workgroup's first item's are)
        # (Tried in round 4 and NOT kept: the eight requests handed out one at a time between the eight conversion stretches below - the tail grew
        # from 6 330 to 7 870 cycles: a request costs the lone wave its ~85 cycles wherever it is issued, and the last ones then land late.)
        o += ["s_bitcmp1_b32 %[ctl], 11", "s_cbranch_scc1 .Lf3_noqdma%d_%%=" % g]
        o += ["v_readfirstlane_b32 %s, %%[nqsoff%d]" % (S_SOFFQ, g), "s_mov_b32 m0, %s" % S_QST, "s_nop 2"]
        for c in range(8):      # (no immediate offset: an LDS-DMA load adds it to the LDS address as well)
            if c & 1:
                o += ["s_add_u32 %s, %s, 128" % (S_TMP, S_SOFFQ)]
            o += ["buffer_load_dwordx4 %%[nqv%s%d], %s, %s offen lds" % ("AB"[(c >> 1) & 1], g, sq(QDESC), S_TMP if c & 1 else S_SOFFQ),
                  "s_add_u32 m0, m0, 1024", "s_nop 0"]
            if c & 1:
                o += ["s_add_u32 %s, %s, %%[rows8]" % (S_SOFFQ, S_SOFFQ)]
        o += [".Lf3_noqdma%d_%%=:" % g]
        if g == 0:
            o += stamp(1, "tail")
        # ---- this item's rows of the group
        l, lt, lse, fl, loff, inv = T0, T0 + 1, T0 + 4, T0 + 5, T0 + 6, T0 + 30      # (inv: a pair, v_pk_mul's factor)
        W, A, R = T0 + 8, T0 + 12, T0 + 40                                           # 4 packed words; 8 accumulator values; 4 x 4 registers read back
        o += ["v_pk_add_f32 %s, %s, %s" % (vr(L2[g], 2), vr(L2[g], 2), vr(L2[g] + 2, 2)), "v_add_f32_e32 v%d, v%d, v%d" % (l, L2[g], L2[g] + 1),
              "v_mov_b32_e32 v%d, v%d" % (lt, l), "s_nop 1", "v_permlane32_swap_b32_e32 v%d, v%d" % (l, lt), "s_nop 1", "v_add_f32_e32 v%d, v%d, v%d" % (lt, l, lt),      # l of the whole row
              "v_and_b32_e32 v%d, 3, %%[loff%d]" % (fl, g), "v_and_b32_e32 v%d, -4, %%[loff%d]" % (loff, g),
              "v_rcp_f32_e32 v%d, v%d" % (inv, lt), "v_log_f32_e32 v%d, v%d" % (lse, lt),
              "v_cmp_lt_u32_e32 vcc, 1, v%d" % fl, "s_mov_b64 %s, vcc" % sp(SRC),                                       # bit 1: the row is a row of the sequence
              "v_cmp_lt_f32_e32 vcc, 0, v%d" % lt, "s_and_b64 vcc, vcc, %s" % sp(SRC), "v_cndmask_b32_e32 v%d, 0, v%d, vcc" % (inv, inv),
              "v_add_f32_e32 v%d, v%d, v%d" % (lse, lse, MREF[g]), "v_mul_f32_e32 v%d, 0x3f317218, v%d" % (lse, lse),
              "s_mov_b64 vcc, %s" % sp(SRC), "v_cndmask_b32_e32 v%d, 0, v%d, vcc" % (lse, lse),
              "v_mov_b32_e32 v%d, v%d" % (inv + 1, inv),
              # the group's first output row (+ 8 rows per store), how many of its rows exist in the tensor (nt01: group 0 | group 1 << 8)
              "s_mul_i32 %s, %%[rows8o], %d" % (S_TMP, 4 * g), "s_add_u32 s%d, s%d, %s" % (S_OBJ[0], OB[0], S_TMP), "s_addc_u32 s%d, s%d, 0" % (S_OBJ[1], OB[1]),
              "s_bfe_u32 %s, %%[nt01], 0x8%04x" % (S_TMP2, 8 * g)]
        for p in range(2):
            for dtl in range(2):
                for gp in range(2):
                    acc0 = 16 * (4 * g + 2 * p + dtl) + 8 * gp
                    o += ["v_accvgpr_read_b32 v%d, a%d" % (A + j, acc0 + j) for j in range(8)]
                    o += ["v_pk_mul_f32 %s, %s, %s" % (vr(A + 2 * j, 2), vr(A + 2 * j, 2), vr(inv, 2)) for j in range(4)]
                    o += ["v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (W + j, A + 2 * j, A + 2 * j + 1) for j in range(4)]
                    for k in range(2):      # 8 bytes: head_dim 64 p + 32 dtl + 16 gp + 8 k + 4 h .. + 3 = chunk 4 dtl + 2 gp + k of the pass, half h
                        c = 4 * dtl + 2 * gp + k
                        o += ["v_xor_b32_e32 v%d, %d, v%d" % (t, 16 * c, wmask), "v_add_u32_e32 v%d, v%d, v%d" % (t, t, wbase), "ds_write_b64 v%d, %s" % (t, vr(W + 2 * k, 2))]
            o += ["s_waitcnt lgkmcnt(0)"]
            o += ["ds_read_b128 %s, v%d offset:%d" % (vr(R + 4 * j, 4), (rbase, rbase_odd)[j & 1], 1024 * j) for j in range(4)]
            for j in range(4):      # rows 8 j .. 8 j + 7 of the group, 128 bytes each: lanes of rows beyond the tensor switched off
                o += ["s_waitcnt lgkmcnt(%d)" % (3 - j), "s_sub_u32 %s, %s, %d" % (S_TMP, S_TMP2, 8 * j), "s_max_i32 %s, %s, 0" % (S_TMP, S_TMP),
                      "v_cmp_gt_u32_e32 vcc, %s, v%d" % (S_TMP, l3), "s_and_saveexec_b64 %s, vcc" % sp(SRC),
                      "global_store_dwordx4 %%[ooffc], %s, s[%d:%d] offset:%d%s" % (vr(R + 4 * j, 4), S_OBJ[0], S_OBJ[1], 128 * p, O_STORE_POLICY),
                      "s_mov_b64 exec, %s" % sp(SRC)]
                if j < 3:
                    o += ["s_add_u32 s%d, s%d, %%[rows8o]" % (S_OBJ[0], S_OBJ[0]), "s_addc_u32 s%d, s%d, 0" % (S_OBJ[1], S_OBJ[1])]
            if p == 0:      # back to the group's first row for the second pass (3 x 8 rows down)
                o += ["s_mul_i32 %s, %%[rows8o], 3" % S_TMP, "s_sub_u32 s%d, s%d, %s" % (S_OBJ[0], S_OBJ[0], S_TMP), "s_subb_u32 s%d, s%d, 0" % (S_OBJ[1], S_OBJ[1])]
        # the lse of the rows that exist (bit 0 of the flags), from the lanes of half 0
        o += ["v_and_b32_e32 v%d, 1, v%d" % (t, fl), "v_cmp_eq_u32_e32 vcc, 1, v%d" % t, "s_mov_b64 %s, vcc" % sp(SRC), "v_cmp_gt_u32_e32 vcc, 32, v%d" % lane,
              "s_and_b64 vcc, vcc, %s" % sp(SRC), "s_and_saveexec_b64 %s, vcc" % sp(SRC), "global_store_dword v%d, v%d, %s" % (loff, lse, sp(LB)),
              "s_mov_b64 exec, %s" % sp(SRC)]
        # ---- the group's fragments, straight into the Q registers (lane (r, h): row r, chunk 2 ks + h of the half-tile image): the requests were
        # issued in front of 8 row stores and the lse store
        # (a group with rows beyond the tensor may have issued fewer: then everything has to land)
        if g == 0:
            o += stamp(2, "tail")      # the group's rows and lse are stored (issued): what follows is the wait for its Q rows
        o += ["s_bitcmp1_b32 %[ctl], 11", "s_cbranch_scc1 .Lf3_qgather%d_%%=" % g, "s_cmp_lt_u32 %s, 32" % S_TMP2, "s_cbranch_scc1 .Lf3_qw0%d_%%=" % g,
              "s_waitcnt vmcnt(9)", "s_branch .Lf3_qw%d_%%=" % g, ".Lf3_qw0%d_%%=:" % g, "s_waitcnt vmcnt(0)", ".Lf3_qw%d_%%=:" % g]
        o += ["ds_read_b128 %%[q%d], v%d offset:%d" % (8 * g + ks, (KRE, KRO)[ks & 1], 512 * (ks >> 1)) for ks in range(8)]
        o += ["s_waitcnt lgkmcnt(0)", "s_branch .Lf3_qdone%d_%%=" % g, ".Lf3_qgather%d_%%=:" % g]
        o += ["global_load_dwordx4 %%[q%d], %%[nqg%d], off%s" % (8 * g + ks, g, (" offset:%d" % (32 * ks)) if ks else "") for ks in range(8)]
        o += ["s_waitcnt vmcnt(0)", ".Lf3_qdone%d_%%=:" % g]
        if g == 0:
            o += stamp(3, "tail")      # group 0's fragments are in their registers
    return o


STAMP = False


STAMP_TAIL = os.environ.get("FWD3_STAMP_TAIL", "0") == "1"      # stamps 1..3 inside the tail (row group 0) instead of the prologue: tools/stamp_fwd3.py STAMP_TAIL=1


def stamp(k, where="main"):
    """diagnostic builds (sdpa_fwd3_loop_stamp.inc, -DHALVA_STAMP): the low word of s_memtime into output operand st<k> (k = 0..7)"""
    if not STAMP or (k in (1, 2, 3) and (where == "tail") != STAMP_TAIL):
        return []
    return ["s_memtime s[68:69]", "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 %%[st%d], s68" % k]


def main():
    global STAMP
    STAMP = False
    emit(os.environ.get("FWD3_OUT", "sdpa_fwd3_loop.inc"))
    STAMP = True
    emit(os.environ.get("FWD3_OUT", "sdpa_fwd3_loop.inc").replace(".inc", "_stamp.inc"))


def emit(out):
    global MASKED, PHASE
    M, body_p, ool_p = variant(False, "b")
    _, body_m, ool_m = variant(True, "m")
    L = []
    # ---------------- entry
    L += stamp(0)
    L += ["s_mov_b32 %s, m0" % S_M0SAVE,
          "v_readfirstlane_b32 s%d, %%[k_lo]" % DESC0[0], "v_readfirstlane_b32 s%d, %%[k_hi]" % (DESC0[0] + 1), "v_readfirstlane_b32 s%d, %%[nrec]" % (DESC0[0] + 2),
          "s_mov_b32 s%d, 0x00020000" % (DESC0[0] + 3), "v_readfirstlane_b32 %s, %%[soff0]" % S_SOFF0,
          "s_bfe_u32 %s, %%[ctl], 0x10004" % S_PF, "s_mov_b32 %s, 0" % S_REDO]
    # ---------------- one pass over the item (re-entered, S_REDO counting the repeats, when a row sum shows that P overflowed: at most MAX_REDO times -
    # inputs with NaN / inf scores would repeat for ever; they end with the NaN / inf they ask for instead)
    L += [".Lf3_pass_%=:", "s_mov_b64 s[%d:%d], s[%d:%d]" % (DESC[0], DESC[0] + 1, DESC0[0], DESC0[0] + 1), "s_mov_b64 s[%d:%d], s[%d:%d]" % (DESC[0] + 2, DESC[0] + 3, DESC0[0] + 2, DESC0[0] + 3),
          "s_mov_b32 %s, %s" % (S_SOFFK, S_SOFF0),
          "s_and_b32 %s, %%[ctl], 3" % S_T,                                   # ring slot of tile 0
          "s_bfe_u32 %s, %%[ctl], 0x20005" % S_NPF,                              # tiles of the next item to request behind this item's last
          "s_and_b32 %s, %%[nreq], 0xffff" % S_RLEFT, "s_lshr_b32 %s, %%[nreq], 16" % S_TOJUMP,
          # the repeat walks from tile 0 whatever the call said: give the three tiles the predecessor had requested back to the walk
          "s_bitcmp1_b32 %[ctl], 4", "s_cselect_b32 %s, 3, 0" % S_TMP, "s_cmp_eq_u32 %s, 0" % S_REDO, "s_cselect_b32 %s, 0, %s" % (S_TMP, S_TMP),
          "s_add_u32 %s, %s, %s" % (S_RLEFT, S_RLEFT, S_TMP),
          "s_cmp_eq_u32 %s, 0xffff" % S_TOJUMP, "s_cselect_b32 %s, 0, %s" % (S_TMP2, S_TMP), "s_add_u32 %s, %s, %s" % (S_TOJUMP, S_TOJUMP, S_TMP2),
          "s_cmp_eq_u32 %s, 0xffff" % S_TOJUMP, "s_cselect_b32 %s, -1, %s" % (S_TOJUMP, S_TOJUMP),
          "s_mul_i32 %s, %%[piece], %s" % (S_TMP2, S_TMP), "s_lshl_b32 %s, %s, 2" % (S_TMP2, S_TMP2), "s_sub_u32 %s, %s, %s" % (S_SOFFK, S_SOFFK, S_TMP2)]
    ool_pro = []
    # S_PF: the previous item's block has requested this item's tiles 0..2 already (nreq / k then describe the walk from tile 3 on)
    L += ["s_cmp_eq_u32 %s, 1" % S_PF, "s_cbranch_scc1 .Lf3_noreq_%="]
    for i in range(3):      # tiles 0..2 of the walk into slots S_T + i (every wave has passed the barrier behind the previous reads of the ring)
        pre, groups, post, ool = request_tile("p%d" % i, i)
        L += pre
        for g in groups:
            L += g
        L += post
        ool_pro += ool
    L += [".Lf3_noreq_%=:"]
    L += stamp(1)
    # ---------------- while they fly: zero O^T, state
    L += ["v_mov_b32_e32 v%d, 0" % (RING + j) for j in range(4)] + ["s_nop 4"]
    for o in range(8):
        L += ["v_mfma_f32_32x32x16_bf16 a[%d:%d], v[%d:%d], v[%d:%d], 0" % (16 * o, 16 * o + 15, RING, RING + 3, RING, RING + 3)]
    for g in (0, 1):
        L += ["v_mov_b32_e32 v%d, 0" % (PB[(g, 1)] + i) for i in range(8)]
        L += ["v_mov_b32_e32 v%d, 0" % (L2[g] + i) for i in range(4)] + (["v_mov_b32_e32 v%d, 0xff800000" % MX[g]] if TRACK_MAX else [])
    L += ["v_mov_b32_e32 v%d, 0xff800000" % V_NINF]
    L += ["s_and_b32 %s, %s, 3" % (S_TMP, S_T), "s_lshl_b32 %s, %s, 14" % (S_TOFFK, S_TMP), "s_mov_b32 %s, %s" % (S_TOFFV, S_TOFFK)]      # V "tile -1" := tile 0's slot (finite data; its P is zero)
    L += [i.text for i in k_addr_from_toff()] + [i.text for i in v_addr_from_toff()]
    # ---------------- this item's Q fragments and tile 0 have landed.  Requested here: [Q, the workgroup's first item only | the predecessor's last
    # row stores] [tiles 0, 1, 2]: the 16 requests of tiles 1, 2 may stay in flight.  Requested by the predecessor: it waited for its own Q requests,
    # which were younger than the tiles - nothing to wait for (only its last row stores are still out).
    L += stamp(2)
    L += ["s_cmp_eq_u32 %s, 1" % S_PF, "s_cbranch_scc1 .Lf3_landed_%=", "s_waitcnt vmcnt(16)", ".Lf3_landed_%=:", "s_barrier"]
    L += stamp(3)
    lines, fifo = chain_block(M, range(48, 64), 0, [])      # S(0), keys 0..31
    assert not fifo
    L += lines
    L += ["s_nop 7", "s_nop 7", "s_nop 7"]
    # its mask (always applied here: 64 passes everything) and the reference: m_ref = sc * max over the half tile (0 for a lane that sees nothing)
    # (which: ctl bits 8-9 - 0: the first tile is a plain one, 1: it opens masked run 1, 2: masked run 2)
    L += ["s_bfe_u32 %s, %%[ctl], 0x20008" % S_TMP, "v_mov_b32_e32 v%d, 64" % RANGE[0], "v_mov_b32_e32 v%d, 64" % RANGE[1],
          "s_cmp_eq_u32 %s, 1" % S_TMP, "s_cbranch_scc0 .Lf3_rsp1_%=", "v_mov_b32_e32 v%d, %%[rsA]" % RANGE[0], "v_mov_b32_e32 v%d, %%[rsA]" % RANGE[1], ".Lf3_rsp1_%=:",
          "s_cmp_eq_u32 %s, 2" % S_TMP, "s_cbranch_scc0 .Lf3_rsp2_%=", "v_mov_b32_e32 v%d, %%[rsB0]" % RANGE[0], "v_mov_b32_e32 v%d, %%[rsB1]" % RANGE[1], ".Lf3_rsp2_%=:"]
    for g in (0, 1):
        L += [i.text for i in mask_ops(g, 0, RANGE[g])]
    L += ["s_cmp_lg_u32 %s, 0" % S_REDO, "s_cbranch_scc1 .Lf3_mrdone_%="]      # (a repeat: the references are the first pass's, some of them raised)
    for g, t in ((0, V_T0), (1, V_T1)):
        x = XS[(g, 0)]
        L += ["v_max3_f32 v%d, v%d, v%d, v%d" % (t, x, x + 1, V_NINF)] + ["v_max3_f32 v%d, v%d, v%d, v%d" % (t, x + 2 * i, x + 2 * i + 1, t) for i in range(1, 8)]
        L += ["v_mov_b32_e32 v%d, v%d" % (MREF[g], t), "s_nop 1", "v_permlane32_swap_b32_e32 v%d, v%d" % (t, MREF[g]), "s_nop 1",
              "v_max_f32_e32 v%d, v%d, v%d" % (t, t, MREF[g]), "v_mul_f32_e32 v%d, %%[sc], v%d" % (MREF[g], t),
              "v_cmp_lt_f32_e32 vcc, v%d, v%d" % (V_NINF, MREF[g]), "v_cndmask_b32_e32 v%d, 0, v%d, vcc" % (MREF[g], MREF[g])]
    L += [".Lf3_mrdone_%=:"]
    # the loop's first A operands
    for c in range(LOOKAHEAD // 2):
        L += [l.text for l in a_loads(M[2 * c]["a"], c % 8)]
    L += stamp(4)
    # ---------------- the item's iterations: [plain n0][masked n1 from rsA][plain n2][masked n3 from rsB]
    L += ["s_mov_b32 %s, 0" % S_SEG, ".Lf3_seg_%=:", "s_cmp_eq_u32 %s, 0" % S_SEG, "s_cselect_b32 %s, %%[n01], %%[n23]" % S_FLD,
          "s_and_b32 %s, %s, 0xffff" % (S_CNT, S_FLD), "s_cmp_eq_u32 %s, 0" % S_CNT, "s_cbranch_scc1 .Lf3_noplain_%=", ".Lf3_plain_%=:"]
    L += body_p
    L += ["s_sub_u32 %s, %s, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 %s, 0" % S_CNT, "s_cbranch_scc1 .Lf3_plain_%=", ".Lf3_noplain_%=:",
          "s_lshr_b32 %s, %s, 16" % (S_CNT, S_FLD), "s_cmp_eq_u32 %s, 0" % S_CNT, "s_cbranch_scc1 .Lf3_nomasked_%=",
          "s_cmp_eq_u32 %s, 0" % S_SEG, "s_cbranch_scc0 .Lf3_rsb_%=",
          "v_mov_b32_e32 v%d, %%[rsA]" % RANGE[0], "v_mov_b32_e32 v%d, %%[rsA]" % RANGE[1], "s_branch .Lf3_masked_%=",
          ".Lf3_rsb_%=:", "v_mov_b32_e32 v%d, %%[rsB0]" % RANGE[0], "v_mov_b32_e32 v%d, %%[rsB1]" % RANGE[1], ".Lf3_masked_%=:"]
    L += body_m
    L += ["s_sub_u32 %s, %s, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 %s, 0" % S_CNT, "s_cbranch_scc1 .Lf3_masked_%=", ".Lf3_nomasked_%=:",
          "s_add_u32 %s, %s, 1" % (S_SEG, S_SEG), "s_cmp_lt_u32 %s, 2" % S_SEG, "s_cbranch_scc1 .Lf3_seg_%="]
    L += stamp(5)
    # ---------------- drain: O(last), keys 32..63 (its first A operands are in flight)
    lines, fifo = chain_block(M, range(0, 16), LOOKAHEAD // 2, carried_reads(M))
    L += lines
    L += ["s_branch .Lf3_end_%="] + ool_pro + ool_p + ool_m + [".Lf3_end_%=:", "s_waitcnt lgkmcnt(0)"]
    # ---------------- has any row's exponent reference been outgrown?  (TRACK_MAX above: judged by the row sums.)  Workgroup-wide - the waves share the
    # tile ring -: every wave leaves its answer in the item's vote words (LDS, double buffered by ctl bit 10), barrier, everybody reads all four.
    # The barrier is also the one behind which the ring may be requested into again.  The two lanes of a row (keys 4h.. of every 8) must move
    # their reference together: the larger of their partial sums decides for both.
    T = (V_T0, V_T1)
    R0, R1, R2 = RING, RING + 1, RING + 2
    FLAG = (SRC, (82, 83))      # which lanes repeat, per group (s[82:83] = S_SEG / S_FLD: loop-only)
    L += ["s_cmp_ge_u32 %s, %d" % (S_REDO, MAX_REDO), "s_cbranch_scc1 .Lf3_exit_%="]      # (uniform over the workgroup: every wave has counted the same repeats)
    for g in (0, 1):
        L += ["v_max3_f32 v%d, v%d, v%d, v%d" % (T[g], L2[g], L2[g] + 1, L2[g] + 2), "v_max_f32_e32 v%d, v%d, v%d" % (T[g], T[g], L2[g] + 3),
              "v_mov_b32_e32 v%d, v%d" % (R0, T[g]), "s_nop 1", "v_permlane32_swap_b32_e32 v%d, v%d" % (T[g], R0), "s_nop 1",
              "v_max_f32_e32 v%d, v%d, v%d" % (T[g], T[g], R0), "v_mov_b32_e32 v%d, 0x%08x" % (R0, REDO_LIMIT), "v_cmp_nlt_f32_e32 vcc, v%d, v%d" % (T[g], R0),
              "s_mov_b64 %s, vcc" % sp(FLAG[g])]
    L += ["s_or_b64 vcc, %s, %s" % (sp(FLAG[0]), sp(FLAG[1])),
          "s_cmp_lg_u64 vcc, 0", "s_cselect_b32 %s, 1, 0" % S_TMP, "v_mov_b32_e32 v%d, %s" % (R1, S_TMP),
          "s_bfe_u32 %s, %%[ctl], 0x1000a" % S_TMP, "s_lshl_b32 %s, %s, 4" % (S_TMP, S_TMP), "s_add_u32 %s, %s, %d" % (S_TMP, S_TMP, MAIL_LDS),
          "s_lshl_b32 %s, %%[wave], 2" % S_TMP2, "s_add_u32 %s, %s, %s" % (S_TMP2, S_TMP2, S_TMP), "v_mov_b32_e32 v%d, %s" % (R2, S_TMP2),
          "ds_write_b32 v%d, v%d" % (R2, R1), "s_waitcnt lgkmcnt(0)", "s_barrier",
          "v_mov_b32_e32 v%d, %s" % (R2, S_TMP), "ds_read_b128 v[%d:%d], v%d" % (RING + 4, RING + 7, R2), "s_waitcnt lgkmcnt(0)",
          "v_or3_b32 v%d, v%d, v%d, v%d" % (R1, RING + 4, RING + 5, RING + 6), "v_or_b32_e32 v%d, v%d, v%d" % (R1, R1, RING + 7), "s_nop 1",
          "v_readfirstlane_b32 %s, v%d" % (S_TMP, R1), "s_nop 3", "s_cmp_eq_u32 %s, 0" % S_TMP, "s_cbranch_scc1 .Lf3_exit_%="]
    # rare: the whole row block again, cold, the references of the rows that overflowed raised; what this pass requested for the next item must
    # have landed before the next pass requests into the same slots
    for g in (0, 1):
        L += ["s_mov_b64 vcc, %s" % sp(FLAG[g]), "v_add_f32_e32 v%d, 0x%08x, v%d" % (T[g], REDO_STEP, MREF[g]), "v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (MREF[g], MREF[g], T[g])]
    L += ["s_add_u32 %s, %s, 1" % (S_REDO, S_REDO), "s_mov_b32 %s, 0" % S_PF, "s_waitcnt vmcnt(0)", "s_branch .Lf3_pass_%="]
    # ---------------- way out.  Every wave has passed the vote's barrier: the ring slot of the item's last tile is free (the other three hold the next
    # item's first tiles).  Per row group g: [the NEXT item's Q rows of the group requested by LDS-DMA into this wave's 8 KiB of that slot - whole
    # 1-KiB pieces through a descriptor over the next sequence's Q rows, rows outside it arrive as zeros] [this item's rows of the group: O = O^T / l,
    # bf16, 16-byte pieces traded between the two lanes of a row (v_permlane32_swap), 8 row stores + the lse - in the shadow of the request]
    # [the fragments read back from LDS straight into the Q registers].  The 64 x 32-byte gather this replaces took the CU's address unit
    # ~4 500 cycles per item for its four waves (and, issued in front of the row stores, held those up as well).
    L += [".Lf3_exit_%=:"]
    L += stamp(6)
    L += tail_code()
    L += ["s_mov_b32 m0, %s" % S_M0SAVE]
    L += stamp(7)
    L = [x for l in L for x in l.replace("\\n\\t", "\n").split("\n")]
    diag = os.environ.get("FWD3_DIAG", "")      # timing experiments only (results are wrong): nodma / nobar, comma separated
    if "nodma" in diag:
        L = [l for l in L if not l.startswith("global_load_lds")]
    if "nobar" in diag:
        L = [l for l in L if l != "s_barrier" and not l.startswith("s_waitcnt vmcnt")]
    with open(out, "w") as f:
        f.write("// generated by gen_fwd3_loop.py - do not edit (python3 gen_fwd3_loop.py)\n")
        for l in L:
            f.write('"%s\\n\\t"\n' % l)
    with open(out.replace(".inc", "_clobbers.inc"), "w") as f:
        f.write("// generated by gen_fwd3_loop.py - do not edit\n")
        f.write(", ".join('"v%d"' % i for i in range(64, V_LAST + 1)) + ",\n" + ", ".join('"a%d"' % i for i in range(0, 128)) + ",\n" +
                ", ".join('"s%d"' % i for i in range(S_FIRST - 4, S_LAST + 1)) + ', "vcc", "scc", "memory"\n')
    print("%s: %d asm lines" % (out, len(L)))


if __name__ == "__main__":
    main()
