#!/usr/bin/env python3
"""Generates sdpa_dkv3_loop.inc: the interior-step loop of sdpa_bwd_dkv3 (sdpa_dkv3.h) as ONE inline-asm block.

A wave owns 32 keys (K / V fragments and the dK^T / dV^T accumulators live in the accumulator file); per 64-row step it runs the four
products itself - S = Q K^T, dP' = dO V^T - delta, dV^T += dO^T P, dK^T += Q^T dZ: 64 MFMAs - with P = exp2(S sc - lse2) formed in place
over S and dZ = P dP' in place over dP'.  This script assigns every instruction of the step to an MFMA gap:
  * A operands travel through a ring of 8 four-register slots; the LDS read(s) for MFMA n are issued in gap n - LOOKAHEAD;
  * the vector work is spread over the gaps behind the chain it depends on, at most CAP issue units per gap (a transcendental counts 2);
  * every s_waitcnt lgkmcnt(N) comes from a simulation of the in-order LDS queue over two consecutive iterations (steady state);
  * tile t+1 is waited for (counted vmcnt + s_barrier) in gap 57 of step t, in front of the reads of step t+1's first MFMAs; behind that
    barrier every wave has issued its last read of tile t, so its slot (a ring of FOUR) takes the requests of tile t+4 - 8 LDS-DMA pieces
    per wave and two statistics rows, gaps 58-63: three whole steps of lead; the dS of the step is stored behind its packs;
  * checks: no vector instruction reads an MFMA result earlier than two MFMAs behind the chain's last one, nothing consumes a v_exp
    result in the next issue slot, a packed operand is written >= 4 instructions before the MFMA that reads it.
The stream was developed and timed in experiments/dkv3 (2 419 cycles per step against 3 650 for hipcc's own schedule, bit-identical).

Operands of the asm block (sdpa_dkv3_call.h), by NAME in the emitted text (%[name]; the %NN in this script are shorthand resolved in main()):
accV0-3 / accK0-3 the dV^T / dK^T accumulators ("+a" / "=a", fixed registers a0-a127); the K / V fragments are NO operands (round 6) but the literal
registers a128-a191, named in the clobber list (possibly still in flight when the block is entered - its first call waits with s_waitcnt vmcnt(0));
rowrel, colrel, statrel the row-read, transposed-read and statistics lane offsets, voff_q / voff_do / stat_voff the lane offsets of the Q / dO tile
pieces and of the statistics request ("v"); sc = scale * log2 e, n02 / n1 the steps of the three phases (masked | plain | masked: the diagonal steps,
the interior, the tail; n02 = first | last << 16), ndma the steps that request a tile (+ the next item's tiles << 8, see dma_groups), wave,
q_piece / do_piece the bytes between a wave's pieces (16 rows) ("s");
lo0 / range the masks' per-lane bounds ("v"): a score of key kl and row ql = qt0 + 4 h + c (c = the register's row inside the step) survives
iff unsigned(c - lo) < range with lo = kl - qt0 - 4 h of the first step (the loop subtracts 64 per step) and range = len - kl (0 = lane off);
ctl the control word (CTL_* below; bit 11: the mail-box slot) ("s"); drawn ("=&v") what the home queue's counter at sched_ptr ("v", 64 bit) answered
(first call, thread 0); rec_ptr ("v", 64 bit) where wave 0's lane l < 16 finds its 16 bytes of the record of the item after next;
rec / nrec ("v") THE ITEM RECORDS of this item and the next (round 6; lane l = dword l, REC_FIELDS below): the bases / extents of the Q / dO /
statistics buffer descriptors are picked out of them with v_readlane_b32 here; what changes from call to call arrives as uniform values in
VECTOR registers (scalar operands are scarce, and when they run out the compiler silently hands the asm a vector register): q_soff / do_soff /
st_soff the first tile / statistics record to request in bytes from the descriptor's base, ds_lo / ds_hi the dS pointer of the call's first step (this
wave's strip).  The rows of the workgroup's NEXT item are requested by the block's last three steps (ndma bits 8-15: how many of its tiles).
The masked phases cost three more vector instructions per score."""
import os
import sys

LOOKAHEAD, CAP, CAP_MASKED = 6, 5, 9
WAIT_AGE = int(os.environ.get("DKV3_WAIT_AGE", "4"))      # 0: one wait per first use (rounds 2-3)
CARRY_K = 4
# cache policy of the dS stores: " nt" (rounds 2-4: written once, read once) or "" (experiments/ds_residency: does the Infinity Cache keep them
# for a consumer that follows closely?)
DS_STORE_POLICY = {"nt": " nt", "plain": "", "sc1": " sc1"}[os.environ.get("DKV3_DS_POLICY", "nt")]
X = [64, 96]; Y = [80, 112]; PB = [128, 144]; ZB = [136, 152]; SL = 160; RING = 176
QRE, QRO, DRE, DRO, QC0, QC1, DC0, DC1, STAT, DSOFF, LANE4, LANE, V_LO, V_T, V_NINF = 208, 209, 210, 211, 212, 213, 214, 215, 216, 217, 218, 219, 220, 221, 222
S_SLOT, S_CNT, S_DMALEFT, S_TOFF, S_TMP, S_TMP2 = "s72", "s73", "s74", "s75", "s80", "s81"
DSP, SDESC = (82, 83), (84, 87)      # the step's dS chunk (pointer); the statistics of this (sequence, head) as a buffer descriptor
S_DSTQ, S_M0SAVE, S_DSTS, S_SOFFS = "s88", "s89", "s90", "s91"      # (s96 free)
SRC = (98, 99)
# Q / dO tiles arrive through bounds-checked buffer descriptors over the sequence's rows of this head (round 4: as sdpa_fwd3's K / V tiles): a row past
# the sequence arrives as ZEROS (experiments/fwd3/oob_probe.hip), so the partial last tile and the requests past the block's end need no case of
# their own - three instructions per 1-KiB piece instead of ten.  The statistics (round 4, second pass): the delta kernel writes lse2 and -delta of a
# 64-row step side by side - [lse2 x 64][-delta x 64] = 512 bytes per step, steps in sequence coordinates (sdpa.hip:sdpa_bwd_delta_kernel) - so ONE
# `buffer_load_dwordx4 ... lds` per step brings both (lanes 0..31; the upper lanes' offsets lie outside the descriptor and write zeros into the unused
# half of the 1-KiB statistics slot), a step past the sequence brings zeros, and the selects, the dummy target and the clamped-lane path of the two
# pointer-form requests are gone: 4 instructions instead of 18 + 6, one vector-memory operation instead of two (a request costs the lone wave ~25-60
# cycles, experiments/issue_cost).
QDESC, DDESC = (92, 95), (76, 79)
S_SOFFQ, S_SOFFD = "s70", "s71"
MASKED = False
PHASE = "a"
DO_LDS, LSE_LDS, ND_DELTA, NSLOT = 65536, 131072, 256, 4      # statistics slot t & 3: [lse2 x 64][-delta x 64][512 bytes unused] at LSE_LDS + 1024 slot
VM_STEADY = 30                      # vector-memory operations issued behind the requests of tile t+1 when step t waits for it: 3 x 4 stores + 2 x 9 requests
WAIT_GAP = 64 - LOOKAHEAD - 1
# ctl, the call's control word: bit 0 first call of the key block, bits 1-2 tiles to request up front, bits 3-5 which of them is the partial
# last tile (7: none), bit 8 the run's last request is the partial
# last tile, bits 9-10 ring slot of the first step's tile;  s97 takes the field being looked at
CTL_NPRO, CTL_PROALT = "s_bfe_u32 s97, %[ctl], 0x20001", "s_bfe_u32 s97, %[ctl], 0x30003"
# Round 6: the ITEM RECORD.  Everything about an item (sequence, head, key block) that is uniform over the workgroup - its geometry, the run lengths
# of its first call, the bases / extents of its buffer descriptors, its dS strip, its K / dK rows - is worked out ONCE, by the delta pass of the same
# C-ABI call (sdpa.hip:dkv3_build_record), as 64 dwords in queue order.  A wave holds an item's record in ONE vector register (lane l = dword l,
# a ds_read_b32 of the workgroup's mail box) and the block picks its fields with v_readlane_b32; before, ~1 450 compiler instructions per item
# re-derived them (profiles/r06_dkv3_anatomy.log: 6 600 cycles in front of every block with nothing to overlap them).  REC: dword index by name.
REC_FIELDS = ["valid", "s", "hd", "kb", "start", "len", "br_a", "br_b", "kblk_min", "q_begin", "ntiles", "last_partial", "lr", "prefetchable",
              "n02", "n1", "t_side", "ndma", "part", "ctl0",
              "q_lo", "q_hi", "do_lo", "do_hi", "q_rec", "do_rec", "q_soff0", "do_soff0", "q_soff3", "do_soff3",
              "st_lo", "st_hi", "st_rec", "st_soff0", "st_soff3", "ds_lo", "ds_hi", "k_lo", "k_hi", "dk_lo", "dk_hi",
              "rope_pos0", "rope_pos1", "rope_pos2", "rope_pos3", "rows_ok0", "rows_ok1", "rows_ok2", "rows_ok3", "q_in_b0", "all_valid"]
REC = {n: i for i, n in enumerate(REC_FIELDS)}
assert len(REC) <= 64
MAIL_LDS = 131072 + 4096             # DKV3_SCHED (sdpa_dkv3.h): two records of 256 bytes
V_RECLOAD = RING + 8                 # v[184:187]: wave 0's lanes 0..15 fetch the record of the item after next here (prologue of a block's first call)
DMA_GAPS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]      # the quiet head of the step (no vector work yet): tile t+3 into the slot tile t-1 left at the last barrier


def vr(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1) if n > 1 else "v%d" % lo


def sp(pair):
    return "s[%d:%d]" % pair


class Ins:
    def __init__(self, text, kind, reads=(), writes=(), lds_defs=None, cost=0):
        self.text, self.kind, self.reads, self.writes, self.lds_defs, self.cost = text, kind, set(reads), set(writes), lds_defs, cost


def regs(lo, n):
    return ["v%d" % i for i in range(lo, lo + n)]


def mfma_list():
    out = []
    for sub in (0, 1):
        for ks in range(8):
            out.append(dict(prod="S", sub=sub, ks=ks, a=("row", "q", sub, ks)))
        for ks in range(8):
            out.append(dict(prod="dP", sub=sub, ks=ks, a=("row", "do", sub, ks)))
    for sub in (0, 1):
        for i in range(8):
            out.append(dict(prod="dV", sub=sub, i=i, a=("col", "do", sub, i)))
        for i in range(8):
            out.append(dict(prod="dK", sub=sub, i=i, a=("col", "q", sub, i)))
    return out


def a_loads(desc, slot):
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
    slot = RING + 4 * (n % 8)
    a = vr(slot, 4)
    if m["prod"] == "S":
        d = vr(X[m["sub"]], 16)
        c = "0" if m["ks"] == 0 else d
        return Ins("v_mfma_f32_32x32x16_bf16 %s, %s, %%%d, %s" % (d, a, 8 + m["ks"], c), "mfma",
                   reads=regs(slot, 4) + (regs(X[m["sub"]], 16) if m["ks"] else []), writes=regs(X[m["sub"]], 16))
    if m["prod"] == "dP":
        d = vr(Y[m["sub"]], 16)
        return Ins("v_mfma_f32_32x32x16_bf16 %s, %s, %%%d, %s" % (d, a, 16 + m["ks"], d), "mfma", reads=regs(slot, 4) + regs(Y[m["sub"]], 16),
                   writes=regs(Y[m["sub"]], 16))
    k16, dt = m["i"] // 4, m["i"] % 4
    if m["prod"] == "dV":
        b = PB[m["sub"]] + 4 * k16
        return Ins("v_mfma_f32_32x32x16_bf16 %%%d, %s, %s, %%%d" % (dt, a, vr(b, 4), dt), "mfma", reads=regs(slot, 4) + regs(b, 4))
    b = ZB[m["sub"]] + 4 * k16
    return Ins("v_mfma_f32_32x32x16_bf16 %%%d, %s, %s, %%%d" % (4 + dt, a, vr(b, 4), 4 + dt), "mfma", reads=regs(slot, 4) + regs(b, 4))


def valu_ops(sub):
    x, y, pb, zb = X[sub], Y[sub], PB[sub], ZB[sub]
    A = lambda r: Ins("v_fma_f32 v%d, v%d, %%30, -v%d" % (x + r, x + r, SL + r), "valu", reads=["v%d" % (x + r), "v%d" % (SL + r)], writes=["v%d" % (x + r)], cost=1)
    B = lambda r: Ins("v_exp_f32_e32 v%d, v%d" % (x + r, x + r), "trans", reads=["v%d" % (x + r)], writes=["v%d" % (x + r)], cost=2)
    C = lambda r: Ins("v_mul_f32_e32 v%d, v%d, v%d" % (y + r, x + r, y + r), "valu", reads=["v%d" % (x + r), "v%d" % (y + r)], writes=["v%d" % (y + r)], cost=1)
    Dp = lambda i: Ins("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (pb + i, x + 2 * i, x + 2 * i + 1), "valu", reads=["v%d" % (x + 2 * i), "v%d" % (x + 2 * i + 1)],
                       writes=["v%d" % (pb + i)], cost=1)
    E = lambda i: Ins("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (zb + i, y + 2 * i, y + 2 * i + 1), "valu", reads=["v%d" % (y + 2 * i), "v%d" % (y + 2 * i + 1)],
                      writes=["v%d" % (zb + i)], cost=1)
    def Mk(r):      # x = -inf unless unsigned(c_r - lo) < range: three dependent instructions kept together (they share VCC)
        c = 32 * sub + (r & 3) + 8 * (r >> 2)
        return [Ins("v_sub_u32_e32 v%d, %d, v%d" % (V_T, c, V_LO), "valu", reads=["v%d" % V_LO], writes=["v%d" % V_T], cost=1),
                Ins("v_cmp_lt_u32_e32 vcc, v%d, %%42" % V_T, "valu", reads=["v%d" % V_T], cost=1),
                Ins("v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (x + r, V_NINF, x + r), "valu", reads=["v%d" % (x + r), "v%d" % V_NINF], writes=["v%d" % (x + r)], cost=1)]
    AM = (lambda r: Mk(r) + [A(r)]) if MASKED else (lambda r: [A(r)])
    s1 = AM(0) + AM(1) + AM(2) + AM(3)
    for r in range(12):
        s1 += [B(r)] + AM(r + 4)
    s1 += [B(12), B(13), B(14), B(15)]
    s2 = []
    for i in range(8):
        s2 += [C(2 * i), C(2 * i + 1), Dp(i)]
        if i >= 1:
            s2.append(E(i - 1))
    s2.append(E(7))
    return [(1, o) for o in s1] + [(2, o) for o in s2]


def addr_setup():
    """address registers of the tile in ring slot S_SLOT (S_TOFF = slot * 16 KiB)"""
    o = []
    o.append(Ins("v_add_u32_e32 v%d, %s, %%24" % (QRE, S_TOFF), "valu", writes=["v%d" % QRE], cost=1))
    o.append(Ins("v_xor_b32_e32 v%d, 32, v%d" % (QRO, QRE), "valu", reads=["v%d" % QRE], writes=["v%d" % QRO], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DRE, DO_LDS, QRE), "valu", reads=["v%d" % QRE], writes=["v%d" % DRE], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DRO, DO_LDS, QRO), "valu", reads=["v%d" % QRO], writes=["v%d" % DRO], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %s, %%25" % (QC0, S_TOFF), "valu", writes=["v%d" % QC0], cost=1))
    o.append(Ins("v_xor_b32_e32 v%d, 32, v%d" % (QC1, QC0), "valu", reads=["v%d" % QC0], writes=["v%d" % QC1], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DC0, DO_LDS, QC0), "valu", reads=["v%d" % QC0], writes=["v%d" % DC0], cost=1))
    o.append(Ins("v_add_u32_e32 v%d, %d, v%d" % (DC1, DO_LDS, QC1), "valu", reads=["v%d" % QC1], writes=["v%d" % DC1], cost=1))
    o.append(Ins("s_lshl_b32 %s, %s, 10" % (S_TMP, S_SLOT), "salu"))
    o.append(Ins("s_add_u32 %s, %s, %d" % (S_TMP, S_TMP, LSE_LDS), "salu"))
    o.append(Ins("v_add_u32_e32 v%d, %s, %%26" % (STAT, S_TMP), "valu", writes=["v%d" % STAT], cost=1))
    return o


def sq(quad):
    return "s[%d:%d]" % quad


def dma_groups():
    """[list of instruction texts] x 9: the requests of the tile three steps ahead - WITHOUT a taken branch on the common path (a taken
    branch costs the wave tens of cycles).  Q / dO pieces and the statistics: `buffer_load_dwordx4 ... offen lds` through the sequence's
    descriptors - a row / step past the sequence brings zeros, a request past the block's last tile lands in a ring slot nobody reads any more:
    no special case, and every step has the same vector-memory operations, so the wait for tile t+1 is ONE counted vmcnt."""
    groups, ool = [], []
    # S_DMALEFT: bits 0-7 tiles of this block still to request; bits 8-15 tiles of the workgroup's NEXT item to request behind them (the steps
    # of a block that have no tile of their own left to ask for - its last three - ask for the next item's first tiles:
    # same slot rotation, so the next block simply starts on a rotated ring; free of charge, where a prefetch block of its own behind the
    # steps cost 3 700 cycles per item); bit 16: that switch has been made
    pre = ["s_and_b32 s97, %s, 0xff" % S_DMALEFT, "s_cmp_eq_u32 s97, 0", "s_cbranch_scc1 .Ldkv3_sw%s_%%=" % PHASE, ".Ldkv3_swb%s_%%=:" % PHASE,
           "s_add_u32 %s, %s, 3" % (S_TMP, S_SLOT), "s_and_b32 %s, %s, 3" % (S_TMP, S_TMP), "s_lshl_b32 %s, %s, 14" % (S_DSTQ, S_TMP),
           "s_lshl_b32 %s, %%34, 10" % S_TMP2, "s_add_u32 %s, %s, %s" % (S_DSTQ, S_DSTQ, S_TMP2), "s_lshl_b32 %s, %s, 10" % (S_DSTS, S_TMP),
           "s_add_u32 %s, %s, %d" % (S_DSTS, S_DSTS, LSE_LDS),
           "s_mov_b32 m0, %s" % S_DSTQ]
    k = 0
    for which, desc, soff, voff, piece, base in (("q", QDESC, S_SOFFQ, "%27", "%37", 0), ("do", DDESC, S_SOFFD, "%28", "%38", DO_LDS)):
        for i in range(4):
            g = (["s_add_u32 m0, %s, %d" % (S_DSTQ, base), "s_nop 0"] if (base and i == 0) else [])
            g += ["buffer_load_dwordx4 %s, %s, %s offen lds" % (voff, sq(desc), soff), "s_add_u32 %s, %s, %s" % (soff, soff, piece), "s_add_u32 m0, m0, 4096"]
            groups.append((pre if k == 0 else []) + g)
            k += 1
    stat = ["s_mov_b32 m0, %s" % S_DSTS, "s_nop 0", "buffer_load_dwordx4 %%[stat_voff], %s, %s offen lds" % (sq(SDESC), S_SOFFS), "s_add_u32 %s, %s, 512" % (S_SOFFS, S_SOFFS)]
    groups.append([] if "nostat" in os.environ.get("DKV3_DIAG", "") else stat)      # (timing experiment: the statistics request and its scalar code gone)
    ool += [".Ldkv3_sw%s_%%=:" % PHASE, "s_bitcmp1_b32 %s, 16" % S_DMALEFT, "s_cbranch_scc1 .Ldkv3_swb%s_%%=" % PHASE,      # already switched: nothing left
            "s_lshr_b32 %s, %s, 8" % (S_DMALEFT, S_DMALEFT), "s_or_b32 %s, %s, 0x10000" % (S_DMALEFT, S_DMALEFT),
            "s_and_b32 s97, %s, 0xff" % S_DMALEFT, "s_cmp_eq_u32 s97, 0", "s_cbranch_scc1 .Ldkv3_swb%s_%%=" % PHASE]            # no next item to serve
    def nrl(sreg, field):
        return "v_readlane_b32 %s, %%[nrec], %d" % (sreg, REC[field])
    ool += [nrl("s%d" % QDESC[0], "q_lo"), nrl("s%d" % (QDESC[0] + 1), "q_hi"), nrl("s%d" % (QDESC[0] + 2), "q_rec"),
            nrl("s%d" % DDESC[0], "do_lo"), nrl("s%d" % (DDESC[0] + 1), "do_hi"), nrl("s%d" % (DDESC[0] + 2), "do_rec"),
            nrl(S_SOFFQ, "q_soff0"), nrl(S_SOFFD, "do_soff0"),
            nrl("s%d" % SDESC[0], "st_lo"), nrl("s%d" % (SDESC[0] + 1), "st_hi"), nrl("s%d" % (SDESC[0] + 2), "st_rec"),
            nrl(S_SOFFS, "st_soff0")]
    ool += ["s_nop 3", "s_branch .Ldkv3_swb%s_%%=" % PHASE]
    return groups, ool


def build_body():
    M = mfma_list()
    gaps = [[] for _ in range(64)]
    for n in range(64):
        g = n - LOOKAHEAD
        if g >= 0:
            gaps[g] += a_loads(M[n]["a"], n % 8)

    def stat(sub, j, which):
        off = (ND_DELTA if which == "nd" else 0) + 128 * sub + 32 * j
        dst = (Y[sub] if which == "nd" else SL) + 4 * j
        return Ins("ds_read_b128 %s, v%d offset:%d" % (vr(dst, 4), STAT, off), "lds", reads=["v%d" % STAT], writes=regs(dst, 4), lds_defs=regs(dst, 4))
    for j in range(4):
        gaps[0 + j].append(stat(0, j, "nd"))
        gaps[4 + j].append(stat(0, j, "lse"))
        gaps[15 + j].append(stat(1, j, "nd"))
        gaps[19 + j].append(stat(1, j, "lse"))
    chain_end = {("S", 0): 7, ("dP", 0): 15, ("S", 1): 23, ("dP", 1): 31}
    used = [sum(i.cost for i in g) for g in gaps]
    for sub in (0, 1):
        g = 0
        for stage, ins in valu_ops(sub):
            earliest = chain_end[("S", sub)] + 2 if stage == 1 else max(chain_end[("dP", sub)] + 2, g)
            g = max(g, earliest)
            while used[g] + ins.cost > (CAP_MASKED if MASKED else CAP):
                g += 1
            gaps[g].append(ins)
            used[g] += ins.cost
        assert g < 48, "vector work of a sub-tile ran past its packs' consumers"

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
            assert g0 + half < WAIT_GAP           # (the vmcnt arithmetic assumes the step's stores are issued in front of its wait)
            gaps[g0 + half].append(Ins("global_store_dwordx4 v%d, %s, %s offset:%d%s" % (DSOFF, vr(ZB[sub] + 4 * half, 4), sp(DSP), 2048 * sub + 1024 * half, DS_STORE_POLICY),
                                       "vmem", reads=regs(ZB[sub] + 4 * half, 4) + ["v%d" % DSOFF]))
    dma, ool = dma_groups()
    # gap WAIT_GAP (behind the reads of MFMA 63): tile t+1 has landed for every wave; move on to its slot
    # the wait: tile t+1 was requested three steps ago; while this step still requested a tile, VM_STEADY younger operations may stay in flight
    w = ["s_waitcnt vmcnt(%d)" % VM_STEADY, "s_barrier",
         "s_add_u32 %s, %s, 1" % (S_SLOT, S_SLOT), "s_and_b32 %s, %s, %d" % (S_SLOT, S_SLOT, NSLOT - 1), "s_lshl_b32 %s, %s, 14" % (S_TOFF, S_SLOT),
         "s_add_u32 s%d, s%d, 16384" % (DSP[0], DSP[0]), "s_addc_u32 s%d, s%d, 0" % (DSP[1], DSP[1]),
         "v_subrev_u32_e32 v%d, 64, v%d" % (V_LO, V_LO)]      # the next step's rows lie 64 further down: lo -= 64 (kept in both variants)
    gaps[WAIT_GAP] += [Ins(t, "raw") for t in w] + addr_setup()
    for n in range(LOOKAHEAD):
        gaps[64 - LOOKAHEAD + n] += a_loads(M[n]["a"], n % 8)
    for k, grp in enumerate(dma):
        gaps[DMA_GAPS[k]] += [Ins(t, "raw") for t in grp]
    gaps[DMA_GAPS[-1]] += [Ins(t, "raw") for t in ("s_and_b32 s97, %s, 0xff" % S_DMALEFT, "s_min_u32 s97, s97, 1", "s_sub_u32 %s, %s, s97" % (S_DMALEFT, S_DMALEFT))]
    return M, gaps, ool


def linearize(M, gaps):
    seq = []
    for n in range(64):
        seq.append(mfma_ins(n, M[n]))
        seq += gaps[n]
    return seq


def insert_waits(seq, carried, final_keep=None):
    """s_waitcnt lgkmcnt(N) in front of the first user of an LDS read, N from the in-order queue.  A wait that is due anyway also covers every
    younger read issued at least WAIT_AGE MFMAs ago (long landed): every instruction of the single wave, a wait included, is an issue slot of
    ~4 cycles, and the stream had one wait per MFMA (round 4; ages in MFMAs relative to the iteration's first)."""
    fifo = [dict(e, age=e["age"] - 64 if e.get("age", -99) > 0 else e.get("age", -99)) for e in carried]
    pending = {}
    for e in fifo:
        for r in e["defs"]:
            pending[r] = e
    lines = []
    prev = None
    now = 0
    for ins in seq:
        if ins.kind == "mfma":
            now += 1
        need = [pending[r] for r in (ins.reads | ins.writes) if r in pending]
        if need:
            last = max(fifo.index(e) for e in need)
            while WAIT_AGE and last + 1 < len(fifo) and now - fifo[last + 1]["age"] >= WAIT_AGE:
                last += 1
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
            e = {"defs": set(ins.lds_defs), "age": now}
            fifo.append(e)
            for r in e["defs"]:
                pending[r] = e
        assert len(fifo) <= 15, "more than 15 LDS reads in flight"
        if ins.kind not in ("salu", "raw"):
            prev = ins
    if final_keep is not None and len(fifo) > final_keep:      # a canonical queue at the iteration's end: only the youngest reads stay in flight
        lines.append("s_waitcnt lgkmcnt(%d)" % final_keep)
        fifo = fifo[-final_keep:]
    return lines, fifo


def check(seq):
    pos_mfma = [i for i, s in enumerate(seq) if s.kind == "mfma"]
    last_writer = {}
    for i, s in enumerate(seq):
        if s.kind in ("valu", "trans", "vmem"):
            for r in s.reads:
                if r in last_writer and last_writer[r][0] == "mfma":
                    assert sum(1 for p in pos_mfma if last_writer[r][1] < p < i) >= 2, "%s reads %s too close behind its MFMA chain" % (s.text, r)
        if s.kind == "mfma":
            for r in s.reads:
                if r in last_writer and last_writer[r][0] in ("valu", "trans"):
                    assert i - last_writer[r][1] >= 4, "%s reads %s right behind the vector write" % (s.text, r)
        for r in s.writes:
            last_writer[r] = (s.kind, i)


def variant(masked, phase):
    global MASKED, PHASE
    MASKED, PHASE = masked, phase
    M, gaps, ool = build_body()
    seq = linearize(M, gaps)
    check(seq + seq)
    # every phase's iteration starts and ends with the SAME reads in flight - the youngest CARRY_K of the next iteration's first operands - so that
    # the phases can follow each other in any order; whoever enters a loop (the prologue, main()) has waited down to that state
    carried = []
    for n in range(LOOKAHEAD):
        for l in a_loads(M[n]["a"], n % 8):
            carried.append({"defs": set(l.lds_defs), "age": n - LOOKAHEAD + 1})
    carried = carried[-CARRY_K:]
    lines1, fifo1 = insert_waits(seq, carried, CARRY_K)
    lines2, fifo2 = insert_waits(seq, fifo1, CARRY_K)
    assert [sorted(e["defs"]) for e in fifo1] == [sorted(e["defs"]) for e in carried], "the iteration's last reads are not the next one's first operands"
    assert lines1 == lines2 and [sorted(e["defs"]) for e in fifo1] == [sorted(e["defs"]) for e in fifo2], "loop is not in steady state"
    nv = sum(1 for s in seq if s.kind in ("valu", "trans"))
    nl = sum(1 for s in seq if s.kind == "lds")
    print("phase %s (%s): 64 MFMAs, %d vector, %d LDS reads, %d asm lines per iteration" % (phase, "masked" if masked else "plain", nv, nl, len(lines1)))
    return M, lines1, ool


def main():
    import os
    out = os.environ.get("DKV3_OUT", "sdpa_dkv3_loop.inc")
    M, body_a, ool_a = variant(True, "a")
    _, body_b, ool_b = variant(False, "b")
    _, body_c, ool_c = variant(True, "c")
    pro = ["s_waitcnt lgkmcnt(0)", "s_mov_b32 %s, m0" % S_M0SAVE,
           "v_mbcnt_lo_u32_b32 v%d, -1, 0" % LANE, "v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (LANE, LANE),
           "v_lshlrev_b32_e32 v%d, 4, v%d" % (DSOFF, LANE), "v_lshlrev_b32_e32 v%d, 2, v%d" % (LANE4, LANE),
           "s_bfe_u32 %s, %%[ctl], 0x20009" % S_SLOT, "s_mov_b32 %s, %%32" % S_DMALEFT] + \
          ["v_readfirstlane_b32 s%d, %%[ds_%s]" % (DSP[w], "lo" if w == 0 else "hi") for w in (0, 1)] + [
           # the statistics of this (sequence, head): 512 bytes per 64-row step from the sequence's first row on; st_soff: the first step to request
           "v_readlane_b32 s%d, %%[rec], %d" % (SDESC[0], REC["st_lo"]), "v_readlane_b32 s%d, %%[rec], %d" % (SDESC[0] + 1, REC["st_hi"]),
           "v_readlane_b32 s%d, %%[rec], %d" % (SDESC[0] + 2, REC["st_rec"]),
           "s_mov_b32 s%d, 0x00020000" % (SDESC[0] + 3), "v_readfirstlane_b32 %s, %%[st_soff]" % S_SOFFS,
           # the sequence's Q / dO rows of this head as buffer descriptors (base, 0 stride, bytes up to the end of the last row, raw 32-bit format);
           # q_soff / do_soff: the first tile to request, in bytes from the sequence's first row
           "v_readlane_b32 s%d, %%[rec], %d" % (QDESC[0], REC["q_lo"]), "v_readlane_b32 s%d, %%[rec], %d" % (QDESC[0] + 1, REC["q_hi"]),
           "v_readlane_b32 s%d, %%[rec], %d" % (QDESC[0] + 2, REC["q_rec"]),
           "s_mov_b32 s%d, 0x00020000" % (QDESC[0] + 3),
           "v_readlane_b32 s%d, %%[rec], %d" % (DDESC[0], REC["do_lo"]), "v_readlane_b32 s%d, %%[rec], %d" % (DDESC[0] + 1, REC["do_hi"]),
           "v_readlane_b32 s%d, %%[rec], %d" % (DDESC[0] + 2, REC["do_rec"]),
           "s_mov_b32 s%d, 0x00020000" % (DDESC[0] + 3),
           "v_readfirstlane_b32 %s, %%[q_soff]" % S_SOFFQ, "v_readfirstlane_b32 %s, %%[do_soff]" % S_SOFFD,

           "s_lshl_b32 %s, %s, 14" % (S_TOFF, S_SLOT), "v_mov_b32_e32 v%d, %%41" % V_LO, "v_mov_b32_e32 v%d, 0xff800000" % V_NINF]
    global PHASE
    PHASE = "p"
    # ---- first call of a key block (control bit 0): zero the accumulators and request the block's first (<= 3) tiles + statistics rows here,
    #      with the loop's own cheap form of a request (the caller has passed the barrier behind the previous block's last reads); tile CTL_PROALT
    #      of them (7 = none) is the sequence's partial last tile and takes the clamped lane offsets.  Then everything has to land.
    pro += ["s_bitcmp0_b32 %[ctl], 0", "s_cbranch_scc1 .Ldkv3_nofirst_%="]
    # the workgroup's scheduler (sdpa_dkv3.h): thread 0 draws the item after the item after next HERE - the counter's answer comes back under
    # the s_waitcnt vmcnt(0) below and leaves the block as an ordinary output.  Asked for by the compiler's own atomic it had to be waited for
    # with vmcnt(0) in the middle of the next round (the compiler cannot count past an asm block): ~6 000 cycles per item.
    pro += ["s_cmp_lg_u32 %[wave], 0", "s_cbranch_scc1 .Ldkv3_nodraw_%=", "s_mov_b64 %s, exec" % sp(SRC), "s_mov_b64 exec, 1", "v_mov_b32_e32 v%d, 1" % RING,
            "global_atomic_add %%[drawn], %%[sched_ptr], v%d, off sc0" % RING,
            # ... and its lanes 0..15 fetch the RECORD of the item after next (rec_ptr: record + 16 lane; thread 0 turned the previous answer of the
            # counter into a record index in front of the block): 256 bytes that come back under the same wait and go into the mail box below
            "s_mov_b64 exec, 0xffff", "global_load_dwordx4 %s, %%[rec_ptr], off" % vr(V_RECLOAD, 4),
            "s_mov_b64 exec, %s" % sp(SRC), ".Ldkv3_nodraw_%=:"]
    # (an accumulator tuple operand cannot be sliced into single registers from here: it is zeroed by an MFMA of zero fragments, D = 0 * 0 + 0)
    pro += ["v_mov_b32_e32 v%d, 0" % (RING + j) for j in range(4)] + ["s_nop 4"]
    for opnd in range(8):
        pro += ["ZERO_TUPLE %%%d" % opnd]
    for i in range(3):
        pro += [CTL_NPRO, "s_cmp_le_u32 s97, %d" % i, "s_cbranch_scc1 .Ldkv3_prodone_%="]
        pro += ["s_add_u32 %s, %s, %d" % (S_TMP, S_SLOT, i), "s_and_b32 %s, %s, 3" % (S_TMP, S_TMP), "s_lshl_b32 %s, %s, 14" % (S_DSTQ, S_TMP),
                "s_lshl_b32 %s, %%34, 10" % S_TMP2, "s_add_u32 %s, %s, %s" % (S_DSTQ, S_DSTQ, S_TMP2), "s_lshl_b32 %s, %s, 10" % (S_DSTS, S_TMP),
                "s_add_u32 %s, %s, %d" % (S_DSTS, S_DSTS, LSE_LDS)]
        for which, desc, soff, voff, piece, base in (("q", QDESC, S_SOFFQ, "%27", "%37", 0), ("do", DDESC, S_SOFFD, "%28", "%38", DO_LDS)):
            pro += ["s_add_u32 m0, %s, %d" % (S_DSTQ, base), "s_nop 0"]
            for k in range(4):
                pro += ["buffer_load_dwordx4 %s, %s, %s offen lds" % (voff, sq(desc), soff), "s_add_u32 %s, %s, %s" % (soff, soff, piece), "s_add_u32 m0, m0, 4096", "s_nop 0"]
        pro += ["s_mov_b32 m0, %s" % S_DSTS, "s_nop 0", "buffer_load_dwordx4 %%[stat_voff], %s, %s offen lds" % (sq(SDESC), S_SOFFS), "s_add_u32 %s, %s, 512" % (S_SOFFS, S_SOFFS)]
    # the mail box: slot ctl bit 11, 16 bytes per lane; every wave reads it (ds_read_b32, lane l = dword l) behind the round's last barrier
    pro += [".Ldkv3_prodone_%=:", "s_waitcnt vmcnt(0)", "s_cmp_lg_u32 %[wave], 0", "s_cbranch_scc1 .Ldkv3_nomail_%=",
            "s_bfe_u32 %s, %%[ctl], 0x1000b" % S_TMP, "s_lshl_b32 %s, %s, 8" % (S_TMP, S_TMP), "s_add_u32 %s, %s, %d" % (S_TMP, S_TMP, MAIL_LDS),
            "v_add_u32_e32 v%d, %s, v%d" % (V_T, S_TMP, DSOFF), "s_mov_b64 %s, exec" % sp(SRC), "s_mov_b64 exec, 0xffff",
            "ds_write_b128 v%d, %s" % (V_T, vr(V_RECLOAD, 4)), "s_mov_b64 exec, %s" % sp(SRC), ".Ldkv3_nomail_%=:",
            "s_barrier", ".Ldkv3_nofirst_%=:"]
    pro += [i.text for i in addr_setup()]
    for n in range(LOOKAHEAD):
        pro += [l.text for l in a_loads(M[n]["a"], n % 8)]
    pro += ["s_waitcnt lgkmcnt(%d)" % CARRY_K]      # (the state every iteration starts in: variant())
    lines = list(pro)
    for phase, body, count in (("a", body_a, "s_and_b32 %s, %%[n02], 0xffff" % S_CNT), ("b", body_b, "s_mov_b32 %s, %%[n1]" % S_CNT), ("c", body_c, "s_lshr_b32 %s, %%[n02], 16" % S_CNT)):
        lines += [count, "s_cmp_eq_u32 %s, 0" % S_CNT, "s_cbranch_scc1 .Ldkv3_skip%s_%%=" % phase, ".Ldkv3_loop%s_%%=:" % phase]
        lines += body
        lines += ["s_sub_u32 %s, %s, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 %s, 0" % S_CNT, "s_cbranch_scc1 .Ldkv3_loop%s_%%=" % phase, ".Ldkv3_skip%s_%%=:" % phase]
    lines += ["s_branch .Ldkv3_end_%="] + ool_a + ool_b + ool_c + [".Ldkv3_end_%=:", "s_waitcnt lgkmcnt(0)", "s_mov_b32 m0, %s" % S_M0SAVE]
    names = {24: "rowrel", 25: "colrel", 26: "statrel", 27: "voff_q", 28: "voff_do", 30: "sc", 32: "ndma", 34: "wave", 37: "q_piece", 38: "do_piece", 41: "lo0", 42: "range"}
    names.update({i: "accV%d" % i for i in range(4)}); names.update({4 + i: "accK%d" % i for i in range(4)})
    names.update({8 + i: "kq%d" % i for i in range(8)}); names.update({16 + i: "vq%d" % i for i in range(8)})
    import re
    def named(l):      # the asm statement's operands are NAMED (sdpa_dkv3.h): the numbers above are this script's shorthand
        # (round 6: the K / V fragments are no operands any more but the literal registers a128-a191: the clobber list names them, which keeps the
        # compiler's own spills out of them, and tools/check_dkv3_isa.py holds it to that.  It lets an experiment load them across the item loop's back
        # edge - experiments/dkv3_item_boundary - where, as loop-carried operands, the compiler moved them through vector registers)
        def one(m):
            n = int(m.group(1))
            if 8 <= n < 24:
                return "a[%d:%d]" % (128 + 4 * (n - 8), 131 + 4 * (n - 8))
            return "%%[%s]" % names[n]
        return re.sub(r"%(\d+)", one, l)
    lines = [named(l) for l in lines]
    expanded = []
    for l in lines:
        if l.startswith("ZERO_TUPLE"):      # D = 0 * 0 + 0: the operand tuple cannot be sliced into single registers from here
            expanded.append("v_mfma_f32_32x32x16_bf16 %s, v[%d:%d], v[%d:%d], 0" % (l.split()[1], RING, RING + 3, RING, RING + 3))
        else:
            expanded.append(l)
    lines = expanded
    diag = os.environ.get("DKV3_DIAG", "")      # timing experiments only (results are wrong): nodma / nobar / nostore, comma separated
    if "nodma" in diag:
        lines = [l for l in lines if not l.startswith("global_load_lds") and not (l.startswith("buffer_load") and l.endswith(" lds"))]
    if "nobar" in diag:
        lines = [l for l in lines if l != "s_barrier" and not l.startswith("s_waitcnt vmcnt")]
    if "nostore" in diag:
        lines = [l for l in lines if not l.startswith("global_store")]
    with open(out, "w") as f:
        f.write("// generated by gen_dkv3_loop.py - do not edit (python3 gen_dkv3_loop.py)\n")
        for l in lines:
            f.write('"%s\\n\\t"\n' % l)
    with open(os.environ.get("DKV3_OUT", "sdpa_dkv3_loop.inc").replace(".inc", "_clobbers.inc"), "w") as f:
        f.write("// generated by gen_dkv3_loop.py - do not edit\n")
        f.write(", ".join('"v%d"' % i for i in range(64, 223)) + ",\n" + ", ".join('"a%d"' % i for i in range(128, 192)) + ",\n" +
                ", ".join('"s%d"' % i for i in range(70, 100)) + ', "vcc", "scc", "memory"\n')
    with open(os.environ.get("DKV3_OUT", "sdpa_dkv3_loop.inc").replace(".inc", "_rec.inc"), "w") as f:
        f.write("// generated by gen_dkv3_loop.py - do not edit: dword index of every field of an item record (gen_dkv3_loop.py:REC_FIELDS)\n")
        f.write("enum Dkv3Rec : int {\n" + "".join("    DKV3_REC_%s = %d,\n" % (n.upper(), i) for n, i in REC.items()) + "    DKV3_REC_DWORDS = 64\n};\n")
    print("%s: %d asm lines" % (out, len(lines)))


if __name__ == "__main__":
    main()
