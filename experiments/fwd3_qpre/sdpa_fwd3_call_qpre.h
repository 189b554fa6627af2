// The call of the generated block of the Q-prefetching build (sdpa_fwd3_loop_qpre.inc <- FWD3_QPRE=1 python3 gen_fwd3_loop.py) - included TWICE by
// sdpa_fwd3_item<true> (sdpa_fwd3.h).  The two fragment sets have FIXED homes in every statement - qf in a[128:191], qf2 in a[192:255], both "+" - so
// that the compiler never has a reason to move them (an output-only / input-only binding made it copy them through vector registers at the loop's
// back edge: tools/check_fwd3_isa.py); what differs is the NAME the block knows them by: FWD3_QSET 0 - q (this item's fragments) = qf, nq (the next
// item's, asked for by the block's prologue) = qf2; FWD3_QSET 1 - the other way round.  A workgroup's consecutive items alternate.  The block owns the
// accumulators (a[0:127], clobbered) and stores the item's rows and lse itself.  (Text written by tools/r05/gen_fwd3_call_qpre.py.)
asm volatile(
#include "sdpa_fwd3_loop_qpre.inc"
#if FWD3_QSET == 0
    : [q0] "+{a[128:131]}"(qf[0]), [q1] "+{a[132:135]}"(qf[1]), [q2] "+{a[136:139]}"(qf[2]), [q3] "+{a[140:143]}"(qf[3]), [q4] "+{a[144:147]}"(qf[4]), [q5] "+{a[148:151]}"(qf[5]), [q6] "+{a[152:155]}"(qf[6]), [q7] "+{a[156:159]}"(qf[7]), [q8] "+{a[160:163]}"(qf[8]), [q9] "+{a[164:167]}"(qf[9]), [q10] "+{a[168:171]}"(qf[10]), [q11] "+{a[172:175]}"(qf[11]), [q12] "+{a[176:179]}"(qf[12]), [q13] "+{a[180:183]}"(qf[13]), [q14] "+{a[184:187]}"(qf[14]), [q15] "+{a[188:191]}"(qf[15]),
      [nq0] "+{a[192:195]}"(qf2[0]), [nq1] "+{a[196:199]}"(qf2[1]), [nq2] "+{a[200:203]}"(qf2[2]), [nq3] "+{a[204:207]}"(qf2[3]), [nq4] "+{a[208:211]}"(qf2[4]), [nq5] "+{a[212:215]}"(qf2[5]), [nq6] "+{a[216:219]}"(qf2[6]), [nq7] "+{a[220:223]}"(qf2[7]), [nq8] "+{a[224:227]}"(qf2[8]), [nq9] "+{a[228:231]}"(qf2[9]), [nq10] "+{a[232:235]}"(qf2[10]), [nq11] "+{a[236:239]}"(qf2[11]), [nq12] "+{a[240:243]}"(qf2[12]), [nq13] "+{a[244:247]}"(qf2[13]), [nq14] "+{a[248:251]}"(qf2[14]), [nq15] "+{a[252:255]}"(qf2[15])
#else
    : [q0] "+{a[192:195]}"(qf2[0]), [q1] "+{a[196:199]}"(qf2[1]), [q2] "+{a[200:203]}"(qf2[2]), [q3] "+{a[204:207]}"(qf2[3]), [q4] "+{a[208:211]}"(qf2[4]), [q5] "+{a[212:215]}"(qf2[5]), [q6] "+{a[216:219]}"(qf2[6]), [q7] "+{a[220:223]}"(qf2[7]), [q8] "+{a[224:227]}"(qf2[8]), [q9] "+{a[228:231]}"(qf2[9]), [q10] "+{a[232:235]}"(qf2[10]), [q11] "+{a[236:239]}"(qf2[11]), [q12] "+{a[240:243]}"(qf2[12]), [q13] "+{a[244:247]}"(qf2[13]), [q14] "+{a[248:251]}"(qf2[14]), [q15] "+{a[252:255]}"(qf2[15]),
      [nq0] "+{a[128:131]}"(qf[0]), [nq1] "+{a[132:135]}"(qf[1]), [nq2] "+{a[136:139]}"(qf[2]), [nq3] "+{a[140:143]}"(qf[3]), [nq4] "+{a[144:147]}"(qf[4]), [nq5] "+{a[148:151]}"(qf[5]), [nq6] "+{a[152:155]}"(qf[6]), [nq7] "+{a[156:159]}"(qf[7]), [nq8] "+{a[160:163]}"(qf[8]), [nq9] "+{a[164:167]}"(qf[9]), [nq10] "+{a[168:171]}"(qf[10]), [nq11] "+{a[172:175]}"(qf[11]), [nq12] "+{a[176:179]}"(qf[12]), [nq13] "+{a[180:183]}"(qf[13]), [nq14] "+{a[184:187]}"(qf[14]), [nq15] "+{a[188:191]}"(qf[15])
#endif
    : [nqg0] "v"(nqg0), [nqg1] "v"(nqg1),
      [o_lo] "v"((unsigned)o_base), [o_hi] "v"((unsigned)(o_base >> 32)), [lse_lo] "v"((unsigned)lse_base), [lse_hi] "v"((unsigned)(lse_base >> 32)),
      [ooffc] "v"(ooffc), [rows8o] "s"(rows8o), [nt01] "s"(nt01), [loff0] "v"(loff[0]), [loff1] "v"(loff[1]), [rowrel] "v"(rowrel), [colrel] "v"(colrel), [voff] "v"(voff), [rsA] "v"(rsA),
      [rsB0] "v"(rsB[0]), [rsB1] "v"(rsB[1]), [k_lo] "v"((unsigned)k_base),
      [k_hi] "v"((unsigned)(k_base >> 32)), [nrec] "v"(nrec), [soff0] "v"(soff0), [nk_lo] "v"((unsigned)nk_base), [nk_hi] "v"((unsigned)(nk_base >> 32)),
      [nnrec] "v"(nnrec), [nsoff0] "v"(nsoff0), [vdlo] "s"(vdlo), [sc] "s"(sc), [n01] "s"(n01),
      [n23] "s"(n23), [nreq] "s"(nreq), [jlo] "s"(jlo), [wave] "s"(wave_u), [piece] "s"(piece), [ctl] "s"(ctl0)
    :
#include "sdpa_fwd3_loop_qpre_clobbers.inc"
);
