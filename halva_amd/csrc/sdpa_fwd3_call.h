// The call of the generated block (sdpa_fwd3_loop.inc) for the current row block (sdpa_fwd3_item, sdpa_fwd3.h).  The block owns the accumulators
// (a[0:127], clobbered) and stores the item's rows and lse itself.  The Q fragments go in as this item's and come out as the NEXT item's, landed.
asm volatile(
#ifdef HALVA_STAMP
#include "sdpa_fwd3_loop_stamp.inc"
#else
#include "sdpa_fwd3_loop.inc"
#endif
    : [q0] "+{a[128:131]}"(qf[0]), [q1] "+{a[132:135]}"(qf[1]), [q2] "+{a[136:139]}"(qf[2]), [q3] "+{a[140:143]}"(qf[3]), [q4] "+{a[144:147]}"(qf[4]), [q5] "+{a[148:151]}"(qf[5]), [q6] "+{a[152:155]}"(qf[6]), [q7] "+{a[156:159]}"(qf[7]), [q8] "+{a[160:163]}"(qf[8]), [q9] "+{a[164:167]}"(qf[9]), [q10] "+{a[168:171]}"(qf[10]), [q11] "+{a[172:175]}"(qf[11]), [q12] "+{a[176:179]}"(qf[12]), [q13] "+{a[180:183]}"(qf[13]), [q14] "+{a[184:187]}"(qf[14]), [q15] "+{a[188:191]}"(qf[15])
#ifdef HALVA_STAMP
      , [st0] "=&v"(st_[0]), [st1] "=&v"(st_[1]), [st2] "=&v"(st_[2]), [st3] "=&v"(st_[3]), [st4] "=&v"(st_[4]), [st5] "=&v"(st_[5]), [st6] "=&v"(st_[6]), [st7] "=&v"(st_[7])
#endif
    : [nq_lo] "v"((unsigned)nq_base), [nq_hi] "v"((unsigned)(nq_base >> 32)), [nqrec] "v"(nqrec), [nqsoff0] "v"(nqsoff[0]), [nqsoff1] "v"(nqsoff[1]),
      [nqvA0] "v"(nqv[0][0]), [nqvB0] "v"(nqv[0][1]), [nqvA1] "v"(nqv[1][0]), [nqvB1] "v"(nqv[1][1]), [rows8] "s"(rows8), [nqg0] "v"(nqg0), [nqg1] "v"(nqg1),
      [o_lo] "v"((unsigned)o_base), [o_hi] "v"((unsigned)(o_base >> 32)), [lse_lo] "v"((unsigned)lse_base), [lse_hi] "v"((unsigned)(lse_base >> 32)),
      [ooffc] "v"(ooffc), [rows8o] "s"(rows8o), [nt01] "s"(nt01), [loff0] "v"(loff[0]), [loff1] "v"(loff[1]), [rowrel] "v"(rowrel), [colrel] "v"(colrel), [voff] "v"(voff), [rsA] "v"(rsA),
      [rsB0] "v"(rsB[0]), [rsB1] "v"(rsB[1]), [k_lo] "v"((unsigned)k_base),
      [k_hi] "v"((unsigned)(k_base >> 32)), [nrec] "v"(nrec), [soff0] "v"(soff0), [nk_lo] "v"((unsigned)nk_base), [nk_hi] "v"((unsigned)(nk_base >> 32)),
      [nnrec] "v"(nnrec), [nsoff0] "v"(nsoff0), [vdlo] "s"(vdlo), [sc] "s"(sc), [n01] "s"(n01),
      [n23] "s"(n23), [nreq] "s"(nreq), [jlo] "s"(jlo), [wave] "s"(wave_u), [piece] "s"(piece), [ctl] "s"(ctl0)
    :
#include "sdpa_fwd3_loop_clobbers.inc"
);
