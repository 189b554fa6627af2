// One call of the generated block (sdpa_fwd3_loop.inc) for the current row block - included TWICE by sdpa_fwd3_item (sdpa_fwd3.h): the ordinary
// pass and the rare repeat against the true row maxima (FWD3_CTL: the control word; FWD3_NREQ / FWD3_KPTR: the walk as this call requests it).  The accumulators are outputs only: the block zeroes them.
asm volatile(
#ifdef HALVA_STAMP
#include "sdpa_fwd3_loop_stamp.inc"
#else
#include "sdpa_fwd3_loop.inc"
#endif
    : [o0] "={a[0:15]}"(acc[0]), [o1] "={a[16:31]}"(acc[1]), [o2] "={a[32:47]}"(acc[2]), [o3] "={a[48:63]}"(acc[3]), [o4] "={a[64:79]}"(acc[4]),
      [o5] "={a[80:95]}"(acc[5]), [o6] "={a[96:111]}"(acc[6]), [o7] "={a[112:127]}"(acc[7]), [l0] "=&v"(l[0]), [mx0] "=&v"(mx[0]), [mr0] "=&v"(mr[0]),
      [l1] "=&v"(l[1]), [mx1] "=&v"(mx[1]), [mr1] "=&v"(mr[1])
#ifdef HALVA_STAMP
      , [st0] "=&v"(st_[0]), [st1] "=&v"(st_[1]), [st2] "=&v"(st_[2]), [st3] "=&v"(st_[3]), [st4] "=&v"(st_[4]), [st5] "=&v"(st_[5]), [st6] "=&v"(st_[6])
#endif
    : [q0] "{a[128:131]}"(qf[0]), [q1] "{a[132:135]}"(qf[1]), [q2] "{a[136:139]}"(qf[2]), [q3] "{a[140:143]}"(qf[3]), [q4] "{a[144:147]}"(qf[4]), [q5] "{a[148:151]}"(qf[5]), [q6] "{a[152:155]}"(qf[6]), [q7] "{a[156:159]}"(qf[7]), [q8] "{a[160:163]}"(qf[8]), [q9] "{a[164:167]}"(qf[9]), [q10] "{a[168:171]}"(qf[10]), [q11] "{a[172:175]}"(qf[11]), [q12] "{a[176:179]}"(qf[12]), [q13] "{a[180:183]}"(qf[13]), [q14] "{a[184:187]}"(qf[14]), [q15] "{a[188:191]}"(qf[15]),
      [rowrel] "v"(rowrel), [colrel] "v"(colrel), [voff] "v"(voff),
      [alt0] "v"(alt[0]), [alt1] "v"(alt[1]), [alt2] "v"(alt[2]), [alt3] "v"(alt[3]), [rsA] "v"(rsA),
      [rsB0] "v"(rsB[0]), [rsB1] "v"(rsB[1]), [mri0] "v"(mri0), [mri1] "v"(mri1), [k_lo] "v"((unsigned)FWD3_KPTR),
      [k_hi] "v"((unsigned)(FWD3_KPTR >> 32)), [nk_lo] "v"((unsigned)nk_ptr), [nk_hi] "v"((unsigned)(nk_ptr >> 32)),
      [safe_k] "s"(safe_k), [vdlo] "s"(vdlo), [vdhi] "s"(vdhi), [sc] "s"(sc), [n01] "s"(n01),
      [n23] "s"(n23), [nreq] "s"(FWD3_NREQ), [jlo] "s"(jlo), [jhi] "s"(jhi), [wave] "s"(wave_u), [piece] "s"(piece), [ctl] "s"(FWD3_CTL)
    :
#include "sdpa_fwd3_loop_clobbers.inc"
);
