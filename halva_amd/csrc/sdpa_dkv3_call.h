// One call of the generated block (sdpa_dkv3_loop.inc) for the steps t .. t_side of the current item - included TWICE by sdpa_bwd_dkv3_items
// (sdpa_dkv3.h): once for the item's first call (DKV3_FIRST 1), where the accumulators are outputs only (DKV3_ACC_MOD "=": the block zeroes them
// itself) and the call's run lengths come ready-made from the item's record (sdpa_dkv3_items.h), and once inside the loop over the rare further
// calls (a key block cut by a branch point), where they are read and written ("+") and the run lengths are worked out here.  With ONE site inside a
// loop the accumulators became a loop-carried value that the compiler kept in vector registers: 128 copies out of and 128 into the accumulator
// file around every call.
{
            const int qt0 = q_begin + t * BQ;
#if DKV3_FIRST
            const bool q_in_b = DKV3_F(rec, Q_IN_B0) != 0;
            const int t_side = DKV3_F(rec, T_SIDE), ndma = DKV3_F(rec, NDMA);
            const unsigned n02_u = (unsigned)DKV3_F(rec, N02), n1_u = (unsigned)DKV3_F(rec, N1);
            // the next tile to request: tile 3 - or tile 0 on a cold first call, which requests tiles 0..2 up front
            const bool cold = !prefetched;
            const unsigned q_soff = (unsigned)(cold ? DKV3_F(rec, Q_SOFF0) : DKV3_F(rec, Q_SOFF3)), do_soff = (unsigned)(cold ? DKV3_F(rec, DO_SOFF0) : DKV3_F(rec, DO_SOFF3));
            const unsigned st_soff = (unsigned)(cold ? DKV3_F(rec, ST_SOFF0) : DKV3_F(rec, ST_SOFF3));
            // the call's control word: first call of the block | tiles it requests itself | which of those is the partial last tile
            // | the run's last request is the partial last tile | ring slot of the first step's tile | mail-box slot   (gen_dkv3_loop.py: CTL_*)
            const unsigned ctl_u = dkv3_uni((unsigned)DKV3_F(rec, CTL0) | ((cold ? (unsigned)min(3, ntiles) : 0u) << 1) | ((unsigned)(ring_base & 3) << 9) |
                                            ((unsigned)(round & 1) << 11));
            const unsigned long long ds_ptr = ds0;
#else
            const bool q_in_b = qt0 >= br.b;                     // br.b and qt0 are multiples of 64: uniform over the step
            // ONE asm block for the steps up to the branch point (or the end): masked steps (the diagonal), interior steps, masked steps (the tail)
            const Dkv3Call cp = dkv3_call_params<CAUSAL>(t, q_begin, kblk_min, len, ntiles, DKV3_F(rec, LAST_PARTIAL), br);
            const int t_side = cp.t_side, ndma = cp.ndma;
            const unsigned n02_u = dkv3_uni((unsigned)cp.n0 | ((unsigned)cp.n2 << 16)), n1_u = dkv3_uni((unsigned)cp.n1);      // (the launcher keeps T < 2^22)
            const int tq = t + 3;
            const unsigned q_soff = (unsigned)((q_begin + (int64_t)tq * BQ) * p.ld_qkv * 2), do_soff = (unsigned)((q_begin + (int64_t)tq * BQ) * p.ld_do * 2);
            const unsigned st_soff = (unsigned)((q_begin / BQ + tq) * 512);      // the first step whose statistics this call requests (512 bytes per step)
            const unsigned ctl_u = dkv3_uni((7u << 3) | ((cp.part ? 1u : 0u) << 8) | ((unsigned)((ring_base + t) & 3) << 9));
            const unsigned long long ds_ptr = ds0 + (unsigned long long)t * 16384;
#endif
            const int n = t_side - t;
            const bool lane_off = !k_valid || (q_in_b && key_hidden);
            const unsigned lo0 = (unsigned)(kl - qt0 - 4 * h), range = lane_off ? 0u : (unsigned)(len - kl);
            // the next item's first tiles ride on this call's last three steps (which have no tile of their own left to ask for) - when it has three
            const unsigned pf = (t_side == ntiles && n - ndma == 3) ? (unsigned)npro_next : 0u;
            requested_next = requested_next || pf != 0;
            const unsigned ndma_u = dkv3_uni((unsigned)ndma | (pf << 8));
#ifdef HALVA_STAMP
            if (t == 0) DKV3_NOW(t1_);
#endif
            asm volatile(
#include "sdpa_dkv3_loop.inc"
                : [accV0] DKV3_ACC_MOD "{a[0:15]}"(accV[0]), [accV1] DKV3_ACC_MOD "{a[16:31]}"(accV[1]), [accV2] DKV3_ACC_MOD "{a[32:47]}"(accV[2]), [accV3] DKV3_ACC_MOD "{a[48:63]}"(accV[3]),
                  [accK0] DKV3_ACC_MOD "{a[64:79]}"(accK[0]), [accK1] DKV3_ACC_MOD "{a[80:95]}"(accK[1]), [accK2] DKV3_ACC_MOD "{a[96:111]}"(accK[2]), [accK3] DKV3_ACC_MOD "{a[112:127]}"(accK[3]),
                  [drawn] "=&v"(drawn_out)
                : [sched_ptr] "v"(home_counter), [rec_ptr] "v"(rec_ptr), [rowrel] "v"(rowrel), [colrel] "v"(colrel), [statrel] "v"(statrel), [voff_q] "v"(voff_q), [voff_do] "v"(voff_do),
                  [sc] "s"(sc), [n02] "s"(n02_u), [n1] "s"(n1_u), [ndma] "s"(ndma_u), [wave] "s"(wave_u), [q_piece] "s"(q_piece), [do_piece] "s"(do_piece),
                  [lo0] "v"(lo0), [range] "v"(range), [stat_voff] "v"(stat_voff),
                  [ctl] "s"(ctl_u),
                  // the item's record and the next item's (lane l = dword l): the block picks the descriptor bases / extents with v_readlane_b32
                  [rec] "v"(rec), [nrec] "v"(nrec),
                  // what changes from call to call, as uniform values in vector registers (the block reads them with v_readfirstlane: scalar operands are scarce)
                  [q_soff] "v"(q_soff), [do_soff] "v"(do_soff), [st_soff] "v"(st_soff), [ds_lo] "v"((unsigned)ds_ptr), [ds_hi] "v"((unsigned)(ds_ptr >> 32))
                :
#include "sdpa_dkv3_loop_clobbers.inc"
            );
            t = t_side;
#ifdef HALVA_STAMP
            DKV3_NOW(t2_);
#endif
}
