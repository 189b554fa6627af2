// sdpa_bwd_dkv3's ITEM RECORDS (round 6) - included by sdpa.hip inside its anonymous namespace, in front of the delta pass that writes them.
//
// An item of sdpa_bwd_dkv3 (sdpa_dkv3.h) is one (sequence, head, key block of 128).  Up to round 5 every persistent workgroup re-derived, per
// item and in compiler-generated code with nothing to overlap it, all that is uniform about the item: thread 0 turned the work queue's answer into
// (sequence, head, key block) with three divisions, looked the sequence up, posted eight words to an LDS mail box; every wave collected them, redid
// the geometry, the run lengths of the masked / plain / masked phases, eight 64-bit base addresses, the extents of three buffer descriptors -
// ~1 450 instructions at 4-5 cycles each = 6 600 cycles in front of every generated block (profiles/r06_dkv3_anatomy.log).
// All of it is a function of the launch parameters and the item's index alone, so the delta pass of the SAME C-ABI call (sdpa_bwd_delta_kernel, which
// runs first anyway) now writes it once, as one 256-byte record per item in queue order; the workgroups fetch a record with one 16-lane load inside
// the generated block (under a wait that exists anyway), hand it round through LDS and pick fields with v_readlane_b32.
// Replaces (with sdpa_dkv3.h) the backward of flash_attn_varlen_qkvpacked_func, reference llava/train/llama_flash_attn_monkey_patch.py:85-91.
#include "sdpa_dkv3_loop_rec.inc"      // enum Dkv3Rec: the dword index of every field (written by gen_dkv3_loop.py, which reads the same fields in asm)

__device__ __host__ __forceinline__ int dkv3_queue_len(int x, int G, int nkb) { return x < G ? ((G - x + 7) >> 3) * nkb : 0; }
// first record of queue x (the queues lie one behind the other)
__device__ __host__ __forceinline__ int dkv3_queue_base(int x, int G, int nkb) {
    int b = 0;
#pragma unroll 1
    for (int y = 0; y < x; ++y) b += dkv3_queue_len(y, G, nkb);
    return b;
}
// item j of queue x -> (pair g = sequence * H + head, key block).  Queue x holds the pairs x, x + 8, ...; inside it, by `order`:
//   0  pair-major;   1  key-block major over the whole queue (measured: loses the L2's reuse of Q / dO, +30 % per step);
//   2  the long half of every pair, pair by pair, then the short halves key-block major: what is left for the end is short (the default)
__device__ __forceinline__ void dkv3_item_of(int x, int j, int G, int nkb, int order, int& g, int& kb) {
    const int ng = (G - x + 7) >> 3;
    int gi;
    if (order == 1) {
        kb = j / ng, gi = j - kb * ng;
    } else if (order == 2) {
        const int hl = (nkb + 1) >> 1, nl = ng * hl;
        if (j < nl) {
            gi = j / hl, kb = j - gi * hl;
        } else {
            const int q = (j - nl) / ng;
            kb = hl + q, gi = j - nl - q * ng;
        }
    } else {
        gi = j / nkb, kb = j - gi * nkb;
    }
    g = x + 8 * gi;
}

// The steps t .. t_side of a key block run as ONE call of the generated block: n0 masked steps (the diagonal), n1 interior steps (every key visible
// to every row), n2 masked steps (sequence tail / branch edge); a key block cut by a branch point takes a second call behind br.b.  ndma: the steps of
// the call that request tile t' + 3; part: the last of those is the sequence's partial last tile.
struct Dkv3Call {
    int n0, n1, n2, t_side, ndma, part;
};
template <bool CAUSAL>
__device__ __forceinline__ Dkv3Call dkv3_call_params(int t, int q_begin, int kblk_min, int len, int ntiles, int last_partial, const Branch br) {
    constexpr int BQ = 64;
    const int qt0 = q_begin + t * BQ;
    const bool q_in_b = qt0 >= br.b;                     // br.b and qt0 are multiples of 64: uniform over the step
    Dkv3Call c;
    c.t_side = q_in_b ? ntiles : min(ntiles, (int)(((int64_t)br.b - q_begin + BQ - 1) / BQ));      // first step at or behind br.b
#ifdef HALVA_DKV3_ALL_MASKED      // diagnostic: every step through the masked phase (same results: an interior step's masks pass everything)
    c.n0 = c.t_side - t, c.n1 = 0, c.n2 = 0;
#else
    {      // interior(t') on this side of br.b = side_ok && t_diag <= t' < t_full: three runs, no scan
        const bool all_valid = kblk_min >= 0 && kblk_min + 128 <= len;      // workgroup-uniform: no padded key in the block
        const bool side_ok = all_valid && !(q_in_b && kblk_min < br.b && kblk_min + 127 >= br.a);
        const int t_diag = CAUSAL ? max(0, (kblk_min + 127 - q_begin + BQ - 1) / BQ) : 0;      // first t' with qt0 >= kblk_min + 127
        const int t_full = max(0, (len - q_begin) / BQ);                                          // first t' with qt0 + 64 > len
        const int lo = min(c.t_side, max(t, t_diag)), hi = min(c.t_side, max(lo, t_full));
        c.n0 = side_ok ? lo - t : c.t_side - t;
        c.n1 = side_ok ? hi - lo : 0;
        c.n2 = c.t_side - t - c.n0 - c.n1;
    }
#endif
    const int n = c.t_side - t;
    c.ndma = min(n, max(0, ntiles - 3 - t));
    c.part = last_partial && c.ndma > 0 && (t + c.ndma - 1 + 3 == ntiles - 1);
    return c;
}

// Record `idx` of the launch (0 <= idx <= total; record `total` is the all-zero "the queues are empty" record), by one thread.
template <bool CAUSAL>
__device__ __forceinline__ void dkv3_build_record(const SdpaParams& p, int idx, int total, int* out) {
    constexpr int D = 128, BQ = 64;
    int r[DKV3_REC_DWORDS];
#pragma unroll
    for (int i = 0; i < DKV3_REC_DWORDS; ++i) r[i] = 0;
    if (idx < total) {
        const int G = p.npairs, nkb = p.nblk;
        int x = 0, j = idx;
        while (x < 7 && j >= dkv3_queue_len(x, G, nkb)) j -= dkv3_queue_len(x, G, nkb), ++x;
        int g, kb;
        dkv3_item_of(x, j, G, nkb, p.sched_order, g, kb);
        const int s = g / p.H, hd = g - s * p.H;
        const int start = p.seq_start ? p.seq_start[s] : 0, len = p.seq_len ? p.seq_len[s] : p.T;
        const Branch br = load_branch(p, s);
        const int kblk_min = kb * 128 - start;                                  // first key of the block in sequence coordinates
        const int q_begin = CAUSAL ? max(0, kblk_min) / BQ * BQ : 0;            // first query row that sees it
        const bool block_has_keys = (kblk_min < len) && (kblk_min + 128 > 0);
        const int q_stop = (kblk_min >= br.a && kblk_min + 127 < br.b) ? min(len, br.b) : len;
        const int ntiles = (block_has_keys && q_stop > q_begin) ? (q_stop - q_begin + BQ - 1) / BQ : 0;      // its 64-row steps
        const int last_rows = len - (q_begin + (ntiles - 1) * BQ);              // >= 64: whole (or the block's rows stop at br.b)
        const int last_partial = last_rows < BQ;
        const Dkv3Call c = dkv3_call_params<CAUSAL>(0, q_begin, kblk_min, len, ntiles, last_partial, br);
        r[DKV3_REC_VALID] = 1, r[DKV3_REC_S] = s, r[DKV3_REC_HD] = hd, r[DKV3_REC_KB] = kb, r[DKV3_REC_START] = start, r[DKV3_REC_LEN] = len;
        r[DKV3_REC_BR_A] = br.a, r[DKV3_REC_BR_B] = br.b, r[DKV3_REC_KBLK_MIN] = kblk_min, r[DKV3_REC_Q_BEGIN] = q_begin, r[DKV3_REC_NTILES] = ntiles;
        r[DKV3_REC_LAST_PARTIAL] = last_partial, r[DKV3_REC_LR] = last_partial ? last_rows : BQ;
        // the block's first three tiles are whole ones: the previous item's asm block may request them on its way out
        r[DKV3_REC_PREFETCHABLE] = (ntiles > 0 && !(last_partial && ntiles <= 3)) ? min(3, ntiles) : 0;
        r[DKV3_REC_N02] = (int)((unsigned)c.n0 | ((unsigned)c.n2 << 16)), r[DKV3_REC_N1] = c.n1, r[DKV3_REC_T_SIDE] = c.t_side, r[DKV3_REC_NDMA] = c.ndma;
        r[DKV3_REC_PART] = c.part;
        // the first call's control word without its run-time bits (cold tiles, ring slot, mail-box slot): gen_dkv3_loop.py:CTL_*
        r[DKV3_REC_CTL0] = (int)(1u | (((last_partial && ntiles <= 3) ? (unsigned)(ntiles - 1) : 7u) << 3) | ((c.part ? 1u : 0u) << 8));
        const int64_t seq_row0 = (int64_t)s * p.T, qrow0 = seq_row0 + start;
        auto put64 = [&](int at, unsigned long long v) { r[at] = (int)(unsigned)v, r[at + 1] = (int)(unsigned)(v >> 32); };
        // the sequence's Q / dO rows of this head as buffer descriptors: base row, bytes up to the end of the last row; *_soff0 / *_soff3: tile 0 / tile 3
        put64(DKV3_REC_Q_LO, (unsigned long long)(size_t)(p.q + hd * D + qrow0 * p.ld_qkv));
        put64(DKV3_REC_DO_LO, (unsigned long long)(size_t)(p.d_o + hd * D + qrow0 * p.ld_do));
        r[DKV3_REC_Q_REC] = len > 0 ? (int)(unsigned)((int64_t)(len - 1) * p.ld_qkv * 2 + D * 2) : 0;
        r[DKV3_REC_DO_REC] = len > 0 ? (int)(unsigned)((int64_t)(len - 1) * p.ld_do * 2 + D * 2) : 0;
        r[DKV3_REC_Q_SOFF0] = (int)(unsigned)((int64_t)q_begin * p.ld_qkv * 2), r[DKV3_REC_DO_SOFF0] = (int)(unsigned)((int64_t)q_begin * p.ld_do * 2);
        r[DKV3_REC_Q_SOFF3] = (int)(unsigned)((q_begin + (int64_t)3 * BQ) * p.ld_qkv * 2), r[DKV3_REC_DO_SOFF3] = (int)(unsigned)((q_begin + (int64_t)3 * BQ) * p.ld_do * 2);
        // the statistics of this (sequence, head): 512 bytes per 64-row step in sequence coordinates (sdpa_bwd_delta_kernel)
        put64(DKV3_REC_ST_LO, (unsigned long long)(size_t)(p.lse2 + ((int64_t)s * p.H + hd) * p.stat_nt * 128));
        r[DKV3_REC_ST_REC] = ((len + BQ - 1) / BQ) * 512, r[DKV3_REC_ST_SOFF0] = (q_begin / BQ) * 512, r[DKV3_REC_ST_SOFF3] = (q_begin / BQ + 3) * 512;
        // dS of the block's first step, wave 0's strip (a wave adds 4096 wave, a step 16384)
        put64(DKV3_REC_DS_LO, (unsigned long long)(size_t)(p.ds_ws + ((((int64_t)s * p.H + hd) * p.ds_nkb + kb) * p.ds_nt + q_begin / BQ) * 16384));
        // K of the sequence's first key, this head (a lane adds its key's row and 16 h bytes; V lies p.v - p.k behind); dK of the block's first key row
        put64(DKV3_REC_K_LO, (unsigned long long)(size_t)(p.k + qrow0 * p.ld_qkv + hd * D));
        put64(DKV3_REC_DK_LO, (unsigned long long)(size_t)(p.dk + (seq_row0 + (int64_t)kb * 128) * p.ld_qkv + hd * D));
        for (int w = 0; w < 4; ++w) {
            const int wrow0 = kb * 128 + 32 * w;
            r[DKV3_REC_ROPE_POS0 + w] = rope_position(min(wrow0, p.T - 1), br);
            r[DKV3_REC_ROWS_OK0 + w] = min(32, max(0, p.T - wrow0));
        }
        r[DKV3_REC_Q_IN_B0] = q_begin >= br.b, r[DKV3_REC_ALL_VALID] = kblk_min >= 0 && kblk_min + 128 <= len;
    }
    typedef int i4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < DKV3_REC_DWORDS / 4; ++i) reinterpret_cast<i4*>(out)[i] = i4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
}
