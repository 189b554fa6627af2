// sdpa_fwd3: the causal forward (head_dim 128) with ONE wave per SIMD - included by sdpa.hip inside its anonymous namespace.
//
// Replaces sdpa_fwd_kernel<128, true> for flash_attn_varlen_qkvpacked_func's forward (reference llava/train/llama_flash_attn_monkey_patch.py:85-91).
// Why another structure: sdpa_fwd_kernel puts two 256-register waves of 32 query rows on every SIMD; they meet at the matrix pipe, in
// their softmax stretches and at the tile's barrier in lock step: ~5 100 cycles per 64 MFMAs and SIMD for 2 048 cycles of matrix work (DESIGN
// 6, 6b).  Here a workgroup is 4 waves = 256 query rows, a wave owns 64 rows (two groups of 32) with the Q fragments (64 registers) and the
// O^T accumulators (128) in the accumulator file and runs the whole tile step itself, alone on its SIMD: S^T = K Q^T puts the QUERY on the
// lane, so the softmax statistics need no cross-lane work inside the loop, and both groups share every K / V^T operand read from LDS.
// hipcc cannot schedule that (experiments/fwd3: 3 565 cycles per step against 2 804 hand-placed), so every tile step of a row block runs
// inside ONE generated inline-asm block (sdpa_fwd3_loop.inc <- gen_fwd3_loop.py): prologue (requests of the first three K / V tiles, zeroed
// accumulators, the first score half-tile, the exponent reference), the iterations (plain | masked bodies), the last O product.
// The exponent reference m_ref of a query is FIXED for the row block: P = exp2(S sc - m_ref), O^T and l accumulate un-rescaled (O^T lives in the
// accumulator file, which the vector unit cannot touch) and NOTHING is tracked inside the loop.  m_ref comes from the row's first 32 visible
// keys; any reference within ~100 log2 units of the row's true maximum gives the same result.  At the end of the pass the workgroup votes on
// the row SUMS: a partial sum not below 2^100 repeats the WHOLE row block, inside the same asm block, with the reference of those rows raised
// by 120, at most MAX_REDO = 64 times (gen_fwd3_loop.py; test_sdpa_exponent_reference_moves_when_later_keys_dominate).  A finite row whose
// maximum lies further out than 64 x 120 log2 units (~5 300 nats above its first keys) leaves this kernel with l = inf: NaN output rows, lse = inf -
// loud, not silently wrong.  HALVA_FWD3_REPAIR=1 makes launch_fwd follow every sdpa_fwd3 launch with the running-maximum kernel in repair mode
// (SdpaParams::repair), which redoes exactly the row blocks that hold a non-finite lse with no bound at all (as flash-attn); it is OFF by default:
// the pass costs ~30 us per launch (4 % of the forward: 2 048 workgroups that each read their lse rows and leave) for inputs no training run produces
// (test_sdpa_forward_far_beyond_the_repeat_budget_is_repaired).
// LDS: K ring [4][64][128] bf16 at 0, V ring at 64 KiB, the workgroup's vote words, 4 KiB per wave for the transposition of its output rows.

constexpr int FWD3_TILE = 64 * 128 * 2;
constexpr int FWD3_MAIL = 8 * FWD3_TILE;      // (gen_fwd3_loop.py: MAIL_LDS)
constexpr int FWD3_OSTAGE = FWD3_MAIL + 128;      // (gen_fwd3_loop.py: OSTAGE_LDS) 4 x 4 KiB: the waves' staging areas for their output rows
constexpr int FWD3_LDS = FWD3_OSTAGE + 4 * 4096;

// Everything the generated block needs to know about one (sequence, head, 256-row block) item; wave-uniform unless noted.
struct Fwd3Geom {
    int N;                 // tiles of the walk (0: the block sees no key)
    int kv_first;          // first key of the first walked tile (local)
    int n1req;             // requests before the walk's jump (0xffff: none)
    int64_t jump_rows;     // rows the walk jumps over
    int partial, lr;       // the last walked tile is the sequence's partial last tile: lr of its 64 rows exist
    int n0, n1, n2, n3;    // per WAVE: [plain n0][masked n1][plain n2][masked n3]
    int kvA, kvB;          // first key of the first tile of masked run 1 / 2
    int wave_in_b, wq_min;
};

__device__ __forceinline__ Fwd3Geom fwd3_geom(const SdpaParams& p, int qb, int start, int len, const Branch& br, int wave) {
    constexpr int BN = 64, BM = 256;
    Fwd3Geom g;
    const int g0 = qb * BM;
    const int kv_end = min(len, g0 + BM - start);
    const int ntiles = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;
    int skip_lo = ntiles, skip_hi = ntiles;
    const bool wholly_b = g0 - start >= br.b;
    if (wholly_b) {
        skip_lo = min(ntiles, (br.a + BN - 1) / BN);
        skip_hi = max(skip_lo, min(ntiles, br.b / BN));
    }
    g.N = skip_lo + (ntiles - skip_hi);
    g.kv_first = skip_lo > 0 ? 0 : skip_hi * BN;
    const bool jumps = skip_lo > 0 && skip_hi > skip_lo && skip_hi < ntiles;
    g.n1req = jumps ? skip_lo : 0xffff;
    g.jump_rows = jumps ? (int64_t)(skip_hi - skip_lo) * BN : 0;
    const int last_kv0 = (skip_hi < ntiles ? ntiles - 1 : skip_lo - 1) * BN;      // (N > 0)
    g.partial = g.N > 0 && last_kv0 + BN > len;
    g.lr = g.partial ? len - last_kv0 : BN;
    // this wave's rows: wq_min .. wq_min + 63 (local)
    g.wq_min = g0 + 64 * wave - start;
    g.wave_in_b = g.wq_min >= br.b;
    int cntA = 0, kvB0 = 0;
    if (g.wave_in_b) {      // the tiles in front of br.b are seen "up to br.a"
        cntA = wholly_b ? skip_lo : min(ntiles, br.b / BN);
        kvB0 = wholly_b ? skip_hi * BN : br.b;
    }
    const int cntB = g.N - cntA;
    g.n0 = min(cntA, br.a / BN);
    g.n1 = cntA - g.n0;
    g.kvA = BN * g.n0;
    const int fullB = min(len, g.wq_min + 1);      // keys every row of the wave sees
    g.n2 = min(cntB, max(0, fullB - kvB0) / BN);
    g.n3 = cntB - g.n2;
    g.kvB = kvB0 + BN * g.n2;
    return g;
}

// A workgroup's position in its static sequence of row blocks: virtual block vb = blockIdx.x + k * gridDim.x (the blocks the static launch of
// sdpa_fwd_kernel would have started: map_block / paired_blocks - heavy block first, then the light one), all wave-uniform.
struct Fwd3Cursor {
    int vb, which, first, second, cur_qb, s, hd, start, len, seen_s;      // seen_s: the sequence start / len / br were loaded for (-1: none yet)
    Branch br;
    __device__ __forceinline__ int qb() const { return cur_qb; }      // (a stored scalar: `which ? second : first` was compiled into a scratch array)
};
__device__ __forceinline__ void fwd3_cursor_load(const SdpaParams& p, Fwd3Cursor& c, int total) {
    if (c.vb >= total) return;
    int b;
    map_block(c.vb, (p.nblk + 1) / 2, p.H, p.npairs, false, c.s, c.hd, b);
    // the sequence's geometry by SCALAR loads (they return through lgkmcnt): as vector loads the compiler waits for each of them with vmcnt(0) - it
    // knows nothing of what the asm blocks have in flight -, i.e. for the rows just stored and the tiles requested for the next item
    const int sq = (int)dkv3_uni((unsigned)c.s);
    if (sq != c.seen_s) {      // (a workgroup's consecutive virtual blocks mostly belong to one sequence)
        // four loads, ONE wait (an array the launch does not have reads a word of q instead and takes its default afterwards)
        const int32_t* dummy = reinterpret_cast<const int32_t*>(p.q);
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(c.start), "=&s"(c.len), "=&s"(c.br.a), "=&s"(c.br.b)
                     : "s"(dkv3_uni64(p.seq_start ? p.seq_start + sq : dummy)), "s"(dkv3_uni64(p.seq_len ? p.seq_len + sq : dummy)),
                       "s"(dkv3_uni64(p.br_a ? p.br_a + sq : dummy)), "s"(dkv3_uni64(p.br_b ? p.br_b + sq : dummy))
                     : "memory");
        if (!p.seq_start) c.start = 0;
        if (!p.seq_len) c.len = p.T;
        if (!p.br_a) c.br.a = 0x7fffffff;
        if (!p.br_b) c.br.b = 0x7fffffff;
        c.seen_s = sq;
    }
    paired_blocks(p.nblk, c.start, c.br, b, c.first, c.second);
    c.which = 0, c.cur_qb = c.first;
}
__device__ __forceinline__ void fwd3_cursor_advance(const SdpaParams& p, Fwd3Cursor& c, int total) {
    if (c.which == 0 && c.second != c.first) {
        c.which = 1, c.cur_qb = c.second;
    } else {
        c.vb += gridDim.x;
        fwd3_cursor_load(p, c, total);
    }
}

// The Q fragments of one item, asked for by hand straight into the registers the block reads them from (a row outside the sequence reads the
// nearest one inside: its result is not stored): for the workgroup's first item in front of the loop, for every other one behind its
// predecessor's block, in front of the predecessor's row stores.  Nobody waits for them before the block's own counted wait.
__device__ __forceinline__ const bf16_t* fwd3_q_row(const SdpaParams& p, const Fwd3Cursor& c, int wave, int lane, int gi) {
    const int ql = c.qb() * 256 + 64 * wave + 32 * gi + (lane & 31) - c.start;
    return p.q + ((int64_t)c.s * p.T + c.start + min(max(ql, 0), max(c.len - 1, 0))) * p.ld_qkv + c.hd * 128 + 8 * (lane >> 5);
}
__device__ __forceinline__ void fwd3_load_q(u32x4 (&qf)[16], const bf16_t* q0, const bf16_t* q1) {
    asm volatile(
        "global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:32\n\tglobal_load_dwordx4 %2, %16, off offset:64\n\t"
        "global_load_dwordx4 %3, %16, off offset:96\n\tglobal_load_dwordx4 %4, %16, off offset:128\n\tglobal_load_dwordx4 %5, %16, off offset:160\n\t"
        "global_load_dwordx4 %6, %16, off offset:192\n\tglobal_load_dwordx4 %7, %16, off offset:224\n\t"
        "global_load_dwordx4 %8, %17, off\n\tglobal_load_dwordx4 %9, %17, off offset:32\n\tglobal_load_dwordx4 %10, %17, off offset:64\n\t"
        "global_load_dwordx4 %11, %17, off offset:96\n\tglobal_load_dwordx4 %12, %17, off offset:128\n\tglobal_load_dwordx4 %13, %17, off offset:160\n\t"
        "global_load_dwordx4 %14, %17, off offset:192\n\tglobal_load_dwordx4 %15, %17, off offset:224"
        : "={a[128:131]}"(qf[0]), "={a[132:135]}"(qf[1]), "={a[136:139]}"(qf[2]), "={a[140:143]}"(qf[3]), "={a[144:147]}"(qf[4]), "={a[148:151]}"(qf[5]),
          "={a[152:155]}"(qf[6]), "={a[156:159]}"(qf[7]), "={a[160:163]}"(qf[8]), "={a[164:167]}"(qf[9]), "={a[168:171]}"(qf[10]), "={a[172:175]}"(qf[11]),
          "={a[176:179]}"(qf[12]), "={a[180:183]}"(qf[13]), "={a[184:187]}"(qf[14]), "={a[188:191]}"(qf[15])
        : "v"(q0), "v"(q1)
        : "memory");
}

// One row block.  `parity` alternates between a workgroup's consecutive items (the vote words are double buffered).  nxt (valid: nxt.vb < total)
// is the workgroup's next item: this item's block requests its first three K / V tiles from its last iterations (which have no tile of their
// own left to ask for) when they are three whole ordinary tiles.  prefetched / ring_base: in - what the previous item did for this one; out - for the next.
__device__ __forceinline__ void sdpa_fwd3_item(const SdpaParams& p, char* smem, const Fwd3Cursor& cur, const Fwd3Cursor& nxt, bool nxt_valid, bool& prefetched,
                                               int& ring_base, u32x4 (&qf)[16], int wave, int lane_in, int parity) {
    constexpr int D = 128, BM = 256;
#ifdef HALVA_STAMP
    unsigned long long stamp_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t0)::"memory");
#endif
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5, r = lane & 31;
    const int s = cur.s, hd = cur.hd, qb = cur.qb(), start = cur.start, len = cur.len;
    const Branch br = cur.br;
    Fwd3Geom g = fwd3_geom(p, qb, start, len, br, wave);
    const int N = (int)dkv3_uni((unsigned)g.N);
    const int64_t seq_row0 = (int64_t)s * p.T;
    const int g0 = qb * BM;
    int ql[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) ql[gi] = g0 + 64 * wave + 32 * gi + r - start;
    // (a block that sees no key, N == 0, takes the same road: no iteration, O^T = 0, l = 0 -> zero rows; its requests go straight to the next item's tiles)
    // ---- lane constants of the generated block (formed per item: kept across it they are spilled, and a scratch reload waits for every request in flight)
    const unsigned rowrel = 2048 * (r >> 3) + 64 * (r & 7) + 16 * (h ^ ((r >> 2) & 3));
    const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, h2 = g16 >> 1;
    const unsigned colrel = 64 * (4 * h2 + q4) + 16 * ((2 * (g16 & 1) + (pp >> 1)) ^ h2) + 8 * (pp & 1);
    const unsigned voff = dkv3_piece_voff(p.ld_qkv, wave, lane, 0, 64);
    // visible-key counts (minus the lane half's row offset 4 h) of the first tile of each masked run, and of the item's first tile
    const int rsA = br.a - g.kvA - 4 * h;
    int rsB[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) rsB[gi] = min(len, ql[gi] + 1) - g.kvB - 4 * h;
    const unsigned first_state = (g.n0 + g.n1 > 0) ? (g.n0 > 0 ? 0u : 1u) : (g.n2 > 0 ? 0u : 2u);      // the first tile: plain | opens masked run 1 | run 2
    const float sc = p.scale * kLog2e;
    const unsigned n01 = dkv3_uni((unsigned)g.n0 | ((unsigned)g.n1 << 16)), n23 = dkv3_uni((unsigned)g.n2 | ((unsigned)g.n3 << 16));
    const unsigned jlo = dkv3_uni((unsigned)(g.jump_rows * p.ld_qkv * 2));
    const unsigned wave_u = dkv3_uni((unsigned)wave), piece = dkv3_uni((unsigned)(16 * p.ld_qkv * 2));
    // K / V tiles arrive through ONE buffer descriptor per (sequence, head): base = the sequence's first K row of this head, num_records = up to the
    // end of its last V row; V rows = K rows + (v - k).  Rows past the sequence are out of range -> zeros (no clamped offsets, no dummy target).
    const unsigned vdlo = dkv3_uni((unsigned)((const char*)p.v - (const char*)p.k));
    auto kv_base = [&](const Fwd3Cursor& c) { return (unsigned long long)(size_t)(p.k + c.hd * D + ((int64_t)c.s * p.T + c.start) * p.ld_qkv); };
    auto kv_records = [&](const Fwd3Cursor& c) { return c.len > 0 ? (unsigned)((int64_t)(c.len - 1) * p.ld_qkv * 2 + vdlo + D * 2) : 0u; };
    const unsigned long long k_base = kv_base(cur);
    const unsigned nrec = kv_records(cur);
    // the walk as the block is to request it: from tile 0 (cold), or from tile 3 when the previous item's block has requested tiles 0..2
    const bool pf = prefetched;
    const int t_req = pf ? 3 : 0;
    const unsigned nreq = dkv3_uni((unsigned)(N - t_req) | ((unsigned)(g.n1req == 0xffff ? 0xffff : g.n1req - t_req) << 16));
    const unsigned soff0 = (unsigned)((int64_t)(g.kv_first + 64 * t_req) * p.ld_qkv * 2);
    // the next item's first three tiles (requested by this block's last iterations when they are three whole ordinary tiles)
    unsigned npf = 0, nnrec = 0, nsoff0 = 0;
    unsigned long long nk_base = k_base;
    if (nxt_valid) {
        const Fwd3Geom gn = fwd3_geom(p, nxt.qb(), nxt.start, nxt.len, nxt.br, wave);
        if (gn.N >= 3 && (gn.n1req == 0xffff || gn.n1req >= 3)) {
            npf = 3;
            nk_base = kv_base(nxt), nnrec = kv_records(nxt), nsoff0 = (unsigned)((int64_t)gn.kv_first * p.ld_qkv * 2);
        }
    }
    npf = dkv3_uni(npf);

    // the NEXT item's Q rows (none: a descriptor without records - zeros): fetched by the block's tail through LDS, one row group after the other.
    // Piece c of a group's half-tile image = rows 8 (c / 2) .. + 7, chunks 8 (c % 2) .. + 7: the scalar offset carries the band, the
    // instruction's immediate the chunk half, the lane offset row-in-band, chunk and the image's swizzle (two variants: band parity).  A row in
    // front of the sequence (left padding) puts its lane beyond the descriptor's range through a negative lane offset: zeros, like a row behind it.
    const Fwd3Cursor& qn = nxt_valid ? nxt : cur;
    const unsigned long long nq_base = (unsigned long long)(size_t)(p.q + qn.hd * D + ((int64_t)qn.s * p.T + qn.start) * p.ld_qkv);
    const unsigned nqrec = (nxt_valid && qn.len > 0) ? (unsigned)((int64_t)(qn.len - 1) * p.ld_qkv * 2 + D * 2) : 0u;
    unsigned nqsoff[2], nqv[2][2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int qlg0 = qn.qb() * BM + 64 * wave + 32 * gi - qn.start;
        nqsoff[gi] = (unsigned)((int64_t)max(qlg0, 0) * p.ld_qkv * 2);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
            nqv[gi][pb] = (unsigned)((int64_t)(min(qlg0, 0) + (r >> 2)) * p.ld_qkv * 2) + 16u * (4u * h + ((lane & 3) ^ ((2 * pb + (r >> 4)) & 3)));
    }
    const unsigned rows8 = dkv3_uni((unsigned)(8 * p.ld_qkv * 2));
    // (a next row block that begins in front of its sequence - left padding - has its fragments gathered row by row with clamped pointers instead)
    const bool q_gather = nxt_valid && qn.qb() * BM < qn.start;
    const bf16_t* nqg0 = fwd3_q_row(p, qn, wave, lane, 0);
    const bf16_t* nqg1 = fwd3_q_row(p, qn, wave, lane, 1);
    // this item's rows: the wave's first output row / lse entry (uniform) + per-lane offsets; the lse offset's low bits say whether the row
    // exists in the tensor (bit 0) and is a row of the sequence (bit 1)
    const unsigned long long o_base = (unsigned long long)(size_t)(p.o + hd * D + (seq_row0 + g0 + 64 * wave) * p.ld_o);
    const unsigned long long lse_base = (unsigned long long)(size_t)(p.lse + ((int64_t)s * p.H + hd) * p.T + g0 + 64 * wave);
    unsigned loff[2];
    int nT[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int gqs = g0 + 64 * wave + 32 * gi + r;
        const bool in_T = gqs < p.T, valid = in_T && ql[gi] >= 0 && ql[gi] < len;
        loff[gi] = 4u * (32 * gi + r) | (in_T ? 1u : 0u) | (valid ? 2u : 0u);
        nT[gi] = min(32, max(0, p.T - (g0 + 64 * wave + 32 * gi)));      // rows of the group that exist in the tensor
    }
    // (the row stores: a store instruction writes rows 8 j + (lane >> 3), 16 bytes (lane & 7) of a 128-byte half row each)
    const unsigned ooffc = (unsigned)((int64_t)(lane >> 3) * p.ld_o * 2) + 16u * (lane & 7);
    const unsigned rows8o = dkv3_uni((unsigned)(8 * p.ld_o * 2)), nt01 = dkv3_uni((unsigned)nT[0] | ((unsigned)nT[1] << 8));
    const unsigned ctl0 = dkv3_uni((unsigned)((ring_base & 3) | (pf ? 16 : 0) | (parity ? 1024 : 0) | (q_gather ? 2048 : 0)) | (npf << 5) | (first_state << 8));
#ifdef HALVA_STAMP
    unsigned st_[8];
    unsigned long long tc_[6];
    tc_[0] = stamp_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc_[1])::"memory");
#endif
#include "sdpa_fwd3_call.h"
#ifdef HALVA_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc_[2])::"memory");
    tc_[3] = tc_[4] = tc_[2];
#endif
    prefetched = npf != 0;
    ring_base = (ring_base + N) & 3;      // (a repeat pass starts from the same slot)
#ifdef HALVA_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc_[5])::"memory");
    if (p.dbg && lane == 0 && blockIdx.x % 9 == 0 && blockIdx.x / 9 < 30 && cur.vb < (int)(2 * gridDim.x)) {      // slot: [block][item][wave][16]: the first four items
        unsigned long long* o_ = p.dbg + (((blockIdx.x / 9) * 4 + 2 * (cur.vb >= (int)gridDim.x) + cur.which) * 4 + wave) * 16;
        for (int i = 0; i < 6; ++i) o_[i] = tc_[i];
        for (int i = 0; i < 8; ++i) o_[6 + i] = st_[i];
        o_[14] = (unsigned long long)N | ((unsigned long long)qb << 16) | ((unsigned long long)blockIdx.x << 32);
        o_[15] = (unsigned long long)g.n0 | ((unsigned long long)g.n1 << 16) | ((unsigned long long)g.n2 << 32) | ((unsigned long long)g.n3 << 48);
    }
#endif
}

// Persistent workgroups, one per CU: workgroup w walks the virtual blocks w, w + G, w + 2 G, ... of the static launch (two row blocks each, ranked by
// work: heavy with light), so that the blocks of one (sequence, head) pair - which stream the same K / V - still run side by side on one XCD, and
// pipelines across them: an item's last iterations request the next item's first tiles.
template <bool CAUSAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sdpa_fwd3_kernel(const SdpaParams p) {
    static_assert(CAUSAL, "sdpa_fwd3 is the causal head_dim-128 instantiation");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = ((p.nblk + 1) / 2) * p.npairs;
    WG_CLOCK_BEGIN();
    Fwd3Cursor cur;
    cur.seen_s = -1;
    cur.vb = blockIdx.x;
    fwd3_cursor_load(p, cur, total);
    bool prefetched = false;
    int ring_base = 0, parity = 0;
    u32x4 qf[16];      // the Q fragments, in a[128:191] from here on: loaded here for the first item, by every block for its successor
    if (cur.vb < total) fwd3_load_q(qf, fwd3_q_row(p, cur, wave, lane, 0), fwd3_q_row(p, cur, wave, lane, 1));
#pragma unroll 1
    while (cur.vb < total) {
        Fwd3Cursor nxt = cur;
        fwd3_cursor_advance(p, nxt, total);
        sdpa_fwd3_item(p, smem, cur, nxt, nxt.vb < total, prefetched, ring_base, qf, wave, lane, parity);
        cur = nxt;
        parity ^= 1;
    }
    WG_CLOCK_END(p.dbg, 0);
}
