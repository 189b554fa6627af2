// sdpa_bwd_dkv3: dK / dV (+ dS store) with ONE wave per SIMD - included by sdpa.hip inside its anonymous namespace (head_dim 128 only).
//
// Replaces sdpa_bwd_dkv2 on the dS-workspace path of halva_sdpa_branch_bwd_ws (the backward of flash_attn_varlen_qkvpacked_func,
// reference llava/train/llama_flash_attn_monkey_patch.py:85-91).  Why another structure: the two-role kernel puts two 256-register waves
// on every SIMD and they meet - at the matrix pipe, in their vector stretches and at the step's barrier - in lock step: ~3 850 cycles
// per 64-row step for 2 048 cycles of matrix work, whatever was done to either wave's own stream.  Here a workgroup is
// 4 waves = 128 keys, a wave owns 32 keys with BOTH accumulators (dK^T, dV^T: 128 accumulator registers), the K and V fragments (64) and
// runs all four products of a step itself, alone on its SIMD with the whole 512-register file.  hipcc cannot schedule that (every MFMA
// destination lands in the accumulator file and each score is copied out before the vector unit may touch it: 3 650 cycles per step), so
// EVERY step of a key block runs inside generated inline-asm (sdpa_dkv3_loop.inc <- gen_dkv3_loop.py, included through sdpa_dkv3_call.h):
// a masked phase (the diagonal steps), the interior phase (whole tile, every key visible to every row: 2 419 cycles per step in
// isolation, bit-identical to the plain code - experiments/dkv3), a masked phase (sequence tail / branch edge).  The plain-HIP twin of
// the step (dkv3_hip_step, HALVA_DKV3_ASM=0) states the same arithmetic readably and is held to the asm bit for bit by a test.
// Both keep the same protocol on a ring of FOUR Q / dO tile slots (+ their lse2 / -delta rows):
//     at the start of step t tile t has landed and is visible to every wave; tiles t+1, t+2 have been requested;
//     at the head of step t tile t+3 is requested into the slot tile t-1 left at the last barrier; the step ends with "tile t+1 has landed" + s_barrier.
// Statistics: the delta pass writes -delta and lse2 = lse * log2(e) (SdpaParams::lse2, the tail of the workspace), so that both are plain
// rows an LDS-DMA dword request can fetch and -delta is directly the initial accumulator of the dP chain.
// dS leaves in the same image sdpa_bwd_dkv2 writes (strip = wave), so sdpa_bwd_dq2 is unchanged.
// The kernel is launched as persistent workgroups (one per CU) that draw (sequence, head, key block) items from a work queue per XCD
// and pipeline across items: sdpa_bwd_dkv3_items below.  What must hold for that pipelining - no compiler instruction touches the K / V
// fragment registers, no scratch - is checked on the compiled kernel by tools/check_dkv3_isa.py (tests/test_cabi_symbols.py).

constexpr int DKV3_TILE = 64 * 128 * 2;                      // one Q or dO tile
constexpr int DKV3_DO = 4 * DKV3_TILE;                       // dO ring behind the Q ring
constexpr int DKV3_LSE = 8 * DKV3_TILE;                      // [4][64] lse2, then [4][64] -delta
constexpr int DKV3_ND = DKV3_LSE + 4 * 64 * 4;              // (the plain-HIP twin's own layout; the generated loop: four 1-KiB slots [lse2 x 64][-delta x 64][unused])
constexpr int DKV3_SCHED = DKV3_LSE + 4096;                  // [2][64] ints: the persistent workgroup's item mail box (two item records, sdpa_dkv3_items.h; gen_dkv3_loop.py:MAIL_LDS)
constexpr int DKV3_STAGE = DKV3_SCHED + 512;                 // [4 waves][4 KiB]: a wave's staging area for its dK / dV rows (dkv3_store_rows_lds)
constexpr int DKV3_LDS = DKV3_STAGE + 4 * 4096;
#ifndef DKV3_ROWS_VIA_LDS
#define DKV3_ROWS_VIA_LDS 1      // 0: the dK / dV rows stored straight from the accumulator layout (store_rows_T), rounds 2-3
#endif

__device__ __forceinline__ unsigned dkv3_uni(unsigned x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned long long dkv3_uni64(const void* ptr) {
    const unsigned long long v = (unsigned long long)(size_t)ptr;
    return (unsigned long long)dkv3_uni((unsigned)v) | ((unsigned long long)dkv3_uni((unsigned)(v >> 32)) << 32);
}

struct Dkv3State {
    u32x4 kq[8], vq[8];          // K / V fragments of this wave's 32 keys (B operands)
    f32x16 accV[4], accK[4];     // dV^T, dK^T: row = d, lane = key
};

// One step in plain HIP, with every mask: the diagonal, the sequence tail, padded keys, the branch edge.
template <bool CAUSAL>
__device__ __forceinline__ void dkv3_hip_step(Dkv3State& st, const char* qt, const char* dot, const float* lse_t, const float* nd_t, char* ds_step,
                                              float sc, int qt0, int len, int kl, bool lane_off, bool masked, int lane) {
    constexpr int D = 128, KS = 8, DT = 4;
    const int h = lane >> 5;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        f32x4 sl[4], sd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sl[j] = *reinterpret_cast<const f32x4*>(lse_t + 32 * sub + 8 * j + 4 * h);
            sd[j] = *reinterpret_cast<const f32x4*>(nd_t + 32 * sub + 8 * j + 4 * h);
        }
        f32x16 x, y;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f, y[r] = sd[r >> 2][r & 3];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x = mfma32(frag_rows<D>(qt, 32 * sub, ks, lane), __builtin_bit_cast(s16x8, st.kq[ks]), x);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) y = mfma32(frag_rows<D>(dot, 32 * sub, ks, lane), __builtin_bit_cast(s16x8, st.vq[ks]), y);
        if (masked) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = qt0 + 32 * sub + acc_row(r, h);
                if (ql >= len || (CAUSAL && kl > ql) || lane_off) x[r] = -INFINITY;      // -> P = 0, dZ = 0
            }
        }
        u32x4 pb[2], zb[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * i], sc, -sl[i >> 1][(2 * i) & 3]));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * i + 1], sc, -sl[i >> 1][(2 * i + 1) & 3]));
            pb[i >> 2][i & 3] = pack_bf16x2(p0, p1);
            zb[i >> 2][i & 3] = pack_bf16x2(p0 * y[2 * i], p1 * y[2 * i + 1]);
        }
        if (ds_step) {      // nontemporal: written once, read once by the dQ kernel
            __builtin_nontemporal_store(zb[0], reinterpret_cast<u32x4*>(ds_step + 2048 * sub + lane * 16));
            __builtin_nontemporal_store(zb[1], reinterpret_cast<u32x4*>(ds_step + 2048 * sub + 1024 + lane * 16));
        }
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i)
            st.accV[i % DT] = mfma32(frag_cols<D, false>(dot, 32 * sub + 16 * (i / DT), 32 * (i % DT), lane), __builtin_bit_cast(s16x8, pb[i / DT]), st.accV[i % DT]);
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i)
            st.accK[i % DT] = mfma32(frag_cols<D, false>(qt, 32 * sub + 16 * (i / DT), 32 * (i % DT), lane), __builtin_bit_cast(s16x8, zb[i / DT]), st.accK[i % DT]);
    }
}

// byte offset, from a tile's first row, of this lane's 16 bytes of the wave's piece i (chunk wave + 4 i of the tile image: TileDma for four
// waves); rows >= nrows repeat row nrows - 1 (a partial last tile)
__device__ __forceinline__ unsigned dkv3_piece_voff(int64_t ld, int wave, int lane, int i, int nrows) {
    const int o = 1024 * (wave + 4 * i) + 16 * lane;
    const int band = o / 2048, rem = o % 2048;
    const int row = 8 * band + ((rem % 512) >> 6);
    const int ch = 4 * (rem / 512) + (((rem >> 4) & 3) ^ ((row >> 2) & 3));
    return (unsigned)((min(row, nrows - 1) * ld + ch * 8) * 2);
}

// A wave's [128 x 32] transposed accumulator (lane = key row, registers = columns d: store_rows_T's layout) as bf16 rows, TRANSPOSED through 4 KiB of
// LDS of the wave's own (round 4; the forward's tail does the same, gen_fwd3_loop.py:tail_code): two passes of 64 columns; a lane writes its eight
// 8-byte pieces of the pass to [row][128 bytes] with the 16-byte chunk index XORed by (row >> 1) & 7, reads back 16 bytes of row 8 j + (lane >> 3),
// chunk lane & 7, and every store instruction writes eight whole 128-byte row pieces - instead of 16 bytes into each of 32 rows, which the CU's address
// unit takes a lane at a time (rows converted and stored: 5 180 cycles per item, profiles/r04_dkv3_anatomy.log).  row0: the wave's first row; rows_ok:
// how many of its 32 rows exist in the tensor.  mul: the scale of the rows (MUL = false: none, the dV rows; a key outside the sequence needs no 0: every
// step of such a key is a masked one, its P is exactly 0 and so are its accumulators).
// ROPE (round 5, the dK rows of halva_sdpa_branch_bwd_rope): the inverse rotation applied on the way (store_rows_T_rope's arithmetic: row rounded to
// bf16, rotated with the bf16 table entries of the KEY's position, rounded again); elements d and d + 64 are the same register of tiles dt and dt + 2.
// rope_cos / rope_sin: the wave's 4-KiB blocks of table rows in LDS (dkv3_rope_request_lds below), landed.
template <bool ROPE, bool MUL>
__device__ __forceinline__ void dkv3_store_rows_lds(char* smem, int wave, bf16_t* row0, int64_t ld, const f32x16 (&acc)[4], float mul, int rows_ok, int lane,
                                                    const char* rope_cos, const char* rope_sin) {
    typedef __attribute__((address_space(3))) char lchar;
    lchar* stage = (lchar*)(smem + DKV3_STAGE + wave * 4096);
    const int r = lane & 31, h = lane >> 5;
    lchar* wr = stage + r * 128 + 8 * h;
    const int wmask = ((r >> 1) & 7) << 4;
    const int rr = lane >> 3, rc = lane & 7;      // read side: row 8 j + rr, chunk rc
    // (row 8 j + rr: (row >> 1) & 7 = ((rr >> 1) + 4 j) & 7 = (rr >> 1) ^ 4 for odd j)
    lchar* rd[2] = {stage + rr * 128 + ((rc ^ ((rr >> 1) & 7)) << 4), stage + rr * 128 + ((rc ^ ((rr >> 1) & 7) ^ 4) << 4)};
    // ROPE: every element rotated ONCE, two at a time (packed f32: a lone wave pays ~4.5 cycles per instruction whatever it does; against the first
    // version - scalar, recomputed per pass - it made no measurable difference: +32 .. +50 us per call either way, profiles/r05_rope_cost.log).  rot[dt][p] = columns 32 dt + 8 (p >> 1) +
    // 4 h + 2 (p & 1) .. + 1 of this lane's row.
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 rot[4][8];
    if (ROPE) {
#pragma unroll
        for (int dtl = 0; dtl < 2; ++dtl)
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                // (words 2 (pr & 1) of chunk 4 dtl + (pr >> 1) of this lane's row; the chunk sits at position chunk ^ (row & 7))
                const int off = r * 128 + (((4 * dtl + (pr >> 1)) ^ (r & 7)) << 4) + 8 * h + 4 * (pr & 1);
                const unsigned cw = *reinterpret_cast<__attribute__((address_space(3))) const unsigned*>((lchar*)(rope_cos + off));
                const unsigned sw = *reinterpret_cast<__attribute__((address_space(3))) const unsigned*>((lchar*)(rope_sin + off)) ^ 0x80008000u;      // (the inverse rotation: s = -sin, exactly)
                const f2 c = {bf16_lo(cw), bf16_hi(cw)}, sn = {bf16_lo(sw), bf16_hi(sw)};
                const f2 m2 = {mul, mul};
                const f2 a1 = f2{acc[dtl][2 * pr], acc[dtl][2 * pr + 1]} * m2, a2 = f2{acc[dtl + 2][2 * pr], acc[dtl + 2][2 * pr + 1]} * m2;
                const unsigned w1 = pack_bf16x2(a1[0], a1[1]), w2 = pack_bf16x2(a2[0], a2[1]);      // the rows as the separate launch would have read them
                const f2 x1 = {bf16_lo(w1), bf16_hi(w1)}, x2 = {bf16_lo(w2), bf16_hi(w2)};
                rope_pair(x1, x2, c, sn, rot[dtl][pr], rot[dtl + 2][pr]);
            }
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int dtl = 0; dtl < 2; ++dtl)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dt = 2 * pass + dtl;
                u32x2 w;
                if (ROPE) {
                    w[0] = pack_bf16x2(rot[dt][2 * g][0], rot[dt][2 * g][1]);
                    w[1] = pack_bf16x2(rot[dt][2 * g + 1][0], rot[dt][2 * g + 1][1]);
                } else {
                    w[0] = MUL ? pack_bf16x2(acc[dt][4 * g + 0] * mul, acc[dt][4 * g + 1] * mul) : pack_bf16x2(acc[dt][4 * g + 0], acc[dt][4 * g + 1]);
                    w[1] = MUL ? pack_bf16x2(acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul) : pack_bf16x2(acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
                }
                *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(wr + (((4 * dtl + g) << 4) ^ wmask)) = w;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(rd[j & 1] + 1024 * j);
            if (8 * j + rr < rows_ok) HALVA_ROW_STORE(reinterpret_cast<u32x4*>(row0 + (int64_t)(8 * j + rr) * ld + 64 * pass + 8 * rc), v);
        }
    }
}
// The table rows of the wave's 32 keys (positions pos0 .. pos0 + 31, consecutive: a 32-key strip never straddles a branch point) = 4 KiB of cos + 4 KiB of
// sin, contiguous in the tables, brought by LDS-DMA (four 1-KiB pieces per table, coalesced) into the ring slot the item's LAST tile has just left -
// the one slot the next item's prefetched tiles do not use; the wave's 4 KiB of its Q half for cos, of its dO half for sin - with the 16-byte chunk
// position XORed by row & 7 on the SOURCE side (LDS-DMA writes lane l at byte 16 l), so that the lanes' 8-byte reads of their own rows spread over
// the banks.  No register is in flight (the first version asked for the lane's own 16 + 16 table words with 8-byte gathers - 32 different lines per
// instruction - into accumulator registers: 68 us of the kernel per call, profiles/r05_rope_cost.log).  The requests are invisible to the compiler:
// dkv3_rope_landed() waits for them by count.
__device__ __forceinline__ void dkv3_rope_request_lds(char* cos_dst, char* sin_dst, const bf16_t* cos, const bf16_t* sin, int pos0, int max_pos, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), ch = (lane & 7) ^ (row & 7);
        const int64_t at = (int64_t)min(pos0 + row, max_pos - 1) * 64 + ch * 8;
        const bf16_t* sc = cos + at;
        const bf16_t* ss = sin + at;
        const unsigned dc = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(cos_dst + 1024 * i);
        const unsigned ds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(sin_dst + 1024 * i);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(sc), "v"(ss), "s"(dc), "s"(ds) : "memory");
    }
}
template <int N>      // all but the wave's N youngest vector-memory operations are done
__device__ __forceinline__ void dkv3_rope_landed() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// One key block in plain HIP (HALVA_DKV3_ASM=0: the readable twin of the generated loop, with the same ring protocol; every item starts cold)
template <bool CAUSAL>
__device__ __forceinline__ void sdpa_bwd_dkv3_block_hip(const SdpaParams& p, char* smem, int s, int hd, int kb, int wave, int lane, int start, int len,
                                                    const Branch br) {
    constexpr int D = 128, KS = 8, DT = 4, BQ = 64;
    const int h = lane >> 5;
    const int64_t seq_row0 = (int64_t)s * p.T;
    const int gk = kb * 128 + 32 * wave + (lane & 31);
    const int kl = gk - start;
    const bool k_in_T = gk < p.T;
    const bool k_valid = k_in_T && kl >= 0 && kl < len;
    const int kblk_min = kb * 128 - start;
    const int q_begin = CAUSAL ? max(0, kblk_min) / BQ * BQ : 0;
    const bool block_has_keys = (kblk_min < len) && (kblk_min + 128 > 0);
    const int q_stop = (kblk_min >= br.a && kblk_min + 127 < br.b) ? min(len, br.b) : len;
    const int ntiles = (block_has_keys && q_stop > q_begin) ? (q_stop - q_begin + BQ - 1) / BQ : 0;
    bf16_t* dk_row = p.dk + (seq_row0 + gk) * p.ld_qkv + hd * D;
    bf16_t* dv_row = p.dv + (seq_row0 + gk) * p.ld_qkv + hd * D;
    if (ntiles == 0) {
        if (k_in_T) {
            store_rows_zero<D>(dk_row, lane);
            store_rows_zero<D>(dv_row, lane);
        }
        return;
    }
    const bool key_hidden = kl >= br.a && kl < br.b;
    const int wk_min = kblk_min + 32 * wave;
    const bool block_all_keys_valid = kblk_min >= 0 && kblk_min + 128 <= len;      // workgroup-uniform: no padded key in the block
    Dkv3State st;
    {
        const bf16_t* krow = p.k + (seq_row0 + gk) * p.ld_qkv + hd * D;
        const bf16_t* vrow = p.v + (seq_row0 + gk) * p.ld_qkv + hd * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            st.kq[ks] = k_valid ? *reinterpret_cast<const u32x4*>(krow + 16 * ks + 8 * h) : u32x4{0u, 0u, 0u, 0u};
            st.vq[ks] = k_valid ? *reinterpret_cast<const u32x4*>(vrow + 16 * ks + 8 * h) : u32x4{0u, 0u, 0u, 0u};
        }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st.accV[dt][r] = 0.f, st.accK[dt][r] = 0.f;
    const bf16_t* qp = p.q + hd * D;
    const bf16_t* dop = p.d_o + hd * D;
    const int64_t qrow0 = seq_row0 + start;
    const float* stat_g = p.lse2 + ((int64_t)s * p.H + hd) * p.stat_nt * 128;      // [step][lse2 x 64 | -delta x 64], sequence coordinates
    char* q_lds = smem;
    char* do_lds = smem + DKV3_DO;
    float* lse_lds = reinterpret_cast<float*>(smem + DKV3_LSE);
    float* nd_lds = reinterpret_cast<float*>(smem + DKV3_ND);
    const float sc = p.scale * kLog2e;
    char* ds_block = p.ds_ws + ((((int64_t)s * p.H + hd) * p.ds_nkb + kb) * p.ds_nt + q_begin / BQ) * 16384 + wave * 4096;

    auto request_tile = [&](int i) {      // tile i of this block into ring slot i & 3 (rows past the end repeat the last row; callers mask them)
        const int slot = i & 3;
        stage_tile_dma<D, 4>(q_lds + slot * DKV3_TILE, qp, p.ld_qkv, qrow0, q_begin + i * BQ, len, wave, lane);
        stage_tile_dma<D, 4>(do_lds + slot * DKV3_TILE, dop, p.ld_do, qrow0, q_begin + i * BQ, len, wave, lane);
    };
    auto load_stats = [&](int i, float& a, float& b) {
        const int ql = min(q_begin + i * BQ + lane, len - 1);
        a = stat_g[(ql >> 6) * 128 + (ql & 63)];
        b = stat_g[(ql >> 6) * 128 + 64 + (ql & 63)];
    };
    auto store_stats = [&](int i, float a, float b) {
        lse_lds[(i & 3) * 64 + lane] = a;
        nd_lds[(i & 3) * 64 + lane] = b;
    };
    // ---- prologue: the previous block's readers are done; request tiles 0..2, prepare everything else, then wait
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
        float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};      // every request first, ONE wait: a statistic stored right behind its load
#pragma unroll                                                      // would wait for the tile requests in front of it as well (vmcnt is in order)
        for (int i = 0; i < 3; ++i)
            if (i < ntiles && wave == 0) load_stats(i, sa[i], sb[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < ntiles) request_tile(i);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < ntiles && wave == 0) store_stats(i, sa[i], sb[i]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int t = 0;
#pragma unroll 1
    while (t < ntiles) {
        const int qt0 = q_begin + t * BQ;
        const bool q_in_b = qt0 >= br.b;                     // br.b and qt0 are multiples of 64: uniform over the step
        // ---- plain HIP step (debug build of the kernel, ASM = false): request tile t+3, compute with the masks, end with "tile t+1 has landed"
        float sa = 0.f, sb = 0.f;
        if (t + 3 < ntiles) {      // (behind step t-1's barrier the slot of tile t-1 is free: tile t+3; tiles 0..2 came with the prologue)
            request_tile(t + 3);
            if (wave == 0) load_stats(t + 3, sa, sb);
        }
        const bool hidden = q_in_b && wk_min >= br.a && wk_min + 31 < br.b;      // this wave's strip wholly hidden from the step's rows
        if (!hidden) {
            const int slot = t & 3;
            const bool masked = !((qt0 + BQ <= len) && (!CAUSAL || qt0 >= wk_min + 31) && !__any(!k_valid) &&
                                  !(q_in_b && wk_min < br.b && wk_min + 31 >= br.a));      // wave-uniform
            dkv3_hip_step<CAUSAL>(st, q_lds + slot * DKV3_TILE, do_lds + slot * DKV3_TILE, lse_lds + slot * 64, nd_lds + slot * 64,
                                  ds_block + (int64_t)t * 16384, sc, qt0, len, kl, !k_valid || (q_in_b && key_hidden), masked, lane);
        }
        if (t + 3 < ntiles && wave == 0) store_stats(t + 3, sa, sb);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        ++t;
    }
    if (k_in_T) {
        store_rows_T<D>(dv_row, st.accV, k_valid ? 1.f : 0.f, true, lane);
        store_rows_T<D>(dk_row, st.accK, k_valid ? p.scale : 0.f, true, lane);
    }
}

// The item loop of the generated-asm build.  Per item: [the asm block: every step of the key block; on its way out it requests the NEXT
// item's first tiles] -> the next item's K / V fragments are asked for -> this item's dK / dV rows are converted and stored while those
// loads and requests fly -> next item.  (Before this pipelining an item paid ~16 000 cycles around its steps - K / V fragments 2 200, the
// first tiles 4 000, address arithmetic 2 500, the stores and their acknowledgements 4 700, the scheduler 2 500 - with nothing else resident
// on the CU to fill them: a fifth of the kernel.)
// Round 6: what is uniform about an item arrives as its RECORD (sdpa_dkv3_items.h: written by the delta pass of the same call), held by every
// wave in ONE vector register (lane l = dword l).  The scheduler is: thread 0 turns the home queue's answer into a record index (one compare and
// one addition while the home queue lasts); wave 0's lanes 0..15 fetch that record INSIDE the block's first call (gen_dkv3_loop.py: beside the
// counter's draw, back under the wait for the K / V fragments) and leave it in the LDS mail box; behind the round's barrier every wave reads its
// dword of it.  Measured before (profiles/r06_dkv3_anatomy.log, 16 x 2048): 6 616 cycles of compiler code in front of every block (scheduler
// 2 180, run lengths / addresses 1 935, K / V fetch 2 441) + 643 to collect the next geometry behind it.
#define DKV3_F(rec, field) __builtin_amdgcn_readlane((rec), DKV3_REC_##field)
template <bool CAUSAL>
__device__ __forceinline__ void sdpa_bwd_dkv3_items(const SdpaParams& p, char* smem, int wave, int lane_in) {
    constexpr int D = 128, BQ = 64;
    // (an LDS-typed pointer: through a generic one the accesses are FLAT instructions, which wait on the vector-memory counter as well - i.e.
    // for the acknowledgements of the rows just stored and the next item's tiles, all of which this loop is arranged not to wait for)
    typedef __attribute__((address_space(3))) volatile int LdsInt;
    LdsInt* mail = (LdsInt*)(__attribute__((address_space(3))) char*)(smem + DKV3_SCHED);      // [2][64]: two item records
    const int home = blockIdx.x & 7;
    const int nkb = p.nblk, G = p.npairs, total = G * nkb;      // record `total`: the all-zero "queues are empty" record
    const int home_len = dkv3_queue_len(home, G, nkb), home_base = dkv3_queue_base(home, G, nkb);
    auto resolve = [&](int j) {      // thread 0: what the home queue's counter returned -> record index
        if (j < home_len) return home_base + j;
#pragma unroll 1
        for (int k = 1; k < 8; ++k) {      // the home queue is empty: the others, nearest first
            const int x = (home + k) & 7, n = dkv3_queue_len(x, G, nkb);
            if (n == 0) continue;
            const int jj = atomicAdd(p.sched + 32 * x, 1);
            if (jj < n) return dkv3_queue_base(x, G, nkb) + jj;
        }
        return total;
    };
    // a record into a mail-box slot by compiler code (start-up, and the rare item without steps: nothing is in flight that a vmcnt(0) would wait for
    // in vain); wave 0 only
    auto fetch_record = [&](int idx, int slot) {
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (lane_in < 16) {
            typedef int i4 __attribute__((ext_vector_type(4)));
            const i4 v = *reinterpret_cast<const i4*>(p.items + (int64_t)idx * DKV3_REC_DWORDS + 4 * lane_in);
            *reinterpret_cast<__attribute__((address_space(3))) i4*>((__attribute__((address_space(3))) char*)(smem + DKV3_SCHED) + 256 * slot + 16 * lane_in) = v;
        }
    };
    int drawn = 0;      // thread 0: the home counter's answer for the item after next
    if (wave == 0) {
        int i0 = 0, i1 = 0;
        if (threadIdx.x == 0) {
            i0 = resolve(atomicAdd(p.sched + 32 * home, 1));
            i1 = resolve(atomicAdd(p.sched + 32 * home, 1));
            drawn = atomicAdd(p.sched + 32 * home, 1);
        }
        fetch_record(i0, 0);
        fetch_record(i1, 1);
    }
    __syncthreads();
    int rec = mail[lane_in], nrec = mail[64 + lane_in];
    // every wave has READ both slots before wave 0 overwrites slot 0 in round 0 (later rounds: the end-of-round barrier separates a
    // slot's last read from its next write; here nothing did - a delayed wave could have taken the third item's record for its first)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const unsigned q_piece = dkv3_uni((unsigned)(16 * p.ld_qkv * 2)), do_piece = dkv3_uni((unsigned)(16 * p.ld_do * 2));
    const unsigned wave_u = dkv3_uni((unsigned)wave);
    const float sc = p.scale * kLog2e;
    const unsigned ld2 = dkv3_uni((unsigned)(p.ld_qkv * 2));      // bytes between two rows of q / k / v (the launcher keeps it below 2^31)
    const int64_t v_minus_k = (const char*)p.v - (const char*)p.k, dv_minus_dk = (const char*)p.dv - (const char*)p.dk;

    bool prefetched = false;      // cur's first tiles were requested by the previous item's asm block (its last three steps) ...
    int ring_base = 0;            // ... into the ring slots ring_base, ring_base + 1, ...
#ifdef HALVA_STAMP
#define DKV3_NOW(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
    unsigned long long acc_pre_ = 0, acc_asm_ = 0, acc_post_ = 0, n_items_ = 0, t0_, t1_, t2_, t3_, tk_ = 0, ts_ = 0, acc_k_ = 0, acc_s_ = 0, tp_ = 0, tst_ = 0, tb_ = 0, acc_p_ = 0, acc_st_ = 0, acc_b_ = 0;
#endif
#pragma unroll 1
    for (int round = 0; DKV3_F(rec, VALID); ++round) {
#ifdef HALVA_STAMP
        DKV3_NOW(t0_);
        t1_ = t2_ = t0_;
#endif
        // the lane parts of the generated loop's LDS / global addresses are the same for every item, and are formed per item all the same:
        // kept across the asm block (which leaves the compiler 97 vector registers) they were spilled to scratch, and a scratch reload waits
        // for every tile request in flight (vmcnt counts in order)
        int lane = lane_in;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5;
        const int ntiles = DKV3_F(rec, NTILES), kblk_min = DKV3_F(rec, KBLK_MIN), len = DKV3_F(rec, LEN);
        const int kl = kblk_min + 32 * wave + (lane & 31);      // this lane's key in sequence coordinates
        const bool k_valid = kl >= 0 && kl < len && DKV3_F(rec, KB) * 128 + 32 * wave + (lane & 31) < p.T;
        bool requested_next = false;
        if (ntiles > 0) {
            // this item's K / V fragments, asked for FIRST and by hand, straight into the registers the block reads them from - a128-a191, literal
            // registers here and in the block, not values the compiler knows about (a key outside the sequence reads the nearest one inside: its lane
            // is masked).  Nobody waits for them before the block's own s_waitcnt vmcnt(0).  Loaded by the compiler they cost a vmcnt(0) wherever it
            // chose to move them (it cannot count past an asm block).  What they cost: a fragment load is 64 separate 16-byte pieces (a lane = a key =
            // a row of its own, 24 KiB apart) and holds the lone wave's issue for ~170 cycles - 2 750 per item wherever the sixteen stand - and
            // what the code between here and the block does not cover of their ~4 000 cycles of latency is waited for inside the block's first call.
            // Round 6 tried to get rid of both (experiments/dkv3_item_boundary: the loads in front of / between / behind the PREVIOUS item's store
            // conversions; a second register set a192-a255 filled by the previous block's first call): the issue time cannot be hidden by a wave
            // that is alone on its SIMD, and none of the five variants beat this one.
#ifdef DKV3_DIAG_NO_KV      // (timing experiment, results wrong: what would FREE K / V fragments buy? - the ceiling of any other way to deliver them)
            if (round == 0)
#endif
            {
                const unsigned long long k0 = ((unsigned long long)(unsigned)DKV3_F(rec, K_HI) << 32) | (unsigned)DKV3_F(rec, K_LO);
                const char* k_ptr = (const char*)(size_t)k0 + (unsigned long long)(unsigned)min(max(kl, 0), len - 1) * ld2 + 16 * h;
                const char* v_ptr = k_ptr + v_minus_k;
                asm volatile(
                    "global_load_dwordx4 a[128:131], %0, off\n\tglobal_load_dwordx4 a[132:135], %0, off offset:32\n\tglobal_load_dwordx4 a[136:139], %0, off offset:64\n\t"
                    "global_load_dwordx4 a[140:143], %0, off offset:96\n\tglobal_load_dwordx4 a[144:147], %0, off offset:128\n\tglobal_load_dwordx4 a[148:151], %0, off offset:160\n\t"
                    "global_load_dwordx4 a[152:155], %0, off offset:192\n\tglobal_load_dwordx4 a[156:159], %0, off offset:224\n\t"
                    "global_load_dwordx4 a[160:163], %1, off\n\tglobal_load_dwordx4 a[164:167], %1, off offset:32\n\tglobal_load_dwordx4 a[168:171], %1, off offset:64\n\t"
                    "global_load_dwordx4 a[172:175], %1, off offset:96\n\tglobal_load_dwordx4 a[176:179], %1, off offset:128\n\tglobal_load_dwordx4 a[180:183], %1, off offset:160\n\t"
                    "global_load_dwordx4 a[184:187], %1, off offset:192\n\tglobal_load_dwordx4 a[188:191], %1, off offset:224"
                    :
                    : "v"(k_ptr), "v"(v_ptr)
                    : "memory");
            }
#ifdef HALVA_STAMP
            DKV3_NOW(tk_);
#endif
            const unsigned rowrel = 2048 * ((lane & 31) >> 3) + 64 * (lane & 7) + 16 * (h ^ (((lane & 31) >> 2) & 3));
            const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, h2 = g16 >> 1;
            const unsigned colrel = 64 * (4 * h2 + q4) + 16 * ((2 * (g16 & 1) + (pp >> 1)) ^ h2) + 8 * (pp & 1);
            const unsigned statrel = 16 * h;
            const unsigned voff_q = dkv3_piece_voff(p.ld_qkv, wave, lane, 0, 64), voff_do = dkv3_piece_voff(p.ld_do, wave, lane, 0, 64);
            // lanes 32..63 of a statistics request lie outside the descriptor on purpose (they fill the unused half of the slot)
            const unsigned stat_voff = lane < 32 ? 16u * lane : 0x7ffffff0u;
            const int q_begin = DKV3_F(rec, Q_BEGIN);
            const Branch br{DKV3_F(rec, BR_A), DKV3_F(rec, BR_B)};
            const bool key_hidden = kl >= br.a && kl < br.b;
            const unsigned long long ds0 = (((unsigned long long)(unsigned)DKV3_F(rec, DS_HI) << 32) | (unsigned)DKV3_F(rec, DS_LO)) + 4096u * wave_u;
            const int npro_next = DKV3_F(nrec, PREFETCHABLE);      // (0 for the all-zero record)
            // (fixed accumulator registers, the same in every asm statement that touches them: no copies; the block zeroes them on its first call)
            f32x16 accV[4], accK[4];      // dV^T, dK^T (local to the item: carried from round to round they travelled through vector registers)
#ifdef HALVA_STAMP
            DKV3_NOW(ts_);
#endif
            // thread 0: the item after next (the counter's answer came out of the previous item's asm block) - its record's address for the
            // block's first call, whose wave-0 lanes 0..15 fetch it (16 bytes each) into the mail-box slot `cur` was read from
            const int nn_idx = threadIdx.x == 0 ? resolve(drawn) : 0;
            const int* rec_ptr = p.items + (int64_t)__builtin_amdgcn_readfirstlane(nn_idx) * DKV3_REC_DWORDS + 4 * (lane & 15);
#ifdef HALVA_STAMP
            DKV3_NOW(tp_);
#endif
            int* home_counter = p.sched + 32 * home;
            int drawn_out;      // the next answer of the home queue's counter: asked for by the block's first call (see gen_dkv3_loop.py)
            int t = 0;
#define DKV3_FIRST 1
#define DKV3_ACC_MOD "="
#include "sdpa_dkv3_call.h"
#undef DKV3_ACC_MOD
#undef DKV3_FIRST
            if (threadIdx.x == 0) drawn = drawn_out;
#pragma unroll 1
            while (t < ntiles) {
                int drawn_out;      // (not a first call: nothing is drawn)
#define DKV3_FIRST 0
#define DKV3_ACC_MOD "+"
#include "sdpa_dkv3_call.h"
#undef DKV3_ACC_MOD
#undef DKV3_FIRST
            }
            // (straight behind the loop, in the same block: carried to a common tail the accumulators travelled through vector registers)
            if (DKV3_ROWS_VIA_LDS) {      // (rows of the wave in the tensor: uniform)
                const int rows_ok = wave_u == 0 ? DKV3_F(rec, ROWS_OK0) : wave_u == 1 ? DKV3_F(rec, ROWS_OK1) : wave_u == 2 ? DKV3_F(rec, ROWS_OK2) : DKV3_F(rec, ROWS_OK3);
                const unsigned long long dk0 = ((unsigned long long)(unsigned)DKV3_F(rec, DK_HI) << 32) | (unsigned)DKV3_F(rec, DK_LO);
                // (through a GLOBAL-typed pointer: an address built from integers is a generic one to the compiler, and generic stores are FLAT
                // instructions - they count on the LDS counter too, every LDS read behind one waited for vmcnt(0): 7 000 cycles per item, measured)
                bf16_t* w0 = (bf16_t*)(__attribute__((address_space(1))) bf16_t*)(size_t)(dk0 + (unsigned long long)(32u * wave_u) * ld2);
                // the ring slot the item's last tile has left: free until the next item's first step (its prefetched tiles sit in the other three)
                const int free_slot = (ring_base + ntiles - 1) & 3;
                char* rope_c = smem + free_slot * DKV3_TILE + wave * 4096;
                char* rope_s = smem + DKV3_DO + free_slot * DKV3_TILE + wave * 4096;
                if (p.rope_cos) {      // (uniform) the table rows of this wave's keys, asked for now, needed behind the dV rows
                    const int pos0 = wave_u == 0 ? DKV3_F(rec, ROPE_POS0) : wave_u == 1 ? DKV3_F(rec, ROPE_POS1) : wave_u == 2 ? DKV3_F(rec, ROPE_POS2) : DKV3_F(rec, ROPE_POS3);
                    dkv3_rope_request_lds(rope_c, rope_s, p.rope_cos, p.rope_sin, pos0, p.rope_max_pos, lane);
                }
                // (dV of a key outside the sequence: every step of such a key is a masked one and its P is exactly 0, so the accumulator is
                // 0 already - the multiplication by 1 / 0 of rounds 2-5 changed nothing and cost 64 instructions per item)
                dkv3_store_rows_lds<false, false>(smem, wave, (bf16_t*)((char*)w0 + dv_minus_dk), p.ld_qkv, accV, 1.f, rows_ok, lane, nullptr, nullptr);
                if (p.rope_cos) {
                    // the dV rows' stores are younger than the table requests; how many the compiler issues is its own business (it may merge or
                    // branch over them): wait for everything (profiles/r05_rope_cost.log: no measurable difference to a counted wait)
                    dkv3_rope_landed<0>();
                    dkv3_store_rows_lds<true, true>(smem, wave, w0, p.ld_qkv, accK, p.scale, rows_ok, lane, rope_c, rope_s);
                } else {
                    dkv3_store_rows_lds<false, true>(smem, wave, w0, p.ld_qkv, accK, p.scale, rows_ok, lane, nullptr, nullptr);
                }
            } else {
                const int gk = DKV3_F(rec, KB) * 128 + 32 * wave + (lane & 31);
                bf16_t* dk_row = p.dk + ((int64_t)DKV3_F(rec, S) * p.T + gk) * p.ld_qkv + DKV3_F(rec, HD) * D;
                if (gk < p.T) {
                    store_rows_T<D>((bf16_t*)((char*)dk_row + dv_minus_dk), accV, k_valid ? 1.f : 0.f, true, lane);
                    store_rows_T<D>(dk_row, accK, k_valid ? p.scale : 0.f, true, lane);
                }
            }
            ring_base = (ring_base + ntiles) & 3;      // (the next item's prefetched tiles continue the slot rotation)
        } else {
            // an item without steps (a key block past its sequence, or wholly hidden): nothing is in flight, the scheduler's part in compiler code
            if (wave == 0) {
                int nn_idx = 0;
                if (threadIdx.x == 0) {
                    nn_idx = resolve(drawn);
                    drawn = atomicAdd(p.sched + 32 * home, 1);
                }
                fetch_record(nn_idx, round & 1);
            }
            const int gk = DKV3_F(rec, KB) * 128 + 32 * wave + (lane & 31);
            bf16_t* dk_row = p.dk + ((int64_t)DKV3_F(rec, S) * p.T + gk) * p.ld_qkv + DKV3_F(rec, HD) * D;
            if (gk < p.T) {
                store_rows_zero<D>(dk_row, lane);
                store_rows_zero<D>((bf16_t*)((char*)dk_row + dv_minus_dk), lane);
            }
        }
#ifdef HALVA_STAMP
        DKV3_NOW(tst_);
#endif
        // the mail box is LDS: wait for the LDS write only.  (__syncthreads() also waits for every vector-memory operation in flight - the
        // acknowledgements of the rows just stored, the next item's tiles - which is exactly what this loop is arranged not to wait for.)
        // Also: every wave is done with this item's LDS before a cold first call of the next one requests into it.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef HALVA_STAMP
        DKV3_NOW(tb_);
#endif
        rec = nrec;
        prefetched = requested_next;
        nrec = mail[64 * (round & 1) + lane_in];
#ifdef HALVA_STAMP
        DKV3_NOW(t3_);
        acc_pre_ += t1_ - t0_, acc_asm_ += t2_ - t1_, acc_post_ += t3_ - t2_, ++n_items_;
        if (tk_ > t0_) acc_k_ += tk_ - t0_, acc_s_ += ts_ - tk_, acc_p_ += tp_ - ts_, acc_st_ += tst_ - t2_, acc_b_ += tb_ - tst_;
#endif
    }
#ifdef HALVA_STAMP
    if (p.dbg && threadIdx.x == 0 && blockIdx.x < 120)
        p.dbg[7300 + blockIdx.x * 3] = acc_p_, p.dbg[7300 + blockIdx.x * 3 + 1] = acc_st_, p.dbg[7300 + blockIdx.x * 3 + 2] = acc_b_,
        p.dbg[7000 + blockIdx.x * 2] = acc_k_, p.dbg[7000 + blockIdx.x * 2 + 1] = acc_s_, p.dbg[6144 + blockIdx.x * 4] = acc_pre_, p.dbg[6144 + blockIdx.x * 4 + 1] = acc_asm_, p.dbg[6144 + blockIdx.x * 4 + 2] = acc_post_, p.dbg[6144 + blockIdx.x * 4 + 3] = n_items_;
#endif
}

// Persistent workgroups, one per CU, that draw (sequence, head, key block) items from eight queues - one per XCD, so that the workgroups of an
// XCD work on the same few (sequence, head) pairs and find their Q / dO tiles in that XCD's L2 - and help the other queues out when their own
// is empty.  Measured on the static 2048-workgroup launch this replaces (tools/stamp_rounds.py): two of the eight XCDs of a box ran every
// workgroup 6-8 % slower than the rest (12 % more cycles per item), and since the hardware deals workgroups to XCDs round-robin the launch
// ended when THEY were done.  Queue x holds the pairs x, x + 8, ...; the order inside it: SdpaParams::sched_order.
// p.sched: eight counters 128 B apart, zeroed by the delta pass of the same call.
template <int D, bool CAUSAL, bool ASM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sdpa_bwd_dkv3_kernel(const SdpaParams p) {
    static_assert(D == 128, "sdpa_bwd_dkv3 is the head_dim-128 instantiation");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WG_CLOCK_BEGIN();
    if (ASM) {
        sdpa_bwd_dkv3_items<CAUSAL>(p, smem, wave, lane);
    } else {      // the plain build: queue 0..7 in turn, group-major, one atomic per item, no look-ahead
        volatile int* mail = reinterpret_cast<volatile int*>(smem + DKV3_SCHED);
        const int nkb = p.nblk, total = p.npairs * nkb;
#pragma unroll 1
        for (int round = 0;; ++round) {
            if (threadIdx.x == 0) mail[round & 1] = atomicAdd(p.sched, 1);
            __syncthreads();
            const int item = mail[round & 1];
            if (item >= total) break;
            const int g = item / nkb, kb = item - g * nkb, s = g / p.H, hd = g - s * p.H;
            const int start = p.seq_start ? p.seq_start[s] : 0;
            const int len = p.seq_len ? p.seq_len[s] : p.T;
            sdpa_bwd_dkv3_block_hip<CAUSAL>(p, smem, s, hd, kb, wave, lane, start, len, load_branch(p, s));
        }
    }
    WG_CLOCK_END(p.dbg, 3);
}
