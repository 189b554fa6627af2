// sdpa_bwd_dkv3: dK / dV (+ dS store) with ONE wave per SIMD - included by sdpa.hip inside its anonymous namespace (head_dim 128 only).
//
// Replaces sdpa_bwd_dkv2 on the dS-workspace path of halva_sdpa_branch_bwd_ws (the backward of flash_attn_varlen_qkvpacked_func,
// reference llava/train/llama_flash_attn_monkey_patch.py:85-91).  Why another structure: the two-role kernel puts two 256-register waves
// on every SIMD and they meet - at the matrix pipe, in their vector stretches and at the step's barrier - in lock step: ~3 850 cycles
// per 64-row step for 2 048 cycles of matrix work, whatever was done to either wave's own stream.  Here a workgroup is
// 4 waves = 128 keys, a wave owns 32 keys with BOTH accumulators (dK^T, dV^T: 128 accumulator registers), the K and V fragments (64) and
// runs all four products of a step itself, alone on its SIMD with the whole 512-register file.  hipcc cannot schedule that (every MFMA
// destination lands in the accumulator file and each score is copied out before the vector unit may touch it: 3 650 cycles per step), so
// the INTERIOR steps of a key block - whole tile, every key of the block visible to every row, no padding, no branch edge: all but the
// two diagonal steps and the sequence tail - run in one generated inline-asm loop (sdpa_dkv3_loop.inc <- gen_dkv3_loop.py; 2 419 cycles
// per step in isolation, bit-identical to the plain code: experiments/dkv3).  The boundary steps run the plain HIP step below, which
// carries the masks.  Both keep the same protocol on a ring of FOUR Q / dO tile slots (+ their lse2 / -delta rows):
//     at the start of step t tile t has landed and is visible to every wave; tiles t+1, t+2 have been requested;
//     at the head of step t tile t+3 is requested into the slot tile t-1 left at the last barrier; the step ends with "tile t+1 has landed" + s_barrier.
// Statistics: the delta pass writes -delta and lse2 = lse * log2(e) (SdpaParams::lse2, the tail of the workspace), so that both are plain
// rows an LDS-DMA dword request can fetch and -delta is directly the initial accumulator of the dP chain.
// dS leaves in the same image sdpa_bwd_dkv2 writes (strip = wave), so sdpa_bwd_dq2 is unchanged.

constexpr int DKV3_TILE = 64 * 128 * 2;                      // one Q or dO tile
constexpr int DKV3_DO = 4 * DKV3_TILE;                       // dO ring behind the Q ring
constexpr int DKV3_LSE = 8 * DKV3_TILE;                      // [4][64] lse2, then [4][64] -delta
constexpr int DKV3_ND = DKV3_LSE + 4 * 64 * 4;
constexpr int DKV3_DUMMY = DKV3_ND + 4 * 64 * 4;              // 1 KiB: where the requests of the last steps of a block (no tile left) land
constexpr int DKV3_LDS = DKV3_DUMMY + 1024;

#define DKV3_PIN_A(x) asm volatile("" : "+a"(x))

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

template <bool CAUSAL, bool ASM>
__device__ __forceinline__ void sdpa_bwd_dkv3_block(const SdpaParams& p, char* smem, int s, int hd, int kb, int wave, int lane, int start, int len,
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
    if (ASM) {      // the generated block zeroes them itself on its first call (no copies into its operand registers)
        // (fixed accumulator registers, the same in every asm statement that touches them: no copies between the compiler's choice and the block's)
        asm volatile("" : "={a[0:15]}"(st.accV[0]), "={a[16:31]}"(st.accV[1]), "={a[32:47]}"(st.accV[2]), "={a[48:63]}"(st.accV[3]), "={a[64:79]}"(st.accK[0]), "={a[80:95]}"(st.accK[1]), "={a[96:111]}"(st.accK[2]),
                     "={a[112:127]}"(st.accK[3]));
    } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.accV[dt][r] = 0.f, st.accK[dt][r] = 0.f;
    }
    const bf16_t* qp = p.q + hd * D;
    const bf16_t* dop = p.d_o + hd * D;
    const int64_t qrow0 = seq_row0 + start;
    const float* lse2_g = p.lse2 + ((int64_t)s * p.H + hd) * p.T + start;
    const float* nd_g = p.delta + ((int64_t)s * p.H + hd) * p.T + start;
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
        a = lse2_g[ql];
        b = nd_g[ql];
    };
    auto store_stats = [&](int i, float a, float b) {
        lse_lds[(i & 3) * 64 + lane] = a;
        nd_lds[(i & 3) * 64 + lane] = b;
    };
#ifdef HALVA_STAMP
#define DKV3_NOW(x) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
    unsigned long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // 0 entry, 1 loop start, 2 asm cycles, 3 plain steps, 5 masked steps,
    DKV3_NOW(stamp[0]);                                                       // 6 before stores, 7 done, 8 first barrier passed, 9 requests issued, 10 tiles landed
#endif
    // a step the UNMASKED phase may run: whole tile, every key of the block visible to every row, no padded key, no branch edge
    // (the HIP build classifies per wave inside its step; the ASM build derives the three run lengths from the same conditions below)
    // lane parts of the LDS / global addresses of the generated loop
    const unsigned rowrel = 2048 * ((lane & 31) >> 3) + 64 * (lane & 7) + 16 * (h ^ (((lane & 31) >> 2) & 3));
    const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, h2 = g16 >> 1;
    const unsigned colrel = 64 * (4 * h2 + q4) + 16 * ((2 * (g16 & 1) + (pp >> 1)) ^ h2) + 8 * (pp & 1);
    const unsigned statrel = 16 * h;
    const unsigned voff_q = dkv3_piece_voff(p.ld_qkv, wave, lane, 0, 64), voff_do = dkv3_piece_voff(p.ld_do, wave, lane, 0, 64);
    // a partial last tile: its rows are clamped to the sequence, piece by piece
    const int last_rows = len - (q_begin + (ntiles - 1) * BQ);              // >= 64: whole (or the block's rows stop at br.b); 1..63: the sequence ends inside the tile
    const bool last_partial = last_rows < BQ;
    const int lr = last_partial ? last_rows : BQ;
    unsigned alt_q[4], alt_do[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        alt_q[i] = dkv3_piece_voff(p.ld_qkv, wave, lane, i, lr);
        alt_do[i] = dkv3_piece_voff(p.ld_do, wave, lane, i, lr);
    }
    const unsigned alt_stat = 4 * min(lane, lr - 1);

    // ---- prologue: the previous block's readers are done; request tiles 0..2, prepare everything else, then wait
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef HALVA_STAMP
    DKV3_NOW(stamp[8]);
#endif
    if (!ASM) {
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
    } else {      // (the generated block requests tiles 0..2 itself on its first call, and waits for them)
        asm volatile("" : "+{a[128:131]}"(st.kq[0]), "+{a[132:135]}"(st.kq[1]), "+{a[136:139]}"(st.kq[2]), "+{a[140:143]}"(st.kq[3]), "+{a[144:147]}"(st.kq[4]), "+{a[148:151]}"(st.kq[5]), "+{a[152:155]}"(st.kq[6]), "+{a[156:159]}"(st.kq[7]),
                     "+{a[160:163]}"(st.vq[0]), "+{a[164:167]}"(st.vq[1]), "+{a[168:171]}"(st.vq[2]), "+{a[172:175]}"(st.vq[3]), "+{a[176:179]}"(st.vq[4]), "+{a[180:183]}"(st.vq[5]), "+{a[184:187]}"(st.vq[6]), "+{a[188:191]}"(st.vq[7]));
    }
#ifdef HALVA_STAMP
    DKV3_NOW(stamp[9]);
    DKV3_NOW(stamp[10]);
    DKV3_NOW(stamp[1]);
#endif
    int t = 0;
#pragma unroll 1
    while (t < ntiles) {
        const int qt0 = q_begin + t * BQ;
        const bool q_in_b = qt0 >= br.b;                     // br.b and qt0 are multiples of 64: uniform over the step
        if (ASM) {
            // ONE asm block for the steps up to the branch point (or the end): masked steps (the diagonal), interior steps, masked steps (the
            // tail); a layout with more alternations than that takes another round of this loop
            const int t_side = q_in_b ? ntiles : min(ntiles, (int)(((int64_t)br.b - q_begin + BQ - 1) / BQ));      // first step at or behind br.b
            int t1 = t, n0 = 0, n1 = 0, n2 = 0;
#ifdef HALVA_DKV3_ALL_MASKED      // diagnostic: every step through the masked phase (same results: an interior step's masks pass everything)
            n0 = t_side - t, t1 = t_side;
#else
            {      // interior(t') on this side of br.b = side_ok && t_diag <= t' < t_full: three runs, no scan
                const bool side_ok = block_all_keys_valid && !(q_in_b && kblk_min < br.b && kblk_min + 127 >= br.a);
                const int t_diag = CAUSAL ? max(0, (kblk_min + 127 - q_begin + BQ - 1) / BQ) : 0;      // first t' with qt0 >= kblk_min + 127
                const int t_full = max(0, (len - q_begin) / BQ);                                          // first t' with qt0 + 64 > len
                const int lo = min(t_side, max(t, t_diag)), hi = min(t_side, max(lo, t_full));
                n0 = side_ok ? lo - t : t_side - t;
                n1 = side_ok ? hi - lo : 0;
                n2 = t_side - t - n0 - n1;
                t1 = t_side;
            }
#endif
            const int n = t1 - t;
            const int ndma = min(n, max(0, ntiles - 3 - t));      // steps t' of the call with a tile t'+3 to request
            const bool part = last_partial && ndma > 0 && (t + ndma - 1 + 3 == ntiles - 1);
            const bool lane_off = !k_valid || (q_in_b && key_hidden);
            const unsigned lo0 = (unsigned)(kl - qt0 - 4 * h), range = lane_off ? 0u : (unsigned)(len - kl);
            const unsigned long long ds_ptr = dkv3_uni64(ds_block + (int64_t)t * 16384);
            // the next tile to request: tile t+3 - or tile 0 on the block's first call, which requests tiles 0..2 up front
            const int tq = t == 0 ? 0 : t + 3;
            const unsigned long long q_ptr = dkv3_uni64(qp + (qrow0 + q_begin + (int64_t)tq * BQ) * p.ld_qkv);
            const unsigned long long do_ptr = dkv3_uni64(dop + (qrow0 + q_begin + (int64_t)tq * BQ) * p.ld_do);
            const unsigned long long lse_ptr = dkv3_uni64(lse2_g + q_begin + tq * BQ);
            const unsigned long long nd_ptr = dkv3_uni64(nd_g + q_begin + tq * BQ);
            const unsigned first_u = dkv3_uni(t == 0 ? 1u : 0u), npro_u = dkv3_uni((unsigned)min(3, ntiles));
            const unsigned proalt_u = dkv3_uni((last_partial && ntiles <= 3) ? (unsigned)(ntiles - 1) : 7u);
            const unsigned q_piece = dkv3_uni((unsigned)(16 * p.ld_qkv * 2)), do_piece = dkv3_uni((unsigned)(16 * p.ld_do * 2));
            const unsigned n0_u = dkv3_uni((unsigned)n0), n1_u = dkv3_uni((unsigned)n1), n2_u = dkv3_uni((unsigned)n2);
            const unsigned ndma_u = dkv3_uni((unsigned)ndma), slot_u = dkv3_uni((unsigned)(t & 3)), wave_u = dkv3_uni((unsigned)wave);
            const unsigned part_u = dkv3_uni(part ? 1u : 0u);
            // always-valid sources for the requests of the steps with no tile left (they land in the dummy chunk): the tensors' first 16 rows
            // (the launcher requires S * T >= 16) and this pair's first statistics row (the lse2 region is padded by a row)
            const unsigned long long safe_q = dkv3_uni64(qp), safe_do = dkv3_uni64(dop);
            const unsigned long long safe_l = dkv3_uni64(p.lse2 + ((int64_t)s * p.H + hd) * p.T), safe_n = safe_l;
#ifdef HALVA_STAMP
            unsigned long long run0, run1;
            DKV3_NOW(run0);
#endif
            asm volatile(
#include "sdpa_dkv3_loop.inc"
                : "+{a[0:15]}"(st.accV[0]), "+{a[16:31]}"(st.accV[1]), "+{a[32:47]}"(st.accV[2]), "+{a[48:63]}"(st.accV[3]), "+{a[64:79]}"(st.accK[0]), "+{a[80:95]}"(st.accK[1]), "+{a[96:111]}"(st.accK[2]), "+{a[112:127]}"(st.accK[3])
                : "{a[128:131]}"(st.kq[0]), "{a[132:135]}"(st.kq[1]), "{a[136:139]}"(st.kq[2]), "{a[140:143]}"(st.kq[3]), "{a[144:147]}"(st.kq[4]), "{a[148:151]}"(st.kq[5]), "{a[152:155]}"(st.kq[6]), "{a[156:159]}"(st.kq[7]),
                  "{a[160:163]}"(st.vq[0]), "{a[164:167]}"(st.vq[1]), "{a[168:171]}"(st.vq[2]), "{a[172:175]}"(st.vq[3]), "{a[176:179]}"(st.vq[4]), "{a[180:183]}"(st.vq[5]), "{a[184:187]}"(st.vq[6]), "{a[188:191]}"(st.vq[7]), "v"(rowrel), "v"(colrel),
                  "v"(statrel), "v"(voff_q), "v"(voff_do), "s"(ds_ptr), "s"(sc), "s"(n0_u), "s"(ndma_u), "s"(slot_u), "s"(wave_u), "s"(q_ptr), "s"(do_ptr),
                  "s"(q_piece), "s"(do_piece), "s"(lse_ptr), "s"(nd_ptr), "v"(lo0), "v"(range), "v"(alt_q[0]), "v"(alt_q[1]), "v"(alt_q[2]), "v"(alt_q[3]),
                  "v"(alt_do[0]), "v"(alt_do[1]), "v"(alt_do[2]), "v"(alt_do[3]), "v"(alt_stat), "s"(part_u), "s"(n1_u), "s"(n2_u), "s"(safe_q), "s"(safe_do),
                  "s"(safe_l), "s"(safe_n), "s"(first_u), "s"(npro_u), "s"(proalt_u)
                :
#include "sdpa_dkv3_loop_clobbers.inc"
            );
#ifdef HALVA_STAMP
            DKV3_NOW(run1);
            stamp[2] += run1 - run0;
            stamp[3] += n1;
            stamp[5] += n0 + n2;
#endif
            t = t1;
            continue;
        }
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
#ifdef HALVA_STAMP
    DKV3_NOW(stamp[6]);
#endif
    if (k_in_T) {
        store_rows_T<D>(dv_row, st.accV, k_valid ? 1.f : 0.f, true, lane);
        store_rows_T<D>(dk_row, st.accK, k_valid ? p.scale : 0.f, true, lane);
    }
#ifdef HALVA_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DKV3_NOW(stamp[7]);
    if (p.dbg && lane == 0 && s == 1 && hd == 3 && wave == 0)      // (a pair in the middle of the launch, not its very first workgroups)
        for (int i = 0; i < 12; ++i) p.dbg[1024 + kb * 12 + i] = stamp[i];
#endif
}

template <int D, bool CAUSAL, bool ASM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sdpa_bwd_dkv3_kernel(const SdpaParams p) {
    static_assert(D == 128, "sdpa_bwd_dkv3 is the head_dim-128 instantiation");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int s, hd, b;
    map_block(blockIdx.x, CAUSAL ? (p.nblk + 1) / 2 : p.nblk, p.H, p.npairs, false, s, hd, b);
    const int start = p.seq_start ? p.seq_start[s] : 0;
    const int len = p.seq_len ? p.seq_len[s] : p.T;
    const Branch br = load_branch(p, s);
    // under the causal mask key block b is visited by (nblk - b) query blocks: pair b with nblk-1-b (ONE copy of the block code: a loop)
    const int second = (CAUSAL && b != p.nblk - 1 - b) ? p.nblk - 1 - b : -1;
    WG_CLOCK_BEGIN();
#pragma unroll 1
    for (int pass = 0; pass < (second >= 0 ? 2 : 1); ++pass)
        sdpa_bwd_dkv3_block<CAUSAL, ASM>(p, smem, s, hd, pass ? second : b, wave, lane, start, len, br);
    WG_CLOCK_END(p.dbg, 3);
}
