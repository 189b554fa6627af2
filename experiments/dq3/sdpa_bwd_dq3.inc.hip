// experiments/dq3 (round 5): sdpa_bwd_dq3 as it was measured - NOT shipped (README.md here).  This text sat in halva_amd/csrc/sdpa.hip in front of
// `#include "sdpa_dkv3.h"`; launch_bwd selected it in place of sdpa_bwd_dq2 with
//     const size_t lds_dq3 = 2 * 64 * D * 2 + 8 * 2 * DS_LDS_SLOT;
//     launch_one(sdpa_bwd_dq3_kernel<D, false>, p, true, 256, 512, lds_dq3, S, st, "sdpa_bwd_dq3");
// sdpa_bwd_dq3 (round 5): the same product, dS and K fetched THREE tiles ahead through registers.  sdpa_bwd_dq2 keeps two tiles (96 KiB) per CU in
// flight through its LDS ring and is paced by exactly that (experiments/ds_residency: a workgroup alone on its CU reads dS at 25 GB/s whether the bytes
// come from the Infinity Cache or from HBM; the chip-wide 4.7 TB/s is 72 KiB per CU over ~3 us of loaded latency).  LDS cannot hold a deeper ring
// (3 x 52.9 KiB = 158 of 160 KiB), registers can: a wave asks for its 4 KiB of dS of tile i + 3 with four 16-byte loads per lane and for its share of
// the K tile with two (24 registers per tile, three tiles = 144 KiB per CU in flight), writes a landed tile into LDS just before the step that reads it
// (dS into a PRIVATE double buffer of the wave's own - only the wave itself reads it back, transposed, as before -, K into a double buffer the
// workgroup shares) and multiplies as sdpa_bwd_dq2 does.  Plain loads, counted by the compiler: hand-issued (asm) register loads cannot be kept in
// flight across compiler-scheduled code - the compiler copies registers it believes written (measured on the first version: garbage dS).  For its
// counts to be exact every step issues the same six loads: a tile that is not live for the wave (above its diagonal) reads the wave's first piece
// again (an L2 hit), and the walk is padded to a multiple of three steps with repeats of its last tile.
template <int D, bool SLOW_TR>
__device__ __forceinline__ void sdpa_bwd_dq3_block(const SdpaParams& p, char* smem, int s, int hd, int qb, int wave, int lane) {
    constexpr int NW = 8, BN = 64, DT = D / 32, BM = 32 * NW;
    constexpr int TILE_BYTES = BN * D * 2;
    typedef __attribute__((address_space(3))) char lchar;
    char* k_lds = smem;                                                  // [2][BN][D]
    char* ds_lds = smem + 2 * TILE_BYTES + wave * (2 * DS_LDS_SLOT);     // per wave: [2][2 strips][2 pieces of 1 KiB + 128 B]
    const int h = lane >> 5;
    const int start = p.seq_start ? p.seq_start[s] : 0;
    const int len = p.seq_len ? p.seq_len[s] : p.T;
    const int64_t seq_row0 = (int64_t)s * p.T;
    const int lq0 = qb * BM;
    if (lq0 >= len) return;                              // workgroup-uniform
    const int wr0 = lq0 + 32 * wave;
    const int lq = wr0 + (lane & 31);
    const bool q_valid = lq < len;
    const bool wave_live = wr0 < len;
    const Branch br = load_branch(p, s);
    const bool wave_in_b = wr0 >= br.b;
    const int first_tile = max(0, start) / BN;
    const int last_key_local = min(len, lq0 + BM) - 1;
    const int ntile_end = (start + last_key_local) / BN + 1;
    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const bf16_t* kp = p.k + hd * D;
    const int64_t krow0 = seq_row0 + start;
    const int step = wr0 / 64, sub = (wr0 / 32) & 1;
    const char* ds_pair = p.ds_ws + ((int64_t)s * p.H + hd) * p.ds_nkb * p.ds_nt * 16384;
    auto tile_live = [&](int kt) { return wave_live && (kt * BN - start) <= wr0 + 31; };
    auto strip_hidden = [&](int kt, int si) {
        const int k0 = kt * BN + 32 * si - start;
        return wave_in_b && k0 >= br.a && k0 + 31 < br.b;
    };
    int skip_lo = ntile_end, skip_hi = ntile_end;      // (the walk's jump over the tiles of [br.a, br.b): sdpa_bwd_dq2_block)
    if (start == 0 && lq0 >= br.b) {
        skip_lo = min(ntile_end, max(first_tile, (br.a + BN - 1) / BN));
        skip_hi = max(skip_lo, min(ntile_end, br.b / BN));
    }
    const int n_lo = skip_lo - first_tile, n_walk = n_lo + (ntile_end - skip_hi);
    if (n_walk <= 0) {      // (workgroup-uniform; cannot happen for a block with rows - its own diagonal tile exists)
        if (q_valid) store_rows_T<D>(p.dq + (seq_row0 + start + lq) * p.ld_qkv + hd * D, acc, 0.f, true, lane);
        return;
    }
    auto tile_at = [&](int i) {      // positions past the walk repeat its last tile (never multiplied)
        i = min(i, n_walk - 1);
        return i < n_lo ? first_tile + i : skip_hi + (i - n_lo);
    };
    const char* ds_mine = ds_pair + (int64_t)step * 16384 + sub * 2048 + lane * 16;      // this lane's 16 bytes of (key block 0, strip 0, piece 0)
    typedef Stage<D, BN, 64 * NW> KStage;
    auto issue = [&](int i, KStage& ks, u32x4 (&set)[4]) {
        const int kt = tile_at(i);
        ks.load_clamped(kp, p.ld_qkv, krow0, kt * BN - start, len);
        // a tile that is not live for this wave: the same six loads, from bytes that are in the L2 (counts stay exact, nothing is used)
        const char* src = (i < n_walk && tile_live(kt)) ? ds_mine + (int64_t)(kt >> 1) * p.ds_nt * 16384 + 2 * (kt & 1) * 4096 : ds_mine + (int64_t)(tile_at(0) >> 1) * p.ds_nt * 16384 + 2 * (tile_at(0) & 1) * 4096;
#pragma unroll
        for (int c = 0; c < 4; ++c) set[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + (c >> 1) * 4096 + (c & 1) * 1024));
    };
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, hb = g >> 1;
    const int ds_rd0 = ds_lds_off(4 * hb + q4, 4 * (g & 1) + pp);
    auto consume = [&](int i, KStage& ks, u32x4 (&set)[4]) {
        const int kt = tile_at(i);
        const bool live = i < n_walk && tile_live(kt);
        char* ktile = k_lds + (i & 1) * TILE_BYTES;
        char* mine = ds_lds + (i & 1) * DS_LDS_SLOT;
        ks.store(ktile);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((lchar*)(mine + (c >> 1) * DS_LDS_STRIP + (c & 1) * DS_LDS_PIECE + lane * 16)) = set[c];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // K tile i is there for every wave; the slot of tile i + 1 has been read by all
        issue(i + 3, ks, set);                                                     // (into the registers just written out)
        if (live) {
#pragma unroll
            for (int ks4 = 0; ks4 < 4; ++ks4) {               // 16 keys each: strip ks4 >> 1, half ks4 & 1
                if (strip_hidden(kt, ks4 >> 1)) continue;
                s16x8 zb;
                if (SLOW_TR) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = 16 * (ks4 & 1) + 8 * (j >> 2) + 4 * h + (j & 3), qq = lane & 31;
                        zb[j] = *reinterpret_cast<const short*>(mine + (ks4 >> 1) * DS_LDS_STRIP + ds_lds_off(key, qq >> 2) + (qq & 3) * 2);
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const char* a = mine + (ks4 >> 1) * DS_LDS_STRIP + ds_rd0 + 16 * (16 * (ks4 & 1) + 8 * jj);
                        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
                        zb[4 * jj + 0] = t[0];
                        zb[4 * jj + 1] = t[1];
                        zb[4 * jj + 2] = t[2];
                        zb[4 * jj + 3] = t[3];
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma32(frag_cols<D, SLOW_TR>(ktile, 16 * ks4, 32 * dt, lane), zb, acc[dt]);
            }
        }
    };
    KStage k0, k1, k2;
    u32x4 set0[4], set1[4], set2[4];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the previous row block's readers are done
    issue(0, k0, set0);
    issue(1, k1, set1);
    issue(2, k2, set2);
#pragma unroll 1
    for (int i = 0; i < n_walk; i += 3) {
        consume(i, k0, set0);
        consume(i + 1, k1, set1);
        consume(i + 2, k2, set2);
    }
    if (q_valid) {
        bf16_t* dq_row = p.dq + (seq_row0 + start + lq) * p.ld_qkv + hd * D;
        if constexpr (D == 128) {
            if (p.rope_cos) {
                const int pos = rope_position(start + lq, br);
                store_rows_T_rope<D>(dq_row, acc, p.scale, true, lane, p.rope_cos + (int64_t)pos * (D / 2), p.rope_sin + (int64_t)pos * (D / 2));
                return;
            }
        }
        store_rows_T<D>(dq_row, acc, p.scale, true, lane);
    }
}

template <int D, bool SLOW_TR>
__global__ __launch_bounds__(512) void sdpa_bwd_dq3_kernel(const SdpaParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int s, hd, b;
    map_block(blockIdx.x, (p.nblk + 1) / 2, p.H, p.npairs, false, s, hd, b);
    int heavy, light;
    paired_blocks(p.nblk, p.seq_start ? p.seq_start[s] : 0, load_branch(p, s), b, heavy, light);
    const int npass = (heavy != light) ? 2 : 1;
    WG_CLOCK_BEGIN();
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) sdpa_bwd_dq3_block<D, SLOW_TR>(p, smem, s, hd, pass ? light : heavy, wave, lane);
    WG_CLOCK_END(p.dbg, 2);
}

