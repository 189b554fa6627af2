// Fused self-attention for gfx950 (CDNA4): forward, dQ and dK/dV kernels on v_mfma_f32_32x32x16_bf16.
//
// Replaces flash_attn_varlen_qkvpacked_func + unpad_input/pad_input of the reference
// (llava/train/llama_flash_attn_monkey_patch.py:71-91) for Llama (causal, head_dim 128) and HF CLIPAttention's
// softmax(QK^T)V for the frozen vision tower (non-causal, head_dim 64, forward only).
//
// Structure (all three kernels): a workgroup is 4 waves (256 threads); each wave owns a 32-row strip of the
// "stationary" operand in registers (Q rows in fwd/dQ, K/V rows in dK/dV) and the workgroup streams 64-row
// tiles of the other operand through LDS (double buffered; a 3-slot ring in dK/dV), fetched by LDS-DMA: the requests for
// tile i+1 are issued before the MFMAs of tile i and waited for at the barrier that ends it.
//
//   fwd    S^T = K Q^T (key on the accumulator rows, query on the lane) so the softmax row statistics are
//          lane-local: 32 in-register max/add + one cross-half shuffle; P^T is cast to bf16 in registers and is
//          already the B operand of O^T += V^T P^T; V^T comes from the row-major LDS tile via ds_read_b64_tr_b16.
//   dQ     same skeleton: S^T, dP^T = V dO^T, dZ^T = P^T (dP^T - delta), dQ^T += K^T dZ^T (K^T by tr reads); forms delta.
//   dK/dV  S = Q K^T and dP = dO V^T with the KEY on the lane (K/V fragments stay in registers); P and dZ
//          accumulators are directly the B operands of dV^T += dO^T P and dK^T += Q^T dZ (Q^T/dO^T by tr reads).  The two
//          waves of a SIMD split a key strip between them (sdpa_bwd_dkv2_*).
// LDS tiles use one XOR swizzle that is conflict-free for both ds_read_b128 row reads and transposed reads.
// No atomics: dQ has its own kernel, so results are bitwise reproducible.
#include "common.h"

#include <atomic>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

#ifdef HALVA_STAMP
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        stamp_acc[i] += t_ - stamp_prev;                                                           \
        stamp_prev = t_;                                                                           \
    } while (0)
unsigned long long* g_dbg = nullptr;
extern "C" unsigned long long* halva_dbg_buffer() {
    if (!g_dbg) {
        (void)hipMalloc(&g_dbg, 8192 * 8);
        (void)hipMemset(g_dbg, 0, 8192 * 8);
    }
    return g_dbg;
}
// whole-workgroup clock stamp: cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of every 8th workgroup -> in-kernel clock of THIS kernel
// (MI355X_MICROARCH.md, DVFS give-back item 6); slot k of the debug buffer's upper half: {cycles, ticks, start tick, block}
#ifndef HALVA_STAMP_STRIDE
#define HALVA_STAMP_STRIDE 8      // 8: the first 120 workgroups of XCD 0;  17: every XCD and every round of a 2048-workgroup launch
#endif
#define WG_CLOCK_BEGIN() const unsigned long long wgc0_ = __builtin_amdgcn_s_memtime(), wgr0_ = __builtin_amdgcn_s_memrealtime()
#define WG_CLOCK_END(dbg, region)                                                                                     \
    do {                                                                                                              \
        if ((dbg) && threadIdx.x == 0 && blockIdx.x % HALVA_STAMP_STRIDE == 0 && blockIdx.x / HALVA_STAMP_STRIDE < 120) { \
            unsigned long long* o_ = (dbg) + 4096 + (region)*480 + (blockIdx.x / HALVA_STAMP_STRIDE) * 4;             \
            o_[0] = __builtin_amdgcn_s_memtime() - wgc0_;                                                             \
            o_[1] = __builtin_amdgcn_s_memrealtime() - wgr0_;                                                         \
            o_[2] = wgr0_;                                                                                            \
            o_[3] = blockIdx.x;                                                                                       \
        }                                                                                                             \
    } while (0)
#else
#define STAMP(i)
#define WG_CLOCK_BEGIN()
#define WG_CLOCK_END(dbg, region)
#endif

struct SdpaParams {
    const bf16_t* q;      // [S, T, ...] row stride ld_qkv, head offset hd * D
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* o;            // fwd: out; bwd: unused
    const bf16_t* o_in;   // bwd: forward output
    const bf16_t* d_o;    // bwd: grad of out, row stride ld_o
    bf16_t* dq;           // bwd outputs, row stride ld_qkv (packed like q/k/v)
    bf16_t* dk;
    bf16_t* dv;
    float* lse;           // [S, H, T]
    float* delta;         // [S, H, T]
    float* lse2;          // sdpa_bwd_dkv3's row statistics, written by the delta pass (tail of the dS workspace; nullptr without one): per (sequence,
                          // head) stat_nt records of 512 bytes, one per 64-row step in SEQUENCE coordinates: [lse * log2(e) x 64][-delta x 64]
    int stat_nt;          // records per (sequence, head) = ceil(T / 64)
    int sched_order;      // order of the items inside a queue (sdpa_dkv3.h)
    int* sched;           // sdpa_bwd_dkv3's eight work-queue counters, 128 B apart (behind lse2 in the workspace), zeroed by the delta pass
    int* items;           // sdpa_bwd_dkv3's item records, 64 dwords each in queue order + one all-zero record (behind the counters), written by the delta pass
    const int32_t* seq_start;
    const int32_t* seq_len;
    const int32_t* br_a;  // optional per-sequence branch points (local indices; br_b a multiple of 64), include/halva_hip.h:
    const int32_t* br_b;  // rows [br_b, len) do not attend to rows [br_a, br_b)
    int64_t ld_qkv;       // elements between consecutive tokens in q/k/v
    int64_t ld_o;         // elements between consecutive tokens in out
    int64_t ld_do;        // elements between consecutive tokens in dout
    int T, H;
    int nblk, npairs;     // row blocks per (sequence, head) pair; number of pairs (S * H)
    unsigned long long* dbg;   // diagnostic builds only
    char* ds_ws;          // backward: dS = P o (dP - delta) as bf16 in the dK/dV kernel's register layout (see ds_chunk), or nullptr
    int ds_nkb, ds_nt;    // key blocks of 128 / query steps of 64 per (sequence, head) in ds_ws
    float scale;          // softmax scale
    // backward, optional: the inverse RoPE of dq / dk applied in the store epilogues of sdpa_bwd_dq2 / sdpa_bwd_dkv3 (halva_sdpa_branch_bwd_rope);
    // a row's position follows from its index and the branch points (rope_position below)
    const bf16_t* rope_cos;      // [max_pos, D / 2] bf16 tables of halva_rope_qk, or nullptr = dq / dk leave un-rotated
    const bf16_t* rope_sin;
    int rope_max_pos;            // rows of the tables
    int repair;           // sdpa_fwd_kernel behind sdpa_fwd3: redo only the row blocks that hold a valid row with a non-finite lse (launch_fwd)
};

// Two responses sharing one prefix are packed as [prefix | A | pad | B] in one sequence: B (rows >= b, b a multiple of 64 so
// that no 32-row strip and no 64-key tile straddles it) must not see [a, b) = A and the padding.  For a strip of B rows a
// key tile is therefore either untouched, wholly hidden (dropped), or cut at `a` - which is the ordinary "sequence ends at a"
// mask.  Without branch points a = b = INT_MAX and nothing changes.
struct Branch {
    int a, b;
};
__device__ __forceinline__ Branch load_branch(const SdpaParams& p, int s) {
    Branch br;
    br.a = p.br_a ? p.br_a[s] : 0x7fffffff;
    br.b = p.br_b ? p.br_b[s] : 0x7fffffff;
    return br;
}

// RoPE position of row t (index inside its T rows) of a sequence: its index - and for a branch-packed row [prefix | A | pad | B] the rows of B
// continue from the prefix (include/halva_hip.h, halva_sdpa_branch_fwd: "RoPE positions of branch B restart at br_a"; halva_amd/splice.py:pack_pairs)
__device__ __forceinline__ int rope_position(int t, const Branch& br) { return t >= br.b ? br.a + (t - br.b) : t; }

// Byte offset of 16-byte chunk `ch` of row `row` in a [rows][D] bf16 LDS tile.  The tile is cut into 8-row x 32-column
// subtiles of 512 B; inside a subtile the four chunks of a row are XOR-ed with (row >> 2) & 3.  Conflict-free for the
// ds_read_b128 row reads of an MFMA A/B operand and for ds_read_b64_tr_b16 transposed reads alike, and - unlike a
// whole-row XOR - every fragment address is one of TWO per-lane bases plus an immediate (ch >> 2 and row >> 3 only add
// multiples of 512 B), which keeps ~40 VGPRs of address arithmetic out of the main loops.
template <int D>
__device__ __forceinline__ int tile_off(int row, int ch) {
    constexpr int SUBROW = (D / 32) * 512;   // bytes of one 8-row band
    return SUBROW * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

__device__ __forceinline__ f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A/B fragment of a row-major tile: lane (r = lane & 31, h = lane >> 5) gets row (row0 + r), elements 16*ks + 8*h .. +7
template <int D>
__device__ __forceinline__ s16x8 frag_rows(const char* tile, int row0, int ks, int lane) {
    const int r = row0 + (lane & 31);
    return *reinterpret_cast<const s16x8*>(tile + tile_off<D>(r, 2 * ks + (lane >> 5)));
}

// Transposed fragment for a product that sums over the tile's ROW index with an accumulator tile as the other
// operand.  Lane (c = lane & 31, h = lane >> 5) gets column (col0 + c) of rows
//   row0 + 8*jj + 4*h + e,  jj = 0,1, e = 0..3   (element j = 4*jj + e)
// which is exactly the row order of registers 8*s'..8*s'+7 of a 32x32 accumulator (row0 = 16*s' + tile base).
template <int D, bool SLOW>
__device__ __forceinline__ s16x8 frag_cols(const char* tile, int row0, int col0, int lane) {
    s16x8 out;
    if (SLOW) {
        const int c = col0 + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = row0 + 8 * (j >> 2) + 4 * h + (j & 3);
            out[j] = *reinterpret_cast<const short*>(tile + tile_off<D>(r, c >> 3) + (c & 7) * 2);
        }
    } else {
        // ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block; lane 4q+p supplies the address of
        // row q, columns 4p..4p+3 and receives column (lane & 15), rows 0..3.
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = g >> 1;
        const int c = col0 + 16 * (g & 1) + 4 * pp;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int r = row0 + 8 * jj + 4 * h + q;
            const int off = tile_off<D>(r, c >> 3) + (c & 7) * 2;
            const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off));
            out[4 * jj + 0] = t[0];
            out[4 * jj + 1] = t[1];
            out[4 * jj + 2] = t[2];
            out[4 * jj + 3] = t[3];
        }
    }
    return out;
}

// registers 8*s..8*s+7 of a 32x32 f32 accumulator -> bf16 fragment usable as the B (or A) operand
__device__ __forceinline__ s16x8 acc_to_frag(const f32x16& x, int s) {
    s16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (short)f32_to_bf16(x[8 * s + j]);
    return out;
}

// row index (0..31) inside a 32x32 accumulator tile of register `reg` on a lane of half h
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------------
// cooperative tile staging: ROWS x D bf16, 256 threads, register staged
// ---------------------------------------------------------------------------------------------------
template <int D, int ROWS, int NT = 256>
struct Stage {
    static constexpr int NCH = D / 8;
    static constexpr int PER_THREAD = ROWS * NCH / NT;
    static_assert(ROWS * NCH % NT == 0, "tile must split evenly over the workgroup");
    u32x4 r[PER_THREAD];

    // rows outside [0, limit) read the nearest valid row (finite data; callers mask those rows): branch-free
    __device__ __forceinline__ void load_clamped(const bf16_t* base, int64_t ld, int64_t grow_local0, int local0, int limit) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int cid = threadIdx.x + NT * i;
            const int row = cid / NCH, ch = cid % NCH;
            const int loc = min(max(local0 + row, 0), limit - 1);
            r[i] = *reinterpret_cast<const u32x4*>(base + (grow_local0 + loc) * ld + ch * 8);
        }
    }
    __device__ __forceinline__ void store(char* tile) const {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int cid = threadIdx.x + NT * i;
            const int row = cid / NCH, ch = cid % NCH;
            *reinterpret_cast<u32x4*>(tile + tile_off<D>(row, ch)) = r[i];
        }
    }
};

// Fill one [64][D] tile image (the tile_off layout) straight from global memory, no register staging: each
// global_load_lds_dwordx4 writes 1 KiB of LDS at (wave-uniform base + 16 * lane), so lane l of chunk c fetches the 16 bytes
// whose tile_off is 1024 c + 16 l - the swizzle is applied to the SOURCE address.  Rows outside [0, limit) read the nearest
// valid row, as Stage::load_clamped.  NW waves share the tile's chunks.
template <int D, int NW, int ROWS = 64>      // ROWS = 128: two consecutive 64-row images
__device__ __forceinline__ void stage_tile_dma(char* tile, const bf16_t* base, int64_t ld, int64_t grow_local0, int local0, int limit,
                                               int wave, int lane) {
    constexpr int CHUNKS = ROWS * D * 2 / 1024, SUBROW = (D / 32) * 512;
    static_assert(CHUNKS % NW == 0, "chunks must split evenly over the waves");
#pragma unroll
    for (int i = 0; i < CHUNKS / NW; ++i) {
        const int c = wave + NW * i;
        const int o = 1024 * c + 16 * lane;
        const int band = o / SUBROW, rem = o % SUBROW;
        const int row = 8 * band + ((rem % 512) >> 6);
        const int ch = 4 * (rem / 512) + (((rem >> 4) & 3) ^ ((row >> 2) & 3));
        const int loc = min(max(local0 + row, 0), limit - 1);
        const bf16_t* src = base + (grow_local0 + loc) * ld + ch * 8;
        // written as asm: the builtin makes hipcc drain vmcnt(0) before the next ds_read_b64_tr_b16, i.e. in the middle of the step
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(tile + 1024 * c);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
}
// The same fill for a run of whole tiles 64 rows apart, with next to no per-lane state: a wave's chunks c = wave + NW i hold the same
// (row, chunk) pattern shifted by a whole number of rows, so ONE 32-bit lane offset serves all of them and the rest of the address -
// tile origin + that row shift - is scalar (the saddr form of the instruction).  No 64-bit vector arithmetic in the loop.
template <int D, int NW, int ROWS = 64>
struct TileDma {
    static constexpr int CHUNKS = ROWS * D * 2 / 1024, SUBROW = (D / 32) * 512, PER_WAVE = CHUNKS / NW;
    static_assert((1024 * NW) % SUBROW == 0, "chunks of one wave must differ by whole 8-row bands");
    static constexpr int ROWS_PER_I = 8 * (1024 * NW / SUBROW);
    unsigned voff;          // byte offset of this lane's 16 bytes of chunk `wave` from the tile's first row
    const char* origin;     // the next tile's first row (wave-uniform)
    __device__ __forceinline__ void init(const bf16_t* base, int64_t ld, int64_t grow_local0, int local0, int wave, int lane) {
        const int o = 1024 * wave + 16 * lane;
        const int band = o / SUBROW, rem = o % SUBROW;
        const int row = 8 * band + ((rem % 512) >> 6);
        const int ch = 4 * (rem / 512) + (((rem >> 4) & 3) ^ ((row >> 2) & 3));
        voff = (unsigned)((row * ld + ch * 8) * 2);
        origin = reinterpret_cast<const char*>(base + (grow_local0 + local0) * ld);
    }
    // fetch the tile at `origin` (all ROWS rows must exist), then move one tile down
    __device__ __forceinline__ void issue_and_advance(char* tile, int64_t ld, int wave) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(tile + 1024 * (wave + NW * i));
            const char* rows = origin + (int64_t)i * ROWS_PER_I * ld * 2;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rows) : "memory");
        }
        origin += ROWS * ld * 2;
    }
};
// the loads above are invisible to the compiler's counters: wait for them by hand before the barrier that publishes the tile
// (s_waitcnt vmcnt(0) as the builtin, not asm: the compiler then also knows that nothing of its own is pending afterwards)
__device__ __forceinline__ void stage_tile_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Write a [D x 32] transposed accumulator (lane = row of the output, registers = columns d) as bf16 rows.  A lane holds the columns
// 32*dt + 8*g + 4*h + (0..3) of its row (h = lane / 32: the two lanes of a row sit 32 apart), i.e. 8-byte pieces: 16 stores per lane,
// and the store tail of a row block is bound by the NUMBER of store instructions (measured: ~5 600 cycles for a V-side wave of the
// dK/dV kernel, ~10 000 for the K-side wave that finishes last).  v_permlane32_swap trades the pieces of two neighbouring groups
// between the two lanes of a row, after which each holds 16 contiguous bytes: 8 stores per lane, same bytes, same addresses.
// the dq / dk / dv rows leave through this.  -DHALVA_ROWS_NT=1 (with FWD3_O_NT=1 for the forward's generator) writes them nontemporally: measured
// round 5 and NOT kept - sdpa_bwd_dq2 +4 %, sdpa_bwd_dkv3 +0.5 %, sdpa_fwd3 +0.6 % at the step's shapes (alternating runs, rocprofv3)
#ifndef HALVA_ROWS_NT
#define HALVA_ROWS_NT 0
#endif
#if HALVA_ROWS_NT
#define HALVA_ROW_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define HALVA_ROW_STORE(ptr, val) (*(ptr) = (val))
#endif
// one [32 x 32] tile of a transposed accumulator (columns 32 dt .. 32 dt + 31 of the lanes' rows): two 16-byte stores per lane
__device__ __forceinline__ void store_tile_T(bf16_t* row_ptr_dt, const f32x16& t, float mul, int h) {
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        unsigned w[2][2];      // [group 2gp, 2gp+1][word]
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int g = 2 * gp + k;
            w[k][0] = pack_bf16x2(t[4 * g + 0] * mul, t[4 * g + 1] * mul);
            w[k][1] = pack_bf16x2(t[4 * g + 2] * mul, t[4 * g + 3] * mul);
        }
        // upper lanes' group-2gp words <-> lower lanes' group-(2gp+1) words
        const auto x = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        HALVA_ROW_STORE(reinterpret_cast<u32x4*>(row_ptr_dt + 16 * gp + 8 * h), (u32x4{x[0], y[0], x[1], y[1]}));
    }
}
template <int D>
__device__ __forceinline__ void store_rows_T(bf16_t* row_ptr, const f32x16 (&acc)[D / 32], float mul, bool valid, int lane) {
    if (!valid) return;      // (both lanes of a row take the same side: the swap below never pairs an active lane with an inactive one)
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) store_tile_T(row_ptr + 32 * dt, acc[dt], mul, h);
}
// The same rows with the INVERSE RoPE applied on the way out (halva_sdpa_branch_bwd_rope): what halva_rope_qk(inverse = 1) would do to the
// stored row in a launch of its own - the row is rounded to bf16 first, rotated in fp32 with the bf16 table entries of its position, rounded
// again (rope_pair, common.h: the same expression as rope_qk_kernel).  Elements d and d + 64 of a row sit in the same lane, same register
// index, accumulator tiles dt and dt + 2.  cr / sr: this lane's row of the cos / sin tables ([D / 2] bf16).
// The table rows of a wave's 32 consecutive positions (pos0 .. pos0 + 31: a 32-row group never straddles a branch point) = 4 KiB of cos + 4 KiB of sin,
// CONTIGUOUS in the tables: fetched with four coalesced 16-byte loads per lane and table and handed to the lanes through `scratch` (9 KiB of LDS that
// only this wave touches, rows 144 bytes apart).  The first version let every lane gather its own row - 16 loads of 8 bytes per lane, 32 different
// 128-byte lines per instruction: +28 .. +40 us per sdpa_bwd_dq2 launch (profiles/r05_rope_cost.log).
constexpr int ROPE_LDS_ROW = 144, ROPE_LDS_BYTES = 2 * 32 * ROPE_LDS_ROW;
__device__ __forceinline__ void rope_rows_to_lds(char* scratch, const bf16_t* cos, const bf16_t* sin, int pos0, int max_pos, int lane) {
    typedef __attribute__((address_space(3))) char lchar;
    u32x4 c[4], sn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), ch = lane & 7;
        const int64_t at = (int64_t)min(pos0 + row, max_pos - 1) * 64 + ch * 8;
        c[i] = *reinterpret_cast<const u32x4*>(cos + at);
        sn[i] = *reinterpret_cast<const u32x4*>(sin + at);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), ch = lane & 7;
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((lchar*)(scratch + row * ROPE_LDS_ROW + ch * 16)) = c[i];
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((lchar*)(scratch + 32 * ROPE_LDS_ROW + row * ROPE_LDS_ROW + ch * 16)) = sn[i];
    }
}
// The same rows with the INVERSE RoPE applied on the way out (halva_sdpa_branch_bwd_rope): what halva_rope_qk(inverse = 1) would do to the
// stored row in a launch of its own - the row is rounded to bf16 first, rotated in fp32 with the bf16 table entries of its position, rounded
// again (rope_pair, common.h: the same expression as rope_qk_kernel).  Elements d and d + 64 of a row sit in the same lane, same register
// index, accumulator tiles dt and dt + 2.  scratch: rope_rows_to_lds' block of this wave (row lane & 31 = this lane's row).
template <int D>
__device__ __forceinline__ void store_rows_T_rope(bf16_t* row_ptr, const f32x16 (&acc)[D / 32], float mul, bool valid, int lane, const char* scratch) {
    static_assert(D == 128, "the rotating store is the head_dim-128 instantiation");
    typedef __attribute__((address_space(3))) const char lchar;
    const int h = lane >> 5;
    lchar* cr = (lchar*)(scratch + (lane & 31) * ROPE_LDS_ROW + 8 * h);
    lchar* sr = cr + 32 * ROPE_LDS_ROW;
#pragma unroll
    for (int dtl = 0; dtl < 2; ++dtl) {
        f32x16 lo, hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2 cw = *reinterpret_cast<__attribute__((address_space(3))) const u32x2*>(cr + 64 * dtl + 16 * g);
            const u32x2 sw = *reinterpret_cast<__attribute__((address_space(3))) const u32x2*>(sr + 64 * dtl + 16 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = (j & 1) ? bf16_hi(cw[j >> 1]) : bf16_lo(cw[j >> 1]);
                const float sn = (j & 1) ? bf16_hi(sw[j >> 1]) : bf16_lo(sw[j >> 1]);
                float y1, y2;
                rope_pair(bf16_round(acc[dtl][4 * g + j] * mul), bf16_round(acc[dtl + 2][4 * g + j] * mul), c, sn * -1.f, y1, y2);
                lo[4 * g + j] = y1, hi[4 * g + j] = y2;
            }
        }
        if (valid) {      // (both lanes of a row take the same side: the swap in store_tile_T never pairs an active lane with an inactive one)
            store_tile_T(row_ptr + 32 * dtl, lo, 1.f, h);
            store_tile_T(row_ptr + 32 * (dtl + 2), hi, 1.f, h);
        }
    }
}
template <int D>
__device__ __forceinline__ void store_rows_zero(bf16_t* row_ptr, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x2*>(row_ptr + 32 * dt + 8 * g + 4 * h) = u32x2{0u, 0u};
}

// Workgroup -> (sequence, head, row block).  1-D grid; workgroups are dealt round-robin over the 8 XCDs, so the blocks
// L, L+8, L+16, ... share an L2: give those the row blocks of ONE (sequence, head) pair, whose K/V (or Q/dO) tiles they all
// stream, heaviest (latest) block first under the causal mask.  Placement only changes speed, never results.
__device__ __forceinline__ void map_block(int L, int nblk, int H, int npairs, bool heavy_first, int& s, int& hd, int& blk) {
    int pair, o;
    if ((npairs & 7) == 0) {
        const int slot = L >> 3;
        pair = (slot / nblk) * 8 + (L & 7);
        o = slot % nblk;
    } else {
        pair = L / nblk;
        o = L % nblk;
    }
    blk = heavy_first ? nblk - 1 - o : o;
    hd = pair % H;
    s = pair / H;
}

// Which two 256-row blocks of a (sequence, head) pair a workgroup takes under the causal mask.  Plain causal: block b needs b + 1 units of
// key tiles, so b goes with nblk-1-b and every workgroup does nblk + 1 units.  A packed sequence [prefix | A | pad | B] breaks that: the blocks
// wholly inside B (first row >= br.b) do not visit the tiles inside [br.a, br.b), so their work is (b + 1) - hid with hid = (br.b - br.a) / 256 units -
// for the bench's row [668 | 1380 | 1380] the old pairing gave six workgroups of 37-39 units and one of 60 per pair, and the launch waited for
// the sixties.  Here the blocks are RANKED by that work (two increasing runs merged in closed form) and pair k takes the k-th heaviest and the
// k-th lightest (44 units at most in the example).  Without a branch the ranks are the block numbers: the old pairing.
__host__ __device__ __forceinline__ int block_rank(int qb, int n1, int n2, int hid) {      // n1 blocks below br.b, n2 inside B
    if (qb < n1) return qb + min(max(qb - n1 + hid, 0), n2);
    const int j = qb - n1;
    return j + min(max(n1 + j - hid + 1, 0), n1);
}
__host__ __device__ __forceinline__ void paired_blocks(int nblk, int start, const Branch& br, int k, int& heavy, int& light) {
    int n1 = nblk, hid = 0;
    if (br.b != 0x7fffffff && br.b > br.a) {
        n1 = (br.b + start + 255) / 256;      // first block whose first row is >= br.b (local rows: row - start)
        n1 = n1 < nblk ? n1 : nblk;
        hid = (br.b - br.a + 128) / 256;
    }
    const int n2 = nblk - n1, want_h = nblk - 1 - k, want_l = k;
    heavy = nblk - 1 - k, light = k;
    if (n2 == 0 || hid == 0) return;
    for (int qb = 0; qb < nblk; ++qb) {
        const int r = block_rank(qb, n1, n2, hid);
        if (r == want_h) heavy = qb;
        if (r == want_l) light = qb;
    }
}

// ===================================================================================================
// forward
// ===================================================================================================
// 8 waves x 32 query rows (two waves per SIMD), 64-key K/V tiles double-buffered in LDS, register staged (the loads of
// tile t+1 are issued before the MFMAs of tile t and written to LDS after them; one barrier per tile).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float xhalf_max(float v) {   // max with the lane 32 away (v_permlane32_swap: no LDS round trip)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The tile body is two straight-line blocks in which the wave's own vector work rides in the shadow of its own MFMAs
// (the interleave is written out in the source and pinned with scheduling fences):
//   A:  S0 = K[0:32] Q^T ; S1 = K[32:64] Q^T  ||  max(S0), P0 = exp2(S0*sc - m_ref), sum      then max(S1)
//   B:  O^T += V^T[:, 0:32] P0^T              ||  P1 = exp2(S1*sc - m_ref), sum ;  O^T += V^T[:, 32:64] P1^T
// The exponent reference m_ref is only moved when a row's maximum exceeds it by more than 2^RESCALE_AT (the first tile
// always does): that rare path sits between A and B, rescales O / l and recomputes P0 from the untouched S0; O is not
// multiplied every tile.  Cross-half reductions use v_permlane32_swap (no LDS round trip), the exp2 argument / row sums use
// packed fp32 ops, full tiles are fetched by pointer bumps (no per-tile 64-bit address arithmetic).
// Variants built and measured on MI355X at S=8,T=2048,H=32 (numbers and PMC breakdown in DESIGN.md), none faster: 4 waves x
// 64 rows at one wave per SIMD; a ping-pong of GEMM-only / softmax-only phases between the two waves of a SIMD (two barriers
// per tile; the later-dispatched wave of each SIMD is starved whatever s_setprio says); the same with the two blocks in
// rotated order and one barrier; 4-wave workgroups at two per CU.
// ---------------------------------------------------------------------------------------------------
template <bool MASK, bool CAUSAL>
__device__ __forceinline__ void mask_half(f32x16& st, int kbase, int h, int len, int ql) {
    if (!MASK) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kl = kbase + acc_row(r, h);
        if (kl >= len || (CAUSAL && kl > ql)) st[r] = -INFINITY;
    }
}
__device__ __forceinline__ float half_max(const f32x16& st) {
    float a = fmaxf(st[0], st[1]), b = fmaxf(st[2], st[3]);
#pragma unroll
    for (int r = 4; r < 16; r += 4) {
        a = fmaxf(fmaxf(a, st[r]), st[r + 1]);
        b = fmaxf(fmaxf(b, st[r + 2]), st[r + 3]);
    }
    return fmaxf(a, b);
}
// One 2-element slice of P = exp2(S * sc - m): elements r, r+1 of the accumulator -> bf16 pair in the B-operand fragment
__device__ __forceinline__ void exp_pair(const f32x16& st, int r, f32x2 sc2, f32x2 ms2, f32x2& ps, s16x8& p_lo, s16x8& p_hi) {
    f32x2 x = {st[r], st[r + 1]};
    x = x * sc2 + ms2;
    const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
    ps = ps + e;
    const unsigned w = pack_bf16x2(e[0], e[1]);
    if (r < 8) {
        p_lo[r] = (short)(w & 0xffffu);
        p_lo[r + 1] = (short)(w >> 16);
    } else {
        p_hi[r - 8] = (short)(w & 0xffffu);
        p_hi[r - 7] = (short)(w >> 16);
    }
}
__device__ __forceinline__ float half_exp(const f32x16& st, float sc, float m_sub, s16x8& p_lo, s16x8& p_hi) {
    const f32x2 sc2 = {sc, sc}, ms2 = {-m_sub, -m_sub};
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) exp_pair(st, r, sc2, ms2, ps, p_lo, p_hi);
    return ps[0] + ps[1];
}

// The same for scores that come out of the MFMA chain already scaled and shifted (the chain's initial accumulator holds -m_ref and
// Q was multiplied by scale * log2(e) once per row block): P = exp2(S'), no per-element multiply-add at all.
__device__ __forceinline__ void exp_pair0(const f32x16& st, int r, f32x2& ps, s16x8& p_lo, s16x8& p_hi) {
    const f32x2 e = {__builtin_amdgcn_exp2f(st[r]), __builtin_amdgcn_exp2f(st[r + 1])};
    ps = ps + e;
    const unsigned w = pack_bf16x2(e[0], e[1]);
    if (r < 8) {
        p_lo[r] = (short)(w & 0xffffu);
        p_lo[r + 1] = (short)(w >> 16);
    } else {
        p_hi[r - 8] = (short)(w & 0xffffu);
        p_hi[r - 7] = (short)(w >> 16);
    }
}
__device__ __forceinline__ float half_exp0(const f32x16& st, s16x8& p_lo, s16x8& p_hi) {
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) exp_pair0(st, r, ps, p_lo, p_hi);
    return ps[0] + ps[1];
}
// bf16 fragment * c, rounded to bf16 again (once per row block, on the Q fragments)
__device__ __forceinline__ s16x8 scale_frag(const s16x8& f, float c) {
    const u32x4 w = __builtin_bit_cast(u32x4, f);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(bf16_lo(w[i]) * c, bf16_hi(w[i]) * c);
    return __builtin_bit_cast(s16x8, o);
}

// scheduling fence that only LDS reads and scalar ops may cross: pins the MFMA / VALU interleave written in the source
#define FENCE() __builtin_amdgcn_sched_barrier(0x100 | 0x004)

#ifdef HALVA_STAMP
#define STAMP_ARGS , unsigned long long (&stamp_acc)[6], unsigned long long& stamp_prev
#define STAMP_PASS , stamp_acc, stamp_prev
#else
#define STAMP_ARGS
#define STAMP_PASS
#endif
#ifndef HALVA_FWD_CINIT
#define HALVA_FWD_CINIT 0
#endif
// HALVA_FWD_CINIT=1 (measured in round 3, NOT the default): the row constant rides in the MFMA chain.  `qf` holds Q * (scale * log2 e) (bf16, rounded once per row
// block) and both score chains start from the accumulator `minit` = -m_ref in every register (the query sits on the lane), so a score
// leaves the matrix pipe as S' = log2(e) * scale * q.k - m_ref and P = exp2(S') costs ONE vector instruction per element instead of a
// multiply-add plus the exponential (32 fewer vector instructions per wave and tile).  Measured on MI355X: forward 379 -> 372 us (-1.8 %),
// and REJECTED for its numerics: the extra bf16 rounding of Q * c moves a score by ~2^-9 of its magnitude, i.e. P by up to ~1 % for
// scores of a few tens - test_sdpa_exponent_reference_moves_when_later_keys_dominate (scores of 90..230 nat) leaves its 1e-2 bound
// (2.1e-2) and the full-width grouping / prefix-sharing invariance tests see 3x their usual loss noise.  flash-attn keeps the scale
// in fp32 after the product for the same reason; so does the default build.
template <int D, bool CAUSAL, bool MASK, bool SLOW_TR>
__device__ __forceinline__ void fwd_tile(const char* kt, const char* vt, const s16x8 (&qf)[D / 16], f32x16 (&oacc)[D / 32],
                                          float& m_ref, float& l_run, f32x16& minit, float sc, int kv0, int len, int ql, int lane STAMP_ARGS) {
    constexpr int KS = D / 16, DT = D / 32;
    constexpr float RESCALE_AT = 64.f;     // log2 units: P stays below 2^64, far inside fp32 / bf16 range
    static_assert(KS == 8 || KS == 4, "head_dim 128 or 64");
    constexpr int PPS = 8 / KS;            // exp pairs handled per S1 MFMA (1 for D=128, 2 for D=64)
    const int h = lane >> 5;
    f32x16 s0, s1;
    s16x8 p0a, p0b, p1a, p1b;
    const f32x2 sc2 = {sc, sc};
    (void)sc2;
    // ---------------- block A: S0 bare, then S1 with softmax(S0) in its shadow ----------------
#if HALVA_FWD_CINIT
    s0 = minit, s1 = minit;
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
#endif
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s0 = mfma32(frag_rows<D>(kt, 0, ks, lane), qf[ks], s0);
    mask_half<MASK, CAUSAL>(s0, kv0, h, len, ql);
    STAMP(1);
    const float msub_a = (m_ref == -INFINITY) ? 0.f : m_ref;
    const f32x2 ms2a = {-msub_a, -msub_a};
    (void)ms2a;
    f32x2 ps0 = {0.f, 0.f};
    float mx0a = -INFINITY, mx0b = -INFINITY;
    FENCE();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        s1 = mfma32(frag_rows<D>(kt, 32, ks, lane), qf[ks], s1);
        FENCE();
#pragma unroll
        for (int q = 0; q < PPS; ++q) {
            const int r = 2 * (ks * PPS + q);
#if HALVA_FWD_CINIT
            exp_pair0(s0, r, ps0, p0a, p0b);
#else
            exp_pair(s0, r, sc2, ms2a, ps0, p0a, p0b);
#endif
            if (q & 1 || PPS == 1 ? (ks & 1) : false) mx0b = fmaxf(fmaxf(mx0b, s0[r]), s0[r + 1]);
            else mx0a = fmaxf(fmaxf(mx0a, s0[r]), s0[r + 1]);
        }
        FENCE();
    }
    mask_half<MASK, CAUSAL>(s1, kv0 + 32, h, len, ql);
#ifndef HALVA_FWD_NO_PIN
    // P0 and its row sums are dead on the rare path below (which recomputes them), so LLVM sinks the whole exponential block out of
    // the S1 chain into the common successor - behind the chain, where no MFMA covers it.  Pin the values where they are produced.
    asm volatile("" : "+v"(p0a), "+v"(p0b), "+v"(ps0));
#endif
    float sum0 = ps0[0] + ps0[1];
#if HALVA_FWD_CINIT
    const float tmax = xhalf_max(fmaxf(fmaxf(mx0a, mx0b), half_max(s1))) + msub_a;      // scores are relative to msub_a: back to absolute
#else
    const float tmax = xhalf_max(fmaxf(fmaxf(mx0a, mx0b), half_max(s1))) * sc;
#endif
    STAMP(2);
    // ---------------- rare: move the exponent reference ----------------
    if (__any(tmax > m_ref + RESCALE_AT)) {
        const float m_next = fmaxf(m_ref, tmax);
        const float alpha = (m_next == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_ref - m_next);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        m_ref = m_next;
#if HALVA_FWD_CINIT
        // both score tiles were formed against the old reference: shift them (and every later chain's initial accumulator) to the new one
        const float msub_n = (m_ref == -INFINITY) ? 0.f : m_ref;
        const float adj = msub_a - msub_n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] += adj;
            s1[r] += adj;
            minit[r] = -msub_n;
        }
        sum0 = half_exp0(s0, p0a, p0b);
#else
        sum0 = half_exp(s0, sc, (m_ref == -INFINITY) ? 0.f : m_ref, p0a, p0b);
#endif
    }
    STAMP(3);
    // ---------------- block B: PV(P0) with softmax(S1) in its shadow, then PV(P1) bare ----------------
    const float msub_b = (m_ref == -INFINITY) ? 0.f : m_ref;
    const f32x2 ms2b = {-msub_b, -msub_b};
    (void)ms2b;
    f32x2 ps1 = {0.f, 0.f};
    FENCE();
#pragma unroll
    for (int i = 0; i < 2 * DT; ++i) {
        const int ks = i / DT, dt = i % DT;
        oacc[dt] = mfma32(frag_cols<D, SLOW_TR>(vt, 16 * ks, 32 * dt, lane), ks ? p0b : p0a, oacc[dt]);
        FENCE();
#pragma unroll
        for (int q = 0; q < 8 / (2 * DT); ++q) {
#if HALVA_FWD_CINIT
            exp_pair0(s1, 2 * (i * (8 / (2 * DT)) + q), ps1, p1a, p1b);
#else
            exp_pair(s1, 2 * (i * (8 / (2 * DT)) + q), sc2, ms2b, ps1, p1a, p1b);
#endif
        }
        FENCE();
    }
#pragma unroll
    for (int i = 0; i < 2 * DT; ++i) {
        const int ks = i / DT, dt = i % DT;
        oacc[dt] = mfma32(frag_cols<D, SLOW_TR>(vt, 32 + 16 * ks, 32 * dt, lane), ks ? p1b : p1a, oacc[dt]);
    }
    l_run += sum0 + ps1[0] + ps1[1];
    STAMP(4);
}

// Geometry of one 256-row block of one (sequence, head) for this lane / wave.
struct FwdGeom {
    int g0, gq, ql, ntiles, skip_lo, skip_hi, wq_min, wq_max;
    bool q_in_T, q_valid;
    __device__ __forceinline__ int first_tile() const { return skip_lo > 0 ? 0 : skip_hi; }      // == ntiles when the block has no tile
};
template <bool CAUSAL>
__device__ __forceinline__ FwdGeom fwd_geom(const SdpaParams& p, int qb, int start, int len, const Branch& br, int wave, int lane) {
    constexpr int BN = 64, BM = 256;
    FwdGeom g;
    g.g0 = qb * BM;
    g.gq = g.g0 + 32 * wave + (lane & 31);
    g.ql = g.gq - start;
    g.q_in_T = g.gq < p.T;
    g.q_valid = g.q_in_T && g.ql >= 0 && g.ql < len;
    int kv_end = len;
    if (CAUSAL) kv_end = min(len, g.g0 + BM - start);
    g.ntiles = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;
    g.wq_min = g.g0 + 32 * wave - start;
    g.wq_max = g.wq_min + 31;
    // A row block wholly in branch B does not even stage the key tiles that lie wholly inside [a, b): the tile range is walked
    // in (up to) two segments [0, skip_lo) and [skip_hi, ntiles), each a plain double-buffered loop (one barrier per tile).
    g.skip_lo = g.skip_hi = g.ntiles;
    if (g.g0 - start >= br.b) {
        g.skip_lo = min(g.ntiles, (br.a + BN - 1) / BN);
        g.skip_hi = max(g.skip_lo, min(g.ntiles, br.b / BN));
    }
    return g;
}

#ifndef FWD_DMA
#define FWD_DMA 1
#endif
// One row block.  `qf` (the Q fragments) belongs to the caller so that a block can fetch its SUCCESSOR's operands: when the tile loop
// of a block has passed its last barrier the LDS ring is free and the Q registers are dead, so the next block's Q rows and first K/V
// tile are requested THEN - in front of this block's store tail - instead of in the next block's prologue (measured per block,
// s_memtime: 7 000-8 800 cycles from the Q request to the first tile in LDS, 3 000-4 700 for the store tail, against ~4 300 per tile).
//   qb_next    : the block this workgroup runs next (-1: none)          prefetched : this block's operands were requested by its predecessor
//   first_in_wg: nothing of this workgroup has touched the LDS ring yet
template <int D, bool CAUSAL, bool SLOW_TR>
__device__ __forceinline__ void sdpa_fwd_block(const SdpaParams& p, char* smem, int s, int hd, int qb, int qb_next, bool prefetched,
                                               bool first_in_wg, s16x8 (&qf)[D / 16], int start, int len, const Branch br) {
    constexpr int NW = 8, BN = 64, KS = D / 16, DT = D / 32, NT = 64 * NW;
    constexpr int TILE_BYTES = BN * D * 2;
    char* k_lds = smem;                    // [2][BN][D]
    char* v_lds = smem + 2 * TILE_BYTES;   // [2][BN][D]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int64_t seq_row0 = (int64_t)s * p.T;
    const FwdGeom g = fwd_geom<CAUSAL>(p, qb, start, len, br, wave, lane);
    const int gq = g.gq, ql = g.ql, ntiles = g.ntiles, skip_lo = g.skip_lo, skip_hi = g.skip_hi, wq_min = g.wq_min, wq_max = g.wq_max;
    const bool q_in_T = g.q_in_T, q_valid = g.q_valid;

    const bf16_t* kp = p.k + hd * D;
    const bf16_t* vp = p.v + hd * D;
    bf16_t* orow = p.o + (seq_row0 + gq) * p.ld_o + hd * D;
    // K/V tiles arrive by LDS-DMA (no staging registers, no ds_write pass); the slow-transpose debug build keeps register staging
    constexpr bool DMA = !SLOW_TR && FWD_DMA;
    const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // used for the DMA addresses only
    const int64_t krow0 = seq_row0 + start;
    auto request_q = [&](const FwdGeom& gg) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (gg.q_valid)
                qf[ks] = *reinterpret_cast<const s16x8*>(p.q + hd * D + (seq_row0 + gg.gq) * p.ld_qkv + 16 * ks + 8 * h);
            else
                qf[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto request_first_tile = [&](const FwdGeom& gg) {      // DMA builds only; the caller vouches that nobody reads slot 0 any more
        const int t = gg.first_tile();
        if (t < gg.ntiles) {
            stage_tile_dma<D, NW>(k_lds, kp, p.ld_qkv, krow0, t * BN, len, wave_u, lane);
            stage_tile_dma<D, NW>(v_lds, vp, p.ld_qkv, krow0, t * BN, len, wave_u, lane);
        }
    };
    auto prefetch_next = [&]() {
        if (DMA && qb_next >= 0) {
            const FwdGeom gn = fwd_geom<CAUSAL>(p, qb_next, start, len, br, wave, lane);
            request_q(gn);
            request_first_tile(gn);
        }
    };
    const bool pre = DMA && prefetched;

    if (ntiles == 0) {
        if (q_in_T) {
            store_rows_zero<D>(orow, lane);
            if (h == 0 && p.lse) p.lse[((int64_t)s * p.H + hd) * p.T + gq] = 0.f;
        }
        if (!pre && !first_in_wg) __syncthreads();      // (a predecessor's readers; a prefetching predecessor has passed its last barrier)
        prefetch_next();
        return;
    }
    if (!pre) request_q(g);

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_ref = -INFINITY, l_run = 0.f;
    const float sc = p.scale * kLog2e;
    f32x16 minit;
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = 0.f;       // -m_ref, or 0 while the row has seen no key (HALVA_FWD_CINIT)
#if HALVA_FWD_CINIT
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = scale_frag(qf[ks], sc);      // Q * scale * log2(e), once per row block
#endif
    const bool wave_in_b = wq_min >= br.b;             // wave-uniform (br.b is a multiple of 64, strips are 32 rows)
    Stage<D, BN, NT> kst, vst;
    TileDma<D, NW> kdma, vdma;
#ifdef HALVA_STAMP
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_prev, blk_t[4];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
    blk_t[0] = stamp_prev;      // (Q requested, scalars known)
    blk_t[1] = 0;
#endif
    bool first_seg = true;
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const int t0 = seg ? skip_hi : 0, t1 = seg ? ntiles : skip_lo;
        if (t0 >= t1) continue;
        const bool staged = pre && first_seg;      // tile t0 was requested by the previous block
        if (!staged && !(first_in_wg && first_seg)) __syncthreads();      // earlier readers of the LDS slots (previous segment / row block) are done
        if (DMA) {
            if (!staged) {
                stage_tile_dma<D, NW>(k_lds, kp, p.ld_qkv, krow0, t0 * BN, len, wave_u, lane);
                stage_tile_dma<D, NW>(v_lds, vp, p.ld_qkv, krow0, t0 * BN, len, wave_u, lane);
            }
            kdma.init(kp, p.ld_qkv, krow0, (t0 + 1) * BN, wave_u, lane);
            vdma.init(vp, p.ld_qkv, krow0, (t0 + 1) * BN, wave_u, lane);
            stage_tile_dma_wait();
        } else {
            kst.load_clamped(kp, p.ld_qkv, krow0, t0 * BN, len);
            vst.load_clamped(vp, p.ld_qkv, krow0, t0 * BN, len);
            kst.store(k_lds);
            vst.store(v_lds);
        }
        first_seg = false;
        __syncthreads();
#ifdef HALVA_STAMP
        if (blk_t[1] == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[1])::"memory");      // first tile in LDS
#endif
#pragma unroll 1
        for (int it = t0; it < t1; ++it) {
            const int kv0 = it * BN;
            const int slot = (it - t0) & 1;
            const char* kt = k_lds + slot * TILE_BYTES;
            const char* vt = v_lds + slot * TILE_BYTES;
            if (it + 1 < t1) {
                if (!DMA) {
                    kst.load_clamped(kp, p.ld_qkv, krow0, kv0 + BN, len);
                    vst.load_clamped(vp, p.ld_qkv, krow0, kv0 + BN, len);
                } else if (kv0 + 2 * BN <= len) {
                    kdma.issue_and_advance(k_lds + (slot ^ 1) * TILE_BYTES, p.ld_qkv, wave_u);
                    vdma.issue_and_advance(v_lds + (slot ^ 1) * TILE_BYTES, p.ld_qkv, wave_u);
                } else {      // the sequence's last, partial tile
                    stage_tile_dma<D, NW>(k_lds + (slot ^ 1) * TILE_BYTES, kp, p.ld_qkv, krow0, kv0 + BN, len, wave_u, lane);
                    stage_tile_dma<D, NW>(v_lds + (slot ^ 1) * TILE_BYTES, vp, p.ld_qkv, krow0, kv0 + BN, len, wave_u, lane);
                }
            }
            STAMP(0);
            const bool hidden = wave_in_b && kv0 >= br.a && kv0 < br.b;                       // tile wholly inside [a, b)
            const int len_t = (wave_in_b && kv0 < br.a && kv0 + BN > br.a) ? br.a : len;       // tile cut at a
            if ((!CAUSAL || kv0 <= wq_max) && !hidden) {
                if ((kv0 + BN > len_t) || (CAUSAL && kv0 + BN - 1 > wq_min))      // wave-uniform: boundary tiles only
                    fwd_tile<D, CAUSAL, true, SLOW_TR>(kt, vt, qf, oacc, m_ref, l_run, minit, sc, kv0, len_t, ql, lane STAMP_PASS);
                else
                    fwd_tile<D, CAUSAL, false, SLOW_TR>(kt, vt, qf, oacc, m_ref, l_run, minit, sc, kv0, len_t, ql, lane STAMP_PASS);
            }
            if (DMA) {
                stage_tile_dma_wait();
            } else if (it + 1 < t1) {
                kst.store(k_lds + (slot ^ 1) * TILE_BYTES);
                vst.store(v_lds + (slot ^ 1) * TILE_BYTES);
            }
            __syncthreads();
            STAMP(5);
        }
    }
#ifdef HALVA_STAMP
    if (p.dbg && lane == 0 && s == 0 && hd == 0 && qb == p.nblk - 1) {
        for (int i = 0; i < 6; ++i) p.dbg[wave * 8 + i] = stamp_acc[i];
        p.dbg[wave * 8 + 6] = ntiles;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[2])::"memory");      // tile loop done
#endif
    // every wave has passed the last tile's barrier: the ring is free and Q is dead - fetch the next block's operands in front of the stores
    prefetch_next();
    const float l_tot = xhalf_sum(l_run);
    const float inv = (q_valid && l_tot > 0.f) ? 1.f / l_tot : 0.f;
    if (q_in_T) {
        store_rows_T<D>(orow, oacc, inv, true, lane);
        if (h == 0 && p.lse) p.lse[((int64_t)s * p.H + hd) * p.T + gq] = q_valid ? (m_ref + log2f(l_tot)) * kLn2 : 0.f;
    }
#ifdef HALVA_STAMP
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[3])::"memory");      // rows stored
    if (p.dbg && lane == 0 && s == 0 && hd == 0)
        for (int i = 0; i < 4; ++i) p.dbg[2048 + (qb * 8 + wave) * 4 + i] = blk_t[i];
#endif
}

// Under the causal mask row block b needs (b+1) units of work; one workgroup takes blocks b and nblk-1-b so every
// workgroup does the same (nblk+1) units and the grid has no heavy tail.
template <int D, bool CAUSAL, bool SLOW_TR>
__global__ __launch_bounds__(512) void sdpa_fwd_kernel(const SdpaParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int s, hd, b;
    map_block(blockIdx.x, CAUSAL ? (p.nblk + 1) / 2 : p.nblk, p.H, p.npairs, false, s, hd, b);
    // the sequence's geometry is read once per workgroup
    const int start = p.seq_start ? p.seq_start[s] : 0;
    const int len = p.seq_len ? p.seq_len[s] : p.T;
    const Branch br = load_branch(p, s);
    s16x8 qf[D / 16];
    WG_CLOCK_BEGIN();
    if (CAUSAL) {
        int first, second;
        paired_blocks(p.nblk, start, br, b, first, second);
        if (second == first) second = -1;
        if (p.repair) {
            // sdpa_fwd3 fixes a row block's exponent reference on the row's first visible keys and repeats the block a bounded number of times when
            // later keys outgrow it (gen_fwd3_loop.py: MAX_REDO); a FINITE row whose maximum lies further out than that leaves it with l = inf, i.e.
            // lse = inf and NaN for output.  This kernel tracks a running maximum and has no such bound (as flash-attn): launched behind every
            // sdpa_fwd3 launch, a workgroup looks at the lse of its two blocks' valid rows and recomputes a block only if one of them is not finite
            // (NaN / inf inputs are recomputed to the same NaN / inf).  Ordinary activations: 2 x 256 loads and an exit.
            bool did = false;
            for (int k = 0; k < 2; ++k) {
                const int qb = k ? second : first;
                if (qb < 0) continue;
                const int gq = qb * 256 + (threadIdx.x & 255), ql = gq - start;
                const bool bad = threadIdx.x < 256 && gq < p.T && ql >= 0 && ql < len && !__builtin_isfinite(p.lse[((int64_t)s * p.H + hd) * p.T + gq]);
                if (!__syncthreads_or(bad)) continue;
                sdpa_fwd_block<D, CAUSAL, SLOW_TR>(p, smem, s, hd, qb, -1, false, !did, qf, start, len, br);
                did = true;
            }
            return;
        }
        sdpa_fwd_block<D, CAUSAL, SLOW_TR>(p, smem, s, hd, first, second, false, true, qf, start, len, br);
        if (second >= 0) sdpa_fwd_block<D, CAUSAL, SLOW_TR>(p, smem, s, hd, second, -1, true, false, qf, start, len, br);
    } else {
        sdpa_fwd_block<D, CAUSAL, SLOW_TR>(p, smem, s, hd, b, -1, false, true, qf, start, len, br);
    }
    WG_CLOCK_END(p.dbg, 0);
}

// ===================================================================================================
// backward, part 1: dQ  (same skeleton as the forward), and delta[s, h, t] = sum_d dO * O for both backward kernels
// ===================================================================================================
template <int D, bool CAUSAL, bool SLOW_TR, int NW>
__device__ __forceinline__ void sdpa_bwd_dq_block(const SdpaParams& p, char* smem, int s, int hd, int qb) {
    constexpr int BN = 64, KS = D / 16, DT = D / 32, BM = 32 * NW, NT = 64 * NW;
    constexpr int TILE_BYTES = BN * D * 2;
    char* k_lds = smem;
    char* v_lds = smem + 2 * TILE_BYTES;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5;
    const int start = p.seq_start ? p.seq_start[s] : 0;
    const int len = p.seq_len ? p.seq_len[s] : p.T;
    const int g0 = qb * BM;
    const int64_t seq_row0 = (int64_t)s * p.T;
    const int gq = g0 + 32 * wave + (lane & 31);
    const int ql = gq - start;
    const bool q_in_T = gq < p.T;
    const bool q_valid = q_in_T && ql >= 0 && ql < len;

    int kv_end = len;
    if (CAUSAL) kv_end = min(len, g0 + BM - start);
    const int ntiles = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;

    const bf16_t* qp = p.q + hd * D;
    const bf16_t* kp = p.k + hd * D;
    const bf16_t* vp = p.v + hd * D;
    bf16_t* dqrow = p.dq + (seq_row0 + gq) * p.ld_qkv + hd * D;
    if (ntiles == 0) {
        if (q_in_T) store_rows_zero<D>(dqrow, lane);
        return;
    }

    s16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (q_valid) {
            qf[ks] = *reinterpret_cast<const s16x8*>(qp + (seq_row0 + gq) * p.ld_qkv + 16 * ks + 8 * h);
            dof[ks] = *reinterpret_cast<const s16x8*>(p.d_o + (seq_row0 + gq) * p.ld_do + hd * D + 16 * ks + 8 * h);
        } else {
            qf[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
            dof[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    const int64_t stat = ((int64_t)s * p.H + hd) * p.T + gq;
    const float lse2 = q_valid ? p.lse[stat] * kLog2e : INFINITY;      // padded query rows: P = exp2(-inf) = 0
    // delta = rowsum(O o dO): this lane holds half of its row of dO already; the other half sits on lane ^ 32.  Written out for the
    // dK/dV kernel, which runs after this one (this used to be a launch of its own: 52 us of the backward's 1.2 ms).
    float dsum = 0.f;
    if (q_valid) {
        const bf16_t* orow = p.o_in + (seq_row0 + gq) * p.ld_o + hd * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 ov = *reinterpret_cast<const u32x4*>(orow + 16 * ks + 8 * h);
            const u32x4 dv = __builtin_bit_cast(u32x4, dof[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) dsum += bf16_lo(ov[i]) * bf16_lo(dv[i]) + bf16_hi(ov[i]) * bf16_hi(dv[i]);
        }
    }
    const float dlt = xhalf_sum(dsum);
    if (q_valid && h == 0) p.delta[stat] = dlt;
    const float sc = p.scale * kLog2e;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;

    const int wq_min = g0 + 32 * wave - start, wq_max = wq_min + 31;
    const Branch br = load_branch(p, s);
    const bool wave_in_b = wq_min >= br.b;             // wave-uniform (br.b is a multiple of 64, strips are 32 rows)
    // K/V tiles arrive by LDS-DMA (no staging registers, no ds_write pass); the slow-transpose debug build keeps register staging
    constexpr bool DMA = !SLOW_TR && NW == 8;
    Stage<D, BN, NT> kst, vst;
    TileDma<D, NW> kdma, vdma;
    const int64_t krow0 = seq_row0 + start;
    // key-tile range in (up to) two segments around the tiles a branch-B row block never needs (see the forward)
    int skip_lo = ntiles, skip_hi = ntiles;
    if (g0 - start >= br.b) {
        skip_lo = min(ntiles, (br.a + BN - 1) / BN);
        skip_hi = max(skip_lo, min(ntiles, br.b / BN));
    }
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const int t0 = seg ? skip_hi : 0, t1 = seg ? ntiles : skip_lo;
        if (t0 >= t1) continue;
        __syncthreads();      // earlier readers of the LDS slots (previous segment / previous row block) are done
        if (DMA) {
            stage_tile_dma<D, NW>(k_lds, kp, p.ld_qkv, krow0, t0 * BN, len, wave, lane);
            stage_tile_dma<D, NW>(v_lds, vp, p.ld_qkv, krow0, t0 * BN, len, wave, lane);
            kdma.init(kp, p.ld_qkv, krow0, (t0 + 1) * BN, wave, lane);
            vdma.init(vp, p.ld_qkv, krow0, (t0 + 1) * BN, wave, lane);
            stage_tile_dma_wait();
        } else {
            kst.load_clamped(kp, p.ld_qkv, krow0, t0 * BN, len);
            vst.load_clamped(vp, p.ld_qkv, krow0, t0 * BN, len);
            kst.store(k_lds);
            vst.store(v_lds);
        }
        __syncthreads();
#pragma unroll 1
        for (int it = t0; it < t1; ++it) {
        const int kv0 = it * BN;
        const int slot = (it - t0) & 1;
        const char* kt = k_lds + slot * TILE_BYTES;
        const char* vt = v_lds + slot * TILE_BYTES;
        if (it + 1 < t1) {
            if (!DMA) {
                kst.load_clamped(kp, p.ld_qkv, krow0, kv0 + BN, len);
                vst.load_clamped(vp, p.ld_qkv, krow0, kv0 + BN, len);
            } else if (kv0 + 2 * BN <= len) {
                kdma.issue_and_advance(k_lds + (slot ^ 1) * TILE_BYTES, p.ld_qkv, wave);
                vdma.issue_and_advance(v_lds + (slot ^ 1) * TILE_BYTES, p.ld_qkv, wave);
            } else {      // the sequence's last, partial tile
                stage_tile_dma<D, NW>(k_lds + (slot ^ 1) * TILE_BYTES, kp, p.ld_qkv, krow0, kv0 + BN, len, wave, lane);
                stage_tile_dma<D, NW>(v_lds + (slot ^ 1) * TILE_BYTES, vp, p.ld_qkv, krow0, kv0 + BN, len, wave, lane);
            }
        }
        const bool hidden = wave_in_b && kv0 >= br.a && kv0 < br.b;                       // tile wholly inside [a, b)
        const int len_t = (wave_in_b && kv0 < br.a && kv0 + BN > br.a) ? br.a : len;       // tile cut at a
        const bool active = (!CAUSAL || kv0 <= wq_max) && !hidden;
        if (active) {
            f32x16 st[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[t][r] = 0.f;
                    dp[t][r] = 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    st[t] = mfma32(frag_rows<D>(kt, 32 * t, ks, lane), qf[ks], st[t]);
                    dp[t] = mfma32(frag_rows<D>(vt, 32 * t, ks, lane), dof[ks], dp[t]);
                }
            }
            if ((kv0 + BN > len_t) || (CAUSAL && kv0 + BN - 1 > wq_min)) {      // wave-uniform: boundary tiles only
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = kv0 + 32 * t + acc_row(r, h);
                        if (kl >= len_t || (CAUSAL && kl > ql)) st[t][r] = -INFINITY;      // -> P = 0
                    }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][r], sc, -lse2));
                    st[t][r] = pr * (dp[t][r] - dlt);   // dZ^T
                }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const s16x8 zb = acc_to_frag(st[ks >> 1], ks & 1);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    dqacc[dt] = mfma32(frag_cols<D, SLOW_TR>(kt, 16 * ks, 32 * dt, lane), zb, dqacc[dt]);
            }
        }
        if (DMA) {
            stage_tile_dma_wait();
        } else if (it + 1 < t1) {
            kst.store(k_lds + (slot ^ 1) * TILE_BYTES);
            vst.store(v_lds + (slot ^ 1) * TILE_BYTES);
        }
        __syncthreads();
        }
    }
    if (q_in_T) store_rows_T<D>(dqrow, dqacc, q_valid ? p.scale : 0.f, true, lane);
}

template <int D, bool CAUSAL, bool SLOW_TR, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 2 : 1)) void sdpa_bwd_dq_kernel(const SdpaParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int s, hd, b;
    if (CAUSAL) {
        map_block(blockIdx.x, (p.nblk + 1) / 2, p.H, p.npairs, false, s, hd, b);
        sdpa_bwd_dq_block<D, CAUSAL, SLOW_TR, NW>(p, smem, s, hd, p.nblk - 1 - b);
        if (b != p.nblk - 1 - b) sdpa_bwd_dq_block<D, CAUSAL, SLOW_TR, NW>(p, smem, s, hd, b);
    } else {
        map_block(blockIdx.x, p.nblk, p.H, p.npairs, false, s, hd, b);
        sdpa_bwd_dq_block<D, CAUSAL, SLOW_TR, NW>(p, smem, s, hd, b);
    }
}

// ===================================================================================================
// backward, part 2: dK, dV, two-role form.  A workgroup owns 128 keys and streams 64-row tiles of Q and dO (one dual-use LDS image
// each) from the diagonal to the end of the sequence.  Keeping K, V, dK and dV of a 32-key strip in one wave needs 362 registers,
// i.e. one wave per SIMD; that form (round 1) was replaced by the split below and is gone.

// The two waves of a SIMD split that state: wave w (0..3, "V side") owns K fragments and dV of key strip w, wave w+4
// ("K side") owns V fragments and dK of the SAME strip, so both fit the 256-register budget of two waves per SIMD and one's
// MFMAs run beside the other's vector work.  Per 32-row sub-tile:
//   V side:  S = Q K^T  ->  P = exp2(S*sc - lse) (masks applied here)  ->  P to LDS (bf16, accumulator layout)  ->  dV^T += dO^T P
//   K side:  dP = dO V^T  ->  P from LDS  ->  dZ = P o (dP - delta)  ->  dK^T += Q^T dZ
// With a dS workspace (SdpaParams::ds_ws) the K side also stores dZ = dS - which it holds as packed bf16 anyway, as the B operand of
// dK^T += Q^T dZ - for the dQ kernel that follows (sdpa_bwd_dq2): four 1-KiB stores per step and wave, each lane's 16 bytes next to its
// neighbour's.  dS is then formed ONCE in the whole backward (5 matrix products per (query, key) tile pair instead of 7).
// The K side runs one staged step behind the V side, so the workgroup barrier of the step (needed for the Q/dO ring anyway) is the
// only synchronisation; the Q/dO ring has 3 slots (steps t-1, t and the one being fetched), P has 2 (by step parity).
// ---------------------------------------------------------------------------------------------------
// Geometry of one 128-key block for this lane (key = lane % 32 of strip `strip`).
struct DkvGeom {
    int gk, kl, kblk_min, q_begin, q_stop, ntiles;
    bool k_in_T, k_valid;
};
template <bool CAUSAL>
__device__ __forceinline__ DkvGeom dkv_geom(const SdpaParams& p, int kb, int strip, int start, int len, const Branch& br, int lane) {
    constexpr int BQ = 64;
    DkvGeom g;
    g.gk = kb * 128 + 32 * strip + (lane & 31);
    g.kl = g.gk - start;
    g.k_in_T = g.gk < p.T;
    g.k_valid = g.k_in_T && g.kl >= 0 && g.kl < len;
    g.kblk_min = kb * 128 - start;
    g.q_begin = 0;
    if (CAUSAL) g.q_begin = max(0, g.kblk_min) / BQ * BQ;
    const bool block_has_keys = (g.kblk_min < len) && (g.kblk_min + 128 > 0);
    g.q_stop = (g.kblk_min >= br.a && g.kblk_min + 127 < br.b) ? min(len, br.b) : len;
    g.ntiles = (block_has_keys && g.q_stop > g.q_begin) ? (g.q_stop - g.q_begin + BQ - 1) / BQ : 0;
    return g;
}

// `sf` (the stationary K or V fragments) and `st` (the row statistics in flight) belong to the caller so that a key block can fetch its
// SUCCESSOR's operands: once the step loop has passed its last barrier the Q/dO ring is free and the stationary registers are dead, so
// the next block's K/V rows, first Q/dO tile and first statistics are requested then - in front of this block's store tail - instead
// of in the next block's prologue (measured per block, s_memtime: 10 000-12 000 cycles from entry to the first step, of which three
// memory round trips one behind the other; a step is ~3 850).
//   kb_next: the key block this workgroup runs next (-1: none)      prefetched: this block's operands were requested by its predecessor
#ifndef HALVA_DKV_CINIT
#define HALVA_DKV_CINIT 1
#endif
// HALVA_DKV_CINIT (default): a row constant as the initial accumulator (cdna_hip_programming.md, attention backward), on the K side: the
// dP = dO V^T chain starts from -delta (exact: the same fp32 additions in another order), so dZ = P * dP' needs no subtraction - 32
// vector instructions fewer per K-side wave and step, and the 32 registers that held delta during the vector work are free after each
// chain's first MFMA.  The V side's analogue (S chain from -lse, K pre-multiplied by scale * log2 e so that P = exp2(S')) is NOT
// taken: like HALVA_FWD_CINIT it needs a second bf16 rounding of an operand, which moves P by up to ~1 % (see fwd_tile).
template <int D, bool CAUSAL, bool SLOW_TR, int ROLE>
__device__ __forceinline__ void sdpa_bwd_dkv2_block(const SdpaParams& p, char* smem, int s, int hd, int kb, int kb_next, bool prefetched,
                                                    int strip, int start, int len, const Branch br, s16x8 (&sf)[D / 16], float (&st)[2]) {
    constexpr int BQ = 64, SUB = 2, KS = D / 16, DT = D / 32;
    constexpr int TILE_BYTES = BQ * D * 2;
    char* q_lds = smem;                                    // [3][BQ][D]
    char* do_lds = smem + 3 * TILE_BYTES;                  // [3][BQ][D]
    float* lse_lds = reinterpret_cast<float*>(smem + 6 * TILE_BYTES);   // [3][BQ]
    float* dlt_lds = lse_lds + 3 * BQ;                                  // [3][BQ]
    char* p_lds = reinterpret_cast<char*>(dlt_lds + 3 * BQ);            // [2 parity][4 strips][SUB][64 lanes][2][16 B]

    const int lane = threadIdx.x & 63, h = lane >> 5;
#ifdef HALVA_STAMP
    unsigned long long blk_t[8];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[0])::"memory");
    blk_t[5] = blk_t[6] = blk_t[7] = blk_t[0];
#endif
    const int64_t seq_row0 = (int64_t)s * p.T;
    const DkvGeom g = dkv_geom<CAUSAL>(p, kb, strip, start, len, br, lane);
    const int gk = g.gk, kl = g.kl, kblk_min = g.kblk_min, q_begin = g.q_begin, ntiles = g.ntiles;
    const bool k_in_T = g.k_in_T, k_valid = g.k_valid;
    bf16_t* outrow = (ROLE ? p.dk : p.dv) + (seq_row0 + gk) * p.ld_qkv + hd * D;
    const bool key_hidden = kl >= br.a && kl < br.b;

    const bf16_t* qp = p.q + hd * D;
    const bf16_t* dop = p.d_o + hd * D;
    const int64_t qrow0 = seq_row0 + start;
    const float* lse_g = p.lse + ((int64_t)s * p.H + hd) * p.T + start;
    const float* dlt_g = p.delta + ((int64_t)s * p.H + hd) * p.T + start;
    float& st_lse = st[0];
    float& st_dlt = st[1];
    // the row statistics of a tile are fetched by ONE wave that requests no tiles (K side, strip 0): on a tile-requesting wave the wait
    // for these two loads in the prologue also waited for the stationary operand and held back the first tile request - a second
    // memory round trip in front of the loop
    const bool stats_wave = ROLE == 1 && strip == 0;
    auto load_stats = [&](int q0) {
        if (stats_wave) {
            const int ql = min(q0 + lane, len - 1);
            st_lse = lse_g[ql];
            st_dlt = dlt_g[ql];
        }
    };
    auto store_stats = [&](int buf) {
        if (stats_wave) {
            lse_lds[buf * BQ + lane] = st_lse * kLog2e;
#if HALVA_DKV_CINIT      // stored NEGATED: the values are the initial accumulator of the K side's dP chain (see below)
            dlt_lds[buf * BQ + lane] = -st_dlt;
#else
            dlt_lds[buf * BQ + lane] = st_dlt;
#endif
        }
    };
    // The four V-side waves fetch the tiles: they finish a step's arithmetic ahead of their K-side partners (measured against
    // all eight waves taking a share: -0.5 %).
    constexpr int NDMA = 4;
    constexpr bool dma_wave = ROLE == 0;
    const int dma_id = strip;
    // this wave's stationary operand: K fragments (V side) or V fragments (K side)
    auto request_stationary = [&](const DkvGeom& gg) {
        const bf16_t* stat = (ROLE ? p.v : p.k) + (seq_row0 + gg.gk) * p.ld_qkv + hd * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            sf[ks] = gg.k_valid ? *reinterpret_cast<const s16x8*>(stat + 16 * ks + 8 * h) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    };
    auto request_first_tile = [&](const DkvGeom& gg) {      // the caller vouches that nobody reads ring slot 0 any more
        if (dma_wave && gg.ntiles > 0) {
            stage_tile_dma<D, NDMA>(q_lds, qp, p.ld_qkv, qrow0, gg.q_begin, len, dma_id, lane);
            stage_tile_dma<D, NDMA>(do_lds, dop, p.ld_do, qrow0, gg.q_begin, len, dma_id, lane);
        }
    };
    auto prefetch_next = [&]() {
        if (kb_next >= 0) {
            const DkvGeom gn = dkv_geom<CAUSAL>(p, kb_next, strip, start, len, br, lane);
            if (gn.ntiles > 0) {
                request_stationary(gn);
                load_stats(gn.q_begin);
            }
            request_first_tile(gn);
        }
    };

    if (ntiles == 0) {
        if (k_in_T) store_rows_zero<D>(outrow, lane);
        prefetch_next();      // (no step loop ran here, and a predecessor - if any - has passed its last barrier: the rings are free)
        return;
    }
    if (!prefetched) {
        request_stationary(g);
        load_stats(q_begin);
    }
#ifdef HALVA_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[5])::"memory");      // scalars known, stationary loads issued
#endif
    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const float sc = p.scale * kLog2e;
    const int wk_min = kblk_min + 32 * strip;
    const bool wave_has_pad_keys = __any(!k_valid);

    if (!prefetched) request_first_tile(g);      // (not prefetched = first block of the workgroup: nobody is reading the rings)
    TileDma<D, NDMA> qdma, dodma;
    if (dma_wave) {
        qdma.init(qp, p.ld_qkv, qrow0, q_begin + BQ, dma_id, lane);      // stand on tile 1
        dodma.init(dop, p.ld_do, qrow0, q_begin + BQ, dma_id, lane);
    }
#ifdef HALVA_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[7])::"memory");      // first tile requested
#endif
    store_stats(0);
    stage_tile_dma_wait();
    __syncthreads();

#ifdef HALVA_STAMP
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
#ifdef HALVA_STAMP
    blk_t[4] = stamp_prev;            // loop start
#endif
    int slot = ROLE ? 2 : 0;          // ring slot of step t - ROLE (the K side's first pass, t = 0, is idle)
    int slot_next = 1;                // ring slot the fetch of step t + 1 goes to
#pragma unroll 1
    for (int t = 0; t <= ntiles; ++t) {
        STAMP(5);
        if (t + 1 < ntiles) {
            const int qn = q_begin + (t + 1) * BQ;
            load_stats(qn);
            // The V side requests the whole tile: its waves finish a step's arithmetic some 500 cycles before the K side's, and a request
            // costs its wave about 100 cycles wherever it is placed (tried: between the MFMAs of the first block - no cheaper).
            if (dma_wave) {
                if (qn + BQ <= len) {
                    qdma.issue_and_advance(q_lds + slot_next * TILE_BYTES, p.ld_qkv, dma_id);
                    dodma.issue_and_advance(do_lds + slot_next * TILE_BYTES, p.ld_do, dma_id);
                } else {      // the sequence's last, partial tile
                    stage_tile_dma<D, NDMA>(q_lds + slot_next * TILE_BYTES, qp, p.ld_qkv, qrow0, qn, len, dma_id, lane);
                    stage_tile_dma<D, NDMA>(do_lds + slot_next * TILE_BYTES, dop, p.ld_do, qrow0, qn, len, dma_id, lane);
                }
            }
        }
        STAMP(0);
        const int tt = t - ROLE;                           // the step this wave works on
        if (tt >= 0 && tt < ntiles) {
            const int qt0 = q_begin + tt * BQ;
            const char* qt = q_lds + slot * TILE_BYTES;
            const char* dot = do_lds + slot * TILE_BYTES;
            const float* lse_t = lse_lds + slot * BQ;
            const float* dlt_t = dlt_lds + slot * BQ;
            char* pt = p_lds + (((tt & 1) * 4 + strip) * SUB) * 2048 + lane * 32;
            const char* rows_tile = ROLE ? dot : qt;       // S = Q K^T  |  dP = dO V^T
            const char* cols_tile = ROLE ? qt : dot;       // dV^T += dO^T P  |  dK^T += Q^T dZ
            const bool q_in_b = qt0 >= br.b;               // br.b and qt0 are multiples of 64: uniform over the step
            const bool hidden = q_in_b && wk_min >= br.a && wk_min + 31 < br.b;
            // this (key block, query step, strip)'s 4 KiB of the dS workspace
            char* ds_step = (ROLE && p.ds_ws) ? p.ds_ws + ((((int64_t)s * p.H + hd) * p.ds_nkb + kb) * p.ds_nt + qt0 / BQ) * 16384 + strip * 4096
                                              : nullptr;
            // Both sub-tiles are always computed; a step that is not whole and unmasked (sequence tail, causal diagonal, pad keys, the
            // rows of branch B meeting keys of branch A) turns the affected scores into -inf on the V side, which makes P - and with
            // it dZ on the K side - exactly zero there.  A strip entirely hidden from this step's rows skips the step.
            const bool interior = (qt0 + BQ <= len) && (!CAUSAL || qt0 >= wk_min + 31) && !wave_has_pad_keys &&
                                  !(q_in_b && wk_min < br.b && wk_min + 31 >= br.a);      // wave-uniform
            if (!hidden) {
                // Every LDS read is placed by hand one block ahead of its use and nothing may cross a slot boundary: left to itself the
                // scheduler hoists all reads to the top of the step and the register allocator spills.
#define SLOT() __builtin_amdgcn_sched_barrier(0)
                constexpr bool CINIT = HALVA_DKV_CINIT && ROLE == 1;          // K side: the dP chain starts from -delta
                constexpr int PPS = 8 / KS, PPD = 8 / (2 * DT), NS = KS;      // NS slots per block (KS == 2 * DT)
                static_assert(KS == 2 * DT, "slot count");
                const float* stat_t = ROLE ? dlt_t : lse_t;
                auto pair = [&](const f32x16& x, int r, const f32x4& st, unsigned pw) -> unsigned {
                    const float t0 = st[r & 3], t1 = st[(r & 3) + 1];
                    (void)t0, (void)t1;
                    if (ROLE == 0)
                        return pack_bf16x2(__builtin_amdgcn_exp2f(__builtin_fmaf(x[r], sc, -t0)), __builtin_amdgcn_exp2f(__builtin_fmaf(x[r + 1], sc, -t1)));
                    if (CINIT) return pack_bf16x2(bf16_lo(pw) * x[r], bf16_hi(pw) * x[r + 1]);      // x = dP - delta already
                    return pack_bf16x2(bf16_lo(pw) * (x[r] - t0), bf16_hi(pw) * (x[r + 1] - t1));
                };
                auto mask_scores = [&](f32x16& x, int q0) {
                    const bool lane_off = !k_valid || (q_in_b && key_hidden);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ql = q0 + acc_row(r, h);
                        if (ql >= len || (CAUSAL && kl > ql) || lane_off) x[r] = -INFINITY;      // -> P = 0
                    }
                };
                auto col_frag = [&](int sub, int i) { return frag_cols<D, SLOW_TR>(cols_tile, 32 * sub + 16 * (i / DT), 32 * (i % DT), lane); };
                s16x8 fa[NS], fb[NS], fc[NS], fd[NS];
                f32x4 st0[4], st1[4];
                u32x4 p0[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, p1[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
                u32x4 w0[2], w1[2];
                f32x16 x0, x1;
                if (CINIT) {
                // ---- block 1: sub-tile 0's statistics ARE the first chain's initial accumulator (register r <-> query acc_row(r, h)): read
                //      them with the first row fragments; sub-tile 1's are fetched during the chain
#pragma unroll
                for (int j = 0; j < 4; ++j) st0[j] = *reinterpret_cast<const f32x4*>(stat_t + 8 * j + 4 * h);
#pragma unroll
                for (int ks = 0; ks < NS; ++ks) fa[ks] = frag_rows<D>(rows_tile, 0, ks, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) x0[r] = st0[r >> 2][r & 3];
                SLOT();
#pragma unroll
                for (int ks = 0; ks < NS; ++ks) {
                    x0 = mfma32(fa[ks], sf[ks], x0);
                    fb[ks] = frag_rows<D>(rows_tile, 32, ks, lane);
                    if (ks % (NS / 4) == 0) st1[ks / (NS / 4)] = *reinterpret_cast<const f32x4*>(stat_t + 32 + 8 * (ks / (NS / 4)) + 4 * h);
                    if (ROLE && ks == NS - 2) p0[0] = *reinterpret_cast<const u32x4*>(pt);
                    if (ROLE && ks == NS - 1) p0[1] = *reinterpret_cast<const u32x4*>(pt + 16);
                    SLOT();
                }
                } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) x0[r] = 0.f, x1[r] = 0.f;
                // ---- block 1: first product of sub-tile 0; fetch the second sub-tile's row fragments and sub-tile 0's statistics
#pragma unroll
                for (int ks = 0; ks < NS; ++ks) fa[ks] = frag_rows<D>(rows_tile, 0, ks, lane);
                SLOT();
#pragma unroll
                for (int ks = 0; ks < NS; ++ks) {
                    x0 = mfma32(fa[ks], sf[ks], x0);
                    fb[ks] = frag_rows<D>(rows_tile, 32, ks, lane);
                    if (ks % (NS / 4) == 0) st0[ks / (NS / 4)] = *reinterpret_cast<const f32x4*>(stat_t + 8 * (ks / (NS / 4)) + 4 * h);
                    if (ROLE && ks == NS - 2) p0[0] = *reinterpret_cast<const u32x4*>(pt);
                    if (ROLE && ks == NS - 1) p0[1] = *reinterpret_cast<const u32x4*>(pt + 16);
                    SLOT();
                }
                }
                if (ROLE == 0 && !interior) mask_scores(x0, qt0);
                STAMP(1);
                SLOT();
                // ---- block 2: first product of sub-tile 1 || vector work of sub-tile 0; fetch sub-tile 0's column fragments
                if (CINIT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x1[r] = st1[r >> 2][r & 3];
                }
#pragma unroll
                for (int ks = 0; ks < NS; ++ks) {
                    x1 = mfma32(fb[ks], sf[ks], x1);
                    fc[ks] = col_frag(0, ks);
#pragma unroll
                    for (int q = 0; q < PPS; ++q) {
                        const int i = ks * PPS + q;      // pair i: accumulator registers 2i, 2i+1
                        w0[i >> 2][i & 3] = pair(x0, 2 * i, st0[i >> 1], p0[i >> 2][i & 3]);
                    }
                    if (!CINIT && ks % (NS / 4) == NS / 4 - 1) st1[ks / (NS / 4)] = *reinterpret_cast<const f32x4*>(stat_t + 32 + 8 * (ks / (NS / 4)) + 4 * h);
                    if (ROLE && ks == NS / 2 - 1) p1[0] = *reinterpret_cast<const u32x4*>(pt + 2048);
                    if (ROLE && ks == NS - 1) p1[1] = *reinterpret_cast<const u32x4*>(pt + 2048 + 16);
                    SLOT();
                }
                if (!ROLE) {
                    *reinterpret_cast<u32x4*>(pt) = w0[0];
                    *reinterpret_cast<u32x4*>(pt + 16) = w0[1];
                    if (!interior) mask_scores(x1, qt0 + 32);
                } else if (ds_step) {      // nontemporal: written once, read once by another kernel - keep Q / dO / K / V in L2
                    __builtin_nontemporal_store(w0[0], reinterpret_cast<u32x4*>(ds_step + lane * 16));
                    __builtin_nontemporal_store(w0[1], reinterpret_cast<u32x4*>(ds_step + 1024 + lane * 16));
                }
                STAMP(2);
                SLOT();
                // ---- block 3: second product of sub-tile 0 || vector work of sub-tile 1; fetch sub-tile 1's column fragments
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    acc[i % DT] = mfma32(fc[i], __builtin_bit_cast(s16x8, w0[i / DT]), acc[i % DT]);
                    fd[i] = col_frag(1, i);
#pragma unroll
                    for (int q = 0; q < PPD; ++q) {
                        const int j = i * PPD + q;
                        w1[j >> 2][j & 3] = pair(x1, 2 * j, st1[j >> 1], p1[j >> 2][j & 3]);
                    }
                    SLOT();
                }
                if (!ROLE) {
                    *reinterpret_cast<u32x4*>(pt + 2048) = w1[0];
                    *reinterpret_cast<u32x4*>(pt + 2048 + 16) = w1[1];
                } else if (ds_step) {
                    __builtin_nontemporal_store(w1[0], reinterpret_cast<u32x4*>(ds_step + 2048 + lane * 16));
                    __builtin_nontemporal_store(w1[1], reinterpret_cast<u32x4*>(ds_step + 3072 + lane * 16));
                }
                // ---- block 4: second product of sub-tile 1
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[i % DT] = mfma32(fd[i], __builtin_bit_cast(s16x8, w1[i / DT]), acc[i % DT]);
#undef SLOT
            }
        }
#ifdef HALVA_STAMP
        asm volatile("" : "+v"(acc[0][15]), "+v"(acc[DT - 1][15]));
#endif
        STAMP(3);
        if (t + 1 < ntiles) store_stats(slot_next);
        if (dma_wave) stage_tile_dma_wait();      // (the K side has only its dS stores in flight: nothing of this step waits for them)
        STAMP(4);
        slot = (slot == 2) ? 0 : slot + 1;
        slot_next = (slot_next == 2) ? 0 : slot_next + 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // not __syncthreads(): its fence would drain the dS stores
    }
#ifdef HALVA_STAMP
    if (p.dbg && lane == 0 && kb == 0 && s == 0 && hd < 4) {
        const int wave = strip + 4 * ROLE;
        for (int i = 0; i < 6; ++i) p.dbg[(hd * 8 + wave) * 8 + i] = stamp_acc[i];
        p.dbg[(hd * 8 + wave) * 8 + 6] = ntiles;
    }
    blk_t[1] = stamp_prev;      // (the loop's last stamp)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[2])::"memory");
#endif
    // every wave has passed the last step's barrier: the rings are free and the stationary fragments dead
    prefetch_next();
    if (k_in_T) store_rows_T<D>(outrow, acc, k_valid ? (ROLE ? p.scale : 1.f) : 0.f, true, lane);
#ifdef HALVA_STAMP
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t[3])::"memory");
    if (p.dbg && lane == 0 && s == 0 && hd == 0) {      // whole-block anatomy: [entry, loop end, before the stores, after the stores]
        const int wave = strip + 4 * ROLE;
        for (int i = 0; i < 8; ++i) p.dbg[1024 + (kb * 8 + wave) * 8 + i] = blk_t[i];
    }
#endif
}

template <int D, bool CAUSAL, bool SLOW_TR, int ROLE>
__device__ __forceinline__ void sdpa_bwd_dkv2_role(const SdpaParams& p, char* smem, int strip) {
    int s, hd, b;
    map_block(blockIdx.x, CAUSAL ? (p.nblk + 1) / 2 : p.nblk, p.H, p.npairs, false, s, hd, b);
    // the sequence's geometry is read ONCE per workgroup (both key blocks belong to the same sequence): in a block's prologue these
    // scalar loads are a memory round trip of their own in front of everything else
    const int start = p.seq_start ? p.seq_start[s] : 0;
    const int len = p.seq_len ? p.seq_len[s] : p.T;
    const Branch br = load_branch(p, s);
    s16x8 sf[D / 16];
    float st[2] = {0.f, 0.f};
    // under the causal mask key block b is visited by (nblk - b) query blocks: pair b with nblk-1-b
    const int second = (CAUSAL && b != p.nblk - 1 - b) ? p.nblk - 1 - b : -1;
    sdpa_bwd_dkv2_block<D, CAUSAL, SLOW_TR, ROLE>(p, smem, s, hd, b, second, false, strip, start, len, br, sf, st);
    if (second >= 0) sdpa_bwd_dkv2_block<D, CAUSAL, SLOW_TR, ROLE>(p, smem, s, hd, second, -1, true, strip, start, len, br, sf, st);
}

template <int D, bool CAUSAL, bool SLOW_TR>
__global__ __launch_bounds__(512) void sdpa_bwd_dkv2_kernel(const SdpaParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the role is a per-wave constant: branch on it once, on the scalar unit, so that each side gets its own register allocation
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WG_CLOCK_BEGIN();
    if (wave < 4) sdpa_bwd_dkv2_role<D, CAUSAL, SLOW_TR, 0>(p, smem, wave);
    else sdpa_bwd_dkv2_role<D, CAUSAL, SLOW_TR, 1>(p, smem, wave - 4);      // (raising these waves' s_setprio changes nothing)
    WG_CLOCK_END(p.dbg, 1);
}

// ===================================================================================================
// backward with a dS workspace: delta / zero-fill pass, then dK/dV (+ dS store), then dQ = dS K
// ===================================================================================================
#include "sdpa_dkv3_items.h"
// delta[s, h, t] = sum_d dO o O for every valid query row (both later kernels read it), and zeros into dq of the PADDED rows (the dQ
// kernel below walks a sequence in its own coordinates and only writes its valid rows).  One wave per (row, 4 heads); HBM-bound.
// for_dkv3: also what sdpa_bwd_dkv3 needs before it starts - its work-queue counters zeroed and (round 6) its item records (sdpa_dkv3_items.h).
template <int D>
__global__ __launch_bounds__(256) void sdpa_bwd_delta_kernel(const SdpaParams p, int S, int for_dkv3) {
    constexpr int HPW = 512 / D;                       // heads per wave pass: 64 lanes x 8 elements
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (for_dkv3 && blockIdx.x == 0 && threadIdx.x < 8) p.sched[32 * threadIdx.x] = 0;
    if (for_dkv3) {      // one thread per record (8 192 - 13 824 of them at the step's shapes: the first few dozen workgroups)
        const int total = p.npairs * p.nblk;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= total; i += (int64_t)gridDim.x * 256)
            dkv3_build_record<true>(p, (int)i, total, p.items + i * DKV3_REC_DWORDS);
    }
    if (row >= (int64_t)S * p.T) return;
    const int s = (int)(row / p.T), t = (int)(row % p.T);
    const int start = p.seq_start ? p.seq_start[s] : 0;
    const int len = p.seq_len ? p.seq_len[s] : p.T;
    const bool valid = t >= start && t < start + len;
    const int sub = lane / (D / 8), e = (lane % (D / 8)) * 8;      // head within the pass, first element
    for (int h0 = 0; h0 < p.H; h0 += HPW) {
        const int hd = h0 + sub;
        if (hd >= p.H) continue;
        if (!valid) {
            *reinterpret_cast<u32x4*>(p.dq + row * p.ld_qkv + hd * D + e) = u32x4{0u, 0u, 0u, 0u};
            continue;
        }
        // (out is read here for the last time; dO once more, tile by tile, by the dK/dV kernel - from HBM either way: 0.27 - 0.45 GB per tensor)
        const u32x4 ov = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.o_in + row * p.ld_o + hd * D + e));
        const u32x4 dv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.d_o + row * p.ld_do + hd * D + e));
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += bf16_lo(ov[i]) * bf16_lo(dv[i]) + bf16_hi(ov[i]) * bf16_hi(dv[i]);
#pragma unroll
        for (int o = D / 16; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane % (D / 8) == 0) {
            const int64_t at = ((int64_t)s * p.H + hd) * p.T + t;
            if (for_dkv3) {      // sdpa_bwd_dkv3 starts its dP chain from -delta and fetches both statistics of a 64-row step with ONE request (sdpa_dkv3.h)
                const int loc = t - start;
                float* rec = p.lse2 + (((int64_t)s * p.H + hd) * p.stat_nt + (loc >> 6)) * 128;
                rec[loc & 63] = p.lse[at] * kLog2e;
                rec[64 + (loc & 63)] = -acc;
                if (loc == len - 1)      // the rest of the sequence's last record: zeros (a padded query row then yields P = 1, dS = 0 - finite)
                    for (int j = (loc & 63) + 1; j < 64; ++j) rec[j] = 0.f, rec[64 + j] = 0.f;
            } else {
                p.delta[at] = acc;
            }
        }
    }
}

// dQ = scale * dS K with dS read back from the workspace the dK/dV kernel filled.  A workgroup takes 256 query rows of one (sequence,
// head) in SEQUENCE coordinates (row 0 = seq_start: the grid of query steps the producer used), a wave 32 of them = one 32-row half of
// a producer step; per 64-key tile (GLOBAL key coordinates = the producer's 128-key blocks) a wave multiplies
//     dQ^T[128 x 32] += K^T[128 x 64] dS^T[64 x 32]          (16 MFMAs: A = K^T by transposed reads of the shared K tile,
//                                                             B = dS^T by transposed reads of the wave's own 4 KiB of dS)
// The producer's layout is, per (key block, step, 32-key strip, 32-row half, register half j): 64 lanes x 16 bytes, lane (key n, h')
// holding the query rows 16 j + 8 g + 4 h' + (0..3), g = 0, 1: the 8-byte unit (one key, four consecutive queries) is exactly what
// ds_read_b64_tr_b16 gathers from, so the image is copied to LDS byte for byte (LDS-DMA) and transposed by the read.
// HBM-bound: 2 bytes per (query, key) pair, the same pairs the producer wrote.
__device__ __forceinline__ int ds_piece_off(int key, int qgroup) {      // byte offset of (key 0..31, queries 4 G .. 4 G + 3) in a strip's 2 KiB
    return 1024 * (qgroup >> 2) + 16 * (key + 32 * (qgroup & 1)) + 8 * ((qgroup >> 1) & 1);
}
// The LDS copy (round 4): the same image with its 1-KiB pieces 1152 bytes apart (strips 2304, ring slots 4608).  One transposed read has, in
// each half of the wave, its lanes on byte pairs that differ in G >> 2 (the piece) and G & 1 (512 bytes apart inside the piece) only: 1 KiB and
// 512 bytes are multiples of the 256-byte bank row - the same banks FOUR times (SQ_LDS_BANK_CONFLICT = 37 % of this kernel's LDS cycles,
// profiles/r04_sdpa_all_pmc.json).  The 128 bytes of padding move the odd pieces half a bank row: two-way.  Measured and NOT kept: the last
// factor of two by an XOR of the chunk position, applied by the LDS-DMA lanes (lane l fetches chunk l ^ swizzle; LDS-DMA writes lane l to
// byte 16 l) - zero conflicts, but the requests no longer ask for their 1 KiB in lane order and the kernel, which is HBM-bound, ran 1.6 % SLOWER
// (950-959 -> 971-973 us at the step's shapes, same box).
#ifndef DQ2_FAST_TILE
#define DQ2_FAST_TILE 1      // (0: every tile through the general per-strip path, as before round 5)
#endif
#ifndef HALVA_DQ2_DS_POLICY
#define HALVA_DQ2_DS_POLICY " nt"      // (experiments/ds_residency builds it with "" as well)
#endif
constexpr int DS_LDS_PIECE = 1024 + 128, DS_LDS_STRIP = 2 * DS_LDS_PIECE, DS_LDS_SLOT = 2 * DS_LDS_STRIP;
__device__ __forceinline__ int ds_lds_off(int key, int qgroup) {      // (inside a strip)
    return DS_LDS_PIECE * (qgroup >> 2) + 16 * (key + 32 * (qgroup & 1)) + 8 * ((qgroup >> 1) & 1);
}
template <int D, bool SLOW_TR, bool FAST>
__device__ __forceinline__ void sdpa_bwd_dq2_block(const SdpaParams& p, char* smem, int s, int hd, int qb, int wave, int lane) {
    constexpr int NW = 8, BN = 64, DT = D / 32, BM = 32 * NW, RING = 3;
    constexpr int TILE_BYTES = BN * D * 2;
    char* k_lds = smem;                                            // [RING][BN][D]
    char* ds_lds = smem + RING * TILE_BYTES + wave * (RING * DS_LDS_SLOT);  // per wave: [RING][2 strips][2 pieces of 1 KiB + 128 B]
    const int h = lane >> 5;
    const int start = p.seq_start ? p.seq_start[s] : 0;
    const int len = p.seq_len ? p.seq_len[s] : p.T;
    const int64_t seq_row0 = (int64_t)s * p.T;
    const int lq0 = qb * BM;                             // first local row of the block
    if (lq0 >= len) return;                              // workgroup-uniform
    const int wr0 = lq0 + 32 * wave;                     // this wave's first local row
    const int lq = wr0 + (lane & 31);
    const bool q_valid = lq < len;
    const bool wave_live = wr0 < len;
    const Branch br = load_branch(p, s);
    const bool wave_in_b = wr0 >= br.b;
    // global key tiles that hold a key some row of the block may see: local key index kv0g - start <= last row of the block
    const int first_tile = max(0, start) / BN;
    const int last_key_local = min(len, lq0 + BM) - 1;                 // causal: keys <= row
    const int ntile_end = (start + last_key_local) / BN + 1;           // exclusive, global tile index
    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const bf16_t* kp = p.k + hd * D;
    const int64_t krow0 = seq_row0 + start;
    const int step = wr0 / 64, sub = (wr0 / 32) & 1;
    const char* ds_pair = p.ds_ws + ((int64_t)s * p.H + hd) * p.ds_nkb * p.ds_nt * 16384;
    const unsigned ds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ds_lds;
    // a tile is skipped by this wave when none of its rows sees any of its keys (tiles are walked upwards: once skipped, always skipped);
    // a 32-key strip when the branch mask hides it wholly (the producer then wrote nothing for it: its bytes are fetched but not used)
    auto tile_live = [&](int kt) { return wave_live && (kt * BN - start) <= wr0 + 31; };
    auto strip_hidden = [&](int kt, int si) {
        const int k0 = kt * BN + 32 * si - start;
        return wave_in_b && k0 >= br.a && k0 + 31 < br.b;
    };
    // requests of one tile by this wave: its 2 chunks of the shared K tile, and - while the tile is live for it - its own 4 KiB of dS
    // (-DHALVA_DQ2_DIAG=<bits>, timing experiments only - results are wrong: 1 no matrix work, 2 no dS requests, 4 no K requests, 8 no barriers)
#ifndef HALVA_DQ2_DIAG
#define HALVA_DQ2_DIAG 0
#endif
    auto stage = [&](int kt, int slot) {
        if (!(HALVA_DQ2_DIAG & 4)) stage_tile_dma<D, NW>(k_lds + slot * TILE_BYTES, kp, p.ld_qkv, krow0, kt * BN - start, len, wave, lane);
        if (!(HALVA_DQ2_DIAG & 2) && tile_live(kt)) {
            const char* src = ds_pair + ((int64_t)(kt >> 1) * p.ds_nt + step) * 16384 + 2 * (kt & 1) * 4096 + sub * 2048;
#pragma unroll
            for (int c = 0; c < 4; ++c) {                 // strip c >> 1, register half c & 1
                const unsigned dst = ds_dst + slot * DS_LDS_SLOT + (c >> 1) * DS_LDS_STRIP + (c & 1) * DS_LDS_PIECE;
                const unsigned voff = lane * 16;
                const char* rows = src + (c >> 1) * 4096 + (c & 1) * 1024;
                unsigned keep;
                // nt: these bytes are read once, by this CU only - they must not evict the K tiles the XCD's workgroups share in L2
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" HALVA_DQ2_DS_POLICY "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rows) : "memory");
            }
        }
    };
    // per-lane byte offsets of the transposed reads of dS^T (see frag_cols for the lane roles): key 8 jj + 4 hb + q4, query group 4 (g & 1) + pp
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, hb = g >> 1;
    const int ds_rd0 = ds_lds_off(4 * hb + q4, 4 * (g & 1) + pp);           // jj = 0; jj = 1 adds 8 keys = 128 bytes
    // frag_cols' addresses in the K tile, as two per-lane bases (jj = 0, 1: rows 8 jj + 4 hb + q4 of a 16-key group) + 4096 ks + 512 dt (tile_off)
    const unsigned k_u32 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)k_lds;
    unsigned klane[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
        klane[jj] = 2048 * jj + 64 * (4 * hb + q4) + 16 * ((2 * (g & 1) + (pp >> 1)) ^ ((2 * jj + hb) & 3)) + 8 * (pp & 1);
    // A row block wholly in branch B never needs the key tiles that lie wholly inside [a, b) (the producer wrote no dS for them either):
    // the walk jumps from tile skip_lo - 1 to tile skip_hi, as the forward's does.  In the bench's packed rows [668 | 1380 | 1380] that
    // is 21 of the 33..54 tiles of each of the six B blocks - 30 % of this kernel's tile steps, each a 16-KiB K tile and up to 32 KiB
    // of dS fetched for nothing.  (Branch points come with start == 0: halva_amd/splice.py packs right-padded rows only.)
    int skip_lo = ntile_end, skip_hi = ntile_end;
#ifndef HALVA_DQ2_NO_SKIP      // (A/B switch: -DHALVA_DQ2_NO_SKIP walks every tile as round 2 did)
    if (start == 0 && lq0 >= br.b)
#else
    if (false)
#endif
    {
        skip_lo = min(ntile_end, max(first_tile, (br.a + BN - 1) / BN));
        skip_hi = max(skip_lo, min(ntile_end, br.b / BN));
    }
    const int n_lo = skip_lo - first_tile, n_walk = n_lo + (ntile_end - skip_hi);
    auto tile_at = [&](int i) { return i < n_lo ? first_tile + i : skip_hi + (i - n_lo); };
    // TWO tiles in flight (HBM-bound kernel: the queue must not run dry while a tile is multiplied): ring of three slots, counted waits
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the previous row block's readers are done
    if (n_walk > 0) stage(tile_at(0), 0);
    if (n_walk > 1) stage(tile_at(1), 1);
#pragma unroll 1
    for (int i = 0; i < n_walk; ++i) {
        const int kt = tile_at(i);
        const int slot = i % RING;
        // tile i has landed: everything but the requests of tile i + 1 (2 pieces, 6 while that tile is live for this wave)
        if (i + 1 < n_walk) {
            if (HALVA_DQ2_DIAG & 6) {      // (diagnostic builds: fewer requests per tile)
                const bool live = tile_live(tile_at(i + 1));
                if ((HALVA_DQ2_DIAG & 6) == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (HALVA_DQ2_DIAG & 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (live) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (tile_live(tile_at(i + 1))) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (HALVA_DQ2_DIAG & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... for every wave; and tile i - 1 has been read by all
        if (i + 2 < n_walk) stage(tile_at(i + 2), (slot + 2) % RING);         // into the slot of tile i - 1
        if (!(HALVA_DQ2_DIAG & 1) && tile_live(kt)) {
            const char* ktile = k_lds + slot * TILE_BYTES;
            const char* dst_t = ds_lds + slot * DS_LDS_SLOT;
            if (DQ2_FAST_TILE && FAST && !SLOW_TR) {
                // All 40 operand reads of the tile from three per-lane bases + immediates, asked for ahead of the 16 products (the same products in the
                // same order as the general path below: identical sums).  Compiled from the loop below, every product waited for operand reads issued just
                // in front of it (s_waitcnt lgkmcnt(0) x 16 per tile) and each read cost two vector instructions of address arithmetic: without any dS
                // traffic at all the kernel still took 0.82 of its time (experiments/ds_residency: dq2 anatomy by removal, profiles/r05_dq2_anatomy.log).
                const unsigned a0 = k_u32 + slot * TILE_BYTES + klane[0], a1 = k_u32 + slot * TILE_BYTES + klane[1];
                const unsigned da = ds_dst + slot * DS_LDS_SLOT + ds_rd0;
                auto rd = [](unsigned addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr); };
                s16x4 zt[4][2], kt4[4][DT][2];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    zt[ks][0] = rd(da + (ks >> 1) * DS_LDS_STRIP + 256 * (ks & 1));
                    zt[ks][1] = rd(da + (ks >> 1) * DS_LDS_STRIP + 256 * (ks & 1) + 128);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        kt4[ks][dt][0] = rd(a0 + 4096 * ks + 512 * dt);
                        kt4[ks][dt][1] = rd(a1 + 4096 * ks + 512 * dt);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s16x8 zb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) zb[j] = zt[ks][0][j], zb[4 + j] = zt[ks][1][j];
                    // a strip the branch mask hides wholly was never written by the producer: its bytes are whatever the workspace held - zeros instead
                    // (the general path skips its products; adding exact zeros leaves the sums as they are).  ONE path for every tile: with a second one
                    // the accumulators changed registers between the two (32 moves per tile).
                    // Branch-free (an AND with a scalar mask): the tile stays one basic block and the order asked for below holds.
                    {
                        u32x4 w = __builtin_bit_cast(u32x4, zb);
                        const unsigned keep = strip_hidden(kt, ks >> 1) ? 0u : 0xffffffffu;
                        w[0] &= keep, w[1] &= keep, w[2] &= keep, w[3] &= keep;
                        zb = __builtin_bit_cast(s16x8, w);
                    }
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        s16x8 ka;
#pragma unroll
                        for (int j = 0; j < 4; ++j) ka[j] = kt4[ks][dt][0][j], ka[4 + j] = kt4[ks][dt][1][j];
                        acc[dt] = mfma32(ka, zb, acc[dt]);
                    }
                }
                // the order asked of the scheduler: ten reads (the first product's operands and the next one's), then a product per two reads
                __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            } else
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {               // 16 keys each: strip ks >> 1, half ks & 1
                if (strip_hidden(kt, ks >> 1)) continue;
                s16x8 zb;
                if (SLOW_TR) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3), qq = lane & 31;
                        zb[j] = *reinterpret_cast<const short*>(dst_t + (ks >> 1) * DS_LDS_STRIP + ds_lds_off(key, qq >> 2) + (qq & 3) * 2);
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const char* a = dst_t + (ks >> 1) * DS_LDS_STRIP + ds_rd0 + 16 * (16 * (ks & 1) + 8 * jj);
                        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
                        zb[4 * jj + 0] = t[0];
                        zb[4 * jj + 1] = t[1];
                        zb[4 * jj + 2] = t[2];
                        zb[4 * jj + 3] = t[3];
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma32(frag_cols<D, SLOW_TR>(ktile, 16 * ks, 32 * dt, lane), zb, acc[dt]);
            }
        }
    }
    bf16_t* dq_row = p.dq + (seq_row0 + start + lq) * p.ld_qkv + hd * D;
    if constexpr (D == 128) {
        if (p.rope_cos) {      // (workgroup-uniform) positions: halva_rope_qk's convention; the wave's 32 rows sit at consecutive positions
            if (wave_live) {   // (wave-uniform)
                char* scratch = ds_lds;      // the wave's own dS ring: its last tile has been read (the loop's MFMAs consumed it), nobody else touches it
                rope_rows_to_lds(scratch, p.rope_cos, p.rope_sin, rope_position(start + wr0, br), p.rope_max_pos, lane);
                store_rows_T_rope<D>(dq_row, acc, p.scale, q_valid, lane, scratch);
            }
            return;
        }
    }
    if (q_valid) store_rows_T<D>(dq_row, acc, p.scale, true, lane);
}

template <int D, bool SLOW_TR, bool FAST = true>      // (FAST = false: every tile through the general per-strip loop - HALVA_DQ2_FAST_TILE=0, the bitwise twin of the fast tile)
__global__ __launch_bounds__(512) void sdpa_bwd_dq2_kernel(const SdpaParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int s, hd, b;
    map_block(blockIdx.x, (p.nblk + 1) / 2, p.H, p.npairs, false, s, hd, b);
    int heavy, light;      // (the same pairing as the forward: the blocks' work is the same count of key tiles)
    paired_blocks(p.nblk, p.seq_start ? p.seq_start[s] : 0, load_branch(p, s), b, heavy, light);
    const int npass = (heavy != light) ? 2 : 1;
    WG_CLOCK_BEGIN();
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) sdpa_bwd_dq2_block<D, SLOW_TR, FAST>(p, smem, s, hd, pass ? light : heavy, wave, lane);
    WG_CLOCK_END(p.dbg, 2);
}

#include "sdpa_dkv3.h"
#include "sdpa_fwd3.h"
#include "sdpa_fwd3_twin.h"

// HALVA_SDPA_DKV3=0: the two-role dK/dV kernel of rounds 1-2 instead of sdpa_bwd_dkv3; HALVA_DKV3_ASM=0: sdpa_bwd_dkv3 with every step in
// plain HIP (the generated loop off) - A/B and debugging switches, read on every call like the one below.
bool env_flag_on(const char* name) {
    const char* e = getenv(name);
    return !(e && e[0] == '0');
}

// HALVA_SDPA_SLOW_TR=1 (tests): scalar transposed reads instead of ds_read_b64_tr_b16.  Read on every call - a test flips it inside
// one process - but it is a plain getenv, no allocation; everything else launch-related is cached per kernel below.
bool slow_tr_requested() {
    const char* e = getenv("HALVA_SDPA_SLOW_TR");
    return e && e[0] == '1';
}

template <typename KernelT>
int launch_one(KernelT kern, SdpaParams p, bool causal, int rows_per_block, int threads, size_t lds, int S, hipStream_t st,
               const char* name) {
    p.nblk = (p.T + rows_per_block - 1) / rows_per_block;
    p.npairs = S * p.H;
    const int wg_per_pair = causal ? (p.nblk + 1) / 2 : p.nblk;
    // The dynamic-LDS limit is a per-(kernel, device) attribute: remembered per device in atomics (one table per template instantiation
    // = per kernel), set once and again only to grow.  Two host threads racing here both set the same value - harmless; a process
    // that drives several devices sets it on each.
    constexpr int kMaxDev = 64;
    static std::atomic<size_t> lds_set[kMaxDev];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int slot = (dev >= 0 && dev < kMaxDev) ? dev : -1;
    if (slot < 0 || lds > lds_set[slot].load(std::memory_order_acquire)) {
        const hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) {
            halva_set_error("%s: hipFuncSetAttribute(%zu B of LDS) failed: %s", name, lds, hipGetErrorString(ea));
            return HALVA_ERR_LAUNCH;
        }
        if (slot >= 0) {
            size_t cur = lds_set[slot].load(std::memory_order_relaxed);
            while (cur < lds && !lds_set[slot].compare_exchange_weak(cur, lds, std::memory_order_release)) {
            }
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(wg_per_pair * p.npairs)), dim3(threads), lds, st, p);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) {
        halva_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));
        return HALVA_ERR_LAUNCH;
    }
    return HALVA_OK;
}

// sdpa_fwd3: 4 waves, one per SIMD (512 registers), 256 query rows per workgroup, 129 KiB of LDS (four K / V tile slots); persistent
// workgroups, one per CU, over the virtual blocks of the static launch (sdpa_fwd3.h)
int launch_fwd3(SdpaParams p, int S, hipStream_t st) {
    p.nblk = (p.T + 255) / 256;
    p.npairs = S * p.H;
    static std::atomic<int> attr_set[64];
    static std::atomic<int> n_cu[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool slot = dev >= 0 && dev < 64;
    if (!slot || !attr_set[dev].load(std::memory_order_acquire)) {
        const hipError_t ea = hipFuncSetAttribute((const void*)sdpa_fwd3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FWD3_LDS);
        if (ea != hipSuccess) {
            halva_set_error("sdpa_fwd3: hipFuncSetAttribute(%d B of LDS) failed: %s", FWD3_LDS, hipGetErrorString(ea));
            return HALVA_ERR_LAUNCH;
        }
        if (slot) attr_set[dev].store(1, std::memory_order_release);
    }
    int cus = slot ? n_cu[dev].load(std::memory_order_acquire) : 0;
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (slot) n_cu[dev].store(cus, std::memory_order_release);
    }
    {
        const char* e = getenv("HALVA_FWD3_GRID");      // diagnostic: workgroups of the persistent launch (0 = one per virtual block: no pipelining across items)
        if (e) cus = atoi(e) > 0 ? atoi(e) : (1 << 30);
    }
    {
        const char* e = getenv("HALVA_FWD3_ASM");      // 0: the plain-HIP twin of the generated block (sdpa_fwd3_twin.h; one workgroup per row block)
        if (e && e[0] == '0') {
            hipLaunchKernelGGL((sdpa_fwd3_twin_kernel<true>), dim3((unsigned)((int64_t)p.nblk * p.npairs)), dim3(256), 0, st, p);
            HALVA_CHECK_LAUNCH("sdpa_fwd3_twin");
            return HALVA_OK;
        }
    }
    const int64_t total = (int64_t)((p.nblk + 1) / 2) * p.npairs;
    hipLaunchKernelGGL((sdpa_fwd3_kernel<true>), dim3((unsigned)std::min<int64_t>(total, cus)), dim3(256), FWD3_LDS, st, p);
    HALVA_CHECK_LAUNCH("sdpa_fwd3");
    return HALVA_OK;
}

template <int D, bool CAUSAL>
int launch_fwd(const SdpaParams& p_in, int S, hipStream_t st) {
    const size_t lds = 4 * 64 * D * 2;
    SdpaParams p = p_in;
#ifdef HALVA_STAMP
    p.dbg = halva_dbg_buffer();
#endif
    if constexpr (D == 128 && CAUSAL) {
        // sdpa_fwd3: one wave per SIMD, the tile loop in generated asm (sdpa_fwd3.h).  HALVA_SDPA_FWD3=0: the two-waves-per-SIMD kernel of
        // rounds 1-3 (A/B and debugging switch, read on every call).  (Its tile counts travel as 16-bit fields, a sequence's K / V rows are
        // addressed through one 32-bit buffer descriptor.)
        if (!slow_tr_requested() && p.T < (1 << 21) && (int64_t)p.T * p.ld_qkv * 2 < (1ll << 31) && env_flag_on("HALVA_SDPA_FWD3")) {
            const int rc = launch_fwd3(p, S, st);
            // HALVA_FWD3_REPAIR=1: the unbounded-range repair pass over the row blocks sdpa_fwd3 gave up on (sdpa_fwd_kernel, p.repair).  Off by
            // default: measured 32 us per launch (fwd_in_step 0.777 -> 0.809 ms, profiles/r05_ab_rope_repair.log) against rows that would need scores
            // 5 300 nats above their first keys (sdpa_fwd3.h); such rows come back as NaN, not as wrong numbers.
            const char* rep = getenv("HALVA_FWD3_REPAIR");
            if (rc != HALVA_OK || !(rep && rep[0] == '1')) return rc;
            p.repair = 1;
            return launch_one(sdpa_fwd_kernel<D, CAUSAL, false>, p, CAUSAL, 256, 512, lds, S, st, "sdpa_fwd (repair)");
        }
    }
    return slow_tr_requested() ? launch_one(sdpa_fwd_kernel<D, CAUSAL, true>, p, CAUSAL, 256, 512, lds, S, st, "sdpa_fwd")
                               : launch_one(sdpa_fwd_kernel<D, CAUSAL, false>, p, CAUSAL, 256, 512, lds, S, st, "sdpa_fwd");
}

// sdpa_bwd_dkv3: 4 waves, one per SIMD (512 registers), 128 keys per workgroup, 130 KiB of LDS (four Q / dO tile slots + statistics)
template <bool CAUSAL>
int launch_dkv3(SdpaParams p, int S, hipStream_t st, bool use_asm) {
    // (p.nblk / p.npairs / p.sched_order: set by launch_bwd in front of the delta pass, which writes the item records from them)
    static std::atomic<int> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t ea = hipFuncSetAttribute((const void*)sdpa_bwd_dkv3_kernel<128, CAUSAL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV3_LDS);
        if (ea == hipSuccess)
            ea = hipFuncSetAttribute((const void*)sdpa_bwd_dkv3_kernel<128, CAUSAL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV3_LDS);
        if (ea != hipSuccess) {
            halva_set_error("sdpa_bwd_dkv3: hipFuncSetAttribute(%d B of LDS) failed: %s", DKV3_LDS, hipGetErrorString(ea));
            return HALVA_ERR_LAUNCH;
        }
        if (dev >= 0 && dev < 64) attr_set[dev].store(1, std::memory_order_release);
    }
    // one persistent workgroup per CU (it takes the CU's whole register file and 131 KiB of its LDS), fewer when there is less to do
    static std::atomic<int> n_cu[64];
    int cus = (dev >= 0 && dev < 64) ? n_cu[dev].load(std::memory_order_acquire) : 0;
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (dev >= 0 && dev < 64) n_cu[dev].store(cus, std::memory_order_release);
    }
    const int64_t items = (int64_t)p.nblk * p.npairs;
    const dim3 grid((unsigned)std::min<int64_t>(items, cus));
    if (use_asm) hipLaunchKernelGGL((sdpa_bwd_dkv3_kernel<128, CAUSAL, true>), grid, dim3(256), DKV3_LDS, st, p);
    else hipLaunchKernelGGL((sdpa_bwd_dkv3_kernel<128, CAUSAL, false>), grid, dim3(256), DKV3_LDS, st, p);
    HALVA_CHECK_LAUNCH("sdpa_bwd_dkv3");
    return HALVA_OK;
}

template <int D, bool CAUSAL>
int launch_bwd(const SdpaParams& p_in, int S, hipStream_t st, bool* fused_rope = nullptr) {
    SdpaParams p = p_in;
    if (fused_rope) *fused_rope = false;
#ifdef HALVA_STAMP
    p.dbg = halva_dbg_buffer();
#endif
    const size_t lds_dq = 4 * 64 * D * 2;
    const size_t lds_dkv = 6 * 64 * D * 2 + 6 * 64 * sizeof(float) + 2 * 4 * 2 * 2048;
    const bool slow = slow_tr_requested();
    if (p.ds_ws != nullptr && D == 128) {      // dS formed once: delta (+ zero-fill of padded dq rows) -> dK/dV (+ dS store) -> dQ = dS K
        const int64_t rows = (int64_t)S * p.T;
        // (its step counts travel as 16-bit fields, its scheduler divides in fp32, and a sequence's Q / dO rows are addressed through 32-bit
        // buffer descriptors - q_rec / do_rec / *_soff in sdpa_dkv3.h are (rows * ld * 2) as unsigned: the same bound as launch_fwd's)
        const bool dkv3 = !slow && p.lse2 != nullptr && rows >= 16 && p.T < (1 << 22) && (int64_t)S * p.H * ((p.T + 127) / 128) < (1 << 23) &&
                          (int64_t)p.T * std::max(p.ld_qkv, p.ld_do) * 2 < (1ll << 31) && env_flag_on("HALVA_SDPA_DKV3");
        // the inverse RoPE of dq / dk rides in the store epilogues of sdpa_bwd_dq2 and of sdpa_bwd_dkv3's generated build; any other kernel
        // combination leaves them un-rotated and halva_sdpa_branch_bwd_rope follows up with halva_rope_qk_branch (p_in.rope_cos stays set)
        const bool use_asm = env_flag_on("HALVA_DKV3_ASM");
        // (DKV3_ROWS_VIA_LDS: the dK rows are rotated by dkv3_store_rows_lds only - a -DDKV3_ROWS_VIA_LDS=0 build stores them un-rotated, so it
        // must not report the rotation as done)
        if (!(dkv3 && use_asm && !slow && DKV3_ROWS_VIA_LDS && env_flag_on("HALVA_ROPE_FUSED_BWD"))) p.rope_cos = p.rope_sin = nullptr;
        if (fused_rope) *fused_rope = p.rope_cos != nullptr;
        if (dkv3) {
            static_assert(CAUSAL || D != 128, "sdpa_bwd_dkv3's item records are written for the causal backward (dkv3_build_record<true>)");
            p.nblk = (p.T + 127) / 128;
            p.npairs = S * p.H;
            const char* e = getenv("HALVA_DKV3_ORDER");
            p.sched_order = e ? atoi(e) : (CAUSAL ? 2 : 0);
        }
        hipLaunchKernelGGL((sdpa_bwd_delta_kernel<D>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, S, dkv3 ? 1 : 0);
        HALVA_CHECK_LAUNCH("sdpa_bwd_delta");      // (a failed launch would leave stale delta / unzeroed padded dq rows for the two kernels below)
        int rc2;
        if (dkv3) rc2 = launch_dkv3<CAUSAL>(p, S, st, use_asm);
        else
            rc2 = slow ? launch_one(sdpa_bwd_dkv2_kernel<D, CAUSAL, true>, p, CAUSAL, 128, 512, lds_dkv, S, st, "sdpa_bwd_dkv2")
                       : launch_one(sdpa_bwd_dkv2_kernel<D, CAUSAL, false>, p, CAUSAL, 128, 512, lds_dkv, S, st, "sdpa_bwd_dkv2");
        if (rc2 != HALVA_OK) return rc2;
#ifdef HALVA_DS_EVICT_EXP      // experiments/ds_residency only: push the dS just written out of the Infinity Cache before the dQ kernel reads it
        if (const char* ev = getenv("HALVA_DS_EVICT_MB")) {
            static char* scratch = nullptr;
            static size_t scratch_bytes = 0;
            const size_t want = (size_t)atoll(ev) << 20;
            if (want > scratch_bytes) {
                if (scratch) (void)hipFree(scratch);
                (void)hipMalloc(&scratch, want);
                scratch_bytes = want;
            }
            if (want) (void)hipMemsetAsync(scratch, 0, want, st);
        }
#endif
        const size_t lds_dq2 = 3 * 64 * D * 2 + 8 * 3 * DS_LDS_SLOT;
        if (slow) return launch_one(sdpa_bwd_dq2_kernel<D, true>, p, true, 256, 512, lds_dq2, S, st, "sdpa_bwd_dq2");
        return env_flag_on("HALVA_DQ2_FAST_TILE") ? launch_one(sdpa_bwd_dq2_kernel<D, false>, p, true, 256, 512, lds_dq2, S, st, "sdpa_bwd_dq2")
                                                  : launch_one(sdpa_bwd_dq2_kernel<D, false, false>, p, true, 256, 512, lds_dq2, S, st, "sdpa_bwd_dq2 (general tile)");
    }
    const int rc = slow ? launch_one(sdpa_bwd_dq_kernel<D, CAUSAL, true, 8>, p, CAUSAL, 256, 512, lds_dq, S, st, "sdpa_bwd_dq")
                        : launch_one(sdpa_bwd_dq_kernel<D, CAUSAL, false, 8>, p, CAUSAL, 256, 512, lds_dq, S, st, "sdpa_bwd_dq");
    if (rc != HALVA_OK) return rc;
    return slow ? launch_one(sdpa_bwd_dkv2_kernel<D, CAUSAL, true>, p, CAUSAL, 128, 512, lds_dkv, S, st, "sdpa_bwd_dkv2")
                : launch_one(sdpa_bwd_dkv2_kernel<D, CAUSAL, false>, p, CAUSAL, 128, 512, lds_dkv, S, st, "sdpa_bwd_dkv2");
}

}  // namespace

extern "C" int halva_sdpa_block_pairs(int nblk, int start, int br_a, int br_b, int32_t* out) {
    HALVA_CHECK_ARG(nblk > 0 && out, "sdpa_block_pairs: bad arguments");
    const Branch br{br_a, br_b};
    for (int k = 0; k < (nblk + 1) / 2; ++k) {
        int heavy, light;
        paired_blocks(nblk, start, br, k, heavy, light);
        out[2 * k] = heavy, out[2 * k + 1] = light;
    }
    return HALVA_OK;
}

extern "C" int halva_sdpa_causal_fwd(const void* qkv, void* out, float* lse, const int32_t* seq_start, const int32_t* seq_len,
                                     int S, int T, int H, int D, float scale, void* stream) {
    return halva_sdpa_causal_fwd_ld(qkv, out, (int64_t)H * D, lse, seq_start, seq_len, S, T, H, D, scale, stream);
}

extern "C" int halva_sdpa_causal_fwd_ld(const void* qkv, void* out, int64_t ld_out, float* lse, const int32_t* seq_start,
                                        const int32_t* seq_len, int S, int T, int H, int D, float scale, void* stream) {
    return halva_sdpa_branch_fwd(qkv, out, ld_out, lse, seq_start, seq_len, nullptr, nullptr, S, T, H, D, scale, stream);
}

extern "C" int halva_sdpa_branch_fwd(const void* qkv, void* out, int64_t ld_out, float* lse, const int32_t* seq_start,
                                     const int32_t* seq_len, const int32_t* br_a, const int32_t* br_b, int S, int T, int H, int D,
                                     float scale, void* stream) {
    HALVA_CHECK_ARG((br_a == nullptr) == (br_b == nullptr), "sdpa_branch_fwd: br_a and br_b go together");
    HALVA_CHECK_ARG(qkv && out && lse, "sdpa_causal_fwd: null pointer");
    HALVA_CHECK_ARG(ld_out >= (int64_t)H * D && ld_out % 8 == 0, "sdpa_causal_fwd: bad output row stride %lld", (long long)ld_out);
    HALVA_CHECK_ARG(D == 128 || D == 64, "sdpa_causal_fwd: head_dim %d not supported (64 or 128)", D);
    HALVA_CHECK_ARG(S > 0 && T > 0 && H > 0, "sdpa_causal_fwd: bad sizes");
    SdpaParams p{};
    const bf16_t* base = (const bf16_t*)qkv;
    p.q = base;
    p.k = base + (int64_t)H * D;
    p.v = base + 2 * (int64_t)H * D;
    p.o = (bf16_t*)out;
    p.lse = lse;
    p.seq_start = seq_start;
    p.seq_len = seq_len;
    p.br_a = br_a;
    p.br_b = br_b;
    p.ld_qkv = 3 * (int64_t)H * D;
    p.ld_o = ld_out;
    p.T = T;
    p.H = H;
    p.scale = scale > 0.f ? scale : 1.f / sqrtf((float)D);
    return D == 128 ? launch_fwd<128, true>(p, S, (hipStream_t)stream) : launch_fwd<64, true>(p, S, (hipStream_t)stream);
}

extern "C" int halva_sdpa_causal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                     float* delta_ws, float* dq_ws, const int32_t* seq_start, const int32_t* seq_len, int S,
                                     int T, int H, int D, float scale, void* stream) {
    return halva_sdpa_causal_bwd_ld(qkv, out, (int64_t)H * D, dout, (int64_t)H * D, lse, dqkv, delta_ws, dq_ws, seq_start, seq_len,
                                    S, T, H, D, scale, stream);
}

extern "C" int halva_sdpa_causal_bwd_ld(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout,
                                        const float* lse, void* dqkv, float* delta_ws, float* dq_ws, const int32_t* seq_start,
                                        const int32_t* seq_len, int S, int T, int H, int D, float scale, void* stream) {
    (void)dq_ws;
    return halva_sdpa_branch_bwd(qkv, out, ld_out, dout, ld_dout, lse, dqkv, delta_ws, seq_start, seq_len, nullptr, nullptr, S, T, H, D,
                                 scale, stream);
}

static int64_t lse2_region_bytes(int S, int T, int H) { return (((int64_t)S * H * ((T + 63) / 64) * 512 + 256) + 127) / 128 * 128; }
static int64_t ds_region_bytes(int S, int T, int H) { return (int64_t)S * H * ((T + 127) / 128) * ((T + 63) / 64) * 16384; }
static int64_t items_region_bytes(int S, int T, int H) { return ((int64_t)S * H * ((T + 127) / 128) + 1) * 256; }      // sdpa_dkv3_items.h: one record per item + the empty one

extern "C" int64_t halva_sdpa_bwd_ws_bytes(int S, int T, int H, int D) {
    if (D != 128) return 0;                                      // the dS path is the head_dim-128 instantiation; others use the 3-product dQ kernel
    // dS, then [S, H, T] f32 lse * log2(e) for sdpa_bwd_dkv3 (+ one padding row), then its work-queue counters, then its item records
    return ds_region_bytes(S, T, H) + lse2_region_bytes(S, T, H) + 1024 + items_region_bytes(S, T, H);
}

extern "C" int halva_sdpa_branch_bwd(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout,
                                     const float* lse, void* dqkv, float* delta_ws, const int32_t* seq_start, const int32_t* seq_len,
                                     const int32_t* br_a, const int32_t* br_b, int S, int T, int H, int D, float scale, void* stream) {
    return halva_sdpa_branch_bwd_ws(qkv, out, ld_out, dout, ld_dout, lse, dqkv, delta_ws, nullptr, 0, seq_start, seq_len, br_a, br_b, S, T, H, D,
                                    scale, stream);
}

extern "C" int halva_sdpa_branch_bwd_ws(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout,
                                        const float* lse, void* dqkv, float* delta_ws, void* ds_ws, int64_t ds_ws_bytes,
                                        const int32_t* seq_start, const int32_t* seq_len, const int32_t* br_a, const int32_t* br_b, int S,
                                        int T, int H, int D, float scale, void* stream) {
    return halva_sdpa_branch_bwd_rope(qkv, out, ld_out, dout, ld_dout, lse, dqkv, delta_ws, ds_ws, ds_ws_bytes, seq_start, seq_len, br_a, br_b,
                                      nullptr, nullptr, 0, S, T, H, D, scale, stream);
}

extern "C" int halva_sdpa_branch_bwd_rope(const void* qkv, const void* out, int64_t ld_out, const void* dout, int64_t ld_dout,
                                          const float* lse, void* dqkv, float* delta_ws, void* ds_ws, int64_t ds_ws_bytes,
                                          const int32_t* seq_start, const int32_t* seq_len, const int32_t* br_a, const int32_t* br_b,
                                          const void* rope_cos, const void* rope_sin, int max_pos, int S, int T, int H, int D, float scale,
                                          void* stream) {
    HALVA_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr), "sdpa_branch_bwd_rope: cos and sin go together");
    HALVA_CHECK_ARG(rope_cos == nullptr || T <= max_pos, "sdpa_branch_bwd_rope: T=%d exceeds the cos/sin table (%d rows)", T, max_pos);
    HALVA_CHECK_ARG(ds_ws == nullptr || ds_ws_bytes >= halva_sdpa_bwd_ws_bytes(S, T, H, D),
                    "sdpa_branch_bwd_ws: workspace of %lld bytes, %lld needed", (long long)ds_ws_bytes,
                    (long long)halva_sdpa_bwd_ws_bytes(S, T, H, D));
    HALVA_CHECK_ARG((br_a == nullptr) == (br_b == nullptr), "sdpa_branch_bwd: br_a and br_b go together");
    HALVA_CHECK_ARG(ld_out >= (int64_t)H * D && ld_out % 8 == 0 && ld_dout >= (int64_t)H * D && ld_dout % 8 == 0,
                    "sdpa_causal_bwd: bad row strides %lld / %lld", (long long)ld_out, (long long)ld_dout);   // reserved for an atomics-based dQ variant; the shipped dQ kernel needs no scratch
    HALVA_CHECK_ARG(qkv && out && dout && lse && dqkv && delta_ws, "sdpa_causal_bwd: null pointer");
    HALVA_CHECK_ARG(D == 128 || D == 64, "sdpa_causal_bwd: head_dim %d not supported (64 or 128)", D);
    HALVA_CHECK_ARG(S > 0 && T > 0 && H > 0, "sdpa_causal_bwd: bad sizes");
    SdpaParams p{};
    const bf16_t* base = (const bf16_t*)qkv;
    p.q = base;
    p.k = base + (int64_t)H * D;
    p.v = base + 2 * (int64_t)H * D;
    p.o_in = (const bf16_t*)out;
    p.d_o = (const bf16_t*)dout;
    bf16_t* dbase = (bf16_t*)dqkv;
    p.dq = dbase;
    p.dk = dbase + (int64_t)H * D;
    p.dv = dbase + 2 * (int64_t)H * D;
    p.lse = const_cast<float*>(lse);
    p.delta = delta_ws;
    p.seq_start = seq_start;
    p.seq_len = seq_len;
    p.br_a = br_a;
    p.br_b = br_b;
    p.ld_qkv = 3 * (int64_t)H * D;
    p.ld_o = ld_out;
    p.ld_do = ld_dout;
    p.T = T;
    p.H = H;
    p.ds_ws = D == 128 ? (char*)ds_ws : nullptr;
    p.lse2 = p.ds_ws ? reinterpret_cast<float*>(p.ds_ws + ds_region_bytes(S, T, H)) : nullptr;
    p.sched = p.ds_ws ? reinterpret_cast<int*>(p.ds_ws + ds_region_bytes(S, T, H) + lse2_region_bytes(S, T, H)) : nullptr;
    p.items = p.ds_ws ? reinterpret_cast<int*>(p.ds_ws + ds_region_bytes(S, T, H) + lse2_region_bytes(S, T, H) + 1024) : nullptr;
    p.ds_nkb = (T + 127) / 128;
    p.ds_nt = (T + 63) / 64;
    p.stat_nt = (T + 63) / 64;
    p.scale = scale > 0.f ? scale : 1.f / sqrtf((float)D);
    p.rope_cos = (const bf16_t*)rope_cos;
    p.rope_sin = (const bf16_t*)rope_sin;
    p.rope_max_pos = max_pos;
    bool fused = false;
    const int rc = D == 128 ? launch_bwd<128, true>(p, S, (hipStream_t)stream, &fused) : launch_bwd<64, true>(p, S, (hipStream_t)stream, &fused);
    if (rc != HALVA_OK || rope_cos == nullptr || fused) return rc;
    // a kernel combination without the rotating epilogues (head_dim 64, no dS workspace, the debugging builds): the rotation as its own launch
    return halva_rope_qk_branch(dqkv, rope_cos, rope_sin, br_a, br_b, (int64_t)S * T, T, H, D, max_pos, 1, stream);
}

extern "C" int halva_sdpa_full_fwd(const void* qkv, void* out, int N, int S, int H, int D, float scale, void* stream) {
    HALVA_CHECK_ARG(qkv && out, "sdpa_full_fwd: null pointer");
    HALVA_CHECK_ARG(D == 128 || D == 64, "sdpa_full_fwd: head_dim %d not supported (64 or 128)", D);
    HALVA_CHECK_ARG(N > 0 && S > 0 && H > 0, "sdpa_full_fwd: bad sizes");
    SdpaParams p{};
    const bf16_t* base = (const bf16_t*)qkv;
    p.q = base;
    p.k = base + (int64_t)H * D;
    p.v = base + 2 * (int64_t)H * D;
    p.o = (bf16_t*)out;
    p.lse = nullptr;
    p.ld_qkv = 3 * (int64_t)H * D;
    p.ld_o = (int64_t)H * D;
    p.T = S;
    p.H = H;
    p.scale = scale > 0.f ? scale : 1.f / sqrtf((float)D);
    return D == 128 ? launch_fwd<128, false>(p, N, (hipStream_t)stream) : launch_fwd<64, false>(p, N, (hipStream_t)stream);
}
