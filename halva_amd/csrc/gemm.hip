// bf16 GEMM on v_mfma_f32_32x32x16_bf16 for the small dense contractions of the vision side of the DPA step:
// the mm_projector MLP (fwd: bias + GELU epilogue; bwd: NN and TN forms) and CLIP's patch-embed (im2col + GEMM).
//   C[M,N] = epi( opA(A) @ opB(B)^T + bias ),   opA(A) = A[M,K] or A given as [K,M];  opB(B) = B[N,K] or [K,N]
// Workgroup = 4 waves as 2x2, tile 128x128x64, each wave 64x64 (2x2 MFMA tiles); operands are register-staged
// into double-buffered swizzled LDS tiles; transposed operands are read with ds_read_b64_tr_b16.
#include "common.h"
#include <cstdlib>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int off64(int row, int ch) {    // [rows][64] bf16 tile, 128-byte rows
    return row * 128 + ((ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4);
}
__device__ __forceinline__ int off128(int row, int ch) {   // [rows][128] bf16 tile, 256-byte rows
    return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// One 128(rows of the output dim) x 64(k) operand tile.  TRANS=false: source is [dim][K] (k contiguous) and the LDS
// image is [128][64]; TRANS=true: source is [K][dim] and the LDS image is [64][128], read transposed.
template <bool TRANS>
struct Operand {
    u32x4 r[4];
    __device__ __forceinline__ void load(const bf16_t* src, int64_t ld, int dim0, int dim_lim, int k0, int K) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cid = threadIdx.x + 256 * i;
            bool ok;
            const bf16_t* ptr;
            if (!TRANS) {
                const int row = cid >> 3, ch = cid & 7;
                ok = (dim0 + row < dim_lim) && (k0 + ch * 8 < K);
                ptr = src + (int64_t)(dim0 + row) * ld + k0 + ch * 8;
            } else {
                const int row = cid >> 4, ch = cid & 15;
                ok = (k0 + row < K) && (dim0 + ch * 8 < dim_lim);
                ptr = src + (int64_t)(k0 + row) * ld + dim0 + ch * 8;
            }
            r[i] = ok ? *reinterpret_cast<const u32x4*>(ptr) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __device__ __forceinline__ void store(char* tile) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cid = threadIdx.x + 256 * i;
            const int off = TRANS ? off128(cid >> 4, cid & 15) : off64(cid >> 3, cid & 7);
            *reinterpret_cast<u32x4*>(tile + off) = r[i];
        }
    }
    // fragment for rows [row0, row0+32) of the output dim, k-step ks (16 deep): lane (r, h) holds k = 16*ks + 8*h + j
    static __device__ __forceinline__ s16x8 frag(const char* tile, int row0, int ks, int lane) {
        if (!TRANS) {
            return *reinterpret_cast<const s16x8*>(tile + off64(row0 + (lane & 31), 2 * ks + (lane >> 5)));
        } else {
            s16x8 out;
            const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = g >> 1;
            const int c = row0 + 16 * (g & 1) + 4 * pp;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int kr = 16 * ks + 8 * h + 4 * jj + q;   // natural k order: element j = 4*jj + e <-> k = 8*h + j
                const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(tile + off128(kr, c >> 3) + (c & 7) * 2));
                out[4 * jj + 0] = t[0];
                out[4 * jj + 1] = t[1];
                out[4 * jj + 2] = t[2];
                out[4 * jj + 3] = t[3];
            }
            return out;
        }
    }
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    const bf16_t* bias;
    void* C;
    void* pre;      // optional pre-activation output (same dtype/shape as C)
    int64_t lda, ldb, ldc;
    int M, N, K;
    int epilogue, out_f32, accumulate;
    int ksplit;     // > 0: blockIdx.z owns k in [z * ksplit, min(K, (z + 1) * ksplit)) and writes its f32 partial to C + z * M * ldc
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    constexpr int TILE = 128 * 64 * 2;   // bytes of one operand tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_lds = smem;               // [2][TILE]
    char* b_lds = smem + 2 * TILE;    // [2][TILE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Operand<TA> sa;
    Operand<TB> sb;
    GemmParams q = p;      // this block's view of the problem (a k-slab of it when the launch is split over k)
    if (p.ksplit > 0) {
        const int64_t kb = (int64_t)blockIdx.z * p.ksplit;
        q.K = (int)min((int64_t)p.ksplit, p.K - kb);
        q.A = p.A + (TA ? kb * p.lda : kb);
        q.B = p.B + (TB ? kb * p.ldb : kb);
        q.C = (float*)p.C + (int64_t)blockIdx.z * p.M * p.ldc;
    }
    const int nk = (q.K + 63) / 64;
    sa.load(q.A, q.lda, m0, q.M, 0, q.K);
    sb.load(q.B, q.ldb, n0, q.N, 0, q.K);
    sa.store(a_lds);
    sb.store(b_lds);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const char* at = a_lds + (it & 1) * TILE;
        const char* bt = b_lds + (it & 1) * TILE;
        if (it + 1 < nk) {
            sa.load(q.A, q.lda, m0, q.M, (it + 1) * 64, q.K);
            sb.load(q.B, q.ldb, n0, q.N, (it + 1) * 64, q.K);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = Operand<TA>::frag(at, 64 * wm + 32 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Operand<TB>::frag(bt, 64 * wn + 32 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                       __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
        }
        if (it + 1 < nk) {
            sa.store(a_lds + ((it + 1) & 1) * TILE);
            sb.store(b_lds + ((it + 1) & 1) * TILE);
        }
        __syncthreads();
    }
    // epilogue: accumulator column (lane & 31) = n, rows = m
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 64 * wn + 32 * j + (lane & 31);
        if (n >= p.N) continue;
        const float bv = p.bias ? bf16_to_f32(p.bias[n]) : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                const int64_t idx = (int64_t)m * p.ldc + n;
                if (p.pre) {
                    if (p.out_f32) ((float*)p.pre)[idx] = v; else ((bf16_t*)p.pre)[idx] = f32_to_bf16(v);
                }
                if (p.epilogue == 1) v = gelu_erf(p.out_f32 ? v : bf16_round(v));
                if (p.out_f32) {
                    float* c = (float*)q.C + idx;
                    *c = p.accumulate ? *c + v : v;
                } else {
                    bf16_t* c = (bf16_t*)p.C + idx;
                    *c = f32_to_bf16(p.accumulate ? bf16_to_f32(*c) + v : v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the TN form for the LoRA weight gradients (halva_wgrad_accumulate: C[M, N] f32 += A^T B over `rows`, A [rows, M] and B [rows, N] column
// windows of row-major activations, M and N multiples of 128) with the operand tiles brought in by LDS-DMA instead of through registers:
//   * a k-tile = 64 rows of 128 columns (256 bytes each) per operand = sixteen 1-KiB pieces of four rows; wave w requests pieces w, w + 4, w + 8,
//     w + 12 of both operands with `buffer_load_dwordx4 ... offen lds` through a bounds-checked descriptor over the k-slab (a row past the slab
//     arrives as ZEROS: no tail case; experiments/fwd3/oob_probe.hip) - no staging registers, no ds_write (8 x 13 cycles per thread and tile before);
//   * the LDS image keeps the XOR swizzle of Operand<true> (its ds_read_b64_tr_b16 fragments are conflict-free on it): LDS-DMA writes lane l
//     to byte 16 l of the piece, so the swizzle is applied to WHICH chunk of its row a lane fetches - constant per wave (piece & 3 == wave);
//   * two stages (64 KiB: two workgroups per CU): iteration t waits for its own requests of tile t, barrier (tile t complete for everyone; everyone
//     has finished tile t - 1), requests tile t + 1 into the other stage, multiplies tile t.
// Same arithmetic and summation order inside a slab as gemm_kernel<true, true>: bitwise the same partials.
// KT = 64: two stages, one tile in flight per workgroup while it multiplies.  KT = 32: FOUR stages of half the height - tile t + 3 is requested
// when tile t has landed, so three tiles (48 KiB per workgroup) stay in flight all the time, at twice the barriers (HALVA_WGRAD_KT=32).
// (the body of one workgroup = one [128 x 128] tile of one k-slab: bx / by / bz = the tile's column, row and slab - blockIdx of the single-problem launch,
// decoded from a linear index by the batched one)
template <int KT, int NT_MODE>      // NT_MODE bit 0: A streamed nontemporally, bit 1: B (see `request`)
__device__ __forceinline__ void wgrad_dma_body(const GemmParams& p, int bx, int by, int bz, char* smem) {
    constexpr int TILE = 128 * KT * 2, NST = 128 / KT, PPW = KT / 16;      // bytes of an operand tile; stages; 1-KiB pieces per wave and operand
    char* a_lds = smem;                 // [NST][TILE]
    char* b_lds = smem + NST * TILE;    // [NST][TILE]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * 128, n0 = bx * 128;
    const int64_t kb = (int64_t)bz * p.ksplit;
    const int K = (int)min((int64_t)p.ksplit, p.K - kb);
    const int nk = (K + KT - 1) / KT;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // descriptors over the slab's rows of this tile's 128 columns: base, stride 0, bytes up to the end of the last row's 128 columns, raw dwords
    auto desc = [&](const bf16_t* base, int64_t ld) {
        const uint64_t a = (uint64_t)(size_t)base;
        u32x4 d;
        d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        d[2] = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)(K - 1) * ld + 128) * 2));
        d[3] = 0x00020000u;
        return d;
    };
    const u32x4 da = desc(p.A + kb * p.lda + m0, p.lda), db = desc(p.B + kb * p.ldb + n0, p.ldb);
    // lane (row l >> 4 of the piece, position l & 15) fetches chunk (position ^ swizzle(row)) of its row: off128's image
    const int prow = lane >> 4, chunk = (lane & 15) ^ ((prow << 2) | wave);
    const unsigned voa = (unsigned)(prow * p.lda * 2 + chunk * 16), vob = (unsigned)(prow * p.ldb * 2 + chunk * 16);
    const unsigned lds_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)a_lds, lds_b = (unsigned)(size_t)(__attribute__((address_space(3))) char*)b_lds;
    auto request = [&](int it, int stage) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + 4 * i;
            const unsigned sa = (unsigned)(((int64_t)it * KT + 4 * piece) * p.lda * 2), sb = (unsigned)(((int64_t)it * KT + 4 * piece) * p.ldb * 2);
            const unsigned dst_a = lds_a + stage * TILE + piece * 1024, dst_b = lds_b + stage * TILE + piece * 1024;
            // An operand whose 128-column slab is read by ONE workgroup of the grid (the wide side of a LoRA factor's gradient: gridDim.x == 1 for A,
            // gridDim.y == 1 for B: NT_MODE, chosen by the launcher) is streamed nontemporally - read once, it would only push the other operand's slab, which every workgroup of the
            // row / column re-reads, out of the L2 (round 5: the row kernels gained 6-16 % from the same policy, experiments/rowops_stream).
            unsigned keep;
#define WGRAD_REQ(NTA, NTB)                                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %7 offen" NTA " lds\n\t"                  \
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %8 offen" NTB " lds\n\ts_mov_b32 m0, %0"                      \
                 : "=&s"(keep) : "v"(voa), "v"(vob), "s"(da), "s"(db), "s"(dst_a), "s"(dst_b), "s"(sa), "s"(sb) : "memory")
            if constexpr (NT_MODE == 0) WGRAD_REQ("", "");
            else if constexpr (NT_MODE == 1) WGRAD_REQ(" nt", "");
            else if constexpr (NT_MODE == 2) WGRAD_REQ("", " nt");
            else WGRAD_REQ(" nt", " nt");
#undef WGRAD_REQ
        }
    };
    // NST - 1 tiles ahead (every wave issues 2 PPW requests per tile, whether the tile exists or not: rows past the slab bring zeros and the
    // counted wait stays the same to the end)
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) request(t, t);
#pragma unroll 1
    for (int it = 0; it < nk; ++it) {
        // tile `it` is complete for every wave (the NST - 2 younger tiles may stay in flight); every wave is through with tile it - 1
        if (NST == 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");      // (NST = 4: 2 tiles x 4 requests)
        request(it + NST - 1, (it + NST - 1) % NST);
        const char* at = a_lds + (it % NST) * TILE;
        const char* bt = b_lds + (it % NST) * TILE;
#ifdef HALVA_WGRAD_DIAG_NOMFMA      // (timing experiment: the request path alone; results are wrong)
        if (p.K < 0)
#endif
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            s16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = Operand<true>::frag(at, 64 * wm + 32 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Operand<true>::frag(bt, 64 * wn + 32 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                       __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the requests past the slab's end)
    float* cz = (float*)p.C + (int64_t)bz * p.M * p.ldc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 64 * wn + 32 * j + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                cz[(int64_t)m * p.ldc + n] = acc[i][j][r];
            }
    }
}
template <int KT, int NT_MODE>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_dma_body<KT, NT_MODE>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Round 6: the weight gradients of ONE LoRA group - its A factor and its one to three B factors, two to four products that
// become available together in the group's backward (halva_amd/llama.py:_LoraGroupLinear) - as ONE launch of the tile kernel and ONE of the reduction
// instead of two launches per factor (1 408 -> 512 per bench step): every workgroup finds its problem from a prefix table in the kernel arguments and
// runs the single-problem body on it, with the slab counts of the single-problem launcher - the partials, their order and the results are BITWISE those
// of halva_wgrad_accumulate called once per factor (tests/test_hip_kernels.py).  What it saves is kernel boundaries (each one drains and refills the
// chip), not work.
constexpr int WGRAD_BATCH_MAX = 4;
struct WgradBatch {
    GemmParams p[WGRAD_BATCH_MAX];      // p[q].C: the partials of problem q in the workspace
    float* C[WGRAD_BATCH_MAX];
    float alpha[WGRAD_BATCH_MAX];
    int64_t mn[WGRAD_BATCH_MAX];
    int splits[WGRAD_BATCH_MAX], nt_mode[WGRAD_BATCH_MAX], gx[WGRAD_BATCH_MAX], gy[WGRAD_BATCH_MAX];
    int first_block[WGRAD_BATCH_MAX + 1];       // prefix sums of gx * gy * splits
    int first_rblock[WGRAD_BATCH_MAX + 1];      // prefix sums of the reduction's blocks (1 024 elements each)
    int n;
};
__global__ __launch_bounds__(256, 2) void wgrad_dma_batch_kernel(const WgradBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int q = 0;
    while (q + 1 < b.n && (int)blockIdx.x >= b.first_block[q + 1]) ++q;      // (uniform: scalar loads from the kernel arguments)
    const int i = (int)blockIdx.x - b.first_block[q], gx = b.gx[q], gy = b.gy[q];
    const int bx = i % gx, by = (i / gx) % gy, bz = i / (gx * gy);
    const GemmParams& p = b.p[q];
    switch (b.nt_mode[q]) {
    case 0: wgrad_dma_body<64, 0>(p, bx, by, bz, smem); break;
    case 1: wgrad_dma_body<64, 1>(p, bx, by, bz, smem); break;
    case 2: wgrad_dma_body<64, 2>(p, bx, by, bz, smem); break;
    default: wgrad_dma_body<64, 3>(p, bx, by, bz, smem); break;
    }
}

// images [n, 3, hw, hw] bf16 -> col [n * (hw/p)^2, Kp] bf16, k = (c, ky, kx), zero padded to Kp
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ col, int n, int hw,
                                                     int p, int Kp, int64_t total) {
    const int np1 = hw / p, K = 3 * p * p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const int64_t row = i / Kp;
        bf16_t v = 0;
        if (k < K) {
            const int px = (int)(row % np1), py = (int)((row / np1) % np1);
            const int64_t b = row / ((int64_t)np1 * np1);
            const int c = k / (p * p), ky = (k / p) % p, kx = k % p;
            v = img[((b * 3 + c) * hw + (py * p + ky)) * hw + px * p + kx];
        }
        col[i] = v;
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ hpre,
                                                       bf16_t* __restrict__ dh, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float x = bf16_to_f32(hpre[i]);
        const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        dh[i] = f32_to_bf16(bf16_to_f32(dy[i]) * (cdf + x * pdf));
    }
}

// out[n] += sum_m x[m][n]; grid (ceil(N/256), splits); each block sums a slab of rows, one atomic per column
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int64_t M, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int64_t per = (M + gridDim.y - 1) / gridDim.y;
    const int64_t lo = blockIdx.y * per, hi = lo + per < M ? lo + per : M;
    float s = 0.f;
    for (int64_t m = lo; m < hi; ++m) s += bf16_to_f32(x[m * N + n]);
    atomicAdd(out + n, s);
}

// C[i] += alpha * sum_z ws[z][i]  (fixed order: the split-k weight gradients stay bitwise reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t mn, int splits,
                                                            float alpha) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= mn) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(ws + z * mn + i);
    f32x4* c = reinterpret_cast<f32x4*>(C + i);
    *c = *c + s * alpha;
}

__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const WgradBatch b) {
    int q = 0;
    while (q + 1 < b.n && (int)blockIdx.x >= b.first_rblock[q + 1]) ++q;
    const int64_t i = ((int64_t)((int)blockIdx.x - b.first_rblock[q]) * 256 + threadIdx.x) * 4, mn = b.mn[q];
    if (i >= mn) return;
    const float* ws = (const float*)b.p[q].C;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < b.splits[q]; ++z) s += *reinterpret_cast<const f32x4*>(ws + z * mn + i);
    f32x4* c = reinterpret_cast<f32x4*>(b.C[q] + i);
    *c = *c + s * b.alpha[q];
}

template <bool TA, bool TB>
int launch_gemm(const GemmParams& p, hipStream_t st) {
    const dim3 grid((p.N + 127) / 128, (p.M + 127) / 128, p.ksplit > 0 ? (p.K + p.ksplit - 1) / p.ksplit : 1), block(256);
    const size_t lds = 4 * 128 * 64 * 2;
    (void)hipFuncSetAttribute((const void*)gemm_kernel<TA, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemm_kernel<TA, TB>), grid, block, lds, st, p);
    HALVA_CHECK_LAUNCH("gemm_bf16");
    return HALVA_OK;
}

}  // namespace

extern "C" int halva_gemm_bf16(const void* A, const void* B, const void* bias, void* C, void* pre_act, int M, int N, int K,
                               int trans_a, int trans_b, int epilogue, halva_dtype out_dtype, int accumulate, void* stream) {
    HALVA_CHECK_ARG(A && B && C, "gemm_bf16: null pointer");
    HALVA_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm_bf16: bad sizes %d %d %d", M, N, K);
    HALVA_CHECK_ARG(epilogue == 0 || epilogue == 1, "gemm_bf16: unknown epilogue %d", epilogue);
    // 16-byte row chunks: the contiguous dimension of every operand must be a multiple of 8 elements
    HALVA_CHECK_ARG((trans_a ? M : K) % 8 == 0, "gemm_bf16: contiguous dim of A must be a multiple of 8");
    HALVA_CHECK_ARG((trans_b ? N : K) % 8 == 0, "gemm_bf16: contiguous dim of B must be a multiple of 8");
    HALVA_CHECK_ARG(!(trans_a && !trans_b), "gemm_bf16: (trans_a, !trans_b) is not used on this path");
    GemmParams p{};
    p.A = (const bf16_t*)A;
    p.B = (const bf16_t*)B;
    p.bias = (const bf16_t*)bias;
    p.C = C;
    p.pre = pre_act;
    p.lda = trans_a ? M : K;
    p.ldb = trans_b ? N : K;
    p.ldc = N;
    p.M = M;
    p.N = N;
    p.K = K;
    p.epilogue = epilogue;
    p.out_f32 = out_dtype == HALVA_F32;
    p.accumulate = accumulate;
    if (!trans_a && !trans_b) return launch_gemm<false, false>(p, (hipStream_t)stream);
    if (!trans_a && trans_b) return launch_gemm<false, true>(p, (hipStream_t)stream);
    return launch_gemm<true, true>(p, (hipStream_t)stream);
}

// the k-slabs of one weight-gradient product: as many as keep the whole grid resident at once (2 workgroups per CU x 256 CUs): a second, partial
// round of workgroups costs more than the extra parallelism brings (measured at 256..1536 workgroups, tools/bench_wgrad.py)
static void wgrad_slabs(int M, int N, int64_t rows, int64_t ws_floats, int& splits, int& ksplit) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int64_t mn = (int64_t)M * N;
    splits = (int)max((int64_t)1, min((int64_t)min(64, 512 / tiles), ws_floats / mn));
    ksplit = (int)(((rows + splits - 1) / splits + 63) / 64 * 64);
    splits = (int)((rows + ksplit - 1) / ksplit);
}
static bool wgrad_dma_ok(int M, int N, int64_t lda, int64_t ldb, int ksplit) {      // HALVA_WGRAD_DMA=0: the register-staged gemm_kernel<true, true> of rounds 2-3 (also what odd shapes take)
    const char* e_dma = getenv("HALVA_WGRAD_DMA");
    const int64_t slab_bytes = (int64_t)ksplit * (lda > ldb ? lda : ldb) * 2;
    return !(e_dma && e_dma[0] == '0') && M % 128 == 0 && N % 128 == 0 && slab_bytes < (1ll << 31);
}

extern "C" int halva_wgrad_accumulate(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int M, int N, int64_t rows,
                                      float alpha, float* ws, int64_t ws_floats, void* stream) {
    HALVA_CHECK_ARG(A && B && C && ws, "wgrad_accumulate: null pointer");
    HALVA_CHECK_ARG(M > 0 && N > 0 && rows > 0 && rows < (1ll << 31), "wgrad_accumulate: bad sizes %d %d %lld", M, N, (long long)rows);
    HALVA_CHECK_ARG(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N,
                    "wgrad_accumulate: M, lda, ldb must be multiples of 8 and the strides cover the columns");
    HALVA_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)ws) & 15) == 0, "wgrad_accumulate: 16-byte aligned pointers");
    const int64_t mn = (int64_t)M * N;
    HALVA_CHECK_ARG(ws_floats >= mn, "wgrad_accumulate: workspace of %lld floats, need at least M * N = %lld", (long long)ws_floats,
                    (long long)mn);
    int splits, ksplit;
    wgrad_slabs(M, N, rows, ws_floats, splits, ksplit);
    GemmParams p{};
    p.A = (const bf16_t*)A;
    p.B = (const bf16_t*)B;
    p.C = ws;
    p.lda = lda;
    p.ldb = ldb;
    p.ldc = N;
    p.M = M;
    p.N = N;
    p.K = (int)rows;
    p.out_f32 = 1;
    p.ksplit = ksplit;
    if (wgrad_dma_ok(M, N, lda, ldb, ksplit)) {
        const dim3 grid(N / 128, M / 128, splits);
        const size_t lds = 4 * 128 * 64 * 2;
        const char* e_kt = getenv("HALVA_WGRAD_KT");
        const bool kt32 = e_kt && atoi(e_kt) == 32;
        const int nt_mode = (grid.x == 1 ? 1 : 0) | (grid.y == 1 ? 2 : 0);
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, p);
        };
        if (kt32) go(wgrad_dma_kernel<32, 0>);
        else if (nt_mode == 0) go(wgrad_dma_kernel<64, 0>);
        else if (nt_mode == 1) go(wgrad_dma_kernel<64, 1>);
        else if (nt_mode == 2) go(wgrad_dma_kernel<64, 2>);
        else go(wgrad_dma_kernel<64, 3>);
        HALVA_CHECK_LAUNCH("wgrad_dma");
    } else {
        const int rc = launch_gemm<true, true>(p, (hipStream_t)stream);
        if (rc != HALVA_OK) return rc;
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, C, mn, splits,
                       alpha);
    HALVA_CHECK_LAUNCH("splitk_reduce");
    return HALVA_OK;
}

extern "C" int halva_wgrad_accumulate_batch(int n, const halva_wgrad_item* items, float* ws, int64_t ws_floats, void* stream) {
    HALVA_CHECK_ARG(n > 0 && items && ws, "wgrad_accumulate_batch: no items / null pointer");
    // one launch pair when every product takes the LDS-DMA tile kernel and all partials fit the workspace side by side - with the slab counts the
    // single-problem call would choose for each, so that the results are bitwise the same; anything else: the single-problem calls, one by one
    WgradBatch b{};
    const char* e_kt = getenv("HALVA_WGRAD_KT");
    bool batched = n <= WGRAD_BATCH_MAX && !(e_kt && atoi(e_kt) == 32) && (((uintptr_t)ws) & 15) == 0;
    int64_t ws_at = 0;
    for (int q = 0; q < n && batched; ++q) {
        const halva_wgrad_item& it = items[q];
        const int64_t mn = (int64_t)it.M * it.N;
        if (!(it.A && it.B && it.C && it.M > 0 && it.N > 0 && it.rows > 0 && it.rows < (1ll << 31) && it.lda % 8 == 0 && it.ldb % 8 == 0 && it.lda >= it.M &&
              it.ldb >= it.N && (((uintptr_t)it.A | (uintptr_t)it.B | (uintptr_t)it.C) & 15) == 0 && ws_floats >= mn)) {
            batched = false;
            break;
        }
        int splits, ksplit;
        wgrad_slabs(it.M, it.N, it.rows, ws_floats, splits, ksplit);
        if (!wgrad_dma_ok(it.M, it.N, it.lda, it.ldb, ksplit) || ws_at + (int64_t)splits * mn > ws_floats) {
            batched = false;
            break;
        }
        GemmParams& p = b.p[q];
        p.A = (const bf16_t*)it.A, p.B = (const bf16_t*)it.B, p.C = ws + ws_at;
        p.lda = it.lda, p.ldb = it.ldb, p.ldc = it.N, p.M = it.M, p.N = it.N, p.K = (int)it.rows, p.out_f32 = 1, p.ksplit = ksplit;
        b.C[q] = it.C, b.alpha[q] = it.alpha, b.mn[q] = mn, b.splits[q] = splits, b.gx[q] = it.N / 128, b.gy[q] = it.M / 128;
        b.nt_mode[q] = (b.gx[q] == 1 ? 1 : 0) | (b.gy[q] == 1 ? 2 : 0);
        b.first_block[q + 1] = b.first_block[q] + b.gx[q] * b.gy[q] * splits;
        b.first_rblock[q + 1] = b.first_rblock[q] + (int)((mn / 4 + 255) / 256);
        ws_at += ((int64_t)splits * mn + 3) / 4 * 4;
    }
    if (!batched) {
        for (int q = 0; q < n; ++q) {
            const halva_wgrad_item& it = items[q];
            const int rc = halva_wgrad_accumulate(it.A, it.lda, it.B, it.ldb, it.C, it.M, it.N, it.rows, it.alpha, ws, ws_floats, stream);
            if (rc != HALVA_OK) return rc;
        }
        return HALVA_OK;
    }
    b.n = n;
    const size_t lds = 4 * 128 * 64 * 2;
    (void)hipFuncSetAttribute((const void*)wgrad_dma_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wgrad_dma_batch_kernel, dim3((unsigned)b.first_block[n]), dim3(256), lds, (hipStream_t)stream, b);
    HALVA_CHECK_LAUNCH("wgrad_dma_batch");
    hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3((unsigned)b.first_rblock[n]), dim3(256), 0, (hipStream_t)stream, b);
    HALVA_CHECK_LAUNCH("splitk_reduce_batch");
    return HALVA_OK;
}

extern "C" int halva_vit_patch_embed(const void* images, const void* weight_kp, const void* bias, void* col_ws, void* out, int n,
                                     int hw, int p, int d, int Kp, void* stream) {
    HALVA_CHECK_ARG(images && weight_kp && col_ws && out, "vit_patch_embed: null pointer");
    HALVA_CHECK_ARG(n > 0 && p > 0 && hw >= p, "vit_patch_embed: bad image/patch size");
    HALVA_CHECK_ARG(Kp % 8 == 0 && Kp >= 3 * p * p, "vit_patch_embed: Kp=%d must be a multiple of 8 and >= 3*p*p", Kp);
    const int np = (hw / p) * (hw / p);   // 'valid' convolution: a trailing partial patch is dropped (384 = 27*14 + 6)
    const int64_t total = (int64_t)n * np * Kp;
    int64_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)images,
                       (bf16_t*)col_ws, n, hw, p, Kp, total);
    HALVA_CHECK_LAUNCH("im2col");
    return halva_gemm_bf16(col_ws, weight_kp, bias, out, nullptr, n * np, d, Kp, 0, 0, 0, HALVA_BF16, 0, stream);
}

extern "C" int halva_clip_patch_embed(const void* images, const void* weight_kp, void* col_ws, void* out, int n, int hw, int p,
                                      int d, int Kp, void* stream) {
    HALVA_CHECK_ARG(p > 0 && hw % p == 0, "clip_patch_embed: bad image/patch size");
    return halva_vit_patch_embed(images, weight_kp, nullptr, col_ws, out, n, hw, p, d, Kp, stream);
}

extern "C" int halva_gelu_bwd(const void* dy, const void* h, void* dh, int64_t M, int N, void* stream) {
    HALVA_CHECK_ARG(dy && h && dh, "gelu_bwd: null pointer");
    const int64_t total = M * N;
    if (total <= 0) return HALVA_OK;
    int64_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)h, (bf16_t*)dh, total);
    HALVA_CHECK_LAUNCH("gelu_bwd");
    return HALVA_OK;
}

// dst[c][r] = src[r][c] for a [rows x cols] bf16 matrix with row strides ld_src / ld_dst (elements): 64 x 64 tiles through LDS (33-word rows:
// the transposed reads of a wave fall on different banks), 16-byte global loads and stores on both sides.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int64_t ld_src, bf16_t* __restrict__ dst, int64_t ld_dst,
                                                             int rows, int cols) {
    __shared__ unsigned tile[64][33];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, v = threadIdx.x & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (threadIdx.x >> 3) + 32 * i, gr = r0 + r, gc = c0 + 8 * v;
        u32x4 x = {0u, 0u, 0u, 0u};
        if (gr < rows && gc < cols) x = *reinterpret_cast<const u32x4*>(src + (int64_t)gr * ld_src + gc);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[r][4 * v + j] = x[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (threadIdx.x >> 3) + 32 * i, gc = c0 + c, gr = r0 + 8 * v;
        if (gc < cols && gr < rows) {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned a = tile[8 * v + 2 * j][c >> 1], b = tile[8 * v + 2 * j + 1][c >> 1];
                o[j] = (c & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
            }
            *reinterpret_cast<u32x4*>(dst + (int64_t)gc * ld_dst + gr) = o;
        }
    }
}

extern "C" int halva_transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, void* stream) {
    HALVA_CHECK_ARG(src && dst, "transpose_bf16: null pointer");
    HALVA_CHECK_ARG(rows >= 0 && cols >= 0 && rows % 8 == 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ld_src >= cols && ld_dst >= rows,
                    "transpose_bf16: rows, cols and both row strides must be multiples of 8 (16-byte accesses); got %d x %d, strides %lld / %lld", rows,
                    cols, (long long)ld_src, (long long)ld_dst);
    HALVA_CHECK_ARG(((size_t)src | (size_t)dst) % 16 == 0, "transpose_bf16: 16-byte aligned pointers");
    if (rows == 0 || cols == 0) return HALVA_OK;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src,
                       (bf16_t*)dst, ld_dst, rows, cols);
    HALVA_CHECK_LAUNCH("transpose_bf16");
    return HALVA_OK;
}

extern "C" int halva_colsum(const void* x, float* out, int64_t M, int N, void* stream) {
    HALVA_CHECK_ARG(x && out, "colsum: null pointer");
    if (M <= 0 || N <= 0) return HALVA_OK;
    int splits = (int)((M + 127) / 128);
    if (splits > 64) splits = 64;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, splits), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, M,
                       N);
    HALVA_CHECK_LAUNCH("colsum");
    return HALVA_OK;
}
