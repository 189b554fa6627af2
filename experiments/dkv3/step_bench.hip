// Micro-benchmark (round 3): the step body of a ONE-WAVE-PER-SIMD dK/dV kernel - a wave owns 32 keys (K, V fragments and both
// accumulators in registers: 4 waves x 32 keys per workgroup) and runs all four products of a 64-row step itself - isolated from tile
// DMA, barriers and block prologues: Q / dO tiles and the row statistics sit in LDS, the loop is stamped with s_memtime.
// Question: how many cycles per step does a hand-placed stream of this body take (matrix work: 64 MFMAs = 2 048 cycles), against the
// ~3 850 cycles per step and SIMD of the shipped two-role kernel (two waves per SIMD, the same 64 MFMAs per SIMD and step)?
//   MODE 0: plain order (reference for bit-equality)      MODE 1: software-pipelined groups with scheduling fences
// build: hipcc -O3 --offload-arch=gfx950 -o step_bench step_bench.hip ; run: ./step_bench [nsteps] [nwg]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 128, KS = 8, DT = 4, NTILE = 4, TILE_BYTES = 64 * D * 2;

__device__ __forceinline__ int tile_off(int row, int ch) {
    constexpr int SUBROW = (D / 32) * 512;
    return SUBROW * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
__device__ __forceinline__ f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ s16x8 frag_rows(const char* tile, int row0, int ks, int lane) {
    const int r = row0 + (lane & 31);
    return *reinterpret_cast<const s16x8*>(tile + tile_off(r, 2 * ks + (lane >> 5)));
}
__device__ __forceinline__ s16x8 frag_cols(const char* tile, int row0, int col0, int lane) {
    s16x8 out;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = g >> 1;
    const int c = col0 + 16 * (g & 1) + 4 * pp;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int r = row0 + 8 * jj + 4 * h + q;
        const int off = tile_off(r, c >> 3) + (c & 7) * 2;
        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off));
        out[4 * jj + 0] = t[0]; out[4 * jj + 1] = t[1]; out[4 * jj + 2] = t[2]; out[4 * jj + 3] = t[3];
    }
    return out;
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ f32x16 tuple_of(const f32x4 (&s)[4]) {
    f32x16 x;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = s[r >> 2][r & 3];
    return x;
}
#define SLOT() __builtin_amdgcn_sched_barrier(0)
// register-file pins (empty asm): "a" = accumulator file (MFMA-only state: the four dK / dV accumulators, the stationary K / V
// fragments), "v" = vector file (everything the vector unit touches: the score tiles)
#define PIN_A(x) asm volatile("" : "+a"(x))
#define PIN_V(x) asm volatile("" : "+v"(x))

struct State {
    s16x8 kf[KS], vf[KS];
    f32x16 accV[DT], accK[DT];
};

// ---- MODE 0: one sub-tile after the other, nothing overlapped by hand
__device__ __forceinline__ void step_plain(State& st, const char* qt, const char* dot, const float* lse_t, const float* nd_t, char* ds_step,
                                           float sc, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        f32x4 sl[4], sd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sl[j] = *reinterpret_cast<const f32x4*>(lse_t + 32 * sub + 8 * j + 4 * h);
            sd[j] = *reinterpret_cast<const f32x4*>(nd_t + 32 * sub + 8 * j + 4 * h);
        }
        f32x16 x, y = tuple_of(sd);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x = mfma32(frag_rows(qt, 32 * sub, ks, lane), st.kf[ks], x);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) y = mfma32(frag_rows(dot, 32 * sub, ks, lane), st.vf[ks], y);
        u32x4 pb[2], zb[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * i], sc, -sl[i >> 1][(2 * i) & 3]));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(x[2 * i + 1], sc, -sl[i >> 1][(2 * i + 1) & 3]));
            pb[i >> 2][i & 3] = pack_bf16x2(p0, p1);
            zb[i >> 2][i & 3] = pack_bf16x2(p0 * y[2 * i], p1 * y[2 * i + 1]);
        }
        __builtin_nontemporal_store(zb[0], reinterpret_cast<u32x4*>(ds_step + 2048 * sub + lane * 16));
        __builtin_nontemporal_store(zb[1], reinterpret_cast<u32x4*>(ds_step + 2048 * sub + 1024 + lane * 16));
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i)
            st.accV[i % DT] = mfma32(frag_cols(dot, 32 * sub + 16 * (i / DT), 32 * (i % DT), lane), __builtin_bit_cast(s16x8, pb[i / DT]), st.accV[i % DT]);
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i)
            st.accK[i % DT] = mfma32(frag_cols(qt, 32 * sub + 16 * (i / DT), 32 * (i % DT), lane), __builtin_bit_cast(s16x8, zb[i / DT]), st.accK[i % DT]);
    }
}

// ---- MODE 1: eight groups of eight MFMAs; every LDS read is issued one group ahead of its use, the vector work of a sub-tile rides
//      under the chains of the other products; scheduling fences keep the order written here
__device__ __forceinline__ void step_piped(State& st, const char* qt, const char* dot, const float* lse_t, const float* nd_t, char* ds_step,
                                           float sc, int lane) {
    const int h = lane >> 5;
    s16x8 fa[KS], fb[KS], fc[KS], fd[KS];
    f32x4 sl0[4], sd0[4], sl1[4], sd1[4];
    f32x16 x0, y0, x1, y1;
    float p0[16], p1[16];
    u32x4 pb0[2], zb0[2], pb1[2], zb1[2];
    // prologue: row fragments of Q (sub-tile 0) and the statistics of sub-tile 0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sl0[j] = *reinterpret_cast<const f32x4*>(lse_t + 8 * j + 4 * h);
        sd0[j] = *reinterpret_cast<const f32x4*>(nd_t + 8 * j + 4 * h);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fa[ks] = frag_rows(qt, 0, ks, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) x0[r] = 0.f, x1[r] = 0.f;
    y0 = tuple_of(sd0);
    PIN_V(x0); PIN_V(x1); PIN_V(y0);
    SLOT();
    // G1: S0 = Q0 K^T      || fetch dO rows of sub-tile 0
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        x0 = mfma32(fa[ks], st.kf[ks], x0);
        fb[ks] = frag_rows(dot, 0, ks, lane);
        SLOT();
    }
    // G2: dP0 = dO0 V^T    || fetch Q rows of sub-tile 1, statistics of sub-tile 1; P0 = exp2(S0 sc - lse)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        y0 = mfma32(fb[ks], st.vf[ks], y0);
        fa[ks] = frag_rows(qt, 32, ks, lane);
        if (ks < 4) sl1[ks] = *reinterpret_cast<const f32x4*>(lse_t + 32 + 8 * ks + 4 * h);
        else sd1[ks - 4] = *reinterpret_cast<const f32x4*>(nd_t + 32 + 8 * (ks - 4) + 4 * h);
        p0[2 * ks] = __builtin_amdgcn_exp2f(__builtin_fmaf(x0[2 * ks], sc, -sl0[ks >> 1][(2 * ks) & 3]));
        p0[2 * ks + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(x0[2 * ks + 1], sc, -sl0[ks >> 1][(2 * ks + 1) & 3]));
        SLOT();
    }
    y1 = tuple_of(sd1);
    PIN_V(y1); PIN_V(x0); PIN_V(y0);
    SLOT();
    // G3: S1 = Q1 K^T      || fetch dO rows of sub-tile 1; dZ0 = P0 dP0, packs
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        x1 = mfma32(fa[ks], st.kf[ks], x1);
        fb[ks] = frag_rows(dot, 32, ks, lane);
        pb0[ks >> 2][ks & 3] = pack_bf16x2(p0[2 * ks], p0[2 * ks + 1]);
        zb0[ks >> 2][ks & 3] = pack_bf16x2(p0[2 * ks] * y0[2 * ks], p0[2 * ks + 1] * y0[2 * ks + 1]);
        SLOT();
    }
    // G4: dP1 = dO1 V^T    || column fragments of dO (sub-tile 0) for dV; P1; dS0 out
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        y1 = mfma32(fb[ks], st.vf[ks], y1);
        fc[ks] = frag_cols(dot, 16 * (ks / DT), 32 * (ks % DT), lane);
        p1[2 * ks] = __builtin_amdgcn_exp2f(__builtin_fmaf(x1[2 * ks], sc, -sl1[ks >> 1][(2 * ks) & 3]));
        p1[2 * ks + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(x1[2 * ks + 1], sc, -sl1[ks >> 1][(2 * ks + 1) & 3]));
        if (ks == 0) __builtin_nontemporal_store(zb0[0], reinterpret_cast<u32x4*>(ds_step + lane * 16));
        if (ks == 1) __builtin_nontemporal_store(zb0[1], reinterpret_cast<u32x4*>(ds_step + 1024 + lane * 16));
        SLOT();
    }
    PIN_V(x1); PIN_V(y1);
    // G5: dV^T += dO0^T P0 || column fragments of Q (sub-tile 0) for dK; dZ1, packs
#pragma unroll
    for (int i = 0; i < 2 * DT; ++i) {
        st.accV[i % DT] = mfma32(fc[i], __builtin_bit_cast(s16x8, pb0[i / DT]), st.accV[i % DT]);
        fd[i] = frag_cols(qt, 16 * (i / DT), 32 * (i % DT), lane);
        pb1[i >> 2][i & 3] = pack_bf16x2(p1[2 * i], p1[2 * i + 1]);
        zb1[i >> 2][i & 3] = pack_bf16x2(p1[2 * i] * y1[2 * i], p1[2 * i + 1] * y1[2 * i + 1]);
        SLOT();
    }
    // G6: dK^T += Q0^T dZ0 || column fragments of dO (sub-tile 1); dS1 out
#pragma unroll
    for (int i = 0; i < 2 * DT; ++i) {
        st.accK[i % DT] = mfma32(fd[i], __builtin_bit_cast(s16x8, zb0[i / DT]), st.accK[i % DT]);
        fc[i] = frag_cols(dot, 32 + 16 * (i / DT), 32 * (i % DT), lane);
        if (i == 0) __builtin_nontemporal_store(zb1[0], reinterpret_cast<u32x4*>(ds_step + 2048 + lane * 16));
        if (i == 1) __builtin_nontemporal_store(zb1[1], reinterpret_cast<u32x4*>(ds_step + 3072 + lane * 16));
        SLOT();
    }
    // G7: dV^T += dO1^T P1 || column fragments of Q (sub-tile 1)
#pragma unroll
    for (int i = 0; i < 2 * DT; ++i) {
        st.accV[i % DT] = mfma32(fc[i], __builtin_bit_cast(s16x8, pb1[i / DT]), st.accV[i % DT]);
        fd[i] = frag_cols(qt, 32 + 16 * (i / DT), 32 * (i % DT), lane);
        SLOT();
    }
    // G8: dK^T += Q1^T dZ1
#pragma unroll
    for (int i = 0; i < 2 * DT; ++i) {
        st.accK[i % DT] = mfma32(fd[i], __builtin_bit_cast(s16x8, zb1[i / DT]), st.accK[i % DT]);
        SLOT();
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { PIN_A(st.accV[dt]); PIN_A(st.accK[dt]); }
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void step_kernel(
    const bf16_t* q, const bf16_t* dout, const bf16_t* k, const bf16_t* v, const float* lse2, const float* ndelta, float* out, char* ds_out,
    unsigned long long* cycles, int nsteps, float sc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_lds = smem;
    char* do_lds = smem + NTILE * TILE_BYTES;
    float* lse_lds = reinterpret_cast<float*>(smem + 2 * NTILE * TILE_BYTES);
    float* nd_lds = lse_lds + NTILE * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    // fill the tiles (tile_off image) and the statistics: every workgroup the same data
    for (int c = threadIdx.x; c < NTILE * 64 * 16; c += 256) {
        const int t = c / 1024, row = (c % 1024) / 16, ch = c % 16;
        *reinterpret_cast<u32x4*>(q_lds + t * TILE_BYTES + tile_off(row, ch)) = *reinterpret_cast<const u32x4*>(q + ((size_t)(t * 64 + row)) * D + ch * 8);
        *reinterpret_cast<u32x4*>(do_lds + t * TILE_BYTES + tile_off(row, ch)) = *reinterpret_cast<const u32x4*>(dout + ((size_t)(t * 64 + row)) * D + ch * 8);
    }
    for (int c = threadIdx.x; c < NTILE * 64; c += 256) lse_lds[c] = lse2[c], nd_lds[c] = ndelta[c];
    State st;
    const int key = 32 * wave + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        st.kf[ks] = *reinterpret_cast<const s16x8*>(k + (size_t)key * D + 16 * ks + 8 * h);
        st.vf[ks] = *reinterpret_cast<const s16x8*>(v + (size_t)key * D + 16 * ks + 8 * h);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st.accV[dt][r] = 0.f, st.accK[dt][r] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { PIN_A(st.accV[dt]); PIN_A(st.accK[dt]); }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { PIN_A(st.kf[ks]); PIN_A(st.vf[ks]); }
    __syncthreads();
    char* ds_wave = ds_out + ((size_t)blockIdx.x * 4 + wave) * 8 * 4096;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 2) {
        // lane parts of the LDS addresses (see gen_step_asm.py): row reads (ks even; odd = ^32), transposed reads (jj = 0; jj = 1 = ^32, +2048), statistics
        const int r = lane & 31;
        const int rowrel = 2048 * (r >> 3) + 64 * (r & 7) + 16 * (h ^ ((r >> 2) & 3));
        const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, h2 = g >> 1;
        const int colrel = 64 * (4 * h2 + q4) + 16 * ((2 * (g & 1) + (pp >> 1)) ^ h2) + 8 * (pp & 1);
        const int statrel = 16 * h;
        const unsigned long long ds_base = __builtin_amdgcn_readfirstlane((unsigned)((size_t)ds_wave & 0xffffffffu)) |
                                           ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((size_t)ds_wave >> 32)) << 32);
        u32x4 kq[KS], vq[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kq[ks] = __builtin_bit_cast(u32x4, st.kf[ks]), vq[ks] = __builtin_bit_cast(u32x4, st.vf[ks]);
        asm volatile(
#include "step_asm.inc"
            : "+a"(st.accV[0]), "+a"(st.accV[1]), "+a"(st.accV[2]), "+a"(st.accV[3]), "+a"(st.accK[0]), "+a"(st.accK[1]), "+a"(st.accK[2]), "+a"(st.accK[3])
            : "a"(kq[0]), "a"(kq[1]), "a"(kq[2]), "a"(kq[3]), "a"(kq[4]), "a"(kq[5]), "a"(kq[6]), "a"(kq[7]),
              "a"(vq[0]), "a"(vq[1]), "a"(vq[2]), "a"(vq[3]), "a"(vq[4]), "a"(vq[5]), "a"(vq[6]), "a"(vq[7]),
              "v"(rowrel), "v"(colrel), "v"(statrel), "s"(ds_base), "s"(sc), "s"(nsteps)
            :
#include "step_asm_clobbers.inc"
        );
    } else
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        const int ti = t & (NTILE - 1);
        if (MODE == 0) step_plain(st, q_lds + ti * TILE_BYTES, do_lds + ti * TILE_BYTES, lse_lds + ti * 64, nd_lds + ti * 64, ds_wave + (t & 7) * 4096, sc, lane);
        else step_piped(st, q_lds + ti * TILE_BYTES, do_lds + ti * TILE_BYTES, lse_lds + ti * 64, nd_lds + ti * 64, ds_wave + (t & 7) * 4096, sc, lane);
    }
    asm volatile("" : "+v"(st.accV[0]), "+v"(st.accK[DT - 1]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = (t1 - t0);
    // accumulators out: [wg][wave][2][DT][16][64]
    float* o = out + ((size_t)blockIdx.x * 4 + wave) * 2 * DT * 16 * 64;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[(dt * 16 + r) * 64 + lane] = st.accV[dt][r];
            o[((DT + dt) * 16 + r) * 64 + lane] = st.accK[dt][r];
        }
}

static bf16_t f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (bf16_t)(u >> 16);
}
static float bf2f(bf16_t b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 64, nwg = argc > 2 ? atoi(argv[2]) : 512;
    const float sc = 1.4426950408889634f / sqrtf((float)D);
    std::vector<bf16_t> hq(NTILE * 64 * D), hdo(NTILE * 64 * D), hk(128 * D), hv(128 * D);
    std::vector<float> hl(NTILE * 64), hn(NTILE * 64);
    srand(1);
    auto rnd = [] { float s = 0; for (int i = 0; i < 6; ++i) s += rand() / (float)RAND_MAX; return (s - 3.f) * 1.41f; };
    for (auto& x : hq) x = f2bf(rnd());
    for (auto& x : hdo) x = f2bf(rnd());
    for (auto& x : hk) x = f2bf(rnd());
    for (auto& x : hv) x = f2bf(rnd());
    for (auto& x : hl) x = 4.f + rnd();
    for (auto& x : hn) x = 0.1f * rnd();
    bf16_t *q, *dO, *k, *v; float *l, *n, *out0, *out1; char* ds; unsigned long long* cyc;
    const size_t outn = (size_t)nwg * 4 * 2 * DT * 16 * 64;
    hipMalloc(&q, hq.size() * 2); hipMalloc(&dO, hdo.size() * 2); hipMalloc(&k, hk.size() * 2); hipMalloc(&v, hv.size() * 2);
    hipMalloc(&l, hl.size() * 4); hipMalloc(&n, hn.size() * 4); hipMalloc(&out0, outn * 4); hipMalloc(&out1, outn * 4);
    hipMalloc(&ds, (size_t)nwg * 4 * 8 * 4096); hipMalloc(&cyc, nwg * 4 * 8);
    hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dO, hdo.data(), hdo.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), hv.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(l, hl.data(), hl.size() * 4, hipMemcpyHostToDevice); hipMemcpy(n, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = 2 * NTILE * TILE_BYTES + 2 * NTILE * 64 * 4;
    hipFuncSetAttribute((const void*)step_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)step_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)step_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<unsigned long long> hc(nwg * 4);
    float* out2; hipMalloc(&out2, outn * 4);
    for (int mode = 0; mode < 3; ++mode) {
        float* out = mode == 0 ? out0 : mode == 1 ? out1 : out2;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(step_kernel<0>, dim3(nwg), dim3(256), lds, 0, q, dO, k, v, l, n, out, ds, cyc, nsteps, sc);
            else if (mode == 1) hipLaunchKernelGGL(step_kernel<1>, dim3(nwg), dim3(256), lds, 0, q, dO, k, v, l, n, out, ds, cyc, nsteps, sc);
            else hipLaunchKernelGGL(step_kernel<2>, dim3(nwg), dim3(256), lds, 0, q, dO, k, v, l, n, out, ds, cyc, nsteps, sc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto c : hc) s += (double)c;
            const double flop = (double)nwg * 4 * nsteps * 64 * 2.0 * 32 * 32 * 16;
            printf("mode %d rep %d: %.3f ms, %.0f cycles per step per wave (matrix work 2048), %.1f TFLOP/s, err %s\n", mode, rep, ms,
                   s / hc.size() / nsteps, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
        }
    }
    std::vector<float> h0(outn), h1(outn);
    hipMemcpy(h0.data(), out0, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), out1, outn * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; double mx = 0;
    for (size_t i = 0; i < outn; ++i) { if (memcmp(&h0[i], &h1[i], 4)) ++diff; mx = fmax(mx, fabs(h0[i])); }
    printf("mode 1 vs mode 0: %zu of %zu accumulator values differ bitwise (max |value| %.3f)\n", diff, outn, mx);
    {
        std::vector<float> h2(outn);
        hipMemcpy(h2.data(), out2, outn * 4, hipMemcpyDeviceToHost);
        size_t d2 = 0; double md = 0;
        for (size_t i = 0; i < outn; ++i) { if (memcmp(&h0[i], &h2[i], 4)) { ++d2; md = fmax(md, fabs(h0[i] - h2[i])); } }
        printf("mode 2 (asm) vs mode 0: %zu of %zu accumulator values differ bitwise (max |diff| %.3e)\n", d2, outn, md);
    }
    // CPU check of wave 0 of workgroup 0 (keys 0..31): dV^T[d][key] and dK^T[d][key] accumulated over the steps
    {
        std::vector<double> dv(D * 32, 0.0), dk(D * 32, 0.0);
        for (int t = 0; t < nsteps; ++t) {
            const int ti = t & (NTILE - 1);
            for (int qi = 0; qi < 64; ++qi)
                for (int key = 0; key < 32; ++key) {
                    float s = 0.f, dp = hn[ti * 64 + qi];
                    for (int d = 0; d < D; ++d) {
                        s += bf2f(hq[(ti * 64 + qi) * D + d]) * bf2f(hk[key * D + d]);
                        dp += bf2f(hdo[(ti * 64 + qi) * D + d]) * bf2f(hv[key * D + d]);
                    }
                    const float p = exp2f(s * sc - hl[ti * 64 + qi]);
                    const float pb = bf2f(f2bf(p)), zb = bf2f(f2bf(p * dp));
                    for (int d = 0; d < D; ++d) {
                        dv[d * 32 + key] += (double)bf2f(hdo[(ti * 64 + qi) * D + d]) * pb;
                        dk[d * 32 + key] += (double)bf2f(hq[(ti * 64 + qi) * D + d]) * zb;
                    }
                }
        }
        // accumulator layout: acc[dt][r] on lane: row (d) = 32 dt + (r&3) + 8 (r>>2) + 4 h, column (key) = lane & 31
        double ev = 0, ek = 0, nv = 0, nk = 0;
        for (int dt = 0; dt < DT; ++dt)
            for (int r = 0; r < 16; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), key = lane & 31;
                    const double gv = h0[(size_t)(dt * 16 + r) * 64 + lane], gk = h0[(size_t)((DT + dt) * 16 + r) * 64 + lane];
                    ev += (gv - dv[d * 32 + key]) * (gv - dv[d * 32 + key]); nv += dv[d * 32 + key] * dv[d * 32 + key];
                    ek += (gk - dk[d * 32 + key]) * (gk - dk[d * 32 + key]); nk += dk[d * 32 + key] * dk[d * 32 + key];
                }
        printf("mode 0 vs CPU (wave 0): rel err dV %.2e dK %.2e\n", sqrt(ev / nv), sqrt(ek / nk));
    }
    return 0;
}
