// Micro-benchmark (round 3, for the next round's forward kernel): the step body of a ONE-WAVE-PER-SIMD forward attention kernel in the
// orientation sdpa_bwd_dkv3 uses - a wave owns 64 queries (two groups of 32: Q fragments and O^T accumulators in registers), streams 64-key
// K / V tiles from LDS and computes S^T = K Q^T (lane = query: row statistics need no cross-lane work), P = exp2(S^T sc - m_ref), l += P,
// O^T += V^T P - isolated from tile DMA, barriers, masks and the rescale path: the tiles sit in LDS, the loop is stamped with s_memtime.
// Question: how many cycles per 64-key step does a hand-placed stream take (matrix work: 64 MFMAs = 2 048 cycles), against the ~5 100 cycles
// per 64 MFMAs and SIMD of the shipped sdpa_causal_fwd (two 32-row waves per SIMD)?
//   MODE 0: plain HIP, one tile after the other          MODE 2: the loop as one generated asm block (gen_fwd_step.py)
//   MODE 1 / 4: MODE 0 / 2 with every score tested against the lane's visible-key count (causal diagonal at tile nsteps / 2: plain tiles,
//           the diagonal tile and wholly hidden tiles all occur)
//   MODE 3: MODE 2 with the K / V tiles streamed from global memory by LDS-DMA through the ring of four slots (8 requests per wave and
//           step, one counted vmcnt + s_barrier per step), every workgroup its own run of tiles
// build: python3 gen_fwd_step.py && hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o step_bench step_bench.hip ; run: ./step_bench [nsteps] [nwg]
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
    s16x8 qf[2][KS];
    f32x16 acc[2][DT];
    float l[2], mx[2];
};

// one tile, plain order: for each query group the two key halves' scores, their softmax numerators, then the O products
template <bool WITH_O, bool MASKED>
__device__ __forceinline__ void step_plain(State& st, const char* kt, const char* vt, float sc, int lane, int vis0, int vis1) {
    const int hh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        u32x4 pb[2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            f32x16 x;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) x = mfma32(frag_rows(kt, 32 * kh, ks, lane), st.qf[g][ks], x);
            if (MASKED) {      // key (32 kh + row of the register) of this tile is visible to the lane's query iff it is < vis
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * kh + (r & 3) + 8 * (r >> 2) + 4 * hh >= (g ? vis1 : vis0)) x[r] = -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                st.mx[g] = fmaxf(st.mx[g], fmaxf(x[2 * i], x[2 * i + 1]));
                const float p0 = __builtin_amdgcn_exp2f(x[2 * i] * sc), p1 = __builtin_amdgcn_exp2f(x[2 * i + 1] * sc);
                st.l[g] += p0 + p1;
                pb[kh][i >> 2][i & 3] = pack_bf16x2(p0, p1);
            }
        }
        if (WITH_O) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 2 * DT; ++i)
                    st.acc[g][i % DT] = mfma32(frag_cols(vt, 32 * kh + 16 * (i / DT), 32 * (i % DT), lane), __builtin_bit_cast(s16x8, pb[kh][i / DT]), st.acc[g][i % DT]);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void step_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, float* out,
                                                                                              float* stats, unsigned long long* cycles, int nsteps, float sc,
                                                                                              const bf16_t* kstream, const bf16_t* vstream, int stream_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + NTILE * TILE_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    for (int c = threadIdx.x; c < NTILE * 64 * 16; c += 256) {      // the tiles (tile_off image): every workgroup the same data
        const int t = c / 1024, row = (c % 1024) / 16, ch = c % 16;
        *reinterpret_cast<u32x4*>(k_lds + t * TILE_BYTES + tile_off(row, ch)) = *reinterpret_cast<const u32x4*>(k + ((size_t)(t * 64 + row)) * D + ch * 8);
        *reinterpret_cast<u32x4*>(v_lds + t * TILE_BYTES + tile_off(row, ch)) = *reinterpret_cast<const u32x4*>(v + ((size_t)(t * 64 + row)) * D + ch * 8);
    }
    State st;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int query = 64 * wave + 32 * g + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st.qf[g][ks] = *reinterpret_cast<const s16x8*>(q + (size_t)query * D + 16 * ks + 8 * h);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.acc[g][dt][r] = 0.f;
        st.l[g] = 0.f, st.mx[g] = -INFINITY;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) PIN_A(st.acc[g][dt]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) PIN_A(st.qf[g][ks]);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // masked modes: the lane's query 64 Td + 32 g + (lane & 31) sees the keys up to itself (causal), tile t holds the keys 64 t .. 64 t + 63
    const int Td = nsteps / 2, v0 = 64 * Td + (lane & 31) + 1, v1 = v0 + 32;
    if (MODE == 2 || MODE == 3 || MODE == 4) {
        const int r = lane & 31;
        const int rowrel = 2048 * (r >> 3) + 64 * (r & 7) + 16 * (h ^ ((r >> 2) & 3));
        const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, h2 = g16 >> 1;
        const int colrel = 64 * (4 * h2 + q4) + 16 * ((2 * (g16 & 1) + (pp >> 1)) ^ h2) + 8 * (pp & 1);
        u32x4 q0[KS], q1[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) q0[ks] = __builtin_bit_cast(u32x4, st.qf[0][ks]), q1[ks] = __builtin_bit_cast(u32x4, st.qf[1][ks]);
        const int iters = nsteps + 1;
        if (MODE == 3) {
            // lane offset of this lane's 16 bytes of the wave's chunk (1 KiB of the tile image: TileDma for four waves); the wave's pieces lie 16 rows apart
            const int o = 1024 * wave + 16 * lane, band = o / 2048, rem = o % 2048, row = 8 * band + ((rem % 512) >> 6);
            const int ch = 4 * (rem / 512) + (((rem >> 4) & 3) ^ ((row >> 2) & 3));
            const unsigned voff = (unsigned)((row * D + ch * 8) * 2);
            const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave), piece = 16 * D * 2;
            const unsigned long long kp = (unsigned long long)(size_t)(kstream + (size_t)blockIdx.x * stream_tiles * 64 * D);
            const unsigned long long vp = (unsigned long long)(size_t)(vstream + (size_t)blockIdx.x * stream_tiles * 64 * D);
            asm volatile(
#include "fwd_step_dma_asm.inc"
                : "+a"(st.acc[0][0]), "+a"(st.acc[0][1]), "+a"(st.acc[0][2]), "+a"(st.acc[0][3]), "+a"(st.acc[1][0]), "+a"(st.acc[1][1]), "+a"(st.acc[1][2]), "+a"(st.acc[1][3]),
                  "=v"(st.l[0]), "=v"(st.mx[0]), "=v"(st.l[1]), "=v"(st.mx[1])
                : "a"(q0[0]), "a"(q0[1]), "a"(q0[2]), "a"(q0[3]), "a"(q0[4]), "a"(q0[5]), "a"(q0[6]), "a"(q0[7]),
                  "a"(q1[0]), "a"(q1[1]), "a"(q1[2]), "a"(q1[3]), "a"(q1[4]), "a"(q1[5]), "a"(q1[6]), "a"(q1[7]),
                  "v"(rowrel), "v"(colrel), "s"(sc), "s"(iters), "v"(voff), "v"(voff), "s"(wave_u), "s"(piece),
                  "v"((unsigned)kp), "v"((unsigned)(kp >> 32)), "v"((unsigned)vp), "v"((unsigned)(vp >> 32))
                :
#include "fwd_step_dma_asm_clobbers.inc"
            );
        } else if (MODE == 4) {
            const int r0 = v0 + 64 - 4 * h, r1 = v1 + 64 - 4 * h;      // (the loop takes 64 off before the first tile; the lane half's row offset is folded in)
            asm volatile(
#include "fwd_step_masked_asm.inc"
                : "+a"(st.acc[0][0]), "+a"(st.acc[0][1]), "+a"(st.acc[0][2]), "+a"(st.acc[0][3]), "+a"(st.acc[1][0]), "+a"(st.acc[1][1]), "+a"(st.acc[1][2]), "+a"(st.acc[1][3]),
                  "=v"(st.l[0]), "=v"(st.mx[0]), "=v"(st.l[1]), "=v"(st.mx[1])
                : "a"(q0[0]), "a"(q0[1]), "a"(q0[2]), "a"(q0[3]), "a"(q0[4]), "a"(q0[5]), "a"(q0[6]), "a"(q0[7]),
                  "a"(q1[0]), "a"(q1[1]), "a"(q1[2]), "a"(q1[3]), "a"(q1[4]), "a"(q1[5]), "a"(q1[6]), "a"(q1[7]),
                  "v"(rowrel), "v"(colrel), "s"(sc), "s"(iters), "v"(r0), "v"(r1)
                :
#include "fwd_step_masked_asm_clobbers.inc"
            );
        } else
        asm volatile(
#include "fwd_step_asm.inc"
            : "+a"(st.acc[0][0]), "+a"(st.acc[0][1]), "+a"(st.acc[0][2]), "+a"(st.acc[0][3]), "+a"(st.acc[1][0]), "+a"(st.acc[1][1]), "+a"(st.acc[1][2]), "+a"(st.acc[1][3]),
              "=v"(st.l[0]), "=v"(st.mx[0]), "=v"(st.l[1]), "=v"(st.mx[1])
            : "a"(q0[0]), "a"(q0[1]), "a"(q0[2]), "a"(q0[3]), "a"(q0[4]), "a"(q0[5]), "a"(q0[6]), "a"(q0[7]),
              "a"(q1[0]), "a"(q1[1]), "a"(q1[2]), "a"(q1[3]), "a"(q1[4]), "a"(q1[5]), "a"(q1[6]), "a"(q1[7]),
              "v"(rowrel), "v"(colrel), "s"(sc), "s"(iters)
            :
#include "fwd_step_asm_clobbers.inc"
        );
    } else {
#pragma unroll 1
        for (int t = 0; t < nsteps; ++t)
            step_plain<true, MODE == 1>(st, k_lds + (t & (NTILE - 1)) * TILE_BYTES, v_lds + (t & (NTILE - 1)) * TILE_BYTES, sc, lane, v0 - 64 * t, v1 - 64 * t);
        step_plain<false, MODE == 1>(st, k_lds + (nsteps & (NTILE - 1)) * TILE_BYTES, v_lds, sc, lane, v0 - 64 * nsteps, v1 - 64 * nsteps);      // (the asm loop's last iteration scores one more tile)
    }
    asm volatile("" : "+v"(st.acc[0][0]), "+v"(st.acc[1][DT - 1]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = (t1 - t0);
    float* o = out + ((size_t)blockIdx.x * 4 + wave) * 2 * DT * 16 * 64;      // accumulators out: [wg][wave][g][DT][16][64]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((g * DT + dt) * 16 + r) * 64 + lane] = st.acc[g][dt][r];
    float* so = stats + ((size_t)blockIdx.x * 4 + wave) * 4 * 64;
    so[lane] = st.l[0], so[64 + lane] = st.mx[0], so[128 + lane] = st.l[1], so[192 + lane] = st.mx[1];
}

__global__ void fill_stream(const bf16_t* k, const bf16_t* v, bf16_t* ks, bf16_t* vs, int stream_tiles) {
    const size_t dst = (size_t)blockIdx.x * 64 * D, src = (size_t)((blockIdx.x % stream_tiles) & 3) * 64 * D;
    for (int i = threadIdx.x; i < 64 * D / 8; i += 256) {
        reinterpret_cast<u32x4*>(ks + dst)[i] = reinterpret_cast<const u32x4*>(k + src)[i];
        reinterpret_cast<u32x4*>(vs + dst)[i] = reinterpret_cast<const u32x4*>(v + src)[i];
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
    std::vector<bf16_t> hq(256 * D), hk(NTILE * 64 * D), hv(NTILE * 64 * D);
    srand(1);
    auto rnd = [] { float s = 0; for (int i = 0; i < 6; ++i) s += rand() / (float)RAND_MAX; return (s - 3.f) * 1.41f; };
    for (auto& x : hq) x = f2bf(rnd());
    for (auto& x : hk) x = f2bf(rnd());
    for (auto& x : hv) x = f2bf(rnd());
    bf16_t *q, *k, *v; float *out0, *out2, *st0, *st2; unsigned long long* cyc;
    const size_t outn = (size_t)nwg * 4 * 2 * DT * 16 * 64, stn = (size_t)nwg * 4 * 4 * 64;
    hipMalloc(&q, hq.size() * 2); hipMalloc(&k, hk.size() * 2); hipMalloc(&v, hv.size() * 2);
    hipMalloc(&out0, outn * 4); hipMalloc(&out2, outn * 4); hipMalloc(&st0, stn * 4); hipMalloc(&st2, stn * 4); hipMalloc(&cyc, nwg * 4 * 8);
    hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(v, hv.data(), hv.size() * 2, hipMemcpyHostToDevice);
    // MODE 3 streams its tiles from global memory: per workgroup a run of nsteps + 4 tiles, tile t = base tile t & 3 (so that every mode computes the same)
    const int stream_tiles = nsteps + 4;
    bf16_t *ks, *vs;
    hipMalloc(&ks, (size_t)nwg * stream_tiles * 64 * D * 2); hipMalloc(&vs, (size_t)nwg * stream_tiles * 64 * D * 2);
    hipLaunchKernelGGL(fill_stream, dim3((unsigned)(nwg * stream_tiles)), dim3(256), 0, 0, k, v, ks, vs, stream_tiles);
    hipDeviceSynchronize();
    const size_t lds = 2 * NTILE * TILE_BYTES;
    hipFuncSetAttribute((const void*)step_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)step_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)step_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)step_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)step_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float *out1, *out4, *st1, *st4; hipMalloc(&out1, outn * 4); hipMalloc(&out4, outn * 4); hipMalloc(&st1, stn * 4); hipMalloc(&st4, stn * 4);
    float *out3, *st3; hipMalloc(&out3, outn * 4); hipMalloc(&st3, stn * 4);
    std::vector<unsigned long long> hc(nwg * 4);
    for (int mode : {0, 2, 3, 1, 4}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(step_kernel<0>, dim3(nwg), dim3(256), lds, 0, q, k, v, out0, st0, cyc, nsteps, sc, ks, vs, stream_tiles);
            else if (mode == 2) hipLaunchKernelGGL(step_kernel<2>, dim3(nwg), dim3(256), lds, 0, q, k, v, out2, st2, cyc, nsteps, sc, ks, vs, stream_tiles);
            else if (mode == 3) hipLaunchKernelGGL(step_kernel<3>, dim3(nwg), dim3(256), lds, 0, q, k, v, out3, st3, cyc, nsteps, sc, ks, vs, stream_tiles);
            else if (mode == 1) hipLaunchKernelGGL(step_kernel<1>, dim3(nwg), dim3(256), lds, 0, q, k, v, out1, st1, cyc, nsteps, sc, ks, vs, stream_tiles);
            else hipLaunchKernelGGL(step_kernel<4>, dim3(nwg), dim3(256), lds, 0, q, k, v, out4, st4, cyc, nsteps, sc, ks, vs, stream_tiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto c : hc) s += (double)c;
            const double flop = (double)nwg * 4 * nsteps * 64 * 2.0 * 32 * 32 * 16;
            printf("mode %d rep %d: %.3f ms, %.0f cycles per step per wave (matrix work 2048), %.1f TFLOP/s, err %s\n", mode, rep, ms,
                   s / hc.size() / nsteps, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
        }
    }
    std::vector<float> h0(outn), h2(outn), s0(stn), s2(stn);
    hipMemcpy(h0.data(), out0, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), out2, outn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s0.data(), st0, stn * 4, hipMemcpyDeviceToHost); hipMemcpy(s2.data(), st2, stn * 4, hipMemcpyDeviceToHost);
    size_t d2 = 0; double md = 0, mx = 0;
    for (size_t i = 0; i < outn; ++i) { if (memcmp(&h0[i], &h2[i], 4)) { ++d2; md = fmax(md, fabs(h0[i] - h2[i])); } mx = fmax(mx, fabs(h0[i])); }
    printf("mode 2 (asm) vs mode 0: %zu of %zu accumulator values differ bitwise (max |diff| %.3e, max |value| %.3e)\n", d2, outn, md, mx);
    {
        std::vector<float> h3(outn);
        hipMemcpy(h3.data(), out3, outn * 4, hipMemcpyDeviceToHost);
        size_t d3 = 0;
        for (size_t i = 0; i < outn; ++i) if (memcmp(&h0[i], &h3[i], 4)) ++d3;
        printf("mode 3 (asm, tiles streamed by LDS-DMA) vs mode 0: %zu of %zu accumulator values differ bitwise\n", d3, outn);
    }
    {
        std::vector<float> h1(outn), h4(outn);
        hipMemcpy(h1.data(), out1, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(h4.data(), out4, outn * 4, hipMemcpyDeviceToHost);
        size_t d4 = 0, dm = 0;
        for (size_t i = 0; i < outn; ++i) { if (memcmp(&h1[i], &h4[i], 4)) ++d4; if (memcmp(&h1[i], &h0[i], 4)) ++dm; }
        printf("mode 4 (masked asm) vs mode 1 (masked HIP): %zu of %zu accumulator values differ bitwise (the masks change %zu values against mode 0)\n", d4, outn, dm);
    }
    double ml = 0, mm = 0;
    for (size_t i = 0; i < stn; ++i) { const double rel = fabs(s0[i] - s2[i]) / fmax(1e-30, fabs(s0[i])); if ((i / 64) % 2 == 0) ml = fmax(ml, rel); else mm = fmax(mm, fabs(s0[i] - s2[i])); }
    printf("l: max relative difference %.2e; running maximum: max difference %.2e\n", ml, mm);
    {   // CPU check of wave 0 of workgroup 0 (queries 0..63): O^T[d][query] accumulated over the steps
        std::vector<double> o(D * 64, 0.0);
        for (int t = 0; t < nsteps; ++t) {
            const int ti = t & (NTILE - 1);
            for (int qi = 0; qi < 64; ++qi)
                for (int key = 0; key < 64; ++key) {
                    float s = 0.f;
                    for (int d = 0; d < D; ++d) s += bf2f(hk[(ti * 64 + key) * D + d]) * bf2f(hq[qi * D + d]);
                    const float pb = bf2f(f2bf(exp2f(s * sc)));
                    for (int d = 0; d < D; ++d) o[d * 64 + qi] += (double)bf2f(hv[(ti * 64 + key) * D + d]) * pb;
                }
        }
        double e = 0, n = 0;      // accumulator layout: acc[g][dt][r] on lane: row (d) = 32 dt + (r & 3) + 8 (r >> 2) + 4 h, column (query) = 32 g + (lane & 31)
        for (int g = 0; g < 2; ++g)
            for (int dt = 0; dt < DT; ++dt)
                for (int r = 0; r < 16; ++r)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), qi = 32 * g + (lane & 31);
                        const double got = h0[(size_t)((g * DT + dt) * 16 + r) * 64 + lane];
                        e += (got - o[d * 64 + qi]) * (got - o[d * 64 + qi]); n += o[d * 64 + qi] * o[d * 64 + qi];
                    }
        printf("mode 0 vs CPU (wave 0): rel err O %.2e\n", sqrt(e / n));
    }
    return 0;
}
