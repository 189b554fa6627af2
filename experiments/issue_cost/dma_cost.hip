// What an LDS-DMA request costs the lone wave that issues it (round 4): one workgroup of four waves per CU; each wave times 16 requests of
// 1 KiB (buffer_load_dwordx4 ... offen lds / global_load_lds_dwordx4) from its own 64 KiB of an L2-resident buffer, in several forms:
//   A  M0 rewritten for every request (s_add m0 + s_nop 0), as the kernels do
//   B  M0 set once, every request to the same LDS place (is it the M0 write that costs?)
//   C  M0 set once, immediate offsets 0 / 1024 / 2048 / 3072 (groups of four: LDS and global address both move by the offset)
//   D  plain global_load_dwordx4 into registers (no LDS)
//   E  form A with 16 independent v_fma between two requests (does other work hide the cost?)
//   G  (round 6) form E WITHOUT a lane offset (`off`: every lane the same address - timing only): is it the address register's read that costs?
//   H  (round 6) form E with the lane offset made by the BUFFER DESCRIPTOR (ADD_TID_ENABLE, stride 16: lane l reads base + soffset + 16 l, no
//      vector register involved) + a check that the 1 KiB really lands as it does through `offen` with voff = 16 l
//   hipcc -O3 --offload-arch=gfx950 -o dma_cost dma_cost.hip && ./dma_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1(k)                                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");          \
    asm volatile("s_memtime %0\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=s"(t2)::"memory"); \
    if (lane == 0) out[(blockIdx.x * 4 + wave) * 16 + 2 * (k)] = (unsigned)(t1 - t0), out[(blockIdx.x * 4 + wave) * 16 + 2 * (k) + 1] = (unsigned)(t2 - t0);
#define R4(x) x x x x
__global__ __launch_bounds__(256) void probe(const char* src, unsigned* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + ((size_t)blockIdx.x * 4 + wave) * 65536;
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32));
    d[2] = 65536u;
    d[3] = 0x00020000u;
    const unsigned voff = 16u * lane, lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
    unsigned long long t0, t1, t2;
    // warm the lines (L2) once
    { float a = 0; for (int i = lane * 4; i < 65536; i += 256) a += *(const float*)(base + i); if (a == 12345.f) sink[0] = a; }
    // A
    T0();
    asm volatile("s_mov_b32 s40, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 R4(R4("buffer_load_dwordx4 %0, %1, s40 offen lds\n\ts_add_u32 s40, s40, 1024\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\t"))
                 :: "v"(voff), "s"(d), "s"(lds0) : "s40", "memory");
    T1(0)
    // B
    T0();
    asm volatile("s_mov_b32 s40, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 R4(R4("buffer_load_dwordx4 %0, %1, s40 offen lds\n\ts_add_u32 s40, s40, 1024\n\t"))
                 :: "v"(voff), "s"(d), "s"(lds0) : "s40", "memory");
    T1(1)
    // C
    T0();
    asm volatile("s_mov_b32 s40, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 R4("buffer_load_dwordx4 %0, %1, s40 offen lds\n\tbuffer_load_dwordx4 %0, %1, s40 offen offset:1024 lds\n\t"
                    "buffer_load_dwordx4 %0, %1, s40 offen offset:2048 lds\n\tbuffer_load_dwordx4 %0, %1, s40 offen offset:3072 lds\n\t"
                    "s_add_u32 s40, s40, 4096\n\ts_add_u32 m0, m0, 4096\n\ts_nop 0\n\t")
                 :: "v"(voff), "s"(d), "s"(lds0) : "s40", "memory");
    T1(2)
    // D
    T0();
    asm volatile("s_mov_b32 s40, 0\n\t"
                 R4("buffer_load_dwordx4 v[100:103], %0, %1, s40 offen\n\tbuffer_load_dwordx4 v[104:107], %0, %1, s40 offen offset:1024\n\t"
                    "buffer_load_dwordx4 v[108:111], %0, %1, s40 offen offset:2048\n\tbuffer_load_dwordx4 v[112:115], %0, %1, s40 offen offset:3072\n\t"
                    "s_add_u32 s40, s40, 4096\n\t")
                 :: "v"(voff), "s"(d) : "s40", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "memory");
    T1(3)
    // E
    T0();
    asm volatile("s_mov_b32 s40, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 R4(R4("buffer_load_dwordx4 %0, %1, s40 offen lds\n\ts_add_u32 s40, s40, 1024\n\ts_add_u32 m0, m0, 1024\n\t"
                       R4(R4("v_fma_f32 v100, v101, v102, v103\n\t"))))
                 :: "v"(voff), "s"(d), "s"(lds0) : "s40", "v100", "memory");
    T1(4)
    // G: no lane offset at all
    T0();
    asm volatile("s_mov_b32 s40, 0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 R4(R4("buffer_load_dwordx4 off, %0, s40 lds\n\ts_add_u32 s40, s40, 1024\n\ts_add_u32 m0, m0, 1024\n\t"
                       R4(R4("v_fma_f32 v100, v101, v102, v103\n\t"))))
                 :: "s"(d), "s"(lds0) : "s40", "v100", "memory");
    T1(6)
    // H: the lane offset from the descriptor (word1: stride 16 in bits 16-29; word3: ADD_TID_ENABLE = bit 23, DATA_FORMAT then holds stride[17:14] = 0)
    u32x4 dt = d;
    dt[1] = d[1] | (16u << 16);
    dt[3] = 0x00800000u;
    T0();
    asm volatile("s_mov_b32 s40, 0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 R4(R4("buffer_load_dwordx4 off, %0, s40 lds\n\ts_add_u32 s40, s40, 1024\n\ts_add_u32 m0, m0, 1024\n\t"
                       R4(R4("v_fma_f32 v100, v101, v102, v103\n\t"))))
                 :: "s"(dt), "s"(lds0) : "s40", "v100", "memory");
    T1(7)
    {   // what landed: dword k of lane l's 16 bytes of request i must be source dword 256 i + 4 l + k
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned* mine = (const unsigned*)(smem + wave * 16384);
        const unsigned* want = (const unsigned*)base;
        unsigned bad = 0;
        for (int i = 0; i < 16; ++i)
            for (int k = 0; k < 4; ++k) bad += mine[256 * i + 4 * lane + k] != want[256 * i + 4 * lane + k];
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
        if (lane == 0) out[(blockIdx.x * 4 + wave) * 16 + 12] = bad;
    }
    // F: the 256 v_fma alone
    T0();
    asm volatile(R4(R4(R4(R4("v_fma_f32 v100, v101, v102, v103\n\t")))) ::: "v100", "memory");
    T1(5)
    if (lane == 0 && wave == 0) sink[1] = ((float*)smem)[5];
}
int main() {
    const int nwg = 256;
    char* src; unsigned* d; float* s;
    hipMalloc(&src, (size_t)nwg * 4 * 65536);
    {   // a pattern (round 6: form H checks what lands): dword j holds a hash of j
        std::vector<unsigned> pat((size_t)nwg * 4 * 16384);
        for (size_t j = 0; j < pat.size(); ++j) pat[j] = (unsigned)(j * 2654435761u);
        hipMemcpy(src, pat.data(), pat.size() * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&d, nwg * 4 * 16 * 4); hipMalloc(&s, 64);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 65536, 0, src, d, s);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 4 * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    const char* names[] = {"A: 16 requests, M0 rewritten each time", "B: 16 requests, M0 set once (same LDS place)", "C: 16 requests, M0 per four + immediate offsets",
                           "D: 16 buffer_load_dwordx4 into registers", "E: form A, 16 v_fma between two requests", "F: the 256 v_fma of E alone",
                           "G: form E without a lane offset (off)", "H: form E, lane offset by the descriptor (ADD_TID)"};
    for (int k = 0; k < 8; ++k) {
        std::vector<unsigned> a, b;
        for (int w = 0; w < nwg * 4; ++w) a.push_back(h[w * 16 + 2 * k]), b.push_back(h[w * 16 + 2 * k + 1]);
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("%-52s issued after %6u cycles (%5.1f per request), all landed after %6u\n", names[k], a[a.size() / 2], a[a.size() / 2] / 16.0, b[b.size() / 2]);
    }
    unsigned long long bad = 0;
    for (int w = 0; w < nwg * 4; ++w) bad += h[w * 16 + 12];
    printf("form H: %llu dwords of %d landed somewhere else than with `offen`, voff = 16 lane\n", bad, nwg * 4 * 16 * 256);
    return 0;
}
