// experiments/rowops_stream (round 5): can the HBM-bound row kernels stream faster than the 5.0 - 5.3 TB/s they run at?  The swiglu_bwd access pattern
// (per 16-byte chunk: read dout, read gate, read up, write dgate, write dup; rows x F = 54 848 x 11 008 as in the 7B step) with the launch / access
// variants below; same arithmetic as halva_amd/csrc/rowops.hip:swiglu_bwd_kernel.   hipcc -O3 --offload-arch=gfx950 -o stream_variants stream_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pk(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}
__device__ __forceinline__ void body(const u32x4& d, const u32x4& g, const u32x4& u, u32x4& dg, u32x4& du) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float o[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float dd = k ? hi(d[i]) : lo(d[i]), gg = k ? hi(g[i]) : lo(g[i]), uu = k ? hi(u[i]) : lo(u[i]);
            const float sg = 1.f / (1.f + __expf(-gg));
            o[1][k] = dd * gg * sg;
            o[0][k] = dd * uu * sg * (1.f + gg * (1.f - sg));
        }
        dg[i] = pk(o[0][0], o[0][1]);
        du[i] = pk(o[1][0], o[1][1]);
    }
}
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k_stride(const u32x4* __restrict__ dout, const u32x4* __restrict__ gu, u32x4* __restrict__ dgu, int chunksF, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * UNROLL) {
        u32x4 d[UNROLL], g[UNROLL], u[UNROLL];
        int64_t row[UNROLL]; int c[UNROLL]; bool ok[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const int64_t i = i0 + k * stride;
            ok[k] = i < total;
            row[k] = ok[k] ? i / chunksF : 0; c[k] = ok[k] ? (int)(i - row[k] * chunksF) : 0;
            const u32x4* pd = dout + row[k] * chunksF + c[k];
            const u32x4* pg = gu + row[k] * 2 * chunksF + c[k];
            if (NT) { d[k] = __builtin_nontemporal_load(pd); g[k] = __builtin_nontemporal_load(pg); u[k] = __builtin_nontemporal_load(pg + chunksF); }
            else { d[k] = *pd; g[k] = *pg; u[k] = pg[chunksF]; }
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            u32x4 dg, du;
            body(d[k], g[k], u[k], dg, du);
            if (!ok[k]) continue;
            u32x4* po = dgu + row[k] * 2 * chunksF + c[k];
            if (NT) { __builtin_nontemporal_store(dg, po); __builtin_nontemporal_store(du, po + chunksF); }
            else { *po = dg; po[chunksF] = du; }
        }
    }
}
int main() {
    const int64_t rows = 54848; const int F = 11008, ch = F / 8; const int64_t total = rows * ch;
    u32x4 *dout, *gu, *dgu;
    hipMalloc(&dout, total * 16); hipMalloc(&gu, total * 32); hipMalloc(&dgu, total * 32);
    hipMemset(dout, 0x3c, total * 16); hipMemset(gu, 0x3c, total * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = 5.0 * rows * F * 2;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %.2f TB/s  (%s)\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    };
    const int full = (int)((total + 255) / 256);
    for (int rep = 0; rep < 2; ++rep) {
        run("shipped: grid 8192, plain, 1 chunk", [&] { hipLaunchKernelGGL((k_stride<false, 1>), dim3(8192), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("grid 8192, nontemporal, 1 chunk", [&] { hipLaunchKernelGGL((k_stride<true, 1>), dim3(8192), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("grid 8192, plain, 2 chunks in flight", [&] { hipLaunchKernelGGL((k_stride<false, 2>), dim3(8192), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("grid 8192, nontemporal, 2 chunks", [&] { hipLaunchKernelGGL((k_stride<true, 2>), dim3(8192), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("grid 8192, nontemporal, 4 chunks", [&] { hipLaunchKernelGGL((k_stride<true, 4>), dim3(8192), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("grid 2048 (8 / CU), plain, 4 chunks", [&] { hipLaunchKernelGGL((k_stride<false, 4>), dim3(2048), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("grid 2048 (8 / CU), nontemporal, 4 chunks", [&] { hipLaunchKernelGGL((k_stride<true, 4>), dim3(2048), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("grid 1024 (4 / CU), nontemporal, 4 chunks", [&] { hipLaunchKernelGGL((k_stride<true, 4>), dim3(1024), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("one chunk per thread (no loop), plain", [&] { hipLaunchKernelGGL((k_stride<false, 1>), dim3(full), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
        run("one chunk per thread (no loop), nontemporal", [&] { hipLaunchKernelGGL((k_stride<true, 1>), dim3(full), dim3(256), 0, 0, dout, gu, dgu, ch, total); });
    }
    return 0;
}
