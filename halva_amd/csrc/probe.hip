// Hardware-layout probes (diagnostics for the GPU tests): the MFMA 32x32x16 bf16 operand/accumulator maps and
// the ds_read_b64_tr_b16 gather the attention and GEMM kernels rely on.
#include "common.h"

namespace {
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe_kernel(int32_t* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    // section 0 [0, 256): transposed read, lane l supplies the address of elements 4l .. 4l+3
    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(&lds[lane * 4]));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = t[e];
    // section 1 [256, 256 + 1024): C = A B with small exact integers, operands loaded with the documented maps
    bf16x8 a, b;
    const int r = lane & 31, h = lane >> 5;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        a[j] = (__bf16)(float)(((r * 7 + k * 3) % 5) - 2);    // A[i = r][k]
        b[j] = (__bf16)(float)(((k * 5 + r * 11) % 7) - 3);   // B[k][j = r]
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[256 + lane * 16 + i] = (int)c[i];
}
// In-kernel shader clock: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz reference, so their ratio over a short spin
// is the clock the CU runs at WHILE whatever else is on the chip runs (MI355X_MICROARCH.md, DVFS give-back item 6: board power and sysfs
// pp_dpm_sclk are not the test).  One wave per block; out[4 b + {0, 1, 2, 3}] = {shader cycles, 100 MHz ticks, start tick, XCC id}.
__global__ void clock_probe_kernel(unsigned long long* out, int spin_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < (unsigned long long)spin_ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[4 * blockIdx.x + 0] = c1 - c0;
    out[4 * blockIdx.x + 1] = r1 - r0;
    out[4 * blockIdx.x + 2] = r0;
    out[4 * blockIdx.x + 3] = xcc & 0xf;
}
}  // namespace

extern "C" int halva_clock_probe(uint64_t* out, int blocks, int spin_ticks, void* stream) {
    HALVA_CHECK_ARG(out && blocks > 0 && blocks <= 1024 && spin_ticks > 0 && spin_ticks <= 100000000,
                    "clock_probe: out=%p blocks=%d spin_ticks=%d", (void*)out, blocks, spin_ticks);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, spin_ticks);
    HALVA_CHECK_LAUNCH("clock_probe");
    return HALVA_OK;
}

extern "C" int halva_probe_layouts(int32_t* out, int n, void* stream) {
    HALVA_CHECK_ARG(out && n >= 256 + 1024, "probe_layouts: need %d ints", 256 + 1024);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    HALVA_CHECK_LAUNCH("probe_layouts");
    return HALVA_OK;
}
