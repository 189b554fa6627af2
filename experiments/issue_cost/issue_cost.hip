// What ONE wave alone on its SIMD pays per instruction (round 4): 256 threads per workgroup, one workgroup per CU, each wave times a run of 256
// independent instructions of one kind with s_memtime.  Prints cycles per instruction (median over waves).
//   hipcc -O3 --offload-arch=gfx950 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define TIMED(name, body)                                                                                         \
    {                                                                                                             \
        unsigned long long t0, t1;                                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
        asm volatile(REP16(REP16(body)) ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "a0", "a1", "a2", "a3", "memory"); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                               \
        if (lane == 0) out[(blockIdx.x * 4 + wave) * 32 + k] = (unsigned)(t1 - t0);                               \
        ++k;                                                                                                      \
    }

__global__ __launch_bounds__(256) void probe(unsigned* out, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int k = 0;
    __shared__ float lds[4096];
    lds[threadIdx.x] = 0.f;
    TIMED("v_mov", "v_mov_b32 v100, v101\n\t")
    TIMED("v_mul_f32", "v_mul_f32 v100, v101, v102\n\tv_mul_f32 v103, v104, v105\n\t")
    TIMED("v_pk_mul_f32", "v_pk_mul_f32 v[100:101], v[102:103], v[104:105]\n\tv_pk_mul_f32 v[106:107], v[108:109], v[110:111]\n\t")
    TIMED("v_pk_add_f32", "v_pk_add_f32 v[100:101], v[102:103], v[104:105]\n\tv_pk_add_f32 v[106:107], v[108:109], v[110:111]\n\t")
    TIMED("v_pk_fma_f32", "v_pk_fma_f32 v[100:101], v[102:103], v[104:105], v[106:107]\n\tv_pk_fma_f32 v[108:109], v[102:103], v[104:105], v[106:107]\n\t")
    TIMED("v_accvgpr_read", "v_accvgpr_read_b32 v100, a0\n\tv_accvgpr_read_b32 v101, a1\n\t")
    TIMED("v_accvgpr_write", "v_accvgpr_write_b32 a0, v100\n\tv_accvgpr_write_b32 a1, v101\n\t")
    TIMED("v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 v100, v101, v102\n\tv_cvt_pk_bf16_f32 v103, v104, v105\n\t")
    TIMED("v_permlane32_swap", "v_permlane32_swap_b32 v100, v101\n\tv_permlane32_swap_b32 v102, v103\n\t")
    TIMED("v_exp_f32", "v_exp_f32 v100, v101\n\tv_exp_f32 v102, v103\n\t")
    TIMED("v_rcp_f32", "v_rcp_f32 v100, v101\n\tv_rcp_f32 v102, v103\n\t")
    TIMED("v_fma_f32", "v_fma_f32 v100, v101, v102, v103\n\tv_fma_f32 v104, v105, v106, v107\n\t")
    TIMED("v_max3_f32", "v_max3_f32 v100, v101, v102, v103\n\tv_max3_f32 v104, v105, v106, v107\n\t")
    TIMED("s_add_u32", "s_add_u32 s40, s41, 1\n\ts_add_u32 s42, s43, 1\n\t")
    TIMED("s_nop 0", "s_nop 0\n\ts_nop 0\n\t")
    TIMED("v_xor+v_add", "v_xor_b32 v100, 16, v101\n\tv_add_u32 v102, v103, v104\n\t")
    TIMED("v_cmp+cndmask", "v_cmp_lt_i32 vcc, 5, v101\n\tv_cndmask_b32 v102, v103, v104, vcc\n\t")
    if (lane == 0 && wave == 0) sink[blockIdx.x] = lds[5];
}

int main() {
    const int nwg = 256, nk = 17, per[] = {1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2};
    const char* names[] = {"v_mov_b32", "v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_cvt_pk_bf16_f32",
                           "v_permlane32_swap_b32", "v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_max3_f32", "s_add_u32", "s_nop 0", "v_xor_b32 / v_add_u32", "v_cmp / v_cndmask"};
    unsigned* d;
    float* s;
    hipMalloc(&d, nwg * 4 * 32 * 4);
    hipMalloc(&s, nwg * 4);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, d, s);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 4 * 32);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int k = 0; k < nk; ++k) {
        std::vector<unsigned> v;
        for (int w = 0; w < nwg * 4; ++w) v.push_back(h[w * 32 + k]);
        std::sort(v.begin(), v.end());
        printf("%-24s %6.2f cycles per instruction (median of %zu waves; 256 x %d in a row)\n", names[k], v[v.size() / 2] / (256.0 * per[k]), v.size(), per[k]);
    }
    return 0;
}
