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
}  // namespace

extern "C" int halva_probe_layouts(int32_t* out, int n, void* stream) {
    HALVA_CHECK_ARG(out && n >= 256 + 1024, "probe_layouts: need %d ints", 256 + 1024);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    HALVA_CHECK_LAUNCH("probe_layouts");
    return HALVA_OK;
}
