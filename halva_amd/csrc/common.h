// Shared device/host helpers for libhalva_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/halva_hip.h"

typedef unsigned short bf16_t;   // raw bf16 bits
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

void halva_set_error(const char* fmt, ...);

#define HALVA_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            halva_set_error(__VA_ARGS__);     \
            return HALVA_ERR_INVALID_ARG;     \
        }                                     \
    } while (0)

#define HALVA_CHECK_LAUNCH(name)                                                  \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            halva_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return HALVA_ERR_LAUNCH;                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// one v_cvt_pk_bf16_f32 (the scalar form above only sometimes folds into it)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// One RoPE pair, rope_qk_kernel's arithmetic (rowops.hip) - shared with the attention backward's store epilogues (sdpa.hip), which apply
// the INVERSE rotation (s = -sin) to the freshly rounded dq / dk rows: x1, x2 = elements d and d + D/2 of a head row, already bf16 values.
template <class T>      // float, or a pair of floats (ext_vector_type(2): v_pk_mul_f32 / v_pk_fma_f32 - half the instructions, the same values)
__device__ __forceinline__ void rope_pair(T x1, T x2, T c, T s, T& y1, T& y2) {
    y1 = x1 * c - x2 * s;
    y2 = x2 * c + x1 * s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blocks of NW waves; `red` is NW floats of LDS. Every thread gets the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}
