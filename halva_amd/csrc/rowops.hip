// HBM-bound row kernels of the Llama decoder on gfx950: RMSNorm fwd/bwd, RoPE (in place on packed qkv),
// SwiGLU fwd/bwd and the splice row gather.  All bf16 I/O with 16-byte (8 x bf16) accesses per lane,
// one 64-lane wave per row for the reductions (shuffle only, no LDS, no barrier).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxChunks = 16;   // 16 chunks x 64 lanes x 8 elements = rows up to d = 8192 stay in registers
// The row buffers are sized by a template parameter (10 chunks for d <= 5120, else 16): with the 16-chunk buffers the backward
// needs 178 registers = 2 waves per SIMD, too few rows in flight to cover the HBM latency of its load -> reduce -> store chain
// (10 chunks: 122 registers, 4 waves; an 8-chunk instantiation makes hipcc cache the unpacked floats - 256 registers - so d = 4096
// runs the 10-chunk one with two predicated-off iterations).
#define RMSNORM_DISPATCH(d, CALL)          \
    do {                                   \
        if ((d) <= 5120) { CALL(10); }     \
        else { CALL(16); }                 \
    } while (0)

__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_lo(v[i]);
        f[2 * i + 1] = bf16_hi(v[i]);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return v;
}
// bytes that are read once / written once by a streaming row kernel: nontemporal (they would only push the GEMMs' operands out of the caches)
__device__ __forceinline__ u32x4 stream_load(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(u32x4* p, const u32x4& v) { __builtin_nontemporal_store(v, p); }

// ---------------------------------------------------------------------------------------------------
// RMSNorm (modelling_llama.py:65-70): y = bf16(w * x * rsqrt(mean(x^2) + eps)), one rounding.  (The reference module rounds
// x * rstd to the input dtype before multiplying by w; on the fp32 CPU path that parity is measured against that is a no-op,
// and emulating the bf16 double rounding moved the fixture loss by 1.2e-3 - more than every other bf16 effect together.)
// ---------------------------------------------------------------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ w,
                                                          u32x4* __restrict__ y, float* __restrict__ rstd, int64_t rows,
                                                          int nchunk, int64_t ldy_chunks, float eps, float inv_d, int module_rounding,
                                                          u32x4* __restrict__ x_copy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4* xr = x + row * nchunk;
    u32x4 buf[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            buf[i] = stream_load(xr + c);
            if (x_copy) stream_store(x_copy + row * nchunk + c, buf[i]);      // the residual stream's own buffer (see halva_rmsnorm_fwd_fork_ld)
            float f[8];
            unpack8(buf[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss * inv_d + eps);
    if (lane == 0) rstd[row] = r;
    u32x4* yr = y + row * ldy_chunks;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float f[8], g[8];
            unpack8(buf[i], f);
            unpack8(w[c], g);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = g[j] * (module_rounding ? bf16_round(f[j] * r) : f[j] * r);
            stream_store(yr + c, pack8(f));
        }
    }
}

// dx = r * (g - n * mean(g * n)),  g = dy * w,  n = x * r
template <int MAXC, bool HAS_RES>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ x,
                                                          const u32x4* __restrict__ w, const float* __restrict__ rstd,
                                                          const u32x4* dres, u32x4* dx, int64_t rows, int nchunk,
                                                          int64_t lddy_chunks, float inv_d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4* xr = x + row * nchunk;
    const u32x4* dyr = dy + row * lddy_chunks;
    const float r = rstd[row];
    u32x4 bx[MAXC], bg[MAXC];   // bg holds g = dy * w packed back as two-halves? keep dy, recompute g
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            bx[i] = stream_load(xr + c);
            bg[i] = stream_load(dyr + c);
            float fx[8], fd[8], fw[8];
            unpack8(bx[i], fx);
            unpack8(bg[i], fd);
            unpack8(w[c], fw);
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += fd[j] * fw[j] * fx[j] * r;
        }
    }
    dot = wave_sum(dot) * inv_d;
    u32x4* dxr = dx + row * nchunk;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float fx[8], fd[8], fw[8];
            unpack8(bx[i], fx);
            unpack8(bg[i], fd);
            unpack8(w[c], fw);
#pragma unroll
            for (int j = 0; j < 8; ++j) fx[j] = r * (fd[j] * fw[j] - fx[j] * r * dot);
            if (HAS_RES) {      // + the gradient arriving through the residual connection (bf16 sum of two bf16 gradients, as autograd's)
                float fr[8];
                unpack8(stream_load(dres + row * nchunk + c), fr);
#pragma unroll
                for (int j = 0; j < 8; ++j) fx[j] = bf16_round(fx[j]) + fr[j];
            }
            stream_store(dxr + c, pack8(fx));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// RoPE in place on q and k of packed qkv [rows, 3, H, D]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_qk_kernel(u32x4* __restrict__ qkv, const u32x4* __restrict__ cosb,
                                                      const u32x4* __restrict__ sinb, const int32_t* __restrict__ pos,
                                                      const int32_t* __restrict__ br_a, const int32_t* __restrict__ br_b,
                                                      int64_t rows, int T, int H, int chunks_half, float sgn, int64_t total) {
    // one thread = 8 elements of the first half of one head + the matching 8 of the second half
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks_half);
        int64_t t = i / chunks_half;
        const int h = (int)(t % H);
        t /= H;
        const int part = (int)(t & 1);   // 0 = q, 1 = k
        const int64_t row = t >> 1;
        int p = pos ? pos[row] : (int)(row % T);
        if (br_b) {      // branch-packed rows: the rows of branch B continue from the prefix (halva_rope_qk_branch)
            const int64_t sq = row / T;
            if (p >= br_b[sq]) p = br_a[sq] + (p - br_b[sq]);
        }
        const int64_t base = ((row * 3 + part) * H + h) * (2 * chunks_half) + c;
        float x1[8], x2[8], cs[8], sn[8];
        unpack8(stream_load(qkv + base), x1);
        unpack8(stream_load(qkv + base + chunks_half), x2);
        unpack8(cosb[(int64_t)p * chunks_half + c], cs);
        unpack8(sinb[(int64_t)p * chunks_half + c], sn);
        float y1[8], y2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            rope_pair(x1[j], x2[j], cs[j], sn[j] * sgn, y1[j], y2[j]);
        }
        stream_store(qkv + base, pack8(y1));
        stream_store(qkv + base + chunks_half, pack8(y2));
    }
}

// ---------------------------------------------------------------------------------------------------
// SwiGLU on packed gate|up rows
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const u32x4* __restrict__ gu, u32x4* __restrict__ out,
                                                         int chunksF, int64_t ldo_chunks, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / chunksF;
        const int c = (int)(i - row * chunksF);
        float g[8], u[8];
        unpack8(stream_load(gu + row * 2 * chunksF + c), g);
        unpack8(stream_load(gu + row * 2 * chunksF + chunksF + c), u);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = bf16_round(g[j] / (1.f + __expf(-g[j]))) * u[j];
        stream_store(out + row * ldo_chunks + c, pack8(g));
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const u32x4* __restrict__ dout, const u32x4* __restrict__ gu,
                                                         u32x4* __restrict__ dgu, int chunksF, int64_t lddo_chunks, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / chunksF;
        const int c = (int)(i - row * chunksF);
        float g[8], u[8], d[8], dg[8], du[8];
        unpack8(stream_load(gu + row * 2 * chunksF + c), g);
        unpack8(stream_load(gu + row * 2 * chunksF + chunksF + c), u);
        unpack8(stream_load(dout + row * lddo_chunks + c), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sg = 1.f / (1.f + __expf(-g[j]));
            du[j] = d[j] * g[j] * sg;
            dg[j] = d[j] * u[j] * sg * (1.f + g[j] * (1.f - sg));
        }
        stream_store(dgu + row * 2 * chunksF + c, pack8(dg));
        stream_store(dgu + row * 2 * chunksF + chunksF + c, pack8(du));
    }
}

// ---------------------------------------------------------------------------------------------------
// splice: one wave per output row; src >= 0 token embedding, src <= -2 image feature row, src == -1 zero pad
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splice_rows_kernel(const u32x4* __restrict__ embed, const u32x4* __restrict__ feats,
                                                          const int32_t* __restrict__ src, u32x4* __restrict__ out,
                                                          int64_t rows, int nchunk) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int s = src[row];
    const u32x4* from = s >= 0 ? embed + (int64_t)s * nchunk : (s <= -2 ? feats + (int64_t)(-s - 2) * nchunk : nullptr);
    u32x4* to = out + row * nchunk;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (int c = lane; c < nchunk; c += 64) to[c] = from ? from[c] : zero;
}

// One 16-byte chunk per thread, no grid-stride loop (round 5, experiments/rowops_stream: the swiglu_bwd pattern at the step's shape streams at
// 5.3 TB/s that way against 5.0 with the grid capped at 8192 blocks, and at 5.8 with nontemporal accesses on top - stream_load / stream_store);
// the kernels keep their grid-stride loops for what lies beyond the largest launch: HIP rejects a grid whose gridDim.x * blockDim.x exceeds
// 2^32 - 1 threads (hipErrorInvalidConfiguration), so the cap is (2^32 - 1) / block BLOCKS (a cap of 2^31 - 1 blocks would have
// failed to launch where the 8 192-block grids of rounds 1-4 simply looped - unreachable at today's shapes, a 64-GiB tensor).
inline int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 0xffffffffll / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int halva_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int d, float eps,
                                 void* stream) {
    return halva_rmsnorm_fwd_ld(x, w, y, d, rstd, rows, d, eps, stream);
}

extern "C" int halva_rmsnorm_fwd_ld(const void* x, const void* w, void* y, int64_t ldy, float* rstd, int64_t rows, int d,
                                    float eps, void* stream) {
    return halva_rmsnorm_fwd_fork_ld(x, w, y, ldy, rstd, nullptr, rows, d, eps, stream);
}

extern "C" int halva_rmsnorm_fwd_fork_ld(const void* x, const void* w, void* y, int64_t ldy, float* rstd, void* x_copy, int64_t rows,
                                         int d, float eps, void* stream) {
    HALVA_CHECK_ARG(x && w && y && rstd, "rmsnorm_fwd: null pointer");
    HALVA_CHECK_ARG(x_copy != x, "rmsnorm_fwd_fork: x_copy must not alias x");
    HALVA_CHECK_ARG(ldy >= d && ldy % 8 == 0, "rmsnorm_fwd: output row stride %lld must be >= d and a multiple of 8", (long long)ldy);
    HALVA_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 8 * 64 * kMaxChunks, "rmsnorm_fwd: d=%d must be a multiple of 8 and <= %d", d,
                    8 * 64 * kMaxChunks);
    if (rows <= 0) return HALVA_OK;
    const int blocks = (int)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
#define FWD(C)                                                                                                           \
    hipLaunchKernelGGL(rmsnorm_fwd_kernel<C>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (const u32x4*)w, \
                       (u32x4*)y, rstd, rows, d / 8, ldy / 8, eps, 1.f / d, module_rounding, (u32x4*)x_copy)
    // HALVA_RMSNORM_MODULE_ROUNDING=1: round x * rstd to bf16 BEFORE the multiplication with the weight, exactly as the module does on a
    // bf16 device (modelling_llama.py:69-70: `self.weight * hidden_states.to(input_dtype)`), for bf16-vs-bf16 comparisons with the
    // reference's GPU numerics.  Default 0: one rounding of the fp32 value (what the fp32 CPU path parity is defined against computes).
    static const int module_rounding = [] { const char* e = getenv("HALVA_RMSNORM_MODULE_ROUNDING"); return (e && e[0] == '1') ? 1 : 0; }();
    RMSNORM_DISPATCH(d, FWD);
#undef FWD
    HALVA_CHECK_LAUNCH("rmsnorm_fwd");
    return HALVA_OK;
}

extern "C" int halva_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, int64_t rows,
                                 int d, void* stream) {
    return halva_rmsnorm_bwd_ld(dy, d, x, w, rstd, dx, rows, d, stream);
}

extern "C" int halva_rmsnorm_bwd_ld(const void* dy, int64_t lddy, const void* x, const void* w, const float* rstd, void* dx,
                                    int64_t rows, int d, void* stream) {
    return halva_rmsnorm_bwd_res_ld(dy, lddy, x, w, rstd, nullptr, dx, rows, d, stream);
}

extern "C" int halva_rmsnorm_bwd_res_ld(const void* dy, int64_t lddy, const void* x, const void* w, const float* rstd,
                                        const void* dres, void* dx, int64_t rows, int d, void* stream) {
    HALVA_CHECK_ARG(dy && x && w && rstd && dx, "rmsnorm_bwd: null pointer");
    HALVA_CHECK_ARG(lddy >= d && lddy % 8 == 0, "rmsnorm_bwd: dy row stride %lld must be >= d and a multiple of 8", (long long)lddy);
    HALVA_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 8 * 64 * kMaxChunks, "rmsnorm_bwd: d=%d must be a multiple of 8 and <= %d", d,
                    8 * 64 * kMaxChunks);
    if (rows <= 0) return HALVA_OK;
    const int blocks = (int)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
#define BWD(C)                                                                                                                 \
    if (dres)                                                                                                                  \
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<C, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)dy,   \
                           (const u32x4*)x, (const u32x4*)w, rstd, (const u32x4*)dres, (u32x4*)dx, rows, d / 8, lddy / 8, 1.f / d); \
    else                                                                                                                       \
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<C, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)dy,  \
                           (const u32x4*)x, (const u32x4*)w, rstd, (const u32x4*)nullptr, (u32x4*)dx, rows, d / 8, lddy / 8, 1.f / d)
    RMSNORM_DISPATCH(d, BWD);
#undef BWD
    HALVA_CHECK_LAUNCH("rmsnorm_bwd");
    return HALVA_OK;
}

extern "C" int halva_rope_qk(void* qkv, const void* cos, const void* sin, const int32_t* pos, int64_t rows, int T, int H,
                             int D, int max_pos, int inverse, void* stream) {
    HALVA_CHECK_ARG(qkv && cos && sin, "rope_qk: null pointer");
    HALVA_CHECK_ARG(D > 0 && D % 16 == 0, "rope_qk: head_dim=%d must be a multiple of 16", D);
    HALVA_CHECK_ARG(T > 0 && H > 0, "rope_qk: bad T/H");
    HALVA_CHECK_ARG(pos || T <= max_pos, "rope_qk: T=%d exceeds the cos/sin table (%d rows)", T, max_pos);
    if (rows <= 0) return HALVA_OK;
    const int ch = D / 16;
    const int64_t total = rows * 2 * H * ch;
    hipLaunchKernelGGL(rope_qk_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (u32x4*)qkv,
                       (const u32x4*)cos, (const u32x4*)sin, pos, nullptr, nullptr, rows, T, H, ch, inverse ? -1.f : 1.f, total);
    HALVA_CHECK_LAUNCH("rope_qk");
    return HALVA_OK;
}

extern "C" int halva_rope_qk_branch(void* qkv, const void* cos, const void* sin, const int32_t* br_a, const int32_t* br_b, int64_t rows, int T,
                                    int H, int D, int max_pos, int inverse, void* stream) {
    HALVA_CHECK_ARG(qkv && cos && sin, "rope_qk_branch: null pointer");
    HALVA_CHECK_ARG((br_a == nullptr) == (br_b == nullptr), "rope_qk_branch: br_a and br_b go together");
    HALVA_CHECK_ARG(D > 0 && D % 16 == 0, "rope_qk_branch: head_dim=%d must be a multiple of 16", D);
    HALVA_CHECK_ARG(T > 0 && H > 0 && rows % T == 0, "rope_qk_branch: bad T/H/rows");
    HALVA_CHECK_ARG(T <= max_pos, "rope_qk_branch: T=%d exceeds the cos/sin table (%d rows)", T, max_pos);
    if (rows <= 0) return HALVA_OK;
    const int ch = D / 16;
    const int64_t total = rows * 2 * H * ch;
    hipLaunchKernelGGL(rope_qk_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (u32x4*)qkv,
                       (const u32x4*)cos, (const u32x4*)sin, nullptr, br_a, br_b, rows, T, H, ch, inverse ? -1.f : 1.f, total);
    HALVA_CHECK_LAUNCH("rope_qk_branch");
    return HALVA_OK;
}

extern "C" int halva_swiglu_fwd(const void* gu, void* out, int64_t rows, int F, void* stream) {
    return halva_swiglu_fwd_ld(gu, out, F, rows, F, stream);
}

extern "C" int halva_swiglu_fwd_ld(const void* gu, void* out, int64_t ldo, int64_t rows, int F, void* stream) {
    HALVA_CHECK_ARG(gu && out, "swiglu_fwd: null pointer");
    HALVA_CHECK_ARG(ldo >= F && ldo % 8 == 0, "swiglu_fwd: output row stride %lld must be >= F and a multiple of 8", (long long)ldo);
    HALVA_CHECK_ARG(F > 0 && F % 8 == 0, "swiglu_fwd: F=%d must be a multiple of 8", F);
    if (rows <= 0) return HALVA_OK;
    const int64_t total = rows * (F / 8);
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)gu,
                       (u32x4*)out, F / 8, ldo / 8, total);
    HALVA_CHECK_LAUNCH("swiglu_fwd");
    return HALVA_OK;
}

// CLIP's quick_gelu, x * sigmoid(1.702 x), with the roundings of the three bf16 tensor ops it replaces (scale, sigmoid, product)
__global__ __launch_bounds__(256) void quick_gelu_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        unpack8(x[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = bf16_round(1.702f * v[j]);
            v[j] = v[j] * bf16_round(1.f / (1.f + __expf(-t)));
        }
        out[i] = pack8(v);
    }
}

extern "C" int halva_quick_gelu(const void* x, void* out, int64_t n, void* stream) {
    HALVA_CHECK_ARG(x && out, "quick_gelu: null pointer");
    HALVA_CHECK_ARG(n >= 0 && n % 8 == 0 && ((size_t)x | (size_t)out) % 16 == 0, "quick_gelu: n must be a multiple of 8 and the pointers 16-byte aligned");
    if (n == 0) return HALVA_OK;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (u32x4*)out, n / 8);
    HALVA_CHECK_LAUNCH("quick_gelu");
    return HALVA_OK;
}

extern "C" int halva_swiglu_bwd(const void* dout, const void* gu, void* dgu, int64_t rows, int F, void* stream) {
    return halva_swiglu_bwd_ld(dout, F, gu, dgu, rows, F, stream);
}

extern "C" int halva_swiglu_bwd_ld(const void* dout, int64_t lddo, const void* gu, void* dgu, int64_t rows, int F, void* stream) {
    HALVA_CHECK_ARG(dout && gu && dgu, "swiglu_bwd: null pointer");
    HALVA_CHECK_ARG(lddo >= F && lddo % 8 == 0, "swiglu_bwd: dout row stride %lld must be >= F and a multiple of 8", (long long)lddo);
    HALVA_CHECK_ARG(F > 0 && F % 8 == 0, "swiglu_bwd: F=%d must be a multiple of 8", F);
    if (rows <= 0) return HALVA_OK;
    const int64_t total = rows * (F / 8);
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)dout,
                       (const u32x4*)gu, (u32x4*)dgu, F / 8, lddo / 8, total);
    HALVA_CHECK_LAUNCH("swiglu_bwd");
    return HALVA_OK;
}

extern "C" int halva_splice_rows(const void* embed, const void* feats, const int32_t* src, void* out, int64_t rows, int d,
                                 void* stream) {
    HALVA_CHECK_ARG(embed && src && out, "splice_rows: null pointer");
    HALVA_CHECK_ARG(d > 0 && d % 8 == 0, "splice_rows: d=%d must be a multiple of 8", d);
    if (rows <= 0) return HALVA_OK;
    const int blocks = (int)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(splice_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)embed,
                       (const u32x4*)feats, src, (u32x4*)out, rows, d / 8);
    HALVA_CHECK_LAUNCH("splice_rows");
    return HALVA_OK;
}
