// Row kernels of the vision side (CLIP / SigLIP towers and VILA's mlp_downsample projector) on gfx950:
// LayerNorm forward (+ parameter gradients), the 2x2 DownSampleBlock gather.  HBM-bound bf16 work with 16-byte
// accesses; one 64-lane wave per row for the LayerNorm reductions (shuffles only).
#include "common.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxChunks = 16;   // rows up to d = 8192 stay in registers

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

// y = bf16((x - mean) * rstd * w + b), statistics in fp32 over the bf16 row (torch.nn.LayerNorm on bf16 input)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ w,
                                                            const u32x4* __restrict__ b, u32x4* __restrict__ y,
                                                            float* __restrict__ stats, int64_t rows, int nchunk, float eps,
                                                            float inv_d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4* xr = x + row * nchunk;
    u32x4 buf[kMaxChunks];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            buf[i] = xr[c];
            float f[8];
            unpack8(buf[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[j];
        }
    }
    const float mean = wave_sum(s) * inv_d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float f[8];
            unpack8(buf[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += (f[j] - mean) * (f[j] - mean);
        }
    }
    const float r = rsqrtf(wave_sum(ss) * inv_d + eps);
    if (stats && lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = r;
    }
    u32x4* yr = y + row * nchunk;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float f[8], g[8], h[8];
            unpack8(buf[i], f);
            unpack8(w[c], g);
            unpack8(b[c], h);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * r * g[j] + h[j];
            yr[c] = pack8(f);
        }
    }
}

// dw[j] += sum_r dy[r][j] * (x[r][j] - mean[r]) * rstd[r];  db[j] += sum_r dy[r][j]
// grid (ceil(d/256), splits): one column per thread, one slab of rows per block, one atomic pair per thread.
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                   const float* __restrict__ stats, float* __restrict__ dw,
                                                                   float* __restrict__ db, int64_t rows, int d) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t lo = blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
    float sw = 0.f, sb = 0.f;
    for (int64_t r = lo; r < hi; ++r) {
        const float g = bf16_to_f32(dy[r * d + j]);
        const float xh = (bf16_to_f32(x[r * d + j]) - stats[2 * r]) * stats[2 * r + 1];
        sw += g * xh;
        sb += g;
    }
    atomicAdd(dw + j, sw);
    atomicAdd(db + j, sb);
}

// DownSampleBlock (vila base_projector.py:33-54): x [n, g*g, c] -> out [n, G*G, 4c], G = ceil(g/2);
// out[n, b2*G + a2, (f*2 + e)*c + ch] = x[n, (2*a2 + f)*g + (2*b2 + e), ch], zero where the padded row/col is read.
__global__ __launch_bounds__(256) void downsample2x2_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, int g, int G,
                                                            int cchunks, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % cchunks);
        int64_t t = i / cchunks;
        const int fe = (int)(t & 3);
        t >>= 2;
        const int a2 = (int)(t % G);
        t /= G;
        const int b2 = (int)(t % G);
        const int64_t n = t / G;
        const int r = 2 * a2 + (fe >> 1), c = 2 * b2 + (fe & 1);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < g && c < g) v = x[((n * g + r) * g + c) * cchunks + ch];
        out[i] = v;
    }
}

}  // namespace

extern "C" int halva_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* stats, int64_t rows, int d,
                                   float eps, void* stream) {
    HALVA_CHECK_ARG(x && w && b && y, "layernorm_fwd: null pointer");
    HALVA_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 8 * 64 * kMaxChunks, "layernorm_fwd: d=%d must be a multiple of 8 and <= %d", d,
                    8 * 64 * kMaxChunks);
    if (rows <= 0) return HALVA_OK;
    const unsigned grid = (unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(64 * kWavesPerBlock), 0, (hipStream_t)stream, (const u32x4*)x,
                       (const u32x4*)w, (const u32x4*)b, (u32x4*)y, stats, rows, d / 8, eps, 1.0f / (float)d);
    HALVA_CHECK_LAUNCH("layernorm_fwd");
    return HALVA_OK;
}

extern "C" int halva_layernorm_bwd_params(const void* dy, const void* x, const float* stats, float* dw, float* db, int64_t rows,
                                          int d, void* stream) {
    HALVA_CHECK_ARG(dy && x && stats && dw && db, "layernorm_bwd_params: null pointer");
    if (rows <= 0 || d <= 0) return HALVA_OK;
    int splits = (int)((rows + 63) / 64);
    if (splits > 128) splits = 128;
    hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3((d + 255) / 256, splits), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, (const bf16_t*)x, stats, dw, db, rows, d);
    HALVA_CHECK_LAUNCH("layernorm_bwd_params");
    return HALVA_OK;
}

extern "C" int halva_downsample2x2(const void* x, void* out, int n, int g, int c, void* stream) {
    HALVA_CHECK_ARG(x && out, "downsample2x2: null pointer");
    HALVA_CHECK_ARG(n > 0 && g > 0 && c > 0 && c % 8 == 0, "downsample2x2: bad sizes n=%d g=%d c=%d (c must be a multiple of 8)", n,
                    g, c);
    const int G = (g + 1) / 2;
    const int64_t total = (int64_t)n * G * G * 4 * (c / 8);
    int64_t grid = (total + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(downsample2x2_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (u32x4*)out,
                       g, G, c / 8, total);
    HALVA_CHECK_LAUNCH("downsample2x2");
    return HALVA_OK;
}

// ---------------------------------------------------------------------------------------------------
// Image preprocessing: Pillow's 8-bit bicubic resample + the HF processor's crop / rescale / normalise, for
// a batch of decoded uint8 images.  Two passes like ImagingResample: horizontal into a uint8 scratch (only the rows the
// vertical pass reads), vertical fused with the centre crop and the per-channel 256-entry normalise table.  The
// expand2square canvas is virtual (pixels outside the pasted image read the fill colour).  Integer arithmetic is Pillow's
// (22-bit fixed-point weights from the host, int32 accumulate with +2^21, shift, clip) so the uint8 results are bit-exact.
// ---------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int clip8(int v) {
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void image_resample_h_kernel(const uint8_t* __restrict__ src, const HalvaImageDesc* __restrict__ descs,
                                                               const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                                               uint8_t* __restrict__ tmp) {
    const HalvaImageDesc d = descs[blockIdx.y];
    const int64_t total = (int64_t)d.tmp_rows * d.out_w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xx = (int)(i % d.out_w), row = (int)(i / d.out_w);
        const int cy = d.row0 + row - d.pad_y;                 // row inside the decoded image (may fall in the padding)
        const bool row_in = cy >= 0 && cy < d.src_h;
        const uint8_t* srow = src + d.src_off + (int64_t)(row_in ? cy : 0) * d.src_w * 3;
        int o0, o1, o2;
        if (d.need_h) {
            const int xmin = bounds[d.bh_off + 2 * xx], n = bounds[d.bh_off + 2 * xx + 1];
            const int32_t* k = coef + d.kh_off + (int64_t)xx * d.ksize_h;
            int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
            for (int j = 0; j < n; ++j) {
                const int cx = xmin + j - d.pad_x;
                int p0 = d.bg[0], p1 = d.bg[1], p2 = d.bg[2];
                if (row_in && cx >= 0 && cx < d.src_w) {
                    p0 = srow[3 * cx];
                    p1 = srow[3 * cx + 1];
                    p2 = srow[3 * cx + 2];
                }
                a0 += p0 * k[j];
                a1 += p1 * k[j];
                a2 += p2 * k[j];
            }
            o0 = clip8(a0), o1 = clip8(a1), o2 = clip8(a2);
        } else {
            const int cx = xx - d.pad_x;
            o0 = d.bg[0], o1 = d.bg[1], o2 = d.bg[2];
            if (row_in && cx >= 0 && cx < d.src_w) {
                o0 = srow[3 * cx];
                o1 = srow[3 * cx + 1];
                o2 = srow[3 * cx + 2];
            }
        }
        uint8_t* t = tmp + d.tmp_off + ((int64_t)row * d.out_w + xx) * 3;
        t[0] = (uint8_t)o0;
        t[1] = (uint8_t)o1;
        t[2] = (uint8_t)o2;
    }
}

template <bool BF16OUT>
__global__ __launch_bounds__(256) void image_resample_v_kernel(const uint8_t* __restrict__ tmp, const HalvaImageDesc* __restrict__ descs,
                                                               const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                                               const float* __restrict__ lut, void* __restrict__ out, int crop_h,
                                                               int crop_w) {
    __shared__ float s_lut[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const HalvaImageDesc d = descs[blockIdx.y];
    const int64_t plane = (int64_t)crop_h * crop_w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % crop_w), y = (int)(i / crop_w);
        const int rx = x + d.crop_x, ry = y + d.crop_y;
        const uint8_t* t = tmp + d.tmp_off;
        int o0, o1, o2;
        if (d.need_v) {
            const int ymin = bounds[d.bv_off + 2 * ry], n = bounds[d.bv_off + 2 * ry + 1];
            const int32_t* k = coef + d.kv_off + (int64_t)ry * d.ksize_v;
            int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
            for (int j = 0; j < n; ++j) {
                const uint8_t* px = t + ((int64_t)(ymin + j) * d.out_w + rx) * 3;
                a0 += px[0] * k[j];
                a1 += px[1] * k[j];
                a2 += px[2] * k[j];
            }
            o0 = clip8(a0), o1 = clip8(a1), o2 = clip8(a2);
        } else {
            const uint8_t* px = t + ((int64_t)ry * d.out_w + rx) * 3;
            o0 = px[0], o1 = px[1], o2 = px[2];
        }
        const int64_t base = (int64_t)blockIdx.y * 3 * plane + i;
        const float f0 = s_lut[o0], f1 = s_lut[256 + o1], f2 = s_lut[512 + o2];
        if (BF16OUT) {
            bf16_t* o = (bf16_t*)out;
            o[base] = f32_to_bf16(f0);
            o[base + plane] = f32_to_bf16(f1);
            o[base + 2 * plane] = f32_to_bf16(f2);
        } else {
            float* o = (float*)out;
            o[base] = f0;
            o[base + plane] = f1;
            o[base + 2 * plane] = f2;
        }
    }
}

}  // namespace

extern "C" int halva_image_preprocess(const void* src_pack, const HalvaImageDesc* descs, const int32_t* coef, const int32_t* bounds,
                                      const float* lut, void* tmp, void* out, int n_images, int max_tmp_pixels, int crop_h,
                                      int crop_w, halva_dtype out_dtype, void* stream) {
    HALVA_CHECK_ARG(src_pack && descs && coef && bounds && lut && tmp && out, "image_preprocess: null pointer");
    HALVA_CHECK_ARG(n_images > 0 && crop_h > 0 && crop_w > 0 && max_tmp_pixels > 0, "image_preprocess: bad sizes");
    HALVA_CHECK_ARG(out_dtype == HALVA_BF16 || out_dtype == HALVA_F32, "image_preprocess: bad output dtype");
    int gx = (max_tmp_pixels + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(image_resample_h_kernel, dim3(gx, n_images), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src_pack, descs,
                       coef, bounds, (uint8_t*)tmp);
    HALVA_CHECK_LAUNCH("image_resample_h");
    int gy = (crop_h * crop_w + 255) / 256;
    if (gy > 4096) gy = 4096;
    if (out_dtype == HALVA_BF16)
        hipLaunchKernelGGL(image_resample_v_kernel<true>, dim3(gy, n_images), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)tmp, descs,
                           coef, bounds, lut, out, crop_h, crop_w);
    else
        hipLaunchKernelGGL(image_resample_v_kernel<false>, dim3(gy, n_images), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)tmp, descs,
                           coef, bounds, lut, out, crop_h, crop_w);
    HALVA_CHECK_LAUNCH("image_resample_v");
    return HALVA_OK;
}
