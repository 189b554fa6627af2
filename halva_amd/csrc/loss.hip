// Loss kernels of the DPA step on gfx950 (all HBM-bound; one 256-thread workgroup per logits row):
//   token_logp   log p(target) with one online-softmax pass over the row           (halva_trainer.py:406-407)
//   kl_rows      KL(ref || policy) of a row with one fused two-input online pass   (halva_trainer.py:583-588)
//   phrase_sum   masked segmented sum of token log-probs by phrase id              (halva_trainer.py:411-419,556-557)
// Rows are read with 16-byte accesses per lane, reduced with wave shuffles + one LDS exchange per workgroup.
#include "common.h"
#include <atomic>
#include <cstdint>
#include <cstdlib>

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kNW = 4;   // waves per workgroup

// Round 6: the logits rows are streamed - a 0.5 GB chunk is read once per pass and overwritten in place by the gradient - so
// the row kernels' policy applies: nontemporal loads and stores (-DHALVA_LOSS_NT=0: the plain accesses of rounds 1-5, A/B builds).
#ifndef HALVA_LOSS_NT
#define HALVA_LOSS_NT 1
#endif
#if HALVA_LOSS_NT
#define LOSS_LOAD(p) __builtin_nontemporal_load(p)
#define LOSS_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define LOSS_LOAD(p) (*(p))
#define LOSS_STORE(p, v) (*(p) = (v))
#endif

template <typename T>
struct RowIO;
template <>
struct RowIO<bf16_t> {
    static constexpr int W = 8;
    typedef u32x4 Raw;      // one 16-byte chunk as it lies in memory (kl_rows keeps the rows' chunks in LDS between its two passes)
    __device__ static void cvt(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = bf16_lo(v[i]);
            f[2 * i + 1] = bf16_hi(v[i]);
        }
    }
    __device__ static void load(const bf16_t* p, float (&f)[8]) { cvt(LOSS_LOAD(reinterpret_cast<const u32x4*>(p)), f); }
    __device__ static void store(bf16_t* p, const float (&f)[8]) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
        LOSS_STORE(reinterpret_cast<u32x4*>(p), v);
    }
    __device__ static float get(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static void put(bf16_t* p, float f) { *p = f32_to_bf16(f); }
};
template <>
struct RowIO<float> {
    static constexpr int W = 4;
    typedef f32x4 Raw;
    __device__ static void cvt(const f32x4& v, float (&f)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = v[i];
    }
    __device__ static void load(const float* p, float (&f)[4]) { cvt(LOSS_LOAD(reinterpret_cast<const f32x4*>(p)), f); }
    __device__ static void store(float* p, const float (&f)[4]) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = f[i];
        LOSS_STORE(reinterpret_cast<f32x4*>(p), v);
    }
    __device__ static float get(const float* p) { return *p; }
    __device__ static void put(float* p, float f) { *p = f; }
};

// running (max, sum) in the log2 domain
struct MS {
    float m, s;
};
__device__ __forceinline__ void ms_merge(MS& a, float m2, float s2) {
    const float M = fmaxf(a.m, m2);
    const float fa = (a.m == -INFINITY) ? 0.f : exp2f(a.m - M);
    const float fb = (m2 == -INFINITY) ? 0.f : exp2f(m2 - M);
    a.s = a.s * fa + s2 * fb;
    a.m = M;
}
__device__ __forceinline__ MS ms_wave(MS a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(a.m, o, 64), s2 = __shfl_xor(a.s, o, 64);
        ms_merge(a, m2, s2);
    }
    return a;
}

template <typename T>
__device__ __forceinline__ bool row_vec_ok(const T* p, int64_t ld) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld * (int64_t)sizeof(T)) % 16 == 0);
}

// ---------------------------------------------------------------------------------------------------
// KCH > 0 (round 6): every thread asks for ALL of its (<= KCH) 16-byte chunks of the row before it touches the first - sixteen loads in flight per
// thread instead of one or two behind the dependent online-softmax update; same chunks in the same order per thread: the same bits.  The launcher
// picks it when the row fits KCH * 256 chunks (V <= 32 768 in bf16).
template <typename T, int KCH>
__global__ __launch_bounds__(256) void token_logp_fwd_kernel(const T* __restrict__ logits, int64_t ld,
                                                             const int32_t* __restrict__ target, float* __restrict__ logp,
                                                             float* __restrict__ lse, int V) {
    constexpr int W = RowIO<T>::W;
    typedef typename RowIO<T>::Raw Raw;
    __shared__ float red_m[kNW], red_s[kNW];
    const int64_t r = blockIdx.x;
    const T* row = logits + r * ld;
    const int nvec = row_vec_ok(logits, ld) ? V / W : 0;
    MS a{-INFINITY, 0.f};
    auto chunk = [&](const float (&f)[W]) {
        float cm = f[0];
#pragma unroll
        for (int j = 1; j < W; ++j) cm = fmaxf(cm, f[j]);
        cm *= kLog2e;
        const float M = fmaxf(a.m, cm);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < W; ++j) s += exp2f(f[j] * kLog2e - M);
        a.s = a.s * ((a.m == -INFINITY) ? 0.f : exp2f(a.m - M)) + s;
        a.m = M;
    };
    if (KCH > 0) {
        Raw keep[KCH > 0 ? KCH : 1];
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < nvec) keep[i] = LOSS_LOAD(reinterpret_cast<const Raw*>(row + (int64_t)c * W));
        }
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < nvec) {
                float f[W];
                RowIO<T>::cvt(keep[i], f);
                chunk(f);
            }
        }
    } else {
        for (int c = threadIdx.x; c < nvec; c += 256) {
            float f[W];
            RowIO<T>::load(row + (int64_t)c * W, f);
            chunk(f);
        }
    }
    for (int v = nvec * W + threadIdx.x; v < V; v += 256) ms_merge(a, RowIO<T>::get(row + v) * kLog2e, 1.f);
    a = ms_wave(a);
    if ((threadIdx.x & 63) == 0) {
        red_m[threadIdx.x >> 6] = a.m;
        red_s[threadIdx.x >> 6] = a.s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MS t{red_m[0], red_s[0]};
#pragma unroll
        for (int i = 1; i < kNW; ++i) ms_merge(t, red_m[i], red_s[i]);
        const float l = (t.m + log2f(t.s)) * kLn2;
        lse[r] = l;
        logp[r] = RowIO<T>::get(row + target[r]) - l;
    }
}

template <typename T, int KCH>
__global__ __launch_bounds__(256) void token_logp_bwd_kernel(const T* logits, int64_t ld,
                                                             const int32_t* __restrict__ target, const float* __restrict__ lse,
                                                             const float* __restrict__ g, T* dlogits, int V) {
    constexpr int W = RowIO<T>::W;
    typedef typename RowIO<T>::Raw Raw;
    const int64_t r = blockIdx.x;
    const T* row = logits + r * ld;
    T* drow = dlogits + r * ld;
    const float gr = g[r];
    const float l2 = lse[r] * kLog2e;
    const int tgt = target[r];
    const int nvec = (row_vec_ok(logits, ld) && row_vec_ok(dlogits, ld)) ? V / W : 0;
    auto grad = [&](int c, float (&f)[W]) {
#pragma unroll
        for (int j = 0; j < W; ++j) f[j] = gr * ((c * W + j == tgt ? 1.f : 0.f) - exp2f(f[j] * kLog2e - l2));
    };
    if (KCH > 0 && gr != 0.f) {      // (a row without a gradient is not read at all: below)
        Raw keep[KCH > 0 ? KCH : 1];
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < nvec) keep[i] = LOSS_LOAD(reinterpret_cast<const Raw*>(row + (int64_t)c * W));
        }
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < nvec) {
                float f[W];
                RowIO<T>::cvt(keep[i], f);
                grad(c, f);
                RowIO<T>::store(drow + (int64_t)c * W, f);
            }
        }
    } else {
        for (int c = threadIdx.x; c < nvec; c += 256) {
            float f[W];
            if (gr != 0.f) {
                RowIO<T>::load(row + (int64_t)c * W, f);
                grad(c, f);
            } else {
#pragma unroll
                for (int j = 0; j < W; ++j) f[j] = 0.f;
            }
            RowIO<T>::store(drow + (int64_t)c * W, f);
        }
    }
    for (int v = nvec * W + threadIdx.x; v < V; v += 256) {
        const float x = RowIO<T>::get(row + v);
        RowIO<T>::put(drow + v, gr == 0.f ? 0.f : gr * ((v == tgt ? 1.f : 0.f) - exp2f(x * kLog2e - l2)));
    }
}

// ---------------------------------------------------------------------------------------------------
// KL(ref || pol) of one row.  Running state per thread: (m_r, s_r, a_r) for the reference with
// a_r = sum exp2(t_r - m_r) * (z_r - z_p), and (m_p, s_p) for the policy.
// KEEP (round 4): every thread holds its 16-byte chunks of the two rows IN REGISTERS between the statistics pass and the gradient pass (NT = 512
// threads x KCH = 8 chunks x 2 rows = 64 registers, 122 in all: two workgroups per CU; all sixteen loads are issued before the first use), so
// that every logit is fetched from HBM ONCE: the second pass of the 256-thread form is not served by L2 / MALL (5 row passes reach HBM for 3
// algorithmic, profiles/r03_rowops_pmc.json).  tools/bench_kl_rows.py, 8192 rows of 32 000: 362-391 us against 453-457 (4.0-4.3 instead of
// 3.45 TB/s of the three algorithmic passes).  Forms that leave ONE workgroup on a CU were slower than reading twice - nothing is in flight
// while it reduces and stores: the rows in 128 000 B of LDS 561 us, in the registers of 1024 threads (92 each) 481-545; 256 x 16 chunks 385-401.
template <typename T, int NT, bool KEEP, int KCH = 4>
__global__ __launch_bounds__(NT) void kl_rows_kernel(const T* pol, const T* __restrict__ ref, int64_t ld,
                                                     const float* __restrict__ w, float* __restrict__ kl, T* dpol,
                                                     float gscale, int V) {
    constexpr int W = RowIO<T>::W, kNW = NT / 64;
    typedef typename RowIO<T>::Raw Raw;
    Raw keep_r[KEEP ? KCH : 1], keep_p[KEEP ? KCH : 1];      // (the launcher guarantees V / W <= KCH * NT)
    __shared__ float red[5][kNW];
    __shared__ float bc[2];
    const int64_t r = blockIdx.x;
    const T* prow = pol + r * ld;
    const T* rrow = ref + r * ld;
    const float wr = w ? w[r] : 1.f;
    const bool vec = row_vec_ok(pol, ld) && row_vec_ok(ref, ld) && (!dpol || row_vec_ok(dpol, ld));
    const int nvec = vec ? V / W : 0;
    if (wr == 0.f) {   // masked row: contributes nothing and has zero gradient; never read it
        if (threadIdx.x == 0) kl[r] = 0.f;
        if (dpol) {
            T* drow = dpol + r * ld;
            float z[W];
#pragma unroll
            for (int j = 0; j < W; ++j) z[j] = 0.f;
            for (int c = threadIdx.x; c < nvec; c += NT) RowIO<T>::store(drow + (int64_t)c * W, z);
            for (int v = nvec * W + threadIdx.x; v < V; v += NT) RowIO<T>::put(drow + v, 0.f);
        }
        return;
    }
    float mr = -INFINITY, sr = 0.f, ar = 0.f, mp = -INFINITY, sp = 0.f;
    auto step = [&](float zr, float zp) {
        const float tr = zr * kLog2e, tp = zp * kLog2e;
        if (tr > mr) {
            const float f = (mr == -INFINITY) ? 0.f : exp2f(mr - tr);
            sr *= f;
            ar *= f;
            mr = tr;
        }
        const float e = exp2f(tr - mr);
        sr += e;
        ar += e * (zr - zp);
        if (tp > mp) {
            sp *= (mp == -INFINITY) ? 0.f : exp2f(mp - tp);
            mp = tp;
        }
        sp += exp2f(tp - mp);
    };
    if (KEEP) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < nvec) keep_r[i] = LOSS_LOAD(reinterpret_cast<const Raw*>(rrow + (int64_t)c * W)), keep_p[i] = LOSS_LOAD(reinterpret_cast<const Raw*>(prow + (int64_t)c * W));
        }
    }
#pragma unroll
    for (int i = 0; i < (KEEP ? KCH : 1 << 30); ++i) {
        const int c = threadIdx.x + i * NT;
        if (c >= nvec) break;
        float fr[W], fp[W];
        if (KEEP) {
            RowIO<T>::cvt(keep_r[i], fr);
            RowIO<T>::cvt(keep_p[i], fp);
        } else {
            RowIO<T>::load(rrow + (int64_t)c * W, fr);
            RowIO<T>::load(prow + (int64_t)c * W, fp);
        }
        // chunk-wise rescale: one max per chunk keeps the exp count at W + 2 per input
        float cr = fr[0], cp = fp[0];
#pragma unroll
        for (int j = 1; j < W; ++j) {
            cr = fmaxf(cr, fr[j]);
            cp = fmaxf(cp, fp[j]);
        }
        cr *= kLog2e;
        cp *= kLog2e;
        if (cr > mr) {
            const float f = (mr == -INFINITY) ? 0.f : exp2f(mr - cr);
            sr *= f;
            ar *= f;
            mr = cr;
        }
        if (cp > mp) {
            sp *= (mp == -INFINITY) ? 0.f : exp2f(mp - cp);
            mp = cp;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float e = exp2f(fr[j] * kLog2e - mr);
            sr += e;
            ar += e * (fr[j] - fp[j]);
            sp += exp2f(fp[j] * kLog2e - mp);
        }
    }
    for (int v = nvec * W + threadIdx.x; v < V; v += NT) step(RowIO<T>::get(rrow + v), RowIO<T>::get(prow + v));
    // wave then workgroup merge
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float mr2 = __shfl_xor(mr, o, 64), sr2 = __shfl_xor(sr, o, 64), ar2 = __shfl_xor(ar, o, 64);
        const float mp2 = __shfl_xor(mp, o, 64), sp2 = __shfl_xor(sp, o, 64);
        const float Mr = fmaxf(mr, mr2), Mp = fmaxf(mp, mp2);
        const float f1 = (mr == -INFINITY) ? 0.f : exp2f(mr - Mr), f2 = (mr2 == -INFINITY) ? 0.f : exp2f(mr2 - Mr);
        sr = sr * f1 + sr2 * f2;
        ar = ar * f1 + ar2 * f2;
        mr = Mr;
        const float g1 = (mp == -INFINITY) ? 0.f : exp2f(mp - Mp), g2 = (mp2 == -INFINITY) ? 0.f : exp2f(mp2 - Mp);
        sp = sp * g1 + sp2 * g2;
        mp = Mp;
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = mr;
        red[1][wv] = sr;
        red[2][wv] = ar;
        red[3][wv] = mp;
        red[4][wv] = sp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float Mr = red[0][0], Sr = red[1][0], Ar = red[2][0], Mp = red[3][0], Sp = red[4][0];
#pragma unroll
        for (int i = 1; i < kNW; ++i) {
            const float M = fmaxf(Mr, red[0][i]);
            const float f1 = (Mr == -INFINITY) ? 0.f : exp2f(Mr - M), f2 = (red[0][i] == -INFINITY) ? 0.f : exp2f(red[0][i] - M);
            Sr = Sr * f1 + red[1][i] * f2;
            Ar = Ar * f1 + red[2][i] * f2;
            Mr = M;
            const float N = fmaxf(Mp, red[3][i]);
            const float g1 = (Mp == -INFINITY) ? 0.f : exp2f(Mp - N), g2 = (red[3][i] == -INFINITY) ? 0.f : exp2f(red[3][i] - N);
            Sp = Sp * g1 + red[4][i] * g2;
            Mp = N;
        }
        const float lse_r = (Mr + log2f(Sr)) * kLn2, lse_p = (Mp + log2f(Sp)) * kLn2;
        kl[r] = wr * (Ar / Sr - lse_r + lse_p);
        bc[0] = lse_r * kLog2e;
        bc[1] = lse_p * kLog2e;
    }
    if (!dpol) return;
    __syncthreads();
    const float lr2 = bc[0], lp2 = bc[1];
    const float gs = wr * gscale;
    T* drow = dpol + r * ld;
#pragma unroll
    for (int i = 0; i < (KEEP ? KCH : 1 << 30); ++i) {
        const int c = threadIdx.x + i * NT;
        if (c >= nvec) break;
        float fr[W], fp[W];
        if (KEEP) {
            RowIO<T>::cvt(keep_r[i], fr);
            RowIO<T>::cvt(keep_p[i], fp);
        } else {
            RowIO<T>::load(rrow + (int64_t)c * W, fr);
            RowIO<T>::load(prow + (int64_t)c * W, fp);
        }
#pragma unroll
        for (int j = 0; j < W; ++j) fp[j] = gs * (exp2f(fp[j] * kLog2e - lp2) - exp2f(fr[j] * kLog2e - lr2));
        RowIO<T>::store(drow + (int64_t)c * W, fp);
    }
    for (int v = nvec * W + threadIdx.x; v < V; v += NT) {
        const float zr = RowIO<T>::get(rrow + v), zp = RowIO<T>::get(prow + v);
        RowIO<T>::put(drow + v, gs * (exp2f(zp * kLog2e - lp2) - exp2f(zr * kLog2e - lr2)));
    }
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phrase_sum_fwd_kernel(const float* __restrict__ logp, const int64_t* __restrict__ labels,
                                                             const int64_t* __restrict__ signs,
                                                             const int64_t* __restrict__ slot_ids, int P,
                                                             float* __restrict__ acc, int T1) {
    __shared__ float red[kNW];
    const int b = blockIdx.x;
    const float* lp = logp + (int64_t)b * T1;
    const int64_t* lb = labels + (int64_t)b * T1;
    const int64_t* sg = signs + (int64_t)b * T1;
    for (int p = 0; p < P; ++p) {
        const int64_t id = slot_ids[p];
        float s = 0.f;
        for (int t = threadIdx.x; t < T1; t += 256) {
            const int64_t v = sg[t] == -100 ? 0 : sg[t];   // halva_trainer.py:560 (masked_fill(-100 -> 0))
            if (v == id && lb[t] != -100) s += lp[t];
        }
        s = block_sum<kNW>(s, red);
        if (threadIdx.x == 0) acc[(int64_t)b * P + p] = s;
    }
}

__global__ __launch_bounds__(256) void phrase_sum_bwd_kernel(const float* __restrict__ dacc, const int64_t* __restrict__ labels,
                                                             const int64_t* __restrict__ signs,
                                                             const int64_t* __restrict__ slot_ids, int P,
                                                             float* __restrict__ dlogp, int T1, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / T1;
        float g = 0.f;
        const int64_t s = signs[i] == -100 ? 0 : signs[i];
        if (labels[i] != -100) {
            for (int p = 0; p < P; ++p)
                if (slot_ids[p] == s) g = dacc[b * P + p];
        }
        dlogp[i] = g;
    }
}

}  // namespace

extern "C" int halva_token_logp_fwd(const void* logits, halva_dtype dt, int64_t ld, const int32_t* target, float* logp,
                                    float* lse, int64_t R, int V, void* stream) {
    HALVA_CHECK_ARG(logits && target && logp && lse, "token_logp_fwd: null pointer");
    HALVA_CHECK_ARG(V > 0 && ld >= V, "token_logp_fwd: bad V=%d / ld=%lld", V, (long long)ld);
    HALVA_CHECK_ARG(R < (1ll << 31), "token_logp_fwd: too many rows");
    if (R <= 0) return HALVA_OK;
    if (dt == HALVA_BF16 && V / 8 <= 16 * 256)
        hipLaunchKernelGGL((token_logp_fwd_kernel<bf16_t, 16>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)logits, ld, target, logp, lse, V);
    else if (dt == HALVA_BF16)
        hipLaunchKernelGGL((token_logp_fwd_kernel<bf16_t, 0>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)logits, ld, target, logp, lse, V);
    else if (dt == HALVA_F32)
        hipLaunchKernelGGL((token_logp_fwd_kernel<float, 0>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const float*)logits, ld, target, logp, lse, V);
    else
        HALVA_CHECK_ARG(false, "token_logp_fwd: unsupported dtype %d", (int)dt);
    HALVA_CHECK_LAUNCH("token_logp_fwd");
    return HALVA_OK;
}

extern "C" int halva_token_logp_bwd(const void* logits, halva_dtype dt, int64_t ld, const int32_t* target, const float* lse,
                                    const float* g, void* dlogits, int64_t R, int V, void* stream) {
    HALVA_CHECK_ARG(logits && target && lse && g && dlogits, "token_logp_bwd: null pointer");
    HALVA_CHECK_ARG(V > 0 && ld >= V, "token_logp_bwd: bad V=%d / ld=%lld", V, (long long)ld);
    HALVA_CHECK_ARG(R < (1ll << 31), "token_logp_bwd: too many rows");
    if (R <= 0) return HALVA_OK;
    if (dt == HALVA_BF16 && V / 8 <= 16 * 256)
        hipLaunchKernelGGL((token_logp_bwd_kernel<bf16_t, 16>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)logits, ld, target, lse, g, (bf16_t*)dlogits, V);
    else if (dt == HALVA_BF16)
        hipLaunchKernelGGL((token_logp_bwd_kernel<bf16_t, 0>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)logits, ld, target, lse, g, (bf16_t*)dlogits, V);
    else if (dt == HALVA_F32)
        hipLaunchKernelGGL((token_logp_bwd_kernel<float, 0>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                           (const float*)logits, ld, target, lse, g, (float*)dlogits, V);
    else
        HALVA_CHECK_ARG(false, "token_logp_bwd: unsupported dtype %d", (int)dt);
    HALVA_CHECK_LAUNCH("token_logp_bwd");
    return HALVA_OK;
}

extern "C" int halva_kl_rows(const void* pol, const void* ref, halva_dtype dt, int64_t ld, const float* w, float* kl,
                             void* dpol, float gscale, int64_t R, int V, void* stream) {
    HALVA_CHECK_ARG(pol && ref && kl, "kl_rows: null pointer");
    HALVA_CHECK_ARG(V > 0 && ld >= V, "kl_rows: bad V=%d / ld=%lld", V, (long long)ld);
    HALVA_CHECK_ARG(R < (1ll << 31), "kl_rows: too many rows");
    if (R <= 0) return HALVA_OK;
    // the rows held in registers between the two passes (bf16 with a gradient, whole 16-byte chunks, at most 8 chunks per thread of 512);
    // HALVA_KL_KEEP=0 = the 256-thread form that reads them twice
    const char* kl_env = getenv("HALVA_KL_KEEP");      // (read on every call: a test flips it inside one process)
    const bool keep_on = !(kl_env && kl_env[0] == '0');
    const bool vec16 = ((((uintptr_t)pol | (uintptr_t)ref | (uintptr_t)dpol) & 15) == 0) && ld % 8 == 0 && V % 8 == 0;
    if (dt == HALVA_BF16 && dpol && keep_on && vec16 && V / 8 <= 8 * 512) {
        hipLaunchKernelGGL((kl_rows_kernel<bf16_t, 512, true, 8>), dim3((unsigned)R), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)pol,
                           (const bf16_t*)ref, ld, w, kl, (bf16_t*)dpol, gscale, V);
    } else if (dt == HALVA_BF16)
        hipLaunchKernelGGL((kl_rows_kernel<bf16_t, 256, false>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pol,
                           (const bf16_t*)ref, ld, w, kl, (bf16_t*)dpol, gscale, V);
    else if (dt == HALVA_F32)
        hipLaunchKernelGGL((kl_rows_kernel<float, 256, false>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, (const float*)pol,
                           (const float*)ref, ld, w, kl, (float*)dpol, gscale, V);
    else
        HALVA_CHECK_ARG(false, "kl_rows: unsupported dtype %d", (int)dt);
    HALVA_CHECK_LAUNCH("kl_rows");
    return HALVA_OK;
}

extern "C" int halva_phrase_sum_fwd(const float* logp, const int64_t* labels, const int64_t* signs, const int64_t* slot_ids,
                                    int P, float* acc, int B, int T1, void* stream) {
    HALVA_CHECK_ARG(logp && labels && signs && (P == 0 || (slot_ids && acc)), "phrase_sum_fwd: null pointer");
    if (B <= 0 || P <= 0 || T1 <= 0) return HALVA_OK;
    hipLaunchKernelGGL(phrase_sum_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logp, labels, signs, slot_ids, P, acc,
                       T1);
    HALVA_CHECK_LAUNCH("phrase_sum_fwd");
    return HALVA_OK;
}

extern "C" int halva_phrase_sum_bwd(const float* dacc, const int64_t* labels, const int64_t* signs, const int64_t* slot_ids,
                                    int P, float* dlogp, int B, int T1, void* stream) {
    HALVA_CHECK_ARG(labels && signs && dlogp && (P == 0 || (slot_ids && dacc)), "phrase_sum_bwd: null pointer");
    if (B <= 0 || T1 <= 0) return HALVA_OK;
    const int64_t total = (int64_t)B * T1;
    int64_t grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(phrase_sum_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, dacc, labels, signs,
                       slot_ids, P, dlogp, T1, total);
    HALVA_CHECK_LAUNCH("phrase_sum_bwd");
    return HALVA_OK;
}
