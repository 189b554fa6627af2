// sdpa_fwd3 in plain HIP (HALVA_FWD3_ASM=0): the readable twin of the generated block (sdpa_fwd3.h, gen_fwd3_loop.py), held to it BIT FOR BIT by
// tests/test_sdpa_bench_shapes_gpu.py::test_fwd3_plain_hip_twin_matches_the_generated_loop.  Same algorithm, same arithmetic in the same order - and
// nothing of the machinery: no tile ring, no LDS, no persistence; every operand fragment is gathered from global memory where it is needed (slow:
// a test and documentation kernel).  What has to agree for the bits to agree, and does:
//   * S^T[key][query] = K Q^T per (32-query group, 32-key half-tile): eight v_mfma_f32_32x32x16_bf16 over head_dim (A = K rows, B = Q fragments),
//     lane (query, h), register r <-> key 8 (r >> 2) + 4 h + (r & 3);
//   * the row's exponent reference: sc * max over the VISIBLE scores of the first half-tile of the walk (0 if it sees none), fixed for the row block;
//   * P = exp2(fma(S, sc, -m_ref)) (hidden keys: S = -inf -> 0); the row sum in FOUR partial sums per lane (registers r with r & 3 == 0 / 1 / 2 / 3);
//     P packed to bf16 in register pairs, so that operand slot (h, j) of key chunk k16 holds key 16 k16 + 8 (j >> 2) + 4 h + (j & 3);
//   * O^T[d][query] += V^T P: per half-tile, chunk k16 = 0, 1 and head_dim tile dt = 0..3 one MFMA (A = V^T gathered in the same slot order);
//   * after the last tile the workgroup votes on the partial sums (any not < 2^100: the whole row block again, those rows' references + 120);
//   * l = ((p0 + p2) + (p1 + p3)) + the same of the row's other lane; O / l by v_rcp_f32 and a multiply; lse = (log2(l) + m_ref) * ln 2.
// The walk (which 64-key tiles, in which order, incl. the jump of a block wholly inside branch B) is fwd3_geom's; visibility is computed per
// (row, key) from the definition - the generated loop's "row sees the first n keys of this tile" masks are equivalent to it.
template <bool CAUSAL>
__global__ __launch_bounds__(256) void sdpa_fwd3_twin_kernel(const SdpaParams p) {
    static_assert(CAUSAL, "sdpa_fwd3 is the causal head_dim-128 instantiation");
    constexpr int D = 128, BM = 256;
    __shared__ int vote;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r32 = lane & 31;
    const int qb = blockIdx.x % p.nblk, pair = blockIdx.x / p.nblk, s = pair / p.H, hd = pair % p.H;
    const int start = p.seq_start ? p.seq_start[s] : 0, len = p.seq_len ? p.seq_len[s] : p.T;
    const Branch br = load_branch(p, s);
    const Fwd3Geom g = fwd3_geom(p, qb, start, len, br, wave);
    const int64_t seq_row0 = (int64_t)s * p.T;
    const float sc = p.scale * kLog2e;
    const bf16_t* qh = p.q + hd * D;
    const bf16_t* kh_ = p.k + hd * D;
    const bf16_t* vh = p.v + hd * D;
    auto visible = [&](int ql, int k) { return k >= 0 && k <= ql && k < len && !(ql >= br.b && k >= br.a && k < br.b); };
    auto load8 = [&](const bf16_t* ptr, bool ok) {      // 8 bf16 (16 bytes) or zeros
        const u32x4 v = ok ? *reinterpret_cast<const u32x4*>(ptr) : u32x4{0u, 0u, 0u, 0u};
        return __builtin_bit_cast(bf16x8, v);
    };
    int ql[2];
    bf16x8 qf[2][8];
    for (int gi = 0; gi < 2; ++gi) {
        ql[gi] = qb * BM + 64 * wave + 32 * gi + r32 - start;
        const bool ok = ql[gi] >= 0 && ql[gi] < len;
        for (int ks = 0; ks < 8; ++ks) qf[gi][ks] = load8(qh + (seq_row0 + start + ql[gi]) * p.ld_qkv + 16 * ks + 8 * h, ok);
    }
    float mref[2] = {0.f, 0.f};
    f32x16 acc[2][4];
    float l2[2][4];
    for (int pass = 0;; ++pass) {
        for (int gi = 0; gi < 2; ++gi) {
            for (int dt = 0; dt < 4; ++dt)
                for (int r = 0; r < 16; ++r) acc[gi][dt][r] = 0.f;
            for (int i = 0; i < 4; ++i) l2[gi][i] = 0.f;
        }
        for (int tw = 0; tw < g.N; ++tw) {
            const int key0 = g.kv_first + 64 * tw + (tw >= g.n1req ? (int)g.jump_rows : 0);
            for (int khalf = 0; khalf < 2; ++khalf) {
                const int kb0 = key0 + 32 * khalf;
                bf16x8 ka[8];
                for (int ks = 0; ks < 8; ++ks) {
                    const int key = kb0 + r32;
                    ka[ks] = load8(kh_ + (seq_row0 + start + key) * p.ld_qkv + 16 * ks + 8 * h, key >= 0 && key < len);
                }
                unsigned pb[2][8];
                for (int gi = 0; gi < 2; ++gi) {
                    f32x16 x;
                    for (int r = 0; r < 16; ++r) x[r] = 0.f;
                    for (int ks = 0; ks < 8; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[gi][ks], x, 0, 0, 0);
                    for (int r = 0; r < 16; ++r)
                        if (!visible(ql[gi], kb0 + 8 * (r >> 2) + 4 * h + (r & 3))) x[r] = -INFINITY;
                    if (tw == 0 && khalf == 0 && pass == 0) {      // the exponent reference of the row block
                        float m = -INFINITY;
                        for (int r = 0; r < 16; ++r) m = fmaxf(m, x[r]);
                        m = fmaxf(m, __shfl_xor(m, 32, 64));
                        const float mr = sc * m;
                        mref[gi] = (mr > -INFINITY) ? mr : 0.f;
                    }
                    for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r], sc, -mref[gi]));
                    for (int i = 0; i < 8; ++i) {
                        l2[gi][2 * (i & 1) + 0] += x[2 * i];
                        l2[gi][2 * (i & 1) + 1] += x[2 * i + 1];
                        pb[gi][i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
                    }
                }
                for (int k16 = 0; k16 < 2; ++k16)
                    for (int dt = 0; dt < 4; ++dt) {
                        s16x8 va;      // V^T fragment: lane (d = 32 dt + r32, h), slot j <-> key kb0 + 16 k16 + 8 (j >> 2) + 4 h + (j & 3)
                        for (int j = 0; j < 8; ++j) {
                            const int key = kb0 + 16 * k16 + 8 * (j >> 2) + 4 * h + (j & 3);
                            va[j] = (key >= 0 && key < len) ? (short)vh[(seq_row0 + start + key) * p.ld_qkv + 32 * dt + r32] : (short)0;
                        }
                        for (int gi = 0; gi < 2; ++gi) {
                            const u32x4 pw = u32x4{pb[gi][4 * k16], pb[gi][4 * k16 + 1], pb[gi][4 * k16 + 2], pb[gi][4 * k16 + 3]};
                            acc[gi][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pw), acc[gi][dt], 0, 0, 0);
                        }
                    }
            }
        }
        // the vote: has any partial row sum outgrown 2^100 (P overflowed, or is about to)?
        bool again[2];
        int mine = 0;
        for (int gi = 0; gi < 2; ++gi) {
            float t = fmaxf(fmaxf(fmaxf(l2[gi][0], l2[gi][1]), l2[gi][2]), l2[gi][3]);
            t = fmaxf(t, __shfl_xor(t, 32, 64));
            again[gi] = !(t < 1.2676506002282294e30f);      // 2^100 (true for inf and NaN too)
            mine |= again[gi] ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) vote = 0;
        __syncthreads();
        if (mine) vote = 1;
        __syncthreads();
        if (!vote || pass >= 8) break;      // (MAX_REDO, gen_fwd3_loop.py)
        for (int gi = 0; gi < 2; ++gi)
            if (again[gi]) mref[gi] = mref[gi] + 120.f;
    }
    for (int gi = 0; gi < 2; ++gi) {
        const int gq = qb * BM + 64 * wave + 32 * gi + r32;      // row in [0, T)
        const bool in_T = gq < p.T, valid = in_T && ql[gi] >= 0 && ql[gi] < len;
        const float lh = (l2[gi][0] + l2[gi][2]) + (l2[gi][1] + l2[gi][3]);
        const float lt = lh + __shfl_xor(lh, 32, 64);
        const float inv = (valid && lt > 0.f) ? __builtin_amdgcn_rcpf(lt) : 0.f;
        float lse = (__builtin_amdgcn_logf(lt) + mref[gi]) * 0.6931471824645996f;      // (0x3f317218)
        if (!valid) lse = 0.f;
        if (!in_T) continue;
        bf16_t* orow = p.o + (seq_row0 + gq) * p.ld_o + hd * D;
        for (int dt = 0; dt < 4; ++dt)
            for (int i = 0; i < 8; ++i) {
                const int d = 32 * dt + 8 * ((2 * i) >> 2) + 4 * h + ((2 * i) & 3);
                *reinterpret_cast<unsigned*>(orow + d) = pack_bf16x2(acc[gi][dt][2 * i] * inv, acc[gi][dt][2 * i + 1] * inv);
            }
        if (h == 0) p.lse[((int64_t)s * p.H + hd) * p.T + gq] = lse;
    }
}
