// Which of the library's GEMM kernels is fastest for a shape UNDER SUSTAINED LOAD?  (tools/tune_gemms_sustained.py drives this through ctypes.)
//
// PyTorch's TunableOp - what tools/tune_gemms.sh used for halva_amd/tuned/gfx950_tunableop.csv - ranks the candidates by a short burst; on this
// part the burst winner runs at ~2.0 PFLOP/s for a few launches and at ~1.5 once the power limit pulls the clock down, and the step (1 300 GEMM
// launches back to back, 77 % of its time) only ever sees the second number.  This harness screens every hipBLASLt algorithm and every rocBLAS
// solution that supports the problem by burst time, then times the best `topk` of them back to back for `secs` seconds each, in two interleaved
// rounds, and reports both.  The problem description (column-major m, n, k, lda, ldb, ldc, transposes, bf16 in / out, f32 accumulation, alpha 1,
// beta 0) and the solution numbering are the ones PyTorch's GemmTunableOp uses (aten/src/ATen/cuda/tunable/GemmHipblaslt.h, GemmRocblas.h), so a
// winner can be written into the table as Gemm_Hipblaslt_<index> / Gemm_Rocblas_<index>.  Loaded into a Python process AFTER `import torch`, so the
// BLAS libraries (and the HIP runtime) are the ones PyTorch itself uses - solution indices are not portable across library builds.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        auto e_ = (x);                                                                         \
        if ((int)e_ != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); return -1; } \
    } while (0)

namespace {
struct Cand {
    int lib;      // 0 hipBLASLt, 1 rocBLAS
    int index;
    hipblasLtMatmulAlgo_t algo;
    size_t ws;
    float burst_ms;
    float sustained_ms[2];
    std::string name;
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}      // namespace

// out: up to topk rows of (lib, index, burst_ms, sustained_ms round 0, sustained_ms round 1) as doubles, 5 per row; returns the row count (< 0: error)
extern "C" int lt_sustained(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc, int topk, double secs,
                            int with_rocblas, double* out, char* names, int name_bytes) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipblasLtHandle_t lt;
    CK(hipblasLtCreate(&lt));
    const size_t a_elems = (size_t)lda * (trans_a ? m : k), b_elems = (size_t)ldb * (trans_b ? k : n), c_elems = (size_t)ldc * n;
    const size_t WS = 256u << 20;
    void *A, *B, *C, *W;
    CK(hipMalloc(&A, a_elems * 2)); CK(hipMalloc(&B, b_elems * 2)); CK(hipMalloc(&C, c_elems * 2)); CK(hipMalloc(&W, WS));
    {      // bf16 values in (-1, 1): the power a GEMM draws depends on how many bits toggle - zeros would flatter every kernel
        std::vector<uint16_t> h(1 << 20);
        uint32_t s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; float f = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
        for (size_t o = 0; o < a_elems; o += h.size()) CK(hipMemcpy((char*)A + o * 2, h.data(), std::min(h.size(), a_elems - o) * 2, hipMemcpyHostToDevice));
        for (size_t o = 0; o < b_elems; o += h.size()) CK(hipMemcpy((char*)B + o * 2, h.data() + 7, std::min(h.size() - 7, b_elems - o) * 2, hipMemcpyHostToDevice));
    }
    const hipblasOperation_t opa = trans_a ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = trans_b ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    hipblasLtMatmulDesc_t desc;
    CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)));
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, trans_a ? k : m, trans_a ? m : k, lda));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, trans_b ? n : k, trans_b ? k : n, ldb));
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, m, n, ldc));
    const float alpha = 1.f, beta = 0.f;

    std::vector<Cand> cands;
    {
        std::vector<hipblasLtMatmulHeuristicResult_t> all;
        CK(hipblaslt_ext::getAllAlgos(lt, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, opa, opb, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF, HIPBLAS_COMPUTE_32F, all));
        for (auto& r : all) {
            size_t ws = 0;
            if (hipblaslt_ext::matmulIsAlgoSupported(lt, desc, &alpha, la, lb, &beta, lc, lc, r.algo, ws) != HIPBLAS_STATUS_SUCCESS || ws > WS) continue;
            Cand c{};
            c.lib = 0; c.index = hipblaslt_ext::getIndexFromAlgo(r.algo); c.algo = r.algo; c.ws = ws;
            cands.push_back(c);
        }
        fprintf(stderr, "hipBLASLt: %zu algorithms, %zu support the problem\n", all.size(), cands.size());
    }
    rocblas_handle rb = nullptr;
    const rocblas_operation ra = trans_a ? rocblas_operation_transpose : rocblas_operation_none, rbo = trans_b ? rocblas_operation_transpose : rocblas_operation_none;
    if (with_rocblas) {
        CK(rocblas_create_handle(&rb));
        CK(rocblas_set_stream(rb, st));
        rocblas_int cnt = 0;
        CK(rocblas_gemm_ex_get_solutions(rb, ra, rbo, (int)m, (int)n, (int)k, &alpha, A, rocblas_datatype_bf16_r, (int)lda, B, rocblas_datatype_bf16_r, (int)ldb, &beta,
                                         C, rocblas_datatype_bf16_r, (int)ldc, C, rocblas_datatype_bf16_r, (int)ldc, rocblas_datatype_f32_r,
                                         rocblas_gemm_algo_solution_index, rocblas_gemm_flags_none, nullptr, &cnt));
        std::vector<rocblas_int> ids(cnt);
        CK(rocblas_gemm_ex_get_solutions(rb, ra, rbo, (int)m, (int)n, (int)k, &alpha, A, rocblas_datatype_bf16_r, (int)lda, B, rocblas_datatype_bf16_r, (int)ldb, &beta,
                                         C, rocblas_datatype_bf16_r, (int)ldc, C, rocblas_datatype_bf16_r, (int)ldc, rocblas_datatype_f32_r,
                                         rocblas_gemm_algo_solution_index, rocblas_gemm_flags_none, ids.data(), &cnt));
        for (int i = 0; i < cnt; ++i) { Cand c{}; c.lib = 1; c.index = ids[i]; cands.push_back(c); }
        fprintf(stderr, "rocBLAS: %d solutions\n", (int)cnt);
    }
    auto launch = [&](Cand& c) -> int {
        if (c.lib == 0)
            return (int)hipblasLtMatmul(lt, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &c.algo, W, c.ws, st);
        return (int)rocblas_gemm_ex(rb, ra, rbo, (int)m, (int)n, (int)k, &alpha, A, rocblas_datatype_bf16_r, (int)lda, B, rocblas_datatype_bf16_r, (int)ldb, &beta, C,
                                    rocblas_datatype_bf16_r, (int)ldc, C, rocblas_datatype_bf16_r, (int)ldc, rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index,
                                    c.index, rocblas_gemm_flags_none);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // screening: one warm-up + one timed launch; a second timed one only for candidates within 1.5 x of the best so far
    float best = 1e30f;
    std::vector<Cand> ok;
    for (auto& c : cands) {
        if (launch(c) != 0) { (void)hipGetLastError(); continue; }
        CK(hipEventRecord(e0, st));
        if (launch(c) != 0) continue;
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < 1.5f * best) {
            CK(hipEventRecord(e0, st));
            launch(c); launch(c);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
            ms = std::min(ms, ms2 / 2);
        }
        c.burst_ms = ms;
        best = std::min(best, ms);
        ok.push_back(c);
    }
    std::sort(ok.begin(), ok.end(), [](const Cand& a, const Cand& b) { return a.burst_ms < b.burst_ms; });
    if ((int)ok.size() > topk) ok.resize(topk);
    for (auto& c : ok) c.name = c.lib == 0 ? hipblaslt_ext::getKernelNameFromAlgo(lt, c.algo) : std::string("rocblas");
    for (int round = 0; round < 2; ++round)
        for (size_t i = 0; i < ok.size(); ++i) {
            Cand& c = ok[round ? ok.size() - 1 - i : i];      // second round in reverse order: a drift over the run shows up as disagreement between the rounds
            const double t_warm = now();
            while (now() - t_warm < 0.3 * secs) { for (int j = 0; j < 8; ++j) launch(c); CK(hipStreamSynchronize(st)); }
            int nl = 0;
            CK(hipEventRecord(e0, st));
            const double t0 = now();
            while (now() - t0 < secs) { for (int j = 0; j < 8; ++j) launch(c); nl += 8; CK(hipStreamSynchronize(st)); }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            c.sustained_ms[round] = ms / nl;
        }
    std::string all_names;
    for (size_t i = 0; i < ok.size(); ++i) {
        out[5 * i + 0] = ok[i].lib; out[5 * i + 1] = ok[i].index; out[5 * i + 2] = ok[i].burst_ms; out[5 * i + 3] = ok[i].sustained_ms[0]; out[5 * i + 4] = ok[i].sustained_ms[1];
        all_names += ok[i].name + "\n";
    }
    if (names && name_bytes > 0) { strncpy(names, all_names.c_str(), name_bytes - 1); names[name_bytes - 1] = 0; }
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(W);
    hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc); hipblasLtMatmulDescDestroy(desc);
    if (rb) rocblas_destroy_handle(rb);
    hipblasLtDestroy(lt);
    (void)hipStreamDestroy(st);
    return (int)ok.size();
}
