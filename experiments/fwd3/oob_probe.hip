// Probe (round 4): what does `buffer_load_dwordx4 ... offen lds` do with lanes whose address lies beyond the descriptor's num_records - and is the
// scalar offset part of the range check?  One wave; the LDS chunk is pre-filled with 0xAAAAAAAA; results: per lane the first dword it "loaded".
//   hipcc -O2 --offload-arch=gfx950 -o oob_probe oob_probe.hip && ./oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* src, unsigned* out, unsigned num_records, unsigned soff, unsigned vstride) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    const unsigned long long base = (unsigned long long)(size_t)src;
    u32x4 desc = {(unsigned)base, (unsigned)(base >> 32), num_records, 0x00020000u};
    desc[0] = __builtin_amdgcn_readfirstlane(desc[0]); desc[1] = __builtin_amdgcn_readfirstlane(desc[1]);
    desc[2] = __builtin_amdgcn_readfirstlane(desc[2]); desc[3] = __builtin_amdgcn_readfirstlane(desc[3]);
    const unsigned voff = lane * vstride;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(desc), "s"(dst), "s"(soff) : "memory", "m0");
    __syncthreads();
    out[lane] = lds[4 * lane];
    out[64 + lane] = lds[4 * lane + 3];
}

int main() {
    const int n = 1 << 16;      // (256 KiB: soffset 1024 + 3 * 16384 still lies inside the allocation - a missing check would show as loaded data, not a fault)
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;      // dword i holds i
    unsigned *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 128 * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    struct { unsigned nrec, soff, vstride; const char* what; } cases[] = {
        {64 * 16, 0, 16, "all 64 lanes in range (1024 B), soffset 0"},
        {32 * 16, 0, 16, "num_records = 512 B: lanes 32..63 beyond it (through the VECTOR offset)"},
        {1024 + 512, 1024, 16, "num_records = 1536 B, soffset 1024: lanes 32..63 beyond it ONLY if the scalar offset counts"},
        {1024, 1024, 16, "num_records = 1024 B, soffset 1024: every lane beyond it if the scalar offset counts"},
        {40 * 16 + 8, 0, 16, "num_records = 648 B: lane 40 straddles (its first 8 bytes in range)"},
        // round-4 advice: soffset BEYOND num_records - what sdpa_fwd3 (tile requests up to three tiles past the sequence), sdpa_bwd_dkv3 (statistics of steps
        // past the last record) and wgrad_dma_kernel (rows past the slab) issue.  A raw-buffer check written as offset >= num_records - soffset would wrap.
        {1024, 1024 + 3 * 16384, 16, "num_records = 1024 B, soffset 1024 + 3 * 16384: every lane far beyond it"},
        {1024, 2048, 16, "num_records = 1024 B, soffset 2048: every lane beyond it (soffset - num_records = 1024 = what a wrapped check would accept)"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, c.nrec, c.soff, c.vstride);
        std::vector<unsigned> r(128);
        hipMemcpy(r.data(), out, 128 * 4, hipMemcpyDeviceToHost);
        printf("%s  (%s)\n  lane: first dword / last dword:", c.what, hipGetErrorString(hipGetLastError()));
        for (int l : {0, 1, 31, 32, 33, 39, 40, 41, 63}) printf("  %d: %x/%x", l, r[l], r[64 + l]);
        printf("\n");
    }
    return 0;
}
