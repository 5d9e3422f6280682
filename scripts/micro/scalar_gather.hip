// scalar_gather.hip -- microbenchmark (round 6): can the scalar data cache's miss path carry random 4-byte gathers BESIDE the vector
// L1's?  The headline kernel is bound by the CU's vector miss path (TA busy 93 %, ~97 L1 misses in flight per CU at ~750 cycles:
// profiles/counters_headline_r06.txt); its a side is one 4-byte read per pair from a 4 MiB table.  If s_load_dword gathers (wave-uniform
// addresses, 15 outstanding per wave by lgkmcnt) have a miss capacity of their own, the a side could move there.
//   mode 0: vector gathers only   (every lane one random 4-byte read per iteration, UNROLL in flight)
//   mode 1: scalar gathers only   (every wave 14 random s_load_dword per iteration)
//   mode 2: both in the same wave (the capacities add, or they do not)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/scalar_gather scripts/micro/scalar_gather.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef const uint32_t __attribute__((address_space(4))) *const_words;

__device__ __forceinline__ uint32_t lcg(uint32_t s) { return s * 1664525u + 1013904223u; }

template <int MODE>
__global__ __launch_bounds__(1024) void k_gather(const uint32_t *__restrict__ table, uint32_t line_mask, int iters, uint32_t *out)
{
    const_words ctab = (const_words)(uintptr_t)table;
    uint32_t vseed = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t sseed = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 40503u + 977u));
    uint32_t vacc = 0, sacc = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE != 1) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                vseed = lcg(vseed);
                v[k] = table[((vseed >> 7) & line_mask) * 16u];      // one 4-byte read per 64-byte line
            }
#pragma unroll
            for (int k = 0; k < 4; k++) vacc ^= v[k];
        }
        if (MODE != 0) {
            uint32_t s[14];
#pragma unroll
            for (int k = 0; k < 14; k++) {
                sseed = lcg(sseed);
                s[k] = ctab[((sseed >> 7) & line_mask) * 16u];       // wave-uniform address in the constant address space: s_load_dword
            }
#pragma unroll
            for (int k = 0; k < 14; k++) sacc ^= s[k];
        }
    }
    if ((vacc ^ sacc) == 0x9E3779B9u) out[0] = vacc ^ sacc;
}

// mode 3: the headline kernel's mix if its a side went scalar -- per wave and iteration 64 x VPL vector reads (one or two per lane: the b
// records, the ids) in flight WHILE 56 scalar reads (four batches of 14: lgkmcnt holds 15) are issued and waited for
template <int VPL>
__global__ __launch_bounds__(1024) void k_mix(const uint32_t *__restrict__ table, uint32_t line_mask, int iters, uint32_t *out)
{
    const_words ctab = (const_words)(uintptr_t)table;
    uint32_t vseed = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t sseed = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 40503u + 977u));
    uint32_t vacc = 0, sacc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v[VPL];
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            vseed = lcg(vseed);
            v[k] = table[((vseed >> 7) & line_mask) * 16u];
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t s[14];
#pragma unroll
            for (int k = 0; k < 14; k++) {
                sseed = lcg(sseed);
                s[k] = ctab[((sseed >> 7) & line_mask) * 16u];
            }
#pragma unroll
            for (int k = 0; k < 14; k++) sacc ^= s[k];
        }
#pragma unroll
        for (int k = 0; k < VPL; k++) vacc ^= v[k];
    }
    if ((vacc ^ sacc) == 0x9E3779B9u) out[0] = vacc ^ sacc;
}

template <int VPL>
static void run_mix(const uint32_t *table, uint32_t line_mask, uint32_t *out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mix<VPL>, dim3(blocks), dim3(1024), 0, 0, table, line_mask, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double waves = (double)blocks * 16;
    const double vreads = waves * 64 * iters * VPL, sreads = waves * iters * 56;
    std::printf("mix: %d vector/lane + 56 scalar/wave blocks %4d: %.3f ms  vector %.2f G/s  scalar %.2f G/s  total %.2f G/s\n", VPL, blocks, best,
                vreads / best / 1e6, sreads / best / 1e6, (vreads + sreads) / best / 1e6);
}

template <int MODE>
static void run(const uint32_t *table, uint32_t line_mask, uint32_t *out, int blocks, int iters, const char *what)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(1024), 0, 0, table, line_mask, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double waves = (double)blocks * 16, lanes = waves * 64;
    const double vreads = MODE != 1 ? lanes * iters * 4 : 0, sreads = MODE != 0 ? waves * iters * 14 : 0;
    std::printf("%-28s blocks %4d: %.3f ms  vector %.2f G reads/s  scalar %.2f G reads/s  total %.2f G/s\n", what, blocks, best, vreads / best / 1e6,
                sreads / best / 1e6, (vreads + sreads) / best / 1e6);
}

int main(int argc, char **argv)
{
    const double mib = argc > 1 ? std::atof(argv[1]) : 20.0;
    uint32_t lines = 1;
    while ((double)lines * 2 * 64 <= mib * 1048576.0) lines *= 2;      // power of two lines <= the footprint
    const size_t bytes = (size_t)lines * 64;
    uint32_t *table, *out;
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(table, 1, bytes));
    std::printf("table %.1f MiB (%u lines)\n", bytes / 1048576.0, lines);
    for (int blocks : {256, 512}) {
        run<0>(table, lines - 1, out, blocks, 64, "vector gathers only");
        run<1>(table, lines - 1, out, blocks, 64 * 16, "scalar gathers only");
        run<2>(table, lines - 1, out, blocks, 64, "both in every wave");
        run_mix<1>(table, lines - 1, out, blocks, 256);
        run_mix<2>(table, lines - 1, out, blocks, 256);
    }
    return 0;
}
