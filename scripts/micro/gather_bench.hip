// gather_bench.hip -- what does the chip deliver for uniformly random small reads?
// Calibrates the ceiling of the canopy kernel's record fetches (two random 64-byte
// sectors per pair).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bench scripts/micro/gather_bench.hip && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// each lane does `iters` x UNROLL independent reads of BYTES bytes at random 64-byte-aligned offsets
template <int BYTES, int UNROLL>
__global__ __launch_bounds__(1024) void k_gather(const uint8_t *__restrict__ table, uint32_t mask64, int iters, uint32_t *out)
{
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t off[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) off[k] = rng(s) & mask64;
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const uint8_t *p = table + (size_t)off[k] * 64;
            if (BYTES == 4) acc += *reinterpret_cast<const uint32_t *>(p);
            else if (BYTES == 16) { uint4 v = *reinterpret_cast<const uint4 *>(p); acc += v.x + v.w; }
            else if (BYTES == 32) { uint4 v = *reinterpret_cast<const uint4 *>(p); uint4 w = *reinterpret_cast<const uint4 *>(p + 16); acc += v.x + w.w; }
            else { uint4 v = *reinterpret_cast<const uint4 *>(p); uint4 w = *reinterpret_cast<const uint4 *>(p + 48); acc += v.x + w.w; }
        }
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}

template <int BYTES, int UNROLL>
static int run(const uint8_t *d_table, size_t table_bytes, uint32_t *d_out, int blocks)
{
    const uint32_t mask64 = (uint32_t)(table_bytes / 64 - 1);
    const int iters = 256 / UNROLL;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_gather<BYTES, UNROLL>), dim3(blocks), dim3(1024), 0, 0, d_table, mask64, iters, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double reads = (double)blocks * 1024 * iters * UNROLL;
    printf("table %7.0f MiB  %2d B/read  unroll %d  blocks %4d : %8.3f ms  %7.2f G reads/s  %7.2f TB/s (64-B sectors)\n",
           table_bytes / 1048576.0, BYTES, UNROLL, blocks, best, reads / best / 1e6, reads * 64 / best / 1e9);
    return 0;
}

int main()
{
    const size_t max_bytes = (size_t)4 << 30;
    uint8_t *d_table; uint32_t *d_out;
    CK(hipMalloc(&d_table, max_bytes)); CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_table, 1, max_bytes));
    for (size_t mb : {2, 16, 64, 128, 512, 4096}) {
        size_t bytes = mb << 20;
        for (int blocks : {256, 512}) {
            if (run<4, 4>(d_table, bytes, d_out, blocks)) return 1;
            if (run<16, 4>(d_table, bytes, d_out, blocks)) return 1;
            if (run<32, 4>(d_table, bytes, d_out, blocks)) return 1;
            if (run<64, 4>(d_table, bytes, d_out, blocks)) return 1;
        }
        if (run<4, 8>(d_table, bytes, d_out, 512)) return 1;
        if (run<32, 8>(d_table, bytes, d_out, 512)) return 1;
    }
    return 0;
}
