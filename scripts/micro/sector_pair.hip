// sector_pair.hip -- does it pay to let two adjacent lanes fetch the two 64-byte sectors of one 128-byte
// line in the SAME instruction (one 128-byte request from the L1?) instead of one lane fetching
// them one after the other (two 64-byte requests)?  The question behind a lane-pair form of the
// lineage-length stream of k_walk_sorted, whose counters say: bound by the L1's outstanding misses.
//   A  every lane: random 128-byte line, sector 0 (4 x 16 B), then sector 1 (4 x 16 B)      [streams today]
//   B  lane pair (2k, 2k+1): one random line, lane 2k sector 0, lane 2k+1 sector 1, same instructions
//   C  every lane: one sector (4 x 16 B) of a random line                                     [reference]
//   D  every lane: one 16-byte quad of a random line (with the L1 counters: does C's sector cost one L1 -> L2
//      request or four?)
// Tables: 32 MiB (Infinity Cache) and 2 GiB (HBM).  Rates in G sectors/s.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/sector_pair scripts/micro/sector_pair.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ table, uint32_t line_mask, int iters, uint32_t *out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (MODE == 1 ? tid >> 1 : tid) * 2654435761u + 12345u;      // B: both lanes of a pair draw the same lines
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint8_t *line = table + (size_t)(rng(s) & line_mask) * 128;
        if (MODE == 0) {
            const uint4 *p = reinterpret_cast<const uint4 *>(line);
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            const uint4 e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        } else if (MODE == 3) {      // D: one 16-byte quad of a random line (how many L1 -> L2 requests does C's sector cost?)
            const uint4 a = *reinterpret_cast<const uint4 *>(line);
            acc += a.x + a.w;
        } else {
            const uint4 *p = reinterpret_cast<const uint4 *>(line + (MODE == 1 ? (tid & 1) * 64 : 0));
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        }
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}

int main()
{
    uint32_t *d_out; CK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t bytes : {(size_t)32 << 20, (size_t)2 << 30}) {
        uint8_t *t; CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 1, bytes));
        const uint32_t mask = (uint32_t)(bytes / 128 - 1);
        const int blocks = 256 * 8, iters = 256;
        for (int mode = 0; mode < 4; mode++) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, t, mask, iters, d_out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, t, mask, iters, d_out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, t, mask, iters, d_out);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, t, mask, iters, d_out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double lanes = (double)blocks * 256 * iters;
            const double sectors = mode == 0 ? 2 * lanes : lanes;
            printf("table %5zu MiB  %s  %7.3f ms  %6.1f G sectors/s  %5.2f TB/s\n", bytes >> 20,
                   mode == 0 ? "A one lane, both sectors in turn " : mode == 1 ? "B lane pair, one sector each     " : mode == 2 ? "C one lane, one sector           " : "D one lane, one 16-byte quad     ",
                   best, sectors / best / 1e6, sectors * 64 / best / 1e9);
        }
        CK(hipFree(t));
    }
    return 0;
}
