// link_pack_bench.hip -- would narrower wire formats lift the host path's link-side ceiling?  A kernel that reads its
// pairs from pinned host memory and writes its results to pinned host memory (the zero-copy host path), per pair:
//   (A) 8 B in (int32 x 2), 4 B + 4 B out (float32 distance, int32 MRCA id)          [today]
//   (B) 6 B in (24-bit ids), 4 B + 3 B out (float32, 24-bit MRCA id), moved as aligned 16-byte accesses per wave
//   (C) 8 B in, 4 B out (distances only)          (D) 6 B in, 4 B out
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/link_pack_bench scripts/micro/link_pack_bench.hip && /tmp/link_pack_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// per wave of 64 pairs: IN16 / OUTD16 / OUTM16 aligned 16-byte accesses (traffic only; the values are made up)
template <int IN16, int OUTD16, int OUTM16>
__global__ __launch_bounds__(1024) void k_wire(const uint4 *__restrict__ in, uint4 *__restrict__ out_d, uint4 *__restrict__ out_m, long long n_waves)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, stride = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long w = wave0; w < n_waves; w += stride) {
        uint4 v = make_uint4(1, 2, 3, 4);
        if (lane < IN16) v = in[w * IN16 + lane];
        v.x += __shfl(v.y, (lane * 7) & 63);      // (something depends on the loads)
        if (lane < OUTD16) out_d[w * OUTD16 + lane] = v;
        if (OUTM16 && lane < OUTM16) out_m[w * OUTM16 + lane] = v;
    }
}

// per-lane form of the 6-byte input: three 2-byte loads at 6 i (no cooperation between lanes)
__global__ __launch_bounds__(1024) void k_wire_lane6(const uint16_t *__restrict__ in, float *__restrict__ out_d, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint16_t *p = in + 3 * i;
        const uint32_t a = p[0] | ((uint32_t)(p[1] & 0xFF) << 16), b = (p[1] >> 8) | ((uint32_t)p[2] << 8);
        out_d[i] = (float)(a + b);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const long long n = 64ll << 20;      // pairs
    void *h_in, *h_d, *h_m;
    CK(hipHostMalloc(&h_in, (size_t)n * 8, hipHostMallocDefault));
    CK(hipHostMalloc(&h_d, (size_t)n * 4, hipHostMallocDefault));
    CK(hipHostMalloc(&h_m, (size_t)n * 4, hipHostMallocDefault));
    memset(h_in, 1, (size_t)n * 8); memset(h_d, 0, (size_t)n * 4); memset(h_m, 0, (size_t)n * 4);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run = [&](const char *what, auto kern, int in_b, int out_b) -> int {
        double best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            const double t0 = now();
            hipLaunchKernelGGL(kern, dim3(1024), dim3(1024), 0, s, (const uint4 *)h_in, (uint4 *)h_d, (uint4 *)h_m, n / 64);
            CK(hipStreamSynchronize(s));
            const double t = now() - t0;
            if (rep && t < best) best = t;
        }
        printf("%-58s %7.2f ms  %.3e pairs/s  in %5.1f GB/s  out %5.1f GB/s\n", what, best * 1e3, n / best, n * (double)in_b / best / 1e9, n * (double)out_b / best / 1e9);
        return 0;
    };
    if (run("(A) 8 B in, 4 + 4 B out  [today]", k_wire<32, 16, 16>, 8, 8)) return 1;
    if (run("(B) 6 B in, 4 + 3 B out", k_wire<24, 16, 12>, 6, 7)) return 1;
    if (run("(C) 8 B in, 4 B out  [today, distances only]", k_wire<32, 16, 0>, 8, 4)) return 1;
    if (run("(D) 6 B in, 4 B out", k_wire<24, 16, 0>, 6, 4)) return 1;
    {
        double best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            const double t0 = now();
            hipLaunchKernelGGL(k_wire_lane6, dim3(1024), dim3(1024), 0, s, (const uint16_t *)h_in, (float *)h_d, n);
            CK(hipStreamSynchronize(s));
            const double t = now() - t0;
            if (rep && t < best) best = t;
        }
        printf("%-58s %7.2f ms  %.3e pairs/s\n", "(E) 6 B in as three 2-byte loads per lane, 4 B out", best * 1e3, n / best);
    }
    return 0;
}
