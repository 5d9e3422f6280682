// pcie_bench.hip -- what the host link delivers, by transfer engine.  The staged host path
// moved ~46 GB/s as the SUM of both directions whatever the number of copy threads
// (profiles/host_path_r02.jsonl); this separates the candidates:
//   (1) hipMemcpyAsync H2D alone, D2H alone, both at once on two streams (SDMA engines)
//   (2) a kernel reading pinned host memory and writing pinned host memory directly
//       (8 B in + 8 B out per lane -- the shape of the host path's wire format)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pcie_bench scripts/micro/pcie_bench.hip && /tmp/pcie_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_host_rw(const uint2 *__restrict__ in, float *__restrict__ out_d,
                                                  int *__restrict__ out_m, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint2 v = in[i];
        out_d[i] = (float)v.x;
        out_m[i] = (int)v.y;
    }
}

__global__ __launch_bounds__(1024) void k_host_r(const uint2 *__restrict__ in, uint2 *__restrict__ dev, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dev[i] = in[i];
}

__global__ __launch_bounds__(1024) void k_host_w(const uint2 *__restrict__ dev, uint2 *__restrict__ out, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = dev[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const long long n = 64ll << 20;                 // 64M elements of 8 B = 512 MiB per direction
    const size_t bytes = (size_t)n * 8;
    void *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes));
    memset(h_in, 1, bytes); memset(h_out, 2, bytes);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    auto report = [&](const char *what, double t, double gb) { printf("%-64s %7.2f ms  %6.1f GB/s\n", what, t * 1e3, gb / t); };
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
        t = now() - t; if (rep) report("hipMemcpyAsync H2D alone", t, bytes / 1e9);
        t = now();
        CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
        t = now() - t; if (rep) report("hipMemcpyAsync D2H alone", t, bytes / 1e9);
        t = now();
        CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0));
        CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        t = now() - t; if (rep) report("hipMemcpyAsync H2D + D2H at once (two streams), sum", t, 2 * bytes / 1e9);
        // chunked like the pipe: 32 MiB pieces alternating on two streams
        t = now();
        const size_t piece = 32u << 20;
        for (size_t off = 0, k = 0; off < bytes; off += piece, k++) {
            hipStream_t s = (k & 1) ? s1 : s0;
            CK(hipMemcpyAsync((char *)d_a + off, (char *)h_in + off, piece, hipMemcpyHostToDevice, s));
            CK(hipMemcpyAsync((char *)h_out + off, (char *)d_b + off, piece, hipMemcpyDeviceToHost, s));
        }
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        t = now() - t; if (rep) report("32 MiB pieces, H2D then D2H per piece, two streams, sum", t, 2 * bytes / 1e9);
    }
    for (int blocks : {256, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            double t = now();
            hipLaunchKernelGGL(k_host_r, dim3(blocks), dim3(1024), 0, s0, (const uint2 *)h_in, (uint2 *)d_a, n);
            CK(hipStreamSynchronize(s0));
            t = now() - t; if (rep) { char b[96]; snprintf(b, 96, "kernel reads pinned host (8 B/lane), %d blocks", blocks); report(b, t, bytes / 1e9); }
            t = now();
            hipLaunchKernelGGL(k_host_w, dim3(blocks), dim3(1024), 0, s0, (const uint2 *)d_b, (uint2 *)h_out, n);
            CK(hipStreamSynchronize(s0));
            t = now() - t; if (rep) { char b[96]; snprintf(b, 96, "kernel writes pinned host (8 B/lane), %d blocks", blocks); report(b, t, bytes / 1e9); }
            t = now();
            hipLaunchKernelGGL(k_host_rw, dim3(blocks), dim3(1024), 0, s0, (const uint2 *)h_in, (float *)h_out, (int *)h_out + n, n);
            CK(hipStreamSynchronize(s0));
            t = now() - t; if (rep) { char b[96]; snprintf(b, 96, "kernel reads 8 B + writes 4+4 B per lane, host both, %d blocks, sum", blocks); report(b, t, 2 * bytes / 1e9); }
        }
    }
    return 0;
}
