// microbench.hip -- measurement helpers, NOT part of the product path.
//
// bench.py loads libst_microbench.so to measure, in the same process and on the same GPU
// as the timed kernel, the two hardware ceilings its roofline block quotes:
//
//   stmb_random_sector_reads   uniformly random 64-byte-sector reads from a table of a given
//                              size (the access pattern of the canopy kernel's record fetches)
//   stmb_stream_copy           a plain 16-byte-per-lane streaming copy (achievable HBM rate)
//
// scripts/micro/gather_bench.hip wraps the same kernels in a main() that sweeps table sizes.
// Built for gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define STMB_CK(x)                                      \
    do {                                                \
        hipError_t e_ = (x);                            \
        if (e_ != hipSuccess) {                         \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                   \
        }                                               \
    } while (0)

namespace stmb {

__device__ __forceinline__ uint32_t rng(uint32_t &s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

// each lane does iters x UNROLL independent reads of BYTES bytes at random 64-byte-aligned offsets
template <int BYTES, int UNROLL>
__global__ __launch_bounds__(1024) void k_gather(const uint8_t *__restrict__ table, uint32_t n_sectors, int iters,
                                                 uint32_t *out)
{
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t off[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) off[k] = __umulhi(rng(s), n_sectors);      // (any table size, not only powers of two)
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const uint8_t *p = table + (size_t)off[k] * 64;
            if (BYTES == 4) acc += *reinterpret_cast<const uint32_t *>(p);
            else if (BYTES == 8) { uint2 v = *reinterpret_cast<const uint2 *>(p); acc += v.x + v.y; }
            else if (BYTES == 16) { uint4 v = *reinterpret_cast<const uint4 *>(p); acc += v.x + v.w; }
            else if (BYTES == 32) { uint4 v = *reinterpret_cast<const uint4 *>(p); uint4 w = *reinterpret_cast<const uint4 *>(p + 16); acc += v.x + w.w; }
            else { uint4 v = *reinterpret_cast<const uint4 *>(p); uint4 w = *reinterpret_cast<const uint4 *>(p + 48); acc += v.x + w.w; }
        }
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}

__global__ __launch_bounds__(1024) void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long n16)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// Streaming copy, shape by template: every lane moves UNROLL 16-byte words per trip -- all loads
// first, then all stores -- the words of one trip one block-width apart (coalesced per
// instruction); NT: non-temporal loads and stores (the copied bytes are used once).
template <int UNROLL, bool NT>
__global__ __launch_bounds__(1024) void k_copy_shape(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long n16)
{
    const long long per_block = (long long)blockDim.x * UNROLL;
    const long long stride = (long long)gridDim.x * per_block;
    for (long long base = (long long)blockIdx.x * per_block + threadIdx.x; base < n16; base += stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const long long i = base + (long long)k * blockDim.x;
            if (i < n16) {
                if (NT) {
                    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(src + i);
                    unsigned long long lo = __builtin_nontemporal_load(p), hi = __builtin_nontemporal_load(p + 1);
                    v[k] = make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
                } else {
                    v[k] = src[i];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const long long i = base + (long long)k * blockDim.x;
            if (i < n16) {
                if (NT) {
                    unsigned long long *p = reinterpret_cast<unsigned long long *>(dst + i);
                    __builtin_nontemporal_store((unsigned long long)v[k].x | ((unsigned long long)v[k].y << 32), p);
                    __builtin_nontemporal_store((unsigned long long)v[k].z | ((unsigned long long)v[k].w << 32), p + 1);
                } else {
                    dst[i] = v[k];
                }
            }
        }
    }
}

template <int BYTES, int UNROLL>
static int run_gather(const uint8_t *d_table, size_t table_bytes, uint32_t *d_out, int blocks, int reps,
                      double *greads_per_s, double *ms_out, int threads = 1024)
{
    const uint32_t n_sectors = (uint32_t)(table_bytes / 64);
    const int iters = 256 / UNROLL;
    hipEvent_t e0, e1;
    STMB_CK(hipEventCreate(&e0));
    STMB_CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < reps + 1; rep++) {
        STMB_CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_gather<BYTES, UNROLL>), dim3(blocks), dim3(threads), 0, 0, d_table, n_sectors, iters, d_out);
        STMB_CK(hipEventRecord(e1));
        STMB_CK(hipEventSynchronize(e1));
        float ms;
        STMB_CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;   // the first launch warms the caches
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double reads = (double)blocks * threads * iters * UNROLL;
    if (greads_per_s) *greads_per_s = reads / best / 1e6;
    if (ms_out) *ms_out = best;
    return 0;
}

}  // namespace stmb

extern "C" {

// Uniformly random reads of `bytes_per_read` (4, 8, 16, 32 or 64) bytes at 64-byte-aligned
// offsets of a `table_bytes` (a multiple of 64) table, `blocks` x 1024 lanes x 256 reads.
// Returns 0 and the best-of-`reps` rate in G reads/s.
int stmb_random_sector_reads(int device, long long table_bytes, int bytes_per_read, int blocks, int reps,
                             double *greads_per_s)
{
    if (table_bytes < 64 || (table_bytes & 63) || table_bytes / 64 > 0xFFFFFFFFLL || blocks < 1 || reps < 1) return 2;
    STMB_CK(hipSetDevice(device));
    uint8_t *d_table = nullptr;
    uint32_t *d_out = nullptr;
    STMB_CK(hipMalloc(&d_table, (size_t)table_bytes));
    STMB_CK(hipMalloc(&d_out, 64));
    STMB_CK(hipMemset(d_table, 1, (size_t)table_bytes));
    int rc;
    switch (bytes_per_read) {
        case 4: rc = stmb::run_gather<4, 4>(d_table, (size_t)table_bytes, d_out, blocks, reps, greads_per_s, nullptr); break;
        case 8: rc = stmb::run_gather<8, 4>(d_table, (size_t)table_bytes, d_out, blocks, reps, greads_per_s, nullptr); break;
        case 16: rc = stmb::run_gather<16, 4>(d_table, (size_t)table_bytes, d_out, blocks, reps, greads_per_s, nullptr); break;
        case 32: rc = stmb::run_gather<32, 4>(d_table, (size_t)table_bytes, d_out, blocks, reps, greads_per_s, nullptr); break;
        case 64: rc = stmb::run_gather<64, 4>(d_table, (size_t)table_bytes, d_out, blocks, reps, greads_per_s, nullptr); break;
        default: rc = 2;
    }
    (void)hipFree(d_table);
    (void)hipFree(d_out);
    return rc;
}

// Streaming copy of `bytes` (multiple of 16): best-of-`reps` GB/s counting read + written bytes.
int stmb_stream_copy(int device, long long bytes, int reps, double *gbytes_per_s)
{
    if (bytes < 16 || (bytes & 15) || reps < 1) return 2;
    STMB_CK(hipSetDevice(device));
    uint4 *src = nullptr, *dst = nullptr;
    STMB_CK(hipMalloc(&src, (size_t)bytes));
    STMB_CK(hipMalloc(&dst, (size_t)bytes));
    STMB_CK(hipMemset(src, 1, (size_t)bytes));
    hipEvent_t e0, e1;
    STMB_CK(hipEventCreate(&e0));
    STMB_CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < reps + 1; rep++) {
        STMB_CK(hipEventRecord(e0));
        hipLaunchKernelGGL(stmb::k_copy, dim3(2048), dim3(1024), 0, 0, src, dst, bytes / 16);
        STMB_CK(hipEventRecord(e1));
        STMB_CK(hipEventSynchronize(e1));
        float ms;
        STMB_CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(dst);
    if (gbytes_per_s) *gbytes_per_s = 2.0 * (double)bytes / best / 1e6;
    return 0;
}

// The same measurement with the launch shape chosen by the caller: `unroll` (4, 8 or 16) independent
// reads in flight per lane, `blocks` x `threads` lanes.  bench.py sweeps these and quotes the best
// rate as the ceiling (a ceiling measured at one shape is only a floor).
int stmb_random_sector_reads_shape(int device, long long table_bytes, int bytes_per_read, int unroll, int blocks,
                                   int threads, int reps, double *greads_per_s)
{
    if (table_bytes < 64 || (table_bytes & 63) || table_bytes / 64 > 0xFFFFFFFFLL || blocks < 1 || reps < 1 || threads < 64 ||
        threads > 1024 || (threads & 63))
        return 2;
    if (bytes_per_read != 32 && bytes_per_read != 4) return 2;
    STMB_CK(hipSetDevice(device));
    uint8_t *d_table = nullptr;
    uint32_t *d_out = nullptr;
    STMB_CK(hipMalloc(&d_table, (size_t)table_bytes));
    STMB_CK(hipMalloc(&d_out, 64));
    STMB_CK(hipMemset(d_table, 1, (size_t)table_bytes));
    int rc = 2;
    const size_t tb = (size_t)table_bytes;
    if (bytes_per_read == 32) {
        if (unroll == 4) rc = stmb::run_gather<32, 4>(d_table, tb, d_out, blocks, reps, greads_per_s, nullptr, threads);
        else if (unroll == 8) rc = stmb::run_gather<32, 8>(d_table, tb, d_out, blocks, reps, greads_per_s, nullptr, threads);
        else if (unroll == 16) rc = stmb::run_gather<32, 16>(d_table, tb, d_out, blocks, reps, greads_per_s, nullptr, threads);
    } else {
        if (unroll == 4) rc = stmb::run_gather<4, 4>(d_table, tb, d_out, blocks, reps, greads_per_s, nullptr, threads);
        else if (unroll == 8) rc = stmb::run_gather<4, 8>(d_table, tb, d_out, blocks, reps, greads_per_s, nullptr, threads);
        else if (unroll == 16) rc = stmb::run_gather<4, 16>(d_table, tb, d_out, blocks, reps, greads_per_s, nullptr, threads);
    }
    (void)hipFree(d_table);
    (void)hipFree(d_out);
    return rc;
}

// Streaming copy with the launch shape chosen by the caller: `unroll` (1, 2, 4, 8) 16-byte words per
// lane per trip, `blocks` x `threads` lanes (blocks = 0: exactly enough blocks for one trip each),
// `nt` = non-temporal loads and stores.  Best-of-`reps` GB/s counting read + written bytes.
int stmb_stream_copy_shape(int device, long long bytes, int reps, int unroll, int blocks, int threads, int nt,
                           double *gbytes_per_s)
{
    if (bytes < 16 || (bytes & 15) || reps < 1 || threads < 64 || threads > 1024 || (threads & 63)) return 2;
    STMB_CK(hipSetDevice(device));
    uint4 *src = nullptr, *dst = nullptr;
    STMB_CK(hipMalloc(&src, (size_t)bytes));
    STMB_CK(hipMalloc(&dst, (size_t)bytes));
    STMB_CK(hipMemset(src, 1, (size_t)bytes));
    const long long n16 = bytes / 16;
    if (blocks <= 0) blocks = (int)((n16 + (long long)threads * unroll - 1) / ((long long)threads * unroll));
    hipEvent_t e0, e1;
    STMB_CK(hipEventCreate(&e0));
    STMB_CK(hipEventCreate(&e1));
    float best = 1e30f;
    int rc = 0;
    for (int rep = 0; rep < reps + 1 && rc == 0; rep++) {
        STMB_CK(hipEventRecord(e0));
#define STMB_COPY(U, N) hipLaunchKernelGGL((stmb::k_copy_shape<U, N>), dim3(blocks), dim3(threads), 0, 0, src, dst, n16)
        if (unroll == 1) { if (nt) STMB_COPY(1, true); else STMB_COPY(1, false); }
        else if (unroll == 2) { if (nt) STMB_COPY(2, true); else STMB_COPY(2, false); }
        else if (unroll == 4) { if (nt) STMB_COPY(4, true); else STMB_COPY(4, false); }
        else if (unroll == 8) { if (nt) STMB_COPY(8, true); else STMB_COPY(8, false); }
        else rc = 2;
#undef STMB_COPY
        STMB_CK(hipEventRecord(e1));
        STMB_CK(hipEventSynchronize(e1));
        float ms;
        STMB_CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(dst);
    if (rc == 0 && gbytes_per_s) *gbytes_per_s = 2.0 * (double)bytes / best / 1e6;
    return rc;
}

}  // extern "C"
