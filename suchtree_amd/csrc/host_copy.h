// host_copy.h -- the CPU side of the host-buffer path: the three memory passes that
// every pair pays on the host (pack ids -> pinned, unpack distances, unpack MRCA ids)
// and the pre-faulting of freshly allocated result arrays.
//
// The host path is CPU-memory-bound (the GPU and PCIe have headroom),
// so each pass moves as few bytes as it can: results are written with non-temporal
// stores (no read-for-ownership of lines that are overwritten whole), ids are narrowed
// to int32 while they are copied, and pages of a fresh result array are populated by
// the copy pool (MADV_POPULATE_WRITE on huge-page-advised ranges) while the GPU works,
// instead of one 4 KiB fault at a time inside the unpack loop.
#pragma once
#include <atomic>
#include <cerrno>
#include <emmintrin.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23   // Linux 5.14
#endif

namespace st {

// The wide forms below are compiled for AVX2 whatever the build's baseline (function-level target attribute) and
// chosen at run time; the SSE2 forms stay as the fallback.
inline bool cpu_has_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}

__attribute__((target("avx2"))) inline int64_t widen_f32_to_f64_avx2(double *dst, const float *src, int64_t k, int64_t n)
{
    while (k < n && (reinterpret_cast<uintptr_t>(dst + k) & 31)) { dst[k] = (double)src[k]; k++; }
    for (; k + 8 <= n; k += 8) {
        const __m256 v = _mm256_loadu_ps(src + k);
        _mm256_stream_pd(dst + k, _mm256_cvtps_pd(_mm256_castps256_ps128(v)));
        _mm256_stream_pd(dst + k + 4, _mm256_cvtps_pd(_mm256_extractf128_ps(v, 1)));
    }
    return k;
}

// float32 -> float64, dst written with streaming stores.
inline void widen_f32_to_f64(double *dst, const float *src, int64_t n)
{
    int64_t k = 0;
    if (cpu_has_avx2()) k = widen_f32_to_f64_avx2(dst, src, 0, n);
    while (k < n && (reinterpret_cast<uintptr_t>(dst + k) & 15)) { dst[k] = (double)src[k]; k++; }
    for (; k + 4 <= n; k += 4) {
        const __m128 v = _mm_loadu_ps(src + k);
        _mm_stream_pd(dst + k, _mm_cvtps_pd(v));
        _mm_stream_pd(dst + k + 2, _mm_cvtps_pd(_mm_movehl_ps(v, v)));
    }
    for (; k < n; k++) dst[k] = (double)src[k];
    _mm_sfence();
}

// 24-bit ids (device_common.h::MrcaSink: pair i at bytes [3 i, 3 i + 3), 0xFFFFFF = -1) -> int32, ids
// [first, first + n) of the packed stream `src`; dst (the caller's array, already offset to `first`) written with
// streaming stores.  The packed buffer is readable for at least 4 bytes past its last id (a staging slot of 4 bytes
// per pair).
__attribute__((target("avx2"))) inline int64_t unpack_ids24_avx2(int32_t *dst, const uint8_t *src, int64_t k, int64_t n)
{
    while (k < n && (reinterpret_cast<uintptr_t>(dst + k) & 31)) {
        uint32_t w;
        std::memcpy(&w, src + 3 * k, 4);
        w &= 0xFFFFFFu;
        dst[k] = w == 0xFFFFFFu ? -1 : (int32_t)w;
        k++;
    }
    const __m256i pick = _mm256_setr_epi8(0, 1, 2, -1, 3, 4, 5, -1, 6, 7, 8, -1, 9, 10, 11, -1,
                                          0, 1, 2, -1, 3, 4, 5, -1, 6, 7, 8, -1, 9, 10, 11, -1);
    const __m256i none = _mm256_set1_epi32(0xFFFFFF);
    for (; k + 10 <= n; k += 8) {      // (16-byte loads at 3 k and 3 k + 12: the last one ends 4 bytes past id k + 7)
        const __m128i lo = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 3 * k));
        const __m128i hi = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 3 * k + 12));
        __m256i v = _mm256_shuffle_epi8(_mm256_inserti128_si256(_mm256_castsi128_si256(lo), hi, 1), pick);
        v = _mm256_or_si256(v, _mm256_slli_epi32(_mm256_cmpeq_epi32(v, none), 24));      // 0xFFFFFF -> -1
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k), v);
    }
    return k;
}

inline void unpack_ids24(int32_t *dst, const uint8_t *src, int64_t first, int64_t n)
{
    const uint8_t *p = src + 3 * first;
    int64_t k = 0;
    if (cpu_has_avx2()) k = unpack_ids24_avx2(dst, p, 0, n);
    for (; k < n; k++) {
        uint32_t w;
        std::memcpy(&w, p + 3 * k, 4);
        w &= 0xFFFFFFu;
        dst[k] = w == 0xFFFFFFu ? -1 : (int32_t)w;
    }
    _mm_sfence();
}

// plain copy with streaming stores (dst is written once and read much later)
inline void copy_stream(void *dst_, const void *src_, int64_t bytes)
{
    char *dst = static_cast<char *>(dst_);
    const char *src = static_cast<const char *>(src_);
    int64_t k = 0;
    const int64_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
    if (head && head <= bytes) { std::memcpy(dst, src, (size_t)head); k = head; }
    for (; k + 64 <= bytes; k += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k + 32));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + k), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + k + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + k + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + k + 48), d);
    }
    if (k < bytes) std::memcpy(dst + k, src + k, (size_t)(bytes - k));
    _mm_sfence();
}

// C-order int64 (m,2) -> int32 (m,2).  Values outside int32 are clamped to INT32_MAX /
// INT32_MIN (still out of range for the kernel) and their exact extremes are reported in
// hi / lo, so the id the caller finally reports is the reference's (MuchTree.pyx:897-903).
inline void narrow_pairs_i64(int32_t *dst, const int64_t *src, int64_t m, long long &hi, long long &lo)
{
    const int64_t n = 2 * m;          // scalars
    constexpr int64_t kBlock = 2048;  // scalars per checked block
    for (int64_t base = 0; base < n; base += kBlock) {
        const int64_t end = base + kBlock < n ? base + kBlock : n;
        int64_t k = base;
        __m128i bad = _mm_setzero_si128();
        if ((reinterpret_cast<uintptr_t>(dst + k) & 15) == 0) {
            for (; k + 4 <= end; k += 4) {
                const __m128 v0 = _mm_castsi128_ps(_mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k)));
                const __m128 v1 = _mm_castsi128_ps(_mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k + 2)));
                const __m128i low = _mm_castps_si128(_mm_shuffle_ps(v0, v1, _MM_SHUFFLE(2, 0, 2, 0)));
                const __m128i high = _mm_castps_si128(_mm_shuffle_ps(v0, v1, _MM_SHUFFLE(3, 1, 3, 1)));
                // fits int32  <=>  high word == sign extension of the low word
                bad = _mm_or_si128(bad, _mm_xor_si128(high, _mm_srai_epi32(low, 31)));
                _mm_stream_si128(reinterpret_cast<__m128i *>(dst + k), low);
            }
        }
        const bool redo = _mm_movemask_epi8(_mm_cmpeq_epi32(bad, _mm_setzero_si128())) != 0xFFFF;
        for (int64_t q = redo ? base : k; q < end; q++) {
            const long long v = src[q];
            int32_t w = (int32_t)v;
            if (v > INT32_MAX) { w = INT32_MAX; if (v > hi) hi = v; }
            else if (v < INT32_MIN) { w = INT32_MIN; if (v < lo) lo = v; }
            dst[q] = w;
        }
    }
    _mm_sfence();
}

// (m,2) ids, element strides s0 / s1 -> 24 bits per id, 6 bytes per pair (the wire format of trees with fewer than
// 2^24 nodes; device_common.h::SrcContig32::packed48).  `first` = index of the first pair in the destination (the
// packed stream is written in 8-byte words where it can).  Ids outside [0, n_nodes) become 0xFFFFFF -- out of range
// for the kernel too, which answers NaN / -1 -- and hi / lo keep the largest id >= n_nodes and the smallest negative
// one: in this format the host is the judge of the range (the reference's choice of the id to report,
// MuchTree.pyx:897-903, is made from these).
// Four pairs of a C-order (m,2) array at once (AVX2): 0 if any of the eight ids is outside [0, n_nodes) -- the
// caller then takes the scalar form for these four, which also records the offender -- else their 24 packed bytes.
__attribute__((target("avx2"))) inline bool pack4_pairs48_i64(const int64_t *src, long long n_nodes, uint64_t out[3])
{
    const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src));          // a0 b0 a1 b1
    const __m256i v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 4));      // a2 b2 a3 b3
    const __m256i top = _mm256_set1_epi64x(n_nodes - 1), zero = _mm256_setzero_si256();
    const __m256i bad = _mm256_or_si256(_mm256_or_si256(_mm256_cmpgt_epi64(v0, top), _mm256_cmpgt_epi64(zero, v0)),
                                        _mm256_or_si256(_mm256_cmpgt_epi64(v1, top), _mm256_cmpgt_epi64(zero, v1)));
    if (!_mm256_testz_si256(bad, bad)) return false;
    const __m256i pick = _mm256_setr_epi8(0, 1, 2, 8, 9, 10, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          0, 1, 2, 8, 9, 10, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i w0 = _mm256_shuffle_epi8(v0, pick), w1 = _mm256_shuffle_epi8(v1, pick);
    const uint64_t p0 = (uint64_t)_mm256_extract_epi64(w0, 0), p1 = (uint64_t)_mm256_extract_epi64(w0, 2);
    const uint64_t p2 = (uint64_t)_mm256_extract_epi64(w1, 0), p3 = (uint64_t)_mm256_extract_epi64(w1, 2);
    out[0] = p0 | (p1 << 48);
    out[1] = (p1 >> 16) | (p2 << 32);
    out[2] = (p2 >> 32) | (p3 << 16);
    return true;
}

__attribute__((target("avx2"))) inline bool pack4_pairs48_i32(const int32_t *src, long long n_nodes, uint64_t out[3])
{
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src));           // a0 b0 a1 b1 | a2 b2 a3 b3
    const __m256i top = _mm256_set1_epi32((int)(n_nodes - 1)), zero = _mm256_setzero_si256();
    const __m256i bad = _mm256_or_si256(_mm256_cmpgt_epi32(v, top), _mm256_cmpgt_epi32(zero, v));
    if (!_mm256_testz_si256(bad, bad)) return false;
    const __m256i pick = _mm256_setr_epi8(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1,
                                          0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1);
    const __m256i w = _mm256_shuffle_epi8(v, pick);
    const uint64_t l0 = (uint64_t)_mm256_extract_epi64(w, 0), l1 = (uint64_t)_mm256_extract_epi64(w, 1) & 0xFFFFFFFFull;
    const uint64_t h0 = (uint64_t)_mm256_extract_epi64(w, 2), h1 = (uint64_t)_mm256_extract_epi64(w, 3) & 0xFFFFFFFFull;
    out[0] = l0;
    out[1] = l1 | (h0 << 32);
    out[2] = (h0 >> 32) | (h1 << 32);
    return true;
}

template <typename Id>
inline void pack_pairs48(uint8_t *dst, int64_t first, const Id *src, int64_t m, int64_t s0, int64_t s1, long long n_nodes,
                         long long &hi, long long &lo)
{
    auto one = [&](int64_t k) -> uint64_t {      // pair k as 48 bits
        uint64_t w = 0;
        for (int c = 0; c < 2; c++) {
            const long long v = (long long)src[k * s0 + c * s1];
            uint64_t id = (uint64_t)v;
            if ((unsigned long long)v >= (unsigned long long)n_nodes) {
                id = 0xFFFFFFu;
                if (v < 0) { if (v < lo) lo = v; }
                else if (v > hi) hi = v;
            }
            w |= id << (24 * c);
        }
        return w;
    };
    int64_t k = 0;
    uint8_t *out = dst + 6 * first;
    // head: up to the next pair whose byte offset is a multiple of 24 (four pairs = three 8-byte words)
    for (; k < m && ((first + k) & 3) != 0; k++) {
        const uint64_t w = one(k);
        std::memcpy(out + 6 * k, &w, 6);
    }
    const bool wide = cpu_has_avx2() && s0 == 2 && s1 == 1 && n_nodes >= 1 && n_nodes <= 0xFFFFFF;
    for (; k + 4 <= m; k += 4) {
        long long *q = reinterpret_cast<long long *>(out + 6 * k);      // 8-byte aligned: dst is 16-byte aligned, 6 (first + k) is a multiple of 24
        uint64_t w[3];
        bool done = false;
        if (wide) {
            if (sizeof(Id) == 8) done = pack4_pairs48_i64(reinterpret_cast<const int64_t *>(src) + 2 * k, n_nodes, w);
            else done = pack4_pairs48_i32(reinterpret_cast<const int32_t *>(src) + 2 * k, n_nodes, w);
        }
        if (!done) {
            const uint64_t p0 = one(k), p1 = one(k + 1), p2 = one(k + 2), p3 = one(k + 3);
            w[0] = p0 | (p1 << 48);
            w[1] = (p1 >> 16) | (p2 << 32);
            w[2] = (p2 >> 32) | (p3 << 16);
        }
        _mm_stream_si64(q + 0, (long long)w[0]);
        _mm_stream_si64(q + 1, (long long)w[1]);
        _mm_stream_si64(q + 2, (long long)w[2]);
    }
    for (; k < m; k++) {
        const uint64_t w = one(k);
        std::memcpy(out + 6 * k, &w, 6);
    }
    _mm_sfence();
}

// Transparent huge pages available to madvise(MADV_HUGEPAGE) ranges ("always" or "madvise" in sysfs)?
inline bool thp_available()
{
    static const bool yes = [] {
        FILE *f = std::fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        if (!f) return false;
        char buf[128] = {0};
        const size_t got = std::fread(buf, 1, sizeof buf - 1, f);
        std::fclose(f);
        buf[got] = 0;
        return std::strstr(buf, "[always]") != nullptr || std::strstr(buf, "[madvise]") != nullptr;
    }();
    return yes;
}

// Make [p, p + bytes) resident and writable without taking one page fault per 4 KiB inside
// the copy loops.  Safe on any memory the caller is about to overwrite anyway.
// Two forms, by what the host offers (scripts/micro/populate_bench.cpp, profiles/populate_bench_r04.log: 600 MB on
// 16 threads of the pool's hosts): with transparent huge pages one store per 4 KiB page -- the first one of a huge
// page faults in (zeroes) all 2 MiB of it, the rest find it there: 2.4 ms = 258 GB/s -- where MADV_POPULATE_WRITE
// takes 18.7 ms (34 GB/s: it does not scale past 4 threads there); without huge pages MADV_POPULATE_WRITE (Linux
// 5.14), else the stores.  Callers hand blocks of whole huge pages to one thread each (kPopulateGrainBytes).
constexpr int64_t kPopulateGrainBytes = (int64_t)2 << 20;

inline void populate_for_write(void *p, int64_t bytes)
{
    static const long page = sysconf(_SC_PAGESIZE);
    const uintptr_t b = (reinterpret_cast<uintptr_t>(p) + (uintptr_t)page - 1) & ~((uintptr_t)page - 1);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes) & ~((uintptr_t)page - 1);
    if (e <= b) return;
    // 1 = not probed yet / available, 0 = this kernel does not know MADV_POPULATE_WRITE (EINVAL on
    // the FIRST call ever: Linux < 5.14).  Any other failure (a file-backed or hugetlb range, ENOMEM,
    // a later EINVAL) concerns this range only: nothing is touched -- the copy loops then take their
    // page faults -- and the next call tries again.  Called from all copy threads: atomic.
    static std::atomic<int> have_populate{1};
    static std::atomic<int> probed{0};
    if (!thp_available() && have_populate.load(std::memory_order_relaxed)) {
        const int rc = madvise(reinterpret_cast<void *>(b), e - b, MADV_POPULATE_WRITE);
        const int err = rc == 0 ? 0 : errno;
        const bool first = probed.exchange(1, std::memory_order_relaxed) == 0;
        if (rc == 0) return;
        if (!(first && err == EINVAL)) return;
        have_populate.store(0, std::memory_order_relaxed);
    }
    // touch one byte per page, leaving its value as it is (the range belongs to the caller): a locked OR with zero
    // is one access with write intent -- one write fault per page, where `*c = *c` would take a read fault (mapping
    // the shared zero page) and then a write fault
    // (inline assembly: the compiler turns an idempotent __atomic_fetch_or(p, 0) into a fenced LOAD, which takes the
    // read fault after all -- measured: "populated" 600 MB in 0.14 ms, and the unpack passes then paid 5 ms of faults)
    for (uintptr_t q = b; q < e; q += (uintptr_t)page)
        asm volatile("lock; orb $0, %0" : "+m"(*reinterpret_cast<unsigned char *>(q)) : : "cc");
}

// Are the pages of [p, p + bytes) already resident?  Judged from its first, middle and last
// page (mincore): result arrays are either fresh from mmap (nothing resident: populate them)
// or recycled by the allocator / the caller (everything resident: MADV_POPULATE_WRITE would
// still walk every page, several hundred microseconds per million pairs).  A wrong guess on a
// partly touched range only costs ordinary page faults in the unpack loops.
inline bool looks_resident(const void *p, int64_t bytes)
{
    static const long page = sysconf(_SC_PAGESIZE);
    if (bytes <= 0) return true;
    const uintptr_t first = reinterpret_cast<uintptr_t>(p) & ~((uintptr_t)page - 1);
    const uintptr_t last = (reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes - 1) & ~((uintptr_t)page - 1);
    const uintptr_t mid = (first + (last - first) / 2) & ~((uintptr_t)page - 1);
    for (uintptr_t q : {first, mid, last}) {
        unsigned char vec = 0;
        if (mincore(reinterpret_cast<void *>(q), (size_t)page, &vec) != 0 || !(vec & 1)) return false;
    }
    return true;
}

// Populates the pages of freshly allocated result arrays on threads of its own while the call's pipeline runs: the
// first touch of 600 MB (5e7 pairs: float64 + int32) is 5 ms of page zeroing by the kernel, which the copy pool
// used to do chunk by chunk BETWEEN its pack and unpack passes -- half of the host thread's time in a call that
// returns fresh arrays.  Here K threads walk the arrays front to back in blocks of 2^18 pairs (thread t: blocks t,
// t + K, ...), so the populated frontier stays ahead of the unpack passes, which start two chunks into the call; an
// unpack pass that does catch up takes ordinary page faults for a while (populating a page twice is harmless).
// join() before the call returns: the arrays are the caller's.
class AsyncPrefault {
public:
    AsyncPrefault() = default;
    AsyncPrefault(const AsyncPrefault &) = delete;
    AsyncPrefault &operator=(const AsyncPrefault &) = delete;
    ~AsyncPrefault() { join(); }

    // dist (8 bytes per pair) and / or mrca (4 bytes per pair), n pairs; either may be NULL
    void start(double *dist, int32_t *mrca, int64_t n, int n_threads)
    {
        if ((!dist && !mrca) || n <= 0 || n_threads < 1) return;
        constexpr int64_t kBlock = (int64_t)1 << 18;
        const int64_t blocks = (n + kBlock - 1) / kBlock;
        const int k = (int)std::min<int64_t>(n_threads, blocks);
        try {
            for (int t = 0; t < k; t++)
                threads_.emplace_back([=] {
                    for (int64_t b = t; b < blocks; b += k) {
                        const int64_t lo = b * kBlock, m = std::min(kBlock, n - lo);
                        if (dist) populate_for_write(dist + lo, m * 8);
                        if (mrca) populate_for_write(mrca + lo, m * 4);
                    }
                });
        } catch (...) {      // (thread limit: the unpack passes take the page faults themselves)
        }
    }

    bool active() const { return !threads_.empty(); }

    void join()
    {
        for (auto &t : threads_) t.join();
        threads_.clear();
    }

private:
    std::vector<std::thread> threads_;
};

inline void advise_huge(void *p, int64_t bytes)
{
    const uintptr_t huge = (uintptr_t)2 << 20;
    const uintptr_t b = (reinterpret_cast<uintptr_t>(p) + huge - 1) & ~(huge - 1);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes) & ~(huge - 1);
    if (e > b) (void)madvise(reinterpret_cast<void *>(b), e - b, MADV_HUGEPAGE);
}

}  // namespace st
