// host_pipe.h -- the host-buffer path of the C ABI: pinned, multi-buffered, zero-copy.
//
// Callers of st_distances_host / st_triangle_host own ordinary (pageable) memory, which
// the HIP runtime copies at ~10 GB/s.  The pipe keeps a few slots of pinned staging memory
// (kPipeSlots of them, one stream each) and a pool of copy threads:
//
//   pack(c)   : caller's pairs  -> pinned (parallel narrowing copy / strided gather: 24 bits per id on trees of
//               fewer than 2^24 nodes, else int32)
//   gpu(c)    : the packed pairs go to a device copy of the slot through the copy engine; one kernel on the
//               slot's stream reads them there and stores its results DIRECTLY into the pinned slot over PCIe:
//               distances as float32 (they are float32 sums), MRCA ids as 24 bits each (or int32)
//   unpack(c) : pinned -> caller's result arrays (parallel widen to float64 / ids to int32)
//
// unpack(c-2) and pack(c+1) run on the CPU while gpu(c-1) and gpu(c) are in flight.  Why zero-copy on the way back
// (scripts/micro/pcie_bench.hip, profiles/pcie_bench_r02.log): per-chunk H2D -> kernel -> D2H sequences on two
// streams fall into lock step (both streams copy in the same direction at the same time) and reach 59 GB/s as the
// sum of both directions, while a kernel that writes its results to pinned host memory beside another slot's copy
// in moves both directions at once.  (Trees served by the tile-sorted kernel's non-lineage-sum forms stage their
// slots in device memory both ways: ensure_device_stage.)
#pragma once
#include <hip/hip_runtime.h>

#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace st {

// A fixed pool of threads that split [0, n) into contiguous ranges.  A host-path call runs
// three or four short parallel phases per chunk (pack, pre-fault, unpack), a few hundred
// microseconds apart, so waking sleeping threads through a condition variable for each of
// them (~50 us a time) would cost as much as the phases themselves on batches of a million
// pairs: workers spin on the generation counter for kSpinMicros after finishing a phase and
// only then go to sleep; the caller spins for the phase's completion.
class CopyPool {
public:
    CopyPool() = default;
    ~CopyPool() { stop(); }
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;

    void start(int n_threads)
    {
        if (!workers_.empty()) return;
        n_ = std::max(1, n_threads);
        quit_.store(false);
        const uint64_t born = generation_.load();     // (a restarted pool does not start from zero)
        const int want = n_;
        try {
            workers_.reserve((size_t)want);
            for (int i = 0; i < want - 1; i++) workers_.emplace_back([this, born] { loop(born); });
        } catch (...) {
            // (thread limit reached, out of memory: the pool works with the threads it has -- the phases
            // are split into n_ parts and wait for n_ - 1 workers, so n_ must be what actually runs)
            stop();
            n_ = 1;
        }
    }

    void stop()
    {
        quit_.store(true);
        generation_.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> g(m_); }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
        workers_.clear();
    }

    // fn(begin, end) over a partition of [0, n) into blocks of `grain` items (a multiple of 1024; 0 = about eight
    // blocks per thread), handed out by an atomic counter: a thread that lands on a busy core or is descheduled
    // for a while (the pool's hosts are shared) takes fewer blocks instead of holding the whole phase up, which a
    // static split into one range per thread did.  Returns when every block is done.
    void parallel_for(int64_t n, const std::function<void(int64_t, int64_t)> &fn, int64_t grain = 0)
    {
        if (n <= 0) return;
        if (workers_.empty() || n < (int64_t)1 << 14) { fn(0, n); return; }
        if (grain <= 0) grain = std::max<int64_t>(4096, (n / (8 * (int64_t)n_) + 1023) / 1024 * 1024);
        fn_ = &fn;
        total_ = n;
        grain_ = grain;
        next_.store(0, std::memory_order_relaxed);
        pending_.store(n_ - 1, std::memory_order_relaxed);
        generation_.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> g(m_); }     // a worker about to sleep has either seen the new generation or is waiting
        cv_.notify_all();
        run_blocks();
        for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0; spins++) {
            if ((spins & 1023) == 1023) std::this_thread::yield();
            else _mm_pause();
        }
        fn_ = nullptr;
    }

    void copy(void *dst, const void *src, int64_t bytes)
    {
        parallel_for(bytes, [=](int64_t b, int64_t e) {
            std::memcpy(static_cast<char *>(dst) + b, static_cast<const char *>(src) + b, (size_t)(e - b));
        });
    }

private:
    static constexpr int kSpinMicros = 200;

    void run_blocks()
    {
        for (;;) {
            const int64_t b = next_.fetch_add(grain_, std::memory_order_relaxed);
            if (b >= total_) return;
            (*fn_)(b, std::min(b + grain_, total_));
        }
    }

    void loop(uint64_t seen)
    {
        for (;;) {
            // spin a little: the next phase of the same call is usually moments away
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; generation_.load(std::memory_order_acquire) == seen; spins++) {
                _mm_pause();
                if ((spins & 255) == 255 &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinMicros)) {
                    std::unique_lock<std::mutex> g(m_);
                    cv_.wait(g, [&] { return generation_.load(std::memory_order_acquire) != seen; });
                    break;
                }
            }
            seen = generation_.load(std::memory_order_acquire);
            if (quit_.load()) return;
            run_blocks();
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }

    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(int64_t, int64_t)> *fn_ = nullptr;
    int64_t total_ = 0, grain_ = 1;
    std::atomic<int64_t> next_{0};
    int n_ = 1;
    std::atomic<int> pending_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> quit_{false};
};

struct PipeSlot {
    void *h_in = nullptr, *h_d = nullptr, *h_m = nullptr;   // pinned host memory, read / written by the kernels
    void *d_in = nullptr;                                   // device copy of h_in  (quartets; tile-sorted kernels: ensure_device_stage)
    void *d_d = nullptr, *d_m = nullptr;                    // device-side result staging (tile-sorted kernels only)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int64_t off = 0, m = 0;
    bool busy = false;
};

// Three slots: while the host unpacks chunk c-2 and packs chunk c+1, chunks c-1 and c are on
// the GPU / the link.  With two, the GPU idled through every unpack + pack + launch of the
// host thread (a third of a million-pair call; SUCHTREE_AMD_TRACE_PIPE shows it as "wait").
constexpr int kPipeSlots = 3;

struct HostPipe {
    PipeSlot slot[kPipeSlots];
    int64_t cap = 0;          // pairs per slot
    void *h_fault = nullptr;  // pinned copy of the tree's fault word, fetched behind the last chunk (16 bytes)
    void *d_ids = nullptr;    // id list of the all-pairs generator
    int64_t ids_cap = 0;
    CopyPool pool;

    hipError_t ensure(int64_t pairs)
    {
        if (pairs <= cap) return hipSuccess;
        release_buffers();
        for (auto &s : slot) {
            hipError_t e;
            if ((e = hipHostMalloc(&s.h_in, (size_t)pairs * 16, hipHostMallocDefault)) != hipSuccess) return e;
            if ((e = hipHostMalloc(&s.h_d, (size_t)pairs * 4, hipHostMallocDefault)) != hipSuccess) return e;   // float32 transport
            if ((e = hipHostMalloc(&s.h_m, (size_t)pairs * 4, hipHostMallocDefault)) != hipSuccess) return e;
            if (!s.stream && (e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess) return e;
            if (!s.done && (e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return e;
        }
        if (!h_fault) {
            const hipError_t e = hipHostMalloc(&h_fault, 64, hipHostMallocDefault);
            if (e != hipSuccess) return e;
        }
        cap = pairs;
        // copy threads: a quarter of the hardware threads, at most 16 (the three memory passes
        // of the host path saturate the host's memory system well before they run out of cores);
        const unsigned hw = std::thread::hardware_concurrency();
        int n_threads = (int)std::min<unsigned>(16, std::max<unsigned>(1, hw / 4));
        pool.start(n_threads);
        return hipSuccess;
    }

    // device-resident copies of the input slots: only the quartet path wants them (every
    // quartet row is read by six lanes; over PCIe that would be six fetches)
    hipError_t ensure_device_in()
    {
        for (auto &s : slot) {
            if (s.d_in) continue;
            const hipError_t e = hipMalloc(&s.d_in, (size_t)cap * 16);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }

    // Device-side staging of a whole slot: kernels that read their pairs twice and store their
    // results in sorted (scattered) order must not do that over PCIe; they work on these
    // buffers and coalesced copy kernels move the slot between pinned and device memory.
    hipError_t ensure_device_stage()
    {
        hipError_t e = ensure_device_in();
        for (auto &s : slot) {
            if (e == hipSuccess && !s.d_d) e = hipMalloc(&s.d_d, (size_t)cap * 4);
            if (e == hipSuccess && !s.d_m) e = hipMalloc(&s.d_m, (size_t)cap * 4);
        }
        return e;
    }

    hipError_t ensure_ids(int64_t n)
    {
        if (n <= ids_cap) return hipSuccess;
        (void)hipFree(d_ids);
        d_ids = nullptr;
        ids_cap = 0;
        hipError_t e = hipMalloc(&d_ids, (size_t)std::max<int64_t>(n, 2) * 8);
        if (e == hipSuccess) ids_cap = n;
        return e;
    }

    void release_buffers()
    {
        for (auto &s : slot) {
            (void)hipHostFree(s.h_in); (void)hipHostFree(s.h_d); (void)hipHostFree(s.h_m);
            (void)hipFree(s.d_in); (void)hipFree(s.d_d); (void)hipFree(s.d_m);
            s.h_in = s.h_d = s.h_m = s.d_in = s.d_d = s.d_m = nullptr;
            s.busy = false;
        }
        cap = 0;
    }

    void destroy()
    {
        release_buffers();
        for (auto &s : slot) {
            if (s.stream) (void)hipStreamDestroy(s.stream);
            if (s.done) (void)hipEventDestroy(s.done);
            s.stream = nullptr;
            s.done = nullptr;
        }
        (void)hipFree(d_ids);
        d_ids = nullptr;
        ids_cap = 0;
        (void)hipHostFree(h_fault);
        h_fault = nullptr;
        pool.stop();
    }
};

}  // namespace st
