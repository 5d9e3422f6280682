// host_pipe.h -- the host-buffer path of the C ABI: pinned, double-buffered, overlapped.
//
// Callers of st_distances_host / st_triangle_host own ordinary (pageable) memory, which
// the HIP runtime copies at ~10 GB/s.  The pipe keeps two slots of pinned staging +
// device buffers on two streams and a small pool of copy threads:
//
//   pack(c)   : caller's pairs  -> pinned (parallel memcpy / strided gather)
//   gpu(c)    : H2D, kernel, D2H into pinned, on stream c&1   (async); distances travel
//               as float32 (they are float32 sums), MRCA ids as int32
//   unpack(c) : pinned -> caller's result arrays (parallel widen to float64 / memcpy)
//
// unpack(c-1) and pack(c+1) run on the CPU while gpu(c) is in flight, and the two
// streams let the H2D of one chunk overlap the D2H of the other (PCIe is full duplex).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace st {

// A fixed pool of threads that split [0, n) into contiguous ranges.
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
        for (int i = 0; i < n_ - 1; i++) workers_.emplace_back([this, i] { loop(i + 1); });
    }

    void stop()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
        workers_.clear();
        quit_ = false;
    }

    // fn(begin, end) over a partition of [0, n); returns when every part is done.
    void parallel_for(int64_t n, const std::function<void(int64_t, int64_t)> &fn)
    {
        if (n <= 0) return;
        if (workers_.empty() || n < (int64_t)1 << 14) { fn(0, n); return; }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            total_ = n;
            pending_ = n_ - 1;
            ++generation_;
        }
        cv_.notify_all();
        run_part(0);
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

    void copy(void *dst, const void *src, int64_t bytes)
    {
        parallel_for(bytes, [=](int64_t b, int64_t e) {
            std::memcpy(static_cast<char *>(dst) + b, static_cast<const char *>(src) + b, (size_t)(e - b));
        });
    }

private:
    void run_part(int part)
    {
        const int64_t b = total_ * part / n_, e = total_ * (part + 1) / n_;
        if (e > b) (*fn_)(b, e);
    }

    void loop(int part)
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (quit_) return;
            }
            run_part(part);
            {
                std::lock_guard<std::mutex> g(m_);
                --pending_;
            }
            done_cv_.notify_one();
        }
    }

    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int64_t, int64_t)> *fn_ = nullptr;
    int64_t total_ = 0;
    int n_ = 1, pending_ = 0;
    uint64_t generation_ = 0;
    bool quit_ = false;
};

struct PipeSlot {
    void *h_in = nullptr, *h_d = nullptr, *h_m = nullptr;   // pinned
    void *d_in = nullptr, *d_d = nullptr, *d_m = nullptr;   // device
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int64_t off = 0, m = 0;
    bool busy = false;
};

struct HostPipe {
    PipeSlot slot[2];
    int64_t cap = 0;          // pairs per slot
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
            if ((e = hipMalloc(&s.d_in, (size_t)pairs * 16)) != hipSuccess) return e;
            if ((e = hipMalloc(&s.d_d, (size_t)pairs * 4)) != hipSuccess) return e;
            if ((e = hipMalloc(&s.d_m, (size_t)pairs * 4)) != hipSuccess) return e;
            if (!s.stream && (e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess) return e;
            if (!s.done && (e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return e;
        }
        cap = pairs;
        const unsigned hw = std::thread::hardware_concurrency();
        pool.start((int)std::min<unsigned>(16, std::max<unsigned>(1, hw / 4)));
        return hipSuccess;
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
        pool.stop();
    }
};

}  // namespace st
