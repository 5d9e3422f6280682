// host_path.h -- part of suchtree_hip.hip (included in this order: host_tree.h, host_launch.h, host_path.h,
// host_upload.h).
// The host-buffer path: chunking, the small-batch mailbox, copy kernels, the slot pipeline, multi-device dealing.
#pragma once

template <typename T>
static int upload(T **dst, const std::vector<T> &src, int64_t *bytes)
{
    const size_t sz = std::max<size_t>(src.size() * sizeof(T), 16);
    ST_HIP(hipMalloc(reinterpret_cast<void **>(dst), sz));
    if (!src.empty()) ST_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes += (int64_t)sz;
    return ST_OK;
}

constexpr int64_t kHostChunk = (int64_t)1 << 22;      // most pairs per pipeline chunk
constexpr int64_t kHostChunkMin = (int64_t)1 << 18;   // fewest, when a batch is dealt over several GPUs
constexpr int kDeepCanopyDepth = kShallowCanopyDepth;     // canopies deeper than this (edges) are "deep" (tree_prep.h)
constexpr int kDeepCanopyNodes = 10238;   // the largest ladder image (16 B per node) that leaves k_canopy_ladder its 32 bytes of flags in 160 KiB of LDS
static_assert(st::ladder_kernel_lds_bytes(kDeepCanopyNodes) <= st::kLdsBytesPerCu, "a deep canopy must fit the scalar ladder kernel's LDS");
constexpr int64_t kMaxWalkLineageEntries = (int64_t)1 << 30;   // 4 GiB of lineage sums at most on trees the canopy family refuses
constexpr int64_t kMaxLineageEntries = (int64_t)1 << 28;   // 1 GiB of lineage sums at most (ml.tree: 48 MB)
constexpr int64_t kMailboxPairs = 8192;   // largest batch served through the mailbox (beyond it the staged pipe's fixed ~60 us pay off)

// How a host batch of n pairs is cut into pipeline chunks and dealt over n_dev devices:
// chunk c covers [c*chunk, min(n, (c+1)*chunk)) and belongs to device index c % n_dev.  With
// several devices the chunk shrinks (down to kHostChunkMin) so that every device gets work.
static int64_t host_chunk_pairs(int64_t n, int n_dev)
{
    if (n <= kHostChunkMin) return std::max<int64_t>(n, 1);
    // at least eight chunks per device, so that packing, the link and unpacking overlap even
    // on batches of a few million pairs; never below kHostChunkMin, never above kHostChunk
    int64_t chunk = (n + 8 * (int64_t)n_dev - 1) / (8 * (int64_t)n_dev);
    chunk = std::min(std::max(chunk, kHostChunkMin), kHostChunk);
    return (chunk + 1023) / 1024 * 1024;
}

struct ChunkSeq {
    int64_t n, chunk;
    int first, step;    // this device handles chunks first, first + step, ...
};

// Small batches (a scalar distance(a,b) call is a batch of one) are all latency: instead of
// H2D copy + kernel + D2H copy + fault read-back, the walk kernel reads the pairs from and
// writes the results to pinned host memory mapped into the device, so a call is one launch
// and one stream synchronisation.  Ids are range-checked here on the host (the batch is
// tiny), with the reference's choice of the id to report (MuchTree.pyx:897-903).
template <typename Id>
static int small_batch(st_tree *t, const Id *pairs, int64_t n, int64_t stride0, int64_t stride1,
                       double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    std::lock_guard<std::mutex> lock(t->mb_mutex);
    if (!t->mb_host) {
        // set up in locals and committed to the handle only when every step has succeeded: a partly
        // initialised mailbox (memory but no stream, no block counter) must never be launched on
        const size_t bytes = (size_t)kMailboxPairs * (16 + 8 + 4) + 64;     // + the completion word
        void *host = nullptr, *dev = nullptr;
        Fault *counter = nullptr;
        hipStream_t stream = nullptr;
        hipError_t e = hipHostMalloc(&host, bytes, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, host, 0);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&counter), sizeof(Fault));      // [0]: block counter of the mailbox kernel
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        // cleared ON the mailbox stream: a hipMemset on the null stream is not ordered before
        // kernels of a non-blocking stream, and recycled device memory is not zero
        if (e == hipSuccess) e = hipMemsetAsync(counter, 0, sizeof(Fault), stream);
        if (e != hipSuccess) {
            if (stream) (void)hipStreamDestroy(stream);
            if (counter) (void)hipFree(counter);
            if (host) (void)hipHostFree(host);
            return fail(ST_ERR_HIP, std::string("mailbox setup: ") + hipGetErrorString(e));
        }
        *reinterpret_cast<volatile unsigned *>(static_cast<char *>(host) + (size_t)kMailboxPairs * 28) = 0;
        t->mb_dev = dev;
        t->d_fault_mb = counter;
        t->mb_stream = stream;
        t->mb_host = host;
    }
    int64_t *h_pairs = static_cast<int64_t *>(t->mb_host);
    double *h_dist = reinterpret_cast<double *>(h_pairs + 2 * kMailboxPairs);
    int32_t *h_mrca = reinterpret_cast<int32_t *>(h_dist + kMailboxPairs);
    long long max_id = std::numeric_limits<long long>::min(), min_id = std::numeric_limits<long long>::max();
    for (int64_t k = 0; k < n; k++) {
        const long long a = pairs[k * stride0], b = pairs[k * stride0 + stride1];
        h_pairs[2 * k] = a;
        h_pairs[2 * k + 1] = b;
        max_id = std::max(max_id, std::max(a, b));
        min_id = std::min(min_id, std::min(a, b));
    }
    if (min_id < 0 || max_id >= t->n_nodes) {
        const long long bad = max_id >= t->n_nodes ? max_id : min_id;
        if (bad_id) *bad_id = bad;
        return fail(ST_ERR_BOUNDS, "Node ID " + std::to_string(bad) + " out of bounds (tree size: " +
                                       std::to_string(t->n_nodes) + ")");
    }
    char *d_base = static_cast<char *>(t->mb_dev);
    double *d_dist = reinterpret_cast<double *>(d_base + (size_t)kMailboxPairs * 16);
    int32_t *d_mrca = reinterpret_cast<int32_t *>(d_base + (size_t)kMailboxPairs * 24);
    unsigned *d_done = reinterpret_cast<unsigned *>(d_base + (size_t)kMailboxPairs * 28);
    volatile unsigned *h_done = reinterpret_cast<volatile unsigned *>(static_cast<char *>(t->mb_host) + (size_t)kMailboxPairs * 28);
    unsigned seq = ++t->mb_seq;
    if (seq == 0) seq = ++t->mb_seq;     // (0 is the word's initial value)
    ST_HIP(launch_walk_mailbox(t, reinterpret_cast<const long long *>(d_base), (int)n, out_dist ? d_dist : nullptr,
                               out_mrca ? d_mrca : nullptr, reinterpret_cast<unsigned *>(t->d_fault_mb), d_done, seq, t->mb_stream));
    // poll the completion word (pinned host memory); if it does not show up within a few
    // milliseconds something is wrong: let the runtime report it
    {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(const_cast<unsigned *>(h_done), __ATOMIC_ACQUIRE) != seq) {
            _mm_pause();
            if ((++spins & 4095) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
                ST_HIP(hipStreamSynchronize(t->mb_stream));
                if (__atomic_load_n(const_cast<unsigned *>(h_done), __ATOMIC_ACQUIRE) != seq)
                    return fail(ST_ERR_HIP, "mailbox kernel finished without publishing its completion word");
                break;
            }
        }
    }
    if (out_dist) std::memcpy(out_dist, h_dist, (size_t)n * 8);
    if (out_mrca) std::memcpy(out_mrca, h_mrca, (size_t)n * 4);
    return ST_OK;
}

// Coalesced word copy between pinned host memory and device memory (either direction).
__global__ __launch_bounds__(1024) void k_words_copy(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, long long n_words)
{
    // 16 bytes per lane when both ends are 16-byte aligned (staging slots always are; a caller's
    // pinned result array need not be), else word by word
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const long long n4 = vec ? n_words >> 2 : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) dst[i] = src[i];
}

// Copy kernels run beside the compute kernels of the other slots: a few dozen workgroups keep
// the link busy and leave the CUs to them (with 512 the host path of ml.tree is 10 % slower).
constexpr int64_t kCopyKernelBlocks = 32;

static hipError_t enqueue_words_copy(const void *src, void *dst, int64_t n_words, hipStream_t stream)
{
    if (n_words <= 0) return hipSuccess;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n_words / 4 + 1023) / 1024, kCopyKernelBlocks));
    hipLaunchKernelGGL(k_words_copy, dim3((unsigned)blocks), dim3(1024), 0, stream, static_cast<const uint32_t *>(src),
                       static_cast<uint32_t *>(dst), (long long)n_words);
    return hipGetLastError();
}

// int32 MRCA ids (device) -> the 24-bit wire format (pinned host), coalesced: the staged form of a direct result write
__global__ __launch_bounds__(1024) void k_pack24_copy(const int *__restrict__ src, unsigned char *__restrict__ dst, long long n)
{
    const MrcaSink out{nullptr, dst};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long base = (long long)blockIdx.x * blockDim.x; base < n; base += stride) {
        const long long i = base + threadIdx.x;
        store_mrca_wave(out, i, i < n ? src[i] : 0, i < n);
    }
}

// Where the results of a host-path call go: the caller's arrays (any host memory; results always come back through the
// pinned slots as float32 + 24-bit / int32 ids and are widened into them by the unpack pass.  Round 4 let the kernels
// write float64 + int32 straight into result arrays that were themselves pinned -- 12 bytes per pair over the link
// instead of 7: 2.65e9 pairs/s where the staged form does 4.98e9, profiles/bench_r04_selfrun.json -- removed).
struct HostOut {
    double *dist = nullptr;
    int32_t *mrca = nullptr;
    bool wire24 = false;      // MRCA ids cross the link as 24 bits each (device_common.h::MrcaSink)
};

// MRCA ids come back as 24 bits each on trees of fewer than 2^24 nodes (option wire24)
static bool wire24_of(const st_tree *t, const HostOut &out)
{
    return out.mrca && t->wire24 && t->n_nodes <= 0xFFFFFF;
}

static HostOut make_host_out(double *out_dist, int32_t *out_mrca)
{
    HostOut o;
    o.dist = out_dist;
    o.mrca = out_mrca;
    return o;
}

// The tile-sorted kernel reads every pair twice and stores results in sorted order: fine in
// HBM, ruinous over PCIe (scattered 4-byte writes).  For trees that use it the host path keeps
// the slot in device memory and moves it with the copy kernel above (launch_policy.h: wants_device_stage).

// One chunk of a host-path call on slot s: `make_src(in)` builds the pair source from the
// chunk's input pointer (NULL for generated sources), results go to the slot's pinned arrays.
template <typename MakeSrc>
static int launch_chunk(st_tree *r, PipeSlot &s, int64_t off, int64_t m, int in_bytes_per_pair, const HostOut &out,
                        MakeSrc make_src)
{
    if (!wants_device_stage(r, m) || (!out.dist && mrca_ranks_ready(r))) {
        DistSink sink{nullptr, nullptr};
        if (out.dist) sink.f32 = static_cast<float *>(s.h_d);
        MrcaSink mrca{nullptr, nullptr};
        if (out.mrca) {
            if (out.wire24) mrca.m24 = static_cast<unsigned char *>(s.h_m);
            else mrca.m32 = static_cast<int32_t *>(s.h_m);
        }
        // The chunk's packed pairs come in through the copy engine into a device copy of the slot (the kernel reads them
        // there; its results still go straight to the pinned slot): the way in then runs at the engine's rate beside the
        // kernels of the other slots instead of at what 16 waves per CU keep in flight over the link -- link side alone
        // 5.87 -> 6.50e9 pairs/s (both outputs), whole calls: distances alone +8 % (6.1-6.4 -> 6.6-7.1e9), both outputs
        // even (the CPU passes bind there).
        const void *in = s.h_in;
        if (in_bytes_per_pair) {
            hipError_t e = r->dp->pipe.ensure_device_in();
            if (e == hipSuccess) e = hipMemcpyAsync(s.d_in, s.h_in, (size_t)m * in_bytes_per_pair, hipMemcpyHostToDevice, s.stream);
            if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("device staging: ") + hipGetErrorString(e));
            in = s.d_in;
        }
        return enqueue_src(r, make_src(in), m, sink, mrca, r->d_fault_host, s.stream, false);
    }
    // Pairs come in through the copy engine, results go out through copy kernels: the two
    // directions then overlap and the engine takes no CUs from the tile-sorted kernel (ml.tree,
    // 2e7 pairs, both outputs: input by copy kernel as well 2.7e9 pairs/s, this way 3.7-4.0e9,
    // both directions by the copy engine 3.3-3.5e9).
    hipError_t e = r->dp->pipe.ensure_device_stage();
    if (e == hipSuccess && in_bytes_per_pair)
        e = hipMemcpyAsync(s.d_in, s.h_in, (size_t)m * in_bytes_per_pair, hipMemcpyHostToDevice, s.stream);
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("device staging: ") + hipGetErrorString(e));
    const int rc = enqueue_src(r, make_src(s.d_in), m, DistSink{nullptr, out.dist ? static_cast<float *>(s.d_d) : nullptr},
                               MrcaSink{out.mrca ? static_cast<int32_t *>(s.d_m) : nullptr, nullptr}, r->d_fault_host, s.stream);
    if (rc != ST_OK) return rc;
    if (out.dist) e = enqueue_words_copy(s.d_d, s.h_d, m, s.stream);
    if (e == hipSuccess && out.mrca && out.wire24) {
        hipLaunchKernelGGL(k_pack24_copy, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((m + 1023) / 1024, kCopyKernelBlocks))),
                           dim3(1024), 0, s.stream, static_cast<const int *>(s.d_m), static_cast<unsigned char *>(s.h_m), (long long)m);
        e = hipGetLastError();
    } else if (e == hipSuccess && out.mrca) {
        e = enqueue_words_copy(s.d_m, s.h_m, m, s.stream);
    }
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("device staging: ") + hipGetErrorString(e));
    return ST_OK;
}

// Push this device's chunks of a batch through the slots of the pipe (host_pipe.h).
// pack(slot, off, m) fills slot.h_in for chunk [off, off+m); launch(slot, off, m) enqueues
// the kernel on slot.stream, reading slot.h_in and writing slot.h_d / slot.h_m -- pinned
// host memory, accessed by the kernel over PCIe (see host_pipe.h).  Caller holds the device pipe's mutex.
template <typename Pack, typename Launch>
static int run_pipe(st_tree *t, const ChunkSeq &seq, Pack pack, Launch launch, const HostOut &out, Fault &fault)
{
    fault = kFaultInit;
    // Handle option "measure" (st_tree_set_option; MEASUREMENT ONLY -- a call made with a skip bit set returns
    // ST_ERR_MEASURE_ONLY instead of ST_OK, because its result arrays hold garbage): 1 = trace, one line per call on
    // stderr with the host thread's time by phase; 2 = no pack and no unpack passes (what the GPU / link side of the
    // pipeline takes when the host's memory system is otherwise idle); 4 = the pack / pre-fault / unpack passes alone,
    // nothing launched (what the host side sustains when it never waits for a GPU: the most a multi-device handle,
    // one pipeline per GPU all fed by this host's memory system, can scale to).  Nothing in the environment reaches here.
    const bool trace = (t->measure & 1) != 0;
    const bool skip_cpu = (t->measure & 2) != 0;
    const bool skip_gpu = (t->measure & 4) != 0;
    using Clock = std::chrono::steady_clock;
    double t_wait = 0, t_unpack = 0, t_pack = 0, t_launch = 0, t_prefault = 0;
    const Clock::time_point t_begin = Clock::now();
    auto lap = [&](double &acc, Clock::time_point &since) {
        if (!trace) return;
        const Clock::time_point now = Clock::now();
        acc += std::chrono::duration<double, std::micro>(now - since).count();
        since = now;
    };
    // results the kernels write directly need neither unpacking nor pre-faulting
    double *const out_dist = out.dist;
    int32_t *const out_mrca = out.mrca;
    HostPipe &P = t->dp->pipe;
    {
        const hipError_t e = P.ensure(std::max<int64_t>(seq.chunk, 1024));
        if (e != hipSuccess) {
            P.release_buffers();
            return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
        }
    }
    auto drain = [&](PipeSlot &s) -> hipError_t {
        if (!s.busy) return hipSuccess;
        s.busy = false;
        Clock::time_point tp = trace ? Clock::now() : Clock::time_point();
        if (!skip_gpu) {
            const hipError_t e = hipEventSynchronize(s.done);
            if (e != hipSuccess) return e;
        }
        lap(t_wait, tp);
        // distances crossed PCIe as float32 and are widened into the caller's float64 array;
        // MRCA ids are copied; one pass of the pool over the chunk does both
        const float *src_d = static_cast<const float *>(s.h_d);
        const int32_t *src_m = static_cast<const int32_t *>(s.h_m);
        double *dst_d = out_dist ? out_dist + s.off : nullptr;
        int32_t *dst_m = out_mrca ? out_mrca + s.off : nullptr;
        const bool wire24 = out.wire24;
        if ((dst_d || dst_m) && !skip_cpu)
            P.pool.parallel_for(s.m, [=](int64_t b, int64_t e) {
                if (dst_d) widen_f32_to_f64(dst_d + b, src_d + b, e - b);
                if (dst_m && wire24) unpack_ids24(dst_m + b, reinterpret_cast<const uint8_t *>(src_m), b, e - b);
                else if (dst_m) copy_stream(dst_m + b, src_m + b, (e - b) * 4);
            });
        lap(t_unpack, tp);
        return hipSuccess;
    };
    // pages of a freshly allocated result array are populated here, by the pool, while the
    // chunk is on the GPU -- not one fault at a time inside the unpack loops
    // (only pages that are not there yet: a recycled result array is resident already, and
    // populating resident pages costs more than everything else a mid-sized call does)
    // large fresh arrays: on threads of their own, ahead of the unpack passes (host_copy.h: AsyncPrefault); the
    // per-piece form below then finds the pages resident (or nearly so) and does nothing
    AsyncPrefault async_prefault;      // (joins on every way out of this function)
    bool async_populates = false;      // this call's fresh pages are populated by such threads (of this replica or of the first one)
    {
        constexpr int64_t kAsyncMinBytes = (int64_t)64 << 20;
        double *const pd = out_dist && !looks_resident(out_dist, seq.n * 8) ? out_dist : nullptr;
        int32_t *const pm = out_mrca && !looks_resident(out_mrca, seq.n * 4) ? out_mrca : nullptr;
        const unsigned hw = std::thread::hardware_concurrency();
        // Only with the page-touch form of populate_for_write (transparent huge pages): MADV_POPULATE_WRITE from
        // threads of its own beside the faulting unpack passes measured 3x SLOWER than doing nothing (5e7 pairs, fresh
        // arrays: 1.1-1.6e9 pairs/s; the touch form 4.5-4.9e9; populating between the passes, either form, 3.7-3.9e9;
        // profiles/fresh_array_r04.log).  Multi-device handles: the replica that owns the first chunk populates for all.
        async_populates = thp_available() && !skip_cpu && (pd ? seq.n * 8 : 0) + (pm ? seq.n * 4 : 0) >= kAsyncMinBytes && hw >= 8;
        if (async_populates && seq.first == 0) async_prefault.start(pd, pm, seq.n, (int)std::min<unsigned>(16, hw / 4));
    }
    auto prefault = [&](int64_t off, int64_t m) {
        if (async_populates) return;
        double *const pd = out_dist && !looks_resident(out_dist + off, m * 8) ? out_dist : nullptr;
        int32_t *const pm = out_mrca && !looks_resident(out_mrca + off, m * 4) ? out_mrca : nullptr;
        if (!pd && !pm) return;
        // (blocks of 2^18 pairs: whole huge pages of the float64 array to one thread each -- host_copy.h)
        P.pool.parallel_for(m, [=](int64_t b, int64_t e) {
            if (pd) populate_for_write(pd + off + b, (e - b) * 8);
            if (pm) populate_for_write(pm + off + b, (e - b) * 4);
        }, kPopulateGrainBytes / 8);
    };
    auto bail = [&](int code, const std::string &msg) {
        for (auto &s : P.slot) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            s.busy = false;
        }
        return fail(code, msg);
    };
    // This device's chunks, the first and the last of them cut into pieces of 1/8, 1/8, 1/4, 1/2 (and the mirror
    // image): the GPU has nothing to do while the first piece is packed, and the host nothing to overlap with while
    // the last one is unpacked, so those two are kept small (5e7 pairs in 2^22-pair chunks: ~0.5 of 10 ms).
    struct Piece { int64_t off, m; bool last; };
    std::vector<Piece> pieces;
    {
        std::vector<std::pair<int64_t, int64_t>> chunks;
        for (int64_t c = seq.first; c * seq.chunk < seq.n; c += seq.step)
            chunks.emplace_back(c * seq.chunk, std::min(seq.chunk, seq.n - c * seq.chunk));
        constexpr int64_t kRampMin = (int64_t)1 << 20;      // chunks below this are not worth cutting
        for (size_t q = 0; q < chunks.size(); q++) {
            const int64_t off = chunks[q].first, m = chunks[q].second;
            const bool first = q == 0, last = q + 1 == chunks.size();
            if ((!first && !last) || m < kRampMin) { pieces.push_back({off, m, false}); continue; }
            const int64_t e = (m / 8 + 1023) / 1024 * 1024;
            std::vector<int64_t> cuts;
            if (first && last) cuts = {e, e, 2 * e, 2 * e, e};            // (the rest: one more small piece)
            else if (first) cuts = {e, e, 2 * e};                         // (the rest: about half)
            else cuts = {4 * e, 2 * e, e};                                // (the rest: the last eighth)
            int64_t at = 0;
            for (const int64_t c : cuts) {
                if (at + c >= m) break;
                pieces.push_back({off + at, c, false});
                at += c;
            }
            pieces.push_back({off + at, m - at, false});
        }
        if (!pieces.empty()) pieces.back().last = true;
    }
    int64_t k = 0;
    for (const Piece &pc : pieces) {
        const int64_t off = pc.off, m = pc.m;
        PipeSlot &s = P.slot[k % kPipeSlots];
        hipError_t e = drain(s);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
        Clock::time_point tp = trace ? Clock::now() : Clock::time_point();
        if (!skip_cpu) pack(s, off, m);
        lap(t_pack, tp);
        if (!skip_gpu) {
            const int rc = launch(s, off, m);
            if (rc != ST_OK) return bail(rc, g_last_error);
            if (pc.last) {
                // last piece of this device: fetch the fault word behind it (and behind the pieces
                // still in flight on the other streams), so that one wait covers results and faults
                for (PipeSlot &other : P.slot)
                    if (&other != &s && other.busy && e == hipSuccess) e = hipStreamWaitEvent(s.stream, other.done, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(P.h_fault, t->d_fault_host, sizeof(Fault), hipMemcpyDeviceToHost, s.stream);
                if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
            }
            e = hipEventRecord(s.done, s.stream);
            if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
        }
        s.busy = true;
        s.off = off;
        s.m = m;
        lap(t_launch, tp);
        prefault(off, m);
        lap(t_prefault, tp);
        k++;
        e = drain(P.slot[k % kPipeSlots]);   // unpack the oldest piece while the newer ones are in flight
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
    }
    for (int j = 0; j < kPipeSlots; j++) {     // oldest first
        const hipError_t e = drain(P.slot[(k + j) % kPipeSlots]);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
    }
    if (k > 0 && !skip_gpu) {
        fault = *static_cast<const Fault *>(P.h_fault);
        if (fault.max_bad != kFaultInit.max_bad || fault.min_bad != kFaultInit.min_bad) {     // fired: re-arm
            hipStream_t s0 = P.slot[0].stream;
            ST_HIP(hipMemcpyAsync(t->d_fault_host, &kFaultInit, sizeof(Fault), hipMemcpyHostToDevice, s0));
            ST_HIP(hipStreamSynchronize(s0));
        }
    }
    t->host_fault_dirty = false;
    if (trace)
        std::fprintf(stderr, "[pipe] n %lld chunk %lld chunks %lld total %.1f us: pack %.1f launch %.1f prefault %.1f wait %.1f unpack %.1f\n",
                     (long long)seq.n, (long long)seq.chunk, (long long)k,      // (pieces: the first and last chunk are cut)
                     std::chrono::duration<double, std::micro>(Clock::now() - t_begin).count(), t_pack, t_launch, t_prefault,
                     t_wait, t_unpack);
    if (skip_cpu || skip_gpu) return fail(ST_ERR_MEASURE_ONLY, "handle option 'measure' is set: this call skipped part of the pipeline, its results are not valid");
    return ST_OK;
}

// Run `work(tree, seq, fault)` for every replica of a (possibly multi-device) handle, each
// on its own host thread with its own device's pipe locked, and merge the fault words.
// work returns ST_OK or an error code (message in that thread's g_last_error).
template <typename Work>
static int for_each_replica(st_tree *t, int64_t n, Fault &fault, Work work)
{
    const int n_dev = 1 + (int)t->peers.size();
    const int64_t chunk = host_chunk_pairs(n, n_dev);
    fault = kFaultInit;
    auto one = [&](st_tree *r, int index, Fault &f, std::string &err) -> int {
        DeviceScope scope(r->device);
        if (scope.error() != hipSuccess) {
            err = std::string("hipSetDevice: ") + hipGetErrorString(scope.error());
            return ST_ERR_HIP;
        }
        std::lock_guard<std::mutex> lock(r->dp->m);
        const ChunkSeq seq{n, chunk, index, n_dev};
        f = kFaultInit;
        const int rc = work(r, seq, f);
        if (rc != ST_OK) err = g_last_error;
        return rc;
    };
    if (n_dev == 1) {
        std::string err;
        const int rc = one(t, 0, fault, err);
        return rc == ST_OK ? ST_OK : fail(rc, err);
    }
    std::vector<int> rcs((size_t)n_dev, ST_OK);
    std::vector<Fault> faults((size_t)n_dev, kFaultInit);
    std::vector<std::string> errs((size_t)n_dev);
    std::vector<std::thread> threads;
    for (int d = 1; d < n_dev; d++)
        threads.emplace_back([&, d] { rcs[(size_t)d] = one(t->peers[(size_t)d - 1], d, faults[(size_t)d], errs[(size_t)d]); });
    rcs[0] = one(t, 0, faults[0], errs[0]);
    for (auto &th : threads) th.join();
    for (int d = 0; d < n_dev; d++) {
        if (rcs[(size_t)d] != ST_OK) return fail(rcs[(size_t)d], "device " + std::to_string(d == 0 ? t->device : t->peers[(size_t)d - 1]->device) + ": " + errs[(size_t)d]);
        merge_fault(fault, faults[(size_t)d]);
    }
    return ST_OK;
}
