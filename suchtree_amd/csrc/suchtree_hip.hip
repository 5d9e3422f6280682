// suchtree_hip.hip -- gfx950 kernels and the C ABI of libsuchtree_hip.so.
//
// Hot path replaced: SuchTree._distances + SuchTree._mrca
// (/root/reference/SuchTree/MuchTree.pyx:911-943, 999-1030).  One wavefront
// lane per node pair.  Two kernel families, bit-identical results:
//
//   walk    pointer chase over the 8-byte {parent,dist} table with a depth
//           cut; works for any rooted tree.
//   canopy  the top of the tree lives in LDS (BFS-numbered); everything below it is
//           folded into one fixed-stride understory record per node (split into an
//           8-byte a-side table and a b-side table), so a pair costs two record reads
//           plus an LDS climb instead of ~h dependent global gathers.
//           k_canopy_ilp     1-2 pairs per lane, predicated (the default)
//           k_canopy_sorted  deep canopies: ladder form of the canopy (three edges per
//                            16-byte LDS entry), pairs sorted by climb length within a
//                            workgroup tile; with in-order ids the meeting node comes from
//                            a sparse table and a's side from per-node lineage sums
//           k_canopy         scalar, branchy (records longer than 128 bytes)
//
// Every kernel is templated on a pair source (SrcContig / SrcContig32 / SrcStrided /
// SrcTriangle / SrcGrid / SrcQuartet): an explicit (n,2) array, or pairs derived from their
// index (all-pairs triangle, rows x columns grid, the six pairs of a quartet).
//
// The kernels live in headers included below, in this order: device_common.h (fault word, pair
// sources, result sinks), kernels_walk.h, kernels_canopy.h, kernels_misc.h; this file holds the
// error plumbing, the copy kernels of the host pipe and everything host-side.
//
// Host side of the C ABI: tree upload to one or several GPUs (tree_prep.cpp builds the
// tables), the zero-copy host path (host_pipe.h, host_copy.h: kernels read and write pinned
// host memory), the small-batch mailbox, fault read-back, k-nearest selection, graph matrices.
//
// Built for gfx950 only: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/suchtree_hip.h"
#include "host_copy.h"
#include "host_pipe.h"
#include "pair_math.h"
#include "tree_prep.h"

namespace st {

// --------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define ST_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail(ST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Selects a device for the scope of one C-ABI call and puts the caller's current device
// back afterwards (callers such as PyTorch keep their own notion of "current device").
class DeviceScope {
public:
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        err_ = (prev_ == device) ? hipSuccess : hipSetDevice(device);
        changed_ = err_ == hipSuccess && prev_ != device;
    }
    ~DeviceScope()
    {
        if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
    }
    hipError_t error() const { return err_; }

private:
    int prev_ = -1;
    hipError_t err_ = hipSuccess;
    bool changed_ = false;
};

#define ST_DEVICE(dev)                                                                     \
    DeviceScope device_scope_(dev);                                                        \
    if (device_scope_.error() != hipSuccess)                                               \
        return fail(ST_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_scope_.error()))

}  // namespace st

#include "device_common.h"
#include "kernels_walk.h"
#include "kernels_canopy.h"
#include "kernels_misc.h"


// --------------------------------------------------------------------------
// host side: the tree handle
// --------------------------------------------------------------------------
using namespace st;

// One staging pipe (pinned + device buffers, one stream per slot, copy pool) per GPU, shared by every
// tree of the process on that GPU: SuchLinkedTrees holds two trees, applications hold many,
// and the staging is ~200 MB of pinned memory and up to 15 threads per pipe.  Reference
// counted; the mutex admits one host-path call at a time per device.
struct DevicePipe {
    std::mutex m;
    HostPipe pipe;
    int refs = 0;
};
static std::mutex g_pipes_mutex;
static std::map<int, DevicePipe *> g_pipes;

static DevicePipe *pipe_acquire(int device)
{
    std::lock_guard<std::mutex> g(g_pipes_mutex);
    DevicePipe *&p = g_pipes[device];
    if (!p) p = new DevicePipe();
    p->refs++;
    return p;
}

static void pipe_release(int device)
{
    std::lock_guard<std::mutex> g(g_pipes_mutex);
    auto it = g_pipes.find(device);
    if (it == g_pipes.end()) return;
    if (--it->second->refs > 0) return;
    {
        DeviceScope scope(device);
        it->second->pipe.destroy();
    }
    delete it->second;
    g_pipes.erase(it);
}

struct st_tree {
    int device = 0;
    int strategy = ST_STRATEGY_WALK;       // family in use
    bool has_canopy = false;
    int n_cu = 256;
    st_tree_info info{};
    // device tables
    Node8 *d_nodes = nullptr;
    int32_t *d_depth = nullptr;
    Stride3 *d_stride = nullptr;
    uint64_t *d_tree_rmq = nullptr;   // whole-tree sparse table (in-order ids, small trees), else NULL
    CanopyEntry *d_canopy = nullptr;
    int32_t *d_canopy_id = nullptr;
    uint8_t *d_rec_a = nullptr, *d_rec_b = nullptr, *d_rec_i = nullptr;
    uint8_t *d_rec_p = nullptr;       // lineage sums (deep canopies with a sparse table), else NULL
    uint64_t *d_rmq64 = nullptr;
    uint32_t *d_rec_r = nullptr;      // MRCA-only queries (in-order ids), else NULL
    float *d_lineage = nullptr;
    // two fault words: the device-pointer entry points are not serialised against anything,
    // so the host path keeps its own (reset at the start of every host call, read under the
    // device pipe's mutex) and is never confused by a caller who skipped st_fault_check
    Fault *d_fault = nullptr;        // st_distances_device / st_triangle_device / st_fault_check
    Fault *d_fault_host = nullptr;   // st_*_host
    bool host_fault_dirty = false;   // a host call ended before reading its fault word back: re-arm it first
    // canopy geometry
    int32_t canopy_nodes = 0, rec_bytes = 0, rec_cap = 0, parity = 0;
    int64_t n_nodes = 0, n_leaves = 0;
    int pairs_per_lane = 1;   // tuning: 0 = scalar (branchy) kernel, 1/2 = predicated ILP kernel with that many pairs per lane
    int tile_sort = 0;        // tuning: 1 = tile-sorted kernel over the ladder form of the canopy (default for deep canopies)
    int mrca_ranks = 1;       // tuning: 0 = MRCA-only requests go through the distance kernels
    int lineage_sums = 1;     // tuning: 0 = the tile-sorted kernel climbs a's canopy lineage even when the lineage-sum table exists
    LadderEntry *d_ladder = nullptr;
    uint16_t *d_cdepth = nullptr;
    uint16_t *d_cpos = nullptr;     // sparse table for the meeting node (in-order ids only)
    uint32_t *d_rmq = nullptr;
    int canopy_depth = 0;     // deepest canopy node (edges)
    int small_batch_path = 1; // tuning: batches <= kMailboxPairs go through the pinned mailbox
    // staging of the host entry points: the device's shared pipe
    DevicePipe *dp = nullptr;
    void *q_tmp = nullptr;        // MRCA ids of the quartet path (6 int32 per quartet)
    int64_t q_tmp_cap = 0;
    // mailbox of the small-batch path: pinned host memory the kernel reads and writes directly
    std::mutex mb_mutex;
    void *mb_host = nullptr;      // [pairs int64 x2 | dist double | mrca int32] x kMailboxPairs
    void *mb_dev = nullptr;       // device alias of mb_host
    Fault *d_fault_mb = nullptr;  // 16 device bytes of that path: the mailbox kernel's block counter
    unsigned mb_seq = 0;          // sequence number of the last mailbox call (its completion word)
    hipStream_t mb_stream = nullptr;
    // multi-device handle (st_tree_create_multi): replicas of this tree on the other devices.
    // Host-path calls deal their chunks over {this, peers...}; everything else uses this tree.
    std::vector<st_tree *> peers;
};

static const Fault kFaultInit = {std::numeric_limits<long long>::min(),
                                 std::numeric_limits<long long>::max()};

static size_t canopy_lds_bytes(const st_tree *t)
{
    return (size_t)((t->canopy_nodes + 1) / 2) * 16;
}

template <typename Kern, typename Src>
static hipError_t launch_canopy_k(Kern kern, int ppl, const st_tree *t, const CanopyParams &P,
                                  const Src &src, int64_t n, DistSink out_d, int32_t *out_m,
                                  Fault *fault, hipStream_t stream, size_t lds = 0)
{
    if (lds == 0) lds = canopy_lds_bytes(t);
    if (lds > 64 * 1024) {
        // dynamic LDS above 64 KiB has to be granted per kernel (cheap host-side call)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // one or two 1024-lane workgroups per CU, whatever the LDS image allows
    const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int64_t tile = (int64_t)kCanopyBlock * ppl;
    int64_t blocks = (n + tile - 1) / tile;
    blocks = std::min<int64_t>(blocks, (int64_t)t->n_cu * wg_per_cu);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kCanopyBlock), lds, stream, P, src,
                       (long long)n, out_d, out_m, fault);
    return hipGetLastError();
}

// Shape of the tile-sorted launch: pairs per lane (2 when image + scratch fit half the LDS, i.e.
// two workgroups per CU; else 4 with one workgroup per CU) and whether the meeting nodes come
// from the sparse table (in-order ids, and the extra 4 bytes per pair of scratch still leave
// room for the same tile) or from the lock-step search.  q = 0: the ladder image does not fit.
struct SortedShape {
    int q;
    bool rmq;
    bool sums;   // a's side from the lineage-sum table (needs rmq and 4 more bytes of scratch per pair)
};

static SortedShape sorted_shape(const st_tree *t)
{
    const size_t image = ladder_image_bytes(t->canopy_nodes);
    static const int forced = std::getenv("SUCHTREE_AMD_SORT_Q") ? std::atoi(std::getenv("SUCHTREE_AMD_SORT_Q")) : 0;   // tuning experiments
    const bool table = t->d_rmq != nullptr;
    const bool lineage = table && t->d_lineage != nullptr && t->lineage_sums;
    struct Mode { bool rmq, sums; };
    // lineage sums first (they are worth a smaller tile), then the sparse table alone, then
    // the lock-step search; within a mode the largest tile that fits, two workgroups per CU if possible
    for (const Mode m : {Mode{true, true}, Mode{true, false}, Mode{false, false}}) {
        if ((m.rmq && !table) || (m.sums && !lineage)) continue;
        if ((forced == 1 || forced == 2 || forced == 4) && image + sort_scratch_bytes(forced, m.rmq, m.sums) <= 160 * 1024)
            return {forced, m.rmq, m.sums};
        // two workgroups per CU where that is possible -- except with lineage sums: that form of the
        // kernel needs more than 64 VGPRs, so only one workgroup fits a CU anyway, and the larger
        // tile wins (caterpillar of 2048 leaves: 1.35e10 pairs/s with 4096-pair tiles, 9.8e9 with 2048)
        if (!m.sums && image + sort_scratch_bytes(2, m.rmq, m.sums) <= 80 * 1024) return {2, m.rmq, m.sums};
        // (measured on nj.tree, 9111 canopy nodes: lineage sums with 1024-pair tiles 1.37e10 pairs/s,
        // lock-step search with 2048-pair tiles 1.19e10, sparse table alone with 2048-pair tiles 1.03e10)
        for (const int q : {4, 2, 1}) {
            if (q == 1 && !m.sums) continue;
            if (q == 2 && m.rmq && !m.sums) continue;
            if (image + sort_scratch_bytes(q, m.rmq, m.sums) <= 160 * 1024) return {q, m.rmq, m.sums};
        }
    }
    return {0, false, false};
}

static int sorted_q(const st_tree *t) { return sorted_shape(t).q; }

// Smallest batch the canopy kernels take.  The tile-sorted kernel has a fixed cost (every
// workgroup stages a ladder image of up to 150 KiB, sorts, and on the host path its slot is
// staged through device memory), and with lineage sums the walk kernel does 8e9 pairs/s on deep
// trees: 10,000 pairs of ml.tree through the host path 73 us sorted, 40 us walked.
constexpr int64_t kCanopyMinPairs = 4096;
constexpr int64_t kSortedMinPairs = 32768;        // deep canopies with lineage sums: below this the walk kernel wins
constexpr int64_t kSortedMinPairsHost = 131072;   // ... on the host path, where the tile-sorted kernel also needs its slot staged in device memory

static int64_t canopy_min_pairs(const st_tree *t)
{
    return t->tile_sort && t->d_lineage && t->lineage_sums && sorted_q(t) > 0 ? kSortedMinPairs : kCanopyMinPairs;
}

static bool mrca_ranks_ready(const st_tree *t)
{
    return t->strategy == ST_STRATEGY_CANOPY && t->mrca_ranks && t->d_rec_r && t->d_rmq64;
}

static bool wants_device_stage(const st_tree *t, int64_t m)
{
    if (t->strategy != ST_STRATEGY_CANOPY || !t->tile_sort || sorted_q(t) <= 0) return false;
    return m >= (canopy_min_pairs(t) == kSortedMinPairs ? kSortedMinPairsHost : kCanopyMinPairs);
}

template <int CAP, typename Src>
static hipError_t launch_canopy_sorted(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n,
                                       DistSink out_d, int32_t *out_m, Fault *fault, hipStream_t stream)
{
    const SortedShape shape = sorted_shape(t);
    const int q = shape.q;
    const size_t lds = ladder_image_bytes(t->canopy_nodes) + sort_scratch_bytes(q, shape.rmq, shape.sums);
    CanopyParams Pk = P;
    if (!shape.rmq) { Pk.cpos = nullptr; Pk.rmq = nullptr; }
    if (!shape.sums) { Pk.rec_p = nullptr; Pk.lineage = nullptr; }
    const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int64_t tile = (int64_t)q * kCanopyBlock;
    int64_t blocks = (n + tile - 1) / tile;
    blocks = std::max<int64_t>(std::min<int64_t>(blocks, (int64_t)t->n_cu * wg_per_cu), 1);
    int key_shift = 0;     // keys are edge counts: of both canopy lineages, or (lineage sums) of b's whole lineage
    const int key_max = shape.sums ? t->canopy_depth + t->rec_cap : 2 * t->canopy_depth;
    while ((key_max >> key_shift) >= kSortBuckets) key_shift++;
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kCanopyBlock), lds, stream, Pk, src,
                           (long long)n, out_d, out_m, fault, key_shift);
        return hipGetLastError();
    };
    if (shape.sums)
        return q == 1 ? go(k_canopy_sorted<CAP, 1, true, Src>) : q == 2 ? go(k_canopy_sorted<CAP, 2, true, Src>)
                                                                         : go(k_canopy_sorted<CAP, 4, true, Src>);
    return q == 2 ? go(k_canopy_sorted<CAP, 2, false, Src>) : go(k_canopy_sorted<CAP, 4, false, Src>);
}

template <int CAP, typename Src>
static hipError_t launch_canopy_t(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n,
                                  DistSink out_d, int32_t *out_m, Fault *fault, hipStream_t stream)
{
    // tile-sorted kernel: the default of deep canopies, when its scratch fits next to the canopy image
    if (t->tile_sort && sorted_q(t) > 0)
        return launch_canopy_sorted<CAP>(t, P, src, n, out_d, out_m, fault, stream);
    if constexpr (CAP == 0) {
        return launch_canopy_k(k_canopy<0, Src>, 1, t, P, src, n, out_d, out_m, fault, stream);
    } else {
        if (t->pairs_per_lane == 0) return launch_canopy_k(k_canopy<CAP, Src>, 1, t, P, src, n, out_d, out_m, fault, stream);
        // two pairs per lane: a measured-equal variant kept selectable for explicit pair arrays only
        if constexpr (std::is_same<Src, SrcContig>::value || std::is_same<Src, SrcContig32>::value) {
            if (t->pairs_per_lane == 2)
                return launch_canopy_k(k_canopy_ilp<CAP, 2, Src>, 2, t, P, src, n, out_d, out_m, fault, stream);
        }
        return launch_canopy_k(k_canopy_ilp<CAP, 1, Src>, 1, t, P, src, n, out_d, out_m, fault, stream);
    }
}

template <typename Src>
static hipError_t launch_canopy(const st_tree *t, const Src &src, int64_t n, DistSink out_d,
                                int32_t *out_m, Fault *fault, hipStream_t stream)
{
    CanopyParams P;
    P.canopy = t->d_canopy;
    P.canopy_id = t->d_canopy_id;
    P.ladder = t->d_ladder;
    P.cdepth = t->d_cdepth;
    P.cpos = t->d_cpos;
    P.rmq = t->d_rmq;
    P.rec_a = t->d_rec_a;
    P.rec_b = t->d_rec_b;
    P.rec_i = t->d_rec_i;
    P.rec_p = t->d_rec_p;
    P.rmq64 = t->d_rmq64;
    P.rec_r = t->d_rec_r;
    P.lineage = t->d_lineage;
    P.n_nodes = t->n_nodes;
    P.n_leaves = t->n_leaves;
    P.canopy_nodes = t->canopy_nodes;
    P.rec_bytes = t->rec_bytes;
    P.parity = t->parity;
    if (!out_d.any() && out_m && P.rec_r && P.rmq64 && t->mrca_ranks) {
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)t->n_cu * 8));
        hipLaunchKernelGGL(k_mrca_ranks<Src>, dim3((unsigned)blocks), dim3(256), 0, stream, P, src, (long long)n, out_m, fault);
        return hipGetLastError();
    }
    // 31-slot chains are register resident only in the tile-sorted kernel when it runs one
    // workgroup per CU (128 VGPRs per lane); everywhere else they are read through a pointer
    if (t->rec_cap == 31 && t->tile_sort && sorted_q(t) > 0 &&
        ladder_image_bytes(t->canopy_nodes) + sort_scratch_bytes(sorted_q(t), sorted_shape(t).rmq, sorted_shape(t).sums) > 80 * 1024)
        return launch_canopy_sorted<31>(t, P, src, n, out_d, out_m, fault, stream);
    switch (t->rec_cap) {
        case 1: return launch_canopy_t<1>(t, P, src, n, out_d, out_m, fault, stream);
        case 3: return launch_canopy_t<3>(t, P, src, n, out_d, out_m, fault, stream);
        case 7: return launch_canopy_t<7>(t, P, src, n, out_d, out_m, fault, stream);
        case 15: return launch_canopy_t<15>(t, P, src, n, out_d, out_m, fault, stream);
        default: return launch_canopy_t<0>(t, P, src, n, out_d, out_m, fault, stream);
    }
}

static WalkParams walk_params(const st_tree *t)
{
    WalkParams P;
    P.nodes = t->d_nodes;
    P.depth = t->d_depth;
    P.stride = t->d_stride;
    P.rmq = t->d_tree_rmq;
    P.n_nodes = t->n_nodes;
    if (t->d_lineage && t->lineage_sums) {
        P.lineage.rec_p = t->d_rec_p;
        P.lineage.sums = t->d_lineage;
        P.lineage.n_leaves = t->n_leaves;
        P.lineage.parity = t->parity != 0;
    }
    return P;
}

template <typename Src>
static hipError_t launch_walk(const st_tree *t, const Src &src, int64_t n, DistSink out_d,
                              int32_t *out_m, Fault *fault, hipStream_t stream)
{
    const WalkParams P = walk_params(t);
    int64_t blocks = (n + 255) / 256;
    blocks = std::min<int64_t>(blocks, (int64_t)t->n_cu * 16);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(k_walk<Src>, dim3((unsigned)blocks), dim3(256), 0, stream, P, src,
                       (long long)n, out_d, out_m, fault);
    return hipGetLastError();
}

// Small batches are not worth staging 128 KiB of canopy per workgroup.

template <typename Src>
static int enqueue_src(st_tree *t, const Src &src, int64_t n, DistSink d_out, int32_t *d_mrca,
                       Fault *fault, hipStream_t stream, bool allow_sorted = true)
{
    if (n == 0) return ST_OK;
    // MRCA-only requests (d_out == NULL) also go through the canopy kernels: the id comes out
    // of the same climb, and that is ~7x faster than walking the global table
    // (allow_sorted = false: pairs and results are in pinned host memory, which the tile-sorted
    // kernel must not work on -- it reads every pair twice and scatters its stores)
    // MRCA ids only, rank table available: k_mrca_ranks whatever the tree's depth (it reads every
    // pair once and stores coalesced, so it may also work on pinned host memory)
    const bool ranks_only = !d_out.any() && d_mrca && mrca_ranks_ready(t) && n >= kCanopyMinPairs;
    const bool canopy = ranks_only ||
                        (t->strategy == ST_STRATEGY_CANOPY && n >= canopy_min_pairs(t) &&
                         (allow_sorted || !(t->tile_sort && sorted_q(t) > 0)));
    const hipError_t e = canopy ? launch_canopy(t, src, n, d_out, d_mrca, fault, stream)
                                : launch_walk(t, src, n, d_out, d_mrca, fault, stream);
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return ST_OK;
}

static int enqueue(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t s0, int64_t s1,
                   DistSink d_out, int32_t *d_mrca, hipStream_t stream)
{
    const long long *p = reinterpret_cast<const long long *>(d_pairs);
    if (s0 == 2 && s1 == 1 && (reinterpret_cast<uintptr_t>(d_pairs) & 15) == 0)
        return enqueue_src(t, SrcContig{p}, n, d_out, d_mrca, t->d_fault, stream);
    return enqueue_src(t, SrcStrided{p, (long long)s0, (long long)s1}, n, d_out, d_mrca, t->d_fault, stream);
}

// Copy a fault word back (synchronises `stream`) and re-arm it if it had fired.
static int fetch_fault(Fault *d_word, hipStream_t stream, Fault &f)
{
    ST_HIP(hipMemcpyAsync(&f, d_word, sizeof(Fault), hipMemcpyDeviceToHost, stream));
    ST_HIP(hipStreamSynchronize(stream));
    if (f.max_bad == kFaultInit.max_bad && f.min_bad == kFaultInit.min_bad) return ST_OK;
    ST_HIP(hipMemcpyAsync(d_word, &kFaultInit, sizeof(Fault), hipMemcpyHostToDevice, stream));
    ST_HIP(hipStreamSynchronize(stream));
    return ST_OK;
}

// The host path's fault word is clean between calls (fetch_fault re-arms it when it fired), so
// a call does not pay a reset + synchronisation up front -- unless the previous call on this
// tree ended early.  begin_host_faults marks the word as in use, end_host_faults reads it back.
static int begin_host_faults(st_tree *t, hipStream_t stream)
{
    if (t->host_fault_dirty) {
        ST_HIP(hipMemcpyAsync(t->d_fault_host, &kFaultInit, sizeof(Fault), hipMemcpyHostToDevice, stream));
        ST_HIP(hipStreamSynchronize(stream));
    }
    t->host_fault_dirty = true;
    return ST_OK;
}

static int end_host_faults(st_tree *t, hipStream_t stream, Fault &f)
{
    const int rc = fetch_fault(t->d_fault_host, stream, f);
    if (rc == ST_OK) t->host_fault_dirty = false;
    return rc;
}

static void merge_fault(Fault &into, const Fault &f)
{
    into.max_bad = std::max(into.max_bad, f.max_bad);
    into.min_bad = std::min(into.min_bad, f.min_bad);
}

// ST_OK, or ST_ERR_BOUNDS with the id the reference reports: max_id when it is too large,
// else min_id (MuchTree.pyx:897-903)
static int report_fault(int64_t n_nodes, const Fault &f, int64_t *bad_id)
{
    if (f.max_bad == kFaultInit.max_bad && f.min_bad == kFaultInit.min_bad) return ST_OK;
    const long long bad = f.max_bad >= n_nodes ? f.max_bad : f.min_bad;
    if (bad_id) *bad_id = bad;
    return fail(ST_ERR_BOUNDS, "Node ID " + std::to_string(bad) + " out of bounds (tree size: " +
                                   std::to_string(n_nodes) + ")");
}

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
constexpr int kDeepCanopyDepth = 100;     // canopies deeper than this (edges) are "deep"
constexpr int kDeepCanopyNodes = 10240;   // 80 KiB LDS image: two 1024-lane workgroups per CU
constexpr int64_t kMaxLineageEntries = (int64_t)1 << 28;   // 1 GiB of lineage sums at most (ml.tree: 48 MB)
constexpr int64_t kMailboxPairs = 8192;   // largest batch served through the mailbox (beyond it the staged pipe's fixed ~60 us pay off)

// How a host batch of n pairs is cut into pipeline chunks and dealt over n_dev devices:
// chunk c covers [c*chunk, min(n, (c+1)*chunk)) and belongs to device index c % n_dev.  With
// several devices the chunk shrinks (down to kHostChunkMin) so that every device gets work.
static int64_t host_chunk_pairs(int64_t n, int n_dev)
{
    static const int64_t forced = [] {     // tuning experiments
        const char *env = std::getenv("SUCHTREE_AMD_HOST_CHUNK");
        return env ? std::max<int64_t>(1024, std::atoll(env)) / 1024 * 1024 : (int64_t)0;
    }();
    if (forced) return n <= forced ? std::max<int64_t>(n, 1) : forced;
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
        const size_t bytes = (size_t)kMailboxPairs * (16 + 8 + 4) + 64;     // + the completion word
        ST_HIP(hipHostMalloc(&t->mb_host, bytes, hipHostMallocMapped));
        ST_HIP(hipHostGetDevicePointer(&t->mb_dev, t->mb_host, 0));
        ST_HIP(hipMalloc(reinterpret_cast<void **>(&t->d_fault_mb), sizeof(Fault)));      // [0]: block counter of the mailbox kernel
        ST_HIP(hipStreamCreateWithFlags(&t->mb_stream, hipStreamNonBlocking));
        // cleared ON the mailbox stream: a hipMemset on the null stream is not ordered before
        // kernels of a non-blocking stream, and recycled device memory is not zero
        ST_HIP(hipMemsetAsync(t->d_fault_mb, 0, sizeof(Fault), t->mb_stream));
        *reinterpret_cast<volatile unsigned *>(static_cast<char *>(t->mb_host) + (size_t)kMailboxPairs * 28) = 0;
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
    const WalkParams P = walk_params(t);
    double *d_dist = reinterpret_cast<double *>(d_base + (size_t)kMailboxPairs * 16);
    int32_t *d_mrca = reinterpret_cast<int32_t *>(d_base + (size_t)kMailboxPairs * 24);
    unsigned *d_done = reinterpret_cast<unsigned *>(d_base + (size_t)kMailboxPairs * 28);
    volatile unsigned *h_done = reinterpret_cast<volatile unsigned *>(static_cast<char *>(t->mb_host) + (size_t)kMailboxPairs * 28);
    unsigned seq = ++t->mb_seq;
    if (seq == 0) seq = ++t->mb_seq;     // (0 is the word's initial value)
    hipLaunchKernelGGL(k_walk_mailbox, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, t->mb_stream, P,
                       reinterpret_cast<const long long *>(d_base), (int)n, out_dist ? d_dist : nullptr,
                       out_mrca ? d_mrca : nullptr, reinterpret_cast<unsigned *>(t->d_fault_mb), d_done, seq);
    ST_HIP(hipGetLastError());
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

// float32 (device) -> float64 (pinned host), coalesced: the staged form of a direct result write
__global__ __launch_bounds__(1024) void k_widen_copy(const float *__restrict__ src, double *__restrict__ dst, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (double)src[i];
}

// Is [p, p + bytes) pinned host memory the GPU can address (hipHostMalloc / hipHostRegister)?
// Result arrays like that -- st_host_alloc blocks, pinned torch tensors -- are written by the
// kernels directly: no staging slot, no unpack pass, no page faults.
static bool device_visible_host(const void *p, int64_t bytes)
{
    if (!p || bytes <= 0) return false;
    for (const char *q : {static_cast<const char *>(p), static_cast<const char *>(p) + bytes - 1}) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, q) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (attr.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// Where the results of a host-path call go: the caller's arrays, and whether the kernels can
// write each of them directly.
struct HostOut {
    double *dist = nullptr;
    int32_t *mrca = nullptr;
    bool direct_d = false, direct_m = false;
};

static HostOut make_host_out(double *out_dist, int32_t *out_mrca, int64_t n)
{
    HostOut o;
    o.dist = out_dist;
    o.mrca = out_mrca;
    o.direct_d = device_visible_host(out_dist, n * 8);
    o.direct_m = device_visible_host(out_mrca, n * 4);
    return o;
}

// The tile-sorted kernel reads every pair twice and stores results in sorted order: fine in
// HBM, ruinous over PCIe (scattered 4-byte writes).  For trees that use it the host path keeps
// the slot in device memory and moves it with the copy kernel above.
static bool wants_device_stage(const st_tree *t, int64_t m);

// One chunk of a host-path call on slot s: `make_src(in)` builds the pair source from the
// chunk's input pointer (NULL for generated sources), results go to the slot's pinned arrays.
template <typename MakeSrc>
static int launch_chunk(st_tree *r, PipeSlot &s, int64_t off, int64_t m, int in_words_per_pair, const HostOut &out,
                        MakeSrc make_src)
{
    if (!wants_device_stage(r, m) || (!out.dist && mrca_ranks_ready(r))) {
        DistSink sink{nullptr, nullptr};
        if (out.dist) {
            if (out.direct_d) sink.d64 = out.dist + off;
            else sink.f32 = static_cast<float *>(s.h_d);
        }
        int32_t *mrca = !out.mrca ? nullptr : out.direct_m ? out.mrca + off : static_cast<int32_t *>(s.h_m);
        return enqueue_src(r, make_src(s.h_in), m, sink, mrca, r->d_fault_host, s.stream, false);
    }
    // Pairs come in through the copy engine, results go out through copy kernels: the two
    // directions then overlap and the engine takes no CUs from the tile-sorted kernel (ml.tree,
    // 2e7 pairs, both outputs: input by copy kernel as well 2.7e9 pairs/s, this way 3.7-4.0e9,
    // both directions by the copy engine 3.3-3.5e9).
    hipError_t e = r->dp->pipe.ensure_device_stage();
    if (e == hipSuccess && in_words_per_pair)
        e = hipMemcpyAsync(s.d_in, s.h_in, (size_t)m * in_words_per_pair * 4, hipMemcpyHostToDevice, s.stream);
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("device staging: ") + hipGetErrorString(e));
    const int rc = enqueue_src(r, make_src(s.d_in), m, DistSink{nullptr, out.dist ? static_cast<float *>(s.d_d) : nullptr},
                               out.mrca ? static_cast<int32_t *>(s.d_m) : nullptr, r->d_fault_host, s.stream);
    if (rc != ST_OK) return rc;
    if (out.dist && out.direct_d) {
        hipLaunchKernelGGL(k_widen_copy, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((m + 1023) / 1024, kCopyKernelBlocks))),
                           dim3(1024), 0, s.stream, static_cast<const float *>(s.d_d), out.dist + off, (long long)m);
        e = hipGetLastError();
    } else if (out.dist) {
        e = enqueue_words_copy(s.d_d, s.h_d, m, s.stream);
    }
    if (e == hipSuccess && out.mrca) e = enqueue_words_copy(s.d_m, out.direct_m ? static_cast<void *>(out.mrca + off) : s.h_m, m, s.stream);
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("device staging: ") + hipGetErrorString(e));
    return ST_OK;
}

// Push this device's chunks of a batch through the slots of the pipe (host_pipe.h).
// pack(slot, off, m) fills slot.h_in for chunk [off, off+m); launch(slot, off, m) enqueues
// the kernel on slot.stream, reading slot.h_in and writing slot.h_d / slot.h_m -- pinned
// host memory, accessed by the kernel over PCIe (see host_pipe.h) -- or, where the caller's
// own result array is pinned (HostOut::direct_*), that array itself.  Caller holds the
// device pipe's mutex.
template <typename Pack, typename Launch>
static int run_pipe(st_tree *t, const ChunkSeq &seq, Pack pack, Launch launch, const HostOut &out, Fault &fault)
{
    fault = kFaultInit;
    // SUCHTREE_AMD_TRACE_PIPE=1: one line per call on stderr with the host thread's time by phase
    static const bool trace = std::getenv("SUCHTREE_AMD_TRACE_PIPE") != nullptr;
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
    double *const out_dist = out.direct_d ? nullptr : out.dist;
    int32_t *const out_mrca = out.direct_m ? nullptr : out.mrca;
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
        const hipError_t e = hipEventSynchronize(s.done);
        if (e != hipSuccess) return e;
        lap(t_wait, tp);
        // distances crossed PCIe as float32 and are widened into the caller's float64 array;
        // MRCA ids are copied; one pass of the pool over the chunk does both
        const float *src_d = static_cast<const float *>(s.h_d);
        const int32_t *src_m = static_cast<const int32_t *>(s.h_m);
        double *dst_d = out_dist ? out_dist + s.off : nullptr;
        int32_t *dst_m = out_mrca ? out_mrca + s.off : nullptr;
        if (dst_d || dst_m)
            P.pool.parallel_for(s.m, [=](int64_t b, int64_t e) {
                if (dst_d) widen_f32_to_f64(dst_d + b, src_d + b, e - b);
                if (dst_m) copy_stream(dst_m + b, src_m + b, (e - b) * 4);
            });
        lap(t_unpack, tp);
        return hipSuccess;
    };
    // pages of a freshly allocated result array are populated here, by the pool, while the
    // chunk is on the GPU -- not one fault at a time inside the unpack loops
    // (only pages that are not there yet: a recycled result array is resident already, and
    // populating resident pages costs more than everything else a mid-sized call does)
    auto prefault = [&](int64_t off, int64_t m) {
        double *const pd = out_dist && !looks_resident(out_dist + off, m * 8) ? out_dist : nullptr;
        int32_t *const pm = out_mrca && !looks_resident(out_mrca + off, m * 4) ? out_mrca : nullptr;
        if (!pd && !pm) return;
        P.pool.parallel_for(m, [=](int64_t b, int64_t e) {
            if (pd) populate_for_write(pd + off + b, (e - b) * 8);
            if (pm) populate_for_write(pm + off + b, (e - b) * 4);
        });
    };
    auto bail = [&](int code, const std::string &msg) {
        for (auto &s : P.slot) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            s.busy = false;
        }
        return fail(code, msg);
    };
    int64_t k = 0;
    for (int64_t c = seq.first; c * seq.chunk < seq.n; c += seq.step, k++) {
        const int64_t off = c * seq.chunk;
        const int64_t m = std::min(seq.chunk, seq.n - off);
        PipeSlot &s = P.slot[k % kPipeSlots];
        hipError_t e = drain(s);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
        Clock::time_point tp = trace ? Clock::now() : Clock::time_point();
        pack(s, off, m);
        lap(t_pack, tp);
        const int rc = launch(s, off, m);
        if (rc != ST_OK) return bail(rc, g_last_error);
        if ((c + seq.step) * seq.chunk >= seq.n) {
            // last chunk of this device: fetch the fault word behind it (and behind the chunk
            // still in flight on the other stream), so that one wait covers results and faults
            for (PipeSlot &other : P.slot)
                if (&other != &s && other.busy && e == hipSuccess) e = hipStreamWaitEvent(s.stream, other.done, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(P.h_fault, t->d_fault_host, sizeof(Fault), hipMemcpyDeviceToHost, s.stream);
            if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
        }
        e = hipEventRecord(s.done, s.stream);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
        s.busy = true;
        s.off = off;
        s.m = m;
        lap(t_launch, tp);
        prefault(off, m);
        lap(t_prefault, tp);
        e = drain(P.slot[(k + 1) % kPipeSlots]);   // unpack the oldest chunk while the newer ones are in flight
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
    }
    for (int j = 0; j < kPipeSlots; j++) {     // oldest first
        const hipError_t e = drain(P.slot[(k + j) % kPipeSlots]);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("pipeline: ") + hipGetErrorString(e));
    }
    if (k > 0) {
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
                     (long long)seq.n, (long long)seq.chunk, (long long)k,
                     std::chrono::duration<double, std::micro>(Clock::now() - t_begin).count(), t_pack, t_launch, t_prefault,
                     t_wait, t_unpack);
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

// Tables are built once on the host, then uploaded to every device of the handle.
struct BuiltTables {
    TreeTables T;
    bool canopy_ok = false;
    bool deep = false;
};

static int build_tables(const int32_t *parent, const float *distance, int64_t n_nodes, int strategy, BuiltTables &B)
{
    std::string err;
    if (!prepare_basic(parent, distance, n_nodes, B.T, err)) return fail(ST_ERR_TREE, err);
    int max_canopy = 0;
    if (const char *env = std::getenv("SUCHTREE_AMD_CANOPY_NODES")) max_canopy = std::atoi(env);   // tuning experiments
    if (strategy != ST_STRATEGY_WALK) {
        B.canopy_ok = prepare_canopy(parent, distance, B.T, max_canopy);
        // Deep canopies (real, unbalanced phylogenies: hundreds of levels) spend their time in
        // the LDS climb, not in memory.  There a canopy image small enough for two workgroups
        // per CU, a longer understory (more of each lineage pre-summed in its record) and the
        // branchy scalar kernel (finished lanes stop issuing LDS reads) measured 13-30 % faster.
        if (B.canopy_ok && max_canopy == 0) {
            int cdepth = 0;
            for (const CanopyEntry &e : B.T.canopy) cdepth = std::max<int>(cdepth, (int)(e.link >> 16));
            if (cdepth > kDeepCanopyDepth) {
                B.deep = true;
                int deep_nodes = kDeepCanopyNodes;
                if (const char *env = std::getenv("SUCHTREE_AMD_DEEP_NODES")) deep_nodes = std::atoi(env);   // tuning experiments
                if (B.T.canopy_nodes > deep_nodes) {
                    TreeTables T2 = B.T;
                    if (prepare_canopy(parent, distance, T2, deep_nodes)) B.T = std::move(T2);
                }
                // a's side of every pair from one read (tree_prep.h: lineage sums); 4 bytes per
                // node and level, so only while the table stays below kMaxLineageEntries
                (void)prepare_lineage_sums(B.T, kMaxLineageEntries);
            }
        }
    }
    if (B.canopy_ok) (void)prepare_rank_table(B.T);      // MRCA-only queries of in-order trees
    if (strategy == ST_STRATEGY_CANOPY && !B.canopy_ok)
        return fail(ST_ERR_TREE, "tree does not admit the canopy family (understory deeper than a record)");
    return ST_OK;
}

static int upload_tree(BuiltTables &B, int device, st_tree **out)
{
    TreeTables &T = B.T;
    int n_dev = 0;
    ST_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev)
        return fail(ST_ERR_HIP, "device " + std::to_string(device) + " not available (" +
                                    std::to_string(n_dev) + " visible)");
    ST_DEVICE(device);
    hipDeviceProp_t prop;
    ST_HIP(hipGetDeviceProperties(&prop, device));

    st_tree *t = new (std::nothrow) st_tree();
    if (!t) return fail(ST_ERR_NOMEM, "out of host memory");
    t->device = device;
    t->dp = pipe_acquire(device);
    t->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    t->n_nodes = T.n;
    t->n_leaves = T.n_leaves;
    if (B.deep) { t->pairs_per_lane = 0; t->tile_sort = 1; }
    int64_t bytes = 0;
    int rc = upload(&t->d_nodes, T.nodes, &bytes);
    if (rc == ST_OK) rc = upload(&t->d_depth, T.depth, &bytes);
    if (rc == ST_OK) rc = upload(&t->d_stride, T.stride, &bytes);
    if (rc == ST_OK && !T.tree_rmq.empty()) rc = upload(&t->d_tree_rmq, T.tree_rmq, &bytes);
    if (rc == ST_OK && B.canopy_ok) {
        t->has_canopy = true;
        t->canopy_nodes = T.canopy_nodes;
        t->rec_bytes = T.record_bytes;
        t->rec_cap = T.record_cap;
        t->parity = T.parity_layout ? 1 : 0;
        for (const CanopyEntry &e : T.canopy) t->canopy_depth = std::max<int>(t->canopy_depth, (int)(e.link >> 16));
        std::vector<CanopyEntry> image = T.canopy;
        if (image.size() & 1) image.push_back(CanopyEntry{0.0f, 0u});   // 16-byte staging granule
        rc = upload(&t->d_canopy, image, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_canopy_id, T.canopy_id, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_ladder, T.ladder, &bytes);
        if (rc == ST_OK) {
            std::vector<uint16_t> cd = T.canopy_depth;
            cd.resize((cd.size() + 7) / 8 * 8, 0);     // 16-byte staging granule
            rc = upload(&t->d_cdepth, cd, &bytes);
        }
        if (rc == ST_OK && B.deep && T.inorder_ids && !T.canopy_rmq.empty()) {
            rc = upload(&t->d_cpos, T.canopy_pos, &bytes);
            if (rc == ST_OK) rc = upload(&t->d_rmq, T.canopy_rmq, &bytes);
        }
        if (rc == ST_OK) rc = upload(&t->d_rec_a, T.rec_a, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_rec_b, T.rec_b, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_rec_i, T.rec_i, &bytes);
        if (rc == ST_OK && !T.rec_r.empty()) {
            rc = upload(&t->d_rec_r, T.rec_r, &bytes);
            if (rc == ST_OK) rc = upload(&t->d_rmq64, T.canopy_rmq64, &bytes);
        }
        if (rc == ST_OK && t->d_rmq && t->d_rmq64 && !T.lineage_sum.empty()) {
            rc = upload(&t->d_rec_p, T.rec_p, &bytes);
            if (rc == ST_OK) rc = upload(&t->d_lineage, T.lineage_sum, &bytes);
        }
    }
    if (rc == ST_OK) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_fault), 2 * sizeof(Fault));
        const Fault init2[2] = {kFaultInit, kFaultInit};
        if (e == hipSuccess) e = hipMemcpy(t->d_fault, init2, sizeof(init2), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(ST_ERR_HIP, std::string("tree setup: ") + hipGetErrorString(e));
        else t->d_fault_host = t->d_fault + 1;
    }
    if (rc != ST_OK) {
        std::string keep = g_last_error;
        st_tree_destroy(t);
        g_last_error = keep;
        return rc;
    }
    t->strategy = B.canopy_ok ? ST_STRATEGY_CANOPY : ST_STRATEGY_WALK;
    t->info.n_nodes = T.n;
    t->info.n_leaves = T.n_leaves;
    t->info.root = T.root;
    t->info.depth = T.tree_depth;
    t->info.device = device;
    t->info.canopy_nodes = B.canopy_ok ? T.canopy_nodes : 0;
    t->info.understory_max = B.canopy_ok ? T.understory_max : 0;
    t->info.record_bytes = B.canopy_ok ? T.record_bytes : 0;
    t->info.n_devices = 1;
    t->info.device_bytes = bytes;
    t->info.lineage_entries = t->d_lineage ? (int64_t)T.lineage_sum.size() : 0;
    *out = t;
    return ST_OK;
}

extern "C" {

const char *st_last_error(void) { return g_last_error.c_str(); }

int st_device_count(int *count)
{
    if (!count) return fail(ST_ERR_ARG, "count is NULL");
    *count = 0;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ST_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return ST_OK;
}

int st_host_depths(const int32_t *parent, int64_t n_nodes, int32_t *out_depths, int32_t *out_tree_depth)
{
    if (!parent || n_nodes <= 0) return fail(ST_ERR_ARG, "parent is NULL or n_nodes <= 0");
    std::vector<float> zeros((size_t)n_nodes, 0.0f);
    TreeTables T;
    std::string err;
    if (!prepare_basic(parent, zeros.data(), n_nodes, T, err)) return fail(ST_ERR_TREE, err);
    if (out_depths) std::memcpy(out_depths, T.depth.data(), (size_t)n_nodes * 4);
    if (out_tree_depth) *out_tree_depth = T.tree_depth;
    return ST_OK;
}

int st_host_chunk_plan(int64_t n, int n_devices, int64_t *chunk_pairs, int64_t *n_chunks)
{
    if (n < 0 || n_devices < 1) return fail(ST_ERR_ARG, "n < 0 or n_devices < 1");
    const int64_t chunk = host_chunk_pairs(n, n_devices);
    if (chunk_pairs) *chunk_pairs = chunk;
    if (n_chunks) *n_chunks = (n + chunk - 1) / chunk;
    return ST_OK;
}

int st_host_chunk_owner(int64_t n, int n_devices, int64_t chunk_index, int *device_index,
                        int64_t *first_pair, int64_t *n_pairs)
{
    if (n < 0 || n_devices < 1 || chunk_index < 0) return fail(ST_ERR_ARG, "bad arguments");
    const int64_t chunk = host_chunk_pairs(n, n_devices);
    if (chunk_index * chunk >= n) return fail(ST_ERR_ARG, "chunk index past the end of the batch");
    // the same arithmetic as ChunkSeq / run_pipe
    if (device_index) *device_index = (int)(chunk_index % n_devices);
    if (first_pair) *first_pair = chunk_index * chunk;
    if (n_pairs) *n_pairs = std::min(chunk, n - chunk_index * chunk);
    return ST_OK;
}

static int check_create_args(const int32_t *parent, const float *distance, int64_t n_nodes, int strategy, st_tree **out)
{
    if (!out) return fail(ST_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!parent || !distance || n_nodes <= 0) return fail(ST_ERR_ARG, "parent/distance NULL or n_nodes <= 0");
    if (strategy != ST_STRATEGY_AUTO && strategy != ST_STRATEGY_WALK && strategy != ST_STRATEGY_CANOPY)
        return fail(ST_ERR_ARG, "unknown strategy " + std::to_string(strategy));
    return ST_OK;
}

int st_tree_create(const int32_t *parent, const float *distance, int64_t n_nodes, int device,
                   int strategy, st_tree **out)
{
    int rc = check_create_args(parent, distance, n_nodes, strategy, out);
    if (rc != ST_OK) return rc;
    BuiltTables B;
    rc = build_tables(parent, distance, n_nodes, strategy, B);
    if (rc != ST_OK) return rc;
    return upload_tree(B, device, out);
}

int st_tree_create_multi(const int32_t *parent, const float *distance, int64_t n_nodes,
                         const int *devices, int n_devices, int strategy, st_tree **out)
{
    int rc = check_create_args(parent, distance, n_nodes, strategy, out);
    if (rc != ST_OK) return rc;
    if (!devices || n_devices < 1) return fail(ST_ERR_ARG, "devices is NULL or n_devices < 1");
    // (SUCHTREE_AMD_ALLOW_DUPLICATE_DEVICES=1: testing aid -- several replicas on one GPU, so that
    // the dealing of chunks over replicas can run on a single-GPU box; they take turns on its pipe)
    const char *dup = std::getenv("SUCHTREE_AMD_ALLOW_DUPLICATE_DEVICES");
    if (!dup || dup[0] != '1')
        for (int i = 0; i < n_devices; i++)
            for (int j = 0; j < i; j++)
                if (devices[i] == devices[j]) return fail(ST_ERR_ARG, "device " + std::to_string(devices[i]) + " listed twice");
    BuiltTables B;
    rc = build_tables(parent, distance, n_nodes, strategy, B);
    if (rc != ST_OK) return rc;
    st_tree *primary = nullptr;
    rc = upload_tree(B, devices[0], &primary);
    if (rc != ST_OK) return rc;
    for (int i = 1; i < n_devices; i++) {
        st_tree *peer = nullptr;
        rc = upload_tree(B, devices[i], &peer);
        if (rc != ST_OK) {
            std::string keep = g_last_error;
            st_tree_destroy(primary);
            g_last_error = keep;
            return rc;
        }
        primary->peers.push_back(peer);
    }
    primary->info.n_devices = n_devices;
    *out = primary;
    return ST_OK;
}

void st_tree_destroy(st_tree *t)
{
    if (!t) return;
    for (st_tree *p : t->peers) st_tree_destroy(p);
    t->peers.clear();
    {
        DeviceScope scope(t->device);
        (void)hipFree(t->d_nodes);
        (void)hipFree(t->d_depth);
        (void)hipFree(t->d_stride);
        (void)hipFree(t->d_tree_rmq);
        (void)hipFree(t->d_canopy);
        (void)hipFree(t->d_canopy_id);
        (void)hipFree(t->d_ladder);
        (void)hipFree(t->d_cdepth);
        (void)hipFree(t->d_cpos);
        (void)hipFree(t->d_rmq);
        (void)hipFree(t->d_rec_a);
        (void)hipFree(t->d_rec_b);
        (void)hipFree(t->d_rec_i);
        (void)hipFree(t->d_rec_p);
        (void)hipFree(t->d_rmq64);
        (void)hipFree(t->d_rec_r);
        (void)hipFree(t->d_lineage);
        (void)hipFree(t->d_fault);
        (void)hipFree(t->q_tmp);
        if (t->mb_host) (void)hipHostFree(t->mb_host);
        (void)hipFree(t->d_fault_mb);
        if (t->mb_stream) (void)hipStreamDestroy(t->mb_stream);
    }
    if (t->dp) pipe_release(t->device);
    delete t;
}

int st_tree_info_get(const st_tree *t, st_tree_info *info)
{
    if (!t || !info) return fail(ST_ERR_ARG, "tree or info is NULL");
    *info = t->info;
    info->strategy = t->strategy;
    return ST_OK;
}

int st_tree_devices(const st_tree *t, int *devices, int capacity, int *n_devices)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    const int n = 1 + (int)t->peers.size();
    if (n_devices) *n_devices = n;
    if (devices)
        for (int i = 0; i < n && i < capacity; i++) devices[i] = i == 0 ? t->device : t->peers[(size_t)i - 1]->device;
    return ST_OK;
}

int st_tree_set_strategy(st_tree *t, int strategy)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (strategy == ST_STRATEGY_AUTO) strategy = t->has_canopy ? ST_STRATEGY_CANOPY : ST_STRATEGY_WALK;
    if (strategy == ST_STRATEGY_CANOPY && !t->has_canopy)
        return fail(ST_ERR_ARG, "tree was built without canopy tables");
    if (strategy != ST_STRATEGY_CANOPY && strategy != ST_STRATEGY_WALK)
        return fail(ST_ERR_ARG, "unknown strategy " + std::to_string(strategy));
    t->strategy = strategy;
    for (st_tree *p : t->peers) p->strategy = strategy;
    return ST_OK;
}

static int set_option_one(st_tree *t, const char *name, int64_t value)
{
    if (std::strcmp(name, "pairs_per_lane") == 0) {
        if (value != 0 && value != 1 && value != 2)
            return fail(ST_ERR_ARG, "pairs_per_lane must be 0, 1 or 2");
        t->pairs_per_lane = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "tile_sort") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "tile_sort must be 0 or 1");
        t->tile_sort = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "mrca_ranks") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "mrca_ranks must be 0 or 1");
        t->mrca_ranks = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "lineage_sums") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "lineage_sums must be 0 or 1");
        t->lineage_sums = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "small_batch_path") == 0) {
        t->small_batch_path = value != 0;
        return ST_OK;
    }
    return fail(ST_ERR_ARG, std::string("unknown option ") + name);
}

int st_tree_set_option(st_tree *t, const char *name, int64_t value)
{
    if (!t || !name) return fail(ST_ERR_ARG, "tree or name is NULL");
    int rc = set_option_one(t, name, value);
    for (st_tree *p : t->peers)
        if (rc == ST_OK) rc = set_option_one(p, name, value);
    return rc;
}

int st_distances_device(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t stride0,
                        int64_t stride1, double *d_out_dist, int32_t *d_out_mrca, void *stream)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !d_pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!d_out_dist && !d_out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    ST_DEVICE(t->device);
    return enqueue(t, d_pairs, n, stride0, stride1, DistSink{d_out_dist, nullptr}, d_out_mrca,
                   reinterpret_cast<hipStream_t>(stream));
}

int st_distances_device_f32(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t stride0,
                            int64_t stride1, float *d_out_dist, int32_t *d_out_mrca, void *stream)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !d_pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!d_out_dist && !d_out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    ST_DEVICE(t->device);
    return enqueue(t, d_pairs, n, stride0, stride1, DistSink{nullptr, d_out_dist}, d_out_mrca,
                   reinterpret_cast<hipStream_t>(stream));
}

int st_fault_check(st_tree *t, void *stream, int64_t *bad_id)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    ST_DEVICE(t->device);
    Fault f;
    const int rc = fetch_fault(t->d_fault, reinterpret_cast<hipStream_t>(stream), f);
    if (rc != ST_OK) return rc;
    return report_fault(t->n_nodes, f, bad_id);
}

}  // extern "C"

// Host-buffer entry point for int64 ids (the reference's dtype) and int32 ids.
template <typename Id>
static int distances_host_impl(st_tree *t, const Id *pairs, int64_t n, int64_t stride0, int64_t stride1,
                               double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!out_dist && !out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    if (n == 0) return ST_OK;
    // (deep canopies: beyond the walk/canopy switch the tile-sorted kernel beats the mailbox's walk)
    // (deep canopies without lineage sums: the walk is slow there, the tile-sorted kernel takes over at 4096 pairs)
    if (n <= (t->tile_sort && !(t->d_lineage && t->lineage_sums) ? kCanopyMinPairs - 1 : kMailboxPairs) && t->small_batch_path) {
        ST_DEVICE(t->device);
        return small_batch(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);
    }
    const HostOut out = make_host_out(out_dist, out_mrca, n);
    // fresh result arrays: ask for huge pages before the first touch (a no-op on resident memory)
    if (out_dist && !out.direct_d && !looks_resident(out_dist, n * 8)) advise_huge(out_dist, n * 8);
    if (out_mrca && !out.direct_m && !looks_resident(out_mrca, n * 4)) advise_huge(out_mrca, n * 4);

    // Ids cross PCIe as int32 (half the H2D bytes).  Values that do not fit are clamped to
    // INT32_MAX / INT32_MIN -- still out of range for the kernel -- and their exact extremes
    // are kept here so that the reported id is the one the reference would report.
    auto work = [&](st_tree *r, const ChunkSeq &seq, Fault &fault) -> int {
        CopyPool &pool = r->dp->pipe.pool;
        hipStream_t s0 = nullptr;
        std::mutex wide_mutex;
        Fault wide = kFaultInit;
        auto pack = [&](PipeSlot &s, int64_t off, int64_t m) {
            const Id *src = pairs + off * stride0;
            int32_t *dst = static_cast<int32_t *>(s.h_in);
            const bool c_order = stride0 == 2 && stride1 == 1;
            if (sizeof(Id) == 4 && c_order) {   // int32 C-order: already the wire format
                pool.parallel_for(m, [=](int64_t b, int64_t e) { copy_stream(dst + 2 * b, src + 2 * b, (e - b) * 8); });
                return;
            }
            pool.parallel_for(m, [&, src, dst](int64_t b, int64_t e) {
                long long hi = kFaultInit.max_bad, lo = kFaultInit.min_bad;
                if (sizeof(Id) == 8 && c_order) {
                    narrow_pairs_i64(dst + 2 * b, reinterpret_cast<const int64_t *>(src) + 2 * b, e - b, hi, lo);
                } else {
                    for (int64_t k = b; k < e; k++) {
                        for (int c = 0; c < 2; c++) {
                            const long long v = src[k * stride0 + c * stride1];
                            int32_t w = (int32_t)v;
                            if (v > INT32_MAX) { w = INT32_MAX; hi = std::max(hi, v); }
                            else if (v < INT32_MIN) { w = INT32_MIN; lo = std::min(lo, v); }
                            dst[2 * k + c] = w;
                        }
                    }
                }
                if (hi != kFaultInit.max_bad || lo != kFaultInit.min_bad) {
                    std::lock_guard<std::mutex> g(wide_mutex);
                    wide.max_bad = std::max(wide.max_bad, hi);
                    wide.min_bad = std::min(wide.min_bad, lo);
                }
            });
        };
        auto launch = [&](PipeSlot &s, int64_t off, int64_t m) {
            return launch_chunk(r, s, off, m, 2, out,
                                [](const void *in) { return SrcContig32{static_cast<const int *>(in)}; });
        };
        {   // (the pipe may not exist yet: ensure() inside run_pipe creates the streams)
            const hipError_t e = r->dp->pipe.ensure(std::max<int64_t>(seq.chunk, 1024));
            if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
            s0 = r->dp->pipe.slot[0].stream;
        }
        int rc = begin_host_faults(r, s0);
        if (rc == ST_OK) rc = run_pipe(r, seq, pack, launch, out, fault);
        if (rc != ST_OK) return rc;
        // a clamped id always trips the device check as well; its exact value replaces the clamp
        if (fault.max_bad != kFaultInit.max_bad || fault.min_bad != kFaultInit.min_bad) merge_fault(fault, wide);
        return ST_OK;
    };
    Fault fault;
    const int rc = for_each_replica(t, n, fault, work);
    if (rc != ST_OK) return rc;
    return report_fault(t->n_nodes, fault, bad_id);
}

extern "C" {

int st_distances_host(st_tree *t, const int64_t *pairs, int64_t n, int64_t stride0, int64_t stride1,
                      double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    return distances_host_impl(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);
}

int st_distances_host_i32(st_tree *t, const int32_t *pairs, int64_t n, int64_t stride0, int64_t stride1,
                          double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    return distances_host_impl(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);
}

static int triangle_args(st_tree *t, const int64_t *ids, int64_t m, int64_t k_begin, int64_t k_count,
                         const void *out_d, const void *out_m)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (m < 0 || k_begin < 0 || k_count < 0) return fail(ST_ERR_ARG, "negative size");
    if (m > 3000000000LL) return fail(ST_ERR_ARG, "m too large");
    const int64_t total = m * (m - 1) / 2;
    if (k_begin + k_count > total) return fail(ST_ERR_ARG, "pair range exceeds m(m-1)/2");
    if (k_count > 0 && !ids) return fail(ST_ERR_ARG, "ids is NULL");
    if (!out_d && !out_m) return fail(ST_ERR_ARG, "both outputs are NULL");
    return ST_OK;
}

int st_triangle_device(st_tree *t, const int64_t *d_ids, int64_t m, int64_t id_stride,
                       int64_t k_begin, int64_t k_count, double *d_out_dist, int32_t *d_out_mrca,
                       void *stream)
{
    int rc = triangle_args(t, d_ids, m, k_begin, k_count, d_out_dist, d_out_mrca);
    if (rc != ST_OK) return rc;
    ST_DEVICE(t->device);
    const SrcTriangle src{reinterpret_cast<const long long *>(d_ids), (long long)id_stride, (long long)k_begin};
    return enqueue_src(t, src, k_count, DistSink{d_out_dist, nullptr}, d_out_mrca, t->d_fault,
                       reinterpret_cast<hipStream_t>(stream));
}

int st_triangle_host(st_tree *t, const int64_t *ids, int64_t m, int64_t id_stride, int64_t k_begin,
                     int64_t k_count, double *out_dist, int32_t *out_mrca, int64_t *bad_id)
{
    int rc = triangle_args(t, ids, m, k_begin, k_count, out_dist, out_mrca);
    if (rc != ST_OK) return rc;
    if (k_count == 0) return ST_OK;
    const HostOut out = make_host_out(out_dist, out_mrca, k_count);
    if (out_dist && !out.direct_d && !looks_resident(out_dist, k_count * 8)) advise_huge(out_dist, k_count * 8);
    if (out_mrca && !out.direct_m && !looks_resident(out_mrca, k_count * 4)) advise_huge(out_mrca, k_count * 4);
    // the id list goes up once per device (packed); results stream back through the pipe
    std::vector<int64_t> packed;
    const int64_t *src_ids = ids;
    if (id_stride != 1) {
        packed.resize((size_t)m);
        for (int64_t i = 0; i < m; i++) packed[(size_t)i] = ids[i * id_stride];
        src_ids = packed.data();
    }
    auto work = [&](st_tree *r, const ChunkSeq &seq, Fault &fault) -> int {
        HostPipe &P = r->dp->pipe;
        hipError_t e = P.ensure(std::max<int64_t>(seq.chunk, 1024));
        if (e == hipSuccess) e = P.ensure_ids(m);
        if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
        hipStream_t s0 = P.slot[0].stream;
        if (begin_host_faults(r, s0) != ST_OK) return ST_ERR_HIP;
        ST_HIP(hipMemcpy(P.d_ids, src_ids, (size_t)m * 8, hipMemcpyHostToDevice));
        auto pack = [](PipeSlot &, int64_t, int64_t) {};
        auto launch = [&](PipeSlot &s, int64_t off, int64_t c) {
            const SrcTriangle src{static_cast<const long long *>(P.d_ids), 1, (long long)(k_begin + off)};
            return launch_chunk(r, s, off, c, 0, out, [&](const void *) { return src; });
        };
        return run_pipe(r, seq, pack, launch, out, fault);
    };
    Fault fault;
    rc = for_each_replica(t, k_count, fault, work);
    if (rc != ST_OK) return rc;
    return report_fault(t->n_nodes, fault, bad_id);
}

static int grid_args(st_tree *t, const int64_t *rows, int64_t n_rows, const int64_t *cols, int64_t n_cols,
                     int64_t e_begin, int64_t e_count, const void *out_d, const void *out_m)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n_rows < 0 || n_cols < 0 || e_begin < 0 || e_count < 0) return fail(ST_ERR_ARG, "negative size");
    if (n_rows > 3000000000LL || n_cols > 3000000000LL) return fail(ST_ERR_ARG, "grid too large");
    if (e_begin + e_count > n_rows * n_cols) return fail(ST_ERR_ARG, "element range exceeds n_rows * n_cols");
    if (e_count > 0 && (!rows || !cols)) return fail(ST_ERR_ARG, "id list is NULL");
    if (!out_d && !out_m) return fail(ST_ERR_ARG, "both outputs are NULL");
    return ST_OK;
}

int st_grid_host(st_tree *t, const int64_t *row_ids, int64_t n_rows, const int64_t *col_ids, int64_t n_cols,
                 int symmetric, int64_t e_begin, int64_t e_count, double *out_dist, int32_t *out_mrca,
                 int64_t *bad_id)
{
    int rc = grid_args(t, row_ids, n_rows, col_ids, n_cols, e_begin, e_count, out_dist, out_mrca);
    if (rc != ST_OK) return rc;
    if (symmetric && (n_rows != n_cols)) return fail(ST_ERR_ARG, "symmetric grid needs n_rows == n_cols");
    if (e_count == 0) return ST_OK;
    const HostOut out = make_host_out(out_dist, out_mrca, e_count);
    if (out_dist && !out.direct_d && !looks_resident(out_dist, e_count * 8)) advise_huge(out_dist, e_count * 8);
    if (out_mrca && !out.direct_m && !looks_resident(out_mrca, e_count * 4)) advise_huge(out_mrca, e_count * 4);
    auto work = [&](st_tree *r, const ChunkSeq &seq, Fault &fault) -> int {
        HostPipe &P = r->dp->pipe;
        hipError_t e = P.ensure(std::max<int64_t>(seq.chunk, 1024));
        if (e == hipSuccess) e = P.ensure_ids(n_rows + n_cols);
        if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
        hipStream_t s0 = P.slot[0].stream;
        if (begin_host_faults(r, s0) != ST_OK) return ST_ERR_HIP;
        long long *d_rows = static_cast<long long *>(P.d_ids), *d_cols = d_rows + n_rows;
        ST_HIP(hipMemcpy(d_rows, row_ids, (size_t)n_rows * 8, hipMemcpyHostToDevice));
        ST_HIP(hipMemcpy(d_cols, col_ids, (size_t)n_cols * 8, hipMemcpyHostToDevice));
        auto pack = [](PipeSlot &, int64_t, int64_t) {};
        auto launch = [&](PipeSlot &s, int64_t off, int64_t c) {
            const SrcGrid src{d_rows, d_cols, (long long)n_cols, (long long)(e_begin + off), symmetric};
            return launch_chunk(r, s, off, c, 0, out, [&](const void *) { return src; });
        };
        return run_pipe(r, seq, pack, launch, out, fault);
    };
    Fault fault;
    rc = for_each_replica(t, e_count, fault, work);
    if (rc != ST_OK) return rc;
    return report_fault(t->n_nodes, fault, bad_id);
}

int st_knn_host(st_tree *t, const int64_t *queries, int64_t n_queries, const int64_t *cands, int64_t n_cands,
                int k, int skip_self, int64_t *out_index, double *out_dist, int64_t *bad_id)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n_queries < 0 || n_cands < 0) return fail(ST_ERR_ARG, "negative size");
    if (k < 1 || k > kKnnMaxK) return fail(ST_ERR_ARG, "k must be in [1, " + std::to_string(kKnnMaxK) + "]");
    if (n_cands > 0xFFFFFFFFLL) return fail(ST_ERR_ARG, "too many candidates");
    if (n_queries > 0 && (!queries || !out_index || !out_dist)) return fail(ST_ERR_ARG, "queries or output is NULL");
    if (n_cands > 0 && !cands) return fail(ST_ERR_ARG, "cands is NULL");
    if (n_queries == 0) return ST_OK;
    if (n_cands == 0) {
        for (int64_t i = 0; i < n_queries * k; i++) { out_index[i] = -1; out_dist[i] = std::numeric_limits<double>::quiet_NaN(); }
        return ST_OK;
    }
    ST_DEVICE(t->device);
    std::lock_guard<std::mutex> lock(t->dp->m);
    // rows per launch: the distance block (float32, device only) stays under 256 MiB
    const int64_t rows_per_block = std::max<int64_t>(1, std::min<int64_t>(n_queries, ((int64_t)1 << 26) / n_cands));
    long long *d_q = nullptr, *d_c = nullptr, *d_oi = nullptr;
    float *d_tmp = nullptr;
    double *d_od = nullptr;
    hipStream_t stream = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_q); (void)hipFree(d_c); (void)hipFree(d_oi); (void)hipFree(d_tmp); (void)hipFree(d_od);
        if (stream) (void)hipStreamDestroy(stream);
    };
    hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_q), (size_t)n_queries * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_c), (size_t)n_cands * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_tmp), (size_t)rows_per_block * (size_t)n_cands * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_oi), (size_t)n_queries * k * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_od), (size_t)n_queries * k * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_q, queries, (size_t)n_queries * 8, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, cands, (size_t)n_cands * 8, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { cleanup(); return fail(ST_ERR_HIP, std::string("knn setup: ") + hipGetErrorString(e)); }
    if (begin_host_faults(t, stream) != ST_OK) { cleanup(); return ST_ERR_HIP; }
    for (int64_t r0 = 0; r0 < n_queries; r0 += rows_per_block) {
        const int64_t rows = std::min(rows_per_block, n_queries - r0);
        const SrcGrid src{d_q + r0, d_c, (long long)n_cands, 0, 0};
        const int rc = enqueue_src(t, src, rows * n_cands, DistSink{nullptr, d_tmp}, nullptr, t->d_fault_host, stream);
        if (rc != ST_OK) { (void)hipStreamSynchronize(stream); cleanup(); return rc; }
        hipLaunchKernelGGL(k_knn_select, dim3((unsigned)rows), dim3(256), 0, stream, d_tmp, (long long)n_cands,
                           d_q + r0, d_c, skip_self, k, d_oi + r0 * k, d_od + r0 * k);
        e = hipGetLastError();
        if (e != hipSuccess) { (void)hipStreamSynchronize(stream); cleanup(); return fail(ST_ERR_HIP, std::string("knn launch: ") + hipGetErrorString(e)); }
    }
    e = hipMemcpyAsync(out_index, d_oi, (size_t)n_queries * k * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_dist, d_od, (size_t)n_queries * k * 8, hipMemcpyDeviceToHost, stream);
    Fault f = kFaultInit;
    int rc = e == hipSuccess ? end_host_faults(t, stream, f) : fail(ST_ERR_HIP, std::string("knn D2H: ") + hipGetErrorString(e));
    cleanup();
    if (rc != ST_OK) return rc;
    return report_fault(t->n_nodes, f, bad_id);
}

int st_quartets_host(st_tree *t, const int64_t *quartets, int64_t n, int64_t stride0, int64_t stride1,
                     int64_t *out_topologies, int64_t *bad_id)
{
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && (!quartets || !out_topologies)) return fail(ST_ERR_ARG, "quartets or output is NULL");
    if (n == 0) return ST_OK;
    ST_DEVICE(t->device);
    std::lock_guard<std::mutex> lock(t->dp->m);
    HostPipe &pipe = t->dp->pipe;
    // (n,4) int64 in and out are each the size of two pair rows.  Every slot of the pipe carries
    // one chunk: its pinned buffer holds the chunk's quartets (first half) and, later, its
    // topologies (second half, written by the kernel); while chunk c is on the GPU the host
    // unpacks chunk c-2 and packs chunk c+1.
    const int64_t chunk = std::min<int64_t>(n, kHostChunk / 8);
    {
        hipError_t e = pipe.ensure(std::max<int64_t>(4 * chunk, 1024));
        if (e == hipSuccess) e = pipe.ensure_device_in();
        if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("host staging allocation: ") + hipGetErrorString(e));
    }
    if (begin_host_faults(t, pipe.slot[0].stream) != ST_OK) return ST_ERR_HIP;
    const WalkParams P = walk_params(t);
    const bool canopy = t->strategy == ST_STRATEGY_CANOPY && 6 * chunk >= canopy_min_pairs(t);
    if (canopy && t->q_tmp_cap < kPipeSlots * chunk) {      // six MRCA ids per quartet, per slot
        (void)hipFree(t->q_tmp);
        t->q_tmp = nullptr;
        t->q_tmp_cap = 0;
        ST_HIP(hipMalloc(&t->q_tmp, (size_t)kPipeSlots * (size_t)chunk * 24));
        t->q_tmp_cap = kPipeSlots * chunk;
    }
    auto out_of = [&](PipeSlot &s) { return reinterpret_cast<int64_t *>(static_cast<char *>(s.h_in) + (size_t)chunk * 32); };
    auto drain = [&](PipeSlot &s) -> hipError_t {
        if (!s.busy) return hipSuccess;
        s.busy = false;
        const hipError_t e = hipEventSynchronize(s.done);
        if (e != hipSuccess) return e;
        pipe.pool.copy(out_topologies + s.off * 4, out_of(s), s.m * 32);     // topologies were written straight into pinned memory
        return hipSuccess;
    };
    auto bail = [&](int code, const std::string &msg) {
        for (auto &s : pipe.slot) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            s.busy = false;
        }
        return fail(code, msg);
    };
    int64_t k = 0;
    for (int64_t off = 0; off < n; off += chunk, k++) {
        const int64_t m = std::min(chunk, n - off);
        const int slot_index = (int)(k % kPipeSlots);
        PipeSlot &s = pipe.slot[slot_index];
        hipError_t e = drain(s);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("quartet pipeline: ") + hipGetErrorString(e));
        int64_t *h = static_cast<int64_t *>(s.h_in);
        const int64_t *src = quartets + off * stride0;
        pipe.pool.parallel_for(m, [=](int64_t b, int64_t e2) {
            for (int64_t q = b; q < e2; q++)
                for (int c = 0; c < 4; c++) h[4 * q + c] = src[q * stride0 + c * stride1];
        });
        e = hipMemcpyAsync(s.d_in, s.h_in, (size_t)m * 32, hipMemcpyHostToDevice, s.stream);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("quartet pipeline: ") + hipGetErrorString(e));
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((m + 255) / 256, (int64_t)t->n_cu * 16));
        if (canopy && 6 * m >= canopy_min_pairs(t)) {
            // six MRCA ids per quartet out of the canopy / rank-table kernels, then the pick
            int32_t *tmp = static_cast<int32_t *>(t->q_tmp) + (size_t)slot_index * (size_t)chunk * 6;
            const int rc = enqueue_src(t, SrcQuartet{static_cast<const long long *>(s.d_in)}, 6 * m,
                                       DistSink{nullptr, nullptr}, tmp, t->d_fault_host, s.stream);
            if (rc != ST_OK) return bail(rc, g_last_error);
            hipLaunchKernelGGL(k_quartet_pick, dim3((unsigned)blocks), dim3(256), 0, s.stream,
                               static_cast<const long long *>(s.d_in), static_cast<const int *>(tmp),
                               (long long)m, reinterpret_cast<long long *>(out_of(s)));
        } else {
            hipLaunchKernelGGL(k_quartets, dim3((unsigned)blocks), dim3(256), 0, s.stream, P,
                               static_cast<const long long *>(s.d_in), (long long)m, 4LL, 1LL,
                               reinterpret_cast<long long *>(out_of(s)), t->d_fault_host);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(s.done, s.stream);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("quartet pipeline: ") + hipGetErrorString(e));
        s.busy = true;
        s.off = off;
        s.m = m;
        e = drain(pipe.slot[(k + 1) % kPipeSlots]);      // the oldest chunk, while the newer ones are in flight
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("quartet pipeline: ") + hipGetErrorString(e));
    }
    for (int j = 0; j < kPipeSlots; j++) {
        const hipError_t e = drain(pipe.slot[(k + j) % kPipeSlots]);
        if (e != hipSuccess) return bail(ST_ERR_HIP, std::string("quartet pipeline: ") + hipGetErrorString(e));
    }
    Fault f;
    const int rc = end_host_faults(t, pipe.slot[0].stream, f);
    if (rc != ST_OK) return rc;
    return report_fault(t->n_nodes, f, bad_id);
}

int st_graph_matrices_host(int device, int64_t n, int64_t n_edges, const int32_t *u, const int32_t *v,
                           const double *w, double *out_adjacency, double *out_laplacian)
{
    if (n <= 0 || n_edges < 0) return fail(ST_ERR_ARG, "bad sizes");
    if (n_edges > 0 && (!u || !v || !w)) return fail(ST_ERR_ARG, "edge arrays are NULL");
    if (!out_adjacency && !out_laplacian) return fail(ST_ERR_ARG, "both outputs are NULL");
    for (int64_t e = 0; e < n_edges; e++)
        if (u[e] < 0 || u[e] >= n || v[e] < 0 || v[e] >= n) return fail(ST_ERR_ARG, "edge endpoint out of range");
    ST_DEVICE(device);
    // one workspace: A, L (only if asked for), column sums, edge list.  The dense matrices
    // must fit the GPU's free memory with room to spare; beyond that the caller should not
    // be building a dense Laplacian at all (the reference's numpy version would need the same
    // bytes of host memory).
    const size_t mat = (size_t)n * (size_t)n * 8;
    const size_t edges_b = (size_t)std::max<int64_t>(n_edges, 1);
    const size_t need = mat * (out_laplacian ? 2 : 1) + (size_t)n * 8 + edges_b * 16 + 4096;
    size_t free_b = 0, total_b = 0;
    ST_HIP(hipMemGetInfo(&free_b, &total_b));
    if (need > free_b / 10 * 9)
        return fail(ST_ERR_NOMEM, "dense " + std::to_string(n) + " x " + std::to_string(n) + " graph matrices need " +
                                      std::to_string(need >> 20) + " MiB of device memory, " + std::to_string(free_b >> 20) + " MiB free");
    // one grow-only workspace per device, kept between calls (a 422 x 422 Laplacian is all
    // allocation time otherwise); calls on one device take turns
    struct Workspace { std::mutex m; char *p = nullptr; size_t cap = 0; };
    static std::mutex ws_map_mutex;
    static std::map<int, Workspace *> ws_map;
    Workspace *W;
    {
        std::lock_guard<std::mutex> g(ws_map_mutex);
        Workspace *&slot = ws_map[device];
        if (!slot) slot = new Workspace();
        W = slot;
    }
    std::lock_guard<std::mutex> ws_lock(W->m);
    if (W->cap < need) {
        (void)hipFree(W->p);
        W->p = nullptr;
        W->cap = 0;
        ST_HIP(hipMalloc(reinterpret_cast<void **>(&W->p), need));
        W->cap = need;
    }
    char *ws = W->p;
    double *d_A = reinterpret_cast<double *>(ws);
    double *d_L = out_laplacian ? d_A + (size_t)n * n : nullptr;
    double *d_deg = reinterpret_cast<double *>(ws + mat * (out_laplacian ? 2 : 1));
    double *d_w = d_deg + n;
    int *d_u = reinterpret_cast<int *>(d_w + edges_b), *d_v = d_u + edges_b;
    hipError_t e = hipMemsetAsync(d_A, 0, mat, nullptr);
    if (e == hipSuccess && n_edges) e = hipMemcpyAsync(d_u, u, (size_t)n_edges * 4, hipMemcpyHostToDevice, nullptr);
    if (e == hipSuccess && n_edges) e = hipMemcpyAsync(d_v, v, (size_t)n_edges * 4, hipMemcpyHostToDevice, nullptr);
    if (e == hipSuccess && n_edges) e = hipMemcpyAsync(d_w, w, (size_t)n_edges * 8, hipMemcpyHostToDevice, nullptr);
    if (e == hipSuccess) {
        if (n_edges)
            hipLaunchKernelGGL(k_graph_scatter, dim3((unsigned)std::min<int64_t>((n_edges + 255) / 256, 4096)),
                               dim3(256), 0, nullptr, d_A, (long long)n, (long long)n_edges, d_u, d_v, d_w);
        if (out_laplacian) {
            // one lane per column, rows in increasing order (numpy's sum(axis=0) order); waves of
            // 64 consecutive columns read each row coalesced, one wave per workgroup so that a
            // few hundred columns already spread over the chip
            hipLaunchKernelGGL(k_graph_degree, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, d_A, (long long)n, d_deg);
            hipLaunchKernelGGL(k_graph_laplacian, dim3((unsigned)std::min<int64_t>((n * n + 255) / 256, 65536)),
                               dim3(256), 0, nullptr, d_A, d_deg, (long long)n, d_L);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess && out_adjacency) e = hipMemcpy(out_adjacency, d_A, mat, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_laplacian) e = hipMemcpy(out_laplacian, d_L, mat, hipMemcpyDeviceToHost);
    if (W->cap > ((size_t)256 << 20)) {     // do not sit on a multi-GB workspace
        (void)hipFree(W->p);
        W->p = nullptr;
        W->cap = 0;
    }
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("graph matrices: ") + hipGetErrorString(e));
    return ST_OK;
}

int st_host_alloc(int64_t bytes, void **out)
{
    if (!out || bytes < 0) return fail(ST_ERR_ARG, "bad arguments");
    *out = nullptr;
    // portable: addressable by every GPU of the process (multi-device handles write into it too)
    ST_HIP(hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocPortable));
    return ST_OK;
}

int st_host_free(void *ptr)
{
    if (ptr) ST_HIP(hipHostFree(ptr));
    return ST_OK;
}

int st_device_malloc(int device, int64_t bytes, void **out)
{
    if (!out || bytes < 0) return fail(ST_ERR_ARG, "bad arguments");
    ST_DEVICE(device);
    ST_HIP(hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16)));
    return ST_OK;
}

int st_device_free(int device, void *ptr)
{
    ST_DEVICE(device);
    ST_HIP(hipFree(ptr));
    return ST_OK;
}

int st_memcpy_h2d(int device, void *dst, const void *src, int64_t bytes)
{
    ST_DEVICE(device);
    ST_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return ST_OK;
}

int st_memcpy_d2h(int device, void *dst, const void *src, int64_t bytes)
{
    ST_DEVICE(device);
    ST_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return ST_OK;
}

int st_device_synchronize(int device)
{
    ST_DEVICE(device);
    ST_HIP(hipDeviceSynchronize());
    return ST_OK;
}

}  // extern "C"
