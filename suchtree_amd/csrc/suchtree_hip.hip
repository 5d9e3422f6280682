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
//           k_canopy_ilp     one pair per lane, predicated (shallow canopies: the default)
//           k_canopy_ladder  deep canopies, large batches, long records: records read once, chain in
//                            registers, meeting node from a sparse table, both sides climbed on the
//                            ladder form of the canopy (three edges per 16-byte LDS entry)
//           k_canopy_sorted  deep canopies: pairs sorted by climb length within a workgroup tile; with
//                            in-order ids the meeting node comes from a sparse table and a's side from
//                            per-node lineage sums
//
// Every kernel is templated on a pair source (SrcContig / SrcContig32 / SrcStrided /
// SrcTriangle / SrcGrid / SrcQuartet): an explicit (n,2) array, or pairs derived from their
// index (all-pairs triangle, rows x columns grid, the six pairs of a quartet).
//
// Four translation units (compiled in parallel, build.py): launch_walk.hip (kernels_walk.h), launch_canopy.hip
// (kernels_canopy.h), launch_canopy_sorted.hip (kernels_canopy_sorted.h) -- each a kernel family with its launch
// functions, exported through launch_decl.h -- and this file: error plumbing, the C ABI and the host side in
// between: st_tree.h / host_tree.h (handle, pipe registry), launch_policy.h + host_launch.h (which family a
// request gets, enqueueing, faults), host_path.h (host-buffer pipeline, mailbox, copy kernels), host_upload.h
// (tables -> device), kernels_misc.h (k nearest, graph matrices).
//
// Host side of the C ABI: tree upload to one or several GPUs (tree_prep.cpp builds the tables, under a table budget
// if one is given), the host path (host_pipe.h, host_copy.h: packed ids in through the copy engine, kernels write
// float32 + 24-bit ids straight into pinned host memory), the small-batch mailbox, fault read-back, the creation-time
// choice of a deep tree's kernels (host_tune.h), k-nearest selection, graph matrices.
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

// Nothing may be thrown through the C ABI (the callers are ctypes, cgo, JNI ...): every int-returning export is
// a function-try-block that ends in ST_CATCH_ALL.
static int on_exception() noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        try { g_last_error = "out of host memory"; } catch (...) {}
        return ST_ERR_NOMEM;
    } catch (const std::exception &e) {
        try { g_last_error = std::string("internal error: ") + e.what(); } catch (...) {}
        return ST_ERR_HIP;
    } catch (...) {
        try { g_last_error = "internal error"; } catch (...) {}
        return ST_ERR_HIP;
    }
}
#define ST_CATCH_ALL catch (...) { return st::on_exception(); }

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
#include "kernels_misc.h"


// --------------------------------------------------------------------------
// host side: the tree handle
// --------------------------------------------------------------------------
using namespace st;

#include "host_tree.h"
#include "host_launch.h"
#include "host_path.h"
#include "host_tune.h"
#include "host_upload.h"

extern "C" {

const char *st_last_error(void) { return g_last_error.c_str(); }

int st_device_count(int *count)
try {
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
} ST_CATCH_ALL

int st_host_depths(const int32_t *parent, int64_t n_nodes, int32_t *out_depths, int32_t *out_tree_depth)
try {
    if (!parent || n_nodes <= 0) return fail(ST_ERR_ARG, "parent is NULL or n_nodes <= 0");
    std::vector<float> zeros((size_t)n_nodes, 0.0f);
    TreeTables T;
    std::string err;
    if (!prepare_basic(parent, zeros.data(), n_nodes, T, err)) return fail(ST_ERR_TREE, err);
    if (out_depths) std::memcpy(out_depths, T.depth.data(), (size_t)n_nodes * 4);
    if (out_tree_depth) *out_tree_depth = T.tree_depth;
    return ST_OK;
} ST_CATCH_ALL

int st_link_sample_pairs(uint64_t *state, const int64_t *linklist, int64_t n_links, int64_t count, int64_t *query_a,
                         int64_t *query_b)
try {
    if (!state || !linklist || n_links < 1 || count < 0 || (count > 0 && (!query_a || !query_b)))
        return fail(ST_ERR_ARG, "state / linklist / query arrays are NULL, n_links < 1 or count < 0");
    uint64_t s = *state;
    auto draw = [&]() -> int64_t {      // MuchTree.pyx:2946-2949
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return (int64_t)((s * 2685821657736338717ull) % (uint64_t)n_links);
    };
    for (int64_t k = 0; k < count; k++) {
        const int64_t l1 = draw(), l2 = draw();
        query_a[2 * k] = linklist[2 * l1 + 1];
        query_a[2 * k + 1] = linklist[2 * l2 + 1];
        query_b[2 * k] = linklist[2 * l1];
        query_b[2 * k + 1] = linklist[2 * l2];
    }
    *state = s;
    return ST_OK;
} ST_CATCH_ALL

int st_bucket_moments(const double *dist, int64_t buckets, int64_t n, double *sums, double *sumsq)
try {
    if (buckets < 0 || n < 0 || (buckets > 0 && n > 0 && (!dist || !sums || !sumsq)))
        return fail(ST_ERR_ARG, "dist / sums / sumsq are NULL or a size is negative");
    for (int64_t i = 0; i < buckets; i++)
        for (int64_t j = 0; j < n; j++) {
            const double d = dist[i * n + j];
            sums[i] += d;
            sumsq[i] += std::pow(d, 2.0);
        }
    return ST_OK;
} ST_CATCH_ALL

int st_host_chunk_plan(int64_t n, int n_devices, int64_t *chunk_pairs, int64_t *n_chunks)
try {
    if (n < 0 || n_devices < 1) return fail(ST_ERR_ARG, "n < 0 or n_devices < 1");
    const int64_t chunk = host_chunk_pairs(n, n_devices);
    if (chunk_pairs) *chunk_pairs = chunk;
    if (n_chunks) *n_chunks = (n + chunk - 1) / chunk;
    return ST_OK;
} ST_CATCH_ALL

int st_host_chunk_owner(int64_t n, int n_devices, int64_t chunk_index, int *device_index,
                        int64_t *first_pair, int64_t *n_pairs)
try {
    if (n < 0 || n_devices < 1 || chunk_index < 0) return fail(ST_ERR_ARG, "bad arguments");
    const int64_t chunk = host_chunk_pairs(n, n_devices);
    if (chunk_index * chunk >= n) return fail(ST_ERR_ARG, "chunk index past the end of the batch");
    // the same arithmetic as ChunkSeq / run_pipe
    if (device_index) *device_index = (int)(chunk_index % n_devices);
    if (first_pair) *first_pair = chunk_index * chunk;
    if (n_pairs) *n_pairs = std::min(chunk, n - chunk_index * chunk);
    return ST_OK;
} ST_CATCH_ALL

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
try {
    int rc = check_create_args(parent, distance, n_nodes, strategy, out);
    if (rc != ST_OK) return rc;
    BuiltTables B;
    rc = build_tables(parent, distance, n_nodes, strategy, B);
    if (rc != ST_OK) return rc;
    return upload_tree(B, device, out);
} ST_CATCH_ALL

static int create_multi(const int32_t *parent, const float *distance, int64_t n_nodes, const int *devices, int n_devices,
                        int strategy, int64_t budget_option, st_tree **out);

int st_tree_create_multi(const int32_t *parent, const float *distance, int64_t n_nodes,
                         const int *devices, int n_devices, int strategy, st_tree **out)
try {
    return create_multi(parent, distance, n_nodes, devices, n_devices, strategy, 0, out);
} ST_CATCH_ALL

int st_tree_create_ex(const int32_t *parent, const float *distance, int64_t n_nodes, const int *devices, int n_devices,
                      int strategy, const st_tree_options *opts, st_tree **out)
try {
    if (opts && opts->table_budget_bytes < 0) return fail(ST_ERR_ARG, "table_budget_bytes < 0");
    return create_multi(parent, distance, n_nodes, devices, n_devices, strategy, opts ? opts->table_budget_bytes : 0, out);
} ST_CATCH_ALL

int st_host_table_plan(const int32_t *parent, const float *distance, int64_t n_nodes, int strategy, int64_t table_budget_bytes,
                       int64_t *device_bytes, int32_t *dropped_tables, int32_t *family)
try {
    st_tree *unused = nullptr;
    int rc = check_create_args(parent, distance, n_nodes, strategy, &unused);
    if (rc != ST_OK) return rc;
    if (table_budget_bytes < 0) return fail(ST_ERR_ARG, "table_budget_bytes < 0");
    BuiltTables B;
    rc = build_tables(parent, distance, n_nodes, strategy, B, table_budget_bytes);
    if (rc != ST_OK) return rc;
    if (device_bytes) *device_bytes = device_bytes_of(B);
    if (dropped_tables) *dropped_tables = B.dropped;
    if (family) *family = B.canopy_ok ? ST_STRATEGY_CANOPY : ST_STRATEGY_WALK;
    return ST_OK;
} ST_CATCH_ALL

}  // extern "C"

static int create_multi(const int32_t *parent, const float *distance, int64_t n_nodes, const int *devices, int n_devices,
                        int strategy, int64_t budget_option, st_tree **out)
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
    rc = build_tables(parent, distance, n_nodes, strategy, B, budget_option);
    if (rc != ST_OK) return rc;
    st_tree *primary = nullptr;
    rc = upload_tree(B, devices[0], &primary);
    if (rc != ST_OK) return rc;
    TreeOwner owner{primary};      // (destroys the primary and the peers it holds unless every replica was built)
    primary->peers.reserve((size_t)n_devices);
    for (int i = 1; i < n_devices; i++) {
        st_tree *peer = nullptr;
        rc = upload_tree(B, devices[i], &peer, false);
        if (rc != ST_OK) return rc;
        if (B.deep) copy_tuned_settings(peer, primary);      // one measurement per handle: every replica runs the same kernel
        primary->peers.push_back(peer);
    }
    primary->info.n_devices = n_devices;
    owner.t = nullptr;
    *out = primary;
    return ST_OK;
}

extern "C" {

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
        (void)hipFree(t->d_rec_a4);
        (void)hipFree(t->d_leaf_blocks);
        (void)hipFree(t->d_rec_c);
        (void)hipFree(t->d_rec_b);
        (void)hipFree(t->d_rec_i);
        (void)hipFree(t->d_rec_p);
        (void)hipFree(t->d_rmq64);
        (void)hipFree(t->d_rec_r);
        (void)hipFree(t->d_lineage);
        (void)hipFree(t->d_lineage_len);
        (void)hipFree(t->d_lineage_node_rec);
        (void)hipFree(t->d_crown_rmq);
        (void)hipFree(t->d_crown_ladder);
        (void)hipFree(t->d_fault);
        (void)hipFree(t->d_work);
        for (hipEvent_t ev : t->work_done)
            if (ev) (void)hipEventDestroy(ev);
        (void)hipFree(t->d_choice);
        (void)hipFree(t->q_tmp);
        if (t->mb_host) (void)hipHostFree(t->mb_host);
        (void)hipFree(t->d_fault_mb);
        if (t->mb_stream) (void)hipStreamDestroy(t->mb_stream);
    }
    if (t->dp) pipe_release(t->device);
    delete t;
}

int st_tree_info_get(const st_tree *t, st_tree_info *info)
try {
    if (!t || !info) return fail(ST_ERR_ARG, "tree or info is NULL");
    *info = t->info;
    info->strategy = t->strategy;
    info->big_batch_kernel = big_batch_kernel_of(t);      // (follows the handle's current options)
    info->a_side_bytes = t->has_canopy ? ((t->rec_a4 && t->d_rec_a4 && t->d_leaf_blocks) ? 4 : 8) : 0;
    info->b_table_bytes_per_leaf = t->has_canopy ? ((t->rec_a4 && t->cherries && t->d_rec_c && t->d_leaf_blocks) ? t->rec_bytes / 4 : t->rec_bytes / 2) : 0;
    info->ladder_sums = (t->ladder_sums && ladder_sums_ready(t)) ? 1 : 0;      // (for batches of up to ladder_sums_max_pairs, if that is set)
    info->ladder_sums_max_pairs = info->ladder_sums ? t->ladder_sums_max_pairs : 0;
    info->host_wire_bytes_in = t->wire48 && t->n_nodes <= 0xFFFFFF ? 6 : 8;
    info->host_wire_bytes_out = t->wire24 && t->n_nodes <= 0xFFFFFF ? 7 : 8;
    return ST_OK;
} ST_CATCH_ALL

int st_api_version(void) { return ST_API_VERSION; }

int st_tree_info_get_sized(const st_tree *t, void *info, int64_t info_bytes)
try {
    if (!t || !info) return fail(ST_ERR_ARG, "tree or info is NULL");
    if (info_bytes < 8 || info_bytes % 4) return fail(ST_ERR_ARG, "info_bytes must be a multiple of 4, at least 8");
    st_tree_info full;
    const int rc = st_tree_info_get(t, &full);
    if (rc != ST_OK) return rc;
    std::memcpy(info, &full, (size_t)std::min<int64_t>(info_bytes, (int64_t)sizeof full));
    return ST_OK;
} ST_CATCH_ALL

int st_tree_devices(const st_tree *t, int *devices, int capacity, int *n_devices)
try {
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    const int n = 1 + (int)t->peers.size();
    if (n_devices) *n_devices = n;
    if (devices)
        for (int i = 0; i < n && i < capacity; i++) devices[i] = i == 0 ? t->device : t->peers[(size_t)i - 1]->device;
    return ST_OK;
} ST_CATCH_ALL

int st_tree_set_strategy(st_tree *t, int strategy)
try {
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (strategy == ST_STRATEGY_AUTO) strategy = t->has_canopy ? ST_STRATEGY_CANOPY : ST_STRATEGY_WALK;
    if (strategy == ST_STRATEGY_CANOPY && !t->has_canopy)
        return fail(ST_ERR_ARG, "tree was built without canopy tables");
    if (strategy != ST_STRATEGY_CANOPY && strategy != ST_STRATEGY_WALK)
        return fail(ST_ERR_ARG, "unknown strategy " + std::to_string(strategy));
    t->strategy = strategy;
    for (st_tree *p : t->peers) p->strategy = strategy;
    return ST_OK;
} ST_CATCH_ALL

static int set_option_one(st_tree *t, const char *name, int64_t value)
{
    if (std::strcmp(name, "tile_sort") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "tile_sort must be 0 or 1");
        t->tile_sort = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "ladder_scalar") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "ladder_scalar must be 0 or 1");
        t->ladder_scalar = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "reserve_cus") == 0) {
        // the persistent kernels are sized to the CUs of the device; on a rank that also receives (the root of the
        // multi-GPU gather) RCCL's kernels need somewhere to run beside a 1024-lane / 144 KiB workgroup per CU
        if (value < 0 || value >= t->n_cu_device) return fail(ST_ERR_ARG, "reserve_cus must be in [0, CUs of the device)");
        t->n_cu = t->n_cu_device - (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "ladder_dynamic") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "ladder_dynamic must be 0 or 1");
        t->ladder_dynamic = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "cherries") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "cherries must be 0 or 1");
        t->cherries = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "batch_probe") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "batch_probe must be 0 or 1");
        t->batch_probe = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "measure") == 0) {      // (host_path.h::run_pipe; peers of a multi-device handle follow)
        if (value < 0 || value > 7) return fail(ST_ERR_ARG, "measure must be a combination of 1 (trace), 2 (skip CPU passes), 4 (skip GPU side)");
        t->measure = (int)value;
        for (st_tree *p : t->peers) p->measure = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "ladder_min_pairs") == 0) {
        if (value < 0) return fail(ST_ERR_ARG, "ladder_min_pairs must be >= 0");
        t->ladder_min_pairs = value;
        return ST_OK;
    }
    if (std::strcmp(name, "tree_rmq") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "tree_rmq must be 0 or 1");
        t->tree_rmq = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "mrca_ranks") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "mrca_ranks must be 0 or 1");
        t->mrca_ranks = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "rec_a4") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "rec_a4 must be 0 or 1");
        t->rec_a4 = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "walk_ladder") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "walk_ladder must be 0 or 1");
        t->walk_ladder = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "prefer_walk_sorted") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "prefer_walk_sorted must be 0 or 1");
        t->prefer_walk_sorted = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "wire48") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "wire48 must be 0 or 1");
        t->wire48 = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "wire24") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "wire24 must be 0 or 1");
        t->wire24 = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "sort_tile") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(ST_ERR_ARG, "sort_tile must be 0, 1, 2 or 4");
        t->sort_tile = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "walk_sort_min") == 0) {
        if (value < 0) return fail(ST_ERR_ARG, "walk_sort_min must be >= 0");
        t->walk_sort_min = value;
        return ST_OK;
    }
    if (std::strcmp(name, "walk_crown") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "walk_crown must be 0 or 1");
        t->walk_crown = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "walk_sort") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "walk_sort must be 0 or 1");
        t->walk_sort = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "lineage_lens") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "lineage_lens must be 0 or 1");
        t->lineage_lens = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "lineage_sums") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "lineage_sums must be 0 or 1");
        t->lineage_sums = (int)value;
        return ST_OK;
    }
    if (std::strcmp(name, "ladder_sums") == 0) {
        if (value != 0 && value != 1) return fail(ST_ERR_ARG, "ladder_sums must be 0 or 1");
        t->ladder_sums = (int)value;
        t->ladder_sums_max_pairs = 0;      // (an explicit setting holds for every batch size)
        return ST_OK;
    }
    if (std::strcmp(name, "small_batch_path") == 0) {
        t->small_batch_path = value != 0;
        return ST_OK;
    }
    return fail(ST_ERR_ARG, std::string("unknown option ") + name);
}

int st_tree_set_option(st_tree *t, const char *name, int64_t value)
try {
    if (!t || !name) return fail(ST_ERR_ARG, "tree or name is NULL");
    int rc = set_option_one(t, name, value);
    for (st_tree *p : t->peers)
        if (rc == ST_OK) rc = set_option_one(p, name, value);
    return rc;
} ST_CATCH_ALL

int st_distances_device(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t stride0,
                        int64_t stride1, double *d_out_dist, int32_t *d_out_mrca, void *stream)
try {
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !d_pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!d_out_dist && !d_out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    ST_DEVICE(t->device);
    return enqueue(t, d_pairs, n, stride0, stride1, DistSink{d_out_dist, nullptr}, MrcaSink{d_out_mrca, nullptr},
                   reinterpret_cast<hipStream_t>(stream));
} ST_CATCH_ALL

int st_distances_device_f32(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t stride0,
                            int64_t stride1, float *d_out_dist, int32_t *d_out_mrca, void *stream)
try {
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !d_pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!d_out_dist && !d_out_mrca) return fail(ST_ERR_ARG, "both outputs are NULL");
    ST_DEVICE(t->device);
    return enqueue(t, d_pairs, n, stride0, stride1, DistSink{nullptr, d_out_dist}, MrcaSink{d_out_mrca, nullptr},
                   reinterpret_cast<hipStream_t>(stream));
} ST_CATCH_ALL

int st_distances_device_wire(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t stride0, int64_t stride1,
                             float *d_out_dist, uint8_t *d_out_mrca24, void *stream)
try {
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n > 0 && !d_pairs) return fail(ST_ERR_ARG, "pairs is NULL");
    if (!d_out_dist && !d_out_mrca24) return fail(ST_ERR_ARG, "both outputs are NULL");
    if (t->n_nodes > 0xFFFFFF) return fail(ST_ERR_ARG, "24-bit MRCA ids need a tree of fewer than 2^24 nodes");
    if (reinterpret_cast<uintptr_t>(d_out_mrca24) & 3) return fail(ST_ERR_ARG, "d_out_mrca24 must be 4-byte aligned");
    ST_DEVICE(t->device);
    return enqueue(t, d_pairs, n, stride0, stride1, DistSink{nullptr, d_out_dist}, MrcaSink{nullptr, d_out_mrca24},
                   reinterpret_cast<hipStream_t>(stream));
} ST_CATCH_ALL

int st_unpack_mrca24_device(int device, const uint8_t *d_packed, int64_t n, int32_t *d_out_mrca, void *stream)
try {
    if (n < 0) return fail(ST_ERR_ARG, "n < 0");
    if (n == 0) return ST_OK;
    if (!d_packed || !d_out_mrca) return fail(ST_ERR_ARG, "buffer is NULL");
    ST_DEVICE(device);
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(k_unpack24, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_packed, (long long)n, d_out_mrca);
    ST_HIP(hipGetLastError());
    return ST_OK;
} ST_CATCH_ALL

int st_fault_check(st_tree *t, void *stream, int64_t *bad_id)
try {
    if (!t) return fail(ST_ERR_ARG, "tree is NULL");
    ST_DEVICE(t->device);
    Fault f;
    const int rc = fetch_fault(t->d_fault, reinterpret_cast<hipStream_t>(stream), f);
    if (rc != ST_OK) return rc;
    return report_fault(t->n_nodes, f, bad_id);
} ST_CATCH_ALL

int st_probe_last_choice(st_tree *t, void *stream, int *choice)
try {
    if (!t || !choice) return fail(ST_ERR_ARG, "tree or choice is NULL");
    *choice = -1;
    const unsigned issued = t->choice_next.load(std::memory_order_relaxed);
    if (!t->d_choice || issued == 0) return ST_OK;
    ST_DEVICE(t->device);
    ST_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    ST_HIP(hipMemcpy(choice, t->d_choice + (issued - 1) % kWorkSlots, sizeof(int), hipMemcpyDeviceToHost));
    return ST_OK;
} ST_CATCH_ALL

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
    HostOut out = make_host_out(out_dist, out_mrca);
    out.wire24 = wire24_of(t, out);
    // fresh result arrays: ask for huge pages before the first touch (a no-op on resident memory)
    if (out_dist && !looks_resident(out_dist, n * 8)) advise_huge(out_dist, n * 8);
    if (out_mrca && !looks_resident(out_mrca, n * 4)) advise_huge(out_mrca, n * 4);

    // Ids cross PCIe as int32 (half the H2D bytes).  Values that do not fit are clamped to
    // INT32_MAX / INT32_MIN -- still out of range for the kernel -- and their exact extremes
    // are kept here so that the reported id is the one the reference would report.
    auto work = [&](st_tree *r, const ChunkSeq &seq, Fault &fault) -> int {
        CopyPool &pool = r->dp->pipe.pool;
        hipStream_t s0 = nullptr;
        std::mutex wide_mutex;
        Fault wide = kFaultInit;
        // trees with fewer than 2^24 nodes: 24 bits per id on the wire (6 instead of 8 bytes per pair over the link;
        // the range check is then the host's, host_copy.h::pack_pairs48)
        const bool wire48 = r->wire48 && r->n_nodes <= 0xFFFFFF;
        auto pack = [&](PipeSlot &s, int64_t off, int64_t m) {
            const Id *src = pairs + off * stride0;
            int32_t *dst = static_cast<int32_t *>(s.h_in);
            const bool c_order = stride0 == 2 && stride1 == 1;
            if (wire48) {
                const long long n_nodes = r->n_nodes;
                pool.parallel_for(m, [&, src, dst, n_nodes](int64_t b, int64_t e) {
                    long long hi = kFaultInit.max_bad, lo = kFaultInit.min_bad;
                    pack_pairs48(reinterpret_cast<uint8_t *>(dst), b, src + b * stride0, e - b, stride0, stride1, n_nodes, hi, lo);
                    if (hi != kFaultInit.max_bad || lo != kFaultInit.min_bad) {
                        std::lock_guard<std::mutex> g(wide_mutex);
                        wide.max_bad = std::max(wide.max_bad, hi);
                        wide.min_bad = std::min(wide.min_bad, lo);
                    }
                });
                return;
            }
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
            return launch_chunk(r, s, off, m, wire48 ? 6 : 8, out,
                                [wire48](const void *in) { return SrcContig32{static_cast<const int *>(in), wire48 ? 1 : 0}; });
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
        // (24-bit wire format: every id out of range was seen, and its exact value kept, by the packing step)
        if (wire48) fault = wide;
        else if (fault.max_bad != kFaultInit.max_bad || fault.min_bad != kFaultInit.min_bad) merge_fault(fault, wide);
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
try {
    return distances_host_impl(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);
} ST_CATCH_ALL

int st_distances_host_i32(st_tree *t, const int32_t *pairs, int64_t n, int64_t stride0, int64_t stride1,
                          double *out_dist, int32_t *out_mrca, int64_t *bad_id)
try {
    return distances_host_impl(t, pairs, n, stride0, stride1, out_dist, out_mrca, bad_id);
} ST_CATCH_ALL

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
try {
    int rc = triangle_args(t, d_ids, m, k_begin, k_count, d_out_dist, d_out_mrca);
    if (rc != ST_OK) return rc;
    ST_DEVICE(t->device);
    const SrcTriangle src{reinterpret_cast<const long long *>(d_ids), (long long)id_stride, (long long)k_begin};
    return enqueue_src(t, src, k_count, DistSink{d_out_dist, nullptr}, MrcaSink{d_out_mrca, nullptr}, t->d_fault,
                       reinterpret_cast<hipStream_t>(stream));
} ST_CATCH_ALL

int st_triangle_host(st_tree *t, const int64_t *ids, int64_t m, int64_t id_stride, int64_t k_begin,
                     int64_t k_count, double *out_dist, int32_t *out_mrca, int64_t *bad_id)
try {
    int rc = triangle_args(t, ids, m, k_begin, k_count, out_dist, out_mrca);
    if (rc != ST_OK) return rc;
    if (k_count == 0) return ST_OK;
    HostOut out = make_host_out(out_dist, out_mrca);
    out.wire24 = wire24_of(t, out);
    if (out_dist && !looks_resident(out_dist, k_count * 8)) advise_huge(out_dist, k_count * 8);
    if (out_mrca && !looks_resident(out_mrca, k_count * 4)) advise_huge(out_mrca, k_count * 4);
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
} ST_CATCH_ALL

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
try {
    int rc = grid_args(t, row_ids, n_rows, col_ids, n_cols, e_begin, e_count, out_dist, out_mrca);
    if (rc != ST_OK) return rc;
    if (symmetric && (n_rows != n_cols)) return fail(ST_ERR_ARG, "symmetric grid needs n_rows == n_cols");
    if (e_count == 0) return ST_OK;
    HostOut out = make_host_out(out_dist, out_mrca);
    out.wire24 = wire24_of(t, out);
    if (out_dist && !looks_resident(out_dist, e_count * 8)) advise_huge(out_dist, e_count * 8);
    if (out_mrca && !looks_resident(out_mrca, e_count * 4)) advise_huge(out_mrca, e_count * 4);
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
} ST_CATCH_ALL

int st_knn_host(st_tree *t, const int64_t *queries, int64_t n_queries, const int64_t *cands, int64_t n_cands,
                int k, int skip_self, int64_t *out_index, double *out_dist, int64_t *bad_id)
try {
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
        const int rc = enqueue_src(t, src, rows * n_cands, DistSink{nullptr, d_tmp}, MrcaSink{nullptr, nullptr}, t->d_fault_host, stream);
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
} ST_CATCH_ALL

int st_quartets_host(st_tree *t, const int64_t *quartets, int64_t n, int64_t stride0, int64_t stride1,
                     int64_t *out_topologies, int64_t *bad_id)
try {
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
        if (canopy && 6 * m >= canopy_min_pairs(t)) {
            // six MRCA ids per quartet out of the canopy / rank-table kernels, then the pick
            int32_t *tmp = static_cast<int32_t *>(t->q_tmp) + (size_t)slot_index * (size_t)chunk * 6;
            const int rc = enqueue_src(t, SrcQuartet{static_cast<const long long *>(s.d_in)}, 6 * m,
                                       DistSink{nullptr, nullptr}, MrcaSink{tmp, nullptr}, t->d_fault_host, s.stream);
            if (rc != ST_OK) return bail(rc, g_last_error);
            e = launch_quartet_pick(t, static_cast<const long long *>(s.d_in), static_cast<const int *>(tmp), m,
                                    reinterpret_cast<long long *>(out_of(s)), s.stream);
        } else {
            e = launch_quartets_walk(t, static_cast<const long long *>(s.d_in), m, reinterpret_cast<long long *>(out_of(s)),
                                     t->d_fault_host, s.stream);
        }
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
} ST_CATCH_ALL

int st_graph_matrices_host(int device, int64_t n, int64_t n_edges, const int32_t *u, const int32_t *v,
                           const double *w, double *out_adjacency, double *out_laplacian)
try {
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
} ST_CATCH_ALL

int st_device_malloc(int device, int64_t bytes, void **out)
try {
    if (!out || bytes < 0) return fail(ST_ERR_ARG, "bad arguments");
    ST_DEVICE(device);
    ST_HIP(hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16)));
    return ST_OK;
} ST_CATCH_ALL

int st_device_free(int device, void *ptr)
try {
    ST_DEVICE(device);
    ST_HIP(hipFree(ptr));
    return ST_OK;
} ST_CATCH_ALL

int st_memcpy_h2d(int device, void *dst, const void *src, int64_t bytes)
try {
    ST_DEVICE(device);
    ST_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return ST_OK;
} ST_CATCH_ALL

int st_memcpy_d2h(int device, void *dst, const void *src, int64_t bytes)
try {
    ST_DEVICE(device);
    ST_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return ST_OK;
} ST_CATCH_ALL

int st_device_synchronize(int device)
try {
    ST_DEVICE(device);
    ST_HIP(hipDeviceSynchronize());
    return ST_OK;
} ST_CATCH_ALL

}  // extern "C"
