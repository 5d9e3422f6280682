// host_launch.h -- part of suchtree_hip.hip (included in this order: host_tree.h, host_launch.h, host_path.h,
// host_upload.h).  Which family a request gets (launch_policy.h has the thresholds, the launch units
// launch_walk.hip / launch_canopy.hip / launch_canopy_sorted.hip the kernels), enqueueing, fault-word read-back.
#pragma once
#include "launch_decl.h"
#include "launch_policy.h"

static const Fault kFaultInit = {std::numeric_limits<long long>::min(),
                                 std::numeric_limits<long long>::max()};

// Small batches are not worth staging 128 KiB of canopy per workgroup.

template <typename Src>
static int enqueue_src(st_tree *t, const Src &src, int64_t n, DistSink d_out, MrcaSink d_mrca,
                       Fault *fault, hipStream_t stream, bool allow_sorted = true)
{
    if (n == 0) return ST_OK;
    // MRCA-only requests (d_out == NULL) also go through the canopy kernels: the id comes out
    // of the same climb, and that is ~7x faster than walking the global table
    // (allow_sorted = false: pairs and results are in pinned host memory, which the tile-sorted
    // kernel must not work on -- it reads every pair twice and scatters its stores)
    // MRCA ids only, rank table available: k_mrca_ranks whatever the tree's depth (it reads every
    // pair once and stores coalesced, so it may also work on pinned host memory)
    const bool ranks_only = !d_out.any() && d_mrca.any() && mrca_ranks_ready(t) && n >= kCanopyMinPairs;
    const bool canopy = ranks_only ||
                        (t->strategy == ST_STRATEGY_CANOPY && n >= canopy_min_pairs(t) &&
                         (allow_sorted || !(t->tile_sort && sorted_q(t) > 0) || sorted_zero_copy(t)) &&
                         !prefers_walk_sorted(t, n, d_out.any()));
    // Large batches of explicit pairs that the scalar ladder kernel would take, on a tree whose tile-sorted walk kernel is
    // ready as well: the batch itself decides (pair_math.h: probe_says_walk) -- both kernels are enqueued, every workgroup
    // of either samples the batch, the kernel the sample does not choose returns at once.  From kProbeMinPairs pairs: the empty
    // dispatch and the sampling cost 8-10 us per batch (round 5's separate probe kernel: 20; profiles/default_vs_matrix_r06.log).
    // Pair generators (triangle, grid, quartets) keep the handle's choice: clade triangles run fastest on the ladder kernel
    // (profiles/clade_triangles_r04.log).
    constexpr bool explicit_pairs = std::is_same<Src, SrcContig>::value || std::is_same<Src, SrcContig32>::value || std::is_same<Src, SrcStrided>::value;
    if (explicit_pairs && canopy && !ranks_only && allow_sorted && t->batch_probe && t->d_choice && d_out.any() && ladder_applies(t, n) &&
        n >= std::max<int64_t>(walk_sorted_min_pairs(t), kProbeMinPairs) && t->walk_ladder && t->d_crown_ladder && t->walk_crown && walk_sorted_ready(t)) {
        // (the word only reports the verdict -- st_probe_last_choice --, nothing waits on it: a ring of slots so that launches in
        // flight on several streams do not write the one a reader is about to fetch)
        const unsigned slot = t->choice_next.fetch_add(1, std::memory_order_relaxed) % kWorkSlots;
        int *choice = t->d_choice + slot;
        hipError_t e = launch_canopy(t, src, n, d_out, d_mrca, fault, stream, choice);
        if (e == hipSuccess) e = launch_walk(t, src, n, d_out, d_mrca, fault, stream, choice);
        if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        return ST_OK;
    }
    const hipError_t e = canopy ? launch_canopy(t, src, n, d_out, d_mrca, fault, stream)
                                : launch_walk(t, src, n, d_out, d_mrca, fault, stream);
    if (e != hipSuccess) return fail(ST_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return ST_OK;
}

static int enqueue(st_tree *t, const int64_t *d_pairs, int64_t n, int64_t s0, int64_t s1,
                   DistSink d_out, MrcaSink d_mrca, hipStream_t stream)
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
