// host_launch.h -- part of libsuchtree_hip.so's single translation unit (included by suchtree_hip.hip after
// the kernels, in this order: host_tree.h, host_launch.h, host_path.h, host_upload.h).
// Which kernel a request gets (launch shapes, thresholds), enqueueing, fault-word read-back.
#pragma once

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

// Deep-canopy trees whose canopy image leaves the tile-sorted canopy kernel only small tiles (nj.tree: 9111
// canopy nodes = 146 KiB, 1024-pair tiles) are served faster by the tile-sorted WALK kernel once its crown
// ladder exists: a crown of <= 5120 nodes, 4096-pair tiles (nj.tree, 1e7 pairs: 1.73e10 against 1.60e10
// pairs/s; ml.tree keeps the canopy kernel: 2.15e10 against 1.73e10).  Large batches with distances only.
static bool walk_sorted_ready(const st_tree *t);
constexpr int64_t kWalkSortedMinPairs = 524288;
static bool prefers_walk_sorted(const st_tree *t, int64_t n, bool want_dist)
{
    if (t->strategy != ST_STRATEGY_CANOPY || !t->tile_sort || !want_dist || n < kWalkSortedMinPairs) return false;
    const int q = sorted_q(t);
    return q > 0 && q < 4 && t->walk_ladder && t->d_crown_ladder && t->walk_crown && walk_sorted_ready(t);
}

// In lineage-sum mode the tile-sorted canopy kernel reads every pair once (key phase; shared-portal pairs, rare,
// a second time) and all its stores are coalesced: it may work on the host path's pinned slots directly.
// SUCHTREE_AMD_SORTED_ZERO_COPY=0 puts the device staging back (measurement).
static bool sorted_zero_copy(const st_tree *t)
{
    static const bool on = !(std::getenv("SUCHTREE_AMD_SORTED_ZERO_COPY") && std::getenv("SUCHTREE_AMD_SORTED_ZERO_COPY")[0] == '0');
    return on && sorted_shape(t).sums;
}

static bool wants_device_stage(const st_tree *t, int64_t m)
{
    if (t->strategy != ST_STRATEGY_CANOPY || !t->tile_sort || sorted_q(t) <= 0) return false;
    if (prefers_walk_sorted(t, m, true)) return false;      // (that kernel reads every pair once and stores coalesced)
    if (sorted_zero_copy(t)) return false;
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
        // explicit pair arrays on trees with the four-byte a side: 4-byte gathers from a table half the size
        if constexpr (std::is_same<Src, SrcContig>::value || std::is_same<Src, SrcContig32>::value) {
            if (P.rec_a4 && P.leaf_blocks)
                return launch_canopy_k(k_canopy_ilp<CAP, 1, Src, true>, 1, t, P, src, n, out_d, out_m, fault, stream,
                                       canopy_lds_bytes(t) + leaf_block_image_bytes(P.leaf_block_count));
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
    P.rec_a4 = t->rec_a4 ? t->d_rec_a4 : nullptr;
    P.leaf_blocks = t->rec_a4 ? t->d_leaf_blocks : nullptr;
    P.leaf_block_shift = t->leaf_block_shift;
    P.leaf_block_count = t->leaf_block_count;
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
    P.rmq = t->tree_rmq ? t->d_tree_rmq : nullptr;
    P.n_nodes = t->n_nodes;
    P.crown_ladder = nullptr;
    if (t->d_lineage && t->d_lineage_node_rec && t->lineage_sums) {
        P.lineage.node_rec = t->d_lineage_node_rec;
        P.lineage.sums = t->d_lineage;
        P.lineage.lens = t->lineage_lens ? t->d_lineage_len : nullptr;
        P.lineage.shared_blocks = t->walk_crown != 0;
        if (t->walk_crown && t->d_crown_rmq) {
            P.lineage.crown_rmq = t->d_crown_rmq;
            P.lineage.crown_nodes = t->crown_nodes;
            if (t->walk_ladder) P.crown_ladder = t->d_crown_ladder;
        }
    }
    return P;
}

// Smallest batch the tile-sorted walk kernel takes: below it k_walk's finer grain wins (a 1e5-pair
// batch is 25 tiles of 4096 pairs on 256 CUs; measured on ml.tree, pairs per second unsorted / sorted:
// 1e5 pairs 4.9e9 / 1.3e9, 4e5 8.0e9 / 5.3e9, 8e5 9.1e9 / 1.03e10, 3.2e6 1.0e10 / 1.08e10, 1e7 1.1e10 /
// 1.35e10).  Tiles: 4096 pairs on trees with canopy tables, 2048 on trees only the walk family serves
// (1e6-leaf depth-338 tree, 1e7 / 4e7 pairs: 5.93e9 / 6.43e9 against 5.76e9 / 5.90e9 with 4096;
// ml.tree: 1.28e10 / 1.34e10 against 1.32e10 / 1.41e10).  (kWalkSortedMinPairs = 524288, above.)

static bool walk_sorted_ready(const st_tree *t)
{
    return t->walk_sort && t->tree_rmq && t->d_tree_rmq && t->lineage_sums && t->d_lineage && t->d_lineage_node_rec &&
           t->lineage_lens && t->d_lineage_len;
}

template <int Q, bool LADDER, typename Src>
static hipError_t launch_walk_sorted(const st_tree *t, const WalkParams &P, const Src &src, int64_t n, DistSink out_d,
                                     int32_t *out_m, Fault *fault, hipStream_t stream)
{
    constexpr int64_t tile = (int64_t)Q * kWalkSortBlock;
    const size_t lds = walk_sort_scratch_bytes(Q) + (LADDER ? (size_t)P.lineage.crown_nodes * 16 : 0);
    const int wg_per_cu = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / lds)));
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + tile - 1) / tile, (int64_t)t->n_cu * wg_per_cu));
    int key_shift = 0;      // keys are edge counts of b's lineage below the meeting node
    while ((t->info.depth >> key_shift) >= kWalkSortBuckets) key_shift++;
    auto kern = k_walk_sorted<Q, LADDER, Src>;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kWalkSortBlock), lds, stream, P, src, (long long)n, out_d, out_m, fault, key_shift);
    return hipGetLastError();
}

template <typename Src>
static hipError_t launch_walk(const st_tree *t, const Src &src, int64_t n, DistSink out_d,
                              int32_t *out_m, Fault *fault, hipStream_t stream)
{
    const WalkParams P = walk_params(t);
    if (out_d.any() && n >= kWalkSortedMinPairs && walk_sorted_ready(t)) {
        static const int forced = std::getenv("SUCHTREE_AMD_WALK_SORT_Q") ? std::atoi(std::getenv("SUCHTREE_AMD_WALK_SORT_Q")) : 0;   // tuning experiments
        if (P.crown_ladder) {
            // the crown's ladder in LDS: the largest tile that fits beside it
            const size_t image = (size_t)P.lineage.crown_nodes * 16;
            const int q = forced ? forced : image + walk_sort_scratch_bytes(4) <= 160 * 1024 ? 4 : image + walk_sort_scratch_bytes(2) <= 160 * 1024 ? 2 : 1;
            if (image + walk_sort_scratch_bytes(q) <= 160 * 1024) {
                if (q == 4) return launch_walk_sorted<4, true>(t, P, src, n, out_d, out_m, fault, stream);
                if (q == 2) return launch_walk_sorted<2, true>(t, P, src, n, out_d, out_m, fault, stream);
                return launch_walk_sorted<1, true>(t, P, src, n, out_d, out_m, fault, stream);
            }
        }
        const int q = forced ? forced : t->has_canopy ? 4 : 2;
        if (q == 4) return launch_walk_sorted<4, false>(t, P, src, n, out_d, out_m, fault, stream);
        if (q == 2) return launch_walk_sorted<2, false>(t, P, src, n, out_d, out_m, fault, stream);
        return launch_walk_sorted<1, false>(t, P, src, n, out_d, out_m, fault, stream);
    }
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
                         (allow_sorted || !(t->tile_sort && sorted_q(t) > 0) || sorted_zero_copy(t)) &&
                         !prefers_walk_sorted(t, n, d_out.any()));
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
