// host_tune.h -- part of suchtree_hip.hip (included after host_path.h, before host_upload.h).
// Which kernel the large distance batches of a deep tree get, decided by timing when the tree is created.
//
// A deep tree (canopy more than kDeepCanopyDepth edges deep) has up to three kernels that produce the same bits:
// the tile-sorted canopy kernel, the predicated canopy kernel and the tile-sorted walk kernel.  Which one is fastest
// depends on the shape of the tree in ways no single statistic captured (launch_policy.h has the numbers), and the
// spread is 2-4x, so the handle times them once on a sample of 2^22 random leaf pairs drawn on the device -- 6 ms on ml.tree next to the 0.07-2 s
// the tables of such a tree take to build -- and sets its defaults (tile_sort, pairs_per_lane, prefer_walk_sorted)
// to the fastest.  st_tree_set_option / st_tree_set_strategy still override them.  SUCHTREE_AMD_AUTOTUNE=0: the
// fixed rule instead.  Never an error: if anything here fails the rule's defaults stay.
#pragma once

constexpr int64_t kTunePairs = (int64_t)1 << 22;      // (large enough for every candidate's largest tiles on every CU: with 2^20 a 13 % gap at 2e7 pairs went unseen)

// What the handle's current settings select for a large batch with distances.
static int big_batch_kernel_of(const st_tree *t)
{
    if (t->strategy != ST_STRATEGY_CANOPY) return ST_KERNEL_WALK;
    if (prefers_walk_sorted(t, (int64_t)1 << 40, true)) return ST_KERNEL_WALK_SORTED;
    if (t->tile_sort && sorted_q(t) > 0) return ST_KERNEL_CANOPY_SORTED;
    return t->pairs_per_lane == 0 ? ST_KERNEL_CANOPY_SCALAR : ST_KERNEL_CANOPY;
}

static void rule_for_deep_tree(st_tree *t)
{
    t->pairs_per_lane = 0;
    t->tile_sort = 1;
    // 63-slot chains: the tile-sorted canopy kernel reads them through a pointer and never won a measurement
    if (t->rec_cap > 31 || sorted_q(t) <= 0) { t->pairs_per_lane = 1; t->tile_sort = 0; }
    t->prefer_walk_sorted = walk_sorted_by_rule(t) ? 1 : 0;
}

static void tune_deep_tree(st_tree *t, const TreeTables &T)
{
    rule_for_deep_tree(t);
    if (const char *env = std::getenv("SUCHTREE_AMD_AUTOTUNE"))
        if (env[0] == '0') return;
    // sample: uniform random leaf pairs (the reference's typical query, and the bench's)
    std::vector<int32_t> leaves;
    {
        std::vector<uint8_t> has_child((size_t)T.n, 0);
        for (int64_t x = 0; x < T.n; x++)
            if (T.nodes[(size_t)x].parent >= 0) has_child[(size_t)T.nodes[(size_t)x].parent] = 1;
        for (int64_t x = 0; x < T.n; x++)
            if (!has_child[(size_t)x]) leaves.push_back((int32_t)x);
    }
    if (leaves.size() < 2) return;
    const int64_t n = kTunePairs;
    int32_t *d_leaves = nullptr;
    long long *d_pairs = nullptr;
    double *d_dist = nullptr;
    int32_t *d_mrca = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const std::string keep = g_last_error;
    bool ok = hipMalloc(reinterpret_cast<void **>(&d_leaves), leaves.size() * 4) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&d_pairs), (size_t)n * 16) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&d_dist), (size_t)n * 8) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&d_mrca), (size_t)n * 4) == hipSuccess &&
              hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
              hipMemcpyAsync(d_leaves, leaves.data(), leaves.size() * 4, hipMemcpyHostToDevice, stream) == hipSuccess;
    if (ok) {      // the sample is drawn on the device (k_sample_leaf_pairs, kernels_misc.h)
        hipLaunchKernelGGL(k_sample_leaf_pairs, dim3(1024), dim3(256), 0, stream, d_leaves, (unsigned)leaves.size(), d_pairs, (long long)n);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    }
    // milliseconds of the fastest of three launches after one warm-up, or a negative number
    auto time_settings = [&](int tile_sort, int ppl, int walk) -> float {
        t->tile_sort = tile_sort;
        t->pairs_per_lane = ppl;
        t->prefer_walk_sorted = walk;
        float best = -1.0f;
        for (int rep = 0; rep < 4; rep++) {
            if (hipEventRecord(e0, stream) != hipSuccess) return -1.0f;
            if (enqueue_src(t, SrcContig{d_pairs}, n, DistSink{d_dist, nullptr}, d_mrca, t->d_fault, stream) != ST_OK) return -1.0f;
            if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0f;
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0f;
            if (rep > 0 && (best < 0.0f || ms < best)) best = ms;
        }
        return best;
    };
    if (ok) {
        const int rule_sort = t->tile_sort, rule_ppl = t->pairs_per_lane, rule_walk = t->prefer_walk_sorted;
        // the canopy family's two forms first; the walk kernel has to beat the better of them
        int best_sort = rule_sort, best_ppl = rule_ppl;
        float best_ms = -1.0f;
        if (sorted_q(t) > 0) best_ms = time_settings(1, 0, 0), best_sort = 1, best_ppl = 0;
        const float ilp_ms = time_settings(0, 1, 0);
        if (ilp_ms > 0.0f && (best_ms < 0.0f || ilp_ms < best_ms)) best_ms = ilp_ms, best_sort = 0, best_ppl = 1;
        int best_walk = 0;
        t->prefer_walk_sorted = 1;
        if (prefers_walk_sorted(t, n, true)) {
            const float walk_ms = time_settings(best_sort, best_ppl, 1);
            if (walk_ms > 0.0f && (best_ms < 0.0f || walk_ms < best_ms)) best_ms = walk_ms, best_walk = 1;
        }
        if (best_ms > 0.0f) {
            t->tile_sort = best_sort;
            t->pairs_per_lane = best_ppl;
            t->prefer_walk_sorted = best_walk;
            t->info.tuned = 1;
        } else {
            t->tile_sort = rule_sort;
            t->pairs_per_lane = rule_ppl;
            t->prefer_walk_sorted = rule_walk;
        }
    }
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (stream) (void)hipStreamDestroy(stream);
    (void)hipFree(d_leaves);
    (void)hipFree(d_pairs);
    (void)hipFree(d_dist);
    (void)hipFree(d_mrca);
    g_last_error = keep;
}
