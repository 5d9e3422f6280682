// host_tune.h -- part of suchtree_hip.hip (included after host_path.h, before host_upload.h).
// Which kernel the large distance batches of a deep tree get, decided by timing when the tree is created.
//
// A deep tree (canopy more than kDeepCanopyDepth edges deep) has up to three kernels that produce the same bits:
// the tile-sorted canopy kernel, the predicated canopy kernel and the tile-sorted walk kernel.  Which one is fastest
// depends on the shape of the tree in ways no single statistic captured (launch_policy.h has the numbers), and the
// spread is 2-4x, so the handle times them once on a sample of 2^23 random leaf pairs drawn on the device -- ~15 ms on ml.tree next to the 0.07-2 s
// the tables of such a tree take to build -- and sets its defaults (tile_sort, prefer_walk_sorted, ladder_scalar)
// to the fastest -- if it beats the rule's own choice by more than 5 % (kTuneMargin), else the rule stands.  The decision
// is recorded per (tree digest, device, library build) and read back by later handles and processes, so a handle's
// kernel is stable across runs (info.tuned: 1 = timed now, 2 = read from the record); multi-device handles time on
// the primary and copy the settings to the peers.  st_tree_set_option / st_tree_set_strategy still override them.
// SUCHTREE_AMD_AUTOTUNE=0: the fixed rule instead.  Results are bit-identical whatever is chosen.  Never an error:
// if anything here fails the rule's defaults stay.
#pragma once
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

constexpr int64_t kTunePairs = (int64_t)1 << 23;      // (large enough for every candidate's largest tiles on every CU: with 2^20 a 13 % gap at 2e7 pairs went unseen, with 2^22 nj.tree's 16 % gap between the scalar ladder kernel and the tile-sorted walk kernel shrank below the margin)

// What the handle's current settings select for a large batch with distances.
static int big_batch_kernel_of(const st_tree *t)
{
    if (t->strategy != ST_STRATEGY_CANOPY) return ST_KERNEL_WALK;
    if (ladder_scalar_ready(t)) return ST_KERNEL_CANOPY_LADDER;      // (launch_canopy.hip: the first choice of large batches)
    if (prefers_walk_sorted(t, (int64_t)1 << 40, true)) return ST_KERNEL_WALK_SORTED;
    if (t->tile_sort && sorted_q(t) > 0) return ST_KERNEL_CANOPY_SORTED;
    return ST_KERNEL_CANOPY;
}

static void rule_for_deep_tree(st_tree *t)
{
    t->tile_sort = 1;
    // 63-slot chains: the tile-sorted canopy kernel reads them through a pointer and never won a measurement
    if (t->rec_cap > 31 || sorted_q(t) <= 0) t->tile_sort = 0;
    t->prefer_walk_sorted = walk_sorted_by_rule(t) ? 1 : 0;
    // where its image fits the scalar ladder kernel has won every tree measured (profiles/kernel_win_matrix_r06.json; 1 KB
    // records: no other canopy kernel reads them well); its joint form is a matter of timing -- not by rule
    t->ladder_scalar = ladder_tables_ready(t) ? 1 : 0;
    t->ladder_sums = 0;
    t->ladder_sums_max_pairs = 0;
}

// ---- persistent record of what a tree measured -------------------------------------------------------------
// Results are identical whatever the choice, but a handle's kernel should not change from run to run because a
// busy GPU flipped a close race: the decision is written to a small file keyed by (tree digest, device name,
// build of this library) and read back by later processes.  $SUCHTREE_AMD_CACHE_DIR, else $XDG_CACHE_HOME/suchtree_amd,
// else ~/.cache/suchtree_amd; SUCHTREE_AMD_TUNE_CACHE=0: neither read nor written.  Never an error.
static uint64_t tree_digest(const TreeTables &T)
{
    uint64_t h = 1469598103934665603ull;      // FNV-1a over the {parent, distance} table, eight bytes at a time
    const uint64_t *w = reinterpret_cast<const uint64_t *>(T.nodes.data());
    for (int64_t k = 0; k < T.n; k++) { h ^= w[k]; h *= 1099511628211ull; }
    h ^= (uint64_t)T.n; h *= 1099511628211ull;
    return h;
}

static std::string tune_cache_path(const st_tree *t, const TreeTables &T, const char *device_name)
{
    if (const char *env = std::getenv("SUCHTREE_AMD_TUNE_CACHE"))
        if (env[0] == '0') return std::string();
    std::string dir;
    if (const char *env = std::getenv("SUCHTREE_AMD_CACHE_DIR")) dir = env;
    else if (const char *xdg = std::getenv("XDG_CACHE_HOME")) dir = std::string(xdg) + "/suchtree_amd";
    else if (const char *home = std::getenv("HOME")) dir = std::string(home) + "/.cache/suchtree_amd";
    if (dir.empty()) return std::string();
    uint64_t h = tree_digest(T);
    for (const char *c = device_name; c && *c; c++) { h ^= (uint64_t)(unsigned char)*c; h *= 1099511628211ull; }
    for (const char *c = __DATE__ " " __TIME__; *c; c++) { h ^= (uint64_t)(unsigned char)*c; h *= 1099511628211ull; }      // (a rebuilt library measures again)
    h ^= (uint64_t)t->n_cu; h *= 1099511628211ull;
    char name[64];
    std::snprintf(name, sizeof name, "/tune-%016llx.txt", (unsigned long long)h);
    return dir + name;
}

static bool tune_cache_read(const std::string &path, int &tile_sort, int &walk, int &ladder, long long &ladder_min, int &ladder_sums,
                            long long &ladder_sums_max)
{
    if (path.empty()) return false;
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    int a = -1, c = -1, d = -1, g = -1;
    long long e = -1, h = -1;
    const int got = std::fscanf(f, "%d %d %d %lld %d %lld", &a, &c, &d, &e, &g, &h);
    std::fclose(f);
    if (got != 6 || (a != 0 && a != 1) || (c != 0 && c != 1) || (d != 0 && d != 1) || e < 0 || (g != 0 && g != 1) || h < 0) return false;
    tile_sort = a; walk = c; ladder = d; ladder_min = e; ladder_sums = g; ladder_sums_max = h;
    return true;
}

static void tune_cache_write(const std::string &path, int tile_sort, int walk, int ladder, long long ladder_min, int ladder_sums,
                             long long ladder_sums_max)
{
    if (path.empty()) return;
    const size_t slash = path.rfind('/');
    const std::string dir = path.substr(0, slash);
    for (size_t k = 1; k <= dir.size(); k++)      // mkdir -p
        if (k == dir.size() || dir[k] == '/') (void)::mkdir(dir.substr(0, k).c_str(), 0777);
    const std::string tmp = path + ".tmp." + std::to_string((long long)::getpid());
    FILE *f = std::fopen(tmp.c_str(), "w");
    if (!f) return;
    std::fprintf(f, "%d %d %d %lld %d %lld\n", tile_sort, walk, ladder, ladder_min, ladder_sums, ladder_sums_max);
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}

// A candidate has to beat the rule's own choice by this much before the handle departs from it (timing noise on a
// shared GPU is a few per cent; the gaps that matter are 1.3-4x).
constexpr float kTuneMargin = 0.95f;

static void copy_tuned_settings(st_tree *to, const st_tree *from)
{
    to->tile_sort = from->tile_sort;
    to->prefer_walk_sorted = from->prefer_walk_sorted;
    to->ladder_scalar = from->ladder_scalar;
    to->ladder_min_pairs = from->ladder_min_pairs;
    to->ladder_sums = from->ladder_sums;
    to->ladder_sums_max_pairs = from->ladder_sums_max_pairs;
    to->info.tuned = from->info.tuned;
}

static void tune_deep_tree(st_tree *t, const TreeTables &T, const char *device_name)
{
    rule_for_deep_tree(t);
    if (const char *env = std::getenv("SUCHTREE_AMD_AUTOTUNE"))
        if (env[0] == '0') return;
    const std::string cache = tune_cache_path(t, T, device_name);
    {
        int a, c, d, g;
        long long e, h;
        if (tune_cache_read(cache, a, c, d, e, g, h)) {
            // (a recorded choice the handle cannot serve -- other table budget, other options -- is ignored)
            const int keep_sort = t->tile_sort, keep_walk = t->prefer_walk_sorted, keep_ladder = t->ladder_scalar;
            t->tile_sort = a; t->prefer_walk_sorted = c; t->ladder_scalar = 0;
            const bool ok = (!a || sorted_q(t) > 0) && (!c || prefers_walk_sorted(t, kTunePairs, true)) && (!d || ladder_tables_ready(t)) &&
                            (!g || ladder_sums_ready(t));
            if (ok) { t->ladder_scalar = d; t->ladder_min_pairs = e; t->ladder_sums = g; t->ladder_sums_max_pairs = h; t->info.tuned = 2; return; }
            t->tile_sort = keep_sort; t->prefer_walk_sorted = keep_walk; t->ladder_scalar = keep_ladder;
        }
    }
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
    int64_t n = kTunePairs;
    {   // 28 bytes per sampled pair: a quarter of the sample when that is more than 1/8 of the free HBM, none below 1/2
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
        if ((size_t)n * 28 > free_b / 8) n >>= 2;
        if ((size_t)n * 28 + leaves.size() * 4 > free_b / 2) return;
    }
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
    // milliseconds of the fastest of three launches of m pairs after one warm-up, or a negative number
    auto time_settings = [&](int tile_sort, int walk, int ladder, int64_t m) -> float {
        t->tile_sort = tile_sort;
        t->prefer_walk_sorted = walk;
        t->ladder_scalar = ladder;
        t->ladder_min_pairs = 0;
        float best = -1.0f;
        for (int rep = 0; rep < 4; rep++) {
            if (hipEventRecord(e0, stream) != hipSuccess) return -1.0f;
            if (enqueue_src(t, SrcContig{d_pairs}, m, DistSink{d_dist, nullptr}, MrcaSink{d_mrca, nullptr}, t->d_fault, stream) != ST_OK) return -1.0f;
            if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0f;
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0f;
            if (rep > 0 && (best < 0.0f || ms < best)) best = ms;
        }
        return best;
    };
    const int keep_probe = t->batch_probe;
    t->batch_probe = 0;      // (the candidates are timed bare: the probe's fixed cost is the same for whichever kernel the handle keeps, and it would decide the mid-size comparisons)
    if (ok) {
        const int rule_sort = t->tile_sort, rule_walk = t->prefer_walk_sorted, rule_ladder = t->ladder_scalar;
        // Two batch sizes: the whole sample (what bulk callers send) and a quarter of it.  The scalar ladder kernel sorts
        // nothing, so its waves finish unevenly and few tiles per wave leave a long tail: it can win at 2^23 pairs and
        // lose at 2^21 (launch_policy.h) -- then it only takes the batches beyond the size in between.
        struct Cand { int sort, walk, ladder; float ms, ms_small; };
        const int64_t n_small = n / 4;
        std::vector<Cand> cands;
        // The scalar ladder kernel has two forms (kernels_canopy.h): both sides climbed, or a's side read from the lineage sums
        // (one fabric read more, one LDS climb less: nj.tree +17 %, ml.tree -13 %, profiles/ladder_joint_r06.log).  The faster
        // one -- the joint form only when it is ahead by the margin -- is what the ladder candidate below runs.
        // Timed at the whole sample and at a sixteenth of it: on ml.tree the joint form leads below 2^20 pairs (where the
        // tile-sorted kernel used to serve, 8-13 % ahead of the climbing form) and trails above -- it then takes the batches
        // up to an eighth of the sample only.
        t->ladder_sums = 0;
        t->ladder_sums_max_pairs = 0;
        if (ladder_sums_ready(t)) {
            const float climb_big = time_settings(0, 0, 1, n), climb_small = time_settings(0, 0, 1, n / 16);
            t->ladder_sums = 1;
            const float sums_big = time_settings(0, 0, 1, n), sums_small = time_settings(0, 0, 1, n / 16);
            const bool big = climb_big > 0.0f && sums_big > 0.0f && sums_big < kTuneMargin * climb_big;
            const bool small = climb_small > 0.0f && sums_small > 0.0f && sums_small < kTuneMargin * climb_small;
            t->ladder_sums = (big || small) ? 1 : 0;
            t->ladder_sums_max_pairs = (!big && small) ? n / 8 : 0;
        }
        t->ladder_scalar = 0;
        if (sorted_q(t) > 0) cands.push_back({1, 0, 0, -1.0f, -1.0f});
        cands.push_back({0, 0, 0, -1.0f, -1.0f});
        if (ladder_tables_ready(t)) cands.push_back({0, 0, 1, -1.0f, -1.0f});
        t->prefer_walk_sorted = 1;
        if (prefers_walk_sorted(t, n_small, true)) cands.push_back({rule_sort, 1, 0, -1.0f, -1.0f});
        for (Cand &c : cands) {
            c.ms = time_settings(c.sort, c.walk, c.ladder, n);
            c.ms_small = time_settings(c.sort, c.walk, c.ladder, n_small);
        }
        const Cand *best = nullptr, *rule = nullptr, *base = nullptr, *base_small = nullptr;
        for (const Cand &c : cands) {
            if (c.ms <= 0.0f || c.ms_small <= 0.0f) continue;
            if (!best || c.ms < best->ms) best = &c;
            if (!c.ladder && (!base || c.ms < base->ms)) base = &c;                            // fastest without the ladder kernel, whole sample ...
            if (!c.ladder && (!base_small || c.ms_small < base_small->ms_small)) base_small = &c;   // ... and at the smaller size
            // the rule's choice among the candidates (with the walk kernel chosen, the canopy settings behind it do not matter)
            if (c.walk == rule_walk && (c.walk || (c.ladder == rule_ladder && c.sort == rule_sort))) rule = &c;
        }
        if (best && rule && best != rule && best->ms > kTuneMargin * rule->ms) best = rule;      // too close to call: the rule stands
        if (best) {
            // what serves when the ladder kernel does not: its own competitor at the smaller size, else the winner
            const Cand *other = best->ladder ? (base_small ? base_small : base) : best;
            // (the walk kernel only takes batches of 524288 pairs and more: below that a canopy kernel serves)
            const Cand *canopy = other && !other->walk ? other : nullptr;
            if (!canopy)
                for (const Cand &c : cands)
                    if (!c.walk && !c.ladder && c.ms_small > 0.0f && (!canopy || c.ms_small < canopy->ms_small)) canopy = &c;
            t->tile_sort = canopy ? canopy->sort : rule_sort;
            t->prefer_walk_sorted = other && other->walk ? 1 : 0;
            t->ladder_scalar = best->ladder;
            t->ladder_min_pairs = 0;
            if (best->ladder && base_small && best->ms_small > kTuneMargin * base_small->ms_small) {
                t->ladder_min_pairs = n / 2;
            } else if (best->ladder && base_small) {
                // ahead at a quarter of the sample too: one more size, a sixteenth (ml.tree: the tile-sorted kernel is
                // 10-20 % ahead from 2^17 to 2^19 pairs, even at 2^20, behind from 2^21 -- profiles/ladder_midsize_r04.log)
                const Cand ladder = *best, other_small = *base_small;      // (time_settings overwrites the handle's settings)
                const float ms_ladder = time_settings(ladder.sort, ladder.walk, ladder.ladder, n / 16);
                const float ms_other = time_settings(other_small.sort, other_small.walk, other_small.ladder, n / 16);
                t->tile_sort = canopy ? canopy->sort : rule_sort;
                    t->prefer_walk_sorted = other && other->walk ? 1 : 0;
                t->ladder_scalar = best->ladder;
                t->ladder_min_pairs = (ms_ladder > 0.0f && ms_other > 0.0f && ms_ladder > kTuneMargin * ms_other) ? n / 8 : 0;
            }
            if (t->rec_bytes > kMaxRecordBytes) t->ladder_scalar = 1;      // (1 KB records: the family's other kernels read them through a pointer, far slower)
            t->info.tuned = 1;
            tune_cache_write(cache, t->tile_sort, t->prefer_walk_sorted, t->ladder_scalar, (long long)t->ladder_min_pairs, t->ladder_sums,
                             (long long)t->ladder_sums_max_pairs);
        } else {
            t->tile_sort = rule_sort;
            t->prefer_walk_sorted = rule_walk;
            t->ladder_scalar = rule_ladder;
            t->ladder_min_pairs = 0;
            t->ladder_sums = 0;
            t->ladder_sums_max_pairs = 0;
        }
    } else {
        rule_for_deep_tree(t);
    }
    t->batch_probe = keep_probe;
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
