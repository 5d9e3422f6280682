// launch_policy.h -- host-only decisions shared by suchtree_hip.hip and the launch units: LDS budgets of a
// tree's launches, which form of the tile-sorted kernels fits, the batch-size thresholds between families.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "launch_geometry.h"
#include "st_tree.h"

namespace st {

static inline size_t canopy_lds_bytes(const st_tree *t)
{
    return (size_t)((t->canopy_nodes + 1) / 2) * 16;
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

static inline SortedShape sorted_shape(const st_tree *t)
{
    // chains of at most seven slots only (round 6): on records of 128 bytes and more the scalar ladder kernel has the same
    // image in LDS, reads every record once and won every cell of profiles/kernel_win_matrix_r06.json -- the tile-sorted
    // kernel's 15- / 31-slot and pointer forms are gone; on 16- to 64-byte records (small deep trees: a 2048-leaf
    // caterpillar, 3000 leaves at depth 292) it leads the walk family by 20-30 % and stays
    if (t->rec_cap > 7) return {0, false, false};
    const size_t image = ladder_image_bytes(t->canopy_nodes);
    const bool table = t->d_rmq != nullptr;
    const bool lineage = table && t->d_lineage != nullptr && t->d_rec_p != nullptr && t->lineage_sums;      // (rec_p: the canopy family's offsets into the table)
    struct Mode { bool rmq, sums; };
    // lineage sums first (they are worth a smaller tile), then the sparse table alone, then
    // the lock-step search; within a mode the largest tile that fits, two workgroups per CU if possible
    for (const Mode m : {Mode{true, true}, Mode{true, false}, Mode{false, false}}) {
        if ((m.rmq && !table) || (m.sums && !lineage)) continue;
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

static inline int sorted_q(const st_tree *t) { return sorted_shape(t).q; }

// Tile of one launch of a tile-sorted kernel, in units of 1024 pairs: q_max is what LDS admits (the tile of
// large batches: fewer image stagings, better sorted groups); a batch that would leave CUs without a tile at
// that size is cut finer, down to q_min (the smallest instantiated tile of the mode) -- a 262,144-pair batch on
// ml.tree is 64 tiles of 4096 pairs (45 us on a quarter of the CUs) or 256 of 1024 (22 us).  `forced`: the
// sort_tile option.  profiles/midsize_r03.log.
static inline int batch_tile_q(int q_max, int q_min, int forced, int64_t n, int n_cu)
{
    if ((forced == 1 || forced == 2 || forced == 4) && forced >= q_min && forced <= q_max) return forced;
    int q = q_max;
    while (q > q_min && (n + (int64_t)q * 1024 - 1) / ((int64_t)q * 1024) < (int64_t)n_cu) q >>= 1;
    return q;
}

// Smallest batch the canopy kernels take.  The tile-sorted kernel has a fixed cost (every
// workgroup stages a ladder image of up to 150 KiB, sorts, and on the host path its slot is
// staged through device memory), and with lineage sums the walk kernel does 8e9 pairs/s on deep
// trees: 10,000 pairs of ml.tree through the host path 73 us sorted, 40 us walked.
constexpr int64_t kCanopyMinPairs = 4096;
// deep canopies with lineage sums: below this the walk kernel wins (device-resident batches, us per call,
// k_walk / tile-sorted canopy kernel with 1024-pair tiles: ml.tree 32768 pairs 11.0 / 13.7, 65536 14.4 / 16.3,
// 131072 19.7 / 19.1, 262144 31.0 / 22.3; nj.tree 9.6 / 13.3, 12.8 / 17.0, 17.9 / 19.6, 26.9 / 23.2;
// profiles/midsize_r03.log)
constexpr int64_t kSortedMinPairs = 131072;

// k_canopy_ladder (launch_canopy.hip): records of 128 bytes and more, ladder image fits LDS.  The kernel reads every
// record once, keeps no scratch and sorts nothing, which pays on large batches only: a pass through it is a chain of
// four dependent global round trips and an LDS climb (17-30 us for ANY batch, k_walk: 8 us), and without a sort its
// waves finish unevenly, so few tiles per wave leave a long tail (us per call, ladder / tile-sorted canopy / walk
// kernels: ml.tree 2^17 pairs 20.8 / 19.0 / 20.0, 2^21 pairs 128 / 94 / 121, 2e7 pairs 700 / 880 / 1110; 1e6 leaves
// depth 173: 23 / 31 / 27 at 2^17, 136 / 240 / 300 at 2^21; 1e6 leaves depth 338: 300 / - / 304 at 2^21, 2270 / - /
// 2890 at 2e7; profiles/ladder_midsize_r04.log).  Its smallest batch is therefore timed too (host_tune.h).
constexpr int64_t kLadderMinPairs = 131072;
static inline bool ladder_tables_ready(const st_tree *t)
{
    return t->strategy == ST_STRATEGY_CANOPY && t->d_ladder && (t->rec_cap == 15 || t->rec_cap == 31 || t->rec_cap >= 63) &&
           ladder_kernel_lds_bytes(t->canopy_nodes) <= kLdsBytesPerCu;      // (the flags behind the image count too)
}
// the joint form of that kernel (kernels_canopy.h: ladder_pair_sums) needs the lineage sums, rec_p and the 64-bit sparse table
static inline bool ladder_sums_ready(const st_tree *t)
{
    return ladder_tables_ready(t) && t->d_rec_p && t->d_rmq64 && t->d_lineage && t->lineage_sums;
}
// ... and the handle may have found it ahead on batches up to a size only (host_tune.h: ml.tree below 2^20 pairs)
static inline bool ladder_sums_applies(const st_tree *t, int64_t n)
{
    return t->ladder_sums && (t->ladder_sums_max_pairs <= 0 || n <= t->ladder_sums_max_pairs) && ladder_sums_ready(t);
}
static inline bool ladder_applies(const st_tree *t, int64_t n)
{
    return t->ladder_scalar && n >= std::max<int64_t>(t->ladder_min_pairs, kLadderMinPairs) && ladder_tables_ready(t);
}
static inline bool ladder_scalar_ready(const st_tree *t) { return ladder_applies(t, (int64_t)1 << 40); }

// Smallest batch the canopy kernels take: 4096 pairs -- or, on deep trees whose heavy kernels (tile-sorted, scalar
// ladder) have a fixed cost of 15-30 us and whose walk kernel has a's side in one read (lineage sums by node id),
// kSortedMinPairs.
static inline int64_t canopy_min_pairs(const st_tree *t)
{
    const bool walk_is_quick = t->d_lineage && t->d_lineage_node_rec && t->lineage_sums;
    const bool sorted = t->tile_sort && t->d_rec_p && sorted_q(t) > 0;
    const bool ladder = t->ladder_scalar && ladder_tables_ready(t);
    // (deep canopies without either -- an image beyond the ladder's 10238 nodes: 3e5 leaves at depth 311 -- leave the predicated
    // kernel a 120+ KiB image to stage per workgroup and a climb of hundreds of rounds: 38 us for ANY batch up to 65536 pairs where
    // k_walk takes 15, profiles/default_vs_matrix_r06.log)
    const bool deep = t->canopy_depth > kShallowCanopyDepth;
    // (... and where the ladder kernel is the only canopy kernel fit for a deep canopy -- records of 128 bytes and more, since
    // round 6 -- the walk family also serves the batches below the ladder kernel's smallest: the predicated kernel there was 2-3x
    // behind k_walk / k_walk_sorted on every such tree of the matrix)
    if (walk_is_quick && deep && ladder && !sorted) return std::max<int64_t>(kSortedMinPairs, std::max<int64_t>(t->ladder_min_pairs, kLadderMinPairs));
    // (... and a deep canopy with neither leaves the predicated kernel, which the walk family beats up to ~4e5 pairs on every such tree
    // measured -- 3e5 leaves at depth 311: 0.047 against 0.034 ms at 2^17 pairs, 1e5 leaves at depth 423: 0.067 against 0.026 --;
    // from 524288 pairs the handle's timed choice between it and the tile-sorted walk kernel applies)
    if (walk_is_quick && deep && !ladder && !sorted) return 524288;
    return walk_is_quick && (sorted || ladder || deep) ? kSortedMinPairs : kCanopyMinPairs;
}

static inline bool mrca_ranks_ready(const st_tree *t)
{
    return t->strategy == ST_STRATEGY_CANOPY && t->mrca_ranks && t->d_rec_r && t->d_rmq64;
}

// Which kernel large distance batches of a deep tree get is decided when the tree is created (host_tune.h): the
// candidates -- tile-sorted canopy kernel, predicated canopy kernel, tile-sorted walk kernel -- are timed on a
// sample of random leaf pairs and the handle's defaults (tile_sort, prefer_walk_sorted, ladder_scalar) follow the
// fastest; nothing else separates them reliably (ml.tree: 2.2e10 / 5.7e9 / 1.7e10 pairs/s in that order, a 1e6-leaf
// tree of depth 252: 5.2e9 / 1.1e10 / 6.8e9, a 1e5-leaf tree of depth 423: 3.8e9 / 5.7e9 / 1.1e10;
// profiles/kernel_choice_r03.log).  Batches of 524288 pairs and more (below, the canopy kernels' finer tiles are the
// right size anyway).
static inline bool walk_sorted_ready(const st_tree *t);
constexpr int64_t kWalkSortedMinPairs = 262144;
constexpr int64_t kProbeMinPairs = (int64_t)1 << 22;      // smallest batch the batch probe looks at (host_launch.h): its 8-10 us are 5 % of such a batch on nj.tree
static inline int64_t walk_sorted_min_pairs(const st_tree *t) { return t->walk_sort_min > 0 ? t->walk_sort_min : kWalkSortedMinPairs; }
static inline bool prefers_walk_sorted(const st_tree *t, int64_t n, bool want_dist)
{
    if (want_dist && ladder_applies(t, n)) return false;      // (large batches of a handle that measured the ladder kernel fastest)
    if (t->strategy != ST_STRATEGY_CANOPY || !t->prefer_walk_sorted || !want_dist || n < std::max<int64_t>(walk_sorted_min_pairs(t), 524288)) return false;
    return t->walk_ladder && t->d_crown_ladder && t->walk_crown && walk_sorted_ready(t);
}

// The rule used when the candidates are not timed (SUCHTREE_AMD_AUTOTUNE=0): deep-canopy trees whose canopy image
// leaves the tile-sorted canopy kernel only small tiles (nj.tree: 9111 canopy nodes = 146 KiB, 1024-pair tiles) go to
// the tile-sorted walk kernel once its crown ladder exists (a crown of <= 5120 nodes, 4096-pair tiles).
static inline bool walk_sorted_by_rule(const st_tree *t)
{
    const int q = sorted_q(t);
    return t->strategy == ST_STRATEGY_CANOPY && t->tile_sort && q > 0 && q < 4 && t->d_crown_ladder && walk_sorted_ready(t);
}

// In lineage-sum mode the tile-sorted canopy kernel reads every pair once (key phase; shared-portal pairs, rare,
// a second time) and all its stores are coalesced: it may work on the host path's pinned slots directly.
static inline bool sorted_zero_copy(const st_tree *t) { return sorted_shape(t).sums; }

static inline bool wants_device_stage(const st_tree *t, int64_t m)
{
    if (t->strategy != ST_STRATEGY_CANOPY || !t->tile_sort || sorted_q(t) <= 0) return false;
    if (ladder_applies(t, m)) return false;                 // (the scalar ladder kernel reads every pair once and stores coalesced)
    if (prefers_walk_sorted(t, m, true)) return false;      // (and so does this one)
    if (sorted_zero_copy(t)) return false;
    return m >= canopy_min_pairs(t);
}

// Smallest batch the tile-sorted walk kernel takes: below it k_walk's finer grain wins.  With tiles by batch
// size (batch_tile_q), us per call on ml.tree, k_walk / k_walk_sorted: 131072 pairs 19.7 / 22.8, 262144 31.0 /
// 25.2, 524288 62.7 / 37.5, 1048576 112 / 62; nj.tree 17.8 / 21.2, 26.7 / 23.2, 54.7 / 35.4, 100 / 62
// (profiles/midsize_r03.log).  Largest tiles: 4096 pairs on trees with canopy tables, 2048 on trees only the walk
// family serves (1e6-leaf depth-338 tree, 1e7 / 4e7 pairs: 5.93e9 / 6.43e9 against 5.76e9 / 5.90e9 with 4096;
// ml.tree: 1.28e10 / 1.34e10 against 1.32e10 / 1.41e10).  (kWalkSortedMinPairs = 262144, above.)

static inline bool walk_sorted_ready(const st_tree *t)
{
    return t->walk_sort && t->tree_rmq && t->d_tree_rmq && t->lineage_sums && t->d_lineage && t->d_lineage_node_rec &&
           t->lineage_lens && t->d_lineage_len;
}

}  // namespace st
