// st_tree.h -- the tree handle (struct st_tree): device tables, geometry, tuning switches.  Shared by every
// translation unit of libsuchtree_hip.so (the launch units read it, suchtree_hip.hip owns it).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <vector>

#include "../../include/suchtree_hip.h"
#include "device_common.h"
#include "tree_prep.h"

struct DevicePipe;      // host_tree.h: the device's staging pipe

using st::CanopyEntry;
using st::Fault;
using st::LadderEntry;
using st::Node8;
using st::Stride3;

constexpr unsigned kWorkSlots = 64;      // work-counter slots of a handle (k_canopy_ladder); a slot's reuse waits for the launch that drew from it last (work_done)

struct st_tree {
    int device = 0;
    int strategy = ST_STRATEGY_WALK;       // family in use
    bool has_canopy = false;
    int n_cu = 256;           // CUs the launches are sized for (option "reserve_cus": the device's count minus what is left to others)
    int n_cu_device = 256;
    st_tree_info info{};
    // device tables
    Node8 *d_nodes = nullptr;
    int32_t *d_depth = nullptr;
    Stride3 *d_stride = nullptr;
    uint64_t *d_tree_rmq = nullptr;   // whole-tree sparse table (in-order ids, small trees), else NULL
    CanopyEntry *d_canopy = nullptr;
    int32_t *d_canopy_id = nullptr;
    uint8_t *d_rec_a = nullptr, *d_rec_b = nullptr, *d_rec_i = nullptr;
    float *d_rec_a4 = nullptr;            // four-byte form of the a side (balanced-like trees), else NULL
    uint16_t *d_leaf_blocks = nullptr;
    uint8_t *d_rec_c = nullptr;           // cherry records (one per pair of sibling leaves; tree_prep.h), else NULL
    int cherries = 1;         // tuning: 0 = the predicated kernel reads every b record from rec_b even where cherry records exist
    int32_t leaf_block_shift = 0, leaf_block_count = 0;
    int rec_a4 = 1;           // tuning: 0 = the predicated canopy kernel reads the 8-byte rec_a entries even when the four-byte form exists
    uint8_t *d_rec_p = nullptr;       // lineage sums (deep canopies with a sparse table), else NULL
    uint64_t *d_rmq64 = nullptr;
    uint16_t *d_rec_r = nullptr;      // rank of every node's portal: MRCA-only queries (in-order ids), else NULL
    float *d_lineage = nullptr;
    float *d_lineage_len = nullptr;           // lineage lengths (same blocks as d_lineage), else NULL
    uint32_t *d_lineage_node_rec = nullptr;   // {depth, lineage offset, portal's lineage offset, nb | portal rank << 8} by node id
    uint64_t *d_crown_rmq = nullptr;          // sparse table over the walk family's crown (in-order ids), else NULL
    LadderEntry *d_crown_ladder = nullptr;    // ladder form of the crown by rank (crowns that fit LDS), else NULL
    int walk_ladder = 1;      // tuning: 0 = k_walk_sorted streams the crown part of b's side from the portal's block instead of climbing it in LDS
    int32_t crown_nodes = 0;
    // two fault words: the device-pointer entry points are not serialised against anything,
    // so the host path keeps its own (reset at the start of every host call, read under the
    // device pipe's mutex) and is never confused by a caller who skipped st_fault_check
    Fault *d_fault = nullptr;        // st_distances_device / st_triangle_device / st_fault_check
    Fault *d_fault_host = nullptr;   // st_*_host
    bool host_fault_dirty = false;   // a host call ended before reading its fault word back: re-arm it first
    // canopy geometry
    int32_t canopy_nodes = 0, rec_bytes = 0, rec_cap = 0, parity = 0;
    int64_t n_nodes = 0, n_leaves = 0;
    int tile_sort = 0;        // tuning: 1 = tile-sorted kernel over the ladder form of the canopy (default for deep canopies)
    int ladder_scalar = 0;    // tuning: 1 = distance batches of >= ladder_min_pairs pairs on records of 128 bytes and more go to k_canopy_ladder: the scalar kernel over the ladder image, meeting nodes from the sparse table (set when the tree is created: timed)
    int ladder_dynamic = 1;   // tuning: 0 = the scalar ladder kernel deals its pairs statically whatever the batch size
    unsigned long long *d_work = nullptr;          // kWorkSlots x 64 counters (eight used per launch, 64 bytes apart)
    mutable std::atomic<unsigned> work_next{0};
    hipEvent_t work_done[kWorkSlots] = {};         // recorded behind the launch that used slot k: the next user's memset waits for it, whatever its stream
    int *d_choice = nullptr;                       // kWorkSlots words that receive the batch probe's verdicts (pair_math.h: probe_says_walk; st_probe_last_choice), or NULL
    mutable std::atomic<unsigned> choice_next{0};
    int batch_probe = 1;      // tuning: 0 = large explicit batches of a deep tree always go to the kernel the handle chose when it was created
    int64_t ladder_min_pairs = 0;   // smallest batch of that kernel; 0 = kLadderMinPairs (set when the tree is created: timed at two batch sizes)
    int prefer_walk_sorted = 0;   // large distance batches of a canopy-strategy tree go to k_walk_sorted (set when the tree is created: timed, or by rule)
    int wire48 = 1;           // tuning: 0 = host-path ids always cross the link as int32 (8 bytes per pair), also on trees of fewer than 2^24 nodes
    int wire24 = 1;           // tuning: 0 = host-path MRCA ids always come back as int32 (8 bytes per pair with the float32 distance), also on trees of fewer than 2^24 nodes
    int sort_tile = 0;        // tuning: tile of both tile-sorted kernels in units of 1024 pairs, 1 / 2 / 4 (when it fits LDS); 0 = by batch size
    int64_t walk_sort_min = 0;   // tuning: smallest batch of the tile-sorted walk kernel; 0 = kWalkSortedMinPairs
    int tree_rmq = 1;         // tuning: 0 = the walk family searches the meeting node by climbing even when the whole-tree sparse table exists
    int mrca_ranks = 1;       // tuning: 0 = MRCA-only requests go through the distance kernels
    int walk_crown = 1;       // tuning: 0 = the walk family streams b's side from b's own block alone and finds meeting nodes in the whole-tree table
    int walk_sort = 1;        // tuning: 0 = the walk family never uses its tile-sorted kernel (k_walk_sorted)
    int lineage_lens = 1;     // tuning: 0 = the walk family climbs b's lineage through the stride-3 image even when the lineage-length table exists
    int lineage_sums = 1;     // tuning: 0 = the tile-sorted kernel climbs a's canopy lineage even when the lineage-sum table exists
    int64_t ladder_sums_max_pairs = 0;   // largest batch of that form; 0 = every batch (set with ladder_sums when the tree is created)
    int ladder_sums = 0;      // 1 = the scalar ladder kernel reads a's whole side from the lineage sums too (kernels_canopy.h: ladder_pair_sums)
    LadderEntry *d_ladder = nullptr;
    uint16_t *d_cdepth = nullptr;
    uint16_t *d_cpos = nullptr;     // sparse table for the meeting node (in-order ids only)
    uint32_t *d_rmq = nullptr;
    int canopy_depth = 0;     // deepest canopy node (edges)
    int small_batch_path = 1; // tuning: batches <= kMailboxPairs go through the pinned mailbox
    int measure = 0;          // measurement switches of the host path (host_path.h::run_pipe): 1 trace, 2 skip the CPU passes, 4 skip the GPU side; calls with 2 / 4 return ST_ERR_MEASURE_ONLY
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
