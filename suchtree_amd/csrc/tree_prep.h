// tree_prep.h -- host-side preparation of the device tables (plain C++17).
//
// Input is the reference's flat tree (parent:int32[N], distance:float32[N];
// /root/reference/SuchTree/MuchTree.pyx:55-60,182-216).  Output is everything
// the gfx950 kernels read:
//
//   nodes    {parent,dist} 8-byte table ........ walk kernel (single steps)
//   depth    edges to root per node ............ walk kernel (depth cut)
//   stride   three edges + third ancestor ...... walk kernel (three levels per gather)
//   canopy   the top of the tree, BFS-numbered so that parent index < child
//            index; staged into LDS by every workgroup .... canopy kernel
//   records  one fixed-stride "understory" record per node: which canopy
//            node its lineage enters (portal), the float32 running sum of its
//            own lineage below the canopy (pbot) and that lineage's branch
//            lengths / node ids ............................ canopy kernel
//
// No GPU calls in here: the "not gpu" tests exercise this file on the CPU.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ST_HD __host__ __device__ __forceinline__
#else
#define ST_HD inline
#endif

namespace st {

struct Node8 {
    int32_t parent;
    float dist;
};
static_assert(sizeof(Node8) == 8, "Node8 must be 8 bytes");

// Stride-3 image of the whole tree for the walk family: the branch lengths of three
// consecutive edges of a node's lineage and its third ancestor, so that a walk of k edges is
// k/3 dependent 16-byte gathers instead of k (single steps use Node8::parent).  Beyond the
// root the ancestor is clamped to the root and the lengths are 0 (never added: walks know
// their edge counts).
struct Stride3 {
    float d0, d1, d2;   // dist[x], dist[parent(x)], dist[parent(parent(x))]
    int32_t p3;         // third ancestor (clamped to the root)
};
static_assert(sizeof(Stride3) == 16, "Stride3 must be 16 bytes");

struct CanopyEntry {
    float dist;        // branch length above this canopy node
    uint32_t link;     // bits 0..15: canopy index of its parent (root: 0); bits 16..31: depth (edges to the root)
};
constexpr uint32_t kCanopyParentMask = 0xFFFFu;
static_assert(sizeof(CanopyEntry) == 8, "CanopyEntry must be 8 bytes");

// Ladder form of the canopy, for deep trees (hundreds of levels): one 16-byte entry per canopy
// node carries the branch lengths of THREE consecutive edges of its lineage -- its own, its
// parent's and its grandparent's -- and where its third ancestor's entry is, so a climb of k
// edges costs k/3 LDS reads and k float adds (in lineage order, exactly as before) instead of
// k reads.  Single steps (the lock-step search of trees without the sparse table) take the
// parent from the 8-byte CanopyEntry table in global memory.
// The third ancestor is kept as the PLACE of its entry -- the byte offset in the image (index * 16); a kernel that
// stages the image into LDS adds the image's LDS address (pair_math.h: PtrLadder / device_common.h: LdsLadder) -- so a
// climb's next read takes the link as it comes.  An ancestor above the root is kLadderAbove: negative as int32, below
// every place.  Canopy images number parents before children, so along one lineage places grow with depth and a
// climb towards a known ancestor m runs "while link >= place(m)" (signed): the climb loops of the ladder kernels are
// bound by VALU issue (ml.tree: 94 % busy at 7 instructions per round -- shift, mask, count, compare and the three
// adds; profiles/counters_ml_ladder_r04.txt) and this leaves the compare and the adds.
struct alignas(16) LadderEntry {
    float d0, d1, d2;   // dist[v], dist[parent(v)], dist[parent(parent(v))] (0 above the root)
    uint32_t link;      // place of the third ancestor's entry, or kLadderAbove
};
constexpr uint32_t kLadderAbove = 0x80000000u;
static_assert(sizeof(LadderEntry) == 16, "LadderEntry must be 16 bytes");

// Understory record of a node, cap = R/8 - 1 chain slots, kept in three tables so that
// each side of a pair touches the fewest bytes (all indexed by record slot):
//   rec_a  8 bytes  : word0 = portal (canopy index, bits 0..15) | chain length << 16,
//                     pbot (float32 running sum of the chain from 0)
//   rec_b  R/2 bytes: word0 again, then cap float32 branch lengths, the node's own first
//   rec_i  R/2 bytes: pbot again, then the int32 node ids of the same chain in the LAST slots of the cap
//                     (slot cap - 1 = the portal's child, slot cap - nb = the node itself): two chains under one
//                     portal are compared slot by slot from the end, with indices known at compile time
// A pair reads rec_a[a] (8 B of a table small enough to be partly L2 resident) and
// rec_b[b]; rec_i is only needed when both lineages share a portal.  R ("record_bytes")
// is kept as the name of the geometry: rec_b and rec_i have stride R/2.
constexpr int kMaxCanopyNodes = 16384;   // 16384 * 8 B = 128 KiB of the 160 KiB LDS
constexpr int kMaxRecordBytes = 512;
constexpr int kLongRecordBytes = 1024;   // 127-slot chains, read through a pointer by the scalar ladder kernel alone (TreeTables::max_record_bytes)
constexpr int kMinRecordBytes = 16;
constexpr int64_t kMaxTreeRmqBytes = (int64_t)64 << 20;   // whole-tree sparse table: trees of up to ~450k nodes
constexpr int64_t kMaxTreeRmqBytesWalkOnly = (int64_t)4 << 30;   // ... up to ~20M nodes when the walk family is all a tree has
inline int record_cap_for(int rec_bytes) { return rec_bytes / 8 - 1; }

struct TreeTables {
    int64_t n = 0;
    int64_t n_leaves = 0;
    int32_t root = -1;
    int32_t tree_depth = 0;             // nodes on the longest leaf->root path
    bool parity_layout = false;         // leaves are exactly the even ids
    bool inorder_ids = false;           // strictly binary and ids = in-order positions (every tree of the reference)
    std::vector<Node8> nodes;           // [n]
    std::vector<int32_t> depth;         // [n] edges to root
    std::vector<Stride3> stride;        // [n] stride-3 image (walk family)
    // Whole-tree sparse table for the walk family's meeting node (in-order ids only, and only
    // while it stays small: kMaxTreeRmqBytes): tree_rmq[k * n + i] = depth << 32 | id of the
    // shallowest node among ids [i, i + 2^k).  The MRCA of a and b is the shallowest node
    // whose id lies between theirs: two 8-byte reads instead of a climb.
    std::vector<uint64_t> tree_rmq;     // [tree_rmq_levels * n] or empty
    int32_t tree_rmq_levels = 0;
    std::vector<int32_t> bfs_order;     // [n] scratch: parents before children
    std::vector<int32_t> height;        // [n] scratch: nodes down to the deepest leaf (leaf = 1)
    // canopy family (empty when the tree does not admit it)
    bool has_canopy = false;
    int32_t canopy_nodes = 0;
    int32_t understory_max = 0;         // longest chain below the canopy
    int32_t record_bytes = 0;
    int32_t record_cap = 0;             // chain slots per record
    std::vector<CanopyEntry> canopy;    // [canopy_nodes]
    std::vector<int32_t> canopy_id;     // [canopy_nodes] canopy index -> node id
    std::vector<LadderEntry> ladder;    // [canopy_nodes] ladder form of the same canopy (see LadderEntry)
    std::vector<uint16_t> canopy_depth; // [canopy_nodes] edges to the root
    // Meeting node of two canopy nodes in O(1): node ids are in-order positions of a strictly
    // binary tree, so the MRCA of u and v is THE minimum-depth node among those whose id lies
    // between theirs -- and it is a canopy node (the canopy is closed under "parent of").
    // canopy_pos[c] = rank of canopy node c when the canopy is sorted by node id;
    // canopy_rmq[k * canopy_nodes + i] = (depth << 16 | canopy index) of the shallowest node
    // among ranks [i, i + 2^k): a sparse table, one 4-byte read per half of a query.
    std::vector<uint16_t> canopy_pos;   // [canopy_nodes]
    std::vector<uint32_t> canopy_rmq;   // [rmq_levels * canopy_nodes]
    int32_t rmq_levels = 0;
    std::vector<uint8_t> rec_a;         // [n * 8], slot order
    // Four-byte form of the a side (prepare_leaf_blocks; leaves-first layout): rec_a4[slot] = pbot
    // alone, the portal from a small table kept in LDS -- leaf slots are in id order, the leaves
    // below one portal are consecutive, so leaf_block_portal[slot >> leaf_block_shift] names the
    // portal of a whole aligned block of leaf slots (0xFFFF: the block straddles two portals; such
    // leaves, and all internal nodes, read the 8-byte rec_a entry instead).  Built only when at
    // least 99 % of the leaves sit in uniform blocks (balanced and near-balanced trees): the a side
    // of a pair then gathers 4 bytes from a table half the size -- more of it stays in L2.
    std::vector<float> rec_a4;                 // [n] slot order, or empty
    std::vector<uint16_t> leaf_block_portal;   // [ceil(n_leaves >> leaf_block_shift)] or empty
    int32_t leaf_block_shift = 0;
    // Cherry records (prepare_cherries; needs the block table): two sibling leaves share every edge of their understory
    // chains but their own, so ONE record of rec_b's size serves both -- {length of the first leaf, length of the
    // second, chain slots 1 .. cap-1 of either} -- and the b side of a leaf pair gathers from a table HALF the size of
    // rec_b (the fabric serves random sectors of a smaller table faster, and more of it stays in L2).  rec_c[c] is the
    // record of leaf slots 2c and 2c + 1; a block of the table above whose leaves all sit in such pairs has bit 15
    // (kLeafBlockCherries) set beside its portal; other leaves, internal nodes and the shared-portal case read rec_b.
    // Built when at least 99 % of the leaves sit in marked blocks (balanced and near-balanced trees).
    std::vector<uint8_t> rec_c;                // [ceil(n_leaves / 2) * record_bytes/2] or empty
    std::vector<uint8_t> rec_b;         // [n * record_bytes/2]
    std::vector<uint8_t> rec_i;         // [n * record_bytes/2], or empty: left out under a table budget (pairs that share a
                                        // portal are then walked on the tree itself: kernels_canopy.h::same_portal_by_walk)
    // Table budget (host_upload.h): bytes the record tables rec_a + rec_b (+ rec_i) may take; 0 = no limit.
    // prepare_canopy refuses the canopy family when rec_a + rec_b alone exceed it and leaves rec_i out when all three do.
    // (< 0: geometry only -- prepare_canopy says whether the tree would admit the family and builds nothing.)
    int64_t record_budget_bytes = 0;
    int32_t max_record_bytes = kMaxRecordBytes;   // prepare_canopy refuses trees that need longer records (kLongRecordBytes: opt-in, no id chains)
    // Lineage sums (deep canopies with a sparse table; prepare_lineage_sums): the a side of a
    // pair is a sum that STARTS at a -- d = 0; d += dist[n] for n = a, parent(a), ... (pyx:934-936)
    // -- so every prefix of it can be tabulated per node, bit for bit: lineage_sum[off(x) + k] =
    // the reference's accumulator after the first k edges of x's lineage (k = 0 .. depth(x)).
    // With the meeting node known from the sparse table the whole a side of a pair is one 4-byte
    // read; only b's edges, which continue a's sum, are still added one by one.
    // rec_p[slot] = {rank of x's portal (canopy_pos) | depth(x) << 16,
    //               off(x) | (16-byte chunks of x's rec_b entry that hold its chain, minus one) << 28}:
    // all the key phase of the deep kernel reads of either node of a pair -- the meeting node
    // comes from the two ranks, a's edge count and b's from the depths.
    // canopy_rmq64 = canopy_rmq with entries depth << 32 | NODE ID of the shallowest node (the
    // MRCA id of a pair falls out of the same two reads that find its meeting depth).
    // Every node's block of the table holds depth + 1 entries rounded up to a multiple of sixteen, so
    // that blocks start on 64-byte boundaries: a stream of lineage lengths touches whole sectors.
    // Lineage lengths, same blocks and offsets: lineage_len[off(x) + k] = branch length of the k-th
    // node of x's lineage (k = 0: x itself; k < depth(x)), i.e. the operands of the reference's b-side
    // loop (pyx:939-942) laid out contiguously.  With the meeting node known, b's side of a pair is
    // k_b consecutive floats added in order -- independent 16-byte loads, four per 64-byte sector --
    // instead of k_b / 3 dependent gathers through the stride-3 image.
    std::vector<float> lineage_sum;     // [sum over nodes of round_up16(depth + 1)] or empty
    std::vector<float> lineage_len;     // same shape, or empty
    std::vector<uint8_t> rec_p;         // [n * 8], slot order, or empty
    std::vector<uint64_t> canopy_rmq64; // [rmq_levels * canopy_nodes] or empty
    // rec_r[slot] = rank of x's portal | depth(x) << 16 (the first word of rec_p, on its own: 4
    // bytes per node).  With canopy_rmq64 it answers MRCA-only queries without touching the
    // understory records or LDS: two 4-byte reads and two table entries per pair
    // (prepare_rank_table; in-order ids).
    std::vector<uint32_t> rec_r;        // [n], slot order, or empty
    // Trees that only the walk family serves (prepare_walk_lineage): the same lineage sums,
    // offsets by node id (no records there to carry them).
    std::vector<uint32_t> lineage_node_off;   // [n] or empty
    // Crown of the walk family (prepare_walk_crown): crown(H) = { x : height(x) > H }, closed under
    // "parent of"; portal(x) = the lowest node of x's lineage (x included) that is in the crown;
    // nb(x) = nodes of the lineage below the portal (<= H <= 255).  A node's lineage lengths are
    // its own first nb entries followed by ITS PORTAL'S lineage lengths -- the same floats as the
    // tail of its own block, but in a block that every node below that portal shares, so that the
    // long part of every b-side stream falls on a hot set of a few MB (L2 / Infinity Cache) instead
    // of a table of hundreds of MB.  H is the smallest height whose crown blocks fit crown_hot_bytes.
    // lineage_node_rec[4 x + 0..3] = {depth, offset of x's block, offset of its portal's block,
    // nb | rank of the portal << 8}: all the walk family reads of either node, one 16-byte gather.
    // crown_rmq (in-order ids): sparse table over the crown sorted by node id, entries depth << 32 |
    // node id; rank = position in that order.  Two nodes with different portals meet in the crown:
    // their MRCA is the shallowest crown node between the two ranks (a table of a few hundred KB
    // instead of the whole-tree table's hundreds of MB); equal portals use the whole-tree form.
    std::vector<uint32_t> lineage_node_rec;   // [4 * n] or empty
    std::vector<uint64_t> crown_rmq;          // [crown_levels * crown_nodes] or empty
    // Ladder form of the crown, indexed by RANK (ladder entries only name parents and third ancestors,
    // so any numbering serves): when the crown is small enough for LDS (prepare_walk_crown's
    // max_ladder_nodes) the tile-sorted walk kernel climbs the crown part of b's side there -- three
    // edges per 16-byte LDS read, no cache line at all -- and streams only the nodes below the portal.
    std::vector<LadderEntry> crown_ladder;    // [crown_nodes] or empty
    int32_t crown_nodes = 0, crown_levels = 0, crown_height = 0;
    int64_t crown_hot_bytes = 0;              // lineage-length bytes of the crown's blocks
};

// Record slot of node id x.  With the parity layout leaf records come first
// so that leaf-pair queries touch a dense half of the table.
ST_HD int64_t record_slot(int64_t x, bool parity, int64_t n_leaves) {
    return parity ? ((x & 1) ? n_leaves + (x >> 1) : (x >> 1)) : x;
}
// ... and back
ST_HD int64_t record_node(int64_t slot, bool parity, int64_t n_leaves) {
    return parity ? (slot < n_leaves ? 2 * slot : 2 * (slot - n_leaves) + 1) : slot;
}

// (Re)builds the whole-tree sparse table if ids are in-order and it stays within max_bytes
// (prepare_basic calls it with kMaxTreeRmqBytes).  Needs depth and inorder_ids.
bool build_tree_rmq(TreeTables &T, int64_t max_bytes);

// Validates the parent array and fills nodes/depth; returns false with `err`
// set when it is not a single rooted tree.
bool prepare_basic(const int32_t *parent, const float *distance, int64_t n,
                   TreeTables &T, std::string &err);

// rec_a4 + leaf_block_portal (see TreeTables); false (tables left empty) when the tree has no canopy,
// no leaves-first layout, or too few leaves in portal-uniform blocks of at most max_blocks blocks.
bool prepare_leaf_blocks(TreeTables &T, int max_blocks = 8192);

// rec_c and the cherry bits of leaf_block_portal (see TreeTables); false (rec_c left empty, no bit set) otherwise.
constexpr uint16_t kLeafBlockCherries = 0x8000u, kLeafBlockPortalMask = 0x3FFFu, kLeafBlockMixed = 0xFFFFu;
static_assert(kMaxCanopyNodes <= (int)kLeafBlockPortalMask + 1,
              "a block-table entry keeps its portal in 14 bits (k_canopy_ilp masks every entry with kLeafBlockPortalMask)");
bool prepare_cherries(TreeTables &T);

// rec_r and canopy_rmq64 (see TreeTables) for trees with a canopy and in-order ids; false otherwise.
bool prepare_rank_table(TreeTables &T);

// Lineage sums for a tree without canopy tables: lineage_sum + lineage_node_off, when the table
// has at most max_entries (< 2^32) entries.  Any node numbering.  Built on several threads (one
// root-ward walk per node).
// with_lens: also the lineage-length table (same size again).
bool prepare_walk_lineage(TreeTables &T, int64_t max_entries, bool with_lens = true);

// Lineage sums of every node (see TreeTables), for trees with a canopy and a sparse table and
// at most max_entries table entries; returns false (tables left empty) otherwise.
bool prepare_lineage_sums(TreeTables &T, int64_t max_entries, bool with_lens = true);

// node_rec + crown tables for the walk family (see TreeTables); needs lineage_node_off (either
// prepare_*lineage* function fills it).  hot_bytes: budget of the crown's lineage-length blocks.
// max_ladder_nodes > 0: when some H <= 255 leaves at most that many nodes in the crown, that H is taken
// (instead of the one from hot_bytes) and crown_ladder is built as well.
bool prepare_walk_crown(TreeTables &T, int64_t hot_bytes, int max_ladder_nodes = 0);

// Entries of a node's block in the lineage tables.
ST_HD int64_t lineage_block(int32_t depth) { return ((int64_t)depth + 1 + 15) & ~(int64_t)15; }

// Chooses the canopy and builds canopy + records.  Returns false (without
// error) when the tree does not admit the canopy family (lineages below any
// 16k-node canopy longer than a record can hold).
// max_canopy_nodes <= 0: the LDS limit (kMaxCanopyNodes); smaller values trade a longer
// understory (bigger records) for a smaller LDS image, i.e. more workgroups per CU.
bool prepare_canopy(const int32_t *parent, const float *distance, TreeTables &T, int max_canopy_nodes = 0);
constexpr int kShallowCanopyDepth = 100;   // canopies deeper than this (edges) are "deep": tile-sorted kernels, tuned at creation

}  // namespace st
