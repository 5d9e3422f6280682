/*
 * suchtree_hip.h -- C ABI of libsuchtree_hip.so, the MI355X (gfx950) bulk
 * patristic-distance / MRCA engine behind suchtree_amd.SuchTree.
 *
 * The reference (ryneches/SuchTree) has no FFI layer for this path: the
 * boundary is its Cython class surface and the two cdef methods behind it.
 * Each entry point below names the reference interface it stands in for
 * (paths relative to /root/reference).  Signatures are plain C: pointers and
 * sizes only, no torch / numpy types.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns ST_OK (0) or an ST_ERR_* code; the message for
 *     the calling thread's last failure is st_last_error().
 *   - node ids are the reference's ids: positions in the in-order traversal
 *     of the strictly binary tree (SuchTree/MuchTree.pyx:171-180).
 *   - pairs are int64 (n,2) views given as a base pointer and two ELEMENT
 *     strides (the reference takes any `long[:,:]` memoryview,
 *     MuchTree.pyx:913); C order is stride0=2, stride1=1.
 *   - distances come back as float64 holding the float32 value the
 *     reference accumulates (MuchTree.pyx:922,943); MRCA ids as int32.
 *   - "_host" entry points take host buffers and do the transfers; "_device"
 *     entry points take device buffers on the tree's GPU, enqueue on the
 *     caller's hipStream_t (passed as void*, NULL = default stream) and do
 *     not synchronise.
 */
#ifndef SUCHTREE_HIP_H
#define SUCHTREE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * ABI version: bumped whenever an exported struct grows, or an export, option or constant goes away.  A caller built against
 * an older header must not be handed a larger st_tree_info: compare ST_API_VERSION with st_api_version() at load (the ctypes
 * binding does) and use st_tree_info_get_sized, which writes at most the bytes the caller says it has.
 *   6 (round 6): st_api_version, st_tree_info_get_sized, st_probe_last_choice, option "ladder_sums" added; st_tree_info.reserved0
 *                is now ladder_sums, ladder_sums_max_pairs appended (8 bytes); option "tile_sort" selects nothing on records of 128 bytes and more (kernel forms removed).
 *   5 (round 5): st_tree_info grew by 8 bytes (b_table_bytes_per_leaf, reserved0); st_host_alloc / st_host_free,
 *                ST_KERNEL_CANOPY_SCALAR, the options pairs_per_lane and ladder_dynamic = 2 removed.
 */
#define ST_API_VERSION 6
int st_api_version(void);

#define ST_OK          0
#define ST_ERR_ARG     1   /* bad argument (NULL, negative size, bad strategy) */
#define ST_ERR_HIP     2   /* a HIP runtime call failed / no usable GPU */
#define ST_ERR_BOUNDS  3   /* a node id in `pairs` is outside [0, n_nodes) */
#define ST_ERR_NOMEM   4   /* host allocation failed */
#define ST_ERR_TREE    5   /* parent array is not a single rooted binary tree */
#define ST_ERR_MEASURE_ONLY 6   /* the handle's "measure" option made this host-path call skip work: timing only, results invalid */

/* kernel families (st_tree_create `strategy`, st_tree_info.strategy) */
#define ST_STRATEGY_AUTO    0  /* canopy when the tree admits it, else walk */
#define ST_STRATEGY_WALK    1  /* pointer chase over the {parent,dist} table */
#define ST_STRATEGY_CANOPY  2  /* top of tree in LDS + per-node understory records */

typedef struct st_tree st_tree;   /* opaque: device-resident tree */

typedef struct st_tree_info {
    int64_t n_nodes;
    int64_t n_leaves;
    int32_t root;
    int32_t depth;            /* nodes on the longest leaf->root path (MuchTree.pyx:218-225) */
    int32_t device;
    int32_t strategy;         /* ST_STRATEGY_WALK or ST_STRATEGY_CANOPY actually in use */
    int32_t canopy_nodes;     /* nodes staged in LDS (0 for walk) */
    int32_t understory_max;   /* longest chain below the canopy, in nodes */
    int32_t record_bytes;     /* stride of one understory record */
    int32_t n_devices;        /* GPUs holding a replica (1 unless st_tree_create_multi) */
    int64_t device_bytes;     /* HBM held by this tree */
    int64_t lineage_entries;  /* float32 entries of the lineage-sum table (deep canopies, in-order ids), else 0 */
    int32_t big_batch_kernel; /* kernel of large distance batches: ST_KERNEL_* below */
    int32_t tuned;            /* 1 = that kernel was chosen by timing the candidates when the tree was created, 2 = read
                                 from the record an earlier handle of the same tree on the same device left, 0 = by rule */
    int32_t host_wire_bytes_in;   /* bytes per pair the st_*_host entry points ship over the link: ids in (6 or 8) ... */
    int32_t host_wire_bytes_out;  /* ... float32 distance + MRCA id back (7 or 8); follow the wire48 / wire24 options */
    int32_t a_side_bytes;     /* canopy family: bytes gathered for the first node of a pair, 4 (rec_a4 + block table) or 8 */
    int32_t dropped_tables;   /* ST_TABLE_* bits: what the table budget left out (slower forms of the kernels take over) */
    int64_t table_budget_bytes;   /* the budget this handle was built under (0 = none) */
    int32_t b_table_bytes_per_leaf;   /* canopy family: bytes per leaf of the table the second node of a leaf pair gathers from:
                                         record_bytes / 2, or record_bytes / 4 where sibling leaves share a cherry record */
    int32_t ladder_sums;      /* (was reserved0 until version 6) 1 = the scalar ladder kernel reads the first node's whole side from the
                                 lineage sums (option "ladder_sums", set by timing when a deep tree is created), 0 = it climbs both sides */
    int64_t ladder_sums_max_pairs;   /* largest batch the joint form takes when ladder_sums is 1; 0 = every batch (ml.tree: 2^20 -- beyond it the
                                        climbing form runs) */
} st_tree_info;

/* st_tree_info.dropped_tables, in the order in which a table budget (st_tree_options.table_budget_bytes, else
 * SUCHTREE_AMD_TABLE_MB) drops them -- every one is an accelerator, results are the same bits without it: */
#define ST_TABLE_LINEAGE_LEN   1   /* lineage lengths + crown tables: the walk family climbs b's side instead of streaming it */
#define ST_TABLE_LINEAGE_SUM   2   /* lineage sums: a's side is climbed, the tile-sorted kernels lose their lineage-sum form */
#define ST_TABLE_TREE_RMQ      4   /* whole-tree sparse table: the walk family finds meeting nodes by climbing */
#define ST_TABLE_REC_I         8   /* id chains of the understory records (as large as the b-side records: judged when the
                                      records are sized): pairs under one portal are walked on the tree */
#define ST_TABLE_REC_A4       16   /* four-byte a side of the predicated kernel and the cherry records of its b side: the 8-byte entries and rec_b serve */
#define ST_TABLE_RANKS        32   /* rank table of MRCA-only requests: they go through the distance kernels */
#define ST_TABLE_CANOPY       64   /* every canopy table: the walk family serves the tree */

/* Creation options (st_tree_create_ex); zero-initialise, then set what is wanted. */
typedef struct st_tree_options {
    int64_t table_budget_bytes;   /* device bytes the tree's tables may take; 0 = SUCHTREE_AMD_TABLE_MB (MiB) if set, else no
                                     limit.  Below the floor (28 bytes per node: parent/distance, depth and the three-level
                                     image the walk kernel needs) the floor is what is uploaded. */
    int64_t reserved[7];
} st_tree_options;

/* st_tree_info.big_batch_kernel */
#define ST_KERNEL_WALK            0   /* k_walk / k_walk_sorted: trees the canopy family does not serve */
#define ST_KERNEL_CANOPY          1   /* predicated canopy kernel (one pair per lane, chain in registers); 2: unused since round 5 */
#define ST_KERNEL_CANOPY_SORTED   3   /* tile-sorted canopy kernel (ladder form of the canopy in LDS; chains of at most seven slots) */
#define ST_KERNEL_WALK_SORTED     4   /* tile-sorted walk kernel on a tree that also has canopy tables */
#define ST_KERNEL_CANOPY_LADDER   5   /* scalar canopy kernel over the ladder form (long records in registers, read once) */

/* Last error message of the calling thread ("" if none). */
const char *st_last_error(void);

/* Number of visible HIP devices. */
int st_device_count(int *count);

/*
 * Upload a tree.  Replaces the reference's in-object `Node* data` filled at
 * SuchTree/MuchTree.pyx:158-216 (parent / distance columns) and the `depth`
 * scan at :218-225.  `parent[root] == -1`; `distance` are the float32 branch
 * lengths exactly as the reference stores them (root entry ignored).
 * All derived tables (depths, canopy, understory records) are built here,
 * once, and stay resident in HBM until st_tree_destroy.
 */
int st_tree_create(const int32_t *parent, const float *distance, int64_t n_nodes,
                   int device, int strategy, st_tree **out);

/*
 * Same tree replicated on several GPUs of one node, driven from ONE process: the reference's
 * user calls T.distances(ids) from a single process (SuchTree/MuchTree.pyx:872-909) and
 * parallelises with a fork pool over contiguous chunks
 * (docs/examples/SuchTree_examples.md:462-497); here the "_host" entry points deal their
 * pipeline chunks round-robin over `devices` (one host thread, one staging pipe and one
 * PCIe link per GPU; st_host_chunk_plan / st_host_chunk_owner state the map).  The tables
 * are built once and uploaded to every listed device.  Device-pointer entry points, the
 * small-batch mailbox and st_quartets_host use devices[0].
 */
int st_tree_create_multi(const int32_t *parent, const float *distance, int64_t n_nodes,
                         const int *devices, int n_devices, int strategy, st_tree **out);

/*
 * The same with creation options: a budget for the device tables.  A tree costs 28 bytes per node on the device (the
 * floor); everything beyond that -- 60 to 1500 times the reference's 20-byte Node -- is accelerator tables, and under
 * a budget they are left out in the order of the ST_TABLE_* bits above until the rest fits; st_tree_info reports
 * device_bytes and dropped_tables.  opts may be NULL (= st_tree_create_multi).  No counterpart in the reference,
 * whose tree is one calloc of n nodes (SuchTree/MuchTree.pyx:114, 160-165).
 */
int st_tree_create_ex(const int32_t *parent, const float *distance, int64_t n_nodes,
                      const int *devices, int n_devices, int strategy, const st_tree_options *opts, st_tree **out);

/* Host-only helper, no GPU needed: what st_tree_create_ex would build for this tree under `table_budget_bytes`
 * (0 = SUCHTREE_AMD_TABLE_MB if set, else no limit): the device bytes, the ST_TABLE_* bits left out and the family. */
int st_host_table_plan(const int32_t *parent, const float *distance, int64_t n_nodes, int strategy,
                       int64_t table_budget_bytes, int64_t *device_bytes, int32_t *dropped_tables, int32_t *family);

/* Devices of a handle, devices[0] first (devices may be NULL to ask for the count only). */
int st_tree_devices(const st_tree *tree, int *devices, int capacity, int *n_devices);

/*
 * How a host batch of n pairs is cut into pipeline chunks and which device of a
 * multi-device handle gets which chunk (no GPU needed): chunk c covers
 * [c * chunk_pairs, min(n, (c+1) * chunk_pairs)) and is computed by devices[c % n_devices].
 */
int st_host_chunk_plan(int64_t n, int n_devices, int64_t *chunk_pairs, int64_t *n_chunks);
int st_host_chunk_owner(int64_t n, int n_devices, int64_t chunk_index, int *device_index,
                        int64_t *first_pair, int64_t *n_pairs);

/* Replaces SuchTree.__dealloc__ (MuchTree.pyx:230-232). */
void st_tree_destroy(st_tree *tree);

/* Fills *info (sizeof(st_tree_info) bytes of THIS header's struct: callers compiled against it only). */
int st_tree_info_get(const st_tree *tree, st_tree_info *info);
/* The same for a caller that states how large ITS st_tree_info is: at most info_bytes bytes are written (fields are only ever
 * appended, so a shorter struct is a prefix); info_bytes < 8 or not a multiple of 4 is ST_ERR_ARG. */
int st_tree_info_get_sized(const st_tree *tree, void *info, int64_t info_bytes);

/*
 * Bulk distances (+ MRCA ids) for host-resident pairs.  Replaces
 * SuchTree._distances (SuchTree/MuchTree.pyx:911-943, which calls _mrca
 * :999-1030) as invoked by distances_bulk (:872-909).
 * out_dist and out_mrca may each be NULL (not both).  On ST_ERR_BOUNDS
 * *bad_id (if non-NULL) receives the id the reference would report
 * (max id if it is >= n_nodes, else the min id; MuchTree.pyx:897-903) and the
 * outputs are unspecified.
 */
int st_distances_host(st_tree *tree, const int64_t *pairs, int64_t n,
                      int64_t stride0, int64_t stride1,
                      double *out_dist, int32_t *out_mrca, int64_t *bad_id);

/* Same for int32 ids (element strides of the int32 view): half the host-side read traffic,
 * and a C-order array is already what is sent over PCIe. */
int st_distances_host_i32(st_tree *tree, const int32_t *pairs, int64_t n,
                          int64_t stride0, int64_t stride1,
                          double *out_dist, int32_t *out_mrca, int64_t *bad_id);

/*
 * Same computation on device-resident buffers, enqueued on `stream`
 * (a hipStream_t; NULL = default stream).  Does not synchronise.  Out-of-range
 * ids never dereference the tree: such pairs produce NaN / -1 and are
 * recorded in the tree's device-path fault word, read back by st_fault_check
 * (the "_host" entry points keep a fault word of their own).
 */
int st_distances_device(st_tree *tree, const int64_t *d_pairs, int64_t n,
                        int64_t stride0, int64_t stride1,
                        double *d_out_dist, int32_t *d_out_mrca, void *stream);

/* Same, writing the distances as the float32 values they are (half the output bytes; the
 * float64 form above holds exactly these values widened). */
int st_distances_device_f32(st_tree *tree, const int64_t *d_pairs, int64_t n,
                            int64_t stride0, int64_t stride1,
                            float *d_out_dist, int32_t *d_out_mrca, void *stream);

/*
 * Wire format of result slices that travel (multi-GPU gather over xGMI, suchtree_amd/sharding.py::run_sharded;
 * the host path ships the same format over PCIe): float32 distances and MRCA ids as 24 bits each -- id i at bytes
 * [3 i, 3 i + 3) of d_out_mrca24, little endian, -1 (an id out of range) as 0xFFFFFF -- 7 bytes per pair instead of
 * the 12 of float64 + int32.  The kernels assemble the packed stream themselves (no packing pass).  Trees of fewer
 * than 2^24 nodes only (ST_ERR_ARG otherwise); d_out_mrca24 must be 4-byte aligned and hold 3 n bytes rounded up to
 * a multiple of 4 (the last dword is written whole).  Either output may be NULL.  No counterpart in the reference,
 * whose only parallel recipe is a fork pool (docs/examples/SuchTree_examples.md:462-497).
 */
int st_distances_device_wire(st_tree *tree, const int64_t *d_pairs, int64_t n,
                             int64_t stride0, int64_t stride1,
                             float *d_out_dist, uint8_t *d_out_mrca24, void *stream);

/* The receiving side: n packed 24-bit ids at d_packed (any byte alignment) -> int32 at d_out_mrca (0xFFFFFF -> -1),
 * enqueued on `stream` of `device`. */
int st_unpack_mrca24_device(int device, const uint8_t *d_packed, int64_t n, int32_t *d_out_mrca, void *stream);

/*
 * Synchronise `stream`, then report and clear the fault word written by
 * earlier st_distances_device calls: ST_OK, or ST_ERR_BOUNDS with *bad_id set
 * as for st_distances_host.
 */
int st_fault_check(st_tree *tree, void *stream, int64_t *bad_id);

/*
 * Diagnostic: which kernel the batch probe gave the handle's most recent probed batch (large device-resident batches of
 * explicit pairs on deep trees decide per batch, on the device: option batch_probe).  Synchronises `stream`.
 * *choice = 0 scalar ladder kernel, 1 tile-sorted walk kernel, -1 no batch of this handle has been probed yet.
 */
int st_probe_last_choice(st_tree *tree, void *stream, int *choice);

/*
 * All-pairs generator: for an id list ids[0..m) (element stride id_stride) computes
 * pair k = (ids[j], ids[i]), k = i(i-1)/2 + j, 0 <= j < i < m, for k in
 * [k_begin, k_begin + k_count); out[k - k_begin] receives the result.  No pair array
 * exists anywhere: the kernel derives (i, j) from k.  Replaces the nested pair loops of
 * SuchLinkedTrees.linked_distances (SuchTree/MuchTree.pyx:2918-2925, ids = a linklist
 * column) and, up to enumeration order, of SuchTree.pairwise_distances (:1106-1114),
 * followed by _distances (:911-943).  The k-range lets callers shard the triangle by
 * equal pair counts across GPUs and stream it in tiles.
 */
int st_triangle_device(st_tree *tree, const int64_t *d_ids, int64_t m, int64_t id_stride,
                       int64_t k_begin, int64_t k_count,
                       double *d_out_dist, int32_t *d_out_mrca, void *stream);
int st_triangle_host(st_tree *tree, const int64_t *ids, int64_t m, int64_t id_stride,
                     int64_t k_begin, int64_t k_count,
                     double *out_dist, int32_t *out_mrca, int64_t *bad_id);

/*
 * Grid generator: for id lists row_ids[0..n_rows) and col_ids[0..n_cols) (contiguous int64)
 * computes element e = r * n_cols + c, the pair (row_ids[r], col_ids[c]), for e in
 * [e_begin, e_begin + e_count); out[e - e_begin] receives the result.  With `symmetric` != 0
 * (same list on both sides) elements below the diagonal take their mirror image's argument
 * order, so the whole range [0, n^2) IS the symmetric matrix of SuchTree.pairwise_distances
 * (SuchTree/MuchTree.pyx:1084-1124: pairs (ids[i], ids[j]), i < j, scattered to [i,j] and
 * [j,i]; zero diagonal), written straight into the caller's (n,n) float64 array -- the
 * reference's Python pair list and its scatter loop (:1106-1122) have no counterpart.
 * A rectangular grid is the distance block of nearest_neighbors (:1069-1072).
 */
int st_grid_host(st_tree *tree, const int64_t *row_ids, int64_t n_rows,
                 const int64_t *col_ids, int64_t n_cols, int symmetric,
                 int64_t e_begin, int64_t e_count,
                 double *out_dist, int32_t *out_mrca, int64_t *bad_id);

/*
 * k nearest candidates of every query: distances query -> cands on the device, then a
 * per-row selection of the k smallest (ties: lower candidate index first), both on the GPU.
 * Replaces the pair list, distances_bulk call and np.argsort of SuchTree.nearest_neighbors
 * (SuchTree/MuchTree.pyx:1069-1082) for many queries at once.  out_index (n_queries, k) holds
 * positions in `cands` (-1 where fewer than k candidates exist), out_dist (n_queries, k) the
 * distances, ascending.  skip_self != 0 ignores candidates equal to the query id (the
 * reference drops a leaf query from its default candidate list, :1058-1062).  1 <= k <= 256.
 */
int st_knn_host(st_tree *tree, const int64_t *queries, int64_t n_queries,
                const int64_t *cands, int64_t n_cands, int k, int skip_self,
                int64_t *out_index, double *out_dist, int64_t *bad_id);

/*
 * Quartet topologies: for each row (a,b,c,d) of the int64 (n,4) view the row re-ordered so
 * that columns (0,1) and (2,3) are the sister pairs.  Replaces
 * SuchTree._quartet_topologies (SuchTree/MuchTree.pyx:1331-1376) as called by
 * quartet_topologies_bulk (:1271-1329).  out_topologies is C-order int64 (n,4).
 */
int st_quartets_host(st_tree *tree, const int64_t *quartets, int64_t n,
                     int64_t stride0, int64_t stride1,
                     int64_t *out_topologies, int64_t *bad_id);

/*
 * Dense graph matrices of SuchLinkedTrees: adjacency A (A[u][v] = A[v][u] = w per edge) and
 * Laplacian L = diag(column sums of A) - A, both n x n float64, C order; either output may
 * be NULL.  Replaces the numpy assembly at the end of SuchLinkedTrees.adjacency / .laplacian
 * (SuchTree/MuchTree.pyx:3110-3145); the edge list (tree edges normalised by the largest
 * edge, link edges at the mean weight, :3113-3129) is prepared by the caller.  No tree
 * handle involved.
 */
int st_graph_matrices_host(int device, int64_t n, int64_t n_edges, const int32_t *u,
                           const int32_t *v, const double *w,
                           double *out_adjacency, double *out_laplacian);

/* Select the kernel family for subsequent calls (tests / benchmarking).
 * ST_ERR_ARG if the tree was built without that family's tables. */
int st_tree_set_strategy(st_tree *tree, int strategy);

/* Tuning knobs (benchmarking / tests).
 * "tile_sort": 1 = on deep canopies whose records hold at most seven chain slots (16- to 64-byte records: small deep trees, a
 * few thousand leaves) every workgroup sorts its tile of pairs by expected climb length so that a wave's lanes finish together
 * (the default where it measured fastest when the tree was created, see "prefer_walk_sorted" below); 0 = pairs in input order.
 * On longer records the option selects nothing since version 6: the tile-sorted canopy kernel's 15- / 31-slot and pointer forms
 * won no cell of profiles/kernel_win_matrix_r06.json against the scalar ladder kernel ("ladder_scalar") and were removed.
 * "tree_rmq": 1 (default) = the walk family takes the meeting node from the whole-tree sparse table
 * where the tree has one (in-order ids; up to 64 MB, more -- within SUCHTREE_AMD_WALK_TABLE_MB -- when
 * the canopy family is not available); 0 = it searches it by climbing both lineages.
 * "mrca_ranks": 1 (default) = MRCA-only requests on trees with in-order ids are answered from a
 * per-node rank table and a sparse table over the canopy (no LDS, no understory records);
 * 0 = they go through the distance kernels.
 * "lineage_sums": 1 (default) = the first node's whole side of a pair comes from a table of per-node
 * lineage sums (one 4-byte read) wherever the tree has that table: in the tile-sorted canopy kernel
 * (deep canopies with in-order ids, table below 1 GiB) and in the walk family (the same trees, and
 * trees only the walk family serves); 0 = that side is climbed as well (this also switches off
 * everything below that builds on the table).
 * "lineage_lens": 1 (default) = the walk family adds the second node's side from the lineage-length
 * table, consecutive floats, instead of climbing the stride-3 image; 0 = it climbs.
 * "walk_crown": 1 (default) = the walk family reads the long upper part of that stream from the
 * block of the node's portal (shared by all nodes below it: a cache-resident hot set) and takes
 * meeting nodes of different portals from the crown's own sparse table; 0 = own block, whole-tree table.
 * "walk_ladder": 1 (default) = where the crown is small enough (<= 8192 nodes) the tile-sorted walk kernel keeps
 * its ladder form in LDS and climbs the upper part of the second node's side there (three edges per 16-byte LDS
 * read) instead of streaming it; 0 = it streams it from the portal's block.
 * "walk_sort": 1 (default) = batches of >= 262144 pairs on trees with the sparse table and both lineage
 * tables run the tile-sorted walk kernel (a wave's 64 pairs have streams of similar length); 0 = k_walk.
 * "prefer_walk_sorted": 1 = distance batches of >= 524288 pairs on a canopy-strategy tree that also has the walk
 * family's tables (deep trees) go to the tile-sorted walk kernel; 0 = they stay with the canopy kernels.  Like
 * "tile_sort" its default is set when the tree is created, on deep trees by timing the candidate
 * kernels on a sample of random leaf pairs (st_tree_info.tuned; SUCHTREE_AMD_AUTOTUNE=0: by a fixed rule).
 * "ladder_scalar": 1 = distance batches of at least "ladder_min_pairs" pairs (0 = 131072) on records of 128 bytes and
 * more are served by the scalar kernel over the ladder form of the canopy (records read once, no sort: large batches);
 * both defaults are set with the three above when a deep tree is created (timed at two or three batch sizes, on
 * uniform random leaf pairs: a caller whose batches are all close relatives -- every pair within a few leaves, short
 * paths -- does better with "ladder_scalar" 0 and "prefer_walk_sorted" 1: the walk family's cost follows the path
 * length; nj.tree, leaves within 8 of each other: 1.4e10 -> 1.9e10 pairs/s, ml.tree 1.7e10 -> 2.0e10).
 * "ladder_sums": 1 = that kernel reads the first node's whole side of a pair from the lineage sums (one 4-byte read in place of
 * its understory entry and its climb in LDS; the meeting node from the 64-bit sparse table by the two portal ranks, the second
 * node's record by the 16-byte chunks that hold chain slots in use) and climbs the second node's canopy edges only; 0 = both
 * sides are climbed.  Needs the lineage sums ("lineage_sums" 1, table built); the default is set by timing when a deep tree is
 * created, possibly for batches up to a size only (nj.tree: 1, +17 %; ml.tree: 1 for batches below 2^20 pairs, where it leads
 * by 8-13 %, 0 above -- there the extra fabric read costs more than its second workgroup's climbs save); setting the option by
 * hand applies it to every batch size.
 * "ladder_dynamic": 1 (default) = on records of 512 bytes and more, batches of 2^22 pairs and more (2^21 on 1 KB
 * records) of that kernel draw their work from per-XCD counters instead of a static deal; 0 = never.
 * "walk_sort_min": smallest batch (pairs) that kernel takes; 0 (default) = 262144.
 * "sort_tile": tile of both tile-sorted kernels in units of 1024 pairs: 1, 2 or 4 (taken when it fits LDS and the
 * kernel's form has that tile), 0 (default) = the largest tile LDS admits, cut finer for batches that would
 * otherwise leave CUs without a tile.
 * "rec_a4": 1 (default) = on trees whose leaves sit in portal-uniform aligned blocks of leaf slots (balanced
 * and near-balanced trees) the predicated canopy kernel gathers 4 bytes for the first node of a pair
 * (its understory sum; the portal comes from a block table in LDS) instead of the 8-byte entry; 0 = 8 bytes.
 * "cherries": 1 (default) = on such trees the second node of a pair, when it is a leaf whose block of leaf slots consists
 * of sibling pairs, is read from the pair's cherry record (one record of rec_b's size per two leaves: half the table);
 * 0 = from rec_b.
 * "wire48": 1 (default) = on trees of fewer than 2^24 nodes the host entry points ship ids over the link as 24 bits
 * each (6 bytes per pair instead of 8; the packing step then checks the range and keeps the id to report); 0 = int32.
 * "wire24": 1 (default) = on such trees MRCA ids come back over the link as 24 bits each (7 bytes per pair with the
 * float32 distance instead of 8; assembled by the kernels, widened by the host's unpack pass); 0 = int32.
 * "reserve_cus": CUs the launches leave to others (default 0): the kernels are persistent workgroups sized to the
 * device; a rank that receives result slices while it computes (the root of the multi-GPU gather) can leave RCCL's
 * kernels a few CUs of their own.
 * "small_batch_path": 1 (default) = host batches of <= 8192 pairs go through a pinned,
 * device-mapped mailbox (one launch; completion is polled in host memory), 0 = through the
 * staged pipe.
 * "batch_probe": 1 (default) = on deep trees whose handle sends large distance batches to the scalar ladder kernel and
 * whose tile-sorted walk kernel is ready as well, every device-resident batch of >= 2^22 explicit pairs is sampled on the device
 * (1024 pairs, one from every 1024th of the batch at a hashed offset: do both nodes share their portal?) and goes to the walk
 * kernel when a quarter of the sample does -- batches of close relatives -- else to the ladder kernel.  Both kernels are enqueued
 * and every workgroup of either takes the sample itself; the kernel it does not choose returns at once: no probe launch, no host
 * round trip, 8-10 us per batch (version 5's separate probe kernel, from 524288 pairs: 20 us).  0 = always the handle's choice.
 * "measure" (default 0; MEASUREMENT ONLY): bits 1 = one line per host-path call on stderr with the host thread's time by
 * phase, 2 = the host path skips its pack / unpack passes, 4 = it launches nothing.  A call made with bit 2 or 4 set
 * returns ST_ERR_MEASURE_ONLY, never ST_OK: its result arrays are not valid. */
int st_tree_set_option(st_tree *tree, const char *name, int64_t value);

/*
 * Host-only helper, no GPU needed: edges-to-root for every node and the
 * reference's `depth` (MuchTree.pyx:218-225).  out_depths may be NULL.
 */
int st_host_depths(const int32_t *parent, int64_t n_nodes, int32_t *out_depths,
                   int32_t *out_tree_depth);

/*
 * Host-only helper, no GPU needed: the link-pair draws of SuchLinkedTrees.sample_linked_distances
 * (SuchTree/MuchTree.pyx:3025-3038) with the reference's own generator, xorshift64* (`_random_int`, :2937-2949:
 * state ^= state >> 12; state ^= state << 25; state ^= state >> 27; draw = (state * 2685821657736338717) % n_links).
 * For each of `count` samples two draws l1, l2 in that order; query_a[k] = (linklist[l1][1], linklist[l2][1]),
 * query_b[k] = (linklist[l1][0], linklist[l2][0]); `linklist` is (n_links, 2) int64, row-major.  `state` is read and
 * left at the generator's state after the last draw, so consecutive calls continue one sequence.
 */
int st_link_sample_pairs(uint64_t *state, const int64_t *linklist, int64_t n_links, int64_t count,
                         int64_t *query_a, int64_t *query_b);

/*
 * Host-only helper: the per-bucket running moments of sample_linked_distances (SuchTree/MuchTree.pyx:3044-3048 as
 * compiled, SuchTree/MuchTree.c:65197-65242): for i < buckets, j < n in that order
 * sums[i] += dist[i * n + j]; sumsq[i] += pow(dist[i * n + j], 2.0) -- doubles, the C library's pow.
 */
int st_bucket_moments(const double *dist, int64_t buckets, int64_t n, double *sums, double *sumsq);

/*
 * Native Newick ingest (host only).  Replaces the dendropy calls of SuchTree.__init__
 * (SuchTree/MuchTree.pyx:138-157, 171-216): first tree of the text, polytomies resolved,
 * nodes numbered in order.  st_newick_open parses and reports sizes; st_newick_fill copies
 * the flat arrays (any pointer may be NULL) -- leaf names come back concatenated, in
 * increasing leaf-id order, with n_leaves+1 byte offsets; st_newick_close frees.
 * ST_ERR_TREE means "not handled here" (syntax error or a token whose Python meaning is
 * not reproduced): callers fall back to suchtree_amd/newick.py, which owns the errors.
 */
typedef struct st_newick st_newick;
int st_newick_open(const char *text, int64_t len, st_newick **out, int64_t *n_nodes,
                   int64_t *n_leaves, int64_t *names_bytes, int32_t *root, int32_t *depth);
int st_newick_fill(const st_newick *h, int32_t *parent, int32_t *left, int32_t *right,
                   float *support, float *distance, int32_t *leaf_ids, char *names,
                   int64_t *name_offsets);
void st_newick_close(st_newick *h);

/* Thin device-memory helpers so callers without torch can stage buffers. */
int st_device_malloc(int device, int64_t bytes, void **out);
int st_device_free(int device, void *ptr);
int st_memcpy_h2d(int device, void *dst, const void *src, int64_t bytes);
int st_memcpy_d2h(int device, void *dst, const void *src, int64_t bytes);
int st_device_synchronize(int device);

#ifdef __cplusplus
}
#endif
#endif /* SUCHTREE_HIP_H */
