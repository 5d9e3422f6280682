// kernels_canopy.h -- part of libsuchtree_hip.so's single translation unit (included by suchtree_hip.hip,
// in this order: device_common.h, kernels_walk.h, kernels_canopy.h, kernels_misc.h).
// Canopy family: k_canopy, k_canopy_ilp, k_mrca_ranks, k_canopy_sorted.
#pragma once

namespace st {

// --------------------------------------------------------------------------
// canopy kernels
// --------------------------------------------------------------------------
struct CanopyParams {
    const CanopyEntry *canopy;     // [canopy_nodes] global copy, staged to LDS
    const int32_t *canopy_id;      // [canopy_nodes]
    const LadderEntry *ladder;     // [canopy_nodes] ladder form (deep canopies), staged to LDS instead of `canopy`
    const uint16_t *cdepth;        // [canopy_nodes (padded to 8)] canopy depths
    const uint16_t *cpos;          // [canopy_nodes] rank by node id; NULL unless ids are in-order positions
    const uint32_t *rmq;           // [levels * canopy_nodes] sparse table of shallowest nodes (tree_prep.h)
    const uint8_t *rec_a;          // [n_nodes * 8]            {word0, pbot}
    const float *rec_a4;           // [n_nodes]                pbot alone (tree_prep.h: four-byte form of the a side), or NULL
    const uint16_t *leaf_blocks;   // [leaf_block_count]       portal of every aligned block of leaf slots (staged to LDS), or NULL
    int32_t leaf_block_shift, leaf_block_count;
    const uint8_t *rec_b;          // [n_nodes * rec_bytes/2]  {word0, chain lengths}
    const uint8_t *rec_i;          // [n_nodes * rec_bytes/2]  {pbot, chain node ids}
    const uint8_t *rec_p;          // [n_nodes * 8]            {portal rank | depth << 16, lineage offset | chunks << 28}; NULL without lineage sums
    const uint64_t *rmq64;         // [levels * canopy_nodes] sparse table with node ids (tree_prep.h); in-order ids only
    const uint32_t *rec_r;         // [n_nodes] portal rank | depth << 16 (MRCA-only queries); in-order ids only
    const float *lineage;          // lineage sums (tree_prep.h): a's whole side of a pair in one read
    long long n_nodes;
    long long n_leaves;
    int32_t canopy_nodes;
    int32_t rec_bytes;
    int32_t parity;                // 1: leaf records first (leaves are the even ids)
};

constexpr int kCanopyBlock = 1024;

// stage the canopy image into LDS: 16 bytes (two entries) per lane per step, coalesced
__device__ __forceinline__ void stage_canopy(const CanopyParams &P, unsigned char *lds_raw)
{
    const int n16 = (P.canopy_nodes + 1) / 2;
    const uint4 *src = reinterpret_cast<const uint4 *>(P.canopy);
    uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
    for (int k = threadIdx.x; k < n16; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
}

// One pair, scalar: the record loads and the climb, for a valid pair with record slots sa / sb.
// CAP = chain slots per record (rec_bytes = 8*(CAP+1)); CAP == 0 is the generic form for records
// longer than CAP allows in registers, which reads b's chain through a pointer.  LADDER: `image`
// is the ladder form of the canopy (tree_prep.h), else the plain 8-byte entries.
// The record reads of one pair: word0 + pbot of a (8 bytes of rec_a), word0 + chain of b (rec_b).
template <int CAP>
struct PairRecs {
    uint32_t wa, wb;
    float pbot_a;
    float Db[CAP > 0 ? CAP : 1];
    const uint8_t *rb;
    __device__ __forceinline__ const float *chain() const { return CAP > 0 ? Db : reinterpret_cast<const float *>(rb + 4); }
};

// b's record (L.rb set): word0 + chain
template <int CAP>
__device__ __forceinline__ void load_rec_b(PairRecs<CAP> &L)
{
    if (CAP == 1) {
        const uint2 v = *reinterpret_cast<const uint2 *>(L.rb);
        L.wb = v.x;
        L.Db[0] = __uint_as_float(v.y);
    } else if (CAP > 1) {
        uint32_t w[CAP + 1];
#pragma unroll
        for (int q = 0; q < (CAP + 1) / 4; q++) {
            const uint4 v = reinterpret_cast<const uint4 *>(L.rb)[q];
            w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        L.wb = w[0];
#pragma unroll
        for (int q = 0; q < CAP; q++) L.Db[q] = __uint_as_float(w[q + 1]);
    } else {
        L.wb = *reinterpret_cast<const uint32_t *>(L.rb);
        L.Db[0] = 0.0f;
    }
}

// b's record (L.rb set), only its first `chunks` 16-byte chunks (the rest of the chain slots are
// never added: zero).  A lane that does not load a chunk does not cost a cache lookup.
template <int CAP>
__device__ __forceinline__ void load_rec_b_chunks(PairRecs<CAP> &L, uint32_t chunks)
{
    static_assert(CAP == 0 || CAP == 1 || (CAP + 1) % 4 == 0, "record layout");
    if (CAP <= 1) {
        load_rec_b<CAP>(L);
    } else {
        uint32_t w[CAP + 1];
#pragma unroll
        for (int q = 0; q < (CAP + 1) / 4; q++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((uint32_t)q < chunks) v = reinterpret_cast<const uint4 *>(L.rb)[q];
            w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        L.wb = w[0];
#pragma unroll
        for (int q = 0; q < CAP; q++) L.Db[q] = __uint_as_float(w[q + 1]);
    }
}

template <int CAP>
__device__ __forceinline__ void load_pair_recs(const CanopyParams &P, long long sa, long long sb, int rec_bytes, PairRecs<CAP> &L)
{
    L.rb = P.rec_b + sb * (rec_bytes / 2);
    const uint2 va = reinterpret_cast<const uint2 *>(P.rec_a)[sa];
    L.wa = va.x;
    L.pbot_a = __uint_as_float(va.y);
    load_rec_b<CAP>(L);
}

// One pair, scalar, after its record reads: the climb, for a valid pair with record slots sa / sb.
// CAP = chain slots per record (rec_bytes = 8*(CAP+1)); CAP == 0 is the generic form for records
// longer than CAP allows in registers, which reads b's chain through a pointer.  LADDER: `image`
// is the ladder form of the canopy (tree_prep.h), else the plain 8-byte entries.
template <int CAP, bool LADDER>
__device__ __forceinline__ PairResult canopy_pair_finish(const CanopyParams &P, const unsigned char *image,
                                                         const PairRecs<CAP> &L, long long sa, long long sb,
                                                         int rec_bytes, uint32_t meet)
{
    const uint32_t pa = L.wa & 0xFFFFu, pb = L.wb & 0xFFFFu;
    if (pa != pb) {
        if (LADDER) {
            const LadderEntry *lad = reinterpret_cast<const LadderEntry *>(image);
            if (meet != 0xFFFFFFFFu)      // meeting node known from the sparse table: only the sums remain
                return pair_ladder_sums<CAP>(lad, P.canopy_id, meet, pa, P.cdepth[pa], L.pbot_a, pb, P.cdepth[pb], L.chain(), L.wb >> 16);
            return pair_ladder_split<CAP>(lad, P.cdepth, P.canopy_id, pa, L.pbot_a, pb, L.chain(), L.wb >> 16);
        }
        return pair_canopy_split<CAP>(reinterpret_cast<const CanopyEntry *>(image), P.canopy_id, pa, L.pbot_a, pb,
                                      L.chain(), L.wb >> 16);
    }
    const RecTables R{P.rec_a, P.rec_b, P.rec_i, rec_bytes / 2};
    return pair_canopy_same_portal(P.canopy_id, rec_view(R, sa), rec_view(R, sb));
}

template <int CAP, bool LADDER>
__device__ __forceinline__ PairResult canopy_pair_scalar(const CanopyParams &P, const unsigned char *image,
                                                         long long sa, long long sb, int rec_bytes,
                                                         uint32_t meet = 0xFFFFFFFFu)
{
    PairRecs<CAP> L;
    load_pair_recs<CAP>(P, sa, sb, rec_bytes, L);
    return canopy_pair_finish<CAP, LADDER>(P, image, L, sa, sb, rec_bytes, meet);
}

// LDS image of the ladder form: canopy_nodes 16-byte entries (the depths stay in global
// memory: they are read twice per pair, from a table of a few KiB)
__host__ __device__ inline size_t ladder_image_bytes(int canopy_nodes)
{
    return (size_t)canopy_nodes * 16;
}

__device__ __forceinline__ void stage_ladder(const CanopyParams &P, unsigned char *lds_raw)
{
    uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
    const uint4 *src_e = reinterpret_cast<const uint4 *>(P.ladder);
    for (int k = threadIdx.x; k < P.canopy_nodes; k += blockDim.x) dst[k] = src_e[k];
    __syncthreads();
}

// Scalar, branchy kernel (one pair per lane, input order): records longer than 128 bytes and
// the pairs_per_lane = 0 setting.  (Over the ladder image it measured no faster than the
// predicated kernel on 2^17-leaf trees -- those are bound by record fetches too -- so the
// ladder is only used by the tile-sorted kernel.)
template <int CAP, typename Src>
__global__ __launch_bounds__(kCanopyBlock) void k_canopy(CanopyParams P, Src src, long long n,
                                                         DistSink out_d,
                                                         int *__restrict__ out_m, Fault *fault)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    stage_canopy(P, lds_raw);

    const int rec_bytes = CAP > 0 ? 8 * (CAP + 1) : P.rec_bytes;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long a, b;
        src.load(i, a, b);
        if ((unsigned long long)a >= (unsigned long long)P.n_nodes ||
            (unsigned long long)b >= (unsigned long long)P.n_nodes) {
            record_fault(fault, a, b, P.n_nodes);
            store_result(out_d, out_m, i, __builtin_nanf(""), -1);
            continue;
        }
        const long long sa = record_slot(a, P.parity != 0, P.n_leaves);
        const long long sb = record_slot(b, P.parity != 0, P.n_leaves);
        const PairResult r = canopy_pair_scalar<CAP, false>(P, lds_raw, sa, sb, rec_bytes);
        store_result(out_d, out_m, i, r.dist, r.mrca);
    }
}

// Same computation with PPL pairs in flight per lane.  Every lane carries PPL independent
// pairs: their pair and record loads are issued together and their canopy climbs advance in
// the same loop iteration as independent ds_read_b64 (low word = dist bits, high word =
// parent index).  All updates are predicated selects (a finished climb keeps re-reading its
// meeting node), so the PPL chains never serialise behind a branch.
// A4: the four-byte form of the a side (tree_prep.h): pbot from rec_a4 (4 bytes, a table half the size
// of rec_a), the portal from the block table of leaf slots, staged into LDS behind the canopy image;
// leaves of straddling blocks and internal nodes take the 8-byte entry (a wave-uniform rare branch).
__host__ __device__ inline size_t leaf_block_image_bytes(int count) { return ((size_t)count * 2 + 15) & ~(size_t)15; }

template <int CAP, int PPL, typename Src, bool A4 = false>
__global__ __launch_bounds__(kCanopyBlock) void k_canopy_ilp(CanopyParams P, Src src, long long n,
                                                             DistSink out_d,
                                                             int *__restrict__ out_m, Fault *fault)
{
    static_assert(CAP == 1 || CAP == 3 || CAP == 7 || CAP == 15, "register-resident chains only");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const unsigned long long *can = reinterpret_cast<const unsigned long long *>(lds_raw);
    const uint16_t *BLK = reinterpret_cast<const uint16_t *>(lds_raw + (size_t)((P.canopy_nodes + 1) / 2) * 16);
    if (A4) {      // (stage_canopy's barrier covers these stores too)
        const uint4 *src16 = reinterpret_cast<const uint4 *>(P.leaf_blocks);
        uint4 *dst16 = reinterpret_cast<uint4 *>(lds_raw + (size_t)((P.canopy_nodes + 1) / 2) * 16);
        const int n16 = (int)(leaf_block_image_bytes(P.leaf_block_count) / 16);
        for (int k = threadIdx.x; k < n16; k += blockDim.x) dst16[k] = src16[k];
    }
    stage_canopy(P, lds_raw);

    constexpr int rec_bytes = 8 * (CAP + 1);
    const bool parity = P.parity != 0;
    const long long tile = (long long)blockDim.x * PPL;
    for (long long base = (long long)blockIdx.x * tile; base < n; base += (long long)gridDim.x * tile) {
        long long idx[PPL], sa[PPL], sb[PPL], ida[PPL], idb[PPL];
        bool live[PPL], valid[PPL];
        // all PPL pair loads are issued before anything looks at them
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const long long i = base + (long long)j * blockDim.x + threadIdx.x;
            live[j] = i < n;
            idx[j] = live[j] ? i : n - 1;
            src.load(idx[j], ida[j], idb[j]);
        }
        bool any_bad = false;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            valid[j] = (unsigned long long)ida[j] < (unsigned long long)P.n_nodes &&
                       (unsigned long long)idb[j] < (unsigned long long)P.n_nodes;
            any_bad |= !valid[j] && live[j];
            const long long a = valid[j] ? ida[j] : 0, b = valid[j] ? idb[j] : 0;
            sa[j] = record_slot(a, parity, P.n_leaves);
            sb[j] = record_slot(b, parity, P.n_leaves);
        }
        if (any_bad) {
#pragma unroll
            for (int j = 0; j < PPL; j++)
                if (!valid[j] && live[j]) record_fault(fault, ida[j], idb[j], P.n_nodes);
        }
        uint32_t u[PPL], v[PPL], pa[PPL], pb[PPL], nb[PPL];
        float s[PPL], Db[PPL][CAP];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint8_t *rb = P.rec_b + sb[j] * (rec_bytes / 2);
            uint32_t wa;
            if (A4) {
                s[j] = P.rec_a4[sa[j]];
                wa = sa[j] < P.n_leaves ? (uint32_t)BLK[sa[j] >> P.leaf_block_shift] : 0xFFFFu;
                if (wa == 0xFFFFu) wa = reinterpret_cast<const uint32_t *>(P.rec_a)[2 * sa[j]];      // rare
            } else {
                const uint2 va = reinterpret_cast<const uint2 *>(P.rec_a)[sa[j]];
                wa = va.x;
                s[j] = __uint_as_float(va.y);
            }
            uint32_t wb;
            if (CAP == 1) {
                const uint2 q = *reinterpret_cast<const uint2 *>(rb);
                wb = q.x;
                Db[j][0] = __uint_as_float(q.y);
            } else {
                uint32_t w[CAP + 1];
#pragma unroll
                for (int q = 0; q < (CAP + 1) / 4; q++) {
                    const uint4 x = reinterpret_cast<const uint4 *>(rb)[q];
                    w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
                }
                wb = w[0];
#pragma unroll
                for (int q = 0; q < CAP; q++) Db[j][q] = __uint_as_float(w[q + 1]);
            }
            u[j] = wa & 0xFFFFu;
            v[j] = wb & 0xFFFFu;
            pa[j] = u[j];
            pb[j] = v[j];
            nb[j] = wb >> 16;
        }

        // climb 1: find the meeting node; the a-side sum rides along
        bool go = false;
#pragma unroll
        for (int j = 0; j < PPL; j++) go |= u[j] != v[j];
        while (go) {
            go = false;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                // depth cut inside the canopy: both entries are read every round, the deeper
                // side moves (both on a tie), so the climb takes max(ka,kb) rounds, not ka+kb
                const unsigned long long eu = can[u[j]], ev = can[v[j]];
                const uint32_t lu = (uint32_t)(eu >> 32), lv = (uint32_t)(ev >> 32);
                const bool act = u[j] != v[j];
                const bool mu = act && (lu >> 16) >= (lv >> 16);
                const bool mv = act && (lv >> 16) >= (lu >> 16);
                const float s_next = s[j] + __uint_as_float((uint32_t)eu);
                s[j] = mu ? s_next : s[j];
                u[j] = mu ? (lu & kCanopyParentMask) : u[j];
                v[j] = mv ? (lv & kCanopyParentMask) : v[j];
                go |= u[j] != v[j];
            }
        }
        // b's understory, then climb 2 over b's canopy lineage
#pragma unroll
        for (int j = 0; j < PPL; j++) {
#pragma unroll
            for (int q = 0; q < CAP; q++) {
                const float s_next = s[j] + Db[j][q];
                s[j] = (uint32_t)q < nb[j] ? s_next : s[j];
            }
            v[j] = pb[j];
        }
        go = false;
#pragma unroll
        for (int j = 0; j < PPL; j++) go |= v[j] != u[j];
        while (go) {
            go = false;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const bool act = v[j] != u[j];
                const unsigned long long e = can[v[j]];
                const float s_next = s[j] + __uint_as_float((uint32_t)e);
                s[j] = act ? s_next : s[j];
                v[j] = act ? ((uint32_t)(e >> 32) & kCanopyParentMask) : v[j];
                go |= v[j] != u[j];
            }
        }
        int m[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) m[j] = P.canopy_id[u[j]];
        // shared portal (rare for random pairs): the MRCA is the portal or below it
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (pa[j] == pb[j]) {
                const RecTables R{P.rec_a, P.rec_b, P.rec_i, rec_bytes / 2};
                const PairResult r = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa[j]), rec_view(R, sb[j]));
                s[j] = r.dist;
                m[j] = r.mrca;
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (live[j]) {
                if (valid[j]) store_result(out_d, out_m, idx[j], s[j], m[j]);
                else store_result(out_d, out_m, idx[j], __builtin_nanf(""), -1);
            }
        }
    }
}

// MRCA ids only (common_ancestors_bulk, the six pairs of a quartet), trees with in-order ids: the
// MRCA of two nodes is the shallowest node whose id lies between theirs, and unless both hang
// below the same portal it is a canopy node -- two 4-byte reads (rank of either portal) and two
// entries of the 64-bit sparse table (depth << 32 | node id).  No LDS, no understory records:
// 4.8e10 ids/s on ml.tree where the canopy kernels' MRCA-only mode did 3.0e10.
template <typename Src>
__global__ __launch_bounds__(256) void k_mrca_ranks(CanopyParams P, Src src, long long n, int *__restrict__ out_m, Fault *fault)
{
    const bool parity = P.parity != 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long a, b;
        src.load(i, a, b);
        if ((unsigned long long)a >= (unsigned long long)P.n_nodes || (unsigned long long)b >= (unsigned long long)P.n_nodes) {
            record_fault(fault, a, b, P.n_nodes);
            out_m[i] = -1;
            continue;
        }
        const long long sa = record_slot(a, parity, P.n_leaves), sb = record_slot(b, parity, P.n_leaves);
        const uint32_t ra = P.rec_r[sa] & 0xFFFFu, rb = P.rec_r[sb] & 0xFFFFu;
        if (ra != rb) {
            out_m[i] = (int)(uint32_t)canopy_meet_ranks64(P.rmq64, P.canopy_nodes, ra, rb);
        } else {      // shared portal: the MRCA is the portal or lies in the understory
            const RecTables R{P.rec_a, P.rec_b, P.rec_i, P.rec_bytes / 2};
            out_m[i] = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa), rec_view(R, sb)).mrca;
        }
    }
}

// Tile-sorted ladder form for deep canopies (the default there).  On trees like
// data/bigtrees/ml.tree a pair's climb is anything from a few to several hundred LDS rounds, so
// in the kernels above a wave is as slow as its longest lineage and keeps ~30 % of its lanes
// busy.  Here a workgroup takes a tile of Q * 1024 pairs, computes a work estimate per pair (key
// phase, input order), counting-sorts the tile by that key in LDS, and hands every wave 64 pairs
// of similar length (sorted phase): waves, not lanes, differ in run time, and a wave's
// instructions serve 64 active lanes.  Wave w processes sorted groups w, 31-w (, 32+w, 63-w):
// short with long, so the waves of a workgroup finish together.  The canopy sits in LDS in its
// ladder form (tree_prep.h: three edges per 16-byte entry), so a climb of k edges is k/3 LDS reads.
// Three modes, by what the tree offers (sorted_shape):
//   lock-step    any node numbering: the key is the depth of the deeper portal; the meeting node
//                is searched on the ladder (pair_math.h: pair_ladder_split)
//   sparse table in-order ids: the meeting node of every pair comes from canopy_pos / canopy_rmq
//                in the key phase (exact key); both sides are then climbed with known counts
//   lineage sums in-order ids + lineage table (SUMS): a's whole side is one table read in the key
//                phase, the MRCA id leaves there too; the sorted phase climbs b's edges only and
//                the distances leave together, coalesced (see below)
// Not one float addition changes: same operands, same order.
constexpr int kSortBuckets = 256;
// LDS scratch of a tile of Q * 1024 pairs: per pair one uint16 (the sorted order), with the
// sparse table one uint32 (the pair's meeting node; b's edge count in lineage-sum mode), with
// lineage sums two more words (a's side, later the distance; b's record slot), then the bucket
// array and the scan carries
__host__ __device__ constexpr size_t sort_scratch_bytes(int q, bool rmq, bool sums = false)
{
    return (size_t)q * kCanopyBlock * (2 + (rmq ? 4 : 0) + (sums ? 8 : 0)) + (size_t)kSortBuckets * 4 + 64;
}

// SUMS: the lineage-sum mode (a separate instantiation: it needs about 120 VGPRs, the other
// modes stay below 64, which is what lets two of their workgroups share a CU).
template <int CAP, int Q, bool SUMS, typename Src>
__global__ __launch_bounds__(kCanopyBlock) void k_canopy_sorted(CanopyParams P, Src src, long long n,
                                                                DistSink out_d, int *__restrict__ out_m,
                                                                Fault *fault, int key_shift)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const uint16_t *cdep = P.cdepth;
    constexpr int kSortTile = Q * kCanopyBlock;
    stage_ladder(P, lds_raw);
    unsigned char *scratch = lds_raw + ladder_image_bytes(P.canopy_nodes);
    const bool have_rmq = P.rmq != nullptr;
    constexpr bool have_sums = SUMS;
    uint32_t *HIST = reinterpret_cast<uint32_t *>(scratch);          // [kSortBuckets] counts, then exclusive starts
    uint32_t *WSUM = HIST + kSortBuckets;                            // [4] scan carries, [4] = pairs to process
    uint16_t *PERM = reinterpret_cast<uint16_t *>(WSUM + 16);        // [kSortTile] sorted position -> pair of the tile
    uint32_t *MEET = reinterpret_cast<uint32_t *>(PERM + kSortTile); // [kSortTile] meeting node (depth << 16 | index), sparse-table mode; b's edge count, lineage-sum mode
    float *SIDE_A = reinterpret_cast<float *>(MEET + kSortTile);     // [kSortTile] a's side of the pair, then its distance (lineage-sum mode)
    uint32_t *SLOT_B = reinterpret_cast<uint32_t *>(SIDE_A + kSortTile);   // [kSortTile] b's record slot | chunks of its record that matter << 28 (lineage-sum mode)

    const int rec_bytes = CAP > 0 ? 8 * (CAP + 1) : P.rec_bytes;
    const int half = rec_bytes / 2;
    const bool parity = P.parity != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * kSortTile; base < n; base += (long long)gridDim.x * kSortTile) {
        if (threadIdx.x < kSortBuckets) HIST[threadIdx.x] = 0;
        __syncthreads();
        // keys, in units of 2^key_shift levels: the canopy edges the pair will climb (its meeting
        // node comes out of the sparse table right here), or -- ids not in order, no table -- the
        // depth of its deeper portal
        uint32_t key[Q], rank[Q];
        if constexpr (have_sums) {
            // Lineage-sum mode.  rec_p of either node = {rank of its portal | its depth << 16, offset
            // of its lineage sums | record chunks << 28}: the meeting node (depth << 32 | node id)
            // comes from the two ranks, a's whole side is one table read, and the sorted phase only
            // climbs b's edges.  The gathers of the lane's Q pairs are issued level by level --
            // pairs, records, sparse table, lineage sums -- without branches in between, so that
            // all Q chains are in flight together (written pair by pair, each chain waited for the
            // one before).  What the sorted phase needs besides b's record stays in LDS: b's edges
            // below the meeting node, a's side, b's slot.
            long long a_[Q], b_[Q];
            bool in_[Q], valid_[Q];
            uint2 va_[Q], vb_[Q];
            uint64_t e1_[Q], e2_[Q];
            float side_[Q];
            const bool want_d = out_d.any();
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const long long i = base + (int)threadIdx.x + q * kCanopyBlock;
                in_[q] = i < n;
                a_[q] = 0;
                b_[q] = 0;
                if (in_[q]) src.load(i, a_[q], b_[q]);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                valid_[q] = (unsigned long long)a_[q] < (unsigned long long)P.n_nodes &&
                            (unsigned long long)b_[q] < (unsigned long long)P.n_nodes;
                const long long sa = record_slot(valid_[q] ? a_[q] : 0, parity, P.n_leaves);
                const long long sb = record_slot(valid_[q] ? b_[q] : 0, parity, P.n_leaves);
                va_[q] = reinterpret_cast<const uint2 *>(P.rec_p)[sa];
                vb_[q] = reinterpret_cast<const uint2 *>(P.rec_p)[sb];
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const uint32_t ra = va_[q].x & 0xFFFFu, rb = vb_[q].x & 0xFFFFu;
                const uint32_t l = ra < rb ? ra : rb, r = ra < rb ? rb : ra;
                const uint32_t len = r - l + 1;
                const uint32_t k = 31u - (uint32_t)__clz((int)len);      // floor(log2(len))
                e1_[q] = P.rmq64[(size_t)k * (size_t)P.canopy_nodes + l];
                e2_[q] = P.rmq64[(size_t)k * (size_t)P.canopy_nodes + (r + 1 - (1u << k))];
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                e1_[q] = (e2_[q] >> 32) < (e1_[q] >> 32) ? e2_[q] : e1_[q];     // the meeting node
                side_[q] = 0.0f;
                if (want_d)
                    side_[q] = P.lineage[(size_t)(va_[q].y & 0x0FFFFFFFu) + ((va_[q].x >> 16) - (uint32_t)(e1_[q] >> 32))];
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = (int)threadIdx.x + q * kCanopyBlock;
                const long long i = base + j;
                key[q] = 0xFFFFFFFFu;
                rank[q] = 0;
                if (!in_[q]) continue;
                if (!valid_[q]) {
                    record_fault(fault, a_[q], b_[q], P.n_nodes);
                    SIDE_A[j] = __builtin_nanf("");      // (the distances of the tile leave LDS together, below)
                    if (out_m) out_m[i] = -1;
                    continue;
                }
                uint32_t k = 0;
                if ((va_[q].x & 0xFFFFu) == (vb_[q].x & 0xFFFFu)) {     // shared portal: left to the general form
                    MEET[j] = 0xFFFFFFFFu;
                } else {
                    // the MRCA id is known here and leaves at once, coalesced
                    if (out_m) out_m[i] = (int)(uint32_t)e1_[q];
                    if (!want_d) continue;      // MRCA ids only: this pair is done
                    const uint32_t kb = (vb_[q].x >> 16) - (uint32_t)(e1_[q] >> 32);
                    MEET[j] = kb;
                    SIDE_A[j] = side_[q];
                    SLOT_B[j] = (uint32_t)record_slot(b_[q], parity, P.n_leaves) | (vb_[q].y & 0xF0000000u);   // (+ how many 16-byte chunks of b's record matter)
                    k = kb >> key_shift;
                }
                key[q] = k < (uint32_t)kSortBuckets - 1 ? k : (uint32_t)kSortBuckets - 1;
                rank[q] = atomicAdd(&HIST[key[q]], 1u);
            }
        } else {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = (int)threadIdx.x + q * kCanopyBlock;
            const long long i = base + j;
            key[q] = 0xFFFFFFFFu;
            rank[q] = 0;
            if (i < n) {
                long long a, b;
                src.load(i, a, b);
                if ((unsigned long long)a >= (unsigned long long)P.n_nodes ||
                    (unsigned long long)b >= (unsigned long long)P.n_nodes) {
                    record_fault(fault, a, b, P.n_nodes);
                    store_result(out_d, out_m, i, __builtin_nanf(""), -1);
                } else {
                    const long long sa = record_slot(a, parity, P.n_leaves);
                    const long long sb = record_slot(b, parity, P.n_leaves);
                    uint32_t k;
                    const uint32_t pa = *reinterpret_cast<const uint32_t *>(P.rec_a + sa * 8) & 0xFFFFu;
                    const uint32_t pb = *reinterpret_cast<const uint32_t *>(P.rec_b + sb * half) & 0xFFFFu;
                    const uint32_t da = cdep[pa], db = cdep[pb];
                    if (have_rmq) {
                        const uint32_t meet = canopy_meet(P.cpos, P.rmq, P.canopy_nodes, pa, pb);
                        MEET[j] = meet;
                        k = (da + db - 2 * (meet >> 16)) >> key_shift;
                    } else {
                        k = (2 * (da > db ? da : db)) >> key_shift;
                    }
                    key[q] = k < (uint32_t)kSortBuckets - 1 ? k : (uint32_t)kSortBuckets - 1;
                    rank[q] = atomicAdd(&HIST[key[q]], 1u);
                }
            }
        }
        }
        __syncthreads();
        // exclusive scan of the 256 bucket counts (4 waves of 64)
        uint32_t cnt = 0, incl = 0;
        if (threadIdx.x < kSortBuckets) {
            cnt = HIST[threadIdx.x];
            incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            if (lane == 63) WSUM[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < kSortBuckets) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += WSUM[w];
            HIST[threadIdx.x] = carry + incl - cnt;
            if (threadIdx.x == kSortBuckets - 1) WSUM[4] = carry + incl;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; q++)
            if (key[q] != 0xFFFFFFFFu) PERM[HIST[key[q]] + rank[q]] = (uint16_t)((int)threadIdx.x + q * kCanopyBlock);
        __syncthreads();
        const uint32_t total = WSUM[4];
        // wave w: sorted groups w, 31 - w, 32 + w, 63 - w (short pairs with long pairs)
        if constexpr (have_sums) {
            // lineage-sum mode: per pair one global read is left (b's record), issued one
            // group ahead of the climb that uses it
            int jq[Q];
            bool ok[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const uint32_t pos = (uint32_t)((q * 16 + ((q & 1) ? 15 - wave : wave)) * 64 + lane);
                ok[q] = pos < total;
                jq[q] = ok[q] ? (int)PERM[pos] : 0;
            }
            PairRecs<CAP> cur, nxt;
            auto fetch = [&](PairRecs<CAP> &L, int q) {
                const uint32_t w = ok[q] && MEET[jq[q]] != 0xFFFFFFFFu ? SLOT_B[jq[q]] : 0u;
                L.rb = P.rec_b + (long long)(w & 0x0FFFFFFFu) * half;
                if (ok[q] && MEET[jq[q]] != 0xFFFFFFFFu) load_rec_b_chunks<CAP>(L, (w >> 28) + 1);
            };
            fetch(cur, 0);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                if (q + 1 < Q) fetch(nxt, q + 1);
                if (ok[q]) {
                    const int j = jq[q];
                    const uint32_t kb = MEET[j];
                    float dist;
                    if (kb != 0xFFFFFFFFu) {
                        dist = ladder_sum_b<CAP>(reinterpret_cast<const LadderEntry *>(lds_raw), kb - (cur.wb >> 16), SIDE_A[j],
                                                 cur.wb & 0xFFFFu, cur.chain(), cur.wb >> 16);
                    } else {     // shared portal
                        long long a, b;
                        src.load(base + j, a, b);
                        const PairResult r = canopy_pair_scalar<CAP, true>(P, lds_raw, record_slot(a, parity, P.n_leaves),
                                                                           record_slot(b, parity, P.n_leaves), rec_bytes, 0xFFFFFFFFu);
                        dist = r.dist;
                        if (out_m) out_m[base + j] = r.mrca;
                    }
                    SIDE_A[j] = dist;      // the pair's scratch word has served: its distance waits there
                }
                if (q + 1 < Q) cur = nxt;
            }
            // distances leave in input order, coalesced (scattered stores straight from the sorted
            // phase cost a cache lookup per lane and wrote every output line several times)
            __syncthreads();
            if (out_d.any()) {
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const int j = (int)threadIdx.x + q * kCanopyBlock;
                    if (base + j < n) store_result(out_d, nullptr, base + j, SIDE_A[j], 0);
                }
            }
        } else {
#pragma unroll 1
            for (int q = 0; q < Q; q++) {
                const uint32_t pos = (uint32_t)((q * 16 + ((q & 1) ? 15 - wave : wave)) * 64 + lane);
                if (pos >= total) continue;
                const int j = PERM[pos];
                long long a, b;
                src.load(base + j, a, b);     // (the tile was read a moment ago: an L2 hit; validated then)
                const long long sa = record_slot(a, parity, P.n_leaves), sb = record_slot(b, parity, P.n_leaves);
                const PairResult r = canopy_pair_scalar<CAP, true>(P, lds_raw, sa, sb, rec_bytes, have_rmq ? MEET[j] : 0xFFFFFFFFu);
                store_result(out_d, out_m, base + j, r.dist, r.mrca);
            }
        }
        __syncthreads();     // the next tile overwrites PERM and MEET
    }
}

}  // namespace st
