// kernels_walk.h -- walk family: k_walk, k_walk_sorted, the mailbox kernel, the quartet kernels
// (launch_walk.hip).  Include after device_common.h and pair_math.h.
#pragma once
#include "launch_geometry.h"

namespace st {

// --------------------------------------------------------------------------
// walk kernel
// --------------------------------------------------------------------------
struct WalkParams {
    const Node8 *nodes;
    const int32_t *depth;
    const Stride3 *stride;
    const uint64_t *rmq;     // whole-tree sparse table for the meeting node, or NULL
    long long n_nodes;
    LineageView lineage;     // a's side in one read (deep trees with lineage sums), else empty
    const LadderEntry *crown_ladder;   // ladder form of the crown by rank (k_walk_sorted stages it into LDS), or NULL
    const uint16_t *probe_rec_r;       // batch probe (pair_math.h: probe_says_walk): portal ranks by record slot, leaves, layout
    long long probe_n_leaves;
    int probe_parity;
};

template <typename Src>
__global__ __launch_bounds__(256) void k_walk(WalkParams P, Src src, long long n,
                                              DistSink out_d, MrcaSink out_m,
                                              Fault *fault)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long base = (long long)blockIdx.x * blockDim.x; base < n; base += stride) {      // (uniform trip count: store_mrca_wave)
        const long long i = base + threadIdx.x;
        const bool live = i < n;
        PairResult r;
        r.dist = __builtin_nanf("");
        r.mrca = -1;
        if (live) {
            long long a, b;
            src.load(i, a, b);
            if ((unsigned long long)a >= (unsigned long long)P.n_nodes ||
                (unsigned long long)b >= (unsigned long long)P.n_nodes) {
                record_fault(fault, a, b, P.n_nodes);
            } else if (out_d.any()) {
                r = pair_walk(P.nodes, P.depth, P.stride, (int32_t)a, (int32_t)b, P.rmq, P.n_nodes, P.lineage);
            } else {
                r.mrca = pair_walk_mrca(P.nodes, P.depth, P.stride, (int32_t)a, (int32_t)b, nullptr, P.rmq, P.n_nodes);
            }
        }
        store_result_wave(out_d, out_m, i, r.dist, r.mrca, live);
    }
}

// Tile-sorted form of k_walk for trees that have the whole-tree sparse table and both lineage tables
// (tree_prep.h): large AND deep trees the canopy family refuses, and deep-canopy trees when the walk
// family is asked for.  There a pair costs two table reads for the meeting node, one read for a's
// whole side and k_b consecutive floats for b's side -- and k_b is anything from 0 to the depth of
// the tree, so in k_walk a wave is as slow as its longest lane (counters on data/bigtrees/ml.tree:
// 103 scattered load instructions per 64 pairs where the mean lane needs 28).  Here a workgroup
// takes a tile of Q * 1024 pairs.  Key phase (input order, the Q chains of a lane in flight
// together): {depth, offset} of both nodes, the meeting node, a's side; the MRCA id leaves at once,
// coalesced; a's side, b's offset and k_b stay in LDS.  The tile is counting-sorted by k_b, every
// wave streams 64 pairs of similar length (sorted groups w, 31 - w, ...: short with long), and the
// distances leave together in input order, coalesced.  Every pair is read once and every store is
// coalesced, so the kernel may also work on pinned host memory.  Same operands, same order of
// additions as k_walk.
// (tile scratch: launch_geometry.h::walk_sort_scratch_bytes)

// LADDER: the crown is small enough for LDS (tree_prep.h: crown_ladder): its ladder form is staged in
// front of the scratch and the crown part of every b side is climbed there, three edges per 16-byte LDS
// read; only the nodes below the portal are streamed from global memory.
template <int Q, bool LADDER, typename Src>
__global__ __launch_bounds__(kWalkSortBlock) void k_walk_sorted(WalkParams P, Src src, long long n, DistSink out_d,
                                                                 MrcaSink out_m, Fault *fault, int key_shift, int *choice)
{
    extern __shared__ __align__(16) unsigned char walk_lds_all[];
    // `choice` (or NULL): a probed batch (pair_math.h: probe_says_walk) -- this kernel is the batch's kernel when the sample says
    // "close relatives", the scalar ladder kernel, launched beside it and looking at the same sample, when it does not
    if (choice && !probe_says_walk(P.probe_rec_r, src, n, P.n_nodes, P.probe_n_leaves, P.probe_parity != 0, reinterpret_cast<int *>(walk_lds_all))) return;
    const LdsLadder LAD(walk_lds_all);
    unsigned char *walk_lds = walk_lds_all + (LADDER ? (size_t)P.lineage.crown_nodes * 16 : 0);
    if (LADDER) {
        stage_ladder_image(walk_lds_all, P.crown_ladder, P.lineage.crown_nodes);
        __syncthreads();
    }
    constexpr int kTile = Q * kWalkSortBlock;
    uint32_t *HIST = reinterpret_cast<uint32_t *>(walk_lds);     // [kWalkSortBuckets] counts, then exclusive starts
    uint32_t *WSUM = HIST + kWalkSortBuckets;                    // [4] scan carries, [4] = pairs to stream
    uint32_t *KB = WSUM + 16;                                    // [kTile] b's edges below the meeting node | nodes of b below its portal << 24
    float *SIDE = reinterpret_cast<float *>(KB + kTile);         // [kTile] a's side of the pair, then its distance
    uint32_t *OFFB = reinterpret_cast<uint32_t *>(SIDE + kTile); // [kTile] offset of b's block in the lineage tables
    uint32_t *OFFP = OFFB + kTile;                               // [kTile] offset of the block of b's portal
    uint16_t *PERM = reinterpret_cast<uint16_t *>(OFFP + kTile); // [kTile] sorted position -> pair of the tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LineageView &lin = P.lineage;
    for (long long base = (long long)blockIdx.x * kTile; base < n; base += (long long)gridDim.x * kTile) {
        if (threadIdx.x < kWalkSortBuckets) HIST[threadIdx.x] = 0;
        __syncthreads();
        uint32_t key[Q], rank[Q];
        {
            long long a_[Q], b_[Q];
            bool in_[Q], valid_[Q];
            NodeKey ka_[Q], kb_[Q];
            uint64_t e1_[Q], e2_[Q];
            float side_[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const long long i = base + (int)threadIdx.x + q * kWalkSortBlock;
                in_[q] = i < n;
                a_[q] = 0;
                b_[q] = 0;
                if (in_[q]) src.load(i, a_[q], b_[q]);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                valid_[q] = (unsigned long long)a_[q] < (unsigned long long)P.n_nodes &&
                            (unsigned long long)b_[q] < (unsigned long long)P.n_nodes;
                ka_[q] = lineage_key(lin, valid_[q] ? (int32_t)a_[q] : 0);
                kb_[q] = lineage_key(lin, valid_[q] ? (int32_t)b_[q] : 0);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                // the meeting node: of the two portals in the crown's table when they differ (a few
                // hundred KB, cache resident), else of the two nodes in the whole-tree table
                const uint32_t ra = ka_[q].nb_rank >> 8, rb = kb_[q].nb_rank >> 8;
                const bool crown = lin.crown_rmq != nullptr && ra != rb;
                const uint32_t a = crown ? ra : (valid_[q] ? (uint32_t)a_[q] : 0u), b = crown ? rb : (valid_[q] ? (uint32_t)b_[q] : 0u);
                const uint64_t *tab = crown ? lin.crown_rmq : P.rmq;
                const size_t width = crown ? (size_t)lin.crown_nodes : (size_t)P.n_nodes;
                const uint32_t l = a < b ? a : b, r = a < b ? b : a;
                const uint32_t len = r - l + 1;
                const uint32_t k = 31u - (uint32_t)__clz((int)len);      // floor(log2(len))
                e1_[q] = tab[(size_t)k * width + l];
                e2_[q] = tab[(size_t)k * width + (r + 1 - (1u << k))];
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                e1_[q] = e2_[q] < e1_[q] ? e2_[q] : e1_[q];      // the meeting node: depth << 32 | id
                side_[q] = lin.sums[(size_t)ka_[q].off + (size_t)(ka_[q].depth - (uint32_t)(e1_[q] >> 32))];
            }
            // the MRCA ids leave at once, coalesced (converged: every lane of the workgroup is here)
            if (out_m.any()) {
#pragma unroll
                for (int q = 0; q < Q; q++)
                    store_mrca_wave(out_m, base + (int)threadIdx.x + q * kWalkSortBlock, valid_[q] ? (int)(uint32_t)e1_[q] : -1, in_[q]);
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = (int)threadIdx.x + q * kWalkSortBlock;
                key[q] = 0xFFFFFFFFu;
                rank[q] = 0;
                if (!in_[q]) continue;
                if (!valid_[q]) {
                    record_fault(fault, a_[q], b_[q], P.n_nodes);
                    SIDE[j] = __builtin_nanf("");
                    continue;
                }
                const uint32_t kb = kb_[q].depth - (uint32_t)(e1_[q] >> 32);
                KB[j] = kb | ((kb_[q].nb_rank & 0xFFu) << 24);      // (k_b < 2^24: the lineage tables exist)
                SIDE[j] = side_[q];
                OFFB[j] = kb_[q].off;
                OFFP[j] = LADDER ? (kb_[q].nb_rank >> 8) : kb_[q].portal_off;      // (ladder: the portal's rank = its ladder index)
                const uint32_t k = kb >> key_shift;
                key[q] = k < (uint32_t)kWalkSortBuckets - 1 ? k : (uint32_t)kWalkSortBuckets - 1;
                rank[q] = atomicAdd(&HIST[key[q]], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of the 256 bucket counts (4 waves of 64)
        uint32_t cnt = 0, incl = 0;
        if (threadIdx.x < kWalkSortBuckets) {
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
        if (threadIdx.x < kWalkSortBuckets) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += WSUM[w];
            HIST[threadIdx.x] = carry + incl - cnt;
            if (threadIdx.x == kWalkSortBuckets - 1) WSUM[4] = carry + incl;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; q++)
            if (key[q] != 0xFFFFFFFFu) PERM[HIST[key[q]] + rank[q]] = (uint16_t)((int)threadIdx.x + q * kWalkSortBlock);
        __syncthreads();
        const uint32_t total = WSUM[4];
#pragma unroll 1
        for (int q = 0; q < Q; q++) {
            const uint32_t pos = (uint32_t)((q * 16 + ((q & 1) ? 15 - wave : wave)) * 64 + lane);
            if (pos >= total) continue;
            const int j = PERM[pos];
            const int32_t kb = (int32_t)(KB[j] & 0xFFFFFFu), nb = (int32_t)(KB[j] >> 24);
            float s = SIDE[j];
            if (LADDER) {
                s = stream_b_ladder(lin.lens, LAD, OFFB[j], (uint32_t)nb, OFFP[j], s, kb);
            } else if (!lin.shared_blocks || kb <= nb) {
                s = stream_sum(lin.lens + OFFB[j], s, kb);
            } else {      // below the portal from b's own block, above it from the portal's (shared, cache resident)
                s = stream_sum(lin.lens + OFFB[j], s, nb);
                s = stream_sum(lin.lens + OFFP[j], s, kb - nb);
            }
            SIDE[j] = s;
        }
        __syncthreads();
        if (out_d.any()) {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = (int)threadIdx.x + q * kWalkSortBlock;
                if (base + j < n) store_dist(out_d, base + j, SIDE[j]);
            }
        }
        __syncthreads();     // the next tile overwrites the scratch
    }
}

// The mailbox form of k_walk (small host batches, one lane per pair, no grid stride): pairs and
// results live in pinned host memory, and so does a completion word -- the last workgroup to
// finish publishes the call's sequence number there (system-scope release after every block's
// system-scope fence), so the host learns of completion by polling its own memory instead of
// paying a stream synchronisation (the driver's wake-up costs as much as the whole kernel).
__global__ __launch_bounds__(64) void k_walk_mailbox(WalkParams P, const long long *__restrict__ pairs, int n,
                                                     double *__restrict__ out_d, int *__restrict__ out_m,
                                                     unsigned *block_counter, unsigned *done_word, unsigned seq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const longlong2 v = reinterpret_cast<const longlong2 *>(pairs)[i];   // (ids were range-checked on the host)
        if (out_d) {
            const PairResult r = pair_walk(P.nodes, P.depth, P.stride, (int32_t)v.x, (int32_t)v.y, P.rmq, P.n_nodes, P.lineage);
            out_d[i] = (double)r.dist;
            if (out_m) out_m[i] = r.mrca;
        } else {
            out_m[i] = pair_walk_mrca(P.nodes, P.depth, P.stride, (int32_t)v.x, (int32_t)v.y, nullptr, P.rmq, P.n_nodes);
        }
    }
    __threadfence_system();            // this lane's results are visible to the host ...
    __syncthreads();                   // ... and so are the whole block's
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(block_counter, 1u);
        if (done == gridDim.x - 1) {   // last block of the launch
            *block_counter = 0;        // (launches on the mailbox stream are serial)
            __threadfence_system();
            __hip_atomic_store(done_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Quartet topologies (MuchTree.pyx:1331-1376): six MRCAs per quartet (ab ac ad bc bd cd),
// the first MRCA id that occurs exactly once names the sister pair; the row is re-ordered
// by the matching line of the table I = {0123, 0213, 0312, 1203, 1302, 2301}.  If no id is
// unique the reference's loop leaves j = 5, reproduced here.  Integer work only.
__global__ __launch_bounds__(256) void k_quartets(WalkParams P, const long long *__restrict__ q,
                                                  long long n, long long s0, long long s1,
                                                  long long *__restrict__ out, Fault *fault)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long id[4];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            id[k] = q[i * s0 + k * s1];
            ok &= (unsigned long long)id[k] < (unsigned long long)P.n_nodes;
        }
        if (!ok) {
            record_fault(fault, id[0], id[1], P.n_nodes);
            record_fault(fault, id[2], id[3], P.n_nodes);
#pragma unroll
            for (int k = 0; k < 4; k++) out[i * 4 + k] = -1;
            continue;
        }
        const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
        int M[6];
#pragma unroll
        for (int j = 0; j < 6; j++)
            M[j] = pair_walk_mrca(P.nodes, P.depth, P.stride, (int32_t)id[pa[j]], (int32_t)id[pb[j]], nullptr, P.rmq, P.n_nodes);
        int pick = 5;
#pragma unroll
        for (int j = 5; j >= 0; j--) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) c += M[j] == M[k];
            if (c == 1) pick = j;
        }
        // I[pick] = {pa, pb, the other two in increasing order}
        const int a = pa[pick], b = pb[pick];
        int rest[2], r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k != a && k != b) { if (r < 2) rest[r] = k; r++; }
        out[i * 4 + 0] = id[a];
        out[i * 4 + 1] = id[b];
        out[i * 4 + 2] = id[rest[0]];
        out[i * 4 + 3] = id[rest[1]];
    }
}

// Second half of the quartet path when the MRCA ids came from a canopy launch over
// SrcQuartet: M[6*i .. 6*i+5] are the ids of quartet i (-1 where an id was out of range).
__global__ __launch_bounds__(256) void k_quartet_pick(const long long *__restrict__ q, const int *__restrict__ M,
                                                      long long n, long long *__restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int m[6];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 6; j++) { m[j] = M[i * 6 + j]; bad |= m[j] < 0; }
        long long id[4];
#pragma unroll
        for (int k = 0; k < 4; k++) id[k] = q[i * 4 + k];
        if (bad) {
#pragma unroll
            for (int k = 0; k < 4; k++) out[i * 4 + k] = -1;
            continue;
        }
        int pick = 5;
#pragma unroll
        for (int j = 5; j >= 0; j--) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) c += m[j] == m[k];
            if (c == 1) pick = j;
        }
        const int a = (0x940 >> (2 * pick)) & 3, b = (0xFB9 >> (2 * pick)) & 3;
        int rest[2], r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k != a && k != b) { if (r < 2) rest[r] = k; r++; }
        out[i * 4 + 0] = id[a];
        out[i * 4 + 1] = id[b];
        out[i * 4 + 2] = id[rest[0]];
        out[i * 4 + 3] = id[rest[1]];
    }
}

}  // namespace st
