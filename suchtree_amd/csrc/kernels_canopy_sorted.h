// kernels_canopy_sorted.h -- k_canopy_sorted, the tile-sorted ladder kernel of deep canopies
// (launch_canopy_sorted.hip).  Include after kernels_canopy.h (CanopyParams and the per-pair helpers).
#pragma once

namespace st {

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
// (tile scratch: launch_geometry.h::sort_scratch_bytes)
// SUMS: the lineage-sum mode (a separate instantiation: it needs about 120 VGPRs, the other
// modes stay below 64, which is what lets two of their workgroups share a CU).
template <int CAP, int Q, bool SUMS, typename Src>
__global__ __launch_bounds__(kCanopyBlock) void k_canopy_sorted(CanopyParams P, Src src, long long n,
                                                                DistSink out_d, MrcaSink out_m,
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
            // The MRCA ids are known here and leave at once, coalesced (converged: every lane of the workgroup is here).
            // Pairs that share a portal (their id follows from the sorted phase: store_mrca) leave a zero.
            if (out_m.any()) {
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const bool shared = (va_[q].x & 0xFFFFu) == (vb_[q].x & 0xFFFFu);
                    store_mrca_wave(out_m, base + (int)threadIdx.x + q * kCanopyBlock,
                                    !valid_[q] ? -1 : shared ? 0 : (int)(uint32_t)e1_[q], in_[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = (int)threadIdx.x + q * kCanopyBlock;
                key[q] = 0xFFFFFFFFu;
                rank[q] = 0;
                if (!in_[q]) continue;
                if (!valid_[q]) {
                    record_fault(fault, a_[q], b_[q], P.n_nodes);
                    SIDE_A[j] = __builtin_nanf("");      // (the distances of the tile leave LDS together, below)
                    continue;
                }
                uint32_t k = 0;
                if ((va_[q].x & 0xFFFFu) == (vb_[q].x & 0xFFFFu)) {     // shared portal: left to the general form
                    MEET[j] = 0xFFFFFFFFu;
                } else {
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
        // packed MRCA ids: the key phase's dword stores have left a zero where a shared-portal pair's id will go, and
        // the sorted phase writes those three bytes from another wave -- make sure the dwords have landed first
        if (have_sums && out_m.m24) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                        dist = ladder_sum_b<CAP>(LdsLadder(lds_raw), kb - (cur.wb >> 16), SIDE_A[j],
                                                 cur.wb & 0xFFFFu, cur.chain(), cur.wb >> 16);
                    } else {     // shared portal
                        long long a, b;
                        src.load(base + j, a, b);
                        PairResult r;
                        if (!P.rec_i) {      // id chains left out under a table budget
                            r = pair_walk(P.nodes, P.depth, P.stride, (int32_t)a, (int32_t)b);
                        } else if constexpr (CAP == 1 || CAP == 3 || CAP == 7 || CAP == 15) {      // chains in registers, no dependent loads
                            const RecTables R{P.rec_a, P.rec_b, P.rec_i, rec_bytes / 2};
                            r = pair_same_portal_regs<CAP>(P.canopy_id, R, record_slot(a, parity, P.n_leaves), record_slot(b, parity, P.n_leaves));
                        } else {
                            r = canopy_pair_scalar<CAP, true>(P, lds_raw, record_slot(a, parity, P.n_leaves),
                                                              record_slot(b, parity, P.n_leaves), rec_bytes, 0xFFFFFFFFu);
                        }
                        dist = r.dist;
                        store_mrca(out_m, base + j, r.mrca);
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
                    if (base + j < n) store_dist(out_d, base + j, SIDE_A[j]);
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
