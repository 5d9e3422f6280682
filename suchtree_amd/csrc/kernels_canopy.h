// kernels_canopy.h -- canopy family: CanopyParams, the per-pair device helpers, k_canopy_ladder, k_canopy_ilp,
// k_mrca_ranks (launch_canopy.hip); k_canopy_sorted lives in kernels_canopy_sorted.h
// (launch_canopy_sorted.hip).  Include after device_common.h and pair_math.h.
#pragma once
#include "launch_geometry.h"

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
    const uint8_t *rec_c;          // [ceil(n_leaves / 2) * rec_bytes/2] cherry records (tree_prep.h), or NULL
    int32_t leaf_block_shift, leaf_block_count;
    const uint8_t *rec_b;          // [n_nodes * rec_bytes/2]  {word0, chain lengths}
    const uint8_t *rec_i;          // [n_nodes * rec_bytes/2]  {pbot, chain node ids}; NULL: left out under a table budget
    const Node8 *nodes;            // the walk family's tables: pairs that share a portal are walked on the tree itself when
    const int32_t *depth;          //   rec_i is NULL (same_portal_by_walk)
    const Stride3 *stride;
    const uint8_t *rec_p;          // [n_nodes * 8]            {portal rank | depth << 16, lineage offset | chunks << 28}; NULL without lineage sums
    const uint64_t *rmq64;         // [levels * canopy_nodes] sparse table with node ids (tree_prep.h); in-order ids only
    const uint16_t *rec_r;         // [n_nodes] rank of the node's portal (MRCA-only queries); in-order ids only
    const float *lineage;          // lineage sums (tree_prep.h): a's whole side of a pair in one read
    long long n_nodes;
    long long n_leaves;
    int32_t canopy_nodes;
    int32_t rec_bytes;
    int32_t parity;                // 1: leaf records first (leaves are the even ids)
};

// Both lineages enter the canopy at the same node and the id chains (rec_i) were left out under a table budget: the
// pair is walked on the tree itself (pair_math.h::pair_walk: depth cut over the stride-3 image, then both sums in the
// reference's order -- the same bits).  Rare for random pairs; all there is for pairs of nearby leaves.
__device__ __forceinline__ PairResult same_portal_by_walk(const CanopyParams &P, long long sa, long long sb)
{
    const bool parity = P.parity != 0;
    return pair_walk(P.nodes, P.depth, P.stride, (int32_t)record_node(sa, parity, P.n_leaves), (int32_t)record_node(sb, parity, P.n_leaves));
}

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
        L.Db[0] = __uint_as_float(kChainPad);
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
            uint4 v = make_uint4(kChainPad, kChainPad, kChainPad, kChainPad);      // (a chunk that is not loaded holds no chain slot in use)
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
            const LdsLadder lad(image);
            if (meet != 0xFFFFFFFFu)      // meeting node known from the sparse table: only the sums remain
                return pair_ladder_sums<CAP>(lad, P.canopy_id, meet, pa, P.cdepth[pa], L.pbot_a, pb, P.cdepth[pb], L.chain(), L.wb >> 16);
            return pair_ladder_split<CAP>(lad, P.cdepth, P.canopy, P.canopy_id, pa, L.pbot_a, pb, L.chain(), L.wb >> 16);
        }
        return pair_canopy_split<CAP>(reinterpret_cast<const CanopyEntry *>(image), P.canopy_id, pa, L.pbot_a, pb,
                                      L.chain(), L.wb >> 16);
    }
    if (!P.rec_i) return same_portal_by_walk(P, sa, sb);
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

__device__ __forceinline__ void stage_ladder(const CanopyParams &P, unsigned char *lds_raw)
{
    stage_ladder_image(lds_raw, P.ladder, P.canopy_nodes);
    __syncthreads();
}

// b's record (L.rb set) for long chains: the first 128-byte line at once, of a 63-slot chain's second line only the
// 16-byte chunks that hold slots in use, once the length is there (k_canopy_ilp reads its records the same way).
template <int CAP>
__device__ __forceinline__ void load_rec_b_lazy(PairRecs<CAP> &L)
{
    if constexpr (CAP <= 31) {
        load_rec_b<CAP>(L);
    } else {
        uint32_t w[CAP + 1];
        constexpr int kChunks = (CAP + 1) / 4, kEager = 8;
#pragma unroll
        for (int q = 0; q < kEager; q++) {
            const uint4 x = reinterpret_cast<const uint4 *>(L.rb)[q];
            w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
        }
        L.wb = w[0];
#pragma unroll
        for (int q = kEager; q < kChunks; q++) {
            uint4 x = make_uint4(kChainPad, kChainPad, kChainPad, kChainPad);
            if ((uint32_t)(4 * q) <= (L.wb >> 16)) x = reinterpret_cast<const uint4 *>(L.rb)[q];      // slots 4q-1 .. 4q+2
            w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int q = 0; q < CAP; q++) L.Db[q] = __uint_as_float(w[q + 1]);
    }
}

// Scalar kernel over the LADDER image (deep canopies whose records are too long for the tile-sorted kernel's second
// read of them: 1e6-leaf trees a few hundred levels deep, 63-slot chains).  Records are read once, in input order,
// the chain stays in registers; with in-order ids the meeting node comes from the canopy's sparse table (two rank
// reads and two table reads, all cache resident), so both sides climb towards a known node, three edges per
// 16-byte LDS read (pair_math.h: ladder_climb_to -- read, compare, three adds): max k_a / 3 + max k_b / 3 dependent
// LDS reads per wave where the predicated kernel's depth cut takes max(k_a, k_b) + max k_b rounds of two reads each.
// Without the table: the lock-step search on the ladder.  What binds it (round 5, profiles/ladder_ablation_r05.log): the
// kernel is balanced -- on nj.tree its memory side alone (ids, records, ranks, two sparse-table entries: 5.7 reads that
// miss L1 per pair) takes 0.41 of the 0.50 ms per 1e7 pairs, its climbs about as long, and a CU's 16 or 32 waves overlap
// the two.  Forms that shrank one side only were bit-exact and no faster: two pairs per lane, climbs dealt again among
// the workgroup's lanes, a software pipeline over the passes (profiles/ladder_*_r04.log); lane-granular refill, K climbs
// per lane in lock step, fewer lookups, fewer dependent round trips, a block form of the meeting-node query
// (profiles/ladder_refill_r05.log, ladder_ablation_r05.log).  Round 6 shrank both sides in one form -- SUMS, ladder_pair_sums
// below: nj.tree +17 %, chosen per tree and batch size by the handle's timing -- and measured two more forms on top of it, K pairs
// per lane level by level and the K climbs of a lane in one loop: slower (profiles/ladder_joint_r06.log).
// (launch bounds: two 1024-lane workgroups per CU = 8 waves per SIMD need at most 64 VGPRs AND at most 80 SGPRs --
// the hardware admits floor(800 / (sgprs rounded up to 16 + 16)) waves per SIMD -- so the short-record form, whose
// ladder image can leave room for two workgroups (ml.tree: 74 KiB), is compiled for 8; at 92 SGPRs it ran one
// workgroup per CU and ml.tree at 2.0e10 pairs/s instead of 2.85e10)
constexpr int kLadderChunk = 128;      // pairs a wave takes per visit to its work counter (two passes of 64)

// One pair of the scalar ladder kernel (valid ids): records, meeting node, both climbs.
template <int CAP>
__device__ __forceinline__ PairResult ladder_pair(const CanopyParams &P, const unsigned char *lds_raw, long long a, long long b,
                                                  bool parity, int rec_bytes)
{
    const long long sa = record_slot(a, parity, P.n_leaves), sb = record_slot(b, parity, P.n_leaves);
    PairRecs<CAP> L;
    L.rb = P.rec_b + sb * (rec_bytes / 2);
    const uint2 va = reinterpret_cast<const uint2 *>(P.rec_a)[sa];
    L.wa = va.x;
    L.pbot_a = __uint_as_float(va.y);
    load_rec_b_lazy<CAP>(L);
    const uint32_t pa = L.wa & 0xFFFFu, pb = L.wb & 0xFFFFu;
    const uint32_t meet = (P.rmq && pa != pb) ? canopy_meet(P.cpos, P.rmq, P.canopy_nodes, pa, pb) : 0xFFFFFFFFu;
    return canopy_pair_finish<CAP, true>(P, lds_raw, L, sa, sb, rec_bytes, meet);
}

// The same pair with a's WHOLE side read from the lineage-sum table (tree_prep.h; round 6, the "joint form": the memory side
// and the climb side shrink together).  rec_p of either node = {rank of its portal | its depth << 16, offset of its lineage
// sums | record chunks << 28}: the meeting node (depth << 32 | node id) comes from the two ranks and the 64-bit sparse
// table -- no rank, depth or canopy_id lookup (seven gathers of ladder_pair become two) --, a's side is ONE 4-byte read
// (the reference's accumulator after a's edges up to the meeting node, bit for bit), b's record is read by the chunks that
// hold chain slots in use, and only b's canopy edges are climbed in LDS: one climb per pair instead of two, so a wave waits
// for the longest of 64 climbs once.  Same operands, same order as ladder_pair: the same bits.
template <int CAP>
__device__ __forceinline__ PairResult ladder_pair_sums(const CanopyParams &P, const unsigned char *lds_raw, long long a, long long b,
                                                       bool parity, int rec_bytes)
{
    const long long sa = record_slot(a, parity, P.n_leaves), sb = record_slot(b, parity, P.n_leaves);
    const uint2 va = reinterpret_cast<const uint2 *>(P.rec_p)[sa];
    const uint2 vb = reinterpret_cast<const uint2 *>(P.rec_p)[sb];
    const uint32_t ra = va.x & 0xFFFFu, rb = vb.x & 0xFFFFu;
    const uint32_t l = ra < rb ? ra : rb, r = ra < rb ? rb : ra;
    const uint32_t k = 31u - (uint32_t)__clz((int)(r - l + 1));      // floor(log2(len))
    const uint64_t e1 = P.rmq64[(size_t)k * (size_t)P.canopy_nodes + l];
    const uint64_t e2 = P.rmq64[(size_t)k * (size_t)P.canopy_nodes + (r + 1 - (1u << k))];
    PairRecs<CAP> L;
    L.rb = P.rec_b + sb * (rec_bytes / 2);
    load_rec_b_chunks<CAP>(L, (vb.y >> 28) + 1);      // (a lane that does not load a chunk does not cost a cache lookup)
    const uint64_t e = (e2 >> 32) < (e1 >> 32) ? e2 : e1;      // the meeting node (shared portal: the portal itself, every index below stays in range)
    const uint32_t dm = (uint32_t)(e >> 32);
    const float side = P.lineage[(size_t)(va.y & 0x0FFFFFFFu) + ((va.x >> 16) - dm)];
    PairResult res;
    res.dist = ladder_sum_b<CAP>(LdsLadder(lds_raw), (vb.x >> 16) - dm - (L.wb >> 16), side, L.wb & 0xFFFFu, L.chain(), L.wb >> 16);
    res.mrca = (int32_t)(uint32_t)e;
    if (ra == rb) res = canopy_pair_scalar<CAP, true>(P, lds_raw, sa, sb, rec_bytes);      // shared portal (rare): the general form
    return res;
}

// `work`: NULL = pairs are dealt statically (workgroup b takes tiles b, b + G, ...), else eight counters (one per XCD,
// zeroed before the launch) from which every WAVE draws chunks of kLadderChunk pairs: the kernel sorts nothing, a wave
// is as slow as its longest lane, and with a static deal the launch ends when the unluckiest wave does.  XCD x owns
// the x-th eighth of the batch (its waves draw from counter x first, then help the next XCDs), so one counter sees an
// eighth of the requests.  Pays where a pair is heavy (profiles/ladder_dynamic_r04.log, 1e7 pairs: 1e6 leaves at depth
// 338, 1 KB records 8.45e9 -> 9.46e9 pairs/s; depth 173, 512-byte records 1.68 -> 1.77e10; nj.tree even; ml.tree
// 2.73 -> 2.53e10: a draw's round trip is as long as its 128 pairs): launch_canopy.hip turns it on by record size.
template <int CAP, typename Src, bool SUMS = false>
__global__ __launch_bounds__(kCanopyBlock, (CAP == 15 ? 8 : 4)) void k_canopy_ladder(CanopyParams P, Src src, long long n,
                                                                DistSink out_d, MrcaSink out_m, Fault *fault,
                                                                unsigned long long *work, int *choice)
{
    static_assert(CAP == 0 || CAP == 15 || CAP == 31 || CAP == 63, "long chains in registers, or (0) through a pointer");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if (choice) {      // a probed batch (pair_math.h: probe_says_walk): the tile-sorted walk kernel, launched beside this one, reaches the same verdict
        const bool walk = probe_says_walk(P.rec_r, src, n, P.n_nodes, P.n_leaves, P.parity != 0, reinterpret_cast<int *>(lds_raw));
        if (blockIdx.x == 0 && threadIdx.x == 0) *choice = walk ? 1 : 0;      // (diagnostic: st_probe_last_choice)
        if (walk) return;
    }
    stage_ladder(P, lds_raw);
    const int rec_bytes = CAP > 0 ? 8 * (CAP + 1) : P.rec_bytes;
    const bool parity = P.parity != 0;
    auto one = [&](long long i) {      // (converged: the whole wave, consecutive pair numbers)
        const bool live = i < n;
        PairResult r;
        r.dist = __builtin_nanf("");
        r.mrca = -1;
        if (live) {
            long long a, b;
            src.load(i, a, b);
            if ((unsigned long long)a >= (unsigned long long)P.n_nodes || (unsigned long long)b >= (unsigned long long)P.n_nodes)
                record_fault(fault, a, b, P.n_nodes);
            else if constexpr (SUMS)
                r = ladder_pair_sums<CAP>(P, lds_raw, a, b, parity, rec_bytes);
            else
                r = ladder_pair<CAP>(P, lds_raw, a, b, parity, rec_bytes);
        }
        store_result_wave(out_d, out_m, i, r.dist, r.mrca, live);
    };
    if (!work) {
        const long long stride = (long long)gridDim.x * blockDim.x;
        for (long long base = (long long)blockIdx.x * blockDim.x; base < n; base += stride) one(base + threadIdx.x);
        return;
    }
    const int lane = threadIdx.x & 63;
    const unsigned xcd = __builtin_amdgcn_s_getreg(20 | ((4 - 1) << 11)) & 7u;      // HW_REG_XCC_ID
    const long long chunks = (n + kLadderChunk - 1) / kLadderChunk;
    // which counters this workgroup has seen run dry: one failed draw per counter and WORKGROUP instead of one per
    // wave (the 57,000 failing draws of 8192 waves at the end of a launch took 0.1-0.2 ms; a plain load of the counter
    // before every draw is worse still -- 5e9 pairs/s flat: loads and atomics of all waves queue on one line)
    // (they live behind the image, which then starts at LDS address 0: a masked link IS the address of the next read)
    unsigned *dry = reinterpret_cast<unsigned *>(lds_raw + (size_t)P.canopy_nodes * sizeof(LadderEntry));
    if (threadIdx.x < 8) dry[threadIdx.x] = 0;
    __syncthreads();
    for (unsigned turn = 0; turn < 8; turn++) {
        const unsigned x = (xcd + turn) & 7u;
        const long long first = chunks * x / 8, last = chunks * (x + 1) / 8;      // this counter's chunks
        const unsigned long long count = (unsigned long long)(last - first);
        // (drawing the next chunk while this one is computed measured no better: the extra draw every wave then wastes
        // at the end costs what the hidden round trips save)
        for (;;) {
            unsigned long long c = ~0ull;
            if (lane == 0 && !reinterpret_cast<volatile unsigned *>(dry)[x]) {
                c = atomicAdd(&work[x * 8], 1ull);      // (counters 64 bytes apart)
                if (c >= count) reinterpret_cast<volatile unsigned *>(dry)[x] = 1;
            }
            c = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)c) |
                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(c >> 32)) << 32);
            if (c >= count) break;      // (also the "nothing drawn" value ~0)
            const long long base = (first + (long long)c) * kLadderChunk;
#pragma unroll 1
            for (int j = 0; j < kLadderChunk / 64; j++) one(base + j * 64 + lane);
        }
    }
}

// The predicated kernel (shallow canopies; the headline): one pair per lane, ds_read_b64 per canopy entry (low word = dist
// bits, high word = parent index).  All updates are predicated selects (a finished climb keeps re-reading its meeting
// node), so nothing serialises behind a branch.  (PPL: pairs per lane, 1 -- two measured equal and were dropped in round 5,
// as was the branchy scalar kernel k_canopy, last on every tree of profiles/kernel_choice_r0{4,5}.log.)
// A4: the four-byte form of the a side (tree_prep.h): pbot from rec_a4 (4 bytes, a table half the size
// of rec_a), the portal from the block table of leaf slots, staged into LDS behind the canopy image;
// leaves of straddling blocks and internal nodes take the 8-byte entry (a wave-uniform rare branch).
// (launch bounds: short records on canopies of at most 80 KiB run two workgroups per CU = 8 waves per SIMD, which the
// hardware only admits at <= 64 VGPRs and <= 80 SGPRs; see k_canopy_ladder)
template <int CAP, int PPL, typename Src, bool A4 = false>
__global__ __launch_bounds__(kCanopyBlock, ((CAP <= 7 && PPL == 1) ? 8 : 4)) void k_canopy_ilp(CanopyParams P, Src src, long long n,
                                                             DistSink out_d,
                                                             MrcaSink out_m, Fault *fault)
{
    static_assert(CAP == 1 || CAP == 3 || CAP == 7 || CAP == 15 || ((CAP == 31 || CAP == 63) && PPL == 1), "register-resident chains only");
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
        uint32_t u[PPL], v[PPL], pa[PPL], pb[PPL];
        float s[PPL], Db[PPL][CAP];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint8_t *rb = P.rec_b + sb[j] * (rec_bytes / 2);
            uint32_t wa;
            // b's record: its own, or -- a leaf of a block whose leaves all sit in sibling pairs -- the pair's cherry record
            // (tree_prep.h: rec_c, a table half the size of rec_b): the portal then comes from the block table
            uint32_t cherry_portal = 0xFFFFFFFFu;
            if (A4) {
                s[j] = P.rec_a4[sa[j]];
                wa = sa[j] < P.n_leaves ? (uint32_t)BLK[sa[j] >> P.leaf_block_shift] : (uint32_t)kLeafBlockMixed;
                if (wa == kLeafBlockMixed) wa = reinterpret_cast<const uint32_t *>(P.rec_a)[2 * sa[j]];      // rare
                else wa &= kLeafBlockPortalMask;
                if (P.rec_c) {
                    const uint32_t eb = sb[j] < P.n_leaves ? (uint32_t)BLK[sb[j] >> P.leaf_block_shift] : (uint32_t)kLeafBlockMixed;
                    if (eb != kLeafBlockMixed && (eb & kLeafBlockCherries)) {
                        cherry_portal = eb & kLeafBlockPortalMask;
                        rb = P.rec_c + (sb[j] >> 1) * (rec_bytes / 2);
                    }
                }
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
                if (A4 && cherry_portal != 0xFFFFFFFFu) {      // {first leaf's length, second leaf's}
                    Db[j][0] = __uint_as_float((sb[j] & 1) ? q.y : q.x);
                    wb = cherry_portal | (1u << 16);
                }
            } else {
                uint32_t w[CAP + 1];
                // the first 128-byte line of the chain at once; of a 63-slot chain's second line only the 16-byte
                // chunks that hold slots in use, once the length is there (mean understory of a 1e6-leaf tree
                // with 512-byte records: 34 nodes; 43 % of its leaves need no second line, the rest a part of it)
                constexpr int kChunks = (CAP + 1) / 4, kEager = kChunks < 8 ? kChunks : 8;
#pragma unroll
                for (int q = 0; q < kEager; q++) {
                    const uint4 x = reinterpret_cast<const uint4 *>(rb)[q];
                    w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
                }
                wb = w[0];
#pragma unroll
                for (int q = kEager; q < kChunks; q++) {
                    uint4 x = make_uint4(kChainPad, kChainPad, kChainPad, kChainPad);
                    if ((uint32_t)(4 * q) <= (wb >> 16)) x = reinterpret_cast<const uint4 *>(rb)[q];      // slots 4q-1 .. 4q+2
                    w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
                }
#pragma unroll
                for (int q = 0; q < CAP; q++) Db[j][q] = __uint_as_float(w[q + 1]);
                if (A4 && cherry_portal != 0xFFFFFFFFu) {      // {first leaf's length, second leaf's, slots 1 .. CAP-1}
                    Db[j][0] = __uint_as_float((sb[j] & 1) ? w[1] : w[0]);
                    wb = cherry_portal | ((uint32_t)CAP << 16);
                }
            }
            u[j] = wa & 0xFFFFu;
            v[j] = wb & 0xFFFFu;
            pa[j] = u[j];
            pb[j] = v[j];
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
            for (int q = 0; q < CAP; q++) s[j] += Db[j][q];      // (slots beyond the chain hold -0.0f: pair_math.h, kChainPad)
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
                PairResult r;
                if (!P.rec_i) r = same_portal_by_walk(P, sa[j], sb[j]);
                else if constexpr (CAP <= 31) r = pair_same_portal_regs<CAP>(P.canopy_id, R, sa[j], sb[j]);
                else r = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa[j]), rec_view(R, sb[j]));
                s[j] = r.dist;
                m[j] = r.mrca;
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; j++)      // (converged: every lane of the workgroup is here, with consecutive pair numbers)
            store_result_wave(out_d, out_m, base + (long long)j * blockDim.x + threadIdx.x, valid[j] ? s[j] : __builtin_nanf(""),
                              valid[j] ? m[j] : -1, live[j]);
    }
}

// MRCA ids only (common_ancestors_bulk, the six pairs of a quartet), trees with in-order ids: the
// MRCA of two nodes is the shallowest node whose id lies between theirs, and unless both hang
// below the same portal it is a canopy node -- two 4-byte reads (rank of either portal) and two
// entries of the 64-bit sparse table (depth << 32 | node id).  No LDS, no understory records:
// 4.8e10 ids/s on ml.tree where the canopy kernels' MRCA-only mode did 3.0e10.
// CAP: chain slots of the tree's records when the shared-portal case compares them in registers (1 ... 31), 0: by
// the loop (longer chains).
template <int CAP, typename Src>
__global__ __launch_bounds__(256) void k_mrca_ranks(CanopyParams P, Src src, long long n, MrcaSink out_m, Fault *fault)
{
    const bool parity = P.parity != 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long base = (long long)blockIdx.x * blockDim.x; base < n; base += stride) {      // (uniform trip count: store_mrca_wave)
        const long long i = base + threadIdx.x;
        const bool live = i < n;
        int m = -1;
        if (live) {
            long long a, b;
            src.load(i, a, b);
            if ((unsigned long long)a >= (unsigned long long)P.n_nodes || (unsigned long long)b >= (unsigned long long)P.n_nodes) {
                record_fault(fault, a, b, P.n_nodes);
            } else {
                const long long sa = record_slot(a, parity, P.n_leaves), sb = record_slot(b, parity, P.n_leaves);
                const uint32_t ra = P.rec_r[sa], rb = P.rec_r[sb];
                if (ra != rb) {
                    m = (int)(uint32_t)canopy_meet_ranks64(P.rmq64, P.canopy_nodes, ra, rb);
                } else {      // shared portal: the MRCA is the portal or lies in the understory
                    const RecTables R{P.rec_a, P.rec_b, P.rec_i, CAP > 0 ? 4 * (CAP + 1) : P.rec_bytes / 2};
                    if (!P.rec_i) m = pair_walk_mrca(P.nodes, P.depth, P.stride, (int32_t)a, (int32_t)b);
                    else if constexpr (CAP > 0) m = mrca_same_portal_regs<CAP>(P.canopy_id, R, sa, sb);
                    else m = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa), rec_view(R, sb)).mrca;
                }
            }
        }
        store_mrca_wave(out_m, i, m, live);
    }
}

}  // namespace st
