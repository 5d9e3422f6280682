// pair_math.h -- the per-pair arithmetic, shared by the gfx950 kernels and by
// the host-side emulator that the CPU test-suite uses to check the table
// construction (tests only; the product never computes on the CPU).
//
// Contract restated from the reference (/root/reference/SuchTree/MuchTree.pyx):
//   mrca(a,b)  = first node of b's root-ward lineage that is also on a's
//                lineage, a and b themselves included (:999-1030)
//   d(a,b)     = float32 accumulator, 0, += dist[n] for n = a .. below mrca,
//                then += dist[n] for n = b .. below mrca, in that order
//                (:930-943); stored as double.
// Float adds only, strictly left to right: no reassociation, no FMA.
#pragma once
#include <cstdint>

#include "tree_prep.h"

namespace st {

struct PairResult {
    float dist;
    int32_t mrca;
};

// ---- walk family -----------------------------------------------------------
// Works on any rooted tree, no tables but the tree itself.  The meeting node is found first,
// with integer work only: the deeper endpoint is lifted to the other's depth, then both climb
// in lock step -- three levels per gather (Stride3::p3) while their third ancestors differ, one
// level (Node8::parent) once they agree.  Both sums then know their edge counts and add three
// edges per 16-byte gather, in lineage order: a's edges from 0, then b's onto the same
// accumulator, exactly the reference's two loops (MuchTree.pyx:934-942).
// `rmq` (optional): the whole-tree sparse table of tree_prep.h -- the meeting node and its
// depth from two 8-byte reads, no climbing.
ST_HD int32_t pair_walk_mrca(const Node8 *__restrict__ nodes, const int32_t *__restrict__ depth,
                             const Stride3 *__restrict__ stride, int32_t a, int32_t b, int32_t *depth_of_mrca = nullptr,
                             const uint64_t *__restrict__ rmq = nullptr, int64_t n_nodes = 0)
{
    if (rmq) {
        const uint32_t l = (uint32_t)(a < b ? a : b), r = (uint32_t)(a < b ? b : a);
        const uint32_t len = r - l + 1;
        uint32_t k = 0;
        while ((2ull << k) <= len) k++;
        const uint64_t e1 = rmq[(size_t)k * (size_t)n_nodes + l];
        const uint64_t e2 = rmq[(size_t)k * (size_t)n_nodes + (r + 1 - (1u << k))];
        const uint64_t e = e2 < e1 ? e2 : e1;
        if (depth_of_mrca) *depth_of_mrca = (int32_t)(e >> 32);
        return (int32_t)(uint32_t)e;
    }
    int32_t x = a, y = b;
    int32_t dx = depth[x], dy = depth[y];
    while (dx > dy) {
        if (dx - dy >= 3) { x = stride[x].p3; dx -= 3; } else { x = nodes[x].parent; dx -= 1; }
    }
    while (dy > dx) {
        if (dy - dx >= 3) { y = stride[y].p3; dy -= 3; } else { y = nodes[y].parent; dy -= 1; }
    }
    while (x != y) {
        const int32_t px = stride[x].p3, py = stride[y].p3;
        if (px != py) { x = px; y = py; dx -= 3; }
        else { x = nodes[x].parent; y = nodes[y].parent; dx -= 1; }
    }
    if (depth_of_mrca) *depth_of_mrca = dx;
    return x;
}

ST_HD float walk_sum(const Stride3 *__restrict__ stride, float s, int32_t u, int32_t k)
{
    while (k >= 3) {
        const Stride3 e = stride[u];
        s += e.d0; s += e.d1; s += e.d2;
        u = e.p3;
        k -= 3;
    }
    if (k) {
        const Stride3 e = stride[u];
        s += e.d0;
        if (k == 2) s += e.d1;
    }
    return s;
}

// Lineage tables for the walk family (tree_prep.h; NULL when the tree has none): a's side of a
// pair is sums[offset of a + edges of a below the meeting node]; b's side continues that
// accumulator with the first k_b lineage lengths of b, consecutive floats: b's own block for the
// nodes below its portal, then its portal's block (the same floats as the rest of its own block,
// but shared by every node below that portal: a hot set of a few MB).  node_rec[4 x ..] = {depth,
// offset of x's block, offset of its portal's block, nodes below the portal | rank of the portal
// << 8}.  crown_rmq: the meeting node of two nodes with different portals from a table over the
// crown alone.
struct LineageView {
    const uint32_t *node_rec = nullptr;
    const float *sums = nullptr;
    const float *lens = nullptr;          // lineage lengths (same blocks), or NULL: b's side climbs the stride-3 image
    const uint64_t *crown_rmq = nullptr;  // or NULL: meeting nodes from the whole-tree table / by climbing
    int32_t crown_nodes = 0;
    bool shared_blocks = false;           // b's stream switches to its portal's block above the portal
};

struct alignas(16) NodeKey {
    uint32_t depth, off, portal_off, nb_rank;
};

ST_HD NodeKey lineage_key(const LineageView &lin, int32_t x)
{
    return *reinterpret_cast<const NodeKey *>(lin.node_rec + (size_t)x * 4);
}

// Meeting node (depth << 32 | node id) of two nodes whose portals differ, from their ranks.
ST_HD uint64_t crown_meet(const uint64_t *__restrict__ rmq, int32_t n_crown, uint32_t ra, uint32_t rb)
{
    const uint32_t l = ra < rb ? ra : rb, r = ra < rb ? rb : ra;
    const uint32_t len = r - l + 1;
    uint32_t k = 0;
    while ((2u << k) <= len) k++;
    const uint64_t e1 = rmq[(size_t)k * (size_t)n_crown + l];
    const uint64_t e2 = rmq[(size_t)k * (size_t)n_crown + (r + 1 - (1u << k))];
    return e2 < e1 ? e2 : e1;
}

struct alignas(16) Quad {
    float x, y, z, w;
};

// s += p[0]; s += p[1]; ... s += p[k-1], in that order (the reference's b-side loop, pyx:939-942,
// over operands laid out contiguously).  p is 16-byte aligned and its block is padded to whole
// quads, so every load is one aligned 16-byte read; the loads do not depend on each other.
ST_HD float stream_sum(const float *__restrict__ p, float s, int32_t k)
{
    const Quad *__restrict__ q = reinterpret_cast<const Quad *>(p);
    while (k >= 16) {      // one 64-byte sector per trip
        const Quad a = q[0], b = q[1], c = q[2], d = q[3];
        q += 4;
        k -= 16;
        s += a.x; s += a.y; s += a.z; s += a.w;
        s += b.x; s += b.y; s += b.z; s += b.w;
        s += c.x; s += c.y; s += c.z; s += c.w;
        s += d.x; s += d.y; s += d.z; s += d.w;
    }
    while (k >= 4) {
        const Quad a = *q++;
        k -= 4;
        s += a.x; s += a.y; s += a.z; s += a.w;
    }
    if (k) {
        const Quad a = *q;
        s += a.x;
        if (k > 1) s += a.y;
        if (k > 2) s += a.z;
    }
    return s;
}

// b's side: k_b lineage lengths of b onto s, in lineage order -- in one piece from b's own block,
// or (shared blocks) the part below the portal from b's block and the rest from the portal's.
ST_HD float stream_b(const LineageView &lin, const NodeKey &kb, float s, int32_t k)
{
    const int32_t nb = (int32_t)(kb.nb_rank & 0xFFu);
    if (!lin.shared_blocks || k <= nb) return stream_sum(lin.lens + kb.off, s, k);
    s = stream_sum(lin.lens + kb.off, s, nb);
    return stream_sum(lin.lens + kb.portal_off, s, k - nb);
}

// ---- ladder images (tree_prep.h: LadderEntry) -------------------------------
// Where an image lives: PtrLadder = anywhere, places are byte offsets from its first entry (host emulation, tests);
// the kernels read theirs from LDS through LdsLadder (device_common.h), whose places are LDS addresses.
struct PtrLadder {
    const LadderEntry *p;
    ST_HD LadderEntry at(uint32_t place) const { return *reinterpret_cast<const LadderEntry *>(reinterpret_cast<const unsigned char *>(p) + place); }
    ST_HD uint32_t place(uint32_t index) const { return index * (uint32_t)sizeof(LadderEntry); }
    ST_HD uint32_t index(uint32_t place) const { return place / (uint32_t)sizeof(LadderEntry); }
};

#if defined(__HIPCC__)
// An image staged into LDS by stage_ladder_image: places are LDS addresses (the links were moved by the image's own
// address as they were copied), so a read takes its address from the entry before it without arithmetic.
typedef uint32_t LadderWords __attribute__((ext_vector_type(4)));
struct LdsLadder {
    uint32_t base;      // LDS address of entry 0
    __device__ __forceinline__ explicit LdsLadder(const void *image) : base((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)image) {}
    __device__ __forceinline__ LadderEntry at(uint32_t place) const      // (one ds_read_b128)
    {
        const LadderWords w = *(const __attribute__((address_space(3))) LadderWords *)(uintptr_t)place;
        return LadderEntry{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), w.w};
    }
    __device__ __forceinline__ uint32_t place(uint32_t index) const { return base + index * (uint32_t)sizeof(LadderEntry); }
    __device__ __forceinline__ uint32_t index(uint32_t place) const { return (place - base) / (uint32_t)sizeof(LadderEntry); }
};

// (all lanes of the workgroup; the caller's barrier follows)
__device__ __forceinline__ void stage_ladder_image(unsigned char *lds_image, const LadderEntry *from, int entries)
{
    const uint32_t base = LdsLadder(lds_image).base;
    uint4 *dst = reinterpret_cast<uint4 *>(lds_image);
    const uint4 *src = reinterpret_cast<const uint4 *>(from);
    for (int k = threadIdx.x; k < entries; k += blockDim.x) {
        uint4 e = src[k];
        if (e.w != kLadderAbove) e.w += base;
        dst[k] = e;
    }
}
#endif

#if defined(__HIPCC__)
// Batch probe (round 6: folded into the two kernels it chooses between).  Which kernel a large batch of explicit pairs gets on a
// deep tree is decided per batch, on the device: the handle's timing uses uniform random pairs, where the scalar ladder kernel
// leads; batches of close relatives -- both nodes under one portal, short paths -- run 20-35 % faster on the tile-sorted walk
// kernel, whose cost follows the path length (profiles/near_pairs_r0{4,5}.log).  Both kernels are launched; EVERY workgroup of
// either looks at the same kProbePairs pairs of the batch -- one from every stretch of n / kProbePairs, at a hashed offset inside it
// (a fixed stride aliases with batches of periodic structure) -- and reaches the same verdict: the walk kernel when at least a quarter
// of them share their portal (uniform pairs: 0.1 %, leaves within 1024 of each other: 2-3 %, within 64: a third, within 8: 70-80 %).
// The kernel that is not chosen returns before it stages anything.  No probe launch, no host round trip, no word shared between the
// two kernels (round 5's separate probe kernel, its events and the third dispatch cost 20 us per batch; this form costs the empty
// dispatch and two dependent reads per workgroup).  Called by all lanes of a 1024-lane workgroup BEFORE it uses its dynamic LDS:
// `lds_word` is four bytes of it.  rec_r: rank of every node's portal by record slot.
constexpr int kProbePairs = 1024;
template <typename Src>
__device__ __forceinline__ bool probe_says_walk(const uint16_t *__restrict__ rec_r, const Src &src, long long n, long long n_nodes,
                                                long long n_leaves, bool parity, int *lds_word)
{
    if (threadIdx.x == 0) *lds_word = 0;
    __syncthreads();
    const long long step = n / kProbePairs > 0 ? n / kProbePairs : 1;
    bool shared = false;
    if ((int)threadIdx.x < kProbePairs) {
        uint32_t h = (uint32_t)threadIdx.x * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const long long i = (long long)threadIdx.x * step + (long long)((unsigned long long)h % (unsigned long long)step);
        if (i < n) {
            long long a, b;
            src.load(i, a, b);
            if ((unsigned long long)a < (unsigned long long)n_nodes && (unsigned long long)b < (unsigned long long)n_nodes)
                shared = rec_r[record_slot(a, parity, n_leaves)] == rec_r[record_slot(b, parity, n_leaves)];
        }
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(shared);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(lds_word, (int)__builtin_popcountll(m));
    __syncthreads();
    const bool walk = *lds_word * 4 >= kProbePairs;
    __syncthreads();      // (the word belongs to the caller's LDS again)
    return walk;
}

#endif

// k edges of a lineage from the entry at `at`, added onto s in lineage order, three per 16-byte read; the climb counts
// its edges (any numbering of the image).
template <typename Lad>
ST_HD float ladder_climb_counted(const Lad &lad, uint32_t at, uint32_t k, float s)
{
    while (k >= 3) {
        const LadderEntry e = lad.at(at);
        s += e.d0; s += e.d1; s += e.d2;
        at = e.link;
        k -= 3;
    }
    if (k) {
        const LadderEntry e = lad.at(at);
        s += e.d0;
        if (k == 2) s += e.d1;
    }
    return s;
}

// The same on an image that numbers parents before children (the canopy's), towards a known ancestor whose entry is at
// `stop`, k edges above: a round is taken while the third ancestor is still at or below that node -- one signed
// compare of the link as it was read (kLadderAbove is negative) -- and the k % 3 edges left come from the entry the
// loop ends on, which is already there.  Per round: one LDS read, the compare, three adds.
template <typename Lad>
ST_HD float ladder_climb_to(const Lad &lad, uint32_t at, uint32_t stop, uint32_t k, float s)
{
    const uint32_t left = k - 3u * ((k * 0xAAABu) >> 17);      // k % 3 (k < 2^16)
    LadderEntry e = lad.at(at);
    while ((int32_t)e.link >= (int32_t)stop) {
        s += e.d0; s += e.d1; s += e.d2;
        e = lad.at(e.link);
    }
    if (left) {
        s += e.d0;
        if (left == 2) s += e.d1;
    }
    return s;
}

// The same with the crown part climbed on the crown's ladder (tree_prep.h: crown_ladder, indexed by rank;
// LDS on the device) instead of streamed: k edges of b's lineage = its first min(k, nb) lineage lengths,
// then k - nb ladder edges from its portal, three per 16-byte entry, in lineage order.
template <typename Lad>
ST_HD float stream_b_ladder(const float *__restrict__ lens, const Lad &lad, uint32_t off_b, uint32_t nb, uint32_t portal_rank,
                            float s, int32_t k)
{
    if (k <= (int32_t)nb) return stream_sum(lens + off_b, s, k);
    s = stream_sum(lens + off_b, s, (int32_t)nb);
    return ladder_climb_counted(lad, lad.place(portal_rank), (uint32_t)k - nb, s);
}

ST_HD PairResult pair_walk(const Node8 *__restrict__ nodes, const int32_t *__restrict__ depth,
                           const Stride3 *__restrict__ stride, int32_t a, int32_t b,
                           const uint64_t *__restrict__ rmq = nullptr, int64_t n_nodes = 0,
                           const LineageView &lin = LineageView())
{
    int32_t dm, da, db, m;
    NodeKey ka{}, kb{};
    if (lin.sums) {
        ka = lineage_key(lin, a);
        kb = lineage_key(lin, b);
        da = (int32_t)ka.depth;
        db = (int32_t)kb.depth;
    } else {
        da = depth[a];
        db = depth[b];
    }
    if (lin.sums && lin.crown_rmq && (ka.nb_rank >> 8) != (kb.nb_rank >> 8)) {
        const uint64_t e = crown_meet(lin.crown_rmq, lin.crown_nodes, ka.nb_rank >> 8, kb.nb_rank >> 8);
        dm = (int32_t)(e >> 32);
        m = (int32_t)(uint32_t)e;
    } else {
        m = pair_walk_mrca(nodes, depth, stride, a, b, &dm, rmq, n_nodes);
    }
    float s;
    if (lin.sums) s = lin.sums[(size_t)ka.off + (size_t)(da - dm)];
    else s = walk_sum(stride, 0.0f, a, da - dm);
    if (lin.sums && lin.lens) s = stream_b(lin, kb, s, db - dm);
    else s = walk_sum(stride, s, b, db - dm);
    PairResult r;
    r.dist = s;
    r.mrca = m;
    return r;
}

// ---- canopy family ---------------------------------------------------------
// Chains in registers (CAP > 0): every slot is added, unconditionally -- slots beyond the chain's length hold -0.0f
// (tree_prep.cpp writes it into the records, the kernels into chunks they do not load), and s + (-0.0f) == s bit for
// bit for every float s (+0, -0, denormals, infinities and NaNs included): one v_add_f32 per slot instead of add + select.
constexpr uint32_t kChainPad = 0x80000000u;
// `can` is the canopy table (LDS on the device), BFS-numbered: parent index <
// child index, so "move the larger index up" can never step past the meeting
// point.  `rec_*` are the understory records of a and b (global memory).
// s += D[0]; s += D[1]; ... s += D[nb - 1] for a chain read through a pointer INTO ITS RECORD (rec_b: word0, then the
// chain; 16-byte aligned): sixteen bytes per load instead of four, and chains of 8 slots and more -- they sit in
// records of at least 16 words, whose size is a multiple of 16 words from there on -- in blocks of four such loads
// that do not wait for each other (a 63-slot chain: at most four round trips instead of sixteen, or sixty-three).
ST_HD float chain_sum_ptr(const float *__restrict__ D, uint32_t nb, float s)
{
    if (nb <= 1) return nb ? s + D[0] : s;      // (one-slot records are 8 bytes: no 16-byte read there)
    const Quad *q = reinterpret_cast<const Quad *>(D - 1);      // {word0, D[0], D[1], D[2]}, {D[3] ...}, ...
    if (nb >= 8) {
        for (uint32_t base = 0; base <= nb; base += 16) {      // the block holds slots base - 1 .. base + 14
            const Quad a = q[0], b = q[1], c = q[2], d = q[3];
            q += 4;
            if (base) s += a.x;      // (base - 1 < nb: the loop condition)
            if (base + 0 < nb) s += a.y;
            if (base + 1 < nb) s += a.z;
            if (base + 2 < nb) s += a.w;
            if (base + 3 < nb) s += b.x;
            if (base + 4 < nb) s += b.y;
            if (base + 5 < nb) s += b.z;
            if (base + 6 < nb) s += b.w;
            if (base + 7 < nb) s += c.x;
            if (base + 8 < nb) s += c.y;
            if (base + 9 < nb) s += c.z;
            if (base + 10 < nb) s += c.w;
            if (base + 11 < nb) s += d.x;
            if (base + 12 < nb) s += d.y;
            if (base + 13 < nb) s += d.z;
            if (base + 14 < nb) s += d.w;
        }
        return s;
    }
    Quad v = q[0];
    s += v.y;
    if (nb > 1) s += v.z;
    if (nb > 2) s += v.w;
    if (nb > 3) {      // (nb <= 7: the record has a second quad)
        v = q[1];
        s += v.x;
        if (nb > 4) s += v.y;
        if (nb > 5) s += v.z;
        if (nb > 6) s += v.w;
    }
    return s;
}

// The same additions with the next block's four loads issued before the current block is added (one block ahead):
// a 127-slot chain is eight round trips to memory otherwise, each waited for in turn.  The hot path of the kernels
// that read long chains through a pointer (CAP == 0 below); sixteen more registers than the plain form, which the
// cold paths keep (pair_canopy_same_portal: its registers would otherwise set every kernel's allocation).
ST_HD float chain_sum_ptr_ahead(const float *__restrict__ D, uint32_t nb, float s)
{
    if (nb < 8) return chain_sum_ptr(D, nb, s);
    const Quad *q = reinterpret_cast<const Quad *>(D - 1);
    Quad a = q[0], b = q[1], c = q[2], d = q[3];
    for (uint32_t base = 0; base <= nb; base += 16) {      // the block holds slots base - 1 .. base + 14
        q += 4;
        Quad na = a, nb4 = b, nc = c, nd = d;
        if (base + 16 <= nb) { na = q[0]; nb4 = q[1]; nc = q[2]; nd = q[3]; }
        if (base) s += a.x;
        if (base + 0 < nb) s += a.y;
        if (base + 1 < nb) s += a.z;
        if (base + 2 < nb) s += a.w;
        if (base + 3 < nb) s += b.x;
        if (base + 4 < nb) s += b.y;
        if (base + 5 < nb) s += b.z;
        if (base + 6 < nb) s += b.w;
        if (base + 7 < nb) s += c.x;
        if (base + 8 < nb) s += c.y;
        if (base + 9 < nb) s += c.z;
        if (base + 10 < nb) s += c.w;
        if (base + 11 < nb) s += d.x;
        if (base + 12 < nb) s += d.y;
        if (base + 13 < nb) s += d.z;
        if (base + 14 < nb) s += d.w;
        a = na; b = nb4; c = nc; d = nd;
    }
    return s;
}

struct RecView {
    uint32_t portal;
    uint32_t nb;
    float pbot;
    const float *D;      // nb branch lengths, node itself first
    const int32_t *I;    // cap id slots, the chain's nb ids in the last ones (I[cap - 1] = the portal's child)
    int32_t cap;
};

// The three record tables (tree_prep.h) and the stride of rec_b / rec_i.
struct RecTables {
    const uint8_t *a;    // [n * 8]
    const uint8_t *b;    // [n * half]
    const uint8_t *i;    // [n * half]
    int32_t half;        // record_bytes / 2
};

ST_HD RecView rec_view(const RecTables &R, int64_t slot)
{
    const uint8_t *rb = R.b + slot * (int64_t)R.half;
    const uint8_t *ri = R.i + slot * (int64_t)R.half;
    RecView v;
    const uint32_t w0 = *reinterpret_cast<const uint32_t *>(rb);
    v.portal = w0 & 0xFFFFu;
    v.nb = w0 >> 16;
    v.pbot = *reinterpret_cast<const float *>(ri);
    v.D = reinterpret_cast<const float *>(rb + 4);
    v.I = reinterpret_cast<const int32_t *>(ri + 4);
    v.cap = R.half / 4 - 1;
    return v;
}

// Both lineages enter the canopy at different nodes: the MRCA is a canopy
// node.  s_in = a's understory total (pbot), D_b/nb_b = b's understory.
// CAP > 0: D_b is a register array of CAP floats (fully unrolled, predicated);
// CAP == 0: D_b is read through the pointer with a run-time trip count.
template <int CAP, typename CanPtr>
ST_HD PairResult pair_canopy_split(CanPtr can, const int32_t *__restrict__ canopy_id,
                                   uint32_t pa, float pbot_a, uint32_t pb,
                                   const float *D_b, uint32_t nb_b)
{
    float s = pbot_a;
    uint32_t u = pa, v = pb;
    while (u != v) {
        const bool up_a = u > v;
        const CanopyEntry e = can[up_a ? u : v];
        if (up_a) {
            s += e.dist;
            u = e.link & kCanopyParentMask;
        } else {
            v = e.link & kCanopyParentMask;
        }
    }
    const uint32_t mc = u;
    if (CAP > 0) {
#pragma unroll
        for (int i = 0; i < CAP; i++) s += D_b[i];      // (slots beyond nb_b hold -0.0f: kChainPad)
    } else {
        s = chain_sum_ptr_ahead(D_b, nb_b, s);
    }
    v = pb;
    while (v != mc) {
        const CanopyEntry e = can[v];
        s += e.dist;
        v = e.link & kCanopyParentMask;
    }
    PairResult r;
    r.dist = s;
    r.mrca = canopy_id[mc];
    return r;
}

// Meeting node of two canopy nodes from the sparse table (tree_prep.h): two 2-byte rank reads
// and two 4-byte table reads, no climbing.  Returns depth << 16 | canopy index.
// (from the ranks of the two canopy nodes: one 4-byte table read per half of the query)
ST_HD uint32_t canopy_meet_ranks(const uint32_t *__restrict__ rmq, int32_t n_canopy, uint32_t ra, uint32_t rb)
{
    const uint32_t l = ra < rb ? ra : rb, r = ra < rb ? rb : ra;
    const uint32_t len = r - l + 1;
    uint32_t k = 0;
    while ((2u << k) <= len) k++;                       // floor(log2(len)); len <= 16384
    const uint32_t e1 = rmq[(size_t)k * (size_t)n_canopy + l];
    const uint32_t e2 = rmq[(size_t)k * (size_t)n_canopy + (r + 1 - (1u << k))];
    return (e2 >> 16) < (e1 >> 16) ? e2 : e1;
}

// The same query on the 64-bit table (tree_prep.h: canopy_rmq64): depth << 32 | node id.
ST_HD uint64_t canopy_meet_ranks64(const uint64_t *__restrict__ rmq, int32_t n_canopy, uint32_t ra, uint32_t rb)
{
    const uint32_t l = ra < rb ? ra : rb, r = ra < rb ? rb : ra;
    const uint32_t len = r - l + 1;
    uint32_t k = 0;
    while ((2u << k) <= len) k++;
    const uint64_t e1 = rmq[(size_t)k * (size_t)n_canopy + l];
    const uint64_t e2 = rmq[(size_t)k * (size_t)n_canopy + (r + 1 - (1u << k))];
    return (e2 >> 32) < (e1 >> 32) ? e2 : e1;
}

ST_HD uint32_t canopy_meet(const uint16_t *__restrict__ pos, const uint32_t *__restrict__ rmq, int32_t n_canopy,
                           uint32_t pa, uint32_t pb)
{
    return canopy_meet_ranks(rmq, n_canopy, pos[pa], pos[pb]);
}

// Ladder form of pair_canopy_split (deep canopies).  `lad` is the ladder table (LDS on the
// device).  The meeting node `meet` (depth << 16 | canopy index, from canopy_meet) and the
// depths of the two portals are known, so both sums know how many edges they climb and add
// them three per 16-byte entry, in lineage order: a's canopy edges onto pbot_a, b's
// understory, b's canopy edges.
template <int CAP, typename Lad>
ST_HD PairResult pair_ladder_sums(const Lad &lad, const int32_t *__restrict__ canopy_id, uint32_t meet,
                                  uint32_t pa, uint32_t da, float pbot_a, uint32_t pb, uint32_t db,
                                  const float *D_b, uint32_t nb_b)
{
    const uint32_t dm = meet >> 16, stop = lad.place(meet & 0xFFFFu);
    const int32_t mrca = canopy_id[meet & 0xFFFFu];      // (asked for first: the read is in flight while both sides climb)
    float s = ladder_climb_to(lad, lad.place(pa), stop, da - dm, pbot_a);
    if (CAP > 0) {
#pragma unroll
        for (int i = 0; i < CAP; i++) s += D_b[i];      // (slots beyond nb_b hold -0.0f: kChainPad)
    } else {
        s = chain_sum_ptr_ahead(D_b, nb_b, s);
    }
    s = ladder_climb_to(lad, lad.place(pb), stop, db - dm, s);
    PairResult r;
    r.dist = s;
    r.mrca = mrca;
    return r;
}

// b's side of a pair whose a side came from the lineage-sum table (tree_prep.h): `s_a` is the
// reference's accumulator after a's edges up to the meeting node; b's understory and its `kb`
// canopy edges continue it, in lineage order, three canopy edges per 16-byte entry.
template <int CAP, typename Lad>
ST_HD float ladder_sum_b(const Lad &lad, uint32_t kb, float s_a, uint32_t pb, const float *D_b, uint32_t nb_b)
{
    float s = s_a;
    if (CAP > 0) {
#pragma unroll
        for (int i = 0; i < CAP; i++) s += D_b[i];      // (slots beyond nb_b hold -0.0f: kChainPad)
    } else {
        s = chain_sum_ptr_ahead(D_b, nb_b, s);
    }
    return ladder_climb_counted(lad, lad.place(pb), kb, s);
}

// Ladder form of pair_canopy_split for trees whose ids are NOT an in-order numbering (no
// sparse table): lock-step search for the meeting node.  `lad` is the ladder image, `cdepth` the
// canopy depths, `canopy` the 8-byte entries (parents; global memory on the device).  Phase 1 finds
// the meeting node with integer work only: the deeper lineage is lifted to the other's depth, then
// both climb in lock step three levels at a time while their third ancestors differ, one level at a
// time once they agree (or lie above the root).  Phase 2 then knows where each side ends and adds its
// edges, three per entry, in lineage order: a's canopy edges onto pbot_a, b's understory, b's canopy edges.
template <int CAP, typename Lad, typename DepthPtr>
ST_HD PairResult pair_ladder_split(const Lad &lad, DepthPtr cdepth, const CanopyEntry *__restrict__ canopy,
                                   const int32_t *__restrict__ canopy_id,
                                   uint32_t pa, float pbot_a, uint32_t pb,
                                   const float *D_b, uint32_t nb_b)
{
    uint32_t u = pa, v = pb;
    const uint32_t da = cdepth[pa], db = cdepth[pb];
    uint32_t du = da, dv = db;
    while (du > dv) {
        if (du - dv >= 3) { u = lad.index(lad.at(lad.place(u)).link); du -= 3; }      // (du >= 3: the link is a place)
        else { u = canopy[u].link & kCanopyParentMask; du -= 1; }
    }
    while (dv > du) {
        if (dv - du >= 3) { v = lad.index(lad.at(lad.place(v)).link); dv -= 3; }
        else { v = canopy[v].link & kCanopyParentMask; dv -= 1; }
    }
    while (u != v) {
        const uint32_t lu = lad.at(lad.place(u)).link, lv = lad.at(lad.place(v)).link;
        if (lu != lv) { u = lad.index(lu); v = lad.index(lv); du -= 3; }      // (two places; both above the root compare equal)
        else { u = canopy[u].link & kCanopyParentMask; v = canopy[v].link & kCanopyParentMask; du -= 1; }
    }
    const uint32_t stop = lad.place(u);
    float s = ladder_climb_to(lad, lad.place(pa), stop, da - du, pbot_a);
    if (CAP > 0) {
#pragma unroll
        for (int i = 0; i < CAP; i++) s += D_b[i];      // (slots beyond nb_b hold -0.0f: kChainPad)
    } else {
        s = chain_sum_ptr_ahead(D_b, nb_b, s);
    }
    s = ladder_climb_to(lad, lad.place(pb), stop, db - du, s);
    PairResult r;
    r.dist = s;
    r.mrca = canopy_id[u];
    return r;
}

// Both lineages enter the canopy at the same node: the MRCA is that portal or
// lies in the understory.  The two id chains are compared from the portal end.
// (four id slots per round trip: the slots end on the record's last 16 bytes, so chunk q from the end is one aligned
// 16-byte read of each record -- pairs of nearby leaves share nearly their whole chains, and one dependent pair of
// 4-byte loads per shared level was what the scalar kernels spent on them: nj.tree, leaves within 8 of each other,
// 9.4e9 pairs/s where the tile-sorted kernel's register form did 1.85e10)
struct alignas(16) IdQuad { int32_t x, y, z, w; };

ST_HD PairResult pair_canopy_same_portal(const int32_t *__restrict__ canopy_id,
                                         const RecView &A, const RecView &B)
{
    uint32_t c = 0;
    int32_t last = 0;      // the id in slot cap - c (the deepest common node so far)
    const uint32_t lim = A.nb < B.nb ? A.nb : B.nb;
    if (A.cap >= 3) {
        const IdQuad *qa = reinterpret_cast<const IdQuad *>(A.I + A.cap), *qb = reinterpret_cast<const IdQuad *>(B.I + B.cap);
        // chunk q = slots cap-1-4q (.w) down to cap-4-4q (.x); its first `room` slots count (the shorter chain ends there;
        // the record's first word is pbot).  Shared ancestry is a run from the portal end, so "chunk q matches whole" is
        // monotone in q: the first chunk decides most pairs that are not close relatives, a bisection over the others
        // finds the chunk where the run ends in log2 steps (31-slot chains: 4 dependent round trips at most, not 8)
        auto matches = [&](uint32_t q, IdQuad &x, uint32_t &room) {
            x = qa[-1 - (int32_t)q];
            const IdQuad y = qb[-1 - (int32_t)q];
            room = lim - 4 * q < 4 ? lim - 4 * q : 4;
            const uint32_t m = x.w != y.w ? 0u : x.z != y.z ? 1u : x.y != y.y ? 2u : x.x != y.x ? 3u : 4u;
            return m < room ? m : room;
        };
        const uint32_t chunks = (lim + 3) / 4;
        IdQuad x{0, 0, 0, 0}, xe{0, 0, 0, 0};      // xe: the last chunk seen that ends the run
        uint32_t room = 0, m = 0, lo = 0, hi = chunks, me = 0, qe = chunks;
        if (chunks) {
            m = matches(0, x, room);
            if (m < 4 || chunks == 1) { qe = 0; me = m; xe = x; hi = 0; }
            else lo = 1;
        }
        while (lo < hi) {      // invariant: chunks below lo match whole; chunk qe (= hi, if hi < chunks) ends the run with me slots
            const uint32_t mid = (lo + hi) / 2;
            m = matches(mid, x, room);
            if (m == 4) lo = mid + 1;
            else { hi = mid; qe = mid; me = m; xe = x; }
        }
        if (qe < chunks) {
            c = 4 * qe + me;
            if (me) last = me == 1 ? xe.w : me == 2 ? xe.z : me == 3 ? xe.y : xe.x;
            else if (qe) last = qa[-(int32_t)qe].x;      // the run ends exactly between two chunks: the id is the previous chunk's last slot
        } else if (chunks) {      // every chunk matched whole (lim is a multiple of four)
            c = lim;
            last = qa[-(int32_t)chunks].x;
        }
    } else {      // 8-byte records: one slot
        while (c < lim && A.I[A.cap - 1 - (int32_t)c] == B.I[B.cap - 1 - (int32_t)c]) {
            last = A.I[A.cap - 1 - (int32_t)c];
            c++;
        }
    }
    const uint32_t ia = A.nb - c, ib = B.nb - c;
    float s = chain_sum_ptr(A.D, ia, 0.0f);
    s = chain_sum_ptr(B.D, ib, s);
    PairResult r;
    r.dist = s;
    r.mrca = c ? last : canopy_id[A.portal];
    return r;
}

// The same for chains of at most CAP <= 15 slots, without a load that depends on a comparison: the four half
// records are read whole (independent 16-byte loads), the ids compared slot by slot from the portal end, the
// sums are predicated adds -- every index is a compile-time constant, so the chains live in registers.  Pairs of
// nearby leaves (sister taxa, nearest-neighbour candidates) all come this way: 2e7 pairs within 8 leaves of each
// other on the 2^20-leaf tree 6.4e9 pairs/s with the loop form above (serial loads, divergent trip counts).
struct alignas(16) Words4 {
    uint32_t x, y, z, w;
};

template <int CAP>
ST_HD void load_half_record(const uint8_t *p, uint32_t (&w)[CAP + 1])
{
    if (CAP == 1) {
        w[0] = reinterpret_cast<const uint32_t *>(p)[0];
        w[1] = reinterpret_cast<const uint32_t *>(p)[1];
    } else {
#pragma unroll
        for (int q = 0; q < (CAP + 1) / 4; q++) {
            const Words4 v = reinterpret_cast<const Words4 *>(p)[q];
            w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    }
}

ST_HD float word_as_float(uint32_t u)
{
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}

// Common part: length of the common portal-end run of two id chains (ids in words 1..CAP, slot CAP - 1 - k = the
// k-th node below the portal) and the last id of that run.
template <int CAP>
ST_HD uint32_t common_chain_run(const uint32_t (&IA)[CAP + 1], const uint32_t (&IB)[CAP + 1], uint32_t na, uint32_t nb, uint32_t &last)
{
    // first slot from the portal end at which the chains differ (selects with short-lived conditions: a chain of
    // boolean flags carried through the loop kept a 64-bit lane mask alive per step and spilled scalar registers)
    uint32_t first = (uint32_t)CAP;
#pragma unroll
    for (int k = CAP - 1; k >= 0; k--) first = IA[CAP - k] != IB[CAP - k] ? (uint32_t)k : first;
    const uint32_t shorter = na < nb ? na : nb;
    const uint32_t c = first < shorter ? first : shorter;
    last = 0;
#pragma unroll
    for (int k = 0; k < CAP; k++) last = c == (uint32_t)(k + 1) ? IA[CAP - k] : last;
    return c;
}

template <int CAP>
ST_HD PairResult pair_same_portal_regs(const int32_t *__restrict__ canopy_id, const RecTables &R, int64_t sa, int64_t sb)
{
    static_assert(CAP == 1 || CAP == 3 || CAP == 7 || CAP == 15 || CAP == 31, "chains in registers");
    uint32_t c, last, na, nb, portal;
    uint32_t DA[CAP + 1], DB[CAP + 1];
    if (CAP <= 15) {      // everything in one round trip
        uint32_t IA[CAP + 1], IB[CAP + 1];
        load_half_record<CAP>(R.i + sa * (int64_t)R.half, IA);
        load_half_record<CAP>(R.i + sb * (int64_t)R.half, IB);
        load_half_record<CAP>(R.b + sa * (int64_t)R.half, DA);
        load_half_record<CAP>(R.b + sb * (int64_t)R.half, DB);
        na = DA[0] >> 16;
        nb = DB[0] >> 16;
        portal = DA[0] & 0xFFFFu;
        c = common_chain_run<CAP>(IA, IB, na, nb, last);
    } else {              // 31 slots: the ids first, the lengths after them (64 registers at a time, not 128)
        const uint32_t wa = *reinterpret_cast<const uint32_t *>(R.a + sa * 8), wb = *reinterpret_cast<const uint32_t *>(R.a + sb * 8);
        na = wa >> 16;
        nb = wb >> 16;
        portal = wa & 0xFFFFu;
        {
            uint32_t IA[CAP + 1], IB[CAP + 1];
            load_half_record<CAP>(R.i + sa * (int64_t)R.half, IA);
            load_half_record<CAP>(R.i + sb * (int64_t)R.half, IB);
            c = common_chain_run<CAP>(IA, IB, na, nb, last);
        }
        asm volatile("" ::: "memory");      // (keeps the second pair of loads behind the comparison)
        load_half_record<CAP>(R.b + sa * (int64_t)R.half, DA);
        load_half_record<CAP>(R.b + sb * (int64_t)R.half, DB);
    }
    const uint32_t ia = na - c, ib = nb - c;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < CAP; i++) {
        const float t = s + word_as_float(DA[i + 1]);
        s = (uint32_t)i < ia ? t : s;
    }
#pragma unroll
    for (int i = 0; i < CAP; i++) {
        const float t = s + word_as_float(DB[i + 1]);
        s = (uint32_t)i < ib ? t : s;
    }
    PairResult r;
    r.dist = s;
    r.mrca = c ? (int32_t)last : canopy_id[portal];
    return r;
}

// The MRCA id alone (k_mrca_ranks): the two id chains and the two 8-byte a entries.
template <int CAP>
ST_HD int32_t mrca_same_portal_regs(const int32_t *__restrict__ canopy_id, const RecTables &R, int64_t sa, int64_t sb)
{
    uint32_t IA[CAP + 1], IB[CAP + 1];
    load_half_record<CAP>(R.i + sa * (int64_t)R.half, IA);
    load_half_record<CAP>(R.i + sb * (int64_t)R.half, IB);
    const uint32_t wa = *reinterpret_cast<const uint32_t *>(R.a + sa * 8), wb = *reinterpret_cast<const uint32_t *>(R.a + sb * 8);
    uint32_t last;
    const uint32_t c = common_chain_run<CAP>(IA, IB, wa >> 16, wb >> 16, last);
    return c ? (int32_t)last : canopy_id[wa & 0xFFFFu];
}

}  // namespace st
