// kernels_canopy_refill.h -- k_canopy_refill: the scalar ladder kernel with LANE-granular refill (launch_canopy.hip).
//
// k_canopy_ladder gives every lane one pair per pass; a wave then climbs for as long as the longest of its 64 a-side
// climbs, adds the chains, and climbs for as long as the longest of its 64 b-side climbs (nj.tree: 129 rounds per pass
// where a lane's mean is 35).  Here a wave takes K * 64 pairs at a time and its lanes draw CLIMBS from that pool: a
// lane whose climb has ended delivers the sum and takes the next climb nobody has started.  What makes that
// affordable on this ISA:
//   * global loads stay at wave-uniform points (vmcnt counts a wave's loads in order, so a lane cannot wait for "its"
//     load while others' are in flight): the KEY phase reads ids, records' first words, ranks, sparse-table entries
//     and depths of all K * 64 pairs together -- K loads in flight per lane and round trip instead of one;
//   * a climb is two words -- the running sum, and {entry | meeting entry << 14 | edges % 3 << 28} -- kept in 2 K
//     registers per lane; the lane that takes climb number c reads them from lane c % 64 of set c / 64 with
//     ds_bpermute (set index wave-uniform: climbs are handed out in order);
//   * sums come back through 4 bytes of LDS per pair behind the image (K * 256 bytes per wave);
//   * b's chain cannot ride along (31 registers per pair), and a's sum must be complete before its first slot is
//     added (pyx:934-942: one float32 accumulator, a's edges, then b's): so a pair is two climbs with a wave-uniform
//     CHAIN phase between them that reads every b record (second read: L2) and adds all K * 64 chains, no lane idle;
//   * the hand-over block is only entered when at least `threshold` lanes wait (or nothing climbs): its ~30
//     instructions are paid once per several rounds, not once per round.
// Results are bit-identical to k_canopy_ladder: the same float32 adds in the same order per pair.
#pragma once
#include "kernels_canopy.h"

namespace st {

// one climb, packed: canopy index of the entry the climb starts at (14 bits: kDeepCanopyNodes < 16384), of the
// meeting node's entry, the edges % 3 left after the last whole round, and (b side) "no chain" for pairs whose
// distance is already final (shared portal, ids out of range)
constexpr uint32_t kRefillNoChain = 1u << 30;
ST_HD uint32_t refill_pack(uint32_t at_index, uint32_t stop_index, uint32_t edges)
{
    return at_index | (stop_index << 14) | ((edges - 3u * ((edges * 0xAAABu) >> 17)) << 28);      // edges % 3 (edges < 2^16)
}

#if defined(__HIPCC__)
struct RefillClimb {
    float s;
    uint32_t at, stop, left, home;
};

__device__ __forceinline__ void refill_decode(uint32_t lo, uint32_t hi, uint32_t base, RefillClimb &c)
{
    c.s = __uint_as_float(lo);
    c.at = base + ((hi & 0x3FFFu) << 4);
    c.stop = base + (((hi >> 14) & 0x3FFFu) << 4);
    c.left = (hi >> 28) & 3u;
}

// All K * 64 climbs of (lo, hi), each lane drawing the next one when its own has ended; climb number c leaves its sum
// in res[c].  Every lane of the wave is here (converged), and leaves together.
template <int K>
__device__ __forceinline__ void refill_climbs(const uint32_t (&lo)[K], const uint32_t (&hi)[K], uint32_t image_base,
                                              float *res, int threshold)
{
    const unsigned lane = threadIdx.x & 63u;
    RefillClimb c;
    refill_decode(lo[0], hi[0], image_base, c);
    c.home = lane;
    uint32_t next = 64;      // climbs handed out so far (wave-uniform)
    uint32_t out = 0;        // lanes in state 3 (wave-uniform)
    // lane state: 0 = climbing, 1 = climb ended, sum not delivered, 2 = delivered, wants a climb, 3 = nothing left to take
    uint32_t st = 0;
    LadderEntry e{0.0f, 0.0f, 0.0f, 0u};
    for (;;) {
        // ---- rounds, until `threshold` lanes wait or nothing climbs: while more than `limit` lanes climb ----
        const int32_t room_for_waiting = 64 - (int32_t)out - threshold;
        const uint32_t limit = room_for_waiting > 0 ? (uint32_t)room_for_waiting : 0u;
        uint32_t climbing;
        do {
            if (st == 0) {
                const LadderWords w = *(const __attribute__((address_space(3))) LadderWords *)(uintptr_t)c.at;
                e = LadderEntry{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), w.w};
                if ((int32_t)e.link >= (int32_t)c.stop) {
                    c.s += e.d0; c.s += e.d1; c.s += e.d2;
                    c.at = e.link;
                } else {
                    st = 1;
                }
            }
            climbing = (uint32_t)__builtin_popcountll(__ballot(st == 0));
        } while (climbing > limit);
        const uint32_t waiting = 64u - out - climbing;
        if (waiting == 0) break;      // (nothing climbs, nobody waits: every sum delivered, nothing left)
        // ---- hand-over (every lane is here) ----
        if (st == 1) {      // deliver: the edges % 3 left come from the entry the climb ended on
            if (c.left) {
                c.s += e.d0;
                if (c.left == 2) c.s += e.d1;
            }
            res[c.home] = c.s;
            st = 2;
        }
        const uint32_t q0 = next >> 6;      // the set the next climbs come from (uniform)
        if (q0 >= (uint32_t)K) {
            if (st == 2) st = 3;
            out += waiting;
            continue;
        }
        const unsigned long long want = __ballot(st == 2);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
        const uint32_t room = 64u - (next & 63u);
        const uint32_t take = waiting < room ? waiting : room;
        const uint32_t number = next + rank;
        uint32_t nlo = 0, nhi = 0;
#pragma unroll
        for (int q = 0; q < K; q++) {
            if (q0 == (uint32_t)q) {      // (uniform; every lane executes the permutes: a lane that is masked off would supply nothing)
                nlo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((number & 63u) << 2), (int)lo[q]);
                nhi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((number & 63u) << 2), (int)hi[q]);
            }
        }
        if (st == 2 && rank < take) {
            refill_decode(nlo, nhi, image_base, c);
            c.home = number;
            st = 0;
        }
        next += take;
    }
}

// The K climbs of every lane in lock step, K LDS reads in flight per round: nothing is dealt again -- a round lasts
// while ANY of the wave's K * 64 climbs has one left -- but a wave's dependent LDS round trips per pair drop K-fold
// (with 16 waves per CU it is their latency, not LDS bandwidth, that a one-workgroup image leaves exposed), and a set
// whose 64 climbs have all ended is skipped.  Sums come back in lo[].
template <int K>
__device__ __forceinline__ void lockstep_climbs(uint32_t (&lo)[K], const uint32_t (&hi)[K], uint32_t image_base)
{
    float s[K];
    uint32_t stop[K];
    LadderEntry e[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        RefillClimb c;
        refill_decode(lo[j], hi[j], image_base, c);
        s[j] = c.s;
        stop[j] = c.stop;
        const LadderWords w = *(const __attribute__((address_space(3))) LadderWords *)(uintptr_t)c.at;
        e[j] = LadderEntry{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), w.w};
    }
    for (;;) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < K; j++) {
            if ((int32_t)e[j].link >= (int32_t)stop[j]) {
                s[j] += e[j].d0; s[j] += e[j].d1; s[j] += e[j].d2;
                const LadderWords w = *(const __attribute__((address_space(3))) LadderWords *)(uintptr_t)e[j].link;
                e[j] = LadderEntry{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), w.w};
                any = true;
            }
        }
        if (!__any(any)) break;
    }
#pragma unroll
    for (int j = 0; j < K; j++) {      // the edges % 3 left come from the entry the climb ended on
        const uint32_t left = (hi[j] >> 28) & 3u;
        if (left) {
            s[j] += e[j].d0;
            if (left == 2) s[j] += e[j].d1;
        }
        lo[j] = __float_as_uint(s[j]);
    }
}

// K: pairs per lane and visit (sets); CAP as in k_canopy_ladder.  LDS: the ladder image, then K * 64 floats per wave.
// REFILL: climbs drawn lane by lane (refill_climbs), else K climbs per lane in lock step (lockstep_climbs: no result slots).
template <int CAP, int K, typename Src, bool REFILL>
__global__ __launch_bounds__(kCanopyBlock, 4) void k_canopy_refill(CanopyParams P, Src src, long long n, DistSink out_d, MrcaSink out_m,
                                                                   Fault *fault, int threshold)
{
    static_assert(CAP == 0 || CAP == 15 || CAP == 31 || CAP == 63, "as k_canopy_ladder");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    stage_ladder(P, lds_raw);
    const LdsLadder lad(lds_raw);
    const int rec_bytes = CAP > 0 ? 8 * (CAP + 1) : P.rec_bytes;
    const bool parity = P.parity != 0;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float *res = reinterpret_cast<float *>(lds_raw + ladder_image_bytes(P.canopy_nodes)) + (size_t)wave * (K * 64);
    const long long per_visit = (long long)K * 64;
    const long long visits = (n + per_visit - 1) / per_visit;
    const long long waves = (long long)gridDim.x * (kCanopyBlock / 64);
    for (long long visit = (long long)blockIdx.x * (kCanopyBlock / 64) + wave; visit < visits; visit += waves) {
        const long long base = visit * per_visit;
        uint32_t lo[K], hi[K], hb[K], sb32[K];
        // ---- key phase: everything of all K pairs of this lane, loads of one kind issued together ----
        long long ia[K], ib[K];
        bool live[K], valid[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            const long long i = base + (long long)j * 64 + lane;
            live[j] = i < n;
            src.load(live[j] ? i : n - 1, ia[j], ib[j]);
        }
        uint32_t wa[K], wb[K], sa32[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            valid[j] = (unsigned long long)ia[j] < (unsigned long long)P.n_nodes && (unsigned long long)ib[j] < (unsigned long long)P.n_nodes;
            if (!valid[j] && live[j]) record_fault(fault, ia[j], ib[j], P.n_nodes);
            sa32[j] = (uint32_t)record_slot(valid[j] ? ia[j] : 0, parity, P.n_leaves);
            sb32[j] = (uint32_t)record_slot(valid[j] ? ib[j] : 0, parity, P.n_leaves);
            const uint2 va = reinterpret_cast<const uint2 *>(P.rec_a)[sa32[j]];
            wa[j] = va.x;
            lo[j] = va.y;      // pbot of a: where a's climb starts from
            wb[j] = *reinterpret_cast<const uint32_t *>(P.rec_b + (size_t)sb32[j] * (size_t)(rec_bytes / 2));
        }
        uint32_t ra[K], rb[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            ra[j] = P.cpos[wa[j] & 0xFFFFu];
            rb[j] = P.cpos[wb[j] & 0xFFFFu];
        }
        uint32_t meet[K];
#pragma unroll
        for (int j = 0; j < K; j++) meet[j] = canopy_meet_ranks(P.rmq, P.canopy_nodes, ra[j], rb[j]);
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t pa = wa[j] & 0xFFFFu, pb = wb[j] & 0xFFFFu;
            const uint32_t mi = meet[j] & 0xFFFFu, dm = meet[j] >> 16;
            const uint32_t da = P.cdepth[pa], db = P.cdepth[pb];
            int m = P.canopy_id[mi];
            hi[j] = refill_pack(pa, mi, da - dm);
            hb[j] = refill_pack(pb, mi, db - dm);
            if (!valid[j] || pa == pb) {      // rare: the distance is final here; both climbs end at once (the root's entry, no edges)
                PairResult r;
                r.dist = __builtin_nanf("");
                r.mrca = -1;
                if (valid[j]) {
                    if (!P.rec_i) {
                        r = same_portal_by_walk(P, sa32[j], sb32[j]);
                    } else {
                        const RecTables R{P.rec_a, P.rec_b, P.rec_i, rec_bytes / 2};
                        r = pair_canopy_same_portal(P.canopy_id, rec_view(R, sa32[j]), rec_view(R, sb32[j]));
                    }
                }
                lo[j] = __float_as_uint(r.dist);
                m = r.mrca;
                hi[j] = 0;
                hb[j] = kRefillNoChain;
            }
            store_mrca_wave(out_m, base + (long long)j * 64 + lane, m, live[j]);
        }
        // ---- a's climbs ----
        if (REFILL) refill_climbs<K>(lo, hi, lad.base, res, threshold);
        else lockstep_climbs<K>(lo, hi, lad.base);
        // ---- chain phase: a's sum, then every slot of b's chain, in order ----
#pragma unroll
        for (int j = 0; j < K; j++) {
            PairRecs<CAP> L;
            L.rb = P.rec_b + (size_t)sb32[j] * (size_t)(rec_bytes / 2);
            load_rec_b_lazy<CAP>(L);
            const float s_a = REFILL ? res[j * 64 + lane] : __uint_as_float(lo[j]);
            float s = s_a;
            if (CAP > 0) {
#pragma unroll
                for (int q = 0; q < CAP; q++) s += L.Db[q];      // (slots beyond the chain hold -0.0f: kChainPad)
            } else {
                s = chain_sum_ptr_ahead(L.chain(), L.wb >> 16, s);
            }
            lo[j] = __float_as_uint((hb[j] & kRefillNoChain) ? s_a : s);
        }
        // ---- b's climbs ----
        if (REFILL) refill_climbs<K>(lo, hb, lad.base, res, threshold);
        else lockstep_climbs<K>(lo, hb, lad.base);
#pragma unroll
        for (int j = 0; j < K; j++) {
            const long long i = base + (long long)j * 64 + lane;
            if (live[j]) store_dist(out_d, i, REFILL ? res[j * 64 + lane] : __uint_as_float(lo[j]));
        }
    }
}
#endif

}  // namespace st
