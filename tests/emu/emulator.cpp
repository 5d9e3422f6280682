// emulator.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Runs the product's host-side table construction (suchtree_amd/csrc/tree_prep.cpp)
// and the product's per-pair device functions (suchtree_amd/csrc/pair_math.h),
// compiled for the host, one pair at a time with a plain array standing in for
// LDS.  The CPU test-suite uses it to check the canopy / understory tables and
// the pair arithmetic against the oracle without a GPU.  It is never loaded by
// suchtree_amd: the product computes on the GPU or fails.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../suchtree_amd/csrc/pair_math.h"
#include "../../suchtree_amd/csrc/tree_prep.h"

using namespace st;

extern "C" {

struct emu_info {
    int64_t n_leaves;
    int32_t root, depth, has_canopy, canopy_nodes, understory_max, record_bytes, parity, pad;
};

static std::string g_err;
const char *emu_last_error() { return g_err.c_str(); }

// strategy: 1 walk, 2 canopy.  Returns 0 ok, 1 bad tree, 2 canopy not admitted.
int emu_distances(const int32_t *parent, const float *distance, int64_t n_nodes, int strategy,
                  const int64_t *pairs, int64_t n, double *out_d, int32_t *out_m, emu_info *info)
{
    TreeTables T;
    if (!prepare_basic(parent, distance, n_nodes, T, g_err)) return 1;
    bool canopy = false;
    bool lineage = false, ranks = false, leaf_blocks = false;
    // the walk family's lineage tables, built both ways (offsets as the deep-canopy path assigns them,
    // and the walk-only path), each with crowns of several sizes
    struct WalkTables {
        std::vector<float> sums, lens;
        std::vector<uint32_t> node_rec;
        std::vector<uint64_t> crown_rmq;
        int32_t crown_nodes = 0;
        std::vector<LadderEntry> ladder;
    };
    std::vector<WalkTables> walk_tables;
    if (strategy == 1) {
        for (const int64_t hot : {(int64_t)2048, (int64_t)4 << 20}) {
            TreeTables C = T;
            // (the larger budget also asks for a ladder over a crown of at most 300 nodes)
            const int ladder_nodes = hot > 4096 ? 300 : 0;
            if (prepare_canopy(parent, distance, C) && prepare_lineage_sums(C, (int64_t)1 << 27) && prepare_walk_crown(C, hot, ladder_nodes))
                walk_tables.push_back({C.lineage_sum, C.lineage_len, C.lineage_node_rec, C.crown_rmq, C.crown_nodes, C.crown_ladder});
            TreeTables W = T;
            if (prepare_walk_lineage(W, (int64_t)1 << 27) && prepare_walk_crown(W, hot, ladder_nodes))
                walk_tables.push_back({W.lineage_sum, W.lineage_len, W.lineage_node_rec, W.crown_rmq, W.crown_nodes, W.crown_ladder});
        }
    }
    if (strategy == 2) {
        canopy = prepare_canopy(parent, distance, T);
        if (!canopy) { g_err = "canopy not admitted"; return 2; }
        leaf_blocks = prepare_leaf_blocks(T, 8192) || prepare_leaf_blocks(T, 1 << 30);   // (the second form: one leaf per block, always uniform)
        if (leaf_blocks) (void)prepare_cherries(T);
        lineage = prepare_lineage_sums(T, (int64_t)1 << 27);   // (in-order ids only)
        ranks = prepare_rank_table(T);                         // (in-order ids only)
    }
    if (info) {
        info->n_leaves = T.n_leaves;
        info->root = T.root;
        info->depth = T.tree_depth;
        info->has_canopy = canopy;
        info->canopy_nodes = T.canopy_nodes;
        info->understory_max = T.understory_max;
        info->record_bytes = T.record_bytes;
        info->parity = T.parity_layout;
    }
    for (int64_t i = 0; i < n; i++) {
        const int64_t a = pairs[2 * i], b = pairs[2 * i + 1];
        PairResult r;
        if (!canopy) {
            r = pair_walk(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b);
            if (!T.tree_rmq.empty()) {     // the sparse-table form of the meeting node must agree
                const PairResult q = pair_walk(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b,
                                               T.tree_rmq.data(), T.n);
                if (q.mrca != r.mrca || std::memcmp(&q.dist, &r.dist, 4) != 0) {
                    g_err = "walk: sparse-table meeting node disagrees with the lock-step search";
                    return 7;
                }
            }
            // every form of the lineage tables: a's side from the sums; b's side climbed, streamed from its
            // own block, or streamed from its own block and then its portal's; the meeting node by
            // climbing, from the whole-tree table, or from the crown's table
            for (const WalkTables &W : walk_tables) {
                if (!W.ladder.empty()) {      // k_walk_sorted's ladder mode: a's side from the sums, b's crown part climbed on the ladder
                    LineageView lin;
                    lin.node_rec = W.node_rec.data();
                    const NodeKey ka = lineage_key(lin, (int32_t)a), kb = lineage_key(lin, (int32_t)b);
                    int32_t dm;
                    (void)pair_walk_mrca(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b, &dm);
                    float s = W.sums[(size_t)ka.off + (size_t)((int32_t)ka.depth - dm)];
                    s = stream_b_ladder(W.lens.data(), PtrLadder{W.ladder.data()}, kb.off, kb.nb_rank & 0xFFu, kb.nb_rank >> 8, s, (int32_t)kb.depth - dm);
                    if (std::memcmp(&s, &r.dist, 4) != 0) {
                        g_err = "walk: ladder form of the crown part disagrees with the climb";
                        return 16;
                    }
                }
                for (int form = 0; form < 16; form++) {
                    const bool use_rmq = form & 1, use_lens = form & 2, shared = form & 4, crown = form & 8;
                    if (use_rmq && T.tree_rmq.empty()) continue;
                    if (crown && W.crown_rmq.empty()) continue;
                    if (shared && !use_lens) continue;
                    LineageView lin;
                    lin.node_rec = W.node_rec.data();
                    lin.sums = W.sums.data();
                    lin.lens = use_lens ? W.lens.data() : nullptr;
                    lin.shared_blocks = shared;
                    if (crown) { lin.crown_rmq = W.crown_rmq.data(); lin.crown_nodes = W.crown_nodes; }
                    const PairResult q = pair_walk(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b,
                                                   use_rmq ? T.tree_rmq.data() : nullptr, T.n, lin);
                    if (q.mrca != r.mrca || std::memcmp(&q.dist, &r.dist, 4) != 0) {
                        g_err = "walk: lineage-table form " + std::to_string(form) + " disagrees with the climb";
                        return 8;
                    }
                }
            }
            if (out_m && !out_d) r.mrca = pair_walk_mrca(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b);
        } else {
            const int64_t sa = record_slot(a, T.parity_layout, T.n_leaves);
            const int64_t sb = record_slot(b, T.parity_layout, T.n_leaves);
            const RecTables R{T.rec_a.data(), T.rec_b.data(), T.rec_i.data(), T.record_bytes / 2};
            const RecView A = rec_view(R, sa);
            const RecView B = rec_view(R, sb);
            // the a side of a kernel reads only rec_a: {word0, pbot}
            uint32_t wa;
            float pbot_a;
            std::memcpy(&wa, R.a + sa * 8, 4);
            std::memcpy(&pbot_a, R.a + sa * 8 + 4, 4);
            if (wa != (A.portal | (A.nb << 16)) || std::memcmp(&pbot_a, &A.pbot, 4) != 0) {
                g_err = "rec_a disagrees with rec_b / rec_i";
                return 3;
            }
            if (leaf_blocks) {     // the four-byte a side: pbot from rec_a4, the portal from the block table of leaf slots
                if (std::memcmp(&T.rec_a4[(size_t)sa], &pbot_a, 4) != 0) { g_err = "rec_a4 disagrees with rec_a"; return 14; }
                if (sa < T.n_leaves) {
                    const uint16_t e = T.leaf_block_portal[(size_t)(sa >> T.leaf_block_shift)];
                    if (e != kLeafBlockMixed && (uint32_t)(e & kLeafBlockPortalMask) != (wa & 0xFFFFu)) { g_err = "leaf block table names another portal"; return 15; }
                }
                // the cherry record of b (a leaf of a block of sibling pairs): the same chain as rec_b's, the portal from the block table
                if (!T.rec_c.empty() && sb < T.n_leaves) {
                    const uint16_t e = T.leaf_block_portal[(size_t)(sb >> T.leaf_block_shift)];
                    if (e != kLeafBlockMixed && (e & kLeafBlockCherries)) {
                        const size_t half = (size_t)T.record_bytes / 2;
                        const uint8_t *q = T.rec_c.data() + (size_t)(sb >> 1) * half;
                        std::vector<uint8_t> rebuilt(half);
                        const uint32_t w0 = (uint32_t)(e & kLeafBlockPortalMask) | (B.nb << 16);
                        std::memcpy(rebuilt.data(), &w0, 4);
                        std::memcpy(rebuilt.data() + 4, q + ((sb & 1) ? 4 : 0), 4);
                        if (half > 8) std::memcpy(rebuilt.data() + 8, q + 8, half - 8);
                        if (std::memcmp(rebuilt.data(), T.rec_b.data() + (size_t)sb * half, half) != 0) { g_err = "cherry record disagrees with rec_b"; return 22; }
                    }
                }
            }
            if (ranks && A.portal != B.portal) {     // the MRCA-only kernel's form: two rank reads, two table entries
                const uint64_t m64 = canopy_meet_ranks64(T.canopy_rmq64.data(), T.canopy_nodes, T.rec_r[(size_t)sa] & 0xFFFFu,
                                                         T.rec_r[(size_t)sb] & 0xFFFFu);
                const PairResult ref = pair_canopy_split<0>(T.canopy.data(), T.canopy_id.data(), wa & 0xFFFFu, pbot_a, B.portal, B.D, B.nb);
                if ((int32_t)(uint32_t)m64 != ref.mrca || (T.rec_r[(size_t)sa] >> 16) != (uint32_t)T.depth[(size_t)a]) {
                    g_err = "rank-table MRCA disagrees with the canopy climb";
                    return 10;
                }
            }
            if (ranks && A.portal == B.portal && (T.rec_r[(size_t)sa] & 0xFFFFu) != (T.rec_r[(size_t)sb] & 0xFFFFu)) {
                g_err = "rank table: equal portals, different ranks";
                return 11;
            }
            if (A.portal != B.portal) {
                r = pair_canopy_split<0>(T.canopy.data(), T.canopy_id.data(), wa & 0xFFFFu, pbot_a,
                                         B.portal, B.D, B.nb);
                // the ladder form of the same canopy must agree bit for bit
                // in-order trees: meeting node from the sparse table, then the two sums with known
                // edge counts; other trees: lock-step search on the ladder
                const uint32_t pa = wa & 0xFFFFu, pb = B.portal;
                PairResult l;
                if (T.inorder_ids) {
                    const uint32_t meet = canopy_meet(T.canopy_pos.data(), T.canopy_rmq.data(), T.canopy_nodes, pa, pb);
                    l = pair_ladder_sums<0>(PtrLadder{T.ladder.data()}, T.canopy_id.data(), meet, pa, T.canopy_depth[pa],
                                            pbot_a, pb, T.canopy_depth[pb], B.D, B.nb);
                    const PairResult l2 = pair_ladder_split<0>(PtrLadder{T.ladder.data()}, T.canopy_depth.data(), T.canopy.data(), T.canopy_id.data(),
                                                               pa, pbot_a, pb, B.D, B.nb);
                    if (l2.mrca != l.mrca || std::memcmp(&l2.dist, &l.dist, 4) != 0) {
                        g_err = "sparse-table form disagrees with the lock-step ladder form";
                        return 5;
                    }
                    if (lineage) {     // what the deep kernel reads: rec_p of both nodes, two 64-bit table entries, the lineage sum of a
                        uint32_t wpa, ya, wpb, yb;
                        std::memcpy(&wpa, T.rec_p.data() + sa * 8, 4);
                        std::memcpy(&ya, T.rec_p.data() + sa * 8 + 4, 4);
                        std::memcpy(&wpb, T.rec_p.data() + sb * 8, 4);
                        std::memcpy(&yb, T.rec_p.data() + sb * 8 + 4, 4);
                        const uint64_t m64 = canopy_meet_ranks64(T.canopy_rmq64.data(), T.canopy_nodes, wpa & 0xFFFFu, wpb & 0xFFFFu);
                        const uint32_t dm = (uint32_t)(m64 >> 32);
                        const uint32_t k_a = (wpa >> 16) - dm, kb_total = (wpb >> 16) - dm;
                        const uint32_t chunks_b = (yb >> 28) + 1;       // 16-byte chunks of b's record the kernel loads
                        if (T.record_cap <= 63 && 4 * chunks_b < B.nb + 1) {
                            g_err = "rec_p: chunk count does not cover b's chain";
                            return 9;
                        }
                        const float d3 = ladder_sum_b<0>(PtrLadder{T.ladder.data()}, kb_total - B.nb, T.lineage_sum[(size_t)(ya & 0x0FFFFFFFu) + k_a], pb, B.D, B.nb);
                        if (dm != (meet >> 16) || (int32_t)(uint32_t)m64 != l.mrca || std::memcmp(&d3, &l.dist, 4) != 0) {
                            g_err = "lineage-sum form disagrees with the ladder form";
                            return 6;
                        }
                    }
                } else {
                    l = pair_ladder_split<0>(PtrLadder{T.ladder.data()}, T.canopy_depth.data(), T.canopy.data(), T.canopy_id.data(), pa, pbot_a, pb, B.D, B.nb);
                }
                if (l.mrca != r.mrca || std::memcmp(&l.dist, &r.dist, 4) != 0) {
                    g_err = "sparse-table / ladder form disagrees with the plain canopy climb";
                    return 4;
                }
            }
            else {
                r = pair_canopy_same_portal(T.canopy_id.data(), A, B);
                // the register form of the predicated kernel (chains of up to 15 slots) must agree bit for bit
                PairResult q = r;
                switch (T.record_cap) {
                    case 1: q = pair_same_portal_regs<1>(T.canopy_id.data(), R, sa, sb); break;
                    case 3: q = pair_same_portal_regs<3>(T.canopy_id.data(), R, sa, sb); break;
                    case 7: q = pair_same_portal_regs<7>(T.canopy_id.data(), R, sa, sb); break;
                    case 15: q = pair_same_portal_regs<15>(T.canopy_id.data(), R, sa, sb); break;
                    case 31: q = pair_same_portal_regs<31>(T.canopy_id.data(), R, sa, sb); break;
                    default: break;
                }
                int32_t qm = r.mrca;
                switch (T.record_cap) {
                    case 1: qm = mrca_same_portal_regs<1>(T.canopy_id.data(), R, sa, sb); break;
                    case 3: qm = mrca_same_portal_regs<3>(T.canopy_id.data(), R, sa, sb); break;
                    case 7: qm = mrca_same_portal_regs<7>(T.canopy_id.data(), R, sa, sb); break;
                    case 15: qm = mrca_same_portal_regs<15>(T.canopy_id.data(), R, sa, sb); break;
                    case 31: qm = mrca_same_portal_regs<31>(T.canopy_id.data(), R, sa, sb); break;
                    default: break;
                }
                if (qm != r.mrca) {
                    g_err = "same portal: the register form of the MRCA id disagrees with the loop form";
                    return 18;
                }
                if (q.mrca != r.mrca || std::memcmp(&q.dist, &r.dist, 4) != 0) {
                    g_err = "same portal: the register form disagrees with the loop form";
                    return 18;
                }
            }
        }
        if (out_d) out_d[i] = (double)r.dist;
        if (out_m) out_m[i] = r.mrca;
    }
    return 0;
}

}  // extern "C"
