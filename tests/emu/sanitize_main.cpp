// sanitize_main.cpp -- TEST INFRASTRUCTURE ONLY.
// Address/UB-sanitizer run of the product's host-side C++ (table construction, per-pair
// functions compiled for the host, native Newick ingest) on generated trees.  Built and run
// by tests/test_sanitizers.py with g++ -fsanitize=address,undefined; exits non-zero on any
// finding or on a mismatch between the walk and canopy families.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/suchtree_hip.h"
#include "../../suchtree_amd/csrc/host_copy.h"
#include "../../suchtree_amd/csrc/pair_math.h"
#include "../../suchtree_amd/csrc/tree_prep.h"

using namespace st;

static void random_tree(std::mt19937_64 &rng, int n_leaves, double skew, std::vector<int32_t> &parent,
                        std::vector<float> &dist)
{
    const int n = 2 * n_leaves - 1;
    parent.assign(n, -1);
    dist.assign(n, -1.0f);
    struct Item { int lo, hi, par; };
    std::vector<Item> stack{{0, n - 1, -1}};
    std::uniform_real_distribution<double> U(0.0, 1.0);
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.lo == it.hi) { parent[it.lo] = it.par; continue; }
        const int leaves = (it.hi - it.lo) / 2 + 1;
        int left = U(rng) < skew ? (U(rng) < 0.5 ? 1 : leaves - 1) : 1 + (int)(U(rng) * (leaves - 1));
        if (left < 1) left = 1;
        if (left > leaves - 1) left = leaves - 1;
        const int node = it.lo + 2 * left - 1;
        parent[node] = it.par;
        stack.push_back({it.lo, node - 1, node});
        stack.push_back({node + 1, it.hi, node});
    }
    for (int i = 0; i < n; i++)
        if (parent[i] >= 0) dist[i] = U(rng) < 0.1 ? 2.220446e-16f : (float)(U(rng) * 2.0);
}

static int check_tree(std::mt19937_64 &rng, int n_leaves, double skew, int max_canopy)
{
    std::vector<int32_t> parent;
    std::vector<float> dist;
    random_tree(rng, n_leaves, skew, parent, dist);
    TreeTables T;
    std::string err;
    if (!prepare_basic(parent.data(), dist.data(), (int64_t)parent.size(), T, err)) {
        std::printf("prepare_basic failed: %s\n", err.c_str());
        return 1;
    }
    const bool canopy = prepare_canopy(parent.data(), dist.data(), T, max_canopy);
    const bool lineage = canopy && prepare_lineage_sums(T, (int64_t)1 << 24);       // (in-order ids: always, unless too large)
    const bool crown = lineage && prepare_walk_crown(T, (int64_t)4 << 10);
    const int64_t n = T.n;
    std::uniform_int_distribution<int64_t> pick(0, n - 1);
    for (int k = 0; k < 20000; k++) {
        const int64_t a = pick(rng), b = k % 3 == 0 ? std::min<int64_t>(n - 1, a + k % 17) : pick(rng);
        const PairResult w = pair_walk(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b);
        if (pair_walk_mrca(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b) != w.mrca) return 2;
        if (!canopy) continue;
        const RecTables R{T.rec_a.data(), T.rec_b.data(), T.rec_i.data(), T.record_bytes / 2};
        const RecView A = rec_view(R, record_slot(a, T.parity_layout, T.n_leaves));
        const RecView B = rec_view(R, record_slot(b, T.parity_layout, T.n_leaves));
        const PairResult c = A.portal != B.portal
            ? pair_canopy_split<0>(T.canopy.data(), T.canopy_id.data(), A.portal, A.pbot, B.portal, B.D, B.nb)
            : pair_canopy_same_portal(T.canopy_id.data(), A, B);
        // the padding contract of chains kept in registers (pair_math.h: kChainPad): slots beyond the chain hold
        // -0.0f, and adding every slot -- what the kernels do -- gives the bits of adding the chain alone
        {
            float s_all = A.pbot, s_chain = A.pbot;
            for (int32_t q = 0; q < B.cap; q++) {
                uint32_t bits;
                std::memcpy(&bits, B.D + q, 4);
                if ((uint32_t)q >= B.nb && bits != kChainPad) return 12;
                volatile float t = s_all + B.D[q];
                s_all = t;
                if ((uint32_t)q < B.nb) { volatile float u = s_chain + B.D[q]; s_chain = u; }
            }
            if (std::memcmp(&s_all, &s_chain, 4) != 0) return 13;
        }
        if (c.mrca != w.mrca || std::memcmp(&c.dist, &w.dist, 4) != 0) {
            std::printf("mismatch leaves=%d skew=%g pair (%lld,%lld)\n", n_leaves, skew, (long long)a, (long long)b);
            return 3;
        }
        if (A.portal == B.portal) {      // the register forms of the shared-portal case read whole half records
            const int64_t sa = record_slot(a, T.parity_layout, T.n_leaves), sb = record_slot(b, T.parity_layout, T.n_leaves);
            PairResult q = c;
            int32_t qm = c.mrca;
            switch (T.record_cap) {
                case 1: q = pair_same_portal_regs<1>(T.canopy_id.data(), R, sa, sb); qm = mrca_same_portal_regs<1>(T.canopy_id.data(), R, sa, sb); break;
                case 3: q = pair_same_portal_regs<3>(T.canopy_id.data(), R, sa, sb); qm = mrca_same_portal_regs<3>(T.canopy_id.data(), R, sa, sb); break;
                case 7: q = pair_same_portal_regs<7>(T.canopy_id.data(), R, sa, sb); qm = mrca_same_portal_regs<7>(T.canopy_id.data(), R, sa, sb); break;
                case 15: q = pair_same_portal_regs<15>(T.canopy_id.data(), R, sa, sb); qm = mrca_same_portal_regs<15>(T.canopy_id.data(), R, sa, sb); break;
                case 31: q = pair_same_portal_regs<31>(T.canopy_id.data(), R, sa, sb); qm = mrca_same_portal_regs<31>(T.canopy_id.data(), R, sa, sb); break;
                default: break;
            }
            if (q.mrca != w.mrca || qm != w.mrca || std::memcmp(&q.dist, &w.dist, 4) != 0) return 7;
        }
        if (lineage) {      // the walk with a's side from the lineage sums, and the deep kernel's reads
            if (!crown) return 6;
            LineageView lin;      // every table of the walk family at once: sums, streamed lengths, shared blocks, crown table
            lin.node_rec = T.lineage_node_rec.data();
            lin.sums = T.lineage_sum.data();
            lin.lens = T.lineage_len.data();
            lin.shared_blocks = true;
            lin.crown_rmq = T.crown_rmq.empty() ? nullptr : T.crown_rmq.data();
            lin.crown_nodes = T.crown_nodes;
            const PairResult q = pair_walk(T.nodes.data(), T.depth.data(), T.stride.data(), (int32_t)a, (int32_t)b,
                                           T.tree_rmq.empty() ? nullptr : T.tree_rmq.data(), T.n, lin);
            if (q.mrca != w.mrca || std::memcmp(&q.dist, &w.dist, 4) != 0) return 4;
            if (A.portal != B.portal) {
                uint32_t wpa, ya, wpb;
                std::memcpy(&wpa, T.rec_p.data() + record_slot(a, T.parity_layout, T.n_leaves) * 8, 4);
                std::memcpy(&ya, T.rec_p.data() + record_slot(a, T.parity_layout, T.n_leaves) * 8 + 4, 4);
                std::memcpy(&wpb, T.rec_p.data() + record_slot(b, T.parity_layout, T.n_leaves) * 8, 4);
                const uint64_t m64 = canopy_meet_ranks64(T.canopy_rmq64.data(), T.canopy_nodes, wpa & 0xFFFFu, wpb & 0xFFFFu);
                const uint32_t dm = (uint32_t)(m64 >> 32);
                const float d = ladder_sum_b<0>(PtrLadder{T.ladder.data()}, (wpb >> 16) - dm - B.nb,
                                                T.lineage_sum[(size_t)(ya & 0x0FFFFFFFu) + ((wpa >> 16) - dm)], B.portal, B.D, B.nb);
                if ((int32_t)(uint32_t)m64 != w.mrca || std::memcmp(&d, &w.dist, 4) != 0) return 5;
            }
        }
    }
    return 0;
}

static int check_newick(std::mt19937_64 &rng)
{
    const char *texts[] = {"(A,B,(C,D));", "((a:1,b:2)0.9:3,(c:1,(d:1,e:2,f:3,g:4):0)x:2);", "A;", "((A,B);",
                           "(A:1,B:2", "[c](A[x]:1,'q''r':2)[y];", "(,);", "(A:1e400,B:-0.0);", "", ")("};
    for (const char *t : texts) {
        st_newick *h = nullptr;
        int64_t n = 0, nl = 0, nb = 0;
        int32_t root = 0, depth = 0;
        if (st_newick_open(t, (int64_t)std::strlen(t), &h, &n, &nl, &nb, &root, &depth) == ST_OK) {
            std::vector<int32_t> p(n), l(n), r(n), ids(nl);
            std::vector<float> s(n), d(n);
            std::vector<char> names((size_t)nb + 1);
            std::vector<int64_t> off(nl + 1);
            st_newick_fill(h, p.data(), l.data(), r.data(), s.data(), d.data(), ids.data(), names.data(), off.data());
            st_newick_close(h);
        }
    }
    // random bytes must never crash the parser
    std::uniform_int_distribution<int> ch(0, 11);
    const char alphabet[] = "(),:;'[]A1. ";
    for (int k = 0; k < 3000; k++) {
        std::string t;
        const int len = 1 + (int)(rng() % 60);
        for (int i = 0; i < len; i++) t.push_back(alphabet[ch(rng)]);
        st_newick *h = nullptr;
        if (st_newick_open(t.data(), (int64_t)t.size(), &h, nullptr, nullptr, nullptr, nullptr, nullptr) == ST_OK)
            st_newick_close(h);
    }
    return 0;
}

// The host path's memory passes (host_copy.h) at random sizes and alignments, against the
// plain loops they replace; residency probe and pre-faulting on a fresh mapping.
static int check_host_copy(std::mt19937_64 &rng)
{
    for (int round = 0; round < 300; round++) {
        const int64_t m = (int64_t)(rng() % 5000);
        const int skew_in = (int)(rng() % 4), skew_out = (int)(rng() % 4);
        std::vector<int64_t> src((size_t)(2 * m + 4));
        for (auto &v : src) v = (int64_t)(rng() % 3000000) - 5;
        long long want_hi = std::numeric_limits<long long>::min(), want_lo = std::numeric_limits<long long>::max();
        if (m > 0 && round % 3 == 0) {          // a few ids that do not fit int32
            for (int k = 0; k < 3; k++) {
                const long long big = (round % 2 ? 1 : -1) * ((long long)1 << (32 + k)) + (long long)(rng() % 1000);
                src[(size_t)skew_in + (size_t)(rng() % (uint64_t)(2 * m))] = big;
            }
        }
        std::vector<int32_t> got((size_t)(2 * m + 8), 77), want((size_t)(2 * m + 8), 77);
        for (int64_t q = 0; q < 2 * m; q++) {
            const long long v = src[(size_t)(skew_in + q)];
            int32_t w = (int32_t)v;
            if (v > INT32_MAX) { w = INT32_MAX; want_hi = std::max(want_hi, v); }
            else if (v < INT32_MIN) { w = INT32_MIN; want_lo = std::min(want_lo, v); }
            want[(size_t)(skew_out + q)] = w;
        }
        long long hi = std::numeric_limits<long long>::min(), lo = std::numeric_limits<long long>::max();
        st::narrow_pairs_i64(got.data() + skew_out, src.data() + skew_in, m, hi, lo);
        if (got != want || hi != want_hi || lo != want_lo) return 21;

        std::vector<float> f((size_t)(m + 4));
        for (auto &v : f) v = (float)((double)(rng() % 100000) / 977.0);
        std::vector<double> dgot((size_t)(m + 4), -1.0), dwant((size_t)(m + 4), -1.0);
        const int so = (int)(rng() % 3);
        for (int64_t q = 0; q < m; q++) dwant[(size_t)(so + q)] = (double)f[(size_t)(skew_in + q)];
        st::widen_f32_to_f64(dgot.data() + so, f.data() + skew_in, m);
        if (std::memcmp(dgot.data(), dwant.data(), dgot.size() * 8) != 0) return 22;

        std::vector<char> a((size_t)(m * 4 + 70)), b((size_t)(m * 4 + 70), 3), c((size_t)(m * 4 + 70), 3);
        for (auto &v : a) v = (char)rng();
        const int64_t bytes = (int64_t)(rng() % (uint64_t)(m * 4 + 1));
        const int ao = (int)(rng() % 33), bo = (int)(rng() % 33);
        std::memcpy(c.data() + bo, a.data() + ao, (size_t)bytes);
        st::copy_stream(b.data() + bo, a.data() + ao, bytes);
        if (b != c) return 23;
    }
    // the 24-bit wire formats: ids in (6 bytes per pair, packed by the host) and MRCA ids out (3 bytes per id, unpacked
    // by the host), wide and scalar forms against byte-by-byte references, at every alignment of the packed stream
    for (int round = 0; round < 300; round++) {
        const int64_t m = (int64_t)(rng() % 3000);
        const int64_t first = (int64_t)(rng() % 9);
        const long long n_nodes = round % 5 == 0 ? 0xFFFFFF : 1 + (long long)(rng() % 3000000);
        std::vector<int64_t> src64((size_t)(2 * m + 2));
        std::vector<int32_t> src32((size_t)(2 * m + 2));
        for (size_t q = 0; q < src64.size(); q++) {
            src64[q] = (int64_t)(rng() % (uint64_t)n_nodes);
            src32[q] = (int32_t)src64[q];
        }
        long long want_hi = std::numeric_limits<long long>::min(), want_lo = std::numeric_limits<long long>::max();
        if (m > 0 && round % 2 == 0) {
            for (int k = 0; k < 4; k++) {
                const size_t at = (size_t)(rng() % (uint64_t)(2 * m));
                const long long bad = k == 0 ? n_nodes : k == 1 ? -1 - (long long)(rng() % 99) : k == 2 ? n_nodes + (long long)(rng() % 5000) : 0x7FFFFFF0;
                src64[at] = bad;
                src32[at] = (int32_t)bad;
                if (bad < 0) want_lo = std::min(want_lo, bad); else want_hi = std::max(want_hi, bad);
            }
        }
        std::vector<uint8_t> want((size_t)(6 * (first + m) + 16), 0xAB), got64 = want, got32 = want, got_str = want;
        for (int64_t k = 0; k < m; k++)
            for (int c = 0; c < 2; c++) {
                const long long v = src64[(size_t)(2 * k + c)];
                const uint32_t id = (unsigned long long)v >= (unsigned long long)n_nodes ? 0xFFFFFFu : (uint32_t)v;
                for (int b = 0; b < 3; b++) want[(size_t)(6 * (first + k) + 3 * c + b)] = (uint8_t)(id >> (8 * b));
            }
        // the packed stream starts 16-byte aligned in the product (a pinned slot): std::vector's storage is
        alignas(16) static uint8_t slot[6 * 3016 + 32];
        auto run = [&](auto *src, int64_t s0, int64_t s1, std::vector<uint8_t> &out, int code) -> int {
            std::memset(slot, 0xAB, sizeof slot);
            long long hi = std::numeric_limits<long long>::min(), lo = std::numeric_limits<long long>::max();
            st::pack_pairs48(slot, first, src, m, s0, s1, n_nodes, hi, lo);
            std::memcpy(out.data(), slot, out.size());
            if (std::memcmp(out.data() + 6 * first, want.data() + 6 * first, (size_t)(6 * m)) != 0) return code;
            for (int64_t q = 0; q < 6 * first; q++) if (out[(size_t)q] != 0xAB) return code + 1;       // nothing before the range
            for (size_t q = (size_t)(6 * (first + m)); q < out.size(); q++) if (out[q] != 0xAB) return code + 1;      // nothing after
            if (hi != want_hi || lo != want_lo) return code + 2;
            return 0;
        };
        if (const int rc = run(src64.data(), 2, 1, got64, 30)) return rc;
        if (const int rc = run(src32.data(), 2, 1, got32, 40)) return rc;
        {   // strided view: every other row of a (2m, 2) array, columns swapped back
            std::vector<int64_t> wide((size_t)(4 * m + 4), -77);
            for (int64_t k = 0; k < m; k++) { wide[(size_t)(4 * k + 1)] = src64[(size_t)(2 * k)]; wide[(size_t)(4 * k)] = src64[(size_t)(2 * k + 1)]; }
            if (const int rc = run(wide.data() + 1, 4, -1, got_str, 50)) return rc;
        }
        // MRCA ids back: 3 bytes per id, -1 as 0xFFFFFF
        std::vector<int32_t> ids((size_t)m);
        for (auto &v : ids) v = rng() % 7 == 0 ? -1 : (int32_t)(rng() % 0xFFFFFF);
        if (m > 2) { ids[0] = 0xFFFFFE; ids[(size_t)m - 1] = 0; }
        std::vector<uint8_t> packed((size_t)(4 * (first + m) + 4), 0xCD);      // (a staging slot has 4 bytes per pair)
        for (int64_t k = 0; k < m; k++)
            for (int b = 0; b < 3; b++) packed[(size_t)(3 * (first + k) + b)] = (uint8_t)((uint32_t)ids[(size_t)k] >> (8 * b));
        const int so = (int)(rng() % 9);
        std::vector<int32_t> back((size_t)(m + 12), 55), back_want((size_t)(m + 12), 55);
        for (int64_t k = 0; k < m; k++) back_want[(size_t)(so + k)] = ids[(size_t)k];
        st::unpack_ids24(back.data() + so, packed.data(), first, m);
        if (back != back_want) return 60;
    }
    // a fresh anonymous mapping is not resident; after populate_for_write it is
    const size_t len = (size_t)8 << 20;
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return 24;
    char *q = static_cast<char *>(p) + 24;      // (numpy-like: data starts a few bytes into the mapping)
    if (st::looks_resident(q, (int64_t)len - 24)) return 25;
    st::advise_huge(q, (int64_t)len - 24);
    st::populate_for_write(q, (int64_t)len - 24);
    q[0] = 1;                                   // the partial first and last pages are the copy loops' to touch
    q[len - 25] = 1;
    if (!st::looks_resident(q, (int64_t)len - 24)) return 26;
    if (!st::looks_resident(q, 0)) return 27;
    munmap(p, len);
    return 0;
}

int main()
{
    std::mt19937_64 rng(12345);
    if (const int rc = check_host_copy(rng)) { std::printf("FAILED host_copy rc=%d\n", rc); return rc; }
    const int sizes[] = {1, 2, 3, 7, 64, 1000, 20000};
    const double skews[] = {0.0, 0.5, 0.9, 0.999};
    for (int n : sizes)
        for (double s : skews)
            for (int cap : {0, 64, 1}) {
                const int rc = check_tree(rng, n, s, cap);
                if (rc) { std::printf("FAILED rc=%d\n", rc); return rc; }
            }
    if (check_newick(rng)) return 9;
    std::printf("sanitize ok\n");
    return 0;
}
