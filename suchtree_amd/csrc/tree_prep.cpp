// tree_prep.cpp -- see tree_prep.h.  Host only; built with -ffp-contract=off
// because pbot must be the same float32 left-to-right sum the reference
// accumulates (/root/reference/SuchTree/MuchTree.pyx:922,934-938).
#include "tree_prep.h"

#include <thread>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <utility>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23   // Linux 5.14
#endif

namespace st {

// fn(begin, end) over contiguous parts of [0, n) on up to 32 threads (one part per `grain` items at least); a thread
// that cannot be started leaves its part to the calling thread, nothing joinable is abandoned, nothing is thrown.
template <typename Fn>
static void parallel_ranges(int64_t n, int64_t grain, Fn fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 32), n / std::max<int64_t>(grain, 1)));
    std::vector<std::thread> threads;
    std::vector<std::pair<int64_t, int64_t>> mine;
    for (int t = 0; t < n_threads; t++) {
        const int64_t x0 = n * t / n_threads, x1 = n * (t + 1) / n_threads;
        if (t + 1 == n_threads) { mine.emplace_back(x0, x1); break; }
        try {
            threads.emplace_back(fn, x0, x1);
        } catch (const std::exception &) {
            mine.emplace_back(x0, x1);
        }
    }
    for (const auto &r : mine) fn(r.first, r.second);
    for (auto &th : threads) th.join();
}

bool build_tree_rmq(TreeTables &T, int64_t max_bytes)
{
    T.tree_rmq.clear();
    T.tree_rmq_levels = 0;
    if (!T.inorder_ids) return false;
    const int64_t n = T.n;
    int32_t levels = 1;
    while (((int64_t)1 << levels) <= n) levels++;
    if ((int64_t)levels * n * 8 > max_bytes) return false;
    T.tree_rmq_levels = levels;
    T.tree_rmq.resize((size_t)levels * (size_t)n);
    for (int64_t i = 0; i < n; i++) T.tree_rmq[(size_t)i] = ((uint64_t)(uint32_t)T.depth[(size_t)i] << 32) | (uint64_t)(uint32_t)i;
    for (int32_t k = 1; k < levels; k++) {
        const uint64_t *lo = T.tree_rmq.data() + (size_t)(k - 1) * (size_t)n;
        uint64_t *cur = T.tree_rmq.data() + (size_t)k * (size_t)n;
        const int64_t half = (int64_t)1 << (k - 1);
        for (int64_t i = 0; i < n; i++) {
            const uint64_t a = lo[i], b = i + half < n ? lo[i + half] : a;
            cur[i] = b < a ? b : a;        // depth in the high word: the shallower entry is the smaller
        }
    }
    return true;
}

bool prepare_basic(const int32_t *parent, const float *distance, int64_t n,
                   TreeTables &T, std::string &err)
{
    T = TreeTables();
    if (n <= 0) { err = "tree has no nodes"; return false; }
    if (n > INT32_MAX) { err = "tree has more than 2^31-1 nodes"; return false; }
    T.n = n;

    // child CSR
    std::vector<int32_t> n_child((size_t)n + 1, 0);
    int64_t n_roots = 0;
    for (int64_t c = 0; c < n; c++) {
        int32_t p = parent[c];
        if (p < 0) { n_roots++; T.root = (int32_t)c; continue; }
        if (p >= n || p == c) { err = "parent id out of range at node " + std::to_string(c); return false; }
        n_child[(size_t)p]++;
    }
    if (n_roots != 1) { err = "expected exactly one root, found " + std::to_string(n_roots); return false; }
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; i++) off[(size_t)i + 1] = off[(size_t)i] + n_child[(size_t)i];
    std::vector<int32_t> child((size_t)(n > 1 ? n - 1 : 1));
    {
        std::vector<int64_t> fill(off.begin(), off.end() - 1);
        for (int64_t c = 0; c < n; c++) {      // increasing id: left child before right
            int32_t p = parent[c];
            if (p >= 0) child[(size_t)fill[(size_t)p]++] = (int32_t)c;
        }
    }

    // BFS: parents before children; also detects cycles / unreachable nodes
    T.bfs_order.resize((size_t)n);
    T.depth.assign((size_t)n, 0);
    int64_t head = 0, tail = 0;
    T.bfs_order[(size_t)tail++] = T.root;
    while (head < tail) {
        int32_t x = T.bfs_order[(size_t)head++];
        for (int64_t k = off[(size_t)x]; k < off[(size_t)x + 1]; k++) {
            int32_t c = child[(size_t)k];
            T.depth[(size_t)c] = T.depth[(size_t)x] + 1;
            T.bfs_order[(size_t)tail++] = c;
        }
    }
    if (tail != n) { err = "parent array contains a cycle or unreachable nodes"; return false; }

    // heights (leaf = 1), leaves, parity layout, reference depth
    T.height.assign((size_t)n, 1);
    bool parity = true;
    int32_t max_leaf_depth = 0;
    for (int64_t k = n - 1; k >= 0; k--) {
        int32_t x = T.bfs_order[(size_t)k];
        bool leaf = n_child[(size_t)x] == 0;
        if (leaf) {
            T.n_leaves++;
            max_leaf_depth = std::max(max_leaf_depth, T.depth[(size_t)x]);
        }
        if (leaf != ((x & 1) == 0)) parity = false;
        int32_t p = parent[x];
        if (p >= 0) T.height[(size_t)p] = std::max(T.height[(size_t)p], T.height[(size_t)x] + 1);
    }
    T.parity_layout = parity && (T.n_leaves == (n + 1) / 2);
    {   // in-order numbering: every internal node x has exactly two children whose subtrees
        // occupy the id ranges [lo, x-1] and [x+1, hi]
        std::vector<int32_t> lo((size_t)n), hi((size_t)n);
        bool ok = true;
        for (int64_t k = n - 1; k >= 0 && ok; k--) {
            const int32_t x = T.bfs_order[(size_t)k];
            const int64_t c0 = off[(size_t)x], c1 = off[(size_t)x + 1];
            if (c1 == c0) { lo[(size_t)x] = hi[(size_t)x] = x; continue; }
            if (c1 - c0 != 2) { ok = false; break; }
            const int32_t l = child[(size_t)c0], r = child[(size_t)c0 + 1];     // increasing id
            if (hi[(size_t)l] != x - 1 || lo[(size_t)r] != x + 1) { ok = false; break; }
            lo[(size_t)x] = lo[(size_t)l];
            hi[(size_t)x] = hi[(size_t)r];
        }
        T.inorder_ids = ok;
    }
    build_tree_rmq(T, kMaxTreeRmqBytes);
    T.tree_depth = max_leaf_depth + 1;   // MuchTree.pyx:218-225 counts nodes

    T.nodes.resize((size_t)n);
    T.stride.resize((size_t)n);
    // (three dependent scattered reads per node: on several threads for large trees)
    parallel_ranges(n, (int64_t)1 << 18, [&](int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; i++) {
        T.nodes[(size_t)i].parent = parent[i];
        T.nodes[(size_t)i].dist = distance[i];
    }
    // stride-3 image: the root's own "length" (-1 in the reference's table) is never an edge
    for (int64_t i = i0; i < i1; i++) {
        const int32_t p1 = parent[i] >= 0 ? parent[i] : (int32_t)i;
        const int32_t p2 = parent[p1] >= 0 ? parent[p1] : p1;
        const int32_t p3 = parent[p2] >= 0 ? parent[p2] : p2;
        Stride3 &e = T.stride[(size_t)i];
        e.d0 = parent[i] >= 0 ? distance[i] : 0.0f;
        e.d1 = (p1 != (int32_t)i && parent[p1] >= 0) ? distance[p1] : 0.0f;
        e.d2 = (p2 != p1 && parent[p2] >= 0) ? distance[p2] : 0.0f;
        e.p3 = p3;
    }
    });
    return true;
}

// v = `count` zero elements.  For the large record tables the cost of that is the kernel zero-filling fresh pages one
// fault at a time under a single-threaded memset (8.4 M nodes with 256-byte records: 7-17 s of a 9 s build on a busy
// host): the pages are populated first, by several threads (MADV_POPULATE_WRITE on the reserved,
// still empty buffer; failures are ignored), and the memset then runs over resident memory.
template <typename T>
static void assign_zero(std::vector<T> &v, size_t count)
{
    const size_t bytes = count * sizeof(T);
    v.clear();
    if (bytes >= ((size_t)64 << 20)) {
        v.reserve(count);
        const long page = sysconf(_SC_PAGESIZE);
        const uintptr_t b = (reinterpret_cast<uintptr_t>(v.data()) + (uintptr_t)page - 1) & ~((uintptr_t)page - 1);
        const uintptr_t e = (reinterpret_cast<uintptr_t>(v.data()) + bytes) & ~((uintptr_t)page - 1);
        if (e > b) {
            const unsigned hw = std::thread::hardware_concurrency();
            const int n_threads = (int)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 16), (e - b) >> 26));
            auto populate = [=](int t) {
                const uintptr_t granule = (uintptr_t)2 << 20;      // whole huge pages per thread
                const uintptr_t span = ((e - b) / (uintptr_t)n_threads + granule - 1) & ~(granule - 1);
                const uintptr_t lo = std::min(e, b + span * (uintptr_t)t), hi = std::min(e, lo + span);
                if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_POPULATE_WRITE);
            };
            std::vector<std::thread> threads;
            for (int t = 1; t < n_threads; t++) {
                try {
                    threads.emplace_back(populate, t);
                } catch (const std::exception &) {
                    populate(t);
                }
            }
            populate(0);
            for (auto &th : threads) th.join();
        }
    }
    v.assign(count, T());
}

static int32_t pow2_ceil(int32_t v) {
    int32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

bool prepare_canopy(const int32_t *parent, const float *distance, TreeTables &T, int max_canopy_nodes)
{
    const int64_t n = T.n;
    T.has_canopy = false;
    if (max_canopy_nodes <= 0 || max_canopy_nodes > kMaxCanopyNodes) max_canopy_nodes = kMaxCanopyNodes;

    // canopy(H) = { x : height(x) > H } is closed under "parent of", and the
    // longest lineage left below it has exactly H nodes.  Smallest H whose
    // canopy fits LDS:
    int32_t hmax = 0;
    for (int64_t i = 0; i < n; i++) hmax = std::max(hmax, T.height[(size_t)i]);
    std::vector<int64_t> count_ge((size_t)hmax + 2, 0);   // nodes with height == h, then suffix sums
    for (int64_t i = 0; i < n; i++) count_ge[(size_t)T.height[(size_t)i]]++;
    for (int32_t h = hmax - 1; h >= 0; h--) count_ge[(size_t)h] += count_ge[(size_t)h + 1];
    // count(height > H) = count_ge[H+1]
    int32_t h_min = 0;
    while (h_min < hmax && count_ge[(size_t)h_min + 1] > max_canopy_nodes) h_min++;
    if (count_ge[(size_t)h_min + 1] > max_canopy_nodes) return false;   // cannot happen (count_ge[hmax+1] = 0)
    int32_t rec_bytes = std::max(kMinRecordBytes, pow2_ceil(8 + 8 * h_min));
    if (rec_bytes > std::max<int32_t>(kMinRecordBytes, std::min<int32_t>(T.max_record_bytes, kLongRecordBytes))) return false;
    // table budget: the record tables are what the canopy family costs (8 + R/2 bytes per node, R/2 more with the
    // id chains of the shared-portal case)
    bool with_ids = true;
    if (T.record_budget_bytes < 0) return true;      // (geometry only: would the tree admit the family? nothing is built)
    T.rec_i.clear();
    if (rec_bytes > kMaxRecordBytes) with_ids = false;      // (1 KB records: a GB per million nodes for the rare shared-portal case)
    if (T.record_budget_bytes > 0) {
        const int64_t core = n * (8 + (int64_t)rec_bytes / 2);
        if (core > T.record_budget_bytes) return false;
        with_ids = core + n * ((int64_t)rec_bytes / 2) <= T.record_budget_bytes;
    }
    // use the whole record: a longer understory means a smaller canopy
    int32_t cap = record_cap_for(rec_bytes);
    int32_t H = std::min(cap, hmax);
    // ... except under long records on trees that stay with the predicated kernel (canopy at most
    // kShallowCanopyDepth edges deep): there the largest canopy that fits wins -- shorter chains to add, fewer
    // record sectors to fetch (1e6 leaves, depth 108: understories of 32 instead of 63 nodes, 1.18e10 -> 1.68e10
    // pairs/s; 2^22 leaves of random shape: 21 instead of 31, 2.34e10 -> 2.40e10).  Deep canopies keep the small
    // image: their tile-sorted kernel needs the LDS for its tiles.
    if (cap >= 31 && h_min < H) {
        int32_t canopy_depth = 0;
        for (int64_t i = 0; i < n; i++)
            if (T.height[(size_t)i] > h_min) canopy_depth = std::max(canopy_depth, T.depth[(size_t)i]);
        if (canopy_depth <= kShallowCanopyDepth) H = h_min;
    }
    // the root must stay in the canopy so that every lineage has a portal
    if (H >= hmax) H = hmax - 1;
    if (H < 0) H = 0;

    // BFS-number the canopy: parent index < child index
    std::vector<int32_t> cidx((size_t)n, -1);
    T.canopy.clear();
    T.canopy_id.clear();
    for (int64_t k = 0; k < n; k++) {
        int32_t x = T.bfs_order[(size_t)k];
        if (T.height[(size_t)x] <= H) continue;
        int32_t p = parent[x];
        CanopyEntry e;
        e.dist = distance[x];
        // a canopy of <= 16384 nodes closed under "parent of" is at most 16383 edges deep
        e.link = (p < 0 ? 0u : (uint32_t)cidx[(size_t)p]) | ((uint32_t)T.depth[(size_t)x] << 16);
        cidx[(size_t)x] = (int32_t)T.canopy.size();
        T.canopy.push_back(e);
        T.canopy_id.push_back(x);
    }
    T.canopy_nodes = (int32_t)T.canopy.size();
    if (T.canopy_nodes < 1 || T.canopy_nodes > kMaxCanopyNodes || T.canopy_nodes > 65535) return false;
    T.canopy_pos.clear();
    T.canopy_rmq.clear();
    T.rmq_levels = 0;
    if (T.inorder_ids) {   // sparse table for the meeting node (see tree_prep.h)
        const int32_t C = T.canopy_nodes;
        std::vector<int32_t> by_id((size_t)C);
        for (int32_t c = 0; c < C; c++) by_id[(size_t)c] = c;
        std::sort(by_id.begin(), by_id.end(), [&](int32_t a, int32_t b) { return T.canopy_id[(size_t)a] < T.canopy_id[(size_t)b]; });
        T.canopy_pos.assign((size_t)C, 0);
        for (int32_t r = 0; r < C; r++) T.canopy_pos[(size_t)by_id[(size_t)r]] = (uint16_t)r;
        int32_t levels = 1;
        while ((1 << levels) <= C) levels++;
        T.rmq_levels = levels;
        T.canopy_rmq.assign((size_t)levels * (size_t)C, 0u);
        for (int32_t r = 0; r < C; r++) {
            const int32_t c = by_id[(size_t)r];
            T.canopy_rmq[(size_t)r] = ((T.canopy[(size_t)c].link >> 16) << 16) | (uint32_t)c;
        }
        for (int32_t k = 1; k < levels; k++) {
            const uint32_t *lo = T.canopy_rmq.data() + (size_t)(k - 1) * (size_t)C;
            uint32_t *cur = T.canopy_rmq.data() + (size_t)k * (size_t)C;
            const int32_t half = 1 << (k - 1);
            for (int32_t i = 0; i < C; i++) {
                const uint32_t a = lo[i], b = i + half < C ? lo[i + half] : a;
                cur[i] = (b >> 16) < (a >> 16) ? b : a;
            }
        }
    }
    T.canopy[0].dist = 0.0f;   // root: never added
    // ladder form: parents precede children in BFS order, so parent entries are complete
    T.ladder.assign((size_t)T.canopy_nodes, LadderEntry{0.0f, 0.0f, 0.0f, 0u});
    T.canopy_depth.assign((size_t)T.canopy_nodes, 0);
    for (int32_t c = 0; c < T.canopy_nodes; c++) {
        const uint32_t p1 = T.canopy[(size_t)c].link & kCanopyParentMask;          // root: 0 (itself)
        const uint32_t p2 = T.canopy[(size_t)p1].link & kCanopyParentMask;
        const uint32_t p3 = T.canopy[(size_t)p2].link & kCanopyParentMask;
        const uint32_t dc = T.canopy[(size_t)c].link >> 16;
        LadderEntry &e = T.ladder[(size_t)c];
        e.d0 = T.canopy[(size_t)c].dist;                       // (root: 0, see above)
        e.d1 = dc >= 2 ? T.canopy[(size_t)p1].dist : 0.0f;
        e.d2 = dc >= 3 ? T.canopy[(size_t)p2].dist : 0.0f;
        e.link = dc >= 3 ? p3 * (uint32_t)sizeof(LadderEntry) : kLadderAbove;
        if (dc >= 1 && p1 >= (uint32_t)c) return false;        // (parents first: what "while link >= place(m)" rests on)
        T.canopy_depth[(size_t)c] = (uint16_t)(T.canopy[(size_t)c].link >> 16);
    }
    T.understory_max = H;
    T.record_bytes = rec_bytes;
    T.record_cap = cap;

    // records: every node's chain is gathered by walking up from the node itself (at most H steps over the 8-byte node
    // table), so the nodes are independent and the table is built on several threads (copying the parent's finished
    // record instead -- 2 x R/2 bytes from a cache-cold slot per node, one node after the other -- took 9.4 s for
    // 8.4 M nodes with 256-byte records, this form 0.5 s on 8 threads)
    const size_t half = (size_t)rec_bytes / 2;
    assign_zero(T.rec_a, (size_t)n * 8);
    assign_zero(T.rec_b, (size_t)n * half);
    if (with_ids) assign_zero(T.rec_i, (size_t)n * half);
    std::atomic<bool> too_long{false};
    auto build_records = [&](int64_t x0, int64_t x1) {
        std::vector<int32_t> ids_scratch((size_t)cap + 1, 0);      // (the id chain of a node when rec_i is left out)
        for (int64_t x = x0; x < x1; x++) {
            const size_t slot = (size_t)record_slot(x, T.parity_layout, T.n_leaves);
            uint8_t *rb = T.rec_b.data() + slot * half;
            uint8_t *ri = with_ids ? T.rec_i.data() + slot * half : nullptr;
            float *D = reinterpret_cast<float *>(rb + 4);
            int32_t *I = with_ids ? reinterpret_cast<int32_t *>(ri + 4) : ids_scratch.data();
            uint32_t w0;
            float pbot = 0.0f;
            if (cidx[(size_t)x] >= 0) {
                w0 = (uint32_t)cidx[(size_t)x];   // chain length 0: the node is its own portal
            } else {
                uint32_t nb = 0;
                int32_t y = (int32_t)x;
                while (cidx[(size_t)y] < 0) {     // (the root is in the canopy)
                    if ((int32_t)nb >= cap) { too_long.store(true); break; }      // cannot happen: nb <= H <= cap
                    D[nb] = distance[y];
                    I[nb] = y;                    // left-aligned for now
                    nb++;
                    y = parent[y];
                }
                if (too_long.load()) return;
                // ids sit at the END of their slots (slot cap - 1 = the portal's child, slot cap - nb = the node itself)
                if ((int32_t)nb < cap) {
                    std::memmove(I + (cap - (int32_t)nb), I, 4 * (size_t)nb);
                    std::memset(I, 0, 4 * (size_t)(cap - (int32_t)nb));
                }
                w0 = (uint32_t)cidx[(size_t)y] | (nb << 16);
                // the reference's accumulator: d = 0; d += dist[n] up the lineage (pyx:934-938)
                volatile float acc = 0.0f;
                for (uint32_t i = 0; i < nb; i++) acc = acc + D[i];
                pbot = acc;
            }
            // unused chain slots hold -0.0f: s + (-0.0f) == s bit for bit for every s, so kernels that keep the chain in
            // registers add all `cap` slots unconditionally (one v_add_f32 per slot, no select: pair_math.h)
            {
                const uint32_t nb_used = w0 >> 16, neg_zero = 0x80000000u;
                for (uint32_t i = nb_used; i < (uint32_t)cap; i++) std::memcpy(D + i, &neg_zero, 4);
            }
            std::memcpy(rb, &w0, 4);
            if (ri) std::memcpy(ri, &pbot, 4);
            std::memcpy(T.rec_a.data() + slot * 8, &w0, 4);
            std::memcpy(T.rec_a.data() + slot * 8 + 4, &pbot, 4);
        }
    };
    parallel_ranges(n, (int64_t)1 << 16, build_records);
    if (too_long.load()) return false;
    T.has_canopy = true;
    return true;
}

bool prepare_leaf_blocks(TreeTables &T, int max_blocks)
{
    T.rec_a4.clear();
    T.leaf_block_portal.clear();
    T.leaf_block_shift = 0;
    if (!T.has_canopy || !T.parity_layout || T.n_leaves < 1 || max_blocks < 1) return false;
    int32_t shift = 0;
    while (((T.n_leaves + ((int64_t)1 << shift) - 1) >> shift) > max_blocks) shift++;
    const int64_t blocks = (T.n_leaves + ((int64_t)1 << shift) - 1) >> shift;
    std::vector<uint16_t> table((size_t)blocks, 0xFFFFu);
    std::vector<uint8_t> seen((size_t)blocks, 0);
    for (int64_t slot = 0; slot < T.n_leaves; slot++) {      // leaves-first layout: slots [0, n_leaves) are the leaves
        uint32_t w0;
        std::memcpy(&w0, T.rec_a.data() + (size_t)slot * 8, 4);
        const uint16_t portal = (uint16_t)(w0 & 0xFFFFu);
        const size_t blk = (size_t)(slot >> shift);
        if (!seen[blk]) { seen[blk] = 1; table[blk] = portal; }
        else if (table[blk] != portal) table[blk] = 0xFFFFu;
    }
    int64_t covered = 0;
    for (int64_t slot = 0; slot < T.n_leaves; slot++) covered += table[(size_t)(slot >> shift)] != 0xFFFFu;
    if (covered * 100 < T.n_leaves * 99) return false;
    T.leaf_block_portal = std::move(table);
    T.leaf_block_shift = shift;
    T.rec_a4.resize((size_t)T.n);
    for (int64_t slot = 0; slot < T.n; slot++) std::memcpy(&T.rec_a4[(size_t)slot], T.rec_a.data() + (size_t)slot * 8 + 4, 4);
    return true;
}

bool prepare_cherries(TreeTables &T)
{
    T.rec_c.clear();
    if (T.leaf_block_portal.empty() || T.leaf_block_shift < 1 || !T.parity_layout || T.canopy_nodes > (int32_t)kLeafBlockPortalMask + 1 ||
        T.record_cap < 1 || T.rec_b.empty())
        return false;
    const size_t half = (size_t)T.record_bytes / 2;      // bytes of one rec_b entry: word0 + cap chain slots
    const int64_t cherries = (T.n_leaves + 1) / 2;
    std::vector<uint8_t> ok((size_t)cherries, 0);
    std::vector<uint8_t> rec_c((size_t)cherries * half, 0);
    for (int64_t c = 0; c < cherries; c++) {
        const int64_t s0 = 2 * c, s1 = 2 * c + 1;
        if (s1 >= T.n_leaves) continue;
        const int64_t x0 = record_node(s0, true, T.n_leaves), x1 = record_node(s1, true, T.n_leaves);
        if (T.nodes[(size_t)x0].parent < 0 || T.nodes[(size_t)x0].parent != T.nodes[(size_t)x1].parent) continue;
        const uint8_t *r0 = T.rec_b.data() + (size_t)s0 * half, *r1 = T.rec_b.data() + (size_t)s1 * half;
        // same portal, same chain length, the same slots above the leaves' own (they are the parent's lineage)
        if (std::memcmp(r0, r1, 4) != 0 || std::memcmp(r0 + 8, r1 + 8, half - 8) != 0) continue;
        uint8_t *q = rec_c.data() + (size_t)c * half;
        std::memcpy(q, r0 + 4, 4);                  // the first leaf's own length
        std::memcpy(q + 4, r1 + 4, 4);              // the second leaf's
        std::memcpy(q + 8, r0 + 8, half - 8);       // slots 1 .. cap-1
        ok[(size_t)c] = 1;
    }
    const int32_t shift = T.leaf_block_shift;
    int64_t covered = 0;
    std::vector<uint16_t> table = T.leaf_block_portal;
    for (size_t b = 0; b < table.size(); b++) {
        if (table[b] == kLeafBlockMixed) continue;
        const int64_t lo = (int64_t)b << shift, hi = std::min<int64_t>(T.n_leaves, ((int64_t)b + 1) << shift);
        bool all = (lo & 1) == 0 && (hi & 1) == 0;      // (whole cherries only)
        for (int64_t c = lo / 2; all && c < hi / 2; c++) all = ok[(size_t)c] != 0;
        if (all) { table[b] |= kLeafBlockCherries; covered += hi - lo; }
    }
    if (covered * 100 < T.n_leaves * 99) return false;
    T.leaf_block_portal = std::move(table);
    T.rec_c = std::move(rec_c);
    return true;
}

static void build_rmq64(TreeTables &T)
{
    T.canopy_rmq64.resize(T.canopy_rmq.size());
    for (size_t i = 0; i < T.canopy_rmq.size(); i++) {
        const uint32_t e = T.canopy_rmq[i];
        T.canopy_rmq64[i] = ((uint64_t)(e >> 16) << 32) | (uint32_t)T.canopy_id[(size_t)(e & 0xFFFFu)];
    }
}

bool prepare_rank_table(TreeTables &T)
{
    T.rec_r.clear();
    if (!T.has_canopy || !T.inorder_ids || T.canopy_rmq.empty() || T.tree_depth > 65535) return false;
    build_rmq64(T);
    T.rec_r.assign((size_t)T.n, 0u);
    for (int64_t x = 0; x < T.n; x++) {
        const size_t slot = (size_t)record_slot(x, T.parity_layout, T.n_leaves);
        uint32_t w0;
        std::memcpy(&w0, T.rec_a.data() + slot * 8, 4);
        T.rec_r[slot] = (uint32_t)T.canopy_pos[(size_t)(w0 & 0xFFFFu)] | ((uint32_t)T.depth[(size_t)x] << 16);
    }
    return true;
}

// One node's block of the lineage tables: the reference's a-side accumulator after each prefix of
// x's lineage (d = 0; d += dist[n] up the lineage, pyx:934-938) and, optionally, the operands.
static inline void fill_lineage_block(const TreeTables &T, int64_t x, float *sums, float *lens)
{
    volatile float acc = 0.0f;
    int32_t v = (int32_t)x;
    const int32_t k_max = T.depth[(size_t)x];
    const int64_t block = lineage_block(k_max);
    sums[0] = 0.0f;
    for (int32_t k = 1; k <= k_max; k++) {
        const float d = T.nodes[(size_t)v].dist;
        acc = acc + d;
        sums[k] = acc;
        if (lens) lens[k - 1] = d;
        v = T.nodes[(size_t)v].parent;
    }
    for (int64_t k = (int64_t)k_max + 1; k < block; k++) sums[k] = 0.0f;
    if (lens)
        for (int64_t k = k_max; k < block; k++) lens[k] = 0.0f;
}

bool prepare_lineage_sums(TreeTables &T, int64_t max_entries, bool with_lens)
{
    T.lineage_sum.clear();
    T.lineage_len.clear();
    T.rec_p.clear();
    T.lineage_node_off.clear();
    T.lineage_node_rec.clear();
    T.crown_rmq.clear();
    if (!T.has_canopy || !T.inorder_ids || T.canopy_rmq.empty()) return false;
    const int64_t n = T.n;
    int64_t entries = 0;
    for (int64_t x = 0; x < n; x++) entries += lineage_block(T.depth[(size_t)x]);
    // (offsets share their word with a 4-bit chunk count; slots are kept in 28 bits by the kernel)
    if (entries > max_entries || entries >= ((int64_t)1 << 28) || n >= ((int64_t)1 << 28) || T.tree_depth > 65535) return false;
    build_rmq64(T);
    assign_zero(T.lineage_sum, (size_t)entries);
    if (with_lens) assign_zero(T.lineage_len, (size_t)entries);
    T.rec_p.assign((size_t)n * 8, 0);
    T.lineage_node_off.resize((size_t)n);
    int64_t off = 0;
    for (int64_t x = 0; x < n; x++) {
        T.lineage_node_off[(size_t)x] = (uint32_t)off;
        const size_t slot = (size_t)record_slot(x, T.parity_layout, T.n_leaves);
        uint32_t w0a;
        std::memcpy(&w0a, T.rec_a.data() + slot * 8, 4);
        const uint32_t chain = w0a >> 16;                                 // nodes of x's understory chain
        const uint32_t chunks = T.record_cap <= 63 ? (chain + 1 + 3) / 4 : 1;   // word0 + chain floats, 16 bytes at a time (a 4-bit field: up to 63 slots; longer chains are read through a pointer)
        const uint32_t off32 = (uint32_t)off | ((chunks - 1) << 28);
        uint32_t w0;
        std::memcpy(&w0, T.rec_a.data() + slot * 8, 4);
        const uint32_t wp = (uint32_t)T.canopy_pos[(size_t)(w0 & 0xFFFFu)] | ((uint32_t)T.depth[(size_t)x] << 16);
        std::memcpy(T.rec_p.data() + slot * 8, &wp, 4);
        std::memcpy(T.rec_p.data() + slot * 8 + 4, &off32, 4);
        fill_lineage_block(T, x, T.lineage_sum.data() + off, with_lens ? T.lineage_len.data() + off : nullptr);
        off += lineage_block(T.depth[(size_t)x]);
    }
    return true;
}

bool prepare_walk_lineage(TreeTables &T, int64_t max_entries, bool with_lens)
{
    T.lineage_sum.clear();
    T.lineage_len.clear();
    T.lineage_node_off.clear();
    T.lineage_node_rec.clear();
    T.crown_rmq.clear();
    const int64_t n = T.n;
    int64_t entries = 0;
    for (int64_t x = 0; x < n; x++) entries += lineage_block(T.depth[(size_t)x]);
    if (entries > max_entries || entries >= ((int64_t)1 << 32)) return false;
    T.lineage_node_off.resize((size_t)n);
    int64_t off = 0;
    for (int64_t x = 0; x < n; x++) {
        T.lineage_node_off[(size_t)x] = (uint32_t)off;
        off += lineage_block(T.depth[(size_t)x]);
    }
    assign_zero(T.lineage_sum, (size_t)entries);
    if (with_lens) assign_zero(T.lineage_len, (size_t)entries);
    auto fill = [&](int64_t lo, int64_t hi) {
        for (int64_t x = lo; x < hi; x++) {
            const size_t o = T.lineage_node_off[(size_t)x];
            fill_lineage_block(T, x, T.lineage_sum.data() + o, with_lens ? T.lineage_len.data() + o : nullptr);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 32), entries >> 22));
    // ranges of equal table size, not equal node count (depths differ)
    std::vector<std::pair<int64_t, int64_t>> ranges;
    int64_t lo = 0;
    for (int t = 1; t <= n_threads; t++) {
        int64_t hi = n;
        if (t < n_threads) {
            const uint32_t want = (uint32_t)((uint64_t)entries * (uint64_t)t / (uint64_t)n_threads);
            hi = std::lower_bound(T.lineage_node_off.begin(), T.lineage_node_off.end(), want) - T.lineage_node_off.begin();
            if (hi < lo) hi = lo;
        }
        ranges.emplace_back(lo, hi);
        lo = hi;
    }
    // a thread that cannot be started (std::system_error) leaves its range to the caller's thread:
    // nothing joinable is ever abandoned, nothing is thrown past this function
    std::vector<std::thread> threads;
    std::vector<std::pair<int64_t, int64_t>> mine;
    for (size_t t = 0; t < ranges.size(); t++) {
        if (t + 1 == ranges.size()) { mine.push_back(ranges[t]); break; }
        try {
            threads.emplace_back(fill, ranges[t].first, ranges[t].second);
        } catch (const std::exception &) {
            mine.push_back(ranges[t]);
        }
    }
    for (const auto &r : mine) fill(r.first, r.second);
    for (auto &th : threads) th.join();
    return true;
}

bool prepare_walk_crown(TreeTables &T, int64_t hot_bytes, int max_ladder_nodes)
{
    T.lineage_node_rec.clear();
    T.crown_rmq.clear();
    T.crown_ladder.clear();
    T.crown_nodes = T.crown_levels = T.crown_height = 0;
    T.crown_hot_bytes = 0;
    const int64_t n = T.n;
    if (T.lineage_node_off.size() != (size_t)n || T.height.size() != (size_t)n) return false;
    // bytes of the crown's blocks by H: suffix sums over heights
    int32_t hmax = 0;
    for (int64_t i = 0; i < n; i++) hmax = std::max(hmax, T.height[(size_t)i]);
    std::vector<int64_t> bytes_ge((size_t)hmax + 2, 0);      // blocks of nodes with height == h, then suffix sums
    for (int64_t i = 0; i < n; i++) bytes_ge[(size_t)T.height[(size_t)i]] += 4 * lineage_block(T.depth[(size_t)i]);
    for (int32_t h = hmax - 1; h >= 0; h--) bytes_ge[(size_t)h] += bytes_ge[(size_t)h + 1];
    // crown(H) = height > H; the root stays in the crown (H <= hmax - 1); nb <= H must fit 8 bits
    int32_t H = 0;
    const int32_t h_top = std::min(hmax - 1, 255);
    while (H < h_top && bytes_ge[(size_t)H + 1] > hot_bytes) H++;
    bool want_ladder = false;
    if (max_ladder_nodes > 0 && T.inorder_ids) {
        // a crown that fits LDS: the smallest H (shortest streams below the portals) with few enough nodes
        std::vector<int64_t> count_ge((size_t)hmax + 2, 0);
        for (int64_t i = 0; i < n; i++) count_ge[(size_t)T.height[(size_t)i]]++;
        for (int32_t h = hmax - 1; h >= 0; h--) count_ge[(size_t)h] += count_ge[(size_t)h + 1];
        int32_t HL = 0;
        while (HL < h_top && count_ge[(size_t)HL + 1] > max_ladder_nodes) HL++;
        if (count_ge[(size_t)HL + 1] <= max_ladder_nodes && count_ge[(size_t)HL + 1] <= 65535) {
            H = HL;
            want_ladder = true;
        }
    }
    T.crown_height = H;
    T.crown_hot_bytes = bytes_ge[(size_t)H + 1];
    // crown nodes by id, their ranks
    std::vector<int32_t> rank((size_t)n, -1);
    int64_t C = 0;
    for (int64_t x = 0; x < n; x++)
        if (T.height[(size_t)x] > H) rank[(size_t)x] = (int32_t)C++;
    if (C < 1 || C >= ((int64_t)1 << 24)) return false;
    T.crown_nodes = (int32_t)C;
    // portal and nb, parents first
    std::vector<int32_t> portal((size_t)n);
    std::vector<uint8_t> nb((size_t)n);
    for (int64_t k = 0; k < n; k++) {
        const int32_t x = T.bfs_order[(size_t)k];
        if (rank[(size_t)x] >= 0) { portal[(size_t)x] = x; nb[(size_t)x] = 0; continue; }
        const int32_t p = T.nodes[(size_t)x].parent;      // (the root is in the crown: p >= 0 here)
        portal[(size_t)x] = portal[(size_t)p];
        nb[(size_t)x] = (uint8_t)(nb[(size_t)p] + 1);
    }
    T.lineage_node_rec.resize((size_t)n * 4);
    for (int64_t x = 0; x < n; x++) {
        uint32_t *r = T.lineage_node_rec.data() + (size_t)x * 4;
        const int32_t p = portal[(size_t)x];
        r[0] = (uint32_t)T.depth[(size_t)x];
        r[1] = T.lineage_node_off[(size_t)x];
        r[2] = T.lineage_node_off[(size_t)p];
        r[3] = (uint32_t)nb[(size_t)x] | ((uint32_t)rank[(size_t)p] << 8);
    }
    if (want_ladder) {
        // rank numbering: entry r describes the crown node of rank r; root: parent and third ancestor clamp to itself
        std::vector<int32_t> node_of((size_t)C);
        for (int64_t x = 0; x < n; x++)
            if (rank[(size_t)x] >= 0) node_of[(size_t)rank[(size_t)x]] = (int32_t)x;
        T.crown_ladder.assign((size_t)C, LadderEntry{0.0f, 0.0f, 0.0f, 0u});
        for (int64_t r = 0; r < C; r++) {
            const int32_t x = node_of[(size_t)r];
            const int32_t q1 = T.nodes[(size_t)x].parent >= 0 ? T.nodes[(size_t)x].parent : x;
            const int32_t q2 = T.nodes[(size_t)q1].parent >= 0 ? T.nodes[(size_t)q1].parent : q1;
            const int32_t q3 = T.nodes[(size_t)q2].parent >= 0 ? T.nodes[(size_t)q2].parent : q2;
            LadderEntry &e = T.crown_ladder[(size_t)r];
            e.d0 = T.nodes[(size_t)x].parent >= 0 ? T.nodes[(size_t)x].dist : 0.0f;
            e.d1 = (q1 != x && T.nodes[(size_t)q1].parent >= 0) ? T.nodes[(size_t)q1].dist : 0.0f;
            e.d2 = (q2 != q1 && T.nodes[(size_t)q2].parent >= 0) ? T.nodes[(size_t)q2].dist : 0.0f;
            // (ranks are not ordered by depth: climbs on this image count their edges)
            e.link = T.depth[(size_t)x] >= 3 ? (uint32_t)rank[(size_t)q3] * (uint32_t)sizeof(LadderEntry) : kLadderAbove;
        }
    }
    if (T.inorder_ids) {
        int32_t levels = 1;
        while (((int64_t)1 << levels) <= C) levels++;
        T.crown_levels = levels;
        T.crown_rmq.resize((size_t)levels * (size_t)C);
        for (int64_t x = 0; x < n; x++)
            if (rank[(size_t)x] >= 0)
                T.crown_rmq[(size_t)rank[(size_t)x]] = ((uint64_t)(uint32_t)T.depth[(size_t)x] << 32) | (uint64_t)(uint32_t)x;
        for (int32_t k = 1; k < levels; k++) {
            const uint64_t *lo = T.crown_rmq.data() + (size_t)(k - 1) * (size_t)C;
            uint64_t *cur = T.crown_rmq.data() + (size_t)k * (size_t)C;
            const int64_t half = (int64_t)1 << (k - 1);
            for (int64_t i = 0; i < C; i++) {
                const uint64_t a = lo[i], b = i + half < C ? lo[i + half] : a;
                cur[i] = b < a ? b : a;
            }
        }
    }
    return true;
}

}  // namespace st
