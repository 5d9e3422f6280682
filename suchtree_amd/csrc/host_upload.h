// host_upload.h -- part of suchtree_hip.hip (included in this order: host_tree.h, host_launch.h, host_path.h,
// host_upload.h).
// Table construction (tree_prep.cpp) and upload of a tree to one device.
#pragma once

// Tables are built once on the host, then uploaded to every device of the handle.
struct BuiltTables {
    TreeTables T;
    bool canopy_ok = false;
    bool deep = false;
    int64_t budget = 0;        // device bytes the tables may take (0 = no limit)
    int32_t dropped = 0;       // ST_TABLE_* bits left out under the budget
};

// Table budget of a handle: the creation option, else SUCHTREE_AMD_TABLE_MB (MiB), else none.
static int64_t table_budget_from(int64_t option_bytes)
{
    if (option_bytes > 0) return option_bytes;
    if (const char *env = std::getenv("SUCHTREE_AMD_TABLE_MB")) {
        const long long mb = std::atoll(env);
        if (mb > 0) return (int64_t)mb << 20;
    }
    return 0;
}

// The floor: parent/distance (8), depth (4) and the three-level image (16) per node -- what the walk kernel needs.
static int64_t table_floor_bytes(int64_t n_nodes) { return n_nodes * 28 + 64; }

// Budget of the OPTIONAL walk-family tables of a tree the canopy family does not serve (whole-tree
// sparse table, lineage sums, lineage lengths), in bytes: SUCHTREE_AMD_WALK_TABLE_MB (default 12288;
// 0 = build none of them: the walk kernel then climbs).  They are aids, never requirements: a table
// over budget, or one that later does not fit the device, is simply left out.
constexpr int kLineageEntriesPerNode = 2048;

static int64_t walk_table_budget()
{
    int64_t mb = 12288;
    if (const char *env = std::getenv("SUCHTREE_AMD_WALK_TABLE_MB")) mb = std::atoll(env);
    return std::max<int64_t>(0, mb) << 20;
}

// Lineage-length bytes the crown's (shared) blocks may occupy: small enough to live in an XCD's
// 4 MiB L2 next to the streams.
static int64_t crown_hot_budget()
{
    return (int64_t)1024 << 10;
}

// Largest crown (nodes) the tile-sorted walk kernel keeps in LDS as a ladder, 16 bytes per node beside
// the sort scratch.  Trees only the walk family serves: as large as LDS allows (8192 nodes + a 1024-pair
// tile, or ~6100 nodes + 2048 pairs) -- there the short streams below the portals are what counts (1e6-leaf
// depth-338 tree: 7.37e9 pairs/s with 8192 or 6144, 7.04e9 with 4096, 6.67e9 with 2048).  Deep-canopy trees
// (the walk family is their second family): 5120 nodes, so that 4096-pair tiles fit (ml.tree 1.71e10, nj.tree
// 1.74e10; with 8192: 1.53e10 / 1.40e10).  profiles/walk_ladder_sweep_r03.log.
static int crown_ladder_nodes(bool has_canopy)
{
    return has_canopy ? 5120 : 8192;
}

// The walk family's tables with offsets by node id: the whole-tree sparse table beyond prepare_basic's 64 MB -- the
// meeting node in two reads instead of a lock-step climb of both lineages matters most on large, deep trees -- and
// the lineage tables (a's side in one read, b's side as a stream), all within walk_table_budget() and of at most
// kLineageEntriesPerNode table entries per node: a 30,000-leaf caterpillar (mean depth 15,000) would otherwise get
// 7 GB of lineage tables for 2 MB of tree; it climbs instead.  For trees that only the walk family serves (canopy
// refused, or the caller asked for the walk family) and for deep canopy trees whose lineage tables are too large
// for the 28-bit offsets of the canopy family's form.
static int64_t device_bytes_of(const BuiltTables &B);

static void build_walk_only_tables(BuiltTables &B, int64_t n_nodes, bool has_canopy)
{
    int64_t budget = walk_table_budget();
    if (B.budget > 0) {      // what the handle's table budget leaves beside the floor (and the canopy family's tables)
        if (device_bytes_of(B) > B.budget && !B.T.tree_rmq.empty()) {
            std::vector<uint64_t>().swap(B.T.tree_rmq);
            B.T.tree_rmq_levels = 0;
            B.dropped |= ST_TABLE_TREE_RMQ;
        }
        budget = std::min<int64_t>(budget, std::max<int64_t>(0, B.budget - device_bytes_of(B)));
    }
    const int64_t by_size = (int64_t)kLineageEntriesPerNode * n_nodes;
    if (budget > 0 && (int64_t)B.T.tree_rmq.size() == 0) (void)build_tree_rmq(B.T, std::min<int64_t>(budget, kMaxTreeRmqBytesWalkOnly));
    if (budget > 0) {
        budget -= (int64_t)B.T.tree_rmq.size() * 8;
        // sums + lens when both fit, else the sums alone
        if (!prepare_walk_lineage(B.T, std::min<int64_t>({budget / 8, kMaxWalkLineageEntries, by_size}), true))
            (void)prepare_walk_lineage(B.T, std::min<int64_t>({budget / 4, kMaxWalkLineageEntries, by_size}), false);
        if (!B.T.lineage_sum.empty()) (void)prepare_walk_crown(B.T, crown_hot_budget(), crown_ladder_nodes(has_canopy));
    }
    if (B.budget > 0) {      // (what the handle's budget kept this tree from getting)
        if (B.T.tree_rmq.empty() && B.T.inorder_ids) B.dropped |= ST_TABLE_TREE_RMQ;
        if (B.T.lineage_sum.empty()) B.dropped |= ST_TABLE_LINEAGE_SUM | ST_TABLE_LINEAGE_LEN;
        else if (B.T.lineage_len.empty()) B.dropped |= ST_TABLE_LINEAGE_LEN;
    }
}

// Device bytes of what upload_tree would upload of B (the same conditions, table by table).
static int64_t device_bytes_of(const BuiltTables &B)
{
    const TreeTables &T = B.T;
    auto sz = [](const auto &v) -> int64_t { return v.empty() ? 0 : (int64_t)std::max<size_t>(v.size() * sizeof(v[0]), 16); };
    int64_t b = sz(T.nodes) + sz(T.depth) + sz(T.stride) + sz(T.tree_rmq) + 32;
    const bool walk_lineage = !T.lineage_node_rec.empty() && !T.lineage_sum.empty();
    auto walk_lineage_bytes = [&]() -> int64_t {
        return walk_lineage ? sz(T.lineage_node_rec) + sz(T.lineage_len) + sz(T.crown_rmq) + sz(T.crown_ladder) : 0;
    };
    if (!B.canopy_ok) return b + (walk_lineage ? sz(T.lineage_sum) : 0) + walk_lineage_bytes();
    b += sz(T.canopy) + 8 + sz(T.canopy_id) + sz(T.ladder) + sz(T.canopy_depth) + 16 + (int64_t)kWorkSlots * 64 * 8;
    if (B.deep && T.inorder_ids && !T.canopy_rmq.empty()) b += sz(T.canopy_pos) + sz(T.canopy_rmq);
    b += sz(T.rec_a) + sz(T.rec_b) + sz(T.rec_i);
    if (!T.rec_a4.empty()) b += sz(T.rec_a4) + sz(T.leaf_block_portal) + 16 + sz(T.rec_c);
    if (!T.rec_r.empty()) b += sz(T.rec_r) / 2 + sz(T.canopy_rmq64);      // (uploaded as 2-byte ranks)
    if (!T.lineage_sum.empty()) b += sz(T.lineage_sum) + sz(T.rec_p) + walk_lineage_bytes();
    return b;
}

// Leaves optional tables out, in the order of the ST_TABLE_* bits, until the rest fits B.budget.  (The record
// tables were sized against the budget before they were built: prepare_canopy / TreeTables::record_budget_bytes.)
static void apply_table_budget(BuiltTables &B)
{
    TreeTables &T = B.T;
    if (B.canopy_ok && T.rec_i.empty()) B.dropped |= ST_TABLE_REC_I;      // (prepare_canopy left the id chains out)
    if (B.budget <= 0) return;
    auto over = [&] { return device_bytes_of(B) > B.budget; };
    auto clear = [](auto &v) { std::decay_t<decltype(v)>().swap(v); };
    auto drop_crown = [&] { clear(T.crown_rmq); clear(T.crown_ladder); T.crown_nodes = 0; };
    if (over() && !T.lineage_len.empty()) { clear(T.lineage_len); drop_crown(); B.dropped |= ST_TABLE_LINEAGE_LEN; }
    if (over() && !T.lineage_sum.empty()) {
        clear(T.lineage_sum); clear(T.lineage_len); clear(T.rec_p); clear(T.lineage_node_rec); clear(T.lineage_node_off);
        drop_crown();
        B.dropped |= ST_TABLE_LINEAGE_SUM | ST_TABLE_LINEAGE_LEN;
    }
    if (over() && !T.tree_rmq.empty()) { clear(T.tree_rmq); T.tree_rmq_levels = 0; B.dropped |= ST_TABLE_TREE_RMQ; }
    if (over() && B.canopy_ok && !T.rec_i.empty()) { clear(T.rec_i); B.dropped |= ST_TABLE_REC_I; }
    if (over() && !T.rec_a4.empty()) { clear(T.rec_a4); clear(T.leaf_block_portal); clear(T.rec_c); B.dropped |= ST_TABLE_REC_A4; }
    if (over() && !T.rec_r.empty()) { clear(T.rec_r); clear(T.canopy_rmq64); B.dropped |= ST_TABLE_RANKS; }
    if (over() && B.canopy_ok) {
        clear(T.canopy); clear(T.canopy_id); clear(T.ladder); clear(T.canopy_depth); clear(T.canopy_pos); clear(T.canopy_rmq);
        clear(T.rec_a); clear(T.rec_b); clear(T.rec_i); clear(T.rec_a4); clear(T.leaf_block_portal); clear(T.rec_c); clear(T.rec_r);
        clear(T.canopy_rmq64); clear(T.rec_p);
        T.has_canopy = false;
        B.canopy_ok = false;
        B.deep = false;
        B.dropped |= ST_TABLE_CANOPY | ST_TABLE_REC_I | ST_TABLE_RANKS | ST_TABLE_REC_A4;
    }
}

static int build_tables_impl(const int32_t *parent, const float *distance, int64_t n_nodes, int strategy, BuiltTables &B)
{
    std::string err;
    if (!prepare_basic(parent, distance, n_nodes, B.T, err)) return fail(ST_ERR_TREE, err);
    const int max_canopy = 0;      // (prepare_canopy: the LDS limit)
    if (B.budget > 0) {
        // (tables beyond the floor: the whole-tree sparse table prepare_basic may have built is the first thing to go
        // when the records would not fit beside it)
        B.T.record_budget_bytes = std::max<int64_t>(1, B.budget - table_floor_bytes(n_nodes));
    }
    // Records: up to 512 bytes (63-slot chains) with every kernel of the family; trees that need more -- up to 127
    // levels below the canopy: 1e6 leaves at depth 340 -- get 1 KB records without id chains, read through a pointer by
    // the scalar ladder kernel alone (k_canopy_ladder<0>): 9.4e9 pairs/s on that tree where the walk family's tables
    // give 7.5e9 (profiles/kernel_choice_r04.log).
    const int max_record = kLongRecordBytes;
    if (strategy != ST_STRATEGY_WALK) {
        B.T.max_record_bytes = std::min<int>(max_record, kMaxRecordBytes);
        B.canopy_ok = prepare_canopy(parent, distance, B.T, max_canopy);
        if (!B.canopy_ok && max_record > kMaxRecordBytes && max_canopy == 0) {
            // (the ladder image of the kernel that reads such records: at most 10240 canopy nodes)
            B.T.max_record_bytes = kLongRecordBytes;
            B.canopy_ok = prepare_canopy(parent, distance, B.T, kDeepCanopyNodes);
        }
        // Deep canopies (real, unbalanced phylogenies: hundreds of levels) spend their time in
        // the LDS climb, not in memory.  There a canopy image small enough for two workgroups
        // per CU, a longer understory (more of each lineage pre-summed in its record) and the
        // branchy scalar kernel (finished lanes stop issuing LDS reads) measured 13-30 % faster.
        if (B.canopy_ok && max_canopy == 0) {
            int cdepth = 0;
            for (const CanopyEntry &e : B.T.canopy) cdepth = std::max<int>(cdepth, (int)(e.link >> 16));
            if (cdepth > kDeepCanopyDepth) {
                B.deep = true;
                const int deep_nodes = kDeepCanopyNodes;      // (other sizes: profiles/deep_nodes_r05.log)
                if (B.T.canopy_nodes > deep_nodes) {
                    TreeTables T2 = B.T;
                    if (prepare_canopy(parent, distance, T2, deep_nodes)) B.T = std::move(T2);
                }
                // a's side of every pair from one read (tree_prep.h: lineage sums), b's side of the walk
                // family as a stream (lineage lengths); 4 bytes per node and level each, so only while
                // the table stays below kMaxLineageEntries
                // (tables beyond that: the walk family's own form, offsets by node id -- the walk kernels then serve this
                // tree as they serve trees without a canopy; the tile-sorted canopy kernel goes without lineage sums)
                int64_t max_entries = kMaxLineageEntries;
                if (const char *env = std::getenv("SUCHTREE_AMD_LINEAGE_MAX_ENTRIES")) max_entries = std::min<int64_t>(max_entries, std::max<int64_t>(0, std::atoll(env)));   // tests: force the other form
                if (prepare_lineage_sums(B.T, max_entries)) (void)prepare_walk_crown(B.T, crown_hot_budget(), crown_ladder_nodes(true));
                else build_walk_only_tables(B, n_nodes, true);
            }
        }
    }
    if (strategy == ST_STRATEGY_CANOPY && !B.canopy_ok)
        return fail(ST_ERR_TREE, B.budget > 0 ? "tree does not admit the canopy family under this table budget"
                                              : "tree does not admit the canopy family (understory deeper than a record)");
    if (strategy != ST_STRATEGY_WALK && !B.canopy_ok && B.budget > 0) {
        // refused by the budget, or by the tree itself?  (without a budget the tree decides)
        const int64_t keep = B.T.record_budget_bytes;
        B.T.record_budget_bytes = -1;      // (sentinel: geometry only, see prepare_canopy)
        if (prepare_canopy(parent, distance, B.T, max_canopy)) B.dropped |= ST_TABLE_CANOPY | ST_TABLE_REC_I | ST_TABLE_RANKS | ST_TABLE_REC_A4;
        B.T.record_budget_bytes = keep;
    }
    if (B.canopy_ok) (void)prepare_rank_table(B.T);      // MRCA-only queries of in-order trees
    // four-byte a side for the predicated kernel (shallow canopies): 32 KiB of LDS are left beside a full canopy image
    if (B.canopy_ok && !B.deep && B.T.record_cap <= 15 && prepare_leaf_blocks(B.T, 8192)) (void)prepare_cherries(B.T);      // (+ one record per pair of sibling leaves)
    if (!B.canopy_ok) build_walk_only_tables(B, n_nodes, false);
    apply_table_budget(B);
    return ST_OK;
}

// Nothing may be thrown through the C ABI: allocation failures of the (large) optional tables
// become ST_ERR_NOMEM.
static int build_tables(const int32_t *parent, const float *distance, int64_t n_nodes, int strategy, BuiltTables &B,
                        int64_t budget_option = 0)
{
    B.budget = table_budget_from(budget_option);
    try {
        return build_tables_impl(parent, distance, n_nodes, strategy, B);
    } catch (const std::bad_alloc &) {
        return fail(ST_ERR_NOMEM, "out of host memory while building the tree tables");
    } catch (const std::exception &e) {
        return fail(ST_ERR_NOMEM, std::string("building the tree tables failed: ") + e.what());
    }
}

// An optional table: uploaded when it fits comfortably (at most half of the device memory that is
// free right now), left out -- pointer NULL, kernels use the form without it -- otherwise, and also
// when its allocation or copy fails.  Never an error.
template <typename P, typename V>
static bool upload_optional(P **dst, const V &v, int64_t *bytes)
{
    *dst = nullptr;
    if (v.empty()) return false;
    size_t free_b = 0, total_b = 0;
    const size_t need = v.size() * sizeof(v[0]);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b / 2) {
        (void)hipGetLastError();
        return false;
    }
    const std::string keep = g_last_error;
    if (upload(dst, v, bytes) != ST_OK) {
        if (*dst) (void)hipFree(*dst);
        *dst = nullptr;
        (void)hipGetLastError();
        g_last_error = keep;
        return false;
    }
    return true;
}

// Destroys a handle that is still under construction, on every way out but success; the error message survives.
struct TreeOwner {
    st_tree *t;
    ~TreeOwner()
    {
        if (!t) return;
        std::string keep;
        try { keep = g_last_error; } catch (...) {}
        st_tree_destroy(t);
        try { g_last_error = keep; } catch (...) {}
    }
};

// tune: time the candidate kernels of a deep tree on this device (host_tune.h); peers of a multi-device handle take
// the primary's settings instead.
static int upload_tree(BuiltTables &B, int device, st_tree **out, bool tune = true)
{
    TreeTables &T = B.T;
    int n_dev = 0;
    ST_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev)
        return fail(ST_ERR_HIP, "device " + std::to_string(device) + " not available (" +
                                    std::to_string(n_dev) + " visible)");
    ST_DEVICE(device);
    hipDeviceProp_t prop;
    ST_HIP(hipGetDeviceProperties(&prop, device));

    st_tree *t = new (std::nothrow) st_tree();
    if (!t) return fail(ST_ERR_NOMEM, "out of host memory");
    TreeOwner owner{t};      // (a std::bad_alloc below unwinds to the C ABI's catch: the half-built handle goes with it)
    t->device = device;
    t->dp = pipe_acquire(device);
    t->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    t->n_cu_device = t->n_cu;
    t->n_nodes = T.n;
    t->n_leaves = T.n_leaves;
    int64_t bytes = 0;
    int rc = upload(&t->d_nodes, T.nodes, &bytes);
    if (rc == ST_OK) rc = upload(&t->d_depth, T.depth, &bytes);
    if (rc == ST_OK) rc = upload(&t->d_stride, T.stride, &bytes);
    if (rc == ST_OK) (void)upload_optional(&t->d_tree_rmq, T.tree_rmq, &bytes);
    // lineage tables of the walk family: {node_rec, sums} or nothing; lengths and the crown's table are further options
    auto upload_walk_lineage = [&]() {
        if (T.lineage_node_rec.empty() || T.lineage_sum.empty()) return;
        if (!upload_optional(&t->d_lineage_node_rec, T.lineage_node_rec, &bytes)) return;
        if (!t->d_lineage && !upload_optional(&t->d_lineage, T.lineage_sum, &bytes)) {
            (void)hipFree(t->d_lineage_node_rec);
            t->d_lineage_node_rec = nullptr;
            return;
        }
        (void)upload_optional(&t->d_lineage_len, T.lineage_len, &bytes);
        if (upload_optional(&t->d_crown_rmq, T.crown_rmq, &bytes)) {
            t->crown_nodes = T.crown_nodes;
            (void)upload_optional(&t->d_crown_ladder, T.crown_ladder, &bytes);
        }
    };
    if (rc == ST_OK && !B.canopy_ok) upload_walk_lineage();
    if (rc == ST_OK && B.canopy_ok) {
        t->has_canopy = true;
        t->canopy_nodes = T.canopy_nodes;
        t->rec_bytes = T.record_bytes;
        t->rec_cap = T.record_cap;
        t->parity = T.parity_layout ? 1 : 0;
        for (const CanopyEntry &e : T.canopy) t->canopy_depth = std::max<int>(t->canopy_depth, (int)(e.link >> 16));
        std::vector<CanopyEntry> image = T.canopy;
        if (image.size() & 1) image.push_back(CanopyEntry{0.0f, 0u});   // 16-byte staging granule
        rc = upload(&t->d_canopy, image, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_canopy_id, T.canopy_id, &bytes);
        if (rc == ST_OK) rc = upload(&t->d_ladder, T.ladder, &bytes);
        if (rc == ST_OK) {
            std::vector<uint16_t> cd = T.canopy_depth;
            cd.resize((cd.size() + 7) / 8 * 8, 0);     // 16-byte staging granule
            rc = upload(&t->d_cdepth, cd, &bytes);
        }
        if (rc == ST_OK && B.deep && T.inorder_ids && !T.canopy_rmq.empty()) {
            rc = upload(&t->d_cpos, T.canopy_pos, &bytes);
            if (rc == ST_OK) rc = upload(&t->d_rmq, T.canopy_rmq, &bytes);
        }
        if (rc == ST_OK) rc = upload(&t->d_rec_a, T.rec_a, &bytes);
        if (rc == ST_OK && !T.rec_a4.empty() &&
            canopy_lds_bytes(t) + leaf_block_image_bytes((int)T.leaf_block_portal.size()) <= 160 * 1024) {
            std::vector<uint16_t> blocks = T.leaf_block_portal;
            blocks.resize(leaf_block_image_bytes((int)blocks.size()) / 2, 0xFFFFu);     // 16-byte staging granule
            if (upload_optional(&t->d_rec_a4, T.rec_a4, &bytes)) {
                if (upload_optional(&t->d_leaf_blocks, blocks, &bytes)) {
                    t->leaf_block_shift = T.leaf_block_shift;
                    t->leaf_block_count = (int32_t)T.leaf_block_portal.size();
                    // cherry records: the block table carries their bits, so without the table on the device they must go too
                    if (!T.rec_c.empty() && !upload_optional(&t->d_rec_c, T.rec_c, &bytes)) {
                        for (uint16_t &e : blocks)
                            if (e != kLeafBlockMixed) e &= (uint16_t)~kLeafBlockCherries;
                        (void)hipMemcpy(t->d_leaf_blocks, blocks.data(), blocks.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
                    }
                } else {
                    (void)hipFree(t->d_rec_a4);
                    t->d_rec_a4 = nullptr;
                }
            }
        }
        if (rc == ST_OK) rc = upload(&t->d_rec_b, T.rec_b, &bytes);
        if (rc == ST_OK && !T.rec_i.empty()) rc = upload(&t->d_rec_i, T.rec_i, &bytes);      // (empty: left out under a table budget)
        if (rc == ST_OK && !T.rec_r.empty()) {
            // the MRCA-only kernel needs the rank alone: 2 bytes per node, so that the leaves' half of the
            // table (2 MB for 2^20 leaves) stays in an XCD's L2 (4-byte entries: 4.4e10 ids/s)
            std::vector<uint16_t> ranks(T.rec_r.size());
            for (size_t k = 0; k < ranks.size(); k++) ranks[k] = (uint16_t)(T.rec_r[k] & 0xFFFFu);
            rc = upload(&t->d_rec_r, ranks, &bytes);
            if (rc == ST_OK) rc = upload(&t->d_rmq64, T.canopy_rmq64, &bytes);
        }
        if (rc == ST_OK && T.rec_p.empty() && !T.lineage_sum.empty()) {
            upload_walk_lineage();      // (lineage tables in the walk family's own form only: build_walk_only_tables)
        } else if (rc == ST_OK && t->d_rmq && t->d_rmq64 && !T.lineage_sum.empty()) {
            if (upload_optional(&t->d_rec_p, T.rec_p, &bytes)) {
                if (!upload_optional(&t->d_lineage, T.lineage_sum, &bytes)) {
                    (void)hipFree(t->d_rec_p);
                    t->d_rec_p = nullptr;
                } else {
                    upload_walk_lineage();      // (the walk family's view of the same sums)
                }
            }
        }
    }
    if (rc == ST_OK) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_fault), 2 * sizeof(Fault));
        const Fault init2[2] = {kFaultInit, kFaultInit};
        if (e == hipSuccess) e = hipMemcpy(t->d_fault, init2, sizeof(init2), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(ST_ERR_HIP, std::string("tree setup: ") + hipGetErrorString(e));
        else t->d_fault_host = t->d_fault + 1;
    }
    if (rc == ST_OK && B.canopy_ok && !T.ladder.empty()) {      // work counters of the scalar ladder kernel (optional: static deal without)
        if (hipMalloc(reinterpret_cast<void **>(&t->d_work), (size_t)kWorkSlots * 64 * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            t->d_work = nullptr;
        } else {
            bytes += (int64_t)kWorkSlots * 64 * 8;
            for (unsigned k = 0; k < kWorkSlots; k++) {      // (an event that cannot be created leaves the static deal: launch_canopy.hip)
                if (hipEventCreateWithFlags(&t->work_done[k], hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    t->work_done[k] = nullptr;
                }
            }
            // words of the batch probe (host_launch.h::enqueue_src): optional as well
            const bool ok = t->d_rec_r && hipMalloc(reinterpret_cast<void **>(&t->d_choice), kWorkSlots * sizeof(int)) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                t->d_choice = nullptr;
            } else {
                bytes += (int64_t)kWorkSlots * 4;
            }
        }
    }
    if (rc != ST_OK) return rc;      // (owner destroys t and keeps the message)
    t->strategy = B.canopy_ok ? ST_STRATEGY_CANOPY : ST_STRATEGY_WALK;
    t->info.n_nodes = T.n;
    t->info.n_leaves = T.n_leaves;
    t->info.root = T.root;
    t->info.depth = T.tree_depth;
    t->info.device = device;
    t->info.canopy_nodes = B.canopy_ok ? T.canopy_nodes : 0;
    t->info.understory_max = B.canopy_ok ? T.understory_max : 0;
    t->info.record_bytes = B.canopy_ok ? T.record_bytes : 0;
    t->info.n_devices = 1;
    t->info.device_bytes = bytes;
    t->info.dropped_tables = B.dropped;
    t->info.table_budget_bytes = B.budget;
    t->info.lineage_entries = t->d_lineage ? (int64_t)T.lineage_sum.size() : 0;
    if (t->rec_bytes > kMaxRecordBytes) {      // 1 KB records: the scalar ladder kernel
        t->tile_sort = 0;
        t->ladder_scalar = 1;
    }
    if (B.deep && !tune) rule_for_deep_tree(t);
    if (B.deep && tune) {      // (host_tune.h: the kernel of large batches, by timing the candidates; the rule's defaults if that fails)
        try {
            tune_deep_tree(t, T, prop.gcnArchName);
        } catch (...) {
            rule_for_deep_tree(t);
            t->info.tuned = 0;
        }
    }
    owner.t = nullptr;
    *out = t;
    return ST_OK;
}
