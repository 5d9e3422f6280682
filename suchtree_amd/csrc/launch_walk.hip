// launch_walk.hip -- translation unit of the walk family: k_walk, k_walk_sorted, the mailbox kernel, the quartet
// kernels, and their launch functions.  Built for gfx950 only: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#define ST_LAUNCH_UNIT 1
#include "device_common.h"
#include "pair_math.h"
#include "tree_prep.h"
#include "st_tree.h"
#include "launch_decl.h"
#include "launch_policy.h"
#include "kernels_walk.h"

namespace st {

static WalkParams walk_params(const st_tree *t)
{
    WalkParams P;
    P.nodes = t->d_nodes;
    P.depth = t->d_depth;
    P.stride = t->d_stride;
    P.rmq = t->tree_rmq ? t->d_tree_rmq : nullptr;
    P.n_nodes = t->n_nodes;
    P.crown_ladder = nullptr;
    P.probe_rec_r = t->d_rec_r;
    P.probe_n_leaves = t->n_leaves;
    P.probe_parity = t->parity;
    if (t->d_lineage && t->d_lineage_node_rec && t->lineage_sums) {
        P.lineage.node_rec = t->d_lineage_node_rec;
        P.lineage.sums = t->d_lineage;
        P.lineage.lens = t->lineage_lens ? t->d_lineage_len : nullptr;
        P.lineage.shared_blocks = t->walk_crown != 0;
        if (t->walk_crown && t->d_crown_rmq) {
            P.lineage.crown_rmq = t->d_crown_rmq;
            P.lineage.crown_nodes = t->crown_nodes;
            if (t->walk_ladder) P.crown_ladder = t->d_crown_ladder;
        }
    }
    return P;
}

template <int Q, bool LADDER, typename Src>
static hipError_t launch_walk_sorted(const st_tree *t, const WalkParams &P, const Src &src, int64_t n, DistSink out_d,
                                     MrcaSink out_m, Fault *fault, hipStream_t stream, int *choice)
{
    constexpr int64_t tile = (int64_t)Q * kWalkSortBlock;
    const size_t lds = walk_sort_scratch_bytes(Q) + (LADDER ? (size_t)P.lineage.crown_nodes * 16 : 0);
    const int wg_per_cu = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / lds)));
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + tile - 1) / tile, (int64_t)t->n_cu * wg_per_cu));
    int key_shift = 0;      // keys are edge counts of b's lineage below the meeting node
    while ((t->info.depth >> key_shift) >= kWalkSortBuckets) key_shift++;
    auto kern = k_walk_sorted<Q, LADDER, Src>;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kWalkSortBlock), lds, stream, P, src, (long long)n, out_d, out_m, fault, key_shift, choice);
    return hipGetLastError();
}

template <typename Src>
hipError_t launch_walk(const st_tree *t, const Src &src, int64_t n, DistSink out_d,
                              MrcaSink out_m, Fault *fault, hipStream_t stream, int *choice)
{
    const WalkParams P = walk_params(t);
    if (out_d.any() && n >= walk_sorted_min_pairs(t) && walk_sorted_ready(t)) {
        if (P.crown_ladder) {
            // the crown's ladder in LDS: the largest tile that fits beside it, cut finer for batches that would
            // leave CUs idle (launch_policy.h: batch_tile_q)
            const size_t image = (size_t)P.lineage.crown_nodes * 16;
            const int q_max = image + walk_sort_scratch_bytes(4) <= 160 * 1024 ? 4 : image + walk_sort_scratch_bytes(2) <= 160 * 1024 ? 2 : 1;
            if (image + walk_sort_scratch_bytes(q_max) <= 160 * 1024) {
                const int q = batch_tile_q(q_max, 1, t->sort_tile, n, t->n_cu);
                if (q == 4) return launch_walk_sorted<4, true>(t, P, src, n, out_d, out_m, fault, stream, choice);
                if (q == 2) return launch_walk_sorted<2, true>(t, P, src, n, out_d, out_m, fault, stream, choice);
                return launch_walk_sorted<1, true>(t, P, src, n, out_d, out_m, fault, stream, choice);
            }
        }
        const int q = batch_tile_q(t->has_canopy ? 4 : 2, 1, t->sort_tile, n, t->n_cu);
        if (q == 4) return launch_walk_sorted<4, false>(t, P, src, n, out_d, out_m, fault, stream, choice);
        if (q == 2) return launch_walk_sorted<2, false>(t, P, src, n, out_d, out_m, fault, stream, choice);
        return launch_walk_sorted<1, false>(t, P, src, n, out_d, out_m, fault, stream, choice);
    }
    if (choice) return hipErrorInvalidValue;      // (a probed batch is one the tile-sorted kernel takes: host_launch.h)
    int64_t blocks = (n + 255) / 256;
    blocks = std::min<int64_t>(blocks, (int64_t)t->n_cu * 16);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(k_walk<Src>, dim3((unsigned)blocks), dim3(256), 0, stream, P, src,
                       (long long)n, out_d, out_m, fault);
    return hipGetLastError();
}


#define ST_INSTANTIATE_WALK(S) \
    template hipError_t launch_walk<S>(const st_tree *, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t, int *);
ST_FOR_EACH_SRC(ST_INSTANTIATE_WALK)

hipError_t launch_walk_mailbox(const st_tree *t, const long long *d_pairs, int n, double *d_dist, int *d_mrca,
                               unsigned *block_counter, unsigned *d_done, unsigned seq, hipStream_t stream)
{
    const WalkParams P = walk_params(t);
    hipLaunchKernelGGL(k_walk_mailbox, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, P, d_pairs, n, d_dist, d_mrca,
                       block_counter, d_done, seq);
    return hipGetLastError();
}

hipError_t launch_quartets_walk(const st_tree *t, const long long *d_quartets, int64_t n, long long *d_out, Fault *fault,
                                hipStream_t stream)
{
    const WalkParams P = walk_params(t);
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)t->n_cu * 16));
    hipLaunchKernelGGL(k_quartets, dim3((unsigned)blocks), dim3(256), 0, stream, P, d_quartets, (long long)n, 4LL, 1LL, d_out, fault);
    return hipGetLastError();
}

hipError_t launch_quartet_pick(const st_tree *t, const long long *d_quartets, const int *d_mrca6, int64_t n, long long *d_out,
                               hipStream_t stream)
{
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)t->n_cu * 16));
    hipLaunchKernelGGL(k_quartet_pick, dim3((unsigned)blocks), dim3(256), 0, stream, d_quartets, d_mrca6, (long long)n, d_out);
    return hipGetLastError();
}

}  // namespace st
