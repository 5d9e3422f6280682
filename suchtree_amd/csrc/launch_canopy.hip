// launch_canopy.hip -- translation unit of the canopy family's plain kernels (k_canopy_ilp, k_canopy_ladder, k_mrca_ranks)
// and of launch_canopy<Src>, the family's one entry point (the tile-sorted kernel sits behind it in
// launch_canopy_sorted.hip).  Built for gfx950 only: hipcc --offload-arch=gfx950 -ffp-contract=off.
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
#include "kernels_canopy.h"
#include "launch_canopy_sorted.h"

namespace st {

template <typename Kern, typename Src>
static hipError_t launch_canopy_k(Kern kern, int ppl, const st_tree *t, const CanopyParams &P,
                                  const Src &src, int64_t n, DistSink out_d, MrcaSink out_m,
                                  Fault *fault, hipStream_t stream, size_t lds = 0)
{
    if (lds == 0) lds = canopy_lds_bytes(t);
    if (lds > 64 * 1024) {
        // dynamic LDS above 64 KiB has to be granted per kernel (cheap host-side call)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // one or two 1024-lane workgroups per CU, whatever the LDS image allows
    const int wg_per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int64_t tile = (int64_t)kCanopyBlock * ppl;
    int64_t blocks = (n + tile - 1) / tile;
    blocks = std::min<int64_t>(blocks, (int64_t)t->n_cu * wg_per_cu);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kCanopyBlock), lds, stream, P, src,
                       (long long)n, out_d, out_m, fault);
    return hipGetLastError();
}

// k_canopy_ladder: on long records batches of kLadderDynamicMin pairs and more draw their work from counters (a slot of
// the handle's ring, zeroed on the stream in front of the launch); everything else is dealt statically.
constexpr int64_t kLadderDynamicMin = (int64_t)1 << 22;      // (records of 512 bytes and more; 1 KB records: half of it)
template <int CAP, typename Src>
static hipError_t launch_canopy_ladder(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n,
                                       DistSink out_d, MrcaSink out_m, Fault *fault, hipStream_t stream, int *choice = nullptr)
{
    const size_t lds = ladder_kernel_lds_bytes(t->canopy_nodes);      // (image + the kernel's eight "counter ran dry" flags)
    if (lds > kLdsBytesPerCu) return hipErrorInvalidValue;            // (launch_policy.h::ladder_tables_ready keeps such trees away)
    // a's side from the lineage sums (kernels_canopy.h: ladder_pair_sums) where the handle has the tables and chose the form
    const bool sums = ladder_sums_applies(t, n);
    auto kern = sums ? k_canopy_ladder<CAP, Src, true> : k_canopy_ladder<CAP, Src, false>;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int wg_per_cu = (lds <= 80 * 1024 && CAP == 15) ? 2 : 1;      // (the short-record form is compiled for 8 waves per SIMD)
    int64_t blocks = std::min<int64_t>((n + kCanopyBlock - 1) / kCanopyBlock, (int64_t)t->n_cu * wg_per_cu);
    blocks = std::max<int64_t>(blocks, 1);
    unsigned long long *work = nullptr;
    hipEvent_t done = nullptr;
    if (t->d_work && t->ladder_dynamic && t->rec_bytes >= 512 && n >= (t->rec_bytes > 512 ? kLadderDynamicMin / 2 : kLadderDynamicMin)) {
        const unsigned slot = t->work_next.fetch_add(1, std::memory_order_relaxed) % kWorkSlots;
        done = t->work_done[slot];
        if (done) {      // (no event: the static deal)
            work = t->d_work + (size_t)slot * 64;      // eight counters, 64 bytes apart
            // the launch that drew from this slot 64 launches ago may still be queued on ANOTHER stream: zeroing its counters
            // under it would deal chunks twice or not at all -- this stream waits for it first (an event never recorded is complete)
            hipError_t e = hipStreamWaitEvent(stream, done, 0);
            if (e == hipSuccess) e = hipMemsetAsync(work, 0, 64 * sizeof(unsigned long long), stream);
            if (e != hipSuccess) return e;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kCanopyBlock), lds, stream, P, src, (long long)n, out_d, out_m, fault, work, choice);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && work) e = hipEventRecord(done, stream);
    return e;
}

template <int CAP, typename Src>
static hipError_t launch_canopy_t(const st_tree *t, const CanopyParams &P, const Src &src, int64_t n,
                                  DistSink out_d, MrcaSink out_m, Fault *fault, hipStream_t stream, int *choice)
{
    // the scalar ladder kernel: large distance batches of handles that chose it (launch_policy.h: ladder_applies;
    // records of more than 512 bytes: every batch the family takes -- nothing else reads them well)
    if constexpr (CAP == 0 || CAP == 15 || CAP == 31 || CAP == 63) {
        if ((out_d.any() && ladder_applies(t, n)) || (t->rec_bytes > 512 && t->ladder_scalar && ladder_tables_ready(t)))
            return launch_canopy_ladder<CAP>(t, P, src, n, out_d, out_m, fault, stream, choice);
    }
    if (choice) return hipErrorInvalidValue;      // (a probed batch is one the scalar ladder kernel takes: host_launch.h)
    // tile-sorted kernel: deep canopies with chains of at most seven slots, when its scratch fits next to the canopy image
    if constexpr (CAP == 1 || CAP == 3 || CAP == 7) {
        if (t->tile_sort && sorted_q(t) > 0) return launch_canopy_sorted<CAP>(t, P, src, n, out_d, out_m, fault, stream);
    }
    if constexpr (CAP == 0) {
        // 1 KB records exist for the scalar ladder kernel alone (host_upload.h builds them with a canopy that fits its image)
        return ladder_tables_ready(t) ? launch_canopy_ladder<CAP>(t, P, src, n, out_d, out_m, fault, stream) : hipErrorInvalidValue;
    } else if constexpr (CAP == 31 || CAP == 63) {
        // 256- and 512-byte records (large trees of random shape: 2^22 leaves, depth 57, understories of up to 31
        // nodes; 1e6 leaves with some skew: up to 63): the predicated kernel with the chain in 31 / 63 registers
        return launch_canopy_k(k_canopy_ilp<CAP, 1, Src>, 1, t, P, src, n, out_d, out_m, fault, stream);
    } else {
        // explicit pair arrays on trees with the four-byte a side: 4-byte gathers from a table half the size
        if constexpr (std::is_same<Src, SrcContig>::value || std::is_same<Src, SrcContig32>::value) {
            if (P.rec_a4 && P.leaf_blocks)
                return launch_canopy_k(k_canopy_ilp<CAP, 1, Src, true>, 1, t, P, src, n, out_d, out_m, fault, stream,
                                       canopy_lds_bytes(t) + leaf_block_image_bytes(P.leaf_block_count));
        }
        return launch_canopy_k(k_canopy_ilp<CAP, 1, Src>, 1, t, P, src, n, out_d, out_m, fault, stream);
    }
}

template <typename Src>
hipError_t launch_canopy(const st_tree *t, const Src &src, int64_t n, DistSink out_d,
                                MrcaSink out_m, Fault *fault, hipStream_t stream, int *choice)
{
    CanopyParams P;
    P.canopy = t->d_canopy;
    P.canopy_id = t->d_canopy_id;
    P.ladder = t->d_ladder;
    P.cdepth = t->d_cdepth;
    P.cpos = t->d_cpos;
    P.rmq = t->d_rmq;
    P.rec_a = t->d_rec_a;
    P.rec_a4 = t->rec_a4 ? t->d_rec_a4 : nullptr;
    P.leaf_blocks = t->rec_a4 ? t->d_leaf_blocks : nullptr;
    P.rec_c = (t->rec_a4 && t->cherries) ? t->d_rec_c : nullptr;
    P.leaf_block_shift = t->leaf_block_shift;
    P.leaf_block_count = t->leaf_block_count;
    P.rec_b = t->d_rec_b;
    P.rec_i = t->d_rec_i;
    P.nodes = t->d_nodes;
    P.depth = t->d_depth;
    P.stride = t->d_stride;
    P.rec_p = t->d_rec_p;
    P.rmq64 = t->d_rmq64;
    P.rec_r = t->d_rec_r;
    P.lineage = t->d_lineage;
    P.n_nodes = t->n_nodes;
    P.n_leaves = t->n_leaves;
    P.canopy_nodes = t->canopy_nodes;
    P.rec_bytes = t->rec_bytes;
    P.parity = t->parity;
    if (!out_d.any() && out_m.any() && P.rec_r && P.rmq64 && t->mrca_ranks) {
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)t->n_cu * 8));
        switch (t->rec_cap) {      // (chains of up to 31 slots: the shared-portal case compares them in registers)
            case 1: hipLaunchKernelGGL((k_mrca_ranks<1, Src>), dim3((unsigned)blocks), dim3(256), 0, stream, P, src, (long long)n, out_m, fault); break;
            case 3: hipLaunchKernelGGL((k_mrca_ranks<3, Src>), dim3((unsigned)blocks), dim3(256), 0, stream, P, src, (long long)n, out_m, fault); break;
            case 7: hipLaunchKernelGGL((k_mrca_ranks<7, Src>), dim3((unsigned)blocks), dim3(256), 0, stream, P, src, (long long)n, out_m, fault); break;
            case 15: hipLaunchKernelGGL((k_mrca_ranks<15, Src>), dim3((unsigned)blocks), dim3(256), 0, stream, P, src, (long long)n, out_m, fault); break;
            case 31: hipLaunchKernelGGL((k_mrca_ranks<31, Src>), dim3((unsigned)blocks), dim3(256), 0, stream, P, src, (long long)n, out_m, fault); break;
            default: hipLaunchKernelGGL((k_mrca_ranks<0, Src>), dim3((unsigned)blocks), dim3(256), 0, stream, P, src, (long long)n, out_m, fault); break;
        }
        return hipGetLastError();
    }
    switch (t->rec_cap) {
        case 1: return launch_canopy_t<1>(t, P, src, n, out_d, out_m, fault, stream, choice);
        case 3: return launch_canopy_t<3>(t, P, src, n, out_d, out_m, fault, stream, choice);
        case 7: return launch_canopy_t<7>(t, P, src, n, out_d, out_m, fault, stream, choice);
        case 15: return launch_canopy_t<15>(t, P, src, n, out_d, out_m, fault, stream, choice);
        case 31: return launch_canopy_t<31>(t, P, src, n, out_d, out_m, fault, stream, choice);
        case 63: return launch_canopy_t<63>(t, P, src, n, out_d, out_m, fault, stream, choice);
        default: return launch_canopy_t<0>(t, P, src, n, out_d, out_m, fault, stream, choice);
    }
}


#define ST_INSTANTIATE_CANOPY(S) \
    template hipError_t launch_canopy<S>(const st_tree *, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t, int *);
// The instantiations are the slowest part of the build since round 6 (the scalar ladder kernel has two forms): the file is compiled
// three times, -DST_CANOPY_PART=0 / 1 / 2, each part with two of the six pair sources (build.py).
#ifndef ST_CANOPY_PART
ST_FOR_EACH_SRC(ST_INSTANTIATE_CANOPY)      // (one piece: a plain `hipcc -c` of this file still works)
#elif ST_CANOPY_PART == 0
ST_INSTANTIATE_CANOPY(SrcContig)
ST_INSTANTIATE_CANOPY(SrcContig32)
#elif ST_CANOPY_PART == 1
ST_INSTANTIATE_CANOPY(SrcStrided)
ST_INSTANTIATE_CANOPY(SrcTriangle)
#else
ST_INSTANTIATE_CANOPY(SrcGrid)
ST_INSTANTIATE_CANOPY(SrcQuartet)
#endif

}  // namespace st
