// launch_decl.h -- what the launch units of libsuchtree_hip.so export to suchtree_hip.hip: one launch function
// per kernel family and pair source, explicitly instantiated in launch_walk.hip / launch_canopy.hip (the
// kernels themselves are compiled there, in parallel with this unit).  All of them only enqueue.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_common.h"
#include "st_tree.h"

namespace st {

// canopy family (launch_canopy.hip; k_canopy_sorted behind it in launch_canopy_sorted.hip)
// `choice` (or NULL): a probed batch (pair_math.h: probe_says_walk) -- callers that pass it launch the scalar ladder kernel AND the
// tile-sorted walk kernel; every workgroup of either samples the batch itself, the kernel the sample does not choose returns at
// once; *choice receives the verdict (0 ladder, 1 walk) for st_probe_last_choice
template <typename Src>
hipError_t launch_canopy(const st_tree *t, const Src &src, int64_t n, DistSink out_d, MrcaSink out_m, Fault *fault,
                         hipStream_t stream, int *choice = nullptr);
// walk family (launch_walk.hip): k_walk or, for large batches on trees with the tables, k_walk_sorted
template <typename Src>
hipError_t launch_walk(const st_tree *t, const Src &src, int64_t n, DistSink out_d, MrcaSink out_m, Fault *fault,
                       hipStream_t stream, int *choice = nullptr);

#define ST_FOR_EACH_SRC(X) X(SrcContig) X(SrcContig32) X(SrcStrided) X(SrcTriangle) X(SrcGrid) X(SrcQuartet)
#ifndef ST_LAUNCH_UNIT
#define ST_EXTERN_LAUNCH(S)                                                                                              \
    extern template hipError_t launch_canopy<S>(const st_tree *, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t, int *); \
    extern template hipError_t launch_walk<S>(const st_tree *, const S &, int64_t, DistSink, MrcaSink, Fault *, hipStream_t, int *);
ST_FOR_EACH_SRC(ST_EXTERN_LAUNCH)
#undef ST_EXTERN_LAUNCH
#endif

// the mailbox form of k_walk (host_path.h: small batches): n <= kMailboxPairs pairs at d_pairs, results to
// d_dist / d_mrca (either may be NULL), completion word `d_done` <- seq
hipError_t launch_walk_mailbox(const st_tree *t, const long long *d_pairs, int n, double *d_dist, int *d_mrca,
                               unsigned *block_counter, unsigned *d_done, unsigned seq, hipStream_t stream);
// quartet topologies by the walk family (k_quartets) / the pick after a canopy launch over SrcQuartet (k_quartet_pick)
hipError_t launch_quartets_walk(const st_tree *t, const long long *d_quartets, int64_t n, long long *d_out, Fault *fault,
                                hipStream_t stream);
hipError_t launch_quartet_pick(const st_tree *t, const long long *d_quartets, const int *d_mrca6, int64_t n, long long *d_out,
                               hipStream_t stream);

}  // namespace st
