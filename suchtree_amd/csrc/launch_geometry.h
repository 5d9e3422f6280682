// launch_geometry.h -- workgroup sizes and LDS budgets that the kernels and the host's launch logic
// must agree on (included by the kernel headers and by launch_policy.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace st {

constexpr int kCanopyBlock = 1024;       // lanes of a canopy-family workgroup
constexpr int kSortBuckets = 256;        // counting-sort buckets of the tile-sorted canopy kernel
constexpr int kWalkSortBlock = 1024;     // lanes of a k_walk_sorted workgroup
constexpr int kWalkSortBuckets = 256;

// LDS image of the ladder form: canopy_nodes 16-byte entries (the depths stay in global
// memory: they are read twice per pair, from a table of a few KiB)
__host__ __device__ constexpr size_t ladder_image_bytes(int canopy_nodes)
{
    return (size_t)canopy_nodes * 16;
}

// LDS of one k_canopy_ladder workgroup: the image and, behind it, the eight "counter ran dry" flags of the dynamic deal
constexpr size_t kLdsBytesPerCu = 160 * 1024;
constexpr size_t kLadderFlagBytes = 32;
__host__ __device__ constexpr size_t ladder_kernel_lds_bytes(int canopy_nodes) { return ladder_image_bytes(canopy_nodes) + kLadderFlagBytes; }

// block table of the four-byte a side (k_canopy_ilp<..., true>), padded to the 16-byte staging granule
__host__ __device__ inline size_t leaf_block_image_bytes(int count) { return ((size_t)count * 2 + 15) & ~(size_t)15; }

// LDS scratch of a k_canopy_sorted tile of Q * 1024 pairs: per pair one uint16 (the sorted order), with the
// sparse table one uint32 (the pair's meeting node; b's edge count in lineage-sum mode), with
// lineage sums two more words (a's side, later the distance; b's record slot), then the bucket
// array and the scan carries
__host__ __device__ constexpr size_t sort_scratch_bytes(int q, bool rmq, bool sums = false)
{
    return (size_t)q * kCanopyBlock * (2 + (rmq ? 4 : 0) + (sums ? 8 : 0)) + (size_t)kSortBuckets * 4 + 64;
}

// LDS scratch of a k_walk_sorted tile of Q * 1024 pairs (18 bytes per pair + buckets + carries)
__host__ __device__ constexpr size_t walk_sort_scratch_bytes(int q)
{
    return (size_t)q * kWalkSortBlock * 18 + (size_t)kWalkSortBuckets * 4 + 64;
}

}  // namespace st
