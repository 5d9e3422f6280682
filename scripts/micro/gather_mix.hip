// gather_mix.hip -- would splitting the understory records pay?  Emulates the memory side of
// one pair: (A) two random 64-B-sector reads from one 64 MiB table [today], (B) one random
// 8-byte read from an 8 MiB table + one random 32-byte read from a 32 MiB table [split],
// each with the 16-byte pair stream in and 12 bytes out.  Round 3: (C) a 4-byte a entry (4 MiB table).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_mix scripts/micro/gather_mix.hip && /tmp/gather_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, bool NT, bool NTB = false>
__global__ __launch_bounds__(1024) void k_mix(const uint8_t *__restrict__ ta, uint32_t mask_a, int stride_a,
                                              const uint8_t *__restrict__ tb, uint32_t mask_b, int stride_b,
                                              const longlong2 *__restrict__ pairs, long long n,
                                              double *__restrict__ out_d, int *__restrict__ out_m)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        longlong2 p;
        if (NT) { const ll2 v = __builtin_nontemporal_load(reinterpret_cast<const ll2 *>(&pairs[i])); p.x = v.x; p.y = v.y; }
        else p = pairs[i];
        const uint32_t ia = hash((uint32_t)p.x) & mask_a, ib = hash((uint32_t)p.y) & mask_b;
        const uint8_t *pa = ta + (size_t)ia * stride_a, *pb = tb + (size_t)ib * stride_b;
        uint32_t acc;
        if (MODE == 0) {   // today: word0 + pbot of a (two dwords of one 64-B record), 32 B of b's record
            acc = *reinterpret_cast<const uint32_t *>(pa) + *reinterpret_cast<const uint32_t *>(pa + 32);
        } else if (MODE == 1) {   // split: 8-byte a entry
            const uint2 v = *reinterpret_cast<const uint2 *>(pa);
            acc = v.x + v.y;
        } else if (MODE == 2) {   // 4-byte a entry (pbot alone; the portal recovered from the leaf-id range of each portal)
            acc = *reinterpret_cast<const uint32_t *>(pa);
        } else if (MODE == 3) {   // pbot (4 B, 4 MiB table) + portal (2 B, 2 MiB table): two small gathers
            acc = *reinterpret_cast<const uint32_t *>(ta + (size_t)ia * 4) +
                  *reinterpret_cast<const uint16_t *>(ta + (64u << 20) + (size_t)ia * 2);
        } else {                  // 6-byte packed a entry: pbot + 16-bit portal, unaligned (6 MiB table)
            const uint8_t *p6 = ta + (size_t)ia * 6;
            acc = *reinterpret_cast<const uint16_t *>(p6) + *reinterpret_cast<const uint16_t *>(p6 + 2) +
                  *reinterpret_cast<const uint16_t *>(p6 + 4);
        }
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        if (NTB) {
            const u4 b0 = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(pb));
            const u4 b1 = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(pb + 16));
            acc += b0.x + b1.w;
        } else {
            const uint4 b0 = *reinterpret_cast<const uint4 *>(pb), b1 = *reinterpret_cast<const uint4 *>(pb + 16);
            acc += b0.x + b1.w;
        }
        if (NT) { __builtin_nontemporal_store((double)acc, &out_d[i]); __builtin_nontemporal_store((int)acc, &out_m[i]); }
        else { out_d[i] = (double)acc; out_m[i] = (int)acc; }
    }
}

// What an XCD-partitioned second pass could reach: every workgroup reads a only from the
// 1/8 slice and b only from the 1/8 (or 1/16) slice of its own XCD (HW_REG_XCC_ID), so both
// slices fit that XCD's 4 MiB L2.  Pairs in: 8 B (int32 x2), results out: 8 B (f32 + i32).
__global__ __launch_bounds__(1024) void k_sliced(const uint8_t *__restrict__ ta, uint32_t slice_mask_a, int shift_a,
                                                 const uint8_t *__restrict__ tb, uint32_t slice_mask_b, int shift_b,
                                                 const int2 *__restrict__ pairs, long long n,
                                                 float *__restrict__ out_d, int *__restrict__ out_m)
{
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | ((4 - 1) << 11)) & 7u;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int2 p = pairs[i];
        const uint32_t ia = (hash((uint32_t)p.x) & slice_mask_a) | (xcc << shift_a);
        const uint32_t ib = (hash((uint32_t)p.y) & slice_mask_b) | (xcc << shift_b);
        const uint2 v = *reinterpret_cast<const uint2 *>(ta + (size_t)ia * 8);
        const uint8_t *pb = tb + (size_t)ib * 32;
        const uint4 b0 = *reinterpret_cast<const uint4 *>(pb), b1 = *reinterpret_cast<const uint4 *>(pb + 16);
        const uint32_t acc = v.x + v.y + b0.x + b1.w;
        out_d[i] = (float)acc;
        out_m[i] = (int)acc;
    }
}

int main()
{
    const long long n = 100000000;
    uint8_t *ta, *tb; longlong2 *pairs; double *od; int *om;
    CK(hipMalloc(&ta, 128u << 20)); CK(hipMalloc(&tb, 128u << 20));
    CK(hipMalloc(&pairs, n * 16)); CK(hipMalloc(&od, n * 8)); CK(hipMalloc(&om, n * 4));
    CK(hipMemset(ta, 1, 128u << 20)); CK(hipMemset(tb, 1, 128u << 20));
    {   // pseudo-random pair contents
        uint64_t *h = (uint64_t *)malloc(n * 16);
        uint64_t s = 88172645463325252ull;
        for (long long i = 0; i < 2 * n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s & 0xFFFFF; }
        CK(hipMemcpy(pairs, h, n * 16, hipMemcpyHostToDevice)); free(h);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kern, uint32_t ma, int sa, const uint8_t *tbp, uint32_t mb, int sb) -> int {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, ta, ma, sa, tbp, mb, sb, pairs, n, od, om);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-64s %7.3f ms  %6.2f G pairs/s\n", name, best, n / best / 1e6);
        return 0;
    };
    const uint32_t M1 = (1u << 20) - 1;   // 1M leaves
    if (run("today : a,b both from one 64 MiB table of 64-B records", k_mix<0, false>, M1, 64, ta, M1, 64)) return 1;
    if (run("today + non-temporal stream loads/stores", k_mix<0, true>, M1, 64, ta, M1, 64)) return 1;
    if (run("split : a from 8 MiB (8 B), b from 32 MiB (32 B)", k_mix<1, false>, M1, 8, tb, M1, 32)) return 1;
    if (run("split + non-temporal stream loads/stores", k_mix<1, true>, M1, 8, tb, M1, 32)) return 1;
    if (run("split + nt streams + nt b-record loads", k_mix<1, true, true>, M1, 8, tb, M1, 32)) return 1;
    if (run("split + nt b-record loads only", k_mix<1, false, true>, M1, 8, tb, M1, 32)) return 1;
    if (run("r03   : a from 4 MiB (4 B), b from 32 MiB (32 B)", k_mix<2, false>, M1, 4, tb, M1, 32)) return 1;
    if (run("r03   : a from 4 MiB (4 B) + nt streams", k_mix<2, true>, M1, 4, tb, M1, 32)) return 1;
    if (run("r03   : a from 2 MiB (2 B stride: what a 16-bit entry would touch), b from 32 MiB", k_mix<2, false>, M1, 2, tb, M1, 32)) return 1;
    if (run("r03   : a = pbot 4 B (4 MiB) + portal 2 B (2 MiB), b from 32 MiB (32 B)", k_mix<3, false>, M1, 4, tb, M1, 32)) return 1;
    if (run("r03   : a = packed 6 B entry (6 MiB), b from 32 MiB (32 B)", k_mix<4, false>, M1, 6, tb, M1, 32)) return 1;
    if (run("split : a from 8 MiB (8 B), b from 64 MiB (64-B records)", k_mix<1, false>, M1, 8, tb, M1, 64)) return 1;
    if (run("floor : a and b from 2 MiB tables (L2 resident)", k_mix<1, false>, (1u << 15) - 1, 8, tb, (1u << 15) - 1, 32)) return 1;
    {
        // sliced pass: a slice = 2^17 entries x 8 B = 1 MiB, b slice = 2^17 x 32 B = 4 MiB (or 2^16: 2 MiB)
        for (int bbits : {17, 16}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_sliced, dim3(512), dim3(1024), 0, 0, ta, (1u << 17) - 1, 17, tb, (1u << bbits) - 1, bbits,
                                   reinterpret_cast<const int2 *>(pairs), n, reinterpret_cast<float *>(od), om);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("XCD-sliced pass 2: a 1 MiB slice, b %d MiB slice per XCD, 8 B in / 8 B out    %7.3f ms  %6.2f G pairs/s\n",
                   bbits == 17 ? 4 : 2, best, n / best / 1e6);
        }
    }
    return 0;
}
