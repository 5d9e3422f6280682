// gather_bench.hip -- what does the chip deliver for uniformly random small reads?
// Sweep over table sizes of the kernels in suchtree_amd/csrc/microbench.hip (the same
// kernels bench.py runs in-process for its `random_sector` block).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bench scripts/micro/gather_bench.hip && /tmp/gather_bench
#include "../../suchtree_amd/csrc/microbench.hip"

int main()
{
    for (long long mb : {2, 8, 16, 32, 64, 128, 512, 4096}) {
        for (int bytes : {4, 8, 32, 64}) {
            for (int blocks : {256, 512}) {
                double g = 0;
                if (stmb_random_sector_reads(0, mb << 20, bytes, blocks, 3, &g)) return 1;
                std::printf("table %5lld MiB  %2d B/read  blocks %4d : %7.2f G reads/s  %6.2f TB/s (64-B sectors)\n",
                            mb, bytes, blocks, g, g * 64 / 1e3);
            }
        }
    }
    double c = 0;
    if (stmb_stream_copy(0, 1ll << 30, 5, &c)) return 1;
    std::printf("stream copy 1 GiB: %.0f GB/s (read + write)\n", c);
    return 0;
}
