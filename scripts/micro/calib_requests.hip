// calib_requests.hip -- known byte counts for calibrating rocprofv3's TCC request counters on
// gfx950 (MI355X_MICROARCH.md, section HBM: "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Two launches with exactly known traffic:
//   stmb::k_copy    1 GiB streamed in (16 B per lane, coalesced) and 1 GiB streamed out
//   stmb::k_gather  512 x 1024 lanes x 256 random 32-byte reads, one per 64-byte sector, from a
//                   64 MiB table (the canopy kernel's record-fetch pattern)
// Run under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace` (and
// FETCH_SIZE / WRITE_SIZE passes); scripts/summarize_profile.py turns the counts into bytes per
// request for each pattern (profiles/calibration_rNN.json).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/calib_requests scripts/micro/calib_requests.hip
#include "../../suchtree_amd/csrc/microbench.hip"

int main()
{
    double g = 0, c = 0;
    // reps = 1 -> two launches of each kernel (one warm-up + one timed); the summary averages per launch
    if (stmb_stream_copy(0, 1ll << 30, 1, &c)) return 1;
    if (stmb_random_sector_reads(0, 64ll << 20, 32, 512, 1, &g)) return 1;
    std::printf("{\"copy_bytes_read_per_launch\": %lld, \"copy_bytes_written_per_launch\": %lld, "
                "\"gather_reads_per_launch\": %lld, \"gather_table_MiB\": 64, \"copy_GBps\": %.1f, \"gather_Greads_per_s\": %.2f}\n",
                1ll << 30, 1ll << 30, 512ll * 1024 * 256, c, g);
    return 0;
}
