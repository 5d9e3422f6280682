// pool_main.cpp -- TEST INFRASTRUCTURE ONLY: exercises suchtree_amd/csrc/host_pipe.h's CopyPool
// (spin-then-sleep dispatch) on the host: many parallel phases of random sizes back to back and
// after pauses long enough for the workers to fall asleep; every element must be visited once.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../../suchtree_amd/csrc/host_pipe.h"

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 4000;
    st::CopyPool pool;
    pool.start(8);
    std::mt19937_64 rng(5);
    std::vector<int> hits;
    for (int round = 0; round < rounds; round++) {
        const int64_t n = (int64_t)(rng() % 300000) + 1;
        hits.assign((size_t)n, 0);
        pool.parallel_for(n, [&](int64_t b, int64_t e) { for (int64_t k = b; k < e; k++) hits[(size_t)k]++; });
        for (int64_t k = 0; k < n; k++)
            if (hits[(size_t)k] != 1) { std::printf("round %d: element %lld visited %d times\n", round, (long long)k, hits[(size_t)k]); return 1; }
        if (round % 200 == 199) std::this_thread::sleep_for(std::chrono::milliseconds(3));   // let the workers go to sleep
    }
    char a[1 << 20], b[1 << 20];
    for (size_t i = 0; i < sizeof(a); i++) a[i] = (char)(i * 7);
    pool.copy(b, a, sizeof(a));
    if (std::memcmp(a, b, sizeof(a)) != 0) return 2;
    pool.stop();
    pool.start(3);            // restartable
    int64_t sum = 0;
    std::mutex m;
    pool.parallel_for(1 << 20, [&](int64_t lo, int64_t hi) { std::lock_guard<std::mutex> g(m); sum += hi - lo; });
    if (sum != (1 << 20)) return 3;
    std::printf("pool ok\n");
    return 0;
}
