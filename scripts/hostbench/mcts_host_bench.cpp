// Development: host-only timing of Tree::collect / finish_batch (no GPU): a fake evaluator returns a peaked random policy.
//   hipcc -O3 -std=c++17 -x hip --offload-arch=gfx950 scripts/hostbench/mcts_host_bench.cpp crazyara_amd/csrc/search/mcts.cpp
//         crazyara_amd/csrc/chess/{position,policy,planes_host}.cpp -o /tmp/mcts_host_bench
#include <chrono>
#include <cstdio>
#include <random>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../crazyara_amd/csrc/search/mcts.h"

using namespace cra;
using namespace cra::search;

static void run_tree(int sims, int quota, const std::string& fen, double* out_collect_us, int seed) {
    SearchSettings s;
    s.batch_size = quota;
    chess::Position root;
    root.set(fen, false, chess::V_CRAZYHOUSE);
    Tree tree(root, s);
    const int nbp = 5184;
    std::vector<float> probs(size_t(quota) * nbp), values(quota);
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    auto fill = [&](int n) {
        for (int i = 0; i < n; ++i) {
            values[i] = u(rng) * 0.4f - 0.2f;
            float* p = probs.data() + size_t(i) * nbp;
            for (int k = 0; k < nbp; ++k) { const float x = u(rng); p[k] = x * x * x * x * 1e-3f; }
        }
    };
    fill(1);
    tree.set_root_result(values[0], probs.data());
    std::vector<BoardDesc> descs(quota);
    double t_collect = 0;
    long leaves = 0;
    fill(quota);
    while (tree.root_visits() < uint32_t(sims)) {
        const auto a = std::chrono::steady_clock::now();
        const int n = tree.collect(quota, descs.data());
        const auto b = std::chrono::steady_clock::now();
        tree.finish_batch(values.data(), probs.data(), nbp);
        t_collect += std::chrono::duration<double>(b - a).count();
        leaves += n;
    }
    *out_collect_us = t_collect / leaves * 1e6;
}

int main(int argc, char** argv) {
    const int sims = argc > 1 ? atoi(argv[1]) : 100000, quota = argc > 2 ? atoi(argv[2]) : 16;
    const int threads = argc > 3 ? atoi(argv[3]) : 1;
    const std::string fen = argc > 4 ? std::string(argv[4]) : chess::start_fen(chess::V_CRAZYHOUSE);
    std::vector<double> us(threads);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; ++t) th.emplace_back(run_tree, sims, quota, fen, &us[t], t + 1);
    for (auto& t : th) t.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double mn = 1e9, mx = 0, sm = 0;
    for (double v : us) { mn = std::min(mn, v); mx = std::max(mx, v); sm += v; }
    printf("threads %d: collect us/leaf min %.2f avg %.2f max %.2f  (wall %.2f s)\n", threads, mn, sm / threads, mx, wall);
    return 0;
}
