// Development: host-only timing of ONE tree shared by k collectors (Tree::set_collectors: per-node locks, the reference's `Threads`
// SearchThreads on one tree) -- no GPU, a fake evaluator returns a peaked random policy at once.  Shows what the collectors alone can
// deliver per second when the network costs nothing.
//   g++ -O3 -std=c++17 -pthread scripts/hostbench/shared_tree_bench.cpp crazyara_amd/csrc/search/mcts.cpp \
//       crazyara_amd/csrc/chess/{position,policy,planes_host}.cpp -o /tmp/shared_tree_bench && /tmp/shared_tree_bench 200000 32
#include <atomic>
#include <chrono>
#include <cstdio>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../crazyara_amd/csrc/search/mcts.h"

using namespace cra;
using namespace cra::search;

int main(int argc, char** argv) {
    const int sims = argc > 1 ? atoi(argv[1]) : 200000, quota = argc > 2 ? atoi(argv[2]) : 32;
    const std::string fen = argc > 3 ? std::string(argv[3]) : chess::start_fen(chess::V_CRAZYHOUSE);
    const int nbp = 5184;
    for (int k : {1, 2, 4, 8}) {
        SearchSettings s;
        s.batch_size = quota;
        chess::Position root;
        root.set(fen, false, chess::V_CRAZYHOUSE);
        Tree tree(root, s);
        tree.set_collectors(k);
        std::vector<float> probs(size_t(quota) * nbp);
        std::mt19937 rng(7);
        std::uniform_real_distribution<float> u(0.f, 1.f);
        for (int i = 0; i < quota; ++i) {
            float* p = probs.data() + size_t(i) * nbp;
            for (int j = 0; j < nbp; ++j) { const float x = u(rng); p[j] = x * x * x * x * 1e-3f; }
        }
        float v0 = 0.05f;
        tree.set_root_result(v0, probs.data());
        tree.begin_search();
        std::atomic<long> leaves{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int c = 0; c < k; ++c)
            th.emplace_back([&, c]() {
                std::vector<BoardDesc> descs(quota);
                std::vector<float> values(quota);
                std::mt19937 r2(100 + c);
                std::uniform_real_distribution<float> uv(-0.2f, 0.2f);
                while (tree.root_visits() < uint32_t(sims)) {
                    const int n = tree.collect(quota, descs.data(), c);
                    for (int i = 0; i < n; ++i) values[i] = uv(r2);
                    tree.finish_batch(values.data(), probs.data(), nbp, c);
                    leaves += n;
                    if (n == 0 && tree.root_visits() == 0) break;
                }
            });
        for (auto& t : th) t.join();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d collector(s) x %d leaves on one tree: %.0f leaves/s (%.2f us per leaf and collector), %u root visits, %u nodes, %.2f s\n", k, quota,
               leaves.load() / wall, wall * k / leaves.load() * 1e6, tree.root_visits(), tree.node_count(), wall);
    }
    return 0;
}
