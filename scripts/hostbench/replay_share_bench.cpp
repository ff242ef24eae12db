// Round 6: what would MCTS_STORE_STATES-style stored leaf states remove on ONE deep tree?  The product's collector
// clones the root position and replays the selected path move by move for every simulation (SearchThread::get_new_child_to_evaluate without
// MCTS_STORE_STATES, searchthread.cpp:198-213; here Tree::get_new_child_to_evaluate, incremental keys).  Host-only: a fake evaluator
// with a peaked policy and a small value noise grows a narrow, deep tree; mcts.cpp is compiled with -DCRA_REPLAY_PROFILE, which counts the
// ticks of the clone and of every replay step by depth against the whole of Tree::collect.
//   g++ -O3 -std=c++17 -DCRA_REPLAY_PROFILE -Icrazyara_amd/csrc scripts/hostbench/replay_share_bench.cpp crazyara_amd/csrc/search/mcts.cpp
//       crazyara_amd/csrc/chess/{position,policy,planes_host}.cpp -lpthread -o /tmp/replay_share_bench
//   /tmp/replay_share_bench <simulations = 25600> <batch = 256> <policy sharpness = 8> <state budget = default> "<fen>"
// without -DCRA_REPLAY_PROFILE the same program reports wall time only (no tick counters in the descent)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "search/mcts.h"

#ifndef CRA_REPLAY_PROFILE
namespace cra { namespace search {
unsigned long long g_replay_ticks[128], g_replay_steps[128], g_clone_ticks, g_expand_ticks, g_collect_ticks = 1, g_leaf_depth_hist[128];   // (plain build: wall time only)
} }
#endif
namespace cra { namespace search {
extern unsigned long long g_replay_ticks[128], g_replay_steps[128], g_clone_ticks, g_expand_ticks, g_collect_ticks, g_leaf_depth_hist[128];
} }

using namespace cra;
using namespace cra::search;

int main(int argc, char** argv) {
    const int sims = argc > 1 ? atoi(argv[1]) : 25600, quota = argc > 2 ? atoi(argv[2]) : 256;
    const double sharp = argc > 3 ? atof(argv[3]) : 8.0;
    const long budget = argc > 4 ? atol(argv[4]) : -1;          // stored leaf states per tree (Tree::set_state_budget); -1 = the default
    const std::string fen = argc > 5 ? argv[5] : "r1b2bk1/pp3ppp/2pn1bn1/4r3/3Q3P/2N1PB1p/PPP1PPP1/3RK2R/NQp w K - 0 24";
    SearchSettings s;
    s.batch_size = quota;
    chess::Position root;
    root.set(fen, false, chess::V_CRAZYHOUSE);
    Tree tree(root, s);
    if (budget >= 0) tree.set_state_budget(uint32_t(budget));
    const int nbp = 5184;
    std::vector<float> probs(size_t(quota) * nbp), values(quota);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    auto fill = [&](int n) {                                 // a peaked policy: p ~ x^sharp (a few moves carry the mass), values near 0
        for (int i = 0; i < n; ++i) {
            values[i] = u(rng) * 0.2f - 0.1f;
            float* p = probs.data() + size_t(i) * nbp;
            for (int k = 0; k < nbp; ++k) p[k] = std::pow(u(rng), float(sharp)) * 1e-3f;
        }
    };
    fill(1);
    tree.set_root_result(values[0], probs.data());
    std::vector<BoardDesc> descs(quota);
    fill(quota);
    memset(g_replay_ticks, 0, sizeof(g_replay_ticks));
    memset(g_replay_steps, 0, sizeof(g_replay_steps));
    g_clone_ticks = g_expand_ticks = g_collect_ticks = 0;
    const auto t0 = std::chrono::steady_clock::now();
    long leaves = 0;
    double t_finish = 0;
    while (tree.root_visits() < uint32_t(sims)) {
        const int n = tree.collect(quota, descs.data());
        const auto a = std::chrono::steady_clock::now();
        tree.finish_batch(values.data(), probs.data(), nbp);
        t_finish += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        leaves += n;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long replay = 0, steps = 0;
    for (int d = 0; d < 128; ++d) { replay += g_replay_ticks[d]; steps += g_replay_steps[d]; }
    double dsum = 0, dn = 0;
    int dmax = 0;
    for (int d = 0; d < 128; ++d) { dsum += double(d) * g_leaf_depth_hist[d]; dn += g_leaf_depth_hist[d]; if (g_leaf_depth_hist[d]) dmax = d; }
    printf("one tree, %d simulations, batch %d, policy sharpness %.0f, state budget %ld (%u stored): %ld new leaves, leaf depth avg %.1f max %d, wall %.3f s = %.2f us per simulation (collect %.1f %%, finish_batch %.1f %%)\n",
           sims, quota, sharp, budget, tree.stored_states(), leaves, dsum / (dn > 0 ? dn : 1), dmax, wall, wall / sims * 1e6, 100.0 * (wall - t_finish) / wall, 100.0 * t_finish / wall);
#ifndef CRA_REPLAY_PROFILE
    return 0;
#endif
    printf("  of Tree::collect: clone of the root position %.1f %%, replay down the path %.1f %% (%.1f steps per simulation, %.0f ticks per step), "
           "expansion (do_move + move generation + descriptor) %.1f %%, selection and the rest %.1f %%\n",
           100.0 * g_clone_ticks / g_collect_ticks, 100.0 * replay / g_collect_ticks, double(steps) / double(sims), double(replay) / double(steps ? steps : 1),
           100.0 * g_expand_ticks / g_collect_ticks, 100.0 * (double(g_collect_ticks) - g_clone_ticks - replay - g_expand_ticks) / g_collect_ticks);
    printf("  clone + replay = %.1f %% of collect = %.1f %% of the host's search time; replay share by depth of the step:", 100.0 * (g_clone_ticks + replay) / g_collect_ticks,
           100.0 * (g_clone_ticks + replay) / g_collect_ticks * (wall - t_finish) / wall);
    for (int d0 = 1; d0 < 128; d0 += 8) {
        unsigned long long t = 0;
        for (int d = d0; d < d0 + 8 && d < 128; ++d) t += g_replay_ticks[d];
        if (t) printf("  [%d-%d] %.1f %%", d0, d0 + 7, 100.0 * t / g_collect_ticks);
    }
    printf("\n");
    return 0;
}
