// Development: the search pool (spinning fork/join workers, two lanes, many trees) under ThreadSanitizer with the callback evaluator.
// The HIP lane is stubbed out (never constructed):
//   g++ -std=c++17 -O1 -g -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include scripts/hostbench/tsan_pool.cpp \
//       crazyara_amd/csrc/search/{pool,mcts}.cpp crazyara_amd/csrc/chess/{position,policy,planes_host}.cpp -lpthread -o /tmp/tsan_pool
#include <cstdio>
#include <cstring>
#include <chrono>
#include <random>
#include <thread>

#include "../../crazyara_amd/csrc/nn/rise_net.h"
#include "../../crazyara_amd/csrc/search/pool.h"

// link-time stubs for the symbols pool.cpp's HIP lane references
extern "C" hipError_t hipHostMalloc(void**, size_t, unsigned) { return hipErrorNotSupported; }
extern "C" hipError_t hipHostFree(void*) { return hipSuccess; }
namespace cra {
void RiseNet::submit_boards(const void*, int, int, float*, float*, float*) {}
void RiseNet::submit_boards_gathered(const void*, int, int, const uint16_t*, const uint32_t*, uint32_t, float*, float*, float*) {}
void RiseNet::wait() {}
}  // namespace cra

using namespace cra;
using namespace cra::search;

static int eval(void*, const void* descs, int n, float* value, float* probs) {
    const BoardDesc* d = static_cast<const BoardDesc*>(descs);
    for (int i = 0; i < n; ++i) {
        uint64_t h = 1469598103934665603ull;
        for (int k = 0; k < 12; ++k) h = (h ^ d[i].bb[k]) * 1099511628211ull;
        std::minstd_rand0 r(uint32_t(h) | 1u);
        value[i] = float(r() % 2000) / 1000.f - 1.f;
        for (int k = 0; k < 5184; ++k) probs[size_t(i) * 5184 + k] = float(r() % 1000) * 1e-6f;
    }
    return 0;
}

int main() {
    SearchSettings s;
    s.batch_size = 8;
    s.epsilon_greedy_counter = 9;
    s.dirichlet_epsilon = 0.25f;
    SearchPool pool(s, make_callback_evaluator(eval, nullptr, 64, 5184), make_callback_evaluator(eval, nullptr, 64, 5184));
    chess::Position p;
    p.set(chess::start_fen(chess::V_CRAZYHOUSE), false, chess::V_CRAZYHOUSE);
    for (int i = 0; i < 16; ++i) pool.add_position(p);
    SearchStats st;
    for (int round = 0; round < 3; ++round) {
        pool.run(120 * (round + 1), 0, 6, &st);
        std::printf("round %d: %llu nodes, %llu batches\n", round, (unsigned long long)st.nodes, (unsigned long long)st.batches);
        for (int i = 0; i < 16; i += 2) {
            const int best = pool.tree(i).best_move_index();
            if (best >= 0) pool.tree(i).apply_move(pool.tree(i).root().actions[size_t(best)]);
        }
        pool.set_active(1, round == 0);
    }
    // one tree, many collectors (the reference's SearchThreads on one tree): per-node locks, CHILD_PENDING expansion, atomics on the
    // fields other nodes' owners read -- two lanes x 4 collectors on 8 threads, solver and exploration on, tree reuse in between
    {
        SearchPool shared(s, make_callback_evaluator(eval, nullptr, 64, 5184), make_callback_evaluator(eval, nullptr, 64, 5184));
        shared.add_position(p);
        chess::Position mate;
        mate.set("4R2b/1N3rkb/1p2P1pp/p2P4/2P1P3/8/PP4Q1/3R3K[QRBBNNNPPPPpp] w - - 2 53", false, chess::V_CRAZYHOUSE);
        shared.add_position(mate);
        shared.set_shared_collectors(4);
        for (int round = 0; round < 3; ++round) {
            shared.run(1500 * (round + 1), 0, 8, &st);
            std::printf("shared round %d: %llu nodes, %llu batches, root visits %u / %u\n", round, (unsigned long long)st.nodes,
                        (unsigned long long)st.batches, shared.tree(0).root_visits(), shared.tree(1).root_visits());
            const int best = shared.tree(0).best_move_index();
            if (best >= 0) shared.tree(0).apply_move(shared.tree(0).root().actions[size_t(best)]);
        }
        // visit conservation after quiescence
        std::vector<uint32_t> words;
        shared.tree(0).dump(words);
        for (size_t i = 0; i < words.size();) {
            const uint32_t m = words[i], visit_sum = words[i + 1];
            uint32_t sum = 0;
            for (uint32_t c = 0; c < m; ++c) {
                sum += words[i + 8 + 6 * c + 1];
                if (words[i + 8 + 6 * c + 2] != 0) { std::printf("virtual loss left behind\n"); return 1; }
            }
            if (sum != visit_sum) { std::printf("visit counts do not add up\n"); return 1; }
            i += 8 + 6 * m;
        }
    }
    // movetime and an asynchronous stop (halt_ is an atomic read by the driving thread inside run, written by the stopping thread),
    // then the principal variation of the kept tree
    {
        SearchPool timed(s, make_callback_evaluator(eval, nullptr, 64, 5184), make_callback_evaluator(eval, nullptr, 64, 5184));
        for (int i = 0; i < 4; ++i) timed.add_position(p);
        timed.run(0, 0, 4, &st, 150);
        std::printf("movetime 150 ms: %llu simulations in %.3f s\n", (unsigned long long)st.simulations, st.seconds);
        std::thread stopper([&] {
            std::this_thread::sleep_for(std::chrono::milliseconds(120));
            timed.request_stop();
        });
        timed.run(50000000, 0, 4, &st);
        stopper.join();
        std::printf("stopped: %llu simulations in %.3f s\n", (unsigned long long)st.simulations, st.seconds);
        std::vector<chess::Move> pv;
        int mate = 0, cp = 0;
        timed.tree(0).principal_variation(pv, &mate, &cp);
        std::printf("pv length %zu, cp %d, mate %d\n", pv.size(), cp, mate);
        if (pv.empty() || st.seconds > 5.0) return 1;
    }
    // the stop protocol with overlapping searches (round 5): a commanding thread announces, stops, announces the next go and stops it too
    // while the search thread is still inside / between run() calls; generations are guarded by a mutex, stop_gen_ is an atomic
    {
        SearchPool gen(s, make_callback_evaluator(eval, nullptr, 64, 5184), make_callback_evaluator(eval, nullptr, 64, 5184));
        for (int i = 0; i < 4; ++i) gen.add_position(p);
        SearchStats st1, st2, st3;
        gen.announce_go();
        std::thread searcher([&] {
            gen.run(50000000, 0, 4, &st1);                               // search 1: stopped from outside
            gen.run(50000000, 0, 4, &st2);                               // search 2: announced and stopped while search 1 was running
            gen.run(uint32_t(gen.tree(0).root_visits()) + 300, 0, 4, &st3);   // search 3: never stopped
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        gen.request_stop();
        gen.announce_go();
        gen.request_stop();
        searcher.join();
        std::printf("generations: search 1 %llu, search 2 %llu, search 3 %llu simulations\n", (unsigned long long)st1.simulations,
                    (unsigned long long)st2.simulations, (unsigned long long)st3.simulations);
        if (st2.simulations > 4 * 64 || st3.simulations < 100) { std::printf("stop protocol: a stop was lost or leaked\n"); return 1; }
    }
    std::printf("done\n");
    return 0;
}
