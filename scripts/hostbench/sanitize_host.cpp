// Development: the host-side C++ (positions of every variant, SAN, plane builder, policy map, the search tree with solver / noise /
// tree reuse, the ONNX importer on damaged files) under AddressSanitizer + UBSan.  No GPU, no HIP:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined scripts/hostbench/sanitize_host.cpp \
//       crazyara_amd/csrc/search/mcts.cpp crazyara_amd/csrc/chess/{position,policy,planes_host}.cpp \
//       crazyara_amd/csrc/nn/{onnx_import,netfile}.cpp -o /tmp/sanitize_host && /tmp/sanitize_host [onnx files...]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../crazyara_amd/csrc/chess/planes_host.h"
#include "../../crazyara_amd/csrc/chess/policy.h"
#include "../../crazyara_amd/csrc/nn/onnx_import.h"
#include "../../crazyara_amd/csrc/search/mcts.h"

using namespace cra;

static void playouts(const char* variant, int mode, int layout, int games, std::mt19937& rng) {
    const chess::Variant v = chess::variant_from_name(variant);
    std::vector<float> planes(size_t(layout_channels(layout)) * 64);
    long plies = 0;
    for (int g = 0; g < games; ++g) {
        chess::Position p;
        p.set(chess::start_fen(v), false, v);
        for (int ply = 0; ply < 200; ++ply) {
            std::vector<chess::Move> legal = p.legal_moves();
            if (p.is_terminal(legal.size()) != chess::TERMINAL_NONE || legal.empty()) break;
            chess::board_to_planes(p, layout, true, planes.data());
            for (chess::Move m : legal) {
                (void)p.move_to_san(m);
                (void)p.move_to_uci(m);
                (void)p.gives_check(m);
            }
            chess::Position q;
            q.set(p.fen(), false, v);
            if (q.fen() != p.fen()) { std::printf("FEN round trip differs: %s\n", p.fen().c_str()); std::exit(1); }
            p.do_move(legal[rng() % legal.size()]);
            ++plies;
        }
    }
    std::printf("%-14s mode %d layout %d: %d games, %ld plies\n", variant, mode, layout, games, plies);
}

static void run_search(const char* variant, int mode, int major, int minor, bool noise, std::mt19937& rng) {
    using namespace cra::search;
    const chess::Variant v = chess::variant_from_name(variant);
    SearchSettings s;
    s.batch_size = 8;
    s.mode = mode;
    s.version_major = major;
    s.version_minor = minor;
    s.epsilon_greedy_counter = 7;
    s.epsilon_checks_counter = 11;
    s.dirichlet_epsilon = noise ? 0.25f : 0.f;
    chess::Position root;
    root.set(chess::start_fen(v), false, v);
    Tree tree(root, s);
    const int nbp = mode == 0 ? 5184 : mode == 1 ? 4864 : 5376;
    std::vector<float> probs(size_t(8) * nbp), values(8);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    auto fill = [&](int n) {
        for (int i = 0; i < n; ++i) {
            values[size_t(i)] = u(rng) * 1.6f - 0.8f;
            float* p = probs.data() + size_t(i) * nbp;
            for (int k = 0; k < nbp; ++k) p[k] = u(rng) * 1e-3f;
        }
    };
    std::vector<BoardDesc> descs(8);
    long nodes = 0;
    for (int move = 0; move < 12; ++move) {
        if (tree.root_needs_eval()) {
            BoardDesc d;
            tree.root_desc(d);
            fill(1);
            tree.set_root_result(values[0], probs.data());
        }
        if (tree.root().terminal) break;
        tree.begin_search();
        const uint32_t target = tree.root_visits() + 150;
        while (tree.root_visits() < target && !tree.root_solved()) {
            const int n = tree.collect(8, descs.data());
            fill(n);
            tree.finish_batch(values.data(), probs.data(), nbp);
        }
        nodes += tree.node_count();
        const int best = tree.best_move_index();
        {
            std::vector<chess::Move> pv;
            int mate = 0, cp = 0;
            tree.principal_variation(pv, &mate, &cp);                    // EvalInfo pv / centipawns / movesToMate
            chess::Position walk = tree.root_position();
            for (chess::Move m : pv) {                                    // every move of the line is legal where it is played
                bool ok = false;
                for (chess::Move l : walk.legal_moves()) ok = ok || l == m;
                if (!ok) { std::printf("illegal move in the principal variation\n"); std::exit(1); }
                walk.do_move(m);
            }
        }
        if (best < 0) break;
        tree.apply_move(tree.root().actions[size_t(best)]);
    }
    std::printf("search %-14s v%d.%d noise %d: %ld nodes\n", variant, major, minor, int(noise), nodes);
}

static void onnx_damage(const char* path, std::mt19937& rng) {
    std::ifstream f(path, std::ios::binary);
    std::vector<char> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (good.empty()) { std::printf("cannot read %s\n", path); return; }
    int ok = 0, bad = 0;
    for (int it = 0; it < 1500; ++it) {
        std::vector<char> b = good;
        if (it % 3 == 0) b.resize(rng() % b.size());
        else if (it % 3 == 1) for (int k = 0; k < 3; ++k) b[rng() % std::min<size_t>(b.size(), 12000)] = char(rng());
        else { const size_t i = rng() % b.size(); b.erase(b.begin() + long(i), b.begin() + long(std::min(b.size(), i + 1 + rng() % 48))); }
        try {
            NetFile nf;
            import_onnx_bytes(b.data(), b.size(), "m-v1.0.onnx", nf);
            ++ok;
        } catch (const std::exception&) { ++bad; }
    }
    std::printf("onnx %s: %d imported, %d rejected\n", path, ok, bad);
}

int main(int argc, char** argv) {
    std::mt19937 rng(12345);
    playouts("crazyhouse", 0, LAYOUT_CZ_V3, 6, rng);
    playouts("chess", 1, LAYOUT_CHESS_V28, 6, rng);
    playouts("3check", 2, LAYOUT_LICHESS_V3, 4, rng);
    playouts("kingofthehill", 2, LAYOUT_LICHESS_V2, 4, rng);
    playouts("antichess", 2, LAYOUT_LICHESS_V3, 6, rng);
    playouts("atomic", 2, LAYOUT_LICHESS_V3, 6, rng);
    playouts("horde", 2, LAYOUT_LICHESS_V3, 4, rng);
    playouts("racingkings", 2, LAYOUT_LICHESS_V3, 6, rng);
    run_search("crazyhouse", 0, 1, 0, false, rng);
    run_search("crazyhouse", 0, 3, 0, true, rng);
    run_search("chess", 1, 2, 8, true, rng);
    run_search("atomic", 2, 3, 0, false, rng);
    run_search("antichess", 2, 3, 0, true, rng);
    run_search("racingkings", 2, 3, 0, false, rng);
    for (int i = 1; i < argc; ++i) onnx_damage(argv[i], rng);
    std::printf("done\n");
    return 0;
}
