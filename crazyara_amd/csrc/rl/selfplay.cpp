// Native self-play / arena loops -- see selfplay.h for the reference lines each step follows.
#include "selfplay.h"

#include <fstream>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cfloat>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace cra {
namespace rl {

using chess::Move;
using chess::Position;
using search::SearchStats;
using search::Tree;

// ---------------------------------------------------------------------------------------------------------------------
// blazeutil.h / agent.cpp helpers
// ---------------------------------------------------------------------------------------------------------------------
void apply_temperature(std::vector<double>& p, double t) {           // blazeutil.h:77-87: p^(1/T) renormalised; T == 1 leaves p untouched
    if (t == 1.0) return;
    double sum = 0;
    for (double& v : p) { v = std::pow(v, 1.0 / t); sum += v; }
    for (double& v : p) v /= sum;
}

// get_quantile (blazeutil.h:188-212), literally: sort ascending; 0 if the smallest entry already reaches the quantile; else accumulate
// (in float32) from the SECOND smallest entry and return the entry BEFORE the one that crosses the quantile, plus FLT_EPSILON.  A
// quantile above the mass behind the smallest entry: the reference asserts(false); release builds fall through (nothing is clipped).
double get_quantile(const std::vector<double>& p, double quantile) {
    std::vector<double> order(p);
    std::sort(order.begin(), order.end());
    if (order.empty() || order[0] >= quantile) return 0.0;
    float acc = 0.f;
    for (size_t i = 1; i < order.size(); ++i) {
        acc = float(double(acc) + order[i]);
        if (acc >= quantile) return order[i - 1] + double(FLT_EPSILON);
    }
    return -1.0;
}

void apply_quantile_clipping(double quantile, std::vector<double>& p) {    // agent.cpp:121-130
    const double thresh = get_quantile(p, quantile);
    double sum = 0;
    for (double& v : p) { if (v < thresh) v = 0.0; sum += v; }              // sequential double sum, as the reference's loop adds it
    for (double& v : p) v /= sum;
}

void sharpen_distribution(std::vector<double>& p, double thresh) {          // blazeutil.h:94-105
    if (p.empty() || *std::max_element(p.begin(), p.end()) < thresh) return;
    double sum = 0;
    for (double& v : p) { if (v < thresh) v = 0.0; sum += v; }
    for (double& v : p) v /= sum;
}

namespace {
size_t sample_index(std::mt19937_64& rng, const std::vector<double>& p) {   // random_choice: inverse CDF on a uniform draw
    double sum = 0;
    for (double v : p) sum += v;
    const double u = std::uniform_real_distribution<double>(0.0, 1.0)(rng) * sum;
    double acc = 0;
    for (size_t i = 0; i < p.size(); ++i) {
        acc += p[i];
        if (u < acc) return i;
    }
    for (size_t i = p.size(); i-- > 0;)
        if (p[i] > 0) return i;
    return 0;
}
double uniform01(std::mt19937_64& rng) { return std::uniform_real_distribution<double>(0.0, 1.0)(rng); }

Position make_position(const std::string& fen, bool is960, chess::Variant v) {
    Position p;
    p.set(fen.empty() ? chess::start_fen(v) : fen, is960, v);
    return p;
}

// play_move_and_update's bookkeeping for one move: SAN with '#' for a decisive end, the result from White's point of view
int result_for_white(const Position& pos, chess::TerminalType t) {
    const bool stm_white = pos.side_to_move() == chess::WHITE;
    if (t == chess::TERMINAL_DRAW) return 0;
    if (t == chess::TERMINAL_LOSS) return stm_white ? -1 : 1;              // the side to move has lost
    return stm_white ? 1 : -1;
}
std::string mark_mate(std::string san, chess::TerminalType t) {
    if (t == chess::TERMINAL_WIN || t == chess::TERMINAL_LOSS) {
        if (!san.empty() && san.back() == '+') san.back() = '#';
        else san += "#";
    }
    return san;
}
chess::TerminalType terminal_of(const Position& pos) {
    std::vector<Move> mv;
    pos.legal_moves(mv);
    return pos.is_terminal(mv.size());
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// self-play
// ---------------------------------------------------------------------------------------------------------------------
SelfPlayDriver::SelfPlayDriver(search::SearchPool* pool, const SelfPlaySettings& s, int concurrent, chess::Variant variant, bool is960,
                               TrainDataExporter* exporter)
    : pool_(pool), s_(s), concurrent_(concurrent), variant_(variant), is960_(is960), exporter_(exporter) {
    if (!pool || concurrent < 1) throw std::invalid_argument("self-play needs a pool and at least one concurrent game");
    if (pool->n_trees() != 0) throw std::invalid_argument("the pool must be empty when the game loop takes it over");
    if (!s.simulations && !s.nodes) throw std::invalid_argument("self-play needs a simulations or a nodes budget");
    if (s.num_phases < 1) throw std::invalid_argument("self-play needs at least one game phase");
    if (s.game_phase_definition == 0 && s.num_phases != 1 && s.num_phases != 3)
        throw std::invalid_argument("the lichess game-phase definition has three phases (board.cpp:544)");
    if (s.game_phase_definition != 0 && s.game_phase_definition != 1) throw std::invalid_argument("game phase definition: 0 lichess, 1 movecount");
    exporters_.assign(size_t(s.num_phases), nullptr);
    exporters_[0] = exporter;
    games_.resize(size_t(concurrent));
    for (int slot = 0; slot < concurrent; ++slot) {
        const int t = pool->add_position(make_position("", is960, variant));
        if (t != slot) throw std::logic_error("tree slots out of order");
        pool->set_active(slot, false);
    }
}

void SelfPlayDriver::finish(Game& g, int result, const char* why) {
    g.rec.result = result;
    g.rec.termination = why;
    if (exporter_) {                                                       // generate_game: samples are written once the result is known
        // new_game on every exporter, every sample to the exporter of its phase, export_game_samples on every exporter (selfplay.cpp:204-206,
        // 232-238,248-250).  The phase COLUMN of a sample is the position's phase under the definition, also with one exporter
        // (save_cur_phase, traindataexporter.cpp:91-103).
        for (TrainDataExporter* e : exporters_) e->new_game();
        for (const Game::Sample& sm : g.samples)
            exporters_[s_.num_phases > 1 ? size_t(sm.phase) : 0]->save_sample(sm.pos, sm.moves, sm.policy.data(), sm.policy.size(), sm.q, sm.phase);
        for (TrainDataExporter* e : exporters_) stats_.samples += e->export_game_samples(result > 0 ? WHITE_WIN : result < 0 ? BLACK_WIN : DRAWN);
    }
    finished_.push_back(std::move(g.rec));
    // the slot's tree sits out the following runs until a new game takes it (a finished game's tree would otherwise be searched to the
    // full budget every round and its visits counted as nodes)
    pool_->set_active(g.slot, false);
    games_[size_t(g.slot)].reset();
}

bool SelfPlayDriver::check_over(Game& g, const std::string* san, const std::string* uci) {
    const chess::TerminalType t = terminal_of(g.pos);
    if (san) {
        g.rec.san.push_back(mark_mate(*san, t));
        g.rec.uci.push_back(*uci);
    }
    if (t == chess::TERMINAL_NONE) return false;
    finish(g, result_for_white(g.pos, t), "terminal");
    return true;
}

std::vector<std::string> read_epd_file(const std::string& path) {
    std::vector<std::string> lines;
    if (path.empty() || path == "<empty>") return lines;                  // load_random_fen, rl/selfplay.cpp:60-62
    std::ifstream f(path);
    if (!f) throw std::invalid_argument("EPD file cannot be read: " + path);
    std::string line;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        if (!line.empty()) lines.push_back(line);
    }
    if (lines.empty()) throw std::invalid_argument("EPD file holds no position: " + path);
    return lines;
}

// refill free slots; init_starting_state_from_raw_policy for all games that start in this round, ply by ply, one batch per ply
void SelfPlayDriver::start_games(size_t n_games) {
    std::vector<Game*> fresh;
    for (int slot = 0; slot < concurrent_; ++slot) {
        if (games_[size_t(slot)] || started_ >= n_games) continue;
        const size_t idx = started_++;
        std::unique_ptr<Game> g(new Game);
        g->slot = slot;
        std::seed_seq seq{uint32_t(s_.seed), uint32_t(s_.seed >> 32), uint32_t(idx), uint32_t(uint64_t(idx) >> 32)};
        g->rng.seed(seq);
        // load position from file if epd filepath was set (rl/selfplay.cpp:200-202)
        g->pos = make_position(!epd_lines_.empty() ? pick_epd_line(epd_lines_, g->rng)
                               : start_fens_.empty() ? std::string() : start_fens_[idx % start_fens_.size()], is960_, variant_);
        g->rec.start_fen = g->pos.fen();
        pool_->reset_position(slot, g->pos);
        pool_->set_active(slot, true);
        if (s_.mean_init_ply > 0) {                                        // plies ~ round(Exp(mean)), clipped (selfplay.cpp:196-197)
            const double e = std::exponential_distribution<double>(1.0 / s_.mean_init_ply)(g->rng);
            int ply = int(e + 0.5);
            // clip_ply (selfplay.cpp:466-472): a draw beyond the maximum is REPLACED by a uniform one in [0, max) -- the tail of the
            // exponential is spread over all opening lengths, not piled onto the longest
            if (ply > s_.max_init_ply) ply = s_.max_init_ply > 0 ? std::uniform_int_distribution<int>(0, s_.max_init_ply - 1)(g->rng) : 0;
            g->opening_left = ply;
            g->in_opening = g->opening_left > 0;
        }
        fresh.push_back(g.get());
        games_[size_t(slot)] = std::move(g);
    }
    for (;;) {
        bool any = false;
        for (Game* g : fresh) any = any || g->in_opening;
        if (!any) break;
        SearchStats st;
        pool_->evaluate_new_roots(&st);                                     // the raw policy of every position that needs one, batched
        stats_.nn_evals += st.nn_evals;
        for (Game* g : fresh) {
            if (!g->in_opening) continue;
            Tree& t = pool_->tree(g->slot);
            const search::Node& root = t.root();
            const size_t n = root.actions.size();
            if (n == 0 || root.terminal) { g->in_opening = false; continue; }
            std::vector<double> p(n, 1.0);
            if (n > 1) {
                for (size_t i = 0; i < n; ++i) p[i] = double(root.priors[i]);
            }
            if (uniform01(g->rng) < s_.raw_policy_prob_temperature) {     // apply_raw_policy_temp (selfplay.cpp:474-488)
                const double u = uniform01(g->rng);
                apply_temperature(p, u < 0.05 ? 10.0 : u < 0.25 ? 5.0 : 2.0);
            }
            const Move mv = root.actions[sample_index(g->rng, p)];
            Position nxt = g->pos;
            nxt.do_move(mv);
            if (terminal_of(nxt) != chess::TERMINAL_NONE) { g->in_opening = false; continue; }     // leads_to_terminal: keep the game alive
            g->rec.san.push_back(g->pos.move_to_san(mv) + " {book}");
            g->rec.uci.push_back(g->pos.move_to_uci(mv));
            g->pos = nxt;
            t.apply_move(mv);                                              // the tree restarts at the new position, move history kept
            if (--g->opening_left == 0) g->in_opening = false;
        }
    }
    for (Game* g : fresh) {
        g->rec.book_plies = int(g->rec.uci.size());
        g->allow_resign = s_.resign_probability >= 0.01 && uniform01(g->rng) < s_.resign_probability;
        check_over(*g, nullptr, nullptr);                                  // a start position can already be decided
    }
}

void SelfPlayDriver::set_phase_exporter(int phase, TrainDataExporter* exporter) {
    if (phase < 1 || phase >= s_.num_phases) throw std::invalid_argument("phase exporter: phase out of range (phase 0 is the constructor's)");
    if (!exporter_) throw std::invalid_argument("phase exporters need the phase-0 exporter");
    exporters_[size_t(phase)] = exporter;
}

size_t SelfPlayDriver::sample_capacity() const { return exporter_ ? exporter_->get_number_samples() : 0; }

size_t SelfPlayDriver::play(size_t n_games, int threads) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool until_full = n_games == 0;                                  // SelfPlay::go(0): `while (generatedSamples < max_samples_per_iteration())`
    if (until_full && !exporter_) throw std::invalid_argument("self-play until the export file is full needs an exporter");
    if (exporter_)
        for (TrainDataExporter* e : exporters_)
            if (!e) throw std::invalid_argument("self-play with several game phases needs an exporter for every phase (set_phase_exporter)");
    // search settings of a normal and of a quick move (update_q_value_weight / update_dirichlet_epsilon, selfplay.cpp:217-222,184-190)
    const search::SearchSettings normal = pool_->settings();
    search::SearchSettings quick = normal;
    quick.q_value_weight = float(s_.quick_search_q_value_weight);
    quick.dirichlet_epsilon = float(s_.quick_dirichlet_epsilon);
    const size_t capacity = sample_capacity();
    for (;;) {
        // go(N): N games; go(0): new games while the sample count is below the file's capacity -- running games are always played out
        const size_t target = until_full ? (samples_taken_ < capacity ? started_ + size_t(concurrent_) : started_) : n_games;
        if (!until_full && finished_.size() >= n_games) break;
        start_games(target);
        std::vector<Game*> active;
        for (auto& g : games_) if (g) active.push_back(g.get());
        if (active.empty()) {
            if (started_ >= target) break;
            continue;                                                      // every fresh game was over at once: refill again
        }
        // every game's limits and settings for THIS move (generate_game's loop head, selfplay.cpp:209-221): a draw for the jitter, a
        // draw for the quick search, quick: nodes = quickSearchNodes + its Q weight and Dirichlet epsilon; then adjust_node_count
        // (:146-152) on whichever node budget applies.  The simulations limit stays as configured (SearchLimits keeps it).
        for (Game* g : active) {
            const uint32_t rand_int = uint32_t(std::uniform_int_distribution<uint32_t>(0, 0x7fffffffu)(g->rng));      // rand()
            g->quick = s_.quick_search_probability >= 0.01 && uniform01(g->rng) < s_.quick_search_probability;
            uint64_t nodes = g->quick ? s_.quick_search_nodes : s_.nodes;
            const uint64_t max_random = uint64_t(double(nodes) * s_.node_random_factor);
            if (max_random != 0) nodes = nodes + (uint64_t(rand_int) % max_random) - max_random / 2;
            uint32_t sims = s_.simulations;
            if (g->quick && !nodes) nodes = 1;                                 // (a quick search without a node budget would be a normal one)
            if (!sims && !nodes) nodes = 1;
            pool_->set_tree_limits(g->slot, sims, uint32_t(nodes));
            search::SearchSettings st = g->quick ? quick : normal;
            st.seed = normal.seed + uint32_t(g->slot);
            pool_->tree(g->slot).set_search_settings(st);
            if (g->quick) ++stats_.quick_searches;
        }
        const uint32_t sims = s_.simulations, nodes = s_.nodes ? s_.nodes : (s_.simulations ? 0u : 1u);
        // which of this round's positions are exported (`!isQuickSearch && generatedSamples < max_samples_per_iteration()`, :224): decided
        // in game order, the shared count is not touched from the parallel step
        std::vector<char> take(active.size(), 0);
        for (size_t gi = 0; gi < active.size(); ++gi) {
            if (!exporter_ || active[gi]->quick) continue;
            if (samples_taken_ < capacity) { take[gi] = 1; ++samples_taken_; }
            else ++stats_.samples_dropped;
        }
        SearchStats st;
        const auto r0 = std::chrono::steady_clock::now();
        pool_->run(sims, nodes, threads, &st);
        stats_.run_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
        stats_.nodes += st.nodes;
        stats_.nn_evals += st.nn_evals;
        // every running game picks and plays its move: the games touch nothing but their own tree, position and generator, so this
        // part runs on the pool's worker threads; what is shared (the list of finished games, the exporter, the counters) follows serially
        struct Played { std::string san, uci; chess::TerminalType term = chess::TERMINAL_NONE; float best_q = 0; bool kept = false; bool moved_tree = false; };
        std::vector<Played> played(active.size());
        const auto m0 = std::chrono::steady_clock::now();
        pool_->parallel_for(int(active.size()), threads, [&](int gi) {
            Game& g = *active[size_t(gi)];
            Played& pl = played[size_t(gi)];
            Tree& t = pool_->tree(g.slot);
            // Agent::set_best_move (agent.cpp:38-55) on the root's MCTS policy
            std::vector<double> policy;
            const int best = t.best_move_index(&policy);
            if (best < 0) throw std::logic_error("self-play: a running game's tree has no searched root");
            pl.best_q = t.eval_best_move_q();
            const size_t ply = g.rec.uci.size();                           // steps_from_null of the game state
            size_t pick;
            if (int(ply) < s_.temperature_moves && s_.init_temperature > 0.01) {
                std::vector<double> p(policy);
                apply_temperature(p, s_.init_temperature * std::pow(s_.temperature_decay, double(ply)));
                if (s_.quantile_clipping != 0) apply_quantile_clipping(s_.quantile_clipping, p);
                pick = sample_index(g.rng, p);
            } else {
                pick = size_t(std::max_element(policy.begin(), policy.end()) - policy.begin());
            }
            const Move mv = t.root().actions[pick];
            if (take[size_t(gi)]) {                                        // save_sample before the move (selfplay.cpp:225-240)
                std::vector<double> exported(policy);
                if (s_.low_policy_clip_threshold > 0) sharpen_distribution(exported, s_.low_policy_clip_threshold);   // the move was picked from the unsharpened policy
                g.samples.push_back(Game::Sample{g.pos, t.root().actions, std::move(exported), pl.best_q,
                                                 g.pos.game_phase(unsigned(s_.num_phases), s_.game_phase_definition)});
            }
            pl.uci = g.pos.move_to_uci(mv);
            pl.san = g.pos.move_to_san(mv);
            g.pos.do_move(mv);
            pl.term = terminal_of(g.pos);
            const bool goes_on = pl.term == chess::TERMINAL_NONE && !(g.allow_resign && pl.best_q < s_.resign_threshold) &&
                                 int(g.rec.uci.size()) + 1 < s_.max_plies;
            if (goes_on) {
                pl.moved_tree = true;
                if (s_.reuse_tree) pl.kept = t.apply_move(mv);
                else pool_->reset_position(g.slot, g.pos);
            }
        });
        stats_.move_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - m0).count();
        for (size_t gi = 0; gi < active.size(); ++gi) {
            Game& g = *active[gi];
            const Played& pl = played[gi];
            ++stats_.moves;
            g.rec.san.push_back(mark_mate(pl.san, pl.term));               // play_move_and_update
            g.rec.uci.push_back(pl.uci);
            // check_for_resignation runs AFTER play_move_and_update and overwrites its verdict (selfplay.cpp:241-243,168-182): a mover whose
            // best move scores below the threshold has resigned even when that move ended the game on the board
            if (g.allow_resign && pl.best_q < s_.resign_threshold) {       // (after the move: the side to move wins)
                finish(g, g.pos.side_to_move() == chess::WHITE ? 1 : -1, "resignation");
                continue;
            }
            if (pl.term != chess::TERMINAL_NONE) {
                finish(g, result_for_white(g.pos, pl.term), "terminal");
                continue;
            }
            if (int(g.rec.uci.size()) >= s_.max_plies) {
                finish(g, 0, "ply limit");
                continue;
            }
            if (pl.kept) ++stats_.kept_subtrees; else ++stats_.restarts;
        }
    }
    stats_.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return finished_.size();
}

// ---------------------------------------------------------------------------------------------------------------------
// arena
// ---------------------------------------------------------------------------------------------------------------------
ArenaDriver::ArenaDriver(search::SearchPool* pool_a, search::SearchPool* pool_b, const SelfPlaySettings& s, int concurrent,
                         chess::Variant variant, bool is960)
    : s_(s), concurrent_(concurrent), variant_(variant), is960_(is960) {
    pools_[0] = pool_a;
    pools_[1] = pool_b;
    if (!pool_a || !pool_b || concurrent < 1) throw std::invalid_argument("an arena needs two pools and at least one concurrent game");
    if (!s.simulations && !s.nodes) throw std::invalid_argument("an arena needs a simulations or a nodes budget");
    games_.resize(size_t(concurrent));
    for (int slot = 0; slot < concurrent; ++slot)
        for (search::SearchPool* p : pools_) {
            if (p->add_position(make_position("", is960, variant)) != slot) throw std::invalid_argument("the pools must be empty when the arena takes them over");
            p->set_active(slot, false);
        }
}

size_t ArenaDriver::play(size_t n_games, int threads) {
    const auto t0 = std::chrono::steady_clock::now();
    while (finished_.size() < n_games) {
        for (int slot = 0; slot < concurrent_; ++slot) {
            Game& g = games_[size_t(slot)];
            if (g.active || started_ >= n_games) continue;
            const size_t idx = started_++;
            const size_t pair = idx / 2;
            if (idx % 2 == 0) {
                std::string fen = start_fens_.empty() ? std::string() : start_fens_[pair % start_fens_.size()];
                if (!epd_lines_.empty()) {                                  // load_random_fen(rlSettings->epdFilePath), rl/selfplay.cpp:396
                    std::seed_seq seq{uint32_t(s_.seed), uint32_t(s_.seed >> 32), uint32_t(pair), 0xE9Du};
                    std::mt19937_64 rng(seq);
                    fen = pick_epd_line(epd_lines_, rng);
                }
                g.pos = make_position(fen, is960_, variant_);
                if (pair_fen_.size() <= pair) pair_fen_.resize(pair + 1);
                pair_fen_[pair] = g.pos.fen();
            } else {
                g.pos = make_position(pair_fen_.at(pair), is960_, variant_);   // gamePGN.fen of the game before
            }
            g.idx = idx;
            g.rec = GameRecord();
            g.rec.start_fen = g.pos.fen();
            g.rec.contender_white = idx % 2 == 0;
            g.active = true;
            for (search::SearchPool* p : pools_) p->reset_position(slot, g.pos);
        }
        bool any = false;
        std::vector<int> mover(size_t(concurrent_), -1);                   // which player searches this game now
        for (int slot = 0; slot < concurrent_; ++slot) {
            const Game& g = games_[size_t(slot)];
            if (!g.active) continue;
            any = true;
            const bool white_to_move = g.pos.side_to_move() == chess::WHITE;
            mover[size_t(slot)] = white_to_move == g.rec.contender_white ? 0 : 1;
        }
        if (!any) break;
        // Both players search AT THE SAME TIME, each pool on its own driving thread with half of the host threads: the games in which
        // the contender is to move and the games in which the baseline is to move are disjoint (colour-swapped pairs make the two sets
        // equally large), every pool has its own nets and streams, and while one pool collects leaves the other pool's batch is on the
        // GPU.  (One pool after the other left the chip idle for every collection step of either: 331k nodes/s where the same trees
        // searched without a game around them gave 1.18M.)
        bool has[2] = {false, false};
        for (int pi = 0; pi < 2; ++pi)
            for (int slot = 0; slot < concurrent_; ++slot) {
                const bool on = mover[size_t(slot)] == pi;
                pools_[pi]->set_active(slot, on);
                has[pi] = has[pi] || on;
            }
        SearchStats st[2];
        const auto r0 = std::chrono::steady_clock::now();
        const bool concurrent_pools = threads >= 2 && getenv("CRA_ARENA_SERIAL") == nullptr;     // CRA_ARENA_SERIAL=1: one pool after the other (A/B)
        // each pool keeps ONE worker count for the whole arena (its searches and the move step below): a pool rebuilds its worker
        // threads whenever it is asked for a different number
        const int tb = concurrent_pools ? threads / 2 : threads, ta = concurrent_pools ? threads - tb : threads;
        if (has[0] && has[1] && concurrent_pools) {
            std::exception_ptr err;
            std::thread other([&] {
                try { pools_[1]->run(s_.simulations, s_.simulations ? 0 : s_.nodes, tb, &st[1]); } catch (...) { err = std::current_exception(); }
            });
            try {
                pools_[0]->run(s_.simulations, s_.simulations ? 0 : s_.nodes, ta, &st[0]);
            } catch (...) {
                other.join();
                throw;
            }
            other.join();
            if (err) std::rethrow_exception(err);
        } else {
            for (int pi = 0; pi < 2; ++pi)
                if (has[pi]) pools_[pi]->run(s_.simulations, s_.simulations ? 0 : s_.nodes, pi == 0 ? ta : tb, &st[pi]);
        }
        stats_.run_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
        for (int pi = 0; pi < 2; ++pi) {
            stats_.nodes += st[pi].nodes;
            stats_.nn_evals += st[pi].nn_evals;
        }
        // the games play their moves on the worker threads (a game touches its own position and its tree slot in both pools)
        std::vector<chess::TerminalType> terms(size_t(concurrent_), chess::TERMINAL_NONE);
        std::vector<char> over(size_t(concurrent_), 0);
        const auto m0 = std::chrono::steady_clock::now();
        pools_[0]->parallel_for(concurrent_, ta, [&](int slot) {
            Game& g = games_[size_t(slot)];
            if (!g.active) return;
            Tree& t = pools_[mover[size_t(slot)]]->tree(slot);
            std::vector<double> policy;
            if (t.best_move_index(&policy) < 0) throw std::logic_error("arena: a running game's tree has no searched root");
            const Move mv = t.root().actions[size_t(std::max_element(policy.begin(), policy.end()) - policy.begin())];
            const std::string uci = g.pos.move_to_uci(mv), san = g.pos.move_to_san(mv);
            g.pos.do_move(mv);
            const chess::TerminalType term = terminal_of(g.pos);
            terms[size_t(slot)] = term;
            g.rec.san.push_back(mark_mate(san, term));
            g.rec.uci.push_back(uci);
            over[size_t(slot)] = term != chess::TERMINAL_NONE || int(g.rec.uci.size()) >= s_.max_plies;
            if (!over[size_t(slot)])
                for (search::SearchPool* p : pools_) p->tree(slot).apply_move(mv);   // own move in one tree, the opponent's move in the other
        });
        stats_.move_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - m0).count();
        for (int slot = 0; slot < concurrent_; ++slot) {
            Game& g = games_[size_t(slot)];
            if (!g.active) continue;
            ++stats_.moves;
            if (!over[size_t(slot)]) continue;
            const chess::TerminalType term = terms[size_t(slot)];
            g.rec.result = term == chess::TERMINAL_NONE ? 0 : result_for_white(g.pos, term);
            g.rec.termination = term != chess::TERMINAL_NONE ? "terminal" : "ply limit";
            const int a_score = g.rec.contender_white ? g.rec.result : -g.rec.result;
            if (a_score > 0) ++wins_; else if (a_score < 0) ++losses_; else ++draws_;
            finished_.push_back(g.rec);
            g.active = false;
            for (search::SearchPool* p : pools_) p->set_active(slot, false);
        }
    }
    stats_.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return finished_.size();
}

// ---------------------------------------------------------------------------------------------------------------------
std::string game_pgn(const GameRecord& g, const std::string& variant_tag, const std::string& event, const std::string& white,
                     const std::string& black, const std::string& date) {
    static const char* kResult[3] = {"0-1", "1/2-1/2", "1-0"};
    const std::string res = kResult[g.result + 1];
    std::string out;
    out += "[Variant \"" + variant_tag + "\"]\n[Event \"" + event + "\"]\n[Date \"" + date + "\"]\n[Site \"MI355X\"]\n[Round \"?\"]\n";
    out += "[FEN \"" + g.start_fen + "\"]\n[White \"" + white + "\"]\n[Black \"" + black + "\"]\n[Result \"" + res + "\"]\n";
    out += "[PlyCount \"" + std::to_string(g.san.size()) + "\"]\n[TimeControl \"-\"]\n\n";
    for (size_t ply = 0; ply < g.san.size(); ++ply) {
        if (ply % 2 == 0) out += std::to_string(ply / 2 + 1) + ". ";
        out += g.san[ply] + " ";
        if ((ply + 1) % 8 == 0) out += "\n";
    }
    out += res + "\n\n";
    return out;
}

}  // namespace rl
}  // namespace cra
