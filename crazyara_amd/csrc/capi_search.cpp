// C ABI: MCTS search pool -- see include/crazyara_hip.h.
#include "../../include/crazyara_hip.h"

#include <malloc.h>

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>

#include "capi_common.h"
#include "capi_net.h"
#include "capi_search_handle.h"
#include "capi_traindata.h"
#include "search/pool.h"

using namespace cra;
using namespace cra::search;

namespace {
SearchSettings convert(const mi_search_settings& m) {
    SearchSettings s;
    s.batch_size = m.batch_size;
    s.cpuct_init = m.cpuct_init;
    s.cpuct_base = m.cpuct_base;
    s.node_policy_temperature = m.node_policy_temperature;
    s.virtual_style = m.virtual_style;
    s.virtual_mix_threshold = m.virtual_mix_threshold;
    s.virtual_offset_strength = m.virtual_offset_strength;
    s.q_value_weight = m.q_value_weight;
    s.q_veto_delta = m.q_veto_delta;
    s.mode = m.mode;
    s.version_major = m.version_major;
    s.version_minor = m.version_minor;
    s.is_policy_map = m.is_policy_map != 0;
    s.clone_keeps_last_moves = m.clone_keeps_last_moves;
    s.epsilon_greedy_counter = m.epsilon_greedy_counter;
    s.epsilon_checks_counter = m.epsilon_checks_counter;
    s.seed = m.seed;
    s.mcts_solver = m.mcts_solver != 0;
    s.dirichlet_epsilon = m.dirichlet_epsilon;
    s.dirichlet_alpha = m.dirichlet_alpha;
    return s;
}
}  // namespace

extern "C" {

void mi_search_default_settings(mi_search_settings* m) {
    if (!m) return;
    const SearchSettings s;
    m->batch_size = s.batch_size;
    m->cpuct_init = s.cpuct_init;
    m->cpuct_base = s.cpuct_base;
    m->node_policy_temperature = s.node_policy_temperature;
    m->virtual_style = s.virtual_style;
    m->virtual_mix_threshold = s.virtual_mix_threshold;
    m->virtual_offset_strength = s.virtual_offset_strength;
    m->q_value_weight = s.q_value_weight;
    m->q_veto_delta = s.q_veto_delta;
    m->mode = s.mode;
    m->version_major = s.version_major;
    m->version_minor = s.version_minor;
    m->is_policy_map = s.is_policy_map ? 1 : 0;
    m->clone_keeps_last_moves = s.clone_keeps_last_moves;
    m->epsilon_greedy_counter = s.epsilon_greedy_counter;
    m->epsilon_checks_counter = s.epsilon_checks_counter;
    m->seed = s.seed;
    m->mcts_solver = s.mcts_solver ? 1 : 0;
    m->dirichlet_epsilon = s.dirichlet_epsilon;
    m->dirichlet_alpha = s.dirichlet_alpha;
}

// Many threads growing their trees at once extend glibc's heaps in 128 KiB steps by default; every extension takes the process's
// mmap lock for writing and stalls the page faults of all other threads (measured on the 2 x 64-core host of the GPU box: 16 trees
// collected in parallel ran 2.5x slower per leaf than one alone, and exactly as fast as one alone with this setting).  The pad is
// address space, not resident memory.  CRA_NO_MALLOPT=1 leaves the allocator alone.
static void widen_heap_growth_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("CRA_NO_MALLOPT") == nullptr) (void)mallopt(M_TOP_PAD, 256 << 20);
    });
}

mi_search* mi_search_create(const mi_search_settings* s, mi_net* net_a, mi_net* net_b, mi_eval_fn fn, void* user, int fn_batch, int fn_nb_policy) {
    mi_search* h = nullptr;
    if (cra_guard([&] {
            if (!s) throw std::invalid_argument("null settings");
            widen_heap_growth_once();
            std::unique_ptr<Evaluator> a, b;
            if (fn) {
                if (fn_batch <= 0 || fn_nb_policy <= 0) throw std::invalid_argument("callback lane needs batch and nb_policy");
                a = make_callback_evaluator(fn, user, fn_batch, fn_nb_policy);
            } else {
                if (!net_a) throw std::invalid_argument("mi_search_create needs a net or an evaluator callback");
                a = make_hip_evaluator(&net_a->net);
                if (net_b) b = make_hip_evaluator(&net_b->net);
            }
            h = new mi_search;
            h->pool.reset(new SearchPool(convert(*s), std::move(a), std::move(b)));
        })) {
        delete h;
        return nullptr;
    }
    return h;
}

int mi_search_add_lane(mi_search* sp, mi_net* net) {
    if (!sp || !net) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] { sp->pool->add_lane(make_hip_evaluator(&net->net)); });
}

void mi_search_destroy(mi_search* sp) { delete sp; }

int mi_search_add_position(mi_search* sp, const char* fen, int is_chess960, const char* variant) {
    int id = -1;
    if (!sp) return id;
    cra_guard([&] {
        const chess::Variant v = chess::variant_from_name(variant && *variant ? variant : "chess");
        chess::Position p;
        p.set(fen && *fen ? std::string(fen) : chess::start_fen(v), is_chess960 != 0, v);
        id = sp->pool->add_position(p);
    });
    return id;
}

int mi_search_apply_move(mi_search* sp, int tree, const char* uci, int* kept) {
    if (!sp || !uci) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        Tree& t = sp->pool->tree(tree);
        const chess::Move m = t.root_position().uci_to_move(uci);
        if (m == chess::MOVE_NONE) throw std::invalid_argument(std::string("not a legal move here: ") + uci);
        const bool k = t.apply_move(m);
        if (kept) *kept = k ? 1 : 0;
    });
}

int mi_search_tree_fen(mi_search* sp, int tree, char* fen, int cap) {
    if (!sp || !fen) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        const std::string s = sp->pool->tree(tree).root_position().fen();
        if (int(s.size()) + 1 > cap) throw std::invalid_argument("fen buffer too small");
        std::memcpy(fen, s.c_str(), s.size() + 1);
    });
}

int mi_search_announce_go(mi_search* sp) {
    if (!sp) { cra_set_error("null search"); return 1; }
    sp->pool->announce_go();
    return 0;
}

int mi_search_cancel_go(mi_search* sp) {
    if (!sp) { cra_set_error("null search"); return -1; }
    return sp->pool->cancel_go() ? 1 : 0;
}

int mi_search_stop(mi_search* sp) {
    if (!sp) { cra_set_error("null search"); return 1; }
    sp->pool->request_stop();          // a short mutex + one atomic store: safe beside a running mi_search_run of another thread
    return 0;
}

int mi_search_run(mi_search* sp, unsigned simulations, unsigned nodes, int threads, mi_search_stats* stats) {
    return mi_search_run_timed(sp, simulations, nodes, 0, threads, stats);
}

int mi_search_run_timed(mi_search* sp, unsigned simulations, unsigned nodes, unsigned movetime_ms, int threads, mi_search_stats* stats) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] {
        SearchStats st;
        sp->pool->run(simulations, nodes, threads, &st, movetime_ms);
        if (stats) {
            stats->nodes = st.nodes;
            stats->nn_evals = st.nn_evals;
            stats->batches = st.batches;
            stats->simulations = st.simulations;
            stats->seconds = st.seconds;
            stats->depth_avg = st.depth_avg;
            stats->depth_max = st.depth_max;
        }
    });
}

int mi_search_root_children(mi_search* sp, int tree, int cap, uint32_t* moves, uint32_t* visits, float* q, float* priors) {
    int n = -1;
    if (!sp) return n;
    cra_guard([&] {
        const Node& r = sp->pool->tree(tree).root();
        n = r.has_data ? int(r.child_visits.size()) : 0;
        for (int i = 0; i < n && i < cap; ++i) {
            if (moves) moves[i] = r.actions[i];
            if (visits) visits[i] = r.child_visits[i];
            if (q) q[i] = r.q[i];
            if (priors) priors[i] = r.priors[i];
        }
    });
    return n;
}

int mi_search_tree_info(mi_search* sp, int tree, unsigned* root_visits, unsigned* node_count, unsigned* allocated_nodes, float* root_value) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] {
        Tree& t = sp->pool->tree(tree);
        if (root_visits) *root_visits = t.root_visits();
        if (node_count) *node_count = t.node_count();
        if (allocated_nodes) *allocated_nodes = unsigned(t.node_count_allocated());
        if (root_value) *root_value = t.root().real_visits ? t.root().value() : 0.0f;
    });
}

int mi_search_root_solved(mi_search* sp, int tree, int* node_type, int* end_in_ply, int* checkmate_idx) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] {
        const Tree& t = sp->pool->tree(tree);
        if (node_type) *node_type = t.root().node_type;
        if (end_in_ply) *end_in_ply = t.root().end_in_ply;
        if (checkmate_idx) *checkmate_idx = t.root().checkmate_idx;
    });
}

int mi_search_root_policy(mi_search* sp, int tree, int cap, double* policy, float* best_move_q) {
    int n = -1;
    if (!sp) { cra_set_error("null search"); return n; }
    cra_guard([&] {
        Tree& t = sp->pool->tree(tree);
        std::vector<double> pol;
        const int b = t.best_move_index(&pol);
        if (b < 0) { n = 0; return; }
        if (int(pol.size()) > cap) throw std::invalid_argument("policy buffer too small");
        if (policy) std::copy(pol.begin(), pol.end(), policy);
        if (best_move_q) *best_move_q = t.eval_best_move_q();
        n = int(pol.size());
    });
    return n;
}

int mi_time_for_move(const mi_go_limits* l, int side, int move_number) {
    if (!l || side < 0 || side > 1) { cra_set_error("mi_time_for_move: null limits or side not 0 / 1"); return -1; }
    constexpr int EXPECTED_GAME_LENGTH = 38, THRESH_MOVE = 35, PROP_MOVES_TO_GO = 14, BUFFER_FACTOR = 30;   // constants.h:94-98
    constexpr float INCREMENT_FACTOR = 0.7f;
    if (l->infinite) return 0;
    if ((l->nodes != 0 || l->simulations != 0 || l->depth != 0) && l->movetime == 0) return 0;
    const int safe_remaining = std::max(l->time[side] - l->move_overhead * BUFFER_FACTOR, 1);              // get_safe_remaining_time
    auto constant_movetime = [&](int moves_to_go) { return int(safe_remaining / moves_to_go + INCREMENT_FACTOR * l->inc[side]); };
    int cur;
    if (l->movetime != 0) cur = int(l->movetime);
    else if (l->movestogo != 0) cur = constant_movetime(l->movestogo);
    else if (l->time[side] != 0) cur = move_number < THRESH_MOVE ? constant_movetime(EXPECTED_GAME_LENGTH - move_number) : constant_movetime(PROP_MOVES_TO_GO);
    else cur = 1000;                                                                                        // "No limit specification given"
    cur -= l->move_overhead;
    if (cur <= 0) cur = l->move_overhead * 2;
    if (l->time[side] != 0) return std::min(safe_remaining, cur);
    return cur;
}

int mi_search_pv(mi_search* sp, int tree, char* uci_line, int cap, int* centipawns, int* moves_to_mate) {
    int n = -1;
    if (!sp || !uci_line) { cra_set_error("null argument"); return n; }
    cra_guard([&] {
        Tree& t = sp->pool->tree(tree);
        std::vector<chess::Move> pv;
        t.principal_variation(pv, moves_to_mate, centipawns);
        chess::Position pos = t.root_position();
        std::string line;
        for (chess::Move m : pv) {
            if (!line.empty()) line += ' ';
            line += pos.move_to_uci(m);
            pos.do_move(m);
        }
        if (int(line.size()) + 1 > cap) throw std::invalid_argument("pv buffer too small");
        std::memcpy(uci_line, line.c_str(), line.size() + 1);
        n = int(pv.size());
    });
    return n;
}

int mi_search_pv_multi(mi_search* sp, int tree, int idx, int multipv, char* uci_line, int cap, int* centipawns, int* moves_to_mate, float* best_move_q) {
    int n = -1;
    if (!sp || !uci_line) { cra_set_error("null argument"); return n; }
    cra_guard([&] {
        Tree& t = sp->pool->tree(tree);
        std::vector<chess::Move> pv;
        if (cap > 0) uci_line[0] = 0;
        if (!t.principal_variation_multi(idx, multipv, pv, moves_to_mate, centipawns, best_move_q)) { n = 0; return; }
        chess::Position pos = t.root_position();
        std::string line;
        for (chess::Move m : pv) {
            if (!line.empty()) line += ' ';
            line += pos.move_to_uci(m);
            pos.do_move(m);
        }
        if (int(line.size()) + 1 > cap) throw std::invalid_argument("pv buffer too small");
        std::memcpy(uci_line, line.c_str(), line.size() + 1);
        n = int(pv.size());
    });
    return n;
}

long mi_search_debug_replay(mi_search* sp, char* report, long cap) {
    long n = -1;
    if (!sp) { cra_set_error("null search"); return n; }
    cra_guard([&] {
        std::string text;
        const size_t bad = sp->pool->debug_replay(&text);
        if (report && cap > 0) {
            const size_t m = std::min(text.size(), size_t(cap - 1));
            std::memcpy(report, text.data(), m);
            report[m] = 0;
        }
        n = long(bad);
    });
    return n;
}

long mi_search_tree_dump(mi_search* sp, int tree, uint32_t* out, long cap) {
    long n = -1;
    if (!sp) { cra_set_error("null search"); return n; }
    cra_guard([&] {
        std::vector<uint32_t> words;
        sp->pool->tree(tree).dump(words);
        if (long(words.size()) > cap) throw std::invalid_argument("tree dump buffer too small");
        if (out) std::copy(words.begin(), words.end(), out);
        n = long(words.size());
    });
    return n;
}

// save_sample(state, evalInfo) of the self-play loop (selfplay.cpp:150-160) taken straight from a searched tree: the root position,
// its legal moves in the root's order, EvalInfo::policyProbSmall (Node::get_mcts_policy, unexpanded moves 0) and bestMoveQ
int mi_search_save_sample(mi_search* sp, int tree, mi_traindata* td) {
    if (!sp || !td) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        Tree& t = sp->pool->tree(tree);
        std::vector<double> pol;
        const int b = t.best_move_index(&pol);
        if (b < 0) throw std::logic_error("mi_search_save_sample: the tree has not been searched");
        td->exp.save_sample(t.root_position(), t.root().actions, pol.data(), pol.size(), t.eval_best_move_q());
    });
}

int mi_search_set_shared_collectors(mi_search* sp, int k) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] { sp->pool->set_shared_collectors(k); });
}

int mi_search_set_state_budget(mi_search* sp, unsigned budget) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] { sp->pool->set_state_budget(budget); });
}

int mi_search_set_adaptive_quota(mi_search* sp, int cap) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] { sp->pool->set_adaptive_quota(cap); });
}

int mi_search_set_active(mi_search* sp, int tree, int active) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] { sp->pool->set_active(tree, active != 0); });
}

int mi_search_reset_position(mi_search* sp, int tree, const char* fen, int is_chess960, const char* variant) {
    if (!sp) { cra_set_error("null search"); return 1; }
    return cra_guard([&] {
        const chess::Variant v = chess::variant_from_name(variant && *variant ? variant : "chess");
        chess::Position p;
        p.set(fen && *fen ? std::string(fen) : chess::start_fen(v), is_chess960 != 0, v);
        sp->pool->reset_position(tree, p);
    });
}

int mi_search_best_move(mi_search* sp, int tree, char* uci, int cap) {
    if (!sp || !uci) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        Tree& t = sp->pool->tree(tree);
        const int b = t.best_move_index();
        const std::string s = b < 0 ? "(none)" : t.root_position().move_to_uci(t.root().actions[b]);
        if (int(s.size()) >= cap) throw std::invalid_argument("buffer too small");
        std::memcpy(uci, s.c_str(), s.size() + 1);
    });
}

}  // extern "C"
