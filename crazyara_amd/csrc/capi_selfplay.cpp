// C ABI: native self-play / arena loops -- see include/crazyara_hip.h.
#include "../../include/crazyara_hip.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>

#include "capi_common.h"
#include "capi_search_handle.h"
#include "capi_traindata.h"
#include "rl/selfplay.h"

using namespace cra;

struct mi_selfplay {
    std::unique_ptr<rl::SelfPlayDriver> self;
    std::unique_ptr<rl::ArenaDriver> arena;
    const std::vector<rl::GameRecord>& finished() const { return self ? self->finished() : arena->finished(); }
};

extern "C" {

void mi_selfplay_default_settings(mi_selfplay_settings* m) {
    if (!m) return;
    const rl::SelfPlaySettings d;
    m->simulations = d.simulations;
    m->nodes = d.nodes;
    m->node_random_factor = float(d.node_random_factor);
    m->mean_init_ply = float(d.mean_init_ply);
    m->max_init_ply = d.max_init_ply;
    m->raw_policy_prob_temperature = float(d.raw_policy_prob_temperature);
    m->init_temperature = float(d.init_temperature);
    m->temperature_moves = d.temperature_moves;
    m->temperature_decay = float(d.temperature_decay);
    m->quantile_clipping = float(d.quantile_clipping);
    m->resign_probability = float(d.resign_probability);
    m->resign_threshold = float(d.resign_threshold);
    m->reuse_tree = d.reuse_tree ? 1 : 0;
    m->max_plies = d.max_plies;
    m->seed = d.seed;
    m->quick_search_probability = float(d.quick_search_probability);
    m->quick_search_nodes = d.quick_search_nodes;
    m->quick_search_q_value_weight = float(d.quick_search_q_value_weight);
    m->quick_dirichlet_epsilon = float(d.quick_dirichlet_epsilon);
    m->low_policy_clip_threshold = float(d.low_policy_clip_threshold);
    m->num_phases = d.num_phases;
    m->game_phase_definition = d.game_phase_definition;
}

mi_selfplay* mi_selfplay_create(mi_search* pool_a, mi_search* pool_b, const mi_selfplay_settings* m, int concurrent, const char* variant,
                                int is_chess960, mi_traindata* exporter) {
    if (!pool_a || !m) { cra_set_error("null argument to mi_selfplay_create"); return nullptr; }
    mi_selfplay* out = nullptr;
    cra_guard([&] {
        rl::SelfPlaySettings s;
        s.simulations = m->simulations;
        s.nodes = m->nodes;
        s.node_random_factor = m->node_random_factor;
        s.mean_init_ply = m->mean_init_ply;
        s.max_init_ply = m->max_init_ply;
        s.raw_policy_prob_temperature = m->raw_policy_prob_temperature;
        s.init_temperature = m->init_temperature;
        s.temperature_moves = m->temperature_moves;
        s.temperature_decay = m->temperature_decay;
        s.quantile_clipping = m->quantile_clipping;
        s.resign_probability = m->resign_probability;
        s.resign_threshold = m->resign_threshold;
        s.reuse_tree = m->reuse_tree != 0;
        s.max_plies = m->max_plies;
        s.seed = m->seed;
        s.quick_search_probability = m->quick_search_probability;
        s.quick_search_nodes = m->quick_search_nodes;
        s.quick_search_q_value_weight = m->quick_search_q_value_weight;
        s.quick_dirichlet_epsilon = m->quick_dirichlet_epsilon;
        s.low_policy_clip_threshold = m->low_policy_clip_threshold;
        s.num_phases = m->num_phases;
        s.game_phase_definition = m->game_phase_definition;
        const chess::Variant v = chess::variant_from_name(variant && *variant ? variant : "chess");
        std::unique_ptr<mi_selfplay> sp(new mi_selfplay);
        if (pool_b) {
            if (exporter) throw std::invalid_argument("an arena does not export training samples");
            sp->arena.reset(new rl::ArenaDriver(pool_a->pool.get(), pool_b->pool.get(), s, concurrent, v, is_chess960 != 0));
        } else {
            sp->self.reset(new rl::SelfPlayDriver(pool_a->pool.get(), s, concurrent, v, is_chess960 != 0, exporter ? &exporter->exp : nullptr));
        }
        out = sp.release();
    });
    return out;
}

void mi_selfplay_destroy(mi_selfplay* sp) { delete sp; }

int mi_selfplay_set_start_fens(mi_selfplay* sp, const char* fens) {
    if (!sp || !fens) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        std::vector<std::string> list;
        std::stringstream ss(fens);
        std::string line;
        while (std::getline(ss, line)) list.push_back(line);
        if (sp->self) sp->self->set_start_fens(std::move(list));
        else sp->arena->set_start_fens(std::move(list));
    });
}

int mi_selfplay_set_epd_file(mi_selfplay* sp, const char* path) {
    if (!sp || !path) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        if (sp->self) sp->self->set_epd_file(path);
        else sp->arena->set_epd_file(path);
    });
}

int mi_selfplay_set_phase_exporter(mi_selfplay* sp, int phase, mi_traindata* exporter) {
    if (!sp || !exporter) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        if (!sp->self) throw std::invalid_argument("an arena does not export training samples");
        sp->self->set_phase_exporter(phase, &exporter->exp);
    });
}

int mi_selfplay_play(mi_selfplay* sp, int n_games, int threads) {
    if (!sp || n_games < 0) { cra_set_error("bad argument to mi_selfplay_play"); return -1; }
    int n = -1;
    cra_guard([&] { n = int(sp->self ? sp->self->play(size_t(n_games), threads) : sp->arena->play(size_t(n_games), threads)); });
    return n;
}

long mi_selfplay_game(mi_selfplay* sp, int index, int* result, int* book_plies, int* contender_white, char* text, long cap) {
    if (!sp) { cra_set_error("null self-play handle"); return -1; }
    long n = -1;
    cra_guard([&] {
        const rl::GameRecord& g = sp->finished().at(size_t(index));
        if (result) *result = g.result;
        if (book_plies) *book_plies = g.book_plies;
        if (contender_white) *contender_white = g.contender_white ? 1 : 0;
        std::string s = g.start_fen + "\n" + g.termination + "\n";
        for (size_t i = 0; i < g.san.size(); ++i) s += (i ? "\t" : "") + g.san[i];
        s += "\n";
        for (size_t i = 0; i < g.uci.size(); ++i) s += (i ? "\t" : "") + g.uci[i];
        s += "\n";
        if (text && cap > 0) {
            if (long(s.size()) >= cap) throw std::invalid_argument("game text buffer too small");
            std::memcpy(text, s.c_str(), s.size() + 1);
        }
        n = long(s.size());
    });
    return n;
}

int mi_selfplay_get_stats(mi_selfplay* sp, mi_selfplay_stats* out) {
    if (!sp || !out) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        const rl::LoopStats& st = sp->self ? sp->self->stats() : sp->arena->stats();
        out->moves = st.moves;
        out->nodes = st.nodes;
        out->nn_evals = st.nn_evals;
        out->kept_subtrees = st.kept_subtrees;
        out->restarts = st.restarts;
        out->samples = st.samples;
        out->seconds = st.seconds;
        out->wins = sp->arena ? sp->arena->wins() : 0;
        out->draws = sp->arena ? sp->arena->draws() : 0;
        out->losses = sp->arena ? sp->arena->losses() : 0;
        out->reserved = 0;
        out->run_seconds = st.run_seconds;
        out->move_seconds = st.move_seconds;
        out->quick_searches = st.quick_searches;
        out->samples_dropped = st.samples_dropped;
    });
}

void mi_policy_apply_temperature(double* p, int n, double temperature) {
    std::vector<double> v(p, p + n);
    rl::apply_temperature(v, temperature);
    std::copy(v.begin(), v.end(), p);
}
double mi_policy_get_quantile(const double* p, int n, double quantile) { return rl::get_quantile(std::vector<double>(p, p + n), quantile); }
void mi_policy_sharpen_distribution(double* p, int n, double thresh) {
    std::vector<double> v(p, p + n);
    rl::sharpen_distribution(v, thresh);
    std::copy(v.begin(), v.end(), p);
}
void mi_policy_apply_quantile_clipping(double* p, int n, double quantile) {
    std::vector<double> v(p, p + n);
    rl::apply_quantile_clipping(quantile, v);
    std::copy(v.begin(), v.end(), p);
}

}  // extern "C"
