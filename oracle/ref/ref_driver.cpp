// ORACLE (test infrastructure only): C entry points over the REFERENCE'S OWN search code, compiled from /root/reference by
// oracle/ref/build_ref.py (MCTSAgent / SearchThread / Node / NodeData / EvalInfo / NeuralNetAPI, unmodified, see
// shim/pommermanstate.h for what is theirs and what is ours).  tests/test_mcts_reference_build.py drives the same searches
// through this library and through the product (`mi_search_*`) with one evaluator callback and compares the trees bit for bit.
//
// Everything an engine's UCI loop does around a `go` is done here the way engine/src/uci/crazyara.cpp does it:
//   settings   crazyara.cpp:731-803 (init_search_settings / init_play_settings) -- every field set explicitly from ref_settings
//   nets       crazyara.cpp:548-563: one batch-1 net for the agent's root evaluation, one batch-N net per SearchThread
//   go         crazyara.cpp:203-230: agent->set_search_settings(state, limits, evalInfo); agent->perform_action()
//   move made  run_agent_thread (agent.cpp:107-112) + CrazyAra::position (apply_move_to_tree for both sides)
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "agents/mctsagent.h"
#include "util/blazeutil.h"
#include "pommermanstate.h"

#include "../../include/crazyara_hip.h"      // mi_search_settings (one settings struct for both sides of the comparison), mi_eval_fn

#ifdef REF_WITH_HIPAPI
#include "../../integration/hipapi.h"        // the reference-side binding of the product library, compiled against the reference's base class
#endif

// ---------------------------------------------------------------------------------------------------------------------
// environment-side globals of the shim
// ---------------------------------------------------------------------------------------------------------------------
namespace refshim {
Config& config() {
    static Config c;
    return c;
}
std::vector<cra::BoardDesc>& pending_descs() {
    static thread_local std::vector<cra::BoardDesc> v;
    return v;
}
}  // namespace refshim

std::string StateConstantsPommerman::action_to_uci(Action action, bool is960) {
    using namespace cra::chess;
    const Move m = Move(action);
    static const char* pc = " PNBRQK";
    auto sq = [](int s) { return std::string{char('a' + (s & 7)), char('1' + (s >> 3))}; };
    if (kind_of(m) == DROP) return std::string{pc[piece_of(m)], '@'} + sq(to_sq(m));
    int from = from_sq(m), to = to_sq(m);
    if (kind_of(m) == CASTLING && !is960) to = (from & 56) + (to > from ? 6 : 2);
    std::string s = sq(from) + sq(to);
    if (kind_of(m) == PROMOTION) s += char(pc[piece_of(m)] + 32);
    return s;
}

MoveIdx StateConstantsPommerman::lookup(Action action, bool mirror, bool policy_map) {
    using namespace cra::chess;
    const PolicyTables& t = policy_tables(refshim::config().mode);
    const Move m = Move(action);
    const int flip = mirror ? 56 : 0;
    int li;
    if (kind_of(m) == DROP) {
        li = t.drop[piece_of(m) - PAWN][to_sq(m) ^ flip];
    } else {
        int from = from_sq(m), to = to_sq(m);
        if (kind_of(m) == CASTLING && !refshim::config().is960) to = (from & 56) + (to > from ? 6 : 2);   // sfutil.cpp:243-285
        from ^= flip;
        to ^= flip;
        li = kind_of(m) == PROMOTION ? t.promo[from][to][piece_of(m) - KNIGHT] : t.normal[from][to];
    }
    if (li < 0) throw std::logic_error("move without a policy label");
    return MoveIdx(policy_map ? t.flat_plane_idx[li] : li);
}

// ---------------------------------------------------------------------------------------------------------------------
// rand(): the reference's epsilon exploration draws from rand() (searchthread.cpp:126,170,177,499), seeded with the clock by
// TimeManager (timemanager.cpp:42-43).  The library is linked -Bsymbolic so that those calls bind to this definition: the
// ANSI-C example generator, the same one every product tree owns (csrc/search/mcts.cpp next_rand) -- searches replay.
// ---------------------------------------------------------------------------------------------------------------------
static uint32_t g_rand_state = 1;
extern "C" int rand(void) {
    g_rand_state = g_rand_state * 1103515245u + 12345u;
    return int((g_rand_state >> 16) & 0x7fffu);
}
extern "C" void srand(unsigned seed) { (void)seed; }      // TimeManager's clock seed is ignored; ref_agent_seed sets the state

void refshim_seed_node_generator(unsigned seed);          // node_tu.cpp: the std::default_random_engine of node.cpp's translation unit

// ---------------------------------------------------------------------------------------------------------------------
// evaluator: a NeuralNetAPI that hands the batch to a callback (the product's mi_eval_fn signature)
// ---------------------------------------------------------------------------------------------------------------------
class CallbackNet : public NeuralNetAPI
{
    mi_eval_fn fn;
    void* user;
    int channels, nbPolicy;

    void load_model() override {}
    void init_nn_design() override {
        nnDesign.inputShape.nbDims = 4;
        nnDesign.inputShape.v[0] = int(batchSize);
        nnDesign.inputShape.v[1] = channels;
        nnDesign.inputShape.v[2] = 8;
        nnDesign.inputShape.v[3] = 8;
        nnDesign.valueOutputShape.nbDims = 2;
        nnDesign.valueOutputShape.v[0] = int(batchSize);
        nnDesign.valueOutputShape.v[1] = 1;
        nnDesign.policyOutputShape.nbDims = 2;
        nnDesign.policyOutputShape.v[0] = int(batchSize);
        nnDesign.policyOutputShape.v[1] = nbPolicy;
        nnDesign.auxiliaryOutputShape.nbDims = 2;
        nnDesign.auxiliaryOutputShape.v[0] = int(batchSize);
        nnDesign.auxiliaryOutputShape.v[1] = 0;
        nnDesign.hasAuxiliaryOutputs = false;
        nnDesign.isPolicyMap = unsigned(nbPolicy) != StateConstants::NB_LABELS();          // tensorrtapi.cpp:157
    }
    void load_parameters() override {}
    void bind_executor() override {}

public:
    size_t calls = 0, evals = 0;
    CallbackNet(unsigned batch, int channels, int nbPolicy, const std::string& name, mi_eval_fn fn, void* user) :
        NeuralNetAPI("cpu", 0, batch, "/refnet/", false), fn(fn), user(user), channels(channels), nbPolicy(nbPolicy)
    {
        modelName = name;
        initialize();
    }
    void predict(float* inputPlanes, float* valueOutput, float* probOutputs, float* auxiliaryOutputs) override {
        (void)inputPlanes;
        (void)auxiliaryOutputs;
        std::vector<cra::BoardDesc>& d = refshim::pending_descs();
        ++calls;
        evals += d.size();
        if (!d.empty() && fn(user, d.data(), int(d.size()), valueOutput, probOutputs) != 0)
            throw std::runtime_error("evaluator callback failed");
        d.clear();
    }
};

struct ref_agent {
    SearchSettings ss;
    PlaySettings ps;
    SearchLimits limits;
    EvalInfo eval;
    std::vector<std::unique_ptr<NeuralNetAPI>> netSingle;
    std::vector<std::vector<std::unique_ptr<NeuralNetAPI>>> netBatches;
    std::unique_ptr<MCTSAgent> agent;
    std::unique_ptr<StateObj> state;
    std::string error;
};

static thread_local std::string g_error;

template <typename F>
static int guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

extern "C" {

const char* ref_last_error(void) { return g_error.c_str(); }

}  // extern "C"

// UCI option Search_Type (crazyara.cpp:736: useMCGS = Search_Type == "mcgs"; the option's default IS mcgs, optionsuci.cpp).  The product has no
// such switch: Node::add_new_node_to_tree's transposition link reads its candidate from the child slot it is about to fill (node.cpp:730-731),
// which SearchThread only calls for an EMPTY slot (searchthread.cpp:194-211), so the flag changes no tree.  ref_set_use_mcgs lets the tests
// run the compiled reference under both values and compare the dumps (tests/test_mcts_reference_build.py).  Applies to agents created afterwards.
static bool g_use_mcgs = false;
extern "C" void ref_set_use_mcgs(int on) { g_use_mcgs = on != 0; }

// SearchSettings / PlaySettings exactly as CrazyAra::init_search_settings fills them, from the product's settings struct
// search_threads = the UCI option `Threads`: that many SearchThreads on the one tree, each with its own batch net (crazyara.cpp:548-563)
template <typename MakeNet>
static ref_agent* create_agent(const mi_search_settings* s, MakeNet&& make_net, unsigned search_threads = 1) {
    ref_agent* a = nullptr;
    if (guard([&] {
            refshim::Config& c = refshim::config();
            c.mode = s->mode;
            c.version_major = s->version_major;
            c.version_minor = s->version_minor;
            c.layout = cra::layout_for(s->mode, s->version_major, s->version_minor);
            c.is_policy_map = s->is_policy_map != 0;
            c.clone_keeps_last_moves = s->clone_keeps_last_moves < 0 ? s->mode != cra::MODE_CRAZYHOUSE : s->clone_keeps_last_moves != 0;
            a = new ref_agent;
            SearchSettings& ss = a->ss;
            ss.multiPV = 1;
            ss.threads = search_threads;
            ss.batchSize = unsigned(s->batch_size);
            ss.useMCGS = g_use_mcgs;                            // default false; ref_set_use_mcgs(1) = the UCI default Search_Type mcgs (above)
            ss.searchPlayerMode = MODE_TWO_PLAYER;
            ss.qValueWeight = s->q_value_weight;
            ss.qVetoDelta = s->q_veto_delta;
            ss.epsilonChecksCounter = uint_fast8_t(s->epsilon_checks_counter);
            ss.epsilonGreedyCounter = uint_fast8_t(s->epsilon_greedy_counter);
            ss.cpuctInit = s->cpuct_init;
            ss.cpuctBase = s->cpuct_base;
            ss.dirichletEpsilon = s->dirichlet_epsilon;
            ss.dirichletAlpha = s->dirichlet_alpha;
            ss.nodePolicyTemperature = s->node_policy_temperature;
            ss.randomMoveFactor = 0.0f;
            ss.allowEarlyStopping = false;
            ss.useNPSTimemanager = false;
            ss.useTablebase = false;
            ss.reuseTree = true;
            ss.mctsSolver = s->mcts_solver != 0;
            ss.virtualStyle = VirtualStyle(s->virtual_style);   // same enum order (searchsettings.h:40-45)
            ss.virtualMixThreshold = s->virtual_mix_threshold;
            ss.virtualOffsetStrenght = s->virtual_offset_strength;
            ss.gamePhaseDefinition = MOVECOUNT;
            a->ps.initTemperature = 0.0;                        // best move = pv[0][0] (agent.cpp:38-55)
            a->ps.temperatureMoves = 0;
            a->ps.temperatureDecayFactor = 1.0;
            a->ps.quantileClipping = 0.0;
            a->netSingle.emplace_back(make_net(1u));            // crazyara.cpp:548-563: batch-1 net for the agent, batch-N per SearchThread
            for (unsigned t = 0; t < search_threads; ++t) {
                a->netBatches.emplace_back();
                a->netBatches[t].emplace_back(make_net(unsigned(s->batch_size)));
            }
            StateConstants::init(a->netSingle[0]->is_policy_map(), false);
            a->agent.reset(new MCTSAgent(a->netSingle, a->netBatches, &a->ss, &a->ps));
            g_rand_state = s->seed;
            refshim_seed_node_generator(s->seed);
        })) {
        delete a;
        return nullptr;
    }
    return a;
}

extern "C" {

ref_agent* ref_agent_create(const mi_search_settings* s, mi_eval_fn fn, void* user, int nb_policy) {
    const int channels = cra::layout_channels(cra::layout_for(s->mode, s->version_major, s->version_minor));
    const std::string name = "callback-v" + std::to_string(s->version_major) + "." + std::to_string(s->version_minor);
    return create_agent(s, [&](unsigned batch) { return new CallbackNet(batch, channels, nb_policy, name, fn, user); });
}

void ref_agent_destroy(ref_agent* a) { delete a; }

// `position fen ...` / ucinewgame: a new game (crazyara.cpp: clear_game_history + state->set)
int ref_agent_set_position(ref_agent* a, const char* fen, int is960, const char* variant) {
    return guard([&] {
        const cra::chess::Variant v = cra::chess::variant_from_name(variant);
        a->state.reset(new StateObj());
        std::string f = fen && *fen ? fen : cra::chess::start_fen(v);
        a->state->set(f, is960 != 0, int(v));
        refshim::config().is960 = is960 != 0;
        a->agent->clear_game_history();
    });
}

// `go`: limits as the UCI loop sets them (simulations / nodes only, no clock)
int ref_agent_go(ref_agent* a, unsigned simulations, unsigned nodes) {
    return guard([&] {
        a->limits.reset();
        a->limits.simulations = simulations;
        a->limits.nodes = nodes;
        a->agent->set_search_settings(a->state.get(), &a->limits, &a->eval);
        a->agent->set_must_wait(true);
        a->agent->perform_action();
    });
}

// a move is played on the board: the agent that searched hears about it as its own move (run_agent_thread, agent.cpp:107-112)
int ref_agent_apply_move(ref_agent* a, const char* uci) {
    return guard([&] {
        std::string u = uci;
        const Action m = a->state->uci_to_action(u);
        if (m == 0) throw std::invalid_argument("illegal move " + u);
        a->agent->apply_move_to_tree(m, true);
        a->state->do_action(m);
    });
}

int ref_agent_fen(ref_agent* a, char* out, int cap) {
    const std::string f = a->state->fen();
    if (int(f.size()) + 1 > cap) return 1;
    std::memcpy(out, f.c_str(), f.size() + 1);
    return 0;
}

// root statistics: the first noVisitIdx children in the node's (sorted) order
int ref_agent_root_children(ref_agent* a, int cap, uint32_t* moves, uint32_t* visits, float* q, float* priors) {
    Node* r = a->agent->get_root_node();
    if (!r || !r->is_playout_node()) return 0;
    const int n = int(r->get_no_visit_idx());
    const std::vector<Action> acts = r->get_legal_actions();
    for (int i = 0; i < n && i < cap; ++i) {
        moves[i] = uint32_t(acts[i]);
        visits[i] = r->get_child_number_visits(ChildIdx(i));
        q[i] = r->get_q_value(ChildIdx(i));
        priors[i] = r->get_policy_prob_small()[i];
    }
    return n;
}

int ref_agent_root_info(ref_agent* a, unsigned* root_visits, unsigned* node_count, float* root_value, int* node_type, int* end_in_ply,
                        int* checkmate_idx, unsigned* free_visits, int* n_legal) {
    Node* r = a->agent->get_root_node();
    if (!r || !r->is_playout_node()) return 1;
    *root_visits = r->get_visits();
    *node_count = r->get_node_count();                         // visits - freeVisits (node.cpp:1303-1306): the `nodes` of evalinfo.cpp
    *root_value = r->get_value();
    *node_type = int(r->get_node_type());
    *end_in_ply = int(r->get_end_in_ply());
    *checkmate_idx = r->get_checkmate_idx() == NO_CHECKMATE ? -1 : r->get_checkmate_idx();
    *free_visits = r->get_free_visits();
    *n_legal = int(r->get_number_child_nodes());
    return 0;
}

// EvalInfo of the last go (evalinfo.cpp:184-243): MCTS policy over ALL legal moves in root order, best move, bestMoveQ, nodes
int ref_agent_eval(ref_agent* a, int cap, double* policy, char* best_uci, int uci_cap, float* best_q, unsigned* nodes, unsigned* sel_depth) {
    const EvalInfo& e = a->eval;
    const int n = int(e.policyProbSmall.size());
    for (int i = 0; i < n && i < cap; ++i) policy[i] = e.policyProbSmall[i];
    const std::string u = StateConstants::action_to_uci(e.bestMove, a->state->is_chess960());
    if (int(u.size()) + 1 > uci_cap) return -1;
    std::memcpy(best_uci, u.c_str(), u.size() + 1);
    *best_q = e.bestMoveQ.empty() ? 0.0f : e.bestMoveQ[0];
    *nodes = unsigned(e.nodes);
    *sel_depth = unsigned(e.selDepth);
    return n;
}

// Multi_PV: the option only changes what update_eval_info writes (evalinfo.cpp:195-260)
void ref_agent_set_multipv(ref_agent* a, int k) { a->ss.multiPV = uint16_t(k < 1 ? 1 : k); }
// line idx of the last go's EvalInfo; returns the number of moves, 0 when the line does not exist
int ref_agent_pv_at(ref_agent* a, int idx, char* line, int cap, int* centipawns, int* moves_to_mate, float* q) {
    const EvalInfo& e = a->eval;
    if (idx < 0 || size_t(idx) >= e.pv.size() || e.pv[size_t(idx)].empty()) return 0;
    std::string out;
    int n = 0;
    for (Action m : e.pv[size_t(idx)]) {
        if (!out.empty()) out += ' ';
        out += StateConstants::action_to_uci(m, a->state->is_chess960());
        ++n;
    }
    if (int(out.size()) + 1 > cap) return -1;
    std::memcpy(line, out.c_str(), out.size() + 1);
    *centipawns = e.centipawns[size_t(idx)];
    *moves_to_mate = e.movesToMate[size_t(idx)];
    *q = e.bestMoveQ[size_t(idx)];
    return n;
}

// TimeManager::get_time_for_move on a SearchLimits filled from the product's mi_go_limits (randomMoveFactor 0)
int ref_time_for_move(const mi_go_limits* l, int side, int move_number) {
    SearchLimits lim;
    lim.movetime = l->movetime;
    lim.nodes = l->nodes;
    lim.simulations = l->simulations;
    lim.movestogo = l->movestogo;
    lim.depth = l->depth;
    lim.time[0] = l->time[0]; lim.time[1] = l->time[1];
    lim.inc[0] = l->inc[0]; lim.inc[1] = l->inc[1];
    lim.moveOverhead = l->move_overhead;
    lim.infinite = l->infinite != 0;
    TimeManager tm(0.0f);
    return tm.get_time_for_move(&lim, SideToMove(side), move_number);
}

// EvalInfo::pv[0] as a space-separated UCI line, centipawns[0], movesToMate[0] of the last go (update_eval_info, evalinfo.cpp:195-260)
int ref_agent_pv(ref_agent* a, char* line, int cap, int* centipawns, int* moves_to_mate) {
    const EvalInfo& e = a->eval;
    std::string out;
    int n = 0;
    if (!e.pv.empty()) {
        for (Action m : e.pv[0]) {
            if (!out.empty()) out += ' ';
            out += StateConstants::action_to_uci(m, a->state->is_chess960());
            ++n;
        }
    }
    if (int(out.size()) + 1 > cap) return -1;
    std::memcpy(line, out.c_str(), out.size() + 1);
    *centipawns = e.centipawns.empty() ? 0 : e.centipawns[0];
    *moves_to_mate = e.movesToMate.empty() ? 0 : e.movesToMate[0];
    return n;
}

// Whole-tree dump in depth-first preorder over the expanded children, one record per PLAYOUT node (a node that was selected at
// least once: it owns NodeData):  [n_expanded, visit_sum, real_visits, free_visits, node_type, end_in_ply, terminal,
// float bits of value, then per expanded child: move, visits, virtual-loss counter, float bits of Q, float bits of prior,
// child state (0 = no node yet, 1 = node without NodeData, 2 = playout node -> its record follows in order)].  Returns the number
// of uint32 words written, or -1 if cap is too small.
static long dump_node(Node* n, uint32_t* out, long cap, long w) {
    auto put = [&](uint32_t v) {
        if (w >= 0) {
            if (w < cap) out[w++] = v;
            else w = -1;
        }
    };
    auto fbits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    const int m = n->is_terminal() ? 0 : int(n->get_no_visit_idx());
    put(uint32_t(m));
    put(n->get_visits());
    put(n->get_real_visits());
    put(n->get_free_visits());
    put(uint32_t(n->get_node_type()));
    put(uint32_t(n->get_end_in_ply()));
    put(n->is_terminal() ? 1u : 0u);
    put(fbits(n->get_value()));
    const std::vector<Action> acts = n->get_legal_actions();
    std::vector<Node*> follow;
    for (int i = 0; i < m; ++i) {
        put(uint32_t(acts[i]));
        put(n->get_child_number_visits(ChildIdx(i)));
        put(uint32_t(n->get_virtual_loss_counter(ChildIdx(i))));
        put(fbits(n->get_q_value(ChildIdx(i))));
        put(fbits(n->get_policy_prob_small()[i]));
        Node* c = n->get_child_node(ChildIdx(i));
        const uint32_t st = c == nullptr ? 0u : c->is_playout_node() ? 2u : 1u;
        put(st);
        if (st == 2u) follow.push_back(c);
    }
    for (Node* c : follow) {
        if (w < 0) break;
        w = dump_node(c, out, cap, w);
    }
    return w;
}

long ref_agent_tree_dump(ref_agent* a, uint32_t* out, long cap) {
    Node* r = a->agent->get_root_node();
    if (!r || !r->is_playout_node()) return 0;
    return dump_node(r, out, cap, 0);
}

// counters of the two evaluator nets: predict() calls and positions evaluated (root net, batch net)
void ref_agent_net_counters(ref_agent* a, unsigned long long* out4) {
    const CallbackNet* s = dynamic_cast<const CallbackNet*>(a->netSingle[0].get());
    const CallbackNet* b = dynamic_cast<const CallbackNet*>(a->netBatches[0][0].get());
    if (!s || !b) { out4[0] = out4[1] = out4[2] = out4[3] = 0; return; }
    out4[0] = s->calls;
    out4[1] = s->evals;
    out4[2] = b->calls;
    out4[3] = b->evals;
}

// ---- single functions of node.cpp / blazeutil.h for the worked examples of the tests ----
float ref_get_current_cput(float visits, float cpuct_init, float cpuct_base) {
    SearchSettings s;
    s.cpuctInit = cpuct_init;
    s.cpuctBase = cpuct_base;
    return get_current_cput(visits, &s);
}

void ref_first_and_second_max(const float* v, int n, int end_idx, float* out_first, float* out_second, int* first_arg, int* second_arg) {
    blaze::DynamicVector<float> d{static_cast<size_t>(n)};
    for (int i = 0; i < n; ++i) d[i] = v[i];
    float f, s;
    size_t fa, sa;
    first_and_second_max(d, size_t(end_idx), f, s, fa, sa);
    *out_first = f;
    *out_second = s;
    *first_arg = int(fa);
    *second_arg = int(sa);
}

float ref_get_quantile(const double* v, int n, float quantile) {
    blaze::DynamicVector<double> d{static_cast<size_t>(n)};
    for (int i = 0; i < n; ++i) d[i] = v[i];
    return float(get_quantile(d, quantile));
}

void ref_apply_quantile_clipping(double* v, int n, float quantile) {
    blaze::DynamicVector<double> d{static_cast<size_t>(n)};
    for (int i = 0; i < n; ++i) d[i] = v[i];
    apply_quantile_clipping(quantile, d);
    for (int i = 0; i < n; ++i) v[i] = d[i];
}

void ref_sharpen_distribution(double* v, int n, float thresh) {       // blazeutil.h:94-105 (rl/selfplay.cpp:229-231)
    blaze::DynamicVector<double> d{static_cast<size_t>(n)};
    for (int i = 0; i < n; ++i) d[i] = v[i];
    sharpen_distribution(d, thresh);
    for (int i = 0; i < n; ++i) v[i] = d[i];
}

int ref_value_to_centipawn(float v) { return value_to_centipawn(v); }

#ifdef REF_WITH_HIPAPI
// ---- integration/hipapi.h exercised through the reference's own base class and users -----------------------------------------
// the reference's MCTSAgent / SearchThread on HipAPI nets: `go` computes on the GPU through mi_net_predict
ref_agent* ref_agent_create_hip(const mi_search_settings* s, const char* model_dir, int device_id, const char* precision) {
    return create_agent(s, [&](unsigned batch) { return new HipAPI(device_id, batch, model_dir, precision); });
}
// the same with `Threads` SearchThreads (measurement leg of bench.py: the throughput a CrazyAra build with this back end gets)
ref_agent* ref_agent_create_hip_threads(const mi_search_settings* s, const char* model_dir, int device_id, const char* precision, int threads) {
    if (threads < 1 || threads > 64) return nullptr;
    return create_agent(s, [&](unsigned batch) { return new HipAPI(device_id, batch, model_dir, precision); }, unsigned(threads));
}

struct ref_hipapi {
    std::vector<std::unique_ptr<NeuralNetAPI>> nets;
};

// `mode` fixes the label set the reference binary would have been built with (MODE_CRAZYHOUSE / MODE_CHESS / MODE_LICHESS)
ref_hipapi* ref_hipapi_create(const char* model_dir, int device_id, unsigned batch, const char* precision, int mode) {
    ref_hipapi* h = nullptr;
    if (guard([&] {
            refshim::config().mode = mode;
            h = new ref_hipapi;
            h->nets.emplace_back(new HipAPI(device_id, batch, model_dir, precision));
            const Version v = h->nets[0]->get_version();
            refshim::config().version_major = int(version::get_major(v));
            refshim::config().version_minor = int(version::get_minor(v));
            refshim::config().layout = cra::layout_for(mode, int(version::get_major(v)), int(version::get_minor(v)));
        })) {
        delete h;
        return nullptr;
    }
    return h;
}

void ref_hipapi_destroy(ref_hipapi* h) {
    if (h) {
        for (auto& n : h->nets) delete static_cast<HipAPI*>(n.release());      // the base class has no virtual destructor
        delete h;
    }
}

// out[0..8): version, is_policy_map, nb_input_values_total, nb_policy_values, batch_size, nb_auxiliary_outputs, has_auxiliary_outputs,
// game_phase -- all through the BASE CLASS getters (neuralnetapi.h:230-299)
void ref_hipapi_info(ref_hipapi* h, long* out) {
    const NeuralNetAPI* n = h->nets[0].get();
    out[0] = long(n->get_version());
    out[1] = n->is_policy_map() ? 1 : 0;
    out[2] = long(n->get_nb_input_values_total());
    out[3] = long(n->get_nb_policy_values());
    out[4] = long(n->get_batch_size());
    out[5] = long(n->get_nb_auxiliary_outputs());
    out[6] = n->has_auxiliary_outputs() ? 1 : 0;
    out[7] = long(n->get_game_phase());
}

int ref_hipapi_model_name(ref_hipapi* h, char* out, int cap) {
    const std::string s = h->nets[0]->get_model_name();
    if (int(s.size()) + 1 > cap) return 1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}

// NeuralNetAPI::validate_neural_network (neuralnetapi.cpp:137-162) logs its findings and returns void; the same conditions are
// re-evaluated here with the reference's check_condition so that the test sees the verdict: number of failed conditions
int ref_hipapi_validate(ref_hipapi* h) {
    NeuralNetAPI* n = h->nets[0].get();
    n->validate_neural_network();
    int failed = 0;
    failed += !check_condition(unsigned(n->get_nb_input_values_total()), StateConstants::NB_VALUES_TOTAL(), "nbNNInputValues", "NB_VALUES_TOTAL()");
    failed += !check_condition(unsigned(n->get_nb_policy_values()),
                               n->is_policy_map() ? StateConstants::NB_LABELS_POLICY_MAP() : StateConstants::NB_LABELS(), "nbPolicyValues", "labels");
    return failed;
}

// the reference's `inference` loop (crazyara.cpp:156-181): NeuralNetAPIUser owns the host buffers and calls predict back to back
struct ExposedUser : public NeuralNetAPIUser {
    using NeuralNetAPIUser::NeuralNetAPIUser;
    float* in() { return inputPlanes; }
    float* value() { return valueOutputs; }
    float* probs() { return probOutputs; }
    float* aux() { return auxiliaryOutputs; }
};

int ref_hipapi_run_inference(ref_hipapi* h, int iterations, const float* planes, float* value, float* probs, float* aux, double* seconds) {
    return guard([&] {
        ExposedUser user(h->nets);
        const NeuralNetAPI* n = h->nets[0].get();
        const size_t B = n->get_batch_size();
        std::memcpy(user.in(), planes, B * n->get_nb_input_values_total() * sizeof(float));
        const auto t0 = std::chrono::steady_clock::now();
        user.run_inference(uint_fast16_t(iterations));                      // neuralnetapiuser.cpp:104-109
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::memcpy(value, user.value(), B * sizeof(float));
        std::memcpy(probs, user.probs(), B * n->get_nb_policy_values() * sizeof(float));
        if (aux && user.aux() && n->has_auxiliary_outputs()) std::memcpy(aux, user.aux(), B * n->get_nb_auxiliary_outputs() * sizeof(float));
    });
}
#endif  // REF_WITH_HIPAPI

}  // extern "C"
