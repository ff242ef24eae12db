#include "mcts.h"

#include <sys/mman.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <stdexcept>

#include "../chess/planes_host.h"

namespace cra {
namespace search {

using chess::Move;
using chess::Position;

constexpr float Q_INIT = -1.0f;                 // constants.h:85
constexpr float LOSS_VALUE = -1.0f, DRAW_VALUE = 0.0f, WIN_VALUE = 1.0f;   // constants.h:78-80
constexpr int TERMINAL_NODE_CACHE_FACTOR = 2;   // SearchThread: terminalNodeCache = 2 * batchSize

constexpr int32_t CHILD_NONE = -1;              // no node behind this move yet
constexpr int32_t CHILD_PENDING = -2;           // a collector is creating the node right now (shared trees): a collision for everybody else

// ---- shared trees --------------------------------------------------------------------------------------------------------------
// Fields of a node that OTHER nodes' owners read without that node's lock (the reference reads them the same way, unsynchronised:
// node.cpp:108-172 child->d->nodeType, searchthread.cpp:151 nextNode->get_visits(), :239 has_nn_results()): accessed through
// relaxed / acquire-release atomics so that the shared tree is free of data races (ThreadSanitizer: scripts/hostbench/tsan_pool.cpp).
namespace {
inline int8_t ld_type(const Node& n) { return __atomic_load_n(&n.node_type, __ATOMIC_RELAXED); }
inline uint16_t ld_end(const Node& n) { return __atomic_load_n(&n.end_in_ply, __ATOMIC_RELAXED); }
inline void st_end(Node& n, uint16_t v) { __atomic_store_n(&n.end_in_ply, v, __ATOMIC_RELAXED); }
inline bool ld_has_data(const Node& n) { return __atomic_load_n(&n.has_data, __ATOMIC_ACQUIRE); }
inline bool ld_has_nn(const Node& n) { return __atomic_load_n(&n.has_nn, __ATOMIC_ACQUIRE); }
inline uint32_t ld_visits(const Node& n) { return __atomic_load_n(&n.visit_sum, __ATOMIC_RELAXED); }

struct NodeLock {                                // Node::lock() / unlock() (node.cpp:936-944); a no-op for single-collector trees
    Node* n;
    NodeLock(Node& node, bool on) : n(on ? &node : nullptr) {
        if (n)
            while (__atomic_test_and_set(&n->lock, __ATOMIC_ACQUIRE))
                while (__atomic_load_n(&n->lock, __ATOMIC_RELAXED)) __builtin_ia32_pause();
    }
    ~NodeLock() { if (n) __atomic_clear(&n->lock, __ATOMIC_RELEASE); }
    NodeLock(const NodeLock&) = delete;
    NodeLock& operator=(const NodeLock&) = delete;
};
}  // namespace

// The chunk-pointer table (1 MiB of address space) is an anonymous mapping of its own: zero pages the kernel hands over one by one when
// they are first touched, whatever the allocator's thresholds are at the moment (a calloc of this size can come from the heap once
// glibc has raised its mmap threshold, and is then memset in full on every played move); the atomics are constructed in place.
static_assert(sizeof(std::atomic<Node*>) == sizeof(Node*) && std::atomic<Node*>::is_always_lock_free, "the table is raw zeroed memory");
NodeArena::NodeArena() {
    void* p = mmap(nullptr, size_t(kMaxChunks) * sizeof(std::atomic<Node*>), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) throw std::bad_alloc();
    table_ = static_cast<std::atomic<Node*>*>(p);          // all-zero pages: every entry is a null pointer; entries are created where used
}
NodeArena::~NodeArena() {
    clear();
    munmap(static_cast<void*>(table_), size_t(kMaxChunks) * sizeof(std::atomic<Node*>));
}
void NodeArena::clear() {
    const size_t chunks = (size_t(size_.load()) + kChunk - 1) >> kChunkBits;
    for (size_t c = 0; c < size_t(kMaxChunks) && (c < chunks || table_[c].load(std::memory_order_relaxed)); ++c) {
        delete[] table_[c].load(std::memory_order_relaxed);
        table_[c].store(nullptr, std::memory_order_relaxed);
    }
    size_.store(0);
}
void NodeArena::swap(NodeArena& o) {
    std::swap(table_, o.table_);
    const uint32_t a = size_.load(), b = o.size_.load();
    size_.store(b);
    o.size_.store(a);
}
int NodeArena::emplace_back() {
    const uint32_t i = size_.fetch_add(1, std::memory_order_acq_rel);
    const size_t c = size_t(i) >> kChunkBits;
    if (c >= size_t(kMaxChunks)) throw std::length_error("search tree: node arena exhausted");
    if (table_[c].load(std::memory_order_acquire) == nullptr) {
        std::lock_guard<std::mutex> lk(grow_);
        if (table_[c].load(std::memory_order_acquire) == nullptr) {
            // (no placement-new of the entry here: its write of a null pointer would be a NON-atomic write racing with the unlocked
            // first check of other collectors -- ThreadSanitizer's report of round 5.  std::atomic<Node*> is an implicit-lifetime type
            // whose all-zero bytes from mmap are the null pointer; every access to an entry is an atomic operation)
            table_[c].store(new Node[kChunk], std::memory_order_release);
        }
    }
    return int(i);
}

float get_current_cput(float visits, const SearchSettings& s) {
    return std::log((visits + s.cpuct_base + 1) / s.cpuct_base) + s.cpuct_init;
}

VirtualStyle get_virtual_style(const SearchSettings& s, uint32_t visits) {
    if (s.virtual_style == VIRTUAL_MIX) return visits > s.virtual_mix_threshold ? VIRTUAL_LOSS : VIRTUAL_VISIT;
    return VirtualStyle(s.virtual_style);
}

Tree::Tree(const Position& root, const SearchSettings& settings) : s_(settings), root_pos_(root) {
    if (s_.epsilon_greedy_counter < 0 || s_.epsilon_checks_counter < 0) throw std::invalid_argument("epsilon counters must be >= 0");
    collectors_.emplace_back(new Collector);
    collectors_[0]->rng = s_.seed;
    noise_rng_.seed(s_.seed);
    tables_ = &chess::policy_tables(s_.mode);
    layout_ = layout_for(s_.mode, s_.version_major, s_.version_minor);
    keep_last_moves_ = s_.clone_keeps_last_moves < 0 ? s_.mode != MODE_CRAZYHOUSE : s_.clone_keeps_last_moves != 0;
    new_node(root_pos_);
}

// Node::Node + check_for_terminal (node.cpp:82-106, 880-904)
int Tree::new_node(const Position& pos) {
    const int index = nodes_.emplace_back();
    Node& n = nodes_[index];
    pos.legal_moves(n.actions);
    n.plies = uint16_t(pos.game_ply());
    n.key = pos.key();
    n.stm = uint8_t(pos.side_to_move());
    const chess::TerminalType tt = pos.is_terminal(n.actions.size());
    if (tt != chess::TERMINAL_NONE) {
        n.terminal = true;           // mark_as_terminal: NodeData with noVisitIdx = 0, sorted
        n.has_data = true;
        n.sorted = true;
        n.no_visit_idx = 0;
        switch (tt) {
            case chess::TERMINAL_WIN: n.set_value(WIN_VALUE); n.node_type = NT_WIN; break;
            case chess::TERMINAL_DRAW: n.set_value(DRAW_VALUE); n.node_type = NT_DRAW; n.actions.clear(); break;
            case chess::TERMINAL_LOSS: n.set_value(LOSS_VALUE); n.node_type = NT_LOSS; break;
            default: break;
        }
    }
    n.priors.assign(n.actions.size(), 0.0f);
    if (!n.terminal) {
        n.policy_idx.resize(n.actions.size());
        for (size_t i = 0; i < n.actions.size(); ++i) {
            const int idx = chess::policy_index(*tables_, pos, n.actions[i], s_.is_policy_map);
            if (idx < 0) throw std::logic_error("legal move without a policy label: " + pos.move_to_uci(n.actions[i]));
            n.policy_idx[i] = uint16_t(idx);
        }
    }
    return index;
}

void Tree::set_collectors(int k) {
    if (pending_new() > 0 || pending_collisions() > 0) throw std::logic_error("set_collectors with a batch in flight");
    k = std::max(1, k);
    while (int(collectors_.size()) > k) collectors_.pop_back();
    while (int(collectors_.size()) < k) {
        collectors_.emplace_back(new Collector);
        collectors_.back()->rng = s_.seed + 7919u * uint32_t(collectors_.size() - 1);      // every collector its own exploration stream
    }
    concurrent_ = k > 1;
}

int Tree::pending_new(int ctx) const {
    if (ctx >= 0) return int(collectors_.at(size_t(ctx))->new_nodes.size());
    int n = 0;
    for (const auto& c : collectors_) n += int(c->new_nodes.size());
    return n;
}
int Tree::pending_collisions(int ctx) const {
    if (ctx >= 0) return int(collectors_.at(size_t(ctx))->collision_trajectories.size());
    int n = 0;
    for (const auto& c : collectors_) n += int(c->collision_trajectories.size());
    return n;
}
uint64_t Tree::depth_sum() const {
    uint64_t v = 0;
    for (const auto& c : collectors_) v += c->depth_sum;
    return v;
}
uint32_t Tree::depth_max() const {
    uint32_t v = 0;
    for (const auto& c : collectors_) v = std::max(v, c->depth_max);
    return v;
}
void Tree::reset_depth_max() {
    for (auto& c : collectors_) c->depth_max = 0;
}

void Tree::root_desc(BoardDesc& d) const { chess::pack_desc(root_pos_, d, layout_needs_move_features(layout_)); }

// fill_nn_results (searchthread.cpp:290-299): gather priors, temperature, value
void Tree::fill_nn_result(Node& n, float value, const float* probs) {
    for (size_t i = 0; i < n.actions.size(); ++i) n.priors[i] = probs[n.policy_idx[i]];   // set_probabilities_for_moves, node.cpp:961-979
    finish_node(n, value);
}

void Tree::fill_nn_result_gathered(Node& n, float value, const float* priors) {
    for (size_t i = 0; i < n.actions.size(); ++i) n.priors[i] = priors[i];               // the same values, gathered before the copy back
    finish_node(n, value);
}

void Tree::finish_node(Node& n, float value) {
    std::vector<uint16_t>().swap(n.policy_idx);
    const float t = s_.node_policy_temperature;                                           // apply_temperature, blazeutil.h:77-87
    if (t != 1) {
        const float e = 1.0f / t;
        float sum = 0.0f;
        for (float& p : n.priors) { p = std::pow(p, e); sum += p; }
        for (float& p : n.priors) p /= sum;
    }
    n.set_value(value);                                                                   // node_assign_value, searchthread.cpp:475-489
    __atomic_store_n(&n.has_nn, true, __ATOMIC_RELEASE);      // last: from here on other collectors walk through this node
}

void Tree::set_root_result(float value, const float* probs) {
    fill_nn_result(nodes_[0], value, probs);
    prepare_node_for_visits(nodes_[0], *collectors_[0]);                                                   // mctsagent.cpp:195
}

// apply_dirichlet_noise_to_prior_policy (node.cpp:950-954) with get_dirichlet_noise (blazeutil.h:113-124: one fresh
// std::gamma_distribution<float>(alpha, 1) per entry, normalised by the float sum), then fully_expand_node (node.cpp:582-593):
// every child gets its NodeData slot and the current order is kept (the noised priors are no longer sorted).
void Tree::begin_search() {
    Node& n = nodes_[0];
    if (!(s_.dirichlet_epsilon > 0.009f) || n.terminal || !n.has_nn || n.actions.empty()) return;
    std::vector<float> noise(n.actions.size());
    float sum = 0.0f;
    for (float& v : noise) {
        std::gamma_distribution<float> distribution(s_.dirichlet_alpha, 1.0f);
        v = distribution(noise_rng_);
        sum += v;
    }
    const float keep = 1 - s_.dirichlet_epsilon;
    for (size_t i = 0; i < noise.size(); ++i) {
        const float a = keep * n.priors[i], b = s_.dirichlet_epsilon * (noise[i] / sum);
        n.priors[i] = a + b;
    }
    if (!n.sorted) prepare_node_for_visits(n, *collectors_[0]);
    while (size_t(n.no_visit_idx) < n.actions.size()) increment_no_visit_idx(n);
}

// Tree reuse.  The reference keeps shared_ptr subtrees (pick_next_node) and lets the rest of the old tree die; here the kept
// subtree is copied breadth-first into a fresh node array (indices are the links), everything else is dropped.
bool Tree::apply_move(Move m) {
    if (pending_new() > 0 || pending_collisions() > 0) throw std::logic_error("apply_move with a batch in flight");
    const Node& r = nodes_[0];
    int child = -1;
    if (r.has_data) {
        for (int i = 0; i < int(r.no_visit_idx); ++i)
            if (r.actions[i] == m) { child = r.child[i]; break; }
    }
    bool legal = false;
    for (Move a : r.actions) legal |= a == m;
    if (!legal && !r.terminal) {
        std::vector<Move> lm;
        root_pos_.legal_moves(lm);
        for (Move a : lm) legal |= a == m;
    }
    if (!legal) throw std::invalid_argument("apply_move: illegal move " + root_pos_.move_to_uci(m));
    root_pos_.do_move(m);
    // get_root_node_from_tree: the candidate must be a playout node (has NodeData) with at least one visit below it
    const bool keep = child >= 0 && nodes_[child].has_data && nodes_[child].has_nn && !nodes_[child].terminal && nodes_[child].visit_sum > 0;
    if (keep) {
        // Stored states of the kept subtree: their key history is the game's (absolute), but in a build whose root clone starts the
        // move history afresh (crazyhouse: Board::operator= drops lastMoves, board.cpp:106-108) their last-move lists reach back beyond
        // the NEW root, where a replay from it would not -- for a layout with last-move planes they are dropped (the nodes then replay
        // from the root like nodes beyond the budget); every other layout never reads the list.
        const bool drop_states = !keep_last_moves_ && layout_needs_move_features(layout_);
        uint32_t kept_states = 0;
        NodeArena fresh;
        std::vector<int> order{child};               // breadth-first copy; remap[i] = new index of old node order[i]
        for (size_t head = 0; head < order.size(); ++head) {
            Node& n = fresh[fresh.emplace_back()];
            n = std::move(nodes_[order[head]]);
            if (drop_states || head == 0) n.state.reset();       // (the root's position is root_pos_)
            kept_states += n.state != nullptr;
            for (int32_t& c : n.child)
                if (c >= 0) { order.push_back(c); c = int32_t(order.size()) - 1; }
        }
        nodes_.swap(fresh);
        stored_states_.store(kept_states, std::memory_order_relaxed);
    } else {
        nodes_.clear();
        stored_states_.store(0, std::memory_order_relaxed);
        new_node(root_pos_);
    }
    for (auto& c : collectors_) { c->depth_sum = 0; c->depth_max = 0; }
    return keep;
}

// sort_moves_by_probabilities + init_node_data (node.cpp:464-470, 634-643, nodedata.cpp:40-57).
// The reference uses an unstable std::sort with greater<float>; ties are broken by the original index here.
void Tree::prepare_node_for_visits(Node& n, Collector& col) {
    // stable insertion sort of the indices by descending prior (a few dozen moves; same order as std::stable_sort)
    std::vector<int>& perm = col.sort_perm;
    std::vector<Move>& sort_moves_ = col.sort_moves;
    std::vector<float>& sort_priors_ = col.sort_priors;
    perm.resize(n.actions.size());
    for (int i = 0; i < int(perm.size()); ++i) {
        const float p = n.priors[i];
        int j = i;
        while (j > 0 && n.priors[perm[j - 1]] < p) { perm[j] = perm[j - 1]; --j; }
        perm[j] = i;
    }
    sort_moves_.assign(n.actions.begin(), n.actions.end());
    sort_priors_.assign(n.priors.begin(), n.priors.end());
    for (size_t i = 0; i < perm.size(); ++i) { n.actions[i] = sort_moves_[perm[i]]; n.priors[i] = sort_priors_[perm[i]]; }
    n.sorted = true;
    if (!n.has_data) {
        n.no_visit_idx = 1;
        {   // room for the first few children at once: most nodes expand 2-6 of them, one allocation each instead of 3 regrowths
            const size_t room = std::min<size_t>(n.actions.size(), 6);
            n.child_visits.reserve(room); n.q.reserve(room); n.child.reserve(room); n.vl.reserve(room); n.child_types.reserve(room);
        }
        n.child_visits.assign(1, 0u);
        n.q.assign(1, Q_INIT);
        n.child.assign(1, CHILD_NONE);
        n.vl.assign(1, 0);
        n.child_types.assign(1, NT_UNSOLVED);
        n.unsolved_children = uint16_t(n.actions.size());                                 // NodeData(numberChildNodes), nodedata.cpp:70-75
        __atomic_store_n(&n.has_data, true, __ATOMIC_RELEASE);                            // last: parents test it without this node's lock
    }
}

void Tree::increment_no_visit_idx(Node& n) {                                              // node.cpp:571-580
    if (n.no_visit_idx < n.actions.size()) {
        ++n.no_visit_idx;
        n.child_visits.push_back(0u);
        n.q.push_back(Q_INIT);
        n.child.push_back(CHILD_NONE);
        n.vl.push_back(0);
        n.child_types.push_back(NT_UNSOLVED);
    }
}

// Node::select_child_node + get_current_u_values (node.cpp:1056-1063, 1150-1167):
//   argmax_i<noVisitIdx ( Q_i + float( double(cpuct * P_i) * (sqrt(double(N)) / (n_i + 1.0)) ) ), first maximum wins.
int Tree::select_child(Node& n, Collector& col) {
    if (!n.sorted) prepare_node_for_visits(n, col);
    if (n.no_visit_idx == 1) return 0;
    if (n.checkmate_idx >= 0) return n.checkmate_idx;                                     // has_forced_win
    const float cpuct = get_current_cput(float(n.visit_sum), s_);
    const double sq = std::sqrt(double(n.visit_sum));
    const int m = int(n.no_visit_idx);
    col.select_buf.resize(size_t(m));
    float* __restrict__ val = col.select_buf.data();
    const float* __restrict__ pr = n.priors.data();
    const float* __restrict__ qv = n.q.data();
    const uint32_t* __restrict__ cv = n.child_visits.data();
    for (int i = 0; i < m; ++i)                      // element-wise, vectorisable; each element rounds exactly as the scalar form
        val[i] = qv[i] + float(double(cpuct * pr[i]) * (sq / (double(cv[i]) + 1.0)));
    int best = 0;
    float best_v = -std::numeric_limits<float>::infinity();
    for (int i = 0; i < m; ++i)
        if (val[i] > best_v) { best_v = val[i]; best = i; }
    return best;
}

void Tree::apply_virtual_loss(Node& n, int c) {                                           // node.cpp:507-529
    switch (get_virtual_style(s_, n.child_visits[c])) {
        case VIRTUAL_LOSS: n.q[c] = float((double(n.q[c]) * n.child_visits[c] - 1) / double(n.child_visits[c] + 1)); break;
        case VIRTUAL_OFFSET: n.q[c] = float(n.q[c] - s_.virtual_offset_strength); break;
        default: break;
    }
    ++n.child_visits[c];
    __atomic_store_n(&n.visit_sum, n.visit_sum + 1, __ATOMIC_RELAXED);       // written under the node's lock, read by limit checks without
    ++n.vl[c];
}

void Tree::revert_virtual_loss(Node& n, int c) {                                          // node.cpp:661-679
    switch (get_virtual_style(s_, n.child_visits[c])) {
        case VIRTUAL_LOSS: n.q[c] = float((double(n.q[c]) * n.child_visits[c] + 1) / (n.child_visits[c] - 1)); break;
        case VIRTUAL_OFFSET: n.q[c] = float(n.q[c] + s_.virtual_offset_strength); break;
        default: break;
    }
    --n.child_visits[c];
    __atomic_store_n(&n.visit_sum, n.visit_sum - 1, __ATOMIC_RELAXED);
    --n.vl[c];
}

void Tree::revert_virtual_loss_and_update(Node& n, int c, float value, bool free_backup, bool solve) {   // node.h:199-246
    n.value_sum += value;
    ++n.real_visits;
    if (n.child_visits[c] == 1) {
        n.q[c] = value;
    } else {
        switch (get_virtual_style(s_, n.child_visits[c])) {
            case VIRTUAL_LOSS: n.q[c] = float((double(n.q[c]) * n.child_visits[c] + 1 + value) / n.child_visits[c]); break;
            case VIRTUAL_VISIT: {
                const uint32_t r = n.real_child_visits(c);
                n.q[c] = float((double(n.q[c]) * r + value) / (r + 1));
                break;
            }
            case VIRTUAL_OFFSET: {
                // (the reference reads childRealVisit uninitialised in this branch, node.h:228-231; real visits are meant)
                const uint32_t r = n.real_child_visits(c);
                double nq = double(n.q[c]) + n.vl[c] * s_.virtual_offset_strength;
                nq = (nq * r + value) / (r + 1.0);
                n.q[c] = float(nq - ((n.vl[c] - 1) * s_.virtual_offset_strength));
                break;
            }
            default: break;
        }
    }
    --n.vl[c];
    if (free_backup) __atomic_store_n(&n.free_visits, n.free_visits + 1, __ATOMIC_RELAXED);
    if (solve) solve_for_terminal(n, c);
}

// Node::solve_for_terminal (node.cpp:365-453) without tablebases, MODE_TWO_PLAYER: a child's WIN is this node's LOSS.
//   WIN  <- one child is a LOSS (solved_win, node.cpp:108-117); remembers the mating child in checkmate_idx
//   LOSS <- every child is a WIN (solved_loss, node.cpp:155-172)
//   DRAW <- every child is a WIN or a DRAW, at least one DRAW (solved_draw / at_least_one_drawn_child, node.cpp:119-148)
// end_in_ply: define_end_ply_for_solved_terminal (node.cpp:268-289); update_solved_terminal (:291-297) overwrites the node
// value (set_value counts one more real visit) and the edge's Q.
bool Tree::solve_for_terminal(Node& n, int c) {
    const int ci = n.child[c];
    if (ci < 0 || !ld_has_data(nodes_[ci])) return false;                                 // !childNode->is_playout_node()
    const Node& ch = nodes_[ci];
    // one look at the child's verdict (another collector may be proving it right now): type first, then the distance that was
    // stored before it
    const int8_t ch_type = __atomic_load_n(&ch.node_type, __ATOMIC_ACQUIRE);
    const uint16_t ch_end = ld_end(ch);
    if (ch_type == NT_UNSOLVED) return false;
    if (n.node_type != NT_UNSOLVED) return false;                                         // already solved
    if (n.child_types[c] == NT_UNSOLVED) {
        --n.unsolved_children;
        n.child_types[c] = ch_type;
        if (ch_type == NT_WIN) {                                                          // disable_action, node.cpp:1006-1010
            n.priors[c] = 0.0f;
            n.q[c] = float(-2147483647);
        }
    }
    auto all_children = [&](auto&& pred) {
        for (int i = 0; i < int(n.child.size()); ++i) {
            if (n.child[i] < 0 || !pred(nodes_[n.child[i]])) return false;
        }
        return true;
    };
    auto solved = [&](int8_t type, uint16_t end_in_ply) {                                 // distance first, verdict last (release)
        st_end(n, end_in_ply);
        __atomic_store_n(&n.node_type, type, __ATOMIC_RELEASE);
    };
    if (ch_type == NT_LOSS) {
        solved(NT_WIN, uint16_t(ch_end + 1));
        n.set_value(WIN_VALUE);
        n.q[c] = WIN_VALUE;
        n.checkmate_idx = c;
        return true;
    }
    if (n.unsolved_children == 0 && ch_type == NT_WIN && all_children([](const Node& k) { return ld_type(k) == NT_WIN; })) {
        uint16_t longest = n.end_in_ply;
        for (int i : n.child) longest = std::max<uint16_t>(longest, uint16_t(ld_end(nodes_[i]) + 1));   // longest line
        solved(NT_LOSS, longest);
        n.set_value(LOSS_VALUE);
        n.q[c] = LOSS_VALUE;
        return true;
    }
    if (n.unsolved_children == 0 && ch_type != NT_LOSS) {
        bool drawn = false;
        const bool ok = all_children([&](const Node& k) {
            const int8_t kt = ld_type(k);
            if (!ld_has_data(k) || (kt != NT_DRAW && kt != NT_WIN)) return false;
            drawn |= kt == NT_DRAW;
            return true;
        });
        if (ok && drawn) {
            // shortest drawn line: `child.end + 1 < end` with end starting at 0 never fires (node.cpp:277-285): end_in_ply stays 0
            solved(NT_DRAW, n.end_in_ply);
            n.set_value(DRAW_VALUE);
            n.q[c] = DRAW_VALUE;
            return true;
        }
    }
    return false;
}

// backup_value<freeBackup> (node.h:819-843) for trees (no transposition nodes: targetQValue stays 0)
void Tree::backup_value(float value, const Trajectory& t, bool free_backup, bool solve) {
    for (auto it = t.rbegin(); it != t.rend(); ++it) {
        value = -value;                                                                   // MODE_TWO_PLAYER
        Node& n = nodes_[it->node];
        NodeLock lk(n, concurrent_);                                                      // revert_virtual_loss_and_update locks, node.h:201,244
        revert_virtual_loss_and_update(n, it->child_idx, value, free_backup, solve);
    }
}

// ---- epsilon exploration helpers ----------------------------------------------------------------------------------
// rand(): the classic ANSI-C generator, one stream per tree
uint32_t Tree::next_rand(Collector& col) {
    col.rng = col.rng * 1103515245u + 12345u;
    return (col.rng >> 16) & 0x7fffu;
}

// get_random_depth (searchthread.cpp:497-501): ceil(-log2(1 - r/100) - 1), r uniform in 1..100.  r = 100 makes that +infinity, and
// the reference converts it to size_t -- undefined behaviour; the x86-64 code GCC emits for the conversion yields 0 (measured on the
// reference's own function, oracle/_ref), i.e. the playout starts at the root, and that is what is restated here
size_t Tree::get_random_depth(Collector& col) {
    const int r = int(next_rand(col) % 100u) + 1;
    if (r == 100) return 0;
    return size_t(std::ceil(-std::log2(1 - r / 100.0) - 1));
}

// get_best_action_index(fast = true) (node.cpp:1123-1148): the mating child, else (proven loss) the child that delays the mate
// longest, else the most-visited child (first maximum)
int Tree::best_action_index_fast(const Node& n) const {
    if (n.checkmate_idx >= 0) return n.checkmate_idx;
    int best = 0;
    if (n.node_type == NT_LOSS) {
        uint16_t longest = 0;
        for (int i = 0; i < int(n.child.size()); ++i)
            if (n.child[i] >= 0 && ld_end(nodes_[n.child[i]]) > longest) { longest = ld_end(nodes_[n.child[i]]); best = i; }
        return best;
    }
    for (int i = 1; i < int(n.no_visit_idx); ++i)
        if (n.child_visits[i] > n.child_visits[best]) best = i;
    return best;
}

// CRA_REPLAY_PROFILE (scripts/hostbench/replay_share_bench.cpp only; never in the library): ticks spent on what MCTS_STORE_STATES would
// remove -- the clone of the root position and the do_move replay down the selected path -- by depth, against the whole of collect().
#ifdef CRA_REPLAY_PROFILE
#include <x86intrin.h>
unsigned long long g_replay_ticks[128], g_replay_steps[128], g_clone_ticks, g_expand_ticks, g_collect_ticks, g_leaf_depth_hist[128];
#define CRA_TICK(var) const unsigned long long var = __rdtsc()
#define CRA_TOCK(acc, var) (acc) += __rdtsc() - (var)
#else
#define CRA_TICK(var) do { } while (0)
#define CRA_TOCK(acc, var) do { } while (0)
#endif

// The position of a simulation on its way down, lazily: `base` = the stored state of the deepest node passed that has one (else the
// root position) and the moves since; get() materialises it in the collector's scratch position -- one copy and only the moves behind the
// last stored state, instead of a clone of the root and a do_move per ply (SearchThread::get_new_child_to_evaluate with / without
// MCTS_STORE_STATES, searchthread.cpp:198-213).  After get() the scratch position is kept current move by move.
struct Tree::LazyPos {
    Tree& t;
    Collector& col;
    const Position* base;
    bool from_root = true, live = false;
    LazyPos(Tree& tree, Collector& c) : t(tree), col(c), base(&tree.root_pos_) { col.replay.clear(); }
    void step(int next, Move mv) {                            // the descent moves on to node `next` by `mv`
        const Node& n = t.nodes_[next];
        if (live) { col.scratch_pos.do_move(mv, &n.key); return; }
        if (n.state) { base = n.state.get(); from_root = false; col.replay.clear(); return; }
        col.replay.emplace_back(mv, &n.key);
    }
    Position& get() {
        if (!live) {
            CRA_TICK(t_clone);
            col.scratch_pos = *base;                          // rootState->clone() / currentNode->get_state()->clone()
            if (from_root && !t.keep_last_moves_) col.scratch_pos.clear_last_moves();
            CRA_TOCK(g_clone_ticks, t_clone);
            CRA_TICK(t_replay);
            for (const auto& mk : col.replay) col.scratch_pos.do_move(mk.first, mk.second);     // actionsBuffer replay
            CRA_TOCK(g_replay_ticks[col.replay.size() < 127 ? col.replay.size() : 127], t_replay);
#ifdef CRA_REPLAY_PROFILE
            g_replay_steps[col.replay.size() < 127 ? col.replay.size() : 127] += col.replay.size();
#endif
            col.replay.clear();
            live = true;
        }
        return col.scratch_pos;
    }
};

// get_starting_node (searchthread.cpp:144-162): walk the most-visited line for a random number of plies.  No virtual loss and
// no trajectory entries on the way down: the value found below is only backed up from the starting node.
int Tree::get_starting_node(Collector& col, int cur, uint32_t& depth, int& child_idx, LazyPos& lp) {
    const size_t d = get_random_depth(col);
    for (size_t cd = 0; cd < d; ++cd) {
        Node& n = nodes_[cur];
        int next;
        Move mv = chess::MOVE_NONE;
        {
            NodeLock lk(n, concurrent_);                     // currentNode->lock(), searchthread.cpp:148
            const int best = best_action_index_fast(n);
            child_idx = best;
            next = n.no_visit_idx ? n.child[best] : -1;
            if (next >= 0) mv = n.actions[best];
        }
        if (next < 0 || !ld_has_data(nodes_[next]) || ld_visits(nodes_[next]) < uint32_t(s_.epsilon_greedy_counter) ||
            ld_type(nodes_[next]) != NT_UNSOLVED)
            break;
        lp.step(next, mv);
        cur = next;
        ++depth;
    }
    return cur;
}

// random_playout (searchthread.cpp:124-142); the caller holds the node's lock
void Tree::random_playout(Collector& col, int cur, int& child_idx) {
    Node& n = nodes_[cur];
    if (size_t(n.no_visit_idx) == n.actions.size()) {            // is_fully_expanded
        const int idx = int(next_rand(col) % uint32_t(n.actions.size()));
        const int child = n.child[idx];
        if (child < 0 || !ld_has_data(nodes_[child])) { child_idx = idx; return; }
        if (ld_type(nodes_[child]) == NT_UNSOLVED) { child_idx = idx; return; }
        child_idx = -1;
    } else {
        child_idx = int(std::min(size_t(n.no_visit_idx), n.actions.size() - 1));
        increment_no_visit_idx(n);
    }
}

// select_enhanced_move (searchthread.cpp:451-473): make sure a checking move has been tried once; the caller holds the node's lock
int Tree::select_enhanced_move(int cur, const Position& pos) {
    Node& n = nodes_[cur];
    if (n.has_data && !n.inspected && !n.terminal) {
        const size_t first = n.no_visit_idx;
        for (size_t ci = first; ci < n.actions.size(); ++ci) {
            if (pos.gives_check(n.actions[ci])) {
                for (size_t idx = first; idx < ci + 1; ++idx) increment_no_visit_idx(n);
                return int(ci);
            }
        }
        n.inspected = true;
    }
    return -1;
}

// SearchThread::get_new_child_to_evaluate (searchthread.cpp:164-271), tree variant (useMCGS = false).  With several collectors the
// bookkeeping of a step -- selection, virtual loss, the trajectory entry, reading / reserving the child link -- happens under the
// node's lock as in the reference; the expansion itself (do_move, move generation, terminal test, policy indices) does not: the
// child link is marked CHILD_PENDING meanwhile, and a collector that runs into it has a collision exactly as if it had found the
// node without network results (what the reference's second thread finds once it gets the lock).
int Tree::get_new_child_to_evaluate(Collector& col, NodeBackup& type, uint32_t& depth, BoardDesc* desc_out) {
    depth = 0;
    int cur = 0;
    LazyPos lp(*this, col);
    int forced = -1;                             // childIdx chosen by the exploration step (uint16_t(-1) = none)
    if (s_.epsilon_greedy_counter && nodes_[0].has_data && next_rand(col) % uint32_t(s_.epsilon_greedy_counter) == 0) {
        cur = get_starting_node(col, cur, depth, forced, lp);
        NodeLock lk(nodes_[cur], concurrent_);
        random_playout(col, cur, forced);
    } else if (s_.epsilon_checks_counter && nodes_[0].has_data && next_rand(col) % uint32_t(s_.epsilon_checks_counter) == 0) {
        cur = get_starting_node(col, cur, depth, forced, lp);
        const Position& at_start = lp.get();
        NodeLock lk(nodes_[cur], concurrent_);
        forced = select_enhanced_move(cur, at_start);
        if (forced < 0) random_playout(col, cur, forced);
    }
    while (true) {
        int c, next;
        Move mv;
        {
            Node& n = nodes_[cur];
            NodeLock lk(n, concurrent_);
            c = forced >= 0 ? forced : select_child(n, col);
            forced = -1;
            apply_virtual_loss(n, c);
            col.trajectory_buffer.push_back(NodeAndIdx{cur, uint16_t(c)});
            next = n.child[c];
            mv = n.actions[c];
            ++depth;
            if (next == CHILD_NONE) {
                if (concurrent_) n.child[c] = CHILD_PENDING;
                increment_no_visit_idx(n);
            }
        }
        if (next == CHILD_NONE) {
            Position& pos = lp.get();
            CRA_TICK(t_expand);
            pos.do_move(mv);
            const int nn = new_node(pos);
            // MCTS_STORE_STATES: the new node keeps its position (before it is linked: nobody else sees the node yet)
            if (!nodes_[nn].terminal && stored_states_.load(std::memory_order_relaxed) < state_budget_) {
                nodes_[nn].state.reset(new Position(pos));
                stored_states_.fetch_add(1, std::memory_order_relaxed);
            }
            {
                Node& n = nodes_[cur];
                NodeLock lk(n, concurrent_);
                n.child[c] = nn;
            }
            if (nodes_[nn].terminal) {           // SearchThread::add_new_node_to_tree, searchthread.cpp:93-96
                type = NODE_TERMINAL;
                CRA_TOCK(g_expand_ticks, t_expand);
                return nn;
            }
            // newState->get_state_planes(true, ...), searchthread.cpp:229; the new node holds the legal moves already
            chess::pack_desc(pos, *desc_out, layout_needs_move_features(layout_), &nodes_[nn].actions);
            type = NODE_NEW_NODE;
            CRA_TOCK(g_expand_ticks, t_expand);
#ifdef CRA_REPLAY_PROFILE
            ++g_leaf_depth_hist[depth < 127 ? depth : 127];
#endif
            return nn;
        }
        if (next == CHILD_PENDING) { type = NODE_COLLISION; return -1; }
        if (nodes_[next].terminal) { type = NODE_TERMINAL; return next; }
        if (!ld_has_nn(nodes_[next])) { type = NODE_COLLISION; return next; }
        lp.step(next, mv);                        // (no position needed yet: the move is noted, or a stored state becomes the new base)
        cur = next;
    }
}

int Tree::collect(int quota, BoardDesc* descs, int ctx) {
#ifdef CRA_REPLAY_PROFILE
    struct Whole { unsigned long long t0 = __rdtsc(); ~Whole() { g_collect_ticks += __rdtsc() - t0; } } whole;
#endif
    Collector& col = *collectors_.at(size_t(ctx));
    size_t num_terminal = 0;
    const size_t terminal_cache = size_t(TERMINAL_NODE_CACHE_FACTOR) * size_t(std::max(quota, 1));
    int n_new = 0;
    if (nodes_[0].terminal || !nodes_[0].has_nn || root_solved()) return 0;                // is_root_node_unsolved, searchthread.cpp:333-340
    while (n_new < quota && col.collision_trajectories.size() != size_t(quota) && num_terminal < terminal_cache) {
        col.trajectory_buffer.clear();
        NodeBackup type;
        uint32_t depth;
        const int leaf = get_new_child_to_evaluate(col, type, depth, descs + n_new);
        col.depth_sum += depth;
        col.depth_max = std::max(col.depth_max, depth);
        if (type == NODE_TERMINAL) {
            ++num_terminal;
            backup_value(nodes_[leaf].value(), col.trajectory_buffer, true, s_.mcts_solver);   // backup_value<true>: terminal visits are free (searchthread.cpp:364-367)
        } else if (type == NODE_COLLISION) {
            col.collision_trajectories.push_back(col.trajectory_buffer);
        } else {
            col.new_nodes.push_back(leaf);
            col.new_trajectories.push_back(col.trajectory_buffer);
            ++n_new;
        }
    }
    return n_new;
}

void Tree::pending_policy_indices(int k, const uint16_t** idx, int* count, int ctx) const {
    const Node& n = nodes_[collectors_.at(size_t(ctx))->new_nodes.at(size_t(k))];
    *idx = n.policy_idx.data();
    *count = int(n.policy_idx.size());
}

void Tree::finish_batch(const float* values, const float* probs, int nb_policy, int ctx) {
    Collector& col = *collectors_.at(size_t(ctx));
    for (size_t i = 0; i < col.new_nodes.size(); ++i) fill_nn_result(nodes_[col.new_nodes[i]], values[i], probs + i * size_t(nb_policy));
    backup_batch(col);
}

void Tree::finish_batch_gathered(const float* values, const float* gathered, uint32_t stride, int ctx) {
    Collector& col = *collectors_.at(size_t(ctx));
    for (size_t i = 0; i < col.new_nodes.size(); ++i) fill_nn_result_gathered(nodes_[col.new_nodes[i]], values[i], gathered + i * size_t(stride));
    backup_batch(col);
}

void Tree::backup_batch(Collector& col) {
    for (size_t i = 0; i < col.new_nodes.size(); ++i) backup_value(nodes_[col.new_nodes[i]].value(), col.new_trajectories[i], false);
    col.new_nodes.clear();
    col.new_trajectories.clear();
    for (const Trajectory& t : col.collision_trajectories)                                // backup_collision, node.cpp:655-659
        for (auto it = t.rbegin(); it != t.rend(); ++it) {
            Node& n = nodes_[it->node];
            NodeLock lk(n, concurrent_);                                                  // Node::revert_virtual_loss locks, node.cpp:663
            revert_virtual_loss(n, it->child_idx);
        }
    col.collision_trajectories.clear();
}

// Node::get_mcts_policy (node.cpp:1070-1109)
int Tree::best_move_index(std::vector<double>* policy_out) const {
    const Node& n = nodes_[0];
    if (!n.has_data || n.no_visit_idx == 0) return -1;
    const int m = n.no_visit_idx;
    std::vector<double> pol(m);
    auto normalise = [&]() {
        double sum = 0;
        for (double v : pol) sum += v;
        if (sum == 0) {
            // every visited child is a proven win for the opponent while unvisited ones remain: the reference divides 0 / 0 here
            // (node.cpp:1107) and plays index 0 off a NaN policy; fall back to the plain visit counts instead
            for (int i = 0; i < m; ++i) { pol[i] = n.child_visits[i]; sum += pol[i]; }
            if (sum == 0) { pol.assign(size_t(m), 1.0 / m); sum = 1.0; }
        }
        int best = 0;
        for (int i = 0; i < m; ++i) {
            pol[i] /= sum;
            if (pol[i] > pol[best]) best = i;
        }
        if (policy_out) *policy_out = pol;
        return best;
    };
    if (n.node_type == NT_WIN) {                                 // mcts_policy_based_on_wins (node.cpp:299-322): every mating child
        for (int i = 0; i < m; ++i)
            pol[i] = (n.child[i] >= 0 && nodes_[n.child[i]].has_data && nodes_[n.child[i]].node_type == NT_LOSS) ? 1.0 : 0.0;
        return normalise();
    }
    if (n.node_type == NT_LOSS) {                                // mcts_policy_based_on_losses (node.cpp:324-341): delay the mate
        int longest_idx = 0;
        uint16_t longest = 0;
        for (int i = 0; i < m; ++i)
            if (n.child[i] >= 0 && nodes_[n.child[i]].has_data && nodes_[n.child[i]].end_in_ply > longest) {
                longest = nodes_[n.child[i]].end_in_ply;
                longest_idx = i;
            }
        pol[longest_idx] = 1.0;
        return normalise();
    }
    for (int i = 0; i < m; ++i) pol[i] = n.child_visits[i];
    if (n.unsolved_children != n.actions.size()) {               // prune_losses_in_mcts_policy (node.cpp:343-363)
        for (int i = 0; i < m; ++i)
            if (n.child[i] >= 0 && nodes_[n.child[i]].has_data && nodes_[n.child[i]].node_type == NT_WIN) pol[i] = 0;
    }
    int best_q = 0;
    for (int i = 1; i < m; ++i) if (n.q[i] > n.q[best_q]) best_q = i;
    // first_and_second_max (blazeutil.h:155-178): runner-up seeded with numeric_limits<double>::min()
    double first = pol[0], second = std::numeric_limits<double>::min();
    int first_arg = 0, second_arg = 0;
    for (int i = 1; i < m; ++i) {
        if (pol[i] > first) { second = first; second_arg = first_arg; first = pol[i]; first_arg = i; }
        else if (pol[i] > second) { second = pol[i]; second_arg = i; }
    }
    if (s_.q_value_weight > 0) {
        if (s_.q_veto_delta != 0 && best_q != first_arg && n.q[best_q] > n.q[first_arg] + s_.q_veto_delta && n.child_visits[best_q] > 1) {
            if (pol[first_arg] > pol[best_q]) std::swap(pol[best_q], pol[first_arg]);
        } else if (first_arg != second_arg && n.q[second_arg] > n.q[first_arg]) {
            const float q_diff = n.q[second_arg] - n.q[first_arg];
            pol[second_arg] += q_diff * s_.q_value_weight * pol[first_arg];
        }
    }
    return normalise();
}

void Tree::handle_single_move() {                               // mctsagent.cpp:277-290
    Node& r = nodes_[0];
    float target = last_value_eval_;
    if (last_stm_ != r.stm) target = -last_value_eval_;
    r.set_value(target);
    if (r.has_data && !r.q.empty()) r.q[0] = target;
}

float Tree::eval_best_move_q() const {
    const Node& r = nodes_[0];
    if (r.actions.size() == 1 && r.visit_sum == 0) {            // evalinfo.cpp:218-224: single move, blank root -> get_value_display
        if (r.has_data && r.node_type == NT_WIN) return WIN_VALUE;
        if (r.has_data && r.node_type == NT_LOSS) return LOSS_VALUE;
        if (r.has_data && r.node_type == NT_DRAW) return DRAW_VALUE;
        return r.real_visits ? r.value() : Q_INIT;
    }
    return best_move_q(best_move_index());
}

// value_to_centipawn (evalinfo.cpp:103-112): logarithmic pseudo-centipawns, base VALUE_TO_CENTI_PARAM (constants.h:89-93: 1.4 in
// MODE_CHESS builds, 1.2 otherwise); float arithmetic as there
static int value_to_centipawn(float value, int mode) {
    const int sg = (value > 0.f) - (value < 0.f);
    if (std::abs(value) >= 1) return sg * 9999;
    const float base = mode == MODE_CHESS ? 1.4f : 1.2f;
    return int(-(sg * std::log(1.0f - std::abs(value)) / std::log(base)) * 100.0f);
}

// set_eval_for_single_pv (evalinfo.cpp:120-182) for root child b: the line below it, bestMoveQ, mate distance, centipawns
void Tree::line_below_root_child(int b, std::vector<chess::Move>& pv, int* moves_to_mate, int* centipawns, float* q_out) const {
    const Node& r = nodes_[0];
    int mate = 0, cp = 0;
    pv.push_back(r.actions[size_t(b)]);
    float q = Q_INIT;
    bool scored = true;
    const int32_t first = b < int(r.child.size()) ? r.child[size_t(b)] : -1;
    if (first >= 0) {
        int32_t cur = first;
        while (cur >= 0 && nodes_[size_t(cur)].has_data && !nodes_[size_t(cur)].terminal) {      // Node::get_principal_variation
            const Node& n = nodes_[size_t(cur)];
            const int i = best_action_index_fast(n);
            pv.push_back(n.actions[size_t(i)]);
            cur = i < int(n.child.size()) ? n.child[size_t(i)] : -1;
        }
        const Node& next = nodes_[size_t(first)];
        q = best_move_q(b);
        if (next.has_data && next.node_type == NT_LOSS) { mate = (int(pv.size()) + 1) / 2; scored = false; }
        else if (next.has_data && next.node_type == NT_WIN) { mate = -(int(pv.size()) + 1) / 2; scored = false; }
    }
    if (scored) cp = value_to_centipawn(q, s_.mode);
    if (moves_to_mate) *moves_to_mate = mate;
    if (centipawns) *centipawns = cp;
    if (q_out) *q_out = q;
}

void Tree::principal_variation(std::vector<chess::Move>& pv, int* moves_to_mate, int* centipawns) const {
    pv.clear();
    if (moves_to_mate) *moves_to_mate = 0;
    if (centipawns) *centipawns = 0;
    const Node& r = nodes_[0];
    if (r.terminal || r.actions.empty()) return;
    if (r.actions.size() == 1 && r.visit_sum == 0) {                // single move, blank root (evalinfo.cpp:226-232)
        pv.push_back(r.actions[0]);
        if (centipawns) *centipawns = value_to_centipawn(eval_best_move_q(), s_.mode);
        return;
    }
    if (!r.has_data) return;
    const int b = best_move_index();
    if (b >= 0) line_below_root_child(b, pv, moves_to_mate, centipawns, nullptr);
}

bool Tree::principal_variation_multi(int idx, int multipv, std::vector<chess::Move>& pv, int* moves_to_mate, int* centipawns, float* q) const {
    pv.clear();
    if (moves_to_mate) *moves_to_mate = 0;
    if (centipawns) *centipawns = 0;
    if (q) *q = 0.f;
    const Node& r = nodes_[0];
    if (idx < 0 || multipv < 1 || r.terminal || r.actions.empty()) return false;
    if (r.actions.size() == 1 && r.visit_sum == 0) {                // single move, blank root: only pv[0] is filled
        if (idx != 0) return false;
        pv.push_back(r.actions[0]);
        const float v = eval_best_move_q();
        if (centipawns) *centipawns = value_to_centipawn(v, s_.mode);
        if (q) *q = v;
        return true;
    }
    if (!r.has_data) return false;
    const int max_idx = std::min(multipv, int(r.no_visit_idx));
    if (idx >= max_idx) return false;
    std::vector<double> pol;
    const int best = best_move_index(&pol);
    if (best < 0) return false;
    int b = best;
    if (idx > 0) {                                                   // sort_eval_lists: rank by the policy over ALL legal moves, as float
        pol.resize(r.actions.size(), 0.0);
        std::vector<int> order(r.actions.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = int(i);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return float(pol[size_t(x)]) > float(pol[size_t(y)]); });
        b = order[size_t(idx)];
    }
    line_below_root_child(b, pv, moves_to_mate, centipawns, q);
    return true;
}

void Tree::end_search() {                                       // tail of evaluate_board_state (mctsagent.cpp:337-339) over update_eval_info
    const Node& r = nodes_[0];
    if (r.terminal || r.actions.empty()) return;
    last_value_eval_ = eval_best_move_q();
    last_stm_ = r.stm;
}

float Tree::best_move_q(int b) const {
    const Node& r = nodes_[0];
    if (b < 0 || !r.has_data || b >= int(r.child.size()) || r.child[b] < 0) return Q_INIT;
    const Node& c = nodes_[r.child[b]];
    if (c.has_data) {                                           // Node::get_value_display (node.cpp:600-612)
        if (c.node_type == NT_WIN) return -float(WIN_VALUE);
        if (c.node_type == NT_LOSS) return -float(LOSS_VALUE);
        if (c.node_type == NT_DRAW) return -float(DRAW_VALUE);
    }
    return -c.value();
}

void Tree::dump(std::vector<uint32_t>& out) const {
    auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    std::vector<int32_t> stack{0};
    if (!nodes_[0].has_data) return;
    std::vector<int32_t> follow;
    while (!stack.empty()) {
        const Node& n = nodes_[stack.back()];
        stack.pop_back();
        const int m = n.terminal ? 0 : int(n.no_visit_idx);
        out.push_back(uint32_t(m));
        out.push_back(n.visit_sum);
        out.push_back(n.real_visits);
        out.push_back(n.free_visits);
        out.push_back(uint32_t(n.node_type));
        out.push_back(uint32_t(n.end_in_ply));
        out.push_back(n.terminal ? 1u : 0u);
        out.push_back(fbits(n.real_visits ? n.value() : 0.0f));
        follow.clear();
        for (int i = 0; i < m; ++i) {
            out.push_back(n.actions[i]);
            out.push_back(n.child_visits[i]);
            out.push_back(n.vl[i]);
            out.push_back(fbits(n.q[i]));
            out.push_back(fbits(n.priors[i]));
            const int32_t c = n.child[i];
            const uint32_t st = c < 0 ? 0u : nodes_[c].has_data ? 2u : 1u;
            out.push_back(st);
            if (st == 2u) follow.push_back(c);
        }
        for (auto it = follow.rbegin(); it != follow.rend(); ++it) stack.push_back(*it);   // preorder: first child's record next
    }
}

}  // namespace search
}  // namespace cra
