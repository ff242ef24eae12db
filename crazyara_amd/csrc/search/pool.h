// SearchPool: many independent trees feeding shared GPU batches through two pipeline lanes (see mcts.h header).
#pragma once
#include <atomic>
#include <exception>
#include <string>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "mcts.h"

namespace cra {
class RiseNet;
namespace search {

// One evaluation lane: submit() starts the evaluation of `n` descriptors and returns; wait() blocks until the
// value / probs host buffers of that submit are valid.
class Evaluator {
public:
    virtual ~Evaluator() {}
    virtual int batch_size() const = 0;
    virtual int nb_policy() const = 0;
    virtual BoardDesc* descs() = 0;        // [batch]   host (pinned for the HIP lane)
    virtual const float* values() = 0;     // [batch]
    virtual const float* probs() = 0;      // [batch][nb_policy]
    virtual void submit(int n_valid, int layout) = 0;
    virtual void wait() = 0;
    // Gathered priors: instead of bringing all nb_policy probabilities of every slot back to the host (5.3 MB per batch of 256
    // crazyhouse boards) the lane receives the policy indices of the new nodes' legal moves and returns just those entries.
    // Slot s owns gather_idx()[s * stride .. + gather_cnt()[s]) and the same range of gathered(): fixed stride, so that the trees
    // of a batch write their slots in parallel and nothing has to be compacted.  stride 0 = the lane does not gather.
    virtual uint32_t gather_stride() const = 0;
    virtual uint16_t* gather_idx() = 0;                // [batch][stride]
    virtual uint32_t* gather_cnt() = 0;                // [batch]
    virtual const float* gathered() = 0;               // [batch][stride]
    virtual void submit_gathered(int n_valid, int layout) = 0;   // values() and gathered() are valid after wait(); probs() is not
    // diagnostic (CRA_LANE_RECORD=1, HIP lanes): every batch this lane evaluated is sent through it again, alone on the device, and
    // compared bit for bit with what the host saw at the time; returns the number of differing words, details appended to `report`
    virtual size_t debug_replay(std::string* report) { (void)report; return 0; }
};

// HIP lane: RiseNet::submit_boards on the net's side stream (192 B/position H2D, planes built on the GPU, D2H of results)
std::unique_ptr<Evaluator> make_hip_evaluator(RiseNet* net);
// user-supplied lane (tests / alternative back ends): fn(user, descs, n, value, probs) fills the outputs synchronously
typedef int (*EvalFn)(void* user, const void* descs, int n, float* value, float* probs);
std::unique_ptr<Evaluator> make_callback_evaluator(EvalFn fn, void* user, int batch, int nb_policy);

struct SearchStats {
    uint64_t nodes = 0;          // sum over trees of (visits - freeVisits) gained in this run (evalinfo.cpp:73-80)
    uint64_t nn_evals = 0;       // positions sent to the evaluator (leaves + roots)
    uint64_t batches = 0;        // evaluator submits
    uint64_t simulations = 0;    // sum of root visits gained
    uint64_t collisions = 0;
    uint64_t terminal_visits = 0;
    double seconds = 0;
    double depth_avg = 0;
    uint32_t depth_max = 0;
};

// Fork/join over a fixed set of threads for work items of a few tens of microseconds (one tree's leaf collection).
// Workers spin on a generation counter (a condition-variable wake-up costs more than the work) and fall back to short sleeps
// when nothing arrives for a while; item i always runs on thread i % threads, so a tree stays in the caches of one core.
class WorkerPool {
public:
    explicit WorkerPool(int threads);
    ~WorkerPool();
    void parallel_for(int n, const std::function<void(int)>& fn);   // blocks; fn(i) for i in [0,n); rethrows the first exception
    int threads() const { return nthreads_; }
private:
    void worker_loop(int index);
    void run_items(int index);
    int nthreads_ = 1;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0;
    std::mutex err_m_;
    std::exception_ptr err_;
    std::atomic<int> generation_{0}, done_{0};
    std::atomic<bool> stop_{false};
};

class SearchPool {
public:
    SearchPool(const SearchSettings& s, std::unique_ptr<Evaluator> lane_a, std::unique_ptr<Evaluator> lane_b);
    // one more evaluator lane (before any position is added): one more batch in flight while the others are collected
    void add_lane(std::unique_ptr<Evaluator> lane);
    int add_position(const chess::Position& pos);
    // runs until every tree reached `simulations` root visits (if > 0) and/or `nodes` counted nodes (if > 0)
    // (SearchThread::nodes_limits_ok, searchthread.cpp:326-331)
    // movetime_ms > 0: the searches also end when that much wall time has passed since the start of the call (the timer of
    // ThreadManager::stop_search_based_on_limits, threadmanager.cpp:69-97, without its early-stopping heuristics); at least one of the
    // three limits must be given.  Batches in flight are applied before the call returns: the trees are consistent.
    void run(uint32_t simulations, uint32_t nodes, int threads, SearchStats* stats, uint32_t movetime_ms = 0);
    // Stop protocol (SearchThread::stop, searchthread.cpp:109-112; MCTSAgent::stop, mctsagent.cpp:364-373).  Every search has a
    // generation number.  announce_go() -- called by the thread that decides to search, BEFORE it hands run() to another thread --
    // opens the next generation; run() adopts the oldest announced generation that no run has taken yet (or opens one itself);
    // request_stop() from any thread stops EVERY generation announced or running at that moment (stop_gen_ = the newest one; a run
    // ends as if its limits had been reached when stop_gen_ >= its generation), also a search that has not entered run() yet: a stop is
    // sticky for its search and never reaches a later one.  There is no shared "state" word to lose: searches may overlap at the
    // protocol level -- UCI `stop` followed directly by `go` (or ponderhit) announces the next go while the previous run() is still
    // returning; that run's exit touches nothing the announcement or a later stop depends on (round 4's version stored "idle" over it).
    // With nothing announced or running a stop names only generations that are over (MCTSAgent::stop: `if (!isRunning) return`).
    void announce_go() {
        std::lock_guard<std::mutex> g(gen_mu_);
        ++go_gen_;
    }
    // An announced search that will NOT be run (the commanding thread dropped it: its search thread failed to start, or the go was
    // stopped and discarded before it began): without this the announcement would stay the oldest un-run generation for the life of
    // the pool, and the next run() would adopt it -- together with any stop it has collected.  Returns false when no
    // announced generation is waiting for its run.
    bool cancel_go() {
        std::lock_guard<std::mutex> g(gen_mu_);
        if (adopted_gen_ >= go_gen_) return false;
        ++adopted_gen_;                                                // the oldest waiting generation is over
        return true;
    }
    void request_stop() {
        std::lock_guard<std::mutex> g(gen_mu_);
        stop_gen_.store(go_gen_, std::memory_order_release);          // go_gen_ only grows: so does stop_gen_
    }
    // evaluates the roots that have no network result yet (new games, restarted trees) through the first lane, as run() does first;
    // a game loop reads the raw policy of fresh positions from the root priors this leaves (RawNetAgent::evaluate_board_state)
    void evaluate_new_roots(SearchStats* stats) { SearchStats st; evaluate_roots(&st.nn_evals, &st.batches); if (stats) *stats = st; }
    Tree& tree(int i) { return *trees_.at(i); }
    // fn(i) for i in [0, n) on the pool's worker threads (the game loops play the moves of their concurrent games this way: each game
    // touches only its own tree); between runs only
    void parallel_for(int n, int threads, const std::function<void(int)>& fn);
    void reset_position(int i, const chess::Position& pos);   // a new game in slot i (same lane, same exploration stream seed)
    // trees that sit out the following runs (an arena game whose other player is to move): they keep their state
    void set_active(int i, bool active);
    // One tree, many collectors -- the reference's `Threads` SearchThreads on ONE tree (crazyara.cpp:555-561, searchthread.cpp:403-416):
    // every tree gets k >= 1 collectors IN EVERY LANE; a lane's batch is the concatenation of its collectors' mini-batches, collected in
    // parallel on the worker threads under per-node locks with virtual loss keeping them apart.  k = 0 (default): every tree has one
    // collector and lives in one lane (the many-trees mode: no locks).  Call between runs.
    void set_shared_collectors(int k);
    int shared_collectors() const { return shared_k_; }
    // Many-trees mode, throughput setting (self-play): cap > 0 lets the trees of a lane that are still searching share the WHOLE batch
    // -- each gets batch / (running trees) slots, at most `cap`, and never more than it still needs to reach its limit -- instead of the
    // fixed batch / (trees of the lane) of the default (0), under which a tree's batches, and so its statistics, do not depend on the
    // other trees.  With the absolute limits of tree reuse the trees of a round need very different numbers of simulations; the fixed
    // quota then ends a round with mostly empty batches for the one tree that needs the most.
    void set_adaptive_quota(int cap) { adaptive_cap_ = cap < 0 ? 0 : cap; }
    // stored leaf states per tree (Tree::set_state_budget; the reference's MCTS_STORE_STATES): 0 = every simulation replays its path from
    // the root.  Applies to the trees of the pool and to the ones added later; between runs.
    void set_state_budget(uint32_t budget) {
        state_budget_ = int64_t(budget);
        for (auto& t : trees_) t->set_state_budget(budget);
    }
    // Per-tree limits for the following runs, replacing run()'s simulations / nodes for that tree (both 0 = back to run()'s): the
    // concurrent games of a self-play loop search with their own node budgets -- quick searches and the per-move node jitter of
    // SelfPlay::generate_game (selfplay.cpp:146-152,213-221), which the reference sets on its one SearchLimits before every move.
    void set_tree_limits(int i, uint32_t simulations, uint32_t nodes) {
        (void)trees_.at(size_t(i));
        if (tree_limits_.size() < trees_.size()) tree_limits_.resize(trees_.size(), {0u, 0u});
        tree_limits_[size_t(i)] = {simulations, nodes};
    }
    int adaptive_quota() const { return adaptive_cap_; }
    int n_trees() const { return int(trees_.size()); }
    size_t debug_replay(std::string* report) {                   // Evaluator::debug_replay of every lane (between runs)
        size_t bad = 0;
        for (Lane& lane : lanes_) bad += lane.eval->debug_replay(report);
        return bad;
    }
    const SearchSettings& settings() const { return s_; }

private:
    struct Item { int tree; int ctx; };    // one collector of one tree: the unit that owns a slot range of a batch
    struct Lane {
        std::unique_ptr<Evaluator> eval;
        std::vector<int> trees;            // item ids (collectors) assigned to this lane; one per tree unless trees are shared
        std::vector<int> slot_begin, slot_count, n_new, batch_ids;
        bool in_flight = false;
        bool gathered = false;             // the batch in flight returns gathered priors
        bool same_trees_next = false;      // the next batch can keep this batch's trees and slots (no rotation, no tree finished)
    };
    bool tree_done(const Tree& t, uint32_t simulations, uint32_t nodes) const;
    // the limits tree `id` searches under: its own (set_tree_limits) or the run's
    void limits_of(int id, uint32_t& simulations, uint32_t& nodes) const {
        if (size_t(id) < tree_limits_.size() && (tree_limits_[size_t(id)].first || tree_limits_[size_t(id)].second)) {
            simulations = tree_limits_[size_t(id)].first;
            nodes = tree_limits_[size_t(id)].second;
        }
    }
    std::vector<std::pair<uint32_t, uint32_t>> tree_limits_;
    void evaluate_roots(uint64_t* evals, uint64_t* batches);
    void rebuild_items();
    Tree& item_tree(int item) { return *trees_[size_t(items_[size_t(item)].tree)]; }
    int item_ctx(int item) const { return items_[size_t(item)].ctx; }
    std::vector<Item> items_;
    int shared_k_ = 0;
    int adaptive_cap_ = 0;
    int64_t state_budget_ = -1;                          // set_state_budget; -1 = the Tree's default
    std::mutex gen_mu_;                                  // stop protocol (above): guards go_gen_ / adopted_gen_; never taken on the search's hot path
    uint64_t go_gen_ = 0, adopted_gen_ = 0;              // the newest generation announced or opened; the newest one a run() has taken
    std::atomic<uint64_t> stop_gen_{0};                  // every generation <= this one has been told to stop
    uint64_t open_generation();                          // run(): the oldest announced generation no run has taken, or a new one
    SearchSettings s_;
    int layout_;
    std::vector<std::unique_ptr<Tree>> trees_;
    std::vector<Lane> lanes_;
    std::vector<uint8_t> paused_;
    std::unique_ptr<WorkerPool> workers_;   // kept between runs (a game loop calls run() once per move)
    bool is_paused(int id) const { return size_t(id) < paused_.size() && paused_[id] != 0; }
};

}  // namespace search
}  // namespace cra
