#include "pool.h"

#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../nn/rise_net.h"

namespace cra {
namespace search {

// ---------------------------------------------------------------------------------------------------------------------
// evaluators
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// room for the gathered priors of a slot: 160 legal moves (crazyhouse middlegames have 40-120; a batch with a position that has
// more falls back to the full probability vectors)
constexpr uint32_t kGatherPerSlotDefault = 160;
uint32_t gather_per_slot() {            // CRA_GATHER_PER_SLOT: 0 = always whole probability vectors (A/B, tests), n = room per slot
    const char* e = getenv("CRA_GATHER_PER_SLOT");
    return e ? uint32_t(std::max(0, atoi(e))) : kGatherPerSlotDefault;
}

class HipEvaluator : public Evaluator {
public:
    explicit HipEvaluator(RiseNet* net) : net_(net) {
        const RiseDesign& d = net->design();
        batch_ = d.batch;
        nb_policy_ = d.nb_policy;
        auto pinned = [](size_t bytes) {
            void* p = nullptr;
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) throw std::runtime_error("hipHostMalloc failed");
            return p;
        };
        descs_ = static_cast<BoardDesc*>(pinned(sizeof(BoardDesc) * batch_));
        std::memset(descs_, 0, sizeof(BoardDesc) * batch_);
        values_ = static_cast<float*>(pinned(sizeof(float) * batch_));
        probs_ = static_cast<float*>(pinned(sizeof(float) * size_t(batch_) * nb_policy_));
        aux_ = d.nb_aux ? static_cast<float*>(pinned(sizeof(float) * size_t(batch_) * d.nb_aux)) : nullptr;
        gstride_ = gather_per_slot();
        const size_t gcap = std::max<size_t>(size_t(batch_) * gstride_, 1);
        gidx_ = static_cast<uint16_t*>(pinned(sizeof(uint16_t) * gcap));
        gcnt_ = static_cast<uint32_t*>(pinned(sizeof(uint32_t) * size_t(batch_)));
        gout_ = static_cast<float*>(pinned(sizeof(float) * gcap));
    }
    ~HipEvaluator() override {
        (void)hipHostFree(descs_);
        (void)hipHostFree(values_);
        (void)hipHostFree(probs_);
        if (aux_) (void)hipHostFree(aux_);
        (void)hipHostFree(gidx_);
        (void)hipHostFree(gcnt_);
        (void)hipHostFree(gout_);
    }
    int batch_size() const override { return batch_; }
    int nb_policy() const override { return nb_policy_; }
    BoardDesc* descs() override { return descs_; }
    const float* values() override { return values_; }
    const float* probs() override { return probs_; }
    void submit(int n_valid, int layout) override {
        if (record_) before_submit(n_valid, layout, false);
        net_->submit_boards(descs_, n_valid, layout, values_, probs_, aux_);
    }
    void wait() override {
        net_->wait();
        if (record_ && pending_) after_wait();
    }
    uint32_t gather_stride() const override { return gstride_; }
    uint16_t* gather_idx() override { return gidx_; }
    uint32_t* gather_cnt() override { return gcnt_; }
    const float* gathered() override { return gout_; }
    void submit_gathered(int n_valid, int layout) override {
        if (record_) before_submit(n_valid, layout, true);
        net_->submit_boards_gathered(descs_, n_valid, layout, gidx_, gcnt_, gstride_, values_, gout_, aux_);
    }
    size_t debug_replay(std::string* report) override;

private:
    // ---- CRA_LANE_RECORD=1 (diagnostic: tests/test_lane_determinism_gpu.py, scripts/lane_divergence.py) ----
    // Every batch of the lane is kept: what went in (descriptors, gather lists) and what the host found in the output buffers when
    // wait() returned (`seen`) and again when the next batch was submitted or the replay started (`late`: a result that lands after
    // the stream was reported idle shows up as a difference between the two).  The outputs are poisoned before every submit, so a
    // result the GPU never wrote is visible as such.  debug_replay() then sends every recorded batch through the lane again, alone
    // on the device, and compares bit for bit.
    struct Record {
        int n_valid = 0, layout = 0;
        bool gathered = false;
        std::vector<BoardDesc> descs;
        std::vector<uint16_t> gidx;
        std::vector<uint32_t> gcnt;
        std::vector<uint32_t> seen_v, seen_o, late_v, late_o;     // bit patterns of values / (gathered priors | whole vectors)
    };
    static constexpr uint32_t kPoison = 0x7fc0dead;
    size_t out_words(const Record& r) const { return r.gathered ? size_t(r.n_valid) * gstride_ : size_t(r.n_valid) * nb_policy_; }
    const float* out_buf(const Record& r) const { return r.gathered ? gout_ : probs_; }
    void snapshot(const Record& r, std::vector<uint32_t>& v, std::vector<uint32_t>& o) const {
        v.resize(size_t(r.n_valid));
        std::memcpy(v.data(), values_, v.size() * 4);
        o.resize(out_words(r));
        std::memcpy(o.data(), out_buf(r), o.size() * 4);
    }
    void poison(const Record& r) {
        uint32_t* v = reinterpret_cast<uint32_t*>(values_);
        for (int i = 0; i < r.n_valid; ++i) v[i] = kPoison;
        uint32_t* o = reinterpret_cast<uint32_t*>(const_cast<float*>(out_buf(r)));
        for (size_t i = 0; i < out_words(r); ++i) o[i] = kPoison;
    }
    void close_last() {
        if (!records_.empty() && records_.back().late_v.empty() && !pending_) snapshot(records_.back(), records_.back().late_v, records_.back().late_o);
    }
    void before_submit(int n_valid, int layout, bool gathered) {
        close_last();
        Record r;
        r.n_valid = n_valid;
        r.layout = layout;
        r.gathered = gathered;
        r.descs.assign(descs_, descs_ + n_valid);
        if (gathered) {
            r.gidx.assign(gidx_, gidx_ + size_t(n_valid) * gstride_);
            r.gcnt.assign(gcnt_, gcnt_ + n_valid);
        }
        poison(r);
        records_.push_back(std::move(r));
        pending_ = true;
    }
    void after_wait() {
        pending_ = false;
        snapshot(records_.back(), records_.back().seen_v, records_.back().seen_o);
    }
    bool record_ = getenv("CRA_LANE_RECORD") != nullptr;
    bool pending_ = false;
    std::vector<Record> records_;

    RiseNet* net_;
    int batch_ = 0, nb_policy_ = 0;
    BoardDesc* descs_ = nullptr;
    float *values_ = nullptr, *probs_ = nullptr, *aux_ = nullptr;
    uint32_t gstride_ = 0;
    uint16_t* gidx_ = nullptr;
    uint32_t* gcnt_ = nullptr;
    float* gout_ = nullptr;
};

class CallbackEvaluator : public Evaluator {
public:
    CallbackEvaluator(EvalFn fn, void* user, int batch, int nb_policy)
        : fn_(fn), user_(user), batch_(batch), nb_policy_(nb_policy), descs_(batch), values_(batch), probs_(size_t(batch) * nb_policy),
          gstride_(gather_per_slot()), gidx_(size_t(batch) * gstride_), gcnt_(size_t(batch)), gout_(size_t(batch) * gstride_) {
        std::memset(descs_.data(), 0, sizeof(BoardDesc) * batch);
    }
    int batch_size() const override { return batch_; }
    int nb_policy() const override { return nb_policy_; }
    BoardDesc* descs() override { return descs_.data(); }
    const float* values() override { return values_.data(); }
    const float* probs() override { return probs_.data(); }
    void submit(int n_valid, int) override {
        if (fn_(user_, descs_.data(), n_valid, values_.data(), probs_.data()) != 0) throw std::runtime_error("evaluator callback failed");
    }
    void wait() override {}
    // the callback fills whole probability vectors; the gather the HIP lane runs on the GPU happens here on the host, so that
    // every pool driven by a callback (the CPU tests, the CPU baseline) takes the same path through the pool as the GPU lanes
    uint32_t gather_stride() const override { return gstride_; }
    uint16_t* gather_idx() override { return gidx_.data(); }
    uint32_t* gather_cnt() override { return gcnt_.data(); }
    const float* gathered() override { return gout_.data(); }
    void submit_gathered(int n_valid, int layout) override {
        submit(n_valid, layout);
        for (int s = 0; s < n_valid; ++s)
            for (uint32_t j = 0; j < gcnt_[size_t(s)]; ++j)
                gout_[size_t(s) * gstride_ + j] = probs_[size_t(s) * nb_policy_ + gidx_[size_t(s) * gstride_ + j]];
    }

private:
    EvalFn fn_;
    void* user_;
    int batch_, nb_policy_;
    std::vector<BoardDesc> descs_;
    std::vector<float> values_, probs_;
    uint32_t gstride_;
    std::vector<uint16_t> gidx_;
    std::vector<uint32_t> gcnt_;
    std::vector<float> gout_;
};
size_t HipEvaluator::debug_replay(std::string* report) {
    if (!record_) return 0;
    if (pending_) throw std::logic_error("debug_replay with a batch in flight");
    close_last();
    record_ = false;                                  // the replays themselves are not recorded
    size_t bad = 0;
    char line[512];
    auto say = [&](const char* what, size_t batch, int slot, long entry, uint32_t a, uint32_t b, const char* note) {
        ++bad;
        if (!report || report->size() > (1u << 16)) return;
        float fa, fb;
        std::memcpy(&fa, &a, 4);
        std::memcpy(&fb, &b, 4);
        snprintf(line, sizeof line, "%s batch %zu slot %d entry %ld: %08x (%.9g) vs %08x (%.9g)%s\n", what, batch, slot, entry, a, double(fa), b, double(fb), note);
        *report += line;
    };
    for (size_t k = 0; k < records_.size(); ++k) {
        const Record& r = records_[k];
        if (r.seen_v.empty() && r.n_valid > 0) continue;
        std::memcpy(descs_, r.descs.data(), r.descs.size() * sizeof(BoardDesc));
        if (r.gathered) {
            std::memcpy(gidx_, r.gidx.data(), r.gidx.size() * sizeof(uint16_t));
            std::memset(gcnt_, 0, sizeof(uint32_t) * size_t(batch_));
            std::memcpy(gcnt_, r.gcnt.data(), r.gcnt.size() * sizeof(uint32_t));
        }
        poison(r);
        if (r.gathered) net_->submit_boards_gathered(descs_, r.n_valid, r.layout, gidx_, gcnt_, gstride_, values_, gout_, aux_);
        else net_->submit_boards(descs_, r.n_valid, r.layout, values_, probs_, aux_);
        net_->wait();
        std::vector<uint32_t> v, o;
        snapshot(r, v, o);
        const size_t per = r.gathered ? gstride_ : size_t(nb_policy_);
        for (int s = 0; s < r.n_valid; ++s) {
            const size_t cnt = r.gathered ? r.gcnt[size_t(s)] : per;
            if (r.gathered && cnt == 0) continue;          // (a slot without a new node: nothing is written, nothing is read)
            // where does a wrong value come from?  the batch before (stale buffer), another slot of this batch, poison (never written)
            auto origin = [&](uint32_t got) -> const char* {
                if (got == kPoison) return "  [poison: never written]";
                if (k > 0 && size_t(s) < records_[k - 1].seen_v.size() && records_[k - 1].seen_v[size_t(s)] == got) return "  [= this slot's value of the batch before]";
                for (int t = 0; t < r.n_valid; ++t)
                    if (t != s && v[size_t(t)] == got) return "  [= another slot's value of this batch]";
                return "";
            };
            if (r.seen_v[size_t(s)] != v[size_t(s)]) say("value   seen/replay", k, s, -1, r.seen_v[size_t(s)], v[size_t(s)], origin(r.seen_v[size_t(s)]));
            if (r.late_v[size_t(s)] != r.seen_v[size_t(s)]) say("value   seen/late  ", k, s, -1, r.seen_v[size_t(s)], r.late_v[size_t(s)], "");
            for (size_t j = 0; j < cnt; ++j) {
                const size_t i = size_t(s) * per + j;
                if (r.seen_o[i] != o[i]) say("prior   seen/replay", k, s, long(j), r.seen_o[i], o[i], r.seen_o[i] == kPoison ? "  [poison: never written]" : "");
                if (r.late_o[i] != r.seen_o[i]) say("prior   seen/late  ", k, s, long(j), r.seen_o[i], r.late_o[i], "");
            }
        }
    }
    if (report) {
        snprintf(line, sizeof line, "lane replay: %zu batches, %zu differing words\n", records_.size(), bad);
        *report += line;
    }
    records_.clear();
    record_ = true;
    return bad;
}
}  // namespace

std::unique_ptr<Evaluator> make_hip_evaluator(RiseNet* net) { return std::unique_ptr<Evaluator>(new HipEvaluator(net)); }
std::unique_ptr<Evaluator> make_callback_evaluator(EvalFn fn, void* user, int batch, int nb_policy) {
    return std::unique_ptr<Evaluator>(new CallbackEvaluator(fn, user, batch, nb_policy));
}

// ---------------------------------------------------------------------------------------------------------------------
// worker pool
// ---------------------------------------------------------------------------------------------------------------------
WorkerPool::WorkerPool(int threads) : nthreads_(std::max(1, threads)) {
    workers_.reserve(size_t(nthreads_ - 1));
    for (int i = 1; i < nthreads_; ++i) workers_.emplace_back([this, i] { worker_loop(i); });
}
void WorkerPool::run_items(int index) {
    try {
        for (int i = index; i < n_; i += nthreads_) (*fn_)(i);
    } catch (...) {
        std::lock_guard<std::mutex> lk(err_m_);
        if (!err_) err_ = std::current_exception();
    }
}
WorkerPool::~WorkerPool() {
    stop_.store(true, std::memory_order_release);
    for (auto& t : workers_) t.join();
}
void WorkerPool::worker_loop(int index) {
    int seen = 0;
    while (true) {
        // Idle policy by elapsed time (a `pause` is 10-60 ns depending on the core, so counting them is not a clock): spin for
        // 0.8 ms -- longer than a forward, so within a run a worker never sleeps between the jobs of consecutive batches (a nap
        // costs its wake-up latency on the next batch's critical path) --, then 50 us naps for 100 ms (between two runs of a game
        // loop), then 2 ms naps (idle pool).
        int spins = 0;
        auto idle_since = std::chrono::steady_clock::now();
        int phase = 0;
        while (generation_.load(std::memory_order_acquire) == seen) {
            if (stop_.load(std::memory_order_acquire)) return;
            if (phase == 0) {
                __builtin_ia32_pause();
                if ((++spins & 63) == 0 && std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(800)) phase = 1;
            } else if (phase == 1) {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
                if (std::chrono::steady_clock::now() - idle_since > std::chrono::milliseconds(100)) phase = 2;
            } else {
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        seen = generation_.load(std::memory_order_acquire);
        run_items(index);
        done_.fetch_add(1, std::memory_order_release);
    }
}
void WorkerPool::parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (workers_.empty() || n == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    fn_ = &fn;
    n_ = n;
    done_.store(0, std::memory_order_relaxed);
    generation_.fetch_add(1, std::memory_order_release);
    run_items(0);
    while (done_.load(std::memory_order_acquire) != int(workers_.size())) __builtin_ia32_pause();
    if (err_) {
        std::exception_ptr e = err_;
        err_ = nullptr;
        std::rethrow_exception(e);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// pool
// ---------------------------------------------------------------------------------------------------------------------
SearchPool::SearchPool(const SearchSettings& s, std::unique_ptr<Evaluator> lane_a, std::unique_ptr<Evaluator> lane_b) : s_(s) {
    layout_ = layout_for(s.mode, s.version_major, s.version_minor);
    if (!lane_a) throw std::invalid_argument("SearchPool needs at least one evaluator lane");
    lanes_.emplace_back();
    lanes_.back().eval = std::move(lane_a);
    if (lane_b) {
        lanes_.emplace_back();
        lanes_.back().eval = std::move(lane_b);
    }
}

void SearchPool::add_lane(std::unique_ptr<Evaluator> lane) {
    if (!lane) throw std::invalid_argument("null evaluator lane");
    if (!trees_.empty()) throw std::logic_error("lanes must be added before positions");
    if (lane->batch_size() != lanes_[0].eval->batch_size() || lane->nb_policy() != lanes_[0].eval->nb_policy())
        throw std::invalid_argument("evaluator lanes must agree in batch size and policy length");
    lanes_.emplace_back();
    lanes_.back().eval = std::move(lane);
}

int SearchPool::add_position(const chess::Position& pos) {
    SearchSettings st = s_;
    st.seed = s_.seed + uint32_t(trees_.size());     // every tree owns its exploration stream
    trees_.emplace_back(new Tree(pos, st));
    if (state_budget_ >= 0) trees_.back()->set_state_budget(uint32_t(state_budget_));
    rebuild_items();
    return int(trees_.size()) - 1;
}

void SearchPool::parallel_for(int n, int threads, const std::function<void(int)>& fn) {
    if (!workers_ || workers_->threads() != std::max(1, threads)) workers_.reset(new WorkerPool(std::max(1, threads)));
    workers_->parallel_for(n, fn);
}

void SearchPool::reset_position(int i, const chess::Position& pos) {
    SearchSettings st = s_;
    st.seed = s_.seed + uint32_t(i);
    trees_.at(i).reset(new Tree(pos, st));
    if (state_budget_ >= 0) trees_[i]->set_state_budget(uint32_t(state_budget_));
    if (shared_k_ > 0) trees_[i]->set_collectors(shared_k_ * int(lanes_.size()));
}

void SearchPool::set_shared_collectors(int k) {
    if (k < 0 || k > 256) throw std::invalid_argument("shared collectors per tree and lane must be in [0, 256]");
    for (const Lane& lane : lanes_)
        if (lane.in_flight) throw std::logic_error("set_shared_collectors with a batch in flight");
    shared_k_ = k;
    rebuild_items();
}

// items = (tree, collector) pairs and their lanes.  Many-trees mode (k = 0): one item per tree, tree t in lane t % lanes.  Shared mode
// (k >= 1): every tree has k collectors in every lane (collector index = lane * k + j).
void SearchPool::rebuild_items() {
    items_.clear();
    for (Lane& lane : lanes_) lane.trees.clear();
    const int L = int(lanes_.size());
    for (int t = 0; t < int(trees_.size()); ++t) {
        if (shared_k_ <= 0) {
            trees_[t]->set_collectors(1);
            items_.push_back(Item{t, 0});
            lanes_[size_t(t % L)].trees.push_back(int(items_.size()) - 1);
        } else {
            trees_[t]->set_collectors(shared_k_ * L);
            for (int l = 0; l < L; ++l)
                for (int j = 0; j < shared_k_; ++j) {
                    items_.push_back(Item{t, l * shared_k_ + j});
                    lanes_[size_t(l)].trees.push_back(int(items_.size()) - 1);
                }
        }
    }
}

void SearchPool::set_active(int i, bool active) {
    (void)trees_.at(i);
    if (paused_.size() < trees_.size()) paused_.resize(trees_.size(), 0);
    paused_[i] = active ? 0 : 1;
}

bool SearchPool::tree_done(const Tree& t, uint32_t simulations, uint32_t nodes) const {
    if (t.root().terminal || t.root_solved()) return true;
    if (simulations && t.root_visits() >= simulations) return true;
    if (nodes && t.node_count() >= nodes) return true;
    return false;
}

// roots without network results (new games, restarted trees): evaluated through the first lane, a batch at a time
void SearchPool::evaluate_roots(uint64_t* evals, uint64_t* batches) {
    Evaluator& ev = *lanes_[0].eval;
    std::vector<int> todo;
    for (int id = 0; id < int(trees_.size()); ++id)
        if (trees_[id]->root_needs_eval() && !is_paused(id)) todo.push_back(id);
    for (size_t off = 0; off < todo.size(); off += ev.batch_size()) {
        const int n = int(std::min(todo.size() - off, size_t(ev.batch_size())));
        for (int i = 0; i < n; ++i) trees_[todo[off + i]]->root_desc(ev.descs()[i]);
        ev.submit(n, layout_);
        ev.wait();
        for (int i = 0; i < n; ++i) trees_[todo[off + i]]->set_root_result(ev.values()[i], ev.probs() + size_t(i) * ev.nb_policy());
        ++*batches;
    }
    *evals += todo.size();
}

uint64_t SearchPool::open_generation() {
    std::lock_guard<std::mutex> g(gen_mu_);
    if (adopted_gen_ < go_gen_) return ++adopted_gen_;            // an announced go nobody has run yet (a stop sent since names it already)
    adopted_gen_ = ++go_gen_;
    return adopted_gen_;
}

void SearchPool::run(uint32_t simulations, uint32_t nodes, int threads, SearchStats* stats, uint32_t movetime_ms) {
    // this run's generation: the announced one (a stop sent since the announcement already names it) or a new one
    const uint64_t my_gen = open_generation();
    if (!simulations && !nodes && !movetime_ms) throw std::invalid_argument("run needs a simulations, a nodes or a movetime limit");
    if (!workers_ || workers_->threads() != std::max(1, threads)) workers_.reset(new WorkerPool(std::max(1, threads)));
    WorkerPool& workers = *workers_;
    SearchStats st;
    std::vector<uint32_t> nodes_pre(trees_.size()), visits_pre(trees_.size());
    std::vector<uint64_t> depth_pre(trees_.size());
    const auto t0 = std::chrono::steady_clock::now();
    evaluate_roots(&st.nn_evals, &st.batches);
    std::vector<char> single_move(trees_.size(), 0);
    for (size_t i = 0; i < trees_.size(); ++i) {
        if (!is_paused(int(i))) {
            if (trees_[i]->single_move_root()) {                    // "Only single move available -> early stopping", mctsagent.cpp:303-307
                trees_[i]->handle_single_move();
                single_move[i] = 1;
            } else {
                trees_[i]->begin_search();                          // Dirichlet noise + full expansion of the root (RL settings)
            }
        }
        nodes_pre[i] = trees_[i]->node_count();
        visits_pre[i] = trees_[i]->root_visits();
        depth_pre[i] = trees_[i]->depth_sum();
        trees_[i]->reset_depth_max();                               // reset_stats() at the start of a go (searchthread.cpp:283-288)
    }
    // simulations / nodes limits are ABSOLUTE on the root's counters, as SearchThread::nodes_limits_ok has them
    // (searchthread.cpp:326-331: rootNode->get_visits() < simulations, get_node_count() < nodes): visits inherited through tree
    // reuse count towards the limit of the next go
    const auto deadline = t0 + std::chrono::milliseconds(movetime_ms);
    std::atomic<bool> time_up_flag{false};
    auto halted = [&]() {                                           // request_stop() or the movetime: every tree is "done"
        if (stop_gen_.load(std::memory_order_acquire) >= my_gen) return true;
        if (time_up_flag.load(std::memory_order_relaxed)) return true;
        if (movetime_ms && std::chrono::steady_clock::now() >= deadline) {
            time_up_flag.store(true, std::memory_order_relaxed);
            return true;
        }
        return false;
    };
    auto done = [&](int item) {                                     // per item = per (tree, collector): the tree's verdict
        const int id = items_[size_t(item)].tree;
        if (is_paused(id) || single_move[id]) return true;
        if (halted()) return true;
        const Tree& t = *trees_[id];
        if (t.root().terminal || t.root_solved()) return true;      // is_root_node_unsolved(), searchthread.cpp:333-340
        uint32_t sims_t = simulations, nodes_t = nodes;
        limits_of(id, sims_t, nodes_t);
        return tree_done(t, sims_t, nodes_t);
    };

    double t_par = 0, t_submit = 0, t_item_max = 0, t_item_sum = 0, t_wait = 0;
    const bool timing = getenv("CRA_POOL_TIMING") != nullptr;
    const bool two_step = getenv("CRA_POOL_TWO_STEP") != nullptr;     // development A/B (once per run, not per lane step)
    std::atomic<bool> gather_overflow{false};
    // simulations / nodes the tree still lacks (>= 1 for a tree that is not done)
    auto remaining_need = [&](const Tree& t, int tree_id) -> int {
        uint32_t need = 0xffffffffu, sims_t = simulations, nodes_t = nodes;
        limits_of(tree_id, sims_t, nodes_t);
        if (sims_t) need = std::min(need, sims_t > t.root_visits() ? sims_t - t.root_visits() : 1u);
        if (nodes_t) need = std::min(need, nodes_t > t.node_count() ? nodes_t - t.node_count() : 1u);
        return int(std::min<uint32_t>(need, 1u << 20));
    };
    // one tree's share of a batch: leaves into its slots; the policy indices of the new nodes' legal moves go straight into the
    // lane's gather list (this thread just wrote them)
    auto collect_item = [&](Lane& lane, int i, int id) {
        Evaluator& ev = *lane.eval;
        const uint32_t gstride = ev.gather_stride();
        Tree& tree = item_tree(id);
        const int ctx = item_ctx(id);
        int take = lane.slot_count[i];
        if (adaptive_cap_ > 0 && shared_k_ == 0) take = std::min(take, remaining_need(tree, items_[size_t(id)].tree));   // no overshoot beyond the limit
        lane.n_new[i] = tree.collect(take, ev.descs() + lane.slot_begin[i], ctx);
        if (gstride) {
            for (int k = 0; k < lane.slot_count[i]; ++k) {
                const size_t slot = size_t(lane.slot_begin[i] + k);
                uint32_t cnt = 0;
                if (k < lane.n_new[i]) {
                    const uint16_t* src = nullptr;
                    int c = 0;
                    tree.pending_policy_indices(k, &src, &c, ctx);
                    if (uint32_t(c) > gstride) gather_overflow.store(true, std::memory_order_relaxed);
                    else {
                        std::memcpy(ev.gather_idx() + slot * gstride, src, size_t(c) * sizeof(uint16_t));
                        cnt = uint32_t(c);
                    }
                }
                ev.gather_cnt()[slot] = cnt;
            }
        }
    };
    // the results of the batch a tree took part in: priors + values into its new nodes, backups
    auto finish_item = [&](Lane& lane, bool gathered, int i, int id) {
        Evaluator& ev = *lane.eval;
        if (gathered) item_tree(id).finish_batch_gathered(ev.values() + lane.slot_begin[i], ev.gathered() + size_t(lane.slot_begin[i]) * ev.gather_stride(), ev.gather_stride(), item_ctx(id));
        else item_tree(id).finish_batch(ev.values() + lane.slot_begin[i], ev.probs() + size_t(lane.slot_begin[i]) * ev.nb_policy(), ev.nb_policy(), item_ctx(id));
    };
    // after the trees of `ids` have collected: count, submit (or finish at once when there is nothing to evaluate)
    auto submit_batch = [&](Lane& lane, const std::vector<int>& ids) {
        Evaluator& ev = *lane.eval;
        const int n_use = int(ids.size());
        int total_new = 0, last_used = 0;
        for (int i = 0; i < n_use; ++i) {
            total_new += lane.n_new[i];
            if (lane.n_new[i]) last_used = lane.slot_begin[i] + lane.n_new[i];
        }
        lane.batch_ids = ids;   // which tree owns which slot range, for the apply step
        lane.in_flight = false;
        if (total_new == 0) {
            // nothing to evaluate (all terminal / collisions): finish immediately
            workers.parallel_for(n_use, [&](int i) { item_tree(ids[i]).finish_batch(nullptr, nullptr, ev.nb_policy(), item_ctx(ids[i])); });
            return;
        }
        const auto s0 = std::chrono::steady_clock::now();
        lane.gathered = ev.gather_stride() > 0 && !gather_overflow.load(std::memory_order_relaxed);
        if (lane.gathered) ev.submit_gathered(last_used, layout_);
        else ev.submit(last_used, layout_);
        t_submit += std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
        lane.in_flight = true;
        st.nn_evals += total_new;
        ++st.batches;
    };
    auto note_items = [&](const std::vector<double>& item_t) {
        double mx = 0, sm = 0;
        for (double v : item_t) { mx = std::max(mx, v); sm += v; }
        t_item_max += mx;
        t_item_sum += sm;
    };
    auto collect_lane = [&](Lane& lane) -> bool {
        std::vector<int> active;
        for (int id : lane.trees)
            if (!done(id)) active.push_back(id);
        if (active.empty()) return false;
        Evaluator& ev = *lane.eval;
        const int B = ev.batch_size();
        // fixed per-tree quota (= the reference's per-search Batch_Size): a tree's statistics do not depend on which other
        // trees are still running.  More trees than slots: the rest waits a round (rotation below).
        int quota = std::max(1, B / int(lane.trees.size()));
        // adaptive (set_adaptive_quota): the trees still running share the whole batch
        if (adaptive_cap_ > 0 && shared_k_ == 0) quota = std::max(quota, std::min(adaptive_cap_, B / int(active.size())));
        const int n_use = std::min<int>(int(active.size()), B / quota);
        lane.slot_begin.assign(n_use, 0);
        lane.slot_count.assign(n_use, 0);
        lane.n_new.assign(n_use, 0);
        std::vector<int> ids(active.begin(), active.begin() + n_use);
        for (int i = 0; i < n_use; ++i) { lane.slot_begin[i] = i * quota; lane.slot_count[i] = quota; }
        const auto c0 = std::chrono::steady_clock::now();
        std::vector<double> item_t(timing ? n_use : 0);
        gather_overflow.store(false, std::memory_order_relaxed);
        workers.parallel_for(n_use, [&](int i) {
            const auto i0 = std::chrono::steady_clock::now();
            collect_item(lane, i, ids[i]);
            if (timing) item_t[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - i0).count();
        });
        if (timing) note_items(item_t);
        t_par += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
        // rotate so that waiting trees get their turn next round
        const bool rotated = int(active.size()) > n_use;
        if (rotated) std::rotate(lane.trees.begin(), lane.trees.begin() + 1, lane.trees.end());
        lane.same_trees_next = !rotated;       // the next batch of this lane can keep trees and slots (fused step below)
        submit_batch(lane, ids);
        return true;
    };
    auto apply_lane = [&](Lane& lane) {
        Evaluator& ev = *lane.eval;
        const auto w0 = std::chrono::steady_clock::now();
        ev.wait();
        t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        const int n_use = int(lane.slot_begin.size());
        workers.parallel_for(n_use, [&](int i) { finish_item(lane, lane.gathered, i, lane.batch_ids[i]); });
        lane.in_flight = false;
    };
    // The usual step of a lane whose trees all fit into one batch: results of the batch in flight and the next leaf collection in
    // ONE fork/join -- every tree finishes its share and collects again at once (same thread, its nodes still in that core's
    // cache), keeping its slots.  A tree that reaches its limit leaves its slots empty; the step after that goes the long way
    // (apply_lane + collect_lane), which compacts the remaining trees to the front.
    auto fused_step = [&](Lane& lane) -> bool {
        Evaluator& ev = *lane.eval;
        const auto w0 = std::chrono::steady_clock::now();
        ev.wait();
        const auto c0 = std::chrono::steady_clock::now();
        t_wait += std::chrono::duration<double>(c0 - w0).count();
        const std::vector<int> ids = lane.batch_ids;
        const int n_use = int(ids.size());
        const bool was_gathered = lane.gathered;
        std::vector<double> item_t(timing ? n_use : 0);
        std::atomic<int> still_running{0};
        gather_overflow.store(false, std::memory_order_relaxed);
        workers.parallel_for(n_use, [&](int i) {
            const auto i0 = std::chrono::steady_clock::now();
            const int id = ids[i];
            finish_item(lane, was_gathered, i, id);
            if (done(id)) {
                lane.n_new[i] = 0;
                if (ev.gather_stride())
                    for (int k = 0; k < lane.slot_count[i]; ++k) ev.gather_cnt()[size_t(lane.slot_begin[i] + k)] = 0;
            } else {
                still_running.fetch_add(1, std::memory_order_relaxed);
                collect_item(lane, i, id);
            }
            if (timing) item_t[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - i0).count();
        });
        if (timing) note_items(item_t);
        t_par += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
        const int running = still_running.load();
        lane.same_trees_next = running == n_use;
        submit_batch(lane, ids);
        return running > 0;
    };

    // development: CRA_POOL_TIMING=1 prints where the driver thread spends its time (wait = blocked on the GPU)
    double t_apply = 0, t_collect = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    bool any = true;
    // A failure in the middle of a run (evaluator callback, device error, allocation) must not leave half-applied batches behind:
    // every lane is drained, and a tree that still holds leaves without results or unreverted virtual losses restarts from its
    // root position -- its statistics would be wrong otherwise, and Tree::apply_move would refuse it from then on.
    auto recover = [&]() {
        for (Lane& lane : lanes_) {
            if (lane.in_flight) {
                try { lane.eval->wait(); } catch (...) {}
                lane.in_flight = false;
            }
            lane.same_trees_next = false;
        }
        for (size_t i = 0; i < trees_.size(); ++i)
            if (trees_[i]->pending_new() > 0 || trees_[i]->pending_collisions() > 0) reset_position(int(i), trees_[i]->root_position());
    };
    try {
    while (any) {
        any = false;
        for (Lane& lane : lanes_) {
            const auto a0 = now();
            if (lane.in_flight && lane.same_trees_next && !two_step) {
                const double w_before = t_wait;
                if (fused_step(lane)) any = true;
                if (timing) {
                    t_apply += t_wait - w_before;
                    t_collect += std::chrono::duration<double>(now() - a0).count() - (t_wait - w_before);
                }
                continue;
            }
            if (lane.in_flight) apply_lane(lane);
            const auto a1 = now();
            if (collect_lane(lane)) any = true;
            if (timing) {
                t_apply += std::chrono::duration<double>(a1 - a0).count();
                t_collect += std::chrono::duration<double>(now() - a1).count();
            }
        }
    }
    if (timing) fprintf(stderr, "pool timing: wait %.1f ms, apply %.1f ms, collect+submit %.1f ms (parallel collect %.1f [items: sum %.1f, sum of per-batch max %.1f], submit %.1f), batches %llu\n",
                        t_wait * 1e3, (t_apply - t_wait) * 1e3, t_collect * 1e3, t_par * 1e3, t_item_sum * 1e3, t_item_max * 1e3, t_submit * 1e3,
                        (unsigned long long)st.batches);
    for (Lane& lane : lanes_)
        if (lane.in_flight) apply_lane(lane);
    } catch (...) {
        recover();
        throw;
    }
    for (size_t i = 0; i < trees_.size(); ++i)
        if (!is_paused(int(i))) trees_[i]->end_search();
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t dsum = 0;
    for (size_t i = 0; i < trees_.size(); ++i) {
        st.nodes += trees_[i]->node_count() - nodes_pre[i];
        st.simulations += trees_[i]->root_visits() - visits_pre[i];
        dsum += trees_[i]->depth_sum() - depth_pre[i];
        st.depth_max = std::max(st.depth_max, trees_[i]->depth_max());
    }
    st.depth_avg = st.simulations ? double(dsum) / double(st.simulations) : 0.0;
    if (stats) *stats = st;
}

}  // namespace search
}  // namespace cra
