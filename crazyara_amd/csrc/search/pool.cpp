#include "pool.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <stdexcept>

#include "../nn/rise_net.h"

namespace cra {
namespace search {

// ---------------------------------------------------------------------------------------------------------------------
// evaluators
// ---------------------------------------------------------------------------------------------------------------------
namespace {
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
    }
    ~HipEvaluator() override {
        (void)hipHostFree(descs_);
        (void)hipHostFree(values_);
        (void)hipHostFree(probs_);
        if (aux_) (void)hipHostFree(aux_);
    }
    int batch_size() const override { return batch_; }
    int nb_policy() const override { return nb_policy_; }
    BoardDesc* descs() override { return descs_; }
    const float* values() override { return values_; }
    const float* probs() override { return probs_; }
    void submit(int n_valid, int layout) override { net_->submit_boards(descs_, n_valid, layout, values_, probs_, aux_); }
    void wait() override { net_->wait(); }

private:
    RiseNet* net_;
    int batch_ = 0, nb_policy_ = 0;
    BoardDesc* descs_ = nullptr;
    float *values_ = nullptr, *probs_ = nullptr, *aux_ = nullptr;
};

class CallbackEvaluator : public Evaluator {
public:
    CallbackEvaluator(EvalFn fn, void* user, int batch, int nb_policy)
        : fn_(fn), user_(user), batch_(batch), nb_policy_(nb_policy), descs_(batch), values_(batch), probs_(size_t(batch) * nb_policy) {
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

private:
    EvalFn fn_;
    void* user_;
    int batch_, nb_policy_;
    std::vector<BoardDesc> descs_;
    std::vector<float> values_, probs_;
};
}  // namespace

std::unique_ptr<Evaluator> make_hip_evaluator(RiseNet* net) { return std::unique_ptr<Evaluator>(new HipEvaluator(net)); }
std::unique_ptr<Evaluator> make_callback_evaluator(EvalFn fn, void* user, int batch, int nb_policy) {
    return std::unique_ptr<Evaluator>(new CallbackEvaluator(fn, user, batch, nb_policy));
}

// ---------------------------------------------------------------------------------------------------------------------
// worker pool
// ---------------------------------------------------------------------------------------------------------------------
WorkerPool::WorkerPool(int threads) {
    for (int i = 1; i < threads; ++i) workers_.emplace_back([this] { worker_loop(); });
}
WorkerPool::~WorkerPool() {
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
}
void WorkerPool::worker_loop() {
    int seen = 0;
    while (true) {
        const std::function<void(int)>* fn;
        int n;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            fn = fn_;
            n = n_;
        }
        for (int i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
        {
            std::lock_guard<std::mutex> lk(m_);
            if (--active_ == 0) done_cv_.notify_all();
        }
    }
}
void WorkerPool::parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (workers_.empty() || n == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(m_);
        fn_ = &fn;
        n_ = n;
        next_.store(0);
        active_ = int(workers_.size());
        ++generation_;
    }
    cv_.notify_all();
    for (int i; (i = next_.fetch_add(1)) < n;) fn(i);
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return active_ == 0; });
}

// ---------------------------------------------------------------------------------------------------------------------
// pool
// ---------------------------------------------------------------------------------------------------------------------
SearchPool::SearchPool(const SearchSettings& s, std::unique_ptr<Evaluator> lane_a, std::unique_ptr<Evaluator> lane_b) : s_(s) {
    layout_ = layout_for(s.mode, s.version_major);
    if (!lane_a) throw std::invalid_argument("SearchPool needs at least one evaluator lane");
    lanes_.emplace_back();
    lanes_.back().eval = std::move(lane_a);
    if (lane_b) {
        lanes_.emplace_back();
        lanes_.back().eval = std::move(lane_b);
    }
}

int SearchPool::add_position(const chess::Position& pos) {
    SearchSettings st = s_;
    st.seed = s_.seed + uint32_t(trees_.size());     // every tree owns its exploration stream
    trees_.emplace_back(new Tree(pos, st));
    const int id = int(trees_.size()) - 1;
    lanes_[id % lanes_.size()].trees.push_back(id);
    return id;
}

bool SearchPool::tree_done(const Tree& t, uint32_t simulations, uint32_t nodes) const {
    if (t.root().terminal || t.root_solved()) return true;
    if (simulations && t.root_visits() >= simulations) return true;
    if (nodes && t.node_count() >= nodes) return true;
    return false;
}

void SearchPool::evaluate_roots(Lane& lane) {
    Evaluator& ev = *lane.eval;
    std::vector<int> todo;
    for (int id : lane.trees)
        if (trees_[id]->root_needs_eval()) todo.push_back(id);
    for (size_t off = 0; off < todo.size(); off += ev.batch_size()) {
        const int n = int(std::min(todo.size() - off, size_t(ev.batch_size())));
        for (int i = 0; i < n; ++i) trees_[todo[off + i]]->root_desc(ev.descs()[i]);
        ev.submit(n, layout_);
        ev.wait();
        for (int i = 0; i < n; ++i) trees_[todo[off + i]]->set_root_result(ev.values()[i], ev.probs() + size_t(i) * ev.nb_policy());
    }
}

void SearchPool::run(uint32_t simulations, uint32_t nodes, int threads, SearchStats* stats) {
    if (!simulations && !nodes) throw std::invalid_argument("run needs a simulations or a nodes limit");
    WorkerPool workers(std::max(1, threads));
    SearchStats st;
    std::vector<uint32_t> nodes_pre(trees_.size()), visits_pre(trees_.size());
    const auto t0 = std::chrono::steady_clock::now();
    for (Lane& lane : lanes_) {
        size_t before = 0;
        for (int id : lane.trees) before += trees_[id]->root_needs_eval();
        evaluate_roots(lane);
        st.nn_evals += before;
        st.batches += (before + lane.eval->batch_size() - 1) / lane.eval->batch_size();
    }
    for (size_t i = 0; i < trees_.size(); ++i) {
        trees_[i]->begin_search();                                  // Dirichlet noise + full expansion of the root (RL settings)
        nodes_pre[i] = trees_[i]->node_count();
        visits_pre[i] = trees_[i]->root_visits();
    }
    // simulations/nodes limits are per `go`: measured from the pre-search counters (tree reuse keeps old visits)
    auto done = [&](int id) {
        const Tree& t = *trees_[id];
        if (t.root().terminal || t.root_solved()) return true;      // is_root_node_unsolved(), searchthread.cpp:333-340
        if (simulations && t.root_visits() - visits_pre[id] >= simulations) return true;
        if (nodes && t.node_count() - nodes_pre[id] >= nodes) return true;
        return false;
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
        const int quota = std::max(1, B / int(lane.trees.size()));
        const int n_use = std::min<int>(int(active.size()), B / quota);
        lane.slot_begin.assign(n_use, 0);
        lane.slot_count.assign(n_use, 0);
        lane.n_new.assign(n_use, 0);
        std::vector<int> ids(active.begin(), active.begin() + n_use);
        for (int i = 0; i < n_use; ++i) { lane.slot_begin[i] = i * quota; lane.slot_count[i] = quota; }
        workers.parallel_for(n_use, [&](int i) {
            lane.n_new[i] = trees_[ids[i]]->collect(lane.slot_count[i], ev.descs() + lane.slot_begin[i]);
        });
        // rotate so that waiting trees get their turn next round
        if (int(active.size()) > n_use) std::rotate(lane.trees.begin(), lane.trees.begin() + 1, lane.trees.end());
        int total_new = 0, last_used = 0;
        for (int i = 0; i < n_use; ++i) {
            total_new += lane.n_new[i];
            if (lane.n_new[i]) last_used = lane.slot_begin[i] + lane.n_new[i];
        }
        lane.batch_ids = ids;   // which tree owns which slot range, for the apply step
        if (total_new == 0) {
            // nothing to evaluate (all terminal / collisions): finish immediately
            workers.parallel_for(n_use, [&](int i) { trees_[ids[i]]->finish_batch(nullptr, nullptr, ev.nb_policy()); });
            return true;
        }
        ev.submit(last_used, layout_);
        lane.in_flight = true;
        st.nn_evals += total_new;
        ++st.batches;
        return true;
    };
    auto apply_lane = [&](Lane& lane) {
        Evaluator& ev = *lane.eval;
        ev.wait();
        const int n_use = int(lane.slot_begin.size());
        workers.parallel_for(n_use, [&](int i) {
            const int id = lane.batch_ids[i];
            trees_[id]->finish_batch(ev.values() + lane.slot_begin[i], ev.probs() + size_t(lane.slot_begin[i]) * ev.nb_policy(), ev.nb_policy());
        });
        lane.in_flight = false;
    };

    bool any = true;
    while (any) {
        any = false;
        for (Lane& lane : lanes_) {
            if (lane.in_flight) apply_lane(lane);
            if (collect_lane(lane)) any = true;
        }
    }
    for (Lane& lane : lanes_)
        if (lane.in_flight) apply_lane(lane);
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t dsum = 0;
    for (size_t i = 0; i < trees_.size(); ++i) {
        st.nodes += trees_[i]->node_count() - nodes_pre[i];
        st.simulations += trees_[i]->root_visits() - visits_pre[i];
        dsum += trees_[i]->depth_sum;
        st.depth_max = std::max(st.depth_max, trees_[i]->depth_max);
    }
    st.depth_avg = st.simulations ? double(dsum) / double(st.simulations) : 0.0;
    if (stats) *stats = st;
}

}  // namespace search
}  // namespace cra
