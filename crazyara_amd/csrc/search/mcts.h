// Host-side MCTS leaf collection: select / expand / virtual loss / backup, restating the arithmetic of
// engine/src/node.{h,cpp}, nodedata.{h,cpp} and engine/src/searchthread.cpp.
//
// MI355X-first structure (not the reference's): instead of >=2 SearchThreads sharing one tree to hide a blocking
// predict(), a SearchPool keeps MANY independent trees (games / opening positions) and fills each GPU batch with
// leaves from all of them; trees are split into two pipeline halves so that one half's leaves are collected while the
// other half's batch is on the GPU (side stream, pinned async copies, 192-byte descriptors instead of float planes).
// Every tree is touched by exactly one thread at a time -> no per-node mutex, no global hash-table mutex.
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "../chess/planes.h"
#include "../chess/policy.h"
#include "../chess/position.h"

namespace cra {
namespace search {

enum VirtualStyle : int { VIRTUAL_LOSS = 0, VIRTUAL_VISIT = 1, VIRTUAL_OFFSET = 2, VIRTUAL_MIX = 3 };   // node.h:78-95
enum NodeType : int8_t { NT_WIN = 0, NT_DRAW = 1, NT_LOSS = 2, NT_UNSOLVED = 6 };

// engine/src/agents/config/searchsettings.{h,cpp}; defaults = the UCI defaults a `go` sees (optionsuci.cpp:66-219)
struct SearchSettings {
    int batch_size = 16;                  // Batch_Size
    float cpuct_init = 2.5f;              // Centi_CPuct_Init 250
    float cpuct_base = 19652.0f;          // CPuct_Base
    float node_policy_temperature = 1.7f; // Centi_Node_Temperature 170 (RL builds: 100)
    int virtual_style = VIRTUAL_MIX;      // MCTS_Virtual_Style (optionsuci.cpp:194-195)
    uint32_t virtual_mix_threshold = 1000;
    double virtual_offset_strength = 0.001;
    float q_value_weight = 1.0f;          // Centi_Q_Value_Weight 100
    float q_veto_delta = 0.4f;            // Centi_Q_Veto_Delta 40
    int mode = MODE_CRAZYHOUSE;           // build flavour: label set + plane layout family
    int version_major = 1;                // input representation version of the loaded net
    int version_minor = 0;                //   (chess 2.7 / 2.8 differ in the minor number)
    bool is_policy_map = true;
    // Board::operator= copies lastMoves only in MODE_CHESS / MODE_LICHESS binaries (board.cpp:106-108): in a crazyhouse
    // binary every leaf's move history restarts at the root clone.  -1 = follow the mode, 0/1 = force.
    int clone_keeps_last_moves = -1;
    // epsilon exploration (searchthread.cpp:124-185, 451-473, 497-501): one simulation in `counter` starts with a random
    // playout (or an unexplored checking move) from a random depth of the principal variation.  0 = off.  The reference draws
    // from rand() (seeded with the time: not reproducible); here every tree owns a seeded ANSI-C LCG, so searches replay.
    // UCI defaults: Centi_Epsilon_Greedy 5 -> 20, Centi_Epsilon_Checks 1 -> 100 (optionsuci.cpp:89-90, crazyara.cpp:748-749).
    int epsilon_greedy_counter = 0;
    int epsilon_checks_counter = 0;
    uint32_t seed = 1;
    // MCTS_Solver (optionsuci.cpp:129, default true): terminal backups mark parents WIN / LOSS / DRAW once their children
    // prove it (Node::solve_for_terminal, node.cpp:365-453); a solved root ends the search (searchthread.cpp:333-340).
    // Search_Type "mcgs" needs no switch here: in the reference the transposition link of add_new_node_to_tree is
    // unreachable (node.cpp:730-731 reads the candidate from the still-empty child slot), so mcgs and mcts search the same tree --
    // pinned on the compiled reference run under both values of useMCGS (tests/test_mcts_reference_build.py::
    // test_mcgs_flag_searches_the_same_tree: identical dumps on transposition-rich positions, equal to this tree).
    bool mcts_solver = true;
    // Dirichlet noise on the root priors (mctsagent.cpp:311-316, node.cpp:950-954, blazeutil.h:113-124): applied at the start of
    // every search when epsilon > 0.009, followed by fully_expand_node.  UCI defaults: Centi_Dirichlet_Epsilon 0 (25 in RL builds),
    // Centi_Dirichlet_Alpha 20 (optionsuci.cpp:84-88).  The reference draws from a std::default_random_engine seeded by
    // std::random_device; here each tree owns one seeded with `seed` + tree index.
    float dirichlet_epsilon = 0.0f;
    float dirichlet_alpha = 0.2f;
};

struct Node {
    std::vector<chess::Move> actions;     // legalActions (sorted by prior on first selection)
    std::vector<float> priors;            // policyProbSmall
    std::vector<uint16_t> policy_idx;     // MV_LOOKUP index per action (consumed when the NN result arrives)
    // NodeData (nodedata.h:88-121): entries [0, no_visit_idx]
    std::vector<uint32_t> child_visits;
    std::vector<float> q;
    std::vector<int32_t> child;           // node index or -1
    std::vector<uint8_t> vl;              // virtualLossCounter
    std::vector<int8_t> child_types;      // nodeTypes: what the solver has recorded about each expanded child
    double value_sum = 0.0;
    uint32_t real_visits = 0;
    uint32_t visit_sum = 0;
    uint32_t free_visits = 0;
    uint16_t no_visit_idx = 0;
    uint64_t key = 0;                     // hash key of the position (Node::key, node.cpp:84): lets the descent skip its recomputation
    uint16_t plies = 0;
    uint16_t unsolved_children = 0;       // numberUnsolvedChildNodes
    uint16_t end_in_ply = 0;              // endInPly: distance to the proven terminal
    int32_t checkmate_idx = -1;           // checkmateIdx (NO_CHECKMATE)
    int8_t node_type = NT_UNSOLVED;
    bool terminal = false, has_nn = false, sorted = false, has_data = false, inspected = false;
    uint8_t stm = 0;
    uint8_t lock = 0;                     // per-node spin lock (Node::mtx, node.h:100): taken only by trees with several collectors
    // MCTS_STORE_STATES (node.h:111,530; searchthread.cpp:198-213): the position this node stands for, kept from its creation so that a
    // later expansion below it starts HERE instead of cloning the root and replaying the whole path.  Optional per node (a tree over its
    // state budget, or a kept subtree whose move history no longer starts at the root, simply replays from the nearest ancestor that
    // has one); written before the node is linked into its parent, read-only afterwards.
    std::unique_ptr<chess::Position> state;

    float value() const { return float(value_sum / real_visits); }                         // node.cpp:595-598
    void set_value(float v) { ++real_visits; value_sum = double(v * float(real_visits)); } // node.cpp:716-720
    uint32_t real_child_visits(int i) const { return child_visits[i] - vl[i]; }            // node.cpp:650-653
};

struct NodeAndIdx {
    int32_t node;
    uint16_t child_idx;
};
typedef std::vector<NodeAndIdx> Trajectory;

enum NodeBackup { NODE_COLLISION, NODE_TERMINAL, NODE_NEW_NODE, NODE_TRANSPOSITION };

float get_current_cput(float visits, const SearchSettings& s);                             // node.cpp:1243-1246
VirtualStyle get_virtual_style(const SearchSettings& s, uint32_t visits);                  // node.h:87-95

// What one SearchThread of the reference owns (searchthread.h:52-80): the trajectories of its mini-batch in flight, its running
// position and scratch buffers, its exploration stream.  A tree searched by ONE collector (the pool's many-trees mode) has exactly one;
// a tree shared by k collectors -- k SearchThreads on one tree, crazyara.cpp:555-561 -- has k, and takes the per-node locks.
struct Collector {
    Trajectory trajectory_buffer;
    std::vector<int32_t> new_nodes;
    std::vector<Trajectory> new_trajectories, collision_trajectories;
    chess::Position scratch_pos;          // the simulation's running position (assigned from the root: keeps its buffers)
    std::vector<std::pair<chess::Move, const chess::Key*>> replay;   // moves since the last stored state on the way down (Tree::LazyPos)
    std::vector<int> sort_perm;
    std::vector<chess::Move> sort_moves;
    std::vector<float> sort_priors;
    std::vector<float> select_buf;
    uint32_t rng = 1;
    uint64_t depth_sum = 0;
    uint32_t depth_max = 0;
};

// Node storage with stable addresses and lock-free indexing: chunks of 1024 nodes behind a fixed table of chunk pointers, so that
// collectors can allocate nodes while others walk the tree (a std::vector would move every node on growth).
class NodeArena {
public:
    // 256 nodes per chunk, room for 33.5 M nodes.  The chunk-pointer table (1 MiB) is an anonymous mapping: fresh zero pages that the OS
    // hands over only when they are touched, so a tree costs its first chunk (64 KiB of nodes) and one page of table -- not a 256 KiB memset
    // plus 1024 constructed nodes per tree, which a game loop paid on every played move of every game (Tree::apply_move, reset_position)
    static constexpr int kChunkBits = 8, kChunk = 1 << kChunkBits, kMaxChunks = 1 << 17;
    NodeArena();
    ~NodeArena();
    NodeArena(const NodeArena&) = delete;
    NodeArena& operator=(const NodeArena&) = delete;
    Node& operator[](int i) { return table_[size_t(i) >> kChunkBits].load(std::memory_order_acquire)[i & (kChunk - 1)]; }
    const Node& operator[](int i) const { return table_[size_t(i) >> kChunkBits].load(std::memory_order_acquire)[i & (kChunk - 1)]; }
    int emplace_back();                   // thread-safe; the node is default-constructed
    size_t size() const { return size_.load(std::memory_order_acquire); }
    void clear();                         // single-threaded phases only
    void swap(NodeArena& o);              // single-threaded phases only
private:
    std::atomic<Node*>* table_;           // mmap'ed zero pages: all-zero bytes are null pointers
    std::atomic<uint32_t> size_{0};
    std::mutex grow_;
};

class Tree {
public:
    Tree(const chess::Position& root, const SearchSettings& settings);

    // --- root (MCTSAgent::create_new_root_node / set_root_node_predictions, mctsagent.cpp:166-196) ---
    bool root_needs_eval() const { return !nodes_[0].has_nn && !nodes_[0].terminal; }
    void root_desc(BoardDesc& d) const;
    void set_root_result(float value, const float* probs);
    // start of a `go` (MCTSAgent::evaluate_board_state, mctsagent.cpp:311-316): Dirichlet noise + full expansion of the root
    void begin_search();
    // A move was played on the board (MCTSAgent::apply_move_to_tree + get_root_node_from_tree, mctsagent.cpp:230-247,130-164):
    // the subtree below that move becomes the tree if the child is a playout node with visits, otherwise the tree restarts
    // from the new position.  Returns true when the subtree was kept.  The move must be legal in the root position.
    bool apply_move(chess::Move m);
    // MCTSAgent::evaluate_board_state around the search proper (mctsagent.cpp:292-342): a root with ONE legal move is not searched
    // ("Only single move available -> early stopping"): handle_single_move (:277-290) gives it the value of the previous search
    // (lastValueEval, from the side that is now to move) instead; end_search records lastValueEval = bestMoveQ[0] and the side.
    bool single_move_root() const { return nodes_[0].actions.size() == 1 && !nodes_[0].terminal; }
    void handle_single_move();
    void end_search();
    // replace the search settings for the following searches (quick-search switches of self-play, selfplay.cpp:217-222)
    void set_search_settings(const SearchSettings& s) { s_ = s; }
    // Stored leaf states (the reference's MCTS_STORE_STATES build option): every new non-terminal node keeps its position, up to `budget`
    // nodes per tree; 0 = off, the default here as in the reference's default build: every simulation clones the root and replays its
    // path.  Same trees bit for bit either way (tests/test_mcts.py::test_stored_leaf_states_grow_the_same_trees).  MEASURED before it was
    // made a default (round 6, scripts/hostbench/replay_share_bench.cpp, profiles/r06/g_*): on ONE tree of 25,600 simulations (leaf depth
    // 6.8 on average, 26 at most) the clone of the root and the replay of the path are 5.5 % of Tree::collect = 4.5 % of the host's
    // search time (the replay is an incremental do_move with the child's key known, ~80 ticks a ply; a first reading of 22 - 25 % was
    // the tick counter around every single ply), and a stored state is a Position with three heap vectors: with states the same search
    // took 6.3 us per simulation against 5.6 without (-8 ... -13 %; level at 102,400 simulations).  Kept as an option, not the default.
    void set_state_budget(uint32_t budget) { state_budget_ = budget; }
    uint32_t stored_states() const { return stored_states_.load(std::memory_order_relaxed); }

    // --- SearchThread::create_mini_batch (searchthread.cpp:347-380) with `quota` in the role of batchSize ---
    // Writes one BoardDesc per NEW leaf to descs[0..returned).  Terminals are backed up immediately, collisions are
    // remembered and reverted in finish_batch().
    // `ctx` = which collector of the tree (0 for a tree with one): concurrent calls with DIFFERENT ctx are allowed once
    // set_collectors(k > 1) was called -- the k SearchThreads of the reference sharing one tree.
    int collect(int quota, BoardDesc* descs, int ctx = 0);
    // set_nn_results_to_child_nodes + backup_value_outputs + backup_collisions (searchthread.cpp:301-324)
    void finish_batch(const float* values, const float* probs, int nb_policy, int ctx = 0);
    // number of collectors (>= 1).  More than one switches the per-node locks on: select / virtual loss / expansion bookkeeping and the
    // backups of a node happen under that node's lock, as in the reference (searchthread.cpp:187-267, node.h:199-246); the expensive
    // part of an expansion (move generation, legality, terminal test, policy indices) runs outside any lock.
    void set_collectors(int k);
    int n_collectors() const { return int(collectors_.size()); }
    // The same with the priors already gathered (on the GPU) for the legal moves of each new node:
    // pending_policy_indices(k) = the policy indices of the k-th new node of the last collect(), in move order; `gathered + k * stride`
    // holds probs[index] for exactly those (what set_probabilities_for_moves reads, node.cpp:961-979).
    void pending_policy_indices(int k, const uint16_t** idx, int* count, int ctx = 0) const;
    void finish_batch_gathered(const float* values, const float* gathered, uint32_t stride, int ctx = 0);

    // --- queries ---
    const Node& root() const { return nodes_[0]; }
    const Node& node(int i) const { return nodes_[i]; }
    size_t node_count_allocated() const { return nodes_.size(); }
    // read by the pool's limit checks while collectors run: relaxed atomic loads of the root's counters
    uint32_t root_visits() const { return __atomic_load_n(&nodes_[0].visit_sum, __ATOMIC_RELAXED); }
    uint32_t node_count() const {                                                          // Node::get_node_count, node.cpp:1303-1306
        return __atomic_load_n(&nodes_[0].visit_sum, __ATOMIC_RELAXED) - __atomic_load_n(&nodes_[0].free_visits, __ATOMIC_RELAXED);
    }
    const chess::Position& root_position() const { return root_pos_; }
    int pending_new(int ctx = -1) const;           // new nodes without results; ctx < 0: over all collectors
    int pending_collisions(int ctx = -1) const;
    uint64_t depth_sum() const;                    // summed over the collectors
    uint32_t depth_max() const;
    void reset_depth_max();
    // Node::get_mcts_policy + argmax (node.cpp:1070-1109): best child index of the root and the visit policy
    int best_move_index(std::vector<double>* policy = nullptr) const;
    // EvalInfo::bestMoveQ of root child b (set_eval_for_single_pv + get_best_move_q, evalinfo.cpp:110-182): the negated value of
    // the CHILD NODE (its proven result if solved, else valueSum / realVisits) -- not the edge's Q; Q_INIT without a child node
    float best_move_q(int b) const;
    // EvalInfo::bestMoveQ[0] as update_eval_info leaves it (evalinfo.cpp:184-243): best_move_q of the chosen child, or the root's own
    // value for a single-move root that was not searched
    float eval_best_move_q() const;
    // EvalInfo::pv[0], movesToMate[0], centipawns[0] as update_eval_info / set_eval_for_single_pv leave them (evalinfo.cpp:120-243):
    // the best root move, then Node::get_principal_variation (node.cpp:1111-1121) below it -- mating child, else the child that delays a
    // proven loss longest, else the most-visited child, while the node has been selected at least once and is not terminal.
    // moves_to_mate: +(len + 1) / 2 if the position after the best move is a proven LOSS for the side to move there, -(len + 1) / 2
    // (C++ integer division) if a proven WIN, else 0; centipawns = value_to_centipawn(bestMoveQ) (evalinfo.cpp:103-112; 0 beside a mate)
    void principal_variation(std::vector<chess::Move>& pv, int* moves_to_mate, int* centipawns) const;
    // line idx of a Multi_PV output: update_eval_info with searchSettings->multiPV = multipv (evalinfo.cpp:195-260).  Line 0 is the one
    // above; line idx >= 1 starts with the root move of rank idx in the MCTS policy over ALL legal moves (sort_eval_lists,
    // evalinfo.cpp:184-193: descending, compared as float; equal entries keep their root order here, the reference's std::sort leaves
    // them unspecified).  Lines exist for idx < min(multipv, expanded root children); returns false beyond.  q = bestMoveQ[idx].
    bool principal_variation_multi(int idx, int multipv, std::vector<chess::Move>& pv, int* moves_to_mate, int* centipawns, float* q) const;
    // Whole tree as a flat word list (inspection / parity tests; the reference's counterpart is MCTSAgent::export_search_tree,
    // mctsagent.cpp:420-448): depth-first preorder over the expanded children, one record per node that owns NodeData:
    // [n_expanded, visit_sum, real_visits, free_visits, node_type, end_in_ply, terminal, bits(value)] then per expanded child
    // [move, visits, virtual-loss counter, bits(Q), bits(prior), state 0 no node / 1 node without data / 2 record follows]
    void dump(std::vector<uint32_t>& out) const;

    // exposed for the arithmetic parity tests
    int select_child(Node& n) { return select_child(n, *collectors_[0]); }
    int select_child(Node& n, Collector& col);
    void apply_virtual_loss(Node& n, int child_idx);
    void revert_virtual_loss(Node& n, int child_idx);
    void revert_virtual_loss_and_update(Node& n, int child_idx, float value, bool free_backup, bool solve = false);
    void backup_value(float value, const Trajectory& t, bool free_backup, bool solve = false);
    bool solve_for_terminal(Node& n, int child_idx);
    bool root_solved() const { return __atomic_load_n(&nodes_[0].node_type, __ATOMIC_RELAXED) != NT_UNSOLVED; }

private:
    int new_node(const chess::Position& pos);
    void prepare_node_for_visits(Node& n, Collector& col);
    void increment_no_visit_idx(Node& n);
    void fill_nn_result(Node& n, float value, const float* probs);
    void fill_nn_result_gathered(Node& n, float value, const float* priors);
    void finish_node(Node& n, float value);
    void backup_batch(Collector& col);
    int get_new_child_to_evaluate(Collector& col, NodeBackup& type, uint32_t& depth, BoardDesc* desc_out);
    uint32_t next_rand(Collector& col);
    size_t get_random_depth(Collector& col);
    // the simulation's position, materialised only when something needs it (an expansion, the exploration steps): the nearest stored
    // state on the way down + the moves since (mcts.cpp)
    struct LazyPos;
    int get_starting_node(Collector& col, int cur, uint32_t& depth, int& child_idx, LazyPos& lp);
    void random_playout(Collector& col, int cur, int& child_idx);
    int select_enhanced_move(int cur, const chess::Position& pos);
    int best_action_index_fast(const Node& n) const;
    void line_below_root_child(int b, std::vector<chess::Move>& pv, int* moves_to_mate, int* centipawns, float* q) const;

    SearchSettings s_;
    chess::Position root_pos_;
    const chess::PolicyTables* tables_;
    int layout_;
    bool keep_last_moves_;
    NodeArena nodes_;
    std::vector<std::unique_ptr<Collector>> collectors_;   // [0] always exists
    bool concurrent_ = false;                               // several collectors: per-node locks are taken
    uint32_t state_budget_ = 0;                             // stored leaf states per tree (set_state_budget); 0 = off
    std::atomic<uint32_t> stored_states_{0};
    float last_value_eval_ = -1.0f;            // MCTSAgent::lastValueEval (mctsagent.cpp:47), reset with the game (clear_game_history)
    uint8_t last_stm_ = 0;                     // MCTSAgent::lastSideToMove
    std::minstd_rand0 noise_rng_;              // std::default_random_engine of libstdc++ (randomgen.h:35)
};

}  // namespace search
}  // namespace cra
