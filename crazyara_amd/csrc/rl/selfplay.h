// Self-play / arena game loops on top of the search pool, native.
//
// Follows engine/src/rl/selfplay.cpp (`generate_game` :192-265, `generate_arena_game` :267-308, `go_arena` :387-424,
// `init_starting_state_from_raw_policy` :426-452, `play_move_and_update` :38-54, `check_for_resignation` :169-182,
// `adjust_node_count` :146-152), agents/agent.cpp (`set_best_move` :38-55, `apply_quantile_clipping` :121-130), util/blazeutil.h
// (`apply_temperature` :77-87, `get_quantile` :188-212) and rl/gamepgn.cpp (:28-56) -- restructured for the many-trees pool: G games
// run CONCURRENTLY, every game owns one tree slot, one `run` of the pool searches the next move of all of them in shared GPU
// batches, then every game picks and plays its move (the searched subtree is kept: Tree::apply_move) and finished games are
// replaced.  The raw-policy opening plies of the games that start in the same round are evaluated together through the pool's
// first lane (the root priors of a freshly reset tree ARE RawNetAgent's policy over the legal moves).
//
// Where the reference draws from rand() / std::random_device (opening plies, temperature sampling, resignation, node-count
// jitter) a game draws from its own seeded std::mt19937_64, so a run replays; the distributions are the reference's.
#pragma once
#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../chess/position.h"
#include "../search/pool.h"
#include "traindata.h"

namespace cra {
namespace rl {

struct SelfPlaySettings {
    uint32_t simulations = 800;             // search budget per move (SearchLimits::simulations; `nodes` works the same way)
    uint32_t nodes = 0;
    double node_random_factor = 0.0;        // RLSettings::nodeRandomFactor
    double mean_init_ply = 0.0;             // PlaySettings::meanInitPly
    int max_init_ply = 30;                  // PlaySettings::maxInitPly
    double raw_policy_prob_temperature = 0.0;   // RLSettings::rawPolicyProbabilityTemperature
    double init_temperature = 0.0;          // PlaySettings::initTemperature
    int temperature_moves = 0;              // PlaySettings::temperatureMoves
    double temperature_decay = 1.0;         // PlaySettings::temperatureDecayFactor
    double quantile_clipping = 0.0;         // PlaySettings::quantileClipping
    double resign_probability = 0.0;        // RLSettings::resignProbability
    double resign_threshold = -0.9;         // RLSettings::resignThreshold
    bool reuse_tree = true;                 // RLSettings::reuseTreeForSelpay
    int max_plies = 600;                    // safety net: adjudicated as a draw
    uint64_t seed = 1;
    // quick searches (is_quick_search, selfplay.cpp:154-159; :213-221): with this probability a move is searched with quick_search_nodes
    // nodes, its own Q-value weight and Dirichlet epsilon, and its position is NOT exported (:224); below 0.01 = never
    double quick_search_probability = 0.0;  // RLSettings::quickSearchProbability (Centi_Quick_Probability)
    uint32_t quick_search_nodes = 100;      // RLSettings::quickSearchNodes (Quick_Nodes)
    double quick_search_q_value_weight = 0.7;   // RLSettings::quickSearchQValueWeight (Centi_Quick_Q_Value_Weight)
    double quick_dirichlet_epsilon = 0.0;   // RLSettings::quickDirichletEpsilon (Centi_Quick_Dirichlet_Epsilon)
    // sharpen_distribution on the EXPORTED policy (blazeutil.h:94-105, selfplay.cpp:229-231): entries below the threshold are zeroed and
    // the rest renormalised, unless the largest entry is itself below it; 0 = off (Milli_Policy_Clip_Thresh)
    double low_policy_clip_threshold = 0.0;
    // game phases (MCTSAgent::get_num_phases, SearchSettings::gamePhaseDefinition): with more than one phase a sample goes to the
    // exporter of its position's phase (selfplay.cpp:232-238); definition 0 = lichess (1 or 3 phases), 1 = movecount
    int num_phases = 1;
    int game_phase_definition = 0;
};

struct GameRecord {
    std::string start_fen;
    std::vector<std::string> san, uci;
    int book_plies = 0;
    int result = 0;                         // +1 white win, -1 black win, 0 draw
    std::string termination;                // "terminal" | "resignation" | "ply limit"
    bool contender_white = true;            // arena games only
};

struct LoopStats {
    uint64_t moves = 0, nodes = 0, nn_evals = 0, kept_subtrees = 0, restarts = 0, samples = 0;
    uint64_t quick_searches = 0;            // moves searched in quick mode (their positions were not exported)
    uint64_t samples_dropped = 0;           // searched positions that found their export file full (generatedSamples >= max_samples_per_iteration)
    double seconds = 0;
    double run_seconds = 0, move_seconds = 0;      // inside SearchPool::run / inside the parallel move step
};

// blazeutil.h / agent.cpp helpers on double vectors (exposed for the tests)
void apply_temperature(std::vector<double>& p, double t);
double get_quantile(const std::vector<double>& p, double quantile);
void apply_quantile_clipping(double quantile, std::vector<double>& p);
void sharpen_distribution(std::vector<double>& p, double thresh);          // blazeutil.h:94-105

// RLSettings::epdFilePath / UCI option EPD_File_Path (optionsuci.cpp:206; load_random_fen, rl/selfplay.cpp:58-80): the lines of an EPD
// file, one start position each.  "" and "<empty>" mean "no file" (an empty list); a file that cannot be read throws.  Empty lines are
// skipped; pick_epd_line() draws one line uniformly with the caller's generator (the reference's reservoir draw over the lines seeds a
// fresh random_device per call; here the game's own seeded generator, so a run replays) and drops ONE trailing ';' as the reference does.
std::vector<std::string> read_epd_file(const std::string& path);
template <typename Rng> std::string pick_epd_line(const std::vector<std::string>& lines, Rng& rng) {
    if (lines.empty()) return std::string();
    std::string r = lines[size_t(std::uniform_int_distribution<size_t>(0, lines.size() - 1)(rng))];
    if (!r.empty() && r.back() == ';') r.pop_back();
    return r;
}

class SelfPlayDriver {
public:
    // takes over an EMPTY pool: adds one tree slot per concurrent game.  exporter may be null.
    SelfPlayDriver(search::SearchPool* pool, const SelfPlaySettings& s, int concurrent, chess::Variant variant, bool is960,
                   TrainDataExporter* exporter);
    // game i starts from start_fens[i % size] ("" = the variant's start position); default: always the start position
    void set_start_fens(std::vector<std::string> fens) { start_fens_ = std::move(fens); }
    // epdFilePath: every game starts from a random line of the file (drawn with the game's generator); overrides set_start_fens
    void set_epd_file(const std::string& path) { epd_lines_ = read_epd_file(path); }
    // one more exporter per further game phase (phase 0 = the constructor's): SelfPlay's `exporters` (selfplay.cpp:115-125)
    void set_phase_exporter(int phase, TrainDataExporter* exporter);
    // n_games > 0: plays until n_games are finished in total (over all calls) -- SelfPlay::go(N): every game is played out, positions
    // beyond the export file's capacity are searched and dropped (stats().samples_dropped);  n_games == 0: SelfPlay::go(0)
    // (selfplay.cpp:374-377), games are started until the phase-0 export file is full, the running ones are played out.
    // Returns the total of finished games.
    size_t play(size_t n_games, int threads);
    const std::vector<GameRecord>& finished() const { return finished_; }
    const LoopStats& stats() const { return stats_; }

private:
    struct Game {
        int slot = 0;
        GameRecord rec;
        chess::Position pos;
        std::mt19937_64 rng;
        bool allow_resign = false;
        bool in_opening = false;
        int opening_left = 0;
        bool quick = false;                 // this move's search is a quick one
        struct Sample { chess::Position pos; std::vector<chess::Move> moves; std::vector<double> policy; float q; int phase; };
        std::vector<Sample> samples;
    };
    void start_games(size_t n_games);
    bool check_over(Game& g, const std::string* san, const std::string* uci);
    void finish(Game& g, int result, const char* why);
    search::SearchPool* pool_;
    SelfPlaySettings s_;
    int concurrent_;
    chess::Variant variant_;
    bool is960_;
    TrainDataExporter* exporter_;           // phase 0
    std::vector<TrainDataExporter*> exporters_;      // by phase; [0] == exporter_
    size_t samples_taken_ = 0;              // generatedSamples: positions accepted for export so far (buffered in their games or written)
    size_t sample_capacity() const;         // max_samples_per_iteration() = the export file's capacity
    std::vector<std::string> start_fens_, epd_lines_;
    std::vector<std::unique_ptr<Game>> games_;
    std::vector<GameRecord> finished_;
    size_t started_ = 0;
    LoopStats stats_;
};

// go_arena: two pools (two nets); game 2i has the contender (pool A) as White from a fresh start position, game 2i+1 replays the SAME
// start position with colours swapped; the player to move searches (the other pool's tree of that game pauses), both apply the move;
// always the best move, no resignation.
class ArenaDriver {
public:
    ArenaDriver(search::SearchPool* pool_a, search::SearchPool* pool_b, const SelfPlaySettings& s, int concurrent, chess::Variant variant,
                bool is960);
    void set_start_fens(std::vector<std::string> fens) { start_fens_ = std::move(fens); }   // pair i uses start_fens[i % size]
    // epdFilePath (go_arena, rl/selfplay.cpp:396): every PAIR of games starts from a random line of the file; overrides set_start_fens
    void set_epd_file(const std::string& path) { epd_lines_ = read_epd_file(path); }
    size_t play(size_t n_games, int threads);
    const std::vector<GameRecord>& finished() const { return finished_; }
    const LoopStats& stats() const { return stats_; }
    int wins() const { return wins_; }
    int draws() const { return draws_; }
    int losses() const { return losses_; }

private:
    struct Game {
        size_t idx = 0;
        GameRecord rec;
        chess::Position pos;
        bool active = false;
    };
    search::SearchPool* pools_[2];
    SelfPlaySettings s_;
    int concurrent_;
    chess::Variant variant_;
    bool is960_;
    std::vector<std::string> start_fens_, epd_lines_;
    std::vector<Game> games_;
    std::vector<std::string> pair_fen_;
    std::vector<GameRecord> finished_;
    size_t started_ = 0;
    int wins_ = 0, draws_ = 0, losses_ = 0;
    LoopStats stats_;
};

// GamePGN's operator<< (gamepgn.cpp:28-56)
std::string game_pgn(const GameRecord& g, const std::string& variant_tag, const std::string& event, const std::string& white,
                     const std::string& black, const std::string& date);

}  // namespace rl
}  // namespace cra
