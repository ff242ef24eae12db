// ORACLE BUILD SUPPORT (test infrastructure only).
//
// The reference selects its environment at compile time in engine/src/stateobj.h:39-68; the MODE_POMMERMAN branch includes
// "pommermanstate.h" and expects `PommermanState` / `StateConstantsPommerman` from OUTSIDE the reference tree (no such file
// exists in engine/src).  oracle/ref/build_ref.py uses exactly that hook to compile the reference's own search sources
// (node.cpp, nodedata.cpp, searchthread.cpp, agents/mctsagent.cpp, evalinfo.cpp, nn/neuralnetapi.cpp ...) where they lie,
// with this header supplying the environment: the chess family through this repository's bitboard Position
// (crazyara_amd/csrc/chess/), because the reference's own environment needs the un-vendored Stockfish fork.
//
// What is the reference's and what is ours in the resulting library:
//   reference (compiled from /root/reference, unmodified): Node, NodeData, SearchThread, MCTSAgent, Agent, ThreadManager,
//       TimeManager, tree manager, EvalInfo, blazeutil, NeuralNetAPI / NeuralNetAPIUser / NeuralNetDesign, State, SearchSettings
//   ours: this State adapter (mirrors BoardState, boardstate.cpp:42-277, call by call), the blaze stand-in (shim/blaze/Math.h)
//
// Action = this repository's Move integer, so move lists of both sides compare directly.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include <cstring>

#include "state.h"

#include "chess/planes.h"
#include "chess/planes_host.h"
#include "chess/policy.h"
#include "chess/position.h"

namespace refshim {

// Build-flavour constants of the reference (MODE_CRAZYHOUSE / MODE_CHESS / MODE_LICHESS, VERSION) made run-time values so that
// one library serves every configuration the tests compare.
struct Config {
    int mode = 0;                 // 0 crazyhouse, 1 chess, 2 lichess (mi_search_settings.mode)
    int version_major = 1, version_minor = 0;
    int layout = 0;               // plane layout id of crazyara_amd/csrc/chess/planes.h
    bool is_policy_map = true;
    bool is960 = false;
    // Board::operator= copies lastMoves only in MODE_CHESS / MODE_LICHESS binaries (board.cpp:106-108)
    bool clone_keeps_last_moves = false;
};
Config& config();

// side channel for evaluators that key on the position rather than on the float planes: every get_state_planes() call appends
// the 192-byte descriptor of the position it was asked to encode (same order as the planes in the batch)
std::vector<cra::BoardDesc>& pending_descs();

}  // namespace refshim


class StateConstantsPommerman : public StateConstantsInterface<StateConstantsPommerman>
{
public:
    static uint BOARD_WIDTH() { return 8; }
    static uint BOARD_HEIGHT() { return 8; }
    static uint NB_CHANNELS_TOTAL() { return uint(cra::layout_channels(refshim::config().layout)); }
    static uint NB_LABELS() { return uint(cra::chess::policy_tables(refshim::config().mode).nb_labels()); }
    static uint NB_LABELS_POLICY_MAP() { return uint(cra::chess::policy_tables(refshim::config().mode).nb_policy_map()); }
    static uint NB_AUXILIARY_OUTPUTS() { return 0U; }                                  // boardstate.h:64-66
    static uint NB_PLAYERS() { return 2; }
    static std::string action_to_uci(Action action, bool is960);
    // OutputRepresentation::MV_LOOKUP* (outputrepresentation.cpp:39-56): a pure function of the move; the 960 flag fixed by init()
    template<PolicyType p, MirrorType m>
    static MoveIdx action_to_index(Action action) {
        return lookup(action, m == mirrored, p == normal && refshim::config().is_policy_map);
    }
    static void init(bool isPolicyMap, bool is960) {
        refshim::config().is_policy_map = isPolicyMap;
        refshim::config().is960 = is960;
    }
    static std::vector<std::string> available_variants() {
        return {"chess", "crazyhouse", "kingofthehill", "3check", "antichess", "atomic", "horde", "racingkings"};   // = cra::chess::Variant
    }
    static std::string start_fen(int variant) { return cra::chess::start_fen(cra::chess::Variant(variant)); }

private:
    static MoveIdx lookup(Action action, bool mirror, bool policy_map);
};


class PommermanState : public State
{
    cra::chess::Position pos;

public:
    PommermanState() = default;
    PommermanState(const PommermanState& o) : State(), pos(o.pos) {
        if (!refshim::config().clone_keeps_last_moves) pos.clear_last_moves();            // board.cpp:76-110
    }
    const cra::chess::Position& position() const { return pos; }

    // BoardState::mirror_policy -> flip_board (inputrepresentation.h:58-66): side != WHITE except racing kings
    bool mirror_policy(SideToMove sideToMove) const {
        return sideToMove != FIRST_PLAYER_IDX && pos.variant() != cra::chess::V_RACE;
    }
    std::vector<Action> legal_actions() const override {
        std::vector<cra::chess::Move> mv;
        pos.legal_moves(mv);
        return std::vector<Action>(mv.begin(), mv.end());
    }
    void set(const std::string& fenStr, bool isChess960, int variant) override {
        pos.set(fenStr, isChess960, cra::chess::Variant(variant));
    }
    void get_state_planes(bool normalize, float* inputPlanes, Version version) const override {
        (void)version;                                   // the layout was fixed with the configuration (it is derived from the same version)
        cra::chess::board_to_planes(pos, refshim::config().layout, normalize, inputPlanes);
        cra::BoardDesc d;
        cra::chess::pack_desc(pos, d, cra::layout_needs_move_features(refshim::config().layout));
        refshim::pending_descs().push_back(d);
    }
    // what a maintainer adds to BoardState for the descriptor-fed SearchThread (integration/searchthread_hip.patch): the position as the
    // 192-byte descriptor the GPU plane builder reads (include/crazyara_hip.h: mi_pos_desc)
    void fill_board_desc(void* out) const {
        cra::BoardDesc d;
        cra::chess::pack_desc(pos, d, cra::layout_needs_move_features(refshim::config().layout));
        std::memcpy(out, &d, sizeof(d));
    }
    unsigned int steps_from_null() const override { return unsigned(pos.game_ply()); }     // boardstate.cpp:82-85
    bool is_chess960() const override { return pos.is_chess960(); }
    std::string fen() const override { return pos.fen(); }
    void do_action(Action action) override { pos.do_move(cra::chess::Move(action)); }
    void undo_action(Action) override { throw std::logic_error("undo_action is not used by the search"); }
    void prepare_action() override {}
    unsigned int number_repetitions() const override { return unsigned(pos.number_repetitions()); }
    int side_to_move() const override { return int(pos.side_to_move()); }
    Key hash_key() const override { return pos.key(); }
    void flip() override { throw std::logic_error("flip is not used by the search"); }
    Action uci_to_action(std::string& uciStr) const override { return Action(pos.uci_to_move(uciStr)); }
    std::string action_to_san(Action action, const std::vector<Action>&, bool, bool) const override {
        return pos.move_to_san(cra::chess::Move(action));
    }
    TerminalType is_terminal(size_t numberLegalMoves, float&) const override {
        return TerminalType(int(pos.is_terminal(numberLegalMoves)));                       // same enum values (state.h)
    }
    bool gives_check(Action action) const override { return pos.gives_check(cra::chess::Move(action)); }
    void print(std::ostream& os) const override { os << pos.fen(); }
    Tablebase::WDLScore check_for_tablebase_wdl(Tablebase::ProbeState& result) override {
        result = Tablebase::FAIL;
        return Tablebase::WDLDraw;
    }
    void set_auxiliary_outputs(const float*) override {}
    PommermanState* clone() const override { return new PommermanState(*this); }
    void init(int variant, bool isChess960) override {
        pos.set(cra::chess::start_fen(cra::chess::Variant(variant)), isChess960, cra::chess::Variant(variant));
    }
    GamePhase get_phase(unsigned int, GamePhaseDefinition) const override { return GamePhase(0); }   // one net per search here
};
