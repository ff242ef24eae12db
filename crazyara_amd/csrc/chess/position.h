// Own bitboard position for the chess family the BASELINE configs use: chess, chess960, crazyhouse, 3check, KOTH.
// Replaces what the reference takes from its un-vendored multi-variant Stockfish fork (engine/3rdparty/Stockfish, empty
// in the mount; call sites: engine/src/environments/chess_related/board.cpp:35-275, boardstate.cpp:61-248).
// Move integers are this library's own encoding -- parity with the reference is defined on UCI strings and FENs.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace cra {
namespace chess {

typedef uint64_t Bitboard;
typedef uint64_t Key;
typedef uint32_t Move;      // bits 0-5 to | 6-11 from | 12-14 kind | 15-17 piece type (promotion target / dropped piece)

enum Color : int { WHITE = 0, BLACK = 1 };
enum PieceType : int { NO_PIECE_TYPE = 0, PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING, PIECE_TYPE_NB };
enum MoveKind : int { NORMAL = 0, PROMOTION = 1, ENPASSANT = 2, CASTLING = 3, DROP = 4 };
enum Variant : int { V_CHESS = 0, V_CRAZYHOUSE = 1, V_KOTH = 2, V_THREECHECK = 3,
                     // lichess variants of the MultiAra build: antichess (captures compulsory, no check, king promotion,
                     // win by losing everything or being stalemated), horde (White = 36 pawns without a king), racing kings
                     // (no checks, first king on the eighth rank), atomic (captures explode, kings never capture, touching kings
                     // are immune to check, blowing up the enemy king wins at once)
                     V_ANTI = 4, V_ATOMIC = 5, V_HORDE = 6, V_RACE = 7 };
enum CastlingRight : int { WHITE_OO = 1, WHITE_OOO = 2, BLACK_OO = 4, BLACK_OOO = 8 };
enum TerminalType : int { TERMINAL_LOSS = 0, TERMINAL_DRAW = 1, TERMINAL_WIN = 2, TERMINAL_CUSTOM = 3, TERMINAL_NONE = 4 };  // state.h

constexpr int SQ_NONE = 64;
constexpr Move MOVE_NONE = 0;

inline int to_sq(Move m) { return m & 63; }
inline int from_sq(Move m) { return (m >> 6) & 63; }
inline MoveKind kind_of(Move m) { return MoveKind((m >> 12) & 7); }
inline PieceType piece_of(Move m) { return PieceType((m >> 15) & 7); }   // promotion type or dropped type
inline Move make_move(int from, int to, MoveKind k = NORMAL, PieceType pt = NO_PIECE_TYPE) {
    return Move(to | (from << 6) | (int(k) << 12) | (int(pt) << 15));
}
inline Move make_drop(int to, PieceType pt) { return make_move(to, to, DROP, pt); }

inline int file_of(int sq) { return sq & 7; }
inline int rank_of(int sq) { return sq >> 3; }
inline Bitboard sq_bb(int sq) { return Bitboard(1) << sq; }
inline int popcount(Bitboard b) { return __builtin_popcountll(b); }
inline int lsb(Bitboard b) { return __builtin_ctzll(b); }
inline int pop_lsb(Bitboard& b) { int s = lsb(b); b &= b - 1; return s; }

Variant variant_from_name(const std::string& name);    // UCI::variant_from_name call site: boardstate.h:318-320
const char* variant_name(Variant v);
std::string start_fen(Variant v);                       // StateConstantsBoard::start_fen, boardstate.h:322-385
std::string chess960_start_fen(int index);              // Scharnagl numbering 0..959 (boardstate.cpp:260-271 picks one at random)

class Position {
public:
    Position();
    // Board::set (board.cpp:262-266).  Accepts crazyhouse pockets as "[..]" or as a 9th "/" field, promoted marks "~",
    // 3check counters "3+3" (remaining checks, before the clocks) or lichess "+1+2" (given, trailing), X-FEN / Shredder
    // castling letters.  Throws std::invalid_argument on malformed input.
    void set(const std::string& fen, bool is_chess960, Variant v);
    std::string fen() const;

    // ---- queries used by the input planes (inputrepresentation.cpp) ----
    Color side_to_move() const { return stm_; }
    Bitboard pieces() const { return by_color_[0] | by_color_[1]; }
    Bitboard pieces(Color c) const { return by_color_[c]; }
    Bitboard pieces(PieceType pt) const { return by_type_[pt]; }
    Bitboard pieces(Color c, PieceType pt) const { return by_color_[c] & by_type_[pt]; }
    int piece_on(int sq) const { return board_[sq]; }          // 0 empty, else color*8 + type
    int count(Color c, PieceType pt) const { return popcount(pieces(c, pt)); }
    int count_all() const { return popcount(pieces()); }
    Bitboard promoted_pieces() const { return promoted_; }
    int pocket_count(Color c, PieceType pt) const { return in_hand_[c][pt]; }
    int ep_square() const { return ep_; }
    bool can_castle(int cr) const { return (castling_ & cr) != 0; }
    int rule50_count() const { return rule50_; }
    int game_ply() const { return game_ply_; }
    int plies_from_null() const { return int(keys_.size()) - 1; }
    bool is_chess960() const { return chess960_; }
    Variant variant() const { return variant_; }
    bool is_house() const { return variant_ == V_CRAZYHOUSE; }
    int checks_given(Color c) const { return checks_given_[c]; }
    Bitboard checkers() const { return checkers_; }
    bool opposite_bishops() const;
    Key key() const { return keys_.back(); }
    int king_square(Color c) const { return pieces(c, KING) ? lsb(pieces(c, KING)) : SQ_NONE; }
    int castling_rook_square(int cr) const;
    // Board::number_repetitions (board.cpp:132-141): 0 or 1 only -- the "== 2" branch is unreachable in the reference
    int number_repetitions() const { return repetition_ != 0 ? 1 : 0; }
    bool can_claim_3fold_repetition() const { return repetition_ < 0; }        // board.cpp:143-149
    // last moves, most recent first, capped at 8 (Board::add_move_to_list, board.cpp:223-232)
    const std::vector<Move>& last_moves() const { return last_moves_; }
    void clear_last_moves() { last_moves_.clear(); }

    // ---- move generation / execution ----
    Bitboard attackers_to(int sq, Bitboard occ) const;
    void legal_moves(std::vector<Move>& out) const;            // MoveList<LEGAL>
    std::vector<Move> legal_moves() const { std::vector<Move> v; legal_moves(v); return v; }
    bool gives_check(Move m) const;
    void do_move(Move m) { do_move(m, nullptr); }
    // known_key: the position key after the move if the caller has it (a search tree node remembers its key): saves the recomputation
    void do_move(Move m, const Key* known_key);
    Move uci_to_move(const std::string& uci) const;            // UCI::to_move; MOVE_NONE if not legal
    // pgn_move (board.cpp:277-359) without the {book} / '#' decorations: the reference's own SAN dialect -- promotions without
    // '=', pawn drops as "P@e4", disambiguation by is_pgn_move_ambiguous (board.cpp:362-383), '+' when the move gives check
    std::string move_to_san(Move m) const;
    std::string move_to_uci(Move m) const;                     // UCI::move (castling: e1g1 classic, king-takes-rook in 960)
    // origin/destination as the policy labels see them (castling: classic -> king's two-step target, 960 -> rook square)
    void label_squares(Move m, int& from, int& to) const;

    // ---- terminal rules: BoardState::is_terminal (boardstate.cpp:143-226) with Board helpers (board.cpp:151-221) ----
    bool is_50_move_rule_draw(size_t n_legal) const;
    bool draw_by_insufficient_material() const;
    TerminalType is_terminal(size_t n_legal) const;

    // Board::get_phase (board.cpp:540-587) with its helpers get_majors_and_minors_count / is_backrank_sparse / get_mixedness
    // (board.cpp:446-538; the lichess / scalachess Divider): definition 0 = lichess (three phases: opening 0, middlegame 1, endgame 2),
    // 1 = movecount (num_phases equal slices of an average game of 42.85 moves).  One phase: always 0.
    int game_phase(unsigned num_phases, int definition) const;
    int majors_and_minors() const { return popcount(by_type_[QUEEN] | by_type_[ROOK] | by_type_[KNIGHT] | by_type_[BISHOP]); }
    bool backrank_sparse() const;
    int mixedness() const;

    uint64_t perft(int depth) const;

private:
    void put_piece(Color c, PieceType pt, int sq);
    void remove_piece(int sq);
    void move_piece(int from, int to);
    bool pseudo_is_legal(Move m) const;
    void gen_pseudo(std::vector<Move>& out) const;
    void set_castling_right(Color c, int rook_sq);
    void update_checkers();
    void compute_repetition();
    Key compute_key() const;

    Bitboard by_type_[PIECE_TYPE_NB];
    Bitboard by_color_[2];
    uint8_t board_[64];
    Bitboard promoted_;
    int in_hand_[2][PIECE_TYPE_NB];
    int castling_;
    int castling_rook_[4];       // index: 0 WHITE_OO, 1 WHITE_OOO, 2 BLACK_OO, 3 BLACK_OOO
    uint8_t castling_mask_[64];
    int ep_, rule50_, game_ply_;
    Color stm_;
    bool chess960_;
    Variant variant_;
    int checks_given_[2];
    Bitboard checkers_;
    int repetition_;
    std::vector<Key> keys_;      // keys since the last null/setup, oldest first (the StateInfo chain's key column)
    std::vector<uint8_t> rep_flags_;   // per key: that state's repetition != 0
    std::vector<Move> last_moves_;
};

void init_bitboards();   // idempotent; called by Position()

}  // namespace chess
}  // namespace cra
