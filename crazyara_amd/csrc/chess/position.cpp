#include "position.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <mutex>
#include <sstream>
#include <stdexcept>

namespace cra {
namespace chess {

// =====================================================================================================================
// tables
// =====================================================================================================================
namespace {
Bitboard g_knight[64], g_king[64], g_pawn_att[2][64], g_ray[8][64], g_between[64][64];
Key z_psq[2][PIECE_TYPE_NB][64], z_ep[8], z_castling[16], z_side, z_hand[2][PIECE_TYPE_NB][32], z_checks[2][4];
std::once_flag g_once;

const int kDirDf[8] = {0, 1, 1, 1, 0, -1, -1, -1};   // N NE E SE S SW W NW
const int kDirDr[8] = {1, 1, 0, -1, -1, -1, 0, 1};
constexpr Bitboard kCenter = (Bitboard(1) << 27) | (Bitboard(1) << 28) | (Bitboard(1) << 35) | (Bitboard(1) << 36);  // d4 e4 d5 e5

inline bool dir_positive(int d) { return d == 0 || d == 1 || d == 2 || d == 7; }

inline Bitboard ray_attack(int d, int sq, Bitboard occ) {
    Bitboard att = g_ray[d][sq];
    const Bitboard blockers = att & occ;
    if (blockers) {
        const int b = dir_positive(d) ? lsb(blockers) : 63 - __builtin_clzll(blockers);
        att ^= g_ray[d][b];
    }
    return att;
}
inline Bitboard bishop_attacks(int sq, Bitboard occ) {
    return ray_attack(1, sq, occ) | ray_attack(3, sq, occ) | ray_attack(5, sq, occ) | ray_attack(7, sq, occ);
}
inline Bitboard rook_attacks(int sq, Bitboard occ) {
    return ray_attack(0, sq, occ) | ray_attack(2, sq, occ) | ray_attack(4, sq, occ) | ray_attack(6, sq, occ);
}

uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void init_tables() {
    for (int sq = 0; sq < 64; ++sq) {
        const int f = file_of(sq), r = rank_of(sq);
        auto add = [&](Bitboard& b, int df, int dr) {
            const int nf = f + df, nr = r + dr;
            if (nf >= 0 && nf < 8 && nr >= 0 && nr < 8) b |= sq_bb(nr * 8 + nf);
        };
        const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
        g_knight[sq] = g_king[sq] = 0;
        for (auto& k : kn) add(g_knight[sq], k[0], k[1]);
        for (int d = 0; d < 8; ++d) add(g_king[sq], kDirDf[d], kDirDr[d]);
        g_pawn_att[WHITE][sq] = g_pawn_att[BLACK][sq] = 0;
        add(g_pawn_att[WHITE][sq], -1, 1);
        add(g_pawn_att[WHITE][sq], 1, 1);
        add(g_pawn_att[BLACK][sq], -1, -1);
        add(g_pawn_att[BLACK][sq], 1, -1);
        for (int d = 0; d < 8; ++d) {
            Bitboard b = 0;
            int nf = f + kDirDf[d], nr = r + kDirDr[d];
            while (nf >= 0 && nf < 8 && nr >= 0 && nr < 8) {
                b |= sq_bb(nr * 8 + nf);
                nf += kDirDf[d];
                nr += kDirDr[d];
            }
            g_ray[d][sq] = b;
        }
    }
    for (int a = 0; a < 64; ++a)
        for (int b = 0; b < 64; ++b) {
            g_between[a][b] = 0;
            for (int d = 0; d < 8; ++d)
                if (g_ray[d][a] & sq_bb(b)) g_between[a][b] = g_ray[d][a] & g_ray[(d + 4) & 7][b];
        }
    uint64_t seed = 0x5EEDC0DE1234ull;
    for (int c = 0; c < 2; ++c)
        for (int pt = 0; pt < PIECE_TYPE_NB; ++pt) {
            for (int s = 0; s < 64; ++s) z_psq[c][pt][s] = splitmix(seed);
            for (int n = 0; n < 32; ++n) z_hand[c][pt][n] = splitmix(seed);
        }
    for (auto& k : z_ep) k = splitmix(seed);
    for (auto& k : z_castling) k = splitmix(seed);
    z_side = splitmix(seed);
    for (int c = 0; c < 2; ++c)
        for (int n = 0; n < 4; ++n) z_checks[c][n] = splitmix(seed);
}

const char kPieceChars[] = " PNBRQK";
inline PieceType type_from_char(char c) {
    switch (std::toupper(static_cast<unsigned char>(c))) {
        case 'P': return PAWN;
        case 'N': return KNIGHT;
        case 'B': return BISHOP;
        case 'R': return ROOK;
        case 'Q': return QUEEN;
        case 'K': return KING;
    }
    return NO_PIECE_TYPE;
}
inline int cr_index(int cr) { return cr == WHITE_OO ? 0 : cr == WHITE_OOO ? 1 : cr == BLACK_OO ? 2 : 3; }
inline std::string sq_str(int sq) { return std::string{char('a' + file_of(sq)), char('1' + rank_of(sq))}; }
}  // namespace

void init_bitboards() { std::call_once(g_once, init_tables); }

Variant variant_from_name(const std::string& n) {
    if (n == "chess" || n == "standard" || n == "fischerandom" || n == "chess960") return V_CHESS;
    if (n == "crazyhouse") return V_CRAZYHOUSE;
    if (n == "kingofthehill" || n == "koth") return V_KOTH;
    if (n == "3check" || n == "threecheck") return V_THREECHECK;
    if (n == "antichess" || n == "giveaway") return V_ANTI;
    if (n == "atomic") return V_ATOMIC;
    if (n == "horde") return V_HORDE;
    if (n == "racingkings") return V_RACE;
    throw std::invalid_argument("unknown variant '" + n + "'");
}

const char* variant_name(Variant v) {
    static const char* names[] = {"chess", "crazyhouse", "kingofthehill", "3check", "antichess", "atomic", "horde", "racingkings"};
    return names[v];
}

std::string start_fen(Variant v) {
    switch (v) {
        case V_CRAZYHOUSE: return "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR[] w KQkq - 0 1";
        case V_THREECHECK: return "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 3+3 0 1";
        case V_ANTI: return "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w - - 0 1";
        case V_HORDE: return "rnbqkbnr/pppppppp/8/1PP2PP1/PPPPPPPP/PPPPPPPP/PPPPPPPP/PPPPPPPP w kq - 0 1";
        case V_RACE: return "8/8/8/8/8/8/krbnNBRK/qrbnNBRQ w - - 0 1";
        default: return "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1";
    }
}

std::string chess960_start_fen(int index) {
    int n = ((index % 960) + 960) % 960;
    char p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    p[(n % 4) * 2 + 1] = 'B';
    n /= 4;
    p[(n % 4) * 2] = 'B';
    n /= 4;
    auto place_nth_free = [&](int k, char c) {
        for (int i = 0; i < 8; ++i)
            if (!p[i] && k-- == 0) { p[i] = c; return; }
    };
    place_nth_free(n % 6, 'Q');
    n /= 6;
    static const int kn[10][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {1, 2}, {1, 3}, {1, 4}, {2, 3}, {2, 4}, {3, 4}};
    place_nth_free(kn[n][1], 'N');   // place the higher slot first so the lower index stays valid
    place_nth_free(kn[n][0], 'N');
    place_nth_free(0, 'R');
    place_nth_free(0, 'K');
    place_nth_free(0, 'R');
    std::string white(p, 8), black(white);
    std::transform(black.begin(), black.end(), black.begin(), ::tolower);
    return black + "/pppppppp/8/8/8/8/PPPPPPPP/" + white + " w KQkq - 0 1";
}

// =====================================================================================================================
// setup
// =====================================================================================================================
Position::Position() {
    init_bitboards();
    std::memset(by_type_, 0, sizeof(by_type_));
    std::memset(by_color_, 0, sizeof(by_color_));
    std::memset(board_, 0, sizeof(board_));
    std::memset(in_hand_, 0, sizeof(in_hand_));
    std::memset(castling_rook_, 0, sizeof(castling_rook_));
    std::memset(castling_mask_, 0, sizeof(castling_mask_));
    promoted_ = 0;
    castling_ = 0;
    ep_ = SQ_NONE;
    rule50_ = game_ply_ = 0;
    stm_ = WHITE;
    chess960_ = false;
    variant_ = V_CHESS;
    checks_given_[0] = checks_given_[1] = 0;
    checkers_ = 0;
    repetition_ = 0;
}

void Position::put_piece(Color c, PieceType pt, int sq) {
    board_[sq] = uint8_t(c * 8 + pt);
    by_type_[pt] |= sq_bb(sq);
    by_color_[c] |= sq_bb(sq);
}
void Position::remove_piece(int sq) {
    const int pc = board_[sq];
    by_type_[pc & 7] &= ~sq_bb(sq);
    by_color_[pc >> 3] &= ~sq_bb(sq);
    board_[sq] = 0;
}
void Position::move_piece(int from, int to) {
    const int pc = board_[from];
    const Bitboard ft = sq_bb(from) | sq_bb(to);
    by_type_[pc & 7] ^= ft;
    by_color_[pc >> 3] ^= ft;
    board_[from] = 0;
    board_[to] = uint8_t(pc);
}

int Position::castling_rook_square(int cr) const { return castling_rook_[cr_index(cr)]; }

void Position::set_castling_right(Color c, int rsq) {
    const int ksq = king_square(c);
    const int cr = (c == WHITE ? 1 : 4) << (ksq < rsq ? 0 : 1);
    castling_ |= cr;
    castling_mask_[ksq] |= cr;
    castling_mask_[rsq] |= cr;
    castling_rook_[cr_index(cr)] = rsq;
}

void Position::set(const std::string& fen, bool is_chess960, Variant v) {
    *this = Position();
    chess960_ = is_chess960;
    variant_ = v;
    std::istringstream ss(fen);
    std::string placement, side, castling, ep, tok;
    ss >> placement >> side >> castling >> ep;
    if (placement.empty() || side.empty()) throw std::invalid_argument("malformed FEN: " + fen);
    std::vector<std::string> rest;
    while (ss >> tok) rest.push_back(tok);

    // 1. piece placement (+ pockets)
    int sq = 56, slashes = 0;
    bool in_pocket = false;
    int last_sq = SQ_NONE;
    for (char ch : placement) {
        if (ch == '[') { in_pocket = true; continue; }
        if (ch == ']') { in_pocket = false; continue; }
        if (in_pocket) {
            if (ch == '-') continue;
            const PieceType pt = type_from_char(ch);
            if (pt == NO_PIECE_TYPE) throw std::invalid_argument("bad pocket char in FEN: " + fen);
            ++in_hand_[std::isupper(static_cast<unsigned char>(ch)) ? WHITE : BLACK][pt];
            continue;
        }
        if (ch == '/') {
            if (++slashes == 8) { in_pocket = true; continue; }   // lichess style: 9th field is the pocket
            sq -= 16;
            continue;
        }
        if (std::isdigit(static_cast<unsigned char>(ch))) { sq += ch - '0'; continue; }
        if (ch == '~') {
            if (last_sq != SQ_NONE) promoted_ |= sq_bb(last_sq);
            continue;
        }
        const PieceType pt = type_from_char(ch);
        if (pt == NO_PIECE_TYPE || sq < 0 || sq > 63) throw std::invalid_argument("bad piece placement in FEN: " + fen);
        put_piece(std::isupper(static_cast<unsigned char>(ch)) ? WHITE : BLACK, pt, sq);
        last_sq = sq++;
    }
    // 2. side to move
    stm_ = side == "w" ? WHITE : BLACK;
    // 3. castling (KQkq, Shredder-FEN / X-FEN letters)
    for (char ch : castling) {
        if (ch == '-') continue;
        const Color c = std::isupper(static_cast<unsigned char>(ch)) ? WHITE : BLACK;
        const int back = c == WHITE ? 0 : 56;
        const int rook_pc = c * 8 + ROOK;
        if (king_square(c) == SQ_NONE) continue;
        const char up = char(std::toupper(static_cast<unsigned char>(ch)));
        int rsq = SQ_NONE;
        if (up == 'K') { for (rsq = back + 7; rsq >= back && board_[rsq] != rook_pc; --rsq) {} if (rsq < back) rsq = SQ_NONE; }
        else if (up == 'Q') { for (rsq = back; rsq <= back + 7 && board_[rsq] != rook_pc; ++rsq) {} if (rsq > back + 7) rsq = SQ_NONE; }
        else if (up >= 'A' && up <= 'H') { rsq = back + (up - 'A'); if (board_[rsq] != rook_pc) rsq = SQ_NONE; }
        if (rsq != SQ_NONE && rank_of(king_square(c)) == rank_of(back)) set_castling_right(c, rsq);
    }
    // 4. en passant: kept only if an enemy... i.e. a pawn of the side to move attacks it and the pushed pawn stands behind it
    if (ep.size() == 2 && ep[0] >= 'a' && ep[0] <= 'h' && (ep[1] == '3' || ep[1] == '6')) {
        const int e = (ep[1] - '1') * 8 + (ep[0] - 'a');
        const Color them = Color(stm_ ^ 1);
        const int behind = stm_ == WHITE ? e - 8 : e + 8;
        if ((g_pawn_att[them][e] & pieces(stm_, PAWN)) && (pieces(them, PAWN) & sq_bb(behind)) && !(pieces() & sq_bb(e))) ep_ = e;
    }
    // 5. check counters / clocks
    std::vector<int> nums;
    for (const std::string& t : rest) {
        const size_t plus = t.find('+');
        if (plus != std::string::npos) {
            if (plus == 0) {   // lichess: "+w+b" = checks already given
                const size_t p2 = t.find('+', 1);
                if (p2 != std::string::npos) {
                    checks_given_[WHITE] = std::stoi(t.substr(1, p2 - 1));
                    checks_given_[BLACK] = std::stoi(t.substr(p2 + 1));
                }
            } else {           // "3+3" = remaining checks (white+black)
                checks_given_[WHITE] = std::max(0, 3 - std::stoi(t.substr(0, plus)));
                checks_given_[BLACK] = std::max(0, 3 - std::stoi(t.substr(plus + 1)));
            }
        } else if (!t.empty() && (std::isdigit(static_cast<unsigned char>(t[0])) || t[0] == '-')) {
            nums.push_back(std::stoi(t));
        }
    }
    rule50_ = nums.size() > 0 ? nums[0] : 0;
    const int fullmove = nums.size() > 1 ? nums[1] : 1;
    game_ply_ = std::max(2 * (fullmove - 1), 0) + (stm_ == BLACK ? 1 : 0);
    update_checkers();
    keys_.assign(1, compute_key());
    rep_flags_.assign(1, 0);
    repetition_ = 0;
}

std::string Position::fen() const {
    std::ostringstream os;
    for (int r = 7; r >= 0; --r) {
        int empty = 0;
        for (int f = 0; f < 8; ++f) {
            const int sq = r * 8 + f, pc = board_[sq];
            if (!pc) { ++empty; continue; }
            if (empty) { os << empty; empty = 0; }
            const char ch = kPieceChars[pc & 7];
            os << char((pc >> 3) == WHITE ? ch : std::tolower(ch));
            if (is_house() && (promoted_ & sq_bb(sq))) os << '~';
        }
        if (empty) os << empty;
        if (r) os << '/';
    }
    if (is_house()) {
        os << '[';
        for (int c = 0; c < 2; ++c)
            for (int pt = QUEEN; pt >= PAWN; --pt)
                for (int n = 0; n < in_hand_[c][pt]; ++n) os << char(c == WHITE ? kPieceChars[pt] : std::tolower(kPieceChars[pt]));
        os << ']';
    }
    os << (stm_ == WHITE ? " w " : " b ");
    if (!castling_) os << '-';
    else {
        static const int crs[4] = {WHITE_OO, WHITE_OOO, BLACK_OO, BLACK_OOO};
        static const char std_ch[4] = {'K', 'Q', 'k', 'q'};
        for (int i = 0; i < 4; ++i)
            if (castling_ & crs[i]) {
                if (chess960_) os << char((i < 2 ? 'A' : 'a') + file_of(castling_rook_[i]));
                else os << std_ch[i];
            }
    }
    os << ' ' << (ep_ == SQ_NONE ? std::string("-") : sq_str(ep_));
    if (variant_ == V_THREECHECK) os << ' ' << (3 - checks_given_[WHITE]) << '+' << (3 - checks_given_[BLACK]);
    os << ' ' << rule50_ << ' ' << 1 + (game_ply_ - (stm_ == BLACK ? 1 : 0)) / 2;
    return os.str();
}

Key Position::compute_key() const {
    Key k = 0;
    Bitboard occ = pieces();
    while (occ) {
        const int sq = pop_lsb(occ);
        k ^= z_psq[board_[sq] >> 3][board_[sq] & 7][sq];
    }
    if (ep_ != SQ_NONE) k ^= z_ep[file_of(ep_)];
    k ^= z_castling[castling_];
    if (stm_ == BLACK) k ^= z_side;
    if (is_house())
        for (int c = 0; c < 2; ++c)
            for (int pt = PAWN; pt <= QUEEN; ++pt) k ^= z_hand[c][pt][in_hand_[c][pt] & 31];
    if (variant_ == V_THREECHECK) k ^= z_checks[0][checks_given_[0] & 3] ^ z_checks[1][checks_given_[1] & 3];
    return k;
}

bool Position::opposite_bishops() const {
    if (count(WHITE, BISHOP) != 1 || count(BLACK, BISHOP) != 1) return false;
    const int a = lsb(pieces(WHITE, BISHOP)), b = lsb(pieces(BLACK, BISHOP));
    return ((file_of(a) + rank_of(a)) & 1) != ((file_of(b) + rank_of(b)) & 1);
}

// =====================================================================================================================
// attacks / legality
// =====================================================================================================================
Bitboard Position::attackers_to(int sq, Bitboard occ) const {
    return (g_pawn_att[BLACK][sq] & pieces(WHITE, PAWN)) | (g_pawn_att[WHITE][sq] & pieces(BLACK, PAWN)) |
           (g_knight[sq] & by_type_[KNIGHT]) | (g_king[sq] & by_type_[KING]) |
           (bishop_attacks(sq, occ) & (by_type_[BISHOP] | by_type_[QUEEN])) |
           (rook_attacks(sq, occ) & (by_type_[ROOK] | by_type_[QUEEN]));
}

void Position::update_checkers() {
    const int ksq = king_square(stm_);
    // antichess: the king is an ordinary piece, there is no check; horde: White has no king; atomic: a king next to the enemy
    // king cannot be captured (the capture would blow up the capturer's own king), so it is never in check there
    const bool exempt = ksq == SQ_NONE || variant_ == V_ANTI ||
                        (variant_ == V_ATOMIC && (g_king[ksq] & pieces(Color(stm_ ^ 1), KING)));
    checkers_ = exempt ? 0 : attackers_to(ksq, pieces()) & pieces(Color(stm_ ^ 1));
}

bool Position::pseudo_is_legal(Move m) const {
    const Color us = stm_, them = Color(us ^ 1);
    const int ksq = king_square(us);
    if (ksq == SQ_NONE || variant_ == V_ANTI) return true;
    if (variant_ == V_ATOMIC && kind_of(m) != CASTLING) {
        // play it on a copy (captures explode): my king must survive; blowing up the enemy king wins on the spot and overrides any
        // check; otherwise my king must not be attacked -- unless the kings stand next to each other
        Position p(*this);
        p.do_move(m);
        const int k2 = p.king_square(us);
        if (k2 == SQ_NONE) return false;
        if (p.king_square(them) == SQ_NONE) return true;
        if (g_king[k2] & p.pieces(them, KING)) return true;
        return !(p.attackers_to(k2, p.pieces()) & p.pieces(them));
    }
    const Bitboard occ = pieces();
    const int to = to_sq(m);
    switch (kind_of(m)) {
        case DROP:
            if (!checkers_) return true;
            return !(attackers_to(ksq, occ | sq_bb(to)) & pieces(them));
        case ENPASSANT: {
            const int from = from_sq(m), cap = us == WHITE ? to - 8 : to + 8;
            const Bitboard occ2 = (occ ^ sq_bb(from) ^ sq_bb(cap)) | sq_bb(to);
            return !(attackers_to(ksq, occ2) & pieces(them) & ~sq_bb(cap));
        }
        case CASTLING: return true;   // fully checked at generation
        default: {
            const int from = from_sq(m);
            const Bitboard occ2 = (occ ^ sq_bb(from)) | sq_bb(to);
            const int k2 = from == ksq ? to : ksq;
            return !(attackers_to(k2, occ2) & pieces(them) & ~sq_bb(to));
        }
    }
}

void Position::gen_pseudo(std::vector<Move>& out) const {
    const Color us = stm_, them = Color(us ^ 1);
    const Bitboard occ = pieces(), own = pieces(us), enemy = pieces(them);
    // pawns
    Bitboard pawns = pieces(us, PAWN);
    const int up = us == WHITE ? 8 : -8;
    const int promo_rank = us == WHITE ? 7 : 0, start_rank = us == WHITE ? 1 : 6;
    auto add_pawn = [&](int from, int to) {
        if (rank_of(to) == promo_rank) {
            for (PieceType pt : {QUEEN, ROOK, BISHOP, KNIGHT}) out.push_back(make_move(from, to, PROMOTION, pt));
            if (variant_ == V_ANTI) out.push_back(make_move(from, to, PROMOTION, KING));     // antichess: promotion to a king
        } else {
            out.push_back(make_move(from, to));
        }
    };
    while (pawns) {
        const int from = pop_lsb(pawns);
        const int one = from + up;
        if (one >= 0 && one < 64 && !(occ & sq_bb(one))) {
            add_pawn(from, one);
            // horde: the white pawns of the first rank may also advance two squares
            const bool two = rank_of(from) == start_rank || (variant_ == V_HORDE && us == WHITE && rank_of(from) == 0);
            if (two && !(occ & sq_bb(one + up))) out.push_back(make_move(from, one + up));
        }
        Bitboard caps = g_pawn_att[us][from] & enemy;
        while (caps) add_pawn(from, pop_lsb(caps));
        if (ep_ != SQ_NONE && (g_pawn_att[us][from] & sq_bb(ep_))) out.push_back(make_move(from, ep_, ENPASSANT));
    }
    auto add_targets = [&](int from, Bitboard t) {
        t &= ~own;
        while (t) out.push_back(make_move(from, pop_lsb(t)));
    };
    Bitboard b = pieces(us, KNIGHT);
    while (b) { const int s = pop_lsb(b); add_targets(s, g_knight[s]); }
    b = pieces(us, BISHOP);
    while (b) { const int s = pop_lsb(b); add_targets(s, bishop_attacks(s, occ)); }
    b = pieces(us, ROOK);
    while (b) { const int s = pop_lsb(b); add_targets(s, rook_attacks(s, occ)); }
    b = pieces(us, QUEEN);
    while (b) { const int s = pop_lsb(b); add_targets(s, bishop_attacks(s, occ) | rook_attacks(s, occ)); }
    b = pieces(us, KING);
    while (b) { const int s = pop_lsb(b); add_targets(s, variant_ == V_ATOMIC ? g_king[s] & ~enemy : g_king[s]); }   // atomic kings never capture

    // castling (king-takes-rook encoding), legality checked here the way Stockfish's legal() does; none in antichess
    if (castling_ && !checkers_ && variant_ != V_ANTI) {
        const int ksq = king_square(us);
        for (int side = 0; side < 2; ++side) {
            const int cr = (us == WHITE ? 1 : 4) << side;
            if (!(castling_ & cr)) continue;
            const int rsq = castling_rook_[cr_index(cr)];
            const int back = us == WHITE ? 0 : 56;
            const int kto = back + (side == 0 ? 6 : 2), rto = back + (side == 0 ? 5 : 3);
            const Bitboard path = (g_between[ksq][kto] | sq_bb(kto) | g_between[rsq][rto] | sq_bb(rto)) & ~(sq_bb(ksq) | sq_bb(rsq));
            if (path & occ) continue;
            bool ok = true;
            const int step = kto > ksq ? 1 : -1;
            for (int s = kto; s != ksq; s -= step)
                if (attackers_to(s, occ) & enemy) { ok = false; break; }
            if (!ok) continue;
            if (chess960_ && (rook_attacks(kto, occ ^ sq_bb(rsq)) & pieces(them) & (by_type_[ROOK] | by_type_[QUEEN]))) continue;
            out.push_back(make_move(ksq, rsq, CASTLING));
        }
    }
    // drops
    if (is_house()) {
        const Bitboard empty = ~occ;
        for (PieceType pt : {PAWN, KNIGHT, BISHOP, ROOK, QUEEN}) {
            if (!in_hand_[us][pt]) continue;
            Bitboard t = empty;
            if (pt == PAWN) t &= ~(0xFFull | (0xFFull << 56));
            while (t) out.push_back(make_drop(pop_lsb(t), pt));
        }
    }
}

void Position::legal_moves(std::vector<Move>& out) const {
    out.clear();
    thread_local std::vector<Move> pseudo;           // one buffer per thread: this runs once per new search node
    pseudo.clear();
    gen_pseudo(pseudo);
    if (variant_ == V_ANTI) {                        // antichess: if a capture exists, a capture must be played
        out.reserve(pseudo.size());
        bool any_capture = false;
        for (Move m : pseudo) any_capture |= kind_of(m) == ENPASSANT || board_[to_sq(m)] != 0;
        for (Move m : pseudo)
            if (!any_capture || kind_of(m) == ENPASSANT || board_[to_sq(m)] != 0) out.push_back(m);
        return;
    }
    out.reserve(pseudo.size());                      // one allocation for a fresh vector instead of a doubling chain
    // Not in check: a move of a piece that is neither the king nor pinned to it cannot expose the king, and a drop never does --
    // only king moves, moves of pinned pieces and en-passant captures need the attack test (with ~3 instead of ~30 per position).
    // Atomic has its own legality (explosions), positions without a king (horde's white side) have nothing to protect.
    const Color us = stm_, them = Color(us ^ 1);
    const int ksq = king_square(us);
    const bool fast = !checkers_ && ksq != SQ_NONE && variant_ != V_ATOMIC;
    Bitboard pinned = 0;
    if (fast) {
        const Bitboard occ = pieces();
        const Bitboard orth = g_ray[0][ksq] | g_ray[2][ksq] | g_ray[4][ksq] | g_ray[6][ksq];
        const Bitboard diag = g_ray[1][ksq] | g_ray[3][ksq] | g_ray[5][ksq] | g_ray[7][ksq];
        Bitboard snipers = (orth & (pieces(them, ROOK) | pieces(them, QUEEN))) | (diag & (pieces(them, BISHOP) | pieces(them, QUEEN)));
        while (snipers) {
            const Bitboard b = g_between[ksq][pop_lsb(snipers)] & occ;
            if (b && !(b & (b - 1))) pinned |= b;              // exactly one piece in between (an enemy one there pins nothing we move)
        }
    }
    for (Move m : pseudo) {
        const MoveKind k = kind_of(m);
        const bool safe = fast && (k == DROP || ((k == NORMAL || k == PROMOTION) && from_sq(m) != ksq && !(pinned & sq_bb(from_sq(m)))));
        if (!safe && !pseudo_is_legal(m)) continue;
        if (variant_ == V_RACE && gives_check(m)) continue;     // racing kings: giving check is forbidden
        out.push_back(m);
    }
}

bool Position::gives_check(Move m) const {
    Position p(*this);
    p.do_move(m);
    return p.checkers() != 0;
}

// =====================================================================================================================
// do_move
// =====================================================================================================================
void Position::compute_repetition() {
    repetition_ = 0;
    const int n = int(keys_.size());
    const int end = is_house() ? n - 1 : std::min(rule50_, n - 1);
    if (end >= 4) {
        const Key k = keys_.back();
        for (int i = 4; i <= end; i += 2) {
            const int idx = n - 1 - i;
            if (keys_[idx] == k) {
                repetition_ = rep_flags_[idx] ? -i : i;
                break;
            }
        }
    }
}

void Position::do_move(Move m, const Key* known_key) {
    const Color us = stm_, them = Color(us ^ 1);
    const int to = to_sq(m), from = from_sq(m);
    const MoveKind kind = kind_of(m);
    // Board::do_move -> add_move_to_list (board.cpp:223-237)
    last_moves_.insert(last_moves_.begin(), m);
    if (last_moves_.size() > 8) last_moves_.pop_back();
    ++game_ply_;
    ++rule50_;
    int new_ep = SQ_NONE;

    if (kind == DROP) {
        put_piece(us, piece_of(m), to);
        --in_hand_[us][piece_of(m)];
    } else if (kind == CASTLING) {
        const int back = us == WHITE ? 0 : 56;
        const bool oo = to > from;
        const int kto = back + (oo ? 6 : 2), rto = back + (oo ? 5 : 3);
        remove_piece(from);
        remove_piece(to);
        put_piece(us, KING, kto);
        put_piece(us, ROOK, rto);
        castling_ &= ~(castling_mask_[from] | castling_mask_[to]);
    } else {
        const int moving = board_[from] & 7;
        int capsq = to;
        if (kind == ENPASSANT) capsq = us == WHITE ? to - 8 : to + 8;
        const bool atomic_blast = variant_ == V_ATOMIC && board_[capsq] != 0;
        if (board_[capsq]) {
            int cap_type = board_[capsq] & 7;
            if (is_house()) {
                if (promoted_ & sq_bb(capsq)) cap_type = PAWN;
                ++in_hand_[us][cap_type];
            }
            promoted_ &= ~sq_bb(capsq);
            remove_piece(capsq);
            rule50_ = 0;
        }
        move_piece(from, to);
        if (promoted_ & sq_bb(from)) promoted_ = (promoted_ & ~sq_bb(from)) | sq_bb(to);
        if (moving == PAWN) {
            rule50_ = 0;
            // horde: a double step from the first rank cannot be captured en passant
            if ((to ^ from) == 16 && !(variant_ == V_HORDE && rank_of(from) == (us == WHITE ? 0 : 7))) {
                const int mid = (to + from) / 2;
                if (g_pawn_att[us][mid] & pieces(them, PAWN)) new_ep = mid;
            }
            if (kind == PROMOTION) {
                remove_piece(to);
                put_piece(us, piece_of(m), to);
                if (is_house()) promoted_ |= sq_bb(to);
            }
        }
        castling_ &= ~(castling_mask_[from] | castling_mask_[to]);
        if (atomic_blast) {
            // atomic: the capturing piece and every non-pawn piece on the squares around the destination are removed as well
            remove_piece(to);
            Bitboard ring = g_king[to] & pieces() & ~by_type_[PAWN];
            while (ring) {
                const int sq = pop_lsb(ring);
                castling_ &= ~castling_mask_[sq];
                remove_piece(sq);
            }
            new_ep = SQ_NONE;
        }
    }
    ep_ = new_ep;
    stm_ = them;
    update_checkers();
    if (variant_ == V_THREECHECK && checkers_) ++checks_given_[us];
    keys_.push_back(known_key ? *known_key : compute_key());
    rep_flags_.push_back(0);
    compute_repetition();
    rep_flags_.back() = repetition_ != 0;
}

// =====================================================================================================================
// UCI strings
// =====================================================================================================================
void Position::label_squares(Move m, int& from, int& to) const {
    from = from_sq(m);
    to = to_sq(m);
    if (kind_of(m) == CASTLING && !chess960_) {
        const int back = rank_of(from) * 8;
        to = back + (to > from ? 6 : 2);
    }
}

std::string Position::move_to_uci(Move m) const {
    if (m == MOVE_NONE) return "(none)";
    if (kind_of(m) == DROP) return std::string{kPieceChars[piece_of(m)], '@'} + sq_str(to_sq(m));
    int from, to;
    label_squares(m, from, to);
    std::string s = sq_str(from) + sq_str(to);
    if (kind_of(m) == PROMOTION) s += char(std::tolower(kPieceChars[piece_of(m)]));
    return s;
}

std::string Position::move_to_san(Move m) const {
    if (m == MOVE_NONE) return "(none)";
    const int from = from_sq(m), to = to_sq(m);
    const MoveKind kind = kind_of(m);
    std::string s;
    if (kind == CASTLING) {
        s = file_of(from) < file_of(to) ? "O-O" : "O-O-O";         // king-takes-rook encoding: the rook's file tells the side
    } else if (kind == DROP) {
        s = std::string{kPieceChars[piece_of(m)], '@'} + sq_str(to);
    } else {
        const int piece = board_[from], type = piece & 7;
        const bool capture = board_[to] != 0 || kind == ENPASSANT;
        std::string amb;
        if (type != PAWN) {
            bool ambiguous = false, same_file = false, same_rank = false;
            std::vector<Move> moves;
            legal_moves(moves);
            for (Move o : moves) {
                if (kind_of(o) == DROP) continue;
                const int of = from_sq(o);
                if (to_sq(o) == to && of != from && board_[of] == piece) {
                    ambiguous = true;
                    same_file |= file_of(of) == file_of(from);
                    same_rank |= rank_of(of) == rank_of(from);
                }
            }
            if (ambiguous) amb = same_file && same_rank ? sq_str(from) : same_file ? std::string(1, char('1' + rank_of(from))) : std::string(1, char('a' + file_of(from)));
        }
        if (type == PAWN) s = capture ? std::string(1, char('a' + file_of(from))) + "x" + sq_str(to) : sq_str(to);
        else s = std::string(1, kPieceChars[type]) + amb + (capture ? "x" : "") + sq_str(to);
        if (kind == PROMOTION) s += kPieceChars[piece_of(m)];
    }
    if (gives_check(m)) s += "+";
    return s;
}

Move Position::uci_to_move(const std::string& uci) const {
    std::string s = uci;
    if (s.size() == 5) s[4] = char(std::tolower(static_cast<unsigned char>(s[4])));
    std::vector<Move> moves;
    legal_moves(moves);
    for (Move m : moves)
        if (move_to_uci(m) == s) return m;
    return MOVE_NONE;
}

// =====================================================================================================================
// terminal rules
// =====================================================================================================================
bool Position::is_50_move_rule_draw(size_t n_legal) const {
    if (is_house()) return false;                                  // board.cpp:151-160
    return rule50_ > 99 && (!checkers_ || n_legal > 0);
}

bool Position::draw_by_insufficient_material() const {             // board.cpp:175-221
    if (variant_ != V_CHESS && variant_ != V_ATOMIC) return false;
    const int n = count_all();
    if (n > 4) return false;
    const int nb = popcount(by_type_[BISHOP]), nn = popcount(by_type_[KNIGHT]);
    return n == 2 || (n == 3 && nb == 1) || (n == 3 && nn == 1) ||
           (n == 4 && (count(WHITE, KNIGHT) == 2 || count(BLACK, KNIGHT) == 2));
}

TerminalType Position::is_terminal(size_t n_legal) const {          // boardstate.cpp:143-226
    const Color them = Color(stm_ ^ 1);
    if (variant_ == V_ATOMIC) {                                     // is_atomic_win / is_atomic_loss: a king has been blown up
        if (!pieces(them, KING)) return TERMINAL_WIN;
        if (!pieces(stm_, KING)) return TERMINAL_LOSS;
    }
    if (variant_ == V_ANTI) {                                       // is_anti_win / is_anti_loss: whoever has no piece left has won
        if (!pieces(stm_)) return TERMINAL_WIN;
        if (!pieces(them)) return TERMINAL_LOSS;
    }
    if (variant_ == V_HORDE) {                                      // is_horde_loss: the side without a king has lost all its pieces
        const Color horde = pieces(WHITE, KING) ? BLACK : WHITE;
        if (horde == stm_ && !pieces(horde)) return TERMINAL_LOSS;
    }
    if (variant_ == V_RACE && pieces(stm_, KING) && pieces(them, KING)) {   // is_race_win / _draw / _loss
        const int mine = king_square(stm_), theirs = king_square(them);
        if (rank_of(mine) == 7) return rank_of(theirs) == 7 ? TERMINAL_DRAW : TERMINAL_WIN;
        if (rank_of(theirs) == 7) {
            // White arrived first: Black may still equalise if its king can step to the eighth rank right now
            if (rank_of(mine) < (stm_ == WHITE ? 7 : 6)) return TERMINAL_LOSS;
            Bitboard b = g_king[mine] & (0xFFull << 56) & ~pieces(stm_);
            bool can_follow = false;
            while (b) can_follow |= !(attackers_to(pop_lsb(b), pieces()) & pieces(them));
            if (!can_follow) return TERMINAL_LOSS;
        }
    }
    if (variant_ == V_KOTH) {
        if (pieces(stm_, KING) & kCenter) return TERMINAL_WIN;
        if (pieces(Color(stm_ ^ 1), KING) & kCenter) return TERMINAL_LOSS;
    }
    if (variant_ == V_THREECHECK) {
        if (checks_given_[stm_] >= 3) return TERMINAL_WIN;
        if (checks_given_[stm_ ^ 1] >= 3) return TERMINAL_LOSS;
    }
    if (n_legal == 0) {
        if (variant_ == V_ANTI) return TERMINAL_WIN;                // a stalemate is a win in antichess
        return checkers_ ? TERMINAL_LOSS : TERMINAL_DRAW;
    }
    if (can_claim_3fold_repetition() || is_50_move_rule_draw(n_legal) || draw_by_insufficient_material()) return TERMINAL_DRAW;
    return TERMINAL_NONE;
}

// ---- game phase (training-data exporter choice of self-play, selfplay.cpp:232-238) ----
bool Position::backrank_sparse() const {                       // three or fewer pieces left on either side's own first rank
    return popcount(by_color_[WHITE] & 0xffull) <= 3 || popcount(by_color_[BLACK] & (0xffull << 56)) <= 3;
}

// Mixedness of the Divider: every 2x2 window of the board (lower-left corner on ranks 1-7, files a-g) scores by how many white and
// black pieces it holds and how far up the board it sits.  The score of a window is `base + slope term`, tabulated by (white, black)
// count; y = rank of the window's lower row, 1-based.
int Position::mixedness() const {
    auto window_score = [](int w, int b, int y) -> int {
        switch (w * 5 + b) {
            case 1 * 5 + 0: return 1 + (8 - y);
            case 2 * 5 + 0: return 2 + std::max(y - 2, 0);
            case 3 * 5 + 0: return 3 + std::max(y - 1, 0);
            case 4 * 5 + 0: return 3 + std::max(y - 1, 0);
            case 0 * 5 + 1: return 1 + y;
            case 1 * 5 + 1: return 5 + std::abs(3 - y);
            case 2 * 5 + 1: return 4 + y;
            case 3 * 5 + 1: return 5 + y;
            case 0 * 5 + 2: return 2 + std::max(6 - y, 0);
            case 1 * 5 + 2: return 4 + (6 - y);
            case 2 * 5 + 2: return 7;
            case 0 * 5 + 3: return 3 + std::max(7 - y, 0);
            case 1 * 5 + 3: return 5 + (6 - y);
            case 0 * 5 + 4: return 3 + std::max(7 - y, 0);
            default: return 0;
        }
    };
    int mix = 0;
    for (int r = 0; r < 7; ++r)
        for (int f = 0; f < 7; ++f) {
            const Bitboard window = (Bitboard(0x303) << (r * 8 + f));          // squares (f, r), (f+1, r), (f, r+1), (f+1, r+1)
            mix += window_score(popcount(by_color_[WHITE] & window), popcount(by_color_[BLACK] & window), r + 1);
        }
    return mix;
}

int Position::game_phase(unsigned num_phases, int definition) const {
    if (definition == 0) {                                     // lichess: three phases whatever num_phases says (the reference only asserts it)
        const int mm = majors_and_minors();
        if (mm <= 6) return 2;
        if (mm <= 10 || backrank_sparse() || mixedness() > 150) return 1;
        return 0;
    }
    if (definition == 1) {                                     // movecount
        if (num_phases <= 1) return 0;
        const double phase_length = std::round(42.85 / double(num_phases));
        const double g = double(size_t(game_ply_ / 2)) / phase_length;        // total_move_cout() = gamePly / 2 (board.cpp:127-130)
        return g > double(num_phases - 1) ? int(num_phases - 1) : int(g);
    }
    return 0;
}

uint64_t Position::perft(int depth) const {
    std::vector<Move> moves;
    legal_moves(moves);
    if (depth <= 1) return depth == 1 ? moves.size() : 1;
    uint64_t n = 0;
    for (Move m : moves) {
        Position p(*this);
        p.do_move(m);
        n += p.perft(depth - 1);
    }
    return n;
}

}  // namespace chess
}  // namespace cra
