#include "policy.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

namespace cra {
namespace chess {
namespace {

// plane_policy_representation.py:34-224
int queen_plane(int dy, int dx) {
    const int len = std::max(std::abs(dx), std::abs(dy)) - 1;
    int dir;
    if (dx == 0 && dy > 0) dir = 0;          // N
    else if (dx > 0 && dy > 0) dir = 1;      // NE
    else if (dx > 0 && dy == 0) dir = 2;     // E
    else if (dy < 0 && dx > 0) dir = 3;      // SE
    else if (dx == 0 && dy < 0) dir = 4;     // S
    else if (dx < 0 && dy < 0) dir = 5;      // SW
    else if (dx < 0 && dy == 0) dir = 6;     // W
    else dir = 7;                            // NW
    return dir * 7 + len;
}
int knight_plane(int dy, int dx) {
    static const int cases[8][2] = {{2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1}, {-1, -2}, {1, -2}, {2, -1}};
    for (int i = 0; i < 8; ++i)
        if (cases[i][0] == dy && cases[i][1] == dx) return 56 + i;
    return -1;
}
int piece_id(char c) {   // python-chess order minus one: P0 N1 B2 R3 Q4 K5
    switch (c) {
        case 'p': case 'P': return 0;
        case 'n': case 'N': return 1;
        case 'b': case 'B': return 2;
        case 'r': case 'R': return 3;
        case 'q': case 'Q': return 4;
        default: return 5;
    }
}

void build(PolicyTables& t, int mode) {
    t.mode = mode;
    const std::string files = "abcdefgh", ranks = "12345678";
    std::vector<std::string> promo_pieces = {"q", "r", "b", "n"};
    if (mode == MODE_LICHESS) promo_pieces.push_back("k");           // outputrepresentation.cpp:70-76
    // classical moves (outputrepresentation.cpp:78-127)
    static const int kfo[8] = {-2, -1, -2, 1, 2, -1, 2, 1}, kro[8] = {-1, -2, 1, -2, -1, 2, 1, 2};
    for (int f = 0; f < 8; ++f)
        for (int r = 0; r < 8; ++r) {
            std::vector<std::pair<int, int>> dest;
            for (int i = 0; i < 8; ++i) dest.push_back({i, r});
            for (int i = 0; i < 8; ++i) dest.push_back({f, i});
            for (int i = -7; i < 8; ++i) dest.push_back({f + i, r + i});
            for (int i = -7; i < 8; ++i) dest.push_back({f + i, r - i});
            for (int i = 0; i < 8; ++i) dest.push_back({f + kfo[i], r + kro[i]});
            for (auto& d : dest) {
                const int f2 = d.first, r2 = d.second;
                if ((f != f2 || r != r2) && f2 >= 0 && f2 < 8 && r2 >= 0 && r2 < 8)
                    t.labels.push_back(std::string{files[f], ranks[r], files[f2], ranks[r2]});
            }
        }
    // promotions (:128-145)
    for (int f = 0; f < 8; ++f)
        for (const std::string& p : promo_pieces) {
            const char fc = files[f];
            t.labels.push_back(std::string{fc, '2', fc, '1'} + p);
            t.labels.push_back(std::string{fc, '7', fc, '8'} + p);
            if (f > 0) {
                t.labels.push_back(std::string{fc, '2', files[f - 1], '1'} + p);
                t.labels.push_back(std::string{fc, '7', files[f - 1], '8'} + p);
            }
            if (f < 7) {
                t.labels.push_back(std::string{fc, '2', files[f + 1], '1'} + p);
                t.labels.push_back(std::string{fc, '7', files[f + 1], '8'} + p);
            }
        }
    // drops (:147-163)
    if (mode != MODE_CHESS) {
        const std::string pcs = "PNBRQ";
        for (int f = 0; f < 8; ++f)
            for (int r = 0; r < 8; ++r)
                for (char pc : pcs)
                    if (pc != 'P' || !(r == 0 || r == 7)) t.labels.push_back(std::string{pc, '@', files[f], ranks[r]});
    }
    const size_t expect = mode == MODE_CRAZYHOUSE ? 2272 : mode == MODE_LICHESS ? 2316 : 1968;   // boardstate.h:51-60
    if (t.labels.size() != expect) throw std::logic_error("policy label count mismatch");
    t.nb_channels_policy_map = mode == MODE_CRAZYHOUSE ? 81 : mode == MODE_LICHESS ? 84 : 76;
    // Parity quirk: the shipped MODE_LICHESS table (policymaprepresentation.h:2314-4631) keeps drops on planes 76..80
    // although king promotions occupy 76..78 (plane_policy_representation.py would put them at 79..83); indices collide
    // ("a2a1k" == "N@a2") exactly as in the reference.
    const int drop_base = 76;

    std::memset(t.normal, 0xFF, sizeof(t.normal));
    std::memset(t.promo, 0xFF, sizeof(t.promo));
    std::memset(t.drop, 0xFF, sizeof(t.drop));
    t.labels_mirrored.resize(t.labels.size());
    t.flat_plane_idx.resize(t.labels.size());
    for (size_t i = 0; i < t.labels.size(); ++i) {
        const std::string& l = t.labels[i];
        std::string m = l;                                                // mirror_move, sfutil.cpp:183-197
        for (char& ch : m)
            if (ch >= '1' && ch <= '8') ch = char('1' + ('8' - ch));
        t.labels_mirrored[i] = m;
        if (l[1] == '@') {
            const int to = (l[3] - '1') * 8 + (l[2] - 'a');
            const int pid = piece_id(l[0]);
            t.drop[pid][to] = int16_t(i);
            t.flat_plane_idx[i] = uint16_t((drop_base + pid) * 64 + to);
            continue;
        }
        const int from = (l[1] - '1') * 8 + (l[0] - 'a'), to = (l[3] - '1') * 8 + (l[2] - 'a');
        const int dy = rank_of(to) - rank_of(from), dx = file_of(to) - file_of(from);
        if (l.size() == 5) {
            const int pid = piece_id(l[4]);                               // N1 B2 R3 Q4 K5
            t.promo[from][to][pid - 1] = int16_t(i);
            t.flat_plane_idx[i] = uint16_t((64 + (pid - 1) * 3 + dx + 1) * 64 + from);
        } else {
            t.normal[from][to] = int16_t(i);
            const int kp = knight_plane(dy, dx);
            t.flat_plane_idx[i] = uint16_t((kp >= 0 ? kp : queen_plane(dy, dx)) * 64 + from);
        }
    }
}

PolicyTables g_tables[3];
std::once_flag g_flags[3];
}  // namespace

const PolicyTables& policy_tables(int mode) {
    if (mode < 0 || mode > 2) throw std::invalid_argument("bad mode");
    std::call_once(g_flags[mode], [mode] { build(g_tables[mode], mode); });
    return g_tables[mode];
}

int label_index(const PolicyTables& t, const Position& pos, Move m, bool mirror) {
    const int flip = mirror ? 56 : 0;
    if (kind_of(m) == DROP) {
        if (piece_of(m) < PAWN || piece_of(m) > QUEEN) return -1;
        return t.drop[piece_of(m) - PAWN][to_sq(m) ^ flip];
    }
    int from, to;
    pos.label_squares(m, from, to);     // castling: classic e1g1 / 960 king-takes-rook (sfutil.cpp:243-285)
    from ^= flip;
    to ^= flip;
    if (kind_of(m) == PROMOTION) return t.promo[from][to][piece_of(m) - KNIGHT];
    return t.normal[from][to];
}

int policy_index(const PolicyTables& t, const Position& pos, Move m, bool is_policy_map) {
    const bool mirror = pos.side_to_move() != WHITE && pos.variant() != V_RACE;
    const int li = label_index(t, pos, m, mirror);
    if (li < 0) return -1;
    return is_policy_map ? t.flat_plane_idx[li] : li;
}

}  // namespace chess
}  // namespace cra
