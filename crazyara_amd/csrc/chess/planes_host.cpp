#include "planes_host.h"

#include <cstring>

namespace cra {
namespace chess {

void pack_desc(const Position& pos, BoardDesc& d, bool move_features, const std::vector<Move>* legal) {
    std::memset(&d, 0, sizeof(d));
    if (move_features) {
        std::vector<Move> own;
        if (!legal) {
            pos.legal_moves(own);
            legal = &own;
        }
        for (Move m : *legal)
            if (pos.gives_check(m)) {        // castling counts with the king-takes-rook squares of the fork's Move, as for the last moves
                d.check_from |= sq_bb(from_sq(m));
                d.check_to |= sq_bb(to_sq(m));
            }
        d.mobility = uint8_t(legal->size() > 255 ? 255 : legal->size());
    }
    for (int c = 0; c < 2; ++c)
        for (int pt = PAWN; pt <= KING; ++pt) d.bb[c * 6 + pt - 1] = pos.pieces(Color(c), PieceType(pt));
    d.promoted = pos.promoted_pieces();
    d.checkers = pos.checkers();
    for (int c = 0; c < 2; ++c)
        for (int pt = PAWN; pt <= QUEEN; ++pt) d.pocket[c][pt - 1] = uint8_t(pos.pocket_count(Color(c), PieceType(pt)));
    d.stm = uint8_t(pos.side_to_move());
    d.castling = uint8_t((pos.can_castle(WHITE_OO) ? 1 : 0) | (pos.can_castle(WHITE_OOO) ? 2 : 0) |
                         (pos.can_castle(BLACK_OO) ? 4 : 0) | (pos.can_castle(BLACK_OOO) ? 8 : 0));
    d.ep_square = uint8_t(pos.ep_square());
    d.repetitions = uint8_t(pos.number_repetitions());
    d.is960 = pos.is_chess960() ? 1 : 0;
    d.variant = uint8_t(pos.variant());
    d.checks_given[0] = uint8_t(pos.checks_given(WHITE));
    d.checks_given[1] = uint8_t(pos.checks_given(BLACK));
    const std::vector<Move>& lm = pos.last_moves();
    d.n_last = uint8_t(lm.size() > 8 ? 8 : lm.size());
    for (int i = 0; i < d.n_last; ++i) {
        d.last_from[i] = kind_of(lm[i]) == DROP ? 255 : uint8_t(from_sq(lm[i]));   // castling: king-takes-rook squares, as the fork's Move
        d.last_to[i] = uint8_t(to_sq(lm[i]));
    }
    d.rule50 = uint16_t(pos.rule50_count());
    d.fullmove = uint16_t(pos.game_ply() / 2 + 1);
}

void board_to_planes(const Position& pos, int layout, bool normalize, float* out, int repetitions) {
    BoardDesc d;
    pack_desc(pos, d, layout_needs_move_features(layout));
    if (repetitions >= 0) d.repetitions = uint8_t(repetitions);
    const int C = layout_channels(layout);
    for (int ch = 0; ch < C; ++ch)
        for (int sq = 0; sq < 64; ++sq) out[ch * 64 + sq] = plane_value(d, layout, normalize, ch, sq);
}

}  // namespace chess
}  // namespace cra
