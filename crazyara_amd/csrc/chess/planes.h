// Input-plane builder shared by host and device (one definition, compiled for both).
//
// Restates board_to_planes() and its layout builders (engine/src/environments/chess_related/inputrepresentation.cpp:
// 33-109 helpers, 112-417 plane setters, 426-624 layouts, 628-680 dispatch) as a pure function
//      value = plane_value(desc, layout, normalize, channel, square)
// over a compact 192-byte board descriptor, so that the GPU expands descriptors straight into the batch tensor
// (coalesced stores) instead of receiving 8-20 KB of mostly-zero floats per position over PCIe.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CRA_HD __host__ __device__ __forceinline__
#else
#define CRA_HD inline
#endif

namespace cra {

// Build mode of the reference binary (engine/CMakeLists.txt:28-64): fixes label set, normalisers and layout family.
enum Mode : int { MODE_CRAZYHOUSE = 0, MODE_CHESS = 1, MODE_LICHESS = 2 };

// (mode, input-representation major version) -> layout; version 0/1 (and 2 for lichess) are the default layout
enum PlaneLayout : int {
    LAYOUT_CZ_V1 = 0,      // 34 ch  default_board_to_planes, MODE_CRAZYHOUSE      (:426-501)
    LAYOUT_CZ_V2 = 1,      // 51 ch  board_to_planes_crazyhouse_v2                  (:579-595)
    LAYOUT_CZ_V3 = 2,      // 64 ch  board_to_planes_crazyhouse_v3                  (:569-577)
    LAYOUT_CHESS_V1 = 3,   // 39 ch  default_board_to_planes, MODE_CHESS
    LAYOUT_CHESS_V3 = 4,   // 52 ch  board_to_planes_chess_v3                       (:536-566)
    LAYOUT_LICHESS_V2 = 5, // 63 ch  default_board_to_planes, MODE_LICHESS (v1/v2)
    LAYOUT_LICHESS_V3 = 6, // 80 ch  board_to_planes_lichess_v3                     (:599-624)
    LAYOUT_CHESS_V27 = 7,  // 33 ch  board_to_planes_chess_v_2_7                     (:503-524)  needs the move features of the descriptor
    LAYOUT_CHESS_V28 = 8,  // 38 ch  board_to_planes_chess_v_2_8                     (:526-533)  v2.7 + material count
    LAYOUT_NB = 9
};

CRA_HD int layout_channels(int layout) {
    switch (layout) {
        case LAYOUT_CZ_V1: return 34;
        case LAYOUT_CZ_V2: return 51;
        case LAYOUT_CZ_V3: return 64;
        case LAYOUT_CHESS_V1: return 39;
        case LAYOUT_CHESS_V3: return 52;
        case LAYOUT_LICHESS_V2: return 63;
        case LAYOUT_LICHESS_V3: return 80;
        case LAYOUT_CHESS_V27: return 33;
        case LAYOUT_CHESS_V28: return 38;
    }
    return 0;
}

// layouts whose planes depend on the legal moves of the position (check-giving moves, mobility): pack_desc fills those
// descriptor fields only on request, they cost a gives_check() per legal move
CRA_HD bool layout_needs_move_features(int layout) { return layout == LAYOUT_CHESS_V27 || layout == LAYOUT_CHESS_V28; }

// dispatch of board_to_planes (:628-680); chess 2.x: make_version<2,7,0> / <2,8,0>
inline int layout_for(int mode, int version_major, int version_minor = 0) {
    switch (mode) {
        case MODE_CRAZYHOUSE: return version_major == 2 ? LAYOUT_CZ_V2 : version_major == 3 ? LAYOUT_CZ_V3 : LAYOUT_CZ_V1;
        case MODE_CHESS:
            if (version_major == 2) return version_minor == 8 ? LAYOUT_CHESS_V28 : LAYOUT_CHESS_V27;
            return version_major == 3 ? LAYOUT_CHESS_V3 : LAYOUT_CHESS_V1;
        default: return version_major == 3 ? LAYOUT_LICHESS_V3 : LAYOUT_LICHESS_V2;
    }
}

struct BoardDesc {               // 192 bytes, 8-byte aligned; absolute colours, a1 = bit 0
    uint64_t bb[12];             // white P N B R Q K, black P N B R Q K
    uint64_t promoted;           // crazyhouse promoted-piece mask
    uint64_t checkers;           // pieces giving check to the side to move
    uint8_t pocket[2][5];        // [colour][P N B R Q]
    uint8_t stm;                 // 0 white, 1 black
    uint8_t castling;            // bit0 WHITE_OO, bit1 WHITE_OOO, bit2 BLACK_OO, bit3 BLACK_OOO
    uint8_t ep_square;           // 64 = none
    uint8_t repetitions;         // Board::number_repetitions(): 0 or 1 from the engine (2 only via the test/data-export API)
    uint8_t is960;
    uint8_t variant;             // chess::Variant
    uint8_t checks_given[2];     // [colour]
    uint8_t n_last;              // number of valid last moves (<= 8), most recent first
    uint8_t last_from[8];        // 255 = drop (its "from" plane is skipped, inputrepresentation.cpp:272-277)
    uint8_t last_to[8];
    uint8_t pad0;
    uint16_t rule50;
    uint16_t fullmove;           // game_ply / 2 + 1
    // move features (chess v2.7 / v2.8 only; zero unless requested from pack_desc): origin / destination squares of the legal
    // moves that give check (set_check_moves, :382-393) and the number of legal moves (set_mobility, :395-398)
    uint64_t check_from;
    uint64_t check_to;
    uint8_t mobility;
    uint8_t pad1[23];
};
static_assert(sizeof(BoardDesc) == 192, "BoardDesc must be 192 bytes");

namespace planes_detail {

CRA_HD int popc(uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(b);
#else
    return __builtin_popcountll(b);
#endif
}
CRA_HD uint64_t bswap(uint64_t b) { return __builtin_bswap64(b); }   // flip_vertical, sfutil.cpp:178-181

struct Ctx {
    const BoardDesc* d;
    bool flip;        // flip_board(): side to move is black, except racing kings (inputrepresentation.h:58-66)
    bool normalize;
    int me, you;
    int sq;           // output square (NCHW index inside the plane)
};

CRA_HD float bit_plane(const Ctx& c, uint64_t bb) {
    if (c.flip) bb = bswap(bb);
    return float((bb >> c.sq) & 1ull);
}
CRA_HD float single_square(const Ctx& c, int target, float v = 1.0f) {
    if (target >= 64) return 0.0f;
    return ((c.flip ? (target ^ 56) : target) == c.sq) ? v : 0.0f;
}
CRA_HD uint64_t color_bb(const BoardDesc& d, int col) {
    const uint64_t* p = d.bb + col * 6;
    return p[0] | p[1] | p[2] | p[3] | p[4] | p[5];
}

// --- plane groups; `k` is the channel index inside the group ---
CRA_HD float g_pieces(const Ctx& c, int k) {           // :112-122  me{P,N,B,R,Q,K}, you{...}
    const int col = k < 6 ? c.me : c.you;
    return bit_plane(c, c.d->bb[col * 6 + (k % 6)]);
}
CRA_HD float g_repetition(const Ctx& c, int k) {       // :124-136
    return c.d->repetitions >= k + 1 ? 1.0f : 0.0f;
}
CRA_HD float g_pockets(const Ctx& c, int k, float max_prisoners) {   // :139-151
    const int col = k < 5 ? c.me : c.you;
    const int cnt = c.d->pocket[col][k % 5];
    if (cnt <= 0) return 0.0f;
    return c.normalize ? cnt / max_prisoners : float(cnt);
}
CRA_HD float g_promoted(const Ctx& c, int k) {         // :153-157
    return bit_plane(c, c.d->promoted & color_bb(*c.d, k == 0 ? c.me : c.you));
}
CRA_HD float g_ep(const Ctx& c) { return single_square(c, c.d->ep_square); }                 // :160-166
CRA_HD float g_color(const Ctx& c) { return c.me == 0 ? 1.0f : 0.0f; }                        // :168-175
CRA_HD float g_total_moves(const Ctx& c) {                                                   // :177-181
    return c.normalize ? c.d->fullmove / 500.0f : float(c.d->fullmove);
}
CRA_HD float g_castling(const Ctx& c, int k) {         // :183-221  me-OO, me-OOO, you-OO, you-OOO
    const int col = k < 2 ? c.me : c.you;
    return (c.d->castling >> (col * 2 + (k & 1))) & 1 ? 1.0f : 0.0f;
}
CRA_HD float g_no_progress(const Ctx& c, float max_no_progress) {                            // :223-226
    return c.normalize ? c.d->rule50 / max_no_progress : float(c.d->rule50);
}
CRA_HD float g_remaining_checks(const Ctx& c, int k) {  // :229-247
    if (c.d->variant != 3) return 0.0f;                 // V_THREECHECK
    const int given = c.d->checks_given[k < 2 ? c.me : c.you];
    return given >= (k & 1) + 1 ? 1.0f : 0.0f;
}
CRA_HD float g_variant_and_960(const Ctx& c, int k) {   // :251-263, CHANNEL_MAPPING_VARIANTS boardstate.h:283-294
    if (k == 0) return c.d->is960 ? 1.0f : 0.0f;
    int slot;
    switch (c.d->variant) {        // chess 1, crazyhouse 2, koth 3, 3check 4, anti 5, atomic 6, horde 7, race 8
        case 0: slot = 1; break;
        case 1: slot = 2; break;
        case 2: slot = 3; break;
        case 3: slot = 4; break;
        case 4: slot = 5; break;
        case 5: slot = 6; break;
        case 6: slot = 7; break;
        default: slot = 8; break;
    }
    return k == slot ? 1.0f : 0.0f;
}
CRA_HD float g_last_moves(const Ctx& c, int k) {        // :266-282  plane 2i = from (skipped for drops), 2i+1 = to
    const int i = k >> 1;
    if (i >= c.d->n_last) return 0.0f;
    if (k & 1) return single_square(c, c.d->last_to[i]);
    return c.d->last_from[i] == 255 ? 0.0f : single_square(c, c.d->last_from[i]);
}
CRA_HD float g_is960(const Ctx& c) { return c.d->is960 ? 1.0f : 0.0f; }                       // :284-290
CRA_HD float g_piece_masks(const Ctx& c, int k) { return bit_plane(c, color_bb(*c.d, k == 0 ? c.me : c.you)); }   // :292-300
CRA_HD float g_checkerboard(const Ctx& c) {             // :302-314, written un-flipped for both sides
    return (((c.sq & 7) + (c.sq >> 3)) & 1) ? 1.0f : 0.0f;
}
CRA_HD float rel_count(const Ctx& c, int cnt) {          // set_single_relative_count :316-322
    if (cnt == 0) return 0.0f;
    return c.normalize ? float(cnt) / 8.0f : float(cnt);
}
CRA_HD float g_material_diff(const Ctx& c, int k) {      // :324-345 (k = P,N,B,R,Q[,K])
    return rel_count(c, popc(c.d->bb[c.me * 6 + k]) - popc(c.d->bb[c.you * 6 + k]));
}
CRA_HD float g_opposite_bishops(const Ctx& c) {          // :401-406, Position::opposite_bishops()
    const uint64_t wb = c.d->bb[2], bb = c.d->bb[8];
    if (popc(wb) != 1 || popc(bb) != 1) return 0.0f;
    const uint64_t dark = 0xAA55AA55AA55AA55ull;
    return ((wb & dark) != 0) != ((bb & dark) != 0) ? 1.0f : 0.0f;
}
CRA_HD float g_checkers(const Ctx& c) { return bit_plane(c, c.d->checkers); }                 // :376-379
CRA_HD float g_material_count(const Ctx& c, int k) { return rel_count(c, popc(c.d->bb[c.me * 6 + k])); }   // :407-424
CRA_HD float g_check_moves(const Ctx& c, int k) { return bit_plane(c, k == 0 ? c.d->check_from : c.d->check_to); }   // :382-393
CRA_HD float g_mobility(const Ctx& c) {                                                        // :395-398, NORMALIZE_MOBILITY 64
    return c.normalize ? c.d->mobility / 64.0f : float(c.d->mobility);
}

}  // namespace planes_detail

// value of input plane `ch` at NCHW square `sq` for the given layout
CRA_HD float plane_value(const BoardDesc& d, int layout, bool normalize, int ch, int sq) {
    using namespace planes_detail;
    Ctx c;
    c.d = &d;
    c.flip = d.stm == 1 && d.variant != 7;   // racing kings is never flipped
    c.normalize = normalize;
    c.me = d.stm;
    c.you = d.stm ^ 1;
    c.sq = sq;
    switch (layout) {
        case LAYOUT_CZ_V1:
        case LAYOUT_CZ_V2: {
            if (ch < 12) return g_pieces(c, ch);
            if (ch < 14) return g_repetition(c, ch - 12);
            if (ch < 24) return g_pockets(c, ch - 14, 32.0f);
            if (ch < 26) return g_promoted(c, ch - 24);
            if (ch == 26) return g_ep(c);
            if (ch == 27) return g_color(c);
            if (ch == 28) return g_total_moves(c);
            if (ch < 33) return g_castling(c, ch - 29);
            if (ch == 33) return g_no_progress(c, 40.0f);
            if (ch == 34) return g_is960(c);
            return g_last_moves(c, ch - 35);
        }
        case LAYOUT_CHESS_V1: {
            if (ch < 12) return g_pieces(c, ch);
            if (ch < 14) return g_repetition(c, ch - 12);
            if (ch == 14) return g_ep(c);
            if (ch == 15) return g_color(c);
            if (ch == 16) return g_total_moves(c);
            if (ch < 21) return g_castling(c, ch - 17);
            if (ch == 21) return g_no_progress(c, 50.0f);
            if (ch == 22) return g_is960(c);
            return g_last_moves(c, ch - 23);
        }
        case LAYOUT_CHESS_V3:
        case LAYOUT_CZ_V3: {
            if (ch < 12) return g_pieces(c, ch);
            if (ch < 14) return g_repetition(c, ch - 12);
            if (ch == 14) return g_ep(c);
            if (ch < 19) return g_castling(c, ch - 15);
            if (ch == 19) return g_no_progress(c, layout == LAYOUT_CZ_V3 ? 40.0f : 50.0f);
            if (ch < 36) return g_last_moves(c, ch - 20);
            if (ch == 36) return g_is960(c);
            if (ch < 39) return g_piece_masks(c, ch - 37);
            if (ch == 39) return g_checkerboard(c);
            if (ch < 45) return g_material_diff(c, ch - 40);
            if (ch == 45) return g_opposite_bishops(c);
            if (ch == 46) return g_checkers(c);
            if (ch < 52) return g_material_count(c, ch - 47);
            if (ch < 62) return g_pockets(c, ch - 52, 32.0f);
            return g_promoted(c, ch - 62);
        }
        case LAYOUT_CHESS_V27:
        case LAYOUT_CHESS_V28: {           // no repetition, colour, move-count or no-progress planes; one move of history
            if (ch < 12) return g_pieces(c, ch);
            if (ch == 12) return g_ep(c);
            if (ch < 17) return g_castling(c, ch - 13);
            if (ch < 19) return g_last_moves(c, ch - 17);
            if (ch == 19) return g_is960(c);
            if (ch < 22) return g_piece_masks(c, ch - 20);
            if (ch == 22) return g_checkerboard(c);
            if (ch < 28) return g_material_diff(c, ch - 23);
            if (ch == 28) return g_opposite_bishops(c);
            if (ch == 29) return g_checkers(c);
            if (ch < 32) return g_check_moves(c, ch - 30);
            if (ch == 32) return g_mobility(c);
            return g_material_count(c, ch - 33);
        }
        case LAYOUT_LICHESS_V2:
        case LAYOUT_LICHESS_V3: {
            if (ch < 12) return g_pieces(c, ch);
            if (ch < 14) return g_repetition(c, ch - 12);
            if (ch < 24) return g_pockets(c, ch - 14, 16.0f);
            if (ch < 26) return g_promoted(c, ch - 24);
            if (ch == 26) return g_ep(c);
            if (ch == 27) return layout == LAYOUT_LICHESS_V3 ? 0.0f : g_color(c);         // v3 skips colour / move count (:609-610)
            if (ch == 28) return layout == LAYOUT_LICHESS_V3 ? 0.0f : g_total_moves(c);
            if (ch < 33) return g_castling(c, ch - 29);
            if (ch == 33) return g_no_progress(c, 50.0f);
            if (ch < 38) return g_remaining_checks(c, ch - 34);
            if (ch < 47) return g_variant_and_960(c, ch - 38);
            if (ch < 63) return g_last_moves(c, ch - 47);
            if (ch < 65) return g_piece_masks(c, ch - 63);
            if (ch == 65) return g_checkerboard(c);
            if (ch < 72) return g_material_diff(c, ch - 66);      // with king
            if (ch == 72) return g_opposite_bishops(c);
            if (ch == 73) return g_checkers(c);
            return g_material_count(c, ch - 74);                  // with king
        }
    }
    return 0.0f;
}

}  // namespace cra
