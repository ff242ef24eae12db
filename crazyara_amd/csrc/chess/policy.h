// Policy output map: label list, policy-map plane indices and move -> index lookup.
// Restates engine/src/environments/chess_related/outputrepresentation.cpp:39-184 (labels, lookup construction),
// sfutil.cpp:142-285 (move <-> label rules) and regenerates the FLAT_PLANE_IDX table of policymaprepresentation.h:39-6602
// from its generating formula (DeepCrazyhouse/src/domain/variants/plane_policy_representation.py:22-224).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "planes.h"
#include "position.h"

namespace cra {
namespace chess {

struct PolicyTables {
    int mode = 0;
    std::vector<std::string> labels;          // OutputRepresentation::LABELS
    std::vector<std::string> labels_mirrored; // LABELS_MIRRORED (rank-mirrored strings)
    std::vector<uint16_t> flat_plane_idx;     // FLAT_PLANE_IDX[label]
    int nb_channels_policy_map = 0;           // 81 / 76 / 84 (boardstate.h:246-254)
    int nb_labels() const { return int(labels.size()); }
    int nb_policy_map() const { return nb_channels_policy_map * 64; }
    // label index for (from,to[,promotion piece]) / (drop piece,to); -1 if no such label
    int16_t normal[64][64];
    int16_t promo[64][64][5];                 // KNIGHT..KING -> 0..4 (king only in lichess mode)
    int16_t drop[5][64];                      // PAWN..QUEEN
};

const PolicyTables& policy_tables(int mode);   // built once per mode (StateConstants::init, boardstate.h:98-101)

// Node::set_probabilities_for_moves lookup (engine/src/node.cpp:961-979): index into the NN policy vector for a legal
// move of `pos`.  mirror = pos.side_to_move() != WHITE except racing kings (BoardState::mirror_policy, boardstate.cpp:56-59).
// is_policy_map: MV_LOOKUP holds FLAT_PLANE_IDX[label] for policy-map nets, the label index otherwise.
int policy_index(const PolicyTables& t, const Position& pos, Move m, bool is_policy_map);
int label_index(const PolicyTables& t, const Position& pos, Move m, bool mirror);

}  // namespace chess
}  // namespace cra
