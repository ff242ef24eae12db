// Host side of the plane builder: descriptor packing and the CPU-callable board_to_planes (same plane_value() as the GPU).
#pragma once
#include "planes.h"
#include "position.h"

namespace cra {
namespace chess {

// move_features: also fill the legal-move fields the chess v2.7 / v2.8 layouts read (layout_needs_move_features); `legal` = the
// position's legal moves when the caller has them already (a search leaf being expanded), else they are generated here
void pack_desc(const Position& pos, BoardDesc& d, bool move_features = false, const std::vector<Move>* legal = nullptr);

// board_to_planes(pos, boardRepetition, normalize, inputPlanes, version) (inputrepresentation.cpp:628-680):
// writes layout_channels(layout)*64 floats, NCHW.  repetitions < 0: use pos.number_repetitions().
void board_to_planes(const Position& pos, int layout, bool normalize, float* out, int repetitions = -1);

}  // namespace chess

// GPU builder: n descriptors (device memory) -> float NCHW planes [n][C][64]
void launch_planes_from_desc(const BoardDesc* d_desc, int n, int layout, int normalize, float* d_planes, void* stream);

}  // namespace cra
