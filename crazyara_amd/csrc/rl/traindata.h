// Training-sample exporter of the self-play loop: restates engine/src/rl/traindataexporter.{h,cpp}.
//
// One zarr (format 2) group -- what z5::createFile(file, /*zarr*/ true) + z5::createDataset produce with their default "raw"
// compressor (traindataexporter.cpp:262-283) -- with the arrays
//     start_indices  int32   [N]              sample index at which game g starts                                (:226-233)
//     x              int16   [N][C][8][8]     un-normalised input planes, get_state_planes(normalize = false)    (:175-196)
//     y_value        int16   [N]              +1 / -1 / 0: the game result seen from the side to move            (:66-78, 287-296)
//     y_policy       float32 [N][NB_LABELS]   the MCTS policy scattered to the CLASSIC label index, mirrored for Black (:198-224)
//     y_best_move_q  float32 [N]              EvalInfo::bestMoveQ[0]                                              (:49-63)
//     plys_to_end    int16   [N]              plies from the sample to the end of its game                        (:80-93, 298-302)
//     phase_vector   int16   [N]              game phase of the sample                                            (:95-108)
// N = numberChunks * chunkSize, chunked [chunkSize] along N.  Chunks are plain little-endian C-order blocks ("compressor": null),
// chunk file names "i.0.0.0" -- the layout every zarr-v2 reader opens (tests/zarr_v2_reader.py reads it from the published spec).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../chess/position.h"

namespace cra {
namespace rl {

enum Result : int { DRAWN = 0, WHITE_WIN = 1, BLACK_WIN = 2 };      // engine/src/state.h

class ZarrArray {
public:
    ZarrArray() = default;
    // creates <root>/<name>/.zarray unless the array directory already holds one (open_dataset_from_file, :236-245)
    ZarrArray(const std::string& root, const std::string& name, const std::string& dtype, size_t elem_bytes, std::vector<size_t> shape,
              size_t chunk_rows);
    // rows [start, start + n) of the first axis; data = n * row_bytes() bytes (writeSubarray with offset {start, 0, ...})
    void write_rows(size_t start, const void* data, size_t n) const;
    size_t row_bytes() const { return row_bytes_; }

private:
    std::string dir_;
    std::vector<size_t> shape_;
    size_t chunk_rows_ = 0, row_bytes_ = 0;
};

class TrainDataExporter {
public:
    // TrainDataExporter(fileName, numPhases, gamePhaseDefinition, numberChunks = 200, chunkSize = 128) (:136-156); mode / version fix
    // the plane layout and the label set the reference gets from its build flavour (StateConstants)
    TrainDataExporter(const std::string& file_name, int mode, int version_major, int version_minor, size_t number_chunks = 200,
                      size_t chunk_size = 128);
    // save_sample(pos, eval) (:33-47): planes, policy over `legal_moves` (policy[i] for i < n_policy, 0 beyond: EvalInfo pads moves the
    // search never expanded), bestMoveQ, side to move, running sample index, phase
    // phase < 0 (default): save_cur_phase (:91-103) = pos->get_phase(numPhases, gamePhaseDefinition) with the exporter's own two
    // settings, as the reference's exporter holds them (set_phases; 1 phase / lichess definition unless told otherwise)
    void save_sample(const chess::Position& pos, const std::vector<chess::Move>& legal_moves, const double* policy, size_t n_policy,
                     float best_move_q, int phase = -1);
    void set_phases(int num_phases, int game_phase_definition);       // TrainDataExporter(fileName, numPhases, gamePhaseDefinition, ...) (:136)
    // export_game_samples(result) (:110-134): applies the result to the values, turns the sample indices into plies-to-end, writes the
    // game's rows behind the previous games' and records the next start index.  Returns the number of samples written.
    size_t export_game_samples(int result);
    void new_game();                                                   // :167-171
    bool is_file_full() const { return start_idx_ >= number_samples_; }   // :162-165
    size_t get_number_samples() const { return number_samples_; }
    size_t start_index() const { return start_idx_; }
    size_t game_index() const { return game_idx_; }
    int nb_labels() const { return nb_labels_; }
    int channels() const { return channels_; }

private:
    void save_start_idx();
    int mode_, layout_, channels_, nb_labels_;
    int num_phases_ = 1, game_phase_definition_ = 0;
    size_t number_chunks_, chunk_size_, number_samples_;
    bool first_move_ = true;
    size_t game_idx_ = 0, start_idx_ = 0, cur_sample_idx_ = 0;
    ZarrArray d_start_, d_x_, d_value_, d_policy_, d_best_q_, d_plys_, d_phase_;
    std::vector<int16_t> game_x_, game_value_, game_plys_, game_phase_;
    std::vector<float> game_policy_, game_best_q_;
};

}  // namespace rl
}  // namespace cra
