#include "traindata.h"

#include <cmath>

#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "../chess/planes.h"
#include "../chess/planes_host.h"
#include "../chess/policy.h"

namespace cra {
namespace rl {

namespace {
bool exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}
void make_dir(const std::string& p) {
    if (::mkdir(p.c_str(), 0777) != 0 && !exists(p)) throw std::runtime_error("cannot create directory " + p);
}
void write_file(const std::string& p, const void* data, size_t n) {
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + p);
    const size_t w = n ? std::fwrite(data, 1, n, f) : 0;
    std::fclose(f);
    if (w != n) throw std::runtime_error("short write on " + p);
}
}  // namespace

ZarrArray::ZarrArray(const std::string& root, const std::string& name, const std::string& dtype, size_t elem_bytes, std::vector<size_t> shape,
                     size_t chunk_rows)
    : dir_(root + "/" + name), shape_(std::move(shape)), chunk_rows_(chunk_rows) {
    row_bytes_ = elem_bytes;
    for (size_t i = 1; i < shape_.size(); ++i) row_bytes_ *= shape_[i];
    make_dir(dir_);
    const std::string meta = dir_ + "/.zarray";
    if (exists(meta)) return;
    std::string shp, chk;
    for (size_t i = 0; i < shape_.size(); ++i) {
        shp += (i ? ", " : "") + std::to_string(shape_[i]);
        chk += (i ? ", " : "") + std::to_string(i ? shape_[i] : chunk_rows_);
    }
    const std::string json = "{\n \"zarr_format\": 2,\n \"shape\": [" + shp + "],\n \"chunks\": [" + chk + "],\n \"dtype\": \"" + dtype +
                             "\",\n \"compressor\": null,\n \"fill_value\": 0,\n \"order\": \"C\",\n \"filters\": null\n}\n";
    write_file(meta, json.data(), json.size());
}

void ZarrArray::write_rows(size_t start, const void* data, size_t n) const {
    const char* src = static_cast<const char*>(data);
    std::vector<char> chunk(chunk_rows_ * row_bytes_);
    size_t pos = 0;
    while (pos < n) {
        const size_t row = start + pos, ci = row / chunk_rows_, off = row % chunk_rows_;
        const size_t take = std::min(chunk_rows_ - off, n - pos);
        std::string path = dir_ + "/" + std::to_string(ci);
        for (size_t i = 1; i < shape_.size(); ++i) path += ".0";
        bool loaded = false;
        if (off != 0 || take != chunk_rows_) {                         // partial chunk: keep what earlier games wrote
            if (FILE* f = std::fopen(path.c_str(), "rb")) {
                loaded = std::fread(chunk.data(), 1, chunk.size(), f) == chunk.size();
                std::fclose(f);
            }
        }
        if (!loaded) std::memset(chunk.data(), 0, chunk.size());       // fill_value 0
        std::memcpy(chunk.data() + off * row_bytes_, src + pos * row_bytes_, take * row_bytes_);
        write_file(path, chunk.data(), chunk.size());
        pos += take;
    }
}

TrainDataExporter::TrainDataExporter(const std::string& file_name, int mode, int version_major, int version_minor, size_t number_chunks,
                                     size_t chunk_size)
    : mode_(mode), number_chunks_(number_chunks), chunk_size_(chunk_size), number_samples_(number_chunks * chunk_size) {
    if (number_chunks == 0 || chunk_size == 0) throw std::invalid_argument("TrainDataExporter: empty data set");
    layout_ = layout_for(mode, version_major, version_minor);
    channels_ = layout_channels(layout_);
    nb_labels_ = chess::policy_tables(mode).nb_labels();
    // "Export file already exists. It will be overwritten" (:149-152): the arrays are reopened and written from sample 0 again
    make_dir(file_name);
    const std::string zgroup = file_name + "/.zgroup";
    if (!exists(zgroup)) {
        const std::string g = "{\n \"zarr_format\": 2\n}\n";
        write_file(zgroup, g.data(), g.size());
    }
    const size_t n = number_samples_, c = chunk_size_;
    d_start_ = ZarrArray(file_name, "start_indices", "<i4", 4, {n}, c);
    d_x_ = ZarrArray(file_name, "x", "<i2", 2, {n, size_t(channels_), 8, 8}, c);
    d_value_ = ZarrArray(file_name, "y_value", "<i2", 2, {n}, c);
    d_policy_ = ZarrArray(file_name, "y_policy", "<f4", 4, {n, size_t(nb_labels_)}, c);
    d_best_q_ = ZarrArray(file_name, "y_best_move_q", "<f4", 4, {n}, c);
    d_plys_ = ZarrArray(file_name, "plys_to_end", "<i2", 2, {n}, c);
    d_phase_ = ZarrArray(file_name, "phase_vector", "<i2", 2, {n}, c);
    save_start_idx();
}

void TrainDataExporter::new_game() {
    first_move_ = true;
    cur_sample_idx_ = 0;
    game_x_.clear();
    game_value_.clear();
    game_plys_.clear();
    game_phase_.clear();
    game_policy_.clear();
    game_best_q_.clear();
}

void TrainDataExporter::set_phases(int num_phases, int game_phase_definition) {
    if (num_phases < 1) throw std::invalid_argument("number of game phases must be at least 1");
    if (game_phase_definition != 0 && game_phase_definition != 1) throw std::invalid_argument("game phase definition: 0 lichess, 1 movecount");
    num_phases_ = num_phases;
    game_phase_definition_ = game_phase_definition;
}

void TrainDataExporter::save_sample(const chess::Position& pos, const std::vector<chess::Move>& legal_moves, const double* policy,
                                    size_t n_policy, float best_move_q, int phase) {
    if (start_idx_ + cur_sample_idx_ >= number_samples_) return;      // "Extended number of maximum samples"
    if (first_move_) new_game();
    if (phase < 0) phase = pos.game_phase(unsigned(num_phases_), game_phase_definition_);       // save_cur_phase, :91-103
    // save_planes: float planes of the un-normalised representation, truncated to int16
    std::vector<float> planes(size_t(channels_) * 64);
    chess::board_to_planes(pos, layout_, false, planes.data());
    for (float v : planes) game_x_.push_back(int16_t(v));
    // save_policy: StateConstants::action_to_index<classic, mirrored / notMirrored>
    const chess::PolicyTables& t = chess::policy_tables(mode_);
    const bool mirror = pos.side_to_move() != chess::WHITE && pos.variant() != chess::V_RACE;   // pos->mirror_policy(side_to_move)
    const size_t base = game_policy_.size();
    game_policy_.resize(base + size_t(nb_labels_), 0.0f);
    for (size_t i = 0; i < legal_moves.size(); ++i) {
        const int li = chess::label_index(t, pos, legal_moves[i], mirror);
        if (li < 0) throw std::logic_error("legal move without a policy label: " + pos.move_to_uci(legal_moves[i]));
        if (i < n_policy && !std::isfinite(policy[i])) throw std::invalid_argument("training sample with a non-finite policy entry");
        game_policy_[base + size_t(li)] = i < n_policy ? float(policy[i]) : 0.0f;
    }
    game_best_q_.push_back(best_move_q);
    game_value_.push_back(int16_t(-(int(pos.side_to_move()) * 2 - 1)));           // save_side_to_move: -(col * 2 - 1)
    game_plys_.push_back(int16_t(cur_sample_idx_));                               // save_cur_sample_index
    game_phase_.push_back(int16_t(phase));                                        // save_cur_phase
    ++cur_sample_idx_;
    first_move_ = false;
}

size_t TrainDataExporter::export_game_samples(int result) {
    if (cur_sample_idx_ == 0) return 0;                                // "No samples have been recorded, skip export."
    if (start_idx_ >= number_samples_) return 0;                       // "Extended number of maximum samples"
    // apply_result_to_value (:287-296), apply_result_to_plys_to_end (:298-302)
    for (int16_t& v : game_value_) v = result == BLACK_WIN ? int16_t(-v) : result == DRAWN ? int16_t(0) : v;
    for (int16_t& p : game_plys_) p = int16_t(-(p - int16_t(cur_sample_idx_)));
    const size_t n = cur_sample_idx_;
    d_x_.write_rows(start_idx_, game_x_.data(), n);
    d_value_.write_rows(start_idx_, game_value_.data(), n);
    d_best_q_.write_rows(start_idx_, game_best_q_.data(), n);
    d_policy_.write_rows(start_idx_, game_policy_.data(), n);
    d_plys_.write_rows(start_idx_, game_plys_.data(), n);
    d_phase_.write_rows(start_idx_, game_phase_.data(), n);
    start_idx_ += n;
    ++game_idx_;
    save_start_idx();
    new_game();
    return n;
}

void TrainDataExporter::save_start_idx() {
    if (game_idx_ >= number_samples_) return;
    const int32_t v = int32_t(start_idx_);
    d_start_.write_rows(game_idx_, &v, 1);
}

}  // namespace rl
}  // namespace cra
