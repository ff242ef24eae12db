// C ABI: environment, input planes, policy map -- see include/crazyara_hip.h.
#include "../../include/crazyara_hip.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "capi_common.h"
#include "capi_traindata.h"
#include "chess/planes_host.h"
#include "chess/policy.h"
#include "chess/position.h"
#include "nn/rise_net.h"

using namespace cra;
using namespace cra::chess;

struct mi_pos {
    Position pos;
};

namespace {
int copy_out(const std::string& s, char* buf, int cap) {
    if (!buf || cap <= int(s.size())) { cra_set_error("buffer too small"); return -1; }
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return int(s.size());
}
}  // namespace

extern "C" {

mi_pos* mi_pos_create(const char* fen, int is_chess960, const char* variant) {
    mi_pos* p = nullptr;
    if (cra_guard([&] {
            const Variant v = variant_from_name(variant && *variant ? variant : "chess");
            p = new mi_pos;
            p->pos.set(fen && *fen ? std::string(fen) : start_fen(v), is_chess960 != 0, v);
        })) {
        delete p;
        return nullptr;
    }
    return p;
}
mi_pos* mi_pos_clone(const mi_pos* pos) { return pos ? new mi_pos(*pos) : nullptr; }
void mi_pos_destroy(mi_pos* pos) { delete pos; }
int mi_pos_fen(const mi_pos* pos, char* buf, int cap) { return pos ? copy_out(pos->pos.fen(), buf, cap) : -1; }
int mi_pos_side_to_move(const mi_pos* pos) { return pos ? int(pos->pos.side_to_move()) : -1; }
int mi_pos_legal_moves(const mi_pos* pos, uint32_t* moves, int cap) {
    if (!pos) return -1;
    std::vector<Move> v;
    pos->pos.legal_moves(v);
    if (moves)
        for (int i = 0; i < int(v.size()) && i < cap; ++i) moves[i] = v[i];
    return int(v.size());
}
uint32_t mi_pos_uci_to_move(const mi_pos* pos, const char* uci) { return pos && uci ? pos->pos.uci_to_move(uci) : 0; }
int mi_pos_move_to_uci(const mi_pos* pos, uint32_t move, char* buf, int cap) {
    return pos ? copy_out(pos->pos.move_to_uci(move), buf, cap) : -1;
}
int mi_pos_do_move(mi_pos* pos, uint32_t move) {
    if (!pos || !move) { cra_set_error("null position or move"); return 1; }
    return cra_guard([&] { pos->pos.do_move(move); });
}
int mi_pos_terminal(const mi_pos* pos) {
    if (!pos) return -1;
    std::vector<Move> v;
    pos->pos.legal_moves(v);
    return int(pos->pos.is_terminal(v.size()));
}
int mi_pos_move_to_san(const mi_pos* pos, uint32_t move, char* buf, int cap) {
    if (!pos || !buf) return -1;
    int n = -1;
    cra_guard([&] {
        const std::string s = pos->pos.move_to_san(move);
        if (int(s.size()) + 1 > cap) throw std::invalid_argument("buffer too small");
        std::memcpy(buf, s.c_str(), s.size() + 1);
        n = int(s.size());
    });
    return n;
}

int mi_pos_game_phase(const mi_pos* pos, int num_phases, int definition) {
    if (!pos || num_phases < 1) return -1;
    int phase = -1;
    cra_guard([&] { phase = pos->pos.game_phase(unsigned(num_phases), definition); });
    return phase;
}
int mi_pos_insufficient_material(const mi_pos* pos) { return pos && pos->pos.draw_by_insufficient_material() ? 1 : 0; }
int mi_pos_plies_from_null(const mi_pos* pos) { return pos ? pos->pos.plies_from_null() : 0; }
int mi_pos_in_check(const mi_pos* pos) { return pos && pos->pos.checkers() != 0 ? 1 : 0; }

int mi_pos_number_repetitions(const mi_pos* pos) { return pos ? pos->pos.number_repetitions() : -1; }
unsigned long long mi_pos_perft(const mi_pos* pos, int depth) { return pos ? pos->pos.perft(depth) : 0; }
const char* mi_chess960_start_fen(int idx) {
    thread_local std::string s;
    s = chess960_start_fen(idx);
    return s.c_str();
}

int mi_planes_layout(int mode, int version_major) { return layout_for(mode, version_major); }
int mi_planes_layout_minor(int mode, int version_major, int version_minor) { return layout_for(mode, version_major, version_minor); }
int mi_planes_channels(int layout) { return layout_channels(layout); }
int mi_pos_planes(const mi_pos* pos, int layout, int normalize, int repetitions, float* out) {
    if (!pos || !out || layout_channels(layout) == 0) { cra_set_error("bad argument to mi_pos_planes"); return 1; }
    board_to_planes(pos->pos, layout, normalize != 0, out, repetitions);
    return 0;
}
int mi_pos_desc(const mi_pos* pos, void* desc192) {
    if (!pos || !desc192) { cra_set_error("null argument"); return 1; }
    pack_desc(pos->pos, *static_cast<BoardDesc*>(desc192));
    return 0;
}
int mi_pos_desc_for(const mi_pos* pos, int layout, void* desc192) {
    if (!pos || !desc192 || layout_channels(layout) == 0) { cra_set_error("bad argument to mi_pos_desc_for"); return 1; }
    pack_desc(pos->pos, *static_cast<BoardDesc*>(desc192), layout_needs_move_features(layout));
    return 0;
}
int mi_planes_from_descs_host(const void* descs, int n, int layout, int normalize, float* out) {
    if (!descs || !out || n < 0) { cra_set_error("bad argument"); return 1; }
    return cra_guard([&] {
        const int C = layout_channels(layout);
        if (C <= 0) throw std::invalid_argument("unknown plane layout");
        const BoardDesc* d = static_cast<const BoardDesc*>(descs);
        for (int b = 0; b < n; ++b)
            for (int i = 0; i < C * 64; ++i) out[size_t(b) * C * 64 + i] = plane_value(d[b], layout, normalize != 0, i >> 6, i & 63);
    });
}

int mi_planes_from_descs_device(const void* descs_host, int n, int layout, int normalize, float* d_planes, int device_id) {
    if (!descs_host || !d_planes || n <= 0 || layout_channels(layout) == 0) { cra_set_error("bad argument"); return 1; }
    return cra_guard([&] {
        auto ck = [](hipError_t e, const char* what) {
            if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
        };
        ck(hipSetDevice(device_id), "hipSetDevice");
        void* d = nullptr;
        ck(hipMalloc(&d, size_t(n) * sizeof(BoardDesc)), "hipMalloc");
        hipError_t e = hipMemcpy(d, descs_host, size_t(n) * sizeof(BoardDesc), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            launch_planes_from_desc(static_cast<const BoardDesc*>(d), n, layout, normalize, d_planes, nullptr);
            e = hipDeviceSynchronize();
        }
        (void)hipFree(d);
        ck(e, "planes_from_desc");
    });
}

int mi_policy_nb_labels(int mode) {
    int n = -1;
    cra_guard([&] { n = policy_tables(mode).nb_labels(); });
    return n;
}
int mi_policy_nb_policy_map(int mode) {
    int n = -1;
    cra_guard([&] { n = policy_tables(mode).nb_policy_map(); });
    return n;
}
const char* mi_policy_label(int mode, int idx, int mirrored) {
    const char* r = nullptr;
    cra_guard([&] {
        const PolicyTables& t = policy_tables(mode);
        r = (mirrored ? t.labels_mirrored : t.labels).at(idx).c_str();
    });
    return r;
}
int mi_policy_flat_plane_idx(int mode, int idx) {
    int n = -1;
    cra_guard([&] { n = policy_tables(mode).flat_plane_idx.at(idx); });
    return n;
}
int mi_pos_policy_index(const mi_pos* pos, uint32_t move, int mode, int is_policy_map) {
    int n = -1;
    if (!pos) return n;
    cra_guard([&] { n = policy_index(policy_tables(mode), pos->pos, move, is_policy_map != 0); });
    return n;
}

// ---- training-sample exporter (engine/src/rl/traindataexporter.cpp) ----
mi_traindata* mi_traindata_create(const char* path, int mode, int version_major, int version_minor, unsigned number_chunks, unsigned chunk_size) {
    mi_traindata* t = nullptr;
    if (cra_guard([&] {
            if (!path || !*path) throw std::invalid_argument("mi_traindata_create: empty path");
            t = new mi_traindata(path, mode, version_major, version_minor, number_chunks, chunk_size);
        })) {
        return nullptr;
    }
    return t;
}
void mi_traindata_destroy(mi_traindata* t) { delete t; }
int mi_traindata_set_phases(mi_traindata* t, int num_phases, int game_phase_definition) {
    if (!t) { cra_set_error("null exporter"); return 1; }
    return cra_guard([&] { t->exp.set_phases(num_phases, game_phase_definition); });
}
int mi_traindata_new_game(mi_traindata* t) {
    if (!t) { cra_set_error("null exporter"); return 1; }
    return cra_guard([&] { t->exp.new_game(); });
}
int mi_traindata_save_sample(mi_traindata* t, const mi_pos* pos, const uint32_t* moves, int n_moves, const double* policy, int n_policy,
                             float best_move_q) {
    if (!t || !pos || (n_moves > 0 && !moves) || (n_policy > 0 && !policy)) { cra_set_error("null argument"); return 1; }
    return cra_guard([&] {
        std::vector<Move> mv(moves, moves + n_moves);
        t->exp.save_sample(pos->pos, mv, policy, size_t(n_policy), best_move_q);
    });
}
int mi_traindata_export_game_samples(mi_traindata* t, int result, unsigned* written) {
    if (!t) { cra_set_error("null exporter"); return 1; }
    return cra_guard([&] {
        if (result < 0 || result > 2) throw std::invalid_argument("result must be 0 DRAWN, 1 WHITE_WIN or 2 BLACK_WIN");
        const size_t n = t->exp.export_game_samples(result);
        if (written) *written = unsigned(n);
    });
}
int mi_traindata_info(const mi_traindata* t, unsigned* number_samples, unsigned* start_index, unsigned* game_index, int* nb_labels, int* channels,
                      int* is_full) {
    if (!t) { cra_set_error("null exporter"); return 1; }
    if (number_samples) *number_samples = unsigned(t->exp.get_number_samples());
    if (start_index) *start_index = unsigned(t->exp.start_index());
    if (game_index) *game_index = unsigned(t->exp.game_index());
    if (nb_labels) *nb_labels = t->exp.nb_labels();
    if (channels) *channels = t->exp.channels();
    if (is_full) *is_full = t->exp.is_file_full() ? 1 : 0;
    return 0;
}

}  // extern "C"
