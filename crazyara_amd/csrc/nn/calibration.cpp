// Precision int8's default calibration positions: the plies of the reference's calibration games (the UCI move lists of ChessBatchStream,
// engine/src/environments/chess_related/chessbatchstream.cpp:44-94 -- 232 crazyhouse plies, 104 chess plies -- which TensorRT's
// Int8EntropyCalibrator2 is fed with, tensorrtapi.cpp:349-357), kept as DATA in crazyara_amd/data/opening_games.json and played here on the
// library's own Position; the planes come from the library's own board_to_planes.  Host code only.
#include "rise_net.h"

#include <dlfcn.h>

#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "../chess/planes_host.h"
#include "../chess/position.h"

namespace cra {
namespace {

std::string data_dir() {
    if (const char* e = getenv("CRA_DATA_DIR")) return e;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&data_dir), &info) && info.dli_fname) {
        std::string so = info.dli_fname;                         // .../crazyara_amd/lib/libcrazyara_hip.so -> .../crazyara_amd/data
        const size_t sl = so.find_last_of('/');
        const std::string libdir = sl == std::string::npos ? "." : so.substr(0, sl);
        return libdir + "/../data";
    }
    return "crazyara_amd/data";
}

// the move lists of one variant: the value of "<key>": [[...], [...]] in the JSON file (strings without escapes)
std::vector<std::vector<std::string>> games_of(const std::string& text, const std::string& key) {
    std::vector<std::vector<std::string>> games;
    size_t p = text.find("\"" + key + "\"");
    if (p == std::string::npos) return games;
    p = text.find('[', p);
    if (p == std::string::npos) return games;
    int depth = 0;
    for (; p < text.size(); ++p) {
        const char c = text[p];
        if (c == '[') {
            if (++depth == 2) games.emplace_back();
        } else if (c == ']') {
            if (--depth == 0) break;
        } else if (c == '"' && depth == 2) {
            const size_t q = text.find('"', p + 1);
            if (q == std::string::npos) break;
            games.back().push_back(text.substr(p + 1, q - p - 1));
            p = q;
        }
    }
    return games;
}

}  // namespace

// planes [n][channels][64] of the default calibration positions for a net with `channels` input planes of input version `version`
// (make_version: major * 1e6 + minor * 1e3); throws when no layout of the three build modes has that many channels
std::vector<float> default_calibration_planes(int channels, int version, int* n_boards) {
    const int vmaj = version / 1000000, vmin = (version / 1000) % 1000;
    int mode = -1, layout = -1;
    for (int m = 0; m < 3 && mode < 0; ++m) {
        const int l = layout_for(m, vmaj, vmin);
        if (layout_channels(l) == channels) { mode = m; layout = l; }
    }
    if (mode < 0) throw std::runtime_error("no input-plane layout with " + std::to_string(channels) + " channels for input version " + std::to_string(vmaj) + "." + std::to_string(vmin));
    const std::string path = data_dir() + "/opening_games.json";
    std::ifstream f(path);
    if (!f) throw std::runtime_error("calibration games not found: " + path + " (CRA_DATA_DIR names the directory)");
    std::stringstream ss;
    ss << f.rdbuf();
    const bool house = mode == MODE_CRAZYHOUSE;
    const std::vector<std::vector<std::string>> games = games_of(ss.str(), house ? "crazyhouse" : "chess");
    if (games.empty()) throw std::runtime_error("no calibration games in " + path);
    const chess::Variant variant = house ? chess::V_CRAZYHOUSE : chess::V_CHESS;
    std::vector<float> planes;
    int n = 0;
    for (const auto& g : games) {
        chess::Position pos;
        pos.set(chess::start_fen(variant), false, variant);
        for (size_t ply = 0; ply <= g.size(); ++ply) {           // the position in front of every move and behind the last one
            planes.resize(size_t(n + 1) * channels * 64);
            chess::board_to_planes(pos, layout, true, planes.data() + size_t(n) * channels * 64);
            ++n;
            if (ply == g.size()) break;
            const chess::Move m = pos.uci_to_move(g[ply]);
            if (m == chess::MOVE_NONE) throw std::runtime_error("illegal move " + g[ply] + " in a calibration game of " + path);
            pos.do_move(m);
        }
    }
    *n_boards = n;
    return planes;
}

}  // namespace cra
