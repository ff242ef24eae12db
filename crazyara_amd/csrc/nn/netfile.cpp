#include "netfile.h"

#include <dirent.h>
#include <algorithm>
#include <cctype>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace cra {

void NetFile::load(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open model file " + path);
    char magic[8];
    uint64_t hlen = 0;
    f.read(magic, 8);
    f.read(reinterpret_cast<char*>(&hlen), 8);
    if (!f || std::memcmp(magic, "CRANET01", 8) != 0 || hlen > (1u << 26))
        throw std::runtime_error("not a CRANET01 model file: " + path);
    std::string header(hlen, '\0');
    f.read(&header[0], hlen);
    blob.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    std::istringstream hs(header);
    std::string line;
    while (std::getline(hs, line)) {
        if (line.empty()) continue;
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "tensor") {
            std::string name;
            int nd = 0;
            ls >> name >> nd;
            if (!ls || nd < 0 || nd > 8) throw std::runtime_error("bad tensor record in " + path + ": " + line);
            TensorView tv;
            uint64_t numel = 1;
            for (int i = 0; i < nd; ++i) {
                int64_t d = -1;
                ls >> d;
                if (!ls || d < 0 || d > (int64_t(1) << 31)) throw std::runtime_error("bad dimension of tensor " + name + " in " + path);
                tv.shape.push_back(d);
                numel *= uint64_t(d);
                if (numel > (uint64_t(1) << 34)) throw std::runtime_error("tensor " + name + " is implausibly large in " + path);
            }
            uint64_t off = 0;
            ls >> off;
            // no wrap-around: compare against what is left behind the offset
            if (!ls || off % 4 != 0 || off > blob.size() || numel * 4 > blob.size() - off)
                throw std::runtime_error("tensor " + name + " exceeds blob in " + path);
            tv.data = reinterpret_cast<const float*>(blob.data() + off);
            tensors[name] = tv;
        } else {
            std::string val;
            std::getline(ls, val);
            size_t p = val.find_first_not_of(' ');
            meta[key] = p == std::string::npos ? "" : val.substr(p);
        }
    }
}

const TensorView& NetFile::get(const std::string& name) const {
    auto it = tensors.find(name);
    if (it == tensors.end()) throw std::runtime_error("model file lacks tensor " + name);
    return it->second;
}

std::string NetFile::str(const std::string& key, const std::string& dflt) const {
    auto it = meta.find(key);
    return it == meta.end() ? dflt : it->second;
}

int64_t NetFile::num(const std::string& key, int64_t dflt) const {
    auto it = meta.find(key);
    return it == meta.end() ? dflt : std::stoll(it->second);
}

std::vector<std::string> NetFile::list(const std::string& key) const {
    std::vector<std::string> out;
    std::string s = str(key), cur;
    std::istringstream ss(s);
    while (std::getline(ss, cur, ',')) out.push_back(cur);
    return out;
}

static bool has_suffix(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

static std::string pick_model_file(const std::vector<std::string>& files, const std::string& model_dir, int batch_size, const std::string& ext) {
    const std::string want = "-bsize-" + std::to_string(batch_size) + ext;
    for (auto& f : files) if (has_suffix(f, want)) return f;
    for (auto& f : files) {
        if (has_suffix(f, ext)) {
            if (f.find("-bsize-") != std::string::npos)
                throw std::invalid_argument("The given directory at " + model_dir + " should either contain a " + ext.substr(1) +
                    " file supporting the current batch size or one without -bsize-");
            return f;
        }
    }
    return "";
}

std::string find_model_file(const std::string& model_dir, int batch_size) {
    std::vector<std::string> files;
    if (DIR* d = opendir(model_dir.c_str())) {
        while (dirent* e = readdir(d)) files.emplace_back(e->d_name);
        closedir(d);
    } else {
        throw std::invalid_argument("The given directory at " + model_dir + " cannot be opened");
    }
    std::sort(files.begin(), files.end());
    // a converted container wins over the ONNX it came from (as the reference prefers its cached .trt engine, tensorrtapi.cpp:297-332)
    for (const char* ext : {".cranet", ".onnx"}) {
        std::string f = pick_model_file(files, model_dir, batch_size, ext);
        if (!f.empty()) return f;
    }
    throw std::invalid_argument("The given directory at " + model_dir + " doesn't contain a file ending with .cranet or .onnx");
}

int read_version_from_string(const std::string& s) {
    for (size_t p = s.find("-v"); p != std::string::npos; p = s.find("-v", p + 1)) {
        size_t i = p + 2, j;
        if (i >= s.size() || !std::isdigit((unsigned char)s[i])) continue;
        for (j = i; j < s.size() && std::isdigit((unsigned char)s[j]); ++j) {}
        if (j + 1 >= s.size() || !std::isdigit((unsigned char)s[j + 1])) continue;   // any separator char, like regex '.'
        size_t k = j + 1, e;
        for (e = k; e < s.size() && std::isdigit((unsigned char)s[e]); ++e) {}
        return std::stoi(s.substr(i, j - i)) * 1000000 + std::stoi(s.substr(k, e - k)) * 1000;   // make_version(), version.h:53-55
    }
    return 0;
}

int read_game_phase_from_string(const std::string& dir) {
    if (dir.size() < 2) return 0;
    char c = dir[dir.size() - 2];
    return std::isdigit((unsigned char)c) ? c - '0' : 0;
}

}  // namespace cra
