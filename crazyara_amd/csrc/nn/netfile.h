// Model container read by the HIP backend (replaces the ONNX file the reference's TensorRT backend parses,
// engine/src/nn/tensorrtapi.cpp:239-295; onnx is not available in this environment).
//
// Layout:  "CRANET01" | u64 header_len | header text | raw little-endian fp32 blob
// Header text lines:   "<key> <value>"   and   "tensor <name> <ndim> <d0> .. <offset_bytes>"
// Tensor names are the reference PyTorch state-dict keys (rise_mobile_v3.py / builder_util.py), BN un-folded.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace cra {

struct TensorView {
    std::vector<int64_t> shape;
    const float* data = nullptr;
    int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
};

struct NetFile {
    std::map<std::string, std::string> meta;
    std::map<std::string, TensorView> tensors;
    std::vector<char> blob;

    // throws std::runtime_error / std::invalid_argument
    void load(const std::string& path);
    const TensorView& get(const std::string& name) const;
    bool has(const std::string& name) const { return tensors.count(name) != 0; }
    std::string str(const std::string& key, const std::string& dflt = "") const;
    int64_t num(const std::string& key, int64_t dflt = 0) const;
    std::vector<std::string> list(const std::string& key) const;   // comma separated
};

// get_onnx_model_name() (engine/src/nn/neuralnetapi.cpp:57-73) for "*.cranet" and, when the directory holds none, "*.onnx":
// prefers "*-bsize-<B><ext>", else the first "*<ext>" without "-bsize-"; throws invalid_argument otherwise.
std::string find_model_file(const std::string& model_dir, int batch_size);

// Mirrors read_version_from_string() (neuralnetapi.cpp:194-227): "-v<maj>.<min>" -> make_version(maj,min,0)
// = maj*1000000 + min*1000 (engine/src/version.h:37-55), else 0.
int read_version_from_string(const std::string& model_file_name);

// Mirrors read_game_phase_from_string() (neuralnetapi.cpp:229-239).
int read_game_phase_from_string(const std::string& model_dir_with_slash);

}  // namespace cra
