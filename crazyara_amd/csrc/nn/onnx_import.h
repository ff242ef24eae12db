// ONNX model import: the reference's on-disk model format of record read straight into the HIP backend's
// NetFile, so a model directory of a CrazyAra / ClassicAra / MultiAra release ("<prefix>-v<maj>.<min>[-bsize-<B>].onnx",
// trainer_agent_pytorch.py:588-633) loads where the reference's TensorRT backend parses it (tensorrtapi.cpp:239-295).
//
// The importer does not interpret arbitrary ONNX: it recognises the graphs the reference's model zoo exports --
//   stem conv -> residual tower (mobile bottleneck blocks with optional ca_se / eca_se gates, ClassicalResidualBlock or
//   AlphaZero ResidualBlock) -> value head (tanh or WDL + plies-to-end) and policy head (policy map or flat labels)
// -- in the flavours exporters leave them: BatchNormalization nodes or BN already folded into the convolution's weight and bias,
// Gemm or MatMul(+Add) for Linear, shape plumbing as Reshape / Flatten with constant or computed (Shape-Gather-Concat) shapes.
// Anything else is rejected with a message naming the node.  No protobuf library: the few messages needed are decoded from the
// wire format (field numbers of the published onnx.proto3 are listed in onnx_import.cpp).
#pragma once
#include <string>

#include "netfile.h"

namespace cra {

// Fills `nf` (meta keys as crazyara_amd/netfile.py:export_rise writes them, tensors under the reference's state-dict names).
// Every conv+BN pair is emitted in the normalised form  BN.weight = scale, BN.bias = shift, running_mean = 0, running_var = 1 - eps
// (the loader's fold reproduces scale and shift to 1 ulp).  Throws std::runtime_error on malformed or unsupported input.
void import_onnx(const std::string& path, NetFile& nf);
void import_onnx_bytes(const void* data, size_t size, const std::string& model_file_name, NetFile& nf);

// Writes `nf` as a CRANET01 file (the conversion the reference caches as a .trt engine next to the ONNX, tensorrtapi.cpp:297-332).
void write_cranet(const NetFile& nf, const std::string& path);

}  // namespace cra
