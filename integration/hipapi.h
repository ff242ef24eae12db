/*
  hipapi.h -- the reference-side binding of libcrazyara_hip.so: a NeuralNetAPI back end for the CrazyAra engine.

  This is the file a CrazyAra maintainer drops into engine/src/nn/ (INTEGRATION.md section 2).  It is compiled in this repository
  too: oracle/ref/build_ref.py builds it against the reference's own nn/neuralnetapi.{h,cpp}, nn/neuralnetapiuser.cpp and
  agents/mctsagent.cpp (oracle/_ref/libcrazyara_ref_hip.so), and tests/test_hipapi_shim_gpu.py runs the reference's
  NeuralNetAPIUser::run_inference and a whole MCTSAgent search on top of it on the GPU.

  Selected like the other back ends (engine/src/uci/crazyara.cpp:650-665):
      #elif defined HIP_BACKEND
          return make_unique<HipAPI>(deviceId, batchSize, modelDirectory, Options["Precision"]);
*/
#ifndef HIPAPI_H
#define HIPAPI_H

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "neuralnetapi.h"
#include "crazyara_hip.h"          /* include/crazyara_hip.h of the MI355X library */

class HipAPI : public NeuralNetAPI
{
private:
    mi_net* net = nullptr;
    std::string precision;

    static void set_shape(nn_api::Shape& shape, std::initializer_list<int> dims) {     // cf. set_shape of tensorrtapi.h:194-198
        shape.nbDims = int(dims.size());
        int i = 0;
        for (int d : dims) shape.v[i++] = d;
    }

    // the four private virtuals of NeuralNetAPI (neuralnetapi.h:172-192).  mi_net_create does all of the work (file discovery,
    // ONNX / .cranet parse, BN folding, weight packing, upload, stream + graph); the hooks keep initialize()'s template-method
    // shape (neuralnetapi.cpp:93-99).
    void load_model() override {
        // `Precision int8` (optionsuci.cpp:143-147): TensorRT's entropy-calibrated INT8 runs its calibrator over the engine's ChessBatchStream
        // when no engine cache exists (tensorrtapi.cpp:297-360).  Here: a calibration file beside the model (<model>.int8calib), made on
        // first use from the same positions -- the plies of the reference's calibration games, which the library holds as data.
        if (precision == "int8" && mi_net_has_int8_calibration(modelDir.c_str()) == 0) {
            info_string("run INT8 quantization calibration");
            if (mi_net_calibrate_int8(modelDir.c_str(), deviceID, nullptr, 0) != 0) {
                throw std::runtime_error(std::string("HipAPI: INT8 calibration failed: ") + mi_last_error());
            }
        }
        net = mi_net_create(modelDir.c_str(), deviceID, int(batchSize), precision.c_str());
        if (net == nullptr) {
            throw std::runtime_error(std::string("HipAPI: ") + mi_last_error());          // ctor errors throw, neuralnetapi.cpp:65-70
        }
        modelName = mi_net_model_name(net);            // carries the "-v<major>.<minor>" read_version_from_string parses
        modelFilePath = modelDir + modelName;
    }
    void init_nn_design() override {
        int in[4], nbPolicy = 0, nbAux = 0, ver = 0, phase = 0;
        mi_net_design(net, in, &nbPolicy, &nbAux, &ver, &phase);
        set_shape(nnDesign.inputShape, {in[0], in[1], in[2], in[3]});                    // cf. tensorrtapi.cpp:128-158
        set_shape(nnDesign.valueOutputShape, {in[0], 1});
        set_shape(nnDesign.policyOutputShape, {in[0], nbPolicy});
        nnDesign.hasAuxiliaryOutputs = nbAux > 0;
        set_shape(nnDesign.auxiliaryOutputShape, {in[0], nbAux});
        nnDesign.isPolicyMap = unsigned(nbPolicy) != StateConstants::NB_LABELS();        // tensorrtapi.cpp:157
    }
    void load_parameters() override {}
    void bind_executor() override {}

public:
    HipAPI(int deviceID, unsigned int batchSize, const std::string& modelDirectory, const std::string& strPrecision) :
        NeuralNetAPI("gpu", deviceID, batchSize, modelDirectory, false), precision(strPrecision)
    {
        initialize();                     // the call TensorrtAPI's constructor makes (tensorrtapi.cpp:62)
    }
    ~HipAPI() { mi_net_destroy(net); }    // NeuralNetAPI has no virtual destructor (neuralnetapi.h:148-311): as ~TensorrtAPI

    HipAPI(const HipAPI&) = delete;
    HipAPI& operator=(const HipAPI&) = delete;

    void predict(float* inputPlanes, float* valueOutput, float* probOutputs, float* auxiliaryOutputs) override {
        // predict() has no error channel in the reference (void); a device failure ends the process like CUDA's CHECK macro
        if (mi_net_predict(net, inputPlanes, valueOutput, probOutputs, auxiliaryOutputs) != 0) {
            info_string_important("HipAPI::predict:", mi_last_error());
            std::abort();
        }
    }
    // the overlap path for a SearchThread that double-buffers
    void submit(float* in, float* value, float* probs, float* aux) { mi_net_submit(net, in, value, probs, aux); }
    void wait() { mi_net_wait(net); }
    mi_net* handle() const { return net; }

    // Descriptor-fed batches (integration/searchthread_hip.patch: SearchThread under HIP_BACKEND).  A leaf is handed
    // over as its 192-byte board descriptor (BoardState::fill_board_desc) and the input planes are built on the GPU; the layout id follows
    // the engine's build mode and the version in the model's file name, like board_to_planes' dispatch (inputrepresentation.cpp:628-680).
    int planes_layout() const {
#if defined(HIP_ENGINE_MODE)
        const int mode = HIP_ENGINE_MODE;               // (a build whose mode is a run-time value)
#elif defined(MODE_CRAZYHOUSE)
        const int mode = MI_MODE_CRAZYHOUSE;
#elif defined(MODE_LICHESS)
        const int mode = MI_MODE_LICHESS;
#else
        const int mode = MI_MODE_CHESS;
#endif
        return mi_planes_layout_minor(mode, int(version::get_major(version)), int(version::get_minor(version)));
    }
    // whole probability vectors (value / probs / aux as predict() fills them; any host memory)
    void predict_boards(const void* descs, int nValid, float* valueOutput, float* probOutputs, float* auxiliaryOutputs) {
        if (mi_net_submit_boards(net, descs, nValid, planes_layout(), valueOutput, probOutputs, auxiliaryOutputs) != 0 || mi_net_wait(net) != 0) {
            info_string_important("HipAPI::predict_boards:", mi_last_error());
            std::abort();
        }
    }
    // only the priors the search reads: gathered[s * stride + j] = probs[s][idx[s * stride + j]], j < cnt[s] (what
    // Node::set_probabilities_for_moves picks out, node.cpp:961-979).  Every buffer must come from mi_host_alloc.
    void predict_boards_gathered(const void* descs, int nValid, const unsigned short* idx, const unsigned* cnt, unsigned stride,
                                 float* valueOutput, float* gathered) {
        if (mi_net_submit_boards_gathered(net, descs, nValid, planes_layout(), idx, cnt, stride, valueOutput, gathered, nullptr) != 0 ||
            mi_net_wait(net) != 0) {
            info_string_important("HipAPI::predict_boards_gathered:", mi_last_error());
            std::abort();
        }
    }
};

#endif // HIPAPI_H
