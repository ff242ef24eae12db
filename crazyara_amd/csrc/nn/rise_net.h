// RiseNet: the MI355X-native executor behind the NeuralNetAPI boundary.
// Lifecycle mirrors TensorrtAPI (engine/src/nn/tensorrtapi.cpp:43-63,160-237): construct on a device with a fixed batch
// size -> load_model (read .cranet) -> init_nn_design (shapes) -> load_parameters (fold BN, pack MFMA fragments, upload)
// -> bind_executor (stream, device buffers, hipGraph capture of the whole forward) -> predict().
#pragma once
#include <hip/hip_runtime.h>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "netfile.h"

namespace cra {

struct RiseDesign {
    int batch = 0;
    int nb_input_channels = 0;     // C of the [B,C,8,8] input
    int nb_policy = 0;             // policyOutputShape[1]
    int nb_aux = 0;                // auxiliaryOutputShape[1] (0 = none)
    int version = 0;               // make_version(maj,min,0) parsed from the file name (neuralnetapi.cpp:194-227)
    int game_phase = 0;
    double flops_per_position = 0; // 2*MACs, recomputed from the layer list
};

// the host-side weight quantiser of Precision fp8: float -> OCP e4m3fn byte, round to nearest even, clamped at +-448
uint8_t float_to_e4m3(float v);
uint8_t float_to_e5m2(float v);

class RiseNet {
public:
    // model_path: a .cranet file, or a directory searched like get_onnx_model_name() (neuralnetapi.cpp:57-73).
    // precision: "float16" (f16 MFMA operands, f32 accumulate; the reference TensorRT default, optionsuci.cpp:143-147)
    //            or "fp8" ("float8"; "int8" is accepted as the reference's name for its reduced-precision mode): float16 with e4m3
    //            operands in the GEMMs of the residual tower (256-channel bottleneck nets only)
    //            or "float16x3" (float activations, every dense contraction as three f16 MFMAs on hi/lo split operands: ~1e-5 on the
    //            logits at several times the float32 rate, x3.hip)
    //            or "float32" (exact f32 MFMA); float16 runs the residual tower kernel (tower.hip: runs of 3x3 blocks in one launch);
    //            suffix "-perblock" selects one fused launch per bottleneck block, "-unfused" the layer-granular kernels
    //            (both kept for A/B measurements and as independent implementations in the parity tests).   Throws std::invalid_argument / std::runtime_error.
    RiseNet(const std::string& model_path, int device_id, int batch_size, const std::string& precision);
    ~RiseNet();
    RiseNet(const RiseNet&) = delete;
    RiseNet& operator=(const RiseNet&) = delete;

    const RiseDesign& design() const { return design_; }
    const std::string& model_name() const { return model_name_; }
    const std::string& model_file_path() const { return model_file_path_; }
    bool fp16() const { return fp16_; }
    int device() const { return device_; }
    hipStream_t stream() const { return stream_; }

    // NeuralNetAPI::predict contract (neuralnetapi.h:230-237): whole fixed batch, host pointers, blocking;
    // value after tanh, policy after softmax over all nb_policy entries.
    void predict(const float* in_planes, float* value, float* probs, float* aux);
    // asynchronous split of the same call: submit() enqueues H2D + forward + D2H on the net's side stream and returns;
    // wait() blocks until results are in the host buffers.  Host buffers should come from mi_host_alloc (pinned).
    void submit(const float* in_planes, float* value, float* probs, float* aux);
    void wait();
    // descriptor-fed variant: 192-byte BoardDesc per position, planes expanded on the GPU (csrc/chess/planes_kernel.hip)
    void submit_boards(const void* descs_host, int n_valid, int layout, float* value, float* probs, float* aux);
    // the same, but only the probabilities the search will read come back: idx[s * stride .. + cnt[s]) are the policy indices of slot s's
    // legal moves, gathered[] (same layout) receives probs[s][idx].  Every host buffer (descs, idx, cnt, value, gathered, aux) must come
    // from mi_host_alloc / hipHostMalloc: the kernels read and write them in place, there is no copy.
    void submit_boards_gathered(const void* descs_host, int n_valid, int layout, const uint16_t* idx, const uint32_t* cnt, uint32_t stride,
                                float* value, float* gathered, float* aux);

    // INT8 calibration (the reference: Int8EntropyCalibrator2 over ChessBatchStream, tensorrtapi.cpp:334-360): on a net made with Precision
    // "float16-unfused" -- every tensor of a block passes through HBM there -- runs the n boards (float planes, NCHW, as predict() takes
    // them) and returns per bottleneck block the largest |value| of the (gated) stream in front of it and of its depthwise output.
    std::vector<std::pair<float, float>> calibration_maxima(const float* planes_host, int n_boards);

    // Device-resident path: the captured forward reads d_planes() and writes d_value()/d_probs()/d_aux()/d_logits().
    float* d_planes() const { return d_planes_; }     // [B][C][64] float (NCHW, as predict() takes it)
    float* d_value() const { return d_value_; }       // [B]
    float* d_probs() const { return d_probs_; }       // [B][nb_policy]
    float* d_logits() const { return d_logits_; }     // [B][nb_policy] pre-softmax policy_out: valid after a forward made with keep_logits(true)
    void keep_logits(bool on);
    // test hook: the one-launch bottleneck tower also stores the f16 residual stream in front of its first block and behind every
    // block (kernels.h: TowerArgs::block_dump).  Returns the device buffer [n_tiles][B][64][256] f16; throws when the net has no such
    // tower (or more than one run of blocks).
    void* enable_block_dump(int* n_tiles);
    void dev_launch_op(int op, int iters);                        // development: op `op` of the forward `iters` times into the net's stream, no wait
    float* value_head_debug() const { return value_head_dbg_; }   // development: [B][8] stage checksums (CRA_VALUE_HEAD_DEBUG), else null
    // development: the co-residency screen (scripts/coresidency_screen.py, profiles/NOTES.md round 5).  prepare: with a forward of
    // OTHER planes behind it, runs the forward op by op and records, per op, which of the net's mutable device buffers it changes, their
    // contents before and after, and whether the op gives the same bits when launched again on its own output (else its buffers are put
    // back before every launch).  run: op `op` alone `launches` times on the net's stream, every launch compared word for word on the
    // device with the recorded result; returns the number of launches that differed (words: how many 16-byte pieces in all).
    int dev_screen_prepare();
    long dev_screen_run(int op, int launches, long* words);
    std::string dev_screen_info(int op) const;
    float* d_aux() const { return d_aux_; }           // [B][nb_aux] or nullptr
    void forward_async();
    void launch_forward_in_stream();     // the forward as part of a stream's in-order work (submit*, see rise_net.hip)                              // graph replay on stream(); no copies, no sync
    // same forward enqueued kernel-by-kernel on a caller stream (no graph) -- used for profiling / event timing
    void forward_on(hipStream_t s);

    // per-launch bookkeeping (bench.py roofline: live hipEvent timing of each op on the net stream)
    int launches_per_forward() const { return launches_; }
    const char* op_name(int i) const;
    void time_ops(int iters, float* ms);               // ms[i] += elapsed of op i, summed over iters (un-graphed launches)
    float time_forward(int iters);                     // graph replays between two events, returns ms

    // predict()/submit() with ALL of the caller's buffers in pinned host memory (mi_host_alloc / hipHostMalloc / hipHostRegister, as
    // NeuralNetAPIUser allocates them under its TENSORRT branch, neuralnetapiuser.cpp:50-60): no copy commands -- the first kernel reads
    // the planes and the last kernels write value / probabilities / aux straight through PCIe, one queue, nothing handed to the DMA
    // engines and back.  Pageable buffers keep the three hipMemcpyAsync.  CRA_PREDICT_COPY=1 forces the copies (A/B measurements).
    struct IoOverride {
        const float* planes = nullptr;
        float* value = nullptr;
        float* probs = nullptr;
        float* aux = nullptr;
    };
    bool last_submit_was_zero_copy() const { return last_zero_copy_; }

private:
    struct Impl;
    template <typename T> void build(const NetFile& nf);
    template <typename T> void enqueue(hipStream_t s, const IoOverride* io = nullptr);
    template <typename T> void launch_op(int i, hipStream_t s, const IoOverride* io = nullptr);
    void capture();
    bool buffers_are_pinned(const float* in_planes, float* value, float* probs, float* aux);
    bool last_zero_copy_ = false;
    bool counted_in_flight_ = false;   // this net's predict is counted in the device's predicts-in-flight (submit ... wait)
    int staged_calls_left_ = 0;        // predict on pinned buffers: calls that still take the staged form after another user was met in flight
    // Development switches: read from the environment ONCE, when the net is constructed -- never on the per-batch path,
    // where several lane threads would otherwise scan `environ` per kernel launch beside a host program that may call setenv.
    struct DevSwitches {
        int conv_dev = -1;              // CRA_X3_CONV_DEV: bisecting switches of conv_gemm_x3_kernel
        bool device_graph = false;      // CRA_DEVICE_GRAPH: replay the graph also for a one-launch forward
        bool lane_graph = false;        // CRA_LANE_GRAPH: the lane step replays the graph
        bool lane_no_graph = false;     // CRA_LANE_NO_GRAPH: the lane step never replays the graph
        bool predict_copy = false;      // CRA_PREDICT_COPY: predict() stages through device buffers also for pinned caller buffers
        bool predict_zero_copy = false; // CRA_PREDICT_ZERO_COPY: predict() on pinned buffers never stages, whatever else is in flight
        char lane_launches = 0;         // CRA_LANE_LAUNCHES: '1' / '2' / '3' force the shape of the lane step
        bool lane_sync = false;         // CRA_LANE_SYNC: a stream sync between forward and gather
        bool x3_symmetric = false;      // CRA_X3_TOWER=symmetric: the float16x3 tower with every wave running all three phases
        int x3_split_dev = 0;           // CRA_X3_SPLIT_DEV: timing switches of block_x3_split_kernel (x3.hip; bits 2 and 4 give wrong results)
        int x3_split_max_g = 0;         // CRA_X3_SPLIT_MAX_G: upper bound on the workgroups per board of the split-board blocks
        int x3_split_max_batch = 0;     // CRA_X3_SPLIT_MAX_BATCH: the largest batch that runs split-board (default kBoardSplitMaxBatch)
        bool no_small_path = false;     // CRA_NO_SMALL_PATH: partial batches run the whole batch's forward as before round 6 (A/B)
        int small_conv_split = 2;       // CRA_SMALL_BATCH_CONV_SPLIT: the wide convs of a small-batch net as 1 = two workgroups of 128 couts per board, 2 = four of 64
        bool own_stream = false;        // CRA_OWN_STREAM_PER_NET: a stream created (and destroyed) per net, as before the streams of the library (A/B)
        DevSwitches();
    } dev_;
    float* value_head_dbg_ = nullptr;
    bool keep_logits_ = false;   // the one-launch head also writes policy_out (pre-softmax) to d_logits() (parity tests); nets whose heads
                                 // run as separate launches always have it there (the softmax launch reads it)

    RiseDesign design_;
    std::string model_name_, model_file_path_;
    bool fp16_ = true;
    bool x3_ = false;            // Precision float16x3: float activations, split-operand f16 MFMAs (x3.hip); fp16_ is false
    bool p8_ = false;            // Precision float16p8: float16x3 whose one-launch tower takes the cross terms of both 1x1 GEMMs through e5m2 MFMAs
    bool fp8_tower_ = false;     // Precision fp8 / int8: 8-bit operands in the residual tower's GEMMs, everything else as float16
    bool int8_ = false;          // Precision int8: the calibrated INT8 mode (tower.hip Q = 2); fp8_tower_ is set too (same streams and tiles)
    std::vector<std::pair<float, float>> int8_calib_;   // per block: max |stream in front of it|, max depthwise output (read_int8_calibration)
    bool fused_ = true;
    bool tower_ = true;
    // Small batches (round 6): float16x3 / float16p8 nets made for at most kBoardSplitMaxBatch boards (64: measured faster up to 96, profiles/r06/d_*) run their 3x3 bottleneck blocks one per
    // launch with up to C_op / 128 workgroups per board (x3.hip: block_x3_split_kernel, float16x3 arithmetic in both modes) instead of the
    // one-workgroup-per-board tower, whose latency a small batch pays in full on a handful of CUs.  "-1wg" after the precision keeps the
    // one-workgroup-per-board tower (A/B, and the parity tests of the tower kernels on the small fixtures).
    bool board_split_ = true;
    static constexpr int kBoardSplitMaxBatch = 64;
    // A net made for MORE boards still meets small batches: the root of a `go` (one board), the first batches of a single tree, the tail of
    // a game loop.  submit_boards / submit_boards_gathered with at most kBoardSplitMaxBatch valid boards go to a companion net of that batch
    // size (same model and precision, made on first use) whose launches take the number of boards of THIS call -- a forward of n boards
    // instead of one of the whole batch (0.33 ms instead of 0.70 for one board of RISEv2-19).  float16x3 / float16p8 only.
    std::unique_ptr<RiseNet> small_;  // works in this net's stream (owns_stream_ = false there)
    bool owns_stream_ = true;         // this net took stream_ itself (from the library's set, or created it: stream_slot_ < 0)
    int stream_slot_ = -1;            // which stream of the library's per-device set this net works in (rise_net.hip: NetStreams)
    void touch_stream() const;        // this net is submitting work: its stream was used NOW (what the choice for the next new net looks at)
    std::string precision_arg_;       // what the constructor was given (the companion is made with the same)
    int dyn_n_ = 0, dyn_prev_g_ = 1;  // > 0 while a forward of dyn_n_ boards is being enqueued (launch_op)
    bool small_path_ok() const;
    RiseNet& small_net();
    bool one_launch_ = true;     // stem + tower + head in one launch when the net is exactly that chain ("-3k": three launches)
    bool rt_thin_waves_ = false; // dense residual tower: 8 waves x 32 couts ("-8w") instead of 4 x 64
    int boards_per_wg_ = 0;      // dense residual tower: 0 = by batch size (2 from 512 boards), 1 / 2 = forced ("-1b" / "-2b")
    struct Turn;                       // forwards of different streams take turns when one fills the chip (rise_net.hip)
    int cu_count_ = 256;
    int device_ = 0;
    int launches_ = 0;
    hipStream_t stream_ = nullptr;
    hipGraph_t graph_ = nullptr;
    hipGraphExec_t graph_exec_ = nullptr;
    void* d_desc_ = nullptr;
    float *d_planes_ = nullptr, *d_value_ = nullptr, *d_probs_ = nullptr, *d_logits_ = nullptr, *d_aux_ = nullptr;
    std::unique_ptr<Impl> impl_;
};

// Precision int8's calibration file beside the model (TensorRT keeps its calibration cache the same way): <model file>.int8calib, text --
// "crazyara-int8-calibration 1", "boards <n>", "blocks <m>", then m lines "<max |stream|> <max depthwise output>".
std::string int8_calibration_path(const std::string& model_file_path);
std::vector<std::pair<float, float>> read_int8_calibration(const std::string& model_file_path);      // empty if the file is missing
// runs the calibration pass (a float16 layer-kernel net of its own on `device_id`) and writes the file; returns its path.
// planes_host == nullptr: the default calibration positions -- the plies of the reference's calibration games (calibration.cpp)
std::string calibrate_int8(const std::string& model_path, int device_id, const float* planes_host, int n_boards);
std::vector<float> default_calibration_planes(int channels, int version, int* n_boards);

}  // namespace cra
