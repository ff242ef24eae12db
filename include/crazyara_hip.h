/*
 * crazyara_hip.h -- C ABI of the MI355X-native evaluator (libcrazyara_hip.so).
 *
 * This is the drop-in boundary for CrazyAra's NN plugin surface: a ~100-line `HipAPI : NeuralNetAPI` shim
 *  binds exactly these entry points.  Every function cites the reference interface it replaces
 * (paths relative to the reference repository root).  Plain pointers and sizes only; no C++/torch types.
 *
 * Error model: functions returning int give 0 on success, non-zero on failure; functions returning a pointer give
 * NULL on failure; mi_last_error() then holds a thread-local message.  (The reference throws from constructors --
 * engine/src/nn/neuralnetapi.cpp:65-70 -- and has no error channel in predict(); the shim re-throws / aborts.)
 */
#ifndef CRAZYARA_HIP_H
#define CRAZYARA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------------------------------------- */
const char* mi_last_error(void);
const char* mi_version(void);
/* number of visible HIP devices (UCI options First_Device_ID / Last_Device_ID, engine/src/uci/optionsuci.cpp:114-127) */
int mi_device_count(void);

/* Pinned host buffers: replaces cudaMallocHost / cudaFreeHost in NeuralNetAPIUser
 * (engine/src/nn/neuralnetapiuser.cpp:50-60,80-88). */
void* mi_host_alloc(size_t bytes);
void mi_host_free(void* p);

/* ------------------------------------------------------------------------------------------------------------------
 * Network handle == one NeuralNetAPI instance (engine/src/nn/neuralnetapi.h:148-311)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_net mi_net;

/* TensorrtAPI::TensorrtAPI(deviceID, batchSize, modelDirectory, strPrecision) + initialize()
 * (engine/src/nn/tensorrtapi.cpp:43-63; neuralnetapi.cpp:93-99).  model_dir: directory holding a "*.cranet" file
 * (chosen like get_onnx_model_name(), neuralnetapi.cpp:57-73) or a direct path to one.
 * precision: "float32" | "float16" (UCI option Precision, optionsuci.cpp:143-147). */
mi_net* mi_net_create(const char* model_dir, int device_id, int batch_size, const char* precision);
void mi_net_destroy(mi_net* net);

/* nnDesign + derived getters (neuralnetapi.cpp:81-91, neuralnetapi.h:252-299):
 * in_shape = {batch, C, 8, 8}; nb_policy = get_nb_policy_values(); nb_aux = get_nb_auxiliary_outputs();
 * version = get_version() as make_version(maj,min,0) (engine/src/version.h:53-55); game_phase = get_game_phase(). */
int mi_net_design(const mi_net* net, int in_shape[4], int* nb_policy, int* nb_aux, int* version, int* game_phase);
const char* mi_net_model_name(const mi_net* net);        /* get_model_name(), neuralnetapi.cpp:116-119 */
double mi_net_flops_per_position(const mi_net* net);     /* 2*MACs of the loaded layer list (for roofline reporting) */

/* NeuralNetAPI::predict(float* inputPlanes, float* valueOutput, float* probOutputs, float* auxiliaryOutputs)
 * (neuralnetapi.h:230-237; behaviour of tensorrtapi.cpp:195-237): whole fixed batch, fp32 NCHW host planes,
 * blocking; value after tanh, probs = softmax over all nb_policy entries, aux only if nb_aux > 0. */
int mi_net_predict(mi_net* net, const float* in_planes, float* value, float* probs, float* aux);

/* Asynchronous split of predict(): the MI355X restatement of "two SearchThreads per GPU hide the blocking call"
 * (engine/src/searchthread.cpp:403-416; optionsuci.cpp:182): submit enqueues H2D + forward + D2H on the net's
 * side stream, wait blocks until the host buffers are valid. */
int mi_net_submit(mi_net* net, const float* in_planes, float* value, float* probs, float* aux);
int mi_net_wait(mi_net* net);

/* Device-resident path (no PCIe): pointers to the buffers the captured forward reads/writes.
 * d_planes [B][C][64] float, d_value [B], d_probs [B][nb_policy], d_logits [B][nb_policy] (pre-softmax policy_out),
 * d_aux [B][4] or NULL.  Any out-pointer may be NULL. */
int mi_net_device_buffers(mi_net* net, float** d_planes, float** d_value, float** d_probs, float** d_logits, float** d_aux);
int mi_net_forward_device(mi_net* net);          /* hipGraph replay on the net stream, asynchronous */
int mi_net_sync(mi_net* net);                    /* hipStreamSynchronize(net stream) */
void* mi_net_stream(mi_net* net);                /* the hipStream_t */

/* Run `iters` forwards (graph replays) bracketed by hipEvents on the net's own stream; *ms_total = elapsed.
 * This is the NN-only shape of CrazyAra::inference (engine/src/uci/crazyara.cpp:156-181) minus the PCIe copies. */
int mi_net_time_forward(mi_net* net, int iters, float* ms_total);
/* Per-launch timing: each op of the forward launched un-graphed between two hipEvents, `iters` times.
 * names: op_count pointers to static strings; ms: op_count accumulated milliseconds (sum over iters). */
int mi_net_op_count(const mi_net* net);
int mi_net_time_ops(mi_net* net, int iters, const char** names, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* CRAZYARA_HIP_H */
