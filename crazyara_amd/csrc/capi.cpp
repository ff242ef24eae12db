// C ABI of libcrazyara_hip.so -- see include/crazyara_hip.h for the contract and reference citations.
#include "../../include/crazyara_hip.h"

#include <hip/hip_runtime.h>

#include <exception>
#include <string>

#include "capi_common.h"
#include "nn/onnx_import.h"
#include "nn/rise_net.h"

namespace {
thread_local std::string g_err;
template <typename F> int guard(F&& f) { return cra_guard(static_cast<F&&>(f)); }
}  // namespace

void cra_set_error(const std::string& msg) { g_err = msg; }
const char* cra_get_error() { return g_err.c_str(); }

#include "capi_net.h"

extern "C" {

const char* mi_last_error(void) { return g_err.c_str(); }
const char* mi_version(void) { return "crazyara_amd 0.1 (gfx950)"; }

int mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        g_err = "hipGetDeviceCount failed (no HIP device visible)";
        return 0;
    }
    return n;
}

void* mi_host_alloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) {
        g_err = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return nullptr;
    }
    return p;
}
void mi_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

mi_net* mi_net_create(const char* model_dir, int device_id, int batch_size, const char* precision) {
    mi_net* h = nullptr;
    if (guard([&] { h = new mi_net(model_dir, device_id, batch_size, precision); })) return nullptr;
    return h;
}
void mi_net_destroy(mi_net* net) { delete net; }

int mi_net_calibrate_int8(const char* model_dir, int device_id, const float* planes, int n_boards) {
    if (!model_dir) { g_err = "null argument to mi_net_calibrate_int8"; return 1; }
    return guard([&] { (void)cra::calibrate_int8(model_dir, device_id, planes, n_boards); });
}
int mi_net_has_int8_calibration(const char* model_dir) {
    if (!model_dir) { g_err = "null argument to mi_net_has_int8_calibration"; return -1; }
    int found = 0;
    if (guard([&] {
            std::string path = model_dir;
            const bool is_file = path.size() > 5 && (path.compare(path.size() - 5, 5, ".onnx") == 0 || (path.size() > 7 && path.compare(path.size() - 7, 7, ".cranet") == 0));
            if (!is_file) {
                if (path.empty()) throw std::invalid_argument("The given directory must not be empty.");
                if (path.back() != '/') path += "/";
                path += cra::find_model_file(path, 0);
            }
            found = cra::read_int8_calibration(path).empty() ? 0 : 1;
        }))
        return -1;
    return found;
}

int mi_e4m3_from_float(float v) { return int(cra::float_to_e4m3(v)); }
int mi_e5m2_from_float(float v) { return int(cra::float_to_e5m2(v)); }

int mi_onnx_to_cranet(const char* onnx_path, const char* cranet_path) {
    if (!onnx_path || !cranet_path) { g_err = "null argument to mi_onnx_to_cranet"; return 1; }
    return guard([&] {
        cra::NetFile nf;
        cra::import_onnx(onnx_path, nf);
        cra::write_cranet(nf, cranet_path);
    });
}

int mi_net_design(const mi_net* net, int in_shape[4], int* nb_policy, int* nb_aux, int* version, int* game_phase) {
    if (!net) { g_err = "null net"; return 1; }
    const cra::RiseDesign& d = net->net.design();
    if (in_shape) { in_shape[0] = d.batch; in_shape[1] = d.nb_input_channels; in_shape[2] = 8; in_shape[3] = 8; }
    if (nb_policy) *nb_policy = d.nb_policy;
    if (nb_aux) *nb_aux = d.nb_aux;
    if (version) *version = d.version;
    if (game_phase) *game_phase = d.game_phase;
    return 0;
}
const char* mi_net_model_name(const mi_net* net) { return net ? net->net.model_name().c_str() : ""; }
double mi_net_flops_per_position(const mi_net* net) { return net ? net->net.design().flops_per_position : 0.0; }

int mi_net_predict(mi_net* net, const float* in_planes, float* value, float* probs, float* aux) {
    if (!net || !in_planes || !value || !probs) { g_err = "null argument to mi_net_predict"; return 1; }
    return guard([&] { net->net.predict(in_planes, value, probs, aux); });
}
int mi_net_submit(mi_net* net, const float* in_planes, float* value, float* probs, float* aux) {
    if (!net || !in_planes || !value || !probs) { g_err = "null argument to mi_net_submit"; return 1; }
    return guard([&] { net->net.submit(in_planes, value, probs, aux); });
}
int mi_net_submit_boards(mi_net* net, const void* descs_host, int n_valid, int layout, float* value, float* probs, float* aux) {
    if (!net || (!descs_host && n_valid > 0) || !value || !probs) { g_err = "null argument to mi_net_submit_boards"; return 1; }
    return guard([&] { net->net.submit_boards(descs_host, n_valid, layout, value, probs, aux); });
}
int mi_net_submit_boards_gathered(mi_net* net, const void* descs_host, int n_valid, int layout, const unsigned short* idx,
                                  const unsigned* cnt, unsigned stride, float* value, float* gathered, float* aux) {
    if (!net || (n_valid > 0 && (!descs_host || !idx || !cnt || !gathered)) || !value) { g_err = "null argument to mi_net_submit_boards_gathered"; return 1; }
    return guard([&] { net->net.submit_boards_gathered(descs_host, n_valid, layout, idx, cnt, stride, value, gathered, aux); });
}
int mi_net_wait(mi_net* net) {
    if (!net) { g_err = "null net"; return 1; }
    return guard([&] { net->net.wait(); });
}
int mi_net_last_submit_zero_copy(const mi_net* net) { return net && net->net.last_submit_was_zero_copy() ? 1 : 0; }

int mi_net_device_buffers(mi_net* net, float** d_planes, float** d_value, float** d_probs, float** d_logits, float** d_aux) {
    if (!net) { g_err = "null net"; return 1; }
    if (d_planes) *d_planes = net->net.d_planes();
    if (d_value) *d_value = net->net.d_value();
    if (d_probs) *d_probs = net->net.d_probs();
    if (d_logits) *d_logits = net->net.d_logits();
    if (d_aux) *d_aux = net->net.d_aux();
    return 0;
}
int mi_net_keep_logits(mi_net* net, int on) {
    if (!net) { g_err = "null net"; return 1; }
    net->net.keep_logits(on != 0);
    return 0;
}
void* mi_net_block_dump(mi_net* net, int* n_tiles) {
    if (!net) { g_err = "null net"; return nullptr; }
    void* p = nullptr;
    if (guard([&] { p = net->net.enable_block_dump(n_tiles); })) return nullptr;
    return p;
}
// development hook (not in the public header): device pointer to the value head's [B][8] stage checksums, null unless the process
// runs with CRA_VALUE_HEAD_DEBUG (scripts/lane_divergence.py)
int mi_dev_launch_op(mi_net* net, int op, int iters) {       // development hook: one op of the forward, `iters` times, no wait
    if (!net) { g_err = "null net"; return 1; }
    return guard([&] { net->net.dev_launch_op(op, iters); });
}
// development hooks of the co-residency screen (scripts/coresidency_screen.py; RiseNet::dev_screen_*)
int mi_dev_screen_prepare(mi_net* net) {
    if (!net) { g_err = "null net"; return -1; }
    int n = -1;
    if (guard([&] { n = net->net.dev_screen_prepare(); })) return -1;
    return n;
}
long mi_dev_screen_run(mi_net* net, int op, int launches, long* words) {
    if (!net) { g_err = "null net"; return -1; }
    long n = -1;
    if (guard([&] { n = net->net.dev_screen_run(op, launches, words); })) return -1;
    return n;
}
int mi_dev_screen_info(mi_net* net, int op, char* out, int cap) {
    if (!net || !out || cap <= 0) { g_err = "null argument"; return 1; }
    const std::string s = net->net.dev_screen_info(op);
    snprintf(out, size_t(cap), "%s", s.c_str());
    return 0;
}
void* mi_dev_value_head_debug(mi_net* net) { return net ? static_cast<void*>(net->net.value_head_debug()) : nullptr; }
int mi_net_forward_device(mi_net* net) {
    if (!net) { g_err = "null net"; return 1; }
    return guard([&] { net->net.forward_async(); });
}
int mi_net_sync(mi_net* net) {
    if (!net) { g_err = "null net"; return 1; }
    return guard([&] { net->net.wait(); });
}
void* mi_net_stream(mi_net* net) { return net ? static_cast<void*>(net->net.stream()) : nullptr; }

int mi_net_time_forward(mi_net* net, int iters, float* ms_total) {
    if (!net || !ms_total) { g_err = "null argument"; return 1; }
    return guard([&] { *ms_total = net->net.time_forward(iters); });
}
int mi_net_op_count(const mi_net* net) { return net ? net->net.launches_per_forward() : 0; }
int mi_net_time_ops(mi_net* net, int iters, const char** names, float* ms) {
    if (!net || !ms) { g_err = "null argument"; return 1; }
    return guard([&] {
        const int n = net->net.launches_per_forward();
        for (int i = 0; i < n; ++i) {
            if (names) names[i] = net->net.op_name(i);
            ms[i] = 0.f;
        }
        net->net.time_ops(iters, ms);
    });
}

}  // extern "C"
