#include "rise_net.h"

#include <atomic>
#include <mutex>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <type_traits>
#include <stdexcept>

#include "kernels.h"
#include "onnx_import.h"
#include "../chess/planes_host.h"

namespace cra {

#define HIP_CHECK(expr)                                                                                      \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(_e) + " at " #expr);      \
    } while (0)

namespace {
constexpr double kBnEps = 1e-5;   // torch.nn.BatchNorm2d default; the reference never overrides it
// the float16x3 forward's value head: false = conv GEMM + FC GEMM + value_final (three launches), true = value_head_kernel (one)
constexpr bool kX3ValueHeadOneLaunch = true;

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// conv weight [cout][cin_g][k][k] + BN -> folded double weights / bias
struct Folded {
    std::vector<double> w;   // same layout as the input conv weight
    std::vector<double> b;   // [cout]
};

// float -> OCP e4m3fn byte: round to nearest even, subnormals down to 2^-9, beyond +-448 clamps (the device conversion runs with
// MODE.FP16_OVFL = 1 and does the same)
uint8_t to_e4m3(double v) {
    const uint8_t sign = std::signbit(v) ? 0x80 : 0;
    double a = std::fabs(v);
    if (!(a == a)) return uint8_t(sign | 0x7f);
    if (a >= 448.0) return uint8_t(sign | 0x7e);
    if (a < std::ldexp(1.0, -10)) return sign;                   // below half the smallest subnormal (a tie at 2^-10 rounds to even = 0)
    int e;
    (void)std::frexp(a, &e);                                     // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;                                              // a = (1 + f) * 2^ex
    if (ex < -6) ex = -6;                                        // subnormal range: fixed quantum 2^-9
    const double q = std::ldexp(1.0, ex - 3);                    // spacing of representable values around a
    double n = std::nearbyint(a / q);                            // default rounding mode: nearest even
    int mant = int(n);                                           // in units of q: normal numbers 8..16, subnormals 0..8
    if (ex == -6 && mant < 8) return uint8_t(sign | mant);
    if (mant == 16) { mant = 8; ++ex; }
    if (ex > 8 || (ex == 8 && mant - 8 > 6)) return uint8_t(sign | 0x7e);
    return uint8_t(sign | ((ex + 7) << 3) | (mant - 8));
}
// power of two that brings max |w| of a row into [1, 2); 1 for an all-zero row
double row_scale_pow2(double max_abs) {
    if (!(max_abs > 0.0)) return 1.0;
    int e;
    (void)std::frexp(max_abs, &e);
    e -= 1;
    if (e < -24) e = -24;
    return std::ldexp(1.0, e);
}

Folded fold_bn(const NetFile& nf, const std::string& conv, const std::string& bn) {
    const TensorView& w = nf.get(conv + ".weight");
    const int64_t cout = w.shape[0], per = w.numel() / cout;
    Folded f;
    f.w.resize(w.numel());
    f.b.assign(cout, 0.0);
    if (bn.empty()) {
        for (int64_t i = 0; i < w.numel(); ++i) f.w[i] = w.data[i];
        return f;
    }
    const float *g = nf.get(bn + ".weight").data, *be = nf.get(bn + ".bias").data, *m = nf.get(bn + ".running_mean").data,
                *v = nf.get(bn + ".running_var").data;
    for (int64_t co = 0; co < cout; ++co) {
        const double sc = double(g[co]) / std::sqrt(double(v[co]) + kBnEps);
        for (int64_t i = 0; i < per; ++i) f.w[co * per + i] = double(w.data[co * per + i]) * sc;
        f.b[co] = double(be[co]) - double(m[co]) * sc;
    }
    return f;
}

template <typename T> T cast_w(double v);
template <> half_t cast_w<half_t>(double v) { return half_t(float(v)); }
template <> float cast_w<float>(double v) { return float(v); }

// MFMA A-fragment image, see kernels.h
template <typename T>
std::vector<T> pack_dense(const Folded& f, int cout, int cin, int ks, int cout_pad, int cin_pad) {
    const int kt = ks * ks * cin_pad, nslab = kt / 32, nct = cout_pad / 16;
    std::vector<T> out(size_t(cout_pad) * kt);
    for (int ct = 0; ct < nct; ++ct)
        for (int s = 0; s < nslab; ++s)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int co = ct * 16 + (l & 15);
                    const int k = s * 32 + (l >> 4) * 8 + j;
                    const int tap = k / cin_pad, ci = k % cin_pad;
                    double v = 0.0;
                    if (co < cout && ci < cin) v = f.w[(size_t(co) * cin + ci) * ks * ks + tap];
                    out[((size_t(ct) * nslab + s) * 64 + l) * 8 + j] = cast_w<T>(v);
                }
    return out;
}

// Precision float16x3: w = hi + lo, both f16 (hi = the nearest f16, lo = the nearest f16 to the rest), as two fragment images
struct SplitPack { std::vector<half_t> hi, lo; };
SplitPack pack_dense_split(const Folded& f, int cout, int cin, int ks, int cout_pad, int cin_pad) {
    Folded fh = f, fl = f;
    for (size_t i = 0; i < f.w.size(); ++i) {
        const half_t h = half_t(float(f.w[i]));
        fh.w[i] = double(float(h));
        fl.w[i] = f.w[i] - fh.w[i];
    }
    return SplitPack{pack_dense<half_t>(fh, cout, cin, ks, cout_pad, cin_pad), pack_dense<half_t>(fl, cout, cin, ks, cout_pad, cin_pad)};
}

// e5m2 ("bf8": f16's exponent field, two mantissa bits), round to nearest even from the float value, subnormals kept, saturating
uint8_t to_e5m2(float f) {
    const uint8_t sign = std::signbit(f) ? 0x80 : 0;
    const double a = std::fabs(double(f));
    if (!(a > 0.0)) return sign;
    int e = 0;
    (void)std::frexp(a, &e);
    int E = e - 1;                                                      // a = 1.m * 2^E
    if (E < -14) {                                                      // subnormal: units of 2^-16
        const int q = int(std::nearbyint(std::ldexp(a, 16)));
        return uint8_t(sign | (q >= 4 ? 0x04 : q));
    }
    int mant = int(std::nearbyint((std::ldexp(a, -E) - 1.0) * 4.0));
    if (mant == 4) { mant = 0; ++E; }
    if (E > 15) return uint8_t(sign | 0x7B);                            // the largest finite value (1.75 * 2^15)
    return uint8_t(sign | ((E + 15) << 2) | mant);
}

// Precision float16p8, expand / project weights of a tower block (x3.hip: tower_p8_kernel; kernels.h: X3TowerBlock; oracle: _p8_conv).
// W' = w * 2^p with p = 11 - floor(log2(max |w|)) (the largest weight lands in [2048, 4096)): hi = rne_f16(W') is the main term's operand;
// the 8-bit image holds, per cout tile and 64 k, a lane's 32 bytes -- lane groups 0, 1: e5m2((W' - hi) * c) for k [0, 32), [32, 64) (they
// meet the activations' hi8), groups 2, 3: e5m2(hi * c) for the same k (they meet the activations' lo8) -- bytes 0-15 in "slab" 2 J, bytes
// 16-31 in "slab" 2 J + 1 of the lo image's geometry.  c = 1 / (1 - ln 2 / 8): the kernel's activation bytes are TRUNCATED f16 values (their
// high bytes), which lose 2^e / 8 on average; the weight images take the mean back.  All three products carry the factor 2^p; *inv = 2^-p.
constexpr double kP8TruncCompensation = 1.0 / (1.0 - 0.125 * 0.6931471805599453);
SplitPack pack_dense_p8(const Folded& f, int cout, int cin, int ks, int cout_pad, int cin_pad, double* inv) {
    double mx = 0.0;
    for (double v : f.w) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.0) { (void)std::frexp(mx, &e); e -= 1; }               // mx = m * 2^e, m in [1, 2)
    const int p = 11 - e;
    *inv = std::ldexp(1.0, -p);
    Folded fs = f;
    for (double& v : fs.w) v = std::ldexp(v, p);
    SplitPack out;
    out.hi = pack_dense<half_t>(fs, cout, cin, ks, cout_pad, cin_pad);
    const int nslab = ks * ks * cin_pad / 32, nct = cout_pad / 16;     // k = tap * cin_pad + ci, as pack_dense walks it
    if (cin_pad % 64 != 0) throw std::runtime_error("float16p8: K per tap must be a multiple of 64");
    std::vector<uint8_t> bytes(size_t(cout_pad) * ks * ks * cin_pad * 2, 0);
    for (int ct = 0; ct < nct; ++ct)
        for (int J = 0; J < nslab / 2; ++J)
            for (int l = 0; l < 64; ++l)
                for (int bb = 0; bb < 32; ++bb) {
                    const int co = ct * 16 + (l & 15), lg = l >> 4, k = 64 * J + (lg & 1) * 32 + bb;
                    const int tap = k / cin_pad, ci = k % cin_pad;
                    uint8_t q = 0;
                    if (co < cout && ci < cin) {
                        const double W = fs.w[(size_t(co) * cin + ci) * ks * ks + tap];
                        const double hi = double(float(half_t(float(W))));
                        q = to_e5m2(float((lg < 2 ? W - hi : hi) * kP8TruncCompensation));
                    }
                    bytes[((size_t(ct) * nslab + 2 * J + (bb >> 4)) * 64 + l) * 16 + (bb & 15)] = q;
                }
    out.lo.resize(bytes.size() / 2);
    std::memcpy(out.lo.data(), bytes.data(), bytes.size());
    return out;
}

// Precision float16x3, depthwise 3x3 records of a block (x3.hip: X3Depthwise): per tile of 16 expanded channels 16 rows of 16 floats (1 KiB,
// one 16-byte load per lane) -- rows 0-2 the folded taps of column dx = -1 (dy = -1, 0, 1), rows 3-5 dx = 0, rows 6-8 dx = +1, row 9 the BN1
// bias, row 10 the BN2 bias, rows 11-15 zeros: a lane on file a / h reads its dx = -1 / +1 weights from rows 11-13
std::vector<float> pack_x3_depthwise_records(const Folded& bn1, const Folded& dw, int cop, int cop_pad) {
    std::vector<float> rec(size_t(cop_pad) * 16, 0.f);
    for (int c = 0; c < cop; ++c) {
        float* tile = rec.data() + size_t(c / 16) * 256 + (c % 16);
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) tile[(dx * 3 + dy) * 16] = float(dw.w[size_t(c) * 9 + dy * 3 + dx]);
        tile[9 * 16] = float(bn1.b[c]);
        tile[10 * 16] = float(dw.b[c]);
    }
    return rec;
}

// the same for a 5x5 depthwise (x3.hip: X3Depthwise5): per tile 32 rows of 16 floats (2 KiB) -- rows 5 g + i the folded taps of column dx = g - 2
// (dy = i - 2), row 25 the BN1 bias, row 26 the BN2 bias, rows 27-31 zeros
std::vector<float> pack_x3_depthwise_records5(const Folded& bn1, const Folded& dw, int cop, int cop_pad) {
    std::vector<float> rec(size_t(cop_pad) * 32, 0.f);
    for (int c = 0; c < cop; ++c) {
        float* tile = rec.data() + size_t(c / 16) * 512 + (c % 16);
        for (int dy = 0; dy < 5; ++dy)
            for (int dx = 0; dx < 5; ++dx) tile[(dx * 5 + dy) * 16] = float(dw.w[size_t(c) * 25 + dy * 5 + dx]);
        tile[25 * 16] = float(bn1.b[c]);
        tile[26 * 16] = float(dw.b[c]);
    }
    return rec;
}

enum class OpKind { PlanesToAct, Conv, Depthwise, SE, ValueHead, Softmax, Block, ValueFinal, SEGate, Tower, Head, Stem, ResTower, Forward, TowerX3, BlockX3Split, X3SplitFinish, HeadsSmall };

struct Op {
    OpKind kind;
    ConvArgs conv{};
    bool from_planes = false;     // float16x3 stem conv: reads the NCHW input planes (their address is a launch-time value too)
    bool fused_softmax = false;   // float16x3 policy-map conv: the softmax runs in its launch (the probabilities' address is a launch-time value)
    // depthwise / se
    const void* x = nullptr;
    void* y = nullptr;
    const float *w0 = nullptr, *w1 = nullptr, *b0 = nullptr;
    int C = 0, ks = 0, se_kind = 0;
    ValueHeadArgs vh{};
    BlockArgs blk{};
    ValueFinalArgs vf{};
    TowerArgs tw{};
    HeadArgs hd{};
    ResTowerArgs rt{};
    StemArgs st{};
    X3TowerArgs tx{};
    X3SplitArgs xs{};             // BlockX3Split; X3SplitFinish: x_parts, gin, batch and (xs_y) the float stream
    float* xs_y = nullptr;
};
}  // namespace

// development: what the co-residency screen knows about one op (RiseNet::dev_screen_prepare)
struct ScreenOp {
    struct Buf { char* live; char* before; char* after; size_t bytes; };
    std::vector<Buf> writes;          // the mutable buffers the op changes
    bool idempotent = true;           // launched again on its own result it gives the same bits
};

struct RiseNet::Impl {
    std::vector<void*> allocs;
    std::vector<std::pair<char*, size_t>> mutables;      // allocations that are not uploaded constants: activations, outputs, scratch
    std::vector<Op> ops;
    std::vector<ScreenOp> screen;
    std::vector<void*> screen_allocs;
    unsigned* screen_bad = nullptr;
    int cin_pad = 0;

    void* dalloc(size_t bytes, bool constant = false) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
        allocs.push_back(p);
        if (!constant) mutables.emplace_back(static_cast<char*>(p), bytes ? bytes : 16);
        return p;
    }
    template <typename U> U* upload(const std::vector<U>& h) {
        U* d = static_cast<U*>(dalloc(h.size() * sizeof(U), true));
        HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice));
        return d;
    }
    float* upload_d2f(const std::vector<double>& h, size_t pad_to = 0) {
        std::vector<float> f(std::max(h.size(), pad_to), 0.f);
        for (size_t i = 0; i < h.size(); ++i) f[i] = float(h[i]);
        return upload(f);
    }
    ~Impl() {
        for (void* p : allocs) (void)hipFree(p);
        for (void* p : screen_allocs) (void)hipFree(p);
    }
};

RiseNet::DevSwitches::DevSwitches() {
    if (const char* e = getenv("CRA_X3_CONV_DEV")) conv_dev = atoi(e);
    device_graph = getenv("CRA_DEVICE_GRAPH") != nullptr;
    lane_graph = getenv("CRA_LANE_GRAPH") != nullptr;
    lane_no_graph = getenv("CRA_LANE_NO_GRAPH") != nullptr;
    predict_copy = getenv("CRA_PREDICT_COPY") != nullptr;
    predict_zero_copy = getenv("CRA_PREDICT_ZERO_COPY") != nullptr;
    if (const char* e = getenv("CRA_LANE_LAUNCHES")) lane_launches = e[0];
    lane_sync = getenv("CRA_LANE_SYNC") != nullptr;
    if (const char* e = getenv("CRA_X3_TOWER")) x3_symmetric = e[0] == 's';
    if (const char* e = getenv("CRA_X3_SPLIT_DEV")) x3_split_dev = atoi(e);
    if (const char* e = getenv("CRA_X3_SPLIT_MAX_G")) x3_split_max_g = atoi(e);
    if (const char* e = getenv("CRA_X3_SPLIT_MAX_BATCH")) x3_split_max_batch = atoi(e);
    no_small_path = getenv("CRA_NO_SMALL_PATH") != nullptr;
    own_stream = getenv("CRA_OWN_STREAM_PER_NET") != nullptr;
    if (const char* e = getenv("CRA_SMALL_BATCH_CONV_SPLIT")) small_conv_split = atoi(e);
}

// The streams nets work in.  The runtime binds every stream to one of GPU_MAX_HW_QUEUES (4) hardware queues -- the one with the fewest
// streams on it, whether those streams do anything or not -- and two streams of one queue run strictly one after the other
// (scripts/ubench/stream_queues.hip, profiles/r06/v_stream_queues.txt: two 2 ms kernels take 4.0 ms on streams 0 and 7 of eight, 2.0 ms on
// any two of different queues).  With a stream created per net, which queue two evaluator lanes (or two NeuralNetAPIUsers) shared was
// decided by how many nets the process had opened before and not yet closed: the same two-lane search measured 35k or 63k nodes/s, the
// same two predict() users 302k or 359k evals/s, depending on nets that were idle at the time (profiles/r06/t_*, u_*).  So the library
// keeps one stream per hardware queue and device, made together on first need and never destroyed (their queues stay four different
// ones), and a new net takes the one that has gone unused the longest: idle nets do not keep a queue busy, and up to four nets that work
// at the same time work on four queues.  More nets than queues share streams as they shared queues before -- in order, which is correct
// for everything a net does (each net's work is in-order in its stream; graphs are captured on a stream of their own, see capture()).
namespace {
struct NetStreams {
    std::mutex mu;
    int n = 0;
    hipStream_t s[16] = {};
    std::atomic<uint64_t> last[16] = {};
};
NetStreams g_net_streams[64];                  // per device
std::atomic<uint64_t> g_stream_tick{1};

int take_net_stream(int device, hipStream_t* out) {
    NetStreams& ns = g_net_streams[device];
    std::lock_guard<std::mutex> lk(ns.mu);
    if (ns.n == 0) {
        int n = 4;                              // the runtime's default number of hardware queues per process and device
        if (const char* e = getenv("GPU_MAX_HW_QUEUES")) n = atoi(e);
        n = n < 1 ? 1 : n > 16 ? 16 : n;
        for (int i = 0; i < n; ++i) HIP_CHECK(hipStreamCreateWithFlags(&ns.s[i], hipStreamNonBlocking));
        ns.n = n;
    }
    int best = 0;
    for (int i = 1; i < ns.n; ++i)
        if (ns.last[i].load(std::memory_order_relaxed) < ns.last[best].load(std::memory_order_relaxed)) best = i;
    ns.last[best].store(g_stream_tick.fetch_add(1, std::memory_order_relaxed), std::memory_order_relaxed);
    *out = ns.s[best];
    return best;
}
}  // namespace

void RiseNet::touch_stream() const {
    if (stream_slot_ >= 0) g_net_streams[device_].last[stream_slot_].store(g_stream_tick.fetch_add(1, std::memory_order_relaxed), std::memory_order_relaxed);
}

static thread_local hipStream_t g_companion_stream = nullptr;   // set by a constructor for the constructor of its companion net (same thread, next statement)
static thread_local int g_companion_slot = -1;

RiseNet::RiseNet(const std::string& model_path, int device_id, int batch_size, const std::string& precision)
    : device_(device_id), impl_(new Impl) {
    if (batch_size <= 0) throw std::invalid_argument("batch size must be positive");
    precision_arg_ = precision;
    std::string prec = precision;
    // "-3k": stem, tower and head as three launches instead of one (forward.hip); per-kernel timing and A/B reference
    if (prec.size() > 3 && prec.compare(prec.size() - 3, 3, "-3k") == 0) {
        one_launch_ = false;
        prec.resize(prec.size() - 3);
    }
    // "-1b" / "-2b": boards per workgroup of the dense residual tower (restower.hip); default by batch size
    if (prec.size() > 3 && prec.compare(prec.size() - 3, 3, "-8w") == 0) {   // dense tower: 8 thin waves instead of 4 fat ones
        rt_thin_waves_ = true;
        prec.resize(prec.size() - 3);
    }
    if (prec.size() > 4 && prec.compare(prec.size() - 4, 4, "-1wg") == 0) {  // float16x3 / float16p8: one workgroup per board also for small batches
        board_split_ = false;
        prec.resize(prec.size() - 4);
    }
    if (prec.size() > 3 && (prec.compare(prec.size() - 3, 3, "-1b") == 0 || prec.compare(prec.size() - 3, 3, "-2b") == 0)) {
        boards_per_wg_ = prec[prec.size() - 2] - '0';
        prec.resize(prec.size() - 3);
    }
    const std::string unfused_tag = "-unfused";   // layer-granular kernels (A/B reference for the fused block kernel)
    const std::string perblock_tag = "-perblock"; // one launch per bottleneck block (A/B reference for the tower kernel)
    if (prec.size() > unfused_tag.size() && prec.compare(prec.size() - unfused_tag.size(), unfused_tag.size(), unfused_tag) == 0) {
        fused_ = false;
        tower_ = false;
        prec.resize(prec.size() - unfused_tag.size());
    } else if (prec.size() > perblock_tag.size() && prec.compare(prec.size() - perblock_tag.size(), perblock_tag.size(), perblock_tag) == 0) {
        tower_ = false;
        prec.resize(prec.size() - perblock_tag.size());
    }
    if (prec == "float16" || prec == "fp16" || prec == "half") fp16_ = true;
    // Precision int8: the reference's calibrated reduced-precision mode (TensorRT INT8, entropy-calibrated on the plies of two recorded
    // games: tensorrtapi.cpp:334-360, chessbatchstream.cpp:44-94; UCI option Precision = int8).  Here: int8 operands in the two GEMMs of
    // every bottleneck block (v_mfma_i32_32x32x32_i8, tower.hip Q = 2), one activation step per tensor and block from a calibration pass
    // (mi_net_calibrate_int8 -> <model file>.int8calib beside the model, like TensorRT's calibration cache), one weight step per output
    // row; everything else as float16.  Round 6's study on int8 itself (scripts/studies/int8_calibration_study.py: value within 6 - 8e-3
    // of fp32, e4m3's 1 - 3e-2) replaced round 3's refusal, which rested on an e4m3 study.
    else if (prec == "int8") {
        fp16_ = true;
        fp8_tower_ = true;
        int8_ = true;
    }
    else if (prec == "fp8" || prec == "float8") {
        fp16_ = true;
        fp8_tower_ = true;
    }
    else if (prec == "float32" || prec == "fp32") fp16_ = false;
    // the fast mode that meets "logits within 1e-3 of fp32": float activations, every dense contraction as three f16 MFMAs on split
    // operands (x3.hip)
    else if (prec == "float16x3" || prec == "fp16x3" || prec == "f16x3") { fp16_ = false; x3_ = true; }
    // float16x3 with the cross terms of the one-launch tower's two 1x1 GEMMs on ONE e5m2 MFMA per 64 k and the residual stream in the PROJECT
    // waves' registers (x3.hip: tower_p8_kernel): logits within 3e-4 of fp32 (emulated 5e-5 ... 1.3e-4 on the parity nets)
    else if (prec == "float16p8" || prec == "fp16p8" || prec == "f16p8") { fp16_ = false; x3_ = true; p8_ = true; }
    else throw std::invalid_argument("unsupported precision '" + precision + "' (float16 | float16x3 | float16p8 | float32 | fp8 | int8)");
    design_.batch = batch_size;

    // model discovery (TensorrtAPI ctor, tensorrtapi.cpp:53-58)
    std::string dir, file;
    auto ends_with = [&](const char* ext) { const size_t n = strlen(ext); return model_path.size() > n && model_path.compare(model_path.size() - n, n, ext) == 0; };
    if (ends_with(".cranet") || ends_with(".onnx")) {
        const size_t sl = model_path.find_last_of('/');
        dir = sl == std::string::npos ? "./" : model_path.substr(0, sl + 1);
        file = sl == std::string::npos ? model_path : model_path.substr(sl + 1);
    } else {
        if (model_path.empty()) throw std::invalid_argument("The given directory must not be empty.");
        dir = model_path.back() == '/' ? model_path : model_path + "/";
        file = find_model_file(dir, batch_size);
    }
    model_name_ = file;
    model_file_path_ = dir + file;
    design_.version = read_version_from_string(model_name_);
    design_.game_phase = read_game_phase_from_string(dir);

    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) throw std::invalid_argument("device id out of range");
    HIP_CHECK(hipSetDevice(device_id));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) cu_count_ = cus;
    }

    NetFile nf;                                  // load_model: our container, or the reference's ONNX parsed in place (onnx_import.h)
    if (model_file_path_.size() > 5 && model_file_path_.compare(model_file_path_.size() - 5, 5, ".onnx") == 0) import_onnx(model_file_path_, nf);
    else nf.load(model_file_path_);
    if (nf.str("arch") != "rise") throw std::runtime_error("unsupported arch '" + nf.str("arch") + "' in " + model_file_path_);
    if (g_companion_stream) {                    // the companion net of a larger one works in ITS stream (never at the same time: a call goes to one of them)
        stream_ = g_companion_stream;
        stream_slot_ = g_companion_slot;
        owns_stream_ = false;
        g_companion_stream = nullptr;
    } else if (dev_.own_stream || device_id >= 64) {
        HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    } else {
        stream_slot_ = take_net_stream(device_id, &stream_);
    }
    if (int8_) {
        int8_calib_ = read_int8_calibration(model_file_path_);
        if (int8_calib_.empty())
            throw std::runtime_error("Precision int8 needs a calibration of this model: " + int8_calibration_path(model_file_path_) +
                                     " is missing -- mi_net_calibrate_int8 makes it (integration/hipapi.h does that with the engine's calibration positions)");
    }
    if (fp16_) build<half_t>(nf); else build<float>(nf);   // init_nn_design + load_parameters + buffers
    capture();                                   // bind_executor
    // the companion net for calls with few boards (rise_net.h: small_) is made HERE, on the thread that makes this net: made on first use it
    // was made by whichever SearchThread came first, and two of them making nets at once -- one capturing its graph, one uploading weights
    // through the legacy stream -- is an error of the runtime ("would make the legacy stream depend on a capturing blocking stream")
    // It shares this net's stream: a stream of its own shifted which hardware queue every later stream of the process got, and two lanes
    // of a later search landed on ONE queue (config 1 with two lanes: 35k nodes/s instead of 63k, profiles/r06/t_*).
    if (!dev_.no_small_path && x3_ && tower_ && fused_ && board_split_ && design_.batch > kBoardSplitMaxBatch) {
        g_companion_stream = stream_;
        g_companion_slot = stream_slot_;
        small_.reset(new RiseNet(model_file_path_, device_id, kBoardSplitMaxBatch, precision_arg_));
    }
}

static void turns_forget_stream(int device, hipStream_t s);     // below, next to RiseNet::Turn
// predict()s in flight per device (submit ... wait of any net): what decides between the two forms of a predict on pinned buffers (submit)
namespace {
std::atomic<int> g_predicts_in_flight[64];
}

RiseNet::~RiseNet() {
    if (counted_in_flight_) g_predicts_in_flight[device_].fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(device_);
    small_.reset();                              // (it works in this net's stream)
    if (stream_) {
        (void)hipStreamSynchronize(stream_);
        if (owns_stream_ && stream_slot_ < 0) turns_forget_stream(device_, stream_);
    }
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    impl_.reset();
    if (stream_ && owns_stream_ && stream_slot_ < 0) (void)hipStreamDestroy(stream_);
}

template <typename T> void RiseNet::build(const NetFile& nf) {
    Impl& im = *impl_;
    const int B = design_.batch;
    const int cin = int(nf.num("nb_input_channels"));
    const int C = int(nf.num("channels", 256));
    const int cop_init = int(nf.num("channels_operating_init"));
    const int cexp = int(nf.num("channel_expansion"));
    const int cv = int(nf.num("channels_value_head", 8));
    const int fc = int(nf.num("value_fc_size", 256));
    const int cp = int(nf.num("channels_policy_head"));
    const bool wdl = nf.num("use_wdl") != 0 && nf.num("use_plys_to_end") != 0;
    std::vector<std::string> kernels = nf.list("kernels"), se_types = nf.list("se_types");
    if (kernels.empty() || kernels.size() != se_types.size()) throw std::runtime_error("kernels/se_types mismatch in model file");
    // residual block family: RiseV3's mobile bottleneck (default), ClassicalResidualBlock (builder_util.py:401-434) or
    // AlphaZeroResnet's ResidualBlock (a0_resnet.py:72-107); the last two are towers of dense 3x3 convolutions
    const std::string conv_block = nf.str("conv_block", "mobile_bottlekneck_res_block");
    const bool dense_blocks = conv_block == "classical_res_block" || conv_block == "a0_res_block";
    if (!dense_blocks && conv_block != "mobile_bottlekneck_res_block") throw std::runtime_error("unsupported conv_block '" + conv_block + "'");
    // SE inside dense residual blocks (ClassicalResidualBlock(se_type), builder_util.py:401-434: gate on the block INPUT, hard-sigmoid;
    // AlphaZero ResidualBlock(use_se), a0_resnet.py:72-107: gate on the body OUTPUT, plain sigmoid): such nets run their blocks on the
    // layer kernels (conv GEMM + SE kernel), not on the one-launch dense tower
    bool dense_se = false;
    if (dense_blocks)
        for (const std::string& t : se_types) dense_se |= !(t == "none" || t.empty());
    if (C % 64 != 0 || C > 512) throw std::runtime_error("channels must be a multiple of 64 and <= 512");
    if (fc > 256 && fc % 256 != 0) throw std::runtime_error("unsupported value_fc_size");

    design_.nb_input_channels = cin;
    // _PolicyHead form (builder_util.py:206-243): policy map (the P planes, channel-major) or flat labels (Linear on top)
    const bool policy_map = nf.num("select_policy_from_plane", 1) != 0;
    const int n_labels = int(nf.num("n_labels", 0));
    if (!policy_map && (n_labels <= 0 || (cp * kSquares) % 32 != 0)) throw std::runtime_error("flat policy head needs n_labels and P*64 % 32 == 0");
    design_.nb_policy = policy_map ? cp * kSquares : n_labels;
    design_.nb_aux = wdl ? 4 : 0;
    const int cin_pad = round_up(cin, 32);
    im.cin_pad = cin_pad;

    // C_op schedule: rise_mobile_v3.py:36-78 (kernel_5_channel_ratio=None)
    std::vector<int> cops, ks;
    int cop_run = cop_init, cop_max = 32;
    const std::vector<std::string> cop_list = nf.list("channels_operating");     // imported models carry the widths they were found with
    if (!cop_list.empty() && cop_list.size() != kernels.size()) throw std::runtime_error("channels_operating/kernels mismatch in model file");
    for (size_t i = 0; i < kernels.size(); ++i) {
        const int k = std::stoi(kernels[i]);
        if (k != 3 && k != 5) throw std::runtime_error("unsupported depthwise kernel size " + kernels[i]);
        const int c = !cop_list.empty() ? std::stoi(cop_list[i]) : k == 5 ? cop_run - 32 * int(i / 2) : cop_run;
        if (c % 32 != 0 || c <= 0) throw std::runtime_error("channels_operating must be a positive multiple of 32");
        cops.push_back(c);
        ks.push_back(k);
        cop_max = std::max(cop_max, c);
        cop_run += cexp;
    }

    // ---- device buffers ----
    d_desc_ = im.dalloc(size_t(B) * sizeof(BoardDesc));
    d_planes_ = static_cast<float*>(im.dalloc(size_t(B) * cin * kSquares * sizeof(float)));
    d_value_ = static_cast<float*>(im.dalloc(size_t(B) * sizeof(float)));
    d_probs_ = static_cast<float*>(im.dalloc(size_t(B) * design_.nb_policy * sizeof(float)));
    d_logits_ = static_cast<float*>(im.dalloc(size_t(B) * design_.nb_policy * sizeof(float)));
    d_aux_ = wdl ? static_cast<float*>(im.dalloc(size_t(B) * 4 * sizeof(float))) : nullptr;
    T* x0 = static_cast<T*>(im.dalloc(size_t(B) * kSquares * cin_pad * sizeof(T)));
    T* a0 = static_cast<T*>(im.dalloc(size_t(B) * kSquares * C * sizeof(T)));
    T* a1 = static_cast<T*>(im.dalloc(size_t(B) * kSquares * C * sizeof(T)));
    T* e = static_cast<T*>(im.dalloc(size_t(B) * kSquares * cop_max * sizeof(T)));
    T* f = static_cast<T*>(im.dalloc(size_t(B) * kSquares * cop_max * sizeof(T)));

    double macs = 0;
    // packed A-fragment images of a dense layer: T, or the f16 hi / lo pair of Precision float16x3
    auto set_conv_weights = [&](ConvArgs& c, const Folded& fd, int co, int ci, int k, int co_pad, int ci_pad, bool p8 = false) {
        if (p8 && p8_ && k == 3 && ci_pad % 128 == 0) {     // Precision float16p8: the policy head's 3x3 convs (x3.hip: conv3x3_p8_kernel)
            double inv = 1.0;
            SplitPack sp = pack_dense_p8(fd, co, ci, k, co_pad, ci_pad, &inv);
            c.wpk = im.upload(sp.hi);
            c.wpk_lo = im.upload(sp.lo);
            c.p8 = 1;
            c.acc_scale = float(inv);
        } else if (x3_) {
            SplitPack sp = pack_dense_split(fd, co, ci, k, co_pad, ci_pad);
            c.wpk = im.upload(sp.hi);
            c.wpk_lo = im.upload(sp.lo);
        } else {
            c.wpk = im.upload(pack_dense<T>(fd, co, ci, k, co_pad, ci_pad));
        }
    };
    auto add_conv = [&](const std::string& conv, const std::string& bn, const T* x, T* out, const T* resid, int ci, int ci_pad,
                        int co, int k, int relu, float* out_policy, bool p8 = false) {
        Folded fd = fold_bn(nf, conv, bn);
        const int co_pad = round_up(co, 16);
        Op op;
        op.kind = OpKind::Conv;
        op.conv.x = x;
        set_conv_weights(op.conv, fd, co, ci, k, co_pad, ci_pad, p8);
        op.conv.bias = im.upload_d2f(fd.b, co_pad);
        op.conv.resid = resid;
        op.conv.out = out_policy ? static_cast<void*>(out_policy) : static_cast<void*>(out);
        op.conv.batch = B;
        op.conv.cin = ci_pad;
        op.conv.cout_pad = co_pad;
        op.conv.cout_real = co;
        op.conv.cout_ld = co_pad;
        op.conv.ks = k;
        op.conv.relu = relu;
        op.conv.out_policy_f32 = out_policy ? 1 : 0;
        im.ops.push_back(op);
        macs += double(kSquares) * ci * co * k * k;
    };

    const int cin_pad16 = std::max(48, round_up(cin, 16));
    if (tower_ && fused_ && std::is_same<T, half_t>::value && C == 256 && cin_pad16 <= 96) {
        // stem kernel: planes -> conv3x3 + BN + ReLU -> NHWC f16 in one launch (stem.hip)
        Folded fs = fold_bn(nf, "body_spatial.0.body.0", "body_spatial.0.body.1");
        const int nks = cin_pad16 / 16;
        std::vector<half_t> sw;
        std::vector<float> sb;
        for (int wv = 0; wv < 8; ++wv) {
            for (int tap = 0; tap < 9; ++tap)
                for (int ksx = 0; ksx < nks; ++ksx)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int co = wv * 32 + (l & 31), ci = ksx * 16 + (l >> 5) * 8 + j;
                            sw.push_back(half_t(ci < cin ? float(fs.w[(size_t(co) * cin + ci) * 9 + tap]) : 0.f));
                        }
            sw.insert(sw.end(), size_t(16) * 512, half_t(0.f));
            for (int lh = 0; lh < 2; ++lh)
                for (int v = 0; v < 16; ++v) sb.push_back(float(fs.b[wv * 32 + (v % 4) + 8 * (v / 4) + 4 * lh]));
        }
        Op op;
        op.kind = OpKind::Stem;
        op.st.planes = d_planes_;
        op.st.x = a0;
        op.st.stem_w = im.upload(sw);
        op.st.stem_b = im.upload(sb);
        op.st.stem_wave_frags = 9 * nks + 16;
        op.st.cin = cin;
        op.st.cin_pad = cin_pad16;
        op.st.batch = B;
        im.ops.push_back(op);
        macs += double(kSquares) * cin * C * 9;
    } else {
        if (!x3_) {   // input layout transform (Precision float16x3: the stem conv reads the planes itself)
            Op op;
            op.kind = OpKind::PlanesToAct;
            op.x = d_planes_;
            op.y = x0;
            op.C = cin;
            im.ops.push_back(op);
        }
        add_conv("body_spatial.0.body.0", "body_spatial.0.body.1", x0, a0, nullptr, cin, cin_pad, C, 3, true, nullptr);   // _Stem
        if (x3_) {
            im.ops.back().from_planes = true;
            im.ops.back().conv.planes_c = cin;
        }
    }
    T *cur = a0, *nxt = a1;
    // SE plumbing for the fused paths: the squeeze (per-channel sums) is produced by the previous block / tower kernel's
    // epilogue, a small gate kernel turns it into gate[b][c], and the consumer's prologue multiplies it into x while
    // loading the tile.  Inside a tower the whole SE runs in-kernel.
    float *se_pool = nullptr, *se_gate = nullptr;
    const float* pending_gate = nullptr;
    int prod_op = -1;                 // last op that produced the residual stream and can emit its channel sums
    constexpr bool kHalf = std::is_same<T, half_t>::value;
    const bool tower_ok = tower_ && fused_ && kHalf && C == 256 && !dense_se;
    if (fp8_tower_ && (!tower_ok || dense_blocks))
        throw std::runtime_error("Precision fp8 runs on the one-launch bottleneck tower only (256-channel RISE nets): use float16 for this model");
    std::vector<TowerBlockDesc> tower_blocks;
    std::vector<half_t> tower_ws[4];          // per matrix wave: MFMA A fragments in consumption order (kernels.h: TowerArgs)
    std::vector<float> tower_bs[4];
    std::vector<half_t> tower_ps[4];          // per vector wave: packed f16 depthwise weights (kernels.h: pstream)
    std::vector<uint8_t> tower_w8e[4], tower_w8p[4];   // Precision fp8: per matrix wave the expand / project streams (kernels.h: TowerArgs::fp8)
    const float* tower_gate = nullptr;
    auto flush_tower = [&]() {
        if (tower_blocks.empty()) return;
        Op op;
        op.kind = OpKind::Tower;
        op.tw.x = cur;
        op.tw.y = nxt;
        op.tw.blocks = im.upload(tower_blocks);
        op.tw.nblocks = int(tower_blocks.size());
        {   // close the streams: the kernel's windows run one window / one chunk past the end
            std::vector<half_t> ws, ps;
            std::vector<float> bs;
            std::vector<uint8_t> w8;
            if (fp8_tower_) {
                for (int w = 0; w < 4; ++w) {        // [expand stream + a window of zeros][project stream + a window of zeros]
                    tower_w8e[w].resize(tower_w8e[w].size() + 8 * 1024, 0);
                    tower_w8p[w].resize(tower_w8p[w].size() + 8 * 1024, 0);
                    w8.insert(w8.end(), tower_w8e[w].begin(), tower_w8e[w].end());
                    w8.insert(w8.end(), tower_w8p[w].begin(), tower_w8p[w].end());
                }
                op.tw.fp8 = int8_ ? 2 : 1;
                op.tw.wstream_e_frags = (long long)(tower_w8e[0].size() / 1024);
            }
            for (int w = 0; w < 4; ++w) {
                tower_ws[w].resize(tower_ws[w].size() + size_t(kTowerWindow) * 512, half_t(0.f));
                tower_ps[w].resize(tower_ps[w].size() + 1024, half_t(0.f));
                tower_bs[w].resize(tower_bs[w].size() + 32, 0.f);
                ws.insert(ws.end(), tower_ws[w].begin(), tower_ws[w].end());
                bs.insert(bs.end(), tower_bs[w].begin(), tower_bs[w].end());
                ps.insert(ps.end(), tower_ps[w].begin(), tower_ps[w].end());
            }
            if (fp8_tower_) op.tw.wstream = im.upload(w8);
            else op.tw.wstream = im.upload(ws);
            op.tw.bstream = im.upload(bs);
            op.tw.pstream = im.upload(ps);
            op.tw.wstream_wave_frags = fp8_tower_ ? (long long)((tower_w8e[0].size() + tower_w8p[0].size()) / 1024) : (long long)(tower_ws[0].size() / 512);
            op.tw.bstream_wave_floats = (long long)tower_bs[0].size();
            op.tw.pstream_wave_bytes = (long long)(tower_ps[0].size() * sizeof(half_t));
            for (int w = 0; w < 4; ++w) { tower_ws[w].clear(); tower_bs[w].clear(); tower_ps[w].clear(); tower_w8e[w].clear(); tower_w8p[w].clear(); }
        }
        op.tw.batch = B;
        op.tw.gate_in = tower_gate;
        if (getenv("CRA_TOWER_TRACE")) op.tw.trace = static_cast<unsigned long long*>(im.dalloc(2 * 256 * sizeof(unsigned long long)));
        prod_op = int(im.ops.size());
        im.ops.push_back(op);
        tower_blocks.clear();
        tower_gate = nullptr;
        std::swap(cur, nxt);
    };
    // a block runs on the fused per-block kernel (else on the layer kernels): Precision float16x3 has a fused kernel for 3x3 blocks only
    auto block_fused = [&](int k) { return fused_ && C == 256 && !(x3_ && k != 3); };
    auto add_se = [&](Op op, bool consumer_fused) {
        if (consumer_fused && prod_op >= 0) {
            if (!se_pool) {
                se_pool = static_cast<float*>(im.dalloc(size_t(B) * C * sizeof(float)));
                se_gate = static_cast<float*>(im.dalloc(size_t(B) * C * sizeof(float)));
            }
            if (im.ops[prod_op].kind == OpKind::Tower) im.ops[prod_op].tw.pool_out = se_pool;
            else im.ops[prod_op].blk.pool_out = se_pool;
            op.kind = OpKind::SEGate;
            op.x = se_pool;
            op.y = se_gate;
            pending_gate = se_gate;
        } else {
            op.kind = OpKind::SE;      // in-place scaling kernel (input produced by the stem conv, or layer-granular path)
            op.y = cur;
        }
        im.ops.push_back(op);
    };
    // Precision float16x3: runs of consecutive 3x3 blocks in one launch (x3.hip: tower_x3_kernel)
    std::vector<X3TowerBlock> x3_blocks;
    int x3_run_ks = 3;                     // a run is all 3x3 or all 5x5 blocks (tower_x3_roles_kernel<KS>, tower_p8_kernel<KS>)
    // small batches: 3x3 runs one block per launch, several workgroups per board (kernels.h: X3SplitArgs)
    const bool x3_split = x3_ && tower_ && fused_ && C == 256 && board_split_ && B <= (dev_.x3_split_max_batch > 0 ? dev_.x3_split_max_batch : kBoardSplitMaxBatch);
    if (x3_split)
        for (Op& o : im.ops)
            if (o.kind == OpKind::Conv && o.from_planes) o.conv.few_boards = dev_.small_conv_split;      // the stem's couts over several workgroups per board
    float* split_parts[2] = {nullptr, nullptr};
    constexpr int kSplitMaxG = 10;
    auto flush_x3_tower = [&]() {
        if (x3_blocks.empty()) return;
        if (x3_split && x3_run_ks == 3) {
            if (!split_parts[0])
                for (auto& q : split_parts) q = static_cast<float*>(im.dalloc(size_t(B) * kSplitMaxG * kSquares * C * sizeof(float)));
            int max_g = std::max(1, std::min(kSplitMaxG, cu_count_ / B));
            if (dev_.x3_split_max_g > 0) max_g = std::min(max_g, dev_.x3_split_max_g);
            const int nb = int(x3_blocks.size());
            int gin = 1;
            for (int k = 0; k < nb; ++k) {
                Op op;
                op.kind = OpKind::BlockX3Split;
                op.xs.blk = x3_blocks[k];
                op.xs.x_parts = k == 0 ? reinterpret_cast<const float*>(cur) : split_parts[k % 2];
                op.xs.y_parts = split_parts[(k + 1) % 2];
                op.xs.gin = gin;
                op.xs.batch = B;
                op.xs.G = std::min(max_g, x3_blocks[k].cop_pad / block_x3_chunk_channels());
                op.xs.dev = dev_.x3_split_dev;
                if (k > 0 && x3_blocks[k].se_kind != 0 && !(dev_.x3_split_dev & 8)) {      // the launch before a gated block leaves its images' channel sums
                    float* pools = static_cast<float*>(im.dalloc(size_t(B) * kSplitMaxG * C * sizeof(float)));
                    im.ops.back().xs.pool_out = pools;
                    op.xs.pool_in = pools;
                }
                gin = op.xs.G;
                im.ops.push_back(op);
            }
            Op fin;
            fin.kind = OpKind::X3SplitFinish;
            fin.xs.x_parts = split_parts[nb % 2];
            fin.xs.gin = gin;
            fin.xs.batch = B;
            fin.xs_y = reinterpret_cast<float*>(nxt);
            im.ops.push_back(fin);
            x3_blocks.clear();
            prod_op = -1;
            std::swap(cur, nxt);
            return;
        }
        Op op;
        op.kind = OpKind::TowerX3;
        op.tx.x = reinterpret_cast<const float*>(cur);
        op.tx.y = reinterpret_cast<float*>(nxt);
        op.tx.blocks = im.upload(x3_blocks);
        op.tx.nblocks = int(x3_blocks.size());
        op.tx.batch = B;
        op.tx.p8 = p8_ ? 1 : 0;
        op.tx.ks = x3_run_ks;
        op.tx.symmetric = dev_.x3_symmetric ? 1 : 0;
        im.ops.push_back(op);
        x3_blocks.clear();
        prod_op = -1;                      // this launch does not emit channel sums: a gate behind it is an SE launch of its own
        std::swap(cur, nxt);
    };
    auto to_half = [](const std::vector<float>& v) {
        std::vector<half_t> h(v.size());
        for (size_t i = 0; i < v.size(); ++i) h[i] = half_t(v[i]);
        return h;
    };
    // tower kernel, SE gate weights in thread order: thread t of 512 reads its 32 half2 weights as 8 coalesced 16-byte loads,
    // load i of thread t at uint4 index i*512 + t (tower.hip: se_phase).  idx(t, k) = half2 index of thread t's k-th weight.
    auto pack_se_threads = [](const std::vector<float>& src, auto idx) {
        std::vector<half_t> out(size_t(8) * 512 * 8);
        for (int i = 0; i < 8; ++i)
            for (int t = 0; t < 512; ++t)
                for (int j = 0; j < 4; ++j) {
                    const size_t h2 = idx(t, 4 * i + j);
                    out[((size_t(i) * 512 + t) * 4 + j) * 2 + 0] = half_t(src[2 * h2]);
                    out[((size_t(i) * 512 + t) * 4 + j) * 2 + 1] = half_t(src[2 * h2 + 1]);
                }
        return out;
    };
    // the float16x3 tower's gate matrices in thread order (x3.hip: x3_se_phase): thread t of 512 reads 16 float4, load i at float4 index
    // i * 512 + t = (a[2i], b[2i], a[2i+1], b[2i+1]) of its two output rows a, b; w(row, k) returns the matrix entry for the thread's k-th input
    auto pack_se_threads_f32 = [](auto w) {
        std::vector<float> out(size_t(16) * 512 * 4);
        for (int i = 0; i < 16; ++i)
            for (int t = 0; t < 512; ++t)
                for (int e = 0; e < 4; ++e) out[(size_t(i) * 512 + t) * 4 + e] = w(t, e & 1, 2 * i + (e >> 1));
        return out;
    };
    if (dense_blocks && tower_ok) {
        // all blocks in one launch (restower.hip; stream layouts in kernels.h: ResTowerArgs)
        if constexpr (kHalf) {
            // wave shape (restower.hip): 4 fat waves of 64 couts by default, "-8w" = 8 waves of 32 couts (the first version)
            const int NR = rt_thin_waves_ ? 1 : 2, n_waves = 8 / NR;
            std::vector<half_t> ws;
            std::vector<float> bs;
            std::vector<Folded> f1s, f2s;
            for (size_t i = 0; i < cops.size(); ++i) {
                const std::string p = "body_spatial." + std::to_string(i + 1);
                f1s.push_back(fold_bn(nf, p + ".body.0", p + ".body.1"));
                f2s.push_back(fold_bn(nf, p + ".body.3", p + ".body.4"));
            }
            for (int wv = 0; wv < n_waves; ++wv) {
                for (size_t i = 0; i < cops.size(); ++i) {
                    for (int cv2 = 0; cv2 < 2; ++cv2) {
                        const Folded& fd = cv2 ? f2s[i] : f1s[i];
                        for (int tap = 0; tap < 9; ++tap)
                            for (int ksx = 0; ksx < 16; ++ksx)
                                for (int rt = 0; rt < NR; ++rt)
                                    for (int l = 0; l < 64; ++l)
                                        for (int j = 0; j < 8; ++j) {
                                            const int co = (wv * NR + rt) * 32 + (l & 31), kpos = ksx * 16 + (l >> 5) * 8 + j;
                                            const int ci = cv2 ? (kpos / 32) * 32 + tower_row_of_position(kpos % 32) : kpos;
                                            ws.push_back(half_t(float(fd.w[(size_t(co) * C + ci) * 9 + tap])));
                                        }
                        for (int rt = 0; rt < NR; ++rt)
                            for (int lh = 0; lh < 2; ++lh)
                                for (int v = 0; v < 16; ++v) bs.push_back(float(fd.b[(wv * NR + rt) * 32 + (v % 4) + 8 * (v / 4) + 4 * lh]));
                    }
                }
                ws.insert(ws.end(), size_t(16) * 512, half_t(0.f));
            }
            Op op;
            op.kind = OpKind::ResTower;
            op.rt.x = cur;
            op.rt.y = nxt;
            op.rt.wstream = im.upload(ws);
            op.rt.bstream = im.upload(bs);
            op.rt.wstream_wave_frags = (long long)(cops.size() * 2 * 9 * 16 * NR + 16);
            op.rt.bstream_wave_floats = (long long)(cops.size() * 64 * NR);
            op.rt.cout_tiles_per_wave = NR;
            op.rt.nblocks = int(cops.size());
            op.rt.relu_after_add = conv_block == "a0_res_block" ? 1 : 0;
            op.rt.batch = B;
            // two boards per workgroup halve the weight stream per board but fill only B/2 CUs: from 512 boards on, or on request
            // (two evaluator lanes of 256 keep 512 boards in flight)
            op.rt.boards_per_workgroup = boards_per_wg_ ? boards_per_wg_ : (B >= 512 ? 2 : 1);
            im.ops.push_back(op);
            macs += double(cops.size()) * 2.0 * kSquares * C * C * 9;
            std::swap(cur, nxt);
        }
    }
    // gate of a dense block as an in-place SE op on `target` (+ optional shortcut `res`: target = relu(res + target * gate))
    auto dense_se_op = [&](const std::string& p, const std::string& type, T* target, const T* res, bool plain_sigmoid) {
        Op op;
        op.kind = OpKind::SE;
        op.C = C;
        op.y = target;
        op.x = res;
        if (type == "ca_se" || type == "se") {
            const TensorView &w1 = nf.get(p + ".se.fc.0.weight"), &w2 = nf.get(p + ".se.fc.2.weight");
            const int H = C / 2;
            std::vector<float> w1t(size_t(C) * H), w2t(size_t(H) * C);
            for (int j = 0; j < H; ++j) for (int c = 0; c < C; ++c) w1t[size_t(c) * H + j] = w1.data[size_t(j) * C + c];
            for (int c = 0; c < C; ++c) for (int j = 0; j < H; ++j) w2t[size_t(j) * C + c] = w2.data[size_t(c) * H + j];
            op.se_kind = 1;
            op.w0 = im.upload(w1t);
            op.w1 = im.upload(w2t);
            macs += 2.0 * C * H;
        } else if (type == "eca_se") {
            const TensorView& w = nf.get(p + ".se.body.0.weight");
            const int kk = int(w.shape[2]), mid = kk / 2;
            std::vector<float> wt(size_t(C) * C), b(C);
            for (int o = 0; o < C; ++o) for (int c = 0; c < C; ++c) wt[size_t(c) * C + o] = w.data[(size_t(o) * C + c) * kk + mid];
            const float* bs = nf.get(p + ".se.body.0.bias").data;
            for (int o = 0; o < C; ++o) b[o] = bs[o];
            op.se_kind = 2;
            op.w0 = im.upload(wt);
            op.b0 = im.upload(b);
            macs += double(C) * C;
        } else {
            throw std::runtime_error("unsupported se_type " + type);
        }
        if (plain_sigmoid) op.se_kind |= 16;
        im.ops.push_back(op);
    };
    for (size_t i = 0; dense_blocks && !tower_ok && i < cops.size(); ++i) {
        // x -> conv3x3 + BN + ReLU -> conv3x3 + BN -> classical: x + ReLU(.)   a0: ReLU(x + .)
        const std::string p = "body_spatial." + std::to_string(i + 1);
        const bool gated = !(se_types[i] == "none" || se_types[i].empty());
        const bool a0 = conv_block == "a0_res_block";
        if (gated && !a0) dense_se_op(p, se_types[i], cur, nullptr, false);       // classical: x = se(x) first (builder_util.py:431-433)
        add_conv(p + ".body.0", p + ".body.1", cur, nxt, nullptr, C, C, C, 3, 1, nullptr);
        T* out = e;                                   // e: scratch of at least C channels per square
        if (gated && a0) {
            // out = BN(conv(.)) without shortcut, then out = relu(x + se(out)) in the gate kernel (a0_resnet.py:104-107)
            add_conv(p + ".body.3", p + ".body.4", nxt, out, nullptr, C, C, C, 3, 0, nullptr);
            dense_se_op(p, se_types[i], out, cur, true);
        } else {
            add_conv(p + ".body.3", p + ".body.4", nxt, out, cur, C, C, C, 3, conv_block == "classical_res_block" ? 2 : 1, nullptr);
        }
        // keep (cur, nxt) = (block output, scratch): rotate the three buffers
        T* old = cur;
        cur = out;
        e = old;
    }
    for (size_t i = 0; !dense_blocks && i < cops.size(); ++i) {
        const std::string p = "body_spatial." + std::to_string(i + 1);
        const int cop = cops[i], k = ks[i];
        const bool in_tower = tower_ok && (k == 3 || k == 5);
        if (!in_tower) flush_tower();
        const bool se_in_kernel = in_tower && !tower_blocks.empty();
        TowerBlockDesc td{};
        // the 5x5 blocks (RISEv3.3) run in tower launches of their own (tower_*_kernel<5>); Precision float16p8 also computes a run's first gate in the launch
        const bool in_x3_tower = x3_ && tower_ && fused_ && C == 256 && (k == 3 || k == 5);
        if (!in_x3_tower || (!x3_blocks.empty() && x3_run_ks != k)) flush_x3_tower();
        if (in_x3_tower && x3_blocks.empty()) x3_run_ks = k;
        const bool split_block = in_x3_tower && x3_split && k == 3;        // small batches: this block runs on block_x3_split_kernel (float16x3 images, own gate)
        const bool x3_se_in_kernel = in_x3_tower && (!x3_blocks.empty() || p8_ || split_block);  // float16x3: the first block of a run takes its gate from an SE launch
        X3TowerBlock xb{};
        if (se_types[i] == "ca_se" || se_types[i] == "se") {           // _ChannelAttentionModule, builder_util.py:83-114
            const TensorView &w1 = nf.get(p + ".se.fc.0.weight"), &w2 = nf.get(p + ".se.fc.2.weight");
            const int H = C / 2;
            std::vector<float> w1t(size_t(C) * H), w2t(size_t(H) * C);
            for (int j = 0; j < H; ++j) for (int c = 0; c < C; ++c) w1t[size_t(c) * H + j] = w1.data[size_t(j) * C + c];
            for (int c = 0; c < C; ++c) for (int j = 0; j < H; ++j) w2t[size_t(j) * C + c] = w2.data[size_t(c) * H + j];
            if (x3_se_in_kernel) {
                xb.se_kind = 1;
                // FC1: thread t -> hidden rows 2*(t/8), +1 over inputs c = 32*(t%8) + k;  FC2: gate rows 2*(t/4), +1 over hidden j = 32*(t%4) + k
                xb.se_w1t = im.upload(pack_se_threads_f32([&](int t, int row, int k) { return w1t[size_t(32 * (t & 7) + k) * H + 2 * (t >> 3) + row]; }));
                xb.se_w2t = im.upload(pack_se_threads_f32([&](int t, int row, int k) { return w2t[size_t(32 * (t & 3) + k) * C + 2 * (t >> 2) + row]; }));
            } else if (se_in_kernel) {
                td.se_kind = 1;
                // FC1: thread t -> outputs 2*(t/8), +1 over inputs c in [32*(t%8), +32); FC2: outputs 2*(t/4), +1 over j in [32*(t%4), +32):
                // the threads of an output pair are neighbouring lanes (in-wave reduction, tower.hip: se_phase)
                td.se_w1 = im.upload(pack_se_threads(w1t, [](int t, int k) { return size_t((t & 7) * 32 + k) * 64 + (t >> 3); }));
                td.se_w2 = im.upload(pack_se_threads(w2t, [](int t, int k) { return size_t((t & 3) * 32 + k) * 128 + (t >> 2); }));
            } else {
                Op op;
                op.se_kind = 1;
                op.w0 = im.upload(w1t);
                op.w1 = im.upload(w2t);
                op.C = C;
                add_se(op, block_fused(k) && !in_x3_tower);
            }
            macs += 2.0 * C * H;
        } else if (se_types[i] == "eca_se") {                           // _EfficientChannelAttentionModule, builder_util.py:49-80
            const TensorView& w = nf.get(p + ".se.body.0.weight");     // [C][C][kk]; the length-1 sequence only sees the centre tap
            const int kk = int(w.shape[2]), mid = kk / 2;
            std::vector<float> wt(size_t(C) * C), b(C);
            for (int o = 0; o < C; ++o) for (int c = 0; c < C; ++c) wt[size_t(c) * C + o] = w.data[(size_t(o) * C + c) * kk + mid];
            const float* bs = nf.get(p + ".se.body.0.bias").data;
            for (int o = 0; o < C; ++o) b[o] = bs[o];
            if (x3_se_in_kernel) {
                xb.se_kind = 2;
                // thread t -> gate rows 2*(t/4), +1 over inputs i = 64*(t%4) + k: the first 32 inputs, then (second image) the other 32
                std::vector<float> pk = pack_se_threads_f32([&](int t, int row, int k) { return wt[size_t(64 * (t & 3) + k) * C + 2 * (t >> 2) + row]; });
                const std::vector<float> pk2 = pack_se_threads_f32([&](int t, int row, int k) { return wt[size_t(64 * (t & 3) + 32 + k) * C + 2 * (t >> 2) + row]; });
                pk.insert(pk.end(), pk2.begin(), pk2.end());
                xb.se_w1t = im.upload(pk);
                xb.se_b = im.upload(b);
            } else if (se_in_kernel) {
                td.se_kind = 2;
                // thread t -> outputs 2*(t/4), +1 over inputs i in [64*(t%4), +64): first 32 inputs, then the second 32
                std::vector<half_t> pk = pack_se_threads(wt, [](int t, int k) { return size_t((t & 3) * 64 + k) * 128 + (t >> 2); });
                const std::vector<half_t> pk2 = pack_se_threads(wt, [](int t, int k) { return size_t((t & 3) * 64 + 32 + k) * 128 + (t >> 2); });
                pk.insert(pk.end(), pk2.begin(), pk2.end());
                td.se_w1 = im.upload(pk);
                td.se_b = im.upload(b);
            } else {
                Op op;
                op.se_kind = 2;
                op.w0 = im.upload(wt);
                op.b0 = im.upload(b);
                op.C = C;
                add_se(op, block_fused(k) && !in_x3_tower);
            }
            macs += double(C) * C;
        } else if (se_types[i] != "none" && !se_types[i].empty()) {
            throw std::runtime_error("unsupported se_type " + se_types[i]);
        }
        if (in_tower) {
            // residual tower: this block joins the current run of 3x3 blocks (one launch per run, kernels.h: TowerArgs)
            if constexpr (kHalf) {
                const int cop_pad = round_up(cop, 128);
                Folded f1 = fold_bn(nf, p + ".body.0", p + ".body.1");
                Folded f2 = fold_bn(nf, p + ".body.3", p + ".body.4");
                Folded f3 = fold_bn(nf, p + ".body.6", p + ".body.7");
                const int n = cop_pad / 128;
                // Precision fp8: power-of-two scale per expand channel / per cout; s1 goes into the depthwise weights (ReLU commutes with
                // a positive factor), b1 / s1 is where the expand accumulator starts, y = x + s3 * (acc + b3 / s3)
                // Precision int8 (oracle/rise_oracle.py: int8_block is the definition): s1 / s3 = max |row| / 127, weights rounded half to
                // even; with the block's calibrated activation steps 1 / qx_inv (stream) and 1 / qt_inv (depthwise output) a unit of the
                // expand accumulator is worth k1 = s1 / qx_inv, of the project accumulator k3 = s3 / qt_inv: the biases enter the
                // accumulators as round(b / k), t1 = relu(acc) * 2^-7, k1 * 2^7 goes into the depthwise weights, y = x + k3 * acc
                std::vector<double> s1(size_t(cop_pad), 1.0), s3(size_t(C), 1.0);
                std::vector<double> k1(size_t(cop_pad), 1.0), k3(size_t(C), 1.0);
                constexpr double kInt8Escale = 1.0 / 128.0;
                double qx_inv = 0.0, qt_inv = 0.0;
                if (int8_) {
                    if (i >= int8_calib_.size()) throw std::runtime_error("INT8 calibration file holds fewer blocks than the model");
                    qx_inv = double(float(half_t(float(127.0 / std::max(double(int8_calib_[i].first), 1e-6)))));     // f16 numbers: the kernel's quantiser multiplies in f16
                    qt_inv = double(float(half_t(float(255.0 / std::max(double(int8_calib_[i].second), 1e-6)))));
                    td.qx_inv = float(qx_inv);
                    td.qt_inv = float(qt_inv);
                    td.escale = float(kInt8Escale);
                }
                auto weight_byte = [&](double v) -> uint8_t {               // v = w / row step
                    if (!int8_) return to_e4m3(v);
                    const double r = std::max(-127.0, std::min(127.0, std::nearbyint(v)));
                    return uint8_t(int8_t(int(r)));
                };
                if (fp8_tower_) {
                    for (int ch = 0; ch < cop; ++ch) {
                        double m = 0;
                        for (int k2 = 0; k2 < C; ++k2) m = std::max(m, std::fabs(f1.w[size_t(ch) * C + k2]));
                        s1[ch] = int8_ ? std::max(m, 1e-30) / 127.0 : row_scale_pow2(m);
                        k1[ch] = s1[ch] / qx_inv;
                    }
                    for (int co = 0; co < C; ++co) {
                        double m = 0;
                        for (int ch = 0; ch < cop; ++ch) m = std::max(m, std::fabs(f3.w[size_t(co) * cop + ch]));
                        s3[co] = int8_ ? std::max(m, 1e-30) / 127.0 : row_scale_pow2(m);
                        k3[co] = s3[co] / qt_inv;
                    }
                    for (int w = 0; w < 4; ++w)
                        for (int c = 0; c < n; ++c) {
                            for (int ks = 0; ks < 4; ++ks)           // expand: [k-step of 64][half][lane][16 B]
                                for (int hf = 0; hf < 2; ++hf)
                                    for (int l = 0; l < 64; ++l)
                                        for (int t = 0; t < 16; ++t) {
                                            const int ch = c * 128 + w * 32 + (l & 31), k2 = ks * 64 + (l >> 5) * 32 + hf * 16 + t;
                                            tower_w8e[w].push_back(ch < cop ? weight_byte(f1.w[size_t(ch) * C + k2] / s1[ch]) : uint8_t(0));
                                        }
                            for (int ks = 0; ks < 2; ++ks)           // project: [k-step of 64][row tile][half][lane][16 B]
                                for (int rt = 0; rt < 2; ++rt)
                                    for (int hf = 0; hf < 2; ++hf)
                                        for (int l = 0; l < 64; ++l)
                                            for (int t = 0; t < 16; ++t) {
                                                const int co = w * 64 + rt * 32 + (l & 31);
                                                const int ch = tower_k_channel(c * 128 + ks * 64 + (l >> 5) * 32 + hf * 16 + t);
                                                tower_w8p[w].push_back(ch < cop ? weight_byte(f3.w[size_t(co) * cop + ch] / s3[co]) : uint8_t(0));
                                            }
                        }
                    if (int8_) {
                        // project accumulators: int32, started at round(b3 / k3) + 128 x the row's weight sum (the depthwise output u is held as
                        // u - 128); y = x + k3 * acc.  The bit patterns travel in the float arrays the fp8 path uses.
                        std::vector<float> b3bits, k3f;
                        b3bits.resize(size_t(C));
                        k3f.resize(size_t(C));
                        for (int co = 0; co < C; ++co) {
                            long long rowsum = 0;
                            for (int ch = 0; ch < cop; ++ch) rowsum += (long long)(int8_t(weight_byte(f3.w[size_t(co) * cop + ch] / s3[co])));
                            const int32_t start = int32_t(std::nearbyint(f3.b[co] / k3[co])) + int32_t(128 * rowsum);
                            std::memcpy(&b3bits[co], &start, 4);
                            k3f[co] = float(k3[co]);
                        }
                        td.s3 = im.upload(k3f);
                        td.b3 = im.upload(b3bits);
                    } else {
                        for (int co = 0; co < C; ++co) f3.b[co] /= s3[co];
                        std::vector<float> s3f(s3.begin(), s3.end());
                        td.s3 = im.upload(s3f);
                    }
                }
                for (int w = 0; w < 4; ++w) {
                    std::vector<half_t>& ws = tower_ws[w];
                    for (int kk = -1; kk <= n; ++kk) {                 // interval: E(kk+1) then P(kk-1)
                        if (kk + 1 < n) {                              // expand A fragments [k-step]: rows = my 32 channels, k = input channel
                            const int c = kk + 1;
                            for (int ks = 0; ks < 16; ++ks)
                                for (int l = 0; l < 64; ++l)
                                    for (int j = 0; j < 8; ++j) {
                                        const int ch = c * 128 + w * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + j;
                                        ws.push_back(half_t(ch < cop ? float(f1.w[size_t(ch) * C + k]) : 0.f));
                                    }
                        }
                        if (kk - 1 >= 0) {                             // project A fragments [k-step][row tile]: rows = my 64 couts, k = tower K position
                            const int c = kk - 1;
                            for (int ks = 0; ks < 8; ++ks)
                                for (int rt = 0; rt < 2; ++rt)
                                    for (int l = 0; l < 64; ++l)
                                        for (int j = 0; j < 8; ++j) {
                                            const int co = w * 64 + rt * 32 + (l & 31);
                                            const int ch = tower_k_channel(c * 128 + ks * 16 + (l >> 5) * 8 + j);
                                            ws.push_back(half_t(ch < cop ? float(f3.w[size_t(co) * cop + ch]) : 0.f));
                                        }
                        }
                    }
                    for (int c = 0; c < n; ++c) {
                        for (int lh = 0; lh < 2; ++lh)                  // BN1 biases [lane/32][accumulator element v]
                            for (int v = 0; v < 16; ++v) {
                                const int ch = c * 128 + w * 32 + (v % 4) + 8 * (v / 4) + 4 * lh;
                                if (int8_) {                        // int32 bit pattern of the BN1 bias in the accumulator's unit
                                    const int32_t start = ch < cop ? int32_t(std::nearbyint(f1.b[ch] / k1[ch])) : 0;
                                    float bits;
                                    std::memcpy(&bits, &start, 4);
                                    tower_bs[w].push_back(bits);
                                } else
                                tower_bs[w].push_back(ch < cop ? float(f1.b[ch] / s1[ch]) : 0.f);
                            }
                        // depthwise weights for K positions w*32 + lg*8 + pi*2 + {0,1}, entries = k*k taps then the BN2 bias:
                        //   5 x 5: [32 entries][lg][pair pi][2]   (entry-major: the four lane groups of a broadcast read sit in four bank slots)
                        //   3 x 3: [10 entries][lg][file variant][pair pi][2], variant 0 = file a (taps with dx = -1 zeroed), 1 = files b..g,
                        //          2 = file h (dx = +1 zeroed); zero padding up to the chunk's 2 KiB
                        auto dwv = [&](int lgk, int ent, int pi, int hh) {
                            const int ch = tower_k_channel(c * 128 + w * 32 + lgk * 8 + pi * 2 + hh);
                            double v = 0.0;
                            if (ch < cop && ent <= k * k) v = ent < k * k ? f2.w[size_t(ch) * k * k + ent] * (int8_ ? k1[ch] / kInt8Escale : s1[ch]) : f2.b[ch];
                            return v;
                        };
                        const size_t chunk_begin = tower_ps[w].size();
                        if (k == 3) {
                            for (int ent = 0; ent < 10; ++ent)
                                for (int lgk = 0; lgk < 4; ++lgk)
                                    for (int var = 0; var < 3; ++var)
                                        for (int pi = 0; pi < 4; ++pi)
                                            for (int hh = 0; hh < 2; ++hh) {
                                                const bool off_board = ent < 9 && ((var == 0 && ent % 3 == 0) || (var == 2 && ent % 3 == 2));
                                                tower_ps[w].push_back(half_t(off_board ? 0.f : float(dwv(lgk, ent, pi, hh))));
                                            }
                        } else {
                            for (int ent = 0; ent < 32; ++ent)
                                for (int lgk = 0; lgk < 4; ++lgk)
                                    for (int pi = 0; pi < 4; ++pi)
                                        for (int hh = 0; hh < 2; ++hh) tower_ps[w].push_back(half_t(float(dwv(lgk, ent, pi, hh))));
                        }
                        tower_ps[w].resize(chunk_begin + 1024, half_t(0.f));
                    }
                }
                if (!int8_) td.b3 = im.upload_d2f(f3.b, C);
                td.cop_pad = cop_pad;
                td.ks = k;
                if (tower_blocks.empty()) {
                    tower_gate = pending_gate;     // gate computed by the launches before this run (or none)
                    pending_gate = nullptr;
                }
                tower_blocks.push_back(td);
                macs += double(kSquares) * cop * (2.0 * C + k * k);
            }
        } else if (in_x3_tower) {
            const int cop_pad = round_up(cop, block_x3_chunk_channels());
            Folded f1 = fold_bn(nf, p + ".body.0", p + ".body.1");
            Folded f2 = fold_bn(nf, p + ".body.3", p + ".body.4");
            Folded f3 = fold_bn(nf, p + ".body.6", p + ".body.7");
            double w1_inv = 1.0, w3_inv = 1.0;
            const bool p8_images = p8_ && !split_block;
            SplitPack s1 = p8_images ? pack_dense_p8(f1, cop, C, 1, cop_pad, C, &w1_inv) : pack_dense_split(f1, cop, C, 1, cop_pad, C);
            SplitPack s3 = p8_images ? pack_dense_p8(f3, C, cop, 1, C, cop_pad, &w3_inv) : pack_dense_split(f3, C, cop, 1, C, cop_pad);
            xb.w1pk = im.upload(s1.hi);
            xb.w1pk_lo = im.upload(s1.lo);
            xb.w3pk = im.upload(s3.hi);
            xb.w3pk_lo = im.upload(s3.lo);
            xb.dwpk = im.upload(k == 5 ? pack_x3_depthwise_records5(f1, f2, cop, cop_pad) : pack_x3_depthwise_records(f1, f2, cop, cop_pad));
            xb.b3 = im.upload_d2f(f3.b, C);
            xb.w1_inv = float(w1_inv);                                     // float16p8: the accumulators run in the weights' scales
            xb.w3_inv = float(w3_inv);
            xb.w3_scale = float(1.0 / w3_inv);
            xb.cop_pad = cop_pad;
            x3_blocks.push_back(xb);
            macs += double(kSquares) * cop * (2.0 * C + k * k);
        } else if (block_fused(k)) {
            // fused bottleneck block: expand -> depthwise -> project -> +x in one launch (kernels.hip: block_kernel; x3.hip: block_x3_kernel)
            const int cop_pad = round_up(cop, x3_ ? block_x3_chunk_channels() : block_chunk_channels<T>());
            Folded f1 = fold_bn(nf, p + ".body.0", p + ".body.1");
            Folded f2 = fold_bn(nf, p + ".body.3", p + ".body.4");
            Folded f3 = fold_bn(nf, p + ".body.6", p + ".body.7");
            std::vector<float> w(size_t(k) * k * cop_pad, 0.f);
            for (int c = 0; c < cop; ++c) for (int t = 0; t < k * k; ++t) w[size_t(t) * cop_pad + c] = float(f2.w[size_t(c) * k * k + t]);
            Op op;
            op.kind = OpKind::Block;
            BlockArgs& ba = op.blk;
            ba.x = cur;
            ba.y = nxt;
            if (x3_) {
                SplitPack s1 = pack_dense_split(f1, cop, C, 1, cop_pad, C), s3 = pack_dense_split(f3, C, cop, 1, C, cop_pad);
                ba.w1pk = im.upload(s1.hi);
                ba.w1pk_lo = im.upload(s1.lo);
                ba.w3pk = im.upload(s3.hi);
                ba.w3pk_lo = im.upload(s3.lo);
            } else {
                ba.w1pk = im.upload(pack_dense<T>(f1, cop, C, 1, cop_pad, C));
                ba.w3pk = im.upload(pack_dense<T>(f3, C, cop, 1, C, cop_pad));
            }
            ba.b1 = im.upload_d2f(f1.b, cop_pad);
            ba.wdw = im.upload(w);
            ba.b2 = im.upload_d2f(f2.b, cop_pad);
            ba.b3 = im.upload_d2f(f3.b, C);
            ba.batch = B;
            ba.C = C;
            ba.cop_pad = cop_pad;
            ba.ks = k;
            if (k == 3) {   // per-channel record for the DPP depthwise kernel: 9 taps, BN1 bias, BN2 bias, pad
                if (x3_) {
                    ba.dwpk = im.upload(pack_x3_depthwise_records(f1, f2, cop, cop_pad));
                } else {
                    std::vector<float> rec(size_t(cop_pad) * 12, 0.f);
                    for (int c = 0; c < cop; ++c) {
                        for (int t = 0; t < 9; ++t) rec[size_t(c) * 12 + t] = float(f2.w[size_t(c) * 9 + t]);
                        rec[size_t(c) * 12 + 9] = float(f1.b[c]);
                        rec[size_t(c) * 12 + 10] = float(f2.b[c]);
                    }
                    ba.dwpk = im.upload(rec);
                }
            }
            ba.gate = pending_gate;
            pending_gate = nullptr;
            prod_op = int(im.ops.size());
            im.ops.push_back(op);
            macs += double(kSquares) * cop * (2.0 * C + k * k);
            std::swap(cur, nxt);
        } else {
            add_conv(p + ".body.0", p + ".body.1", cur, e, nullptr, C, C, cop, 1, true, nullptr);   // 1x1 expand + BN + ReLU
            {   // depthwise k x k + BN + ReLU
                Folded fd = fold_bn(nf, p + ".body.3", p + ".body.4");
                std::vector<float> w(size_t(k) * k * cop);
                for (int c = 0; c < cop; ++c) for (int t = 0; t < k * k; ++t) w[size_t(t) * cop + c] = float(fd.w[size_t(c) * k * k + t]);
                Op op;
                op.kind = OpKind::Depthwise;
                op.x = e;
                op.y = f;
                op.w0 = im.upload(w);
                op.b0 = im.upload_d2f(fd.b);
                op.C = cop;
                op.ks = k;
                im.ops.push_back(op);
                macs += double(kSquares) * cop * k * k;
            }
            add_conv(p + ".body.6", p + ".body.7", f, nxt, cur, cop, cop, C, 1, false, nullptr);    // 1x1 project + BN + residual
            std::swap(cur, nxt);
            prod_op = -1;                  // the residual stream now comes from a layer kernel: nobody emits its channel sums
        }
    }
    flush_tower();
    flush_x3_tower();
    // value heads with fewer than 8 channels (AlphaZeroResnet: 4) run as 8 with zero rows: ReLU(0) = 0 meets zero FC weights
    const bool head_ok = tower_ok && policy_map && cv >= 1 && cv <= 8 && cp <= 96 && (wdl || fc == 256);
    if (head_ok) {
        // policy + value head in one launch (head.hip; stream layouts in kernels.h: HeadArgs)
        if constexpr (kHalf) {
            Folded f1 = fold_bn(nf, "policy_head.body.0", "policy_head.body.1");
            Folded f2 = fold_bn(nf, "policy_head.body.3", "");
            Folded fv = fold_bn(nf, "value_head.body.0", "value_head.body.1");
            std::vector<half_t> s1, s2;
            std::vector<float> b1;
            const half_t hz = half_t(0.f);
            for (int wv = 0; wv < 8; ++wv) {
                for (int tap = 0; tap < 9; ++tap)
                    for (int ks = 0; ks < 16; ++ks)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int co = wv * 32 + (l & 31), ci = ks * 16 + (l >> 5) * 8 + j;
                                s1.push_back(half_t(float(f1.w[(size_t(co) * C + ci) * 9 + tap])));
                            }
                for (int ks = 0; ks < 16; ++ks)          // value head 1x1 conv (wave 0), rows 0..7
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int row = l & 31, ci = ks * 16 + (l >> 5) * 8 + j;
                            s1.push_back(wv == 0 && row < cv ? half_t(float(fv.w[size_t(row) * C + ci])) : hz);
                        }
                s1.insert(s1.end(), size_t(16) * 512, hz);
                for (int lh = 0; lh < 2; ++lh)
                    for (int v = 0; v < 16; ++v) b1.push_back(float(f1.b[wv * 32 + (v % 4) + 8 * (v / 4) + 4 * lh]));
                for (int i = 0; i < 18; ++i) {
                    const int u = wv * 18 + i, tap = u >> 4, ks = u & 15;
                    for (int rt = 0; rt < 3; ++rt)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int co = rt * 32 + (l & 31), kpos = ks * 16 + (l >> 5) * 8 + j;
                                const int ci = (kpos / 32) * 32 + tower_row_of_position(kpos % 32);
                                s2.push_back(co < cp ? half_t(float(f2.w[(size_t(co) * C + ci) * 9 + tap])) : hz);
                            }
                }
                s2.insert(s2.end(), size_t(9) * 512, hz);
            }
            Op op;
            op.kind = OpKind::Head;
            HeadArgs& h = op.hd;
            h.x = cur;
            h.logits = d_logits_;
            h.probs = d_probs_;
            h.value = d_value_;
            h.aux = d_aux_;
            h.s1 = im.upload(s1);
            h.b1 = im.upload(b1);
            h.s2 = im.upload(s2);
            h.s1_wave_frags = 9 * 16 + 16 + 16;
            h.s2_wave_frags = 18 * 3 + 9;
            h.vconv_bias = im.upload_d2f(fv.b, 8);
            h.cp = cp;
            h.batch = B;
            if (getenv("CRA_TOWER_TRACE")) h.trace = static_cast<unsigned long long*>(im.dalloc(64 * sizeof(unsigned long long)));
            const int nfl = kSquares * cv;
            if (wdl) {
                const TensorView &ww = nf.get("value_head.body_wdl.0.weight"), &wp = nf.get("value_head.body_plys.0.weight");
                std::vector<float> w4(size_t(4) * 512, 0.f);                      // [4][512], rows zero-padded beyond nfl
                for (int r = 0; r < 3; ++r) std::copy(ww.data + size_t(r) * nfl, ww.data + size_t(r + 1) * nfl, w4.begin() + size_t(r) * 512);
                std::copy(wp.data, wp.data + nfl, w4.begin() + size_t(3) * 512);
                const float* bw = nf.get("value_head.body_wdl.0.bias").data;
                h.fc1_w = im.upload(w4);
                h.wdl_b[0] = bw[0]; h.wdl_b[1] = bw[1]; h.wdl_b[2] = bw[2];
                h.wdl_b[3] = nf.get("value_head.body_plys.0.bias").data[0];
                h.wdlp = 1;
                macs += 4.0 * nfl;
            } else {
                const TensorView &w1 = nf.get("value_head.body_final.0.weight"), &w2 = nf.get("value_head.body_final.2.weight");
                // thread order (head.hip, phase 4): thread t of 512 owns outputs 2*(t/4), +1 over k in [128*(t%4), +128); its load i is the
                // uint4 at index i*512 + t = the (w[2j2][k], w[2j2+1][k]) pairs of k = 128*(t%4) + 4i .. 4i+3; k >= nfl: zeros
                std::vector<half_t> w1t(size_t(512) * fc, half_t(0.f));
                for (int i = 0; i < 32; ++i)
                    for (int t = 0; t < 512; ++t)
                        for (int j = 0; j < 4; ++j) {
                            const int k = 128 * (t & 3) + 4 * i + j, o = 2 * (t >> 2);
                            if (k >= nfl) continue;
                            w1t[((size_t(i) * 512 + t) * 4 + j) * 2 + 0] = half_t(w1.data[size_t(o) * nfl + k]);
                            w1t[((size_t(i) * 512 + t) * 4 + j) * 2 + 1] = half_t(w1.data[size_t(o + 1) * nfl + k]);
                        }
                const float* bb = nf.get("value_head.body_final.0.bias").data;
                h.fc1_w = im.upload(w1t);
                h.fc1_b = im.upload(std::vector<float>(bb, bb + fc));
                h.fc2_w = im.upload(std::vector<float>(w2.data, w2.data + fc));
                h.fc2_b = nf.get("value_head.body_final.2.bias").data[0];
                macs += double(nfl) * fc + fc;
            }
            macs += double(kSquares) * 9 * (double(C) * C + double(C) * cp) + double(kSquares) * C * cv;
            im.ops.push_back(op);
        }
    } else {
    // _PolicyHead (select_policy_from_plane), builder_util.py:206-243
    // Precision float16p8, policy map at 256 channels: both convs of the head in ONE launch (x3.hip: conv3x3_p8_chain_kernel); CRA_P8_NO_HEAD_CHAIN: development A/B
    // (a small batch: float16x3's convs, the first one's couts over four workgroups per board, the second beside the value head -- the
    // chain's 0.049 ms at batch 1 became 0.017 + 0.024, the latter shared with the value head: profiles/r06/f_*, y_*)
    const bool head_chain = p8_ && policy_map && C == 256 && round_up(cp, 16) <= 128 && getenv("CRA_P8_NO_HEAD_CHAIN") == nullptr &&
                            !(x3_split && getenv("CRA_SMALL_BATCH_HEAD_CHAIN") == nullptr);
    // Precision float16x3 has the same head as one launch since round 6 (x3.hip: conv3x3_x3_chain_kernel, the same bits as the two launches);
    // CRA_X3_NO_HEAD_CHAIN: development A/B.  Small-batch nets keep the two launches (the first conv's couts over four workgroups per board).
    const bool head_chain_x3 = x3_ && !p8_ && fused_ && policy_map && C == 256 && round_up(cp, 16) <= 128 && !x3_split &&
                               getenv("CRA_X3_NO_HEAD_CHAIN") == nullptr;
    if (head_chain_x3) {
        Folded f1 = fold_bn(nf, "policy_head.body.0", "policy_head.body.1");
        SplitPack s1 = pack_dense_split(f1, C, C, 3, C, C);
        add_conv("policy_head.body.3", "", cur, nullptr, nullptr, C, C, cp, 3, false, d_logits_, false);       // (its x: the tower's output)
        ConvArgs& c = im.ops.back().conv;
        c.pre_wpk = im.upload(s1.hi);
        c.pre_wpk_lo = im.upload(s1.lo);
        c.pre_bias = im.upload_d2f(f1.b, C);
        c.pre_acc_scale = 1.f;
        macs += double(kSquares) * C * C * 9;
    } else
    if (head_chain) {
        Folded f1 = fold_bn(nf, "policy_head.body.0", "policy_head.body.1");
        double inv1 = 1.0;
        SplitPack s1 = pack_dense_p8(f1, C, C, 3, C, C, &inv1);
        add_conv("policy_head.body.3", "", cur, nullptr, nullptr, C, C, cp, 3, false, d_logits_, true);      // (its x: the tower's output)
        ConvArgs& c = im.ops.back().conv;
        c.pre_wpk = im.upload(s1.hi);
        c.pre_wpk_lo = im.upload(s1.lo);
        c.pre_bias = im.upload_d2f(f1.b, C);
        c.pre_acc_scale = float(inv1);
        macs += double(kSquares) * C * C * 9;
    } else {
        // (a small batch: float16x3's convs in both modes, like its blocks -- the cross terms on e5m2 buy nothing where a launch is its latency)
        add_conv("policy_head.body.0", "policy_head.body.1", cur, nxt, nullptr, C, C, C, 3, true, nullptr, !x3_split);
        im.ops.back().conv.few_boards = x3_split ? dev_.small_conv_split : 0;
    }
    if (head_chain || head_chain_x3) {
    } else if (policy_map) {
        add_conv("policy_head.body.3", "", nxt, nullptr, nullptr, C, C, cp, 3, false, d_logits_, !x3_split);
    } else {
        // flat labels: conv3x3(C->P) + BN + ReLU written channel-major flat (x.view(-1, nb_flatten)), then Linear(P*64 -> n_labels)
        // as a GEMM over the BATCH (64 boards play the 64 "squares" of a workgroup tile), float logits row per board
        const int nfl = cp * kSquares, Bpad = round_up(B, 64), co_pad = round_up(cp, 16);
        T* pflat = static_cast<T*>(im.dalloc(size_t(Bpad) * nfl * sizeof(T)));
        HIP_CHECK(hipMemset(pflat, 0, size_t(Bpad) * nfl * sizeof(T)));
        {
            Folded fd = fold_bn(nf, "policy_head.body.3", "policy_head.body2.0");
            Op op;
            op.kind = OpKind::Conv;
            op.conv.x = nxt;
            set_conv_weights(op.conv, fd, cp, C, 3, co_pad, C);
            op.conv.bias = im.upload_d2f(fd.b, co_pad);
            op.conv.out = pflat;
            op.conv.batch = B;
            op.conv.cin = C;
            op.conv.cout_pad = co_pad;
            op.conv.cout_real = cp;
            op.conv.cout_ld = co_pad;
            op.conv.ks = 3;
            op.conv.relu = 1;
            op.conv.out_flat = 1;
            op.conv.flat_pitch = nfl;
            im.ops.push_back(op);
            macs += double(kSquares) * C * cp * 9;
        }
        {
            const TensorView& w = nf.get("policy_head.body3.0.weight");
            const float* bb = nf.get("policy_head.body3.0.bias").data;
            Folded fl;
            fl.w.assign(w.data, w.data + size_t(n_labels) * nfl);
            fl.b.assign(bb, bb + n_labels);
            const int nl_pad = round_up(n_labels, 16);
            Op op;
            op.kind = OpKind::Conv;
            op.conv.x = pflat;
            set_conv_weights(op.conv, fl, n_labels, nfl, 1, nl_pad, nfl);
            op.conv.bias = im.upload_d2f(fl.b, nl_pad);
            op.conv.out = d_logits_;
            op.conv.batch = Bpad / 64;
            op.conv.cin = nfl;
            op.conv.cout_pad = nl_pad;
            op.conv.cout_real = n_labels;
            op.conv.cout_ld = nl_pad;
            op.conv.ks = 1;
            op.conv.relu = 0;
            op.conv.out_rows_f32 = 1;
            op.conv.rows_valid = B;
            im.ops.push_back(op);
            macs += double(nfl) * n_labels;
        }
    }
    // Precision float16x3, policy map: the policy conv holds a board's whole logit vector in one workgroup and runs the softmax itself
    // (conv_gemm_x3_kernel; the launcher takes one workgroup per board up to 256 couts, the staging tiles hold 8192 logits)
    if (x3_ && !im.ops.empty() && im.ops.back().kind == OpKind::Conv && im.ops.back().conv.out_policy_f32 &&
        im.ops.back().conv.cout_pad <= 256 && im.ops.back().conv.cout_real * kSquares <= 8192 &&
        getenv("CRA_X3_NO_FUSED_SOFTMAX") == nullptr) {                    // (development: the softmax as its own launch)
        im.ops.back().fused_softmax = true;
    } else {
        Op op;
        op.kind = OpKind::Softmax;
        im.ops.push_back(op);
    }
    // CRA_X3_VALUE_HEAD=one / three: the float16x3 forward's value head as the one-launch f32 kernel or as the three launches below
    // (development: A/B and the lane determinism stress test, tests/test_lane_determinism_gpu.py)
    const char* x3_vh = getenv("CRA_X3_VALUE_HEAD");
    const bool x3_value_one_launch = x3_ && fused_ && (x3_vh ? x3_vh[0] == 'o' : kX3ValueHeadOneLaunch);
    if (fused_ && !x3_value_one_launch) {
        // _ValueHead (builder_util.py:246-326) as three MFMA/wave-level launches instead of one latency-bound VALU kernel (Precision
        // float16 / fp8 layer paths; float16x3 on request).  Precision float16x3 runs the one-launch f32 kernel below (0.022 ms against
        // 0.039): in round 3 it made two-lane searches irreproducible -- its FC1 ran on v_pk_fma_f32, which goes wrong beside the MFMA
        // waves of the other lane's policy conv on the same SIMD (profiles/NOTES.md round 5); FC1 is on v_fmac_f32 since.
        //   (1) conv1x1(C->cv)+BN+ReLU on the conv-GEMM kernel, written channel-major flat  (x.view(-1, nb_flatten))
        //   (2) FC(nfl->fc)+ReLU as a GEMM over the BATCH: 64 boards play the role of the 64 "squares" of one workgroup tile
        //   (3) FC(fc->1)+tanh, or the WDLP outputs, one wave per board
        const int nfl = kSquares * cv;
        const int Bpad = round_up(B, 64);
        T* vflat = static_cast<T*>(im.dalloc(size_t(Bpad) * nfl * sizeof(T)));
        HIP_CHECK(hipMemset(vflat, 0, size_t(Bpad) * nfl * sizeof(T)));
        {
            Folded fd = fold_bn(nf, "value_head.body.0", "value_head.body.1");
            const int co_pad = round_up(cv, 16);
            Op op;
            op.kind = OpKind::Conv;
            op.conv.x = cur;
            set_conv_weights(op.conv, fd, cv, C, 1, co_pad, C);
            op.conv.bias = im.upload_d2f(fd.b, co_pad);
            op.conv.out = vflat;
            op.conv.batch = B;
            op.conv.cin = C;
            op.conv.cout_pad = co_pad;
            op.conv.cout_real = cv;
            op.conv.cout_ld = co_pad;
            op.conv.ks = 1;
            op.conv.relu = 1;
            op.conv.out_flat = 1;
            op.conv.flat_pitch = nfl;
            im.ops.push_back(op);
            macs += double(kSquares) * C * cv;
        }
        Op fin;
        fin.kind = OpKind::ValueFinal;
        ValueFinalArgs& vf = fin.vf;
        vf.value = d_value_;
        vf.aux = d_aux_;
        vf.batch = B;
        if (wdl) {
            const TensorView &ww = nf.get("value_head.body_wdl.0.weight"), &wp = nf.get("value_head.body_plys.0.weight");
            std::vector<float> w4(size_t(4) * nfl);
            std::copy(ww.data, ww.data + 3 * nfl, w4.begin());
            std::copy(wp.data, wp.data + nfl, w4.begin() + 3 * nfl);
            const float* bw = nf.get("value_head.body_wdl.0.bias").data;
            vf.in = vflat;
            vf.n = nfl;
            vf.w = im.upload(w4);
            vf.b[0] = bw[0]; vf.b[1] = bw[1]; vf.b[2] = bw[2];
            vf.b[3] = nf.get("value_head.body_plys.0.bias").data[0];
            vf.wdlp = 1;
            macs += 4.0 * nfl;
        } else {
            if (nfl % 32 != 0) throw std::runtime_error("value head flatten size must be a multiple of 32");
            const TensorView &w1 = nf.get("value_head.body_final.0.weight"), &w2 = nf.get("value_head.body_final.2.weight");
            const float* b1 = nf.get("value_head.body_final.0.bias").data;
            Folded f1;
            f1.w.assign(w1.data, w1.data + size_t(fc) * nfl);
            f1.b.assign(b1, b1 + fc);
            const int fc_pad = round_up(fc, 16);
            T* vh = static_cast<T*>(im.dalloc(size_t(Bpad) * fc_pad * sizeof(T)));
            Op op;
            op.kind = OpKind::Conv;
            op.conv.x = vflat;
            set_conv_weights(op.conv, f1, fc, nfl, 1, fc_pad, nfl);
            op.conv.bias = im.upload_d2f(f1.b, fc_pad);
            op.conv.out = vh;
            op.conv.batch = Bpad / 64;        // 64 boards per workgroup tile
            op.conv.cin = nfl;
            op.conv.cout_pad = fc_pad;
            op.conv.cout_real = fc;
            op.conv.cout_ld = fc_pad;
            op.conv.ks = 1;
            op.conv.relu = 1;
            im.ops.push_back(op);
            vf.in = vh;
            vf.n = fc_pad;
            std::vector<float> w2p(fc_pad, 0.f);
            std::copy(w2.data, w2.data + fc, w2p.begin());
            vf.w = im.upload(w2p);
            vf.b[0] = nf.get("value_head.body_final.2.bias").data[0];
            vf.wdlp = 0;
            macs += double(nfl) * fc + fc;
        }
        im.ops.push_back(fin);
    } else
    {   // _ValueHead, builder_util.py:246-326
        Folded fd = fold_bn(nf, "value_head.body.0", "value_head.body.1");
        const int nfl = kSquares * cv;
        Op op;
        op.kind = OpKind::ValueHead;
        ValueHeadArgs& v = op.vh;
        v.x = cur;
        v.wconv = im.upload_d2f(fd.w);
        v.bconv = im.upload_d2f(fd.b);
        v.value = d_value_;
        v.aux = d_aux_;
        v.batch = B;
        v.C = C;
        v.cv = cv;
        v.fc = fc;
        if (wdl) {
            const TensorView &ww = nf.get("value_head.body_wdl.0.weight"), &wp = nf.get("value_head.body_plys.0.weight");
            v.wwdl = im.upload(std::vector<float>(ww.data, ww.data + 3 * nfl));
            const float* bw = nf.get("value_head.body_wdl.0.bias").data;
            v.bwdl = im.upload(std::vector<float>(bw, bw + 3));
            v.wplys = im.upload(std::vector<float>(wp.data, wp.data + nfl));
            v.bplys = nf.get("value_head.body_plys.0.bias").data[0];
            macs += 4.0 * nfl;
        } else {
            const TensorView &w1 = nf.get("value_head.body_final.0.weight"), &w2 = nf.get("value_head.body_final.2.weight");
            std::vector<float> w1t(size_t(nfl) * fc);
            for (int t = 0; t < fc; ++t) for (int i = 0; i < nfl; ++i) w1t[size_t(i) * fc + t] = w1.data[size_t(t) * nfl + i];
            v.w1t = im.upload(w1t);
            const float* b1 = nf.get("value_head.body_final.0.bias").data;
            v.b1 = im.upload(std::vector<float>(b1, b1 + fc));
            v.w2 = im.upload(std::vector<float>(w2.data, w2.data + fc));
            v.b2 = nf.get("value_head.body_final.2.bias").data[0];
            macs += double(nfl) * fc + fc;
        }
        macs += double(kSquares) * C * cv;
        if (getenv("CRA_VALUE_HEAD_DEBUG") != nullptr) {              // development: stage checksums of every launch (ValueHeadArgs::dbg)
            // [B][8 + 1024] checksums and FC1 sums, then (variant & 16, the PROBE instantiation) [B][16 + 3 * 1024] words
            const size_t dbg_bytes = size_t(B) * ((8 + 1024) + (16 + 3 * 1024)) * sizeof(float);
            v.dbg = static_cast<float*>(im.dalloc(dbg_bytes));
            HIP_CHECK(hipMemset(v.dbg, 0, dbg_bytes));
            value_head_dbg_ = v.dbg;
        }
        v.lds_pad = -1;                                                // no LDS fence (kernels.hip: round 5's root cause)
        if (const char* pad = getenv("CRA_VALUE_HEAD_LDS_PAD")) v.lds_pad = atoi(pad);
        if (const char* var = getenv("CRA_VALUE_HEAD_VARIANT")) v.variant = atoi(var);
        prepare_value_head<T>(op.vh);
        im.ops.push_back(op);
    }
    // a small batch: the policy conv that ends in the softmax and the value head side by side in one launch (x3.hip: heads_small_kernel);
    // CRA_SMALL_BATCH_HEADS_APART: development A/B
    if (x3_split && x3_ && im.ops.size() >= 2 && im.ops.back().kind == OpKind::ValueHead && im.ops[im.ops.size() - 2].kind == OpKind::Conv &&
        im.ops[im.ops.size() - 2].fused_softmax && heads_small_fits(im.ops[im.ops.size() - 2].conv, im.ops.back().vh) &&
        getenv("CRA_SMALL_BATCH_HEADS_APART") == nullptr) {
        Op vh = im.ops.back();
        im.ops.pop_back();
        Op& op = im.ops.back();
        op.kind = OpKind::HeadsSmall;
        op.vh = vh.vh;
    }
    }   // !head_ok
    init_block_kernel_attributes<T>();
    init_x3_kernel_attributes();
    init_tower_kernel_attributes();
    init_restower_kernel_attributes();
    init_head_kernel_attributes();
    // stem -> tower -> head with nothing in between and nothing handed to other launches: one launch, the board tile stays in LDS
    if (one_launch_ && im.ops.size() == 3 && im.ops[0].kind == OpKind::Stem && im.ops[1].kind == OpKind::Tower && im.ops[2].kind == OpKind::Head &&
        im.ops[1].tw.gate_in == nullptr && im.ops[1].tw.pool_out == nullptr) {
        Op op;
        op.kind = OpKind::Forward;
        op.st = im.ops[0].st;
        op.tw = im.ops[1].tw;
        op.hd = im.ops[2].hd;
        im.ops.assign(1, op);
        init_forward_kernel_attributes();
    }
    design_.flops_per_position = 2.0 * macs;
    launches_ = int(im.ops.size());
}

template <typename T> void RiseNet::launch_op(int i, hipStream_t s, const IoOverride* io) {
    Impl& im = *impl_;
    const int B = dyn_n_ > 0 ? dyn_n_ : design_.batch;       // (a forward of fewer boards than the net was made for: small_net())
    const Op& op = im.ops[i];
    auto boards = [&](ConvArgs c) {                            // a board-batched conv of such a forward
        if (dyn_n_ > 0 && c.batch == design_.batch && !c.out_rows_f32) c.batch = dyn_n_;
        return c;
    };
    // io: the caller's pinned host buffers stand in for the device-side input / output tensors of this forward (zero-copy predict)
    const float* planes = io ? io->planes : d_planes_;
    float* value = io ? io->value : d_value_;
    float* probs = io ? io->probs : d_probs_;
    float* aux = (io && d_aux_) ? (io->aux ? io->aux : d_aux_) : d_aux_;
    switch (op.kind) {
        case OpKind::PlanesToAct:
            launch_planes_to_act<T>(op.x == d_planes_ ? planes : static_cast<const float*>(op.x), static_cast<T*>(op.y), B, op.C, im.cin_pad, s);
            break;
        case OpKind::Conv:
            if (x3_ && dev_.conv_dev >= 0) {                                // development: bisecting switches of conv_gemm_x3_kernel
                ConvArgs c = boards(op.conv);
                c.dev = dev_.conv_dev;
                if (op.from_planes) c.planes = planes;
                if (op.fused_softmax) { c.softmax_out = probs; if (!keep_logits_) c.out = nullptr; }
                launch_conv_gemm_x3(c, s);
            } else if (x3_ && op.from_planes) {
                ConvArgs c = boards(op.conv);
                c.planes = planes;
                launch_conv_gemm_x3(c, s);
            } else if (x3_ && op.fused_softmax) {
                ConvArgs c = boards(op.conv);
                c.softmax_out = probs;
                if (!keep_logits_) c.out = nullptr;          // (the logits stay in LDS unless a test / analysis asked for them)
                launch_conv_gemm_x3(c, s);
            } else if (x3_) launch_conv_gemm_x3(boards(op.conv), s);
            else launch_conv_gemm<T>(op.conv, s);
            break;
        case OpKind::Depthwise:
            launch_depthwise<T>(static_cast<const T*>(op.x), static_cast<T*>(op.y), op.w0, op.b0, B, op.C, op.ks, s);
            break;
        case OpKind::SE: launch_se<T>(static_cast<T*>(op.y), op.se_kind, op.w0, op.w1, op.b0, B, op.C, s, static_cast<const T*>(op.x)); break;
        case OpKind::ValueHead: {
            ValueHeadArgs v = op.vh;
            v.value = value;
            v.aux = aux;
            if (dyn_n_ > 0) v.batch = dyn_n_;
            launch_value_head<T>(v, s);
            break;
        }
        case OpKind::HeadsSmall: {
            HeadsSmallArgs h;
            h.conv = boards(op.conv);
            h.conv.softmax_out = probs;
            if (!keep_logits_) h.conv.out = nullptr;
            h.vh = op.vh;
            h.vh.value = value;
            h.vh.aux = aux;
            h.vh.batch = h.conv.batch;
            launch_heads_small(h, s);
            break;
        }
        case OpKind::Softmax: launch_softmax(d_logits_, probs, B, design_.nb_policy, s); break;
        case OpKind::Block:
            if (x3_) launch_block_x3(op.blk, s);
            else launch_block<T>(op.blk, s);
            break;
        case OpKind::ValueFinal: {
            ValueFinalArgs v = op.vf;
            v.value = value;
            v.aux = aux;
            launch_value_final<T>(v, s);
            break;
        }
        case OpKind::Tower: launch_tower(op.tw, s); break;
        case OpKind::Head: {
            HeadArgs h = op.hd;
            h.value = value;
            h.probs = probs;
            h.aux = aux;
            h.logits = keep_logits_ ? d_logits_ : nullptr;
            launch_head(h, s);
            break;
        }
        case OpKind::ResTower: launch_restower(op.rt, s); break;
        case OpKind::TowerX3:
            if (dyn_n_ > 0) {
                X3TowerArgs t = op.tx;
                t.batch = dyn_n_;
                launch_tower_x3(t, s);
            } else launch_tower_x3(op.tx, s);
            break;
        case OpKind::BlockX3Split:
            if (dyn_n_ > 0) {                                    // the workgroups per board follow the boards of THIS forward; a launch reads
                X3SplitArgs a = op.xs;                           // as many images per board as the launch before it wrote
                a.batch = dyn_n_;
                a.G = std::max(1, std::min(std::min(10, cu_count_ / dyn_n_), a.blk.cop_pad / block_x3_chunk_channels()));
                if (dev_.x3_split_max_g > 0) a.G = std::min(a.G, dev_.x3_split_max_g);
                a.gin = (i == 0 || im.ops[i - 1].kind != OpKind::BlockX3Split) ? 1 : dyn_prev_g_;      // (a run's first block reads the float stream)
                dyn_prev_g_ = a.G;
                launch_block_x3_split(a, s);
            } else launch_block_x3_split(op.xs, s);
            break;
        case OpKind::X3SplitFinish: launch_x3_split_finish(op.xs.x_parts, dyn_n_ > 0 ? dyn_prev_g_ : op.xs.gin, op.xs_y, B, s); break;
        case OpKind::Stem: {
            StemArgs st = op.st;
            st.planes = planes;
            launch_stem(st, s);
            break;
        }
        case OpKind::Forward: {
            StemArgs st = op.st;
            HeadArgs h = op.hd;
            st.planes = planes;
            h.value = value;
            h.probs = probs;
            h.aux = aux;
            h.logits = keep_logits_ ? d_logits_ : nullptr;
            launch_forward(st, op.tw, h, s);
            break;
        }
        case OpKind::SEGate:
            launch_se_gate(static_cast<const float*>(op.x), static_cast<float*>(op.y), op.se_kind, op.w0, op.w1, op.b0, B, op.C, s);
            break;
    }
}

template <typename T> void RiseNet::enqueue(hipStream_t s, const IoOverride* io) {
    // (Round 6 tried the value head of a small batch on a side stream beside the policy head -- two branches of the captured graph: the
    // forward got SLOWER, 0.354 against 0.335 ms at batch 1, the cross-queue joins cost more than the 24 us they hide: profiles/r06/e_*.)
    dyn_prev_g_ = 1;
    for (int i = 0; i < int(impl_->ops.size()); ++i) launch_op<T>(i, s, io);
    HIP_CHECK(hipGetLastError());
}

const char* RiseNet::op_name(int i) const {
    const Op& op = impl_->ops.at(i);
    switch (op.kind) {
        case OpKind::PlanesToAct: return "planes_to_act";
        case OpKind::Conv: return x3_ ? (op.conv.ks == 1 ? "conv_gemm_x3_1x1" : "conv_gemm_x3_3x3") : (op.conv.ks == 1 ? "conv_gemm_1x1" : "conv_gemm_3x3");
        case OpKind::Depthwise: return "depthwise";
        case OpKind::SE: return "se";
        case OpKind::ValueHead: return "value_head";
        case OpKind::Softmax: return "softmax";
        case OpKind::Block: return x3_ ? "block_x3" : "fused_block";
        case OpKind::ValueFinal: return "value_final";
        case OpKind::SEGate: return "se_gate";
        case OpKind::Tower: return "tower";
        case OpKind::Head: return "head";
        case OpKind::ResTower: return "restower";
        case OpKind::Stem: return "stem";
        case OpKind::Forward: return "forward";
        case OpKind::TowerX3: return op.tx.p8 ? "tower_p8" : "tower_x3";
        case OpKind::BlockX3Split: return "block_x3_split";
        case OpKind::X3SplitFinish: return "x3_split_finish";
        case OpKind::HeadsSmall: return "heads_small";
    }
    return "?";
}

void RiseNet::time_ops(int iters, float* ms) {
    HIP_CHECK(hipSetDevice(device_));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    const int n = int(impl_->ops.size());
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < n; ++i) {
            HIP_CHECK(hipEventRecord(e0, stream_));
            if (fp16_) launch_op<half_t>(i, stream_); else launch_op<float>(i, stream_);
            HIP_CHECK(hipEventRecord(e1, stream_));
            HIP_CHECK(hipEventSynchronize(e1));
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            ms[i] += t;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (const Op& op : impl_->ops)
        if ((op.kind == OpKind::Head || op.kind == OpKind::Forward) && op.hd.trace) {
            unsigned long long h[16];
            HIP_CHECK(hipMemcpy(h, op.hd.trace, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "head trace (load, conv1, pack, conv2, atomics, softmax, value):");
            for (int i = 1; i < 8; ++i) fprintf(stderr, " %llu", h[i] - h[i - 1]);
            fprintf(stderr, "\n");
        }
    for (const Op& op : impl_->ops)
        if ((op.kind == OpKind::Tower || op.kind == OpKind::Forward) && op.tw.trace) {
            std::vector<unsigned long long> h(512);
            HIP_CHECK(hipMemcpy(h.data(), op.tw.trace, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int wv = 0; wv < 2; ++wv) {
                fprintf(stderr, "tower trace wave %d:", wv * 4);
                for (int i = 1; i < 256 && h[wv * 256 + i]; ++i) fprintf(stderr, " %llu", h[wv * 256 + i] - h[wv * 256 + i - 1]);
                fprintf(stderr, "\n");
            }
        }
}

// ---- development: the co-residency screen ----
namespace {
// 16-byte pieces of two buffers compared in place; every differing piece counts into *bad (one word per launch of the screened op)
__global__ __launch_bounds__(256) void screen_compare_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t n16, unsigned* bad) {
    unsigned diff = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256) {
        const uint4 x = a[i], y = b[i];
        diff += (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) ? 1u : 0u;
    }
    if (diff) atomicAdd(bad, diff);
}
void screen_compare(const char* a, const char* b, size_t bytes, unsigned* bad, hipStream_t s) {
    const size_t n16 = bytes / 16;                       // (allocations are multiples of 16 bytes or compared up to the last whole piece)
    if (!n16) return;
    const int blocks = int(std::min<size_t>(512, (n16 + 255) / 256));
    hipLaunchKernelGGL(screen_compare_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(a), reinterpret_cast<const uint4*>(b), n16, bad);
}
}  // namespace

int RiseNet::dev_screen_prepare() {
    HIP_CHECK(hipSetDevice(device_));
    Impl& im = *impl_;
    if (!im.screen.empty()) return int(im.screen.size());
    auto salloc = [&](size_t bytes) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
        im.screen_allocs.push_back(p);
        return static_cast<char*>(p);
    };
    im.screen_bad = reinterpret_cast<unsigned*>(salloc(sizeof(unsigned) * 65536));
    unsigned* flag = reinterpret_cast<unsigned*>(salloc(sizeof(unsigned) * im.mutables.size()));
    std::vector<char*> snap(im.mutables.size());
    for (size_t i = 0; i < im.mutables.size(); ++i) snap[i] = salloc(im.mutables[i].second);
    HIP_CHECK(hipStreamSynchronize(stream_));
    im.screen.resize(im.ops.size());
    std::vector<unsigned> hflag(im.mutables.size());
    for (int k = 0; k < int(im.ops.size()); ++k) {
        for (size_t i = 0; i < im.mutables.size(); ++i)
            HIP_CHECK(hipMemcpyAsync(snap[i], im.mutables[i].first, im.mutables[i].second, hipMemcpyDeviceToDevice, stream_));
        HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(unsigned) * im.mutables.size(), stream_));
        dev_launch_op(k, 1);
        for (size_t i = 0; i < im.mutables.size(); ++i) screen_compare(snap[i], im.mutables[i].first, im.mutables[i].second, flag + i, stream_);
        HIP_CHECK(hipMemcpyAsync(hflag.data(), flag, sizeof(unsigned) * hflag.size(), hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        ScreenOp& so = im.screen[k];
        for (size_t i = 0; i < im.mutables.size(); ++i) {
            if (!hflag[i]) continue;
            ScreenOp::Buf b{im.mutables[i].first, salloc(im.mutables[i].second), salloc(im.mutables[i].second), im.mutables[i].second};
            HIP_CHECK(hipMemcpyAsync(b.before, snap[i], b.bytes, hipMemcpyDeviceToDevice, stream_));
            HIP_CHECK(hipMemcpyAsync(b.after, b.live, b.bytes, hipMemcpyDeviceToDevice, stream_));
            so.writes.push_back(b);
        }
        // the op once more, on its own result
        HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(unsigned), stream_));
        dev_launch_op(k, 1);
        for (const ScreenOp::Buf& b : so.writes) screen_compare(b.after, b.live, b.bytes, flag, stream_);
        HIP_CHECK(hipMemcpyAsync(hflag.data(), flag, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        so.idempotent = hflag[0] == 0;
        if (!so.idempotent)                                   // leave the forward's state behind the op as it was
            for (const ScreenOp::Buf& b : so.writes) HIP_CHECK(hipMemcpyAsync(b.live, b.after, b.bytes, hipMemcpyDeviceToDevice, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
    }
    return int(im.screen.size());
}

long RiseNet::dev_screen_run(int op, int launches, long* words) {
    HIP_CHECK(hipSetDevice(device_));
    Impl& im = *impl_;
    if (im.screen.empty()) throw std::runtime_error("dev_screen_run: dev_screen_prepare first");
    if (op < 0 || op >= int(im.screen.size())) throw std::invalid_argument("op index out of range");
    launches = std::min(launches, 65536);
    const ScreenOp& so = im.screen[op];
    HIP_CHECK(hipMemsetAsync(im.screen_bad, 0, sizeof(unsigned) * launches, stream_));
    for (int l = 0; l < launches; ++l) {
        if (!so.idempotent)
            for (const ScreenOp::Buf& b : so.writes) HIP_CHECK(hipMemcpyAsync(b.live, b.before, b.bytes, hipMemcpyDeviceToDevice, stream_));
        dev_launch_op(op, 1);
        for (const ScreenOp::Buf& b : so.writes) screen_compare(b.after, b.live, b.bytes, im.screen_bad + l, stream_);
        if ((l & 63) == 63) HIP_CHECK(hipStreamSynchronize(stream_));      // (keeps the queue short; the neighbour's stream runs on)
    }
    std::vector<unsigned> bad(launches);
    HIP_CHECK(hipMemcpyAsync(bad.data(), im.screen_bad, sizeof(unsigned) * launches, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    long n = 0, w = 0;
    for (unsigned b : bad) { n += b != 0; w += b; }
    if (words) *words = w;
    return n;
}

std::string RiseNet::dev_screen_info(int op) const {
    const Impl& im = *impl_;
    if (op < 0 || op >= int(im.screen.size())) return "";
    size_t bytes = 0;
    for (const ScreenOp::Buf& b : im.screen[op].writes) bytes += b.bytes;
    return std::string(op_name(op)) + " writes " + std::to_string(im.screen[op].writes.size()) + " buffers / " + std::to_string(bytes) + " bytes" +
           (im.screen[op].idempotent ? "" : ", not idempotent (buffers restored before every launch)");
}

void RiseNet::dev_launch_op(int op, int iters) {
    HIP_CHECK(hipSetDevice(device_));
    if (op < 0 || op >= int(impl_->ops.size())) throw std::invalid_argument("op index out of range");
    for (int it = 0; it < iters; ++it) {
        if (fp16_) launch_op<half_t>(op, stream_); else launch_op<float>(op, stream_);
    }
    HIP_CHECK(hipGetLastError());
}

float RiseNet::time_forward(int iters) {
    HIP_CHECK(hipSetDevice(device_));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, stream_));
    for (int it = 0; it < iters; ++it) HIP_CHECK(hipGraphLaunch(graph_exec_, stream_));
    HIP_CHECK(hipEventRecord(e1, stream_));
    HIP_CHECK(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return t;
}

void RiseNet::keep_logits(bool on) {
    if (on == keep_logits_) return;
    keep_logits_ = on;
    if (launches_ > 1 && graph_exec_) {      // forwards of several launches replay a captured graph: capture again with the new head arguments
        HIP_CHECK(hipStreamSynchronize(stream_));
        (void)hipGraphExecDestroy(graph_exec_);
        if (graph_) (void)hipGraphDestroy(graph_);
        graph_exec_ = nullptr;
        graph_ = nullptr;
        capture();
    }
}

void RiseNet::capture() {
    // on a stream of its own: stream_ may be shared with a net that another thread is working with right now (NetStreams), and whatever
    // that thread launched between Begin and End would land in THIS graph
    hipStream_t cs = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipError_t err = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (err != hipSuccess) {
        (void)hipStreamDestroy(cs);
        HIP_CHECK(err);
    }
    try {
        forward_on(cs);
    } catch (...) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(cs, &g);
        if (g) (void)hipGraphDestroy(g);
        (void)hipStreamDestroy(cs);
        throw;
    }
    err = hipStreamEndCapture(cs, &graph_);
    (void)hipStreamDestroy(cs);
    HIP_CHECK(err);
    HIP_CHECK(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0));
}

void RiseNet::forward_on(hipStream_t s) {
    if (fp16_) enqueue<half_t>(s); else enqueue<float>(s);
}

// Forwards of DIFFERENT streams take turns when a forward fills the chip on its own (one workgroup per board, 160 KiB of LDS: one
// per CU).  Two evaluator lanes (or two NeuralNetAPIUsers) keep two batches in flight on two streams; when the streams sit on
// different hardware queues the dispatcher interleaves the workgroups of both forward kernels, both batches then finish together after
// 2 x 0.31 ms, the host collects for both lanes with nothing queued, and the chip idles for every collect: measured 0.42 ms per batch
// instead of 0.32 on the headline search leg (615k against 750k nodes/s), in one mode or the other for a whole process depending
// on which queues the runtime handed out.  Taking turns (submission order) keeps one batch executing and one queued.  Small batches
// are left alone: a batch of 8 occupies 8 CUs and SHOULD overlap with its neighbour.  The copy path of predict() gains too (its D2H copies
// now run beside the other user's forward: two users 559k -> 767k evals/s); zero-copy predict is exempt (see submit()).
namespace {
struct ForwardTurns {
    std::mutex mu;
    hipEvent_t ev[64];
    bool made = false, any = false;
    bool multi = false;                   // a second stream has shown up: from then on every forward records its event
    int last = 0;
    hipStream_t last_stream = nullptr;
};
ForwardTurns g_turns[64];     // per device
}  // namespace

static void turns_forget_stream(int device, hipStream_t s) {    // the stream is about to be destroyed (and has been drained)
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lk(g_turns[device].mu);
    if (g_turns[device].last_stream == s) {
        g_turns[device].last_stream = nullptr;
        g_turns[device].any = false;
    }
}

struct RiseNet::Turn {
    ForwardTurns* t = nullptr;
    hipStream_t s = nullptr;
    std::unique_lock<std::mutex> lk;       // a member: released also when the constructor throws
    Turn(RiseNet& n) {
        static const bool off = getenv("CRA_NO_FORWARD_TURNS") != nullptr;      // development: A/B
        static const bool always = getenv("CRA_FORCE_FORWARD_TURNS") != nullptr;
        if (off || n.device_ < 0 || n.device_ >= 64 || (!always && int(n.design_.batch) * 4 < n.cu_count_ * 3)) return;
        ForwardTurns* ft = &g_turns[n.device_];
        lk = std::unique_lock<std::mutex>(ft->mu);
        if (!ft->made) {
            HIP_CHECK(hipSetDevice(n.device_));
            for (hipEvent_t& e : ft->ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ft->made = true;
        }
        if (!ft->multi) {
            // one stream on this device so far (a device-resident loop over one net: the headline measurement): nothing to order, and
            // an event record per forward is not free (measured 3 us per 0.33 ms step)
            if (ft->last_stream == nullptr || ft->last_stream == n.stream_) {
                ft->last_stream = n.stream_;
                lk.unlock();
                return;
            }
            // a second stream: everything the first one has been given so far goes in front of this forward
            ft->multi = true;
            if (hipEventRecord(ft->ev[0], ft->last_stream) == hipSuccess) {
                ft->last = 0;
                ft->any = true;
            } else {
                (void)hipGetLastError();          // that stream is gone (its net was closed): nothing of it can be in flight
            }
        }
        if (ft->any && ft->last_stream != n.stream_) HIP_CHECK(hipStreamWaitEvent(n.stream_, ft->ev[ft->last], 0));
        t = ft;
        s = n.stream_;
    }
    ~Turn() {
        if (!t) return;
        const int next = (t->last + 1) & 63;
        if (hipEventRecord(t->ev[next], s) == hipSuccess) {
            t->last = next;
            t->last_stream = s;
            t->any = true;
        }
    }
};

// Device-resident replay.  A forward that is ONE kernel gains nothing from a graph (there is no launch sequence to save) and loses the
// graph launch's own cost between consecutive replays: it goes into the stream as a plain launch.  Everything else replays the graph.
// CRA_DEVICE_GRAPH=1 forces the graph (A/B timing).
void RiseNet::forward_async() {
    touch_stream();
    Turn turn(*this);
    if (launches_ == 1 && !dev_.device_graph) {
        forward_on(stream_);
        HIP_CHECK(hipGetLastError());
        return;
    }
    HIP_CHECK(hipGraphLaunch(graph_exec_, stream_));
}

// The forward between other work of the same stream (descriptor expansion before, gather / copies after).  A graph launch runs its
// nodes on the graph's own queue and is tied to the launching stream by cross-queue dependencies, which this runtime resolves from
// the host: measured, a lane's next kernel did not start until the host called into the runtime again (0.09-0.15 ms per batch lost
// whenever the host was busy collecting).  With the whole forward in one to five kernels there is nothing left for a graph to save,
// so these paths put the kernels straight into the stream: one queue, in-order, no host in the loop (float16p8's five launches: config 2
// searched at 373k nodes/s against 370k through the graph on an idle host, profiles/r04/ac_*).
void RiseNet::launch_forward_in_stream() {
    touch_stream();
    Turn turn(*this);
    if (dyn_n_ > 0 || (launches_ <= 5 && !dev_.lane_graph) || dev_.lane_no_graph) forward_on(stream_);      // (a forward of fewer boards: its own arguments)
    else HIP_CHECK(hipGraphLaunch(graph_exec_, stream_));
}

// every buffer of the call in pinned (device-visible) host memory?  Asked of the runtime on EVERY call (hipPointerGetAttributes: a
// microsecond or two per pointer against a forward of 100+ us): a remembered answer would outlive a hipHostFree / hipHostUnregister of
// the caller's buffers, and a later call with pageable memory at the same addresses would then be read and written by the kernels.
bool RiseNet::buffers_are_pinned(const float* in_planes, float* value, float* probs, float* aux) {
    if (dev_.predict_copy) return false;     // (CRA_PREDICT_COPY when the net was made: bench.py times the copy path on nets of its own)
    const void* set[4] = {in_planes, value, probs, (d_aux_ && aux) ? aux : nullptr};
    for (const void* p : set) {
        if (!p) continue;
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;     // the kernels read / write the buffers in 16-byte units
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();               // a pageable pointer is reported as an error by some runtimes: not ours to keep
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

void RiseNet::submit(const float* in_planes, float* value, float* probs, float* aux) {
    HIP_CHECK(hipSetDevice(device_));   // every predict selects its device, tensorrtapi.cpp:198
    const size_t B = design_.batch;
    // Zero-copy or staged?  With pinned buffers the kernels can read the planes and write value / probabilities across PCIe themselves: no
    // copy commands, the best form for ONE user (338k against 331k evals/s at batch 256, profiles/r05/p_*).  With a second user's forward
    // on the device the staged form wins by 6 - 12 % in every measurement (422k against 377k: the copies of one user run on the DMA
    // engines beside the other user's forward, while a zero-copy forward holds its CUs for the whole PCIe write): so the form is chosen
    // from what is in flight when the call arrives -- the reference's default is Threads = 2 (optionsuci.cpp), i.e. two users.
    // CRA_PREDICT_COPY / CRA_PREDICT_ZERO_COPY (when the net was made) force one form.
    const bool others_in_flight = device_ >= 0 && device_ < 64 && g_predicts_in_flight[device_].load(std::memory_order_relaxed) > (counted_in_flight_ ? 1 : 0);
    if (!counted_in_flight_ && device_ >= 0 && device_ < 64) {
        g_predicts_in_flight[device_].fetch_add(1, std::memory_order_relaxed);
        counted_in_flight_ = true;
    }
    // with hysteresis: a user that has met another one in flight stays on the staged form for its next 64 calls (two blocking users drift in
    // and out of phase: one of them would otherwise find the device "empty" at every other call and alternate between the forms)
    if (others_in_flight) staged_calls_left_ = 64;
    else if (staged_calls_left_ > 0) --staged_calls_left_;
    last_zero_copy_ = buffers_are_pinned(in_planes, value, probs, aux) && (dev_.predict_zero_copy || staged_calls_left_ == 0);
    if (last_zero_copy_) {
        IoOverride io;
        io.planes = in_planes;
        io.value = value;
        io.probs = probs;
        io.aux = (d_aux_ && aux) ? aux : nullptr;
        // no turn-taking here: these kernels write 5 MB of probabilities per batch across PCIe from inside the forward, and two users
        // in flight hide each other's write phase only when their kernels interleave (measured: 770k against 585k evals/s)
        if (fp16_) enqueue<half_t>(stream_, &io); else enqueue<float>(stream_, &io);
        return;
    }
    HIP_CHECK(hipMemcpyAsync(d_planes_, in_planes, B * design_.nb_input_channels * kSquares * sizeof(float), hipMemcpyHostToDevice, stream_));
    launch_forward_in_stream();
    HIP_CHECK(hipMemcpyAsync(value, d_value_, B * sizeof(float), hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipMemcpyAsync(probs, d_probs_, B * design_.nb_policy * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (d_aux_ && aux) HIP_CHECK(hipMemcpyAsync(aux, d_aux_, B * 4 * sizeof(float), hipMemcpyDeviceToHost, stream_));
}

// a float16x3 / float16p8 net made for more than kBoardSplitMaxBatch boards, asked for at most that many: the companion net's business
bool RiseNet::small_path_ok() const {
    return small_ != nullptr;
}
RiseNet& RiseNet::small_net() { return *small_; }

void RiseNet::submit_boards(const void* descs_host, int n_valid, int layout, float* value, float* probs, float* aux) {
    HIP_CHECK(hipSetDevice(device_));
    const size_t B = design_.batch;
    if (n_valid < 0 || size_t(n_valid) > B) throw std::invalid_argument("n_valid out of range");
    if (layout_channels(layout) != design_.nb_input_channels)
        throw std::invalid_argument("plane layout has " + std::to_string(layout_channels(layout)) + " channels, net expects " +
                                    std::to_string(design_.nb_input_channels));
    if (n_valid > 0 && n_valid <= kBoardSplitMaxBatch && small_path_ok()) {
        small_net().submit_boards(descs_host, n_valid, layout, value, probs, aux);      // (into this net's stream: wait() as ever)
        return;
    }
    // (this net as the companion of a larger one: a forward of n_valid boards, and only their results go back)
    const bool partial = n_valid > 0 && x3_ && board_split_ && design_.batch <= kBoardSplitMaxBatch && size_t(n_valid) < B && !dev_.no_small_path;
    const size_t rows = partial ? size_t(n_valid) : B;
    if (n_valid > 0) {
        HIP_CHECK(hipMemcpyAsync(d_desc_, descs_host, size_t(n_valid) * sizeof(BoardDesc), hipMemcpyHostToDevice, stream_));
        launch_planes_from_desc(static_cast<const BoardDesc*>(d_desc_), n_valid, layout, 1, d_planes_, stream_);
    }
    dyn_n_ = partial ? n_valid : 0;
    launch_forward_in_stream();
    dyn_n_ = 0;
    HIP_CHECK(hipMemcpyAsync(value, d_value_, rows * sizeof(float), hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipMemcpyAsync(probs, d_probs_, rows * design_.nb_policy * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (d_aux_ && aux) HIP_CHECK(hipMemcpyAsync(aux, d_aux_, rows * 4 * sizeof(float), hipMemcpyDeviceToHost, stream_));
}

void RiseNet::submit_boards_gathered(const void* descs_host, int n_valid, int layout, const uint16_t* idx, const uint32_t* cnt, uint32_t stride,
                                     float* value, float* gathered, float* aux) {
    HIP_CHECK(hipSetDevice(device_));
    const size_t B = design_.batch;
    if (n_valid < 0 || size_t(n_valid) > B) throw std::invalid_argument("n_valid out of range");
    if (stride == 0) throw std::invalid_argument("gather stride must be positive");
    if (layout_channels(layout) != design_.nb_input_channels)
        throw std::invalid_argument("plane layout has " + std::to_string(layout_channels(layout)) + " channels, net expects " +
                                    std::to_string(design_.nb_input_channels));
    if (n_valid > 0 && n_valid <= kBoardSplitMaxBatch && small_path_ok()) {          // few boards on a net made for many: the companion net
        small_net().submit_boards_gathered(descs_host, n_valid, layout, idx, cnt, stride, value, gathered, aux);
        return;
    }
    // No copy commands at all: the descriptors and the gather lists are read by the kernels straight from the caller's pinned
    // (device-visible, coherent) buffers, and the gather kernel writes values, gathered priors and aux straight into them.  A batch
    // moves ~50 KB in and ~170 KB out, so PCIe bandwidth is irrelevant; what a copy costs is the hand-over between the DMA engine
    // and the compute queue -- five copies around three kernel groups were 0.1-0.5 ms of latency per batch (host dependent), more
    // than the forward itself on a loaded host.  One queue, three launches back to back; the host polls the stream.
    Impl& im = *impl_;
    const char lane_mode_c[2] = {dev_.lane_launches, 0};             // CRA_LANE_LAUNCHES "1" / "2" / "3": force the shape of the lane step
    const char* lane_mode = dev_.lane_launches ? lane_mode_c : nullptr;
    if (im.ops.size() == 1 && im.ops[0].kind == OpKind::Forward && !(lane_mode && lane_mode[0] == '3')) {
        // The forward kernel's head writes the gathered priors, value and aux of a board straight into the caller's buffers: a search
        // reads nothing else (set_probabilities_for_moves, node.cpp:961-979), so neither the 20.7 KB probability vector nor the logits of
        // a board leave its CU, and there is no gather launch behind the forward.
        //  * SMALL batches, where the launches themselves are what a lane step costs (a batch of 8 occupies 8 CUs for 0.1 ms), run as
        //    ONE launch: the kernel's stem builds the planes of a board from its descriptor (profiles/r02/t_*: single-tree search at
        //    batch 8 +4-5 % with 1 to 8 collectors).
        //  * LARGE batches keep the plane builder as a launch of its own: inside the 0.31 ms kernel it cost more than the small launch
        //    does beside the other lane's forward (-4 % on the headline search leg).
        const bool one_launch = lane_mode ? lane_mode[0] == '1' : B <= 64;
        const Op& op = im.ops[0];
        StemArgs st = op.st;
        HeadArgs h = op.hd;
        if (one_launch) {
            st.descs = descs_host;
            st.layout = layout;
            st.n_valid = n_valid;
        } else if (n_valid > 0) {
            launch_planes_from_desc(static_cast<const BoardDesc*>(descs_host), n_valid, layout, 1, d_planes_, stream_);
        }
        h.value = value;
        h.probs = nullptr;
        h.logits = nullptr;
        h.aux = (d_aux_ && aux) ? aux : d_aux_;
        h.g_idx = idx;
        h.g_cnt = cnt;
        h.g_out = gathered;
        h.g_stride = int(stride);
        h.g_n_valid = n_valid;
        touch_stream();
        Turn turn(*this);                         // forwards that fill the chip take turns (small batches: a no-op)
        launch_forward(st, op.tw, h, stream_);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (n_valid > 0) launch_planes_from_desc(static_cast<const BoardDesc*>(descs_host), n_valid, layout, 1, d_planes_, stream_);
    // (this net as the companion of a larger one, or any small-batch net with fewer valid boards than its batch: a forward of n_valid boards)
    dyn_n_ = (n_valid > 0 && x3_ && board_split_ && design_.batch <= kBoardSplitMaxBatch && size_t(n_valid) < B && !dev_.no_small_path) ? n_valid : 0;
    launch_forward_in_stream();
    dyn_n_ = 0;
    if (dev_.lane_sync) HIP_CHECK(hipStreamSynchronize(stream_));     // development: bisecting the lane step's ordering
    launch_gather_probs(d_probs_, design_.nb_policy, idx, cnt, int(stride), n_valid, gathered, d_value_, value, int(B),
                        (d_aux_ && aux) ? d_aux_ : nullptr, aux, stream_);
}

void RiseNet::wait() {
    struct Done {                                                       // the predict is over however this call ends
        RiseNet& n;
        ~Done() {
            if (n.counted_in_flight_) {
                g_predicts_in_flight[n.device_].fetch_sub(1, std::memory_order_relaxed);
                n.counted_in_flight_ = false;
            }
        }
    } done{*this};
    // CRA_WAIT_POLL=1 polls hipStreamQuery instead (development: on the hosts measured so far the runtime's own wait was not the
    // source of the per-batch latency; both give the same pipeline rate)
    static const bool poll = getenv("CRA_WAIT_POLL") != nullptr;
    if (poll) {
        for (;;) {
            const hipError_t e = hipStreamQuery(stream_);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) HIP_CHECK(e);
            __builtin_ia32_pause();
        }
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void RiseNet::predict(const float* in_planes, float* value, float* probs, float* aux) {
    submit(in_planes, value, probs, aux);
    wait();
}

void* RiseNet::enable_block_dump(int* n_tiles) {
    HIP_CHECK(hipSetDevice(device_));
    Op* tower = nullptr;
    for (Op& op : impl_->ops)
        if (op.kind == OpKind::Tower || op.kind == OpKind::Forward) {
            if (tower) throw std::runtime_error("block dump: more than one tower launch in this net");
            tower = &op;
        }
    if (!tower) throw std::runtime_error("block dump: this net / precision does not run the one-launch bottleneck tower");
    const int tiles = tower->tw.nblocks + 1;
    if (!tower->tw.block_dump) {
        HIP_CHECK(hipStreamSynchronize(stream_));
        tower->tw.block_dump = impl_->dalloc(size_t(tiles) * design_.batch * kSquares * 256 * sizeof(half_t));
        if (launches_ > 1) {                 // forwards of several launches replay a captured graph: capture again with the pointer set
            if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
            if (graph_) (void)hipGraphDestroy(graph_);
            graph_exec_ = nullptr;
            graph_ = nullptr;
            capture();
        }
    }
    if (n_tiles) *n_tiles = tiles;
    return tower->tw.block_dump;
}

std::string int8_calibration_path(const std::string& model_file_path) { return model_file_path + ".int8calib"; }

std::vector<std::pair<float, float>> read_int8_calibration(const std::string& model_file_path) {
    std::vector<std::pair<float, float>> out;
    std::ifstream f(int8_calibration_path(model_file_path));
    if (!f) return out;
    std::string magic, word;
    int version = 0, boards = 0, blocks = 0;
    f >> magic >> version >> word >> boards >> word >> blocks;
    if (magic != "crazyara-int8-calibration" || version != 1 || blocks <= 0 || blocks > 4096)
        throw std::runtime_error("malformed INT8 calibration file " + int8_calibration_path(model_file_path));
    for (int i = 0; i < blocks; ++i) {
        float a = 0.f, b = 0.f;
        if (!(f >> a >> b) || !(a >= 0.f) || !(b >= 0.f)) throw std::runtime_error("malformed INT8 calibration file " + int8_calibration_path(model_file_path));
        out.emplace_back(a, b);
    }
    return out;
}

std::vector<std::pair<float, float>> RiseNet::calibration_maxima(const float* planes_host, int n_boards) {
    if (fused_ || !fp16_ || fp8_tower_) throw std::logic_error("calibration_maxima: a net made with Precision float16-unfused");
    if (!planes_host || n_boards <= 0) throw std::invalid_argument("calibration needs at least one board");
    HIP_CHECK(hipSetDevice(device_));
    Impl& im = *impl_;
    const int B = design_.batch;
    const size_t per_board = size_t(design_.nb_input_channels) * kSquares;
    std::vector<std::pair<float, float>> out;
    std::vector<half_t> host;
    auto absmax = [&](const void* dev, size_t count) {
        host.resize(count);
        HIP_CHECK(hipMemcpy(host.data(), dev, count * sizeof(half_t), hipMemcpyDeviceToHost));
        float m = 0.f;
        for (size_t i = 0; i < count; ++i) m = std::max(m, std::fabs(float(host[i])));
        return m;
    };
    std::vector<float> chunk(size_t(B) * per_board);
    for (int b0 = 0; b0 < n_boards; b0 += B) {
        for (int j = 0; j < B; ++j)                          // the last chunk repeats boards: a maximum does not mind
            std::memcpy(chunk.data() + size_t(j) * per_board, planes_host + size_t((b0 + j) % n_boards) * per_board, per_board * sizeof(float));
        HIP_CHECK(hipMemcpy(d_planes_, chunk.data(), chunk.size() * sizeof(float), hipMemcpyHostToDevice));
        size_t blk = 0;
        for (int k = 0; k < int(im.ops.size()); ++k) {
            launch_op<half_t>(k, stream_);
            HIP_CHECK(hipStreamSynchronize(stream_));
            const Op& op = im.ops[k];
            if (op.kind != OpKind::Depthwise) continue;
            if (k == 0 || im.ops[k - 1].kind != OpKind::Conv || im.ops[k - 1].conv.out != op.x)
                throw std::logic_error("calibration_maxima: a depthwise op without its expand conv in front");
            const Op& ex = im.ops[k - 1];                     // the expand conv has run: its input (the gated stream) is untouched
            const float mx = absmax(ex.conv.x, size_t(B) * kSquares * size_t(ex.conv.cin));
            const float mt = absmax(op.y, size_t(B) * kSquares * size_t(op.C));
            if (blk == out.size()) out.emplace_back(0.f, 0.f);
            out[blk].first = std::max(out[blk].first, mx);
            out[blk].second = std::max(out[blk].second, mt);
            ++blk;
        }
    }
    return out;
}

std::string calibrate_int8(const std::string& model_path, int device_id, const float* planes_host, int n_boards) {
    if (planes_host && n_boards <= 0) throw std::invalid_argument("calibration needs at least one board");
    RiseNet net(model_path, device_id, planes_host ? std::min(n_boards, 64) : 64, "float16-unfused");
    std::vector<float> own;
    if (!planes_host) {
        own = default_calibration_planes(net.design().nb_input_channels, net.design().version, &n_boards);
        planes_host = own.data();
    }
    const std::vector<std::pair<float, float>> mx = net.calibration_maxima(planes_host, n_boards);
    if (mx.empty()) throw std::runtime_error("Precision int8 runs on the one-launch bottleneck tower only: this model has no bottleneck blocks");
    const std::string path = int8_calibration_path(net.model_file_path());
    std::ofstream f(path);
    if (!f) throw std::runtime_error("cannot write " + path);
    f << "crazyara-int8-calibration 1\nboards " << n_boards << "\nblocks " << mx.size() << "\n";
    f.precision(9);
    for (const auto& p : mx) f << p.first << " " << p.second << "\n";
    if (!f) throw std::runtime_error("cannot write " + path);
    return path;
}

uint8_t float_to_e4m3(float v) { return to_e4m3(double(v)); }
uint8_t float_to_e5m2(float v) { return to_e5m2(v); }

}  // namespace cra
