// ONNX -> NetFile importer, see onnx_import.h.
//
// Wire-format field numbers used (onnx.proto3):
//   ModelProto      producer_name=2 graph=7 opset_import=8 {domain=1 version=2}
//   GraphProto      node=1 initializer=5 input=11 output=12
//   NodeProto       input=1 output=2 name=3 op_type=4 attribute=5
//   AttributeProto  name=1 f=2 i=3 s=4 t=5 floats=7 ints=8
//   TensorProto     dims=1 data_type=2 float_data=4 int32_data=5 int64_data=7 name=8 raw_data=9 double_data=10 external: 13/14
//   ValueInfoProto  name=1
#include "onnx_import.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>

namespace cra {
namespace {

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error("ONNX import: " + msg); }

// ---------------------------------------------------------------------------------------------------------------------------
// protobuf wire format
// ---------------------------------------------------------------------------------------------------------------------------
struct Slice {
    const uint8_t* p = nullptr;
    size_t n = 0;
    std::string str() const { return std::string(reinterpret_cast<const char*>(p), n); }
};

struct Field {
    int num = 0, wt = 0;
    uint64_t v = 0;      // varint / fixed payload
    Slice s;             // length-delimited payload
};

class Reader {
public:
    explicit Reader(Slice s) : p_(s.p), end_(s.p + s.n) {}
    bool next(Field& f) {
        if (p_ >= end_) return false;
        const uint64_t key = varint();
        f.num = int(key >> 3);
        f.wt = int(key & 7);
        switch (f.wt) {
            case 0: f.v = varint(); break;
            case 1: need(8); std::memcpy(&f.v, p_, 8); p_ += 8; break;
            case 5: { need(4); uint32_t t; std::memcpy(&t, p_, 4); f.v = t; p_ += 4; break; }
            case 2: {
                const uint64_t n = varint();
                need(n);
                f.s.p = p_;
                f.s.n = size_t(n);
                p_ += n;
                break;
            }
            default: fail("unsupported protobuf wire type " + std::to_string(f.wt));
        }
        return true;
    }
    uint64_t varint() {
        uint64_t r = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            need(1);
            const uint8_t b = *p_++;
            r |= uint64_t(b & 0x7F) << shift;
            if (!(b & 0x80)) return r;
        }
        fail("varint too long");
    }
    bool done() const { return p_ >= end_; }

private:
    void need(uint64_t n) const { if (uint64_t(end_ - p_) < n) fail("truncated file"); }
    const uint8_t *p_, *end_;
};

float f32_of(uint64_t bits) { uint32_t b = uint32_t(bits); float f; std::memcpy(&f, &b, 4); return f; }

float half_to_float(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000) << 16, exp = (h >> 10) & 31, man = h & 1023;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else {
            int e = -1;
            uint32_t m = man;
            do { ++e; m <<= 1; } while (!(m & 1024));
            out = sign | uint32_t(127 - 15 - e) << 23 | (m & 1023) << 13;
        }
    } else if (exp == 31) out = sign | 0x7F800000u | man << 13;
    else out = sign | (exp + 112) << 23 | man << 13;
    float f;
    std::memcpy(&f, &out, 4);
    return f;
}

struct Tensor {
    std::vector<int64_t> dims;
    std::vector<float> f;      // floating tensors
    std::vector<int64_t> i;    // integer tensors (shapes, indices)
    int64_t numel() const { int64_t n = 1; for (auto d : dims) n *= d; return n; }
};

void packed_ints(const Field& f, std::vector<int64_t>& out) {
    if (f.wt == 0) { out.push_back(int64_t(f.v)); return; }
    Reader r(f.s);
    while (!r.done()) out.push_back(int64_t(r.varint()));
}

void packed_floats(const Field& f, std::vector<float>& out) {
    if (f.wt == 5) { out.push_back(f32_of(f.v)); return; }
    if (f.s.n % 4) fail("malformed packed float field");
    const size_t n = f.s.n / 4, at = out.size();
    out.resize(at + n);
    std::memcpy(out.data() + at, f.s.p, n * 4);
}

Tensor parse_tensor(Slice s, std::string* name_out) {
    Tensor t;
    int dtype = 0;
    Slice raw;
    bool has_raw = false;
    std::vector<float> floats;
    std::vector<int64_t> ints;
    std::vector<double> doubles;
    std::string name;
    Reader r(s);
    Field f;
    while (r.next(f)) {
        switch (f.num) {
            case 1: packed_ints(f, t.dims); break;
            case 2: dtype = int(f.v); break;
            case 4: packed_floats(f, floats); break;
            case 5: case 7: packed_ints(f, ints); break;
            case 8: name = f.s.str(); break;
            case 9: raw = f.s; has_raw = true; break;
            case 10:
                if (f.wt == 1) { double d; std::memcpy(&d, &f.v, 8); doubles.push_back(d); }
                else { const size_t n = f.s.n / 8, at = doubles.size(); doubles.resize(at + n); std::memcpy(doubles.data() + at, f.s.p, n * 8); }
                break;
            case 13: case 14: fail("tensor " + name + " keeps its data in an external file; store the weights inside the .onnx");
            default: break;
        }
    }
    if (name_out) *name_out = name;
    {
        int64_t total = 1;                                        // bounded product: dims of a damaged file must not overflow numel()
        for (int64_t d : t.dims) {
            if (d < 0 || d > (int64_t(1) << 31)) fail("tensor " + name + " has an implausible dimension");
            total *= std::max<int64_t>(d, 1);
            if (total > (int64_t(1) << 31)) fail("tensor " + name + " is implausibly large");
        }
    }
    const int64_t n = t.numel();
    auto check = [&](size_t have) { if (int64_t(have) != n) fail("tensor " + name + " has " + std::to_string(have) + " elements for its dims"); };
    switch (dtype) {
        case 1:   // FLOAT
            if (has_raw) { check(raw.n / 4); t.f.resize(size_t(n)); std::memcpy(t.f.data(), raw.p, size_t(n) * 4); }
            else { check(floats.size()); t.f = std::move(floats); }
            break;
        case 10:  // FLOAT16 (raw, or bit patterns in int32_data)
            check(has_raw ? raw.n / 2 : ints.size());
            t.f.resize(size_t(n));
            if (has_raw) { for (int64_t k = 0; k < n; ++k) { uint16_t h; std::memcpy(&h, raw.p + 2 * k, 2); t.f[size_t(k)] = half_to_float(h); } }
            else { for (int64_t k = 0; k < n; ++k) t.f[size_t(k)] = half_to_float(uint16_t(ints[size_t(k)])); }
            break;
        case 11:  // DOUBLE
            check(has_raw ? raw.n / 8 : doubles.size());
            t.f.resize(size_t(n));
            if (has_raw) { for (int64_t k = 0; k < n; ++k) { double d; std::memcpy(&d, raw.p + 8 * k, 8); t.f[size_t(k)] = float(d); } }
            else { for (int64_t k = 0; k < n; ++k) t.f[size_t(k)] = float(doubles[size_t(k)]); }
            break;
        case 7:   // INT64
            if (has_raw) { check(raw.n / 8); t.i.resize(size_t(n)); std::memcpy(t.i.data(), raw.p, size_t(n) * 8); }
            else { check(ints.size()); t.i = std::move(ints); }
            break;
        case 6:   // INT32
            if (has_raw) { check(raw.n / 4); t.i.resize(size_t(n)); for (int64_t k = 0; k < n; ++k) { int32_t v; std::memcpy(&v, raw.p + 4 * k, 4); t.i[size_t(k)] = v; } }
            else { check(ints.size()); t.i = std::move(ints); }
            break;
        default:  // other types never carry weights of these nets
            break;
    }
    return t;
}

struct Attr {
    bool has_f = false, has_i = false;
    float f = 0;
    int64_t i = 0;
    std::vector<int64_t> ints;
    std::vector<float> floats;
    std::string s;
    Tensor t;
    bool has_t = false;
};

struct Node {
    std::string op, name;
    std::vector<std::string> in, out;
    std::map<std::string, Attr> attrs;
    int64_t int_attr(const std::string& k, int64_t dflt) const { auto it = attrs.find(k); return it != attrs.end() && it->second.has_i ? it->second.i : dflt; }
    float float_attr(const std::string& k, float dflt) const { auto it = attrs.find(k); return it != attrs.end() && it->second.has_f ? it->second.f : dflt; }
    const std::vector<int64_t>* ints_attr(const std::string& k) const { auto it = attrs.find(k); return it != attrs.end() ? &it->second.ints : nullptr; }
    std::string label() const { return op + " '" + (name.empty() ? (out.empty() ? std::string() : out[0]) : name) + "'"; }
};

Node parse_node(Slice s) {
    Node n;
    Reader r(s);
    Field f;
    while (r.next(f)) {
        switch (f.num) {
            case 1: n.in.push_back(f.s.str()); break;
            case 2: n.out.push_back(f.s.str()); break;
            case 3: n.name = f.s.str(); break;
            case 4: n.op = f.s.str(); break;
            case 5: {
                Attr a;
                std::string key;
                Reader ar(f.s);
                Field af;
                while (ar.next(af)) {
                    switch (af.num) {
                        case 1: key = af.s.str(); break;
                        case 2: a.f = f32_of(af.v); a.has_f = true; break;
                        case 3: a.i = int64_t(af.v); a.has_i = true; break;
                        case 4: a.s = af.s.str(); break;
                        case 5: a.t = parse_tensor(af.s, nullptr); a.has_t = true; break;
                        case 7: packed_floats(af, a.floats); break;
                        case 8: packed_ints(af, a.ints); break;
                        default: break;
                    }
                }
                n.attrs[key] = std::move(a);
                break;
            }
            default: break;
        }
    }
    return n;
}

std::string value_info_name(Slice s) {
    Reader r(s);
    Field f;
    while (r.next(f)) if (f.num == 1 && f.wt == 2) return f.s.str();
    return "";
}

// ---------------------------------------------------------------------------------------------------------------------------
// graph + matcher
// ---------------------------------------------------------------------------------------------------------------------------
struct ConvUnit {            // Conv [+ BatchNormalization] [+ Relu]
    const Tensor* w = nullptr;
    std::vector<double> scale, shift;   // per output channel, applied after the convolution
    bool relu = false;
    std::string out;
    int k = 0, group = 1, cout = 0, cin_g = 0;
    std::string label;
};

struct Linear {
    std::vector<float> w;    // [nout][nin]
    std::vector<float> b;    // [nout]
    bool has_bias = false;
    int nout = 0, nin = 0;
    std::string out;
};

constexpr float kLoaderBnVar = 0.99999f;   // running_var such that var + kBnEps (rise_net.hip) == 1

class Importer {
public:
    Importer(Slice file, const std::string& model_file_name, NetFile& nf) : nf_(nf), file_name_(model_file_name) { parse(file); }
    void run();

private:
    void parse(Slice file);
    void index();
    const std::string& resolve(const std::string& name) const {
        const std::string* cur = &name;
        for (auto it = alias_.find(*cur); it != alias_.end(); it = alias_.find(*cur)) cur = &it->second;
        return *cur;
    }
    const Tensor* init(const std::string& name) const {
        auto it = inits_.find(resolve(name));
        return it == inits_.end() ? nullptr : &it->second;
    }
    const std::vector<int>& cons(const std::string& t) const {
        static const std::vector<int> none;
        auto it = cons_.find(resolve(t));
        return it == cons_.end() ? none : it->second;
    }
    const Node* sole(const std::string& t, const char* op) const {
        const std::vector<int>& c = cons(t);
        return c.size() == 1 && nodes_[size_t(c[0])].op == op ? &nodes_[size_t(c[0])] : nullptr;
    }
    int pick(const std::vector<int>& c, const char* op) const {      // the only node of that type among c, -1 if none
        int found = -1;
        for (int i : c)
            if (nodes_[size_t(i)].op == op) {
                if (found >= 0) return -2;
                found = i;
            }
        return found;
    }
    bool is_output(const std::string& tensor, const char* name) const { return outputs_.count(name) && resolve(tensor) == resolve(name); }

    ConvUnit conv_unit(int idx) const;
    bool is_linear(const Node& n) const { return n.op == "Gemm" || n.op == "MatMul"; }
    Linear linear(const Node& n) const;
    std::string se_gate(const std::string& x, const std::string& prefix, std::string& se_type, bool plain_sigmoid = false);

    void put(const std::string& name, std::vector<int64_t> dims, std::vector<float> data) { out_.push_back({name, std::move(dims), std::move(data)}); }
    void put_conv_bn(const std::string& conv, const std::string& bn, const ConvUnit& u);
    void put_linear(const std::string& name, const Linear& l, bool with_bias);
    void finish();

    struct Out { std::string name; std::vector<int64_t> dims; std::vector<float> data; };
    NetFile& nf_;
    std::string file_name_, producer_;
    std::vector<Node> nodes_;
    std::map<std::string, Tensor> inits_;
    std::vector<std::string> inputs_;
    std::set<std::string> outputs_;
    std::map<std::string, std::string> alias_;
    std::map<std::string, std::vector<int>> cons_;
    std::vector<Out> out_;
};

void Importer::parse(Slice file) {
    Slice graph;
    Reader r(file);
    Field f;
    while (r.next(f)) {
        if (f.num == 7 && f.wt == 2) graph = f.s;
        else if (f.num == 2 && f.wt == 2) producer_ = f.s.str();
    }
    if (!graph.p) fail("no graph in the file (is this an ONNX model?)");
    Reader g(graph);
    std::vector<std::string> inputs;
    while (g.next(f)) {
        if (f.wt != 2) continue;
        switch (f.num) {
            case 1: nodes_.push_back(parse_node(f.s)); break;
            case 5: { std::string name; Tensor t = parse_tensor(f.s, &name); inits_[name] = std::move(t); break; }
            case 11: inputs.push_back(value_info_name(f.s)); break;
            case 12: outputs_.insert(value_info_name(f.s)); break;
            default: break;
        }
    }
    for (const std::string& i : inputs)
        if (!inits_.count(i)) inputs_.push_back(i);       // IR < 4 lists the initializers among the inputs
    for (Node& n : nodes_)                                  // Constant nodes are initializers by another name
        if (n.op == "Constant" && !n.out.empty()) {
            auto it = n.attrs.find("value");
            if (it != n.attrs.end() && it->second.has_t) inits_[n.out[0]] = it->second.t;
        }
    index();
}

void Importer::index() {
    // view ops: the tensor a reshape-like node forwards is the one it reads; shape arithmetic (everything downstream of a Shape
    // node that only mixes shapes and constants) is plumbing for those views and takes no part in the match
    static const std::set<std::string> views = {"Reshape", "Flatten", "Expand", "Squeeze", "Unsqueeze", "Identity", "Dropout", "Cast"};
    static const std::map<std::string, size_t> min_inputs = {{"Conv", 2}, {"BatchNormalization", 5}, {"Gemm", 2}, {"MatMul", 2}, {"Add", 2},
                                                             {"Mul", 2}, {"Relu", 1}, {"HardSigmoid", 1}, {"Sigmoid", 1}, {"Tanh", 1},
                                                             {"GlobalAveragePool", 1}, {"ReduceMean", 1}, {"AveragePool", 1}};
    std::set<std::string> shape_valued;
    for (size_t idx = 0; idx < nodes_.size(); ++idx) {
        const Node& n = nodes_[idx];
        if (n.op == "Constant" || n.out.empty()) continue;
        {
            auto mi = min_inputs.find(n.op);
            if (mi != min_inputs.end() && n.in.size() < mi->second) fail(n.label() + " has too few inputs");
        }
        bool any_shape = false, all_shape_or_const = true;
        for (const std::string& i : n.in) {
            if (i.empty()) continue;
            const std::string& r = resolve(i);
            if (inits_.count(r)) continue;
            if (shape_valued.count(r)) any_shape = true; else all_shape_or_const = false;
        }
        if (n.op == "Shape" || (any_shape && all_shape_or_const)) {
            for (const std::string& o : n.out) shape_valued.insert(o);
            continue;
        }
        if (views.count(n.op)) {
            if (n.in.empty()) fail(n.label() + " has no input");
            const std::string target = resolve(n.in[0]);
            if (target == n.out[0]) fail(n.label() + " aliases its own output");     // a damaged file: resolve() would never end
            alias_[n.out[0]] = target;
            continue;
        }
        for (const std::string& i : n.in) {
            if (i.empty()) continue;
            const std::string& r = resolve(i);
            if (inits_.count(r) || shape_valued.count(r)) continue;
            std::vector<int>& c = cons_[r];
            if (c.empty() || c.back() != int(idx)) c.push_back(int(idx));
        }
    }
}

ConvUnit Importer::conv_unit(int idx) const {
    const Node& n = nodes_[size_t(idx)];
    if (n.op != "Conv") fail("expected a Conv, found " + n.label());
    ConvUnit u;
    u.label = n.label();
    u.w = n.in.size() > 1 ? init(n.in[1]) : nullptr;
    if (!u.w || u.w->dims.size() != 4 || u.w->f.empty()) fail(u.label + ": weight must be a 4-d float initializer");
    u.cout = int(u.w->dims[0]);
    u.cin_g = int(u.w->dims[1]);
    u.k = int(u.w->dims[2]);
    if (u.w->dims[3] != u.k || (u.k != 1 && u.k != 3 && u.k != 5)) fail(u.label + ": kernel must be 1x1, 3x3 or 5x5");
    u.group = int(n.int_attr("group", 1));
    if (const auto* v = n.ints_attr("strides")) for (int64_t s : *v) if (s != 1) fail(u.label + ": stride must be 1");
    if (const auto* v = n.ints_attr("dilations")) for (int64_t s : *v) if (s != 1) fail(u.label + ": dilation must be 1");
    const auto* pads = n.ints_attr("pads");
    auto ap = n.attrs.find("auto_pad");
    const bool same = ap != n.attrs.end() && (ap->second.s == "SAME_UPPER" || ap->second.s == "SAME_LOWER");
    if (!same) {
        if (u.k > 1 && (!pads || pads->size() != 4)) fail(u.label + ": needs 'same' padding");
        if (pads) for (int64_t p : *pads) if (p != u.k / 2) fail(u.label + ": needs 'same' padding");
    }
    u.scale.assign(size_t(u.cout), 1.0);
    u.shift.assign(size_t(u.cout), 0.0);
    if (n.in.size() > 2 && !n.in[2].empty()) {
        const Tensor* b = init(n.in[2]);
        if (!b || int(b->f.size()) != u.cout) fail(u.label + ": bias must be a float initializer of size cout");
        for (int c = 0; c < u.cout; ++c) u.shift[size_t(c)] = b->f[size_t(c)];
    }
    u.out = n.out[0];
    if (const Node* bn = sole(u.out, "BatchNormalization")) {
        const Tensor *g = init(bn->in[1]), *be = init(bn->in[2]), *m = init(bn->in[3]), *v = init(bn->in[4]);
        if (!g || !be || !m || !v || int(g->f.size()) != u.cout || int(be->f.size()) != u.cout || int(m->f.size()) != u.cout || int(v->f.size()) != u.cout)
            fail(bn->label() + ": scale / bias / mean / var must be float initializers of size cout");
        const double eps = bn->float_attr("epsilon", 1e-5f);
        for (int c = 0; c < u.cout; ++c) {
            const double s = double(g->f[size_t(c)]) / std::sqrt(double(v->f[size_t(c)]) + eps);
            u.shift[size_t(c)] = double(be->f[size_t(c)]) + (u.shift[size_t(c)] - double(m->f[size_t(c)])) * s;
            u.scale[size_t(c)] = s;
        }
        u.out = bn->out[0];
    }
    if (const Node* r = sole(u.out, "Relu")) {
        u.relu = true;
        u.out = r->out[0];
    }
    return u;
}

Linear Importer::linear(const Node& n) const {
    Linear l;
    const Tensor* w = n.in.size() > 1 ? init(n.in[1]) : nullptr;
    if (!w || w->dims.size() != 2 || w->f.empty()) fail(n.label() + ": weight must be a 2-d float initializer");
    bool w_is_out_in = false;
    if (n.op == "Gemm") {
        if (n.float_attr("alpha", 1.f) != 1.f || n.float_attr("beta", 1.f) != 1.f || n.int_attr("transA", 0) != 0) fail(n.label() + ": alpha/beta/transA not supported");
        w_is_out_in = n.int_attr("transB", 0) != 0;
    }
    l.nout = int(w_is_out_in ? w->dims[0] : w->dims[1]);
    l.nin = int(w_is_out_in ? w->dims[1] : w->dims[0]);
    l.w.resize(size_t(l.nout) * l.nin);
    for (int o = 0; o < l.nout; ++o)
        for (int i = 0; i < l.nin; ++i)
            l.w[size_t(o) * l.nin + i] = w_is_out_in ? w->f[size_t(o) * l.nin + i] : w->f[size_t(i) * l.nout + o];
    l.b.assign(size_t(l.nout), 0.f);
    l.out = n.out[0];
    const Tensor* b = nullptr;
    if (n.op == "Gemm") {
        if (n.in.size() > 2 && !n.in[2].empty() && !(b = init(n.in[2]))) fail(n.label() + ": bias must be an initializer");
    } else if (const Node* add = sole(l.out, "Add")) {       // MatMul + Add(bias)
        for (const std::string& i : add->in) if (const Tensor* t = init(i)) b = t;
        if (b) l.out = add->out[0];
    }
    if (b) {
        if (int(b->f.size()) != l.nout) fail(n.label() + ": bias size does not match");
        l.b = b->f;
        l.has_bias = true;
    }
    return l;
}

void Importer::put_conv_bn(const std::string& conv, const std::string& bn, const ConvUnit& u) {
    put(conv + ".weight", u.w->dims, u.w->f);
    std::vector<float> s(u.scale.begin(), u.scale.end()), b(u.shift.begin(), u.shift.end());
    const std::vector<int64_t> d{int64_t(u.cout)};
    put(bn + ".weight", d, std::move(s));
    put(bn + ".bias", d, std::move(b));
    put(bn + ".running_mean", d, std::vector<float>(size_t(u.cout), 0.f));
    put(bn + ".running_var", d, std::vector<float>(size_t(u.cout), kLoaderBnVar));
}

void Importer::put_linear(const std::string& name, const Linear& l, bool with_bias) {
    put(name + ".weight", {int64_t(l.nout), int64_t(l.nin)}, l.w);
    if (with_bias) put(name + ".bias", {int64_t(l.nout)}, l.b);
    else if (l.has_bias) fail(name + ": this layer has no bias in the reference's modules");
}

// x -> GlobalAveragePool -> (eca: Conv1d | ca: Linear, Relu, Linear) -> HardSigmoid -> Mul(x, gate).  Returns the Mul's output.
// _EfficientChannelAttentionModule / _ChannelAttentionModule, builder_util.py:49-114.  plain_sigmoid: the gate of AlphaZero's
// ResidualBlock(use_se) = get_se("se", use_hard_sigmoid=False) on the body output (a0_resnet.py:94-95): ca form, Sigmoid activation.
std::string Importer::se_gate(const std::string& x, const std::string& prefix, std::string& se_type, bool plain_sigmoid) {
    const std::vector<int>& c = cons(x);
    int gap = pick(c, "GlobalAveragePool");
    if (gap < 0) gap = pick(c, "ReduceMean");
    if (gap < 0) gap = pick(c, "AveragePool");
    const int mul = pick(c, "Mul");
    if (gap < 0 || mul < 0 || c.size() != 2) fail("channel gate at " + prefix + ": expected exactly a pooling node and a Mul on the block input");
    std::string t = nodes_[size_t(gap)].out[0];
    const std::vector<int>& gc = cons(t);
    if (gc.size() != 1) fail("channel gate at " + prefix + ": the pooled vector must feed one layer");
    const Node& first = nodes_[size_t(gc[0])];
    if (first.op == "Conv") {
        const Tensor* w = init(first.in[1]);
        const Tensor* b = first.in.size() > 2 ? init(first.in[2]) : nullptr;
        if (!w || w->dims.size() != 3 || !b) fail(first.label() + ": eca gate needs a 1-d convolution with bias");
        const auto* pads = first.ints_attr("pads");
        if (!pads || pads->size() != 2 || (*pads)[0] != w->dims[2] / 2 || (*pads)[1] != w->dims[2] / 2) fail(first.label() + ": eca gate needs 'same' padding");
        put(prefix + ".se.body.0.weight", w->dims, w->f);
        put(prefix + ".se.body.0.bias", b->dims, b->f);
        se_type = "eca_se";
        t = first.out[0];
    } else if (is_linear(first)) {
        Linear l1 = linear(first);
        const Node* relu = sole(l1.out, "Relu");
        if (!relu) fail(first.label() + ": channel gate needs Linear -> Relu -> Linear");
        const std::vector<int>& c2 = cons(relu->out[0]);
        if (c2.size() != 1 || !is_linear(nodes_[size_t(c2[0])])) fail(first.label() + ": channel gate needs Linear -> Relu -> Linear");
        Linear l2 = linear(nodes_[size_t(c2[0])]);
        put_linear(prefix + ".se.fc.0", l1, false);
        put_linear(prefix + ".se.fc.2", l2, false);
        se_type = "ca_se";
        t = l2.out;
    } else {
        fail(first.label() + ": unsupported channel gate (ca_se and eca_se are)");
    }
    const Node* hs = sole(t, plain_sigmoid ? "Sigmoid" : "HardSigmoid");
    if (!hs) fail("channel gate at " + prefix + ": the gate activation must be " + (plain_sigmoid ? "Sigmoid" : "HardSigmoid"));
    if (plain_sigmoid) {
        if (se_type != "ca_se") fail("channel gate at " + prefix + ": the gate on a residual branch's output is Linear -> Relu -> Linear -> Sigmoid");
    } else if (std::fabs(hs->float_attr("alpha", 0.2f) - 1.f / 6.f) > 1e-6f || std::fabs(hs->float_attr("beta", 0.5f) - 0.5f) > 1e-6f) {
        fail(hs->label() + ": expected torch.nn.Hardsigmoid (alpha 1/6, beta 1/2)");
    }
    const Node& m = nodes_[size_t(mul)];
    const std::string &a = resolve(m.in[0]), &b = resolve(m.in[1]), &xs = resolve(x), &gs = resolve(hs->out[0]);
    if (!((a == xs && b == gs) || (a == gs && b == xs))) fail(m.label() + ": expected block input x gate");
    return m.out[0];
}

void Importer::run() {
    if (inputs_.size() != 1) fail("expected one graph input ('data'), found " + std::to_string(inputs_.size()));
    if (!outputs_.count("value_out") || !outputs_.count("policy_out"))
        fail("expected outputs 'value_out' and 'policy_out' (main_config value_output / policy_output)");

    // ---- stem ----
    const std::vector<int>& c0 = cons(inputs_[0]);
    if (c0.size() != 1) fail("the input must feed exactly the stem convolution");
    const ConvUnit stem = conv_unit(c0[0]);
    if (!stem.relu || stem.group != 1 || stem.k != 3) fail(stem.label + ": stem must be conv3x3 + BN + ReLU");
    const int C = stem.cout, cin = stem.cin_g;
    put_conv_bn("body_spatial.0.body.0", "body_spatial.0.body.1", stem);

    // ---- residual tower ----
    std::string cur = stem.out;
    std::vector<int> cops, kernels;
    std::vector<std::string> se_types;
    std::string conv_block;
    int head_value = -1, head_policy = -1;
    for (int blk = 1;; ++blk) {
        const std::string p = "body_spatial." + std::to_string(blk);
        std::string se = "none", x = cur;
        {
            const std::vector<int>& c = cons(cur);
            if (pick(c, "Mul") >= 0) x = se_gate(cur, p, se);
        }
        const std::vector<int> c = cons(x);
        const int add = pick(c, "Add");
        if (add < 0) {                                     // no residual join: the heads start here
            if (se != "none") fail("a channel gate feeds no residual block at " + p);
            for (int i : c) {
                const Node& n = nodes_[size_t(i)];
                if (n.op != "Conv") fail("unexpected " + n.label() + " on the tower output");
                const Tensor* w = init(n.in[1]);
                if (w && w->dims.size() == 4 && w->dims[2] == 1) { if (head_value >= 0) fail("two 1x1 heads"); head_value = i; }
                else { if (head_policy >= 0) fail("two 3x3 heads"); head_policy = i; }
            }
            if (head_value < 0 || head_policy < 0) fail("tower output must feed the value head (conv1x1) and the policy head (conv3x3)");
            break;
        }
        const int first = pick(c, "Conv");
        if (first < 0 || c.size() != 2) fail("residual block " + p + ": block input must feed one convolution and the residual Add");
        std::vector<ConvUnit> units;
        units.push_back(conv_unit(first));
        for (;;) {
            const std::vector<int>& n = cons(units.back().out);
            if (n.size() == 1 && nodes_[size_t(n[0])].op == "Conv") units.push_back(conv_unit(n[0])); else break;
            if (units.size() > 3) fail("residual block " + p + ": more than three convolutions");
        }
        const Node& an = nodes_[size_t(add)];
        // AlphaZero's ResidualBlock(use_se): the branch output goes through a channel gate (plain sigmoid) before the shortcut
        std::string branch_out = units.back().out, se_out = "none";
        {
            const std::vector<int>& n = cons(branch_out);
            if (n.size() == 2 && pick(n, "Mul") >= 0) branch_out = se_gate(branch_out, p, se_out, true);
        }
        {
            const std::vector<int>& n = cons(branch_out);
            const std::string &a = resolve(an.in[0]), &b = resolve(an.in[1]), &xs = resolve(x), &us = resolve(branch_out);
            if (n.size() != 1 || n[0] != add || !((a == xs && b == us) || (a == us && b == xs)))
                fail("residual block " + p + ": branch must end in the Add with the block input");
        }
        cur = an.out[0];
        bool post_relu = false;
        if (const Node* r = sole(cur, "Relu")) { post_relu = true; cur = r->out[0]; }
        std::string kind;
        if (units.size() == 3) {
            const ConvUnit &e = units[0], &d = units[1], &pr = units[2];
            const bool ok = e.k == 1 && e.group == 1 && e.relu && e.cin_g == C && d.group == e.cout && d.cin_g == 1 && d.cout == e.cout && d.k > 1 && d.relu &&
                            pr.k == 1 && pr.group == 1 && !pr.relu && pr.cin_g == e.cout && pr.cout == C && !post_relu;
            if (!ok) fail("residual block " + p + ": not a mobile bottleneck (1x1+ReLU, depthwise+ReLU, 1x1)");
            if (se_out != "none") fail("residual block " + p + ": a bottleneck block gates its input, not its branch output");
            kind = "mobile_bottlekneck_res_block";
            put_conv_bn(p + ".body.0", p + ".body.1", e);
            put_conv_bn(p + ".body.3", p + ".body.4", d);
            put_conv_bn(p + ".body.6", p + ".body.7", pr);
            cops.push_back(e.cout);
            kernels.push_back(d.k);
        } else if (units.size() == 2) {
            const ConvUnit &a = units[0], &b = units[1];
            if (!(a.k == 3 && b.k == 3 && a.group == 1 && b.group == 1 && a.cin_g == C && a.cout == C && b.cin_g == C && b.cout == C && a.relu))
                fail("residual block " + p + ": not a pair of dense 3x3 convolutions");
            if (b.relu && !post_relu) kind = "classical_res_block";           // x + ReLU(BN(conv(..)))        builder_util.py:401-434
            else if (!b.relu && post_relu) kind = "a0_res_block";             // ReLU(x + BN(conv(..)))        a0_resnet.py:72-107
            else fail("residual block " + p + ": unknown activation placement around the residual Add");
            // where the reference's two dense blocks put their gate: ClassicalResidualBlock on the block input (hard-sigmoid,
            // builder_util.py:416,431-433), AlphaZero's ResidualBlock on the body output (plain sigmoid, a0_resnet.py:94-107)
            if (kind == "classical_res_block" && se_out != "none") fail("residual block " + p + ": a classical residual block gates its input, not its branch output");
            if (kind == "a0_res_block" && se != "none") fail("residual block " + p + ": an AlphaZero residual block gates its branch output, not its input");
            if (kind == "a0_res_block") se = se_out;
            put_conv_bn(p + ".body.0", p + ".body.1", a);
            put_conv_bn(p + ".body.3", p + ".body.4", b);
            cops.push_back(C);
            kernels.push_back(3);
        } else {
            fail("residual block " + p + ": expected two or three convolutions");
        }
        if (conv_block.empty()) conv_block = kind;
        else if (conv_block != kind) fail("residual block " + p + ": mixed block families");
        se_types.push_back(se);
    }
    if (conv_block.empty()) conv_block = "mobile_bottlekneck_res_block";

    // ---- policy head (_PolicyHead, builder_util.py:206-243) ----
    const ConvUnit p0 = conv_unit(head_policy);
    if (!(p0.k == 3 && p0.group == 1 && p0.relu && p0.cin_g == C)) fail(p0.label + ": policy head must start with conv3x3 + BN + ReLU");
    const Node* p1n = sole(p0.out, "Conv");
    if (!p1n) fail(p0.label + ": policy head needs a second convolution");
    const ConvUnit p1 = conv_unit(int(p1n - nodes_.data()));
    if (p1.cin_g != p0.cout || p1.group != 1) fail(p1.label + ": policy head channel mismatch");
    put_conv_bn("policy_head.body.0", "policy_head.body.1", p0);
    bool policy_map;
    int n_labels = 0;
    if (is_output(p1.out, "policy_out")) {
        policy_map = true;
        for (double s : p1.shift) if (s != 0.0) fail(p1.label + ": the policy-map convolution has no bias in the reference's head");
        std::vector<float> w(p1.w->f);
        const size_t per = w.size() / size_t(p1.cout);
        for (int c = 0; c < p1.cout; ++c) for (size_t i = 0; i < per; ++i) w[size_t(c) * per + i] = float(double(w[size_t(c) * per + i]) * p1.scale[size_t(c)]);
        put("policy_head.body.3.weight", p1.w->dims, std::move(w));
    } else {
        policy_map = false;
        if (!p1.relu) fail(p1.label + ": flat policy head needs BN + ReLU before the Linear layer");
        const std::vector<int>& c = cons(p1.out);
        if (c.size() != 1 || !is_linear(nodes_[size_t(c[0])])) fail(p1.label + ": flat policy head needs one Linear layer");
        const Linear l = linear(nodes_[size_t(c[0])]);
        if (!is_output(l.out, "policy_out") || l.nin != p1.cout * 64) fail("flat policy head does not end in policy_out");
        put_conv_bn("policy_head.body.3", "policy_head.body2.0", p1);
        put_linear("policy_head.body3.0", l, true);
        n_labels = l.nout;
    }

    // ---- value head (_ValueHead, builder_util.py:246-310) ----
    const ConvUnit v0 = conv_unit(head_value);
    if (!(v0.k == 1 && v0.group == 1 && v0.relu && v0.cin_g == C)) fail(v0.label + ": value head must start with conv1x1 + BN + ReLU");
    put_conv_bn("value_head.body.0", "value_head.body.1", v0);
    const std::vector<int> vc = cons(v0.out);
    bool wdl = false;
    int fc = 0;
    if (vc.size() == 1 && is_linear(nodes_[size_t(vc[0])]) && sole(linear(nodes_[size_t(vc[0])]).out, "Relu")) {
        const Linear l1 = linear(nodes_[size_t(vc[0])]);
        const Node* relu = sole(l1.out, "Relu");
        const std::vector<int>* c2 = relu ? &cons(relu->out[0]) : nullptr;
        if (!c2 || c2->size() != 1 || !is_linear(nodes_[size_t((*c2)[0])])) fail("value head: expected Linear -> ReLU -> Linear -> Tanh");
        const Linear l2 = linear(nodes_[size_t((*c2)[0])]);
        const Node* th = sole(l2.out, "Tanh");
        if (!th || !is_output(th->out[0], "value_out") || l2.nout != 1 || l1.nin != v0.cout * 64) fail("value head: expected Linear -> ReLU -> Linear -> Tanh -> value_out");
        put_linear("value_head.body_final.0", l1, true);
        put_linear("value_head.body_final.2", l2, true);
        fc = l1.nout;
    } else {
        // WDL + plies-to-end: value_out = softmax(wdl)[win] - softmax(wdl)[loss] is derived, not learned (rise_mobile_v3.py forward)
        const Linear *lw = nullptr, *lp = nullptr;
        Linear a, b;
        if (vc.size() == 2 && is_linear(nodes_[size_t(vc[0])]) && is_linear(nodes_[size_t(vc[1])])) {
            a = linear(nodes_[size_t(vc[0])]);
            b = linear(nodes_[size_t(vc[1])]);
            lw = a.nout == 3 ? &a : b.nout == 3 ? &b : nullptr;
            lp = a.nout == 1 ? &a : b.nout == 1 ? &b : nullptr;
        }
        Linear zero_plys;
        if (vc.size() == 1 && is_linear(nodes_[size_t(vc[0])])) {              // plies-to-end branch pruned from the file: it is no output then
            a = linear(nodes_[size_t(vc[0])]);
            if (a.nout == 3) {
                lw = &a;
                zero_plys.nout = 1;
                zero_plys.nin = a.nin;
                zero_plys.w.assign(size_t(a.nin), 0.f);
                zero_plys.b.assign(1, 0.f);
                lp = &zero_plys;
            }
        }
        if (!lw || !lp || lw->nin != v0.cout * 64 || lp->nin != v0.cout * 64) fail("value head: expected either the tanh head or the WDL (3) + plies-to-end (1) pair");
        if (lp != &zero_plys && !sole(lp->out, "Sigmoid")) fail("value head: plies-to-end output must end in Sigmoid");
        put_linear("value_head.body_wdl.0", *lw, true);
        put_linear("value_head.body_plys.0", *lp, true);
        wdl = true;
    }

    // ---- meta (keys of crazyara_amd/netfile.py:export_rise) ----
    auto join = [](const std::vector<std::string>& v) { std::string s; for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + v[i]; return s; };
    std::vector<std::string> ks, cs;
    for (int k : kernels) ks.push_back(std::to_string(k));
    for (int c : cops) cs.push_back(std::to_string(c));
    std::map<std::string, std::string>& m = nf_.meta;
    m.clear();
    m["arch"] = "rise";
    m["source"] = "onnx";
    m["producer"] = producer_;
    std::string ver;
    {
        const int v = read_version_from_string(file_name_);
        ver = std::to_string(v / 1000000) + "." + std::to_string(v / 1000 % 1000);
    }
    m["input_version"] = ver;
    m["nb_input_channels"] = std::to_string(cin);
    m["channels"] = std::to_string(C);
    m["channels_operating_init"] = std::to_string(cops.empty() ? C : cops[0]);
    m["channel_expansion"] = "0";
    m["channels_operating"] = join(cs);        // per block, as found (the loader prefers it over init/expansion)
    m["kernels"] = join(ks);
    m["se_types"] = join(se_types);
    m["channels_value_head"] = std::to_string(v0.cout);
    m["value_fc_size"] = std::to_string(fc);
    m["channels_policy_head"] = std::to_string(p1.cout);
    m["use_wdl"] = wdl ? "1" : "0";
    m["use_plys_to_end"] = wdl ? "1" : "0";
    m["conv_block"] = conv_block;
    m["select_policy_from_plane"] = policy_map ? "1" : "0";
    m["n_labels"] = std::to_string(n_labels);
    finish();
}

void Importer::finish() {
    size_t total = 0;
    for (const Out& o : out_) total += (o.data.size() * 4 + 15) / 16 * 16;
    nf_.blob.assign(total, 0);
    nf_.tensors.clear();
    size_t off = 0;
    for (const Out& o : out_) {
        if (nf_.tensors.count(o.name)) fail("duplicate tensor " + o.name);
        std::memcpy(nf_.blob.data() + off, o.data.data(), o.data.size() * 4);
        TensorView tv;
        tv.shape = o.dims;
        tv.data = reinterpret_cast<const float*>(nf_.blob.data() + off);
        if (tv.numel() != int64_t(o.data.size())) fail("tensor " + o.name + " size mismatch");
        nf_.tensors[o.name] = tv;
        off += (o.data.size() * 4 + 15) / 16 * 16;
    }
}

}  // namespace

void import_onnx_bytes(const void* data, size_t size, const std::string& model_file_name, NetFile& nf) {
    Slice s;
    s.p = static_cast<const uint8_t*>(data);
    s.n = size;
    Importer imp(s, model_file_name, nf);
    imp.run();
}

void import_onnx(const std::string& path, NetFile& nf) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open model file " + path);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const size_t sl = path.find_last_of('/');
    import_onnx_bytes(raw.data(), raw.size(), sl == std::string::npos ? path : path.substr(sl + 1), nf);
}

void write_cranet(const NetFile& nf, const std::string& path) {
    // tensors in offset order so that the blob is reproduced byte for byte
    std::vector<std::pair<const float*, std::string>> order;
    for (const auto& kv : nf.tensors) order.push_back({kv.second.data, kv.first});
    std::sort(order.begin(), order.end());
    std::ostringstream h;
    for (const auto& kv : nf.meta) h << kv.first << ' ' << kv.second << '\n';
    std::vector<char> blob;
    for (const auto& o : order) {
        const TensorView& tv = nf.tensors.at(o.second);
        h << "tensor " << o.second << ' ' << tv.shape.size();
        for (int64_t d : tv.shape) h << ' ' << d;
        h << ' ' << blob.size() << '\n';
        const char* src = reinterpret_cast<const char*>(tv.data);
        blob.insert(blob.end(), src, src + size_t(tv.numel()) * 4);
        blob.resize((blob.size() + 15) / 16 * 16, 0);
    }
    const std::string header = h.str();
    const uint64_t hlen = header.size();
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("cannot write " + path);
    f.write("CRANET01", 8);
    f.write(reinterpret_cast<const char*>(&hlen), 8);
    f.write(header.data(), std::streamsize(header.size()));
    f.write(blob.data(), std::streamsize(blob.size()));
    if (!f) throw std::runtime_error("write failed: " + path);
}

}  // namespace cra
