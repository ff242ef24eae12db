// ORACLE BUILD SUPPORT (test infrastructure only) -- a minimal stand-in for <blaze/Math.h>.
//
// The reference's search code (engine/src/node.{h,cpp}, nodedata.{h,cpp}, util/blazeutil.h, evalinfo.cpp, searchthread.cpp,
// agents/*.cpp) uses blaze only as a dense-vector container with a handful of element-wise expressions
// (DynamicVector<T> + argmax, max, sum, pow, softmax, subvector and the arithmetic operators).  blaze itself
// is an empty submodule in the mount (engine/3rdparty/blaze), so oracle/ref/build_ref.py compiles the reference's own
// sources against this header instead.  Nothing here is copied from blaze (its source is not available); the semantics are:
//
//   * every expression is evaluated eagerly, element by element, in the natural C++ promotion of its operand types
//     (float * float -> float, float * double -> double, uint32_t + double -> double ...), left to right as written in the
//     reference's source -- i.e. blaze's expression templates WITHOUT its algebraic restructuring of scalar factors;
//   * reductions (sum, max, argmax) run sequentially from index 0 in the element type; argmax returns the FIRST maximum
//     (blaze's documented behaviour);
//   * softmax(v) = exp(v) / sum(exp(v)).
// Where blaze would round differently (SIMD reduction order of sum(), restructured scalar products) results may differ in
// the last ulp of a float; the parity tests state this next to the comparisons it could touch (prior renormalisation).
#pragma once
// standard headers the reference's sources get transitively through the real <blaze/Math.h> and do not include themselves
#include <cassert>
#include <cfloat>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <numeric>
#include <random>
#include <string>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <type_traits>
#include <utility>
#include <vector>

namespace blaze {

template <typename T, size_t N> class StaticVector;   // named by `using` declarations only
template <typename T, size_t N> class HybridVector;

template <typename T>
class DynamicVector {
    std::vector<T> v_;

public:
    using ElementType = T;
    using iterator = typename std::vector<T>::iterator;
    using const_iterator = typename std::vector<T>::const_iterator;

    DynamicVector() = default;
    explicit DynamicVector(size_t n) : v_(n) {}
    DynamicVector(size_t n, const T& init) : v_(n, init) {}
    DynamicVector(const DynamicVector&) = default;
    DynamicVector(DynamicVector&&) = default;
    template <typename U>
    DynamicVector(const DynamicVector<U>& o) : v_(o.size()) {
        for (size_t i = 0; i < v_.size(); ++i) v_[i] = static_cast<T>(o[i]);
    }
    DynamicVector& operator=(const DynamicVector&) = default;
    DynamicVector& operator=(DynamicVector&&) = default;
    template <typename U>
    DynamicVector& operator=(const DynamicVector<U>& o) {
        v_.resize(o.size());
        for (size_t i = 0; i < v_.size(); ++i) v_[i] = static_cast<T>(o[i]);
        return *this;
    }
    template <typename S, typename = std::enable_if_t<std::is_arithmetic<S>::value>>
    DynamicVector& operator=(S scalar) {             // homogeneous assignment: every element = scalar
        for (T& e : v_) e = static_cast<T>(scalar);
        return *this;
    }

    size_t size() const { return v_.size(); }
    size_t capacity() const { return v_.capacity(); }
    void resize(size_t n, bool preserve = true) {
        if (!preserve) v_.clear();
        v_.resize(n);
    }
    void extend(size_t n, bool preserve = true) { resize(v_.size() + n, preserve); }
    void reserve(size_t n) { v_.reserve(n); }
    T* data() { return v_.data(); }
    const T* data() const { return v_.data(); }
    T& operator[](size_t i) { return v_[i]; }
    const T& operator[](size_t i) const { return v_[i]; }
    iterator begin() { return v_.begin(); }
    iterator end() { return v_.end(); }
    const_iterator begin() const { return v_.begin(); }
    const_iterator end() const { return v_.end(); }

    template <typename S, typename = std::enable_if_t<std::is_arithmetic<S>::value>>
    DynamicVector& operator/=(S s) {
        for (T& e : v_) e = static_cast<T>(e / s);
        return *this;
    }
    template <typename S, typename = std::enable_if_t<std::is_arithmetic<S>::value>>
    DynamicVector& operator*=(S s) {
        for (T& e : v_) e = static_cast<T>(e * s);
        return *this;
    }
    template <typename U>
    DynamicVector& operator+=(const DynamicVector<U>& o) {
        for (size_t i = 0; i < v_.size(); ++i) v_[i] = static_cast<T>(v_[i] + o[i]);
        return *this;
    }
};

template <typename A, typename B> using MulT = decltype(std::declval<A>() * std::declval<B>());
template <typename A, typename B> using AddT = decltype(std::declval<A>() + std::declval<B>());
template <typename A, typename B> using DivT = decltype(std::declval<A>() / std::declval<B>());
template <typename S> using IfScalar = std::enable_if_t<std::is_arithmetic<S>::value>;

// scalar * vector, vector * scalar
template <typename S, typename T, typename = IfScalar<S>>
DynamicVector<MulT<S, T>> operator*(S s, const DynamicVector<T>& v) {
    DynamicVector<MulT<S, T>> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = s * v[i];
    return r;
}
template <typename S, typename T, typename = IfScalar<S>>
DynamicVector<MulT<T, S>> operator*(const DynamicVector<T>& v, S s) {
    DynamicVector<MulT<T, S>> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = v[i] * s;
    return r;
}
// vector * vector: component-wise (two column vectors)
template <typename A, typename B>
DynamicVector<MulT<A, B>> operator*(const DynamicVector<A>& a, const DynamicVector<B>& b) {
    DynamicVector<MulT<A, B>> r(a.size());
    for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] * b[i];
    return r;
}
template <typename A, typename B>
DynamicVector<AddT<A, B>> operator+(const DynamicVector<A>& a, const DynamicVector<B>& b) {
    DynamicVector<AddT<A, B>> r(a.size());
    for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] + b[i];
    return r;
}
template <typename A, typename B>
DynamicVector<AddT<A, B>> operator-(const DynamicVector<A>& a, const DynamicVector<B>& b) {
    DynamicVector<AddT<A, B>> r(a.size());
    for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] - b[i];
    return r;
}
// vector + scalar
template <typename T, typename S, typename = IfScalar<S>>
DynamicVector<AddT<T, S>> operator+(const DynamicVector<T>& v, S s) {
    DynamicVector<AddT<T, S>> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = v[i] + s;
    return r;
}
// scalar / vector, vector / scalar
template <typename S, typename T, typename = IfScalar<S>>
DynamicVector<DivT<S, T>> operator/(S s, const DynamicVector<T>& v) {
    DynamicVector<DivT<S, T>> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = s / v[i];
    return r;
}
template <typename T, typename S, typename = IfScalar<S>>
DynamicVector<DivT<T, S>> operator/(const DynamicVector<T>& v, S s) {
    DynamicVector<DivT<T, S>> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = v[i] / s;
    return r;
}

template <typename T>
DynamicVector<T> subvector(const DynamicVector<T>& v, size_t start, size_t n) {
    DynamicVector<T> r(n);
    for (size_t i = 0; i < n; ++i) r[i] = v[start + i];
    return r;
}

template <typename T>
T sum(const DynamicVector<T>& v) {
    T s = T(0);
    for (size_t i = 0; i < v.size(); ++i) s += v[i];
    return s;
}
template <typename T>
T max(const DynamicVector<T>& v) {
    T m = v[0];
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] > m) m = v[i];
    return m;
}
template <typename T>
size_t argmax(const DynamicVector<T>& v) {
    if (v.size() == 0) return 0;
    size_t best = 0;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] > v[best]) best = i;
    return best;
}
template <typename T, typename S, typename = IfScalar<S>>
DynamicVector<T> pow(const DynamicVector<T>& v, S e) {
    DynamicVector<T> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = static_cast<T>(std::pow(v[i], e));
    return r;
}
template <typename T>
DynamicVector<T> exp(const DynamicVector<T>& v) {
    DynamicVector<T> r(v.size());
    for (size_t i = 0; i < v.size(); ++i) r[i] = std::exp(v[i]);
    return r;
}
template <typename T>
DynamicVector<T> softmax(const DynamicVector<T>& v) {
    DynamicVector<T> r = exp(v);
    r /= sum(r);
    return r;
}

}  // namespace blaze
