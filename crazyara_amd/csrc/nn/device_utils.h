// Device-side helpers shared by the gfx950 kernels: MFMA fragment types, f16/f32 vector load/store, DPP moves.
#pragma once
#include "kernels.h"

namespace cra {

typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) float8 {
    f32x4 lo, hi;
};

template <typename T> struct VT;
template <> struct VT<half_t> {
    typedef half8 frag;                // 8 k-values of one MFMA operand row/col
    static constexpr int KC = 256;     // channels staged in LDS per chunk
};
template <> struct VT<float> {
    typedef float8 frag;
    static constexpr int KC = 128;
};

// D(16x16) += A(16 x 32) * B(32 x 16); both operands hold k = (lane>>4)*8 + j in element j.
__device__ __forceinline__ void mma_k32(const half8& a, const half8& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// exact-f32 path: 8 x (16x16x4).  MFMA j consumes element j of both fragments, i.e. k = (lane>>4)*8 + j --
// the same bijection on both operands, so the K-sum is complete whatever order the hardware walks it in.
__device__ __forceinline__ void mma_k32(const float8& a, const float8& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], c, 0, 0, 0);
}

// An MFMA's result registers must not be read by a VALU instruction for (passes + 2) issue slots after the MFMA.  The compiler pads
// its OWN instructions with s_nop; it neither looks inside inline asm (the v_fma_mix / v_cvt_pk epilogues of these kernels) nor keeps
// an `asm volatile("s_nop ...")` behind the MFMAs -- without a data dependence it schedules MFMAs across it (seen in the one-launch fp8
// forward: the stem's last MFMA landed two slots in front of the asm pack, which then read the accumulator one k-step short on some
// issues).  Hence the accumulator goes THROUGH the wait: MFMA -> this asm -> reader is a dependence chain the scheduler cannot reorder.
// 20 slots cover the 16-pass v_mfma_f32_32x32x64_f8f6f4.  Every accumulator an asm epilogue reads goes through one of these, behind
// its last MFMA.  (scripts/isa_mfma_hazards.py checks the compiled kernels for such distances; tests/test_isa_hazards.py runs it.)
template <typename Acc> __device__ __forceinline__ void mfma_retire(Acc& a) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a)); }
template <typename Acc> __device__ __forceinline__ void mfma_retire(Acc& a, Acc& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }
template <typename Acc> __device__ __forceinline__ void mfma_retire(Acc& a, Acc& b, Acc& c, Acc& d) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

__device__ __forceinline__ float to_f(half_t v) { return float(v); }
__device__ __forceinline__ float to_f(float v) { return v; }

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<half_t>(const half_t* p, float (&v)[8]) {
    half8 h = *reinterpret_cast<const half8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = float(h[j]);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<half_t>(half_t* p, const float (&v)[8]) {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = half_t(v[j]);
    *reinterpret_cast<half8*>(p) = h;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<half_t>(const half_t* p, float (&v)[4]) {
    half4 h = *reinterpret_cast<const half4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = float(h[j]);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = a[j];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<half_t>(half_t* p, const float (&v)[4]) {
    half4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = half_t(v[j]);
    *reinterpret_cast<half4*>(p) = h;
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = v[j];
    *reinterpret_cast<f32x4*>(p) = a;
}

__device__ __forceinline__ float hard_sigmoid(float v) { return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f); }

// DPP lane moves inside a 16-lane row (bound_ctrl: lanes shifted in from outside the row read 0)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_ROR8 = 0x128;

}  // namespace cra
