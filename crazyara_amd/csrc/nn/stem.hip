// Stem kernel for gfx950: NCHW fp32 planes -> conv3x3(cin -> 256) + BN + ReLU -> NHWC f16 residual stream, one launch,
// one workgroup per board (replaces the layout-transform launch + the generic conv launch).
//
// Reference semantics: _Stem (DeepCrazyhouse/src/domain/neural_net/architectures/pytorch/builder_util.py:154-178) applied to
// the planes NeuralNetAPI::predict receives (engine/src/nn/neuralnetapi.h:230-237).
//
// The planes of a board (cin x 64 floats) are transposed into a [65][cin_pad + 8] f16 tile in LDS (row 64 = zeros); the conv
// is 9 shifted GEMMs on v_mfma_f32_32x32x16_f16, wave v owning couts 32v..32v+31 for all 64 squares; the result tile is
// staged in LDS and leaves as 16-byte coalesced stores.  HBM per board: cin*256 B in, 32 KB out; weights 9*cin_pad*512 B from L2.
#include "kernels.h"
#include "device_utils.h"
#include "../chess/planes.h"

namespace cra {

namespace {
constexpr int ST_OROW = 256 + 8;                     // halves, output staging tile pitch
constexpr int ST_OUT_BYTES = 64 * ST_OROW * 2;
constexpr int ST_WIN = 16;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t st_pack_relu_h2(float a, float ca, float b, float cb) {
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2\n\tv_fma_mixhi_f16 %0, %3, 1.0, %4\n\tv_pk_max_f16 %0, %0, 0" : "=&v"(r) : "v"(a), "v"(ca), "v"(b), "v"(cb));
    return r;
}
}  // namespace

// to_global = false: the result stays in LDS as the [64][ST_OROW] tile at offset 0 of the dynamic segment, which is exactly the
// residual-stream tile the tower opens with (forward.hip runs stem, tower and head of a board in one launch)
template <int NKS>
__device__ __forceinline__ void stem_body(const StemArgs& a, const bool to_global) {
    using frag = half8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* ot = reinterpret_cast<half_t*>(smem);                    // [64][ST_OROW] output staging
    half_t* pl = reinterpret_cast<half_t*>(smem + ST_OUT_BYTES);     // [65][prow] planes tile
    constexpr int prow = NKS * 16 + 8;               // halves; (cin_pad + 8) / 8 is odd: 32 consecutive rows -> distinct bank slots
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    const frag* sp = reinterpret_cast<const frag*>(a.stem_w) + size_t(wv) * a.stem_wave_frags * 64 + lane;
    frag win[ST_WIN];
#pragma unroll
    for (int i = 0; i < ST_WIN; ++i) win[i] = sp[i * 64];
    if (a.descs != nullptr) {
        // search lane: the planes of this board are computed here from its 192-byte descriptor (one launch instead of a plane-builder
        // launch + 8.7-20 KB per board through HBM); same function, same values as planes_from_desc_kernel + the conversion below
        __shared__ BoardDesc sd;
        const bool valid = b < a.n_valid;
        if (valid && tid < int(sizeof(BoardDesc) / 8))
            reinterpret_cast<uint64_t*>(&sd)[tid] = reinterpret_cast<const uint64_t*>(static_cast<const BoardDesc*>(a.descs) + b)[tid];
        for (int i = tid; i < 65 * prow / 2; i += 512) reinterpret_cast<uint32_t*>(pl)[i] = 0u;
        __syncthreads();
        if (valid) {
#pragma unroll 1
            for (int e = tid; e < a.cin * 64; e += 512) pl[(e & 63) * prow + (e >> 6)] = half_t(plane_value(sd, a.layout, true, e >> 6, e & 63));
        }
    } else {
        const float* pb = a.planes + size_t(b) * a.cin * 64;
        float pv[12];                                // cin <= 96 -> at most 12 values per thread, all loads in flight together
#pragma unroll
        for (int i = 0; i < 12; ++i) pv[i] = (tid + i * 512) < a.cin * 64 ? pb[tid + i * 512] : 0.f;
        for (int i = tid; i < 65 * prow / 2; i += 512) reinterpret_cast<uint32_t*>(pl)[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int e = tid + i * 512;
            if (e < a.cin * 64) pl[(e & 63) * prow + (e >> 6)] = half_t(pv[i]);
        }
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
#pragma unroll
    for (int q = 0; q < 9 * NKS; ++q) {              // (tap, k-step) units, tap-major; everything about a unit is compile-time
        const int tap = q / NKS, ks = q % NKS;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int sq0 = l31, sq1 = 32 + l31;
        const int y0 = (sq0 >> 3) + dy, x0 = (sq0 & 7) + dx, y1 = (sq1 >> 3) + dy;
        const bool okx = unsigned(x0) < 8u;
        const int r0 = (okx && unsigned(y0) < 8u) ? sq0 + dy * 8 + dx : 64;
        const int r1 = (okx && unsigned(y1) < 8u) ? sq1 + dy * 8 + dx : 64;
        const frag b0 = *reinterpret_cast<const frag*>(pl + r0 * prow + ks * 16 + lh * 8);
        const frag b1 = *reinterpret_cast<const frag*>(pl + r1 * prow + ks * 16 + lh * 8);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(win[q % ST_WIN], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(win[q % ST_WIN], b1, acc[1], 0, 0, 0);
        win[q % ST_WIN] = sp[(q + ST_WIN) * 64];     // (the stream ends with ST_WIN zero fragments)
    }
    mfma_retire(acc[0], acc[1]);                     // the last MFMAs retire before the asm pack reads them (device_utils.h)
    f32x4 bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = reinterpret_cast<const f32x4*>(a.stem_b + (wv * 2 + lh) * 16)[i];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {             // 4 consecutive couts: rows 8*g4 + 4*lh + 0..3
            uint2 o;
            o.x = st_pack_relu_h2(acc[ct][g4 * 4 + 0], bias[g4][0], acc[ct][g4 * 4 + 1], bias[g4][1]);
            o.y = st_pack_relu_h2(acc[ct][g4 * 4 + 2], bias[g4][2], acc[ct][g4 * 4 + 3], bias[g4][3]);
            *reinterpret_cast<uint2*>(ot + (ct * 32 + l31) * ST_OROW + wv * 32 + g4 * 8 + lh * 4) = o;
        }
    __syncthreads();
    if (!to_global) return;
    half_t* xb = reinterpret_cast<half_t*>(a.x) + size_t(b) * 64 * 256;
    for (int i = tid; i < 64 * 32; i += 512) {
        const int r = i >> 5, v = i & 31;
        *reinterpret_cast<uint4*>(xb + size_t(r) * 256 + v * 8) = *reinterpret_cast<const uint4*>(ot + r * ST_OROW + v * 8);
    }
}

#ifndef CRA_FORWARD_TU
template <int NKS>
__global__ __launch_bounds__(512) void stem_kernel(const StemArgs a) { stem_body<NKS>(a, true); }

void launch_stem(const StemArgs& a, hipStream_t s) {
    const int nks = a.cin_pad / 16;
    const size_t lds = ST_OUT_BYTES + size_t(65) * (a.cin_pad + 8) * 2;
    switch (nks) {
        case 3: hipLaunchKernelGGL(stem_kernel<3>, dim3(a.batch), dim3(512), lds, s, a); break;
        case 4: hipLaunchKernelGGL(stem_kernel<4>, dim3(a.batch), dim3(512), lds, s, a); break;
        case 5: hipLaunchKernelGGL(stem_kernel<5>, dim3(a.batch), dim3(512), lds, s, a); break;
        default: hipLaunchKernelGGL(stem_kernel<6>, dim3(a.batch), dim3(512), lds, s, a); break;
    }
}

#endif  // CRA_FORWARD_TU

}  // namespace cra
