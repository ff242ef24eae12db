// Dense residual tower for gfx950: every ClassicalResidualBlock / AlphaZero ResidualBlock of a net in ONE launch.
//
// Reference semantics:
//   ClassicalResidualBlock  x + ReLU(BN(conv3x3(ReLU(BN(conv3x3(x))))))      builder_util.py:401-434
//   ResidualBlock (A0)      ReLU(x + BN(conv3x3(ReLU(BN(conv3x3(x))))))      a0_resnet.py:72-107
//
// One workgroup (8 waves) = one board for the whole tower.  The 64 x 256 residual stream and the intermediate of a block
// stay in LDS as f16 tiles (64 squares + one zero row for the board edge); a 3x3 convolution is 9 shifted GEMMs on
// v_mfma_f32_32x32x16_f16: wave v owns couts 32v..32v+31 for all 64 squares (two 32 x 32 accumulators), A = its weight
// stream (host-packed in consumption order, 16 fragments in flight, lines touched in L2 two taps early by the workgroups of
// an XCD in turn -- tower.hip explains why), B = neighbour rows of the tile.  Per block and wave: 2 x 288 MFMAs.
//   conv 1 epilogue: + BN bias, ReLU -> f16, a lane's 16 rows stored as 16 consecutive K positions of the intermediate
//                    (two 16-byte stores); conv 2's weights are packed in that K order (kernels.h: tower_row_of_position)
//   conv 2 epilogue: + BN bias, activation / shortcut in the block type's order, rounded once to f16 into the stream tile in
//                    natural channel order (the head and the next block read it as is)
#include "kernels.h"
#include "device_utils.h"

namespace cra {

namespace {
constexpr int RT_C = 256;
constexpr int RT_ROW = RT_C + 8;                     // halves; 528-byte pitch: 32 consecutive rows hit distinct 16-byte bank slots
constexpr int RT_TILE_BYTES = 65 * RT_ROW * 2;       // 64 squares + a zero row
constexpr int RT_LDS_BYTES = 2 * RT_TILE_BYTES;
constexpr int RT_WIN = 16;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pack_relu_h2(float a, float ca, float b, float cb) {
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2\n\tv_fma_mixhi_f16 %0, %3, 1.0, %4\n\tv_pk_max_f16 %0, %0, 0" : "=&v"(r) : "v"(a), "v"(ca), "v"(b), "v"(cb));
    return r;
}
__device__ __forceinline__ void mma32(const half8& a, const half8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// LDS row of the neighbour of square sq for tap (dy, dx), or the zero row 64
__device__ __forceinline__ int nbr_row(int sq, int dy, int dx) {
    const int y = (sq >> 3) + dy, x = (sq & 7) + dx;
    return (unsigned(y) < 8u && unsigned(x) < 8u) ? sq + dy * 8 + dx : 64;
}
}  // namespace

size_t restower_lds_bytes() { return RT_LDS_BYTES; }

__global__ __launch_bounds__(512) void restower_kernel(const ResTowerArgs a) {
    using frag = half8;
    constexpr int ROW = RT_ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* X = reinterpret_cast<half_t*>(smem);
    half_t* T = reinterpret_cast<half_t*>(smem + RT_TILE_BYTES);

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- open my weight stream, then bring the board in ----
    const frag* sp = reinterpret_cast<const frag*>(a.wstream) + size_t(wv) * a.wstream_wave_frags * 64 + lane;
    const char* sline = reinterpret_cast<const char*>(a.wstream) + size_t(wv) * a.wstream_wave_frags * 1024;   // warm-up cursor
    const float* bp = a.bstream + size_t(wv) * a.bstream_wave_floats + lh * 16;
    frag win[RT_WIN];
#pragma unroll
    for (int q = 0; q < RT_WIN; ++q) win[q] = sp[q * 64];
    {
        const half_t* xb = reinterpret_cast<const half_t*>(a.x) + size_t(b) * 64 * RT_C;
        for (int i = tid; i < 64 * 32; i += 512) {
            const int r = i >> 5, v = i & 31;
            *reinterpret_cast<uint4*>(X + r * ROW + v * 8) = *reinterpret_cast<const uint4*>(xb + size_t(r) * RT_C + v * 8);
        }
        if (tid < ROW / 2) {                          // zero rows of both tiles
            reinterpret_cast<uint32_t*>(X + 64 * ROW)[tid] = 0u;
            reinterpret_cast<uint32_t*>(T + 64 * ROW)[tid] = 0u;
        }
    }
    __syncthreads();

    const int pf_slot = (b >> 3) & 31;
    int pf_old = 0, pf_sink = 0;
    const long long stream_bytes = a.wstream_wave_frags * 1024;
    long long consumed = 0;                          // bytes of my stream the taps so far have used

    // one 3x3 convolution of the tile `src` into acc (9 taps x 8 steps x 4 MFMAs)
    auto conv3x3 = [&](const half_t* src, f32x16 (&acc)[2]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            pf_sink ^= pf_old;
            {   // L2 warm-up: my share of the lines two taps (32 KiB of stream) ahead
                const long long off = consumed + 2 * 16384 + (lane * 32 + pf_slot) * 128;
                pf_old = (lane < 4 && off < stream_bytes) ? *reinterpret_cast<const int*>(sline + off) : 0;
            }
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const half_t* r0 = src + nbr_row(l31, dy, dx) * ROW + lh * 8;
            const half_t* r1 = src + nbr_row(32 + l31, dy, dx) * ROW + lh * 8;
            frag bfa[4], bfb[4];                     // [k-step parity][square tile]
            bfa[0] = *reinterpret_cast<const frag*>(r0);      bfa[1] = *reinterpret_cast<const frag*>(r1);
            bfa[2] = *reinterpret_cast<const frag*>(r0 + 16); bfa[3] = *reinterpret_cast<const frag*>(r1 + 16);
#pragma unroll
            for (int s = 0; s < 8; ++s) {            // step = k-steps 2s, 2s+1
                frag (&cur)[4] = (s & 1) ? bfb : bfa;
                frag (&nxt)[4] = (s & 1) ? bfa : bfb;
                if (s + 1 < 8) {
                    nxt[0] = *reinterpret_cast<const frag*>(r0 + (s + 1) * 32);      nxt[1] = *reinterpret_cast<const frag*>(r1 + (s + 1) * 32);
                    nxt[2] = *reinterpret_cast<const frag*>(r0 + (s + 1) * 32 + 16); nxt[3] = *reinterpret_cast<const frag*>(r1 + (s + 1) * 32 + 16);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) mma32(win[s * 2 + (i >> 1)], cur[i], acc[i & 1]);
#pragma unroll
                for (int e = 0; e < 2; ++e) win[s * 2 + e] = sp[(s * 2 + e + RT_WIN) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            sp += 16 * 64;
            consumed += 16384;
        }
    };

    for (int blk = 0; blk < a.nblocks; ++blk) {
        f32x16 acc[2];
        // ---------------- conv 1 + BN + ReLU : X -> T ----------------
        {
            f32x4 bias[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bias[i] = reinterpret_cast<const f32x4*>(bp)[i];
            conv3x3(X, acc);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the asm pack reads them
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                uint32_t o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = pack_relu_h2(acc[ct][2 * i], bias[i >> 1][(2 * i) & 3], acc[ct][2 * i + 1], bias[i >> 1][(2 * i + 1) & 3]);
                uint4* dst = reinterpret_cast<uint4*>(T + (ct * 32 + l31) * ROW + wv * 32 + lh * 16);
                dst[0] = uint4{o[0], o[1], o[2], o[3]};
                dst[1] = uint4{o[4], o[5], o[6], o[7]};
            }
        }
        __syncthreads();
        // ---------------- conv 2 + BN, activation and shortcut : T (+ X) -> X ----------------
        {
            f32x4 bias[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bias[i] = reinterpret_cast<const f32x4*>(bp + 32)[i];
            bp += 64;
            conv3x3(T, acc);
            // rows 8*g4 + 4*lh + 0..3 of my cout tile = accumulator elements 4*g4 + 0..3: four consecutive channels, one 8-byte
            // read-modify-write of the stream tile each.  Only this wave touches these couts, and nobody reads X during conv 2.
            uint2 rv[2][4];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    rv[ct][g4] = *reinterpret_cast<const uint2*>(X + (ct * 32 + l31) * ROW + wv * 32 + g4 * 8 + lh * 4);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const half_t* rh = reinterpret_cast<const half_t*>(&rv[ct][g4]);
                    half_t oh[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = acc[ct][g4 * 4 + j] + bias[g4][j];
                        if (!a.relu_after_add) t = fmaxf(t, 0.f);          // classical: the body ends with the activation
                        t += float(rh[j]);
                        if (a.relu_after_add) t = fmaxf(t, 0.f);           // A0: final_act(x + out)
                        oh[j] = half_t(t);
                    }
                    *reinterpret_cast<uint2*>(X + (ct * 32 + l31) * ROW + wv * 32 + g4 * 8 + lh * 4) = *reinterpret_cast<const uint2*>(oh);
                }
        }
        __syncthreads();
    }
    pf_sink ^= pf_old;
    asm volatile("" ::"v"(pf_sink));

    // ---- residual stream -> HBM ----
    {
        half_t* yb = reinterpret_cast<half_t*>(a.y) + size_t(b) * 64 * RT_C;
        for (int i = tid; i < 64 * 32; i += 512) {
            const int r = i >> 5, v = i & 31;
            *reinterpret_cast<uint4*>(yb + size_t(r) * RT_C + v * 8) = *reinterpret_cast<const uint4*>(X + r * ROW + v * 8);
        }
    }
}

void init_restower_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&restower_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BYTES);
}

void launch_restower(const ResTowerArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(restower_kernel, dim3(a.batch), dim3(512), RT_LDS_BYTES, s, a);
}

}  // namespace cra
