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
constexpr int RT_WIN = 16;
template <int NB> constexpr int rt_lds_bytes() { return 2 * NB * RT_TILE_BYTES; }

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

size_t restower_lds_bytes() { return rt_lds_bytes<2>(); }

// NB boards per workgroup, NR cout tiles (of 32) per wave; 8 / NR waves.  A wave's MFMAs form an NR x (2 NB) grid of 32 x 32 tiles
// that shares its operand fragments: per k-step NR weight fragments + 2 NB tile fragments feed 2 NB NR MFMAs, i.e.
// (NR + 2 NB) / (2 NB NR) KiB of operands per MFMA -- 1.5 (NB 1, NR 1), 1.25 (2, 1), 1.0 (1, 2), 0.75 (2, 2).  What bounds these
// kernels is the rate at which operands return into a SIMD's registers while its MFMAs run (measured 24-33 B/clk per SIMD,
// scripts/ubench/mfma_mem_issue.hip: about 1 KiB per 32-cycle MFMA), so fewer, fatter waves beat more, thinner ones.
template <int NB, int NR>
__global__ __launch_bounds__(512 / NR) void restower_kernel(const ResTowerArgs a) {
    constexpr int TH = 512 / NR;                                               // threads
    using frag = half8;
    constexpr int ROW = RT_ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* X = reinterpret_cast<half_t*>(smem);                               // NB stream tiles, then NB intermediate tiles
    half_t* T = reinterpret_cast<half_t*>(smem + NB * RT_TILE_BYTES);
    constexpr int TILE = RT_TILE_BYTES / 2;                                    // halves between the tiles of consecutive boards
    constexpr int NT = 2 * NB;                                                 // 32-square tiles (columns of the wave's grid)

    const int b = blockIdx.x * NB;                                             // first board of this workgroup
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- open my weight stream, then bring the board in ----
    const frag* sp = reinterpret_cast<const frag*>(a.wstream) + size_t(wv) * a.wstream_wave_frags * 64 + lane;
    const char* sline = reinterpret_cast<const char*>(a.wstream) + size_t(wv) * a.wstream_wave_frags * 1024;   // warm-up cursor
    const float* bp = a.bstream + size_t(wv) * a.bstream_wave_floats + lh * 16;   // per conv: [rt][lane/32][16]
    frag win[RT_WIN];
#pragma unroll
    for (int q = 0; q < RT_WIN; ++q) win[q] = sp[q * 64];
    for (int nb = 0; nb < NB; ++nb) {
        const bool live = b + nb < a.batch;          // an odd batch leaves the last workgroup's second board empty (zeros)
        const half_t* xb = reinterpret_cast<const half_t*>(a.x) + size_t(b + nb) * 64 * RT_C;
        for (int i = tid; i < 64 * 32; i += TH) {
            const int r = i >> 5, v = i & 31;
            *reinterpret_cast<uint4*>(X + nb * TILE + r * ROW + v * 8) =
                live ? *reinterpret_cast<const uint4*>(xb + size_t(r) * RT_C + v * 8) : uint4{0u, 0u, 0u, 0u};
        }
        if (tid < ROW / 2) {                          // zero rows of both tiles (ROW / 2 = 132 <= TH)
            reinterpret_cast<uint32_t*>(X + nb * TILE + 64 * ROW)[tid] = 0u;
            reinterpret_cast<uint32_t*>(T + nb * TILE + 64 * ROW)[tid] = 0u;
        }
    }
    __syncthreads();

    const int pf_slot = (blockIdx.x >> 3) & 31;
    int pf_old = 0, pf_sink = 0;
    const long long stream_bytes = a.wstream_wave_frags * 1024;
    long long consumed = 0;                          // bytes of my stream the taps so far have used

    // one 3x3 convolution of the NB tiles at `src` into acc[rt][t] (9 taps x 16 k-steps x NR*NT MFMAs); tile column t = board
    // t/2, squares 32*(t%2) .. +31.  Stream order per tap: [k-step][rt]; fragment f of the tap sits in window slot f % 16.
    auto conv3x3 = [&](const half_t* src, f32x16 (&acc)[NR][NT]) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[rt][t][v] = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            pf_sink ^= pf_old;
            {   // L2 warm-up: my share of the lines two windows (32 KiB of stream) ahead
                const long long off = consumed + 2 * 16384 + (lane * 32 + pf_slot) * 128;
                pf_old = (lane < 4 * NR && off < stream_bytes) ? *reinterpret_cast<const int*>(sline + off) : 0;
            }
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const half_t* rows[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) rows[t] = src + (t >> 1) * TILE + nbr_row((t & 1) * 32 + l31, dy, dx) * ROW + lh * 8;
            frag bfa[2 * NT], bfb[2 * NT];           // [k-step parity within the step][tile column]
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bfa[t] = *reinterpret_cast<const frag*>(rows[t]);
                bfa[NT + t] = *reinterpret_cast<const frag*>(rows[t] + 16);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {            // step = k-steps 2s, 2s+1
                frag (&cur)[2 * NT] = (s & 1) ? bfb : bfa;
                frag (&nxt)[2 * NT] = (s & 1) ? bfa : bfb;
                if (s + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        nxt[t] = *reinterpret_cast<const frag*>(rows[t] + (s + 1) * 32);
                        nxt[NT + t] = *reinterpret_cast<const frag*>(rows[t] + (s + 1) * 32 + 16);
                    }
                }
                // next step's tile fragments are ISSUED before this step's MFMAs (tower.hip, matrix_interval explains the fence)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int rt = 0; rt < NR; ++rt)
#pragma unroll
                        for (int t = 0; t < NT; ++t) mma32(win[((s * 2 + kk) * NR + rt) & 15], cur[kk * NT + t], acc[rt][t]);
#pragma unroll
                for (int e = 0; e < 2 * NR; ++e) win[(s * 2 * NR + e) & 15] = sp[(s * 2 * NR + e + RT_WIN) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            sp += 16 * NR * 64;
            consumed += 16384 * NR;
        }
    };

    for (int blk = 0; blk < a.nblocks; ++blk) {
        f32x16 acc[NR][NT];
        // ---------------- conv 1 + BN + ReLU : X -> T ----------------
        {
            f32x4 bias[NR][4];
#pragma unroll
            for (int rt = 0; rt < NR; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) bias[rt][i] = reinterpret_cast<const f32x4*>(bp + rt * 32)[i];
            conv3x3(X, acc);
#pragma unroll
            for (int rt = 0; rt < NR; ++rt)          // the last MFMAs retire before the asm pack reads them (device_utils.h)
#pragma unroll
                for (int t = 0; t < NT; ++t) mfma_retire(acc[rt][t]);
#pragma unroll
            for (int rt = 0; rt < NR; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    uint32_t o[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        o[i] = pack_relu_h2(acc[rt][t][2 * i], bias[rt][i >> 1][(2 * i) & 3], acc[rt][t][2 * i + 1], bias[rt][i >> 1][(2 * i + 1) & 3]);
                    uint4* dst = reinterpret_cast<uint4*>(T + (t >> 1) * TILE + ((t & 1) * 32 + l31) * ROW + (wv * NR + rt) * 32 + lh * 16);
                    dst[0] = uint4{o[0], o[1], o[2], o[3]};
                    dst[1] = uint4{o[4], o[5], o[6], o[7]};
                }
        }
        __syncthreads();
        // ---------------- conv 2 + BN, activation and shortcut : T (+ X) -> X ----------------
        {
            f32x4 bias[NR][4];
#pragma unroll
            for (int rt = 0; rt < NR; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) bias[rt][i] = reinterpret_cast<const f32x4*>(bp + NR * 32 + rt * 32)[i];
            bp += 2 * NR * 32;
            conv3x3(T, acc);
            // rows 8*g4 + 4*lh + 0..3 of a cout tile = accumulator elements 4*g4 + 0..3: four consecutive channels, one 8-byte
            // read-modify-write of the stream tile each.  Only this wave touches these couts, and nobody reads X during conv 2.
#pragma unroll
            for (int rt = 0; rt < NR; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    half_t* xrow = X + (t >> 1) * TILE + ((t & 1) * 32 + l31) * ROW + (wv * NR + rt) * 32 + lh * 4;
                    uint2 rv[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) rv[g4] = *reinterpret_cast<const uint2*>(xrow + g4 * 8);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const half_t* rh = reinterpret_cast<const half_t*>(&rv[g4]);
                        half_t oh[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = acc[rt][t][g4 * 4 + j] + bias[rt][g4][j];
                            if (!a.relu_after_add) v = fmaxf(v, 0.f);          // classical: the body ends with the activation
                            v += float(rh[j]);
                            if (a.relu_after_add) v = fmaxf(v, 0.f);           // A0: final_act(x + out)
                            oh[j] = half_t(v);
                        }
                        *reinterpret_cast<uint2*>(xrow + g4 * 8) = *reinterpret_cast<const uint2*>(oh);
                    }
                }
        }
        __syncthreads();
    }
    pf_sink ^= pf_old;
    asm volatile("" ::"v"(pf_sink));

    // ---- residual stream -> HBM ----
    for (int nb = 0; nb < NB; ++nb) {
        if (b + nb >= a.batch) break;
        half_t* yb = reinterpret_cast<half_t*>(a.y) + size_t(b + nb) * 64 * RT_C;
        for (int i = tid; i < 64 * 32; i += TH) {
            const int r = i >> 5, v = i & 31;
            *reinterpret_cast<uint4*>(yb + size_t(r) * RT_C + v * 8) = *reinterpret_cast<const uint4*>(X + nb * TILE + r * ROW + v * 8);
        }
    }
}

void init_restower_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&restower_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, rt_lds_bytes<1>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&restower_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, rt_lds_bytes<2>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&restower_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, rt_lds_bytes<1>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&restower_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, rt_lds_bytes<2>());
}

void launch_restower(const ResTowerArgs& a, hipStream_t s) {
    const int nb = a.boards_per_workgroup == 2 ? 2 : 1, nr = a.cout_tiles_per_wave == 2 ? 2 : 1;
    const dim3 grid((a.batch + nb - 1) / nb), block(512 / nr);
    if (nb == 1 && nr == 1) hipLaunchKernelGGL((restower_kernel<1, 1>), grid, block, rt_lds_bytes<1>(), s, a);
    else if (nb == 2 && nr == 1) hipLaunchKernelGGL((restower_kernel<2, 1>), grid, block, rt_lds_bytes<2>(), s, a);
    else if (nb == 1 && nr == 2) hipLaunchKernelGGL((restower_kernel<1, 2>), grid, block, rt_lds_bytes<1>(), s, a);
    else hipLaunchKernelGGL((restower_kernel<2, 2>), grid, block, rt_lds_bytes<2>(), s, a);
}

}  // namespace cra
