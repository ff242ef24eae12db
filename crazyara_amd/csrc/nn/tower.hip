// Residual-tower kernel for gfx950: a run of consecutive 3x3 mobile-bottleneck blocks in ONE launch.
//
// Reference semantics: _BottlekneckResidualBlock / _ChannelAttentionModule / _EfficientChannelAttentionModule
// (DeepCrazyhouse/src/domain/neural_net/architectures/pytorch/builder_util.py:437-475, 83-114, 49-80).
//
// One workgroup = one board for the whole run of blocks: the 64 x 256 residual stream stays in LDS (f16) from the first
// block to the last, so between blocks there is no HBM round trip, no launch boundary and no separate SE-gate launch.
// Inside a block the C_op-wide intermediate is produced and consumed in chunks of 128 channels and never leaves the CU.
//
// 8 waves = 4 MATRIX waves (0-3) + 4 VECTOR waves (4-7); wave w and w+4 share a SIMD, so every SIMD always has one wave
// feeding the matrix pipe and one wave on the vector ALU.  Per interval (one workgroup barrier each):
//   matrix wave w : E(chunk k+1)  expand 1x1, MFMA 16x16x32, 32 channels x 64 squares, +BN1 bias, ReLU -> f16 tile t1 (LDS)
//                   P(chunk k-1)  project 1x1 into its persistent 64 couts x 64 squares register accumulator, B = tile t2 (LDS)
//   vector wave w : D(chunk k)    depthwise 3x3 of ITS 32 channels on values held in the MFMA D layout (square = lane & 15 of
//                                 tile t, 4 channels per lane group): vertical neighbours are the other tile registers / a
//                                 row_ror:8 DPP move, horizontal neighbours row_shr:1 / row_shl:1; +BN2 bias, ReLU -> t2
// The weights of the whole tower are pre-packed on the host into per-wave STREAMS in exact consumption order (MFMA A
// fragments for the matrix waves, 48-byte per-channel depthwise records for the vector waves), so every wave keeps a
// fixed-depth window of loads in flight with addresses that are just "stream position + lane" -- across chunk, block and
// SE boundaries alike.
#include "kernels.h"
#include "device_utils.h"

namespace cra {

namespace {
constexpr int TW_C = 256;
constexpr int TW_XROW = TW_C + 16;                   // halves; 544-byte pitch: 16-row fragment reads hit 16 distinct 16-B slots
constexpr int TW_CK = 128;                           // C_op channels per chunk
constexpr int TW_TROW = TW_CK + 16;                  // 288-byte pitch, same property
constexpr int TW_XS_BYTES = 64 * TW_XROW * 2;        // 34816  residual stream tile
constexpr int TW_T_BYTES = 64 * TW_TROW * 2;         // 18432  one chunk tile (t1: expand output, t2: depthwise output), x2 each
constexpr int TW_T1_OFF = TW_XS_BYTES;
constexpr int TW_T2_OFF = TW_T1_OFF + 2 * TW_T_BYTES;
constexpr int TW_POOL_OFF = TW_T2_OFF + 2 * TW_T_BYTES;        // float [256] channel sums of the new stream
constexpr int TW_SE_OFF = TW_POOL_OFF + 256 * 4;              // float mean[256], part[1024], h[128], gate[256]
constexpr int TW_B3_OFF = TW_SE_OFF + (256 + 1024 + 128 + 256) * 4;   // float [256] BN3 bias of the current block
constexpr int TW_PRM_OFF = TW_B3_OFF + 256 * 4;                        // 4 vector waves x 384 floats: depthwise records of a chunk
constexpr int TW_LDS_BYTES = TW_PRM_OFF + 4 * 384 * 4;
constexpr int TW_AHEAD = 96;                          // L2 warm-up distance in fragments per stream (3 full intervals, 384 KiB)
constexpr int TW_WIN = kTowerWindow;                  // weight fragments in flight per matrix wave (16 KiB)

typedef half_t half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    half2_t h = {half_t(a), half_t(b)};              // round-to-nearest-even, as every other f16 store of the path
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float h_lo(uint32_t u) { return float(__builtin_bit_cast(half2_t, u)[0]); }
__device__ __forceinline__ float h_hi(uint32_t u) { return float(__builtin_bit_cast(half2_t, u)[1]); }

// every instruction class except VMEM may be scheduled across: keeps the stream loads where they are written, so that
// exactly the window (and not the compiler's idea of "as early as possible") is in flight next to the accumulators
#define TW_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x38F)
#define TW_FT(i) do { if (ft) ft[i] = __builtin_amdgcn_s_memtime(); } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// matrix role, one interval: E(next chunk: ET expand tiles per wave) then P(previous chunk: PS k-slabs of 32)
//   win   : TW_WIN weight fragments in flight; slot q holds stream fragment (position + q); consumed slots are refilled
//           with the fragment TW_WIN positions ahead.
//   The stream always carries 16 expand fragments ([k-slab s][tile e]) and 16 project fragments ([k-slab s2][cout tile j])
//   per chunk; for a 64-channel tail chunk the tile-1 / slab-2,3 fragments are unused padding, so the refill pattern (and
//   with it every register index) is the same for every chunk and the tail only skips MFMAs under wave-uniform branches.
__device__ __forceinline__ void matrix_interval(int et, int ps, f32x4 (&accP)[4][4], half8 (&win)[TW_WIN], const half8* __restrict__& sp,
                                                const float* __restrict__& bp, const half_t* xsr, half_t* t1w, const half_t* t2r,
                                                unsigned long long* ft) {
    using frag = half8;
    constexpr int XROW = TW_XROW, TROW = TW_TROW;
    TW_FT(0);
    // B fragments are double buffered by hand: step s issues the LDS reads of step s+1 before its own MFMAs, and nothing is
    // scheduled across a step boundary, so the reads of the next step always have a full step of matrix work to land in.
    f32x4 accE[2][4];
    f32x4 bias0, bias1;
    if (et != 0) {
        bias0 = *reinterpret_cast<const f32x4*>(bp);           // BN1 bias of my channels lg*4 + r, tile 0 / tile 1
        bias1 = *reinterpret_cast<const f32x4*>(bp + 4);
        bp += 32;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) accE[e][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        frag bfa[4], bfb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bfa[t] = *reinterpret_cast<const frag*>(xsr + t * 16 * XROW);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            frag (&cur)[4] = (s & 1) ? bfb : bfa;
            frag (&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) nxt[t] = *reinterpret_cast<const frag*>(xsr + t * 16 * XROW + (s + 1) * 32);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) mma_k32(win[s * 2], cur[t], accE[0][t]);
            if (et == 2) {
#pragma unroll
                for (int t = 0; t < 4; ++t) mma_k32(win[s * 2 + 1], cur[t], accE[1][t]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) win[s * 2 + e] = sp[(s * 2 + e + TW_WIN) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        sp += 16 * 64;
    }
    TW_FT(1);
    // expand epilogue for square tiles [t0, t1): BN1 bias + ReLU -> f16 tile in the layout the project GEMM reads as its B operand:
    // expand tile v = w*2+e (full chunk) / w (tail) sits at K positions (v/2)*32 + lg*8 + (v%2)*4 + r, i.e. one 16 (8)-byte
    // store per lane and row.  It is issued inside the first project steps so that its VALU work hides behind their MFMAs.
    auto expand_epilogue = [&](int t0, int t1) {
#pragma unroll
        for (int t = t0; t < t1; ++t) {
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 bs = e == 0 ? bias0 : bias1;
                o[e * 2 + 0] = pack_h2(fmaxf(accE[e][t][0] + bs[0], 0.f), fmaxf(accE[e][t][1] + bs[1], 0.f));
                o[e * 2 + 1] = pack_h2(fmaxf(accE[e][t][2] + bs[2], 0.f), fmaxf(accE[e][t][3] + bs[3], 0.f));
            }
            if (et == 2) *reinterpret_cast<uint4*>(t1w + t * 16 * TROW) = uint4{o[0], o[1], o[2], o[3]};
            else *reinterpret_cast<uint2*>(t1w + t * 16 * TROW) = uint2{o[0], o[1]};
        }
    };
    if (ps != 0) {
        frag bfa[4], bfb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bfa[t] = *reinterpret_cast<const frag*>(t2r + t * 16 * TROW);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            frag (&cur)[4] = (s2 & 1) ? bfb : bfa;
            frag (&nxt)[4] = (s2 & 1) ? bfa : bfb;
            if (s2 + 1 < 4) {                        // (slabs 2, 3 of a tail chunk hold stale-but-finite values: read, not used)
#pragma unroll
                for (int t = 0; t < 4; ++t) nxt[t] = *reinterpret_cast<const frag*>(t2r + t * 16 * TROW + (s2 + 1) * 32);
            }
            if (s2 < 2 || ps == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) mma_k32(win[s2 * 4 + j], cur[t], accP[j][t]);
                }
            }
            if (et != 0 && s2 < 2) expand_epilogue(s2 * 2, s2 * 2 + 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) win[s2 * 4 + j] = sp[(s2 * 4 + j + TW_WIN) * 64];
            __builtin_amdgcn_sched_barrier(0);
            if (s2 == 1) TW_FT(2);
        }
        sp += 16 * 64;
    } else if (et != 0) {
        expand_epilogue(0, 4);
    }
    TW_FT(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// vector role, one interval: D(chunk) for this wave's TILES x 16 channels.
//   The chunk's 8 per-channel records (12 floats = 9 folded taps, BN2 bias, 0, 0; stream order [step = e*4 + r][lg]) are
//   fetched one chunk ahead with three coalesced 8-byte loads per lane, parked in a wave-private LDS slice and read back
//   as broadcast 16-byte reads (a direct per-lane-group load would cost the CU's vector-memory pipe as much as the
//   matrix waves' weight stream although it moves 20x fewer bytes).
// ---------------------------------------------------------------------------------------------------------------------
struct DwParams {
    f32x4 p0, p1, p2;
};

// one channel, all 64 squares (4 tiles of this lane): out[t] = relu(b2 + sum of 9 taps).  up/down neighbours of tile t are
// the cross rows s[t] / s[t+1]; left/right neighbours come in through the DPP operand of v_fmac_f32 (row_shr:1 / row_shl:1,
// lanes outside the 16-lane row read 0), with the file-a / file-h wrap of the two board rows masked out of the weights.
// Tap-major order: four independent accumulation chains.
__device__ __forceinline__ void dw_channel(const float (&ev)[4], const float (&s)[5], const DwParams& q, float mL, float mR, float (&o)[4]) {
    const float w0 = q.p0[0] * mL, w1 = q.p0[1], w2 = q.p0[2] * mR;
    const float w3 = q.p0[3] * mL, w4 = q.p1[0], w5 = q.p1[1] * mR;
    const float w6 = q.p1[2] * mL, w7 = q.p1[3], w8 = q.p2[0] * mR;
    const float b2 = q.p2[1];
    float a0 = fmaf(w4, ev[0], b2), a1 = fmaf(w4, ev[1], b2), a2 = fmaf(w4, ev[2], b2), a3 = fmaf(w4, ev[3], b2);
    // operands: %0-3 acc, %4-8 s0..s4, %9-12 ev0..ev3, %13.. w0 w1 w2 w3 w5 w6 w7 w8
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %4, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %5, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %6, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %7, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_e32 %0, %4, %14\n\t"
        "v_fmac_f32_e32 %1, %5, %14\n\t"
        "v_fmac_f32_e32 %2, %6, %14\n\t"
        "v_fmac_f32_e32 %3, %7, %14\n\t"
        "v_fmac_f32_dpp %0, %4, %15 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %5, %15 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %6, %15 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %7, %15 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %9, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %10, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %11, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %12, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %9, %17 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %10, %17 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %11, %17 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %12, %17 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %5, %18 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %6, %18 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %7, %18 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %8, %18 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_e32 %0, %5, %19\n\t"
        "v_fmac_f32_e32 %1, %6, %19\n\t"
        "v_fmac_f32_e32 %2, %7, %19\n\t"
        "v_fmac_f32_e32 %3, %8, %19\n\t"
        "v_fmac_f32_dpp %0, %5, %20 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %6, %20 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %7, %20 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %8, %20 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(ev[0]), "v"(ev[1]), "v"(ev[2]), "v"(ev[3]),
          "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w5), "v"(w6), "v"(w7), "v"(w8));
    o[0] = fmaxf(a0, 0.f); o[1] = fmaxf(a1, 0.f); o[2] = fmaxf(a2, 0.f); o[3] = fmaxf(a3, 0.f);
}

// cross rows of one channel: s[t] is the board row above tile t's squares for lanes l15 < 8 and the row below tile t-1's
// squares for lanes l15 >= 8 -- both are "lane (l15 + 8) % 16 of tile t-1 / t", one row_ror:8 move per tile
__device__ __forceinline__ void cross_rows(const float (&ev)[4], bool hi, float (&s)[5]) {
    float rot[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rot[t] = dpp_mov<DPP_ROW_ROR8>(ev[t]);
    s[0] = hi ? rot[0] : 0.f;
#pragma unroll
    for (int t = 1; t < 4; ++t) s[t] = hi ? rot[t] : rot[t - 1];
    s[4] = hi ? 0.f : rot[3];
}

template <int TILES>
__device__ __forceinline__ void vector_interval(float2 (&pre)[3], const float2* __restrict__& pp, float* prml, int lane, int lg,
                                                const half_t* t1r, half_t* t2w, bool hi, float mL, float mR, unsigned long long* ft) {
    constexpr int TROW = TW_TROW;
    TW_FT(0);
    // park this chunk's records, request the next chunk's
#pragma unroll
    for (int k = 0; k < 3; ++k) reinterpret_cast<float2*>(prml)[lane * 3 + k] = pre[k];
    pp += 192;
#pragma unroll
    for (int k = 0; k < 3; ++k) pre[k] = pp[k];
    uint32_t in[4][2 * TILES];                       // [square tile][packed channel pair e*2 + r/2]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if constexpr (TILES == 2) {
            const uint4 u = *reinterpret_cast<const uint4*>(t1r + t * 16 * TROW);
            in[t][0] = u.x; in[t][1] = u.y; in[t][2] = u.z; in[t][3] = u.w;
        } else {
            const uint2 u = *reinterpret_cast<const uint2*>(t1r + t * 16 * TROW);
            in[t][0] = u.x; in[t][1] = u.y;
        }
    }
    auto read_pair = [&](int step, DwParams (&q)[2]) {
        const f32x4* rp = reinterpret_cast<const f32x4*>(prml + (step * 4 + lg) * 12);
        q[0].p0 = rp[0]; q[0].p1 = rp[1]; q[0].p2 = rp[2];
        q[1].p0 = rp[12]; q[1].p1 = rp[13]; q[1].p2 = rp[14];       // step + 1: 4 lane groups x 3 float4 further
    };
    DwParams qa[2], qb[2];
    read_pair(0, qa);
    TW_FT(1);
    uint32_t outp[4][2 * TILES];
#pragma unroll
    for (int step = 0; step < 4 * TILES; step += 2) {            // channel pairs (e, r = 0/1 or 2/3)
        DwParams (&cur)[2] = (step & 2) ? qb : qa;
        DwParams (&nxt)[2] = (step & 2) ? qa : qb;
        if (step + 2 < 4 * TILES) read_pair(step + 2, nxt);
        float ev0[4], ev1[4], s0[5], s1[5], o0[4], o1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ev0[t] = h_lo(in[t][step >> 1]);
            ev1[t] = h_hi(in[t][step >> 1]);
        }
        cross_rows(ev0, hi, s0);
        cross_rows(ev1, hi, s1);
        __builtin_amdgcn_sched_barrier(0);           // every DPP source is written before the hand-scheduled blocks start
        dw_channel(ev0, s0, cur[0], mL, mR, o0);
        dw_channel(ev1, s1, cur[1], mL, mR, o1);
#pragma unroll
        for (int t = 0; t < 4; ++t) outp[t][step >> 1] = pack_h2(o0[t], o1[t]);
        __builtin_amdgcn_sched_barrier(0);
        TW_FT(2 + (step >> 1));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if constexpr (TILES == 2) *reinterpret_cast<uint4*>(t2w + t * 16 * TROW) = uint4{outp[t][0], outp[t][1], outp[t][2], outp[t][3]};
        else *reinterpret_cast<uint2*>(t2w + t * 16 * TROW) = uint2{outp[t][0], outp[t][1]};
    }
}

// SE gate of a block from the channel sums the previous epilogue left in LDS; executed by all 512 threads
__device__ __forceinline__ void se_phase(const TowerBlockDesc& d, int tid, half_t* xs, const float* pool_sum, float* se_mean,
                                     float* se_part, float* se_h, float* se_gate) {
    constexpr int XROW = TW_XROW;
    {
        if (tid < 256) se_mean[tid] = pool_sum[tid] * (1.f / 64.f);
        __syncthreads();
        if (d.se_kind == 1) {        // ca_se: relu(W1 mean) -> W2 -> hard-sigmoid (builder_util.py:83-114)
            {
                const half2_t* w1 = reinterpret_cast<const half2_t*>(d.se_w1);     // [c][128] halves
                const int j2 = tid & 63, kq = tid >> 6;                           // outputs 2*j2, 2*j2+1; c in [kq*32, +32)
                half2_t wv[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) wv[k] = w1[(kq * 32 + k) * 64 + j2];
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float m = se_mean[kq * 32 + k];
                    s0 = fmaf(float(wv[k][0]), m, s0);
                    s1 = fmaf(float(wv[k][1]), m, s1);
                }
                se_part[kq * 128 + 2 * j2] = s0;
                se_part[kq * 128 + 2 * j2 + 1] = s1;
            }
            __syncthreads();
            if (tid < 128) {
                float s = 0.f;
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) s += se_part[kq * 128 + tid];
                se_h[tid] = fmaxf(s, 0.f);
            }
            __syncthreads();
            {
                const half2_t* w2 = reinterpret_cast<const half2_t*>(d.se_w2);     // [j][256] halves
                const int c2 = tid & 127, kq = tid >> 7;                          // outputs 2*c2, 2*c2+1; j in [kq*32, +32)
                half2_t wv[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) wv[k] = w2[(kq * 32 + k) * 128 + c2];
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float h = se_h[kq * 32 + k];
                    s0 = fmaf(float(wv[k][0]), h, s0);
                    s1 = fmaf(float(wv[k][1]), h, s1);
                }
                se_part[kq * 256 + 2 * c2] = s0;
                se_part[kq * 256 + 2 * c2 + 1] = s1;
            }
            __syncthreads();
            if (tid < 256) se_gate[tid] = hard_sigmoid(se_part[tid] + se_part[256 + tid] + se_part[512 + tid] + se_part[768 + tid]);
        } else {                     // eca_se: centre-tap linear + bias -> hard-sigmoid (builder_util.py:49-80)
            const half2_t* wc = reinterpret_cast<const half2_t*>(d.se_w1);         // [i][256] halves
            const int c2 = tid & 127, kq = tid >> 7;                              // i in [kq*64, +64)
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                half2_t wv[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) wv[k] = wc[(kq * 64 + h * 32 + k) * 128 + c2];
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float m = se_mean[kq * 64 + h * 32 + k];
                    s0 = fmaf(float(wv[k][0]), m, s0);
                    s1 = fmaf(float(wv[k][1]), m, s1);
                }
            }
            se_part[kq * 256 + 2 * c2] = s0;
            se_part[kq * 256 + 2 * c2 + 1] = s1;
            __syncthreads();
            if (tid < 256)
                se_gate[tid] = hard_sigmoid(d.se_b[tid] + se_part[tid] + se_part[256 + tid] + se_part[512 + tid] + se_part[768 + tid]);
        }
        __syncthreads();
        for (int i = tid; i < 64 * 32; i += 512) {     // x := x * gate (the residual uses the gated x, builder_util.py:473-475)
            const int r = i >> 5, v = i & 31;
            float xv[8], gv[8];
            load8<half_t>(xs + r * XROW + v * 8, xv);
            load8<float>(se_gate + v * 8, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] *= gv[j];
            store8<half_t>(xs + r * XROW + v * 8, xv);
        }
        __syncthreads();
    }

}
}  // namespace

size_t tower_lds_bytes() { return TW_LDS_BYTES; }

__global__ __launch_bounds__(512) void tower_kernel(const TowerArgs a) {
    using frag = half8;
    constexpr int C = TW_C, XROW = TW_XROW, TROW = TW_TROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* xs = reinterpret_cast<half_t*>(smem);
    float* pool_sum = reinterpret_cast<float*>(smem + TW_POOL_OFF);
    float* se_mean = reinterpret_cast<float*>(smem + TW_SE_OFF);
    float* se_part = se_mean + 256;
    float* se_h = se_part + 1024;
    float* se_gate = se_h + 128;
    float* b3s = reinterpret_cast<float*>(smem + TW_B3_OFF);

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_matrix = wave < 4;
    const int w = wave & 3;                              // both roles: my 32-channel slice of a chunk / my 64 couts
    const bool hi = l15 >= 8;                            // second board row of a 16-square tile
    const float mL = (l15 & 7) != 0 ? 1.f : 0.f;         // a left / right neighbour exists on the board
    const float mR = (l15 & 7) != 7 ? 1.f : 0.f;

    // ---- residual stream tile -> LDS (optionally gated: the first block's SE gate was computed by a previous launch) ----
    auto load_board = [&]() {
        const half_t* xb = reinterpret_cast<const half_t*>(a.x) + size_t(b) * 64 * C;
        if (a.gate_in == nullptr) {
            for (int i = tid; i < 64 * 32; i += 512) {
                const int r = i >> 5, v = i & 31;
                *reinterpret_cast<uint4*>(xs + r * XROW + v * 8) = *reinterpret_cast<const uint4*>(xb + size_t(r) * C + v * 8);
            }
        } else {
            const float* gt = a.gate_in + size_t(b) * C;
            for (int i = tid; i < 64 * 32; i += 512) {
                const int r = i >> 5, v = i & 31;
                float xv[8], gv[8];
                load8<half_t>(xb + size_t(r) * C + v * 8, xv);
                load8<float>(gt + v * 8, gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] *= gv[j];
                store8<half_t>(xs + r * XROW + v * 8, xv);
            }
        }
    };

    // per-lane LDS addresses: B-operand fragment rows (row l15 of a 16-square tile, k offset lg*8) and my chunk-tile slice
    const half_t* xsr = xs + l15 * XROW + lg * 8;
    const int toff_full = l15 * TROW + w * 32 + lg * 8;                       // expand tiles w*2, w*2+1
    const int toff_tail = l15 * TROW + (w >> 1) * 32 + lg * 8 + (w & 1) * 4;  // expand tile w of a 64-channel tail chunk
    const int toff_b = l15 * TROW + lg * 8;
    half_t* t1 = reinterpret_cast<half_t*>(smem + TW_T1_OFF);
    half_t* t2 = reinterpret_cast<half_t*>(smem + TW_T2_OFF);
    constexpr int TSTRIDE = TW_T_BYTES / 2;                                    // halves between the two buffers of a tile

    unsigned long long* trc = (a.trace != nullptr && b == 0 && (wave == 0 || wave == 4) && lane == 0) ? a.trace + (wave >> 2) * 256 : nullptr;
    int trn = 0;
#define TW_STAMP() do { if (trc) trc[trn++] = __builtin_amdgcn_s_memtime(); } while (0)
    TW_STAMP();
    // The two roles run completely separate control flow (same barrier sequence), so that neither carries the other's
    // register state.  Per block: [SE gate, all threads] then intervals k = -1 .. n, one barrier each:
    //   matrix: E(k+1) -> t1[(k+1)&1], then P(k-1) <- t2[(k-1)&1]        vector: D(k): t1[k&1] -> t2[k&1]
    // Only the last chunk of a block can be a 64-channel tail (1 expand tile per wave, 2 project slabs).
    if (is_matrix) {
        // open the weight stream first: its window flies while the board tile comes in
        const frag* sp = reinterpret_cast<const frag*>(a.wstream) + size_t(w) * a.wstream_wave_frags * 64 + lane;
        const float* bp = a.bstream + size_t(w) * a.bstream_wave_floats + lg * 8;
        frag win[TW_WIN];
#pragma unroll
        for (int q = 0; q < TW_WIN; ++q) win[q] = sp[q * 64];
        load_board();
        __syncthreads();
        TW_STAMP();
        for (int blk = 0; blk < a.nblocks; ++blk) {
            const TowerBlockDesc& d = a.blocks[blk];
            const int cop = d.cop_pad;
            if (blk > 0 && d.se_kind != 0) se_phase(d, tid, xs, pool_sum, se_mean, se_part, se_h, se_gate);
            TW_STAMP();
            const int n = (cop + TW_CK - 1) / TW_CK;
            const bool tail = (cop & (TW_CK - 1)) != 0;
            b3s[tid] = d.b3[tid];        // (matrix waves are threads 0..255) read back in the epilogue, many barriers later
            f32x4 accP[4][4];            // [cout tile j][square tile t]: couts (w*4+j)*16 .. +15, all 64 squares
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) accP[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k = -1; k <= n; ++k) {
                const int et = k + 1 < n ? ((tail && k + 1 == n - 1) ? 1 : 2) : 0;
                const int ps = k >= 1 ? ((tail && k - 1 == n - 1) ? 2 : 4) : 0;
                half_t* t1w = t1 + ((k + 1) & 1) * TSTRIDE + (et == 1 ? toff_tail : toff_full);
                const half_t* t2r = t2 + ((k - 1) & 1) * TSTRIDE + toff_b;
                unsigned long long* ft = (trc != nullptr && blk == a.nblocks - 1 && k >= 4 && k <= 6) ? trc + 200 + (k - 4) * 5 : nullptr;
                matrix_interval(et, ps, accP, win, sp, bp, xsr, t1w, t2r, ft);
                __syncthreads();
                if (ft) ft[4] = __builtin_amdgcn_s_memtime();
            }
            TW_STAMP();
            // ---- block epilogue: y = x + BN3(project); new residual stream back to LDS, pooled for a following SE ----
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co0 = (w * 4 + j) * 16 + lg * 4;
                float bs[4];
                load4<float>(b3s + co0, bs);
                float pool[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int sq = t * 16 + l15;
                    float rv[4], v[4];
                    load4<half_t>(xs + sq * XROW + co0, rv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = accP[j][t][r] + bs[r] + rv[r];
                        pool[r] += float(half_t(v[r]));         // pool what the next block will read
                    }
                    store4<half_t>(xs + sq * XROW + co0, v);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {        // sum over the 16 squares of the DPP row (rotations: every lane ends with the total)
                    pool[r] += dpp_mov<0x128>(pool[r]);   // row_ror:8
                    pool[r] += dpp_mov<0x124>(pool[r]);   // row_ror:4
                    pool[r] += dpp_mov<0x122>(pool[r]);   // row_ror:2
                    pool[r] += dpp_mov<0x121>(pool[r]);   // row_ror:1
                }
                if (l15 == 0) store4<float>(pool_sum + co0, pool);
            }
            __syncthreads();
            TW_STAMP();
        }
    } else {
        const float2* pp = reinterpret_cast<const float2*>(a.pstream + size_t(w) * a.pstream_wave_floats) + lane * 3;
        float* prml = reinterpret_cast<float*>(smem + TW_PRM_OFF) + w * 384;
        float2 pre[3];                   // my 24 bytes of the NEXT chunk's 1536-byte record block
#pragma unroll
        for (int k = 0; k < 3; ++k) pre[k] = pp[k];
        // L2 warm-up for the matrix waves' weight streams.  All 256 workgroups consume the same 4 streams in near lockstep, so
        // without help every line is a miss-in-flight for everyone (one XCD-L2 fill, 31 requests queued behind it) and the
        // stream runs at miss latency.  The workgroups of an XCD therefore share the job of touching each line TW_AHEAD
        // fragments early: workgroup b (XCD b % 8 under round-robin dispatch -- an assumption about speed only) takes the
        // lines whose index is (b / 8) % 32 modulo 32, one 32-lane load per interval from vector wave 0.
        const char* wsb = reinterpret_cast<const char*>(a.wstream);
        const long long wsF = a.wstream_wave_frags;
        const int pf_slot = (b >> 3) & 31;
        long long mpos = 0;              // fragments the matrix waves have consumed (same arithmetic as theirs)
        int pf_old = 0, pf_sink = 0;
        auto prefetch = [&](long long first, int nfr) -> int {   // fragments [first, first + nfr) of all 4 streams, nfr = 16 or 32
            int v = 0;
            if (w == 0 && lane < 32) {
                const int idx = lane * 32 + pf_slot, sh = nfr == 32 ? 8 : 7;
                const int wq = idx >> sh, rem = idx & ((1 << sh) - 1);
                const long long fr = first + (rem >> 3);
                if (wq < 4 && fr < wsF) v = *reinterpret_cast<const int*>(wsb + ((wq * wsF + fr) << 10) + ((rem & 7) << 7));
            }
            return v;
        };
        for (int f0 = 0; f0 < TW_AHEAD; f0 += 32) pf_sink ^= prefetch(f0, 32);
        load_board();
        __syncthreads();
        TW_STAMP();
        for (int blk = 0; blk < a.nblocks; ++blk) {
            const TowerBlockDesc& d = a.blocks[blk];
            const int cop = d.cop_pad;
            if (blk > 0 && d.se_kind != 0) se_phase(d, tid, xs, pool_sum, se_mean, se_part, se_h, se_gate);
            TW_STAMP();
            const int n = (cop + TW_CK - 1) / TW_CK;
            const bool tail = (cop & (TW_CK - 1)) != 0;
            for (int k = -1; k <= n; ++k) {
                {
                    const int adv = (k + 1 < n ? 16 : 0) + (k >= 1 ? 16 : 0);
                    pf_sink ^= pf_old;                         // last interval's load is consumed a whole interval later
                    pf_old = adv != 0 ? prefetch(mpos + TW_AHEAD, adv) : 0;
                    mpos += adv;
                }
                if (k >= 0 && k < n) {
                    const bool tl = tail && k == n - 1;
                    const int off = (k & 1) * TSTRIDE + (tl ? toff_tail : toff_full);
                    unsigned long long* ft = (trc != nullptr && blk == a.nblocks - 1 && k == 5) ? trc + 220 : nullptr;
                    if (tl) vector_interval<1>(pre, pp, prml, lane, lg, t1 + off, t2 + off, hi, mL, mR, ft);
                    else vector_interval<2>(pre, pp, prml, lane, lg, t1 + off, t2 + off, hi, mL, mR, ft);
                    if (ft) ft[6] = __builtin_amdgcn_s_memtime();
                }
                __syncthreads();
            }
            TW_STAMP();
            __syncthreads();             // the matrix waves' block epilogue
            TW_STAMP();
        }
        pf_sink ^= pf_old;
        asm volatile("" ::"v"(pf_sink));     // keeps the warm-up loads alive; their values are never used
    }

    // ---- residual stream -> HBM; channel sums for an SE gate computed by a later launch ----
    {
        half_t* yb = reinterpret_cast<half_t*>(a.y) + size_t(b) * 64 * C;
        for (int i = tid; i < 64 * 32; i += 512) {
            const int r = i >> 5, v = i & 31;
            *reinterpret_cast<uint4*>(yb + size_t(r) * C + v * 8) = *reinterpret_cast<const uint4*>(xs + r * XROW + v * 8);
        }
        if (a.pool_out != nullptr && tid < 256) a.pool_out[size_t(b) * C + tid] = pool_sum[tid];
    }
}

void init_tower_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES);
}

void launch_tower(const TowerArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(tower_kernel, dim3(a.batch), dim3(512), TW_LDS_BYTES, s, a);
}

}  // namespace cra
