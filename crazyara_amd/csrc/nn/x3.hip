// Precision float16x3: the fast mode that meets north_star's "logits within 1e-3 of fp32" (it measures ~1e-5).
//
// Every dense contraction of the network (stem, expand / project 1x1, dense 3x3 towers, policy convs, value head conv and FCs, flat
// policy Linear) runs on the f16 matrix pipe with SPLIT operands: a = a_hi + a_lo, a_hi = rne_f16(a), a_lo = rne_f16(a - a_hi), and
//     a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (three v_mfma_f32_16x16x32_f16, f32 accumulate; a_lo*b_lo ~ 2^-22 is dropped)
// so a product carries ~22 significand bits (17+ for operands small enough that a_lo is an f16 subnormal, |a| < 2^-3: absolute error
// <= 2^-25) where Precision float16 carries 11.  The matrix pipe does 3 MFMAs per product at 16x the exact-f32 MFMA rate
// (v_mfma_f32_16x16x4_f32, Precision float32): a ceiling of 2500 / 3 = 833 TFLOP/s against 157.
//
// Layout: activations live in HBM as float [B][64][C] exactly as in Precision float32 (depthwise, SE gates, softmax and the last value
// FC are the float32 mode's own kernels); a board tile is split ONCE while it is staged into LDS (two f16 tiles, hi and lo), weights
// are split once on the host after BN folding in double (two A-fragment images, rise_net.hip: pack_dense_split).
//
// Reference semantics: the same modules as kernels.hip (builder_util.py:154-178, 437-475, 206-326).
#include "kernels.h"
#include "device_utils.h"
#include "value_head_body.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <type_traits>

// CRA_X3_ABL: development switches that TIME parts of the tower's chunk loop (scripts/ubench/x3_tower_ablate.hip); every bit computes wrong
// results on purpose, so they only compile in a development build.  1: no depthwise arithmetic, 2: no expand MFMAs, 4: no project MFMAs,
// 8: no LDS operand reads (expand and project), 16: no weight loads, 32: no chunk barriers, 64: no t2 stores, 128: the expand GEMM issues
// the mixed split's instruction mix (per 64 k two f16 MFMAs and one 8-bit 16x16x128 on whatever the registers hold); tower_p8_kernel
// honours 1, 2, 4, 16, 64 and (round 6, the weight-port question) 512: the 8-bit weight images are fetched at HALF size -- one 16-byte
// piece per lane and 64-k step, the other half a register copy -- i.e. the L2 -> CU stream of a 3-bytes-per-weight layout with its byte
// permutes stood in for by the copies; 1024: no 8-bit weight fetches at all (2 bytes per weight)
#ifndef CRA_X3_ABL
#define CRA_X3_ABL 0
#endif
// tower_p8_kernel<3>, E + D intervals: how many vector instructions the scheduling recipe places behind every MFMA.  10 measured 2.4 % faster than
// 4 over five interleaved rounds (2: 1.8 %; 3, 5, 6: 3 - 4 % SLOWER -- the recipe steers the machine scheduler, not the hardware: profiles/r04/ap-ar)
#ifndef X3_SGB_VALU
#define X3_SGB_VALU 10
#endif
#if CRA_X3_ABL != 0 && !defined(CRA_DEVELOPMENT)
#error "CRA_X3_ABL is a development switch (wrong results): build with -DCRA_DEVELOPMENT"
#endif
// CRA_X3_TRACE=<block>: development, the two-role tower stamps the shader clock at its phase boundaries while it runs block <block>
// (workgroups 0 and 131, every wave; scripts/ubench/x3_tower_ablate.hip prints the timeline)
#if defined(CRA_X3_TRACE) && !defined(CRA_DEVELOPMENT)
#error "CRA_X3_TRACE is a development switch: build with -DCRA_DEVELOPMENT"
#endif

namespace cra {

#ifdef CRA_X3_TRACE
__device__ unsigned long long x3_trace[2][8][128][2];      // [workgroup 0 | 131][wave][stamp](clock, interval * 16 + phase id)
#define X3_STAMP(id)                                                                                    \
    do {                                                                                                \
        if (tracing && trace_n < 128) {                                                                 \
            const unsigned long long t_ = __builtin_readcyclecounter();                                 \
            if (lane == 0) {                                                                            \
                x3_trace[b != 0][wave][trace_n][0] = t_;                                                \
                x3_trace[b != 0][wave][trace_n][1] = (unsigned long long)((kk + 1) * 16 + (id));        \
            }                                                                                           \
            ++trace_n;                                                                                  \
        }                                                                                               \
    } while (0)
#define X3_STAMP_SLAB(i_, sl_) do { const int kk = (i_) - 1; X3_STAMP(6); (void)(sl_); } while (0)      // (a k-slab of an EXPAND interval begins)
#else
#define X3_STAMP(id) do { } while (0)
#define X3_STAMP_SLAB(i_, sl_) do { } while (0)
#endif

namespace {
constexpr int X3_ABL = CRA_X3_ABL;
__device__ __forceinline__ void x3_mfma(const half8& a, const half8& b, f32x4& c, bool on) {
    if (on) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    else asm volatile("" : "+v"(c) : "v"(a), "v"(b));
}

// (a, b) -> packed f16 pairs hi = rne(a | b), lo = rne((a | b) - hi): 4 instructions (pack-convert, two mix-precision FMAs that read the
// f16 halves in place, pack-convert) where the compiler's form of the same arithmetic takes about ten.  The difference a - hi is exact.
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
        "v_fma_mix_f32 %1, %0, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(ra), "=&v"(rb)
        : "v"(a), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
}
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(half8, u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(half8, u32x4{l[0], l[1], l[2], l[3]});
}
__device__ __forceinline__ void split4(const float (&v)[4], half4& hi, half4& lo) {
    uint32_t h[2], l[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) split_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(half4, u32x2{h[0], h[1]});
    lo = __builtin_bit_cast(half4, u32x2{l[0], l[1]});
}

// Precision float16p8 (tower_p8_kernel): a value v goes to the matrix unit as the f16 pair of the float16x3 split, hi = rne(v) for the MAIN
// product, and as two e5m2 bytes for the CROSS products: hi8 / lo8 = the HIGH BYTES of hi and of lo = rne_f16(v - hi).  e5m2 has f16's
// exponent field, so the byte IS the value truncated to two mantissa bits: one v_perm_b32 per four values and image, no conversion instruction
// (v_cvt_scalef32_pk_fp8_* issue at a fraction of the VALU rate: profiles/NOTES.md, round 4); the mean loss of the truncation is taken back on
// the host-made weight images (rise_net.hip: pack_dense_p8).  4 values -> 2 dwords of f16, 1 dword of hi8, 1 dword of lo8.
typedef int i32x4_x3 __attribute__((ext_vector_type(4)));
typedef int i32x8_x3 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t x3_high_bytes(uint32_t p01, uint32_t p23) {      // the high bytes of four packed f16, in order
    return __builtin_amdgcn_perm(p23, p01, 0x07050301u);
}
__device__ __forceinline__ void split4_b8(const float (&v)[4], half4& hi, uint32_t& h8, uint32_t& l8) {
    uint32_t h[2], l[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) split_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(half4, u32x2{h[0], h[1]});
    h8 = x3_high_bytes(h[0], h[1]);
    l8 = x3_high_bytes(l[0], l[1]);
}
// D(16x16) += A(16 x 128) * B(128 x 16), e5m2 operands (cbsz = blgp = 1): lane l holds row / column l % 16 and the 32 bytes k = (l / 16) * 32 + t of the
// step -- the same labelling on both operands, so the sum is the plain product whatever order the hardware walks its k in
__device__ __forceinline__ void x3_mfma8(const i32x8_x3& a, const i32x8_x3& b, f32x4& c, bool on) {
    if (on) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, 0, 0, 0, 0);
    else asm volatile("" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ i32x8_x3 x3_cat(const half8& lo16, const half8& hi16) {       // a lane's 32 operand bytes from two 16-byte pieces
    return __builtin_shufflevector(__builtin_bit_cast(i32x4_x3, lo16), __builtin_bit_cast(i32x4_x3, hi16), 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc += w * x[lane -/+ 1 within the 16-lane row] (lanes shifted in from outside the row contribute 0).  Through the builtin, not inline
// assembly: a DPP operand written by a VALU instruction needs two wait states in front of the DPP read, the compiler inserts them for
// its own instructions and does not look inside an asm statement (a hand-written v_fmac_f32_dpp here read stale registers: r03c).
__device__ __forceinline__ float fmac_shr1(float acc, float x, float w) { return fmaf(w, dpp_mov<DPP_ROW_SHR1>(x), acc); }
__device__ __forceinline__ float fmac_shl1(float acc, float x, float w) { return fmaf(w, dpp_mov<DPP_ROW_SHL1>(x), acc); }

// one product tile: the two cross terms first, the main term last
__device__ __forceinline__ void mma_x3(const half8& ah, const half8& al, const half8& bh, const half8& bl, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
}

constexpr int X3_KC = 128;                 // input channels staged per pass of the conv GEMM
constexpr int X3_ROWP = X3_KC + 8;         // halves per LDS row (+16 B: the 16 rows of a fragment read land in 16 bank groups)

}  // namespace

// ================================================================================================================
// Dense conv (1x1 / 3x3) as implicit GEMM -- conv_gemm_kernel<float> (kernels.hip) with split operands.
// ================================================================================================================
// The epilogue of the conv GEMMs: conv_gemm_kernel<float>'s, word for word (bias, ReLU before / after the shortcut, the four output layouts,
// the fused row softmax); acc_scale: the accumulators carry the weights' power-of-two scale (Precision float16p8)
template <int MT, int NW>
__device__ __forceinline__ void conv_x3_finish(const ConvArgs& a, f32x4 (&acc)[MT][4], const bool (&active)[MT], char* smem, int b, int co_tile0, float acc_scale) {
    constexpr int NTHR = 64 * NW;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    if (a.softmax_out) __syncthreads();                      // the board's logits gather in the staging tiles: every wave is done reading them
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (!active[m]) continue;
        const int co0 = (co_tile0 + m) * 16 + lg * 4;
        float bs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[r] = a.bias[co0 + r];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int sq = t * 16 + l15;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[m][t][r], acc_scale, bs[r]);
            if (a.relu == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (a.resid) {
                float rv[4];
                load4<float>(reinterpret_cast<const float*>(a.resid) + (size_t(b) * kSquares + sq) * a.cout_ld + co0, rv);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rv[r];
            }
            if (a.relu == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (a.out_policy_f32) {
                float* o = reinterpret_cast<float*>(a.out) + size_t(b) * a.cout_real * kSquares;
                float* lds_logits = reinterpret_cast<float*>(smem);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < a.cout_real) {
                        if (a.out) o[(co0 + r) * kSquares + sq] = v[r];
                        if (a.softmax_out) lds_logits[(co0 + r) * kSquares + sq] = v[r];
                    }
            } else if (a.out_rows_f32) {
                const int row = b * kSquares + sq;
                if (row < a.rows_valid) {
                    float* o = reinterpret_cast<float*>(a.out) + size_t(row) * a.cout_real;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co0 + r < a.cout_real) o[co0 + r] = v[r];
                }
            } else if (a.out_flat) {
                float* o = reinterpret_cast<float*>(a.out) + size_t(b) * a.flat_pitch;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < a.cout_real) o[(co0 + r) * kSquares + sq] = v[r];
            } else {
                store4<float>(reinterpret_cast<float*>(a.out) + (size_t(b) * kSquares + sq) * a.cout_ld + co0, v);
            }
        }
    }
    if (a.softmax_out) {
        // row softmax of the board's logits (softmax_kernel, kernels.hip; apply_softmax(), neuralnetapi.cpp:241-260): exp(x - (max + log(sum)))
        __syncthreads();
        const float* in = reinterpret_cast<const float*>(smem);
        float* red = reinterpret_cast<float*>(smem) + 8192;  // behind the logits (at most 8192 of them: 32 KiB of the 35 KiB)
        const int n = a.cout_real * kSquares;
        float* out = a.softmax_out + size_t(b) * n;
        float m = -INFINITY;
        for (int i = tid; i < n; i += NTHR) m = fmaxf(m, in[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) m = fmaxf(m, red[i]);
        float sum = 0.f;
        for (int i = tid; i < n; i += NTHR) sum += expf(in[i] - m);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        __syncthreads();
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        sum = red[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) sum += red[i];
        const float c = m + logf(sum);
        for (int i = tid; i < n; i += NTHR) out[i] = expf(in[i] - c);
    }
}

// Workgroup = one board x (NW waves x MT cout tiles of 16): the board's channels are staged and split ONCE per workgroup, so wide
// layers take the whole cout range in one workgroup (NW = 8, MT = 2: 256 couts -- stem, policy conv 1; with 64 couts per workgroup the
// split was redone four times per board), and a stream fragment read from LDS feeds MT x 3 MFMAs.
// NS > 0: every staged pass has exactly NS k-slabs (cin a multiple of 32 * NS, at most KC per pass) and runs a static schedule:
// weight fragments through a window of three (tap, k-slab) steps -- requested before the pass is staged, refilled right behind
// their MFMAs -- and the stream fragments of the next step read from LDS before this step's MFMAs (as in the tower).  NS = 0: any cin.
template <int KS, int MT, int NW, int NS>
__device__ __forceinline__ void conv_gemm_x3_body(const ConvArgs& a, char* smem, const int bx, const int b) {
    half_t* xh = reinterpret_cast<half_t*>(smem);            // [65][ROWP] hi
    half_t* xl = xh + 65 * X3_ROWP;                          // [65][ROWP] lo
    constexpr int ROWP = X3_ROWP, KC = X3_KC, NTHR = 64 * NW;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int co_tile0 = (bx * NW + wave) * MT;
    bool active[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) active[m] = (co_tile0 + m) * 16 < a.cout_pad;
    const float* xb = reinterpret_cast<const float*>(a.x) + size_t(b) * kSquares * a.cin;
    const int nslab_ci = a.cin >> 5;
    const int nslab = KS * KS * nslab_ci;
    const half8 *wph[MT], *wpl[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        wph[m] = reinterpret_cast<const half8*>(a.wpk) + size_t(active[m] ? co_tile0 + m : 0) * nslab * 64 + lane;
        wpl[m] = reinterpret_cast<const half8*>(a.wpk_lo) + size_t(active[m] ? co_tile0 + m : 0) * nslab * 64 + lane;
    }

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < ROWP; i += NTHR) {                 // row 64: what out-of-board taps read
        xh[64 * ROWP + i] = half_t(0.f);
        xl[64 * ROWP + i] = half_t(0.f);
    }

    for (int kc0 = 0; kc0 < a.cin; kc0 += KC) {
        const int kcl = NS > 0 ? 32 * NS : min(KC, a.cin - kc0);
        constexpr int NSTEP = KS * KS * (NS > 0 ? NS : 1), D = 3;
        half8 wh[D][MT], wl[D][MT];
        auto wload = [&](int st) {                           // step st = tap st / NS, k-slab st % NS of this pass
            const size_t wo = size_t((st / (NS > 0 ? NS : 1)) * nslab_ci + (kc0 >> 5) + st % (NS > 0 ? NS : 1)) * 64;
#pragma unroll
            for (int m = 0; m < MT; ++m) { wh[st % D][m] = wph[m][wo]; wl[st % D][m] = wpl[m][wo]; }
        };
        if constexpr (NS > 0) {
            if (!(a.dev & 2) || active[0]) {
#pragma unroll
                for (int st = 0; st < D && st < NSTEP; ++st) wload(st);
            }
        }
        __syncthreads();
        const int vec_per_row = kcl >> 3;                    // 8 floats -> 8 + 8 halves
        for (int i = tid; i < kSquares * vec_per_row; i += NTHR) {
            const int r = i / vec_per_row, v = i - r * vec_per_row;
            float f[8];
            if (a.planes) {                                  // NCHW planes: channel c of square r
                const float* pb = a.planes + size_t(b) * a.planes_c * kSquares + r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = kc0 + v * 8 + j;
                    f[j] = c < a.planes_c ? pb[c * kSquares] : 0.f;
                }
            } else if (a.dev & 16) {                         // (development, timing only: no loads of the board)
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = 0.f;
            } else {
                load8<float>(xb + size_t(r) * a.cin + kc0 + v * 8, f);
            }
            half8 h, l;
            split8(f, h, l);
            *reinterpret_cast<half8*>(xh + r * ROWP + v * 8) = h;
            *reinterpret_cast<half8*>(xl + r * ROWP + v * 8) = l;
        }
        __syncthreads();
        if constexpr (NS > 0) {
            if (active[0] && !(a.dev & 8)) {                 // (development bit 8, timing only: no K loop)
                half8 bh[2][4], bl[2][4];
                auto read_frag = [&](int st) {
                    const int tap = st / NS, sl = st % NS, dy = tap / KS - KS / 2, dx = tap % KS - KS / 2;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int sq = t * 16 + l15;
                        const int ny = (sq >> 3) + dy, nx = (sq & 7) + dx;
                        const bool ok = (unsigned(ny) < 8u) && (unsigned(nx) < 8u);
                        const int off = (ok ? ny * 8 + nx : 64) * ROWP + lg * 8 + sl * 32;
                        bh[st & 1][t] = *reinterpret_cast<const half8*>(xh + off);
                        bl[st & 1][t] = *reinterpret_cast<const half8*>(xl + off);
                    }
                };
                read_frag(0);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) read_frag(st + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[st % D][m], bh[st & 1][t], acc[m][t], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st % D][m], bl[st & 1][t], acc[m][t], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st % D][m], bh[st & 1][t], acc[m][t], 0, 0, 0);
                    if (st + D < NSTEP) wload(st + D);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else
        if (active[0]) {
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap) {
                const int dy = tap / KS - KS / 2, dx = tap % KS - KS / 2;
                int rowoff[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int sq = t * 16 + l15;
                    const int ny = (sq >> 3) + dy, nx = (sq & 7) + dx;
                    const bool ok = (unsigned(ny) < 8u) && (unsigned(nx) < 8u);
                    rowoff[t] = (ok ? ny * 8 + nx : 64) * ROWP + lg * 8;
                }
                const size_t wo = size_t(tap * nslab_ci + (kc0 >> 5)) * 64;
                const int ns = kcl >> 5;
                half8 ah[MT], al[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) { ah[m] = wph[m][wo]; al[m] = wpl[m][wo]; }
                for (int sl = 0; sl < ns; ++sl) {
                    half8 ch[MT], cl[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) { ch[m] = ah[m]; cl[m] = al[m]; }
                    if (sl + 1 < ns) {                       // next slab's fragments fly while this slab's MFMAs run
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            ah[m] = wph[m][wo + size_t(sl + 1) * 64];
                            al[m] = wpl[m][wo + size_t(sl + 1) * 64];
                        }
                    }
                    half8 bh[4], bl[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        bh[t] = *reinterpret_cast<const half8*>(xh + rowoff[t] + sl * 32);
                        bl[t] = *reinterpret_cast<const half8*>(xl + rowoff[t] + sl * 32);
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl[m], bh[t], acc[m][t], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[m], bl[t], acc[m][t], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[m], bh[t], acc[m][t], 0, 0, 0);
                }
            }
        }
    }

    conv_x3_finish<MT, NW>(a, acc, active, smem, b, co_tile0, 1.f);
    if (a.dev & 1) __syncthreads();
}
template <int KS, int MT, int NW, int NS>
__global__ __launch_bounds__(64 * NW) void conv_gemm_x3_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_gemm_x3_body<KS, MT, NW, NS>(a, smem, blockIdx.x, blockIdx.y);
}

// The heads of a small batch in ONE launch (nets made for at most 64 boards, rise_net.hip): one workgroup of board b runs the second policy conv
// with the board's softmax -- conv_gemm_x3_kernel<3, 1, 8, 4>'s work -- and another one the value head (value_head_kernel_8w's).  Behind one another the two cost a batch of one 20 + 25 us on two CUs of 256; side by side the longer of
// the two.  (For a FULL batch the same launch -- or the value head beside policy conv 1 -- gains nothing: 0.074 ms against 0.049 + 0.025,
// the chip is full either way and the dispatcher, not the kernel, decides which workgroups share a CU; profiles/NOTES.md round 6.)  (On two streams instead: slower than in sequence, the joins across queues cost more than they hide -- profiles/r06/e_*.)
__global__ __launch_bounds__(512) void heads_small_kernel(const HeadsSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Workgroups go to the eight XCDs round-robin by their linear id: with (role, board) = (id % 2, id / 2) every conv role would sit on the
    // even XCDs, two to a CU, and every value head on the odd ones.  id = xcd + 8 slot: role = slot % 2, board = xcd + 8 (slot / 2) --
    // both roles of a board on one XCD (they read the same tile), each XCD half and half.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = xcd + 8 * (slot >> 1);
    if (b >= a.conv.batch) return;
    if ((slot & 1) == 0) {
        conv_gemm_x3_body<3, 1, 8, 4>(a.conv, smem, 0, b);
    } else {
        value_head_body<float, false, true, 512>(a.vh, smem, b);
    }
}

// Precision float16p8, dense 3x3 conv with cin a multiple of 128 (the two policy convs): conv_gemm_x3_kernel<3, MT, NW, 4> with the cross terms on
// e5m2 MFMAs -- per (tap, 64 k) two f16 MFMAs hi x hi and one v_mfma_f32_16x16x128_f8f6f4 on [hi8 | lo8] x [w_lo8 ; w_hi8] per cout and
// square tile instead of six f16 MFMAs (tower_p8_kernel's arithmetic; weights: rise_net.hip pack_dense_p8, the accumulators carry 2^p,
// a.acc_scale = 2^-p).  The board is staged as the f16 hi tile + a byte tile [hi8, 128 B | lo8, 128 B] per square (the high bytes of the
// split's f16 pair), double-buffered: the next pass's 32 KB are requested from HBM before this pass's MFMAs and split behind them.
namespace {
struct ConvP8 {                                                  // geometry of the staged tiles and of a pass over one of them
    static constexpr int ROWP = X3_ROWP, KC = X3_KC, R8 = 272, R8LO = 144, NS = KC / 32, NSTEP = 9 * NS, D = 3;
    static_assert(ROWP * 2 == R8 && KC == 128, "the byte tile has the f16 tile's row pitch");
    static constexpr size_t TILE = size_t(65) * ROWP * sizeof(half_t);
    static constexpr size_t lds_bytes = 4 * TILE;               // two (f16 tile + byte tile) buffers: 70 KB
    static __device__ __forceinline__ half_t* xh_of(char* smem, int buf) { return reinterpret_cast<half_t*>(smem + size_t(buf) * 2 * TILE); }
    static __device__ __forceinline__ char* x8_of(char* smem, int buf) { return smem + size_t(buf) * 2 * TILE + TILE; }
};
template <int MT> struct ConvP8Window {                          // weight fragments in flight: three (tap, k-slab) steps of f16, two 64-k steps of bytes
    half8 wh[ConvP8::D][MT];
    i32x8_x3 w8[2][MT];
};
template <int MT>
__device__ __forceinline__ void conv_p8_wload(ConvP8Window<MT>& W, const half8* const (&wph)[MT], int kc0, int nslab_ci, int st) {
    const size_t wo = size_t((st / ConvP8::NS) * nslab_ci + (kc0 >> 5) + st % ConvP8::NS) * 64;    // step st = tap st / NS, k-slab st % NS of this pass
#pragma unroll
    for (int m = 0; m < MT; ++m) W.wh[st % ConvP8::D][m] = wph[m][wo];
}
template <int MT>
__device__ __forceinline__ void conv_p8_wload8(ConvP8Window<MT>& W, const half8* const (&wp8)[MT], int kc0, int nslab_ci, int q) {
    // 64-k step q = tap q / 2, slabs 2 (q % 2), 2 (q % 2) + 1 of this pass: a lane's 32 bytes
    const size_t wo = size_t((q / (ConvP8::NS / 2)) * nslab_ci + (kc0 >> 5) + 2 * (q % (ConvP8::NS / 2))) * 64;
#pragma unroll
    for (int m = 0; m < MT; ++m) W.w8[q & 1][m] = x3_cat(wp8[m][wo], wp8[m][wo + 64]);
}
template <int MT>
__device__ __forceinline__ void conv_p8_prime(ConvP8Window<MT>& W, const half8* const (&wph)[MT], const half8* const (&wp8)[MT], int kc0, int nslab_ci) {
#pragma unroll
    for (int st = 0; st < ConvP8::D; ++st) conv_p8_wload<MT>(W, wph, kc0, nslab_ci, st);
    conv_p8_wload8<MT>(W, wp8, kc0, nslab_ci, 0);
    conv_p8_wload8<MT>(W, wp8, kc0, nslab_ci, 1);
}
// the 9 taps x 4 k-slabs of one staged pass (input channels kc0 ... kc0 + 127 in the tiles xh / x8), window primed by the caller
template <int MT>
__device__ __forceinline__ void conv_p8_pass(ConvP8Window<MT>& W, const half_t* xh, const char* x8, const half8* const (&wph)[MT],
                                             const half8* const (&wp8)[MT], int kc0, int nslab_ci, int l15, int lg, f32x4 (&acc)[MT][4]) {
    constexpr int ROWP = ConvP8::ROWP, R8 = ConvP8::R8, R8LO = ConvP8::R8LO, NS = ConvP8::NS, NSTEP = ConvP8::NSTEP, D = ConvP8::D;
    half8 bh[2][4];
    i32x8_x3 b8[4];
    auto row_of = [&](int tap, int t) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int sq = t * 16 + l15;
        const int ny = (sq >> 3) + dy, nx = (sq & 7) + dx;
        return (unsigned(ny) < 8u) && (unsigned(nx) < 8u) ? ny * 8 + nx : 64;
    };
    auto read_frag = [&](int st) {
        const int tap = st / NS, sl = st % NS;
#pragma unroll
        for (int t = 0; t < 4; ++t) bh[st & 1][t] = *reinterpret_cast<const half8*>(xh + row_of(tap, t) * ROWP + lg * 8 + sl * 32);
    };
    auto read_8 = [&](int q) {                                  // lane group lg: 0, 1 = hi8 of k [0, 32), [32, 64) of the step; 2, 3 = lo8 of the same
        const int tap = q / (NS / 2), J = q % (NS / 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const char* pp = x8 + row_of(tap, t) * R8 + (lg >> 1) * R8LO + J * 64 + (lg & 1) * 32;
            b8[t] = x3_cat(*reinterpret_cast<const half8*>(pp), *reinterpret_cast<const half8*>(pp + 16));
        }
    };
    read_frag(0);
    read_8(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) read_frag(st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.wh[st % D][m], bh[st & 1][t], acc[m][t], 0, 0, 0);
        if (st & 1) {
            const int q = st >> 1;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) x3_mfma8(W.w8[q & 1][m], b8[t], acc[m][t], true);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < NSTEP / 2) read_8(q + 1);
            if (q + 2 < NSTEP / 2) conv_p8_wload8<MT>(W, wp8, kc0, nslab_ci, q + 2);
        }
        if (st + D < NSTEP) conv_p8_wload<MT>(W, wph, kc0, nslab_ci, st + D);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// stages one board's input channels [kc0, kc0 + 128) as operand tiles: `request` into registers, `split_store` from them (NTHR = 512)
struct ConvP8Stage {
    static constexpr int NV = kSquares * (ConvP8::KC / 8) / 512;
    float pre[NV][8];
    __device__ __forceinline__ void request(const float* xb, int cin, int kc0, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * 512, r = i / (ConvP8::KC / 8), v = i - r * (ConvP8::KC / 8);
            load8<float>(xb + size_t(r) * cin + kc0 + v * 8, pre[j]);
        }
    }
    __device__ __forceinline__ void split_store(half_t* xh, char* x8, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * 512, r = i / (ConvP8::KC / 8), v = i - r * (ConvP8::KC / 8);
            half8 h, l;
            split8(pre[j], h, l);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x4 hw = __builtin_bit_cast(u32x4, h), lw = __builtin_bit_cast(u32x4, l);
            *reinterpret_cast<half8*>(xh + r * ConvP8::ROWP + v * 8) = h;
            *reinterpret_cast<u32x2*>(x8 + r * ConvP8::R8 + v * 8) = u32x2{x3_high_bytes(hw[0], hw[1]), x3_high_bytes(hw[2], hw[3])};
            *reinterpret_cast<u32x2*>(x8 + r * ConvP8::R8 + ConvP8::R8LO + v * 8) = u32x2{x3_high_bytes(lw[0], lw[1]), x3_high_bytes(lw[2], lw[3])};
        }
    }
};
__device__ __forceinline__ void conv_p8_zero_rows(char* smem, int tid) {      // row 64 of both buffers: what out-of-board taps read (halves of zeros = bytes of zeros)
    for (int i = tid; i < 2 * ConvP8::ROWP; i += 512) {
        const int buf = i / ConvP8::ROWP, c = i - buf * ConvP8::ROWP;
        ConvP8::xh_of(smem, buf)[64 * ConvP8::ROWP + c] = half_t(0.f);
        reinterpret_cast<half_t*>(ConvP8::x8_of(smem, buf))[64 * ConvP8::ROWP + c] = half_t(0.f);
    }
}
}  // namespace

template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void conv3x3_p8_kernel(const ConvArgs a) {
    static_assert(NW == 8, "512 threads stage a pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = ConvP8::KC;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const int co_tile0 = (blockIdx.x * NW + wave) * MT;
    bool active[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) active[m] = (co_tile0 + m) * 16 < a.cout_pad;
    const float* xb = reinterpret_cast<const float*>(a.x) + size_t(b) * kSquares * a.cin;
    const int nslab_ci = a.cin >> 5;
    const int nslab = 9 * nslab_ci;
    const half8 *wph[MT], *wp8[MT];                          // f16 image / 8-bit image: fragment (cout tile, slab) = 64 lanes x 16 B
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        wph[m] = reinterpret_cast<const half8*>(a.wpk) + size_t(active[m] ? co_tile0 + m : 0) * nslab * 64 + lane;
        wp8[m] = reinterpret_cast<const half8*>(a.wpk_lo) + size_t(active[m] ? co_tile0 + m : 0) * nslab * 64 + lane;
    }
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    conv_p8_zero_rows(smem, tid);
    ConvP8Stage stage;
    stage.request(xb, a.cin, 0, tid);
    stage.split_store(ConvP8::xh_of(smem, 0), ConvP8::x8_of(smem, 0), tid);
    const int npass = a.cin / KC;
    for (int pass = 0; pass < npass; ++pass) {
        const int kc0 = pass * KC;
        ConvP8Window<MT> W;
        if (active[0]) conv_p8_prime<MT>(W, wph, wp8, kc0, nslab_ci);
        __syncthreads();                                     // the pass's tiles are staged (and the other buffer is free: everybody is through the pass before)
        if (pass + 1 < npass) stage.request(xb, a.cin, kc0 + KC, tid);
        if (active[0]) conv_p8_pass<MT>(W, ConvP8::xh_of(smem, pass & 1), ConvP8::x8_of(smem, pass & 1), wph, wp8, kc0, nslab_ci, l15, lg, acc);
        if (pass + 1 < npass) stage.split_store(ConvP8::xh_of(smem, (pass + 1) & 1), ConvP8::x8_of(smem, (pass + 1) & 1), tid);
    }
    conv_x3_finish<MT, NW>(a, acc, active, smem, b, co_tile0, a.acc_scale);
}

// The policy head of a policy-map net in ONE launch (Precision float16p8): conv 3x3 256 -> 256 + BN + ReLU (a.pre_*: conv3x3_p8_kernel<2, 8>'s
// work), its output written straight into the two staging buffers as operand tiles (channels 0-127 / 128-255: exactly the two passes of the
// next conv), then conv 3x3 256 -> P on them (conv3x3_p8_kernel<1, 8>'s work: no staging, no barrier between its passes) and its epilogue
// (the board's softmax).  Saves a launch, 128 KB of HBM traffic per board and the second conv's exposed staging.
__global__ __launch_bounds__(512) void conv3x3_p8_chain_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = ConvP8::KC, NW = 8, C = 256;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const float* xb = reinterpret_cast<const float*>(a.x) + size_t(b) * kSquares * C;
    constexpr int nslab_ci = C >> 5, nslab = 9 * nslab_ci;
    {   // ---- conv 1: this wave's two cout tiles of the 16
        const half8 *wph[2], *wp8[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            wph[m] = reinterpret_cast<const half8*>(a.pre_wpk) + size_t(wave * 2 + m) * nslab * 64 + lane;
            wp8[m] = reinterpret_cast<const half8*>(a.pre_wpk_lo) + size_t(wave * 2 + m) * nslab * 64 + lane;
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        conv_p8_zero_rows(smem, tid);
        ConvP8Stage stage;
        stage.request(xb, C, 0, tid);
        stage.split_store(ConvP8::xh_of(smem, 0), ConvP8::x8_of(smem, 0), tid);
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int kc0 = pass * KC;
            ConvP8Window<2> W;
            conv_p8_prime<2>(W, wph, wp8, kc0, nslab_ci);
            __syncthreads();
            if (pass == 0) stage.request(xb, C, KC, tid);
            conv_p8_pass<2>(W, ConvP8::xh_of(smem, pass), ConvP8::x8_of(smem, pass), wph, wp8, kc0, nslab_ci, l15, lg, acc);
            if (pass == 0) stage.split_store(ConvP8::xh_of(smem, 1), ConvP8::x8_of(smem, 1), tid);
        }
        __syncthreads();                                         // every wave is through with the input tiles: they take conv 1's output
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c0 = (wave * 2 + m) * 16 + lg * 4;         // 4 consecutive output channels of this lane
            const f32x4 bs = *reinterpret_cast<const f32x4*>(a.pre_bias + c0);
            half_t* xh = ConvP8::xh_of(smem, c0 >> 7);
            char* x8 = ConvP8::x8_of(smem, c0 >> 7);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sq = t * 16 + l15;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[m][t][r], a.pre_acc_scale, bs[r]), 0.f);
                half4 h;
                uint32_t h8, l8;
                split4_b8(v, h, h8, l8);
                *reinterpret_cast<half4*>(xh + sq * ConvP8::ROWP + (c0 & 127)) = h;
                *reinterpret_cast<uint32_t*>(x8 + sq * ConvP8::R8 + (c0 & 127)) = h8;
                *reinterpret_cast<uint32_t*>(x8 + sq * ConvP8::R8 + ConvP8::R8LO + (c0 & 127)) = l8;
            }
        }
    }
    // ---- conv 2: one cout tile per wave
    bool active[1] = {wave * 16 < a.cout_pad};
    const half8 *wph[1], *wp8[1];
    wph[0] = reinterpret_cast<const half8*>(a.wpk) + size_t(active[0] ? wave : 0) * nslab * 64 + lane;
    wp8[0] = reinterpret_cast<const half8*>(a.wpk_lo) + size_t(active[0] ? wave : 0) * nslab * 64 + lane;
    f32x4 acc2[1][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    ConvP8Window<1> W;
    if (active[0]) conv_p8_prime<1>(W, wph, wp8, 0, nslab_ci);
    __syncthreads();                                             // conv 1's output tiles are written
    if (active[0]) {
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) conv_p8_prime<1>(W, wph, wp8, KC, nslab_ci);
            conv_p8_pass<1>(W, ConvP8::xh_of(smem, pass), ConvP8::x8_of(smem, pass), wph, wp8, pass * KC, nslab_ci, l15, lg, acc2);
        }
    }
    conv_x3_finish<1, NW>(a, acc2, active, smem, b, wave, a.acc_scale);
}

// ---- Precision float16x3: the policy head of a policy-map net in ONE launch (the float16x3 twin of conv3x3_p8_chain_kernel) ----
// conv 3x3 256 -> 256 + BN + ReLU (a.pre_*: conv_gemm_x3_kernel<3, 2, 8, 4>'s arithmetic, step for step) with both passes of the board staged
// in two buffers (the second pass's loads fly during the first pass's K loop: the two-launch form stages them one after the other, exposed);
// its output split straight into the same two buffers as operand tiles (channels 0-127 / 128-255: the two passes of the next conv -- the
// same halves the two-launch form makes of the floats it reads back); then conv 3x3 256 -> P on them (conv_gemm_x3_kernel<3, 1, 8, 4>'s
// arithmetic) and its epilogue, the board's softmax.  Same bits as the two launches; saves a launch, 128 KB of HBM traffic per board and
// every exposed staging pass but the first.
namespace {
struct ConvX3 {
    static constexpr int KC = X3_KC, ROWP = X3_ROWP, NS = KC / 32, NSTEP = 9 * NS, D = 3;
    static constexpr size_t buf_bytes = size_t(2) * 65 * ROWP * sizeof(half_t);          // hi tile + lo tile, 65 rows (row 64: zeros)
    static constexpr size_t lds_bytes = 2 * buf_bytes;                                     // 70.7 KB
    static __device__ __forceinline__ half_t* xh_of(char* smem, int buf) { return reinterpret_cast<half_t*>(smem + buf * buf_bytes); }
    static __device__ __forceinline__ half_t* xl_of(char* smem, int buf) { return xh_of(smem, buf) + 65 * ROWP; }
};
template <int MT> struct ConvX3Window { half8 wh[ConvX3::D][MT], wl[ConvX3::D][MT]; };
template <int MT>
__device__ __forceinline__ void conv_x3_wload(ConvX3Window<MT>& W, const half8* const (&wph)[MT], const half8* const (&wpl)[MT], int kc0, int nslab_ci, int st) {
    const size_t wo = size_t((st / ConvX3::NS) * nslab_ci + (kc0 >> 5) + st % ConvX3::NS) * 64;      // step st = tap st / NS, k-slab st % NS of this pass
#pragma unroll
    for (int m = 0; m < MT; ++m) { W.wh[st % ConvX3::D][m] = wph[m][wo]; W.wl[st % ConvX3::D][m] = wpl[m][wo]; }
}
template <int MT>
__device__ __forceinline__ void conv_x3_prime(ConvX3Window<MT>& W, const half8* const (&wph)[MT], const half8* const (&wpl)[MT], int kc0, int nslab_ci) {
#pragma unroll
    for (int st = 0; st < ConvX3::D; ++st) conv_x3_wload<MT>(W, wph, wpl, kc0, nslab_ci, st);
}
// the 9 taps x 4 k-slabs of one staged pass: conv_gemm_x3_body's static schedule (NS = 4), window primed by the caller
template <int MT>
__device__ __forceinline__ void conv_x3_pass(ConvX3Window<MT>& W, const half_t* xh, const half_t* xl, const half8* const (&wph)[MT],
                                             const half8* const (&wpl)[MT], int kc0, int nslab_ci, int l15, int lg, f32x4 (&acc)[MT][4]) {
    constexpr int ROWP = ConvX3::ROWP, NS = ConvX3::NS, NSTEP = ConvX3::NSTEP, D = ConvX3::D;
    half8 bh[2][4], bl[2][4];
    auto read_frag = [&](int st) {
        const int tap = st / NS, sl = st % NS, dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int sq = t * 16 + l15;
            const int ny = (sq >> 3) + dy, nx = (sq & 7) + dx;
            const bool ok = (unsigned(ny) < 8u) && (unsigned(nx) < 8u);
            const int off = (ok ? ny * 8 + nx : 64) * ROWP + lg * 8 + sl * 32;
            bh[st & 1][t] = *reinterpret_cast<const half8*>(xh + off);
            bl[st & 1][t] = *reinterpret_cast<const half8*>(xl + off);
        }
    };
    read_frag(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) read_frag(st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.wl[st % D][m], bh[st & 1][t], acc[m][t], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.wh[st % D][m], bl[st & 1][t], acc[m][t], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.wh[st % D][m], bh[st & 1][t], acc[m][t], 0, 0, 0);
        if (st + D < NSTEP) conv_x3_wload<MT>(W, wph, wpl, kc0, nslab_ci, st + D);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// stages one board's input channels [kc0, kc0 + 128) as split tiles: `request` into registers, `split_store` from them (512 threads)
struct ConvX3Stage {
    static constexpr int NV = kSquares * (ConvX3::KC / 8) / 512;
    float pre[NV][8];
    __device__ __forceinline__ void request(const float* xb, int cin, int kc0, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * 512, r = i / (ConvX3::KC / 8), v = i - r * (ConvX3::KC / 8);
            load8<float>(xb + size_t(r) * cin + kc0 + v * 8, pre[j]);
        }
    }
    __device__ __forceinline__ void split_store(half_t* xh, half_t* xl, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * 512, r = i / (ConvX3::KC / 8), v = i - r * (ConvX3::KC / 8);
            half8 h, l;
            split8(pre[j], h, l);
            *reinterpret_cast<half8*>(xh + r * ConvX3::ROWP + v * 8) = h;
            *reinterpret_cast<half8*>(xl + r * ConvX3::ROWP + v * 8) = l;
        }
    }
};
}  // namespace

__global__ __launch_bounds__(512) void conv3x3_x3_chain_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = ConvX3::KC, ROWP = ConvX3::ROWP, NW = 8, C = 256;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;
    const float* xb = reinterpret_cast<const float*>(a.x) + size_t(b) * kSquares * C;
    constexpr int nslab_ci = C >> 5, nslab = 9 * nslab_ci;
    {   // ---- conv 1: this wave's two cout tiles of the 16
        const half8 *wph[2], *wpl[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            wph[m] = reinterpret_cast<const half8*>(a.pre_wpk) + size_t(wave * 2 + m) * nslab * 64 + lane;
            wpl[m] = reinterpret_cast<const half8*>(a.pre_wpk_lo) + size_t(wave * 2 + m) * nslab * 64 + lane;
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 4 * ROWP; i += 512) {              // row 64 of all four tiles: what out-of-board taps read
            const int tile = i / ROWP, c = i - tile * ROWP;
            (tile & 1 ? ConvX3::xl_of(smem, tile >> 1) : ConvX3::xh_of(smem, tile >> 1))[64 * ROWP + c] = half_t(0.f);
        }
        ConvX3Stage stage;
        stage.request(xb, C, 0, tid);
        stage.split_store(ConvX3::xh_of(smem, 0), ConvX3::xl_of(smem, 0), tid);
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int kc0 = pass * KC;
            ConvX3Window<2> W;
            conv_x3_prime<2>(W, wph, wpl, kc0, nslab_ci);
            __syncthreads();
            if (pass == 0) stage.request(xb, C, KC, tid);
            conv_x3_pass<2>(W, ConvX3::xh_of(smem, pass), ConvX3::xl_of(smem, pass), wph, wpl, kc0, nslab_ci, l15, lg, acc);
            if (pass == 0) stage.split_store(ConvX3::xh_of(smem, 1), ConvX3::xl_of(smem, 1), tid);
        }
        __syncthreads();                                         // every wave is through with the input tiles: they take conv 1's output
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c0 = (wave * 2 + m) * 16 + lg * 4;         // 4 consecutive output channels of this lane
            const f32x4 bs = *reinterpret_cast<const f32x4*>(a.pre_bias + c0);
            half_t* xh = ConvX3::xh_of(smem, c0 >> 7);
            half_t* xl = ConvX3::xl_of(smem, c0 >> 7);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sq = t * 16 + l15;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[m][t][r], 1.f, bs[r]), 0.f);      // (conv_x3_finish's form: fma by acc_scale = 1, ReLU)
                half4 h, l;
                split4(v, h, l);
                *reinterpret_cast<half4*>(xh + sq * ROWP + (c0 & 127)) = h;
                *reinterpret_cast<half4*>(xl + sq * ROWP + (c0 & 127)) = l;
            }
        }
    }
    // ---- conv 2: one cout tile per wave
    bool active[1] = {wave * 16 < a.cout_pad};
    const half8 *wph[1], *wpl[1];
    wph[0] = reinterpret_cast<const half8*>(a.wpk) + size_t(active[0] ? wave : 0) * nslab * 64 + lane;
    wpl[0] = reinterpret_cast<const half8*>(a.wpk_lo) + size_t(active[0] ? wave : 0) * nslab * 64 + lane;
    f32x4 acc2[1][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    ConvX3Window<1> W;
    if (active[0]) conv_x3_prime<1>(W, wph, wpl, 0, nslab_ci);
    __syncthreads();                                             // conv 1's output tiles are written
    if (active[0]) {
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) conv_x3_prime<1>(W, wph, wpl, KC, nslab_ci);
            conv_x3_pass<1>(W, ConvX3::xh_of(smem, pass), ConvX3::xl_of(smem, pass), wph, wpl, pass * KC, nslab_ci, l15, lg, acc2);
        }
    }
    conv_x3_finish<1, NW>(a, acc2, active, smem, b, wave, 1.f);
}

static void launch_conv3x3_p8(const ConvArgs& a, hipStream_t s) {
    if (a.ks != 3 || a.cin % X3_KC != 0 || a.planes || a.out_rows_f32) throw std::invalid_argument("conv3x3_p8_kernel: a dense 3x3 conv with cin a multiple of 128");
    const size_t shmem = ConvP8::lds_bytes;                              // 70 KB: two (f16 tile + byte tile) buffers
    const int tiles = a.cout_pad / 16;
    if (a.pre_wpk) {                                                     // the policy head's two convs in one launch
        if (a.cin != 256 || tiles > 8 || a.resid) throw std::invalid_argument("conv3x3_p8_chain_kernel: 256 -> 256 -> at most 128 couts, no shortcut");
        hipLaunchKernelGGL(conv3x3_p8_chain_kernel, dim3(1, a.batch), dim3(512), shmem, s, a);
        return;
    }
    // (a.few_boards: a small batch leaves CUs idle, so a 256-cout layer goes as two workgroups of 128 couts per board)
    if ((tiles >= 12 && !a.few_boards) || (a.softmax_out && tiles > 8)) hipLaunchKernelGGL((conv3x3_p8_kernel<2, 8>), dim3((tiles + 15) / 16, a.batch), dim3(512), shmem, s, a);
    else hipLaunchKernelGGL((conv3x3_p8_kernel<1, 8>), dim3((tiles + 7) / 8, a.batch), dim3(512), shmem, s, a);
}
template <int KS, int NS> static void launch_conv_gemm_x3_ks(const ConvArgs& a, hipStream_t s) {
    const size_t shmem = size_t(2) * 65 * X3_ROWP * sizeof(half_t);      // 35 KB
    const int tiles = a.cout_pad / 16;
    if (a.out_rows_f32) {                   // an FC over the batch: few "boards" (64 rows each), so as many workgroups as the couts give
        hipLaunchKernelGGL((conv_gemm_x3_kernel<KS, 1, 4, NS>), dim3((tiles + 3) / 4, a.batch), dim3(256), shmem, s, a);
    } else if (a.few_boards >= 2 && tiles >= 8 && !a.softmax_out) {   // a small batch: the couts of a wide layer over tiles / 4 workgroups per board (CUs are idle, the launch is its latency)
        hipLaunchKernelGGL((conv_gemm_x3_kernel<KS, 1, 4, NS>), dim3((tiles + 3) / 4, a.batch), dim3(256), shmem, s, a);
    } else if ((tiles >= 12 && !a.few_boards) || (a.softmax_out && tiles > 8)) {   // 192 couts and more (or a fused softmax: the board in one workgroup): 8 waves x 2 tiles, the whole cout range of a 256-wide layer in one workgroup
        hipLaunchKernelGGL((conv_gemm_x3_kernel<KS, 2, 8, NS>), dim3((tiles + 15) / 16, a.batch), dim3(512), shmem, s, a);
    } else if (tiles >= 5) {                // 80 ... 176 couts: 8 waves x 1 tile (measured against 4 waves x 2 tiles: 0.036 / 0.042 ms for 96 couts)
        hipLaunchKernelGGL((conv_gemm_x3_kernel<KS, 1, 8, NS>), dim3((tiles + 7) / 8, a.batch), dim3(512), shmem, s, a);
    } else {
        hipLaunchKernelGGL((conv_gemm_x3_kernel<KS, 1, 4, NS>), dim3((tiles + 3) / 4, a.batch), dim3(256), shmem, s, a);
    }
}
template <int KS> static void launch_conv_gemm_x3_cin(const ConvArgs& a, hipStream_t s) {
    if (a.cin % X3_KC == 0) launch_conv_gemm_x3_ks<KS, X3_KC / 32>(a, s);     // every pass is a full one
    else if (a.cin == 64) launch_conv_gemm_x3_ks<KS, 2>(a, s);                // the stem's padded planes
    else launch_conv_gemm_x3_ks<KS, 0>(a, s);
}
void launch_conv_gemm_x3(const ConvArgs& a, hipStream_t s) {
    if (a.pre_wpk && !a.p8) {                                            // float16x3: the policy head's two convs in one launch
        if (a.ks != 3 || a.cin != 256 || a.cout_pad > 128 || a.resid || a.planes || a.out_rows_f32 || !a.out_policy_f32)
            throw std::invalid_argument("conv3x3_x3_chain_kernel: 256 -> 256 -> at most 128 couts of a policy map, no shortcut");
        hipLaunchKernelGGL(conv3x3_x3_chain_kernel, dim3(1, a.batch), dim3(512), ConvX3::lds_bytes, s, a);
        return;
    }
    if (a.p8) launch_conv3x3_p8(a, s);
    else if (a.ks == 1) launch_conv_gemm_x3_cin<1>(a, s);
    else launch_conv_gemm_x3_cin<3>(a, s);
}
bool heads_small_fits(const ConvArgs& c, const ValueHeadArgs& v) {
    return !c.p8 && c.ks == 3 && c.cin % X3_KC == 0 && !c.planes && !c.out_rows_f32 && c.out_policy_f32 && !c.resid && c.cout_pad <= 128 && !v.dbg && v.lds_pad < 0 &&
           v.variant == 0 && value_head_lds_bytes(v) <= 64 * 1024;
}
void launch_heads_small(const HeadsSmallArgs& a, hipStream_t s) {
    if (!heads_small_fits(a.conv, a.vh) || !a.conv.softmax_out || a.conv.batch != a.vh.batch) throw std::invalid_argument("heads_small_kernel: a 3x3 policy conv of at most 128 couts with its softmax, and the one-launch value head");
    const size_t shmem = std::max(size_t(2) * 65 * X3_ROWP * sizeof(half_t), value_head_lds_bytes(a.vh));
    const dim3 grid(16 * ((a.conv.batch + 7) / 8));
    hipLaunchKernelGGL(heads_small_kernel, grid, dim3(512), shmem, s, a);
}

// ================================================================================================================
// Fused 3x3 mobile-bottleneck blocks -- block_kernel_dpp (kernels.hip) with split operands: expand (MFMA x3) -> BN1 + ReLU + depthwise
// 3x3 + BN2 + ReLU on the f32 accumulators by DPP lane shifts (no LDS round trip, exact f32) -> split -> LDS -> project (MFMA x3)
// into the register accumulator -> + BN3 bias + x.  8 waves, one board per workgroup.
//   block_x3_kernel : one block per launch, x and y float [B][64][256] in HBM
//   tower_x3_kernel : a run of consecutive 3x3 blocks in ONE launch -- the residual stream stays in LDS as its hi / lo f16 pair
//                     (x to 2^-22) from the first block to the last, SE gates are computed in-kernel (exact f32)
// ================================================================================================================
namespace {
struct X3Block {
    static constexpr int C = 256, NW = 8, NE = 1, CK = 16 * NW * NE, NTHR = 64 * NW, NJ = C / 16 / NW;
    static constexpr int XROW = C + 16;      // halves; 32-byte row pad (16 and 48 bytes measured the same: profiles/r03/s_*)
    static constexpr int TROW = CK + 16;
    static constexpr size_t dws_bytes = size_t(NW) * NE * 1024;
    // NE = 1: the depthwise output tile is double-buffered -- ONE barrier per chunk (the depthwise of chunk k + 1 writes the other
    // buffer while slower waves still read chunk k's in their project phase), and the waves of a SIMD drift apart: one is in a
    // matrix phase while its partner runs the depthwise
    static constexpr int T2BUF = NE == 1 ? 2 : 1;
    static constexpr size_t lds_bytes = (size_t(2) * 64 * XROW + size_t(2) * T2BUF * 64 * TROW) * sizeof(half_t) + dws_bytes;   // NE = 1: 151,552 B; NE = 2: 155,648 B
};
static_assert(X3Block::lds_bytes + 8192 <= 160 * 1024, "LDS budget (tower_p8_kernel<5> takes 8 KiB more for its records)");

struct X3Tiles {
    half_t *xh, *xl;        // [64][XROW] block input = residual stream, hi / lo
    half_t *t2h, *t2l;      // [T2BUF][64][TROW] depthwise output of a chunk, hi / lo (buffer = chunk parity)
    float* dws;             // [8 waves][NE][256 floats] the waves' depthwise records of the chunk (16 rows x 16 channels each)
};
__device__ __forceinline__ X3Tiles x3_tiles(char* smem) {
    X3Tiles t;
    t.xh = reinterpret_cast<half_t*>(smem);
    t.xl = t.xh + 64 * X3Block::XROW;
    t.t2h = t.xl + 64 * X3Block::XROW;
    t.t2l = t.t2h + X3Block::T2BUF * 64 * X3Block::TROW;
    t.dws = reinterpret_cast<float*>(t.t2l + X3Block::T2BUF * 64 * X3Block::TROW);
    return t;
}

// Row order of the board tiles in LDS.  A 16-square MFMA tile t (rows t * 16 ... + 15 of the x and t2 tiles) holds board ranks t (lanes
// l15 = 0-7, files a-h) and t + 4 (lanes 8-15): the rank above / below a lane's square is then the SAME lane of tile t - 1 / t + 1, so the
// depthwise finds its vertical neighbours in the neighbouring accumulator registers without a lane shuffle or a select; only rank 4's upper
// and rank 3's lower neighbour cross the two halves (one row_ror:8 each).  Everything between staging and the store to HBM works on tile
// rows and never needs to know which square a row is.
__device__ __forceinline__ int x3_row(int sq) { return ((sq >> 3) & 3) * 16 + (sq >> 5) * 8 + (sq & 7); }      // square -> tile row
__device__ __forceinline__ int x3_square(int row) { return ((row >> 4) + 4 * ((row >> 3) & 1)) * 8 + (row & 7); }   // tile row -> square

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ f32x2 dpp_mov2(f32x2 v) { return f32x2{dpp_mov<CTRL>(v.x), dpp_mov<CTRL>(v.y)}; }
__device__ __forceinline__ f32x2 pair_of(const f32x4& v, int p) { return p == 0 ? f32x2{v[0], v[1]} : f32x2{v[2], v[3]}; }

// D of one 16-channel tile: BN1 bias + ReLU on the expand accumulators, depthwise 3x3, BN2 bias + ReLU, exact f32 in the tap order of
// block_kernel_dpp (kernels.hip).  A lane holds 4 channels (accumulator rows r) of one file on ranks t / t + 4 (x3_row); channels go two
// at a time (P = 0, 1).
// Horizontal neighbours: row_shr:1 / row_shl:1 copies, each feeding the three outputs it is up / mid / down neighbour of; lane 8 would
// read lane 7 (file h of the other rank) and lane 0 a zero: a lane on file a / h reads its dx = -1 / +1 weights from the record's ZERO rows
// (an address offset computed once per kernel, x3_edge_offsets; as six multiplies per channel the masks were 7 % of the depthwise).
// In pieces (load, gather<P>, taps<P>) so that a caller can spread them over a stretch of MFMAs.
//   rec: this tile's records in LDS, [16 rows: taps dx = -1 (dy = -1, 0, 1), dx = 0, dx = +1, BN1 bias, BN2 bias, 5 rows of zeros][16 channels]
struct X3EdgeOffsets { int left, right; };                       // in floats: 11 rows / 5 rows from the dx = -1 / +1 rows to the zero rows, or 0
__device__ __forceinline__ X3EdgeOffsets x3_edge_offsets(int l15) { return X3EdgeOffsets{(l15 & 7) == 0 ? 11 * 16 : 0, (l15 & 7) == 7 ? 5 * 16 : 0}; }
struct X3Depthwise {
    f32x2 w[11];                                                 // the current channel pair's records (rows 0 ... 10)
    f32x2 S[6], L[6], R[6];                                      // rank - 1 ... rank + 4 of this lane's half: S[1 + t] = tile t
    float outv[4][4];                                            // [tile][channel r]

    template <int P> __device__ __forceinline__ void load(const float* rec, int lg, const X3EdgeOffsets& e) {
#pragma unroll
        for (int q = 0; q < 11; ++q) w[q] = *reinterpret_cast<const f32x2*>(rec + (q < 3 ? e.left : q >= 6 && q < 9 ? e.right : 0) + q * 16 + lg * 4 + 2 * P);
    }
    // acc_scale: the accumulators carry the weights' power-of-two scale (Precision float16p8): S = relu(acc * acc_scale + bias), one FMA instead of the add
    template <int P, bool SCALED = false> __device__ __forceinline__ void gather(const f32x4 (&acc)[4], bool upper, int c0 = 0, int c1 = 2, float acc_scale = 1.f) {
        if constexpr (X3_ABL & 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) S[1 + t] = pair_of(acc[t], P) + w[0];
            return;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c < c0 || c >= c1) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t) S[1 + t][c] = SCALED ? fmaxf(fmaf(acc[t][2 * P + c], acc_scale, w[9][c]), 0.f) : fmaxf(acc[t][2 * P + c] + w[9][c], 0.f);
            const float across_up = dpp_mov<DPP_ROW_ROR8>(S[4][c]), across_dn = dpp_mov<DPP_ROW_ROR8>(S[1][c]);
            S[0][c] = upper ? across_up : 0.f;                    // above rank 4 lies rank 3 (tile 3, other half); above rank 0 the edge
            S[5][c] = upper ? 0.f : across_dn;                    // below rank 3 lies rank 4 (tile 0, other half); below rank 7 the edge
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                L[j][c] = dpp_mov<DPP_ROW_SHR1>(S[j][c]);
                R[j][c] = dpp_mov<DPP_ROW_SHL1>(S[j][c]);
            }
        }
    }
    // plain v_fmac_f32, NOT v_pk_fma_f32: a packed f32 FMA does not run in the shadow of MFMAs (scripts/ubench/mix_kinds.hip: an MFMA
    // followed by two of them 38.5 cycles, by two v_fmac_f32 18.5; beside another wave's MFMAs 14.7 cycles each against 8.75)
    template <int P> __device__ __forceinline__ void taps(int t0, int t1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < t0 || t >= t1) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (X3_ABL & 1) {
                    outv[t][2 * P + c] = S[1 + t][c];
                    continue;
                }
                float a = w[10][c];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    a = fmaf(w[dy][c], L[t + dy][c], a);
                    a = fmaf(w[3 + dy][c], S[t + dy][c], a);
                    a = fmaf(w[6 + dy][c], R[t + dy][c], a);
                }
                outv[t][2 * P + c] = fmaxf(a, 0.f);
            }
        }
    }
    // keeps the values computed so far where they were written (a piece set between MFMAs is otherwise sunk to its first use)
    __device__ __forceinline__ void pin_taps(int t0, int t1, int P) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (t >= t0 && t < t1) asm volatile("" : "+v"(outv[t][2 * P + c]));
    }
};
__device__ __forceinline__ void x3_depthwise(const f32x4 (&acc)[4], const float* rec, int lg, bool upper, const X3EdgeOffsets& e, float (&outv)[4][4]) {
    X3Depthwise dw;
    dw.template load<0>(rec, lg, e);
    dw.template gather<0>(acc, upper);
    dw.template taps<0>(0, 4);
    dw.template load<1>(rec, lg, e);
    dw.template gather<1>(acc, upper);
    dw.template taps<1>(0, 4);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) outv[t][r] = dw.outv[t][r];
}
// D of one 16-channel tile with a 5x5 depthwise (RISEv3.3's wide blocks; Precision float16p8's tower_p8_kernel<5>): X3Depthwise's scheme
// on ranks - 2 ... + 2 and files - 2 ... + 2, one channel at a time (25 weights in registers).  The rank above / below a lane's square is the
// same lane of the neighbouring tile; two rows on either side of the rank 3 / 4 seam come from the other half of the row (row_ror:8).
// Horizontal neighbours are row_shr / row_shl copies by 1 and 2; a lane whose file lacks a neighbour reads that column's weights from the
// record's zero rows (X3EdgeOffsets5).
//   rec: this tile's records in LDS, [32 rows: taps column dx = -2 (dy = -2 ... 2), dx = -1, 0, +1, +2, BN1 bias, BN2 bias, 5 rows of zeros][16 channels]
struct X3EdgeOffsets5 { int o[5]; };                             // in floats, per tap column: to the zero rows 27 ... 31, or 0
__device__ __forceinline__ X3EdgeOffsets5 x3_edge_offsets5(int l15) {
    const int f = l15 & 7;
    X3EdgeOffsets5 e;
    e.o[0] = f < 2 ? 27 * 16 : 0;
    e.o[1] = f < 1 ? 22 * 16 : 0;
    e.o[2] = 0;
    e.o[3] = f > 6 ? 12 * 16 : 0;
    e.o[4] = f > 5 ? 7 * 16 : 0;
    return e;
}
struct X3Depthwise5 {
    float w[27];                                                 // the current channel's records (rows 0 ... 26)
    float S[8];                                                  // rank - 2 ... rank + 5 of this lane's half: S[2 + t] = tile t
    float outv[4][4];                                            // [tile][channel r]

    template <int CH> __device__ __forceinline__ void load(const float* rec, int lg, const X3EdgeOffsets5& e) {
#pragma unroll
        for (int q = 0; q < 27; ++q) w[q] = rec[(q < 25 ? e.o[q / 5] : 0) + q * 16 + lg * 4 + CH];
    }
    template <int CH> __device__ __forceinline__ void gather(const f32x4 (&acc)[4], bool upper, float acc_scale) {
#pragma unroll
        for (int t = 0; t < 4; ++t) S[2 + t] = fmaxf(fmaf(acc[t][CH], acc_scale, w[25]), 0.f);
        const float u3 = dpp_mov<DPP_ROW_ROR8>(S[5]), u2 = dpp_mov<DPP_ROW_ROR8>(S[4]);
        const float d0 = dpp_mov<DPP_ROW_ROR8>(S[2]), d1 = dpp_mov<DPP_ROW_ROR8>(S[3]);
        S[1] = upper ? u3 : 0.f;                                  // above rank 4 lies rank 3 (tile 3, other half), above that rank 2; above rank 0 the edge
        S[0] = upper ? u2 : 0.f;
        S[6] = upper ? 0.f : d0;                                  // below rank 3 lie ranks 4, 5 (tiles 0, 1, other half); below rank 7 the edge
        S[7] = upper ? 0.f : d1;
    }
    template <int CH> __device__ __forceinline__ void taps() {
        float a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = w[26];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v[5] = {dpp_mov<DPP_ROW_SHR2>(S[j]), dpp_mov<DPP_ROW_SHR1>(S[j]), S[j], dpp_mov<DPP_ROW_SHL1>(S[j]), dpp_mov<DPP_ROW_SHL2>(S[j])};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int dy = j - t;
                if (dy < 0 || dy > 4) continue;
#pragma unroll
                for (int g = 0; g < 5; ++g) a[t] = fmaf(w[g * 5 + dy], v[g], a[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) outv[t][CH] = fmaxf(a[t], 0.f);
    }
};
// float board tile [64][256] (optionally x := x * gate[c], _ChannelAttentionModule.forward, builder_util.py:114) -> split tiles
__device__ __forceinline__ void x3_stage_tile(const X3Tiles& T, const float* xb, const float* g, int tid) {
    constexpr int C = X3Block::C, XROW = X3Block::XROW;
#pragma unroll 1
    for (int i = tid; i < 64 * (C / 8); i += X3Block::NTHR) {
        const int sq = i / (C / 8), v = i - sq * (C / 8), r = x3_row(sq);
        float f[8];
        load8<float>(xb + size_t(sq) * C + v * 8, f);
        if (g) {
            float gv[8];
            load8<float>(g + v * 8, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= gv[j];
        }
        half8 h, l;
        split8(f, h, l);
        *reinterpret_cast<half8*>(T.xh + r * XROW + v * 8) = h;
        *reinterpret_cast<half8*>(T.xl + r * XROW + v * 8) = l;
    }
}

// The chunk loop of one block: accP[j][t] += project(depthwise(expand(x))) for this wave's 32 couts x 64 squares.  The caller has put a
// barrier behind the last write of the x tiles.  On return every wave is done with the x tiles (the last expand phase lies before the
// last chunk barrier); other waves may still be reading t2 in their last project phase.
// Weights are read with raw buffer loads: resource descriptor + wave-uniform byte offset in SGPRs, the lane part one constant VGPR.
// (Through the pointers of a descriptor array in device memory the compiler can only emit FLAT loads, which count on the LDS
// counter too: every wait for an LDS operand then also waits for the weight fragments requested slabs ahead.)
struct X3Weights {
    __amdgpu_buffer_rsrc_t w1h, w1l, w3h, w3l;   // packed expand / project weights, hi / lo (kernels.h: packed-weight geometry)
    __amdgpu_buffer_rsrc_t dw;                   // [cop_pad / 16 tiles][16 rows: taps, BN1 bias, BN2 bias, zeros (X3Depthwise)][16 channels] floats
    int cop_pad;
};
// The pointer is wave-uniform, but read from a descriptor array in device memory the compiler has it in VGPRs and wraps EVERY buffer load
// in a waterfall loop (readfirstlane / compare / saveexec / branch: ~12 instructions per load, 33 loads per chunk): say so explicitly.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint64_t u = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))))) << 32) |
                       uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ X3Weights x3_weights(const void* w1h, const void* w1l, const void* w3h, const void* w3l, const float* dwpk, int cop_pad) {
    X3Weights W;
    W.w1h = x3_rsrc(w1h); W.w1l = x3_rsrc(w1l); W.w3h = x3_rsrc(w3h); W.w3l = x3_rsrc(w3l);
    W.dw = x3_rsrc(dwpk);
    W.cop_pad = __builtin_amdgcn_readfirstlane(cop_pad);
    return W;
}
__device__ __forceinline__ half8 x3_frag(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t frag) {      // fragment = 64 lanes x 16 B
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, frag * 1024u, 0));
}
// The chunk loop of one block: accP[j][t] += project(depthwise(expand(x))) for this wave's 32 couts x 64 squares.  The caller has put a
// barrier behind the last write of the x tiles.  On return every wave is done with the x tiles (the last expand phase lies before the
// last chunk barrier); other waves may still be reading t2 in their last project phase.
// Measured and NOT kept (profiles/r03/o_*): P(ch - 1) and D(ch) as one instruction stream per wave (the depthwise cut in four pieces
// between the project MFMAs, sched_group_barrier recipes): 0.754 ms against 0.710 for this form -- 256 registers with spills, and the
// scheduler interleaved only half of the stretches.
// ch0 / ch1: the chunk range [ch0, ch1) of the block (block_x3_split_kernel); default: all of them.  PRE: the caller has requested the
// first chunk's expand window already (x3_expand_window_request, before it staged the board) and hands the registers in.
struct X3ExpandWindow { half8 h[X3Block::NE == 1 ? 4 : 2][X3Block::NE], l[X3Block::NE == 1 ? 4 : 2][X3Block::NE]; };
__device__ __forceinline__ void x3_expand_window_request(X3ExpandWindow& E, const X3Weights& W, int ch) {
    constexpr int EW = X3Block::NE == 1 ? 4 : 2, NE = X3Block::NE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
#pragma unroll
    for (int s = 0; s < EW; ++s)
#pragma unroll
        for (int ne = 0; ne < NE; ++ne) {
            const uint32_t f = uint32_t(ch * (X3Block::CK / 16) + wave * NE + ne) * (X3Block::C / 32) + uint32_t(s);
            E.h[s][ne] = x3_frag(W.w1h, uint32_t(lane) * 16u, f);
            E.l[s][ne] = x3_frag(W.w1l, uint32_t(lane) * 16u, f);
        }
}
template <bool PRE = false>
__device__ __forceinline__ void x3_chunks(const X3Tiles& T, const X3Weights& W, f32x4 (&accP)[X3Block::NJ][4], int ch0 = 0, int ch1 = -1,
                                          const X3ExpandWindow* pre = nullptr) {
    using G = X3Block;
    constexpr int C = G::C, CK = G::CK, XROW = G::XROW, TROW = G::TROW, NJ = G::NJ, NE = G::NE;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane_off = uint32_t(lane) * 16u;
    const int nchunk = ch1 < 0 ? W.cop_pad / CK : ch1;
    const int nslab3 = W.cop_pad >> 5;
    const bool hi = l15 >= 8;                                  // the tile's second rank (t + 4, x3_row)
    const X3EdgeOffsets edge = x3_edge_offsets(l15);             // a lane on file a / h has no left / right neighbour on the board

    // Weight fragments come through two small rolling windows (a phase's whole set in registers leaves the compiler no room in the
    // one-launch tower): the expand window holds EW of the 8 k-slabs (hi + lo, NE channel tiles), the project window PW of CK / 32.
    // A slot is refilled as soon as its MFMAs are issued, EW (PW) slabs ahead of its next use; the first slabs of a phase are
    // requested a whole phase ahead.
    constexpr int EW = NE == 1 ? 4 : 2, PW = 2;
    half8 e_h[EW][NE], e_l[EW][NE];
    if constexpr (X3_ABL & 16) {
#pragma unroll
        for (int s = 0; s < EW; ++s)
#pragma unroll
            for (int ne = 0; ne < NE; ++ne) e_h[s][ne] = e_l[s][ne] = *reinterpret_cast<const half8*>(T.xh + lane * 8);
    }
    auto load_expand = [&](int ch, int s) {                    // cout tile (16 channels) of (chunk, wave, ne) = ch * CK / 16 + wave * NE + ne
        if constexpr (X3_ABL & 16) return;
#pragma unroll
        for (int ne = 0; ne < NE; ++ne) {
            const uint32_t f = uint32_t(ch * (CK / 16) + wave * NE + ne) * (C / 32) + uint32_t(s);
            e_h[s % EW][ne] = x3_frag(W.w1h, lane_off, f);
            e_l[s % EW][ne] = x3_frag(W.w1l, lane_off, f);
        }
    };
    if constexpr (PRE) {
#pragma unroll
        for (int s = 0; s < EW; ++s)
#pragma unroll
            for (int ne = 0; ne < NE; ++ne) { e_h[s][ne] = pre->h[s][ne]; e_l[s][ne] = pre->l[s][ne]; }
    } else {
#pragma unroll
        for (int s = 0; s < EW; ++s) load_expand(ch0, s);
    }
    for (int ch = ch0; ch < nchunk; ++ch) {
        half_t* const t2h = T.t2h + (G::T2BUF == 2 ? (ch & 1) * 64 * TROW : 0);
        half_t* const t2l = T.t2l + (G::T2BUF == 2 ? (ch & 1) * 64 * TROW : 0);
        // ---------------- E: expand, NE x 16 channels x 64 squares per wave, K = C; a tile fragment of the stream feeds NE channel tiles ----
        f32x4 accE[NE][4];
#pragma unroll
        for (int ne = 0; ne < NE; ++ne)
#pragma unroll
            for (int t = 0; t < 4; ++t) accE[ne][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Depthwise records (per 16-channel tile 16 rows of 16 floats: X3Depthwise) of my NE tiles: ONE 16-byte load per lane and tile
        // (the tile's 1 KiB), parked in a wave-private LDS scratch half-way through the expand MFMAs and read back per lane as 11
        // broadcast reads per channel pair.  (Loaded per lane straight from L2 -- 12 loads of which 16 lanes each fetch the same
        // bytes -- the records were a quarter of all bytes on the 64 B/clk L2 -> CU path, which this loop nearly saturates.)
        f32x4 dw_raw[NE];
#pragma unroll
        for (int ne = 0; ne < NE; ++ne)
            dw_raw[ne] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W.dw, lane_off, uint32_t(ch * CK + (wave * NE + ne) * 16) * 64u, 0));
        float* my_dws = T.dws + (wave * NE) * 256;
        auto park_dw = [&]() {
#pragma unroll
            for (int ne = 0; ne < NE; ++ne) *reinterpret_cast<f32x4*>(my_dws + ne * 256 + lane * 4) = dw_raw[ne];
        };
        // A slab = 12 * NE MFMAs on the stream fragments of one k-slab.  The NEXT slab's fragments are read from LDS before this slab's
        // MFMAs issue and the window refills right behind them; the fences keep the machine scheduler from sinking either to just in
        // front of their consumers (it does, to shorten live ranges: every slab then waits out an LDS or L2 latency).
        half8 bh[2][4], bl[2][4];
        auto read_stream = [&](int s, half8 (&h)[4], half8 (&l)[4]) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (X3_ABL & 8) {
                    h[t] = e_h[s % EW][0];
                    l[t] = e_l[s % EW][0];
                } else {
                    h[t] = *reinterpret_cast<const half8*>(T.xh + (t * 16 + l15) * XROW + s * 32 + lg * 8);
                    l[t] = *reinterpret_cast<const half8*>(T.xl + (t * 16 + l15) * XROW + s * 32 + lg * 8);
                }
            }
        };
        read_stream(0, bh[0], bl[0]);
#pragma unroll
        for (int s = 0; s < C / 32; ++s) {
            if (s + 1 < C / 32) read_stream(s + 1, bh[(s + 1) & 1], bl[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ne = 0; ne < NE; ++ne)
#pragma unroll
                for (int t = 0; t < 4; ++t) x3_mfma(e_l[s % EW][ne], bh[s & 1][t], accE[ne][t], !(X3_ABL & 2));
#pragma unroll
            for (int ne = 0; ne < NE; ++ne)
#pragma unroll
                for (int t = 0; t < 4; ++t) x3_mfma(e_h[s % EW][ne], bl[s & 1][t], accE[ne][t], !(X3_ABL & 2));
#pragma unroll
            for (int ne = 0; ne < NE; ++ne)
#pragma unroll
                for (int t = 0; t < 4; ++t) x3_mfma(e_h[s % EW][ne], bh[s & 1][t], accE[ne][t], !(X3_ABL & 2));
            if (s + EW < C / 32) load_expand(ch, s + EW);
            if (s == C / 64) park_dw();                        // the records' load is half a phase old by now
            __builtin_amdgcn_sched_barrier(0);
        }
        // the first slabs of this chunk's project fragments: they land while the depthwise runs
        half8 p_h[PW][NJ], p_l[PW][NJ];
        if constexpr (X3_ABL & 16) {
#pragma unroll
            for (int s2 = 0; s2 < PW; ++s2)
#pragma unroll
                for (int j = 0; j < NJ; ++j) p_h[s2][j] = p_l[s2][j] = *reinterpret_cast<const half8*>(T.xl + lane * 8);
        }
        auto load_project = [&](int s2) {
            if constexpr (X3_ABL & 16) return;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const uint32_t f = uint32_t(wave * NJ + j) * uint32_t(nslab3) + uint32_t(ch * (CK / 32) + s2);
                p_h[s2 % PW][j] = x3_frag(W.w3h, lane_off, f);
                p_l[s2 % PW][j] = x3_frag(W.w3l, lane_off, f);
            }
        };
#pragma unroll
        for (int s2 = 0; s2 < PW; ++s2) load_project(s2);

        // ---------------- D: BN1 + ReLU, depthwise 3x3 on the accumulators (block_kernel_dpp), BN2 + ReLU, exact f32 ----------------
#pragma unroll
        for (int ne = 0; ne < NE; ++ne) {
            float outv[4][4];                                   // [tile][channel r]
            x3_depthwise(accE[ne], my_dws + ne * 256, lg, hi, edge, outv);
            const int cl = (wave * NE + ne) * 16 + lg * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                half4 h, l;
                split4(outv[t], h, l);
                if constexpr (X3_ABL & 64) {
                    asm volatile("" ::"v"(h), "v"(l));
                } else {
                    *reinterpret_cast<half4*>(t2h + (t * 16 + l15) * TROW + cl) = h;
                    *reinterpret_cast<half4*>(t2l + (t * 16 + l15) * TROW + cl) = l;
                }
            }
        }
        if constexpr (!(X3_ABL & 32)) __syncthreads();
        if (ch + 1 < nchunk) {                                  // the next chunk's first expand slabs land while the project MFMAs run
#pragma unroll
            for (int s = 0; s < EW; ++s) load_expand(ch + 1, s);
        }
        // ---------------- P: project, 32 couts x 64 squares per wave, K = CK (accumulates over chunks) ----------------
        auto read_t2 = [&](int s2, half8 (&h)[4], half8 (&l)[4]) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (X3_ABL & 8) {
                    h[t] = p_h[s2 % PW][0];
                    l[t] = p_l[s2 % PW][0];
                } else {
                    h[t] = *reinterpret_cast<const half8*>(t2h + (t * 16 + l15) * TROW + s2 * 32 + lg * 8);
                    l[t] = *reinterpret_cast<const half8*>(t2l + (t * 16 + l15) * TROW + s2 * 32 + lg * 8);
                }
            }
        };
        read_t2(0, bh[0], bl[0]);
#pragma unroll
        for (int s2 = 0; s2 < CK / 32; ++s2) {
            if (s2 + 1 < CK / 32) read_t2(s2 + 1, bh[(s2 + 1) & 1], bl[(s2 + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) x3_mfma(p_l[s2 % PW][j], bh[s2 & 1][t], accP[j][t], !(X3_ABL & 4));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) x3_mfma(p_h[s2 % PW][j], bl[s2 & 1][t], accP[j][t], !(X3_ABL & 4));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) x3_mfma(p_h[s2 % PW][j], bh[s2 & 1][t], accP[j][t], !(X3_ABL & 4));
            if (s2 + PW < CK / 32) load_project(s2 + PW);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!(X3_ABL & 32) && G::T2BUF == 1) __syncthreads();     // single t2 buffer: it is rewritten by the next chunk's depthwise
    }
}
}  // namespace

__global__ __launch_bounds__(512) void block_x3_kernel(const BlockArgs a) {
    using G = X3Block;
    constexpr int C = G::C, XROW = G::XROW, NJ = G::NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const X3Tiles T = x3_tiles(smem);
    const int b = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const X3Weights W = x3_weights(a.w1pk, a.w1pk_lo, a.w3pk, a.w3pk_lo, a.dwpk, a.cop_pad);
    x3_stage_tile(T, reinterpret_cast<const float*>(a.x) + size_t(b) * 64 * C, a.gate ? a.gate + size_t(b) * C : nullptr, tid);
    __syncthreads();

    f32x4 accP[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) accP[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    x3_chunks(T, W, accP);

    // ---------------- epilogue: + BN3 bias + residual (hi + lo of the staged input = x to 2^-22) ----------------
    float* yb = reinterpret_cast<float*>(a.y) + size_t(b) * 64 * C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int co0 = (wave * NJ + j) * 16 + lg * 4;
        float bs[4];
        load4<float>(a.b3 + co0, bs);
        float pool[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int sq = t * 16 + l15;
            float rh[4], rl[4], v[4];
            load4<half_t>(T.xh + sq * XROW + co0, rh);
            load4<half_t>(T.xl + sq * XROW + co0, rl);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = accP[j][t][r] + bs[r] + (rh[r] + rl[r]);
                pool[r] += v[r];
            }
            store4<float>(yb + size_t(x3_square(sq)) * C + co0, v);
        }
        if (a.pool_out) {                                       // squeeze (AdaptiveAvgPool2d) of the block output, fused here
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) pool[r] += __shfl_xor(pool[r], off, 64);
            if (l15 == 0) store4<float>(a.pool_out + size_t(b) * C + co0, pool);
        }
    }
}

// ---- the run of blocks in one launch ----
namespace {
// SE gate of a block on the stream in LDS, exact f32 (the arithmetic of se_gate_kernel / se_kernel, kernels.hip): squeeze over the 64
// squares, ca_se: relu(W1 mean) -> W2 -> hard-sigmoid, eca_se: centre-tap linear + bias -> hard-sigmoid, then x := x * gate, re-split.
// The scheme of the float16 tower's SE phase (tower.hip) with float weights: the gate matrices are host-packed in THREAD order
// (rise_net.hip: pack_se_threads_f32), a thread's 64 (eca: 128) weights are 16 (32) coalesced 16-byte loads that are all in flight
// before the squeeze ends; the threads that share an output pair are neighbouring lanes and reduce by DPP.
//   ca_se : FC1 thread t -> hidden 2*(t/8), +1 over c in [32*(t%8), +32);  FC2 thread t -> gate 2*(t/4), +1 over j in [32*(t%4), +32)
//   eca_se: thread t -> gate 2*(t/4), +1 over inputs i in [64*(t%4), +64)
// scratch (the t2 tiles, idle between blocks): mean 8 x 36, hidden 4 x 36, gate 256 floats.  Ends with a barrier.
// pre_wa: the thread's first 16 weight loads, requested by the caller before it staged the board (block_x3_split_kernel), else nullptr
//
// x3_se_fcs: the gate from the channel means (se_mean, LDS, written and barrier'd by the caller): both FC stages of ca_se / the one of eca_se; ends with a barrier
__device__ __forceinline__ void x3_se_fcs(const X3TowerBlock& d, const float* se_mean, float* se_h, float* se_gate, const f32x4 (&wa)[16], const f32x4 (&wb)[16], int tid) {
    constexpr int GRP = 36;
    auto dot32 = [](const f32x4 (&w)[16], const float* v, float& s0, float& s1) {   // v: 32 floats, 16-byte aligned; w[i] = (a, b, a', b') of k = 2i, 2i+1
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(v + 4 * k4);
            s0 = fmaf(w[2 * k4][0], m[0], s0); s1 = fmaf(w[2 * k4][1], m[0], s1);
            s0 = fmaf(w[2 * k4][2], m[1], s0); s1 = fmaf(w[2 * k4][3], m[1], s1);
            s0 = fmaf(w[2 * k4 + 1][0], m[2], s0); s1 = fmaf(w[2 * k4 + 1][1], m[2], s1);
            s0 = fmaf(w[2 * k4 + 1][2], m[3], s0); s1 = fmaf(w[2 * k4 + 1][3], m[3], s1);
        }
    };
    if (d.se_kind == 1) {
        {
            const int j2 = tid >> 3, kq = tid & 7;
            float s0 = 0.f, s1 = 0.f;
            dot32(wa, se_mean + kq * GRP, s0, s1);
            s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
            s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
            s0 += dpp_mov<0x114>(s0); s1 += dpp_mov<0x114>(s1);             // lane 7 of the group of 8: the whole sum
            if (kq == 7) {                                                    // hidden j = 2*j2, +1 at (j / 32) * 36 + j % 32
                float* h = se_h + (j2 >> 4) * GRP + 2 * (j2 & 15);
                h[0] = fmaxf(s0, 0.f);
                h[1] = fmaxf(s1, 0.f);
            }
        }
        __syncthreads();
        {
            const int c2 = tid >> 2, kq = tid & 3;
            float s0 = 0.f, s1 = 0.f;
            dot32(wb, se_h + kq * GRP, s0, s1);
            s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
            s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);             // lane 3 of the group of 4
            if (kq == 3) {
                se_gate[2 * c2] = hard_sigmoid(s0);
                se_gate[2 * c2 + 1] = hard_sigmoid(s1);
            }
        }
    } else {
        const int c2 = tid >> 2, kq = tid & 3;
        float s0 = 0.f, s1 = 0.f;
        dot32(wa, se_mean + (2 * kq) * GRP, s0, s1);
        dot32(wb, se_mean + (2 * kq + 1) * GRP, s0, s1);
        s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
        s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
        if (kq == 3) {
            se_gate[2 * c2] = hard_sigmoid(d.se_b[2 * c2] + s0);
            se_gate[2 * c2 + 1] = hard_sigmoid(d.se_b[2 * c2 + 1] + s1);
        }
    }
    __syncthreads();
}
struct X3SeFirstWeights { f32x4 w[16]; };
__device__ __forceinline__ void x3_se_first_weights_request(X3SeFirstWeights& S, const X3TowerBlock& d, int tid) {
    const f32x4* pk = reinterpret_cast<const f32x4*>(d.se_w1t) + tid;
#pragma unroll
    for (int i = 0; i < 16; ++i) S.w[i] = pk[i * 512];
}
__device__ __forceinline__ void x3_se_phase(const X3Tiles& T, const X3TowerBlock& d, float* scratch, int tid, const X3SeFirstWeights* pre_wa = nullptr) {
    constexpr int C = X3Block::C, XROW = X3Block::XROW, GRP = 36;     // floats per group of 32 means / hidden values (bank spread)
    float* se_mean = scratch;              // [8][36]
    float* se_h = scratch + 8 * GRP;       // [4][36]
    float* se_gate = se_h + 4 * GRP;       // [256]
    f32x4 wa[16], wb[16];
    auto load_thread_weights = [&](const float* base, f32x4 (&dst)[16]) {
        const f32x4* pk = reinterpret_cast<const f32x4*>(base) + tid;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i] = pk[i * 512];
    };
    if (pre_wa) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wa[i] = pre_wa->w[i];
        load_thread_weights(d.se_kind == 1 ? d.se_w2t : d.se_w1t + size_t(16) * 512 * 4, wb);      // (the second set flies during the squeeze)
    } else load_thread_weights(d.se_w1t, wa);
    {   // squeeze: a wave owns 32 channels: lane = (4 groups of 8 channels) x (16 groups of 4 squares); 16-lane DPP row reduction
        const int lane = tid & 63, wv = tid >> 6, cg = lane >> 4, sg = lane & 15;
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float fh[8], fl[8];
            load8<half_t>(T.xh + x3_row(sg * 4 + q) * XROW + wv * 32 + cg * 8, fh);     // squares in board order: the sum's rounding does not depend on the tile-row order
            load8<half_t>(T.xl + x3_row(sg * 4 + q) * XROW + wv * 32 + cg * 8, fl);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] += fh[j] + fl[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sum[j] += dpp_mov<0x111>(sum[j]);    // row_shr:1
            sum[j] += dpp_mov<0x112>(sum[j]);    // row_shr:2
            sum[j] += dpp_mov<0x114>(sum[j]);    // row_shr:4
            sum[j] += dpp_mov<0x118>(sum[j]);    // row_shr:8 -> lane 15 of the row holds the row's sum
        }
        if (sg == 15) {
#pragma unroll
            for (int j = 0; j < 8; ++j) se_mean[wv * GRP + cg * 8 + j] = sum[j] * (1.f / 64.f);      // channel c at (c / 32) * 36 + c % 32
        }
    }
    if (!pre_wa) load_thread_weights(d.se_kind == 1 ? d.se_w2t : d.se_w1t + size_t(16) * 512 * 4, wb);      // flies during FC1 (eca: the second half)
    __syncthreads();
    x3_se_fcs(d, se_mean, se_h, se_gate, wa, wb, tid);
#pragma unroll 1
    for (int i = tid; i < 64 * (C / 8); i += X3Block::NTHR) {      // x := x * gate (the residual uses the gated x, builder_util.py:473-475)
        const int r = i / (C / 8), v = i - r * (C / 8);
        float fh[8], fl[8], gv[8];
        load8<half_t>(T.xh + r * XROW + v * 8, fh);
        load8<half_t>(T.xl + r * XROW + v * 8, fl);
        load8<float>(se_gate + v * 8, gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) fh[j] = (fh[j] + fl[j]) * gv[j];
        half8 h, l;
        split8(fh, h, l);
        *reinterpret_cast<half8*>(T.xh + r * XROW + v * 8) = h;
        *reinterpret_cast<half8*>(T.xl + r * XROW + v * 8) = l;
    }
    __syncthreads();
}
}  // namespace

__global__ __launch_bounds__(512) void tower_x3_kernel(const X3TowerArgs a) {
    using G = X3Block;
    constexpr int C = G::C, XROW = G::XROW, NJ = G::NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const X3Tiles T = x3_tiles(smem);
    const int b = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    x3_stage_tile(T, a.x + size_t(b) * 64 * C, nullptr, tid);
    __syncthreads();
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const X3TowerBlock& d = a.blocks[blk];
        if (blk > 0 && d.se_kind != 0) x3_se_phase(T, d, reinterpret_cast<float*>(T.t2h), tid);
        const X3Weights W = x3_weights(d.w1pk, d.w1pk_lo, d.w3pk, d.w3pk_lo, d.dwpk, d.cop_pad);
        f32x4 accP[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {                          // the project accumulators start at the BN3 bias of their 4 couts
            const f32x4 bs = *reinterpret_cast<const f32x4*>(d.b3 + (wave * NJ + j) * 16 + lg * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) accP[j][t] = bs;
        }
        x3_chunks(T, W, accP);
        // block epilogue: new stream = x + body(x), split again, in place.  A wave rewrites exactly the columns (its 32 couts) it reads
        // here, and the chunk loop's closing barrier is behind every other read of the tiles.
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co0 = (wave * NJ + j) * 16 + lg * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sq = t * 16 + l15;
                float rh[4], rl[4], v[4];
                load4<half_t>(T.xh + sq * XROW + co0, rh);
                load4<half_t>(T.xl + sq * XROW + co0, rl);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = accP[j][t][r] + (rh[r] + rl[r]);
                half4 h, l;
                split4(v, h, l);
                *reinterpret_cast<half4*>(T.xh + sq * XROW + co0) = h;
                *reinterpret_cast<half4*>(T.xl + sq * XROW + co0) = l;
            }
        }
        __syncthreads();
    }
    // stream -> HBM as float, 32-byte pieces per thread
    float* yb = a.y + size_t(b) * 64 * C;
#pragma unroll 1
    for (int i = tid; i < 64 * (C / 8); i += G::NTHR) {
        const int sq = i / (C / 8), v = i - sq * (C / 8), r = x3_row(sq);
        float fh[8], fl[8];
        load8<half_t>(T.xh + r * XROW + v * 8, fh);
        load8<half_t>(T.xl + r * XROW + v * 8, fl);
#pragma unroll
        for (int j = 0; j < 8; ++j) fh[j] += fl[j];
        store8<float>(yb + size_t(sq) * C + v * 8, fh);
    }
}

// ---- small batches: one block per launch, G workgroups per board (kernels.h: X3SplitArgs) ----
namespace {
// The board tile as the sum of `gin` float images [64][256] (the previous launch's partial sums, added in the order of their index: the same
// bits on every launch and in every workgroup of the board; gin = 1: a plain float tile) -> split tiles.  A thread's loads of one pass are all
// requested before the first add: the images come from HBM / the memory-side cache, and dependent round trips are what a launch of this
// kernel mostly consists of.
// gate: nullptr, or the board's 256 channel gates (LDS): the tile staged is x * gate (a gated block whose gate is known before the board is)
template <int GIN>
__device__ __forceinline__ void x3_stage_tile_sum(const X3Tiles& T, const float* parts, int gin, int tid, const float* gate = nullptr) {
    constexpr int C = X3Block::C, XROW = X3Block::XROW, IT = 64 * (C / 8) / X3Block::NTHR;
    constexpr int NG = GIN > 0 ? GIN : 16;
    constexpr int FLY = GIN > 0 && GIN <= 2 ? 4 : GIN > 0 && GIN <= 5 ? 2 : 1;       // passes requested together (8 NG registers each)
    static_assert(IT % FLY == 0, "passes");
#pragma unroll
    for (int it0 = 0; it0 < IT; it0 += FLY) {
        f32x4 q[FLY][NG][2];
#pragma unroll
        for (int u = 0; u < FLY; ++u) {
            const int i = tid + (it0 + u) * X3Block::NTHR, sq = i / (C / 8), v = i - sq * (C / 8);
            const float* p0 = parts + size_t(sq) * C + v * 8;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (GIN > 0 || g < gin) {
                    const f32x4* p = reinterpret_cast<const f32x4*>(p0 + size_t(g) * 64 * C);
                    q[u][g][0] = p[0];
                    q[u][g][1] = p[1];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FLY; ++u) {
            const int i = tid + (it0 + u) * X3Block::NTHR, sq = i / (C / 8), v = i - sq * (C / 8), r = x3_row(sq);
            f32x4 s0 = q[u][0][0], s1 = q[u][0][1];
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                if (GIN > 0 || g < gin) {
                    s0 += q[u][g][0];
                    s1 += q[u][g][1];
                }
            }
            float f[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (gate) {
                float gv[8];
                load8<float>(gate + v * 8, gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] *= gv[j];
            }
            half8 h, l;
            split8(f, h, l);
            *reinterpret_cast<half8*>(T.xh + r * XROW + v * 8) = h;
            *reinterpret_cast<half8*>(T.xl + r * XROW + v * 8) = l;
        }
    }
}
}  // namespace

__global__ __launch_bounds__(512) void block_x3_split_kernel(const X3SplitArgs a) {
    using G = X3Block;
    constexpr int C = G::C, CK = G::CK, XROW = G::XROW, NJ = G::NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const X3Tiles T = x3_tiles(smem);
    // Workgroup -> (board, share): workgroups go to the eight XCDs round-robin by their linear id; the G workgroups of a board read the same
    // partial sums, which one XCD's L2 then fetches once.  Board b lives on XCD b % 8: id = xcd + 8 j, b = xcd + 8 (j / G), g = j % G; ids
    // beyond the batch leave at once.  (Placement is a matter of speed only.)
    const int G_ = a.G;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int g = jj % G_, b = xcd + 8 * (jj / G_);
    if (b >= a.batch) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const X3TowerBlock& d = a.blk;
    // the first chunk's expand weights are requested before the board is staged: they come from the memory-side cache (every chunk of the
    // net is read by exactly one workgroup per forward) and land while the partial sums are added up
    const X3Weights W = x3_weights(d.w1pk, d.w1pk_lo, d.w3pk, d.w3pk_lo, d.dwpk, d.cop_pad);
    const int n = W.cop_pad / CK;
    const int ch0 = __builtin_amdgcn_readfirstlane(g * n / G_), ch1 = __builtin_amdgcn_readfirstlane((g + 1) * n / G_);
    X3ExpandWindow win;
    x3_expand_window_request(win, W, ch0);
    const bool gate_first = d.se_kind != 0 && a.pool_in != nullptr;     // the launch before left the channel sums of its images: the gate comes BEFORE the board
    X3SeFirstWeights se_first;                                          // a gated block: the gate matrices' first half as well (128 KB per workgroup)
    if (d.se_kind != 0 && !gate_first) x3_se_first_weights_request(se_first, d, tid);
    const float* gate = nullptr;
    if (gate_first) {
        constexpr int GRP = 36;
        float* scratch = reinterpret_cast<float*>(T.t2h);               // x3_se_phase's scratch: mean 8 x 36, hidden 4 x 36, gate 256
        float *se_mean = scratch, *se_h = scratch + 8 * GRP, *se_gate = se_h + 4 * GRP;
        f32x4 wa[16], wb[16];                                           // both gate matrices (256 KB per workgroup), in flight from here
        {
            const f32x4* pa = reinterpret_cast<const f32x4*>(d.se_w1t) + tid;
            const f32x4* pb = reinterpret_cast<const f32x4*>(d.se_kind == 1 ? d.se_w2t : d.se_w1t + size_t(16) * 512 * 4) + tid;
#pragma unroll
            for (int i = 0; i < 16; ++i) wa[i] = pa[i * 512];
#pragma unroll
            for (int i = 0; i < 16; ++i) wb[i] = pb[i * 512];
        }
        if (tid < C) {                                                  // mean of channel tid = the images' channel sums in index order / 64
            const float* pin = a.pool_in + size_t(b) * a.gin * C + tid;
            float ps[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) ps[q] = q < a.gin ? pin[q * C] : 0.f;
            float sum = ps[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) sum += ps[q];
            se_mean[(tid >> 5) * GRP + (tid & 31)] = sum * (1.f / 64.f);
        }
        __syncthreads();
        x3_se_fcs(d, se_mean, se_h, se_gate, wa, wb, tid);
        gate = se_gate;
    }
    {
        const float* parts = a.x_parts + size_t(b) * a.gin * 64 * C;
        switch (a.gin) {                                                // (the usual counts with every load of a pass in flight at once)
            case 1: x3_stage_tile_sum<1>(T, parts, 1, tid, gate); break;
            case 2: x3_stage_tile_sum<2>(T, parts, 2, tid, gate); break;
            case 3: x3_stage_tile_sum<3>(T, parts, 3, tid, gate); break;
            case 4: x3_stage_tile_sum<4>(T, parts, 4, tid, gate); break;
            case 5: x3_stage_tile_sum<5>(T, parts, 5, tid, gate); break;
            case 6: x3_stage_tile_sum<6>(T, parts, 6, tid, gate); break;
            case 7: x3_stage_tile_sum<7>(T, parts, 7, tid, gate); break;
            case 8: x3_stage_tile_sum<8>(T, parts, 8, tid, gate); break;
            case 9: x3_stage_tile_sum<9>(T, parts, 9, tid, gate); break;
            case 10: x3_stage_tile_sum<10>(T, parts, 10, tid, gate); break;
            default: x3_stage_tile_sum<0>(T, parts, a.gin, tid, gate); break;
        }
    }
    __syncthreads();
    if (d.se_kind != 0 && !gate_first) x3_se_phase(T, d, reinterpret_cast<float*>(T.t2h), tid, &se_first);      // every workgroup of the board: the same gate, the same gated tiles
    f32x4 accP[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 bs = g == 0 ? *reinterpret_cast<const f32x4*>(d.b3 + (wave * NJ + j) * 16 + lg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) accP[j][t] = bs;
    }
    if (!(a.dev & 4)) x3_chunks<true>(T, W, accP, ch0, ch1, &win);       // (development bit 4, timing only: no chunk loop)
    // epilogue: this workgroup's part of x + b3 + body(x) (workgroup 0 carries x and b3) -> its own image
    float* yb = a.y_parts + (size_t(b) * G_ + g) * 64 * C;
    f32x4 psum[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) psum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int co0 = (wave * NJ + j) * 16 + lg * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int sq = t * 16 + l15;
            f32x4 v = accP[j][t];
            if (g == 0) {
                float rh[4], rl[4];
                load4<half_t>(T.xh + sq * XROW + co0, rh);
                load4<half_t>(T.xl + sq * XROW + co0, rl);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rh[r] + rl[r];
            }
            *reinterpret_cast<f32x4*>(yb + size_t(x3_square(sq)) * C + co0) = v;
            if (a.pool_out) {
#pragma unroll
                for (int r = 0; r < 4; ++r) psum[j][r] += v[r];
            }
        }
    }
    if (a.pool_out) {                                                   // the channel sums of this image (squares: tiles 0..3 in a lane, then the 16 lanes of the row)
        float* po = a.pool_out + (size_t(b) * G_ + g) * C;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float p[4] = {psum[j][0], psum[j][1], psum[j][2], psum[j][3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] += dpp_mov<0x111>(p[r]);
                p[r] += dpp_mov<0x112>(p[r]);
                p[r] += dpp_mov<0x114>(p[r]);
                p[r] += dpp_mov<0x118>(p[r]);
            }
            if (l15 == 15) *reinterpret_cast<f32x4*>(po + (wave * NJ + j) * 16 + lg * 4) = f32x4{p[0], p[1], p[2], p[3]};
        }
    }
}

// the last block's partial sums -> the float stream [B][64][256] the heads read (the same order of addition as the staging above)
__global__ __launch_bounds__(256) void x3_split_finish_kernel(const float* parts, int gin, float* y, int batch) {
    constexpr int per_board = 64 * X3Block::C / 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < batch * per_board; i += gridDim.x * 256) {
        const int b = i / per_board, e = i - b * per_board;
        const f32x4* p = reinterpret_cast<const f32x4*>(parts + size_t(b) * gin * 64 * X3Block::C) + e;
        f32x4 s = p[0];
        for (int g = 1; g < gin; ++g) s += p[size_t(g) * per_board];
        reinterpret_cast<f32x4*>(y)[i] = s;
    }
}

// ---- the same run of blocks with the waves in two ROLES ----
// Waves 0-3 expand and run the depthwise (EXPAND waves), waves 4-7 project (PROJECT waves); wave w and wave w + 4 share a SIMD.
// A chunk = 128 expanded channels; an EXPAND wave owns two 16-channel tiles of it.  Interval i (one workgroup barrier):
//   EXPAND wave w : E(i) = 32 channels x 64 squares, K = 256 (both tiles share every stream fragment read from LDS) with the depthwise
//                   D(i - 1) of the chunk before INSIDE the MFMA stream (its accumulators were set aside), t2[(i - 1) & 1] written
//   PROJECT wave v: P(i - 2) = 64 couts x 64 squares, K = 128 from t2[i & 1] into its persistent accumulator
// The depthwise inside the MFMA stream: a wave issues two VALU instructions behind each of its own MFMAs at no cost
// (scripts/ubench/mix_kinds.hip: MFMA + 2 VALU 18.5 cycles against 17.3), whereas depthwise work that runs behind the MFMA loop is the
// interval's bare critical path -- the SIMD serves the MFMAs of its two waves one at a time whichever wave they come from (r03k
// timeline: depthwise behind the expand MFMAs 4.4k of the interval's 9.9k cycles, matrix pipe 64 % busy).
// Against the symmetric form (x3_chunks): a stream fragment read from LDS in the project phase feeds twice the MFMAs, and the matrix
// pipe has the PROJECT wave's MFMAs whenever the EXPAND wave's stream thins out.  The price is the pipeline's fill and drain once per
// block (the next block's expand needs this block's output): two intervals run one role only.
// Also measured and NOT kept (profiles/r03/q_*, r_*): the chunk's two tiles in two passes of half an interval each (no second
// accumulator set, but every stream fragment read from LDS twice): 0.72 ms against 0.69 -- the operand reads are a quarter of an
// expand pass (4.6k cycles with, 3.4k without them).
// Measured and NOT kept: L2 warm-up touches of the weight stream two chunks ahead (the float16 tower's trick) made this kernel 9 %
// slower (profiles/r03/h_*) -- alone, its weight stream already runs at 27 TB/s, 80 % of the L2 -> CU peak; s_setprio on the EXPAND
// waves, a barrier that holds the PROJECT waves back until the expand MFMAs are through (profiles/r03/m_*): nothing / slower;
// v_pk_fma_f32 for the depthwise: it does not run in the shadow of MFMAs (mix_kinds.hip: 38.5 cycles for MFMA + 2 of them).
template <int KS>
__global__ __launch_bounds__(512) void tower_x3_roles_kernel(const X3TowerArgs a) {
    static_assert(KS == 3 || KS == 5, "depthwise 3x3 or 5x5 (X3Depthwise / X3Depthwise5)");
    constexpr int REC = KS == 3 ? 256 : 512;                            // floats of depthwise records per 16-channel tile
    using G = X3Block;
    static_assert(G::NE == 1 && G::T2BUF == 2 && G::CK == 128, "the role kernel uses the NE = 1 tile geometry (two t2 buffers of 128 channels)");
    constexpr int C = G::C, CK = G::CK, XROW = G::XROW, TROW = G::TROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const X3Tiles T = x3_tiles(smem);
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool expand_role = wave < 4;
    const int w = wave & 3;
    const uint32_t lane_off = uint32_t(lane) * 16u;
    x3_stage_tile(T, a.x + size_t(b) * 64 * C, nullptr, tid);
    __syncthreads();
    // EXPAND role: the expand weight window (below).  Its first two k-slabs of a block are requested at the END of the block before (unless a
    // gate phase comes in between: that needs the registers), where the EXPAND waves wait 9 k ticks for the PROJECT waves' last chunk and
    // epilogue.  A block's first interval runs E(0) alone -- one wave per SIMD, 384 cycles of MFMAs per k-slab behind a window of two slabs:
    // a fragment is asked for 770 cycles before it is needed and takes ~1500, so the interval runs at the L2's latency (5.9 k ticks for 3.1 k
    // of matrix pipe, 2.3 k with the weight loads switched off; profiles/r06/J_*) -- and asked for at the block's start its first fragments
    // were a whole latency late on top: 5.9 k -> 4.5 k.  (The rest of the first chunk held in registers as well -- the depthwise's are idle
    // in that interval -- spills: 16 to 96 more registers measured, 144 to 572 bytes of scratch, no faster or slower.)
#if defined(CRA_DEVELOPMENT) && defined(CRA_X3_EW)
    constexpr int EW = CRA_X3_EW;
#else
    constexpr int EW = 2;
#endif
    half8 e_h[EW][2], e_l[EW][2];
    bool first_chunk_requested = false;
    // The two roles run the block loop separately (the same barriers in the same order): with one loop around an if / else, a value an EXPAND
    // wave carries from one block into the next -- the first chunk's fragments -- counts as live through the PROJECT branch of the
    // iteration between (the allocator does not know that a wave never changes its role), and that branch has no register to spare.
    if (expand_role) {
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const X3TowerBlock& d = a.blocks[blk];
        if (blk > 0 && d.se_kind != 0) {
            x3_se_phase(T, d, reinterpret_cast<float*>(T.t2h), tid);
            // (never requested in front of a gate phase.  The flag alone does not tell the register allocator: an empty definition of every
            // fragment here ends their live ranges in front of the phase, which needs the registers)
            first_chunk_requested = false;
#pragma unroll
            for (int ne = 0; ne < 2; ++ne) {
#pragma unroll
                for (int q = 0; q < EW; ++q) { asm volatile("" : "=v"(e_h[q][ne])); asm volatile("" : "=v"(e_l[q][ne])); }
            }
        }
        const X3Weights W = x3_weights(d.w1pk, d.w1pk_lo, d.w3pk, d.w3pk_lo, d.dwpk, d.cop_pad);
        const int n = W.cop_pad / CK;
        const int nslab3 = W.cop_pad >> 5;
#ifdef CRA_X3_TRACE
        const bool tracing = (b == 0 || b == 131) && blk == CRA_X3_TRACE;
        int trace_n = 0;
#endif
        // barriers of a block, the same for both roles: one behind each of the halves 0 ... 2n, then the one behind the epilogue
        {
            const bool hi = l15 >= 8;                                  // the tile's second rank (t + 4, x3_row)
            const X3EdgeOffsets edge = x3_edge_offsets(l15);             // a lane on file a / h has no left / right neighbour on the board
            // expand weight window: EW of the 8 k-slabs x 2 channel tiles x (hi, lo); the stream runs on across chunk boundaries: slab s of
            // chunk i sits in slot s % EW and is refilled with the slab EW positions ahead right behind its MFMAs
            auto load_e = [&](int i, int s) {                          // cout tile (16 channels) of (chunk i, wave w, ne) = i * 8 + w * 2 + ne
                if constexpr (X3_ABL & 16) return;
#pragma unroll
                for (int ne = 0; ne < 2; ++ne) {
                    const uint32_t f = uint32_t(i * (CK / 16) + w * 2 + ne) * (C / 32) + uint32_t(s);
                    e_h[s % EW][ne] = x3_frag(W.w1h, lane_off, f);
                    e_l[s % EW][ne] = x3_frag(W.w1l, lane_off, f);
                }
            };
            auto load_first_chunk = [&](const X3Weights& Wx) {          // the window's slots with the first k-slabs of chunk 0's two tiles
                if constexpr (X3_ABL & 16) return;
#pragma unroll
                for (int s = 0; s < EW; ++s)
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne) {
                        const uint32_t f = uint32_t(w * 2 + ne) * (C / 32) + uint32_t(s);
                        e_h[s][ne] = x3_frag(Wx.w1h, lane_off, f);
                        e_l[s][ne] = x3_frag(Wx.w1l, lane_off, f);
                    }
            };
            if constexpr (X3_ABL & 16) {
#pragma unroll
                for (int s = 0; s < EW; ++s)
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne) e_h[s][ne] = e_l[s][ne] = *reinterpret_cast<const half8*>(T.xh + lane * 8);
            }
            if (!first_chunk_requested) load_first_chunk(W);
            float* const my_dws = T.dws + (w * 2) * REC;               // this wave's two record tiles
            f32x4 accE[2][4], accD[2][4];                               // chunk i being expanded / chunk i - 1 in the depthwise
            X3Depthwise dw;
            X3Depthwise5 dw5;
            const X3EdgeOffsets5 edge5 = x3_edge_offsets5(l15);
            // Interval i: E(i) (HASE) with D(i - 1) (HASD) cut into sixteen pieces, two per k-slab: tile 0 in slabs 0-3, tile 1 in 4-7.
            auto interval = [&](auto hase_c, auto hasd_c, int i) {
                constexpr bool HASE = decltype(hase_c)::value, HASD = decltype(hasd_c)::value;
                // stream fragments (B operands) through a ring of four (k-slab, square tile) steps: a step's pair (hi, lo) is requested
                // three steps = 18 MFMAs ahead (a slab's eight pairs double-buffered would be 64 registers beside the depthwise's state)
                half8 ring_h[4], ring_l[4];
                auto read_step = [&](int st) {                          // step st = k-slab st / 4, square tile st % 4
                    if constexpr (X3_ABL & 8) {
                        ring_h[st % 4] = e_h[0][0];
                        ring_l[st % 4] = e_l[0][0];
                    } else {
                        ring_h[st % 4] = *reinterpret_cast<const half8*>(T.xh + ((st & 3) * 16 + l15) * XROW + (st >> 2) * 32 + lg * 8);
                        ring_l[st % 4] = *reinterpret_cast<const half8*>(T.xl + ((st & 3) * 16 + l15) * XROW + (st >> 2) * 32 + lg * 8);
                    }
                };
                f32x4 dw_raw[2][REC / 256];
                if constexpr (HASE) {
                    // depthwise records of chunk i's two tiles: one 16-byte load per lane and tile (x3_chunks); they go to LDS at the end of
                    // the interval, behind the depthwise that still reads chunk i - 1's
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int h2 = 0; h2 < REC / 256; ++h2)
                            dw_raw[ne][h2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W.dw, lane_off, uint32_t(i * CK + (w * 2 + ne) * 16) * uint32_t(REC / 4) + uint32_t(h2) * 1024u, 0));
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int t = 0; t < 4; ++t) accE[ne][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    read_step(0); read_step(1); read_step(2);
                }
                if constexpr (HASD && KS == 3) dw.template load<0>(my_dws, lg, edge);
                half_t* const t2h = T.t2h + ((i - 1) & 1) * 64 * TROW;
                half_t* const t2l = T.t2l + ((i - 1) & 1) * 64 * TROW;
#pragma unroll
                for (int sl = 0; sl < C / 32; ++sl) {
                    const int dt = sl / 4, ph = sl % 4;                 // the depthwise's tile and quarter
                    if constexpr (HASD) {
                        if constexpr (KS == 3) {
                            if (ph == 2) dw.template load<1>(my_dws + dt * 256, lg, edge);
                            if (sl == 4) dw.template load<0>(my_dws + 256, lg, edge);  // (tile 0's last pieces ran in slab 3)
                        } else {                                        // 5x5: one channel of the tile per k-slab
                            if (ph == 0) dw5.template load<0>(my_dws + dt * REC, lg, edge5);
                            if (ph == 1) dw5.template load<1>(my_dws + dt * REC, lg, edge5);
                            if (ph == 2) dw5.template load<2>(my_dws + dt * REC, lg, edge5);
                            if (ph == 3) dw5.template load<3>(my_dws + dt * REC, lg, edge5);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (HASE) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int st = sl * 4 + t;
                            if (st + 3 < 4 * (C / 32)) read_step(st + 3);
#pragma unroll
                            for (int ne = 0; ne < 2; ++ne) {
                                if constexpr ((X3_ABL & 128) != 0) {      // TIMING ONLY (wrong results): the expand GEMM's instruction mix under the mixed split
                                    x3_mfma(e_h[sl % EW][ne], ring_h[st % 4], accE[ne][t], true);
                                    if (sl & 1) x3_mfma8(x3_cat(e_l[0][ne], e_l[1][ne]), x3_cat(ring_l[st % 4], ring_l[(st + 1) % 4]), accE[ne][t], true);
                                    continue;
                                }
                                x3_mfma(e_l[sl % EW][ne], ring_h[st % 4], accE[ne][t], !(X3_ABL & 2));
                                x3_mfma(e_h[sl % EW][ne], ring_l[st % 4], accE[ne][t], !(X3_ABL & 2));
                                x3_mfma(e_h[sl % EW][ne], ring_h[st % 4], accE[ne][t], !(X3_ABL & 2));
                            }
                        }
                        if (sl + EW < C / 32) load_e(i, sl + EW);
                        else load_e(i + 1 < n ? i + 1 : i, sl + EW - C / 32);    // (behind the last chunk: a valid address, no branch in the stretch)
                    }
                    if constexpr (HASD) {
                        if constexpr (KS == 3) {
                            if (ph == 0) dw.template gather<0>(accD[dt], hi);
                            if (ph == 1) { dw.template taps<0>(0, 4); dw.pin_taps(0, 4, 0); }
                            if (ph == 2) dw.template gather<1>(accD[dt], hi);
                        } else {
                            if (ph == 0) { dw5.template gather<0>(accD[dt], hi, 1.f); dw5.template taps<0>(); }
                            if (ph == 1) { dw5.template gather<1>(accD[dt], hi, 1.f); dw5.template taps<1>(); }
                            if (ph == 2) { dw5.template gather<2>(accD[dt], hi, 1.f); dw5.template taps<2>(); }
                            if (ph == 3) { dw5.template gather<3>(accD[dt], hi, 1.f); dw5.template taps<3>(); }
                        }
                        if (ph == 3) {
                            if constexpr (KS == 3) dw.template taps<1>(0, 4);
                            const int cl = (w * 2 + dt) * 16 + lg * 4;  // split -> t2 of chunk i - 1
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                half4 h, l;
                                split4(KS == 3 ? dw.outv[t] : dw5.outv[t], h, l);
                                if constexpr (X3_ABL & 64) {
                                    asm volatile("" ::"v"(h), "v"(l));
                                } else {
                                    *reinterpret_cast<half4*>(t2h + (t * 16 + l15) * TROW + cl) = h;
                                    *reinterpret_cast<half4*>(t2l + (t * 16 + l15) * TROW + cl) = l;
                                }
                            }
                        }
                    }
                    if constexpr (HASE && HASD) {
#pragma unroll
                        for (int r = 0; r < 24; ++r) {                   // behind every MFMA up to four VALU instructions
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (HASE) {                                    // the depthwise is through with chunk i - 1: its records and accumulators make room
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int h2 = 0; h2 < REC / 256; ++h2) *reinterpret_cast<f32x4*>(my_dws + ne * REC + h2 * 256 + lane * 4) = dw_raw[ne][h2];
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int t = 0; t < 4; ++t) accD[ne][t] = accE[ne][t];
                }
            };
            for (int i = 0; i <= n; ++i) {
                const int kk = i - 1;                                    // (stamp bookkeeping)
                X3_STAMP(0);
                if (i == 0) interval(std::true_type{}, std::false_type{}, i);
                else if (i < n) interval(std::true_type{}, std::true_type{}, i);
                else interval(std::false_type{}, std::true_type{}, i);
                X3_STAMP(3);
                if constexpr (!(X3_ABL & 32)) __syncthreads();
                X3_STAMP(4);
            }
            // the next block's first fragments (unless a gate phase comes first): they land while the PROJECT waves finish this block
            first_chunk_requested = false;
            if (blk + 1 < a.nblocks && a.blocks[blk + 1].se_kind == 0) {
                const X3TowerBlock& dn = a.blocks[blk + 1];
                load_first_chunk(x3_weights(dn.w1pk, dn.w1pk_lo, dn.w3pk, dn.w3pk_lo, dn.dwpk, dn.cop_pad));
                first_chunk_requested = true;
            }
            __syncthreads();                                            // the PROJECT waves' block epilogue
            { const int kk = n; X3_STAMP(5); }
        }
    }
    } else {
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const X3TowerBlock& d = a.blocks[blk];
        if (blk > 0 && d.se_kind != 0) x3_se_phase(T, d, reinterpret_cast<float*>(T.t2h), tid);
        const X3Weights W = x3_weights(d.w1pk, d.w1pk_lo, d.w3pk, d.w3pk_lo, d.dwpk, d.cop_pad);
        const int n = W.cop_pad / CK;
        const int nslab3 = W.cop_pad >> 5;
#ifdef CRA_X3_TRACE
        const bool tracing = (b == 0 || b == 131) && blk == CRA_X3_TRACE;
        int trace_n = 0;
#endif
        {
            // project weight window: 2 of a chunk's 4 k-slabs x 4 cout tiles x (hi, lo), running on across chunk boundaries
#if defined(CRA_DEVELOPMENT) && defined(CRA_X3_PW)
            constexpr int PW = CRA_X3_PW, NJ = 4;
#else
            constexpr int PW = 2, NJ = 4;
#endif
            half8 p_h[PW][NJ], p_l[PW][NJ];
            auto load_p = [&](int k, int s2) {                         // cout tile = w * 4 + j, K slab = k * 4 + s2
                if constexpr (X3_ABL & 16) return;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const uint32_t f = uint32_t(w * NJ + j) * uint32_t(nslab3) + uint32_t(k * (CK / 32) + s2);
                    p_h[s2 % PW][j] = x3_frag(W.w3h, lane_off, f);
                    p_l[s2 % PW][j] = x3_frag(W.w3l, lane_off, f);
                }
            };
            if constexpr (X3_ABL & 16) {
#pragma unroll
                for (int s2 = 0; s2 < PW; ++s2)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) p_h[s2][j] = p_l[s2][j] = *reinterpret_cast<const half8*>(T.xl + lane * 8);
            }
            f32x4 accP[NJ][4];                                          // couts (w * 4 + j) * 16 + lg * 4 .. + 3, squares t * 16 + l15
#pragma unroll
            for (int j = 0; j < NJ; ++j) {                              // the accumulators start at the BN3 bias of their 4 couts
                const f32x4 bs = *reinterpret_cast<const f32x4*>(d.b3 + (w * NJ + j) * 16 + lg * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) accP[j][t] = bs;
            }
#pragma unroll
            for (int s2 = 0; s2 < PW; ++s2) load_p(0, s2);
            if constexpr (!(X3_ABL & 32)) {
                __syncthreads();                                        // intervals 0 and 1: chunk 0 is expanded, then run through the depthwise
                __syncthreads();
            }
            for (int kk = 0; kk < n; ++kk) {                            // P(kk) runs in interval kk + 2
                const half_t* const t2h = T.t2h + (kk & 1) * 64 * TROW;
                const half_t* const t2l = T.t2l + (kk & 1) * 64 * TROW;
                X3_STAMP(8);
                half8 bh[2][4], bl[2][4];
                auto read_t2 = [&](int s2, half8 (&h)[4], half8 (&l)[4]) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if constexpr (X3_ABL & 8) {
                            h[t] = p_h[s2 % PW][0];
                            l[t] = p_l[s2 % PW][0];
                        } else {
                            h[t] = *reinterpret_cast<const half8*>(t2h + (t * 16 + l15) * TROW + s2 * 32 + lg * 8);
                            l[t] = *reinterpret_cast<const half8*>(t2l + (t * 16 + l15) * TROW + s2 * 32 + lg * 8);
                        }
                    }
                };
                read_t2(0, bh[0], bl[0]);
#pragma unroll
                for (int s2 = 0; s2 < CK / 32; ++s2) {
                    if (s2 + 1 < CK / 32) read_t2(s2 + 1, bh[(s2 + 1) & 1], bl[(s2 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int t = 0; t < 4; ++t) x3_mfma(p_l[s2 % PW][j], bh[s2 & 1][t], accP[j][t], !(X3_ABL & 4));
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int t = 0; t < 4; ++t) x3_mfma(p_h[s2 % PW][j], bl[s2 & 1][t], accP[j][t], !(X3_ABL & 4));
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int t = 0; t < 4; ++t) x3_mfma(p_h[s2 % PW][j], bh[s2 & 1][t], accP[j][t], !(X3_ABL & 4));
                    if (s2 + PW < CK / 32) load_p(kk, s2 + PW);
                    else load_p(kk + 1 < n ? kk + 1 : kk, s2 + PW - CK / 32);
                    if (s2 == 1) X3_STAMP(9);
                    __builtin_amdgcn_sched_barrier(0);
                }
                X3_STAMP(10);
                if (kk + 1 < n) {
                    if constexpr (!(X3_ABL & 32)) __syncthreads();      // (the last chunk's project phase has no partner: the epilogue's barrier follows)
                }
                X3_STAMP(11);
            }
            // block epilogue: new stream = x + body(x), split again, in place (every EXPAND wave is behind its last read of the tiles: it
            // waits at the barrier below)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int co0 = (w * NJ + j) * 16 + lg * 4;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int sq = t * 16 + l15;
                    float rh[4], rl[4], v[4];
                    load4<half_t>(T.xh + sq * XROW + co0, rh);
                    load4<half_t>(T.xl + sq * XROW + co0, rl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = accP[j][t][r] + (rh[r] + rl[r]);
                    half4 h, l;
                    split4(v, h, l);
                    *reinterpret_cast<half4*>(T.xh + sq * XROW + co0) = h;
                    *reinterpret_cast<half4*>(T.xl + sq * XROW + co0) = l;
                }
            }
            { const int kk = n; X3_STAMP(12); }
            __syncthreads();
            { const int kk = n; X3_STAMP(13); }
        }
    }
    }
    // stream -> HBM as float, 32-byte pieces per thread
    float* yb = a.y + size_t(b) * 64 * C;
#pragma unroll 1
    for (int i = tid; i < 64 * (C / 8); i += G::NTHR) {
        const int sq = i / (C / 8), v = i - sq * (C / 8), r = x3_row(sq);
        float fh[8], fl[8];
        load8<half_t>(T.xh + r * XROW + v * 8, fh);
        load8<half_t>(T.xl + r * XROW + v * 8, fl);
#pragma unroll
        for (int j = 0; j < 8; ++j) fh[j] += fl[j];
        store8<float>(yb + size_t(sq) * C + v * 8, fh);
    }
}

// ================================================================================================================
// Precision float16p8: the two-role tower with both GEMMs on the mixed split
// ================================================================================================================
// What changes against tower_x3_roles_kernel (measurements: profiles/NOTES.md round 4):
//  * Both GEMMs: per 64 k two f16 MFMAs (main term hi x hi) and ONE v_mfma_f32_16x16x128_f8f6f4 (e5m2 operands) on [hi8 | lo8] x [w_lo8 ; w_hi8]
//    (both cross terms) instead of six f16 MFMAs.  The interval is the matrix pipe time of a SIMD's two waves plus what the EXPAND wave issues
//    beyond its MFMAs' shadow: both shrink.  The 8-bit operand bytes are the HIGH BYTES of the float16x3 split's f16 pairs (split4_b8: a byte
//    permute; the e4m3 form of the first build paid for its conversions what the matrix pipe gained, sets j and k).
//  * The tiles in LDS hold the operand forms only: the stream tile xh = rne_f16(x) + a byte row [hi8, 256 B | lo8, 256 B] where float16x3 keeps
//    the lo half; the depthwise output t2h + a byte row [hi8, 128 B | lo8, 128 B] (272-byte pitch) where float16x3 keeps t2l.
//  * The residual stream itself lives in the PROJECT waves' accumulators: wave v holds x of its 64 couts x 64 squares in f32 (exact, where
//    float16x3 rebuilds x = hi + lo from LDS to 2^-22), the project sums of a block are accumulated ON it -- in the project weights' scale:
//    x := (x + b3) * 2^p in front of the block, x := x * 2^-p behind it, both exact -- and the block epilogue only writes the operand forms of
//    the new x.  SE gates: squeeze from the registers, gate as before, x *= gate in the registers.
// The roles are separated at the top level (the PROJECT waves' 64 registers of x must not be live in the EXPAND waves' code).
namespace {
// mean[c] (LDS scratch, written by the PROJECT waves) -> gate[c] in LDS: the middle of x3_se_phase, every thread of the workgroup.
// Ends behind a barrier with se_gate valid.
__device__ __forceinline__ void x3_se_gate_from_mean(const X3TowerBlock& d, float* scratch, int tid) {
    constexpr int GRP = 36;
    float* se_mean = scratch;              // [8][36]
    float* se_h = scratch + 8 * GRP;       // [4][36]
    float* se_gate = se_h + 4 * GRP;       // [256]
    f32x4 wa[16], wb[16];
    auto load_thread_weights = [&](const float* base, f32x4 (&dst)[16]) {
        const f32x4* pk = reinterpret_cast<const f32x4*>(base) + tid;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i] = pk[i * 512];
    };
    load_thread_weights(d.se_w1t, wa);
    load_thread_weights(d.se_kind == 1 ? d.se_w2t : d.se_w1t + size_t(16) * 512 * 4, wb);
    __syncthreads();                                                    // the means are in
    auto dot32 = [](const f32x4 (&w)[16], const float* v, float& s0, float& s1) {
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(v + 4 * k4);
            s0 = fmaf(w[2 * k4][0], m[0], s0); s1 = fmaf(w[2 * k4][1], m[0], s1);
            s0 = fmaf(w[2 * k4][2], m[1], s0); s1 = fmaf(w[2 * k4][3], m[1], s1);
            s0 = fmaf(w[2 * k4 + 1][0], m[2], s0); s1 = fmaf(w[2 * k4 + 1][1], m[2], s1);
            s0 = fmaf(w[2 * k4 + 1][2], m[3], s0); s1 = fmaf(w[2 * k4 + 1][3], m[3], s1);
        }
    };
    if (d.se_kind == 1) {
        {
            const int j2 = tid >> 3, kq = tid & 7;
            float s0 = 0.f, s1 = 0.f;
            dot32(wa, se_mean + kq * GRP, s0, s1);
            s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
            s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
            s0 += dpp_mov<0x114>(s0); s1 += dpp_mov<0x114>(s1);
            if (kq == 7) {
                float* h = se_h + (j2 >> 4) * GRP + 2 * (j2 & 15);
                h[0] = fmaxf(s0, 0.f);
                h[1] = fmaxf(s1, 0.f);
            }
        }
        __syncthreads();
        {
            const int c2 = tid >> 2, kq = tid & 3;
            float s0 = 0.f, s1 = 0.f;
            dot32(wb, se_h + kq * GRP, s0, s1);
            s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
            s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
            if (kq == 3) {
                se_gate[2 * c2] = hard_sigmoid(s0);
                se_gate[2 * c2 + 1] = hard_sigmoid(s1);
            }
        }
    } else {
        const int c2 = tid >> 2, kq = tid & 3;
        float s0 = 0.f, s1 = 0.f;
        dot32(wa, se_mean + (2 * kq) * GRP, s0, s1);
        dot32(wb, se_mean + (2 * kq + 1) * GRP, s0, s1);
        s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
        s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
        if (kq == 3) {
            se_gate[2 * c2] = hard_sigmoid(d.se_b[2 * c2] + s0);
            se_gate[2 * c2 + 1] = hard_sigmoid(d.se_b[2 * c2 + 1] + s1);
        }
    }
    __syncthreads();
}
}  // namespace

// byte position of channel c inside a byte row of the residual stream: the 16-byte pieces of a 64-channel step in the order A0 B0 A1 B1
__device__ __forceinline__ int x8_pos(int c) { return (c & ~0x30) | ((c & 0x10) << 1) | ((c & 0x20) >> 1); }

template <int KS>
__global__ __launch_bounds__(512) void tower_p8_kernel(const X3TowerArgs a) {
    static_assert(KS == 3 || KS == 5, "depthwise 3x3 or 5x5");
    constexpr int REC = KS == 3 ? 256 : 512;                            // floats of depthwise records per 16-channel tile (X3Depthwise / X3Depthwise5)
    using G = X3Block;
    static_assert(G::NE == 1 && G::T2BUF == 2 && G::CK == 128, "the role kernels use the NE = 1 tile geometry");
    constexpr int C = G::C, CK = G::CK, XROW = G::XROW, TROW = G::TROW, X8ROW = XROW * 2, T8ROW = 288, T8LO = 144, NJ = 4;
    static_assert(T8ROW <= TROW * 2 && T8LO + CK <= T8ROW, "the byte rows of t2 live where float16x3 keeps t2l");
    // t2's rows (f16 tile and byte tile, both at a 288-byte pitch) are SWIZZLED: bit 0 of a row's 16-byte slot index is XORed with bit 2, and
    // bit 2 with bit 3, of the row's index inside its square tile (g0 / g2 below; both slot bits are lane constants or wave constants on
    // the storing side, so a lane keeps ONE column per tile and the second channel tile of a wave is an immediate away), and the byte tile
    // holds a 64-channel step as the pieces A0 B0 A1 B1 like the stream's byte rows (x8_pos).  With that the PROJECT waves' ds_read_b128
    // put every lane group on 16 different slots (4 LDS cycles; was 8 for the bytes at the 272-byte pitch, and 8 for the f16 tile at any
    // pitch that serves the stores), and the EXPAND waves' 8- and 4-byte stores stay at the two lanes per bank that their row-per-lane
    // mapping allows (scripts/studies/lds_bank_model.py: 384 -> 256 LDS cycles per interval and role pair for t2).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const X3Tiles T = x3_tiles(smem);
    char* const x8 = reinterpret_cast<char*>(T.xl);                    // [64][528 B]: hi8 bytes [0, 256), lo8 bytes [272, 528) of a row
    char* const t28 = reinterpret_cast<char*>(T.t2l);                  // [2][64][288 B]: hi8 bytes [0, 128), lo8 bytes [144, 272) of a row, slots swizzled (above)
    float* const se_scratch = reinterpret_cast<float*>(T.t2h);         // idle between blocks
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);                        // MODE.FP16_OVFL: conversions to f16 clamp instead of overflowing
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & 3;
    const uint32_t lane_off = uint32_t(lane) * 16u;
    const int g0 = (l15 >> 2) & 1, g2 = (l15 >> 3) & 1;                 // t2's slot swizzle of this lane's rows

    if (wave < 4) {
        // =================================================== EXPAND waves ===================================================
        const bool hi = l15 >= 8;
        // where this lane's four channels of channel tile w * 2 (dt = 0) go inside a t2 row: f16 tile in halves (slot = channel / 8 =
        // w * 4 + dt * 2 + lg / 2), byte tile in bytes (the tile's 16 bytes are slot (w / 2) * 4 + dt * 2 + w % 2 of the A0 B0 A1 B1
        // order); slots swizzled; tile w * 2 + 1 (dt = 1) is two slots further in both
        const int t2h_col = ((w ^ g2) * 4 + ((lg >> 1) ^ g0)) * 8 + (lg & 1) * 4;
        const int t2b_col = (((w >> 1) ^ g2) * 4 + ((w & 1) ^ g0)) * 16 + lg * 4;
        const X3EdgeOffsets edge = x3_edge_offsets(l15);
        const X3EdgeOffsets5 edge5 = x3_edge_offsets5(l15);
        // this lane's operand rows in the stream tile (square tile 0): even / odd k-slabs of xh, even / odd 64-k steps of the byte rows -- the
        // stream tile is swizzled like t2 (same 32-byte-mod-256 pitch), which takes the block epilogue's stores from four lanes per bank to two
        const half_t* const xh_even = T.xh + l15 * XROW + g2 * 32 + (lg ^ g0) * 8;
        const half_t* const xh_odd = T.xh + l15 * XROW + (g2 ^ 1) * 32 + (lg ^ g0) * 8;
        const char* const x8_even = x8 + l15 * X8ROW + (lg >> 1) * 272 + g2 * 64 + (((lg & 1) ^ g0) << 4);
        const char* const x8_odd = x8 + l15 * X8ROW + (lg >> 1) * 272 + (g2 ^ 1) * 64 + (((lg & 1) ^ g0) << 4);
        __syncthreads();                                                // the PROJECT waves have written block 0's operand tiles
        for (int blk = 0; blk < a.nblocks; ++blk) {
            const X3TowerBlock& d = a.blocks[blk];
            if (d.se_kind != 0) {                                       // (also the run's first block: its gate is computed in this launch)
                x3_se_gate_from_mean(d, se_scratch, tid);               // (the squeeze and the rescaling are the PROJECT waves')
                __syncthreads();                                        // the gated operand tiles are written
            }
            const X3Weights W = x3_weights(d.w1pk, d.w1pk_lo, d.w3pk, d.w3pk_lo, d.dwpk, d.cop_pad);
            const int n = W.cop_pad / CK;
            const float e_inv = d.w1_inv;
#ifdef CRA_X3_TRACE
            const bool tracing = (b == 0 || b == 131) && blk == CRA_X3_TRACE;
            int trace_n = 0;
#endif
            // window: f16 fragments of two k-slabs (slot = slab parity) and the e5m2 fragments of one 64-k step, for the wave's two channel tiles
            half8 e_h[2][2];
            i32x8_x3 e_8[2];
            auto load_eh = [&](int i, int s) {                          // cout tile of (chunk i, wave w, ne) = i * 8 + w * 2 + ne
                if constexpr (X3_ABL & 16) return;
#pragma unroll
                for (int ne = 0; ne < 2; ++ne) e_h[s & 1][ne] = x3_frag(W.w1h, lane_off, uint32_t(i * (CK / 16) + w * 2 + ne) * (C / 32) + uint32_t(s));
            };
            auto load_e8 = [&](int i, int J) {                          // a lane's 32 bytes: its 16 of "slab" 2 J and its 16 of "slab" 2 J + 1 of the 8-bit image
#pragma unroll
                for (int ne = 0; ne < 2; ++ne) {
                    const uint32_t f = uint32_t(i * (CK / 16) + w * 2 + ne) * (C / 32) + uint32_t(2 * J);
                    if constexpr (X3_ABL & (16 | 1024)) continue;
                    if constexpr (X3_ABL & 512) {
                        const half8 piece = x3_frag(W.w1l, lane_off, f);
                        e_8[ne] = x3_cat(piece, piece);
                        continue;
                    }
                    e_8[ne] = x3_cat(x3_frag(W.w1l, lane_off, f), x3_frag(W.w1l, lane_off, f + 1));
                }
            };
            if constexpr (X3_ABL & (16 | 1024)) {                       // (timing switches: the windows hold whatever LDS holds)
#pragma unroll
                for (int ne = 0; ne < 2; ++ne) {
                    const half8 any = *reinterpret_cast<const half8*>(T.xh + lane * 8 + ne * 512);
                    e_8[ne] = x3_cat(any, any);
                    if constexpr (X3_ABL & 16) e_h[0][ne] = e_h[1][ne] = any;
                }
            }
            load_eh(0, 0);
            load_eh(0, 1);
            load_e8(0, 0);
            float* const my_dws = T.dws + (w * 2) * REC;
            f32x4 accE[2][4], accD[2][4];
            X3Depthwise dw;
            X3Depthwise5 dw5;
            // Interval i: E(i) (HASE) with D(i - 1) (HASD) in sixteen pieces, two per k-slab, as in tower_x3_roles_kernel.  A k-slab issues its 8
            // f16 MFMAs; an ODD slab then the 8 e5m2 MFMAs of its 64-k step.
            auto interval = [&](auto hase_c, auto hasd_c, int i) {
                constexpr bool HASE = decltype(hase_c)::value, HASD = decltype(hasd_c)::value;
                half8 ring_h[4];                                        // f16 operand of step st = slab * 4 + square tile, requested 3 steps ahead
                i32x8_x3 ring_8[3];                                     // e5m2 operand of step q = (64-k step) * 4 + square tile, requested 2 steps ahead
                auto read_h = [&](int st) {                             // slot slab * 4 + lg of the row, swizzled like t2: (slab ^ g2) * 4 + (lg ^ g0)
                    const half_t* const pe = (st >> 2) & 1 ? xh_odd : xh_even;
                    ring_h[st % 4] = *reinterpret_cast<const half8*>(pe + (st & 3) * 16 * XROW + (st >> 3) * 64);
                };
                auto read_8 = [&](int q) {                              // lane group lg: 0, 1 = hi8 of k [0, 32), [32, 64) of the step; 2, 3 = lo8 of the same
                    // the byte rows of x hold a 64-k step as the 16-byte pieces A0 B0 A1 B1 (A = k [0, 32), B = k [32, 64): x8_pos): the
                    // two lane groups of a pair then read NEIGHBOURING slots, which with the 544-byte pitch puts the 16 lanes of every
                    // ds_read_b128 group on 16 different slots (linear A0 A1 B0 B1: two lanes per slot, 8 LDS cycles instead of 4 -- a
                    // third of the kernel's bank-conflict cycles, scripts/studies/lds_bank_model.py)
                    const char* pp = ((q >> 2) & 1 ? x8_odd : x8_even) + (q & 3) * 16 * X8ROW + (q >> 3) * 128;      // (and swizzled like t2: x8_even / x8_odd)
                    ring_8[q % 3] = x3_cat(*reinterpret_cast<const half8*>(pp), *reinterpret_cast<const half8*>(pp + 32));
                };
                f32x4 dw_raw[2][REC / 256];
                const int inext = i + 1 < n ? i + 1 : i;                // (behind the last chunk: a valid address, no branch in the stretch)
                if constexpr (HASE) {
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int h2 = 0; h2 < REC / 256; ++h2)
                            dw_raw[ne][h2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W.dw, lane_off, uint32_t(i * CK + (w * 2 + ne) * 16) * uint32_t(REC / 4) + uint32_t(h2) * 1024u, 0));
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int t = 0; t < 4; ++t) accE[ne][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    read_h(0); read_h(1); read_h(2);
                    read_8(0); read_8(1);
                }
                if constexpr (HASD && KS == 3) dw.template load<0>(my_dws, lg, edge);
                half_t* const t2h = T.t2h + ((i - 1) & 1) * 64 * TROW;
                char* const t2b = t28 + ((i - 1) & 1) * 64 * T8ROW;
#pragma unroll
                for (int sl = 0; sl < C / 32; ++sl) {
                    const int dt = sl / 4, ph = sl % 4;
                    X3_STAMP_SLAB(i, sl);
                    if constexpr (HASD && KS == 3) {
                        if (ph == 2) dw.template load<1>(my_dws + dt * 256, lg, edge);
                        if (sl == 4) dw.template load<0>(my_dws + 256, lg, edge);
                    }
                    if constexpr (HASD && KS == 5) {                    // one channel of the tile per k-slab: its 27 records, then (behind the slab's MFMAs) gather and taps
                        if (ph == 0) dw5.template load<0>(my_dws + dt * REC, lg, edge5);
                        if (ph == 1) dw5.template load<1>(my_dws + dt * REC, lg, edge5);
                        if (ph == 2) dw5.template load<2>(my_dws + dt * REC, lg, edge5);
                        if (ph == 3) dw5.template load<3>(my_dws + dt * REC, lg, edge5);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (HASE) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int st = sl * 4 + t;
                            if (st + 3 < 4 * (C / 32)) read_h(st + 3);
#pragma unroll
                            for (int ne = 0; ne < 2; ++ne) x3_mfma(e_h[sl & 1][ne], ring_h[st % 4], accE[ne][t], !(X3_ABL & 2));
                            if (sl & 1) {
                                const int q = (sl >> 1) * 4 + t;
                                if (q + 2 < 4 * (C / 64)) read_8(q + 2);
#pragma unroll
                                for (int ne = 0; ne < 2; ++ne) x3_mfma8(e_8[ne], ring_8[q % 3], accE[ne][t], !(X3_ABL & 2));
                            }
                        }
                        if (sl + 2 < C / 32) load_eh(i, sl + 2); else load_eh(inext, sl + 2 - C / 32);
                        if (sl & 1) { if ((sl >> 1) + 1 < C / 64) load_e8(i, (sl >> 1) + 1); else load_e8(inext, 0); }
                    }
                    if constexpr (HASD) {
                        if constexpr (KS == 3) {
                            if (ph == 0) dw.template gather<0, true>(accD[dt], hi, 0, 2, e_inv);
                            if (ph == 1) { dw.template taps<0>(0, 4); dw.pin_taps(0, 4, 0); }
                            if (ph == 2) dw.template gather<1, true>(accD[dt], hi, 0, 2, e_inv);
                            if (ph == 3) dw.template taps<1>(0, 4);
                        } else {
                            if (ph == 0) { dw5.template gather<0>(accD[dt], hi, e_inv); dw5.template taps<0>(); }
                            if (ph == 1) { dw5.template gather<1>(accD[dt], hi, e_inv); dw5.template taps<1>(); }
                            if (ph == 2) { dw5.template gather<2>(accD[dt], hi, e_inv); dw5.template taps<2>(); }
                            if (ph == 3) { dw5.template gather<3>(accD[dt], hi, e_inv); dw5.template taps<3>(); }
                        }
                        if (ph == 3) {                                  // split -> t2 of chunk i - 1: the f16 hi and the two byte rows (t2h_col / t2b_col)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                half4 h;
                                uint32_t h8, l8;
                                split4_b8(KS == 3 ? dw.outv[t] : dw5.outv[t], h, h8, l8);
                                if constexpr ((X3_ABL & 64) != 0) {
                                    asm volatile("" ::"v"(h), "v"(h8), "v"(l8));
                                    continue;
                                }
                                *reinterpret_cast<half4*>(t2h + (t * 16 + l15) * TROW + t2h_col + dt * 16) = h;
                                *reinterpret_cast<uint32_t*>(t2b + (t * 16 + l15) * T8ROW + t2b_col + dt * 32) = h8;
                                *reinterpret_cast<uint32_t*>(t2b + (t * 16 + l15) * T8ROW + T8LO + t2b_col + dt * 32) = l8;
                            }
                        }
                    }
                    if constexpr (HASE && HASD) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {                   // behind every MFMA up to X3_SGB_VALU (5x5: twelve) VALU instructions
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, KS == 3 ? X3_SGB_VALU : 12, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (HASE) {                                    // the depthwise is through with chunk i - 1: its records and accumulators make room
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int h2 = 0; h2 < REC / 256; ++h2) *reinterpret_cast<f32x4*>(my_dws + ne * REC + h2 * 256 + lane * 4) = dw_raw[ne][h2];
#pragma unroll
                    for (int ne = 0; ne < 2; ++ne)
#pragma unroll
                        for (int t = 0; t < 4; ++t) accD[ne][t] = accE[ne][t];
                }
            };
            for (int i = 0; i <= n; ++i) {
                const int kk = i - 1;                                    // (stamp bookkeeping)
                X3_STAMP(0);
                if (i == 0) interval(std::true_type{}, std::false_type{}, i);
                else if (i < n) interval(std::true_type{}, std::true_type{}, i);
                else interval(std::false_type{}, std::true_type{}, i);
                X3_STAMP(3);
                __syncthreads();
                X3_STAMP(4);
            }
            __syncthreads();                                            // the PROJECT waves' block epilogue
            { const int kk = n; X3_STAMP(5); }
        }
        return;
    }

    // =================================================== PROJECT waves ===================================================
    // the residual stream of this wave's 64 couts: accX[j][t][r] = x[square of tile row t * 16 + l15][cout (w * 4 + j) * 16 + lg * 4 + r]
    f32x4 accX[NJ][4];
    {
        const float* xb = a.x + size_t(b) * 64 * C;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                accX[j][t] = *reinterpret_cast<const f32x4*>(xb + size_t(x3_square(t * 16 + l15)) * C + (w * NJ + j) * 16 + lg * 4);
    }
    auto write_tiles = [&]() {                                          // the operand forms of x: xh, hi8, lo8 (this wave's 64 channel columns)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co0 = (w * NJ + j) * 16 + lg * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rr = t * 16 + l15;
                float v[4] = {accX[j][t][0], accX[j][t][1], accX[j][t][2], accX[j][t][3]};
                half4 h;
                uint32_t h8, l8;
                split4_b8(v, h, h8, l8);
                // xh: slot (w * 4 + j) * 2 + lg / 2 of the row, bit 0 ^ g0, bit 2 ^ g2; byte rows: slot w * 4 + (j & 1) * 2 + j / 2 of the
                // A0 B0 A1 B1 order (x8_pos(co0) / 16), the same two bits swizzled
                const int ch = ((w * 4 + j) * 2 + (lg >> 1)) ^ g0 ^ (g2 << 2);
                const int cb = ((w * 4 + (j & 1) * 2 + (j >> 1)) ^ g0 ^ (g2 << 2)) * 16 + lg * 4;
                *reinterpret_cast<half4*>(T.xh + rr * XROW + ch * 8 + (lg & 1) * 4) = h;
                *reinterpret_cast<uint32_t*>(x8 + rr * X8ROW + cb) = h8;
                *reinterpret_cast<uint32_t*>(x8 + rr * X8ROW + 272 + cb) = l8;
            }
        }
    };
    // this lane's operand rows in t2 (buffer 0, square tile 0): the f16 tile's even / odd k-slabs and the byte tile's two 64-k steps (swizzle)
    const half_t* const t2h_even = T.t2h + l15 * TROW + g2 * 32 + (lg ^ g0) * 8;
    const half_t* const t2h_odd = T.t2h + l15 * TROW + (g2 ^ 1) * 32 + (lg ^ g0) * 8;
    const char* const t2b_j0 = t28 + l15 * T8ROW + (lg >> 1) * T8LO + g2 * 64 + (((lg & 1) ^ g0) << 4);
    const char* const t2b_j1 = t28 + l15 * T8ROW + (lg >> 1) * T8LO + (g2 ^ 1) * 64 + (((lg & 1) ^ g0) << 4);
    if (a.blocks[0].se_kind == 0) write_tiles();                       // (a gated first block writes them behind its gate)
    __syncthreads();
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const X3TowerBlock& d = a.blocks[blk];
        if (d.se_kind != 0) {
            // squeeze (AdaptiveAvgPool2d) from the registers: sum over the four square tiles, then over the 16 lanes of the row
            constexpr int GRP = 36;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float sum[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] = (accX[j][0][r] + accX[j][1][r]) + (accX[j][2][r] + accX[j][3][r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sum[r] += dpp_mov<0x111>(sum[r]);
                    sum[r] += dpp_mov<0x112>(sum[r]);
                    sum[r] += dpp_mov<0x114>(sum[r]);
                    sum[r] += dpp_mov<0x118>(sum[r]);                    // lane 15 of the row holds the row's sum
                }
                if (l15 == 15) {
                    const int c = (w * NJ + j) * 16 + lg * 4;            // channel c at (c / 32) * 36 + c % 32
#pragma unroll
                    for (int r = 0; r < 4; ++r) se_scratch[((c + r) >> 5) * GRP + ((c + r) & 31)] = sum[r] * (1.f / 64.f);
                }
            }
            x3_se_gate_from_mean(d, se_scratch, tid);
            const float* se_gate = se_scratch + 12 * GRP;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {                              // x := x * gate (the residual uses the gated x, builder_util.py:473-475)
                const f32x4 g = *reinterpret_cast<const f32x4*>(se_gate + (w * NJ + j) * 16 + lg * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) accX[j][t] *= g;
            }
            write_tiles();
            __syncthreads();
        }
        const X3Weights W = x3_weights(d.w1pk, d.w1pk_lo, d.w3pk, d.w3pk_lo, d.dwpk, d.cop_pad);
        const int n = W.cop_pad / CK;
        const int nslab3 = W.cop_pad >> 5;
#ifdef CRA_X3_TRACE
        const bool tracing = (b == 0 || b == 131) && blk == CRA_X3_TRACE;
        int trace_n = 0;
#endif
        // window: f16 fragments of two K slabs (slot = slab parity) and the e5m2 fragments of one 64-k step, for the wave's four cout tiles
        half8 p_h[2][NJ];
        i32x8_x3 p_8[NJ];
        auto load_ph = [&](int k, int s2) {                            // cout tile = w * 4 + j, K slab = k * 4 + s2
            if constexpr (X3_ABL & 16) return;
#pragma unroll
            for (int j = 0; j < NJ; ++j) p_h[s2 & 1][j] = x3_frag(W.w3h, lane_off, uint32_t(w * NJ + j) * uint32_t(nslab3) + uint32_t(k * (CK / 32) + s2));
        };
        auto load_p8 = [&](int k, int J) {                             // a lane's 32 bytes of the 64-k step J of chunk k: "slabs" 2 J and 2 J + 1 of the 8-bit image
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const uint32_t f = uint32_t(w * NJ + j) * uint32_t(nslab3) + uint32_t(k * (CK / 32) + 2 * J);
                if constexpr (X3_ABL & (16 | 1024)) continue;
                if constexpr (X3_ABL & 512) {
                    const half8 piece = x3_frag(W.w3l, lane_off, f);
                    p_8[j] = x3_cat(piece, piece);
                    continue;
                }
                p_8[j] = x3_cat(x3_frag(W.w3l, lane_off, f), x3_frag(W.w3l, lane_off, f + 1));
            }
        };
        if constexpr (X3_ABL & (16 | 1024)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const half8 any = *reinterpret_cast<const half8*>(T.xh + lane * 8 + j * 512);
                p_8[j] = x3_cat(any, any);
                if constexpr (X3_ABL & 16) p_h[0][j] = p_h[1][j] = any;
            }
        }
        {   // the residual stream enters the project weights' scale: x := (x + b3) * 2^p; the project sums of the block are accumulated on it
            const float ps = d.w3_scale;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 bs = *reinterpret_cast<const f32x4*>(d.b3 + (w * NJ + j) * 16 + lg * 4) * ps;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accX[j][t][r] = fmaf(accX[j][t][r], ps, bs[r]);
            }
        }
        load_ph(0, 0);
        load_ph(0, 1);
        load_p8(0, 0);
        __syncthreads();                                                // intervals 0 and 1: chunk 0 is expanded, then run through the depthwise
        __syncthreads();
        for (int kk = 0; kk < n; ++kk) {                                // P(kk) runs in interval kk + 2
            X3_STAMP(8);
            half8 bh[2][4];
            i32x8_x3 b8[4];
            auto read_h = [&](int s2) {                                 // slot s2 * 4 + lg of the row, swizzled: (s2 ^ g2) * 4 + (lg ^ g0)
                const half_t* const pe = s2 & 1 ? t2h_odd : t2h_even;
#pragma unroll
                for (int t = 0; t < 4; ++t) bh[s2 & 1][t] = *reinterpret_cast<const half8*>(pe + (kk & 1) * 64 * TROW + t * 16 * TROW + (s2 >> 1) * 64);
            };
            auto read_8 = [&](int J) {                                  // lane group lg: 0, 1 = hi8 of k [0, 32), [32, 64) of the step; 2, 3 = lo8 of the same
                const char* const pj = J ? t2b_j1 : t2b_j0;              // the lane's 32 bytes: pieces (lg & 1) and 2 + (lg & 1) of the step's A0 B0 A1 B1, swizzled
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const char* pp = pj + (kk & 1) * 64 * T8ROW + t * 16 * T8ROW;
                    b8[t] = x3_cat(*reinterpret_cast<const half8*>(pp), *reinterpret_cast<const half8*>(pp + 32));
                }
            };
            const int knext = kk + 1 < n ? kk + 1 : kk;                 // (behind the last chunk: a valid address, no branch in the stretch)
            read_h(0);
            read_8(0);
#pragma unroll
            for (int s2 = 0; s2 < CK / 32; ++s2) {
                if (s2 + 1 < CK / 32) read_h(s2 + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) x3_mfma(p_h[s2 & 1][j], bh[s2 & 1][t], accX[j][t], !(X3_ABL & 4));
                if (s2 & 1) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int t = 0; t < 4; ++t) x3_mfma8(p_8[j], b8[t], accX[j][t], !(X3_ABL & 4));
                    __builtin_amdgcn_sched_barrier(0);
                    if (s2 == 1) { read_8(1); load_p8(kk, 1); } else load_p8(knext, 0);
                }
                if (s2 + 2 < CK / 32) load_ph(kk, s2 + 2); else load_ph(knext, s2 + 2 - CK / 32);
                __builtin_amdgcn_sched_barrier(0);
            }
            X3_STAMP(10);
            if (kk + 1 < n) __syncthreads();
            X3_STAMP(11);
        }
        {   // back out of the project weights' scale
            const float pi = d.w3_inv;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) accX[j][t] *= pi;
        }
        // block epilogue: accX IS the new stream; its operand forms go to LDS unless the next block gates it first (the SE phase writes them then)
        const bool next_gated = blk + 1 < a.nblocks && a.blocks[blk + 1].se_kind != 0;
        if (!next_gated) write_tiles();
        { const int kk = n; X3_STAMP(12); }
        __syncthreads();
        { const int kk = n; X3_STAMP(13); }
    }
    // the stream -> HBM straight from the registers (64-byte pieces per square and lane group)
    float* yb = a.y + size_t(b) * 64 * C;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<f32x4*>(yb + size_t(x3_square(t * 16 + l15)) * C + (w * NJ + j) * 16 + lg * 4) = accX[j][t];
}

void init_x3_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(X3Block::lds_bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(X3Block::lds_bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block_x3_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(X3Block::lds_bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_x3_roles_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, int(X3Block::lds_bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_x3_roles_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, int(X3Block::lds_bytes + 8192));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_p8_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, int(X3Block::lds_bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_p8_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, int(X3Block::lds_bytes + 8192));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(ConvX3::lds_bytes));
    const int conv_p8_lds = int(ConvP8::lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p8_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, conv_p8_lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p8_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, conv_p8_lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p8_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, conv_p8_lds);
}
int block_x3_chunk_channels() { return X3Block::CK; }

void launch_block_x3(const BlockArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(block_x3_kernel, dim3(a.batch), dim3(X3Block::NTHR), X3Block::lds_bytes, s, a);
}
void launch_block_x3_split(const X3SplitArgs& a, hipStream_t s) {
    const int n = a.blk.cop_pad / X3Block::CK;
    if (a.G < 1 || a.G > n || a.G > 16 || a.gin < 1 || a.gin > 16) throw std::invalid_argument("block_x3_split: 1 <= G <= min(chunks, 16), 1 <= gin <= 16");
    hipLaunchKernelGGL(block_x3_split_kernel, dim3(8 * a.G * ((a.batch + 7) / 8)), dim3(X3Block::NTHR), X3Block::lds_bytes, s, a);
}
void launch_x3_split_finish(const float* parts, int gin, float* y, int batch, hipStream_t s) {
    const int n4 = batch * 64 * X3Block::C / 4;
    hipLaunchKernelGGL(x3_split_finish_kernel, dim3(std::min(1024, (n4 + 255) / 256)), dim3(256), 0, s, parts, gin, y, batch);
}
void launch_tower_x3(const X3TowerArgs& a, hipStream_t s) {
    // CRA_X3_TOWER=symmetric (read when the net is made, rise_net.h DevSwitches): every wave runs all three phases (A/B reference);
    // default: the two-role kernel.  The two add up every output in the same order: same bits.
    const bool symmetric = a.symmetric != 0;
    if (a.p8) {
        if (symmetric) throw std::invalid_argument("Precision float16p8 runs the two-role tower only");
        if (a.ks == 5) hipLaunchKernelGGL(tower_p8_kernel<5>, dim3(a.batch), dim3(X3Block::NTHR), X3Block::lds_bytes + 8192, s, a);    // (2 KiB of records per tile)
        else hipLaunchKernelGGL(tower_p8_kernel<3>, dim3(a.batch), dim3(X3Block::NTHR), X3Block::lds_bytes, s, a);
    } else if (a.ks == 5) {                                              // (5x5 runs exist in the two-role form only: the A/B switch leaves them alone)
        hipLaunchKernelGGL(tower_x3_roles_kernel<5>, dim3(a.batch), dim3(X3Block::NTHR), X3Block::lds_bytes + 8192, s, a);
    } else if (symmetric) hipLaunchKernelGGL(tower_x3_kernel, dim3(a.batch), dim3(X3Block::NTHR), X3Block::lds_bytes, s, a);
    else hipLaunchKernelGGL(tower_x3_roles_kernel<3>, dim3(a.batch), dim3(X3Block::NTHR), X3Block::lds_bytes, s, a);
}

}  // namespace cra
