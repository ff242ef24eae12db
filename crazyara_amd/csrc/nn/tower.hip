// Residual-tower kernel for gfx950: a run of consecutive 3x3 mobile-bottleneck blocks in ONE launch.
//
// Reference semantics: _BottlekneckResidualBlock / _ChannelAttentionModule / _EfficientChannelAttentionModule
// (DeepCrazyhouse/src/domain/neural_net/architectures/pytorch/builder_util.py:437-475, 83-114, 49-80).
//
// One workgroup = one board for the whole run of blocks: the 64 x 256 residual stream stays in LDS (f16) from the first
// block to the last, so between blocks there is no HBM round trip, no launch boundary and no separate SE-gate launch.
// Inside a block the C_op-wide intermediate is produced and consumed in chunks of 128 channels and never leaves the CU.
//
// 8 waves = 4 MATRIX waves (0-3) + 4 VECTOR waves (4-7); wave w and w+4 share a SIMD.  Per interval (one workgroup barrier):
//   matrix wave w : E(chunk k+1)  expand 1x1: 32 channels x 64 squares, K = 256, v_mfma_f32_32x32x16_f16, A = weight stream,
//                                 B = residual-stream rows (LDS); +BN1 bias, ReLU -> f16 tile t1 (LDS)
//                   P(chunk k-1)  project 1x1 into its persistent 64 couts x 64 squares accumulator, K = 128, B = tile t2 (LDS)
//   vector wave w : D(chunk k)    depthwise 3x3 of 32 channels, packed f16: 9 shifted LDS reads of t1 per square, v_pk_fma_f16,
//                                 +BN2 bias, ReLU -> t2
// Measured on MI355X and designed for (scripts/ubench/): a wave's own non-MFMA instructions do NOT overlap its 16x16x32
// MFMAs (the 16-cycle instruction leaves no free issue slot), a 32x32x16 MFMA hides ~5 of them; next to a saturated MFMA
// stream the partner wave of the SIMD gets ~2 VALU issues per MFMA.  Hence: 32x32x16 tiles, every per-interval instruction
// counted, packed-f16 depthwise, and all weights pre-packed on the host into per-wave STREAMS in exact consumption order
// (addresses are "stream position + lane"; a fixed window of loads stays in flight across chunk, block and SE boundaries).
// All 256 workgroups read the same streams in near lockstep, so the vector waves also touch every stream line a few
// intervals early (each workgroup 1/32 of the lines): without that, every line is an L2 miss-in-flight for everyone.
#include "kernels.h"
#include "device_utils.h"

#include <type_traits>

// The TW_DEV_* / TW_TRACE_* switches below exist to TIME parts of this kernel (scripts/build_variant.sh); several of them compute wrong
// results on purpose.  They only compile in a development build: a stray -DTW_DEV_... in a product build is a compile error, not a
// silently broken library.
#if !defined(CRA_DEVELOPMENT) && (defined(TW_DEV_NO_MFMA) || defined(TW_DEV_HALF_E_READS) || defined(TW_DEV_NO_WLOAD) || \
    defined(TW_DEV_VEC_NO_LDS) || defined(TW_DEV_VEC_NO_VALU) || defined(TW_DEV_VEC_TILES) || defined(TW_DEV_PRIO) || \
    defined(TW_DEV_NO_MATRIX) || defined(TW_DEV_NO_VECTOR) || defined(TW_DEV_NO_WARMUP) || defined(TW_TRACE_SE) || defined(TW_TRACE_BARRIERS) || \
    defined(TW_DEV_STAGGER) || defined(TW_DEV_DEPTH) || defined(TW_DEV_NO_BARRIER))
#error "TW_DEV_* / TW_TRACE_* are development switches (some compute wrong results): build with -DCRA_DEVELOPMENT, see scripts/build_variant.sh"
#endif

namespace cra {

namespace {
constexpr int TW_C = 256;
constexpr int TW_CK = 128;                           // C_op channels per chunk
// LDS row pitches (halves).  A 16-lane group of a ds_read_b128 must hit 16 distinct 16-byte slots of the 256-byte bank row:
//   MFMA B fragments (row = lane % 32, +16 B for lanes >= 32)    -> pitch = 1 slot  (mod 16): xs 528 B, t2 272 B
//   depthwise neighbour reads (row = l15 + const, +16 B per lg)   -> pitch = 2 slots (mod 16): t1 288 B
constexpr int TW_XROW = TW_C + 8;
constexpr int TW_T1ROW = TW_CK + 16;
constexpr int TW_T2ROW = TW_CK + 8;
constexpr int TW_XS_BYTES = 64 * TW_XROW * 2;        // 33792  residual stream tile
constexpr int TW_T1_BYTES = 67 * TW_T1ROW * 2;       // 19296  expand output of a chunk: zero row, 64 squares, two zero rows; x2
constexpr int TW_T2_BYTES = 64 * TW_T2ROW * 2;       // 17408  depthwise output of a chunk; x2
constexpr int TW_T1_OFF = TW_XS_BYTES;
constexpr int TW_T2_OFF = TW_T1_OFF + 2 * TW_T1_BYTES;
constexpr int TW_POOL_OFF = TW_T2_OFF + 2 * TW_T2_BYTES;      // float [256] channel sums of the new stream
constexpr int TW_SE_OFF = TW_POOL_OFF + 256 * 4;              // float mean[256], part[1024], h[128], gate[256]
constexpr int TW_B3_OFF = TW_SE_OFF + (256 + 1024 + 128 + 256) * 4;   // float [256] BN3 bias of the current block
constexpr int TW_DYN_LDS_BYTES = TW_B3_OFF + 256 * 4;         // the part above is the launch's dynamic LDS (Precision float16)
// Precision fp8 (F8 = true below): the two GEMMs of a block run on v_mfma_f32_32x32x64_f8f6f4 with e4m3 operands.  The residual stream
// stays f16 (xs); an e4m3 copy of it (xq) is the expand GEMM's B operand, and the depthwise writes its output t2 as e4m3 (rows of
// 128 bytes in the same two t2 regions).  Row pitches: an odd number of 16-byte slots, as for the f16 tiles.
constexpr int TW_XQROW = TW_C + 16;                           // bytes: 17 slots
constexpr int TW_T2ROW8 = TW_CK + 16;                         // bytes: 9 slots
constexpr int TW_XQ_OFF = TW_DYN_LDS_BYTES;
constexpr int TW_XQ_BYTES = 64 * TW_XQROW;                    // 17408
constexpr int TW_DYN_LDS_BYTES_F8 = TW_XQ_OFF + TW_XQ_BYTES;
// Depthwise weights: a static LDS array, 4 vector waves x 2 buffers x 2 KiB, filled by LDS-DMA.
constexpr int TW_PRM_ENT = 64;                                // bytes per entry of a 5 x 5 buffer: [entry][lane group][16 B]
constexpr int TW_PRM_ENT3 = 192;                              // 3 x 3: [entry][lane group][file variant: a | b..g | h][16 B]
constexpr int TW_PRM_BUF = 32 * TW_PRM_ENT;
constexpr int TW_PRM_BYTES = 4 * 2 * TW_PRM_BUF;
constexpr int TW_LDS_BYTES = TW_DYN_LDS_BYTES + TW_PRM_BYTES;
static_assert(TW_LDS_BYTES <= 160 * 1024 && TW_DYN_LDS_BYTES_F8 + TW_PRM_BYTES <= 160 * 1024, "LDS budget");
constexpr int TW_AHEAD = 96;                          // L2 warm-up distance in fragments per stream (3 full intervals, 384 KiB)
constexpr int TW_WIN = kTowerWindow;                  // weight fragments in flight per matrix wave (16 KiB)
static_assert(TW_WIN == 16, "one E or P phase consumes exactly one window");
// development (timing experiment): only TW_DEV_DEPTH of the window's 16 loads in flight (slot q % depth, refill `depth` positions ahead):
// does the weight stream run at "bytes in flight / loaded L2 latency"?
#ifdef TW_DEV_DEPTH
constexpr int TW_DEPTH = TW_DEV_DEPTH;
static_assert(TW_DEPTH == 8 || TW_DEPTH == 4, "the depth must divide the phase length");
#else
constexpr int TW_DEPTH = TW_WIN;
#endif

typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// (a, b) -> ReLU -> packed f16 pair, 2 VALU: v_cvt_pk_f16_f32 (gfx950) rounds both to nearest even; the BN bias is already in the
// accumulator (the expand accumulators START at the bias).
__device__ __forceinline__ uint32_t pack_relu_cvt(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_f16 %0, %0, 0" : "=&v"(r) : "v"(a), "v"(b));
    return r;
}

// A weight stream of one wave, read with raw buffer loads: resource descriptor + byte position live in SGPRs, the lane offset is
// one constant VGPR, so a refill costs no vector address arithmetic (a flat load needs a 64-bit add per 4 KiB of stream).
struct WStream {
    // The window start is a wave-uniform POINTER that moves once per phase, and every load of a phase addresses "window start + a
    // compile-time constant": the constant goes into the instruction's immediate / a loop-invariant SGPR.  (With a running byte
    // position in the soffset operand each of the 32 loads of an interval cost an s_add of its own -- 8 % of the matrix wave's
    // instructions, and a wave pays ~5 cycles of issue time for any instruction.)
    const char* base;    // window start (wave-uniform)
    uint32_t lane_off;   // lane * 16
    __device__ __forceinline__ half8 frag_at(int q) const {      // fragment q positions after the window start
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, q * 1024, 0);
        return __builtin_bit_cast(half8, v);
    }
    __device__ __forceinline__ void advance(int bytes) { base += bytes; }
};

// D(32x32) += A(32 x 16) * B(16 x 32): lane l holds A[row l%32][k = (l/32)*8 + j], B[k = (l/32)*8 + j][col l%32], j = 0..7;
// D[row (v%4) + 8*(v/4) + 4*(l/32)][col l%32] in element v.
__device__ __forceinline__ void mma32(const half8& a, const half8& b, f32x16& c) {
#ifdef TW_DEV_NO_MFMA
    asm volatile("" : "+v"(c) : "v"(a), "v"(b));
#else
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// the first MFMA of an accumulator: C = the bias tuple, result in the accumulator's own registers
__device__ __forceinline__ f32x16 mma32_init(const half8& a, const half8& b, const f32x16& c) {
#ifdef TW_DEV_NO_MFMA
    f32x16 r = c;
    asm volatile("" : "+v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
// BN1 bias of a lane's 16 rows (bias stream: [lane/32][element v]) as one 16-register tuple
__device__ __forceinline__ void load_bias(f32x16& bias, const float* __restrict__ bp) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 q = reinterpret_cast<const f32x4*>(bp)[i];
        bias[4 * i + 0] = q[0]; bias[4 * i + 1] = q[1]; bias[4 * i + 2] = q[2]; bias[4 * i + 3] = q[3];
    }
}

// ---- Precision fp8 ----
// D(32x32) += A(32 x 64) * B(64 x 32), e4m3 operands: lane l holds row / column l % 32 and the 32 bytes k = (l/32)*32 + t of a k-step
// (any labelling of k that is the same on both sides gives the same sum; scripts/ubench/fp8_probe.hip checks this one), D as above.
// 64 cycles: twice the MACs per cycle of the f16 instruction.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8 cat32(const half8& lo, const half8& hi) {
    const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
    return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ void mma64(const i32x8& a, const i32x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma64_init(const i32x8& a, const i32x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}
// four packed f16 pairs -> 8 e4m3 bytes (round to nearest even; the kernel runs with MODE.FP16_OVFL = 1, so values beyond +-448 clamp
// instead of becoming NaN): v_cvt_scalef32_pk_fp8_f16 fills one 16-bit half of its destination per instruction
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f16x4_to_e4m3(uint32_t lo, uint32_t hi) {
    s16x2 q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, __builtin_bit_cast(half2_t, lo), 1.0f, false);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, __builtin_bit_cast(half2_t, hi), 1.0f, true);
    return __builtin_bit_cast(uint32_t, q);
}

// ---- Precision int8 (Q = 2): the calibrated INT8 mode (the reference's TensorRT INT8, tensorrtapi.cpp:334-360), Precision fp8's structure with
// int8 operands on v_mfma_i32_32x32x32_i8.  An fp8 MFMA's 32 bytes per lane are two int8 MFMAs' 16 + 16 (the same bytes meet on both
// operands, so the sum is the same dot product); the accumulators are int32 in the registers the fp8 path holds floats in.
//   quantiser: z = x * inv + magic in ONE f16 FMA -- magic = 1536 (1152) puts z into [1024, 2048), where an f16's unit is 1: the FMA's
//   single rounding IS the rounding to the integer grid (half to even), and the LOW BYTE of z's bits is the integer (two's complement:
//   1536 = 6 * 256; 1152 = 4 * 256 + 128 stores an unsigned u as u - 128); clamped to +-127 (0 ... 255) by a packed min / max.
typedef int i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mma64_i8(const i32x8& a, const i32x8& b, const f32x16& c) {
    i32x16 ci = __builtin_bit_cast(i32x16, c);
    ci = __builtin_amdgcn_mfma_i32_32x32x32_i8(i32x4{a[0], a[1], a[2], a[3]}, i32x4{b[0], b[1], b[2], b[3]}, ci, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_i32_32x32x32_i8(i32x4{a[4], a[5], a[6], a[7]}, i32x4{b[4], b[5], b[6], b[7]}, ci, 0, 0, 0);
    return __builtin_bit_cast(f32x16, ci);
}
template <int Q> __device__ __forceinline__ void mma8(const i32x8& a, const i32x8& b, f32x16& c) {
    if constexpr (Q == 2) c = mma64_i8(a, b, c); else mma64(a, b, c);
}
template <int Q> __device__ __forceinline__ f32x16 mma8_init(const i32x8& a, const i32x8& b, const f32x16& c) {
    if constexpr (Q == 2) return mma64_i8(a, b, c); else return mma64_init(a, b, c);
}
struct Q8 { uint32_t inv2, magic2, lo2, hi2; };      // packed f16 pairs: 1 / step, the rounding constant, the clamps
__device__ __forceinline__ Q8 q8_signed(float inv) {       // the residual stream: +-127
    const uint32_t h = __builtin_bit_cast(unsigned short, half_t(inv));
    return Q8{h | (h << 16), 0x66006600u, 0x65816581u, 0x667f667fu};
}
__device__ __forceinline__ Q8 q8_unsigned(float inv) {     // the depthwise output (post-ReLU): 0 ... 255, stored as u - 128
    const uint32_t h = __builtin_bit_cast(unsigned short, half_t(inv));
    return Q8{h | (h << 16), 0x64806480u, 0x64806480u, 0x657f657fu};
}
__device__ __forceinline__ uint32_t f16x4_to_i8(uint32_t lo, uint32_t hi, const Q8& q) {
    uint32_t a, b;
    asm("v_pk_fma_f16 %0, %2, %4, %5\n\tv_pk_fma_f16 %1, %3, %4, %5\n\t"
        "v_pk_min_f16 %0, %0, %7\n\tv_pk_min_f16 %1, %1, %7\n\tv_pk_max_f16 %0, %0, %6\n\tv_pk_max_f16 %1, %1, %6"
        : "=&v"(a), "=&v"(b)
        : "v"(lo), "v"(hi), "v"(q.inv2), "v"(q.magic2), "v"(q.lo2), "v"(q.hi2));
    return __builtin_amdgcn_perm(b, a, 0x06040200u);                 // the low bytes of the four halves, in order
}
// the two byte forms behind one name: Q = 1 e4m3, Q = 2 int8 at the tensor's calibrated step
template <int Q> __device__ __forceinline__ uint32_t f16x4_to_q8(uint32_t lo, uint32_t hi, const Q8& q) {
    if constexpr (Q == 2) return f16x4_to_i8(lo, hi, q); else return f16x4_to_e4m3(lo, hi);
}
template <int Q> __device__ __forceinline__ uint2 f16x8_to_q8(const uint4& h, const Q8& q) {
    return uint2{f16x4_to_q8<Q>(h.x, h.y, q), f16x4_to_q8<Q>(h.z, h.w, q)};
}
// expand epilogue, int8: the int32 accumulator (it started at the BN1 bias in its own unit) -> relu -> x 2^-7 -> f16 (the per-row value of
// a unit and the 2^7 are folded into the depthwise weights on the host)
__device__ __forceinline__ uint32_t pack_relu_cvt_i32(float a_bits, float b_bits, float scale) {
    return pack_relu_cvt(float(__builtin_bit_cast(int, a_bits)) * scale, float(__builtin_bit_cast(int, b_bits)) * scale);
}

// matrix role, Precision fp8, one phase.  A fragment (32 rows x 64 k) is two loads of 1 KiB ([half][lane][16 B]: a lane's bytes 0..15
// and 16..31), so a phase needs 8 loads where the f16 phase needs 16.  The expand and the project fragments therefore come as TWO streams
// per wave, each with its own window of 8 loads (16 KiB in flight as before): a phase consumes exactly its window and refills it for
// the next phase of its kind, whatever the other kind does in between (the first and last intervals of a block run only one of them).
//   expand : 4 k-steps x {1 fragment, 2 square tiles}: 8 MFMAs          project: 2 k-steps x {2 row tiles, 2 square tiles}: 8 MFMAs
constexpr int TW_WIN8 = 8;
template <int Q>
__device__ __forceinline__ void expand_phase8(f32x16 (&accE)[2], half8 (&win)[TW_WIN8], WStream& sp, f32x16& bias,
                                              const float* __restrict__& bp, const char* xqr) {
    constexpr int BASE = 0;
    using frag = half8;
    frag ba[4], bb[4];                               // [square tile][16-byte half] of a k-step
#pragma unroll
    for (int i = 0; i < 4; ++i) ba[i] = *reinterpret_cast<const frag*>(xqr + (i >> 1) * 32 * TW_XQROW + (i & 1) * 16);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        frag (&cur)[4] = (s & 1) ? bb : ba;
        frag (&nxt)[4] = (s & 1) ? ba : bb;
        if (s + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nxt[i] = *reinterpret_cast<const frag*>(xqr + (i >> 1) * 32 * TW_XQROW + (s + 1) * 64 + (i & 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        const i32x8 a = cat32(win[BASE + 2 * s], win[BASE + 2 * s + 1]);
        // youngest operands first, the bias as the first MFMA's C operand (see matrix_interval)
        if (s == 0) {
            accE[1] = mma8_init<Q>(a, cat32(cur[2], cur[3]), bias);
            accE[0] = mma8_init<Q>(a, cat32(cur[0], cur[1]), bias);
        } else {
            mma8<Q>(a, cat32(cur[2], cur[3]), accE[1]);
            mma8<Q>(a, cat32(cur[0], cur[1]), accE[0]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) win[BASE + 2 * s + e] = sp.frag_at(2 * s + e + TW_WIN8);
        if (s == 0) {
            asm volatile("" ::"v"(bias));            // the tuple outlives both MFMAs (see matrix_interval)
            bp += 32;
            load_bias(bias, bp);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    sp.advance(8 * 1024);
}
template <int Q, typename EPI>
__device__ __forceinline__ void project_phase8(f32x16 (&accP)[2][2], half8 (&win)[TW_WIN8], WStream& sp, const char* t2r, const EPI& epilogue) {
    using frag = half8;
    constexpr int BASE = 0;
    frag ba[4], bb[4];                               // [square tile][16-byte half]
#pragma unroll
    for (int i = 0; i < 4; ++i) ba[i] = *reinterpret_cast<const frag*>(t2r + (i >> 1) * 32 * TW_T2ROW8 + (i & 1) * 16);
#pragma unroll
    for (int s = 0; s < 2; ++s) {                    // k-step s: loads [row tile rt][half]
        frag (&cur)[4] = (s & 1) ? bb : ba;
        frag (&nxt)[4] = (s & 1) ? ba : bb;
        if (s + 1 < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nxt[i] = *reinterpret_cast<const frag*>(t2r + (i >> 1) * 32 * TW_T2ROW8 + (s + 1) * 64 + (i & 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 1; rt >= 0; --rt) {            // youngest operands first (see matrix_interval)
            const i32x8 a = cat32(win[BASE + s * 4 + rt * 2], win[BASE + s * 4 + rt * 2 + 1]);
            mma8<Q>(a, cat32(cur[2], cur[3]), accP[rt][1]);
            mma8<Q>(a, cat32(cur[0], cur[1]), accP[rt][0]);
        }
        epilogue(s);
#pragma unroll
        for (int e = 0; e < 4; ++e) win[BASE + s * 4 + e] = sp.frag_at(s * 4 + e + TW_WIN8);
        __builtin_amdgcn_sched_barrier(0);
    }
    sp.advance(8 * 1024);
}

// ---------------------------------------------------------------------------------------------------------------------
// matrix role, one interval: E(next chunk) then P(previous chunk)
//   win : 16 weight fragments (1 KiB each) in flight; slot q holds stream fragment (position + q) and is refilled with the
//         fragment 16 positions ahead as soon as its MFMAs are issued.  E consumes 16 fragments ([k-step]), P 16 ([k-step][row tile]).
//   A step = 4 MFMAs + the LDS reads of the NEXT step's B fragments + 2 refills; nothing is scheduled across step boundaries.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void matrix_interval(bool do_e, bool do_p, f32x16 (&accP)[2][2], half8 (&win)[TW_WIN], WStream& sp,
                                                f32x16& bias, const float* __restrict__& bp, const half_t* xsr, half_t* t1w,
                                                const half_t* t2r) {
    using frag = half8;
    constexpr int XROW = TW_XROW, T1ROW = TW_T1ROW, T2ROW = TW_T2ROW;
    f32x16 accE[2];                                  // [square tile of 32]
    // bias: BN1 bias of my 16 rows (v%4) + 8*(v/4) + 4*(lane/32) for THIS expand phase, loaded one phase ahead (below).  Loaded at
    // the start of the phase that uses it, the wave would have to drain every weight load in flight (vmcnt counts in order) and
    // then sit out an L2 round trip before its first MFMA, once per interval.
    if (do_e) {
        frag bfa[4], bfb[4];                         // [k-step parity within the step][square tile]
#pragma unroll
        for (int i = 0; i < 4; ++i) bfa[i] = *reinterpret_cast<const frag*>(xsr + (i & 1) * 32 * XROW + (i >> 1) * 16);
#pragma unroll
        for (int s = 0; s < 8; ++s) {                // step = k-steps 2s, 2s+1
            frag (&cur)[4] = (s & 1) ? bfb : bfa;
            frag (&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 8) {
#ifdef TW_DEV_HALF_E_READS
                // development (wrong results): half of the B-fragment reads of the expand phase, to time the operand traffic
#pragma unroll
                for (int i = 0; i < 2; ++i) nxt[i] = *reinterpret_cast<const frag*>(xsr + (i & 1) * 32 * XROW + ((s + 1) * 2 + (i >> 1)) * 16);
                nxt[2] = nxt[0]; nxt[3] = nxt[1];
#else
#pragma unroll
                for (int i = 0; i < 4; ++i) nxt[i] = *reinterpret_cast<const frag*>(xsr + (i & 1) * 32 * XROW + ((s + 1) * 2 + (i >> 1)) * 16);
#endif
            }
            // The reads of the NEXT step's B fragments must be ISSUED before this step's MFMAs (they then have a whole step to land).
            // Without this fence the scheduler sinks them below the MFMAs to save registers -- to the end of the step, directly in
            // front of the MFMAs that consume them -- and every step waits out a full LDS latency (measured: 179 cycles per step).
            __builtin_amdgcn_sched_barrier(0);
            // The step's MFMAs run YOUNGEST OPERANDS FIRST (i = 3 .. 0): loads and LDS reads return in order, so the one s_waitcnt in
            // front of the first MFMA covers the other three (oldest first, every MFMA had a wait of its own: 4 per step, and a wave
            // pays issue time for a wait like for any instruction).  The first MFMA of a tile takes the BN1 bias as its C operand
            // (a register tuple of its own: accumulators that START as a copy of the bias cost 24 moves per interval).
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                if (s == 0 && i >= 2) accE[i & 1] = mma32_init(win[(s * 2 + (i >> 1)) % TW_DEPTH], cur[i], bias);
                else mma32(win[(s * 2 + (i >> 1)) % TW_DEPTH], cur[i], accE[i & 1]);
            }
#ifndef TW_DEV_NO_WLOAD
#pragma unroll
            for (int e = 0; e < 2; ++e) win[(s * 2 + e) % TW_DEPTH] = sp.frag_at(s * 2 + e + TW_DEPTH);
#endif
            if (s == 0) {                            // the first MFMAs have read the bias registers: fetch the next phase's
                asm volatile("" ::"v"(bias));        // (the tuple outlives both MFMAs, or the second one accumulates in place in it)
                bp += 32;                            // (the stream is closed with one chunk of zeros, rise_net.hip)
                load_bias(bias, bp);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        sp.advance(16 * 1024);
    }
    // expand epilogue for square tile ct: BN1 bias + ReLU -> f16.  A lane's 16 rows are 16 consecutive K positions of the tile
    // the project GEMM reads as its B operand (position p <-> row (p%4) + 8*((p%16)/4) + 4*(p/16), kernels.h: tower_k_channel),
    // i.e. two 16-byte stores.  Issued inside the first project steps so that its VALU work hides behind their MFMAs.
    auto expand_epilogue = [&](int ct) {
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = pack_relu_cvt(accE[ct][2 * i], accE[ct][2 * i + 1]);
        uint4* dst = reinterpret_cast<uint4*>(t1w + ct * 32 * T1ROW);
        dst[0] = uint4{o[0], o[1], o[2], o[3]};
        dst[1] = uint4{o[4], o[5], o[6], o[7]};
    };
    if (do_p) {
        frag bfa[4], bfb[4];                         // [k-step parity][square tile]
#pragma unroll
        for (int i = 0; i < 4; ++i) bfa[i] = *reinterpret_cast<const frag*>(t2r + (i & 1) * 32 * T2ROW + (i >> 1) * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {                // step = k-steps 2s, 2s+1: 8 MFMAs
            frag (&cur)[4] = (s & 1) ? bfb : bfa;
            frag (&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) nxt[i] = *reinterpret_cast<const frag*>(t2r + (i & 1) * 32 * T2ROW + ((s + 1) * 2 + (i >> 1)) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);       // as in the expand loop: next step's reads are issued before this step's MFMAs
#pragma unroll
            for (int kk = 1; kk >= 0; --kk)          // youngest operands first (see the expand loop)
#pragma unroll
                for (int rt = 1; rt >= 0; --rt)
#pragma unroll
                    for (int ct = 1; ct >= 0; --ct) mma32(win[((s * 2 + kk) * 2 + rt) % TW_DEPTH], cur[kk * 2 + ct], accP[rt][ct]);
            if (do_e && s < 2) expand_epilogue(s);
#ifndef TW_DEV_NO_WLOAD
#pragma unroll
            for (int e = 0; e < 4; ++e) win[(s * 4 + e) % TW_DEPTH] = sp.frag_at(s * 4 + e + TW_DEPTH);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        sp.advance(16 * 1024);
    } else if (do_e) {
        mfma_retire(accE[0], accE[1]);               // the last expand MFMAs retire before the asm epilogue reads them (device_utils.h)
        expand_epilogue(0);
        expand_epilogue(1);
    }
}

// the same interval in Precision fp8 (window halves: [0, 8) expand stream, [8, 16) project stream)
template <int Q>
__device__ __forceinline__ void matrix_interval8(bool do_e, bool do_p, f32x16 (&accP)[2][2], half8 (&winE)[TW_WIN8], half8 (&winP)[TW_WIN8],
                                                 WStream& spE, WStream& spP, f32x16& bias, const float* __restrict__& bp, const char* xqr,
                                                 half_t* t1w, const char* t2r, float escale) {
    constexpr int T1ROW = TW_T1ROW;
    f32x16 accE[2];
    if (do_e) expand_phase8<Q>(accE, winE, spE, bias, bp, xqr);
    auto expand_epilogue = [&](int ct) {             // as in matrix_interval: the accumulators started at the (scaled) BN1 bias
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = Q == 2 ? pack_relu_cvt_i32(accE[ct][2 * i], accE[ct][2 * i + 1], escale) : pack_relu_cvt(accE[ct][2 * i], accE[ct][2 * i + 1]);
        uint4* dst = reinterpret_cast<uint4*>(t1w + ct * 32 * T1ROW);
        dst[0] = uint4{o[0], o[1], o[2], o[3]};
        dst[1] = uint4{o[4], o[5], o[6], o[7]};
    };
    if (do_p) {
        auto epi = [&](int s) { if (do_e) expand_epilogue(s); };
        project_phase8<Q>(accP, winP, spP, t2r, epi);
    } else if (do_e) {
        mfma_retire(accE[0], accE[1]);               // the last expand MFMAs (16 passes) retire first
        expand_epilogue(0);
        expand_epilogue(1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// vector role, one interval: D(chunk) for this wave's 32 channels, packed f16 math.
//   * a lane owns square (tile t of 16, l15) and 8 consecutive K positions = 4 channel PAIRS held as packed halves;
//   * the 9 neighbours are 9 LDS reads of 16 bytes from the t1 tile (rows above the board / below it are the tile's
//     two zero rows; file wrap-around is cancelled by zeroed weights), no cross-lane traffic, no conversions;
//   * v_pk_fma_f16 accumulates two channels per instruction (f16 accumulate: +0.2e-3 on the logits against f32
//     accumulation in the oracle's emulation, tolerance 6e-3 -- DESIGN.md), v_pk_max_f16 is the ReLU.
// Weights: per chunk and wave 2 KiB in the stream = [32 entries: k*k taps, BN2 bias, pad][lane group lg][4 pairs] half2, brought
// one chunk ahead straight into a wave-private LDS buffer by two LDS-DMA loads (no VGPRs in between), read back as 10 (26)
// broadcast reads.  Neighbour rows are read one square tile ahead of the FMAs that use them; the bottom row of a tile is the
// top row of the next (27 reads per chunk, not 36); the four channel-pair chains are interleaved tap by tap.
// ---------------------------------------------------------------------------------------------------------------------
struct VecAddr {
    const char* tap[9];  // (LDS) my neighbour row for tap (dy+1)*3 + (dx+1) in tile 0 of buffer 0, rank-valid case
    const char* top[3];  // tile 0, dy = -1: the zero row above the board for l15 < 8
    const char* bot[3];  // tile 3, dy = +1: the zero row below the board for l15 >= 8 (pre-biased by -3 tiles)
};

// The wave's depthwise-weight stream: chunk g sits in LDS buffer g & 1 one interval before it is used.
struct VParams {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t pos;        // byte position of the chunk that is being computed (wave-uniform)
    uint32_t lane_off;   // lane * 16
    uint32_t buf;        // LDS buffer of that chunk
    char* lds;           // my two buffers
    __device__ __forceinline__ void fetch(uint32_t chunk_pos, uint32_t b) const {     // 2 x (64 lanes x 16 B) -> buffer b
        auto dst = (__attribute__((address_space(3))) void*)(lds + b * TW_PRM_BUF);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, lane_off, chunk_pos, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, lane_off, chunk_pos, 1024, 0);
    }
    // start of an interval: my chunk has landed (it was requested a whole interval ago)
    __device__ __forceinline__ const char* open() const {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return lds + buf * TW_PRM_BUF;
    }
    // end of an interval: request the next chunk into the other buffer
    __device__ __forceinline__ void fetch_next() {
        pos += 2048;
        buf ^= 1;
        fetch(pos, buf);
    }
};

// development switches (timing only): TW_DEV_VEC_NO_LDS replaces the neighbour reads by a register constant, TW_DEV_VEC_NO_VALU
// keeps the reads and skips the FMAs
__device__ __forceinline__ uint4 vec_row_read(const char* p) {
#ifdef TW_DEV_VEC_NO_LDS
    uint4 r;
    asm volatile("v_mov_b32 %0, 0x3c003c00\n\tv_mov_b32 %1, 0x3c003c00\n\tv_mov_b32 %2, 0x3c003c00\n\tv_mov_b32 %3, 0x3c003c00" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w));
    return r;
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}

// t2w: byte address of my output slot in tile 0 of t2 buffer 0 (f16: row pitch T2ROW halves, 16 bytes; fp8: TW_T2ROW8 bytes, 8 bytes)
template <int Q>
__device__ __forceinline__ void vector_store(char* t2w, int parity, int t, const uint32_t (&o)[4], const Q8& qt) {
    constexpr bool F8 = Q != 0;
    if constexpr (F8) *reinterpret_cast<uint2*>(t2w + parity * TW_T2_BYTES + t * 16 * TW_T2ROW8) = uint2{f16x4_to_q8<Q>(o[0], o[1], qt), f16x4_to_q8<Q>(o[2], o[3], qt)};
    else *reinterpret_cast<uint4*>(t2w + parity * TW_T2_BYTES + t * 16 * TW_T2ROW * 2) = uint4{o[0], o[1], o[2], o[3]};
}

// prm_off: byte offset of my weights inside an entry = lane group * 48 + file variant * 16.  The board has no file left of a and none
// right of h: instead of multiplying the taps that leave the board by zero in every interval (24 packed multiplies), the stream
// carries the weights three times -- as they are, with the dx = -1 taps zeroed (file a), with the dx = +1 taps zeroed (file h).
template <int PARITY, int Q>
__device__ __forceinline__ void vector_interval(VParams& vp, const char* prm_buf, int prm_off, const VecAddr& va, char* t2w, const Q8& qt) {
    constexpr int T1ROW = TW_T1ROW;
    constexpr int TILE = 16 * T1ROW * 2;             // bytes between square tiles of a t1 buffer
    const char* prm = prm_buf + prm_off;
    // rows of neighbours: top(t) | mid(t) | bot(t), 3 reads each; bot(t) == top(t + 1)
    uint4 top[3], mid[3], bot[3], nmid[3], nbot[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        top[i] = vec_row_read(va.top[i] + PARITY * TW_T1_BYTES);
        mid[i] = vec_row_read(va.tap[3 + i] + PARITY * TW_T1_BYTES);
        bot[i] = vec_row_read(va.tap[6 + i] + PARITY * TW_T1_BYTES);
    }
    half2_t W[10][4];
#pragma unroll
    for (int e = 0; e < 10; ++e) {
        const uint4 u = *reinterpret_cast<const uint4*>(prm + e * TW_PRM_ENT3);
        W[e][0] = __builtin_bit_cast(half2_t, u.x); W[e][1] = __builtin_bit_cast(half2_t, u.y);
        W[e][2] = __builtin_bit_cast(half2_t, u.z); W[e][3] = __builtin_bit_cast(half2_t, u.w);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
#ifdef TW_DEV_VEC_TILES
    for (int t = 0; t < TW_DEV_VEC_TILES; ++t) {     // development: only part of the depthwise work (timing experiment)
#else
    for (int t = 0; t < 4; ++t) {
#endif
        if (t < 3) {                                 // next tile's rows go out before this tile's FMAs
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                nmid[i] = vec_row_read(va.tap[3 + i] + PARITY * TW_T1_BYTES + (t + 1) * TILE);
                nbot[i] = vec_row_read((t == 2 ? va.bot[i] : va.tap[6 + i]) + PARITY * TW_T1_BYTES + (t + 1) * TILE);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        half2_t acc[4] = {W[9][0], W[9][1], W[9][2], W[9][3]};
#ifdef TW_DEV_VEC_NO_VALU
#pragma unroll
        for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(top[i]), "v"(mid[i]), "v"(bot[i]));
#else
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int pi = 0; pi < 4; ++pi)
                acc[pi] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&top[i])[pi]), W[i][pi], acc[pi]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int pi = 0; pi < 4; ++pi)
                acc[pi] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&mid[i])[pi]), W[3 + i][pi], acc[pi]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int pi = 0; pi < 4; ++pi)
                acc[pi] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&bot[i])[pi]), W[6 + i][pi], acc[pi]);
#endif
        uint32_t o[4];
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) o[pi] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(acc[pi], half2_t{0, 0}));
        vector_store<Q>(t2w, PARITY, t, o, qt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { top[i] = bot[i]; mid[i] = nmid[i]; bot[i] = nbot[i]; }
    }
    vp.fetch_next();                                 // last: the compiler orders every later LDS access behind a pending LDS-DMA
}

// 5 x 5 depthwise (RISEv3.3 blocks 7, 11, 12, 13): same scheme, 25 neighbour reads and 100 packed FMAs per square tile.
// Entries 0..24 = taps (dy+2)*5 + (dx+2), entry 25 = BN2 bias.  Rows two above / below the board: the tile's zero rows.
struct VecAddr5 {
    const char* tap[25]; // my neighbour row for each tap in tile 0 of buffer 0, rank-valid case
    const char* top[5];  // tile 0, dy = -1: zero row for l15 < 8         (dy = -2 in tile 0: always the zero row)
    const char* bot[5];  // tile 3, dy = +1: zero row for l15 >= 8, pre-biased by -3 tiles   (dy = +2 in tile 3: always the zero row)
    const char* zero0;   // the zero row above the board at my columns
    const char* zero3;   // the zero row below the board at my columns, pre-biased by -3 tiles
};

template <int PARITY, int Q>
__device__ __forceinline__ void vector_interval5(VParams& vp, const char* prm_buf, int lg, const VecAddr5& va, char* t2w, const half2_t (&mk)[5], const Q8& qt) {
    constexpr int T1ROW = TW_T1ROW;
    const char* prm = prm_buf + lg * 16;
    half2_t W[26][4];
#pragma unroll
    for (int e = 0; e < 26; ++e) {
        const uint4 u = *reinterpret_cast<const uint4*>(prm + e * TW_PRM_ENT);
        W[e][0] = __builtin_bit_cast(half2_t, u.x); W[e][1] = __builtin_bit_cast(half2_t, u.y);
        W[e][2] = __builtin_bit_cast(half2_t, u.z); W[e][3] = __builtin_bit_cast(half2_t, u.w);
    }
#pragma unroll
    for (int tap = 0; tap < 25; ++tap)               // file wrap-around: taps whose column leaves the board get a zero weight
        if (tap % 5 != 2) {
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) W[tap][pi] *= mk[tap % 5];
        }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        half2_t acc[4] = {W[25][0], W[25][1], W[25][2], W[25][3]};
#pragma unroll
        for (int g = 0; g < 5; ++g) {                // one board row of taps at a time
            uint4 R[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const char* base = va.tap[g * 5 + c];
                if (t == 0 && g == 0) base = va.zero0;
                if (t == 0 && g == 1) base = va.top[c];
                if (t == 3 && g == 3) base = va.bot[c];
                if (t == 3 && g == 4) base = va.zero3;
                R[c] = *reinterpret_cast<const uint4*>(base + PARITY * TW_T1_BYTES + t * 16 * T1ROW * 2);
            }
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int pi = 0; pi < 4; ++pi)
                    acc[pi] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&R[c])[pi]), W[g * 5 + c][pi], acc[pi]);
        }
        uint32_t o[4];
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) o[pi] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(acc[pi], half2_t{0, 0}));
        vector_store<Q>(t2w, PARITY, t, o, qt);
    }
    vp.fetch_next();
}

// SE gate of a block (squeeze over the residual stream in LDS, excitation MLP, scale in place); executed by all 512 threads
template <int Q>
__device__ __forceinline__ void se_phase(const TowerBlockDesc& d, int tid, half_t* xs, char* xq, const float* pool_sum, float* se_mean,
                                     float* se_part, float* se_h, float* se_gate, unsigned long long* trc, int& trn) {
    constexpr bool F8 = Q != 0;
    const Q8 qx = q8_signed(d.qx_inv);               // (int8: the gated stream is quantised at this block's calibrated step)
    constexpr int XROW = TW_XROW, SE_GRP = 36;       // floats per group of 32 means / hidden values (bank spread, see below)
    {
        // The gate weights do not depend on the data: a thread's first 32 dwords go out before the squeeze and fly while it runs,
        // the second 32 behind the squeeze (they fly during FC1).
        // Host-packed in thread order (rise_net.hip: pack_se_threads): 8 coalesced 16-byte loads per thread and matrix.
        half2_t wa[32], wb[32];
        auto load_thread_weights = [&](const void* base, half2_t (&dst)[32]) {
            const uint4* pk = reinterpret_cast<const uint4*>(base) + tid;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 u = pk[i * 512];
                dst[4 * i + 0] = __builtin_bit_cast(half2_t, u.x); dst[4 * i + 1] = __builtin_bit_cast(half2_t, u.y);
                dst[4 * i + 2] = __builtin_bit_cast(half2_t, u.z); dst[4 * i + 3] = __builtin_bit_cast(half2_t, u.w);
            }
        };
        // Excitation without a trip through LDS for the partial sums: the threads that share an output pair are NEIGHBOURING LANES
        // (8 for FC1, 4 for FC2 / the eca matrix), each takes 32 inputs (eca: 2 x 32), and a DPP row_shr reduction leaves the total in
        // the last lane of the group.  Barriers per SE phase: squeeze | FC1 | FC2 | scale (were six).  The vectors a group's lanes
        // read (means, hidden) sit in LDS with 36 floats per 32 inputs, so that the 8 (4) concurrent 16-byte reads hit distinct banks.
        //   ca_se : FC1 thread t -> hidden 2*(t/8), +1 over c in [32*(t%8), +32);  FC2 thread t -> gate 2*(t/4), +1 over j in [32*(t%4), +32)
        //   eca_se: thread t -> gate 2*(t/4), +1 over inputs i in [64*(t%4), +64): first 32, then the second 32
        load_thread_weights(d.se_w1, wa);
        {   // squeeze: mean over the 64 squares of the residual stream as it sits in LDS.  A wave owns 32 channels: lane =
            // (4 groups of 8 channels) x (16 groups of 4 squares); 16-byte reads, then a 16-lane DPP row reduction.
            const int lane = tid & 63, wv = tid >> 6, cg = lane >> 4, sg = lane & 15;
            float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float xv[8];
                load8<half_t>(xs + (sg * 4 + q) * XROW + wv * 32 + cg * 8, xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum[j] += xv[j];
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
                for (int j = 0; j < 8; ++j) se_mean[wv * SE_GRP + cg * 8 + j] = sum[j] * (1.f / 64.f);      // channel c at (c / 32) * 36 + c % 32
            }
        }
        if (d.se_kind == 1) load_thread_weights(d.se_w2, wb);                     // flies during FC1
        else load_thread_weights(reinterpret_cast<const char*>(d.se_w1) + 8 * 512 * 16, wb);
        __syncthreads();
#ifdef TW_TRACE_SE
        if (trc) trc[trn++] = __builtin_amdgcn_s_memtime();
#endif
        auto dot32 = [](const half2_t (&w)[32], const float* v, float& s0, float& s1) {   // v: 32 floats, 16-byte aligned
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(v + 4 * k4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s0 = fmaf(float(w[4 * k4 + j][0]), m[j], s0);
                    s1 = fmaf(float(w[4 * k4 + j][1]), m[j], s1);
                }
            }
        };
        if (d.se_kind == 1) {        // ca_se: relu(W1 mean) -> W2 -> hard-sigmoid (builder_util.py:83-114)
            {
                const int j2 = tid >> 3, kq = tid & 7;
                float s0 = 0.f, s1 = 0.f;
                dot32(wa, se_mean + kq * SE_GRP, s0, s1);
                s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
                s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
                s0 += dpp_mov<0x114>(s0); s1 += dpp_mov<0x114>(s1);             // lane 7 of the group of 8: the whole sum
                if (kq == 7) {                                                    // hidden j = 2*j2, +1 at (j / 32) * 36 + j % 32
                    float* h = se_h + (j2 >> 4) * SE_GRP + 2 * (j2 & 15);
                    h[0] = fmaxf(s0, 0.f);
                    h[1] = fmaxf(s1, 0.f);
                }
            }
            __syncthreads();
#ifdef TW_TRACE_SE
            if (trc) trc[trn++] = __builtin_amdgcn_s_memtime();
#endif
            {
                const int c2 = tid >> 2, kq = tid & 3;
                float s0 = 0.f, s1 = 0.f;
                dot32(wb, se_h + kq * SE_GRP, s0, s1);
                s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
                s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);             // lane 3 of the group of 4
                if (kq == 3) {
                    se_gate[2 * c2] = hard_sigmoid(s0);
                    se_gate[2 * c2 + 1] = hard_sigmoid(s1);
                }
            }
        } else {                     // eca_se: centre-tap linear + bias -> hard-sigmoid (builder_util.py:49-80)
            const int c2 = tid >> 2, kq = tid & 3;
            float s0 = 0.f, s1 = 0.f;
            dot32(wa, se_mean + (2 * kq) * SE_GRP, s0, s1);
            dot32(wb, se_mean + (2 * kq + 1) * SE_GRP, s0, s1);
            s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
            s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
            if (kq == 3) {
                se_gate[2 * c2] = hard_sigmoid(d.se_b[2 * c2] + s0);
                se_gate[2 * c2 + 1] = hard_sigmoid(d.se_b[2 * c2 + 1] + s1);
            }
        }
        __syncthreads();
#ifdef TW_TRACE_SE
        if (trc) trc[trn++] = __builtin_amdgcn_s_memtime();
#endif
        for (int i = tid; i < 64 * 32; i += 512) {     // x := x * gate (the residual uses the gated x, builder_util.py:473-475)
            const int r = i >> 5, v = i & 31;
            float xv[8], gv[8];
            load8<half_t>(xs + r * XROW + v * 8, xv);
            load8<float>(se_gate + v * 8, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] *= gv[j];
            store8<half_t>(xs + r * XROW + v * 8, xv);
            if constexpr (F8)                        // the e4m3 copy follows the ROUNDED f16 values (what the oracle's emulation quantises)
                *reinterpret_cast<uint2*>(xq + r * TW_XQROW + v * 8) = f16x8_to_q8<Q>(*reinterpret_cast<const uint4*>(xs + r * XROW + v * 8), qx);
        }
        __syncthreads();
#ifdef TW_TRACE_SE
        if (trc) trc[trn++] = __builtin_amdgcn_s_memtime();
#endif
    }

}
}  // namespace

#ifndef CRA_FORWARD_TU
size_t tower_lds_bytes() { return TW_LDS_BYTES; }
#endif

// x_in_lds: the board's residual-stream tile is already at offset 0 of the dynamic LDS segment (left there by the stem of the same
// launch, forward.hip); y_to_global = false: it stays there for the head of the same launch.  Both need gate_in == pool_out == nullptr.
template <int Q>
__device__ __forceinline__ void tower_body(const TowerArgs& a, const bool x_in_lds, const bool y_to_global) {
    constexpr bool F8 = Q != 0;                      // the byte tiles, the two weight streams, the windows of 8: Precision fp8 and int8 alike
    using frag = half8;
    if constexpr (F8) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // MODE.FP16_OVFL: conversions to f16 / e4m3 clamp instead of overflowing
    constexpr int C = TW_C, XROW = TW_XROW, T1ROW = TW_T1ROW, T2ROW = TW_T2ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) char prm_lds[TW_PRM_BYTES];
    half_t* xs = reinterpret_cast<half_t*>(smem);
    char* xq = smem + TW_XQ_OFF;                     // Precision fp8 only
    float* pool_sum = reinterpret_cast<float*>(smem + TW_POOL_OFF);
    float* se_mean = reinterpret_cast<float*>(smem + TW_SE_OFF);
    float* se_part = se_mean + 256;                  // (unused since the partial sums stay in the wave; the padded vectors live here)
    float* se_h = se_mean + 512;                     // 4 x 36 floats (se_mean: 8 x 36)
    float* se_gate = se_part + 1024 + 128;

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef TW_DEV_STAGGER
    // development: the workgroups of an XCD start the tower TW_DEV_STAGGER x 64 cycles apart (do 32 CUs that ask for the same L2 lines
    // in the same cycle hold each other up?)
    for (int i = 0; i < ((b >> 3) & (TW_DEV_STAGGER_GROUPS - 1)) * TW_DEV_STAGGER; ++i) __builtin_amdgcn_s_sleep(1);
#endif
    const bool is_matrix = wave < 4;
    const int w = wave & 3;                              // both roles: my 32-channel slice of a chunk / my 64 couts

    unsigned long long* trc = (a.trace != nullptr && b == 0 && (wave == 0 || wave == 4) && lane == 0) ? a.trace + (wave >> 2) * 256 : nullptr;
    int trn = 0;
    unsigned long long bwait = 0;
#define TW_STAMP() do { if (trc) trc[trn++] = __builtin_amdgcn_s_memtime(); } while (0)
    TW_STAMP();

    // ---- residual stream tile -> LDS (optionally gated: the first block's SE gate was computed by a previous launch) ----
    const Q8 qx0 = q8_signed(a.blocks[0].qx_inv);    // int8: the first block's stream step
    auto load_board = [&]() {
        if (tid < 6 * T1ROW / 2) {       // zero rows 0, 65 and 66 of both t1 buffers (6 rows of T1ROW halves, as 32-bit words)
            const int rowi = tid / (T1ROW / 2), col = tid % (T1ROW / 2), r3 = rowi % 3;
            reinterpret_cast<uint32_t*>(smem + TW_T1_OFF + (rowi / 3) * TW_T1_BYTES + (r3 == 0 ? 0 : 64 + r3) * T1ROW * 2)[col] = 0u;
        }
        if (x_in_lds) {
            if constexpr (F8) {                      // the stem of this launch left the f16 tile: make its e4m3 copy
                for (int i = tid; i < 64 * 32; i += 512) {
                    const int r = i >> 5, v = i & 31;
                    *reinterpret_cast<uint2*>(xq + r * TW_XQROW + v * 8) = f16x8_to_q8<Q>(*reinterpret_cast<const uint4*>(xs + r * XROW + v * 8), qx0);
                }
            }
            return;
        }
        const half_t* xb = reinterpret_cast<const half_t*>(a.x) + size_t(b) * 64 * C;
        if (a.gate_in == nullptr) {
            for (int i = tid; i < 64 * 32; i += 512) {
                const int r = i >> 5, v = i & 31;
                const uint4 u = *reinterpret_cast<const uint4*>(xb + size_t(r) * C + v * 8);
                *reinterpret_cast<uint4*>(xs + r * XROW + v * 8) = u;
                if constexpr (F8) *reinterpret_cast<uint2*>(xq + r * TW_XQROW + v * 8) = f16x8_to_q8<Q>(u, qx0);
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
                if constexpr (F8)
                    *reinterpret_cast<uint2*>(xq + r * TW_XQROW + v * 8) = f16x8_to_q8<Q>(*reinterpret_cast<const uint4*>(xs + r * XROW + v * 8), qx0);
            }
        }
    };

    half_t* t1 = reinterpret_cast<half_t*>(smem + TW_T1_OFF);
    half_t* t2 = reinterpret_cast<half_t*>(smem + TW_T2_OFF);
    // test hook: the stream tile as it stands behind a workgroup barrier -> block_dump[tile][b] (all 512 threads, both roles call it at
    // the same points; nothing writes the tile before the next barrier)
    auto dump_tile = [&](int tile) {
        if (a.block_dump == nullptr) return;
        half_t* db = reinterpret_cast<half_t*>(a.block_dump) + (size_t(tile) * a.batch + b) * 64 * C;
        for (int i = tid; i < 64 * 32; i += 512) {
            const int r = i >> 5, v = i & 31;
            *reinterpret_cast<uint4*>(db + size_t(r) * C + v * 8) = *reinterpret_cast<const uint4*>(xs + r * XROW + v * 8);
        }
    };

    // The two roles run completely separate control flow (same barrier sequence), so that neither carries the other's
    // register state.  Per block: [SE gate, all threads] then intervals k = -1 .. n, one barrier each:
    //   matrix: E(k+1) -> t1[(k+1)&1], then P(k-1) <- t2[(k-1)&1]        vector: D(k): t1[k&1] -> t2[k&1]
    // C_op is padded to whole chunks of 128 (zero weights), so every chunk is full.
    if (is_matrix) {
#ifdef TW_DEV_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        // open the weight stream first: its window flies while the board tile comes in
        WStream sp;
        sp.base = reinterpret_cast<const char*>(a.wstream) + size_t(w) * a.wstream_wave_frags * 1024;
        sp.lane_off = lane * 16;
        const float* bp = a.bstream + size_t(w) * a.bstream_wave_floats + lh * 16;
        // Precision fp8: two streams (expand loads first, project loads from a.wstream_e_frags on), a window of 8 loads each
        WStream spP = sp;
        spP.base += size_t(a.wstream_e_frags) * 1024u;
        frag win[TW_WIN];
#pragma unroll
        for (int q = 0; q < (F8 ? TW_WIN : TW_DEPTH); ++q) win[q] = (F8 && q >= TW_WIN8) ? spP.frag_at(q - TW_WIN8) : sp.frag_at(q);
        f32x16 bias;                                 // BN1 bias of the next expand phase (matrix_interval): the first MFMAs' C operand
        load_bias(bias, bp);
        load_board();
        __syncthreads();
        dump_tile(0);
        TW_STAMP();
        // per-lane LDS addresses: B-operand fragments (row lane%32 of a 32-square tile, k offset (lane/32)*8), my t1 store slot
        const half_t* xsr = xs + l31 * XROW + lh * 8;
        const int t1off = (1 + l31) * T1ROW + w * 32 + lh * 16;        // + 1: row 0 of a t1 buffer is a zero row
        const int t2off = l31 * T2ROW + lh * 8;
        // Precision fp8: a lane's 32 bytes of a k-step in its row of the e4m3 tiles; window slot of the next phase's first load
        const char* xqr = xq + l31 * TW_XQROW + lh * 32;
        const int t2off8 = l31 * TW_T2ROW8 + lh * 32;
        for (int blk = 0; blk < a.nblocks; ++blk) {
            const TowerBlockDesc& d = a.blocks[blk];
            if (blk > 0 && d.se_kind != 0) se_phase<Q>(d, tid, xs, xq, pool_sum, se_mean, se_part, se_h, se_gate, trc, trn);
            TW_STAMP();
            const int n = d.cop_pad / TW_CK;
            // project accumulators start at the BN3 bias of their cout: row (v%4) + 8*(v/4) + 4*lh of tile rt
            f32x16 accP[2][2];           // [cout tile rt][square tile ct]: couts w*64 + rt*32 + row, squares ct*32 + lane%32
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(d.b3 + w * 64 + rt * 32 + g4 * 8 + lh * 4);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int j = 0; j < 4; ++j) accP[rt][ct][g4 * 4 + j] = bs[j];
                }
            for (int k = -1; k <= n; ++k) {
                half_t* t1w = t1 + ((k + 1) & 1) * (TW_T1_BYTES / 2) + t1off;
                const half_t* t2r = t2 + ((k - 1) & 1) * (TW_T2_BYTES / 2) + t2off;
#ifndef TW_DEV_NO_MATRIX
                if constexpr (F8)
                    matrix_interval8<Q>(k + 1 < n, k >= 1, accP, reinterpret_cast<frag(&)[TW_WIN8]>(win[0]), reinterpret_cast<frag(&)[TW_WIN8]>(win[TW_WIN8]),
                                        sp, spP, bias, bp, xqr, t1w, smem + TW_T2_OFF + ((k - 1) & 1) * TW_T2_BYTES + t2off8, d.escale);
                else
                    matrix_interval(k + 1 < n, k >= 1, accP, win, sp, bias, bp, xsr, t1w, t2r);
#endif
#ifdef TW_TRACE_BARRIERS
                if (trc) {                           // development: time spent waiting at the interval barriers (perturbs the loop)
                    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
                    __syncthreads();
                    bwait += __builtin_amdgcn_s_memtime() - w0;
                } else
#endif
#ifndef TW_DEV_NO_BARRIER
                    __syncthreads();
#else
                    __builtin_amdgcn_sched_barrier(0);   // development (wrong results): the interval barrier's cost
#endif
            }
            if (trc) { trc[trn++] = __builtin_amdgcn_s_memtime() - bwait; bwait = 0; }   // loop end stamp minus barrier waits = busy time
            TW_STAMP();
            // ---- block epilogue: y = x + BN3(project), new residual stream back to LDS ----
            // 4 consecutive couts per step: rows 8*g4 + 4*lh + 0..3 = accumulator elements 4*g4 + 0..3; the f16 residual is read
            // straight out of its packed register by the mix-precision FMA, which also rounds the sum (once, RNE) into place.
            // All residual reads of a cout tile go out before its first store (a store would order the later reads behind it).
            // Precision fp8: the accumulators are in units of the cout's weight scale (they started at b3 / s3): y = x + s3 * acc, and the
            // new stream's e4m3 copy is made from the rounded f16 values.
            mfma_retire(accP[0][0], accP[0][1], accP[1][0], accP[1][1]);      // asm readers below (device_utils.h)
            // int8: the new stream is quantised at the NEXT block's step (a gated next block quantises it again behind its gate)
            const Q8 qxn = q8_signed(a.blocks[blk + 1 < a.nblocks ? blk + 1 : blk].qx_inv);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                uint2 rv[4][2];
                f32x4 sc[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if constexpr (F8) sc[g4] = *reinterpret_cast<const f32x4*>(d.s3 + w * 64 + rt * 32 + g4 * 8 + lh * 4);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        rv[g4][ct] = *reinterpret_cast<const uint2*>(xs + (ct * 32 + l31) * XROW + w * 64 + rt * 32 + g4 * 8 + lh * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int co0 = w * 64 + rt * 32 + g4 * 8 + lh * 4;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        half_t* px = xs + (ct * 32 + l31) * XROW + co0;
                        float t0 = accP[rt][ct][g4 * 4 + 0], t1 = accP[rt][ct][g4 * 4 + 1];
                        float t2 = accP[rt][ct][g4 * 4 + 2], t3 = accP[rt][ct][g4 * 4 + 3];
                        if constexpr (Q == 2) {      // int32 sums (they started at the BN3 bias in their unit): y = x + (step of t2 * row step) * acc
                            t0 = float(__builtin_bit_cast(int, t0)); t1 = float(__builtin_bit_cast(int, t1));
                            t2 = float(__builtin_bit_cast(int, t2)); t3 = float(__builtin_bit_cast(int, t3));
                        }
                        uint2 o;
                        if constexpr (F8) {
                            asm("v_fma_mixlo_f16 %0, %2, %8, %6 op_sel_hi:[0,0,1]\n\t"
                                "v_fma_mixhi_f16 %0, %3, %9, %6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                                "v_fma_mixlo_f16 %1, %4, %10, %7 op_sel_hi:[0,0,1]\n\t"
                                "v_fma_mixhi_f16 %1, %5, %11, %7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                : "=&v"(o.x), "=&v"(o.y)
                                : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(rv[g4][ct].x), "v"(rv[g4][ct].y), "v"(sc[g4][0]), "v"(sc[g4][1]),
                                  "v"(sc[g4][2]), "v"(sc[g4][3]));
                            *reinterpret_cast<uint32_t*>(xq + (ct * 32 + l31) * TW_XQROW + co0) = f16x4_to_q8<Q>(o.x, o.y, qxn);
                        } else {
                            asm("v_fma_mixlo_f16 %0, %2, 1.0, %6 op_sel_hi:[0,0,1]\n\t"
                                "v_fma_mixhi_f16 %0, %3, 1.0, %6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                                "v_fma_mixlo_f16 %1, %4, 1.0, %7 op_sel_hi:[0,0,1]\n\t"
                                "v_fma_mixhi_f16 %1, %5, 1.0, %7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                : "=&v"(o.x), "=&v"(o.y)
                                : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(rv[g4][ct].x), "v"(rv[g4][ct].y));
                        }
                        *reinterpret_cast<uint2*>(px) = o;
                    }
                }
            }
            __syncthreads();
            dump_tile(blk + 1);
            TW_STAMP();
        }
    } else {
        if constexpr (F8) __builtin_amdgcn_s_setprio(2);     // fp8: the depthwise waves bound the interval (measured 1-2 %; no effect in f16)
        VParams vp;
        vp.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.pstream)) + size_t(w) * a.pstream_wave_bytes, 0, 0x7fffffff, 0x00020000);
        vp.pos = 0;
        vp.lane_off = lane * 16;
        vp.buf = 0;
        vp.lds = prm_lds + w * 2 * TW_PRM_BUF;
        vp.fetch(0, 0);                  // the first chunk's weights
        const bool hi = l15 >= 8;        // second board row of a 16-square tile
        const half2_t one2 = {half_t(1.f), half_t(1.f)}, zero2 = {half_t(0.f), half_t(0.f)};
        const int prm_off3 = lg * 48 + ((l15 & 7) == 0 ? 0 : (l15 & 7) == 7 ? 32 : 16);    // 3 x 3 weights: my lane group, my file's variant
        // neighbour rows of my square in tile 0 (buffer rows: 0 = zero row, 1 + sq, 65 and 66 = zero rows); file wrap-around reads
        // a wrong-but-FINITE row that meets a zero weight.  Finite matters: Inf or NaN times the zero weight is a NaN, which the ReLU
        // (v_pk_max_f16 returns the other operand) silently turns into 0 for the whole output.  The 5 x 5 taps reach row -1 (the last
        // 288 bytes in front of the buffer: the stream tile's last row for buffer 0, buffer 0's zero rows for buffer 1) and row 66
        // (h8 + 2 files, h7 + 1 rank + 2 files); without the second zero row that was the first bytes of the t2 region for buffer 1
        // -- f16 activations in Precision float16, but e4m3 bytes and 16 never-written pad bytes per row in Precision fp8
        VecAddr va;
        {
            const char* t1b = smem + TW_T1_OFF;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int row = 1 + l15 + (tap / 3 - 1) * 8 + (tap % 3 - 1);
                va.tap[tap] = t1b + (row * T1ROW + w * 32 + lg * 8) * 2;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                va.top[i] = hi ? va.tap[i] : t1b + (w * 32 + lg * 8) * 2;                                    // row 0: zeros
                va.bot[i] = hi ? t1b + ((65 - 48) * T1ROW + w * 32 + lg * 8) * 2 : va.tap[6 + i];            // + 3 tiles (48 rows) = row 65
            }
        }
        char* t2w = F8 ? smem + TW_T2_OFF + l15 * TW_T2ROW8 + w * 32 + lg * 8 : reinterpret_cast<char*>(t2 + l15 * T2ROW + w * 32 + lg * 8);
        // the same for the 5 x 5 blocks (addresses only: 36 more VGPRs, this role has them to spare)
        VecAddr5 va5;
        half2_t mk5[5];
        {
            const char* t1b = smem + TW_T1_OFF;
            const int col2 = (w * 32 + lg * 8) * 2, file = l15 & 7;
#pragma unroll
            for (int tap = 0; tap < 25; ++tap) va5.tap[tap] = t1b + (1 + l15 + (tap / 5 - 2) * 8 + (tap % 5 - 2)) * T1ROW * 2 + col2;
            va5.zero0 = t1b + col2;
            va5.zero3 = t1b + (65 - 48) * T1ROW * 2 + col2;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                va5.top[i] = hi ? va5.tap[5 + i] : va5.zero0;
                va5.bot[i] = hi ? va5.zero3 : va5.tap[15 + i];
                const int x = file + i - 2;
                mk5[i] = (x >= 0 && x < 8) ? one2 : zero2;
            }
        }
        // L2 warm-up for the matrix waves' weight streams.  All 256 workgroups consume the same 4 streams in near lockstep, so
        // without help every line is a miss-in-flight for everyone (one XCD-L2 fill, 31 requests queued behind it) and the
        // stream runs at miss latency.  The workgroups of an XCD therefore share the job of touching each line TW_AHEAD
        // fragments early: workgroup b (XCD b % 8 under round-robin dispatch -- an assumption about speed only) takes the
        // lines whose index is (b / 8) % 32 modulo 32, one 32-lane load per interval from vector wave 0.
        const char* wsb = reinterpret_cast<const char*>(a.wstream);
        const long long wsF = a.wstream_wave_frags;
        const int pf_slot = (b >> 3) & 31;
        long long mpos = 0, mposP = 0;   // loads the matrix waves have consumed (same arithmetic as theirs; fp8: per stream)
        int pf_old = 0, pf_sink = 0;
        auto prefetch = [&](long long first, int nfr) -> int {   // loads [first, first + nfr) of all 4 waves' streams, nfr = 8, 16 or 32
            int v = 0;
            if (w == 0 && lane < 32) {
                const int idx = lane * 32 + pf_slot, sh = nfr == 32 ? 8 : nfr == 16 ? 7 : 6;
                const int wq = idx >> sh, rem = idx & ((1 << sh) - 1);
                const long long fr = first + (rem >> 3);
                if (wq < 4 && fr < wsF) v = *reinterpret_cast<const int*>(wsb + ((wq * wsF + fr) << 10) + ((rem & 7) << 7));
            }
            return v;
        };
        if constexpr (F8) {
            for (int f0 = 0; f0 < TW_AHEAD / 2; f0 += 16) pf_sink ^= prefetch(f0, 16) ^ prefetch(a.wstream_e_frags + f0, 16);
        } else {
            for (int f0 = 0; f0 < TW_AHEAD; f0 += 32) pf_sink ^= prefetch(f0, 32);
        }
        load_board();
        __syncthreads();
        dump_tile(0);
        TW_STAMP();
        for (int blk = 0; blk < a.nblocks; ++blk) {
            const TowerBlockDesc& d = a.blocks[blk];
            if (blk > 0 && d.se_kind != 0) se_phase<Q>(d, tid, xs, xq, pool_sum, se_mean, se_part, se_h, se_gate, trc, trn);
            TW_STAMP();
            const int n = d.cop_pad / TW_CK;
            // the NEXT block's SE-gate weights (128 KiB, read by every workgroup at the same moment) get the same treatment:
            // each workgroup of an XCD touches 32 of the 1024 lines a whole block early (vector wave 1, one load)
            if (w == 1 && lane < 32 && blk + 1 < a.nblocks && a.blocks[blk + 1].se_kind != 0) {
                const TowerBlockDesc& dn = a.blocks[blk + 1];
                const int line = pf_slot * 32 + lane;                                   // 0..1023
                const char* base = dn.se_kind == 1 && line >= 512 ? reinterpret_cast<const char*>(dn.se_w2) - 512 * 128
                                                                  : reinterpret_cast<const char*>(dn.se_w1);
                pf_sink ^= *reinterpret_cast<const int*>(base + line * 128);
            }
            const Q8 qt = q8_unsigned(d.qt_inv);     // int8: this block's depthwise-output step
            const bool five = __builtin_amdgcn_readfirstlane(d.ks) == 5;    // read ONCE per block: inside the loop the compiler re-loads it
                                                                            // from global memory every interval and waits for vmcnt(0),
                                                                            // i.e. also for the L2 warm-up load it has just issued
            for (int k = -1; k <= n; ++k) {
                // Vector-memory returns retire in order, so a wait for ANY load also waits for every older one.  The warm-up touch is
                // a cold miss by design: it is issued right AFTER the interval's only vmcnt wait (the depthwise weights' DMA of an
                // interval ago, vp.open()) and consumed a whole interval later; issued before that wait it stalled this wave for a full
                // miss latency in every interval.
                pf_sink ^= pf_old;
#ifndef TW_DEV_NO_VECTOR
                const bool work = k >= 0 && k < n;
#else
                const bool work = false;
#endif
                const char* prm_buf = work ? vp.open() : nullptr;
                {
                    if constexpr (F8) {              // two streams: the expand loads, then (from wstream_e_frags on) the project loads
#ifndef TW_DEV_NO_WARMUP
                        pf_old = (k + 1 < n ? prefetch(mpos + TW_AHEAD / 2, 8) : 0) ^ (k >= 1 ? prefetch(a.wstream_e_frags + mposP + TW_AHEAD / 2, 8) : 0);
#endif
                        mpos += k + 1 < n ? 8 : 0;
                        mposP += k >= 1 ? 8 : 0;
                    } else {
                        const int adv = (k + 1 < n ? 16 : 0) + (k >= 1 ? 16 : 0);
#ifndef TW_DEV_NO_WARMUP
                        pf_old = adv != 0 ? prefetch(mpos + TW_AHEAD, adv) : 0;
#endif
                        mpos += adv;
                    }
                }
                if (work) {
                    if (five) {
                        if (k & 1) vector_interval5<1, Q>(vp, prm_buf, lg, va5, t2w, mk5, qt);
                        else vector_interval5<0, Q>(vp, prm_buf, lg, va5, t2w, mk5, qt);
                    } else {
                        if (k & 1) vector_interval<1, Q>(vp, prm_buf, prm_off3, va, t2w, qt);
                        else vector_interval<0, Q>(vp, prm_buf, prm_off3, va, t2w, qt);
                    }
                }
#ifdef TW_TRACE_BARRIERS
                if (trc) {
                    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
                    __syncthreads();
                    bwait += __builtin_amdgcn_s_memtime() - w0;
                } else
#endif
#ifndef TW_DEV_NO_BARRIER
                    __syncthreads();
#else
                    __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (trc) { trc[trn++] = __builtin_amdgcn_s_memtime() - bwait; bwait = 0; }
            TW_STAMP();
            __syncthreads();             // the matrix waves' block epilogue
            dump_tile(blk + 1);
            TW_STAMP();
        }
        pf_sink ^= pf_old;
        asm volatile("" ::"v"(pf_sink));     // keeps the warm-up loads alive; their values are never used
    }

    // ---- residual stream -> HBM; channel sums for an SE gate computed by a later launch ----
    if (y_to_global) {
        half_t* yb = reinterpret_cast<half_t*>(a.y) + size_t(b) * 64 * C;
        for (int i = tid; i < 64 * 32; i += 512) {
            const int r = i >> 5, v = i & 31;
            *reinterpret_cast<uint4*>(yb + size_t(r) * C + v * 8) = *reinterpret_cast<const uint4*>(xs + r * XROW + v * 8);
        }
        if (a.pool_out != nullptr && tid < 256) {        // channel sums for an SE gate computed by a later launch
            float sum = 0.f;
            for (int sq = 0; sq < 64; ++sq) sum += float(xs[sq * XROW + tid]);
            a.pool_out[size_t(b) * C + tid] = sum;
        }
    }
}

#ifndef CRA_FORWARD_TU
__global__ __launch_bounds__(512) void tower_kernel(const TowerArgs a) { tower_body<0>(a, false, true); }
__global__ __launch_bounds__(512) void tower_kernel_fp8(const TowerArgs a) { tower_body<1>(a, false, true); }
__global__ __launch_bounds__(512) void tower_kernel_int8(const TowerArgs a) { tower_body<2>(a, false, true); }

void init_tower_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TW_DYN_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_kernel_fp8), hipFuncAttributeMaxDynamicSharedMemorySize, TW_DYN_LDS_BYTES_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_kernel_int8), hipFuncAttributeMaxDynamicSharedMemorySize, TW_DYN_LDS_BYTES_F8);
}

void launch_tower(const TowerArgs& a, hipStream_t s) {
    if (a.fp8 == 2) hipLaunchKernelGGL(tower_kernel_int8, dim3(a.batch), dim3(512), TW_DYN_LDS_BYTES_F8, s, a);
    else if (a.fp8) hipLaunchKernelGGL(tower_kernel_fp8, dim3(a.batch), dim3(512), TW_DYN_LDS_BYTES_F8, s, a);
    else hipLaunchKernelGGL(tower_kernel, dim3(a.batch), dim3(512), TW_DYN_LDS_BYTES, s, a);
}

#endif  // CRA_FORWARD_TU

}  // namespace cra
