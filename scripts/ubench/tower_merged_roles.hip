// EXPERIMENT RECORD (round 2) -- not part of the library, not compiled by crazyara_amd/build.py.
//
// Residual tower with merged roles: the tower of crazyara_amd/csrc/nn/tower.hip with FOUR waves per workgroup instead of eight.
// Wave w runs the matrix role (expand rows / project couts w) AND the vector role (depthwise of the same 32 channels) in ONE
// instruction stream; at one wave per SIMD a lane owns up to 512 registers (256 VGPRs + 256 AGPRs).
//
// Hypothesis: scripts/ubench/merged_wave.hip measures 4 MFMAs + 2 weight loads + 4 B-fragment reads + 12 packed FMAs of ONE wave at
// 171 cycles, while the shipped kernel's step (the same work split over two waves of a SIMD) takes 244 -- so one wave doing both
// roles with a hand-written instruction order (16 groups of 4 MFMAs per interval, a group = four slots {MFMA, <= 4 VALU}, a
// scheduling fence after every slot) should bring the interval from 3.9k to about 2.8k cycles.
//
// Result (profiles/r02/k_merged_roles_experiment.txt): SLOWER.  Tower alone, RISEv2-19, batch 256: 0.388-0.401 ms against 0.293-0.297 ms
// for the 8-wave kernel (batch 512: 0.766 against 0.568), i.e. about 5.1k cycles per interval.  Why, from the ISA of the steady-state
// interval: beside its 64 MFMAs the single stream carries ~570 other instructions (200 packed FMA/MUL, 85 ds_read, 36 loads, 70
// waits, 100 v_accvgpr_read/write that move the project accumulators and the expand results between the register files because VALU
// instructions cannot address AGPRs, ...).  A wave issues one instruction per 4-5 cycles and an MFMA that is not yet allowed to issue
// blocks everything behind it, so 570 x 4.5 cycles do not fit under 64 x 32 MFMA cycles: the two-wave form wins because the SIMD
// issues the two streams independently.  The microbenchmark's 171 cycles hold only for its 18 extra instructions per 4 MFMAs; the
// real interval has 36.
//
// What was learned on the way (kept for whoever tries again):
//   * the compiler keeps a loop-carried weight window in AGPRs only as a spill area (load into VGPRs, v_accvgpr_write at the top
//     of the next interval => vmcnt(0) there); loads written as inline assembly with an "a" output constraint do go straight to
//     AGPRs and feed the MFMA A operand from there;
//   * BUT a value produced by an inline-assembly load must not be copied before its hand-written s_waitcnt: the register allocator
//     inserts v_accvgpr_mov / v_mov copies at merge points of the 12 interval variants, and a copy made before the load has landed
//     copies stale data.  That is the state of this file: with the BN1-bias loads as assembly the outputs are wrong everywhere; with
//     every load compiler-visible (-DTWM_PLAIN_LOADS) a 3-block net agrees to 7.5e-4 on the probabilities and a 19-block net still
//     does not.  The timing above does not depend on which variant runs; parity was NOT reached and was not pursued further once
//     the timing was known;
//   * LDS-DMA through inline assembly (s_mov m0 + buffer_load ... lds) works and avoids the vmcnt(0) the builtin forces before the
//     next LDS access and at every barrier.
//
// To rerun: put the file back beside tower.hip as towerm.hip, declare launch_tower_m / init_tower_m_kernel_attributes in kernels.h,
// route OpKind::Tower to launch_tower_m for runs whose blocks are all 3 x 3 under a precision suffix ("-m3k" was used, with
// one_launch_ = false), and run scripts/ubench/tower_merged_roles_check.py on the GPU box.
//
#define CRA_FORWARD_TU 1
#include "tower.hip"
#undef CRA_FORWARD_TU

namespace cra {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// two LDS-DMA loads (64 lanes x 16 B each) of the 2 KiB at byte `pos` of the stream into LDS at byte address `lds_addr`
__device__ __forceinline__ void dma_chunk_params(i32x4 rsrc, uint32_t lds_addr, uint32_t lane_off, uint32_t pos) {
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                 "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
                 :
                 : "s"(lds_addr), "v"(lane_off), "s"(rsrc), "s"(pos)
                 : "memory", "m0");
}

#define TWM_FENCE() __builtin_amdgcn_sched_barrier(0)

// The weight window lives in accumulation registers (AGPRs): a fragment is only ever written by its load and read as the A operand
// of MFMAs, both of which address AGPRs directly, and the 256 architectural VGPRs are needed by the depthwise.  The compiler does
// not make that choice by itself (it loads into VGPRs and copies, which also costs a full vmcnt drain per interval), so the loads
// are inline assembly with an "a" constraint -- and because the compiler cannot count loads it does not see, every vector-memory
// wait of the interval loop is written by hand:
//   * the stream is consumed strictly in order and every consumed slot is refilled at once, so the load of the fragment a slot
//     holds is always followed by at least 15 younger loads by the time it is needed: s_waitcnt vmcnt(15) (returns retire in order;
//     the BN1-bias loads and the depthwise DMA in between only make the wait stricter than necessary);
//   * the wait names the fragment as an in/out operand, so the MFMAs that read it cannot be moved above it.
struct WStreamA {
    i32x4 rs;
    uint32_t pos;        // byte position of the window start (wave-uniform)
    uint32_t lane_off;   // lane * 16
};
#ifdef TWM_PLAIN_LOADS      // development: compiler-visible loads (VGPR window, compiler-made waits), to tell a wait bug from a data-flow bug
__device__ __forceinline__ void aload(half8& dst, const WStreamA& sp, int q) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(uint64_t(uint32_t(sp.rs[0])) | (uint64_t(uint32_t(sp.rs[1])) << 32)), 0, 0x7fffffff, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, sp.lane_off, sp.pos + uint32_t(q) * 1024u, 0);
    dst = __builtin_bit_cast(half8, v);
}
__device__ __forceinline__ void await_frag(half8&) {}
#else
__device__ __forceinline__ void aload(half8& dst, const WStreamA& sp, int q) {          // slot <- fragment q positions after the window start
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(dst) : "v"(sp.lane_off), "s"(sp.rs), "s"(sp.pos + uint32_t(q) * 1024u));
}
__device__ __forceinline__ void await_frag(half8& f) { asm volatile("s_waitcnt vmcnt(15)" : "+a"(f)); }
#endif

// (a + ca, b + cb) -> ReLU -> packed f16 pair: the f32 sums are rounded once
__device__ __forceinline__ uint32_t pack_bias_relu(float a, float ca, float b, float cb) {
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2\n\tv_fma_mixhi_f16 %0, %3, 1.0, %4\n\tv_pk_max_f16 %0, %0, 0" : "=&v"(r) : "v"(a), "v"(ca), "v"(b), "v"(cb));
    return r;
}

// One interval of wave w: E(chunk k+1), D(chunk k), P(chunk k-1), whichever exist.
//   groups 0..7  : expand k-step pairs (4 MFMAs each), groups 8..15: project (k-step 2s + kk: 4 MFMAs each)
//   depthwise    : group 0 = file masks on the weights, tile t = groups 1+3t .. 3+3t (top row, middle row, bottom row + ReLU + store)
//   expand epilogue (BN1 bias is in the accumulator, ReLU, f16, store to t1): one pack per slot in groups 8..11
template <bool DO_E, bool DO_D, bool DO_P, int PARITY>
__device__ __forceinline__ void merged_interval(f32x16 (&accP)[2][2], half8 (&win)[TW_WIN], WStreamA& sp,
                                                const float* __restrict__& bp, const half_t* xsr, half_t* t1w, const half_t* t2r,
                                                const char* prm, const VecAddr& va, half_t* t2w, half2_t mLp, half2_t mRp) {
    using frag = half8;
    constexpr int XROW = TW_XROW, T1ROW = TW_T1ROW, T2ROW = TW_T2ROW;
    constexpr int TILE = 16 * T1ROW * 2;             // bytes between square tiles of a t1 buffer
    constexpr int PB = PARITY * TW_T1_BYTES;

    // ---- depthwise state: three row slots S (role r of tile t sits in slot (r + 2t) % 3: the bottom row of a tile is the top row of
    // the next, and the next tile's middle / bottom rows are read into the slots of this tile's top / middle rows as soon as those
    // have been multiplied), weights, accumulators ----
    uint4 S[3][3];
    half2_t W[10][4];
    half2_t acc[4];
    if constexpr (DO_D) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // my chunk's weights have landed (file header)
#pragma unroll
        for (int e = 0; e < 10; ++e) {
            const uint4 u = *reinterpret_cast<const uint4*>(prm + e * TW_PRM_ENT);
            W[e][0] = __builtin_bit_cast(half2_t, u.x); W[e][1] = __builtin_bit_cast(half2_t, u.y);
            W[e][2] = __builtin_bit_cast(half2_t, u.z); W[e][3] = __builtin_bit_cast(half2_t, u.w);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            S[0][i] = *reinterpret_cast<const uint4*>(va.top[i] + PB);
            S[1][i] = *reinterpret_cast<const uint4*>(va.tap[3 + i] + PB);
            S[2][i] = *reinterpret_cast<const uint4*>(va.tap[6 + i] + PB);
        }
    }
    auto vec_reads = [&](int g) {                    // before the MFMAs of group g: rows of the next tile into slots that are free
#ifdef TWM_SEQ_VECTOR
        return;
#endif
        if constexpr (DO_D) {
            if (g >= 2 && g <= 9 && (g - 1) % 3 != 0) {
                const int t = (g - 1) / 3, r = (g - 1) % 3;     // tile t is in its row r (1: middle, 2: bottom); rows < r are done
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (r == 1)                      // next tile's middle row -> this tile's top slot
                        S[(0 + 2 * t) % 3][i] = *reinterpret_cast<const uint4*>(va.tap[3 + i] + PB + (t + 1) * TILE);
                    else                             // next tile's bottom row -> this tile's middle slot
                        S[(1 + 2 * t) % 3][i] = *reinterpret_cast<const uint4*>((t == 2 ? va.bot[i] : va.tap[6 + i]) + PB + (t + 1) * TILE);
                }
            }
        }
    };
    auto vec_piece = [&](int g, int p) {             // the depthwise VALU work of slot p of group g
#ifdef TWM_SEQ_VECTOR
        return;
#endif
        if constexpr (DO_D) {
            if (g == 0) {                            // file a has no left neighbour, file h no right neighbour: 6 of the 24 products
                constexpr int taps[6] = {0, 3, 6, 2, 5, 8};
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int f = p * 6 + q, tap = taps[f >> 2], pi = f & 3;
                    W[tap][pi] *= (f >> 2) < 3 ? mLp : mRp;
                }
            } else if (g <= 12) {
                const int t = (g - 1) / 3, r = (g - 1) % 3;      // tile, row of taps
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int f = p * 3 + q, i = f >> 2, pi = f & 3;
                    const half2_t x = __builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&S[(r + 2 * t) % 3][i])[pi]);
                    acc[pi] = __builtin_elementwise_fma(x, W[r * 3 + i][pi], (r == 0 && i == 0) ? W[9][pi] : acc[pi]);
                }
                if (r == 2 && p == 3) {
                    uint32_t o[4];
#pragma unroll
                    for (int pi = 0; pi < 4; ++pi) o[pi] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(acc[pi], half2_t{0, 0}));
                    *reinterpret_cast<uint4*>(t2w + PARITY * (TW_T2_BYTES / 2) + t * 16 * T2ROW) = uint4{o[0], o[1], o[2], o[3]};
                }
            }
        }
    };

    // ---- matrix state ----
    f32x16 accE[2];                                  // [square tile of 32]
    frag bA[4], bB[4];                               // B fragments of the step being multiplied / of the next step
    auto read_e = [&](frag (&dst)[4], int s) {       // expand step s: [k-step parity][square tile]
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const frag*>(xsr + (i & 1) * 32 * XROW + (s * 2 + (i >> 1)) * 16);
    };
    auto read_p = [&](frag (&dst)[4], int s) {       // project step s: [k-step parity][square tile]
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const frag*>(t2r + (i & 1) * 32 * T2ROW + (s * 2 + (i >> 1)) * 16);
    };
    f32x4 bias[4];                                   // BN1 bias of my 16 rows (v%4) + 8*(v/4) + 4*(lane/32): loaded in group 0, added in
                                                     // the epilogue (>= 16 window loads later)
    if constexpr (DO_E) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) accE[ct][v] = 0.f;
        read_e(bA, 0);
    } else if constexpr (DO_P) {
        read_p(bA, 0);
    }
    uint32_t eo[4];                                  // packed expand outputs on their way to t1
    TWM_FENCE();

#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const bool e_group = g < 8;
        const int s = e_group ? g : (g - 8) >> 1, kk = (g - 8) & 1;
        const int u = e_group ? g : 8 + s;           // step counter over both phases: the B-fragment buffers alternate per step
        frag (&cur)[4] = (u & 1) ? bB : bA;
        frag (&nxt)[4] = (u & 1) ? bA : bB;
        // ---- LDS reads that must be in flight before this group's MFMAs ----
        if (e_group) {
            if constexpr (DO_E) {
                if (g + 1 < 8) read_e(nxt, g + 1);
                else if constexpr (DO_P) read_p(nxt, 0);
            }
        } else if constexpr (DO_P) {
            if (kk == 0 && s + 1 < 4) read_p(nxt, s + 1);
        }
        vec_reads(g);
        TWM_FENCE();
        // ---- four slots ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (e_group) {
                if constexpr (DO_E) {
                    if (g == 0 && i == 0) {          // this phase's BN1 bias: in front of the phase's 16 refills
#ifdef TWM_PLAIN_LOADS
#pragma unroll
                        for (int q = 0; q < 4; ++q) bias[q] = reinterpret_cast<const f32x4*>(bp)[q];
#else
                        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                                     "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                                     : "=&v"(bias[0]), "=&v"(bias[1]), "=&v"(bias[2]), "=&v"(bias[3]) : "v"(bp));
#endif
                        bp += 32;
                    }
                    if ((i & 1) == 0) await_frag(win[s * 2 + (i >> 1)]);
                    mma32(win[s * 2 + (i >> 1)], cur[i], accE[i & 1]);
                    if (i & 1) aload(win[s * 2 + (i >> 1)], sp, s * 2 + (i >> 1) + TW_WIN);
                }
            } else {
                if constexpr (DO_P) {
                    const int rt = i >> 1, ct = i & 1;
                    if (ct == 0) await_frag(win[(s * 2 + kk) * 2 + rt]);
                    mma32(win[(s * 2 + kk) * 2 + rt], cur[kk * 2 + ct], accP[rt][ct]);
                    if (ct == 1) aload(win[(s * 2 + kk) * 2 + rt], sp, (s * 2 + kk) * 2 + rt + TW_WIN);
                    if constexpr (DO_E) {
                        if (g < 12) {                // expand epilogue of square tile (g - 8) / 2: 8 packs over 8 slots, then 2 stores
                            const int ct2 = (g - 8) >> 1, hf = (g - 8) & 1, j = hf * 4 + i;
#ifndef TWM_PLAIN_LOADS
                            if (g == 8 && i == 0)    // the bias loads of group 0 are 16 refills old
                                asm volatile("s_waitcnt vmcnt(15)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]));
#endif
                            eo[i] = pack_bias_relu(accE[ct2][2 * j], bias[j >> 1][(2 * j) & 3], accE[ct2][2 * j + 1], bias[j >> 1][(2 * j + 1) & 3]);
                            if (i == 3) reinterpret_cast<uint4*>(t1w + ct2 * 32 * T1ROW)[hf] = uint4{eo[0], eo[1], eo[2], eo[3]};
                        }
                    }
                }
            }
            vec_piece(g, i);
            TWM_FENCE();
        }
        if (g == 7) {
            if constexpr (DO_E) {
                sp.pos += 16 * 1024;
                if constexpr (!DO_P) {               // no project phase to hide it in: the epilogue right here
#ifdef TWM_PLAIN_LOADS
                    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#else
                    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(15)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]) : : "memory");
#endif
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int jj = hf * 4 + j;
                                eo[j] = pack_bias_relu(accE[ct][2 * jj], bias[jj >> 1][(2 * jj) & 3], accE[ct][2 * jj + 1], bias[jj >> 1][(2 * jj + 1) & 3]);
                            }
                            reinterpret_cast<uint4*>(t1w + ct * 32 * T1ROW)[hf] = uint4{eo[0], eo[1], eo[2], eo[3]};
                        }
                }
            }
        }
    }
    if constexpr (DO_P) sp.pos += 16 * 1024;
#ifdef TWM_SEQ_VECTOR       // development: tower.hip's depthwise loop after the matrix groups (tells a depthwise bug from a matrix bug)
    if constexpr (DO_D) {
        uint4 top[3], mid[3], bot[3];
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            W[0][pi] *= mLp; W[3][pi] *= mLp; W[6][pi] *= mLp;
            W[2][pi] *= mRp; W[5][pi] *= mRp; W[8][pi] *= mRp;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                top[i] = *reinterpret_cast<const uint4*>((t == 0 ? va.top[i] : va.tap[i]) + PB + t * TILE);
                mid[i] = *reinterpret_cast<const uint4*>(va.tap[3 + i] + PB + t * TILE);
                bot[i] = *reinterpret_cast<const uint4*>((t == 3 ? va.bot[i] : va.tap[6 + i]) + PB + t * TILE);
            }
            half2_t a4[4] = {W[9][0], W[9][1], W[9][2], W[9][3]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) {
                    a4[pi] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&top[i])[pi]), W[i][pi], a4[pi]);
                    a4[pi] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&mid[i])[pi]), W[3 + i][pi], a4[pi]);
                    a4[pi] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, reinterpret_cast<const uint32_t*>(&bot[i])[pi]), W[6 + i][pi], a4[pi]);
                }
            uint32_t o[4];
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) o[pi] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(a4[pi], half2_t{0, 0}));
            *reinterpret_cast<uint4*>(t2w + PARITY * (TW_T2_BYTES / 2) + t * 16 * T2ROW) = uint4{o[0], o[1], o[2], o[3]};
        }
    }
#endif
}

// SE gate of a block with 256 threads: tower.hip's se_phase, every thread playing threads tid and tid + 256 of its 512
__device__ __forceinline__ void se_phase_m(const TowerBlockDesc& d, int tid, half_t* xs, float* se_mean, float* se_part, float* se_h,
                                           float* se_gate) {
    constexpr int XROW = TW_XROW;
    half2_t wa[2][32], wb[2][32];
    auto load_thread_weights = [&](const void* base, int vt, half2_t (&dst)[32]) {
        const uint4* pk = reinterpret_cast<const uint4*>(base) + vt;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 u = pk[i * 512];
            dst[4 * i + 0] = __builtin_bit_cast(half2_t, u.x); dst[4 * i + 1] = __builtin_bit_cast(half2_t, u.y);
            dst[4 * i + 2] = __builtin_bit_cast(half2_t, u.z); dst[4 * i + 3] = __builtin_bit_cast(half2_t, u.w);
        }
    };
#pragma unroll
    for (int v = 0; v < 2; ++v) load_thread_weights(d.se_w1, tid + 256 * v, wa[v]);
#pragma unroll
    for (int v = 0; v < 2; ++v) {                    // squeeze
        const int vt = tid + 256 * v, lane = vt & 63, wv = vt >> 6, cg = lane >> 4, sg = lane & 15;
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
            sum[j] += dpp_mov<0x111>(sum[j]);
            sum[j] += dpp_mov<0x112>(sum[j]);
            sum[j] += dpp_mov<0x114>(sum[j]);
            sum[j] += dpp_mov<0x118>(sum[j]);
        }
        if (sg == 15) {
#pragma unroll
            for (int j = 0; j < 8; ++j) se_mean[wv * 32 + cg * 8 + j] = sum[j] * (1.f / 64.f);
        }
    }
    __syncthreads();
    if (d.se_kind == 1) {            // ca_se
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int vt = tid + 256 * v, j2 = vt & 63, kq = vt >> 6;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const float m = se_mean[kq * 32 + k];
                s0 = fmaf(float(wa[v][k][0]), m, s0);
                s1 = fmaf(float(wa[v][k][1]), m, s1);
            }
            load_thread_weights(d.se_w2, vt, wb[v]);
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
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int vt = tid + 256 * v, c2 = vt & 127, kq = vt >> 7;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const float h = se_h[kq * 32 + k];
                s0 = fmaf(float(wb[v][k][0]), h, s0);
                s1 = fmaf(float(wb[v][k][1]), h, s1);
            }
            se_part[kq * 256 + 2 * c2] = s0;
            se_part[kq * 256 + 2 * c2 + 1] = s1;
        }
        __syncthreads();
        se_gate[tid] = hard_sigmoid(se_part[tid] + se_part[256 + tid] + se_part[512 + tid] + se_part[768 + tid]);
    } else {                         // eca_se
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int vt = tid + 256 * v, c2 = vt & 127, kq = vt >> 7;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const float m = se_mean[kq * 64 + k];
                s0 = fmaf(float(wa[v][k][0]), m, s0);
                s1 = fmaf(float(wa[v][k][1]), m, s1);
            }
            load_thread_weights(reinterpret_cast<const char*>(d.se_w1) + 8 * 512 * 16, vt, wb[v]);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const float m = se_mean[kq * 64 + 32 + k];
                s0 = fmaf(float(wb[v][k][0]), m, s0);
                s1 = fmaf(float(wb[v][k][1]), m, s1);
            }
            se_part[kq * 256 + 2 * c2] = s0;
            se_part[kq * 256 + 2 * c2 + 1] = s1;
        }
        __syncthreads();
        se_gate[tid] = hard_sigmoid(d.se_b[tid] + se_part[tid] + se_part[256 + tid] + se_part[512 + tid] + se_part[768 + tid]);
    }
    __syncthreads();
    for (int i = tid; i < 64 * 32; i += 256) {       // x := x * gate
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

}  // namespace

__global__ __launch_bounds__(256) void tower_kernel_m(const TowerArgs a) {
    using frag = half8;
    constexpr int C = TW_C, XROW = TW_XROW, T1ROW = TW_T1ROW, T2ROW = TW_T2ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) char prm_lds[TW_PRM_BYTES];
    half_t* xs = reinterpret_cast<half_t*>(smem);
    float* se_mean = reinterpret_cast<float*>(smem + TW_SE_OFF);
    float* se_part = se_mean + 256;
    float* se_h = se_part + 1024;
    float* se_gate = se_h + 128;

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4, l31 = lane & 31, lh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- streams: depthwise weights (LDS-DMA, first chunk now), weight window, BN1 bias ----
    const char* psb = reinterpret_cast<const char*>(a.pstream) + size_t(w) * a.pstream_wave_bytes;
    const uint64_t psa = reinterpret_cast<uint64_t>(psb);
    const i32x4 prs = {__builtin_amdgcn_readfirstlane(int(uint32_t(psa))), __builtin_amdgcn_readfirstlane(int(uint32_t(psa >> 32) & 0xffffu)),
                       0x7fffffff, 0x00020000};
    const uint32_t prm_base = uint32_t(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)(prm_lds + w * 2 * TW_PRM_BUF)));
    const uint32_t prm_addr = __builtin_amdgcn_readfirstlane(prm_base);
    uint32_t ppos = 0, pbuf = 0;                     // stream position / LDS buffer of the chunk D computes next
    dma_chunk_params(prs, prm_addr, lane * 16, 0);
    WStreamA sp;
    {
        const uint64_t wsa = reinterpret_cast<uint64_t>(reinterpret_cast<const char*>(a.wstream) + size_t(w) * a.wstream_wave_frags * 1024);
        sp.rs = i32x4{__builtin_amdgcn_readfirstlane(int(uint32_t(wsa))), __builtin_amdgcn_readfirstlane(int(uint32_t(wsa >> 32) & 0xffffu)),
                      0x7fffffff, 0x00020000};
    }
    sp.pos = 0;
    sp.lane_off = lane * 16;
    const float* bp = a.bstream + size_t(w) * a.bstream_wave_floats + lh * 16;
    frag win[TW_WIN];
#pragma unroll
    for (int q = 0; q < TW_WIN; ++q) aload(win[q], sp, q);

    // ---- residual stream tile -> LDS ----
    for (int i = tid; i < 4 * T1ROW / 2; i += 256) { // zero rows 0 and 65 of both t1 buffers
        const int rowi = i / (T1ROW / 2), col = i % (T1ROW / 2);
        reinterpret_cast<uint32_t*>(smem + TW_T1_OFF + (rowi >> 1) * TW_T1_BYTES + (rowi & 1) * 65 * T1ROW * 2)[col] = 0u;
    }
    {
        const half_t* xb = reinterpret_cast<const half_t*>(a.x) + size_t(b) * 64 * C;
        if (a.gate_in == nullptr) {
            for (int i = tid; i < 64 * 32; i += 256) {
                const int r = i >> 5, v = i & 31;
                *reinterpret_cast<uint4*>(xs + r * XROW + v * 8) = *reinterpret_cast<const uint4*>(xb + size_t(r) * C + v * 8);
            }
        } else {
            const float* gt = a.gate_in + size_t(b) * C;
            for (int i = tid; i < 64 * 32; i += 256) {
                const int r = i >> 5, v = i & 31;
                float xv[8], gv[8];
                load8<half_t>(xb + size_t(r) * C + v * 8, xv);
                load8<float>(gt + v * 8, gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] *= gv[j];
                store8<half_t>(xs + r * XROW + v * 8, xv);
            }
        }
    }
    __syncthreads();

    half_t* t1 = reinterpret_cast<half_t*>(smem + TW_T1_OFF);
    half_t* t2 = reinterpret_cast<half_t*>(smem + TW_T2_OFF);
    // matrix role addresses
    const half_t* xsr = xs + l31 * XROW + lh * 8;
    const int t1off = (1 + l31) * T1ROW + w * 32 + lh * 16;
    const int t2off = l31 * T2ROW + lh * 8;
    // vector role addresses
    const bool hi = l15 >= 8;
    const half2_t one2 = {half_t(1.f), half_t(1.f)}, zero2 = {half_t(0.f), half_t(0.f)};
    const half2_t mLp = (l15 & 7) != 0 ? one2 : zero2, mRp = (l15 & 7) != 7 ? one2 : zero2;
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
            va.top[i] = hi ? va.tap[i] : t1b + (w * 32 + lg * 8) * 2;
            va.bot[i] = hi ? t1b + ((65 - 48) * T1ROW + w * 32 + lg * 8) * 2 : va.tap[6 + i];
        }
    }
    half_t* t2w = t2 + l15 * T2ROW + w * 32 + lg * 8;

    for (int blk = 0; blk < a.nblocks; ++blk) {
        const TowerBlockDesc& d = a.blocks[blk];
        if (blk > 0 && d.se_kind != 0) se_phase_m(d, tid, xs, se_mean, se_part, se_h, se_gate);
        const int n = __builtin_amdgcn_readfirstlane(d.cop_pad / TW_CK);
        f32x16 accP[2][2];
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
            const bool do_e = k + 1 < n, do_d = k >= 0 && k < n, do_p = k >= 1;
            const char* prm = prm_lds + (w * 2 + pbuf) * TW_PRM_BUF + lg * 16;
            if (do_d) {                              // the NEXT chunk's depthwise weights go out first: >= 16 window loads follow
                ppos += 2048;
                pbuf ^= 1;
                dma_chunk_params(prs, prm_addr + pbuf * TW_PRM_BUF, lane * 16, ppos);
            }
#define TWM_CALL(E, D, P, PAR) merged_interval<E, D, P, PAR>(accP, win, sp, bp, xsr, t1w, t2r, prm, va, t2w, mLp, mRp)
            if (do_e && do_d && do_p) { if (k & 1) TWM_CALL(true, true, true, 1); else TWM_CALL(true, true, true, 0); }
            else if (do_e && do_d) { if (k & 1) TWM_CALL(true, true, false, 1); else TWM_CALL(true, true, false, 0); }
            else if (do_d && do_p) { if (k & 1) TWM_CALL(false, true, true, 1); else TWM_CALL(false, true, true, 0); }
            else if (do_e) TWM_CALL(true, false, false, 0);
            else if (do_p) TWM_CALL(false, false, true, 0);
            else if (do_d) { if (k & 1) TWM_CALL(false, true, false, 1); else TWM_CALL(false, true, false, 0); }
#undef TWM_CALL
            __syncthreads();
        }
        // ---- block epilogue: y = x + BN3(project), new residual stream back to LDS (tower.hip) ----
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            uint2 rv[4][2];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    rv[g4][ct] = *reinterpret_cast<const uint2*>(xs + (ct * 32 + l31) * XROW + w * 64 + rt * 32 + g4 * 8 + lh * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co0 = w * 64 + rt * 32 + g4 * 8 + lh * 4;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    half_t* px = xs + (ct * 32 + l31) * XROW + co0;
                    const float t0 = accP[rt][ct][g4 * 4 + 0], t1v = accP[rt][ct][g4 * 4 + 1];
                    const float t2v = accP[rt][ct][g4 * 4 + 2], t3 = accP[rt][ct][g4 * 4 + 3];
                    uint2 o;
                    asm("v_fma_mixlo_f16 %0, %2, 1.0, %6 op_sel_hi:[0,0,1]\n\t"
                        "v_fma_mixhi_f16 %0, %3, 1.0, %6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                        "v_fma_mixlo_f16 %1, %4, 1.0, %7 op_sel_hi:[0,0,1]\n\t"
                        "v_fma_mixhi_f16 %1, %5, 1.0, %7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(o.x), "=&v"(o.y)
                        : "v"(t0), "v"(t1v), "v"(t2v), "v"(t3), "v"(rv[g4][ct].x), "v"(rv[g4][ct].y));
                    *reinterpret_cast<uint2*>(px) = o;
                }
            }
        }
        __syncthreads();
    }
    // the last DMA (the padding chunk behind the stream) must not outlive the workgroup's LDS allocation
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- residual stream -> HBM; channel sums for an SE gate computed by a later launch ----
    half_t* yb = reinterpret_cast<half_t*>(a.y) + size_t(b) * 64 * C;
    for (int i = tid; i < 64 * 32; i += 256) {
        const int r = i >> 5, v = i & 31;
        *reinterpret_cast<uint4*>(yb + size_t(r) * C + v * 8) = *reinterpret_cast<const uint4*>(xs + r * XROW + v * 8);
    }
    if (a.pool_out != nullptr) {
        float sum = 0.f;
        for (int sq = 0; sq < 64; ++sq) sum += float(xs[sq * XROW + tid]);
        a.pool_out[size_t(b) * C + tid] = sum;
    }
}

void init_tower_m_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_kernel_m), hipFuncAttributeMaxDynamicSharedMemorySize, TW_DYN_LDS_BYTES);
}

void launch_tower_m(const TowerArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(tower_kernel_m, dim3(a.batch), dim3(256), TW_DYN_LDS_BYTES, s, a);
}

}  // namespace cra
