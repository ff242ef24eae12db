// Policy + value head kernel for gfx950: everything after the residual tower in ONE launch, one workgroup per board.
//
// Reference semantics: _PolicyHead (select_policy_from_plane) and _ValueHead
// (DeepCrazyhouse/src/domain/neural_net/architectures/pytorch/builder_util.py:206-243, 246-326), the softmax the GPU back
// end appends to policy_out (engine/src/nn/tensorrtapi.cpp:378-392; formula of apply_softmax, engine/src/nn/neuralnetapi.cpp:241-260).
//
//   phase 1  policy conv 3x3 256->256 + BN + ReLU as 9 shifted GEMMs on v_mfma_f32_32x32x16_f16: wave v owns couts 32v..32v+31
//            for all 64 squares, A = its weight stream ([tap][k-step] fragments), B = rows of the board tile in LDS (neighbour
//            square or the zero row); result -> f16 tile P1 in LDS.  Wave 0 also runs the value head's 1x1 conv (8 couts).
//   phase 2  policy conv 3x3 256->P (P <= 96): the 144 (tap, k-step) units are dealt 18 per wave, each wave accumulates all
//            P x 64 partial logits of its units; the 8 partial sets are summed through LDS, one row tile at a time.
//   phase 3  softmax over the P*64 logits in LDS; the probabilities go to HBM once (the logits too, on request; a search lane takes
//            only the gathered priors of its legal moves).
//   phase 4  value head: FC(512->256)+ReLU -> FC(256->1) -> tanh, or the WDLP outputs.
// The board tile, P1 and the logits never leave the CU; HBM traffic per board is 32 KB in and 2 * P*256 B + 4 B out.
#include "kernels.h"
#include "device_utils.h"

namespace cra {

namespace {
constexpr int HD_C = 256;
constexpr int HD_ROW = HD_C + 8;                     // halves; 528-byte pitch: 32 consecutive rows hit distinct 16-byte bank slots
constexpr int HD_TILE_BYTES = 65 * HD_ROW * 2;       // 64 squares + a zero row
constexpr int HD_X_OFF = 0;
constexpr int HD_P1_OFF = HD_TILE_BYTES;
constexpr int HD_LOGIT_OFF = 2 * HD_TILE_BYTES;      // float [96][64]
constexpr int HD_VFLAT_OFF = HD_LOGIT_OFF + 96 * 64 * 4;      // float [4][132]: value conv output, channel-major flat, 128 per row
constexpr int HD_FC_OFF = HD_VFLAT_OFF + 528 * 4;    // (spare: the FC1 partial sums stay in the wave)
constexpr int HD_RED_OFF = HD_FC_OFF + 5 * 256 * 4;  // float [16] reductions
constexpr int HD_LDS_BYTES = HD_RED_OFF + 64;
constexpr int HD_WIN = 16;

typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pack_relu_h2(float a, float ca, float b, float cb) {
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2\n\tv_fma_mixhi_f16 %0, %3, 1.0, %4\n\tv_pk_max_f16 %0, %0, 0" : "=&v"(r) : "v"(a), "v"(ca), "v"(b), "v"(cb));
    return r;
}
__device__ __forceinline__ void hd_mma32(const half8& a, const half8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// LDS row of the neighbour of square sq for tap (dy, dx), or the zero row 64
__device__ __forceinline__ int nbr_row(int sq, int dy, int dx) {
    const int y = (sq >> 3) + dy, x = (sq & 7) + dx;
    return (unsigned(y) < 8u && unsigned(x) < 8u) ? sq + dy * 8 + dx : 64;
}
// flat index k of the value conv output -> float index in LDS: rows of 128 padded to 132, so that the four lanes of an FC1 group
// (k = 128 * (lane % 4) + ...) read their 16 bytes from distinct banks
__device__ __forceinline__ int vflat_at(int k) { return k + 4 * (k >> 7); }
__device__ __forceinline__ float block_reduce_512(float v, float* red, bool is_max) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}
}  // namespace

#ifndef CRA_FORWARD_TU
size_t head_lds_bytes() { return HD_LDS_BYTES; }
#endif

// x_in_lds: the board tile is already at offset 0 of the dynamic LDS segment ([64][HD_ROW] f16, the tower's residual-stream tile)
__device__ __forceinline__ void head_body(const HeadArgs& a, const bool x_in_lds) {
    using frag = half8;
    constexpr int ROW = HD_ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* X = reinterpret_cast<half_t*>(smem + HD_X_OFF);
    half_t* P1 = reinterpret_cast<half_t*>(smem + HD_P1_OFF);
    float* logit = reinterpret_cast<float*>(smem + HD_LOGIT_OFF);
    float* vflat = reinterpret_cast<float*>(smem + HD_VFLAT_OFF);
    float* red = reinterpret_cast<float*>(smem + HD_RED_OFF);

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    unsigned long long* trc = (a.trace != nullptr && b == 0 && tid == 0) ? a.trace : nullptr;
    int trn = 0;
#define HD_STAMP() do { if (trc) trc[trn++] = __builtin_amdgcn_s_memtime(); } while (0)
    HD_STAMP();
    // ---- open my conv1 stream, then bring the board in ----
    const frag* sp = reinterpret_cast<const frag*>(a.s1) + size_t(wv) * a.s1_wave_frags * 64 + lane;
    frag win[HD_WIN];
#pragma unroll
    for (int q = 0; q < HD_WIN; ++q) win[q] = sp[q * 64];
    {
        if (!x_in_lds) {
            const half_t* xb = reinterpret_cast<const half_t*>(a.x) + size_t(b) * 64 * HD_C;
            for (int i = tid; i < 64 * 32; i += 512) {
                const int r = i >> 5, v = i & 31;
                *reinterpret_cast<uint4*>(X + r * ROW + v * 8) = *reinterpret_cast<const uint4*>(xb + size_t(r) * HD_C + v * 8);
            }
        }
        if (tid < ROW / 2) {                          // zero rows of both tiles
            reinterpret_cast<uint32_t*>(X + 64 * ROW)[tid] = 0u;
            reinterpret_cast<uint32_t*>(P1 + 64 * ROW)[tid] = 0u;
        }
    }
    __syncthreads();
    HD_STAMP();

    // ================= phase 1: policy conv 1 (+ value conv on wave 0) =================
    {
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
        f32x4 bias[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[i] = reinterpret_cast<const f32x4*>(a.b1 + (wv * 2 + lh) * 16)[i];
        const int ntap = wv == 0 ? 10 : 9;           // tap 9 = the value head's 1x1 conv (centre tap geometry, own accumulators)
        f32x16 accv[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) accv[ct][v] = 0.f;
        // L2 warm-up (see tower.hip): the workgroups of an XCD share the job of touching every stream line early; this
        // workgroup takes the lines whose index is (b / 8) % 32 modulo 32 of the 16 fragments it will need two taps from now
        const char* s1b = reinterpret_cast<const char*>(a.s1) + size_t(wv) * a.s1_wave_frags * 1024;
        const int pf_slot = (b >> 3) & 31;
        int pf_old = 0, pf_sink = 0;
        for (int tap = 0; tap < ntap; ++tap) {
            pf_sink ^= pf_old;
            pf_old = (lane < 4 && tap + 2 < ntap) ? *reinterpret_cast<const int*>(s1b + size_t(tap + 2) * 16384 + (lane * 32 + pf_slot) * 128) : 0;
            const int dy = tap < 9 ? tap / 3 - 1 : 0, dx = tap < 9 ? tap % 3 - 1 : 0;
            const half_t* r0 = X + nbr_row(l31, dy, dx) * ROW + lh * 8;
            const half_t* r1 = X + nbr_row(32 + l31, dy, dx) * ROW + lh * 8;
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
                // next step's reads are ISSUED before this step's MFMAs (tower.hip, matrix_interval: without the fence the scheduler sinks
                // them to the end of the step, right in front of their consumers, and every step waits out an LDS latency)
                __builtin_amdgcn_sched_barrier(0);
                if (tap < 9) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hd_mma32(win[s * 2 + (i >> 1)], cur[i], acc[i & 1]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hd_mma32(win[s * 2 + (i >> 1)], cur[i], accv[i & 1]);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) win[s * 2 + e] = sp[(s * 2 + e + HD_WIN) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            sp += 16 * 64;
        }
        pf_sink ^= pf_old;
        asm volatile("" ::"v"(pf_sink));
        HD_STAMP();
        mfma_retire(acc[0], acc[1]);                 // the last MFMAs retire before the asm pack reads them (device_utils.h)
        // BN bias + ReLU -> P1; my 16 rows of a square are 16 consecutive K positions of conv 2 (kernels.h: tower_row_of_position)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            uint32_t o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = pack_relu_h2(acc[ct][2 * i], bias[i >> 1][(2 * i) & 3], acc[ct][2 * i + 1], bias[i >> 1][(2 * i + 1) & 3]);
            uint4* dst = reinterpret_cast<uint4*>(P1 + (ct * 32 + l31) * ROW + wv * 32 + lh * 16);
            dst[0] = uint4{o[0], o[1], o[2], o[3]};
            dst[1] = uint4{o[4], o[5], o[6], o[7]};
        }
        if (wv == 0) {                               // value conv: rows 0..7 = lanes' elements 0..3 (rows 4*lh + 0..3)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cv = lh * 4 + r;
                    vflat[vflat_at(cv * 64 + ct * 32 + l31)] = fmaxf(accv[ct][r] + a.vconv_bias[cv], 0.f);
                }
        }
    }
    __syncthreads();
    HD_STAMP();

    // ================= phase 2: policy conv 2 =================
    {
        const frag* s2 = reinterpret_cast<const frag*>(a.s2) + size_t(wv) * a.s2_wave_frags * 64 + lane;
        f32x16 acc[3][2];                            // [row tile of 32 policy channels][square tile]
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[rt][ct][v] = 0.f;
        frag wf[3][3];                               // [unit mod 3][row tile]: fragments of units u, u+1, u+2 in flight
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) wf[q][rt] = s2[(q * 3 + rt) * 64];
        auto read_b = [&](int u, frag (&bq)[2]) {     // unit u = (tap, k-step): B rows of both square tiles
            const int tap = u >> 4, ks = u & 15, dy = tap / 3 - 1, dx = tap % 3 - 1;
            bq[0] = *reinterpret_cast<const frag*>(P1 + nbr_row(l31, dy, dx) * ROW + ks * 16 + lh * 8);
            bq[1] = *reinterpret_cast<const frag*>(P1 + nbr_row(32 + l31, dy, dx) * ROW + ks * 16 + lh * 8);
        };
        frag bq[2][2];
        read_b(wv * 18, bq[0]);
        for (int i0 = 0; i0 < 18; i0 += 6) {
#pragma unroll
            for (int q6 = 0; q6 < 6; ++q6) {
                const int i = i0 + q6, q = q6 % 3;
                if (i + 1 < 18) read_b(wv * 18 + i + 1, bq[(q6 + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // reads of the next unit before this unit's MFMAs
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    hd_mma32(wf[q][rt], bq[q6 & 1][0], acc[rt][0]);
                    hd_mma32(wf[q][rt], bq[q6 & 1][1], acc[rt][1]);
                }
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) wf[q][rt] = s2[((i + 3) * 3 + rt) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        HD_STAMP();
        // cross-wave reduction of the 8 partial logit sets, one row tile at a time through the (now free) tile area of LDS:
        // slab[wave][(ct*4 + v/4)*64 + lane] = 4 consecutive rows of one square; thread t then owns float4 slot t of every slab.
        // (LDS float atomics would do this in one pass but run at ~190 cycles per wave instruction on this chip.)
        f32x4* slab = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            __syncthreads();                         // every wave is done reading P1 / the previous round's slabs
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4)
                    slab[wv * 512 + (ct * 4 + v4) * 64 + lane] =
                        f32x4{acc[rt][ct][v4 * 4 + 0], acc[rt][ct][v4 * 4 + 1], acc[rt][ct][v4 * 4 + 2], acc[rt][ct][v4 * 4 + 3]};
            __syncthreads();
            f32x4 sum = slab[tid];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const f32x4 p = slab[w * 512 + tid];
                sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; sum[3] += p[3];
            }
            const int ct = tid >> 8, v4 = (tid >> 6) & 3, ln = tid & 63;
            const int co = rt * 32 + 8 * v4 + 4 * (ln >> 5), sq = ct * 32 + (ln & 31);
#pragma unroll
            for (int r = 0; r < 4; ++r) logit[(co + r) * 64 + sq] = sum[r];
        }
    }
    __syncthreads();
    HD_STAMP();

    // ================= phase 3: softmax over the cp * 64 logits =================
    // The tanh head's FC1 weights (256 KiB per workgroup) are requested first and fly while the softmax runs: thread t owns outputs
    // 2*(t/4), +1 over k in [128*(t%4), +128), host-packed in thread order (rise_net.hip): load i of thread t = uint4 i*512 + t = the
    // half2 weights of k = 128*(t%4) + 4i .. 4i+3.
    uint4 vw[32];
    if (!a.wdlp) {
        const uint4* pk = reinterpret_cast<const uint4*>(a.fc1_w) + tid;
#pragma unroll
        for (int i = 0; i < 32; ++i) vw[i] = pk[i * 512];
    }
    {
        const int n4 = a.cp * 16;                    // float4 units
        const f32x4* lg4 = reinterpret_cast<const f32x4*>(logit);
        f32x4* lo = a.logits ? reinterpret_cast<f32x4*>(a.logits + size_t(b) * n4 * 4) : nullptr;
        f32x4* po = a.probs ? reinterpret_cast<f32x4*>(a.probs + size_t(b) * n4 * 4) : nullptr;
        float m = -INFINITY;
        for (int i = tid; i < n4; i += 512) {
            const f32x4 x = lg4[i];
            m = fmaxf(fmaxf(m, fmaxf(x[0], x[1])), fmaxf(x[2], x[3]));
        }
        m = block_reduce_512(m, red, true);
        float sum = 0.f;
        for (int i = tid; i < n4; i += 512) {
            const f32x4 x = lg4[i];
            sum += (__expf(x[0] - m) + __expf(x[1] - m)) + (__expf(x[2] - m) + __expf(x[3] - m));
        }
        sum = block_reduce_512(sum, red, false);
        const float c = m + logf(sum);               // exp(x - (max + log(sum))) as apply_softmax(), neuralnetapi.cpp:241-260
        // policy_out itself (pre-softmax) leaves the CU only when somebody asked for it (mi_net_keep_logits: parity tests), and a search
        // lane that takes the gathered priors below needs neither vector: 2 x 20.7 KB per board that nobody reads
        if (a.probs != nullptr) {
            for (int i = tid; i < n4; i += 512) {
                const f32x4 x = lg4[i];
                if (a.logits != nullptr) lo[i] = x;
                po[i] = f32x4{__expf(x[0] - c), __expf(x[1] - c), __expf(x[2] - c), __expf(x[3] - c)};
            }
        } else if (a.logits != nullptr) {
            for (int i = tid; i < n4; i += 512) lo[i] = lg4[i];
        }
        if (a.g_out != nullptr && b < a.g_n_valid) {     // search lane: the priors of the new node's legal moves, straight from the tile
            const uint32_t cnt = a.g_cnt[b];
            const size_t base = size_t(b) * a.g_stride;
            for (uint32_t j = tid; j < cnt; j += 512) a.g_out[base + j] = __expf(logit[a.g_idx[base + j]] - c);
        }
    }

    HD_STAMP();
    // ================= phase 4: value head =================
    if (!a.wdlp) {
        float part = 0.f;
        {   // FC1 512 -> 256 + ReLU, FC2 256 -> 1: the four lanes of an output pair reduce by DPP, the last one carries on
            const int j2 = tid >> 2, kq = tid & 3;
            const float* vf = vflat + kq * 132;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(vf + 4 * i);
                const half2_t w0 = __builtin_bit_cast(half2_t, vw[i].x), w1 = __builtin_bit_cast(half2_t, vw[i].y);
                const half2_t w2 = __builtin_bit_cast(half2_t, vw[i].z), w3 = __builtin_bit_cast(half2_t, vw[i].w);
                s0 = fmaf(float(w0[0]), f[0], s0); s1 = fmaf(float(w0[1]), f[0], s1);
                s0 = fmaf(float(w1[0]), f[1], s0); s1 = fmaf(float(w1[1]), f[1], s1);
                s0 = fmaf(float(w2[0]), f[2], s0); s1 = fmaf(float(w2[1]), f[2], s1);
                s0 = fmaf(float(w3[0]), f[3], s0); s1 = fmaf(float(w3[1]), f[3], s1);
            }
            s0 += dpp_mov<0x111>(s0); s1 += dpp_mov<0x111>(s1);
            s0 += dpp_mov<0x112>(s0); s1 += dpp_mov<0x112>(s1);
            if (kq == 3)
                part = fmaxf(s0 + a.fc1_b[2 * j2], 0.f) * a.fc2_w[2 * j2] + fmaxf(s1 + a.fc1_b[2 * j2 + 1], 0.f) * a.fc2_w[2 * j2 + 1];
        }
        const float tot = block_reduce_512(part, red, false);
        if (tid == 0) a.value[b] = tanhf(tot + a.fc2_b);
    } else {        // WDLP: wdl = W flat + b, plys = sigmoid(w flat + b); value = -softmax(wdl)[0] + softmax(wdl)[2]
        const float* w = reinterpret_cast<const float*>(a.fc1_w);            // [4][512] float
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = w[k * 512 + tid] * vflat[vflat_at(tid)];
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = block_reduce_512(p[k], red, false);
        if (tid == 0) {
            const float l0 = r[0] + a.wdl_b[0], l1 = r[1] + a.wdl_b[1], l2 = r[2] + a.wdl_b[2];
            const float m = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
            const float inv = 1.f / (e0 + e1 + e2);
            a.value[b] = -e0 * inv + e2 * inv;
            if (a.aux) {
                a.aux[b * 4 + 0] = l0;
                a.aux[b * 4 + 1] = l1;
                a.aux[b * 4 + 2] = l2;
                a.aux[b * 4 + 3] = 1.f / (1.f + expf(-(r[3] + a.wdl_b[3])));
            }
        }
    }
    HD_STAMP();
}

#ifndef CRA_FORWARD_TU
__global__ __launch_bounds__(512) void head_kernel(const HeadArgs a) { head_body(a, false); }

void init_head_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HD_LDS_BYTES);
}

void launch_head(const HeadArgs& a, hipStream_t s) { hipLaunchKernelGGL(head_kernel, dim3(a.batch), dim3(512), HD_LDS_BYTES, s, a); }

#endif  // CRA_FORWARD_TU

}  // namespace cra
