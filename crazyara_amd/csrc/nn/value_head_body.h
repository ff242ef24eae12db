// The value head of one board as a device function (the body of value_head_kernel, kernels.hip): also one of the two roles of the
// small-batch head launch (x3.hip: heads_small_kernel), where it runs BESIDE the second policy conv instead of behind it.
#pragma once
#include "kernels.h"
#include "device_utils.h"

namespace cra {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* s_red) {      // (NT = 256: ((w0 + w1) + w2) + w3, the order this kernel has always had)
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = s_red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) t += s_red[i];
    return t;
}

// (the comments on PROBE and SCALAR_FMA are in front of value_head_kernel, kernels.hip)
__device__ __forceinline__ float block_sum_256(float v, float* s_red) { return block_sum<256>(v, s_red); }

// NT threads: 256 (the PROBE instantiation, whose record layout is four FC1 groups) or 512 -- NT / 64 groups of conv outputs and of FC1
// inputs; the product form runs 512 (a board's head is a chain of latencies, and eight waves keep twice the loads in flight)
template <typename T, bool PROBE, bool SCALAR_FMA, int NT = 256>
__device__ __forceinline__ void value_head_body(const ValueHeadArgs& a, char* smem, const int b) {
    constexpr int NG = NT / 64;
    static_assert(NT == 256 || (NT == 512 && !PROBE), "256 or 512 threads");
    const int CH = a.C / 2;                                    // the board is staged in two halves of its channels (LDS stays below 64 KiB)
    const int XP = CH + 4;                                    // floats per staged row: rows step 4 banks
    float* xs = reinterpret_cast<float*>(smem);               // [64][C / 2 + 4]
    float* ws = xs + kSquares * XP;                            // [cv][C]
    float* s_flat = ws + a.cv * a.C;                           // [64 * cv]
    float* s_red = s_flat + kSquares * a.cv;                   // [8]
    const int tid = threadIdx.x;
    const T* xb = reinterpret_cast<const T*>(a.x) + size_t(b) * kSquares * a.C;
    const int nf = kSquares * a.cv;

    for (int i0 = tid * 4; i0 < a.cv * a.C; i0 += 4 * 4 * NT) {    // the folded conv weights, 16 bytes per thread and piece, four pieces in flight
        f32x4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 4 * NT < a.cv * a.C) wv[u] = *reinterpret_cast<const f32x4*>(a.wconv + i0 + u * 4 * NT);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 4 * NT < a.cv * a.C) *reinterpret_cast<f32x4*>(ws + i0 + u * 4 * NT) = wv[u];
    }
    float dbg_in = 0.f;                                        // (development) what this thread staged of the board
    {   // conv 1x1 + BN + ReLU: thread = square tid % 64, channels tid / 64, + 4, ... (at most four per thread)
        const int sq = tid & 63, g = tid >> 6;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* xr = xs + sq * XP;
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();                         // everyone is through with the first half of the tile
            // (four pieces per thread at C = 256, all loads in flight before the first LDS write: taken one by one the loop is four
            // HBM round trips long)
            for (int i0 = tid; i0 < kSquares * (CH / 8); i0 += 4 * NT) {
                float f[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * NT;
                    if (i < kSquares * (CH / 8)) load8<T>(xb + size_t(i / (CH / 8)) * a.C + half * CH + (i % (CH / 8)) * 8, f[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * NT;
                    if (i < kSquares * (CH / 8)) {
                        const int r = i / (CH / 8), v = i % (CH / 8);
                        *reinterpret_cast<f32x4*>(xs + r * XP + v * 8) = f32x4{f[u][0], f[u][1], f[u][2], f[u][3]};
                        *reinterpret_cast<f32x4*>(xs + r * XP + v * 8 + 4) = f32x4{f[u][4], f[u][5], f[u][6], f[u][7]};
                        if (a.dbg)
                            for (int e = 0; e < 8; ++e) dbg_in += f[u][e];
                    }
                }
            }
            __syncthreads();
            for (int c = 0; c < CH; c += 4) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = g + NG * k;
                    if (co < a.cv) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + co * a.C + half * CH + c);
                        acc[k] = fmaf(wv[0], xv[0], acc[k]);
                        acc[k] = fmaf(wv[1], xv[1], acc[k]);
                        acc[k] = fmaf(wv[2], xv[2], acc[k]);
                        acc[k] = fmaf(wv[3], xv[3], acc[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = g + NG * k;
            if (co < a.cv) s_flat[co * kSquares + sq] = fmaxf(acc[k] + a.bconv[co], 0.f);
        }
    }
    __syncthreads();

    if (a.wwdl) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < nf; i += NT) {
            const float f = s_flat[i];
            p[0] = fmaf(a.wwdl[i], f, p[0]);
            p[1] = fmaf(a.wwdl[nf + i], f, p[1]);
            p[2] = fmaf(a.wwdl[2 * nf + i], f, p[2]);
            p[3] = fmaf(a.wplys[i], f, p[3]);
        }
        float r[4];
        for (int k = 0; k < 4; ++k) r[k] = block_sum<NT>(p[k], s_red);
        if (tid == 0) {
            const float l0 = r[0] + a.bwdl[0], l1 = r[1] + a.bwdl[1], l2 = r[2] + a.bwdl[2];
            const float m = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
            const float inv = 1.f / (e0 + e1 + e2);
            a.value[b] = -e0 * inv + e2 * inv;
            if (a.aux) {
                a.aux[b * 4 + 0] = l0;
                a.aux[b * 4 + 1] = l1;
                a.aux[b * 4 + 2] = l2;
                a.aux[b * 4 + 3] = 1.f / (1.f + expf(-(r[3] + a.bplys)));
            }
        }
        return;
    }

    if (a.dbg) {                                               // (development) stage checksums, see ValueHeadArgs::dbg
        float* o = a.dbg + size_t(b) * 8;
        const float s_in = block_sum<NT>(dbg_in, s_red);
        float w_sum = 0.f, f_sum = 0.f;
        for (int i = tid; i < a.cv * a.C; i += NT) w_sum += ws[i];
        for (int i = tid; i < nf; i += NT) f_sum += s_flat[i];
        const float s_w = block_sum<NT>(w_sum, s_red), s_f = block_sum<NT>(f_sum, s_red);
        if (tid == 0) { o[0] = s_in; o[1] = s_w; o[2] = s_f; }
        __syncthreads();
    }

    // FC1 + ReLU + FC2: thread = (four consecutive outputs, quarter of the inputs): a row of the transposed matrix is one 16-byte load per
    // lane (a wave reads 1 KiB), 32 of them in flight; the quarters' partial sums meet in LDS (the staged board is dead by now)
    float* s_part = (a.variant & 1) ? s_red + 8 : xs;           // [4 quarters][fc]
    float part = 0.f;
    const int kq = tid >> 6, nq = nf / NG;
    float* probe = PROBE ? a.dbg + size_t(a.batch) * (8 + 1024) + size_t(b) * (16 + 3 * 1024) : nullptr;
    for (int j4 = tid & 63; 4 * j4 < a.fc; j4 += 64) {
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        uint32_t cs[4] = {0u, 0u, 0u, 0u};
        const float* wt = a.w1t + size_t(kq) * nq * a.fc + 4 * j4;
        const float* fl = s_flat + kq * nq;
        int i = 0;
        for (; i + 32 <= nq; i += 32) {
            f32x4 w[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (a.variant & 8) w[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wt + size_t(i + j) * a.fc));
                else w[j] = *reinterpret_cast<const f32x4*>(wt + size_t(i + j) * a.fc);
            }
            if (a.variant & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (development) every load back before the first product
            if constexpr (PROBE) {
#pragma unroll
                for (int j = 0; j < 32; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float we = w[j][e];          // (bit_cast straight on the vector element reads element 0: round 5's first probe)
                        cs[e] += __builtin_bit_cast(uint32_t, we);
                    }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(fl + i + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (SCALAR_FMA) {
                        float h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
                        const float w0 = w[4 * q + e][0], w1 = w[4 * q + e][1], w2 = w[4 * q + e][2], w3 = w[4 * q + e][3], fe = f[e];
                        asm volatile("v_fmac_f32 %0, %4, %8\n\tv_fmac_f32 %1, %5, %8\n\tv_fmac_f32 %2, %6, %8\n\tv_fmac_f32 %3, %7, %8"
                                     : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(fe));
                        h = f32x4{h0, h1, h2, h3};
                    } else {
                        h[0] = fmaf(w[4 * q + e][0], f[e], h[0]);
                        h[1] = fmaf(w[4 * q + e][1], f[e], h[1]);
                        h[2] = fmaf(w[4 * q + e][2], f[e], h[2]);
                        h[3] = fmaf(w[4 * q + e][3], f[e], h[3]);
                    }
                    if (a.variant & 2) asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
                }
            }
        }
        for (; i < nq; ++i) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wt + size_t(i) * a.fc);
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = fmaf(w[e], fl[i], h[e]);
        }
        if constexpr (PROBE) {
            if (kq * a.fc + 4 * j4 + 3 < 1024) {
                *reinterpret_cast<f32x4*>(probe + 16 + 1024 + kq * a.fc + 4 * j4) = h;               // (2) from the registers
                float* oc = probe + 16 + 2048 + kq * a.fc + 4 * j4;
#pragma unroll
                for (int e = 0; e < 4; ++e) oc[e] = __builtin_bit_cast(float, cs[e]);                 // (3) what the loads returned
            }
        }
        *reinterpret_cast<f32x4*>(s_part + kq * a.fc + 4 * j4) = h;
    }
    __syncthreads();
    if constexpr (PROBE) {
        for (int i = tid; i < 4 * a.fc && i < 1024; i += NT) probe[16 + i] = s_part[i];              // (1) first read back from LDS
        if ((tid & 63) == 0) probe[tid >> 6] = __builtin_bit_cast(float, __builtin_amdgcn_s_getreg((31 << 11) | 4));
        if (tid == 0) probe[4] = __builtin_bit_cast(float, __builtin_amdgcn_s_getreg((31 << 11) | 20));
    }
    for (int t = tid; t < a.fc; t += NT) {
        float ps = (s_part[t] + s_part[a.fc + t]) + (s_part[2 * a.fc + t] + s_part[3 * a.fc + t]);
        if constexpr (NG == 8) ps += (s_part[4 * a.fc + t] + s_part[5 * a.fc + t]) + (s_part[6 * a.fc + t] + s_part[7 * a.fc + t]);
        const float h = a.b1[t] + ps;
        part = fmaf(a.w2[t], fmaxf(h, 0.f), part);
    }
    const float tot = block_sum<NT>(part, s_red);
    if (tid == 0) a.value[b] = tanhf(tot + a.b2);
    if (a.dbg) {
        float p_sum = 0.f;
        for (int i = tid; i < NG * a.fc; i += NT) p_sum += s_part[i];
        const float s_p = block_sum<NT>(p_sum, s_red);
        for (int i = tid; i < 4 * a.fc && i < 1024; i += NT) a.dbg[size_t(a.batch) * 8 + size_t(b) * 1024 + i] = s_part[i];   // the partial sums themselves
        if (tid == 0) {
            float* o = a.dbg + size_t(b) * 8;
            o[3] = s_p;
            o[4] = tot;
            o[5] = tanhf(tot + a.b2);
            o[6] = __builtin_bit_cast(float, __builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_ID: wave / simd / cu / sh / se (raw bits)
            o[7] = __builtin_bit_cast(float, __builtin_amdgcn_s_getreg((31 << 11) | 20));     // XCC_ID
        }
    }
}

}  // namespace cra
