// REPRODUCER (gfx950 / MI355X, ROCm 7.2): v_pk_fma_f32 gives wrong results when a wave of ANOTHER workgroup on the same SIMD issues
// v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 neighbour_mfma.hip -o neighbour_mfma.bin
//   ./neighbour_mfma.bin 0 3000 8 600 64 8     ->  ~9 million wrong sums in 3000 launches (always lanes 48-63, always the LOW half of the packed result)
//   ./neighbour_mfma.bin 0 3000 8 600 64 6     ->  0   (the same chain with only op_sel_hi:[1,0,1]; form 7: only op_sel:[0,1,0]; form 5: plain; form 1: v_fmac_f32)
//   ./neighbour_mfma.bin 5 3000 8 600 64 8     ->  0   (neighbour on v_mfma_f32_32x32x16_f16; 6: 16x16x128 f8f6f4; 7: 16x16x4 f32; 4: v_fma_f32; -1: none)
//   ./neighbour_mfma.bin 100 3000 8 600 64 8   ->  0   (the same two roles as waves of ONE workgroup)
// Victim condition: a dependent chain of v_pk_fma_f32 that reads ONE VGPR pair as src1 first with op_sel_hi:[1,0,1] (low dword for both halves)
// and then with op_sel:[0,1,0] (high dword for both halves) -- what hipcc makes of `h += w[i] * f[i]; h += w[i + 1] * f[i + 1]` on float4
// accumulators with (f[i], f[i + 1]) in one register pair (form 0).  How full the SIMD's register file is does not matter (kind 8).
//
// History: which kind of neighbour on the compute unit makes an FC1-shaped wave lose a word?  (profiles/NOTES.md rounds 4 and 5:
// value_head_kernel's FC1 sums came out wrong in lanes 48-63 of one register whenever workgroups of conv_gemm_x3_kernel<3, 1, 8, 4> shared
// its compute unit; round 4's passive victim beside VALU / LDS-permute / store neighbours stayed clean, but never had a neighbour that
// issues MFMAs, streams 16-byte loads or reads 16-byte LDS rows -- the three things that conv does.)
//
//   victim    : value_head_kernel's FC1 loop and geometry -- 256 threads = one wave per SIMD, ~200 VGPRs, 44 KB of LDS; per round every lane
//               requests 32 x 16 bytes (a 1 KiB row per wave and load, all 32 in flight), then
//                 (L) checks every loaded WORD against the pattern the buffer holds (word = f(address): no arithmetic in between),
//                 (A) runs the FMA chain on four accumulators (the words are small integers as floats, the other factor is 1.0 from a
//                     broadcast 16-byte LDS read: the sums are exact and known),
//                 (S) stores the accumulators with one ds_write_b128, reads them back behind a barrier and compares them with the registers.
//               Every mismatch is reported as [kind L/A/S, block, wave, lane, load j, component, expected, found, HW_ID].
//   aggressor : 512 threads = two waves per SIMD, 152 VGPRs (the victim compiles to 202 -> 208 allocated; 2 x 152 + 208 = 512: the pair fits a
//               SIMD as the policy conv's 2 x 168 and the value head's 160 do),
//               35 KB of LDS, looping ONE kind of work:
//                 0 v_mfma_f32_16x16x32_f16 back to back     1 ds_read_b128 rows     2 global_load_dwordx4 stream     3 all three interleaved
//                 4 v_fma_f32 only (control)                 -1 no aggressor
//                 5 v_mfma_f32_32x32x16_f16     6 v_mfma_scale_f32_16x16x128_f8f6f4     7 v_mfma_f32_16x16x4_f32
//                 8 = kind 0 with 120 registers (the SIMD's register file is then not full beside the victim)
//                 100 = no second kernel: the MIXED kernel, victim waves 0-3 and MFMA waves 4-7 in one workgroup
//
// usage: neighbour_mfma.bin <kind> [launches] [victim rounds] [aggressor iterations] [blocks] [victim form 0-11]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int ROWS = 512;                       // rows of 256 words x 4 = 1 KiB x 4 waves ... the buffer: [4 waves][ROWS][64 lanes][4 words] (a lane's 32 loads 1 KiB apart, as FC1's)
__host__ __device__ inline float word_at(uint32_t idx) { return float(int((idx * 2654435761u) >> 27) - 16); }     // integers -16 ... 15

__device__ __forceinline__ void report_one(uint32_t* report, uint32_t* count, uint32_t kind, uint32_t j, uint32_t e, float want, float got) {
    const uint32_t k = atomicAdd(count, 1u);
    if (k < 512) {
        uint32_t* o = report + k * 9;
        o[0] = kind; o[1] = blockIdx.x; o[2] = threadIdx.x >> 6; o[3] = threadIdx.x & 63; o[4] = j; o[5] = e;
        o[6] = __builtin_bit_cast(uint32_t, want); o[7] = __builtin_bit_cast(uint32_t, got);
        o[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

// VK: 0 = the FMA chain as the compiler writes it (two v_pk_fma_f32 per weight row), 1 = four v_fmac_f32 (inline asm), 2 = the packed chain on
// words made in registers (no loads in the round at all)
//     3 = v_pk_mul_f32 (word * 1.0 from LDS, the products' bit patterns summed), 4 = v_pk_add_f32 (word + 0.0 from LDS, likewise)
// BARRIERS = false: no (S) stage and no workgroup barriers (the victim role inside the mixed kernel below)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int VK, bool BARRIERS> __device__ __forceinline__ void victim_rounds(const float* __restrict__ buf, uint32_t* report, uint32_t* count, float* lds,
                                                                                int rounds, int salt) {
    float* s_flat = lds;                         // [128] ones (the "flattened conv output"), broadcast reads; [128..255] zeros
    float* s_part = lds + 1024;                  // [4][256]
    const int tid = threadIdx.x, lane = tid & 63, kq = (tid >> 6) & 3;
    if constexpr (VK == 9 || VK == 10) asm volatile("v_mov_b32 v199, 0" ::: "v199");          // the allocation of the failing compiler form: 200 registers
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int row0 = ((r + salt) * 32) % ROWS;
        // what the round must see, from the pattern alone (a rolled loop: no registers held across the loads' shadow)
        uint32_t want_cs[4] = {0u, 0u, 0u, 0u};
        float want_h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int j = 0; j < 32; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = word_at(uint32_t((kq * ROWS + row0 + j) * 64 + lane) * 4 + e);
                want_cs[e] += __builtin_bit_cast(uint32_t, v);
                want_h[e] += v;
            }
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        f32x4 w[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if constexpr (VK == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[j][e] = word_at(uint32_t((kq * ROWS + row0 + j) * 64 + lane) * 4 + e);
            } else {
                w[j] = *reinterpret_cast<const f32x4*>(buf + ((size_t(kq) * ROWS + row0 + j) * 64 + lane) * 4);
            }
        }
        // (L) the words as they arrived: sum of their bit patterns per component; a mismatch reports both sums (one wrong word: their difference = found - expected bits, searched on the host)
        uint32_t cs[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 32; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float we = w[j][e];                       // (bit_cast straight on a vector element reads element 0)
                cs[e] += __builtin_bit_cast(uint32_t, we);
            }
        if (cs[0] != want_cs[0] || cs[1] != want_cs[1] || cs[2] != want_cs[2] || cs[3] != want_cs[3]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (cs[e] != want_cs[e]) report_one(report, count, 'L', uint32_t(row0), e, __builtin_bit_cast(float, want_cs[e]), __builtin_bit_cast(float, cs[e]));
        }
        // (A) the FMA chain of FC1
        uint32_t pcs[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(s_flat + 4 * q + (r & 3) * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (VK == 3 || VK == 4) {
                    // one packed multiply (by 1.0) or add (of 0.0) per pair of words: the result must be the word itself
                    const f32x2 lo = {w[4 * q + e][0], w[4 * q + e][1]}, hi = {w[4 * q + e][2], w[4 * q + e][3]};
                    const float g = VK == 3 ? f[e] : s_flat[128 + 4 * q + e];
                    const f32x2 gg = {g, g};
                    f32x2 r0, r1;
                    if constexpr (VK == 3) {
                        asm volatile("v_pk_mul_f32 %0, %2, %4\n\tv_pk_mul_f32 %1, %3, %4" : "=&v"(r0), "=&v"(r1) : "v"(lo), "v"(hi), "v"(gg));
                    } else {
                        asm volatile("v_pk_add_f32 %0, %2, %4\n\tv_pk_add_f32 %1, %3, %4" : "=&v"(r0), "=&v"(r1) : "v"(lo), "v"(hi), "v"(gg));
                    }
                    pcs[0] += __builtin_bit_cast(uint32_t, float(r0[0]));
                    pcs[1] += __builtin_bit_cast(uint32_t, float(r0[1]));
                    pcs[2] += __builtin_bit_cast(uint32_t, float(r1[0]));
                    pcs[3] += __builtin_bit_cast(uint32_t, float(r1[1]));
                } else if constexpr (VK == 5) {            // v_pk_fma_f32 written out (no op_sel: the factor sits in both halves of a pair)
                    f32x2 hl = {h[0], h[1]}, hh = {h[2], h[3]};
                    const f32x2 lo = {w[4 * q + e][0], w[4 * q + e][1]}, hi = {w[4 * q + e][2], w[4 * q + e][3]}, gg = {f[e], f[e]};
                    asm volatile("v_pk_fma_f32 %0, %2, %4, %0\n\tv_pk_fma_f32 %1, %3, %4, %1" : "+v"(hl), "+v"(hh) : "v"(lo), "v"(hi), "v"(gg));
                    h = f32x4{hl[0], hl[1], hh[0], hh[1]};
                } else if constexpr (VK == 11) {             // the same alternation on PACKED F16 (v_pk_fma_f16: the float16 kernels' depthwise instruction)
                    if ((e & 1) == 0) {
                        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                        half2v hl = {_Float16(h[0]), _Float16(h[1])}, hh = {_Float16(h[2]), _Float16(h[3])};
                        const half2v lo0 = {_Float16(w[4 * q + e][0]), _Float16(w[4 * q + e][1])}, hi0 = {_Float16(w[4 * q + e][2]), _Float16(w[4 * q + e][3])};
                        const half2v lo1 = {_Float16(w[4 * q + e + 1][0]), _Float16(w[4 * q + e + 1][1])}, hi1 = {_Float16(w[4 * q + e + 1][2]), _Float16(w[4 * q + e + 1][3])};
                        const half2v gg = {_Float16(f[e]), _Float16(f[e + 1])};
                        asm volatile("v_pk_fma_f16 %0, %2, %6, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f16 %1, %3, %6, %1 op_sel_hi:[1,0,1]\n\t"
                                     "v_pk_fma_f16 %0, %4, %6, %0 op_sel:[0,1,0]\n\tv_pk_fma_f16 %1, %5, %6, %1 op_sel:[0,1,0]"
                                     : "+v"(hl), "+v"(hh) : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "v"(gg));
                        h = f32x4{float(hl[0]), float(hl[1]), float(hh[0]), float(hh[1])};
                    }
                } else if constexpr (VK == 8 || VK == 10) {  // BOTH forms alternating on one pair of factors, as the compiler writes FC1's chain
                    if constexpr ((0) == 0) {
                        if ((e & 1) == 0) {
                            f32x2 hl = {h[0], h[1]}, hh = {h[2], h[3]};
                            const f32x2 lo0 = {w[4 * q + e][0], w[4 * q + e][1]}, hi0 = {w[4 * q + e][2], w[4 * q + e][3]};
                            const f32x2 lo1 = {w[4 * q + e + 1][0], w[4 * q + e + 1][1]}, hi1 = {w[4 * q + e + 1][2], w[4 * q + e + 1][3]};
                            const f32x2 gg = {f[e], f[e + 1]};
                            asm volatile("v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %6, %1 op_sel_hi:[1,0,1]\n\t"
                                         "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %1, %5, %6, %1 op_sel:[0,1,0]"
                                         : "+v"(hl), "+v"(hh) : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "v"(gg));
                            h = f32x4{hl[0], hl[1], hh[0], hh[1]};
                        }
                    }
                } else if constexpr (VK == 6 || VK == 7 || VK == 9) {   // the compiler's forms: the factor is ONE dword of a pair, picked by op_sel
                    f32x2 hl = {h[0], h[1]}, hh = {h[2], h[3]};
                    const f32x2 lo = {w[4 * q + e][0], w[4 * q + e][1]}, hi = {w[4 * q + e][2], w[4 * q + e][3]};
                    const f32x2 gg = VK != 7 ? f32x2{f[e], 777.f} : f32x2{777.f, f[e]};
                    if constexpr (VK != 7)       // both halves take src1's LOW dword
                        asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]" : "+v"(hl), "+v"(hh) : "v"(lo), "v"(hi), "v"(gg));
                    else                         // both halves take src1's HIGH dword
                        asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel:[0,1,0]" : "+v"(hl), "+v"(hh) : "v"(lo), "v"(hi), "v"(gg));
                    h = f32x4{hl[0], hl[1], hh[0], hh[1]};
                } else if constexpr (VK == 1) {
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
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (VK == 3 || VK == 4) {
                if (pcs[e] != want_cs[e]) report_one(report, count, 'P', 0, e, __builtin_bit_cast(float, want_cs[e]), __builtin_bit_cast(float, pcs[e]));
            } else {
                if (h[e] != want_h[e]) report_one(report, count, 'A', 0, e, want_h[e], h[e]);
            }
        }
        if constexpr (BARRIERS) {
            // (S) through LDS
            *reinterpret_cast<f32x4*>(s_part + kq * 256 + 4 * lane) = h;
            __syncthreads();
            const f32x4 back = *reinterpret_cast<const f32x4*>(s_part + kq * 256 + 4 * lane);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (__builtin_bit_cast(uint32_t, back[e]) != __builtin_bit_cast(uint32_t, h[e])) report_one(report, count, 'S', 0, e, h[e], back[e]);
            __syncthreads();
        }
    }
}

template <int VK> __global__ __launch_bounds__(256) void victim(const float* __restrict__ buf, uint32_t* report, uint32_t* count, float* dump, int rounds, int salt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x < 256) lds[threadIdx.x] = threadIdx.x < 128 ? 1.0f : 0.0f;
    __syncthreads();
    victim_rounds<VK, true>(buf, report, count, lds, rounds, salt);
}

// ONE workgroup of eight waves: waves 0-3 run the victim's rounds, waves 4-7 (their SIMD partners) issue MFMAs back to back -- the shape of
// the tower kernels' EXPAND / PROJECT roles.  Does the fault need a neighbour of ANOTHER workgroup, or is a partner wave enough?
template <int VK> __global__ __launch_bounds__(512) void mixed(const float* __restrict__ buf, uint32_t* report, uint32_t* count, float* sink, int rounds, int salt,
                                                              int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x < 256) lds[threadIdx.x] = threadIdx.x < 128 ? 1.0f : 0.0f;
    __syncthreads();
    if (threadIdx.x < 256) {
        victim_rounds<VK, false>(buf, report, count, lds, rounds, salt);
    } else {
        const int lane = threadIdx.x & 63;
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        half8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = _Float16(0.01f * float((lane + i) & 15)); b[i] = _Float16(0.02f * float((lane * 3 + i) & 7)); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
        if (s == 123.456f) sink[threadIdx.x] = s;
    }
}

template <int KIND> __global__ __launch_bounds__(512) void aggressor(const float* __restrict__ buf, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    if constexpr (KIND == 8) asm volatile("v_mov_b32 v119, 0" ::: "v119");          // 120 registers: 2 x 120 + 200 = 440, the register file is NOT full
    else asm volatile("v_mov_b32 v151, 0" ::: "v151");            // 152 registers: 2 x 152 + the victim's 200 = 504, the pair fills the SIMD's register file
    for (int i = tid; i < 8192; i += 512) lds[i] = 0.001f * float(i & 255);
    __syncthreads();
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    half8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = _Float16(0.01f * float((lane + i) & 15)); b[i] = _Float16(0.02f * float((lane * 3 + i) & 7)); }
    f32x4 x = {0.1f, 0.2f, 0.3f, 0.4f};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    f32x16 acc16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    i32x8 a8, b8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + lane; b8[i] = 0x34343434 + i; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0 || KIND == 3 || KIND == 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
        }
        if constexpr (KIND == 1 || KIND == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + u * 256 + it * 68) & 8188));
                x += v;
            }
        }
        if constexpr (KIND == 2 || KIND == 3) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(buf + (size_t((it * 4 + u + blockIdx.x * 7) % ROWS) * 4 * 64 + (tid & 255)) * 4);
                x += v;
            }
        }
        if constexpr (KIND == 4) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, 0.5, 0.5" : "+v"(x[u & 3]));
        }
        if constexpr (KIND == 5) {                          // the 32x32 form (16 accumulator registers)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc16, 0, 0, 0);
        }
        if constexpr (KIND == 6) {                          // the 8-bit form, K = 128
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[u & 3], 1, 1, 0, 0, 0, 0);
        }
        if constexpr (KIND == 7) {                          // exact f32 (not on the XDL path's f16 datapath)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0], x[1], acc[u & 3], 0, 0, 0);
        }
        if constexpr (KIND == 3) { a[it & 7] = _Float16(x[0] * 1e-9f); }
    }
    float s = x[0] + x[1] + x[2] + x[3] + acc16[0] + acc16[5] + acc16[15];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    if (s == 123.456f) sink[tid] = s;
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0, launches = argc > 2 ? atoi(argv[2]) : 2000, rounds = argc > 3 ? atoi(argv[3]) : 8,
              iters = argc > 4 ? atoi(argv[4]) : 600, blocks = argc > 5 ? atoi(argv[5]) : 64, vk = argc > 6 ? atoi(argv[6]) : 0;
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    std::vector<float> host(size_t(ROWS) * 4 * 64 * 4);
    for (size_t i = 0; i < host.size(); ++i) host[i] = word_at(uint32_t(i));
    float *buf, *sink;
    uint32_t *report, *count;
    CHECK(hipMalloc(&buf, host.size() * 4));
    CHECK(hipMemcpy(buf, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&report, 512 * 9 * 4));
    CHECK(hipMalloc(&count, 8));
    CHECK(hipMemset(count, 0, 8));
    float* dump = nullptr;
    const size_t vlds = 44064, alds = 35 * 1024;
    for (int l = 0; l < launches; ++l) {
        if (kind == 100) {                                   // the mixed kernel: victim and MFMA roles in ONE workgroup, no second stream
            if (vk == 1) hipLaunchKernelGGL(mixed<1>, dim3(blocks), dim3(512), vlds, sv, buf, report, count, sink, rounds, l, iters);
            else if (vk == 6) hipLaunchKernelGGL(mixed<6>, dim3(blocks), dim3(512), vlds, sv, buf, report, count, sink, rounds, l, iters);
            else if (vk == 7) hipLaunchKernelGGL(mixed<7>, dim3(blocks), dim3(512), vlds, sv, buf, report, count, sink, rounds, l, iters);
            else if (vk == 8) hipLaunchKernelGGL(mixed<8>, dim3(blocks), dim3(512), vlds, sv, buf, report, count, sink, rounds, l, iters);
            else hipLaunchKernelGGL(mixed<5>, dim3(blocks), dim3(512), vlds, sv, buf, report, count, sink, rounds, l, iters);
        }
        else if (vk == 1) hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 2) hipLaunchKernelGGL(victim<2>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 3) hipLaunchKernelGGL(victim<3>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 4) hipLaunchKernelGGL(victim<4>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 5) hipLaunchKernelGGL(victim<5>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 6) hipLaunchKernelGGL(victim<6>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 7) hipLaunchKernelGGL(victim<7>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 8) hipLaunchKernelGGL(victim<8>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 9) hipLaunchKernelGGL(victim<9>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 10) hipLaunchKernelGGL(victim<10>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else if (vk == 11) hipLaunchKernelGGL(victim<11>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        else hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), vlds, sv, buf, report, count, dump, rounds, l);
        switch (kind) {
            case 0: hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 1: hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 2: hipLaunchKernelGGL(aggressor<2>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 3: hipLaunchKernelGGL(aggressor<3>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 4: hipLaunchKernelGGL(aggressor<4>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 5: hipLaunchKernelGGL(aggressor<5>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 6: hipLaunchKernelGGL(aggressor<6>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 7: hipLaunchKernelGGL(aggressor<7>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            case 8: hipLaunchKernelGGL(aggressor<8>, dim3(blocks), dim3(512), alds, sa, buf, sink, iters); break;
            default: break;
        }
        if ((l & 31) == 31) { CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sa)); }
    }
    CHECK(hipDeviceSynchronize());
    uint32_t n = 0;
    std::vector<uint32_t> rep(512 * 9);
    CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(rep.data(), report, rep.size() * 4, hipMemcpyDeviceToHost));
    printf("victim form %d (0 v_pk_fma_f32, 1 v_fmac_f32, 2 v_pk_fma_f32 on register-made words, 3 v_pk_mul_f32, 4 v_pk_add_f32, 5 v_pk_fma_f32 plain, 6 op_sel_hi:[1,0,1], 7 op_sel:[0,1,0], 8 both forms alternating, 9 = 6 and 10 = 8 padded to 200 registers, 11 = 8 on v_pk_fma_f16), aggressor kind %d%s: %u mismatches in %d victim launches (%d blocks x 4 waves x %d rounds x 128 loaded words)\n", vk, kind, kind == 100 ? " (MIXED: both roles in one workgroup)" : "", n, launches, blocks, rounds);
    for (uint32_t k = 0; k < n && k < 40; ++k) {
        const uint32_t* o = rep.data() + k * 9;
        printf("  %c block %u wave %u lane %u load %u word %u: expected %g found %g  hw_id %08x (simd %u cu %u sh %u se %u)\n", char(o[0]), o[1], o[2], o[3], o[4],
               o[5], double(__builtin_bit_cast(float, o[6])), double(__builtin_bit_cast(float, o[7])), o[8], (o[8] >> 4) & 3, (o[8] >> 8) & 15,
               (o[8] >> 12) & 1, (o[8] >> 13) & 7);
    }
    // (L) reports carry checksums: with ONE wrong word in the lane's 32 loads of that component, found - expected = bits(found word) -
    // bits(right word); say which load it can have been and whether the found word is what the same register held a round earlier
    for (uint32_t k = 0; k < n && k < 40; ++k) {
        const uint32_t* o = rep.data() + k * 9;
        if (o[0] != 'L') continue;
        const int kq = int(o[2]), lane = int(o[3]), row0 = int(o[4]), e = int(o[5]);
        const uint32_t diff = o[7] - o[6];
        for (int j = 0; j < 32; ++j) {
            const uint32_t right = __builtin_bit_cast(uint32_t, word_at(uint32_t((kq * ROWS + row0 + j) * 64 + lane) * 4 + e));
            const float found = __builtin_bit_cast(float, right + diff);
            if (found == float(int(found)) && found >= -16.f && found <= 15.f) {
                const int rb = (row0 - 32 + ROWS) % ROWS;
                const bool stale = word_at(uint32_t((kq * ROWS + rb + j) * 64 + lane) * 4 + e) == found;
                printf("    report %u: load %d word %d may have held %g instead of %g%s%s\n", k, j, e, double(found), double(__builtin_bit_cast(float, right)),
                       stale ? "  (= this load's word one round earlier)" : "", found == 0.f ? "  (zero)" : "");
            }
        }
    }
    return 0;
}
