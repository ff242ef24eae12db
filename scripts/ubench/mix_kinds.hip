// Microbenchmark: which VALU instruction kinds run in the shadow of 16x16x32 MFMAs?
//   same wave  : every MFMA followed by two instructions of the kind (one wave per SIMD)
//   other wave : MFMA-only waves 0-3 (the older ones) beside VALU-only waves 4-7, and the other way round
// kinds: v_fmac_f32, v_pk_fma_f32, v_mov_b32_dpp, v_max_f32, v_cvt_pk_f16_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND> __device__ __forceinline__ void valu2(float (&r)[4], f2 (&p)[2], float x, float w, f2 px, f2 pw) {
    if constexpr (KIND == 0) { asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[0]) : "v"(x), "v"(w)); asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[1]) : "v"(x), "v"(w)); }
    if constexpr (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[0]) : "v"(px), "v"(pw)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[1]) : "v"(px), "v"(pw)); }
    if constexpr (KIND == 2) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[0]) : "v"(x)); asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[1]) : "v"(w)); }
    if constexpr (KIND == 3) { asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(r[0]) : "v"(x)); asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(r[1]) : "v"(w)); }
    if constexpr (KIND == 4) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(r[0]) : "v"(x), "v"(w)); asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(r[1]) : "v"(w), "v"(x)); }
}

// mode 0: same wave (waves 0-3 run MFMA + 2 VALU each);  1: waves 0-3 MFMA only, waves 4-7 VALU only;  2: waves 0-3 VALU only, waves 4-7 MFMA only
template <int KIND> __global__ __launch_bounds__(512) void k(const half8* src, float* out, unsigned long long* cyc, int iters, int mode) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[i * 64 + lane]; b[i] = src[256 + i * 64 + lane]; }
    float r[4] = {float(lane), float(lane + 1), float(lane + 2), float(lane + 3)}, x = 0.5f, w = 0.25f;
    f2 p[2] = {{r[0], r[1]}, {r[2], r[3]}}, px = {x, x}, pw = {w, w};
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    const bool do_mfma = mode == 0 ? wave < 4 : (mode == 1 ? wave < 4 : wave >= 4);
    const bool do_valu = mode == 0 ? wave < 4 : (mode == 1 ? wave >= 4 : wave < 4);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        if (do_mfma)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
                    valu2<KIND>(r, p, x, w, px, pw);
                }
            }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) valu2<KIND>(r, p, x, w, px, pw);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = r[0] + r[1] + r[2] + r[3] + p[0].x + p[1].y;
    for (int i = 0; i < 16; ++i) res += acc[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}
template <int KIND> void run(const char* name, const half8* s, float* o, unsigned long long* c) {
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        k<KIND><<<256, 512>>>(s, o, c, 10, mode);
        k<KIND><<<256, 512>>>(s, o, c, iters, mode);
        (void)hipDeviceSynchronize();
        unsigned long long h[8];
        (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        if (mode == 0) printf("%-18s same wave, MFMA + 2 of them: %6.2f cycles per MFMA\n", name, double(h[0]) / (iters * 16.0));
        if (mode == 1) printf("%-18s MFMA waves older : %6.2f cycles per MFMA, %6.2f per VALU instruction\n", name, double(h[0]) / (iters * 16.0), double(h[4]) / (iters * 32.0));
        if (mode == 2) printf("%-18s VALU waves older : %6.2f cycles per MFMA, %6.2f per VALU instruction\n", name, double(h[4]) / (iters * 16.0), double(h[0]) / (iters * 32.0));
    }
}
int main() {
    half8* s; float* o; unsigned long long* c;
    (void)hipMalloc(&s, 512 * 16); (void)hipMalloc(&o, 256 * 512 * 4); (void)hipMalloc(&c, 64);
    (void)hipMemset(s, 0x3c, 512 * 16);
    run<0>("v_fmac_f32", s, o, c);
    run<1>("v_pk_fma_f32", s, o, c);
    run<2>("v_mov_b32_dpp", s, o, c);
    run<3>("v_max_f32", s, o, c);
    run<4>("v_cvt_pk_f16_f32", s, o, c);
    return 0;
}
