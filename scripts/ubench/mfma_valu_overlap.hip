// Microbenchmark: does a VALU-only wave overlap with an MFMA-only wave on the SAME SIMD (waves w and w+4 of a 512-thread
// workgroup share a SIMD)?  Reports cycles per instruction of each role alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// mode bit0: matrix waves (0-3) run, bit1: vector waves (4-7) run, bit2: vector waves get s_setprio 1, bit3: matrix waves get s_setprio 1
__global__ __launch_bounds__(512) void k(const half8* src, float* out, unsigned long long* cyc, int iters, int mode) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = 0, t1 = 0;
    float res = 0;
    if (wave < 4) {
        half8 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = src[i * 64 + lane]; b[i] = src[256 + i * 64 + lane]; }
        f4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
        if (mode & 8) __builtin_amdgcn_s_setprio(1);
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) res += acc[i][0];
    } else {
        float a0 = lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, x = 0.5f, w = 0.25f;
        if (mode & 4) __builtin_amdgcn_s_setprio(1);
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
                asm volatile(REP4("v_fmac_f32_dpp %0, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                  "v_fmac_f32_dpp %2, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                  "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
            }
        t1 = __builtin_amdgcn_s_memtime();
        res = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

int main() {
    half8* s; float* o; unsigned long long* c;
    hipMalloc(&s, 512 * 16); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    hipMemset(s, 0x3c, 512 * 16);
    const int iters = 4000;
    const char* names[] = {"", "MFMA waves only", "VALU waves only", "both", "", "", "", "both, VALU waves s_setprio 1", "", "", "", "both, MFMA waves s_setprio 1"};
    for (int mode : {1, 2, 3, 7, 11}) {
        k<<<256, 512>>>(s, o, c, 10, mode);
        k<<<256, 512>>>(s, o, c, iters, mode);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-32s cycles per MFMA (wave 0): %6.2f   cycles per VALU instr (wave 4): %6.2f\n", names[mode], (mode & 1) ? double(h[0]) / (iters * 16.0) : 0.0,
               (mode & 2) ? double(h[4]) / (iters * 32.0) : 0.0);
    }
    return 0;
}
