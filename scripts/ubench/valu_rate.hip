// Microbenchmark: VALU issue rate on gfx950 for the depthwise inner loop: plain / DPP-fused / packed-f16 FMAs,
// as N independent accumulation chains, one wave per SIMD (256 threads) or two (512).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
    float a0 = src[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float x = src[threadIdx.x + 512], w = src[threadIdx.x + 1024];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 8 independent plain fmac chains, 64 instr
            asm volatile(REP4("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                              "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
                              REP4("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                              "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        } else if (MODE == 1) {   // 8 independent DPP fmac chains
            asm volatile(REP4("v_fmac_f32_dpp %0, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %2, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %4, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %6, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         REP4("v_fmac_f32_dpp %0, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %2, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %4, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %6, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        } else if (MODE == 2) {   // 4 independent plain chains (dependent distance 4)
            asm volatile(REP16("v_fmac_f32_e32 %0, %4, %5\n v_fmac_f32_e32 %1, %4, %5\n v_fmac_f32_e32 %2, %4, %5\n v_fmac_f32_e32 %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(w));
        } else if (MODE == 3) {   // 4 independent DPP chains
            asm volatile(REP16("v_fmac_f32_dpp %0, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %2, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(w));
        } else if (MODE == 4) {   // 8 independent v_pk_fma_f16 chains
            asm volatile(REP4("v_pk_fma_f16 %0, %8, %9, %0\n v_pk_fma_f16 %1, %8, %9, %1\n v_pk_fma_f16 %2, %8, %9, %2\n v_pk_fma_f16 %3, %8, %9, %3\n"
                              "v_pk_fma_f16 %4, %8, %9, %4\n v_pk_fma_f16 %5, %8, %9, %5\n v_pk_fma_f16 %6, %8, %9, %6\n v_pk_fma_f16 %7, %8, %9, %7\n")
                         REP4("v_pk_fma_f16 %0, %8, %9, %0\n v_pk_fma_f16 %1, %8, %9, %1\n v_pk_fma_f16 %2, %8, %9, %2\n v_pk_fma_f16 %3, %8, %9, %3\n"
                              "v_pk_fma_f16 %4, %8, %9, %4\n v_pk_fma_f16 %5, %8, %9, %5\n v_pk_fma_f16 %6, %8, %9, %6\n v_pk_fma_f16 %7, %8, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        } else if (MODE == 5) {   // 8 independent v_mov_b32_dpp (data movement only)
            asm volatile(REP4("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         REP4("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE> void run(const char* name, int threads, const float* s, float* o, unsigned long long* c) {
    const int iters = 2000;
    k<MODE><<<256, threads>>>(s, o, c, 10);
    k<MODE><<<256, threads>>>(s, o, c, iters);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s %d waves/SIMD: %.2f cycles per instruction (wave 0)\n", name, threads / 256, double(h[0]) / (iters * 64.0));
}

int main() {
    float* s; float* o; unsigned long long* c;
    hipMalloc(&s, 2048 * 4); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    hipMemset(s, 0, 2048 * 4);
    for (int thr : {256, 512}) {
        run<0>("v_fmac_f32, 8 chains", thr, s, o, c);
        run<2>("v_fmac_f32, 4 chains", thr, s, o, c);
        run<1>("v_fmac_f32_dpp row_shr:1, 8 chains", thr, s, o, c);
        run<3>("v_fmac_f32_dpp row_shr:1, 4 chains", thr, s, o, c);
        run<4>("v_pk_fma_f16, 8 chains", thr, s, o, c);
        run<5>("v_mov_b32_dpp row_shr:1, 8 indep", thr, s, o, c);
    }
    return 0;
}
