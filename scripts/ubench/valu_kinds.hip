// Microbenchmark: issue cost of the instruction kinds of the float16x3 depthwise, one wave per SIMD, independent instructions:
// v_fmac_f32, v_pk_fma_f32, v_mov_b32_dpp, v_max_f32, v_cndmask_b32, v_cvt_pk_f16_f32, v_fma_mix_f32, and a dpp mov right behind the
// VALU that writes its source (the 2-wait-state hazard).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND> __global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, r4 = lane + 4, r5 = lane + 5, r6 = lane + 6, r7 = lane + 7, x = 0.5f, w = 0.25f;
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, px = {x, x}, pw = {w, w};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) asm volatile(REP8("v_fmac_f32_e32 %0, %4, %5\n v_fmac_f32_e32 %1, %4, %5\n v_fmac_f32_e32 %2, %4, %5\n v_fmac_f32_e32 %3, %4, %5\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x), "v"(w));
        if constexpr (KIND == 1) asm volatile(REP8("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px), "v"(pw));
        if constexpr (KIND == 2) asm volatile(REP8("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x), "v"(w));
        if constexpr (KIND == 3) asm volatile(REP8("v_max_f32_e32 %0, %4, %0\n v_max_f32_e32 %1, %4, %1\n v_max_f32_e32 %2, %5, %2\n v_max_f32_e32 %3, %5, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x), "v"(w));
        if constexpr (KIND == 4) asm volatile(REP8("v_cvt_pk_f16_f32 %0, %4, %5\n v_fma_mix_f32 %1, %0, -1.0, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_cvt_pk_f16_f32 %3, %1, %2\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x), "v"(w));
        if constexpr (KIND == 5) asm volatile(REP8("v_fmac_f32_e32 %0, %4, %5\n s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_e32 %2, %4, %5\n s_nop 1\n v_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x), "v"(w));
        if constexpr (KIND == 6) asm volatile(REP8("v_fmac_f32_dpp %0, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %4, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %2, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %4, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x), "v"(w));
        if constexpr (KIND == 7) asm volatile(REP8("v_pk_mul_f32 %0, %4, %5\n v_pk_add_f32 %1, %4, %5\n v_pk_mul_f32 %2, %4, %5\n v_pk_add_f32 %3, %4, %5\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px), "v"(pw));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + p0.x + p1.y + p2.x + p3.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name, int per_iter, float* o, unsigned long long* c) {
    const int iters = 4000;
    k<KIND><<<256, 256>>>(o, c, 10);
    k<KIND><<<256, 256>>>(o, c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-56s %6.2f cycles per instruction\n", name, double(h) / (double(iters) * per_iter));
}
int main() {
    float* o; unsigned long long* c;
    (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&c, 64);
    run<0>("v_fmac_f32", 32, o, c);
    run<1>("v_pk_fma_f32", 32, o, c);
    run<7>("v_pk_mul_f32 / v_pk_add_f32", 32, o, c);
    run<2>("v_mov_b32_dpp", 32, o, c);
    run<6>("v_fmac_f32_dpp", 32, o, c);
    run<3>("v_max_f32", 32, o, c);
    run<4>("split pair (cvt_pk, 2 fma_mix, cvt_pk; dependent)", 32, o, c);
    run<5>("fmac; s_nop 1; dpp mov of its result (per triple)", 16, o, c);
    return 0;
}
