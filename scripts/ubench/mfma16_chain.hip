// mfma16_chain.hip with the accumulators in AccVGPRs ("+a" operands of an inline-asm v_mfma_f32_16x16x32_f16) beside the same loop with
// them in architectural VGPRs ("+v"): does a LONE wave (one per SIMD) issue MFMAs on several independent accumulators faster when C / D do
// not share the VGPR read ports with A and B?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES, bool AGPR>
__global__ __launch_bounds__(64 * WAVES) void k(const half8* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[2] = {src[lane], src[64 + lane]}, b[2] = {src[128 + lane], src[192 + lane]};
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i % NACC]) : "v"(a[i & 1]), "v"(b[(i >> 1) & 1]));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i % NACC]) : "v"(a[i & 1]), "v"(b[(i >> 1) & 1]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int WAVES, bool AGPR> void run(const half8* s, float* o, unsigned long long* c) {
    const int iters = 5000;
    k<NACC, WAVES, AGPR><<<256, 64 * WAVES>>>(s, o, c, 10);
    k<NACC, WAVES, AGPR><<<256, 64 * WAVES>>>(s, o, c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%d accumulators in %s, %d wave(s) per SIMD: %.1f ticks per MFMA of wave 0\n", NACC, AGPR ? "AccVGPRs" : "VGPRs   ", WAVES / 4, double(h) / (iters * 24.0));
}
int main() {
    half8* s; float* o; unsigned long long* c;
    (void)hipMalloc(&s, 4096); (void)hipMalloc(&o, 256 * 512 * 4); (void)hipMalloc(&c, 64);
    (void)hipMemset(s, 0x3c, 4096);
    run<1, 4, false>(s, o, c); run<1, 4, true>(s, o, c);
    run<2, 4, false>(s, o, c); run<2, 4, true>(s, o, c);
    run<4, 4, false>(s, o, c); run<4, 4, true>(s, o, c);
    run<8, 4, false>(s, o, c); run<8, 4, true>(s, o, c);
    run<8, 8, false>(s, o, c); run<8, 8, true>(s, o, c);
    return 0;
}
