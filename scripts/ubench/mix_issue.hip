// Microbenchmark: MFMAs and VALU instructions in ONE instruction stream, one or two such waves per SIMD -- how many cycles does the SIMD
// spend per MFMA when every MFMA is followed by V independent VALU instructions (grouped G MFMAs, then G * V VALUs)?
// The float16x3 tower's depthwise (~530 VALU per 384 MFMAs and SIMD) is the case: V ~ 1.4.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mix_issue.hip -o /tmp/mix_issue && /tmp/mix_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int V> __device__ __forceinline__ void valus(float (&r)[8], float x, float w) {
    if constexpr (V >= 1) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[0]) : "v"(x), "v"(w));
    if constexpr (V >= 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[1]) : "v"(x), "v"(w));
    if constexpr (V >= 3) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[2]) : "v"(x), "v"(w));
    if constexpr (V >= 4) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[3]) : "v"(x), "v"(w));
}

// BIG = 0: v_mfma_f32_16x16x32_f16, 16 per iteration; BIG = 1: v_mfma_f32_32x32x16_f16, 8 per iteration with 2 V VALUs each
template <int V, int G, int BIG> __global__ __launch_bounds__(512) void k(const half8* src, float* out, unsigned long long* cyc, int iters, int nwaves) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[i * 64 + lane]; b[i] = src[256 + i * 64 + lane]; }
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = lane + i;
    float x = 0.5f, w = 0.25f, res = 0;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    if (wave < nwaves) {
        if constexpr (BIG == 0) {
            f4 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 16 / G; ++g) {
#pragma unroll
                    for (int i = 0; i < G; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[g * G + i]) : "v"(a[i & 3]), "v"(b[g & 3]));
#pragma unroll
                    for (int i = 0; i < G; ++i) valus<V>(r, x, w);
                }
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 16; ++i) res += acc[i][0];
        } else {
            f16v acc[8];
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
            constexpr int GB = G >= 2 ? G / 2 : 1;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 8 / GB; ++g) {
#pragma unroll
                    for (int i = 0; i < GB; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g * GB + i]) : "v"(a[i & 3]), "v"(b[g & 3]));
#pragma unroll
                    for (int i = 0; i < GB; ++i) { valus<V>(r, x, w); valus<V>(r, x, w); }
                }
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 8; ++i) res += acc[i][0];
        }
    }
    for (int i = 0; i < 8; ++i) res += r[i];
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int V, int G, int BIG> void run(const half8* s, float* o, unsigned long long* c) {
    const int iters = 2000;
    for (int nw : {4, 8}) {
        k<V, G, BIG><<<256, 512>>>(s, o, c, 10, nw);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<V, G, BIG><<<256, 512>>>(s, o, c, iters, nw);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double per_simd = double(h[0]) / (iters * 16.0 * (nw / 4));      // cycles of the SIMD per 16x16x32-equivalent MFMA
        printf("%s  V=%d VALU per MFMA-equivalent, groups of %2d, %d wave(s) per SIMD: %6.2f cycles per MFMA-equivalent and SIMD (wave 0: %.2f per own MFMA-eq)\n",
               BIG ? "32x32x16" : "16x16x32", V, G, nw / 4, per_simd, double(h[0]) / (iters * 16.0));
        printf("      wall clock: %.3f ms = %.0f TFLOP/s f16 over 256 CUs, shader clock by the counter %.2f GHz\n", ms, 256.0 * nw * iters * 16 * 16384.0 / (ms * 1e-3) / 1e12,
               double(h[0]) / (ms * 1e-3) / 1e9);
    }
}

int main() {
    half8* s; float* o; unsigned long long* c;
    hipMalloc(&s, 512 * 16); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    hipMemset(s, 0x3c, 512 * 16);
    run<0, 1, 0>(s, o, c); run<1, 1, 0>(s, o, c); run<2, 1, 0>(s, o, c); run<3, 1, 0>(s, o, c); run<4, 1, 0>(s, o, c);
    run<1, 4, 0>(s, o, c); run<2, 4, 0>(s, o, c); run<2, 16, 0>(s, o, c);
    run<0, 2, 1>(s, o, c); run<1, 2, 1>(s, o, c); run<2, 2, 1>(s, o, c); run<3, 2, 1>(s, o, c); run<4, 2, 1>(s, o, c);
    run<2, 8, 1>(s, o, c);
    return 0;
}
