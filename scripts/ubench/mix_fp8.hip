// Microbenchmark for the mixed split (f16 main term + fp8 cross terms, profiles/NOTES.md round 4): issue cost per SIMD of
//   kind 0  v_mfma_f32_16x16x32_f16           (K = 32, the float16x3 tower's instruction)
//   kind 1  v_mfma_f32_16x16x128_f8f6f4       (K = 128, e4m3 x e4m3)
//   kind 2  v_mfma_f32_16x16x32_fp8_fp8       (K = 32, the gfx940 form)
// back to back in one wave, in two waves of a SIMD, and with V independent v_fmac_f32 behind every MFMA (the depthwise in the shadow).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef int i2v __attribute__((ext_vector_type(2)));

template <int V> __device__ __forceinline__ void valus(float (&r)[8], float x, float w) {
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[i & 7]) : "v"(x), "v"(w));
}

template <int KIND, int V> __global__ __launch_bounds__(512) void k(const int* src, float* out, unsigned long long* cyc, int iters, int nwaves) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    i8v a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = src[i * 64 + lane]; b8[i] = src[512 + i * 64 + lane]; }
    half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<const f4*>(&a8)), bh = __builtin_bit_cast(half8, *reinterpret_cast<const f4*>(&b8));
    i2v a2 = {a8[0], a8[1]}, b2 = {b8[0], b8[1]};
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = lane + i;
    float x = 0.5f, w = 0.25f, res = 0;
    unsigned long long t0 = 0, t1 = 0;
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    __syncthreads();
    if (wave < nwaves) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ah), "v"(bh));
                if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
                if constexpr (KIND == 2) asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a2), "v"(b2));
                valus<V>(r, x, w);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) res += acc[i][0];
    }
    for (int i = 0; i < 8; ++i) res += r[i];
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int KIND, int V> void run(const int* src, float* out, unsigned long long* cyc, const char* name) {
    const int iters = 2000;
    for (int nw : {4, 8}) {
        hipLaunchKernelGGL((k<KIND, V>), dim3(256), dim3(512), 0, 0, src, out, cyc, iters, nw);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on gfx9?  report the ratio to kind 0 as well as raw ticks
        printf("%-34s V=%d  %d wave(s) per SIMD: %8.2f ticks per MFMA (wave 0), %8.2f (wave %d)\n", name, V, nw / 4, double(h[0]) / (8.0 * iters),
               double(h[nw - 1]) / (8.0 * iters), nw - 1);
    }
}

int main() {
    int* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 1024 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    int h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0x38383838;          // e4m3 1.0 bytes / harmless f16 bit patterns
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>(src, out, cyc, "v_mfma_f32_16x16x32_f16");
    run<1, 0>(src, out, cyc, "v_mfma_f32_16x16x128_f8f6f4 (fp8)");
    run<2, 0>(src, out, cyc, "v_mfma_f32_16x16x32_fp8_fp8");
    run<0, 2>(src, out, cyc, "v_mfma_f32_16x16x32_f16");
    run<1, 2>(src, out, cyc, "v_mfma_f32_16x16x128_f8f6f4 (fp8)");
    run<1, 4>(src, out, cyc, "v_mfma_f32_16x16x128_f8f6f4 (fp8)");
    run<1, 6>(src, out, cyc, "v_mfma_f32_16x16x128_f8f6f4 (fp8)");
    run<1, 8>(src, out, cyc, "v_mfma_f32_16x16x128_f8f6f4 (fp8)");
    run<2, 1>(src, out, cyc, "v_mfma_f32_16x16x32_fp8_fp8");
    run<2, 2>(src, out, cyc, "v_mfma_f32_16x16x32_fp8_fp8");
    return 0;
}
