// Microbenchmark: effective shader clock (s_memtime cycles / wall time) as a function of MFMA shape and duty cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE, int FILL>   // SHAPE 0: 16x16x32, 1: 32x32x16 ; FILL: v_nop-like VALU fillers per 16x16x32-equivalent MFMA
__global__ __launch_bounds__(256) void k(const half8* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(i * 64 + lane)]; b[i] = src[(256 + i * 64 + lane)]; }
    f4 acc[16]; f16v acc32[4];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0;
    float f0 = lane, f1 = lane + 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < FILL; ++q) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f0) : "v"(f1));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 1], acc32[i & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2 * FILL; ++q) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f0) : "v"(f1));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = f0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 4; ++i) s += acc32[i][0] + acc32[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int FILL> void run(const half8* d_src, float* d_out, unsigned long long* d_cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, FILL><<<256, 256>>>(d_src, d_out, d_cyc, 100);
    hipEventRecord(e0);
    k<SHAPE, FILL><<<256, 256>>>(d_src, d_out, d_cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    const double tf = 256.0 * 4 * iters * 16.0 * 16384 / (ms * 1e-3) / 1e12;
    printf("%s fillers/MFMA-equiv %d: %7.3f ms  %5.0f TFLOP/s  %5.1f cycles per MFMA-equiv  clock %.2f GHz\n", SHAPE ? "32x32x16" : "16x16x32", FILL, ms, tf,
           double(c) / (iters * 16.0), double(c) / (ms * 1e-3) / 1e9);
}

int main() {
    std::vector<_Float16> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = _Float16((rand() % 2001 - 1000) / 1000.f);
    half8* d_src; float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_src, h.size() * 2); hipMalloc(&d_out, 256 * 256 * 4); hipMalloc(&d_cyc, 64);
    hipMemcpy(d_src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0, 0>(d_src, d_out, d_cyc); run<0, 1>(d_src, d_out, d_cyc); run<0, 2>(d_src, d_out, d_cyc); run<0, 4>(d_src, d_out, d_cyc); run<0, 8>(d_src, d_out, d_cyc);
    run<1, 0>(d_src, d_out, d_cyc); run<1, 1>(d_src, d_out, d_cyc); run<1, 2>(d_src, d_out, d_cyc); run<1, 4>(d_src, d_out, d_cyc); run<1, 8>(d_src, d_out, d_cyc);
    return 0;
}
