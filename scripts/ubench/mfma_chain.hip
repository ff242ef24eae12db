// Microbenchmark: 32x32x16 f16 MFMA rate as a function of the number of independent accumulators (dependent-chain distance).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const half8* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[2] = {src[lane], src[64 + lane]}, b[2] = {src[128 + lane], src[192 + lane]};
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], acc[i % NACC], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int WAVES> void run(const half8* s, float* o, unsigned long long* c) {
    const int iters = 5000;
    k<NACC, WAVES><<<256, 64 * WAVES>>>(s, o, c, 10);
    k<NACC, WAVES><<<256, 64 * WAVES>>>(s, o, c, iters);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%d accumulators, %d waves/SIMD: %.1f cycles per 32x32x16 MFMA (wave 0)\n", NACC, WAVES / 4, double(h) / (iters * 16.0));
}
int main() {
    half8* s; float* o; unsigned long long* c;
    hipMalloc(&s, 4096); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    hipMemset(s, 0x3c, 4096);
    run<1, 4>(s, o, c); run<2, 4>(s, o, c); run<4, 4>(s, o, c); run<8, 4>(s, o, c);
    run<1, 8>(s, o, c); run<2, 8>(s, o, c); run<4, 8>(s, o, c);
    return 0;
}
