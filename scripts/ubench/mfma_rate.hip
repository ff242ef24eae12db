// Microbenchmark: sustained MFMA issue rate on gfx950 for the shapes the tower kernel could use.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, int NWAVE_ACTIVE>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(i * 64 + lane)]; b[i] = src[(256 + i * 64 + lane)]; }
    f4 acc[16];
    f16v acc32[4];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < NWAVE_ACTIVE) {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 0) {            // 16 independent 16x16x32 f16
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
            } else if (MODE == 1) {     // 16x16x32 bf16
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[i & 3]), __builtin_bit_cast(bf8, b[i >> 2]), acc[i], 0, 0, 0);
            } else if (MODE == 2) {     // 4 independent 32x32x16 f16 (x2 per iter = same flops as 16 16x16x32)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + r) & 3], acc32[i], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 4; ++i) s += acc32[i][0] + acc32[i][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int NW> void run(const char* name, const half8* d_src, float* d_out, unsigned long long* d_cyc, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NW><<<256, 512>>>(d_src, d_out, d_cyc, 10);
    hipEventRecord(e0);
    k<MODE, NW><<<256, 512>>>(d_src, d_out, d_cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double flop_per_iter_wave = 16.0 * 16384;
    const double tf = 256.0 * NW * iters * flop_per_iter_wave / (ms * 1e-3) / 1e12;
    printf("%-34s waves/WG %d: %.3f ms  %.0f TFLOP/s  s_memtime ticks per 16x16x32-equivalent MFMA: %.1f (wave0) %.1f (wave %d)\n", name, NW, ms, tf,
           double(c[0]) / (iters * 16.0), double(c[NW - 1]) / (iters * 16.0), NW - 1);
}

int main() {
    std::vector<_Float16> h(512 * 8);
    for (auto rnd : {0, 1}) {
        for (size_t i = 0; i < h.size(); ++i) h[i] = rnd ? _Float16((rand() % 2001 - 1000) / 1000.f) : _Float16(0.f);
        half8* d_src; float* d_out; unsigned long long* d_cyc;
        hipMalloc(&d_src, h.size() * 2); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 64);
        hipMemcpy(d_src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        printf("== %s operands\n", rnd ? "random" : "zero");
        const int it = 20000;
        run<0, 4>("16x16x32 f16", d_src, d_out, d_cyc, it);
        run<0, 8>("16x16x32 f16", d_src, d_out, d_cyc, it);
        run<1, 4>("16x16x32 bf16", d_src, d_out, d_cyc, it);
        run<2, 4>("32x32x16 f16", d_src, d_out, d_cyc, it);
        run<2, 8>("32x32x16 f16", d_src, d_out, d_cyc, it);
    }
    return 0;
}
