// Which clock does s_memtime count?  (the towers' intervals are quoted in s_memtime ticks; round 4 read 1.6-1.75 GHz
// out of them, GRBM_GUI_ACTIVE / 8 XCDs / launch time reads 2.3-2.4 GHz.)  Every workgroup of a chip-filling launch stamps s_memtime and
// s_memrealtime (the constant 100 MHz counter) around a loop of one kind of work; ticks of s_memtime per second of s_memrealtime = the
// counter's rate under that load.  kind 0: s_sleep (idle chip)   1: v_fma_f32 (vector only)   2: v_mfma_f32_16x16x32_f16 back to back on
// all eight waves of every CU (the microbenchmarks' load)   3: MFMAs on half the waves at a third of the issue slots (about the towers' duty)
// usage: clock_domains.bin [iterations]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND> __global__ __launch_bounds__(512) void clk(uint64_t* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    half8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = _Float16(0.01f * float((lane + i) & 15)); b[i] = _Float16(0.02f * float((lane * 3 + i) & 7)); }
    float x = 0.001f * float(lane);
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) __builtin_amdgcn_s_sleep(64);
        if constexpr (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, 0.5, 0.5" : "+v"(x));
        }
        if constexpr (KIND == 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
        }
        if constexpr (KIND == 3) {
            if (wave < 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 24; ++u) asm volatile("v_fma_f32 %0, %0, 0.5, 0.5" : "+v"(x));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = x;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][3];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000, blocks = 256;
    uint64_t* out;
    float* sink;
    CHECK(hipMalloc(&out, blocks * 16));
    CHECK(hipMalloc(&sink, 4096));
    std::vector<uint64_t> h(blocks * 2);
    const char* names[4] = {"s_sleep (idle)", "v_fma_f32", "MFMA back to back, 8 waves per CU", "MFMA on 4 of 8 waves + v_fma_f32 (tower-like duty)"};
    for (int kind = 0; kind < 4; ++kind) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            switch (kind) {
                case 0: hipLaunchKernelGGL(clk<0>, dim3(blocks), dim3(512), 0, 0, out, sink, iters); break;
                case 1: hipLaunchKernelGGL(clk<1>, dim3(blocks), dim3(512), 0, 0, out, sink, iters * 4); break;
                case 2: hipLaunchKernelGGL(clk<2>, dim3(blocks), dim3(512), 0, 0, out, sink, iters * 4); break;
                default: hipLaunchKernelGGL(clk<3>, dim3(blocks), dim3(512), 0, 0, out, sink, iters * 4); break;
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
        }
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost));
        double rate = 0, span = 0;
        for (int bI = 0; bI < blocks; ++bI) { rate += double(h[bI * 2]) / (double(h[bI * 2 + 1]) / 100e6); span += double(h[bI * 2 + 1]) / 100e6; }
        printf("%-52s: s_memtime runs at %.3f GHz (mean over %d workgroups; loop %.3f ms by s_memrealtime, launch %.3f ms by events)\n", names[kind],
               rate / blocks / 1e9, blocks, span / blocks * 1e3, double(ms));
    }
    return 0;
}
