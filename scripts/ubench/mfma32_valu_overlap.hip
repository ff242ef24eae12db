// Microbenchmark: issue rate a VALU/LDS wave gets next to an MFMA-saturated wave on the SAME SIMD (waves w and w+4 of a
// 512-thread workgroup), for the two f16 MFMA shapes.  shape 0: 16x16x32 (8 passes), shape 1: 32x32x16 (16 passes).
// vector mixes: 0 = v_pk_fma_f16 only, 1 = 4 pk_fma : 1 ds_read_b128 (the depthwise loop's mix), 2 = ds_read_b128 only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define REP4(x) x x x x

template <int SHAPE, int MIX>
__global__ __launch_bounds__(512) void k(const half8* src, float* out, unsigned long long* cyc, int iters, int mode) {
    __shared__ half8 lds[1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lds[threadIdx.x] = src[threadIdx.x & 255];
    lds[threadIdx.x + 512] = src[threadIdx.x & 255];
    unsigned long long t0 = 0, t1 = 0;
    float res = 0;
    if (wave < 4) {
        half8 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = src[i * 64 + lane]; b[i] = src[256 + i * 64 + lane]; }
        __syncthreads();
        if (SHAPE == 0) {
            f4 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
            t0 = __builtin_amdgcn_s_memtime();
            if (mode & 1)
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
                }
            t1 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 16; ++i) res += acc[i][0];
        } else {
            f16v acc[4];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
            t0 = __builtin_amdgcn_s_memtime();
            if (mode & 1)
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 2], acc[i & 3], 0, 0, 0);
                }
            t1 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 4; ++i) res += acc[i][0];
        }
    } else {
        half2v a0 = {(_Float16)lane, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, x = {0.5f16, 0.25f16}, w = {0.25f16, 0.5f16};
        half8 r0 = {}, r1 = {};
        const unsigned addr = unsigned(lane) * 16u;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
                if (MIX == 0) {
                    asm volatile(REP4("v_pk_fma_f16 %0, %8, %9, %0\n v_pk_fma_f16 %1, %8, %9, %1\n v_pk_fma_f16 %2, %8, %9, %2\n v_pk_fma_f16 %3, %8, %9, %3\n"
                                      "v_pk_fma_f16 %4, %8, %9, %4\n v_pk_fma_f16 %5, %8, %9, %5\n v_pk_fma_f16 %6, %8, %9, %6\n v_pk_fma_f16 %7, %8, %9, %7\n")
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
                } else if (MIX == 1) {   // 32 instr: per 8: 6 pk_fma + 2 ds_read (waitcnt at the end of the group)
                    asm volatile(REP4("ds_read_b128 %10, %12\n v_pk_fma_f16 %0, %8, %9, %0\n v_pk_fma_f16 %1, %8, %9, %1\n v_pk_fma_f16 %2, %8, %9, %2\n"
                                      "ds_read_b128 %11, %12 offset:4096\n v_pk_fma_f16 %4, %8, %9, %4\n v_pk_fma_f16 %5, %8, %9, %5\n v_pk_fma_f16 %6, %8, %9, %6\n")
                                 "s_waitcnt lgkmcnt(0)\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w), "v"(r0), "v"(r1), "v"(addr));
                } else {
                    asm volatile(REP4("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n ds_read_b128 %0, %2 offset:1024\n ds_read_b128 %1, %2 offset:2048\n"
                                      "ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n ds_read_b128 %0, %2 offset:1024\n ds_read_b128 %1, %2 offset:2048\n")
                                 "s_waitcnt lgkmcnt(0)\n"
                                 : "+v"(r0), "+v"(r1) : "v"(addr));
                }
            }
        t1 = __builtin_amdgcn_s_memtime();
        res = float(a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0]) + float(r0[0] + r1[0]);
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int SHAPE, int MIX>
void run(const half8* s, float* o, unsigned long long* c) {
    const int iters = 4000;
    const char* shapes[] = {"16x16x32", "32x32x16"};
    const char* mixes[] = {"pk_fma", "6 pk_fma : 2 ds_read_b128", "ds_read_b128"};
    for (int mode : {1, 2, 3}) {
        k<SHAPE, MIX><<<256, 512>>>(s, o, c, 10, mode);
        k<SHAPE, MIX><<<256, 512>>>(s, o, c, iters, mode);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s | %-26s | %-10s  cyc/MFMA %6.2f   cyc/vector-instr %6.2f\n", shapes[SHAPE], mixes[MIX], mode == 1 ? "MFMA only" : mode == 2 ? "vec only" : "both",
               (mode & 1) ? double(h[0]) / (iters * 16.0) : 0.0, (mode & 2) ? double(h[4]) / (iters * 32.0) : 0.0);
    }
}

int main() {
    half8* s; float* o; unsigned long long* c;
    hipMalloc(&s, 512 * 16); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    hipMemset(s, 0x3c, 512 * 16);
    run<0, 0>(s, o, c); run<1, 0>(s, o, c);
    run<0, 1>(s, o, c); run<1, 1>(s, o, c);
    run<0, 2>(s, o, c); run<1, 2>(s, o, c);
    return 0;
}
