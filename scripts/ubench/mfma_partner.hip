// Microbenchmark: an MFMA wave that ALSO fetches its operands (per 4 MFMAs: NL buffer_load_b128 + ND ds_read_b128, the tower's
// matrix role) next to a partner wave on the same SIMD (waves w and w + 4 of a 512-thread workgroup) that runs
//   0 nothing   1 ds_read_b128 only   2 v_pk_fma_f16 only   3 the depthwise mix (6 pk_fma : 2 ds_read_b128)
// Reports cycles per 4 MFMAs of the MFMA wave and cycles per partner instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define REP4(x) x x x x

template <int NL, int ND, int PARTNER>
__global__ __launch_bounds__(512) void k(const char* src, float* out, unsigned long long* cyc, int steps, unsigned stream_bytes) {
    __shared__ half8 lds[4096];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = half8{1, 1, 1, 1, 1, 1, 1, 1};
    __syncthreads();
    float res = 0;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + size_t(wave) * stream_bytes, 0, 0x7fffffff, 0x00020000);
        half8 win[16];
        unsigned pos = 0;
        for (int q = 0; q < 16; ++q) win[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, pos + q * 1024, 0));
        f16v acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        half8 b[4];
        for (int i = 0; i < 4; ++i) b[i] = lds[i * 64 + lane];
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int s = 0; s < steps; s += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                half8 nb[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
                for (int i = 0; i < ND; ++i) nb[i] = lds[((u * 4 + i) * 64 + lane) & 2047];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(win[(u * 2 + (i >> 1)) & 15], b[i], acc[i], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < NL; ++e)
                    win[(u * 2 + e) & 15] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, pos + ((u * 2 + e) & 15) * 1024 + 16384, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = nb[i];
                __builtin_amdgcn_sched_barrier(0);
            }
            pos += NL * 8 * 1024;
            if (pos + 65536 > stream_bytes) pos = 0;
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) res += acc[i][0];
    } else {
        half2v a0 = {(_Float16)lane, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, x = {0.5f16, 0.25f16}, w = {0.25f16, 0.5f16};
        half8 r0 = {}, r1 = {};
        const unsigned addr = 32768u + unsigned(lane) * 16u;      // the upper half of the LDS array
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        const int iters = PARTNER == 0 ? 0 : steps * 130 / (PARTNER == 1 ? 512 : 256);   // roughly as long as the MFMA waves
        for (int it = 0; it < iters; ++it) {
            if (PARTNER == 1) {
                asm volatile(REP4("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n ds_read_b128 %0, %2 offset:1024\n ds_read_b128 %1, %2 offset:2048\n"
                                  "ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n ds_read_b128 %0, %2 offset:1024\n ds_read_b128 %1, %2 offset:2048\n")
                             "s_waitcnt lgkmcnt(0)\n" : "+v"(r0), "+v"(r1) : "v"(addr));
            } else if (PARTNER == 2) {
                asm volatile(REP4("v_pk_fma_f16 %0, %6, %7, %0\n v_pk_fma_f16 %1, %6, %7, %1\n v_pk_fma_f16 %2, %6, %7, %2\n v_pk_fma_f16 %3, %6, %7, %3\n"
                                  "v_pk_fma_f16 %4, %6, %7, %4\n v_pk_fma_f16 %5, %6, %7, %5\n v_pk_fma_f16 %0, %6, %7, %0\n v_pk_fma_f16 %1, %6, %7, %1\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x), "v"(w));
            } else {
                asm volatile(REP4("ds_read_b128 %8, %10\n v_pk_fma_f16 %0, %6, %7, %0\n v_pk_fma_f16 %1, %6, %7, %1\n v_pk_fma_f16 %2, %6, %7, %2\n"
                                  "ds_read_b128 %9, %10 offset:4096\n v_pk_fma_f16 %3, %6, %7, %3\n v_pk_fma_f16 %4, %6, %7, %4\n v_pk_fma_f16 %5, %6, %7, %5\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x), "v"(w), "v"(r0), "v"(r1), "v"(addr));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        res = float(a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0]) + float(r0[0] + r1[0]);
        if (blockIdx.x == 0 && lane == 0) cyc[8 + (wave - 4)] = iters;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int NL, int ND, int PARTNER>
void run(const char* s, float* o, unsigned long long* c, unsigned stream_bytes) {
    const int steps = 4000;
    const char* names[] = {"idle", "ds_read_b128 stream", "v_pk_fma_f16 stream", "6 pk_fma : 2 ds_read mix"};
    k<NL, ND, PARTNER><<<256, 512>>>(s, o, c, 80, stream_bytes);
    k<NL, ND, PARTNER><<<256, 512>>>(s, o, c, steps, stream_bytes);
    (void)hipDeviceSynchronize();
    unsigned long long h[12];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("MFMA wave: 4 MFMA + %d loads + %d reads | partner %-24s : %7.1f cycles per 4 MFMAs", NL, ND, names[PARTNER], double(h[0]) / steps);
    if (PARTNER) printf("   partner %6.2f cycles per instruction", double(h[4]) / (double(h[8]) * 32.0));
    printf("\n");
}

int main() {
    const unsigned stream_bytes = 1u << 20;
    char* s; float* o; unsigned long long* c;
    (void)hipMalloc(&s, 4 * size_t(stream_bytes) + (1 << 20));
    (void)hipMalloc(&o, 256 * 512 * 4);
    (void)hipMalloc(&c, 128);
    (void)hipMemset(s, 0x3c, 4 * size_t(stream_bytes) + (1 << 20));
    run<0, 0, 0>(s, o, c, stream_bytes); run<0, 0, 1>(s, o, c, stream_bytes); run<0, 0, 2>(s, o, c, stream_bytes); run<0, 0, 3>(s, o, c, stream_bytes);
    run<2, 4, 0>(s, o, c, stream_bytes); run<2, 4, 1>(s, o, c, stream_bytes); run<2, 4, 2>(s, o, c, stream_bytes); run<2, 4, 3>(s, o, c, stream_bytes);
    run<0, 4, 1>(s, o, c, stream_bytes); run<0, 4, 3>(s, o, c, stream_bytes); run<2, 0, 3>(s, o, c, stream_bytes);
    return 0;
}
