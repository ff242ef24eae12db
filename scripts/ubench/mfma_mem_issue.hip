// Microbenchmark: what do vector-memory and LDS instructions cost the ISSUING wave when they sit between its own
// v_mfma_f32_32x32x16_f16?  One wave per SIMD (256-thread workgroups, 256 of them).  Per step: 4 MFMAs + NL buffer_load_b128
// (1 KiB per wave, L2-hot stream, 16 loads in flight) + ND ds_read_b128.  Reports cycles per step; 4 bare MFMAs = 130.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NL, int ND, int SHARE>
__global__ __launch_bounds__(256) void k(const char* src, float* out, unsigned long long* cyc, int steps, unsigned stream_bytes) {
    __shared__ half8 lds[2048];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = half8{1, 1, 1, 1, 1, 1, 1, 1};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + size_t(SHARE ? (wave >> 1) : wave) * stream_bytes, 0, 0x7fffffff, 0x00020000);
    half8 win[16];
    unsigned pos = 0;
    for (int q = 0; q < 16; ++q) win[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, pos + q * 1024, 0));
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    half8 b[4];
    for (int i = 0; i < 4; ++i) b[i] = lds[i * 64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            half8 nb[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const half8 r = lds[((u * 8 + i) * 64 + lane) & 2047];
                if (i < 4) nb[i] = r; else asm volatile("" ::"v"(r));
            }
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
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0;
    for (int i = 0; i < 4; ++i) res += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int NL, int ND, int SHARE = 0>
void run(const char* s, float* o, unsigned long long* c, unsigned stream_bytes) {
    const int steps = 4000;
    k<NL, ND, SHARE><<<256, 256>>>(s, o, c, 80, stream_bytes);
    k<NL, ND, SHARE><<<256, 256>>>(s, o, c, steps, stream_bytes);
    (void)hipDeviceSynchronize();
    unsigned long long h[4];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("%sper step: 4 MFMA + %d buffer_load_b128 + %d ds_read_b128 : %7.1f cycles/step (wave 0), %7.1f (wave 3)\n", SHARE ? "[pairs of waves share a stream] " : "", NL, ND, double(h[0]) / steps, double(h[3]) / steps);
}

int main() {
    const unsigned stream_bytes = 1u << 20;   // per wave; 4 streams = 4 MiB: L2-resident after the warm-up launch
    char* s; float* o; unsigned long long* c;
    (void)hipMalloc(&s, 4 * size_t(stream_bytes) + (1 << 20));
    (void)hipMalloc(&o, 256 * 256 * 4);
    (void)hipMalloc(&c, 64);
    (void)hipMemset(s, 0x3c, 4 * size_t(stream_bytes) + (1 << 20));
    run<0, 0>(s, o, c, stream_bytes);
    run<0, 2>(s, o, c, stream_bytes);
    run<0, 4>(s, o, c, stream_bytes);
    run<1, 0>(s, o, c, stream_bytes);
    run<2, 0>(s, o, c, stream_bytes);
    run<1, 2>(s, o, c, stream_bytes);
    run<2, 4>(s, o, c, stream_bytes);
    run<1, 4>(s, o, c, stream_bytes);
    run<0, 6>(s, o, c, stream_bytes);
    run<0, 8>(s, o, c, stream_bytes);
    run<0, 12>(s, o, c, stream_bytes);
    run<2, 2>(s, o, c, stream_bytes);
    run<2, 0, 1>(s, o, c, stream_bytes);
    run<2, 4, 1>(s, o, c, stream_bytes);
    run<4, 0, 0>(s, o, c, stream_bytes);
    run<4, 0, 1>(s, o, c, stream_bytes);
    return 0;
}
