// Microbenchmark for the "two boards per workgroup, four fat waves" tower candidate: ONE wave per SIMD
// (256-thread workgroups, up to 512 registers per wave) that does its matrix steps AND its share of the depthwise itself.
// Per step: 4 x v_mfma_f32_32x32x16_f16 + NL buffer_load_b128 (1 KiB of weight stream, L2-hot, 16 in flight) + ND ds_read_b128
// + NV v_pk_fma_f16 (8 independent chains).  Reports cycles per step; 4 bare MFMAs = 130.
// The candidate's mix per 4 MFMAs: 1 load (a weight fragment feeds 4 MFMAs), 3 B-fragment reads + 2 depthwise neighbour reads,
// 9-12 packed FMAs + a few moves.  Compare with today's matrix wave: 2 loads + 3 reads per 4 MFMAs next to a partner wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NL, int ND, int NV>
__global__ __launch_bounds__(256) void k(const char* src, float* out, unsigned long long* cyc, int steps, unsigned stream_bytes) {
    __shared__ half8 lds[4096];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = half8{1, 1, 1, 1, 1, 1, 1, 1};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + size_t(wave) * stream_bytes, 0, 0x7fffffff, 0x00020000);
    half8 win[16];
    unsigned pos = 0;
    for (int q = 0; q < 16; ++q) win[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, pos + q * 1024, 0));
    f16v acc[8];                                     // 8 accumulator tiles: the candidate holds 2 x 4 project tiles + expand tiles
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    half8 b[4];
    for (int i = 0; i < 4; ++i) b[i] = lds[i * 64 + lane];
    half2v v[8];
    for (int i = 0; i < 8; ++i) v[i] = half2v{(_Float16)lane, (_Float16)i};
    const half2v wx = {0.5f16, 0.25f16}, ww = {0.25f16, 0.5f16};
    half8 sink = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            half8 nb[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const half8 r = lds[((u * 8 + i) * 64 + lane) & 4095];
                if (i < 4) nb[i] = r; else sink = r;
            }
            __builtin_amdgcn_sched_barrier(0);       // the next step's reads are issued BEFORE this step's MFMAs (else the scheduler sinks them)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[(u & 1) * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(win[(u * 2 + (NL == 1 ? 0 : (i >> 1))) & 15], b[i], acc[(u & 1) * 4 + i], 0, 0, 0);
                // the packed FMAs of this step are spread behind the MFMAs (the compiler keeps this order: sched_barrier below)
#pragma unroll
                for (int e = 0; e < (NV + 3 - i) / 4; ++e) {
                    const int c = (i * ((NV + 3) / 4) + e) & 7;
                    v[c] = __builtin_elementwise_fma(v[c], wx, ww);
                }
            }
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
    float res = float(sink[0]);
    for (int i = 0; i < 8; ++i) res += acc[i][0] + float(v[i][0]);
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int NL, int ND, int NV>
void run(const char* s, float* o, unsigned long long* c, unsigned stream_bytes) {
    const int steps = 4000;
    k<NL, ND, NV><<<256, 256>>>(s, o, c, 80, stream_bytes);
    k<NL, ND, NV><<<256, 256>>>(s, o, c, steps, stream_bytes);
    (void)hipDeviceSynchronize();
    unsigned long long h[4];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("loads %d  ds_reads %d  pk_fma %2d  per 4 MFMAs: %7.1f cycles per step  (%5.1f per MFMA)\n", NL, ND, NV, double(h[0]) / steps, double(h[0]) / steps / 4);
}

int main() {
    const unsigned stream_bytes = 256u << 10;      // 1 MiB per workgroup: stays in the XCD L2 (the tower warms its stream into L2 ahead of use)
    char* s; float* o; unsigned long long* c;
    hipMalloc(&s, size_t(stream_bytes) * 4); hipMalloc(&o, 256 * 256 * 4); hipMalloc(&c, 64);
    hipMemset(s, 0x3c, size_t(stream_bytes) * 4);
    run<0, 0, 0>(s, o, c, stream_bytes);
    run<0, 4, 0>(s, o, c, stream_bytes);
    run<1, 0, 0>(s, o, c, stream_bytes);
    run<2, 0, 0>(s, o, c, stream_bytes);
    run<1, 4, 0>(s, o, c, stream_bytes);
    run<1, 5, 0>(s, o, c, stream_bytes);
    run<1, 5, 4>(s, o, c, stream_bytes);
    run<1, 5, 8>(s, o, c, stream_bytes);
    run<1, 5, 12>(s, o, c, stream_bytes);
    run<1, 5, 16>(s, o, c, stream_bytes);
    run<1, 6, 12>(s, o, c, stream_bytes);
    run<1, 4, 12>(s, o, c, stream_bytes);
    run<2, 4, 0>(s, o, c, stream_bytes);      // today's matrix wave (without its partner)
    run<2, 4, 12>(s, o, c, stream_bytes);
    run<0, 0, 12>(s, o, c, stream_bytes);
    run<0, 0, 16>(s, o, c, stream_bytes);
    return 0;
}
