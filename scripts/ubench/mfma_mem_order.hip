// Microbenchmark: does the ORDER of operand fetches inside a group of 8 v_mfma_f32_32x32x16_f16 matter?  Per group: 8 MFMAs,
// 4 buffer_load_b128 (weights, L2-hot) and 8 ds_read_b128 (B fragments).  Patterns:
//   0  interleaved: every 4 MFMAs -> 4 reads + 2 loads (the tower's E step)          1  all 8 reads first, the 4 loads last
//   2  the 4 loads first, the 8 reads last                                           3  reads before MFMAs 0-3, loads before 4-7
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int PAT>
__global__ __launch_bounds__(256) void k(const char* src, float* out, unsigned long long* cyc, int groups, unsigned stream_bytes) {
    __shared__ half8 lds[2048];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = half8{1, 1, 1, 1, 1, 1, 1, 1};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + size_t(wave) * stream_bytes, 0, 0x7fffffff, 0x00020000);
    half8 win[16];
    unsigned pos = 0;
    for (int q = 0; q < 16; ++q) win[q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, pos + q * 1024, 0));
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    half8 b[8], nb[8];
    for (int i = 0; i < 8; ++i) b[i] = lds[i * 64 + lane];
    auto rd = [&](int u, int i) { nb[i] = lds[((u * 8 + i) * 64 + lane) & 2047]; };
    auto ld = [&](int u, int e) { win[(u * 4 + e) & 15] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, pos + ((u * 4 + e) & 15) * 1024 + 16384, 0)); };
    auto mm = [&](int u, int i) { acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(win[(u * 4 + (i >> 1)) & 15], b[i], acc[i & 3], 0, 0, 0); };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < groups; g += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (PAT == 0) {
                for (int i = 0; i < 4; ++i) rd(u, i);
                for (int i = 0; i < 4; ++i) mm(u, i);
                ld(u, 0); ld(u, 1);
                __builtin_amdgcn_sched_barrier(0);
                for (int i = 4; i < 8; ++i) rd(u, i);
                for (int i = 4; i < 8; ++i) mm(u, i);
                ld(u, 2); ld(u, 3);
            } else if (PAT == 1) {
                for (int i = 0; i < 8; ++i) rd(u, i);
                __builtin_amdgcn_sched_barrier(0);
                for (int i = 0; i < 8; ++i) mm(u, i);
                __builtin_amdgcn_sched_barrier(0);
                for (int e = 0; e < 4; ++e) ld(u, e);
            } else if (PAT == 2) {
                for (int i = 0; i < 4; ++i) mm(u, i);
                for (int e = 0; e < 4; ++e) ld(u, e);
                __builtin_amdgcn_sched_barrier(0);
                for (int i = 4; i < 8; ++i) mm(u, i);
                for (int i = 0; i < 8; ++i) rd(u, i);
            } else {
                for (int i = 0; i < 8; ++i) rd(u, i);
                for (int i = 0; i < 4; ++i) mm(u, i);
                __builtin_amdgcn_sched_barrier(0);
                for (int e = 0; e < 4; ++e) ld(u, e);
                for (int i = 4; i < 8; ++i) mm(u, i);
            }
            for (int i = 0; i < 8; ++i) b[i] = nb[i];
            __builtin_amdgcn_sched_barrier(0);
        }
        pos += 16 * 1024;
        if (pos + 65536 > stream_bytes) pos = 0;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0;
    for (int i = 0; i < 4; ++i) res += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int PAT>
void run(const char* s, float* o, unsigned long long* c, unsigned stream_bytes) {
    const int groups = 2000;
    k<PAT><<<256, 256>>>(s, o, c, 40, stream_bytes);
    k<PAT><<<256, 256>>>(s, o, c, groups, stream_bytes);
    (void)hipDeviceSynchronize();
    unsigned long long h[4];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d: %7.1f cycles per 8 MFMAs (bare: 261)\n", PAT, double(h[0]) / groups);
}

int main() {
    const unsigned stream_bytes = 1u << 20;
    char* s; float* o; unsigned long long* c;
    (void)hipMalloc(&s, 4 * size_t(stream_bytes) + (1 << 20));
    (void)hipMalloc(&o, 256 * 256 * 4);
    (void)hipMalloc(&c, 64);
    (void)hipMemset(s, 0x3c, 4 * size_t(stream_bytes) + (1 << 20));
    run<0>(s, o, c, stream_bytes);
    run<1>(s, o, c, stream_bytes);
    run<2>(s, o, c, stream_bytes);
    run<3>(s, o, c, stream_bytes);
    return 0;
}
