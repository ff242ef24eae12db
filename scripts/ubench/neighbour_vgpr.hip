// Do the VGPRs of a wave survive a neighbour?  (profiles/NOTES.md round 4: value_head_kernel's FC1 sums came out wrong in lanes 48-63 of
// one register whenever workgroups of the policy conv with the fused softmax shared its compute unit.)
//
//   victim    : a wave fills NREG registers with a pattern of (register, lane), keeps them alive through `spin` rounds of an FMA that does
//               not change them (x = fma(x, 1, 0)) with a short sleep in between, then checks every register in every lane and reports
//               the first differences: [block, wave, register, lane, expected bits, found bits]
//   aggressor : waves running one instruction kind in a loop on their own registers:
//               0 v_exp_f32   1 v_log_f32   2 v_fma_f32 (control)   3 ds_bpermute_b32   4 v_rcp_f32   5 global_store   6 v_sqrt_f32
//               7 v_exp_f32 at the very end of the wave (exp, then straight into s_endpgm)
//
// Both kernels are small (no LDS, few registers), so workgroups of both share compute units and SIMDs when they are launched on two
// streams at the same time.  usage: neighbour_vgpr.bin <kind> [launches] [victim spin] [aggressor iterations]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NREG = 48;

__device__ __forceinline__ uint32_t pattern(int reg, int lane, int salt) { return 0x3f000000u | (uint32_t(salt & 0xff) << 14) | (uint32_t(reg) << 8) | uint32_t(lane); }

__global__ __launch_bounds__(256) void victim(uint32_t* report, uint32_t* count, int spin, int salt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = __builtin_bit_cast(float, pattern(i, lane, salt));
    for (int s = 0; s < spin; ++s) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) asm volatile("v_fma_f32 %0, %0, 1.0, 0" : "+v"(r[i]));       // value unchanged, register rewritten
        __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const uint32_t got = __builtin_bit_cast(uint32_t, r[i]), want = pattern(i, lane, salt);
        if (got != want) {
            const uint32_t k = atomicAdd(count, 1u);
            if (k < 256) {
                uint32_t* o = report + k * 6;
                o[0] = blockIdx.x; o[1] = uint32_t(wave); o[2] = uint32_t(i); o[3] = uint32_t(lane); o[4] = want; o[5] = got;
            }
        }
    }
}

template <int KIND> __global__ __launch_bounds__(512) void aggressor(float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * float(lane + i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == 0 || KIND == 7) asm volatile("v_exp_f32 %0, %0\n\ts_nop 1\n\tv_mul_f32 %0, 0.5, %0" : "+v"(x[i]));
            if constexpr (KIND == 1) asm volatile("v_log_f32 %0, %0\n\ts_nop 1\n\tv_max_f32 %0, 0.5, %0" : "+v"(x[i]));
            if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %0, 0.5, 0.5" : "+v"(x[i]));
            if constexpr (KIND == 3) x[i] = __shfl_xor(x[i], 1 << (i % 6), 64);
            if constexpr (KIND == 4) asm volatile("v_rcp_f32 %0, %0\n\ts_nop 1\n\tv_add_f32 %0, 1.0, %0" : "+v"(x[i]));
            if constexpr (KIND == 5) sink[(size_t(blockIdx.x) * 512 + threadIdx.x) * 8 + i] = x[i] + float(it);
            if constexpr (KIND == 6) asm volatile("v_sqrt_f32 %0, %0\n\ts_nop 1\n\tv_add_f32 %0, 1.0, %0" : "+v"(x[i]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) sink[threadIdx.x] = s;             // (keeps the loop alive)
    if constexpr (KIND == 7) {                             // a transcendental as the wave's last instruction: its result is never read
        float y = s;
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_log_f32 %2, %2\n\ts_endpgm" : "+v"(y), "+v"(x[0]), "+v"(x[1]));
    }
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0, launches = argc > 2 ? atoi(argv[2]) : 2000, spin = argc > 3 ? atoi(argv[3]) : 200,
              iters = argc > 4 ? atoi(argv[4]) : 400;
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    uint32_t *report, *count;
    float* sink;
    CHECK(hipMalloc(&report, 256 * 6 * 4));
    CHECK(hipMalloc(&count, 4));
    CHECK(hipMalloc(&sink, size_t(1024) * 512 * 8 * 4));
    CHECK(hipMemset(count, 0, 4));
    const int vblocks = 512, ablocks = 512;
    for (int l = 0; l < launches; ++l) {
        hipLaunchKernelGGL(victim, dim3(vblocks), dim3(256), 0, sv, report, count, spin, l);
        switch (kind) {
            case 0: hipLaunchKernelGGL(aggressor<0>, dim3(ablocks), dim3(512), 0, sa, sink, iters); break;
            case 1: hipLaunchKernelGGL(aggressor<1>, dim3(ablocks), dim3(512), 0, sa, sink, iters); break;
            case 2: hipLaunchKernelGGL(aggressor<2>, dim3(ablocks), dim3(512), 0, sa, sink, iters); break;
            case 3: hipLaunchKernelGGL(aggressor<3>, dim3(ablocks), dim3(512), 0, sa, sink, iters); break;
            case 4: hipLaunchKernelGGL(aggressor<4>, dim3(ablocks), dim3(512), 0, sa, sink, iters); break;
            case 5: hipLaunchKernelGGL(aggressor<5>, dim3(ablocks), dim3(512), 0, sa, sink, iters); break;
            case 6: hipLaunchKernelGGL(aggressor<6>, dim3(ablocks), dim3(512), 0, sa, sink, iters); break;
            case 7: hipLaunchKernelGGL(aggressor<7>, dim3(ablocks), dim3(512), 0, sa, sink, 4); break;
            default: break;                                  // -1: the victim alone
        }
        if ((l & 63) == 63) { CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sa)); }
    }
    CHECK(hipDeviceSynchronize());
    uint32_t n = 0;
    std::vector<uint32_t> rep(256 * 6);
    CHECK(hipMemcpy(&n, count, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(rep.data(), report, rep.size() * 4, hipMemcpyDeviceToHost));
    printf("aggressor kind %d: %u corrupted register lanes in %d victim launches (%d blocks x 4 waves x %d registers)\n", kind, n, launches, vblocks, NREG);
    for (uint32_t k = 0; k < n && k < 24; ++k)
        printf("  block %u wave %u register %u lane %u: expected %08x found %08x (%g)\n", rep[k * 6], rep[k * 6 + 1], rep[k * 6 + 2], rep[k * 6 + 3],
               rep[k * 6 + 4], rep[k * 6 + 5], double(__builtin_bit_cast(float, rep[k * 6 + 5])));
    return 0;
}
