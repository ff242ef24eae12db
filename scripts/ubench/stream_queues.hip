// Which HIP streams of one process share a hardware queue?  Two one-workgroup kernels that each spin for 2 ms finish after 2 ms
// when their streams sit on different queues and after 4 ms on one queue.  Prints the matrix for 8 streams in creation order, then
// destroys two and makes two new ones (what closing and opening nets in one process does).
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/stream_queues.hip -o /tmp/stream_queues && /tmp/stream_queues
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(long long ticks, long long* out) {
    const long long t0 = wall_clock64();
    long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (out) *out = t - t0;
}

static double pair_ms(hipStream_t a, hipStream_t b, long long ticks) {
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    spin<<<1, 64, 0, a>>>(ticks, nullptr);
    spin<<<1, 64, 0, b>>>(ticks, nullptr);
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static void matrix(const std::vector<hipStream_t>& s, long long ticks) {
    printf("      ");
    for (size_t j = 0; j < s.size(); ++j) printf("  s%zu  ", j);
    printf("\n");
    for (size_t i = 0; i < s.size(); ++i) {
        printf("  s%zu  ", i);
        for (size_t j = 0; j < s.size(); ++j) {
            if (j <= i) { printf("   .  "); continue; }
            printf(" %4.1f%c", pair_ms(s[i], s[j], ticks), ' ');
        }
        printf("\n");
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = 2LL * rate_khz;              // 2 ms
    printf("wall clock %d kHz, spin %lld ticks, GPU_MAX_HW_QUEUES=%s\n", rate_khz, ticks, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");
    std::vector<hipStream_t> s(n);
    for (int i = 0; i < n; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    for (int i = 0; i < n; ++i) { spin<<<1, 64, 0, s[i]>>>(100, nullptr); CK(hipStreamSynchronize(s[i])); }     // first use binds the queue
    printf("one kernel alone: %.2f ms\n", pair_ms(s[0], s[0], ticks) / 2);
    printf("-- %d streams in creation order (ms for two 2 ms kernels)\n", n);
    matrix(s, ticks);
    printf("-- against the null stream\n");
    for (int i = 0; i < n; ++i) printf("  s%d %4.1f", i, pair_ms(nullptr, s[i], ticks));
    printf("\n");
    if (n >= 4) {
        CK(hipStreamDestroy(s[1]));
        CK(hipStreamDestroy(s[2]));
        CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&s[2], hipStreamNonBlocking));
        spin<<<1, 64, 0, s[1]>>>(100, nullptr); spin<<<1, 64, 0, s[2]>>>(100, nullptr);
        CK(hipDeviceSynchronize());
        printf("-- s1 and s2 destroyed and made again\n");
        matrix(s, ticks);
    }
    {   // a pool made once and never destroyed, used in pairs: what a library-owned set of streams would give
        std::vector<hipStream_t> p(4);
        for (auto& x : p) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        for (auto& x : p) { spin<<<1, 64, 0, x>>>(100, nullptr); CK(hipStreamSynchronize(x)); }
        printf("-- four more streams made after the eight\n");
        matrix(p, ticks);
    }
    return 0;
}
