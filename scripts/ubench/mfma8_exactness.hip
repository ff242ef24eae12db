// How exact is v_mfma_f32_16x16x128_f8f6f4 on e5m2 (and e4m3) operands?  The products of two 8-bit floats are exact in f32; what the
// instruction does with the 128 of them before they reach the f32 accumulator is not documented.  Random operands with a chosen spread of
// exponents, the MFMA's result against the exact sum (double): max and mean |error| relative to the largest |product| of the dot product and
// relative to the sum of |products|, and the sign of the mean error (a truncating adder tree shows up as a bias).
//   usage: mfma8_exactness.bin [format 0 = e4m3, 1 = e5m2] [exponent spread in binades] [trials]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FMT> __global__ void one_mfma(const uint8_t* A, const uint8_t* B, float* D) {      // A [16][128], B [16 cols][128] row-major in k
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = *reinterpret_cast<const int*>(A + r * 128 + g * 32 + i * 4);
        b[i] = *reinterpret_cast<const int*>(B + r * 128 + g * 32 + i * 4);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FMT, FMT, 0, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(g * 4 + i) * 16 + r] = c[i];         // D[row of A][row (= column) of B]
}

static double decode(uint8_t v, int fmt) {
    const int s = v >> 7;
    double x;
    if (fmt == 1) {
        const int e = (v >> 2) & 31, m = v & 3;
        x = e == 0 ? std::ldexp(m / 4.0, -14) : std::ldexp(1.0 + m / 4.0, e - 15);
    } else {
        const int e = (v >> 3) & 15, m = v & 7;
        x = e == 0 ? std::ldexp(m / 8.0, -6) : std::ldexp(1.0 + m / 8.0, e - 7);
    }
    return s ? -x : x;
}

int main(int argc, char** argv) {
    const int fmt = argc > 1 ? atoi(argv[1]) : 1, spread = argc > 2 ? atoi(argv[2]) : 4, trials = argc > 3 ? atoi(argv[3]) : 200;
    std::mt19937 rng(3);
    uint8_t *dA, *dB;
    float* dD;
    CHECK(hipMalloc(&dA, 2048));
    CHECK(hipMalloc(&dB, 2048));
    CHECK(hipMalloc(&dD, 1024));
    double max_rel_big = 0, max_rel_abs = 0, mean_err = 0, mean_abs = 0;
    long count = 0;
    for (int t = 0; t < trials; ++t) {
        std::vector<uint8_t> A(2048), B(2048);
        const int bias = fmt == 1 ? 15 : 7, mbits = fmt == 1 ? 2 : 3;
        for (auto* v : {&A, &B})
            for (auto& q : *v) {
                const int e = bias - spread / 2 + int(rng() % unsigned(spread + 1));
                q = uint8_t(((rng() & 1) << 7) | (e << mbits) | (rng() & ((1 << mbits) - 1)));
            }
        CHECK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
        if (fmt == 1) hipLaunchKernelGGL(one_mfma<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        else hipLaunchKernelGGL(one_mfma<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        std::vector<float> D(256);
        CHECK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double sum = 0, big = 0, sabs = 0;
                for (int k = 0; k < 128; ++k) {
                    const double p = decode(A[i * 128 + k], fmt) * decode(B[j * 128 + k], fmt);
                    sum += p; sabs += std::fabs(p); big = std::max(big, std::fabs(p));
                }
                const double err = double(D[i * 16 + j]) - sum;
                max_rel_big = std::max(max_rel_big, std::fabs(err) / big);
                max_rel_abs = std::max(max_rel_abs, std::fabs(err) / sabs);
                mean_err += err / sabs; mean_abs += std::fabs(err) / sabs;
                ++count;
            }
    }
    printf("format %s, exponents over %d binades, %ld dot products of 128: max |err| = %.3g x largest |product| (2^%.1f), %.3g x sum |products|; mean err %.3g, mean |err| %.3g (x sum |products|)\n",
           fmt == 1 ? "e5m2" : "e4m3", spread, count, max_rel_big, std::log2(max_rel_big > 0 ? max_rel_big : 1e-300), max_rel_abs, mean_err / count, mean_abs / count);
    return 0;
}
