// Probe for the FP8 (e4m3) tower: what v_mfma_f32_32x32x64_f8f6f4 and v_cvt_scalef32_pk_fp8_f16 do on gfx950.
//   1. operand layout: lane l holds row / column l % 32 and 32 bytes of K; is "lane half lh, byte t <-> k = lh * 32 + t" on BOTH operands a
//      consistent labelling (the product then equals the plain matrix product whatever the hardware calls its k)?  D layout = 32x32x16's?
//   2. cycles per MFMA (one wave per SIMD, independent accumulators)
//   3. the conversion: rounding, values beyond +-448, negatives, op_sel halves
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/fp8_probe.hip -o scripts/ubench/fp8_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

static float e4m3_to_float(uint8_t b) {               // OCP e4m3fn: bias 7, no infinities, 0x7f / 0xff = NaN
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = std::ldexp(float(m), -9);
    else if (e == 15 && m == 7) v = NAN;
    else v = std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}

__global__ void gemm(const uint8_t* A, const uint8_t* B, float* D) {   // A [32][64] row-major bytes, B [64][32] as [col][k] bytes, D [32][32]
    const int l = threadIdx.x, r = l & 31, lh = l >> 5;
    v8i a, b;
    for (int t = 0; t < 8; ++t) {
        a[t] = reinterpret_cast<const int*>(A + r * 64 + lh * 32)[t];
        b[t] = reinterpret_cast<const int*>(B + r * 64 + lh * 32)[t];
    }
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[((v % 4) + 8 * (v / 4) + 4 * lh) * 32 + r] = c[v];
}

__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int n) {
    v8i a, b;
    for (int t = 0; t < 8; ++t) { a[t] = 0x38383838 + threadIdx.x; b[t] = 0x30303030 + t; }
    v16f c[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[j], 0, 0, 0, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int v = 0; v < 16; ++v) s += c[j][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void cvt(const _Float16* in, uint8_t* out, int n, float scale, int ovfl) {
    const int i = threadIdx.x;
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);    // MODE.FP16_OVFL = 1: conversions clamp instead of overflowing
    if (2 * i + 1 >= n + 1) return;
    h2 x = {in[2 * i], in[2 * i + 1]};
    s2 o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, x, scale, false);      // low 16 bits
    s2 o2 = {0x1111, 0x2222};
    o2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o2, x, scale, true);     // high 16 bits, low half kept?
    out[4 * i + 0] = uint8_t(o[0] & 0xff);
    out[4 * i + 1] = uint8_t((o[0] >> 8) & 0xff);
    out[4 * i + 2] = uint8_t(o2[1] & 0xff);
    out[4 * i + 3] = uint8_t(o2[0] & 0xff);       // 0x11 if the low half is preserved
}

int main() {
    // ---- 1. layout ----
    std::vector<uint8_t> A(32 * 64), B(32 * 64);
    srand(1);
    for (auto& v : A) { v = uint8_t(rand() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; }
    for (auto& v : B) { v = uint8_t(rand() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    gemm<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(32 * 32);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += double(e4m3_to_float(A[i * 64 + k])) * double(e4m3_to_float(B[j * 64 + k]));
            worst = std::fmax(worst, std::fabs(s - D[i * 32 + j]));
            ref_max = std::fmax(ref_max, std::fabs(s));
        }
    printf("layout: max |D - A.B^T| = %.3e (max |ref| %.3e) -> %s\n", worst, ref_max, worst <= 1e-4 * ref_max ? "k = lh*32 + t on both operands is consistent" : "MISMATCH");
    // ---- 2. rate ----
    float* o; unsigned long long* c;
    hipMalloc(&o, 256 * 256 * 4); hipMalloc(&c, 8);
    const int n = 20000;
    rate<<<256, 256>>>(o, c, 100);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    rate<<<256, 256>>>(o, c, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double flop = 256.0 * 4 * n * 4 * 2.0 * 32 * 32 * 64;
    printf("rate: %.1f cycles per v_mfma_f32_32x32x64_f8f6f4 (e4m3), %.2f PFLOP/s over the chip, clock %.2f GHz\n", double(cy) / (4.0 * n), flop / (ms * 1e-3) / 1e15,
           double(cy) / (ms * 1e-3) / 1e9);
    // ---- 3. conversion ----
    const float vals[] = {0.f, 1.f, 1.0625f, 1.125f, 1.1875f, 0.1f, 447.f, 448.f, 449.f, 480.f, 500.f, 1000.f, 60000.f, -3.3f, -500.f, 0.001f, 0.0009765625f, 0.002f, 0.0146f, 1.5e-4f};
    const int nv = sizeof(vals) / sizeof(vals[0]);
    std::vector<_Float16> hv(nv);
    for (int i = 0; i < nv; ++i) hv[i] = _Float16(vals[i]);
    _Float16* dv; uint8_t* dq;
    hipMalloc(&dv, nv * 2); hipMalloc(&dq, nv * 2);
    hipMemcpy(dv, hv.data(), nv * 2, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, -1.0f}) {
        const int ovfl = scale < 0;
        if (ovfl) scale = 1.0f;
        cvt<<<1, nv / 2>>>(dv, dq, nv, scale, ovfl);
        if (ovfl) printf("with MODE.FP16_OVFL = 1:\n");
        std::vector<uint8_t> q(nv * 2);
        hipMemcpy(q.data(), dq, nv * 2, hipMemcpyDeviceToHost);
        printf("cvt_scalef32_pk_fp8_f16, scale %.1f:\n", scale);
        for (int i = 0; i < nv / 2; ++i)
            printf("   (%g, %g) -> lo-half (0x%02x = %g, 0x%02x = %g)   hi-half byte0 0x%02x = %g, low half kept: %s\n", float(hv[2 * i]), float(hv[2 * i + 1]), q[4 * i], e4m3_to_float(q[4 * i]),
                   q[4 * i + 1], e4m3_to_float(q[4 * i + 1]), q[4 * i + 2], e4m3_to_float(q[4 * i + 2]), q[4 * i + 3] == 0x11 ? "yes" : "no");
    }
    return 0;
}
