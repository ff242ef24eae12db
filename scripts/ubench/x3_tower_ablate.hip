// Development: where does tower_x3_kernel's time go?  Compiles x3.hip with one CRA_X3_ABL switch set (each computes wrong results on
// purpose, x3.hip) and times the RISEv2-19 tower (19 blocks, C_op 128 ... 1280, 256 boards) with random weights.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCRA_DEVELOPMENT -DCRA_X3_ABL=<bits> -Icrazyara_amd/csrc/nn scripts/ubench/x3_tower_ablate.hip -o /tmp/x3abl_<bits>
// scripts/run_x3_ablation.sh builds and runs the set.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "x3.hip"        // -I crazyara_amd/csrc/nn
namespace cra { size_t value_head_lds_bytes(const ValueHeadArgs&) { return 0; } }      // (kernels.hip's, which this harness does not link: the head launches are not used here)

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at %s\n", hipGetErrorString(_e), #e); exit(1); } } while (0)

template <typename T> T* upload(const std::vector<T>& h) {
    T* d = nullptr;
    CK(hipMalloc(&d, h.size() * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    using namespace cra;
    const int B = argc > 1 ? atoi(argv[1]) : 256, nblocks = argc > 2 ? atoi(argv[2]) : 19, iters = argc > 3 ? atoi(argv[3]) : 20;
    const int p8 = argc > 4 ? atoi(argv[4]) : 0;          // 1: Precision float16p8's project path (the same buffers: timing only)
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-0.05f, 0.05f);
    std::vector<X3TowerBlock> blocks;
    const int chunk = block_x3_chunk_channels();
    double flops = 0;
    for (int i = 0; i < nblocks; ++i) {
        const int cop = 128 + 64 * i, cop_pad = (cop + chunk - 1) / chunk * chunk;
        std::vector<half_t> w1(size_t(cop_pad) * 256), w3(size_t(256) * cop_pad);
        for (auto& v : w1) v = half_t(u(rng));
        for (auto& v : w3) v = half_t(u(rng));
        std::vector<float> rec(size_t(cop_pad) * 16), b3(256);
        for (auto& v : rec) v = u(rng);
        for (auto& v : b3) v = u(rng);
        X3TowerBlock b{};
        b.w1pk = upload(w1); b.w1pk_lo = upload(w1);
        b.w3pk = upload(w3); b.w3pk_lo = upload(w3);
        b.dwpk = upload(rec);
        b.b3 = upload(b3);
        b.cop_pad = cop_pad;
        b.w1_inv = 1.f;
        b.w3_scale = 1.f;
        b.w3_inv = 1.f;
        blocks.push_back(b);
        flops += 2.0 * 64 * cop * (2.0 * 256 + 9) * B;
    }
    std::vector<float> x(size_t(B) * 64 * 256);
    for (auto& v : x) v = u(rng) * 10;
    X3TowerArgs a{};
    a.x = upload(x);
    a.y = upload(x);
    a.blocks = upload(blocks);
    a.nblocks = nblocks;
    a.batch = B;
    a.p8 = p8;
    {   // launch_tower_x3 reads X3TowerArgs::symmetric (RiseNet::build sets it from CRA_X3_TOWER when a net is made); here it comes from the same variable
        const char* tw = getenv("CRA_X3_TOWER");
        a.symmetric = tw && std::string(tw) == "symmetric" ? 1 : 0;
    }
    init_x3_kernel_attributes();
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 3; ++i) launch_tower_x3(a, s);
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_tower_x3(a, s);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("CRA_X3_ABL=%d  B=%d blocks=%d chunk=%d: %.4f ms per tower launch  (%.1f algorithmic TFLOP/s)\n", CRA_X3_ABL, B, nblocks, chunk, ms,
           flops / (ms * 1e-3) / 1e12);
#ifdef CRA_X3_TRACE
    // timeline of block CRA_X3_TRACE in workgroups 0 and 131 (last launch): shader-clock cycles since the workgroup's first stamp.
    // EXPAND waves 0-3: 0 interval start, 1 half of the expand MFMAs, 2 expand done, 3 depthwise done, 4 behind the chunk barrier, 5 block end
    // PROJECT waves 4-7: 8 interval start, 9 half of the project MFMAs, 10 project done, 11 behind the chunk barrier, 12 epilogue done, 13 block end
    static unsigned long long tr[2][8][128][2];
    CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(cra::x3_trace), sizeof(tr)));
    for (int g = 0; g < 2; ++g) {
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 8; ++w)
            if (tr[g][w][0][0] && tr[g][w][0][0] < t0) t0 = tr[g][w][0][0];
        printf("workgroup %d, block %d (C_op %d = %d chunks)\n", g ? 131 : 0, CRA_X3_TRACE, 128 + 64 * CRA_X3_TRACE, (128 + 64 * CRA_X3_TRACE + chunk - 1) / chunk);
        for (int w = 0; w < 8; ++w) {
            printf("  wave %d:", w);
            int last_iv = -99;
            for (int i = 0; i < 128 && tr[g][w][i][0]; ++i) {
                const int iv = int(tr[g][w][i][1] >> 4) - 1, id = int(tr[g][w][i][1] & 15);
                if (iv != last_iv) { printf("\n    interval %2d:", iv); last_iv = iv; }
                printf("  [%d] %6llu", id, tr[g][w][i][0] - t0);
            }
            printf("\n");
        }
    }
#endif
    return 0;
}
