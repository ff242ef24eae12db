// Whole forward of a board in ONE launch for gfx950: stem -> residual tower -> policy + value head, one workgroup per board.
//
// The three kernels (stem.hip, tower.hip, head.hip) already share their geometry: one 512-thread workgroup per board, and the
// board's 64 x 256 f16 activation tile sits at offset 0 of the dynamic LDS segment with a 528-byte row pitch in all of them (the
// stem's output staging tile = the tower's residual stream = the head's input tile).  Run back to back inside one kernel, the tile
// never leaves the CU between them: no 32 KB store + reload per board at either seam, no two launch boundaries (the stem alone was
// a 9-12 us launch for 0.3 us of MFMA work per board).  Their bodies are compiled into this translation unit as device functions.
#define CRA_FORWARD_TU 1
#include "stem.hip"
#include "tower.hip"
#include "head.hip"

namespace cra {

namespace {
constexpr int FW_STEM_LDS = ST_OUT_BYTES + 65 * (96 + 8) * 2;        // largest stem: cin_pad 96
constexpr int fw_max(int a, int b) { return a > b ? a : b; }
constexpr int FW_DYN_LDS = fw_max(TW_DYN_LDS_BYTES, fw_max(HD_LDS_BYTES, FW_STEM_LDS));
constexpr int FW_DYN_LDS_F8 = fw_max(TW_DYN_LDS_BYTES_F8, fw_max(HD_LDS_BYTES, FW_STEM_LDS));
static_assert(ST_OROW == TW_XROW && TW_XROW == HD_ROW, "the three kernels must agree on the pitch of the board tile");
}  // namespace

template <int NKS, int F8 = 0>      // F8: 0 Precision float16, 1 fp8 (e4m3), 2 int8 (tower.hip: Q)
__global__ __launch_bounds__(512) void forward_kernel(const StemArgs sa, const TowerArgs ta, const HeadArgs ha) {
#ifdef CRA_DEV_SEAMS                     // development: bit 0 / bit 1 = hand the tile over through global memory at the first / second seam
    stem_body<NKS>(sa, (CRA_DEV_SEAMS & 1) != 0);
    if (CRA_DEV_SEAMS & 1) { __threadfence(); __syncthreads(); }
    tower_body<F8>(ta, (CRA_DEV_SEAMS & 1) == 0, (CRA_DEV_SEAMS & 2) != 0);
    if (CRA_DEV_SEAMS & 2) { __threadfence(); __syncthreads(); }
    if (CRA_DEV_SEAMS & 4) __syncthreads();
    head_body(ha, (CRA_DEV_SEAMS & 2) == 0);
#else
    stem_body<NKS>(sa, false);           // ends in a workgroup barrier: the tile is complete
    tower_body<F8>(ta, true, false);     // every role ends in the barrier after the last block's epilogue
    head_body(ha, true);
#endif
}

void init_forward_kernel_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<6, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel<6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, FW_DYN_LDS_F8);
}

void launch_forward(const StemArgs& sa, const TowerArgs& ta, const HeadArgs& ha, hipStream_t s) {
    if (ta.fp8 == 2) {
        switch (sa.cin_pad / 16) {
            case 3: hipLaunchKernelGGL((forward_kernel<3, 2>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
            case 4: hipLaunchKernelGGL((forward_kernel<4, 2>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
            case 5: hipLaunchKernelGGL((forward_kernel<5, 2>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
            default: hipLaunchKernelGGL((forward_kernel<6, 2>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
        }
        return;
    }
    if (ta.fp8) {
        switch (sa.cin_pad / 16) {
            case 3: hipLaunchKernelGGL((forward_kernel<3, 1>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
            case 4: hipLaunchKernelGGL((forward_kernel<4, 1>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
            case 5: hipLaunchKernelGGL((forward_kernel<5, 1>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
            default: hipLaunchKernelGGL((forward_kernel<6, 1>), dim3(sa.batch), dim3(512), FW_DYN_LDS_F8, s, sa, ta, ha); break;
        }
        return;
    }
    switch (sa.cin_pad / 16) {
        case 3: hipLaunchKernelGGL(forward_kernel<3>, dim3(sa.batch), dim3(512), FW_DYN_LDS, s, sa, ta, ha); break;
        case 4: hipLaunchKernelGGL(forward_kernel<4>, dim3(sa.batch), dim3(512), FW_DYN_LDS, s, sa, ta, ha); break;
        case 5: hipLaunchKernelGGL(forward_kernel<5>, dim3(sa.batch), dim3(512), FW_DYN_LDS, s, sa, ta, ha); break;
        default: hipLaunchKernelGGL(forward_kernel<6>, dim3(sa.batch), dim3(512), FW_DYN_LDS, s, sa, ta, ha); break;
    }
}

}  // namespace cra
