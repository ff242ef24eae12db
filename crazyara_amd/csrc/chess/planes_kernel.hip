// GPU input-plane builder: one workgroup per board expands a 192-byte descriptor into the C x 64 float planes of the
// batch tensor with fully coalesced stores (thread = consecutive (channel, square) element).  HBM-bound byte work:
// algorithmic bytes per board = 192 read + C*256 written.
#include <hip/hip_runtime.h>

#include "planes_host.h"

namespace cra {

__global__ __launch_bounds__(256) void planes_from_desc_kernel(const BoardDesc* __restrict__ desc, int layout, int normalize,
                                                               float* __restrict__ planes) {
    __shared__ BoardDesc sd;
    const int b = blockIdx.x;
    if (threadIdx.x < sizeof(BoardDesc) / 8)
        reinterpret_cast<uint64_t*>(&sd)[threadIdx.x] = reinterpret_cast<const uint64_t*>(desc + b)[threadIdx.x];
    __syncthreads();
    const int n = layout_channels(layout) * 64;
    float* out = planes + size_t(b) * n;
    for (int i = threadIdx.x; i < n; i += 256) out[i] = plane_value(sd, layout, normalize != 0, i >> 6, i & 63);
}

void launch_planes_from_desc(const BoardDesc* d_desc, int n, int layout, int normalize, float* d_planes, void* stream) {
    hipLaunchKernelGGL(planes_from_desc_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), d_desc, layout,
                       normalize, d_planes);
}

}  // namespace cra
