// Test support: fills the LDS of every CU with a byte pattern (bytes [lo, hi) of the 160 KiB get `pattern`, the rest `base`), so that a
// kernel launched afterwards that reads LDS it never wrote shows it (tests/test_fp8.py; scripts/lds_read_before_write.py bisects
// [lo, hi) down to the bytes that matter).  Built on demand by tests/conftest.py: hipcc -shared, one C entry point.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void poison(unsigned pattern, unsigned base, int lo, int hi, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = (i * 4 >= lo && i * 4 < hi) ? pattern : base;
    __syncthreads();
    if (lds[(threadIdx.x * 7 + blockIdx.x) % (160 * 1024 / 4)] == 0x12345678u) sink[0] = 1;
}
extern "C" int poison_lds(unsigned pattern, unsigned base, int lo, int hi) {
    static unsigned* sink = nullptr;
    if (!sink) hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(poison, dim3(2048), dim3(1024), 160 * 1024, 0, pattern, base, lo, hi, sink);
    return int(hipDeviceSynchronize());
}
