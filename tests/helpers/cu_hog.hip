// Test helper (not part of the product library): a kernel that HOLDS compute units for a given time.
//
// tests/test_hip_modules.py::test_in_launch_exchange_beside_a_persistent_kernel launches it on a second stream while the fused
// cross-attention kernels run their in-launch hand-offs: `blocks` workgroups that each claim `lds_bytes` of LDS (>= 96 KiB: no second
// workgroup of this kernel, and none of the fused kernels' ~130 KiB ones, fits beside it on its CU) and spin on the constant 100 MHz
// wall clock for `milliseconds` - the shape of RCCL's persistent channel workgroups on a rank that exchanges gradients while it computes.
#include <hip/hip_runtime.h>

extern "C" __global__ void cu_hog_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += lds[(threadIdx.x + acc) & 63];
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xffffffffu) sink[0] = acc;          // keeps the loop alive; never true in practice
}

extern "C" int cu_hog_launch(int blocks, unsigned long long lds_bytes, double milliseconds, unsigned* sink, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    const unsigned long long ticks = (unsigned long long)(milliseconds * 1e5);      // wall_clock64: 100 MHz
    cu_hog_kernel<<<dim3(blocks), dim3(64), lds_bytes, stream>>>(ticks, sink);
    return (int)hipGetLastError();
}
