// filler.hip -- developer probe: a low-register (< 32 VGPRs), HBM-bound kernel that can co-reside with three 160-register
// waves per SIMD (512 - 3 x 160 = 32).  Question: does memory-bound work placed BESIDE the VALU-bound k_fast_map run in its
// shadow?  (tools/coresidency_probe.py)   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libfiller.so filler.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void k_fill_copy(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
// the same traffic with some arithmetic per 16 bytes (a stand-in for the resize's ~40 VALU operations per 4 pixels)
__global__ __launch_bounds__(256) void k_fill_work(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n, int ops)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = s[i];
        for (int k = 0; k < ops; ++k) {
            v.x = v.x * 2654435761u + v.y;
            v.y = (v.y >> 3) ^ v.z;
            v.z = v.z + v.w * 40503u;
            v.w = v.w ^ (v.x >> 7);
        }
        d[i] = v;
    }
}
extern "C" int filler_copy(const void *s, void *d, size_t bytes, int blocks, void *stream)
{
    hipLaunchKernelGGL(k_fill_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)s, (uint4 *)d, bytes / 16);
    return (int)hipGetLastError();
}
extern "C" int filler_work(const void *s, void *d, size_t bytes, int blocks, int ops, void *stream)
{
    hipLaunchKernelGGL(k_fill_work, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)s, (uint4 *)d, bytes / 16, ops);
    return (int)hipGetLastError();
}
