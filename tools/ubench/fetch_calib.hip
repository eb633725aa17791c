// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the extractor kernels use
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").  Three kernels stream a 1 GiB
// buffer (4x the Infinity Cache) exactly once:  k_dword  4 B per lane,  k_win12  the FAST/blur window (every lane loads
// the 12 bytes [4l-4, 4l+8) -> each byte requested 3x, unique bytes = buffer),  k_x4  16 B per lane; k_store writes it.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_dword(const uint32_t *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_win12(const uint8_t *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x + 1; i + 2 < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t *q = (const uint32_t *)(p + 4 * i - 4);
        acc ^= q[0] ^ q[1] ^ q[2];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_x4(const uint4 *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_store(uint32_t *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main()
{
    const size_t bytes = 1ull << 30;
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, bytes + 64); hipMalloc(&o, 64);
    hipMemset(d, 1, bytes + 64);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_dword, dim3(4096), dim3(256), 0, 0, (const uint32_t *)d, bytes / 4, o);
        hipLaunchKernelGGL(k_win12, dim3(4096), dim3(256), 0, 0, (const uint8_t *)d, bytes / 4, o);
        hipLaunchKernelGGL(k_x4, dim3(4096), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, o);
        hipLaunchKernelGGL(k_store, dim3(4096), dim3(256), 0, 0, (uint32_t *)d, bytes / 4);
    }
    hipDeviceSynchronize();
    printf("buffer bytes %zu\n", bytes);
    return 0;
}
