// issue rate of the gfx950 i8 / fp8 MFMA shapes: one wave per SIMD, 4 independent accumulator chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef long v2l __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, 6, 7, (int)threadIdx.x};
    v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
    v4i d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        } else if (KIND == 1) {
            d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d3, 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    int *d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ops = (kind == 0 ? 65536.0 : 32768.0) * 4 * iters * 1024;  // per-wave ops x waves (256 WG x 4)
            printf("kind %d: %.3f ms, %.1f TOPS, ns per MFMA per SIMD %.2f\n", kind, ms, ops / ms / 1e9, ms * 1e6 / (4.0 * iters));
        }
    return 0;
}
