// does the VALU overlap v_mfma_i32_32x32x32_i8?  per loop: 2 independent MFMA chains + NV independent v_max/v_min ops
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NV, bool MF>
__global__ __launch_bounds__(256) void k(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, 6, 7, (int)threadIdx.x};
    v16i c0 = {}, c1 = {};
    int x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * (j + 3);
    for (int i = 0; i < iters; ++i) {
        if (MF) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j & 7] = max(x[j & 7], min(x[(j + 1) & 7], i + j));  // 2 VALU each
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, bool MF> void run(int *d, int wpg)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, MF>), dim3(256 * wpg), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%s, %2d min/max pairs per 2 MFMA, %d wave(s)/SIMD: %.1f ns per loop iteration per wave-slot\n", MF ? "MFMA+VALU" : "VALU only", NV, wpg, ms * 1e6 / ((double)iters * wpg));
    }
}
int main()
{
    int *d; hipMalloc(&d, 4096 * 256 * 4);
    run<0, true>(d, 1); run<7, false>(d, 1); run<7, true>(d, 1); run<14, false>(d, 1); run<14, true>(d, 1);
    run<7, true>(d, 2); run<14, true>(d, 2);
    return 0;
}
