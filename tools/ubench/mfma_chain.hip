// latency vs throughput of v_mfma_i32_32x32x32_i8: N independent accumulator chains per wave (1 wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ __launch_bounds__(256) void k(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, 6, 7, (int)threadIdx.x};
    v16i c[NCH];
    for (int j = 0; j < NCH; ++j) c[j] = v16i{};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < NCH; ++j) c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[j], 0, 0, 0);
    int s = 0;
    for (int j = 0; j < NCH; ++j)
        for (int i = 0; i < 16; ++i) s += c[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCH> void run(int *d, int wpg)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NCH>, dim3(256 * wpg), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("chains %d, %d wave(s)/SIMD: %.2f ns per MFMA per SIMD\n", NCH, wpg, ms * 1e6 / ((double)NCH * iters * wpg));
    }
}
int main()
{
    int *d; hipMalloc(&d, 4096 * 256 * 4);
    run<1>(d, 1); run<2>(d, 1); run<4>(d, 1); run<1>(d, 2); run<2>(d, 2);
    return 0;
}
