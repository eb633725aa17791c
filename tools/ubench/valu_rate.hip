// micro-benchmark: issue rate of candidate VALU instructions on gfx950 (wave64), 8 independent chains per lane
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
#define BODY8(INS) \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define KERNEL(NAME, ASMSTR)                                                                         \
    __global__ void NAME(uint32_t *out, uint32_t seed)                                               \
    {                                                                                                \
        uint32_t a[8], b = seed + threadIdx.x, c = seed * 3 + 1;                                     \
        for (int i = 0; i < 8; ++i) a[i] = seed + i * 17 + threadIdx.x;                              \
        for (int it = 0; it < ITERS; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
                asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c));                                  \
        }                                                                                            \
        uint32_t r = 0;                                                                              \
        for (int i = 0; i < 8; ++i) r ^= a[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                              \
    }
KERNEL(k_min3_u32, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_min3_i32, "v_min3_i32 %0, %0, %1, %2")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
KERNEL(k_pk_min_f16, "v_pk_min_f16 %0, %0, %1")
KERNEL(k_pk_minimum3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
KERNEL(k_minimum3_f32, "v_minimum3_f32 %0, %0, %1, %2")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_pk_fma_f32_dummy, "v_add_u32 %0, %0, %1")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, %1, %2")
KERNEL(k_cvt_ubyte1, "v_cvt_f32_ubyte1 %0, %1")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_max3_u32_sdwa_like, "v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
KERNEL(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, %1, %2")

template <typename K> void run(const char *name, K kern, uint32_t *d)
{
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, threads>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, threads>>>(d, r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = 5.0 * blocks * (threads / 64) * (double)ITERS * 8;  // wave-instructions
    double per_cu_per_clk = instr / (ms * 1e-3) / 256 / 2.4e9;         // wave-instr per CU per clock (2.4 GHz nominal)
    printf("%-22s %8.3f ms  %6.3f wave-instr/clk/CU  -> %5.2f clk per wave-instr per SIMD\n", name, ms, per_cu_per_clk, 4.0 / per_cu_per_clk);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(k) run(#k, k, d)
    R(k_min3_u32); R(k_min3_i32); R(k_min_u32); R(k_min3_f32); R(k_min_f32); R(k_pk_min_u16); R(k_pk_min_i16); R(k_pk_min_f16);
    R(k_pk_minimum3_f16); R(k_minimum3_f32); R(k_fma_f32); R(k_pk_fma_f32_dummy); R(k_mad_u32_u24); R(k_bfe_u32); R(k_cvt_ubyte1);
    R(k_pk_mad_u16); R(k_pk_add_u16); R(k_perm_b32); R(k_sad_u8); R(k_max3_u32_sdwa_like); R(k_and_b32); R(k_lshl_or);
    return 0;
}
