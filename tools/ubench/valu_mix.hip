// micro-benchmark: issue cost (shader clocks per wave64 instruction per SIMD) of the opcodes k_fast_map's row loop is
// made of (tools/valu_count.py prints the histogram), 8 independent chains per lane, 8 waves per SIMD.  The shader clock
// is measured (s_memtime ticks per wall second), not assumed.  Output: one line per opcode and a JSON summary that
// bench.py's VALU model is pasted from.  build: hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 4096
#define KERNEL(NAME, ASMSTR)                                                                         \
    __global__ void NAME(uint32_t *out, uint32_t seed, unsigned long long *clk)                      \
    {                                                                                                \
        uint32_t a[8], b = seed + threadIdx.x, c = seed * 3 + 1;                                     \
        for (int i = 0; i < 8; ++i) a[i] = seed + i * 17 + threadIdx.x;                              \
        const unsigned long long t0 = clock64();                                                     \
        for (int it = 0; it < ITERS; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
                asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c));                                  \
        }                                                                                            \
        const unsigned long long t1 = clock64();                                                     \
        uint32_t r = 0;                                                                              \
        for (int i = 0; i < 8; ++i) r ^= a[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;                                   \
    }
KERNEL(k_pk_maximum3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
KERNEL(k_pk_minimum3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
KERNEL(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_alignbit_b32, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_cmp_ne_u32, "v_cmp_ne_u32 vcc, %0, %1")
KERNEL(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")

static double g_mhz = 0;
template <typename K> void run(const char *name, K kern, uint32_t *d, unsigned long long *clk, int count_in_loop)
{
    const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, threads>>>(d, 1, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, threads>>>(d, r, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 5.0 * blocks * (threads / 64) * (double)ITERS * 8 / 1024.0;
    const double clk_per_instr = ms * 1e-3 * g_mhz * 1e6 / instr_per_simd;
    printf("{\"op\": \"%s\", \"ms\": %.3f, \"clk_per_wave_instr_per_simd\": %.3f, \"per_row_step\": %d},\n", name, ms, clk_per_instr, count_in_loop);
}

__global__ void k_clock(unsigned long long *out)
{
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    unsigned long long t1, w1;
    do { t1 = clock64(); w1 = wall_clock64(); } while (w1 - w0 < 20000000ull);  // 0.2 s at the 100 MHz wall clock
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}

int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *clk; hipMallocManaged(&clk, 64);
    // shader clock under load: one busy kernel on every CU next to the timing wave
    k_clock<<<1, 64>>>(clk);
    hipDeviceSynchronize();
    g_mhz = (double)clk[0] / ((double)clk[1] / 100.0);  // wall_clock64 ticks at 100 MHz
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("{\"shader_clock_MHz_measured_idle_chip\": %.1f, \"clock_rate_attr_MHz\": %.1f,\n \"ops\": [\n", g_mhz, khz / 1000.0);
#define R(k, n) run(#k, k, d, clk, n)
    R(k_pk_maximum3_f16, 42); R(k_pk_minimum3_f16, 38); R(k_pk_max_u16, 38); R(k_pk_min_u16, 34); R(k_pk_sub_u16, 8);
    R(k_perm_b32, 32); R(k_alignbit_b32, 3); R(k_cndmask_b32, 18); R(k_add_u32, 16); R(k_add3_u32, 9); R(k_and_b32, 6);
    R(k_mov_b32, 10); R(k_lshlrev_b32, 5); R(k_mbcnt_lo, 8); R(k_cmp_ne_u32, 15); R(k_dot4_u32_u8, 0); R(k_mad_u32_u24, 2);
    printf("]}\n");
    return 0;
}
