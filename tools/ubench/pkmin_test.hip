#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *out)
{
    uint32_t a = 0x00050009u, b = 0x000700FFu, c = 0x00030001u, r0, r1, r2, r3;
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(b));
    asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r3) : "v"(a), "v"(b));
    out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3;
    uint32_t z = 0;
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r0) : "v"(z), "v"(b), "v"(c));
    out[4] = r0;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r1) : "v"(0x44434241u), "v"(0x14131211u), "v"(0x0c070c03u));
    out[5] = r1;
}
int main()
{
    uint32_t *d, h[6];
    hipMalloc(&d, 64);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("min3 %08x (want 00030001)  max3 %08x (want 000700ff)  min %08x (00050009) max %08x (000700ff) min3z %08x (00000000) perm %08x\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
