// timing harness for k_match_bf variants: links the product sources compiled with -DBM_* experiment flags
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "orbfe.h"
int main()
{
    const int B = 256, cap = 1088, n = 1004;
    std::vector<uint8_t> h((size_t)B * cap * 32);
    for (auto &v : h) v = (uint8_t)rand();
    uint8_t *d_desc; orbfe_keypoint *d_kps; int32_t *d_n, *d_q, *d_t, *d_m, *d_nm;
    hipMalloc(&d_desc, h.size()); hipMemcpy(d_desc, h.data(), h.size(), hipMemcpyHostToDevice);
    hipMalloc(&d_kps, (size_t)B * cap * sizeof(orbfe_keypoint)); hipMemset(d_kps, 0, (size_t)B * cap * sizeof(orbfe_keypoint));
    std::vector<int32_t> hn(B, n), hq(B), ht(B);
    for (int i = 0; i < B; ++i) { hq[i] = i; ht[i] = (i + B - 1) % B; }
    hipMalloc(&d_n, B * 4); hipMalloc(&d_q, B * 4); hipMalloc(&d_t, B * 4); hipMalloc(&d_m, (size_t)B * cap * 4); hipMalloc(&d_nm, B * 4);
    hipMemcpy(d_n, hn.data(), B * 4, hipMemcpyHostToDevice); hipMemcpy(d_q, hq.data(), B * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_t, ht.data(), B * 4, hipMemcpyHostToDevice);
    orbfe_matcher *m; if (orbfe_matcher_create(0, &m) != ORBFE_OK) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 200; ++i) orbfe_match_bf_frames_device(m, d_kps, d_desc, d_n, cap, d_q, d_t, B, 0.9f, 100, 0, d_m, d_nm, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < 500; ++i) orbfe_match_bf_frames_device(m, d_kps, d_desc, d_n, cap, d_q, d_t, B, 0.9f, 100, 0, d_m, d_nm, nullptr);
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("us_per_call %.2f\n", ms * 1000 / 500);
    }
    return 0;
}
