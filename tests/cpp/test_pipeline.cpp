// test_pipeline.cpp -- a C++ host driving the sequence pipeline of the C-ABI (include/orbfe.h orbfe_pipeline_*), the shape
// of the reference's frame loop (perfect/Examples/RGB-D/rgbd_tum.cc:77-119: every frame through the extractor, then a match
// against the previous frame).  No Python, no torch: HIP runtime + liborbfe.so only.
//   usage: test_pipeline in.raw W H nframes nfeatures sub_batch npipes calls out.bin [no_join [host]]
// host != 0: the calls go through orbfe_pipeline_extract_match -- plain host buffers in and out (malloc'ed, pageable), the
// copies overlapped inside the library -- instead of the device entry point.
// The sequence of `nframes` frames is pushed through in `calls` consecutive calls (ORBFE_PIPE_CONTINUE from the second on),
// every call re-using the SAME device output blocks (the pipeline protects them); results are copied to the host after each
// call.  out.bin: per frame  int32 n | n x 28 B keypoints | n x 32 B descriptors | int32 nmatches | n x int32 match row.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "orbfe.h"

#define CHECK_HIP(x)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);     \
            return 10;                                                                      \
        }                                                                                   \
    } while (0)
#define CHECK_ORB(x)                                                                                        \
    do {                                                                                                    \
        orbfe_status s_ = (x);                                                                              \
        if (s_ != ORBFE_OK) {                                                                               \
            fprintf(stderr, "%s: %s (%s) line %d\n", #x, orbfe_strerror(s_), orbfe_last_error(), __LINE__); \
            return 11;                                                                                      \
        }                                                                                                   \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 10) {
        fprintf(stderr, "usage: %s in.raw W H nframes nfeatures sub_batch npipes calls out.bin [no_join]\n", argv[0]);
        return 2;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]), N = atoi(argv[4]), nf = atoi(argv[5]), F = atoi(argv[6]), P = atoi(argv[7]);
    const int calls = atoi(argv[8]);
    const bool no_join = argc > 10 && atoi(argv[10]) != 0;
    const bool host_mode = argc > 11 && atoi(argv[11]) != 0;
    if (N < 1 || calls < 1 || calls > N) return 2;
    std::vector<uint8_t> frames((size_t)W * H * N);
    FILE *fi = fopen(argv[1], "rb");
    if (!fi || fread(frames.data(), 1, frames.size(), fi) != frames.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 3; }
    fclose(fi);

    orbfe_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.nfeatures = nf; prm.scale_factor = 1.2f; prm.nlevels = 8; prm.ini_th_fast = 20; prm.min_th_fast = 7;
    prm.max_width = W; prm.max_height = H; prm.max_batch = F; prm.device = 0; prm.blur_rounding = 0;
    orbfe_pipeline *pl = NULL;
    CHECK_ORB(orbfe_pipeline_create(&prm, P, &pl));
    const int cap = orbfe_pipeline_capacity(pl);
    const int per_call = (N + calls - 1) / calls;   // frames of a call (the last one may be shorter)

    uint8_t *d_gray = NULL, *d_desc = NULL;
    orbfe_keypoint *d_kps = NULL;
    int32_t *d_n = NULL, *d_match = NULL, *d_nm = NULL;
    CHECK_HIP(hipMalloc((void **)&d_gray, (size_t)W * H * N));
    CHECK_HIP(hipMemcpy(d_gray, frames.data(), frames.size(), hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void **)&d_kps, (size_t)per_call * cap * sizeof(orbfe_keypoint)));
    CHECK_HIP(hipMalloc((void **)&d_desc, (size_t)per_call * cap * 32));
    CHECK_HIP(hipMalloc((void **)&d_n, (size_t)per_call * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_match, (size_t)per_call * cap * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_nm, (size_t)per_call * sizeof(int32_t)));
    hipStream_t st;
    CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    std::vector<orbfe_keypoint> kps((size_t)per_call * cap);
    std::vector<uint8_t> desc((size_t)per_call * cap * 32);
    std::vector<int32_t> n((size_t)per_call), nm((size_t)per_call), match((size_t)per_call * cap);
    FILE *fo = fopen(argv[9], "wb");
    if (!fo) return 4;
    for (int c = 0, lo = 0; lo < N; ++c, lo += per_call) {
        const int nfr = per_call < N - lo ? per_call : N - lo;
        const int flags = (c > 0 ? ORBFE_PIPE_CONTINUE : 0) | (no_join ? ORBFE_PIPE_NO_JOIN : 0);
        if (host_mode) {
            std::vector<const uint8_t *> ptrs((size_t)nfr);
            for (int f = 0; f < nfr; ++f) ptrs[(size_t)f] = frames.data() + (size_t)(lo + f) * W * H;
            CHECK_ORB(orbfe_pipeline_extract_match(pl, ptrs.data(), nfr, W, H, W, kps.data(), desc.data(), cap, n.data(), match.data(),
                                                   nm.data(), 0.9f, ORBFE_TH_HIGH, 1, c > 0 ? ORBFE_PIPE_CONTINUE : 0));
        } else {
        CHECK_ORB(orbfe_pipeline_extract_match_device(pl, d_gray + (size_t)lo * W * H, nfr, W, H, W, (size_t)W * H, d_kps, d_desc, cap, d_n,
                                                      d_match, d_nm, 0.9f, ORBFE_TH_HIGH, 1, flags, (void *)st));
        if (no_join) CHECK_ORB(orbfe_pipeline_join(pl, (void *)st));   // the copies below run on `st`
        CHECK_HIP(hipMemcpyAsync(n.data(), d_n, (size_t)nfr * 4, hipMemcpyDeviceToHost, st));
        CHECK_HIP(hipMemcpyAsync(nm.data(), d_nm, (size_t)nfr * 4, hipMemcpyDeviceToHost, st));
        CHECK_HIP(hipMemcpyAsync(kps.data(), d_kps, (size_t)nfr * cap * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost, st));
        CHECK_HIP(hipMemcpyAsync(desc.data(), d_desc, (size_t)nfr * cap * 32, hipMemcpyDeviceToHost, st));
        CHECK_HIP(hipMemcpyAsync(match.data(), d_match, (size_t)nfr * cap * 4, hipMemcpyDeviceToHost, st));
        CHECK_HIP(hipStreamSynchronize(st));
        }
        for (int f = 0; f < nfr; ++f) {
            const int k = n[(size_t)f];
            if (k < 0 || k > cap) { fprintf(stderr, "frame %d: count %d outside [0, %d]\n", lo + f, k, cap); return 5; }
            fwrite(&k, 4, 1, fo);
            fwrite(&kps[(size_t)f * cap], sizeof(orbfe_keypoint), (size_t)k, fo);
            fwrite(&desc[(size_t)f * cap * 32], 32, (size_t)k, fo);
            fwrite(&nm[(size_t)f], 4, 1, fo);
            fwrite(&match[(size_t)f * cap], 4, (size_t)k, fo);
        }
    }
    fclose(fo);
    int32_t ovf = 0;
    CHECK_ORB(orbfe_pipeline_get_overflow(pl, &ovf));
    if (ovf) { fprintf(stderr, "device-side capacity overflow %d\n", ovf); return 6; }
    orbfe_pipeline_destroy(pl);
    hipFree(d_gray); hipFree(d_kps); hipFree(d_desc); hipFree(d_n); hipFree(d_match); hipFree(d_nm);
    hipStreamDestroy(st);
    printf("ok %d frames, %d calls, cap %d\n", N, calls, cap);
    return 0;
}
