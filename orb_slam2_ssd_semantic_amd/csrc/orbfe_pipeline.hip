// orbfe_pipeline.hip -- the throughput pipeline of the ORB front-end as a product feature (include/orbfe.h "Pipeline").
//
// What it stands for in the reference: the frame loop of perfect/Examples/RGB-D/rgbd_tum.cc:77-119 -- one frame after the
// other through ORBextractor::operator() (Frame constructor) and a match against the previous frame.  Here a whole resident
// SEQUENCE goes through in one call: it is cut into sub-batches of `max_batch` frames, sub-batch j runs on pipe j mod P -- a
// pipe = one extractor handle + one matcher handle + one stream, so that the VALU-bound FAST pass of one sub-batch shares the
// chip with the HBM / LDS-bound stages of its neighbours -- and frame k is matched against frame k - 1 ACROSS sub-batch
// boundaries and across calls (the last frame of a call is carried over), i.e. a real sequence.
//
// Host code only: every kernel is launched through the extractor / matcher entry points of this library.
#include <new>
#include <vector>

#include "orbfe_common.h"

// csrc/orbfe_api.hip / orbfe_match.hip: handles that live on a stream of the pipeline instead of creating their own
orbfe_status orbfe_internal_create_on_stream(const orbfe_params *p, void *st, void *side, orbfe_handle **out);
orbfe_status orbfe_internal_matcher_create_on_stream(int32_t device, void *st, orbfe_matcher **out);

#ifndef ORBFE_PIPE_SIDE_STREAMS
#define ORBFE_PIPE_SIDE_STREAMS 1   // side streams of a pipeline, shared by its pipes (A/B -DORBFE_PIPE_SIDE_STREAMS=n, 12 pipes: 1: 314-315 k frames/s resident and 0.91 of the PCIe link through the host entry point; 2: 312-313 k / 0.87; 4: 299-301 k; one per pipe: 310 k / 0.90)
#endif

struct PipeGuard {
    int prev = -1, dev = -1;
    explicit PipeGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~PipeGuard()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

struct orbfe_pipeline {
    orbfe_params prm;
    int device = 0;
    int P = 1;     // pipes
    int F = 1;     // frames per sub-batch (prm.max_batch)
    int cap = 0;   // keypoint slots per frame
    std::vector<orbfe_handle *> ext;
    std::vector<orbfe_matcher *> mat;
    std::vector<hipStream_t> st;
    std::vector<hipStream_t> side;     // side streams (the extractors' blur), shared: pipe i uses side[i % side.size()]
    std::vector<hipEvent_t> ev_end;    // per pipe: behind the pipe's last launch of the most recent call
    std::vector<hipEvent_t> ev_ext;    // per sub-batch index: extraction finished (most recent call)
    std::vector<hipEvent_t> ev_match;  // per sub-batch index: matcher finished (most recent call)
    std::vector<char> ev_match_valid;
    std::vector<char> ev_ext_valid;
    // the output slices the events of index j stand for (keypoint block, descriptor block, counts): a later call that lays its
    // sub-batches over other addresses (shifted base pointer, another call size) is ordered by ADDRESS, not by index
    struct Slice { const char *k = nullptr, *d = nullptr, *n = nullptr; size_t kb = 0, db = 0, nb = 0; };
    std::vector<Slice> ev_slice;
    hipEvent_t ev_fork = nullptr;
    // seq[i] = i - 1: qframe = seq + q0 + 1 (q0, q0 + 1, ...), tframe = seq + q0 (q0 - 1, q0, ...)
    int32_t *d_seq = nullptr;
    int seq_len = 0;
    // the last frame of the previous call (keypoints, descriptors, count): two slots, written alternately
    orbfe_keypoint *d_ckps[2] = {nullptr, nullptr};
    uint8_t *d_cdesc[2] = {nullptr, nullptr};
    int32_t *d_cn[2] = {nullptr, nullptr};
    hipEvent_t ev_carry[2] = {nullptr, nullptr};
    hipEvent_t ev_m0[2] = {nullptr, nullptr};   // behind the frame-0 match of a call that READ carry slot k (recorded on that call's pipe)
    bool m0_valid[2] = {false, false};
    bool carry_written[2] = {false, false};
    int carry_cur = 0;          // slot the NEXT call reads
    bool have_carry = false;
    // where the carried frame was copied FROM (the caller's blocks of the previous call): a sub-batch of the next call that
    // writes over these addresses waits for the copy, the others do not (no drain at the call boundary)
    const void *carry_src[3] = {nullptr, nullptr, nullptr};
    size_t carry_src_bytes[3] = {0, 0, 0};
    bool joined = true;
    int rot = 0;                // pipe of sub-batch 0 of the next call: consecutive short calls take turns on the pipes
    int host_pipes = 1;         // pipes the host entry point deals its chunks to (orbfe_pipeline_set_host_pipes)
    // host entry point (orbfe_pipeline_extract_match): device input / output sets, copy streams
    static const int NSETS = 3;
    uint8_t *d_in[NSETS] = {nullptr, nullptr, nullptr};
    orbfe_keypoint *d_okps[NSETS] = {nullptr, nullptr, nullptr};
    uint8_t *d_odesc[NSETS] = {nullptr, nullptr, nullptr};
    int32_t *d_on[NSETS] = {nullptr, nullptr, nullptr}, *d_om[NSETS] = {nullptr, nullptr, nullptr}, *d_onm[NSETS] = {nullptr, nullptr, nullptr};
    size_t in_bytes = 0;
    int out_frames = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_out[NSETS] = {nullptr, nullptr, nullptr};
};

static orbfe_status ensure_events(orbfe_pipeline *pl, int nsub)
{
    while ((int)pl->ev_ext.size() < nsub) {
        hipEvent_t a = nullptr, b = nullptr;
        ORBFE_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        pl->ev_ext.push_back(a);
        ORBFE_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        pl->ev_match.push_back(b);
        pl->ev_match_valid.push_back(0);
        pl->ev_ext_valid.push_back(0);
        pl->ev_slice.push_back(orbfe_pipeline::Slice());
    }
    return ORBFE_OK;
}

static orbfe_status ensure_seq(orbfe_pipeline *pl, int nframes)
{
    if (nframes + 1 <= pl->seq_len) return ORBFE_OK;
    // the index table is about to be replaced: no launch of an earlier call may still read it
    for (hipStream_t s : pl->st) ORBFE_HIP(hipStreamSynchronize(s));
    if (pl->d_seq) ORBFE_HIP(hipFree(pl->d_seq));
    pl->d_seq = nullptr;
    pl->seq_len = 0;
    const int len = std::max(nframes + 1, 4096);
    std::vector<int32_t> h((size_t)len);
    for (int i = 0; i < len; ++i) h[(size_t)i] = i - 1;
    ORBFE_HIP(hipMalloc((void **)&pl->d_seq, (size_t)len * sizeof(int32_t)));
    ORBFE_HIP(hipMemcpy(pl->d_seq, h.data(), (size_t)len * sizeof(int32_t), hipMemcpyHostToDevice));
    pl->seq_len = len;
    return ORBFE_OK;
}

extern "C" void orbfe_pipeline_destroy(orbfe_pipeline *pl)
{
    if (!pl) return;
    PipeGuard g(pl->device);
    for (hipStream_t s : pl->st)
        if (s) (void)hipStreamSynchronize(s);
    for (orbfe_handle *h : pl->ext) orbfe_destroy(h);
    for (orbfe_matcher *m : pl->mat) orbfe_matcher_destroy(m);
    for (hipEvent_t e : pl->ev_end)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->ev_ext)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->ev_match)
        if (e) (void)hipEventDestroy(e);
    if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
    for (int k = 0; k < 2; ++k) {
        if (pl->ev_carry[k]) (void)hipEventDestroy(pl->ev_carry[k]);
        if (pl->ev_m0[k]) (void)hipEventDestroy(pl->ev_m0[k]);
        if (pl->d_ckps[k]) (void)hipFree(pl->d_ckps[k]);
        if (pl->d_cdesc[k]) (void)hipFree(pl->d_cdesc[k]);
        if (pl->d_cn[k]) (void)hipFree(pl->d_cn[k]);
    }
    if (pl->d_seq) (void)hipFree(pl->d_seq);
    if (pl->s_in) (void)hipStreamSynchronize(pl->s_in);
    if (pl->s_out) (void)hipStreamSynchronize(pl->s_out);
    for (int k = 0; k < orbfe_pipeline::NSETS; ++k) {
        void *bufs[] = {pl->d_in[k], pl->d_okps[k], pl->d_odesc[k], pl->d_on[k], pl->d_om[k], pl->d_onm[k]};
        for (void *b : bufs)
            if (b) (void)hipFree(b);
        if (pl->ev_out[k]) (void)hipEventDestroy(pl->ev_out[k]);
    }
    if (pl->s_in) (void)hipStreamDestroy(pl->s_in);
    if (pl->s_out) (void)hipStreamDestroy(pl->s_out);
    for (hipStream_t s : pl->st)
        if (s) (void)hipStreamDestroy(s);
    for (hipStream_t s : pl->side)
        if (s) (void)hipStreamDestroy(s);
    delete pl;
}

extern "C" orbfe_status orbfe_pipeline_create(const orbfe_params *p, int32_t npipes, orbfe_pipeline **out)
{
    if (!p || !out || npipes < 1 || npipes > 64) {
        orbfe_set_error("bad argument to orbfe_pipeline_create (1..64 pipes)");
        return ORBFE_ERR_ARG;
    }
    *out = nullptr;
    orbfe_pipeline *pl = new (std::nothrow) orbfe_pipeline();
    if (!pl) return ORBFE_ERR_NOMEM;
    pl->prm = *p;
    pl->P = npipes;
    pl->F = p->max_batch;
    auto fail = [&](orbfe_status s) {
        orbfe_pipeline_destroy(pl);
        return s;
    };
    {   // the device the handles will resolve (p->device may be -1 = current); without a device the first create fails below
        int dev = p->device;
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        pl->device = dev;
        pl->prm.device = dev;
    }
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
            (void)hipGetLastError();
            orbfe_set_error("no HIP device visible; liborbfe has no CPU fallback");
            return fail(ORBFE_ERR_NODEVICE);
        }
        if (pl->device >= ndev) { orbfe_set_error("device %d out of range (%d visible)", pl->device, ndev); return fail(ORBFE_ERR_ARG); }
    }
    PipeGuard g(pl->device);
    // One stream per pipe, shared by the pipe's extractor and matcher handles (which then create none of their own but the
    // extractor's side stream): the runtime multiplexes all streams of a process onto a few hardware queues, and every idle
    // stream less is a copy stream that does not have to share its queue with a kernel stream.
    for (int i = 0; i < npipes; ++i) {
        hipStream_t st = nullptr;
        hipEvent_t e = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { orbfe_set_error("pipeline stream creation failed"); return fail(ORBFE_ERR_HIP); }
        pl->st.push_back(st);
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { orbfe_set_error("pipeline event creation failed"); return fail(ORBFE_ERR_HIP); }
        pl->ev_end.push_back(e);
    }
    for (int i = 0; i < std::min(npipes, ORBFE_PIPE_SIDE_STREAMS); ++i) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { orbfe_set_error("pipeline stream creation failed"); return fail(ORBFE_ERR_HIP); }
        pl->side.push_back(st);
    }
    for (int i = 0; i < npipes; ++i) {
        orbfe_handle *h = nullptr;
        orbfe_status s = orbfe_internal_create_on_stream(&pl->prm, (void *)pl->st[(size_t)i], (void *)pl->side[(size_t)i % pl->side.size()], &h);
        if (s != ORBFE_OK) return fail(s);
        pl->ext.push_back(h);
        orbfe_matcher *m = nullptr;
        s = orbfe_internal_matcher_create_on_stream(pl->device, (void *)pl->st[(size_t)i], &m);
        if (s != ORBFE_OK) return fail(s);
        pl->mat.push_back(m);
    }
    // Slots per frame of the host sets and the carry blocks: the extractor's capacity for the planned max_width x max_height, and
    // never less than what ANY frame size can ask for -- a level's selection holds max(N_l + 2, 4 * roots) keypoints and the root
    // count follows the frame's aspect ratio (at most ORBFE_MAX_ROOTS): a later call with another w / ht finds room.
    {
        int32_t feat[ORBFE_MAX_LEVELS];
        int worst = 0;
        if (orbfe_get_features_per_level(pl->ext[0], feat) == ORBFE_OK)
            for (int l = 0; l < pl->prm.nlevels; ++l) worst += std::max(feat[l] + 2, 4 * ORBFE_MAX_ROOTS);
        pl->cap = std::max(orbfe_keypoint_capacity(pl->ext[0]), (worst + 63) & ~63);
    }
    if (hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming) != hipSuccess) { orbfe_set_error("pipeline event creation failed"); return fail(ORBFE_ERR_HIP); }
    for (int k = 0; k < 2; ++k) {
        if (hipEventCreateWithFlags(&pl->ev_carry[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&pl->ev_m0[k], hipEventDisableTiming) != hipSuccess ||
            hipMalloc((void **)&pl->d_ckps[k], (size_t)pl->cap * sizeof(orbfe_keypoint)) != hipSuccess ||
            hipMalloc((void **)&pl->d_cdesc[k], (size_t)pl->cap * 32) != hipSuccess ||
            hipMalloc((void **)&pl->d_cn[k], 64) != hipSuccess) {
            orbfe_set_error("pipeline carry buffers: %s", hipGetErrorString(hipGetLastError()));
            return fail(ORBFE_ERR_NOMEM);
        }
        (void)hipMemset(pl->d_cn[k], 0, 64);
    }
    orbfe_status s = ensure_seq(pl, 4095);
    if (s != ORBFE_OK) return fail(s);
    *out = pl;
    return ORBFE_OK;
}

extern "C" int32_t orbfe_pipeline_pipes(const orbfe_pipeline *pl) { return pl ? pl->P : 0; }
extern "C" int32_t orbfe_pipeline_capacity(const orbfe_pipeline *pl) { return pl ? pl->cap : 0; }
extern "C" int32_t orbfe_pipeline_sub_batch(const orbfe_pipeline *pl) { return pl ? pl->F : 0; }
extern "C" orbfe_handle *orbfe_pipeline_extractor(orbfe_pipeline *pl, int32_t pipe)
{
    return (pl && pipe >= 0 && pipe < pl->P) ? pl->ext[(size_t)pipe] : nullptr;
}
extern "C" orbfe_matcher *orbfe_pipeline_matcher(orbfe_pipeline *pl, int32_t pipe)
{
    return (pl && pipe >= 0 && pipe < pl->P) ? pl->mat[(size_t)pipe] : nullptr;
}

extern "C" orbfe_status orbfe_pipeline_set_host_pipes(orbfe_pipeline *pl, int32_t n)
{
    if (!pl || n < 1) return ORBFE_ERR_ARG;
    pl->host_pipes = std::min(n, pl->P);
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_pipeline_reset_sequence(orbfe_pipeline *pl)
{
    if (!pl) return ORBFE_ERR_ARG;
    pl->have_carry = false;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_pipeline_join(orbfe_pipeline *pl, void *stream)
{
    if (!pl) return ORBFE_ERR_ARG;
    PipeGuard g(pl->device);
    for (int p = 0; p < pl->P; ++p) ORBFE_HIP(hipStreamWaitEvent((hipStream_t)stream, pl->ev_end[(size_t)p], 0));
    pl->joined = true;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_pipeline_synchronize(orbfe_pipeline *pl)
{
    if (!pl) return ORBFE_ERR_ARG;
    PipeGuard g(pl->device);
    for (hipStream_t s : pl->st) ORBFE_HIP(hipStreamSynchronize(s));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_pipeline_get_overflow(orbfe_pipeline *pl, int32_t *flags)
{
    if (!pl || !flags) return ORBFE_ERR_ARG;
    *flags = 0;
    for (orbfe_handle *h : pl->ext) {
        int32_t f = 0;
        const orbfe_status s = orbfe_get_overflow(h, &f);
        if (s != ORBFE_OK) return s;
        *flags |= f;
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_pipeline_extract_match_device(orbfe_pipeline *pl, const uint8_t *d_gray, int32_t nframes, int32_t w,
                                                            int32_t ht, int32_t stride, size_t frame_stride, orbfe_keypoint *d_kps,
                                                            uint8_t *d_desc, int32_t cap, int32_t *d_n_out, int32_t *d_match,
                                                            int32_t *d_nmatches, float nnratio, int32_t th, int32_t check_ori,
                                                            int32_t flags, void *stream)
{
    if (!pl || !d_gray || !d_kps || !d_desc || !d_n_out || nframes < 1 || cap < 1 || (d_match && !d_nmatches)) {
        orbfe_set_error("bad argument to orbfe_pipeline_extract_match_device");
        return ORBFE_ERR_ARG;
    }
    PipeGuard g(pl->device);
    const int F = pl->F, P = pl->P;
    const int nsub = (nframes + F - 1) / F;
    orbfe_status s = ensure_events(pl, nsub + 1);
    if (s != ORBFE_OK) return s;
    s = ensure_seq(pl, nframes);
    if (s != ORBFE_OK) return s;
    hipStream_t cs = (hipStream_t)stream;
    const bool match = d_match != nullptr;
    const bool cont = (flags & ORBFE_PIPE_CONTINUE) != 0 && pl->have_carry && match;
    const int rd = pl->carry_cur, wr = pl->carry_cur ^ 1;

    // fork: the pipes start behind whatever the caller's stream holds (the producer of d_gray, the consumer of the output
    // blocks of an earlier call)
    ORBFE_HIP(hipEventRecord(pl->ev_fork, cs));
    for (int p = 0; p < P; ++p) ORBFE_HIP(hipStreamWaitEvent(pl->st[(size_t)p], pl->ev_fork, 0));
    // An error in the middle of the loop leaves launches of this call in flight and the event / carry bookkeeping half
    // updated: drain the pipes and start the sequence over (the next call has no predecessor frame), then report.
    auto bail = [&](orbfe_status e) {
        for (hipStream_t x : pl->st) (void)hipStreamSynchronize(x);
        pl->have_carry = false;
        pl->m0_valid[0] = pl->m0_valid[1] = false;
        pl->joined = true;
        return e;
    };
    auto overlaps = [](const void *a, size_t na, const void *b, size_t nb) {
        const char *pa = (const char *)a, *pb = (const char *)b;
        return a && b && pa < pb + nb && pb < pa + na;
    };

    for (int j = 0; j < nsub; ++j) {
        const int p = (pl->rot + j) % P;
        hipStream_t st = pl->st[(size_t)p];
        const int lo = j * F, nf = std::min(F, nframes - lo);
        // The output blocks may be the ones of the previous call (a host that re-uses its buffers), and the pipes take turns: what
        // the PREVIOUS call did with slices j happened on other pipes' streams.  This sub-batch overwrites them only after that
        // call's extraction of sub-batch j (write after write), its matcher of sub-batch j (queries) and its matcher of
        // sub-batch j + 1 (whose first pair reads the last frame of slice j) have finished.  (Events of finished work cost nothing.)
        {
            const char *ok = (const char *)(d_kps + (size_t)lo * cap), *od = (const char *)(d_desc + (size_t)lo * cap * 32), *on = (const char *)(d_n_out + lo);
            const size_t okb = (size_t)nf * cap * sizeof(orbfe_keypoint), odb = (size_t)nf * cap * 32, onb = (size_t)nf * sizeof(int32_t);
            auto wait_index = [&](size_t i) -> hipError_t {   // everything that wrote or read the slices recorded under index i
                hipError_t e = hipSuccess;
                if (pl->ev_ext_valid[i]) e = hipStreamWaitEvent(st, pl->ev_ext[i], 0);
                if (e == hipSuccess && pl->ev_match_valid[i]) e = hipStreamWaitEvent(st, pl->ev_match[i], 0);
                if (e == hipSuccess && i + 1 < pl->ev_match_valid.size() && pl->ev_match_valid[i + 1]) e = hipStreamWaitEvent(st, pl->ev_match[i + 1], 0);
                return e;
            };
            // index j (its events are about to be re-recorded), and every other index whose recorded slices overlap this sub-batch's:
            // with the usual re-use of the same buffers and call size that is index j alone
            ORBFE_HIP(wait_index((size_t)j));
            for (size_t i = 0; i < pl->ev_slice.size(); ++i) {
                const orbfe_pipeline::Slice &si = pl->ev_slice[i];
                if ((int)i != j && si.k && (overlaps(si.k, si.kb, ok, okb) || overlaps(si.d, si.db, od, odb) || overlaps(si.n, si.nb, on, onb)))
                    ORBFE_HIP(wait_index(i));
            }
        }
        // ... and the copy of the previous call's last frame into the carry slot must have read it before this sub-batch
        // writes over it (only the sub-batch whose output slices cover those addresses waits: no drain at the call boundary)
        if (pl->have_carry &&
            (overlaps(pl->carry_src[0], pl->carry_src_bytes[0], d_kps + (size_t)lo * cap, (size_t)nf * cap * sizeof(orbfe_keypoint)) ||
             overlaps(pl->carry_src[1], pl->carry_src_bytes[1], d_desc + (size_t)lo * cap * 32, (size_t)nf * cap * 32) ||
             overlaps(pl->carry_src[2], pl->carry_src_bytes[2], d_n_out + lo, (size_t)nf * sizeof(int32_t))))
            ORBFE_HIP(hipStreamWaitEvent(st, pl->ev_carry[rd], 0));
        s = orbfe_extract_batch_device(pl->ext[(size_t)p], d_gray + (size_t)lo * frame_stride, nf, w, ht, stride, frame_stride,
                                       d_kps + (size_t)lo * cap, d_desc + (size_t)lo * cap * 32, cap, d_n_out + lo, (void *)st);
        if (s != ORBFE_OK) return bail(s);
        ORBFE_HIP(hipEventRecord(pl->ev_ext[(size_t)j], st));
        pl->ev_ext_valid[(size_t)j] = 1;
        {
            orbfe_pipeline::Slice &sj = pl->ev_slice[(size_t)j];
            sj.k = (const char *)(d_kps + (size_t)lo * cap); sj.kb = (size_t)nf * cap * sizeof(orbfe_keypoint);
            sj.d = (const char *)(d_desc + (size_t)lo * cap * 32); sj.db = (size_t)nf * cap * 32;
            sj.n = (const char *)(d_n_out + lo); sj.nb = (size_t)nf * sizeof(int32_t);
        }
        if (!match) continue;
        if (j > 0) ORBFE_HIP(hipStreamWaitEvent(st, pl->ev_ext[(size_t)j - 1], 0));   // frame lo - 1 comes from the neighbour pipe
        const int q0 = lo == 0 ? 1 : lo;
        const int np = lo + nf - q0;
        if (np > 0) {
            s = orbfe_match_bf_blocks_device(pl->mat[(size_t)p], d_kps, d_desc, d_n_out, d_kps, d_desc, d_n_out, cap, pl->d_seq + q0 + 1,
                                             pl->d_seq + q0, np, nnratio, th, check_ori, d_match + (size_t)q0 * cap, d_nmatches + q0,
                                             (void *)st);
            if (s != ORBFE_OK) return bail(s);
        }
        if (lo == 0) {
            if (cont) {
                ORBFE_HIP(hipStreamWaitEvent(st, pl->ev_carry[rd], 0));
                // query = frame 0 of this call, train = the carried frame (frame 0 of the carry block)
                s = orbfe_match_bf_blocks_device(pl->mat[(size_t)p], d_kps, d_desc, d_n_out, pl->d_ckps[rd], pl->d_cdesc[rd], pl->d_cn[rd], cap,
                                                 pl->d_seq + 1, pl->d_seq + 1, 1, nnratio, th, check_ori, d_match, d_nmatches, (void *)st);
                if (s != ORBFE_OK) return bail(s);
                ORBFE_HIP(hipEventRecord(pl->ev_m0[rd], st));   // slot rd has been read: the NEXT call may write it
                pl->m0_valid[rd] = true;
            } else {  // the first frame of a sequence has no predecessor
                ORBFE_HIP(hipMemsetAsync(d_match, 0xFF, (size_t)cap * sizeof(int32_t), st));
                ORBFE_HIP(hipMemsetAsync(d_nmatches, 0, sizeof(int32_t), st));
            }
        }
        ORBFE_HIP(hipEventRecord(pl->ev_match[(size_t)j], st));
        pl->ev_match_valid[(size_t)j] = 1;
    }
    // (the events of sub-batch indices this call did not use keep their last record: a later, longer call still orders itself
    // behind whatever touched those slices last)

    // carry: the last frame of this call, for the first frame of the next one.  Slot `wr` was read by the PREVIOUS call's
    // frame-0 match, on whatever pipe that call's sub-batch 0 ran (the pipes take turns): the copy waits for that match's own
    // event.
    {
        const int jl = nsub - 1;
        hipStream_t st = pl->st[(size_t)((pl->rot + jl) % P)];
        const size_t last = (size_t)nframes - 1;
        if (pl->m0_valid[wr]) {
            ORBFE_HIP(hipStreamWaitEvent(st, pl->ev_m0[wr], 0));
            pl->m0_valid[wr] = false;
        }
        // ... and behind the previous WRITER of the slot (two calls ago, possibly on another pipe's stream): calls without a frame-0
        // match (extract only, no CONTINUE) record no ev_m0, and nothing else would order the two copies
        if (pl->carry_written[wr]) ORBFE_HIP(hipStreamWaitEvent(st, pl->ev_carry[wr], 0));
        ORBFE_HIP(hipMemcpyAsync(pl->d_ckps[wr], d_kps + last * cap, (size_t)std::min(cap, pl->cap) * sizeof(orbfe_keypoint),
                                 hipMemcpyDeviceToDevice, st));
        ORBFE_HIP(hipMemcpyAsync(pl->d_cdesc[wr], d_desc + last * cap * 32, (size_t)std::min(cap, pl->cap) * 32, hipMemcpyDeviceToDevice, st));
        ORBFE_HIP(hipMemcpyAsync(pl->d_cn[wr], d_n_out + last, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        ORBFE_HIP(hipEventRecord(pl->ev_carry[wr], st));
        pl->carry_written[wr] = true;
        pl->carry_cur = wr;
        pl->have_carry = true;
        pl->carry_src[0] = d_kps + last * cap;
        pl->carry_src_bytes[0] = (size_t)std::min(cap, pl->cap) * sizeof(orbfe_keypoint);
        pl->carry_src[1] = d_desc + last * cap * 32;
        pl->carry_src_bytes[1] = (size_t)std::min(cap, pl->cap) * 32;
        pl->carry_src[2] = d_n_out + last;
        pl->carry_src_bytes[2] = sizeof(int32_t);
    }
    for (int p = 0; p < P; ++p) ORBFE_HIP(hipEventRecord(pl->ev_end[(size_t)p], pl->st[(size_t)p]));
    pl->joined = false;
    pl->rot = (pl->rot + nsub) % P;
    if (!(flags & ORBFE_PIPE_NO_JOIN)) return orbfe_pipeline_join(pl, stream);
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// host frames in, host results out: chunks of one sub-batch, H2D / pipes / D2H overlapped over three device buffer sets
// ---------------------------------------------------------------------------------------------------
static orbfe_status ensure_host_sets(orbfe_pipeline *pl, int w, int ht)
{
    const size_t need = (size_t)pl->F * w * ht;
    if (need <= pl->in_bytes && pl->out_frames == pl->F) return ORBFE_OK;
    // (measured and rejected: copy streams at the highest stream priority -- 159 k frames/s with 3 pipes, 41 k with 12)
    if (!pl->s_in) ORBFE_HIP(hipStreamCreateWithFlags(&pl->s_in, hipStreamNonBlocking));
    if (!pl->s_out) ORBFE_HIP(hipStreamCreateWithFlags(&pl->s_out, hipStreamNonBlocking));
    ORBFE_HIP(hipStreamSynchronize(pl->s_in));
    ORBFE_HIP(hipStreamSynchronize(pl->s_out));
    for (hipStream_t st : pl->st) ORBFE_HIP(hipStreamSynchronize(st));
    const size_t F = (size_t)pl->F, cap = (size_t)pl->cap;
    for (int k = 0; k < orbfe_pipeline::NSETS; ++k) {
        if (pl->d_in[k]) ORBFE_HIP(hipFree(pl->d_in[k]));
        pl->d_in[k] = nullptr;
        ORBFE_HIP(hipMalloc((void **)&pl->d_in[k], need + 64));
        if (!pl->ev_out[k]) ORBFE_HIP(hipEventCreateWithFlags(&pl->ev_out[k], hipEventDisableTiming));
        if (!pl->d_okps[k]) {
            ORBFE_HIP(hipMalloc((void **)&pl->d_okps[k], F * cap * sizeof(orbfe_keypoint)));
            ORBFE_HIP(hipMalloc((void **)&pl->d_odesc[k], F * cap * 32));
            ORBFE_HIP(hipMalloc((void **)&pl->d_on[k], F * sizeof(int32_t)));
            ORBFE_HIP(hipMalloc((void **)&pl->d_om[k], F * cap * sizeof(int32_t)));
            ORBFE_HIP(hipMalloc((void **)&pl->d_onm[k], F * sizeof(int32_t)));
        }
    }
    pl->in_bytes = need;
    pl->out_frames = pl->F;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_pipeline_extract_match(orbfe_pipeline *pl, const uint8_t *const *grays, int32_t nframes, int32_t w, int32_t ht,
                                                     int32_t stride, orbfe_keypoint *kps, uint8_t *desc, int32_t cap, int32_t *n_out,
                                                     int32_t *match, int32_t *nmatches, float nnratio, int32_t th, int32_t check_ori,
                                                     int32_t flags)
{
    if (!pl || !grays || !kps || !desc || !n_out || nframes < 1 || w < 1 || ht < 1 || stride < w || (match && !nmatches)) {
        orbfe_set_error("bad argument to orbfe_pipeline_extract_match");
        return ORBFE_ERR_ARG;
    }
    if (cap < pl->cap) {
        orbfe_set_error("orbfe_pipeline_extract_match: cap %d below orbfe_pipeline_capacity() = %d", cap, pl->cap);
        return ORBFE_ERR_CAP;
    }
    PipeGuard g(pl->device);
    orbfe_status s = ensure_host_sets(pl, w, ht);
    if (s != ORBFE_OK) return s;
    const int F = pl->F, pc = pl->cap;
    const size_t fbytes = (size_t)w * ht;
    const int nchunks = (nframes + F - 1) / F;
    // Whatever way the loop is left, copies that read the caller's frames or write its arrays may be in flight: every exit
    // drains the copy streams and the pipes first.
    struct Drain {
        orbfe_pipeline *pl;
        ~Drain()
        {
            if (pl->s_in) (void)hipStreamSynchronize(pl->s_in);
            for (hipStream_t st : pl->st) (void)hipStreamSynchronize(st);
            if (pl->s_out) (void)hipStreamSynchronize(pl->s_out);
        }
    } drain{pl};
    for (int c = 0; c < nchunks; ++c) {
        const int k = c % orbfe_pipeline::NSETS, lo = c * F, nf = std::min(F, nframes - lo);
        // set k is free once the results of chunk c - NSETS have left it
        if (c >= orbfe_pipeline::NSETS) ORBFE_HIP(hipStreamWaitEvent(pl->s_in, pl->ev_out[k], 0));
        bool contiguous = stride == w;
        for (int f = 1; f < nf && contiguous; ++f) contiguous = grays[lo + f] == grays[lo + f - 1] + fbytes;
        if (contiguous) {
            ORBFE_HIP(hipMemcpyAsync(pl->d_in[k], grays[lo], fbytes * nf, hipMemcpyHostToDevice, pl->s_in));
        } else {
            for (int f = 0; f < nf; ++f)
                ORBFE_HIP(hipMemcpy2DAsync(pl->d_in[k] + fbytes * f, (size_t)w, grays[lo + f], (size_t)stride, (size_t)w, (size_t)ht,
                                           hipMemcpyHostToDevice, pl->s_in));
        }
        // the pipes start behind the copy (the call forks from s_in) and are not joined: the next chunk's copy and pipes follow at once
        const int fl = ((c > 0 || (flags & ORBFE_PIPE_CONTINUE)) ? ORBFE_PIPE_CONTINUE : 0) | ORBFE_PIPE_NO_JOIN;
        pl->rot = c % std::max(1, std::min(pl->host_pipes, pl->P));   // the chunk is one sub-batch: this is its pipe
        s = orbfe_pipeline_extract_match_device(pl, pl->d_in[k], nf, w, ht, w, fbytes, pl->d_okps[k], pl->d_odesc[k], pc, pl->d_on[k],
                                                match ? pl->d_om[k] : nullptr, match ? pl->d_onm[k] : nullptr, nnratio, th, check_ori, fl,
                                                (void *)pl->s_in);
        if (s != ORBFE_OK) return s;
        s = orbfe_pipeline_join(pl, (void *)pl->s_out);   // everything submitted so far, i.e. this chunk and older ones
        if (s != ORBFE_OK) return s;
        // padded blocks straight into the caller's arrays (row pitch cap >= pc)
        ORBFE_HIP(hipMemcpyAsync(n_out + lo, pl->d_on[k], (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, pl->s_out));
        auto rows_out = [&](void *dst, const void *src, size_t elem) -> hipError_t {   // nf rows of pc elements, host pitch cap
            if (cap == pc) return hipMemcpyAsync(dst, src, (size_t)nf * pc * elem, hipMemcpyDeviceToHost, pl->s_out);   // one linear copy
            return hipMemcpy2DAsync(dst, (size_t)cap * elem, src, (size_t)pc * elem, (size_t)pc * elem, (size_t)nf, hipMemcpyDeviceToHost, pl->s_out);
        };
        ORBFE_HIP(rows_out(kps + (size_t)lo * cap, pl->d_okps[k], sizeof(orbfe_keypoint)));
        ORBFE_HIP(rows_out(desc + (size_t)lo * cap * 32, pl->d_odesc[k], 32));
        if (match) {
            ORBFE_HIP(rows_out(match + (size_t)lo * cap, pl->d_om[k], sizeof(int32_t)));
            ORBFE_HIP(hipMemcpyAsync(nmatches + lo, pl->d_onm[k], (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, pl->s_out));
        }
        ORBFE_HIP(hipEventRecord(pl->ev_out[k], pl->s_out));
    }
    ORBFE_HIP(hipStreamSynchronize(pl->s_out));
    int32_t ovf = 0;
    s = orbfe_pipeline_get_overflow(pl, &ovf);
    if (s != ORBFE_OK) return s;
    if (ovf) {
        orbfe_set_error("device-side capacity overflow %d in orbfe_pipeline_extract_match", ovf);
        return ORBFE_ERR_CAP;
    }
    return ORBFE_OK;
}
