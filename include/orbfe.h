/*
 * orbfe.h -- C-ABI of the MI355X-native ORB front-end (liborbfe.so).
 *
 * Drop-in boundary for ONE hot path of Ewenwan/ORB_SLAM2_SSD_Semantic:
 *   ORB_SLAM2::ORBextractor::operator()      (reference include/ORBextractor.h:53-55, src/ORBextractor.cc:1052)
 *   ORB_SLAM2::ORBmatcher Hamming core        (reference include/ORBmatcher.h:41-112, src/ORBmatcher.cc)
 * The reference has no FFI: the "operator API" is two C++ classes.  The C++ shim in
 * orb_slam2_ssd_semantic_amd/shim/ re-declares those classes with identical signatures on top of
 * this header (INTEGRATION.md shows the binding).  Plain C, POD only, caller-owned buffers, every
 * call returns an orbfe_status (0 = ok, negative = error); nothing throws or aborts.
 *
 * Threading: a handle is used by one thread at a time (like an ORBextractor instance, which mutates
 * mvImagePyramid); different handles may be used concurrently (stereo: src/Frame.cc:121-122;
 * matchers are created per call site from three SLAM threads).  There is no global
 * mutable state.
 *
 * All compute runs in hand-written HIP kernels for gfx950.  There is NO CPU fallback: without a
 * usable HIP device orbfe_create / orbfe_matcher_create fail with ORBFE_ERR_NODEVICE.
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_VERSION 100 /* 0.1.0 */

typedef int32_t orbfe_status;
enum {
    ORBFE_OK = 0,
    ORBFE_ERR_ARG = -1,      /* null pointer / negative size / inconsistent arguments          */
    ORBFE_ERR_SIZE = -2,     /* image larger than planned or too small for the 8-level grid    */
    ORBFE_ERR_CAP = -3,      /* caller's keypoint capacity too small; *n_out holds the need    */
    ORBFE_ERR_HIP = -4,      /* a HIP runtime call failed (see orbfe_last_error)               */
    ORBFE_ERR_NOMEM = -5,    /* device or host allocation failed                               */
    ORBFE_ERR_NODEVICE = -6, /* no usable HIP device: there is no CPU path                     */
    ORBFE_ERR_STATE = -7     /* call made in the wrong state (e.g. taps before any extract)    */
};

/* = cv::KeyPoint field order (T1): {Point2f pt; float size, angle, response; int octave, class_id}. 28 B. */
typedef struct orbfe_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbfe_keypoint;

/* Constructor arguments of ORBextractor (include/ORBextractor.h:45-46, values read from the settings
 * YAML at src/Tracking.cc:198-206) plus planning sizes for the device buffers. */
typedef struct orbfe_params {
    int32_t nfeatures;      /* ORBextractor.nFeatures   (TUM3.yaml:41-54: 1000) */
    float scale_factor;     /* ORBextractor.scaleFactor (1.2)                   */
    int32_t nlevels;        /* ORBextractor.nLevels     (8), 1..16              */
    int32_t ini_th_fast;    /* ORBextractor.iniThFAST   (20)                    */
    int32_t min_th_fast;    /* ORBextractor.minThFAST   (7)                     */
    int32_t max_width;      /* largest image width this handle will see  (<= 4096) */
    int32_t max_height;     /* largest image height this handle will see (<= 4096) */
    int32_t max_batch;      /* frames per batched call (>= 1)                   */
    int32_t device;         /* HIP device ordinal, -1 = current device          */
    int32_t blur_rounding;  /* 0 = canonical half-up; 1 = emulate the x86 SSE2 column kernel */
} orbfe_params;

typedef struct orbfe_handle orbfe_handle;   /* one ORBextractor instance */
typedef struct orbfe_matcher orbfe_matcher; /* scratch + stream for ORBmatcher calls */

/* ---------------------------------------------------------------------------------------------
 * Library
 * ------------------------------------------------------------------------------------------- */
int32_t orbfe_version(void);
const char *orbfe_strerror(orbfe_status s);
/* thread-local text of the last failure on the calling thread ("" if none) */
const char *orbfe_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int32_t orbfe_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Extractor  (replaces ORBextractor: ctor src/ORBextractor.cc:399-466, operator() :1052-1114)
 * ------------------------------------------------------------------------------------------- */
/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) */
orbfe_status orbfe_create(const orbfe_params *p, orbfe_handle **out);
void orbfe_destroy(orbfe_handle *h);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (include/ORBextractor.h:58-78); nlevels floats each, any pointer may be NULL. */
orbfe_status orbfe_get_scales(const orbfe_handle *h, float *scale, float *inv_scale, float *sigma2,
                              float *inv_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:426-439); nlevels ints */
orbfe_status orbfe_get_features_per_level(const orbfe_handle *h, int32_t *out);
/* upper bound of keypoints one frame can produce: sum over levels of max(N_l + 2, 4 * nIni_l) -- the quadtree may end
 * a pass with up to 3 nodes more than asked, and never with fewer than the initial roots' children -- rounded up to 64 */
int32_t orbfe_keypoint_capacity(const orbfe_handle *h);

/* ORBextractor::operator()(image, mask, keypoints, descriptors) for one 8-bit gray frame in HOST
 * memory (mask is ignored by the reference, :1052).  kps/desc are caller-owned with room for `cap`
 * keypoints; *n_out receives the count.  w==0 || h==0 || gray==NULL -> ORBFE_OK, outputs untouched
 * (:1055-1056).  Order of the output is the reference's (level-major, quadtree list order). */
orbfe_status orbfe_extract(orbfe_handle *h, const uint8_t *gray, int32_t w, int32_t ht, int32_t stride,
                           orbfe_keypoint *kps, uint8_t *desc /* cap x 32 */, int32_t cap, int32_t *n_out);

/* Batched keyframe mode, HOST buffers: frames are independent.  grays[i] points to
 * frame i (all w x ht, same stride).  Frame i writes kps[i*cap ..], desc[i*cap*32 ..], n_out[i].
 * More than max_batch frames run as a pipeline (H2D of the next chunk, kernels of the current one and D2H of the previous
 * one overlap on three streams).  Page-locked memory (hipHostMalloc / hipHostRegister / torch pin_memory) is used as it
 * is: frames with stride == w are copied straight from the caller's buffers, and page-locked kps / desc / n_out arrays
 * receive the padded device blocks directly (slots >= n_out[i] zero-filled); pageable memory goes through the handle's
 * own pinned staging sets. */
orbfe_status orbfe_extract_batch(orbfe_handle *h, const uint8_t *const *grays, int32_t nframes, int32_t w,
                                 int32_t ht, int32_t stride, orbfe_keypoint *kps, uint8_t *desc, int32_t cap,
                                 int32_t *n_out);

/* Batched keyframe mode, DEVICE buffers (HBM-resident input, what bench.py times):
 *   d_gray   : nframes frames, frame i at d_gray + i*frame_stride, row pitch `stride`; nothing outside
 *              [d_gray, last pixel of the last frame] is read (row windows are pulled back at the right edge)
 *   d_kps    : nframes*cap orbfe_keypoint      d_desc : nframes*cap*32 bytes     d_n_out : nframes int32
 * All work is enqueued on `stream`, a hipStream_t passed as void*.  NULL is HIP's (legacy) default stream
 * -- the stream PyTorch uses unless told otherwise; pass orbfe_get_stream(h) for the handle's own
 * non-blocking stream.  The call returns without synchronising.  Slots >= n_out[i] of a frame are zero-filled so the
 * padded buffers can be all-gathered as they are.
 * Lifetime: level 0 of the pyramid is read IN PLACE from d_gray, also by later orbfe_get_pyramid_level /
 * orbfe_tap_* / orbfe_stereo_matches calls on this handle -- keep d_gray alive and unchanged until the next extract
 * call (or until you are done with those calls).  With the host entry points only the frames of the last chunk of
 * max_batch frames stay addressable. */
orbfe_status orbfe_extract_batch_device(orbfe_handle *h, const uint8_t *d_gray, int32_t nframes, int32_t w,
                                        int32_t ht, int32_t stride, size_t frame_stride,
                                        orbfe_keypoint *d_kps, uint8_t *d_desc, int32_t cap,
                                        int32_t *d_n_out, void *stream);
/* Capacity check for the DEVICE entry point (the host entry points do it themselves and return ORBFE_ERR_CAP):
 * every internal list is sized for its worst case, and the kernels raise a sticky device-side word instead of
 * silently dropping data if a size is ever exceeded.  Waits for the stream of the last batched call, returns and
 * clears the word: bit 0 = a level's FAST survivor list overflowed, bit 1 = a level's quadtree selection overflowed,
 * bit 2 = a frame produced more keypoints than `cap` (d_n_out holds the required count). */
orbfe_status orbfe_get_overflow(orbfe_handle *h, int32_t *flags);
/* FAST kernel variant.  Results are identical in every mode.
 *   0  dense: the 16 nine-arcs of every pixel
 *   1  dense with wave-uniform shortcuts (skips the arc evaluation of 256-pixel row pieces that fail a 4-point necessary
 *      test, and the suppression of rows without strength)
 *   2  lane-compacting: every pixel pair takes the 4-point necessary test; the pairs that pass leave a tag in a queue and are
 *      evaluated, 64 at a time, one per lane, from a ring of the pixel rows the wave holds in LDS -- cheaper than dense when few
 *      pairs pass (FAST stage of 1024 frames of 640x480: -16 % at the 18 % of the camera-like synthetic frames S_tum, -28 % at
 *      8 %, -40 % at 2 %; break-even near 27 %), dearer when more do (+45 % at the 84 % of the corner-saturated frames S), and
 *      dearer for calls that do not fill the GPU (its waves are longer: +15 % on one frame)
 *   3  auto (the default): 0 for calls that do not fill the GPU (less work than about 29 frames of 640x480); otherwise 2, and 0 for the next 16 calls (doubling up to 256 while
 *      it keeps happening) whenever the compacting kernel reported more than 25 % passing pairs
 * collect_stats != 0 counts {row steps, arc skips, NMS skips} (mode 1) / {row steps, batches, parked pairs} of a sample of the
 * waves (mode 2); in mode 3 orbfe_get_fast_stats returns the counters of the last probe that completed (zeros before one has). */
orbfe_status orbfe_set_fast_mode(orbfe_handle *h, int32_t mode, int32_t collect_stats);
orbfe_status orbfe_get_fast_stats(orbfe_handle *h, uint64_t out[3], int32_t reset);
/* work model of the FAST kernel for the current frame size: out[0] = wave row steps per frame (one step = 64 lanes x 4
 * pixels of one row, halo rows included), out[1] = waves per frame.  bench.py prices its VALU ceiling with it. */
orbfe_status orbfe_get_work_counts(const orbfe_handle *h, int64_t out[2]);
/* Launch-shape and A/B options of a handle.  The release library takes them ONLY through this call -- it reads no tuning
 * knob from the process environment (a library inside a SLAM process must not change algorithm because of the host's
 * environment).  Every setting gives byte-identical results (ORBFE_OPT_BLUR_ROUNDING excepted).  Built-in choices, i.e. the value that
 * restores the default: OVERLAP -1; ROWS / ROWS_FAST / ROWS_BLUR / PYR_ROWS / QT_THREADS_* 0; BLUR_PIECES 1 (0 = the older packing, an
 * A/B variant); BLUR_UPDOWN 1 (0 = never, 2 = always); DEBUG 0; the [dev] fusions 0; FUSE_FAST_PYR_LEVELS 0 (= all levels; 1..8 = that
 * many); REUSE_IDENTICAL_INPUT 0.
 * Options marked [dev] select kernel variants that were measured slower than the default and are compiled only into a
 * developer build (-DORBFE_DEVELOPER, tools/ab_build.sh; such a build also honours $ORBFE_<NAME> at orbfe_create): a
 * release build answers ORBFE_ERR_STATE to a non-zero value.  Call between extract calls, not concurrently with one. */
enum {
    ORBFE_OPT_OVERLAP = 1,        /* blur on the side stream: 0 never, 1 from before FAST, 2 beside the quadtree, -1 by batch size */
    ORBFE_OPT_ROWS = 2,           /* rows a FAST / blur wave walks (8..512) */
    ORBFE_OPT_ROWS_FAST = 3,      /* ... FAST only */
    ORBFE_OPT_ROWS_BLUR = 4,      /* ... blur only */
    ORBFE_OPT_BLUR_PIECES = 5,    /* 1 (default): blur lanes laid out in 64-byte pieces */
    ORBFE_OPT_BLUR_UPDOWN = 6,    /* odd row blocks of the blur walk upwards: 0 never, 1 where it adds no wave (default), 2 always */
    ORBFE_OPT_PYR_ROWS = 7,       /* destination rows per lane run of the pyramid kernel (2..16) */
    ORBFE_OPT_QT_THREADS_0 = 8,   /* threads per workgroup of the quadtree's three level groups (64..512, multiple of 64) */
    ORBFE_OPT_QT_THREADS_1 = 9,
    ORBFE_OPT_QT_THREADS_2 = 10,
    ORBFE_OPT_DEBUG = 11,         /* forces production code paths that ordinary frames rarely take (tests): 50 = quadtree by
                                   * streaming key passes only (the deep-tree path), 51 = generic node passes only */
    ORBFE_OPT_PYR_FUSE = 12,      /* [dev] 1: two pyramid levels per launch */
    ORBFE_OPT_FUSE_BLUR_PYR = 13, /* [dev] 1 / 2: blur + resize in one chained pass */
    ORBFE_OPT_FUSE_FAST_PYR = 14, /* [dev] 1 / 2: FAST + resize in one launch per level, 3: FAST of level 0 beside the pyramid */
    ORBFE_OPT_FUSE_FAST_PYR_LEVELS = 15,
    ORBFE_OPT_BLUR_ROUNDING = 16, /* orbfe_params.blur_rounding of an existing handle (0 / 1); this one changes RESULTS */
    ORBFE_OPT_REUSE_IDENTICAL_INPUT = 17 /* single-frame host calls (orbfe_extract; orbfe_extract_batch with one frame), default 0.  1: a call whose
                                   * pixels equal those of the previous such call of this handle (same w, ht, cap; compared on the host against the pinned
                                   * staging copy, stride-independent) returns that call's keypoints / descriptors without
                                   * touching the GPU; pyramid and taps stay those of that frame.  For callers that extract the
                                   * same image twice: perfect/src/Tracking.cc:685 and :716 build two Frames from one mImGray.
                                   * Bit-exact by construction.  Any other use of the handle in between drops the cached frame. */
};
orbfe_status orbfe_set_option(orbfe_handle *h, int32_t option, int32_t value);
/* 1 when the last orbfe_extract call of the handle was answered from the previous call's results (ORBFE_OPT_REUSE_IDENTICAL_INPUT) */
int32_t orbfe_last_call_reused(const orbfe_handle *h);
/* the handle's own non-blocking stream (hipStream_t as void*): the host-buffer entry points run on it */
void *orbfe_get_stream(orbfe_handle *h);
/* block until everything enqueued by this handle on its own stream has finished */
orbfe_status orbfe_synchronize(orbfe_handle *h);

/* mvImagePyramid[level] of the LAST extract call, frame `frame` of the batch (include/ORBextractor.h:80;
 * read by Frame::ComputeStereoMatches src/Frame.cc:649,761-778).  with_border != 0 returns the
 * (w+38) x (h+38) BORDER_REFLECT_101-padded image (src/ORBextractor.cc:1136-1142), else w x h. */
orbfe_status orbfe_get_level_size(const orbfe_handle *h, int32_t level, int32_t *w, int32_t *ht);
orbfe_status orbfe_get_pyramid_level(orbfe_handle *h, int32_t frame, int32_t level, uint8_t *dst,
                                     int32_t dst_stride, int32_t with_border);
/* The same for ALL levels in one go -- the public mvImagePyramid of the reference (src/ORBextractor.cc:1128-1142): level l of
 * frame `frame` with its 19-px BORDER_REFLECT_101 frame as a (w_l + 38) x (h_l + 38) block with tight rows (pitch w_l + 38)
 * at offsets[l] of dst (blocks back to back, each offset a multiple of 64); *total = bytes needed.  dst == NULL: sizes only.
 * One kernel + ONE device-to-host copy (dst may be pageable).  offsets [nlevels] and total may be NULL. */
orbfe_status orbfe_get_pyramid_padded(orbfe_handle *h, int32_t frame, uint8_t *dst, size_t cap, size_t *offsets, size_t *total);

/* ---- stage taps of the last extract call (parity tests / debugging; host copies) ---- */
/* 7x7 sigma-2 blurred level (the private clone of src/ORBextractor.cc:1094-1095) */
orbfe_status orbfe_tap_blurred_level(orbfe_handle *h, int32_t frame, int32_t level, uint8_t *dst,
                                     int32_t dst_stride);
/* FAST candidates handed to DistributeOctTree for one level, reference order (cell-row-major, raster):
 * xyr = n x {x, y, response} floats in detection-window coordinates (src/ORBextractor.cc:831-833) */
orbfe_status orbfe_tap_candidates(orbfe_handle *h, int32_t frame, int32_t level, float *xyr, int32_t cap,
                                  int32_t *n);
/* keypoints kept by DistributeOctTree for one level, list order, level coordinates (border added) */
orbfe_status orbfe_tap_selected(orbfe_handle *h, int32_t frame, int32_t level, float *xyr, int32_t cap,
                                int32_t *n);

/* ---- timing of the last batched call (HIP events on the stream the kernels ran on) ---- */
enum {
    ORBFE_T_PYRAMID = 0, /* all resize launches                                  */
    ORBFE_T_FAST = 1,    /* FAST score + per-cell NMS + threshold fallback       */
    ORBFE_T_OCTREE = 2,  /* DistributeOctTree                                    */
    ORBFE_T_BLUR = 3,    /* 7x7 Gaussian                                         */
    ORBFE_T_DESC = 4,    /* IC_Angle + rBRIEF + keypoint assembly                */
    ORBFE_T_TOTAL = 5,   /* first launch -> last launch                          */
    ORBFE_T_COUNT = 6
};
/* enable != 0: record events around each stage of subsequent calls (adds a few us per call) and reset
 * the averaging window */
orbfe_status orbfe_set_profiling(orbfe_handle *h, int32_t enable);
/* milliseconds per stage, AVERAGED over the calls made since profiling was enabled (the 64 most recent at
 * most); waits for those calls to finish */
orbfe_status orbfe_get_stage_ms(orbfe_handle *h, float ms[ORBFE_T_COUNT]);

/* ---------------------------------------------------------------------------------------------
 * Matcher  (replaces the Hamming core of ORBmatcher, src/ORBmatcher.cc)
 * ------------------------------------------------------------------------------------------- */
#define ORBFE_TH_HIGH 100     /* ORBmatcher::TH_HIGH      src/ORBmatcher.cc:39 */
#define ORBFE_TH_LOW 50       /* ORBmatcher::TH_LOW       src/ORBmatcher.cc:40 */
#define ORBFE_HISTO_LENGTH 30 /* ORBmatcher::HISTO_LENGTH src/ORBmatcher.cc:41 */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1968-1984): host scalar, no device involved */
int32_t orbfe_hamming(const uint8_t a[32], const uint8_t b[32]);

orbfe_status orbfe_matcher_create(int32_t device /* -1 = current */, orbfe_matcher **out);
void orbfe_matcher_destroy(orbfe_matcher *m);

/* BASELINE config 3 "brute-force Hamming match to previous frame" (SURVEY 8(a) M3): for every query
 * row the best / second-best distance over ALL train rows with the update idiom of
 * src/ORBmatcher.cc:280-289 (ties: lowest train index), accepted when best <= th and
 * (float)best < nnratio*(float)second, then the rotation-consistency histogram (:308-316, :338-360,
 * ComputeThreeMaxima :1912-1957) when check_ori != 0.  HOST buffers.
 *   match_q2t[nq] : train index or -1        best/second[nq] : may be NULL        *nmatches : kept */
orbfe_status orbfe_match_bf(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                            const float *q_angle, const float *t_angle, float nnratio, int32_t th,
                            int32_t check_ori, int32_t *match_q2t, int32_t *best, int32_t *second,
                            int32_t *nmatches);
void *orbfe_matcher_get_stream(orbfe_matcher *m);
/* all-pairs kernel behind the three orbfe_match_bf* calls: 0 (default) = exact int8 dot product on the matrix cores
 * (dot = 128 * (128 - hamming)), 1 = xor / popcount.  Results are identical; see DESIGN.md for the measured A/B. */
orbfe_status orbfe_matcher_set_bf_kernel(orbfe_matcher *m, int32_t kernel);
/* orbfe_search_by_projection(_chi2): 0 (default) = the whole search in ONE launch (k_proj_fused: per-query slabs, round 0 of the
 * relaxation decided while the candidates are written, the last workgroup finishes the rounds, results stored straight into
 * page-locked host memory), falling back to 1 = count -> scan -> fill -> resolve when a query has more than 512 candidates or
 * the tables exceed 64 KB of LDS.  Identical results. */
orbfe_status orbfe_matcher_set_projection_kernel(orbfe_matcher *m, int32_t kernel);
/* same, DEVICE buffers, enqueued on `stream` (NULL = HIP's default stream, see orbfe_extract_batch_device;
 * orbfe_matcher_get_stream(m) = the matcher's own stream), no synchronisation;
 * d_nmatches is one int32 in device memory */
orbfe_status orbfe_match_bf_device(orbfe_matcher *m, const uint8_t *d_q, int32_t nq, const uint8_t *d_t,
                                   int32_t nt, const float *d_q_angle, const float *d_t_angle, float nnratio,
                                   int32_t th, int32_t check_ori, int32_t *d_match_q2t, int32_t *d_best,
                                   int32_t *d_second, int32_t *d_nmatches, void *stream);
/* batched form of the above: pair p matches frame p (queries) against frame p+1's predecessor layout:
 *   queries  = d_desc + qframe[p]*cap*32 (nq = d_n[qframe[p]]),  train = d_desc + tframe[p]*cap*32
 * with angles taken from d_kps (orbfe_keypoint.angle).  Outputs are npairs x cap. Used by bench.py. */
orbfe_status orbfe_match_bf_frames_device(orbfe_matcher *m, const orbfe_keypoint *d_kps, const uint8_t *d_desc,
                                          const int32_t *d_n, int32_t cap, const int32_t *d_qframe,
                                          const int32_t *d_tframe, int32_t npairs, float nnratio, int32_t th,
                                          int32_t check_ori, int32_t *d_match_q2t /* npairs*cap */,
                                          int32_t *d_nmatches /* npairs */, void *stream);

/* the same with queries and train frames taken from two DIFFERENT output blocks of the same `cap` (e.g. the frames of this
 * batch against frames of the previous batch): d_qframe indexes (d_qkps, d_qdesc, d_qn), d_tframe indexes (d_tkps, d_tdesc, d_tn) */
orbfe_status orbfe_match_bf_blocks_device(orbfe_matcher *m, const orbfe_keypoint *d_qkps, const uint8_t *d_qdesc,
                                          const int32_t *d_qn, const orbfe_keypoint *d_tkps, const uint8_t *d_tdesc,
                                          const int32_t *d_tn, int32_t cap, const int32_t *d_qframe, const int32_t *d_tframe,
                                          int32_t npairs, float nnratio, int32_t th, int32_t check_ori,
                                          int32_t *d_match_q2t /* npairs*cap */, int32_t *d_nmatches /* npairs */, void *stream);

/* ORBmatcher::SearchByBoW (KeyFrame*, Frame&, ...) src/ORBmatcher.cc:217-363   [strict_lt = 0, validF = NULL]
 * ORBmatcher::SearchByBoW (KeyFrame*, KeyFrame*, ...) src/ORBmatcher.cc:665-812 [strict_lt = 1]
 * DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) is passed as CSR: node[nnodes] ascending,
 * off[nnodes+1], idx[off[nnodes]].  Each feature index must occur in at most one node (true for
 * FeatureVectors produced by DBoW2 transform()); otherwise ORBFE_ERR_ARG.
 *   validKF[nKF] : 1 where the KF feature has a good MapPoint (pMP && !pMP->isBad()), NULL = all
 *   validF[nF]   : same for the second keyframe (M2), NULL for a Frame (M1)
 *   matchF2KF[nF]: KF feature index whose MapPoint is assigned to F feature i, -1 = none
 * HOST buffers. */
orbfe_status orbfe_search_by_bow(orbfe_matcher *m, const uint8_t *descKF, int32_t nKF, const uint8_t *validKF,
                                 const float *angKF, const uint32_t *nodeKF, const uint32_t *offKF,
                                 const uint32_t *idxKF, int32_t nnodesKF, const uint8_t *descF, int32_t nF,
                                 const uint8_t *validF, const float *angF, const uint32_t *nodeF,
                                 const uint32_t *offF, const uint32_t *idxF, int32_t nnodesF, float nnratio,
                                 int32_t th_low, int32_t strict_lt, int32_t check_ori, int32_t *matchF2KF,
                                 int32_t *nmatches);

/* batched best / second-best over per-query candidate lists (CSR), the inner loop of the
 * SearchByProjection / SearchForTriangulation / SearchBySim3 / Fuse family (src/ORBmatcher.cc:63,378,827,
 * 1031,1198,1334,1578,1757): the pose/grid gating stays on the host, the Hamming work comes here.
 * best_idx[nq] = candidate (train row) with the minimum distance, first in list order on ties, -1 if
 * the list is empty; best/second initialised to 256.  HOST buffers. */
orbfe_status orbfe_hamming_csr(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                               const uint32_t *off, const uint32_t *cand, int32_t *best_idx, int32_t *best,
                               int32_t *second);
/* same, plus second_idx[nq] (may be NULL): the candidate that last set the runner-up distance in the reference's update
 * idiom -- what SearchByProjection(Frame&, vector<MapPoint*>&) keeps as bestLevel2 (src/ORBmatcher.cc:128-147): the
 * previous best when a new best arrives, else the first candidate below the runner-up; -1 if fewer than two */
orbfe_status orbfe_hamming_csr_ex(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                                  const uint32_t *off, const uint32_t *cand, int32_t *best_idx, int32_t *best,
                                  int32_t *second, int32_t *second_idx);
/* every distance of every list: dist[off[nq]] (0..256), for the one caller whose rule needs them all --
 * ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:523-651) skips a candidate that an EARLIER query holds at a
 * distance <= its own (:573), so its best / second-best depend on the matches made so far; the distances come from the
 * device, the in-order rule stays with the caller.  HOST buffers. */
orbfe_status orbfe_hamming_csr_all(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                                   const uint32_t *off, const uint32_t *cand, uint16_t *dist);
/* DEVICE buffers (queries / train rows e.g. straight out of an extractor output block), enqueued on `stream`, no
 * validation of the candidate indices (they must be < the number of train rows); d_second_idx may be NULL */
orbfe_status orbfe_hamming_csr_device(orbfe_matcher *m, const uint8_t *d_q, int32_t nq, const uint8_t *d_t,
                                      const uint32_t *d_off, const uint32_t *d_cand, int32_t *d_best_idx, int32_t *d_best,
                                      int32_t *d_second, int32_t *d_second_idx, void *stream);

/* the frame grid index that every projection-gated matcher walks.
 * Frame::AssignFeaturesToGrid (src/Frame.cc:319-334) + Frame::PosInGrid (:522-531): 64 x 48 cells over the undistorted
 * image bounds; keypoint i goes to cell (round((x-minx)*gw_inv), round((y-miny)*gh_inv)) if that is inside the grid.
 *   xy[n*2]            undistorted keypoint positions (mvKeysUn[i].pt)
 *   cell_off[64*48+1]  CSR offsets, cell c = ix*48 + iy (= mGrid[ix][iy]);  cell_idx[n]: keypoint indices, ascending per
 *                      cell (push_back order);  *n_in_grid = number of keypoints inside the grid.  HOST buffers. */
#define ORBFE_GRID_COLS 64 /* FRAME_GRID_COLS include/Frame.h:26 */
#define ORBFE_GRID_ROWS 48 /* FRAME_GRID_ROWS include/Frame.h:25 */
orbfe_status orbfe_assign_grid(orbfe_matcher *m, const float *xy, int32_t n, float minx, float miny, float gw_inv,
                               float gh_inv, uint32_t *cell_off, uint32_t *cell_idx, int32_t *n_in_grid);
/* The same index on the host, without a device or a matcher handle: for hosts that hold the keypoints but not the grid
 * (KeyFrame::mGrid is protected in the reference, include/KeyFrame.h:223 -- the matcher shim rebuilds it from the public
 * mvKeysUn with the values Frame::AssignFeaturesToGrid used: Frame::mnMinX / mnMinY, mfGridElement{Width,Height}Inv). */
orbfe_status orbfe_assign_grid_host(const float *xy, int32_t n, float minx, float miny, float gw_inv, float gh_inv,
                                    uint32_t *cell_off, uint32_t *cell_idx, int32_t *n_in_grid);
/* AssignFeaturesToGrid for every frame of an extractor output block, device-resident: frame f's keypoints are
 * d_kps[f * cap .. f * cap + d_n[f]) as orbfe_extract_batch_device wrote them (mvKeysUn == mvKeys: no lens distortion, as
 * for TUM fr3); d_cell_off [nframes][ORBFE_GRID_COLS * ORBFE_GRID_ROWS + 1], d_cell_idx [nframes][cap], d_n_in_grid
 * [nframes].  Enqueued on `stream`, no host synchronisation. */
orbfe_status orbfe_assign_grid_batch_device(orbfe_matcher *m, const orbfe_keypoint *d_kps, const int32_t *d_n, int32_t cap,
                                            int32_t nframes, float minx, float miny, float gw_inv, float gh_inv,
                                            uint32_t *d_cell_off, uint32_t *d_cell_idx, int32_t *d_n_in_grid, void *stream);

/* Frame::GetFeaturesInArea (src/Frame.cc:465-518) for a batch of nq queries (x, y, r, minLevel, maxLevel):
 *   qxyr[nq*3], qlevels[nq*2] (may be NULL = -1,-1);  octave[n] = mvKeysUn[i].octave
 *   off[nq+1], cand[cap]: per query the keypoint indices in the reference's iteration order (ix, iy, cell order).
 * Returns ORBFE_ERR_CAP (with off[nq] = required total) when cap is too small.  Feed (off, cand) to orbfe_hamming_csr
 * to get SearchByProjection's best / second-best per query.  HOST buffers. */
orbfe_status orbfe_features_in_area(orbfe_matcher *m, const float *xy, const int32_t *octave, int32_t n,
                                    const uint32_t *cell_off, const uint32_t *cell_idx, float minx, float miny,
                                    float gw_inv, float gh_inv, const float *qxyr, const int32_t *qlevels, int32_t nq,
                                    uint32_t *off, uint32_t *cand, int32_t cap);

/* GetFeaturesInArea on device buffers, for ONE frame of an extractor output block: d_kps = that frame's keypoint records
 * (x, y, octave are read in place), d_cell_off / d_cell_idx = its grid (orbfe_assign_grid_batch_device), d_qxyr[nq*3] /
 * d_qlevels[nq*2] (may be NULL) the queries; d_off[nq+1] and d_cand[cap] come out in the order orbfe_hamming_csr_device
 * consumes.  Enqueued on `stream`, nothing validated.  If the total exceeds cap nothing is written to d_cand and
 * d_off[nq] (the required size) tells so. */
orbfe_status orbfe_features_in_area_device(orbfe_matcher *m, const orbfe_keypoint *d_kps, const uint32_t *d_cell_off,
                                           const uint32_t *d_cell_idx, float minx, float miny, float gw_inv, float gh_inv,
                                           const float *d_qxyr, const int32_t *d_qlevels, int32_t nq, uint32_t *d_off,
                                           uint32_t *d_cand, int32_t cap, void *stream);

/* the projection-gated searches of the per-frame tracker --
 *   ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th)       src/ORBmatcher.cc:63-157
 *   ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)   src/ORBmatcher.cc:1578-1724
 *   (and the perfect/ overload that also returns the 2-D point pairs, perfect/src/ORBmatcher.cc:1727-1911).
 * The pose projection and its gates (isBad, mbTrackInView, depth sign, image bounds, forward / backward level window) stay
 * with the caller (shim/ORBmatcher_orbfe.cc does them on cv::Mat exactly as the reference); one query = one MapPoint that
 * passed them.  Per query: Frame::GetFeaturesInArea(u, v, r, min_level, max_level) on the frame's grid, the right-image gate
 * (candidate idx skipped when uRight[idx] > 0 && fabs(ur - uRight[idx]) > r -- both call sites compare against the very
 * expression they pass as the search radius, :114-119 / :1654-1660), best / second-best Hamming with the :128-140 idiom
 * over the candidates whose slot is free, acceptance: best <= th (th <= 255; TH_HIGH / ORBdist in the reference) and, with ratio_rule != 0, not (bestLevel == bestLevel2 &&
 * best > nnratio * second) (:143-146; ratio_rule 0 for the last-frame form, :1673).
 * Queries are NOT independent in the reference: an accepted query puts its MapPoint into F.mvpMapPoints[bestIdx], and a
 * slot holding a MapPoint with Observations() > 0 is skipped by every later candidate scan (:108-110 / :1647-1649).
 * blocked[nF] marks slots taken before the call; ORBFE_PROJ_CLAIMS marks a query whose MapPoint has Observations() > 0.
 * The result equals the reference's sequential loop (fixed point by relaxation).
 *   match[nq]: frame feature assigned to query i, -1 = none.  The caller replays, in query order,
 *              F.mvpMapPoints[match[i]] = pMP_i; nmatches++  (later queries may overwrite a slot whose point has no
 *              observations: the reference counts both), then its rotation histogram (:1683-1721) where it has one.
 *   best / second[nq] (may be NULL): bestDist / bestDist2 of the scan that decided the query (256 = none).
 * HOST buffers; xyF = mvKeysUn[i].pt, octF = mvKeysUn[i].octave, (cell_off, cell_idx) = Frame::mGrid as orbfe_assign_grid
 * lays it out; at most 15360 frame features. */
typedef struct orbfe_proj_query {
    float u, v, r;                   /* GetFeaturesInArea(u, v, r, min_level, max_level) */
    int32_t min_level, max_level;
    float ur;                        /* projected right-image coordinate (mTrackProjXR / u - mbf * invzc) */
    int32_t flags;                   /* ORBFE_PROJ_* */
    int32_t pad;
} orbfe_proj_query;
#define ORBFE_PROJ_CLAIMS 1          /* the query's MapPoint has Observations() > 0: its slot is skipped by later queries */
#define ORBFE_PROJ_RIGHT_GATE 2      /* apply the right-image gate (both Frame overloads do; the KeyFrame forms do not) */
#define ORBFE_PROJ_CHI2_GATE 4       /* Fuse (src/ORBmatcher.cc:1112-1139): skip a candidate whose reprojection error
                                      * e2 * inv_level_sigma2[octave] exceeds 7.8 (uRight[idx] >= 0: e2 = ex^2 + ey^2 + er^2,
                                      * er = ur - uRight[idx]) / 5.99 (monocular keypoint); needs the _chi2 entry point */
orbfe_status orbfe_search_by_projection(orbfe_matcher *m, const uint8_t *descF, const float *xyF, const int32_t *octF,
                                        int32_t nF, const uint32_t *cell_off, const uint32_t *cell_idx, float minx, float miny,
                                        float gw_inv, float gh_inv, const float *uRight /* nF or NULL */,
                                        const uint8_t *blocked /* nF or NULL */, const orbfe_proj_query *q,
                                        const uint8_t *qdesc /* nq x 32 */, int32_t nq, int32_t th, float nnratio,
                                        int32_t ratio_rule, int32_t *match, int32_t *best, int32_t *second);

/* the same search with the per-level table the ORBFE_PROJ_CHI2_GATE queries need (KeyFrame::mvInvLevelSigma2); the other
 * members of the family use it without claims: ORBmatcher::Fuse x2 (src/ORBmatcher.cc:1031, :1198) and SearchBySim3 (:1334)
 * pass flags without ORBFE_PROJ_CLAIMS, blocked = NULL, ratio_rule 0 and read match[i] = the best candidate if its distance is
 * <= th, so that GetFeaturesInArea runs on the device for them too */
orbfe_status orbfe_search_by_projection_chi2(orbfe_matcher *m, const uint8_t *descF, const float *xyF, const int32_t *octF,
                                             int32_t nF, const uint32_t *cell_off, const uint32_t *cell_idx, float minx, float miny,
                                             float gw_inv, float gh_inv, const float *uRight /* nF or NULL */,
                                             const uint8_t *blocked /* nF or NULL */, const float *inv_level_sigma2 /* nlevels or NULL */,
                                             int32_t nlevels, const orbfe_proj_query *q, const uint8_t *qdesc /* nq x 32 */, int32_t nq,
                                             int32_t th, float nnratio, int32_t ratio_rule, int32_t *match, int32_t *best,
                                             int32_t *second);

/* HOST-side geometry of the same family (csrc/orbfe_hostgeom.hip; no device, no handle), so that a host shim only flattens
 * its objects and replays decisions.  Arithmetic order = what the reference's cv::Mat expressions evaluate to: 3x3 * 3x1
 * products accumulate left to right in float, norms and dot products in double, nothing fused.
 *
 * orbfe_project_points: x_c = R * x_w + t (then x_c = R2 * x_c + t2 when R2 != NULL: SearchBySim3's two stages), the depth /
 * image / distance / viewing-angle gates and the projected coordinates of n map points -- src/ORBmatcher.cc:401-433 (Sim3
 * projection), :1060-1101 (Fuse), :1224-1255 (Fuse, Sim3), :1389-1424 (SearchBySim3), :1620-1642 (last frame), :1778-1803
 * (relocalisation).  ok[i] = 1 when point i passed every gate; u, v (and invz, dist, ur = u - bf * invz where non-NULL) are
 * written for those.  min_dist / max_dist NULL: no distance gate; with them dist = |x_w - Ow| (Ow != NULL) or |x_c|; normal
 * != NULL adds the viewing-angle gate PO . n < 0.5 * dist -> out.  R, R2 row-major 3x3. */
#define ORBFE_PJ_SKIP_NEG_DEPTH 1    /* z_c < 0 -> out (:411, :1068, :1234, :1396) */
#define ORBFE_PJ_SKIP_NEG_INVZ 2     /* 1 / z_c < 0 -> out (:1625) */
#define ORBFE_PJ_UV_CHAINED 4        /* u = fx * x_c * invz + cx (:1627, :1789) instead of u = fx * (x_c * invz) + cx */
#define ORBFE_PJ_BOUNDS_CLOSED 8     /* out iff u < minx || u > maxx (Frame statics, :1629) instead of KeyFrame::IsInImage */
orbfe_status orbfe_project_points(const float *R, const float *t, const float *R2, const float *t2, const float *Ow, float fx,
                                  float fy, float cx, float cy, float bf, float minx, float maxx, float miny, float maxy,
                                  int32_t flags, int32_t n, const float *world_pos, const float *normal, const float *min_dist,
                                  const float *max_dist, float *u, float *v, float *invz, float *dist, float *ur, uint8_t *ok);
/* the gating of SearchByProjection(Frame&, const vector<MapPoint*>&, th) (:63-93; Tracking::SearchLocalPoints): points with
 * mbTrackInView and !isBad() become queries (u, v, ur) = proj_uvr[i], r = RadiusByViewingCos(view_cos) [* th] *
 * scale_factors[level], levels [level - 1, level], right-image gate on, CLAIMS iff obs_gt0; q / src compacted, *nq their count */
orbfe_status orbfe_proj_queries_local_map(const float *scale_factors, int32_t n, const uint8_t *in_view, const uint8_t *bad,
                                          const int32_t *level, const float *view_cos, const float *proj_uvr, const uint8_t *obs_gt0,
                                          float th, orbfe_proj_query *q, int32_t *src, int32_t *nq);
/* the rotation-consistency check every matcher ends with: match i votes for bin round((angle_a[i] - angle_b[i], + 360 if
 * negative) * (1 / histo_len)) (:308-313), the three fullest bins stay (ComputeThreeMaxima :1912-1957: an earlier bin wins a
 * tie, the second / third go when under a tenth of the first); drop[i] = 1 for matches outside them.  A match whose bin is
 * not in [0, histo_len] (NaN, an angle outside [0, 360), or histo_len < 19 with ordinary angles) is what the reference asserts
 * against (:314): ORBFE_ERR_ARG, nothing written */
orbfe_status orbfe_rotation_consistency(const float *angle_a, const float *angle_b, int32_t n, int32_t histo_len, uint8_t *drop);
/* SearchForInitialization's in-order rule (:547-617) on the lists of orbfe_window_distances: query qi takes its closest
 * candidate among those no earlier query holds at a distance <= its own, accepted when best <= th and best < second * nnratio;
 * a later query may take a held feature over.  accepted[nq]: the feature a query was accepted with (-1: none) -- whether or
 * not it was taken over later; holder[n2]: the query that holds a feature at the end (-1: none). */
orbfe_status orbfe_initialization_resolve(const uint32_t *off, const uint32_t *ent, int32_t nq, int32_t n2, int32_t th, float nnratio,
                                          int32_t *accepted, int32_t *holder);

/* The window search and the distances alone, as lists: off[nq + 1], ent[off[nq]] with ent = feature | distance << 16 in the
 * reference's candidate order (GetFeaturesInArea's cell walk); flags / ur of the queries are ignored.  ORBFE_ERR_CAP (off[nq] =
 * the size needed) when cap is too small.  ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:523-651) runs its in-order
 * rule -- a candidate held by an earlier query at a smaller distance is skipped, :571-574 -- on these lists.  HOST buffers. */
orbfe_status orbfe_window_distances(orbfe_matcher *m, const uint8_t *descF, const float *xyF, const int32_t *octF, int32_t nF,
                                    const uint32_t *cell_off, const uint32_t *cell_idx, float minx, float miny, float gw_inv,
                                    float gh_inv, const orbfe_proj_query *q, const uint8_t *qdesc, int32_t nq, uint32_t *off,
                                    uint32_t *ent, int32_t cap);

/* the matching core of ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:827-1012,
 * LocalMapping::CreateNewMapPoints).  For every feature of keyframe 1 with elig1 (no MapPoint; stereo if bOnlyStereo) whose
 * vocabulary node also exists in keyframe 2: over that node's keyframe-2 features in FeatureVector order, the ones with elig2
 * (no MapPoint; stereo if bOnlyStereo -- the reference's loop never sets vbMatched2, so the keyframe-1 features are
 * independent), dist <= th_low, dist <= bestDist, farther than the epipole gate ((ex - x)^2 + (ey - y)^2 >= 100 * scale[octave],
 * applied only when both keypoints are monocular, :900-907) and inside CheckDistEpipolarLine (:175-196: F12 row-major,
 * dsqr < 3.84 * mvLevelSigma2[octave], the reference's float operation order) take over the best: the smallest distance, the
 * LAST in order on ties.  match12[n1] = keyframe-2 feature or -1; the caller applies its rotation histogram (:966-985) and
 * builds vMatchedPairs.  FeatureVectors as CSR (ascending node ids); HOST buffers. */
orbfe_status orbfe_search_for_triangulation(orbfe_matcher *m, const uint8_t *desc1, const float *xy1, const uint8_t *elig1,
                                            const uint8_t *stereo1, int32_t n1, const uint32_t *node1, const uint32_t *off1,
                                            const uint32_t *idx1, int32_t nn1, const uint8_t *desc2, const float *xy2,
                                            const int32_t *oct2, const uint8_t *elig2, const uint8_t *stereo2, int32_t n2,
                                            const uint32_t *node2, const uint32_t *off2, const uint32_t *idx2, int32_t nn2,
                                            const float F12[9], float ex, float ey, const float *scale_factors2,
                                            const float *level_sigma2_2, int32_t nlevels2, int32_t th_low, int32_t *match12);

/* Frame::ComputeStereoMatches (src/Frame.cc:642-846): row-band descriptor search in the right image,
 * 11 x 11 SAD refinement over 11 shifts on the two extractors' device-resident pyramids (mvImagePyramid of the LAST
 * call of `left` / `right`, frame 0), parabola fit, disparity -> depth, and the median-based outlier rejection.
 *   kpsL/descL/nL, kpsR/descR/nR  what those two calls returned (mvKeys / mDescriptors, mvKeysRight / mDescriptorsRight)
 *   mbf, mb                       baseline * fx and the baseline; the reference reads mb before assigning it (:682,
 *                                 undefined) -- here minZ = mb is an explicit argument
 *   uRight[nL], depth[nL]         mvuRight / mvDepth, -1 where no match.  HOST buffers. */
orbfe_status orbfe_stereo_matches(orbfe_matcher *m, orbfe_handle *left, orbfe_handle *right, const orbfe_keypoint *kpsL,
                                  const uint8_t *descL, int32_t nL, const orbfe_keypoint *kpsR, const uint8_t *descR,
                                  int32_t nR, float mbf, float mb, float *uRight, float *depth);

/* The same for a whole batch, without leaving the device: frame pair f = frame f of the LAST orbfe_extract_batch_device
 * call of `left` and of `right` (the stereo Frame constructor runs the two extractors side by side, src/Frame.cc:82-87),
 * their output blocks exactly as those calls wrote them -- d_kps* [nframes][cap], d_desc* [nframes][cap][32], d_n*
 * [nframes] -- and d_uRight / d_depth [nframes][cap] floats (slots >= the frame's count are left untouched).  Everything
 * is enqueued on `stream` (the stream the two extract calls ran on, or one ordered after it); d_gray of both calls must
 * still be alive.  No host synchronisation. */
orbfe_status orbfe_stereo_matches_batch_device(orbfe_matcher *m, orbfe_handle *left, orbfe_handle *right,
                                               const orbfe_keypoint *d_kpsL, const uint8_t *d_descL, const int32_t *d_nL,
                                               const orbfe_keypoint *d_kpsR, const uint8_t *d_descR, const int32_t *d_nR,
                                               int32_t cap, int32_t nframes, float mbf, float mb, float *d_uRight,
                                               float *d_depth, void *stream);

/* DBoW2 TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) as called by
 * Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:553, src/KeyFrame.cc:82; levelsup = 4).  DBoW2 is not vendored
 * by the reference; the algorithm is restated from the published one.  The tree is handed over as arrays:
 * children of node i = child_idx[child_off[i] .. child_off[i+1]) in stored order (node 0 = root; child ids must exceed
 * their parent's id), node_desc[nnodes*32], word_id / weight (TF-IDF) meaningful for leaves, L = tree depth. */
typedef struct orbfe_vocabulary orbfe_vocabulary;
orbfe_status orbfe_vocabulary_create(int32_t device, int32_t nnodes, const uint32_t *child_off, const uint32_t *child_idx,
                                     const uint8_t *node_desc, const uint32_t *word_id, const double *weight, int32_t L,
                                     orbfe_vocabulary **out);
void orbfe_vocabulary_destroy(orbfe_vocabulary *v);
/* desc[n*32] (n <= 8192).  Per feature (any may be NULL): f_word / f_node (-1 when the word's weight is 0 and the feature
 * is skipped) / f_weight.  BowVector: bow_id[<= n] ascending, bow_val L1-normalised doubles, *nbow.  FeatureVector as the
 * CSR orbfe_search_by_bow takes: fv_node[<= n] ascending, fv_off[*nfv + 1], fv_idx[<= n].  HOST buffers. */
orbfe_status orbfe_bow_transform(orbfe_matcher *m, const orbfe_vocabulary *v, const uint8_t *desc, int32_t n,
                                 int32_t levelsup, int32_t *f_word, int32_t *f_node, double *f_weight, uint32_t *bow_id,
                                 double *bow_val, int32_t *nbow, uint32_t *fv_node, uint32_t *fv_off, uint32_t *fv_idx,
                                 int32_t *nfv);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:284-345) for a batch of map points.
 *   pool[npool*32]      descriptors (rows of the observing keyframes' mDescriptors)
 *   off[npoints+1], idx map point p observes pool[idx[off[p] .. off[p+1])]  (at most 1024 observations per point)
 *   best_idx[npoints]   position inside the point's list of the descriptor with the least median Hamming distance to
 *                       the others (self distance 0 included, element (int)(0.5*(N-1)) of the sorted row; first on
 *                       ties); median[npoints] that median; both -1 for a point without observations.  HOST buffers. */
orbfe_status orbfe_distinctive_descriptors(orbfe_matcher *m, const uint8_t *pool, int32_t npool, const uint32_t *off,
                                           const uint32_t *idx, int32_t npoints, int32_t *best_idx, int32_t *median);

/* The same on device buffers (the pool is typically the all-gathered descriptor block of the keyframes: observation index
 * = keyframe * cap + slot): nothing is validated or copied, the launch goes to `stream`.  max_obs (1..1024) sizes the
 * kernel's LDS; a point with more observations than that gets -2 / -2 instead of a result. */
orbfe_status orbfe_distinctive_descriptors_device(orbfe_matcher *m, const uint8_t *d_pool, const uint32_t *d_off,
                                                  const uint32_t *d_idx, int32_t npoints, int32_t max_obs,
                                                  int32_t *d_best_idx, int32_t *d_median, void *stream);

/* ---- the same chain DEVICE-RESIDENT and BATCHED: extractor output block -> BoW -> SearchByBoW, one stream, no host
 * round trip (Frame::ComputeBoW src/Frame.cc:546-555 / KeyFrame::ComputeBoW src/KeyFrame.cc:74-84, then
 * ORBmatcher::SearchByBoW src/ORBmatcher.cc:217-363 / :665-812 for a batch of (KeyFrame, Frame) pairs).
 *
 * orbfe_bow_transform_batch_device: frame b owns `cap` slots of every array (cap + 1 of d_fv_off, 4 of d_counts):
 *   in : d_desc [B][cap][32], d_n [B]               (what orbfe_extract_batch_device wrote; cap <= 8192)
 *   out: d_f_word / d_f_node [B][cap] (-1 = no word / padding), d_f_weight [B][cap] doubles,
 *        d_bow_id / d_bow_val [B][cap]  BowVector, ascending ids, L1-normalised,
 *        d_fv_node [B][cap], d_fv_off [B][cap+1], d_fv_idx [B][cap]  FeatureVector CSR,
 *        d_counts [B][4] = {nbow, nfv, number of indexed features, 0}
 * orbfe_search_by_bow_batch_device: pair p = (KeyFrame d_kf[p], Frame d_f[p]), frame indices into the same blocks;
 *   d_valid [B][cap] (1 = good MapPoint) or NULL = all; kf_kf = 0: SearchByBoW(KeyFrame*, Frame&) (`<= th_low`, d_valid
 *   applies to the KeyFrame side only), kf_kf = 1: SearchByBoW(KeyFrame*, KeyFrame*) (`< th_low`, both sides);
 *   d_match [P][cap]: KF feature index assigned to F feature i (-1 none; for kf_kf invert to get vpMatches12),
 *   d_nmatches [P].  Both calls only enqueue on `stream` (NULL = HIP's default stream). */
orbfe_status orbfe_bow_transform_batch_device(orbfe_matcher *m, const orbfe_vocabulary *v, const uint8_t *d_desc,
                                              const int32_t *d_n, int32_t nframes, int32_t cap, int32_t levelsup,
                                              int32_t *d_f_word, int32_t *d_f_node, double *d_f_weight,
                                              uint32_t *d_bow_id, double *d_bow_val, uint32_t *d_fv_node,
                                              uint32_t *d_fv_off, uint32_t *d_fv_idx, int32_t *d_counts, void *stream);
orbfe_status orbfe_search_by_bow_batch_device(orbfe_matcher *m, const orbfe_keypoint *d_kps, const uint8_t *d_desc,
                                              int32_t cap, const uint8_t *d_valid, const uint32_t *d_fv_node,
                                              const uint32_t *d_fv_off, const uint32_t *d_fv_idx, const int32_t *d_counts,
                                              const int32_t *d_kf, const int32_t *d_f, int32_t npairs, float nnratio,
                                              int32_t th_low, int32_t kf_kf, int32_t check_ori, int32_t *d_match,
                                              int32_t *d_nmatches, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Sequence pipeline: the frame loop of the reference's drivers (perfect/Examples/RGB-D/rgbd_tum.cc:77-119: every frame of
 * the sequence through the Frame constructor = ORBextractor::operator(), then matched against its predecessor -- BASELINE
 * config 3) for a device-resident SEQUENCE in one call.
 *
 * A pipeline owns `npipes` pipes; a pipe = one extractor handle + one matcher handle + one stream.  A call cuts its
 * nframes into sub-batches of p->max_batch frames (the last one may be shorter) and runs sub-batch j on pipe j mod npipes,
 * so that the VALU-bound FAST pass of one sub-batch shares the chip with the HBM / LDS-bound stages of its neighbours
 * (one pipe 273 k, three pipes 302 k frames/s at 1024-frame sub-batches).  Frame k is matched against frame
 * k - 1 as orbfe_match_bf does (best <= th, ratio, rotation histogram) ACROSS sub-batch boundaries; frame 0 of a call is
 * matched against the last frame of the previous call when ORBFE_PIPE_CONTINUE is set (the pipeline keeps its own copy of
 * that frame, the caller may re-use its buffers), otherwise -- and on the first call or after
 * orbfe_pipeline_reset_sequence -- it has no predecessor: its match row is -1, its count 0.
 *   d_gray / d_kps / d_desc / cap / d_n_out   exactly as orbfe_extract_batch_device, for all nframes
 *   d_match [nframes][cap], d_nmatches [nframes]   row k = matches of frame k into frame k - 1 (train index or -1);
 *                                                  d_match == NULL: extract only (BASELINE config 2)
 * The call only enqueues.  The pipes start behind what `stream` holds at the time of the call; unless ORBFE_PIPE_NO_JOIN
 * is set, `stream` is made to wait for all pipes before the call returns, so that anything enqueued on it afterwards sees
 * the results.  With ORBFE_PIPE_NO_JOIN consecutive calls run back to back without draining the chip in between (a
 * throughput loop whose results are consumed later): call orbfe_pipeline_join(pl, stream) or orbfe_pipeline_synchronize(pl)
 * before touching the outputs.  Output blocks re-used by a later call are protected inside the pipeline: every sub-batch records the address ranges
 * it writes, and a later sub-batch waits for the extraction and the matchers of every recorded range that overlaps its own (by
 * address, not by position in the call: shifted base pointers and other call sizes are ordered too).
 * A pipeline is used by one thread at a time.  orbfe_pipeline_extractor / _matcher give the pipes' handles for the per-handle
 * settings (orbfe_set_fast_mode, orbfe_set_profiling, orbfe_matcher_set_bf_kernel) and taps.
 * ------------------------------------------------------------------------------------------- */
typedef struct orbfe_pipeline orbfe_pipeline;
#define ORBFE_PIPE_CONTINUE 1 /* frame 0 continues the sequence of the previous call */
#define ORBFE_PIPE_NO_JOIN 2  /* do not order `stream` behind the pipes at the end of the call */
orbfe_status orbfe_pipeline_create(const orbfe_params *p /* max_batch = frames per sub-batch */, int32_t npipes, orbfe_pipeline **out);
void orbfe_pipeline_destroy(orbfe_pipeline *pl);
int32_t orbfe_pipeline_pipes(const orbfe_pipeline *pl);
int32_t orbfe_pipeline_capacity(const orbfe_pipeline *pl);   /* = orbfe_keypoint_capacity of its handles */
int32_t orbfe_pipeline_sub_batch(const orbfe_pipeline *pl);
orbfe_handle *orbfe_pipeline_extractor(orbfe_pipeline *pl, int32_t pipe);
orbfe_matcher *orbfe_pipeline_matcher(orbfe_pipeline *pl, int32_t pipe);
orbfe_status orbfe_pipeline_extract_match_device(orbfe_pipeline *pl, const uint8_t *d_gray, int32_t nframes, int32_t w, int32_t ht,
                                                 int32_t stride, size_t frame_stride, orbfe_keypoint *d_kps, uint8_t *d_desc,
                                                 int32_t cap, int32_t *d_n_out, int32_t *d_match, int32_t *d_nmatches,
                                                 float nnratio, int32_t th, int32_t check_ori, int32_t flags, void *stream);
/* The same with HOST buffers -- the frame loop as the reference's drivers run it (images from the host, results to the host;
 * BASELINE config 3 as SURVEY 8(d) words it): grays[i] = frame i (w x ht, row pitch stride), results into kps [nframes][cap],
 * desc [nframes][cap][32], n_out [nframes] and, with match != NULL, match [nframes][cap] / nmatches [nframes]; cap >=
 * orbfe_pipeline_capacity().  Chunks of one sub-batch: the H2D copy of chunk c + 1, the pipes of chunk c and the D2H copies of
 * chunk c - 1 overlap (three device buffer sets, two copy streams); consecutive chunks take turns on the pipes.  Page-locked
 * frames / result arrays (hipHostMalloc, hipHostRegister, torch pin_memory) are copied asynchronously as they are; pageable
 * memory works, more slowly.  Blocking: returns when the results are on the host.  ORBFE_PIPE_CONTINUE as above. */
orbfe_status orbfe_pipeline_extract_match(orbfe_pipeline *pl, const uint8_t *const *grays, int32_t nframes, int32_t w, int32_t ht,
                                          int32_t stride, orbfe_keypoint *kps, uint8_t *desc, int32_t cap, int32_t *n_out,
                                          int32_t *match, int32_t *nmatches, float nnratio, int32_t th, int32_t check_ori,
                                          int32_t flags);
/* Pipes the host entry point deals its chunks to (default 1, at most orbfe_pipeline_pipes()).  The host path is bound by the
 * PCIe link, not by the kernels: its chunks should FINISH in order (first in, first out) so that their results leave while
 * the next chunk arrives -- several chunks sharing the chip finish together and stall the copies (measured 172 k frames/s
 * with 1, 156 k with 3, 141 k with 12 on a 57 GB/s link). */
orbfe_status orbfe_pipeline_set_host_pipes(orbfe_pipeline *pl, int32_t n);
/* `stream` waits (at stream level) for everything the pipes hold */
orbfe_status orbfe_pipeline_join(orbfe_pipeline *pl, void *stream);
/* the host waits for the pipes */
orbfe_status orbfe_pipeline_synchronize(orbfe_pipeline *pl);
/* the next call starts a new sequence even with ORBFE_PIPE_CONTINUE */
orbfe_status orbfe_pipeline_reset_sequence(orbfe_pipeline *pl);
/* OR of orbfe_get_overflow over the pipes' extractors (waits for them) */
orbfe_status orbfe_pipeline_get_overflow(orbfe_pipeline *pl, int32_t *flags);

/* ---------------------------------------------------------------------------------------------
 * Batched keyframe mode over several devices (north star: "shards independent frames across the 8 GPUs of one
 * node with an RCCL all-gather of descriptors over xGMI", host stays C++).
 *
 * A batch of nframes independent frames is cut into contiguous shards, shard r on rank r.  orbfe_group_shard_range is the
 * SINGLE source of truth for the cut: every rank gets nframes / world frames and the first nframes % world ranks one more
 * (10 frames over 4 ranks: 3, 3, 2, 2 -- not ceil-sized shards with a short last one); a caller that shards the frames
 * itself (orbfe_group_extract_shard_device) must use it.  Every rank owns three padded blocks for the WHOLE batch -- counts
 * [F], keypoints [F][cap], descriptors [F][cap][32], F = world * shard with shard = ceil(max_batch / world) slots per rank;
 * rank r's frames sit at block indices r * shard + (frame - lo_r) (orbfe_group_block_index; the unused slots of a slice are
 * zero) -- it extracts its shard straight into its own slice, and ONE in-place all-gather per block (ncclAllGather, RCCL)
 * leaves the whole batch on every rank.  The consumer of the gather, what KeyFrameDatabase /
 * LoopClosing do serially per candidate keyframe (src/LoopClosing.cc:312-342), is orbfe_group_match: frames of the own shard
 * against candidate frames anywhere in the gathered set.
 *
 * Two ways to form a group; every other call is the same:
 *   orbfe_group_create_local  ONE process drives ndevices devices (ncclCommInitAll): the shape of a C++ SLAM process
 *   orbfe_group_create_rank   one process per device: rank 0 makes an id (orbfe_group_unique_id), hands the 128 bytes to the
 *                             other ranks by any means it has, every rank calls this (ncclCommInitRank)
 * p->max_batch = the largest GLOBAL batch; p->device is ignored.  RCCL is loaded at run time (the copy already mapped into
 * the process if there is one, else librccl.so.1 / $ORBFE_RCCL_LIB); without it the create calls fail with ORBFE_ERR_STATE.
 * A group is used by one thread at a time.  Calls only enqueue work (compute stream + communication stream per member,
 * ordered by events); orbfe_group_synchronize / _get_frame / _match wait.
 *
 * Transport of the exchange step.  ORBFE_GROUP_RCCL (default): ncclAllGather.  ORBFE_GROUP_COPY (local groups only,
 * orbfe_group_create_local_ex): every member pulls the other members' slices with
 * hipMemcpyAsync / hipMemcpyPeerAsync on its communication stream behind the same events -- same bytes at the same offsets.
 * It needs no communicator, so `devices` may name one device several times: several members on ONE GPU, which is how the
 * multi-member paths are tested on a one-GPU box (RCCL refuses two ranks on one device).
 * ------------------------------------------------------------------------------------------- */
typedef struct orbfe_group orbfe_group;
enum { ORBFE_GROUP_RCCL = 0, ORBFE_GROUP_COPY = 1 };
void orbfe_group_shard_range(int32_t nframes, int32_t rank, int32_t world, int32_t *lo, int32_t *hi);
/* pure index arithmetic of the layout above (no group, no device): the rank that owns `frame`, and the frame's slot in the
 * blocks of a group with `shard` slots per rank; -1 on a bad argument */
int32_t orbfe_group_owner_rank(int32_t nframes, int32_t world, int32_t frame);
int32_t orbfe_group_block_index_of(int32_t nframes, int32_t world, int32_t shard, int32_t frame);
orbfe_status orbfe_group_unique_id(uint8_t id[128]);
orbfe_status orbfe_group_create_local(const orbfe_params *p, const int32_t *devices, int32_t ndevices, orbfe_group **out);
orbfe_status orbfe_group_create_local_ex(const orbfe_params *p, const int32_t *devices, int32_t ndevices, int32_t transport,
                                         orbfe_group **out);
orbfe_status orbfe_group_create_rank(const orbfe_params *p, int32_t device, int32_t rank, int32_t world, const uint8_t id[128],
                                     orbfe_group **out);
void orbfe_group_destroy(orbfe_group *g);
int32_t orbfe_group_world(const orbfe_group *g);
int32_t orbfe_group_members(const orbfe_group *g);       /* members THIS process drives (world for a local group, 1 for a rank group) */
int32_t orbfe_group_transport(const orbfe_group *g);
int32_t orbfe_group_capacity(const orbfe_group *g);      /* cap: keypoint slots per frame of the blocks */
int32_t orbfe_group_frames_padded(const orbfe_group *g); /* F = world * shard */
int32_t orbfe_group_block_index(const orbfe_group *g, int32_t nframes, int32_t frame);
/* HOST frames of the global batch (grays[i] = frame i, w x ht, row pitch stride): every member of this process copies and
 * extracts its own shard (a rank group reads only grays[lo .. hi) of its rank) */
orbfe_status orbfe_group_extract_batch(orbfe_group *g, const uint8_t *const *grays, int32_t nframes, int32_t w, int32_t ht,
                                       int32_t stride);
/* DEVICE frames: d_gray = the shard of member `member` (its frames only, on its device), nframes_global = the size of the
 * whole batch (the shard bounds follow from it) */
orbfe_status orbfe_group_extract_shard_device(orbfe_group *g, int32_t member, const uint8_t *d_gray, int32_t nframes_global, int32_t w,
                                              int32_t ht, int32_t stride, size_t frame_stride);
/* the one exchange step: in-place ncclAllGather of the three blocks on the members' communication streams, behind the
 * extraction; later calls on the compute streams are ordered behind it */
orbfe_status orbfe_group_allgather(orbfe_group *g);
orbfe_status orbfe_group_synchronize(orbfe_group *g);
/* device pointers of member `member`'s blocks and its compute stream (hipStream_t), for consumers of the gathered batch */
orbfe_status orbfe_group_blocks(orbfe_group *g, int32_t member, int32_t **d_n, orbfe_keypoint **d_kps, uint8_t **d_desc,
                                void **compute_stream);
/* one frame of the (gathered) batch of the last extract call to the host */
orbfe_status orbfe_group_get_frame(orbfe_group *g, int32_t frame, orbfe_keypoint *kps, uint8_t *desc, int32_t cap, int32_t *n_out);
/* the same out of the blocks of member `member` (orbfe_group_get_frame reads member 0) */
orbfe_status orbfe_group_get_frame_from(orbfe_group *g, int32_t member, int32_t frame, orbfe_keypoint *kps, uint8_t *desc, int32_t cap,
                                        int32_t *n_out);
/* the whole count block [F] of member `member` to the host (slots outside the shards' frames are 0) */
orbfe_status orbfe_group_get_counts(orbfe_group *g, int32_t member, int32_t *n_out);
/* pair p: brute-force match (as orbfe_match_bf: best <= th, ratio, rotation histogram) of frame qframe[p] -- a frame of a
 * shard of THIS process -- against frame tframe[p], any frame of the batch; match [npairs][cap] (train index or -1, slots
 * >= the query frame's count are -1), nmatches [npairs].  HOST arrays; call after orbfe_group_allgather. */
orbfe_status orbfe_group_match(orbfe_group *g, const int32_t *qframe, const int32_t *tframe, int32_t npairs, float nnratio, int32_t th,
                               int32_t check_ori, int32_t *match, int32_t *nmatches);
/* the same on one member with everything on its device: pairs as BLOCK indices (orbfe_group_block_index), results
 * [npairs][cap] / [npairs] in device memory, enqueued on the member's compute stream */
orbfe_status orbfe_group_match_device(orbfe_group *g, int32_t member, const int32_t *d_qblock, const int32_t *d_tblock, int32_t npairs,
                                      float nnratio, int32_t th, int32_t check_ori, int32_t *d_match, int32_t *d_nmatches);

/* ---------------------------------------------------------------------------------------------
 * Formats: the on-disk keyframe records of Map::Save / Map::Load and the ORB vocabulary files
 * ------------------------------------------------------------------------------------------- */
/* Keyframe block of Map::Save (perfect/src/Map.cc:330-381 _WriteKeyFrame, read back by _ReadKeyFrame :143-187):
 *   u64 mnId | f64 mTimeStamp | f32 t_cw[3] | f32 q_cw[4] (x y z w) | i32 N |
 *   N x { f32 pt.x pt.y size angle response | i32 octave | u8 descriptor[32] | u64 map-point index (ULONG_MAX = none) }
 * = 48 + 64 * N bytes; class_id is not stored (the reader leaves -1).  HOST buffers; *written / *consumed = block size. */
size_t orbfe_mapio_keyframe_bytes(int32_t n);
orbfe_status orbfe_mapio_write_keyframe(uint8_t *dst, size_t cap, uint64_t id, double timestamp, const float t_cw[3],
                                        const float q_cw[4], const orbfe_keypoint *kps, const uint8_t *desc,
                                        const uint64_t *mp_index /* NULL = none */, int32_t n, size_t *written);
orbfe_status orbfe_mapio_read_keyframe(const uint8_t *src, size_t len, uint64_t *id, double *timestamp, float t_cw[3],
                                       float q_cw[4], orbfe_keypoint *kps, uint8_t *desc, uint64_t *mp_index, int32_t cap,
                                       int32_t *n, size_t *consumed);
/* the 64-byte feature records straight from an extractor output block in HBM (also what an all-gathered block is turned
 * into before it is written): frame b -> d_out + b*cap*64, slots >= d_n[b] zero; d_mp_index [nframes][cap] or NULL */
orbfe_status orbfe_mapio_pack_records_device(const orbfe_keypoint *d_kps, const uint8_t *d_desc, const int32_t *d_n,
                                             const uint64_t *d_mp_index, int32_t nframes, int32_t cap, uint8_t *d_out,
                                             void *stream);

/* The caller's colour -> gray step before operator() (Tracking::GrabImageRGBD, src/Tracking.cc:339-353) for DEVICE-resident
 * 3-channel interleaved frames: gray = (w0*c0 + 9617*c1 + w2*c2 + 8192) >> 14, (w0, w2) = (4899, 1868) when rgb_flag != 0
 * (cvtColor(CV_RGB2GRAY) on the memory order given -- the reference's path for cv::imread's BGR with Camera.RGB = 1),
 * (1868, 4899) otherwise (CV_BGR2GRAY).  Enqueued on `stream`. */
orbfe_status orbfe_interleaved_to_gray_device(const uint8_t *d_src, int32_t nframes, int32_t w, int32_t h, int32_t src_stride,
                                              size_t src_frame_stride, int32_t rgb_flag, uint8_t *d_gray,
                                              int32_t gray_stride, size_t gray_frame_stride, void *stream);

/* ORB vocabulary files as the reference loads them (src/System.cc:123-129; tool/text2binary.cc converts one into the
 * other).  DBoW2 is not vendored by the reference; the two layouts are ORB-SLAM2's TemplatedVocabulary
 * loadFromTextFile / saveToBinaryFile, restated (csrc/orbfe_io.hip).  Host only: no device is touched before
 * orbfe_vocabulary_create_from_file. */
typedef struct orbfe_vocfile orbfe_vocfile;
orbfe_status orbfe_vocfile_load(const char *path /* ORBvoc.txt or ORBvoc.bin */, orbfe_vocfile **out);
void orbfe_vocfile_free(orbfe_vocfile *v);
orbfe_status orbfe_vocfile_info(const orbfe_vocfile *v, int32_t *k, int32_t *L, int32_t *nnodes, int32_t *nwords,
                                int32_t *scoring, int32_t *weighting);
/* pointers into the parsed file (valid until orbfe_vocfile_free); any may be NULL */
orbfe_status orbfe_vocfile_arrays(const orbfe_vocfile *v, const uint32_t **child_off, const uint32_t **child_idx,
                                  const uint8_t **node_desc, const uint32_t **word_id, const double **weight,
                                  const uint32_t **parent, const uint8_t **is_leaf);
orbfe_status orbfe_vocfile_save_binary(const orbfe_vocfile *v, const char *path);
orbfe_status orbfe_vocabulary_create_from_file(int32_t device, const orbfe_vocfile *v, orbfe_vocabulary **out);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H */
