/*
 * orb_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the ORB front-end of Ewenwan/ORB_SLAM2_SSD_Semantic:
 *   ORBextractor  (/root/reference/src/ORBextractor.cc, include/ORBextractor.h)
 *   ORBmatcher    Hamming core (/root/reference/src/ORBmatcher.cc)
 * plus the OpenCV-3.2 generic-path arithmetic those files call (cv::resize
 * INTER_LINEAR, copyMakeBorder, cv::FAST, GaussianBlur, fastAtan2, cvRound),
 * restated from the published algorithms because OpenCV is a non-vendored,
 * un-pinned third-party dependency.
 *
 * PARITY STATUS: pinned to code compiled from the reference for everything that IS the reference's code;
 * "restated, unpinned by real OpenCV" for the five OpenCV algorithms.
 *   - oracle/_ref/libref_orb.so (recipe: oracle/refbuild/Makefile) is the UNMODIFIED reference
 *     src/ORBextractor.cc and src/ORBmatcher.cc compiled against a cv stub.  tests/test_ref_pin.py checks
 *     oracle == _ref bit for bit, stage by stage, on the golden cases + 200 seeded frames + 300 quadtree sets +
 *     480 SearchByBoW cases: constructor tables, pyramid orchestration, the cell loop and threshold fallback,
 *     DistributeOctTree / DivideNode, IC_Angle, computeOrbDescriptor, rescale + order, DescriptorDistance,
 *     SearchByBoW x2, ComputeThreeMaxima.  tests/golden/orb_golden.npz is generated from _ref.  The (f) rows --
 *     Frame grid + GetFeaturesInArea, ComputeStereoMatches, ComputeDistinctiveDescriptors -- are checked against the
 *     reference's function bodies cut verbatim out of Frame.cc / MapPoint.cc at build time (oracle/refbuild/slice.py).
 *   - cv::resize, copyMakeBorder, cv::FAST, GaussianBlur, fastAtan2 are NOT in /root/reference (OpenCV is a
 *     non-vendored, un-pinned dependency and is not installed): inside _ref they are THIS file's restatements
 *     of the OpenCV 3.2 generic path, pinned only by known-answer tables and definition-level twins
 *     (tests/test_oracle_kat.py, tests/twins.py) and anchored on independent third-party code where that exists
 *     (tests/test_thirdparty.py): scikit-image's own FAST-9 classifies every pixel as this file's score does, its
 *     rBRIEF table / umax / 749-pixel mask equal ours, numpy.pad == the REFLECT_101 border, scipy's int64 correlation
 *     with the q8 taps == the blur byte for byte, float64 half-pixel bilinear (scipy) within one grey level of the
 *     resize, numpy.arctan2 within 0.3 degrees of fastAtan2.  For those five stages parity remains "unpinned by real
 *     OpenCV" (FAST's score value / NMS and the resize's internal truncations have no third-party anchor).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The product (orb_slam2_ssd_semantic_amd/csrc)
 * never does.
 *
 * Two places where the real reference binary is itself not reproducible and
 * this oracle DEFINES the contract (documented in DESIGN.md):
 *   1. DistributeOctTree sorts pair<int,ExtractorNode*> (ORBextractor.cc:686):
 *      equal-size nodes are ordered by heap address.  Contract here: ties keep
 *      creation order (stable), iterated from the back as :687 does.
 *   2. cos/sin at ORBextractor.cc:97 resolve to glibc cosf/sinf (CPU/glibc
 *      dependent last-bit).  Contract here: orc_sincos(), a fixed fp64
 *      algorithm rounded once to fp32.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint field order (T1 in SURVEY 8(a)); 28 bytes. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

/* FAST / quadtree candidate: integer pixel position stored as float like cv::KeyPoint. */
typedef struct {
    float x, y, response;
} orc_cand;

typedef struct orc_extractor orc_extractor;

/* ---- E0: constructor tables (ORBextractor.cc:399-466) ---- */
orc_extractor *orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void orc_destroy(orc_extractor *e);
int orc_nlevels(const orc_extractor *e);
void orc_get_scales(const orc_extractor *e, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2);
void orc_get_features_per_level(const orc_extractor *e, int *out);
void orc_get_umax(int out[16]);
const signed char *orc_get_pattern(void); /* 1024 x int8 */
/* level sizes for a w x h input (ORBextractor.cc:1121-1122) */
void orc_level_sizes(const orc_extractor *e, int w, int h, int *lw, int *lh);
/* FAST cell grid of one level (ORBextractor.cc:780-796): returns 0 if the level is too small */
int orc_cell_grid(int lw, int lh, int *ncols, int *nrows, int *wcell, int *hcell);

/* ---- E2: cv::resize INTER_LINEAR 8UC1 and copyMakeBorder REFLECT_101 (9.2) ---- */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                          int dstride);
void orc_resize_tables(int ssize, int dsize, int is_x, int *ofs, int16_t *coef /* 2*dsize */);
void orc_copy_make_border101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride,
                             int border);

/* ---- E3a: cv::FAST(img, kps, threshold, nonmax=true), FAST-9/16 ---- */
/* returns number of keypoints written (raster order), -1 if cap is too small */
int orc_fast9(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax, orc_cand *out, int cap);
/* threshold-independent corner score map: score(x,y) = max(A_dark, A_bright) - 1 clamped to [0,255]
 * on the detectable interior [3,w-3)x[3,h-3), 0 elsewhere (kernel twin). */
void orc_fast_score_map(const uint8_t *img, int w, int h, int stride, uint8_t *score, int score_stride);

/* ---- E4: DistributeOctTree (ORBextractor.cc:540-765) ---- */
typedef struct {
    int iterations;   /* breadth-first passes executed */
    int phaseb_passes;/* largest-first passes executed */
    int tie_breaks;   /* number of equal-size adjacent pairs crossed by the >=N break (tie-sensitive) */
} orc_octree_stats;
int orc_distribute_octtree(const orc_cand *in, int n, int minx, int maxx, int miny, int maxy, int N,
                           orc_cand *out, int cap, orc_octree_stats *st);

/* ---- E6: IC_Angle (ORBextractor.cc:59-88) + cv::fastAtan2 ---- */
float orc_fast_atan2(float y, float x);
void orc_ic_moments(const uint8_t *img, int stride, int x, int y, int *m10, int *m01);
float orc_ic_angle(const uint8_t *img, int stride, int x, int y);

/* ---- E7: GaussianBlur 7x7 sigma 2 REFLECT_101, 8-bit fixed-point kernel ----
 * mode 0 = canonical integer formula (half-up); mode 1 = emulate the x86 SSE2 column kernel
 * (half-to-even on columns x < (w & ~3)).  *ties (optional) counts exact-half pixels. */
void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int mode,
                        long *ties);

/* ---- E8: steered BRIEF (ORBextractor.cc:92-131) ---- */
void orc_sincos(float angle_deg, float *cos_a, float *sin_b); /* canonical (float)cos/(float)sin of angle*pi/180 */
void orc_sincos_rad(float angle_rad, float *cos_a, float *sin_b); /* the same sequence entered after the *factorPI step */
void orc_descriptor(const uint8_t *blurred, int stride, int x, int y, float angle_deg, uint8_t desc[32]);

/* ---- E1: operator() (ORBextractor.cc:1052-1114) ---- */
/* returns 0 ok, -1 bad args/size, -2 cap too small. Empty image (w==0||h==0) -> 0 with *n_out untouched. */
int orc_extract(orc_extractor *e, const uint8_t *gray, int w, int h, int stride, orc_keypoint *kps,
                uint8_t *desc, int cap, int *n_out);
/* stage taps of the last orc_extract call (pointers owned by e, valid until next call) */
const uint8_t *orc_tap_level(const orc_extractor *e, int level, int *w, int *h, int *stride);
const uint8_t *orc_tap_blurred(const orc_extractor *e, int level, int *w, int *h, int *stride);
const orc_cand *orc_tap_candidates(const orc_extractor *e, int level, int *n);
const orc_cand *orc_tap_selected(const orc_extractor *e, int level, int *n);
long orc_tap_blur_ties(const orc_extractor *e);
int orc_tap_octree_tie_breaks(const orc_extractor *e);
void orc_set_blur_mode(orc_extractor *e, int mode);

/* ---- M0: DescriptorDistance (ORBmatcher.cc:1968-1984) ---- */
int orc_hamming(const uint8_t a[32], const uint8_t b[32]);

/* ---- M5: ComputeThreeMaxima (ORBmatcher.cc:1912-1957) on bin counts ---- */
void orc_three_maxima(const int *counts, int L, int *ind1, int *ind2, int *ind3);
/* rotation bin (ORBmatcher.cc:308-313) */
int orc_rot_bin(float angle1, float angle2);

/* ---- M3: brute-force best/2nd-best + ratio + rotation histogram ---- */
int orc_match_bf(const uint8_t *q, int nq, const uint8_t *t, int nt, const float *q_angle, const float *t_angle,
                 float nnratio, int th, int check_ori, int32_t *match_q2t, int32_t *best, int32_t *second,
                 int *nmatches);

/* ---- M1/M2: SearchByBoW (ORBmatcher.cc:217-363, 665-812) over CSR feature vectors ---- */
int orc_search_by_bow(const uint8_t *descKF, int nKF, const uint8_t *validKF, const float *angKF,
                      const uint32_t *nodeKF, const uint32_t *offKF, const uint32_t *idxKF, int nnodesKF,
                      const uint8_t *descF, int nF, const uint8_t *validF, const float *angF,
                      const uint32_t *nodeF, const uint32_t *offF, const uint32_t *idxF, int nnodesF,
                      float nnratio, int th_low, int strict_lt, int check_ori, int32_t *matchF2KF,
                      int *nmatches);

/* ---- 8(f).1: CSR batched Hamming best/2nd-best (core of the SearchByProjection family) ---- */
int orc_hamming_csr(const uint8_t *q, int nq, const uint8_t *t, int nt, const uint32_t *off /* nq+1 */,
                    const uint32_t *cand, int32_t *best_idx, int32_t *best, int32_t *second);
int orc_hamming_csr2(const uint8_t *q, int nq, const uint8_t *t, int nt, const uint32_t *off, const uint32_t *cand,
                     int32_t *best_idx, int32_t *best, int32_t *second, int32_t *second_idx /* may be NULL */);

/* ---- 8(f).2: Frame::AssignFeaturesToGrid (src/Frame.cc:319-334) + PosInGrid (:522-531) ----
 * 64 x 48 grid; cell c = ix*48 + iy (mGrid[ix][iy]); cell_off has 64*48+1 entries, cell_idx ascending per cell. */
#define ORC_GRID_COLS 64
#define ORC_GRID_ROWS 48
int orc_assign_grid(const float *xy /* n x 2 */, int n, float minx, float miny, float gw_inv, float gh_inv,
                    uint32_t *cell_off, uint32_t *cell_idx);
/* ---- 8(f).2: Frame::GetFeaturesInArea (src/Frame.cc:465-518) for one query; returns the count, -1 if cap too small */
int orc_features_in_area(const float *xy, const int32_t *octave, const uint32_t *cell_off, const uint32_t *cell_idx,
                         float minx, float miny, float gw_inv, float gh_inv, float x, float y, float r, int min_level,
                         int max_level, uint32_t *out, int cap);

/* ---- M4: the projection-gated searches (src/ORBmatcher.cc:63-157 local map, :1578-1724 last frame; perfect/ :1727-1911) ----
 * One query = one MapPoint that survived the reference's gating: GetFeaturesInArea(u, v, r, min_level, max_level) on the
 * frame, the right-image gate (candidate idx skipped when uRight[idx] > 0 and fabs(ur - uRight[idx]) > r, :114-119 /
 * :1654-1660; both call sites compare against the very expression they pass as the search radius), best / second-best
 * with the :128-140 idiom over the candidates whose slot is free.  A slot is taken when the frame feature already holds a
 * MapPoint with Observations() > 0 (blocked[], :108-110 / :1647-1649) or when an EARLIER query of this call put such a
 * point there -- the loop assigns F.mvpMapPoints[bestIdx] = pMP as it goes, so queries are not independent.
 * flags: bit 0 = the query's MapPoint has Observations() > 0 (its assignment takes the slot), bit 1 = right-image gate on.
 * ratio_rule 1 = reject when bestLevel == bestLevel2 && bestDist > nnratio * bestDist2 (:143-146); 0 = none (:1673).
 * match[q] = frame feature the query is assigned to (the host replays F.mvpMapPoints[match] = pMP; nmatches++ in query
 * order), -1 = none. */
typedef struct orc_proj_query {
    float u, v, r;
    int32_t min_level, max_level;
    float ur;
    int32_t flags;
    int32_t pad;
} orc_proj_query;
int orc_search_by_projection(const uint8_t *descF, const float *xyF, const int32_t *octF, int nF, const uint32_t *cell_off,
                             const uint32_t *cell_idx, float minx, float miny, float gw_inv, float gh_inv,
                             const float *uRight /* nF or NULL */, const uint8_t *blocked /* nF or NULL */,
                             const orc_proj_query *q, const uint8_t *qdesc, int nq, int th, float nnratio, int ratio_rule,
                             int32_t *match, int32_t *best, int32_t *second);
int orc_search_by_projection_chi2(const uint8_t *descF, const float *xyF, const int32_t *octF, int nF, const uint32_t *cell_off,
                                  const uint32_t *cell_idx, float minx, float miny, float gw_inv, float gh_inv,
                                  const float *uRight, const uint8_t *blocked, const float *inv_sigma2, int nlevels,
                                  const orc_proj_query *q, const uint8_t *qdesc, int nq, int th, float nnratio, int ratio_rule,
                                  int32_t *match, int32_t *best, int32_t *second);
/* The reference's host steps in front of that core, restated with the float operation order of the code as compiled
 * against oracle/refbuild's cv stub (cv::Mat products accumulate in float, k ascending; real OpenCV's gemm accumulates
 * floats in double -- gemm_double = 1 -- the product's shim uses whatever cv::Mat it is linked with):
 * last-frame variant (:1593-1640): valid[i] = 0 when last feature i yields no query. */
int orc_proj_queries_last_frame(const float Tcw_cur[16], const float Tcw_last[16], float fx, float fy, float cx, float cy,
                                float mbf, float mb, float minx, float maxx, float miny, float maxy,
                                const float *scale_factors, int nlast, const uint8_t *has_mp, const uint8_t *outlier,
                                const float *world_pos /* nlast x 3 */, const int32_t *octave_last, const uint8_t *mp_obs_gt0,
                                float th, int mono, int gemm_double, orc_proj_query *q, uint8_t *valid);
/* local-map variant (:67-92): track_in_view / bad / scale level / view cos / proj x, y, xr per MapPoint */
int orc_proj_queries_local_map(const float *scale_factors, int nmp, const uint8_t *in_view, const uint8_t *bad,
                               const int32_t *scale_level, const float *view_cos, const float *proj_xyr /* nmp x 3 */,
                               const uint8_t *mp_obs_gt0, float th, orc_proj_query *q, uint8_t *valid);

/* ---- M4: SearchForTriangulation (src/ORBmatcher.cc:827-1012), the matching core ----
 * For every feature idx1 of keyframe 1 that is eligible (no MapPoint, stereo if bOnlyStereo) and whose vocabulary node also
 * exists in keyframe 2: over that node's features of keyframe 2, in FeatureVector order, the eligible ones (elig2: no MapPoint,
 * stereo if bOnlyStereo -- vbMatched2 is never set by the reference's loop) with dist <= TH_LOW and dist <= bestDist that lie
 * farther than the epipole gate (:906-911, only when both keypoints are monocular) and pass CheckDistEpipolarLine (:175-196)
 * take over best (so: the smallest distance, LAST in order on ties).  match12[n1] = idx2 or -1, before the rotation check.
 * node1 / off1 / idx1, node2 / off2 / idx2: the two FeatureVectors as CSR (ascending node ids). */
int orc_search_for_triangulation(const uint8_t *desc1, const float *xy1, const uint8_t *elig1, const uint8_t *stereo1, int n1,
                                 const uint32_t *node1, const uint32_t *off1, const uint32_t *idx1, int nn1,
                                 const uint8_t *desc2, const float *xy2, const int32_t *oct2, const uint8_t *elig2,
                                 const uint8_t *stereo2, int n2, const uint32_t *node2, const uint32_t *off2,
                                 const uint32_t *idx2, int nn2, const float F12[9], float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int th_low, int32_t *match12);

/* ---- 8(f).4: MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:284-345) for a batch of map points ----
 * point p observes descriptors pool[idx[off[p] .. off[p+1])]; best_idx[p] = position (inside the point's list) of the
 * descriptor with the least median distance to the others (first on ties), median[p] that median; -1 / -1 if empty. */
int orc_distinctive(const uint8_t *pool, int npool, const uint32_t *off, const uint32_t *idx, int npoints,
                    int32_t *best_idx, int32_t *median);

/* ---- 8(f).2b: Frame::ComputeStereoMatches (src/Frame.cc:642-846) ----
 * eL / eR: extractors whose LAST orc_extract call saw the left / right image (their pyramids are read);
 * kps / desc: what those calls returned.  uRight[nL], depth[nL] = mvuRight / mvDepth (-1 where no match);
 * sad[nL] (optional) = the SAD distance kept for the outlier filter, -1 where none.
 * The reference reads `mb` before it is assigned (:682, undefined); here the caller passes it (minZ = mb). */
int orc_stereo_matches(const orc_extractor *eL, const orc_extractor *eR, const orc_keypoint *kpsL, const uint8_t *descL,
                       int nL, const orc_keypoint *kpsR, const uint8_t *descR, int nR, float mbf, float mb,
                       float *uRight, float *depth, int32_t *sad);

/* ---- 8(f).3: DBoW2 TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) ----
 * DBoW2 is a third-party dependency the reference does not vendor (perfect/Thirdparty/DBoW2 holds a readme only;
 * ORB-SLAM2 ships its own modified copy, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h).  Restated from the published
 * algorithm as used at src/Frame.cc:553 / src/KeyFrame.cc:82 (levelsup = 4, ORBvoc: k = 10, L = 6, TF-IDF, L1):
 * every feature descends the tree taking the child with the smallest Hamming distance (first on ties); word weight 0
 * -> feature skipped; BowVector = per word the sum of the weights in feature order, then L1-normalised in ascending
 * word order (doubles); FeatureVector = per node at level L - levelsup the feature indices in feature order.
 * Tree as arrays: children of node i are child_idx[child_off[i] .. child_off[i+1]) (node 0 = root), node_desc
 * [nnodes][32], word_id / weight meaningful for leaves.  Outputs: per feature word / node (-1 if skipped) / weight;
 * bow_id/bow_val[<= n] ascending; fv_node[<= n] ascending, fv_off[nfv+1], fv_idx[<= n]. */
int orc_bow_transform(int nnodes, const uint32_t *child_off, const uint32_t *child_idx, const uint8_t *node_desc,
                      const uint32_t *word_id, const double *weight, int L, int levelsup, const uint8_t *desc, int n,
                      int32_t *f_word, int32_t *f_node, double *f_weight, uint32_t *bow_id, double *bow_val, int *nbow,
                      uint32_t *fv_node, uint32_t *fv_off, uint32_t *fv_idx, int *nfv);

#ifdef __cplusplus
}
#endif
#endif
