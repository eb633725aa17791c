// orbfe_kernels.h -- launcher interface between the host API (orbfe_api.hip) and the kernels.
#pragma once
#include "orbfe_common.h"

// launch-shape options of a handle (orbfe_set_option); 0 = the built-in choice
struct OrbOpts {
    int32_t pw_rows;   // destination rows per lane run of the pyramid kernels (2..ORBFE_PW_ROWS)
    int32_t pyr_fuse;  // 1: two pyramid levels per launch (k_pyr_walk2; developer builds)
    int32_t qt[3];     // threads per workgroup of the quadtree's three level groups
};

// everything one batched extractor call needs on the device
struct OrbLaunch {
    OrbOpts opts;
    const OrbPlan *h_plan;  // host copy
    const OrbPlan *d_plan;  // device copy
    const OrbTab *d_tabs;
    const OrbLane *d_flanes;
    const OrbLane *d_flanes_c;   // lane list of k_fast_map_c
    const OrbLane *d_blanes;
    const OrbLaneR *d_blanesR;   // the resize job of every blur lane (fused blur + pyramid pass)
    int32_t nframes;
    // input frames (level 0, read in place)
    const uint8_t *d_gray;
    int64_t gray_fstride;
    int32_t gray_pitch;
    // handle-owned blocks, one slice per frame
    uint8_t *d_pyr;
    uint8_t *d_blur;
    int64_t pyr_fstride;
    uint2 *d_skeys;      // unordered NMS survivors {key, ord} per level (k_fast_map)
    int32_t *d_scount;
    uint32_t *d_cflag;   // [B][nlevels][cf_words] bit = the FAST cell has a survivor above iniTh (k_fast_map -> k_octree)
    int32_t cf_words;
    uint16_t *d_knode;
    int16_t *d_qtbox;      // [B][nlevels][qtbox_stride] node boxes of deep quadtrees (global scratch)
    int32_t qtbox_stride;  // int16 elements per (frame, level)
    char *d_qtnodes;       // [B][nlevels][qtnodes_stride] node arrays of quadtrees too large for the LDS (else null)
    int64_t qtnodes_stride;
    uint32_t *d_sel;
    int32_t *d_nsel;
    int32_t *d_nkeys;
    // outputs
    orbfe_keypoint *d_kps;
    uint8_t *d_desc;
    int32_t cap;
    int32_t *d_n_out;
    // sticky overflow word: bit 0 a level's survivor list exceeded key_cap, bit 1 a level's selection exceeded sel_cap,
    // bit 2 a frame's keypoints exceeded `cap`
    int32_t *d_ovf;
    // FAST variant: 1 = wave-uniform shortcuts for sparse-corner frames; d_fstat (optional) counts their effect
    int32_t fast_sparse;
    unsigned long long *d_fstat;
};

// FAST(l) + resize(l -> l + 1) in one launch per level for the levels below `nfused`, the remaining pyramid levels and one
// FAST launch over the remaining waves after them (replaces orbk_launch_pyramid + orbk_launch_fast); spread: see k_fast_pyr
hipError_t orbk_launch_fast_pyr(const OrbLaunch &a, int nfused, int spread, hipStream_t st);
hipError_t orbk_launch_fast_levels(const OrbLaunch &a, int l0, int l1, int clear, hipStream_t st);
hipError_t orbk_upload_constants(const int *umax16);
size_t orbk_octree_lds_bytes(int node_cap, int max_nini, int w, int h, int ncells);
size_t orbk_octree_box_bytes(int node_cap);
size_t orbk_octree_node_bytes(int node_cap);  // global scratch per (frame, level) when the node arrays do not fit the LDS
hipError_t orbk_prepare_octree(int node_cap, int max_nini, int w, int h, int ncells);
size_t orbk_pyramid_lds_bytes(int dh);
#define ORBFE_PW_ROWS 16  // destination rows per lane run of the pyramid kernels (= PW_ROWS)
size_t orbk_pyramid2_lds_bytes(int gx, int gy);  // dynamic LDS of the two-level pyramid kernel for a tile of gx column groups x gy runs  // dynamic LDS of the pyramid kernel for a destination level of dh rows
hipError_t orbk_launch_pyramid(const OrbLaunch &a, hipStream_t st);
hipError_t orbk_launch_fast(const OrbLaunch &a, hipStream_t st);
hipError_t orbk_launch_octree(const OrbLaunch &a, hipStream_t st);
hipError_t orbk_launch_blur(const OrbLaunch &a, hipStream_t st);
// blur of every level and the pyramid in one chained pass (replaces orbk_launch_pyramid + orbk_launch_blur)
hipError_t orbk_launch_blur_pyr(const OrbLaunch &a, hipStream_t st);
hipError_t orbk_launch_describe(const OrbLaunch &a, hipStream_t st);

// device view of the pyramid the handle built in its last call (orbfe_api.hip), for kernels outside the extractor
struct OrbPyrView {
    int32_t nlevels, device;
    const uint8_t *ptr[ORBFE_MAX_LEVELS];
    int32_t pitch[ORBFE_MAX_LEVELS], w[ORBFE_MAX_LEVELS], h[ORBFE_MAX_LEVELS];
    float scale[ORBFE_MAX_LEVELS], inv_scale[ORBFE_MAX_LEVELS];
    int64_t fstride[ORBFE_MAX_LEVELS];  // bytes from a level of one frame of the batch to the same level of the next
    int32_t nframes;                    // frames in the handle's last batch
};
// all levels of one frame with their REFLECT_101 frame of ORBFE_EDGE pixels, packed at off[l] (off[nlevels] = total bytes)
hipError_t orbk_launch_pad_pyramid(const OrbPyrView &v, const uint32_t *off, uint8_t *d_out, hipStream_t st);
struct orbfe_handle;
// fills `v` for frame `frame` of the last batch and waits for the handle's own stream; ORBFE_ERR_STATE before any call
int32_t orbfe_internal_pyramid_view(orbfe_handle *h, int frame, OrbPyrView *v);
int32_t orbfe_internal_order_after_last_call(orbfe_handle *h, void *stream);
