// orbfe_common.h -- internal definitions shared by the HIP translation units of liborbfe.so.
// Product code: never includes anything from oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "orbfe.h"

#define ORBFE_MAX_LEVELS 16
#define ORBFE_MAX_ROOTS 8   // quadtree roots of a level = round(width / height) of its detection window (:545)
#define ORBFE_EDGE 19        // EDGE_THRESHOLD   (reference src/ORBextractor.cc:54)
#define ORBFE_HALF_PATCH 15  // HALF_PATCH_SIZE  (:53)
#define ORBFE_PATCH 31       // PATCH_SIZE       (:52)
#define ORBFE_MINB 16        // minBorderX/Y = EDGE_THRESHOLD-3 (:780)
#define ORBFE_NK_STRIDE 32   // ints between per-(frame, level) key counters: one 128-B line each (atomic targets)
#define ORBFE_LDS_MAX (160 * 1024)  // LDS of one gfx950 CU: the dynamic-LDS attribute of every kernel is set to this
#define ORBFE_TILE_MAX 72    // FAST cell tile edge upper bound (cell+6 <= 66 when nCols == 1)

// ---- thread-local error text ---------------------------------------------------------------------
void orbfe_set_error(const char *fmt, ...);

#define ORBFE_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            orbfe_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,    \
                            __LINE__);                                                           \
            return e__ == hipErrorOutOfMemory ? ORBFE_ERR_NOMEM : ORBFE_ERR_HIP;                 \
        }                                                                                        \
    } while (0)

// ---- device-visible plan -------------------------------------------------------------------------
// One entry per pyramid level.  Level 0 is read in place from the caller's frame (pitch = stride);
// levels >= 1 live in the handle's pyramid block at byte offset `off` of each frame's slice.
struct OrbLevel {
    int32_t w, h;          // level size                      (src/ORBextractor.cc:1121-1122)
    int32_t pitch;         // row pitch in bytes inside the pyramid / blurred blocks
    int32_t off;           // byte offset inside one frame's pyramid (and blurred) slice
    int32_t ncols, nrows;  // FAST grid                        (:792-796)
    int32_t wcell, hcell;
    int32_t cell0, ncells; // this level's cells in the frame's cell table (skipped cells removed)
    int32_t ncc;           // cell columns after the skip rule (cells form an ncells/ncc x ncc grid)
    int32_t nfeat;         // mnFeaturesPerLevel[l]            (:426-439)
    int32_t nini;          // quadtree roots                   (:545)
    float hx;              // root width                       (:547)
    int32_t key_off;       // first key slot of this level in a frame's key scratch
    int32_t key_cap;       // worst-case number of FAST candidates of this level
    int32_t sel_off;       // first slot of this level in a frame's selected-keypoint scratch
    int32_t sel_cap;
    int32_t xtab, ytab;    // offsets into the resize tables (entries), level >= 1
    float scale;           // mvScaleFactor[l]
    float patch_size;      // (float)(int)(PATCH_SIZE * scale)  (:846)
    int32_t root_x[ORBFE_MAX_ROOTS + 1];  // root box x boundaries, nini+1 entries
    int32_t ix1, iy1;      // end of the union of the cells' detectable interiors ([19,ix1) x [19,iy1))
    // two pyramid levels per launch (k_pyr_walk2): this level (B) is produced tile by tile from level l-1 and level l+1 (C)
    // from the tile while it is in LDS.  Tile = p2_gx column groups (4 px) x p2_gy row runs; tiles overlap by one column
    // group and one row; p2_cxs / p2_cys: first column / row of level l+1 each tile column / row owns (int32 tables inside
    // the tap-table block, offsets in OrbTab entries).  p2_tx == 0: not fused.
    int32_t p2_gx, p2_gy, p2_tx, p2_ty, p2_cxs, p2_cys;
};

// one FAST cell = one cv::FAST call of the reference (:798-838)
struct OrbCell {
    uint16_t level;
    uint16_t x0, y0;  // tile origin in level coordinates (iniX, iniY)
    uint16_t tw, th;  // tile size (maxX-iniX, maxY-iniY)
    uint16_t ox, oy;  // j*wCell, i*hCell: added to tile-relative keypoints (:831-832)
    uint16_t pad;
};

// bilinear resize table entry: source index + the two 11-bit coefficients
struct OrbTab {  // one cv::resize tap of one axis: dword 0 = the coefficient pair, dword 1 = the source index
    int16_t c0;  // (1-f)*2048 rounded half-even
    int16_t c1;  // f*2048
    int16_t s;   // sx / sy (un-clamped for y)
    int16_t pad;
};

// one lane of a streaming image pass: a 4-pixel column walked down `nrows` rows
struct OrbLane {
    uint16_t x, ys, nrows;
    uint16_t flags;  // bit 0: halo (computes, does not output); bits 8..15: level
};

// the resize job a blur lane carries in the fused blur + pyramid pass: one destination dword of the next level
struct OrbLaneR {
    uint16_t dj;      // destination dword (4 pixels) of level l + 1
    uint16_t d0, nd;  // destination rows [d0, d0 + nd): those whose upper source row lies in the lane's row block; nd = 0: none
    uint16_t pad;
};

struct OrbPlan {
    int32_t nlevels;
    int32_t w, h;              // level-0 size this plan was built for
    int32_t ncells;            // cells per frame (all levels)
    int32_t cell_cap;          // key slots per cell
    int32_t max_ncells;        // largest FAST cell count of a level (quadtree cell-flag bitmap)
    int32_t keys_per_frame;    // key scratch entries per frame
    int32_t sel_per_frame;     // selected-keypoint scratch entries per frame
    int32_t node_cap;          // quadtree node capacity (multiple of 64)
    int32_t max_nini;          // largest number of quadtree roots over the levels
    int32_t ini_th, min_th;
    int32_t blur_rounding;
    int32_t dbg;               // developer knob (ORBFE_DEBUG env), 0 in production; 50 = quadtree streaming passes only
    int32_t nfwaves;           // FAST waves per frame (64 lane descriptors each)
    int32_t nfwaves_c;         // the same for the lane list of the lane-compacting form (every piece of a strip closed by halo lanes)
    int32_t fast_cellrows;     // the dense lane list is made of whole cell rows, one run of rows per wave: k_fast_map_u (else k_fast_map)
    int32_t fwave_off[ORBFE_MAX_LEVELS + 1];  // first FAST wave of every level (waves are single-level, levels in order)
    int32_t nbwaves;           // blur waves per frame (64 lane descriptors each)
    int32_t bwave_off[ORBFE_MAX_LEVELS + 1];  // first blur wave of every level (the lanes of a level are contiguous)
    int32_t blur_split;        // the blur lane list also holds resize waves (flag bit 2): k_blur_pyr<., 1>; the plain k_blur7 launch skips them
    int64_t pyr_frame_bytes;   // bytes of one frame's pyramid slice (levels 1..n-1; level 0 kept too when owned)
    OrbLevel lv[ORBFE_MAX_LEVELS];
    // k_blur7, border waves: the horizontal taps of a lane with BORDER_REFLECT_101 folded into them.  [level][lane type][3 * j + d] =
    // weight bytes of window dword d for the lane's output pixel j; lane type 0: no reflected column (window = x - 4 .. x + 7), 1: x = 0,
    // 2: 4 < w - x <= 8, 3: w - x <= 4 (window pulled back to end at the row's last pixel)
    uint32_t blur_wt[ORBFE_MAX_LEVELS][4][12];
};

// packed FAST candidate: x (12 bit) | y (12 bit) << 12 | response (8 bit) << 24, detection-window coords
__host__ __device__ inline uint32_t orb_pack_key(int x, int y, int r)
{
    return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)r << 24);
}
__host__ __device__ inline int orb_key_x(uint32_t k) { return (int)(k & 0xFFFu); }
__host__ __device__ inline int orb_key_y(uint32_t k) { return (int)((k >> 12) & 0xFFFu); }
__host__ __device__ inline int orb_key_r(uint32_t k) { return (int)(k >> 24); }

static inline int orb_align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline int64_t orb_align_up64(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
