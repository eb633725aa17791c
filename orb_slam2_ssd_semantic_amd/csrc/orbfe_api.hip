// orbfe_api.hip -- host side of the extractor C-ABI (include/orbfe.h): constructor tables, per-size plan,
// device buffers, stream/event plumbing.  All pixel work happens in orbfe_kernels.hip; there is no CPU path.
//
// Reference behaviour restated here (paths relative to /root/reference):
//   constructor tables            src/ORBextractor.cc:399-466
//   level sizes / pyramid layout  src/ORBextractor.cc:1117-1145
//   FAST cell grid + skip rules   src/ORBextractor.cc:771-816
//   quadtree roots                src/ORBextractor.cc:545-564
//   cv::resize coefficient tables OpenCV 3.2 imgwarp.cpp
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>

#include "orbfe_common.h"
#include "orbfe_kernels.h"

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
static thread_local char t_err[512] = "";

void orbfe_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *orbfe_last_error(void) { return t_err; }
extern "C" int32_t orbfe_version(void) { return ORBFE_VERSION; }

extern "C" const char *orbfe_strerror(orbfe_status s)
{
    switch (s) {
    case ORBFE_OK: return "ok";
    case ORBFE_ERR_ARG: return "invalid argument";
    case ORBFE_ERR_SIZE: return "image size outside the planned range or too small for the pyramid grid";
    case ORBFE_ERR_CAP: return "keypoint capacity too small";
    case ORBFE_ERR_HIP: return "HIP runtime error";
    case ORBFE_ERR_NOMEM: return "out of memory";
    case ORBFE_ERR_NODEVICE: return "no usable HIP device (this library has no CPU path)";
    case ORBFE_ERR_STATE: return "call not valid in the current state";
    default: return "unknown error";
    }
}

extern "C" int32_t orbfe_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------------------------------
// device buffer that only grows
// ---------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            bytes = 0;
            if (e != hipSuccess) return e;
        }
        need = (need + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// ---------------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------------
#define ORBFE_PROF_RING 64
// rows a FAST / blur wave walks down.  Measured on MI355X (256 x 640x480): 24..48 rows are equally fast and ~5 % faster
// than 64+ -- the 6..8 warm-up rows of a block are nearly free, while shorter waves balance the CUs better.
#define ORBFE_ROWS_PER_WAVE 40
// event marks of one profiled call: 0 start, 1 pyramid done, 2 FAST done, 3 quadtree done, 4 describe start, 5 end (launch
// stream); 6 / 7 around the blur (on whichever stream it ran)
#define ORBFE_EV_N 8
// auto FAST mode: above this share of pixel pairs passing the necessary test the dense form is the cheaper one.  Measured per
// 1024 frames of 640x480 (tools/compact_ab.py, profiles/r05_compact_ab.json; dense / lane-compacting, ms): pass rate 0.84 (S)
// 1.47 / 2.14; 0.43 1.40 / 1.62; 0.38 1.42 / 1.55; 0.29 1.36 / 1.39; 0.18 (S_tum) 1.33 / 1.12; 0.076 1.26 / 0.92;
// 0.02 1.22 / 0.74 -- break-even near 0.27
#define ORBFE_AUTO_DENSE_RATE 0.25
// ... and a launch that does not fill the GPU is bound by its longest wave, not by issue slots, and the dense form has the shorter
// wave (FAST stage of ONE 640x480 frame: 18 us dense / 21 us compacting on S_tum, 20 / 26 on S; 8 frames: 29 / 34 and 31 / 44 --
// tools/fast_mode_latency.py): below this many wave row steps per call (about 29 frames of 640x480 with 8 levels: 4 890 each) auto is dense
#define ORBFE_AUTO_MIN_ROW_STEPS 140000
#define ORBFE_AUTO_HOLD_MIN 16    // dense calls after a probe above the rate; doubles with every such probe in a row ...
#define ORBFE_AUTO_HOLD_MAX 256   // ... up to this (a probe call on corner-saturated frames costs +45 % of its FAST stage)
#define ORBFE_AUTO_PROBE_EVERY 8  // compacting calls between two looks at the pass rate

struct orbfe_handle {
    orbfe_params prm;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;      // false: the stream belongs to a pipeline (orbfe_internal_create_on_stream); no host copy streams then
    bool own_side = true;        // false: the side stream (blur) is one the pipeline shares among its pipes
    // constructor tables (src/ORBextractor.cc:404-439)
    float scale[ORBFE_MAX_LEVELS], inv_scale[ORBFE_MAX_LEVELS], sigma2[ORBFE_MAX_LEVELS], inv_sigma2[ORBFE_MAX_LEVELS];
    int feat[ORBFE_MAX_LEVELS];
    // plan for the current frame size
    OrbPlan plan;
    bool plan_valid = false;
    std::vector<OrbCell> cells;
    std::vector<OrbTab> tabs;
    DevBuf d_plan, d_tabs, d_flanes, d_flanes_c, d_blanes, d_blanesR;
    // per-batch blocks
    DevBuf d_pyr, d_blur, d_skeys, d_scount, d_knode, d_qtbox, d_qtnodes, d_sel, d_nsel, d_nkeys, d_pad;
    // sticky overflow word + FAST sparse-variant statistics: [0] int32 overflow bits, [2..7] 3 x uint64 counters
    DevBuf d_misc;
    int fast_mode = 3;            // 0 dense, 1 sparse shortcuts, 2 lane-compacting, 3 auto (default): 2 or 0 by batch size and observed pass rate (orbfe_set_fast_mode)
    bool fast_stats = false;
    // auto mode: the lane-compacting kernel reports {row steps, batches, parked pairs} of a sample of its waves; the counters are
    // copied to pinned host memory behind the kernel and looked at -- without waiting -- by a later call
    PinBuf h_auto;                // 3 x uint64
    hipEvent_t ev_auto = nullptr;
    bool auto_pending = false;
    int auto_dense_left = 0;      // calls still to run dense before the pass rate is probed again
    int auto_hold = ORBFE_AUTO_HOLD_MIN;   // length of the next dense run
    int auto_since = 0;           // compacting calls since the last probe
    int auto_form = 2;            // the form the last probe chose (before the first answer: compacting, the probe's own form)
    uint64_t auto_last[3] = {0, 0, 0};
    int64_t fast_row_steps = 0;
    // The most recent batched call: its stream (only compared, never dereferenced: the caller may have destroyed it) and an
    // event recorded behind its last launch.  All calls of a handle share the scratch blocks, so a call on another stream
    // waits for that event first, and whoever needs the results on the host (taps, mvImagePyramid, re-planning, the
    // overflow word, destroy) synchronises the event, not the stream.
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t ev_last = nullptr;
    // host-API staging
    // two sets, so that the H2D of chunk i+1, the kernels of chunk i and the D2H of chunk i-1 overlap
    DevBuf d_stage[2], d_okps[2], d_odesc[2], d_on[2];
    PinBuf h_stage[2], h_okps[2], h_odesc[2], h_on[2], h_ovf;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_cmp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    // last call (for taps / mvImagePyramid)
    const uint8_t *last_gray = nullptr;
    int64_t last_gray_fstride = 0;
    int32_t last_gray_pitch = 0;
    int32_t last_nframes = 0;
    // profiling: ring of event sets so a timed region of many asynchronous calls can be averaged afterwards
    bool profiling = false;
    hipEvent_t ev[ORBFE_PROF_RING][ORBFE_EV_N];
    int prof_calls = 0;  // calls recorded since profiling was (re-)enabled
    bool ev_ok = false;
    // blur depends on the pyramid only, the quadtree on FAST only: the blur runs on a side stream next to the
    // latency-bound quadtree (overlap 2), next to FAST + quadtree (1), or in line (0).  -1 = by batch size: 2 for
    // batches that fill the chip (>= 128 frames: 3.71 -> 3.62 ms per 1024 frames, the HBM-bound blur fills the
    // quadtree's idle VALU / memory slots; next to the VALU-bound FAST pass it gains nothing), 0 for small ones
    int overlap = -1;
    int fuse_fast_pyr = 0;   // 1 / 2: FAST(l) + resize(l -> l + 1) in one launch per level (ORBFE_FUSE_FAST_PYR; 2 = the two kinds of
                             // workgroups dealt out proportionally over the grid, 1 = resize workgroups first)
    int fuse_fast_pyr_levels = ORBFE_MAX_LEVELS;   // levels fused that way; the rest: plain resizes + one FAST launch
    int fuse_blur_pyr = 0;   // 1: blur + pyramid in one chained pass over the levels (ORBFE_FUSE_BLUR_PYR)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;   // the side-stream FAST of ORBFE_OPT_FUSE_FAST_PYR = 3 (developer builds)
    // tuning options (orbfe_set_option; 0 / -1 = built-in choice).  The plan-shaping ones invalidate the plan.
    int opt_rows = 0, opt_rows_fast = 0, opt_rows_blur = 0;
    int opt_blur_pieces = 1, opt_blur_updown = 1, opt_debug = 0;
    OrbOpts kopts = {0, 0, {0, 0, 0}};
    // ORBFE_OPT_REUSE_IDENTICAL_INPUT (orbfe_extract only): the frame of the last single-frame host call is still in the pinned
    // staging block h_stage[0] and its results in h_okps[0]; a call that brings the same pixels gets those results back without
    // touching the GPU.  reuse_valid is dropped by every other use of the handle (run_batch) and by every option change.
    int opt_reuse = 0;
    bool reuse_valid = false, last_reused = false;
    int reuse_w = 0, reuse_h = 0, reuse_cap = 0;
    int64_t reuse_hits = 0;
};

// waits until the last batched call of the handle has finished, on whichever stream it ran
static hipError_t wait_last_call(orbfe_handle *h)
{
    if (!h->last_stream_valid) return hipSuccess;
    return hipEventSynchronize(h->ev_last);
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }  // cvRound: half-to-even

static void host_umax(int umax[16])
{
    // src/ORBextractor.cc:449-465
    int v, v0;
    const int vmax = (int)floorf(ORBFE_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    const int vmin = (int)ceilf(ORBFE_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = ORBFE_HALF_PATCH * ORBFE_HALF_PATCH;
    for (v = 0; v < 16; ++v) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = (int)lrint(sqrt(hp2 - v * v));
    for (v = ORBFE_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// cv::resize coefficient table of one axis
static void resize_axis(int ssize, int dsize, bool is_x, OrbTab *out)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        auto sat = [](int v) { return (int16_t)std::min(32767, std::max(-32768, v)); };
        out[d].s = (int16_t)s;
        out[d].c0 = sat(cv_round_f((1.f - f) * 2048));
        out[d].c1 = sat(cv_round_f(f * 2048));
        out[d].pad = 0;
    }
}

static orbfe_status build_plan(orbfe_handle *h, int w, int ht)
{
    if (h->plan_valid && h->plan.w == w && h->plan.h == ht) return ORBFE_OK;
    const int nl = h->prm.nlevels;
    OrbPlan P;
    memset(&P, 0, sizeof(P));
    P.nlevels = nl;
    P.w = w;
    P.h = ht;
    P.ini_th = std::min(255, std::max(0, h->prm.ini_th_fast));
    P.min_th = std::min(255, std::max(0, h->prm.min_th_fast));
    P.blur_rounding = h->prm.blur_rounding;
    P.dbg = h->opt_debug;
    // Rows a FAST / blur wave walks.  Long runs amortise the 8 (FAST) / 6 (blur) halo rows -- right for batches, whose waves
    // fill the chip anyway.  A handle made for the online call (a frame or a few per call) is latency-bound instead: one wave's
    // walk IS the kernel's duration, so it takes short runs and more waves: single 640x480 frame, FAST 41 -> 25 -> 21 us and blur
    // 19 -> 11 -> 9 us with 40 -> 16 -> 8 rows (ORBFE_OPT_ROWS overrides, 8..512).
    int rows_per_wave = h->prm.max_batch <= 2 ? 8 : (h->prm.max_batch <= 8 ? 16 : ORBFE_ROWS_PER_WAVE);
    if (h->opt_rows >= 8 && h->opt_rows <= 512) rows_per_wave = h->opt_rows;
    // the FAST and the blur walk can take different run lengths (ORBFE_OPT_ROWS_FAST / ORBFE_OPT_ROWS_BLUR; A/B in
    // profiles/r04_ab_experiments.json): a longer run amortises the 8 (FAST) / 6 (blur) halo steps, a shorter one balances better
    int rows_fast = rows_per_wave, rows_blur = rows_per_wave;
    if (h->opt_rows_fast >= 8 && h->opt_rows_fast <= 512) rows_fast = h->opt_rows_fast;
    if (h->opt_rows_blur >= 8 && h->opt_rows_blur <= 512) rows_blur = h->opt_rows_blur;
    std::vector<OrbCell> cells;
    std::vector<OrbTab> tabs;
    int64_t off = 0;
    int key_off = 0, sel_off = 0, cell_cap = 1, max_sel = 0;
    for (int l = 0; l < nl; ++l) {
        OrbLevel &L = P.lv[l];
        L.w = cv_round_f((float)w * h->inv_scale[l]);   // src/ORBextractor.cc:1122
        L.h = cv_round_f((float)ht * h->inv_scale[l]);
        L.pitch = orb_align_up(L.w, 64);
        L.off = (int32_t)off;
        off = orb_align_up64(off + (int64_t)L.pitch * L.h, 256);
        if (off > 0x7FFFFFFF) { orbfe_set_error("pyramid slice exceeds 2 GiB"); return ORBFE_ERR_SIZE; }
        // FAST grid (src/ORBextractor.cc:780-796)
        const int minb = ORBFE_EDGE - 3;
        const int maxbx = L.w - ORBFE_EDGE + 3, maxby = L.h - ORBFE_EDGE + 3;
        const float width = (float)(maxbx - minb), height = (float)(maxby - minb);
        const float W = 30;
        if (width < W || height < W) {
            orbfe_set_error("level %d (%dx%d) is smaller than one 30-px FAST cell plus borders", l, L.w, L.h);
            return ORBFE_ERR_SIZE;
        }
        L.ncols = (int)(width / W);
        L.nrows = (int)(height / W);
        L.wcell = (int)ceilf(width / L.ncols);
        L.hcell = (int)ceilf(height / L.nrows);
        L.cell0 = (int)cells.size();
        int key_cap = 0;
        for (int i = 0; i < L.nrows; ++i) {
            const float iniY = (float)(minb + i * L.hcell);
            float maxY = iniY + L.hcell + 6;
            if (iniY >= maxby - 3) continue;  // :803
            if (maxY > maxby) maxY = (float)maxby;
            for (int j = 0; j < L.ncols; ++j) {
                const float iniX = (float)(minb + j * L.wcell);
                float maxX = iniX + L.wcell + 6;
                if (iniX >= maxbx - 6) continue;  // :812
                if (maxX > maxbx) maxX = (float)maxbx;
                OrbCell c;
                c.level = (uint16_t)l;
                c.x0 = (uint16_t)iniX;
                c.y0 = (uint16_t)iniY;
                c.tw = (uint16_t)((int)maxX - (int)iniX);
                c.th = (uint16_t)((int)maxY - (int)iniY);
                c.ox = (uint16_t)(j * L.wcell);
                c.oy = (uint16_t)(i * L.hcell);
                c.pad = 0;
                if (c.tw > ORBFE_TILE_MAX || c.th > ORBFE_TILE_MAX) {
                    orbfe_set_error("FAST tile %dx%d exceeds %d", c.tw, c.th, ORBFE_TILE_MAX);
                    return ORBFE_ERR_SIZE;
                }
                // strict 3x3 NMS keeps at most one keypoint per 2x2 block of the detectable interior
                const int iw = std::max(0, (int)c.tw - 6), ih = std::max(0, (int)c.th - 6);
                const int worst = ((iw + 1) / 2) * ((ih + 1) / 2);
                cell_cap = std::max(cell_cap, worst);
                key_cap += worst;
                cells.push_back(c);
            }
        }
        L.ncells = (int)cells.size() - L.cell0;
        {
            // non-skipped cells form a full (rows x cols) sub-grid (the skip rules depend on i or j alone)
            int ncc = 0;
            for (int k = L.cell0; k < (int)cells.size() && cells[k].y0 == cells[L.cell0].y0; ++k) ++ncc;
            L.ncc = ncc;
        }
        {
            const OrbCell &clast = cells.back();
            L.ix1 = L.ncells ? clast.x0 + clast.tw - 3 : ORBFE_EDGE;
            L.iy1 = L.ncells ? clast.y0 + clast.th - 3 : ORBFE_EDGE;
        }
        L.nfeat = h->feat[l];
        // quadtree roots (src/ORBextractor.cc:545-559)
        L.nini = (int)roundf((float)(maxbx - minb) / (float)(maxby - minb));
        if (L.nini < 1 || L.nini > ORBFE_MAX_ROOTS) {  // 0 roots: the reference divides by zero (:547)
            orbfe_set_error("level %d aspect ratio gives %d quadtree roots (supported: 1..%d)", l, L.nini, ORBFE_MAX_ROOTS);
            return ORBFE_ERR_SIZE;
        }
        L.hx = (float)(maxbx - minb) / L.nini;
        for (int i = 0; i <= L.nini; ++i) L.root_x[i] = (int)(L.hx * (float)i);
        L.key_off = key_off;
        L.key_cap = key_cap;
        key_off += orb_align_up(std::max(key_cap, 1), 64);
        L.sel_cap = std::max(L.nfeat + 2, 4 * L.nini);
        L.sel_off = sel_off;
        sel_off += orb_align_up(L.sel_cap, 64);
        max_sel = std::max(max_sel, L.sel_cap);
        L.scale = h->scale[l];
        L.patch_size = (float)(int)(ORBFE_PATCH * h->scale[l]);  // :846
        if (l >= 1) {
            const OrbLevel &S = P.lv[l - 1];
            if (S.w >= 2 * L.w) {  // k_pyr_walk: the 4 source pairs of a lane must fit one 8-byte window
                orbfe_set_error("scale factor too large: level %d is less than half as wide as level %d", l, l - 1);
                return ORBFE_ERR_ARG;
            }
            // tap tables, 4-entry aligned and padded by 8 (a lane reads the taps of its 4 pixels / 8 rows as
            // 16-byte loads; entries past the end repeat the last one)
            auto add_axis = [&](int ssize, int dsize, bool is_x) {
                while (tabs.size() % 4) tabs.push_back(OrbTab{0, 0, 0, 0});
                const int at = (int)tabs.size();
                tabs.resize(tabs.size() + dsize + 8);
                resize_axis(ssize, dsize, is_x, &tabs[at]);
                for (int i = 0; i < 8; ++i) tabs[at + dsize + i] = tabs[at + dsize - 1];
                return at;
            };
            L.xtab = add_axis(S.w, L.w, true);
            L.ytab = add_axis(S.h, L.h, false);
            // k_pyr_walk completes at most one destination row per source row: the source row index must grow strictly
            for (int d = 1; d < L.h; ++d)
                if (tabs[(size_t)L.ytab + d].s <= tabs[(size_t)L.ytab + d - 1].s) {
                    orbfe_set_error("level %d: vertical resize taps are not strictly increasing", l);
                    return ORBFE_ERR_ARG;
                }
            if (orbk_pyramid_lds_bytes(L.h) > 64 * 1024) {
                orbfe_set_error("level %d too tall for the pyramid kernel's LDS tap table", l);
                return ORBFE_ERR_SIZE;
            }
        }
        if (L.w > 4095 + 2 * ORBFE_MINB || L.h > 4095 + 2 * ORBFE_MINB) {
            orbfe_set_error("level %d exceeds the 12-bit key coordinate range", l);
            return ORBFE_ERR_SIZE;
        }
    }
#ifdef ORBFE_DEVELOPER
    // Two pyramid levels per launch (ORBFE_OPT_PYR_FUSE, developer builds): for B = 1, 3, 5, ... with a level C = B + 1 above
    // it, the tiling of B and the first C column / row every tile column / row owns (C pixel (x2, y2) belongs to the tile that
    // holds its top-left tap (sx(x2), sy(y2)) in its own -- non-overlap -- part).
    for (int l = 1; l + 1 < nl; l += 2) {
        OrbLevel &B = P.lv[l];
        const OrbLevel &C = P.lv[l + 1];
        const int ngroups = (B.w + 3) / 4;
        const int ntx0 = (ngroups + 63) / 64;
        int gx = std::max(2, std::min(64, (ngroups + ntx0 - 1) / ntx0 + (ntx0 > 1 ? 1 : 0)));
        int gy = std::max(1, 256 / gx);
        const int trows = gy * ORBFE_PW_ROWS;
        if (trows < 2 || orbk_pyramid2_lds_bytes(gx, gy) > 60 * 1024) continue;
        const int tiles_x = ngroups <= gx ? 1 : (ngroups - 1 + gx - 2) / (gx - 1);
        const int tiles_y = B.h <= trows ? 1 : (B.h - 1 + trows - 2) / (trows - 1);
        auto add_i32 = [&](const std::vector<int32_t> &v) {
            while (tabs.size() % 4) tabs.push_back(OrbTab{0, 0, 0, 0});
            const int at = (int)tabs.size();
            tabs.resize(tabs.size() + (v.size() + 1) / 2 + 1);
            memcpy(&tabs[(size_t)at], v.data(), v.size() * sizeof(int32_t));
            return at;
        };
        std::vector<int32_t> cxs((size_t)tiles_x + 1), cys((size_t)tiles_y + 1);
        for (int t = 0; t <= tiles_x; ++t) {
            int x2 = 0;
            if (t == tiles_x) x2 = C.w;
            else
                while (x2 < C.w && tabs[(size_t)C.xtab + x2].s < t * (gx - 1) * 4) ++x2;
            cxs[(size_t)t] = x2;
        }
        for (int t = 0; t <= tiles_y; ++t) {
            int y2 = 0;
            if (t == tiles_y) y2 = C.h;
            else
                while (y2 < C.h && tabs[(size_t)C.ytab + y2].s < t * (trows - 1)) ++y2;
            cys[(size_t)t] = y2;
        }
        B.p2_gx = gx; B.p2_gy = gy; B.p2_tx = tiles_x; B.p2_ty = tiles_y;
        B.p2_cxs = add_i32(cxs);
        B.p2_cys = add_i32(cys);
    }
#endif
    P.ncells = (int)cells.size();
    P.cell_cap = cell_cap;
    P.max_ncells = 1;
    for (int l = 0; l < nl; ++l) P.max_ncells = std::max(P.max_ncells, P.lv[l].ncells);
    P.keys_per_frame = key_off;
    P.sel_per_frame = sel_off;
    // node arrays: one slot more than the largest list, rounded to 64 (only the sort buffer inside is a power of two)
    const int M = orb_align_up(std::max(max_sel + 1, 64), 64);
    P.node_cap = M;
    P.max_nini = 1;
    for (int l = 0; l < nl; ++l) P.max_nini = std::max(P.max_nini, P.lv[l].nini);
    // Node arrays normally sit in LDS; a level asking for more nodes than fit (about 2400 features on ONE level) keeps them
    // in global scratch.  What remains is the width of the node index the keys of deep trees carry (14 bits).
    if (M > 16383) {
        orbfe_set_error("nfeatures too large: %d quadtree nodes on one level (at most 16383)", max_sel);
        return ORBFE_ERR_ARG;
    }
    for (int l = 0; l < nl; ++l)
        if (P.lv[l].ncells >= (1 << 16) || P.lv[l].wcell > 63 || P.lv[l].hcell > 63) {
            orbfe_set_error("level %d: %d FAST cells / cell size exceed the 16 + 6 + 6 bit candidate-order key", l, P.lv[l].ncells);
            return ORBFE_ERR_SIZE;
        }
    P.pyr_frame_bytes = off;
    if (tabs.empty()) tabs.resize(1);
    // FAST lane list: per level, per (balanced) row block of <= ORBFE_ROWS_PER_WAVE rows, the 4-px columns x = 16, 20, ... < ix1 form a
    // strip; strips are packed back to back into single-level waves of 64 lanes.  Where a wave boundary falls inside a
    // strip, each side gets one halo lane (computes neighbour strengths, outputs nothing).
    std::vector<OrbLane> flanes, clanes;
    {
        std::vector<OrbLane> stream;
        for (int l = 0; l < nl; ++l) {
            const OrbLevel &L = P.lv[l];
            const int rows = L.iy1 - ORBFE_EDGE, ncol = (L.ix1 - 16 + 3) / 4;
            if (rows <= 0 || ncol <= 0) continue;
            const int frb = rows_fast;
            const int nblk = (rows + frb - 1) / frb, rb = (rows + nblk - 1) / nblk;
            for (int k = 0; k < nblk; ++k) {
                const int ys = ORBFE_EDGE + k * rb, nr = std::min(rb, L.iy1 - ys);
                for (int c = 0; c < ncol && nr > 0; ++c) {
                    OrbLane ln;
                    ln.x = (uint16_t)(16 + 4 * c);
                    ln.ys = (uint16_t)ys;
                    ln.nrows = (uint16_t)nr;
                    ln.flags = (uint16_t)(l << 8);
                    stream.push_back(ln);
                }
            }
        }
        auto same_strip = [](const OrbLane &a, const OrbLane &b2) {
            return (a.flags >> 8) == (b2.flags >> 8) && a.ys == b2.ys && b2.x == a.x + 4;
        };
        // The dense kernel of batch handles (k_fast_map_u) walks whole CELL ROWS: a run of rows starts on a cell-row boundary and
        // ends on one (or at the end of the detectable interior), and a wave holds runs of ONE length only -- the reference's FAST
        // never looks across a cell boundary (:798-838), so such a run needs no strength row of its neighbours, and everything that
        // depends on the position inside the run alone is scalar in the kernel.  k cell rows per run, k = rows_fast / hcell rounded (1 for the default
        // 40 rows and the ~31-row cells of every shipped configuration).  Handles made for a few frames per call (rows_fast < 24)
        // keep short balanced runs and the generic kernel: such a call is bound by the length of one wave's walk.
        const bool cellrows = rows_fast >= 24;
        P.fast_cellrows = cellrows ? 1 : 0;
        std::vector<OrbLane> ustream;
        if (cellrows)
            for (int l = 0; l < nl; ++l) {
                const OrbLevel &L = P.lv[l];
                const int rows = L.iy1 - ORBFE_EDGE, ncol = (L.ix1 - 16 + 3) / 4;
                if (rows <= 0 || ncol <= 0) continue;
                const int kc = std::max(1, (rows_fast + L.hcell / 2) / L.hcell), rb = kc * L.hcell;
                for (int ys = ORBFE_EDGE; ys < L.iy1; ys += rb) {
                    const int nr = std::min(rb, L.iy1 - ys);
                    for (int c = 0; c < ncol; ++c) {
                        OrbLane ln;
                        ln.x = (uint16_t)(16 + 4 * c);
                        ln.ys = (uint16_t)ys;
                        ln.nrows = (uint16_t)nr;
                        ln.flags = (uint16_t)(l << 8);
                        ustream.push_back(ln);
                    }
                }
            }
        const std::vector<OrbLane> &dstream = cellrows ? ustream : stream;
        size_t i = 0;
        for (int l = 0; l <= ORBFE_MAX_LEVELS; ++l) P.fwave_off[l] = -1;
        while (i < dstream.size()) {
            const int lvl = dstream[i].flags >> 8;
            const size_t w0 = flanes.size();
            const OrbLane first = dstream[i];
            // cell-row form: runs of ONE length per wave (every run starts on a cell row: the lanes are in step)
            auto fits = [&](const OrbLane &ln) { return !cellrows || ln.nrows == first.nrows; };
            if (P.fwave_off[lvl] < 0) P.fwave_off[lvl] = (int)(w0 / 64);
            if (i > 0 && same_strip(dstream[i - 1], dstream[i])) {  // continuing a cut strip: left halo first
                OrbLane hl = dstream[i - 1];
                hl.flags |= 1;
                flanes.push_back(hl);
            }
            while (i < dstream.size() && (dstream[i].flags >> 8) == lvl && fits(dstream[i]) && flanes.size() - w0 < 64) {
                const bool more = i + 1 < dstream.size() && same_strip(dstream[i], dstream[i + 1]);
                if (flanes.size() - w0 == 63 && more) {  // last slot and the strip goes on: right halo, lane moves on
                    OrbLane hr = dstream[i];
                    hr.flags |= 1;
                    flanes.push_back(hr);
                    break;
                }
                flanes.push_back(dstream[i]);
                ++i;
            }
            while (flanes.size() - w0 < 64) {  // dead lanes (cell-row form: they carry the wave's run, as its scalar row state wants)
                OrbLane d;
                d.x = 16;
                d.ys = cellrows ? first.ys : (uint16_t)ORBFE_EDGE;
                d.nrows = cellrows ? first.nrows : (uint16_t)0;
                d.flags = (uint16_t)((lvl << 8) | 1);
                flanes.push_back(d);
            }
        }
    // Lane list of the lane-compacting form (k_fast_map_c): the same strips, but EVERY piece of a strip inside a wave is closed by a halo
    // lane on both sides (the 4-px column before / behind it, flag bit 0) -- its lanes take their left / right neighbour pixels from
    // the neighbouring lanes, not from memory.  At the image's side borders the halo is the column outside the detectable interior
    // (x = 12 / the column behind the last one: real pixels, nothing inside, nothing output).
    {
        size_t i = 0;
        while (i < stream.size()) {
            const int lvl = stream[i].flags >> 8;
            const size_t w0 = clanes.size();
            while (i < stream.size() && (stream[i].flags >> 8) == lvl && 64 - (clanes.size() - w0) >= 3) {
                OrbLane hl = stream[i];
                hl.x = (uint16_t)(hl.x - 4);
                hl.flags |= 1;
                clanes.push_back(hl);
                size_t room = 64 - (clanes.size() - w0) - 1;   // the right halo takes the last slot
                OrbLane last = stream[i];
                while (room > 0) {
                    last = stream[i];
                    clanes.push_back(last);
                    ++i;
                    --room;
                    if (!(i < stream.size() && same_strip(last, stream[i]))) break;
                }
                OrbLane hr = last;
                hr.x = (uint16_t)(hr.x + 4);
                hr.flags |= 1;
                clanes.push_back(hr);
            }
            while (clanes.size() - w0 < 64) {  // dead lanes
                OrbLane d;
                d.x = 16;
                d.ys = ORBFE_EDGE;
                d.nrows = 0;
                d.flags = (uint16_t)((lvl << 8) | 1);
                clanes.push_back(d);
            }
        }
    }
    }
    P.nfwaves = (int)(flanes.size() / 64);
    P.nfwaves_c = (int)(clanes.size() / 64);
    P.fwave_off[nl] = P.nfwaves;
    for (int l = ORBFE_MAX_LEVELS; l > nl; --l) P.fwave_off[l] = P.nfwaves;
    for (int l = nl - 1; l >= 0; --l)
        if (P.fwave_off[l] < 0) P.fwave_off[l] = P.fwave_off[l + 1];   // a level without FAST rows
    int64_t fast_row_steps = 0;  // wave row steps one frame costs k_fast_map (VALU model of bench.py's roofline)
    for (int wv = 0; wv < P.nfwaves; ++wv) {
        int mx = 0;
        for (int i = 0; i < 64; ++i) mx = std::max(mx, (int)flanes[(size_t)wv * 64 + i].nrows);
        fast_row_steps += mx + 8;
    }
    // blur lane list: every 4-px column of every (balanced, <= ORBFE_ROWS_PER_WAVE rows) row block, single-level waves, no
    // halos.  Lanes do not talk to each other, so a wave can hold columns of different row blocks.  Columns whose 12-byte
    // window [x - 4, x + 8) lies inside the row (flag bit 1: no reflected column) skip the byte rearrangement of the border
    // path, so they get waves of their own.  Everything is laid out in 64-BYTE PIECES (16 lanes): the left and the right piece
    // of a row block go to the border waves whole, the pieces between them to the interior waves -- every store instruction
    // then writes whole 64-byte pieces.  (Border waves holding only the 2 - 3 reflected columns of 20-odd row blocks wrote a
    // lone dword into 64 different lines per store: WRITE_SIZE was 1.19x the output, profiles/r04_ab_experiments.json;
    // ORBFE_OPT_BLUR_PIECES = 0 brings that packing back for the A/B.)
    // k_blur7's border lanes: folded horizontal weights per (level, lane type) -- tap t of output pixel c sits on column
    // reflect101(c - 3 + t), and taps that land on the same column add up (at most 49 + 49: a byte)
    for (int l = 0; l < nl; ++l) {
        const OrbLevel &L = P.lv[l];
        if (L.w < 16) { orbfe_set_error("level %d too narrow for the blur kernel", l); return ORBFE_ERR_SIZE; }
        const int kern[7] = {18, 34, 49, 55, 49, 34, 18};
        const int xlast = ((L.w - 1) / 4) * 4;
        const int xs[4] = {4, 0, xlast - 4, xlast};   // a lane of every type
        for (int ty = 0; ty < 4; ++ty) {
            const int x = xs[ty], base = std::min(std::max(x - 4, 0), L.w - 12);
            for (int j = 0; j < 4; ++j) {
                const int c = std::min(x + j, L.w - 1);   // output pixels past the row's end are computed and not stored
                for (int t = 0; t < 7; ++t) {
                    int col = c - 3 + t;
                    if (col < 0) col = -col;
                    if (col >= L.w) col = 2 * L.w - 2 - col;
                    const int bi = col - base;
                    if (bi < 0 || bi > 11) { orbfe_set_error("level %d: blur window of column %d does not hold column %d", l, x, col); return ORBFE_ERR_SIZE; }
                    P.blur_wt[l][ty][3 * j + bi / 4] += (uint32_t)kern[t] << (8 * (bi % 4));
                }
            }
        }
    }
    std::vector<OrbLane> blanes;
    std::vector<OrbLaneR> blanesR;   // the resize job of every blur lane (fused blur + pyramid pass), same index
    const bool blur_pieces = h->opt_blur_pieces != 0;
    const int blur_updown = std::max(0, std::min(2, h->opt_blur_updown));
    for (int l = 0; l < nl; ++l) {
        const OrbLevel &L = P.lv[l];
        if (L.w < 16) { orbfe_set_error("level %d too narrow for the blur kernel", l); return ORBFE_ERR_SIZE; }
        const int brb = rows_blur;
        const int ncol = (L.w + 3) / 4, nblk = (L.h + brb - 1) / brb, rb = (L.h + nblk - 1) / nblk;
        // first column of the 64-byte piece that holds the first column whose window reaches past the row's right end
        const int right0 = (std::min(ncol - 1, std::max(0, (L.w - 8) / 4 + 1)) / 16) * 16;
        P.bwave_off[l] = (int)(blanes.size() / 64);
        // fused blur + pyramid pass: destination dword j of level l + 1 is carried by the blur lane of source column
        // floor(j * ncol / ncolD) (injective: the level shrinks), i.e. by a lane whose blur window lies over its source pixels
        std::vector<int> dword_of_col((size_t)ncol, -1);
        if (l + 1 < nl) {
            const int ncolD = (P.lv[l + 1].w + 3) / 4;
            for (int j = 0; j < ncolD; ++j) {
                const int c = std::min(ncol - 1, (int)((int64_t)j * ncol / ncolD));
                if (dword_of_col[(size_t)c] >= 0) { orbfe_set_error("level %d: two destination dwords on one blur column", l); return ORBFE_ERR_SIZE; }
                dword_of_col[(size_t)c] = j;
            }
        }
        auto resize_job = [&](int c, int ys, int nr) {
            OrbLaneR r = {0, 0, 0, 0};
            if (l + 1 >= nl || c < 0 || dword_of_col[(size_t)c] < 0 || nr <= 0) return r;
            const OrbLevel &D = P.lv[l + 1];
            int d0 = 0;
            while (d0 < D.h && tabs[(size_t)D.ytab + d0].s < ys) ++d0;          // first destination row whose upper source row is in the block
            int d1 = d0;
            while (d1 < D.h && tabs[(size_t)D.ytab + d1].s < ys + nr) ++d1;
            r.dj = (uint16_t)dword_of_col[(size_t)c];
            r.d0 = (uint16_t)d0;
            r.nd = (uint16_t)(d1 - d0);
            return r;
        };
        auto dead = [&](bool interior_wave) {
            OrbLane d;
            d.x = (uint16_t)(interior_wave ? 4 : 0);
            d.ys = 0;
            d.nrows = 0;
            d.flags = (uint16_t)((l << 8) | 1 | (interior_wave ? 2 : 0));
            return d;
        };
        const int fuse = h->fuse_blur_pyr;   // 0: blur only, 1: every blur lane carries a resize job, 2: resize jobs in waves of their own
        auto blur_lane = [&](int c, int ys, int nr, bool interior) {
            OrbLane ln;
            ln.x = (uint16_t)(4 * c);
            ln.ys = (uint16_t)ys;
            ln.nrows = (uint16_t)nr;
            ln.flags = (uint16_t)((l << 8) | (interior ? 2 : 0));
            return ln;
        };
        auto is_interior = [&](int c) {
            const bool no_reflection = 4 * c >= 4 && 4 * c + 8 <= L.w;
            return blur_pieces ? (no_reflection && c >= 16 && c < right0) : no_reflection;
        };
        if (fuse == 2 && l + 1 < nl) {
            // interior blur waves and resize waves of the same row blocks side by side in the wave list (a workgroup is four
            // consecutive waves): whichever kind touches a source row second finds it in L1 / L2.  Two queues, whole waves of one
            // kind are emitted as soon as they fill, so neither kind runs more than a row block ahead of the other.
            std::vector<OrbLane> qb, qr;
            std::vector<OrbLaneR> qrr;
            const int ncolD = (P.lv[l + 1].w + 3) / 4;
            auto flush = [&](bool all) {
                while (qb.size() >= 64 || qr.size() >= 64 || (all && (!qb.empty() || !qr.empty()))) {
                    if (qb.size() >= 64 || (all && !qb.empty())) {
                        const size_t n = std::min<size_t>(64, qb.size());
                        blanes.insert(blanes.end(), qb.begin(), qb.begin() + n);
                        blanesR.insert(blanesR.end(), n, OrbLaneR{0, 0, 0, 0});
                        qb.erase(qb.begin(), qb.begin() + n);
                        while (blanes.size() % 64) { blanes.push_back(dead(true)); blanesR.push_back(OrbLaneR{0, 0, 0, 0}); }
                    }
                    if (qr.size() >= 64 || (all && !qr.empty())) {
                        const size_t n = std::min<size_t>(64, qr.size());
                        blanes.insert(blanes.end(), qr.begin(), qr.begin() + n);
                        blanesR.insert(blanesR.end(), qrr.begin(), qrr.begin() + n);
                        qr.erase(qr.begin(), qr.begin() + n);
                        qrr.erase(qrr.begin(), qrr.begin() + n);
                        while (blanes.size() % 64) {
                            OrbLane d;
                            d.x = 0; d.ys = 0; d.nrows = 0;
                            d.flags = (uint16_t)((l << 8) | 4 | 1);
                            blanes.push_back(d);
                            blanesR.push_back(OrbLaneR{0, 0, 0, 0});
                        }
                    }
                }
            };
            for (int k = 0; k < nblk; ++k) {
                const int ys = k * rb, nr = std::min(rb, L.h - ys);
                if (nr <= 0) continue;
                for (int c = 0; c < ncol; ++c)
                    if (is_interior(c)) qb.push_back(blur_lane(c, ys, nr, true));
                // the destination rows whose upper source row lies in this row block, for every destination dword
                OrbLaneR rows = resize_job(0, ys, nr);
                if (dword_of_col[0] < 0) {   // resize_job wants a column that carries a dword: take the row range from any such column
                    for (int c = 0; c < ncol; ++c)
                        if (dword_of_col[(size_t)c] >= 0) { rows = resize_job(c, ys, nr); break; }
                }
                for (int j = 0; j < ncolD && rows.nd; ++j) {
                    OrbLane ln;
                    ln.x = 0; ln.ys = (uint16_t)ys; ln.nrows = 0;
                    ln.flags = (uint16_t)((l << 8) | 4);
                    qr.push_back(ln);
                    qrr.push_back(OrbLaneR{(uint16_t)j, rows.d0, rows.nd, 0});
                }
                flush(false);
            }
            flush(true);
            // border blur waves as in the plain layout
            for (int k = 0; k < nblk; ++k) {
                const int ys = k * rb, nr = std::min(rb, L.h - ys);
                if (nr <= 0) continue;
                for (int c = 0; c < ncol; ++c) {
                    if (is_interior(c)) continue;
                    blanes.push_back(blur_lane(c, ys, nr, false));
                    blanesR.push_back(OrbLaneR{0, 0, 0, 0});
                    if (blur_pieces && c == ncol - 1)
                        while (blanes.size() % 16) { blanes.push_back(dead(false)); blanesR.push_back(OrbLaneR{0, 0, 0, 0}); }
                }
            }
            while (blanes.size() % 64) { blanes.push_back(dead(false)); blanesR.push_back(OrbLaneR{0, 0, 0, 0}); }
        } else
        for (int pass = 0; pass < 2; ++pass) {  // 0: interior waves, 1: border waves
            // Odd row blocks walk UPWARDS (flag bit 3, wave-uniform: the even blocks' lanes come first, then the odd blocks'; the
            // kernel is vertically symmetric).  Two neighbouring row blocks then read the rows around their common boundary at the
            // same end of their walks -- both at the start or both at the end; all waves of a frame are in flight together -- and
            // the second reader finds the halo rows in L2 instead of HBM: FETCH_SIZE of the kernel -18 % when every level does it.
            // Keeping the two directions in waves of their own can cost a level one more (partly filled) wave, i.e. instructions,
            // which is what the pipeline as a whole is bound by: a (level, pass) is split only where the wave count stays the same
            // (ORBFE_OPT_BLUR_UPDOWN = 0: never, 2: always; the fused blur + pyramid passes walk downwards only).
            auto emit = [&](int nparity, std::vector<OrbLane> &ol, std::vector<OrbLaneR> &orr) {
                for (int par = 0; par < nparity; ++par) {
                    const uint16_t upflag = par ? 8 : 0;
                    for (int k = 0; k < nblk; ++k) {
                        if (nparity == 2 && (k & 1) != par) continue;
                        const int ys = k * rb, nr = std::min(rb, L.h - ys);
                        if (nr <= 0) continue;
                        for (int c = 0; c < ncol; ++c) {
                            const bool interior = is_interior(c);
                            if (interior != (pass == 0)) continue;
                            ol.push_back(blur_lane(c, ys, nr, interior));
                            ol.back().flags |= upflag;
                            orr.push_back(fuse == 1 ? resize_job(c, ys, nr) : OrbLaneR{0, 0, 0, 0});
                            // the right piece is padded to its 16 slots, so that the next row block's left piece starts a piece again
                            if (blur_pieces && pass == 1 && c == ncol - 1)
                                while (ol.size() % 16) { ol.push_back(dead(false)); ol.back().flags |= upflag; orr.push_back(OrbLaneR{0, 0, 0, 0}); }
                        }
                    }
                    // dead lanes: shadow a column of the wave's kind
                    while (ol.size() % 64) { ol.push_back(dead(pass == 0)); ol.back().flags |= upflag; orr.push_back(OrbLaneR{0, 0, 0, 0}); }
                }
            };
            std::vector<OrbLane> l1, l2;
            std::vector<OrbLaneR> r1, r2;
            emit(1, l1, r1);
            if (fuse == 0 && blur_updown) emit(2, l2, r2);
            const bool split = fuse == 0 && blur_updown && (blur_updown == 2 || l2.size() == l1.size());
            blanes.insert(blanes.end(), (split ? l2 : l1).begin(), (split ? l2 : l1).end());
            blanesR.insert(blanesR.end(), (split ? r2 : r1).begin(), (split ? r2 : r1).end());
        }
        P.bwave_off[l + 1] = (int)(blanes.size() / 64);
    }
    P.nbwaves = (int)(blanes.size() / 64);
    P.blur_split = h->fuse_blur_pyr == 2;
    if (P.ini_th < P.min_th) {
        orbfe_set_error("iniThFAST (%d) must be >= minThFAST (%d)", P.ini_th, P.min_th);
        return ORBFE_ERR_ARG;
    }

    ORBFE_HIP(h->d_plan.ensure(sizeof(OrbPlan)));
    ORBFE_HIP(h->d_tabs.ensure(tabs.size() * sizeof(OrbTab)));
    ORBFE_HIP(h->d_flanes.ensure(std::max<size_t>(flanes.size(), 1) * sizeof(OrbLane)));
    ORBFE_HIP(h->d_flanes_c.ensure(std::max<size_t>(clanes.size(), 1) * sizeof(OrbLane)));
    ORBFE_HIP(h->d_blanes.ensure(std::max<size_t>(blanes.size(), 1) * sizeof(OrbLane)));
    ORBFE_HIP(h->d_blanesR.ensure(std::max<size_t>(blanesR.size(), 1) * sizeof(OrbLaneR)));
    // synchronous copies: plans change rarely (frame size change), never inside the timed region.  Earlier batches may
    // still be in flight on the handle's stream or on the caller's stream of the previous device call: both are drained
    // before the plan tables they read are overwritten.
    ORBFE_HIP(hipStreamSynchronize(h->stream));
    ORBFE_HIP(wait_last_call(h));
    ORBFE_HIP(hipMemcpy(h->d_plan.p, &P, sizeof(OrbPlan), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(h->d_tabs.p, tabs.data(), tabs.size() * sizeof(OrbTab), hipMemcpyHostToDevice));
    if (!flanes.empty())
        ORBFE_HIP(hipMemcpy(h->d_flanes.p, flanes.data(), flanes.size() * sizeof(OrbLane), hipMemcpyHostToDevice));
    if (!clanes.empty())
        ORBFE_HIP(hipMemcpy(h->d_flanes_c.p, clanes.data(), clanes.size() * sizeof(OrbLane), hipMemcpyHostToDevice));
    if (!blanes.empty())
        ORBFE_HIP(hipMemcpy(h->d_blanes.p, blanes.data(), blanes.size() * sizeof(OrbLane), hipMemcpyHostToDevice));
    if (!blanesR.empty())
        ORBFE_HIP(hipMemcpy(h->d_blanesR.p, blanesR.data(), blanesR.size() * sizeof(OrbLaneR), hipMemcpyHostToDevice));
    ORBFE_HIP(orbk_prepare_octree(M, P.max_nini, P.w, P.h, P.max_ncells));
    h->plan = P;
    h->fast_row_steps = fast_row_steps;
    h->cells.swap(cells);
    h->tabs.swap(tabs);
    h->plan_valid = true;
    return ORBFE_OK;
}

static orbfe_status ensure_batch_buffers(orbfe_handle *h, int nframes)
{
    const OrbPlan &P = h->plan;
    const size_t B = (size_t)nframes;
    if (B * (size_t)P.pyr_frame_bytes > h->d_pyr.bytes || B * (size_t)P.keys_per_frame * sizeof(uint2) > h->d_skeys.bytes) {
        // a block is about to be re-allocated: nothing may still be reading the old one
        ORBFE_HIP(hipStreamSynchronize(h->stream));
        ORBFE_HIP(wait_last_call(h));
    }
    ORBFE_HIP(h->d_pyr.ensure(B * (size_t)P.pyr_frame_bytes));
    ORBFE_HIP(h->d_blur.ensure(B * (size_t)P.pyr_frame_bytes));
    ORBFE_HIP(h->d_skeys.ensure(B * (size_t)P.keys_per_frame * sizeof(uint2)));
    // survivor counts and cell flags are zeroed before every FAST pass: one block (counts | flags of the largest batch), so
    // that one memset clears both -- a call with fewer frames passes the flags' offset for ITS frame count (run_batch)
    ORBFE_HIP(h->d_scount.ensure(B * P.nlevels * ORBFE_NK_STRIDE * sizeof(int32_t) +
                                 B * P.nlevels * (size_t)((P.max_ncells + 31) / 32) * sizeof(uint32_t)));
    ORBFE_HIP(h->d_knode.ensure(B * (size_t)P.keys_per_frame * sizeof(uint16_t)));
    ORBFE_HIP(h->d_qtbox.ensure(B * (size_t)P.nlevels * orbk_octree_box_bytes(P.node_cap)));  // deep quadtrees only
    if (orbk_octree_lds_bytes(P.node_cap, std::max(1, P.max_nini), P.w, P.h, P.max_ncells) > (size_t)ORBFE_LDS_MAX)
        ORBFE_HIP(h->d_qtnodes.ensure(B * (size_t)P.nlevels * orbk_octree_node_bytes(P.node_cap)));  // quadtrees beyond the LDS
    ORBFE_HIP(h->d_sel.ensure(B * (size_t)P.sel_per_frame * sizeof(uint32_t)));
    ORBFE_HIP(h->d_nsel.ensure(B * P.nlevels * sizeof(int32_t)));
    ORBFE_HIP(h->d_nkeys.ensure(B * P.nlevels * ORBFE_NK_STRIDE * sizeof(int32_t)));
    if (!h->d_misc.p) {
        ORBFE_HIP(h->d_misc.ensure(1024));
        ORBFE_HIP(hipMemset(h->d_misc.p, 0, 1024));
    }
    return ORBFE_OK;
}

// reads and clears the sticky overflow word; the stream of the last batched call is drained first
static orbfe_status read_overflow(orbfe_handle *h, int32_t *flags)
{
    *flags = 0;
    if (!h->d_misc.p) return ORBFE_OK;
    ORBFE_HIP(wait_last_call(h));
    ORBFE_HIP(hipMemcpy(flags, h->d_misc.p, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (*flags) ORBFE_HIP(hipMemset(h->d_misc.p, 0, sizeof(int32_t)));
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// create / destroy / getters
// ---------------------------------------------------------------------------------------------------
static orbfe_status create_impl(const orbfe_params *p, hipStream_t borrowed, hipStream_t borrowed_side, bool borrow, orbfe_handle **out);
extern "C" orbfe_status orbfe_create(const orbfe_params *p, orbfe_handle **out) { return create_impl(p, nullptr, nullptr, false, out); }
// A handle for a pipe of orbfe_pipeline: `st` (the pipe's stream) serves as the handle's own stream and stays the pipeline's;
// the copy streams of the host entry points are not created (orbfe_extract / orbfe_extract_batch answer ORBFE_ERR_STATE).  Every
// stream a process creates is multiplexed onto a few hardware queues: a pipeline of 12 pipes made 74 of them this way, 26 now.
orbfe_status orbfe_internal_create_on_stream(const orbfe_params *p, void *st, void *side, orbfe_handle **out)
{
    return create_impl(p, (hipStream_t)st, (hipStream_t)side, true, out);
}

static orbfe_status create_impl(const orbfe_params *p, hipStream_t borrowed, hipStream_t borrowed_side, bool borrow, orbfe_handle **out)
{
    if (!p || !out) { orbfe_set_error("null argument"); return ORBFE_ERR_ARG; }
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > ORBFE_MAX_LEVELS || p->nfeatures < 0 || !(p->scale_factor > 1.0f) ||
        p->max_batch < 1 || p->max_width < 1 || p->max_height < 1 || p->max_width > 4096 || p->max_height > 4096) {
        orbfe_set_error("bad orbfe_params (nlevels 1..16, scale_factor > 1, max size <= 4096, max_batch >= 1)");
        return ORBFE_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        orbfe_set_error("no HIP device visible; liborbfe has no CPU fallback");
        return ORBFE_ERR_NODEVICE;
    }
    int dev = p->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) { orbfe_set_error("device %d out of range (%d visible)", dev, ndev); return ORBFE_ERR_ARG; }

    orbfe_handle *h = new (std::nothrow) orbfe_handle();
    if (!h) return ORBFE_ERR_NOMEM;
    h->prm = *p;
    h->device = dev;
    DeviceGuard g(dev);
    // src/ORBextractor.cc:404-421
    const int nl = p->nlevels;
    h->scale[0] = 1.0f;
    h->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; ++i) {
        h->scale[i] = h->scale[i - 1] * p->scale_factor;
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < nl; ++i) {
        h->inv_scale[i] = 1.0f / h->scale[i];
        h->inv_sigma2[i] = 1.0f / h->sigma2[i];
    }
    // src/ORBextractor.cc:426-439
    const float factor = 1.0f / p->scale_factor;
    float desired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; ++l) {
        h->feat[l] = cv_round_f(desired);
        sum += h->feat[l];
        desired *= factor;
    }
    h->feat[nl - 1] = std::max(p->nfeatures - sum, 0);

    auto fail = [&](orbfe_status s) {
        orbfe_destroy(h);
        return s;
    };
    if (borrow) {
        h->stream = borrowed;
        h->own_stream = false;
    } else if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        orbfe_set_error("hipStreamCreate failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(ORBFE_ERR_HIP);
    }
    for (int r = 0; r < ORBFE_PROF_RING; ++r)
        for (int i = 0; i < ORBFE_EV_N; ++i) h->ev[r][i] = nullptr;
    h->ev_ok = true;
    for (int r = 0; r < ORBFE_PROF_RING; ++r)
        for (int i = 0; i < ORBFE_EV_N; ++i)
            if (hipEventCreate(&h->ev[r][i]) != hipSuccess) { orbfe_set_error("hipEventCreate failed"); return fail(ORBFE_ERR_HIP); }
    if (borrow && borrowed_side) {
        h->side = borrowed_side;
        h->own_side = false;
    }
    if ((h->own_side && hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        orbfe_set_error("side stream / event creation failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(ORBFE_ERR_HIP);
    }
    if (h->own_stream && (hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking) != hipSuccess ||
                          hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking) != hipSuccess)) {
        orbfe_set_error("copy stream creation failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(ORBFE_ERR_HIP);
    }
    for (int k = 0; k < 2; ++k)
        if (hipEventCreateWithFlags(&h->ev_in[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_cmp[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_out[k], hipEventDisableTiming) != hipSuccess) {
            orbfe_set_error("pipeline event creation failed");
            return fail(ORBFE_ERR_HIP);
        }
    if (hipEventCreateWithFlags(&h->ev_fork2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_auto, hipEventDisableTiming) != hipSuccess || h->h_auto.ensure(64) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming) != hipSuccess) {
        orbfe_set_error("event creation failed");
        return fail(ORBFE_ERR_HIP);
    }
#ifdef ORBFE_DEVELOPER
    // a developer build (-DORBFE_DEVELOPER; tools/ab_build.sh) also honours the A/B knobs from the environment, under the
    // names of the options (ORBFE_OPT_OVERLAP -> $ORBFE_OVERLAP ...); the release library takes them through orbfe_set_option only
    {
        static const struct { const char *env; int opt; } kEnv[] = {
            {"ORBFE_OVERLAP", ORBFE_OPT_OVERLAP}, {"ORBFE_ROWS", ORBFE_OPT_ROWS}, {"ORBFE_ROWS_FAST", ORBFE_OPT_ROWS_FAST},
            {"ORBFE_ROWS_BLUR", ORBFE_OPT_ROWS_BLUR}, {"ORBFE_BLUR_PIECES", ORBFE_OPT_BLUR_PIECES},
            {"ORBFE_BLUR_UPDOWN", ORBFE_OPT_BLUR_UPDOWN}, {"ORBFE_PW_ROWS", ORBFE_OPT_PYR_ROWS}, {"ORBFE_PYR_FUSE", ORBFE_OPT_PYR_FUSE},
            {"ORBFE_DEBUG", ORBFE_OPT_DEBUG}, {"ORBFE_FUSE_BLUR_PYR", ORBFE_OPT_FUSE_BLUR_PYR},
            {"ORBFE_FUSE_FAST_PYR", ORBFE_OPT_FUSE_FAST_PYR}, {"ORBFE_FUSE_FAST_PYR_LEVELS", ORBFE_OPT_FUSE_FAST_PYR_LEVELS}};
        for (const auto &k : kEnv)
            if (const char *e = getenv(k.env)) (void)orbfe_set_option(h, k.opt, atoi(e));
        if (const char *e = getenv("ORBFE_QT")) {
            int v[3];
            if (sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]) == 3)
                for (int i = 0; i < 3; ++i) (void)orbfe_set_option(h, ORBFE_OPT_QT_THREADS_0 + i, v[i]);
        }
    }
#endif
    int umax[16];
    host_umax(umax);
    if (orbk_upload_constants(umax) != hipSuccess) {
        orbfe_set_error("constant upload failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(ORBFE_ERR_HIP);
    }
    orbfe_status s = build_plan(h, p->max_width, p->max_height);
    if (s != ORBFE_OK) return fail(s);
    s = ensure_batch_buffers(h, p->max_batch);
    if (s != ORBFE_OK) return fail(s);
    *out = h;
    return ORBFE_OK;
}

extern "C" void orbfe_destroy(orbfe_handle *h)
{
    if (!h) return;
    DeviceGuard g(h->device);
    // nothing of this handle may still be running when its buffers go: the caller's last stream, the side stream of the
    // blur, the host pipeline's copy streams
    if (h->last_stream_valid && h->ev_last) (void)hipEventSynchronize(h->ev_last);
    if (h->side) (void)hipStreamSynchronize(h->side);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
    DevBuf *bufs[] = {&h->d_plan, &h->d_tabs, &h->d_flanes, &h->d_flanes_c, &h->d_blanes, &h->d_blanesR, &h->d_pyr, &h->d_blur, &h->d_skeys, &h->d_scount, &h->d_knode, &h->d_qtbox, &h->d_qtnodes, &h->d_sel, &h->d_nsel, &h->d_nkeys, &h->d_pad,
                      &h->d_stage[0], &h->d_okps[0], &h->d_odesc[0], &h->d_on[0], &h->d_stage[1], &h->d_okps[1], &h->d_odesc[1], &h->d_on[1]};
    for (DevBuf *b : bufs) b->release();
    h->d_misc.release();
    PinBuf *pins[] = {&h->h_stage[0], &h->h_okps[0], &h->h_odesc[0], &h->h_on[0], &h->h_stage[1], &h->h_okps[1], &h->h_odesc[1], &h->h_on[1]};
    for (PinBuf *b : pins) b->release();
    h->h_ovf.release();
    for (int k = 0; k < 2; ++k) {
        if (h->ev_in[k]) (void)hipEventDestroy(h->ev_in[k]);
        if (h->ev_cmp[k]) (void)hipEventDestroy(h->ev_cmp[k]);
        if (h->ev_out[k]) (void)hipEventDestroy(h->ev_out[k]);
    }
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    if (h->ev_ok)
        for (int r = 0; r < ORBFE_PROF_RING; ++r)
            for (int i = 0; i < ORBFE_EV_N; ++i)
                if (h->ev[r][i]) (void)hipEventDestroy(h->ev[r][i]);
    if (h->ev_last) (void)hipEventDestroy(h->ev_last);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
    if (h->ev_auto) (void)hipEventDestroy(h->ev_auto);
    h->h_auto.release();
    if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
    if (h->side && h->own_side) (void)hipStreamDestroy(h->side);
    if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" orbfe_status orbfe_get_scales(const orbfe_handle *h, float *scale, float *inv_scale, float *sigma2,
                                         float *inv_sigma2)
{
    if (!h) return ORBFE_ERR_ARG;
    for (int i = 0; i < h->prm.nlevels; ++i) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv_scale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_get_features_per_level(const orbfe_handle *h, int32_t *out)
{
    if (!h || !out) return ORBFE_ERR_ARG;
    for (int i = 0; i < h->prm.nlevels; ++i) out[i] = h->feat[i];
    return ORBFE_OK;
}

extern "C" int32_t orbfe_keypoint_capacity(const orbfe_handle *h)
{
    if (!h) return 0;
    int total = 0;
    for (int l = 0; l < h->plan.nlevels; ++l) total += h->plan.lv[l].sel_cap;
    return orb_align_up(total, 64);
}

extern "C" orbfe_status orbfe_set_profiling(orbfe_handle *h, int32_t enable)
{
    if (!h) return ORBFE_ERR_ARG;
    h->profiling = enable != 0;
    h->prof_calls = 0;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_get_stage_ms(orbfe_handle *h, float ms[ORBFE_T_COUNT])
{
    if (!h || !ms) return ORBFE_ERR_ARG;
    if (h->prof_calls == 0) { orbfe_set_error("no profiled call yet"); return ORBFE_ERR_STATE; }
    DeviceGuard g(h->device);
    const int ncalls = std::min(h->prof_calls, ORBFE_PROF_RING);
    double acc[ORBFE_T_COUNT] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < ncalls; ++c) {
        hipEvent_t *e = h->ev[(h->prof_calls - 1 - c) % ORBFE_PROF_RING];
        ORBFE_HIP(hipEventSynchronize(e[5]));
        ORBFE_HIP(hipEventSynchronize(e[7]));
        static const int from[ORBFE_T_COUNT] = {0, 1, 2, 6, 4, 0}, to[ORBFE_T_COUNT] = {1, 2, 3, 7, 5, 5};
        for (int i = 0; i < ORBFE_T_COUNT; ++i) {
            float t;
            ORBFE_HIP(hipEventElapsedTime(&t, e[from[i]], e[to[i]]));
            acc[i] += t;
        }
    }
    for (int i = 0; i < ORBFE_T_COUNT; ++i) ms[i] = (float)(acc[i] / ncalls);
    return ORBFE_OK;
}

extern "C" void *orbfe_get_stream(orbfe_handle *h) { return h ? (void *)h->stream : nullptr; }

extern "C" orbfe_status orbfe_synchronize(orbfe_handle *h)
{
    if (!h) return ORBFE_ERR_ARG;
    DeviceGuard g(h->device);
    ORBFE_HIP(hipStreamSynchronize(h->stream));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_get_work_counts(const orbfe_handle *h, int64_t out[2])
{
    if (!h || !out) return ORBFE_ERR_ARG;
    out[0] = h->fast_row_steps;
    out[1] = h->plan.nfwaves;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_get_overflow(orbfe_handle *h, int32_t *flags)
{
    if (!h || !flags) return ORBFE_ERR_ARG;
    DeviceGuard g(h->device);
    return read_overflow(h, flags);
}

// developer builds (-DQT_PROFILE): the 1024-byte block behind the overflow word; reset != 0 clears everything but the word
extern "C" orbfe_status orbfe_internal_read_misc(orbfe_handle *h, void *out, int32_t reset)
{
    if (!h || !out || !h->d_misc.p) return ORBFE_ERR_ARG;
    DeviceGuard g(h->device);
    ORBFE_HIP(wait_last_call(h));
    ORBFE_HIP(hipMemcpy(out, h->d_misc.p, 1024, hipMemcpyDeviceToHost));
    if (reset) ORBFE_HIP(hipMemset((char *)h->d_misc.p + 64, 0, 960));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_set_option(orbfe_handle *h, int32_t option, int32_t value)
{
    if (!h) return ORBFE_ERR_ARG;
    auto in = [&](int lo, int hi) { return value >= lo && value <= hi; };
    auto developer_only = [&]() {
#ifdef ORBFE_DEVELOPER
        return false;
#else
        if (value == 0) return false;   // "off" is always accepted
        orbfe_set_error("option %d selects a kernel variant that is compiled into developer builds (-DORBFE_DEVELOPER) only", option);
        return true;
#endif
    };
    bool replan = false;
    switch (option) {
    case ORBFE_OPT_OVERLAP: if (!in(-1, 2)) return ORBFE_ERR_ARG; h->overlap = value; break;
    case ORBFE_OPT_ROWS: if (value && !in(8, 512)) return ORBFE_ERR_ARG; h->opt_rows = value; replan = true; break;
    case ORBFE_OPT_ROWS_FAST: if (value && !in(8, 512)) return ORBFE_ERR_ARG; h->opt_rows_fast = value; replan = true; break;
    case ORBFE_OPT_ROWS_BLUR: if (value && !in(8, 512)) return ORBFE_ERR_ARG; h->opt_rows_blur = value; replan = true; break;
    case ORBFE_OPT_BLUR_PIECES: if (!in(0, 1)) return ORBFE_ERR_ARG; h->opt_blur_pieces = value; replan = true; break;
    case ORBFE_OPT_BLUR_UPDOWN: if (!in(0, 2)) return ORBFE_ERR_ARG; h->opt_blur_updown = value; replan = true; break;
    case ORBFE_OPT_PYR_ROWS: if (value && !in(2, ORBFE_PW_ROWS)) return ORBFE_ERR_ARG; h->kopts.pw_rows = value; break;
    case ORBFE_OPT_QT_THREADS_0:
    case ORBFE_OPT_QT_THREADS_1:
    case ORBFE_OPT_QT_THREADS_2:
        if (value && (!in(64, 512) || value % 64)) return ORBFE_ERR_ARG;
        h->kopts.qt[option - ORBFE_OPT_QT_THREADS_0] = value;
        break;
    case ORBFE_OPT_DEBUG:
#ifdef ORBFE_DEVELOPER
        if (value != 0 && value != 50 && value != 51 && value != 60) return ORBFE_ERR_ARG;   // 60: timing-only, FAST without arcs
#else
        if (value != 0 && value != 50 && value != 51) return ORBFE_ERR_ARG;
#endif
        h->opt_debug = value;
        replan = true;
        break;
    case ORBFE_OPT_PYR_FUSE: if (!in(0, 1)) return ORBFE_ERR_ARG; if (developer_only()) return ORBFE_ERR_STATE; h->kopts.pyr_fuse = value; break;
    case ORBFE_OPT_FUSE_BLUR_PYR:
        if (!in(0, 2)) return ORBFE_ERR_ARG;
        if (developer_only()) return ORBFE_ERR_STATE;
        h->fuse_blur_pyr = value;
        if (value) h->fuse_fast_pyr = 0;   // one fusion at a time
        replan = true;
        break;
    case ORBFE_OPT_FUSE_FAST_PYR:
        if (!in(0, 3)) return ORBFE_ERR_ARG;
        if (developer_only()) return ORBFE_ERR_STATE;
        h->fuse_fast_pyr = value;
        if (value && h->fuse_blur_pyr) { h->fuse_blur_pyr = 0; replan = true; }
        break;
    case ORBFE_OPT_FUSE_FAST_PYR_LEVELS:   // matters for the developer-only variant ORBFE_OPT_FUSE_FAST_PYR alone; 0 = all levels (the default)
        if (!in(0, ORBFE_MAX_LEVELS)) return ORBFE_ERR_ARG;
        if (developer_only()) return ORBFE_ERR_STATE;
        h->fuse_fast_pyr_levels = value ? value : ORBFE_MAX_LEVELS;
        break;
    case ORBFE_OPT_BLUR_ROUNDING: if (!in(0, 1)) return ORBFE_ERR_ARG; h->prm.blur_rounding = value; replan = true; break;
    case ORBFE_OPT_REUSE_IDENTICAL_INPUT: if (!in(0, 1)) return ORBFE_ERR_ARG; h->opt_reuse = value; break;
    default: orbfe_set_error("unknown option %d", option); return ORBFE_ERR_ARG;
    }
    if (replan) h->plan_valid = false;   // rebuilt (behind the handle's outstanding work) by the next call
    h->reuse_valid = false;
    return ORBFE_OK;
}

extern "C" int32_t orbfe_last_call_reused(const orbfe_handle *h) { return h && h->last_reused ? 1 : 0; }

extern "C" orbfe_status orbfe_set_fast_mode(orbfe_handle *h, int32_t mode, int32_t collect_stats)
{
    if (!h || mode < 0 || mode > 3) return ORBFE_ERR_ARG;
    h->fast_mode = mode;
    h->fast_stats = collect_stats != 0;
    h->auto_dense_left = 0;
    h->auto_hold = ORBFE_AUTO_HOLD_MIN;
    h->auto_since = 0;
    h->auto_form = 2;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_get_fast_stats(orbfe_handle *h, uint64_t out[3], int32_t reset)
{
    if (!h || !out) return ORBFE_ERR_ARG;
    out[0] = out[1] = out[2] = 0;
    if (!h->d_misc.p) return ORBFE_OK;
    DeviceGuard g(h->device);
    ORBFE_HIP(wait_last_call(h));
    if (h->fast_mode == 3) {   // auto: the counters belong to the mode selection; report its last completed probe
        if (h->auto_pending && hipEventQuery(h->ev_auto) == hipSuccess) {
            memcpy(h->auto_last, h->h_auto.p, sizeof(h->auto_last));
            h->auto_pending = false;
        }
        for (int i = 0; i < 3; ++i) out[i] = h->auto_last[i];
        return ORBFE_OK;
    }
    ORBFE_HIP(hipMemcpy(out, (char *)h->d_misc.p + 16, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) ORBFE_HIP(hipMemset((char *)h->d_misc.p + 16, 0, 3 * sizeof(uint64_t)));
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// the batched device path (everything else funnels into this)
// ---------------------------------------------------------------------------------------------------
static orbfe_status run_batch(orbfe_handle *h, const uint8_t *d_gray, int nframes, int w, int ht, int stride,
                              size_t frame_stride, orbfe_keypoint *d_kps, uint8_t *d_desc, int cap,
                              int32_t *d_n_out, hipStream_t st)
{
    if (w > h->prm.max_width || ht > h->prm.max_height) {
        orbfe_set_error("frame %dx%d larger than planned %dx%d", w, ht, h->prm.max_width, h->prm.max_height);
        return ORBFE_ERR_SIZE;
    }
    h->reuse_valid = false;   // the scratch blocks and taps belong to this call from here on
    h->last_reused = false;
    orbfe_status s = build_plan(h, w, ht);
    if (s != ORBFE_OK) return s;
    s = ensure_batch_buffers(h, nframes);
    if (s != ORBFE_OK) return s;
    OrbLaunch a;
    a.opts = h->kopts;
    a.h_plan = &h->plan;
    a.d_plan = (const OrbPlan *)h->d_plan.p;
    a.d_tabs = (const OrbTab *)h->d_tabs.p;
    a.d_flanes = (const OrbLane *)h->d_flanes.p;
    a.d_flanes_c = (const OrbLane *)h->d_flanes_c.p;
    a.d_blanes = (const OrbLane *)h->d_blanes.p;
    a.d_blanesR = (const OrbLaneR *)h->d_blanesR.p;
    a.nframes = nframes;
    a.d_gray = d_gray;
    a.gray_fstride = (int64_t)frame_stride;
    a.gray_pitch = stride;
    a.d_pyr = (uint8_t *)h->d_pyr.p;
    a.d_blur = (uint8_t *)h->d_blur.p;
    a.pyr_fstride = h->plan.pyr_frame_bytes;
    a.d_skeys = (uint2 *)h->d_skeys.p;
    a.d_scount = (int32_t *)h->d_scount.p;
    a.d_cflag = (uint32_t *)(a.d_scount + (size_t)nframes * h->plan.nlevels * ORBFE_NK_STRIDE);  // right behind this call's counts
    a.cf_words = (h->plan.max_ncells + 31) / 32;
    a.d_knode = (uint16_t *)h->d_knode.p;
    a.d_qtbox = (int16_t *)h->d_qtbox.p;
    a.qtbox_stride = (int32_t)(orbk_octree_box_bytes(h->plan.node_cap) / sizeof(int16_t));
    a.d_qtnodes = (char *)h->d_qtnodes.p;
    a.qtnodes_stride = (int64_t)orbk_octree_node_bytes(h->plan.node_cap);
    a.d_sel = (uint32_t *)h->d_sel.p;
    a.d_nsel = (int32_t *)h->d_nsel.p;
    a.d_nkeys = (int32_t *)h->d_nkeys.p;
    a.d_kps = d_kps;
    a.d_desc = d_desc;
    a.cap = cap;
    a.d_n_out = d_n_out;
    a.d_ovf = (int32_t *)h->d_misc.p;
    // FAST form of this call.  Auto (3, the default): dense for calls of fewer than ORBFE_AUTO_MIN_ROW_STEPS wave row steps (about 29
    // VGA frames); otherwise
    // lane-compacting unless the last probe found more than ORBFE_AUTO_DENSE_RATE of the pixel pairs passing the necessary test --
    // then dense for the next auto_hold calls (16, doubling up to 256 while the probes keep saying so), after which one compacting
    // call probes again.
    int fmode = h->fast_mode;
    bool auto_probe = false;
    if (fmode == 3 && (int64_t)nframes * h->fast_row_steps < ORBFE_AUTO_MIN_ROW_STEPS) {
        fmode = 0;
    } else if (fmode == 3) {
        if (h->auto_pending && hipEventQuery(h->ev_auto) == hipSuccess) {   // never waits
            memcpy(h->auto_last, h->h_auto.p, sizeof(h->auto_last));
            h->auto_pending = false;
            if (h->auto_last[0] > 0 && (double)h->auto_last[2] > ORBFE_AUTO_DENSE_RATE * 128.0 * (double)h->auto_last[0]) {
                h->auto_form = 0;
                h->auto_dense_left = h->auto_hold;
                h->auto_hold = std::min(2 * h->auto_hold, ORBFE_AUTO_HOLD_MAX);
            } else {
                h->auto_form = 2;
                h->auto_hold = ORBFE_AUTO_HOLD_MIN;
                h->auto_since = 1;
            }
        } else {
            (void)hipGetLastError();   // hipErrorNotReady is not an error of ours
        }
        // A host that runs ahead of the GPU enqueues many calls before a probe's answer arrives: those follow the LAST answer
        // (auto_form), only the probe call itself is compacting when that answer was "dense".
        if (h->auto_form == 0) {
            if (h->auto_dense_left > 0) {
                h->auto_dense_left--;
                fmode = 0;
            } else if (!h->auto_pending) {
                fmode = 2;
                auto_probe = true;
            } else {
                fmode = 0;
            }
        } else {
            fmode = 2;
            auto_probe = !h->auto_pending && (h->auto_since++ % ORBFE_AUTO_PROBE_EVERY) == 0;
        }
    }
#ifdef ORBFE_DEVELOPER
    if (h->fuse_fast_pyr && fmode >= 2) {   // the fused FAST + pyramid kernels exist in the dense forms only
        fmode = 0;
        auto_probe = false;
    }
#endif
    a.fast_sparse = fmode;
    a.d_fstat = (h->fast_stats || auto_probe) ? (unsigned long long *)((char *)h->d_misc.p + 16) : nullptr;
    // every call of a handle uses the same scratch blocks (pyramid, blur, survivor lists, selections): a call on another
    // stream than its predecessor's waits, at stream level, for that predecessor to finish
    if (h->last_stream_valid && h->last_stream != st) ORBFE_HIP(hipStreamWaitEvent(st, h->ev_last, 0));
    hipEvent_t *ev = h->profiling ? h->ev[h->prof_calls % ORBFE_PROF_RING] : nullptr;
    if (ev) ORBFE_HIP(hipEventRecord(ev[0], st));
    auto finish = [&]() -> orbfe_status {   // common tail: profiling bookkeeping, the "last call" state of the handle
        if (ev) {
            ORBFE_HIP(hipEventRecord(ev[5], st));
            h->prof_calls++;
        }
        ORBFE_HIP(hipEventRecord(h->ev_last, st));
        h->last_stream = st;
        h->last_stream_valid = true;
        h->last_gray = d_gray;
        h->last_gray_fstride = (int64_t)frame_stride;
        h->last_gray_pitch = stride;
        h->last_nframes = nframes;
        return ORBFE_OK;
    };
    int ov = h->overlap >= 0 ? h->overlap : (nframes >= 128 ? 2 : 0);
#ifdef ORBFE_DEVELOPER
    if (h->fuse_blur_pyr) {
        // blur(l) and resize(l -> l + 1) in one pass over level l, chained over the levels: level l is read from HBM once for
        // both.  The stage table then shows the fused chain under "pyramid" and nothing under "blur"; `overlap` does not apply
        // (there is no separate blur to put beside anything).
        ORBFE_HIP(orbk_launch_blur_pyr(a, st));
        if (ev) {
            ORBFE_HIP(hipEventRecord(ev[1], st));
            ORBFE_HIP(hipEventRecord(ev[6], st));
            ORBFE_HIP(hipEventRecord(ev[7], st));
        }
        ORBFE_HIP(orbk_launch_fast(a, st));
        if (ev) ORBFE_HIP(hipEventRecord(ev[2], st));
        ORBFE_HIP(orbk_launch_octree(a, st));
        if (ev) {
            ORBFE_HIP(hipEventRecord(ev[3], st));
            ORBFE_HIP(hipEventRecord(ev[4], st));
        }
        ORBFE_HIP(orbk_launch_describe(a, st));
        return finish();
    }
    // ORBFE_OPT_FUSE_FAST_PYR 1 / 2: FAST(l) and resize(l -> l + 1) in one launch per level (k_fast_pyr); the stage table then
    // shows the whole chain under "fast" and nothing under "pyramid".  3: no fused kernel -- FAST of level 0, which needs no
    // pyramid, runs on the side stream BESIDE the pyramid chain (two FAST waves leave room for four resize waves on a SIMD),
    // FAST of the other levels after both; it has its own event pair (ev_fork2 / ev_join2), the blur fork keeps ev_fork / ev_join
    const bool ffp = h->fuse_fast_pyr == 1 || h->fuse_fast_pyr == 2;
    const bool fside = h->fuse_fast_pyr == 3;
    if (fside) {
        ORBFE_HIP(orbk_launch_fast_levels(a, 0, 0, 1, st));   // the clear only
        ORBFE_HIP(hipEventRecord(h->ev_fork2, st));
        ORBFE_HIP(hipStreamWaitEvent(h->side, h->ev_fork2, 0));
        ORBFE_HIP(orbk_launch_fast_levels(a, 0, 1, 0, h->side));
        ORBFE_HIP(hipEventRecord(h->ev_join2, h->side));
    }
    if ((ffp || fside) && ov == 1) ov = 2;   // the blur needs the whole pyramid, which the fused chain finishes last
#else
    const bool ffp = false, fside = false;
    (void)fside;
#endif
    if (!ffp) ORBFE_HIP(orbk_launch_pyramid(a, st));
    if (ev) ORBFE_HIP(hipEventRecord(ev[1], st));
    auto fork_blur = [&]() -> hipError_t {
        hipError_t e = hipEventRecord(h->ev_fork, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->side, h->ev_fork, 0);
        if (e == hipSuccess && ev) e = hipEventRecord(ev[6], h->side);
        if (e == hipSuccess) e = orbk_launch_blur(a, h->side);
        if (e == hipSuccess && ev) e = hipEventRecord(ev[7], h->side);
        if (e == hipSuccess) e = hipEventRecord(h->ev_join, h->side);
        return e;
    };
    if (ov == 1) ORBFE_HIP(fork_blur());
#ifdef ORBFE_DEVELOPER
    if (ffp) ORBFE_HIP(orbk_launch_fast_pyr(a, h->fuse_fast_pyr_levels, h->fuse_fast_pyr == 2, st));
    else if (fside) {
        ORBFE_HIP(orbk_launch_fast_levels(a, 1, h->plan.nlevels, 0, st));
        ORBFE_HIP(hipStreamWaitEvent(st, h->ev_join2, 0));
    } else
#endif
        ORBFE_HIP(orbk_launch_fast(a, st));
    if (auto_probe) {   // the sampled counters of this launch -> pinned host memory; a later call looks at them
        ORBFE_HIP(hipMemcpyAsync(h->h_auto.p, (char *)h->d_misc.p + 16, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        ORBFE_HIP(hipMemsetAsync((char *)h->d_misc.p + 16, 0, 3 * sizeof(uint64_t), st));
        ORBFE_HIP(hipEventRecord(h->ev_auto, st));
        h->auto_pending = true;
    }
    if (ev) ORBFE_HIP(hipEventRecord(ev[2], st));
    if (ov == 2) ORBFE_HIP(fork_blur());
    ORBFE_HIP(orbk_launch_octree(a, st));
    if (ev) ORBFE_HIP(hipEventRecord(ev[3], st));
    if (ov == 0) {
        if (ev) ORBFE_HIP(hipEventRecord(ev[6], st));
        ORBFE_HIP(orbk_launch_blur(a, st));
        if (ev) ORBFE_HIP(hipEventRecord(ev[7], st));
    } else {
        ORBFE_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
    }
    if (ev) ORBFE_HIP(hipEventRecord(ev[4], st));
    ORBFE_HIP(orbk_launch_describe(a, st));
    return finish();
}

extern "C" orbfe_status orbfe_extract_batch_device(orbfe_handle *h, const uint8_t *d_gray, int32_t nframes,
                                                   int32_t w, int32_t ht, int32_t stride, size_t frame_stride,
                                                   orbfe_keypoint *d_kps, uint8_t *d_desc, int32_t cap,
                                                   int32_t *d_n_out, void *stream)
{
    if (!h || !d_gray || !d_kps || !d_desc || !d_n_out || nframes < 1 || w < 1 || ht < 1 || stride < w || cap < 1 ||
        frame_stride < (size_t)stride * (size_t)(ht - 1) + (size_t)w) {
        orbfe_set_error("bad argument to orbfe_extract_batch_device");
        return ORBFE_ERR_ARG;
    }
    DeviceGuard g(h->device);
    return run_batch(h, d_gray, nframes, w, ht, stride, frame_stride, d_kps, d_desc, cap, d_n_out,
                     (hipStream_t)stream);
}

// host buffers, in chunks of max_batch frames.  A single chunk (the online case: one frame) is a plain H2D -> kernels ->
// D2H sequence on the handle's stream.  Several chunks run as a pipeline over two buffer sets and three streams: while
// the kernels of chunk i run, chunk i+1 is copied in (s_in) and chunk i-1 is copied out (s_out) and unpacked by the host.
// Frames in pinned (page-locked / hipHostRegister'ed) memory with stride == w are copied straight from the caller's
// buffers; pageable frames are first gathered into the pinned staging set (the host memcpy then overlaps the GPU work).
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'ed memory: "invalid value", not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

static orbfe_status extract_host(orbfe_handle *h, const uint8_t *const *grays, int nframes, int w, int ht,
                                 int stride, orbfe_keypoint *kps, uint8_t *desc, int cap, int32_t *n_out)
{
    if (!h->own_stream) {
        orbfe_set_error("this extractor is a pipe of an orbfe_pipeline: use orbfe_pipeline_extract_match for host buffers");
        return ORBFE_ERR_STATE;
    }
    DeviceGuard g(h->device);
    const int chunk_max = h->prm.max_batch;
    const int nchunks = (nframes + chunk_max - 1) / chunk_max;
    // The page-locked probes (hipPointerGetAttributes: an error path for plain malloc'ed memory) cost more than they can
    // save on the online single-frame call, which goes through the handle's own pinned staging either way.
    const bool probe = nframes > 1;
    const bool direct = probe && stride == w && is_pinned_host(grays[0]) && is_pinned_host(grays[nframes - 1]);
    // pinned output arrays receive the padded device blocks as they are (slots >= n_out[f] zero-filled): no host unpacking
    const bool direct_out = probe && is_pinned_host(kps) && is_pinned_host(desc) && is_pinned_host(n_out);
    const int pitch = direct ? w : orb_align_up(w, 64);
    const size_t fbytes = (size_t)pitch * ht;
    const int nset = nchunks > 1 ? 2 : 1;
    const size_t nbmax = (size_t)std::min(chunk_max, nframes);
    // Output set k on the device.  When the results go through the handle's pinned staging (the online call), the three
    // arrays are pieces of ONE block -- keypoints | descriptors | counts -- so that they come back in one D2H copy instead
    // of three (each copy is a ~7 us round trip on the stream of a call that takes 0.25 ms in all).
    const size_t kb = (sizeof(orbfe_keypoint) * (size_t)cap * nbmax + 255) & ~(size_t)255;
    const size_t db = ((size_t)32 * cap * nbmax + 255) & ~(size_t)255;
    const size_t cb = (sizeof(int32_t) * nbmax + 255) & ~(size_t)255;
    // ORBFE_OPT_REUSE_IDENTICAL_INPUT: perfect/src/Tracking.cc:685 and :716 build two Frames from the SAME mImGray with the same
    // extractor (the second with the dynamic-object mask, which operator() ignores): the second extraction is the first one's
    // result.  The previous frame sits in the pinned staging block (pitch-aligned rows) and its results in the pinned result
    // block; one pass of memcmp over the rows decides, and a hit touches neither the link nor the GPU (pyramid, taps and the
    // device-side state of the handle are still those of that frame).  Bit-exact by construction.
    if (h->opt_reuse && nframes == 1 && h->reuse_valid && h->reuse_w == w && h->reuse_h == ht && h->reuse_cap == cap && h->plan_valid) {
        const uint8_t *prev = (const uint8_t *)h->h_stage[0].p, *src = grays[0];
        bool same = true;
        if (stride == w && pitch == w) same = memcmp(prev, src, fbytes) == 0;
        else
            for (int y = 0; y < ht && same; ++y) same = memcmp(prev + (size_t)y * pitch, src + (size_t)y * stride, (size_t)w) == 0;
        if (same) {
            const uint8_t *hb = (const uint8_t *)h->h_okps[0].p;   // keypoints | descriptors | counts of that call
            const int n = ((const int32_t *)(hb + kb + db))[0];
            n_out[0] = n;
            memcpy(kps, hb, sizeof(orbfe_keypoint) * (size_t)n);
            memcpy(desc, hb + kb, (size_t)32 * n);
            h->last_reused = true;
            h->reuse_hits++;
            return ORBFE_OK;
        }
    }
    uint8_t *d_ok[2] = {nullptr, nullptr}, *d_od[2] = {nullptr, nullptr}, *d_oc[2] = {nullptr, nullptr};
    for (int k = 0; k < nset; ++k) {
        if (!direct) ORBFE_HIP(h->h_stage[k].ensure(fbytes * nbmax));
        ORBFE_HIP(h->d_stage[k].ensure(fbytes * nbmax + 64));
        if (direct_out) {
            ORBFE_HIP(h->d_okps[k].ensure(kb));
            ORBFE_HIP(h->d_odesc[k].ensure(db));
            ORBFE_HIP(h->d_on[k].ensure(cb));
            d_ok[k] = (uint8_t *)h->d_okps[k].p; d_od[k] = (uint8_t *)h->d_odesc[k].p; d_oc[k] = (uint8_t *)h->d_on[k].p;
        } else {
            ORBFE_HIP(h->d_okps[k].ensure(kb + db + cb));
            ORBFE_HIP(h->h_okps[k].ensure(kb + db + cb));
            d_ok[k] = (uint8_t *)h->d_okps[k].p; d_od[k] = d_ok[k] + kb; d_oc[k] = d_od[k] + db;
        }
    }
    // size the plan and the per-batch blocks once, before anything is in flight
    if (w > h->prm.max_width || ht > h->prm.max_height) {
        orbfe_set_error("frame %dx%d larger than planned %dx%d", w, ht, h->prm.max_width, h->prm.max_height);
        return ORBFE_ERR_SIZE;
    }
    orbfe_status sp = build_plan(h, w, ht);
    if (sp != ORBFE_OK) return sp;
    sp = ensure_batch_buffers(h, (int)nbmax);
    if (sp != ORBFE_OK) return sp;

    const bool piped = nchunks > 1;
    hipStream_t s_in = piped ? h->s_in : h->stream, s_cmp = h->stream, s_out = piped ? h->s_out : h->stream;
    orbfe_status worst = ORBFE_OK;
    auto drain = [&]() {
        (void)hipStreamSynchronize(h->s_in);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->s_out);
    };
    // Every early return below leaves copies / kernels in flight that write into the caller's arrays (direct_out) or read
    // its frames (direct): whatever way the pipeline loop is left, the three streams are drained first.
    struct DrainGuard {
        decltype(drain) &fn;
        bool armed = true;
        ~DrainGuard() { if (armed) fn(); }
    } drain_guard{drain};
    auto unpack = [&](int c) -> orbfe_status {  // results of chunk c: wait for its D2H, hand them to the caller
        const int k = c & (nset - 1), f0 = c * chunk_max, nb = std::min(chunk_max, nframes - f0);
        ORBFE_HIP(hipEventSynchronize(h->ev_out[k]));
        if (direct_out) {
            for (int f = 0; f < nb; ++f)
                if (n_out[f0 + f] > cap) worst = ORBFE_ERR_CAP;
            return ORBFE_OK;
        }
        const uint8_t *hb = (const uint8_t *)h->h_okps[k].p;   // keypoints | descriptors | counts
        for (int f = 0; f < nb; ++f) {
            const int n = ((const int32_t *)(hb + kb + db))[f];
            n_out[f0 + f] = n;
            if (n > cap) { worst = ORBFE_ERR_CAP; continue; }
            memcpy(kps + (size_t)(f0 + f) * cap, (const orbfe_keypoint *)hb + (size_t)f * cap, sizeof(orbfe_keypoint) * (size_t)n);
            memcpy(desc + (size_t)(f0 + f) * cap * 32, hb + kb + (size_t)f * cap * 32, (size_t)32 * n);
        }
        return ORBFE_OK;
    };
    for (int c = 0; c < nchunks; ++c) {
        const int k = c & (nset - 1), f0 = c * chunk_max, nb = std::min(chunk_max, nframes - f0);
        if (c >= 2) {  // set k was last used by chunk c-2: collect its results before its buffers are reused
            orbfe_status su = unpack(c - 2);
            if (su != ORBFE_OK) { drain(); return su; }
        }
        // ---- in ----
        if (piped && c >= 2) ORBFE_HIP(hipStreamWaitEvent(s_in, h->ev_cmp[k], 0));  // kernels of chunk c-2 read d_stage[k]
        if (direct) {
            bool contiguous = true;
            for (int f = 1; f < nb && contiguous; ++f) contiguous = grays[f0 + f] == grays[f0] + fbytes * f;
            if (contiguous) {
                ORBFE_HIP(hipMemcpyAsync(h->d_stage[k].p, grays[f0], fbytes * nb, hipMemcpyHostToDevice, s_in));
            } else {
                for (int f = 0; f < nb; ++f)
                    ORBFE_HIP(hipMemcpyAsync((uint8_t *)h->d_stage[k].p + fbytes * f, grays[f0 + f], fbytes, hipMemcpyHostToDevice, s_in));
            }
        } else {
            // h_stage[k] was read by the H2D of chunk c-2, which the kernels of chunk c-2 waited for and whose results
            // were just unpacked: free to overwrite
            for (int f = 0; f < nb; ++f) {
                uint8_t *dst = (uint8_t *)h->h_stage[k].p + fbytes * f;
                const uint8_t *src = grays[f0 + f];
                if (stride == w && pitch == w) memcpy(dst, src, fbytes);
                else
                    for (int y = 0; y < ht; ++y) memcpy(dst + (size_t)y * pitch, src + (size_t)y * stride, (size_t)w);
            }
            ORBFE_HIP(hipMemcpyAsync(h->d_stage[k].p, h->h_stage[k].p, fbytes * nb, hipMemcpyHostToDevice, s_in));
        }
        if (piped) {
            ORBFE_HIP(hipEventRecord(h->ev_in[k], s_in));
            ORBFE_HIP(hipStreamWaitEvent(s_cmp, h->ev_in[k], 0));
            if (c >= 2) ORBFE_HIP(hipStreamWaitEvent(s_cmp, h->ev_out[k], 0));  // D2H of chunk c-2 read the output set k
        }
        // ---- kernels ----
        orbfe_status s = run_batch(h, (const uint8_t *)h->d_stage[k].p, nb, w, ht, pitch, fbytes, (orbfe_keypoint *)d_ok[k], d_od[k], cap,
                                   (int32_t *)d_oc[k], s_cmp);
        if (s != ORBFE_OK) { drain(); return s; }
        if (piped) {
            ORBFE_HIP(hipEventRecord(h->ev_cmp[k], s_cmp));
            ORBFE_HIP(hipStreamWaitEvent(s_out, h->ev_cmp[k], 0));
        }
        // ---- out ----
        if (direct_out) {
            ORBFE_HIP(hipMemcpyAsync(n_out + f0, d_oc[k], sizeof(int32_t) * nb, hipMemcpyDeviceToHost, s_out));
            ORBFE_HIP(hipMemcpyAsync(kps + (size_t)f0 * cap, d_ok[k], sizeof(orbfe_keypoint) * (size_t)cap * nb, hipMemcpyDeviceToHost, s_out));
            ORBFE_HIP(hipMemcpyAsync(desc + (size_t)f0 * cap * 32, d_od[k], (size_t)32 * cap * nb, hipMemcpyDeviceToHost, s_out));
        } else {
            ORBFE_HIP(hipMemcpyAsync(h->h_okps[k].p, d_ok[k], kb + db + sizeof(int32_t) * nb, hipMemcpyDeviceToHost, s_out));
        }
        if (c == nchunks - 1) {
            // the sticky overflow word travels with the results of the last chunk: an asynchronous 4-byte copy into pinned
            // memory behind the last kernels, in flight BEFORE the host starts waiting for results
            ORBFE_HIP(h->h_ovf.ensure(sizeof(int32_t)));
            ORBFE_HIP(hipMemcpyAsync(h->h_ovf.p, h->d_misc.p, sizeof(int32_t), hipMemcpyDeviceToHost, s_cmp));
        }
        ORBFE_HIP(hipEventRecord(h->ev_out[k], s_out));
    }
    for (int c = std::max(0, nchunks - 2); c < nchunks; ++c) {
        orbfe_status su = unpack(c);
        if (su != ORBFE_OK) { drain(); return su; }
    }
    ORBFE_HIP(hipStreamSynchronize(s_cmp));   // the overflow word (enqueued with the last chunk) has landed
    drain_guard.armed = false;  // everything has been waited for (unpack() synchronised the output events)
    {
        const int32_t ovf = *(const int32_t *)h->h_ovf.p;
        if (ovf) ORBFE_HIP(hipMemset(h->d_misc.p, 0, sizeof(int32_t)));
        if (ovf & 3) {
            orbfe_set_error("internal capacity exceeded (flags %d: 1 = FAST survivor list, 2 = quadtree selection); "
                            "results of this batch are incomplete", ovf);
            return ORBFE_ERR_CAP;
        }
    }
    if (worst == ORBFE_ERR_CAP) orbfe_set_error("cap=%d too small; n_out holds the required counts", cap);
    if (worst == ORBFE_OK && nframes == 1 && !direct && !direct_out) {   // what a later identical frame can be answered from
        h->reuse_valid = true;
        h->reuse_w = w;
        h->reuse_h = ht;
        h->reuse_cap = cap;
    }
    return worst;
}

extern "C" orbfe_status orbfe_extract(orbfe_handle *h, const uint8_t *gray, int32_t w, int32_t ht, int32_t stride,
                                      orbfe_keypoint *kps, uint8_t *desc, int32_t cap, int32_t *n_out)
{
    if (!h) { orbfe_set_error("null handle"); return ORBFE_ERR_ARG; }
    if (!gray || w == 0 || ht == 0) return ORBFE_OK;  // empty image: silent return, outputs untouched (:1055-1056)
    if (!kps || !desc || !n_out || w < 0 || ht < 0 || stride < w || cap < 1) {
        orbfe_set_error("bad argument to orbfe_extract");
        return ORBFE_ERR_ARG;
    }
    return extract_host(h, &gray, 1, w, ht, stride, kps, desc, cap, n_out);
}

extern "C" orbfe_status orbfe_extract_batch(orbfe_handle *h, const uint8_t *const *grays, int32_t nframes, int32_t w,
                                            int32_t ht, int32_t stride, orbfe_keypoint *kps, uint8_t *desc,
                                            int32_t cap, int32_t *n_out)
{
    if (!h) { orbfe_set_error("null handle"); return ORBFE_ERR_ARG; }
    if (nframes == 0 || w == 0 || ht == 0) return ORBFE_OK;
    if (!grays || !kps || !desc || !n_out || nframes < 0 || w < 0 || ht < 0 || stride < w || cap < 1) {
        orbfe_set_error("bad argument to orbfe_extract_batch");
        return ORBFE_ERR_ARG;
    }
    for (int i = 0; i < nframes; ++i)
        if (!grays[i]) { orbfe_set_error("grays[%d] is null", i); return ORBFE_ERR_ARG; }
    return extract_host(h, grays, nframes, w, ht, stride, kps, desc, cap, n_out);
}

// ---------------------------------------------------------------------------------------------------
// mvImagePyramid + stage taps
// ---------------------------------------------------------------------------------------------------
static orbfe_status check_tap(orbfe_handle *h, int frame, int level)
{
    if (!h) return ORBFE_ERR_ARG;
    if (!h->plan_valid || h->last_nframes == 0) { orbfe_set_error("no extract call yet"); return ORBFE_ERR_STATE; }
    if (frame < 0 || frame >= h->last_nframes || level < 0 || level >= h->plan.nlevels) {
        orbfe_set_error("frame/level out of range");
        return ORBFE_ERR_ARG;
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_get_level_size(const orbfe_handle *h, int32_t level, int32_t *w, int32_t *ht)
{
    if (!h || !h->plan_valid || level < 0 || level >= h->plan.nlevels) return ORBFE_ERR_ARG;
    if (w) *w = h->plan.lv[level].w;
    if (ht) *ht = h->plan.lv[level].h;
    return ORBFE_OK;
}

static inline int host_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

static orbfe_status fetch_level(orbfe_handle *h, const uint8_t *base, int pitch, int w, int ht, uint8_t *dst,
                                int dst_stride, int border)
{
    std::vector<uint8_t> tmp((size_t)w * ht);
    ORBFE_HIP(wait_last_call(h));
    ORBFE_HIP(hipMemcpy2D(tmp.data(), (size_t)w, base, (size_t)pitch, (size_t)w, (size_t)ht, hipMemcpyDeviceToHost));
    for (int y = -border; y < ht + border; ++y) {
        const uint8_t *s = tmp.data() + (size_t)host_reflect101(y, ht) * w;
        uint8_t *d = dst + (size_t)(y + border) * dst_stride;
        if (border == 0) memcpy(d, s, (size_t)w);
        else
            for (int x = -border; x < w + border; ++x) d[x + border] = s[host_reflect101(x, w)];  // :1136-1142
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_get_pyramid_level(orbfe_handle *h, int32_t frame, int32_t level, uint8_t *dst,
                                                int32_t dst_stride, int32_t with_border)
{
    orbfe_status s = check_tap(h, frame, level);
    if (s != ORBFE_OK) return s;
    if (!dst) return ORBFE_ERR_ARG;
    DeviceGuard g(h->device);
    const OrbLevel &L = h->plan.lv[level];
    const int border = with_border ? ORBFE_EDGE : 0;
    if (dst_stride < L.w + 2 * border) return ORBFE_ERR_ARG;
    if (level == 0)
        return fetch_level(h, h->last_gray + (int64_t)frame * h->last_gray_fstride, h->last_gray_pitch, L.w, L.h, dst,
                           dst_stride, border);
    return fetch_level(h, (uint8_t *)h->d_pyr.p + (int64_t)frame * h->plan.pyr_frame_bytes + L.off, L.pitch, L.w, L.h,
                       dst, dst_stride, border);
}

// The public mvImagePyramid in one go: every level of frame `frame` with its 19-px BORDER_REFLECT_101 frame, level l as a
// (w_l + 38) x (h_l + 38) block with tight rows at offsets[l] of dst.  One kernel builds the blocks on the device, ONE
// device-to-host copy brings them over (the per-level orbfe_get_pyramid_level path made 8 pageable 2-D copies and filled the
// frames on the host: 9.6 ms for a 640x480 frame against 0.18 ms for the extraction itself).
extern "C" orbfe_status orbfe_get_pyramid_padded(orbfe_handle *h, int32_t frame, uint8_t *dst, size_t cap, size_t *offsets, size_t *total)
{
    orbfe_status s = check_tap(h, frame, 0);
    if (s != ORBFE_OK) return s;
    DeviceGuard g(h->device);
    const int nl = h->plan.nlevels;
    uint32_t off[ORBFE_MAX_LEVELS + 1];
    uint32_t at = 0;
    for (int l = 0; l < nl; ++l) {
        off[l] = at;
        const OrbLevel &L = h->plan.lv[l];
        at += (uint32_t)(((size_t)(L.w + 2 * ORBFE_EDGE) * (size_t)(L.h + 2 * ORBFE_EDGE) + 63) & ~(size_t)63);
    }
    off[nl] = at;
    if (offsets)
        for (int l = 0; l < nl; ++l) offsets[l] = off[l];
    if (total) *total = at;
    if (!dst) return ORBFE_OK;   // sizing call
    if (cap < at) { orbfe_set_error("orbfe_get_pyramid_padded: %zu bytes needed, %zu given", (size_t)at, cap); return ORBFE_ERR_CAP; }
    OrbPyrView v;
    s = orbfe_internal_pyramid_view(h, frame, &v);
    if (s != ORBFE_OK) return s;
    ORBFE_HIP(wait_last_call(h));
    ORBFE_HIP(h->d_pad.ensure(at));
    ORBFE_HIP(orbk_launch_pad_pyramid(v, off, (uint8_t *)h->d_pad.p, h->stream));
    ORBFE_HIP(hipMemcpyAsync(dst, h->d_pad.p, at, hipMemcpyDeviceToHost, h->stream));
    ORBFE_HIP(hipStreamSynchronize(h->stream));
    return ORBFE_OK;
}

// makes `stream` (a hipStream_t) wait for the handle's last batched call, wherever it ran: the pyramid readers of the
// matcher (stereo) order themselves behind the extractor with it, without a host synchronisation
int32_t orbfe_internal_order_after_last_call(orbfe_handle *h, void *stream)
{
    if (!h) return ORBFE_ERR_ARG;
    if (!h->last_stream_valid || h->last_stream == (hipStream_t)stream) return ORBFE_OK;
    DeviceGuard g(h->device);
    ORBFE_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_last, 0));
    return ORBFE_OK;
}

int32_t orbfe_internal_pyramid_view(orbfe_handle *h, int frame, OrbPyrView *v)
{
    orbfe_status s = check_tap(h, frame, 0);
    if (s != ORBFE_OK) return s;
    DeviceGuard g(h->device);
    v->nlevels = h->plan.nlevels;
    v->device = h->device;
    for (int l = 0; l < h->plan.nlevels; ++l) {
        const OrbLevel &L = h->plan.lv[l];
        v->ptr[l] = l == 0 ? h->last_gray + (int64_t)frame * h->last_gray_fstride
                           : (const uint8_t *)h->d_pyr.p + (int64_t)frame * h->plan.pyr_frame_bytes + L.off;
        v->pitch[l] = l == 0 ? h->last_gray_pitch : L.pitch;
        v->w[l] = L.w;
        v->h[l] = L.h;
        v->scale[l] = h->scale[l];
        v->inv_scale[l] = h->inv_scale[l];
        v->fstride[l] = l == 0 ? h->last_gray_fstride : (int64_t)h->plan.pyr_frame_bytes;
    }
    v->nframes = h->last_nframes;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_tap_blurred_level(orbfe_handle *h, int32_t frame, int32_t level, uint8_t *dst,
                                                int32_t dst_stride)
{
    orbfe_status s = check_tap(h, frame, level);
    if (s != ORBFE_OK) return s;
    if (!dst) return ORBFE_ERR_ARG;
    DeviceGuard g(h->device);
    const OrbLevel &L = h->plan.lv[level];
    if (dst_stride < L.w) return ORBFE_ERR_ARG;
    return fetch_level(h, (uint8_t *)h->d_blur.p + (int64_t)frame * h->plan.pyr_frame_bytes + L.off, L.pitch, L.w, L.h,
                       dst, dst_stride, 0);
}

extern "C" orbfe_status orbfe_tap_candidates(orbfe_handle *h, int32_t frame, int32_t level, float *xyr, int32_t cap,
                                             int32_t *n)
{
    orbfe_status s = check_tap(h, frame, level);
    if (s != ORBFE_OK) return s;
    if (!n) return ORBFE_ERR_ARG;
    DeviceGuard g(h->device);
    const OrbPlan &P = h->plan;
    const OrbLevel &L = P.lv[level];
    ORBFE_HIP(wait_last_call(h));
    int32_t nk = 0, nsv = 0;
    ORBFE_HIP(hipMemcpy(&nk, (int32_t *)h->d_nkeys.p + ((size_t)frame * P.nlevels + level) * ORBFE_NK_STRIDE, sizeof(int32_t),
                        hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(&nsv, (int32_t *)h->d_scount.p + ((size_t)frame * P.nlevels + level) * ORBFE_NK_STRIDE, sizeof(int32_t),
                        hipMemcpyDeviceToHost));
    nsv = std::min(nsv, L.key_cap);
    *n = nk;
    if (nk > cap) return ORBFE_ERR_CAP;
    if (nk == 0) return ORBFE_OK;
    if (!xyr) return ORBFE_ERR_ARG;
    // The device keeps the NMS survivors {key, ord} unordered and in place; the per-cell threshold fallback (:818-825) is
    // the same rule k_octree applies: a survivor counts if it is above iniTh or its cell has no survivor above iniTh.
    // `ord` is the rank key of the reference's candidate order.
    std::vector<uint2> sv((size_t)nsv);
    ORBFE_HIP(hipMemcpy(sv.data(), (uint2 *)h->d_skeys.p + (size_t)frame * P.keys_per_frame + L.key_off,
                        sizeof(uint2) * (size_t)nsv, hipMemcpyDeviceToHost));
    std::vector<uint8_t> strong((size_t)L.ncells, 0);
    for (const uint2 &e : sv)
        if ((int)orb_key_r(e.x) >= P.ini_th && (e.y >> 12) < (uint32_t)L.ncells) strong[e.y >> 12] = 1;
    std::vector<uint32_t> kv, ko;
    for (const uint2 &e : sv)
        if ((int)orb_key_r(e.x) >= P.ini_th || ((e.y >> 12) < (uint32_t)L.ncells && !strong[e.y >> 12])) {
            kv.push_back(e.x);
            ko.push_back(e.y);
        }
    if ((int)kv.size() != nk) {
        orbfe_set_error("candidate tap: host filter found %d keys, device counted %d", (int)kv.size(), nk);
        return ORBFE_ERR_STATE;
    }
    std::vector<int> order((size_t)nk);
    for (int i = 0; i < nk; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return ko[a] < ko[b]; });
    for (int i = 0; i < nk; ++i) {
        const uint32_t k = kv[order[i]];
        xyr[3 * i] = (float)orb_key_x(k);
        xyr[3 * i + 1] = (float)orb_key_y(k);
        xyr[3 * i + 2] = (float)orb_key_r(k);
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_tap_selected(orbfe_handle *h, int32_t frame, int32_t level, float *xyr, int32_t cap,
                                           int32_t *n)
{
    orbfe_status s = check_tap(h, frame, level);
    if (s != ORBFE_OK) return s;
    if (!n) return ORBFE_ERR_ARG;
    DeviceGuard g(h->device);
    const OrbPlan &P = h->plan;
    const OrbLevel &L = P.lv[level];
    ORBFE_HIP(wait_last_call(h));
    int32_t ns = 0;
    ORBFE_HIP(hipMemcpy(&ns, (int32_t *)h->d_nsel.p + (size_t)frame * P.nlevels + level, sizeof(int32_t),
                        hipMemcpyDeviceToHost));
    *n = ns;
    if (ns > cap) return ORBFE_ERR_CAP;
    if (ns == 0) return ORBFE_OK;
    if (!xyr) return ORBFE_ERR_ARG;
    std::vector<uint32_t> keys((size_t)ns);
    ORBFE_HIP(hipMemcpy(keys.data(), (uint32_t *)h->d_sel.p + (size_t)frame * P.sel_per_frame + L.sel_off,
                        sizeof(uint32_t) * (size_t)ns, hipMemcpyDeviceToHost));
    for (int i = 0; i < ns; ++i) {
        xyr[3 * i] = (float)orb_key_x(keys[i]);
        xyr[3 * i + 1] = (float)orb_key_y(keys[i]);
        xyr[3 * i + 2] = (float)orb_key_r(keys[i]);
    }
    return ORBFE_OK;
}
