// orbfe_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the ORB extractor.
//
// Reference behaviour restated (paths relative to /root/reference):
//   K1 k_pyr_walk        ComputePyramid + cv::resize INTER_LINEAR   src/ORBextractor.cc:1117-1145
//   K2 k_fast_map        per-cell cv::FAST + NMS + minTh fallback   src/ORBextractor.cc:798-838
//   K3 k_octree          DistributeOctTree / DivideNode             src/ORBextractor.cc:478-765
//   K4 k_blur7           GaussianBlur 7x7 sigma 2 REFLECT_101       src/ORBextractor.cc:1094-1095
//   K5 k_orient_describe IC_Angle + computeOrbDescriptor + rescale  src/ORBextractor.cc:59-131, 846-857, 1103-1110
//
// Integer / byte work bounded by VALU issue (FAST), HBM (pyramid, blur) and LDS latency (quadtree); no MFMA.  Float steps
// that decide an output bit use explicitly rounded single operations (__fmul_rn/__fadd_rn/__fdiv_rn, no FMA contraction).
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

#include "orbfe_common.h"
#include "orbfe_kernels.h"
#include "orbfe_pattern.inc"

__constant__ signed char c_pattern[1024];
// umax[v] of the circular patch (src/ORBextractor.cc:449-465): {15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3}
__constant__ int c_umax[16];

hipError_t orbk_upload_moment_weights(const int *umax16);

hipError_t orbk_upload_constants(const int *umax16)
{
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), orbfe_pattern31_host, 1024);
    if (e != hipSuccess) return e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax16, 16 * sizeof(int));
    if (e != hipSuccess) return e;
    return orbk_upload_moment_weights(umax16);
}

// ---------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------
typedef _Float16 orb_h2 __attribute__((ext_vector_type(2)));
typedef short orb_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short orb_u2 __attribute__((ext_vector_type(2)));

struct FrameSrc {
    const uint8_t *l0;   // level-0 frames (caller's buffer)
    int64_t l0_fstride;  // bytes between frames
    int32_t l0_pitch;    // row pitch of level 0
    uint8_t *pyr;        // levels >= 1 (handle-owned)
    int64_t pyr_fstride;
};

__device__ __forceinline__ const uint8_t *level_ptr(const FrameSrc &fs, const OrbLevel &L, int level, int b,
                                                    int *pitch)
{
    if (level == 0) {
        *pitch = fs.l0_pitch;
        return fs.l0 + (int64_t)b * fs.l0_fstride;
    }
    *pitch = L.pitch;
    return fs.pyr + (int64_t)b * fs.pyr_fstride + L.off;
}

__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan over the block (blockDim.x multiple of 64, <= 1024); s_wave = 17 ints of LDS
__device__ __forceinline__ int block_excl_scan(int v, int *s_wave, int *total)
{
    const int incl = wave_incl_scan(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int t = s_wave[w];
        if (w < wid) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    // cv::borderInterpolate(BORDER_REFLECT_101); len >= 2 in every use here, |overshoot| <= 3
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

// XCD-aware placement for grids of (work item, frame): workgroups are dealt round-robin to the 8 XCDs in launch order
// and every XCD has its own 4 MB L2, while one frame's pyramids are ~2 MB.  With a multiple of 8 frames, all workgroups
// of frame f are steered to XCD f % 8, so rows shared by neighbouring waves / overlapping patches of a frame are
// fetched from HBM once instead of once per XCD.
__device__ __forceinline__ void xcd_frame_remap(int &bx, int &b)
{
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, j = lin >> 3;
        b = xcd + 8 * (j / (int)gridDim.x);
        bx = j % (int)gridDim.x;
    }
}

// ---------------------------------------------------------------------------------------------------
// K1  bilinear pyramid level:  dst(level) = resize(src(level-1))      
// cv::resize(INTER_LINEAR) in its 8-bit fixed-point form: horizontal sums H = S[sx]*a0 + S[sx+1]*a1 with 11-bit
// coefficients, vertical step ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.  The level's geometry travels as kernel
// arguments; the taps (source index + the two coefficients per destination column / row) come from host-built tables
// computed with the exact fp64 / fp32 operation sequence of cv::resize; all kernel math is int32.
// ---------------------------------------------------------------------------------------------------
struct PyrArgs {
    const uint8_t *src;  // level l-1, frame 0
    uint8_t *dst;        // level l, frame 0
    int64_t src_fstride, dst_fstride;
    int32_t sw, sh, spitch;
    int32_t dw, dh, dpitch;
    const OrbTab *xtab, *ytab;  // cv::resize taps of the two axes (host-built, 4-entry aligned, padded by 8 entries)
    int32_t rb, nrblk;          // destination rows per lane run, number of runs
};

// A lane owns 4 adjacent destination pixels and a run of a.rb destination rows and walks down the SOURCE rows
// r = sy(y0), sy(y0)+1, ... once: the four source pairs of its pixels lie in one unaligned 8-byte window per source row
// (level-to-level scale < 2, checked on the host; pair selection by per-lane v_perm selectors, one v_dot2_u32_u16 per
// tap pair), so the horizontal sums H_r of every source row are formed once and kept for one step; the destination row d is completed in the step whose row is sy(d) + 1 -- sy is strictly increasing (the
// level-to-level ratio is >= 1, checked on the host), so a step completes at most one destination row.  Rows are
// fetched PW_PF steps ahead (loads clamp to the last source row: the virtual row sh repeats row sh - 1, which is what
// cv::resize's clamped second tap reads).  The loop issues loads only: completed dwords are parked in LDS
// ([row of the run][thread], conflict-free) and stored in one burst after the walk, so that the wait for a prefetched row
// never includes a store acknowledgement (stores and loads share one in-order counter on this part).
// The level pitch is a multiple of 64: the last column group stores its whole dword.
//
// Waves come in two kinds (round 6).  A ROW-BLOCK wave holds 64 column groups of ONE run of rows: which source steps complete a
// destination row, the row's two vertical coefficients, the source row address and the loop bounds are then the same for all
// lanes -- they live in scalar registers (the vertical taps come through the scalar cache), the vertical arithmetic runs only in
// the steps that complete a row (about 5 of 6; a per-lane predicate had to run it in every step, and in the steps that pad a walk
// to a multiple of four), and the walk ends at its last step.  Per completed row of 4 pixels: 1.2 x 12 (horizontal sums) + 16
// VALU instructions against 1.5 x 40.  The column groups a run has beyond a multiple of 64 (fewer than 37; 37 and more take a
// partly idle row-block wave of their own) are packed into MIXED waves, lanes of several runs side by side, which keep the
// per-lane form with the taps in LDS.
#ifndef PW_PF
#define PW_PF 2
#endif
#define PW_ROWS 16  // destination rows per lane run (8, 24, 32 measured slower)
// row-block waves per run of rows
__host__ __device__ inline int pyr_nfull(int ncol4) { return (ncol4 + 27) >> 6; }
static int pyr_walk_waves(int dw, int nrblk)
{
    const int ncol4 = (dw + 3) >> 2, nfull = pyr_nfull(ncol4), tail = std::max(ncol4 - nfull * 64, 0);
    return nfull * nrblk + (tail * nrblk + 63) / 64;
}
typedef uint32_t pw_u2 __attribute__((ext_vector_type(2)));

// the walk of one lane: column group cg of run rblk.  UNI: rblk (and with it every row quantity) is wave-uniform.
template <bool UNI>
__device__ __forceinline__ void pyr_walk_run(const PyrArgs &a, const int b, const int rblk, const int cg, const uint2 *s_yt, uint32_t *s_out)
{
    const int H = a.dh;
    const int dx0 = cg * 4;
    const int y0 = rblk * a.rb, yend = min(y0 + a.rb, H);
    const uint8_t *src = a.src + (int64_t)b * a.src_fstride;
    uint8_t *dst = a.dst + (int64_t)b * a.dst_fstride;
    auto tap = [&](int d) -> uint2 {   // .x = b0 | b1 << 16, .y = sy (low half); UNI: d is uniform, the entry goes to scalar registers
        if (!UNI) return s_yt[d];
        const uint2 t = ((const uint2 *)a.ytab)[d];
        return make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)t.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)t.y));
    };

    const uint4 tx01 = *(const uint4 *)(a.xtab + dx0), tx23 = *(const uint4 *)(a.xtab + dx0 + 2);
    const uint32_t xc[4] = {tx01.x, tx01.z, tx23.x, tx23.z};
    const int xs[4] = {(int)(short)tx01.y, (int)(short)tx01.w, (int)(short)tx23.y, (int)(short)tx23.w};
    const int sx0 = min(xs[0], a.sw - 8);
    uint32_t sel[4];
    orb_u2 coef[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t o = (uint32_t)min(max(xs[j] - sx0, 0), 7);
        sel[j] = 0x0c000c00u | (min(o + 1u, 7u) << 16) | o;
        coef[j] = __builtin_bit_cast(orb_u2, xc[j]);
    }
    uint2 cur = tap(y0);
    const int r0 = (int)(short)cur.y;
    int nsteps = (int)(short)tap(yend - 1).y + 2 - r0;
    if (!UNI) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, o, 64));
        nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    }
    const uint32_t sp = (uint32_t)a.spitch;
    const int rlast = a.sh - 1;
    // (stride 0, no bounds: the offsets are the kernel's own; word 3 = the untyped 32-bit format of gfx94x / gfx950)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, -1, 0x00020000);
    // uniform base + 32-bit per-lane offset (global_load with an SGPR base; UNI: the row offset is part of the scalar base)
    auto fetch = [&](int s, uint2 &q) {
        if (UNI) {
            // buffer addressing: resource base (the frame's level) + scalar row offset + the lane's column: no vector address arithmetic
            const pw_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, sx0, (int)((uint32_t)min(r0 + s, rlast) * sp), 0);
            q = make_uint2(v.x, v.y);
        } else
            q = *(const uint2 *)(src + (__umul24((uint32_t)min(r0 + s, rlast), sp) + (uint32_t)sx0));
    };
    auto hsum = [&](const uint2 &q, uint32_t (&h)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, __builtin_amdgcn_perm(q.y, q.x, sel[j])), coef[j], 0u, false) >> 4;
            asm volatile("" : "+v"(h[j]));   // formed once: the compiler re-derived the shift from the raw sum wherever the value was used
        }
    };
    // ((b0 * H0) >> 16) + ((b1 * H1) >> 16) + 2: the "+ 2" rides in the second product (+ 2 << 16, no carry into it from
    // below), the two ">> 16" are the SDWA word selects of one add, whose result lands in the low / high half of a pair
    // register; ">> 2" is then one packed shift per pixel pair
    auto vertical = [&](const uint32_t bb, const uint32_t (&Hp)[4], const uint32_t (&Hs)[4]) -> uint32_t {
        const uint32_t b0 = bb & 0xFFFFu, b1 = bb >> 16;
        uint32_t pa[4], pb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pa[j] = __umul24(b0, Hp[j]);
            pb[j] = __umul24(b1, Hs[j]) + 0x20000u;
        }
        uint32_t t01, t23;
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t01) : "v"(pa[0]), "v"(pb[0]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t01) : "v"(pa[1]), "v"(pb[1]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t23) : "v"(pa[2]), "v"(pb[2]));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t23) : "v"(pa[3]), "v"(pb[3]));
        const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t01) >> (orb_u2)(2));
        const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t23) >> (orb_u2)(2));
        return __builtin_amdgcn_perm(q23, q01, 0x06040200u);
    };
    // UNI: lane i keeps the taps of row y0 + i of the run (rb <= PW_ROWS < 64; the table is padded), and the completed row's successor
    // is read from its lane: no memory access for the taps inside the walk.  (Requested before the first rows: loads return in order,
    // so the wait for source row 0 covers it and no wait for it is left inside the loop.)
    uint2 mytap = cur;
    if (UNI) {
        mytap = ((const uint2 *)a.ytab)[min(y0 + (int)(threadIdx.x & 63), H + 7)];
        asm volatile("" : "+v"(mytap.x), "+v"(mytap.y));
    }
    uint2 raw[4];
#pragma unroll
    for (int k = 0; k <= PW_PF; ++k) fetch(k, raw[k]);
    // horizontal sums of the previous and of the current source row, by the parity of the step (the unroll factor is even)
    uint32_t Hh[2][4];
    hsum(raw[0], Hh[0]);  // step 0: source row sy(y0), completes nothing
    int d = y0;
    uint32_t *park = s_out + threadIdx.x;
    // steps run in groups of four (static ring indices).  Mixed waves: no exit inside a group, surplus steps re-read the clamped
    // last row and complete nothing; row-block waves leave at their last step.
    for (int s0 = 1; s0 < nsteps; s0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = s0 + k;
            if (UNI && s >= nsteps) break;
            fetch(s + PW_PF, raw[(k + 1 + PW_PF) % 4]);
            uint32_t(&Hp)[4] = Hh[k & 1];
            uint32_t(&Hs)[4] = Hh[(k + 1) & 1];
            hsum(raw[(k + 1) % 4], Hs);
            const bool emit = d < yend && (int)(short)cur.y + 1 == r0 + s;
            if (UNI) {
                if (emit) {   // wave-uniform
                    *park = vertical(cur.x, Hp, Hs);
                    park += 256;
                    d += 1;
                    cur.x = (uint32_t)__builtin_amdgcn_readlane((int)mytap.x, d - y0);
                    cur.y = (uint32_t)__builtin_amdgcn_readlane((int)mytap.y, d - y0);
                }
            } else {
                const uint32_t v = vertical(cur.x, Hp, Hs);
                if (emit) {
                    *park = v;
                    park += 256;
                    d += 1;
                }
                cur = s_yt[d];
            }
        }
    }
    // burst store of the run (every lane reads back its own LDS column: no barrier)
    const int nrows = yend - y0;
    if (UNI) {
        for (int i0 = 0; i0 < nrows; i0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = s_out[min(i0 + i, a.rb - 1) * 256 + threadIdx.x];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i0 + i < nrows) __builtin_amdgcn_raw_buffer_store_b32(v[i], rdst, dx0, (int)((uint32_t)(y0 + i0 + i) * (uint32_t)a.dpitch), 0);
        }
    } else {
        uint32_t oofs = __umul24((uint32_t)y0, (uint32_t)a.dpitch) + (uint32_t)dx0;
        for (int i0 = 0; i0 < a.rb; i0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = s_out[min(i0 + i, a.rb - 1) * 256 + threadIdx.x];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i0 + i < nrows) *(uint32_t *)(dst + (oofs + (uint32_t)(i0 + i) * (uint32_t)a.dpitch)) = v[i];
        }
    }
}

__global__ __launch_bounds__(256) void k_pyr_walk(PyrArgs a)
{
    extern __shared__ uint2 s_dyn[];
    uint2 *s_yt = s_dyn;                                   // [dh + 8]: the vertical taps, for the mixed waves
    uint32_t *s_out = (uint32_t *)(s_dyn + (a.dh + 8));    // [rb][256]
    const int b = blockIdx.y;
    const int ncol4 = (a.dw + 3) >> 2;
    const int nfull = pyr_nfull(ncol4), tail = max(ncol4 - nfull * 64, 0);
    const int nuni = nfull * a.nrblk, nmix = tail * a.nrblk;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int gw = (int)blockIdx.x * 4 + wv;   // wave of the frame: row-block waves first (run-major), then the mixed waves
    if ((int)blockIdx.x * 4 + 3 >= nuni) {     // a workgroup with a mixed wave (workgroup-uniform): the taps into LDS
        for (int i = threadIdx.x; i < a.dh + 8; i += 256) s_yt[i] = ((const uint2 *)a.ytab)[i];
        __syncthreads();
    }
    if (gw < nuni) {
        const int rblk = gw / nfull;
        pyr_walk_run<true>(a, b, rblk, min((gw - rblk * nfull) * 64 + lane, ncol4 - 1), s_yt, s_out);
    } else if (gw < nuni + ((nmix + 63) >> 6)) {
        const int fl = min((gw - nuni) * 64 + lane, nmix - 1);   // surplus lanes repeat the last lane's work
        const int rblk = fl / tail;
        pyr_walk_run<false>(a, b, rblk, nfull * 64 + fl - rblk * tail, s_yt, s_out);
    }
}

#ifdef ORBFE_DEVELOPER   // measured slower than the default chain; compiled only into developer builds
// Two levels per launch.  The workgroup owns a tile of level B = l: p2_gx column groups x p2_gy runs of a.rb rows, produced
// from level A = l - 1 exactly as k_pyr_walk does (same row walk, same arithmetic); the finished tile stays in LDS
// ([row][p2_gx] dwords), is stored to B in a burst, and level C = l + 1 is then resized from the LDS tile: every C pixel whose
// top-left tap falls on the tile's own part (tiles overlap by one column group and one row, so its other three taps are in
// the tile too) -- level B is never re-read from HBM.  Same fixed-point formula for C: H = (S[sx] * c0 + S[sx+1] * c1) >> 4 on
// both source rows, ((b0 * H0) >> 16) + ((b1 * H1) >> 16) + 2) >> 2.
struct Pyr2Args {
    PyrArgs ab;                 // A -> B as in k_pyr_walk (rb = rows per run; nrblk unused)
    uint8_t *dstC;              // level C, frame 0
    int32_t cw, ch, cpitch;
    const OrbTab *xtabC, *ytabC;
    const int32_t *cxs, *cys;   // [tiles_x + 1], [tiles_y + 1]
    int32_t gx, gy, tiles_x, tiles_y;
};

__global__ __launch_bounds__(256) void k_pyr_walk2(Pyr2Args p)
{
    extern __shared__ uint2 s_dyn[];
    const PyrArgs &a = p.ab;
    const int GX = p.gx, trows = p.gy * a.rb;
    uint2 *s_yt = s_dyn;                                     // [trows + 8] vertical taps of the tile's B rows
    uint32_t *s_tile = (uint32_t *)(s_dyn + (trows + 8));    // [trows][GX] dwords = the B tile
    const int b = blockIdx.y;
    const int tx = (int)blockIdx.x % p.tiles_x, ty = (int)blockIdx.x / p.tiles_x;
    const int W = a.dw, H = a.dh;
    const int g0 = tx * (GX - 1), by0 = ty * (trows - 1);    // first column group / row of the tile
    for (int i = threadIdx.x; i < trows + 8; i += 256) s_yt[i] = ((const uint2 *)a.ytab)[min(by0 + i, H + 7)];
    const int ncol4 = (W + 3) >> 2;
    const int lane_ok = (int)threadIdx.x < GX * p.gy;
    const int tl = min((int)threadIdx.x, GX * p.gy - 1);     // surplus lanes repeat the last lane's work
    const int run = tl / GX, cg = tl - run * GX;
    const int dx0 = min(g0 + cg, ncol4 - 1) * 4;             // column groups past the level repeat its last group
    const int y0 = min(by0 + run * a.rb, H - 1), yend = min(by0 + (run + 1) * a.rb, H);
    const uint8_t *src = a.src + (int64_t)b * a.src_fstride;
    uint8_t *dst = a.dst + (int64_t)b * a.dst_fstride;

    const uint4 tx01 = *(const uint4 *)(a.xtab + dx0), tx23 = *(const uint4 *)(a.xtab + dx0 + 2);
    const uint32_t xc[4] = {tx01.x, tx01.z, tx23.x, tx23.z};
    const int xs[4] = {(int)(short)tx01.y, (int)(short)tx01.w, (int)(short)tx23.y, (int)(short)tx23.w};
    const int sx0 = min(xs[0], a.sw - 8);
    uint32_t sel[4];
    orb_u2 coef[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t o = (uint32_t)min(max(xs[j] - sx0, 0), 7);
        sel[j] = 0x0c000c00u | (min(o + 1u, 7u) << 16) | o;
        coef[j] = __builtin_bit_cast(orb_u2, xc[j]);
    }
    __syncthreads();
    const int yl0 = y0 - by0;                                // tile-local row of the lane's first destination row
    uint2 cur = s_yt[yl0];
    const int r0 = (int)(short)cur.y;
    int nsteps = (int)(short)s_yt[max(yend - 1, y0) - by0].y + 2 - r0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, o, 64));
    nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    const uint32_t sp = (uint32_t)a.spitch;
    const int rlast = a.sh - 1;
    auto fetch = [&](int s, uint2 &q) { q = *(const uint2 *)(src + (__umul24((uint32_t)min(r0 + s, rlast), sp) + (uint32_t)sx0)); };
    auto hsum = [&](const uint2 &q, uint32_t (&h)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            h[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, __builtin_amdgcn_perm(q.y, q.x, sel[j])), coef[j], 0u, false) >> 4;
    };
    uint2 raw[4];
    fetch(0, raw[0]);
    fetch(1, raw[1]);
    fetch(2, raw[2]);
    uint32_t Hp[4];
    hsum(raw[0], Hp);
    int d = y0;
    uint32_t *park = s_tile + yl0 * GX + cg;
    for (int s0 = 1; s0 < nsteps; s0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = s0 + k;
            fetch(s + PW_PF, raw[(k + 1 + PW_PF) % 4]);
            uint32_t Hs[4];
            hsum(raw[(k + 1) % 4], Hs);
            const bool emit = lane_ok && d < yend && (int)(short)cur.y + 1 == r0 + s;
            const uint32_t b0 = cur.x & 0xFFFFu, b1 = cur.x >> 16;
            uint32_t pa[4], pb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pa[j] = __umul24(b0, Hp[j]);
                pb[j] = __umul24(b1, Hs[j]) + 0x20000u;
            }
            uint32_t t01, t23;
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t01) : "v"(pa[0]), "v"(pb[0]));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t01) : "v"(pa[1]), "v"(pb[1]));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t23) : "v"(pa[2]), "v"(pb[2]));
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t23) : "v"(pa[3]), "v"(pb[3]));
            const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t01) >> (orb_u2)(2));
            const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t23) >> (orb_u2)(2));
            if (emit) {
                *park = __builtin_amdgcn_perm(q23, q01, 0x06040200u);
                park += GX;
                d += 1;
            }
            cur = s_yt[min(d, yend) - by0];
#pragma unroll
            for (int j = 0; j < 4; ++j) Hp[j] = Hs[j];
        }
    }
    // burst store of the lane's column of the B tile (its own LDS words: no barrier needed)
    if (lane_ok && g0 + cg < ncol4) {
        uint32_t oofs = __umul24((uint32_t)y0, (uint32_t)a.dpitch) + (uint32_t)dx0;
        const int nrows = yend - y0;
        for (int i0 = 0; i0 < a.rb; i0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = s_tile[(yl0 + min(i0 + i, a.rb - 1)) * GX + cg];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i0 + i < nrows) *(uint32_t *)(dst + (oofs + (uint32_t)(i0 + i) * (uint32_t)a.dpitch)) = v[i];
        }
    }
    __syncthreads();
    // ---- level C from the tile ----
    // A thread owns one aligned 4-pixel column group of the tile's C part (its four horizontal taps live in registers) and a
    // contiguous share of the tile's C rows; per row: the vertical tap from LDS, 16 byte reads of the tile, one dword store.
    const int cx0 = p.cxs[tx], cx1 = p.cxs[tx + 1], cy0 = p.cys[ty], cy1 = p.cys[ty + 1];
    if (cx1 <= cx0 || cy1 <= cy0) return;
    const int gq0 = cx0 >> 2, ngc = ((cx1 - 1) >> 2) - gq0 + 1;
    const int nch = max(1, 256 / ngc);                       // row shares
    const int gi = (int)threadIdx.x % ngc, chn = (int)threadIdx.x / ngc;
    if (chn >= nch) return;
    const int nrowsC = cy1 - cy0, per = (nrowsC + nch - 1) / nch;
    const int ya = cy0 + chn * per, yb = min(ya + per, cy1);
    const uint8_t *T = (const uint8_t *)s_tile;
    const int tpitch = GX * 4, bx0 = g0 * 4;
    const int colmax = min(GX * 4, W - bx0) - 1, rowmax = min(trows, H - by0) - 1;   // clamps of the second taps
    uint8_t *dstC = p.dstC + (int64_t)b * a.dst_fstride;
    const int xg = (gq0 + gi) * 4;
    int c0i[4], c1i[4];
    uint32_t cc0[4], cc1[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x2 = xg + j;
        ok[j] = x2 >= cx0 && x2 < cx1;
        const OrbTab tt = p.xtabC[min(x2, p.cw - 1)];
        c0i[j] = min(max((int)tt.s - bx0, 0), colmax);
        c1i[j] = min(c0i[j] + 1, colmax);
        cc0[j] = (uint32_t)(uint16_t)tt.c0;
        cc1[j] = (uint32_t)(uint16_t)tt.c1;
    }
    const bool full = ok[0] && ok[1] && ok[2] && ok[3];
    OrbTab tyn = p.ytabC[min(ya, p.ch - 1)];
    for (int y2 = ya; y2 < yb; ++y2) {
        const OrbTab ty_ = tyn;
        tyn = p.ytabC[min(y2 + 1, p.ch - 1)];   // next row's vertical tap is in flight while this row is computed
        const int rr0 = (int)ty_.s - by0, rr1 = min(rr0 + 1, rowmax);
        const uint32_t b0 = (uint32_t)(uint16_t)ty_.c0, b1 = (uint32_t)(uint16_t)ty_.c1;
        const uint8_t *R0 = T + rr0 * tpitch, *R1 = T + rr1 * tpitch;
        uint32_t px[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t H0 = (R0[c0i[j]] * cc0[j] + R0[c1i[j]] * cc1[j]) >> 4, H1 = (R1[c0i[j]] * cc0[j] + R1[c1i[j]] * cc1[j]) >> 4;
            px[j] = ((((b0 * H0) >> 16) + ((b1 * H1) >> 16) + 2u) >> 2) & 0xFFu;
        }
        uint8_t *o = dstC + (__umul24((uint32_t)y2, (uint32_t)p.cpitch) + (uint32_t)xg);
        if (full) {
            *(uint32_t *)o = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ok[j]) o[j] = (uint8_t)px[j];
        }
    }
}
#endif  // ORBFE_DEVELOPER

// ---------------------------------------------------------------------------------------------------
// K2  FAST-9/16 with the reference's per-cell semantics: corner at t <=> A > t for the arc strength
//   A = max(A_dark, A_bright), cv score = A - 1, 3x3 strict NMS inside each cell's detectable interior, iniTh list
//   or -- for a cell where that is empty -- minTh list.
// ---------------------------------------------------------------------------------------------------
// The lane mask of a predicate straight from its compare (HIP's __ballot materialises the bool in a VGPR and compares it
// again: two VALU instructions per ballot in the row loops)
__device__ __forceinline__ unsigned long long orb_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// Dense streaming formulation.
//
// k_fast_map   one wave per (level, 256-px strip piece, row block).  Every lane owns 4 adjacent pixels and walks down
//              the rows with the last 7 image rows in registers (each unpacked once into its nine u16 pixel pairs,
//              fast_unpack_row) -- no LDS, no byte loads.  With
//              raw pixel values c[0..15] on the circle,  min over an arc of (v - c) = v - max(c)  and
//              min over an arc of (c - v) = min(c) - v, so
//                  A = max( v - min_k max(c[k..k+8]),  max_k min(c[k..k+8]) - v )
//              needs no per-tap subtraction: two levels of min3/max3 give all 16 nine-arcs (2 ops per arc and
//              polarity).  A is clamped to E = (A > max(minTh,1)) ? A : 0; the 3x3 strict NMS runs in registers on
//              three rows of E (neighbour lanes via shuffles) with the reference's per-cv::FAST-call semantics:
//              neighbours outside the own cell's detectable interior count as 0 (cell seams are per-lane column
//              masks and wave-uniform row flags).  Because iniTh >= minTh, a corner with A > iniTh survives NMS at
//              iniTh iff it survives at minTh, so ONE survivor map  M = survivor ? A : 0  encodes both of the
//              reference's lists: {M > iniTh} and, for cells where that is empty, {M > 0}  (:818-825).
//              Survivors are appended UNORDERED to the level's list {key, ord}; `ord` is computable from the pixel
//              position alone and is a rank key of the reference's candidate order (cell-row-major, raster inside
//              a cell), which is all DistributeOctTree's tie-break needs.  The fallback rule (a cell whose iniTh
//              list is empty contributes its minTh list) is applied per cell in k_octree's prologue.

// ---- packed 16-bit helpers: two pixels per VALU instruction --------------------------------------------
// A pixel pair is held as two u16 halves (values 0..255).  Read as f16 bit patterns those are positive
// denormals, whose order equals the integer order, so gfx950's 3-input packed min/max
// (v_pk_minimum3_f16 / v_pk_maximum3_f16) give exact integer results at two pixels per instruction.

__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c)
{
    const orb_h2 x = __builtin_bit_cast(orb_h2, a), y = __builtin_bit_cast(orb_h2, b), z = __builtin_bit_cast(orb_h2, c);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_minimum(__builtin_elementwise_minimum(x, y), z));  // v_pk_minimum3_f16
}
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c)
{
    const orb_h2 x = __builtin_bit_cast(orb_h2, a), y = __builtin_bit_cast(orb_h2, b), z = __builtin_bit_cast(orb_h2, c);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));  // v_pk_maximum3_f16
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_s2, a) - __builtin_bit_cast(orb_s2, b));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(orb_s2, a), __builtin_bit_cast(orb_s2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, a) - __builtin_bit_cast(orb_u2, b));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(orb_u2, a), __builtin_bit_cast(orb_u2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(orb_u2, a), __builtin_bit_cast(orb_u2, b)));
}
__device__ __forceinline__ uint32_t pk_subsat_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(orb_u2, a), __builtin_bit_cast(orb_u2, b)));
}

// pixels (IDX, IDX+1) of a 12-byte row window as a u16 pair
template <int IDX>
__device__ __forceinline__ uint32_t rowpair(const uint32_t (&w)[3])
{
    static_assert(IDX >= 0 && IDX < 11, "window pair");
    if ((IDX & 3) < 3)
        return __builtin_amdgcn_perm(0u, w[IDX >> 2], 0x0c000c00u | ((uint32_t)((IDX & 3) + 1) << 16) | (uint32_t)(IDX & 3));
    return __builtin_amdgcn_perm(w[(IDX >> 2) + 1 > 2 ? 2 : (IDX >> 2) + 1], w[IDX >> 2], 0x0c040c03u);
}

// Every row of the ring is unpacked ONCE, when it arrives, into the nine pixel pairs (i, i+1), i = 1..9, of its 12-byte
// window (E[i-1]); each pair is then used by up to three centre rows and by both pixel pairs of the lane, instead of
// being re-extracted with a v_perm at every use (34 -> 9 extractions per row step).
#define FM_NE 9
__device__ __forceinline__ void fast_unpack_row(const uint32_t (&w)[3], uint32_t (&e)[FM_NE])
{
    e[0] = rowpair<1>(w);
    e[1] = rowpair<2>(w);
    e[2] = rowpair<3>(w);
    e[3] = rowpair<4>(w);
    e[4] = rowpair<5>(w);
    e[5] = rowpair<6>(w);
    e[6] = rowpair<7>(w);
    e[7] = rowpair<8>(w);
    e[8] = rowpair<9>(w);
}

// thresholded arc strengths S = max(A - t, 0) of the pixel pair (J, J+1) of the lane, as two u16 halves;
// rows are unpacked windows (fast_unpack_row) of image rows y-3 .. y+3
template <int J>
__device__ __forceinline__ uint32_t fast_strength_pair(const uint32_t (&rm3)[FM_NE], const uint32_t (&rm2)[FM_NE],
                                                       const uint32_t (&rm1)[FM_NE], const uint32_t (&r0)[FM_NE],
                                                       const uint32_t (&rp1)[FM_NE], const uint32_t (&rp2)[FM_NE],
                                                       const uint32_t (&rp3)[FM_NE], uint32_t t)
{
    uint32_t c[16];
    c[0] = rp3[3 + J];
    c[1] = rp3[4 + J];
    c[2] = rp2[5 + J];
    c[3] = rp1[6 + J];
    c[4] = r0[6 + J];
    c[5] = rm1[6 + J];
    c[6] = rm2[5 + J];
    c[7] = rm3[4 + J];
    c[8] = rm3[3 + J];
    c[9] = rm3[2 + J];
    c[10] = rm2[1 + J];
    c[11] = rm1[0 + J];
    c[12] = r0[0 + J];
    c[13] = rp1[0 + J];
    c[14] = rp2[1 + J];
    c[15] = rp3[2 + J];
    const uint32_t v = r0[3 + J];
    // Nine-arcs k and k+1 (k even) share the eight pixels c[k+1..k+8], so
    //   max(min arc_k, min arc_k+1) = min(c[k+1..k+8], max(c[k], c[k+9]))     (and dually for the dark polarity):
    // odd-aligned pairs P/Q, three 3-way ops per two arc pairs -- 32 ops per polarity for all 16 arcs.
    uint32_t P[8], Q[8], ex[8], en[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        P[i] = pk_min_u16(c[2 * i + 1], c[(2 * i + 2) & 15]);
        Q[i] = pk_max_u16(c[2 * i + 1], c[(2 * i + 2) & 15]);
        ex[i] = pk_max_u16(c[2 * i], c[(2 * i + 9) & 15]);
        en[i] = pk_min_u16(c[2 * i], c[(2 * i + 9) & 15]);
    }
    // The eight-pixel windows P[i..i+3] of two neighbouring arc pairs share three of their four pairs: one 3-way op
    // (P[i], P[i+1], P[i+2]), i even, serves window i (with P[i+3]) and window i-1 (with P[i-1]) -- 12 ops per polarity for the
    // eight windows instead of 16, the closing op still takes the arc pair's end pixels along.
    uint32_t Wb[8], Wd[8];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const uint32_t ab = pk_min3(P[i], P[(i + 1) & 7], P[(i + 2) & 7]);
        const uint32_t ad = pk_max3(Q[i], Q[(i + 1) & 7], Q[(i + 2) & 7]);
        Wb[i] = pk_min3(ab, P[(i + 3) & 7], ex[i]);
        Wd[i] = pk_max3(ad, Q[(i + 3) & 7], en[i]);
        Wb[(i + 7) & 7] = pk_min3(P[(i + 7) & 7], ab, ex[(i + 7) & 7]);
        Wd[(i + 7) & 7] = pk_max3(Q[(i + 7) & 7], ad, en[(i + 7) & 7]);
    }
    const uint32_t maxmin = pk_max_u16(pk_max3(pk_max3(pk_max3(Wb[0], Wb[1], Wb[2]), Wb[3], Wb[4]), Wb[5], Wb[6]), Wb[7]);
    const uint32_t minmax = pk_min_u16(pk_min3(pk_min3(pk_min3(Wd[0], Wd[1], Wd[2]), Wd[3], Wd[4]), Wd[5], Wd[6]), Wd[7]);
    // S = max(A, t) - t with A = max(v - min_arcs(max), max_arcs(min) - v, 0): zero for non-corners (A <= t), order
    // preserving for corners; t = 0x3FF in a half switches the pixel off (A <= 255)
    const uint32_t m = pk_max3(pk_subsat_u16(v, minmax), pk_subsat_u16(maxmin, v), t);
    return pk_sub_u16(m, t);
}

// Exact NECESSARY condition for A > t on the pixel pair (J, J+1): every nine-arc of the 16-pixel circle contains at
// least one pixel of each opposite pair, so a bright corner needs max(c0, c8) > v + t AND max(c4, c12) > v + t (and
// dually for dark).  Non-zero half <=> that pixel may be a corner.  15 packed ops against 72 for the strength.
template <int J>
__device__ __forceinline__ uint32_t fast_compass_pair(const uint32_t (&rm3)[FM_NE], const uint32_t (&r0)[FM_NE],
                                                      const uint32_t (&rp3)[FM_NE], uint32_t t)
{
    const uint32_t c0 = rp3[3 + J], c8 = rm3[3 + J];
    const uint32_t c4 = r0[6 + J], c12 = r0[0 + J];
    const uint32_t v = r0[3 + J];
    const uint32_t mb = pk_min_u16(pk_max_u16(c0, c8), pk_max_u16(c4, c12));
    const uint32_t md = pk_max_u16(pk_min_u16(c0, c8), pk_min_u16(c4, c12));
    return pk_subsat_u16(pk_max_u16(pk_subsat_u16(mb, v), pk_subsat_u16(v, md)), t);
}

// the network of fast_strength_pair on values -- the same operations in the same order (the lane-compacting form,
// orbfe_fast_body_c.inc, gathers the 16 circle pixel pairs + centre of an item from its LDS pixel ring)
__device__ __forceinline__ uint32_t fast_strength_from(const uint32_t (&c)[16], uint32_t v, uint32_t t)
{
    uint32_t P[8], Q[8], ex[8], en[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        P[i] = pk_min_u16(c[2 * i + 1], c[(2 * i + 2) & 15]);
        Q[i] = pk_max_u16(c[2 * i + 1], c[(2 * i + 2) & 15]);
        ex[i] = pk_max_u16(c[2 * i], c[(2 * i + 9) & 15]);
        en[i] = pk_min_u16(c[2 * i], c[(2 * i + 9) & 15]);
    }
    uint32_t Wb[8], Wd[8];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const uint32_t ab = pk_min3(P[i], P[(i + 1) & 7], P[(i + 2) & 7]);
        const uint32_t ad = pk_max3(Q[i], Q[(i + 1) & 7], Q[(i + 2) & 7]);
        Wb[i] = pk_min3(ab, P[(i + 3) & 7], ex[i]);
        Wd[i] = pk_max3(ad, Q[(i + 3) & 7], en[i]);
        Wb[(i + 7) & 7] = pk_min3(P[(i + 7) & 7], ab, ex[(i + 7) & 7]);
        Wd[(i + 7) & 7] = pk_max3(Q[(i + 7) & 7], ad, en[(i + 7) & 7]);
    }
    const uint32_t maxmin = pk_max_u16(pk_max3(pk_max3(pk_max3(Wb[0], Wb[1], Wb[2]), Wb[3], Wb[4]), Wb[5], Wb[6]), Wb[7]);
    const uint32_t minmax = pk_min_u16(pk_min3(pk_min3(pk_min3(Wd[0], Wd[1], Wd[2]), Wd[3], Wd[4]), Wd[5], Wd[6]), Wd[7]);
    const uint32_t m = pk_max3(pk_subsat_u16(v, minmax), pk_subsat_u16(maxmin, v), t);
    return pk_sub_u16(m, t);
}

__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

#define FM_BUF 384       // per-wave LDS staging of survivors before one atomic reserves their run in the level's list
#define FM_ROW_MAX 140   // a row adds at most 2 per lane + 1 per lane that straddles a cell seam (<= 9 of them)

// The staged survivors go to the level's list, and every survivor above iniTh marks its FAST cell in the WAVE's cell bitmap
// in LDS (the per-cell threshold fallback :818-825 needs "does this cell have an iniTh corner" before any key can be
// judged, so the quadtree kernel would otherwise spend a whole pass over the keys on it).  The wave's bitmap goes to the
// level's bitmap in global memory once, when the wave ends: a wave covers a 256-px x 40-row strip, i.e. a handful of cells.
__device__ __forceinline__ void fm_flush(uint2 *slist, int32_t *scnt, const uint2 *sbuf, int nbuf, int key_cap, int lane,
                                         uint32_t *lflag, int ini)
{
    int base = 0;
    if (lane == 0) base = atomicAdd(scnt, nbuf);
    base = __shfl(base, 0, 64);
    for (int i = lane; i < nbuf; i += 64) {
        const uint2 e = sbuf[i];
        if (base + i < key_cap) slist[base + i] = e;
        if (orb_key_r(e.x) >= ini) {  // cv score = A - 1 >= iniTh  <=>  A > iniTh
            const uint32_t cell = e.y >> 12;
            atomicOr(&lflag[cell >> 5], 1u << (cell & 31));  // LDS, result unused: ds_or_b32
        }
    }
}

// SPARSE = 1: wave-uniform shortcuts for frames whose corners are sparse (real camera images): a row step whose 256
// pixels all fail the compass test skips the arc evaluation, and an NMS row with no strength in its 3-row
// neighbourhood skips the NMS / emission block.  Results are identical; on corner-saturated frames the tests only cost.
// -DFM_WAVES_PER_EU=n caps the kernel's occupancy (A/B: leaving register file to a memory-bound kernel on a side stream)
#if defined(FM_WAVES_PER_EU)
#define FM_OCC __attribute__((amdgpu_waves_per_eu(FM_WAVES_PER_EU, FM_WAVES_PER_EU)))
#else
#define FM_OCC
#endif
#define FM_SHARED_DECLS                                                                                             \
    __shared__ uint2 s_buf[4][FM_BUF];                                                                              \
    extern __shared__ uint32_t s_cf[]; /* [4][cf_words]: per-wave bitmap of the level's cells with a survivor above iniTh */
#ifdef FM_LDS_CONSTS
// Per-lane constants of the row loop (the six half-word masks of the cell seams and the four `ord` column parts) parked in
// LDS and re-read where they are used: ten registers less in the kernel's peak live set.  At <= 152 registers three of its
// waves leave 56 per SIMD lane -- room for one wave of an HBM-bound kernel (k_pyr_walk: 48) beside them, where 160 leave 32
// and nothing fits (A/B in profiles/r04_ab_experiments.json).
#define FM_LDS_DECLS                                                             \
    __shared__ uint4 s_lcm[4][64]; /* in01, in23, lv01, lv23 */                  \
    __shared__ uint2 s_lcr[4][64]; /* rv01, rv23 */                              \
    __shared__ uint4 s_lco[4][64]; /* ordx[0..3] */
#else
#define FM_LDS_DECLS
#endif

template <int SPARSE>
__global__ __launch_bounds__(256) FM_OCC void k_fast_map(const OrbPlan *__restrict__ plan, FrameSrc fs,
                                                  const OrbLane *__restrict__ lanes, int nwaves,
                                                  uint2 *__restrict__ skeys,      // [B][keys_per_frame] {key, ord}
                                                  int32_t *__restrict__ scount,   // [B][nlevels] * NK_STRIDE, zeroed
                                                  uint32_t *__restrict__ cflags,  // [B][nlevels][cf_words], zeroed
                                                  int32_t cf_words,
                                                  unsigned long long *__restrict__ fstat)  // {row steps, arc skips, nms skips} or null
{
    FM_SHARED_DECLS
    FM_LDS_DECLS
    int b = blockIdx.y, bx = blockIdx.x;
    xcd_frame_remap(bx, b);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int t = bx * (blockDim.x >> 6) + wv;
    if (t >= nwaves) return;
#include "orbfe_fast_body.inc"
}

// The same pass over a lane list made of whole CELL ROWS, one run of rows per wave (orbfe_fast_body_u.inc): what batch handles run.
// k_fast_map above takes any run of rows per lane (handles made for a few frames per call walk short runs: a small call is bound by
// the length of one wave's walk).
template <int SPARSE>
__global__ __launch_bounds__(256) FM_OCC void k_fast_map_u(const OrbPlan *__restrict__ plan, FrameSrc fs,
                                                    const OrbLane *__restrict__ lanes, int nwaves,
                                                    uint2 *__restrict__ skeys, int32_t *__restrict__ scount,
                                                    uint32_t *__restrict__ cflags, int32_t cf_words,
                                                    unsigned long long *__restrict__ fstat)
{
    FM_SHARED_DECLS
    int b = blockIdx.y, bx = blockIdx.x;
    xcd_frame_remap(bx, b);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int t = bx * (blockDim.x >> 6) + wv;
    if (t >= nwaves) return;
#include "orbfe_fast_body_u.inc"
}

// The lane-compacting form of the same pass (orbfe_fast_body_c.inc): per wave a ring of the pixel rows its lanes fetched (one dword
// per lane and row, 2 x FC_PN slots of FC_PW dwords), a queue of FC_QCAP one-dword tags of parked pixel pairs, four strength rows
// in flight (a byte per pixel) and the survivor staging buffer -- 7.4 KB per wave and 96 registers: five waves per SIMD.
#define FC_QCAP 128      // tags (power of two); a push of <= 64 always finds room once the fill is <= FC_QCAP - 64
#ifndef FC_LAG
#define FC_LAG 3         // rows the suppression runs behind the front: an item never waits longer
#endif
#define FC_PN (6 + FC_LAG)   // rows the pixel ring holds: what step s parked is evaluated before the row of step s + FC_LAG is written
#define FC_PW 66         // dwords per ring slot: the 64 lanes + a pad on both sides
#define FC_SEAMS 16      // lanes of a wave whose pixel pair straddles a cell seam (cells are >= 30 px wide: <= 9)
#define FC_BUF (128 + FC_SEAMS)   // survivor staging: one row's worth; flushed when the next row might not fit
#ifndef FC_OCC
#define FC_OCC
#endif
static_assert(FC_LAG >= 1 && FC_LAG <= 3, "the strength-row ring (srow / snap, 4 slots) is re-used by the front half of step s + 4: the back half of step s must have read it, i.e. lag <= 3");
static_assert(FC_PN >= 6 + FC_LAG && FC_PN >= 7, "pixel ring: the first row of an item parked at step s is overwritten at step s - 6 + FC_PN");
// fast_compass_pair on values: the four compass pixel pairs and the centre pair of one pixel pair
__device__ __forceinline__ uint32_t fast_compass_from(uint32_t c0, uint32_t c8, uint32_t c4, uint32_t c12, uint32_t v, uint32_t t)
{
    const uint32_t mb = pk_min_u16(pk_max_u16(c0, c8), pk_max_u16(c4, c12));
    const uint32_t md = pk_max_u16(pk_min_u16(c0, c8), pk_min_u16(c4, c12));
    return pk_subsat_u16(pk_max_u16(pk_subsat_u16(mb, v), pk_subsat_u16(v, md)), t);
}
__global__ __launch_bounds__(256) FC_OCC void k_fast_map_c(const OrbPlan *__restrict__ plan, FrameSrc fs,
                                                    const OrbLane *__restrict__ lanes, int nwaves,
                                                    uint2 *__restrict__ skeys, int32_t *__restrict__ scount,
                                                    uint32_t *__restrict__ cflags, int32_t cf_words,
                                                    unsigned long long *__restrict__ fstat)  // {row steps, batches, parked pairs} of sampled waves, or null
{
    __shared__ uint32_t s_pix[4][2 * FC_PN * FC_PW];
#ifdef FC_EXTRA_LDS   // occupancy probe: dead LDS that costs a workgroup slot per CU
    __shared__ uint32_t s_fcpad[FC_EXTRA_LDS / 4];
    if (nwaves < 0) s_fcpad[threadIdx.x] = 1u;
    if (nwaves < -1) scount[0] = (int32_t)s_fcpad[threadIdx.x ^ 1];
#endif
    __shared__ uint32_t s_q[4][FC_QCAP];
    __shared__ uint32_t s_srow[4][4 * 64];
    __shared__ uint2 s_buf[4][FC_BUF];
    extern __shared__ uint32_t s_cf[];
    int b = blockIdx.y, bx = blockIdx.x;
    xcd_frame_remap(bx, b);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int t = bx * (blockDim.x >> 6) + wv;
    if (t >= nwaves) return;
#include "orbfe_fast_body_c.inc"
}

#ifdef ORBFE_DEVELOPER   // measured slower than the default chain; compiled only into developer builds
// FAST on level l and cv::resize l -> l + 1 in ONE launch, the launches chained over the levels ("pyramid inside
// the FAST pass").  FAST's live set leaves no register for a second job in its lanes (157 of the 168 that three waves per SIMD
// allow), so the fusion is by WORKGROUP ROLE: `npyr` of the launch's workgroups per frame are resize workgroups
// (k_pyr_walk's walk, unchanged), the others FAST workgroups over the wave range [wave_lo, wave_lo + nwaves) of the lane list.
// Both read level l, FAST is VALU-bound and the walk HBM-bound, and the two kinds are dealt out proportionally over the block
// index (spread = 1), so a CU holds both at any time and the second reader of a row finds it in L2; spread = 0 puts the resize
// workgroups first.  npyr = 0 on the last level.
template <int SPARSE>
__global__ __launch_bounds__(256) void k_fast_pyr(const OrbPlan *__restrict__ plan, FrameSrc fs, const OrbLane *__restrict__ lanes,
                                                  int wave_lo, int nwaves, uint2 *__restrict__ skeys, int32_t *__restrict__ scount,
                                                  uint32_t *__restrict__ cflags, int32_t cf_words,
                                                  unsigned long long *__restrict__ fstat, PyrArgs pa, int npyr, int spread)
{
    FM_SHARED_DECLS
    FM_LDS_DECLS
    int b = blockIdx.y, bx = blockIdx.x;
    xcd_frame_remap(bx, b);
    int before = min(bx, npyr);              // resize workgroups in front of this one
    bool is_pyr = bx < npyr;
    if (spread) {
        const int T = (int)gridDim.x;
        before = (int)(((uint32_t)bx * (uint32_t)npyr) / (uint32_t)T);
        is_pyr = (int)(((uint32_t)(bx + 1) * (uint32_t)npyr) / (uint32_t)T) > before;
    }
    if (is_pyr) {
        const PyrArgs &a = pa;
        const uint32_t bxi = (uint32_t)before;
#include "orbfe_pyr_body.inc"
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    if ((bx - before) * 4 + wv >= nwaves) return;
    const int t = wave_lo + (bx - before) * 4 + wv;
#include "orbfe_fast_body.inc"
}
#endif  // ORBFE_DEVELOPER

// ---------------------------------------------------------------------------------------------------
// K3  DistributeOctTree on the device.  One workgroup per (frame, level).
//
// The reference keeps a std::list of nodes: a pass visits nodes in some processing order, replaces each
// visited node by its non-empty children (push_front in the order n1,n2,n3,n4) and may stop early once
// the list holds >= N nodes.  Both of its loops are instances of one generic pass:
//     breadth-first loop (:608-667)  processing order = list order,           never stops early
//     largest-first loop (:678-739)  processing order = (size desc, creation desc), stops at >= N
// and the list after a pass that processed ranks 0..R-1 is
//     [children(P[R-1]) n4..n1] ... [children(P[0]) n4..n1]  ++  [unprocessed nodes in old order].
// Positions therefore follow from prefix sums over the processing order and the final "first strongest key" from an
// LDS atomicMax on (response | ~candidate_order | key index).  Keys never move.
// Equal-size ties in the largest-first order are broken by creation order (the reference compares heap
// addresses there, :686 -- see DESIGN.md "quadtree contract").
//
// What a pass needs from the keys is only the number of keys in each child quadrant.  The child boxes are a pure
// function of the root box (DivideNode halves with ceil, :480-481), so every key's quadrant path is known up front:
// the prologue computes a 5-level path code per key and a histogram over the 4^5 leaves of every root; quadrant
// counts of any node down to depth 4 are sums of that histogram.  The first 5 passes -- normally all of them --
// therefore run as node-level bookkeeping only, by ONE wave, without touching the keys and without barriers.
// Only trees that must go deeper fall back to streaming key passes (keys then carry their node index).
// Keys arrive in arbitrary order from k_fast_map; each carries `ord`, its rank in the reference's candidate order.
// ---------------------------------------------------------------------------------------------------
#define QT_MAX 512
// One launch covers a run of consecutive levels whose node lists, root counts and path tables share one LDS carve-up:
// the upper pyramid levels ask for a fraction of level 0's features, so their workgroups are given a smaller carve-up
// and fewer threads and more of them fit a CU (the node bookkeeping is one wave's serial work per workgroup).
struct OctGroup {
    int32_t level0;        // first level of the group; gridDim.x = number of levels in it
    int32_t M;             // node capacity (multiple of 64)
    int32_t nini;          // most roots of a level in the group
    int32_t tw, th;        // largest level size in the group (path tables)
    int32_t ncells;        // most FAST cells of a level in the group
};
#define KNODE_MASK 0x3FFFu
#ifndef KUNROLL
#define KUNROLL 8
#endif
#define FFD 5               // histogram depth
#define FF_PER_ROOT 1364    // 4 + 16 + 64 + 256 + 1024
__host__ __device__ inline int ff_off(int d) { return ((1 << (2 * d)) - 4) / 3; }  // first entry of depth d (1..5)

struct QtShared {
    // two generations of the node list (current / next), addressed arithmetically -- an array of pointers indexed by
    // a run-time generation would live in scratch memory
    int M;
    int32_t *cnt0;       // [2][M]
    uint32_t *path0;     // [2][M] prefix | depth << 12 | root << 16
    __device__ __forceinline__ int32_t *cnt(int g) const { return cnt0 + g * M; }
    __device__ __forceinline__ uint32_t *path(int g) const { return path0 + g * M; }
    // [M*4] quadrant counts of the current pass; the node phase turns the entry of (node, quadrant) into the position of
    // that child in the next list (-1 if empty): the same words, read as `cc` before and as `childpos` after
    int32_t *cc;
    int32_t *P;          // processing order -> node index
    int32_t *rankOf0;    // [2][M] node index -> processing rank or -1 (per list generation)
    __device__ __forceinline__ int32_t *rankOf(int g) const { return rankOf0 + g * M; }
    int32_t *acc;        // inclusive sums over ranks
    int32_t *newIdx;     // next-list position of an unprocessed node, -1 for a processed one
    unsigned long long *skey;
    int32_t *hist;       // [nroots * FF_PER_ROOT] quadrant-path histogram, later the path -> node table
    uint16_t *xtab;      // [w] window column -> root << 10 | x half of the 5-level path code (bits 8,6,4,2,0)
    uint16_t *ytab;      // [h] window row    -> y half of the path code (bits 9,7,5,3,1)
    uint32_t *cflag;     // [ncells / 32] bit = the FAST cell has a survivor above iniTh
    int32_t *misc;       // [64]: 0..15 state, 16..23 root slots, 32..49 scan scratch
    // node boxes exist only for trees that go deeper than the histogram (clustered candidates): two generations of
    // (ulx, uly, urx, bry) in GLOBAL scratch of this (frame, level) -- [2][4][M] int16
    int16_t *gbox;
    __device__ __forceinline__ int16_t *box(int g, int j) const { return gbox + (g * 4 + j) * M; }
};

__host__ __device__ inline int qt_pow2(int v)
{
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

// The node arrays (56 B per node + the sort buffer) normally live in LDS next to the histogram and the tables; a level that
// asks for more nodes than the CU's LDS holds (nfeatures beyond ~2400 per level) keeps them in a global scratch slice of its
// (frame, level) instead -- same code, the accesses become global loads / stores (template parameter of k_octree).
__device__ __forceinline__ void qt_carve(char *lds, char *nodes, int M, int nroots, int w, int h, int ncells, QtShared &q)
{
    char *p = nodes;
    q.skey = (unsigned long long *)p; p += (size_t)qt_pow2(M) * 8;  // the largest-first sort is bitonic: power of two
    q.cc = (int32_t *)p; p += (size_t)M * 16;
    q.M = M;
    q.cnt0 = (int32_t *)p; p += (size_t)M * 8;
    q.path0 = (uint32_t *)p; p += (size_t)M * 8;
    q.P = (int32_t *)p; p += (size_t)M * 4;
    q.rankOf0 = (int32_t *)p; p += (size_t)M * 8;
    q.acc = (int32_t *)p; p += (size_t)M * 4;
    q.newIdx = (int32_t *)p; p += (size_t)M * 4;
    p = nodes == lds ? p : lds;
    q.hist = (int32_t *)p; p += (size_t)nroots * FF_PER_ROOT * 4;
    q.misc = (int32_t *)p; p += 64 * 4;
    q.xtab = (uint16_t *)p; p += (size_t)((w + 1) & ~1) * 2;
    q.ytab = (uint16_t *)p; p += (size_t)((h + 1) & ~1) * 2;
    q.cflag = (uint32_t *)p;
}

size_t orbk_octree_node_bytes(int M) { return (((size_t)qt_pow2(M) * 8 + (size_t)M * (16 + 8 + 8 + 20)) + 255) & ~(size_t)255; }
static size_t octree_table_bytes(int nroots, int w, int h, int ncells)
{
    return (size_t)nroots * FF_PER_ROOT * 4 + 64 * 4 + (size_t)(((w + 1) & ~1) + ((h + 1) & ~1)) * 2 + (size_t)((ncells + 31) / 32) * 4;
}
size_t orbk_octree_lds_bytes(int M, int nroots, int w, int h, int ncells)
{
    return (size_t)qt_pow2(M) * 8 + (size_t)M * (16 + 8 + 8 + 20) + octree_table_bytes(nroots, w, h, ncells);
}

// bytes of global scratch one (frame, level) workgroup may need for the node boxes of a deep tree
size_t orbk_octree_box_bytes(int M) { return (size_t)M * 2 * 4 * sizeof(int16_t); }

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// inclusive scan of arr[0..n) in LDS, in place, by all threads of the workgroup; returns the total.  Ends with a barrier.
__device__ __forceinline__ int bscan_inclusive(int32_t *arr, int n, int32_t *sw)
{
    const int tid = threadIdx.x, QT = blockDim.x, wid = tid >> 6, lane = tid & 63, nw = QT >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += QT) {
        const int i = base + tid;
        const int v = i < n ? arr[i] : 0;
        const int incl = wave_incl_scan(v);
        if (lane == 63) sw[wid] = incl;
        __syncthreads();
        int pre = 0, tot = 0;
        for (int w = 0; w < nw; ++w) {
            const int t = sw[w];
            pre += w < wid ? t : 0;
            tot += t;
        }
        if (i < n) arr[i] = carry + pre + incl;
        carry += tot;
        __syncthreads();
    }
    return carry;
}

// One generic pass at node level, executed by ALL threads of the workgroup (barriers between its steps).
// In:  q.cc[i*4+qd] for every node i in P (quadrant sizes), list `cur` of size S, processing order P[0..m), rankOf.
// Out: list `cur^1` (sizes, paths; boxes when DEEP), the old->new map (newIdx / cc-as-childpos), next P / rankOf, S, m,
//      modeB, finish -- all workgroup-uniform.
// largest-first processing order (:686-687): q.skey[0..Mp) holds (size << 32 | (creation seq + 1) << 16 | list position) for
// the nToExpand multi-key children (zero beyond); sorts descending and writes P / rankOf of list generation nx
__device__ __forceinline__ void qt_sort_assign(const QtShared &q, int Mp, int nToExpand, int S2, int nx)
{
    const int tid = threadIdx.x, QT = blockDim.x;
    for (int kk = 2; kk <= Mp; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < Mp; i += QT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = q.skey[i], c2 = q.skey[ixj];
                    const bool desc = (i & kk) == 0;  // overall descending
                    if (desc ? (a < c2) : (a > c2)) { q.skey[i] = c2; q.skey[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < S2; i += QT) q.rankOf(nx)[i] = -1;
    __syncthreads();
    for (int r = tid; r < nToExpand; r += QT) {
        const int pos = (int)(q.skey[r] & 0xFFFFull);
        q.P[r] = pos;
        q.rankOf(nx)[pos] = r;
    }
}

// Two inclusive scans with shared barriers: a[0..na) and b[0..nb) in LDS, in place, by all threads.  Ends with a barrier.
__device__ __forceinline__ void bscan2_inclusive(int32_t *a, int na, int32_t *b, int nb, int32_t *sw, int &ta, int &tb)
{
    const int tid = threadIdx.x, QT = blockDim.x, wid = tid >> 6, lane = tid & 63, nw = QT >> 6;
    int ca = 0, cb = 0;
    for (int base = 0; base < max(na, nb); base += QT) {
        const int i = base + tid;
        const int va = i < na ? a[i] : 0, vb = i < nb ? b[i] : 0;
        const int ia = wave_incl_scan(va), ib = wave_incl_scan(vb);
        if (lane == 63) { sw[wid] = ia; sw[8 + wid] = ib; }
        __syncthreads();
        int pa = 0, pb = 0, sa = 0, sb = 0;
        for (int w = 0; w < nw; ++w) {
            const int x = sw[w], y = sw[8 + w];
            pa += w < wid ? x : 0;
            pb += w < wid ? y : 0;
            sa += x;
            sb += y;
        }
        if (i < na) a[i] = ca + pa + ia;
        if (i < nb) b[i] = cb + pb + ib;
        ca += sa;
        cb += sb;
        __syncthreads();
    }
    ta = ca;
    tb = cb;
}

// A breadth-first pass (:608-667) of a tree whose node sizes come from the leaf histogram: every multi-key node of the list
// is split, in list order.  Same result as qt_node_phase<false> with modeB == 0, in 4 barriers instead of 15: children
// and multi-key-children counts per rank are scanned together (packed), the unprocessed-node flags in the same barrier pair,
// and the next processing order (list order of the new multi-key nodes, or the sort keys when the largest-first mode
// begins) is written together with the next list -- a multi-key child of rank r sits at multi-rank
// totalMulti - inclMulti[r] + (its index among r's multi-key children in n4..n1 order), creation sequence
// inclMulti[r] - nMulti[r] + (index in n1..n4 order).
__device__ __forceinline__ void qt_pass_bfs_hist(const QtShared &q, int N, int &S, int &m, int &cur, int &modeB, bool &finish)
{
    const int tid = threadIdx.x, QT = blockDim.x;
    const int nx = cur ^ 1;
    int32_t *sw = q.misc + 32;
    int32_t *pk = q.acc, *un = q.newIdx;
    auto quad = [&](int i) {
        const uint32_t pth = q.path(cur)[i];
        const int d = (int)((pth >> 12) & 0xFu), root = (int)(pth >> 16);
        return &q.hist[root * FF_PER_ROOT + ff_off(d + 1) + (int)((pth & 0xFFFu) << 2)];
    };
    for (int r = tid; r < m; r += QT) {
        const int32_t *c = quad(q.P[r]);
        pk[r] = ((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) | (((c[0] > 1) + (c[1] > 1) + (c[2] > 1) + (c[3] > 1)) << 16);
    }
    for (int i = tid; i < S; i += QT) un[i] = q.rankOf(cur)[i] < 0 ? 1 : 0;
    __syncthreads();
    int tpk, nUnproc;
    bscan2_inclusive(pk, m, un, S, sw, tpk, nUnproc);
    const int totalChildren = tpk & 0xFFFF, nToExpand = tpk >> 16;
    const int S2 = totalChildren + nUnproc;
    finish = (S2 >= N) || (S2 == S);  // :671
    const int modeB2 = (!finish && (S2 + 3 * nToExpand > N)) ? 1 : 0;  // :675
    int Mp = 2;
    if (modeB2) {
        while (Mp < nToExpand) Mp <<= 1;
        for (int i = tid; i < Mp; i += QT) q.skey[i] = 0ull;
        __syncthreads();
    }
    for (int i = tid; i < S; i += QT) {
        const int r = q.rankOf(cur)[i];
        const uint32_t pth = q.path(cur)[i];
        if (r >= 0) {
            const int inc = pk[r];
            const int32_t *c = quad(i);
            const int c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
            const int cn[4] = {c0, c1, c2, c3};
            int pos = totalChildren - (inc & 0xFFFF);
            int mr = nToExpand - (inc >> 16);                                                 // multi-rank of the first multi-key child in list order
            int seq = (inc >> 16) - ((c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1));              // creation sequence of n1's slot
            int seqq[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) { seqq[qd] = seq; seq += cn[qd] > 1 ? 1 : 0; }
            const uint32_t cpath = (pth & 0xFFFF0000u) | ((((pth >> 12) & 0xFu) + 1u) << 12) | ((pth & 0xFFFu) << 2);
#pragma unroll
            for (int qd = 3; qd >= 0; --qd) {  // list front holds n4, then n3, n2, n1 (:623-662)
                if (cn[qd] > 0) {
                    q.cnt(nx)[pos] = cn[qd];
                    q.path(nx)[pos] = cpath | (uint32_t)qd;
                    if (cn[qd] > 1) {
                        if (modeB2) {
                            q.skey[seqq[qd]] = ((unsigned long long)(uint32_t)cn[qd] << 32) |
                                               ((unsigned long long)(uint32_t)(seqq[qd] + 1) << 16) | (unsigned long long)pos;
                        } else {
                            q.P[mr] = pos;
                            q.rankOf(nx)[pos] = mr;
                        }
                        ++mr;
                    } else if (!modeB2) {
                        q.rankOf(nx)[pos] = -1;
                    }
                    ++pos;
                }
            }
        } else {
            const int pos = totalChildren + un[i] - 1;
            q.cnt(nx)[pos] = q.cnt(cur)[i];
            q.path(nx)[pos] = pth;
            if (!modeB2) q.rankOf(nx)[pos] = -1;
        }
    }
    __syncthreads();
    if (!finish && modeB2) qt_sort_assign(q, Mp, nToExpand, S2, nx);
    __syncthreads();
    S = S2;
    m = finish ? 0 : nToExpand;
    cur = nx;
    modeB = modeB2;
}

template <bool DEEP>
__device__ __forceinline__ void qt_node_phase(const QtShared &q, int N, int &S, int &m, int &cur, int &modeB, bool &finish)
{
    const int tid = threadIdx.x, QT = blockDim.x, lane = tid & 63;
    const int nx = cur ^ 1;
    int32_t *sw = q.misc + 32;
    int32_t *childpos = q.cc;
    // non-empty children per processing rank, inclusive sums, stop rank R
    for (int r = tid; r < m; r += QT) {
        const int i = q.P[r];
        q.acc[r] = (q.cc[i * 4] > 0) + (q.cc[i * 4 + 1] > 0) + (q.cc[i * 4 + 2] > 0) + (q.cc[i * 4 + 3] > 0);
    }
    if (tid == 0) q.misc[6] = m;
    __syncthreads();
    bscan_inclusive(q.acc, m, sw);
    int R = m;
    if (modeB) {  // first rank whose split brings the list to >= N nodes (:732)
        int rmin = m;
        for (int r = tid; r < m; r += QT)
            if (S + q.acc[r] - (r + 1) >= N) rmin = min(rmin, r + 1);
        rmin = wave_min_i(rmin);
        if (lane == 0 && rmin < m) atomicMin(&q.misc[6], rmin);
        __syncthreads();
        R = q.misc[6];
    }
    const int totalChildren = R > 0 ? q.acc[R - 1] : 0;
    // unprocessed nodes keep their relative order behind the new children
    for (int i = tid; i < S; i += QT) {
        const int r = q.rankOf(cur)[i];
        q.newIdx[i] = (r >= 0 && r < R) ? 0 : 1;
    }
    __syncthreads();
    const int nUnproc = bscan_inclusive(q.newIdx, S, sw);
    const int S2 = totalChildren + nUnproc;
    // write the next list; leave the old->new map (newIdx / childpos) for whoever follows the keys
    for (int i = tid; i < S; i += QT) {
        const int r = q.rankOf(cur)[i];
        const uint32_t pth = q.path(cur)[i];
        if (r >= 0 && r < R) {
            int pos = totalChildren - q.acc[r];
            int ulx = 0, uly = 0, urx = 0, bry = 0, midx = 0, midy = 0;
            if (DEEP) {
                ulx = q.box(cur, 0)[i]; uly = q.box(cur, 1)[i];
                urx = q.box(cur, 2)[i]; bry = q.box(cur, 3)[i];
                midx = ulx + ((urx - ulx + 1) >> 1); midy = uly + ((bry - uly + 1) >> 1);  // ceil(w/2) (:480-481)
            }
            const uint32_t cpath = (pth & 0xFFFF0000u) | ((((pth >> 12) & 0xFu) + 1u) << 12) | ((pth & 0xFFFu) << 2);
            for (int qd = 3; qd >= 0; --qd) {  // list front holds n4, then n3, n2, n1 (:623-662)
                const int cn = q.cc[i * 4 + qd];
                if (cn > 0) {
                    if (DEEP) {
                        q.box(nx, 0)[pos] = (int16_t)((qd & 1) ? midx : ulx);
                        q.box(nx, 1)[pos] = (int16_t)((qd & 2) ? midy : uly);
                        q.box(nx, 2)[pos] = (int16_t)((qd & 1) ? urx : midx);
                        q.box(nx, 3)[pos] = (int16_t)((qd & 2) ? bry : midy);
                    }
                    q.cnt(nx)[pos] = cn;
                    q.path(nx)[pos] = cpath | (uint32_t)qd;
                    childpos[i * 4 + qd] = pos;
                    ++pos;
                } else {
                    childpos[i * 4 + qd] = -1;
                }
            }
            q.newIdx[i] = -1;
        } else {
            const int pos = totalChildren + q.newIdx[i] - 1;
            if (DEEP) {
                q.box(nx, 0)[pos] = q.box(cur, 0)[i];
                q.box(nx, 1)[pos] = q.box(cur, 1)[i];
                q.box(nx, 2)[pos] = q.box(cur, 2)[i];
                q.box(nx, 3)[pos] = q.box(cur, 3)[i];
            }
            q.cnt(nx)[pos] = q.cnt(cur)[i];
            q.path(nx)[pos] = pth;
            q.newIdx[i] = pos;
        }
    }
    __syncthreads();
    // multi-key children in creation order (rank asc, n1..n4): counts per rank -> sequence numbers
    for (int r = tid; r < R; r += QT) {
        const int i = q.P[r];
        int mc = 0;
        for (int qd = 0; qd < 4; ++qd) {
            const int pos = childpos[i * 4 + qd];
            if (pos >= 0 && q.cnt(nx)[pos] > 1) ++mc;
        }
        q.acc[r] = mc;
    }
    __syncthreads();
    const int nToExpand = bscan_inclusive(q.acc, R, sw);
    // termination / next mode (:671-675, :736)
    finish = (S2 >= N) || (S2 == S);
    int modeB2 = modeB;
    if (!modeB && !finish && (S2 + 3 * nToExpand > N)) modeB2 = 1;
    int m2 = 0;
    if (!finish) {
        if (!modeB2) {
            // list order of the multi-key nodes of the new list; P/rankOf of the OLD list are dead now
            int32_t *flag = (int32_t *)q.skey;
            for (int i = tid; i < S2; i += QT) flag[i] = q.cnt(nx)[i] > 1 ? 1 : 0;
            __syncthreads();
            m2 = bscan_inclusive(flag, S2, sw);
            for (int i = tid; i < S2; i += QT) {
                const bool multi = q.cnt(nx)[i] > 1;
                const int r = flag[i] - 1;
                q.rankOf(nx)[i] = multi ? r : -1;
                if (multi) q.P[r] = i;
            }
        } else {
            // sort the new multi-key children by (size desc, creation seq desc) (:686-687)
            int Mp = 2;
            while (Mp < nToExpand) Mp <<= 1;
            for (int i = tid; i < Mp; i += QT) q.skey[i] = 0ull;
            __syncthreads();
            for (int r = tid; r < R; r += QT) {
                const int i = q.P[r];
                int mc = 0;
                for (int qd = 0; qd < 4; ++qd) {
                    const int pos = childpos[i * 4 + qd];
                    if (pos >= 0 && q.cnt(nx)[pos] > 1) ++mc;
                }
                int seq = q.acc[r] - mc;  // acc is inclusive
                for (int qd = 0; qd < 4; ++qd) {
                    const int pos = childpos[i * 4 + qd];
                    if (pos >= 0 && q.cnt(nx)[pos] > 1) {
                        q.skey[seq] = ((unsigned long long)(uint32_t)q.cnt(nx)[pos] << 32) |
                                      ((unsigned long long)(uint32_t)(seq + 1) << 16) | (unsigned long long)pos;
                        ++seq;
                    }
                }
            }
            __syncthreads();
            qt_sort_assign(q, Mp, nToExpand, S2, nx);
            m2 = nToExpand;
        }
    }
    __syncthreads();
    S = S2;
    m = m2;
    cur = nx;
    modeB = modeB2;
}

// -DQT_PROFILE: per-phase clock stamps of k_octree summed into the words behind the overflow word (developer builds only;
// read with orbfe_internal_read_misc, tools/octree_phases.py)
#ifdef QT_PROFILE
#define QT_STAMP(p)                                                                                              \
    do {                                                                                                         \
        if (tid == 0) {                                                                                          \
            const unsigned long long t_now = wall_clock64();                                                     \
            atomicAdd((unsigned long long *)ovf + 8 + (p) + 8 * min(level, 14), t_now - t_prev);               \
            t_prev = t_now;                                                                                      \
        }                                                                                                        \
    } while (0)
#else
#define QT_STAMP(p)
#endif
#ifndef QT_MIN_WAVES
#define QT_MIN_WAVES 6   // waves per SIMD the register allocation must allow (A/B on the GPU box: tools/ab_build.sh)
#endif
template <bool GNODES>
__global__ __launch_bounds__(QT_MAX, QT_MIN_WAVES) void k_octree(const OrbPlan *__restrict__ plan,
                                               const uint2 *__restrict__ skeys,     // [B][keys_per_frame] {key, ord} from k_fast_map
                                               const int32_t *__restrict__ scount,  // [B][nlevels] * NK_STRIDE
                                               const uint32_t *__restrict__ cflags, // [B][nlevels][cf_words] from k_fast_map
                                               int32_t cf_words,
                                               uint16_t *__restrict__ knode,        // [B][keys_per_frame] scratch (deep trees only)
                                               int16_t *__restrict__ qtbox,         // [B][nlevels][box_stride] scratch (deep trees only)
                                               int32_t box_stride,
                                               char *__restrict__ qtnodes,          // [B][nlevels][node_stride] node arrays (GNODES only)
                                               int64_t node_stride,
                                               int32_t *__restrict__ nkeys,         // [B][nlevels] out (taps)
                                               uint32_t *__restrict__ sel,          // [B][sel_per_frame] out
                                               int32_t *__restrict__ nsel,          // [B][nlevels] out
                                               int32_t *__restrict__ ovf,           // sticky overflow word
                                               OctGroup g)                          // the levels this launch covers
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Workgroups go round-robin to the 8 XCDs in launch order; with level = blockIdx.x every XCD would own ONE pyramid
    // level of all frames, and level 0 carries ~10x the keys of level 7.  Rotating the level by the frame index gives
    // every XCD the same mix of levels.
    const int b = blockIdx.y, level = g.level0 + (int)((blockIdx.x + blockIdx.y) % gridDim.x), tid = threadIdx.x;
    const int lane = tid & 63;
    const int QT = blockDim.x;  // 128 .. 512 (per level group)
    const OrbLevel &L = plan->lv[level];
    const int N = L.nfeat;
    QtShared q;
    qt_carve(smem, GNODES ? qtnodes + ((int64_t)b * plan->nlevels + level) * node_stride : smem, g.M, g.nini, g.tw, g.th, g.ncells, q);
    q.gbox = qtbox + ((int64_t)b * plan->nlevels + level) * box_stride;
    int32_t *misc = q.misc;
    const uint2 *SK = skeys + (int64_t)b * plan->keys_per_frame + L.key_off;
    uint16_t *KN = knode + (int64_t)b * plan->keys_per_frame + L.key_off;
    const int nini = L.nini;
    const int ybot = L.h - 2 * ORBFE_MINB;  // maxBorderY - minBorderY
#ifdef QT_PROFILE
    unsigned long long t_prev = wall_clock64();
#endif

    // ---- prologue 1: the reference's per-cell threshold fallback (:818-825): a cell contributes {A > iniTh} if that is
    // non-empty, else all its NMS survivors ({A > minTh}).  Which cells have an iniTh survivor was recorded by k_fast_map as
    // it emitted them (one bit per cell): no pass over the keys is spent on it here.
    const int ns_all = scount[(b * plan->nlevels + level) * ORBFE_NK_STRIDE];
    const int ns = min(ns_all, L.key_cap);
    if (tid == 0 && ns_all > L.key_cap) atomicOr(ovf, 1);  // k_fast_map dropped survivors: results would be truncated
    uint32_t *cflag = q.cflag;  // bitmap over this level's cells; stays valid to the end of the kernel
    const int nwords = (L.ncells + 31) >> 5;
    {
        const uint32_t *gflag = cflags + (int64_t)(b * plan->nlevels + level) * cf_words;
        for (int i = tid; i < nwords; i += QT) cflag[i] = gflag[i];
    }
    for (int i = tid; i < nini * FF_PER_ROOT; i += QT) q.hist[i] = 0;
    if (tid == 0) misc[5] = 0;
    // DivideNode (:478-522) halves x and y independently (mid = UL + ceil(extent / 2)), so a key's 5-level quadrant
    // path is the bit-interleave of a 5-level x path (a function of the key's column and root) and a 5-level y path
    // (a function of its row): two small tables replace five DivideNode steps per key.
    const int winw = L.w - 2 * ORBFE_MINB;
    for (int i = tid; i < winw + ybot; i += QT) {
        const bool isx = i < winw;
        const int v = isx ? i : i - winw;
        int lo = 0, hi = ybot, r = 0;
        if (isx) {  // roots (:545-571): key -> root by (int)(x / hX)
            r = (int)__fdiv_rn((float)v, L.hx);
            r = min(max(r, 0), nini - 1);
            lo = L.root_x[r];
            hi = L.root_x[r + 1];
        }
        uint32_t code = 0;
#pragma unroll
        for (int d = 0; d < FFD; ++d) {
            const int mid = lo + ((hi - lo + 1) >> 1);
            const int hb = v < mid ? 0 : 1;
            code = (code << 2) | (uint32_t)hb;
            lo = hb ? mid : lo;
            hi = hb ? hi : mid;
        }
        if (isx) q.xtab[v] = (uint16_t)(((uint32_t)r << 10) | code);
        else q.ytab[v] = (uint16_t)(code << 1);
    }
    __syncthreads();
    QT_STAMP(0);
    const int ini = plan->ini_th;
    QT_STAMP(1);
    // ---- prologue 2: leaf histogram of the kept keys (a key is kept if it is above iniTh or its cell has no such key).
    // Keys are never moved or copied: whoever needs a key later re-derives "kept" and its path code from the key.
    auto key_kept = [&](const uint2 &e) {
        const uint32_t cell = e.y >> 12;
        return (int)orb_key_r(e.x) >= ini || !((cflag[cell >> 5] >> (cell & 31)) & 1u);
    };
    auto key_code = [&](const uint2 &e) {  // root << 10 | 5-level path code
        return (uint32_t)q.xtab[orb_key_x(e.x)] | (uint32_t)q.ytab[orb_key_y(e.x)];
    };
    int nkept = 0;
    for (int k0 = tid; k0 < ns; k0 += QT * KUNROLL) {
        uint2 e[KUNROLL];
#pragma unroll
        for (int u = 0; u < KUNROLL; ++u) e[u] = SK[min(k0 + u * QT, ns - 1)];
#pragma unroll
        for (int u = 0; u < KUNROLL; ++u)
            if (k0 + u * QT < ns && key_kept(e[u])) {
                const uint32_t rc = key_code(e[u]);
                atomicAdd(&q.hist[(int)(rc >> 10) * FF_PER_ROOT + ff_off(FFD) + (int)(rc & 0x3FFu)], 1);
                ++nkept;
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nkept += __shfl_xor(nkept, o, 64);
    if (lane == 0 && nkept) atomicAdd(&misc[5], nkept);
    __syncthreads();
    const int n = misc[5];
    if (tid == 0) nkeys[(b * plan->nlevels + level) * ORBFE_NK_STRIDE] = n;
    QT_STAMP(2);

    // ---- histogram passes: node-level bookkeeping by the whole workgroup, no key is touched ----
    for (int d = FFD - 1; d >= 1; --d) {  // quadrant sizes of every possible node of depth d-1 .. 4
        const int cntd = nini << (2 * d);
        for (int e = tid; e < cntd; e += QT) {
            const int r = e >> (2 * d), p = e & ((1 << (2 * d)) - 1);
            const int32_t *src = &q.hist[r * FF_PER_ROOT + ff_off(d + 1) + (p << 2)];
            q.hist[r * FF_PER_ROOT + ff_off(d) + p] = src[0] + src[1] + src[2] + src[3];
        }
        __syncthreads();
    }
    if (tid == 0) {  // roots (:545-587): nini boxes, empty roots erased; initial processing order: multi-key roots in list order
        int S0 = 0, m0 = 0;
        for (int r = 0; r < nini; ++r) {
            const int32_t *h1 = &q.hist[r * FF_PER_ROOT];
            const int cn = h1[0] + h1[1] + h1[2] + h1[3];
            if (cn > 0) {
                q.cnt(0)[S0] = cn;
                q.path(0)[S0] = (uint32_t)r << 16;
                if (cn > 1) { q.P[m0] = S0; q.rankOf(0)[S0] = m0; ++m0; }
                else q.rankOf(0)[S0] = -1;
                ++S0;
            }
        }
        misc[0] = S0;
        misc[1] = m0;
    }
    __syncthreads();
    int S = misc[0], m = misc[1], cur = 0, modeB = 0;
    bool finish = false;
    {
        int npass = 0;
        const int ffd = plan->dbg == 50 ? 0 : FFD;  // developer knob: 50 = streaming passes only
        while (!finish && npass < ffd) {  // nodes processed in pass p have depth <= p-1 <= 4: sizes come from the histogram
            if (!modeB && !(plan->dbg == 51)) {  // breadth-first pass (developer knob 51: generic passes only)
                qt_pass_bfs_hist(q, N, S, m, cur, modeB, finish);
                ++npass;
                continue;
            }
            for (int r = tid; r < m; r += QT) {
                const int i = q.P[r];
                const uint32_t pth = q.path(cur)[i];
                const int d = (int)((pth >> 12) & 0xFu), root = (int)(pth >> 16);
                const int32_t *src = &q.hist[root * FF_PER_ROOT + ff_off(d + 1) + (int)((pth & 0xFFFu) << 2)];
                q.cc[i * 4] = src[0];
                q.cc[i * 4 + 1] = src[1];
                q.cc[i * 4 + 2] = src[2];
                q.cc[i * 4 + 3] = src[3];
            }
            __syncthreads();
            qt_node_phase<false>(q, N, S, m, cur, modeB, finish);
            ++npass;
        }
    }
    const bool ff_done = finish;
    QT_STAMP(3);
    // path -> node table of the current list (overwrites the histogram): exactly one node of a key's path exists
    for (int i = tid; i < nini * FF_PER_ROOT; i += QT) q.hist[i] = -1;
    if (tid < ORBFE_MAX_ROOTS) misc[16 + tid] = -1;
    __syncthreads();
    for (int i = tid; i < S; i += QT) {
        const uint32_t pth = q.path(cur)[i];
        const int d = (int)((pth >> 12) & 0xFu), root = (int)(pth >> 16);
        if (d == 0) misc[16 + root] = i;
        else q.hist[root * FF_PER_ROOT + ff_off(d) + (int)(pth & 0xFFFu)] = i;
    }
    __syncthreads();
    // flatten the path -> node table: every leaf learns the one node of its path that exists (the deepest table hit),
    // in place -- a leaf entry is read and written by its own thread only, the shallower levels are read-only here
    for (int i = tid; i < nini << (2 * FFD); i += QT) {
        const int root = i >> (2 * FFD), code = i & ((1 << (2 * FFD)) - 1);
        int idx = misc[16 + root];
#pragma unroll
        for (int d = 1; d < FFD; ++d) {
            const int t = q.hist[root * FF_PER_ROOT + ff_off(d) + (code >> (2 * (FFD - d)))];
            idx = t >= 0 ? t : idx;
        }
        int32_t *leaf = &q.hist[root * FF_PER_ROOT + ff_off(FFD) + code];
        const int t = *leaf;
        *leaf = t >= 0 ? t : idx;
    }
    __syncthreads();
    auto node_of_code = [&](uint32_t kn) { return q.hist[(int)(kn >> 10) * FF_PER_ROOT + ff_off(FFD) + (int)(kn & 0x3FFu)]; };
    QT_STAMP(4);

    if (!ff_done) {
        // ---- deeper trees (clustered candidates): boxes of the current nodes from their paths, keys take their node
        // index, and the passes stream over the keys ----
        for (int i = tid; i < S; i += QT) {
            const uint32_t pth = q.path(cur)[i];
            const int d = (int)((pth >> 12) & 0xFu), root = (int)(pth >> 16);
            int ulx = L.root_x[root], urx = L.root_x[root + 1], uly = 0, bry = ybot;
            for (int s = d - 1; s >= 0; --s) {  // DivideNode along the path (:478-522)
                const int qd = (int)((pth >> (2 * s)) & 3u);
                const int midx = ulx + ((urx - ulx + 1) >> 1), midy = uly + ((bry - uly + 1) >> 1);
                if (qd & 1) ulx = midx; else urx = midx;
                if (qd & 2) uly = midy; else bry = midy;
            }
            q.box(cur, 0)[i] = (int16_t)ulx;
            q.box(cur, 1)[i] = (int16_t)uly;
            q.box(cur, 2)[i] = (int16_t)urx;
            q.box(cur, 3)[i] = (int16_t)bry;
        }
        for (int k = tid; k < ns; k += QT) {
            const uint2 e = SK[k];
            KN[k] = key_kept(e) ? (uint16_t)node_of_code(key_code(e)) : (uint16_t)0xFFFFu;  // 0xFFFF = dropped key
        }
        __syncthreads();
        for (int guard = 0; guard < 64; ++guard) {
            for (int i = tid; i < S * 4; i += QT) q.cc[i] = 0;
            __syncthreads();
            for (int k0 = tid; k0 < ns; k0 += QT * KUNROLL) {
                uint32_t kn[KUNROLL], kv[KUNROLL];
#pragma unroll
                for (int u = 0; u < KUNROLL; ++u) {
                    const int k = min(k0 + u * QT, ns - 1);
                    kn[u] = KN[k];
                    kv[u] = SK[k].x;
                }
#pragma unroll
                for (int u = 0; u < KUNROLL; ++u) {
                    const int k = k0 + u * QT;
                    if (k < ns && kn[u] != 0xFFFFu) {
                        const int i = (int)kn[u];  // node of the current list
                        if (q.cnt(cur)[i] > 1) {
                            const int ulx = q.box(cur, 0)[i], uly = q.box(cur, 1)[i];
                            const int midx = ulx + ((q.box(cur, 2)[i] - ulx + 1) >> 1);  // UL.x + ceil(w/2)  (:480)
                            const int midy = uly + ((q.box(cur, 3)[i] - uly + 1) >> 1);
                            const int qd = (orb_key_x(kv[u]) < midx ? 0 : 1) + (orb_key_y(kv[u]) < midy ? 0 : 2);
                            atomicAdd(&q.cc[i * 4 + qd], 1);
                            KN[k] = (uint16_t)((uint32_t)i | ((uint32_t)qd << 14));
                        }
                    }
                }
            }
            __syncthreads();
            qt_node_phase<true>(q, N, S, m, cur, modeB, finish);
            // keys follow their nodes into the new list (the quadrant counts have just become child positions)
            for (int k = tid; k < ns; k += QT) {
                const uint32_t kn = KN[k];
                if (kn != 0xFFFFu) {
                    const int i = (int)(kn & KNODE_MASK);
                    const int ni = q.newIdx[i];
                    KN[k] = (uint16_t)(ni >= 0 ? ni : q.cc[i * 4 + (int)(kn >> 14)]);
                }
            }
            __syncthreads();
            if (finish) break;
        }
    }

    // ---- keep the strongest key of every node, first in candidate order on ties (:746-762) ----
    // best = response (8 bit) | inverted ord (28 bit: first in candidate order wins ties) | key index (24 bit)
    unsigned long long *best = q.skey;
    for (int i = tid; i < S; i += QT) best[i] = 0ull;
    __syncthreads();
    QT_STAMP(5);
    for (int k0 = tid; k0 < ns; k0 += QT * KUNROLL) {
        uint2 e[KUNROLL];
        uint32_t kn[KUNROLL];
#pragma unroll
        for (int u = 0; u < KUNROLL; ++u) {
            const int k = min(k0 + u * QT, ns - 1);
            e[u] = SK[k];
            kn[u] = ff_done ? 0u : (uint32_t)KN[k];
        }
#pragma unroll
        for (int u = 0; u < KUNROLL; ++u) {
            const int k = k0 + u * QT;
            if (k < ns && (ff_done ? key_kept(e[u]) : kn[u] != 0xFFFFu)) {
                const int i = ff_done ? node_of_code(key_code(e[u])) : (int)kn[u];
                const unsigned long long cand = ((unsigned long long)orb_key_r(e[u].x) << 52) |
                                                ((unsigned long long)(0x0FFFFFFFu - e[u].y) << 24) | (unsigned long long)k;
                // neighbouring keys share nodes: a plain read filters most of them before the (serialising) atomic
                if (cand > best[i]) atomicMax(&best[i], cand);
            }
        }
    }
    __syncthreads();
    QT_STAMP(6);
    uint32_t *out = sel + (int64_t)b * plan->sel_per_frame + L.sel_off;
    const int nout = min(S, L.sel_cap);
    for (int i = tid; i < nout; i += QT) {
        const uint32_t key = SK[(uint32_t)(best[i] & 0xFFFFFFull)].x;
        // + minBorderX / minBorderY (:853-854): level coordinates from here on
        out[i] = orb_pack_key(orb_key_x(key) + ORBFE_MINB, orb_key_y(key) + ORBFE_MINB, orb_key_r(key));
    }
    if (tid == 0) {
        nsel[b * plan->nlevels + level] = nout;
        if (S > L.sel_cap) atomicOr(ovf, 2);
    }
    QT_STAMP(7);
}

// ---------------------------------------------------------------------------------------------------
// K4  7x7 Gaussian, sigma 2, 8-bit fixed-point kernel {18,34,49,55,49,34,18} (sum 257), REFLECT_101 at the
// LEVEL edges.  All levels of all frames in ONE launch.  One wave owns a 256-px wide, BL_RB-row tall
// tile: every lane filters 4 adjacent pixels, walking down the rows with the last 7 row-sums in registers
// (no LDS, no intermediate traffic): per row 3 aligned dword loads (12-byte window), 4 row sums, 4 outputs,
// one dword store.  Reflected borders: rows by a wave-uniform index, columns by a per-byte path on edge lanes.
// ---------------------------------------------------------------------------------------------------
// weights of window dword d (pixels x-4+4d .. x-1+4d) for the output pixel x+j: its taps are window bytes j+1 .. j+7
__host__ __device__ constexpr uint32_t blur_hw(int j, int d)
{
    const int kern[7] = {18, 34, 49, 55, 49, 34, 18};
    uint32_t w = 0u;
    for (int b = 0; b < 4; ++b) {
        const int t = 4 * d + b - j - 1;
        if (t >= 0 && t <= 6) w |= (uint32_t)kern[t] << (8 * b);
    }
    return w;
}

// Work is described per LANE (a 4-pixel column, a short run of rows), packed by the host into single-level waves, so
// no lane idles on narrow levels.  Column borders are branch-free: every lane loads 3 dwords from a per-lane base that
// covers all (reflected) source pixels of its 12-byte window and rearranges them with per-lane byte selectors
// (identity for interior lanes); row borders are a per-lane reflected row index.
#ifndef BL_PF
#define BL_PF 4  // prefetch distance in rows (2 .. 6 measured: profiles/r06_ab_blur.json)
#endif
#ifdef BL_MIN_WAVES
#define BL_BOUNDS __launch_bounds__(256, BL_MIN_WAVES)
#else
#define BL_BOUNDS __launch_bounds__(256)
#endif
// the row walk of one lane; INTERIOR (wave-uniform, compile-time): the 12-byte window holds no reflected column; UP (wave-uniform,
// compile-time): the walk goes from the bottom of the row block to its top.
//
// Instruction budget of a row step (4 pixels per lane): 10 v_dot4 horizontal taps (border waves: 12, below), the vertical taps on
// u16 row-sum PAIRS -- a pair (row 2m, row 2m + 1) is formed once, on the odd step (4 v_lshl_or every other step), and the seven
// rows of an output are three pairs and one single row on even steps, the high half of a pair and three pairs on odd ones, i.e.
// four v_dot2 per pixel either way -- saturate_cast<uchar> by the dot products' own clamp (the accumulator starts at
// 0xFF000000 + the rounding constant, so a value >= 256 runs into 0xFFFFFFFF and byte 2 IS the saturated pixel), three v_perm to
// gather the four bytes, one compare against the lane's row count, one add each for the load and the store offset.
//
// Border waves (reflected columns): BORDER_REFLECT_101 folds the taps that fall outside the row onto pixels inside it, so a
// border lane reads its 12-byte window as it lies (pulled inside the row) and applies FOLDED weights: twelve per-lane weight
// dwords from the plan's table (OrbPlan::blur_wt, four lane types per level) instead of byte selectors -- no v_perm per row.
template <int MODE, bool INTERIOR, bool UP>
__device__ __forceinline__ void blur7_walk(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int pitch, const int dpitch,
                                           const int W, const int H, const int x, const int y0, const int nrows_lane, const bool active,
                                           const int nsteps, const int vec_w, const uint32_t *__restrict__ wtab)
{
    constexpr bool up = UP;
    // all sources of the lane's ten window pixels lie in [base, base + 12) (checked on the host for every level width); at the
    // right edge the window is pulled back so that it ends at the last pixel of the row
    const int base = INTERIOR ? x - 4 : min(max(x - 4, 0), W - 12);
    uint32_t wt[4][3];
    if (!INTERIOR) {
        const int r = W - x;
        const int type = x == 0 ? 1 : (r <= 4 ? 3 : (r <= 8 ? 2 : 0));
        const uint4 *t = (const uint4 *)(wtab + type * 12);
        const uint4 t0 = t[0], t1 = t[1], t2 = t[2];
        wt[0][0] = t0.x; wt[0][1] = t0.y; wt[0][2] = t0.z; wt[1][0] = t0.w;
        wt[1][1] = t1.x; wt[1][2] = t1.y; wt[2][0] = t1.z; wt[2][1] = t1.w;
        wt[2][2] = t2.x; wt[3][0] = t2.y; wt[3][1] = t2.z; wt[3][2] = t2.w;
    }
    const bool full = x + 4 <= W;
    const uint32_t nrows = active ? (uint32_t)nrows_lane : 0u;

    // Row sums are <= 255 * 257 = 65535, i.e. u16.  Q[m & 3] holds, per pixel, the pair (row sum of step 2m, row sum of step
    // 2m + 1) as two u16 halves; hs = the row sums of the even step the pair is waiting for.
    uint32_t Q[4][4], hs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hs[j] = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) Q[k][j] = 0u;

    // raw rows are fetched BL_PF steps ahead into an 8-slot ring (the unroll factor), so a wave keeps several rows in flight
    uint32_t Lr[8][3];
    // A wave none of whose lanes comes within 3 rows of the level's top or bottom (three of four) walks plain rows: the
    // offset advances by the pitch, no reflected row index per step.
    // A lane walks its rows downwards from y0 - 3 or (flag bit 3, odd row blocks) upwards from yend + 2: the taps are symmetric,
    // the sums are the same integers.  Row of step s: ystart + dir * s; the output row of step s lies 3 * dir behind it, i.e. it is
    // output row s - 6 of the lane's block counted from the end the walk started at.
    const int dir = up ? -1 : 1;
    const int yend = y0 + nrows_lane;
    const int ystart = up ? yend + 2 : y0 - 3;
    const int ylast = ystart + dir * (nsteps + BL_PF - 1);   // last row the walk asks for (incl. the prefetch past its end)
    const bool plain_rows = orb_ballot(!(min(ystart, ylast) >= 0 && max(ystart, ylast) < H)) == 0ull;
    uint32_t ro = __umul24((uint32_t)min(max(ystart, 0), H - 1), (uint32_t)pitch) + (uint32_t)base;
    const uint32_t rstep = up ? 0u - (uint32_t)pitch : (uint32_t)pitch;
    uint32_t oo = __umul24((uint32_t)(up ? max(yend - 1, 0) : y0), (uint32_t)dpitch) + (uint32_t)x;   // output offset of step 6
    const uint32_t ostep = up ? 0u - (uint32_t)dpitch : (uint32_t)dpitch;
    auto fetch = [&](int s, uint32_t (&dst3)[3]) {
        const uint8_t *row;
        if (plain_rows) {
            row = src + ro;
            ro += rstep;
        } else {
            const int yy = reflect101(max(min(ystart + dir * s, H + 2), -3), H);
            row = src + (__umul24((uint32_t)yy, (uint32_t)pitch) + (uint32_t)base);
        }
        dst3[0] = *(const uint32_t *)(row);
        dst3[1] = *(const uint32_t *)(row + 4);
        dst3[2] = *(const uint32_t *)(row + 8);
    };
#pragma unroll
    for (int k = 0; k < BL_PF; ++k) fetch(k, Lr[k]);

    constexpr uint32_t ACC0 = 0xFF000000u + 32768u;   // clamp bias (see above) + round half up
    for (int s0 = 0; s0 < nsteps; s0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = s0 + k;   // s0 is a multiple of 8: the parity of s is the parity of k
            if (s >= nsteps) break;  // wave-uniform
            fetch(s + BL_PF, Lr[(k + BL_PF) % 8]);  // rows past the run re-read a valid (reflected / clamped) row
            const uint32_t w[3] = {Lr[k][0], Lr[k][1], Lr[k][2]};
            // horizontal taps as byte dot products against per-(pixel, dword) weight dwords: compile-time constants in interior
            // waves, the lane's folded weights in border waves
            uint32_t hn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t h = 0u;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    if (INTERIOR) {
                        if (blur_hw(j, d) != 0u) h = __builtin_amdgcn_udot4(w[d], blur_hw(j, d), h, false);
                    } else {
                        h = __builtin_amdgcn_udot4(w[d], wt[j][d], h, false);
                    }
                }
                hn[j] = h;
            }
            const int m = k >> 1;
            if (k & 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Q[m][j] = hs[j] | (hn[j] << 16);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) hs[j] = hn[j];
            }
            if (s >= 6) {
                uint32_t tq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // rows y-3 .. y+3 are steps s-6 .. s with taps 18 34 49 55 49 34 18
                    uint32_t acc;
                    if (k & 1) {   // (s-7 | s-6) (s-5 | s-4) (s-3 | s-2) (s-1 | s): the pair just formed is the last
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, Q[(m + 1) & 3][j]), __builtin_bit_cast(orb_u2, 0x00120000u), ACC0, true);
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, Q[(m + 2) & 3][j]), __builtin_bit_cast(orb_u2, 0x00310022u), acc, true);
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, Q[(m + 3) & 3][j]), __builtin_bit_cast(orb_u2, 0x00310037u), acc, true);
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, Q[m][j]), __builtin_bit_cast(orb_u2, 0x00120022u), acc, true);
                    } else {       // (s-6 | s-5) (s-4 | s-3) (s-2 | s-1) and the single row s (a u16 in a dword: its high half is 0)
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, Q[(m + 1) & 3][j]), __builtin_bit_cast(orb_u2, 0x00220012u), ACC0, true);
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, Q[(m + 2) & 3][j]), __builtin_bit_cast(orb_u2, 0x00370031u), acc, true);
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, Q[(m + 3) & 3][j]), __builtin_bit_cast(orb_u2, 0x00220031u), acc, true);
                        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, hn[j]), __builtin_bit_cast(orb_u2, 0x00000012u), acc, true);
                    }
                    tq[j] = acc;  // byte 2 = the value rounded half-up and saturated (0xFFFFFFFF when it was >= 256)
                }
                if (MODE == 1) {
                    // SSE2 half-even: an exact half (low 16 bits zero) rounds to the even value inside the vectorised part of
                    // the row.  One pixel in 65536 is an exact half, so the test is one wave-uniform branch on the smallest
                    // low half of the lane's four sums; the per-pixel correction runs only when some lane has one.  (A saturated
                    // sum has low half 0xFFFF: never corrected, and 256 or 257 saturate to 255 either way.)
                    const uint32_t lowmin = min(min(tq[0] & 0xFFFFu, tq[1] & 0xFFFFu), min(tq[2] & 0xFFFFu, tq[3] & 0xFFFFu));
                    if (orb_ballot(lowmin == 0u) != 0ull) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((tq[j] & 0xFFFFu) == 0u && (x + j) < vec_w && (tq[j] & 0x10000u)) tq[j] -= 0x10000u;
                    }
                }
                const uint32_t p01 = __builtin_amdgcn_perm(tq[1], tq[0], 0x0c0c0602u);
                const uint32_t p23 = __builtin_amdgcn_perm(tq[3], tq[2], 0x06020c0cu);
                const uint32_t packed = p01 | p23;
                if ((uint32_t)(s - 6) < nrows) {
                    if (full) {
                        *(uint32_t *)(dst + oo) = packed;   // uniform base + 32-bit lane offset: no 64-bit address arithmetic
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (x + j < W) dst[oo + (uint32_t)j] = (uint8_t)(packed >> (8 * j));
                    }
                }
                oo += ostep;
            }
        }
    }
}

template <int MODE>
__global__ BL_BOUNDS void k_blur7(const OrbPlan *__restrict__ plan, FrameSrc fs,
                                               const OrbLane *__restrict__ lanes, int nwaves,
                                               uint8_t *__restrict__ blur, int64_t blur_fstride)
{
    int b = blockIdx.y, bx = blockIdx.x;
    xcd_frame_remap(bx, b);
    const int lane = threadIdx.x & 63;
    const int t = bx * 4 + (threadIdx.x >> 6);
    if (t >= nwaves) return;
    const OrbLane ld = lanes[(int64_t)t * 64 + lane];
    const int level = __builtin_amdgcn_readfirstlane((int)(ld.flags >> 8));
    const OrbLevel &L = plan->lv[level];
    int pitch;
    const uint8_t *src = level_ptr(fs, L, level, b, &pitch);
    uint8_t *dst = blur + (int64_t)b * blur_fstride + L.off;
    const int W = L.w, H = L.h;
    const int x = ld.x, y0 = ld.ys, nr = ld.nrows;
    const bool active = !(ld.flags & 1);
    // wave-uniform by construction (the host packs interior and edge columns into separate waves): no reflected column
    const bool interior = __builtin_amdgcn_readfirstlane((int)(ld.flags & 2)) != 0;
    const int vec_w = W & ~3;
    const uint32_t *wtab = plan->blur_wt[level][0];
    int nsteps = ld.nrows;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, o, 64));
    nsteps = __builtin_amdgcn_readfirstlane(nsteps) + 6;  // wave-uniform
    const bool up = __builtin_amdgcn_readfirstlane((int)(ld.flags & 8)) != 0;   // wave-uniform by construction, like `interior`
    if (interior) {
        if (up) blur7_walk<MODE, true, true>(src, dst, pitch, (int)L.pitch, W, H, x, y0, nr, active, nsteps, vec_w, wtab);
        else blur7_walk<MODE, true, false>(src, dst, pitch, (int)L.pitch, W, H, x, y0, nr, active, nsteps, vec_w, wtab);
    } else {
        if (up) blur7_walk<MODE, false, true>(src, dst, pitch, (int)L.pitch, W, H, x, y0, nr, active, nsteps, vec_w, wtab);
        else blur7_walk<MODE, false, false>(src, dst, pitch, (int)L.pitch, W, H, x, y0, nr, active, nsteps, vec_w, wtab);
    }
}

#ifdef ORBFE_DEVELOPER   // measured slower than the default chain; compiled only into developer builds
// Blur of level l AND the resize l -> l + 1 in one pass over level l (ORBFE_FUSE_BLUR_PYR, one launch per level, chained).
// A lane keeps its blur job (a 4-pixel column of a row block, k_blur7's code unchanged) and, in the same row walk, produces
// one 4-pixel destination dword of level l + 1 for the destination rows whose upper source row lies in its row block
// (k_pyr_walk's arithmetic unchanged: horizontal sums of every source row formed once, a destination row completes in the
// step of its lower source row).  The resize part fetches its own unaligned 8-byte window per row -- the rows are the ones
// the workgroup's blur lanes are loading at that moment, so they come from L1 / L2: level l is read from HBM ONCE for both
// jobs instead of once by k_pyr_walk and once by k_blur7.  Vertical taps of level l + 1 sit in LDS.
struct BlurPyrArgs {
    uint8_t *dst;          // level l + 1, frame 0 (null: last level, blur only)
    int64_t dst_fstride;
    int32_t dpitch, dh;
    const OrbTab *xtab, *ytab;
    int32_t sw, sh;        // size of level l
    int32_t wave_lo;       // first wave of level l in the blur lane list
    int32_t split;         // 1: the resize jobs are in waves of their own (lane flag bit 2), the blur lanes only blur
};
template <int MODE, int SPLIT>
__global__ BL_BOUNDS void k_blur_pyr(const OrbPlan *__restrict__ plan, FrameSrc fs,
                                     const OrbLane *__restrict__ lanes, const OrbLaneR *__restrict__ lanesR, int nwaves,
                                     uint8_t *__restrict__ blur, int64_t blur_fstride, int level, BlurPyrArgs pa)
{
    extern __shared__ uint2 s_yt[];   // [dh + 8] of level l + 1: .x = b0 | b1 << 16, .y = sy
    const bool has_next = pa.dst != nullptr;
    const bool inlane = has_next && !SPLIT;   // the lane's own resize job (compiled out of the SPLIT instantiation)
    if (has_next)
        for (int i = threadIdx.x; i < pa.dh + 8; i += 256) s_yt[i] = ((const uint2 *)pa.ytab)[i];
    __syncthreads();
    int b = blockIdx.y, bx = blockIdx.x;
    xcd_frame_remap(bx, b);
    const int lane = threadIdx.x & 63;
    const int t = bx * 4 + (threadIdx.x >> 6);
    if (t >= nwaves) return;
    const OrbLane ld = lanes[(int64_t)(pa.wave_lo + t) * 64 + lane];
    const OrbLevel &L = plan->lv[level];
    const OrbLaneR lr = lanesR[(int64_t)(pa.wave_lo + t) * 64 + lane];
    if (SPLIT && __builtin_amdgcn_readfirstlane((int)(ld.flags & 4)) != 0) {
        // ---- a RESIZE wave (pa.split): k_pyr_walk's walk, one destination dword x a run of destination rows per lane; it sits
        // in the wave list next to the blur waves of the same source rows, so whichever of the two touches a row second finds it
        // in L1 / L2 ----
        int pitch;
        const uint8_t *src = level_ptr(fs, L, level, b, &pitch);
        uint8_t *dstn = pa.dst + (int64_t)b * pa.dst_fstride;
        const int dx0 = 4 * (int)lr.dj, y0 = (int)lr.d0, yend = y0 + (int)lr.nd;
        const uint4 tx01 = *(const uint4 *)(pa.xtab + dx0), tx23 = *(const uint4 *)(pa.xtab + dx0 + 2);
        const uint32_t xc[4] = {tx01.x, tx01.z, tx23.x, tx23.z};
        const int xs[4] = {(int)(short)tx01.y, (int)(short)tx01.w, (int)(short)tx23.y, (int)(short)tx23.w};
        const int sx0 = min(xs[0], pa.sw - 8);
        uint32_t sel[4];
        orb_u2 coef[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t o = (uint32_t)min(max(xs[j] - sx0, 0), 7);
            sel[j] = 0x0c000c00u | (min(o + 1u, 7u) << 16) | o;
            coef[j] = __builtin_bit_cast(orb_u2, xc[j]);
        }
        uint2 cur = s_yt[y0];
        const int r0 = (int)(short)cur.y;
        int nsteps = yend > y0 ? (int)(short)s_yt[yend - 1].y + 2 - r0 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, o, 64));
        nsteps = __builtin_amdgcn_readfirstlane(nsteps);
        const uint32_t sp = (uint32_t)pitch;
        const int rlast = pa.sh - 1;
        auto fetch = [&](int s, uint2 &q) { q = *(const uint2 *)(src + (__umul24((uint32_t)min(r0 + s, rlast), sp) + (uint32_t)sx0)); };
        auto hsum = [&](const uint2 &q, uint32_t (&h)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                h[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, __builtin_amdgcn_perm(q.y, q.x, sel[j])), coef[j], 0u, false) >> 4;
        };
        uint2 raw[4];
        fetch(0, raw[0]);
        fetch(1, raw[1]);
        fetch(2, raw[2]);
        uint32_t Hp[4];
        hsum(raw[0], Hp);
        int d = y0;
        for (int s0 = 1; s0 < nsteps; s0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int s = s0 + k;
                fetch(s + PW_PF, raw[(k + 1 + PW_PF) % 4]);
                uint32_t Hs[4];
                hsum(raw[(k + 1) % 4], Hs);
                const bool emit = d < yend && (int)(short)cur.y + 1 == r0 + s;
                const uint32_t b0 = cur.x & 0xFFFFu, b1 = cur.x >> 16;
                uint32_t va[4], vb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    va[j] = __umul24(b0, Hp[j]);
                    vb[j] = __umul24(b1, Hs[j]) + 0x20000u;
                }
                uint32_t t01, t23;
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t01) : "v"(va[0]), "v"(vb[0]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t01) : "v"(va[1]), "v"(vb[1]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t23) : "v"(va[2]), "v"(vb[2]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t23) : "v"(va[3]), "v"(vb[3]));
                const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t01) >> (orb_u2)(2));
                const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t23) >> (orb_u2)(2));
                if (emit) {
                    *(uint32_t *)(dstn + (__umul24((uint32_t)d, (uint32_t)pa.dpitch) + (uint32_t)dx0)) = __builtin_amdgcn_perm(q23, q01, 0x06040200u);
                    d += 1;
                }
                cur = s_yt[d];
#pragma unroll
                for (int j = 0; j < 4; ++j) Hp[j] = Hs[j];
            }
        }
        return;
    }
    // ---- the resize job of this lane (in-lane fusion, pa.split == 0): destination dword dj of level l + 1, rows [d, dend) ----
    const bool has_dst = inlane && lr.nd != 0;
    const int dj4 = has_dst ? 4 * (int)lr.dj : 0;
    int d = has_dst ? (int)lr.d0 : 0;
    const int dend = has_dst ? (int)lr.d0 + (int)lr.nd : 0;
    uint32_t rsel[4] = {0, 0, 0, 0};
    orb_u2 rcoef[4];
    int rsx0 = 0;
    if (inlane) {
        const uint4 tx01 = *(const uint4 *)(pa.xtab + dj4), tx23 = *(const uint4 *)(pa.xtab + dj4 + 2);
        const uint32_t xc[4] = {tx01.x, tx01.z, tx23.x, tx23.z};
        const int xs[4] = {(int)(short)tx01.y, (int)(short)tx01.w, (int)(short)tx23.y, (int)(short)tx23.w};
        rsx0 = min(xs[0], pa.sw - 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t o = (uint32_t)min(max(xs[j] - rsx0, 0), 7);
            rsel[j] = 0x0c000c00u | (min(o + 1u, 7u) << 16) | o;
            rcoef[j] = __builtin_bit_cast(orb_u2, xc[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) rcoef[j] = __builtin_bit_cast(orb_u2, 0u);
    }
    uint2 rcur = inlane ? s_yt[d] : make_uint2(0u, 0u);
    uint8_t *rdst = inlane ? pa.dst + (int64_t)b * pa.dst_fstride : nullptr;
    int pitch;
    const uint8_t *src = level_ptr(fs, L, level, b, &pitch);
    uint8_t *dst = blur + (int64_t)b * blur_fstride + L.off;
    const int W = L.w, H = L.h;
    const int x = ld.x, y0 = ld.ys, yend = y0 + ld.nrows;
    const bool active = !(ld.flags & 1);
    // wave-uniform by construction (the host packs interior and edge columns into separate waves): no reflected column
    const bool interior = __builtin_amdgcn_readfirstlane((int)(ld.flags & 2)) != 0;
    const int vec_w = W & ~3;
    int nsteps = ld.nrows;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, o, 64));
    nsteps = __builtin_amdgcn_readfirstlane(nsteps) + 6;  // wave-uniform

    // window pixel i (0..11) is level column reflect101(x - 4 + i); i = 0 and 11 are never used
    int srcx[12], lo = W;
#pragma unroll
    for (int i = 1; i < 11; ++i) {
        srcx[i] = reflect101(min(x - 4 + i, W + 2), W);
        lo = min(lo, srcx[i]);
    }
    srcx[0] = srcx[1];
    srcx[11] = srcx[10];
    // all ten sources lie in [base, base + 12) (checked on the host for every level width); at the right edge the
    // window is pulled back so that it ends at the last pixel of the row
    const int base = interior ? x - 4 : min(lo & ~3, W - 12);
    uint32_t selA[3], selB[3], mskB[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        selA[d] = selB[d] = mskB[d] = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bi = min(max(srcx[4 * d + k] - base, 0), 11);  // loaded byte index
            if (bi < 8) selA[d] |= (uint32_t)bi << (8 * k);           // from {w1:w0}
            else {
                selB[d] |= (uint32_t)(bi - 8) << (8 * k);             // from w2
                mskB[d] |= 0xFFu << (8 * k);
            }
        }
    }
    const bool full = x + 4 <= W;
    const int dpitch = L.pitch;

    // Row sums are <= 255 * 257 = 65535, i.e. u16: ring slot k holds, per pixel, the pair (row sum of step s-1, row sum of
    // step s) as two u16 halves, so the vertical pass is three v_dot2_u32_u16 (pairs of taps) plus one multiply-add
    // for the newest row instead of seven multiply / add steps.
    uint32_t S[7][4], Sprev[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Sprev[j] = 0u;
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[k][j] = 0u;

    // raw rows are fetched BL_PF steps ahead into the same 7-slot ring, so a wave keeps several rows in flight
    uint32_t Lr[7][3];
    // A wave none of whose lanes comes within 3 rows of the level's top or bottom (three of four) walks plain rows: the
    // offset advances by the pitch, no reflected row index per step.
    // A lane walks its rows downwards from y0 - 3 or (flag bit 3, odd row blocks) upwards from yend + 2: the taps are symmetric,
    // the sums are the same integers.  Row of step s: ystart + dir * s; the output row of step s lies 3 * dir behind it.
    const bool up = false;   // the fused pass walks every row block downwards (its resize jobs complete rows top to bottom)
    const int dir = up ? -1 : 1;
    const int ystart = up ? yend + 2 : y0 - 3;
    const int ylast = ystart + dir * (nsteps + BL_PF - 1);   // last row the walk asks for (incl. the prefetch past its end)
    const bool plain_rows = orb_ballot(!(min(ystart, ylast) >= 0 && max(ystart, ylast) < H)) == 0ull;
    uint32_t ro = __umul24((uint32_t)min(max(ystart, 0), H - 1), (uint32_t)pitch) + (uint32_t)base;
    const uint32_t rstep = up ? 0u - (uint32_t)pitch : (uint32_t)pitch;
    auto fetch = [&](int s, uint32_t (&dst3)[3]) {
        const uint8_t *row;
        if (plain_rows) {
            row = src + ro;
            ro += rstep;
        } else {
            const int yy = reflect101(max(min(ystart + dir * s, H + 2), -3), H);
            row = src + (__umul24((uint32_t)yy, (uint32_t)pitch) + (uint32_t)base);
        }
        dst3[0] = *(const uint32_t *)(row);
        dst3[1] = *(const uint32_t *)(row + 4);
        dst3[2] = *(const uint32_t *)(row + 8);
    };
#pragma unroll
    for (int k = 0; k < BL_PF; ++k) fetch(k, Lr[k]);
    // resize rows: source row of step s is y0 - 3 + s, clamped into the level (the virtual row sh repeats row sh - 1, which is
    // what cv::resize's clamped second tap reads); same prefetch distance, same 7-slot ring
    uint2 Rr[7];
    uint32_t Hp[4] = {0u, 0u, 0u, 0u};
    auto rfetch = [&](int s, uint2 &q) {
        const int yy = min(max(y0 - 3 + s, 0), H - 1);
        q = *(const uint2 *)(src + (__umul24((uint32_t)yy, (uint32_t)pitch) + (uint32_t)rsx0));
    };
    if (inlane) {
#pragma unroll
        for (int k = 0; k < BL_PF; ++k) rfetch(k, Rr[k]);
    }

    for (int s0 = 0; s0 < nsteps; s0 += 7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int s = s0 + k;
            if (s >= nsteps) break;  // wave-uniform
            const int yin = ystart + dir * s;
            fetch(s + BL_PF, Lr[(k + BL_PF) % 7]);  // rows past the run re-read a valid (reflected / clamped) row
            if (inlane) {   // wave-uniform (kernel argument); absent from the SPLIT instantiation
                rfetch(s + BL_PF, Rr[(k + BL_PF) % 7]);
                uint32_t Hs[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    Hs[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, __builtin_amdgcn_perm(Rr[k].y, Rr[k].x, rsel[j])), rcoef[j], 0u, false) >> 4;
                // destination row d completes in the step whose source row is sy(d) + 1 (sy strictly increasing: at most one per step)
                const bool emit = d < dend && (int)(short)rcur.y + 1 == yin;
                const uint32_t rb0 = rcur.x & 0xFFFFu, rb1 = rcur.x >> 16;
                uint32_t ra[4], rbv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ra[j] = __umul24(rb0, Hp[j]);
                    rbv[j] = __umul24(rb1, Hs[j]) + 0x20000u;
                }
                uint32_t t01, t23;
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t01) : "v"(ra[0]), "v"(rbv[0]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t01) : "v"(ra[1]), "v"(rbv[1]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t23) : "v"(ra[2]), "v"(rbv[2]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t23) : "v"(ra[3]), "v"(rbv[3]));
                const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t01) >> (orb_u2)(2));
                const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(orb_u2, t23) >> (orb_u2)(2));
                if (emit) {
                    *(uint32_t *)(rdst + (__umul24((uint32_t)d, (uint32_t)pa.dpitch) + (uint32_t)dj4)) = __builtin_amdgcn_perm(q23, q01, 0x06040200u);
                    d += 1;
                    rcur = s_yt[d];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) Hp[j] = Hs[j];
            }
            const uint32_t l0 = Lr[k][0], l1 = Lr[k][1], l2 = Lr[k][2];
            uint32_t w[3] = {l0, l1, l2};
            if (!interior) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const uint32_t ta = __builtin_amdgcn_perm(l1, l0, selA[d]);
                    const uint32_t tb = __builtin_amdgcn_perm(l2, l2, selB[d]);
                    w[d] = (tb & mskB[d]) | (ta & ~mskB[d]);
                }
            }
            // horizontal taps as byte dot products against per-(pixel, dword) weight constants
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t h = 0u;
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if (blur_hw(j, d) != 0u) h = __builtin_amdgcn_udot4(w[d], blur_hw(j, d), h, false);
                S[k][j] = Sprev[j] | (h << 16);
                Sprev[j] = h;
            }
            if (s >= 6) {
                const int y = yin - 3;
                uint32_t tq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // rows y-3 .. y+3 are steps s-6 .. s: pairs (s-6, s-5), (s-4, s-3), (s-2, s-1) sit in the slots written at
                    // steps s-5, s-3, s-1; the newest row sum is Sprev
                    uint32_t acc = __umul24(18u, Sprev[j]) + 32768u;  // v_mad_u32_u24
                    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, S[(k + 2) % 7][j]), __builtin_bit_cast(orb_u2, 0x00220012u), acc, false);
                    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, S[(k + 4) % 7][j]), __builtin_bit_cast(orb_u2, 0x00370031u), acc, false);
                    acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u2, S[(k + 6) % 7][j]), __builtin_bit_cast(orb_u2, 0x00220031u), acc, false);
                    tq[j] = acc;  // (acc >> 16) = value rounded half-up, <= 257
                }
                if (MODE == 1) {
                    // SSE2 half-even: an exact half (low 16 bits zero) rounds to the even value inside the vectorised part of
                    // the row.  One pixel in 65536 is an exact half, so the test is one wave-uniform branch on the smallest
                    // low half of the lane's four sums; the per-pixel correction runs only when some lane has one.
                    const uint32_t lowmin = min(min(tq[0] & 0xFFFFu, tq[1] & 0xFFFFu), min(tq[2] & 0xFFFFu, tq[3] & 0xFFFFu));
                    if (orb_ballot(lowmin == 0u) != 0ull) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((tq[j] & 0xFFFFu) == 0u && (x + j) < vec_w && (tq[j] & 0x10000u)) tq[j] -= 0x10000u;
                    }
                }
                // (acc >> 16) <= 257: the high halves of two sums side by side, saturate_cast<uchar> as one packed u16 min, then
                // the four low bytes into one dword
                const uint32_t h01 = pk_min_u16(__builtin_amdgcn_perm(tq[1], tq[0], 0x07060302u), 0x00FF00FFu);
                const uint32_t h23 = pk_min_u16(__builtin_amdgcn_perm(tq[3], tq[2], 0x07060302u), 0x00FF00FFu);
                const uint32_t packed = __builtin_amdgcn_perm(h23, h01, 0x06040200u);
                if (active && y < yend) {
                    uint8_t *o = dst + (__umul24((uint32_t)y, (uint32_t)dpitch) + (uint32_t)x);
                    if (full) {
                        *(uint32_t *)o = packed;
                    } else {
                        for (int j = 0; j < 4 && x + j < W; ++j) o[j] = (uint8_t)(packed >> (8 * j));
                    }
                }
            }
        }
    }
}
#endif  // ORBFE_DEVELOPER

// ---------------------------------------------------------------------------------------------------
// K5  IC_Angle + steered BRIEF + keypoint assembly.  One wave per output slot.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    // cv::fastAtan2 (OpenCV 3.2 atan_f32), every operation rounded separately
    const float s = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = __fmul_rn(0.9997878412794807f, s), p3 = __fmul_rn(-0.3258083974640975f, s);
    const float p5 = __fmul_rn(0.1555786518463281f, s), p7 = __fmul_rn(-0.04432655554792128f, s);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// canonical (float)cos / (float)sin of angle_deg * pi/180: fixed fp64 operation sequence
__device__ __forceinline__ void canon_sincos(float angle_deg, float *ca, float *sb)
{
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = __fmul_rn(angle_deg, factor_pi);
    const double x = (double)angle;
    const double kf = floor(__dadd_rn(__dmul_rn(x, 6.36619772367581382433e-01), 0.5));
    const int k = (int)kf;
    const double r = __dsub_rn(__dsub_rn(x, __dmul_rn(kf, 1.57079632673412561417e+00)),
                               __dmul_rn(kf, 6.07710050650619224932e-11));
    const double z = __dmul_rn(r, r);
    double ps = __dadd_rn(-2.50507602534068634195e-08, __dmul_rn(z, 1.58969099521155010221e-10));
    ps = __dadd_rn(2.75573137070700676789e-06, __dmul_rn(z, ps));
    ps = __dadd_rn(-1.98412698298579493134e-04, __dmul_rn(z, ps));
    ps = __dadd_rn(8.33333333332248946124e-03, __dmul_rn(z, ps));
    ps = __dadd_rn(-1.66666666666666324348e-01, __dmul_rn(z, ps));
    const double sn = __dadd_rn(r, __dmul_rn(__dmul_rn(z, r), ps));
    double pc = __dadd_rn(2.08757232129817482790e-09, __dmul_rn(z, -1.13596475577881948265e-11));
    pc = __dadd_rn(-2.75573143513906633035e-07, __dmul_rn(z, pc));
    pc = __dadd_rn(2.48015872894767294178e-05, __dmul_rn(z, pc));
    pc = __dadd_rn(-1.38888888888741095749e-03, __dmul_rn(z, pc));
    pc = __dadd_rn(4.16666666666666019037e-02, __dmul_rn(z, pc));
    const double cs = __dsub_rn(1.0, __dsub_rn(__dmul_rn(0.5, z), __dmul_rn(__dmul_rn(z, z), pc)));
    double s, c;
    switch (k & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *ca = __double2float_rn(c);
    *sb = __double2float_rn(s);
}

// 16 lanes per keypoint, 4 keypoints per wave, 4 waves per workgroup.
//   A  IC_Angle moments: the 31 rows of the patch as 8 unaligned dwords each; a lane takes dword k = sub & 7 of rows
//      (sub >> 3) + 2i.  m10 and the row sums come from v_dot4_u32_u8 against per-(row, dword) weight bytes
//      ((u + 15) inside the circle, 1 inside the circle), reduced over the 16 lanes -- integer, any order is exact.
//   B  fastAtan2 / canonical sincos per lane (16x redundant instead of 64x).
//   C  rBRIEF: the 37 x 40 blurred patch is staged in LDS with coalesced dword loads; lane `sub` evaluates pairs
//      16*sub .. 16*sub+15, i.e. descriptor bytes 2*sub and 2*sub+1, from LDS byte reads.
#ifndef DS_PP
#define DS_PP 40   // LDS patch pitch (bytes): columns x-18 .. x+21 (44 = an odd number of dwords per row: A/B in DESIGN.md)
#endif
#define DS_PR 37   // patch rows y-18 .. y+18

__constant__ uint2 c_momw[31 * 8];  // per (row v+15, dword k): .x = weights (u+15) or 0, .y = 1 or 0 per byte

// circular patch of IC_Angle (src/ORBextractor.cc:59-88): row v covers u in [-umax[|v|], umax[|v|]]
hipError_t orbk_upload_moment_weights(const int *umax16)
{
    uint2 h[31 * 8];
    for (int row = 0; row < 31; ++row) {
        const int v = row - 15, d = umax16[v < 0 ? -v : v];
        for (int k = 0; k < 8; ++k) {
            uint32_t w10 = 0, w1 = 0;
            for (int i = 0; i < 4; ++i) {
                const int u = 4 * k + i - 15;
                if (u >= -d && u <= d) {
                    w10 |= (uint32_t)(u + 15) << (8 * i);
                    w1 |= 1u << (8 * i);
                }
            }
            h[row * 8 + k] = make_uint2(w10, w1);
        }
    }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_momw), h, sizeof(h));
}

// Keypoints per workgroup (16 lanes each).  The pattern and moment-weight tables (6 KB) are per workgroup and a keypoint's
// blurred patch takes 1480 B of LDS: 16 keypoints -> 30 KB, 5 workgroups = 80 keypoints per CU; 32 keypoints -> 53 KB, 3
// workgroups = 96 keypoints per CU (A/B: profiles/r04_ab_experiments.json).
#ifndef DS_KPW
#define DS_KPW 16
#endif
// Launch geometry: a workgroup belongs to ONE level (its DS_KPW keypoints are consecutive entries of that level's selection), so
// the address of a keypoint's key follows from the block index and the kernel arguments alone -- no count, no plan in memory in
// front of it.  The dependent chain of a workgroup is then TWO round trips: {all level counts, the keys}, then {the 16 moment
// dwords and the 24 blurred-patch dwords of every lane, requested back to back}; it used to be four (counts + plan, key, moment
// pixels, patch pixels), and with a wave's 2.9 k cycles of arithmetic against tens of thousands of cycles of waiting the chain
// length is part of what the kernel's time follows (-7 %).  The output slot of a keypoint (level-major, :1103-1112) needs the
// counts of the levels before its own: they arrive with the key and are used only by the stores at the very end.
struct DescLevelArg { int32_t sel_off, off, pitch, wg0; float scale, patch_size; int32_t pad[2]; };   // wg0: first workgroup of the level
struct DescArgs {
    int32_t wg0[ORBFE_MAX_LEVELS];   // the same, side by side: one scalar load for the level scan (levels past the last: INT_MAX)
    int32_t nwg;                     // workgroups that belong to a level: sum over the levels of ceil(sel_cap / DS_KPW)
    int32_t pad[3];
    DescLevelArg lv[ORBFE_MAX_LEVELS];
};
__global__ __launch_bounds__(DS_KPW * 16) void k_orient_describe(DescArgs da, FrameSrc fs,
                                                         const uint8_t *__restrict__ blur, int64_t blur_fstride,
                                                         const uint32_t *__restrict__ sel,
                                                         const int32_t *__restrict__ nsel,
                                                         orbfe_keypoint *__restrict__ kps,
                                                         uint8_t *__restrict__ desc, int32_t cap,
                                                         int32_t *__restrict__ n_out, int32_t nl,
                                                         int32_t sel_per_frame, int32_t *__restrict__ ovf)
{
#ifndef DS_PATCH_PAD
#define DS_PATCH_PAD 0   // bytes between the patches of neighbouring keypoints (A/B of a bank stagger: profiles/r06_ab_describe_pad.json)
#endif
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[DS_KPW][DS_PR * DS_PP + DS_PATCH_PAD];
    __shared__ uint2 s_momw[31 * 8];
    __shared__ float4 s_pat[256];  // (x0, y0, x1, y1) of every test pair as floats
#ifdef DS_EXTRA_LDS   // occupancy probe: dead LDS that costs a workgroup slot per CU
    __shared__ uint32_t s_pad[DS_EXTRA_LDS / 4];
    if (nl < 0) s_pad[threadIdx.x] = 1u;
    if (nl < -1) ovf[0] = (int32_t)s_pad[threadIdx.x ^ 1];
#endif
    int b = blockIdx.y, bx = blockIdx.x;
    xcd_frame_remap(bx, b);
    b = __builtin_amdgcn_readfirstlane(b);  // workgroup-uniform: frame offsets are scalar 64-bit products
    bx = __builtin_amdgcn_readfirstlane(bx);
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = lane & 15, quad = tid >> 4;  // quad 0..DS_KPW-1 inside the workgroup = one keypoint
    // the level of this workgroup (scalar scan of the kernel arguments); workgroups behind the levels' only pad the output
    int level = -1;
    if (bx < da.nwg) {
        level = 0;
#pragma unroll
        for (int j = 1; j < ORBFE_MAX_LEVELS; ++j) level += bx >= da.wg0[j] ? 1 : 0;   // ascending; INT_MAX past the last level
    }
    const int lv = max(level, 0);
    const DescLevelArg L = da.lv[lv];
    const int idx = (bx - L.wg0) * DS_KPW + quad;   // index inside the level's selection
    // ---- round trip 1: the counts of all levels (lane `sub` of every 16-lane group holds level `sub`'s) and the key ----
    static_assert(ORBFE_MAX_LEVELS == 16, "level counts: one level per lane of a 16-lane group");
    const int cnt_l = sub < nl ? nsel[b * nl + sub] : 0;
    uint32_t key = 0;
    // entries behind the level's count are stale but inside its slice of the scratch (sel_cap rounded up to 64): read, not used
    if (level >= 0) key = sel[(int64_t)b * sel_per_frame + L.sel_off + idx];
    // ... and the two tables, requested in the same round trip (both loads before either LDS store: the load counter is in order)
    static_assert(DS_KPW * 16 >= 256, "table fill: one pattern entry and one moment-weight entry per thread");
    const uint2 mw = c_momw[min(tid, 31 * 8 - 1)];
    const uint32_t pt = ((const uint32_t *)c_pattern)[tid & 255];
    if (tid < 31 * 8) s_momw[tid] = mw;
    if (tid < 256) {
        // pair p = 16 * sub + i is stored at [i][sub]: the 16 lanes of a keypoint read consecutive float4s
        s_pat[(tid & 15) * 16 + (tid >> 4)] = make_float4((float)(int8_t)(pt & 0xFF), (float)(int8_t)((pt >> 8) & 0xFF),
                                                         (float)(int8_t)((pt >> 16) & 0xFF), (float)(int8_t)(pt >> 24));
    }
    // inclusive prefix sums of the level counts across the group's lanes (four DPP row shifts)
    int incl = cnt_l;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1 (lanes without a source add 0)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
    const int total = __builtin_amdgcn_readlane(incl, 15);                 // the same in every group: scalar
    const int mine = __builtin_amdgcn_readlane(cnt_l, lv);                 // keypoints of this workgroup's level
    const int before = lv > 0 ? __builtin_amdgcn_readlane(incl, max(lv - 1, 0)) : 0;   // keypoints of the levels in front of it
    if (bx == 0 && tid == 0) {
        n_out[b] = total;
        if (total > cap) atomicOr(ovf, 4);  // n_out holds the required count; slots >= cap are not written
    }
    {   // zero-fill the padding so the buffers can be all-gathered as they are: workgroup g takes slots total + 16 g ...
        // (the grid has at least cap / DS_KPW workgroups)
        const int zs = total + bx * DS_KPW + quad;
        if (zs < cap) {
            if (sub < 7) ((uint32_t *)(kps + (int64_t)b * cap + zs))[sub] = 0u;
            if (sub < 8) ((uint32_t *)(desc + ((int64_t)b * cap + zs) * 32))[sub] = 0u;
        }
    }
    // A workgroup behind its level's last keypoint (the levels' capacities are what the grid covers), or behind the levels: done.
    // Workgroup-uniform, before the barriers.
    if (level < 0 || (bx - L.wg0) * DS_KPW >= mine) return;
    const int slot = before + idx;
    const bool live = idx < mine && slot < cap;
    orbfe_keypoint *kp = kps + (int64_t)b * cap + slot;
    uint8_t *dd = desc + ((int64_t)b * cap + slot) * 32;
    // dead quads shadow a valid position so that every lane can run the same loads
    const int x = live ? orb_key_x(key) : ORBFE_EDGE, y = live ? orb_key_y(key) : ORBFE_EDGE;
    const int pitch = lv == 0 ? fs.l0_pitch : L.pitch;
    const uint8_t *img = lv == 0 ? fs.l0 + (int64_t)b * fs.l0_fstride : fs.pyr + (int64_t)b * fs.pyr_fstride + L.off;

    // ---- round trip 2: the moment dwords (unblurred level) and the blurred patch of every lane, requested back to back ----
    const int mk = sub & 7, mr0 = sub >> 3;
    uint32_t w[16];
    {
        // 32-bit offsets from 24-bit multiplies (the 32-bit multiply and the 64-bit multiply-add are quarter rate):
        // rows r0, r0 + 2, ...; the 16th row of the odd lanes (31) is clamped to 30 and not used
        const uint8_t *p = img + (__umul24((uint32_t)(y - 15 + mr0), (uint32_t)pitch) + (uint32_t)(x - 15 + 4 * mk));
        // unaligned dwords; the pointer advances by two rows per load (one 64-bit add each instead of a multiply and an add);
        // the last step of the odd lanes is one row (row 30, clamped)
        const uint32_t step2 = 2u * (uint32_t)pitch, step_last = __umul24((uint32_t)(2 - mr0), (uint32_t)pitch);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            w[i] = *(const uint32_t *)p;
            p += i < 14 ? step2 : step_last;
        }
    }
    uint8_t *patch = s_patch[quad];
    uint32_t v[24];
    {
        const int bpitch = L.pitch;
        // uniform base (frame b of the blurred pyramid) + a 32-bit per-lane offset that advances by additions
        typedef const __attribute__((address_space(1))) uint8_t *orb_gptr8;    // global memory, explicitly
        typedef const __attribute__((address_space(1))) uint32_t *orb_gptr32;
        orb_gptr8 bbase;
        {   // pinned to a scalar register pair: the loads below then take it as their SGPR base (the 64-bit product is formed
            // on the vector side, where the compiler no longer knows it is uniform)
            const uint64_t bb = (uint64_t)(blur + (int64_t)b * blur_fstride);
            const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bb);
            const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bb >> 32));
            bbase = (orb_gptr8)(((uint64_t)hi32 << 32) | lo32);
        }
        // dword f = it * 16 + sub of the 37 x 10 dword patch (16 consecutive dwords per step: a row and the start of the
        // next).  With the LDS pitch equal to the 40 patch bytes the LDS offset is simply 4 * f; the row of f is
        // (f * 205) >> 11 (= f / 10 for f < 1029) and the global offset  base + row * (pitch - 40) + 4 * f,  whose
        // "+ 64 * it" rides in the load's immediate offset: three VALU operations per load (the incremental carry logic this
        // replaces took ten).
        static_assert(DS_PR == 37 && DS_PP == 40, "patch staging: LDS pitch == patch bytes");
        uint32_t s205 = (uint32_t)sub * 205u;
        // kept as a value of its own: folded into a multiply-add with the step's constant, every load pays a register move for that
        // constant (the multiply-add cannot take a literal next to its scalar operand); as an addend it takes the literal directly
        asm volatile("" : "+v"(s205));
        const uint32_t bp40 = (uint32_t)bpitch - 40u;
        // uniform base + one 32-bit per-lane offset + immediate: a global_load with an SGPR base, no 64-bit address arithmetic
        const uint32_t o4 = (uint32_t)L.off + __umul24((uint32_t)(y - 18), (uint32_t)bpitch) + (uint32_t)(x - 18 + 4 * sub);
#pragma unroll
        for (int it = 0; it < 23; ++it) {
            const uint32_t row = (s205 + (uint32_t)(it * 16 * 205)) >> 11;
            const uint32_t o = o4 + __umul24(row, bp40);
            v[it] = *(orb_gptr32)((bbase + it * 64) + o);  // unaligned dword
        }
        {   // f = 368 + sub: only f = 368, 369 (row 36, columns 8, 9) exist; the other lanes re-read 369 and store nothing
            const uint32_t fl = 368u + (uint32_t)min(sub, 1);
            v[23] = *(orb_gptr32)(bbase + ((uint32_t)L.off + __umul24((uint32_t)(y + 18), (uint32_t)bpitch) + (uint32_t)(x - 18) + 4u * (fl - 360u)));
        }
    }
    __syncthreads();   // the tables (moment weights, pattern) are in LDS
    // ---- A: moments ----
    int m10 = 0, rs15 = 0, m01 = 0;
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = mr0 + 2 * i;
            if (row <= 30) {
                const uint2 wt = s_momw[row * 8 + mk];
                const uint32_t a = __builtin_amdgcn_udot4(w[i], wt.x, 0u, false);  // sum (u+15) * I
                const uint32_t s1 = __builtin_amdgcn_udot4(w[i], wt.y, 0u, false); // sum I
                m10 += (int)a;
                rs15 += (int)s1;
                m01 += __mul24(row - 15, (int)s1);  // |row - 15| <= 15, s1 <= 1020
            }
        }
        m10 -= 15 * rs15;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            m10 += __shfl_xor(m10, o, 64);
            m01 += __shfl_xor(m01, o, 64);
        }
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // ---- C: blurred patch -> LDS ----
    {
        uint8_t *pl = patch + 4 * sub;
#pragma unroll
        for (int it = 0; it < 23; ++it) *(uint32_t *)(pl + it * 64) = v[it];
        if (sub < 2) *(uint32_t *)(pl + 23 * 64) = v[23];
    }
    float a, bb;
    canon_sincos(angle, &a, &bb);
    __syncthreads();
    const uint8_t *pc = patch + 18 * DS_PP + 18;
    uint32_t bits = 0;
    // rotated sample positions (:100-106): row = cvRound(x*b + y*a), col = cvRound(x*a - y*b), every product and sum
    // rounded separately.  Two coordinates per packed-fp32 instruction; x*a - y*b == x*a + y*(-b) exactly.
    // cvRound by the 1.5 * 2^23 trick: the fp32 add rounds to the nearest integer, ties to even, and leaves it in
    // the low mantissa bits.
    typedef float orb_f2 __attribute__((ext_vector_type(2)));
    const orb_f2 ba = {bb, a}, anb = {a, -bb}, magic = {12582912.f, 12582912.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 pt = s_pat[i * 16 + sub];
        const orb_f2 p0 = orb_f2{pt.x, pt.x} * ba + orb_f2{pt.y, pt.y} * anb + magic;  // (row0, col0) + magic
        const orb_f2 p1 = orb_f2{pt.z, pt.z} * ba + orb_f2{pt.w, pt.w} * anb + magic;
        const int r0 = (int)(short)__float_as_int(p0.x), c0 = __float_as_int(p0.y) - 0x4B400000;
        const int r1 = (int)(short)__float_as_int(p1.x), c1 = __float_as_int(p1.y) - 0x4B400000;
        const int t0 = pc[r0 * DS_PP + c0], t1 = pc[r1 * DS_PP + c1];
        bits |= (uint32_t)(t0 < t1) << i;
    }
    if (live) {
        ((uint16_t *)dd)[sub] = (uint16_t)bits;
        if (sub == 0) {
            float fx = (float)x, fy = (float)y;
            if (level != 0) {  // pt *= mvScaleFactor[level] (:1104-1110)
                fx = __fmul_rn(fx, L.scale);
                fy = __fmul_rn(fy, L.scale);
            }
            kp->x = fx;
            kp->y = fy;
            kp->size = L.patch_size;
            kp->angle = angle;
            kp->response = (float)orb_key_r(key);
            kp->octave = level;
            kp->class_id = -1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K6  the padded pyramid of ONE frame, all levels in one launch: level l as a (w + 2E) x (h + 2E) block with the
// BORDER_REFLECT_101 frame copyMakeBorder gives it (src/ORBextractor.cc:1136-1142), rows tight (pitch w + 2E), levels back to
// back at pad_off[l] -- the memory shape of the reference's public mvImagePyramid, produced for one device-to-host copy.
// One thread per 4 output bytes.
// ---------------------------------------------------------------------------------------------------
struct PadArgs {
    const uint8_t *src[ORBFE_MAX_LEVELS];
    int32_t pitch[ORBFE_MAX_LEVELS], w[ORBFE_MAX_LEVELS], h[ORBFE_MAX_LEVELS];
    uint32_t off[ORBFE_MAX_LEVELS + 1];   // byte offset of a level's block in the output, off[nlevels] = total (multiples of 4)
    int32_t nlevels;
};
__global__ __launch_bounds__(256) void k_pad_pyramid(PadArgs a, uint8_t *__restrict__ out)
{
    const uint32_t o = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (o >= a.off[a.nlevels]) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < ORBFE_MAX_LEVELS; ++k)
        if (k < a.nlevels && o >= a.off[k]) l = k;
    const int W = a.w[l], H = a.h[l], PW = W + 2 * ORBFE_EDGE;
    const uint32_t rel = o - a.off[l], total = (uint32_t)PW * (uint32_t)(H + 2 * ORBFE_EDGE);
    uint32_t word = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t r = rel + (uint32_t)j;
        if (r < total) {   // a block's byte count need not be a multiple of 4: the slack up to the next block stays zero
            const int py = (int)(r / (uint32_t)PW), px = (int)(r - (uint32_t)py * (uint32_t)PW);
            const int sy = reflect101(py - ORBFE_EDGE, H), sx = reflect101(px - ORBFE_EDGE, W);
            word |= (uint32_t)a.src[l][(int64_t)sy * a.pitch[l] + sx] << (8 * j);
        }
    }
    *(uint32_t *)(out + o) = word;
}

hipError_t orbk_launch_pad_pyramid(const OrbPyrView &v, const uint32_t *off, uint8_t *d_out, hipStream_t st)
{
    PadArgs a;
    a.nlevels = v.nlevels;
    for (int l = 0; l < v.nlevels; ++l) {
        a.src[l] = v.ptr[l];
        a.pitch[l] = v.pitch[l];
        a.w[l] = v.w[l];
        a.h[l] = v.h[l];
        a.off[l] = off[l];
    }
    a.off[v.nlevels] = off[v.nlevels];
    const uint32_t nthreads = (off[v.nlevels] + 3u) / 4u;
    hipLaunchKernelGGL(k_pad_pyramid, dim3((nthreads + 255u) / 256u), dim3(256), 0, st, a, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------------------
static FrameSrc make_src(const OrbLaunch &a)
{
    FrameSrc fs;
    fs.l0 = a.d_gray;
    fs.l0_fstride = a.gray_fstride;
    fs.l0_pitch = a.gray_pitch;
    fs.pyr = a.d_pyr;
    fs.pyr_fstride = a.pyr_fstride;
    return fs;
}

size_t orbk_pyramid_lds_bytes(int dh) { return (size_t)(dh + 8) * sizeof(uint2) + (size_t)PW_ROWS * 256 * 4; }

#ifdef ORBFE_DEVELOPER
size_t orbk_pyramid2_lds_bytes(int gx, int gy) { return (size_t)(gy * PW_ROWS + 8) * sizeof(uint2) + (size_t)gy * PW_ROWS * gx * 4; }
#endif

static void pyr_args(const OrbLaunch &a, int l, PyrArgs &pa)
{
    const OrbLevel &D = a.h_plan->lv[l];
    const OrbLevel &S = a.h_plan->lv[l - 1];
    if (l == 1) {
        pa.src = a.d_gray;
        pa.src_fstride = a.gray_fstride;
        pa.spitch = a.gray_pitch;
    } else {
        pa.src = a.d_pyr + S.off;
        pa.src_fstride = a.pyr_fstride;
        pa.spitch = S.pitch;
    }
    pa.dst = a.d_pyr + D.off;
    pa.dst_fstride = a.pyr_fstride;
    pa.sw = S.w; pa.sh = S.h;
    pa.dw = D.w; pa.dh = D.h; pa.dpitch = D.pitch;
    pa.xtab = a.d_tabs + D.xtab;
    pa.ytab = a.d_tabs + D.ytab;
    // rows per lane run: PW_ROWS for batches; a call with a few frames is bound by the length of ONE lane's walk, so it takes
    // short runs and more lanes (ORBFE_PW_ROWS overrides, 2..PW_ROWS)
    int rows = a.nframes <= 8 ? 2 : PW_ROWS;
    if (a.opts.pw_rows >= 2 && a.opts.pw_rows <= PW_ROWS) rows = a.opts.pw_rows;   // ORBFE_OPT_PYR_ROWS
    const int nb = (D.h + rows - 1) / rows;
    pa.rb = (D.h + nb - 1) / nb;               // balanced run length
    pa.nrblk = (D.h + pa.rb - 1) / pa.rb;      // no empty run
}

hipError_t orbk_launch_pyramid(const OrbLaunch &a, hipStream_t st)
{
    // Level l reads level l-1 (:1134): one launch per level.  ORBFE_OPT_PYR_FUSE (developer builds) selects the two-levels-per-launch form instead
    // (k_pyr_walk2: level l from l-1 in tiles, level l+1 from the tile while it is in LDS -- the odd levels are not re-read
    // from HBM, an 8-level pyramid takes 4 launches).  It is byte-exact and tested, and it is SLOWER on this part: 0.574 vs
    // 0.505 ms per 1024 640x480 frames, 0.424 vs 0.397 ms per 128 1080p frames (profiles/r03_ab_experiments.json) -- the
    // chained kernels are latency-bound, not traffic-bound (3.7 of ~5 TB/s), and the second phase lengthens every
    // workgroup's dependent chain by more than the saved 0.37 GB of reads are worth.  Also measured, not kept: all levels
    // of a frame in one launch (a 1024-thread workgroup per frame, workgroup barriers between levels) -- 0.66 ms against
    // 0.63 ms per 1024 frames; with an agent-scope fence between the levels 4.8 ms.
    // (For a single frame the fused form does not win either: 49 us against 48 us for the seven chained launches.)
    const int nl = a.h_plan->nlevels;
    for (int l = 1; l < nl; ++l) {
        const OrbLevel &D = a.h_plan->lv[l];
        PyrArgs pa;
        pyr_args(a, l, pa);
#ifdef ORBFE_DEVELOPER
        if (a.opts.pyr_fuse == 1 && l + 1 < nl && D.p2_tx > 0) {
            const OrbLevel &C = a.h_plan->lv[l + 1];
            Pyr2Args p2;
            p2.ab = pa;
            p2.ab.rb = PW_ROWS;
            p2.dstC = a.d_pyr + C.off;
            p2.cw = C.w; p2.ch = C.h; p2.cpitch = C.pitch;
            p2.xtabC = a.d_tabs + C.xtab;
            p2.ytabC = a.d_tabs + C.ytab;
            p2.cxs = (const int32_t *)(a.d_tabs + D.p2_cxs);
            p2.cys = (const int32_t *)(a.d_tabs + D.p2_cys);
            p2.gx = D.p2_gx; p2.gy = D.p2_gy; p2.tiles_x = D.p2_tx; p2.tiles_y = D.p2_ty;
            hipLaunchKernelGGL(k_pyr_walk2, dim3(D.p2_tx * D.p2_ty, a.nframes), dim3(256), orbk_pyramid2_lds_bytes(D.p2_gx, D.p2_gy), st, p2);
            ++l;
            continue;
        }
#endif
        dim3 grid((pyr_walk_waves(D.w, pa.nrblk) + 3) / 4, a.nframes);
        hipLaunchKernelGGL(k_pyr_walk, grid, dim3(256), orbk_pyramid_lds_bytes(D.h), st, pa);
    }
    return hipGetLastError();
}

hipError_t orbk_launch_fast(const OrbLaunch &a, hipStream_t st)
{
    const FrameSrc fs = make_src(a);
    // the survivor counts and, right behind them, the cell flags of this call's frames: one clear
    if ((const char *)a.d_cflag != (const char *)a.d_scount + sizeof(int32_t) * (size_t)a.nframes * a.h_plan->nlevels * ORBFE_NK_STRIDE)
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(a.d_scount, 0, sizeof(int32_t) * (size_t)a.nframes * a.h_plan->nlevels * ORBFE_NK_STRIDE +
                                                     sizeof(uint32_t) * (size_t)a.nframes * a.h_plan->nlevels * a.cf_words, st);
    if (e != hipSuccess) return e;
    dim3 grid((a.h_plan->nfwaves + 3) / 4, a.nframes);
    if (a.fast_sparse == 2)
        hipLaunchKernelGGL(k_fast_map_c, dim3((a.h_plan->nfwaves_c + 3) / 4, a.nframes), dim3(256), (size_t)a.cf_words * 16, st, a.d_plan, fs,
                           a.d_flanes_c, a.h_plan->nfwaves_c, a.d_skeys, a.d_scount, a.d_cflag, a.cf_words, a.d_fstat);
    else if (a.h_plan->fast_cellrows) {
        if (a.fast_sparse)
            hipLaunchKernelGGL(k_fast_map_u<1>, grid, dim3(256), (size_t)a.cf_words * 16, st, a.d_plan, fs, a.d_flanes, a.h_plan->nfwaves, a.d_skeys,
                               a.d_scount, a.d_cflag, a.cf_words, a.d_fstat);
        else
            hipLaunchKernelGGL(k_fast_map_u<0>, grid, dim3(256), (size_t)a.cf_words * 16, st, a.d_plan, fs, a.d_flanes, a.h_plan->nfwaves, a.d_skeys,
                               a.d_scount, a.d_cflag, a.cf_words, (unsigned long long *)nullptr);
    } else if (a.fast_sparse)
        hipLaunchKernelGGL(k_fast_map<1>, grid, dim3(256), (size_t)a.cf_words * 16, st, a.d_plan, fs, a.d_flanes, a.h_plan->nfwaves, a.d_skeys,
                           a.d_scount, a.d_cflag, a.cf_words, a.d_fstat);
    else
        hipLaunchKernelGGL(k_fast_map<0>, grid, dim3(256), (size_t)a.cf_words * 16, st, a.d_plan, fs, a.d_flanes, a.h_plan->nfwaves, a.d_skeys,
                           a.d_scount, a.d_cflag, a.cf_words, (unsigned long long *)nullptr);
    return hipGetLastError();
}

#ifdef ORBFE_DEVELOPER
static hipError_t fast_clear(const OrbLaunch &a, hipStream_t st)
{
    // the survivor counts and, right behind them, the cell flags of this call's frames: one clear
    const int nl = a.h_plan->nlevels;
    if ((const char *)a.d_cflag != (const char *)a.d_scount + sizeof(int32_t) * (size_t)a.nframes * nl * ORBFE_NK_STRIDE)
        return hipErrorInvalidValue;
    return hipMemsetAsync(a.d_scount, 0, sizeof(int32_t) * (size_t)a.nframes * nl * ORBFE_NK_STRIDE +
                                             sizeof(uint32_t) * (size_t)a.nframes * nl * a.cf_words, st);
}

// one k_fast_pyr launch: FAST waves [w0, w1) of the lane list + the resize to level lpyr (0: none)
static void fast_pyr_one(const OrbLaunch &a, int w0, int w1, int lpyr, int spread, hipStream_t st)
{
    const FrameSrc fs = make_src(a);
    const OrbPlan &P = *a.h_plan;
    unsigned long long *fstat = a.fast_sparse ? a.d_fstat : (unsigned long long *)nullptr;
    PyrArgs pa;
    memset(&pa, 0, sizeof(pa));
    int npyr = 0;
    size_t lds = (size_t)a.cf_words * 16;
    if (lpyr > 0) {
        pyr_args(a, lpyr, pa);
        npyr = (((P.lv[lpyr].w + 3) / 4) * pa.nrblk + 255) / 256;
        lds = std::max(lds, orbk_pyramid_lds_bytes(P.lv[lpyr].h));
    }
    dim3 grid((w1 - w0 + 3) / 4 + npyr, a.nframes);
    if (grid.x == 0) return;
    if (a.fast_sparse)
        hipLaunchKernelGGL(k_fast_pyr<1>, grid, dim3(256), lds, st, a.d_plan, fs, a.d_flanes, w0, w1 - w0, a.d_skeys, a.d_scount,
                           a.d_cflag, a.cf_words, fstat, pa, npyr, spread);
    else
        hipLaunchKernelGGL(k_fast_pyr<0>, grid, dim3(256), lds, st, a.d_plan, fs, a.d_flanes, w0, w1 - w0, a.d_skeys, a.d_scount,
                           a.d_cflag, a.cf_words, fstat, pa, npyr, spread);
}

hipError_t orbk_launch_fast_pyr(const OrbLaunch &a, int nfused, int spread, hipStream_t st)
{
    const OrbPlan &P = *a.h_plan;
    const int nl = P.nlevels;
    hipError_t e = fast_clear(a, st);
    if (e != hipSuccess) return e;
    nfused = std::max(1, std::min(nfused, nl));
    for (int l = 0; l < nfused; ++l) fast_pyr_one(a, P.fwave_off[l], P.fwave_off[l + 1], l + 1 < nl ? l + 1 : 0, spread, st);
    if (nfused < nl) {
        // levels nfused + 1 .. nl - 1 as plain resize launches, then FAST over all remaining levels in one launch
        for (int l = nfused + 1; l < nl; ++l) {
            PyrArgs pa;
            pyr_args(a, l, pa);
            hipLaunchKernelGGL(k_pyr_walk, dim3((pyr_walk_waves(P.lv[l].w, pa.nrblk) + 3) / 4, a.nframes), dim3(256), orbk_pyramid_lds_bytes(P.lv[l].h), st, pa);
        }
        fast_pyr_one(a, P.fwave_off[nfused], P.nfwaves, 0, 0, st);
    }
    return hipGetLastError();
}

// FAST over the levels [l0, l1) of the lane list (no resize workgroups); clear = zero the survivor counts / cell flags first
hipError_t orbk_launch_fast_levels(const OrbLaunch &a, int l0, int l1, int clear, hipStream_t st)
{
    const OrbPlan &P = *a.h_plan;
    if (clear) {
        hipError_t e = fast_clear(a, st);
        if (e != hipSuccess) return e;
    }
    l0 = std::max(0, std::min(l0, (int)P.nlevels));
    l1 = std::max(l0, std::min(l1, (int)P.nlevels));
    fast_pyr_one(a, P.fwave_off[l0], P.fwave_off[l1], 0, 0, st);
    return hipGetLastError();
}
#endif  // ORBFE_DEVELOPER

hipError_t orbk_launch_octree(const OrbLaunch &a, hipStream_t st)
{
    // Level groups [0, nl/8), [nl/8, nl/2), [nl/2, nl) with 512 / 256 / 128 threads: the geometric feature split gives
    // level 0 about 3.6x the features (and many times the candidates) of level 7.  Measured per 1024 frames of 640x480 /
    // 1000 features: one launch of 512-thread workgroups 0.385 ms (dense-corner frames S) / 0.39 ms (camera-like frames
    // S_tum); this grouping 0.335 / 0.28 ms; {1,4,8} x 256 threads 0.33 / 0.315; four or more groups are slower again
    // (every launch has its own tail).
    const OrbPlan &P = *a.h_plan;
    const int nl = P.nlevels;
    // Small batches do not fill the chip: there the launches would only add their latencies (a workgroup's serial
    // bookkeeping, ~45 us each: single-frame host latency 0.26 -> 0.35 ms), so they take one launch.
    const bool grouped = a.nframes >= 128;
    const int cut[4] = {0, grouped ? std::max(1, nl / 8) : nl, grouped ? std::max(1, nl / 2) : nl, nl};
    int qts[3] = {512, 256, 128};
    for (int i = 0; i < 3; ++i)   // ORBFE_OPT_QT_THREADS_0..2: threads per workgroup of the three level groups
        if (a.opts.qt[i] >= 64 && a.opts.qt[i] <= QT_MAX && a.opts.qt[i] % 64 == 0) qts[i] = a.opts.qt[i];
    for (int gi = 0; gi < 3; ++gi) {
        const int l0 = cut[gi], l1 = std::min(cut[gi + 1], nl);
        if (l1 <= l0) continue;
        OctGroup g;
        g.level0 = l0;
        g.M = 64; g.nini = 1; g.tw = 2; g.th = 2; g.ncells = 1;
        for (int l = l0; l < l1; ++l) {
            const OrbLevel &L = P.lv[l];
            g.M = std::max(g.M, (int)orb_align_up(L.sel_cap + 1, 64));
            g.nini = std::max(g.nini, (int)L.nini);
            g.tw = std::max(g.tw, (int)L.w);
            g.th = std::max(g.th, (int)L.h);
            g.ncells = std::max(g.ncells, (int)L.ncells);
        }
        const size_t lds = orbk_octree_lds_bytes(g.M, g.nini, g.tw, g.th, g.ncells);
        const int qt = qts[gi];
        if (lds <= (size_t)ORBFE_LDS_MAX)
            hipLaunchKernelGGL(k_octree<false>, dim3(l1 - l0, a.nframes), dim3(qt), lds, st, a.d_plan, a.d_skeys, a.d_scount, a.d_cflag,
                               a.cf_words, a.d_knode, a.d_qtbox, a.qtbox_stride, (char *)nullptr, (int64_t)0, a.d_nkeys, a.d_sel, a.d_nsel,
                               a.d_ovf, g);
        else  // more nodes than the LDS holds: node arrays in the global scratch slice of each (frame, level)
            hipLaunchKernelGGL(k_octree<true>, dim3(l1 - l0, a.nframes), dim3(QT_MAX), octree_table_bytes(g.nini, g.tw, g.th, g.ncells), st,
                               a.d_plan, a.d_skeys, a.d_scount, a.d_cflag, a.cf_words, a.d_knode, a.d_qtbox, a.qtbox_stride, a.d_qtnodes,
                               a.qtnodes_stride, a.d_nkeys, a.d_sel, a.d_nsel, a.d_ovf, g);
    }
    return hipGetLastError();
}

hipError_t orbk_prepare_octree(int node_cap, int max_nini, int w, int h, int ncells)
{
    // The attribute is per kernel and process-wide: every handle sets it to the SAME value, the most the code can ever
    // request (the CU's 160 KB), so a later, smaller handle can never lower the limit under an earlier, larger one.
    (void)node_cap;
    if (octree_table_bytes(max_nini, w, h, ncells) > (size_t)ORBFE_LDS_MAX) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void *)k_octree<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ORBFE_LDS_MAX);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)k_octree<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ORBFE_LDS_MAX);
}

hipError_t orbk_launch_blur(const OrbLaunch &a, hipStream_t st)
{
    const FrameSrc fs = make_src(a);
    dim3 grid((a.h_plan->nbwaves + 3) / 4, a.nframes);
    if (a.h_plan->blur_rounding == 1)
        hipLaunchKernelGGL(k_blur7<1>, grid, dim3(256), 0, st, a.d_plan, fs, a.d_blanes, a.h_plan->nbwaves, a.d_blur,
                           a.pyr_fstride);
    else
        hipLaunchKernelGGL(k_blur7<0>, grid, dim3(256), 0, st, a.d_plan, fs, a.d_blanes, a.h_plan->nbwaves, a.d_blur,
                           a.pyr_fstride);
    return hipGetLastError();
}

#ifdef ORBFE_DEVELOPER
hipError_t orbk_launch_blur_pyr(const OrbLaunch &a, hipStream_t st)
{
    const FrameSrc fs = make_src(a);
    const int nl = a.h_plan->nlevels;
    for (int l = 0; l < nl; ++l) {
        const OrbLevel &S = a.h_plan->lv[l];
        BlurPyrArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.sw = S.w;
        pa.sh = S.h;
        pa.wave_lo = a.h_plan->bwave_off[l];
        size_t lds = 8;
        if (l + 1 < nl) {
            const OrbLevel &D = a.h_plan->lv[l + 1];
            pa.dst = a.d_pyr + D.off;
            pa.dst_fstride = a.pyr_fstride;
            pa.dpitch = D.pitch;
            pa.dh = D.h;
            pa.xtab = a.d_tabs + D.xtab;
            pa.ytab = a.d_tabs + D.ytab;
            lds = (size_t)(D.h + 8) * sizeof(uint2);
        }
        const int nw = a.h_plan->bwave_off[l + 1] - a.h_plan->bwave_off[l];
        dim3 grid((nw + 3) / 4, a.nframes);
        pa.split = a.h_plan->blur_split;
        const int br = a.h_plan->blur_rounding == 1;
        if (pa.split && br)
            hipLaunchKernelGGL((k_blur_pyr<1, 1>), grid, dim3(256), lds, st, a.d_plan, fs, a.d_blanes, a.d_blanesR, nw, a.d_blur, a.pyr_fstride, l, pa);
        else if (pa.split)
            hipLaunchKernelGGL((k_blur_pyr<0, 1>), grid, dim3(256), lds, st, a.d_plan, fs, a.d_blanes, a.d_blanesR, nw, a.d_blur, a.pyr_fstride, l, pa);
        else if (br)
            hipLaunchKernelGGL((k_blur_pyr<1, 0>), grid, dim3(256), lds, st, a.d_plan, fs, a.d_blanes, a.d_blanesR, nw, a.d_blur, a.pyr_fstride, l, pa);
        else
            hipLaunchKernelGGL((k_blur_pyr<0, 0>), grid, dim3(256), lds, st, a.d_plan, fs, a.d_blanes, a.d_blanesR, nw, a.d_blur, a.pyr_fstride, l, pa);
    }
    return hipGetLastError();
}
#endif  // ORBFE_DEVELOPER

hipError_t orbk_launch_describe(const OrbLaunch &a, hipStream_t st)
{
    const FrameSrc fs = make_src(a);
    const OrbPlan &P = *a.h_plan;
    DescArgs da;
    int wg = 0;
    for (int l = 0; l < ORBFE_MAX_LEVELS; ++l) {
        const OrbLevel &L = P.lv[l < P.nlevels ? l : 0];
        da.lv[l] = DescLevelArg{L.sel_off, L.off, L.pitch, wg, L.scale, L.patch_size, {0, 0}};
        da.wg0[l] = l < P.nlevels ? wg : 0x7FFFFFFF;
        if (l < P.nlevels) wg += (L.sel_cap + DS_KPW - 1) / DS_KPW;
    }
    da.nwg = wg;
    da.pad[0] = da.pad[1] = da.pad[2] = 0;
    // the workgroups of the levels; at least cap / DS_KPW of them: workgroup g also zero-fills the output slots total + 16 g ...
    dim3 grid(std::max(wg, (a.cap + DS_KPW - 1) / DS_KPW), a.nframes);
    hipLaunchKernelGGL(k_orient_describe, grid, dim3(DS_KPW * 16), 0, st, da, fs, a.d_blur, a.pyr_fstride, a.d_sel,
                       a.d_nsel, a.d_kps, a.d_desc, a.cap, a.d_n_out, P.nlevels, P.sel_per_frame, a.d_ovf);
    return hipGetLastError();
}
