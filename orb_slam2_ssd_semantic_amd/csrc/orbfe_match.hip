// orbfe_match.hip -- Hamming matcher kernels + C-ABI (include/orbfe.h "Matcher").
//
// Reference behaviour restated (paths relative to /root/reference):
//   DescriptorDistance                 src/ORBmatcher.cc:1968-1984
//   best / second-best update idiom    src/ORBmatcher.cc:280-289 (and :732-741)
//   SearchByBoW (KF,F) / (KF,KF)       src/ORBmatcher.cc:217-363 / :665-812
//   rotation histogram + prune         src/ORBmatcher.cc:308-316, :338-360
//   ComputeThreeMaxima                 src/ORBmatcher.cc:1912-1957
// Candidate-list matchers (SearchByBoW, CSR Hamming, stereo, BoW descent, distinctive descriptors): xor + v_bcnt_u32_b32.
// All-pairs brute force (M3): by default an EXACT int8 dot product on the matrix cores (k_match_bf); the xor / popcount
// all-pairs kernel the north star names is kept as k_match_popc (orbfe_matcher_set_bf_kernel), bit-identical, slower
// (profiles/r02_match_variants.json).  No CPU path.
#include <stddef.h>
#include <algorithm>
#include <new>

#include <vector>

#include "orbfe_common.h"
#include "orbfe_kernels.h"

// ---------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------
struct Desc8 {
    uint32_t w[8];
};

__device__ __forceinline__ int hamming8(const Desc8 &a, const uint32_t *__restrict__ b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += __popc(a.w[i] ^ b[i]);
    return d;
}

struct Best2 {
    int best, second, idx;
};

// merge of two partial results where `lo` covers the earlier iteration positions (first minimum wins)
__device__ __forceinline__ Best2 merge_best2(const Best2 &lo, const Best2 &hi)
{
    Best2 r;
    if (hi.best < lo.best) {
        r.best = hi.best;
        r.idx = hi.idx;
        r.second = min(lo.best, hi.second);
    } else {
        r.best = lo.best;
        r.idx = lo.idx;
        r.second = min(lo.second, hi.best);
    }
    return r;
}

// ORBmatcher.cc:308-313: rot = a1 - a2 (+360 if < 0); bin = round(rot * (1/HISTO_LENGTH)) (sic)
__device__ __forceinline__ int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / ORBFE_HISTO_LENGTH;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == ORBFE_HISTO_LENGTH) bin = 0;
    return bin;
}

// ---------------------------------------------------------------------------------------------------
// K8  brute force on the matrix cores.
//
// All-pairs Hamming is a GEMM in disguise: with every query bit mapped to the int8 value 2b-1 and every train bit
// to 64(2b-1),
//     dot(t', q') = 64 * (256 - 2 * hamming(q, t)) = 128 * (128 - d),
// exact in int32.  v_mfma_i32_32x32x32_i8 evaluates a 32 x 32 tile of such dots over 32 bit positions per
// instruction; 8 of them cover the 256 bits.  The accumulators do not start at zero but at a 7-bit "field"
// 32 - (train row inside the tile), so an accumulator IS the ranking key
//     key = 128 * (128 - d) + field                      (larger = closer, earlier row wins ties)
// and all that is left for the VALU is to keep, per query, the two largest keys: 2.5 instructions per distance
// instead of the 20 of an xor/popcount loop.  The best key gives (distance, lowest index) and the second key the
// second-smallest distance with multiplicity, which is exactly what the reference's sequential
//     if (d < best) {second = best; best = d; idx = j;} else if (d < second) second = d;
// produces.  Between train tiles the field of the running keys is forced to 127 (an older row beats any newer
// one on equal distance) and the global index of the best is latched whenever the best key changed in a tile.
//
// Workgroup = 4 waves x BM_Q query tiles of 32 = 512 queries; the train descriptors stream through LDS in tiles of
// 32 rows, expanded bit -> +-64 byte on the way in (8-byte table look-up per source byte) and laid out so that the
// A operand of lane (row r, k-half h) at k-step s is one conflict-free ds_read_b128 at ((2s + h) * 32 + r) * 16.
// The query tiles (B operands, 32 VGPRs each) are expanded once and stay in registers.  The bit -> k assignment is
// the same on both sides, which is all a dot product needs.
// The wave is software-pipelined: the MFMA chain of the next query tile (at the end of a train tile: of query tile 0
// of the next train tile) is issued between the ranking instructions of the current one.
// ---------------------------------------------------------------------------------------------------
#define BM_WAVES 4
#define BM_Q 4
#define BM_QW (BM_WAVES * BM_Q * 32)
#define BM_NEG (-(1 << 30))
#define BM_MAX_NT (1 << 22)  // the index is latched as a plain int; only nt * 32 bytes must be addressable in 32 bits

typedef int bm_v4i __attribute__((ext_vector_type(4)));
typedef int bm_v16i __attribute__((ext_vector_type(16)));

// bit b of v -> byte b: +mag if set, -mag if clear (two dwords for the 8 bits)
__device__ __forceinline__ uint2 bm_expand_byte(uint32_t v, uint32_t mag)
{
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        lo |= (((v >> b) & 1u) ? mag : (0x100u - mag)) << (8 * b);
        hi |= (((v >> (4 + b)) & 1u) ? mag : (0x100u - mag)) << (8 * b);
    }
    return make_uint2(lo, hi);
}

__global__ __launch_bounds__(BM_WAVES * 64, 2) void k_match_bf(const uint8_t *__restrict__ q_base,
                                                               const uint8_t *__restrict__ t_base,
                                                               const int32_t *__restrict__ n_arr,  // per-frame counts or NULL
                                                               const int32_t *__restrict__ tn_arr, // counts of the train block
                                                               const int32_t *__restrict__ qframe,
                                                               const int32_t *__restrict__ tframe, int cap, int nq_s,
                                                               int nt_s, float nnratio, int th,
                                                               int32_t *__restrict__ match, int32_t *__restrict__ best_o,
                                                               int32_t *__restrict__ second_o)
{
    __shared__ uint2 s_tabq[256], s_tabt[256];  // query bits -> +-1, train bits -> +-64
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[2][32 * 256];
    const int pair = blockIdx.y;
    const uint8_t *q = q_base, *t = t_base;
    int nq = nq_s, nt = nt_s;
    int64_t out0 = 0;
    if (n_arr) {  // batched-frames form
        const int qf = qframe[pair], tf = tframe[pair];
        q = q_base + (int64_t)qf * cap * 32;
        t = t_base + (int64_t)tf * cap * 32;
        nq = min(n_arr[qf], cap);
        nt = min(tn_arr[tf], cap);
        out0 = (int64_t)pair * cap;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.x * BM_QW;
    const int nslots = n_arr ? cap : nq;  // output slots of this pair
    if (q0 >= nslots) return;
    if (q0 >= nq) {  // only padding slots of the batched form: no match, no compute
        for (int i = q0 + tid; i < min(q0 + BM_QW, nslots); i += BM_WAVES * 64) {
            match[out0 + i] = -1;
            if (best_o) best_o[out0 + i] = 256;
            if (second_o) second_o[out0 + i] = 256;
        }
        return;
    }
    s_tabq[tid] = bm_expand_byte((uint32_t)tid, 1u);
    s_tabt[tid] = bm_expand_byte((uint32_t)tid, 64u);
    __syncthreads();

    const int c = lane & 31, h = lane >> 5;
    // ---- B operands: BM_Q query tiles, lane (c, h) holds bits [128h, 128h + 128) of query c as 8 x 16 bytes ----
    bm_v4i B[BM_Q][8];
#pragma unroll
    for (int u = 0; u < BM_Q; ++u) {
        const int qi = q0 + (wid * BM_Q + u) * 32 + c;
        const uint4 raw = *(const uint4 *)(q + (int64_t)min(qi, nq - 1) * 32 + h * 16);
        const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int sstep = 0; sstep < 8; ++sstep) {
            const uint32_t b0 = (rw[sstep >> 1] >> (16 * (sstep & 1))) & 0xFFu;
            const uint32_t b1 = (rw[sstep >> 1] >> (16 * (sstep & 1) + 8)) & 0xFFu;
            const uint2 e0 = s_tabq[b0], e1 = s_tabq[b1];
            B[u][sstep] = bm_v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
        }
    }
    int kb[BM_Q], ks[BM_Q], bi[BM_Q];
#pragma unroll
    for (int u = 0; u < BM_Q; ++u) {
        kb[u] = ks[u] = BM_NEG;
        bi[u] = -1;
    }
    if (nt > 0) {
        // ---- train tile staging: thread (r = tid & 31, w = tid >> 5) expands source dword w of row r ----
        const int sr = tid & 31, sw = tid >> 5;
        const uint32_t *t32 = (const uint32_t *)t;
        // rows past the end re-read the last row: their keys are masked in the last tile
        auto stage_load = [&](int T) -> uint32_t { return t32[(uint32_t)(min(T + sr, nt - 1) * 8 + sw)]; };
        auto stage_store = [&](uint32_t dw, int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int beta = 4 * sw + i, hh = beta >> 4, bp = beta & 15;
                const int off = (((bp >> 1) * 2 + hh) * 32 + sr) * 16 + (bp & 1) * 8;
                *(uint2 *)(s_tile[buf] + off) = s_tabt[(dw >> (8 * i)) & 0xFFu];
            }
        };
        auto load_a = [&](bm_v4i (&A)[8], int buf) {
#pragma unroll
            for (int sstep = 0; sstep < 8; ++sstep)
                A[sstep] = *(const bm_v4i *)(s_tile[buf] + ((sstep * 2 + h) * 32 + c) * 16);
        };
        // accumulator start values = key fields: 32 - (train row of the accumulator inside the tile)
        bm_v16i cfull;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) cfull[reg] = 32 - ((reg & 3) + 8 * (reg >> 2) + 4 * h);
        auto chain = [&](const bm_v4i (&A)[8], const bm_v4i (&Bu)[8]) -> bm_v16i {
            bm_v16i acc = cfull;
#pragma unroll
            for (int sstep = 0; sstep < 8; ++sstep) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[sstep], Bu[sstep], acc, 0, 0, 0);
            return acc;
        };
        auto interleave = [&]() {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // five VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // two largest keys of query tile u over the 16 rows in `acc`; latch the index when the best changed.
        // A tournament instead of a running (best, second) pair: same instruction count, but depth 5 instead of 16 --
        // with two waves per SIMD a 16-long dependent chain leaves the VALU idle most of the time.
        auto rank = [&](const bm_v16i &acc, int u, int T, bool mask_tail) {
            int hi[8], lo[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                int k0 = acc[2 * p], k1 = acc[2 * p + 1];
                if (mask_tail) {
                    if (T + ((2 * p) & 3) + 8 * ((2 * p) >> 2) + 4 * h >= nt) k0 = BM_NEG;
                    if (T + ((2 * p + 1) & 3) + 8 * ((2 * p + 1) >> 2) + 4 * h >= nt) k1 = BM_NEG;
                }
                hi[p] = max(k0, k1);
                lo[p] = min(k0, k1);
            }
#pragma unroll
            for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                for (int p = 0; p < w; ++p) {  // merge the (largest, second) pairs p and p + w
                    const int h2 = max(hi[p], hi[p + w]);
                    lo[p] = max(min(hi[p], hi[p + w]), max(lo[p], lo[p + w]));
                    hi[p] = h2;
                }
            const int bprev = kb[u] | 127, sprev = ks[u] | 127;
            const int b = max(bprev, hi[0]);
            const int s2 = max(min(bprev, hi[0]), max(sprev, lo[0]));
            if (b != bprev) bi[u] = T + 32 - (b & 127);
            kb[u] = b;
            ks[u] = s2;
        };

        stage_store(stage_load(0), 0);
        uint32_t g1 = stage_load(32);  // source dword of the tile after next, fetched two tiles before it is consumed
        __syncthreads();
        bm_v4i A[8];
        load_a(A, 0);
        bm_v16i acc = chain(A, B[0]);
        int buf = 0, T = 0;
        for (; T + 32 < nt; T += 32) {  // all tiles but the last one
            const uint32_t g2 = stage_load(T + 64);
            stage_store(g1, buf ^ 1);  // tile T + 32 (not read before the barrier below)
            g1 = g2;
#pragma unroll
            for (int u = 0; u < BM_Q; ++u) {
                bm_v16i nacc;
                if (u + 1 < BM_Q) {
                    nacc = chain(A, B[u + 1]);
                } else {  // next train tile
                    __syncthreads();
                    buf ^= 1;
                    load_a(A, buf);
                    nacc = chain(A, B[0]);
                }
                rank(acc, u, T, false);
                interleave();
                acc = nacc;
            }
        }
#pragma unroll
        for (int u = 0; u < BM_Q; ++u) {  // last tile: rows past the end of the train set are masked
            bm_v16i nacc = acc;
            if (u + 1 < BM_Q) nacc = chain(A, B[u + 1]);
            rank(acc, u, T, true);
            acc = nacc;
        }
    }
    // ---- lanes c and c + 32 hold different train rows of the same query: merge the two halves, decode ----
#pragma unroll
    for (int u = 0; u < BM_Q; ++u) {
        const int ob = __shfl_xor(kb[u], 32, 64), os = __shfl_xor(ks[u], 32, 64), oi = __shfl_xor(bi[u], 32, 64);
        // distance part of a key: m = 128 - d (sentinels stay far below)
        const int mb = kb[u] >> 7, mo = ob >> 7;
        const bool take_o = mo > mb || (mo == mb && oi >= 0 && (bi[u] < 0 || oi < bi[u]));
        const int m1 = take_o ? mo : mb, idx = take_o ? oi : bi[u];
        const int m2 = max(min(mb, mo), max(ks[u] >> 7, os >> 7));
        const int qi = q0 + (wid * BM_Q + u) * 32 + c;
        if (h == 0 && qi < nslots) {
            const int best = idx >= 0 ? 128 - m1 : 256;
            const int second = m2 > (BM_NEG >> 8) ? 128 - m2 : 256;
            const bool valid = qi < nq;
            int m = -1;
            if (valid && idx >= 0 && best <= th && (float)best < __fmul_rn(nnratio, (float)second)) m = idx;
            match[out0 + qi] = m;
            if (best_o) best_o[out0 + qi] = valid ? best : 256;
            if (second_o) second_o[out0 + qi] = valid ? second : 256;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K8b  brute force by xor / popcount (the all-pairs formulation BASELINE.json's north star names; kept for the A/B).
// One lane per query (descriptor in 8 VGPRs), the train rows stream through LDS in tiles of 128 and are read as
// broadcasts; per distance 8 v_xor + 8 v_bcnt_u32_b32 (accumulating) + 4 ranking ops on the unique key
// (distance << 22 | train index): b2 = min(b2, max(b1, k)); b1 = min(b1, k) keeps the two smallest keys, i.e. best
// distance with the lowest index and the second-smallest distance with multiplicity -- the reference's update idiom.
// Same arguments, same outputs as k_match_bf.
// ---------------------------------------------------------------------------------------------------
#define BP_TILE 128
__global__ __launch_bounds__(256) void k_match_popc(const uint8_t *__restrict__ q_base, const uint8_t *__restrict__ t_base,
                                                    const int32_t *__restrict__ n_arr, const int32_t *__restrict__ tn_arr,
                                                    const int32_t *__restrict__ qframe,
                                                    const int32_t *__restrict__ tframe, int cap, int nq_s, int nt_s,
                                                    float nnratio, int th, int32_t *__restrict__ match,
                                                    int32_t *__restrict__ best_o, int32_t *__restrict__ second_o)
{
    __shared__ uint4 s_t[BP_TILE][2];
    const int pair = blockIdx.y;
    const uint8_t *q = q_base, *t = t_base;
    int nq = nq_s, nt = nt_s;
    int64_t out0 = 0;
    if (n_arr) {
        const int qf = qframe[pair], tf = tframe[pair];
        q = q_base + (int64_t)qf * cap * 32;
        t = t_base + (int64_t)tf * cap * 32;
        nq = min(n_arr[qf], cap);
        nt = min(tn_arr[tf], cap);
        out0 = (int64_t)pair * cap;
    }
    const int tid = threadIdx.x;
    const int qi = blockIdx.x * 256 + tid;
    const int nslots = n_arr ? cap : nq;
    if ((int)blockIdx.x * 256 >= nslots) return;
    uint32_t qw[8];
    {
        const uint4 *pq = (const uint4 *)(q + (int64_t)min(qi, max(nq - 1, 0)) * 32);
        const uint4 a = nq > 0 ? pq[0] : make_uint4(0, 0, 0, 0), b = nq > 0 ? pq[1] : make_uint4(0, 0, 0, 0);
        qw[0] = a.x; qw[1] = a.y; qw[2] = a.z; qw[3] = a.w; qw[4] = b.x; qw[5] = b.y; qw[6] = b.z; qw[7] = b.w;
    }
    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    for (int T = 0; T < nt; T += BP_TILE) {
        __syncthreads();
        {
            const int r = tid >> 1, hf = tid & 1;   // 256 threads stage 128 rows x 2 halves
            s_t[r][hf] = ((const uint4 *)(t + (int64_t)min(T + r, nt - 1) * 32))[hf];
        }
        __syncthreads();
        const int rows = min(BP_TILE, nt - T);
        for (int r = 0; r < rows; ++r) {
            const uint4 a = s_t[r][0], b = s_t[r][1];
            uint32_t d = __popc(qw[0] ^ a.x);
            d += __popc(qw[1] ^ a.y);
            d += __popc(qw[2] ^ a.z);
            d += __popc(qw[3] ^ a.w);
            d += __popc(qw[4] ^ b.x);
            d += __popc(qw[5] ^ b.y);
            d += __popc(qw[6] ^ b.z);
            d += __popc(qw[7] ^ b.w);
            const uint32_t k = (d << 22) | (uint32_t)(T + r);
            k2 = min(k2, max(k1, k));
            k1 = min(k1, k);
        }
    }
    if (qi < nslots) {
        const bool valid = qi < nq;
        const int best = k1 != 0xFFFFFFFFu ? (int)(k1 >> 22) : 256;
        const int second = k2 != 0xFFFFFFFFu ? (int)(k2 >> 22) : 256;
        const int idx = k1 != 0xFFFFFFFFu ? (int)(k1 & 0x3FFFFFu) : -1;
        int m = -1;
        if (valid && idx >= 0 && best <= th && (float)best < __fmul_rn(nnratio, (float)second)) m = idx;
        match[out0 + qi] = m;
        if (best_o) best_o[out0 + qi] = valid ? best : 256;
        if (second_o) second_o[out0 + qi] = valid ? second : 256;
    }
}

// ---------------------------------------------------------------------------------------------------
// K10  rotation-consistency histogram over accepted matches, ComputeThreeMaxima, prune, count.
// One workgroup per pair.  key i is kept when its bin is one of the three maxima.
// angles: element i of side X is at X_ang[i * stride] (stride 1 for plain arrays, 7 for orbfe_keypoint).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rot_prune(int32_t *__restrict__ match, const float *__restrict__ a_ang,
                                                   const float *__restrict__ b_ang, int ang_stride,
                                                   const int32_t *__restrict__ n_arr,
                                                   const int32_t *__restrict__ aframe,
                                                   const int32_t *__restrict__ bframe, int cap, int n_s,
                                                   int check_ori, int32_t *__restrict__ nmatches)
{
    __shared__ int s_hist[ORBFE_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count;
    const int pair = blockIdx.x, tid = threadIdx.x;
    int n = n_s;
    int32_t *mt = match;
    const float *aa = a_ang, *ba = b_ang;
    if (n_arr) {
        const int af = aframe[pair], bf = bframe[pair];
        n = min(n_arr[af], cap);
        mt = match + (int64_t)pair * cap;
        aa = a_ang + (int64_t)af * cap * ang_stride;
        ba = b_ang + (int64_t)bf * cap * ang_stride;
    }
    if (tid < ORBFE_HISTO_LENGTH) s_hist[tid] = 0;
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (check_ori) {
        for (int i = tid; i < n; i += 256) {
            const int j = mt[i];
            if (j >= 0) atomicAdd(&s_hist[rot_bin(aa[(int64_t)i * ang_stride], ba[(int64_t)j * ang_stride])], 1);
        }
        __syncthreads();
        if (tid == 0) {  // ComputeThreeMaxima (:1912-1957)
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < ORBFE_HISTO_LENGTH; ++i) {
                const int s = s_hist[i];
                if (s > max1) {
                    max3 = max2; max2 = max1; max1 = s;
                    i3 = i2; i2 = i1; i1 = i;
                } else if (s > max2) {
                    max3 = max2; max2 = s;
                    i3 = i2; i2 = i;
                } else if (s > max3) {
                    max3 = s; i3 = i;
                }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
            s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
        }
        __syncthreads();
    }
    int local = 0;
    for (int i = tid; i < n; i += 256) {
        const int j = mt[i];
        if (j < 0) continue;
        if (check_ori) {
            const int bin = rot_bin(aa[(int64_t)i * ang_stride], ba[(int64_t)j * ang_stride]);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                mt[i] = -1;
                continue;
            }
        }
        ++local;
    }
    atomicAdd(&s_count, local);
    __syncthreads();
    if (tid == 0) nmatches[pair] = s_count;
}

// ---------------------------------------------------------------------------------------------------
// K9  SearchByBoW: one thread per KeyFrame vocabulary node.  Nodes own disjoint feature sets, so the
// greedy "F feature already claimed" rule (:273-274, :725) only couples features inside one node and
// is replayed serially there, in the reference's iteration order.
// matchF2KF[iF] = KF feature index, -1 none.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_search_by_bow(const uint8_t *__restrict__ descKF,
                                                      const uint8_t *__restrict__ validKF,
                                                      const uint32_t *__restrict__ nodeKF,
                                                      const uint32_t *__restrict__ offKF,
                                                      const uint32_t *__restrict__ idxKF, int nnodesKF,
                                                      const uint8_t *__restrict__ descF,
                                                      const uint8_t *__restrict__ validF,
                                                      const uint32_t *__restrict__ nodeF,
                                                      const uint32_t *__restrict__ offF,
                                                      const uint32_t *__restrict__ idxF, int nnodesF, float nnratio,
                                                      int th_low, int strict_lt, int32_t *__restrict__ matchF2KF)
{
    const int a = blockIdx.x * 64 + threadIdx.x;
    if (a >= nnodesKF) return;
    const uint32_t node = nodeKF[a];
    int lo = 0, hi = nnodesF - 1, b = -1;  // lower_bound walk of :329-333 == binary search on sorted ids
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t v = nodeF[mid];
        if (v == node) { b = mid; break; }
        if (v < node) lo = mid + 1; else hi = mid - 1;
    }
    if (b < 0) return;
    for (uint32_t ik = offKF[a]; ik < offKF[a + 1]; ++ik) {
        const uint32_t rk = idxKF[ik];
        if (validKF && !validKF[rk]) continue;
        Desc8 dk;
        const uint32_t *pk = (const uint32_t *)(descKF + (int64_t)rk * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) dk.w[i] = pk[i];
        int b1 = 256, b2 = 256, bi = -1;
        for (uint32_t jf = offF[b]; jf < offF[b + 1]; ++jf) {
            const uint32_t rf = idxF[jf];
            if (matchF2KF[rf] >= 0) continue;
            if (validF && !validF[rf]) continue;
            const int d = hamming8(dk, (const uint32_t *)(descF + (int64_t)rf * 32));
            if (d < b1) { b2 = b1; b1 = d; bi = (int)rf; }
            else if (d < b2) { b2 = d; }
        }
        const bool pass = strict_lt ? (b1 < th_low) : (b1 <= th_low);
        if (pass && bi >= 0 && (float)b1 < __fmul_rn(nnratio, (float)b2)) matchF2KF[bi] = (int32_t)rk;
    }
}

// rotation prune for SearchByBoW: key = F feature i, rot = angKF[match[i]] - angF[i] (:308, :759)
__global__ __launch_bounds__(256) void k_rot_prune_bow(int32_t *__restrict__ match, const float *__restrict__ angKF,
                                                       const float *__restrict__ angF, int nF, int check_ori,
                                                       int32_t *__restrict__ nmatches)
{
    __shared__ int s_hist[ORBFE_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count;
    const int tid = threadIdx.x;
    if (tid < ORBFE_HISTO_LENGTH) s_hist[tid] = 0;
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (check_ori) {
        for (int i = tid; i < nF; i += 256) {
            const int j = match[i];
            if (j >= 0) atomicAdd(&s_hist[rot_bin(angKF[j], angF[i])], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < ORBFE_HISTO_LENGTH; ++i) {
                const int s = s_hist[i];
                if (s > max1) {
                    max3 = max2; max2 = max1; max1 = s;
                    i3 = i2; i2 = i1; i1 = i;
                } else if (s > max2) {
                    max3 = max2; max2 = s;
                    i3 = i2; i2 = i;
                } else if (s > max3) {
                    max3 = s; i3 = i;
                }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
            s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
        }
        __syncthreads();
    }
    int local = 0;
    for (int i = tid; i < nF; i += 256) {
        const int j = match[i];
        if (j < 0) continue;
        if (check_ori) {
            const int bin = rot_bin(angKF[j], angF[i]);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                match[i] = -1;
                continue;
            }
        }
        ++local;
    }
    atomicAdd(&s_count, local);
    __syncthreads();
    if (tid == 0) nmatches[0] = s_count;
}

// 8(f).1: best / second-best over a per-query candidate list
__global__ __launch_bounds__(256) void k_hamming_csr(const uint8_t *__restrict__ q, int nq,
                                                     const uint8_t *__restrict__ t, const uint32_t *__restrict__ off,
                                                     const uint32_t *__restrict__ cand, int32_t *__restrict__ best_idx,
                                                     int32_t *__restrict__ best, int32_t *__restrict__ second,
                                                     int32_t *__restrict__ second_idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    Desc8 dq;
    const uint32_t *p = (const uint32_t *)(q + (int64_t)i * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) dq.w[k] = p[k];
    // the reference's update idiom (:128-140) with the bookkeeping it attaches to the runner-up (bestLevel2 = the level of
    // whichever candidate last set bestDist2): si = that candidate
    int b1 = 256, b2 = 256, bi = -1, si = -1;
    for (uint32_t j = off[i]; j < off[i + 1]; ++j) {
        const uint32_t c = cand[j];
        const int d = hamming8(dq, (const uint32_t *)(t + (int64_t)c * 32));
        if (d < b1) { b2 = b1; si = bi; b1 = d; bi = (int)c; }
        else if (d < b2) { b2 = d; si = (int)c; }
    }
    best_idx[i] = bi;
    best[i] = b1;
    second[i] = b2;
    if (second_idx) second_idx[i] = si;
}

// every distance of a candidate list, for callers whose acceptance rule needs more than the two smallest
// (SearchForInitialization, src/ORBmatcher.cc:571-574: a candidate is skipped when an earlier query already holds it at a
// smaller distance).  One wave per query, lanes over its candidates.
__global__ __launch_bounds__(256) void k_hamming_csr_all(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                                         const uint32_t *__restrict__ off, const uint32_t *__restrict__ cand,
                                                         uint16_t *__restrict__ dist)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq) return;
    const int lane = threadIdx.x & 63;
    Desc8 dq;
    const uint32_t *p = (const uint32_t *)(q + (int64_t)i * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) dq.w[k] = p[k];
    const uint32_t e = off[i + 1];
    for (uint32_t j = off[i] + (uint32_t)lane; j < e; j += 64u)
        dist[j] = (uint16_t)hamming8(dq, (const uint32_t *)(t + (int64_t)cand[j] * 32));
}

// ---------------------------------------------------------------------------------------------------
// host API
// ---------------------------------------------------------------------------------------------------
struct MDevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need == 0) need = 4;
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
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

struct MPinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need == 0) need = 4;
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        need = (need * 3 / 2 + 4095) & ~(size_t)4095;
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

struct orbfe_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    MDevBuf b[16];
    bool own_stream = true;  // false: the stream belongs to a pipeline (orbfe_internal_matcher_create_on_stream)
    int bf_kernel = 0;  // 0 = k_match_bf (int8 dot product on the matrix cores), 1 = k_match_popc (xor / popcount)
    // Device entry points that use the scratch blocks b[] run on the CALLER's stream: one that arrives on another stream
    // than its predecessor waits (at stream level) for the event recorded behind that predecessor's last launch.
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    hipEvent_t ev_scratch = nullptr;
    MPinBuf pin_in, pin_out;  // page-locked staging of the latency-bound per-frame calls (orbfe_search_by_projection)
    bool proj_fused = true;   // orbfe_search_by_projection: the one-launch form (k_proj_fused); false = the four-kernel path (tests)
    MDevBuf proj_done;        // k_proj_fused's arrival counter / overflow word (zero between calls)
};

static hipError_t scratch_acquire(orbfe_matcher *m, hipStream_t st)
{
    if (m->scratch_used && m->scratch_stream != st) return hipStreamWaitEvent(st, m->ev_scratch, 0);
    return hipSuccess;
}
static hipError_t scratch_release(orbfe_matcher *m, hipStream_t st)
{
    m->scratch_stream = st;
    m->scratch_used = true;
    return hipEventRecord(m->ev_scratch, st);
}

struct MDeviceGuard {
    int prev = -1, dev = -1;
    explicit MDeviceGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~MDeviceGuard()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

extern "C" int32_t orbfe_hamming(const uint8_t a[32], const uint8_t b[32])
{
    // src/ORBmatcher.cc:1968-1984 (the SWAR popcount there equals a hardware popcount)
    int d = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t x, y;
        memcpy(&x, a + 4 * i, 4);
        memcpy(&y, b + 4 * i, 4);
        d += __builtin_popcount(x ^ y);
    }
    return d;
}

static orbfe_status matcher_create_impl(int32_t device, hipStream_t borrowed, bool borrow, orbfe_matcher **out);
extern "C" orbfe_status orbfe_matcher_create(int32_t device, orbfe_matcher **out) { return matcher_create_impl(device, nullptr, false, out); }
// a matcher for a pipe of orbfe_pipeline: `st` (the pipe's stream) is its own stream and stays the pipeline's
orbfe_status orbfe_internal_matcher_create_on_stream(int32_t device, void *st, orbfe_matcher **out)
{
    return matcher_create_impl(device, (hipStream_t)st, true, out);
}

static orbfe_status matcher_create_impl(int32_t device, hipStream_t borrowed, bool borrow, orbfe_matcher **out)
{
    if (!out) return ORBFE_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        orbfe_set_error("no HIP device visible; liborbfe has no CPU fallback");
        return ORBFE_ERR_NODEVICE;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) { orbfe_set_error("device out of range"); return ORBFE_ERR_ARG; }
    orbfe_matcher *m = new (std::nothrow) orbfe_matcher();
    if (!m) return ORBFE_ERR_NOMEM;
    m->device = device;
    MDeviceGuard g(device);
    m->own_stream = !borrow;
    if (borrow) m->stream = borrowed;
    if ((!borrow && hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&m->ev_scratch, hipEventDisableTiming) != hipSuccess) {
        orbfe_set_error("hipStreamCreate / hipEventCreate failed");
        if (m->stream && m->own_stream) (void)hipStreamDestroy(m->stream);
        delete m;
        return ORBFE_ERR_HIP;
    }
    *out = m;
    return ORBFE_OK;
}

extern "C" void *orbfe_matcher_get_stream(orbfe_matcher *m) { return m ? (void *)m->stream : nullptr; }

extern "C" orbfe_status orbfe_matcher_set_bf_kernel(orbfe_matcher *m, int32_t kernel)
{
    if (!m || kernel < 0 || kernel > 1) return ORBFE_ERR_ARG;
    m->bf_kernel = kernel;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_matcher_set_projection_kernel(orbfe_matcher *m, int32_t kernel)
{
    if (!m || kernel < 0 || kernel > 1) return ORBFE_ERR_ARG;
    m->proj_fused = kernel == 0;
    return ORBFE_OK;
}

extern "C" void orbfe_matcher_destroy(orbfe_matcher *m)
{
    if (!m) return;
    MDeviceGuard g(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->scratch_used) (void)hipEventSynchronize(m->ev_scratch);
    for (auto &b : m->b) b.release();
    m->proj_done.release();
    m->pin_in.release();
    m->pin_out.release();
    if (m->ev_scratch) (void)hipEventDestroy(m->ev_scratch);
    if (m->stream && m->own_stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

static orbfe_status launch_bf(int kernel, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, const float *d_qa,
                              const float *d_ta, int ang_stride, float nnratio, int th, int check_ori,
                              int32_t *d_match, int32_t *d_best, int32_t *d_second, int32_t *d_nm, hipStream_t st)
{
    if (nq > 0) {
        if (kernel == 1)
            hipLaunchKernelGGL(k_match_popc, dim3((nq + 255) / 256, 1), dim3(256), 0, st, d_q, d_t, (const int32_t *)nullptr,
                               (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 0, nq, nt, nnratio, th,
                               d_match, d_best, d_second);
        else
            hipLaunchKernelGGL(k_match_bf, dim3((nq + BM_QW - 1) / BM_QW, 1), dim3(BM_WAVES * 64), 0, st, d_q, d_t,
                               (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                               (const int32_t *)nullptr, 0, nq, nt, nnratio, th, d_match, d_best, d_second);
        ORBFE_HIP(hipGetLastError());
    }
    const int ori = (check_ori && d_qa && d_ta) ? 1 : 0;
    hipLaunchKernelGGL(k_rot_prune, dim3(1), dim3(256), 0, st, d_match, d_qa, d_ta, ang_stride,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 0, nq, ori, d_nm);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_match_bf_device(orbfe_matcher *m, const uint8_t *d_q, int32_t nq, const uint8_t *d_t,
                                              int32_t nt, const float *d_q_angle, const float *d_t_angle,
                                              float nnratio, int32_t th, int32_t check_ori, int32_t *d_match_q2t,
                                              int32_t *d_best, int32_t *d_second, int32_t *d_nmatches, void *stream)
{
    if (!m || nq < 0 || nt < 0 || nt > BM_MAX_NT || !d_match_q2t || !d_nmatches || (nq > 0 && !d_q) || (nt > 0 && !d_t)) {
        orbfe_set_error("bad argument to orbfe_match_bf_device (train set limited to %d descriptors)", BM_MAX_NT);
        return ORBFE_ERR_ARG;
    }
    MDeviceGuard g(m->device);
    return launch_bf(m->bf_kernel, d_q, nq, d_t, nt, d_q_angle, d_t_angle, 1, nnratio, th, check_ori, d_match_q2t, d_best, d_second,
                     d_nmatches, (hipStream_t)stream);
}

extern "C" orbfe_status orbfe_match_bf(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                                       const float *q_angle, const float *t_angle, float nnratio, int32_t th,
                                       int32_t check_ori, int32_t *match_q2t, int32_t *best, int32_t *second,
                                       int32_t *nmatches)
{
    if (!m || nq < 0 || nt < 0 || nt > BM_MAX_NT || (nq > 0 && (!q || !match_q2t)) || (nt > 0 && !t)) {
        orbfe_set_error("bad argument to orbfe_match_bf (train set limited to %d descriptors)", BM_MAX_NT);
        return ORBFE_ERR_ARG;
    }
    if (nq == 0) {
        if (nmatches) *nmatches = 0;
        return ORBFE_OK;
    }
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    const bool ori = check_ori && q_angle && t_angle;
    ORBFE_HIP(m->b[0].ensure((size_t)nq * 32));
    ORBFE_HIP(m->b[1].ensure((size_t)nt * 32));
    ORBFE_HIP(m->b[2].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[3].ensure((size_t)nt * 4));
    ORBFE_HIP(m->b[4].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[5].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[6].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[7].ensure(4));
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nt > 0) ORBFE_HIP(hipMemcpyAsync(m->b[1].p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    if (ori) {
        ORBFE_HIP(hipMemcpyAsync(m->b[2].p, q_angle, (size_t)nq * 4, hipMemcpyHostToDevice, st));
        if (nt > 0) ORBFE_HIP(hipMemcpyAsync(m->b[3].p, t_angle, (size_t)nt * 4, hipMemcpyHostToDevice, st));
    }
    orbfe_status s = launch_bf(m->bf_kernel, (const uint8_t *)m->b[0].p, nq, (const uint8_t *)m->b[1].p, nt,
                               ori ? (const float *)m->b[2].p : nullptr, ori ? (const float *)m->b[3].p : nullptr, 1,
                               nnratio, th, ori ? 1 : 0, (int32_t *)m->b[4].p, (int32_t *)m->b[5].p,
                               (int32_t *)m->b[6].p, (int32_t *)m->b[7].p, st);
    if (s != ORBFE_OK) return s;
    int32_t nm = 0;
    ORBFE_HIP(hipMemcpyAsync(match_q2t, m->b[4].p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    if (best) ORBFE_HIP(hipMemcpyAsync(best, m->b[5].p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    if (second) ORBFE_HIP(hipMemcpyAsync(second, m->b[6].p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(&nm, m->b[7].p, 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    if (nmatches) *nmatches = nm;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_match_bf_blocks_device(orbfe_matcher *m, const orbfe_keypoint *d_qkps, const uint8_t *d_qdesc,
                                                     const int32_t *d_qn, const orbfe_keypoint *d_tkps, const uint8_t *d_tdesc,
                                                     const int32_t *d_tn, int32_t cap, const int32_t *d_qframe,
                                                     const int32_t *d_tframe, int32_t npairs, float nnratio, int32_t th,
                                                     int32_t check_ori, int32_t *d_match_q2t, int32_t *d_nmatches, void *stream)
{
    if (!m || !d_qkps || !d_qdesc || !d_qn || !d_tkps || !d_tdesc || !d_tn || !d_qframe || !d_tframe || !d_match_q2t ||
        !d_nmatches || cap < 1 || cap > BM_MAX_NT || npairs < 0) {
        orbfe_set_error("bad argument to orbfe_match_bf_blocks_device / _frames_device (cap 1..%d)", BM_MAX_NT);
        return ORBFE_ERR_ARG;
    }
    if (npairs == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipStream_t st = (hipStream_t)stream;
    if (m->bf_kernel == 1)
        hipLaunchKernelGGL(k_match_popc, dim3((cap + 255) / 256, npairs), dim3(256), 0, st, d_qdesc, d_tdesc, d_qn, d_tn, d_qframe,
                           d_tframe, cap, 0, 0, nnratio, th, d_match_q2t, (int32_t *)nullptr, (int32_t *)nullptr);
    else
        hipLaunchKernelGGL(k_match_bf, dim3((cap + BM_QW - 1) / BM_QW, npairs), dim3(BM_WAVES * 64), 0, st, d_qdesc, d_tdesc,
                           d_qn, d_tn, d_qframe, d_tframe, cap, 0, 0, nnratio, th, d_match_q2t, (int32_t *)nullptr,
                           (int32_t *)nullptr);
    ORBFE_HIP(hipGetLastError());
    // orbfe_keypoint.angle, stride 7 floats
    hipLaunchKernelGGL(k_rot_prune, dim3(npairs), dim3(256), 0, st, d_match_q2t, &d_qkps->angle, &d_tkps->angle, 7, d_qn, d_qframe,
                       d_tframe, cap, 0, check_ori ? 1 : 0, d_nmatches);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_match_bf_frames_device(orbfe_matcher *m, const orbfe_keypoint *d_kps,
                                                     const uint8_t *d_desc, const int32_t *d_n, int32_t cap,
                                                     const int32_t *d_qframe, const int32_t *d_tframe, int32_t npairs,
                                                     float nnratio, int32_t th, int32_t check_ori,
                                                     int32_t *d_match_q2t, int32_t *d_nmatches, void *stream)
{
    return orbfe_match_bf_blocks_device(m, d_kps, d_desc, d_n, d_kps, d_desc, d_n, cap, d_qframe, d_tframe, npairs, nnratio, th,
                                        check_ori, d_match_q2t, d_nmatches, stream);
}

static bool csr_ok(const uint32_t *node, const uint32_t *off, const uint32_t *idx, int nn, int nfeat,
                   std::vector<uint8_t> &seen)
{
    seen.assign((size_t)std::max(nfeat, 1), 0);
    for (int a = 0; a < nn; ++a) {
        if (a > 0 && node[a] <= node[a - 1]) return false;
        if (off[a + 1] < off[a]) return false;
        for (uint32_t k = off[a]; k < off[a + 1]; ++k) {
            if (idx[k] >= (uint32_t)nfeat || seen[idx[k]]) return false;
            seen[idx[k]] = 1;
        }
    }
    return true;
}

extern "C" orbfe_status orbfe_search_by_bow(orbfe_matcher *m, const uint8_t *descKF, int32_t nKF,
                                            const uint8_t *validKF, const float *angKF, const uint32_t *nodeKF,
                                            const uint32_t *offKF, const uint32_t *idxKF, int32_t nnodesKF,
                                            const uint8_t *descF, int32_t nF, const uint8_t *validF,
                                            const float *angF, const uint32_t *nodeF, const uint32_t *offF,
                                            const uint32_t *idxF, int32_t nnodesF, float nnratio, int32_t th_low,
                                            int32_t strict_lt, int32_t check_ori, int32_t *matchF2KF,
                                            int32_t *nmatches)
{
    if (!m || nKF < 0 || nF < 0 || nnodesKF < 0 || nnodesF < 0 || (nF > 0 && !matchF2KF) ||
        (nnodesKF > 0 && (!nodeKF || !offKF || !descKF)) || (nnodesF > 0 && (!nodeF || !offF || !descF)) ||
        (check_ori && (nKF > 0 && nF > 0) && (!angKF || !angF))) {
        orbfe_set_error("bad argument to orbfe_search_by_bow");
        return ORBFE_ERR_ARG;
    }
    for (int i = 0; i < nF; ++i) matchF2KF[i] = -1;
    if (nmatches) *nmatches = 0;
    if (nKF == 0 || nF == 0 || nnodesKF == 0 || nnodesF == 0) return ORBFE_OK;
    std::vector<uint8_t> seen;
    if (!csr_ok(nodeKF, offKF, idxKF, nnodesKF, nKF, seen) || !csr_ok(nodeF, offF, idxF, nnodesF, nF, seen)) {
        orbfe_set_error("feature vector CSR invalid: node ids must ascend, indices in range and unique");
        return ORBFE_ERR_ARG;
    }
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    const size_t nikf = offKF[nnodesKF], nif = offF[nnodesF];
    const size_t sz[14] = {(size_t)nKF * 32, (size_t)nKF, (size_t)nKF * 4, (size_t)nnodesKF * 4,
                           (size_t)(nnodesKF + 1) * 4, nikf * 4, (size_t)nF * 32, (size_t)nF, (size_t)nF * 4,
                           (size_t)nnodesF * 4, (size_t)(nnodesF + 1) * 4, nif * 4, (size_t)nF * 4, 4};
    const void *src[12] = {descKF, validKF, angKF, nodeKF, offKF, idxKF, descF, validF, angF, nodeF, offF, idxF};
    for (int i = 0; i < 14; ++i) ORBFE_HIP(m->b[i].ensure(sz[i]));
    for (int i = 0; i < 12; ++i)
        if (src[i] && sz[i]) ORBFE_HIP(hipMemcpyAsync(m->b[i].p, src[i], sz[i], hipMemcpyHostToDevice, st));
    ORBFE_HIP(hipMemsetAsync(m->b[12].p, 0xFF, (size_t)nF * 4, st));
    hipLaunchKernelGGL(k_search_by_bow, dim3((nnodesKF + 63) / 64), dim3(64), 0, st, (const uint8_t *)m->b[0].p,
                       validKF ? (const uint8_t *)m->b[1].p : nullptr, (const uint32_t *)m->b[3].p,
                       (const uint32_t *)m->b[4].p, (const uint32_t *)m->b[5].p, nnodesKF, (const uint8_t *)m->b[6].p,
                       validF ? (const uint8_t *)m->b[7].p : nullptr, (const uint32_t *)m->b[9].p,
                       (const uint32_t *)m->b[10].p, (const uint32_t *)m->b[11].p, nnodesF, nnratio, th_low,
                       strict_lt ? 1 : 0, (int32_t *)m->b[12].p);
    ORBFE_HIP(hipGetLastError());
    // histogram key = F feature i, rot = angKF[match[i]] - angF[i] (:308, :759)
    hipLaunchKernelGGL(k_rot_prune_bow, dim3(1), dim3(256), 0, st, (int32_t *)m->b[12].p, (const float *)m->b[2].p,
                       (const float *)m->b[8].p, nF, check_ori ? 1 : 0, (int32_t *)m->b[13].p);
    ORBFE_HIP(hipGetLastError());
    int32_t nm = 0;
    ORBFE_HIP(hipMemcpyAsync(matchF2KF, m->b[12].p, (size_t)nF * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(&nm, m->b[13].p, 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    if (nmatches) *nmatches = nm;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_hamming_csr_ex(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                                             const uint32_t *off, const uint32_t *cand, int32_t *best_idx, int32_t *best,
                                             int32_t *second, int32_t *second_idx)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!q || !off || !best_idx || !best || !second))) {
        orbfe_set_error("bad argument to orbfe_hamming_csr");
        return ORBFE_ERR_ARG;
    }
    if (nq == 0) return ORBFE_OK;
    const size_t nc = off[nq];
    for (int i = 0; i < nq; ++i)
        if (off[i + 1] < off[i]) { orbfe_set_error("CSR offsets must not decrease"); return ORBFE_ERR_ARG; }
    for (size_t k = 0; k < nc; ++k)
        if (cand[k] >= (uint32_t)nt) { orbfe_set_error("candidate index out of range"); return ORBFE_ERR_ARG; }
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    ORBFE_HIP(m->b[0].ensure((size_t)nq * 32));
    ORBFE_HIP(m->b[1].ensure((size_t)nt * 32));
    ORBFE_HIP(m->b[2].ensure((size_t)(nq + 1) * 4));
    ORBFE_HIP(m->b[3].ensure(nc * 4));
    ORBFE_HIP(m->b[4].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[5].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[6].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[7].ensure((size_t)nq * 4));
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nt > 0) ORBFE_HIP(hipMemcpyAsync(m->b[1].p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    ORBFE_HIP(hipMemcpyAsync(m->b[2].p, off, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, st));
    if (nc > 0) ORBFE_HIP(hipMemcpyAsync(m->b[3].p, cand, nc * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_hamming_csr, dim3((nq + 255) / 256), dim3(256), 0, st, (const uint8_t *)m->b[0].p, nq,
                       (const uint8_t *)m->b[1].p, (const uint32_t *)m->b[2].p, (const uint32_t *)m->b[3].p,
                       (int32_t *)m->b[4].p, (int32_t *)m->b[5].p, (int32_t *)m->b[6].p, (int32_t *)m->b[7].p);
    ORBFE_HIP(hipGetLastError());
    if (second_idx) ORBFE_HIP(hipMemcpyAsync(second_idx, m->b[7].p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(best_idx, m->b[4].p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(best, m->b[5].p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(second, m->b[6].p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_hamming_csr_all(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                                              const uint32_t *off, const uint32_t *cand, uint16_t *dist)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!q || !off))) {
        orbfe_set_error("bad argument to orbfe_hamming_csr_all");
        return ORBFE_ERR_ARG;
    }
    if (nq == 0) return ORBFE_OK;
    const size_t nc = off[nq];
    for (int i = 0; i < nq; ++i)
        if (off[i + 1] < off[i]) { orbfe_set_error("CSR offsets must not decrease"); return ORBFE_ERR_ARG; }
    if (nc == 0) return ORBFE_OK;
    if (!cand || !dist || !t) { orbfe_set_error("bad argument to orbfe_hamming_csr_all"); return ORBFE_ERR_ARG; }
    for (size_t k = 0; k < nc; ++k)
        if (cand[k] >= (uint32_t)nt) { orbfe_set_error("candidate index out of range"); return ORBFE_ERR_ARG; }
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));
    ORBFE_HIP(m->b[0].ensure((size_t)nq * 32));
    ORBFE_HIP(m->b[1].ensure((size_t)nt * 32));
    ORBFE_HIP(m->b[2].ensure((size_t)(nq + 1) * 4));
    ORBFE_HIP(m->b[3].ensure(nc * 4));
    ORBFE_HIP(m->b[4].ensure(nc * 2));
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    ORBFE_HIP(hipMemcpyAsync(m->b[1].p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    ORBFE_HIP(hipMemcpyAsync(m->b[2].p, off, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, st));
    ORBFE_HIP(hipMemcpyAsync(m->b[3].p, cand, nc * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_hamming_csr_all, dim3((nq + 3) / 4), dim3(256), 0, st, (const uint8_t *)m->b[0].p, nq,
                       (const uint8_t *)m->b[1].p, (const uint32_t *)m->b[2].p, (const uint32_t *)m->b[3].p, (uint16_t *)m->b[4].p);
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpyAsync(dist, m->b[4].p, nc * 2, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_hamming_csr(orbfe_matcher *m, const uint8_t *q, int32_t nq, const uint8_t *t, int32_t nt,
                                          const uint32_t *off, const uint32_t *cand, int32_t *best_idx, int32_t *best,
                                          int32_t *second)
{
    return orbfe_hamming_csr_ex(m, q, nq, t, nt, off, cand, best_idx, best, second, nullptr);
}

// device-resident form: everything already in HBM (e.g. descriptors of an extractor output block), enqueued on `stream`
extern "C" orbfe_status orbfe_hamming_csr_device(orbfe_matcher *m, const uint8_t *d_q, int32_t nq, const uint8_t *d_t,
                                                 const uint32_t *d_off, const uint32_t *d_cand, int32_t *d_best_idx,
                                                 int32_t *d_best, int32_t *d_second, int32_t *d_second_idx, void *stream)
{
    if (!m || nq < 0 || (nq > 0 && (!d_q || !d_t || !d_off || !d_cand || !d_best_idx || !d_best || !d_second))) {
        orbfe_set_error("bad argument to orbfe_hamming_csr_device");
        return ORBFE_ERR_ARG;
    }
    if (nq == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipLaunchKernelGGL(k_hamming_csr, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_q, nq, d_t, d_off, d_cand,
                       d_best_idx, d_best, d_second, d_second_idx);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// SURVEY 8(f).2  Frame grid index: AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea (src/Frame.cc:319-334, 465-531)
// ---------------------------------------------------------------------------------------------------
#define GRID_NC (ORBFE_GRID_COLS * ORBFE_GRID_ROWS)

__device__ __forceinline__ int grid_cell_of(float x, float y, float minx, float miny, float gwi, float ghi)
{
    const int px = (int)roundf(__fmul_rn(__fsub_rn(x, minx), gwi));  // :525 std::round(float)
    const int py = (int)roundf(__fmul_rn(__fsub_rn(y, miny), ghi));
    if (px < 0 || px >= ORBFE_GRID_COLS || py < 0 || py >= ORBFE_GRID_ROWS) return -1;
    return px * ORBFE_GRID_ROWS + py;
}

// one workgroup: histogram over the 3072 cells (LDS), scan, placement, then each cell's short list is put into
// ascending keypoint order (the reference push_backs in keypoint order)
// xs = floats between consecutive points (2 for packed (x, y), 7 for orbfe_keypoint records).  Batched form (d_n != null):
// workgroup f takes frame f of an extractor output block -- points at f * cap * xs, count d_n[f] -- and writes its own
// cell_off [GRID_NC + 1], cell_idx [cap] and n_in.
__global__ __launch_bounds__(1024) void k_assign_grid(const float *__restrict__ xy, int n, float minx, float miny, float gwi,
                                                      float ghi, uint32_t *__restrict__ cell_off,
                                                      uint32_t *__restrict__ cell_idx, int32_t *__restrict__ n_in, int xs,
                                                      int cap, const int32_t *__restrict__ d_n)
{
    __shared__ uint32_t s_cnt[GRID_NC], s_off[GRID_NC + 1];
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    if (d_n) {
        const int f = blockIdx.x;
        n = min(d_n[f], cap);
        xy += (int64_t)f * cap * xs;
        cell_off += (int64_t)f * (GRID_NC + 1);
        cell_idx += (int64_t)f * cap;
        n_in += f;
    }
    for (int c = tid; c < GRID_NC; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int c = grid_cell_of(xy[(int64_t)xs * i], xy[(int64_t)xs * i + 1], minx, miny, gwi, ghi);
        if (c >= 0) atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    // exclusive scan of 3072 counters: 3 per thread
    uint32_t loc[3], sum = 0;
    for (int k = 0; k < 3; ++k) { loc[k] = s_cnt[tid * 3 + k]; sum += loc[k]; }
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t base = s_part[tid] - sum;
    for (int k = 0; k < 3; ++k) { s_off[tid * 3 + k] = base; base += loc[k]; }
    if (tid == 1023) s_off[GRID_NC] = s_part[1023];
    __syncthreads();
    for (int c = tid; c <= GRID_NC; c += 1024) cell_off[c] = s_off[c];
    if (tid == 0) *n_in = (int32_t)s_off[GRID_NC];
    for (int c = tid; c < GRID_NC; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int c = grid_cell_of(xy[(int64_t)xs * i], xy[(int64_t)xs * i + 1], minx, miny, gwi, ghi);
        if (c >= 0) cell_idx[s_off[c] + atomicAdd(&s_cnt[c], 1u)] = (uint32_t)i;
    }
    __syncthreads();
    __threadfence_block();
    for (int c = tid; c < GRID_NC; c += 1024) {  // insertion sort of the (short) cell lists
        const uint32_t o = s_off[c], e = s_off[c + 1];
        for (uint32_t a = o + 1; a < e; ++a) {
            const uint32_t v = cell_idx[a];
            uint32_t bpos = a;
            while (bpos > o && cell_idx[bpos - 1] > v) { cell_idx[bpos] = cell_idx[bpos - 1]; --bpos; }
            cell_idx[bpos] = v;
        }
    }
}

// GetFeaturesInArea for query i; write == false only counts.  Returns the count.
__device__ int area_query(const float *__restrict__ xy, const int32_t *__restrict__ octave,
                          const uint32_t *__restrict__ cell_off, const uint32_t *__restrict__ cell_idx, float minx,
                          float miny, float gwi, float ghi, float x, float y, float r, int minL, int maxL,
                          uint32_t *out, bool write, int xs = 2, int os = 1)
{
    int nminx = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, minx), r), gwi));  // :470
    nminx = max(nminx, 0);
    if (nminx >= ORBFE_GRID_COLS) return 0;
    int nmaxx = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, minx), r), gwi));
    nmaxx = min(nmaxx, ORBFE_GRID_COLS - 1);
    if (nmaxx < 0) return 0;
    int nminy = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, miny), r), ghi));
    nminy = max(nminy, 0);
    if (nminy >= ORBFE_GRID_ROWS) return 0;
    int nmaxy = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, miny), r), ghi));
    nmaxy = min(nmaxy, ORBFE_GRID_ROWS - 1);
    if (nmaxy < 0) return 0;
    const bool check = (minL > 0) || (maxL >= 0);  // :486
    int cnt = 0;
    for (int ix = nminx; ix <= nmaxx; ++ix)
        for (int iy = nminy; iy <= nmaxy; ++iy) {
            const int c = ix * ORBFE_GRID_ROWS + iy;
            for (uint32_t j = cell_off[c]; j < cell_off[c + 1]; ++j) {
                const uint32_t k = cell_idx[j];
                if (check) {
                    const int o = octave[(size_t)os * k];
                    if (o < minL) continue;
                    if (maxL >= 0 && o > maxL) continue;
                }
                const float dx = __fsub_rn(xy[(size_t)xs * k], x), dy = __fsub_rn(xy[(size_t)xs * k + 1], y);
                if (fabsf(dx) < r && fabsf(dy) < r) {
                    if (write) out[cnt] = k;
                    ++cnt;
                }
            }
        }
    return cnt;
}

__global__ __launch_bounds__(256) void k_area_count(const float *xy, const int32_t *octave, const uint32_t *cell_off,
                                                    const uint32_t *cell_idx, float minx, float miny, float gwi,
                                                    float ghi, const float *qxyr, const int32_t *qlv, int nq,
                                                    uint32_t *cnt, int xs, int os)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    cnt[i] = (uint32_t)area_query(xy, octave, cell_off, cell_idx, minx, miny, gwi, ghi, qxyr[3 * i], qxyr[3 * i + 1],
                                  qxyr[3 * i + 2], qlv ? qlv[2 * i] : -1, qlv ? qlv[2 * i + 1] : -1, nullptr, false, xs, os);
}

// single-workgroup exclusive scan cnt[0..nq) -> off[0..nq]
__global__ __launch_bounds__(1024) void k_scan_u32(const uint32_t *__restrict__ cnt, int nq, uint32_t *__restrict__ off)
{
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nq; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < nq ? cnt[i] : 0u;
        s_part[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t t = tid >= d ? s_part[tid - d] : 0u;
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        if (i < nq) off[i] = s_carry + s_part[tid] - v;
        __syncthreads();
        if (tid == 1023) s_carry += s_part[1023];
        __syncthreads();
    }
    if (tid == 0) off[nq] = s_carry;
}

__global__ __launch_bounds__(256) void k_area_write(const float *xy, const int32_t *octave, const uint32_t *cell_off,
                                                    const uint32_t *cell_idx, float minx, float miny, float gwi,
                                                    float ghi, const float *qxyr, const int32_t *qlv, int nq,
                                                    const uint32_t *off, uint32_t *cand, uint32_t cap, int xs, int os)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq || off[nq] > cap) return;
    area_query(xy, octave, cell_off, cell_idx, minx, miny, gwi, ghi, qxyr[3 * i], qxyr[3 * i + 1], qxyr[3 * i + 2],
               qlv ? qlv[2 * i] : -1, qlv ? qlv[2 * i + 1] : -1, cand + off[i], true, xs, os);
}

extern "C" orbfe_status orbfe_features_in_area_device(orbfe_matcher *m, const orbfe_keypoint *d_kps,
                                                      const uint32_t *d_cell_off, const uint32_t *d_cell_idx, float minx,
                                                      float miny, float gw_inv, float gh_inv, const float *d_qxyr,
                                                      const int32_t *d_qlevels, int32_t nq, uint32_t *d_off,
                                                      uint32_t *d_cand, int32_t cap, void *stream)
{
    if (!m || nq < 0 || cap < 0 || !d_off || (nq > 0 && (!d_kps || !d_cell_off || !d_cell_idx || !d_qxyr || (cap > 0 && !d_cand)))) {
        orbfe_set_error("bad argument to orbfe_features_in_area_device");
        return ORBFE_ERR_ARG;
    }
    MDeviceGuard g(m->device);
    hipStream_t st = (hipStream_t)stream;
    ORBFE_HIP(scratch_acquire(m, st));
    ORBFE_HIP(m->b[7].ensure((size_t)std::max(nq, 1) * 4));  // per-query counts
    const float *xy = (const float *)d_kps;                  // record = 7 floats: (x, y) first, octave sixth
    const int32_t *oct = (const int32_t *)d_kps + 5;
    static_assert(offsetof(orbfe_keypoint, octave) == 20 && sizeof(orbfe_keypoint) == 28, "keypoint record layout");
    if (nq > 0) {
        hipLaunchKernelGGL(k_area_count, dim3((nq + 255) / 256), dim3(256), 0, st, xy, oct, d_cell_off, d_cell_idx, minx, miny, gw_inv,
                           gh_inv, d_qxyr, d_qlevels, nq, (uint32_t *)m->b[7].p, 7, 7);
    }
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, (const uint32_t *)m->b[7].p, nq, d_off);
    if (nq > 0)
        hipLaunchKernelGGL(k_area_write, dim3((nq + 255) / 256), dim3(256), 0, st, xy, oct, d_cell_off, d_cell_idx, minx, miny, gw_inv,
                           gh_inv, d_qxyr, d_qlevels, nq, (const uint32_t *)d_off, d_cand, (uint32_t)cap, 7, 7);
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(scratch_release(m, st));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_assign_grid(orbfe_matcher *m, const float *xy, int32_t n, float minx, float miny,
                                          float gw_inv, float gh_inv, uint32_t *cell_off, uint32_t *cell_idx,
                                          int32_t *n_in_grid)
{
    if (!m || n < 0 || !cell_off || (n > 0 && (!xy || !cell_idx))) {
        orbfe_set_error("bad argument to orbfe_assign_grid");
        return ORBFE_ERR_ARG;
    }
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    ORBFE_HIP(m->b[0].ensure((size_t)n * 8));
    ORBFE_HIP(m->b[1].ensure((size_t)(GRID_NC + 1) * 4));
    ORBFE_HIP(m->b[2].ensure((size_t)n * 4));
    ORBFE_HIP(m->b[3].ensure(4));
    if (n > 0) ORBFE_HIP(hipMemcpyAsync(m->b[0].p, xy, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_assign_grid, dim3(1), dim3(1024), 0, st, (const float *)m->b[0].p, n, minx, miny, gw_inv, gh_inv,
                       (uint32_t *)m->b[1].p, (uint32_t *)m->b[2].p, (int32_t *)m->b[3].p, 2, 0, (const int32_t *)nullptr);
    ORBFE_HIP(hipGetLastError());
    int32_t nin = 0;
    ORBFE_HIP(hipMemcpyAsync(cell_off, m->b[1].p, (size_t)(GRID_NC + 1) * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(&nin, m->b[3].p, 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    if (nin > 0) ORBFE_HIP(hipMemcpy(cell_idx, m->b[2].p, (size_t)nin * 4, hipMemcpyDeviceToHost));
    if (n_in_grid) *n_in_grid = nin;
    return ORBFE_OK;
}

// Host form of the same index (no device, no matcher handle): for callers that hold the keypoints but cannot reach the
// grid itself -- KeyFrame::mGrid is a protected member of the reference (include/KeyFrame.h:223), so the matcher shim
// rebuilds it from the public mvKeysUn.  Counting sort over the 3072 cells; per cell ascending keypoint index.
extern "C" orbfe_status orbfe_assign_grid_host(const float *xy, int32_t n, float minx, float miny, float gw_inv, float gh_inv,
                                               uint32_t *cell_off, uint32_t *cell_idx, int32_t *n_in_grid)
{
    if (n < 0 || !cell_off || (n > 0 && (!xy || !cell_idx))) {
        orbfe_set_error("bad argument to orbfe_assign_grid_host");
        return ORBFE_ERR_ARG;
    }
    auto cell = [&](int i) -> int {
        // src/Frame.cc:525-526: round() of a float product (no contraction: this file is built with -ffp-contract=off)
        const float fx = (xy[2 * (size_t)i] - minx) * gw_inv, fy = (xy[2 * (size_t)i + 1] - miny) * gh_inv;
        const float rx = roundf(fx), ry = roundf(fy);
        if (!(rx >= 0.f && rx < (float)ORBFE_GRID_COLS && ry >= 0.f && ry < (float)ORBFE_GRID_ROWS)) return -1;
        return (int)rx * ORBFE_GRID_ROWS + (int)ry;
    };
    for (int c = 0; c <= GRID_NC; ++c) cell_off[c] = 0u;
    for (int i = 0; i < n; ++i) {
        const int c = cell(i);
        if (c >= 0) cell_off[c + 1]++;
    }
    for (int c = 0; c < GRID_NC; ++c) cell_off[c + 1] += cell_off[c];
    std::vector<uint32_t> fill(cell_off, cell_off + GRID_NC);
    for (int i = 0; i < n; ++i) {
        const int c = cell(i);
        if (c >= 0) cell_idx[fill[(size_t)c]++] = (uint32_t)i;
    }
    if (n_in_grid) *n_in_grid = (int32_t)cell_off[GRID_NC];
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_assign_grid_batch_device(orbfe_matcher *m, const orbfe_keypoint *d_kps, const int32_t *d_n,
                                                       int32_t cap, int32_t nframes, float minx, float miny, float gw_inv,
                                                       float gh_inv, uint32_t *d_cell_off, uint32_t *d_cell_idx,
                                                       int32_t *d_n_in_grid, void *stream)
{
    if (!m || nframes < 0 || cap < 0 || (nframes > 0 && (!d_kps || !d_n || !d_cell_off || !d_cell_idx || !d_n_in_grid))) {
        orbfe_set_error("bad argument to orbfe_assign_grid_batch_device");
        return ORBFE_ERR_ARG;
    }
    if (nframes == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    static_assert(sizeof(orbfe_keypoint) == 7 * sizeof(float), "keypoint record = 7 floats, (x, y) first");
    hipLaunchKernelGGL(k_assign_grid, dim3(nframes), dim3(1024), 0, (hipStream_t)stream, (const float *)d_kps, 0, minx, miny, gw_inv,
                       gh_inv, d_cell_off, d_cell_idx, d_n_in_grid, 7, cap, d_n);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_features_in_area(orbfe_matcher *m, const float *xy, const int32_t *octave, int32_t n,
                                               const uint32_t *cell_off, const uint32_t *cell_idx, float minx,
                                               float miny, float gw_inv, float gh_inv, const float *qxyr,
                                               const int32_t *qlevels, int32_t nq, uint32_t *off, uint32_t *cand,
                                               int32_t cap)
{
    if (!m || n < 0 || nq < 0 || cap < 0 || !cell_off || !off || (nq > 0 && !qxyr) || (n > 0 && (!xy || !octave || !cell_idx))) {
        orbfe_set_error("bad argument to orbfe_features_in_area");
        return ORBFE_ERR_ARG;
    }
    const uint32_t nin = cell_off[GRID_NC];
    if (nin > (uint32_t)n) { orbfe_set_error("cell_off inconsistent with n"); return ORBFE_ERR_ARG; }
    for (int c = 0; c < GRID_NC; ++c)
        if (cell_off[c + 1] < cell_off[c]) { orbfe_set_error("cell_off must not decrease"); return ORBFE_ERR_ARG; }
    for (uint32_t k = 0; k < nin; ++k)
        if (cell_idx[k] >= (uint32_t)n) { orbfe_set_error("cell_idx out of range"); return ORBFE_ERR_ARG; }
    off[0] = 0;
    if (nq == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    const size_t sz[8] = {(size_t)n * 8, (size_t)n * 4, (size_t)(GRID_NC + 1) * 4, (size_t)nin * 4, (size_t)nq * 12,
                          (size_t)nq * 8, (size_t)(nq + 1) * 4, (size_t)nq * 4};
    for (int i = 0; i < 8; ++i) ORBFE_HIP(m->b[i].ensure(sz[i]));
    ORBFE_HIP(m->b[8].ensure((size_t)std::max(cap, 1) * 4));
    const void *src[6] = {xy, octave, cell_off, cell_idx, qxyr, qlevels};
    for (int i = 0; i < 6; ++i)
        if (src[i] && sz[i]) ORBFE_HIP(hipMemcpyAsync(m->b[i].p, src[i], sz[i], hipMemcpyHostToDevice, st));
    const int32_t *dql = qlevels ? (const int32_t *)m->b[5].p : nullptr;
    hipLaunchKernelGGL(k_area_count, dim3((nq + 255) / 256), dim3(256), 0, st, (const float *)m->b[0].p,
                       (const int32_t *)m->b[1].p, (const uint32_t *)m->b[2].p, (const uint32_t *)m->b[3].p, minx, miny,
                       gw_inv, gh_inv, (const float *)m->b[4].p, dql, nq, (uint32_t *)m->b[7].p, 2, 1);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, (const uint32_t *)m->b[7].p, nq, (uint32_t *)m->b[6].p);
    hipLaunchKernelGGL(k_area_write, dim3((nq + 255) / 256), dim3(256), 0, st, (const float *)m->b[0].p,
                       (const int32_t *)m->b[1].p, (const uint32_t *)m->b[2].p, (const uint32_t *)m->b[3].p, minx, miny,
                       gw_inv, gh_inv, (const float *)m->b[4].p, dql, nq, (const uint32_t *)m->b[6].p,
                       (uint32_t *)m->b[8].p, (uint32_t)cap, 2, 1);
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpyAsync(off, m->b[6].p, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    if (off[nq] > (uint32_t)cap) {
        orbfe_set_error("cap=%d too small for %u candidates", cap, off[nq]);
        return ORBFE_ERR_CAP;
    }
    if (off[nq] > 0) ORBFE_HIP(hipMemcpy(cand, m->b[8].p, (size_t)off[nq] * 4, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// the projection-gated searches of the per-frame tracker
//   ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*>&, th)                 src/ORBmatcher.cc:63-157
//   ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) src/ORBmatcher.cc:1578-1724
//   (+ the perfect/ overload that also returns the 2-D point pairs, perfect/src/ORBmatcher.cc:1727-1911)
// The pose projection and its gates stay on the host (they run on cv::Mat in the caller's arithmetic); what comes here is
// one query per surviving MapPoint: GetFeaturesInArea on the frame's grid, the right-image gate, best / second-best Hamming
// over the candidates whose slot is free, the acceptance rule.  The reference's loop is NOT a map over the queries: an
// accepted query writes its MapPoint into F.mvpMapPoints[bestIdx], and later queries skip a slot that holds a point with
// Observations() > 0 (:108-110 / :1647-1649).  That dependency only points backwards (query i sees the assignments of
// j < i), so the sequential result is the unique fixed point of "every query picks its best among the slots no EARLIER
// claiming query took", and it is reached by relaxation: all queries choose in parallel against the owner table of the
// previous round (owner[f] = lowest claiming query matched to f), the table is rebuilt, until no choice changes.  Query i
// is final one round after all j < i are -- rounds = longest dependency chain + 1 (2-4 on real frames, <= nq + 1 always).
// Launch structure: the candidate lists (GetFeaturesInArea) and the Hamming distances are independent per query and are
// spread over the chip with 16 lanes per query (k_proj_count -> k_scan_u32 -> k_proj_fill; every (cand | dist << 16) entry is
// materialised once); the relaxation rounds only re-scan those entries and run in ONE workgroup (k_proj_resolve, 16 lanes
// per query, owner table in LDS).
// ---------------------------------------------------------------------------------------------------
#define PJ_T 1024
#define PJ_L 16                  // lanes per query in the relaxation rounds
#define PJ_LC 64                 // lanes per query in the candidate search (one wave: a search window covers ~100 grid cells)
#define PJ_SKIP 0x1FFu           // distance field of an entry whose slot is blocked before the call / fails the right-image gate
#define PJ_MAX_NF 15360          // owner table in LDS (int32 per frame feature)
struct ProjArgs {
    const uint8_t *descF;
    const float *xyF;
    const int32_t *octF;
    int32_t nF, xs, os;          // xs / os: floats / ints between consecutive points (2 / 1 packed, 7 / 7 keypoint records)
    const uint32_t *cell_off, *cell_idx;
    float minx, miny, gwi, ghi;
    const float *uRight;         // may be null
    const uint8_t *blocked;      // may be null
    const float *inv_sigma2;     // per level, may be null (ORBFE_PROJ_CHI2_GATE then never applies)
    int32_t nlevels;
    const orbfe_proj_query *q;
    const uint8_t *qdesc;
    int32_t nq, th, ratio_rule;
    float nnratio;
    int32_t *match, *best, *second;
    uint32_t *cnt;               // [nq] candidates per query
    uint16_t *lcnt;              // [nq * PJ_LC] candidates found by each lane of the query's wave
    uint32_t *off;               // [nq + 1]
    uint32_t *ent;               // [ent_cap] cand | dist << 16
    uint32_t ent_cap;
    int32_t *status;             // [0] = entries needed when ent_cap is too small (else 0), [1] = rounds run
};

// The cell rectangle of GetFeaturesInArea (:470-484) and the level filter; false = the query has no candidates
struct ProjRect {
    int x0, y0, nx, ny;
    bool check;
};
__device__ __forceinline__ bool proj_rect(const ProjArgs &a, const orbfe_proj_query &Q, ProjRect &R)
{
    int nminx = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.u, a.minx), Q.r), a.gwi));
    nminx = max(nminx, 0);
    if (nminx >= ORBFE_GRID_COLS) return false;
    int nmaxx = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.u, a.minx), Q.r), a.gwi));
    nmaxx = min(nmaxx, ORBFE_GRID_COLS - 1);
    if (nmaxx < 0) return false;
    int nminy = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.v, a.miny), Q.r), a.ghi));
    nminy = max(nminy, 0);
    if (nminy >= ORBFE_GRID_ROWS) return false;
    int nmaxy = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.v, a.miny), Q.r), a.ghi));
    nmaxy = min(nmaxy, ORBFE_GRID_ROWS - 1);
    if (nmaxy < 0) return false;
    R.x0 = nminx; R.y0 = nminy; R.nx = nmaxx - nminx + 1; R.ny = nmaxy - nminy + 1;
    R.check = (Q.min_level > 0) || (Q.max_level >= 0);  // :486
    return R.nx > 0 && R.ny > 0;
}

// Lane `sub` of a query's wave walks its contiguous share of the cell sequence (ix outer, iy inner: the reference's
// order), so lane order = candidate order.  f(k) is called for every feature that passes the level filter and the box test.
template <typename F>
__device__ __forceinline__ void proj_walk(const ProjArgs &a, const orbfe_proj_query &Q, const ProjRect &R, int sub, F f)
{
    const int ncell = R.nx * R.ny, chunk = (ncell + PJ_LC - 1) / PJ_LC;
    const int c0 = sub * chunk, c1 = min(c0 + chunk, ncell);
    for (int c = c0; c < c1; ++c) {
        const int ix = R.x0 + c / R.ny, iy = R.y0 + c % R.ny;
        const int cell = ix * ORBFE_GRID_ROWS + iy;
        for (uint32_t j = a.cell_off[cell]; j < a.cell_off[cell + 1]; ++j) {
            const uint32_t k = a.cell_idx[j];
            if (R.check) {
                const int o = a.octF[(size_t)a.os * k];
                if (o < Q.min_level) continue;
                if (Q.max_level >= 0 && o > Q.max_level) continue;
            }
            const float dx = __fsub_rn(a.xyF[(size_t)a.xs * k], Q.u), dy = __fsub_rn(a.xyF[(size_t)a.xs * k + 1], Q.v);
            if (fabsf(dx) < Q.r && fabsf(dy) < Q.r) f(k);
        }
    }
}

__device__ __forceinline__ int wave_incl_scan_m(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_proj_count(ProjArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x, i = t / PJ_LC, sub = t % PJ_LC;
    if (i >= a.nq) return;   // whole waves leave together
    const orbfe_proj_query Q = a.q[i];
    ProjRect R;
    int n = 0;
    if (proj_rect(a, Q, R)) proj_walk(a, Q, R, sub, [&](uint32_t) { ++n; });
    a.lcnt[(size_t)i * PJ_LC + sub] = (uint16_t)n;
    int tot = n;
#pragma unroll
    for (int o = PJ_LC / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, PJ_LC);
    if (sub == 0) a.cnt[i] = (uint32_t)tot;
}

__global__ __launch_bounds__(256) void k_proj_fill(ProjArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x, i = t / PJ_LC, sub = t % PJ_LC;
    if (i >= a.nq || a.off[a.nq] > a.ent_cap) return;
    const orbfe_proj_query Q = a.q[i];
    ProjRect R;
    if (!proj_rect(a, Q, R)) return;   // wave-uniform
    const int mine = a.lcnt[(size_t)i * PJ_LC + sub];
    uint32_t o = a.off[i] + (uint32_t)(wave_incl_scan_m(mine) - mine);
    Desc8 dq;
    const uint32_t *p = (const uint32_t *)(a.qdesc + (int64_t)i * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) dq.w[k] = p[k];
    const bool gate = (Q.flags & ORBFE_PROJ_RIGHT_GATE) && a.uRight;
    const bool chi2 = (Q.flags & ORBFE_PROJ_CHI2_GATE) && a.inv_sigma2;
    proj_walk(a, Q, R, sub, [&](uint32_t f) {
        bool skip = a.blocked && a.blocked[f];                        // :108-110 / :1647-1649, state before the call
        if (!skip && gate) {                                          // :114-119 / :1654-1660
            const float ur = a.uRight[f];
            skip = ur > 0.f && fabsf(__fsub_rn(Q.ur, ur)) > Q.r;
        }
        if (!skip && chi2) {                                          // Fuse :1112-1139: reprojection error against the level's sigma
            const float ex = __fsub_rn(Q.u, a.xyF[(size_t)a.xs * f]), ey = __fsub_rn(Q.v, a.xyF[(size_t)a.xs * f + 1]);
            float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
            const float kr = a.uRight ? a.uRight[f] : -1.f;
            const int lv = min(max(a.octF[(size_t)a.os * f], 0), a.nlevels - 1);
            double bound = 5.99;
            if (kr >= 0.f) {
                const float er = __fsub_rn(Q.ur, kr);
                e2 = __fadd_rn(e2, __fmul_rn(er, er));
                bound = 7.8;
            }
            skip = (double)__fmul_rn(e2, a.inv_sigma2[lv]) > bound;
        }
        const uint32_t d = skip ? PJ_SKIP : (uint32_t)hamming8(dq, (const uint32_t *)(a.descF + (int64_t)f * 32));
        a.ent[o++] = f | (d << 16);
    });
}

// key of an entry in the reduction: distance (9 bits) above the position in the query's list (first in list order wins ties)
#define PJ_NOKEY 0xFFFFFFFFu
__global__ __launch_bounds__(PJ_T) void k_proj_resolve(ProjArgs a)
{
    extern __shared__ int32_t s_owner[];   // [nF]
    __shared__ int s_changed;
    const int tid = threadIdx.x, sub = tid % PJ_L, grp = tid / PJ_L;
    const int nq = a.nq, nF = a.nF;
    if (tid == 0) {
        const uint32_t total = a.off[nq];
        a.status[0] = total > a.ent_cap ? (int32_t)total : 0;
        a.status[1] = 0;
    }
    if (a.off[nq] > a.ent_cap) return;   // workgroup-uniform: the host grows the scratch and launches again
    for (int f = tid; f < nF; f += PJ_T) s_owner[f] = 0x7FFFFFFF;
    for (int i = tid; i < nq; i += PJ_T) a.match[i] = -1;
    __syncthreads();
    int round = 0;
    for (; round <= nq + 1; ++round) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        bool changed = false;
        for (int i0 = 0; i0 < nq; i0 += PJ_T / PJ_L) {
            const int i = i0 + grp;
            uint32_t k1 = PJ_NOKEY, k2 = PJ_NOKEY;   // the two smallest keys (dist << 16 | position) among the free slots
            uint32_t o = 0, e = 0;
            if (i < nq) { o = a.off[i]; e = a.off[i + 1]; }
            for (uint32_t k = o + sub; k < e; k += PJ_L) {
                const uint32_t en = a.ent[k];
                const uint32_t f = en & 0xFFFFu, d = en >> 16;
                if (d == PJ_SKIP || s_owner[f] < i) continue;   // the slot was taken by an earlier query of this call
                const uint32_t key = (d << 16) | (k - o);
                if (key < k1) { k2 = k1; k1 = key; }
                else if (key < k2) k2 = key;
            }
#pragma unroll
            for (int s = PJ_L / 2; s > 0; s >>= 1) {   // merge the lanes' pairs: the two smallest keys of the group
                const uint32_t o1 = __shfl_xor(k1, s, PJ_L), o2 = __shfl_xor(k2, s, PJ_L);
                const uint32_t lo = min(k1, o1), hi = max(k1, o1);
                k2 = min(hi, min(k2, o2));
                k1 = lo;
            }
            if (i < nq && sub == 0) {
                // :128-140: bestDist = smallest distance, first in list order; bestDist2 / bestLevel2 = the smallest among the
                // others, first in list order
                const int bestDist = k1 == PJ_NOKEY ? 256 : (int)(k1 >> 16), bestDist2 = k2 == PJ_NOKEY ? 256 : (int)(k2 >> 16);
                int mt = -1;
                if (bestDist <= a.th) {            // :143-148 / :1673
                    const int bestIdx = (int)(a.ent[o + (k1 & 0xFFFFu)] & 0xFFFFu);
                    bool reject = false;
                    if (a.ratio_rule) {
                        const int bestLevel = a.octF[(size_t)a.os * bestIdx];
                        const int bestLevel2 = k2 == PJ_NOKEY ? -1 : a.octF[(size_t)a.os * (a.ent[o + (k2 & 0xFFFFu)] & 0xFFFFu)];
                        reject = bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(a.nnratio, (float)bestDist2);
                    }
                    if (!reject) mt = bestIdx;
                }
                if (mt != a.match[i]) {
                    a.match[i] = mt;
                    changed = true;
                }
                a.best[i] = bestDist;
                a.second[i] = bestDist2;
            }
        }
        if (changed) s_changed = 1;
        __syncthreads();
        if (!s_changed) break;                 // workgroup-uniform
        for (int f = tid; f < nF; f += PJ_T) s_owner[f] = 0x7FFFFFFF;
        __syncthreads();
        for (int i = tid; i < nq; i += PJ_T) {
            const int mt = a.match[i];
            if (mt >= 0 && (a.q[i].flags & ORBFE_PROJ_CLAIMS)) atomicMin(&s_owner[mt], i);
        }
        __syncthreads();
    }
    if (tid == 0) a.status[1] = round + 1;
}


// ---------------------------------------------------------------------------------------------------
// The search in TWO launches instead of four and a copy (the per-frame members of Tracking are launch-bound: count -> scan -> fill ->
// resolve plus a result copy cost more than their kernels).  Same arithmetic, same fixed point:
//   * every query's wave walks its cell rectangle twice inside the launch (count, then fill -- the second walk finds its lines
//     in the cache) and writes its entries into a fixed slab of PJ_SLAB slots at i * PJ_SLAB: no scan over the queries, no
//     second launch.  A query with more candidates raises need[] and the host takes the four-kernel path (below) instead.
//   * while it fills, the wave already reduces the two smallest keys: round 0 of the relaxation (owner table empty) is decided
//     here, spread over the chip, for every query at once.
//   * a second, one-workgroup launch (k_proj_rounds) runs the remaining rounds.  A query re-scans its entries in
//     a round only if it has to: when the slot of its best or of its second-best candidate is now owned by an earlier query, or
//     when it ever skipped an owned slot (that slot may have been freed).  Every other query's two smallest free keys are what
//     they were, so its choice is what a full re-scan would return: the rounds and their results are those of k_proj_resolve.
//     On real frames a few dozen of ~800 queries re-scan.
//   * results go straight to page-locked host memory (match | best | second | status): two launches, no copy back, one wait.
// ---------------------------------------------------------------------------------------------------
#define PJ_SLAB 512
#define PJ_FT 256                // threads per workgroup: one wave per query in the fill, 16 lanes per query in the rounds
struct ProjFusedArgs {
    ProjArgs a;
    int32_t *f12;                // [2 * nq] feature of the best / second-best free candidate (-1: none)
    uint8_t *constrained;        // [nq] the query skipped an owned slot in its last scan
    uint32_t *done;              // [1] largest candidate count above PJ_SLAB (0: none); reset by k_proj_rounds
    int32_t *h_out;              // mapped host: match[nq] | best[nq] | second[nq] | status[2]
};

// the decision of :128-148 / :1673 from the two smallest keys of the free candidates and their features
__device__ __forceinline__ int proj_decide(const ProjArgs &a, uint32_t k1, uint32_t k2, int f1, int f2, int &bestDist, int &bestDist2)
{
    bestDist = k1 == PJ_NOKEY ? 256 : (int)(k1 >> 16);
    bestDist2 = k2 == PJ_NOKEY ? 256 : (int)(k2 >> 16);
    if (bestDist > a.th) return -1;
    if (a.ratio_rule) {
        const int bestLevel = a.octF[(size_t)a.os * f1];
        const int bestLevel2 = k2 == PJ_NOKEY ? -1 : a.octF[(size_t)a.os * f2];
        if (bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(a.nnratio, (float)bestDist2)) return -1;
    }
    return f1;
}

// merge (k1, f1, k2, f2) with a partner's over `width` lanes: the two smallest keys and their features
template <int WIDTH>
__device__ __forceinline__ void proj_reduce2(uint32_t &k1, int &f1, uint32_t &k2, int &f2)
{
#pragma unroll
    for (int s = WIDTH / 2; s > 0; s >>= 1) {
        const uint32_t o1 = __shfl_xor(k1, s, WIDTH), o2 = __shfl_xor(k2, s, WIDTH);
        const int g1 = __shfl_xor(f1, s, WIDTH), g2 = __shfl_xor(f2, s, WIDTH);
        // keys are unique inside a query (the position is part of them), NOKEY excepted
        uint32_t hi;
        int fh;
        if (o1 < k1) { hi = k1; fh = f1; k1 = o1; f1 = g1; } else { hi = o1; fh = g1; }
        const uint32_t m2 = min(k2, o2);
        const int fm = k2 <= o2 ? f2 : g2;
        if (hi <= m2) { k2 = hi; f2 = fh; } else { k2 = m2; f2 = fm; }
    }
}

__global__ __launch_bounds__(PJ_FT) void k_proj_fused(ProjFusedArgs p)
{
    const ProjArgs &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nq = a.nq;
    {
        const int i = blockIdx.x * (PJ_FT / 64) + wv;
        if (i < nq) {   // wave-uniform
            const orbfe_proj_query Q = a.q[i];
            ProjRect R;
            const bool any = proj_rect(a, Q, R);
            int n = 0;
            if (any) proj_walk(a, Q, R, lane, [&](uint32_t) { ++n; });
            const int incl = wave_incl_scan_m(n);
            const int tot = __shfl(incl, 63, 64);
            uint32_t k1 = PJ_NOKEY, k2 = PJ_NOKEY;
            int f1 = -1, f2 = -1;
            if (tot > PJ_SLAB) {
                if (lane == 0) atomicMax(&p.done[1], (uint32_t)tot);
            } else if (tot > 0) {
                uint32_t o = (uint32_t)(incl - n);
                uint32_t *ent = a.ent + (size_t)i * PJ_SLAB;
                Desc8 dq;
                const uint32_t *pq = (const uint32_t *)(a.qdesc + (int64_t)i * 32);
#pragma unroll
                for (int k = 0; k < 8; ++k) dq.w[k] = pq[k];
                const bool gate = (Q.flags & ORBFE_PROJ_RIGHT_GATE) && a.uRight;
                const bool chi2 = (Q.flags & ORBFE_PROJ_CHI2_GATE) && a.inv_sigma2;
                proj_walk(a, Q, R, lane, [&](uint32_t f) {
                    bool skip = a.blocked && a.blocked[f];
                    if (!skip && gate) {
                        const float ur = a.uRight[f];
                        skip = ur > 0.f && fabsf(__fsub_rn(Q.ur, ur)) > Q.r;
                    }
                    if (!skip && chi2) {
                        const float ex = __fsub_rn(Q.u, a.xyF[(size_t)a.xs * f]), ey = __fsub_rn(Q.v, a.xyF[(size_t)a.xs * f + 1]);
                        float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        const float kr = a.uRight ? a.uRight[f] : -1.f;
                        const int lv = min(max(a.octF[(size_t)a.os * f], 0), a.nlevels - 1);
                        double bound = 5.99;
                        if (kr >= 0.f) {
                            const float er = __fsub_rn(Q.ur, kr);
                            e2 = __fadd_rn(e2, __fmul_rn(er, er));
                            bound = 7.8;
                        }
                        skip = (double)__fmul_rn(e2, a.inv_sigma2[lv]) > bound;
                    }
                    const uint32_t d = skip ? PJ_SKIP : (uint32_t)hamming8(dq, (const uint32_t *)(a.descF + (int64_t)f * 32));
                    ent[o] = f | (d << 16);
                    if (!skip) {
                        const uint32_t key = (d << 16) | o;
                        if (key < k1) { k2 = k1; f2 = f1; k1 = key; f1 = (int)f; }
                        else if (key < k2) { k2 = key; f2 = (int)f; }
                    }
                    ++o;
                });
                proj_reduce2<64>(k1, f1, k2, f2);
            }
            if (lane == 0) {
                int bd, bd2;
                a.match[i] = proj_decide(a, k1, k2, f1, f2, bd, bd2);   // round 0: every slot free
                a.best[i] = bd;
                a.second[i] = bd2;
                a.cnt[i] = (uint32_t)min(tot, PJ_SLAB);
                p.f12[2 * i] = f1;
                p.f12[2 * i + 1] = f2;
                p.constrained[i] = 0;
            }
        }
    }
}

// the remaining rounds, ONE workgroup, launched behind k_proj_fused (the launch boundary makes the slabs visible; an in-kernel
// hand-over to "the last workgroup to arrive" was measured: the agent-scope fences cost more than the launch, 95 against 82 us).
// One workgroup is latency, not throughput: every dependent trip to memory is a microsecond.  The per-query state (choice, the
// features of the two smallest free keys, flags, distances, list length) is therefore read ONCE into LDS, the rounds run on LDS
// alone except for the entries of the queries that re-scan, and the results leave from LDS; 1024 threads (64 queries re-scan at
// a time).  Round-5 form: three global phases per round with 256 threads, 47 us for 786 queries.
#define PJ_RT 1024
#define PJ_ROUNDS_WORDS 8        // LDS words per query: list, match, f1, f2, flags, best, second, cnt
__global__ __launch_bounds__(PJ_RT) void k_proj_rounds(ProjFusedArgs p)
{
    extern __shared__ int32_t s_dyn[];     // [nF] owner table, then PJ_ROUNDS_WORDS arrays of [nq]
    __shared__ int s_changed, s_nlist;
    const ProjArgs &a = p.a;
    const int tid = threadIdx.x;
    const int nq = a.nq, nF = a.nF;
    int32_t *s_owner = s_dyn, *s_list = s_dyn + nF, *s_match = s_list + nq, *s_f1 = s_match + nq, *s_f2 = s_f1 + nq;
    int32_t *s_flag = s_f2 + nq, *s_best = s_flag + nq, *s_second = s_best + nq, *s_cnt = s_second + nq;
    const uint32_t need = p.done[1];
    for (int i = tid; i < nq; i += PJ_RT) {
        s_match[i] = a.match[i];
        s_f1[i] = p.f12[2 * i];
        s_f2[i] = p.f12[2 * i + 1];
        s_flag[i] = (a.q[i].flags & ORBFE_PROJ_CLAIMS) ? 1 : 0;   // bit 0: the query claims its slot; bit 1: it skipped an owned slot in its last scan
        s_best[i] = a.best[i];
        s_second[i] = a.second[i];
        s_cnt[i] = (int32_t)a.cnt[i];
    }
    int round = 1;
    if (need == 0) {
        for (; round <= nq + 2; ++round) {
            for (int f = tid; f < nF; f += PJ_RT) s_owner[f] = 0x7FFFFFFF;
            if (tid == 0) { s_changed = 0; s_nlist = 0; }
            __syncthreads();
            for (int i = tid; i < nq; i += PJ_RT) {
                const int mt = s_match[i];
                if (mt >= 0 && (s_flag[i] & 1)) atomicMin(&s_owner[mt], i);
            }
            __syncthreads();
            for (int i = tid; i < nq; i += PJ_RT) {
                const int f1 = s_f1[i], f2 = s_f2[i];
                if ((s_flag[i] & 2) || (f1 >= 0 && s_owner[f1] < i) || (f2 >= 0 && s_owner[f2] < i)) s_list[atomicAdd(&s_nlist, 1)] = i;
            }
            __syncthreads();
            const int nl = s_nlist;
            if (nl == 0) break;               // workgroup-uniform
            const int sub = tid % PJ_L, grp = tid / PJ_L;
            bool changed = false;
            for (int l0 = 0; l0 < nl; l0 += PJ_RT / PJ_L) {
                const int li = l0 + grp;
                const int i = li < nl ? s_list[li] : -1;
                uint32_t k1 = PJ_NOKEY, k2 = PJ_NOKEY;
                int f1 = -1, f2 = -1;
                bool skipped = false;
                if (i >= 0) {
                    const uint32_t *ent = a.ent + (size_t)i * PJ_SLAB;
                    const uint32_t e = (uint32_t)s_cnt[i];
                    for (uint32_t k = sub; k < e; k += PJ_L) {
                        const uint32_t en = ent[k];
                        const uint32_t f = en & 0xFFFFu, d = en >> 16;
                        if (d == PJ_SKIP) continue;
                        if (s_owner[f] < i) { skipped = true; continue; }   // taken by an earlier query of this call
                        const uint32_t key = (d << 16) | k;
                        if (key < k1) { k2 = k1; f2 = f1; k1 = key; f1 = (int)f; }
                        else if (key < k2) { k2 = key; f2 = (int)f; }
                    }
                }
                proj_reduce2<PJ_L>(k1, f1, k2, f2);
#pragma unroll
                for (int s = PJ_L / 2; s > 0; s >>= 1) skipped = skipped || __shfl_xor((int)skipped, s, PJ_L) != 0;
                if (i >= 0 && sub == 0) {
                    int bd, bd2;
                    const int mt = proj_decide(a, k1, k2, f1, f2, bd, bd2);
                    if (mt != s_match[i]) {
                        s_match[i] = mt;
                        changed = true;
                    }
                    s_best[i] = bd;
                    s_second[i] = bd2;
                    s_f1[i] = f1;
                    s_f2[i] = f2;
                    s_flag[i] = (s_flag[i] & 1) | (skipped ? 2 : 0);
                }
            }
            if (changed) s_changed = 1;
            __syncthreads();
            if (!s_changed) break;            // workgroup-uniform
            __syncthreads();
        }
    }
    __syncthreads();
    // results to the host; the counters back to zero for the next call
    for (int i = tid; i < nq; i += PJ_RT) {
        p.h_out[i] = s_match[i];
        p.h_out[nq + i] = s_best[i];
        p.h_out[2 * (size_t)nq + i] = s_second[i];
    }
    if (tid == 0) {
        p.h_out[3 * (size_t)nq] = (int32_t)need;   // > 0: some query has that many candidates; nothing above is valid
        p.h_out[3 * (size_t)nq + 1] = round + 1;
        p.done[1] = 0u;
    }
}

extern "C" orbfe_status orbfe_search_by_projection_chi2(orbfe_matcher *m, const uint8_t *descF, const float *xyF, const int32_t *octF,
                                                        int32_t nF, const uint32_t *cell_off, const uint32_t *cell_idx, float minx,
                                                        float miny, float gw_inv, float gh_inv, const float *uRight,
                                                        const uint8_t *blocked, const float *inv_level_sigma2, int32_t nlevels,
                                                        const orbfe_proj_query *q, const uint8_t *qdesc,
                                                        int32_t nq, int32_t th, float nnratio, int32_t ratio_rule, int32_t *match,
                                                        int32_t *best, int32_t *second)
{
    if (inv_level_sigma2 && (nlevels < 1 || nlevels > 64)) {
        orbfe_set_error("bad argument to orbfe_search_by_projection_chi2");
        return ORBFE_ERR_ARG;
    }
    if (!m || nF < 0 || nq < 0 || !cell_off || (nq > 0 && (!q || !qdesc || !match)) || (nF > 0 && (!descF || !xyF || !octF)) ||
        (cell_off[GRID_NC] > 0 && !cell_idx)) {   // an empty grid (no keypoint inside the image bounds) has no cell_idx
        orbfe_set_error("bad argument to orbfe_search_by_projection");
        return ORBFE_ERR_ARG;
    }
    if (nF > PJ_MAX_NF) { orbfe_set_error("orbfe_search_by_projection: at most %d frame features", PJ_MAX_NF); return ORBFE_ERR_SIZE; }
    if (th > 255) { orbfe_set_error("orbfe_search_by_projection: th must be below 256 (256 is the 'no candidate' distance)"); return ORBFE_ERR_ARG; }
    const uint32_t nin = cell_off[GRID_NC];
    if (nin > (uint32_t)nF) { orbfe_set_error("cell_off inconsistent with nF"); return ORBFE_ERR_ARG; }
    for (int c = 0; c < GRID_NC; ++c)
        if (cell_off[c + 1] < cell_off[c]) { orbfe_set_error("cell_off must not decrease"); return ORBFE_ERR_ARG; }
    for (uint32_t k = 0; k < nin; ++k)
        if (cell_idx[k] >= (uint32_t)nF) { orbfe_set_error("cell_idx out of range"); return ORBFE_ERR_ARG; }
    if (nq == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));
    // One pinned staging block in, one out: the per-frame call is latency-bound, nine pageable copies cost more than the
    // kernels.  Layout (256-byte aligned pieces): descF | xyF | octF | cell_off | cell_idx | uRight | blocked | q | qdesc
    const size_t sz[10] = {(size_t)nF * 32, (size_t)nF * 8, (size_t)nF * 4, (size_t)(GRID_NC + 1) * 4, (size_t)nin * 4,
                           uRight ? (size_t)nF * 4 : 0, blocked ? (size_t)nF : 0, (size_t)nq * sizeof(orbfe_proj_query), (size_t)nq * 32,
                           inv_level_sigma2 ? (size_t)nlevels * 4 : 0};
    const void *src[10] = {descF, xyF, octF, cell_off, cell_idx, uRight, blocked, q, qdesc, inv_level_sigma2};
    size_t at[11];
    at[0] = 0;
    for (int i = 0; i < 10; ++i) at[i + 1] = (at[i] + sz[i] + 255) & ~(size_t)255;
    const size_t out_bytes = (size_t)nq * 12 + 8;   // match | best | second | status[2]
    ORBFE_HIP(m->pin_in.ensure(at[10]));
    ORBFE_HIP(m->pin_out.ensure(out_bytes));
    ORBFE_HIP(m->b[0].ensure(at[10]));
    ORBFE_HIP(m->b[1].ensure(out_bytes));
    ORBFE_HIP(m->b[2].ensure((size_t)nq * 4));                 // cnt
    ORBFE_HIP(m->b[3].ensure((size_t)nq * PJ_LC * 2));         // lcnt
    ORBFE_HIP(m->b[4].ensure((size_t)(nq + 1) * 4));           // off
    for (int i = 0; i < 10; ++i)
        if (sz[i]) memcpy((char *)m->pin_in.p + at[i], src[i], sz[i]);
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, m->pin_in.p, at[10], hipMemcpyHostToDevice, st));
    const char *din = (const char *)m->b[0].p;
    ProjArgs a;
    a.descF = (const uint8_t *)(din + at[0]);
    a.xyF = (const float *)(din + at[1]);
    a.octF = (const int32_t *)(din + at[2]);
    a.nF = nF; a.xs = 2; a.os = 1;
    a.cell_off = (const uint32_t *)(din + at[3]);
    a.cell_idx = (const uint32_t *)(din + at[4]);
    a.minx = minx; a.miny = miny; a.gwi = gw_inv; a.ghi = gh_inv;
    a.uRight = uRight ? (const float *)(din + at[5]) : nullptr;
    a.blocked = blocked ? (const uint8_t *)(din + at[6]) : nullptr;
    a.inv_sigma2 = inv_level_sigma2 ? (const float *)(din + at[9]) : nullptr;
    a.nlevels = nlevels;
    a.q = (const orbfe_proj_query *)(din + at[7]);
    a.qdesc = (const uint8_t *)(din + at[8]);
    a.nq = nq; a.th = th; a.ratio_rule = ratio_rule ? 1 : 0; a.nnratio = nnratio;
    a.match = (int32_t *)m->b[1].p;
    a.best = a.match + nq;
    a.second = a.match + 2 * (size_t)nq;
    a.status = a.match + 3 * (size_t)nq;
    a.cnt = (uint32_t *)m->b[2].p;
    a.lcnt = (uint16_t *)m->b[3].p;
    a.off = (uint32_t *)m->b[4].p;
    const int32_t *hout = (const int32_t *)m->pin_out.p;
    // ONE launch (k_proj_fused) when the owner table and the re-scan list fit the LDS and no query overflows its slab; else (or on
    // overflow, reported in status[0]) the four-kernel path below
    const size_t fused_lds = ((size_t)std::max(nF, 1) + (size_t)PJ_ROUNDS_WORDS * (size_t)nq) * 4;
    if (m->proj_fused && fused_lds <= 64 * 1024) {
        ORBFE_HIP(m->b[5].ensure((size_t)nq * PJ_SLAB * 4));
        ORBFE_HIP(m->b[6].ensure((size_t)nq * 8));
        ORBFE_HIP(m->b[7].ensure((size_t)nq));
        if (!m->proj_done.p) {
            ORBFE_HIP(m->proj_done.ensure(256));
            ORBFE_HIP(hipMemsetAsync(m->proj_done.p, 0, 256, st));   // the kernel leaves its counters at zero
        }
        ProjFusedArgs fa;
        fa.a = a;
        fa.a.ent = (uint32_t *)m->b[5].p;
        fa.a.ent_cap = 0xFFFFFFFFu;
        fa.f12 = (int32_t *)m->b[6].p;
        fa.constrained = (uint8_t *)m->b[7].p;
        fa.done = (uint32_t *)m->proj_done.p;
        fa.h_out = (int32_t *)m->pin_out.p;   // page-locked and mapped: the kernel stores the results there
        const int nwg = (nq + PJ_FT / 64 - 1) / (PJ_FT / 64);
        hipLaunchKernelGGL(k_proj_fused, dim3(nwg), dim3(PJ_FT), 0, st, fa);
        hipLaunchKernelGGL(k_proj_rounds, dim3(1), dim3(PJ_RT), fused_lds, st, fa);
        ORBFE_HIP(hipGetLastError());
        ORBFE_HIP(hipStreamSynchronize(st));
        if (hout[3 * (size_t)nq] == 0) {
            memcpy(match, hout, (size_t)nq * 4);
            if (best) memcpy(best, hout + nq, (size_t)nq * 4);
            if (second) memcpy(second, hout + 2 * (size_t)nq, (size_t)nq * 4);
            return ORBFE_OK;
        }
    }
    const int ngrp = (nq * PJ_LC + 255) / 256;
    size_t ent_cap = std::max<size_t>((size_t)nq * 96, 1 << 16);
    for (int attempt = 0; attempt < 2; ++attempt) {
        ORBFE_HIP(m->b[5].ensure(ent_cap * 4));
        a.ent = (uint32_t *)m->b[5].p;
        a.ent_cap = (uint32_t)std::min<size_t>(m->b[5].bytes / 4, 0xFFFFFFFFu);
        if (attempt == 0) {
            hipLaunchKernelGGL(k_proj_count, dim3(ngrp), dim3(256), 0, st, a);
            hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, (const uint32_t *)a.cnt, nq, a.off);
        }
        hipLaunchKernelGGL(k_proj_fill, dim3(ngrp), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(PJ_T), (size_t)std::max(nF, 1) * 4, st, a);
        ORBFE_HIP(hipGetLastError());
        ORBFE_HIP(hipMemcpyAsync(m->pin_out.p, m->b[1].p, out_bytes, hipMemcpyDeviceToHost, st));
        ORBFE_HIP(hipStreamSynchronize(st));
        const int32_t need = hout[3 * (size_t)nq];
        if (need == 0) break;
        if (attempt == 1) { orbfe_set_error("candidate scratch still too small (%d entries)", need); return ORBFE_ERR_NOMEM; }
        ent_cap = (size_t)need;   // the exact need: the second launch cannot fail on it
    }
    memcpy(match, hout, (size_t)nq * 4);
    if (best) memcpy(best, hout + nq, (size_t)nq * 4);
    if (second) memcpy(second, hout + 2 * (size_t)nq, (size_t)nq * 4);
    return ORBFE_OK;
}

// The first two stages of the projection search on their own: GetFeaturesInArea for every query and the Hamming distance of
// every candidate, handed back as lists (entry = feature | distance << 16, in the reference's candidate order).  For callers
// whose acceptance rule is sequential in a way the device core does not implement (SearchForInitialization :571-574).
extern "C" orbfe_status orbfe_window_distances(orbfe_matcher *m, const uint8_t *descF, const float *xyF, const int32_t *octF, int32_t nF,
                                               const uint32_t *cell_off, const uint32_t *cell_idx, float minx, float miny, float gw_inv,
                                               float gh_inv, const orbfe_proj_query *q, const uint8_t *qdesc, int32_t nq, uint32_t *off,
                                               uint32_t *ent, int32_t cap)
{
    if (!m || nF < 0 || nq < 0 || cap < 0 || !cell_off || !off || (nq > 0 && (!q || !qdesc)) || (nF > 0 && (!descF || !xyF || !octF)) ||
        (cell_off[GRID_NC] > 0 && !cell_idx) || (cap > 0 && !ent)) {
        orbfe_set_error("bad argument to orbfe_window_distances");
        return ORBFE_ERR_ARG;
    }
    if (nF > 65535) { orbfe_set_error("orbfe_window_distances: at most 65535 features (16-bit index in an entry)"); return ORBFE_ERR_SIZE; }
    const uint32_t nin = cell_off[GRID_NC];
    if (nin > (uint32_t)nF) { orbfe_set_error("cell_off inconsistent with nF"); return ORBFE_ERR_ARG; }
    for (int c = 0; c < GRID_NC; ++c)
        if (cell_off[c + 1] < cell_off[c]) { orbfe_set_error("cell_off must not decrease"); return ORBFE_ERR_ARG; }
    for (uint32_t k = 0; k < nin; ++k)
        if (cell_idx[k] >= (uint32_t)nF) { orbfe_set_error("cell_idx out of range"); return ORBFE_ERR_ARG; }
    off[0] = 0;
    if (nq == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));
    const size_t sz[7] = {(size_t)nF * 32, (size_t)nF * 8, (size_t)nF * 4, (size_t)(GRID_NC + 1) * 4, (size_t)nin * 4,
                          (size_t)nq * sizeof(orbfe_proj_query), (size_t)nq * 32};
    const void *src[7] = {descF, xyF, octF, cell_off, cell_idx, q, qdesc};
    size_t at[8];
    at[0] = 0;
    for (int i = 0; i < 7; ++i) at[i + 1] = (at[i] + sz[i] + 255) & ~(size_t)255;
    ORBFE_HIP(m->pin_in.ensure(at[7]));
    ORBFE_HIP(m->b[0].ensure(at[7]));
    ORBFE_HIP(m->b[2].ensure((size_t)nq * 4));
    ORBFE_HIP(m->b[3].ensure((size_t)nq * PJ_LC * 2));
    ORBFE_HIP(m->b[4].ensure((size_t)(nq + 1) * 4));
    ORBFE_HIP(m->b[5].ensure(std::max<size_t>((size_t)cap, 1) * 4));
    for (int i = 0; i < 7; ++i)
        if (sz[i]) memcpy((char *)m->pin_in.p + at[i], src[i], sz[i]);
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, m->pin_in.p, at[7], hipMemcpyHostToDevice, st));
    const char *din = (const char *)m->b[0].p;
    ProjArgs a;
    memset(&a, 0, sizeof(a));
    a.descF = (const uint8_t *)(din + at[0]);
    a.xyF = (const float *)(din + at[1]);
    a.octF = (const int32_t *)(din + at[2]);
    a.nF = nF; a.xs = 2; a.os = 1;
    a.cell_off = (const uint32_t *)(din + at[3]);
    a.cell_idx = (const uint32_t *)(din + at[4]);
    a.minx = minx; a.miny = miny; a.gwi = gw_inv; a.ghi = gh_inv;
    a.q = (const orbfe_proj_query *)(din + at[5]);
    a.qdesc = (const uint8_t *)(din + at[6]);
    a.nq = nq;
    a.cnt = (uint32_t *)m->b[2].p;
    a.lcnt = (uint16_t *)m->b[3].p;
    a.off = (uint32_t *)m->b[4].p;
    a.ent = (uint32_t *)m->b[5].p;
    a.ent_cap = (uint32_t)cap;
    const int ngrp = (nq * PJ_LC + 255) / 256;
    hipLaunchKernelGGL(k_proj_count, dim3(ngrp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, (const uint32_t *)a.cnt, nq, a.off);
    hipLaunchKernelGGL(k_proj_fill, dim3(ngrp), dim3(256), 0, st, a);   // writes nothing when the total exceeds cap
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpyAsync(off, a.off, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    if (off[nq] > (uint32_t)cap) { orbfe_set_error("orbfe_window_distances: %u entries needed, cap %d", off[nq], cap); return ORBFE_ERR_CAP; }
    if (off[nq] > 0) {
        ORBFE_HIP(hipMemcpyAsync(ent, a.ent, (size_t)off[nq] * 4, hipMemcpyDeviceToHost, st));
        ORBFE_HIP(hipStreamSynchronize(st));
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_search_by_projection(orbfe_matcher *m, const uint8_t *descF, const float *xyF, const int32_t *octF,
                                                   int32_t nF, const uint32_t *cell_off, const uint32_t *cell_idx, float minx,
                                                   float miny, float gw_inv, float gh_inv, const float *uRight,
                                                   const uint8_t *blocked, const orbfe_proj_query *q, const uint8_t *qdesc,
                                                   int32_t nq, int32_t th, float nnratio, int32_t ratio_rule, int32_t *match,
                                                   int32_t *best, int32_t *second)
{
    return orbfe_search_by_projection_chi2(m, descF, xyF, octF, nF, cell_off, cell_idx, minx, miny, gw_inv, gh_inv, uRight, blocked,
                                           nullptr, 0, q, qdesc, nq, th, nnratio, ratio_rule, match, best, second);
}

// ---------------------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:827-1012, LocalMapping::CreateNewMapPoints), the
// matching core.  The reference's loop never sets vbMatched2, so the features of keyframe 1 are independent: one thread per
// keyframe-1 feature scans its vocabulary node's features of keyframe 2 in FeatureVector order -- Hamming first, then the
// epipole gate and CheckDistEpipolarLine (:175-196) in the reference's float operation order -- and keeps the smallest
// distance, the LAST in order on ties (`dist > bestDist` skips, an equal distance takes over).
// ---------------------------------------------------------------------------------------------------
struct TriArgs {
    const uint8_t *desc1, *desc2;
    const float *xy1, *xy2;
    const int32_t *oct2;
    const uint8_t *elig1, *stereo1, *elig2, *stereo2;
    const int32_t *range1;      // [n1][2]: the node's slice of idx2 for every keyframe-1 feature, (0, 0) = none
    const uint32_t *idx2;
    float F[9], ex, ey;
    const float *scale2, *sigma2_2;
    int32_t n1, th_low;
    int32_t *match12;
};

__global__ __launch_bounds__(256) void k_triangulation(TriArgs a)
{
    const int f1 = blockIdx.x * 256 + threadIdx.x;
    if (f1 >= a.n1) return;
    int best = -1;
    const int lo = a.range1[2 * f1], hi = a.range1[2 * f1 + 1];
    if (hi > lo && a.elig1[f1]) {
        Desc8 d1;
        const uint32_t *p = (const uint32_t *)(a.desc1 + (int64_t)f1 * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) d1.w[k] = p[k];
        const bool st1 = a.stereo1[f1] != 0;
        const float x1 = a.xy1[2 * f1], y1 = a.xy1[2 * f1 + 1];
        // :175-182: a = kp1.x * F(0,0) + kp1.y * F(1,0) + F(2,0), every operation rounded separately
        const float ea = __fadd_rn(__fadd_rn(__fmul_rn(x1, a.F[0]), __fmul_rn(y1, a.F[3])), a.F[6]);
        const float eb = __fadd_rn(__fadd_rn(__fmul_rn(x1, a.F[1]), __fmul_rn(y1, a.F[4])), a.F[7]);
        const float ec = __fadd_rn(__fadd_rn(__fmul_rn(x1, a.F[2]), __fmul_rn(y1, a.F[5])), a.F[8]);
        const float den = __fadd_rn(__fmul_rn(ea, ea), __fmul_rn(eb, eb));
        int bestDist = a.th_low;
        for (int i2 = lo; i2 < hi; ++i2) {
            const uint32_t f2 = a.idx2[i2];
            if (!a.elig2[f2]) continue;
            const int dist = hamming8(d1, (const uint32_t *)(a.desc2 + (int64_t)f2 * 32));
            if (dist > a.th_low || dist > bestDist) continue;   // :895
            const float x2 = a.xy2[2 * f2], y2 = a.xy2[2 * f2 + 1];
            const int o2 = a.oct2[f2];
            if (!st1 && !a.stereo2[f2]) {                         // :900-907
                const float dx = __fsub_rn(a.ex, x2), dy = __fsub_rn(a.ey, y2);
                if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.f, a.scale2[o2])) continue;
            }
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(ea, x2), __fmul_rn(eb, y2)), ec);
            if (den == 0.f) continue;
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if ((double)dsqr < __dmul_rn(3.84, (double)a.sigma2_2[o2])) {   // :195
                best = (int)f2;
                bestDist = dist;
            }
        }
    }
    a.match12[f1] = best;
}

extern "C" orbfe_status orbfe_search_for_triangulation(orbfe_matcher *m, const uint8_t *desc1, const float *xy1, const uint8_t *elig1,
                                                       const uint8_t *stereo1, int32_t n1, const uint32_t *node1, const uint32_t *off1,
                                                       const uint32_t *idx1, int32_t nn1, const uint8_t *desc2, const float *xy2,
                                                       const int32_t *oct2, const uint8_t *elig2, const uint8_t *stereo2, int32_t n2,
                                                       const uint32_t *node2, const uint32_t *off2, const uint32_t *idx2, int32_t nn2,
                                                       const float F12[9], float ex, float ey, const float *scale_factors2,
                                                       const float *level_sigma2_2, int32_t nlevels2, int32_t th_low, int32_t *match12)
{
    if (!m || n1 < 0 || n2 < 0 || nn1 < 0 || nn2 < 0 || nlevels2 < 1 || !F12 || !scale_factors2 || !level_sigma2_2 ||
        (n1 > 0 && (!desc1 || !xy1 || !elig1 || !stereo1 || !match12)) || (n2 > 0 && (!desc2 || !xy2 || !oct2 || !elig2 || !stereo2)) ||
        (nn1 > 0 && (!node1 || !off1 || !idx1)) || (nn2 > 0 && (!node2 || !off2 || !idx2))) {
        orbfe_set_error("bad argument to orbfe_search_for_triangulation");
        return ORBFE_ERR_ARG;
    }
    if (n1 == 0) return ORBFE_OK;
    for (int i = 0; i < n2; ++i)
        if (oct2[i] < 0 || oct2[i] >= nlevels2) { orbfe_set_error("keyframe-2 octave out of range"); return ORBFE_ERR_ARG; }
    // the merge walk over the two FeatureVectors (:849-964) on the host: every keyframe-1 feature learns its node's slice of idx2
    std::vector<int32_t> range((size_t)n1 * 2, 0);
    const uint32_t total2 = nn2 > 0 ? off2[nn2] : 0;
    for (uint32_t k = 0; k < total2; ++k)
        if (idx2[k] >= (uint32_t)n2) { orbfe_set_error("FeatureVector 2 index out of range"); return ORBFE_ERR_ARG; }
    {
        int a = 0, b = 0;
        while (a < nn1 && b < nn2) {
            if (node1[a] == node2[b]) {
                for (uint32_t k = off1[a]; k < off1[a + 1]; ++k) {
                    if (idx1[k] >= (uint32_t)n1) { orbfe_set_error("FeatureVector 1 index out of range"); return ORBFE_ERR_ARG; }
                    range[2 * (size_t)idx1[k]] = (int32_t)off2[b];
                    range[2 * (size_t)idx1[k] + 1] = (int32_t)off2[b + 1];
                }
                ++a;
                ++b;
            } else if (node1[a] < node2[b]) ++a;
            else ++b;
        }
    }
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));
    const size_t sz[11] = {(size_t)n1 * 32, (size_t)n1 * 8, (size_t)n1, (size_t)n1, (size_t)n1 * 8, (size_t)n2 * 32, (size_t)n2 * 8, (size_t)n2 * 4,
                           (size_t)n2, (size_t)n2, (size_t)total2 * 4};
    const void *src[11] = {desc1, xy1, elig1, stereo1, range.data(), desc2, xy2, oct2, elig2, stereo2, idx2};
    size_t at[13];
    at[0] = 0;
    for (int i = 0; i < 11; ++i) at[i + 1] = (at[i] + sz[i] + 255) & ~(size_t)255;
    at[12] = at[11] + (((size_t)nlevels2 * 8 + 255) & ~(size_t)255);
    ORBFE_HIP(m->pin_in.ensure(at[12]));
    ORBFE_HIP(m->pin_out.ensure((size_t)n1 * 4));
    ORBFE_HIP(m->b[0].ensure(at[12]));
    ORBFE_HIP(m->b[1].ensure((size_t)n1 * 4));
    for (int i = 0; i < 11; ++i)
        if (sz[i]) memcpy((char *)m->pin_in.p + at[i], src[i], sz[i]);
    memcpy((char *)m->pin_in.p + at[11], scale_factors2, (size_t)nlevels2 * 4);
    memcpy((char *)m->pin_in.p + at[11] + (size_t)nlevels2 * 4, level_sigma2_2, (size_t)nlevels2 * 4);
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, m->pin_in.p, at[12], hipMemcpyHostToDevice, st));
    const char *d = (const char *)m->b[0].p;
    TriArgs a;
    a.desc1 = (const uint8_t *)(d + at[0]);
    a.xy1 = (const float *)(d + at[1]);
    a.elig1 = (const uint8_t *)(d + at[2]);
    a.stereo1 = (const uint8_t *)(d + at[3]);
    a.range1 = (const int32_t *)(d + at[4]);
    a.desc2 = (const uint8_t *)(d + at[5]);
    a.xy2 = (const float *)(d + at[6]);
    a.oct2 = (const int32_t *)(d + at[7]);
    a.elig2 = (const uint8_t *)(d + at[8]);
    a.stereo2 = (const uint8_t *)(d + at[9]);
    a.idx2 = (const uint32_t *)(d + at[10]);
    a.scale2 = (const float *)(d + at[11]);
    a.sigma2_2 = a.scale2 + nlevels2;
    for (int k = 0; k < 9; ++k) a.F[k] = F12[k];
    a.ex = ex; a.ey = ey;
    a.n1 = n1; a.th_low = th_low;
    a.match12 = (int32_t *)m->pin_out.p;   // page-locked and mapped: the kernel stores its 4 n1 result bytes there, no copy back
    hipLaunchKernelGGL(k_triangulation, dim3((n1 + 255) / 256), dim3(256), 0, st, a);
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipStreamSynchronize(st));
    memcpy(match12, m->pin_out.p, (size_t)n1 * 4);
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// SURVEY 8(f).4  MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:284-345), batched over map points.
// One wave per map point: its observed descriptors are staged in LDS, lane i owns row i of the distance matrix and
// finds that row's median -- element (int)(0.5 * (N - 1)) of the sorted row, self distance 0 included (:332-334) --
// by bisection on the value range [0, 256] with the row recomputed from LDS (no N x N matrix, any N); the point's
// descriptor is the row with the least median, first on ties (:335-339).
// ---------------------------------------------------------------------------------------------------
#define DD_MAX_OBS 1024

__global__ __launch_bounds__(64) void k_distinctive(const uint8_t *__restrict__ pool, const uint32_t *__restrict__ off,
                                                    const uint32_t *__restrict__ idx, int32_t *__restrict__ best_idx,
                                                    int32_t *__restrict__ median, int max_obs)
{
    extern __shared__ uint4 s_obs[];  // [max_obs][2]
    const int p = blockIdx.x, lane = threadIdx.x;
    const uint32_t o0 = off[p];
    const int n = (int)(off[p + 1] - o0);
    if (n <= 0 || n > max_obs) {  // no observation: -1; more than the LDS was sized for (device entry point only): -2
        if (lane == 0) { best_idx[p] = n <= 0 ? -1 : -2; median[p] = n <= 0 ? -1 : -2; }
        return;
    }
    for (int t = lane; t < 2 * n; t += 64) s_obs[t] = ((const uint4 *)pool)[(size_t)idx[o0 + (t >> 1)] * 2 + (t & 1)];
    __syncthreads();
    const int k = (int)(0.5 * (n - 1));
    uint32_t bestkey = 0xFFFFFFFFu;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane, ic = min(i, n - 1);
        const uint4 a0 = s_obs[2 * ic], a1 = s_obs[2 * ic + 1];
        int lo = 0, hi = 256;
        for (int it = 0; it < 9; ++it) {  // 257 values: 9 halvings; lanes whose interval closed early idle harmlessly
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                const uint4 b0 = s_obs[2 * j], b1 = s_obs[2 * j + 1];  // same address in every lane: LDS broadcast
                const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                              __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
                cnt += d <= mid ? 1 : 0;
            }
            if (lo < hi) {
                if (cnt >= k + 1) hi = mid;
                else lo = mid + 1;
            }
        }
        if (i < n) bestkey = min(bestkey, ((uint32_t)lo << 16) | (uint32_t)i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bestkey = min(bestkey, (uint32_t)__shfl_xor((int)bestkey, o, 64));
    if (lane == 0) {
        best_idx[p] = (int32_t)(bestkey & 0xFFFFu);
        median[p] = (int32_t)(bestkey >> 16);
    }
}

extern "C" orbfe_status orbfe_distinctive_descriptors(orbfe_matcher *m, const uint8_t *pool, int32_t npool,
                                                      const uint32_t *off, const uint32_t *idx, int32_t npoints,
                                                      int32_t *best_idx, int32_t *median)
{
    if (!m || npool < 0 || npoints < 0 || (npoints > 0 && (!off || !best_idx || !median))) {
        orbfe_set_error("bad argument to orbfe_distinctive_descriptors");
        return ORBFE_ERR_ARG;
    }
    if (npoints == 0) return ORBFE_OK;
    uint32_t maxn = 0;
    for (int i = 0; i < npoints; ++i) {
        if (off[i + 1] < off[i]) { orbfe_set_error("CSR offsets must not decrease"); return ORBFE_ERR_ARG; }
        maxn = std::max(maxn, off[i + 1] - off[i]);
    }
    if (maxn > DD_MAX_OBS) {
        orbfe_set_error("a map point with %u observations exceeds the supported %d", maxn, DD_MAX_OBS);
        return ORBFE_ERR_ARG;
    }
    const size_t nc = off[npoints];
    if (nc > 0 && (!idx || !pool)) { orbfe_set_error("null pool / idx"); return ORBFE_ERR_ARG; }
    for (size_t k = 0; k < nc; ++k)
        if (idx[k] >= (uint32_t)npool) { orbfe_set_error("observation index out of range"); return ORBFE_ERR_ARG; }
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    ORBFE_HIP(m->b[0].ensure((size_t)std::max(npool, 1) * 32));
    ORBFE_HIP(m->b[2].ensure((size_t)(npoints + 1) * 4));
    ORBFE_HIP(m->b[3].ensure(std::max(nc, (size_t)1) * 4));
    ORBFE_HIP(m->b[4].ensure((size_t)npoints * 4));
    ORBFE_HIP(m->b[5].ensure((size_t)npoints * 4));
    if (npool > 0) ORBFE_HIP(hipMemcpyAsync(m->b[0].p, pool, (size_t)npool * 32, hipMemcpyHostToDevice, st));
    ORBFE_HIP(hipMemcpyAsync(m->b[2].p, off, (size_t)(npoints + 1) * 4, hipMemcpyHostToDevice, st));
    if (nc > 0) ORBFE_HIP(hipMemcpyAsync(m->b[3].p, idx, nc * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_distinctive, dim3(npoints), dim3(64), (size_t)std::max(maxn, 1u) * 32, st,
                       (const uint8_t *)m->b[0].p, (const uint32_t *)m->b[2].p, (const uint32_t *)m->b[3].p,
                       (int32_t *)m->b[4].p, (int32_t *)m->b[5].p, (int)std::max(maxn, 1u));
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpyAsync(best_idx, m->b[4].p, (size_t)npoints * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(median, m->b[5].p, (size_t)npoints * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_distinctive_descriptors_device(orbfe_matcher *m, const uint8_t *d_pool, const uint32_t *d_off,
                                                             const uint32_t *d_idx, int32_t npoints, int32_t max_obs,
                                                             int32_t *d_best_idx, int32_t *d_median, void *stream)
{
    if (!m || npoints < 0 || max_obs < 1 || max_obs > DD_MAX_OBS || (npoints > 0 && (!d_pool || !d_off || !d_idx || !d_best_idx || !d_median))) {
        orbfe_set_error("bad argument to orbfe_distinctive_descriptors_device (max_obs 1..%d)", DD_MAX_OBS);
        return ORBFE_ERR_ARG;
    }
    if (npoints == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipLaunchKernelGGL(k_distinctive, dim3(npoints), dim3(64), (size_t)max_obs * 32, (hipStream_t)stream, d_pool, d_off, d_idx, d_best_idx,
                       d_median, max_obs);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// SURVEY 8(f).2  Frame::ComputeStereoMatches (src/Frame.cc:642-846).
// k_stereo_match: one wave per left keypoint.  Phase A scans the right keypoints in index order -- the reference's
//   per-row candidate lists are the right keypoints whose band [floor(y - r), ceil(y + r)], r = 2 * scale[octave],
//   holds the left keypoint's row, in push_back (= index) order, so the band test replaces the row table -- and keeps
//   the first smallest Hamming distance below TH_HIGH.  Phase B is the 11 x 11 SAD search over 11 shifts on the
//   device-resident pyramids of the two extractors (integer sums: |(l - cl) - (r - cr)| is exact in the reference's
//   float arithmetic), the parabola fit and the disparity / depth bookkeeping, every float operation rounded as there.
// k_stereo_filter: the final outlier rejection (:831-845): median of the kept SAD distances by rank counting.
// ---------------------------------------------------------------------------------------------------
struct StereoArgs {
    OrbPyrView L, R;
    float mbf, mb;
};

// Batched form: grid.y = frame pair f of the two extractors' last batches; keypoints / descriptors / results of frame f
// start at slot f * cap and the counts come from the extractors' own count arrays (d_nL / d_nR, clamped to cap).  The
// single-pair host entry point passes cap = 0 and null count arrays.
__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs a, const orbfe_keypoint *__restrict__ kpsL,
                                                      const uint8_t *__restrict__ descL, int nL,
                                                      const orbfe_keypoint *__restrict__ kpsR,
                                                      const uint8_t *__restrict__ descR, int nR,
                                                      float *__restrict__ uRight, float *__restrict__ depth,
                                                      int32_t *__restrict__ sad, int cap,
                                                      const int32_t *__restrict__ d_nL, const int32_t *__restrict__ d_nR)
{
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.y;
    if (d_nL) {
        nL = min(d_nL[f], cap);
        nR = min(d_nR[f], cap);
        const int64_t o = (int64_t)f * cap;
        kpsL += o; descL += o * 32; kpsR += o; descR += o * 32;
        uRight += o; depth += o; sad += o;
    }
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= nL) return;
    const orbfe_keypoint kL = kpsL[iL];
    const int levelL = kL.octave;
    const float uL = kL.x, vL = kL.y;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    if ((unsigned)levelL >= (unsigned)a.L.nlevels) {  // not an extractor output: no match (the host entry point rejects it)
        if (lane == 0) { uRight[iL] = -1.0f; depth[iL] = -1.0f; sad[iL] = -1; }
        return;
    }
    const float minD = 0.f, maxD = __fdiv_rn(a.mbf, a.mb);
    const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
    const int rowL = (int)vL;
    uint32_t best = 0xFFFFFFFFu;  // dist << 20 | iR
    if (!(maxU < 0)) {
        Desc8 dl;
        {
            const uint32_t *p = (const uint32_t *)(descL + (int64_t)iL * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) dl.w[i] = p[i];
        }
        for (int iR = lane; iR < nR; iR += 64) {
            const orbfe_keypoint kR = kpsR[iR];
            if ((unsigned)kR.octave >= (unsigned)a.R.nlevels) continue;
            const float r = __fmul_rn(2.0f, a.R.scale[kR.octave]);
            const int maxr = (int)ceilf(__fadd_rn(kR.y, r)), minr = (int)floorf(__fsub_rn(kR.y, r));
            if (rowL < minr || rowL > maxr) continue;
            if (kR.octave < levelL - 1 || kR.octave > levelL + 1) continue;
            if (kR.x >= minU && kR.x <= maxU) {
                const int d = hamming8(dl, (const uint32_t *)(descR + (int64_t)iR * 32));
                if (d < ORBFE_TH_HIGH) best = min(best, ((uint32_t)d << 20) | (uint32_t)iR);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
    const int bestDist = (int)(best >> 20), bestIdxR = (int)(best & 0xFFFFFu);
    if (best != 0xFFFFFFFFu && bestDist < (ORBFE_TH_HIGH + ORBFE_TH_LOW) / 2) {
        const float uR0 = kpsR[bestIdxR].x;
        const float sf = a.L.inv_scale[levelL];
        const float scaleduL = roundf(__fmul_rn(kL.x, sf)), scaledvL = roundf(__fmul_rn(kL.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, Ls = 5;
        const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)Ls), (float)w);
        const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)Ls), (float)w), 1.0f);
        if (!(iniu < 0 || endu >= (float)a.R.w[levelL])) {
            const uint8_t *imL = a.L.ptr[levelL] + f * a.L.fstride[levelL], *imR = a.R.ptr[levelL] + f * a.R.fstride[levelL];
            const int pl = a.L.pitch[levelL], pr = a.R.pitch[levelL];
            const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
            const int cl = imL[cv * pl + cu];
            // the lane's two window pixels (121 = 64 + 57)
            const int p0 = lane, p1 = lane + 64;
            const int dy0 = p0 / 11 - w, dx0 = p0 % 11 - w, dy1 = min(p1, 120) / 11 - w, dx1 = min(p1, 120) % 11 - w;
            const int l0 = imL[(cv + dy0) * pl + cu + dx0] - cl, l1 = imL[(cv + dy1) * pl + cu + dx1] - cl;
            int bestSad = 0x7fffffff, bestinc = 0;
            float vDists[11];
#pragma unroll
            for (int inc = -5; inc <= 5; ++inc) {
                const int crc = imR[cv * pr + cr + inc];
                const int r0 = imR[(cv + dy0) * pr + cr + inc + dx0] - crc, r1 = imR[(cv + dy1) * pr + cr + inc + dx1] - crc;
                int acc = abs(l0 - r0) + (p1 < 121 ? abs(l1 - r1) : 0);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
                const float dist = (float)acc;
                if (dist < (float)bestSad) {  // float against int, as :783
                    bestSad = (int)dist;
                    bestinc = inc;
                }
                vDists[inc + 5] = dist;
            }
            if (bestinc != -Ls && bestinc != Ls) {
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int t = 1; t < 10; ++t)
                    if (t == bestinc + 5) { d1 = vDists[t - 1]; d2 = vDists[t]; d3 = vDists[t + 1]; }
                const float deltaR = __fdiv_rn(__fsub_rn(d1, d3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(a.L.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = 0.01f;                                           // float(0.01)
                            bestuR = __double2float_rn(__dsub_rn((double)uL, 0.01));    // double arithmetic, :821
                        }
                        out_d = __fdiv_rn(a.mbf, disparity);
                        out_u = bestuR;
                        out_sad = bestSad;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        uRight[iL] = out_u;
        depth[iL] = out_d;
        sad[iL] = out_sad;
    }
}

__global__ __launch_bounds__(1024) void k_stereo_filter(int nL, float *__restrict__ uRight, float *__restrict__ depth,
                                                        const int32_t *__restrict__ sad, int cap,
                                                        const int32_t *__restrict__ d_nL)
{
    __shared__ int s_n, s_median;
    const int tid = threadIdx.x;
    if (d_nL) {  // batched: one workgroup per frame pair
        nL = min(d_nL[blockIdx.x], cap);
        const int64_t o = (int64_t)blockIdx.x * cap;
        uRight += o; depth += o; sad += o;
    }
    if (tid == 0) { s_n = 0; s_median = -1; }
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < nL; i += 1024) cnt += sad[i] >= 0;
    if (cnt) atomicAdd(&s_n, cnt);
    __syncthreads();
    const int nv = s_n;
    if (nv == 0) return;  // the reference indexes an empty vector here (undefined): nothing to do
    const int target = nv / 2;  // position in the (distance, index)-sorted list
    for (int i = tid; i < nL; i += 1024) {
        const int di = sad[i];
        if (di < 0) continue;
        int rank = 0;
        for (int j = 0; j < nL; ++j) {
            const int dj = sad[j];
            rank += (dj >= 0 && (dj < di || (dj == di && j < i))) ? 1 : 0;
        }
        if (rank == target) s_median = di;
    }
    __syncthreads();
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), (float)s_median);
    for (int i = tid; i < nL; i += 1024) {
        const int di = sad[i];
        if (di >= 0 && !((float)di < thDist)) {
            uRight[i] = -1.0f;
            depth[i] = -1.0f;
        }
    }
}

extern "C" orbfe_status orbfe_stereo_matches(orbfe_matcher *m, orbfe_handle *left, orbfe_handle *right,
                                             const orbfe_keypoint *kpsL, const uint8_t *descL, int32_t nL,
                                             const orbfe_keypoint *kpsR, const uint8_t *descR, int32_t nR, float mbf,
                                             float mb, float *uRight, float *depth)
{
    if (!m || !left || !right || nL < 0 || nR < 0 || nR >= (1 << 20) || (nL > 0 && (!kpsL || !descL || !uRight || !depth)) ||
        (nR > 0 && (!kpsR || !descR))) {
        orbfe_set_error("bad argument to orbfe_stereo_matches");
        return ORBFE_ERR_ARG;
    }
    if (nL == 0) return ORBFE_OK;
    StereoArgs a;
    orbfe_status s = (orbfe_status)orbfe_internal_pyramid_view(left, 0, &a.L);
    if (s != ORBFE_OK) return s;
    s = (orbfe_status)orbfe_internal_pyramid_view(right, 0, &a.R);
    if (s != ORBFE_OK) return s;
    if (a.L.device != m->device || a.R.device != m->device || a.L.nlevels != a.R.nlevels) {
        orbfe_set_error("stereo: the two extractors and the matcher must share a device and a pyramid shape");
        return ORBFE_ERR_ARG;
    }
    for (int i = 0; i < nL; ++i)
        if (kpsL[i].octave < 0 || kpsL[i].octave >= a.L.nlevels) { orbfe_set_error("left octave out of range"); return ORBFE_ERR_ARG; }
    for (int i = 0; i < nR; ++i)
        if (kpsR[i].octave < 0 || kpsR[i].octave >= a.R.nlevels) { orbfe_set_error("right octave out of range"); return ORBFE_ERR_ARG; }
    a.mbf = mbf;
    a.mb = mb;
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    // the kernels read the two extractors' pyramids: behind their last calls, on whichever streams those ran
    s = (orbfe_status)orbfe_internal_order_after_last_call(left, st);
    if (s == ORBFE_OK) s = (orbfe_status)orbfe_internal_order_after_last_call(right, st);
    if (s != ORBFE_OK) return s;
    ORBFE_HIP(m->b[0].ensure((size_t)nL * sizeof(orbfe_keypoint)));
    ORBFE_HIP(m->b[1].ensure((size_t)nL * 32));
    ORBFE_HIP(m->b[2].ensure((size_t)std::max(nR, 1) * sizeof(orbfe_keypoint)));
    ORBFE_HIP(m->b[3].ensure((size_t)std::max(nR, 1) * 32));
    ORBFE_HIP(m->b[4].ensure((size_t)nL * 4));
    ORBFE_HIP(m->b[5].ensure((size_t)nL * 4));
    ORBFE_HIP(m->b[6].ensure((size_t)nL * 4));
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, kpsL, (size_t)nL * sizeof(orbfe_keypoint), hipMemcpyHostToDevice, st));
    ORBFE_HIP(hipMemcpyAsync(m->b[1].p, descL, (size_t)nL * 32, hipMemcpyHostToDevice, st));
    if (nR > 0) {
        ORBFE_HIP(hipMemcpyAsync(m->b[2].p, kpsR, (size_t)nR * sizeof(orbfe_keypoint), hipMemcpyHostToDevice, st));
        ORBFE_HIP(hipMemcpyAsync(m->b[3].p, descR, (size_t)nR * 32, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_stereo_match, dim3((nL + 3) / 4), dim3(256), 0, st, a, (const orbfe_keypoint *)m->b[0].p,
                       (const uint8_t *)m->b[1].p, nL, (const orbfe_keypoint *)m->b[2].p, (const uint8_t *)m->b[3].p, nR,
                       (float *)m->b[4].p, (float *)m->b[5].p, (int32_t *)m->b[6].p, 0, (const int32_t *)nullptr,
                       (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_stereo_filter, dim3(1), dim3(1024), 0, st, nL, (float *)m->b[4].p, (float *)m->b[5].p,
                       (const int32_t *)m->b[6].p, 0, (const int32_t *)nullptr);
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpyAsync(uRight, m->b[4].p, (size_t)nL * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipMemcpyAsync(depth, m->b[5].p, (size_t)nL * 4, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_stereo_matches_batch_device(orbfe_matcher *m, orbfe_handle *left, orbfe_handle *right,
                                                          const orbfe_keypoint *d_kpsL, const uint8_t *d_descL,
                                                          const int32_t *d_nL, const orbfe_keypoint *d_kpsR,
                                                          const uint8_t *d_descR, const int32_t *d_nR, int32_t cap,
                                                          int32_t nframes, float mbf, float mb, float *d_uRight,
                                                          float *d_depth, void *stream)
{
    if (!m || !left || !right || nframes < 0 || cap < 0 || cap >= (1 << 20) ||
        (nframes > 0 && cap > 0 && (!d_kpsL || !d_descL || !d_nL || !d_kpsR || !d_descR || !d_nR || !d_uRight || !d_depth))) {
        orbfe_set_error("bad argument to orbfe_stereo_matches_batch_device");
        return ORBFE_ERR_ARG;
    }
    if (nframes == 0 || cap == 0) return ORBFE_OK;
    StereoArgs a;
    orbfe_status s = (orbfe_status)orbfe_internal_pyramid_view(left, 0, &a.L);
    if (s != ORBFE_OK) return s;
    s = (orbfe_status)orbfe_internal_pyramid_view(right, 0, &a.R);
    if (s != ORBFE_OK) return s;
    if (a.L.device != m->device || a.R.device != m->device || a.L.nlevels != a.R.nlevels) {
        orbfe_set_error("stereo: the two extractors and the matcher must share a device and a pyramid shape");
        return ORBFE_ERR_ARG;
    }
    if (nframes > a.L.nframes || nframes > a.R.nframes) {
        orbfe_set_error("stereo: %d frame pairs asked, the extractors' last batches hold %d / %d frames", nframes, a.L.nframes,
                        a.R.nframes);
        return ORBFE_ERR_ARG;
    }
    a.mbf = mbf;
    a.mb = mb;
    MDeviceGuard g(m->device);
    hipStream_t st = (hipStream_t)stream;
    s = (orbfe_status)orbfe_internal_order_after_last_call(left, st);
    if (s == ORBFE_OK) s = (orbfe_status)orbfe_internal_order_after_last_call(right, st);
    if (s != ORBFE_OK) return s;
    ORBFE_HIP(scratch_acquire(m, st));
    ORBFE_HIP(m->b[6].ensure((size_t)nframes * cap * 4));  // SAD distances of the kept matches, read by the filter
    hipLaunchKernelGGL(k_stereo_match, dim3((cap + 3) / 4, nframes), dim3(256), 0, st, a, d_kpsL, d_descL, 0, d_kpsR, d_descR, 0,
                       d_uRight, d_depth, (int32_t *)m->b[6].p, cap, d_nL, d_nR);
    hipLaunchKernelGGL(k_stereo_filter, dim3(nframes), dim3(1024), 0, st, 0, d_uRight, d_depth, (const int32_t *)m->b[6].p, cap,
                       d_nL);
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(scratch_release(m, st));
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// SURVEY 8(f).3  DBoW2 TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) as called at
// src/Frame.cc:553 and src/KeyFrame.cc:82.  DBoW2 is not vendored by the reference; the algorithm is restated from the
// published one.
// k_bow_descend: thread per feature walks the tree, per level the child with the smallest Hamming distance (first on
//   ties); remembers the node at level L - levelsup; features whose word has weight 0 are dropped.
// k_bow_aggregate: one workgroup turns the per-feature (word, node, weight) into the two containers of the reference:
//   an LDS bitonic sort by (word, feature) gives std::map order, every first-of-its-word thread adds its weights in
//   feature order (doubles, the order `+=` ran in the reference), thread 0 forms the L1 norm in ascending word order,
//   then the same sort by (node, feature) gives the FeatureVector as the CSR orbfe_search_by_bow consumes.
// ---------------------------------------------------------------------------------------------------
#define BOW_MAX_FEATURES 8192

struct orbfe_vocabulary {
    int device = 0, nnodes = 0, L = 0;
    MDevBuf child_off, child_idx, node_desc, word_id, weight;
};

__global__ __launch_bounds__(256) void k_bow_descend(const uint32_t *__restrict__ child_off,
                                                     const uint32_t *__restrict__ child_idx,
                                                     const uint8_t *__restrict__ node_desc,
                                                     const uint32_t *__restrict__ word_id,
                                                     const double *__restrict__ weight, int nid_level,
                                                     const uint8_t *__restrict__ desc, int n,
                                                     int32_t *__restrict__ f_word, int32_t *__restrict__ f_node,
                                                     double *__restrict__ f_weight,
                                                     const int32_t *__restrict__ n_arr, int stride)
{
    // batched form: frame blockIdx.y owns `stride` slots of every array and holds n_arr[frame] features
    if (n_arr) {
        const int b = blockIdx.y;
        n = min(n_arr[b], stride);
        desc += (int64_t)b * stride * 32;
        f_word += (int64_t)b * stride;
        f_node += (int64_t)b * stride;
        f_weight += (int64_t)b * stride;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (n_arr && i >= n && i < stride) {  // padding slots carry "no word" so the buffers can be used as they are
        f_word[i] = -1;
        f_node[i] = -1;
        f_weight[i] = 0.0;
    }
    if (i >= n) return;
    Desc8 q;
    {
        const uint32_t *p = (const uint32_t *)(desc + (int64_t)i * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) q.w[k] = p[k];
    }
    uint32_t fin = 0, nid = 0;
    int level = 0;
    uint32_t c0 = child_off[0], c1 = child_off[1];
    do {  // child ids are larger than their parent's (checked at creation): the walk ends
        ++level;
        fin = child_idx[c0];
        int best = hamming8(q, (const uint32_t *)(node_desc + (int64_t)fin * 32));
        for (uint32_t c = c0 + 1; c < c1; ++c) {
            const uint32_t id = child_idx[c];
            const int d = hamming8(q, (const uint32_t *)(node_desc + (int64_t)id * 32));
            if (d < best) { best = d; fin = id; }
        }
        if (level == nid_level) nid = fin;
        c0 = child_off[fin];
        c1 = child_off[fin + 1];
    } while (c1 != c0);
    const double w = weight[fin];
    const bool keep = w > 0;
    f_word[i] = keep ? (int32_t)word_id[fin] : -1;
    f_node[i] = keep ? (int32_t)nid : -1;
    f_weight[i] = keep ? w : 0.0;
}

__device__ void bow_bitonic_sort(unsigned long long *key, int P, int tid)
{
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P; t += 1024) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const unsigned long long a = key[t], b = key[ixj];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { key[t] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

// exclusive position of every flagged element among P (each thread owns a contiguous chunk); returns the total
__device__ int bow_positions(const unsigned long long *key, int P, int tid, int *s_scan, int *pos_of_first_in_chunk)
{
    const int chunk = (P + 1023) / 1024, j0 = tid * chunk, j1 = min(j0 + chunk, P);
    int cnt = 0;
    for (int j = j0; j < j1; ++j) {
        const unsigned long long kj = key[j];
        if (kj != ~0ull && (j == 0 || (key[j - 1] >> 32) != (kj >> 32))) ++cnt;
    }
    s_scan[tid] = cnt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? s_scan[tid - d] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    *pos_of_first_in_chunk = s_scan[tid] - cnt;
    const int total = s_scan[1023];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(1024) void k_bow_aggregate(int n, int P, const int32_t *__restrict__ f_word,
                                                        const int32_t *__restrict__ f_node,
                                                        const double *__restrict__ f_weight,
                                                        uint32_t *__restrict__ bow_id, double *__restrict__ bow_val,
                                                        uint32_t *__restrict__ fv_node, uint32_t *__restrict__ fv_off,
                                                        uint32_t *__restrict__ fv_idx, int32_t *__restrict__ counts,
                                                        const int32_t *__restrict__ n_arr, int stride)
{
    if (n_arr) {  // batched form: one workgroup per frame, `stride` slots per array (stride + 1 for fv_off, 4 counts)
        const int b = blockIdx.x;
        n = min(n_arr[b], stride);
        f_word += (int64_t)b * stride;
        f_node += (int64_t)b * stride;
        f_weight += (int64_t)b * stride;
        bow_id += (int64_t)b * stride;
        bow_val += (int64_t)b * stride;
        fv_node += (int64_t)b * stride;
        fv_off += (int64_t)b * (stride + 1);
        fv_idx += (int64_t)b * stride;
        counts += (int64_t)b * 4;
    }
    extern __shared__ unsigned long long s_key[];  // [P] keys, then [P] doubles
    double *s_val = (double *)(s_key + P);
    __shared__ int s_scan[1024];
    __shared__ double s_norm;
    const int tid = threadIdx.x;
    const int chunk = (P + 1023) / 1024, j0 = tid * chunk, j1 = min(j0 + chunk, P);
    // ---- BowVector ----
    for (int i = tid; i < P; i += 1024)
        s_key[i] = (i < n && f_word[i] >= 0) ? (((unsigned long long)(uint32_t)f_word[i] << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    bow_bitonic_sort(s_key, P, tid);
    int pos;
    const int nbow = bow_positions(s_key, P, tid, s_scan, &pos);
    for (int j = j0; j < j1; ++j) {
        const unsigned long long kj = s_key[j];
        if (kj != ~0ull && (j == 0 || (s_key[j - 1] >> 32) != (kj >> 32))) {
            double v = 0.0;  // map[word] += weight, in feature order
            for (int e = j; e < P && (s_key[e] >> 32) == (kj >> 32); ++e) v = __dadd_rn(v, f_weight[(uint32_t)s_key[e]]);
            bow_id[pos] = (uint32_t)(kj >> 32);
            s_val[pos] = v;
            ++pos;
        }
    }
    __syncthreads();
    if (tid == 0) {  // BowVector::normalize(L1): ascending word order
        double norm = 0.0;
        for (int o = 0; o < nbow; ++o) norm = __dadd_rn(norm, fabs(s_val[o]));
        s_norm = norm;
    }
    __syncthreads();
    for (int o = tid; o < nbow; o += 1024) bow_val[o] = s_norm > 0.0 ? __ddiv_rn(s_val[o], s_norm) : s_val[o];
    __syncthreads();
    // ---- FeatureVector ----
    for (int i = tid; i < P; i += 1024)
        s_key[i] = (i < n && f_node[i] >= 0) ? (((unsigned long long)(uint32_t)f_node[i] << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    bow_bitonic_sort(s_key, P, tid);
    const int nfv = bow_positions(s_key, P, tid, s_scan, &pos);
    int m = 0;
    for (int j = j0; j < j1; ++j) {
        const unsigned long long kj = s_key[j];
        if (kj == ~0ull) continue;
        fv_idx[j] = (uint32_t)kj;
        if (j == 0 || (s_key[j - 1] >> 32) != (kj >> 32)) {
            fv_node[pos] = (uint32_t)(kj >> 32);
            fv_off[pos] = (uint32_t)j;
            ++pos;
        }
        ++m;
    }
    s_scan[tid] = m;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int t = 0; t < 1024; ++t) tot += s_scan[t];
        fv_off[nfv] = (uint32_t)tot;
        counts[0] = nbow;
        counts[1] = nfv;
        counts[2] = tot;
    }
}

extern "C" orbfe_status orbfe_vocabulary_create(int32_t device, int32_t nnodes, const uint32_t *child_off,
                                                const uint32_t *child_idx, const uint8_t *node_desc,
                                                const uint32_t *word_id, const double *weight, int32_t L,
                                                orbfe_vocabulary **out)
{
    if (!out || nnodes < 1 || !child_off || !node_desc || !word_id || !weight || L < 1) {
        orbfe_set_error("bad argument to orbfe_vocabulary_create");
        return ORBFE_ERR_ARG;
    }
    *out = nullptr;
    const uint32_t nc = child_off[nnodes];
    if (child_off[0] != 0 || (nc > 0 && !child_idx)) { orbfe_set_error("vocabulary: bad child CSR"); return ORBFE_ERR_ARG; }
    for (int i = 0; i < nnodes; ++i) {
        if (child_off[i + 1] < child_off[i]) { orbfe_set_error("vocabulary: child offsets must not decrease"); return ORBFE_ERR_ARG; }
        for (uint32_t c = child_off[i]; c < child_off[i + 1]; ++c)
            if (child_idx[c] <= (uint32_t)i || child_idx[c] >= (uint32_t)nnodes) {
                orbfe_set_error("vocabulary: child ids must be larger than their parent's id and < nnodes");
                return ORBFE_ERR_ARG;
            }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        orbfe_set_error("no HIP device visible; liborbfe has no CPU fallback");
        return ORBFE_ERR_NODEVICE;
    }
    if (device < 0) device = 0;
    if (device >= ndev) { orbfe_set_error("device %d out of range", device); return ORBFE_ERR_ARG; }
    orbfe_vocabulary *v = new (std::nothrow) orbfe_vocabulary();
    if (!v) return ORBFE_ERR_NOMEM;
    v->device = device;
    v->nnodes = nnodes;
    v->L = L;
    MDeviceGuard g(device);
    auto up = [&](MDevBuf &b, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(std::max(bytes, (size_t)4));
        if (e == hipSuccess && bytes) e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(v->child_off, child_off, (size_t)(nnodes + 1) * 4);
    if (e == hipSuccess) e = up(v->child_idx, child_idx, (size_t)nc * 4);
    if (e == hipSuccess) e = up(v->node_desc, node_desc, (size_t)nnodes * 32);
    if (e == hipSuccess) e = up(v->word_id, word_id, (size_t)nnodes * 4);
    if (e == hipSuccess) e = up(v->weight, weight, (size_t)nnodes * 8);
    if (e != hipSuccess) {
        orbfe_set_error("vocabulary upload failed: %s", hipGetErrorString(e));
        orbfe_vocabulary_destroy(v);
        return ORBFE_ERR_HIP;
    }
    *out = v;
    return ORBFE_OK;
}

extern "C" void orbfe_vocabulary_destroy(orbfe_vocabulary *v)
{
    if (!v) return;
    MDeviceGuard g(v->device);
    MDevBuf *bufs[] = {&v->child_off, &v->child_idx, &v->node_desc, &v->word_id, &v->weight};
    for (MDevBuf *b : bufs) b->release();
    delete v;
}

extern "C" orbfe_status orbfe_bow_transform(orbfe_matcher *m, const orbfe_vocabulary *v, const uint8_t *desc, int32_t n,
                                            int32_t levelsup, int32_t *f_word, int32_t *f_node, double *f_weight,
                                            uint32_t *bow_id, double *bow_val, int32_t *nbow, uint32_t *fv_node,
                                            uint32_t *fv_off, uint32_t *fv_idx, int32_t *nfv)
{
    if (!m || !v || n < 0 || n > BOW_MAX_FEATURES || !nbow || !nfv || !fv_off ||
        (n > 0 && (!desc || !bow_id || !bow_val || !fv_node || !fv_idx))) {
        orbfe_set_error("bad argument to orbfe_bow_transform (at most %d features per call)", BOW_MAX_FEATURES);
        return ORBFE_ERR_ARG;
    }
    if (v->device != m->device) { orbfe_set_error("vocabulary and matcher are on different devices"); return ORBFE_ERR_ARG; }
    *nbow = 0;
    *nfv = 0;
    fv_off[0] = 0;
    if (n == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipStream_t st = m->stream;
    ORBFE_HIP(scratch_acquire(m, st));  // a device-buffer call on another stream may still be using the scratch blocks
    int P = 2;
    while (P < n) P <<= 1;
    ORBFE_HIP(m->b[0].ensure((size_t)n * 32));
    ORBFE_HIP(m->b[1].ensure((size_t)n * 4));   // f_word
    ORBFE_HIP(m->b[2].ensure((size_t)n * 4));   // f_node
    ORBFE_HIP(m->b[3].ensure((size_t)n * 8));   // f_weight
    ORBFE_HIP(m->b[4].ensure((size_t)n * 4));   // bow_id
    ORBFE_HIP(m->b[5].ensure((size_t)n * 8));   // bow_val
    ORBFE_HIP(m->b[6].ensure((size_t)n * 4));   // fv_node
    ORBFE_HIP(m->b[7].ensure((size_t)(n + 1) * 4));
    ORBFE_HIP(m->b[8].ensure((size_t)n * 4));   // fv_idx
    ORBFE_HIP(m->b[9].ensure(16));
    ORBFE_HIP(hipMemcpyAsync(m->b[0].p, desc, (size_t)n * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bow_descend, dim3((n + 255) / 256), dim3(256), 0, st, (const uint32_t *)v->child_off.p,
                       (const uint32_t *)v->child_idx.p, (const uint8_t *)v->node_desc.p, (const uint32_t *)v->word_id.p,
                       (const double *)v->weight.p, v->L - levelsup, (const uint8_t *)m->b[0].p, n, (int32_t *)m->b[1].p,
                       (int32_t *)m->b[2].p, (double *)m->b[3].p, (const int32_t *)nullptr, 0);
    const size_t lds = (size_t)P * 16;
    // The dynamic-LDS limit is a process-wide, per-kernel attribute: every caller sets it to the SAME value -- all of the
    // CU's LDS that the kernel's static allocation leaves -- so concurrent matchers can never lower it under one another.
    hipFuncAttributes fa;
    ORBFE_HIP(hipFuncGetAttributes(&fa, (const void *)k_bow_aggregate));
    const size_t lds_max = (size_t)ORBFE_LDS_MAX - fa.sharedSizeBytes;
    if (lds > lds_max) { orbfe_set_error("orbfe_bow_transform: %d features need more than the CU's LDS", n); return ORBFE_ERR_SIZE; }
    if (lds > 64 * 1024)
        ORBFE_HIP(hipFuncSetAttribute((const void *)k_bow_aggregate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    hipLaunchKernelGGL(k_bow_aggregate, dim3(1), dim3(1024), lds, st, n, P, (const int32_t *)m->b[1].p,
                       (const int32_t *)m->b[2].p, (const double *)m->b[3].p, (uint32_t *)m->b[4].p, (double *)m->b[5].p,
                       (uint32_t *)m->b[6].p, (uint32_t *)m->b[7].p, (uint32_t *)m->b[8].p, (int32_t *)m->b[9].p,
                       (const int32_t *)nullptr, 0);
    ORBFE_HIP(hipGetLastError());
    int32_t counts[3] = {0, 0, 0};
    ORBFE_HIP(hipMemcpyAsync(counts, m->b[9].p, 12, hipMemcpyDeviceToHost, st));
    if (f_word) ORBFE_HIP(hipMemcpyAsync(f_word, m->b[1].p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (f_node) ORBFE_HIP(hipMemcpyAsync(f_node, m->b[2].p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (f_weight) ORBFE_HIP(hipMemcpyAsync(f_weight, m->b[3].p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    ORBFE_HIP(hipStreamSynchronize(st));
    *nbow = counts[0];
    *nfv = counts[1];
    if (counts[0] > 0) {
        ORBFE_HIP(hipMemcpy(bow_id, m->b[4].p, (size_t)counts[0] * 4, hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(bow_val, m->b[5].p, (size_t)counts[0] * 8, hipMemcpyDeviceToHost));
    }
    ORBFE_HIP(hipMemcpy(fv_off, m->b[7].p, (size_t)(counts[1] + 1) * 4, hipMemcpyDeviceToHost));
    if (counts[1] > 0) ORBFE_HIP(hipMemcpy(fv_node, m->b[6].p, (size_t)counts[1] * 4, hipMemcpyDeviceToHost));
    if (counts[2] > 0) ORBFE_HIP(hipMemcpy(fv_idx, m->b[8].p, (size_t)counts[2] * 4, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}


// ---------------------------------------------------------------------------------------------------
// Device-resident, batched chain behind Frame::ComputeBoW -> ORBmatcher::SearchByBoW: no host round trip between the
// extractor's output block and the matches.
//
// K9b  k_search_by_bow_rows: SIXTEEN LANES (one DPP row) per KeyFrame vocabulary node, four nodes per wave.  The F
// features of the matching node sit on the lanes; for every KF feature of the node (serial: the greedy "F feature already
// claimed" rule, :273-274 / :725, couples them) all lanes evaluate their xor / popcount distance at once and two row
// reductions (v_min over row_ror DPP moves) give best / first position / second.  Lists longer than a row are walked in
// chunks of 16 in list order, merged with the reference's "earlier position wins" rule.  Claim flags of the first chunk
// live in a register, later chunks re-read the match row (written by this very row only: nodes own disjoint features).
// ---------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t row_ror_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v)  // minimum over the 16 lanes of the DPP row, in every lane
{
    v = min(v, row_ror_u32<0x128>(v));  // row_ror:8
    v = min(v, row_ror_u32<0x124>(v));  // row_ror:4
    v = min(v, row_ror_u32<0x122>(v));  // row_ror:2
    v = min(v, row_ror_u32<0x121>(v));  // row_ror:1
    return v;
}

struct BowBatch {
    const uint8_t *desc;        // [B][cap][32]
    const orbfe_keypoint *kps;  // [B][cap]   (angles)
    const uint8_t *valid;       // [B][cap] or null: 1 = the feature has a good MapPoint
    const uint32_t *fv_node, *fv_off, *fv_idx;  // [B][cap], [B][cap+1], [B][cap]
    const int32_t *counts;      // [B][4] {nbow, nfv, nidx, -}
    const int32_t *kf, *f;      // [P] frame indices of the pairs
    int32_t cap, npairs, th_low, strict_lt, use_valid_f, check_ori;
    float nnratio;
    int32_t *match;             // [P][cap] F feature -> KF feature, -1 none
    int32_t *nmatches;          // [P]
};

__global__ __launch_bounds__(256) void k_search_by_bow_rows(BowBatch a)
{
    __shared__ uint4 s_dk[16][16][2];
    __shared__ uint32_t s_rk[16][16];
    const int p = blockIdx.y;
    const int kf = a.kf[p], f = a.f[p];
    const int lane16 = threadIdx.x & 15, rowb = threadIdx.x >> 4;
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 4, nrows = (gridDim.x * 256) >> 4;
    const int nnK = a.counts[kf * 4 + 1], nnF = a.counts[f * 4 + 1];
    const uint32_t *nodeK = a.fv_node + (int64_t)kf * a.cap, *offK = a.fv_off + (int64_t)kf * (a.cap + 1),
                   *idxK = a.fv_idx + (int64_t)kf * a.cap;
    const uint32_t *nodeF = a.fv_node + (int64_t)f * a.cap, *offF = a.fv_off + (int64_t)f * (a.cap + 1),
                   *idxF = a.fv_idx + (int64_t)f * a.cap;
    const uint8_t *descK = a.desc + (int64_t)kf * a.cap * 32, *descF = a.desc + (int64_t)f * a.cap * 32;
    const uint8_t *validK = a.valid ? a.valid + (int64_t)kf * a.cap : nullptr;
    const uint8_t *validF = (a.valid && a.use_valid_f) ? a.valid + (int64_t)f * a.cap : nullptr;
    int32_t *match = a.match + (int64_t)p * a.cap;
    for (int an = row; an < nnK; an += nrows) {  // row-uniform
        const uint32_t node = nodeK[an];
        int lo = 0, hi = nnF - 1, b = -1;  // lower_bound walk of :329-333 == binary search on sorted ids
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const uint32_t v = nodeF[mid];
            if (v == node) { b = mid; break; }
            if (v < node) lo = mid + 1; else hi = mid - 1;
        }
        if (b < 0) continue;
        const uint32_t f0 = offF[b], nFb = offF[b + 1] - f0;
        const uint32_t k0 = offK[an], nKa = offK[an + 1] - k0;
        // chunk 0 of the F list stays in registers
        Desc8 d0;
        uint32_t rf0 = 0;
        bool ok0 = lane16 < (int)nFb;
        if (ok0) {
            rf0 = idxF[f0 + lane16];
            if (validF && !validF[rf0]) ok0 = false;
        }
        {
            const uint32_t *pf = (const uint32_t *)(descF + (int64_t)(ok0 ? rf0 : 0) * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) d0.w[i] = pf[i];
        }
        for (uint32_t t = 0; t < nKa; ++t) {
            // the KF features of the node are staged 16 at a time in LDS (index, MapPoint flag, descriptor): the serial
            // loop below then depends on LDS latency only, not on two dependent global loads per feature
            if ((t & 15u) == 0u) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // earlier reads of the staging area are done
                uint32_t rk_l = 0xFFFFFFFFu;
                if (t + lane16 < nKa) {
                    rk_l = idxK[k0 + t + lane16];
                    if (validK && !validK[rk_l]) rk_l = 0xFFFFFFFFu;  // !pMP || pMP->isBad() (:256-259)
                }
                s_rk[rowb][lane16] = rk_l;
                const uint4 *pk = (const uint4 *)(descK + (int64_t)(rk_l == 0xFFFFFFFFu ? 0u : rk_l) * 32);
                s_dk[rowb][lane16][0] = pk[0];
                s_dk[rowb][lane16][1] = pk[1];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            const uint32_t rk = s_rk[rowb][t & 15u];
            if (rk == 0xFFFFFFFFu) continue;
            Desc8 dk;
            {
                const uint4 q0 = s_dk[rowb][t & 15u][0], q1 = s_dk[rowb][t & 15u][1];
                dk.w[0] = q0.x; dk.w[1] = q0.y; dk.w[2] = q0.z; dk.w[3] = q0.w;
                dk.w[4] = q1.x; dk.w[5] = q1.y; dk.w[6] = q1.z; dk.w[7] = q1.w;
            }
            uint32_t b1 = 256, b2 = 256, bpos = 0xFFFFFu;  // running result over the chunks seen so far
            for (uint32_t c0 = 0; c0 < nFb; c0 += 16) {
                uint32_t dist = 0x3FFu;  // "no candidate"
                if (c0 == 0) {
                    if (ok0) {
                        int d = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) d += __popc(dk.w[i] ^ d0.w[i]);
                        dist = (uint32_t)d;
                    }
                } else if (c0 + lane16 < nFb) {
                    const uint32_t rf = idxF[f0 + c0 + lane16];
                    const bool free = __hip_atomic_load(&match[rf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0;
                    if (free && !(validF && !validF[rf])) dist = (uint32_t)hamming8(dk, (const uint32_t *)(descF + (int64_t)rf * 32));
                }
                const uint32_t key = (dist << 20) | (c0 + lane16);        // smaller = closer, earlier position wins ties
                const uint32_t k1 = row_min_u32(key);
                const uint32_t k2 = row_min_u32(key == k1 ? 0xFFFFFFFFu : key);  // best of the OTHER candidates of the chunk
                const uint32_t c1 = k1 >> 20, cs = min(k2 >> 20, 256u), cpos = k1 & 0xFFFFFu;
                if (c1 < 0x3FFu) {   // merge: the running result covers earlier positions (first minimum wins)
                    if (c1 < b1) { b2 = min(b1, cs); b1 = c1; bpos = cpos; }
                    else { b2 = min(b2, c1); }
                }
            }
            const bool pass = a.strict_lt ? ((int)b1 < a.th_low) : ((int)b1 <= a.th_low);
            if (pass && bpos != 0xFFFFFu && (float)b1 < __fmul_rn(a.nnratio, (float)b2)) {
                const uint32_t rf = bpos < 16 ? __shfl(rf0, (threadIdx.x & 48) + (int)bpos, 64) : idxF[f0 + bpos];
                if (lane16 == 0) __hip_atomic_store(&match[rf], (int32_t)rk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // later chunk re-reads of this row see the claim
                if (bpos == (uint32_t)lane16) ok0 = false;  // claimed (:273 vpMapPointMatches / :725 vbMatched2)
            }
        }
    }
}

// rotation prune for the batched form: angles come from the keypoint records
__global__ __launch_bounds__(256) void k_rot_prune_bow_batch(BowBatch a)
{
    __shared__ int s_hist[ORBFE_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count;
    const int tid = threadIdx.x, p = blockIdx.x;
    const int kf = a.kf[p], f = a.f[p];
    const orbfe_keypoint *kK = a.kps + (int64_t)kf * a.cap, *kF = a.kps + (int64_t)f * a.cap;
    int32_t *match = a.match + (int64_t)p * a.cap;
    if (tid < ORBFE_HISTO_LENGTH) s_hist[tid] = 0;
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (a.check_ori) {
        for (int i = tid; i < a.cap; i += 256) {
            const int j = match[i];
            if (j >= 0) atomicAdd(&s_hist[rot_bin(kK[j].angle, kF[i].angle)], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < ORBFE_HISTO_LENGTH; ++i) {
                const int s = s_hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
            s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
        }
        __syncthreads();
    }
    int local = 0;
    for (int i = tid; i < a.cap; i += 256) {
        const int j = match[i];
        if (j < 0) continue;
        if (a.check_ori) {
            const int bin = rot_bin(kK[j].angle, kF[i].angle);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) { match[i] = -1; continue; }
        }
        ++local;
    }
    atomicAdd(&s_count, local);
    __syncthreads();
    if (tid == 0) a.nmatches[p] = s_count;
}

extern "C" orbfe_status orbfe_bow_transform_batch_device(orbfe_matcher *m, const orbfe_vocabulary *v, const uint8_t *d_desc,
                                                         const int32_t *d_n, int32_t nframes, int32_t cap, int32_t levelsup,
                                                         int32_t *d_f_word, int32_t *d_f_node, double *d_f_weight,
                                                         uint32_t *d_bow_id, double *d_bow_val, uint32_t *d_fv_node,
                                                         uint32_t *d_fv_off, uint32_t *d_fv_idx, int32_t *d_counts,
                                                         void *stream)
{
    if (!m || !v || !d_desc || !d_n || nframes < 1 || cap < 1 || cap > BOW_MAX_FEATURES || !d_f_word || !d_f_node ||
        !d_f_weight || !d_bow_id || !d_bow_val || !d_fv_node || !d_fv_off || !d_fv_idx || !d_counts) {
        orbfe_set_error("bad argument to orbfe_bow_transform_batch_device (cap <= %d)", BOW_MAX_FEATURES);
        return ORBFE_ERR_ARG;
    }
    if (v->device != m->device) { orbfe_set_error("vocabulary and matcher are on different devices"); return ORBFE_ERR_ARG; }
    MDeviceGuard g(m->device);
    hipStream_t st = (hipStream_t)stream;
    int P = 2;
    while (P < cap) P <<= 1;
    const size_t lds = (size_t)P * 16;
    hipFuncAttributes fa;
    ORBFE_HIP(hipFuncGetAttributes(&fa, (const void *)k_bow_aggregate));
    const size_t lds_max = (size_t)ORBFE_LDS_MAX - fa.sharedSizeBytes;
    if (lds > lds_max) { orbfe_set_error("orbfe_bow_transform_batch_device: cap %d needs more than the CU's LDS", cap); return ORBFE_ERR_SIZE; }
    if (lds > 64 * 1024)
        ORBFE_HIP(hipFuncSetAttribute((const void *)k_bow_aggregate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    hipLaunchKernelGGL(k_bow_descend, dim3((cap + 255) / 256, nframes), dim3(256), 0, st, (const uint32_t *)v->child_off.p,
                       (const uint32_t *)v->child_idx.p, (const uint8_t *)v->node_desc.p, (const uint32_t *)v->word_id.p,
                       (const double *)v->weight.p, v->L - levelsup, d_desc, 0, d_f_word, d_f_node, d_f_weight, d_n, cap);
    hipLaunchKernelGGL(k_bow_aggregate, dim3(nframes), dim3(1024), lds, st, 0, P, (const int32_t *)d_f_word,
                       (const int32_t *)d_f_node, (const double *)d_f_weight, d_bow_id, d_bow_val, d_fv_node, d_fv_off,
                       d_fv_idx, d_counts, d_n, cap);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_search_by_bow_batch_device(orbfe_matcher *m, const orbfe_keypoint *d_kps, const uint8_t *d_desc,
                                                         int32_t cap, const uint8_t *d_valid, const uint32_t *d_fv_node,
                                                         const uint32_t *d_fv_off, const uint32_t *d_fv_idx,
                                                         const int32_t *d_counts, const int32_t *d_kf, const int32_t *d_f,
                                                         int32_t npairs, float nnratio, int32_t th_low, int32_t kf_kf,
                                                         int32_t check_ori, int32_t *d_match, int32_t *d_nmatches,
                                                         void *stream)
{
    if (!m || !d_kps || !d_desc || cap < 1 || !d_fv_node || !d_fv_off || !d_fv_idx || !d_counts || !d_kf || !d_f ||
        npairs < 0 || !d_match || !d_nmatches) {
        orbfe_set_error("bad argument to orbfe_search_by_bow_batch_device");
        return ORBFE_ERR_ARG;
    }
    if (npairs == 0) return ORBFE_OK;
    MDeviceGuard g(m->device);
    hipStream_t st = (hipStream_t)stream;
    BowBatch a;
    a.desc = d_desc; a.kps = d_kps; a.valid = d_valid;
    a.fv_node = d_fv_node; a.fv_off = d_fv_off; a.fv_idx = d_fv_idx; a.counts = d_counts;
    a.kf = d_kf; a.f = d_f;
    a.cap = cap; a.npairs = npairs; a.th_low = th_low; a.strict_lt = kf_kf ? 1 : 0; a.use_valid_f = kf_kf ? 1 : 0;
    a.check_ori = check_ori; a.nnratio = nnratio;
    a.match = d_match; a.nmatches = d_nmatches;
    ORBFE_HIP(hipMemsetAsync(d_match, 0xFF, sizeof(int32_t) * (size_t)npairs * cap, st));
    hipLaunchKernelGGL(k_search_by_bow_rows, dim3(8, npairs), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_rot_prune_bow_batch, dim3(npairs), dim3(256), 0, st, a);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}
