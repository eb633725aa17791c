// orbfe_io.hip -- the data formats either side of the hot path (include/orbfe.h "Formats"): host code + one packing kernel.
//
//   Map::Save / Map::Load keyframe records      /root/reference/perfect/src/Map.cc:330-381 (_WriteKeyFrame),
//                                               :143-187 (_ReadKeyFrame), container :385-430 / :228-300
//   ORB vocabulary files (DBoW2, not vendored)  call sites src/System.cc:123-129, tool/text2binary.cc:24-37; the text and
//                                               binary layouts are those of ORB-SLAM2's TemplatedVocabulary
//                                               loadFromTextFile / saveToBinaryFile / loadFromBinaryFile, restated
// Product code: never includes anything from oracle/.
#include <errno.h>

#include <algorithm>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "orbfe_common.h"

// ---------------------------------------------------------------------------------------------------
// Map::Save keyframe block.  Byte layout written by _WriteKeyFrame (x86-64, little endian, no padding):
//   unsigned long mnId (8) | double mTimeStamp (8) | float px, py, pz (12) | float qx, qy, qz, qw (16) | int N (4)
//   N x { float pt.x, pt.y, size, angle, response (20) | int octave (4) | uchar descriptor[32] | unsigned long mpidx (8) }
// class_id is not stored (the reader leaves cv::KeyPoint's default -1); mpidx = ULONG_MAX for "no map point".
// ---------------------------------------------------------------------------------------------------
#define MAPIO_HEADER 48
#define MAPIO_RECORD 64

extern "C" size_t orbfe_mapio_keyframe_bytes(int32_t n) { return n < 0 ? 0 : (size_t)MAPIO_HEADER + (size_t)n * MAPIO_RECORD; }

static inline void put(uint8_t *&p, const void *src, size_t n)
{
    memcpy(p, src, n);
    p += n;
}
static inline void get(const uint8_t *&p, void *dst, size_t n)
{
    memcpy(dst, p, n);
    p += n;
}

// one feature -> its 64-byte record (shared by the host writer and the device packer's reference in the tests)
static inline void pack_record(uint8_t *dst, const orbfe_keypoint &k, const uint8_t *desc, uint64_t mpidx)
{
    uint8_t *p = dst;
    put(p, &k.x, 4);
    put(p, &k.y, 4);
    put(p, &k.size, 4);
    put(p, &k.angle, 4);
    put(p, &k.response, 4);
    put(p, &k.octave, 4);
    put(p, desc, 32);
    put(p, &mpidx, 8);
}

extern "C" orbfe_status orbfe_mapio_write_keyframe(uint8_t *dst, size_t cap, uint64_t id, double timestamp,
                                                   const float t_cw[3], const float q_cw[4], const orbfe_keypoint *kps,
                                                   const uint8_t *desc, const uint64_t *mp_index, int32_t n,
                                                   size_t *written)
{
    if (!dst || !t_cw || !q_cw || n < 0 || (n > 0 && (!kps || !desc))) { orbfe_set_error("bad argument to orbfe_mapio_write_keyframe"); return ORBFE_ERR_ARG; }
    const size_t need = orbfe_mapio_keyframe_bytes(n);
    if (written) *written = need;
    if (cap < need) { orbfe_set_error("orbfe_mapio_write_keyframe: %zu bytes needed, %zu given", need, cap); return ORBFE_ERR_CAP; }
    uint8_t *p = dst;
    put(p, &id, 8);
    put(p, &timestamp, 8);
    put(p, t_cw, 12);
    put(p, q_cw, 16);
    put(p, &n, 4);
    for (int i = 0; i < n; ++i, p += MAPIO_RECORD)
        pack_record(p, kps[i], desc + (size_t)i * 32, mp_index ? mp_index[i] : ~(uint64_t)0);
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_mapio_read_keyframe(const uint8_t *src, size_t len, uint64_t *id, double *timestamp,
                                                  float t_cw[3], float q_cw[4], orbfe_keypoint *kps, uint8_t *desc,
                                                  uint64_t *mp_index, int32_t cap, int32_t *n, size_t *consumed)
{
    if (!src || !n) { orbfe_set_error("bad argument to orbfe_mapio_read_keyframe"); return ORBFE_ERR_ARG; }
    if (len < MAPIO_HEADER) { orbfe_set_error("keyframe block truncated (header)"); return ORBFE_ERR_SIZE; }
    const uint8_t *p = src;
    uint64_t kid;
    double ts;
    float t[3], q[4];
    int32_t cnt;
    get(p, &kid, 8);
    get(p, &ts, 8);
    get(p, t, 12);
    get(p, q, 16);
    get(p, &cnt, 4);
    if (cnt < 0 || len < orbfe_mapio_keyframe_bytes(cnt)) { orbfe_set_error("keyframe block truncated (%d features)", cnt); return ORBFE_ERR_SIZE; }
    *n = cnt;
    if (consumed) *consumed = orbfe_mapio_keyframe_bytes(cnt);
    if (id) *id = kid;
    if (timestamp) *timestamp = ts;
    if (t_cw) memcpy(t_cw, t, 12);
    if (q_cw) memcpy(q_cw, q, 16);
    if (cnt > cap) { orbfe_set_error("orbfe_mapio_read_keyframe: %d features, capacity %d", cnt, cap); return ORBFE_ERR_CAP; }
    for (int i = 0; i < cnt; ++i) {
        orbfe_keypoint k;
        get(p, &k.x, 4);
        get(p, &k.y, 4);
        get(p, &k.size, 4);
        get(p, &k.angle, 4);
        get(p, &k.response, 4);
        get(p, &k.octave, 4);
        k.class_id = -1;  // cv::KeyPoint() default: the reader never sets it (:170-178)
        if (kps) kps[i] = k;
        if (desc) memcpy(desc + (size_t)i * 32, p, 32);
        p += 32;
        uint64_t mp;
        get(p, &mp, 8);
        if (mp_index) mp_index[i] = mp;
    }
    return ORBFE_OK;
}

// The same records straight from an extractor output block in HBM: frame b -> d_out + b * cap * 64, slots >= d_n[b] zeroed.
// One thread per 16 bytes of output (coalesced 16-byte stores), so what the D2H copy (or the all-gather) moves is already
// the byte stream Map::Save writes between the keyframe headers.
__global__ __launch_bounds__(256) void k_pack_map_records(const orbfe_keypoint *__restrict__ kps,
                                                          const uint8_t *__restrict__ desc, const int32_t *__restrict__ n,
                                                          const uint64_t *__restrict__ mp_index, int cap,
                                                          uint4 *__restrict__ out)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;  // 16-byte chunk inside the frame: 4 per record
    if (t >= cap * 4) return;
    const int i = t >> 2, part = t & 3;
    const int64_t slot = (int64_t)b * cap + i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i < min(n[b], cap)) {
        const uint32_t *k = (const uint32_t *)(kps + slot);           // x y size angle response octave class_id
        const uint32_t *d = (const uint32_t *)(desc + slot * 32);
        if (part == 0) v = make_uint4(k[0], k[1], k[2], k[3]);
        else if (part == 1) v = make_uint4(k[4], k[5], d[0], d[1]);
        else if (part == 2) v = make_uint4(d[2], d[3], d[4], d[5]);
        else {
            const uint64_t mp = mp_index ? mp_index[slot] : ~0ull;
            v = make_uint4(d[6], d[7], (uint32_t)mp, (uint32_t)(mp >> 32));
        }
    }
    out[slot * 4 + part] = v;
}

extern "C" orbfe_status orbfe_mapio_pack_records_device(const orbfe_keypoint *d_kps, const uint8_t *d_desc,
                                                        const int32_t *d_n, const uint64_t *d_mp_index, int32_t nframes,
                                                        int32_t cap, uint8_t *d_out, void *stream)
{
    if (!d_kps || !d_desc || !d_n || !d_out || nframes < 1 || cap < 1) { orbfe_set_error("bad argument to orbfe_mapio_pack_records_device"); return ORBFE_ERR_ARG; }
    hipLaunchKernelGGL(k_pack_map_records, dim3((cap * 4 + 255) / 256, nframes), dim3(256), 0, (hipStream_t)stream, d_kps,
                       d_desc, d_n, d_mp_index, cap, (uint4 *)d_out);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// Tracking::GrabImageRGBD's colour -> gray step (src/Tracking.cc:339-353) on device-resident frames: cvtColor's 14-bit
// fixed point luma, gray = (w0*c0 + 9617*c1 + w2*c2 + 8192) >> 14 with (w0, w2) = (4899, 1868) for CV_RGB2GRAY applied to
// the memory order it is given (what the reference does to cv::imread's BGR with Camera.RGB = 1) and (1868, 4899) for
// CV_BGR2GRAY.  Four pixels per thread: three dword loads, one dword store.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_interleaved_to_gray(const uint8_t *__restrict__ src, int64_t src_fstride, int spitch,
                                                             uint8_t *__restrict__ dst, int64_t dst_fstride, int dpitch, int w,
                                                             int h, uint32_t w0, uint32_t w2)
{
    const int b = blockIdx.z, y = blockIdx.y;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= w) return;
    const uint8_t *row = src + (int64_t)b * src_fstride + (int64_t)y * spitch + (int64_t)x4 * 3;
    uint8_t *o = dst + (int64_t)b * dst_fstride + (int64_t)y * dpitch + x4;
    const int n = min(4, w - x4);
    uint32_t g[4] = {0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
        const uint32_t c0 = row[3 * j], c1 = row[3 * j + 1], c2 = row[3 * j + 2];
        g[j] = (c0 * w0 + c1 * 9617u + c2 * w2 + 8192u) >> 14;
    }
    if (n == 4 && (((uintptr_t)o) & 3) == 0) *(uint32_t *)o = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
    else
        for (int j = 0; j < n; ++j) o[j] = (uint8_t)g[j];
}

extern "C" orbfe_status orbfe_interleaved_to_gray_device(const uint8_t *d_src, int32_t nframes, int32_t w, int32_t h,
                                                         int32_t src_stride, size_t src_frame_stride, int32_t rgb_flag,
                                                         uint8_t *d_gray, int32_t gray_stride, size_t gray_frame_stride,
                                                         void *stream)
{
    if (!d_src || !d_gray || nframes < 1 || w < 1 || h < 1 || src_stride < 3 * w || gray_stride < w) {
        orbfe_set_error("bad argument to orbfe_interleaved_to_gray_device");
        return ORBFE_ERR_ARG;
    }
    const uint32_t w0 = rgb_flag ? 4899u : 1868u, w2 = rgb_flag ? 1868u : 4899u;
    hipLaunchKernelGGL(k_interleaved_to_gray, dim3((w + 1023) / 1024, h, nframes), dim3(256), 0, (hipStream_t)stream, d_src,
                       (int64_t)src_frame_stride, src_stride, d_gray, (int64_t)gray_frame_stride, gray_stride, w, h, w0, w2);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

// ---------------------------------------------------------------------------------------------------
// ORB vocabulary files
//   text   (ORBvoc.txt):  first line "k L scoring weighting"; then one line per node except the root, in node-id order:
//                         "parent_id is_leaf d0 ... d31 weight" (descriptor bytes as decimal numbers, weight as double)
//   binary (ORBvoc.bin):  u32 nb_nodes | u32 size_node (= 41) | i32 k | i32 L | i32 scoring | i32 weighting, then for node
//                         1 .. nb_nodes-1:  u32 parent | u8 descriptor[32] | f32 weight | u8 is_leaf
// Word ids are assigned in node order to the leaves, as both DBoW2 loaders do.
// ---------------------------------------------------------------------------------------------------
struct orbfe_vocfile {
    int32_t k = 0, L = 0, scoring = 0, weighting = 0, nwords = 0;
    std::vector<uint32_t> parent;      // [nnodes], parent[0] = 0
    std::vector<uint8_t> leaf;         // [nnodes]
    std::vector<uint8_t> desc;         // [nnodes][32]
    std::vector<double> weight;        // [nnodes]
    std::vector<uint32_t> word_id;     // [nnodes]
    std::vector<uint32_t> child_off, child_idx;
};

static orbfe_status finish_vocfile(orbfe_vocfile *v)
{
    const size_t nn = v->parent.size();
    if (nn < 1 || nn >= (1u << 31)) { orbfe_set_error("vocabulary: %zu nodes", nn); return ORBFE_ERR_SIZE; }
    std::vector<uint32_t> cnt(nn + 1, 0);
    for (size_t i = 1; i < nn; ++i) {
        if (v->parent[i] >= i) { orbfe_set_error("vocabulary: node %zu has parent %u (parents must precede children)", i, v->parent[i]); return ORBFE_ERR_ARG; }
        cnt[v->parent[i] + 1]++;
    }
    for (size_t i = 0; i < nn; ++i) cnt[i + 1] += cnt[i];
    v->child_off = cnt;
    v->child_idx.assign(nn > 1 ? nn - 1 : 0, 0);
    std::vector<uint32_t> fill(v->child_off.begin(), v->child_off.end() - 1);
    for (size_t i = 1; i < nn; ++i) v->child_idx[fill[v->parent[i]]++] = (uint32_t)i;  // ascending ids = push_back order
    v->word_id.assign(nn, 0);
    v->nwords = 0;
    for (size_t i = 1; i < nn; ++i)
        if (v->leaf[i]) v->word_id[i] = (uint32_t)v->nwords++;
    return ORBFE_OK;
}

static orbfe_status load_text(std::ifstream &f, orbfe_vocfile *v)
{
    std::string line;
    if (!std::getline(f, line)) { orbfe_set_error("vocabulary text file is empty"); return ORBFE_ERR_SIZE; }
    {
        std::stringstream ss(line);
        ss >> v->k >> v->L >> v->scoring >> v->weighting;
        // the range check of TemplatedVocabulary::loadFromTextFile
        if (ss.fail() || v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || v->scoring < 0 || v->scoring > 5 || v->weighting < 0 ||
            v->weighting > 3) {
            orbfe_set_error("vocabulary text header out of range: '%s'", line.c_str());
            return ORBFE_ERR_ARG;
        }
    }
    v->parent.assign(1, 0);
    v->leaf.assign(1, 0);
    v->desc.assign(32, 0);
    v->weight.assign(1, 0.0);
    while (std::getline(f, line)) {
        if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;  // trailing blank line
        const char *p = line.c_str();
        char *e = nullptr;
        errno = 0;
        const long pid = strtol(p, &e, 10);
        if (e == p) { orbfe_set_error("vocabulary: unreadable node line %zu", v->parent.size()); return ORBFE_ERR_ARG; }
        p = e;
        const long is_leaf = strtol(p, &e, 10);
        p = e;
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) {
            const long b = strtol(p, &e, 10);
            if (e == p || b < 0 || b > 255) { orbfe_set_error("vocabulary: bad descriptor byte on node line %zu", v->parent.size()); return ORBFE_ERR_ARG; }
            d[i] = (uint8_t)b;
            p = e;
        }
        const double w = strtod(p, &e);
        if (e == p || pid < 0) { orbfe_set_error("vocabulary: bad weight / parent on node line %zu", v->parent.size()); return ORBFE_ERR_ARG; }
        v->parent.push_back((uint32_t)pid);
        v->leaf.push_back(is_leaf > 0);
        v->desc.insert(v->desc.end(), d, d + 32);
        v->weight.push_back(w);
    }
    return ORBFE_OK;
}

static orbfe_status load_binary(std::ifstream &f, orbfe_vocfile *v)
{
    uint32_t nb = 0, sz = 0;
    f.read((char *)&nb, 4);
    f.read((char *)&sz, 4);
    f.read((char *)&v->k, 4);
    f.read((char *)&v->L, 4);
    f.read((char *)&v->scoring, 4);
    f.read((char *)&v->weighting, 4);
    if (!f || sz != 41 || nb < 1) { orbfe_set_error("vocabulary binary header: nb_nodes %u, size_node %u (expected 41)", nb, sz); return ORBFE_ERR_ARG; }
    {
        // a corrupt node count must not drive the allocations below: the file has to hold (nb - 1) node records
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streampos end = f.tellg();
        f.seekg(here);
        if (!f || end < here || (uint64_t)(end - here) < (uint64_t)(nb - 1) * 41) {
            orbfe_set_error("vocabulary binary file truncated: header says %u nodes, %lld bytes of records present", nb,
                            (long long)(end - here));
            return ORBFE_ERR_SIZE;
        }
    }
    v->parent.assign(nb, 0);
    v->leaf.assign(nb, 0);
    v->desc.assign((size_t)nb * 32, 0);
    v->weight.assign(nb, 0.0);
    std::vector<char> buf((size_t)(nb - 1) * 41);
    f.read(buf.data(), (std::streamsize)buf.size());
    if ((size_t)f.gcount() != buf.size()) { orbfe_set_error("vocabulary binary file truncated"); return ORBFE_ERR_SIZE; }
    for (uint32_t i = 1; i < nb; ++i) {
        const char *r = buf.data() + (size_t)(i - 1) * 41;
        uint32_t pid;
        float w;
        memcpy(&pid, r, 4);
        memcpy(&v->desc[(size_t)i * 32], r + 4, 32);
        memcpy(&w, r + 36, 4);
        v->parent[i] = pid;
        v->weight[i] = (double)w;
        v->leaf[i] = r[40] != 0;
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_vocfile_load(const char *path, orbfe_vocfile **out)
{
    if (!path || !out) { orbfe_set_error("bad argument to orbfe_vocfile_load"); return ORBFE_ERR_ARG; }
    *out = nullptr;
    std::ifstream f(path, std::ios::in | std::ios::binary);
    if (!f) { orbfe_set_error("cannot open vocabulary file %s", path); return ORBFE_ERR_ARG; }
    // the reference picks the loader by file name (src/System.cc:123-129: ".txt" -> text, else binary); sniffing the first
    // bytes gives the same answer: a text header starts with a decimal digit, nb_nodes' low byte as ASCII digit + space
    // would need nb_nodes = 0x20xx.. with size_node 41 impossible at the same time
    char head[8] = {0};
    f.read(head, 8);
    f.clear();
    f.seekg(0);
    uint32_t sz;
    memcpy(&sz, head + 4, 4);
    const bool binary = sz == 41;
    orbfe_vocfile *v = new (std::nothrow) orbfe_vocfile();
    if (!v) return ORBFE_ERR_NOMEM;
    orbfe_status s;
    try {  // nothing may throw through the C boundary
        s = binary ? load_binary(f, v) : load_text(f, v);
        if (s == ORBFE_OK) s = finish_vocfile(v);
    } catch (const std::bad_alloc &) {
        orbfe_set_error("out of memory while loading vocabulary file %s", path);
        s = ORBFE_ERR_NOMEM;
    } catch (const std::exception &e) {
        orbfe_set_error("vocabulary file %s: %s", path, e.what());
        s = ORBFE_ERR_SIZE;
    }
    if (s != ORBFE_OK) {
        delete v;
        return s;
    }
    *out = v;
    return ORBFE_OK;
}

extern "C" void orbfe_vocfile_free(orbfe_vocfile *v) { delete v; }

extern "C" orbfe_status orbfe_vocfile_info(const orbfe_vocfile *v, int32_t *k, int32_t *L, int32_t *nnodes, int32_t *nwords,
                                           int32_t *scoring, int32_t *weighting)
{
    if (!v) return ORBFE_ERR_ARG;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (nnodes) *nnodes = (int32_t)v->parent.size();
    if (nwords) *nwords = v->nwords;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_vocfile_arrays(const orbfe_vocfile *v, const uint32_t **child_off, const uint32_t **child_idx,
                                             const uint8_t **node_desc, const uint32_t **word_id, const double **weight,
                                             const uint32_t **parent, const uint8_t **is_leaf)
{
    if (!v) return ORBFE_ERR_ARG;
    if (child_off) *child_off = v->child_off.data();
    if (child_idx) *child_idx = v->child_idx.data();
    if (node_desc) *node_desc = v->desc.data();
    if (word_id) *word_id = v->word_id.data();
    if (weight) *weight = v->weight.data();
    if (parent) *parent = v->parent.data();
    if (is_leaf) *is_leaf = v->leaf.data();
    return ORBFE_OK;
}

// tool/text2binary.cc:30-32 (saveToBinaryFile)
extern "C" orbfe_status orbfe_vocfile_save_binary(const orbfe_vocfile *v, const char *path)
{
    if (!v || !path) return ORBFE_ERR_ARG;
    std::ofstream f(path, std::ios::out | std::ios::binary);
    if (!f) { orbfe_set_error("cannot write %s", path); return ORBFE_ERR_ARG; }
    const uint32_t nb = (uint32_t)v->parent.size(), sz = 41;
    if (nb < 1) { orbfe_set_error("vocabulary has no root node"); return ORBFE_ERR_ARG; }
    f.write((const char *)&nb, 4);
    f.write((const char *)&sz, 4);
    f.write((const char *)&v->k, 4);
    f.write((const char *)&v->L, 4);
    f.write((const char *)&v->scoring, 4);
    f.write((const char *)&v->weighting, 4);
    std::vector<char> buf;
    try {
        buf.resize((size_t)(nb - 1) * 41);
    } catch (const std::bad_alloc &) {
        return ORBFE_ERR_NOMEM;
    }
    for (uint32_t i = 1; i < nb; ++i) {
        char *r = buf.data() + (size_t)(i - 1) * 41;
        const float w = (float)v->weight[i];
        memcpy(r, &v->parent[i], 4);
        memcpy(r + 4, &v->desc[(size_t)i * 32], 32);
        memcpy(r + 36, &w, 4);
        r[40] = v->leaf[i] ? 1 : 0;
    }
    f.write(buf.data(), (std::streamsize)buf.size());
    return f ? ORBFE_OK : ORBFE_ERR_ARG;
}

extern "C" orbfe_status orbfe_vocabulary_create_from_file(int32_t device, const orbfe_vocfile *v, orbfe_vocabulary **out)
{
    if (!v || !out) return ORBFE_ERR_ARG;
    return orbfe_vocabulary_create(device, (int32_t)v->parent.size(), v->child_off.data(), v->child_idx.data(), v->desc.data(),
                                   v->word_id.data(), v->weight.data(), v->L, out);
}
