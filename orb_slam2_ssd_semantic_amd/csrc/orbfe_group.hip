// orbfe_group.hip -- the batched keyframe mode for C / C++ hosts: a batch of independent frames is cut
// into contiguous shards, one per device; every device extracts its shard into its slice of three padded blocks (counts,
// keypoints, descriptors); ONE all-gather of those blocks (RCCL over xGMI) leaves the whole batch on every device; the
// consumer -- what KeyFrameDatabase / LoopClosing do serially on the CPU (src/LoopClosing.cc:312-342) -- then matches the
// device's own frames against candidate frames anywhere in the gathered set (orbfe_match_bf_frames_device on the block).
//
// Two ways to form a group, same calls afterwards:
//   orbfe_group_create_local  one process drives several devices (ncclCommInitAll) -- the shape of a C++ SLAM process
//   orbfe_group_create_rank   one process per device (ncclCommInitRank with an id made by orbfe_group_unique_id on rank 0
//                             and handed to the other ranks by whatever means the host has): what bench.py launches
// RCCL is loaded at run time (dlopen: the copy already in the process if there is one -- PyTorch ships its own librccl --
// else librccl.so.1 / $ORBFE_RCCL_LIB); liborbfe.so itself has no link-time dependency on it.
//
// Transport of the exchange step (local groups): ORBFE_GROUP_RCCL = three in-place ncclAllGather calls; ORBFE_GROUP_COPY =
// the same exchange as hipMemcpyAsync device-to-device copies (peer copies over xGMI between devices) from every other
// member's slice into the member's blocks, on the same communication streams behind the same events.  The copy transport
// needs no communicator, so several members may share ONE device: that is how the multi-member code paths (slice offsets,
// uneven shards, zero tails, cross-shard matching) run under `-m gpu` on a one-GPU box.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include <rccl/rccl.h>

#include "orbfe_common.h"

// ---------------------------------------------------------------------------------------------------
// RCCL entry points, resolved once
// ---------------------------------------------------------------------------------------------------
namespace
{
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char *names[] = {getenv("ORBFE_RCCL_LIB"), "librccl.so", "librccl.so.1", nullptr};
    for (int pass = 0; pass < 2 && !r.lib; ++pass)  // pass 0: a copy that is already mapped (RTLD_NOLOAD), pass 1: load one
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (r.lib) break;
        }
    if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) return r;
#define ORBFE_SYM(field, name) r.field = (decltype(r.field))dlsym(r.lib, name)
    ORBFE_SYM(GetUniqueId, "ncclGetUniqueId");
    ORBFE_SYM(CommInitRank, "ncclCommInitRank");
    ORBFE_SYM(CommInitAll, "ncclCommInitAll");
    ORBFE_SYM(CommDestroy, "ncclCommDestroy");
    ORBFE_SYM(AllGather, "ncclAllGather");
    ORBFE_SYM(GroupStart, "ncclGroupStart");
    ORBFE_SYM(GroupEnd, "ncclGroupEnd");
    ORBFE_SYM(GetErrorString, "ncclGetErrorString");
#undef ORBFE_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
    return r;
}

#define ORBFE_NCCL(call)                                                                                    \
    do {                                                                                                    \
        const ncclResult_t nr_ = (call);                                                                    \
        if (nr_ != ncclSuccess) {                                                                           \
            orbfe_set_error("%s failed: %s", #call, rccl().GetErrorString ? rccl().GetErrorString(nr_) : "?"); \
            return ORBFE_ERR_HIP;                                                                           \
        }                                                                                                   \
    } while (0)

// the calling thread's device, restored on every exit path
struct GDeviceGuard {
    int prev = -1;
    GDeviceGuard() { (void)hipGetDevice(&prev); }
    ~GDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Member {  // one device of the group that THIS process drives
    int device = 0, rank = 0;
    orbfe_handle *ext = nullptr;
    orbfe_matcher *mat = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t s_cmp = nullptr, s_comm = nullptr;
    hipEvent_t ev_cmp = nullptr, ev_comm = nullptr;
    // the gathered blocks: [world * shard][...]; this member's extractor writes slice `rank`
    int32_t *d_n = nullptr;
    orbfe_keypoint *d_kps = nullptr;
    uint8_t *d_desc = nullptr;
    uint8_t *d_stage = nullptr;  // host-frame staging of the shard
    size_t stage_bytes = 0;
    int32_t *d_pairs = nullptr, *d_match = nullptr, *d_nm = nullptr;  // consumer scratch
    size_t pairs_cap = 0;
};
}  // namespace

struct orbfe_group {
    orbfe_params prm;
    int transport = ORBFE_GROUP_RCCL;
    int world = 1;           // ranks in the communicator
    int shard = 0;           // frames per rank slice of the blocks (= max_batch of every extractor)
    int cap = 0;
    std::vector<Member> mem; // the ranks of this process (all of them for a local group, one for a rank group)
    int last_nframes = 0;    // global batch size of the last extract call
};

extern "C" void orbfe_group_shard_range(int32_t nframes, int32_t rank, int32_t world, int32_t *lo, int32_t *hi)
{
    // contiguous blocks, the remainder frames go to the lowest ranks (= distributed.shard_range)
    if (world < 1) world = 1;
    const int base = nframes / world, rem = nframes % world;
    const int a = rank * base + std::min(rank, rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (rank < rem ? 1 : 0);
}

// rank that owns global frame f of a batch of nframes cut by orbfe_group_shard_range
static inline int owner_rank_of(int world, int nframes, int f)
{
    const int base = nframes / world, rem = nframes % world;
    return f < rem * (base + 1) ? f / (base + 1) : rem + (f - rem * (base + 1)) / std::max(base, 1);
}

extern "C" int32_t orbfe_group_owner_rank(int32_t nframes, int32_t world, int32_t frame)
{
    if (world < 1 || nframes < 1 || frame < 0 || frame >= nframes) return -1;
    return owner_rank_of(world, nframes, frame);
}

extern "C" int32_t orbfe_group_block_index_of(int32_t nframes, int32_t world, int32_t shard, int32_t frame)
{
    if (world < 1 || nframes < 1 || frame < 0 || frame >= nframes || shard * world < nframes) return -1;
    const int r = owner_rank_of(world, nframes, frame);
    int lo, hi;
    orbfe_group_shard_range(nframes, r, world, &lo, &hi);
    return r * shard + (frame - lo);
}

extern "C" orbfe_status orbfe_group_unique_id(uint8_t id[128])
{
    if (!id) return ORBFE_ERR_ARG;
    if (!rccl().ok) { orbfe_set_error("RCCL could not be loaded (librccl.so / $ORBFE_RCCL_LIB)"); return ORBFE_ERR_STATE; }
    ncclUniqueId u;
    ORBFE_NCCL(rccl().GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, 128);
    return ORBFE_OK;
}

static void drain_members(orbfe_group *g)
{
    for (Member &m : g->mem) {
        if (m.device >= 0) (void)hipSetDevice(m.device);
        if (m.s_cmp) (void)hipStreamSynchronize(m.s_cmp);
        if (m.s_comm) (void)hipStreamSynchronize(m.s_comm);
    }
}

static void destroy_member(Member &m)
{
    if (m.device >= 0) (void)hipSetDevice(m.device);
    if (m.s_cmp) (void)hipStreamSynchronize(m.s_cmp);
    if (m.s_comm) (void)hipStreamSynchronize(m.s_comm);
    if (m.comm) (void)rccl().CommDestroy(m.comm);
    if (m.ext) orbfe_destroy(m.ext);
    if (m.mat) orbfe_matcher_destroy(m.mat);
    void *bufs[] = {m.d_n, m.d_kps, m.d_desc, m.d_stage, m.d_pairs, m.d_match, m.d_nm};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (m.ev_cmp) (void)hipEventDestroy(m.ev_cmp);
    if (m.ev_comm) (void)hipEventDestroy(m.ev_comm);
    if (m.s_cmp) (void)hipStreamDestroy(m.s_cmp);
    if (m.s_comm) (void)hipStreamDestroy(m.s_comm);
    m = Member();
}

extern "C" void orbfe_group_destroy(orbfe_group *g)
{
    if (!g) return;
    GDeviceGuard guard;
    // under the copy transport every member pulls the OTHER members' slices on its own communication stream: all streams of
    // all members are drained before the first block is freed
    drain_members(g);
    for (Member &m : g->mem) destroy_member(m);
    delete g;
}

static orbfe_status init_member(orbfe_group *g, Member &m)
{
    ORBFE_HIP(hipSetDevice(m.device));
    orbfe_params p = g->prm;
    p.device = m.device;
    p.max_batch = g->shard;
    orbfe_status s = orbfe_create(&p, &m.ext);
    if (s != ORBFE_OK) return s;
    s = orbfe_matcher_create(m.device, &m.mat);
    if (s != ORBFE_OK) return s;
    g->cap = orbfe_keypoint_capacity(m.ext);
    ORBFE_HIP(hipStreamCreateWithFlags(&m.s_cmp, hipStreamNonBlocking));
    ORBFE_HIP(hipStreamCreateWithFlags(&m.s_comm, hipStreamNonBlocking));
    ORBFE_HIP(hipEventCreateWithFlags(&m.ev_cmp, hipEventDisableTiming));
    ORBFE_HIP(hipEventCreateWithFlags(&m.ev_comm, hipEventDisableTiming));
    const size_t F = (size_t)g->world * g->shard;
    ORBFE_HIP(hipMalloc((void **)&m.d_n, F * sizeof(int32_t)));
    ORBFE_HIP(hipMalloc((void **)&m.d_kps, F * g->cap * sizeof(orbfe_keypoint)));
    ORBFE_HIP(hipMalloc((void **)&m.d_desc, F * g->cap * 32));
    ORBFE_HIP(hipMemset(m.d_n, 0, F * sizeof(int32_t)));
    ORBFE_HIP(hipMemset(m.d_kps, 0, F * g->cap * sizeof(orbfe_keypoint)));
    ORBFE_HIP(hipMemset(m.d_desc, 0, F * g->cap * 32));
    return ORBFE_OK;
}

static orbfe_status group_alloc(const orbfe_params *p, int world, int transport, orbfe_group **out)
{
    if (!p || !out || world < 1 || p->max_batch < 1) { orbfe_set_error("bad argument to orbfe_group_create"); return ORBFE_ERR_ARG; }
    if (transport == ORBFE_GROUP_RCCL && !rccl().ok) { orbfe_set_error("RCCL could not be loaded (librccl.so / $ORBFE_RCCL_LIB)"); return ORBFE_ERR_STATE; }
    orbfe_group *g = new (std::nothrow) orbfe_group();
    if (!g) return ORBFE_ERR_NOMEM;
    g->prm = *p;
    g->transport = transport;
    g->world = world;
    g->shard = (p->max_batch + world - 1) / world;  // max_batch = the largest GLOBAL batch
    *out = g;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_create_local_ex(const orbfe_params *p, const int32_t *devices, int32_t ndevices, int32_t transport,
                                                    orbfe_group **out)
{
    if (out) *out = nullptr;
    if (!devices || ndevices < 1 || (transport != ORBFE_GROUP_RCCL && transport != ORBFE_GROUP_COPY)) {
        orbfe_set_error("bad argument to orbfe_group_create_local");
        return ORBFE_ERR_ARG;
    }
    orbfe_group *g = nullptr;
    orbfe_status s = group_alloc(p, ndevices, transport, &g);
    if (s != ORBFE_OK) return s;
    GDeviceGuard guard;
    g->mem.resize((size_t)ndevices);
    std::vector<ncclComm_t> comms((size_t)ndevices, nullptr);
    std::vector<int> devs(devices, devices + ndevices);
    if (transport == ORBFE_GROUP_RCCL) {
        const ncclResult_t nr = rccl().CommInitAll(comms.data(), ndevices, devs.data());
        if (nr != ncclSuccess) {
            orbfe_set_error("ncclCommInitAll failed: %s", rccl().GetErrorString ? rccl().GetErrorString(nr) : "?");
            g->mem.clear();
            orbfe_group_destroy(g);
            return ORBFE_ERR_HIP;
        }
    }
    for (int r = 0; r < ndevices; ++r) {
        g->mem[(size_t)r].device = devs[(size_t)r];
        g->mem[(size_t)r].rank = r;
        g->mem[(size_t)r].comm = comms[(size_t)r];
    }
    for (int r = 0; r < ndevices && s == ORBFE_OK; ++r) s = init_member(g, g->mem[(size_t)r]);
    if (s == ORBFE_OK && transport == ORBFE_GROUP_COPY)
        // peer access between distinct devices, so that the exchange copies go over xGMI directly (already-enabled is fine)
        for (int a = 0; a < ndevices; ++a)
            for (int b = 0; b < ndevices; ++b) {
                if (devs[(size_t)a] == devs[(size_t)b]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devs[(size_t)a], devs[(size_t)b]) == hipSuccess && can && hipSetDevice(devs[(size_t)a]) == hipSuccess) {
                    const hipError_t e = hipDeviceEnablePeerAccess(devs[(size_t)b], 0);
                    if (e != hipSuccess) (void)hipGetLastError();  // hipErrorPeerAccessAlreadyEnabled
                }
            }
    if (s != ORBFE_OK) { orbfe_group_destroy(g); return s; }
    *out = g;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_create_local(const orbfe_params *p, const int32_t *devices, int32_t ndevices, orbfe_group **out)
{
    return orbfe_group_create_local_ex(p, devices, ndevices, ORBFE_GROUP_RCCL, out);
}

extern "C" orbfe_status orbfe_group_create_rank(const orbfe_params *p, int32_t device, int32_t rank, int32_t world, const uint8_t id[128],
                                                orbfe_group **out)
{
    if (out) *out = nullptr;
    if (!id || rank < 0 || rank >= world) { orbfe_set_error("bad argument to orbfe_group_create_rank"); return ORBFE_ERR_ARG; }
    orbfe_group *g = nullptr;
    orbfe_status s = group_alloc(p, world, ORBFE_GROUP_RCCL, &g);
    if (s != ORBFE_OK) return s;
    GDeviceGuard guard;
    if (device < 0) device = guard.prev >= 0 ? guard.prev : 0;
    g->mem.resize(1);
    Member &m = g->mem[0];
    m.device = device;
    m.rank = rank;
    if (hipSetDevice(device) != hipSuccess) { orbfe_set_error("hipSetDevice(%d) failed", device); orbfe_group_destroy(g); return ORBFE_ERR_HIP; }
    ncclUniqueId u;
    memcpy(&u, id, 128);
    const ncclResult_t nr = rccl().CommInitRank(&m.comm, world, u, rank);
    if (nr != ncclSuccess) {
        orbfe_set_error("ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(nr) : "?");
        m.comm = nullptr;
        orbfe_group_destroy(g);
        return ORBFE_ERR_HIP;
    }
    s = init_member(g, m);
    if (s != ORBFE_OK) { orbfe_group_destroy(g); return s; }
    *out = g;
    return ORBFE_OK;
}

extern "C" int32_t orbfe_group_transport(const orbfe_group *g) { return g ? g->transport : -1; }
extern "C" int32_t orbfe_group_members(const orbfe_group *g) { return g ? (int32_t)g->mem.size() : 0; }
extern "C" int32_t orbfe_group_world(const orbfe_group *g) { return g ? g->world : 0; }
extern "C" int32_t orbfe_group_capacity(const orbfe_group *g) { return g ? g->cap : 0; }
extern "C" int32_t orbfe_group_frames_padded(const orbfe_group *g) { return g ? g->world * g->shard : 0; }

// rank that owns global frame f of a batch of nframes, and the frame's index in the gathered blocks
static inline int owner_rank(const orbfe_group *g, int nframes, int f) { return owner_rank_of(g->world, nframes, f); }
static inline int block_index(const orbfe_group *g, int nframes, int f)
{
    const int r = owner_rank(g, nframes, f);
    int lo, hi;
    orbfe_group_shard_range(nframes, r, g->world, &lo, &hi);
    return r * g->shard + (f - lo);
}

extern "C" int32_t orbfe_group_block_index(const orbfe_group *g, int32_t nframes, int32_t frame)
{
    if (!g || nframes < 1 || frame < 0 || frame >= nframes) return -1;
    return block_index(g, nframes, frame);
}

static orbfe_status extract_member(orbfe_group *g, Member &m, const uint8_t *d_gray, int nsh, int w, int ht, int stride, size_t fstride)
{
    // the extractor writes its shard into slice `rank` of the member's blocks; the unused tail of the slice is cleared so
    // that a short last shard gathers as empty frames.  First the previous gather has to be done with the slice: it read
    // it on the communication stream (copy transport: on EVERY member's communication stream -- the others pull it).
    const size_t at = (size_t)m.rank * g->shard;
    if (g->transport == ORBFE_GROUP_COPY) {
        for (Member &o : g->mem) ORBFE_HIP(hipStreamWaitEvent(m.s_cmp, o.ev_comm, 0));
    } else {
        ORBFE_HIP(hipStreamWaitEvent(m.s_cmp, m.ev_comm, 0));
    }
    if (nsh < g->shard) {
        ORBFE_HIP(hipMemsetAsync(m.d_n + at + nsh, 0, (size_t)(g->shard - nsh) * sizeof(int32_t), m.s_cmp));
        ORBFE_HIP(hipMemsetAsync(m.d_kps + (at + nsh) * g->cap, 0, (size_t)(g->shard - nsh) * g->cap * sizeof(orbfe_keypoint), m.s_cmp));
        ORBFE_HIP(hipMemsetAsync(m.d_desc + (at + nsh) * g->cap * 32, 0, (size_t)(g->shard - nsh) * g->cap * 32, m.s_cmp));
    }
    if (nsh > 0) {
        const orbfe_status s = orbfe_extract_batch_device(m.ext, d_gray, nsh, w, ht, stride, fstride, m.d_kps + at * g->cap,
                                                          m.d_desc + at * g->cap * 32, g->cap, m.d_n + at, (void *)m.s_cmp);
        if (s != ORBFE_OK) return s;
    }
    ORBFE_HIP(hipEventRecord(m.ev_cmp, m.s_cmp));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_extract_batch(orbfe_group *g, const uint8_t *const *grays, int32_t nframes, int32_t w, int32_t ht,
                                                  int32_t stride)
{
    if (!g || !grays || nframes < 1 || nframes > g->world * g->shard || w < 1 || ht < 1 || stride < w) {
        orbfe_set_error("bad argument to orbfe_group_extract_batch (at most %d frames)", g ? g->world * g->shard : 0);
        return ORBFE_ERR_ARG;
    }
    GDeviceGuard guard;
    orbfe_status rs = ORBFE_OK;
    for (Member &m : g->mem) {
        int lo, hi;
        orbfe_group_shard_range(nframes, m.rank, g->world, &lo, &hi);
        const int nsh = hi - lo;
        if (hipSetDevice(m.device) != hipSuccess) { rs = ORBFE_ERR_HIP; break; }
        const size_t fbytes = (size_t)w * ht, need = fbytes * (size_t)std::max(nsh, 1);
        if (need > m.stage_bytes) {
            if (m.d_stage) {
                (void)hipStreamSynchronize(m.s_cmp);  // a previous extraction may still read the old staging block
                (void)hipFree(m.d_stage);
            }
            m.d_stage = nullptr;
            m.stage_bytes = 0;
            if (hipMalloc((void **)&m.d_stage, need + 64) != hipSuccess) { rs = ORBFE_ERR_NOMEM; break; }
            m.stage_bytes = need;
        }
        for (int f = 0; f < nsh && rs == ORBFE_OK; ++f)
            if (hipMemcpy2DAsync(m.d_stage + fbytes * f, (size_t)w, grays[lo + f], (size_t)stride, (size_t)w, (size_t)ht, hipMemcpyHostToDevice,
                                 m.s_cmp) != hipSuccess)
                rs = ORBFE_ERR_HIP;
        if (rs == ORBFE_OK) rs = extract_member(g, m, m.d_stage, nsh, w, ht, w, fbytes);
        if (rs != ORBFE_OK) break;
    }
    if (rs == ORBFE_OK) g->last_nframes = nframes;
    return rs;
}

extern "C" orbfe_status orbfe_group_extract_shard_device(orbfe_group *g, int32_t member, const uint8_t *d_gray, int32_t nframes_global,
                                                         int32_t w, int32_t ht, int32_t stride, size_t frame_stride)
{
    if (!g || member < 0 || member >= (int)g->mem.size() || nframes_global < 1 || nframes_global > g->world * g->shard) {
        orbfe_set_error("bad argument to orbfe_group_extract_shard_device");
        return ORBFE_ERR_ARG;
    }
    Member &m = g->mem[(size_t)member];
    int lo, hi;
    orbfe_group_shard_range(nframes_global, m.rank, g->world, &lo, &hi);
    if (hi > lo && !d_gray) { orbfe_set_error("orbfe_group_extract_shard_device: null frames"); return ORBFE_ERR_ARG; }
    GDeviceGuard guard;
    ORBFE_HIP(hipSetDevice(m.device));
    const orbfe_status s = extract_member(g, m, d_gray, hi - lo, w, ht, stride, frame_stride);
    if (s == ORBFE_OK) g->last_nframes = nframes_global;
    return s;
}

// the RCCL calls of one exchange step; the caller closes the NCCL group on every path
static orbfe_status allgather_rccl_enqueue(orbfe_group *g)
{
    for (Member &m : g->mem) {
        ORBFE_HIP(hipSetDevice(m.device));
        ORBFE_HIP(hipStreamWaitEvent(m.s_comm, m.ev_cmp, 0));
        const size_t at = (size_t)m.rank * g->shard, S = (size_t)g->shard;
        ORBFE_NCCL(rccl().AllGather(m.d_n + at, m.d_n, S * sizeof(int32_t), ncclUint8, m.comm, m.s_comm));
        ORBFE_NCCL(rccl().AllGather(m.d_kps + at * g->cap, m.d_kps, S * g->cap * sizeof(orbfe_keypoint), ncclUint8, m.comm, m.s_comm));
        ORBFE_NCCL(rccl().AllGather(m.d_desc + at * g->cap * 32, m.d_desc, S * g->cap * 32, ncclUint8, m.comm, m.s_comm));
    }
    return ORBFE_OK;
}

// the copy transport: member m pulls slice r of every other member o (rank r) into slice r of its own blocks -- the same
// bytes at the same offsets an in-place all-gather leaves there -- on m's communication stream, behind o's extraction
static orbfe_status allgather_copy_enqueue(orbfe_group *g)
{
    const size_t S = (size_t)g->shard;
    for (Member &m : g->mem) {
        ORBFE_HIP(hipSetDevice(m.device));
        for (Member &o : g->mem) ORBFE_HIP(hipStreamWaitEvent(m.s_comm, o.ev_cmp, 0));
        for (Member &o : g->mem) {
            if (&o == &m) continue;
            const size_t at = (size_t)o.rank * S;
            if (o.device == m.device) {
                ORBFE_HIP(hipMemcpyAsync(m.d_n + at, o.d_n + at, S * sizeof(int32_t), hipMemcpyDeviceToDevice, m.s_comm));
                ORBFE_HIP(hipMemcpyAsync(m.d_kps + at * g->cap, o.d_kps + at * g->cap, S * g->cap * sizeof(orbfe_keypoint), hipMemcpyDeviceToDevice, m.s_comm));
                ORBFE_HIP(hipMemcpyAsync(m.d_desc + at * g->cap * 32, o.d_desc + at * g->cap * 32, S * g->cap * 32, hipMemcpyDeviceToDevice, m.s_comm));
            } else {
                ORBFE_HIP(hipMemcpyPeerAsync(m.d_n + at, m.device, o.d_n + at, o.device, S * sizeof(int32_t), m.s_comm));
                ORBFE_HIP(hipMemcpyPeerAsync(m.d_kps + at * g->cap, m.device, o.d_kps + at * g->cap, o.device, S * g->cap * sizeof(orbfe_keypoint), m.s_comm));
                ORBFE_HIP(hipMemcpyPeerAsync(m.d_desc + at * g->cap * 32, m.device, o.d_desc + at * g->cap * 32, o.device, S * g->cap * 32, m.s_comm));
            }
        }
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_allgather(orbfe_group *g)
{
    if (!g) return ORBFE_ERR_ARG;
    GDeviceGuard guard;
    orbfe_status s;
    if (g->transport == ORBFE_GROUP_COPY) {
        s = allgather_copy_enqueue(g);
    } else {
        // in-place all-gather of the three padded blocks: every rank's slice sits at rank * shard of the receive buffer
        ORBFE_NCCL(rccl().GroupStart());
        s = allgather_rccl_enqueue(g);
        const ncclResult_t ne = rccl().GroupEnd();  // closed on the error path too: a failure must not leave the group open
        if (s == ORBFE_OK && ne != ncclSuccess) {
            orbfe_set_error("ncclGroupEnd failed: %s", rccl().GetErrorString ? rccl().GetErrorString(ne) : "?");
            s = ORBFE_ERR_HIP;
        }
    }
    if (s != ORBFE_OK) return s;
    for (Member &m : g->mem) {
        ORBFE_HIP(hipSetDevice(m.device));
        ORBFE_HIP(hipEventRecord(m.ev_comm, m.s_comm));
        ORBFE_HIP(hipStreamWaitEvent(m.s_cmp, m.ev_comm, 0));  // consumers on the compute stream see the gathered blocks
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_synchronize(orbfe_group *g)
{
    if (!g) return ORBFE_ERR_ARG;
    GDeviceGuard guard;
    for (Member &m : g->mem) {
        ORBFE_HIP(hipSetDevice(m.device));
        ORBFE_HIP(hipStreamSynchronize(m.s_cmp));
        ORBFE_HIP(hipStreamSynchronize(m.s_comm));
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_blocks(orbfe_group *g, int32_t member, int32_t **d_n, orbfe_keypoint **d_kps, uint8_t **d_desc,
                                           void **compute_stream)
{
    if (!g || member < 0 || member >= (int)g->mem.size()) return ORBFE_ERR_ARG;
    Member &m = g->mem[(size_t)member];
    if (d_n) *d_n = m.d_n;
    if (d_kps) *d_kps = m.d_kps;
    if (d_desc) *d_desc = m.d_desc;
    if (compute_stream) *compute_stream = (void *)m.s_cmp;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_get_frame_from(orbfe_group *g, int32_t member, int32_t frame, orbfe_keypoint *kps, uint8_t *desc,
                                                   int32_t cap, int32_t *n_out)
{
    if (!g || member < 0 || member >= (int)g->mem.size() || !n_out || frame < 0 || frame >= g->last_nframes) {
        orbfe_set_error("bad argument to orbfe_group_get_frame");
        return ORBFE_ERR_ARG;
    }
    Member &m = g->mem[(size_t)member];
    const size_t bi = (size_t)block_index(g, g->last_nframes, frame);
    GDeviceGuard guard;
    ORBFE_HIP(hipSetDevice(m.device));
    ORBFE_HIP(hipStreamSynchronize(m.s_cmp));
    ORBFE_HIP(hipStreamSynchronize(m.s_comm));
    int32_t n = 0;
    ORBFE_HIP(hipMemcpy(&n, m.d_n + bi, sizeof(int32_t), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) return ORBFE_ERR_CAP;
    if (n > 0) {
        if (!kps || !desc) return ORBFE_ERR_ARG;
        ORBFE_HIP(hipMemcpy(kps, m.d_kps + bi * g->cap, (size_t)n * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(desc, m.d_desc + bi * g->cap * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_get_counts(orbfe_group *g, int32_t member, int32_t *n_out /* world * shard */)
{
    if (!g || member < 0 || member >= (int)g->mem.size() || !n_out) { orbfe_set_error("bad argument to orbfe_group_get_counts"); return ORBFE_ERR_ARG; }
    Member &m = g->mem[(size_t)member];
    GDeviceGuard guard;
    ORBFE_HIP(hipSetDevice(m.device));
    ORBFE_HIP(hipStreamSynchronize(m.s_cmp));
    ORBFE_HIP(hipStreamSynchronize(m.s_comm));
    ORBFE_HIP(hipMemcpy(n_out, m.d_n, (size_t)g->world * g->shard * sizeof(int32_t), hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_group_get_frame(orbfe_group *g, int32_t frame, orbfe_keypoint *kps, uint8_t *desc, int32_t cap, int32_t *n_out)
{
    return orbfe_group_get_frame_from(g, 0, frame, kps, desc, cap, n_out);
}

// The consumer of the gather (src/LoopClosing.cc:312-342 does this serially per candidate keyframe): pair p matches frame
// qframe[p] -- which must belong to the shard of one of this process's members -- against frame tframe[p], any frame of
// the gathered batch; brute-force Hamming + ratio + rotation histogram (orbfe_match_bf_frames_device on the gathered block).
extern "C" orbfe_status orbfe_group_match(orbfe_group *g, const int32_t *qframe, const int32_t *tframe, int32_t npairs, float nnratio,
                                          int32_t th, int32_t check_ori, int32_t *match /* npairs x cap */, int32_t *nmatches /* npairs */)
{
    if (!g || npairs < 0 || (npairs > 0 && (!qframe || !tframe || !match || !nmatches)) || g->last_nframes < 1) {
        orbfe_set_error("bad argument to orbfe_group_match (call after orbfe_group_extract_* and orbfe_group_allgather)");
        return ORBFE_ERR_ARG;
    }
    const int nf = g->last_nframes;
    for (int p = 0; p < npairs; ++p)
        if (qframe[p] < 0 || qframe[p] >= nf || tframe[p] < 0 || tframe[p] >= nf) { orbfe_set_error("pair %d: frame out of range", p); return ORBFE_ERR_ARG; }
    GDeviceGuard guard;
    std::vector<int> owner_of((size_t)npairs, -1);
    std::vector<std::vector<int>> mine(g->mem.size());
    for (int p = 0; p < npairs; ++p) {
        const int r = owner_rank(g, nf, qframe[p]);
        for (size_t k = 0; k < g->mem.size(); ++k)
            if (g->mem[k].rank == r) { owner_of[(size_t)p] = (int)k; mine[k].push_back(p); }
        if (owner_of[(size_t)p] < 0) { orbfe_set_error("pair %d: query frame %d is not in a shard of this process", p, qframe[p]); return ORBFE_ERR_ARG; }
    }
    orbfe_status rs = ORBFE_OK;
    for (size_t k = 0; k < g->mem.size() && rs == ORBFE_OK; ++k) {
        Member &m = g->mem[k];
        const int np = (int)mine[k].size();
        if (np == 0) continue;
        ORBFE_HIP(hipSetDevice(m.device));
        if ((size_t)np > m.pairs_cap) {
            ORBFE_HIP(hipStreamSynchronize(m.s_cmp));
            for (void *b : {(void *)m.d_pairs, (void *)m.d_match, (void *)m.d_nm})
                if (b) (void)hipFree(b);
            m.d_pairs = m.d_match = m.d_nm = nullptr;
            m.pairs_cap = 0;
            ORBFE_HIP(hipMalloc((void **)&m.d_pairs, (size_t)np * 2 * sizeof(int32_t)));
            ORBFE_HIP(hipMalloc((void **)&m.d_match, (size_t)np * g->cap * sizeof(int32_t)));
            ORBFE_HIP(hipMalloc((void **)&m.d_nm, (size_t)np * sizeof(int32_t)));
            m.pairs_cap = (size_t)np;
        }
        std::vector<int32_t> qt((size_t)np * 2);
        for (int i = 0; i < np; ++i) {
            qt[(size_t)i] = block_index(g, nf, qframe[mine[k][(size_t)i]]);
            qt[(size_t)np + i] = block_index(g, nf, tframe[mine[k][(size_t)i]]);
        }
        ORBFE_HIP(hipMemcpyAsync(m.d_pairs, qt.data(), qt.size() * sizeof(int32_t), hipMemcpyHostToDevice, m.s_cmp));
        ORBFE_HIP(hipStreamSynchronize(m.s_cmp));  // qt is a stack vector; the gathered blocks are ordered behind ev_comm on s_cmp
        rs = orbfe_match_bf_frames_device(m.mat, m.d_kps, m.d_desc, m.d_n, g->cap, m.d_pairs, m.d_pairs + np, np, nnratio, th, check_ori,
                                          m.d_match, m.d_nm, (void *)m.s_cmp);
        if (rs != ORBFE_OK) break;
        std::vector<int32_t> hm((size_t)np * g->cap), hn((size_t)np);
        ORBFE_HIP(hipMemcpyAsync(hm.data(), m.d_match, hm.size() * sizeof(int32_t), hipMemcpyDeviceToHost, m.s_cmp));
        ORBFE_HIP(hipMemcpyAsync(hn.data(), m.d_nm, hn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, m.s_cmp));
        ORBFE_HIP(hipStreamSynchronize(m.s_cmp));
        for (int i = 0; i < np; ++i) {
            const int p = mine[k][(size_t)i];
            memcpy(match + (size_t)p * g->cap, hm.data() + (size_t)i * g->cap, (size_t)g->cap * sizeof(int32_t));
            nmatches[p] = hn[(size_t)i];
        }
    }
    return rs;
}

// device-resident form for one member (bench.py): pairs as BLOCK indices already on the device, results stay there
extern "C" orbfe_status orbfe_group_match_device(orbfe_group *g, int32_t member, const int32_t *d_qblock, const int32_t *d_tblock,
                                                 int32_t npairs, float nnratio, int32_t th, int32_t check_ori, int32_t *d_match,
                                                 int32_t *d_nmatches)
{
    if (!g || member < 0 || member >= (int)g->mem.size()) return ORBFE_ERR_ARG;
    Member &m = g->mem[(size_t)member];
    GDeviceGuard guard;
    ORBFE_HIP(hipSetDevice(m.device));
    return orbfe_match_bf_frames_device(m.mat, m.d_kps, m.d_desc, m.d_n, g->cap, d_qblock, d_tblock, npairs, nnratio, th, check_ori,
                                        d_match, d_nmatches, (void *)m.s_cmp);
}
