// dist.hip — multi-GPU behind the C ABI: shard a batch of images over the ranks of one node (one process per GPU) and
// all-gather the fixed-size per-image records over RCCL/xGMI.
//
// New functionality asked for by the north star (the reference is single image / single device; the host that would call
// it is the evaluate loop of Sources/maskrcnn/EvaluateCommand.swift:146-179).  Each image's pipeline is independent, so
// the batch is split into contiguous blocks (blocks differ by at most one image), weights and anchors are replicated,
// there is NO collective on the data path, and the only exchange is ONE ncclAllGather of zero-padded records
//     record = detections (maxDet × 6 f32) ‖ masks (maxDet × S × S f32)          316 000 B at the defaults
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU host needs no RCCL, and a process that already holds
// a copy (torch ships one) shares it.  The torch.distributed twin of this file is mask-rcnn-coreml_amd/dist.py.
//
// What a rank sends is a SLOT: its records, zero-padded to the largest shard, followed by a 4-word trailer
//     [status, images in the shard, 0, 0]
// so that a rank whose local predict failed (a HIP error, the fp16-range watchdog — data dependent) still takes part in
// the collective — with zeroed records and its status code — and EVERY rank raises after the gather.  Skipping the
// collective on a local failure would leave the other ranks blocked in ncclAllGather for ever.
//
// Layout arithmetic (slot size, where rank r's records start in the gathered buffer, which image rows they become) lives
// in ONE function, plan_entries(), used by the device path and by the host seam mrcnn_dist_simulate_host (tests drive the
// pack → concatenate → unpack code at world sizes 1–8 without a GPU and compare with dist.py).
//
// Overlap: mrcnn_dist_all_gather_records_async issues pack / all-gather / unpack on the handle's own stream behind an
// event on the model's stream and returns; the model's next predict runs under it (the pack reads the caller's result
// buffers: the model's stream waits for the pack, not for the exchange, before it may overwrite them).  mrcnn_dist_wait
// joins and raises the remote statuses.
#include <dlfcn.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <vector>

#include "engine.h"

using namespace mrcnn;

namespace {

// the slice of rccl.h this file needs (ABI of RCCL 2.x: NCCL_UNIQUE_ID_BYTES = 128, ncclFloat = 7, ncclSuccess = 0)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommAbort)(NcclComm) = nullptr;          // optional (last resort of a rank that cannot take part in a collective)
    bool shared = false;                           // the copy the process had already mapped (RTLD_NOLOAD), e.g. PyTorch's
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl()
{
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (r.lib) return r;
    // resolved into a local table first: a missing symbol must not leave a half-bound table behind
    Rccl t;
    // ONE RCCL per process: a copy the host has already mapped — torch's process group loads one under the
    // same soname — is taken first (RTLD_NOLOAD: succeeds only when the library is resident); only a process without one
    // loads its own.  MRCNN_RCCL_PRIVATE=1 skips the first step (tests).
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const char* priv = getenv("MRCNN_RCCL_PRIVATE");
    if (!(priv && atoi(priv) != 0))
        for (const char* n : names) {
            t.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (t.lib) { t.shared = true; break; }
        }
    for (const char* n : names) {
        if (t.lib) break;
        t.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    MRCNN_REQUIRE(t.lib, MRCNN_ERR_CONFIG, "cannot load RCCL (librccl.so.1): %s", dlerror());
    const char* missing = nullptr;
    auto sym = [&](const char* s) {
        void* p = dlsym(t.lib, s);
        if (!p && !missing) missing = s;
        return p;
    };
    t.GetUniqueId = reinterpret_cast<decltype(t.GetUniqueId)>(sym("ncclGetUniqueId"));
    t.CommInitRank = reinterpret_cast<decltype(t.CommInitRank)>(sym("ncclCommInitRank"));
    t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(sym("ncclCommDestroy"));
    t.AllGather = reinterpret_cast<decltype(t.AllGather)>(sym("ncclAllGather"));
    t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(sym("ncclGetErrorString"));
    t.CommAbort = reinterpret_cast<decltype(t.CommAbort)>(dlsym(t.lib, "ncclCommAbort"));      // optional
    if (missing) {
        dlclose(t.lib);
        fail(MRCNN_ERR_CONFIG, "RCCL lacks %s", missing);
    }
    r = t;
    return r;
}
void nccl_check(const Rccl& r, int rc, const char* what)
{
    if (rc != 0) fail(MRCNN_ERR_HIP, "%s failed: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "?");
}

void shard(int total, int world, int rank, int* lo, int* hi)
{
    const int base = total / world, rem = total % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

// ---- the layout of one exchange --------------------------------------------------------------------------------------
constexpr int TRAILER = 4;        // floats: [status, images in the shard, range recoveries of the rank's predict, 0] (int32 bit patterns)
struct Geometry {
    size_t det_len, mask_len, rec;   // floats per image
    int n_max;                       // images of the largest shard
    size_t slot;                     // floats a rank sends: n_max records + trailer
};
Geometry geometry(int global_batch, int world, int max_det, int mask_size)
{
    Geometry g;
    g.det_len = (size_t)max_det * 6;
    g.mask_len = (size_t)max_det * mask_size * mask_size;
    g.rec = g.det_len + g.mask_len;
    g.n_max = (global_batch + world - 1) / world;
    g.slot = (size_t)g.n_max * g.rec + TRAILER;
    return g;
}
struct PlanEntry { int begin, end; size_t recv_off; };     // rank r: its images [begin, end); its slot starts at recv_off floats
std::vector<PlanEntry> plan_entries(int global_batch, int world, const Geometry& g)
{
    std::vector<PlanEntry> p((size_t)world);
    for (int r = 0; r < world; ++r) {
        shard(global_batch, world, r, &p[r].begin, &p[r].end);
        p[r].recv_off = (size_t)r * g.slot;
    }
    return p;
}

// pack / unpack are written once over a 2-D copy primitive: hipMemcpy2DAsync on the device path, memcpy rows on the host seam
struct HostCopy {
    void zero(float* dst, size_t n) const { memset(dst, 0, n * 4); }
    void rows(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t n) const
    {
        for (size_t i = 0; i < n; ++i) memcpy(dst + i * dpitch, src + i * spitch, width * 4);
    }
    void words(float* dst, const int32_t* w, int n) const { memcpy(dst, w, (size_t)n * 4); }
};
struct DeviceCopy {
    hipStream_t s;
    hipMemcpyKind in_kind, out_kind;
    bool unpacking = false;
    void zero(float* dst, size_t n) const { HIP_CHECK(hipMemsetAsync(dst, 0, n * 4, s)); }
    void rows(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t n) const
    {
        HIP_CHECK(hipMemcpy2DAsync(dst, dpitch * 4, src, spitch * 4, width * 4, n, unpacking ? out_kind : in_kind, s));
    }
    void words(float* dst, const int32_t* w, int n) const { HIP_CHECK(hipMemcpyAsync(dst, w, (size_t)n * 4, hipMemcpyHostToDevice, s)); }
};

// this rank's slot: records (zero-padded to n_max) + trailer.  status != 0: the records are sent as zeros.
template <class Copy>
void pack_slot(const Copy& c, const Geometry& g, int n_local, const float* det, const float* masks, const int32_t trailer[TRAILER], float* send)
{
    c.zero(send, g.slot);
    if (n_local > 0 && trailer[0] == 0) {
        c.rows(send, g.rec, det, g.det_len, g.det_len, (size_t)n_local);
        c.rows(send + g.det_len, g.rec, masks, g.mask_len, g.mask_len, (size_t)n_local);
    }
    c.words(send + (size_t)g.n_max * g.rec, trailer, TRAILER);
}
// gathered slots → results in global image order, padding dropped
template <class Copy>
void unpack_slots(const Copy& c, const Geometry& g, const std::vector<PlanEntry>& plan, const float* recv, float* out_det, float* out_masks)
{
    for (const PlanEntry& e : plan) {
        if (e.end == e.begin) continue;
        const float* src = recv + e.recv_off;
        c.rows(out_det + (size_t)e.begin * g.det_len, g.det_len, src, g.rec, g.det_len, (size_t)(e.end - e.begin));
        c.rows(out_masks + (size_t)e.begin * g.mask_len, g.mask_len, src + g.det_len, g.rec, g.mask_len, (size_t)(e.end - e.begin));
    }
}

}  // namespace

struct mrcnn_dist {
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t gs = nullptr;                 // the exchange's own stream (async form)
    hipEvent_t ev_ready = nullptr, ev_packed = nullptr;
    DevBuf send, recv, stage_det, stage_mask, stage_img;
    int32_t trailer[TRAILER] = {0, 0, 0, 0};  // host copy of what was packed (outlives the async H2D)
    std::vector<int32_t> statuses;            // host copy of every rank's trailer after an exchange
    bool pending = false;                     // an async exchange has been issued and not yet joined
    Geometry g{};
    std::vector<PlanEntry> plan;
};

extern "C" int mrcnn_dist_shard(int global_batch, int world, int rank, int* begin, int* end)
{
    return guarded([&] {
        MRCNN_REQUIRE(begin && end && global_batch >= 0 && world >= 1 && rank >= 0 && rank < world, MRCNN_ERR_INVALID, "bad dist_shard argument");
        shard(global_batch, world, rank, begin, end);
    });
}

extern "C" int64_t mrcnn_dist_record_floats(int max_detections, int mask_size)
{
    if (max_detections < 0 || mask_size < 0) return -1;
    return (int64_t)max_detections * 6 + (int64_t)max_detections * mask_size * mask_size;
}

extern "C" int mrcnn_dist_plan(int global_batch, int world, int max_detections, int mask_size, int64_t* table, int64_t* slot_floats)
{
    return guarded([&] {
        MRCNN_REQUIRE(table && global_batch >= 0 && world >= 1 && max_detections >= 0 && mask_size >= 0, MRCNN_ERR_INVALID, "bad dist_plan argument");
        const Geometry g = geometry(global_batch, world, max_detections, mask_size);
        const auto plan = plan_entries(global_batch, world, g);
        for (int r = 0; r < world; ++r) {
            table[4 * r + 0] = plan[r].begin;
            table[4 * r + 1] = plan[r].end;
            table[4 * r + 2] = (int64_t)plan[r].recv_off;
            table[4 * r + 3] = (int64_t)(plan[r].end - plan[r].begin) * (int64_t)g.rec;
        }
        if (slot_floats) *slot_floats = (int64_t)g.slot;
    });
}

extern "C" int mrcnn_dist_simulate_host(int world, int global_batch, int max_detections, int mask_size, const float* const* detections,
                                        const float* const* masks, const int32_t* status, float* out_detections, float* out_masks,
                                        int32_t* status_out)
{
    return guarded([&] {
        MRCNN_REQUIRE(detections && masks && out_detections && out_masks && world >= 1 && global_batch >= 1, MRCNN_ERR_INVALID,
                      "bad dist_simulate_host argument");
        const Geometry g = geometry(global_batch, world, max_detections, mask_size);
        const auto plan = plan_entries(global_batch, world, g);
        static_assert(TRAILER >= 3, "status | records | range recoveries of the rank's predict");
        std::vector<float> gathered((size_t)world * g.slot, -1.0f);       // poisoned: every word must come from a pack
        HostCopy c;
        for (int r = 0; r < world; ++r) {
            const int n = plan[r].end - plan[r].begin;
            // status MRCNN_DIST_ABORTED: a rank that could not even enqueue a zeroed slot and tore the communicator down (release_peers) —
            // its peers' ncclAllGather then FAILS (nccl_check raises on each of them) instead of blocking; nothing is unpacked
            MRCNN_REQUIRE(!(status && status[r] == MRCNN_DIST_ABORTED), MRCNN_ERR_HIP,
                          "ncclAllGather failed: rank %d aborted the communicator (it could not take part in the all-gather)", r);
            MRCNN_REQUIRE(n == 0 || (detections[r] && masks[r]), MRCNN_ERR_INVALID, "rank %d: null local results", r);
            const int32_t tr[TRAILER] = {status ? status[r] : 0, n, 0, 0};
            pack_slot(c, g, n, detections[r], masks[r], tr, gathered.data() + plan[r].recv_off);      // = rank r's ncclAllGather contribution
        }
        unpack_slots(c, g, plan, gathered.data(), out_detections, out_masks);
        if (status_out)
            for (int r = 0; r < world; ++r) memcpy(&status_out[r], gathered.data() + plan[r].recv_off + (size_t)g.n_max * g.rec, 4);
    });
}

extern "C" int mrcnn_dist_rccl_shared(void)
{
    int shared = -1;
    (void)guarded([&] { shared = rccl().shared ? 1 : 0; });
    return shared;
}

extern "C" int mrcnn_dist_unique_id(uint8_t* id128)
{
    return guarded([&] {
        MRCNN_REQUIRE(id128, MRCNN_ERR_INVALID, "null id buffer");
        require_gpu();
        NcclUniqueId id;
        Rccl& r = rccl();
        nccl_check(r, r.GetUniqueId(&id), "ncclGetUniqueId");
        memcpy(id128, id.internal, 128);
    });
}

extern "C" int mrcnn_dist_init(int rank, int world, const uint8_t* id128, mrcnn_dist** out)
{
    return guarded([&] {
        MRCNN_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, MRCNN_ERR_INVALID, "bad dist_init argument");
        require_gpu();
        std::unique_ptr<mrcnn_dist> d(new mrcnn_dist);
        d->rank = rank; d->world = world;
        NcclUniqueId id;
        memcpy(id.internal, id128, 128);
        Rccl& r = rccl();
        nccl_check(r, r.CommInitRank(&d->comm, world, id, rank), "ncclCommInitRank");
        HIP_CHECK(hipStreamCreateWithFlags(&d->gs, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&d->ev_packed, hipEventDisableTiming));
        *out = d.release();
    });
}

extern "C" void mrcnn_dist_destroy(mrcnn_dist* d)
{
    if (!d) return;
    if (d->gs) (void)hipStreamSynchronize(d->gs);
    if (d->comm) (void)rccl().CommDestroy(d->comm);
    if (d->ev_ready) (void)hipEventDestroy(d->ev_ready);
    if (d->ev_packed) (void)hipEventDestroy(d->ev_packed);
    if (d->gs) (void)hipStreamDestroy(d->gs);
    delete d;
}

// Every rank's trailer after the exchange → raise on ALL ranks when any rank reported a failure.
static void raise_remote_status(mrcnn_dist* d)
{
    for (int r = 0; r < d->world; ++r) {
        const int st = d->statuses[(size_t)r * TRAILER];
        if (st != 0)
            fail(st > 0 && st <= MRCNN_ERR_CONFIG ? st : MRCNN_ERR_HIP,
                 "rank %d of %d failed its local predict (status %d%s); the records of this batch are not valid on any rank", r, d->world, st,
                 r == d->rank ? ": see this rank's earlier message" : "");
    }
}

// Buffers of an exchange.  Their sizes depend only on arguments every rank passes alike (record geometry, global batch, world),
// so an allocation failure here is the same failure on every rank — it happens BEFORE the window in which a rank-local error
// would leave the others alone in the collective.
static void reserve_exchange(mrcnn_dist* d, const Geometry& g)
{
    if (d->send.bytes < g.slot * 4) d->send.alloc(g.slot * 4);
    if (d->recv.bytes < (size_t)d->world * g.slot * 4) d->recv.alloc((size_t)d->world * g.slot * 4);
    if (d->statuses.size() != (size_t)d->world * TRAILER) d->statuses.assign((size_t)d->world * TRAILER, 0);    // never re-assigned under an in-flight D2H
}

// Last resort of a rank that cannot reach the collective: tear the communicator down so that the peers' ncclAllGather FAILS instead of
// blocking for ever — ncclCommAbort where librccl exports it, else ncclCommDestroy (which may itself wait for the peers: the documented
// worst case is then a hang of THIS rank, which the job's launcher times out, rather than of every other rank).
static void release_peers(mrcnn_dist* d)
{
    Rccl& r = rccl();
    if (!d->comm) return;
    if (r.CommAbort) (void)r.CommAbort(d->comm);
    else if (r.CommDestroy) (void)r.CommDestroy(d->comm);
    d->comm = nullptr;
}

// local results (n_local records, `in_space`) → every rank's records in global image order (`out_space`).
// Issued on stream `s`; `status` != 0 sends zeroed records.  Does not synchronise.
// The contract: once the arguments every rank sees alike have been checked, NOTHING rank-local may keep this
// rank out of ncclAllGather — a rank-local error (null results, a failing pack copy after a sticky HIP error) becomes the status
// word of a zeroed slot; only when even that cannot be enqueued is the communicator aborted (ncclCommAbort: the peers' collective
// then fails instead of blocking for ever) and the error raised.  *enqueued is set as soon as work targets d's buffers.
static void issue_exchange(mrcnn_dist* d, Model& m, hipStream_t s, const float* det, const float* masks, int in_space, int global_batch,
                           int out_space, float* out_det, float* out_masks, int status, bool* enqueued = nullptr, int recovered = 0)
{
    MRCNN_REQUIRE(m.kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "all_gather_records needs the MaskRCNN model (record geometry)");
    d->g = geometry(global_batch, d->world, m.max_det, 2 * m.mask_pool);
    d->plan = plan_entries(global_batch, d->world, d->g);
    const Geometry& g = d->g;
    const int n_local = d->plan[d->rank].end - d->plan[d->rank].begin;
    // (rank-uniform prologue: a handle whose communicator an earlier failure aborted is refused HERE, before anything is enqueued
    //  and before a peer could be left alone in the collective — every rank of that job saw the same abort)
    MRCNN_REQUIRE(d->comm, MRCNN_ERR_INVALID, "the communicator of this handle was aborted by an earlier failure");
    reserve_exchange(d, g);
    // ---- from here on: rank-local failures travel as the status word ------------------------------------------------
    std::string local_msg;
    if (status == 0 && n_local > 0 && !(det && masks)) { status = MRCNN_ERR_INVALID; local_msg = "null local results"; }
    DeviceCopy c{s, in_space != MRCNN_DEVICE ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                 out_space != MRCNN_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice};
    if (enqueued) *enqueued = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // word 2 (round 6): this rank's predict lowered its split exponents (range recovery) — from then on its bits for a given
        // image differ from its peers', which is what a sharded job that wants rank-independent results must know: mrcnn_dist_recovered
        d->trailer[0] = status; d->trailer[1] = n_local; d->trailer[2] = recovered; d->trailer[3] = 0;
        try {
            pack_slot(c, g, n_local, det, masks, d->trailer, d->send.as<float>());
            break;
        } catch (const Error& e) {
            (void)hipGetLastError();
            if (attempt == 1) {          // not even a zeroed slot can be enqueued: release the peers, then raise
                release_peers(d);
                fail(e.code ? e.code : MRCNN_ERR_HIP, "rank %d cannot take part in the all-gather (%s); communicator aborted", d->rank, e.msg.c_str());
            }
            status = e.code ? e.code : MRCNN_ERR_HIP;      // second attempt: zeroed records + this status
            local_msg = e.msg;
        }
    }
    {
        // a failing event record is a rank-local failure like a failing pack: the peers must not block in the collective
        const hipError_t ee = hipEventRecord(d->ev_packed, s);
        if (ee != hipSuccess) {
            release_peers(d);
            fail(MRCNN_ERR_HIP, "rank %d: hipEventRecord before the all-gather failed (%s); communicator aborted", d->rank, hipGetErrorString(ee));
        }
    }
    // ---- the one collective of the path ---------------------------------------------------------------------------
    Rccl& r = rccl();
    nccl_check(r, r.AllGather(d->send.p, d->recv.p, g.slot, /*ncclFloat*/ 7, d->comm, s), "ncclAllGather");
    c.unpacking = true;
    unpack_slots(c, g, d->plan, d->recv.as<float>(), out_det, out_masks);
    // every rank's trailer → host (one strided copy)
    HIP_CHECK(hipMemcpy2DAsync(d->statuses.data(), TRAILER * 4, d->recv.as<float>() + (size_t)g.n_max * g.rec, g.slot * 4, TRAILER * 4,
                               (size_t)d->world, hipMemcpyDeviceToHost, s));
    if (!local_msg.empty()) set_error("rank %d: %s (sent as the status word of this rank's slot)", d->rank, local_msg.c_str());
}

extern "C" int mrcnn_dist_recovered(mrcnn_dist* d, int32_t* per_rank, int* ranks)
{
    return guarded([&] {
        MRCNN_REQUIRE(d && ranks, MRCNN_ERR_INVALID, "null argument");
        MRCNN_REQUIRE(!d->pending, MRCNN_ERR_INVALID, "an exchange is still pending: call mrcnn_dist_wait first");
        int n = 0;
        for (int r = 0; r < d->world; ++r) {
            const int v = d->statuses.size() == (size_t)d->world * TRAILER ? d->statuses[(size_t)r * TRAILER + 2] : 0;
            if (per_rank) per_rank[r] = v;
            n += v != 0;
        }
        *ranks = n;
    });
}

extern "C" int mrcnn_dist_wait(mrcnn_dist* d)
{
    return guarded([&] {
        MRCNN_REQUIRE(d, MRCNN_ERR_INVALID, "null dist handle");
        if (!d->pending) return;
        d->pending = false;
        HIP_CHECK(hipStreamSynchronize(d->gs));
        raise_remote_status(d);
    });
}

extern "C" int mrcnn_dist_all_gather_records_async(mrcnn_dist* d, mrcnn_model* model, const float* det, const float* masks, int global_batch,
                                                   float* out_det, float* out_masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(d && model && out_det && out_masks && global_batch >= 1, MRCNN_ERR_INVALID, "bad all_gather_records_async argument");
        MRCNN_REQUIRE(!d->pending, MRCNN_ERR_INVALID, "an exchange is still pending: call mrcnn_dist_wait first");
        Model& m = model->m;
        // the exchange starts when the model's stream has produced the results ...
        HIP_CHECK(hipEventRecord(d->ev_ready, m.stream));
        HIP_CHECK(hipStreamWaitEvent(d->gs, d->ev_ready, 0));
        bool enqueued = false;
        try {
            issue_exchange(d, m, d->gs, det, masks, MRCNN_DEVICE, global_batch, MRCNN_DEVICE, out_det, out_masks, 0, &enqueued);
        } catch (...) {
            // work may already target d's buffers and the caller's outputs: join it before the error leaves (a later exchange
            // must never re-use them under an in-flight copy — ADVICE r3)
            if (enqueued) (void)hipStreamSynchronize(d->gs);
            throw;
        }
        d->pending = true;
        // ... and the model's stream may overwrite them (the next predict) once they are packed — not once they are exchanged
        HIP_CHECK(hipStreamWaitEvent(m.stream, d->ev_packed, 0));
    });
}

extern "C" int mrcnn_dist_all_gather_records(mrcnn_dist* d, mrcnn_model* model, const float* det, const float* masks, int global_batch,
                                             int memspace, float* out_det, float* out_masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(d && model && out_det && out_masks && global_batch >= 1, MRCNN_ERR_INVALID, "bad all_gather_records argument");
        MRCNN_REQUIRE(!d->pending, MRCNN_ERR_INVALID, "an exchange is still pending: call mrcnn_dist_wait first");
        issue_exchange(d, model->m, model->m.stream, det, masks, memspace, global_batch, memspace, out_det, out_masks, 0);
        HIP_CHECK(hipStreamSynchronize(model->m.stream));
        raise_remote_status(d);
    });
}

extern "C" int mrcnn_maskrcnn_predict_sharded(mrcnn_dist* d, mrcnn_model* model, const uint8_t* rgb, int global_batch, int height, int width,
                                              int memspace, float* detections, float* masks)
{
    return guarded([&] {
        // argument errors every rank sees identically (same call on every rank): raised before any work, on all ranks alike
        MRCNN_REQUIRE(d && model && rgb && detections && masks && global_batch >= 1, MRCNN_ERR_INVALID, "bad predict_sharded argument");
        MRCNN_REQUIRE(!d->pending, MRCNN_ERR_INVALID, "an exchange is still pending: call mrcnn_dist_wait first");
        Model& m = model->m;
        MRCNN_REQUIRE(m.kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "predict_sharded called on a non-MaskRCNN model");
        const int n_max = (global_batch + d->world - 1) / d->world;
        MRCNN_REQUIRE(n_max <= m.max_batch, MRCNN_ERR_SHAPE, "the largest shard (%d images of %d over %d ranks) exceeds the model's max_batch %d",
                      n_max, global_batch, d->world, m.max_batch);
        MRCNN_REQUIRE(height == m.H && width == m.W, MRCNN_ERR_SHAPE, "image is %dx%d, the model expects %dx%d", height, width, m.H, m.W);
        int lo, hi;
        shard(global_batch, d->world, d->rank, &lo, &hi);
        const int n = hi - lo;
        const int D = m.max_det, S = 2 * m.mask_pool;
        // From here on a failure is LOCAL (a HIP error, the data-dependent fp16-range watchdog): it must not keep this rank out
        // of the collective.  It becomes the status word of this rank's slot, and every rank raises after the gather.
        int status = 0;
        std::string local_msg;
        const long recoveries_before = m.range_recoveries;
        try {
            const size_t nd = (size_t)(n > 0 ? n : 1) * D * 6 * 4, nm = (size_t)(n > 0 ? n : 1) * D * S * S * 4;
            if (d->stage_det.bytes < nd) d->stage_det.alloc(nd);
            if (d->stage_mask.bytes < nm) d->stage_mask.alloc(nm);
            if (n > 0) {
                const uint8_t* src = rgb + (size_t)lo * height * width * 3;
                if (memspace == MRCNN_DEVICE) {
                    m.predict(src, n, height, width, MRCNN_DEVICE, d->stage_det.as<float>(), d->stage_mask.as<float>(), true);
                } else {
                    const size_t ib = (size_t)n * height * width * 3;
                    if (d->stage_img.bytes < ib) d->stage_img.alloc(ib);
                    HIP_CHECK(hipMemcpy(d->stage_img.p, src, ib, hipMemcpyHostToDevice));
                    m.predict(d->stage_img.as<uint8_t>(), n, height, width, MRCNN_DEVICE, d->stage_det.as<float>(), d->stage_mask.as<float>(), true);
                }
            }
        } catch (const Error& e) {
            status = e.code ? e.code : MRCNN_ERR_HIP;
            local_msg = e.msg;
        } catch (const std::exception& e) {
            status = MRCNN_ERR_INVALID;
            local_msg = e.what();
        }
        issue_exchange(d, m, m.stream, d->stage_det.as<float>(), d->stage_mask.as<float>(), MRCNN_DEVICE, global_batch, memspace, detections, masks,
                       status, nullptr, (int)(m.range_recoveries - recoveries_before));
        HIP_CHECK(hipStreamSynchronize(m.stream));
        if (status != 0) fail(status, "rank %d: %s (the other ranks were told through the status word of this rank's slot)", d->rank, local_msg.c_str());
        raise_remote_status(d);
    });
}
