// dist.hip — multi-GPU behind the C ABI: shard a batch of images over the ranks of one node (one process per GPU) and
// all-gather the fixed-size per-image records over RCCL/xGMI.
//
// New functionality asked for by the north star (the reference is single image / single device; the host that would call
// it is the evaluate loop of Sources/maskrcnn/EvaluateCommand.swift:146-179).  Each image's pipeline is independent, so
// the batch is split into contiguous blocks (blocks differ by at most one image), weights and anchors are replicated,
// there is NO collective on the data path, and the only exchange is ONE ncclAllGather of zero-padded records
//     record = detections (maxDet × 6 f32) ‖ masks (maxDet × S × S f32)          316 000 B at the defaults
// issued on the model's own stream.  RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU host needs no RCCL,
// and a process that already holds a copy (torch ships one) shares it.  The torch.distributed twin of this file is
// mask-rcnn-coreml_amd/dist.py (used for the gloo/CPU tests of the same shard arithmetic and record layout).
#include <dlfcn.h>
#include <string.h>

#include <memory>
#include <mutex>

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
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl()
{
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (r.lib) return r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    MRCNN_REQUIRE(r.lib, MRCNN_ERR_CONFIG, "cannot load RCCL (librccl.so.1): %s", dlerror());
    auto sym = [&](const char* s) {
        void* p = dlsym(r.lib, s);
        MRCNN_REQUIRE(p, MRCNN_ERR_CONFIG, "RCCL lacks %s", s);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}
void nccl_check(int rc, const char* what)
{
    if (rc != 0) fail(MRCNN_ERR_HIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
}

void shard(int total, int world, int rank, int* lo, int* hi)
{
    const int base = total / world, rem = total % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

}  // namespace

struct mrcnn_dist {
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    DevBuf send, recv, stage_det, stage_mask, out_det, out_mask;
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

extern "C" int mrcnn_dist_unique_id(uint8_t* id128)
{
    return guarded([&] {
        MRCNN_REQUIRE(id128, MRCNN_ERR_INVALID, "null id buffer");
        require_gpu();
        NcclUniqueId id;
        nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
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
        nccl_check(rccl().CommInitRank(&d->comm, world, id, rank), "ncclCommInitRank");
        *out = d.release();
    });
}

extern "C" void mrcnn_dist_destroy(mrcnn_dist* d)
{
    if (!d) return;
    if (d->comm) (void)rccl().CommDestroy(d->comm);
    delete d;
}

// local results (n_local records, `in_space`) → every rank's records in global image order (`out_space`)
static void gather_records(mrcnn_dist* d, Model& m, const float* det, const float* masks, int in_space, int global_batch, int out_space,
                           float* out_det, float* out_masks)
{
    MRCNN_REQUIRE(m.kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "all_gather_records needs the MaskRCNN model (record geometry)");
    const int D = m.max_det, S = 2 * m.mask_pool;
    const size_t det_len = (size_t)D * 6, mask_len = (size_t)D * S * S, rec = det_len + mask_len;
    int lo, hi;
    shard(global_batch, d->world, d->rank, &lo, &hi);
    const int n_local = hi - lo, n_max = (global_batch + d->world - 1) / d->world;
    MRCNN_REQUIRE(n_local == 0 || (det && masks), MRCNN_ERR_INVALID, "null local results");
    hipStream_t s = m.stream;
    // ---- pack: records of this rank, zero-padded to the largest shard ----------------------------------------------
    if (d->send.bytes < (size_t)n_max * rec * 4) d->send.alloc((size_t)n_max * rec * 4);
    if (d->recv.bytes < (size_t)d->world * n_max * rec * 4) d->recv.alloc((size_t)d->world * n_max * rec * 4);
    HIP_CHECK(hipMemsetAsync(d->send.p, 0, (size_t)n_max * rec * 4, s));
    if (n_local > 0) {
        const hipMemcpyKind k = in_space != MRCNN_DEVICE ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        HIP_CHECK(hipMemcpy2DAsync(d->send.p, rec * 4, det, det_len * 4, det_len * 4, (size_t)n_local, k, s));
        HIP_CHECK(hipMemcpy2DAsync(d->send.as<float>() + det_len, rec * 4, masks, mask_len * 4, mask_len * 4, (size_t)n_local, k, s));
    }
    // ---- the one collective of the path, on the model's stream -----------------------------------------------------
    nccl_check(rccl().AllGather(d->send.p, d->recv.p, (size_t)n_max * rec, /*ncclFloat*/ 7, d->comm, s), "ncclAllGather");
    // ---- unpack in global image order, padding dropped -------------------------------------------------------------
    const hipMemcpyKind k = out_space != MRCNN_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    for (int r = 0; r < d->world; ++r) {
        int rlo, rhi;
        shard(global_batch, d->world, r, &rlo, &rhi);
        if (rhi == rlo) continue;
        const float* src = d->recv.as<float>() + (size_t)r * n_max * rec;
        HIP_CHECK(hipMemcpy2DAsync(out_det + (size_t)rlo * det_len, det_len * 4, src, rec * 4, det_len * 4, (size_t)(rhi - rlo), k, s));
        HIP_CHECK(hipMemcpy2DAsync(out_masks + (size_t)rlo * mask_len, mask_len * 4, src + det_len, rec * 4, mask_len * 4, (size_t)(rhi - rlo), k, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
}

extern "C" int mrcnn_dist_all_gather_records(mrcnn_dist* d, mrcnn_model* model, const float* det, const float* masks, int global_batch,
                                             int memspace, float* out_det, float* out_masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(d && model && out_det && out_masks && global_batch >= 1, MRCNN_ERR_INVALID, "bad all_gather_records argument");
        gather_records(d, model->m, det, masks, memspace, global_batch, memspace, out_det, out_masks);
    });
}

extern "C" int mrcnn_maskrcnn_predict_sharded(mrcnn_dist* d, mrcnn_model* model, const uint8_t* rgb, int global_batch, int height, int width,
                                              int memspace, float* detections, float* masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(d && model && rgb && detections && masks && global_batch >= 1, MRCNN_ERR_INVALID, "bad predict_sharded argument");
        Model& m = model->m;
        MRCNN_REQUIRE(m.kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "predict_sharded called on a non-MaskRCNN model");
        int lo, hi;
        shard(global_batch, d->world, d->rank, &lo, &hi);
        const int n = hi - lo;
        MRCNN_REQUIRE(n <= m.max_batch, MRCNN_ERR_SHAPE, "shard of %d images exceeds the model's max_batch %d", n, m.max_batch);
        const int D = m.max_det, S = 2 * m.mask_pool;
        // this rank's block of the batch → its results stay on the device between predict and the gather
        const size_t nd = (size_t)(n > 0 ? n : 1) * D * 6 * 4, nm = (size_t)(n > 0 ? n : 1) * D * S * S * 4;
        if (d->stage_det.bytes < nd) d->stage_det.alloc(nd);
        if (d->stage_mask.bytes < nm) d->stage_mask.alloc(nm);
        if (n > 0) {
            const uint8_t* src = rgb + (size_t)lo * height * width * 3;
            if (memspace == MRCNN_DEVICE) {
                m.predict(src, n, height, width, MRCNN_DEVICE, d->stage_det.as<float>(), d->stage_mask.as<float>(), true);
            } else {
                if (d->out_det.bytes < (size_t)n * height * width * 3) d->out_det.alloc((size_t)n * height * width * 3);   // image staging
                HIP_CHECK(hipMemcpy(d->out_det.p, src, (size_t)n * height * width * 3, hipMemcpyHostToDevice));
                m.predict(d->out_det.as<uint8_t>(), n, height, width, MRCNN_DEVICE, d->stage_det.as<float>(), d->stage_mask.as<float>(), true);
            }
        }
        gather_records(d, m, d->stage_det.as<float>(), d->stage_mask.as<float>(), MRCNN_DEVICE, global_batch, memspace, detections, masks);
    });
}
