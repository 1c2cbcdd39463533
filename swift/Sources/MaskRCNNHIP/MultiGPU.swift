// Multi-GPU from the Swift host: one process per GPU, a batch of images sharded over the ranks, one RCCL all-gather of
// the per-image records — the C ABI of mask-rcnn-coreml_amd/csrc/dist.hip (mrcnn_dist_*).  The reference processes one
// image on one device (Sources/maskrcnn/EvaluateCommand.swift:146-179); this is the same loop over 8 MI355X.
// Not compiled in this repository (no Swift toolchain); the C entry points are exercised by examples/maskrcnn_predict_mgpu.c
// and tests/test_gpu_layers.py::test_native_dist_world_one_through_rccl.
import CMaskRCNNHIP
import Foundation

public final class MaskRCNNDist {
    private var handle: OpaquePointer?
    public let rank: Int32
    public let world: Int32

    /// Rank 0 creates the 128-byte rendezvous id and ships it to the other ranks (file, environment, pipe — the host's choice).
    public static func uniqueId() throws -> [UInt8] {
        var id = [UInt8](repeating: 0, count: 128)
        if mrcnn_dist_unique_id(&id) != 0 { throw String(cString: mrcnn_last_error()) }
        return id
    }

    /// Joins the communicator on the process's current HIP device (HIP_VISIBLE_DEVICES selects it).
    public init(rank: Int32, world: Int32, uniqueId: [UInt8]) throws {
        precondition(uniqueId.count == 128)
        self.rank = rank
        self.world = world
        if mrcnn_dist_init(rank, world, uniqueId, &handle) != 0 { throw String(cString: mrcnn_last_error()) }
    }
    deinit { mrcnn_dist_destroy(handle) }

    /// [begin, end) of this rank in a batch of `globalBatch` images.
    public func shard(globalBatch: Int32) -> Range<Int32> {
        var lo: Int32 = 0, hi: Int32 = 0
        _ = mrcnn_dist_shard(globalBatch, world, rank, &lo, &hi)
        return lo..<hi
    }

    /// Every rank passes the SAME global batch (RGB8, the model's input size) and receives the whole batch's outputs.
    public func predictSharded(model: OpaquePointer, images: UnsafePointer<UInt8>, globalBatch: Int32, width: Int32, height: Int32,
                               maxDetections: Int, maskSide: Int) throws -> (detections: [Float], mask: [Float]) {
        var det = [Float](repeating: 0, count: Int(globalBatch) * maxDetections * 6)
        var msk = [Float](repeating: 0, count: Int(globalBatch) * maxDetections * maskSide * maskSide)
        if mrcnn_maskrcnn_predict_sharded(handle, model, images, globalBatch, height, width, Int32(MRCNN_HOST.rawValue), &det, &msk) != 0 {
            throw String(cString: mrcnn_last_error())
        }
        return (det, msk)
    }
}
