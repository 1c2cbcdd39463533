// Swift host shim over the C ABI of libmaskrcnn_hip.so.  Not compiled in this repository's CI (no Swift
// toolchain in the image); see swift/README.md and INTEGRATION.md.
import CMaskRCNNHIP
import Foundation

extension String: Error {}   // same convention as the reference (Sources/Mask-RCNN-CoreML/Utils.swift:13)

private func check(_ status: Int32) throws {
    if status != 0 { throw String(cString: mrcnn_last_error()) }
}

/// MaskRCNNConfig.defaultConfig (Sources/Mask-RCNN-CoreML/MaskRCNNConfig.swift:10-18)
public final class MaskRCNNConfig {
    public static let defaultConfig = MaskRCNNConfig()
    public var anchorsURL: URL? { didSet { _ = mrcnn_config_set_anchors_path(anchorsURL?.path) } }
    public var compiledClassifierModelURL: URL? { didSet { _ = mrcnn_config_set_classifier_path(compiledClassifierModelURL?.path) } }
    public var compiledMaskModelURL: URL? { didSet { _ = mrcnn_config_set_mask_path(compiledMaskModelURL?.path) } }
}

/// Compute mode of the convolutions (`compute_dtype` of mrcnn_model_load; include/maskrcnn_hip.h).
public enum ComputeMode {
    /// MRCNN_DEFAULT: the mode the artefact is prepared for — `.f32x3` with the stored exponents when `convert --calibrate` wrote them
    /// into MaskRCNN.mrcw, `.f32` otherwise.  What `MaskRCNN()` (ViewController.swift:37), which names no precision, gets.
    case artefact
    /// fp32 tensors, exact-fp32 MFMA (`v_mfma_f32_32x32x2_f32`): the scale-invariant baseline.
    case f32
    /// fp32 tensors; products on the fp16 matrix cores from a THREE-part split of the activation (exact for 0.5 <= |a| < 65504,
    /// the activation carried to 2^-25 absolute below): the mode `bench.py` reports.
    case f32x3
    /// fp32 tensors; two-part split (22 of 24 significand bits).
    case f32s
    /// fp16 tensors and fp16 MFMA, fp32 accumulate, fp32 box path and outputs (BASELINE configs[3]).
    case f16
    var raw: Int32 {
        switch self {
        case .artefact: return Int32(MRCNN_DEFAULT.rawValue)
        case .f32: return Int32(MRCNN_F32.rawValue)
        case .f32x3: return Int32(MRCNN_F32X3.rawValue)
        case .f32s: return Int32(MRCNN_F32S.rawValue)
        case .f16: return Int32(MRCNN_F16.rawValue)
        }
    }
}

/// Replaces the Xcode-generated `MaskRCNN` class (Example/Source/ViewController.swift:37).
public final class MaskRCNN {
    private var handle: OpaquePointer?
    public let maxDetections: Int
    public let maskSide: Int
    public let width: Int32
    public let height: Int32
    /// The mode the handle runs in (`.artefact` resolved by the library: mrcnn_model_get_int "compute_dtype").
    public let computeMode: ComputeMode

    /// `computeMode` defaults to `.artefact` (MRCNN_DEFAULT): an artefact written by `convert --calibrate` carries the split exponents of
    /// its tensor groups and loads as `.f32x3` — fp32 tensors whose products are formed exactly on the fp16 matrix cores from a three-part
    /// split (2.6x the exact-fp32 mode; the mode `bench.py` measures), fp32-grade at any activation scale with those exponents, and a batch
    /// that leaves the calibrated range is recovered inside the call (INTEGRATION.md §2.4c); an artefact without them loads as `.f32`.
    /// `computeMode` (read-only) says which one it became.
    public init(contentsOf url: URL, maxBatch: Int32 = 1, computeMode: ComputeMode = .artefact) throws {
        try check(mrcnn_model_load(Int32(MRCNN_MODEL_MASKRCNN.rawValue), url.path, maxBatch, computeMode.raw, &handle))
        var v: Int64 = 0
        try check(mrcnn_model_get_int(handle, "max_detections", &v)); maxDetections = Int(v)
        try check(mrcnn_model_get_int(handle, "mask_size", &v)); maskSide = Int(v)
        try check(mrcnn_model_get_int(handle, "image_width", &v)); width = Int32(v)
        try check(mrcnn_model_get_int(handle, "image_height", &v)); height = Int32(v)
        try check(mrcnn_model_get_int(handle, "compute_dtype", &v))
        self.computeMode = v == Int64(MRCNN_F32X3.rawValue) ? .f32x3 : v == Int64(MRCNN_F32S.rawValue) ? .f32s : v == Int64(MRCNN_F16.rawValue) ? .f16 : .f32
    }
    /// Source compatibility with the first version of this shim (`halfPrecision: Bool`): `true` = `.f16`, `false` = `.f32`
    /// (the exact-fp32 engine that initialiser used to select — NOT the new `.f32x3` default).
    @available(*, deprecated, message: "use init(contentsOf:maxBatch:computeMode:); false maps to .f32, true to .f16")
    public convenience init(contentsOf url: URL, maxBatch: Int32 = 1, halfPrecision: Bool) throws {
        try self.init(contentsOf: url, maxBatch: maxBatch, computeMode: halfPrecision ? .f16 : .f32)
    }
    deinit { mrcnn_model_destroy(handle) }

    /// `image`: RGB8, width x height of the model.
    /// Returns the two outputs of the reference graph: "detections" (maxDet x 6) and "mask" (maxDet x 28 x 28).
    public func prediction(image rgb: UnsafePointer<UInt8>) throws -> (detections: [Float], mask: [Float]) {
        var det = [Float](repeating: 0, count: maxDetections * 6)
        var msk = [Float](repeating: 0, count: maxDetections * maskSide * maskSide)
        try check(mrcnn_maskrcnn_predict(handle, rgb, 1, height, width, Int32(MRCNN_HOST.rawValue), &det, &msk))
        return (det, msk)
    }

    /// Scale-aware split (`.f32x3` / `.f32s` only): one calibration predict on a representative image picks a power-of-two exponent per
    /// tensor group (folded into the layers at no run-time cost) so that the mode carries activations like fp32 whatever the
    /// checkpoint's scale; returns the number of inputs the split still cannot carry exactly.  `apply: false` only diagnoses.
    /// OPTIONAL since round 5: an artefact converted with `convert --calibrate` carries the exponents (`exponentsFromArtefact`), and a
    /// prediction that leaves the calibrated range lowers them and runs again inside the call (`rangeRecoveries`) instead of throwing.
    @discardableResult
    public func calibrateSplit(image rgb: UnsafePointer<UInt8>, apply: Bool = true) throws -> Int64 {
        try check(mrcnn_model_calibrate_split(handle, rgb, 1, height, width, Int32(MRCNN_HOST.rawValue), apply ? 1 : 0))
        var n: Int64 = 0
        try check(mrcnn_model_get_int(handle, "split_inexact_inputs", &n))
        return n
    }

    /// Whether `mrcnn_model_load` found the split exponents in MaskRCNN.mrcw, and how many predictions recovered from a range trip so far.
    public var exponentsFromArtefact: Bool {
        var n: Int64 = 0
        return mrcnn_model_get_int(handle, "split_exponents_from_artefact", &n) == 0 && n != 0
    }
    public var rangeRecoveries: Int64 {
        var n: Int64 = 0
        _ = mrcnn_model_get_int(handle, "range_recoveries", &n)
        return n
    }

    /// The evaluate loop with the hand-over of the NEXT image overlapped (EvaluateCommand.swift:167-179): `submit` copies an image on the
    /// handle's copy stream and enqueues its predict, `collect` returns the OLDEST submission's outputs.  At most two in flight; the
    /// buffer passed to `submit` must stay alive until the matching `collect`.
    public func submit(image rgb: UnsafePointer<UInt8>) throws {
        try check(mrcnn_maskrcnn_submit(handle, rgb, 1, height, width))
    }
    public func collect() throws -> (detections: [Float], mask: [Float]) {
        var det = [Float](repeating: 0, count: maxDetections * 6)
        var msk = [Float](repeating: 0, count: maxDetections * maskSide * maskSide)
        var n: Int32 = 0
        try check(mrcnn_maskrcnn_collect(handle, &det, &msk, &n))
        return (det, msk)
    }

    /// `image`: RGB8 of ANY size — what `VNCoreMLRequest` with `.scaleFit` does for the reference (EvaluateCommand.swift:152-157,
    /// ViewController.swift:45): the letterbox runs inside the engine's pre-processing kernel.  Boxes are normalized in the
    /// letterboxed frame, like the reference's; `unletterboxed` maps them back to the source image.
    public func prediction(image rgb: UnsafePointer<UInt8>, width w: Int32, height h: Int32, unletterboxed: Bool = false) throws
        -> (detections: [Float], mask: [Float]) {
        var det = [Float](repeating: 0, count: maxDetections * 6)
        var msk = [Float](repeating: 0, count: maxDetections * maskSide * maskSide)
        try check(mrcnn_maskrcnn_predict_scalefit(handle, rgb, 1, h, w, Int32(MRCNN_HOST.rawValue), &det, &msk))
        if unletterboxed { try check(mrcnn_unletterbox_boxes(&det, Int64(maxDetections), 6, h, w, height, width)) }
        return (det, msk)
    }
}

/// Classifier.prediction(feature_map:) (Conversion/task.py:106-113)
public final class Classifier {
    private var handle: OpaquePointer?
    private let numClasses: Int
    public init(contentsOf url: URL, maxRows: Int32 = 1000) throws {
        try check(mrcnn_model_load(Int32(MRCNN_MODEL_CLASSIFIER.rawValue), url.path, maxRows, Int32(MRCNN_F32.rawValue), &handle))
        var v: Int64 = 0
        try check(mrcnn_model_get_int(handle, "num_classes", &v)); numClasses = Int(v)
    }
    deinit { mrcnn_model_destroy(handle) }
    public func prediction(featureMap: UnsafePointer<Float>, count n: Int32) throws -> (probabilities: [Float], boundingBoxes: [Float]) {
        var p = [Float](repeating: 0, count: Int(n) * numClasses), b = [Float](repeating: 0, count: Int(n) * numClasses * 4)
        try check(mrcnn_classifier_predict(handle, featureMap, n, Int32(MRCNN_HOST.rawValue), &p, &b))
        return (p, b)
    }
}

/// Mask.prediction(feature_map:) (Conversion/task.py:94-101)
public final class Mask {
    private var handle: OpaquePointer?
    private let numClasses: Int
    public init(contentsOf url: URL, maxRows: Int32 = 100) throws {
        try check(mrcnn_model_load(Int32(MRCNN_MODEL_MASK.rawValue), url.path, maxRows, Int32(MRCNN_F32.rawValue), &handle))
        var v: Int64 = 0
        try check(mrcnn_model_get_int(handle, "num_classes", &v)); numClasses = Int(v)
    }
    deinit { mrcnn_model_destroy(handle) }
    public func prediction(featureMap: UnsafePointer<Float>, count n: Int32) throws -> [Float] {
        var m = [Float](repeating: 0, count: Int(n) * numClasses * 28 * 28)
        try check(mrcnn_mask_predict(handle, featureMap, n, Int32(MRCNN_HOST.rawValue), &m))
        return m
    }
}
