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

/// Replaces the Xcode-generated `MaskRCNN` class (Example/Source/ViewController.swift:37).
public final class MaskRCNN {
    private var handle: OpaquePointer?
    public let maxDetections: Int
    public let maskSide: Int
    public let width: Int32
    public let height: Int32

    public init(contentsOf url: URL, maxBatch: Int32 = 1, halfPrecision: Bool = false) throws {
        try check(mrcnn_model_load(Int32(MRCNN_MODEL_MASKRCNN.rawValue), url.path, maxBatch,
                                   Int32(halfPrecision ? MRCNN_F16.rawValue : MRCNN_F32.rawValue), &handle))
        var v: Int64 = 0
        try check(mrcnn_model_get_int(handle, "max_detections", &v)); maxDetections = Int(v)
        try check(mrcnn_model_get_int(handle, "mask_size", &v)); maskSide = Int(v)
        try check(mrcnn_model_get_int(handle, "image_width", &v)); width = Int32(v)
        try check(mrcnn_model_get_int(handle, "image_height", &v)); height = Int32(v)
    }
    deinit { mrcnn_model_destroy(handle) }

    /// `image`: RGB8, width x height of the model (letterbox first: `mrcnn_letterbox_rgb` = `.scaleFit`).
    /// Returns the two outputs of the reference graph: "detections" (maxDet x 6) and "mask" (maxDet x 28 x 28).
    public func prediction(image rgb: UnsafePointer<UInt8>) throws -> (detections: [Float], mask: [Float]) {
        var det = [Float](repeating: 0, count: maxDetections * 6)
        var msk = [Float](repeating: 0, count: maxDetections * maskSide * maskSide)
        try check(mrcnn_maskrcnn_predict(handle, rgb, 1, height, width, Int32(MRCNN_HOST.rawValue), &det, &msk))
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
