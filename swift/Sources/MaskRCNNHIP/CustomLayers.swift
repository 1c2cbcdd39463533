// The five Core ML custom layers of the reference as forwarding classes over the C ABI of libmaskrcnn_hip.so.
//
// Core ML resolves a custom layer BY ITS @objc CLASS NAME from the model spec (className fields written by
// Sources/maskrcnn/Python/Conversion/task.py:25-67): "ProposalLayer", "PyramidROIAlignLayer",
// "TimeDistributedClassifierLayer", "DetectionLayer", "TimeDistributedMaskLayer".  Linking this file instead of the
// reference's Swift implementations (Sources/Mask-RCNN-CoreML/{ProposalLayer,PyramidROIAlignLayer,
// TimeDistributedClassifierLayer,DetectionLayer,TimeDistributedMaskLayer}.swift) keeps the .mlmodel files, the
// parameter dictionaries and MLMultiArray conventions untouched: each class implements the four MLCustomLayer methods
// (ProposalLayer.swift:65,93,97,103) by handing its arguments to mrcnn_layer_create / _set_weight_data /
// _output_shapes / _evaluate, which run the stage on the MI355X.
//
// NOT compiled in this repository (no Swift toolchain / no CoreML framework in the Linux image) — the same C entry
// points are driven by mask-rcnn-coreml_amd/layers.py in every `-m gpu` test, with the same parameter dictionaries, 5-D
// shapes, element strides and dirty output buffers.
#if canImport(CoreML)
import CMaskRCNNHIP
import CoreML
import Foundation

/// [String: Any] → mrcnn_param[]: `as? Int` → intValue, `as? Double` → doubleValue (the typing the reference relies on,
/// ProposalLayer.swift:70-90), anything else as a string.
private func withParams<R>(_ parameters: [String: Any], _ body: (UnsafePointer<mrcnn_param>?, Int32) throws -> R) rethrows -> R {
    var keep: [UnsafeMutablePointer<CChar>] = []
    defer { keep.forEach { free($0) } }
    var params: [mrcnn_param] = []
    for (k, v) in parameters {
        let key = strdup(k)!
        keep.append(key)
        var p = mrcnn_param()
        p.key = UnsafePointer(key)
        if let i = v as? Int {
            p.type = Int32(MRCNN_PARAM_INT.rawValue); p.i = Int64(i)
        } else if let d = v as? Double {
            p.type = Int32(MRCNN_PARAM_DOUBLE.rawValue); p.d = d
        } else {
            let s = strdup(String(describing: v))!
            keep.append(s)
            p.type = Int32(MRCNN_PARAM_STRING.rawValue); p.s = UnsafePointer(s)
        }
        params.append(p)
    }
    return try params.withUnsafeBufferPointer { try body($0.baseAddress, Int32($0.count)) }
}

/// MLMultiArray → mrcnn_tensor: raw dataPointer, 5-D shape, strides in ELEMENTS (Utils.swift:93-99 `floatDataPointer`,
/// the layers index with shape[0], strides[0] / strides[2]).
private func tensor(_ a: MLMultiArray) throws -> mrcnn_tensor {
    guard a.dataType == .float32 else { throw "custom layers take Float32 arrays (ProposalLayer.swift:108)" }
    var t = mrcnn_tensor()
    t.data = a.dataPointer
    t.dtype = Int32(MRCNN_F32.rawValue)
    t.memspace = Int32(MRCNN_HOST.rawValue)
    let rank = a.shape.count
    withUnsafeMutablePointer(to: &t.shape) { $0.withMemoryRebound(to: Int64.self, capacity: 5) { sh in
        withUnsafeMutablePointer(to: &t.strides) { $0.withMemoryRebound(to: Int64.self, capacity: 5) { st in
            for i in 0..<5 {                                    // right-align lower ranks like Core ML does
                let j = i - (5 - rank)
                sh[i] = j >= 0 ? a.shape[j].int64Value : 1
                st[i] = j >= 0 ? a.strides[j].int64Value : (rank > 0 ? a.strides[0].int64Value * a.shape[0].int64Value : 1)
            }
        } }
    } }
    return t
}

/// Shared implementation: one mrcnn_layer handle per Core ML layer instance.
public class HIPCustomLayer: NSObject {
    fileprivate var handle: OpaquePointer?

    fileprivate init(className: String, parameters: [String: Any]) throws {
        super.init()
        let st = withParams(parameters) { p, n in mrcnn_layer_create(className, p, n, &handle) }
        if st != 0 { throw String(cString: mrcnn_last_error()) }
    }
    deinit { mrcnn_layer_destroy(handle) }

    public func setWeightData(_ weights: [Data]) throws {
        // none of the five layers carries weights (ProposalLayer.swift:93-95 is empty): forwarded for completeness
        var ptrs: [UnsafeRawPointer?] = []
        var sizes: [Int] = []
        let pinned = weights.map { NSData(data: $0) }
        for d in pinned { ptrs.append(d.bytes); sizes.append(d.length) }
        let st = ptrs.withUnsafeBufferPointer { pp in sizes.withUnsafeBufferPointer { ss in
            mrcnn_layer_set_weight_data(handle, pp.baseAddress, ss.baseAddress, Int32(weights.count)) } }
        if st != 0 { throw String(cString: mrcnn_last_error()) }
    }

    public func outputShapes(forInputShapes inputShapes: [[NSNumber]]) throws -> [[NSNumber]] {
        var flat = [Int64](repeating: 1, count: inputShapes.count * 5)
        for (i, s) in inputShapes.enumerated() {
            for (j, v) in s.enumerated() where j < 5 { flat[i * 5 + (5 - s.count) + j] = v.int64Value }
        }
        var out = [Int64](repeating: 0, count: 4 * 5)
        var nOut: Int32 = 0
        let st = flat.withUnsafeBufferPointer { f in out.withUnsafeMutableBufferPointer { o in
            f.baseAddress!.withMemoryRebound(to: (Int64, Int64, Int64, Int64, Int64).self, capacity: inputShapes.count) { fi in
                o.baseAddress!.withMemoryRebound(to: (Int64, Int64, Int64, Int64, Int64).self, capacity: 4) { oi in
                    mrcnn_layer_output_shapes(handle, fi, Int32(inputShapes.count), oi, &nOut) } } } }
        if st != 0 { throw String(cString: mrcnn_last_error()) }
        return (0..<Int(nOut)).map { k in (0..<5).map { NSNumber(value: out[k * 5 + $0]) } }
    }

    public func evaluate(inputs: [MLMultiArray], outputs: [MLMultiArray]) throws {
        let ins = try inputs.map(tensor)
        var outs = try outputs.map(tensor)
        // outputs are caller-owned and NOT cleared by Core ML: the library overwrites every element incl. the zero padding
        // (ProposalLayer.swift:188-192, DetectionLayer.swift:226-231, TimeDistributedMaskLayer.swift:87-89)
        let st = ins.withUnsafeBufferPointer { i in outs.withUnsafeMutableBufferPointer { o in
            mrcnn_layer_evaluate(handle, i.baseAddress, Int32(i.count), o.baseAddress, Int32(o.count)) } }
        if st != 0 { throw String(cString: mrcnn_last_error()) }
    }
}

@objc(ProposalLayer) public final class ProposalLayer: HIPCustomLayer, MLCustomLayer {
    public required init(parameters: [String: Any]) throws { try super.init(className: "ProposalLayer", parameters: parameters) }
}

@objc(PyramidROIAlignLayer) public final class PyramidROIAlignLayer: HIPCustomLayer, MLCustomLayer {
    public required init(parameters: [String: Any]) throws { try super.init(className: "PyramidROIAlignLayer", parameters: parameters) }
}

@objc(TimeDistributedClassifierLayer) public final class TimeDistributedClassifierLayer: HIPCustomLayer, MLCustomLayer {
    public required init(parameters: [String: Any]) throws { try super.init(className: "TimeDistributedClassifierLayer", parameters: parameters) }
}

@objc(DetectionLayer) public final class DetectionLayer: HIPCustomLayer, MLCustomLayer {
    public required init(parameters: [String: Any]) throws { try super.init(className: "DetectionLayer", parameters: parameters) }
}

@objc(TimeDistributedMaskLayer) public final class TimeDistributedMaskLayer: HIPCustomLayer, MLCustomLayer {
    public required init(parameters: [String: Any]) throws { try super.init(className: "TimeDistributedMaskLayer", parameters: parameters) }
}
#endif
