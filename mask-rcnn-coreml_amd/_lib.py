"""ctypes binding of libmaskrcnn_hip.so (C ABI: include/maskrcnn_hip.h).

The library is the product; there is no CPU fallback.  ``lib()`` raises ``NativeLibraryMissing`` when
the shared object has not been built (``python __graft_entry__.py`` / ``make -C csrc``), and every
compute entry point returns ``MRCNN_ERR_HIP`` when no gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MRCNN_HIP_LIB") or os.path.join(_HERE, "libmaskrcnn_hip.so")   # override: A/B builds

MRCNN_OK = 0
F32, F64, F16, U8, I32, F32S, F32X3 = 0, 1, 2, 3, 4, 5, 6
DEFAULT = -1          # MRCNN_DEFAULT of mrcnn_model_load: the mode the artefact is prepared for (stored split exponents -> F32X3, else F32)
HOST, DEVICE = 0, 1
PARAM_INT, PARAM_DOUBLE, PARAM_STRING = 0, 1, 2
MODEL_MASKRCNN, MODEL_CLASSIFIER, MODEL_MASK = 0, 1, 2

# declared in include/maskrcnn_hip.h (the drop-in surface)
EXPORTED_SYMBOLS = [
    "mrcnn_last_error", "mrcnn_version", "mrcnn_device_count", "mrcnn_config_set_anchors_path",
    "mrcnn_config_set_classifier_path", "mrcnn_config_set_mask_path", "mrcnn_config_get_anchors_path",
    "mrcnn_config_get_classifier_path", "mrcnn_config_get_mask_path", "mrcnn_layer_create",
    "mrcnn_layer_set_weight_data", "mrcnn_layer_output_shapes", "mrcnn_layer_evaluate", "mrcnn_layer_destroy",
    "mrcnn_iou", "mrcnn_model_load", "mrcnn_model_destroy", "mrcnn_model_set_stream", "mrcnn_maskrcnn_predict",
    "mrcnn_maskrcnn_predict_async", "mrcnn_maskrcnn_submit", "mrcnn_maskrcnn_collect", "mrcnn_classifier_predict", "mrcnn_mask_predict", "mrcnn_model_get_int",
    "mrcnn_model_read_tensor", "mrcnn_model_enable_timing", "mrcnn_model_stage_ms", "mrcnn_model_enable_graph",
    "mrcnn_detections_decode", "mrcnn_mask_to_u8", "mrcnn_paste_masks", "mrcnn_generate_anchors",
    "mrcnn_letterbox_geometry", "mrcnn_letterbox_rgb", "mrcnn_model_check_range", "mrcnn_model_calibrate_split", "mrcnn_model_split_group_stat",
    "mrcnn_model_get_split_exponents", "mrcnn_model_set_split_exponents", "mrcnn_roi_align_nhwc",
    "mrcnn_dist_unique_id", "mrcnn_dist_init", "mrcnn_dist_destroy", "mrcnn_dist_shard", "mrcnn_dist_record_floats",
    "mrcnn_dist_all_gather_records", "mrcnn_maskrcnn_predict_sharded", "mrcnn_mask_to_u8_f64",
    "mrcnn_dist_all_gather_records_async", "mrcnn_dist_wait", "mrcnn_dist_recovered", "mrcnn_dist_rccl_shared", "mrcnn_dist_plan", "mrcnn_dist_simulate_host",
    "mrcnn_maskrcnn_predict_scalefit", "mrcnn_unletterbox_boxes",
]
# declared in include/maskrcnn_hip_test.h (test / measurement entry points of the same library)
TEST_SYMBOLS = [
    "mrcnn_bench_conv", "mrcnn_bench_conv_dtype", "mrcnn_model_conv_profile_enable", "mrcnn_model_conv_profile_get",
    "mrcnn_model_conv_profile_shapes", "mrcnn_conv2d_nhwc", "mrcnn_debug_set", "mrcnn_bottleneck_nhwc", "mrcnn_bench_mfma_probe", "mrcnn_model_conv_profile_group", "mrcnn_bottleneck_first_nhwc", "mrcnn_bottleneck_stage_nhwc", "mrcnn_model_conv_profile_bytes",
]


class NativeLibraryMissing(RuntimeError):
    pass


class MrcnnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[mrcnn status {code}] {msg}")
        self.code = code


class Tensor(C.Structure):          # mrcnn_tensor
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("memspace", C.c_int32),
                ("shape", C.c_int64 * 5), ("strides", C.c_int64 * 5)]


class Param(C.Structure):           # mrcnn_param
    _fields_ = [("key", C.c_char_p), ("type", C.c_int32), ("i", C.c_int64), ("d", C.c_double), ("s", C.c_char_p)]


class SplitGroupStat(C.Structure):   # mrcnn_split_group_stat
    _fields_ = [("name", C.c_char * 48), ("exponent", C.c_int32), ("fixed", C.c_int32), ("absmax", C.c_float),
                ("small_inputs", C.c_int64), ("inexact_inputs", C.c_int64), ("inputs_counted", C.c_int64)]


class DetectionRecord(C.Structure):  # mrcnn_detection
    _fields_ = [("index", C.c_int64), ("x", C.c_double), ("y", C.c_double), ("w", C.c_double), ("h", C.c_double),
                ("class_id", C.c_int64), ("score", C.c_double)]


_lib = None


class ConvShapeStat(C.Structure):    # mrcnn_conv_shape_stat
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("tile", C.c_int32),
                ("launches", C.c_int64), ("total_ms", C.c_double), ("total_flops", C.c_double), ("total_bytes", C.c_double)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise NativeLibraryMissing(
            f"{SO_PATH} not built. Run `python __graft_entry__.py` (or `make -C mask-rcnn-coreml_amd/csrc`). "
            "There is no CPU fallback: the HIP library is the product.")
    # torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).
    # Two HIP runtimes in one process cannot both own the GPU, so when torch is importable it is
    # imported FIRST: the loader then resolves this library's DT_NEEDED libamdhip64.so.7 to the copy
    # torch already mapped, and torch tensors / streams and this library share one runtime.  A host
    # without torch (the Swift host) simply gets /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    vp, cp, i64p, f32p = C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_float)
    L.mrcnn_last_error.restype = cp
    L.mrcnn_version.restype = cp
    L.mrcnn_device_count.restype = C.c_int
    for n in ("anchors", "classifier", "mask"):
        getattr(L, f"mrcnn_config_set_{n}_path").argtypes = [cp]
        getattr(L, f"mrcnn_config_get_{n}_path").restype = cp
    L.mrcnn_layer_create.argtypes = [cp, C.POINTER(Param), C.c_int, C.POINTER(vp)]
    L.mrcnn_layer_set_weight_data.argtypes = [vp, vp, vp, C.c_int]
    L.mrcnn_layer_output_shapes.argtypes = [vp, vp, C.c_int, vp, C.POINTER(C.c_int)]
    L.mrcnn_layer_evaluate.argtypes = [vp, C.POINTER(Tensor), C.c_int, C.POINTER(Tensor), C.c_int]
    L.mrcnn_layer_destroy.argtypes = [vp]
    L.mrcnn_layer_destroy.restype = None
    L.mrcnn_iou.argtypes = [f32p, f32p]
    L.mrcnn_iou.restype = C.c_float
    L.mrcnn_model_load.argtypes = [C.c_int, cp, C.c_int, C.c_int, C.POINTER(vp)]
    L.mrcnn_model_destroy.argtypes = [vp]
    L.mrcnn_model_destroy.restype = None
    L.mrcnn_model_set_stream.argtypes = [vp, vp]
    L.mrcnn_maskrcnn_predict.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.mrcnn_maskrcnn_predict_scalefit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.mrcnn_unletterbox_boxes.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mrcnn_maskrcnn_predict_async.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.mrcnn_maskrcnn_submit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.mrcnn_maskrcnn_collect.argtypes = [vp, vp, vp, C.POINTER(C.c_int)]
    L.mrcnn_classifier_predict.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    L.mrcnn_mask_predict.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.mrcnn_model_get_int.argtypes = [vp, cp, i64p]
    L.mrcnn_model_read_tensor.argtypes = [vp, cp, C.c_int, vp, C.c_int64, i64p]
    L.mrcnn_model_enable_timing.argtypes = [vp, C.c_int]
    L.mrcnn_model_stage_ms.argtypes = [vp, cp, f32p]
    L.mrcnn_model_conv_profile_enable.argtypes = [vp, C.c_int]
    L.mrcnn_model_conv_profile_get.argtypes = [vp, C.c_int, i64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mrcnn_model_conv_profile_group.argtypes = [vp, C.c_int, i64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mrcnn_model_enable_graph.argtypes = [vp, C.c_int]
    L.mrcnn_model_conv_profile_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.mrcnn_model_conv_profile_shapes.argtypes = [vp, C.POINTER(ConvShapeStat), C.c_int, C.POINTER(C.c_int)]
    L.mrcnn_bench_conv.argtypes = [C.c_int] * 8 + [f32p, C.POINTER(C.c_double)]
    L.mrcnn_bench_conv_dtype.argtypes = [C.c_int] * 9 + [f32p, C.POINTER(C.c_double)]
    L.mrcnn_detections_decode.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(DetectionRecord), C.c_int64, i64p]
    L.mrcnn_mask_to_u8.argtypes = [vp, C.c_int64, vp]
    L.mrcnn_mask_to_u8_f64.argtypes = [vp, C.c_int64, vp]
    ip = C.POINTER(C.c_int)
    L.mrcnn_letterbox_geometry.argtypes = [C.c_int] * 4 + [ip] * 4
    L.mrcnn_letterbox_rgb.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.mrcnn_generate_anchors.argtypes = [C.c_int, C.c_int, vp, C.c_int64, i64p]
    L.mrcnn_paste_masks.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp]
    L.mrcnn_conv2d_nhwc.argtypes = [vp] + [C.c_int] * 4 + [vp] + [C.c_int] * 3 + [vp, vp, vp, C.c_int, C.c_int, vp]
    L.mrcnn_debug_set.argtypes = [cp, C.c_int]
    L.mrcnn_bench_mfma_probe.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mrcnn_bottleneck_first_nhwc.argtypes = [vp] + [C.c_int] * 4 + [vp] * 5 + [C.c_int, C.c_int, vp, vp]
    L.mrcnn_bottleneck_stage_nhwc.argtypes = [vp] + [C.c_int] * 4 + [vp] * 4 + [C.c_int, C.c_int, vp, vp, vp]
    L.mrcnn_bottleneck_nhwc.argtypes = [vp] + [C.c_int] * 4 + [vp] * 9 + [C.c_int, C.c_int, vp, vp]
    L.mrcnn_model_check_range.argtypes = [vp, ip]
    L.mrcnn_model_calibrate_split.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mrcnn_model_split_group_stat.argtypes = [vp, C.c_int, C.POINTER(SplitGroupStat)]
    L.mrcnn_model_get_split_exponents.argtypes = [vp, vp, C.c_int, ip]
    L.mrcnn_model_set_split_exponents.argtypes = [vp, vp, C.c_int]
    L.mrcnn_dist_rccl_shared.argtypes = []
    L.mrcnn_dist_rccl_shared.restype = C.c_int
    L.mrcnn_dist_unique_id.argtypes = [vp]
    L.mrcnn_dist_init.argtypes = [C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.mrcnn_dist_destroy.argtypes = [vp]
    L.mrcnn_dist_destroy.restype = None
    L.mrcnn_dist_shard.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
    L.mrcnn_dist_record_floats.argtypes = [C.c_int, C.c_int]
    L.mrcnn_dist_record_floats.restype = C.c_int64
    L.mrcnn_dist_all_gather_records.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.mrcnn_maskrcnn_predict_sharded.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.mrcnn_dist_all_gather_records_async.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp]
    L.mrcnn_dist_wait.argtypes = [vp]
    L.mrcnn_dist_recovered.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.mrcnn_dist_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i64p, i64p]
    L.mrcnn_dist_simulate_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp]
    L.mrcnn_roi_align_nhwc.argtypes = [C.POINTER(vp), ip, ip, C.c_int, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_int, vp, vp]
    _lib = L
    return L


def check(status: int):
    if status != MRCNN_OK:
        raise MrcnnError(status, lib().mrcnn_last_error().decode(errors="replace"))


def make_params(d: dict):
    """dict → (Param array, n).  int → intValue, float → doubleValue, like task.py:25-67."""
    arr = (Param * max(1, len(d)))()
    keep = []
    for i, (k, v) in enumerate(d.items()):
        kb = str(k).encode()
        keep.append(kb)
        arr[i].key = kb
        if isinstance(v, bool) or isinstance(v, int):
            arr[i].type, arr[i].i = PARAM_INT, int(v)
        elif isinstance(v, float):
            arr[i].type, arr[i].d = PARAM_DOUBLE, float(v)
        else:
            sb = str(v).encode()
            keep.append(sb)
            arr[i].type, arr[i].s = PARAM_STRING, sb
    arr._keepalive = keep
    return arr, len(d)
