/*
 * maskrcnn_hip.h — C ABI of libmaskrcnn_hip.so, the MI355X (gfx950) replacement for the hot path of
 * edouardlp/Mask-RCNN-CoreML.  Plain pointers and sizes only; no C++/torch types.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the reference
 * repo).  All functions return an mrcnn_status (0 = OK) and never abort; the message of the last
 * failure on the calling thread is available from mrcnn_last_error() (the reference throws Swift
 * errors — `extension String: Error`, Sources/Mask-RCNN-CoreML/Utils.swift:13 — or crashes on a
 * force-unwrap, ProposalLayer.swift:68).
 *
 * Threading: handles are not thread-safe; one in-flight call per handle; calls are synchronous
 * (they return after the work on the handle's HIP stream has completed), like Core ML's
 * `evaluate` (ProposalLayer.swift:103).  One model handle per GPU for multi-GPU use.
 *
 * Memory: the caller owns every buffer passed in; the callee never frees them and fully
 * overwrites outputs including zero padding ("CoreML does not erase the memory between
 * evaluations", ProposalLayer.swift:188) — except where the reference itself leaves rows
 * unwritten (TimeDistributedMaskLayer, see below).
 */
#ifndef MASKRCNN_HIP_H
#define MASKRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRCNN_API __attribute__((visibility("default")))

typedef enum {
    MRCNN_OK = 0,
    MRCNN_ERR_INVALID = 1,      /* bad argument / unknown key / wrong dtype                        */
    MRCNN_ERR_IO = 2,           /* file missing or malformed (anchors.bin, *.mrcw)                 */
    MRCNN_ERR_HIP = 3,          /* HIP runtime failure, or no gfx950 device                        */
    MRCNN_ERR_SHAPE = 4,        /* shape / stride mismatch                                          */
    MRCNN_ERR_UNSUPPORTED = 5,
    MRCNN_ERR_CONFIG = 6        /* MaskRCNNConfig URL not set before use                            */
} mrcnn_status;

MRCNN_API const char* mrcnn_last_error(void);
MRCNN_API const char* mrcnn_version(void);
/* Number of visible HIP devices (0 when none; never fails). */
MRCNN_API int mrcnn_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * MLMultiArray mirror (CoreML): 5-D [sequence, batch, channel, height, width], row-major unless
 * strides say otherwise; strides are in ELEMENTS.  The custom layers index with shape[0],
 * strides[0] or strides[2] and the raw data pointer (e.g. ProposalLayer.swift:179,
 * TimeDistributedClassifierLayer.swift:63, PyramidROIAlignLayer.swift:124).
 * --------------------------------------------------------------------------------------------- */
typedef enum {
    MRCNN_F32 = 0, MRCNN_F64 = 1, MRCNN_F16 = 2, MRCNN_U8 = 3, MRCNN_I32 = 4,
    MRCNN_F32S = 5,  /* compute mode only (mrcnn_model_load): fp32 tensors, fp16 filters, split-fp16 MFMA — see there */
    MRCNN_F32X3 = 6, /* compute mode only: as MRCNN_F32S with a three-part split (all 24 significand bits for |a| >= 0.5) */
    MRCNN_DEFAULT = -1 /* compute mode only (mrcnn_model_load): what the ARTEFACT is prepared for — MRCNN_F32X3 when MaskRCNN.mrcw carries
                        * stored split exponents (convert.py --calibrate), MRCNN_F32 otherwise; see mrcnn_model_load */
} mrcnn_dtype;
typedef enum { MRCNN_HOST = 0, MRCNN_DEVICE = 1 } mrcnn_memspace;

typedef struct {
    void*   data;
    int32_t dtype;       /* mrcnn_dtype; the layers assert Float32 inputs (ProposalLayer.swift:108) */
    int32_t memspace;    /* mrcnn_memspace: host buffers are staged over PCIe, device buffers used in place */
    int64_t shape[5];
    int64_t strides[5];  /* elements */
} mrcnn_tensor;

/* Custom-layer parameter dictionary entry: `parameters: [String : Any]`
 * (ProposalLayer.swift:65).  Keys and types are fixed by the converter,
 * Sources/maskrcnn/Python/Conversion/task.py:25-67 (intValue / doubleValue). */
typedef enum { MRCNN_PARAM_INT = 0, MRCNN_PARAM_DOUBLE = 1, MRCNN_PARAM_STRING = 2 } mrcnn_param_type;
typedef struct {
    const char* key;
    int32_t     type;    /* mrcnn_param_type */
    int64_t     i;
    double      d;
    const char* s;
} mrcnn_param;

/* ---------------------------------------------------------------------------------------------
 * MaskRCNNConfig.defaultConfig (Sources/Mask-RCNN-CoreML/MaskRCNNConfig.swift:10-18):
 * process-global URLs read by the layers at init/evaluate (ProposalLayer.swift:68,
 * TimeDistributedClassifierLayer.swift:41, TimeDistributedMaskLayer.swift:49).  Must be set before
 * the main model / ProposalLayer is created (Example/Source/AppDelegate.swift:18-20).
 * NULL clears.  Getters return NULL when unset; the pointer stays valid until the next set.
 * --------------------------------------------------------------------------------------------- */
MRCNN_API int mrcnn_config_set_anchors_path(const char* path);      /* anchorsURL                  */
MRCNN_API int mrcnn_config_set_classifier_path(const char* path);   /* compiledClassifierModelURL  */
MRCNN_API int mrcnn_config_set_mask_path(const char* path);         /* compiledMaskModelURL        */
MRCNN_API const char* mrcnn_config_get_anchors_path(void);
MRCNN_API const char* mrcnn_config_get_classifier_path(void);
MRCNN_API const char* mrcnn_config_get_mask_path(void);

/* ---------------------------------------------------------------------------------------------
 * The five MLCustomLayer plugins.  Core ML resolves them BY @objc CLASS NAME from the model spec
 * (task.py:27,39,48,53,59) and drives four methods; mrcnn_layer_* mirror them one to one:
 *
 *   init(parameters:)                 → mrcnn_layer_create      (ProposalLayer.swift:65, PyramidROIAlignLayer.swift:48,
 *                                                                DetectionLayer.swift:63, TimeDistributed*Layer.swift:18)
 *   setWeightData(_:)                 → mrcnn_layer_set_weight_data   (no-op in all five: ProposalLayer.swift:93)
 *   outputShapes(forInputShapes:)     → mrcnn_layer_output_shapes     (ProposalLayer.swift:97, PyramidROIAlignLayer.swift:65,
 *                                                                DetectionLayer.swift:94, TimeDistributedClassifierLayer.swift:26,
 *                                                                TimeDistributedMaskLayer.swift:26)
 *   evaluate(inputs:outputs:)         → mrcnn_layer_evaluate          (ProposalLayer.swift:103, PyramidROIAlignLayer.swift:79,
 *                                                                TimeDistributedClassifierLayer.swift:34, DetectionLayer.swift:107,
 *                                                                TimeDistributedMaskLayer.swift:39)
 *
 * class_name ∈ { "ProposalLayer", "PyramidROIAlignLayer", "TimeDistributedClassifierLayer",
 *                "DetectionLayer", "TimeDistributedMaskLayer" }.
 *
 * evaluate inputs/outputs per layer (all Float32):
 *   ProposalLayer                  in : probs (A,·,2…) contiguous (A,2); deltas (A,4)          out: rois, maxProposals rows of 4, row stride strides[0]
 *   PyramidROIAlignLayer           in : rois (n rows, stride strides[0]); 4 maps [·,·,C,H,W]   out: [n,1,C,pool,pool], row stride strides[0]
 *   TimeDistributedClassifierLayer in : pooled [n,1,C,7,7]                                      out: [n,1,1,1,6] rows (dy,dx,dh,dw,classId,score), row stride strides[2]
 *   DetectionLayer                 in : rois (n,4) row stride strides[0]; cls (n,6) contiguous  out: maxDetections rows of 6, row stride strides[0]
 *   TimeDistributedMaskLayer       in : pooled [D,1,C,14,14]; detections (D rows, strides[0])   out: [1,1,D,28,28], row stride strides[2]; rows the
 *                                        reference never writes (TimeDistributedMaskLayer.swift:58-89) are left untouched
 * --------------------------------------------------------------------------------------------- */
typedef struct mrcnn_layer mrcnn_layer;

MRCNN_API int mrcnn_layer_create(const char* class_name, const mrcnn_param* params, int n_params,
                                 mrcnn_layer** out_layer);
MRCNN_API int mrcnn_layer_set_weight_data(mrcnn_layer* layer, const void* const* blobs,
                                          const size_t* sizes, int n_blobs);
MRCNN_API int mrcnn_layer_output_shapes(mrcnn_layer* layer, const int64_t (*in_shapes)[5], int n_in,
                                        int64_t (*out_shapes)[5], int* n_out);
MRCNN_API int mrcnn_layer_evaluate(mrcnn_layer* layer, const mrcnn_tensor* inputs, int n_in,
                                   mrcnn_tensor* outputs, int n_out);
MRCNN_API void mrcnn_layer_destroy(mrcnn_layer* layer);

/* Stand-alone numerics helpers that are public in the reference:
 *   IOU(_:_:)  Sources/Mask-RCNN-CoreML/Utils.swift:232  (boxes as (y1,x1,y2,x2) floats; Double arithmetic, Float result).
 * Host-only (no GPU involved, as in the reference). */
MRCNN_API float mrcnn_iou(const float a_yxyx[4], const float b_yxyx[4]);

/* ---------------------------------------------------------------------------------------------
 * The three-model surface.  Xcode generates `MaskRCNN`, `Classifier`, `Mask` classes from the
 * .mlmodel files (Example/iOS Example.xcodeproj/project.pbxproj:25-28); the host uses
 * `MaskRCNN().model` (Example/Source/ViewController.swift:37) and reads outputs "detections" and
 * "mask" (task.py:70-72,87-89).  Here a model is a .mrcw artefact (see DESIGN.md "Artefacts").
 * --------------------------------------------------------------------------------------------- */
typedef enum { MRCNN_MODEL_MASKRCNN = 0, MRCNN_MODEL_CLASSIFIER = 1, MRCNN_MODEL_MASK = 2 } mrcnn_model_kind;
typedef struct mrcnn_model mrcnn_model;

/* Loads weights to the current HIP device, folds BatchNorm, repacks kernels, builds the static
 * schedule.  For MRCNN_MODEL_MASKRCNN the anchors / Classifier / Mask artefacts are taken from the
 * config singleton at load time (like ProposalLayer.init, ProposalLayer.swift:68), and loaded ONCE
 * (the reference re-loads the sub-models on every evaluate, TimeDistributedClassifierLayer.swift:41 —
 * deliberately not reproduced).  max_batch sizes the activation arena (images per predict call).
 * compute_dtype: MRCNN_DEFAULT — what a drop-in host passes (`MaskRCNN()` of ViewController.swift:37 names no precision): a MaskRCNN
 * artefact that carries stored split exponents ("split_exp.<group>" metadata, written by `convert --calibrate`) loads as MRCNN_F32X3
 * — fp32 tensors, every product exact, the mode bench.py's `value` is measured in — with those exponents applied; any other artefact
 * (no calibration stored; the stand-alone Classifier / Mask models) loads as MRCNN_F32.  mrcnn_model_get_int "compute_dtype" returns
 * the resolved mode, "compute_dtype_defaulted" 1 when it was chosen this way.  Explicit modes: MRCNN_F32 (exact-fp32 MFMA, fp32
 * activations — the parity baseline), MRCNN_F16 (fp16 activations and filters, fp32 accumulate, fp32 box path and outputs:
 * BASELINE configs[3]) or MRCNN_F32S (everything stays fp32 in memory; each convolution runs as TWO fp16
 * MFMA passes over a hi/lo split of its fp32 activations against the fp16 filters the artefact stores
 * (task.py:90), fp32 accumulate: products are exact, the split carries 22 of the 24 significand bits —
 * fp32-grade results at several times the fp32-MFMA rate.  Requires fp16-representable filters, which is
 * what the converter writes; an artefact with genuine fp32 filters is refused in this mode).  MRCNN_F32X3 is the same
 * with THREE parts — three MFMA passes instead of two.  The bound that holds: an activation with 0.5 <= |a| < 65504 is
 * represented exactly (its product with an fp16 filter is then exact and only the fp32 summation order differs from an fp32
 * engine); a smaller one is carried to 2^-25 ABSOLUTE, rounded to nearest (the third part reaches the fp16 subnormal
 * step; this also relies on the MFMA not flushing fp16 subnormals, which gfx950 does not), i.e. about 14 significant bits
 * at |a| = 1e-3.  An output is therefore off by at most 2^-25 * sum|w| beyond fp32 summation noise: invisible while a layer's
 * activations are O(1) or larger (every tensor of a BatchNorm-folded trunk), but unlike fp32 the RAW split is not scale-invariant
 * (tests/test_gpu_conv_kernels.py, profiles/r03_split_scale_curve.txt).  mrcnn_model_calibrate_split (below) removes the caveat:
 * a power-of-two exponent per tensor group, folded into the layers at no run-time cost, keeps every tensor's maximum in
 * [2^11, 2^12) — the mode is then fp32-grade for a checkpoint at ANY activation scale (tests/test_gpu_split_scale.py) and a
 * checkpoint whose raw activations would leave the fp16 range runs instead of tripping the watchdog.
 * End-to-end tolerance of MRCNN_F16 (BASELINE configs[3]; tests/test_gpu_fullsize.py::test_fp16_mode_end_to_end_bar, full size,
 * batch 8, against the fp32 CPU oracle): >= 95 % of the detections have a partner with the same class id and a box within 2e-3
 * (normalized coordinates), matched scores within 5e-4, matched masks within 3e-2; the fp32 modes' bars are 1e-4 / 1e-5 / 2e-4
 * with >= 99.9 % matched.
 * In MRCNN_F16, MRCNN_F32S and MRCNN_F32X3 every convolution watches its outputs for values leaving the fp16 range (|v| >= 65504,
 * which the next layer could not read; mrcnn_model_get_int key "range_overflows" counts the predicts it trips on).
 *   - Split modes (MRCNN_F32S / MRCNN_F32X3): such a predict is NOT failed — the reference's CPU path is fp32 activations x fp16 weights
 *     (Conversion/task.py:90) and has no such failure.  The batch, still resident on the device, is measured (one calibration pass),
 *     the split exponents are LOWERED to what it needs (never raised), and the batch is computed again before the call returns
 *     (key "range_recoveries"); mrcnn_maskrcnn_collect does the same for a pipelined batch.  Only an enqueue-only predict
 *     (mrcnn_maskrcnn_predict_async + mrcnn_model_check_range) leaves the re-run to the host.
 *   - MRCNN_F16 (fp16 tensors, inherently range-limited): the synchronous predict fails with MRCNN_ERR_UNSUPPORTED instead of
 *     returning saturated results. */
MRCNN_API int mrcnn_model_load(int kind, const char* path, int max_batch, int compute_dtype,
                               mrcnn_model** out_model);
MRCNN_API void mrcnn_model_destroy(mrcnn_model* model);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the model's own. */
MRCNN_API int mrcnn_model_set_stream(mrcnn_model* model, void* hip_stream);

/* Optional hipGraph replay of predict's launch sequence (captured on the second call at a given batch size;
 * ~200 launches per image batch).  Off by default (measured neutral on MI355X); on = 1 enables it,
 * on = 0 switches back to plain stream launches and frees the captured graphs; the environment variable
 * MRCNN_GRAPH=1 sets the default for new handles.  Replay is bypassed automatically while stage timing or the conv
 * profile is enabled, and when the caller is itself capturing the stream (the launches then join the caller's
 * graph).  mrcnn_model_get_int keys "graph_enabled" / "graph_launches" report the state. */
MRCNN_API int mrcnn_model_enable_graph(mrcnn_model* model, int on);

/* MaskRCNN.prediction(image:) — input `image` (task.py:70-75): RGB 8-bit, H×W = the model's
 * input_image_shape, interleaved (B,H,W,3).  The per-channel mean is subtracted on the GPU.
 * Outputs: `detections` (B, maxDetections, 6) rows (y1,x1,y2,x2,classId,score) normalized,
 * zero-padded; `mask` (B, maxDetections, 28, 28).  Batched predict is an extension (the reference
 * is batch 1).  memspace says where image/detections/masks live (host or device). */
MRCNN_API int mrcnn_maskrcnn_predict(mrcnn_model* model, const uint8_t* rgb, int batch, int height,
                                     int width, int memspace, float* detections, float* masks);
/* `.scaleFit` inside predict (VNCoreMLRequest.imageCropAndScaleOption = .scaleFit, EvaluateCommand.swift:152-157,
 * ViewController.swift:45): images of ANY size height×width (the same for the whole batch) are letterboxed — aspect-preserving
 * bilinear resize, centred, black borders: exactly mrcnn_letterbox_rgb's pixels — into the model's input size INSIDE the
 * pre-processing kernel; the results equal mrcnn_letterbox_rgb + mrcnn_maskrcnn_predict bit for bit.  Boxes come back normalized
 * in the letterboxed frame, like the reference's (EvaluateCommand.swift:203-248); mrcnn_unletterbox_boxes maps rows
 * (y1,x1,y2,x2,...) of `stride` floats back to the source image's normalized frame (host arithmetic, in place). */
MRCNN_API int mrcnn_maskrcnn_predict_scalefit(mrcnn_model* model, const uint8_t* rgb, int batch, int height, int width, int memspace,
                                              float* detections, float* masks);
MRCNN_API int mrcnn_unletterbox_boxes(float* detections, int64_t n, int64_t stride, int src_h, int src_w, int model_h, int model_w);
/* Pipelined host entry — the evaluate loop of EvaluateCommand.swift:167-179 (images handed over one call after the other, the
 * hand-over inside the per-image time) with the hand-over of batch i + 1 OVERLAPPED with the computation of batch i:
 *   mrcnn_maskrcnn_submit   copies a batch of host images (pinned memory makes the copy asynchronous) on the handle's copy
 *                           stream into one of two staging buffers and enqueues its predict behind the copy; returns at once.
 *                           At most two submissions may be in flight.
 *   mrcnn_maskrcnn_collect  waits for the OLDEST submission and copies its records to the host buffers (*batch = its size);
 *                           reports that batch's range-watchdog status like the synchronous predict.
 * The loop:  submit(b0); for i: submit(b[i+1]); collect(results of b[i]).   Results are bit-identical to
 * mrcnn_maskrcnn_predict's (the same launches on the same stream); do not interleave it with the synchronous entry while a
 * submission is in flight.  examples/maskrcnn_predict_stream.c is this loop as a plain-C host. */
MRCNN_API int mrcnn_maskrcnn_submit(mrcnn_model* model, const uint8_t* rgb_host, int batch, int height, int width);
MRCNN_API int mrcnn_maskrcnn_collect(mrcnn_model* model, float* detections_host, float* masks_host, int* batch);
/* Same, but only enqueues on the model's stream (no synchronisation); device buffers only. */
MRCNN_API int mrcnn_maskrcnn_predict_async(mrcnn_model* model, const uint8_t* rgb, int batch, int height,
                                           int width, float* detections, float* masks);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU (one process per GPU of a node; new functionality of the MI355X build — the reference runs one image on one
 * device, the host loop that would drive this is Sources/maskrcnn/EvaluateCommand.swift:146-179).  A batch is split into
 * contiguous blocks of images (blocks differ by at most one image), every rank loads the same artefacts and predicts
 * its block, and ONE ncclAllGather over RCCL/xGMI — issued on the model's stream — hands every rank the fixed-size,
 * zero-padded records of the whole batch in image order:
 *     record = detections (maxDetections × 6 f32) ‖ mask (maxDetections × 28 × 28 f32)     316 000 B at the defaults.
 * Per-image results do not depend on the world size (tests).  RCCL is bound at run time (librccl.so.1).
 *   mrcnn_dist_unique_id   rank 0 creates the 128-byte rendezvous id; the HOST ships it to the other ranks (file, env, pipe)
 *   mrcnn_dist_init        joins the communicator on the calling thread's current HIP device
 *   mrcnn_dist_shard       [begin, end) of `rank` in a batch of `global_batch` images (host arithmetic, no GPU needed)
 *   mrcnn_dist_record_floats  floats per image record
 *   mrcnn_dist_all_gather_records  local results (end-begin images, `memspace`) → all `global_batch` results (`memspace`)
 *   mrcnn_maskrcnn_predict_sharded  the whole step: every rank passes the SAME global batch (global_batch, H, W, 3) and
 *                          receives detections (global_batch, maxDet, 6) / masks (global_batch, maxDet, 28, 28)
 * A rank never skips the collective: what it contributes is a SLOT = its records zero-padded to the largest shard + a
 * 4-word trailer [status, images, 0, 0].  A rank whose local predict failed (HIP error, the data-dependent fp16-range
 * watchdog) sends zeroed records with its status, and EVERY rank returns that status after the gather (its own message on
 * the failing rank, "rank r failed ..." on the others) — nobody is left blocked in ncclAllGather.
 *   mrcnn_dist_all_gather_records_async / mrcnn_dist_wait  the same exchange (device buffers only) issued on the handle's
 *                          own stream behind the model's stream: returns at once, the model's NEXT predict overlaps it (the
 *                          model's stream only waits until the results are packed before it may overwrite them);
 *                          mrcnn_dist_wait joins it and reports remote failures.  One exchange in flight per handle.
 *   mrcnn_dist_plan        host arithmetic of one exchange, no GPU needed: table[4*r + {0,1,2,3}] = {first image, end image,
 *                          float offset of rank r's slot in the gathered buffer, record floats rank r contributes};
 *                          *slot_floats = floats per slot
 *   mrcnn_dist_simulate_host  the pack -> concatenate (what ncclAllGather does) -> unpack code of the device path run on
 *                          host buffers for all `world` ranks in one process (detections[r] / masks[r] = rank r's local
 *                          results, status[r] optional): the seam through which the layout is tested at world sizes the
 *                          build machine does not have.  status[r] == MRCNN_DIST_ABORTED models a rank that could not even
 *                          enqueue a zeroed slot and tore the communicator down (ncclCommAbort): the call fails with
 *                          MRCNN_ERR_HIP — what every peer's ncclAllGather does then — and writes nothing */
#define MRCNN_DIST_ABORTED (-1)
typedef struct mrcnn_dist mrcnn_dist;
MRCNN_API int mrcnn_dist_unique_id(uint8_t* id128);
MRCNN_API int mrcnn_dist_init(int rank, int world, const uint8_t* id128, mrcnn_dist** out);
MRCNN_API void mrcnn_dist_destroy(mrcnn_dist* dist);
MRCNN_API int mrcnn_dist_shard(int global_batch, int world, int rank, int* begin, int* end);
MRCNN_API int64_t mrcnn_dist_record_floats(int max_detections, int mask_size);
MRCNN_API int mrcnn_dist_all_gather_records(mrcnn_dist* dist, mrcnn_model* model, const float* detections, const float* masks,
                                            int global_batch, int memspace, float* out_detections, float* out_masks);
MRCNN_API int mrcnn_maskrcnn_predict_sharded(mrcnn_dist* dist, mrcnn_model* model, const uint8_t* rgb, int global_batch,
                                             int height, int width, int memspace, float* detections, float* masks);
MRCNN_API int mrcnn_dist_all_gather_records_async(mrcnn_dist* dist, mrcnn_model* model, const float* detections, const float* masks,
                                                  int global_batch, float* out_detections, float* out_masks);
MRCNN_API int mrcnn_dist_wait(mrcnn_dist* dist);
/* Range recoveries seen by the LAST completed exchange of mrcnn_maskrcnn_predict_sharded (split modes; §"scale-aware split" below): every rank's
 * slot carries, beside its status word, how many times its local predict lowered its split exponents in that call.  *ranks = how many ranks
 * did (0: every rank still holds the exponent vector it was given), per_rank[r] (optional, `world` entries) = rank r's count.  A rank that
 * recovered computes later images with other exponents than its peers: a job that wants per-image results independent of the rank then takes
 * the vector with the LOWEST exponents (mrcnn_model_get_split_exponents on the ranks that recovered, element-wise minimum — the host's own
 * all-reduce or the same record exchange) and gives it to every rank (mrcnn_model_set_split_exponents).  Identical on every rank, no GPU work. */
MRCNN_API int mrcnn_dist_recovered(mrcnn_dist* dist, int32_t* per_rank, int* ranks);
/* Which RCCL this library bound (dist.hip binds it at run time): 1 = the copy the process had already mapped (e.g. the one
 * PyTorch ships under the soname librccl.so.1 — taken first, so that a process holds ONE RCCL), 0 = its own dlopen of
 * librccl.so.1, -1 = RCCL could not be loaded.  Loads RCCL when it is not bound yet; needs no GPU. */
MRCNN_API int mrcnn_dist_rccl_shared(void);
MRCNN_API int mrcnn_dist_plan(int global_batch, int world, int max_detections, int mask_size, int64_t* table, int64_t* slot_floats);
MRCNN_API int mrcnn_dist_simulate_host(int world, int global_batch, int max_detections, int mask_size, const float* const* detections,
                                       const float* const* masks, const int32_t* status, float* out_detections, float* out_masks,
                                       int32_t* status_out);

/* Classifier.prediction(feature_map:) (task.py:106-113): feature_map (n,256,7,7) CHW →
 * probabilities (n, numClasses), bounding_boxes (n, numClasses*4) class-major. */
MRCNN_API int mrcnn_classifier_predict(mrcnn_model* model, const float* feature_map, int n, int memspace,
                                       float* probabilities, float* bounding_boxes);
/* Mask.prediction(feature_map:) (task.py:94-101): feature_map (n,256,14,14) CHW → masks (n,numClasses,28,28). */
MRCNN_API int mrcnn_mask_predict(mrcnn_model* model, const float* feature_map, int n, int memspace,
                                 float* masks);

/* Introspection: "num_classes", "image_height", "image_width", "max_proposals", "max_detections", "num_anchors",
 * "pre_nms_max_proposals", "pre_nms_count" (= min(num_anchors, pre_nms_max_proposals)), "mask_size" (side of the
 * square masks predict returns: 2 × the mask pool size = 28), "max_batch", "compute_dtype", "range_overflows", "range_recoveries",
 * "graph_enabled", "graph_launches", "gpu_busy_us" / "predict_calls" (GPU time between the first and the last command of
 * the synchronous predicts of this handle, HIP events on the model's stream, and their count); any other key is looked up in
 * the artefact's integer metadata. */
MRCNN_API int mrcnn_model_get_int(mrcnn_model* model, const char* key, int64_t* value);

/* Debug taps for parity tests: copies a named intermediate of the last predict (image b) to a host
 * buffer of `capacity` floats and reports its element count.  Names: "rpn_probs" (A,2),
 * "rpn_deltas" (A,4), "P2".."P5" (H,W,256 NHWC), "topk_idx" (int32 stored as float-exact values),
 * "boxes_sorted" (n,4), "rois" (maxProposals,4), "pooled" (maxProposals,7,7,256 NHWC),
 * "cls_probs" (maxProposals,nc), "cls_bbox" (maxProposals,nc*4), "cls6" (maxProposals,6),
 * "detections" (maxDetections,6), "pooled_mask" (maxDetections,14,14,256 NHWC), "mask" (maxDetections,784),
 * "keep_count" (1), "mask_row_flags" (maxDetections: the mask layer's removeZeros predicate per detection row). */
MRCNN_API int mrcnn_model_read_tensor(mrcnn_model* model, const char* name, int image_index,
                                      float* host_dst, int64_t capacity, int64_t* count);

/* fp16-range watchdog for the enqueue-only path: mrcnn_maskrcnn_predict reports an activation that left the fp16
 * range (MRCNN_F16 / MRCNN_F32S / MRCNN_F32X3) as MRCNN_ERR_UNSUPPORTED when it synchronises; _predict_async cannot.
 * Call this after the stream work of an async predict has been ordered before the caller's consumer: it
 * synchronises the model's stream and sets *tripped = 1 when the LAST predict's results are not valid
 * (counted in "range_overflows" like the synchronous path).  Always 0 in MRCNN_F32. */
MRCNN_API int mrcnn_model_check_range(mrcnn_model* model, int* tripped);

/* ---------------------------------------------------------------------------------------------
 * Scale-aware split — what makes MRCNN_F32X3 / MRCNN_F32S fp32-grade at ANY activation scale.
 * The split modes carry an fp32 activation exactly while 0.5 <= |a| < 65504 and to 2^-25 ABSOLUTE below, so a checkpoint
 * whose tensors sit at 1e-3 would lose accuracy (and one at 1e5 would trip the range watchdog) where the reference's CPU path
 * — fp32 activations, fp16 weights (Conversion/task.py:90) — is scale-free.  Every tensor a split convolution reads belongs to
 * a GROUP with an exponent e: it is STORED as 2^e * value, an exact operation folded at no run-time cost into the producer's
 * BatchNorm scale / shift and undone in the consumer's (ReLU, max-pool, the bilinear sampler and the residual add commute).
 * With every e = 0 (the state after mrcnn_model_load) nothing changes.
 *   mrcnn_model_calibrate_split  one predict on the given images collects max |a| per group, picks e with max |a| * 2^e in
 *                          [2^11, 2^12) (16x head room under the fp16 range, everything above 2^-13 of the maximum exact), and a
 *                          second predict verifies the choice and counts the inputs a split still cannot carry exactly;
 *                          apply = 0 only diagnoses (exponents untouched).  Outputs, taps and every stage after the convolutions
 *                          are in true scale either way; results are bit-identical for any batch split as before.
 *   mrcnn_model_get_int    "split_small_inputs" (non-zero inputs below 2^-8 of their tensor's maximum), "split_inexact_inputs"
 *                          (non-zero stored inputs below 0.5: carried to 2^-25 absolute, i.e. <= 2^-36 of the maximum once
 *                          calibrated), "split_inputs_counted", "split_min_exponent" / "split_max_exponent", "split_calibrated",
 *                          "split_groups" — next to "range_overflows"
 *   mrcnn_model_split_group_stat   per group: name, exponent, max |a|, the three counters
 *   mrcnn_model_get/set_split_exponents   the exponents as a vector (one per group, 0 for the groups fp32 arithmetic consumes):
 *                          a sharded job calibrates on one rank — or offline — and sets the same vector on every rank.
 *   stored exponents       `python -m mask-rcnn-coreml_amd.convert ... --calibrate <images>` writes the vector into MaskRCNN.mrcw
 *                          ("split_exp.<group>" metadata); mrcnn_model_load applies it in the split modes, so the drop-in
 *                          MaskRCNN().prediction(image) (ViewController.swift:37) runs calibrated without any extra call
 *                          ("split_exponents_from_artefact" = 1, "split_calibrated" = 1).
 *   range recovery         a batch that leaves the calibrated range lowers the exponents it needs and is computed again inside the
 *                          call (see mrcnn_model_load above; "range_recoveries").  Later batches use the lowered vector: a sharded job
 *                          that wants bit-equal results across ranks after a recovery re-distributes it with get / set.
 * MRCNN_F32 and MRCNN_F16 models return MRCNN_ERR_UNSUPPORTED from calibrate / set. */
typedef struct mrcnn_split_group_stat {
    char    name[48];
    int32_t exponent;
    int32_t fixed;            /* 1: consumed by fp32 arithmetic (logits, deltas, probabilities): exponent stays 0 */
    float   absmax;           /* max |a| of the true values over the calibration images */
    int64_t small_inputs, inexact_inputs, inputs_counted;
} mrcnn_split_group_stat;
MRCNN_API int mrcnn_model_calibrate_split(mrcnn_model* model, const uint8_t* rgb, int batch, int height, int width, int memspace, int apply);
MRCNN_API int mrcnn_model_split_group_stat(mrcnn_model* model, int index, mrcnn_split_group_stat* out);
MRCNN_API int mrcnn_model_get_split_exponents(mrcnn_model* model, int32_t* exponents, int capacity, int* count);
MRCNN_API int mrcnn_model_set_split_exponents(mrcnn_model* model, const int32_t* exponents, int count);

/* PyramidROIAlign (PyramidROIAlignLayer.swift:79-181) on the engine's own layout: four NHWC maps (H_l, W_l, C),
 * dtype MRCNN_F32 or MRCNN_F16; rois rows (y1,x1,y2,x2,...) of `roi_stride` floats; out (n_rois, pool, pool, C)
 * in the maps' dtype.  row_flags (optional, n_rois int32): the removeZeros predicate the mask layer applies to a
 * pooled row (TimeDistributedClassifierLayer.swift:116-127: kept iff every element != 0) evaluated on the fp32
 * samples BEFORE the store rounds them — in fp16 a tiny non-zero sample would otherwise flush to zero and drop a
 * valid detection.  This is what the fused engine uses; the stand-alone MLCustomLayer entry takes CHW fp32. */
MRCNN_API int mrcnn_roi_align_nhwc(const void* const maps[4], const int heights[4], const int widths[4], int channels,
                                   int dtype, const float* rois, int64_t roi_stride, int n_rois, int pool,
                                   double image_w, double image_h, int memspace, void* out, int32_t* row_flags);

/* Per-stage GPU time of the last predict in milliseconds (HIP events on the model's stream).
 * Stage names mirror the reference's os_signpost intervals (ProposalLayer.swift:105-194 etc.):
 * "Trunk", "Proposal-Eval", "PyramidROIAlign-Eval", "TimeDistributedClassifierLayer-Eval",
 * "Detection-Eval", "PyramidROIAlign-Eval-Mask", "TimeDistributedMask-Eval".
 * Only collected after mrcnn_model_enable_timing(model, 1). */
MRCNN_API int mrcnn_model_enable_timing(mrcnn_model* model, int on);
MRCNN_API int mrcnn_model_stage_ms(mrcnn_model* model, const char* stage, float* ms);

/* Test and measurement entry points (the convolution micro-benchmark hook, the per-kernel live profile, the
 * single-convolution parity entry and the process-wide kernel-selection knobs) are exported by the same library but
 * declared in include/maskrcnn_hip_test.h: they are not part of the drop-in surface — nothing in the reference's
 * interface (ProposalLayer.swift:52-103, MaskRCNNConfig.swift:10-18) is a debug knob. */

/* ---------------------------------------------------------------------------------------------
 * Result decoding — Detection.detectionsFromFeatureValue (Sources/Mask-RCNN-CoreML/
 * Detection.swift:23-62) and maskFromFeatureValue (:64-99).  Host-side, like the reference.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t index;        /* row in the detections array                       (Detection.swift:17) */
    double  x, y, w, h;   /* boundingBox CGRect(x: x1, y: y1, width, height)   (Detection.swift:55) */
    int64_t class_id;
    double  score;
} mrcnn_detection;

/* Keeps rows with score > 0.7 (Detection.swift:38).  Writes at most `capacity` records; returns the
 * number kept in *count. */
MRCNN_API int mrcnn_detections_decode(const float* detections, int64_t n_rows, int64_t row_stride,
                                      mrcnn_detection* out, int64_t capacity, int64_t* count);
/* Anchors on demand (the reference's own TODO, MaskRCNNConfig.swift:14): writes the (A,4) float32
 * normalized (y1,x1,y2,x2) anchors that the converter dumps to anchors.bin (task.py:173-176) for the
 * default scales (32..512), ratios (0.5,1,2) and strides (4..64).  Host code.  Call with out = NULL to
 * query *count = A. */
MRCNN_API int mrcnn_generate_anchors(int image_h, int image_w, float* out, int64_t capacity, int64_t* count);

/* Letterbox: Vision's `.scaleFit` (EvaluateCommand.swift:157, ViewController.swift:45) on
 * the GPU — RGB8 (h,w,3) resized with preserved aspect ratio (bilinear, half-pixel centres) and centred in
 * an (H,W,3) canvas with black borders.  _geometry returns the content size and offsets (host arithmetic)
 * so that callers can map normalized boxes back to the source image. */
MRCNN_API int mrcnn_letterbox_geometry(int h, int w, int H, int W, int* nh, int* nw, int* pad_y, int* pad_x);
MRCNN_API int mrcnn_letterbox_rgb(const uint8_t* src, int h, int w, int memspace, uint8_t* dst, int H, int W);

/* Mask paste: per-instance 28×28 sigmoid masks → full-resolution binary masks
 * (n, image_h, image_w) uint8 {0,1}: resize to the detection's box and threshold.  Replaces what the
 * example app does with CoreGraphics when drawing (Example/Source/DetectionRenderer.swift:13-24).
 * Box pixels = round-half-even(y*(H-1)) with +1 on the far edge (Matterport denorm_boxes), bilinear
 * with half-pixel centres, `>= threshold`; rows with score <= 0 give empty masks. image_w % 4 == 0. */
MRCNN_API int mrcnn_paste_masks(const float* detections, int64_t det_stride, const float* masks, int n, int mask_size,
                                int image_h, int image_w, float threshold, int memspace, uint8_t* out);
/* 28×28 mask → 8-bit: UInt8(255 - v/2*255) (Detection.swift:83-85). */
MRCNN_API int mrcnn_mask_to_u8(const float* mask, int64_t n, uint8_t* out);
/* The same on Double input — the type Core ML hands maskFromFeatureValue (Detection.swift:77); for hosts that widen the fp32
 * mask of mrcnn_maskrcnn_predict first the result is identical (float → double is exact). */
MRCNN_API int mrcnn_mask_to_u8_f64(const double* mask, int64_t n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* MASKRCNN_HIP_H */
