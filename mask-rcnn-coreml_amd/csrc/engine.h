// engine.h — model artefacts (.mrcw), weight packing and the static execution plans.
#pragma once
#include <functional>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

namespace mrcnn {

// ---- .mrcw reader (format: mask-rcnn-coreml_amd/weights.py) --------------------------------------
struct MrcwTensor {
    int dtype = 0;                 // 0 = f32, 2 = f16
    std::vector<uint32_t> dims;
    const unsigned char* data = nullptr;
    size_t nbytes = 0;
    size_t count() const { size_t n = 1; for (auto d : dims) n *= d; return n; }
};
struct MrcwFile {
    std::string path;
    std::vector<unsigned char> buf;
    std::map<std::string, int64_t> ints;
    std::map<std::string, double> doubles;
    std::map<std::string, std::string> strings;
    std::map<std::string, MrcwTensor> tensors;
    void load(const std::string& path);
    int64_t get_int(const std::string& k) const;
    int64_t get_int(const std::string& k, int64_t dflt) const;
    double get_double(const std::string& k, double dflt) const;   // accepts ints too
    std::string get_string(const std::string& k) const;
    const MrcwTensor& tensor(const std::string& name) const;
    std::vector<float> floats(const std::string& name) const;     // any dtype → fp32
};

// ---- packed convolution weights on the device ----------------------------------------------------
// compute mode (mrcnn_model_load's compute_dtype) → element type of activations / of filters
// (filters of MRCNN_F32X3 are fp16 too; the value MRCNN_F32X3 on the filter side only tags the three-part split)
inline int mode_act(int mode) { return (mode == MRCNN_F32S || mode == MRCNN_F32X3) ? MRCNN_F32 : mode; }
inline int mode_wgt(int mode) { return mode == MRCNN_F32 ? MRCNN_F32 : (mode == MRCNN_F32X3 ? MRCNN_F32X3 : MRCNN_F16); }

struct PackedConv {
    DevBuf wgt, scale, shift;     // wgt in `wdtype`; scale/shift always fp32
    std::vector<float> h_scale, h_shift;   // host copies of scale / shift (the scale-aware split derives per-op copies from them)
    DevBuf wgt_halo;              // 3x3 layers of the split modes: the same filters re-tiled for the halo kernel (conv_halo_pack)
    DevBuf wgt_c3h;               // fp16 mode, 3x3 layers with 256 | 512 output columns: the filters in k_conv3x3_h's stream order (conv3x3h_pack)
    DevBuf wgt_frag;              // fp16 mode, the layers of C4's identity bottlenecks: the same filters in MFMA-fragment order (bneck_pack_frag)
    int Cin = 0, Cout = 0, KH = 1, KW = 1, Npad = 0;
    int dtype = MRCNN_F32;        // activations
    int wdtype = MRCNN_F32;       // filters (fp16 with fp32 activations = split mode, MRCNN_F32S)
};

struct Tensor4 {   // dense NHWC activation (element type = the model's compute dtype)
    void* p = nullptr;
    int H = 0, W = 0, C = 0;
    long sB() const { return (long)H * W * C; }
};

using Op = std::function<void(hipStream_t, int /*batch*/)>;

// ---- scale-aware split (round 4) ----------------------------------------------------------------------
// The split modes carry an activation exactly only while 0.5 <= |a| < 65504 (below: 2^-25 absolute).  Every tensor a split
// convolution reads therefore belongs to a GROUP with an exponent e: the tensor is stored as 2^e * (its true value), an exact
// operation folded into the producer's scale / shift (scale' = scale * 2^(e_out - e_in), shift' = shift * 2^e_out), undone by
// the consumer the same way.  ReLU, max-pool, the bilinear sampler and the residual add (tensors of one residual chain share a
// group) commute with it, so with every e = 0 nothing changes and with calibrated exponents the mode is fp32-grade at any
// activation scale.  Exponents are chosen by Model::calibrate_split from one predict (max |a| * 2^e in [2^11, 2^12): 16x head
// room under the fp16 range the watchdog guards) or set by the host (identical on every rank of a sharded job).
struct SplitGroup {
    std::string name;
    int exp = 0;
    bool fixed = false;            // consumed by non-split arithmetic (logits, box deltas, probabilities): stays 0
    float absmax = 0.f;            // of the true values, over the calibration batch
    long long small = 0, inexact = 0, counted = 0;   // diagnostics of the last calibrate / diagnose pass (elements)
};
struct ScaledOp {                  // a convolution whose scale / shift carry the exponents of its input / output groups
    const PackedConv* pc = nullptr;
    int g_in = 0, g_out = 0;
    DevBuf scale, shift;           // the per-op copies the launch reads (rewritten in place by apply_split_exponents)
};
// called after a layer has been enqueued while a calibration pass is active (group, tensor, elements)
using SplitObserver = std::function<void(hipStream_t, int, const void*, size_t)>;

struct Arena {
    char* base = nullptr;
    size_t off = 0;
    void* alloc_e(size_t n_elems, int dtype) { return alloc_b(n_elems * (dtype == MRCNN_F16 ? 2 : 4)); }
    float* alloc_f(size_t n_floats)
    {
        size_t bytes = (n_floats * 4 + 255) / 256 * 256;
        char* r = base ? base + off : nullptr;
        off += bytes;
        return reinterpret_cast<float*>(r);
    }
    void* alloc_b(size_t bytes)
    {
        bytes = (bytes + 255) / 256 * 256;
        char* r = base ? base + off : nullptr;
        off += bytes;
        return r;
    }
};

// Box head: Classifier.mlmodel (task.py:106-116) + TimeDistributedClassifierLayer post-processing.
struct ClassifierHead {
    int nc = 0, pool = 7, C = 256, cap = 0, dtype = MRCNN_F32;
    PackedConv fc1, fc2, fc3;
    DevBuf arena;
    void *h1 = nullptr, *h2 = nullptr, *stage_in = nullptr;           // compute dtype
    float *lb = nullptr, *probs = nullptr, *bbox = nullptr, *cls6 = nullptr;
    int mode = MRCNN_F32;          // compute mode the head was loaded with (MRCNN_F32 | MRCNN_F16 | MRCNN_F32S)
    ScaledOp sop[3];               // fc1 / fc2 / fc3 with the exponents of pooled → h1 → h2 → logits (0)
    int grp[3] = {-1, -1, -1};     // split groups of pooled, h1, h2 (owned by the model; -1 = stand-alone head, exponent 0)
    SplitObserver observe;
    void load(const MrcwFile& f, int capacity_rows, int mode);
    // pooled: n rows of pool*pool*C elements in (h,w,c) order, contiguous, compute dtype.
    void forward(hipStream_t s, const void* pooled_nhwc, int n, float* cls6_out, long cls6_stride);
};

// Mask head: Mask.mlmodel (task.py:94-104).
struct MaskHead {
    int nc = 0, pool = 14, C = 256, cap = 0, dtype = MRCNN_F32;
    PackedConv conv[4], deconv, final_full;
    DevBuf final_w, final_b;          // [nc][C], [nc] fp32 for the selected-class kernel
    DevBuf arena;
    void *t0 = nullptr, *t1 = nullptr, *feat = nullptr, *stage_in = nullptr;   // compute dtype
    float* full = nullptr;
    int mode = MRCNN_F32;          // compute mode the head was loaded with (MRCNN_F32 | MRCNN_F16 | MRCNN_F32S)
    ScaledOp sop[5];               // conv1..4 + the deconvolution, exponents pooled_mask → t1 → t2 → t3 → t4 → 0
    int grp[5] = {-1, -1, -1, -1, -1};   // split groups of pooled_mask and of the four 3x3 outputs
    SplitObserver observe;
    void load(const MrcwFile& f, int capacity_rows, int mode);
    // pooled: n rows of 14*14*C NHWC → feat (n, 28*28, C) = ReLU(deconv)
    // sel_partial != nullptr: the deconvolution leaves the selected-class partial dots instead of feat (ConvDesc::sel_partial)
    void forward_features(hipStream_t s, const void* pooled_nhwc, int n, const int32_t* sel_cid = nullptr, float* sel_partial = nullptr);
    // feat → all-class sigmoid masks, NHWC (n, 784, nc) in `full`
    void forward_full(hipStream_t s, int n);
};

bool engine_debug_set(const char* key, int value);     // "mask_fused"

struct StageTimer {
    bool enabled = false;
    std::vector<std::string> names;
    std::vector<hipEvent_t> ev;       // names.size() + 1 events
    std::map<std::string, float> ms;
    void begin(hipStream_t s);
    void mark(hipStream_t s, const char* name);
    void finish();
    ~StageTimer();
};

struct Model {
    int kind = 0;
    int max_batch = 1;
    bool mode_defaulted = false;   // loaded with MRCNN_DEFAULT: `mode` was chosen from the artefact (stored split exponents -> MRCNN_F32X3)
    int mode = MRCNN_F32;       // compute mode: MRCNN_F32 | MRCNN_F16 | MRCNN_F32S
    int dtype = MRCNN_F32;      // element type of the activations = mode_act(mode)
    MrcwFile file;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // config
    std::string arch;
    int H = 0, W = 0, nc = 0, pre_nms = 0, max_prop = 0, max_det = 0, na = 3, A = 0, K = 0;
    float mean[3] = {0, 0, 0};
    float prop_std[4], det_std[4];
    float prop_nms_thr = 0.7f, det_score_thr = 0.7f, det_nms_thr = 0.3f;
    int cls_pool = 7, mask_pool = 14;
    double roi_img_w = 0, roi_img_h = 0;
    // weights
    std::map<std::string, PackedConv> convs;
    DevBuf anchors;
    DevBuf rpn_head_frag;             // split modes: the RPN heads' filters in the halo kernel's head-fragment order (conv_halo_pack_head)
    ClassifierHead cls_head;
    MaskHead mask_head;
    // activations
    DevBuf arena;
    std::vector<Op> trunk_ops;
    std::vector<std::unique_ptr<DevBuf>> stage_tabs;   // fp16 mode: per-block operand tables + tile counters of the whole-stage bottleneck launches (kernels.h: bneck_stage_launch)
    struct Tap { void* base; long per_image; int dtype; int group = -1; };     // group >= 0: stored as 2^e * value (read_tensor undoes it)
    std::map<std::string, Tap> taps;    // name → (base, per-image elements, element type)
    uint8_t* d_rgb = nullptr;
    void* stem_in = nullptr;          // the stem's zero-padded NHWC4 / NHWC8 staging tensor (written by the pre-processing op = trunk_ops[0])
    DevBuf fit_src;                   // predict_scalefit: the source images on the device
    float *rpn_logits = nullptr, *rpn_probs = nullptr, *rpn_deltas = nullptr, *rois = nullptr;
    float *cls6 = nullptr, *detections = nullptr, *mask_out = nullptr;
    void *pooled = nullptr, *pooled_mask = nullptr;     // compute dtype
    Tensor4 P[4];
    ProposalWorkspace prop_ws;
    DetectionWorkspace det_ws;
    MaskSelectWorkspace msel_ws;
    float* mask_partial = nullptr;     // [B][max_det][28*28][2] partial selected-class dots of the fused mask tail
    bool fuse_mask_tail = true;        // MRCNN_FUSE_MASK_TAIL=0 / mrcnn_debug_set("mask_fused", 0): deconvolution output + k_mask_select
    StageTimer timer;
    ConvProfile conv_profile;
    ConvScratch conv_scratch;         // scratch of the convolution family, allocated at load in the split modes (kernels.h), freed with the model
    // Optional: the ~200 launches of one predict captured once per batch size and replayed as a hipGraph
    // (the pipeline is static: every data-dependent count lives in device memory).  Off by default —
    // measured neutral on MI355X (the runtime already keeps the queue full; the gaps
    // between kernels are the GPU's own dispatch) — on with mrcnn_model_enable_graph / MRCNN_GRAPH=1.
    // Bypassed while stage timing / the conv profiler record events between launches and when the
    // caller is itself capturing the stream.
    struct GraphSlot { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int eager_runs = 0; };
    std::map<int, GraphSlot> graphs;
    bool use_graph = false;
    // ---- scale-aware split ------------------------------------------------------------------------------------------
    std::vector<SplitGroup> sgroups;
    std::vector<std::unique_ptr<ScaledOp>> sops;       // the trunk's convolutions (stable addresses: the ops hold pointers)
    int g_P[4] = {-1, -1, -1, -1}, g_rpn[5] = {-1, -1, -1, -1, -1}, g_pooled = -1, g_pooled_mask = -1;
    int calib_phase = 0;              // 0 off | 1 collect max |a| per group | 2 count the inputs the split cannot carry exactly
    DevBuf calib_buf;                 // per group: uint32 max bits | 3 x uint64 counters
    bool split_calibrated = false;
    int new_split_group(const std::string& name, bool fixed = false);
    ScaledOp* new_scaled_op(const PackedConv* pc, int g_in, int g_out);
    void apply_split_exponents();     // rewrites every op's scale / shift copy from the groups' exponents (synchronises the stream)
    void observe_split(hipStream_t s, int group, const void* x, size_t n);
    // one predict with every exponent 0 collecting max |a| per group, exponents chosen, (apply) a second predict that verifies
    // the choice and counts the inputs a split still cannot carry exactly; apply = false: diagnose only, exponents untouched
    void calibrate_split(const uint8_t* rgb, int batch, int h, int w, int memspace, bool apply);
    // phase 1 of a calibration on `batch` images (fit: of size h x w, letterboxed): fills sgroups[].absmax with the true maxima (every
    // exponent is left at the uniform value the pass ran with: the caller chooses and applies the new ones).  resident: the images are
    // the ones the last predict left on the device (d_rgb / fit_src): no copy
    void measure_split_groups(const uint8_t* rgb, int batch, int h, int w, int memspace, bool fit, bool resident);
    static int split_exponent_for(float absmax);          // max |a| * 2^e in [2^11, 2^12)
    // Range recovery (round 5): a predict of a split mode whose activations left the calibrated range is NOT failed — the reference's fp32
    // path has no such failure (Conversion/task.py:90: only the weights are fp16).  The batch still resident on the device is measured
    // (phase 1 above), every group's exponent is LOWERED to what this batch needs (never raised: earlier batches stay inside), and the
    // batch is computed again; "range_recoveries" of mrcnn_model_get_int counts such calls.  false: not a split mode / still out of range.
    bool recover_range(int batch, int h, int w, bool fit);
    long range_recoveries = 0;
    bool exponents_from_artefact = false;      // the exponent vector came with MaskRCNN.mrcw (convert.py --calibrate)
    DevBuf range_flag;          // 4 B: set by the conv epilogues in MRCNN_F16 / MRCNN_F32S when an activation leaves the fp16 range
    long range_overflows = 0;   // predicts that tripped it
    long graph_launches = 0;
    // GPU time of the synchronous predicts of this handle (one event pair per call on the model's stream, read at the
    // call's own synchronisation): "gpu_busy_us" / "predict_calls" of mrcnn_model_get_int — bench.py reports it live
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;
    double gpu_busy_ms = 0;
    long predict_calls = 0;
    // ---- pipelined host entry (mrcnn_maskrcnn_submit / _collect): two batches in flight -----------------------------------
    // The uint8 images of batch i + 1 cross PCIe on a copy stream while batch i computes; the records of a finished batch wait
    // in their own device slot until the host collects them (EvaluateCommand.swift:167-179 hands images over one by one: the
    // hand-over is part of its per-image time).
    struct PipeSlot {
        DevBuf rgb, det, mask, flag;
        hipEvent_t ev_in = nullptr, ev_done = nullptr;
        int batch = 0;
        bool busy = false;
    };
    PipeSlot pipe[2];
    hipStream_t pipe_in = nullptr, pipe_out = nullptr;     // H2D of the images / D2H of the records
    long pipe_submitted = 0, pipe_collected = 0;
    void submit(const uint8_t* rgb_host, int batch, int h, int w);
    void collect(float* det_host, float* masks_host, int* batch_out);
    // Held by the stand-alone TimeDistributed*Layer plugins around stage → forward → unstage: every layer instance
    // shares the cached sub-model's head scratch (stage_in, h1/h2, cls6, feat, full) but launches on its own stream.
    std::mutex eval_mu;

    ~Model();
    void load(int kind, const std::string& path, int max_batch, int dtype);
    void build_maskrcnn();
    // fit = true: the images are h×w of ANY size, letterboxed into the model's H×W inside the pre-processing kernel (.scaleFit)
    // det / masks may be nullptr for internal passes (calibration, recovery measurement): nothing is copied out;
    // resident: the images already sit in d_rgb / fit_src (no input copy)
    void predict(const uint8_t* rgb, int batch, int h, int w, int memspace, float* det, float* masks, bool sync, bool fit = false, bool resident = false);
    // d_rgb (or, with fit geometry, fit_src) → detections / mask_out, launches only
    void enqueue_pipeline(hipStream_t s, int batch, const int* fit = nullptr);
    void drop_graphs();
    void read_tensor(const std::string& name, int image, float* dst, int64_t cap, int64_t* count);
};

// Helpers shared with api.hip
PackedConv pack_conv_oihw(const MrcwFile& f, const std::string& conv, const std::string& bn, int dtype);
void run_conv_dense(hipStream_t s, const PackedConv& pc, const void* in, int B, int H, int W, void* out, int stride,
                    int pad, int act, const void* res = nullptr, int out_f32 = 0, const ScaledOp* sop = nullptr);
void init_scaled_op(ScaledOp& op, const PackedConv* pc, int g_in, int g_out);      // allocates the copies = the base scale / shift

}  // namespace mrcnn

// the opaque handle of the C ABI (include/maskrcnn_hip.h)
struct mrcnn_model {
    mrcnn::Model m;
};
