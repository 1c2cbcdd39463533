// api.hip — the C ABI of include/maskrcnn_hip.h: config singleton, the five custom-layer plugins,
// the three-model surface, result decoding and the convolution micro-benchmark hook.
#include <math.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <random>

#include "engine.h"

using namespace mrcnn;

// ================================================================================================
// misc
// ================================================================================================
extern "C" const char* mrcnn_last_error(void) { return mrcnn::last_error(); }
extern "C" const char* mrcnn_version(void) { return "maskrcnn_hip 0.1 (gfx950)"; }
extern "C" int mrcnn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ================================================================================================
// MaskRCNNConfig.defaultConfig (MaskRCNNConfig.swift:10-18)
// ================================================================================================
namespace {
std::mutex g_cfg_mu;
struct OptStr { bool set = false; std::string v; };
OptStr g_anchors, g_classifier, g_mask;
int set_path(OptStr& o, const char* p)
{
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    o.set = p != nullptr;
    o.v = p ? p : "";
    return MRCNN_OK;
}
const char* get_path(OptStr& o)
{
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    return o.set ? o.v.c_str() : nullptr;
}
}  // namespace
extern "C" int mrcnn_config_set_anchors_path(const char* p) { return set_path(g_anchors, p); }
extern "C" int mrcnn_config_set_classifier_path(const char* p) { return set_path(g_classifier, p); }
extern "C" int mrcnn_config_set_mask_path(const char* p) { return set_path(g_mask, p); }
extern "C" const char* mrcnn_config_get_anchors_path(void) { return get_path(g_anchors); }
extern "C" const char* mrcnn_config_get_classifier_path(void) { return get_path(g_classifier); }
extern "C" const char* mrcnn_config_get_mask_path(void) { return get_path(g_mask); }

// ================================================================================================
// tensor staging helpers
// ================================================================================================
namespace {

void check_f32(const mrcnn_tensor& t, const char* what)
{
    MRCNN_REQUIRE(t.data != nullptr, MRCNN_ERR_INVALID, "%s: null data pointer", what);
    MRCNN_REQUIRE(t.dtype == MRCNN_F32, MRCNN_ERR_INVALID, "%s: dtype must be Float32 (the layers assert it, ProposalLayer.swift:108)", what);
}

// Copies n rows of `len` floats (source row stride `stride` elements) into a dense device buffer.
const float* stage_rows(const void* src, int memspace, long n, long len, long stride, DevBuf& tmp)
{
    if (memspace == MRCNN_DEVICE && stride == len) return static_cast<const float*>(src);
    tmp.alloc((size_t)(n > 0 ? n : 1) * len * 4);
    if (n <= 0) return tmp.as<float>();
    HIP_CHECK(hipMemcpy2D(tmp.p, (size_t)len * 4, src, (size_t)stride * 4, (size_t)len * 4, (size_t)n,
                          memspace == MRCNN_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return tmp.as<float>();
}

// Writes n dense device rows of `len` floats to a destination with row stride `stride`.
void unstage_rows(const float* dev, void* dst, int memspace, long n, long len, long stride)
{
    if (n <= 0) return;
    HIP_CHECK(hipMemcpy2D(dst, (size_t)stride * 4, dev, (size_t)len * 4, (size_t)len * 4, (size_t)n,
                          memspace == MRCNN_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
}

struct Params {
    std::map<std::string, mrcnn_param> m;
    Params(const mrcnn_param* p, int n)
    {
        for (int i = 0; i < n; ++i)
            if (p[i].key) m[p[i].key] = p[i];
    }
    // `parameters["x"] as? Int` / `as? Double` (ProposalLayer.swift:70-90): wrong type → ignored.
    bool get_int(const char* k, int64_t& v) const
    {
        auto it = m.find(k);
        if (it == m.end() || it->second.type != MRCNN_PARAM_INT) return false;
        v = it->second.i;
        return true;
    }
    bool get_double(const char* k, double& v) const
    {
        auto it = m.find(k);
        if (it == m.end() || it->second.type != MRCNN_PARAM_DOUBLE) return false;
        v = it->second.d;
        return true;
    }
    void std_dev(float out[4]) const   // ProposalLayer.swift:70-80 / DetectionLayer.swift:67-77
    {
        int64_t cnt;
        if (!get_int("bboxStdDev_count", cnt) || cnt != 4) return;
        float tmp[4];
        for (int i = 0; i < 4; ++i) {
            double d;
            if (!get_double(("bboxStdDev_" + std::to_string(i)).c_str(), d)) return;
            tmp[i] = (float)d;
        }
        memcpy(out, tmp, sizeof tmp);
    }
};

struct Stream {
    hipStream_t s = nullptr;
    Stream() { require_gpu(); HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
};

// sub-model cache (deliberately NOT re-loading per evaluate, unlike TimeDistributedClassifierLayer.swift:41)
std::mutex g_model_mu;
std::map<std::string, std::shared_ptr<Model>> g_models;
std::shared_ptr<Model> cached_model(int kind, const std::string& path, int min_cap)
{
    std::lock_guard<std::mutex> lk(g_model_mu);
    const std::string key = std::to_string(kind) + ":" + path;
    auto it = g_models.find(key);
    if (it != g_models.end() && it->second->max_batch >= min_cap) return it->second;
    auto m = std::make_shared<Model>();
    m->load(kind, path, min_cap, MRCNN_F32);
    g_models[key] = m;
    return m;
}

}  // namespace

// ================================================================================================
// custom layers
// ================================================================================================
struct mrcnn_layer {
    virtual ~mrcnn_layer() {}
    virtual void output_shapes(const int64_t (*in)[5], int n_in, int64_t (*out)[5], int* n_out) = 0;
    virtual void evaluate(const mrcnn_tensor* in, int n_in, mrcnn_tensor* out, int n_out) = 0;
};

namespace {

// ---- ProposalLayer (ProposalLayer.swift:52-197) ------------------------------------------------
struct ProposalLayerImpl : mrcnn_layer {
    float std4[4] = {0.1f, 0.1f, 0.2f, 0.2f};
    int pre_nms = 6000, max_prop = 1000;
    float nms_thr = 0.7f;
    DevBuf anchors;
    long n_anchors = 0;
    DevBuf ws_buf, out_buf;
    ProposalWorkspace ws;
    Stream st;

    explicit ProposalLayerImpl(const Params& p)
    {
        require_gpu();
        const char* ap = mrcnn_config_get_anchors_path();   // ProposalLayer.swift:68 force-unwraps the URL
        MRCNN_REQUIRE(ap, MRCNN_ERR_CONFIG, "MaskRCNNConfig.anchorsURL is not set (required by ProposalLayer.init)");
        FILE* f = fopen(ap, "rb");
        MRCNN_REQUIRE(f, MRCNN_ERR_IO, "cannot open anchors file '%s'", ap);
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz <= 0 || sz % 16 != 0) { fclose(f); fail(MRCNN_ERR_IO, "anchors file '%s': size %ld is not a positive multiple of 16", ap, sz); }
        std::vector<float> h((size_t)sz / 4);
        const size_t got = fread(h.data(), 1, (size_t)sz, f);
        fclose(f);
        MRCNN_REQUIRE(got == (size_t)sz, MRCNN_ERR_IO, "short read on '%s'", ap);
        n_anchors = sz / 16;
        anchors.alloc((size_t)sz);
        HIP_CHECK(hipMemcpy(anchors.p, h.data(), (size_t)sz, hipMemcpyHostToDevice));
        p.std_dev(std4);
        int64_t v;
        double d;
        if (p.get_int("preNMSMaxProposals", v)) pre_nms = (int)v;
        if (p.get_int("maxProposals", v)) max_prop = (int)v;
        if (p.get_double("nmsIOUThreshold", d)) nms_thr = (float)d;
        MRCNN_REQUIRE(pre_nms >= 1 && max_prop >= 1, MRCNN_ERR_INVALID, "ProposalLayer: preNMSMaxProposals / maxProposals must be >= 1");
    }
    void output_shapes(const int64_t (*in)[5], int n_in, int64_t (*out)[5], int* n_out) override
    {
        MRCNN_REQUIRE(n_in >= 2, MRCNN_ERR_SHAPE, "ProposalLayer expects 2 inputs");
        memcpy(out[0], in[1], sizeof(int64_t) * 5);       // :98-100
        out[0][0] = max_prop;
        *n_out = 1;
    }
    void evaluate(const mrcnn_tensor* in, int n_in, mrcnn_tensor* out, int n_out) override
    {
        MRCNN_REQUIRE(n_in >= 2 && n_out >= 1, MRCNN_ERR_SHAPE, "ProposalLayer expects 2 inputs and 1 output");
        check_f32(in[0], "ProposalLayer probabilities");
        check_f32(in[1], "ProposalLayer deltas");
        check_f32(out[0], "ProposalLayer output");
        const long A = in[0].shape[0];                    // :119
        MRCNN_REQUIRE(A >= 1 && A <= n_anchors, MRCNN_ERR_SHAPE, "ProposalLayer: %ld regions but anchors.bin holds %ld", A, n_anchors);
        MRCNN_REQUIRE(A < (1L << 24) / 4, MRCNN_ERR_UNSUPPORTED, "ProposalLayer: region count exceeds the reference's float-index range");
        const int K = (int)(A < pre_nms ? A : pre_nms);   // :120
        if (ws.A != (int)A || ws.K != K) {
            ws_buf.alloc(ProposalWorkspace::bytes(1, (int)A, K, max_prop));
            ws.bind(ws_buf.p, 1, (int)A, K, max_prop);
        }
        DevBuf t0, t1;
        const float* probs = stage_rows(in[0].data, in[0].memspace, A, 2, 2, t0);
        const float* deltas = stage_rows(in[1].data, in[1].memspace, A, 4, 4, t1);
        const long ostride = out[0].strides[0];           // :179
        MRCNN_REQUIRE(ostride >= 4, MRCNN_ERR_SHAPE, "ProposalLayer: output row stride %ld < 4", ostride);
        if (out[0].memspace == MRCNN_DEVICE) {
            proposal_forward(st.s, ws, probs, 0, deltas, 0, anchors.as<float>(), std4, nms_thr, (float*)out[0].data, 0, ostride);
            HIP_CHECK(hipStreamSynchronize(st.s));
        } else {
            // kept rows receive 4 coordinates only; stage the caller's rows in so the rest is preserved
            out_buf.alloc((size_t)max_prop * ostride * 4);
            HIP_CHECK(hipMemcpy(out_buf.p, out[0].data, (size_t)max_prop * ostride * 4, hipMemcpyHostToDevice));
            proposal_forward(st.s, ws, probs, 0, deltas, 0, anchors.as<float>(), std4, nms_thr, out_buf.as<float>(), 0, ostride);
            HIP_CHECK(hipStreamSynchronize(st.s));
            HIP_CHECK(hipMemcpy(out[0].data, out_buf.p, (size_t)max_prop * ostride * 4, hipMemcpyDeviceToHost));
        }
    }
};

// ---- PyramidROIAlignLayer (PyramidROIAlignLayer.swift:40-183) -----------------------------------
struct PyramidLayerImpl : mrcnn_layer {
    int pool = 7;
    double img_w = 1024, img_h = 1024;
    Stream st;
    explicit PyramidLayerImpl(const Params& p)
    {
        require_gpu();
        int64_t v, w, h;
        if (p.get_int("poolSize", v)) pool = (int)v;
        // :55-58 casts `as? CGFloat`; the converter writes intValue (task.py:41-42).  Accept both.
        double dw, dh;
        if (p.get_int("imageWidth", w) && p.get_int("imageHeight", h)) { img_w = (double)w; img_h = (double)h; }
        else if (p.get_double("imageWidth", dw) && p.get_double("imageHeight", dh)) { img_w = dw; img_h = dh; }
        MRCNN_REQUIRE(pool >= 1, MRCNN_ERR_INVALID, "PyramidROIAlignLayer: poolSize must be >= 1");
    }
    void output_shapes(const int64_t (*in)[5], int n_in, int64_t (*out)[5], int* n_out) override
    {
        MRCNN_REQUIRE(n_in >= 2, MRCNN_ERR_SHAPE, "PyramidROIAlignLayer expects rois + feature maps");
        out[0][0] = in[0][0]; out[0][1] = in[0][1]; out[0][2] = in[1][2]; out[0][3] = pool; out[0][4] = pool;   // :67-76
        *n_out = 1;
    }
    void evaluate(const mrcnn_tensor* in, int n_in, mrcnn_tensor* out, int n_out) override
    {
        MRCNN_REQUIRE(n_in == 5 && n_out >= 1, MRCNN_ERR_SHAPE, "PyramidROIAlignLayer expects rois + 4 feature maps");
        for (int i = 0; i < 5; ++i) check_f32(in[i], "PyramidROIAlignLayer input");
        check_f32(out[0], "PyramidROIAlignLayer output");
        const long n = in[0].shape[0], rstride = in[0].strides[0];
        const int C = (int)in[1].shape[2];
        DevBuf tr, tm[4], to;
        const float* rois = stage_rows(in[0].data, in[0].memspace, n, 4, rstride, tr);
        const long roi_stride = (in[0].memspace == MRCNN_DEVICE && rstride == 4) ? 4 : 4;
        PyramidMaps maps;
        for (int l = 0; l < 4; ++l) {
            const mrcnn_tensor& m = in[1 + l];
            MRCNN_REQUIRE(m.shape[2] == C, MRCNN_ERR_SHAPE, "feature maps disagree on channel count");
            maps.H[l] = (int)m.shape[3]; maps.W[l] = (int)m.shape[4];
            const long len = (long)C * maps.H[l] * maps.W[l];
            maps.data[l] = stage_rows(m.data, m.memspace, 1, len, len, tm[l]);
            maps.sB[l] = len;
        }
        const long row = (long)C * pool * pool, ostride = out[0].strides[0];
        MRCNN_REQUIRE(ostride >= row, MRCNN_ERR_SHAPE, "PyramidROIAlignLayer: output row stride too small");
        if (out[0].memspace == MRCNN_DEVICE) {
            roi_align_forward(st.s, maps, C, 0, rois, 0, roi_stride, (int)n, 1, pool, img_w, img_h, (float*)out[0].data, 0, ostride, MRCNN_F32);
            HIP_CHECK(hipStreamSynchronize(st.s));
        } else {
            to.alloc((size_t)(n > 0 ? n : 1) * row * 4);
            roi_align_forward(st.s, maps, C, 0, rois, 0, roi_stride, (int)n, 1, pool, img_w, img_h, to.as<float>(), 0, row, MRCNN_F32);
            HIP_CHECK(hipStreamSynchronize(st.s));
            unstage_rows(to.as<float>(), out[0].data, MRCNN_HOST, n, row, ostride);
        }
    }
};

// ---- TimeDistributedClassifierLayer (TimeDistributedClassifierLayer.swift:14-92) -----------------
struct ClassifierLayerImpl : mrcnn_layer {
    Stream st;
    explicit ClassifierLayerImpl(const Params&) { require_gpu(); }
    void output_shapes(const int64_t (*in)[5], int n_in, int64_t (*out)[5], int* n_out) override
    {
        MRCNN_REQUIRE(n_in >= 1, MRCNN_ERR_SHAPE, "TimeDistributedClassifierLayer expects 1 input");
        out[0][0] = in[0][0]; out[0][1] = in[0][1]; out[0][2] = 1; out[0][3] = 1; out[0][4] = 6;   // :27-31
        *n_out = 1;
    }
    void evaluate(const mrcnn_tensor* in, int n_in, mrcnn_tensor* out, int n_out) override
    {
        MRCNN_REQUIRE(n_in >= 1 && n_out >= 1, MRCNN_ERR_SHAPE, "TimeDistributedClassifierLayer expects 1 input and 1 output");
        check_f32(in[0], "TimeDistributedClassifierLayer input");
        check_f32(out[0], "TimeDistributedClassifierLayer output");
        const char* cp = mrcnn_config_get_classifier_path();
        MRCNN_REQUIRE(cp, MRCNN_ERR_CONFIG, "MaskRCNNConfig.compiledClassifierModelURL is not set");
        const long n = in[0].shape[0];
        auto model = cached_model(MRCNN_MODEL_CLASSIFIER, cp, (int)(n > 1000 ? n : 1000));
        ClassifierHead& hd = model->cls_head;
        const int C = (int)in[0].shape[2], ph = (int)in[0].shape[3], pw = (int)in[0].shape[4];
        MRCNN_REQUIRE(C == hd.C && ph == hd.pool && pw == hd.pool, MRCNN_ERR_SHAPE, "feature_map is %dx%dx%d, Classifier expects %dx%dx%d", C, ph,
                      pw, hd.C, hd.pool, hd.pool);
        const long row = (long)C * ph * pw;
        DevBuf ti;
        const float* chw = stage_rows(in[0].data, in[0].memspace, n, row, in[0].strides[0], ti);
        std::lock_guard<std::mutex> lk(model->eval_mu);   // the head's scratch is shared by every layer instance
        nchw_to_nhwc_forward(st.s, chw, n, C, ph, pw, hd.stage_in, hd.dtype);
        conv_set_scratch(model->conv_scratch.ks_buf.p ? &model->conv_scratch : nullptr);      // (the mutex above serialises its users; the call synchronises below)
        try { hd.forward(st.s, hd.stage_in, (int)n, hd.cls6, 6); } catch (...) { conv_set_scratch(nullptr); throw; }
        conv_set_scratch(nullptr);
        HIP_CHECK(hipStreamSynchronize(st.s));
        const long ostride = out[0].strides[2];           // :63
        MRCNN_REQUIRE(ostride >= 6, MRCNN_ERR_SHAPE, "TimeDistributedClassifierLayer: output stride %ld < 6", ostride);
        unstage_rows(hd.cls6, out[0].data, out[0].memspace, n, 6, ostride);
    }
};

// ---- DetectionLayer (DetectionLayer.swift:52-236) -----------------------------------------------
struct DetectionLayerImpl : mrcnn_layer {
    float std4[4] = {0.1f, 0.1f, 0.2f, 0.2f};
    int max_det = 100;
    float score_thr = 0.7f, nms_thr = 0.3f;
    DevBuf ws_buf, out_buf;
    DetectionWorkspace ws;
    Stream st;
    explicit DetectionLayerImpl(const Params& p)
    {
        require_gpu();
        p.std_dev(std4);
        int64_t v;
        double d;
        if (p.get_int("maxDetections", v)) max_det = (int)v;
        if (p.get_double("scoreThreshold", d)) score_thr = (float)d;
        if (p.get_double("nmsIOUThreshold", d)) nms_thr = (float)d;
        MRCNN_REQUIRE(max_det >= 1, MRCNN_ERR_INVALID, "DetectionLayer: maxDetections must be >= 1");
    }
    void output_shapes(const int64_t (*in)[5], int n_in, int64_t (*out)[5], int* n_out) override
    {
        MRCNN_REQUIRE(n_in >= 1, MRCNN_ERR_SHAPE, "DetectionLayer expects 2 inputs");
        out[0][0] = max_det; out[0][1] = in[0][1]; out[0][2] = 6; out[0][3] = 1; out[0][4] = 1;   // :96-104
        *n_out = 1;
    }
    void evaluate(const mrcnn_tensor* in, int n_in, mrcnn_tensor* out, int n_out) override
    {
        MRCNN_REQUIRE(n_in >= 2 && n_out >= 1, MRCNN_ERR_SHAPE, "DetectionLayer expects 2 inputs and 1 output");
        check_f32(in[0], "DetectionLayer rois");
        check_f32(in[1], "DetectionLayer classifications");
        check_f32(out[0], "DetectionLayer output");
        const long n = in[0].shape[0];                    // :122
        MRCNN_REQUIRE(n >= 1, MRCNN_ERR_SHAPE, "DetectionLayer: no regions");
        if (ws.N != (int)n) {
            ws_buf.alloc(DetectionWorkspace::bytes(1, (int)n, max_det));
            ws.bind(ws_buf.p, 1, (int)n, max_det);
        }
        DevBuf t0, t1;
        // rois are read through floatDataPointer + broadcast indices, i.e. as contiguous (n,4) (:144-145)
        const float* rois = stage_rows(in[0].data, in[0].memspace, n, 4, 4, t0);
        const float* cls = stage_rows(in[1].data, in[1].memspace, n, 6, 6, t1);     // stride 6 assumed (:125-128)
        const long ostride = out[0].strides[0];           // :213
        MRCNN_REQUIRE(ostride >= 6, MRCNN_ERR_SHAPE, "DetectionLayer: output row stride %ld < 6", ostride);
        if (out[0].memspace == MRCNN_DEVICE) {
            detection_forward(st.s, ws, rois, 0, 4, cls, 0, std4, score_thr, nms_thr, 0, (float*)out[0].data, 0, ostride);
            HIP_CHECK(hipStreamSynchronize(st.s));
        } else {
            out_buf.alloc((size_t)max_det * ostride * 4);
            HIP_CHECK(hipMemcpy(out_buf.p, out[0].data, (size_t)max_det * ostride * 4, hipMemcpyHostToDevice));
            detection_forward(st.s, ws, rois, 0, 4, cls, 0, std4, score_thr, nms_thr, 0, out_buf.as<float>(), 0, ostride);
            HIP_CHECK(hipStreamSynchronize(st.s));
            HIP_CHECK(hipMemcpy(out[0].data, out_buf.p, (size_t)max_det * ostride * 4, hipMemcpyDeviceToHost));
        }
    }
};

// ---- TimeDistributedMaskLayer (TimeDistributedMaskLayer.swift:14-92) -----------------------------
struct MaskLayerImpl : mrcnn_layer {
    Stream st;
    DevBuf ws_buf, nchw, out_buf;
    explicit MaskLayerImpl(const Params&) { require_gpu(); }
    void output_shapes(const int64_t (*in)[5], int n_in, int64_t (*out)[5], int* n_out) override
    {
        MRCNN_REQUIRE(n_in >= 1, MRCNN_ERR_SHAPE, "TimeDistributedMaskLayer expects 2 inputs");
        out[0][0] = 1; out[0][1] = in[0][1]; out[0][2] = in[0][0]; out[0][3] = in[0][3] * 2; out[0][4] = in[0][4] * 2;   // :27-36
        *n_out = 1;
    }
    void evaluate(const mrcnn_tensor* in, int n_in, mrcnn_tensor* out, int n_out) override
    {
        MRCNN_REQUIRE(n_in >= 2 && n_out >= 1, MRCNN_ERR_SHAPE, "TimeDistributedMaskLayer expects 2 inputs and 1 output");
        check_f32(in[0], "TimeDistributedMaskLayer feature maps");
        check_f32(in[1], "TimeDistributedMaskLayer detections");
        check_f32(out[0], "TimeDistributedMaskLayer output");
        const char* mp = mrcnn_config_get_mask_path();
        MRCNN_REQUIRE(mp, MRCNN_ERR_CONFIG, "MaskRCNNConfig.compiledMaskModelURL is not set");
        const long D = in[0].shape[0];
        const long det_count = in[1].shape[0], det_stride = in[1].strides[0];     // :46-47
        MRCNN_REQUIRE(D >= 1 && det_count >= D, MRCNN_ERR_SHAPE, "TimeDistributedMaskLayer: %ld feature rows, %ld detections", D, det_count);
        auto model = cached_model(MRCNN_MODEL_MASK, mp, (int)(D > 100 ? D : 100));
        MaskHead& hd = model->mask_head;
        const int C = (int)in[0].shape[2], ph = (int)in[0].shape[3], pw = (int)in[0].shape[4];
        MRCNN_REQUIRE(C == hd.C && ph == hd.pool && pw == hd.pool, MRCNN_ERR_SHAPE, "feature_map is %dx%dx%d, Mask expects %dx%dx%d", C, ph, pw,
                      hd.C, hd.pool, hd.pool);
        const long row = (long)C * ph * pw;
        const int HW = 4 * ph * pw;
        DevBuf ti, td;
        const float* chw = stage_rows(in[0].data, in[0].memspace, D, row, in[0].strides[0], ti);
        const float* det = stage_rows(in[1].data, in[1].memspace, det_count, det_stride, det_stride, td);
        ws_buf.alloc((size_t)(2 * det_count + 1) * 4 + 1024);
        MaskSelectWorkspace ws;
        ws.flags = ws_buf.as<int32_t>();
        ws.mapping = ws.flags + det_count;
        ws.kept = ws.mapping + det_count;
        std::lock_guard<std::mutex> lk(model->eval_mu);   // the head's scratch is shared by every layer instance
        mask_valid_rows_forward(st.s, chw, 0, row, row, (int)D, 1, ws, MRCNN_F32);          // removeZeros:true (:52)
        nchw_to_nhwc_forward(st.s, chw, D, C, ph, pw, hd.stage_in, hd.dtype);
        hd.forward_features(st.s, hd.stage_in, (int)D);
        hd.forward_full(st.s, (int)D);
        nchw.alloc((size_t)D * hd.nc * HW * 4);
        nhwc_to_nchw_forward(st.s, hd.full, D, hd.nc, 2 * ph, 2 * pw, nchw.as<float>());
        const long ostride = out[0].strides[2];           // :56
        MRCNN_REQUIRE(ostride >= HW, MRCNN_ERR_SHAPE, "TimeDistributedMaskLayer: output stride %ld < %d", ostride, HW);
        // `flags`/`mapping` are sized by D rows; padding (:87-89) runs to detectionCount rows.
        float* o = (float*)out[0].data;
        if (out[0].memspace != MRCNN_DEVICE) {
            out_buf.alloc((size_t)det_count * ostride * 4);
            HIP_CHECK(hipMemcpy(out_buf.p, out[0].data, (size_t)det_count * ostride * 4, hipMemcpyHostToDevice));
            o = out_buf.as<float>();
        }
        mask_select_from_full_forward(st.s, nchw.as<float>(), 0, HW, hd.nc, det, 0, det_stride, (int)det_count, 1, ws, o, 0, ostride);
        HIP_CHECK(hipStreamSynchronize(st.s));
        if (out[0].memspace != MRCNN_DEVICE)
            HIP_CHECK(hipMemcpy(out[0].data, out_buf.p, (size_t)det_count * ostride * 4, hipMemcpyDeviceToHost));
    }
};

}  // namespace

extern "C" int mrcnn_layer_create(const char* class_name, const mrcnn_param* params, int n_params, mrcnn_layer** out_layer)
{
    return guarded([&] {
        MRCNN_REQUIRE(class_name && out_layer, MRCNN_ERR_INVALID, "null argument");
        MRCNN_REQUIRE(n_params == 0 || params, MRCNN_ERR_INVALID, "null parameter array");
        const Params p(params, n_params);
        const std::string c = class_name;
        if (c == "ProposalLayer") *out_layer = new ProposalLayerImpl(p);
        else if (c == "PyramidROIAlignLayer") *out_layer = new PyramidLayerImpl(p);
        else if (c == "TimeDistributedClassifierLayer") *out_layer = new ClassifierLayerImpl(p);
        else if (c == "DetectionLayer") *out_layer = new DetectionLayerImpl(p);
        else if (c == "TimeDistributedMaskLayer") *out_layer = new MaskLayerImpl(p);
        else fail(MRCNN_ERR_INVALID, "unknown custom layer class '%s'", class_name);
    });
}
extern "C" int mrcnn_layer_set_weight_data(mrcnn_layer* layer, const void* const*, const size_t*, int)
{
    return guarded([&] { MRCNN_REQUIRE(layer, MRCNN_ERR_INVALID, "null layer"); });   // no-op (ProposalLayer.swift:93-95)
}
extern "C" int mrcnn_layer_output_shapes(mrcnn_layer* layer, const int64_t (*in_shapes)[5], int n_in, int64_t (*out_shapes)[5], int* n_out)
{
    return guarded([&] {
        MRCNN_REQUIRE(layer && in_shapes && out_shapes && n_out, MRCNN_ERR_INVALID, "null argument");
        layer->output_shapes(in_shapes, n_in, out_shapes, n_out);
    });
}
extern "C" int mrcnn_layer_evaluate(mrcnn_layer* layer, const mrcnn_tensor* inputs, int n_in, mrcnn_tensor* outputs, int n_out)
{
    return guarded([&] {
        MRCNN_REQUIRE(layer && inputs && outputs, MRCNN_ERR_INVALID, "null argument");
        layer->evaluate(inputs, n_in, outputs, n_out);
    });
}
extern "C" void mrcnn_layer_destroy(mrcnn_layer* layer) { delete layer; }

// IOU (Utils.swift:232-246) — host, as in the reference.
extern "C" float mrcnn_iou(const float a[4], const float b[4])
{
    struct R { double x, y, w, h; };
    auto mk = [](const float* d) { R r{(double)d[1], (double)d[0], (double)d[3] - (double)d[1], (double)d[2] - (double)d[0]}; return r; };
    const R A = mk(a), B = mk(b);
    const double areaA = fabs(A.w) * fabs(A.h);
    if (areaA <= 0) return 0;
    const double areaB = fabs(B.w) * fabs(B.h);
    if (areaB <= 0) return 0;
    auto minx = [](const R& r) { return r.w < 0 ? r.x + r.w : r.x; };
    auto maxx = [](const R& r) { return r.w < 0 ? r.x : r.x + r.w; };
    auto miny = [](const R& r) { return r.h < 0 ? r.y + r.h : r.y; };
    auto maxy = [](const R& r) { return r.h < 0 ? r.y : r.y + r.h; };
    const double ix0 = fmax(minx(A), minx(B)), iy0 = fmax(miny(A), miny(B));
    const double ix1 = fmin(maxx(A), maxx(B)), iy1 = fmin(maxy(A), maxy(B));
    const double inter = fmax(iy1 - iy0, 0.0) * fmax(ix1 - ix0, 0.0);
    return (float)(inter / (areaA + areaB - inter));
}

// ================================================================================================
// models
// ================================================================================================
extern "C" int mrcnn_model_load(int kind, const char* path, int max_batch, int compute_dtype, mrcnn_model** out_model)
{
    return guarded([&] {
        MRCNN_REQUIRE(path && out_model, MRCNN_ERR_INVALID, "null argument");
        MRCNN_REQUIRE(kind >= 0 && kind <= 2, MRCNN_ERR_INVALID, "unknown model kind %d", kind);
        MRCNN_REQUIRE(compute_dtype == MRCNN_DEFAULT || compute_dtype == MRCNN_F32 || compute_dtype == MRCNN_F16 || compute_dtype == MRCNN_F32S || compute_dtype == MRCNN_F32X3,
                      MRCNN_ERR_UNSUPPORTED, "compute dtype %d not available (MRCNN_DEFAULT, MRCNN_F32, MRCNN_F16, MRCNN_F32S or MRCNN_F32X3)", compute_dtype);
        std::unique_ptr<mrcnn_model> h(new mrcnn_model);
        h->m.load(kind, path, max_batch, compute_dtype);
        *out_model = h.release();
    });
}
extern "C" void mrcnn_model_destroy(mrcnn_model* model) { delete model; }

extern "C" int mrcnn_model_set_stream(mrcnn_model* model, void* hip_stream)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        if (model->m.own_stream && model->m.stream) (void)hipStreamDestroy(model->m.stream);
        model->m.stream = (hipStream_t)hip_stream;
        model->m.own_stream = false;
    });
}

extern "C" int mrcnn_model_enable_graph(mrcnn_model* model, int on)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        model->m.use_graph = on != 0;
        if (!on) {
            HIP_CHECK(hipStreamSynchronize(model->m.stream));
            model->m.drop_graphs();
        }
    });
}

extern "C" int mrcnn_maskrcnn_predict(mrcnn_model* model, const uint8_t* rgb, int batch, int height, int width, int memspace,
                                      float* detections, float* masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        model->m.predict(rgb, batch, height, width, memspace, detections, masks, true);
    });
}
extern "C" int mrcnn_maskrcnn_predict_scalefit(mrcnn_model* model, const uint8_t* rgb, int batch, int height, int width, int memspace,
                                               float* detections, float* masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        model->m.predict(rgb, batch, height, width, memspace, detections, masks, true, true);
    });
}
extern "C" int mrcnn_unletterbox_boxes(float* detections, int64_t n, int64_t stride, int h, int w, int H, int W)
{
    return guarded([&] {
        MRCNN_REQUIRE(detections && n >= 0 && stride >= 4, MRCNN_ERR_INVALID, "bad unletterbox argument");
        int nh, nw, py, px;
        MRCNN_REQUIRE(mrcnn_letterbox_geometry(h, w, H, W, &nh, &nw, &py, &px) == MRCNN_OK, MRCNN_ERR_INVALID, "bad letterbox geometry");
        // normalized coordinates follow Matterport's norm_boxes: pixel = n * (size - 1), far edge + 1 (anchors, mask paste)
        const double sy = (double)h / nh, sx = (double)w / nw, hy = h > 1 ? h - 1 : 1, wx = w > 1 ? w - 1 : 1;
        auto clip = [](double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); };
        for (int64_t i = 0; i < n; ++i) {
            float* d = detections + i * stride;
            // zero-padded rows (DetectionLayer.swift:224-230 pads the output to maxDetections rows) stay all-zero: mapped through
            // the letterbox an all-zero box would come out with a non-zero far edge and read as a detection
            if (d[0] == 0.f && d[1] == 0.f && d[2] == 0.f && d[3] == 0.f) continue;
            const double y1 = ((double)d[0] * (H - 1) - py) * sy, x1 = ((double)d[1] * (W - 1) - px) * sx;
            const double y2 = ((double)d[2] * (H - 1) + 1.0 - py) * sy, x2 = ((double)d[3] * (W - 1) + 1.0 - px) * sx;
            d[0] = (float)clip(y1 / hy); d[1] = (float)clip(x1 / wx); d[2] = (float)clip((y2 - 1.0) / hy); d[3] = (float)clip((x2 - 1.0) / wx);
        }
    });
}
extern "C" int mrcnn_maskrcnn_submit(mrcnn_model* model, const uint8_t* rgb_host, int batch, int height, int width)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        model->m.submit(rgb_host, batch, height, width);
    });
}
extern "C" int mrcnn_maskrcnn_collect(mrcnn_model* model, float* detections, float* masks, int* batch)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        model->m.collect(detections, masks, batch);
    });
}
extern "C" int mrcnn_maskrcnn_predict_async(mrcnn_model* model, const uint8_t* rgb, int batch, int height, int width,
                                            float* detections, float* masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        model->m.predict(rgb, batch, height, width, MRCNN_DEVICE, detections, masks, false);
    });
}

extern "C" int mrcnn_classifier_predict(mrcnn_model* model, const float* feature_map, int n, int memspace, float* probabilities,
                                        float* bounding_boxes)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && model->m.kind == MRCNN_MODEL_CLASSIFIER, MRCNN_ERR_INVALID, "not a Classifier model");
        MRCNN_REQUIRE(feature_map && probabilities && bounding_boxes && n >= 0, MRCNN_ERR_INVALID, "bad argument");
        ClassifierHead& hd = model->m.cls_head;
        hipStream_t s = model->m.stream;
        const long row = (long)hd.C * hd.pool * hd.pool;
        const hipMemcpyKind back = memspace == MRCNN_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        for (int i0 = 0; i0 < n; i0 += hd.cap) {
            const int c = n - i0 < hd.cap ? n - i0 : hd.cap;
            DevBuf ti;
            const float* chw = stage_rows(feature_map + (size_t)i0 * row, memspace, c, row, row, ti);
            nchw_to_nhwc_forward(s, chw, c, hd.C, hd.pool, hd.pool, hd.stage_in, hd.dtype);
            conv_set_scratch(model->m.conv_scratch.ks_buf.p ? &model->m.conv_scratch : nullptr);
            try { hd.forward(s, hd.stage_in, c, nullptr, 0); } catch (...) { conv_set_scratch(nullptr); throw; }
            conv_set_scratch(nullptr);
            HIP_CHECK(hipMemcpyAsync(probabilities + (size_t)i0 * hd.nc, hd.probs, (size_t)c * hd.nc * 4, back, s));
            HIP_CHECK(hipMemcpyAsync(bounding_boxes + (size_t)i0 * hd.nc * 4, hd.bbox, (size_t)c * hd.nc * 16, back, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
    });
}

extern "C" int mrcnn_mask_predict(mrcnn_model* model, const float* feature_map, int n, int memspace, float* masks)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && model->m.kind == MRCNN_MODEL_MASK, MRCNN_ERR_INVALID, "not a Mask model");
        MRCNN_REQUIRE(feature_map && masks && n >= 0, MRCNN_ERR_INVALID, "bad argument");
        MaskHead& hd = model->m.mask_head;
        hipStream_t s = model->m.stream;
        const long row = (long)hd.C * hd.pool * hd.pool;
        const long orow = (long)hd.nc * 4 * hd.pool * hd.pool;
        DevBuf tmp;
        for (int i0 = 0; i0 < n; i0 += hd.cap) {
            const int c = n - i0 < hd.cap ? n - i0 : hd.cap;
            DevBuf ti;
            const float* chw = stage_rows(feature_map + (size_t)i0 * row, memspace, c, row, row, ti);
            nchw_to_nhwc_forward(s, chw, c, hd.C, hd.pool, hd.pool, hd.stage_in, hd.dtype);
            hd.forward_features(s, hd.stage_in, c);
            hd.forward_full(s, c);
            float* dst = masks + (size_t)i0 * orow;
            if (memspace != MRCNN_DEVICE) { tmp.alloc((size_t)c * orow * 4); dst = tmp.as<float>(); }
            nhwc_to_nchw_forward(s, hd.full, c, hd.nc, 2 * hd.pool, 2 * hd.pool, dst);
            HIP_CHECK(hipStreamSynchronize(s));
            if (memspace != MRCNN_DEVICE) HIP_CHECK(hipMemcpy(masks + (size_t)i0 * orow, tmp.p, (size_t)c * orow * 4, hipMemcpyDeviceToHost));
        }
    });
}

extern "C" int mrcnn_model_get_int(mrcnn_model* model, const char* key, int64_t* value)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && key && value, MRCNN_ERR_INVALID, "null argument");
        const Model& m = model->m;
        const std::string k = key;
        if (k == "num_classes") *value = m.nc;
        else if (k == "max_batch") *value = m.max_batch;
        else if (k == "compute_dtype") *value = m.mode;
        else if (k == "compute_dtype_defaulted") *value = m.mode_defaulted ? 1 : 0;
        else if (m.kind != MRCNN_MODEL_MASKRCNN) *value = m.file.get_int(k);
        else if (k == "image_height") *value = m.H;
        else if (k == "image_width") *value = m.W;
        else if (k == "max_proposals") *value = m.max_prop;
        else if (k == "max_detections") *value = m.max_det;
        else if (k == "num_anchors") *value = m.A;
        else if (k == "pre_nms_max_proposals") *value = m.pre_nms;
        else if (k == "pre_nms_count") *value = m.K;
        else if (k == "mask_size") *value = 2 * m.mask_pool;
        else if (k == "range_overflows") *value = m.range_overflows;
        else if (k == "range_recoveries") *value = m.range_recoveries;
        else if (k == "split_exponents_from_artefact") *value = m.exponents_from_artefact ? 1 : 0;
        // scale-aware split (mrcnn_model_calibrate_split): totals over the tensor groups of the last calibrate / diagnose pass
        else if (k == "split_groups") *value = (int64_t)m.sgroups.size();
        else if (k == "split_calibrated") *value = m.split_calibrated ? 1 : 0;
        else if (k == "split_small_inputs" || k == "split_inexact_inputs" || k == "split_inputs_counted" || k == "split_min_exponent" ||
                 k == "split_max_exponent") {
            long long small = 0, inexact = 0, counted = 0;
            int lo = 0, hi = 0;
            bool first = true;
            for (const auto& g : m.sgroups) {
                if (g.fixed) continue;
                small += g.small; inexact += g.inexact; counted += g.counted;
                if (first || g.exp < lo) lo = g.exp;
                if (first || g.exp > hi) hi = g.exp;
                first = false;
            }
            *value = k == "split_small_inputs" ? small : k == "split_inexact_inputs" ? inexact : k == "split_inputs_counted" ? counted :
                     k == "split_min_exponent" ? lo : hi;
        }
        else if (k == "graph_launches") *value = m.graph_launches;
        else if (k == "gpu_busy_us") *value = (int64_t)(m.gpu_busy_ms * 1e3);
        else if (k == "predict_calls") *value = m.predict_calls;
        else if (k == "graph_enabled") *value = m.use_graph ? 1 : 0;
        else *value = m.file.get_int(k);
    });
}

extern "C" int mrcnn_model_calibrate_split(mrcnn_model* model, const uint8_t* rgb, int batch, int height, int width, int memspace, int apply)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && rgb, MRCNN_ERR_INVALID, "bad calibrate_split argument");
        model->m.calibrate_split(rgb, batch, height, width, memspace, apply != 0);
    });
}
extern "C" int mrcnn_model_split_group_stat(mrcnn_model* model, int index, mrcnn_split_group_stat* out)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && out && index >= 0 && index < (int)model->m.sgroups.size(), MRCNN_ERR_INVALID, "bad split_group_stat argument");
        const SplitGroup& g = model->m.sgroups[(size_t)index];
        memset(out, 0, sizeof *out);
        snprintf(out->name, sizeof out->name, "%s", g.name.c_str());
        out->exponent = g.exp; out->fixed = g.fixed ? 1 : 0; out->absmax = g.absmax;
        out->small_inputs = g.small; out->inexact_inputs = g.inexact; out->inputs_counted = g.counted;
    });
}
extern "C" int mrcnn_model_get_split_exponents(mrcnn_model* model, int32_t* exponents, int capacity, int* count)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && count, MRCNN_ERR_INVALID, "bad get_split_exponents argument");
        const int n = (int)model->m.sgroups.size();
        *count = n;
        if (!exponents) return;
        MRCNN_REQUIRE(capacity >= n, MRCNN_ERR_SHAPE, "get_split_exponents: %d groups, buffer holds %d", n, capacity);
        for (int i = 0; i < n; ++i) exponents[i] = model->m.sgroups[(size_t)i].exp;
    });
}
extern "C" int mrcnn_model_set_split_exponents(mrcnn_model* model, const int32_t* exponents, int count)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && exponents, MRCNN_ERR_INVALID, "bad set_split_exponents argument");
        Model& m = model->m;
        MRCNN_REQUIRE(m.kind == MRCNN_MODEL_MASKRCNN && count == (int)m.sgroups.size(), MRCNN_ERR_SHAPE, "set_split_exponents: the model has %d groups, %d given",
                      (int)m.sgroups.size(), count);
        MRCNN_REQUIRE(m.mode == MRCNN_F32S || m.mode == MRCNN_F32X3, MRCNN_ERR_UNSUPPORTED, "set_split_exponents: only the split modes use exponents");
        for (int i = 0; i < count; ++i) {
            MRCNN_REQUIRE(exponents[i] >= -60 && exponents[i] <= 60, MRCNN_ERR_INVALID, "exponent %d of group %d out of range", exponents[i], i);
            MRCNN_REQUIRE(!m.sgroups[(size_t)i].fixed || exponents[i] == 0, MRCNN_ERR_INVALID, "group %d ('%s') is consumed by fp32 arithmetic: its exponent is 0",
                          i, m.sgroups[(size_t)i].name.c_str());
        }
        for (int i = 0; i < count; ++i) m.sgroups[(size_t)i].exp = exponents[i];
        m.apply_split_exponents();
        m.split_calibrated = true;
    });
}

extern "C" int mrcnn_model_read_tensor(mrcnn_model* model, const char* name, int image_index, float* host_dst, int64_t capacity,
                                       int64_t* count)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && name, MRCNN_ERR_INVALID, "null argument");
        model->m.read_tensor(name, image_index, host_dst, capacity, count);
    });
}

extern "C" int mrcnn_model_check_range(mrcnn_model* model, int* tripped)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && tripped, MRCNN_ERR_INVALID, "null argument");
        Model& m = model->m;
        *tripped = 0;
        if (m.mode == MRCNN_F32 || !m.range_flag.p) return;           // exact-fp32 mode has no fp16 hand-over to watch
        HIP_CHECK(hipStreamSynchronize(m.stream));
        int t = 0;
        HIP_CHECK(hipMemcpy(&t, m.range_flag.p, sizeof(int), hipMemcpyDeviceToHost));
        if (t) ++m.range_overflows;
        *tripped = t ? 1 : 0;
    });
}

// PyramidROIAlign on the engine's own layout (NHWC maps, fp32 or fp16) with the mask layer's removeZeros predicate
extern "C" int mrcnn_roi_align_nhwc(const void* const maps[4], const int heights[4], const int widths[4], int channels, int dtype,
                                    const float* rois, int64_t roi_stride, int n_rois, int pool, double image_w, double image_h,
                                    int memspace, void* out, int32_t* row_flags)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(maps && heights && widths && rois && out, MRCNN_ERR_INVALID, "null argument");
        MRCNN_REQUIRE(dtype == MRCNN_F32 || dtype == MRCNN_F16, MRCNN_ERR_UNSUPPORTED, "roi_align_nhwc: dtype %d", dtype);
        MRCNN_REQUIRE(n_rois >= 0 && pool >= 1 && channels >= 4 && channels % 4 == 0 && roi_stride >= 4, MRCNN_ERR_INVALID, "bad roi_align_nhwc argument");
        if (n_rois == 0) return;
        const size_t es = dtype == MRCNN_F16 ? 2 : 4;
        Stream st;
        DevBuf tm[4], tr, to, tf;
        PyramidMaps pm;
        for (int l = 0; l < 4; ++l) {
            MRCNN_REQUIRE(maps[l] && heights[l] > 0 && widths[l] > 0, MRCNN_ERR_INVALID, "roi_align_nhwc: level %d missing", l);
            const size_t bytes = (size_t)heights[l] * widths[l] * channels * es;
            const void* p = maps[l];
            if (memspace != MRCNN_DEVICE) {
                tm[l].alloc(bytes);
                HIP_CHECK(hipMemcpy(tm[l].p, maps[l], bytes, hipMemcpyHostToDevice));
                p = tm[l].p;
            }
            pm.data[l] = p; pm.H[l] = heights[l]; pm.W[l] = widths[l]; pm.sB[l] = 0;
        }
        const float* r = stage_rows(rois, memspace, n_rois, roi_stride, roi_stride, tr);
        const long row = (long)pool * pool * channels;
        void* o = out;
        int32_t* f = row_flags;
        if (memspace != MRCNN_DEVICE) {
            to.alloc((size_t)n_rois * row * es);
            o = to.p;
            if (row_flags) { tf.alloc((size_t)n_rois * 4); f = tf.as<int32_t>(); }
        }
        roi_align_forward(st.s, pm, channels, 1, r, 0, roi_stride, n_rois, 1, pool, image_w, image_h, o, 0, row, dtype, f);
        HIP_CHECK(hipStreamSynchronize(st.s));
        if (memspace != MRCNN_DEVICE) {
            HIP_CHECK(hipMemcpy(out, to.p, (size_t)n_rois * row * es, hipMemcpyDeviceToHost));
            if (row_flags) HIP_CHECK(hipMemcpy(row_flags, tf.p, (size_t)n_rois * 4, hipMemcpyDeviceToHost));
        }
    });
}

extern "C" int mrcnn_model_enable_timing(mrcnn_model* model, int on)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        model->m.timer.enabled = on != 0;
    });
}
extern "C" int mrcnn_model_stage_ms(mrcnn_model* model, const char* stage, float* ms)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && stage && ms, MRCNN_ERR_INVALID, "null argument");
        auto it = model->m.timer.ms.find(stage);
        MRCNN_REQUIRE(it != model->m.timer.ms.end(), MRCNN_ERR_INVALID, "no timing for stage '%s' (enable timing and run predict first)", stage);
        *ms = it->second;
    });
}

extern "C" int mrcnn_model_conv_profile_enable(mrcnn_model* model, int on)
{
    return guarded([&] {
        MRCNN_REQUIRE(model, MRCNN_ERR_INVALID, "null model");
        ConvProfile& cp = model->m.conv_profile;
        if (on) {
            cp.reset();                                  // a new measurement window
        } else if (cp.active) {
            HIP_CHECK(hipStreamSynchronize(model->m.stream));
            cp.collect();                                // keep the totals readable after the window closes
        }
        cp.active = on != 0;
    });
}
extern "C" int mrcnn_model_conv_profile_get(mrcnn_model* model, int tile, int64_t* launches, double* total_ms, double* total_flops)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && launches && total_ms && total_flops && tile >= 0 && tile < 9, MRCNN_ERR_INVALID, "bad argument");
        HIP_CHECK(hipStreamSynchronize(model->m.stream));
        model->m.conv_profile.collect();
        const auto& sl = model->m.conv_profile.by_tile[tile];
        *launches = sl.launches; *total_ms = sl.ms; *total_flops = sl.flops;
    });
}

extern "C" int mrcnn_model_conv_profile_bytes(mrcnn_model* model, int tile, double* total_bytes)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && total_bytes && tile >= 0 && tile < 9, MRCNN_ERR_INVALID, "bad argument");
        HIP_CHECK(hipStreamSynchronize(model->m.stream));
        model->m.conv_profile.collect();
        *total_bytes = model->m.conv_profile.by_tile[tile].bytes;
    });
}

extern "C" int mrcnn_model_conv_profile_group(mrcnn_model* model, int group, int64_t* launches, double* total_ms, double* total_flops)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && launches && total_ms && total_flops && (group == 0 || group == 1), MRCNN_ERR_INVALID, "bad argument");
        // (the totals are complete only once the pending launches' events have been read — as in _get)
        HIP_CHECK(hipStreamSynchronize(model->m.stream));
        model->m.conv_profile.collect();
        const auto& sl = model->m.conv_profile.by_group[group];
        *launches = sl.launches; *total_ms = sl.ms; *total_flops = sl.flops;
    });
}
extern "C" int mrcnn_model_conv_profile_shapes(mrcnn_model* model, mrcnn_conv_shape_stat* out, int capacity, int* count)
{
    return guarded([&] {
        MRCNN_REQUIRE(model && count && (out || capacity == 0) && capacity >= 0, MRCNN_ERR_INVALID, "bad argument");
        HIP_CHECK(hipStreamSynchronize(model->m.stream));
        model->m.conv_profile.collect();
        const auto& shapes = model->m.conv_profile.by_shape;
        *count = (int)shapes.size();
        int i = 0;
        for (const auto& kv : shapes) {
            if (i >= capacity) break;
            out[i++] = {kv.first.M, kv.first.N, kv.first.K, kv.first.tile, (int64_t)kv.second.launches, kv.second.ms, kv.second.flops, kv.second.bytes};
        }
    });
}

// ================================================================================================
// convolution micro-benchmark (bench.py roofline leg)
// ================================================================================================
extern "C" int mrcnn_bench_conv(int batch, int h, int w, int cin, int cout, int ksize, int stride, int iters, float* avg_ms,
                                double* flops)
{
    return mrcnn_bench_conv_dtype(batch, h, w, cin, cout, ksize, stride, iters, MRCNN_F32, avg_ms, flops);
}

extern "C" int mrcnn_bench_conv_dtype(int batch, int h, int w, int cin, int cout, int ksize, int stride, int iters, int dtype,
                                      float* avg_ms, double* flops)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(avg_ms && flops && iters >= 1 && (ksize == 1 || ksize == 3) && cin % 64 == 0, MRCNN_ERR_INVALID, "bad bench_conv arguments");
        MRCNN_REQUIRE(dtype == MRCNN_F32 || dtype == MRCNN_F16 || dtype == MRCNN_F32S || dtype == MRCNN_F32X3, MRCNN_ERR_UNSUPPORTED, "bench_conv: dtype %d", dtype);
        const size_t es = dtype == MRCNN_F16 ? 2 : 4;             // activations
        const size_t ws = dtype == MRCNN_F32 ? 4 : 2;             // filters
        const int pad = ksize / 2;
        const int oh = (h + 2 * pad - ksize) / stride + 1, ow = (w + 2 * pad - ksize) / stride + 1;
        const int bn = conv_n_tile(cout), npad = (cout + bn - 1) / bn * bn;
        const size_t n_in = (size_t)batch * h * w * cin, n_w = (size_t)npad * ksize * ksize * cin, n_out = (size_t)batch * oh * ow * cout;
        std::mt19937 rng(7);
        std::uniform_real_distribution<float> U(-1.f, 1.f);
        std::vector<float> hs(npad, 1.f), hb(npad, 0.f);
        // random operands in [-1,1) (weights scaled); for fp16 the bit patterns are generated directly
        const size_t pat = 1 << 20;
        std::vector<unsigned char> hin(pat * es), hw(n_w * ws);
        auto fill = [&](unsigned char* dst, size_t n, float scale, size_t esz) {
            for (size_t i = 0; i < n; ++i) {
                const float v = U(rng) * scale;
                if (esz == 4) memcpy(dst + i * 4, &v, 4);
                else { const _Float16 hv = (_Float16)v; memcpy(dst + i * 2, &hv, 2); }
            }
        };
        fill(hin.data(), pat, 1.f, es);
        fill(hw.data(), n_w, 0.05f, ws);
        DevBuf din(n_in * es), dw(n_w * ws), ds(npad * 4), db(npad * 4), dout(n_out * es);
        for (size_t off = 0; off < n_in; off += pat) {
            const size_t c = n_in - off < pat ? n_in - off : pat;
            HIP_CHECK(hipMemcpy((char*)din.p + off * es, hin.data(), c * es, hipMemcpyHostToDevice));
        }
        HIP_CHECK(hipMemcpy(dw.p, hw.data(), n_w * ws, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(ds.p, hs.data(), npad * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(db.p, hb.data(), npad * 4, hipMemcpyHostToDevice));
        ConvDesc d;
        d.dtype = (dtype == MRCNN_F32S || dtype == MRCNN_F32X3) ? MRCNN_F32 : dtype;
        d.wdtype = dtype == MRCNN_F32 ? MRCNN_F32 : (dtype == MRCNN_F32X3 ? MRCNN_F32X3 : MRCNN_F16);
        d.in = din.p; d.B = batch; d.H = h; d.W = w; d.Cin = cin;
        d.in_sW = cin; d.in_sH = (long)w * cin; d.in_sB = (long)h * w * cin;
        d.wgt = dw.p; d.KH = d.KW = ksize; d.stride = stride; d.padH = d.padW = pad;
        d.scale = ds.as<float>(); d.shift = db.as<float>();
        d.OH = oh; d.OW = ow; d.Cout = cout; d.Npad = npad;
        d.out = dout.p; d.out_sP = cout; d.out_sB = (long)oh * ow * cout; d.act = ACT_RELU;
        DevBuf dres;
        if (knob_env("MRCNN_BENCH_RESIDUAL") && atoi(knob_env("MRCNN_BENCH_RESIDUAL"))) {        // the bottleneck blocks' branch2c shape
            dres.alloc(n_out * es);
            HIP_CHECK(hipMemset(dres.p, 0, n_out * es));
            d.res = dres.p; d.res_sB = d.out_sB; d.res_sW = cout; d.res_sH = (long)ow * cout;
        }
        Stream st;
        DevBuf dwh;
        if (ksize == 3 && ws == 2 && es == 4 && conv_halo_packable(ksize, ksize, cin, npad)) {
            conv_halo_pack(st.s, dw.p, npad, cin, dwh);
            d.wgt_halo = dwh.p;
        }
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        DevBuf dw3h;
        if (ws == 2 && es == 2 && stride == 1 && conv3x3h_packable(ksize, ksize, cin, cout, npad)) { conv3x3h_pack(st.s, dw.p, cout, cin, dw3h); d.wgt_c3h = dw3h.p; }
        ConvScratch scratch;                          // the split modes' shared-tile K chunks measure as the engine runs them
        if (ws == 2 && es == 4) { scratch.alloc(); conv_set_scratch(&scratch); }
        try {
            for (int i = 0; i < 2; ++i) conv_forward(st.s, d);
            HIP_CHECK(hipEventRecord(e0, st.s));
            for (int i = 0; i < iters; ++i) conv_forward(st.s, d);
        } catch (...) { conv_set_scratch(nullptr); throw; }
        conv_set_scratch(nullptr);
        HIP_CHECK(hipEventRecord(e1, st.s));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = ms / iters;
        *flops = 2.0 * (double)batch * oh * ow * (double)cout * ksize * ksize * cin;
    });
}

// ================================================================================================
// One convolution of the engine's kernel family on caller data (parity tests of the kernels themselves: every tile
// shape / pipeline variant must give bit-identical results, since the choice depends on the batch size)
// ================================================================================================
static int g_conv2d_alias_res = 0;      // mrcnn_conv2d_nhwc writes its output in place over the residual (tests of the in-place contract)
extern "C" int mrcnn_debug_set(const char* key, int value)
{
    return guarded([&] {
        MRCNN_REQUIRE(key, MRCNN_ERR_INVALID, "null key");
        MRCNN_REQUIRE(test_knobs_armed(), MRCNN_ERR_UNSUPPORTED, "mrcnn_debug_set('%s'): the test / measurement knobs are armed only in a process started with "
                      "MRCNN_TEST_KNOBS=1 (include/maskrcnn_hip_test.h); a production host runs the shipped policy", key);
        if (strcmp(key, "conv2d_alias_res") == 0) { g_conv2d_alias_res = value; return; }
        MRCNN_REQUIRE(conv_debug_set(key, value) || boxes_debug_set(key, value) || engine_debug_set(key, value), MRCNN_ERR_INVALID, "unknown debug key '%s'", key);
    });
}

extern "C" int mrcnn_conv2d_nhwc(const float* in, int batch, int h, int w, int cin, const float* filters, int cout, int ksize, int stride,
                                 const float* scale, const float* shift, const float* residual, int act, int dtype, float* out)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(in && filters && out && batch >= 1 && h >= 1 && w >= 1 && cout >= 1 && (ksize == 1 || ksize == 3) && stride >= 1,
                      MRCNN_ERR_INVALID, "bad conv2d_nhwc argument");
        MRCNN_REQUIRE(dtype == MRCNN_F32 || dtype == MRCNN_F16 || dtype == MRCNN_F32S || dtype == MRCNN_F32X3, MRCNN_ERR_UNSUPPORTED, "conv2d_nhwc: dtype %d", dtype);
        const int adt = dtype == MRCNN_F16 ? MRCNN_F16 : MRCNN_F32;
        const int wdt = dtype == MRCNN_F32 ? MRCNN_F32 : (dtype == MRCNN_F32X3 ? MRCNN_F32X3 : MRCNN_F16);
        MRCNN_REQUIRE(cin % (adt == MRCNN_F16 ? 64 : 32) == 0, MRCNN_ERR_SHAPE, "conv2d_nhwc: Cin %d not a multiple of the K tile", cin);
        const int pad = ksize / 2;
        const int oh = (h + 2 * pad - ksize) / stride + 1, ow = (w + 2 * pad - ksize) / stride + 1;
        const int bn = conv_n_tile(cout), npad = (cout + bn - 1) / bn * bn;
        const size_t n_in = (size_t)batch * h * w * cin, kk = (size_t)ksize * ksize * cin, n_out = (size_t)batch * oh * ow * cout;
        auto to_dev = [&](const float* src, size_t n, size_t n_alloc, bool half, DevBuf& d) {
            if (half) {
                std::vector<_Float16> t(n_alloc, (_Float16)0.f);
                for (size_t i = 0; i < n; ++i) t[i] = (_Float16)src[i];
                d.alloc(n_alloc * 2);
                HIP_CHECK(hipMemcpy(d.p, t.data(), n_alloc * 2, hipMemcpyHostToDevice));
            } else {
                std::vector<float> t(n_alloc, 0.f);
                memcpy(t.data(), src, n * 4);
                d.alloc(n_alloc * 4);
                HIP_CHECK(hipMemcpy(d.p, t.data(), n_alloc * 4, hipMemcpyHostToDevice));
            }
        };
        DevBuf din, dw, ds, db, dres, dout;
        to_dev(in, n_in, n_in, adt == MRCNN_F16, din);
        to_dev(filters, (size_t)cout * kk, (size_t)npad * kk, wdt != MRCNN_F32, dw);
        std::vector<float> hs(npad, 0.f), hb(npad, 0.f);
        for (int o = 0; o < cout; ++o) { hs[o] = scale ? scale[o] : 1.f; hb[o] = shift ? shift[o] : 0.f; }
        ds.alloc((size_t)npad * 4); db.alloc((size_t)npad * 4);
        HIP_CHECK(hipMemcpy(ds.p, hs.data(), (size_t)npad * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(db.p, hb.data(), (size_t)npad * 4, hipMemcpyHostToDevice));
        if (residual) to_dev(residual, n_out, n_out, adt == MRCNN_F16, dres);
        dout.alloc(n_out * 4);
        ConvDesc d;
        d.dtype = adt; d.wdtype = wdt; d.out_f32 = 0;      // fp16 mode stores fp16 (the path the engine's layers take), widened below
        d.in = din.p; d.B = batch; d.H = h; d.W = w; d.Cin = cin;
        d.in_sW = cin; d.in_sH = (long)w * cin; d.in_sB = (long)h * w * cin;
        d.wgt = dw.p; d.KH = d.KW = ksize; d.stride = stride; d.padH = d.padW = pad;
        d.scale = ds.as<float>(); d.shift = db.as<float>();
        d.OH = oh; d.OW = ow; d.Cout = cout; d.Npad = npad;
        d.out = dout.p; d.out_sP = cout; d.out_sB = (long)oh * ow * cout; d.act = act;
        if (residual) { d.res = dres.p; d.res_sB = d.out_sB; d.res_sW = cout; d.res_sH = (long)ow * cout; }
        // mrcnn_debug_set("conv2d_alias_res", 1): the output is written IN PLACE over the residual, the way the engine runs every
        // bottleneck block's branch2c (engine.hip: `to = sc`).  The contract every epilogue must keep (ConvDesc::res in kernels.h):
        // a residual element is loaded by the thread that stores the output element at the same address, before that store.
        if (residual && g_conv2d_alias_res) d.out = dres.p;
        Stream st;
        DevBuf dwh;
        if (ksize == 3 && wdt != MRCNN_F32 && adt == MRCNN_F32 && conv_halo_packable(ksize, ksize, cin, npad)) {
            conv_halo_pack(st.s, dw.p, npad, cin, dwh);
            d.wgt_halo = dwh.p;
        }
        DevBuf dw3h;
        if (adt == MRCNN_F16 && conv3x3h_packable(ksize, ksize, cin, cout, npad)) { conv3x3h_pack(st.s, dw.p, cout, cin, dw3h); d.wgt_c3h = dw3h.p; }
        ConvScratch scratch;                          // one short-lived scratch for the call (kernels.h): the shared-tile K chunks need it
        if (wdt != MRCNN_F32 && adt == MRCNN_F32) { scratch.alloc(); conv_set_scratch(&scratch); }
        try { conv_forward(st.s, d); } catch (...) { conv_set_scratch(nullptr); throw; }
        conv_set_scratch(nullptr);
        HIP_CHECK(hipStreamSynchronize(st.s));
        const void* const result = d.out;
        if (adt == MRCNN_F16) {
            std::vector<_Float16> t(n_out);
            HIP_CHECK(hipMemcpy(t.data(), result, n_out * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n_out; ++i) out[i] = (float)t[i];
        } else {
            HIP_CHECK(hipMemcpy(out, result, n_out * 4, hipMemcpyDeviceToHost));
        }
    });
}

// An identity bottleneck block of the fp16 mode on caller (host) data — the unit tests/test_gpu_bneck.py compares bit for bit:
// fused = 1: the single persistent launch (kernels_bneck.hip); 0: the three launches of the 128-row / ping-pong kernels.
extern "C" int mrcnn_bottleneck_nhwc(const float* x, int batch, int h, int w, int cmid, const float* w1, const float* w2, const float* w3,
                                     const float* s1, const float* h1, const float* s2, const float* h2, const float* s3, const float* h3,
                                     int fused, int iters, float* out, float* avg_ms)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(x && w1 && w2 && w3 && s1 && h1 && s2 && h2 && s3 && h3 && out && batch >= 1 && h >= 1 && w >= 1 && cmid >= 64 && cmid % 64 == 0,
                      MRCNN_ERR_INVALID, "bad bottleneck_nhwc argument");
        const int C = cmid, C4 = 4 * cmid;
        auto half_dev = [&](const float* src, size_t n, DevBuf& d) {
            std::vector<_Float16> t(n);
            for (size_t i = 0; i < n; ++i) t[i] = (_Float16)src[i];
            d.alloc(n * 2);
            HIP_CHECK(hipMemcpy(d.p, t.data(), n * 2, hipMemcpyHostToDevice));
        };
        auto f32_dev = [&](const float* src, size_t n, DevBuf& d) {
            d.alloc(n * 4);
            HIP_CHECK(hipMemcpy(d.p, src, n * 4, hipMemcpyHostToDevice));
        };
        const size_t npix = (size_t)batch * h * w;
        DevBuf dx, dy, dt1, dt2, dw1, dw2, dw3, ds1, dh1, ds2, dh2, ds3, dh3;
        half_dev(x, npix * C4, dx);
        half_dev(w1, (size_t)C * C4, dw1);
        half_dev(w2, (size_t)C * 9 * C, dw2);
        half_dev(w3, (size_t)C4 * C, dw3);
        f32_dev(s1, C, ds1); f32_dev(h1, C, dh1); f32_dev(s2, C, ds2); f32_dev(h2, C, dh2); f32_dev(s3, C4, ds3); f32_dev(h3, C4, dh3);
        dy.alloc(npix * C4 * 2); dt1.alloc(npix * C * 2); dt2.alloc(npix * C * 2);
        HIP_CHECK(hipMemset(dy.p, 0xff, npix * C4 * 2));
        auto desc = [&](const void* in, int cin, const void* wgt, int k, const float* sc, const float* sh, void* o, int cout) {
            ConvDesc d;
            d.dtype = MRCNN_F16; d.wdtype = MRCNN_F16;
            d.in = in; d.B = batch; d.H = h; d.W = w; d.Cin = cin;
            d.in_sW = cin; d.in_sH = (long)w * cin; d.in_sB = (long)h * w * cin;
            d.wgt = wgt; d.KH = d.KW = k; d.stride = 1; d.padH = d.padW = k / 2;
            d.scale = sc; d.shift = sh;
            d.OH = h; d.OW = w; d.Cout = cout; d.Npad = cout;
            d.out = o; d.out_sP = cout; d.out_sB = (long)h * w * cout; d.act = ACT_RELU;
            return d;
        };
        ConvDesc da = desc(dx.p, C4, dw1.p, 1, ds1.as<float>(), dh1.as<float>(), dt1.p, C);
        ConvDesc db = desc(dt1.p, C, dw2.p, 3, ds2.as<float>(), dh2.as<float>(), dt2.p, C);
        ConvDesc dc = desc(dt2.p, C, dw3.p, 1, ds3.as<float>(), dh3.as<float>(), dy.p, C4);
        dc.res = dx.p; dc.res_sW = C4; dc.res_sH = (long)w * C4; dc.res_sB = (long)h * w * C4;
        Stream st;
        DevBuf dw1f, dw2f, dw3f;
        if (bneck_frag_wanted(1, 1, C4, C)) { bneck_pack_frag(st.s, dw1.p, C, C4, dw1f); da.wgt_frag = dw1f.p; }
        if (bneck_frag_wanted(3, 3, C, C)) { bneck_pack_frag(st.s, dw2.p, C, 9 * C, dw2f); db.wgt_frag = dw2f.p; }
        if (bneck_frag_wanted(1, 1, C, C4)) { bneck_pack_frag(st.s, dw3.p, C4, C, dw3f); dc.wgt_frag = dw3f.p; }
        MRCNN_REQUIRE(!fused || conv_bneck_fusable(da, db, dc), MRCNN_ERR_UNSUPPORTED, "bottleneck_nhwc: C %d at %dx%d does not qualify for the fused launch", C, h, w);
        auto run = [&] {
            if (fused) {        // the fused launch whatever the grid size (conv_bneck_forward sends under-filled grids to the three launches)
                static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
                bneck_launch(st.s, C, da.in, dc.out, batch, h, w, da.wgt, db.wgt, dc.wgt, da.scale, da.shift, db.scale, db.shift, dc.scale, dc.shift, nullptr, n_cus,
                             fused == 2 ? nullptr : db.wgt_frag, fused == 2 ? nullptr : dc.wgt_frag, fused == 2 ? nullptr : da.wgt_frag);
            }
            else { conv_forward(st.s, da); conv_forward(st.s, db); conv_forward(st.s, dc); }
        };
        run();
        HIP_CHECK(hipStreamSynchronize(st.s));
        if (iters > 0 && avg_ms) {
            hipEvent_t e0, e1;
            HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, st.s));
            for (int i = 0; i < iters; ++i) run();
            HIP_CHECK(hipEventRecord(e1, st.s));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            *avg_ms = ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        std::vector<_Float16> t(npix * C4);
        HIP_CHECK(hipMemcpy(t.data(), dy.p, t.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < t.size(); ++i) out[i] = (float)t[i];
    });
}

// The stage-entry block (res2a of the fp16 mode: x (B,H,W,C) -> (B,H,W,4C), shortcut = the 1x1 convolution ws of x) — fused = 1: one launch
// (kernels_bneck.hip, FIRST form); 0: the four launches.  tests/test_gpu_bneck.py compares them bit for bit.
extern "C" int mrcnn_bottleneck_first_nhwc(const float* x, int batch, int h, int w, int cmid, const float* w1, const float* w2, const float* w3, const float* ws,
                                           const float* const bn[8], int fused, int iters, float* out, float* avg_ms)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(x && w1 && w2 && w3 && ws && bn && out && batch >= 1 && h >= 1 && w >= 1 && cmid >= 64 && cmid % 64 == 0, MRCNN_ERR_INVALID, "bad bottleneck_first_nhwc argument");
        const int C = cmid, C4 = 4 * cmid;
        auto half_dev = [&](const float* src, size_t n, DevBuf& d) {
            std::vector<_Float16> t(n);
            for (size_t i = 0; i < n; ++i) t[i] = (_Float16)src[i];
            d.alloc(n * 2);
            HIP_CHECK(hipMemcpy(d.p, t.data(), n * 2, hipMemcpyHostToDevice));
        };
        const size_t npix = (size_t)batch * h * w;
        DevBuf dx, dy, dsc, dt1, dt2, dw1, dw2, dw3, dws, dbn[8];
        half_dev(x, npix * C, dx);
        half_dev(w1, (size_t)C * C, dw1); half_dev(w2, (size_t)C * 9 * C, dw2); half_dev(w3, (size_t)C4 * C, dw3); half_dev(ws, (size_t)C4 * C, dws);
        const int bn_n[8] = {C, C, C, C, C4, C4, C4, C4};
        for (int i = 0; i < 8; ++i) { dbn[i].alloc((size_t)bn_n[i] * 4); HIP_CHECK(hipMemcpy(dbn[i].p, bn[i], (size_t)bn_n[i] * 4, hipMemcpyHostToDevice)); }
        dy.alloc(npix * C4 * 2); dsc.alloc(npix * C4 * 2); dt1.alloc(npix * C * 2); dt2.alloc(npix * C * 2);
        HIP_CHECK(hipMemset(dy.p, 0xff, npix * C4 * 2));
        auto desc = [&](const void* in, int cin, const void* wgt, int k, int bi, void* o, int cout, int act) {
            ConvDesc d;
            d.dtype = MRCNN_F16; d.wdtype = MRCNN_F16;
            d.in = in; d.B = batch; d.H = h; d.W = w; d.Cin = cin;
            d.in_sW = cin; d.in_sH = (long)w * cin; d.in_sB = (long)h * w * cin;
            d.wgt = wgt; d.KH = d.KW = k; d.stride = 1; d.padH = d.padW = k / 2;
            d.scale = dbn[bi].as<float>(); d.shift = dbn[bi + 1].as<float>();
            d.OH = h; d.OW = w; d.Cout = cout; d.Npad = cout;
            d.out = o; d.out_sP = cout; d.out_sB = (long)h * w * cout; d.act = act;
            return d;
        };
        ConvDesc da = desc(dx.p, C, dw1.p, 1, 0, dt1.p, C, ACT_RELU);
        ConvDesc db = desc(dt1.p, C, dw2.p, 3, 2, dt2.p, C, ACT_RELU);
        ConvDesc dc = desc(dt2.p, C, dw3.p, 1, 4, dy.p, C4, ACT_RELU);
        ConvDesc ds = desc(dx.p, C, dws.p, 1, 6, dsc.p, C4, ACT_NONE);
        dc.res = dsc.p; dc.res_sW = C4; dc.res_sH = (long)w * C4; dc.res_sB = (long)h * w * C4;
        MRCNN_REQUIRE(!fused || conv_bneck_first_fusable(da, db, dc, ds), MRCNN_ERR_UNSUPPORTED, "bottleneck_first_nhwc: C %d at %dx%d does not qualify for the fused launch", C, h, w);
        Stream st;
        static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
        auto run = [&] {
            if (fused) bneck_launch(st.s, C, da.in, dc.out, batch, h, w, da.wgt, db.wgt, dc.wgt, da.scale, da.shift, db.scale, db.shift, dc.scale, dc.shift, nullptr, n_cus,
                                    nullptr, nullptr, nullptr, ds.wgt, ds.scale, ds.shift);
            else { conv_forward(st.s, da); conv_forward(st.s, ds); conv_forward(st.s, db); conv_forward(st.s, dc); }
        };
        run();
        HIP_CHECK(hipStreamSynchronize(st.s));
        if (iters > 0 && avg_ms) {
            hipEvent_t e0, e1;
            HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, st.s));
            for (int i = 0; i < iters; ++i) run();
            HIP_CHECK(hipEventRecord(e1, st.s));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            *avg_ms = ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        std::vector<_Float16> t(npix * C4);
        HIP_CHECK(hipMemcpy(t.data(), dy.p, t.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < t.size(); ++i) out[i] = (float)t[i];
    });
}

// The identity blocks of a C = 256 stage on caller data (tests/test_gpu_bneck.py): nlayers blocks with stacked operands — w1 (n,C,4C), w2 (n,C,3,3,C),
// w3 (n,4C,C), bn[6] = s1,h1,s2,h2 (n,C) and s3,h3 (n,4C).  form 1: ONE launch (kernels_bneck.hip, STAGE form: tiles wait for their neighbours' previous
// block); form 0: one fused launch per block.  out = the last block's output.  status_flag (optional) receives the launch's flag word (bit 1: no progress).
extern "C" int mrcnn_bottleneck_stage_nhwc(const float* x, int batch, int h, int w, int nlayers, const float* w1, const float* w2, const float* w3,
                                           const float* const bn[6], int form, int iters, float* out, float* avg_ms, int* status_flag)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(x && w1 && w2 && w3 && bn && out && batch >= 1 && h >= 1 && w >= 1 && nlayers >= 1, MRCNN_ERR_INVALID, "bad bottleneck_stage_nhwc argument");
        const int C = 256, C4 = 1024;
        MRCNN_REQUIRE(bneck_geometry_ok(C, h, w), MRCNN_ERR_UNSUPPORTED, "bottleneck_stage_nhwc: %dx%d does not qualify", h, w);
        auto half_dev = [&](const float* src, size_t n, DevBuf& d) {
            std::vector<_Float16> t(n);
            for (size_t i = 0; i < n; ++i) t[i] = (_Float16)src[i];
            d.alloc(n * 2);
            HIP_CHECK(hipMemcpy(d.p, t.data(), n * 2, hipMemcpyHostToDevice));
        };
        const size_t npix = (size_t)batch * h * w;
        DevBuf dx, dx0, dy, dbn[6], flag;
        half_dev(x, npix * C4, dx0);
        dx.alloc(npix * C4 * 2); dy.alloc(npix * C4 * 2);
        flag.alloc(sizeof(int));
        const size_t bn_n[6] = {(size_t)C, (size_t)C, (size_t)C, (size_t)C, (size_t)C4, (size_t)C4};
        for (int i = 0; i < 6; ++i) { dbn[i].alloc(bn_n[i] * nlayers * 4); HIP_CHECK(hipMemcpy(dbn[i].p, bn[i], bn_n[i] * nlayers * 4, hipMemcpyHostToDevice)); }
        Stream st;
        std::vector<DevBuf> dw(3 * (size_t)nlayers), df(3 * (size_t)nlayers);
        const size_t rb = bneck_layer_record_bytes();
        std::vector<unsigned char> recs(rb * nlayers);
        for (int l = 0; l < nlayers; ++l) {
            half_dev(w1 + (size_t)l * C * C4, (size_t)C * C4, dw[3 * l]);
            half_dev(w2 + (size_t)l * C * 9 * C, (size_t)C * 9 * C, dw[3 * l + 1]);
            half_dev(w3 + (size_t)l * C4 * C, (size_t)C4 * C, dw[3 * l + 2]);
            bneck_pack_frag(st.s, dw[3 * l].p, C, C4, df[3 * l]);
            bneck_pack_frag(st.s, dw[3 * l + 1].p, C, 9 * C, df[3 * l + 1]);
            bneck_pack_frag(st.s, dw[3 * l + 2].p, C4, C, df[3 * l + 2]);
            bneck_layer_record(recs.data() + l * rb, df[3 * l].p, df[3 * l + 1].p, df[3 * l + 2].p, dbn[0].as<float>() + (size_t)l * C, dbn[1].as<float>() + (size_t)l * C,
                               dbn[2].as<float>() + (size_t)l * C, dbn[3].as<float>() + (size_t)l * C, dbn[4].as<float>() + (size_t)l * C4, dbn[5].as<float>() + (size_t)l * C4);
        }
        const int ntiles = batch * (h / 8) * (w / 16);
        DevBuf tab(recs.size() + (size_t)ntiles * sizeof(unsigned));
        HIP_CHECK(hipMemcpy(tab.p, recs.data(), recs.size(), hipMemcpyHostToDevice));
        unsigned* const done = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(tab.p) + recs.size());
        static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
        auto run = [&] {
            HIP_CHECK(hipMemcpyAsync(dx.p, dx0.p, npix * C4 * 2, hipMemcpyDeviceToDevice, st.s));      // (the blocks overwrite the input tensor from the second one on)
            HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(int), st.s));
            if (form == 1) bneck_stage_launch(st.s, tab.p, nlayers, dx.p, dy.p, batch, h, w, done, flag.as<int>(), n_cus);
            else
                for (int l = 0; l < nlayers; ++l) {
                    void* const pi = (l & 1) ? dy.p : dx.p;
                    void* const po = (l & 1) ? dx.p : dy.p;
                    bneck_launch(st.s, C, pi, po, batch, h, w, dw[3 * l].p, dw[3 * l + 1].p, dw[3 * l + 2].p, dbn[0].as<float>() + (size_t)l * C, dbn[1].as<float>() + (size_t)l * C,
                                 dbn[2].as<float>() + (size_t)l * C, dbn[3].as<float>() + (size_t)l * C, dbn[4].as<float>() + (size_t)l * C4, dbn[5].as<float>() + (size_t)l * C4,
                                 flag.as<int>(), n_cus, df[3 * l + 1].p, df[3 * l + 2].p, df[3 * l].p);
                }
        };
        run();
        HIP_CHECK(hipStreamSynchronize(st.s));
        if (status_flag) HIP_CHECK(hipMemcpy(status_flag, flag.p, sizeof(int), hipMemcpyDeviceToHost));
        if (iters > 0 && avg_ms) {
            hipEvent_t e0, e1;
            HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, st.s));
            for (int i = 0; i < iters; ++i) run();
            HIP_CHECK(hipEventRecord(e1, st.s));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            *avg_ms = ms / iters;        // (includes the input copy and the two memsets of a run: the same for both forms)
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        std::vector<_Float16> t(npix * C4);
        HIP_CHECK(hipMemcpy(t.data(), (nlayers & 1) ? dy.p : dx.p, t.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < t.size(); ++i) out[i] = (float)t[i];
    });
}

// ================================================================================================
// anchors on demand (MaskRCNNConfig.swift:14 "TODO: generate the anchors on demand")
// Host code.  Restates the published Matterport generator the reference's converter dumps to
// anchors.bin (task.py:173-176): level-major P2..P6, then y, x, ratio; float64 arithmetic, one
// rounding to float32 at the end — bit-identical to mask-rcnn-coreml_amd/anchors.py.
// ================================================================================================
extern "C" int mrcnn_generate_anchors(int image_h, int image_w, float* out, int64_t capacity, int64_t* count)
{
    return guarded([&] {
        MRCNN_REQUIRE(image_h > 0 && image_w > 0 && count, MRCNN_ERR_INVALID, "bad generate_anchors argument");
        const double scales[5] = {32, 64, 128, 256, 512}, ratios[3] = {0.5, 1.0, 2.0};
        const int strides[5] = {4, 8, 16, 32, 64};
        int64_t n = 0;
        for (int l = 0; l < 5; ++l) n += (int64_t)((image_h + strides[l] - 1) / strides[l]) * ((image_w + strides[l] - 1) / strides[l]) * 3;
        *count = n;
        if (!out) return;
        MRCNN_REQUIRE(capacity >= n * 4, MRCNN_ERR_SHAPE, "anchors need %lld floats, buffer holds %lld", (long long)n * 4, (long long)capacity);
        const double sy = (double)(image_h - 1), sx = (double)(image_w - 1);
        float* o = out;
        for (int l = 0; l < 5; ++l) {
            const int fh = (image_h + strides[l] - 1) / strides[l], fw = (image_w + strides[l] - 1) / strides[l];
            for (int y = 0; y < fh; ++y)
                for (int x = 0; x < fw; ++x)
                    for (int r = 0; r < 3; ++r) {
                        const double h = scales[l] / sqrt(ratios[r]), w = scales[l] * sqrt(ratios[r]);
                        const double cy = (double)(y * strides[l]), cx = (double)(x * strides[l]);
                        o[0] = (float)(((cy - 0.5 * h) - 0.0) / sy);
                        o[1] = (float)(((cx - 0.5 * w) - 0.0) / sx);
                        o[2] = (float)(((cy + 0.5 * h) - 1.0) / sy);
                        o[3] = (float)(((cx + 0.5 * w) - 1.0) / sx);
                        o += 4;
                    }
        }
    });
}

// ================================================================================================
// letterbox (`.scaleFit`, EvaluateCommand.swift:157)
// ================================================================================================
extern "C" int mrcnn_letterbox_geometry(int h, int w, int H, int W, int* nh, int* nw, int* pad_y, int* pad_x)
{
    return guarded([&] {
        MRCNN_REQUIRE(h > 0 && w > 0 && H > 0 && W > 0 && nh && nw && pad_y && pad_x, MRCNN_ERR_INVALID, "bad letterbox argument");
        const double sc = fmin((double)W / (double)w, (double)H / (double)h);
        int a = (int)floor((double)h * sc + 0.5), b = (int)floor((double)w * sc + 0.5);
        a = a < 1 ? 1 : (a > H ? H : a);
        b = b < 1 ? 1 : (b > W ? W : b);
        *nh = a; *nw = b; *pad_y = (H - a) / 2; *pad_x = (W - b) / 2;
    });
}

extern "C" int mrcnn_letterbox_rgb(const uint8_t* src, int h, int w, int memspace, uint8_t* dst, int H, int W)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(src && dst, MRCNN_ERR_INVALID, "null buffer");
        int nh, nw, py, px;
        MRCNN_REQUIRE(mrcnn_letterbox_geometry(h, w, H, W, &nh, &nw, &py, &px) == MRCNN_OK, MRCNN_ERR_INVALID, "bad letterbox geometry");
        Stream st;
        DevBuf ts, td;
        const uint8_t* s = src;
        uint8_t* d = dst;
        if (memspace != MRCNN_DEVICE) {
            ts.alloc((size_t)h * w * 3);
            HIP_CHECK(hipMemcpy(ts.p, src, (size_t)h * w * 3, hipMemcpyHostToDevice));
            td.alloc((size_t)H * W * 3);
            s = ts.as<uint8_t>(); d = td.as<uint8_t>();
        }
        letterbox_forward(st.s, s, h, w, d, H, W, nh, nw, py, px);
        HIP_CHECK(hipStreamSynchronize(st.s));
        if (memspace != MRCNN_DEVICE) HIP_CHECK(hipMemcpy(dst, td.p, (size_t)H * W * 3, hipMemcpyDeviceToHost));
    });
}

// ================================================================================================
// mask paste (DetectionRenderer.swift:13-24)
// ================================================================================================
extern "C" int mrcnn_paste_masks(const float* detections, int64_t det_stride, const float* masks, int n, int mask_size, int image_h,
                                 int image_w, float threshold, int memspace, uint8_t* out)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(detections && masks && out && n >= 0 && det_stride >= 6 && mask_size >= 2 && image_h > 0 && image_w > 0,
                      MRCNN_ERR_INVALID, "bad paste_masks argument");
        if (n == 0) return;
        Stream st;
        DevBuf td, tm, to;
        const float* d = stage_rows(detections, memspace, n, det_stride, det_stride, td);
        const float* m = stage_rows(masks, memspace, n, (long)mask_size * mask_size, (long)mask_size * mask_size, tm);
        uint8_t* o = out;
        const size_t bytes = (size_t)n * image_h * image_w;
        if (memspace != MRCNN_DEVICE) { to.alloc(bytes); o = to.as<uint8_t>(); }
        paste_masks_forward(st.s, d, det_stride, m, n, mask_size, image_h, image_w, threshold, o);
        HIP_CHECK(hipStreamSynchronize(st.s));
        if (memspace != MRCNN_DEVICE) HIP_CHECK(hipMemcpy(out, to.p, bytes, hipMemcpyDeviceToHost));
    });
}

// ================================================================================================
// result decoding (Detection.swift:23-99) — host
// ================================================================================================
extern "C" int mrcnn_detections_decode(const float* det, int64_t n_rows, int64_t row_stride, mrcnn_detection* out, int64_t capacity,
                                       int64_t* count)
{
    return guarded([&] {
        MRCNN_REQUIRE(det && count && (out || capacity == 0) && row_stride >= 6, MRCNN_ERR_INVALID, "bad argument");
        int64_t k = 0;
        for (int64_t i = 0; i < n_rows; ++i) {
            const float* r = det + i * row_stride;
            const double score = (double)r[5];
            if (score > 0.7) {                                                  // Detection.swift:38
                if (k < capacity) {
                    const double y1 = r[0], x1 = r[1], y2 = r[2], x2 = r[3];
                    out[k].index = i;
                    out[k].x = x1; out[k].y = y1; out[k].w = x2 - x1; out[k].h = y2 - y1;   // :45-55
                    out[k].class_id = (int64_t)r[4];
                    out[k].score = score;
                }
                ++k;
            }
        }
        *count = k;
    });
}

extern "C" int mrcnn_mask_to_u8(const float* mask, int64_t n, uint8_t* out)
{
    return guarded([&] {
        MRCNN_REQUIRE(mask && out && n >= 0, MRCNN_ERR_INVALID, "bad argument");
        for (int64_t i = 0; i < n; ++i) {
            double v = 255.0 - ((double)mask[i] / 2.0 * 255.0);                 // Detection.swift:83-85
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out[i] = (uint8_t)v;
        }
    });
}

// the same for a host that holds the mask as Double, as Core ML hands it to Detection.maskFromFeatureValue (Detection.swift:77)
extern "C" int mrcnn_mask_to_u8_f64(const double* mask, int64_t n, uint8_t* out)
{
    return guarded([&] {
        MRCNN_REQUIRE(mask && out && n >= 0, MRCNN_ERR_INVALID, "bad argument");
        for (int64_t i = 0; i < n; ++i) {
            double v = 255.0 - (mask[i] / 2.0 * 255.0);                         // Detection.swift:83-85
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out[i] = (uint8_t)v;
        }
    });
}
