/*
 * mrcnn_oracle.c — CPU restatement of the reference's custom-layer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline.  The product path
 * (mask-rcnn-coreml_amd/csrc → libmaskrcnn_hip.so) never links or calls it.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (edouardlp/Mask-RCNN-CoreML) is
 * Swift on Apple frameworks (Core ML / Accelerate / MPS) and ships no tests, golden
 * vectors or fixtures; it can be neither compiled nor run here.
 * This file follows the Swift sources line by line (citations below are
 * file:line under /root/reference/Sources/Mask-RCNN-CoreML/) and is pinned only by
 * hand-computed known-answer cases and independent brute-force re-implementations in
 * tests/test_oracle_*.py.
 *
 * Deliberate tightenings where the reference leaves behaviour unspecified
 * (SURVEY.md §7 "Semantics to preserve", Q-numbers):
 *   Q2  arg-sort ties (vDSP_vsorti)           → score desc, then index asc
 *   Q5  `exp` on Float                        → orc_expf(): an all-IEEE-basic-op algorithm in
 *                                                double, rounded once to float (bit-reproducible
 *                                                on x86 and gfx950; equals a correctly rounded
 *                                                expf except with probability ~2^-29)
 *   Q7  NMS called with 0..<4N (would trap)   → iterate 0..<N
 *   Q11 MPSNNCropAndResizeBilinear sampling   → TensorFlow crop_and_resize (published algorithm,
 *                                                tensorflow/core/kernels/image/crop_and_resize_op.cc),
 *                                                extrapolation value 0
 *   Q12 vDSP_maxvi ties                       → lowest index
 *   Q14 Set<Float> iteration / unstable sort  → classes ascending; stable sort
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * orc_expf — stands in for Swift `exp(_: Float)` (BoxUtils.swift:57-58).
 * Cody-Waite reduction + degree-13 Taylor/Horner in double with explicit fma, scaled by an
 * exactly constructed power of two, rounded once to float.
 * ---------------------------------------------------------------------------------------- */
static const double ORC_LOG2E = 0x1.71547652b82fep+0;
static const double ORC_LN2_HI = 0x1.62e42fefa38p-1;    /* 42 significant bits: n*hi exact */
static const double ORC_LN2_LO = 0x1.ef35793c7673p-45;

ORC_API float orc_expf(float xf)
{
    if (xf != xf) return xf;
    if (xf > 88.72284f) return INFINITY;
    if (xf < -104.0f) return 0.0f;
    double x = (double)xf;
    double n = rint(x * ORC_LOG2E);
    double r = fma(n, -ORC_LN2_HI, x);
    r = fma(n, -ORC_LN2_LO, r);
    double p = 0x1.6124613a86d09p-33;              /* 1/13! */
    p = fma(p, r, 0x1.1eed8eff8d898p-29);           /* 1/12! */
    p = fma(p, r, 0x1.ae64567f544e4p-26);           /* 1/11! */
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);           /* 1/10! */
    p = fma(p, r, 0x1.71de3a556c734p-19);           /* 1/9!  */
    p = fma(p, r, 0x1.a01a01a01a01ap-16);           /* 1/8!  */
    p = fma(p, r, 0x1.a01a01a01a01ap-13);           /* 1/7!  */
    p = fma(p, r, 0x1.6c16c16c16c17p-10);           /* 1/6!  */
    p = fma(p, r, 0x1.1111111111111p-7);            /* 1/5!  */
    p = fma(p, r, 0x1.5555555555555p-5);            /* 1/4!  */
    p = fma(p, r, 0x1.5555555555555p-3);            /* 1/3!  */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int64_t e = (int64_t)n + 1023;                  /* n in [-151, 128] → biased exp in [872, 1151] */
    uint64_t bits = (uint64_t)e << 52;
    double s;
    memcpy(&s, &bits, 8);
    return (float)(p * s);
}

/* ------------------------------------------------------------------------------------------
 * Utils.swift
 * ---------------------------------------------------------------------------------------- */

/* stridedSlice (Utils.swift:17-26): result[i*length + l] = p[begin + l + i*stride] */
ORC_API void orc_strided_slice(const float* p, int64_t begin, int64_t count, int64_t stride,
                               int64_t length, float* result)
{
    for (int64_t l = 0; l < length; ++l)
        for (int64_t i = 0; i < count; ++i)
            result[i * length + l] = p[begin + l + i * stride];
}

typedef struct { float v; uint32_t i; } orc_kv;

static int orc_cmp_desc(const void* a, const void* b)
{
    const orc_kv* x = (const orc_kv*)a;
    const orc_kv* y = (const orc_kv*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);           /* Q2: ties → lower index first */
}

/* sortedIndices(ascending:false) (Utils.swift:56-66): full descending arg-sort. */
ORC_API void orc_sorted_indices_desc(const float* v, int64_t n, uint32_t* idx)
{
    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { kv[i].v = v[i]; kv[i].i = (uint32_t)i; }
    qsort(kv, (size_t)n, sizeof(orc_kv), orc_cmp_desc);
    for (int64_t i = 0; i < n; ++i) idx[i] = kv[i].i;
    free(kv);
}

/* elementWiseMultiply (Utils.swift:173-180): m[r][c] *= vec[c] in fp32. */
ORC_API void orc_elementwise_multiply(float* m, const float* vec, int64_t height, int64_t width)
{
    for (int64_t c = 0; c < width; ++c)
        for (int64_t r = 0; r < height; ++r)
            m[r * width + c] = m[r * width + c] * vec[c];
}

/* CGRect(anchorDatum:) (Utils.swift:220-230) + IOU (Utils.swift:232-246).
 * CGFloat is Double.  CGRect.width/.height/.minX/.maxX are the *standardized* accessors
 * (CGRectGetWidth etc.): width = |x2-x1|, minX = min(x1, x1+w), maxX = max(x1, x1+w). */
typedef struct { double x, y, w, h; } orc_rect;

static orc_rect orc_rect_from(const float* d)
{
    double y1 = (double)d[0], x1 = (double)d[1], y2 = (double)d[2], x2 = (double)d[3];
    orc_rect r = { x1, y1, x2 - x1, y2 - y1 };
    return r;
}
static double orc_w(orc_rect r) { return fabs(r.w); }
static double orc_h(orc_rect r) { return fabs(r.h); }
static double orc_minx(orc_rect r) { return r.w < 0 ? r.x + r.w : r.x; }
static double orc_maxx(orc_rect r) { return r.w < 0 ? r.x : r.x + r.w; }
static double orc_miny(orc_rect r) { return r.h < 0 ? r.y + r.h : r.y; }
static double orc_maxy(orc_rect r) { return r.h < 0 ? r.y : r.y + r.h; }

static float orc_iou_rect(orc_rect a, orc_rect b)
{
    double areaA = orc_w(a) * orc_h(a);
    if (areaA <= 0) return 0;
    double areaB = orc_w(b) * orc_h(b);
    if (areaB <= 0) return 0;
    double ix0 = fmax(orc_minx(a), orc_minx(b));
    double iy0 = fmax(orc_miny(a), orc_miny(b));
    double ix1 = fmin(orc_maxx(a), orc_maxx(b));
    double iy1 = fmin(orc_maxy(a), orc_maxy(b));
    double inter = fmax(iy1 - iy0, 0) * fmax(ix1 - ix0, 0);
    return (float)(inter / (areaA + areaB - inter));
}

/* public func IOU (Utils.swift:232): boxes given as (y1,x1,y2,x2) floats. */
ORC_API float orc_iou(const float* a, const float* b)
{
    return orc_iou_rect(orc_rect_from(a), orc_rect_from(b));
}

/* nonMaxSupression (Utils.swift:185-218). Returns #selected. */
ORC_API int64_t orc_nms(const float* boxes, const int64_t* indices, int64_t n_indices,
                        float iou_threshold, int64_t max, int64_t* selected)
{
    int64_t ns = 0;
    for (int64_t t = 0; t < n_indices; ++t) {
        if (ns >= max) return ns;
        int64_t index = indices[t];
        orc_rect a = orc_rect_from(boxes + index * 4);
        int should = orc_w(a) > 0 && orc_h(a) > 0;
        if (should) {
            for (int64_t s = 0; s < ns; ++s) {
                orc_rect b = orc_rect_from(boxes + selected[s] * 4);
                if (orc_iou_rect(a, b) > iou_threshold) { should = 0; break; }
            }
        }
        if (should) selected[ns++] = index;
    }
    return ns;
}

/* ------------------------------------------------------------------------------------------
 * BoxUtils.swift
 * ---------------------------------------------------------------------------------------- */

/* applyBoxDeltas (BoxUtils.swift:32-71), fp32, Swift does not contract mul+add. */
ORC_API void orc_apply_box_deltas(float* boxes, const float* deltas, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float* b = boxes + i * 4;
        float y1 = b[0], x1 = b[1], y2 = b[2], x2 = b[3];
        float dy = deltas[i * 4], dx = deltas[i * 4 + 1], dh = deltas[i * 4 + 2], dw = deltas[i * 4 + 3];
        float height = y2 - y1;
        float width = x2 - x1;
        float centerY = y1 + 0.5f * height;
        float centerX = x1 + 0.5f * width;
        centerY = centerY + dy * height;
        centerX = centerX + dx * width;
        height = height * orc_expf(dh);
        width = width * orc_expf(dw);
        float ry1 = centerY - 0.5f * height;
        float rx1 = centerX - 0.5f * width;
        float ry2 = ry1 + height;
        float rx2 = rx1 + width;
        b[0] = ry1; b[1] = rx1; b[2] = ry2; b[3] = rx2;
    }
}

/* clip (BoxUtils.swift:73-80): vDSP_vclip to [0,1]. */
ORC_API void orc_clip_boxes(float* boxes, int64_t n)
{
    for (int64_t i = 0; i < n * 4; ++i) {
        float v = boxes[i];
        if (v < 0.0f) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
        boxes[i] = v;
    }
}

/* ------------------------------------------------------------------------------------------
 * ProposalLayer.evaluate (ProposalLayer.swift:103-195)
 *   probs  (A,2) f32, deltas (A,4) f32, anchors (A,4) f32 (anchors.bin), out (max_proposals rows,
 *   row stride out_stride floats).  Every one of max_proposals rows is written (Q9).
 *   Optional debug outputs (may be NULL): topk_idx[n_proc], boxes_sorted[n_proc*4], keep[<=max].
 *   Returns the number of proposals kept.
 * ---------------------------------------------------------------------------------------- */
ORC_API int64_t orc_proposal_layer(const float* probs, const float* deltas, const float* anchors,
                                   int64_t A, int64_t pre_nms, int64_t max_proposals,
                                   float nms_thr, const float* std4, float* out, int64_t out_stride,
                                   uint32_t* dbg_topk_idx, float* dbg_boxes, int64_t* dbg_keep)
{
    int64_t n = A < pre_nms ? A : pre_nms;                           /* :120 */
    float* scores = (float*)malloc(sizeof(float) * (size_t)(A > 0 ? A : 1));
    orc_strided_slice(probs, 1, A, 2, 1, scores);                    /* :124  Q1 */
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(A > 0 ? A : 1));
    orc_sorted_indices_desc(scores, A, order);                       /* :133  Q2 */
    float* sd = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    float* sa = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i)                                  /* :140-149 gather (Q3) */
        for (int j = 0; j < 4; ++j) {
            sd[i * 4 + j] = deltas[(int64_t)order[i] * 4 + j];
            sa[i * 4 + j] = anchors[(int64_t)order[i] * 4 + j];
        }
    orc_elementwise_multiply(sd, std4, n, 4);                        /* :158  Q4 */
    orc_apply_box_deltas(sa, sd, n);                                 /* :162  Q5 */
    orc_clip_boxes(sa, n);                                           /* :163  Q6 */
    int64_t* cand = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) cand[i] = i;                     /* :169-172  Q7 */
    int64_t* keep = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_proposals > 0 ? max_proposals : 1));
    int64_t nk = orc_nms(sa, cand, n, nms_thr, max_proposals, keep); /* Q8 */
    for (int64_t i = 0; i < nk; ++i)                                 /* :181-185 */
        for (int j = 0; j < 4; ++j) out[i * out_stride + j] = sa[keep[i] * 4 + j];
    /* :190-192 pad: zeros from nk*stride for (max-nk)*stride elements */
    for (int64_t e = nk * out_stride; e < max_proposals * out_stride; ++e) out[e] = 0.0f;
    if (dbg_topk_idx) memcpy(dbg_topk_idx, order, sizeof(uint32_t) * (size_t)n);
    if (dbg_boxes) memcpy(dbg_boxes, sa, sizeof(float) * 4 * (size_t)n);
    if (dbg_keep) memcpy(dbg_keep, keep, sizeof(int64_t) * (size_t)nk);
    free(scores); free(order); free(sd); free(sa); free(cand); free(keep);
    return nk;
}

/* ------------------------------------------------------------------------------------------
 * PyramidROIAlignLayer
 * ---------------------------------------------------------------------------------------- */

/* roisToInputItems (PyramidROIAlignLayer.swift:351-396).  level_index[i] in 0..3, or -1 = padding. */
ORC_API void orc_roi_levels(const float* rois, int64_t n, int64_t roi_stride,
                            double image_w, double image_h, int32_t* level_index)
{
    double ratio = 224.0 / sqrt(image_w * image_h);                  /* :357 (factor 224 at :98) */
    for (int64_t i = 0; i < n; ++i) {
        const float* r = rois + i * roi_stride;
        double y1 = (double)r[0], x1 = (double)r[1], y2 = (double)r[2], x2 = (double)r[3];
        double width = x2 - x1, height = y2 - y1;
        double lf = log2(sqrt(width * height) / ratio) + 4.0;        /* :373 */
        int valid = !isnan(lf) && !isinf(lf);                        /* :374 */
        int level = 2;
        if (valid) {
            double rr = round(lf);                                   /* half away from zero */
            if (rr < 2.0) rr = 2.0;
            if (rr > 5.0) rr = 5.0;
            level = (int)rr;                                         /* :376 */
        }
        level_index[i] = valid ? level - 2 : -1;
    }
}

/* TF crop_and_resize (bilinear, extrapolation 0) of one ROI from a CHW map, fp32 (Q11). */
static void orc_crop_and_resize_chw(const float* fmap, int64_t C, int64_t H, int64_t W,
                                    float y1, float x1, float y2, float x2, int64_t P, float* out)
{
    float hs = (P > 1) ? (y2 - y1) * (float)(H - 1) / (float)(P - 1) : 0.0f;
    float ws = (P > 1) ? (x2 - x1) * (float)(W - 1) / (float)(P - 1) : 0.0f;
    for (int64_t py = 0; py < P; ++py) {
        float in_y = (P > 1) ? y1 * (float)(H - 1) + (float)py * hs
                             : 0.5f * (y1 + y2) * (float)(H - 1);
        int y_ok = !(in_y < 0 || in_y > (float)(H - 1));
        float fy = floorf(in_y), cy = ceilf(in_y);
        float ly = in_y - fy;
        for (int64_t px = 0; px < P; ++px) {
            float in_x = (P > 1) ? x1 * (float)(W - 1) + (float)px * ws
                                 : 0.5f * (x1 + x2) * (float)(W - 1);
            int x_ok = !(in_x < 0 || in_x > (float)(W - 1));
            if (!y_ok || !x_ok) {
                for (int64_t c = 0; c < C; ++c) out[(c * P + py) * P + px] = 0.0f;
                continue;
            }
            float fx = floorf(in_x), cx = ceilf(in_x);
            float lx = in_x - fx;
            int64_t t = (int64_t)fy, b = (int64_t)cy, l = (int64_t)fx, r = (int64_t)cx;
            for (int64_t c = 0; c < C; ++c) {
                const float* m = fmap + c * H * W;
                float tl = m[t * W + l], tr = m[t * W + r], bl = m[b * W + l], br = m[b * W + r];
                float top = tl + (tr - tl) * lx;
                float bot = bl + (br - bl) * lx;
                out[(c * P + py) * P + px] = top + (bot - top) * ly;
            }
        }
    }
}

/* PyramidROIAlignLayer.evaluate (PyramidROIAlignLayer.swift:79-181): output row i ↔ ROI i
 * (C,P,P) CHW, padding rows zero (copyOutput :265-272).  fmaps[k] is CHW (C,H[k],W[k]). */
ORC_API void orc_pyramid_roi_align(const float* rois, int64_t n, int64_t roi_stride,
                                   const float* const* fmaps, const int64_t* H, const int64_t* W,
                                   int64_t C, int64_t P, double image_w, double image_h,
                                   float* out, int64_t out_stride)
{
    int32_t* lv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    orc_roi_levels(rois, n, roi_stride, image_w, image_h, lv);
    for (int64_t i = 0; i < n; ++i) {
        float* o = out + i * out_stride;
        if (lv[i] < 0) { for (int64_t e = 0; e < C * P * P; ++e) o[e] = 0.0f; continue; }
        const float* r = rois + i * roi_stride;
        int k = lv[i];
        orc_crop_and_resize_chw(fmaps[k], C, H[k], W[k], r[0], r[1], r[2], r[3], P, o);
    }
    free(lv);
}

/* ------------------------------------------------------------------------------------------
 * TimeDistributedClassifierLayer post-processing (TimeDistributedClassifierLayer.swift:50-88):
 * probs (n, nc) and bbox (n, nc*4) as emitted by Classifier.mlmodel (Double in the reference;
 * the Double→Float cast at :69-71 is the identity on values that are already floats).
 * out row i = (dy,dx,dh,dw, classId, score), row stride out_stride.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_classifier_postprocess(const double* probs, const double* bbox, int64_t n,
                                        int64_t nc, float* out, int64_t out_stride)
{
    for (int64_t i = 0; i < n; ++i) {
        int64_t best = 0;
        float bestv = (float)probs[i * nc];
        for (int64_t c = 1; c < nc; ++c) {                           /* maximumValueWithIndex :177-192, Q12 */
            float v = (float)probs[i * nc + c];
            if (v > bestv) { bestv = v; best = c; }
        }
        float* o = out + i * out_stride;
        o[4] = (float)best;                                          /* :79 */
        o[5] = bestv;                                                /* :80 */
        for (int z = 0; z < 4; ++z) o[z] = (float)bbox[i * nc * 4 + best * 4 + z];   /* :82-85 */
    }
}

/* ------------------------------------------------------------------------------------------
 * DetectionLayer.evaluate (DetectionLayer.swift:107-234)
 *   rois (n,4) contiguous, cls (n,6) contiguous (:125-128 assume stride 6),
 *   out (max_det rows, out_stride).  Returns #detections.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float score; int64_t pos; int64_t id; } orc_det;

static int orc_cmp_det(const void* a, const void* b)
{
    const orc_det* x = (const orc_det*)a;
    const orc_det* y = (const orc_det*)b;
    if (x->score > y->score) return -1;                              /* :199-202 a.1 > b.1 */
    if (x->score < y->score) return 1;
    return (x->pos > y->pos) - (x->pos < y->pos);                    /* Q14: stable */
}

ORC_API int64_t orc_detection_layer(const float* rois, const float* cls, int64_t n,
                                    const float* std4, int64_t max_det, float score_thr,
                                    float nms_thr, float* out, int64_t out_stride)
{
    size_t cap = (size_t)(n > 0 ? n : 1);
    float* deltas = (float*)malloc(sizeof(float) * 4 * cap);
    float* class_ids = (float*)malloc(sizeof(float) * cap);
    float* scores = (float*)malloc(sizeof(float) * cap);
    orc_strided_slice(cls, 0, n, 6, 4, deltas);                      /* :125 */
    orc_strided_slice(cls, 4, n, 6, 1, class_ids);                   /* :127 */
    orc_strided_slice(cls, 5, n, 6, 1, scores);                      /* :128 */
    /* indicesOfRoisWithHighScores (:238-276): vthres zeroes score < thr, vcmprs keeps ramp where
     * gated score != 0  (Q13: score >= thr). Then drop background (:136-140). */
    int64_t* f = (int64_t*)malloc(sizeof(int64_t) * cap);
    int64_t nf = 0;
    for (int64_t i = 0; i < n; ++i) {
        float s = scores[i] >= score_thr ? scores[i] : 0.0f;
        if (s != 0.0f && class_ids[i] > 0) f[nf++] = i;
    }
    float* fr = (float*)malloc(sizeof(float) * 4 * cap);
    float* fd = (float*)malloc(sizeof(float) * 4 * cap);
    float* fs = (float*)malloc(sizeof(float) * cap);
    float* fc = (float*)malloc(sizeof(float) * cap);
    for (int64_t k = 0; k < nf; ++k) {                               /* :144-154 */
        for (int j = 0; j < 4; ++j) { fr[k * 4 + j] = rois[f[k] * 4 + j]; fd[k * 4 + j] = deltas[f[k] * 4 + j]; }
        fs[k] = scores[f[k]];
        fc[k] = class_ids[f[k]];
    }
    orc_elementwise_multiply(fd, std4, nf, 4);                       /* :159 */
    orc_apply_box_deltas(fr, fd, nf);                                /* :163 */
    orc_clip_boxes(fr, nf);                                          /* :164 */
    /* per-class NMS (:166-183), classes ascending (Q14), candidates in ROI order */
    int64_t* nms_ids = (int64_t*)malloc(sizeof(int64_t) * cap);
    int64_t n_ids = 0;
    int64_t* of_class = (int64_t*)malloc(sizeof(int64_t) * cap);
    int64_t* sel = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_det > 0 ? max_det : 1));
    float* uniq = (float*)malloc(sizeof(float) * cap);
    int64_t nu = 0;
    for (int64_t k = 0; k < nf; ++k) {
        int seen = 0;
        for (int64_t u = 0; u < nu; ++u) if (uniq[u] == fc[k]) { seen = 1; break; }
        if (!seen) uniq[nu++] = fc[k];
    }
    for (int64_t a = 1; a < nu; ++a) {                               /* insertion sort ascending */
        float v = uniq[a]; int64_t b = a - 1;
        while (b >= 0 && uniq[b] > v) { uniq[b + 1] = uniq[b]; --b; }
        uniq[b + 1] = v;
    }
    for (int64_t u = 0; u < nu; ++u) {
        int64_t nc = 0;
        for (int64_t k = 0; k < nf; ++k) if (fc[k] == uniq[u]) of_class[nc++] = k;
        int64_t ns = orc_nms(fr, of_class, nc, nms_thr, max_det, sel);
        for (int64_t s = 0; s < ns; ++s) nms_ids[n_ids++] = sel[s];
    }
    /* top max_det by score (:186-209) */
    orc_det* d = (orc_det*)malloc(sizeof(orc_det) * cap);
    for (int64_t k = 0; k < n_ids; ++k) { d[k].score = fs[nms_ids[k]]; d[k].pos = k; d[k].id = nms_ids[k]; }
    qsort(d, (size_t)n_ids, sizeof(orc_det), orc_cmp_det);
    int64_t nd = n_ids < max_det ? n_ids : max_det;
    for (int64_t i = 0; i < nd; ++i) {                               /* :217-224 */
        int64_t r = d[i].id;
        float* o = out + i * out_stride;
        for (int j = 0; j < 4; ++j) o[j] = fr[r * 4 + j];
        o[4] = fc[r];
        o[5] = fs[r];
    }
    for (int64_t e = nd * out_stride; e < max_det * out_stride; ++e) out[e] = 0.0f;   /* :226-231 */
    free(deltas); free(class_ids); free(scores); free(f); free(fr); free(fd); free(fs); free(fc);
    free(nms_ids); free(of_class); free(sel); free(uniq); free(d);
    return nd;
}

/* ------------------------------------------------------------------------------------------
 * TimeDistributedMaskLayer (TimeDistributedMaskLayer.swift:39-91)
 * Split in two so that the caller can run the Mask model in between:
 *   orc_mask_valid_rows  = MultiArrayBatchProvider(removeZeros:true) index mapping
 *                          (TimeDistributedClassifierLayer.swift:116-127): row kept iff every
 *                          element of its `row_len` floats is != 0.
 *   orc_mask_layer_write = the copy/select/pad loop (:58-89).  `masks` holds the Mask model's
 *                          output for the kept rows only: (n_kept, nc, mh*mw) doubles.
 *   `out` is IN/OUT: rows the reference never writes keep whatever the caller had there (Q9/Q15).
 * ---------------------------------------------------------------------------------------- */
ORC_API int64_t orc_mask_valid_rows(const float* pooled, int64_t n, int64_t row_stride, int64_t row_len,
                                    int64_t* index_mapping)
{
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float* r = pooled + i * row_stride;
        int all = 1;
        for (int64_t e = 0; e < row_len; ++e) if (!(r[e] != 0)) { all = 0; break; }
        if (all) index_mapping[k++] = i;
    }
    return k;
}

ORC_API void orc_mask_layer_write(const double* masks, int64_t n_kept, const int64_t* index_mapping,
                                  int64_t nc, int64_t mask_len, const float* detections,
                                  int64_t det_count, int64_t det_stride, float* out, int64_t out_stride)
{
    for (int64_t i = 0; i < n_kept; ++i) {
        int64_t actual = index_mapping[i];                           /* :60 */
        int64_t class_id = (int64_t)detections[det_stride * i + 4];  /* :71 — compact index i (sic) */
        if (class_id < 0) class_id = 0;
        if (class_id >= nc) class_id = nc - 1;                       /* reference would read OOB */
        const double* src = masks + (i * nc + class_id) * mask_len;  /* :75 */
        float* dst = out + out_stride * actual;                      /* :83 */
        for (int64_t e = 0; e < out_stride; ++e) dst[e] = (float)src[e];   /* copies `stride` elements */
    }
    for (int64_t e = n_kept * out_stride; e < det_count * out_stride; ++e) out[e] = 0.0f;  /* :87-89 */
}

/* ------------------------------------------------------------------------------------------
 * Detection.detectionsFromFeatureValue (Detection.swift:23-62) and maskFromFeatureValue (:64-99)
 *   det (count, stride) f32.  Outputs per kept detection: index, (x, y, w, h) doubles, classId, score.
 * ---------------------------------------------------------------------------------------- */
ORC_API int64_t orc_detections_decode(const float* det, int64_t count, int64_t stride,
                                      int64_t* out_index, double* out_xywh, int64_t* out_class,
                                      double* out_score)
{
    int64_t k = 0;
    for (int64_t i = 0; i < count; ++i) {
        double score = (double)det[i * stride + 5];
        if (score > 0.7) {                                           /* :38 (Double literal) */
            double y1 = (double)det[i * stride], x1 = (double)det[i * stride + 1];
            double y2 = (double)det[i * stride + 2], x2 = (double)det[i * stride + 3];
            out_index[k] = i;
            out_xywh[k * 4 + 0] = x1; out_xywh[k * 4 + 1] = y1;
            out_xywh[k * 4 + 2] = x2 - x1; out_xywh[k * 4 + 3] = y2 - y1;
            out_class[k] = (int64_t)det[i * stride + 4];
            out_score[k] = score;
            ++k;
        }
    }
    return k;
}

/* UInt8(255-(v/2*255)) (Detection.swift:83-85); Swift's UInt8(Double) truncates toward zero and
 * traps outside 0..255 — sigmoid outputs lie in [0,1] so the value is in [127.5, 255]. */
ORC_API void orc_mask_to_u8(const double* mask, int64_t n, uint8_t* out)
{
    for (int64_t i = 0; i < n; ++i) {
        double v = 255.0 - (mask[i] / 2.0 * 255.0);
        if (v < 0) v = 0;
        if (v > 255) v = 255;
        out[i] = (uint8_t)v;
    }
}
