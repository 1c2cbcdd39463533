// kernels_roialign.hip — PyramidROIAlignLayer on gfx950.
//
// Replaces (reference, Sources/Mask-RCNN-CoreML/PyramidROIAlignLayer.swift):
//   roisToInputItems        :351-396   FPN level selection in Double, validity
//   evaluate / performBatch / copyOutput :79-274   (89 MB texture upload per call, 64
//                           MPSNNCropAndResizeBilinear encodes per ROI, per-ROI read-back)
//   groupInputItemsByContent / batchInputGroups :399-498 are Metal scheduling artefacts and have
//                           no counterpart: one launch covers every ROI of every image.
//
// The sampler follows TensorFlow's crop_and_resize (bilinear, extrapolation value 0): the MPS
// kernel is closed source, its documented behaviour is that convention (unpinned).
// HBM-bound gather: with the engine's NHWC maps one wave reads 64 × 16 B = 1 KiB of consecutive
// channels per corner, so every access is a full-line coalesced load; the pyramid stays resident
// in HBM (no staging copy at all).  Compiled with -ffp-contract=off.
#include "device_math.h"
#include "kernels.h"

namespace mrcnn {

struct RoiGeom {
    int level;      // 0..3, -1 = padding ROI
    float y1, x1, y2, x2;
};

// roisToInputItems (:351-396).  ratio = 224 / sqrt(imageW*imageH), all in Double.
__device__ __forceinline__ RoiGeom roi_geom(const float* r, double ratio)
{
    RoiGeom g;
    g.y1 = r[0]; g.x1 = r[1]; g.y2 = r[2]; g.x2 = r[3];
    const double width = (double)g.x2 - (double)g.x1;
    const double height = (double)g.y2 - (double)g.y1;
    const double lf = log2(sqrt(width * height) / ratio) + 4.0;
    const bool valid = !isnan(lf) && !isinf(lf);
    int level = 2;
    if (valid) {
        double rr = round(lf);                       // Swift round(): half away from zero
        rr = rr < 2.0 ? 2.0 : (rr > 5.0 ? 5.0 : rr);
        level = (int)rr;
    }
    g.level = valid ? level - 2 : -1;
    return g;
}

struct Sample {
    bool ok;
    int t, b, l, r;
    float ly, lx;
};

__device__ __forceinline__ Sample make_sample(const RoiGeom& g, int H, int W, int P, int py, int px)
{
    Sample s;
    const float hs = (P > 1) ? (g.y2 - g.y1) * (float)(H - 1) / (float)(P - 1) : 0.0f;
    const float ws = (P > 1) ? (g.x2 - g.x1) * (float)(W - 1) / (float)(P - 1) : 0.0f;
    const float in_y = (P > 1) ? g.y1 * (float)(H - 1) + (float)py * hs : 0.5f * (g.y1 + g.y2) * (float)(H - 1);
    const float in_x = (P > 1) ? g.x1 * (float)(W - 1) + (float)px * ws : 0.5f * (g.x1 + g.x2) * (float)(W - 1);
    s.ok = !(in_y < 0 || in_y > (float)(H - 1)) && !(in_x < 0 || in_x > (float)(W - 1));
    const float fy = floorf(in_y), cy = ceilf(in_y), fx = floorf(in_x), cx = ceilf(in_x);
    s.ly = in_y - fy;
    s.lx = in_x - fx;
    s.t = (int)fy; s.b = (int)cy; s.l = (int)fx; s.r = (int)cx;
    return s;
}

__device__ __forceinline__ float bilerp(float tl, float tr, float bl, float br, float lx, float ly)
{
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    return top + (bot - top) * ly;
}

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<_Float16>(const _Float16* p)
{
    const h16x4 h = *reinterpret_cast<const h16x4*>(p);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<_Float16>(_Float16* p, float4 v)
{
    h16x4 h;
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;   // RNE
    *reinterpret_cast<h16x4*>(p) = h;
}

// NHWC: block = one ROI; thread = (channel quad, point group).  fp16 maps: the bilinear arithmetic
// is the same fp32 sequence on the widened samples, rounded once to fp16 on store.
// Round 5: the kernel was INSTRUCTION bound, not memory bound — every thread recomputed its point's sample (two float divisions, floor /
// ceil, the range test) and four run-time integer divisions per element: ~150 VALU instructions around four loads, 125 us for 200 MB
// of fp16 output.  Now the (at most 256) points of a pass are sampled ONCE, one per thread, into an LDS table (corner offsets, weights),
// and the element loop reads its point's entry (wave-uniform address when C >= 256: a broadcast) and indexes by shift / mask.
// The arithmetic per sample and per element is unchanged: the same bits (tests/test_gpu_layers.py against the oracle).
template <typename T>
__global__ __launch_bounds__(256) void k_roi_align_nhwc(PyramidMaps maps, int C, const float* __restrict__ rois,
                                                        long rois_sB, long roi_stride, int P, double ratio,
                                                        T* __restrict__ out, long out_sB, long out_row_stride,
                                                        int32_t* __restrict__ row_flags)
{
    __shared__ int4 s_off[256];        // element offsets of the four corners (top-left, top-right, bottom-left, bottom-right); x < 0: outside the map
    __shared__ float2 s_w[256];        // (lx, ly)
    const int roi = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const RoiGeom g = roi_geom(rois + (size_t)b * rois_sB + (size_t)roi * roi_stride, ratio);
    T* o = out + (size_t)b * out_sB + (size_t)roi * out_row_stride;
    const int C4 = C >> 2;
    const int npts = P * P;
    if (g.level < 0) {
        for (int e = t; e < npts * C4; e += 256) st4<T>(o + (size_t)e * 4, make_float4(0, 0, 0, 0));
        if (row_flags && t == 0) row_flags[(size_t)b * gridDim.x + roi] = 0;
        return;
    }
    // fp16 maps with C % 8 == 0: items of EIGHT channels (16-B loads and stores: half the vector-memory instructions of the gather)
    constexpr bool HALF = sizeof(T) == 2;
    const bool wide = HALF && (C & 7) == 0;
    const int CI = wide ? C >> 3 : C4;                 // items per point
    // removeZeros predicate of the mask layer (TimeDistributedClassifierLayer.swift:116-127: a row is kept iff every
    // element != 0), evaluated on the fp32 samples BEFORE the store rounds them — the decision the reference's fp32
    // pipeline takes; an fp16 store would flush |v| < 3e-8 to zero and drop a valid detection.
    int all_nonzero = 1;
    const int H = maps.H[g.level], W = maps.W[g.level];
    const T* m = static_cast<const T*>(maps.data[g.level]) + (size_t)b * maps.sB[g.level];
    const float mul = maps.mul[g.level];               // 1, or the exact power of two between this level's split exponent and the output's
    const bool pow2 = (CI & (CI - 1)) == 0;
    const int sh = 31 - __builtin_clz((unsigned)(CI > 0 ? CI : 1));
    for (int p0 = 0; p0 < npts; p0 += 256) {
        const int np = min(256, npts - p0);
        if (p0) __syncthreads();
        if (t < np) {
            const int pt = p0 + t;
            const int py = pt / P, px = pt - py * P;
            const Sample s = make_sample(g, H, W, P, py, px);
            s_off[t] = s.ok ? make_int4((s.t * W + s.l) * C, (s.t * W + s.r) * C, (s.b * W + s.l) * C, (s.b * W + s.r) * C) : make_int4(-1, 0, 0, 0);
            s_w[t] = make_float2(s.lx, s.ly);
        }
        __syncthreads();
        const int total = np * CI;
        if constexpr (HALF) {
            if (wide) {
                for (int e = t; e < total; e += 256) {
                    const int pl = pow2 ? e >> sh : e / CI;
                    const int c8 = pow2 ? e & (CI - 1) : e - pl * CI;
                    const int4 of = s_off[pl];
                    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (of.x >= 0) {
                        const float2 w = s_w[pl];
                        const T* mc = m + c8 * 8;
                        const f16x8_t tl = *reinterpret_cast<const f16x8_t*>(mc + of.x), tr = *reinterpret_cast<const f16x8_t*>(mc + of.y);
                        const f16x8_t bl = *reinterpret_cast<const f16x8_t*>(mc + of.z), br = *reinterpret_cast<const f16x8_t*>(mc + of.w);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = bilerp((float)tl[k], (float)tr[k], (float)bl[k], (float)br[k], w.x, w.y) * mul;
                    }
                    f16x8_t hv;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { all_nonzero &= v[k] != 0.0f ? 1 : 0; hv[k] = (_Float16)v[k]; }
                    *reinterpret_cast<f16x8_t*>(o + ((size_t)p0 * CI + e) * 8) = hv;
                }
                continue;
            }
        }
        for (int e = t; e < total; e += 256) {
            const int pl = pow2 ? e >> sh : e / C4;
            const int cq = pow2 ? e & (C4 - 1) : e - pl * C4;
            const int4 of = s_off[pl];
            float4 v = make_float4(0, 0, 0, 0);
            if (of.x >= 0) {
                const float2 w = s_w[pl];
                const T* mc = m + cq * 4;
                const float4 tl = ld4<T>(mc + of.x);
                const float4 tr = ld4<T>(mc + of.y);
                const float4 bl = ld4<T>(mc + of.z);
                const float4 br = ld4<T>(mc + of.w);
                v.x = bilerp(tl.x, tr.x, bl.x, br.x, w.x, w.y);
                v.y = bilerp(tl.y, tr.y, bl.y, br.y, w.x, w.y);
                v.z = bilerp(tl.z, tr.z, bl.z, br.z, w.x, w.y);
                v.w = bilerp(tl.w, tr.w, bl.w, br.w, w.x, w.y);
                v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
            }
            all_nonzero &= (v.x != 0.0f && v.y != 0.0f && v.z != 0.0f && v.w != 0.0f) ? 1 : 0;
            st4<T>(o + ((size_t)p0 * C4 + e) * 4, v);
        }
    }
    if (row_flags) {                                   // wave-uniform branch: kernel argument
        all_nonzero = __syncthreads_and(all_nonzero);
        if (t == 0) row_flags[(size_t)b * gridDim.x + roi] = all_nonzero;
    }
}

// NCHW (Core ML layout, used by the stand-alone custom layer): thread = one output element,
// px fastest so that stores are coalesced.
__global__ __launch_bounds__(256) void k_roi_align_nchw(PyramidMaps maps, int C, const float* __restrict__ rois,
                                                        long rois_sB, long roi_stride, int P, double ratio,
                                                        float* __restrict__ out, long out_sB, long out_row_stride)
{
    const int roi = blockIdx.x, b = blockIdx.y;
    const RoiGeom g = roi_geom(rois + (size_t)b * rois_sB + (size_t)roi * roi_stride, ratio);
    float* o = out + (size_t)b * out_sB + (size_t)roi * out_row_stride;
    const int total = C * P * P;
    if (g.level < 0) {
        for (int e = threadIdx.x; e < total; e += 256) o[e] = 0.0f;
        return;
    }
    const int H = maps.H[g.level], W = maps.W[g.level];
    const float* m = static_cast<const float*>(maps.data[g.level]) + (size_t)b * maps.sB[g.level];
    for (int e = threadIdx.x; e < total; e += 256) {
        const int px = e % P, py = (e / P) % P, c = e / (P * P);
        const Sample s = make_sample(g, H, W, P, py, px);
        float v = 0.0f;
        if (s.ok) {
            const float* mc = m + (size_t)c * H * W;
            v = bilerp(mc[(size_t)s.t * W + s.l], mc[(size_t)s.t * W + s.r], mc[(size_t)s.b * W + s.l],
                       mc[(size_t)s.b * W + s.r], s.lx, s.ly);
        }
        o[e] = v;
    }
}

void roi_align_forward(hipStream_t s, const PyramidMaps& maps, int C, int layout_nhwc, const float* rois,
                       long rois_sB, long roi_stride, int n_rois, int B, int pool, double image_w,
                       double image_h, void* out, long out_sB, long out_row_stride, int dtype, int32_t* row_flags)
{
    if (n_rois <= 0 || B <= 0) return;
    MRCNN_REQUIRE(!row_flags || layout_nhwc, MRCNN_ERR_INVALID, "ROIAlign: row flags are produced by the NHWC kernel only");
    const double ratio = 224.0 / sqrt(image_w * image_h);    // PyramidROIAlignLayer.swift:98,357
    if (layout_nhwc) {
        MRCNN_REQUIRE(C % 4 == 0, MRCNN_ERR_SHAPE, "ROIAlign: channel count %d not a multiple of 4", C);
        if (dtype == MRCNN_F16)
            hipLaunchKernelGGL(k_roi_align_nhwc<_Float16>, dim3(n_rois, B), dim3(256), 0, s, maps, C, rois, rois_sB, roi_stride, pool,
                               ratio, (_Float16*)out, out_sB, out_row_stride, row_flags);
        else
            hipLaunchKernelGGL(k_roi_align_nhwc<float>, dim3(n_rois, B), dim3(256), 0, s, maps, C, rois, rois_sB, roi_stride, pool,
                               ratio, (float*)out, out_sB, out_row_stride, row_flags);
    } else {
        MRCNN_REQUIRE(dtype == MRCNN_F32, MRCNN_ERR_UNSUPPORTED, "ROIAlign: the CHW layout is fp32 only");
        hipLaunchKernelGGL(k_roi_align_nchw, dim3(n_rois, B), dim3(256), 0, s, maps, C, rois, rois_sB, roi_stride, pool,
                           ratio, (float*)out, out_sB, out_row_stride);
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Mask paste: 28×28 sigmoid mask → full-resolution binary instance mask.
// The reference only does this when drawing (Example/Source/DetectionRenderer.swift:13-24 stretches
// the CGImage mask of Detection.swift:83-98 over the box with CoreGraphics); for mask AP the mask has
// to be resized to its box and thresholded.  Conventions (unpinned — CoreGraphics' resampler is
// closed): box pixels as in Matterport's denorm_boxes (round-half-even of y*(H-1), +1 on the far
// edge), bilinear resampling with half-pixel centres and edge clamp, `>= threshold`.
// HBM-bound: n×H×W bytes written once (16 B per lane); the 3 KB mask is read through L1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_paste_masks(const float* __restrict__ det, long det_stride,
                                                     const float* __restrict__ masks, int S, int H, int W, float thr,
                                                     uint8_t* __restrict__ out)
{
    const int inst = blockIdx.y;
    const float* d = det + (size_t)inst * det_stride;
    const float* m = masks + (size_t)inst * S * S;
    // denorm_boxes: around(box * (H-1, W-1, H-1, W-1) + (0, 0, 1, 1))
    const int y1 = (int)rint((double)d[0] * (double)(H - 1));
    const int x1 = (int)rint((double)d[1] * (double)(W - 1));
    const int y2 = (int)rint((double)d[2] * (double)(H - 1) + 1.0);
    const int x2 = (int)rint((double)d[3] * (double)(W - 1) + 1.0);
    const int bh = y2 - y1, bw = x2 - x1;
    const bool empty = bh <= 0 || bw <= 0 || !(d[5] > 0.0f);
    const float sy_scale = empty ? 0.f : (float)S / (float)bh;
    const float sx_scale = empty ? 0.f : (float)S / (float)bw;
    uint8_t* o = out + (size_t)inst * H * W;
    const long quads = (long)H * W / 4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < quads; e += (long)gridDim.x * 256) {
        const long p0 = e * 4;
        const int y = (int)(p0 / W), xb = (int)(p0 - (long)y * W);
        uint32_t packed = 0;
        if (!empty && y >= y1 && y < y2) {
            float sy = ((float)(y - y1) + 0.5f) * sy_scale - 0.5f;
            sy = fminf(fmaxf(sy, 0.0f), (float)(S - 1));
            const int ya = (int)floorf(sy), yb = min(ya + 1, S - 1);
            const float fy = sy - (float)ya;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = xb + k;
                if (x < x1 || x >= x2) continue;
                float sx = ((float)(x - x1) + 0.5f) * sx_scale - 0.5f;
                sx = fminf(fmaxf(sx, 0.0f), (float)(S - 1));
                const int xa = (int)floorf(sx), xc = min(xa + 1, S - 1);
                const float fx = sx - (float)xa;
                const float a = m[ya * S + xa], b = m[ya * S + xc], c = m[yb * S + xa], dd = m[yb * S + xc];
                const float top = a + (b - a) * fx;
                const float bot = c + (dd - c) * fx;
                const float v = top + (bot - top) * fy;
                if (v >= thr) packed |= 1u << (8 * k);
            }
        }
        reinterpret_cast<uint32_t*>(o)[e] = packed;
    }
}

void paste_masks_forward(hipStream_t s, const float* det, long det_stride, const float* masks, int n, int S, int H, int W,
                         float thr, uint8_t* out)
{
    if (n <= 0) return;
    MRCNN_REQUIRE(W % 4 == 0, MRCNN_ERR_SHAPE, "paste_masks: image width %d not a multiple of 4", W);
    const long quads = (long)H * W / 4;
    const int gx = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_paste_masks, dim3(gx, n), dim3(256), 0, s, det, det_stride, masks, S, H, W, thr, out);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Letterbox: the host's `.scaleFit` step (VNCoreMLRequest.imageCropAndScaleOption,
// EvaluateCommand.swift:157, ViewController.swift:45) on the GPU: aspect-preserving bilinear resize
// (half-pixel centres, edge clamp, round-half-up to 8 bit) centred in an H×W canvas, black borders.
// Vision's resampler is closed source → convention unpinned.
// ------------------------------------------------------------------------------------------------
// One letterboxed pixel (the arithmetic k_letterbox and the fused pre-processing share: the two must agree to the bit)
__device__ __forceinline__ void letterbox_pixel(const uint8_t* __restrict__ src, int h, int w, int nh, int nw, float ry, float rx, int y, int x,
                                                uint8_t (&r)[3])
{
    r[0] = r[1] = r[2] = 0;
    if ((unsigned)y < (unsigned)nh && (unsigned)x < (unsigned)nw) {
        float sy = ((float)y + 0.5f) * ry - 0.5f, sx = ((float)x + 0.5f) * rx - 0.5f;
        sy = fminf(fmaxf(sy, 0.0f), (float)(h - 1));
        sx = fminf(fmaxf(sx, 0.0f), (float)(w - 1));
        const int ya = (int)floorf(sy), yb = min(ya + 1, h - 1), xa = (int)floorf(sx), xb = min(xa + 1, w - 1);
        const float fy = sy - (float)ya, fx = sx - (float)xa;
        const uint8_t* a = src + ((long)ya * w + xa) * 3;
        const uint8_t* b = src + ((long)ya * w + xb) * 3;
        const uint8_t* c = src + ((long)yb * w + xa) * 3;
        const uint8_t* d = src + ((long)yb * w + xb) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float top = (float)a[k] + ((float)b[k] - (float)a[k]) * fx;
            const float bot = (float)c[k] + ((float)d[k] - (float)c[k]) * fx;
            const float v = top + (bot - top) * fy;
            r[k] = (uint8_t)floorf(v + 0.5f);
        }
    }
}

__global__ __launch_bounds__(256) void k_letterbox(const uint8_t* __restrict__ src, int h, int w, uint8_t* __restrict__ dst,
                                                   int H, int W, int nh, int nw, int py, int px)
{
    const long total = (long)H * W;
    const float ry = (float)h / (float)nh, rx = (float)w / (float)nw;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int Y = (int)(e / W), X = (int)(e - (long)Y * W);
        uint8_t r[3];
        letterbox_pixel(src, h, w, nh, nw, ry, rx, Y - py, X - px, r);
        dst[e * 3 + 0] = r[0]; dst[e * 3 + 1] = r[1]; dst[e * 3 + 2] = r[2];
    }
}

// `.scaleFit` fused into the network's input staging (EvaluateCommand.swift:152-157): source images of any
// size h×w → the letterboxed 8-bit value (exactly k_letterbox's) minus the channel mean, into the zero-padded NHWC4 (fp32) /
// NHWC8 (fp16) tensor the stem reads — the H×W×3 letterboxed image is never materialised.
template <typename T>
__global__ __launch_bounds__(256) void k_preprocess_scalefit(const uint8_t* __restrict__ src, int B, int h, int w, int H, int W, int nh, int nw,
                                                             int py, int px, int pad, float mr, float mg, float mb, void* __restrict__ out)
{
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long total = (long)B * Hp * Wp;
    const float ry = (float)h / (float)nh, rx = (float)w / (float)nw;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int X = (int)(e % Wp) - pad;
        const int Y = (int)((e / Wp) % Hp) - pad;
        const int b = (int)(e / ((long)Wp * Hp));
        float r = 0.f, g = 0.f, bl = 0.f;
        if ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) {
            uint8_t v[3];
            letterbox_pixel(src + (long)b * h * w * 3, h, w, nh, nw, ry, rx, Y - py, X - px, v);
            r = (float)v[0] - mr; g = (float)v[1] - mg; bl = (float)v[2] - mb;
        }
        if constexpr (sizeof(T) == 4) {
            reinterpret_cast<float4*>(out)[e] = make_float4(r, g, bl, 0.f);
        } else {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 hv;
            hv[0] = (_Float16)r; hv[1] = (_Float16)g; hv[2] = (_Float16)bl;
            hv[3] = hv[4] = hv[5] = hv[6] = hv[7] = (_Float16)0.f;
            reinterpret_cast<h8*>(out)[e] = hv;
        }
    }
}

void preprocess_scalefit_forward(hipStream_t s, const uint8_t* src, int B, int h, int w, int H, int W, int nh, int nw, int py, int px, int pad,
                                 const float mean[3], void* out, int dtype)
{
    const long total = (long)B * (H + 2 * pad) * (W + 2 * pad);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == MRCNN_F16)
        hipLaunchKernelGGL(k_preprocess_scalefit<_Float16>, dim3(grid), dim3(256), 0, s, src, B, h, w, H, W, nh, nw, py, px, pad, mean[0], mean[1], mean[2], out);
    else
        hipLaunchKernelGGL(k_preprocess_scalefit<float>, dim3(grid), dim3(256), 0, s, src, B, h, w, H, W, nh, nw, py, px, pad, mean[0], mean[1], mean[2], out);
    HIP_CHECK(hipGetLastError());
}

void letterbox_forward(hipStream_t s, const uint8_t* src, int h, int w, uint8_t* dst, int H, int W, int nh, int nw, int py, int px)
{
    const long total = (long)H * W;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_letterbox, dim3(grid), dim3(256), 0, s, src, h, w, dst, H, W, nh, nw, py, px);
    HIP_CHECK(hipGetLastError());
}

}  // namespace mrcnn
