// device_math.h — bit-reproducible box arithmetic for the gfx950 kernels.
//
// Compile every translation unit that includes this with -ffp-contract=off: the reference's
// Swift code does not contract a*b+c (BoxUtils.swift:50-66) and the results must be bit-identical
// to the scalar restatement the parity tests compare against.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrcnn {

// exp on Float (BoxUtils.swift:57-58 `exp(deltaY2)`): Cody-Waite reduction and a degree-13 Horner
// polynomial in double with explicit fma, scaled by an exactly constructed power of two and
// rounded once to float.  Only IEEE basic operations, hence the same bits on any conforming
// machine; equals a correctly rounded expf except when the double result falls within 2^-29
// (relative) of a float rounding boundary.
__device__ __forceinline__ float box_expf(float xf)
{
    if (xf != xf) return xf;
    if (xf > 88.72284f) return __builtin_inff();
    if (xf < -104.0f) return 0.0f;
    const double x = (double)xf;
    const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(n, -0x1.62e42fefa38p-1, x);
    r = __builtin_fma(n, -0x1.ef35793c7673p-45, r);
    double p = 0x1.6124613a86d09p-33;
    p = __builtin_fma(p, r, 0x1.1eed8eff8d898p-29);
    p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);
    p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);
    p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const long long e = (long long)n + 1023;
    const double s = __longlong_as_double(e << 52);
    return (float)(p * s);
}

// applyBoxDeltas (BoxUtils.swift:32-71) on one box, deltas already multiplied by the std-dev
// (ProposalLayer.swift:156-158), followed by clip to [0,1] (BoxUtils.swift:73-80).
__device__ __forceinline__ float4 decode_clip_box(float4 box, float4 d)
{
    const float y1 = box.x, x1 = box.y, y2 = box.z, x2 = box.w;
    float height = y2 - y1;
    float width = x2 - x1;
    float cy = y1 + 0.5f * height;
    float cx = x1 + 0.5f * width;
    cy = cy + d.x * height;
    cx = cx + d.y * width;
    height = height * box_expf(d.z);
    width = width * box_expf(d.w);
    float ry1 = cy - 0.5f * height;
    float rx1 = cx - 0.5f * width;
    float ry2 = ry1 + height;
    float rx2 = rx1 + width;
    ry1 = fminf(fmaxf(ry1, 0.0f), 1.0f);
    rx1 = fminf(fmaxf(rx1, 0.0f), 1.0f);
    ry2 = fminf(fmaxf(ry2, 0.0f), 1.0f);
    rx2 = fminf(fmaxf(rx2, 0.0f), 1.0f);
    return make_float4(ry1, rx1, ry2, rx2);
}

// CGRect(anchorDatum:) + IOU (Utils.swift:220-246): Double arithmetic on the standardized rect
// (CGRect.width/.minX/... are the CGRectGet* accessors), result rounded to Float.
struct RectD {
    double x, y, w, h;
};
__device__ __forceinline__ RectD rect_from(float4 b)   // b = (y1,x1,y2,x2)
{
    RectD r;
    r.x = (double)b.y;
    r.y = (double)b.x;
    r.w = (double)b.w - (double)b.y;
    r.h = (double)b.z - (double)b.x;
    return r;
}
__device__ __forceinline__ bool rect_selectable(float4 b)   // width > 0 && height > 0 (Utils.swift:194)
{
    RectD r = rect_from(b);
    return fabs(r.w) > 0 && fabs(r.h) > 0;
}
__device__ __forceinline__ float iou_yxyx(float4 fa, float4 fb)
{
    const RectD a = rect_from(fa), b = rect_from(fb);
    const double areaA = fabs(a.w) * fabs(a.h);
    if (areaA <= 0) return 0.0f;
    const double areaB = fabs(b.w) * fabs(b.h);
    if (areaB <= 0) return 0.0f;
    const double aminx = a.w < 0 ? a.x + a.w : a.x, amaxx = a.w < 0 ? a.x : a.x + a.w;
    const double aminy = a.h < 0 ? a.y + a.h : a.y, amaxy = a.h < 0 ? a.y : a.y + a.h;
    const double bminx = b.w < 0 ? b.x + b.w : b.x, bmaxx = b.w < 0 ? b.x : b.x + b.w;
    const double bminy = b.h < 0 ? b.y + b.h : b.y, bmaxy = b.h < 0 ? b.y : b.y + b.h;
    const double ix0 = fmax(aminx, bminx), iy0 = fmax(aminy, bminy);
    const double ix1 = fmin(amaxx, bmaxx), iy1 = fmin(amaxy, bmaxy);
    const double inter = fmax(iy1 - iy0, 0.0) * fmax(ix1 - ix0, 0.0);
    return (float)(inter / (areaA + areaB - inter));
}

// iou_yxyx(fa, fb) > thr without the fp64 division in all but the borderline cases (the NMS bit-matrix evaluates it for
// every overlapping pair).  The reference's decision is  Float(inter / union) > thr  (Utils.swift:236-246): the quotient is
// rounded to double, then to float, both monotone — so the decision is "quotient above T", T = the midpoint of thr and the
// next float above it, except within a few ulps of T.  inter is compared with T·union under a relative margin of 2^-48
// (the three products round with 2^-53 each); inside the margin the original expression decides.  thr_mid < 0 (a negative
// or non-finite threshold): always the original expression.
__device__ __forceinline__ double iou_threshold_midpoint(float thr)
{
    if (!(thr >= 0.0f) || !(thr < 3.0e38f)) return -1.0;
    return 0.5 * ((double)thr + (double)__uint_as_float(__float_as_uint(thr) + 1u));
}
// The per-box half of the IoU (standardized CGRect extents and area, Utils.swift:220-235), computed once per box.
struct RectExt {
    double minx, maxx, miny, maxy, area;
};
__device__ __forceinline__ RectExt rect_ext(float4 f)
{
    const RectD a = rect_from(f);
    RectExt e;
    e.area = fabs(a.w) * fabs(a.h);
    e.minx = a.w < 0 ? a.x + a.w : a.x; e.maxx = a.w < 0 ? a.x : a.x + a.w;
    e.miny = a.h < 0 ? a.y + a.h : a.y; e.maxy = a.h < 0 ? a.y : a.y + a.h;
    return e;
}
__device__ __forceinline__ bool iou_exceeds(const RectExt& a, const RectExt& b, float thr, double thr_mid)
{
    const double areaA = a.area, areaB = b.area;
    if (areaA <= 0) return 0.0f > thr;
    if (areaB <= 0) return 0.0f > thr;
    const double ix0 = fmax(a.minx, b.minx), iy0 = fmax(a.miny, b.miny);
    const double ix1 = fmin(a.maxx, b.maxx), iy1 = fmin(a.maxy, b.maxy);
    const double inter = fmax(iy1 - iy0, 0.0) * fmax(ix1 - ix0, 0.0);
    const double uni = areaA + areaB - inter;
    if (thr_mid >= 0.0) {
        const double p = thr_mid * uni;
        if (inter > p * (1.0 + 0x1p-48)) return true;
        if (inter < p * (1.0 - 0x1p-48)) return false;
    }
    return (float)(inter / uni) > thr;
}

// Monotone float → uint32 key (ascending key == ascending float); -0 is folded onto +0 so that it
// ties with +0 as in a float comparison.
__device__ __forceinline__ uint32_t order_key(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace mrcnn
