// conv_device.h — device-side pieces shared by the convolution kernels (kernels_conv.hip: the 128-row tiles of every
// compute mode; kernels_conv_pp.hip: the 256-row ping-pong fp16 tiles): launch arguments, 16-B load/store helpers and the
// fused epilogue.  Private to libmaskrcnn_hip.so.
#pragma once
#include <type_traits>
#include "kernels.h"

namespace mrcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const void* in; const void* wgt; const float* scale; const float* shift; const void* res;
    void* out; void* out2;
    long in_sB, in_sH, in_sW;
    long res_sB, res_sH, res_sW;
    long out_sB, out_sP, out_sH, out_sW;
    long out2_sB, out2_sP;
    int B, H, W, Cin, KH, KW, stride, padH, padW;
    int OH, OW, Cout, ncols, Ktot, M;
    int res_shift, act, n_split, deconv2;
    int tiles_m, tiles_n;
    int vec_ok;          // epilogue may use vector stores / residual loads
    int out_f32;         // store fp32 even when the activations are fp16 (RPN outputs, class logits, masks)
    int* range_flag;         // optional: set to 1 when an output leaves the fp16 range (|v| >= 65504 or NaN)
    int dbg;
    int direct;              // the layer's epilogue can go straight from the accumulators (conv_epilogue_direct)
    const float* sel_w; const int32_t* sel_cid; float* sel_partial;      // deconv2: selected-class dot instead of the store (ConvDesc)                 // ablation switches of the ping-pong kernels (measurement only; 0 in production)
    // canonical K chunks (kernels_conv.hip: conv_k_chunks): the layer's sum is ((0 + P0) + P1) + ... over kchunks equal runs of K
    // steps, each Pc a running accumulator from zero — by one block (ksplit == 1: two accumulator sets) or by kchunks blocks per
    // tile (ksplit == kchunks: partial sums through ks_scratch, the last block to arrive folds them in that order)
    int kchunks, ksplit;
    float* ks_scratch; unsigned* ks_count;
    // fused shortcut (kernels_conv.hip: conv_forward's `sc`): the 1x1 convolution whose output this layer would have read as its residual —
    // its input tensor (same batch, sampled with sc_stride), filters and folded BatchNorm; the residual fields above are then unused
    const void* sc_in; const void* sc_wgt; const float* sc_scale; const float* sc_shift;
    long sc_in_sB, sc_in_sH, sc_in_sW;
    int sc_H, sc_W, sc_Cin, sc_stride;
    int sel_part_cols;       // selected-class mode: output columns per partial sum (128: the block-staged form; 64: conv_epilogue_sel_wave)
};

static constexpr int BM_DEFAULT = 128;   // rows of the block tile = WM*TM*32

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPV = 4; };        // elements per 16-B vector
template <> struct Elem<_Float16> { static constexpr int EPV = 8; };

template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load4<_Float16>(const _Float16* p)
{
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void store4<_Float16>(_Float16* p, float4 v)
{
    f16x4 h;
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
    *reinterpret_cast<f16x4*>(p) = h;
}

// 8 consecutive fp16 values as one 16-B access (the fp16 epilogue moves 8 columns per thread)
__device__ __forceinline__ void load8h(const _Float16* p, float4& lo, float4& hi)
{
    const f16x8 h = *reinterpret_cast<const f16x8*>(p);
    lo = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    hi = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
}
__device__ __forceinline__ void store8h(_Float16* p, const float4 lo, const float4 hi)
{
    f16x8 h;
    h[0] = (_Float16)lo.x; h[1] = (_Float16)lo.y; h[2] = (_Float16)lo.z; h[3] = (_Float16)lo.w;
    h[4] = (_Float16)hi.x; h[5] = (_Float16)hi.y; h[6] = (_Float16)hi.z; h[7] = (_Float16)hi.w;
    *reinterpret_cast<f16x8*>(p) = h;
}

// The kernels issue their MFMAs with the FILTER fragment as first operand: a 32×32 result tile is held transposed — lane
// (l31, kk) owns output PIXEL l31 of the tile and channels 8q + 4kk + r (q, r = 0..3), i.e. element e = 4q + r of the
// accumulator vector: runs of four consecutive channels of one pixel.
//
// conv_epilogue (general path): accumulators → LDS (fp32 C tile, rows padded by 16 B so that the 16-B run writes of a
// half-wave — one pixel row per lane — fall in distinct banks) → full-row vector stores with fused scale/shift (BN + bias),
// residual, activation, column split / 2×2 scatter.  conv_epilogue_direct (below): the common case without the LDS round trip.
// CPASS = 1: the whole BM×BN tile is staged at once; CPASS = WN (tiles whose fp32 C tile would not fit
// beside a second block: 128×256): one pass per wave column, BM × TN·32 columns each.
template <typename T, int BN, int TM, int TN, int WM, int WN, int CPASS>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], unsigned char* smem, int m0, int n0)
{
    static_assert(CPASS == 1 || CPASS == WN, "column passes");
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int CW = BN / CPASS;     // columns staged per pass
    constexpr int CWP = CW + 4;        // row length of the LDS C tile (floats)
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    // ---- accumulators → LDS → full-row vector stores ---------------------------------------------
    // (the loop's final barrier guarantees nobody still reads the operand buffers)
    constexpr int CPT = sizeof(T) == 2 ? 8 : 4;   // columns per thread: 16 B of the activation type
    constexpr int NV = CPT / 4;                   // float4 groups per thread
    constexpr int TPR = CW / CPT;     // threads per output row
    constexpr int RPP = NT / TPR;     // rows per pass
    constexpr int NPASS = BM / RPP;
    const int c4 = t % TPR, rr = t / TPR;
    const bool dense_out = a.out_sB == (long)ohw * a.out_sP;
    const bool dense_res = a.res_sB == (long)ohw * a.res_sW && a.res_shift == 0;
    const bool need_bp = !dense_out || a.out2 != nullptr || a.deconv2 || (a.res && !dense_res);
    const bool need_yx = a.deconv2 || (a.res && a.res_shift);
    const T* const res = static_cast<const T*>(a.res);
    float* const Cs = reinterpret_cast<float*>(smem);
    bool out_of_range = false;       // fp16-range watch for the modes whose next layer reads this output through fp16

#pragma unroll
    for (int h = 0; h < CPASS; ++h) {
        const int n = n0 + h * CW + c4 * CPT;
        const bool col_ok = n < a.ncols;
        // Residual / scale / shift are fetched BEFORE the accumulators are staged through LDS: `res` and
        // `out` may alias as far as the compiler knows, so inside the store loop every residual load would
        // wait behind the previous store (16 serialized HBM round trips per thread on the branch2c layers).
        float4 rv[NPASS][NV];
        float4 sc[NV], sh[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) { sc[q] = make_float4(1.f, 1.f, 1.f, 1.f); sh[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (a.vec_ok && col_ok) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (a.scale) sc[q] = *reinterpret_cast<const float4*>(a.scale + n + 4 * q);
                if (a.shift) sh[q] = *reinterpret_cast<const float4*>(a.shift + n + 4 * q);
            }
#ifdef MRCNN_CONV_ABLATE
            if (res && !(a.dbg & 2048)) {
#else
            if (res) {
#endif
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int m = m0 + rr + ps * RPP;
#pragma unroll
                    for (int q = 0; q < NV; ++q) rv[ps][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m < a.M) {
                        long ro;
                        if (dense_res) ro = (long)m * a.res_sW;
                        else {
                            const int b = m / ohw, pix = m - b * ohw;
                            if (a.res_shift) {
                                const int oh = pix / a.OW, ow = pix - oh * a.OW;
                                ro = (long)b * a.res_sB + (long)(oh >> a.res_shift) * a.res_sH + (long)(ow >> a.res_shift) * a.res_sW;
                            } else ro = (long)b * a.res_sB + (long)pix * a.res_sW;
                        }
                        if constexpr (CPT == 8) load8h(reinterpret_cast<const _Float16*>(res) + ro + n, rv[ps][0], rv[ps][NV - 1]);
                        else rv[ps][0] = load4<T>(res + ro + n);
                    }
                }
            }
        }

        if (h > 0) __syncthreads();            // the previous pass has been read out of the C tile
        if (CPASS == 1 || wn == h) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = wm * TM * 32 + i * 32 + l31;                                              // pixel
                        const int col = (CPASS == 1 ? wn * TN * 32 : 0) + j * 32 + 8 * q + 4 * kk;                // channel run
                        *reinterpret_cast<float4*>(&Cs[row * CWP + col]) =
                            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    }
        }
        __syncthreads();

        if (!col_ok) continue;
        if (a.vec_ok) {
            const int qd = a.deconv2 ? n / a.Cout : 0;
            const int co = a.deconv2 ? n - qd * a.Cout : n;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int r = rr + ps * RPP;
                const int m = m0 + r;
                if (m >= a.M) continue;          // (not `break`: an early exit keeps the loop rolled and sends rv[] to scratch memory)
                int b = 0, pix = m, oh = 0, ow = 0;
                if (need_bp) { b = m / ohw; pix = m - b * ohw; }
                if (need_yx) { oh = pix / a.OW; ow = pix - oh * a.OW; }
                float4 v[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    float4 x = *reinterpret_cast<const float4*>(&Cs[r * CWP + c4 * CPT + 4 * q]);
                    x.x = x.x * sc[q].x + sh[q].x; x.y = x.y * sc[q].y + sh[q].y; x.z = x.z * sc[q].z + sh[q].z; x.w = x.w * sc[q].w + sh[q].w;
                    if (res) { x.x += rv[ps][q].x; x.y += rv[ps][q].y; x.z += rv[ps][q].z; x.w += rv[ps][q].w; }
                    if (a.act == ACT_RELU) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
                    else if (a.act == ACT_SIGMOID) {
                        x.x = 1.0f / (1.0f + expf(-x.x)); x.y = 1.0f / (1.0f + expf(-x.y));
                        x.z = 1.0f / (1.0f + expf(-x.z)); x.w = 1.0f / (1.0f + expf(-x.w));
                    }
                    out_of_range = out_of_range || !(fabsf(x.x) < 65504.0f) || !(fabsf(x.y) < 65504.0f) || !(fabsf(x.z) < 65504.0f) || !(fabsf(x.w) < 65504.0f);
                    v[q] = x;
                }
                if (a.sel_partial) {
                    // selected-class dot over this thread's channels, then over the TPR threads of the row (a 128-channel part of
                    // one output pixel); fixed order: identical for every batch size
                    const int cid = a.sel_cid[b];
                    float dot = 0.f;
                    if (cid >= 0) {
                        const float* wr = a.sel_w + (size_t)cid * a.Cout + co;
#pragma unroll
                        for (int q = 0; q < NV; ++q) {
                            float4 x = v[q];
                            if constexpr (sizeof(T) == 2) {          // the value the fp16 tensor would have held
                                x.x = (float)(_Float16)x.x; x.y = (float)(_Float16)x.y; x.z = (float)(_Float16)x.z; x.w = (float)(_Float16)x.w;
                            }
                            const float4 y = *reinterpret_cast<const float4*>(wr + 4 * q);
                            dot += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
                        }
                    }
#pragma unroll
                    for (int o = TPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, TPR);
                    if (c4 == 0 && cid >= 0) {
                        const long P = (long)(2 * oh + (qd >> 1)) * (2 * a.OW) + (2 * ow + (qd & 1));
                        a.sel_partial[((long)b * 4 * ohw + P) * (a.Cout / 128) + co / 128] = dot;
                    }
                    continue;
                }
                long o;
                if (a.deconv2) o = (long)b * a.out_sB + (long)(2 * oh + (qd >> 1)) * a.out_sH + (long)(2 * ow + (qd & 1)) * a.out_sW + co;
                else o = (dense_out ? (long)m * a.out_sP : (long)b * a.out_sB + (long)pix * a.out_sP) + n;
#ifdef MRCNN_CONV_ABLATE
                if ((a.dbg & 4096) && v[0].x != 1.2345e33f) continue;
#endif
                if (a.out_f32) {
#pragma unroll
                    for (int q = 0; q < NV; ++q) store4<float>(static_cast<float*>(a.out) + o + 4 * q, v[q]);
                } else if constexpr (CPT == 8) store8h(reinterpret_cast<_Float16*>(a.out) + o, v[0], v[NV - 1]);
                else store4<T>(static_cast<T*>(a.out) + o, v[0]);
            }
        } else {
            for (int r = rr; r < BM; r += RPP) {
                const int m = m0 + r;
                if (m >= a.M) break;
                const int b = m / ohw, pix = m - b * ohw;
                const int oh = pix / a.OW, ow = pix - oh * a.OW;
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    const int nn = n + c;
                    if (nn >= a.ncols) break;
                    float v = Cs[r * CWP + c4 * CPT + c];
                    v = v * (a.scale ? a.scale[nn] : 1.0f) + (a.shift ? a.shift[nn] : 0.0f);
                    if (res) {
                        long ro;
                        if (a.res_shift) ro = (long)b * a.res_sB + (long)(oh >> a.res_shift) * a.res_sH + (long)(ow >> a.res_shift) * a.res_sW;
                        else ro = (long)b * a.res_sB + (long)pix * a.res_sW;
                        v += (float)res[ro + nn];
                    }
                    if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (a.act == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    out_of_range |= !(fabsf(v) < 65504.0f);
                    long o;
                    void* dst = a.out;
                    if (a.deconv2) {
                        const int qd = nn / a.Cout, co = nn - qd * a.Cout;
                        o = (long)b * a.out_sB + (long)(2 * oh + (qd >> 1)) * a.out_sH + (long)(2 * ow + (qd & 1)) * a.out_sW + co;
                    } else if (a.out2 && nn >= a.n_split) {
                        dst = a.out2;
                        o = (long)b * a.out2_sB + (long)pix * a.out2_sP + (nn - a.n_split);
                    } else {
                        o = (long)b * a.out_sB + (long)pix * a.out_sP + nn;
                    }
                    if (a.out_f32) static_cast<float*>(dst)[o] = v;
                    else static_cast<T*>(dst)[o] = (T)v;
                }
            }
        }
    }
    if (a.range_flag && out_of_range) atomicOr(a.range_flag, 1);
}

// ----------------------------------------------------------------------------------------------------------------
// conv_epilogue_direct: the common epilogue (vectorisable output, no column split, no 2×2 scatter, no sigmoid) straight
// from the transposed accumulators — no LDS staging, no barrier.  Per run of four channels: fused scale/shift (+ residual)
// + ReLU in fp32 in exactly the order conv_epilogue applies them (bit-identical results).  fp32 tensors: one 16-B store per
// run.  fp16 tensors: one rounding, then per pair of runs (q = 2p, 2p+1) a v_permlane32_swap between the half-waves glues
// the pieces into 16 contiguous bytes per lane (cdna guide T21): a wave writes 64-B pieces of a pixel's channel line in
// back-to-back stores.  Short-K layers (the bottleneck blocks' 1×1 convolutions) spend most of their time here: against the
// LDS-staged form this is ≈ 2.5× fewer instructions per output (measured: DESIGN.md §3.1).
//   tab: scale[BN] | shift[BN] of the block's columns in LDS (written by the kernel before its first barrier);
//   row0 = first pixel row of the wave tile (absolute), colrel0 = its first column relative to the block tile.
// The residual of the whole wave tile is requested before the first store (`res` and `out` may alias as far as the
// compiler knows: a load behind a store waits for it).
// ----------------------------------------------------------------------------------------------------------------
template <typename T, int BN, int TMS, int TNS>
__device__ __forceinline__ void conv_epilogue_direct(const ConvArgs& a, f32x16 (&acc)[TMS][TNS], const float* tab, int row0, int n0, int colrel0, int lane)
{
    using Raw = std::conditional_t<sizeof(T) == 4, float4, uint2>;          // four residual elements as loaded
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const T* const res = static_cast<const T*>(a.res);
    T* const out = static_cast<T*>(a.out);
    const bool dense_out = a.out_sB == (long)ohw * a.out_sP;
    const bool dense_res = a.res_sB == (long)ohw * a.res_sW && a.res_shift == 0;
    const bool relu = a.act == ACT_RELU;
    bool out_of_range = false;
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
        const int m = row0 + i * 32 + l31;
        const bool ok_mi = m < a.M;
        long o_row = (long)m * a.out_sP, r_row = (long)m * a.res_sW;
        if (!dense_out || (res && !dense_res)) {
            const int mm = ok_mi ? m : 0;
            const int b = mm / ohw, pix = mm - b * ohw;
            o_row = (long)b * a.out_sB + (long)pix * a.out_sP;
            if (res) {
                if (a.res_shift) {
                    const int oh = pix / a.OW, ow = pix - oh * a.OW;
                    r_row = (long)b * a.res_sB + (long)(oh >> a.res_shift) * a.res_sH + (long)(ow >> a.res_shift) * a.res_sW;
                } else r_row = (long)b * a.res_sB + (long)pix * a.res_sW;
            }
        }
        Raw rr[TNS][2][2];
        if (res) {
#pragma unroll
            for (int j = 0; j < TNS; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int ca = n0 + colrel0 + j * 32 + 16 * p + 4 * kk, cb = ca + 8;
                    rr[j][p][0] = Raw{}; rr[j][p][1] = Raw{};
                    if (ok_mi && ca < a.ncols) rr[j][p][0] = *reinterpret_cast<const Raw*>(res + r_row + ca);
                    if (ok_mi && cb < a.ncols) rr[j][p][1] = *reinterpret_cast<const Raw*>(res + r_row + cb);
                }
        }
#pragma unroll
        for (int j = 0; j < TNS; ++j) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cl = colrel0 + j * 32 + 16 * p + 4 * kk;          // column inside the block tile
                const int ca = n0 + cl, cb = ca + 8;
                const float4 sa = *reinterpret_cast<const float4*>(tab + cl), sb_ = *reinterpret_cast<const float4*>(tab + cl + 8);
                const float4 ha = *reinterpret_cast<const float4*>(tab + BN + cl), hb = *reinterpret_cast<const float4*>(tab + BN + cl + 8);
                float4 va = make_float4(acc[i][j][8 * p + 0], acc[i][j][8 * p + 1], acc[i][j][8 * p + 2], acc[i][j][8 * p + 3]);
                float4 vb = make_float4(acc[i][j][8 * p + 4], acc[i][j][8 * p + 5], acc[i][j][8 * p + 6], acc[i][j][8 * p + 7]);
                va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                const bool ok_a = ok_mi && ca < a.ncols, ok_b = ok_mi && cb < a.ncols;
                if (res) {
                    float4 ra, rb;
                    if constexpr (sizeof(T) == 4) { ra = rr[j][p][0]; rb = rr[j][p][1]; }
                    else {
                        const f16x4 h0 = __builtin_bit_cast(f16x4, rr[j][p][0]), h1 = __builtin_bit_cast(f16x4, rr[j][p][1]);
                        ra = make_float4((float)h0[0], (float)h0[1], (float)h0[2], (float)h0[3]);
                        rb = make_float4((float)h1[0], (float)h1[1], (float)h1[2], (float)h1[3]);
                    }
                    va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
                    vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
                }
                if (relu) {
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                    vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                }
                // fp16-range watchdog (|v| >= 65504, inf or NaN)
                if (ok_a) out_of_range = out_of_range || !(fabsf(va.x) < 65504.0f) || !(fabsf(va.y) < 65504.0f) || !(fabsf(va.z) < 65504.0f) || !(fabsf(va.w) < 65504.0f);
                if (ok_b) out_of_range = out_of_range || !(fabsf(vb.x) < 65504.0f) || !(fabsf(vb.y) < 65504.0f) || !(fabsf(vb.z) < 65504.0f) || !(fabsf(vb.w) < 65504.0f);
                if constexpr (sizeof(T) == 4) {
                    if (ok_a) *reinterpret_cast<float4*>(out + o_row + ca) = va;
                    if (ok_b) *reinterpret_cast<float4*>(out + o_row + cb) = vb;
                } else {
                    f16x4 ha4, hb4;
                    ha4[0] = (_Float16)va.x; ha4[1] = (_Float16)va.y; ha4[2] = (_Float16)va.z; ha4[3] = (_Float16)va.w;
                    hb4[0] = (_Float16)vb.x; hb4[1] = (_Float16)vb.y; hb4[2] = (_Float16)vb.z; hb4[3] = (_Float16)vb.w;
                    const uint2 pa = __builtin_bit_cast(uint2, ha4), pb = __builtin_bit_cast(uint2, hb4);
                    // upper half-wave's q = 2p runs <-> lower half-wave's q = 2p+1 runs
                    const auto sx = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
                    const int n_store = n0 + colrel0 + j * 32 + 16 * p + 8 * kk;
                    if (ok_mi && n_store < a.ncols)
                        *reinterpret_cast<uint4*>(out + o_row + n_store) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
        }
    }
    if (a.range_flag && out_of_range) atomicOr(a.range_flag, 1);
}

// ----------------------------------------------------------------------------------------------------------------
// conv_epilogue_sel_wave — the mask head's deconvolution in selected-class mode (ConvDesc::sel_partial; TimeDistributedMaskLayer.swift:52-89,
// Conversion/task.py:108-116) straight from the accumulators: a lane holds 16 channels of ONE output pixel per column tile
// (lane (pixel l31, kk), slot 4q + r <-> channel 32 j + 8 q + 4 kk + r), so the dot with the selected class's 1x1 filter is 16 TNS FMAs
// per lane in a fixed order, one exchange with the partner lane (kk), and one store per pixel: no LDS tile, no block barrier, no
// five-level cross-lane tree per row (the block-staged form's epilogue took as long as the layer's K loop: DESIGN.md section 6 (11)).
//   partial[(b * 4 ohw + P) * (Cout / PC) + co / PC] = sum over this wave's PC = TNS * 32 columns;   tab: scale[BN] | shift[BN]
// ----------------------------------------------------------------------------------------------------------------
template <typename T, int BN, int TMS, int TNS>
__device__ __forceinline__ void conv_epilogue_sel_wave(const ConvArgs& a, f32x16 (&acc)[TMS][TNS], const float* tab, int row0, int n0, int colrel0, int lane)
{
    constexpr int PC = TNS * 32;
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const int nb = n0 + colrel0;
    const int qd = nb / a.Cout, co0 = nb - qd * a.Cout;          // the wave's columns lie inside one of the four output positions
    const bool relu = a.act == ACT_RELU;
    const int parts = a.Cout / PC;
    bool out_of_range = false;
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
        const int m = row0 + i * 32 + l31;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / ohw, pix = mm - b * ohw, oh = pix / a.OW, ow = pix - oh * a.OW;
        const int cid = ok ? a.sel_cid[b] : -1;
        const float* const wr = a.sel_w + (size_t)(cid < 0 ? 0 : cid) * a.Cout + co0;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < TNS; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cw = j * 32 + 8 * q + 4 * kk;                  // column inside the wave's PC
                const int cl = colrel0 + cw;                             // ... inside the block tile
                const float4 sc = *reinterpret_cast<const float4*>(tab + cl), sh = *reinterpret_cast<const float4*>(tab + BN + cl);
                const float4 w = *reinterpret_cast<const float4*>(wr + cw);
                float x[4] = {acc[i][j][4 * q] * sc.x + sh.x, acc[i][j][4 * q + 1] * sc.y + sh.y, acc[i][j][4 * q + 2] * sc.z + sh.z, acc[i][j][4 * q + 3] * sc.w + sh.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (relu) x[r] = fmaxf(x[r], 0.f);
                    out_of_range = out_of_range || (ok && !(fabsf(x[r]) < 65504.0f));
                    if constexpr (sizeof(T) == 2) x[r] = (float)(_Float16)x[r];          // the value the fp16 tensor would have held
                }
                dot += x[0] * w.x + x[1] * w.y + x[2] * w.z + x[3] * w.w;
            }
        dot += __shfl_xor(dot, 32);                                  // the partner lane's 16 TNS channels of the same pixel
        if (kk == 0 && cid >= 0) {
            const long P = (long)(2 * oh + (qd >> 1)) * (2 * a.OW) + (2 * ow + (qd & 1));
            a.sel_partial[((long)b * 4 * ohw + P) * parts + co0 / PC] = dot;
        }
    }
    if (a.range_flag && out_of_range) atomicOr(a.range_flag, 1);
}

// ----------------------------------------------------------------------------------------------------------------
// conv_epilogue_wave (fp32 tensors): the wave's (TMS·32) × (TNS·32) accumulators through a WAVE-PRIVATE LDS tile of 32 × 32
// outputs at a time — no block barrier — so that every store instruction writes FULL 128-B lines (8 pixel rows × 32
// channels): scattered 16-B pieces (conv_epilogue_direct with fp32 tensors) issue 3× slower per instruction and reach 2.9
// instead of 5.4 TB/s (tools/probes/vmem_probe.hip).  Same arithmetic, same order as the other epilogues: bit-identical.
//   stage: this wave's LDS tile, used as 32 rows × 32 floats with the 16-B chunk c of row r stored at chunk c ^ (r & 7)
//   (round 4; rounds 2-3 padded the rows to 36 floats, which keeps the transposing ds_write_b128 conflict-free but makes
//   every read-back ds_read_b128 a 2-way conflict under the instruction's real lane groups {0-3,12-15,20-27} / {4-11,16-19,
//   28-31} — MI355X_MICROARCH.md §LDS; tools/lds_bank_model.py enumerates both layouts: 32 vs 16 LDS cycles per 32 × 32
//   piece); tab: scale[BN] | shift[BN] of the block's columns.  A wave's LDS writes and reads execute in order, so no wait
//   is needed between the transposing write and the row-wise read-back.
// ----------------------------------------------------------------------------------------------------------------
// RES_FIRST: request the residual of ALL the wave's 32 x 32 pieces before the first piece is staged and stored (the loads of a piece
// otherwise queue behind the previous piece's stores — `res` may alias `out` — one memory round trip per piece).  Legal under the
// in-place contract: a wave loads exactly the elements it will overwrite, all of them before its first store.  Neutral for the
// 128-row kernels (two co-resident blocks cover the latency); used by the fused bottleneck tail, whose
// CU holds two waves per SIMD.
// res_acc / tab2 (fused shortcut): the residual of a piece is not loaded but formed from a second accumulator set, y = res_acc * scale2 + shift2 —
// the arithmetic of the shortcut convolution's own (activation-less) epilogue — through the same wave-private transpose.
template <int BN, int TMS, int TNS, bool RES_FIRST = false>
__device__ __forceinline__ void conv_epilogue_wave(const ConvArgs& a, f32x16 (&acc)[TMS][TNS], float* stage, const float* tab, int row0, int n0,
                                                   int colrel0, int lane, const f32x16 (*res_acc)[TNS] = nullptr, const float* tab2 = nullptr)
{
    constexpr int SW = 32;                               // floats per staged row (unpadded: XOR-swizzled 16-B chunks)
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const float* const res = static_cast<const float*>(a.res);
    float* const out = static_cast<float*>(a.out);
    const bool dense_out = a.out_sB == (long)ohw * a.out_sP;
    const bool dense_res = a.res_sB == (long)ohw * a.res_sW && a.res_shift == 0;
    const bool relu = a.act == ACT_RELU;
    const int rrow = lane >> 3, c4 = (lane & 7) * 4;     // read-back: eight lanes per row, four passes of eight rows
    bool out_of_range = false;
    float4 rv_all[RES_FIRST ? TMS : 1][RES_FIRST ? TNS : 1][4];
    if constexpr (RES_FIRST) {
        if (res) {
#pragma unroll
            for (int i = 0; i < TMS; ++i)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int m = row0 + i * 32 + 8 * ps + rrow;
                    const bool okr = m < a.M;
                    long rr = (long)m * a.res_sW;
                    if (!dense_res) {
                        const int mm = okr ? m : 0;
                        const int b = mm / ohw, pix = mm - b * ohw;
                        if (a.res_shift) {
                            const int oh = pix / a.OW, ow = pix - oh * a.OW;
                            rr = (long)b * a.res_sB + (long)(oh >> a.res_shift) * a.res_sH + (long)(ow >> a.res_shift) * a.res_sW;
                        } else rr = (long)b * a.res_sB + (long)pix * a.res_sW;
                    }
#pragma unroll
                    for (int j = 0; j < TNS; ++j) {
                        const int n = n0 + colrel0 + j * 32 + c4;
                        rv_all[i][j][ps] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (okr && n < a.ncols) rv_all[i][j][ps] = *reinterpret_cast<const float4*>(res + rr + n);
                    }
                }
        }
    }
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
        // rows of this 32-row slab the lane handles in the read-back: row = 8 * pass + rrow
        long o_row[4], r_row[4];
        bool ok[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = row0 + i * 32 + 8 * ps + rrow;
            ok[ps] = m < a.M;
            o_row[ps] = (long)m * a.out_sP; r_row[ps] = (long)m * a.res_sW;
            if (!dense_out || (res && !dense_res)) {
                const int mm = ok[ps] ? m : 0;
                const int b = mm / ohw, pix = mm - b * ohw;
                o_row[ps] = (long)b * a.out_sB + (long)pix * a.out_sP;
                if (res) {
                    if (a.res_shift) {
                        const int oh = pix / a.OW, ow = pix - oh * a.OW;
                        r_row[ps] = (long)b * a.res_sB + (long)(oh >> a.res_shift) * a.res_sH + (long)(ow >> a.res_shift) * a.res_sW;
                    } else r_row[ps] = (long)b * a.res_sB + (long)pix * a.res_sW;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TNS; ++j) {
            const int cl = colrel0 + j * 32 + c4;          // column inside the block tile
            const int n = n0 + cl;
            const bool col_ok = n < a.ncols;
            // the residual of the 32 × 32 piece is requested before the LDS round trip (its latency overlaps it)
            float4 rv[4];
            if (res_acc) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(&stage[l31 * SW + (((2 * q + kk) ^ (l31 & 7)) << 2)]) =
                        make_float4(res_acc[i][j][4 * q], res_acc[i][j][4 * q + 1], res_acc[i][j][4 * q + 2], res_acc[i][j][4 * q + 3]);
                const float4 sc2 = *reinterpret_cast<const float4*>(tab2 + cl), sh2 = *reinterpret_cast<const float4*>(tab2 + BN + cl);
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    float4 y = *reinterpret_cast<const float4*>(&stage[(8 * ps + rrow) * SW + (((lane & 7) ^ rrow) << 2)]);
                    y.x = y.x * sc2.x + sh2.x; y.y = y.y * sc2.y + sh2.y; y.z = y.z * sc2.z + sh2.z; y.w = y.w * sc2.w + sh2.w;
                    out_of_range = out_of_range || !(fabsf(y.x) < 65504.0f) || !(fabsf(y.y) < 65504.0f) || !(fabsf(y.z) < 65504.0f) || !(fabsf(y.w) < 65504.0f);
                    rv[ps] = y;
                }
            } else
            if (res) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    if constexpr (RES_FIRST) rv[ps] = rv_all[i][j][ps];
                    else {
                        rv[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok[ps] && col_ok) rv[ps] = *reinterpret_cast<const float4*>(res + r_row[ps] + n);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(&stage[l31 * SW + (((2 * q + kk) ^ (l31 & 7)) << 2)]) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            const float4 sc = *reinterpret_cast<const float4*>(tab + cl), sh = *reinterpret_cast<const float4*>(tab + BN + cl);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                float4 x = *reinterpret_cast<const float4*>(&stage[(8 * ps + rrow) * SW + (((lane & 7) ^ rrow) << 2)]);
                x.x = x.x * sc.x + sh.x; x.y = x.y * sc.y + sh.y; x.z = x.z * sc.z + sh.z; x.w = x.w * sc.w + sh.w;
                if (res || res_acc) { x.x += rv[ps].x; x.y += rv[ps].y; x.z += rv[ps].z; x.w += rv[ps].w; }
                if (relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
                if (ok[ps] && col_ok) {
                    out_of_range = out_of_range || !(fabsf(x.x) < 65504.0f) || !(fabsf(x.y) < 65504.0f) || !(fabsf(x.z) < 65504.0f) || !(fabsf(x.w) < 65504.0f);
                    *reinterpret_cast<float4*>(out + o_row[ps] + n) = x;
                }
            }
        }
    }
    if (a.range_flag && out_of_range) atomicOr(a.range_flag, 1);
}

// ----------------------------------------------------------------------------------------------------------------
// conv_epilogue_wave_h (fp16 tensors, wave tiles of 32·TMS × 64): the same idea for 2-byte outputs — a row of the wave tile is
// 64 channels = ONE 128-B line, so both 32 × 32 pieces of a row slab go through the wave's private LDS tile (32 × 68 floats)
// together and the read-back moves eight channels per lane: residual loads and stores of 8 full lines per instruction, where
// conv_epilogue_direct issues 64-B store pieces and 8-B residual pieces (3× slower to issue per CU, tools/probes/vmem_probe.hip).
// Same arithmetic in the same order (scale/shift, + residual, ReLU in fp32, ONE rounding): bit-identical.
// ----------------------------------------------------------------------------------------------------------------
template <int BN, int TMS>
__device__ __forceinline__ void conv_epilogue_wave_h(const ConvArgs& a, f32x16 (&acc)[TMS][2], float* stage, const float* tab, int row0, int n0,
                                                     int colrel0, int lane)
{
    constexpr int SW = 64;                               // floats per staged row; 16-B chunk c of row r lives at chunk c ^ (r & 7): conflict-free transposing writes AND read-back (round 4, as conv_epilogue_wave)
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const _Float16* const res = static_cast<const _Float16*>(a.res);
    _Float16* const out = static_cast<_Float16*>(a.out);
    const bool dense_out = a.out_sB == (long)ohw * a.out_sP;
    const bool dense_res = a.res_sB == (long)ohw * a.res_sW && a.res_shift == 0;
    const bool relu = a.act == ACT_RELU;
    const int rrow = lane >> 3, c8 = (lane & 7) * 8;     // read-back: eight lanes per row (one line), four passes of eight rows
    const int cl = colrel0 + c8;                         // column inside the block tile
    const int n = n0 + cl;
    const bool col_ok = n < a.ncols;
    const float4 sc0 = *reinterpret_cast<const float4*>(tab + cl), sc1 = *reinterpret_cast<const float4*>(tab + cl + 4);
    const float4 sh0 = *reinterpret_cast<const float4*>(tab + BN + cl), sh1 = *reinterpret_cast<const float4*>(tab + BN + cl + 4);
    bool out_of_range = false;
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
        long o_row[4], r_row[4];
        bool ok[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = row0 + i * 32 + 8 * ps + rrow;
            ok[ps] = m < a.M;
            o_row[ps] = (long)m * a.out_sP; r_row[ps] = (long)m * a.res_sW;
            if (!dense_out || (res && !dense_res)) {
                const int mm = ok[ps] ? m : 0;
                const int b = mm / ohw, pix = mm - b * ohw;
                o_row[ps] = (long)b * a.out_sB + (long)pix * a.out_sP;
                if (res) {
                    if (a.res_shift) {
                        const int oh = pix / a.OW, ow = pix - oh * a.OW;
                        r_row[ps] = (long)b * a.res_sB + (long)(oh >> a.res_shift) * a.res_sH + (long)(ow >> a.res_shift) * a.res_sW;
                    } else r_row[ps] = (long)b * a.res_sB + (long)pix * a.res_sW;
                }
            }
        }
        // the residual of the slab is requested before the LDS round trip (and before any store: `res` may alias `out`)
        uint4 rv[4];
        if (res) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                rv[ps] = make_uint4(0u, 0u, 0u, 0u);
                if (ok[ps] && col_ok) rv[ps] = *reinterpret_cast<const uint4*>(res + r_row[ps] + n);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(&stage[l31 * SW + (((j * 8 + 2 * q + kk) ^ (l31 & 7)) << 2)]) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            float4 x = *reinterpret_cast<const float4*>(&stage[(8 * ps + rrow) * SW + (((2 * (lane & 7)) ^ rrow) << 2)]);
            float4 y = *reinterpret_cast<const float4*>(&stage[(8 * ps + rrow) * SW + (((2 * (lane & 7) + 1) ^ rrow) << 2)]);
            x.x = x.x * sc0.x + sh0.x; x.y = x.y * sc0.y + sh0.y; x.z = x.z * sc0.z + sh0.z; x.w = x.w * sc0.w + sh0.w;
            y.x = y.x * sc1.x + sh1.x; y.y = y.y * sc1.y + sh1.y; y.z = y.z * sc1.z + sh1.z; y.w = y.w * sc1.w + sh1.w;
            if (res) {
                const f16x8 h = __builtin_bit_cast(f16x8, rv[ps]);
                x.x += (float)h[0]; x.y += (float)h[1]; x.z += (float)h[2]; x.w += (float)h[3];
                y.x += (float)h[4]; y.y += (float)h[5]; y.z += (float)h[6]; y.w += (float)h[7];
            }
            if (relu) {
                x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
                y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
            }
            if (ok[ps] && col_ok) {
                out_of_range = out_of_range || !(fabsf(x.x) < 65504.0f) || !(fabsf(x.y) < 65504.0f) || !(fabsf(x.z) < 65504.0f) || !(fabsf(x.w) < 65504.0f) ||
                               !(fabsf(y.x) < 65504.0f) || !(fabsf(y.y) < 65504.0f) || !(fabsf(y.z) < 65504.0f) || !(fabsf(y.w) < 65504.0f);
                store8h(out + o_row[ps] + n, x, y);
            }
        }
    }
    if (a.range_flag && out_of_range) atomicOr(a.range_flag, 1);
}

// two fp32 → packed fp16, ROUND TO NEAREST EVEN (v_cvt_pk_f16_f32; the builtin pack conversion truncates).  Every part of the
// splits below is rounded to nearest: the parts of an activation that cannot be carried exactly leave a two-sided, unbiased
// error (half the one-sided error of truncation).
// (A vector conversion, which hipcc lowers to the one instruction: written as inline asm the compiler does not pad the
// VALU-write -> MFMA-read hazard behind it and the matrix cores read a stale register — NaNs in the three-part kernels.)
__device__ __forceinline__ uint32_t cvt_pk_rne(float x, float y)
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}

// fp32 → (hi, lo) fp16 pair with hi + lo = a to 2^-23 relative: hi = a rounded to nearest fp16,
// lo = (a - hi) rounded to nearest fp16 (a - hi is exact in fp32; for |a| below ~1e-2 lo lands in the
// fp16 subnormals, whose 2^-24 absolute step is far under the fp32 rounding noise of the sums it feeds).
// fp16 × fp16 products are exact in the MFMA's fp32 accumulate, so hi·w + lo·w reproduces the fp32 product
// a·w for fp16-exact filters w.  |a| must stay below 65504 (as in any fp16 GPU path of the reference).
__device__ __forceinline__ void split_hi_lo(const uint4 u0, const uint4 u1, f16x8& hi, f16x8& lo)
{
    const float a[8] = {__uint_as_float(u0.x), __uint_as_float(u0.y), __uint_as_float(u0.z), __uint_as_float(u0.w),
                        __uint_as_float(u1.x), __uint_as_float(u1.y), __uint_as_float(u1.z), __uint_as_float(u1.w)};
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t h2 = cvt_pk_rne(a[2 * p], a[2 * p + 1]);
        // r = a - hi in ONE mixed-precision FMA (fp16 half of h2 × -1.0 + a): saves the widening conversion
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h2), "v"(a[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h2), "v"(a[2 * p + 1]));
        hw[p] = h2;
        lw[p] = cvt_pk_rne(r0, r1);
    }
    hi = __builtin_bit_cast(f16x8, uint4{hw[0], hw[1], hw[2], hw[3]});
    lo = __builtin_bit_cast(f16x8, uint4{lw[0], lw[1], lw[2], lw[3]});
}

// Three-part variant (MRCNN_F32X3): hi + mid + lo carries all 24 significand bits — exact for 0.5 <= |a| < 65504, and to
// 2^-25 absolute below (where the last part reaches the fp16 subnormal step and is rounded to nearest): for |a| >= 0.5
// the fp32 product a·w is reproduced exactly; the bound for smaller activations is the one include/maskrcnn_hip.h states
// and tests/test_gpu_conv_kernels.py::test_split_modes_scale_curve_stays_inside_the_documented_bound pins.
__device__ __forceinline__ void split_hi_mid_lo(const uint4 u0, const uint4 u1, f16x8& hi, f16x8& mid, f16x8& lo)
{
    const float a[8] = {__uint_as_float(u0.x), __uint_as_float(u0.y), __uint_as_float(u0.z), __uint_as_float(u0.w),
                        __uint_as_float(u1.x), __uint_as_float(u1.y), __uint_as_float(u1.z), __uint_as_float(u1.w)};
    uint32_t hw[4], mw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t h2 = cvt_pk_rne(a[2 * p], a[2 * p + 1]);
        float r0, r1, q0, q1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h2), "v"(a[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h2), "v"(a[2 * p + 1]));
        const uint32_t m2 = cvt_pk_rne(r0, r1);
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(m2), "v"(r0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(m2), "v"(r1));
        hw[p] = h2;
        mw[p] = m2;
        lw[p] = cvt_pk_rne(q0, q1);
    }
    hi = __builtin_bit_cast(f16x8, uint4{hw[0], hw[1], hw[2], hw[3]});
    mid = __builtin_bit_cast(f16x8, uint4{mw[0], mw[1], mw[2], mw[3]});
    lo = __builtin_bit_cast(f16x8, uint4{lw[0], lw[1], lw[2], lw[3]});
}

}  // namespace mrcnn
