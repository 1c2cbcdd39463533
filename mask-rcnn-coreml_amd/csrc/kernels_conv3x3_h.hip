// kernels_conv3x3_h.hip — the 3x3 stride-1 'same' convolutions of the fp16 mode (BASELINE configs[3]) on gfx950: the RPN's shared layer
// (256 -> 512 on P2..P6) and the FPN's output layers (256 -> 256), Sources/maskrcnn/Python/Conversion/task.py:69-92 (layer list:
// the Matterport graph).
//
// Why another kernel (measured): the fp16-MFMA kernels of this engine are bound by the CU's vector-memory
// front end — one 1-KB request per ~37 clocks whatever the instruction — not by the matrix pipe: the 128 x 128 tile needs 64 B/clk at
// full MFMA rate, the 256 x 256 ping-pong tile 32, and both move every input pixel nine times (once per tap).  The fused bottleneck
// (kernels_bneck.hip) showed what removing that traffic buys: its 3x3 phase — activations resident in LDS as a HALO TILE, read by the
// nine taps as shifted windows, filters streamed as MFMA fragments straight into registers, no barrier in the loop — runs at
// 1.29 PFLOP/s with 128-pixel tiles against 1.05-1.09 for the ping-pong kernel on the same layer.  This kernel is that phase as a
// layer of its own, with 256-pixel tiles (half the filter bytes per MFMA again):
//   * a block of 512 threads owns a 16 x 16 output tile x 256 output channels (a "unit"; N = 512 -> two units per tile); wave w owns
//     channels 32 w .. of ALL 256 pixels (eight 32 x 32 accumulators) and streams ITS filter fragments from a copy of the filters in
//     fragment order (conv3x3h_pack: one coalesced 1-KB load per 16-wide K group, six groups ahead);
//   * the input arrives per 64-CHANNEL BLOCK: [18 x 18 halo pixels][128 B] = 41 KB, two stages; the next block is requested a half
//     block ahead into registers (three 16-B pieces per thread and half) and written to the other stage — every load is one the
//     compiler counts, no wait ever drains the filter stream; ONE barrier per 288 MFMAs of a wave;
//   * K order = (64-channel block, tap, 16-wide group): the layer's own canonical order — it differs from the tap-major order of the
//     128-row / ping-pong kernels, so WHICH kernel a layer runs on is a property of the layer (conv3x3h_eligible: dtype, geometry,
//     widths), never of the batch; per-image results do not depend on the batch (tests/test_gpu_conv3x3h.py);
//   * epilogue: scale / shift (+ ReLU) in fp32, one rounding to fp16, the tile leaves through LDS in full 512-B rows.
// LDS bank notes: as kernels_bneck.hip (swizzle keyed on the halo COLUMN, pitch 18).
#include <stdlib.h>
#include <type_traits>

#include "conv_device.h"

namespace mrcnn {

struct C3hArgs {
    const _Float16* in; _Float16* out;
    const uint4* wf;                 // filters in fragment order, channel-block-major (conv3x3h_pack)
    const float *scale, *shift;
    long in_sB, in_sH, in_sW;        // elements
    long out_sB, out_sP;
    int B, H, W, Cin, Cout, act;
    int tiles_x, tiles_y, npass, nunits;
    int* range_flag;
    // HEAD: a 1x1 head over the layer's output fused in (the RPN's class / box heads, ConvDesc::head_*): the layer's own output is not stored
    const uint4* head_wf;            // [32][Cout] head filters in fragment order (bneck_pack_frag)
    const float* head_bias;
    float *head_out, *head_out2;
    long head_out_sB, head_out_sP, head_out2_sB, head_out2_sP;
    int head_split, head_cols;
    float head_mul;
};

#define C3H_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0);

namespace {
constexpr int HWD = 18, HP = 324, STAGE = HP * 128, YB = 256 * 512;
constexpr int OFF_TAB = YB, OFF_HW = YB + 2 * 256 * 4, HW_LANES = 36, HW_BYTES = 32 * HW_LANES * 16, LDS_BYTES = OFF_HW + HW_BYTES;
// (OFF_HW: HEAD — the heads' filter fragments, rows 0..17 of each 1-KB granule only: 18 real head columns, 2 x 18 lanes x 16 B per 16-wide K group, 32 groups)
typedef unsigned c3h_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 c3h_pack16(const float4 va, const float4 vb)
{
    f16x4 ha4, hb4;
    ha4[0] = (_Float16)va.x; ha4[1] = (_Float16)va.y; ha4[2] = (_Float16)va.z; ha4[3] = (_Float16)va.w;
    hb4[0] = (_Float16)vb.x; hb4[1] = (_Float16)vb.y; hb4[2] = (_Float16)vb.z; hb4[3] = (_Float16)vb.w;
    const uint2 pa = __builtin_bit_cast(uint2, ha4), pb = __builtin_bit_cast(uint2, hb4);
    const auto sx = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
    return make_uint4(sx[0], sy[0], sx[1], sy[1]);
}
__device__ __forceinline__ bool c3h_bad(const float4 v)
{
    return !(fabsf(v.x) < 65504.0f) || !(fabsf(v.y) < 65504.0f) || !(fabsf(v.z) < 65504.0f) || !(fabsf(v.w) < 65504.0f);
}
}  // namespace

// HEAD = true (the RPN's shared layer on the levels whose heads ride along): a unit is a whole TILE — its output-column passes run back to
// back in one block — and after each pass's epilogue, with the ReLU'd 256 x 256 piece sitting in LDS as fp16 (exactly what the layer
// would have stored), wave w multiplies pixels 32 w .. of it with the heads' filters: 16 MFMAs per pass into ONE 32 x 32 accumulator
// that runs over all passes (K = Cout in ascending order, 64-channel steps of four groups: the order of the separate head launch, so the
// logits / deltas agree with it bit for bit).  The Cout-channel tensor (1.07 GB at P2 for a batch of 8 in fp32 terms, 537 MB in fp16) is
// neither written nor read back, and the head launches disappear.
template <bool HEAD>
__global__ __launch_bounds__(512) void k_conv3x3_h(const C3hArgs a)
{
    constexpr int D = 4;                       // filter fragments in flight per wave (a K group is eight MFMAs: >= 1 000 clocks of cover)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    float* const tab = reinterpret_cast<float*>(smem + OFF_TAB);
    const int CB = a.Cin >> 6;                 // 64-channel blocks
    const int KGU = CB * 36;                   // 16-wide K groups of a unit: (block, tap, group)

    // staging geometry: piece q = t + 512 m (m < 6) = halo pixel q >> 3, 16-B chunk q & 7; q >= 2592 does not exist
    int s_py[6], s_px[6];
    unsigned s_lds[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int q = t + 512 * m, r = q >> 3, c = q & 7;
        const int py = r / HWD, px = r - py * HWD;
        s_py[m] = q < HP * 8 ? py : -100000;   // (never inside an image)
        s_px[m] = px;
        s_lds[m] = (unsigned)(r * 128 + ((c ^ ((px >> 1) & 7)) << 4));
    }
    // fragment rows: pixel p = i*32 + l31 -> tile row 2i + (l31 >> 4), column l31 & 15; tap (dy, dx): halo pixel (row + dy, col + dx)
    const unsigned frow = (unsigned)(((l31 >> 4) * HWD + (l31 & 15)) * 128);
    const unsigned kk4 = (unsigned)(kk << 4);

    if constexpr (HEAD) {
        // the heads' filters stay in LDS for the life of the block (every unit multiplies with them: 16 dependent MFMAs per pass, and a
        // 1-KB load from L2 in front of each cost 106 us on the P2 level)
        for (int i = t; i < 32 * HW_LANES; i += 512) {
            const int kg = i / HW_LANES, r = i - kg * HW_LANES, hk = r / 18, row = r - hk * 18;
            *reinterpret_cast<uint4*>(smem + OFF_HW + i * 16) = kg * 16 < a.Cout ? a.head_wf[(size_t)kg * 64 + hk * 32 + row] : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
    }
    bool range_trip = false;
    const int q8 = a.nunits >> 3, r8 = a.nunits & 7;
    for (int v = blockIdx.x; v < a.nunits; v += gridDim.x) {
        // XCD-aware bijective walk: the blocks of one XCD own a contiguous run of units (the passes of a tile adjacent)
        const int xcd = v & 7, local = v >> 3;
        const int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
        const int tile = HEAD ? unit : unit / a.npass;
        const int pass0 = HEAD ? 0 : unit - tile * a.npass, pass1 = HEAD ? a.npass : pass0 + 1;
        const int per_img = a.tiles_x * a.tiles_y;
        const int b = tile / per_img, tr = tile - b * per_img;
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int y0 = ty * 16, x0 = tx * 16;
        const _Float16* const img = a.in + (size_t)b * a.in_sB;
        const unsigned img_bytes = (unsigned)((((size_t)(a.H - 1) * a.in_sH + (size_t)(a.W - 1) * a.in_sW) + a.Cin) * 2);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(img), 0, (int)img_bytes, 0x00020000);
        int xo[6];                   // byte offset of the piece at channel block 0, or out of range (-> zeros: the layer's padding)
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int gy = y0 - 1 + s_py[m], gx = x0 - 1 + s_px[m];
            const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            xo[m] = ok ? (int)((((size_t)gy * a.in_sH + (size_t)gx * a.in_sW) + (size_t)(t & 7) * 8) * 2) : (int)0x80000000u;
        }
        f32x16 hacc;                 // (HEAD: assigned in the head section of every pass: never live across a K loop)
        for (int pass = pass0; pass < pass1; ++pass) {
        // this wave's filter stream: granules ((pass * 8 + wave) * KGU + q), q = 0 .. KGU - 1
        const uint4* const wp = a.wf + (size_t)(pass * 8 + wave) * KGU * 64 + lane;
        uint4 wq[D];
#pragma unroll
        for (int d = 0; d < D; ++d) wq[d] = wp[d * 64];
        if (t < 128) {               // scale | shift of the unit's 256 columns
            const int c = (t & 63) * 4;
            const float* src = t < 64 ? a.scale : a.shift;
            const float fill = t < 64 ? 1.0f : 0.0f;
            *reinterpret_cast<float4*>(tab + (t < 64 ? 0 : 256) + c) =
                src ? *reinterpret_cast<const float4*>(src + pass * 256 + c) : make_float4(fill, fill, fill, fill);
        }
        // channel block 0 -> stage 0 (the previous unit's tile has left the LDS: barrier at the end of the loop body)
        c3h_u32x4 xr[3];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int m = 0; m < 3; ++m) xr[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, xo[hf * 3 + m], 0, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (hf == 0 || s_py[3 + m] > -1000) *reinterpret_cast<c3h_u32x4*>(smem + s_lds[hf * 3 + m]) = xr[m];
        }
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
        __syncthreads();
        for (int cb = 0; cb < CB; ++cb) {
            const unsigned char* const sb = smem + (cb & 1) * STAGE;
            unsigned char* const so = smem + ((cb & 1) ^ 1) * STAGE;
            const int cbn = cb + 1 < CB ? cb + 1 : cb;           // (the last block re-requests itself: no branch in the stream; nobody reads the copy)
#pragma unroll
            for (int m = 0; m < 3; ++m) xr[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, xo[m], cbn * 128, 0);
            // Activation fragments: each of the eight registers sets is re-requested (for the NEXT K group) right behind the MFMA that
            // consumed it — the read has 7 MFMA issues (>= 220 clocks) to return, no second register set, no exposed LDS latency.
            // (Left to the compiler's own order under 256 registers: read, wait, MFMA, one at a time — 47 % of the matrix pipe.)
            auto a_off = [&](int k) {
                const int tap = k >> 2, g = k & 3, dy = tap / 3, dx = tap - 3 * dy;
                const unsigned sw = (unsigned)(((((l31 & 15) + dx) >> 1) & 7) << 4) ^ kk4;
                return frow + (unsigned)((dy * HWD + dx) * 128) + (sw ^ (unsigned)(g << 5));
            };
            f16x8 af[8];
            {
                const unsigned ao = a_off(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const f16x8*>(sb + ao + i * 2 * HWD * 128);
            }
#pragma unroll
            for (int k = 0; k < 36; ++k) {
                const f16x8 wfr = __builtin_bit_cast(f16x8, wq[k % D]);
                const unsigned an = a_off(k < 35 ? k + 1 : k);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    C3H_MFMA(wfr, af[i], acc[i])
                    if (k < 35) af[i] = *reinterpret_cast<const f16x8*>(sb + an + i * 2 * HWD * 128);
                }
                int qn = cb * 36 + k + D;
                qn = qn < KGU ? qn : KGU - 1;
                wq[k % D] = wp[(size_t)qn * 64];
                __builtin_amdgcn_sched_barrier(0);
                if (k == 17) {       // first half of the next block lands in the other stage; its second half is requested
#pragma unroll
                    for (int m = 0; m < 3; ++m) *reinterpret_cast<c3h_u32x4*>(so + s_lds[m]) = xr[m];
#pragma unroll
                    for (int m = 0; m < 3; ++m) xr[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, xo[3 + m], cbn * 128, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (s_py[3 + m] > -1000) *reinterpret_cast<c3h_u32x4*>(so + s_lds[3 + m]) = xr[m];
            __syncthreads();
        }
        // ---- epilogue: scale / shift (+ ReLU), fp16, through the LDS tile [256 pixels][512 B], chunk c of pixel p at c ^ (p & 15) ----
        const bool relu = a.act == ACT_RELU;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cl = wave * 32 + 16 * p + 4 * kk;
                const float4 sa = *reinterpret_cast<const float4*>(tab + cl), sb_ = *reinterpret_cast<const float4*>(tab + cl + 8);
                const float4 ha = *reinterpret_cast<const float4*>(tab + 256 + cl), hb = *reinterpret_cast<const float4*>(tab + 256 + cl + 8);
                float4 va = make_float4(acc[i][8 * p + 0], acc[i][8 * p + 1], acc[i][8 * p + 2], acc[i][8 * p + 3]);
                float4 vb = make_float4(acc[i][8 * p + 4], acc[i][8 * p + 5], acc[i][8 * p + 6], acc[i][8 * p + 7]);
                va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                if (relu) {
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                    vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                }
                const int py = 2 * i + (l31 >> 4), px = l31 & 15;
                if (y0 + py < a.H && x0 + px < a.W) range_trip = range_trip || c3h_bad(va) || c3h_bad(vb);
                *reinterpret_cast<uint4*>(smem + (i * 32 + l31) * 512 + (((wave * 4 + 2 * p + kk) ^ (l31 & 15)) << 4)) = c3h_pack16(va, vb);
            }
        __syncthreads();
        if constexpr (HEAD) {
            const int p = wave * 32 + l31;               // this lane's pixel of the tile
            const int hrow = l31 < 18 ? l31 : 17;
            if (pass > pass0) {
                const float4* const hs = reinterpret_cast<const float4*>(a.out) + (size_t)unit * 2048;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float4 v4 = hs[e4 * 512 + t];
                    hacc[4 * e4] = v4.x; hacc[4 * e4 + 1] = v4.y; hacc[4 * e4 + 2] = v4.z; hacc[4 * e4 + 3] = v4.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) hacc[e] = 0.0f;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                uint4 hwv = *reinterpret_cast<const uint4*>(smem + OFF_HW + (((pass * 16 + g) * HW_LANES + kk * 18 + hrow) << 4));
                if (l31 >= 18) hwv = make_uint4(0u, 0u, 0u, 0u);          // head columns 18..31 do not exist
                const f16x8 hw = __builtin_bit_cast(f16x8, hwv);
                const f16x8 av = *reinterpret_cast<const f16x8*>(smem + p * 512 + (((2 * g + kk) ^ (p & 15)) << 4));
                C3H_MFMA(hw, av, hacc)
            }
        } else {
            _Float16* const oimg = a.out + (size_t)b * a.out_sB + (size_t)pass * 256;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = t + 512 * j, px_ = e >> 5, slot = e & 31;          // a wave instruction = two pixels x 512 contiguous bytes
                const int py = px_ >> 4, pxx = px_ & 15;
                const uint4 val = *reinterpret_cast<const uint4*>(smem + px_ * 512 + slot * 16);
                const int c = slot ^ (px_ & 15);
                if (y0 + py < a.H && x0 + pxx < a.W)
                    *reinterpret_cast<uint4*>(oimg + ((size_t)(y0 + py) * a.W + (x0 + pxx)) * a.out_sP + c * 8) = val;
            }
        }
        __syncthreads();             // the tile has left the LDS (or the heads have read it): the next pass / unit may stage
        if constexpr (HEAD) {
            // Between passes the heads' running sums wait in MEMORY, not in registers (live across the K loop they cost 64 spilled dwords
            // inside it): 32 KB per unit in the layer's own output tensor, which exists (the un-fused levels use it) and is not written here.
            if (pass + 1 < pass1) {
                float4* const hs = reinterpret_cast<float4*>(a.out) + (size_t)unit * 2048;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) hs[e4 * 512 + t] = make_float4(hacc[4 * e4], hacc[4 * e4 + 1], hacc[4 * e4 + 2], hacc[4 * e4 + 3]);
            }
        }
        }
        if constexpr (HEAD) {
            // lane (l31, kk): pixel 32 wave + l31, head columns 8q + 4kk + r
            const int p = wave * 32 + l31, py = p >> 4, px = p & 15;
            if (y0 + py < a.H && x0 + px < a.W) {
                const size_t pix = (size_t)(y0 + py) * a.W + (x0 + px);
                float* const o1 = a.head_out + (size_t)b * a.head_out_sB + pix * a.head_out_sP;
                float* const o2 = a.head_out2 + (size_t)b * a.head_out2_sB + pix * a.head_out2_sP;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = 8 * (e >> 2) + 4 * kk + (e & 3);
                    const float vv = hacc[e] * a.head_mul + (c < a.head_cols ? a.head_bias[c] : 0.0f);
                    if (c < a.head_split) o1[c] = vv;
                    else if (c < a.head_cols) o2[c - a.head_split] = vv;
                }
            }
        }
    }
    if (a.range_flag && range_trip) atomicOr(a.range_flag, 1);
}

// [N][9][Cin] fp16 filters (the engine's packed order: tap-major, channel-minor) -> 1-KB granules in the kernel's stream order:
// granule ((nt * CB + cb) * 9 + tap) * 4 + g, lane (l31, kk) <- filter row 32 nt + l31, taps' channels 64 cb + 16 g + 8 kk .. + 7
__global__ void k_conv3x3h_pack(const _Float16* __restrict__ w, int N, int Cin, uint4* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int CB = Cin / 64;
    if (i >= (long)(N / 32) * CB * 36 * 64) return;
    const int lane = (int)(i & 63);
    long gq = i >> 6;
    const int g = (int)(gq & 3); gq >>= 2;
    const int tap = (int)(gq % 9); gq /= 9;
    const int cb = (int)(gq % CB), nt = (int)(gq / CB);
    out[i] = *reinterpret_cast<const uint4*>(w + ((size_t)(nt * 32 + (lane & 31)) * 9 + tap) * Cin + cb * 64 + g * 16 + (lane >> 5) * 8);
}

void conv3x3h_pack(hipStream_t s, const void* wgt_std, int N, int Cin, DevBuf& out)
{
    MRCNN_REQUIRE(N % 32 == 0 && Cin % 64 == 0, MRCNN_ERR_SHAPE, "conv3x3h_pack: [%d][9][%d]", N, Cin);
    const long n = (long)(N / 32) * (Cin / 64) * 36 * 64;
    out.alloc((size_t)n * 16);
    hipLaunchKernelGGL(k_conv3x3h_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const _Float16*>(wgt_std), N, Cin, out.as<uint4>());
    HIP_CHECK(hipGetLastError());
}

bool conv3x3h_packable(int KH, int KW, int Cin, int Cout, int Npad)
{
    return KH == 3 && KW == 3 && Cin % 64 == 0 && Cin >= 64 && Cout % 256 == 0 && Npad == Cout;
}

bool conv3x3h_eligible(const ConvDesc& d)
{
    const int wdt = d.wdtype < 0 ? d.dtype : d.wdtype;
    if (d.dtype != MRCNN_F16 || wdt != MRCNN_F16 || d.out_f32 || !d.wgt_c3h) return false;
    if (!conv3x3h_packable(d.KH, d.KW, d.Cin, d.Cout, d.Npad) || d.stride != 1 || d.padH != 1 || d.padW != 1) return false;
    if (d.OH != d.H || d.OW != d.W || d.res || d.out2 || d.deconv2 || d.sel_partial || d.act == ACT_SIGMOID) return false;
    if (d.head_w && (long)d.B * ((d.H + 15) / 16) * ((d.W + 15) / 16) * 32768 > (long)d.B * d.out_sB * 2) return false;      // (the heads' partial sums wait in the output tensor)
    if (d.head_w && !(d.Cout <= 512 && d.head_out && d.head_out2 && d.head_cols > 0 && d.head_cols <= 18 && d.head_split <= d.head_cols && d.act == ACT_RELU)) return false;
    auto al = [](const void* p, size_t n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
    if (!al(d.in, 16) || !al(d.out, 16) || d.in_sW % 8 || d.in_sH % 8 || d.in_sB % 8 || d.out_sP % 8 || d.out_sB % 8) return false;
    if (d.out_sB < (long)d.OH * d.OW * d.out_sP) return false;
    if (((size_t)(d.H - 1) * d.in_sH + (size_t)(d.W - 1) * d.in_sW + d.Cin) * 2 >= 0x80000000ull) return false;
    return true;
}

void conv3x3h_launch(hipStream_t s, const ConvDesc& d, int* range_flag, int n_cus)
{
    MRCNN_REQUIRE(conv3x3h_eligible(d), MRCNN_ERR_INVALID, "conv3x3h: layer not eligible");
    C3hArgs a;
    a.in = static_cast<const _Float16*>(d.in); a.out = static_cast<_Float16*>(d.out);
    a.wf = static_cast<const uint4*>(d.wgt_c3h);
    a.scale = d.scale; a.shift = d.shift;
    a.in_sB = d.in_sB; a.in_sH = d.in_sH; a.in_sW = d.in_sW;
    a.out_sB = d.out_sB; a.out_sP = d.out_sP;
    a.B = d.B; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Cout = d.Cout; a.act = d.act;
    a.tiles_x = (d.W + 15) / 16; a.tiles_y = (d.H + 15) / 16; a.npass = d.Cout / 256;
    const bool head = d.head_w != nullptr;
    a.nunits = d.B * a.tiles_x * a.tiles_y * (head ? 1 : a.npass);
    a.range_flag = range_flag;
    a.head_wf = static_cast<const uint4*>(d.head_w); a.head_bias = d.head_bias;
    a.head_out = d.head_out; a.head_out2 = d.head_out2;
    a.head_out_sB = d.head_out_sB; a.head_out_sP = d.head_out_sP; a.head_out2_sB = d.head_out2_sB; a.head_out2_sP = d.head_out2_sP;
    a.head_split = d.head_split; a.head_cols = d.head_cols; a.head_mul = d.head_mul;
    int grid = n_cus > 0 ? n_cus / 8 * 8 : 256;
    if (grid <= 0) grid = 8;
    if (a.nunits < grid) grid = a.nunits;
    if (head) hipLaunchKernelGGL(k_conv3x3_h<true>, dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_conv3x3_h<false>, dim3(grid), dim3(512), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace mrcnn
