// kernels_conv_stem.hip — the stem of the trunk in the split modes: conv1 (7x7 stride 2, 3 -> 64, BatchNorm, ReLU) and the 3x3 stride-2
// max-pool behind it as ONE persistent kernel (round 4).
//
// Why: as an implicit GEMM on the 128-row kernel, conv1 stages one kernel row (8 pixels x 4 channels = 128 B) per output pixel and
// K step — neighbouring output pixels overlap by three quarters, so 1.9 GB cross L2 -> LDS for a 136 MB input (batch 8), every
// wave column splits the same values again, and the 537 MB fp32 output is written only to be read back by the pool: 335 + 144 us
// on neither roofline.  Here a block owns a patch of 3 x 16 POOLED outputs: the 19 x 71 input pixels it
// needs are loaded once, split once into fp16 planes in LDS (8 B per pixel and part), the 7 x 33 conv outputs behind the patch are
// computed from shifted windows of those planes, activated, parked in LDS and max-pooled there; conv1's output never exists.
//
// Arithmetic: EXACTLY the 128-row kernel's for this layer — K order = kernel rows ascending, within a row the two 16-wide groups
// (4 pixels x 4 channels each), parts hi / mid / lo per group, one running fp32 accumulator, y = max(acc * scale + shift, 0) —
// so the fused stem is bit-identical to conv1 + max-pool (tests), at every batch.  The 1.2x recomputation of conv outputs shared by
// neighbouring patches changes no value.
//
// Per tile: 8 waves, wave w owns conv outputs 32 w .. 32 w + 31 of the patch (row-major over 7 x 33 = 231, padded to 256) x all 64
// channels: 14 groups x PARTS x 2 MFMAs; the filter fragments of all 14 groups stay in registers for the life of the block (112
// VGPRs); the next tile's input patch is loaded one tile ahead.  LDS: planes 3 x 11 KB + conv tile 64 KB.
#include "conv_device.h"

namespace mrcnn {

namespace {
constexpr int PR = 3, PC = 16;                  // pooled rows / columns of a tile
constexpr int CR = 2 * PR + 1, CC = 2 * PC + 1; // conv outputs behind it: 7 x 33
constexpr int IR = 2 * CR + 5, IC = 72;         // input rows 19, columns 2 * 33 + 5 = 71 (pitch 72)
constexpr int NQ = 256;                         // conv outputs per tile, padded to 8 MFMA row blocks
static_assert(CR * CC <= NQ, "tile shape");

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
}  // namespace

struct StemArgs {
    const void* in;               // zero-padded input (B, Hp, Wp) x 16 B per pixel: NHWC4 fp32 (split modes) or NHWC8 fp16 — pixel - mean, spare channels 0 (k_preprocess)
    const _Float16* wgt;          // [64][7][32 | 64] fp16: k = kw * 4 + ci (split modes) / kw * 8 + ci (fp16) inside a kernel row (pack_conv1)
    const float* scale; const float* shift;
    void* out;                    // pooled NHWC (B, PH, PW, 64), fp32 / fp16
    int B, Hp, Wp, CH, CW, PH, PW;
    int tiles_r, tiles_c, n_tiles;
    int* range_flag;
};

// 4 fp32 -> PARTS x 4 fp16: the round-to-nearest chain of split_hi_mid_lo / split_hi_lo (conv_device.h)
template <int PARTS>
__device__ __forceinline__ void stem_split4(const u32x4_t v, u32x2_t (&part)[3])
{
    const float f[4] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const uint32_t h2 = cvt_pk_rne(f[2 * p2], f[2 * p2 + 1]);
        float e0, e1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(e0) : "v"(h2), "v"(f[2 * p2]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(e1) : "v"(h2), "v"(f[2 * p2 + 1]));
        const uint32_t m2 = cvt_pk_rne(e0, e1);
        part[0][p2] = h2;
        part[1][p2] = m2;
        if constexpr (PARTS == 3) {
            float g0, g1;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(g0) : "v"(m2), "v"(e0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(g1) : "v"(m2), "v"(e1));
            part[2][p2] = cvt_pk_rne(g0, g1);
        }
    }
}

// PARTS = 2 | 3: the split modes (fp32 tensors, 4 channels per pixel, two 16-wide K groups per kernel row = 4 pixels each);
// PARTS = 1: the fp16 mode (fp16 tensors staged as 8 channels per pixel).  COMPACT (round 5, the default): only the first four channels of a
//   staged pixel exist (3 + one zero), so the planes keep 8 B per pixel and a kernel row is the split modes' TWO K groups of 4 pixels
//   each — half the MFMAs, and the 14 x 2 filter fragments fit in registers (gathered once from the 8-channel packing) instead of
//   being re-read from LDS for every tile (84 LDS fragment reads per 56 MFMAs bound the four-group form: 242 us against 260 for the
//   split modes' three passes).  The zero products a group no longer carries change how the 16 products of an MFMA are grouped: results
//   agree with the four-group form / the two launches to fp32 summation noise, not bit for bit (tests: within one fp16 ulp).
//   !COMPACT: round 4's four groups of 2 pixels x 8 channels, filters through LDS ("conv_stem" 2: A/B and the bit-identity test).
//   The compact form also splits the block's work by COLUMN block between the waves (wave = 64 conv outputs x 32 channels instead of 32 x 64): half the
//   filter fragments per wave (56 registers), so that the kernel fits 128 registers and TWO blocks share a CU (75 KB of LDS each) — one block's
//   barrier-separated phases (patch load, MFMAs, activation, pool, store: 4.4 us per tile, 0.5 us of it MFMAs) run under the other's.  Same sums per output.
template <int PARTS, bool COMPACT = true>
__global__ __launch_bounds__(512, (PARTS == 1 && COMPACT) ? 4 : 2) void k_conv_stem(const StemArgs a)
{
    constexpr bool WS = PARTS == 1 && COMPACT;        // waves split by column block: NR row tiles x NJ column blocks per wave
    constexpr int NR = WS ? 2 : 1, NJ = WS ? 1 : 2;
    constexpr bool F16 = PARTS == 1 && !COMPACT;      // the four-group form
    constexpr bool H8 = PARTS == 1;                   // fp16 tensors: a staged pixel is 16 B of fp16 (8 channels)
    constexpr int PXB = F16 ? 16 : 8;                 // bytes of a pixel in a plane
    constexpr int NG = F16 ? 4 : 2;                   // K groups per kernel row
    constexpr int NGR = 7 * NG;                       // ... per output
    constexpr int PLANE = IR * IC * PXB;              // one part of the input patch
    constexpr int WF = F16 ? NGR * 2 * 1024 : 0;      // four-group form: the filter fragments live in LDS (28 groups x 2 would be 224 registers)
    constexpr int CT = NQ * 64 * 4;                   // activated conv outputs [q][64], 16-B chunk c of row q at c ^ (q & 7)
    constexpr int NPX = (IR * IC + 511) / 512;        // input pixels a thread stages per tile (3)
    __shared__ __attribute__((aligned(16))) unsigned char smem[PARTS * PLANE + CT + 2 * 64 * 4 + WF];
    unsigned char* const planes = smem;
    float* const ct = reinterpret_cast<float*>(smem + PARTS * PLANE);
    float* const tab = reinterpret_cast<float*>(smem + PARTS * PLANE + CT);        // scale[64] | shift[64]
    unsigned char* const wf = smem + PARTS * PLANE + CT + 2 * 64 * 4;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kk = lane >> 5;

    // ---- filter fragments stay in REGISTERS for the life of the block (every wave multiplies all 14 groups x 2 column blocks once
    // per tile: through LDS they were 40 % of the fragment reads): group g = 2 kh + G, column block j: lane (n, kk) holds
    // W[32 j + n][kh][16 G + 8 kk .. + 8] ------------------------------------------------------------------------------------------
    const int jb0 = WS ? (wave & 1) : 0;              // first column block of this wave
    uint4 bw[F16 ? 1 : NGR][NJ];
    if constexpr (!F16) {
#pragma unroll
        for (int g = 0; g < NGR; ++g)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (H8) {
                    // gathered from the 8-channel packing [64][7][64] (k = kw * 8 + ci): pixels kw = 4 G + 2 kk and kw + 1, four channels each
                    const _Float16* src = a.wgt + ((size_t)((jb0 + j) * 32 + l31) * 7 + g / NG) * 64 + (4 * (g % NG) + 2 * kk) * 8;
                    const uint2 p0 = *reinterpret_cast<const uint2*>(src), p1 = *reinterpret_cast<const uint2*>(src + 8);
                    bw[g][j] = make_uint4(p0.x, p0.y, p1.x, p1.y);
                } else
                bw[g][j] = *reinterpret_cast<const uint4*>(a.wgt + ((size_t)(j * 32 + l31) * 7 + g / NG) * (16 * NG) + 16 * (g % NG) + 8 * kk);
            }
    } else {
        for (int e = t; e < NGR * 2 * 64; e += 512) {
            const int ln = e & 63, j = (e >> 6) & 1, g = e >> 7;
            *reinterpret_cast<uint4*>(wf + (size_t)e * 16) =
                *reinterpret_cast<const uint4*>(a.wgt + ((size_t)(j * 32 + (ln & 31)) * 7 + g / NG) * (16 * NG) + 16 * (g % NG) + 8 * (ln >> 5));
        }
    }
    if (t < 64) { tab[t] = a.scale ? a.scale[t] : 1.0f; tab[64 + t] = a.shift ? a.shift[t] : 0.0f; }
    // this lane's conv output inside the tile and the plane offset of its tap (0, 0) input pixel pair
    int qv[NR];
    bool q_realv[NR];
    unsigned a_basev[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        qv[r] = (WS ? (wave >> 1) * 64 + r * 32 : wave * 32) + l31;
        q_realv[r] = qv[r] < CR * CC;                   // rows 231..255 of the eighth block are padding: they multiply conv output 0's window and are never read
        const int qa = q_realv[r] ? qv[r] : 0;
        const int qr = qa / CC, qc = qa - qr * CC;
        // split modes / compact fp16: a lane's fragment of a group = 2 neighbouring pixels (8 B each); four-group fp16: one pixel (16 B)
        a_basev[r] = F16 ? (unsigned)(((2 * qr) * IC + 2 * qc + kk) * 16) : (unsigned)(((2 * qr) * IC + 2 * qc + 2 * kk) * 8);
    }
    bool oor = false;
    const int per_img = a.tiles_r * a.tiles_c;

    // the input patch of a tile: NPX pixels per thread, loaded one tile AHEAD into registers (the loads fly under the MFMAs of the
    // current tile) and split into the planes once every wave has left the current tile's MFMAs
    u32x4_t px[NPX];
    auto load_patch = [&](int tile) {
        const int b = tile / per_img, rem = tile - b * per_img, tr = rem / a.tiles_c, tc = rem - tr * a.tiles_c;
        const u32x4_t* const img = static_cast<const u32x4_t*>(a.in) + (size_t)b * a.Hp * a.Wp;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int e = t + 512 * i;
            const int pr_ = e / IC, pc_ = e - pr_ * IC;
            const int y = 4 * tr * PR + pr_, x = 4 * tc * PC + pc_;
            px[i] = u32x4_t{0u, 0u, 0u, 0u};
            if (tile < a.n_tiles && e < IR * IC && y < a.Hp && x < a.Wp) px[i] = img[(size_t)y * a.Wp + x];
        }
    };
    auto park_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int e = t + 512 * i;
            if (e < IR * IC) {
                if constexpr (F16) *reinterpret_cast<u32x4_t*>(planes + e * 16) = px[i];
                else if constexpr (H8) *reinterpret_cast<u32x2_t*>(planes + e * 8) = u32x2_t{px[i][0], px[i][1]};        // channels 0..3 of the pixel
                else {
                    u32x2_t part[3];
                    stem_split4<PARTS>(px[i], part);
#pragma unroll
                    for (int p = 0; p < PARTS; ++p) *reinterpret_cast<u32x2_t*>(planes + p * PLANE + e * 8) = part[p];
                }
            }
        }
    };
    load_patch(blockIdx.x);
    park_patch();
    __syncthreads();

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int b = tile / per_img, rem = tile - b * per_img, tr = rem / a.tiles_c, tc = rem - tr * a.tiles_c;
        const int pr0 = tr * PR, pc0 = tc * PC;
        const int r0 = 2 * pr0, c0 = 2 * pc0;                       // first conv output of the patch
        load_patch(tile + gridDim.x);
        // ---- 7 kernel rows x 2 groups x PARTS x 2 column blocks ----------------------------------------------------------
        f32x16 acc[2];                  // [r * NJ + j]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
#pragma unroll
        for (int g = 0; g < NGR; ++g) {
            uint4 fa[NR][PARTS], fb[NJ];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int p = 0; p < PARTS; ++p)
                    fa[r][p] = *reinterpret_cast<const uint4*>(planes + p * PLANE + a_basev[r] + ((g / NG) * IC + (F16 ? 2 : 4) * (g % NG)) * PXB);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (F16) fb[j] = *reinterpret_cast<const uint4*>(wf + ((g * 2 + j) * 64 + lane) * 16);
                else fb[j] = bw[g][j];
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int p = 0; p < PARTS; ++p)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[r * NJ + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa[r][p]), acc[r * NJ + j], 0, 0, 0);
        }
        // ---- activate, park: lane (q, kk), slot 4 g4 + r  <->  channel 32 j + 8 g4 + 4 kk + r ---------------------------------
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int jb = jb0 + j, q = qv[r];
                const f32x16& ac = acc[r * NJ + j];
                const int c = jb * 32 + 8 * g4 + 4 * kk;
                const float4 sc = *reinterpret_cast<const float4*>(tab + c), sh = *reinterpret_cast<const float4*>(tab + 64 + c);
                float4 y;
                y.x = fmaxf(ac[4 * g4 + 0] * sc.x + sh.x, 0.f);
                y.y = fmaxf(ac[4 * g4 + 1] * sc.y + sh.y, 0.f);
                y.z = fmaxf(ac[4 * g4 + 2] * sc.z + sh.z, 0.f);
                y.w = fmaxf(ac[4 * g4 + 3] * sc.w + sh.w, 0.f);
                oor = oor || (q_realv[r] && (!(y.x < 65504.0f) || !(y.y < 65504.0f) || !(y.z < 65504.0f) || !(y.w < 65504.0f)));
                const int chunk = jb * 8 + 2 * g4 + kk;
                *reinterpret_cast<float4*>(&ct[q * 64 + ((chunk ^ (q & 7)) << 2)]) = y;
            }
        __syncthreads();          // every wave has left the MFMAs (the planes are free) and the conv tile is complete
        park_patch();             // the next tile's input, split into the planes — beside the pooling of this one
        // ---- 3x3 stride-2 max-pool, Keras 'same': the window is clipped at the bottom / right edge of the conv output ------------
        for (int e = t; e < PR * PC * 16; e += 512) {
            const int ch4 = e & 15, pc = (e >> 4) & 15, pr = e >> 8;
            if (pr0 + pr >= a.PH || pc0 + pc >= a.PW) continue;
            float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                if (r0 + 2 * pr + dy >= a.CH) break;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (c0 + 2 * pc + dx >= a.CW) break;
                    const int qq = (2 * pr + dy) * CC + 2 * pc + dx;
                    const float4 v = *reinterpret_cast<const float4*>(&ct[qq * 64 + ((ch4 ^ (qq & 7)) << 2)]);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
            const size_t o = (((size_t)b * a.PH + pr0 + pr) * a.PW + pc0 + pc) * 64 + ch4 * 4;
            if constexpr (H8) store4<_Float16>(static_cast<_Float16*>(a.out) + o, m);         // (max commutes with the monotone rounding: one rounding, like the two launches)
            else *reinterpret_cast<float4*>(static_cast<float*>(a.out) + o) = m;
        }
        __syncthreads();          // the planes hold the next tile; the conv tile may be overwritten
    }
    if (a.range_flag && oor) atomicOr(a.range_flag, 1);
}

// in: padded (B, Hp, Wp) x 16 B; out: pooled (B, PH, PW, 64).  parts = 2 | 3 (split modes, fp32 tensors) | 1 (fp16 tensors).
void conv_stem_launch(hipStream_t s, const void* in, int B, int Hp, int Wp, const void* wgt, const float* scale, const float* shift, int CH, int CW,
                      void* out, int PH, int PW, int parts, int* range_flag, int n_cus, bool compact)
{
    StemArgs a;
    a.in = in; a.wgt = static_cast<const _Float16*>(wgt); a.scale = scale; a.shift = shift; a.out = out;
    a.B = B; a.Hp = Hp; a.Wp = Wp; a.CH = CH; a.CW = CW; a.PH = PH; a.PW = PW;
    a.tiles_r = (PH + PR - 1) / PR; a.tiles_c = (PW + PC - 1) / PC;
    a.n_tiles = B * a.tiles_r * a.tiles_c;
    a.range_flag = range_flag;
    const int slots = n_cus * ((parts == 1 && compact) ? 2 : 1);             // the compact fp16 form: two blocks per CU
    const int grid = a.n_tiles < slots ? a.n_tiles : slots;
    if (parts == 3) hipLaunchKernelGGL(k_conv_stem<3>, dim3(grid), dim3(512), 0, s, a);
    else if (parts == 2) hipLaunchKernelGGL(k_conv_stem<2>, dim3(grid), dim3(512), 0, s, a);
    else if (compact) hipLaunchKernelGGL((k_conv_stem<1, true>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((k_conv_stem<1, false>), dim3(grid), dim3(512), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace mrcnn
