// kernels_conv_halo.hip — the 3×3 stride-1 convolutions of the split modes (fp32 tensors, fp16 matrix cores): persistent
// 128×BN tiles whose input HALO is staged and split ONCE.
//
// Why: in the split modes a fp32 activation is turned into 2 / 3 fp16 parts in registers before it meets
// the matrix cores.  The 128-row implicit-GEMM kernel (kernels_conv.hip) stages one (tap, 32 channels) slab per K step, so a
// 3×3 layer fetches every input pixel NINE times from L2 (once per tap) and every wave column splits it again: the split
// VALU was 18 % of the time of the large 3×3 layers and the L2 → LDS traffic 9× the input — and every K step ends in a block
// barrier.  Here a tile's input region — the rows above / below and the columns left / right of its 128 output pixels — is
// loaded once per 16-channel slab, split once by the whole block on its way into LDS (fp16 hi / mid / lo planes), and the nine
// taps read SHIFTED windows of those planes; the filter fragments do not pass through LDS at all: the filters are re-tiled at
// load into 1-KB granules in MFMA-fragment order (conv_halo_pack), so a fragment is ONE coalesced 1-KB load straight into
// the registers of the wave that multiplies it (L2 / L1 hits: every CU walks the same granules).  Nothing a wave needs during
// a slab's nine taps is produced by another wave, so there is ONE barrier per slab (108 MFMAs per wave) instead of one per step.
//
//   out[m][n] = Σ_h Σ_tap Σ_{c < 16} A[pixel(m) + tap][16 h + c] · W[n][tap][16 h + c]
//
// K ORDER = (16-channel slab h, tap): this is the canonical summation order of every 3×3 stride-1 layer in the split modes,
// for EVERY batch size and tile width (the BN variants below only differ in which columns a block owns), so per-image
// results do not depend on the batch — the sharding contract.  Per (h, tap) step and accumulator: hi·w, mid·w, lo·w, each
// one v_mfma_f32_32x32x16_f16 (16 products + the fp32 accumulate).
//
// Per 16-channel slab of a 128 × 256 tile (8 waves as 2 × 4, wave tile 64 × 64):
//   activations  ≤ 528 input pixels × 64 B, buffer_load_dwordx4 into VGPRs one slab ahead (out-of-image pixels: an
//                out-of-range offset, the hardware returns zeros = the zero padding), split in registers, ds_write_b64 into
//                the other plane buffer while the nine taps of the current slab run;
//   filters      per step and wave two 1-KB fragment loads, three steps ahead (four register sets);
//   MFMAs        9 × 3 × 4 per wave; activation fragments by ds_read_b128 (conflict-free half-swizzle).
// Grids that would not fill the chip run 128 × 128 or 64 × 128 tiles (TN = 1, TM = 1: conv_halo_forward) — same K order, same bits.
// Blocks are persistent: one per CU, each XCD walks a contiguous run of tiles with its 32 CUs on 32 consecutive tiles (vertical
// neighbours share their halo rows in that XCD's L2).
//
// Epilogue: conv_epilogue_wave (conv_device.h) — wave-private LDS transpose, full-line stores — the same arithmetic, in the
// same order, as every other kernel of the family.
#include "conv_device.h"

#include <stdlib.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>

namespace mrcnn {

ConvScratch* conv_current_scratch();      // kernels_conv.hip: the calling thread's scratch (conv_set_scratch)

static int env_int_halo(const char* name, int dflt) { const char* e = knob_env(name); return e ? atoi(e) : dflt; }      // (honoured only with MRCNN_TEST_KNOBS=1)

static constexpr int HALO_MAX_SLOT = 640;        // LDS pixel slots of one plane (region rows x LDS pitch <= this)
static constexpr unsigned HALO_OOB = 0xC0000000u;
// Tile geometries (HaloArgs::geo) — which 64 / 128 output pixels a tile owns.  The K order, hence every output bit, is the same in all.
enum { HALO_GEO_LINEAR = 0,      // BM consecutive output pixels (any W; tiles may span rows and straddle images)
       HALO_GEO_ROW = 1,         // BM consecutive pixels inside ONE image row (OW % BM == 0): region 3 x (BM + 2)
       HALO_GEO_2ROWS = 2,       // 2 rows x 64 columns (BM = 128, OW % 64 == 0, OH % 2 == 0): region 4 x 66 = 264 pixels instead of 3 x 130 = 390
       HALO_GEO_4ROWS = 3 };     // 4 rows x 64 columns (BM = 256: the 64-column layers' 256 x 64 tiles; OH % 4 == 0): region 6 x 66 = 396 pixels

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct HaloArgs {
    ConvArgs a;
    const void* wgt_halo;        // conv_halo_pack layout
    int NH;                      // 16-channel slabs = Cin / 16
    int geo;                     // HALO_GEO_*
    int ecols;                   // region columns staged per region row (W + 2, or the cropped BM + 2 / 66)
    int pitch;                   // LDS pixel slots per region row (>= ecols; HALO_GEO_LINEAR pads it so that a wave's 32 pixels keep
                                 // distinct slots mod 16 across a row wrap: conflict-free ds_read_b128, halo_lds_pitch)
    int img_skew;                // HALO_GEO_LINEAR: extra slots per image boundary inside the region, so that consecutive output pixels keep
                                 // consecutive slots mod 16 across the boundary too (2 W + skew = 0 mod 16)
    int tiles_row, tiles_img;    // HALO_GEO_2ROWS: 64-column blocks per row, tiles per image
    int n_tiles;
    // fused 1x1 head (RPN: rpn_class_raw | rpn_bbox_pred on the ReLU'd output of this 3x3 layer): the layer's
    // own output is NOT stored; head_out / head_out2 receive columns [0, head_split) / [head_split, head_cols) + head_bias
    const void* head_w;          // conv_halo_pack_head layout (nullptr: no head)
    const float* head_bias;      // [32]
    float* head_out; float* head_out2;
    long head_out_sB, head_out_sP, head_out2_sB, head_out2_sP;
    int head_split, head_cols;
    float head_mul;              // head sums * head_mul + bias (ConvDesc::head_mul)
    // fused bottleneck tail (TAIL): the 1x1 layer that consumes this 3x3 layer's 256 ReLU'd output columns — its launch arguments
    // (epilogue fields, residual, output) and its filters in granule order (conv_halo_pack with one tap)
    ConvArgs t;
    const void* t_w;
    void* t_park;                // grid x 64 KB: where the waves of a tile's rows 64..127 park their activated 3x3 outputs during the first half's 1x1
};

// 4 fp32 → PARTS × 4 fp16 (the same round-to-nearest chain as split_hi_mid_lo / split_hi_lo of conv_device.h)
template <int PARTS>
__device__ __forceinline__ void split4(const u32x4 v, u32x2 (&out)[PARTS])
{
    const float a[4] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t h2 = cvt_pk_rne(a[2 * p], a[2 * p + 1]);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h2), "v"(a[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h2), "v"(a[2 * p + 1]));
        const uint32_t m2 = cvt_pk_rne(r0, r1);
        out[0][p] = h2;
        out[1][p] = m2;
        if constexpr (PARTS == 3) {
            float q0, q1;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(m2), "v"(r0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(m2), "v"(r1));
            out[2][p] = cvt_pk_rne(q0, q1);
        }
    }
}

// MAXPC = 16-B staging pieces per thread and slab: the region of a tile (rows x ecols pixels x 4 pieces) must fit MAXPC x 512.
// Every thread loads, splits and parks MAXPC pieces per slab whether the region needs them or not, so the launcher picks the
// smallest that fits: 2 (<= 256 pixels: the mask head, C5, the top pyramid levels), 3 (<= 384: the 4 x 66 regions of C4 and of
// the two-row tiles — rounds 1-3 staged 5 pieces for all of them), 5 (anything up to 640).
//
// TAIL (round 4): the bottleneck block's 1x1 `branch2c` (256 -> 1024 + shortcut + ReLU, Conversion/task.py:69-92) is
// computed by the block that has just produced the 128 x 256 tile of `branch2b` — those 256 columns ARE the full K of the 1x1.
// The ReLU'd tile never goes to memory: per 64-row half it is split ONCE into hi / mid / lo planes in LDS (96 KB, the 3x3 planes
// are dead by then), and every wave multiplies 64 rows x 64 output columns against filter fragments straight from memory
// (16 K groups x 12 MFMAs, fragment reads as in the 3x3 loop), twice per half; the epilogue is conv_epilogue_wave with the
// 1x1 layer's arguments.  K order = 16-channel groups ascending, parts hi / mid / lo per group, one running accumulator: the
// order of the 128-row kernel's 1x1 — the fused launch is BIT-IDENTICAL to the two launches it replaces, so whether a call
// fuses is a launch-time choice (grid fill), like a tile shape.
// WN = 2 (round 4, late): the eight waves as 4 x 2 — 128 x 64 tiles for the 64-column layers (C2's 3x3 64 -> 64), one 32 x 32 accumulator per wave
template <int PARTS, int TN, bool HEAD = false, bool DBG = false, int TM = 2, int MAXPC = 5, bool TAIL = false, int WN = 4>       // TM = 1: 64-row tiles for grids that would not fill the chip
__global__ __launch_bounds__(512, 2) void k_conv_halo(const HaloArgs ha)
{
    const ConvArgs& a = ha.a;
    static_assert(WN == 4 || (WN == 2 && TM == 1 && TN == 1 && !HEAD && !TAIL) || (WN == 1 && TM == 1 && TN == 2 && !HEAD && !TAIL),
                  "wave arrangement 2 x 4; 4 x 2 with one accumulator per wave; 8 x 1 with two (256 x 64 tiles)");
    static_assert(!HEAD || TN == 2, "the fused head walks the K groups of 256-column tiles");
    static_assert(!TAIL || (TM == 2 && TN == 2 && !HEAD), "the fused tail needs the whole 128 x 256 tile in one block");
    static_assert(MAXPC >= 2 && MAXPC <= 5, "staging pieces per thread");
    constexpr int WMR = 8 / WN;                           // wave rows
    constexpr int BM = WMR * TM * 32, BN = WN * TN * 32;
    constexpr int PLANE = (HALO_MAX_SLOT + 1) * 32;       // bytes of one part of one slab (+ one dump slot: pieces beyond the region write there, unconditionally)
    constexpr int PBUF = PARTS * PLANE;                   // one plane buffer (all parts)
    constexpr int STAGE = 8 * 32 * 36 * 4;                // the epilogue's wave-private tiles: they live in plane buffer 1
    constexpr int HPART = 4 * BM * 32 * 4;                // HEAD: the four wave columns' partial head sums of a tile (they live in the planes)
    constexpr int HRUN = HEAD ? BM * 32 * 4 : 0;          // HEAD: running head sum of the M tile over its N tiles
    constexpr int PLANES_ = 2 * PBUF > PBUF + STAGE ? 2 * PBUF : PBUF + STAGE;
    constexpr int TAILPL = 16 * 64 * 32;                  // TAIL: one part of the 64-row half: [16 K groups][64 rows] x 32 B
    constexpr int TAILN = 1024;                           // TAIL: output columns of the 1x1 layer
    constexpr int TPLANES = TAIL ? PARTS * TAILPL + STAGE : 0;            // ... all parts + the wave-private epilogue tiles
    constexpr int PLANES__ = HEAD && HPART > PLANES_ ? HPART : PLANES_;
    constexpr int PLANES = TPLANES > PLANES__ ? TPLANES : PLANES__;
    constexpr int TAB1 = TAIL ? 2 * TAILN * 4 : 0;        // TAIL: scale | shift of the 1x1 layer
    __shared__ __attribute__((aligned(16))) unsigned char smem[PLANES + 2 * 2 * BN * 4 + HRUN + TAB1];
    unsigned char* const planes = smem;
    float* const s_tab0 = reinterpret_cast<float*>(smem + PLANES);         // scale | shift of the tile's columns, by tile parity
    float* const h_run = reinterpret_cast<float*>(smem + PLANES + 2 * 2 * BN * 4);
    float* const tab1 = reinterpret_cast<float*>(smem + PLANES + 2 * 2 * BN * 4 + HRUN);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = WN == 4 ? wave >> 2 : (WN == 2 ? wave >> 1 : wave), wn = WN == 4 ? wave & 3 : (WN == 2 ? wave & 1 : 0);
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const int NH = ha.NH, NS = NH * 9;

    // ---- the tiles of this block: XCD x owns a contiguous run, its CUs take consecutive tiles -----------------------
    // (HEAD: the unit handed to a block is an M tile; its N tiles run back to back on the same block, which sums the head over them)
    const int T = HEAD ? a.tiles_m : ha.n_tiles;
    const int n_inner = HEAD ? a.tiles_n : 1;
    const int nb = gridDim.x;
    const int bid = blockIdx.x;
    int t_first, t_end, t_step;
    {
        const int q = T >> 3, r8 = T & 7;
        const int xcd = bid & 7, j = bid >> 3;
        const int lo = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
        const int cnt = q + (xcd < r8 ? 1 : 0);
        const int per = nb >> 3;                          // blocks per XCD (the host launches a multiple of 8 blocks, or fewer than 8)
        t_first = lo + j; t_end = lo + cnt; t_step = per;
        if (nb < 8) { t_first = bid; t_end = T; t_step = nb; }
    }

    typedef unsigned srd_t __attribute__((ext_vector_type(4)));
    srd_t srdB;
    {
        const unsigned long long wa = (unsigned long long)(uintptr_t)ha.wgt_halo;
        srdB[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
        srdB[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
        srdB[2] = 0xffffffffu;
        srdB[3] = 0x00020000u;
    }
    const unsigned vlane16 = (unsigned)lane * 16u;
    srd_t srdT;                     // TAIL: the 1x1 layer's filters (granule order)
    {
        const unsigned long long ta_ = (unsigned long long)(uintptr_t)(TAIL ? ha.t_w : ha.wgt_halo);
        srdT[0] = __builtin_amdgcn_readfirstlane((unsigned)ta_);
        srdT[1] = __builtin_amdgcn_readfirstlane((unsigned)(ta_ >> 32) & 0xffffu);
        srdT[2] = 0xffffffffu;
        srdT[3] = 0x00020000u;
    }
    if constexpr (TAIL) {           // the 1x1 layer's scale | shift, once per block (first read behind the tile's barriers)
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < TAILN / 4) {
            *reinterpret_cast<float4*>(&tab1[4 * t]) = ha.t.scale ? *reinterpret_cast<const float4*>(ha.t.scale + 4 * t) : one;
            *reinterpret_cast<float4*>(&tab1[TAILN + 4 * t]) = ha.t.shift ? *reinterpret_cast<const float4*>(ha.t.shift + 4 * t) : zero;
        }
    }
    // measurement-only ablations (ha.a.dbg = 0 in production; mrcnn_debug_set("conv_pp_dbg")): 1 no filter loads in the main
    // loop, 2 no barriers, 4 no activation-fragment reads, 8 no MFMAs, 16 no slab staging
    // (compiled in only in the DBG instantiation, which the launcher picks when a.dbg != 0: the production loop carries no switch)
    const int dbg = DBG ? a.dbg : 0;
    const bool dbg_nodma = dbg & 1, dbg_nobar = dbg & 2, dbg_nords = dbg & 4, dbg_nomma = dbg & 8, dbg_nostage = dbg & 16;
    const bool dbg_noloads = dbg & 64, dbg_nosplit = dbg & 128, dbg_nowrite = dbg & 256;       // 64 no slab loads, 128 no split, 256 no plane writes

    int tile_par = 0;
    for (int unit = t_first; unit < t_end; unit += t_step)
    for (int inner = 0; inner < n_inner; ++inner, tile_par ^= 1) {
        const int mt = HEAD ? unit : unit / a.tiles_n, nt = HEAD ? inner : unit - mt * a.tiles_n;
        const int n0 = nt * BN;
        float* const s_tab = s_tab0 + tile_par * 2 * BN;
        // ---- geometry of the tile's input region (uniform) --------------------------------------------------------
        const int Hp = a.H + 2;
        const int pitch = ha.pitch, ecols = ha.ecols;
        int m0, b0, gy_first, col0, rows;                 // m0: first output pixel (linear M index) of the tile — of its first row in HALO_GEO_2ROWS
        if (ha.geo == HALO_GEO_2ROWS || ha.geo == HALO_GEO_4ROWS) {
            const int trows = ha.geo == HALO_GEO_2ROWS ? 2 : 4;
            b0 = mt / ha.tiles_img;
            const int r = mt - b0 * ha.tiles_img, rp = r / ha.tiles_row, cb = r - rp * ha.tiles_row;
            m0 = b0 * ohw + trows * rp * a.OW + 64 * cb;
            gy_first = b0 * Hp + trows * rp + 1;
            col0 = 64 * cb - 1;
            rows = trows + 2;
        } else {
            m0 = mt * BM;
            const int m_last = (m0 + BM - 1 < a.M ? m0 + BM - 1 : a.M - 1);
            b0 = m0 / ohw;
            const int rem0 = m0 - b0 * ohw, oh0 = rem0 / a.OW, ow0 = rem0 - oh0 * a.OW;
            const int b1 = m_last / ohw, rem1 = m_last - b1 * ohw, oh1 = rem1 / a.OW;
            gy_first = b0 * Hp + oh0 + 1;                                 // padded global row of the first output pixel
            const int gy_last = b1 * Hp + oh1 + 1;
            col0 = ha.geo == HALO_GEO_ROW ? ow0 - 1 : -1;                 // input column of region column 0
            rows = gy_last - gy_first + 3;
        }
        const int npx = rows * ecols;                                     // staged pixels: <= MAXPC * 128 (host-checked), rows * pitch <= HALO_MAX_SLOT
        // activations through a buffer resource that starts at image b0 and covers the (at most two) images the tile touches
        srd_t srdA;
        {
            const unsigned long long ia = (unsigned long long)(uintptr_t)(static_cast<const float*>(a.in) + (long)b0 * a.in_sB);
            const unsigned long long rest = (unsigned long long)(a.B - b0) * (unsigned long long)a.in_sB * 4ull;
            srdA[0] = __builtin_amdgcn_readfirstlane((unsigned)ia);
            srdA[1] = __builtin_amdgcn_readfirstlane((unsigned)(ia >> 32) & 0xffffu);
            srdA[2] = __builtin_amdgcn_readfirstlane((unsigned)(rest < 0x80000000ull ? rest : 0x80000000ull));
            srdA[3] = 0x00020000u;
        }
        // ---- this thread's staging pieces: (input pixel, 16-B quarter of its 64-B slab) → source offset, LDS address ----
        unsigned p_off[MAXPC], p_lds[MAXPC];
#pragma unroll
        for (int i = 0; i < MAXPC; ++i) {
            const int j = t + 512 * i;
            const int px = j >> 2, qt = j & 3;
            const int r = px / ecols, c = px - r * ecols;
            const int gy = gy_first - 1 + r;
            const int b = gy / Hp, y = gy - b * Hp - 1;
            const int x = col0 + c;
            const bool ok = px < npx && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && b < a.B;
            p_off[i] = ok ? (unsigned)(((long)(b - b0) * a.in_sB + (long)y * a.in_sH + (long)x * a.in_sW + qt * 4) * 4) : HALO_OOB;
            if (DBG && (a.dbg & 32)) p_off[i] = (unsigned)((px & 63) * 1024 + qt * 16);        // measurement only: a cache-hot source
            const int slot = r * pitch + c + (b - b0) * ha.img_skew;
            p_lds[i] = px < npx ? (unsigned)(slot * 32 + (((qt >> 1) ^ ((slot >> 3) & 1)) << 4) + (qt & 1) * 8) : (unsigned)(HALO_MAX_SLOT * 32 + qt * 8);
        }
        // ---- this lane's output pixels → slot of their tap (0,0) input pixel in the region; the wave's first output row -----
        int base_idx[TM];
        int wave_row0;                                    // linear M index of the wave's first output pixel (its TM * 32 pixels are consecutive)
        if (ha.geo == HALO_GEO_2ROWS || ha.geo == HALO_GEO_4ROWS) {
            if constexpr (WN == 4) {
                wave_row0 = m0 + wm * a.OW;
#pragma unroll
                for (int i = 0; i < TM; ++i) base_idx[i] = wm * pitch + i * 32 + l31;
            } else {                                      // four wave rows of 32 pixels: (tile row wm >> 1, column half wm & 1)
                wave_row0 = m0 + (wm >> 1) * a.OW + (wm & 1) * 32;
                base_idx[0] = (wm >> 1) * pitch + (wm & 1) * 32 + l31;
            }
        } else {
            wave_row0 = m0 + wm * (TM * 32);
            const int ow0 = col0 + 1;                     // HALO_GEO_ROW: the tile's first column
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = wave_row0 + i * 32 + l31;
                const int mm = m < a.M ? m : a.M - 1;
                const int b = mm / ohw, rem = mm - b * ohw, oh = rem / a.OW, ow = rem - oh * a.OW;
                base_idx[i] = ha.geo == HALO_GEO_ROW ? (ow - ow0) : (b * Hp + oh + 1 - gy_first) * pitch + ow + (b - b0) * ha.img_skew;
            }
        }
        // this wave's filter-fragment streams: granule (n0/32 + wn*TN + j), one KB per step
        unsigned sob[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) sob[j] = (unsigned)(((size_t)(n0 / 32 + wn * TN + j) * NS) * 1024u);

        // scale / shift of the tile's columns for the epilogue (double-buffered by tile parity: read after the main loop's barriers)
        if (t < BN / 2) {
            const int c = (t < BN / 4 ? t : t - BN / 4) * 4;
            const float* src = t < BN / 4 ? a.scale : a.shift;
            const float fill = t < BN / 4 ? 1.0f : 0.0f;
            *reinterpret_cast<float4*>(&s_tab[(t < BN / 4 ? 0 : BN) + c]) =
                src ? *reinterpret_cast<const float4*>(src + n0 + c) : make_float4(fill, fill, fill, fill);
        }

#define HALO_LOAD(I) if constexpr ((I) < MAXPC) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st[(I) < MAXPC ? (I) : 0]) : "v"(p_off[(I) < MAXPC ? (I) : 0]), "s"(srdA) : "memory");
#define HALO_LOADS()  { HALO_LOAD(0) HALO_LOAD(1) HALO_LOAD(2) HALO_LOAD(3) HALO_LOAD(4) }
#define HALO_ADVANCE() { _Pragma("unroll") for (int i = 0; i < MAXPC; ++i) p_off[i] += (p_off[i] < HALO_OOB ? 64u : 0u); }
#define HALO_PIN1(I)  if constexpr ((I) < MAXPC) asm volatile("" : "+v"(st[(I) < MAXPC ? (I) : 0]));
#define HALO_PIN()    { HALO_PIN1(0) HALO_PIN1(1) HALO_PIN1(2) HALO_PIN1(3) HALO_PIN1(4) }
#define HALO_WRITE1(BUF, I_)                                                                                     \
    if constexpr ((I_) < MAXPC) {                                                                                \
        constexpr int I = (I_) < MAXPC ? (I_) : 0;                                                               \
        u32x2 parts[PARTS];                                                                                      \
        if (DBG && dbg_nosplit) { _Pragma("unroll") for (int p = 0; p < PARTS; ++p) { parts[p][0] = st[I][0]; parts[p][1] = st[I][p]; } } \
        else split4<PARTS>(st[I], parts);                                                                        \
        if (DBG && dbg_nowrite) { _Pragma("unroll") for (int p = 0; p < PARTS; ++p) asm volatile("" ::"v"(parts[p])); } \
        else {                                                                                                   \
            _Pragma("unroll") for (int p = 0; p < PARTS; ++p)                                                    \
                *reinterpret_cast<u32x2*>(planes + (BUF) * PBUF + p * PLANE + p_lds[I]) = parts[p];              \
        }                                                                                                        \
    }
#define HALO_WRITE(BUF) { HALO_WRITE1(BUF, 0) HALO_WRITE1(BUF, 1) HALO_WRITE1(BUF, 2) HALO_WRITE1(BUF, 3) HALO_WRITE1(BUF, 4) }
// filter fragments of one step: TN coalesced 1-KB loads into a register set (asm: counted by hand)
#define HALO_BLOAD(BV, STEP_)                                                                                    \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                         \
            const unsigned so_ = sob[j] + (unsigned)(STEP_) * 1024u;                                             \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(BV[j]) : "v"(vlane16), "s"(srdB), "s"(so_) : "memory"); \
        }                                                                                                        \
    }
#define HALO_BPIN(BV) { _Pragma("unroll") for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(BV[j])); }
        u32x4 st[MAXPC];
        u32x4 bv0[TN], bv1[TN], bv2[TN], bv3[TN];

        // ---- prologue: slab 0 into plane buffer 0, filter fragments of steps 0 and 1 ---------------------------------------
        HALO_LOADS()
        HALO_ADVANCE()
        HALO_BLOAD(bv0, 0)
        HALO_BLOAD(bv1, 1)
        HALO_BLOAD(bv2, 2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HALO_PIN()
        HALO_BPIN(bv0)
        HALO_BPIN(bv1)
        HALO_BPIN(bv2)
        HALO_WRITE(0)
        __syncthreads();

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

        // ---- main loop ----------------------------------------------------------------------------------------------
        // Step s = (slab h, tap): the MFMAs run on filter set s % 4 while the filter fragments of step s + 3 are requested from
        // memory; once they are issued, the activation fragments of step s + 1 replace the ones just multiplied (the wave's
        // partner on the SIMD owns the matrix pipe meanwhile).
        // vmcnt bookkeeping of a wave, in issue order: TN filter loads per step, the five slab loads right behind those of tap 0.
        // Before step s + 1 multiplies, its filter fragments (requested in step s - 2) must have arrived: behind them come the
        // 2 TN loads of steps s - 1 and s and, in taps 0 to 2, the five slab loads — which the wait of tap 3 therefore retires.
        // ONE barrier per slab, after tap 7: the next slab's planes (written in tap 4 by every wave) are first read by the
        // fragment prefetch of tap 8, and the buffer they replace was last read by the prefetch of the previous slab's tap 7.
#define HALO_AFRAGS(AV, PB_, TAP_)                                                                               \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                         \
            const int idx_ = base_idx[i] + ((TAP_) / 3) * pitch + ((TAP_) % 3);                                  \
            const unsigned a_addr_ = (unsigned)(idx_ << 5) + (unsigned)(((kk << 4) ^ ((idx_ << 1) & 16)));       \
            _Pragma("unroll") for (int p = 0; p < PARTS; ++p) AV[i][p] = *reinterpret_cast<const uint4*>(planes + (PB_) * PBUF + p * PLANE + a_addr_); \
        }                                                                                                        \
    }
#define HALO_STEP(BVC, BVN3, TAP, PB)                                                                            \
    {                                                                                                            \
        /* the filter fragments of step + 3 (past the end: the last step's again — the loop carries no tail case) */  \
        if (!(DBG && dbg_nodma)) HALO_BLOAD(BVN3, step + 3 < NS ? step + 3 : NS - 1)                             \
        if ((TAP) == 0 && next_slab && !(DBG && dbg_noloads)) { HALO_LOADS() HALO_ADVANCE() }                    \
        if (!(DBG && dbg_nomma)) {                                                                               \
        _Pragma("unroll") for (int p = 0; p < PARTS; ++p)                                                        \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, BVC[j]), __builtin_bit_cast(f16x8, av[i][p]), acc[i][j], 0, 0, 0); \
        }                                                                                                        \
        /* the fragments of the next step replace the ones just multiplied (the wave's partner on the SIMD owns the matrix \
           pipe meanwhile): one register set; past the last step the read is harmless */                         \
        if (!(DBG && dbg_nords)) {                                                                               \
            if ((TAP) == 8) HALO_AFRAGS(av, (PB) ^ 1, 0)                                                         \
            else HALO_AFRAGS(av, PB, ((TAP) + 1) % 9)                                                            \
        }                                                                                                        \
        /* the next slab, piece by piece over taps 4..7 (a burst in one step would leave both waves of a SIMD in VALU) */    \
        if ((TAP) == 4 && next_slab) { HALO_PIN() HALO_WRITE1((PB) ^ 1, 0) HALO_WRITE1((PB) ^ 1, 4) }            \
        if ((TAP) == 5 && next_slab) HALO_WRITE1((PB) ^ 1, 1)                                                    \
        if ((TAP) == 6 && next_slab) HALO_WRITE1((PB) ^ 1, 2)                                                    \
        if ((TAP) == 7 && next_slab) HALO_WRITE1((PB) ^ 1, 3)                                                    \
        if (DBG && (dbg_nodma || dbg_noloads)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  \
        else if ((TAP) <= 2 && next_slab) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TN + MAXPC) : "memory");  \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TN) : "memory");                                       \
        HALO_BPIN(bv0) HALO_BPIN(bv1) HALO_BPIN(bv2) HALO_BPIN(bv3)                                              \
        if ((TAP) == 7 && !(DBG && dbg_nobar)) __syncthreads();                                                  \
        ++step;                                                                                                  \
    }
#define HALO_SLAB(B0, B1, B2, B3, PB)                                                                            \
    HALO_STEP(B0, B3, 0, PB) HALO_STEP(B1, B0, 1, PB) HALO_STEP(B2, B1, 2, PB) HALO_STEP(B3, B2, 3, PB) HALO_STEP(B0, B3, 4, PB) \
    HALO_STEP(B1, B0, 5, PB) HALO_STEP(B2, B1, 6, PB) HALO_STEP(B3, B2, 7, PB) HALO_STEP(B0, B3, 8, PB)
        uint4 av[TM][PARTS];
        HALO_AFRAGS(av, 0, 0)
        int step = 0;
        for (int h = 0; h < NH; h += 4) {          // four slabs per iteration: 36 steps, the four filter register sets rotate statically
            { const bool next_slab = !dbg_nostage;                 HALO_SLAB(bv0, bv1, bv2, bv3, 0) }      // steps 0.. 8: sets 0 1 2 3 0 1 2 3 0
            { const bool next_slab = !dbg_nostage;                 HALO_SLAB(bv1, bv2, bv3, bv0, 1) }      // steps 9..17: sets 1 2 3 0 ...
            { const bool next_slab = !dbg_nostage;                 HALO_SLAB(bv2, bv3, bv0, bv1, 0) }
            { const bool next_slab = h + 4 < NH && !dbg_nostage;   HALO_SLAB(bv3, bv0, bv1, bv2, 1) }
        }
        // the loop's last three steps requested filter fragments nobody multiplies (no tail case in the loop): they must have
        // landed before the epilogue reuses their registers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HALO_BPIN(bv0) HALO_BPIN(bv1) HALO_BPIN(bv2) HALO_BPIN(bv3)
#undef HALO_SLAB
#undef HALO_STEP
#undef HALO_AFRAGS
#undef HALO_BPIN
#undef HALO_BLOAD
#undef HALO_WRITE
#undef HALO_WRITE1
#undef HALO_PIN
#undef HALO_PIN1
#undef HALO_ADVANCE
#undef HALO_LOADS
#undef HALO_LOAD
        // Epilogue through a wave-private 32 x 36-float tile inside plane buffer 1 (free since the last slab's barrier; the next
        // tile's prologue only writes plane buffer 0 and the other s_tab, and its first write to buffer 1 comes after its own
        // prologue barrier, i.e. after every wave has left this epilogue): full-line stores, no block barrier.
        if constexpr (TAIL) {
            const ConvArgs& t1 = ha.t;
            unsigned char* const tpl = smem;                               // [part][K group][row] x 32 B, halves swapped when bit 3 of the row is set
            float* const tstage = reinterpret_cast<float*>(smem + PARTS * TAILPL) + wave * (32 * 36);
            const bool relu3 = a.act == ACT_RELU;
            bool oor = false;
            // y = act(acc * scale + shift) — the 3x3 layer's own epilogue arithmetic.  lane (pixel l31, kk), accumulator slot 4q + r
            // <-> channel 32 j + 8 q + 4 kk + r of this wave's 64 columns.  The waves of rows 0..63 (wm = 0) split their y into the
            // LDS planes at once; the waves of rows 64..127 park theirs (fp32, 16 KB per wave, one coalesced KB per store) in the
            // block's slice of a global scratch and fetch it back for the second half: carrying the 64 accumulator registers through
            // the first half's 1x1 left the compiler no room to keep loads in flight (every fragment read and filter load was
            // waited for on the spot: 179 us against 163 for the two launches, gpurun_out/r4f).
            if (t1.dbg & 4) { if (acc[0][0][0] == 123.456f) tab1[1] = acc[1][1][3]; continue; }      // measurement only: the 3x3 loop alone
            float4* const park = static_cast<float4*>(ha.t_park) + ((size_t)blockIdx.x * 4 + wn) * 16 * 64 + lane;
            auto stage_piece = [&](const float (&x)[4], int i, int j, int q) {
                u32x4 v = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
                u32x2 parts[PARTS];
                split4<PARTS>(v, parts);
                const int row = i * 32 + l31, g = wn * 4 + j * 2 + (q >> 1);
                const unsigned addr = (unsigned)((g * 64 + row) * 32 + ((((q & 1) ^ (row >> 3)) & 1) << 4) + 8 * kk);
#pragma unroll
                for (int p = 0; p < PARTS; ++p) *reinterpret_cast<u32x2*>(tpl + p * TAILPL + addr) = parts[p];
            };
            __syncthreads();          // every wave has left the 3x3 loop: its planes may be overwritten
            if (!(t1.dbg & 8))        // measurement only: 8 = no staging / parking of the activated tile
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = wn * 64 + j * 32 + 8 * q + 4 * kk;
                        const float4 sc = *reinterpret_cast<const float4*>(s_tab + cl), sh = *reinterpret_cast<const float4*>(s_tab + BN + cl);
                        float x[4] = {acc[i][j][4 * q + 0] * sc.x + sh.x, acc[i][j][4 * q + 1] * sc.y + sh.y,
                                      acc[i][j][4 * q + 2] * sc.z + sh.z, acc[i][j][4 * q + 3] * sc.w + sh.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (relu3) x[r] = fmaxf(x[r], 0.f);
                            oor = oor || !(fabsf(x[r]) < 65504.0f);
                        }
                        if (wm == 0) stage_piece(x, i, j, q);
                        else park[((i * 2 + j) * 4 + q) * 64] = make_float4(x[0], x[1], x[2], x[3]);
                    }
            if (a.range_flag && oor) atomicOr(a.range_flag, 1);
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && wm == 1 && !(t1.dbg & 8)) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 y = park[((i * 2 + j) * 4 + q) * 64];
                                const float x[4] = {y.x, y.y, y.z, y.w};
                                stage_piece(x, i, j, q);
                            }
                }
                __syncthreads();
                const int half_row0 = ha.geo == HALO_GEO_2ROWS ? m0 + h * a.OW : m0 + h * 64;
#pragma unroll 1
                for (int it = 0; it < 2; ++it) {
                    const int n1 = it * 512 + wave * 64;                   // this wave's 64 output columns of the 1x1
                    f32x16 acc1[2][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc1[i][j][e] = 0.0f;
                    // filter fragments straight from memory, two steps ahead (three register sets): granule (n1 / 32 + j), K group s,
                    // one coalesced KB per load.  The loads and their waits are inline asm like the 3x3 loop's: left to the compiler,
                    // every load was sunk to its use and waited for on the spot (gpurun_out/r4f: the fused launch 10 % SLOWER than the
                    // two it replaces).  vmcnt: at the end of step s the fragments of step s + 1 must have landed; behind them only the
                    // two loads of step s + 2 may be outstanding (stores of the previous epilogue retire independently: loads return in
                    // order, so "at most 2 outstanding" still means every load but the two newest is home).  All drained by step 14:
                    // the epilogue's own loads and stores are counted by the compiler.
                    const unsigned wofs = (unsigned)((n1 / 32) * 16 * 1024);
                    u32x4 bw[3][2];
#define TAIL_BLOAD(SET, S_)                                                                                      \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
            const unsigned so_ = wofs + (unsigned)((j * 16 + (S_)) * 1024);                                      \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bw[SET][j]) : "v"(vlane16), "s"(srdT), "s"(so_) : "memory"); \
        }                                                                                                        \
    }
#define TAIL_BPIN() { _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) _Pragma("unroll") for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(bw[q_][j])); }
                    TAIL_BLOAD(0, 0)
                    TAIL_BLOAD(1, 1)
                    const unsigned a_off = (unsigned)(l31 * 32 + ((kk ^ ((l31 >> 3) & 1)) << 4));
                    uint4 fa[2][PARTS];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int p = 0; p < PARTS; ++p) fa[i][p] = *reinterpret_cast<const uint4*>(tpl + p * TAILPL + (i * 32) * 32 + a_off);
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    TAIL_BPIN()
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        if (t1.dbg & 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; }       // measurement only (mrcnn_debug_set("conv_tail_dbg")): no 1x1 K loop
                        if (s + 2 < 16) TAIL_BLOAD((s + 2) % 3, s + 2)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int p = 0; p < PARTS; ++p)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bw[s % 3][j]), __builtin_bit_cast(f16x8, fa[i][p]),
                                                                                        acc1[i][j], 0, 0, 0);
                        // the fragments of the next K group replace the ones just multiplied (the SIMD's other wave owns the matrix pipe meanwhile)
                        if (s + 1 < 16) {
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int p = 0; p < PARTS; ++p)
                                    fa[i][p] = *reinterpret_cast<const uint4*>(tpl + p * TAILPL + ((s + 1) * 64 + i * 32) * 32 + a_off);
                        }
                        if (s + 2 < 16) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        TAIL_BPIN()
                    }
#undef TAIL_BPIN
#undef TAIL_BLOAD
                    if (!(t1.dbg & 1))          // measurement only: 1 = no 1x1 epilogue (residual loads, stores)
                        conv_epilogue_wave<TAILN, 2, 2, true>(t1, acc1, tstage, tab1, half_row0, 0, n1, lane);
                    else if (acc1[0][0][0] == 123.456f) tab1[0] = acc1[1][1][3];       // (keeps the accumulators alive)
                }
                __syncthreads();          // the planes are rewritten by the other half / the next tile's prologue
            }
        } else if constexpr (!HEAD) {
            conv_epilogue_wave<BN, TM, TN>(a, acc, reinterpret_cast<float*>(planes + PBUF) + wave * (32 * 36), s_tab, wave_row0, n0,
                                           wn * TN * 32, lane);
        } else {
            // ---- fused head: y = act(acc * scale + shift) stays in registers; head[pixel][0..32) += y[pixel][64 channels] . Wh ----
            // The accumulator layout IS an MFMA activation fragment up to a permutation of the 16 channels of a K group
            // (lane (pixel, kk), slot s <-> channel 16 g + 4 kk + (s & 3) + 8 (s >> 2)); the head filters are packed with the
            // same permutation (conv_halo_pack_head), so no data moves: split in registers, multiply.
            // Summation order (the definition of the fused head, the same for every batch): per wave, K groups ascending, parts
            // hi / mid / lo; then wave columns 0..3, then N tiles ascending, then the bias.
            const bool relu = a.act == ACT_RELU;
            bool out_of_range = false;
            f32x16 hp[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) hp[i][e] = 0.0f;
            const unsigned hw_off = (unsigned)(((n0 + wn * TN * 32) / 16) * 1024);         // this wave's first K group of the head
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const uint4 wf = *reinterpret_cast<const uint4*>(static_cast<const char*>(ha.head_w) + hw_off + (j * 2 + g) * 1024 + lane * 16);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        u32x4 v0, v1;
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2) {
                            const int cl = wn * TN * 32 + j * 32 + 8 * (2 * g + q2) + 4 * kk;
                            const float4 sc = *reinterpret_cast<const float4*>(s_tab + cl), sh = *reinterpret_cast<const float4*>(s_tab + BN + cl);
                            float x[4] = {acc[i][j][8 * g + 4 * q2 + 0] * sc.x + sh.x, acc[i][j][8 * g + 4 * q2 + 1] * sc.y + sh.y,
                                          acc[i][j][8 * g + 4 * q2 + 2] * sc.z + sh.z, acc[i][j][8 * g + 4 * q2 + 3] * sc.w + sh.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (relu) x[r] = fmaxf(x[r], 0.f);
                                out_of_range = out_of_range || !(fabsf(x[r]) < 65504.0f);
                                if (q2 == 0) v0[r] = __float_as_uint(x[r]); else v1[r] = __float_as_uint(x[r]);
                            }
                        }
                        u32x2 p0[PARTS], p1[PARTS];
                        split4<PARTS>(v0, p0);
                        split4<PARTS>(v1, p1);
#pragma unroll
                        for (int p = 0; p < PARTS; ++p) {
                            const u32x4 af = {p0[p][0], p0[p][1], p1[p][0], p1[p][1]};
                            hp[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, af), hp[i], 0, 0, 0);
                        }
                    }
                }
            if (a.range_flag && out_of_range) atomicOr(a.range_flag, 1);
            // partial sums of the four wave columns -> LDS (the planes: nobody reads them after the last slab's barrier)
            float* const part = reinterpret_cast<float*>(planes);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    // (16-B chunk c = 2q + kk of row r sits at chunk c ^ (r & 7): eight consecutive rows — one ds_write_b128 lane group —
                    // would otherwise hit ONE bank, 128 B apart: the 0.09 conflict rate of this instantiation in round 3 / VERDICT r3)
                    *reinterpret_cast<float4*>(&part[((wn * BM) + wm * (TM * 32) + i * 32 + l31) * 32 + (((2 * q + kk) ^ (l31 & 7)) << 2)]) =
                        make_float4(hp[i][4 * q], hp[i][4 * q + 1], hp[i][4 * q + 2], hp[i][4 * q + 3]);
            __syncthreads();
            const bool last = inner == n_inner - 1;
#pragma unroll
            for (int k = 0; k < BM / 16; ++k) {               // BM rows x 32 head columns, 512 threads
                const int o = t + 512 * k, row = o >> 5, col = o & 31;
                float sum = inner == 0 ? 0.0f : h_run[o];
                const int pc = ((((col >> 2) ^ (row & 7)) << 2) | (col & 3));          // the swizzled position of column col in row `row`
                sum += part[(0 * BM + row) * 32 + pc];
                sum += part[(1 * BM + row) * 32 + pc];
                sum += part[(2 * BM + row) * 32 + pc];
                sum += part[(3 * BM + row) * 32 + pc];
                if (!last) h_run[o] = sum;
                else {
                    const int m = ha.geo == HALO_GEO_2ROWS ? m0 + (row >> 6) * a.OW + (row & 63) : m0 + row;
                    if (m < a.M && col < ha.head_cols) {
                        const float y = sum * ha.head_mul + ha.head_bias[col];
                        const int b = m / ohw, pix = m - b * ohw;
                        if (col < ha.head_split) ha.head_out[(long)b * ha.head_out_sB + (long)pix * ha.head_out_sP + col] = y;
                        else ha.head_out2[(long)b * ha.head_out2_sB + (long)pix * ha.head_out2_sP + (col - ha.head_split)] = y;
                    }
                }
            }
            __syncthreads();          // the partials (= the planes) are rewritten by the next tile's prologue
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_conv_halo_lat — the same layer, the same K order (slab, tap, part), for grids that leave half the chip idle (round 4;
// single images — C4's 24 3x3 layers are 128 tiles of 64 x 128 on 256 CUs, each SIMD multiplying for two waves).
// 64 x 64 tiles, FOUR waves as 2 x 2 (one 32 x 32 accumulator each): twice the tiles, the same filter bytes per tile row as the
// 64 x 128 form (every tile of the big form reads its 128 columns once; here two tiles read 64 each).  A SIMD holds ONE wave, so
// nothing hides behind a partner — the schedule itself has to:
//   * filter fragments (one coalesced KB per step and wave) are requested ELEVEN steps ahead into twelve register sets: a step is
//     three MFMAs (~100 clk); three steps ahead, the big kernel's distance, is less than an L2 round trip;
//   * the activation fragments of step s + 1 are read from LDS BEFORE the MFMAs of step s issue (two register sets);
//   * the slab loads run TWO slabs ahead (two register sets): issued in tap 0 of slab h for slab h + 2, split and written in taps
//     4..7 of slab h + 1 — waiting for them never drains the filter queue.
// vmcnt of a wave, in issue order: one filter load per step; MAXPC slab loads behind the filter load of tap 0.  Before step s + 1
// multiplies, the filter load issued in step s - 10 must be home: behind it come the ten filter loads of steps s - 9 .. s, the slab
// loads of this slab's tap 0 and — in taps 0 and 1 — those of the previous slab's tap 0 (issued nine steps earlier, behind that
// step's filter load).  The slab loads consumed in tap 4 of slab h were issued in tap 0 of slab h - 1: twelve filter loads and the
// next slab loads lie behind them by the end of tap 3, more than any wait allows outstanding.
// Bit-identical to k_conv_halo (tests/test_gpu_conv_kernels.py: knob "halo_lat" 0 / 1; the full-size batch-8 vs batch-1 identity).
// ------------------------------------------------------------------------------------------------------------------------------
static constexpr int HALO_LAT_MAX_SLOT = 320;
template <int PARTS, int MAXPC>
__global__ __launch_bounds__(256, 2) void k_conv_halo_lat(const HaloArgs ha)
{
    const ConvArgs& a = ha.a;
    static_assert(MAXPC >= 3 && MAXPC <= 5, "staging pieces per thread");
    constexpr int BM = 64, BN = 64, NT = 256, D = 11, NSET = 12;
    constexpr int PLANE = (HALO_LAT_MAX_SLOT + 1) * 32;
    constexpr int PBUF = PARTS * PLANE;
    static_assert(PBUF >= 4 * 32 * 32 * 4, "the four wave-private epilogue tiles live in plane buffer 1");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PBUF + 2 * 2 * BN * 4];
    unsigned char* const planes = smem;
    float* const s_tab0 = reinterpret_cast<float*>(smem + 2 * PBUF);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const int NH = ha.NH, NS = NH * 9;

    const int T = ha.n_tiles;
    const int nb = gridDim.x;
    const int bid = blockIdx.x;
    int t_first, t_end, t_step;
    {
        const int q = T >> 3, r8 = T & 7;
        const int xcd = bid & 7, j = bid >> 3;
        const int lo = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
        const int cnt = q + (xcd < r8 ? 1 : 0);
        const int per = nb >> 3;
        t_first = lo + j; t_end = lo + cnt; t_step = per;
        if (nb < 8) { t_first = bid; t_end = T; t_step = nb; }
    }
    typedef unsigned srd_t __attribute__((ext_vector_type(4)));
    srd_t srdB;
    {
        const unsigned long long wa = (unsigned long long)(uintptr_t)ha.wgt_halo;
        srdB[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
        srdB[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
        srdB[2] = 0xffffffffu;
        srdB[3] = 0x00020000u;
    }
    const unsigned vlane16 = (unsigned)lane * 16u;

    int tile_par = 0;
    for (int unit = t_first; unit < t_end; unit += t_step, tile_par ^= 1) {
        const int mt = unit / a.tiles_n, nt = unit - mt * a.tiles_n;
        const int n0 = nt * BN;
        float* const s_tab = s_tab0 + tile_par * 2 * BN;
        // ---- geometry of the tile's input region (HALO_GEO_LINEAR / HALO_GEO_ROW, as k_conv_halo) -----------------------
        const int Hp = a.H + 2;
        const int pitch = ha.pitch, ecols = ha.ecols;
        const int m0 = mt * BM;
        const int m_last = (m0 + BM - 1 < a.M ? m0 + BM - 1 : a.M - 1);
        const int b0 = m0 / ohw;
        const int rem0 = m0 - b0 * ohw, oh0 = rem0 / a.OW, ow0 = rem0 - oh0 * a.OW;
        const int b1 = m_last / ohw, rem1 = m_last - b1 * ohw, oh1 = rem1 / a.OW;
        const int gy_first = b0 * Hp + oh0 + 1;
        const int gy_last = b1 * Hp + oh1 + 1;
        const int col0 = ha.geo == HALO_GEO_ROW ? ow0 - 1 : -1;
        const int rows = gy_last - gy_first + 3;
        const int npx = rows * ecols;                     // <= MAXPC * 64 pixels, rows * pitch (+ skew) <= HALO_LAT_MAX_SLOT: host-checked
        srd_t srdA;
        {
            const unsigned long long ia = (unsigned long long)(uintptr_t)(static_cast<const float*>(a.in) + (long)b0 * a.in_sB);
            const unsigned long long rest = (unsigned long long)(a.B - b0) * (unsigned long long)a.in_sB * 4ull;
            srdA[0] = __builtin_amdgcn_readfirstlane((unsigned)ia);
            srdA[1] = __builtin_amdgcn_readfirstlane((unsigned)(ia >> 32) & 0xffffu);
            srdA[2] = __builtin_amdgcn_readfirstlane((unsigned)(rest < 0x80000000ull ? rest : 0x80000000ull));
            srdA[3] = 0x00020000u;
        }
        unsigned p_off[MAXPC], p_lds[MAXPC];
#pragma unroll
        for (int i = 0; i < MAXPC; ++i) {
            const int j = t + NT * i;
            const int px = j >> 2, qt = j & 3;
            const int r = px / ecols, c = px - r * ecols;
            const int gy = gy_first - 1 + r;
            const int b = gy / Hp, y = gy - b * Hp - 1;
            const int x = col0 + c;
            const bool ok = px < npx && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && b < a.B;
            p_off[i] = ok ? (unsigned)(((long)(b - b0) * a.in_sB + (long)y * a.in_sH + (long)x * a.in_sW + qt * 4) * 4) : HALO_OOB;
            const int slot = r * pitch + c + (b - b0) * ha.img_skew;
            p_lds[i] = px < npx ? (unsigned)(slot * 32 + (((qt >> 1) ^ ((slot >> 3) & 1)) << 4) + (qt & 1) * 8) : (unsigned)(HALO_LAT_MAX_SLOT * 32 + qt * 8);
        }
        const int wave_row0 = m0 + wm * 32;
        int base_idx;
        {
            const int m = wave_row0 + l31;
            const int mm = m < a.M ? m : a.M - 1;
            const int b = mm / ohw, rem = mm - b * ohw, oh = rem / a.OW, ow = rem - oh * a.OW;
            base_idx = ha.geo == HALO_GEO_ROW ? (ow - ow0) : (b * Hp + oh + 1 - gy_first) * pitch + ow + (b - b0) * ha.img_skew;
        }
        const unsigned sob = (unsigned)(((size_t)(n0 / 32 + wn) * NS) * 1024u);
        if (t < BN / 2) {
            const int c = (t < BN / 4 ? t : t - BN / 4) * 4;
            const float* src = t < BN / 4 ? a.scale : a.shift;
            const float fill = t < BN / 4 ? 1.0f : 0.0f;
            *reinterpret_cast<float4*>(&s_tab[(t < BN / 4 ? 0 : BN) + c]) =
                src ? *reinterpret_cast<const float4*>(src + n0 + c) : make_float4(fill, fill, fill, fill);
        }

#define LAT_LOAD(S, I) if constexpr ((I) < MAXPC) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st[S][(I) < MAXPC ? (I) : 0]) : "v"(p_off[(I) < MAXPC ? (I) : 0]), "s"(srdA) : "memory");
#define LAT_LOADS(S) { LAT_LOAD(S, 0) LAT_LOAD(S, 1) LAT_LOAD(S, 2) LAT_LOAD(S, 3) LAT_LOAD(S, 4) }
#define LAT_ADVANCE() { _Pragma("unroll") for (int i = 0; i < MAXPC; ++i) p_off[i] += (p_off[i] < HALO_OOB ? 64u : 0u); }
#define LAT_PIN1(S, I) if constexpr ((I) < MAXPC) asm volatile("" : "+v"(st[S][(I) < MAXPC ? (I) : 0]));
#define LAT_PIN(S) { LAT_PIN1(S, 0) LAT_PIN1(S, 1) LAT_PIN1(S, 2) LAT_PIN1(S, 3) LAT_PIN1(S, 4) }
#define LAT_WRITE1(S, BUF, I_)                                                                                   \
    if constexpr ((I_) < MAXPC) {                                                                                \
        constexpr int I = (I_) < MAXPC ? (I_) : 0;                                                               \
        u32x2 parts[PARTS];                                                                                      \
        split4<PARTS>(st[S][I], parts);                                                                          \
        _Pragma("unroll") for (int p = 0; p < PARTS; ++p)                                                        \
            *reinterpret_cast<u32x2*>(planes + (BUF) * PBUF + p * PLANE + p_lds[I]) = parts[p];                  \
    }
#define LAT_BLOAD(SET, SO_) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bv[SET]) : "v"(vlane16), "s"(srdB), "s"(SO_) : "memory");
#define LAT_BPIN() { _Pragma("unroll") for (int q_ = 0; q_ < NSET; ++q_) asm volatile("" : "+v"(bv[q_])); }
#define LAT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define LAT_AFRAGS(AV, PB_, TAP_)                                                                                \
    {                                                                                                            \
        const int idx_ = base_idx + ((TAP_) / 3) * pitch + ((TAP_) % 3);                                         \
        const unsigned a_addr_ = (unsigned)(idx_ << 5) + (unsigned)(((kk << 4) ^ ((idx_ << 1) & 16)));           \
        _Pragma("unroll") for (int p = 0; p < PARTS; ++p) AV[p] = *reinterpret_cast<const uint4*>(planes + (PB_) * PBUF + p * PLANE + a_addr_); \
    }
#ifdef MRCNN_CONV_ABLATE      /* measurement build (make ablate; a.dbg = "conv_pp_dbg"): 1 no filter loads in the loop, 2 no MFMAs, 4 no fragment reads, 8 no slab staging, 16 no epilogue, 32 no loop */
#define LAT_ABL(BIT) (!(a.dbg & (BIT)))
#else
#define LAT_ABL(BIT) true
#endif
        u32x4 st[2][MAXPC];
        u32x4 bv[NSET];
        // ---- prologue: slabs 0 and 1 requested, then the filter fragments of steps 0..10; slab 0 into plane buffer 0 -----
        LAT_LOADS(0)
        LAT_ADVANCE()
        const bool two = NH > 1;
        if (two) { LAT_LOADS(1) LAT_ADVANCE() }
        const unsigned so_last = sob + (unsigned)(NS - 1) * 1024u;
        unsigned so_next = sob;                           // byte offset of the newest fragment request (past the last step: the last one again)
#pragma unroll
        for (int q_ = 0; q_ < D; ++q_) {
            if (q_) so_next = so_next + 1024u < so_last ? so_next + 1024u : so_last;
            LAT_BLOAD(q_, so_next)
        }
        if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D + MAXPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        LAT_PIN(0)
        LAT_WRITE1(0, 0, 0) LAT_WRITE1(0, 0, 1) LAT_WRITE1(0, 0, 2) LAT_WRITE1(0, 0, 3) LAT_WRITE1(0, 0, 4)
        LAT_BARRIER();

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        uint4 av[2][PARTS];
        LAT_AFRAGS(av[0], 0, 0)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");        // the fragments of step 0 (the slab-1 loads, if any, are older)
        LAT_BPIN()
// One step = one (slab, tap): the requests first (the filter fragments of step + 11; tap 0: the slab two slabs ahead; the activation
// fragments of step + 1), then the step's dependent MFMAs BACK TO BACK — hi, mid, lo on one accumulator: any instruction between two
// of them costs a 43-clk cliff (MI355X_MICROARCH.md; placing the fillers in the MFMA shadows was tried: C4 23.3 -> 25.3 us,
// gpurun_out/r4y) — then the tap's share of the next slab's staging under the last MFMA, and the wait that makes step + 1's filter
// fragments valid.  taps 4..7 split and write pieces 0..3 of slab h + 1 (tap 3: piece 4).
#define LAT_PIECE(TAP) ((TAP) >= 4 && (TAP) <= 7 ? (TAP) - 4 : ((TAP) == 3 ? 4 : 99))
#define LAT_MFMA(K, P) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[(K) % NSET]), __builtin_bit_cast(f16x8, av[(K) & 1][P]), acc, 0, 0, 0);
#define LAT_STEP(K, TAP, PB, HP)                                                                                 \
    {                                                                                                            \
        constexpr int PI = LAT_PIECE(TAP) < MAXPC ? LAT_PIECE(TAP) : 0;                                          \
        constexpr bool HASP = LAT_PIECE(TAP) < MAXPC;                                                            \
        if (LAT_ABL(1)) { so_next = so_next + 1024u < so_last ? so_next + 1024u : so_last; LAT_BLOAD(((K) + D) % NSET, so_next) } \
        if ((TAP) == 0 && stage_now) { LAT_LOADS(HP) LAT_ADVANCE() }                                             \
        if (LAT_ABL(4)) {                                                                                        \
            if ((TAP) == 8) LAT_AFRAGS(av[((K) + 1) & 1], (PB) ^ 1, 0)                                           \
            else LAT_AFRAGS(av[((K) + 1) & 1], PB, ((TAP) + 1) % 9)                                              \
        }                                                                                                        \
        if (LAT_ABL(2)) { _Pragma("unroll") for (int p = 0; p < PARTS; ++p) LAT_MFMA(K, p) }                     \
        if (HASP && write_next && LAT_ABL(8)) { LAT_PIN1((HP) ^ 1, PI) LAT_WRITE1((HP) ^ 1, (PB) ^ 1, PI) }      \
        if (stage_now) {                                                                                         \
            if ((TAP) <= 1 && stage_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1 + 2 * MAXPC) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1 + MAXPC) : "memory");                            \
        } else {                                                                                                 \
            if ((TAP) <= 1 && stage_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1 + MAXPC) : "memory");   \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");                                    \
        }                                                                                                        \
        LAT_BPIN()                                                                                               \
        if ((TAP) == 7) LAT_BARRIER();                                                                           \
    }
#define LAT_SLAB(K0, PB, HP)                                                                                     \
    LAT_STEP((K0) + 0, 0, PB, HP) LAT_STEP((K0) + 1, 1, PB, HP) LAT_STEP((K0) + 2, 2, PB, HP) LAT_STEP((K0) + 3, 3, PB, HP) LAT_STEP((K0) + 4, 4, PB, HP) \
    LAT_STEP((K0) + 5, 5, PB, HP) LAT_STEP((K0) + 6, 6, PB, HP) LAT_STEP((K0) + 7, 7, PB, HP) LAT_STEP((K0) + 8, 8, PB, HP)
        for (int h = 0; h < NH; h += 4) {          // four slabs per iteration: 36 steps, the twelve filter sets and the two fragment sets rotate statically
            if (!LAT_ABL(32)) break;
            { const bool stage_now = h + 2 < NH && LAT_ABL(8), write_next = h + 1 < NH && LAT_ABL(8), stage_prev = h >= 1 && h + 1 < NH && LAT_ABL(8); LAT_SLAB(0, 0, 0) }
            { const bool stage_now = h + 3 < NH && LAT_ABL(8), write_next = h + 2 < NH && LAT_ABL(8), stage_prev = h + 2 < NH && LAT_ABL(8);           LAT_SLAB(9, 1, 1) }
            { const bool stage_now = h + 4 < NH && LAT_ABL(8), write_next = h + 3 < NH && LAT_ABL(8), stage_prev = h + 3 < NH && LAT_ABL(8);           LAT_SLAB(18, 0, 0) }
            { const bool stage_now = h + 5 < NH && LAT_ABL(8), write_next = h + 4 < NH && LAT_ABL(8), stage_prev = h + 4 < NH && LAT_ABL(8);           LAT_SLAB(27, 1, 1) }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last steps requested fragments nobody multiplies
        LAT_BPIN()
#undef LAT_SLAB
#undef LAT_STEP
#undef LAT_MFMA
#undef LAT_PIECE
#undef LAT_AFRAGS
#undef LAT_BARRIER
#undef LAT_BPIN
#undef LAT_BLOAD
#undef LAT_WRITE1
#undef LAT_PIN
#undef LAT_PIN1
#undef LAT_ADVANCE
#undef LAT_LOADS
#undef LAT_LOAD
        // epilogue through wave-private tiles in plane buffer 1 (the last slab read it, and its barrier is behind every wave; the next
        // tile's prologue writes buffer 0 and the other s_tab, and its first write to buffer 1 comes after its own prologue barrier)
        f32x16 accs[1][1];
        accs[0][0] = acc;
        if (!LAT_ABL(16)) { if (acc[0] == 123.456f) s_tab[0] = acc[3]; continue; }
        conv_epilogue_wave<BN, 1, 1>(a, accs, reinterpret_cast<float*>(planes + PBUF) + wave * (32 * 32), s_tab, wave_row0, n0, wn * 32, lane);
#undef LAT_ABL
    }
}

// ----------------------------------------------------------------------------------------------------------------
// filter re-tiling: [Npad][9][Cin] fp16 (the family's packing) → granules [Npad/32][Cin/16][9][1 KB] in MFMA-fragment order:
// the 16 B of lane (l31, kk) — filter row 32 g + l31, channels 16 h + 8 kk .. + 8 — at byte 16 · (32 kk + l31)
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_pack(const uint4* __restrict__ src, int Npad, int Cin, int taps, uint4* __restrict__ dst)
{
    const int NH = Cin / 16;
    const long total = (long)(Npad / 32) * NH * taps * 64;      // 16-B pieces
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int piece = (int)(e & 63);
        const long gs = e >> 6;                                   // (granule, step)
        const int step = (int)(gs % (NH * taps));
        const int g = (int)(gs / (NH * taps));
        const int h = step / taps, tap = step - h * taps;
        const int r = piece & 31, half = piece >> 5;             // piece = lane of the wave that will load it
        const int n = g * 32 + r;
        dst[e] = src[(((long)n * taps + tap) * Cin + 16 * h + 8 * half) / 8];
    }
}

// head filters [32][Cin] fp16 (rows = head columns) -> [Cin/16][1 KB]: lane (n, kk), slot s holds W[n][16 G + 4 kk + (s & 3) + 8 (s >> 2)]
// — the K permutation under which an accumulator of the main kernel is an activation fragment (k_conv_halo's head epilogue)
__global__ __launch_bounds__(256) void k_halo_pack_head(const uint16_t* __restrict__ src, int Cin, uint16_t* __restrict__ dst)
{
    const long total = (long)(Cin / 16) * 64 * 8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int sl = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const int G = (int)(e >> 9);
        const int n = lane & 31, kk = lane >> 5;
        dst[e] = src[(long)n * Cin + 16 * G + 4 * kk + (sl & 3) + 8 * (sl >> 2)];
    }
}

void conv_halo_pack_head(hipStream_t s, const void* wgt_std, int Npad, int Cin, DevBuf& out)
{
    MRCNN_REQUIRE(Npad == 32 && Cin % 16 == 0, MRCNN_ERR_SHAPE, "halo head packing: Npad %d (must be 32) / Cin %d", Npad, Cin);
    out.alloc((size_t)32 * Cin * 2);
    hipLaunchKernelGGL(k_halo_pack_head, dim3(64), dim3(256), 0, s, static_cast<const uint16_t*>(wgt_std), Cin, static_cast<uint16_t*>(out.p));
    HIP_CHECK(hipGetLastError());
}

void conv_halo_pack(hipStream_t s, const void* wgt_std, int Npad, int Cin, DevBuf& out, int taps)
{
    MRCNN_REQUIRE(Npad % 32 == 0 && Cin % 16 == 0 && (taps == 9 || taps == 1), MRCNN_ERR_SHAPE, "halo packing: Npad %d / Cin %d / taps %d", Npad, Cin, taps);
    const size_t bytes = (size_t)Npad * taps * Cin * 2;
    out.alloc(bytes);
    const long total = (long)bytes / 16;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_halo_pack, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(wgt_std), Npad, Cin, taps, static_cast<uint4*>(out.p));
    HIP_CHECK(hipGetLastError());
}

// ---- tile geometry (host) -------------------------------------------------------------------------------------------
// Rows of the input region a tile of bm CONSECUTIVE output pixels may touch (HALO_GEO_LINEAR): the rows its pixels lie in, one
// above and one below, and two zero rows per image boundary inside the tile.
static int halo_linear_rows(int H, int W, int bm)
{
    const int rows_touched = (bm - 1 + W - 1) / W + 1;
    const long ohw = (long)H * W;
    const int boundaries = ohw % bm == 0 ? 0 : (int)((bm - 1 + ohw - 1) / ohw);
    return rows_touched + 2 * boundaries + 2;
}

static int g_halo_lat = env_int_halo("MRCNN_HALO_LAT", 1);     // grids under 3/4 of the chip even at 64 x 128: 64 x 64 tiles in the latency form (k_conv_halo_lat; bit-identical)
static int g_halo_rounds = env_int_halo("MRCNN_HALO_ROUNDS", 1);     // 64-row tiles where they shorten the last round of a 128 x 128 grid
static int g_halo_n64 = env_int_halo("MRCNN_HALO_N64", 2);     // 64-column 3x3 layers on the halo kernel: 1 as 128 x 64 tiles, 2 also 256 x 64 where the level allows
static int g_halo_geo = env_int_halo("MRCNN_HALO_GEO", 1);     // 0: the round-3 geometries (one-row 3 x 130 regions, five staging pieces, pitch W + 2) — A/B and bit-identity tests

struct HaloGeo { int geo, ecols, pitch, img_skew, tiles_row, tiles_img, rows, maxpc, slots; bool ok; };

// Geometry of the tiles of a layer for block tiles of bm rows.  Which geometry is used never changes a result: every tile sums
// its K in the same (slab, tap, part) order.
static HaloGeo halo_geometry(int H, int W, int bm)
{
    HaloGeo g{};
    int slots_extra = 0;
    if (g_halo_geo == 0) {              // round 3
        const bool row = W % 128 == 0;
        g.geo = row ? HALO_GEO_ROW : HALO_GEO_LINEAR;
        g.ecols = g.pitch = row ? 130 : W + 2;
        g.rows = row ? 3 : halo_linear_rows(H, W, bm);
        g.maxpc = 5;
        g.ok = g.rows * g.ecols <= 528;
        return g;
    }
    if (bm == 256) {                    // (the 64-column layers' 256 x 64 tiles only: conv_halo_forward asks for it where it exists)
        if (W % 64 != 0 || H % 4 != 0) return g;
        g.geo = HALO_GEO_4ROWS; g.ecols = g.pitch = 66; g.rows = 6;
        g.tiles_row = W / 64; g.tiles_img = (H / 4) * g.tiles_row;
    } else if (bm == 128 && W % 64 == 0 && W >= 128 && H % 2 == 0) {
        g.geo = HALO_GEO_2ROWS; g.ecols = g.pitch = 66; g.rows = 4;
        g.tiles_row = W / 64; g.tiles_img = (H / 2) * g.tiles_row;
    } else if (W % bm == 0) {
        g.geo = HALO_GEO_ROW; g.ecols = g.pitch = bm + 2; g.rows = 3;
    } else {
        g.geo = HALO_GEO_LINEAR; g.ecols = W + 2; g.rows = halo_linear_rows(H, W, bm);
        // a wave's 32 consecutive output pixels wrap into the next region row when W % 32 != 0 and their slot index jumps by
        // pitch - W: a multiple of 16 keeps the 16 lanes of every ds_read_b128 group on distinct 16-B slots of the 256-B bank row
        // (the mask head's W = 14 with pitch 16 was a 2-way conflict on every fragment read: tools/lds_bank_model.py)
        // ... and across an image boundary inside a tile (three region rows further: the two zero rows between images) the slot
        // index jumps by 3 pitch - W + 1: img_skew extra slots per boundary make that = 1 mod 16 as well
        g.pitch = W % 32 == 0 ? W + 2 : W + 16;
        const long ohw = (long)H * W;
        const int boundaries = ohw % bm == 0 ? 0 : (int)((bm - 1 + ohw - 1) / ohw);
        g.img_skew = g.pitch == W + 16 ? (16 - (2 * W) % 16) % 16 : 0;
        if (g.rows * g.pitch + boundaries * g.img_skew > HALO_MAX_SLOT) { g.pitch = W + 2; g.img_skew = 0; }
        slots_extra = boundaries * g.img_skew;
    }
    const int px = g.rows * g.ecols;
    g.maxpc = px <= 256 ? 2 : px <= 384 ? 3 : 5;
    g.slots = g.rows * g.pitch + slots_extra;
    g.ok = px <= 640 && g.slots <= HALO_MAX_SLOT;
    return g;
}

bool conv_halo_eligible(const ConvDesc& d)
{
    const int wdtype = d.wdtype < 0 ? d.dtype : d.wdtype;
    const bool split = d.dtype == MRCNN_F32 && (wdtype == MRCNN_F16 || wdtype == MRCNN_F32X3);
    if (!split || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.padH != 1 || d.padW != 1) return false;
    // slabs in fours (the main loop is unrolled by four slabs); 128 output columns or more.  With exactly 128 the wave tile is 64 x 32 and
    // the activation-fragment reads per MFMA double: round 3 measured x0.92 against the 128-row kernel on C3's 128 -> 128 layers and kept
    // them there; with round 4's cheaper staging it is x1.08 (110 -> 102 us, gpurun_out/r4i) and they run here (their K order changes with
    // the kernel, once, for every batch)
    // (late round 4) exactly 64 columns — C2's 64 -> 64 layers, 9 taps re-staged and re-split per tap on the 64-column 128-row kernel
    // (169 us, 2.8 x their HBM floor): 128 x 64 tiles on the same eight waves as 4 x 2 (g_halo_n64)
    if (d.OH != d.H || d.OW != d.W || d.Cin % 64 != 0 || (d.Npad % 128 != 0 && !(d.Npad == 64 && g_halo_n64)) || d.Cout % 4 != 0) return false;
    if (d.deconv2 || d.out2 || d.sel_partial || d.act == ACT_SIGMOID || d.res) return false;
    if (d.H >= 32760 || d.W >= 32760 || (double)d.in_sB * 8.0 >= 2.0e9) return false;
    // A property of the layer, never of the batch: both tile heights the launcher may pick must have a geometry that fits.
    // (Round 3 asked for a region of <= 528 pixels of 128 consecutive output pixels; that bound is kept for the geometries it
    // admitted, so no layer changes its kernel — hence its K order — between rounds except the ones the two-row tiles ADD:
    // W % 64 == 0 with even H, e.g. the 192-wide P3 level of 1536² inputs.)
    const bool legacy_ok = (d.W % 128 == 0 ? 3 * 130 : halo_linear_rows(d.H, d.W, 128) * (d.W + 2)) <= 528;
    if (g_halo_geo == 0) return legacy_ok;
    const bool rows2_ok = d.W % 64 == 0 && d.H % 2 == 0;
    return (legacy_ok || rows2_ok) && halo_geometry(d.H, d.W, 128).ok && halo_geometry(d.H, d.W, 64).ok;
}

template <int PARTS, int TN, bool HEAD, int TM>
static void halo_launch_pc(hipStream_t s, const HaloArgs& ha, int maxpc, int grid)
{
    if (maxpc == 2) {
        if constexpr (!HEAD) hipLaunchKernelGGL((k_conv_halo<PARTS, TN, HEAD, false, TM, 2>), dim3(grid), dim3(512), 0, s, ha);
        else hipLaunchKernelGGL((k_conv_halo<PARTS, TN, HEAD, false, TM, 3>), dim3(grid), dim3(512), 0, s, ha);        // (no fused level has a region this small)
    } else if (maxpc == 3) hipLaunchKernelGGL((k_conv_halo<PARTS, TN, HEAD, false, TM, 3>), dim3(grid), dim3(512), 0, s, ha);
    else hipLaunchKernelGGL((k_conv_halo<PARTS, TN, HEAD, false, TM, 5>), dim3(grid), dim3(512), 0, s, ha);
}

template <int PARTS>
static void halo_launch(hipStream_t s, const HaloArgs& ha, int bm, int bn, int maxpc, int grid)
{
    if (ha.a.dbg && !ha.head_w && bn == 256 && PARTS == 3) {           // ablations (tools/halo_ablate.py)
        if (maxpc <= 3) hipLaunchKernelGGL((k_conv_halo<3, 2, false, true, 2, 3>), dim3(grid), dim3(512), 0, s, ha);
        else hipLaunchKernelGGL((k_conv_halo<3, 2, false, true, 2, 5>), dim3(grid), dim3(512), 0, s, ha);
        return;
    }
    if (bn == 64 && bm == 256) {        // 256 x 64 tiles, the eight waves as 8 x 1 (396-pixel regions: five staging pieces)
        hipLaunchKernelGGL((k_conv_halo<PARTS, 2, false, false, 1, 5, false, 1>), dim3(grid), dim3(512), 0, s, ha);
        return;
    }
    if (bn == 64) {                     // 128 x 64 tiles, the eight waves as 4 x 2
        if (maxpc == 2) hipLaunchKernelGGL((k_conv_halo<PARTS, 1, false, false, 1, 2, false, 2>), dim3(grid), dim3(512), 0, s, ha);
        else if (maxpc == 3) hipLaunchKernelGGL((k_conv_halo<PARTS, 1, false, false, 1, 3, false, 2>), dim3(grid), dim3(512), 0, s, ha);
        else hipLaunchKernelGGL((k_conv_halo<PARTS, 1, false, false, 1, 5, false, 2>), dim3(grid), dim3(512), 0, s, ha);
        return;
    }
    if (ha.head_w && bm == 64) hipLaunchKernelGGL((k_conv_halo<PARTS, 2, true, false, 1, 3>), dim3(grid), dim3(512), 0, s, ha);      // small batches: 64-row tiles (regions of <= 198 pixels)
    else if (ha.head_w) halo_launch_pc<PARTS, 2, true, 2>(s, ha, maxpc, grid);
    else if (bn == 256) halo_launch_pc<PARTS, 2, false, 2>(s, ha, maxpc, grid);
    // (128 x 128 with the waves as 4 x 2 and two accumulators each — one activation fragment per two MFMAs, twice the filter requests —
    //  was measured against this 2 x 4 arrangement on C3's layers: 107.6 -> 104.8 us, bit-identical; not worth six instantiations)
    else if (bm == 128) halo_launch_pc<PARTS, 1, false, 2>(s, ha, maxpc, grid);
    else halo_launch_pc<PARTS, 1, false, 1>(s, ha, maxpc, grid);
}

// the filter shapes conv_halo_eligible can accept: only those are re-tiled at load (engine.hip: pack_conv_oihw)
bool conv_halo_packable(int KH, int KW, int Cin, int Npad) { return KH == 3 && KW == 3 && Cin % 64 == 0 && (Npad % 128 == 0 || Npad == 64); }

bool conv_halo_debug_set(const char* key, int value)
{
    if (std::string(key) == "halo_geo") { g_halo_geo = value; return true; }
    if (std::string(key) == "halo_lat") { g_halo_lat = value; return true; }
    if (std::string(key) == "halo_n64") { g_halo_n64 = value; return true; }
    if (std::string(key) == "halo_rounds") { g_halo_rounds = value; return true; }
    return false;
}

// A fused head needs 256-column tiles (the head's K groups are walked per wave column) and enough M tiles to occupy the chip
// at ANY batch the layer may see — a property of the layer, not of the call: at least 32 M tiles per image.
bool conv_halo_head_eligible(const ConvDesc& d)
{
    return conv_halo_eligible(d) && d.Npad % 256 == 0 && d.Npad == d.Cout && ((long)d.OH * d.OW) % 128 == 0 && (long)d.OH * d.OW >= 4096 &&
           d.head_cols >= 1 && d.head_cols <= 32 && d.head_split >= 0 && d.head_split <= d.head_cols;
}

// the 1x1 filter shapes the fused tail accepts (re-tiled at load with one tap): 256 -> 1024
bool conv_halo_tail_packable(int KH, int KW, int Cin, int Npad) { return KH == 1 && KW == 1 && Cin == 256 && Npad == 1024; }
// ... and the 3x3 geometries: the tail is instantiated for the three-piece staging only (regions of <= 384 pixels: C4 at 1024²)
bool conv_halo_tail_geometry_ok(int H, int W) { const HaloGeo g = halo_geometry(H, W, 128); return g.ok && g.maxpc <= 3; }

// a: filled by conv_forward (M, strides, epilogue fields, vec_ok checked by the caller); returns the N tile used
// t1 / t_w != nullptr: the fused bottleneck tail (conv_forward_tail has checked the pair)
int conv_halo_forward(hipStream_t s, ConvArgs a, const ConvDesc& d, int parts, int n_cus, const ConvArgs* t1, const void* t_w)
{
    HaloArgs ha;
    ha.t_w = t_w;
    ha.t_park = nullptr;
    if (t1) ha.t = *t1;
    // tile shape: the largest whose tiles occupy the chip (the K order, hence the result, does not depend on it): 128 x 256,
    // then 128 x 128, then — small grids: batch 1, the top pyramid levels — 64 x 128
    int bm = 128, bn = d.Npad % 256 == 0 ? 256 : (d.Npad % 128 == 0 ? 128 : 64);
    // 64 columns: 256 x 64 tiles (the waves as 8 x 1, two accumulators each: half the activation-fragment reads per MFMA of the 128 x 64
    // arrangement) where four image rows x 64 columns tile the level and the grid still fills the chip
    if (bn == 64 && g_halo_n64 >= 2 && g_halo_geo && halo_geometry(d.H, d.W, 256).ok && (long)(a.M / 256) >= (long)n_cus * 2) bm = 256;
    if (!d.head_w && !t1 && bn != 64) {
        if (bn > 128 && (long)((a.M + 127) / 128) * (d.Npad / bn) < n_cus) bn = 128;
        if (bn == 128 && (long)((a.M + 127) / 128) * (d.Npad / bn) * 4 < (long)n_cus * 3) bm = 64;
        // ... and where the 128 x 128 grid's last round would run nearly empty (the mask head of a single image: 308 tiles = 2 rounds of
        // 256 blocks, the second 20 % full): 64 x 128 tiles when their rounds, at ~0.575 of a 128-row tile each, come out shorter
        if (bn == 128 && bm == 128 && g_halo_rounds) {
            const long t128 = (long)((a.M + 127) / 128) * (d.Npad / bn), t64 = (long)((a.M + 63) / 64) * (d.Npad / bn);
            const long r128 = (t128 + n_cus - 1) / n_cus, r64 = (t64 + n_cus - 1) / n_cus;
            if (r64 * 575 < r128 * 1000 && halo_geometry(d.H, d.W, 64).ok) bm = 64;
        }
    }
    // a fused head owns whole M tiles (its N tiles run back to back on one block): when the 128-row M tiles alone would leave a
    // quarter of the chip idle — single images at P3 / P4 — 64-row tiles double the blocks (same per-pixel summation order:
    // bit-identical, tests/test_gpu_fullsize.py batch 8 vs batch 1); levels whose rows are not a multiple of 64 wide keep 128
    if (d.head_w && (long)((a.M + 127) / 128) * 4 < (long)n_cus * 3 && d.W % 64 == 0) bm = 64;
    MRCNN_REQUIRE(!t1 || (bn == 256 && d.Npad == 256), MRCNN_ERR_INVALID, "fused tail: the 3x3 layer must have exactly 256 output columns");
    // grids that cover less than 3/8 of the chip even with 64 x 128 tiles — single images at C5 / P5: 64 x 64 tiles in the latency
    // form (four waves per block, deep prefetch: k_conv_halo_lat), when the 64-row region fits its smaller planes.  At one 32 x 32
    // accumulator per wave both forms are bound by the activation-fragment reads (3 KB of LDS per wave and step against three MFMAs:
    // the LDS port is busy 96 of the chain's 96 clk), so C4's 128 tiles gain nothing from becoming 256 (23.3 -> 23.3 us) and stay
    // on the eight-wave form; C5's 64 tiles do (42 -> 34 us).  ("halo_lat" 2: also grids under 3/4 — the A/B of that statement.)
    // A four-wave form with 64 x 128 tiles and two accumulators per wave (half the fragment reads per MFMA) was built and measured as
    // well: C4 23.3 -> 28.7 us, C5 42 -> 45 — with one wave per SIMD the alternating accumulators wait for each other's write-back
    // (a dependent MFMA one instruction behind its producer does not get the back-to-back forwarding); gpurun_out/r4z/lat2.txt.
    int lat_pc = 0;
    if (!d.head_w && !t1 && g_halo_lat && g_halo_geo && bm == 64 && bn == 128 && (long)((a.M + 63) / 64) * (d.Npad / 128) * 8 < (long)n_cus * 3 * g_halo_lat) {
        const HaloGeo g64 = halo_geometry(d.H, d.W, 64);
        const int px = g64.rows * g64.ecols;
        if (g64.ok && g64.geo != HALO_GEO_2ROWS && px <= 320 && g64.slots <= HALO_LAT_MAX_SLOT) { lat_pc = px <= 192 ? 3 : px <= 256 ? 4 : 5; bn = 64; }
    }
    const int tiles_m = (a.M + bm - 1) / bm;
    a.tiles_m = tiles_m;
    a.tiles_n = d.Npad / bn;
    a.direct = 1;
    ha.a = a;
    ha.wgt_halo = d.wgt_halo;
    ha.NH = d.Cin / 16;
    const HaloGeo g = halo_geometry(d.H, d.W, bm);
    MRCNN_REQUIRE(g.ok, MRCNN_ERR_SHAPE, "halo kernel: no tile geometry for %dx%d (conv_halo_eligible should have said so)", d.H, d.W);
    ha.geo = g.geo; ha.ecols = g.ecols; ha.pitch = g.pitch; ha.img_skew = g.img_skew; ha.tiles_row = g.tiles_row; ha.tiles_img = g.tiles_img;
    ha.n_tiles = a.tiles_m * a.tiles_n;
    ha.head_w = d.head_w; ha.head_bias = d.head_bias; ha.head_out = d.head_out; ha.head_out2 = d.head_out2;
    ha.head_out_sB = d.head_out_sB; ha.head_out_sP = d.head_out_sP; ha.head_out2_sB = d.head_out2_sB; ha.head_out2_sP = d.head_out2_sP;
    ha.head_split = d.head_split; ha.head_cols = d.head_cols; ha.head_mul = d.head_mul;
    const int units = d.head_w ? a.tiles_m : ha.n_tiles;
    int grid = units < n_cus ? units : n_cus;
    if (grid >= 8) grid &= ~7;                  // a multiple of 8: every XCD runs the same number of blocks
    if (t1) {
        ha.t.direct = 1;
        // the parking scratch: 64 KB per block, in the launch owner's ConvScratch (kernels.h), grown on demand outside stream captures
        {
            ConvScratch* const sc = conv_current_scratch();
            MRCNN_REQUIRE(sc, MRCNN_ERR_INVALID, "fused tail without a ConvScratch (conv_forward_tail checks)");
            if (sc->park.bytes < (size_t)grid * 65536) {
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                HIP_CHECK(hipStreamIsCapturing(s, &cap));
                MRCNN_REQUIRE(cap == hipStreamCaptureStatusNone, MRCNN_ERR_INVALID, "fused tail: the parking buffer must exist before a stream capture (run the predict once eagerly)");
                HIP_CHECK(hipStreamSynchronize(s));
                sc->park.alloc((size_t)grid * 65536);
            }
            ha.t_park = sc->park.p;
        }
        MRCNN_REQUIRE(g.maxpc <= 3, MRCNN_ERR_INVALID, "fused tail: the tile's input region needs more than three staging pieces (conv_halo_tail_geometry_ok)");
        if (parts == 3) hipLaunchKernelGGL((k_conv_halo<3, 2, false, false, 2, 3, true>), dim3(grid), dim3(512), 0, s, ha);
        else hipLaunchKernelGGL((k_conv_halo<2, 2, false, false, 2, 3, true>), dim3(grid), dim3(512), 0, s, ha);
        return bn;
    }
    if (lat_pc) {
        grid = units < 2 * n_cus ? units : 2 * n_cus;       // 63 KB of LDS per block: two fit a CU
        if (grid >= 8) grid &= ~7;
#define MRCNN_LAT(P, PC) hipLaunchKernelGGL((k_conv_halo_lat<P, PC>), dim3(grid), dim3(256), 0, s, ha)
        if (parts == 3) { if (lat_pc == 3) MRCNN_LAT(3, 3); else if (lat_pc == 4) MRCNN_LAT(3, 4); else MRCNN_LAT(3, 5); }
        else { if (lat_pc == 3) MRCNN_LAT(2, 3); else if (lat_pc == 4) MRCNN_LAT(2, 4); else MRCNN_LAT(2, 5); }
#undef MRCNN_LAT
        return bn;
    }
    if (parts == 3) halo_launch<3>(s, ha, bm, bn, g.maxpc, grid);
    else halo_launch<2>(s, ha, bm, bn, g.maxpc, grid);
    return bn;
}

}  // namespace mrcnn
