// kernels_bneck.hip — an identity ResNet bottleneck block of the fp16 mode as ONE persistent launch on gfx950.
//
// Reference layers: the `res<stage><block>_branch2a / 2b / 2c` convolutions + BatchNorm + ReLU and the shortcut add of every
// non-first block of C2..C5 in MaskRCNN.mlmodel (Sources/maskrcnn/Python/Conversion/task.py:69-92; the layer list is the
// Matterport ResNet graph):
//     t1 = relu(bn(x  * W1))            1x1, 4C -> C
//     t2 = relu(bn(t1 * W2))            3x3, C -> C, zero padding 1
//     y  = relu(bn(t2 * W3) + x)        1x1, C -> 4C, + shortcut
// Three launches of the 128-row kernels move x three times (2a's input, 2c's residual, the next block) and t1 / t2 twice each,
// and every launch pays its own pipeline fill and epilogue drain on a grid that is one or two rounds deep (C4: 29 + 45 + 43 us
// at batch 8 for 73 GFLOP, 621 TFLOP/s; C2: 276 us against a 107 us HBM floor — profiles/r04_conv_shapes_f16.txt).  Here a
// block of 512 threads owns a TH x 16 output tile and runs the three GEMMs back to back with t1 and t2 ON CHIP:
//   phase A  t1 on the tile's halo region ((TH+2) x 18 pixels, recomputed per tile) — x staged through an LDS ring in 32-channel
//            steps (64-B rows), W1 beside it; the result goes to LDS as fp16 with out-of-image pixels ZERO (= the 3x3 layer's padding);
//   phase B  the 3x3 layer as 9 taps x C/64 steps: activation fragments are SHIFTED windows of t1 in LDS (no im2col, no reload),
//            W2 streams through the ring in 128-B rows; t2 goes back to LDS (over t1, which is dead by then);
//   phase C  the 1x1 expansion in four chunks of C output columns: t2 from LDS, W3 through the ring, epilogue + residual + ReLU
//            straight from the accumulators to HBM.
// Only the weights stream from L2 in phases B / C, so the kernel needs fewer vector-memory instructions per MFMA than any
// 128-row tile (the CU's memory front end, not the matrix pipe, bounds the fp16 kernels).
//
// BIT-IDENTICAL to the three launches (tests/test_gpu_bneck.py): every output sums its K in the same order — 64-channel steps,
// tap-major for the 3x3 layer, four 16-wide MFMA groups per step, filter fragment as first MFMA operand — t1 / t2 are rounded to
// fp16 exactly where the fp16 tensors of the three-launch form are, and the epilogue arithmetic is conv_epilogue_direct's.  Which
// form runs is decided by the layer's geometry only (conv_bneck_fusable), never by the batch.
//
// The output must NOT alias the input: a tile reads the halo pixels its neighbours own.  (The three-launch form writes in place;
// the engine gives fused stages a second tensor and ping-pongs.)
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "conv_device.h"

namespace mrcnn {

template <int C>
struct BnCfg {
    static constexpr int TH = C == 256 ? 8 : 16, TW = 16, P = TH * TW;      // output tile (pixels)
    static constexpr int HWD = TW + 2, HP = (TH + 2) * HWD;                 // halo region: pitch 18 (even: see the bank analysis below), pixels
    static constexpr int WM = P / 64, WN = 8 / WM;                          // 8 waves as WM x WN; wave tile 64 pixels x NB channels in phases B / C
    static constexpr int HPAD = 96 * WM;                                    // GEMM rows of phase A (>= HP; wave tile 96 x NB)
    static constexpr int NB = C / WN, TNW = NB / 32;
    static constexpr int CB = C / 64;                                       // 64-channel blocks of the mid tensors
    static constexpr int T1_BYTES = HP * C * 2, T2_BYTES = P * C * 2;
    static constexpr int A_X = HPAD * 64, A_W = C * 64, A_STAGE = A_X + A_W;     // phase A: 32 channels per step, 64-B rows
    static constexpr int B_STAGE = C * 128;                                      // phases B / C: 64 channels per step, 128-B filter rows
    static constexpr int RING = 2 * A_STAGE > 2 * B_STAGE ? 2 * A_STAGE : 2 * B_STAGE;
    // LDS map: the ring FIRST (LDS-DMA destinations stay below 128 KB), then t1 (t2 and phase C's table alias it), then the tables of phases A / B
    static constexpr int OFF_T1 = RING, OFF_TABC = OFF_T1 + T2_BYTES, OFF_TABAB = OFF_T1 + T1_BYTES;
    // C = 64: phase C's shortcut / output tiles of a chunk ([256 pixels][128 B], full-line traffic): one in the ring's idle part, one behind the tables
    static constexpr bool YTILE = C == 64;
    static constexpr int OFF_Y0 = 2 * B_STAGE, OFF_Y1 = OFF_TABAB + 4 * C * 4, Y_BYTES = P * C * 2;
    static constexpr int LDS = OFF_TABAB + 4 * C * 4 + (YTILE ? Y_BYTES : 0);
    static constexpr int NA = 4 * C / 32, NBS = 9 * CB, NCS = 4 * CB;       // K steps of the three phases
    static constexpr int NXD = HPAD / 16, NAD = NXD + C / 16;               // phase A: 1-KB DMAs per step (x rows | W1 rows)
    static_assert(HPAD >= HP && T2_BYTES + 8 * C * 4 * (C == 64 ? 2 : 1) <= T1_BYTES && LDS <= 163840 && NA % 2 == 0, "layout");
    static_assert(!YTILE || OFF_Y0 + Y_BYTES <= RING, "the first chunk tile lives in the ring's idle part");
};

// one identity block of a stage launched as ONE kernel (STAGE form): its filters in fragment order and its folded BatchNorm vectors
struct BneckLayer {
    const uint4 *w1f, *w2f, *w3f;
    const float *s1, *h1, *s2, *h2, *s3, *h3;
};

struct BneckArgs {
    const _Float16* x; _Float16* y;
    const _Float16 *w1, *w2, *w3, *ws;   // ws: the stage-entry form's shortcut filters [4C][C] (branch1)
    const uint4 *w1f, *w2f, *w3f;    // W1 / W2 / W3 in MFMA-fragment order (FRAG form)
    const float *s1, *h1, *s2, *h2, *s3, *h3, *ss, *hs;
    int B, H, W, tiles_x, tiles_y, ntiles;
    int* range_flag;
    int dbg;
    // STAGE form: block l reads pp[l & 1] and writes pp[(l + 1) & 1]; done[tile] = blocks this tile has completed (zeroed by the host before the launch)
    const BneckLayer* layers;
    int nlayers;
    _Float16* pp[2];
    unsigned* done;
};

typedef unsigned bn_srd_t __attribute__((ext_vector_type(4)));
#define BN_BLDS(VOFF, SRD, SOFF, DST)                                                                          \
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(VOFF), "s"(SRD), "s"(SOFF), "s"(DST) : "memory", "m0");
// the same DMA at DEVICE scope (sc1: fetched past the non-coherent L2 lines of this XCD) — STAGE form, whose input was written by other XCDs in this launch
#define BN_BLDS_DEV(VOFF, SRD, SOFF, DST)                                                                      \
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen sc1 lds" ::"v"(VOFF), "s"(SRD), "s"(SOFF), "s"(DST) : "memory", "m0");
#define BN_VMCNT0 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define BN_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0);

// fp32 runs of four channels (q = 2p | 2p + 1 of a 32 x 32 result tile) -> fp16, glued into 16 contiguous bytes per lane:
// lane (l31, 0) channels 16p .. 16p+7, lane (l31, 1) channels 16p+8 .. 16p+15 (conv_epilogue_direct's store layout)
__device__ __forceinline__ uint4 bn_pack16(const float4 va, const float4 vb)
{
    f16x4 ha4, hb4;
    ha4[0] = (_Float16)va.x; ha4[1] = (_Float16)va.y; ha4[2] = (_Float16)va.z; ha4[3] = (_Float16)va.w;
    hb4[0] = (_Float16)vb.x; hb4[1] = (_Float16)vb.y; hb4[2] = (_Float16)vb.z; hb4[3] = (_Float16)vb.w;
    const uint2 pa = __builtin_bit_cast(uint2, ha4), pb = __builtin_bit_cast(uint2, hb4);
    const auto sx = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
    return make_uint4(sx[0], sy[0], sx[1], sy[1]);
}
// the inverse: 16 B in the store layout -> the two fp32 runs a lane accumulates (the swap is an involution)
__device__ __forceinline__ void bn_unpack16(const uint4 r, float4& ra, float4& rb)
{
    const auto sx = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
    const f16x4 h0 = __builtin_bit_cast(f16x4, make_uint2(sx[0], sy[0])), h1 = __builtin_bit_cast(f16x4, make_uint2(sx[1], sy[1]));
    ra = make_float4((float)h0[0], (float)h0[1], (float)h0[2], (float)h0[3]);
    rb = make_float4((float)h1[0], (float)h1[1], (float)h1[2], (float)h1[3]);
}
__device__ __forceinline__ bool bn_bad(const float4 v)
{
    return !(fabsf(v.x) < 65504.0f) || !(fabsf(v.y) < 65504.0f) || !(fabsf(v.z) < 65504.0f) || !(fabsf(v.w) < 65504.0f);
}

// LDS bank notes (MI355X_MICROARCH.md §LDS: ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of
// each half-wave, 64 banks of 4 B):
//   * 128-B rows (ring of phases B / C, t2): 16-B chunk c of row r at c ^ ((r >> 1) & 7) — the kernel family's swizzle;
//   * 64-B rows (ring of phase A): chunk c of row r at c ^ ((r >> 2) & 3);
//   * t1: [64-channel block][halo pixel][128 B].  A 32-lane fragment covers two tile rows of 16 pixels, shifted by the tap: its
//     lane groups read halo columns {dx + 0..3, dx + 12..15} of one halo row and {dx + 4..11} of the next, so the swizzle key is
//     the halo COLUMN: chunk c of pixel (py, px) at c ^ ((px >> 1) & 7), and with an even pitch (18) the 128-B half of the
//     256-B bank line is px & 1 — sixteen distinct (half, key) pairs for any tap.
// FRAG (C = 256): in phases B / C a wave owns 32 output channels x all 128 pixels of the tile and streams ITS filter fragments straight
// from L2 into registers (bneck_pack_frag: one coalesced 1-KB load per 16-wide K group, eight groups ahead) — every filter byte is
// loaded once per block, as through the ring, but there is no ring, no DMA issue in the loop and NO BARRIER inside the two phases
// (t1 / t2 are read-only while they run): the two waves of a SIMD de-phase by themselves and keep the matrix pipe fed.
// FIRST (C = 64, stride 1): the stage's ENTRY block (res2a) — its input has C channels, not 4C, and the shortcut is not x but the 1x1
// convolution `branch1` of x (+ BatchNorm, no ReLU), which the three-launch form stores as an fp16 tensor and reads back as branch2c's
// residual.  Here phase A's K is C, and phase C multiplies the tile's own x pixels (one 32-KB LDS tile) with branch1's filters beside
// t2 x W3: y = relu((t2 W3) s3 + h3 + fp16((x Ws) ss + hs)) — the rounding of the shortcut where the tensor would have been: bit-identical.
// STAGE (FRAG form; round 6): ALL identity blocks of the stage in one launch.  The block that owns tile T runs T through
// block l = 0, 1, .. of the stage; block l of T starts when T's <= 9 neighbour tiles (3 x 3, same image) have PUBLISHED block l - 1 — their
// outputs are the halo pixels phase A reads, and they have finished reading the tensor block l overwrites (the stage ping-pongs between two
// tensors, and the neighbourhood is symmetric: the same condition covers both hazards).  done[tile] counts a tile's finished blocks; outputs are
// stored, and inputs loaded, at DEVICE scope (sc1: written through / fetched past this XCD's non-coherent L2 lines — the protocol of the 128-row
// kernel's K-chunk fold, kernels_conv.hip), a tile publishes behind vmcnt(0) + barrier with a device-scope atomic.  No launch gap, no drain
// behind the slowest tile, and the next block's first filter fragments are requested BEFORE the wait.  MEASURED EQUAL to one launch per block
// (profiles/r06_bneck_stage_ab.txt: 22 blocks at batch 8 2 242 against 2 268 us, whole model x1.000; de-phasing the images by up to a tile time
// gains nothing either): the "fixed" 25-29 us of a block are the tile's OWN dependent chain — cold operand fetches, the
// phase borders' epilogues and barriers — not launch gaps, and one tile per CU leaves nothing to overlap them with.  So the form is OPT-IN
// (mrcnn_debug_set("conv_bneck_stage", 1)); it stays as the evidence and as a tested building block.  Progress needs every block of the grid
// resident (grid <= CUs, one block per CU: bneck_stage_launch); a wait that outlasts ~0.1 s trips bit 1 of the range flag and ends the launch
// (the host reports MRCNN_ERR_HIP) instead of hanging.  Every tile runs exactly the arithmetic of the per-block launches: BIT-IDENTICAL.
template <int C, bool FRAG, bool FIRST = false, bool STAGE = false>
__global__ __launch_bounds__(512) void k_bneck_h(const BneckArgs a)
{
    static_assert(!FIRST || (C == 64 && !FRAG), "the stage-entry form is laid out for C = 64");
    static_assert(!STAGE || FRAG, "the whole-stage form exists for the fragment-streaming kernel");
    constexpr int DEV = STAGE ? 16 : 0;                    // cache policy of the buffer builtins that touch x / y: sc1 = device scope
    constexpr int CIN = FIRST ? C : 4 * C;                 // channels of the block's input
    static_assert(!FRAG || C == 256, "the fragment-streaming form is laid out for C = 256 (eight waves x 32 channels, 128-pixel tiles)");
    using K = BnCfg<C>;
    constexpr int HP = K::HP, HWD = K::HWD, WN = K::WN, NB = K::NB, TNW = K::TNW, CB = K::CB, P = K::P;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[K::LDS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, kk = lane >> 5;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);

    // ---- buffer resources (raw, stride 0): a lane offset beyond num_records deposits zeros in LDS ----
    constexpr unsigned OOB = 0x80000000u;        // (+ any K-step offset stays out of range, no wrap)
    const unsigned img_bytes = (unsigned)((size_t)a.H * a.W * CIN * 2);       // of x; y always has 4C channels
    bn_srd_t srdX, srdW1, srdW2, srdW3;
    auto mk = [](const void* p, unsigned bytes) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        bn_srd_t r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)u);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xffffu);
        r[2] = bytes;
        r[3] = 0x00020000u;
        return r;
    };
    srdW1 = mk(a.w1, (unsigned)(C * CIN * 2));
    srdW2 = mk(a.w2, (unsigned)(C * 9 * C * 2));
    srdW3 = mk(a.w3, (unsigned)(4 * C * C * 2));

    // ---- tables of phases A / B: scale | shift of t1, scale | shift of t2 (once per block) ----
    float* const tabAB = reinterpret_cast<float*>(smem + K::OFF_TABAB);
    float* const tabC = reinterpret_cast<float*>(smem + K::OFF_TABC);

    // ---- loop-invariant DMA lane geometry ----
    // phase A (64-B rows, 16 rows per DMA): DMA u = wave + 8 j; u < NXD: x rows 16u.., else W1 rows 16(u - NXD)..
    const int r16 = lane >> 2, c4 = (lane & 3) ^ ((lane >> 4) & 3);            // row inside the DMA, SOURCE chunk held at position lane & 3
    unsigned w1off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int u = wave + 8 * j;
        const int n = 16 * (u - K::NXD) + r16;
        w1off[j] = (unsigned)(n * CIN * 2 + c4 * 16);
    }
    // phases B / C (128-B rows, 8 rows per DMA): DMA d = wave + 8 j, j < CB
    unsigned w2off[CB], w3off[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int row = 8 * (wave + 8 * j) + (lane >> 3);
        const int c8 = (lane & 7) ^ ((row >> 1) & 7);
        w2off[j] = (unsigned)(row * 9 * C * 2 + c8 * 16);
        w3off[j] = (unsigned)(row * C * 2 + c8 * 16);
    }
    // ---- loop-invariant fragment addresses ----
    const int sw64 = (l31 >> 2) & 3, sw128 = (l31 >> 1) & 7;
    // phase A: x rows wm*96 + i*32 + l31, W1 rows wn*NB + j*32 + l31 (64-B rows); K group g: chunk 2g + kk
    const unsigned axr = (unsigned)((wm * 96 + l31) * 64), awr = (unsigned)(K::A_X + (wn * NB + l31) * 64);
    const unsigned a_c0 = (unsigned)(((0 + kk) ^ sw64) << 4), a_c1 = (unsigned)(((2 + kk) ^ sw64) << 4);
    // phases B / C: filter rows wn*NB + j*32 + l31 (128-B rows)
    const unsigned bwr = (unsigned)((wn * NB + l31) * 128);
    unsigned w_c[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) w_c[g] = (unsigned)(((2 * g + kk) ^ sw128) << 4);
    // phase B: output pixel p = wm*64 + i*32 + l31 -> (py, px) in the tile; tap (dy, dx) reads halo pixel (py + dy, px + dx)
    unsigned t1row[2];          // byte offset of halo pixel (py, px) in block 0 of t1
    int pxi[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wm * 64 + i * 32 + l31;
        const int py = p >> 4, px = p & 15;
        t1row[i] = (unsigned)(K::OFF_T1 + (py * HWD + px) * 128);
        pxi[i] = px;
    }
    // phase C: t2 rows p (128-B rows per 64-channel block)
    const unsigned t2row = (unsigned)(K::OFF_T1 + (wm * 64 + l31) * 128);

    bool range_trip = false;
    const int nblocks = a.ntiles;
    const int q8 = nblocks >> 3, r8 = nblocks & 7;
    __shared__ int s_abort;
    if (t == 0) s_abort = 0;
    for (int layer = 0; layer < (STAGE ? a.nlayers : 1); ++layer) {
    // ---- this block's operands (STAGE: block `layer` of the stage, between the two ping-pong tensors) ----
    // (STAGE reads the record's fields where they are used — scalar loads — instead of holding nine pointers across the phases)
    const BneckLayer* const L = STAGE ? a.layers + layer : nullptr;
#define BN_L(F) (STAGE ? L->F : a.F)
    const _Float16* const Lx = STAGE ? a.pp[layer & 1] : a.x;
    _Float16* const Ly = STAGE ? a.pp[(layer + 1) & 1] : a.y;
    // tables of phases A / B (the previous tile's last readers passed the barrier that ends a tile)
    {
        const float *Ls1 = BN_L(s1), *Lh1 = BN_L(h1), *Ls2 = BN_L(s2), *Lh2 = BN_L(h2);
        for (int i = t; i < 4 * C; i += 512) {
            const int w = i / C, c = i - w * C;
            const float* src = w == 0 ? Ls1 : w == 1 ? Lh1 : w == 2 ? Ls2 : Lh2;
            tabAB[i] = src ? src[c] : ((w & 1) ? 0.0f : 1.0f);
        }
    }
    __syncthreads();
    for (int v = blockIdx.x; v < nblocks; v += gridDim.x) {
        // XCD-aware bijective walk: the blocks of one XCD (blockIdx & 7) own a contiguous run of tiles (neighbours share halo rows and L2)
        const int xcd = v & 7, local = v >> 3;
        const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
        const int per_img = a.tiles_x * a.tiles_y;
        const int b = tile / per_img, tr = tile - b * per_img;
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int y0 = ty * K::TH, x0 = tx * K::TW;
        const _Float16* const ximg = Lx + (size_t)b * a.H * a.W * CIN;
        _Float16* const yimg = Ly + (size_t)b * a.H * a.W * 4 * C;
        srdX = mk(ximg, img_bytes);

        // x rows of this thread's phase-A DMAs: halo pixel r -> image pixel (y0 - 1 + r / 18, x0 - 1 + r % 18)
        unsigned xoff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = wave + 8 * j;
            const int r = 16 * u + r16;
            const int py = r / HWD, px = r - py * HWD;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = r < HP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            xoff[j] = ok ? (unsigned)(((size_t)gy * a.W + gx) * CIN * 2 + c4 * 16) : OOB;
        }

        if constexpr (FRAG) {
        // =========================== phase A, fragment-streaming form ===========================
        // A wave owns 32 channels of t1 for ALL 192 halo rows (six 32-row tiles) and streams ITS W1 fragments into registers, eight
        // K groups ahead; x is shared by the eight waves, so it goes through LDS: 64-channel steps of [192 rows][128 B], two stages,
        // written from registers (three 16-B pieces per thread and step, requested a whole step ahead through the image's buffer
        // resource: out-of-image and padding rows read zeros).  Every load is one the compiler counts: no drained waits.
        constexpr int KG1 = 4 * C / 16, NS1 = 4 * C / 64, D1 = 8;
        f32x16 acc1[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[i][e] = 0.0f;
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(ximg), 0, (int)img_bytes, 0x00020000);
        typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
        int xvo[3];                  // byte offset of this thread's three pieces (row q >> 3, chunk q & 7; q = t + 512 m) at K step 0
        unsigned xls[3];             // ... and where they go in a stage
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int q = t + 512 * m, r = q >> 3, c = q & 7;
            const int py = r / HWD, px = r - py * HWD;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = r < HP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            xvo[m] = ok ? (int)(((size_t)gy * a.W + gx) * 4 * C * 2 + c * 16) : (int)OOB;
            xls[m] = (unsigned)(r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
        }
        const uint4* const wp1 = BN_L(w1f) + (size_t)wave * KG1 * 64 + lane;
        uint4 wq1[D1];
#pragma unroll
        for (int d = 0; d < D1; ++d) wq1[d] = wp1[d * 64];
        if constexpr (STAGE) {
            // the neighbours' previous block: lanes 0..8 of wave 0 poll one tile each (device-scope atomic loads), everybody waits at the barrier
            if (layer > 0) {
                if (t < 9) {
                    const int ny = ty + t / 3 - 1, nx = tx + t % 3 - 1;
                    if ((unsigned)ny < (unsigned)a.tiles_y && (unsigned)nx < (unsigned)a.tiles_x) {
                        const unsigned* const f = a.done + (b * per_img + ny * a.tiles_x + nx);
                        unsigned spins = 0;
                        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)layer) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > (1u << 16) || ((spins & 255u) == 0 && (__hip_atomic_load(a.range_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))) {
                                atomicOr(a.range_flag, 2);      // no progress (a grid that is not resident as a whole?): end the launch, the host fails the call
                                s_abort = 1;
                                break;
                            }
                        }
                    }
                }
                __syncthreads();
                if (s_abort) return;
            }
        }
        bn_u32x4 xr[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) xr[m] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvo[m], 0, DEV);
        __syncthreads();                               // (the previous tile's phase C has read the last of its LDS tile)
#pragma unroll
        for (int m = 0; m < 3; ++m) *reinterpret_cast<bn_u32x4*>(smem + xls[m]) = xr[m];
#pragma unroll
        for (int m = 0; m < 3; ++m) xr[m] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvo[m], 128, DEV);
        __syncthreads();
        const unsigned xfr = (unsigned)(l31 * 128);    // fragment rows i*32 + l31 of a stage
        for (int ks = 0; ks < ((a.dbg & 1) ? 1 : NS1); ks += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {              // two steps per trip: the stages and the eight fragment slots are compile-time
                const unsigned char* const sb = smem + h * (192 * 128);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = h * 4 + g;
                    f16x8 xf[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) xf[i] = *reinterpret_cast<const f16x8*>(sb + xfr + i * 32 * 128 + w_c[g]);
                    const f16x8 wf = __builtin_bit_cast(f16x8, wq1[d]);
#pragma unroll
                    for (int i = 0; i < 6; ++i) BN_MFMA(wf, xf[i], acc1[i])
                    int kgn = (ks + h) * 4 + g + D1;
                    kgn = kgn < KG1 ? kgn : KG1 - 1;
                    wq1[d] = wp1[kgn * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
                // step ks + h + 1 goes to the other stage (last read one step ago, behind the barrier), step ks + h + 2 is requested
                unsigned char* const so = smem + (h ^ 1) * (192 * 128);
#pragma unroll
                for (int m = 0; m < 3; ++m) *reinterpret_cast<bn_u32x4*>(so + xls[m]) = xr[m];
                {
                    int kn = ks + h + 2;
                    kn = kn < NS1 ? kn : NS1 - 1;
#pragma unroll
                    for (int m = 0; m < 3; ++m) xr[m] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvo[m], kn * 128, DEV);
                }
                __syncthreads();
            }
        }
        // epilogue A -> t1 (fp16, zero outside the image): channels 32 wave + 16 p + 8 kk .. of halo row i*32 + l31
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int r = i * 32 + l31;
            const int py = r / HWD, px = r - py * HWD;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool valid = r < HP;
            const bool inimg = valid && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned swz = (unsigned)((px >> 1) & 7);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cl = wave * 32 + 16 * p + 4 * kk;
                const float4 sa = *reinterpret_cast<const float4*>(tabAB + cl), sb_ = *reinterpret_cast<const float4*>(tabAB + cl + 8);
                const float4 ha = *reinterpret_cast<const float4*>(tabAB + C + cl), hb = *reinterpret_cast<const float4*>(tabAB + C + cl + 8);
                float4 va = make_float4(acc1[i][8 * p + 0], acc1[i][8 * p + 1], acc1[i][8 * p + 2], acc1[i][8 * p + 3]);
                float4 vb = make_float4(acc1[i][8 * p + 4], acc1[i][8 * p + 5], acc1[i][8 * p + 6], acc1[i][8 * p + 7]);
                va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                if (inimg) range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                else { va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va; }
                const uint4 pk = bn_pack16(va, vb);
                const int ch0 = wave * 32 + 16 * p + 8 * kk;
                const unsigned off = (unsigned)(K::OFF_T1 + ((ch0 >> 6) * HP + r) * 128) + ((((unsigned)(ch0 & 63) >> 3) ^ swz) << 4);
                if (valid) *reinterpret_cast<uint4*>(smem + off) = pk;
            }
        }
        } else {
        // =========================== phase A: t1 = relu(bn(x * W1)) on the halo region ===========================
        f32x16 acc1[3][TNW];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[i][j][e] = 0.0f;
        // stage 0 at the bottom of the ring, stage 1 at its top
#define BN_ISSUE_A(KS)                                                                                         \
    {                                                                                                          \
        const unsigned sb_ = lds0 + (((KS) & 1) ? (unsigned)(K::RING - K::A_STAGE) : 0u);                      \
        const unsigned ko_ = (unsigned)(KS) * 64u;                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
            const int u = wave + 8 * j;                                                                        \
            if (u < K::NXD) { BN_BLDS(xoff[j] + ko_, srdX, 0, sb_ + u * 1024) }                                \
            else if (u < K::NAD) { BN_BLDS(w1off[j] + ko_, srdW1, 0, sb_ + K::A_X + (u - K::NXD) * 1024) }     \
        }                                                                                                      \
    }
        BN_ISSUE_A(0)
        constexpr int NA_ = CIN / 32;
        static_assert(NA_ % 2 == 0, "phase A ends on ring stage 1");
        for (int ks = 0; ks < ((a.dbg & 1) ? 1 : NA_); ++ks) {
            BN_VMCNT0
            __syncthreads();                           // step ks has landed for everyone; everyone is done reading step ks - 1
            if (ks + 1 < NA_ && !(a.dbg & 1)) BN_ISSUE_A(ks + 1)
            const unsigned char* const sb = smem + ((ks & 1) ? (K::RING - K::A_STAGE) : 0);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const unsigned co = g ? a_c1 : a_c0;
                f16x8 xf[3], wf[TNW];
#pragma unroll
                for (int i = 0; i < 3; ++i) xf[i] = *reinterpret_cast<const f16x8*>(sb + axr + i * 32 * 64 + co);
#pragma unroll
                for (int j = 0; j < TNW; ++j) wf[j] = *reinterpret_cast<const f16x8*>(sb + awr + j * 32 * 64 + co);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j) BN_MFMA(wf[j], xf[i], acc1[i][j])
            }
        }
        // epilogue A -> t1 (fp16, zero outside the image).  Nobody reads t1 / t2 any more: every wave passed the barriers of
        // this phase after its last read of the previous tile's phase C.
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int r = wm * 96 + i * 32 + l31;
            const int py = r / HWD, px = r - py * HWD;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool valid = r < HP;
            const bool inimg = valid && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned swz = (unsigned)((px >> 1) & 7);
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int cl = wn * NB + j * 32 + 16 * p + 4 * kk;
                    const float4 sa = *reinterpret_cast<const float4*>(tabAB + cl), sb_ = *reinterpret_cast<const float4*>(tabAB + cl + 8);
                    const float4 ha = *reinterpret_cast<const float4*>(tabAB + C + cl), hb = *reinterpret_cast<const float4*>(tabAB + C + cl + 8);
                    float4 va = make_float4(acc1[i][j][8 * p + 0], acc1[i][j][8 * p + 1], acc1[i][j][8 * p + 2], acc1[i][j][8 * p + 3]);
                    float4 vb = make_float4(acc1[i][j][8 * p + 4], acc1[i][j][8 * p + 5], acc1[i][j][8 * p + 6], acc1[i][j][8 * p + 7]);
                    va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                    vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                    vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                    if (inimg) range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                    else { va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va; }
                    const uint4 pk = bn_pack16(va, vb);
                    const int ch0 = wn * NB + j * 32 + 16 * p + 8 * kk;
                    const unsigned off = (unsigned)(K::OFF_T1 + ((ch0 >> 6) * HP + r) * 128) + ((((unsigned)(ch0 & 63) >> 3) ^ swz) << 4);
                    if (valid) *reinterpret_cast<uint4*>(smem + off) = pk;
                }
        }
#undef BN_ISSUE_A
        }

        if constexpr (FRAG) {
        // =========================== phases B and C, fragment-streaming form ===========================
        constexpr int KG2 = 9 * C / 16, KG3 = C / 16, D = 8;          // 16-wide K groups of W2 / of one W3 round; fragments in flight
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
        const uint4* const wp2 = BN_L(w2f) + (size_t)wave * KG2 * 64 + lane;
        uint4 wq[D];
#pragma unroll
        for (int d = 0; d < D; ++d) wq[d] = wp2[d * 64];
        // output pixel p = i*32 + l31 -> (py, px) = (2i + (l31 >> 4), l31 & 15); tap (dy, dx) reads halo pixel (py + dy, px + dx)
        const unsigned t1r = (unsigned)(K::OFF_T1 + ((l31 >> 4) * HWD + (l31 & 15)) * 128);
        __syncthreads();                               // t1 complete
        // activation fragments are requested one K group ahead (two register sets), the filter fragment of group kg + 8 right after the
        // MFMAs that consumed this slot's (the slot's register is dead by then: no rotation copies)
        auto a_addr = [&](int kg) {        // byte address of the fragment of pixel tile 0 for K group kg (tile i: + i * 2 * HWD * 128)
            const int tap = kg >> 4, dy = tap / 3, dx = tap - 3 * dy, cb = (kg >> 2) & 3, g = kg & 3;
            const unsigned swx = (unsigned)(((((l31 & 15) + dx) >> 1) & 7) << 4) ^ (unsigned)(kk << 4);
            return t1r + (unsigned)((cb * HP + dy * HWD + dx) * 128) + (swx ^ (unsigned)(g << 5));
        };
        f16x8 af[2][4];
        {
            const unsigned ao = a_addr(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(smem + ao + i * 2 * HWD * 128);
        }
        for (int it = 0; it < ((a.dbg & 2) ? 1 : KG2 / D); ++it) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int kg = it * D + d;
                {
                    const int kn = kg + 1 < KG2 ? kg + 1 : kg;
                    const unsigned ao = a_addr(kn);
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[(d + 1) & 1][i] = *reinterpret_cast<const f16x8*>(smem + ao + i * 2 * HWD * 128);
                }
                const f16x8 wf = __builtin_bit_cast(f16x8, wq[d]);
#pragma unroll
                for (int i = 0; i < 4; ++i) BN_MFMA(wf, af[d & 1][i], acc[i])
                int kgn = kg + D;
                kgn = kgn < KG2 ? kgn : KG2 - 1;       // (the tail re-requests the last fragment: no branch in the stream)
                wq[d] = wp2[kgn * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // W3's first fragments ride under the epilogue; round r: output columns (8 r + wave) * 32 ..
        const uint4* const wp3 = BN_L(w3f) + (size_t)wave * KG3 * 64 + lane;
#pragma unroll
        for (int d = 0; d < D; ++d) wq[d] = wp3[d * 64];
        __syncthreads();                               // every wave is done with t1: t2 and phase C's table may overwrite it
        for (int i = t; i < 8 * C; i += 512) {
            const int w = i / (4 * C), c = i - w * 4 * C;
            const float* src = w == 0 ? BN_L(s3) : BN_L(h3);
            tabC[i] = src ? src[c] : (w ? 0.0f : 1.0f);
        }
        // epilogue B -> t2: channels 32 wave + 16 p + 8 kk .. of pixel i*32 + l31
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cl = wave * 32 + 16 * p + 4 * kk;
                const float4 sa = *reinterpret_cast<const float4*>(tabAB + 2 * C + cl), sb_ = *reinterpret_cast<const float4*>(tabAB + 2 * C + cl + 8);
                const float4 ha = *reinterpret_cast<const float4*>(tabAB + 3 * C + cl), hb = *reinterpret_cast<const float4*>(tabAB + 3 * C + cl + 8);
                float4 va = make_float4(acc[i][8 * p + 0], acc[i][8 * p + 1], acc[i][8 * p + 2], acc[i][8 * p + 3]);
                float4 vb = make_float4(acc[i][8 * p + 4], acc[i][8 * p + 5], acc[i][8 * p + 6], acc[i][8 * p + 7]);
                va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                const uint4 pk = bn_pack16(va, vb);
                const int ch0 = wave * 32 + 16 * p + 8 * kk;
                const unsigned off = (unsigned)(K::OFF_T1 + ((ch0 >> 6) * P + i * 32 + l31) * 128) + ((((unsigned)(ch0 & 63) >> 3) ^ (unsigned)sw128) << 4);
                *reinterpret_cast<uint4*>(smem + off) = pk;
            }
        __syncthreads();                               // t2 and the table are complete
        // Phase C moves its shortcut in and its output out in FULL LINES through a 64-KB tile in LDS (the ring's space, idle since phase A):
        //   ybuf[128 pixels][512 B = the round's 256 columns], 16-B chunk c of pixel p at position c ^ (p & 15) — conflict-free both for the
        //   epilogue's fragment-layout accesses (a lane group's 16 pixels are distinct mod 16) and for the row-wise store-out;
        //   round r: the shortcut of columns 256 r .. arrives by LDS-DMA (two pixels = 1 KB per instruction, requested a whole K loop ahead),
        //   each wave adds its 32 columns in place (read 16 B, unpack, + acc * scale + shift, ReLU, pack, write back), and the block writes
        //   the tile out, two pixels x 512 contiguous bytes per store instruction.  (Straight from the accumulators a store instruction
        //   covers 32 pixels x 32 B — scattered pieces issue 3x slower per CU, tools/probes/vmem_probe.hip — and phase C was the slowest
        //   of the three: 28 us for 7.8 us of MFMAs, gpurun_out/bneck_phases.txt.)
        // DMA / store geometry of this lane: instruction e = 8 wave + j moves pixels 2e, 2e + 1 = tile row `wave`, columns 2j + hi
        const int hi = lane >> 5;
        const unsigned yv = (unsigned)(hi * (4 * C * 2)) + (unsigned)((((lane & 31) ^ hi)) << 4);        // per-lane byte offset; instruction j: ^ (j << 5)
        const unsigned yrow = (unsigned)((((size_t)(y0 + wave) * a.W + x0) * (size_t)(4 * C)) * 2);      // byte offset of the tile row in the image
        const unsigned ybuf = lds0 + (unsigned)(wave * 8 * 1024);
#define BN_RES_DMA(R)                                                                                          \
    {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
            if constexpr (STAGE) { BN_BLDS_DEV(yv ^ (unsigned)(j << 5), srdX, yrow + (unsigned)(j * 2 * 4 * C * 2 + (R) * 512), ybuf + j * 1024) } \
            else { BN_BLDS(yv ^ (unsigned)(j << 5), srdX, yrow + (unsigned)(j * 2 * 4 * C * 2 + (R) * 512), ybuf + j * 1024) } \
        }                                                                                                      \
    }
        const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(yimg, 0, (int)img_bytes, 0x00020000);
        BN_RES_DMA(0)
        for (int r = 0; r < ((a.dbg & 4) ? 1 : 4); ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
            f16x8 af[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(smem + K::OFF_T1 + (i * 32 + l31) * 128 + w_c[0]);
#pragma unroll
            for (int kg = 0; kg < KG3; ++kg) {
                {
                    const int kn = kg + 1 < KG3 ? kg + 1 : kg, kb = kn >> 2, g = kn & 3;
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[(kg + 1) & 1][i] = *reinterpret_cast<const f16x8*>(smem + K::OFF_T1 + (kb * P + i * 32 + l31) * 128 + w_c[g]);
                }
                const f16x8 wf = __builtin_bit_cast(f16x8, wq[kg % D]);
#pragma unroll
                for (int i = 0; i < 4; ++i) BN_MFMA(wf, af[kg & 1][i], acc[i])
                // the stream runs on into the next round's fragments (granule ((8 r' + wave) * KG3 + kg')); the last round re-requests its tail
                int qn = r * KG3 + kg + D;
                qn = qn < 4 * KG3 ? qn : 4 * KG3 - 1;
                wq[kg % D] = wp3[(size_t)((qn / KG3) * 8 * KG3 + (qn % KG3)) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            BN_VMCNT0                                   // this wave's shortcut DMAs have landed (and, alas, its filter prefetch: DESIGN.md)
            __syncthreads();                            // ... and everybody's
            const int n0 = (r * 8 + wave) * 32;         // this wave's output columns of the round
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int cl = n0 + 16 * p + 4 * kk;
                    const float4 sa = *reinterpret_cast<const float4*>(tabC + cl), sb_ = *reinterpret_cast<const float4*>(tabC + cl + 8);
                    const float4 ha = *reinterpret_cast<const float4*>(tabC + 4 * C + cl), hb = *reinterpret_cast<const float4*>(tabC + 4 * C + cl + 8);
                    float4 va = make_float4(acc[i][8 * p + 0], acc[i][8 * p + 1], acc[i][8 * p + 2], acc[i][8 * p + 3]);
                    float4 vb = make_float4(acc[i][8 * p + 4], acc[i][8 * p + 5], acc[i][8 * p + 6], acc[i][8 * p + 7]);
                    va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                    vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                    // pixel i*32 + l31 = tile row 2i + (l31 >> 4), column l31 & 15; chunk 4 wave + 2p + kk of its 512 B
                    uint4* const slot = reinterpret_cast<uint4*>(smem + (i * 32 + l31) * 512 + (((wave * 4 + 2 * p + kk) ^ (l31 & 15)) << 4));
                    float4 ra, rb;
                    bn_unpack16(*slot, ra, rb);
                    va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
                    vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                    vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                    range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                    *slot = bn_pack16(va, vb);
                }
            __syncthreads();                            // the tile is complete
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
                const bn_u32x4 v = *reinterpret_cast<const bn_u32x4*>(smem + (wave * 8 + j) * 1024 + lane * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, srdY, (int)(yv ^ (unsigned)(j << 5)), (int)(yrow + (unsigned)(j * 2 * 4 * C * 2 + r * 512)), DEV);
            }
            __syncthreads();                            // ... and read out: the next round's shortcut may land
            if (r + 1 < 4) BN_RES_DMA(r + 1)
        }
#undef BN_RES_DMA
        } else {
        // =========================== phase B: t2 = relu(bn(conv3x3(t1))) ===========================
        f32x16 acc[2][TNW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
#define BN_ISSUE_W(OFFS, SRD, SOFF, ST)                                                                        \
    {                                                                                                          \
        const unsigned sb_ = lds0 + (unsigned)(ST) * (unsigned)K::B_STAGE;                                     \
        _Pragma("unroll") for (int j = 0; j < CB; ++j) { BN_BLDS(OFFS[j], SRD, SOFF, sb_ + (wave + 8 * j) * 1024) } \
    }
        __syncthreads();                               // t1 complete; the ring's last phase-A reads have retired
        BN_ISSUE_W(w2off, srdW2, 0u, 0)
        {
            int tap = 0, cb = 0, dy = 0, dx = 0;
            for (int s = 0; s < ((a.dbg & 2) ? 1 : K::NBS); ++s) {
                BN_VMCNT0
                __syncthreads();
                {
                    int cb1 = cb + 1, tap1 = tap;
                    if (cb1 == CB) { cb1 = 0; ++tap1; }
                    if (s + 1 < K::NBS) BN_ISSUE_W(w2off, srdW2, (unsigned)((tap1 * C + cb1 * 64) * 2), (s + 1) & 1)
                }
                const unsigned char* const sb = smem + (s & 1) * K::B_STAGE;
                const unsigned tapoff = (unsigned)((cb * HP + dy * HWD + dx) * 128);
                unsigned ax[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) ax[i] = (unsigned)((((pxi[i] + dx) >> 1) & 7) << 4) ^ (unsigned)(kk << 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x8 af[2], wf[TNW];
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(smem + t1row[i] + tapoff + (ax[i] ^ (unsigned)(g << 5)));
#pragma unroll
                    for (int j = 0; j < TNW; ++j) wf[j] = *reinterpret_cast<const f16x8*>(sb + bwr + j * 32 * 128 + w_c[g]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TNW; ++j) BN_MFMA(wf[j], af[i], acc[i][j])
                }
                if (++cb == CB) { cb = 0; ++tap; if (++dx == 3) { dx = 0; ++dy; } }
            }
        }
        __syncthreads();                               // every wave is done with t1: t2 and phase C's table may overwrite it
        // phase C's table: scale | shift of the 4C output columns
        for (int i = t; i < 8 * C; i += 512) {
            const int w = i / (4 * C), c = i - w * 4 * C;
            const float* src = w == 0 ? BN_L(s3) : BN_L(h3);
            tabC[i] = src ? src[c] : (w ? 0.0f : 1.0f);
        }
        // the first W3 step rides under the epilogue (the ring is free: the barrier above retired phase B's last reads)
        if constexpr (!FIRST) BN_ISSUE_W(w3off, srdW3, 0u, K::NBS & 1)
        // epilogue B -> t2
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p_ = wm * 64 + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int cl = wn * NB + j * 32 + 16 * p + 4 * kk;
                    const float4 sa = *reinterpret_cast<const float4*>(tabAB + 2 * C + cl), sb_ = *reinterpret_cast<const float4*>(tabAB + 2 * C + cl + 8);
                    const float4 ha = *reinterpret_cast<const float4*>(tabAB + 3 * C + cl), hb = *reinterpret_cast<const float4*>(tabAB + 3 * C + cl + 8);
                    float4 va = make_float4(acc[i][j][8 * p + 0], acc[i][j][8 * p + 1], acc[i][j][8 * p + 2], acc[i][j][8 * p + 3]);
                    float4 vb = make_float4(acc[i][j][8 * p + 4], acc[i][j][8 * p + 5], acc[i][j][8 * p + 6], acc[i][j][8 * p + 7]);
                    va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                    vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                    vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                    range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                    const uint4 pk = bn_pack16(va, vb);
                    const int ch0 = wn * NB + j * 32 + 16 * p + 8 * kk;
                    const unsigned off = (unsigned)(K::OFF_T1 + ((ch0 >> 6) * P + p_) * 128) + ((((unsigned)(ch0 & 63) >> 3) ^ (unsigned)sw128) << 4);
                    *reinterpret_cast<uint4*>(smem + off) = pk;
                }
        }

        // =========================== phase C: y = relu(bn(t2 * W3) + x), four chunks of C columns ===========================
        {
            int s = K::NBS;              // the ring's step counter runs on from phase B (stage = s & 1)
            size_t grow[2];              // element offset of the lane's pixel in the image
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p_ = wm * 64 + i * 32 + l31;
                grow[i] = ((size_t)(y0 + (p_ >> 4)) * a.W + (x0 + (p_ & 15))) * (size_t)(4 * C);
            }
            if constexpr (FIRST) {
            // ---- stage-entry block: the shortcut is branch1(x), formed here from the tile's own x pixels ----
            static_assert(TNW == 1 && CB == 1, "C = 64");
            constexpr int FST = 2 * K::B_STAGE;          // a stage = W3's step | Ws's step of a chunk (16 KB), two stages
            const bn_srd_t srdWs = mk(a.ws, (unsigned)(4 * C * C * 2));
            float* const tabS = tabC + 8 * C;            // scale | shift of branch1 (behind branch2c's table)
            for (int i = t; i < 8 * C; i += 512) {
                const int w = i / (4 * C), c = i - w * 4 * C;
                const float* src = w == 0 ? a.ss : a.hs;
                tabS[i] = src ? src[c] : (w ? 0.0f : 1.0f);
            }
            // the tile's x pixels -> LDS [256 pixels][128 B] (the chunk-tile area behind the tables), 16-B piece c of pixel p at c ^ ((p >> 1) & 7)
            {
                const int hi8 = lane >> 3;
                const unsigned xs0 = (unsigned)((((size_t)(y0 + 2 * wave) * a.W + x0) * (size_t)CIN) * 2);
                const unsigned xb_ = lds0 + (unsigned)K::OFF_Y1 + (unsigned)(wave * 4 * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned vo = (unsigned)(hi8 * (CIN * 2)) + (unsigned)((((lane & 7) ^ ((4 * (j & 1) + (lane >> 4)) & 7))) << 4);
                    BN_BLDS(vo, srdX, xs0 + (unsigned)((((j >> 1) * a.W + (j & 1) * 8) * (CIN * 2))), xb_ + j * 1024)
                }
            }
#define BN_ISSUE_W2(CH, ST)                                                                                    \
    {                                                                                                          \
        const unsigned sb_ = lds0 + (unsigned)(ST) * (unsigned)FST;                                            \
        BN_BLDS(w3off[0], srdW3, (unsigned)(((CH) * C * C) * 2), sb_ + wave * 1024)                            \
        BN_BLDS(w3off[0], srdWs, (unsigned)(((CH) * C * C) * 2), sb_ + K::B_STAGE + wave * 1024)               \
    }
            BN_ISSUE_W2(0, 0)
            f32x16 accs[2];
            for (int ch = 0; ch < ((a.dbg & 4) ? 1 : 4); ++ch) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc[i][0][e] = 0.0f; accs[i][e] = 0.0f; }
                BN_VMCNT0
                __syncthreads();                            // this chunk's filters (and, first time, t2 / the x tile / the tables) are in place
                if (ch + 1 < 4) BN_ISSUE_W2(ch + 1, (ch + 1) & 1)
                const unsigned char* const sb = smem + (ch & 1) * FST;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x8 af[2], xf[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        af[i] = *reinterpret_cast<const f16x8*>(smem + t2row + (i * 32) * 128 + w_c[g]);
                        xf[i] = *reinterpret_cast<const f16x8*>(smem + K::OFF_Y1 + (wm * 64 + i * 32 + l31) * 128 + w_c[g]);
                    }
                    const f16x8 wf = *reinterpret_cast<const f16x8*>(sb + bwr + w_c[g]);
                    const f16x8 wsf = *reinterpret_cast<const f16x8*>(sb + K::B_STAGE + bwr + w_c[g]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) { BN_MFMA(wf, af[i], acc[i][0]) BN_MFMA(wsf, xf[i], accs[i]) }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int pl = wm * 64 + i * 32 + l31;
                    const size_t go = ((size_t)(y0 + (pl >> 4)) * a.W + (x0 + (pl & 15))) * (size_t)(4 * C);
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int cl = ch * C + wn * NB + 16 * p + 4 * kk;
                        const float4 sa = *reinterpret_cast<const float4*>(tabC + cl), sb_ = *reinterpret_cast<const float4*>(tabC + cl + 8);
                        const float4 ha = *reinterpret_cast<const float4*>(tabC + 4 * C + cl), hb = *reinterpret_cast<const float4*>(tabC + 4 * C + cl + 8);
                        const float4 ta = *reinterpret_cast<const float4*>(tabS + cl), tb = *reinterpret_cast<const float4*>(tabS + cl + 8);
                        const float4 ua = *reinterpret_cast<const float4*>(tabS + 4 * C + cl), ub = *reinterpret_cast<const float4*>(tabS + 4 * C + cl + 8);
                        float4 va = make_float4(acc[i][0][8 * p + 0], acc[i][0][8 * p + 1], acc[i][0][8 * p + 2], acc[i][0][8 * p + 3]);
                        float4 vb = make_float4(acc[i][0][8 * p + 4], acc[i][0][8 * p + 5], acc[i][0][8 * p + 6], acc[i][0][8 * p + 7]);
                        float4 ra = make_float4(accs[i][8 * p + 0], accs[i][8 * p + 1], accs[i][8 * p + 2], accs[i][8 * p + 3]);
                        float4 rb = make_float4(accs[i][8 * p + 4], accs[i][8 * p + 5], accs[i][8 * p + 6], accs[i][8 * p + 7]);
                        va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                        vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                        ra.x = ra.x * ta.x + ua.x; ra.y = ra.y * ta.y + ua.y; ra.z = ra.z * ta.z + ua.z; ra.w = ra.w * ta.w + ua.w;
                        rb.x = rb.x * tb.x + ub.x; rb.y = rb.y * tb.y + ub.y; rb.z = rb.z * tb.z + ub.z; rb.w = rb.w * tb.w + ub.w;
                        range_trip = range_trip || bn_bad(ra) || bn_bad(rb);
                        // the shortcut as the fp16 tensor the three-launch form stores
                        ra.x = (float)(_Float16)ra.x; ra.y = (float)(_Float16)ra.y; ra.z = (float)(_Float16)ra.z; ra.w = (float)(_Float16)ra.w;
                        rb.x = (float)(_Float16)rb.x; rb.y = (float)(_Float16)rb.y; rb.z = (float)(_Float16)rb.z; rb.w = (float)(_Float16)rb.w;
                        va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
                        vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
                        va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                        vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                        range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                        *reinterpret_cast<uint4*>(yimg + go + ch * C + wn * NB + 16 * p + 8 * kk) = bn_pack16(va, vb);
                    }
                }
            }
#undef BN_ISSUE_W2
            } else if constexpr (K::YTILE) {
            // ---- C = 64: the shortcut comes in and the output leaves in FULL LINES through LDS chunk tiles (as the C = 256 form does): straight
            //      from the accumulators a store instruction covers 32 pixels x 32 B, and such scattered pieces issue 3x slower per CU — the
            //      block is memory-bound (276 us as three launches against a 107 us HBM floor), so that was its largest term ----
            static_assert(TNW == 1 && CB == 1, "C = 64");
            const int hi8 = lane >> 3;
            unsigned yv[2];              // per-lane byte offset of DMA / store instruction e = 4 wave + j (pixels 8e .. 8e+7 = tile row 2 wave + (j >> 1), columns 8 (j & 1) + hi8): j & 1
#pragma unroll
            for (int o = 0; o < 2; ++o) yv[o] = (unsigned)(hi8 * (4 * C * 2)) + (unsigned)((((lane & 7) ^ ((4 * o + (lane >> 4)) & 7))) << 4);
            const unsigned ysoff = (unsigned)((((size_t)(y0 + 2 * wave) * a.W + x0) * (size_t)(4 * C)) * 2);
            const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(yimg, 0, (int)img_bytes, 0x00020000);
#define BN_YOFF(J, CH) (ysoff + (unsigned)((((J) >> 1) * a.W + ((J) & 1) * 8) * (4 * C * 2) + (CH) * C * 2))
#define BN_RES_DMA(CH)                                                                                         \
    {                                                                                                          \
        const unsigned yb_ = lds0 + (unsigned)(((CH) & 1) ? K::OFF_Y1 : K::OFF_Y0) + (unsigned)(wave * 4 * 1024); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { BN_BLDS(yv[j & 1], srdX, BN_YOFF(j, CH), yb_ + j * 1024) } \
    }
            BN_RES_DMA(0)
            BN_RES_DMA(1)
            for (int ch = 0; ch < ((a.dbg & 4) ? 1 : 4); ++ch, ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.0f;
                BN_VMCNT0                                   // W3's step and this chunk's shortcut tile have landed (this wave's share)
                __syncthreads();                            // ... everybody's; the tile of chunk ch - 1 has been read out
                if (ch + 1 < 4) BN_ISSUE_W(w3off, srdW3, (unsigned)(((ch + 1) * C * C) * 2), (s + 1) & 1)
                if (ch >= 1 && ch + 1 < 4) BN_RES_DMA(ch + 1)
                const unsigned char* const sb = smem + (s & 1) * K::B_STAGE;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x8 af[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(smem + t2row + (i * 32) * 128 + w_c[g]);
                    const f16x8 wf = *reinterpret_cast<const f16x8*>(sb + bwr + w_c[g]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) BN_MFMA(wf, af[i], acc[i][0])
                }
                unsigned char* const yt = smem + ((ch & 1) ? K::OFF_Y1 : K::OFF_Y0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int cl = ch * C + wn * NB + 16 * p + 4 * kk;
                        const float4 sa = *reinterpret_cast<const float4*>(tabC + cl), sb_ = *reinterpret_cast<const float4*>(tabC + cl + 8);
                        const float4 ha = *reinterpret_cast<const float4*>(tabC + 4 * C + cl), hb = *reinterpret_cast<const float4*>(tabC + 4 * C + cl + 8);
                        float4 va = make_float4(acc[i][0][8 * p + 0], acc[i][0][8 * p + 1], acc[i][0][8 * p + 2], acc[i][0][8 * p + 3]);
                        float4 vb = make_float4(acc[i][0][8 * p + 4], acc[i][0][8 * p + 5], acc[i][0][8 * p + 6], acc[i][0][8 * p + 7]);
                        va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                        vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                        const int pl = wm * 64 + i * 32 + l31;           // tile pixel; its 128 B hold the chunk's 64 columns, 16-B piece c at c ^ ((pl >> 1) & 7)
                        uint4* const slot = reinterpret_cast<uint4*>(yt + pl * 128 + (((wn * 4 + 2 * p + kk) ^ ((pl >> 1) & 7)) << 4));
                        float4 ra, rb;
                        bn_unpack16(*slot, ra, rb);
                        va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
                        vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
                        va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                        vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                        range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                        *slot = bn_pack16(va, vb);
                    }
                __syncthreads();                            // the chunk tile is complete
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
                    const bn_u32x4 v_ = *reinterpret_cast<const bn_u32x4*>(yt + (wave * 4 + j) * 1024 + lane * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v_, srdY, (int)yv[j & 1], (int)BN_YOFF(j, ch), 0);
                }
            }
#undef BN_RES_DMA
#undef BN_YOFF
            } else {
            for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
                // the shortcut of the chunk: requested ahead of its K loop (16 B = eight channels per lane, the store layout)
                uint4 rv[2][TNW][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            rv[i][j][p] = *reinterpret_cast<const uint4*>(ximg + grow[i] + ch * C + wn * NB + j * 32 + 16 * p + 8 * kk);
                for (int kb = 0; kb < CB; ++kb, ++s) {
                    BN_VMCNT0
                    __syncthreads();
                    {
                        int kb1 = kb + 1, ch1 = ch;
                        if (kb1 == CB) { kb1 = 0; ++ch1; }
                        if (ch1 < 4) BN_ISSUE_W(w3off, srdW3, (unsigned)((ch1 * C * C + kb1 * 64) * 2), (s + 1) & 1)
                    }
                    const unsigned char* const sb = smem + (s & 1) * K::B_STAGE;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x8 af[2], wf[TNW];
#pragma unroll
                        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(smem + t2row + (kb * P + i * 32) * 128 + w_c[g]);
#pragma unroll
                        for (int j = 0; j < TNW; ++j) wf[j] = *reinterpret_cast<const f16x8*>(sb + bwr + j * 32 * 128 + w_c[g]);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < TNW; ++j) BN_MFMA(wf[j], af[i], acc[i][j])
                    }
                }
                // epilogue of the chunk: straight from the accumulators
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const int cl = ch * C + wn * NB + j * 32 + 16 * p + 4 * kk;
                            const float4 sa = *reinterpret_cast<const float4*>(tabC + cl), sb_ = *reinterpret_cast<const float4*>(tabC + cl + 8);
                            const float4 ha = *reinterpret_cast<const float4*>(tabC + 4 * C + cl), hb = *reinterpret_cast<const float4*>(tabC + 4 * C + cl + 8);
                            float4 va = make_float4(acc[i][j][8 * p + 0], acc[i][j][8 * p + 1], acc[i][j][8 * p + 2], acc[i][j][8 * p + 3]);
                            float4 vb = make_float4(acc[i][j][8 * p + 4], acc[i][j][8 * p + 5], acc[i][j][8 * p + 6], acc[i][j][8 * p + 7]);
                            va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                            vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                            float4 ra, rb;
                            bn_unpack16(rv[i][j][p], ra, rb);
                            va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
                            vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
                            va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                            vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                            range_trip = range_trip || bn_bad(va) || bn_bad(vb);
                            const uint4 pk = bn_pack16(va, vb);
                            *reinterpret_cast<uint4*>(yimg + grow[i] + ch * C + wn * NB + j * 32 + 16 * p + 8 * kk) = pk;
                        }
            }
        }
            }
#undef BN_ISSUE_W
        }
        if constexpr (STAGE) BN_VMCNT0                 // this thread's output stores have completed ...
        __syncthreads();                               // the next tile's phase A restarts the ring at stage 0 (STAGE: ... and everybody's)
        if constexpr (STAGE) {
            if (t == 0) __hip_atomic_store(a.done + tile, (unsigned)(layer + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    }
#undef BN_L
    if (a.range_flag && range_trip) atomicOr(a.range_flag, 1);
}

__global__ void k_bneck_pack_frag(const _Float16* __restrict__ w, int N, int Kt, uint4* __restrict__ out)
{
    // one thread per 16-B fragment piece: granule (nt, kg), lane (l31, kk) <- w[32 nt + l31][16 kg + 8 kk .. + 7]
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KG = Kt / 16;
    if (i >= (long)(N / 32) * KG * 64) return;
    const int lane = (int)(i & 63);
    const long gq = i >> 6;
    const int kg = (int)(gq % KG), nt = (int)(gq / KG);
    out[i] = *reinterpret_cast<const uint4*>(w + (size_t)(nt * 32 + (lane & 31)) * Kt + kg * 16 + (lane >> 5) * 8);
}

void bneck_pack_frag(hipStream_t s, const void* wgt_std, int N, int Kt, DevBuf& out)
{
    MRCNN_REQUIRE(N % 32 == 0 && Kt % 16 == 0, MRCNN_ERR_SHAPE, "bneck_pack_frag: [%d][%d]", N, Kt);
    const long n = (long)(N / 32) * (Kt / 16) * 64;
    out.alloc((size_t)n * 16);
    hipLaunchKernelGGL(k_bneck_pack_frag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const _Float16*>(wgt_std), N, Kt, out.as<uint4>());
    HIP_CHECK(hipGetLastError());
}

bool bneck_frag_wanted(int KH, int KW, int Cin, int Cout)
{
    return (KH == 3 && KW == 3 && Cin == 256 && Cout == 256) || (KH == 1 && KW == 1 && Cin == 256 && Cout == 1024) || (KH == 1 && KW == 1 && Cin == 1024 && Cout == 256);
}

bool bneck_geometry_ok(int C, int H, int W)
{
    if (C == 256) return H % BnCfg<256>::TH == 0 && W % 16 == 0;
    if (C == 128) return H % BnCfg<128>::TH == 0 && W % 16 == 0;
    if (C == 64) return H % BnCfg<64>::TH == 0 && W % 16 == 0;
    return false;
}

void bneck_launch(hipStream_t s, int C, const void* x, void* y, int B, int H, int W, const void* w1, const void* w2, const void* w3,
                  const float* s1, const float* h1, const float* s2, const float* h2, const float* s3, const float* h3, int* range_flag, int n_cus,
                  const void* w2f, const void* w3f, const void* w1f, const void* ws, const float* ss, const float* hs)
{
    const bool first = ws != nullptr;          // the stage-entry form: x has C channels, the shortcut is the convolution ws of x
    MRCNN_REQUIRE(!first || C == 64, MRCNN_ERR_SHAPE, "bneck: the stage-entry form exists for C = 64");
    MRCNN_REQUIRE(bneck_geometry_ok(C, H, W), MRCNN_ERR_SHAPE, "bneck: C %d at %dx%d", C, H, W);
    MRCNN_REQUIRE(x != y, MRCNN_ERR_INVALID, "bneck: the output must not alias the input");
    MRCNN_REQUIRE((size_t)H * W * 4 * C * 2 < 0x80000000ull, MRCNN_ERR_SHAPE, "bneck: image of %dx%dx%d exceeds the 2-GB offset range", H, W, 4 * C);
    BneckArgs a;
    a.x = static_cast<const _Float16*>(x); a.y = static_cast<_Float16*>(y);
    a.w1 = static_cast<const _Float16*>(w1); a.w2 = static_cast<const _Float16*>(w2); a.w3 = static_cast<const _Float16*>(w3);
    a.w1f = static_cast<const uint4*>(w1f); a.w2f = static_cast<const uint4*>(w2f); a.w3f = static_cast<const uint4*>(w3f);
    a.ws = static_cast<const _Float16*>(ws); a.ss = ss; a.hs = hs;
    a.s1 = s1; a.h1 = h1; a.s2 = s2; a.h2 = h2; a.s3 = s3; a.h3 = h3;
    a.B = B; a.H = H; a.W = W;
    const int th = C == 256 ? 8 : 16;
    a.tiles_x = W / 16; a.tiles_y = H / th; a.ntiles = B * a.tiles_x * a.tiles_y;
    a.range_flag = range_flag;
    // measurement only (tools/bneck_phases.sh; needs MRCNN_TEST_KNOBS=1): 1 / 2 / 4 = phase A / B / C cut to one step — RESULTS INVALID, said once on stderr
    static const int dbg = [] {
        const char* e = knob_env("MRCNN_BNECK_DBG");
        const int v = e ? atoi(e) : 0;
        if (v) fprintf(stderr, "libmaskrcnn_hip: MRCNN_BNECK_DBG=%d — the fused bottleneck blocks skip work: results are NOT valid (measurement only)\n", v);
        return v;
    }();
    a.dbg = dbg;
    int grid = n_cus > 0 ? n_cus / 8 * 8 : 256;
    if (grid <= 0) grid = 8;
    if (a.ntiles < grid) grid = a.ntiles;
    a.layers = nullptr; a.nlayers = 1; a.pp[0] = a.pp[1] = nullptr; a.done = nullptr;
    if (C == 256 && w1f && w2f && w3f) hipLaunchKernelGGL((k_bneck_h<256, true>), dim3(grid), dim3(512), 0, s, a);
    else if (C == 256) hipLaunchKernelGGL((k_bneck_h<256, false>), dim3(grid), dim3(512), 0, s, a);
    else if (C == 128) hipLaunchKernelGGL((k_bneck_h<128, false>), dim3(grid), dim3(512), 0, s, a);
    else if (first) hipLaunchKernelGGL((k_bneck_h<64, false, true>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((k_bneck_h<64, false>), dim3(grid), dim3(512), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

// The identity blocks of a C = 256 stage as ONE launch (STAGE form).  layers: nlayers BneckLayer records on the device; pp0 holds the stage's
// input and, with pp1, the two tensors the blocks ping-pong between (the result is in pp[nlayers & 1]); done: one counter per tile, zeroed here
// on the stream (a memset node: the launch stays graph-capturable).  Needs the whole grid resident: at most one block per CU.
void bneck_stage_launch(hipStream_t s, const void* layers_dev, int nlayers, void* pp0, void* pp1, int B, int H, int W, unsigned* done, int* range_flag, int n_cus)
{
    MRCNN_REQUIRE(bneck_geometry_ok(256, H, W), MRCNN_ERR_SHAPE, "bneck stage: C 256 at %dx%d", H, W);
    MRCNN_REQUIRE(layers_dev && nlayers >= 1 && pp0 && pp1 && pp0 != pp1 && done && range_flag, MRCNN_ERR_INVALID, "bneck stage: bad argument");
    MRCNN_REQUIRE((size_t)H * W * 1024 * 2 < 0x80000000ull, MRCNN_ERR_SHAPE, "bneck stage: image of %dx%dx1024 exceeds the 2-GB offset range", H, W);
    BneckArgs a{};
    a.B = B; a.H = H; a.W = W;
    a.tiles_x = W / 16; a.tiles_y = H / 8; a.ntiles = B * a.tiles_x * a.tiles_y;
    a.range_flag = range_flag;
    a.dbg = 0;
    a.layers = static_cast<const BneckLayer*>(layers_dev); a.nlayers = nlayers;
    a.pp[0] = static_cast<_Float16*>(pp0); a.pp[1] = static_cast<_Float16*>(pp1);
    a.done = done;
    int grid = n_cus > 0 ? n_cus / 8 * 8 : 256;
    if (grid <= 0) grid = 8;
    if (a.ntiles < grid) grid = a.ntiles;
    HIP_CHECK(hipMemsetAsync(done, 0, (size_t)a.ntiles * sizeof(unsigned), s));
    hipLaunchKernelGGL((k_bneck_h<256, true, false, true>), dim3(grid), dim3(512), 0, s, a);
    HIP_CHECK(hipGetLastError());
}
size_t bneck_layer_record_bytes() { return sizeof(BneckLayer); }
void bneck_layer_record(void* dst, const void* w1f, const void* w2f, const void* w3f, const float* s1, const float* h1, const float* s2, const float* h2,
                        const float* s3, const float* h3)
{
    BneckLayer L;
    L.w1f = static_cast<const uint4*>(w1f); L.w2f = static_cast<const uint4*>(w2f); L.w3f = static_cast<const uint4*>(w3f);
    L.s1 = s1; L.h1 = h1; L.s2 = s2; L.h2 = h2; L.s3 = s3; L.h3 = h3;
    memcpy(dst, &L, sizeof(L));
}

}  // namespace mrcnn
