// kernels_conv.hip — the convolution family of the trunk and heads on gfx950 matrix cores, plus
// the small element-wise kernels around it.
//
// Replaces the built-in Core ML layers of MaskRCNN.mlmodel / Classifier.mlmodel / Mask.mlmodel
// (spec emitted by Sources/maskrcnn/Python/Conversion/task.py:69-116 of the reference): 7×7/3×3/1×1
// convolutions with folded BatchNorm, ReLU, residual adds, FPN nearest-neighbour upsample+add, the
// RPN / box / mask heads (inner products as 1×1 convolutions, the 2×2 stride-2 transposed
// convolution as a scatter GEMM), soft-max and sigmoid.
//
// Kernel design (fp32: exact f32 MFMA `v_mfma_f32_32x32x2_f32`; fp16: `v_mfma_f32_32x32x16_f16`):
//   implicit GEMM  out[m][n] = Σ_k A[m][k]·Wt[n][k],  m = (image, oh, ow), k = (tap, cin), n = cout
//   * activations NHWC and filters packed [cout][tap][cin], so BOTH operands are "rows of K":
//     every global load is a 16-B piece of a 128-B contiguous run (32 fp32 / 64 fp16 channels of one tap);
//   * 128×BN block tile, K tile of 128 B, 8 waves (wave64) as 4×2, each wave a (TM×32)×(TN×32) sub-tile
//     of 32×32 MFMA accumulators; K is permuted inside the tile (lane kk∈{0,1} owns one 16-B chunk of
//     every 32-B pair) so a lane fetches its operands for four fp32 MFMA steps / one fp16 MFMA with ONE
//     ds_read_b128 — identical permutation on both operands, so the sum is unchanged;
//   * operands go global→LDS by DMA into two XOR-swizzled buffers (details at the kernel below);
//   * fused epilogue: per-channel scale/shift (folded BN + bias), residual (optionally read at
//     (oh>>1, ow>>1): FPN top-down upsample+add), ReLU / sigmoid, optional column split into two
//     outputs (RPN class + bbox from one GEMM) or 2×2 scatter (transposed conv);
//   * XCD-aware block→tile map: the 8 XCDs get contiguous runs of tiles, N-tiles of one M-tile
//     adjacent, so an A tile is fetched into one XCD's L2 once.
#include <map>
#include <utility>
#include <mutex>

#include "conv_device.h"

namespace mrcnn {

// ------------------------------------------------------------------------------------------------
// The implicit-GEMM kernel.  T = float   : v_mfma_f32_32x32x2_f32  (exact fp32, 157.3 TFLOP/s peak), K tile = 32
//                            T = _Float16: v_mfma_f32_32x32x16_f16 (fp32 accumulate, ~2.5 PFLOP/s peak), K tile = 64
// Operands are staged global→LDS directly (global_load_lds_dwordx4): no VGPR round trip, no ds_write,
// no address/register traffic between the loads and the MFMAs — a register-staged loop (round-1
// history) lost 13 % (fp32) / 40 % (fp16) of the MFMA rate to that path even with
// cache-hot loads.
//   * LDS rows are unpadded 128-B K runs (the DMA writes wave-uniform base + lane×16, so rows cannot
//     be padded); bank conflicts are removed by an XOR swizzle instead: 16-B chunk c of row r lives
//     at chunk position c ^ ((r >> 1) & 7).  The permutation is applied to the per-lane SOURCE
//     address (inside one 128-B line, so coalescing is unchanged) and to the ds_read address.
//     For ds_read_b128's 16-lane groups the pairs (r & 1, (r >> 1) & 7) are all distinct → conflict-free.
//   * the DMAs go through buffer resources (buffer_load_dwordx4 … lds): out-of-image taps (zero padding) and rows beyond
//     M carry an out-of-range offset and the hardware deposits zeros — not predicated, no memory access.
//   * two LDS buffers: the DMA of tile k+1 is issued right after the barrier that retired buffer
//     (k+1)&1 and lands while tile k is being multiplied; one vmcnt(0) + barrier per K step.
// ------------------------------------------------------------------------------------------------
template <typename T, typename TW, int BN, int TM, int TN, int WM, int WN, int STAGES, int PARTS = 2>
__global__ __launch_bounds__(WM * WN * 64, WM * WN >= 8 ? 4 : 2) void k_conv_mfma_glds(const ConvArgs a)     // two blocks per CU
{
    // SPLIT: fp32 activations × fp16 filters as PARTS (2 or 3) fp16 MFMA passes over a split of the activations
    constexpr bool SPLIT = sizeof(T) == 4 && sizeof(TW) == 2;
    static_assert(PARTS == 2 || PARTS == 3, "split parts");
    static_assert(sizeof(T) == sizeof(TW) || SPLIT, "operand types");
    constexpr int BM = WM * TM * 32;
    static_assert(WN * TN * 32 == BN, "tile shape");
    constexpr int EPV = Elem<T>::EPV;
    constexpr int BK = 8 * EPV;
    constexpr int NT = WM * WN * 64;
    constexpr int RPT = NT / 8;
    constexpr int AP = BM / RPT;
    constexpr int BP = SPLIT ? (BN * 4 > NT ? BN * 4 / NT : 1) : BN / RPT;   // SPLIT: 64-B filter rows, 16 B per thread and DMA
    static_assert(AP >= 1 && BP >= 1 && AP <= 4 && BP <= 4 && RPT % 16 == 0, "staging shape");
    static_assert(!SPLIT || BP <= 2, "split mode: the filter tile is staged by at most two DMAs per thread");
    constexpr int ROWB = 128;
    constexpr int BROWB = SPLIT ? 64 : 128;                     // bytes of one filter row per K step
    constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * BROWB;
    constexpr int CPASS = BM * BN * 4 > 64 * 1024 ? WN : 1;    // column passes of the epilogue (fp32 C tile of at most 64 KB)
    constexpr int C_BYTES = BM * (BN / CPASS + 4) * 4;         // rows padded by 16 B (conv_epilogue)
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int NLOADS = SPLIT ? AP : AP + BP;                // DMA instructions per tile per thread (SPLIT: waves past BN/16 issue no filter DMA)
    constexpr int SMEM_OPS = STAGES * (A_STAGE + B_STAGE);      // ring of operand buffers
    // fp16 tensors, 32 x 64 wave tiles: the wave-private epilogue stages 32 x 68 floats per wave (conv_epilogue_wave_h)
    constexpr bool WAVE_H = sizeof(T) == 2 && TN == 2 && BN == 128 && CPASS == 1;
    constexpr int WAVE_H_BYTES = WAVE_H ? WM * WN * 32 * 68 * 4 : 0;
    constexpr int SMEM_ = SMEM_OPS > C_BYTES ? SMEM_OPS : C_BYTES;
    constexpr int SMEM = SMEM_ > WAVE_H_BYTES ? SMEM_ : WAVE_H_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    constexpr bool DIRECT_OK = CPASS == 1 && BN >= 64;                 // (the 32-wide tiles fill the LDS of two blocks to the byte)
    __shared__ __attribute__((aligned(16))) float s_tab[DIRECT_OK ? 2 * BN : 4];       // scale | shift of the block's columns (direct epilogue)
    const T* const in = static_cast<const T*>(a.in);
    const TW* const wgt = static_cast<const TW*>(a.wgt);
#ifdef MRCNN_CONV_ABLATE
    // experiment: de-phase the two blocks of a CU once, at the start of the launch (blocks 256..511 take the second slots)
    if ((a.dbg >> 16) && blockIdx.x >= 256 && blockIdx.x < 512) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)(a.dbg >> 16)) __builtin_amdgcn_s_sleep(16);      // units of 10 ns
    }
#endif

    // KCH: the kernel carries the canonical K chunks of long-K 1x1 layers (ConvArgs::kchunks; conv_k_chunks below).  The 8-wave
    // 128-column instantiation has no room for the second accumulator set (112 of the 128 VGPRs that four waves per SIMD
    // allow): chunked layers take the 4-wave 128-column form instead (conv_forward; same speed on them, gpurun_out/r4v)
    constexpr bool KCH = SPLIT && !(BN == 128 && WM * WN >= 8);
    const int ksplit = KCH ? a.ksplit : 1;
    const int nblocks = a.tiles_m * a.tiles_n * ksplit;
    const int bid = blockIdx.x;
    const int q = nblocks >> 3, r8 = nblocks & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int unit = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + local;
    const int tile = unit / ksplit, chunk = unit - tile * ksplit;          // the blocks of a tile are neighbours in the walk (mostly one XCD)
    const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int r0 = t >> 3;                              // row inside a staging pass
    const int kq = (t & 7) ^ ((r0 >> 1) & 7);           // SOURCE chunk held at LDS chunk position (t & 7)

    // Both operands are addressed through buffer resources (raw, stride 0): 32-bit byte offsets per lane, and a lane
    // whose offset lies beyond the resource deposits ZEROS in LDS — zero padding and rows beyond M cost no memory access
    // and no zero page.  (Also the fastest form of the DMA on this chip: +5-8 % over 64-bit lane addresses in the kernel,
    // tools/probes/dma_probe.hip.)  The activation resource starts at the first image the block touches, so that offsets
    // stay small whatever the batch, and ends with the tensor.
    const int ohw = a.OH * a.OW;
    const int b0 = m0 / ohw;
    constexpr unsigned OOB = 0xffffff00u;
    typedef unsigned srd_t __attribute__((ext_vector_type(4)));
    srd_t srdA, srdB;
    {
        const unsigned long long ia = (unsigned long long)(uintptr_t)(in + (long)b0 * a.in_sB), wa = (unsigned long long)(uintptr_t)wgt;
        const unsigned long long rest = (unsigned long long)(a.M / ohw - b0) * (unsigned long long)a.in_sB * sizeof(T);
        srdA[0] = __builtin_amdgcn_readfirstlane((unsigned)ia);
        srdA[1] = __builtin_amdgcn_readfirstlane((unsigned)(ia >> 32) & 0xffffu);
        srdA[2] = __builtin_amdgcn_readfirstlane((unsigned)(rest < OOB ? rest : OOB));
        srdA[3] = 0x00020000u;
        srdB[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
        srdB[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
        srdB[2] = 0xffffffffu;
        srdB[3] = 0x00020000u;
    }
    // SC (fused shortcut, round 4): behind the layer's own K steps the SAME loop runs the K steps of the block's shortcut convolution — a
    // second 1x1 over the block's input (other tensor, strides, filters, K).  The DMA stream switches source when the layer's channels are
    // exhausted (the ring never drains: no second pipeline fill), the accumulators change hands at that step (`tot` keeps the layer's sums,
    // `acc` restarts for the shortcut's), and the epilogue forms the residual y_sc = acc * scale2 + shift2 where it would have loaded it.
    const bool sc_fused = KCH && DIRECT_OK && a.sc_in != nullptr;
    unsigned a_ob[AP];          // byte offset of tap (0, 0) of the staged row from the resource base (mod 2^32: it may lie before it)
    unsigned a_ob2[KCH ? AP : 1];                         // ... of the shortcut's input row
    int ih0[AP], iw0[AP];
    bool a_ok[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int m = m0 + r0 + RPT * p;
        a_ok[p] = m < a.M;
        const int mm = a_ok[p] ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oh = rem / a.OW, ow = rem - oh * a.OW;
        ih0[p] = oh * a.stride - a.padH;
        iw0[p] = ow * a.stride - a.padW;
        a_ob[p] = (unsigned)(((long)(b - b0) * a.in_sB + (long)ih0[p] * a.in_sH + (long)iw0[p] * a.in_sW + kq * EPV) * (long)sizeof(T));
        if constexpr (KCH)
            a_ob2[p] = (unsigned)(((long)(b - b0) * a.sc_in_sB + (long)(oh * a.sc_stride) * a.sc_in_sH + (long)(ow * a.sc_stride) * a.sc_in_sW + kq * EPV) * (long)sizeof(T));
    }
    srd_t srdA2 = srdA, srdB2 = srdB;
    if constexpr (KCH) {
        if (sc_fused) {
            const unsigned long long ia = (unsigned long long)(uintptr_t)(static_cast<const T*>(a.sc_in) + (long)b0 * a.sc_in_sB), wa = (unsigned long long)(uintptr_t)a.sc_wgt;
            const unsigned long long rest = (unsigned long long)(a.M / ohw - b0) * (unsigned long long)a.sc_in_sB * sizeof(T);
            srdA2[0] = __builtin_amdgcn_readfirstlane((unsigned)ia);
            srdA2[1] = __builtin_amdgcn_readfirstlane((unsigned)(ia >> 32) & 0xffffu);
            srdA2[2] = __builtin_amdgcn_readfirstlane((unsigned)(rest < OOB ? rest : OOB));
            srdB2[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
            srdB2[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
        }
    }
    int cin_tiles = a.Cin / BK;
    const int KT_main = a.KH * a.KW * cin_tiles;
    const int KT_all = KT_main + (sc_fused ? a.sc_Cin / BK : 0);
    const int KT = KT_all / ksplit;                       // K steps of THIS block: one chunk when the tile is shared
    const int kbeg = chunk * KT;
    // wave-uniform LDS destinations: this wave's 8 rows of each staging pass
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);   // LDS byte address of smem

    // ---- DMA address generation, kept off the per-tile critical path ------------------------------
    // A (activations): one 32-bit byte offset per staged row, advanced by a per-row step each tile
    //   (step = one K tile, or 0 when the tap falls outside the image and the row's offset is out of range → zeros);
    //   the bounds test and the select run only when the TAP changes (every Cin/BK tiles).
    // B (filters): scalar offset (advanced by one K tile) + loop-invariant 32-bit per-lane byte offset.
    // The DMA itself goes through inline asm on purpose: with the builtin hipcc treats it as an LDS
    // store it must order against every later ds_read and drains it with vmcnt(0) at the top of the
    // step, which makes the copy synchronous.  In asm the compiler does not count it, so the waits
    // are placed by hand (M0 = wave-uniform LDS byte address; lane i's 16 B land at M0 + 16 i).
    static_assert(AP <= 4 && BP <= 4, "per-row DMA state is spelled out for <= 4 A rows / <= 4 B rows");
    unsigned oa0 = OOB, oa1 = OOB, oa2 = OOB, oa3 = OOB;
    unsigned sa0 = 0, sa1 = 0, sa2 = 0, sa3 = 0;
    // SPLIT: thread t stages 16 B (8 fp16 channels) of filter row t>>2; chunk c of row r sits at position c ^ ((r>>2)&3)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool wave_has_b = !SPLIT || BN * 4 >= NT || wave_u < BN / 16;
    unsigned vb0 = SPLIT ? (unsigned)(((size_t)(t >> 2) * a.Ktot + (((t & 3) ^ ((t >> 4) & 3)) << 3)) * sizeof(TW))
                               : (unsigned)(((size_t)r0 * a.Ktot + kq * EPV) * sizeof(T));
    unsigned vb1 = SPLIT ? vb0 + (unsigned)((size_t)(NT / 4) * a.Ktot * sizeof(TW))
                               : (unsigned)(((size_t)(r0 + RPT) * a.Ktot + kq * EPV) * sizeof(T));
    const unsigned vb2 = (unsigned)(((size_t)(r0 + 2 * RPT) * a.Ktot + kq * EPV) * sizeof(T));
    const unsigned vb3 = (unsigned)(((size_t)(r0 + 3 * RPT) * a.Ktot + kq * EPV) * sizeof(T));
    unsigned sob = (unsigned)((size_t)n0 * a.Ktot * sizeof(TW));      // uniform: byte offset of the K tile in the filter
    int kh = 0, kw = 0, ct = 0;
#define MRCNN_SET_TAP(P)                                                                                       \
    if constexpr (AP > P) {                                                                                    \
        const int ih = ih0[P] + kh, iw = iw0[P] + kw;                                                          \
        const bool ok = a_ok[P] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;               \
        oa##P = ok ? a_ob[P] + (unsigned)(((long)kh * a.in_sH + (long)kw * a.in_sW) * (long)sizeof(T)) : OOB;  \
        sa##P = ok ? BK * (unsigned)sizeof(T) : 0u;                                                            \
    }
    bool sc_pending = sc_fused;
    /* the layer's channels are exhausted: the stream continues with the shortcut's input and filters (1x1, no padding: every staged row in range) */
#define MRCNN_SC_ROW(P) if constexpr (AP > P) { oa##P = a_ok[P] ? a_ob2[P] : OOB; sa##P = a_ok[P] ? BK * (unsigned)sizeof(T) : 0u; }
#define MRCNN_SC_SWITCH()                                                                                      \
    {                                                                                                          \
        if constexpr (KCH) {                                                                                   \
            sc_pending = false;                                                                                \
            MRCNN_SC_ROW(0) MRCNN_SC_ROW(1) MRCNN_SC_ROW(2) MRCNN_SC_ROW(3)                                    \
            srdA = srdA2; srdB = srdB2;                                                                        \
            cin_tiles = a.sc_Cin / BK;                                                                         \
            vb0 = (unsigned)(((size_t)(t >> 2) * a.sc_Cin + (((t & 3) ^ ((t >> 4) & 3)) << 3)) * sizeof(TW));   \
            vb1 = vb0 + (unsigned)((size_t)(NT / 4) * a.sc_Cin * sizeof(TW));                                  \
            sob = (unsigned)((size_t)n0 * a.sc_Cin * sizeof(TW));                                              \
        }                                                                                                      \
    }
#define MRCNN_GLDS_V(VOFF, DST)                                                                                \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(VOFF), "s"(srdA), "s"(DST) : "memory", "m0");
#define MRCNN_GLDS_S(VOFF, SOFF, DST)                                                                          \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(VOFF), "s"(srdB), "s"(SOFF), "s"(DST) : "memory", "m0");
#define MRCNN_DMA_A(P) if constexpr (AP > P) { MRCNN_GLDS_V(oa##P, da + P * RPT * ROWB); oa##P += sa##P; }
#define MRCNN_DMA_TILE(KT_, BUF_)                                                                              \
    {                                                                                                          \
        const unsigned da = lds0 + (BUF_) * A_STAGE + wrow * ROWB;                                             \
        const unsigned db = lds0 + STAGES * A_STAGE + (BUF_) * B_STAGE + (SPLIT ? wave_u * 1024 : wrow * ROWB); \
        MRCNN_DMA_A(0) MRCNN_DMA_A(1) MRCNN_DMA_A(2) MRCNN_DMA_A(3)                                            \
        if (wave_has_b) MRCNN_GLDS_S(vb0, sob, db);                                                            \
        if constexpr (BP > 1) MRCNN_GLDS_S(vb1, sob, db + (SPLIT ? NT * 16 : RPT * ROWB));                     \
        if constexpr (BP > 2) MRCNN_GLDS_S(vb2, sob, db + 2 * RPT * ROWB);                                     \
        if constexpr (BP > 3) MRCNN_GLDS_S(vb3, sob, db + 3 * RPT * ROWB);                                     \
        sob += BK * (unsigned)sizeof(TW);                                                                      \
        if (++ct == cin_tiles) {                                                                               \
            ct = 0;                                                                                            \
            if (KCH && sc_pending) { MRCNN_SC_SWITCH() }                                                       \
            else {                                                                                             \
                if (++kw == a.KW) { kw = 0; ++kh; }                                                            \
                MRCNN_SET_TAP(0) MRCNN_SET_TAP(1) MRCNN_SET_TAP(2) MRCNN_SET_TAP(3)                            \
            }                                                                                                  \
        }                                                                                                      \
    }
    MRCNN_SET_TAP(0) MRCNN_SET_TAP(1) MRCNN_SET_TAP(2) MRCNN_SET_TAP(3)
    if constexpr (KCH) {
        if (kbeg) {                                       // (1x1 layers only: a K step is a channel step)
            oa0 += kbeg * sa0; oa1 += kbeg * sa1; oa2 += kbeg * sa2; oa3 += kbeg * sa3;
            sob += (unsigned)kbeg * BK * (unsigned)sizeof(TW);
        }
    }

    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, kk = lane >> 5;
    const int swz = (l31 >> 1) & 7;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // canonical K chunks, one block per tile: `tot` folds the finished chunks, `acc` restarts from zero at every boundary
    f32x16 tot[KCH ? TM : 1][KCH ? TN : 1];
    const int klen = KCH ? KT_all / a.kchunks : 0;
    int kb = (KCH && a.kchunks > 1 && ksplit == 1) ? klen : (sc_fused ? KT_main : 0x7fffffff);        // (fused shortcut: the accumulators change hands behind the layer's own steps)
    if constexpr (KCH) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) tot[i][j][e] = 0.0f;
    }
#define MRCNN_KFOLD                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                         \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) { tot[i][j][e] += acc[i][j][e]; acc[i][j][e] = 0.0f; }

    // Ring of STAGES operand buffers, STAGES-1 tiles in flight: tile k+STAGES-1 is issued at the top of
    // step k, and only tile k+1 has to have landed at the end of it — counted vmcnt lets the
    // NLOADS·(STAGES-2) most recent DMAs stay outstanding across the barrier.  STAGES = 2 for the
    // 128-wide tile (a step is 2048 MFMA cycles per wave, longer than the DMA latency); the narrow
    // tiles run on under-filled grids with short steps and use deeper rings.
    // direct epilogue (conv_device.h): its scale / shift table goes to LDS before the first barrier
    const bool direct = DIRECT_OK && a.direct;
    if (direct && t < BN / 2) {
        const int c = (t < BN / 4 ? t : t - BN / 4) * 4;
        const float* src = t < BN / 4 ? a.scale : a.shift;
        const float fill = t < BN / 4 ? 1.0f : 0.0f;
        *reinterpret_cast<float4*>(&s_tab[(t < BN / 4 ? 0 : BN) + c]) =
            src ? *reinterpret_cast<const float4*>(src + n0 + c) : make_float4(fill, fill, fill, fill);
    }
    __shared__ __attribute__((aligned(16))) float s_tab2[KCH && DIRECT_OK ? 2 * BN : 4];       // scale | shift of the shortcut convolution's columns
    if constexpr (KCH && DIRECT_OK) {
        if (sc_fused && t < BN / 2) {
            const int c = (t < BN / 4 ? t : t - BN / 4) * 4;
            const float* src = t < BN / 4 ? a.sc_scale : a.sc_shift;
            const float fill = t < BN / 4 ? 1.0f : 0.0f;
            *reinterpret_cast<float4*>(&s_tab2[(t < BN / 4 ? 0 : BN) + c]) =
                src ? *reinterpret_cast<const float4*>(src + n0 + c) : make_float4(fill, fill, fill, fill);
        }
    }
    MRCNN_DMA_TILE(0, 0)
    if (STAGES > 2 && KT > 1) MRCNN_DMA_TILE(1, 1)
    if (STAGES > 3 && KT > 2) MRCNN_DMA_TILE(2, 2)
    if (KT - 1 >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS * (STAGES - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // tile 0 is in LDS

    // One K step on buffer BUF; the DMA of tile KTV + STAGES - 1 goes to buffer NBUF (the one retired
    // by the previous step's barrier).  BUF / NBUF are compile-time constants (loop unrolled by
    // STAGES), so every LDS offset folds into an instruction immediate and the four swizzled lane
    // addresses are loop-invariant.
    const unsigned char* const la = smem + (wm * TM * 32 + l31) * ROWB;
    const unsigned char* const lb = smem + STAGES * A_STAGE + (wn * TN * 32 + l31) * BROWB;
    const int co0 = ((0 + kk) ^ swz) << 4, co1 = ((2 + kk) ^ swz) << 4, co2 = ((4 + kk) ^ swz) << 4, co3 = ((6 + kk) ^ swz) << 4;
    // Operand fetch for the four 32-B K groups of a step is issued up front, ahead of the first MFMA:
    // LDS returns in order, so the waits count down (lgkmcnt) and the reads of group g+1.. are in
    // flight under the MFMAs of group g — needed when a SIMD holds a single wave (narrow tiles on
    // under-filled grids), free otherwise.
    // SPLIT: a step is 32 channels = two fp16 MFMA K groups; lane kk of group g owns channels
    // [8(2g+kk), +8): two fp32 chunks of the activation row, one fp16 chunk of the filter row.
    const int swzb = (l31 >> 2) & 3;
    const int cb0 = ((0 + kk) ^ swzb) << 4, cb1 = ((2 + kk) ^ swzb) << 4;
    const int ca0 = ((0 + 2 * kk) ^ swz) << 4, ca1 = ((1 + 2 * kk) ^ swz) << 4, ca2 = ((4 + 2 * kk) ^ swz) << 4, ca3 = ((5 + 2 * kk) ^ swz) << 4;
#define MRCNN_KLOAD(G, BUF, CO)                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) av[G][i] = *reinterpret_cast<const uint4*>(la + (BUF) * A_STAGE + i * 32 * ROWB + (CO)); \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[G][j] = *reinterpret_cast<const uint4*>(lb + (BUF) * B_STAGE + j * 32 * ROWB + (CO));
#define MRCNN_KLOAD_SPLIT(G, BUF, CA, CA1, CB)                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                           \
        av[2 * G][i] = *reinterpret_cast<const uint4*>(la + (BUF) * A_STAGE + i * 32 * ROWB + (CA));           \
        av[2 * G + 1][i] = *reinterpret_cast<const uint4*>(la + (BUF) * A_STAGE + i * 32 * ROWB + (CA1));      \
    }                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[G][j] = *reinterpret_cast<const uint4*>(lb + (BUF) * B_STAGE + j * 32 * BROWB + (CB));
#define MRCNN_KMATH(G)                                                                                         \
    {                                                                                                          \
        if constexpr (sizeof(T) == 4) {                                                                        \
            _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                      \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
                        const uint32_t au = c == 0 ? av[G][i].x : c == 1 ? av[G][i].y : c == 2 ? av[G][i].z : av[G][i].w;  \
                        const uint32_t bu = c == 0 ? bv[G][j].x : c == 1 ? bv[G][j].y : c == 2 ? bv[G][j].z : bv[G][j].w;  \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(bu), __uint_as_float(au), acc[i][j], 0, 0, 0); \
                    }                                                                                          \
        } else {                                                                                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[G][j]),    \
                                                                       __builtin_bit_cast(f16x8, av[G][i]), acc[i][j], 0, 0, 0); \
        }                                                                                                      \
    }
#define MRCNN_KMATH_SPLIT(G)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                           \
        f16x8 hi, lo, lo2;                                                                                     \
        if constexpr (PARTS == 3) split_hi_mid_lo(av[2 * G][i], av[2 * G + 1][i], hi, lo, lo2);                \
        else split_hi_lo(av[2 * G][i], av[2 * G + 1][i], hi, lo);                                              \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                         \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[G][j]), hi, acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                         \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[G][j]), lo, acc[i][j], 0, 0, 0); \
        if constexpr (PARTS == 3) {                                                                            \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[G][j]), lo2, acc[i][j], 0, 0, 0); \
        }                                                                                                      \
    }
#ifdef MRCNN_CONV_ABLATE      /* measurement build (make ablate): a.dbg 256 no MFMAs (split modes), 512 no DMA in the main loop, 1024 no epilogue, 2048 no residual, 4096 no stores */
#define MRCNN_ABL_NODMA && !(a.dbg & 512)
#define MRCNN_ABL_IFMMA if (!(a.dbg & 256))
#else
#define MRCNN_ABL_NODMA
#define MRCNN_ABL_IFMMA
#endif
#define MRCNN_STEP(BUF, NBUF, KTV)                                                                             \
    {                                                                                                          \
        const bool more = (KTV) + STAGES - 1 < KT MRCNN_ABL_NODMA;                                             \
        if (more) MRCNN_DMA_TILE((KTV) + STAGES - 1, NBUF)                                                     \
        if constexpr (KCH) { if ((KTV) == kb) { MRCNN_KFOLD kb += klen; } }                                    \
        uint4 av[4][TM], bv[4][TN];                                                                            \
        if constexpr (SPLIT) {                                                                                 \
            MRCNN_KLOAD_SPLIT(0, BUF, ca0, ca1, cb0) MRCNN_KLOAD_SPLIT(1, BUF, ca2, ca3, cb1)                  \
            if constexpr (BN < 128) __builtin_amdgcn_sched_barrier(0);                                         \
            MRCNN_ABL_IFMMA { MRCNN_KMATH_SPLIT(0) MRCNN_KMATH_SPLIT(1) }                                      \
        } else {                                                                                               \
            MRCNN_KLOAD(0, BUF, co0) MRCNN_KLOAD(1, BUF, co1) MRCNN_KLOAD(2, BUF, co2) MRCNN_KLOAD(3, BUF, co3) \
            if constexpr (BN < 128) __builtin_amdgcn_sched_barrier(0); /* keep the reads ahead of the MFMAs */ \
            MRCNN_KMATH(0) MRCNN_KMATH(1) MRCNN_KMATH(2) MRCNN_KMATH(3)                                        \
        }                                                                                                      \
        /* tile KTV+1 must have landed before the barrier hands its buffer over */                             \
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS * (STAGES - 2)) : "memory");                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
        __syncthreads(); /* ... and every wave is done reading BUF */                                          \
    }
    for (int kt = 0; kt < KT; kt += STAGES) {
        MRCNN_STEP(0, STAGES - 1, kt)
        if (kt + 1 < KT) MRCNN_STEP(1, 0, kt + 1)
        if constexpr (STAGES > 2) { if (kt + 2 < KT) MRCNN_STEP(2, 1, kt + 2) }
        if constexpr (STAGES > 3) { if (kt + 3 < KT) MRCNN_STEP(3, 2, kt + 3) }
    }
    if constexpr (KCH) {
        if (ksplit > 1) {
            // this block's chunk goes to the scratch (16-B piece q of thread t at [tile][chunk][q][t]: full lines); the block that
            // arrives last — whichever it is — folds the chunks in THE canonical order and runs the epilogue.  The blocks of a tile
            // may sit on different XCDs (one L2 each): the partial sums are stored and loaded at DEVICE scope (sc1: written through /
            // fetched past the non-coherent L2 lines), the stores have completed (vmcnt) before the block counts itself in, and the
            // counter is a device-scope atomic — no cache-wide write-back or invalidate, which is what a __threadfence() costs here
            // (measured: 18.6 -> 86 us on C4's branch2a at batch 1, gpurun_out/r4w).  The counter is left at zero for the next launch.
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            constexpr int NP = TM * TN * 4;                        // 16-B pieces of a thread's accumulators
            constexpr int SC1 = 16;                                // cache policy of the buffer builtins: device scope
            float* const tbase = a.ks_scratch + (size_t)tile * ksplit * NP * NT * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tbase, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        const f32x4 v = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ((chunk * NP + (i * TN + j) * 4 + qd) * NT + t) * 16, 0, SC1);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __shared__ unsigned s_arrived;
            __syncthreads();                                       // every thread's stores have completed
            if (t == 0) s_arrived = __hip_atomic_fetch_add(a.ks_count + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (s_arrived != (unsigned)(ksplit - 1)) return;
            if (t == 0) __hip_atomic_store(a.ks_count + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
            u32x4 cur[NP], nxt[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) cur[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * NT + t) * 16, 0, SC1);
            for (int c = 0; c < ksplit; ++c) {
                if (c + 1 < ksplit) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) nxt[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (((c + 1) * NP + u) * NT + t) * 16, 0, SC1);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * qd + e] += __uint_as_float(cur[(i * TN + j) * 4 + qd][e]);
#pragma unroll
                for (int u = 0; u < NP; ++u) cur[u] = nxt[u];
            }
        } else if (a.kchunks > 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = tot[i][j][e] + acc[i][j][e];
        }
    }
#undef MRCNN_KFOLD
#undef MRCNN_STEP
#undef MRCNN_KMATH_SPLIT
#undef MRCNN_KMATH
#undef MRCNN_KLOAD_SPLIT
#undef MRCNN_KLOAD
#undef MRCNN_DMA_TILE
#undef MRCNN_DMA_A
#undef MRCNN_SC_SWITCH
#undef MRCNN_SC_ROW
#undef MRCNN_GLDS_V
#undef MRCNN_GLDS_S
#undef MRCNN_SET_TAP
#ifdef MRCNN_CONV_ABLATE
    if (a.dbg & 1024) { if (a.dbg == 12345678) static_cast<float*>(a.out)[t] = acc[0][0][0]; return; }
#endif
    if constexpr (DIRECT_OK) {
        if (direct && a.sel_partial) {           // selected-class mode, wave-private form (conv_forward sets direct for it only with 64-column parts)
            if constexpr (BN == 128 && TM == 1 && TN == 2) conv_epilogue_sel_wave<T, BN, TM, TN>(a, acc, s_tab, m0 + wm * TM * 32, n0, wn * TN * 32, lane);
            return;
        }
        if (direct) {
            if constexpr (sizeof(T) == 4) {
                // fp32 tensors (a.direct == 2): through a wave-private 32 x 36-float LDS tile (the operand ring is free: the K loop
                // ended in a barrier) — full-line stores without the two block barriers of the staged epilogue
                static_assert(WM * WN * 32 * 36 * 4 <= SMEM, "wave-private epilogue tiles fit in the operand ring");
                if constexpr (KCH) {
                    if (sc_fused) {           // the layer's sums wait in `tot`, the shortcut's are in `acc`
                        conv_epilogue_wave<BN, TM, TN>(a, tot, reinterpret_cast<float*>(smem) + wave * (32 * 36), s_tab, m0 + wm * TM * 32, n0, wn * TN * 32, lane, acc, s_tab2);
                        return;
                    }
                }
                conv_epilogue_wave<BN, TM, TN>(a, acc, reinterpret_cast<float*>(smem) + wave * (32 * 36), s_tab, m0 + wm * TM * 32, n0, wn * TN * 32, lane);
            } else {
                if constexpr (WAVE_H) {
                    // fp16 tensors (a.direct == 2): a row of the 32 x 64 wave tile is one 128-B line — full-line residual loads and stores
                    if (a.direct == 2) {
                        conv_epilogue_wave_h<BN, TM>(a, acc, reinterpret_cast<float*>(smem) + wave * (32 * 68), s_tab, m0 + wm * TM * 32, n0, wn * TN * 32, lane);
                        return;
                    }
                }
                conv_epilogue_direct<T, BN, TM, TN>(a, acc, s_tab, m0 + wm * TM * 32, n0, wn * TN * 32, lane);
            }
            return;
        }
    }
    conv_epilogue<T, BN, TM, TN, WM, WN, CPASS>(a, acc, smem, m0, n0);
}


static thread_local ConvProfile* g_prof = nullptr;
void conv_set_profiler(ConvProfile* p) { g_prof = p; }
static thread_local int* g_range_flag = nullptr;
void conv_set_range_flag(int* device_flag) { g_range_flag = device_flag; }
void ConvProfile::reset()
{
    for (auto& s : by_tile) s = Slot();
    for (auto& s : by_group) s = Slot();
    by_shape.clear();
    pending.clear();
    used = 0;
}
void ConvProfile::collect()
{
    for (auto& pd : pending) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, pool[pd.e0], pool[pd.e1]));
        by_tile[pd.tile].launches += 1;
        by_tile[pd.tile].ms += ms;
        by_tile[pd.tile].flops += pd.flops;
        by_tile[pd.tile].bytes += pd.bytes;
        Slot& sh = by_shape[pd.shape];
        sh.launches += 1; sh.ms += ms; sh.flops += pd.flops; sh.bytes += pd.bytes;
        Slot& gr = by_group[pd.group == 1 ? 1 : 0];
        gr.launches += 1; gr.ms += ms; gr.flops += pd.flops; gr.bytes += pd.bytes;
    }
    pending.clear();
    used = 0;
}
ConvProfile::~ConvProfile()
{
    for (auto e : pool) (void)hipEventDestroy(e);
}
// ALGORITHMIC bytes of a layer: every operand crosses HBM once — the input pixels the layer reads (a strided 1x1 layer: the sampled ones),
// the filters, the residual / the fused shortcut's input, the outputs it stores (fused heads: their fp32 columns instead of the feature
// tensor; selected-class mode: nothing).  What a memory-bound layer is priced against (bench.py: roofline.by_tile_class[*].frac_of_hbm).
static double conv_algorithmic_bytes(const ConvDesc& d, const ConvDesc* sc = nullptr)
{
    const double es = d.dtype == MRCNN_F16 ? 2.0 : 4.0;
    const int wdt = d.wdtype < 0 ? d.dtype : d.wdtype;
    const double ws = wdt == MRCNN_F32 ? 4.0 : 2.0;
    const double M = (double)d.B * d.OH * d.OW;
    const double ncols = d.deconv2 ? 4.0 * d.Cout : d.Cout;
    double b = 0;
    if (d.algo_k > 0) b += (double)d.B * d.H * d.W * 16.0;                               // the stem's staging tensor: 16 B per padded pixel
    else b += (d.KH * d.KW == 1 ? M : (double)d.B * d.H * d.W) * d.Cin * es;
    b += ncols * d.KH * d.KW * (d.algo_k > 0 ? (double)d.algo_k / (d.KH * d.KW) : (double)d.Cin) * ws;
    if (d.res) b += M * d.Cout * es / (d.res_shift ? 4.0 : 1.0);
    if (sc) b += M * sc->Cin * es + (double)d.Cout * sc->Cin * ws;
    if (d.head_w) b += M * d.head_cols * 4.0;
    else if (!d.sel_partial) b += M * ncols * (d.out_f32 ? 4.0 : es);
    return b;
}
static int prof_event(ConvProfile* p, hipStream_t s)
{
    if (p->used == (int)p->pool.size()) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        p->pool.push_back(e);
    }
    HIP_CHECK(hipEventRecord(p->pool[p->used], s));
    return p->used++;
}

int conv_n_tile(int Cout)
{
    if (Cout > 64) return 128;
    if (Cout > 32) return 64;
    return 32;
}

static int env_int(const char* name, int dflt);
static int g_min_blocks = 448;   // narrow the N tile while the grid has fewer blocks than this: 7/8 of two blocks per CU (the
                                 // box head's 504 tiles of 128 columns beat 1008 of 64: +1.1 % end to end, tools/e2e_ab.py)
static int g_min_blocks_split = 256;      // fp32 tensors (split modes): one block per CU is enough — a narrower tile re-reads the 4-byte activations once more per
                                 // column tile, and on a single image that L2 traffic is what the 1x1 layers wait for: 448 -> 256 is +1.2 .. 1.8 % on one image,
                                 // +0.1 % at batch 8 (profiles/r05_box_path_ab.txt); fp16 tensors keep 448 (-0.3 % at batch 8 with 256).  Same bits either way.
static inline int min_blocks_for(bool split_mode) { return split_mode ? g_min_blocks_split : g_min_blocks; }
static int g_direct = env_int("MRCNN_DIRECT", 3);   // 0: every layer through the block-staged epilogue; 1: fp16 tensors straight from the accumulators;
                                 // 2: also fp32 tensors through wave-private LDS tiles (conv_epilogue_wave); 3: also the fp16 tensors of the
                                 // 128-column kernel (conv_epilogue_wave_h: full-line residual loads and stores; +3.6 % end to end in fp16 mode)
static int g_stem = env_int("MRCNN_STEM", 1);        // conv1 + max-pool as one persistent launch (kernels_conv_stem.hip; split modes: bit-identical to the two launches); 2: fp16 tensors in round 4's four-group form (bit-identical to the two launches; 1 = the compact form: summation noise apart)
static int g_tail_dbg = env_int("MRCNN_TAIL_DBG", 0);        // measurement only: ablation bits of the fused tail's 1x1 phase (1 no epilogue, 2 no K loop)
// Bottleneck tails (3x3 + 1x1) as one persistent launch when the grid fills the chip: bit-identical to the two launches, and
// measured SLOWER (C4, batch 8: 180 us against 88 + 75; ablations: 3x3 loop 91 + barriers / prologues 7 + staging and parking 21 +
// 1x1 K loop 21 + epilogue 37, strictly additive — profiles/r04_tail_ablate_f32x3.txt), so OFF by default;
// MRCNN_TAIL=1 / mrcnn_debug_set("conv_tail", 1) switch it on (tests keep it bit-identical)
static int g_tail = env_int("MRCNN_TAIL", 0);
// fp16 mode: identity bottleneck blocks (branch2a + branch2b + branch2c + shortcut) as ONE persistent launch with the two mid tensors on chip
// (kernels_bneck.hip; bit-identical to the three launches).  MRCNN_BNECK=0 / mrcnn_debug_set("conv_bneck", 0): the three launches.
static int g_bneck = env_int("MRCNN_BNECK", 1);
// ... and (round 6) the consecutive identity blocks of a C = 256 stage — C4: 22 of ResNet-101's blocks — as ONE launch whose tiles wait for their
// neighbours' previous block instead of for a launch boundary (kernels_bneck.hip, STAGE form; bit-identical).  Measured EQUAL to one launch per
// block (profiles/r06_bneck_stage_ab.txt) and dependent on the whole grid being resident, so OFF by default: MRCNN_BNECK_STAGE=1 /
// mrcnn_debug_set("conv_bneck_stage", 1) switch it on.
static int g_bneck_stage = env_int("MRCNN_BNECK_STAGE", 0);
// fp16 mode: 3x3 stride-1 layers with 256 | 512 output columns on the halo-tile / fragment-streaming kernel (kernels_conv3x3_h.hip; its own K order)
// 0: never; 1 (default): where the RPN's heads ride in its epilogue (the engine's P2..P4 levels) — as a plain 3x3 layer it equals the ping-pong kernel
// on the large levels and loses on under-filled grids (profiles/r05_c3h_ab.txt); 2: every eligible layer (tests, A/B); 3: as 1, heads as their own launch (A/B)
static int g_c3h = env_int("MRCNN_C3H", 1);
int conv_c3h_mode() { return g_c3h; }
bool conv_bneck_enabled() { return g_bneck != 0; }
// Canonical K chunks (round 4): the long-K 1x1 layers of the split modes — K >= 2048: C5's `branch2a`, the P5
// lateral, the box head's first inner product (K = 12 544) — sum their K steps as ((0 + P0) + P1) + ..., 4 / 8 equal chunks by the
// layer's shape alone, at EVERY batch.  A grid that fills the chip runs the chunks in one block (a second accumulator set, folded at
// the chunk boundaries); a grid that does not — single images: 32 tiles at C5, 64 in the box head — gives every chunk its own block
// and the last one to finish folds the partial sums in the same order: bit-identical, and the dependent chain of K steps is 4 - 8 x
// shorter (single image: 171 -> 85, 29 -> 18, 28 -> 15 us; batch 8 unchanged).
// MRCNN_KCHUNK=0 / "conv_kchunk" 0: one running sum as in rounds 1-3 (other bits); "conv_ksplit" 0: never share a tile (same bits).
// Fused shortcut (round 4, late): the first block of a ResNet stage convolves the block's input twice — `branch1` (1x1, the shortcut) and,
// three layers later, `branch2c` adds that tensor as its residual.  conv_forward(s, branch2c, &branch1) computes both in ONE launch: the
// shortcut's K loop first (its sums wait in the second accumulator set), then branch2c's, and the epilogue forms the residual from the waiting
// sums with the shortcut's own scale / shift — the same fp32 operations, bit for bit (tests/test_gpu_engine.py), and the 4 x-wide shortcut
// tensor is neither written nor read back (C2: 537 MB each way at batch 8).  "conv_scfuse" 0 / MRCNN_SCFUSE=0: the two launches.
static int g_scfuse = env_int("MRCNN_SCFUSE", 1);
static int g_sel_wave = env_int("MRCNN_SEL_WAVE", 1);         // selected-class mode of the mask head's deconvolution: 1 wave-private epilogue (64-column partial sums), 0 block-staged (128)
int conv_sel_part_cols() { return g_sel_wave ? 64 : 128; }
static int g_kchunk = env_int("MRCNN_KCHUNK", 1);
static int g_ksplit = env_int("MRCNN_KSPLIT", 1);
static int g_ksplit_below = env_int("MRCNN_KSPLIT_BELOW", 256);       // share tiles when the widest-tile grid has fewer blocks than this
static int g_halo = env_int("MRCNN_HALO", 1);        // 3x3 stride-1 layers of the split modes on the halo kernel (kernels_conv_halo.hip) when the filters come re-tiled
static int g_tn4 = -1;       // split modes, 128x128 tile as 4 waves of 32x128: -1 by policy (conv_forward), 0 never, 1 always (tests)
template <typename T, typename TW, int PARTS = 2>
static void conv_launch(hipStream_t s, const ConvArgs& a, int bn, bool wide_waves = false)
{
    // 8 waves as 4 (M) × 2 (N): 128×128 block tile, 32×64 per wave; narrower N tiles keep 128 rows.
    const dim3 grid(a.tiles_m * a.tiles_n * a.ksplit);
#ifndef MRCNN_RING64
#define MRCNN_RING64 3
#endif
#ifndef MRCNN_RING32
#define MRCNN_RING32 4
#endif
#ifndef MRCNN_RING128S
#define MRCNN_RING128S 3    /* split mode: a step is 8 MFMAs per wave, two tiles in flight (+2 %) */
#endif
#ifndef MRCNN_RING128H
#define MRCNN_RING128H 2    /* fp16 tensors: a third stage (96 KB) leaves one block per CU — measured, see DESIGN.md §6 */
#endif
    constexpr int R128 = (sizeof(T) == 4 && sizeof(TW) == 2) ? MRCNN_RING128S : (sizeof(T) == 2 ? MRCNN_RING128H : 2);
    if constexpr (sizeof(T) == 4 && sizeof(TW) == 2) {
        // Split modes, long K: the same 128x128 tile as 4 waves of 32x128 — one register split of an activation fragment feeds
        // four column tiles instead of two (half the split VALU and half the activation-fragment LDS reads per MFMA): +2-4 % on
        // the 3x3 layers, bit-identical (same products, same order per accumulator); short-K layers lose to its 4-wave epilogue.
        if (bn == 128 && wide_waves) { hipLaunchKernelGGL((k_conv_mfma_glds<T, TW, 128, 1, 4, 4, 1, R128, PARTS>), grid, dim3(256), 0, s, a); return; }
    }
    MRCNN_REQUIRE((a.kchunks == 1 && !a.sc_in) || (sizeof(T) == 4 && sizeof(TW) == 2 && bn != 128), MRCNN_ERR_INVALID, "conv: K chunks / a fused shortcut on a kernel that does not carry them");
    if (bn == 128) hipLaunchKernelGGL((k_conv_mfma_glds<T, TW, 128, 1, 2, 4, 2, R128, PARTS>), grid, dim3(512), 0, s, a);
    else if (bn == 64) hipLaunchKernelGGL((k_conv_mfma_glds<T, TW, 64, 1, 1, 4, 2, MRCNN_RING64, PARTS>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((k_conv_mfma_glds<T, TW, 32, 1, 1, 4, 1, MRCNN_RING32, PARTS>), grid, dim3(256), 0, s, a);
}

void conv_pp_launch(hipStream_t s, const ConvArgs& a, int mode);   // kernels_conv_pp.hip: 0 fp16, 2 / 3 split parts

// Run-time switches of the tile choice (A/B measurements through the micro-benchmark hook; defaults = the shipped policy)
static int env_int(const char* name, int dflt)
{
    const char* e = knob_env(name);          // (honoured only with MRCNN_TEST_KNOBS=1: common.h)
    return e && *e ? atoi(e) : dflt;
}
struct PpPolicy { int on, min_tiles, min_kt, dbg, min_fill_pct, split; };
static PpPolicy& pp_policy()
{
    // Shipped policy = where the A/B of tools/conv_ab.py shows a gain (profiles/r02_conv_ab_*.txt): fp16 tensors, at least two
    // full rounds of 256 tiles, K >= 1024 (at K = 512 the in-engine shape table shows a loss: 564 vs 618 TFLOP/s).  The split modes run the same kernel bit-identically but no faster (both kernels
    // sit at the same power-limited MFMA rate): off unless asked for.
    static PpPolicy p = {env_int("MRCNN_PP", 1), env_int("MRCNN_PP_MIN_TILES", 512), env_int("MRCNN_PP_MIN_KT", 16), env_int("MRCNN_PP_DBG", 0),
                         env_int("MRCNN_PP_MIN_FILL", 85), env_int("MRCNN_PP_SPLIT", 0)};
    return p;
}
bool conv_halo_enabled() { return g_halo != 0; }
bool conv_debug_set(const char* key, int value)
{
    const std::string k = key;
    if (k == "conv_pp") pp_policy().on = value;
    else if (k == "conv_pp_min_tiles") pp_policy().min_tiles = value;
    else if (k == "conv_pp_min_kt") pp_policy().min_kt = value;
    else if (k == "conv_pp_dbg") pp_policy().dbg = value;
    else if (k == "conv_pp_min_fill") pp_policy().min_fill_pct = value;
    else if (k == "conv_pp_split") pp_policy().split = value;
    else if (k == "conv_tn4") g_tn4 = value;
    else if (k == "conv_halo") g_halo = value;
    else if (k == "conv_direct") g_direct = value;
    else if (k == "conv_min_blocks") g_min_blocks = g_min_blocks_split = value;
    else if (k == "conv_min_blocks_split") g_min_blocks_split = value;
    else if (k == "conv_scfuse") g_scfuse = value;
    else if (k == "mask_sel_wave") g_sel_wave = value;
    else if (k == "conv_kchunk") g_kchunk = value;
    else if (k == "conv_ksplit") g_ksplit = value;
    else if (k == "conv_ksplit_below") g_ksplit_below = value;
    else if (k == "conv_tail") g_tail = value;
    else if (k == "conv_stem") g_stem = value;
    else if (k == "conv_tail_dbg") g_tail_dbg = value;
    else if (k == "conv_bneck") g_bneck = value;
    else if (k == "conv_bneck_stage") g_bneck_stage = value;
    else if (k == "conv_c3h") g_c3h = value;
    else return conv_halo_debug_set(key, value);
    return true;
}

// The launch arguments every kernel of the family shares (geometry, strides, epilogue fields) from a layer description.
static void conv_fill_args(const ConvDesc& d, ConvArgs& a)
{
    const bool half = d.dtype == MRCNN_F16;
    a.in = d.in; a.wgt = d.wgt; a.scale = d.scale; a.shift = d.shift; a.res = d.res; a.out = d.out; a.out2 = d.out2;
    a.in_sB = d.in_sB; a.in_sH = d.in_sH; a.in_sW = d.in_sW;
    a.res_sB = d.res_sB; a.res_sH = d.res_sH; a.res_sW = d.res_sW;
    a.out_sB = d.out_sB; a.out_sP = d.out_sP; a.out_sH = d.out_sH; a.out_sW = d.out_sW;
    a.out2_sB = d.out2_sB; a.out2_sP = d.out2_sP;
    a.B = d.B; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.KH = d.KH; a.KW = d.KW; a.stride = d.stride; a.padH = d.padH; a.padW = d.padW;
    a.OH = d.OH; a.OW = d.OW; a.Cout = d.Cout;
    a.ncols = d.deconv2 ? 4 * d.Cout : d.Cout;
    a.Ktot = d.KH * d.KW * d.Cin;
    const long M = (long)d.B * d.OH * d.OW;
    MRCNN_REQUIRE(M > 0 && M < (1L << 31) - 256, MRCNN_ERR_SHAPE, "conv: M out of range");
    a.M = (int)M;
    a.res_shift = d.res_shift; a.act = d.act; a.n_split = d.n_split; a.deconv2 = d.deconv2;
    a.out_f32 = (!half || d.out_f32) ? 1 : 0;
    a.range_flag = g_range_flag;
    a.dbg = pp_policy().dbg;
    a.sel_w = d.sel_w; a.sel_cid = d.sel_cid; a.sel_partial = d.sel_partial;
    a.kchunks = 1; a.ksplit = 1; a.ks_scratch = nullptr; a.ks_count = nullptr;
    a.sel_part_cols = 128;
    a.sc_in = nullptr; a.sc_wgt = nullptr; a.sc_scale = nullptr; a.sc_shift = nullptr;
    a.sc_in_sB = a.sc_in_sH = a.sc_in_sW = 0; a.sc_H = a.sc_W = a.sc_Cin = 0; a.sc_stride = 1;
}

// Canonical K chunks of a layer (1 = one running sum): by its shape and mode alone — never by the batch or the grid
int conv_k_chunks(const ConvDesc& d)
{
    const int wdtype = d.wdtype < 0 ? d.dtype : d.wdtype;
    const bool split = d.dtype == MRCNN_F32 && (wdtype == MRCNN_F16 || wdtype == MRCNN_F32X3);
    // (K = 1024 — C4's branch2a, the box head's second inner product — is left as one sum: two chunks gain 1.2 us per launch on a
    //  single image and cost 1.5 us at batch 8 for the second accumulator set, gpurun_out/r4w)
    if (!g_kchunk || !split || d.KH != 1 || d.KW != 1 || d.Cin < 2048 || d.Cin % 32 != 0 || d.head_w) return 1;
    int n = d.Cin >= 8192 ? 8 : 4;
    while ((d.Cin / 32) % n) n >>= 1;
    return n;
}

// Partial sums of shared tiles: 64 MB + 8192 counters per stream, allocated at the stream's first shared launch and kept (a
// captured graph holds the pointers; the engine's first predict at a batch size runs eagerly, so the allocation never falls
// inside a capture).  Launches on one stream are ordered, and a launch leaves its counters at zero.
static constexpr size_t KS_BYTES = 64u << 20;
static constexpr int KS_TILES = 8192;
static thread_local ConvScratch* g_scratch = nullptr;
void conv_set_scratch(ConvScratch* c) { g_scratch = c; }
size_t ConvScratch::ks_bytes() { return KS_BYTES; }
void ConvScratch::alloc()
{
    ks_buf.alloc(KS_BYTES);
    ks_cnt.alloc(KS_TILES * sizeof(unsigned));
    HIP_CHECK(hipMemset(ks_cnt.p, 0, KS_TILES * sizeof(unsigned)));       // (every launch leaves the counters at zero)
}
ConvScratch* conv_current_scratch() { return g_scratch; }

// May the epilogue use 16-B vector stores / residual loads for this layer?
static int conv_vec_ok(const ConvDesc& d, const ConvArgs& a)
{
    const bool half = d.dtype == MRCNN_F16;
    auto al = [](const void* p, size_t n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
    const int cpt = half ? 8 : 4;        // columns per epilogue thread: 16 B of the activation type
    return a.ncols % cpt == 0 && d.out2 == nullptr && d.out_sP % cpt == 0 && d.out_sB % cpt == 0 && al(d.out, 16) &&
               (!d.scale || al(d.scale, 16)) && (!d.shift || al(d.shift, 16)) &&
               (!d.res || (d.res_sW % cpt == 0 && d.res_sH % cpt == 0 && d.res_sB % cpt == 0 && al(d.res, 16))) &&
               (!d.deconv2 || (d.Cout % cpt == 0 && d.out_sH % cpt == 0 && d.out_sW % cpt == 0));
}

// May `main` (a 1x1 layer whose residual is exactly the output of the 1x1 layer `sc`) absorb `sc`?  (conv_forward checks the tile and the epilogue form.)
static bool conv_shortcut_fusable(const ConvDesc& m, const ConvDesc& sc)
{
    const int wm = m.wdtype < 0 ? m.dtype : m.wdtype, ws = sc.wdtype < 0 ? sc.dtype : sc.wdtype;
    if (m.dtype != MRCNN_F32 || sc.dtype != MRCNN_F32 || wm != ws || !(wm == MRCNN_F16 || wm == MRCNN_F32X3)) return false;          // split modes
    if (m.KH != 1 || m.KW != 1 || sc.KH != 1 || sc.KW != 1 || m.padH || m.padW || sc.padH || sc.padW || m.stride != 1) return false;
    if (m.B != sc.B || m.OH != sc.OH || m.OW != sc.OW || m.Cout != sc.Cout || m.Npad != sc.Npad || sc.Cin % 32 != 0) return false;
    if (sc.act != ACT_NONE || sc.res || sc.out2 || sc.deconv2 || sc.sel_partial || sc.head_w || m.out2 || m.deconv2 || m.sel_partial || m.head_w || m.act == ACT_SIGMOID) return false;
    // the residual IS the shortcut's output, element for element
    if (m.res != sc.out || m.res_shift != 0 || m.res_sW != sc.out_sP || m.res_sB != sc.out_sB || m.res_sH != (long)m.OW * m.res_sW) return false;
    if (sc.out_sB != (long)sc.OH * sc.OW * sc.out_sP) return false;
    return true;
}

void conv_forward(hipStream_t s, const ConvDesc& d_in, const ConvDesc* sc)
{
    // a shortcut that cannot ride in this launch runs first, as its own (its output is this layer's residual)
    bool fuse = sc && g_scfuse && conv_shortcut_fusable(d_in, *sc) && conv_k_chunks(d_in) == 1 && conv_k_chunks(*sc) == 1;
    if (fuse) {
        // the fused form needs the direct fp32 epilogue (tiles of 64 columns or more)
        const int bn_max = conv_n_tile(d_in.Cout);
        int bn = bn_max;
        const long tiles_m = ((long)d_in.B * d_in.OH * d_in.OW + BM_DEFAULT - 1) / BM_DEFAULT;
        while (bn > 32 && tiles_m * (d_in.Npad / bn) < min_blocks_for(true)) bn >>= 1;          // (fused shortcuts exist in the split modes only)
        ConvDesc probe = d_in;
        probe.res = nullptr;
        ConvArgs pa;
        conv_fill_args(probe, pa);
        if (bn < 64 || g_direct < 2 || !conv_vec_ok(probe, pa)) fuse = false;
#ifndef MRCNN_CONV_ABLATE
        if (pp_policy().dbg) fuse = false;          // (the measurement build ablates the fused launch too)
#endif
    }
    if (sc && !fuse) conv_forward(s, *sc, nullptr);
    ConvDesc d = d_in;
    if (fuse) { d.res = nullptr; d.res_sB = d.res_sH = d.res_sW = 0; }
    const bool half = d.dtype == MRCNN_F16;
    const int wdtype = d.wdtype < 0 ? d.dtype : d.wdtype;
    if (half && g_c3h && (d.head_w || d.prefer_c3h || g_c3h == 2) && conv3x3h_eligible(d)) {
        static int n_cus_c3h = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
        ConvProfile* prof = (g_prof && g_prof->active) ? g_prof : nullptr;
        const int e0 = prof ? prof_event(prof, s) : 0;
        conv3x3h_launch(s, d, g_range_flag, n_cus_c3h);
        if (prof) {
            const int e1 = prof_event(prof, s);
            const double M = (double)d.B * d.OH * d.OW;
            prof->pending.push_back({8, 2.0 * M * d.Cout * 9.0 * d.Cin, e0, e1, {(int)M, d.Cout, 9 * d.Cin, 8}, d.group, conv_algorithmic_bytes(d)});
        }
        return;
    }
    // fp32 activations, fp16 filters: two-pass (wdtype F16) or exact three-pass (wdtype F32X3, a filter-side tag) fp16 MFMA
    const bool split = d.dtype == MRCNN_F32 && (wdtype == MRCNN_F16 || wdtype == MRCNN_F32X3);
    MRCNN_REQUIRE(d.dtype == MRCNN_F32 || half, MRCNN_ERR_UNSUPPORTED, "conv: dtype %d", d.dtype);
    MRCNN_REQUIRE(wdtype == d.dtype || split, MRCNN_ERR_UNSUPPORTED, "conv: activation dtype %d with filter dtype %d", d.dtype, wdtype);
    const int bk = half ? 64 : 32;
    MRCNN_REQUIRE(d.Cin % bk == 0, MRCNN_ERR_SHAPE, "conv: Cin %d not a multiple of %d", d.Cin, bk);
    ConvArgs a;
    conv_fill_args(d, a);
    MRCNN_REQUIRE(!d.sel_partial || (d.deconv2 && d.sel_w && d.sel_cid && d.Cout % 128 == 0 && d.Npad == 4 * d.Cout && !d.out2),
                  MRCNN_ERR_INVALID, "conv: the selected-class mode needs a 2x2 transposed convolution with Cout a multiple of 128");
    // Tile choice: the widest N tile the packed weights allow, narrowed while the grid would leave
    // the chip under-filled (< 7/8 of 2 blocks per CU) — C5, the top FPN levels and the small RPN levels.
    const int bn_max = conv_n_tile(a.ncols);
    MRCNN_REQUIRE(d.Npad % bn_max == 0 && d.Npad >= a.ncols, MRCNN_ERR_SHAPE, "conv: Npad %d incompatible with tile %d", d.Npad, bn_max);
    a.tiles_m = (a.M + BM_DEFAULT - 1) / BM_DEFAULT;
    int bn = bn_max;
    while (bn > 32 && !d.sel_partial && (long)a.tiles_m * (d.Npad / bn) < min_blocks_for(split)) bn >>= 1;     // (selected-class mode: fixed 128-channel parts)
    if (fuse) {
        a.sc_in = sc->in; a.sc_wgt = sc->wgt; a.sc_scale = sc->scale; a.sc_shift = sc->shift;
        a.sc_in_sB = sc->in_sB; a.sc_in_sH = sc->in_sH; a.sc_in_sW = sc->in_sW;
        a.sc_H = sc->H; a.sc_W = sc->W; a.sc_Cin = sc->Cin; a.sc_stride = sc->stride;
    }
    // (a fused shortcut needs the second accumulator set: the eight-wave 128-column form has no registers for it; of the two forms that do,
    //  the four-wave 128-column one measured C2 / C3 / C4 / C5 321 / 247 / 171 / 150 us against 355 / 282 / 221 / 193 for the eight-wave
    //  64-column one, and 430 / 288 / 180 / 159 for the two launches each replaces: gpurun_out/r5g)
    a.kchunks = d.sel_partial ? 1 : conv_k_chunks(d);
    if (a.kchunks > 1 && g_ksplit && (long)a.tiles_m * (d.Npad / bn_max) < g_ksplit_below) {
        // an under-filled grid: one block per (tile, chunk), the N tile as wide as the shared grid allows
        int bs = bn_max;
        while (bs > 32 && (long)a.tiles_m * (d.Npad / bs) * a.kchunks < min_blocks_for(true)) bs >>= 1;
        const long tiles = (long)a.tiles_m * (d.Npad / bs);
        // (the shared-tile form needs the owner's scratch: a launch without one keeps its chunks in one block — the same bits)
        if (g_scratch && g_scratch->ks_buf.p && tiles <= KS_TILES && (size_t)tiles * a.kchunks * BM_DEFAULT * bs * 4 <= KS_BYTES) {
            bn = bs;
            a.ksplit = a.kchunks;
            a.ks_scratch = g_scratch->ks_buf.as<float>();
            a.ks_count = g_scratch->ks_cnt.as<unsigned>();
        }
    }
    a.vec_ok = conv_vec_ok(d, a);
    a.tiles_n = d.Npad / bn;
    MRCNN_REQUIRE(!d.sel_partial || (bn == 128 && a.vec_ok), MRCNN_ERR_INVALID, "conv: the selected-class mode needs the 128-wide vector epilogue");
    // Epilogue without block barriers wherever the layer allows: fp32 tensors through wave-private LDS tiles (conv_epilogue_wave:
    // full-line stores — a lane's own 16-B store would hold four channels of one pixel, 64 scattered pieces per instruction, which
    // lost 1.4 % — without the two barriers of the block-staged form: +1.6 % end to end, tools/e2e_direct_ab.sh, round 3); fp16
    // tensors of the 128-column kernel the same way (conv_epilogue_wave_h: a row of the 32 x 64 wave tile is one 128-B line;
    // +3.6 % end to end in fp16 mode over the form below, tools/e2e_ab.py f16 conv_direct 2 3), the narrower fp16 tiles straight
    // from the accumulators (conv_epilogue_direct: 32-B pieces per pixel and store, +0.9 % over the block-staged form).
    a.direct = (g_direct && (half || g_direct > 1) && a.vec_ok && !d.out2 && !d.deconv2 && d.act != ACT_SIGMOID && (!half || !a.out_f32)) ? 1 : 0;
    MRCNN_REQUIRE(!fuse || (a.direct && bn >= 64), MRCNN_ERR_INVALID, "conv: fused shortcut without the direct epilogue (conv_forward's own check should have said so)");
    if (a.direct && half && g_direct > 2) a.direct = 2;      // fp16 tensors through wave-private tiles where the wave tile is 32 x 64 (the 128-column kernel)
    // selected-class mode: the wave-private form (64-column partial sums straight from the accumulators) on the eight-wave 128-column kernel
    if (d.sel_partial && g_sel_wave && bn == 128 && a.vec_ok && d.act != ACT_SIGMOID && d.Cout % 64 == 0) { a.direct = 1; a.sel_part_cols = 64; }
    // Layers with a large GEMM: the 256×256 persistent ping-pong kernel (kernels_conv_pp.hip), one block per CU — when the
    // tiles fill whole rounds of the chip well enough (a static walk: the last round costs as much as a full one).
    int pp_bn = 0;
    {
        const PpPolicy& pol = pp_policy();
        const long tiles = (long)((a.M + 255) / 256) * (d.Npad / 256);
        const long rounds = (tiles + 255) / 256;
        const bool fills = tiles >= pol.min_tiles && tiles * 100 >= rounds * 256 * pol.min_fill_pct;
        const int bk_pp = half ? 64 : 32;
        if (pol.on && (half || (split && pol.split)) && d.Cin % bk_pp == 0 && a.Ktot / bk_pp >= pol.min_kt && d.Npad % 256 == 0 && fills &&
            a.kchunks == 1 && !fuse && a.vec_ok && (!half || !a.out_f32) && !d.deconv2 && !d.out2 && d.act != ACT_SIGMOID && d.H < 32760 && d.W < 32760)
            pp_bn = 256;                    // ... and the epilogue / address forms pp_store_tile and PP_SRC_A cover
    }
    if (pp_bn) { a.tiles_m = (a.M + 255) / 256; a.tiles_n = d.Npad / pp_bn; }
    // K >= 2048: the 3x3 layers; and every chunked layer (the 8-wave form has no registers for the second accumulator set)
    const bool wide_waves = split && bn == 128 && (a.kchunks > 1 || fuse || (g_tn4 < 0 ? a.Ktot / bk >= 64 : g_tn4 != 0));
    ConvProfile* prof = (g_prof && g_prof->active) ? g_prof : nullptr;
    const int e0 = prof ? prof_event(prof, s) : 0;
    // 3x3 stride-1 layers of the split modes: the persistent halo kernel, whenever the layer qualifies — by its geometry and
    // mode alone, so that a layer runs in ONE summation order whatever the batch (the kernel's K order is its own).
    const bool halo = g_halo && d.wgt_halo && a.vec_ok && conv_halo_eligible(d);
    MRCNN_REQUIRE(!d.head_w || (halo && conv_halo_head_eligible(d)), MRCNN_ERR_INVALID, "conv: a fused head needs the halo kernel (layer not eligible, or switched off)");
    if (halo) {
        static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
        pp_bn = 0;
        (void)conv_halo_forward(s, a, d, wdtype == MRCNN_F32X3 ? 3 : 2, n_cus);
    } else
    if (pp_bn) conv_pp_launch(s, a, half ? 0 : (wdtype == MRCNN_F32X3 ? 3 : 2));
    else if (half) conv_launch<_Float16, _Float16>(s, a, bn);
    else if (split && wdtype == MRCNN_F32X3) conv_launch<float, _Float16, 3>(s, a, bn, wide_waves);
    else if (split) conv_launch<float, _Float16, 2>(s, a, bn, wide_waves);
    else conv_launch<float, float>(s, a, bn);
    if (prof) {
        const int e1 = prof_event(prof, s);
        const double k = (d.algo_k > 0 ? d.algo_k : a.Ktot) + (fuse ? sc->Cin : 0);          // (a fused shortcut: both K loops)
        const int tile = halo ? 5 : pp_bn == 256 ? 4 : (wide_waves ? 3 : (bn == 128 ? 0 : (bn == 64 ? 1 : 2)));
        prof->pending.push_back({tile, 2.0 * (double)a.M * (double)a.ncols * k, e0, e1, {a.M, a.ncols, a.Ktot, tile}, d.group, conv_algorithmic_bytes(d, fuse ? sc : nullptr)});
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Fused stem (kernels.h: conv_stem_forward)
// ------------------------------------------------------------------------------------------------
bool conv_stem_enabled() { return g_stem != 0; }
bool conv_stem_eligible(const ConvDesc& d)
{
    const int wdtype = d.wdtype < 0 ? d.dtype : d.wdtype;
    const bool split = d.dtype == MRCNN_F32 && (wdtype == MRCNN_F16 || wdtype == MRCNN_F32X3);
    const bool half = d.dtype == MRCNN_F16 && wdtype == MRCNN_F16 && !d.out_f32;
    const int pxc = half ? 8 : 4;          // channels of a staged pixel: 16 B either way
    return (split || half) && d.KH == 7 && d.KW == 1 && d.Cin == 8 * pxc && d.stride == 2 && d.padH == 0 && d.padW == 0 && d.Cout == 64 && d.Npad == 64 &&
           d.act == ACT_RELU && !d.res && !d.out2 && !d.deconv2 && d.in_sW == pxc && d.in_sH == (long)d.W * pxc && d.in_sB == (long)d.H * d.W * pxc &&
           d.OH == (d.H - 7) / 2 + 1 && d.OW == (d.W - 7) / 2 + 1;
}

void conv_stem_forward(hipStream_t s, const ConvDesc& d, void* pooled, int PH, int PW)
{
    MRCNN_REQUIRE(conv_stem_eligible(d) && pooled && PH == (d.OH + 1) / 2 && PW == (d.OW + 1) / 2, MRCNN_ERR_INVALID, "conv_stem_forward: not the stem layer");
    static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
    ConvProfile* prof = (g_prof && g_prof->active) ? g_prof : nullptr;
    const int e0 = prof ? prof_event(prof, s) : 0;
    const int wdtype = d.wdtype < 0 ? d.dtype : d.wdtype;
    conv_stem_launch(s, d.in, d.B, d.H, d.W, d.wgt, d.scale, d.shift, d.OH, d.OW, pooled, PH, PW,
                     d.dtype == MRCNN_F16 ? 1 : (wdtype == MRCNN_F32X3 ? 3 : 2), g_range_flag, n_cus, g_stem != 2);
    if (prof) {
        const int e1 = prof_event(prof, s);
        const long M = (long)d.B * d.OH * d.OW;
        const double k = d.algo_k > 0 ? d.algo_k : d.KH * d.KW * d.Cin;
        // (the 64-column class of the table: the layer it replaces ran there; the pool rides in the same launch)
        // algorithmic bytes: the staging tensor in, the POOLED tensor out (conv1's own output never exists)
        const double by = (double)d.B * d.H * d.W * 16.0 + 64.0 * k * 2.0 + (double)d.B * PH * PW * 64.0 * (d.dtype == MRCNN_F16 ? 2.0 : 4.0);
        prof->pending.push_back({1, 2.0 * (double)M * 64.0 * k, e0, e1, {(int)M, 64, d.KH * d.KW * d.Cin, 1}, d.group, by});
    }
}

// ------------------------------------------------------------------------------------------------
// Fused bottleneck tail (kernels.h: conv_forward_tail)
// ------------------------------------------------------------------------------------------------
bool conv_tail_fusable(const ConvDesc& d3, const ConvDesc& d1)
{
    const int w3 = d3.wdtype < 0 ? d3.dtype : d3.wdtype, w1 = d1.wdtype < 0 ? d1.dtype : d1.wdtype;
    if (d3.dtype != MRCNN_F32 || d1.dtype != MRCNN_F32 || w3 != w1 || !(w3 == MRCNN_F16 || w3 == MRCNN_F32X3)) return false;      // split modes only
    if (!d3.wgt_halo || !d1.wgt_halo || !conv_halo_eligible(d3) || d3.head_w || !conv_halo_tail_geometry_ok(d3.H, d3.W)) return false;
    if (d3.Cout != 256 || d3.Npad != 256 || d3.act != ACT_RELU) return false;
    if (!conv_halo_tail_packable(d1.KH, d1.KW, d1.Cin, d1.Npad) || d1.Cout != 1024 || d1.stride != 1 || d1.padH != 0 || d1.padW != 0) return false;
    if (d1.deconv2 || d1.out2 || d1.sel_partial || d1.act == ACT_SIGMOID || d1.res_shift || d1.head_w) return false;
    // the 1x1 reads exactly what the 3x3 writes: a dense NHWC tensor of 256 channels
    if (d1.in != d3.out || d1.B != d3.B || d1.H != d3.OH || d1.W != d3.OW || d1.OH != d3.OH || d1.OW != d3.OW) return false;
    if (d3.out_sP != 256 || d3.out_sB != (long)d3.OH * d3.OW * 256 || d1.in_sW != 256 || d1.in_sH != (long)d3.OW * 256 || d1.in_sB != d3.out_sB) return false;
    return true;
}

void conv_forward_tail(hipStream_t s, const ConvDesc& d3, const ConvDesc& d1, const ConvDesc* sc)
{
    static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
    const long tiles = ((long)d3.B * d3.OH * d3.OW + 127) / 128;
    ConvArgs a3, a1;
    bool fuse = g_tail && g_halo && g_scratch && conv_tail_fusable(d3, d1) && tiles * 8 >= (long)n_cus * 7;       // a grid that fills the chip: one 128 x 256 tile per block (and an owner for the parking buffer)
    if (sc && fuse) { conv_forward(s, *sc); sc = nullptr; }          // (the fused tail reads its residual from memory: the shortcut runs as its own launch)
    if (fuse) {
        conv_fill_args(d3, a3);
        conv_fill_args(d1, a1);
        a3.vec_ok = conv_vec_ok(d3, a3);
        a1.vec_ok = conv_vec_ok(d1, a1);
        a3.tiles_m = a1.tiles_m = 0; a3.tiles_n = a1.tiles_n = 0; a3.direct = a1.direct = 1;
        a1.dbg = g_tail_dbg;
        fuse = a3.vec_ok && a1.vec_ok && a3.dbg == 0;
    }
    if (!fuse) {
        conv_forward(s, d3);
        conv_forward(s, d1, sc);
        return;
    }
    ConvProfile* prof = (g_prof && g_prof->active) ? g_prof : nullptr;
    const int e0 = prof ? prof_event(prof, s) : 0;
    const int w3 = d3.wdtype < 0 ? d3.dtype : d3.wdtype;
    (void)conv_halo_forward(s, a3, d3, w3 == MRCNN_F32X3 ? 3 : 2, n_cus, &a1, d1.wgt_halo);
    if (prof) {
        const int e1 = prof_event(prof, s);
        // one launch, two layers: algorithmic flops of both; the shape key is the 3x3 layer's M and K with the 1x1's N (tile class 6)
        const double fl = 2.0 * (double)a3.M * ((double)a3.ncols * a3.Ktot + (double)a1.ncols * a1.Ktot);
        ConvDesc d3n = d3; d3n.sel_partial = (float*)1;          // (the tensor between the two layers is not stored: count d3 without its output)
        prof->pending.push_back({6, fl, e0, e1, {a3.M, a1.ncols, a3.Ktot + a1.Ktot, 6}, d3.group, conv_algorithmic_bytes(d3n) + conv_algorithmic_bytes(d1) - (double)a3.M * d1.Cin * 4.0});
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Fused identity bottleneck of the fp16 mode (kernels.h: conv_bneck_forward)
// ------------------------------------------------------------------------------------------------
bool conv_bneck_fusable(const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc)
{
    auto f16 = [](const ConvDesc& d) { return d.dtype == MRCNN_F16 && (d.wdtype < 0 || d.wdtype == MRCNN_F16) && !d.out_f32; };
    if (!f16(da) || !f16(db) || !f16(dc)) return false;
    const int C = da.Cout, H = da.H, W = da.W;
    if (!(C == 64 || C == 128 || C == 256) || !bneck_geometry_ok(C, H, W)) return false;
    auto plain = [](const ConvDesc& d) { return !d.out2 && !d.deconv2 && !d.sel_partial && !d.head_w && d.res_shift == 0 && d.act == ACT_RELU && d.scale && d.shift; };
    if (!plain(da) || !plain(db) || !plain(dc) || da.res || db.res) return false;
    auto dense_in = [](const ConvDesc& d, int h, int w, int c) { return d.H == h && d.W == w && d.Cin == c && d.in_sW == c && d.in_sH == (long)w * c && d.in_sB == (long)h * w * c; };
    auto dense_out = [](const ConvDesc& d, int h, int w, int c) { return d.OH == h && d.OW == w && d.Cout == c && d.out_sP == c && d.out_sB == (long)h * w * c; };
    if (da.KH != 1 || da.KW != 1 || da.stride != 1 || da.padH || da.padW || !dense_in(da, H, W, 4 * C) || !dense_out(da, H, W, C)) return false;
    if (db.KH != 3 || db.KW != 3 || db.stride != 1 || db.padH != 1 || db.padW != 1 || db.in != da.out || !dense_in(db, H, W, C) || !dense_out(db, H, W, C)) return false;
    if (dc.KH != 1 || dc.KW != 1 || dc.stride != 1 || dc.padH || dc.padW || dc.in != db.out || !dense_in(dc, H, W, C) || !dense_out(dc, H, W, 4 * C)) return false;
    if (dc.res != da.in || dc.res_sW != 4 * C || dc.res_sH != (long)W * 4 * C || dc.res_sB != (long)H * W * 4 * C) return false;
    if (dc.out == da.in) return false;                // a tile reads halo pixels its neighbours own
    if (da.B != db.B || da.B != dc.B) return false;
    return true;
}

void conv_bneck_forward(hipStream_t s, const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc)
{
    static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
    // One tile per block: a grid that leaves the chip under-filled (single images: 32 tiles at C4) runs the three launches, whose narrower
    // tiles spread over more CUs — the two forms agree bit for bit, so the choice may follow the batch (measured at batch 1, fp16 mode:
    // 2.66 ms per image with the three launches, 3.59 fused everywhere; batch 8: 9.87 -> 9.52 ms fused)
    const int C_ = da.Cout;
    const long ntiles = (long)da.B * (da.H / (C_ == 256 ? 8 : 16)) * (da.W / 16);
    if (!g_bneck || !conv_bneck_fusable(da, db, dc) || (g_bneck < 3 && ntiles * 8 < (long)n_cus * 7)) {      // ("conv_bneck" 3: fused at every grid size, tests)
        conv_forward(s, da);
        conv_forward_tail(s, db, dc, nullptr);
        return;
    }
    ConvProfile* prof = (g_prof && g_prof->active) ? g_prof : nullptr;
    const int e0 = prof ? prof_event(prof, s) : 0;
    bneck_launch(s, da.Cout, da.in, dc.out, da.B, da.H, da.W, da.wgt, db.wgt, dc.wgt, da.scale, da.shift, db.scale, db.shift, dc.scale, dc.shift,
                 g_range_flag, n_cus, g_bneck >= 2 ? nullptr : db.wgt_frag, g_bneck >= 2 ? nullptr : dc.wgt_frag, g_bneck >= 2 ? nullptr : da.wgt_frag);      // ("conv_bneck" 2: every operand through LDS)
    if (prof) {
        const int e1 = prof_event(prof, s);
        const double M = (double)da.B * da.H * da.W, C = da.Cout;
        const double fl = 2.0 * M * (4 * C * C + 9 * C * C + 4 * C * C);      // algorithmic flops of the three layers (the halo recompute is not work)
        prof->pending.push_back({7, fl, e0, e1, {(int)M, 4 * da.Cout, 17 * da.Cout, 7}, da.group, 2.0 * M * 4 * C * 2.0 + 17.0 * C * C * 2.0});      // x in + y out + the filters
    }
}

void conv_bneck_stage_forward(hipStream_t s, const BneckTriple* blocks, int n, const void* layers_dev, unsigned* done)
{
    static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
    bool stage = n >= 2 && g_bneck == 1 && g_bneck_stage && layers_dev && done && g_range_flag;
    if (stage) {
        const ConvDesc& a0 = blocks[0].a;
        const long ntiles = (long)a0.B * (a0.H / 8) * (a0.W / 16);
        stage = a0.Cout == 256 && ntiles * 8 >= (long)n_cus * 7;           // the per-block rule: an under-filled grid runs the three launches
        for (int i = 0; i < n && stage; ++i) {
            const BneckTriple& t = blocks[i];
            stage = conv_bneck_fusable(t.a, t.b, t.c) && t.a.wgt_frag && t.b.wgt_frag && t.c.wgt_frag && t.a.Cout == 256 && t.a.B == a0.B && t.a.H == a0.H && t.a.W == a0.W;
            if (i > 0) stage = stage && t.a.in == blocks[i - 1].c.out && (i < 2 || t.c.out == blocks[i - 2].c.out);      // a chain between two tensors
        }
        stage = stage && blocks[1].c.out == a0.in;
    }
    if (!stage) {
        for (int i = 0; i < n; ++i) conv_bneck_forward(s, blocks[i].a, blocks[i].b, blocks[i].c);
        return;
    }
    const ConvDesc& a0 = blocks[0].a;
    ConvProfile* prof = (g_prof && g_prof->active) ? g_prof : nullptr;
    const int e0 = prof ? prof_event(prof, s) : 0;
    bneck_stage_launch(s, layers_dev, n, const_cast<void*>(a0.in), blocks[0].c.out, a0.B, a0.H, a0.W, done, g_range_flag, n_cus);
    if (prof) {
        const int e1 = prof_event(prof, s);
        const double M = (double)a0.B * a0.H * a0.W, C = a0.Cout;
        const double fl = 2.0 * M * (4 * C * C + 9 * C * C + 4 * C * C) * n;      // algorithmic flops of the n blocks
        prof->pending.push_back({7, fl, e0, e1, {(int)M, 4 * a0.Cout, 17 * a0.Cout * n, 7}, a0.group, (2.0 * M * 4 * C * 2.0 + 17.0 * C * C * 2.0) * n});
    }
}

bool conv_bneck_first_fusable(const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc, const ConvDesc& ds)
{
    auto f16 = [](const ConvDesc& d) { return d.dtype == MRCNN_F16 && (d.wdtype < 0 || d.wdtype == MRCNN_F16) && !d.out_f32; };
    if (!f16(da) || !f16(db) || !f16(dc) || !f16(ds)) return false;
    const int C = da.Cout, H = da.H, W = da.W;
    if (C != 64 || da.Cin != C || !bneck_geometry_ok(C, H, W)) return false;
    auto plain = [](const ConvDesc& d, int act) { return !d.out2 && !d.deconv2 && !d.sel_partial && !d.head_w && d.res_shift == 0 && d.act == act && d.scale && d.shift; };
    if (!plain(da, ACT_RELU) || !plain(db, ACT_RELU) || !plain(dc, ACT_RELU) || !plain(ds, ACT_NONE) || da.res || db.res || ds.res) return false;
    auto dense_in = [](const ConvDesc& d, int h, int w, int c) { return d.H == h && d.W == w && d.Cin == c && d.in_sW == c && d.in_sH == (long)w * c && d.in_sB == (long)h * w * c; };
    auto dense_out = [](const ConvDesc& d, int h, int w, int c) { return d.OH == h && d.OW == w && d.Cout == c && d.out_sP == c && d.out_sB == (long)h * w * c; };
    auto pw = [](const ConvDesc& d) { return d.KH == 1 && d.KW == 1 && d.stride == 1 && !d.padH && !d.padW; };
    if (!pw(da) || !dense_in(da, H, W, C) || !dense_out(da, H, W, C)) return false;
    if (!pw(ds) || ds.in != da.in || !dense_in(ds, H, W, C) || !dense_out(ds, H, W, 4 * C)) return false;
    if (db.KH != 3 || db.KW != 3 || db.stride != 1 || db.padH != 1 || db.padW != 1 || db.in != da.out || !dense_in(db, H, W, C) || !dense_out(db, H, W, C)) return false;
    if (!pw(dc) || dc.in != db.out || !dense_in(dc, H, W, C) || !dense_out(dc, H, W, 4 * C)) return false;
    if (dc.res != ds.out || dc.res_sW != 4 * C || dc.res_sH != (long)W * 4 * C || dc.res_sB != (long)H * W * 4 * C) return false;
    if (dc.out == da.in || da.B != db.B || da.B != dc.B || da.B != ds.B) return false;
    return true;
}

void conv_bneck_first_forward(hipStream_t s, const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc, const ConvDesc& ds)
{
    static int n_cus = [] { int dev = 0; hipDeviceProp_t p; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
    const long ntiles = (long)da.B * (da.H / 16) * (da.W / 16);
    if (!g_bneck || !conv_bneck_first_fusable(da, db, dc, ds) || (g_bneck < 3 && ntiles * 8 < (long)n_cus * 7)) {
        conv_forward(s, da);
        conv_forward_tail(s, db, dc, &ds);
        return;
    }
    ConvProfile* prof = (g_prof && g_prof->active) ? g_prof : nullptr;
    const int e0 = prof ? prof_event(prof, s) : 0;
    bneck_launch(s, da.Cout, da.in, dc.out, da.B, da.H, da.W, da.wgt, db.wgt, dc.wgt, da.scale, da.shift, db.scale, db.shift, dc.scale, dc.shift,
                 g_range_flag, n_cus, nullptr, nullptr, nullptr, ds.wgt, ds.scale, ds.shift);
    if (prof) {
        const int e1 = prof_event(prof, s);
        const double M = (double)da.B * da.H * da.W, C = da.Cout;
        const double fl = 2.0 * M * (C * C + 9 * C * C + 4 * C * C + 4 * C * C);       // branch2a, 2b, 2c and branch1
        prof->pending.push_back({7, fl, e0, e1, {(int)M, 4 * da.Cout, 18 * da.Cout, 7}, da.group, M * C * 2.0 + M * 4 * C * 2.0 + 18.0 * C * C * 2.0});
    }
}

// ================================================================================================
// element-wise helpers
// ================================================================================================
// fp32: NHWC4 (16 B per pixel); fp16: NHWC8 (16 B per pixel) — either way one 16-B store per pixel
template <typename T>
__global__ __launch_bounds__(256) void k_preprocess(const uint8_t* __restrict__ rgb, int B, int H, int W, int pad,
                                                    float mr, float mg, float mb, void* __restrict__ out)
{
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long total = (long)B * Hp * Wp;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % Wp);
        const int y = (int)((e / Wp) % Hp);
        const int b = (int)(e / ((long)Wp * Hp));
        float r = 0.f, g = 0.f, bl = 0.f;
        const int sy = y - pad, sx = x - pad;
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
            const uint8_t* p = rgb + (((long)b * H + sy) * W + sx) * 3;
            r = (float)p[0] - mr; g = (float)p[1] - mg; bl = (float)p[2] - mb;
        }
        if constexpr (sizeof(T) == 4) {
            reinterpret_cast<float4*>(out)[e] = make_float4(r, g, bl, 0.f);
        } else {
            f16x8 h;
            h[0] = (_Float16)r; h[1] = (_Float16)g; h[2] = (_Float16)bl;
            h[3] = h[4] = h[5] = h[6] = h[7] = (_Float16)0.f;
            reinterpret_cast<f16x8*>(out)[e] = h;
        }
    }
}

void preprocess_forward(hipStream_t s, const uint8_t* rgb, int B, int H, int W, int pad, const float mean[3], void* out, int dtype)
{
    const long total = (long)B * (H + 2 * pad) * (W + 2 * pad);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == MRCNN_F16) hipLaunchKernelGGL(k_preprocess<_Float16>, dim3(grid), dim3(256), 0, s, rgb, B, H, W, pad, mean[0], mean[1], mean[2], out);
    else hipLaunchKernelGGL(k_preprocess<float>, dim3(grid), dim3(256), 0, s, rgb, B, H, W, pad, mean[0], mean[1], mean[2], out);
    HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(256) void k_maxpool3x3s2(const T* __restrict__ in, int B, int H, int W, int C4,
                                                      T* __restrict__ out, int OH, int OW)
{
    const long total = (long)B * OH * OW * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        const int ox = (int)((e / C4) % OW);
        const int oy = (int)((e / ((long)C4 * OW)) % OH);
        const int b = (int)(e / ((long)C4 * OW * OH));
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < 3; ++dy) {
            const int y = 2 * oy + dy;
            if (y >= H) break;
            for (int dx = 0; dx < 3; ++dx) {
                const int x = 2 * ox + dx;
                if (x >= W) break;
                const float4 v = load4<T>(in + ((((long)b * H + y) * W + x) * C4 + c) * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        store4<T>(out + e * 4, m);
    }
}

void maxpool3x3s2_forward(hipStream_t s, const void* in, int B, int H, int W, int C, void* out, int OH, int OW, int dtype)
{
    MRCNN_REQUIRE(C % 4 == 0, MRCNN_ERR_SHAPE, "maxpool: C %% 4 != 0");
    const long total = (long)B * OH * OW * (C / 4);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == MRCNN_F16)
        hipLaunchKernelGGL(k_maxpool3x3s2<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)in, B, H, W, C / 4, (_Float16*)out, OH, OW);
    else hipLaunchKernelGGL(k_maxpool3x3s2<float>, dim3(grid), dim3(256), 0, s, (const float*)in, B, H, W, C / 4, (float*)out, OH, OW);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_softmax_pairs(const float2* __restrict__ logits, float2* __restrict__ probs, long n)
{
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float2 l = logits[e];
        const float m = fmaxf(l.x, l.y);
        const float e0 = expf(l.x - m), e1 = expf(l.y - m);
        const float inv = 1.0f / (e0 + e1);
        probs[e] = make_float2(e0 * inv, e1 * inv);
    }
}

void softmax_pairs_forward(hipStream_t s, const float* logits, float* probs, long n_pairs)
{
    const int grid = (int)((n_pairs + 255) / 256 < 8192 ? (n_pairs + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_softmax_pairs, dim3(grid), dim3(256), 0, s, (const float2*)logits, (float2*)probs, n_pairs);
    HIP_CHECK(hipGetLastError());
}

// one wave per row
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ logits, long ld, int nc, long n,
                                                      float* __restrict__ probs)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* l = logits + row * ld;
    float m = -INFINITY;
    for (int c = lane; c < nc; c += 64) m = fmaxf(m, l[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int c = lane; c < nc; c += 64) sum += expf(l[c] - m);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int c = lane; c < nc; c += 64) probs[row * nc + c] = expf(l[c] - m) * inv;
}

void softmax_rows_forward(hipStream_t s, const float* logits, long ld, int nc, long n, float* probs)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, logits, ld, nc, n, probs);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_copy_columns(const float* __restrict__ src, long ld, int c0, int ncols, long n,
                                                      float* __restrict__ dst)
{
    const long total = n * ncols;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / ncols;
        const int c = (int)(e - r * ncols);
        dst[e] = src[r * ld + c0 + c];
    }
}

void copy_columns_forward(hipStream_t s, const float* src, long ld, int c0, int ncols, long n, float* dst)
{
    if (n <= 0) return;
    const long total = n * ncols;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_copy_columns, dim3(grid), dim3(256), 0, s, src, ld, c0, ncols, n, dst);
    HIP_CHECK(hipGetLastError());
}

// TimeDistributedClassifierLayer.swift:65-86: argmax over all classes (ties → lowest index), score,
// the four deltas of the arg-max class.  One wave per ROI.
__global__ __launch_bounds__(256) void k_classifier_post(const float* __restrict__ probs, const float* __restrict__ bbox,
                                                         int nc, long n, float* __restrict__ out, long out_row_stride)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* p = probs + row * nc;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < nc; c += 64) {
        const float v = p[c];
        if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
    if (bi == 0x7fffffff) { bv = -INFINITY; }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    float* o = out + row * out_row_stride;
    if (lane < 4) o[lane] = bbox[row * nc * 4 + (long)bi * 4 + lane];
    else if (lane == 4) o[4] = (float)bi;
    else if (lane == 5) o[5] = bv;
}

void classifier_postprocess_forward(hipStream_t s, const float* probs, const float* bbox, int nc, long n, float* out,
                                    long out_row_stride)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_classifier_post, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, probs, bbox, nc, n, out, out_row_stride);
    HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, long n, int C, int HW, T* __restrict__ out)
{
    const long total = n * C * HW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int p = (int)((e / C) % HW);
        const long i = e / ((long)C * HW);
        out[e] = (T)in[(i * C + c) * HW + p];
    }
}
__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ in, long n, int C, int HW, float* __restrict__ out)
{
    const long total = n * C * HW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int p = (int)(e % HW);
        const int c = (int)((e / HW) % C);
        const long i = e / ((long)C * HW);
        out[e] = in[(i * HW + p) * C + c];
    }
}
void nchw_to_nhwc_forward(hipStream_t s, const float* in, long n, int C, int H, int W, void* out, int dtype)
{
    const long total = n * C * H * W;
    if (total <= 0) return;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == MRCNN_F16) hipLaunchKernelGGL(k_nchw_to_nhwc<_Float16>, dim3(grid), dim3(256), 0, s, in, n, C, H * W, (_Float16*)out);
    else hipLaunchKernelGGL(k_nchw_to_nhwc<float>, dim3(grid), dim3(256), 0, s, in, n, C, H * W, (float*)out);
    HIP_CHECK(hipGetLastError());
}
void nhwc_to_nchw_forward(hipStream_t s, const float* in, long n, int C, int H, int W, float* out)
{
    const long total = n * C * H * W;
    if (total <= 0) return;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(grid), dim3(256), 0, s, in, n, C, H * W, out);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_copy_rows(const float* __restrict__ src, long src_stride, long n, long len,
                                                   float* __restrict__ dst, long dst_stride)
{
    const long total = n * len;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / len, c = e - r * len;
        dst[r * dst_stride + c] = src[r * src_stride + c];
    }
}
void copy_rows_forward(hipStream_t s, const float* src, long src_stride, long n, long len, float* dst, long dst_stride)
{
    const long total = n * len;
    if (total <= 0) return;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(k_copy_rows, dim3(grid), dim3(256), 0, s, src, src_stride, n, len, dst, dst_stride);
    HIP_CHECK(hipGetLastError());
}

// ================================================================================================
// TimeDistributedMaskLayer
// ================================================================================================
// MultiArrayBatchProvider(removeZeros:true) (TimeDistributedClassifierLayer.swift:116-127): a row is
// kept iff every element is != 0.
template <typename T>
__global__ __launch_bounds__(256) void k_mask_row_flags(const T* __restrict__ pooled, long pooled_sB, long row_stride,
                                                        long row_len, int D, int32_t* __restrict__ flags)
{
    const int d = blockIdx.x, b = blockIdx.y;
    const T* r = pooled + (size_t)b * pooled_sB + (size_t)d * row_stride;
    int ok = 1;
    for (long e = threadIdx.x; e < row_len; e += 256) ok &= ((float)r[e] != 0.0f) ? 1 : 0;
    ok = __syncthreads_and(ok);
    if (threadIdx.x == 0) flags[(size_t)b * D + d] = ok;
}
__global__ void k_mask_row_compact(const int32_t* __restrict__ flags, int D, int32_t* __restrict__ mapping,
                                   int32_t* __restrict__ kept)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    int k = 0;
    for (int d = 0; d < D; ++d)
        if (flags[(size_t)b * D + d]) mapping[(size_t)b * D + k++] = d;
    kept[b] = k;
}

void mask_valid_rows_forward(hipStream_t s, const void* pooled, long pooled_sB, long row_stride, long row_len, int D,
                             int B, const MaskSelectWorkspace& ws, int dtype)
{
    if (D <= 0 || B <= 0) return;
    if (!pooled) { /* flags come from the ROIAlign kernel */ }
    else if (dtype == MRCNN_F16)
        hipLaunchKernelGGL(k_mask_row_flags<_Float16>, dim3(D, B), dim3(256), 0, s, (const _Float16*)pooled, pooled_sB, row_stride, row_len, D, ws.flags);
    else hipLaunchKernelGGL(k_mask_row_flags<float>, dim3(D, B), dim3(256), 0, s, (const float*)pooled, pooled_sB, row_stride, row_len, D, ws.flags);
    hipLaunchKernelGGL(k_mask_row_compact, dim3(B), dim3(64), 0, s, ws.flags, D, ws.mapping, ws.kept);
    HIP_CHECK(hipGetLastError());
}

// TimeDistributedMaskLayer.swift:58-89 with the Mask model's last layer (1×1 conv to numClasses +
// sigmoid, of which the reference keeps one channel) evaluated for the selected class only.
// Compact index i = blockIdx.y: row actual = mapping[i] is written with class detections[i][4]
// (:71 reads the compact index); rows i >= kept are zero padding (:87-89).
template <typename T>
__global__ __launch_bounds__(256) void k_mask_select(const T* __restrict__ feat, long feat_sB, int HW, int C,
                                                     const float* __restrict__ w, const float* __restrict__ bias, int nc,
                                                     const float* __restrict__ det, long det_sB, long det_stride, int D,
                                                     const int32_t* __restrict__ mapping, const int32_t* __restrict__ kept,
                                                     float* __restrict__ out, long out_sB, long out_stride)
{
    const int i = blockIdx.y, b = blockIdx.z;
    const int nk = kept[b];
    float* ob = out + (size_t)b * out_sB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (i >= nk) {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < out_stride; e += (long)gridDim.x * 256) ob[(size_t)i * out_stride + e] = 0.0f;
        return;
    }
    const int actual = mapping[(size_t)b * D + i];
    if (actual >= nk) return;                       // would be overwritten by the zero padding
    int cid = (int)det[(size_t)b * det_sB + (size_t)i * det_stride + 4];
    cid = cid < 0 ? 0 : (cid >= nc ? nc - 1 : cid);
    const float* wr = w + (size_t)cid * C;
    const T* f = feat + (size_t)b * feat_sB + (size_t)actual * HW * C;
    for (int p = blockIdx.x * 4 + wave; p < HW; p += gridDim.x * 4) {
        float sum = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 x = load4<T>(f + (size_t)p * C + c);
            const float4 y = *reinterpret_cast<const float4*>(wr + c);
            sum += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) ob[(size_t)actual * out_stride + p] = 1.0f / (1.0f + expf(-(sum + bias[cid])));
    }
    // the reference copies `stride` elements per row (:83); HW == stride for the 28×28 output
}

void mask_select_forward(hipStream_t s, const void* feat, long feat_sB, int HW, int C, const float* w,
                         const float* bias, int nc, const float* det, long det_sB, long det_stride, int D, int B,
                         const MaskSelectWorkspace& ws, float* out, long out_sB, long out_stride, int dtype)
{
    if (D <= 0 || B <= 0) return;
    MRCNN_REQUIRE(C % 4 == 0, MRCNN_ERR_SHAPE, "mask head: C %% 4 != 0");
    if (dtype == MRCNN_F16)
        hipLaunchKernelGGL(k_mask_select<_Float16>, dim3(49, D, B), dim3(256), 0, s, (const _Float16*)feat, feat_sB, HW, C, w, bias, nc, det,
                           det_sB, det_stride, D, ws.mapping, ws.kept, out, out_sB, out_stride);
    else
        hipLaunchKernelGGL(k_mask_select<float>, dim3(49, D, B), dim3(256), 0, s, (const float*)feat, feat_sB, HW, C, w, bias, nc, det, det_sB,
                           det_stride, D, ws.mapping, ws.kept, out, out_sB, out_stride);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// The fused form of the mask head's tail: the deconvolution leaves, per output pixel, `parts` partial dots with the selected
// class's 1x1 filter (conv_epilogue, ConvDesc::sel_partial) instead of its 256-channel fp32 output (642 MB per batch of 8
// that this layer used to read back).  Same control flow as k_mask_select — what TimeDistributedMaskLayer.swift:58-89 writes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_mask_select_classes(const float* __restrict__ det, long det_sB, long det_stride, int D, int nc,
                                                             const int32_t* __restrict__ mapping, const int32_t* __restrict__ kept,
                                                             int32_t* __restrict__ sel_cid)
{
    const int b = blockIdx.x;
    const int nk = kept[b];
    for (int r = threadIdx.x; r < D; r += blockDim.x) sel_cid[(size_t)b * D + r] = -1;
    __syncthreads();
    for (int i = threadIdx.x; i < nk; i += blockDim.x) {
        const int actual = mapping[(size_t)b * D + i];
        if (actual >= nk) continue;                     // would be overwritten by the zero padding
        int cid = (int)det[(size_t)b * det_sB + (size_t)i * det_stride + 4];      // the COMPACT index's class (:71)
        cid = cid < 0 ? 0 : (cid >= nc ? nc - 1 : cid);
        sel_cid[(size_t)b * D + actual] = cid;
    }
}

__global__ __launch_bounds__(256) void k_mask_select_partials(const float* __restrict__ partial, int parts, int HW,
                                                              const float* __restrict__ bias, int D,
                                                              const int32_t* __restrict__ sel_cid, const int32_t* __restrict__ kept,
                                                              float* __restrict__ out, long out_sB, long out_stride)
{
    const int r = blockIdx.y, b = blockIdx.z;
    const int nk = kept[b];
    float* orow = out + (size_t)b * out_sB + (size_t)r * out_stride;
    const int cid = sel_cid[(size_t)b * D + r];
    if (r >= nk) {                                       // zero padding (:87-89)
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < out_stride; e += (long)gridDim.x * 256) orow[e] = 0.0f;
        return;
    }
    if (cid < 0) return;                                 // a row the layer never writes
    const float* pr = partial + ((size_t)b * D + r) * HW * parts;
    const float bs = bias[cid];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float sum = pr[(size_t)p * parts];
        for (int h = 1; h < parts; ++h) sum += pr[(size_t)p * parts + h];
        orow[p] = 1.0f / (1.0f + expf(-(sum + bs)));
    }
}

void mask_select_classes(hipStream_t s, const float* det, long det_sB, long det_stride, int D, int B, int nc,
                         const MaskSelectWorkspace& ws, int32_t* sel_cid)
{
    if (D <= 0 || B <= 0) return;
    hipLaunchKernelGGL(k_mask_select_classes, dim3(B), dim3(128), 0, s, det, det_sB, det_stride, D, nc, ws.mapping, ws.kept, sel_cid);
    HIP_CHECK(hipGetLastError());
}

void mask_select_from_partials(hipStream_t s, const float* partial, int parts, int HW, const float* bias, int nc, const float* det,
                               long det_sB, long det_stride, int D, int B, const MaskSelectWorkspace& ws, float* out, long out_sB,
                               long out_stride)
{
    (void)nc; (void)det; (void)det_sB; (void)det_stride;
    if (D <= 0 || B <= 0) return;
    hipLaunchKernelGGL(k_mask_select_partials, dim3((HW + 255) / 256, D, B), dim3(256), 0, s, partial, parts, HW, bias, D, ws.sel_cid, ws.kept,
                       out, out_sB, out_stride);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_mask_select_full(const float* __restrict__ masks, long masks_sB, int HW, int nc,
                                                          const float* __restrict__ det, long det_sB, long det_stride,
                                                          int D, const int32_t* __restrict__ mapping,
                                                          const int32_t* __restrict__ kept, float* __restrict__ out,
                                                          long out_sB, long out_stride)
{
    const int i = blockIdx.x, b = blockIdx.y;
    const int nk = kept[b];
    float* ob = out + (size_t)b * out_sB;
    if (i >= nk) {
        for (long e = threadIdx.x; e < out_stride; e += 256) ob[(size_t)i * out_stride + e] = 0.0f;
        return;
    }
    const int actual = mapping[(size_t)b * D + i];
    if (actual >= nk) return;
    int cid = (int)det[(size_t)b * det_sB + (size_t)i * det_stride + 4];
    cid = cid < 0 ? 0 : (cid >= nc ? nc - 1 : cid);
    const float* src = masks + (size_t)b * masks_sB + ((size_t)actual * nc + cid) * HW;
    for (int e = threadIdx.x; e < HW; e += 256) ob[(size_t)actual * out_stride + e] = src[e];
}

void mask_select_from_full_forward(hipStream_t s, const float* masks, long masks_sB, int HW, int nc, const float* det,
                                   long det_sB, long det_stride, int D, int B, const MaskSelectWorkspace& ws, float* out,
                                   long out_sB, long out_stride)
{
    if (D <= 0 || B <= 0) return;
    hipLaunchKernelGGL(k_mask_select_full, dim3(D, B), dim3(256), 0, s, masks, masks_sB, HW, nc, det, det_sB, det_stride,
                       D, ws.mapping, ws.kept, out, out_sB, out_stride);
    HIP_CHECK(hipGetLastError());
}

}  // namespace mrcnn
