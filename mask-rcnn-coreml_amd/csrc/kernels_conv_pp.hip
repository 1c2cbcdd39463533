// kernels_conv_pp.hip — the deep-pipelined implicit-GEMM convolution of the trunk's large layers on gfx950: 256×256 block
// tiles, 8 waves in two groups that PING-PONG between "issue loads" and "issue MFMAs", persistent blocks (one per CU).
//
// Same contract as k_conv_mfma_glds (kernels_conv.hip; reference layers: the 3×3 / 1×1 convolutions of
// MaskRCNN.mlmodel / Mask.mlmodel, Sources/maskrcnn/Python/Conversion/task.py:69-104): NHWC activations, filters packed
// [cout][tap][cin] fp16, out[m][n] = Σ_k A[m][k]·Wt[n][k] accumulated in fp32 by v_mfma_f32_32x32x16_f16, fused
// scale/shift/residual/ReLU epilogue.  MODE 0: fp16 activations (MRCNN_F16).  MODE 2 / 3: fp32 activations split in
// registers into 2 / 3 fp16 parts, one MFMA pass per part (MRCNN_F32S / MRCNN_F32X3, see split_hi_lo in conv_device.h).
// Results are BIT-IDENTICAL to the 128-row kernels (same products, same order of accumulation, same epilogue
// arithmetic): which kernel a layer runs on depends on the batch size, per-image results must not
// (tests/test_gpu_conv_kernels.py).
//
// Why this structure (measured): a CU moves at most ≈27 B/clk from L2 (tools/probes/dma_probe.hip), the
// 128×128 fp16 tile needs 64 B/clk at full MFMA rate, LDS→VGPR reads and DMA writes share the LDS, and the matrix cores
// pull the clock down to ≈1.45 GHz; the block-level fixed cost (prologue latency, LDS-staged epilogue) is exposed once a
// block owns its CU.  Hence:
//   * block tile 256×256, one 128-B run of K per operand row and step (64 fp16 / 32 fp32 channels), wave tile 128×64
//     (2×4 waves): half the L2→LDS bytes and half the LDS→VGPR reads per MFMA of the 128×128 kernel;
//   * a K step is consumed in four PHASES (one 32-row slab of the wave tile × its 64 columns): 8 MFMAs (fp16) or
//     8 / 12 MFMAs + the hi/(mid/)lo split of the slab's activations (split modes); the filter fragments of the whole
//     K step are read once (phase 0) and stay in registers;
//   * the two wave groups (waves 0-3 / 4-7 = one wave of each per SIMD) run ONE BARRIER APART: while one group issues
//     the MFMAs of a phase (s_setprio 1), the other issues its LDS fragment reads and its share of the global→LDS DMA
//     (buffer_load_dwordx4 … lds, XOR-swizzled source chunks, out-of-range offsets = zeros for padding taps) for a K step 1½–2 steps ahead;
//     vmcnt is COUNTED (one `s_waitcnt vmcnt(4 | 3)` per K step, never 0 in steady state);
//   * the epilogue goes straight from the accumulators to HBM (transposed result tiles: a lane owns runs of four
//     consecutive channels of one pixel; fp16: one v_permlane32_swap per 16 B) — no LDS staging, no barrier;
//   * PERSISTENT blocks: a block walks its tiles; the DMA prologue of tile i+1 is issued BEFORE the epilogue of tile i
//     (the epilogue needs no LDS), so neither the prologue's memory latency nor the store drain is exposed.
//
// LDS hazards are excluded by construction.  With "slot" = the interval between two consecutive barrier rendezvous,
// group 0 runs L(ph) in slot 2·ph and M(ph) in slot 2·ph+1 of a K step, group 1 one slot later:
//   RAW  a DMA is visible to a ds_read only after the issuing wave's covering vmcnt AND a barrier the reader passed
//        afterwards: every wave waits for K step kt+1 at the end of its LAST L phase of step kt (group 1: the slot right
//        before group 0's first read of kt+1);
//   WAR  the last reader of slab ph of step kt is group 1, whose reads are issued in slot 2ph+1 and retired by its
//        lgkmcnt(0) at the top of slot 2ph+2: the slab's LDS rows may be overwritten by DMAs issued from slot 2ph+3 on.
//        With two K-step buffers slab ph of step kt+2 is issued in L(ph+2) of step kt (slabs 2, 3 in L(0), L(1) of step
//        kt+1) — slot 2ph+4 at the earliest; the filter tile is read only in phase 0 (retired in slot 2): its pieces
//        travel with slabs 2/3 (→ the other buffer) and 0/1 (→ this buffer, from L(2) on).
#include <type_traits>

#include "conv_device.h"

namespace mrcnn {

#define PP_GLDS_V(SRC, DST)                                                                                    \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(SRC), "s"(DST) : "memory", "m0");
// buffer form: 128-bit resource (scalar) + 32-bit lane offset + scalar offset; lanes whose offset lies beyond the resource's
// num_records deposit ZEROS in LDS (hardware range check) — padding taps and rows beyond M need no zero page
#define PP_BLDS(VOFF, SRD, SOFF, DST)                                                                          \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(VOFF), "s"(SRD), "s"(SOFF), "s"(DST) : "memory", "m0");
#define PP_BLDS0(VOFF, SRD, DST)                                                                               \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(VOFF), "s"(SRD), "s"(DST) : "memory", "m0");
#define PP_DSR(DSTV, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DSTV) : "v"(ADDR), "n"(OFF));
#define PP_BARRIER asm volatile("s_barrier" ::: "memory");
#define PP_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#define PP_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0);
#define PP_U4(X_) __builtin_bit_cast(uint4, X_)

// Source of one staged activation row P for the tap (KH_, KW_) and byte offset KOFF_ (tap + channel step, wave-uniform):
// the 32-bit byte offset of the row's pixel run in the activation resource, or an offset beyond it (→ zeros) when the tap
// falls outside the image (zero padding) / the row is beyond M.  Rebuilt at every issue from 2 registers per row (offset
// of tap (0, 0), packed 16-bit (ih0, iw0)): a handful of VALU in a load phase instead of live registers per row — the
// kernel's budget is 256 registers with a 128-register accumulator.
#define PP_OFF_A(P, KH_, KW_, KOFF_)                                                                           \
    ({                                                                                                         \
        const int ih = (int)(short)(ihw[P] & 0xffff) + (KH_), iw = (ihw[P] >> 16) + (KW_);                     \
        const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;                          \
        ok ? (dbg_hotsrc ? (unsigned)(kq * 16) : obase[P] + (unsigned)(KOFF_)) : OOB;                          \
    })

// ----------------------------------------------------------------------------------------------------------------
// Epilogue straight from the accumulators (no LDS round trip, no barrier).  The MFMAs of this kernel take the FILTER
// fragment as their first operand: the 32×32 result tile is then held transposed — lane (l31, kk) owns output pixel
// l31 of the slab and channels 8q + 4kk + r (q, r = 0..3) of the column tile — so every lane has runs of four
// consecutive channels of one pixel.  Per run: fused scale/shift (+ residual) + ReLU in fp32 exactly as conv_epilogue
// does.  fp32 tensors: one 16-B store per run.  fp16 tensors: one rounding, then per pair of runs (q = 2p, 2p+1) a
// v_permlane32_swap between the half-waves glues the pieces into 16 contiguous bytes per lane (cdna guide T21): lane
// (l31, 0) stores channels 16p..16p+7 of its pixel, lane (l31, 1) channels 16p+8..16p+15 — a wave writes whole
// 128-B lines (its 64 channels of a pixel) in four back-to-back stores.
// ----------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool pp_store_tile(const ConvArgs& a, f32x16 (&acc)[4][2], const float* tab, int m0, int n0, int wrow, int wcol, int lane)
{
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const T* const res = static_cast<const T*>(a.res);
    T* const out = static_cast<T*>(a.out);
    const bool dense_out = a.out_sB == (long)ohw * a.out_sP;
    const bool dense_res = a.res_sB == (long)ohw * a.res_sW && a.res_shift == 0;
    const bool relu = a.act == ACT_RELU;
    bool out_of_range = false;
    // slab outermost, channel runs innermost: the four 32-B pieces of a pixel's 128-B line leave in back-to-back stores
    // (measured: with the channel runs outermost — scale / shift fetched once per run — the stores of a line are spread
    // over the epilogue and the tile's fixed cost grows by 25 %: the L2 no longer merges them before they reach HBM)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wrow * 128 + i * 32 + l31;
        const bool ok_mi = m < a.M;
        long o_rowi = (long)m * a.out_sP, r_rowi = (long)m * a.res_sW;
        if (!dense_out || (res && !dense_res)) {
            const int mm = ok_mi ? m : 0;
            const int b = mm / ohw, pix = mm - b * ohw;
            o_rowi = (long)b * a.out_sB + (long)pix * a.out_sP;
            if (res) {
                if (a.res_shift) {
                    const int oh = pix / a.OW, ow = pix - oh * a.OW;
                    r_rowi = (long)b * a.res_sB + (long)(oh >> a.res_shift) * a.res_sH + (long)(ow >> a.res_shift) * a.res_sW;
                } else r_rowi = (long)b * a.res_sB + (long)pix * a.res_sW;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = n0 + wcol * 64 + j * 32;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int ca = nb + 16 * p + 4 * kk, cb = ca + 8;
                // scale / shift come from the tile's LDS table (DMA'd at the start of the tile): fetched from global memory
                // here, every fetch would queue behind the previous iteration's stores (`out` may alias them as far as the
                // compiler knows; loads and stores share vmcnt) — 16 serialized store-acknowledge + load round trips per
                // tile, measured as 110 of the 300 us fixed cost of the RPN 3x3 layer
                const int cl = wcol * 64 + j * 32 + 16 * p + 4 * kk;
                float4 sa = make_float4(1.f, 1.f, 1.f, 1.f), sb_ = sa, ha = make_float4(0.f, 0.f, 0.f, 0.f), hb = ha;
                if (a.scale) { sa = *reinterpret_cast<const float4*>(tab + cl); sb_ = *reinterpret_cast<const float4*>(tab + cl + 8); }
                if (a.shift) { ha = *reinterpret_cast<const float4*>(tab + 256 + cl); hb = *reinterpret_cast<const float4*>(tab + 256 + cl + 8); }
                float4 va = make_float4(acc[i][j][8 * p + 0], acc[i][j][8 * p + 1], acc[i][j][8 * p + 2], acc[i][j][8 * p + 3]);
                float4 vb = make_float4(acc[i][j][8 * p + 4], acc[i][j][8 * p + 5], acc[i][j][8 * p + 6], acc[i][j][8 * p + 7]);
                va.x = va.x * sa.x + ha.x; va.y = va.y * sa.y + ha.y; va.z = va.z * sa.z + ha.z; va.w = va.w * sa.w + ha.w;
                vb.x = vb.x * sb_.x + hb.x; vb.y = vb.y * sb_.y + hb.y; vb.z = vb.z * sb_.z + hb.z; vb.w = vb.w * sb_.w + hb.w;
                const bool ok_a = ok_mi && ca < a.ncols, ok_b = ok_mi && cb < a.ncols;
                if (res) {
                    if (ok_a) { const float4 r = load4<T>(res + r_rowi + ca); va.x += r.x; va.y += r.y; va.z += r.z; va.w += r.w; }
                    if (ok_b) { const float4 r = load4<T>(res + r_rowi + cb); vb.x += r.x; vb.y += r.y; vb.z += r.z; vb.w += r.w; }
                }
                if (relu) {
                    va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                    vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
                }
                // fp16-range watchdog (|v| >= 65504, inf or NaN)
                if (ok_a) out_of_range = out_of_range || !(fabsf(va.x) < 65504.0f) || !(fabsf(va.y) < 65504.0f) || !(fabsf(va.z) < 65504.0f) || !(fabsf(va.w) < 65504.0f);
                if (ok_b) out_of_range = out_of_range || !(fabsf(vb.x) < 65504.0f) || !(fabsf(vb.y) < 65504.0f) || !(fabsf(vb.z) < 65504.0f) || !(fabsf(vb.w) < 65504.0f);
                if (a.dbg & 32) {                      // measurement only: the epilogue's arithmetic without its stores
                    asm volatile("" ::"v"(va.x), "v"(va.y), "v"(va.z), "v"(va.w), "v"(vb.x), "v"(vb.y), "v"(vb.z), "v"(vb.w));
                } else if constexpr (sizeof(T) == 4) {
                    if (ok_a) *reinterpret_cast<float4*>(out + o_rowi + ca) = va;
                    if (ok_b) *reinterpret_cast<float4*>(out + o_rowi + cb) = vb;
                } else {
                    f16x4 ha4, hb4;
                    ha4[0] = (_Float16)va.x; ha4[1] = (_Float16)va.y; ha4[2] = (_Float16)va.z; ha4[3] = (_Float16)va.w;
                    hb4[0] = (_Float16)vb.x; hb4[1] = (_Float16)vb.y; hb4[2] = (_Float16)vb.z; hb4[3] = (_Float16)vb.w;
                    const uint2 pa = __builtin_bit_cast(uint2, ha4), pb = __builtin_bit_cast(uint2, hb4);
                    // upper half-wave's q = 2p runs <-> lower half-wave's q = 2p+1 runs
                    const auto sx = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
                    const int n_store = nb + 16 * p + 8 * kk;
                    if (ok_mi && n_store < a.ncols)
                        *reinterpret_cast<uint4*>(out + o_rowi + n_store) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
        }
    }
    return out_of_range;      // the caller raises the watchdog flag once, at the end of the kernel (keeps the VMEM count of a tile exact)
}

// ================================================================================================================
template <int MODE>
__global__ __launch_bounds__(512) void k_conv_pp(const ConvArgs a)
{
    constexpr bool SPLIT = MODE != 0;
    using T = std::conditional_t<SPLIT, float, _Float16>;         // activation element
    using TW = _Float16;                                          // filter element
    constexpr int BM = 256, BN = 256;
    constexpr int BK = SPLIT ? 32 : 64;                           // channels per K step = one 128-B run of an activation row
    constexpr int ROWB = 128, BROWB = SPLIT ? 64 : 128;           // LDS row bytes of the activation / filter tile
    constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * BROWB, B_BASE = 2 * A_STAGE;
    constexpr int BJ = 32 * BROWB;                                // LDS bytes between the two column tiles of a wave
    constexpr int STEADY = SPLIT ? 3 : 4;                         // DMAs younger than K step kt+1 at the end of step kt
    constexpr int TILE_STORES = SPLIT ? 32 : 16;                  // 16-B store instructions of pp_store_tile per wave on an interior tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_STAGE + B_STAGE)];     // 128 KB (fp16) / 96 KB (split)
    __shared__ __attribute__((aligned(16))) float s_tab[2][2][BN];      // scale | shift of this tile's and the previous tile's columns
    const char* const in = static_cast<const char*>(a.in);
    const TW* const wgt = static_cast<const TW*>(a.wgt);
    // The DMAs address both operands through buffer resources (raw, stride 0): 32-bit lane offsets, and a lane whose offset
    // lies beyond the resource deposits zeros in LDS (no zero page, no memory access for padding taps).  Measured against
    // 64-bit lane addresses (global_load_lds) in this kernel: +5…8 % on the large fp16 layers (profiles/r02_conv_ab_dmaform.txt).
    // The activation resource is re-based at every tile on the first image the tile touches: offsets stay small whatever the batch.
    constexpr unsigned OOB = 0xffffff00u;
    typedef unsigned srd_t __attribute__((ext_vector_type(4)));
    srd_t srdA, srdB;
    {
        const unsigned long long wa = (unsigned long long)(uintptr_t)wgt;
        srdB[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
        srdB[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
        srdB[2] = 0xffffffffu;
        srdB[3] = 0x00020000u;
        srdA[3] = 0x00020000u;
    }

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;                  // wave row (= ping-pong group) / wave column
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned tab0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&s_tab[0][0][0]);

    // ---- loop-invariant lane geometry -------------------------------------------------------------------------------
    // staging: slab p (0..3) of the A tile = rows {wr·128 + p·32 + 0..31 : wr = 0, 1}; one DMA per thread and slab: wave w
    // stages rows (w>>2)·128 + p·32 + (w&3)·8 + 0..7 (1 KB, lane-linear); 16-B chunk c of row r sits at c ^ ((r>>1)&7)
    const int srow = (wave & 3) * 8 + (lane >> 3);
    const int kq = (lane & 7) ^ ((srow >> 1) & 7);
    const unsigned dA = lds0 + (wr * 128 + (wave & 3) * 8) * ROWB;      // + buf·A_STAGE + p·4096
    // filter tile, fp16 tensors: four 64-row pieces of 128-B rows (row = p·64 + wave·8 + lane>>3, chunk as above);
    //              split modes: two 128-row pieces of 64-B rows (row = q·128 + wave·16 + lane>>2, chunk c at c ^ ((r>>2)&3))
    const unsigned vb = SPLIT ? (unsigned)(((size_t)(wave * 16 + (lane >> 2)) * a.Ktot + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * sizeof(TW))
                              : (unsigned)(((size_t)(wave * 8 + (lane >> 3)) * a.Ktot + kq * 8) * sizeof(TW));
    const unsigned bstr = (unsigned)((size_t)(SPLIT ? 128 : 64) * a.Ktot * sizeof(TW));     // byte distance of the pieces
    const unsigned dB = lds0 + B_BASE + wave * 1024;                    // + buf·B_STAGE + piece·8192
    // fragment reads: lane (l31, kk) of K group g reads 16 B (fp16: chunk 2g+kk) or 32 B (fp32: chunks 4g+2kk, +1) of row l31
    const int l31 = lane & 31, kk = lane >> 5, swz = (l31 >> 1) & 7;
    const unsigned ra_base = lds0 + (wr * 128 + l31) * ROWB;
    const unsigned ra0 = ra_base + (((SPLIT ? 0 + 2 * kk : 0 + kk) ^ swz) << 4), ra1 = ra_base + (((SPLIT ? 1 + 2 * kk : 2 + kk) ^ swz) << 4);
    const unsigned ra2 = ra_base + (((SPLIT ? 4 + 2 * kk : 4 + kk) ^ swz) << 4), ra3 = ra_base + (((SPLIT ? 5 + 2 * kk : 6 + kk) ^ swz) << 4);
    const unsigned rb_base = lds0 + B_BASE + (wc * 64 + l31) * BROWB;
    const int swzb = SPLIT ? (l31 >> 2) & 3 : swz;
    const unsigned rb0 = rb_base + (((0 + kk) ^ swzb) << 4), rb1 = rb_base + (((2 + kk) ^ swzb) << 4);
    const unsigned rb2 = rb_base + (((4 + kk) ^ swz) << 4), rb3 = rb_base + (((6 + kk) ^ swz) << 4);      // fp16 tensors only

    const int ohw = a.OH * a.OW;
    const int cin_tiles = a.Cin / BK;
    const int KT = a.KH * a.KW * cin_tiles;
    const unsigned tapW = (unsigned)(a.in_sW * (long)sizeof(T)), tapH = (unsigned)(a.in_sH * (long)sizeof(T));
    // measurement-only ablations (a.dbg = 0 in production): 1 no s_setprio, 2 no group stagger, 4 no DMA in the main
    // loop, 8 no fragment reads, 16 no MFMAs, 32 epilogue without its stores, 64 no epilogue
    const bool dbg_hotsrc = a.dbg & 256;         // 256: every activation DMA reads the first bytes of the tensor (DMA issue + LDS writes, L1-hot source)
    const bool dbg_noprio = a.dbg & 1, dbg_nostagger = a.dbg & 2, dbg_nodma = a.dbg & 4, dbg_nords = a.dbg & 8, dbg_nomma = a.dbg & 16;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    f16x8 fa0, fa1, fa2, fa3;                                      // activation fragments of the current phase (16 B each)
    f16x8 fb00, fb01, fb10, fb11, fb20, fb21, fb30, fb31;          // filter fragments of the current K step (K group, column tile)

    // ---- persistent walk over the tiles: virtual block v = blockIdx.x + step·gridDim.x through the XCD-aware bijective
    //      map (the N tiles of one M tile adjacent, contiguous runs per XCD; gridDim.x is a multiple of 8 or = #tiles)
    const int nblocks = a.tiles_m * a.tiles_n;
    const int q8 = nblocks >> 3, r8 = nblocks & 7;
    int pm0 = 0, pn0 = 0;
    bool have_prev = false, range_trip = false;
    int tb = 0;                  // table buffer of the tile being computed (the previous tile's epilogue reads tb ^ 1)
    for (int v = blockIdx.x; v < nblocks; v += gridDim.x) {
        const int xcd = v & 7, local = v >> 3;
        const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
        const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
        const int m0 = mt * BM, n0 = nt * BN;

        // scale / shift of the tile's 256 columns → LDS (one 1-KB DMA each, waves 0 and 1): the oldest vector-memory
        // operations of the tile, so every counted wait below covers them; read by the epilogue one tile later, behind
        // this tile's barriers.  Two buffers: a slower wave may still be in the previous tile's epilogue.
        if (wave == 0 && a.scale) { const char* src_ = reinterpret_cast<const char*>(a.scale + n0) + lane * 16; PP_GLDS_V(src_, tab0 + tb * 2048); }
        if (wave == 1 && a.shift) { const char* src_ = reinterpret_cast<const char*>(a.shift + n0) + lane * 16; PP_GLDS_V(src_, tab0 + tb * 2048 + 1024); }
        const int b0 = m0 / ohw;
        {
            const unsigned long long ia = (unsigned long long)(uintptr_t)(in + (long)b0 * a.in_sB * (long)sizeof(T));
            const unsigned long long rest = (unsigned long long)(a.M / ohw - b0) * (unsigned long long)a.in_sB * sizeof(T);
            srdA[0] = __builtin_amdgcn_readfirstlane((unsigned)ia);
            srdA[1] = __builtin_amdgcn_readfirstlane((unsigned)(ia >> 32) & 0xffffu);
            srdA[2] = __builtin_amdgcn_readfirstlane((unsigned)(rest < OOB ? rest : OOB));
        }
        unsigned obase[4];       // byte offset (from the resource base, mod 2^32: it may lie before it) of tap (0, 0), channel chunk kq, of the row this thread stages in slab p
        int ihw[4];              // (ih0 & 0xffff) | (iw0 << 16): input coordinates of tap (0, 0)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + wr * 128 + p * 32 + srow;
            const bool ok = m < a.M;
            const int mm = ok ? m : 0;
            const int b = mm / ohw, rem = mm - b * ohw;
            const int oh = rem / a.OW, ow = rem - oh * a.OW;
            const int ih0 = oh * a.stride - a.padH, iw0 = ow * a.stride - a.padW;
            ihw[p] = ((ok ? ih0 : -32768) & 0xffff) | (iw0 << 16);      // a row beyond M never passes the bounds test
            obase[p] = (unsigned)(((long)(b - b0) * a.in_sB + (long)ih0 * a.in_sH + (long)iw0 * a.in_sW) * (long)sizeof(T) + kq * 16);
        }
        // two issue streams (wave-uniform scalar state): slabs 0, 1 (+ their filter pieces) run one K step ahead of slabs 2, 3
        int ct01 = 0, kh01 = 0, kw01 = 0, ct23 = 0, kh23 = 0, kw23 = 0;
        unsigned ko01 = 0, ko23 = 0;               // byte offset of the stream's current K step from the tap-(0,0) pixel
        unsigned so01 = (unsigned)((size_t)n0 * a.Ktot * sizeof(TW)), so23 = so01;      // byte offset of the stream's K step in the filter
#define PP_ADV(CT_, KH_, KW_, KO_)                                                                             \
    {                                                                                                          \
        KO_ += ROWB;                                                                                           \
        if (++CT_ == cin_tiles) {                                                                              \
            CT_ = 0;                                                                                           \
            if (++KW_ == a.KW) { KW_ = 0; ++KH_; }                                                             \
            KO_ = KH_ * tapH + KW_ * tapW;                                                                     \
        }                                                                                                      \
    }
#define PP_ISSUE_A01(P, BUF_) { const unsigned off_ = PP_OFF_A(P, kh01, kw01, ko01); PP_BLDS0(off_, srdA, dA + (BUF_) * A_STAGE + (P) * 4096) }
#define PP_ISSUE_A23(P, BUF_) { const unsigned off_ = PP_OFF_A(P, kh23, kw23, ko23); PP_BLDS0(off_, srdA, dA + (BUF_) * A_STAGE + (P) * 4096) }
#define PP_ISSUE_B01(P, BUF_) PP_BLDS(vb + (P) * bstr, srdB, so01, dB + (BUF_) * B_STAGE + (P) * 8192)
#define PP_ISSUE_B23(P, BUF_) PP_BLDS(vb + (P) * bstr, srdB, so23, dB + (BUF_) * B_STAGE + (P) * 8192)
        // the DMA of the four load phases: stream 23 → K step kt+1 (other buffer), stream 01 → K step kt+2 (this buffer)
#define PP_L0_ISSUE(BUFN) { PP_ISSUE_A23(2, BUFN) if constexpr (!SPLIT) PP_ISSUE_B23(2, BUFN) }
#define PP_L1_ISSUE(BUFN)                                                                                      \
    {                                                                                                          \
        PP_ISSUE_A23(3, BUFN)                                                                                  \
        if constexpr (!SPLIT) { PP_ISSUE_B23(3, BUFN) } else { PP_ISSUE_B23(1, BUFN) }                         \
        so23 += BK * sizeof(TW);                                                                               \
        PP_ADV(ct23, kh23, kw23, ko23)                                                                         \
    }
#define PP_L2_ISSUE(BUF_) { PP_ISSUE_A01(0, BUF_) if constexpr (!SPLIT) PP_ISSUE_B01(0, BUF_) }
#define PP_L3_ISSUE(BUF_)                                                                                      \
    {                                                                                                          \
        PP_ISSUE_A01(1, BUF_)                                                                                  \
        if constexpr (!SPLIT) { PP_ISSUE_B01(1, BUF_) } else { PP_ISSUE_B01(0, BUF_) }                         \
        so01 += BK * sizeof(TW);                                                                               \
        PP_ADV(ct01, kh01, kw01, ko01)                                                                         \
    }

        // ---- prologue: K step 0 completely, of K step 1 the part the steady state has in flight at a step boundary.
        //      LDS is free: every wave is past the re-join barrier of the previous tile.
        PP_L2_ISSUE(0) PP_L3_ISSUE(0) PP_L0_ISSUE(0) PP_L1_ISSUE(0)
        if (KT > 1) { PP_L2_ISSUE(1) PP_L3_ISSUE(1) }
        // ---- epilogue of the PREVIOUS tile under the flight of those DMAs (registers → HBM, no LDS)
        if (have_prev) {
            if (!(a.dbg & 64)) range_trip |= pp_store_tile<T>(a, acc, &s_tab[tb ^ 1][0][0], pm0, pn0, wr, wc, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
            // K step 0 must have landed; the epilogue's stores are YOUNGER than it and share the counter (gfx9: loads and
            // stores retire in issue order).  On an interior tile every wave issued exactly TILE_STORES stores, so a counted
            // wait leaves them (and the K-step-1 DMAs) in flight — the store drain of 256 CUs bursting together is not
            // waited for.  Edge tiles (some stores skipped: the count is not exact) drain everything.
            const bool interior = pm0 + BM <= a.M && pn0 + BN <= a.ncols;
            if (interior && KT > 1) { PP_VMCNT(TILE_STORES + STEADY) }
            else if (interior) { PP_VMCNT(TILE_STORES) }
            else { PP_VMCNT(0) }
        } else if (KT > 1) {
            PP_VMCNT(STEADY)
        } else {
            PP_VMCNT(0)
        }
        PP_BARRIER
        if (wr == 1 && !dbg_nostagger) PP_BARRIER          // group 1 runs one barrier behind group 0 from here on

#define PP_RD_A(PH, BUF)                                                                                       \
    PP_DSR(fa0, ra0, (BUF) * A_STAGE + (PH) * 4096) PP_DSR(fa1, ra1, (BUF) * A_STAGE + (PH) * 4096)            \
    PP_DSR(fa2, ra2, (BUF) * A_STAGE + (PH) * 4096) PP_DSR(fa3, ra3, (BUF) * A_STAGE + (PH) * 4096)
#define PP_RD_B(BUF)                                                                                           \
    PP_DSR(fb00, rb0, (BUF) * B_STAGE) PP_DSR(fb01, rb0, (BUF) * B_STAGE + BJ)                                 \
    PP_DSR(fb10, rb1, (BUF) * B_STAGE) PP_DSR(fb11, rb1, (BUF) * B_STAGE + BJ)                                 \
    if constexpr (!SPLIT) {                                                                                    \
        PP_DSR(fb20, rb2, (BUF) * B_STAGE) PP_DSR(fb21, rb2, (BUF) * B_STAGE + BJ)                             \
        PP_DSR(fb30, rb3, (BUF) * B_STAGE) PP_DSR(fb31, rb3, (BUF) * B_STAGE + BJ)                             \
    }
// the wait names every fragment as read-write: nothing that consumes one can be scheduled above it
#define PP_WAIT_A asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa0), "+v"(fa1), "+v"(fa2), "+v"(fa3)::"memory");
#define PP_WAIT_AB                                                                                             \
    if constexpr (!SPLIT) {                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                    \
                     : "+v"(fa0), "+v"(fa1), "+v"(fa2), "+v"(fa3), "+v"(fb00), "+v"(fb01), "+v"(fb10), "+v"(fb11), "+v"(fb20),     \
                       "+v"(fb21), "+v"(fb30), "+v"(fb31)::"memory");                                          \
    } else {                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                    \
                     : "+v"(fa0), "+v"(fa1), "+v"(fa2), "+v"(fa3), "+v"(fb00), "+v"(fb01), "+v"(fb10), "+v"(fb11)::"memory");     \
    }
// fp16: 4 K groups × 2 column tiles.  split: 2 K groups × (hi, [mid,] lo) × 2 column tiles, parts in the order of the
// 128-row kernel (every product tile is added to the same accumulator in the same sequence → identical bits).
// (Tried and measured worse: converting the slab in the LOAD phase, under the partner's MFMAs — the load phase, already
// holding the blocking DMA issue, becomes the longer one: 0.87× instead of 0.98× of the 128-row kernel in f32x3.
// Tried and measured neutral: only the FIRST K group's split moved into the load segment, so that the math segment
// opens with MFMAs — 3 262 vs 3 278 µs: the split kernels run at the board's power cap)
#define PP_MATH(PH)                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (!dbg_noprio) __builtin_amdgcn_s_setprio(1);                                                            \
    if (!dbg_nomma) {                                                                                          \
        if constexpr (!SPLIT) {                                                                                \
            PP_MFMA(fb00, fa0, acc[PH][0]) PP_MFMA(fb01, fa0, acc[PH][1]) PP_MFMA(fb10, fa1, acc[PH][0]) PP_MFMA(fb11, fa1, acc[PH][1]) \
            PP_MFMA(fb20, fa2, acc[PH][0]) PP_MFMA(fb21, fa2, acc[PH][1]) PP_MFMA(fb30, fa3, acc[PH][0]) PP_MFMA(fb31, fa3, acc[PH][1]) \
        } else {                                                                                               \
            f16x8 hi, mid, lo;                                                                                 \
            if constexpr (MODE == 3) split_hi_mid_lo(PP_U4(fa0), PP_U4(fa1), hi, mid, lo); else split_hi_lo(PP_U4(fa0), PP_U4(fa1), hi, lo); \
            PP_MFMA(fb00, hi, acc[PH][0]) PP_MFMA(fb01, hi, acc[PH][1])                                        \
            if constexpr (MODE == 3) { PP_MFMA(fb00, mid, acc[PH][0]) PP_MFMA(fb01, mid, acc[PH][1]) }         \
            PP_MFMA(fb00, lo, acc[PH][0]) PP_MFMA(fb01, lo, acc[PH][1])                                        \
            if constexpr (MODE == 3) split_hi_mid_lo(PP_U4(fa2), PP_U4(fa3), hi, mid, lo); else split_hi_lo(PP_U4(fa2), PP_U4(fa3), hi, lo); \
            PP_MFMA(fb10, hi, acc[PH][0]) PP_MFMA(fb11, hi, acc[PH][1])                                        \
            if constexpr (MODE == 3) { PP_MFMA(fb10, mid, acc[PH][0]) PP_MFMA(fb11, mid, acc[PH][1]) }         \
            PP_MFMA(fb10, lo, acc[PH][0]) PP_MFMA(fb11, lo, acc[PH][1])                                        \
        }                                                                                                      \
    }                                                                                                          \
    __builtin_amdgcn_s_setprio(0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);

        // One K step on buffer BUF (compile-time): four L/M phase pairs.  has1 / has2: K steps kt+1 / kt+2 exist.
#define PP_KSTEP(KTV, BUF)                                                                                     \
    {                                                                                                          \
        const bool has1 = (KTV) + 1 < KT && !dbg_nodma, has2 = (KTV) + 2 < KT && !dbg_nodma;                   \
        /* phase 0: slab 0 + the whole filter tile; DMA: slab 2 (+ filter piece) of step kt+1 */               \
        if (!dbg_nords) { PP_RD_A(0, BUF) PP_RD_B(BUF) }                                                       \
        if (has1) PP_L0_ISSUE((BUF) ^ 1)                                                                       \
        PP_BARRIER PP_WAIT_AB PP_MATH(0) PP_BARRIER                                                            \
        /* phase 1: DMA: slab 3 (+ filter piece) of step kt+1 */                                               \
        if (!dbg_nords) { PP_RD_A(1, BUF) }                                                                    \
        if (has1) PP_L1_ISSUE((BUF) ^ 1)                                                                       \
        PP_BARRIER PP_WAIT_A PP_MATH(1) PP_BARRIER                                                             \
        /* phase 2: DMA: slab 0 (+ filter piece) of step kt+2, over this step's own buffer */                  \
        if (!dbg_nords) { PP_RD_A(2, BUF) }                                                                    \
        if (has2) PP_L2_ISSUE(BUF)                                                                             \
        PP_BARRIER PP_WAIT_A PP_MATH(2) PP_BARRIER                                                             \
        /* phase 3: DMA: slab 1 (+ filter piece) of step kt+2; then step kt+1 must have landed (STEADY younger DMAs stay in flight) */ \
        if (!dbg_nords) { PP_RD_A(3, BUF) }                                                                    \
        if (has2) {                                                                                            \
            PP_L3_ISSUE(BUF)                                                                                   \
            PP_VMCNT(STEADY)                                                                                   \
        } else {                                                                                               \
            PP_VMCNT(0)                                                                                        \
        }                                                                                                      \
        PP_BARRIER PP_WAIT_A PP_MATH(3) PP_BARRIER                                                             \
    }
        for (int kt = 0; kt < KT; kt += 2) {
            PP_KSTEP(kt, 0)
            if (kt + 1 < KT) PP_KSTEP(kt + 1, 1)
        }
        if (wr == 0 && !dbg_nostagger) PP_BARRIER          // re-join: every wave's last fragment read has retired behind this rendezvous
        pm0 = m0; pn0 = n0; have_prev = true; tb ^= 1;
    }
    if (have_prev && !(a.dbg & 64)) range_trip |= pp_store_tile<T>(a, acc, &s_tab[tb ^ 1][0][0], pm0, pn0, wr, wc, lane);
    if (a.range_flag && range_trip) atomicOr(a.range_flag, 1);
#undef PP_KSTEP
#undef PP_MATH
#undef PP_WAIT_AB
#undef PP_WAIT_A
#undef PP_RD_B
#undef PP_RD_A
#undef PP_L3_ISSUE
#undef PP_L2_ISSUE
#undef PP_L1_ISSUE
#undef PP_L0_ISSUE
#undef PP_ISSUE_B23
#undef PP_ISSUE_B01
#undef PP_ISSUE_A23
#undef PP_ISSUE_A01
#undef PP_ADV
}

// mode: 0 fp16 tensors, 2 / 3 fp32 tensors split in 2 / 3 fp16 parts.  One persistent block per CU.
void conv_pp_launch(hipStream_t s, const ConvArgs& a, int mode)
{
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t p;
        HIP_CHECK(hipGetDeviceProperties(&p, dev));
        n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount / 8 * 8 : 256;
        if (n_cu <= 0) n_cu = 8;
    }
    const int ntiles = a.tiles_m * a.tiles_n;
    const dim3 grid(ntiles < n_cu ? ntiles : n_cu);
    if (mode == 0) hipLaunchKernelGGL(k_conv_pp<0>, grid, dim3(512), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL(k_conv_pp<2>, grid, dim3(512), 0, s, a);
    else if (mode == 3) hipLaunchKernelGGL(k_conv_pp<3>, grid, dim3(512), 0, s, a);
    else fail(MRCNN_ERR_UNSUPPORTED, "conv_pp: mode %d", mode);
}

}  // namespace mrcnn
