// kernels_conv_halo.hip — the 3×3 stride-1 convolutions of the split modes (fp32 tensors, fp16 matrix cores): persistent
// 128×BN tiles whose input HALO is staged and split ONCE.
//
// Why: in the split modes a fp32 activation is turned into 2 / 3 fp16 parts in registers before it meets
// the matrix cores.  The 128-row implicit-GEMM kernel (kernels_conv.hip) stages one (tap, 32 channels) slab per K step, so a
// 3×3 layer fetches every input pixel NINE times from L2 (once per tap) and every wave column splits it again: the split
// VALU was 18 % of the time of the large 3×3 layers and the L2 → LDS traffic 9× the input.  Here a tile's input region —
// the rows above / below and the columns left / right of its 128 output pixels — is loaded once per 16-channel slab,
// split once by the whole block on its way into LDS (fp16 hi / mid / lo planes), and the nine taps read SHIFTED windows of
// those planes: 9× fewer activation loads, 9·WN× less split work, no VALU between the LDS reads and the MFMAs.
//
//   out[m][n] = Σ_h Σ_tap Σ_{c < 16} A[pixel(m) + tap][16 h + c] · W[n][tap][16 h + c]
//
// K ORDER = (16-channel slab h, tap): this is the canonical summation order of every 3×3 stride-1 layer in the split modes,
// for EVERY batch size and tile width (the BN variants below only differ in which columns a block owns), so per-image
// results do not depend on the batch — the sharding contract.  Per (h, tap) step and accumulator: hi·w, mid·w, lo·w, each
// one v_mfma_f32_32x32x16_f16 (16 products + the fp32 accumulate).
//
// Data movement per 16-channel slab of a 128 × 256 tile:
//   activations  ≤ 528 input pixels × 64 B, buffer_load_dwordx4 into VGPRs one slab ahead (out-of-image pixels: an
//                out-of-range offset, the hardware returns zeros = the zero padding), split in registers, ds_write_b64 into
//                the other plane buffer while the nine taps of the current slab run;
//   filters      9 steps × BN × 32 B, pre-tiled at load into 1-KB granules [32 columns][16 channels] in exactly the LDS image
//                (conv_halo_pack), so a step's filter tile is BN/32 linear 1-KB global→LDS DMAs into a 4-slot ring;
//   MFMAs        9 × 3 × 4 per wave (wave tile 32 × 128), fragments by ds_read_b128 (conflict-free half-swizzle).
// One barrier per (h, tap) step (12 MFMAs per wave).  Blocks are persistent: one per CU, each XCD walks a contiguous run of
// tiles with its 32 CUs on 32 consecutive tiles (vertical neighbours share their halo rows in that XCD's L2).
//
// Epilogue: conv_epilogue_direct (conv_device.h) straight from the accumulators — the same arithmetic, in the same order,
// as every other kernel of the family.
#include "conv_device.h"

namespace mrcnn {

static constexpr int HALO_MAX_PX = 528;          // input pixels a tile may need (rows above/below + columns left/right included)
static constexpr int HALO_RING = 4;              // filter-tile ring slots
static constexpr unsigned HALO_OOB = 0xC0000000u;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct HaloArgs {
    ConvArgs a;
    const void* wgt_halo;        // conv_halo_pack layout
    int NH;                      // 16-channel slabs = Cin / 16
    int single_row;              // OW % 128 == 0: a tile lies inside one image row (cropped halo, pitch 130)
    int n_tiles;
};

// 4 fp32 → PARTS × 4 fp16 (the same round-toward-zero chain as split_hi_mid_lo / split_hi_lo of conv_device.h)
template <int PARTS>
__device__ __forceinline__ void split4(const u32x4 v, u32x2 (&out)[PARTS])
{
    const float a[4] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t h2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a[2 * p], a[2 * p + 1]));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h2), "v"(a[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h2), "v"(a[2 * p + 1]));
        const uint32_t m2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(r0, r1));
        out[0][p] = h2;
        out[1][p] = m2;
        if constexpr (PARTS == 3) {
            float q0, q1;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(m2), "v"(r0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(m2), "v"(r1));
            out[2][p] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(q0, q1));
        }
    }
}

template <int PARTS, int TN>
__global__ __launch_bounds__(512, 2) void k_conv_halo(const HaloArgs ha)
{
    const ConvArgs& a = ha.a;
    constexpr int BM = 128, WN = 2, BN = WN * TN * 32;
    constexpr int NG = BN / 32;                           // filter granules (1 KB) per step
    constexpr int PLANE = HALO_MAX_PX * 32;               // bytes of one part of one slab
    constexpr int PBUF = PARTS * PLANE;                   // one plane buffer (all parts)
    constexpr int RSLOT = BN * 32;                        // one ring slot
    constexpr int MAXPC = (HALO_MAX_PX * 4 + 511) / 512;  // 64-B pieces per thread per slab (5)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PBUF + HALO_RING * RSLOT + 2 * BN * 4];
    unsigned char* const planes = smem;
    unsigned char* const ring = smem + 2 * PBUF;
    float* const s_tab = reinterpret_cast<float*>(smem + 2 * PBUF + HALO_RING * RSLOT);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kk = lane >> 5;
    const int ohw = a.OH * a.OW;
    const int NH = ha.NH, NS = NH * 9;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const bool has_b = wave < NG;                         // this wave issues one filter granule per step

    // ---- the tiles of this block: XCD x owns a contiguous run, its CUs take consecutive tiles -----------------------
    const int T = ha.n_tiles;
    const int nb = gridDim.x;
    const int bid = blockIdx.x;
    int t_first, t_end, t_step;
    {
        const int q = T >> 3, r8 = T & 7;
        const int xcd = bid & 7, j = bid >> 3;
        const int lo = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
        const int cnt = q + (xcd < r8 ? 1 : 0);
        const int per = nb >> 3;                          // blocks per XCD (nb is a multiple of 8, or nb == T < 8·… handled by the host)
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
    // fragment addresses that do not depend on the tile: filter rows of this wave's column tiles inside a ring slot
    const unsigned b_lane = (unsigned)((wn * TN) * 1024 + l31 * 32 + ((kk ^ ((l31 >> 3) & 1)) << 4));

    for (int tile = t_first; tile < t_end; tile += t_step) {
        const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
        const int m0 = mt * BM, n0 = nt * BN;
        // ---- geometry of the tile's input region (uniform) --------------------------------------------------------
        const int m_last = (m0 + BM - 1 < a.M ? m0 + BM - 1 : a.M - 1);
        const int b0 = m0 / ohw, rem0 = m0 - b0 * ohw, oh0 = rem0 / a.OW, ow0 = rem0 - oh0 * a.OW;
        const int b1 = m_last / ohw, rem1 = m_last - b1 * ohw, oh1 = rem1 / a.OW;
        const int Hp = a.H + 2;
        const int gy_first = b0 * Hp + oh0 + 1;                           // padded global row of the first output pixel
        const int gy_last = b1 * Hp + oh1 + 1;
        const int pitch = ha.single_row ? 130 : a.W + 2;
        const int col0 = ha.single_row ? ow0 - 1 : -1;                    // input column of buffer column 0
        const int rows = gy_last - gy_first + 3;
        const int npx = rows * pitch;                                     // <= HALO_MAX_PX (host-checked bound)
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
            const int r = px / pitch, c = px - r * pitch;
            const int gy = gy_first - 1 + r;
            const int b = gy / Hp, y = gy - b * Hp - 1;
            const int x = col0 + c;
            const bool ok = px < npx && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && b < a.B;
            p_off[i] = ok ? (unsigned)(((long)(b - b0) * a.in_sB + (long)y * a.in_sH + (long)x * a.in_sW + qt * 4) * 4) : HALO_OOB;
            p_lds[i] = px < npx ? (unsigned)(px * 32 + (((qt >> 1) ^ ((px >> 3) & 1)) << 4) + (qt & 1) * 8) : 0xffffffffu;
        }
        // ---- this lane's output pixel → index of its tap (0,0) input pixel in the region -----------------------------
        int base_idx;
        {
            const int m = m0 + wm * 32 + l31;
            const int mm = m < a.M ? m : a.M - 1;
            const int b = mm / ohw, rem = mm - b * ohw, oh = rem / a.OW, ow = rem - oh * a.OW;
            base_idx = ha.single_row ? (ow - ow0) : (b * Hp + oh + 1 - gy_first) * pitch + ow;
        }
        unsigned sob = (unsigned)(((size_t)(n0 / 32 + wave) * NS) * 1024u);   // this wave's granule stream (has_b waves only)

        // scale / shift of the tile's columns for the epilogue (read after many barriers)
        if (t < BN / 2) {
            const int c = (t < BN / 4 ? t : t - BN / 4) * 4;
            const float* src = t < BN / 4 ? a.scale : a.shift;
            const float fill = t < BN / 4 ? 1.0f : 0.0f;
            *reinterpret_cast<float4*>(&s_tab[(t < BN / 4 ? 0 : BN) + c]) =
                src ? *reinterpret_cast<const float4*>(src + n0 + c) : make_float4(fill, fill, fill, fill);
        }

#define HALO_LOAD(I) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st[I]) : "v"(p_off[I]), "s"(srdA) : "memory");
#define HALO_LOADS()  { HALO_LOAD(0) HALO_LOAD(1) HALO_LOAD(2) HALO_LOAD(3) HALO_LOAD(4) }
#define HALO_ADVANCE() { _Pragma("unroll") for (int i = 0; i < MAXPC; ++i) p_off[i] += (p_off[i] < HALO_OOB ? 64u : 0u); }
#define HALO_PIN()    asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4]));
#define HALO_WRITE(BUF)                                                                                          \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MAXPC; ++i) {                                                      \
            if (p_lds[i] != 0xffffffffu) {                                                                       \
                u32x2 parts[PARTS];                                                                              \
                split4<PARTS>(st[i], parts);                                                                     \
                _Pragma("unroll") for (int p = 0; p < PARTS; ++p)                                                \
                    *reinterpret_cast<u32x2*>(planes + (BUF) * PBUF + p * PLANE + p_lds[i]) = parts[p];          \
            }                                                                                                    \
        }                                                                                                        \
    }
#define HALO_DMA_B(SLOT)                                                                                         \
    {                                                                                                            \
        if (has_b) {                                                                                             \
            const unsigned dst = lds0 + 2 * PBUF + (SLOT) * RSLOT + wave * 1024;                                 \
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vlane16), "s"(srdB), "s"(sob), "s"(dst) : "memory", "m0"); \
            sob += 1024u;                                                                                        \
        }                                                                                                        \
    }
        static_assert(MAXPC == 5, "the staging statements are spelled out for five pieces per thread");
        u32x4 st[MAXPC];

        // ---- prologue: slab 0 into plane buffer 0, filter tiles of steps 0..RING-2 ------------------------------------
        HALO_LOADS()
        HALO_ADVANCE()
        HALO_DMA_B(0)
        if (NS > 1) HALO_DMA_B(1)
        if (NS > 2) HALO_DMA_B(2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HALO_PIN()
        HALO_WRITE(0)
        __syncthreads();

        f32x16 acc[1][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.0f;

        // ---- main loop ----------------------------------------------------------------------------------------------
        // vmcnt bookkeeping of a has_b wave (in issue order; every op is one wave instruction): per step one filter DMA, plus the
        // five slab loads right behind the DMA of tap 0.  The barrier at the end of step s hands over the filter tile of step
        // s + 1, issued RING - 2 = 2 steps earlier: the ops issued after it are the DMAs of steps s - 1 ... s (2) and, when the
        // slab loads fall in between (taps 0..2), those five.  A wave without a DMA only has the five loads.
        int step = 0;
        for (int h = 0; h < NH; ++h) {
            const int pb = h & 1;
            const bool next_slab = h + 1 < NH;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap, ++step) {
                if (step + HALO_RING - 1 < NS) HALO_DMA_B((step + HALO_RING - 1) & (HALO_RING - 1))
                if (tap == 0 && next_slab) { HALO_LOADS() HALO_ADVANCE() }
                // fragments: this lane's pixel shifted by the tap, three parts; the wave's TN column tiles
                const int idx = base_idx + (tap / 3) * pitch + (tap % 3);
                const unsigned a_addr = (unsigned)(idx << 5) + (unsigned)(((kk << 4) ^ ((idx << 1) & 16)));
                uint4 av[PARTS], bv[TN];
#pragma unroll
                for (int p = 0; p < PARTS; ++p) av[p] = *reinterpret_cast<const uint4*>(planes + pb * PBUF + p * PLANE + a_addr);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const uint4*>(ring + (step & (HALO_RING - 1)) * RSLOT + b_lane + j * 1024);
#pragma unroll
                for (int p = 0; p < PARTS; ++p)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[j]), __builtin_bit_cast(f16x8, av[p]), acc[0][j], 0, 0, 0);
                if (tap == 4 && next_slab) {
                    // the slab loads were retired by the counted wait of tap 3 at the latest (every wave): split and park
                    // them in the other plane buffer, whose last reader finished five barriers ago
                    HALO_PIN()
                    HALO_WRITE(pb ^ 1)
                }
                // the filter tile of the next step must have landed (and at tap 3: the slab loads, in every wave)
                const bool more = step + HALO_RING - 1 < NS;           // a DMA was issued this step
                const bool more1 = step + HALO_RING - 2 < NS;          // ... and in the previous step
                if (tap <= 2 && next_slab) {
                    if (has_b && more && more1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    else if (has_b && (more || more1)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                    if (tap == 2) { /* next wait retires the loads */ }
                } else {
                    if (has_b && more && more1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if (has_b && (more || more1)) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
            }
        }
#undef HALO_DMA_B
#undef HALO_WRITE
#undef HALO_PIN
#undef HALO_ADVANCE
#undef HALO_LOADS
#undef HALO_LOAD
        conv_epilogue_direct<float, BN, 1, TN>(a, acc, s_tab, m0 + wm * 32, n0, wn * TN * 32, lane);
        __syncthreads();          // s_tab / plane buffer 0 are rewritten by the next tile's prologue
    }
}

// ----------------------------------------------------------------------------------------------------------------
// filter re-tiling: [Npad][9][Cin] fp16 (the family's packing) → granules [Npad/32][Cin/16][9][32 rows × 32 B], the 16-B halves
// of a row swapped when (row >> 3) & 1 (the LDS image the kernel's ds_read_b128 expects: conflict-free)
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_pack(const uint4* __restrict__ src, int Npad, int Cin, uint4* __restrict__ dst)
{
    const int NH = Cin / 16;
    const long total = (long)(Npad / 32) * NH * 9 * 64;         // 16-B pieces
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int piece = (int)(e & 63);
        const long gs = e >> 6;                                   // (granule, step)
        const int step = (int)(gs % (NH * 9));
        const int g = (int)(gs / (NH * 9));
        const int h = step / 9, tap = step - h * 9;
        const int r = piece >> 1, pos = piece & 1;
        const int half = pos ^ ((r >> 3) & 1);                    // the half stored at this position
        const int n = g * 32 + r;
        dst[e] = src[(((long)n * 9 + tap) * Cin + 16 * h + 8 * half) / 8];
    }
}

void conv_halo_pack(hipStream_t s, const void* wgt_std, int Npad, int Cin, DevBuf& out)
{
    MRCNN_REQUIRE(Npad % 32 == 0 && Cin % 16 == 0, MRCNN_ERR_SHAPE, "halo packing: Npad %d / Cin %d", Npad, Cin);
    const size_t bytes = (size_t)Npad * 9 * Cin * 2;
    out.alloc(bytes);
    const long total = (long)bytes / 16;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_halo_pack, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(wgt_std), Npad, Cin, static_cast<uint4*>(out.p));
    HIP_CHECK(hipGetLastError());
}

// Largest input region (pixels) any tile of this geometry needs; > HALO_MAX_PX → the layer stays on the 128-row kernel.
static int halo_region_bound(int H, int W)
{
    if (W % 128 == 0) return 3 * 130;
    const int rows_touched = (127 + W - 1) / W + 1;
    const bool straddle = ((long)H * W) % 128 != 0;               // a tile may span two images: two zero rows in between
    return (rows_touched + (straddle ? 2 : 0) + 2) * (W + 2);
}

bool conv_halo_eligible(const ConvDesc& d)
{
    const int wdtype = d.wdtype < 0 ? d.dtype : d.wdtype;
    const bool split = d.dtype == MRCNN_F32 && (wdtype == MRCNN_F16 || wdtype == MRCNN_F32X3);
    if (!split || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.padH != 1 || d.padW != 1) return false;
    if (d.OH != d.H || d.OW != d.W || d.Cin % 16 != 0 || d.Npad % 64 != 0 || d.Cout % 4 != 0) return false;
    if (d.deconv2 || d.out2 || d.sel_partial || d.act == ACT_SIGMOID || d.res) return false;
    if (d.H >= 32760 || d.W >= 32760 || (double)d.in_sB * 8.0 >= 2.0e9) return false;
    return halo_region_bound(d.H, d.W) <= HALO_MAX_PX;
}

template <int PARTS>
static void halo_launch(hipStream_t s, const HaloArgs& ha, int bn, int grid)
{
    if (bn == 256) hipLaunchKernelGGL((k_conv_halo<PARTS, 4>), dim3(grid), dim3(512), 0, s, ha);
    else if (bn == 128) hipLaunchKernelGGL((k_conv_halo<PARTS, 2>), dim3(grid), dim3(512), 0, s, ha);
    else hipLaunchKernelGGL((k_conv_halo<PARTS, 1>), dim3(grid), dim3(512), 0, s, ha);
}

// a: filled by conv_forward (M, strides, epilogue fields, vec_ok checked by the caller); returns the N tile used
int conv_halo_forward(hipStream_t s, ConvArgs a, const ConvDesc& d, int parts, int n_cus)
{
    HaloArgs ha;
    // tile width: the widest whose tiles fill the chip once (the K order, hence the result, does not depend on it)
    const int tiles_m = (a.M + 127) / 128;
    int bn = d.Npad % 256 == 0 ? 256 : (d.Npad % 128 == 0 ? 128 : 64);
    while (bn > 64 && (long)tiles_m * (d.Npad / bn) < n_cus) bn >>= 1;
    a.tiles_m = tiles_m;
    a.tiles_n = d.Npad / bn;
    a.direct = 1;
    ha.a = a;
    ha.wgt_halo = d.wgt_halo;
    ha.NH = d.Cin / 16;
    ha.single_row = d.W % 128 == 0 ? 1 : 0;
    ha.n_tiles = a.tiles_m * a.tiles_n;
    int grid = ha.n_tiles < n_cus ? ha.n_tiles : n_cus;
    if (grid >= 8) grid &= ~7;                  // a multiple of 8: every XCD runs the same number of blocks
    if (parts == 3) halo_launch<3>(s, ha, bn, grid);
    else halo_launch<2>(s, ha, bn, grid);
    return bn;
}

}  // namespace mrcnn
