// ws_1x1_probe.hip — VERDICT r4 item 4, as a bounded PROBE (not product code): does a wave-specialised persistent block bring the split
// modes' HBM-bound 1x1 layers to their floor?  Shape = C4's `branch2c` at batch 8 (M = 32768 pixels, K = 256, N = 1024, fp32 activations,
// fp16 filters, three fp16 MFMA passes per product, y = relu(acc * scale + shift + residual) written in place over the residual):
// 302 MB per launch, 72.3 us in the engine (4.2 TB/s) against ~58 us at the 5.2 TB/s the chip streams.
//
// Block = 8 waves on one CU, persistent over its tiles (128 x 128):
//   waves 0-3 (MFMA): the four-wave 128 x 128 tile of kernels_conv.hip (wave = 32 rows x 128 columns, operands through a 3-stage LDS ring by
//              LDS-DMA, fp32 fragments split in registers) — the ring runs on ACROSS tile boundaries (no drain, no prologue per tile); after
//              a tile's last K step the raw sums go to a 128 x 132-float LDS tile;
//   waves 4-7 (STREAM): own the tile's HBM traffic — the residual of tile i is requested one whole tile ahead into registers (its latency hides
//              under tile i-1's K loop; a wave of the MFMA group could not do this: vector-memory loads return in order, its counted vmcnt
//              waits for the ring would wait for the slow HBM loads too), and while the MFMA waves multiply tile i the stream waves turn tile
//              i-1's sums into outputs: eight slices, one per K step (read 4 rows from the LDS tile, scale / shift / residual / ReLU, full
//              512-B lines out, request the next residual into the registers just consumed).
// One block barrier per K step (the ring's) also paces the slices; two more per tile hand the LDS tile over.
//   hipcc -O3 --offload-arch=gfx950 -Wno-inline-asm -I mask-rcnn-coreml_amd/csrc -I include tools/probes/ws_1x1_probe.hip -o /tmp/ws && /tmp/ws
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv_device.h"

using namespace mrcnn;

namespace {
constexpr int BM = 128, BN = 128, STAGES = 3;
constexpr int A_STAGE = BM * 128, B_STAGE = BN * 64;          // 32 fp32 / 32 fp16 channels per row and step
constexpr int RING = STAGES * (A_STAGE + B_STAGE);            // 72 KB
constexpr int CP = 132;                                       // pitch of the sums tile (floats): 16 lanes of a ds_write_b128 group on 16 distinct bank quads
constexpr int C_BYTES = BM * CP * 4;
constexpr int LDS_BYTES = RING + C_BYTES + 2 * BN * 4;

#define WS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define WS_GLDS(SRC, DST) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(SRC), "s"(DST) : "memory", "m0");
}  // namespace

__global__ __launch_bounds__(512) void k_ws_1x1(const float* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ Y,
                                                const float* __restrict__ scale, const float* __restrict__ shift, int M, int K, int N, int mode)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    float* const ctile = reinterpret_cast<float*>(smem + RING);
    float* const tab = reinterpret_cast<float*>(smem + RING + C_BYTES);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool mfma_wave = wave < 4;
    const int l31 = lane & 31, kk = lane >> 5;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    constexpr int KT = 8;                                        // the probe's K = 256
    const int tiles_n = N / BN, tiles_m = M / BM;
    // this block's tiles: row tiles blockIdx.x, + gridDim.x, ...; all column tiles of a row tile back to back (the activation rows stay in L2)
    const int my_rows = (tiles_m - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int T = my_rows * tiles_n;
    auto tile_m0 = [&](int i) { return ((int)blockIdx.x + (i / tiles_n) * (int)gridDim.x) * BM; };
    auto tile_n0 = [&](int i) { return (i % tiles_n) * BN; };

    // ---------------- MFMA waves: DMA bookkeeping (thread t of 256 stages 16-B pieces) ----------------
    const int r0 = (t & 255) >> 3;                               // A: row inside a 32-row pass
    const int kqa = (t & 7) ^ ((r0 >> 1) & 7);                   // source chunk held at LDS chunk position t & 7
    const int rb = (t & 255) >> 2;                               // B: row inside a 64-row pass
    const int kqb = (t & 3) ^ ((rb >> 2) & 3);
    const int G = T * KT;                                        // K steps of the whole block
    auto issue = [&](int g) {                                     // the DMAs of global step g into ring buffer g % STAGES
        const int i = g / KT, ks = g - i * KT, buf = g % STAGES;
        const int m0 = tile_m0(i), n0 = tile_n0(i);
        const unsigned da = lds0 + buf * A_STAGE + wave * 1024, db = lds0 + STAGES * A_STAGE + buf * B_STAGE + wave * 1024;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float* src = A + (size_t)(m0 + r0 + 32 * p) * K + ks * 32 + kqa * 4;
            WS_GLDS(src, da + p * 4096)
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const _Float16* src = W + (size_t)(n0 + rb + 64 * p) * K + ks * 32 + kqb * 8;
            WS_GLDS(src, db + p * 4096)
        }
    };
    // ---------------- STREAM waves: lane -> (row, 16-B column piece) of a slice ----------------
    const int sv = wave - 4;                                     // rows 32 sv .. 32 sv + 31 of the tile
    const int s_row = lane >> 5, s_col = (lane & 31) * 4;
    float4 res[8][2];
#pragma unroll
    for (int s = 0; s < 8; ++s) { res[s][0] = make_float4(0, 0, 0, 0); res[s][1] = make_float4(0, 0, 0, 0); }
    auto y_ptr = [&](int i, int s, int u) { return Y + (size_t)(tile_m0(i) + 32 * sv + 4 * s + 2 * u + s_row) * N + tile_n0(i) + s_col; };

    f32x16 acc[4];
    if (mfma_wave) {
        if (G > 0) issue(0);
        if (G > 1) issue(1);
        if (G > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (T > 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) res[s][u] = *reinterpret_cast<const float4*>(y_ptr(0, s, u));
    }
    WS_BARRIER
    const int swz = (l31 >> 1) & 7, swzb = (l31 >> 2) & 3;
    const int ca[4] = {((0 + 2 * kk) ^ swz) << 4, ((1 + 2 * kk) ^ swz) << 4, ((4 + 2 * kk) ^ swz) << 4, ((5 + 2 * kk) ^ swz) << 4};
    const int cb[2] = {((0 + kk) ^ swzb) << 4, ((2 + kk) ^ swzb) << 4};

    for (int i = 0; i <= T; ++i) {                                // iteration i: MFMA waves multiply tile i, stream waves finish tile i - 1
        if (mfma_wave) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
        } else if (i >= 1 && i - 1 < T) {
            // scale / shift of tile i - 1's columns (the table is read only by the stream waves, behind the hand-over barrier of iteration i - 1)
        }
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            const int g = i * KT + ks;
            if (mfma_wave) {
                if (i < T) {
                    if (g + 2 < G) issue(g + 2);
                    const int buf = g % STAGES;
                    const unsigned char* la = smem + buf * A_STAGE + (32 * wave + l31) * 128;
                    const unsigned char* lb = smem + STAGES * A_STAGE + buf * B_STAGE + l31 * 64;
#pragma unroll
                    for (int grp = 0; grp < 2; ++grp) {
                        const uint4 a0 = *reinterpret_cast<const uint4*>(la + ca[2 * grp]), a1 = *reinterpret_cast<const uint4*>(la + ca[2 * grp + 1]);
                        uint4 bv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const uint4*>(lb + j * 32 * 64 + cb[grp]);
                        f16x8 hi, mid, lo;
                        split_hi_mid_lo(a0, a1, hi, mid, lo);
                        if (!(mode & 1)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[j]), hi, acc[j], 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[j]), mid, acc[j], 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bv[j]), lo, acc[j], 0, 0, 0);
                        }
                    }
                    if (g + 2 < G) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else if (i >= 1 && !(mode & 2)) {
                // slice ks of tile i - 1 (KT = 8 slices of 4 rows; a probe: K = 256 only)
                const int s = ks;
                const int n0 = tile_n0(i - 1);
                const float4 sc = *reinterpret_cast<const float4*>(scale + n0 + s_col), sh = *reinterpret_cast<const float4*>(shift + n0 + s_col);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 c = *reinterpret_cast<const float4*>(ctile + (32 * sv + 4 * s + 2 * u + s_row) * CP + s_col);
                    const float4 r = res[s][u];
                    float4 y;
                    y.x = fmaxf(c.x * sc.x + sh.x + r.x, 0.f); y.y = fmaxf(c.y * sc.y + sh.y + r.y, 0.f);
                    y.z = fmaxf(c.z * sc.z + sh.z + r.z, 0.f); y.w = fmaxf(c.w * sc.w + sh.w + r.w, 0.f);
                    *reinterpret_cast<float4*>(y_ptr(i - 1, s, u)) = y;
                    if (i < T) res[s][u] = *reinterpret_cast<const float4*>(y_ptr(i, s, u));       // the next tile's residual into the registers just consumed
                }
            }
            WS_BARRIER
        }
        // hand-over: the stream waves are done with the LDS tile (last barrier above); the MFMA waves park tile i's sums
        if (mfma_wave && i < T) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<float4*>(ctile + (32 * wave + l31) * CP + 32 * j + 8 * g4 + 4 * kk) =
                        make_float4(acc[j][4 * g4 + 0], acc[j][4 * g4 + 1], acc[j][4 * g4 + 2], acc[j][4 * g4 + 3]);
        }
        WS_BARRIER
    }
    (void)tab;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 32768, K = 256, N = 1024;
    const int iters = argc > 2 ? atoi(argv[2]) : 50;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::vector<float> hA((size_t)M * K), hY((size_t)M * N), hs(N), hh(N);
    std::vector<_Float16> hW((size_t)N * K);
    unsigned h = 12345u;
    auto rnd = [&] { h = h * 1664525u + 1013904223u; return ((int)(h >> 9) & 0xffff) * (2.0f / 65536.0f) - 1.0f; };
    for (auto& v : hA) v = fmaxf(rnd() * 3.0f, 0.0f);
    for (auto& v : hW) v = (_Float16)(rnd() * 0.0625f);
    for (auto& v : hY) v = rnd();
    for (int n = 0; n < N; ++n) { hs[n] = 0.5f + 0.5f * fabsf(rnd()); hh[n] = 0.1f * rnd(); }
    float *dA, *dY, *ds, *dh; _Float16* dW;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dY, hY.size() * 4)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&ds, N * 4)); CK(hipMalloc(&dh, N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, hh.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ws_1x1), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int grid = M / BM < cus ? M / BM : cus;
    // correctness: one launch from a fresh residual, a few hundred outputs against a double-precision evaluation
    CK(hipMemcpy(dY, hY.data(), hY.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ws_1x1, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dW, dY, ds, dh, M, K, N, 0);
    CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)M * N);
    CK(hipMemcpy(out.data(), dY, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int q = 0; q < 4000; ++q) {
        h = h * 1664525u + 1013904223u; const int m = (int)((h >> 8) % (unsigned)M);
        h = h * 1664525u + 1013904223u; const int n = (int)((h >> 8) % (unsigned)N);
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * (double)(float)hW[(size_t)n * K + k];
        const double want = fmax(s * hs[n] + hh[n] + hY[(size_t)m * N + n], 0.0);
        worst = fmax(worst, fabs(want - out[(size_t)m * N + n]));
    }
    printf("M %d K %d N %d, grid %d x 512 threads, LDS %d B: max |err| over 4000 sampled outputs %.3e\n", M, K, N, grid, LDS_BYTES, worst);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)M * K * 4 + (double)N * K * 2 + 2.0 * M * N * 4, flop = 2.0 * M * N * K;
    for (int mode : {0, 1, 2, 3}) {
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_ws_1x1, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dW, dY, ds, dh, M, K, N, mode);
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_ws_1x1, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dW, dY, ds, dh, M, K, N, mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("mode %d (%s): %7.1f us per launch = %5.2f TB/s algorithmic, %6.1f TFLOP/s algorithmic\n", mode,
               mode == 0 ? "all" : mode == 1 ? "no MFMAs" : mode == 2 ? "no stream slices (sums never leave)" : "ring + barriers only", us, bytes / us * 1e-6, flop / us * 1e-6);
    }
    return 0;
}
