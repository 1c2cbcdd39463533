// mfma_lds_probe.hip — what do the matrix cores sustain under the package power cap when their operands really come from LDS / L2?
// Every wave of the chip runs back-to-back v_mfma_f32_32x32x16_f16 on random fp16 data; per 4 MFMAs it issues R ds_read_b128 (1 KB
// each, conflict-free) whose results ARE the next operands, and optionally G 1-KB global loads from a 2-MB L2-resident buffer.
// D 1-KB global->LDS DMAs per 32 MFMAs model the operand ring.  R = 0 is the register-only roof (profiles/*_mfma_power.txt); R = 3 is the 0.75 fragment reads per MFMA of a 64 x 128 wave tile,
// R = 2 the 0.5 of a 128 x 128 one, R = 4 one read per MFMA (a 32-row wave tile against 128 columns reads 1.25).
//   hipcc -O3 --offload-arch=gfx950 mfma_lds_probe.hip -o /tmp/mfma_lds_probe && /tmp/mfma_lds_probe [seconds per point]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int G, int D = 0>
__global__ __launch_bounds__(512) void k_probe(const uint4* __restrict__ gbuf, float* out, int iters)
{
    __shared__ uint4 lds[4096 + 8 * 64 * 2];      // 64 KB of random fp16 (+ 16 KB of DMA landing area: 2 KB per wave)
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        unsigned w[4];
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            const _Float16 lo = (_Float16)(((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f);
            h = h * 1664525u + 1013904223u;
            const _Float16 hi = (_Float16)(((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f);
            w[k] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        lds[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    uint4 a[4], b[4];
    for (int r = 0; r < 4; ++r) { a[r] = lds[(r * 64 + lane) & 4095]; b[r] = lds[(256 + r * 64 + lane) & 4095]; }
    unsigned off = wave * 448u;
    const uint4* g = gbuf + lane + wave * 8192;
    unsigned goff = 0;
    const unsigned dma_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(lds + 4096 + wave * 128));
    const uint4* gd = gbuf + lane + wave * 8192;
    unsigned doff = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (R >= 1) a[u & 3] = lds[(off + (u * 4 + 0) * 64 + lane) & 4095];
            if (R >= 2) b[(u + 1) & 3] = lds[(off + (u * 4 + 1) * 64 + lane) & 4095];
            if (R >= 3) a[(u + 1) & 3] = lds[(off + (u * 4 + 2) * 64 + lane) & 4095];
            if (R >= 4) b[(u + 2) & 3] = lds[(off + (u * 4 + 3) * 64 + lane) & 4095];
            if (D >= 1 && (u % (8 / D)) == 0) {
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gd + doff), "s"(__builtin_amdgcn_readfirstlane(dma_dst + ((u & 1) << 10))) : "memory", "m0");
                doff = (doff + 64) & 8191;
            }
            if (G >= 1 && (u % (8 / G)) == 0) { b[(u + 3) & 3] = g[goff]; goff = (goff + 64) & 8191; }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[u & 3]), __builtin_bit_cast(f16x8, b[(u + 1) & 3]), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u + 1) & 3]), __builtin_bit_cast(f16x8, b[(u + 2) & 3]), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u + 2) & 3]), __builtin_bit_cast(f16x8, b[(u + 3) & 3]), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u + 3) & 3]), __builtin_bit_cast(f16x8, b[u & 3]), c3, 0, 0, 0);
        }
        off += 2048u + 64u;
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R, int G, int D = 0>
static void run(const char* what, int waves, double seconds, const uint4* gbuf, float* out, int cus)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL((k_probe<R, G, D>), dim3(cus), dim3(64 * waves), 0, 0, gbuf, out, iters); };
    launch(); CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const int total = (int)(seconds * 1e3 / ms) + 3, head = total - total / 3;
    for (int l = 0; l < head; ++l) launch();
    CK(hipEventRecord(e0));
    for (int l = head; l < total; ++l) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)(total - head) * cus * waves * (double)iters * 32.0;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    const double mhz = mfma / (cus * 4.0) / (ms * 1e3) * 32.0;
    printf("%-64s %d waves/CU: %7.1f TFLOP/s  (%4.0f MHz-equivalent at back-to-back issue)\n", what, waves, tf, mhz);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double sec = argc > 1 ? atof(argv[1]) : 2.5;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint4* gbuf; float* out;
    std::vector<unsigned> hostg(8 * 8192 * 4 + 4096);
    unsigned h = 777u;
    for (auto& w : hostg) {
        h = h * 1664525u + 1013904223u; const _Float16 lo = (_Float16)(((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f);
        h = h * 1664525u + 1013904223u; const _Float16 hi = (_Float16)(((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f);
        w = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    CK(hipMalloc(&gbuf, hostg.size() * 4)); CK(hipMemcpy(gbuf, hostg.data(), hostg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 64));
    for (int waves : {8, 4}) {
        run<0, 0>("registers only (0 operand reads per MFMA)", waves, sec, gbuf, out, cus);
        run<2, 0>("0.50 LDS fragment reads per MFMA (128 x 128 wave tile)", waves, sec, gbuf, out, cus);
        run<3, 0>("0.75 LDS fragment reads per MFMA (64 x 128 wave tile)", waves, sec, gbuf, out, cus);
        run<4, 0>("1.00 LDS fragment reads per MFMA", waves, sec, gbuf, out, cus);
        run<3, 2>("0.75 LDS reads + 1 KB from L2 per 16 MFMAs", waves, sec, gbuf, out, cus);
        run<2, 8>("0.50 LDS reads + 1 KB from L2 per 4 MFMAs (fragment streaming)", waves, sec, gbuf, out, cus);
        run<3, 0, 8>("0.75 LDS reads + 1 KB L2->LDS DMA per 4 MFMAs (256 x 256 tile ring)", waves, sec, gbuf, out, cus);
        run<3, 0, 4>("0.75 LDS reads + 1 KB L2->LDS DMA per 8 MFMAs", waves, sec, gbuf, out, cus);
        run<4, 0, 8>("1.00 LDS reads + 1 KB L2->LDS DMA per 4 MFMAs", waves, sec, gbuf, out, cus);
    }
    return 0;
}
