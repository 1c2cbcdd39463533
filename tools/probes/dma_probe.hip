// dma_probe.hip — throughput of the global→LDS DMA path (global_load_lds_dwordx4) per CU on gfx950, against plain
// global_load_dwordx4, for the access patterns of the conv kernels.  Measurement tool: build with
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe tools/probes/dma_probe.hip
// Each wave issues ITERS wave-instructions of 1 KiB (64 lanes × 16 B) with at most DEPTH outstanding.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// pattern 0: 8 rows × 128 B, rows `row_stride` bytes apart, chunks XOR-swizzled inside the row (the conv kernels' source pattern)
// pattern 1: same rows, linear chunks;  pattern 2: one contiguous 1 KiB run
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k_probe(const char* src, size_t window, int row_stride, int pattern, int iters, float* sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned dst = lds0 + wave * 16384;
    const int r = lane >> 3, c = lane & 7;
    size_t off;
    if (pattern == 2) off = (size_t)lane * 16;
    else off = (size_t)r * row_stride + (size_t)((pattern == 0 ? (c ^ ((r >> 1) & 7)) : c) * 16);
    // each block walks its own window (L2-resident after the first pass); waves start at different rows
    const char* base = src + ((size_t)blockIdx.x * window) % ((size_t)1 << 30);
    size_t pos = (size_t)wave * 8 * row_stride;
    const size_t step = pattern == 2 ? 1024 : (size_t)8 * row_stride * 8;   // next 8 rows of this wave (8 waves interleaved)
    uint4 acc = make_uint4(0, 0, 0, 0);
    typedef unsigned srd_t __attribute__((ext_vector_type(4)));
    const unsigned long long b64 = (unsigned long long)base;
    srd_t srd;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)b64);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32) & 0xffffu);      // stride 0
    srd[2] = 0xffffffffu;                                                           // num_records (bytes)
    srd[3] = 0x00020000u;                                                           // raw buffer, dword data format (gfx9 SRD word 3)
    for (int i = 0; i < iters; ++i) {
        const char* p = base + (pos % window) + off;
        if (MODE == 0) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(%2)" ::"v"(p), "s"(dst + (i & 15) * 1024), "n"(DEPTH) : "memory", "m0");
        } else if (MODE == 2) {          // scalar base + 32-bit lane offset
            const unsigned voff = (unsigned)((pos % window) + off);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_waitcnt vmcnt(%3)" ::"v"(voff), "s"(base), "s"(dst + (i & 15) * 1024), "n"(DEPTH) : "memory", "m0");
        } else if (MODE == 3) {          // buffer resource + 32-bit lane offset
            const unsigned voff = (unsigned)((pos % window) + off);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(%3)" ::"v"(voff), "s"(srd), "s"(dst + (i & 15) * 1024), "n"(DEPTH) : "memory", "m0");
        } else if (MODE == 4) {          // dword per lane (256 B per instruction), vaddr form: is the cost per instruction or per byte?
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\ts_waitcnt vmcnt(%2)" ::"v"(p), "s"(dst + (i & 15) * 1024), "n"(DEPTH) : "memory", "m0");
        } else {
            // bursts of 4 loads, waited inside the statement (a destination must not be reused while its load is in flight)
            uint4 v0, v1, v2, v3;
            const char* p1 = base + ((pos + step) % window) + off;
            const char* p2 = base + ((pos + 2 * step) % window) + off;
            const char* p3 = base + ((pos + 3 * step) % window) + off;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\tglobal_load_dwordx4 %2, %6, off\n\t"
                         "global_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p), "v"(p1), "v"(p2), "v"(p3) : "memory");
            acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x;
            pos += 3 * step; i += 3;
        }
        pos += step;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 0x12345678u) sink[0] = 1.0f;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t total = (size_t)1 << 30;
    char* src; float* sink;
    CK(hipMalloc(&src, total + (1 << 22)));
    CK(hipMemset(src, 1, total + (1 << 22)));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, size_t window, int row_stride, int pattern, int blocks) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, src, window, row_stride, pattern, iters, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                const double bytes = (double)blocks * 8 * iters * 1024;
                printf("%-44s window %6zu KB stride %5d: %7.1f GB/s per CU, %6.2f TB/s chip, %5.1f B/clk/CU @2.4GHz\n", name, window >> 10, row_stride,
                       bytes / blocks / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, bytes / blocks / (ms * 1e-3) / 2.4e9);
            }
        }
    };
    const char* pn[3] = {"swizzled rows", "linear rows", "contiguous 1KiB"};
    for (int pattern = 0; pattern < 3; ++pattern)
        for (size_t window : {(size_t)64 << 10, (size_t)2 << 20}) {
            char nm[96];
            snprintf(nm, sizeof nm, "LDS-DMA depth 8, %s", pn[pattern]); run(nm, k_probe<0, 8>, window, 4608, pattern, 256);
            snprintf(nm, sizeof nm, "LDS-DMA depth 16, %s", pn[pattern]); run(nm, k_probe<0, 16>, window, 4608, pattern, 256);
            snprintf(nm, sizeof nm, "to-VGPR depth 8, %s", pn[pattern]); run(nm, k_probe<1, 8>, window, 4608, pattern, 256);
        }
    run("LDS-DMA saddr+voffset depth 8, swizzled", k_probe<2, 8>, (size_t)64 << 10, 4608, 0, 256);
    run("LDS-DMA buffer_load offen depth 8, swizzled", k_probe<3, 8>, (size_t)64 << 10, 4608, 0, 256);
    run("LDS-DMA dword (256 B/instr) vaddr depth 8 [bytes x4 in the rate!]", k_probe<4, 8>, (size_t)64 << 10, 4608, 0, 256);
    run("LDS-DMA depth 8, swizzled, stride 128 (NHWC 64ch)", k_probe<0, 8>, (size_t)2 << 20, 128, 0, 256);
    run("LDS-DMA depth 8, swizzled, stride 512 (NHWC 256ch)", k_probe<0, 8>, (size_t)2 << 20, 512, 0, 256);
    run("LDS-DMA depth 8, swizzled, 1 CU only", k_probe<0, 8>, (size_t)64 << 10, 4608, 0, 1);
    run("LDS-DMA depth 8, swizzled, 32 CUs", k_probe<0, 8>, (size_t)64 << 10, 4608, 0, 32);
    return 0;
}
