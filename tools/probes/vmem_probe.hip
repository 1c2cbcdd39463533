// vmem_probe.hip — issue rate of vector-memory wave-instructions per CU on gfx950: 16-B-per-lane stores and loads, full lines vs
// scattered 16-B pieces (the two shapes an epilogue can have), HBM-streaming windows.  Measurement tool:
//   hipcc --offload-arch=gfx950 -O3 -o vmem_probe tools/probes/vmem_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: store, a wave writes 1 KiB contiguous; 1: store, a wave writes 64 pieces of 16 B, 4 KiB apart (one pixel row each);
// 2: load contiguous; 3: load scattered; 4: load contiguous + store contiguous (alternating); 5: store, 2 rows x 512 B (LDS-staged epilogue shape)
template <int MODE>
__global__ __launch_bounds__(512) void k_vmem(char* buf, size_t per_block, int iters, float* sink)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    char* base = buf + (size_t)blockIdx.x * per_block;
    float4 acc = make_float4(0, 0, 0, 0);
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)t);
    for (int i = 0; i < iters; ++i) {
        size_t off;
        if (MODE == 0 || MODE == 2 || MODE == 4) off = ((size_t)(i * 8 + wave) * 1024 + lane * 16) % per_block;
        else if (MODE == 5) off = ((size_t)(i * 8 + wave) * 8192 + (lane >> 5) * 4096 + (lane & 31) * 16) % per_block;
        else off = ((size_t)lane * 4096 + (size_t)((i * 8 + wave) % 256) * 16 + (size_t)((i * 8 + wave) / 256) * 262144) % per_block;
        float4* p = reinterpret_cast<float4*>(base + off);
        if (MODE == 0 || MODE == 1 || MODE == 5) *p = v;
        else if (MODE == 2 || MODE == 3) { const float4 x = *p; acc.x += x.x; }
        else { if (i & 1) *p = v; else { const float4 x = *p; acc.x += x.x; } }
    }
    if (acc.x == 123.456f) sink[0] = acc.x;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2048;
    const size_t per_block = (size_t)32 << 20;          // 32 MiB per block: streaming (8 GiB over 256 blocks)
    char* buf; float* sink;
    CK(hipMalloc(&buf, per_block * 256));
    CK(hipMemset(buf, 0, per_block * 256));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int blocks) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, buf, per_block, iters, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                const double instr = (double)blocks * 8 * iters;
                printf("%-58s %4d CUs: %7.2f TB/s chip, %6.1f ns per wave-instr per CU (= %5.1f clk @2.0 GHz)\n", name, blocks,
                       instr * 1024 / (ms * 1e-3) / 1e12, ms * 1e6 / (8.0 * iters), ms * 1e6 / (8.0 * iters) * 2.0);
            }
        }
    };
    for (int blocks : {256, 32, 1}) {
        run("store 1 KiB contiguous per wave-instr", k_vmem<0>, blocks);
        run("store 2 rows x 512 B per wave-instr", k_vmem<5>, blocks);
        run("store 64 x 16 B pieces (4 KiB apart) per wave-instr", k_vmem<1>, blocks);
        run("load 1 KiB contiguous per wave-instr", k_vmem<2>, blocks);
        run("load 64 x 16 B pieces per wave-instr", k_vmem<3>, blocks);
        run("alternating load / store, contiguous", k_vmem<4>, blocks);
    }
    return 0;
}
