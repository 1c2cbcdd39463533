// rowpitch_probe.hip — does the ROW PITCH of the activation operand bound a 1x1 layer's HBM rate?  (late round 4)
// Emulates the A-operand stream of the 128-row kernel on C4's branch2a (M = 32 768 pixels, 1024 fp32 channels = 4-KB rows): 512 blocks
// (two per CU), each streams its 128 rows in 32 K steps of 128 B per row via LDS-DMA-sized requests (16 B per lane, 8 lanes per row),
// two steps in flight.  Layouts:  0 = NHWC as the engine stores it (row pitch 4 KB: a step touches 128 different 4-KB rows)
//                                 1 = K-blocked per 128-pixel tile ([tile][K/32][128 rows][32 ch]: a step reads 16 KB contiguous)
//   hipcc --offload-arch=gfx950 -O3 -o rowpitch_probe tools/probes/rowpitch_probe.hip && ./rowpitch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LAYOUT>
__global__ __launch_bounds__(512, 2) void k_stream(const char* __restrict__ buf, int tiles, int ksteps, int row_bytes, float* sink)
{
    const int t = threadIdx.x;
    float acc = 0.f;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const char* base = buf + (size_t)tile * 128 * row_bytes;
        float4 cur[2], nxt[2];
        auto addr = [&](int step, int p) -> const float4* {
            const int r = (t >> 3) + 64 * p, c = t & 7;                       // 512 threads: 64 rows x 8 pieces per pass, two passes
            if (LAYOUT == 0) return reinterpret_cast<const float4*>(base + (size_t)r * row_bytes + step * 128 + c * 16);
            return reinterpret_cast<const float4*>(base + (size_t)step * (128 * 128) + r * 128 + c * 16);
        };
        cur[0] = *addr(0, 0); cur[1] = *addr(0, 1);
        for (int s = 0; s < ksteps; ++s) {
            if (s + 1 < ksteps) { nxt[0] = *addr(s + 1, 0); nxt[1] = *addr(s + 1, 1); }
            acc += cur[0].x + cur[1].w;
            // a K step's worth of "compute": keep the block busy ~0.8 us like 12 MFMAs x 4 waves per SIMD would
            for (int d = 0; d < 24; ++d) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            cur[0] = nxt[0]; cur[1] = nxt[1];
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main()
{
    const int M = 32768 * 4, K = 1024, row_bytes = K * 4, tiles = M / 128, ksteps = K / 32;      // 4 x the layer: 512 MB, well past the caches
    char* buf; float* sink;
    CK(hipMalloc(&buf, (size_t)M * row_bytes));
    CK(hipMemset(buf, 0, (size_t)M * row_bytes));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        for (int layout = 0; layout < 2; ++layout) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) {
                if (layout == 0) hipLaunchKernelGGL(k_stream<0>, dim3(512), dim3(512), 0, 0, buf, tiles, ksteps, row_bytes, sink);
                else hipLaunchKernelGGL(k_stream<1>, dim3(512), dim3(512), 0, 0, buf, tiles, ksteps, row_bytes, sink);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("layout %d (%s): %.1f us per pass over %.0f MB = %.2f TB/s\n", layout, layout ? "K-blocked tiles, 16 KB contiguous per step" : "NHWC, 4-KB row pitch",
                            ms / 5 * 1e3, (double)M * row_bytes / 1e6, (double)M * row_bytes / (ms / 5 * 1e-3) / 1e12);
        }
    return 0;
}
