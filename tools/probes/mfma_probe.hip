// mfma_probe.hip — what the matrix cores of this board sustain with NO data movement at all: every wave of a full
// chip (256 CUs x 8 waves, or x 4) issues v_mfma_f32_32x32x16_f16 (or v_mfma_f32_32x32x2_f32) back to back on
// register operands, four independent accumulators per wave, for a few seconds — long enough for the power
// management to settle.  Prints the rate; tools/power_probe.sh-style sampling of rocm-smi runs beside it
// (tools/refresh_profiles.sh).  The number is the roof the convolution kernels can be held against on a
// power-capped board: nominal peak = 2.5 PFLOP/s (fp16) at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; ./mfma_probe [seconds] [waves_per_block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 2: as KIND 0 but on RANDOM operands that change from one MFMA to the next (four register pairs of pseudo-random
// fp16 values in [-2, 2), cycled): the switching activity of real data — what the cap allows a kernel that does nothing
// but multiply.
__global__ void k_mfma_random(float* out, int iters)
{
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f16x8 a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; a[r][i] = (_Float16)(((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f);
            h = h * 1664525u + 1013904223u; b[r][i] = (_Float16)(((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u + 1) & 3], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 1) & 3], b[(u + 2) & 3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 2) & 3], b[(u + 3) & 3], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 3) & 3], b[u & 3], c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
__global__ void k_mfma(float* out, int iters)
{
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    const float fa = 0.001f * threadIdx.x, fb = 0.002f * (threadIdx.x ^ 5);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c3, 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
static void launch(int blocks, int waves, float* out, int iters)
{
    if (KIND == 2) hipLaunchKernelGGL(k_mfma_random, dim3(blocks), dim3(64 * waves), 0, 0, out, iters);
    else hipLaunchKernelGGL(k_mfma<KIND>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters);
}

template <int KIND>
static void run(const char* name, double flop_per_mfma, double seconds, int waves)
{
    float* out;
    CK(hipMalloc(&out, 4));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 20000;
    launch<KIND>(blocks, waves, out, iters);      // warm-up + calibration
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch<KIND>(blocks, waves, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int launches = (int)(seconds * 1e3 / ms) + 1;
    CK(hipEventRecord(e0));
    for (int l = 0; l < launches; ++l) launch<KIND>(blocks, waves, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)launches * blocks * waves * (double)iters * 32.0;
    printf("%-28s %d CUs x %d waves: %.1f s, %.1f TFLOP/s, %.2f MFMA/SIMD/us -> %.0f MHz-equivalent at back-to-back issue\n", name, blocks, waves,
           ms * 1e-3, mfma * flop_per_mfma / (ms * 1e-3) / 1e12, mfma / (blocks * 4.0) / (ms * 1e3),
           mfma / (blocks * 4.0) / (ms * 1e3) * (KIND == 1 ? 64.0 : 32.0));
    fflush(stdout);
    CK(hipFree(out));
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int waves = argc > 2 ? atoi(argv[2]) : 8;
    const int kind = argc > 3 ? atoi(argv[3]) : -1;
    if (kind < 0 || kind == 0) run<0>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, seconds, waves);
    if (kind < 0 || kind == 1) run<1>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, seconds, waves);
    if (kind < 0 || kind == 2) run<2>("32x32x16_f16, random data", 2.0 * 32 * 32 * 16, seconds, waves);
    return 0;
}
