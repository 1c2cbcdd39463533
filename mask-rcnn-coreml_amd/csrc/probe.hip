// probe.hip — what the matrix cores of THIS board sustain right now (test / measurement entry, include/maskrcnn_hip_test.h).
//
// The boxes of a pool differ by several per cent in the shader clock they hold under the package power cap, so a bench value alone
// cannot tell a +2 % kernel change from a slower box.  mrcnn_bench_mfma_probe runs every wave of the chip (CUs x 8 waves) on
// back-to-back v_mfma_f32_32x32x16_f16 (or v_mfma_f32_32x32x2_f32) with register operands that CHANGE from one MFMA to the next —
// the switching activity of real data, no operand traffic at all — for the requested time and returns the rate: the live roof
// bench.py holds the convolution kernels against in the same process (`roofline.sustained_peak_live`).  The kernel is the one of
// tools/probes/mfma_probe.hip (KIND 2 / KIND 1), moved behind the C ABI.
#include "common.h"

namespace mrcnn {

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));

__global__ void k_probe_mfma_f16(float* out, int iters)
{
    pf32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    pf16x8 a[4], b[4];
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

__global__ void k_probe_mfma_f32(float* out, int iters)
{
    pf32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int r = 0; r < 4; ++r) {
        h = h * 1664525u + 1013904223u; a[r] = ((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
        h = h * 1664525u + 1013904223u; b[r] = ((int)(h >> 9) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 3], b[(u + 1) & 3], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 3], b[(u + 2) & 3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 2) & 3], b[(u + 3) & 3], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 3) & 3], b[u & 3], c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    if (s == 12345.678f) out[0] = s;
}

}  // namespace mrcnn

using namespace mrcnn;

// kind: MRCNN_F16 (2) = v_mfma_f32_32x32x16_f16, MRCNN_F32 (0) = v_mfma_f32_32x32x2_f32.  *tflops = the sustained rate over the last
// `seconds`; *mhz_equivalent = the shader clock that rate means at back-to-back issue (32 / 64 clocks per instruction and SIMD).
extern "C" int mrcnn_bench_mfma_probe(double seconds, int kind, double* tflops, double* mhz_equivalent)
{
    return guarded([&] {
        require_gpu();
        MRCNN_REQUIRE(tflops && seconds > 0 && seconds <= 30 && (kind == MRCNN_F16 || kind == MRCNN_F32), MRCNN_ERR_INVALID, "bad mfma_probe argument");
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t p;
        HIP_CHECK(hipGetDeviceProperties(&p, dev));
        const int blocks = p.multiProcessorCount, waves = 8, iters = 20000;
        DevBuf out(16);
        hipStream_t s = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        auto launch = [&] {
            if (kind == MRCNN_F16) hipLaunchKernelGGL(k_probe_mfma_f16, dim3(blocks), dim3(64 * waves), 0, s, out.as<float>(), iters);
            else hipLaunchKernelGGL(k_probe_mfma_f32, dim3(blocks), dim3(64 * waves), 0, s, out.as<float>(), iters);
        };
        try {
            launch();                                     // warm-up; then one timed launch sizes the run
            HIP_CHECK(hipStreamSynchronize(s));
            float ms = 0;
            HIP_CHECK(hipEventRecord(e0, s));
            launch();
            HIP_CHECK(hipEventRecord(e1, s));
            HIP_CHECK(hipEventSynchronize(e1));
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            // the clocks settle within a few hundred ms: run for `seconds`, rate the LAST third
            const int total = (int)(seconds * 1e3 / ms) + 3, head = total - total / 3;
            for (int l = 0; l < head; ++l) launch();
            HIP_CHECK(hipEventRecord(e0, s));
            for (int l = head; l < total; ++l) launch();
            HIP_CHECK(hipEventRecord(e1, s));
            HIP_CHECK(hipEventSynchronize(e1));
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            HIP_CHECK(hipGetLastError());
            const double mfma = (double)(total - head) * blocks * waves * (double)iters * 32.0;
            const double flop = kind == MRCNN_F16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
            *tflops = mfma * flop / (ms * 1e-3) / 1e12;
            if (mhz_equivalent) *mhz_equivalent = mfma / (blocks * 4.0) / (ms * 1e3) * (kind == MRCNN_F16 ? 32.0 : 64.0);
        } catch (...) {
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
            throw;
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    });
}
