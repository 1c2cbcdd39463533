#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t cvt_pk_rne(float x, float y)
{
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__global__ void k(const float* a, unsigned* o) {
    const float a0 = a[threadIdx.x], a1 = a[threadIdx.x + 64];
    const uint32_t h2 = cvt_pk_rne(a0, a1);
    float r0, r1, q0, q1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h2), "v"(a0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h2), "v"(a1));
    const uint32_t m2 = cvt_pk_rne(r0, r1);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(m2), "v"(r0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(m2), "v"(r1));
    const uint32_t l2 = cvt_pk_rne(q0, q1);
    o[threadIdx.x * 3] = h2; o[threadIdx.x * 3 + 1] = m2; o[threadIdx.x * 3 + 2] = l2;
}
int main() {
    float* a; unsigned* o; hipMalloc(&a, 128 * 4); hipMalloc(&o, 64 * 12);
    float h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.37f * (i + 1) * (i % 3 ? 1.f : -1.f) + 1e-4f * i;
    hipMemcpy(a, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o);
    unsigned r[192]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        auto f = [](unsigned short u) { _Float16 v; __builtin_memcpy(&v, &u, 2); return (double)v; };
        for (int half = 0; half < 2; ++half) {
            const double s = f(r[3 * i] >> (16 * half)) + f(r[3 * i + 1] >> (16 * half)) + f(r[3 * i + 2] >> (16 * half));
            const double want = h[i + 64 * half];
            if (s != want) { ++bad; if (bad < 6) printf("lane %d half %d: parts %g %g %g sum %.9g want %.9g\n", i, half, f(r[3*i] >> (16*half)), f(r[3*i+1] >> (16*half)), f(r[3*i+2] >> (16*half)), s, want); }
        }
    }
    printf("bad %d of 128\n", bad);
    return 0;
}
