// m0_probe.hip — does an LDS-DMA (buffer_load_dwordx4 ... lds) reach LDS addresses at and beyond 128 KB (M0 bit 17)?
//   hipcc --offload-arch=gfx950 -O2 -o m0_probe m0_probe.hip ; ./m0_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned srd_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(const unsigned* src, unsigned* out, unsigned dst)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[163840];
    const int lane = threadIdx.x;
    for (int i = lane; i < 163840 / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    srd_t srd;
    const unsigned long long u = (unsigned long long)(uintptr_t)src;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)u);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xffffu);
    srd[2] = 4096; srd[3] = 0x00020000u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned voff = lane * 16;
    const unsigned d = lds0 + dst;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(srd), "s"(d) : "memory", "m0");
    __syncthreads();
    // where did the first dword (value 0x1000) land?
    unsigned found = 0xffffffffu;
    for (int i = lane; i < 163840 / 4; i += 64) if (reinterpret_cast<unsigned*>(smem)[i] == 0x1000u) found = i * 4;
    for (int o = 32; o > 0; o >>= 1) { unsigned other = __shfl_xor(found, o); found = found < other ? found : other; }
    if (lane == 0) { out[0] = found; out[1] = reinterpret_cast<unsigned*>(smem)[dst / 4]; }
}
int main()
{
    unsigned h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
    unsigned *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, 64);
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    for (unsigned dst : {0u, 65536u, 130048u, 131072u, 140288u, 161792u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, dst);
        unsigned r[2];
        hipMemcpy(r, out, 8, hipMemcpyDeviceToHost);
        printf("dst %6u: first dword landed at %d (%s), smem[dst] = 0x%x\n", dst, (int)r[0], r[0] == dst ? "OK" : "ELSEWHERE", r[1]);
    }
    return 0;
}
