// Developer probe (GPU box): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 --
// (1) do out-of-range lanes write zeros into LDS, (2) does M0 address LDS above 64 KiB.
// build: hipcc -O3 --offload-arch=gfx950 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
}
__global__ void k(const float* in, float* out, int n, int ldsoff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    *(float4*)(smem + ldsoff + threadIdx.x * 16) = make_float4(-7.f, -7.f, -7.f, -7.f);   // sentinel
    __syncthreads();
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000u;        // out of range
    dma16(rs, voff, 0, __builtin_amdgcn_readfirstlane(base + ldsoff + (threadIdx.x >> 6) * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ((float4*)out)[threadIdx.x] = *(const float4*)(smem + ldsoff + threadIdx.x * 16);
}
int main() {
    const int n = 256 * 4;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
    float *din, *dout;
    hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int ldsoff : {0, 60 * 1024, 100 * 1024, 150 * 1024}) {
        hipMemset(dout, 0xff, n * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, din, dout, n, ldsoff);
        std::vector<float> o(n);
        hipError_t e = hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        int ok_in = 0, zero_oob = 0, sentinel_oob = 0, other = 0;
        for (int t = 0; t < 256; ++t)
            for (int j = 0; j < 4; ++j) {
                const float v = o[t * 4 + j];
                if (t & 1) { if (v == 0.f) ++zero_oob; else if (v == -7.f) ++sentinel_oob; else ++other; }
                else { if (v == h[t * 4 + j]) ++ok_in; else ++other; }
            }
        printf("ldsoff %6d: err=%d in-range ok %d/512, oob zero %d/512, oob untouched %d, other %d\n", ldsoff, (int)e, ok_in, zero_oob, sentinel_oob, other);
    }
    return 0;
}
