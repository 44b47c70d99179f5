// Probe: does a 16-byte LDS-DMA (buffer_load ... lds, dwordx4) accept (a) an LDS destination base that is only 4-byte aligned,
// (b) a global source offset that is only 4-byte aligned?   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/lds_dma_align_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void probe(const float* src, float* out, int lds_shift_bytes, int glob_shift_bytes) {
    __shared__ float buf[64 * 4 + 16];
    for (int i = threadIdx.x; i < 64 * 4 + 16; i += 64) buf[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 4096, 0x00020000);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)buf + (unsigned)lds_shift_bytes);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(uintptr_t)base, 16, threadIdx.x * 16 + glob_shift_bytes, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4 + 16; i += 64) out[i] = buf[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int t = 0; t < 4; ++t) {
        const int ls = (t & 1) ? 4 : 0, gs = (t & 2) ? 4 : 0;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, ls, gs);
        if (hipDeviceSynchronize() != hipSuccess) { printf("lds+%d glob+%d: launch failed\n", ls, gs); return 1; }
        std::vector<float> r(272);
        hipMemcpy(r.data(), o, 272 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i + ls / 4] != (float)(i + gs / 4);
        printf("lds+%d glob+%d: %d of 256 wrong; first 10:", ls, gs, bad);
        for (int i = 0; i < 10; ++i) printf(" %g", r[i]);
        printf("\n");
    }
    return 0;
}
