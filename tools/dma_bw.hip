// Calibration: throughput of the global -> LDS DMA path (buffer_load ... lds, 4 and 16 bytes per lane) against
// plain 16-byte loads into registers, on L2-resident and HBM-resident footprints.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_bw.hip -o /tmp/dma_bw && /tmp/dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, const float* lds, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, off, 0, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t r, const float* lds, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 4, off, 0, 0, 0);
}

// MODE 0: dword DMA, 1: dwordx4 DMA, 2: dwordx4 into registers.  Each workgroup streams `per_wg` bytes
// (a private region, or the same region again and again when `stride_wg` == 0 -> L2 hits) per pass.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* src, float* sink, unsigned per_wg, unsigned long long stride_wg, int passes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // 32 KB
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* base = src + (unsigned long long)blockIdx.x * stride_wg / 4;
    const __amdgpu_buffer_rsrc_t r = rsrc(base, per_wg);
    f32x4 accv = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        for (unsigned off = 0; off < per_wg; off += 32768) {        // 32 KB per step = 8 x (256 lanes x 16 B)
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 32; ++i) dma4(r, smem + i * 256 + wave * 64, off + (i * 256 + tid) * 4);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dma16(r, smem + (i * 256 + wave * 64) * 4, off + (i * 256 + tid) * 16);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(base + (off + (i * 256 + tid) * 16) / 4);
                    accv += v;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (MODE == 2) sink[blockIdx.x * 256 + tid] = accv[0] + accv[1] + accv[2] + accv[3];
    else sink[blockIdx.x * 256 + tid] = smem[tid];
}

template <int MODE>
void run(const char* name, const float* src, float* sink, unsigned per_wg, unsigned long long stride, int passes) {
    const int nb = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, 256, 32768>>>(src, sink, per_wg, stride, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<nb, 256, 32768>>>(src, sink, per_wg, stride, passes);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nb * per_wg * passes;
    printf("%-22s per_wg %7u KB stride %8llu KB: %8.3f ms  %7.2f TB/s  (%.1f B/clk/CU at 2.2 GHz)\n", name, per_wg >> 10,
           stride >> 10, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.2e9 / 1e3 * 1e3 / 1e3);
}

int main() {
    float *src, *sink;
    const size_t total = 512ull * 4 * 1024 * 1024;      // 2 GB
    hipMalloc(&src, total);
    hipMemset(src, 0, total);
    hipMalloc(&sink, 512 * 256 * 4);
    // L2-resident: every workgroup re-reads the same 64 KB
    run<0>("dma dword   (L2)", src, sink, 65536, 0, 200);
    run<1>("dma dwordx4 (L2)", src, sink, 65536, 0, 200);
    run<2>("vgpr dwordx4 (L2)", src, sink, 65536, 0, 200);
    // private 64 KB per workgroup, re-read (32 MB total: L2 / MALL)
    run<0>("dma dword   (64K/wg)", src, sink, 65536, 65536, 200);
    run<1>("dma dwordx4 (64K/wg)", src, sink, 65536, 65536, 200);
    run<2>("vgpr dwordx4 (64K/wg)", src, sink, 65536, 65536, 200);
    // HBM stream: 4 MB per workgroup, one pass
    run<0>("dma dword   (HBM)", src, sink, 4u << 20, 4ull << 20, 1);
    run<1>("dma dwordx4 (HBM)", src, sink, 4u << 20, 4ull << 20, 1);
    run<2>("vgpr dwordx4 (HBM)", src, sink, 4u << 20, 4ull << 20, 1);
    return 0;
}
