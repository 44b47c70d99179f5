// Calibration: issue cost of N back-to-back `buffer_load_dwordx4 ... lds` (LDS-DMA) instructions of ONE wave against
// N plain 16-byte loads into registers: cycles until the last one has issued, and until all have completed.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_issue.hip -o tools/dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int MODE>
__global__ void k(const float* src, unsigned long long* out, float* sink, unsigned stride_bytes, int waves) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * (MODE >= 2 ? 0 : (stride_bytes / 4) * N * waves) + (MODE >= 2 ? (size_t)(blockIdx.x % 64) * 2359296 : 0)), 0, MODE >= 2 ? 9437184u : stride_bytes * N * waves, 0x00020000);
    f32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = 0;
    if (wave < waves) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            unsigned off = (unsigned)((wave * N + i) * stride_bytes + lane * 16);
            if (MODE == 2) {
                // the x-tile pattern of the conv / wgrad kernels: quads of rows of 10 quads (160 B), row pitch 1 KB,
                // 41 quads per channel (4 rows + pad), channel pitch 256 KB, tile origin 16 B before a 128-B boundary
                const int q = (wave * N + i) * 64 + lane;
                const int ch = q / 41, within = q % 41, row = within / 10, col = within % 10;
                off = (unsigned)(ch * 262144 + row * 1024 + col * 16 + 112);
            }
            if (MODE == 3) {
                // the same tile with the lanes permuted so that every quad of lanes reads one aligned 64-byte line:
                // per channel 4 rows x 8 aligned quads, then the 8 edge quads (column 0 / 9 of each row)
                const int q = (wave * N + i) * 64 + lane;
                const int ch = q / 40, within = q % 40;
                const int row = within < 32 ? within / 8 : (within - 32) / 2;
                const int col = within < 32 ? 1 + within % 8 : ((within & 1) ? 9 : 0);
                off = (unsigned)(ch * 262144 + row * 1024 + col * 16 + 112);
            }
            if (MODE == 4) {
                // rows widened to aligned 256-byte spans (16 quads), 4 rows per instruction
                const int q = (wave * N + i) * 64 + lane;
                const int ch = q / 64, within = q % 64;
                off = (unsigned)(ch * 262144 + (within / 16) * 1024 + (within % 16) * 16 + 64);
            }
            if (MODE == 0 || MODE >= 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(smem + (wave * N + i) * 256), 16, off, 0, 0, 0);
            else {
                const unsigned u0 = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
                acc[i & 3] += __builtin_bit_cast(float, u0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = t2 - t0;
    }
    sink[blockIdx.x * blockDim.x + tid] = acc[0] + acc[1] + acc[2] + acc[3] + smem[tid];
}

template <int N, int MODE>
void run(const char* name, const float* src, unsigned stride, int waves, bool warm) {
    unsigned long long* out;
    float* sink;
    const int nb = 256;
    hipMalloc(&out, nb * 16);
    hipMalloc(&sink, nb * 256 * 4);
    if (warm) { k<N, MODE><<<nb, 256, 65536>>>(src, out, sink, stride, waves); hipDeviceSynchronize(); }
    k<N, MODE><<<nb, 256, 65536>>>(src, out, sink, stride, waves);
    hipDeviceSynchronize();
    unsigned long long h[nb * 2];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < nb; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    printf("%-10s N=%2d waves=%d %s: issued after %7.0f cycles (%6.1f per instr), complete after %7.0f\n", name, N, waves,
           warm ? "L2-warm" : "cold   ", a / nb, a / nb / N, b / nb);
    hipFree(out); hipFree(sink);
}

int main() {
    float* src;
    hipMalloc(&src, 1ull << 30);
    hipMemset(src, 0, 1ull << 30);
    for (int warm = 1; warm >= 0; --warm) {
        run<1, 0>("lds-dma", src, 1024, 1, warm);
        run<4, 0>("lds-dma", src, 1024, 1, warm);
        run<12, 0>("lds-dma", src, 1024, 1, warm);
        run<12, 0>("lds-dma", src, 1024, 4, warm);
        run<12, 2>("lds-dma-tile", src, 1024, 1, warm);
        run<12, 2>("lds-dma-tile", src, 1024, 4, warm);
        run<5, 2>("lds-dma-tile", src, 1024, 4, warm);
        run<12, 3>("tile-perm", src, 1024, 1, warm);
        run<12, 3>("tile-perm", src, 1024, 4, warm);
        run<5, 3>("tile-perm", src, 1024, 4, warm);
        run<12, 4>("tile-wide", src, 1024, 1, warm);
        run<12, 4>("tile-wide", src, 1024, 4, warm);
        run<1, 1>("vgpr", src, 1024, 1, warm);
        run<4, 1>("vgpr", src, 1024, 1, warm);
        run<12, 1>("vgpr", src, 1024, 1, warm);
        run<12, 1>("vgpr", src, 1024, 4, warm);
    }
    return 0;
}
