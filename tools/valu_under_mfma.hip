// Calibration: how fast does a wave's scalar / vector-ALU / LDS-read code issue while the other waves of its
// SIMD keep the fp32 matrix pipe busy?   hipcc --offload-arch=gfx950 -O3 tools/valu_under_mfma.hip -o tools/valu_under_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// blockDim = 256 * nw: waves 0..3 land on SIMD 0..3, waves 4..7 again ...; wave "slot" = wave / 4.
// slot 0 = probe wave (runs `kind` code, timed), slots >= 1 = MFMA hogs (when hog != 0).
template <int KIND>
__global__ void k(unsigned long long* out, float* sink, int iters, int hog) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, slot = wave >> 2;
    lds[threadIdx.x & 4095] = threadIdx.x;
    __syncthreads();
    if (slot == 0) {
        float a = threadIdx.x, b = 1.0001f;
        int s = blockIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) {          // 64 dependent VALU ops
#pragma unroll
                for (int j = 0; j < 64; ++j) a = a * b + 0.5f;
            } else if (KIND == 1) {   // 64 independent-ish VALU ops (4 chains)
                float c = a + 1, d = a + 2, e = a + 3;
#pragma unroll
                for (int j = 0; j < 16; ++j) { a = a * b + 0.5f; c = c * b + 0.5f; d = d * b + 0.5f; e = e * b + 0.5f; }
                a += c + d + e;
            } else if (KIND == 2) {   // 64 dependent SALU ops
#pragma unroll
                for (int j = 0; j < 64; ++j) s = s * 3 + 1;
            } else {                  // 16 LDS reads, each waited for
#pragma unroll
                for (int j = 0; j < 16; ++j) a += lds[((int)a + j * 64 + (threadIdx.x & 63)) & 4095];
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a + s;
    } else if (hog) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        const float x = threadIdx.x * 1e-3f, y = 1.f;
        for (int i = 0; i < iters * 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[j], 0, 0, 0);
        }
        float r = 0;
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    }
}

template <int KIND>
void run(const char* name, int nslots, int hog, int per_iter) {
    unsigned long long* out;
    float* sink;
    const int nb = 256, threads = 256 * nslots, iters = 200;
    hipMalloc(&out, nb * 4 * 8);
    hipMalloc(&sink, (size_t)nb * threads * 4);
    k<KIND><<<nb, threads>>>(out, sink, 10, hog);
    hipDeviceSynchronize();
    k<KIND><<<nb, threads>>>(out, sink, iters, hog);
    hipDeviceSynchronize();
    unsigned long long h[nb * 4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb * 4; ++i) s += h[i];
    printf("%-28s probe + %d MFMA waves/SIMD: %7.1f cycles per op\n", name, hog ? nslots - 1 : 0, s / (nb * 4) / iters / per_iter);
    hipFree(out);
    hipFree(sink);
}

int main() {
    for (int hogs = 0; hogs <= 3; ++hogs) {
        run<0>("VALU dependent chain", hogs + 1, hogs, 64);
        run<1>("VALU 4 chains", hogs + 1, hogs, 64);
        run<2>("SALU dependent chain", hogs + 1, hogs, 64);
        run<3>("LDS read + wait", hogs + 1, hogs, 16);
    }
    return 0;
}
