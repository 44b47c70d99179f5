// Calibration: fp32 MFMA (32x32x2) rate on this device for w waves/SIMD, with and without LDS operand reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const float* p = s + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (LDS) { a = p[(j * 64 + it * 8) & 4032]; }
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (LDS) b = p[((j * NACC + i) * 64 + it) & 4032];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
        }
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) r += acc[i][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NACC, bool LDS>
void run(int blocks_per_cu, const char* name) {
    float* out;
    int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDS><<<nb, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDS><<<nb, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)nb * 4 * iters * 8 * NACC * 4096.0;
    printf("%s nacc=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", name, NACC, blocks_per_cu, ms, fl / ms / 1e9);
    hipFree(out);
}

int main() {
    run<4, false>(1, "reg"); run<4, false>(2, "reg"); run<4, false>(3, "reg");
    run<4, true>(1, "lds"); run<4, true>(2, "lds"); run<4, true>(3, "lds");
    run<8, true>(2, "lds"); run<1, true>(4, "lds"); run<2, true>(3, "lds");
    return 0;
}
