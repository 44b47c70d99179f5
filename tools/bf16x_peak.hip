// Scoping experiment ("what comes next"): fp32-accurate products on the bf16 matrix cores.
// An fp32 value is the exact sum of three bf16 values (8 + 8 + 8 mantissa bits), so a . b = sum_ij a_i b_j with every
// partial product exact in fp32: 9 bf16 MFMAs (all pairs) reproduce the fp32 product, 6 (i + j <= 2) drop terms below
// 2^-24 |a||b|.  bf16 MFMA peak is 16x the fp32 MFMA peak: the ceiling would be 16/9 = 1.8x or 16/6 = 2.7x -- IF the matrix
// pipes can be fed (3 planes per operand from LDS) and the chip holds its clock.  This loop measures exactly that: a wave
// owns a 64x64 tile (2x2 accumulators of 32x32), reads 3 planes of A and B fragments per K-step of 16 from LDS with
// ds_read_b128 and issues 4 x NP MFMAs.  Output: bf16 TFLOP/s and the fp32-equivalent rate (bf16 rate / NP).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NP>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int random) {
    __shared__ u32x4 s[2048];      // 32 KB
    for (int i = threadIdx.x; i < 2048; i += 256) s[i] = u32x4{0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    if (random) {      // realistic operands: every mantissa / exponent bit toggles (power, hence clock, depends on the data)
        for (int i = threadIdx.x; i < 2048; i += 256) {
            unsigned h = (unsigned)i * 2654435761u + 12345u;
            u32x4 r;
            for (int q = 0; q < 4; ++q) {
                h = h * 1664525u + 1013904223u;
                r[q] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);      // two bf16 in [0.5, 2) with random signs
            }
            s[i] = r;
        }
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const u32x4* p = s + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if (NP == 1 && pl > 0) continue;
                if (NP == 3 && pl > 1) continue;
                a[m][pl] = __builtin_bit_cast(bf16x8, p[((m * 3 + pl) * 64 + it * 64) & 1984]);
                b[m][pl] = __builtin_bit_cast(bf16x8, p[((m * 3 + pl + 6) * 64 + it * 64) & 1984]);
            }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (NP == 1 && (i || j)) continue;
                        if (NP == 3 && i + j > 1) continue;
                        if (NP == 6 && i + j > 2) continue;
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][i], b[n][j], acc[m][n], 0, 0, 0);
                    }
    }
    float r = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) r += acc[i][j][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NP>
void run(int blocks_per_cu, int random) {
    float* out;
    int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * 4);
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NP><<<nb, 256>>>(out, 10, random);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NP><<<nb, 256>>>(out, iters, random);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)nb * 4 * iters * 4 * NP * 32768.0;
    printf("bf16x%d %s blocks/CU=%d: %.3f ms  %.0f TFLOP/s bf16  = %.1f TFLOP/s fp32-equivalent\n", NP, random ? "random data  " : "constant data", blocks_per_cu, ms,
           fl / ms / 1e9, fl / ms / 1e9 / NP);
    hipFree(out);
}

int main() {
    for (int random = 0; random < 2; ++random) {
        run<1>(2, random);
        run<3>(2, random);
        run<6>(1, random); run<6>(2, random);
        run<9>(1, random); run<9>(2, random);
    }
    return 0;
}
