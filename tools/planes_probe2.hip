// Development probe 2 (GPU only): pred_bce_kernel's shape -- 32 planes read, a 32 -> 12 product per pixel, 12 planes written + 12 read (target), a 12 -> 32 product,
// 32 planes written -- with the vector work switched on / off and the channel loop scheduled in different ways.  Does the launch take max(memory, vector) or their sum?
//   hipcc -O3 --offload-arch=gfx950 tools/planes_probe2.hip -o /tmp/planes_probe2 && /tmp/planes_probe2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* cf32;

// MODE 0: loads in batches of NB, products after each batch (the product kernel's loop, unroll 1)
// MODE 1: the same loop fully unrolled (the scheduler may hoist loads as far as the register cap allows)
// MODE 2: explicit double buffer: batch k + 1 is issued before batch k is multiplied
template <int MODE, int NB, bool VALU, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ y, float* __restrict__ dx, const float* __restrict__ wq,
                                               long long hw4, int B) {
    constexpr int CIN = 32, CT = 12;
    const cf32 w = (cf32)wq;
    const long long total = (long long)B * hw4;
    for (long long q = blockIdx.x * 256ll + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int b = (int)(q / hw4);
        const long long p = q - (long long)b * hw4;
        const f4* xp = reinterpret_cast<const f4*>(x) + (long long)b * CIN * hw4 + p;
        f4 acc[CT];
#pragma unroll
        for (int co = 0; co < CT; ++co) acc[co] = f4{0.f, 0.f, 0.f, 0.f};
        auto mul = [&](const f4 (&v)[NB], int c0) {
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) {
                if (VALU) {
#pragma unroll
                    for (int co = 0; co < CT; ++co) acc[co] += v[kk] * w[(c0 + kk) * 64 + co];
                } else {
                    acc[0] += v[kk];
                }
            }
        };
        if (MODE == 0) {
#pragma unroll 1
            for (int c0 = 0; c0 < CIN; c0 += NB) {
                f4 v[NB];
#pragma unroll
                for (int kk = 0; kk < NB; ++kk) v[kk] = xp[(long long)(c0 + kk) * hw4];
                mul(v, c0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int c0 = 0; c0 < CIN; c0 += NB) {
                f4 v[NB];
#pragma unroll
                for (int kk = 0; kk < NB; ++kk) v[kk] = xp[(long long)(c0 + kk) * hw4];
                mul(v, c0);
            }
        } else {
            f4 va[NB], vb[NB];
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) va[kk] = xp[(long long)kk * hw4];
#pragma unroll
            for (int c0 = 0; c0 < CIN; c0 += 2 * NB) {
#pragma unroll
                for (int kk = 0; kk < NB; ++kk) vb[kk] = xp[(long long)(c0 + NB + kk) * hw4];
                __builtin_amdgcn_sched_barrier(0);
                mul(va, c0);
                __builtin_amdgcn_sched_barrier(0);
                if (c0 + 2 * NB < CIN) {
#pragma unroll
                    for (int kk = 0; kk < NB; ++kk) va[kk] = xp[(long long)(c0 + 2 * NB + kk) * hw4];
                }
                __builtin_amdgcn_sched_barrier(0);
                mul(vb, c0 + NB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const long long ob = (long long)b * CT * hw4 + p;
#pragma unroll
        for (int co = 0; co < CT; ++co) {
            const f4 tv = reinterpret_cast<const f4*>(t)[ob + co * hw4];
            reinterpret_cast<f4*>(y)[ob + co * hw4] = acc[co];
            f4 d = acc[co] - tv;
            if (VALU) {      // (a stand-in for the criterion's ~25 operations per element)
#pragma unroll
                for (int r = 0; r < 6; ++r) d = d * d + tv;
            }
            acc[co] = d;
        }
        f4* dp = reinterpret_cast<f4*>(dx) + (long long)b * CIN * hw4 + p;
#pragma unroll 2
        for (int ci = 0; ci < CIN; ++ci) {
            f4 o = {0.f, 0.f, 0.f, 0.f};
            if (VALU) {
#pragma unroll
                for (int co = 0; co < CT; ++co) o += acc[co] * w[ci * 64 + co];
            } else {
                o = acc[0] + (float)ci;
            }
            dp[(long long)ci * hw4] = o;
        }
    }
}

int main() {
    const int B = 32, H = 256, W = 256;
    const long long HW = (long long)H * W;
    float *x, *t, *y, *dx, *w;
    hipMalloc(&x, sizeof(float) * B * 32 * HW);
    hipMalloc(&dx, sizeof(float) * B * 32 * HW);
    hipMalloc(&t, sizeof(float) * B * 12 * HW);
    hipMalloc(&y, sizeof(float) * B * 12 * HW);
    hipMalloc(&w, sizeof(float) * 64 * 64);
    hipMemset(x, 0, sizeof(float) * B * 32 * HW);
    hipMemset(t, 0, sizeof(float) * B * 12 * HW);
    hipMemset(w, 0, sizeof(float) * 64 * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = 88.0 * B * HW * 4;
    auto run = [&](const char* name, auto kern, int grid) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, x, t, y, dx, w, HW / 4, B);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, x, t, y, dx, w, HW / 4, B);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s grid %5d: %7.1f us  %5.2f TB/s\n", name, grid, ms * 1e3 / 5, bytes / (ms * 1e3 / 5) / 1e6);
    };
    for (int grid : {1024, 2048}) {
        run("loop NB 8, no vector work, 4 waves/SIMD", k<0, 8, false, 4>, grid);
        run("loop NB 8, vector work, 4 waves/SIMD", k<0, 8, true, 4>, grid);
        run("unrolled NB 8, vector work, 4 waves/SIMD", k<1, 8, true, 4>, grid);
        run("unrolled NB 4, vector work, 4 waves/SIMD", k<1, 4, true, 4>, grid);
        run("unrolled NB 8, vector work, 3 waves/SIMD", k<1, 8, true, 3>, grid);
        run("double buffer NB 8, vector work, 3 waves/SIMD", k<2, 8, true, 3>, grid);
        run("double buffer NB 4, vector work, 4 waves/SIMD", k<2, 4, true, 4>, grid);
        run("double buffer NB 4, vector work, 3 waves/SIMD", k<2, 4, true, 3>, grid);
        run("double buffer NB 8, no vector work, 3 waves/SIMD", k<2, 8, false, 3>, grid);
    }
    return 0;
}
