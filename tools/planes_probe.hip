// Development probe (GPU only): what bounds a kernel that streams MANY planes at the same pixel offset (pred_bce_kernel: 32 + 12 planes read,
// 12 + 32 written, 256 KB apart at 256^2)?  One thread owns 4 consecutive pixels of every plane, as pred_bce does; grid-stride over the image.
//   hipcc -O3 --offload-arch=gfx950 tools/planes_probe.hip -o /tmp/planes_probe && /tmp/planes_probe
// Variants: number of planes read / written, plane pitch (exact HW or padded), non-temporal accesses, loads batched NB at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NB, bool NT>
__global__ __launch_bounds__(256) void planes_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Wn, long long pitch_in4, long long pitch_out4,
                                                     long long img_in4, long long img_out4, long long hw4, int B) {
    const long long total = (long long)B * hw4;
    for (long long q = blockIdx.x * 256ll + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int b = (int)(q / hw4);
        const long long p = q - (long long)b * hw4;
        const f4* xp = reinterpret_cast<const f4*>(src) + b * img_in4 + p;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < R; c0 += NB) {
            f4 v[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int c = c0 + k < R ? c0 + k : R - 1;
                v[k] = NT ? __builtin_nontemporal_load(xp + c * pitch_in4) : xp[c * pitch_in4];
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) acc += v[k] * (float)(c0 + k + 1);
        }
        f4* yp = reinterpret_cast<f4*>(dst) + b * img_out4 + p;
#pragma unroll 2
        for (int c = 0; c < Wn; ++c) {
            const f4 o = acc * (float)(c + 1);
            if (NT) __builtin_nontemporal_store(o, yp + c * pitch_out4);
            else yp[c * pitch_out4] = o;
        }
    }
}

int main() {
    const int B = 32, H = 256, W = 256;
    const long long HW = (long long)H * W;
    const int maxp = 48;
    const long long pad = 1024 + 64;      // floats
    float *src, *dst;
    hipMalloc(&src, sizeof(float) * B * maxp * (HW + pad));
    hipMalloc(&dst, sizeof(float) * B * maxp * (HW + pad));
    hipMemset(src, 0, sizeof(float) * B * maxp * (HW + pad));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Case { int R, Wn; };
    const Case cases[] = {{1, 1}, {4, 4}, {8, 8}, {16, 16}, {32, 32}, {44, 44}, {44, 12}, {12, 44}, {44, 0}, {0, 44}};
    for (int padded = 0; padded < 2; ++padded)
        for (int nt = 0; nt < 2; ++nt)
            for (int grid : {1024, 4096})
                for (const Case& c : cases) {
                    const long long pitch = HW + (padded ? pad : 0);
                    auto launch = [&]() {
                        const long long pi = pitch / 4, ii = (long long)(c.R ? c.R : 1) * pitch / 4, io = (long long)(c.Wn ? c.Wn : 1) * pitch / 4;
                        if (nt) hipLaunchKernelGGL((planes_kernel<8, true>), dim3(grid), dim3(256), 0, 0, src, dst, c.R, c.Wn, pi, pi, ii, io, HW / 4, B);
                        else hipLaunchKernelGGL((planes_kernel<8, false>), dim3(grid), dim3(256), 0, 0, src, dst, c.R, c.Wn, pi, pi, ii, io, HW / 4, B);
                    };
                    for (int i = 0; i < 2; ++i) launch();
                    hipEventRecord(e0);
                    for (int i = 0; i < 5; ++i) launch();
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms = 0.f;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double us = ms * 1e3 / 5, bytes = (double)(c.R + c.Wn) * B * HW * 4;
                    printf("pad %d nt %d grid %4d  read %2d write %2d planes: %8.1f us  %6.2f TB/s\n", padded, nt, grid, c.R, c.Wn, us, bytes / us / 1e6);
                }
    return 0;
}
