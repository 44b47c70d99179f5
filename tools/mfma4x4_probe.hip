// Operand / result lane map of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4x1 blocks per instruction), probed with exact integer
// data: lane l supplies a = 100 + l and b = 1000 + l ... one-hot operands reveal which (A lane, B lane) pair lands in which
// (lane, register) of D.   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/mfma4x4_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out, int la, int lb) {
    const int l = threadIdx.x;
    const float a = l == la ? 1.f : 0.f, b = l == lb ? 1.f : 0.f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void rate(float* out, unsigned long long* cyc, int n) {
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    const float a = (float)threadIdx.x * 0.001f, b = 1.0f + (float)threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    {
        float* o;
        unsigned long long* c;
        hipMalloc(&o, 64 * 4);
        hipMalloc(&c, 8);
        for (int waves = 1; waves <= 2; ++waves) {      // one / two waves on a SIMD (a 64- / 512-thread workgroup would differ: use 1 block of 64 or 8 x 64)
            hipLaunchKernelGGL(rate, dim3(1), dim3(64 * (waves == 1 ? 1 : 8)), 0, 0, o, c, 10000);
            unsigned long long h = 0;
            hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            printf("4x4x1 x 16 blocks: %d wave(s) per SIMD: %.2f cycles per MFMA (40000 MFMAs per wave, 4 independent accumulators)\n", waves, (double)h / 40000.0);
        }
    }
    float* d;
    hipMalloc(&d, 64 * 4 * 4);
    float h[256];
    // for every A lane la and B lane lb of the SAME block guess (la / 4 == lb / 4) report where the product lands
    int shown = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, la, lb);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; ++i)
                if (h[i] != 0.f) {
                    if (la < 8 && lb < 8) printf("A lane %2d x B lane %2d -> D lane %2d reg %d\n", la, lb, i / 4, i % 4);
                    const int pl = (la / 4) * 4 + (lb % 4), pr = la % 4;      // prediction: block = lane / 4, D[i][j] in lane 4 blk + j, reg i
                    if (la / 4 == lb / 4 && (i / 4 != pl || i % 4 != pr)) { if (shown++ < 10) printf("MISMATCH la %d lb %d -> lane %d reg %d (predicted %d %d)\n", la, lb, i / 4, i % 4, pl, pr); }
                    if (la / 4 != lb / 4 && shown++ < 10) printf("CROSS-BLOCK product la %d lb %d -> lane %d reg %d\n", la, lb, i / 4, i % 4);
                }
        }
    printf("done (%d anomalies)\n", shown);
    return 0;
}
