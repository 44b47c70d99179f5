// PROTOTYPE (GPU only, not part of the library): the 3x3 filter gradient in the Winograd F(2x2, 3x3) domain on the fp32 matrix cores.
//   Y = A^T [ U (.) V ] A with U = G g G^T, V = B^T d B   =>   dU[xi,nu][co][ci] = sum over tiles (A dY A^T)[xi,nu][co] * V[xi,nu][ci],   dg = G^T dU G
// i.e. 16 multiplies per 2x2 output block and channel pair instead of the direct form's 36 -- the K dimension of the MFMAs is the TILE index, and
// between the MFMAs there is nothing but the two transforms.  Measures what DESIGN.md section 8 item 2a prices against wgrad_roll_kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/wgrad_wino_proto.hip -o /tmp/wgrad_wino_proto && /tmp/wgrad_wino_proto
// Layout (version 3): a workgroup of 8 waves = 4 blocks (16 co x 16 ci) x 2 tile groups; a STAGE is FOUR output rows x 32 columns -- two vertically adjacent row pairs, one
// per tile group, sharing two of their input rows: six input rows + four rows of the output gradient = 52 KB, three stages in flight; all waves issue the LDS-DMAs of the stage two
// ahead (rows as [row][channel] groups with a pitch of an odd number of 16-byte units: conflict-free 8-byte reads for lanes = (channel, tile)), one barrier per stage.
// Measured (MI355X, B 32, 32 -> 32): 74 us at 128^2, 282 us at 256^2 = 130-137 TFLOP/s direct-equivalent (wgrad_roll_kernel: 100-119); version 1 (two stages, both operands per
// unit): 114 / 434 us.  The launch is below the ridge (13-16 executed FLOP per byte): its floor is HBM time.  DESIGN.md section 8, item 2a.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int CI = 32, CO = 32;
constexpr int PXB = 176, PDB = 144;                       // pitches of a (row, channel) group: 11 / 9 units of 16 bytes (an odd count = conflict-free 8-byte reads)
constexpr int XBYTES = 6 * CI * PXB + 16, DBYTES = 4 * CO * PDB;
constexpr int STAGE_BYTES = XBYTES + DBYTES;              // a stage: FOUR output rows x 32 columns = six input rows + four rows of the output gradient
constexpr int NBUF = 3;
constexpr int NX = (6 * CI + 4) / 5, ND = (4 * CO + 6) / 7;   // DMA instructions per stage: 5 / 7 groups each
constexpr int NDMA = NX + ND;

struct Args {
    const float* x;     // [B][CI][H][W]
    const float* dy;    // [B][CO][H][W]
    float* part;        // [gridDim.x][16][CO][CI]
    int B, H, W;
};

// version 3: a stage is two vertically adjacent row pairs (they share two of their four input rows), everything arrives by LDS-DMA, three stages in flight,
// one barrier per stage; the two tile groups of the workgroup take one row pair each.
__global__ __launch_bounds__(512, 1) void wgrad_wino_kernel(const Args a) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int H = a.H, W = a.W, HW = H * W;
    const int upr = W / 32, upi = (H / 4) * upr;            // stages per row of stages, per image
    const int stages = a.B * upi;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned lead = (unsigned)((W + 4) * 4);
    const unsigned x_img = (unsigned)(CI * HW * 4) + lead, d_img = (unsigned)(CO * HW * 4);

    const int gx = lane / 11, ux = lane - gx * 11, gd = lane / 9, ud = lane - gd * 9;
    auto dma_stage = [&](int stage, int buf) {
        const int b = stage / upi, rem = stage - b * upi, p = rem / upr, tx = rem - p * upr;
        const int y0 = 4 * p, x0 = 32 * tx;
        const unsigned ub = lds0 + (unsigned)(buf * STAGE_BYTES);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x + (long long)b * CI * HW) - lead), 0, x_img, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (long long)b * CO * HW), 0, d_img, 0x00020000);
        for (int i = wave; i < NDMA; i += 8) {
            if (i < NX) {
                const int grp = 5 * i + gx;
                if (lane < 55 && grp < 6 * CI) {
                    const int r = grp / CI, c = grp - r * CI;
                    const int yy = y0 - 1 + r;
                    const bool zero = ux == 10 || yy < 0 || yy >= H || (ux == 0 && x0 == 0) || (ux == 9 && x0 + 32 == W);
                    const unsigned off = zero ? 0x80000000u : (unsigned)((c * HW + yy * W + x0 - 4 + 4 * ux) * 4) + lead;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(ub + 4u + (unsigned)(5 * i * PXB)), 16, off, 0, 0, 0);
                }
            } else {
                const int jd = i - NX, grp = 7 * jd + gd;
                if (lane < 63 && grp < 4 * CO) {
                    const int r = grp / CO, c = grp - r * CO;
                    const unsigned off = ud == 8 ? 0x80000000u : (unsigned)((c * HW + (y0 + r) * W + x0 + 4 * ud) * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(uintptr_t)(ub + (unsigned)XBYTES + (unsigned)(7 * jd * PDB)), 16, off, 0, 0, 0);
                }
            }
        }
    };
    const int ndma = (NDMA - wave + 7) / 8;                 // how many of a stage's instructions this wave issues

    const int grp2 = wave >> 2, bo = (wave >> 1) & 1, bi = wave & 1, ch = lane & 15, tk = lane >> 4;
    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int G = gridDim.x;
    int s = blockIdx.x, it = 0;
    if (s < stages) dma_stage(s, 0);
    if (s + G < stages) dma_stage(s + G, 1);
    for (; s < stages; s += G, ++it) {
        const int buf = it % NBUF;
        // stage s has landed: what may still be in flight is this wave's share of stage s + G (issued after it)
        if (s + G < stages) {
            if (ndma == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        if (s + 2 * G < stages) dma_stage(s + 2 * G, (it + 2) % NBUF);
        __builtin_amdgcn_sched_barrier(0);      // (the fetches are ISSUED here, not after the products hipcc would rather start with)
        const unsigned char* ub = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int t = 4 * st + tk;
            const unsigned char* xp = ub + 4 + ((2 * grp2) * CI + bi * 16 + ch) * PXB + (2 * t + 3) * 4;
            f32x2 dl[4], dh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dl[r] = *reinterpret_cast<const f32x2*>(xp + r * CI * PXB);
                dh[r] = *reinterpret_cast<const f32x2*>(xp + r * CI * PXB + 8);
            }
            const unsigned char* dp = ub + XBYTES + ((2 * grp2) * CO + bo * 16 + ch) * PDB + (2 * t) * 4;
            const f32x2 e0 = *reinterpret_cast<const f32x2*>(dp), e1 = *reinterpret_cast<const f32x2*>(dp + CO * PDB);
            // V = B^T d B
            const f32x2 tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[1] - dl[3]};
            const f32x2 th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[1] - dh[3]};
            float V[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                V[i][0] = tl[i][0] - th[i][0];
                V[i][1] = tl[i][1] + th[i][0];
                V[i][2] = th[i][0] - tl[i][1];
                V[i][3] = tl[i][1] - th[i][1];
            }
            // Z = A dY A^T
            const f32x2 z[4] = {e0, e0 + e1, e0 - e1, -e1};
            float Z[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Z[i][0] = z[i][0];
                Z[i][1] = z[i][0] + z[i][1];
                Z[i][2] = z[i][0] - z[i][1];
                Z[i][3] = -z[i][1];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Z[i][j], V[i][j], acc[4 * i + j], 0, 0, 0);
        }
    }
    // the two tile groups of a block: group 1 hands its sums over through LDS, group 0 adds and writes the workgroup's partial
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* ex = reinterpret_cast<float*>(smem);
    if (grp2 == 1) {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) ex[(((bo * 2 + bi) * 16 + p) * 4 + r) * 64 + lane] = acc[p][r];
    }
    __syncthreads();
    if (grp2 == 0) {
        float* out = a.part + (long long)blockIdx.x * 16 * CO * CI;
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[p][r] + ex[(((bo * 2 + bi) * 16 + p) * 4 + r) * 64 + lane];
                out[(p * CO + bo * 16 + 4 * tk + r) * CI + bi * 16 + ch] = v;
            }
    }
}

// ---- 16 input channels (the first layer: 14 -> 32 at 256^2): 2 blocks x 4 tile groups (row pair x column half), four stages in flight
namespace c16 {
constexpr int CI2 = 16;
constexpr int XB2 = 6 * CI2 * PXB + 16, STAGE2 = XB2 + DBYTES, NBUF2 = 4;
constexpr int NX2 = (6 * CI2 + 4) / 5, NDMA2 = NX2 + ND;
}
__global__ __launch_bounds__(512, 1) void wgrad_wino16_kernel(const Args a) {
    using namespace c16;
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int H = a.H, W = a.W, HW = H * W;
    const int upr = W / 32, upi = (H / 4) * upr;
    const int stages = a.B * upi;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned lead = (unsigned)((W + 4) * 4);
    const unsigned x_img = (unsigned)(CI2 * HW * 4) + lead, d_img = (unsigned)(CO * HW * 4);
    const int gx = lane / 11, ux = lane - gx * 11, gd = lane / 9, ud = lane - gd * 9;
    auto dma_stage = [&](int stage, int buf) {
        const int b = stage / upi, rem = stage - b * upi, p = rem / upr, tx = rem - p * upr;
        const int y0 = 4 * p, x0 = 32 * tx;
        const unsigned ub = lds0 + (unsigned)(buf * STAGE2);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x + (long long)b * CI2 * HW) - lead), 0, x_img, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (long long)b * CO * HW), 0, d_img, 0x00020000);
        for (int i = wave; i < NDMA2; i += 8) {
            if (i < NX2) {
                const int grp = 5 * i + gx;
                if (lane < 55 && grp < 6 * CI2) {
                    const int r = grp / CI2, c = grp - r * CI2;
                    const int yy = y0 - 1 + r;
                    const bool zero = ux == 10 || yy < 0 || yy >= H || (ux == 0 && x0 == 0) || (ux == 9 && x0 + 32 == W);
                    const unsigned off = zero ? 0x80000000u : (unsigned)((c * HW + yy * W + x0 - 4 + 4 * ux) * 4) + lead;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(ub + 4u + (unsigned)(5 * i * PXB)), 16, off, 0, 0, 0);
                }
            } else {
                const int jd = i - NX2, grp = 7 * jd + gd;
                if (lane < 63 && grp < 4 * CO) {
                    const int r = grp / CO, c = grp - r * CO;
                    const unsigned off = ud == 8 ? 0x80000000u : (unsigned)((c * HW + (y0 + r) * W + x0 + 4 * ud) * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(uintptr_t)(ub + (unsigned)XB2 + (unsigned)(7 * jd * PDB)), 16, off, 0, 0, 0);
                }
            }
        }
    };
    const int ndma = (NDMA2 - wave + 7) / 8;      // 39 instructions: waves 0..6 issue 5, wave 7 issues 4
    const int bo = wave & 1, g = wave >> 1, rp = g >> 1, half = g & 1, ch = lane & 15, tk = lane >> 4;
    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int G = gridDim.x;
    int s = blockIdx.x, it = 0;
    if (s < stages) dma_stage(s, 0);
    if (s + G < stages) dma_stage(s + G, 1);
    if (s + 2 * G < stages) dma_stage(s + 2 * G, 2);
    for (; s < stages; s += G, ++it) {
        const int buf = it % NBUF2;
        // stage s has landed: up to two younger stages of this wave's instructions may be in flight
        const int younger = (s + 2 * G < stages ? 2 : (s + G < stages ? 1 : 0)) * ndma;
        if (younger >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (s + 3 * G < stages) dma_stage(s + 3 * G, (it + 3) % NBUF2);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ub = smem + buf * STAGE2;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int t = 8 * half + 4 * st + tk;
            const unsigned char* xp = ub + 4 + ((2 * rp) * CI2 + ch) * PXB + (2 * t + 3) * 4;
            f32x2 dl[4], dh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dl[r] = *reinterpret_cast<const f32x2*>(xp + r * CI2 * PXB);
                dh[r] = *reinterpret_cast<const f32x2*>(xp + r * CI2 * PXB + 8);
            }
            const unsigned char* dp = ub + XB2 + ((2 * rp) * CO + bo * 16 + ch) * PDB + (2 * t) * 4;
            const f32x2 e0 = *reinterpret_cast<const f32x2*>(dp), e1 = *reinterpret_cast<const f32x2*>(dp + CO * PDB);
            const f32x2 tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[1] - dl[3]};
            const f32x2 th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[1] - dh[3]};
            float V[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                V[i][0] = tl[i][0] - th[i][0];
                V[i][1] = tl[i][1] + th[i][0];
                V[i][2] = th[i][0] - tl[i][1];
                V[i][3] = tl[i][1] - th[i][1];
            }
            const f32x2 z[4] = {e0, e0 + e1, e0 - e1, -e1};
            float Z[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Z[i][0] = z[i][0];
                Z[i][1] = z[i][0] + z[i][1];
                Z[i][2] = z[i][0] - z[i][1];
                Z[i][3] = -z[i][1];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Z[i][j], V[i][j], acc[4 * i + j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* ex = reinterpret_cast<float*>(smem);
    if (g > 0) {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) ex[((((g - 1) * 2 + bo) * 16 + p) * 4 + r) * 64 + lane] = acc[p][r];
    }
    __syncthreads();
    if (g == 0) {
        float* out = a.part + (long long)blockIdx.x * 16 * CO * CI2;
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[p][r];
#pragma unroll
                for (int q = 0; q < 3; ++q) v += ex[(((q * 2 + bo) * 16 + p) * 4 + r) * 64 + lane];
                out[(p * CO + bo * 16 + 4 * tk + r) * CI2 + ch] = v;
            }
    }
}

// dU[16][CO][CI] = sum of the partials; dg = G^T dU G
__global__ void finish_kernel(const float* part, int nparts, float* dw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (co, ci)
    if (idx >= CO * CI) return;
    float u[4][4];
    for (int p = 0; p < 16; ++p) {
        float s = 0.f;
        for (int k = 0; k < nparts; ++k) s += part[((long long)k * 16 + p) * CO * CI + idx];
        u[p >> 2][p & 3] = s;
    }
    // G^T (3x4) = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
    float t[3][4];
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        t[1][j] = 0.5f * (u[1][j] - u[2][j]);
        t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
    for (int i = 0; i < 3; ++i) {
        dw[idx * 9 + i * 3 + 0] = t[i][0] + 0.5f * (t[i][1] + t[i][2]);
        dw[idx * 9 + i * 3 + 1] = 0.5f * (t[i][1] - t[i][2]);
        dw[idx * 9 + i * 3 + 2] = 0.5f * (t[i][1] + t[i][2]) + t[i][3];
    }
}

static void reference(const std::vector<float>& x, const std::vector<float>& dy, std::vector<double>& dw, int B, int H, int W) {
    dw.assign((size_t)CO * CI * 9, 0.0);
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < CO; ++co)
            for (int ci = 0; ci < CI; ++ci)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        double s = 0.0;
                        for (int y = 0; y < H; ++y) {
                            const int yy = y + ky - 1;
                            if (yy < 0 || yy >= H) continue;
                            for (int xx0 = 0; xx0 < W; ++xx0) {
                                const int xx = xx0 + kx - 1;
                                if (xx < 0 || xx >= W) continue;
                                s += (double)dy[((size_t)(b * CO + co) * H + y) * W + xx0] * x[((size_t)(b * CI + ci) * H + yy) * W + xx];
                            }
                        }
                        dw[((size_t)co * CI + ci) * 9 + ky * 3 + kx] += s;
                    }
}

int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * STAGE_BYTES);
    printf("LDS per workgroup: %d bytes (stage %d, %d stages)\n", NBUF * STAGE_BYTES, STAGE_BYTES, NBUF);
    // ---- correctness on a small case
    {
        const int B = 2, H = 32, W = 64;
        std::vector<float> x((size_t)B * CI * H * W), dy((size_t)B * CO * H * W);
        srand(1);
        for (auto& v : x) v = (rand() % 2001 - 1000) / 1000.f;
        for (auto& v : dy) v = (rand() % 2001 - 1000) / 1000.f;
        float *dx_, *ddy, *part, *dw;
        const int grid = 8;
        hipMalloc(&dx_, x.size() * 4);
        hipMalloc(&ddy, dy.size() * 4);
        hipMalloc(&part, (size_t)grid * 16 * CO * CI * 4);
        hipMalloc(&dw, (size_t)CO * CI * 9 * 4);
        hipMemcpy(dx_, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(ddy, dy.data(), dy.size() * 4, hipMemcpyHostToDevice);
        Args a{dx_, ddy, part, B, H, W};
        hipLaunchKernelGGL(wgrad_wino_kernel, dim3(grid), dim3(512), NBUF * STAGE_BYTES, 0, a);
        hipLaunchKernelGGL(finish_kernel, dim3((CO * CI + 255) / 256), dim3(256), 0, 0, part, grid, dw);
        std::vector<float> got((size_t)CO * CI * 9);
        hipMemcpy(got.data(), dw, got.size() * 4, hipMemcpyDeviceToHost);
        printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
        std::vector<double> want;
        reference(x, dy, want, B, H, W);
        double maxerr = 0, maxv = 0;
        for (size_t i = 0; i < got.size(); ++i) {
            maxerr = fmax(maxerr, fabs(got[i] - want[i]));
            maxv = fmax(maxv, fabs(want[i]));
        }
        printf("small case: max |err| %.3e of max |dW| %.3e (%.2e relative)\n", maxerr, maxv, maxerr / maxv);
        hipFree(dx_); hipFree(ddy); hipFree(part); hipFree(dw);
    }
    // ---- timing at the two C2 shapes that are 32 -> 32 (and the 256^2 size for scale)
    for (int sz : {128, 256}) {
        const int B = 32, H = sz, W = sz;
        const size_t n = (size_t)B * 32 * H * W;
        float *dx_, *ddy, *part, *dw;
        const int grid = 256;
        hipMalloc(&dx_, n * 4);
        hipMalloc(&ddy, n * 4);
        hipMalloc(&part, (size_t)grid * 16 * CO * CI * 4);
        hipMalloc(&dw, (size_t)CO * CI * 9 * 4);
        hipMemset(dx_, 0, n * 4);
        hipMemset(ddy, 0, n * 4);
        Args a{dx_, ddy, part, B, H, W};
        hipEvent_t e0, e1, e2;
        hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wgrad_wino_kernel, dim3(grid), dim3(512), NBUF * STAGE_BYTES, 0, a);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(wgrad_wino_kernel, dim3(grid), dim3(512), NBUF * STAGE_BYTES, 0, a);
        hipEventRecord(e1);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(finish_kernel, dim3((CO * CI + 255) / 256), dim3(256), 0, 0, part, grid, dw);
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        float ms1 = 0, ms2 = 0;
        hipEventElapsedTime(&ms1, e0, e1);
        hipEventElapsedTime(&ms2, e1, e2);
        const double direct = 2.0 * 9 * CI * CO * B * H * W;
        printf("B %d %dx%d 32 -> 32: %.1f us (+ finish %.1f us) = %.1f TFLOP/s direct-equivalent, %.1f executed; launch: %s\n", B, H, W, ms1 * 100, ms2 * 100,
               direct / (ms1 * 100) / 1e6, direct * 16 / 36 / (ms1 * 100) / 1e6, hipGetErrorString(hipGetLastError()));
        hipFree(dx_); hipFree(ddy); hipFree(part); hipFree(dw);
    }
    {
        const int B = 32, H = 256, W = 256;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, c16::NBUF2 * c16::STAGE2);
        float *dx_, *ddy, *part;
        const int grid = 256;
        hipMalloc(&dx_, (size_t)B * 16 * H * W * 4);
        hipMalloc(&ddy, (size_t)B * 32 * H * W * 4);
        hipMalloc(&part, (size_t)grid * 16 * CO * 16 * 4);
        hipMemset(dx_, 0, (size_t)B * 16 * H * W * 4);
        hipMemset(ddy, 0, (size_t)B * 32 * H * W * 4);
        Args a{dx_, ddy, part, B, H, W};
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wgrad_wino16_kernel, dim3(grid), dim3(512), c16::NBUF2 * c16::STAGE2, 0, a);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(wgrad_wino16_kernel, dim3(grid), dim3(512), c16::NBUF2 * c16::STAGE2, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms1 = 0;
        hipEventElapsedTime(&ms1, e0, e1);
        const double direct = 2.0 * 9 * 16 * CO * B * H * W;
        printf("B %d %dx%d 16 -> 32 (timing only): %.1f us = %.1f TFLOP/s direct-equivalent, %.1f executed, LDS %d; launch: %s\n", B, H, W, ms1 * 100, direct / (ms1 * 100) / 1e6,
               direct * 16 / 36 / (ms1 * 100) / 1e6, c16::NBUF2 * c16::STAGE2, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
