// HBM-bound kernels of the Y-Net path: pooling, bilinear x2, waypoint pyramid, BCE-with-logits,
// soft-argmax, sigmoid(x/T), heat-map patch gather.  One pass over the data each, coalesced
// (16-byte where alignment allows), wavefront (64-lane) shuffles for reductions.
// Reference call sites are cited per kernel; the public C ABI is include/ynet_hip.h.
#include "ynet_common.h"
#include "bce_element.h"
#include <stdlib.h>

// Plane-wise kernels: blockIdx.y walks the (b,c) planes, blockIdx.x the items of a plane -- the item index stays
// 32-bit and costs one division (by the row length) instead of the three 64-bit divisions of a flat index, which made
// these kernels VALU-bound at ~3.6 TB/s.
static inline dim3 plane_grid(long long planes, long long per_plane) {
    long long gx = (per_plane + 255) / 256;
    if (gx > 4096) gx = 4096;
    long long gy = planes < 65535 ? planes : 65535;
    while (gx * gy > (1ll << 22) && gy > 1) gy = (gy + 1) / 2;
    return dim3((unsigned)gx, (unsigned)gy);
}

static inline int grid_for(long long n, int block, int cap = 2048 * 4) {
    long long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(2,2) forward / backward   (models/ynet.py:202,215,326,340,354,367)
// backward recomputes the argmax from x with ATen's rule: first maximum in window scan order.
// ------------------------------------------------------------------------------------------------
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long N, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, per_plane = Ho * Wo;
    for (long long n = blockIdx.y; n < N; n += gridDim.y)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < per_plane; i += gridDim.x * 256) {
            const int yo = i / Wo, xo = i - yo * Wo;
            const float* p = x + (n * H + 2 * yo) * W + 2 * xo;
            float m = p[0];
            float v = p[1];
            m = (v > m || v != v) ? v : m;
            v = p[W];
            m = (v > m || v != v) ? v : m;
            v = p[W + 1];
            m = (v > m || v != v) ? v : m;
            y[n * per_plane + i] = m;
        }
}

// dx = route(dy) + add0 + add1 (either addend may be NULL): the max-pool backward also folds in the gradients the two
// decoders produced for the same feature map (skip connections), replacing two full-size elementwise adds of the
// autograd engine.  Even H and W, 8-byte aligned rows.
// RELU: x is the post-ReLU output of the conv that receives dx as ITS output gradient; that conv's backward would zero
// dx where x <= 0 (models/ynet.py: nn.ReLU after every encoder conv) by reading x once more next to dx -- x is in this
// kernel's registers already, so the mask is applied here and the conv runs its unmasked dgrad / wgrad kernels.
// CODE: x is not read -- the conv that produced it left one byte per 2 x 2 block (ynet_conv2d_winograd_cat_pool_code: bits 0..1 the arg-max by
// the rule above, bits 2..5 "element is positive" in window scan order), a quarter of a float per element instead of a float.
template <bool RELU, bool CODE = false>
__global__ void maxpool2_bwd_add_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                        const float* __restrict__ add0, const float* __restrict__ add1,
                                        float* __restrict__ dx, long long N, int H, int W) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int Ho = H >> 1, Wo = W >> 1, per_plane = Ho * Wo;
    for (long long n = blockIdx.y; n < N; n += gridDim.y)
    for (int ip = blockIdx.x * 256 + threadIdx.x; ip < per_plane; ip += gridDim.x * 256) {
        const int yo = ip / Wo, xo = ip - yo * Wo;
        const long long i = n * per_plane + ip;
        const long long base = (n * H + 2 * yo) * W + 2 * xo;
        f2 t, b;
        int arg = 0;
        if (CODE) {
            const unsigned code = reinterpret_cast<const unsigned char*>(x)[i];
            arg = (int)(code & 3u);
            t = f2{(code & 4u) ? 1.f : 0.f, (code & 8u) ? 1.f : 0.f};       // (only their sign is looked at below)
            b = f2{(code & 16u) ? 1.f : 0.f, (code & 32u) ? 1.f : 0.f};
        } else {
            t = *reinterpret_cast<const f2*>(x + base);
            b = *reinterpret_cast<const f2*>(x + base + W);
            float m = t[0];
            if (t[1] > m || t[1] != t[1]) { m = t[1]; arg = 1; }
            if (b[0] > m || b[0] != b[0]) { m = b[0]; arg = 2; }
            if (b[1] > m || b[1] != b[1]) { m = b[1]; arg = 3; }
        }
        const float g = dy[i];
        f2 o0 = {arg == 0 ? g : 0.f, arg == 1 ? g : 0.f}, o1 = {arg == 2 ? g : 0.f, arg == 3 ? g : 0.f};
        if (add0) {
            o0 += *reinterpret_cast<const f2*>(add0 + base);
            o1 += *reinterpret_cast<const f2*>(add0 + base + W);
        }
        if (add1) {
            o0 += *reinterpret_cast<const f2*>(add1 + base);
            o1 += *reinterpret_cast<const f2*>(add1 + base + W);
        }
        if (RELU) {
            o0[0] = t[0] > 0.f ? o0[0] : 0.f;
            o0[1] = t[1] > 0.f ? o0[1] : 0.f;
            o1[0] = b[0] > 0.f ? o1[0] : 0.f;
            o1[1] = b[1] > 0.f ? o1[1] : 0.f;
        }
        *reinterpret_cast<f2*>(dx + base) = o0;
        *reinterpret_cast<f2*>(dx + base + W) = o1;
    }
}

__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    float* __restrict__ dx, long long N, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;   // cover odd trailing row/col with zeros
    const long long total = N * Hc * Wc;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wc);
        const int yo = (int)((i / Wc) % Hc);
        const long long n = i / ((long long)Wc * Hc);
        const long long base = (n * H + 2 * yo) * W + 2 * xo;
        if (yo < Ho && xo < Wo) {
            const float* p = x + base;
            float m = p[0];
            int arg = 0;
            float v = p[1];
            if (v > m || v != v) { m = v; arg = 1; }
            v = p[W];
            if (v > m || v != v) { m = v; arg = 2; }
            v = p[W + 1];
            if (v > m || v != v) { m = v; arg = 3; }
            const float g = dy[(n * Ho + yo) * Wo + xo];
            dx[base] = arg == 0 ? g : 0.f;
            dx[base + 1] = arg == 1 ? g : 0.f;
            dx[base + W] = arg == 2 ? g : 0.f;
            dx[base + W + 1] = arg == 3 ? g : 0.f;
        } else {
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    if (2 * yo + a < H && 2 * xo + b < W) dx[base + a * W + b] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False)   (models/ynet.py:463)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void up2_src(int o, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = 0.5f * ((float)o + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long N, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = N * Ho * Wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        const int oy = (int)((i / Wo) % Ho);
        const long long n = i / ((long long)Wo * Ho);
        int h0, h1, w0, w1;
        float hl0, hl1, wl0, wl1;
        up2_src(oy, H, h0, h1, hl0, hl1);
        up2_src(ox, W, w0, w1, wl0, wl1);
        const float* p = x + n * H * W;
        y[i] = hl0 * (wl0 * p[h0 * W + w0] + wl1 * p[h0 * W + w1]) +
               hl1 * (wl0 * p[h1 * W + w0] + wl1 * p[h1 * W + w1]);
    }
}

// 4 consecutive outputs per thread (one 16-byte store); needs W even
__global__ __launch_bounds__(256) void upsample2x_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 long long N, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W, Wq = Wo >> 2, per_plane = Ho * Wq;
    for (long long n = blockIdx.y; n < N; n += gridDim.y)
    for (int ip = blockIdx.x * 256 + threadIdx.x; ip < per_plane; ip += gridDim.x * 256) {
        const int oy = ip / Wq, j = ip - oy * Wq;
        const long long i = n * per_plane + ip;
        int h0, h1;
        float hl0, hl1;
        up2_src(oy, H, h0, h1, hl0, hl1);
        const float* p0 = x + (n * H + h0) * W;
        const float* p1 = x + (n * H + h1) * W;
        // input columns 2j-1 .. 2j+2 (clamped) feed outputs 4j .. 4j+3
        const int c0 = max(2 * j - 1, 0), c1 = 2 * j, c2 = 2 * j + 1, c3 = min(2 * j + 2, W - 1);
        const float a0 = p0[c0], a1 = p0[c1], a2 = p0[c2], a3 = p0[c3];
        const float b0 = p1[c0], b1 = p1[c1], b2 = p1[c2], b3 = p1[c3];
        float4 o;
        // same expression tree as the scalar kernel: hl0*(wl0*v00 + wl1*v01) + hl1*(wl0*v10 + wl1*v11)
        const float w00 = j == 0 ? 1.f : 0.25f, w01 = j == 0 ? 0.f : 0.75f;     // output 4j: (c0|c1) -> at j==0 src clamps to 0
        o.x = j == 0 ? hl0 * (1.f * a1 + 0.f * a2) + hl1 * (1.f * b1 + 0.f * b2)
                     : hl0 * (w00 * a0 + w01 * a1) + hl1 * (w00 * b0 + w01 * b1);
        o.y = hl0 * (0.75f * a1 + 0.25f * a2) + hl1 * (0.75f * b1 + 0.25f * b2);
        o.z = hl0 * (0.25f * a1 + 0.75f * a2) + hl1 * (0.25f * b1 + 0.75f * b2);
        o.w = hl0 * (0.75f * a2 + 0.25f * a3) + hl1 * (0.75f * b2 + 0.25f * b3);
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// 2*RI output rows x 4 output columns per thread from RI + 2 input rows x 4 input columns (one 8-byte load + 2 halo
// loads per row): every store instruction of a wavefront writes 1 KB of one output row, and the load instruction count
// per store falls from 8 (one output row per thread: bound by its load stream at 3.3 TB/s) to 3 (RI = 2) / 2.25 (RI = 4).
// Needs W even, H % RI == 0, 16-byte aligned y.  Every output keeps the scalar kernel's expression tree
// hl0*(wl*A[i0] + wr*A[i1]) + hl1*(wl*B[i0] + wr*B[i1]).
template <int RI>
__global__ __launch_bounds__(256) void upsample2x_fwd_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  long long N, int H, int W) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int Wo = 2 * W, Wq = W >> 1, per_plane = (H / RI) * Wq;
    for (long long n = blockIdx.y; n < N; n += gridDim.y)
    for (int ip = blockIdx.x * 256 + threadIdx.x; ip < per_plane; ip += gridDim.x * 256) {
        const int kk = ip / Wq, j = ip - kk * Wq, k = RI * kk;          // input rows k .. k+RI-1 -> output rows 2k .. 2k+2RI-1
        const float* px = x + n * H * W;
        const int cl = max(2 * j - 1, 0), cr = min(2 * j + 2, W - 1);
        float hx[RI + 2][4];
#pragma unroll
        for (int r = 0; r < RI + 2; ++r) {
            const int ir = r == 0 ? max(k - 1, 0) : (r == RI + 1 ? min(k + RI, H - 1) : k + r - 1);
            const float* row = px + (long long)ir * W;
            const f2 m = *reinterpret_cast<const f2*>(row + 2 * j);
            const float c[4] = {row[cl], m[0], m[1], row[cr]};
            hx[r][0] = j == 0 ? 1.f * c[1] + 0.f * c[2] : 0.25f * c[0] + 0.75f * c[1];      // column 0: source clamps to 0
            hx[r][1] = 0.75f * c[1] + 0.25f * c[2];
            hx[r][2] = 0.25f * c[1] + 0.75f * c[2];
            hx[r][3] = 0.75f * c[2] + 0.25f * c[3];
        }
        float* py = y + (n * 2 * H + 2 * k) * Wo + 4 * j;
#pragma unroll
        for (int r = 0; r < RI; ++r) {          // input row k + r = hx[r + 1]
            float4 o;
            float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int e = 0; e < 4; ++e)         // output row 2(k+r): rows (k+r-1, k+r) x (.25, .75); row 0: (0, 1) x (1, 0)
                ov[e] = (r == 0 && k == 0) ? 1.f * hx[1][e] + 0.f * hx[2][e] : 0.25f * hx[r][e] + 0.75f * hx[r + 1][e];
            *reinterpret_cast<float4*>(py + (2 * r) * Wo) = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = 0.75f * hx[r + 1][e] + 0.25f * hx[r + 2][e];
            *reinterpret_cast<float4*>(py + (2 * r + 1) * Wo) = o;
        }
    }
}

// 4 consecutive input-gradient pixels per thread; needs W % 4 == 0.  1-D weights of input i over outputs
// 2i-1 .. 2i+2 are {.25,.75,.75,.25}, except that output 0 / 2W-1 put their whole weight on input 0 / W-1.
// relu_of (all three kernels; may be NULL): the post-ReLU activation [N][H][W] that was up-sampled -- its ReLU backward is applied
// to dx here, where the value is one extra coalesced read, instead of by the masked dgrad / wgrad kernels of its producer.
__global__ __launch_bounds__(256) void upsample2x_bwd_vec_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                 const float* __restrict__ relu_of, long long N, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W, Wq = W >> 2, per_plane = H * Wq;
    for (long long n = blockIdx.y; n < N; n += gridDim.y)
    for (int ip = blockIdx.x * 256 + threadIdx.x; ip < per_plane; ip += gridDim.x * 256) {
        const int iy = ip / Wq, j = ip - iy * Wq;
        const long long i = n * per_plane + ip;
        const float* g = dy + n * Ho * Wo;
        float wy[4];
        wy[0] = iy > 0 ? 0.25f : 0.f;
        wy[1] = iy > 0 ? 0.75f : 1.f;
        wy[2] = iy < H - 1 ? 0.75f : 1.f;
        wy[3] = iy < H - 1 ? 0.25f : 0.f;
        // column sums: output columns 8j-1 .. 8j+8 are needed for inputs 4j .. 4j+3
        float col[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) col[c] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 2 * iy - 1 + r;
            if (wy[r] == 0.f) continue;
            const float* row = g + (long long)oy * Wo + 8 * j;
            const float4 m0 = *reinterpret_cast<const float4*>(row);
            const float4 m1 = *reinterpret_cast<const float4*>(row + 4);
            const float left = j > 0 ? row[-1] : 0.f;
            const float right = j < Wq - 1 ? row[8] : 0.f;
            col[0] += wy[r] * left;
            col[1] += wy[r] * m0.x;
            col[2] += wy[r] * m0.y;
            col[3] += wy[r] * m0.z;
            col[4] += wy[r] * m0.w;
            col[5] += wy[r] * m1.x;
            col[6] += wy[r] * m1.y;
            col[7] += wy[r] * m1.z;
            col[8] += wy[r] * m1.w;
            col[9] += wy[r] * right;
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ix = 4 * j + e;
            const float w0 = ix > 0 ? 0.25f : 0.f, w1 = ix > 0 ? 0.75f : 1.f;
            const float w2 = ix < W - 1 ? 0.75f : 1.f, w3 = ix < W - 1 ? 0.25f : 0.f;
            o[e] = (w0 * col[2 * e] + w1 * col[2 * e + 1]) + (w2 * col[2 * e + 2] + w3 * col[2 * e + 3]);
        }
        if (relu_of) {
            const float4 a = reinterpret_cast<const float4*>(relu_of)[i];
            o[0] = a.x > 0.f ? o[0] : 0.f;
            o[1] = a.y > 0.f ? o[1] : 0.f;
            o[2] = a.z > 0.f ? o[2] : 0.f;
            o[3] = a.w > 0.f ? o[3] : 0.f;
        }
        reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// RI input-gradient rows x 2 columns per thread from 2*RI + 2 rows x 6 columns of dy (one 16-byte load + 2 halo loads
// per row): every 16-byte load instruction of a wavefront reads 1 KB of one dy row, and a dy row is fetched by
// (2*RI + 2) / (2*RI) threads instead of 2.  Needs W even, H % RI == 0, 16-byte aligned dy planes, 8-byte aligned dx.
template <int RI>
__global__ __launch_bounds__(256) void upsample2x_bwd_rows_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                  const float* __restrict__ relu_of, long long N, int H, int W) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int Ho = 2 * H, Wo = 2 * W, Wq = W >> 1, per_plane = (H / RI) * Wq;
    for (long long n = blockIdx.y; n < N; n += gridDim.y)
    for (int ip = blockIdx.x * 256 + threadIdx.x; ip < per_plane; ip += gridDim.x * 256) {
        const int kk = ip / Wq, l = ip - kk * Wq, iy0 = RI * kk;
        const float* g = dy + n * Ho * Wo;
        // 1-D weights of input column ix over output columns 2ix-1 .. 2ix+2 ({.25,.75,.75,.25}; the border outputs put
        // their whole weight on the border input)
        float wx[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ix = 2 * l + e;
            wx[e][0] = ix > 0 ? 0.25f : 0.f;
            wx[e][1] = ix > 0 ? 0.75f : 1.f;
            wx[e][2] = ix < W - 1 ? 0.75f : 1.f;
            wx[e][3] = ix < W - 1 ? 0.25f : 0.f;
        }
        f2 act[RI];                     // (queued ahead of the dy rows)
        if (relu_of) {
#pragma unroll
            for (int q = 0; q < RI; ++q) act[q] = *reinterpret_cast<const f2*>(relu_of + (n * H + iy0 + q) * W + 2 * l);
        }
        float h[2 * RI + 2][2];        // horizontally combined dy rows 2*iy0-1 .. 2*iy0+2*RI
#pragma unroll
        for (int r = 0; r < 2 * RI + 2; ++r) {
            const int oy = 2 * iy0 - 1 + r;
            if (oy < 0 || oy >= Ho) {
                h[r][0] = h[r][1] = 0.f;
                continue;
            }
            const float* row = g + (long long)oy * Wo + 4 * l;
            const float4 m = *reinterpret_cast<const float4*>(row);
            const float c[6] = {l > 0 ? row[-1] : 0.f, m.x, m.y, m.z, m.w, l < Wq - 1 ? row[4] : 0.f};
#pragma unroll
            for (int e = 0; e < 2; ++e)
                h[r][e] = (wx[e][0] * c[2 * e] + wx[e][1] * c[2 * e + 1]) + (wx[e][2] * c[2 * e + 2] + wx[e][3] * c[2 * e + 3]);
        }
#pragma unroll
        for (int q = 0; q < RI; ++q) {
            const int iy = iy0 + q;
            const float w0 = iy > 0 ? 0.25f : 0.f, w1 = iy > 0 ? 0.75f : 1.f;
            const float w2 = iy < H - 1 ? 0.75f : 1.f, w3 = iy < H - 1 ? 0.25f : 0.f;
            f2 o;
#pragma unroll
            for (int e = 0; e < 2; ++e)
                o[e] = (w0 * h[2 * q][e] + w1 * h[2 * q + 1][e]) + (w2 * h[2 * q + 2][e] + w3 * h[2 * q + 3][e]);
            if (relu_of) {
                o[0] = act[q][0] > 0.f ? o[0] : 0.f;
                o[1] = act[q][1] > 0.f ? o[1] : 0.f;
            }
            *reinterpret_cast<f2*>(dx + (n * H + iy) * W + 2 * l) = o;
        }
    }
}

// dx[i][j] = sum over the (at most 4x4) output pixels whose stencil touches (i,j)
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, const float* __restrict__ relu_of,
                                      long long N, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = N * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W);
        const int iy = (int)((i / W) % H);
        const long long n = i / ((long long)W * H);
        const float* g = dy + n * Ho * Wo;
        float wy[4], wx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int o = 2 * iy - 1 + k, a0, a1;
            float l0, l1;
            wy[k] = 0.f;
            if (o >= 0 && o < Ho) {
                up2_src(o, H, a0, a1, l0, l1);
                wy[k] = (a0 == iy ? l0 : 0.f) + (a1 == iy ? l1 : 0.f);
            }
            o = 2 * ix - 1 + k;
            wx[k] = 0.f;
            if (o >= 0 && o < Wo) {
                up2_src(o, W, a0, a1, l0, l1);
                wx[k] = (a0 == ix ? l0 : 0.f) + (a1 == ix ? l1 : 0.f);
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = 2 * iy - 1 + a;
            if (wy[a] == 0.f) continue;
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ox = 2 * ix - 1 + b;
                if (wx[b] != 0.f) row += wx[b] * g[oy * Wo + ox];
            }
            acc += wy[a] * row;
        }
        dx[i] = (relu_of == nullptr || relu_of[i] > 0.f) ? acc : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// AvgPool2d(2^i) pyramid of the waypoint map, i = 1..nlev  (utils/train_epoch.py:97-100,
// utils/evaluate.py:255-257): one pass, one 32x32 tile per workgroup, levels chained through LDS.
// ------------------------------------------------------------------------------------------------
struct PyrArgs {
    const float* x;
    float* out[5];
    int nlev, H, W;
    long long N;
};

__global__ __launch_bounds__(256) void avgpool_pyramid_kernel(const PyrArgs a) {
    __shared__ float t[32 * 33];
    __shared__ float l1[16 * 16], l2[8 * 8], l3[4 * 4], l4[2 * 2];
    const int tid = threadIdx.x;
    const int tiles_x = a.W / 32, tiles_y = a.H / 32;
    long long bid = blockIdx.x;
    const int txi = (int)(bid % tiles_x);
    const int tyi = (int)((bid / tiles_x) % tiles_y);
    const long long n = bid / ((long long)tiles_x * tiles_y);
    const float* p = a.x + (n * a.H + tyi * 32) * a.W + txi * 32;
    for (int i = tid; i < 1024; i += 256) t[(i >> 5) * 33 + (i & 31)] = p[(i >> 5) * (long long)a.W + (i & 31)];
    __syncthreads();
    {
        const int oy = tid >> 4, ox = tid & 15;
        const float s = (t[(2 * oy) * 33 + 2 * ox] + t[(2 * oy) * 33 + 2 * ox + 1]) +
                        (t[(2 * oy + 1) * 33 + 2 * ox] + t[(2 * oy + 1) * 33 + 2 * ox + 1]);
        l1[tid] = s;
        a.out[0][(n * (a.H / 2) + tyi * 16 + oy) * (a.W / 2) + txi * 16 + ox] = s * 0.25f;
    }
    if (a.nlev < 2) return;
    __syncthreads();
    if (tid < 64) {
        const int oy = tid >> 3, ox = tid & 7;
        const float s = (l1[(2 * oy) * 16 + 2 * ox] + l1[(2 * oy) * 16 + 2 * ox + 1]) +
                        (l1[(2 * oy + 1) * 16 + 2 * ox] + l1[(2 * oy + 1) * 16 + 2 * ox + 1]);
        l2[tid] = s;
        a.out[1][(n * (a.H / 4) + tyi * 8 + oy) * (a.W / 4) + txi * 8 + ox] = s * (1.f / 16.f);
    }
    if (a.nlev < 3) return;
    __syncthreads();
    if (tid < 16) {
        const int oy = tid >> 2, ox = tid & 3;
        const float s = (l2[(2 * oy) * 8 + 2 * ox] + l2[(2 * oy) * 8 + 2 * ox + 1]) +
                        (l2[(2 * oy + 1) * 8 + 2 * ox] + l2[(2 * oy + 1) * 8 + 2 * ox + 1]);
        l3[tid] = s;
        a.out[2][(n * (a.H / 8) + tyi * 4 + oy) * (a.W / 8) + txi * 4 + ox] = s * (1.f / 64.f);
    }
    if (a.nlev < 4) return;
    __syncthreads();
    if (tid < 4) {
        const int oy = tid >> 1, ox = tid & 1;
        const float s = (l3[(2 * oy) * 4 + 2 * ox] + l3[(2 * oy) * 4 + 2 * ox + 1]) +
                        (l3[(2 * oy + 1) * 4 + 2 * ox] + l3[(2 * oy + 1) * 4 + 2 * ox + 1]);
        l4[tid] = s;
        a.out[3][(n * (a.H / 16) + tyi * 2 + oy) * (a.W / 16) + txi * 2 + ox] = s * (1.f / 256.f);
    }
    if (a.nlev < 5) return;
    __syncthreads();
    if (tid == 0) {
        const float s = (l4[0] + l4[1]) + (l4[2] + l4[3]);
        a.out[4][(n * (a.H / 32) + tyi) * (a.W / 32) + txi] = s * (1.f / 1024.f);
    }
}

// ------------------------------------------------------------------------------------------------
// wavefront / block reductions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------------
// BCEWithLogitsLoss(reduction='mean')  (models/trainer.py:206, utils/train_epoch.py:94,106)
//   l = (1 - t) * x - log_sigmoid(x),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
//   dl/dx = (sigmoid(x) - t) * g / n
// ------------------------------------------------------------------------------------------------
// (one element of the loss and of its gradient: bce_element.h)
// GRAD: also writes dx = (sigmoid(x) - t) * gs in the same pass (gs = the expected upstream gradient / n).
template <bool GRAD>
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                      long long n, double* __restrict__ partial,
                                                      float* __restrict__ dx, float gs) {
    __shared__ double ws[4];
    double acc = 0.0;
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    float4* d4 = reinterpret_cast<float4*>(dx);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 xv = x4[i], tv = t4[i];
        float4 o;
        float s = bce_element<GRAD>(xv.x, tv.x, gs, o.x);
        s += bce_element<GRAD>(xv.y, tv.y, gs, o.y);
        s += bce_element<GRAD>(xv.z, tv.z, gs, o.z);
        s += bce_element<GRAD>(xv.w, tv.w, gs, o.w);
        if (GRAD) d4[i] = o;
        acc += (double)s;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float d;
        acc += (double)bce_element<GRAD>(x[i], t[i], gs, d);
        if (GRAD) dx[i] = d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// ------------------------------------------------------------------------------------------------
// Predictor + criterion in one pass (models/ynet.py:450-451,469 `self.predictor(x)`; utils/train_epoch.py:93-94,
// 105-106 `criterion(pred_map, gt_map) * loss_scale`; and, for the backward pass, the predictor's dgrad):
//   y[co]      = bias[co] + sum_ci w[co][ci] * x[ci]                 (the 1x1 predictor, fp32 FMA chain over ci in order)
//   loss      += BCE-with-logits(y[co], t[co])                       (fp64 block partials, summed by the last block)
//   dy[co]     = (sigmoid(y[co]) - t[co]) * gs                       (gs = expected upstream gradient / n)
//   dx[ci]     = sum_co w[co][ci] * dy[co]                           (written when the decoder needs a gradient)
// The activation x (32 channels at full resolution, the largest tensor of the step) is read ONCE; unfused it is read by
// the predictor, the logits are written, re-read by the loss, dy is written, re-read by the predictor's dgrad: 88
// channel planes of HBM traffic instead of 124, and three launches less per decoder.  One thread owns PX consecutive
// pixels and every output channel; filter rows come through scalar loads.
// ------------------------------------------------------------------------------------------------
struct PredBceArgs {
    const float* x;
    long long x_bs;
    const float* wp;        // packed [cin_pad][cout_pad] (forward layout of a 1x1 filter)
    const float* bias;
    const float* t;         // [B][cout][HW]
    const float* t_xy;      // BLOB form of the target: plane (b, co) is the window of the Gaussian template around (x, y) = t_xy[2 * (b * cout + co) ..]
    const float* t_blob;    //   (heatmap_analytic_kernel kind 1: the m x m blob placed at the rounded position, zero elsewhere, all zero when the
    int t_m, t_S, t_W, t_H; //    H x W window would leave the S x S template) -- t is not read
    float* y;               // [B][cout][HW] logits
    float* dx;              // [B][cin][HW] or NULL
    int relu_mask;          // dx is zeroed where x <= 0 (x = post-ReLU output of the conv that receives dx: see maxpool2_bwd_add_kernel)
    float* dy;              // [B][cout][HW] or NULL (wanted when the predictor itself trains)
    double* partial;        // [gridDim.x]
    unsigned* ticket;
    float* loss;
    int cin, cout, cout_pad, B;
    long long hw4, n;       // H*W/4, number of loss elements
    float gs;
};

typedef const __attribute__((address_space(4))) float* glue_const_f32_ptr;      // uniform reads -> scalar loads

// (4 workgroups per CU = 4 waves per SIMD: <= 128 VGPRs.  Left alone hipcc hoists the loads of both loops and takes 175
// registers -- 2 waves per SIMD, too few to hide the HBM latency of a streaming kernel: 213 -> see DESIGN.md)
template <int CT, int PX, bool BLOB = false>
__global__ __launch_bounds__(256, (CT * PX <= 48 ? 4 : (CT * PX <= 60 ? 3 : 2))) void pred_bce_kernel(const PredBceArgs a) {
    typedef float vec_t __attribute__((ext_vector_type(PX)));
    __shared__ double ws[4];
    __shared__ unsigned last;
    const long long hwv = a.hw4 * (4 / PX);
    const long long total = (long long)a.B * hwv;
    const glue_const_f32_ptr w = (glue_const_f32_ptr)a.wp, bias = (glue_const_f32_ptr)a.bias;
    double acc_loss = 0.0;
    for (long long q = blockIdx.x * 256ll + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int b = (int)(q / hwv);
        const long long p = q - (long long)b * hwv;
        const vec_t* xp = reinterpret_cast<const vec_t*>(a.x + (long long)b * a.x_bs) + p;
        vec_t acc[CT];
#pragma unroll
        for (int co = 0; co < CT; ++co) {
            const float bv = (a.bias != nullptr && co < a.cout) ? bias[co] : 0.f;
#pragma unroll
            for (int e = 0; e < PX; ++e) acc[co][e] = bv;
        }
        // NB input planes are fetched before their FMAs start: a streaming kernel lives on bytes in flight (with two
        // loads per thread outstanding it reached 3.3 TB/s -- Little's law at ~2.5 us of loaded HBM latency)
#ifdef YNET_PRED_BCE_NB
        constexpr int NB = YNET_PRED_BCE_NB;      // (development builds: tools/ab_pred_bce.sh)
#else
        constexpr int NB = 8;
#endif
        unsigned xpos[PX];          // bit ci: x[ci] > 0 at this thread's pixel e (cin <= 32): the ReLU mask for dx, kept while x streams by
#pragma unroll
        for (int e = 0; e < PX; ++e) xpos[e] = 0u;
#pragma unroll 1
        for (int c0 = 0; c0 < a.cin; c0 += NB) {
            vec_t v[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int ci = c0 + k < a.cin ? c0 + k : a.cin - 1;      // (clamped: the surplus products are skipped below)
                v[k] = xp[(long long)ci * hwv];
            }
            if (a.relu_mask) {
#pragma unroll
                for (int k = 0; k < NB; ++k)
#pragma unroll
                    for (int e = 0; e < PX; ++e) xpos[e] |= (v[k][e] > 0.f ? 1u : 0u) << ((c0 + k) & 31);
            }
#if defined(YNET_PRED_BCE_DIAG) && (YNET_PRED_BCE_DIAG & 1)
            // (development build, WRONG results: the forward products replaced by one add per plane -- what does the launch cost without its FMAs?)
#pragma unroll
            for (int k = 0; k < NB; ++k)
#pragma unroll
                for (int e = 0; e < PX; ++e) acc[k % CT][e] += v[k][e];
#else
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (c0 + k < a.cin) {
#pragma unroll
                    for (int co = 0; co < CT; ++co) {
                        const float wv = w[(c0 + k) * a.cout_pad + co];
#pragma unroll
                        for (int e = 0; e < PX; ++e) acc[co][e] = __builtin_fmaf(v[k][e], wv, acc[co][e]);
                    }
                }
            }
#endif
        }
        const long long obase = (long long)b * a.cout * hwv + p;
        float s = 0.f;
        int py = 0, px0 = 0;
        if (BLOB) {         // this thread's PX pixels: row py, columns px0 .. px0 + PX - 1 (W % PX == 0)
            const int pix = (int)p * PX;
            py = pix / a.t_W;
            px0 = pix - py * a.t_W;
        }
#pragma unroll
        for (int co = 0; co < CT; ++co) {
            if (co < a.cout) {
                vec_t tv;
                if (BLOB) {
                    const float* pos = a.t_xy + 2ll * ((long long)b * a.cout + co);
                    const int rx = (int)rintf(pos[0]), ry = (int)rintf(pos[1]);
                    const int ox = a.t_S / 2 - rx, oy = a.t_S / 2 - ry;
                    const bool inside = !(ox < 0 || oy < 0 || ox + a.t_W > a.t_S || oy + a.t_H > a.t_S);
                    const int by = py - ry + a.t_m / 2, bx0 = px0 - rx + a.t_m / 2;
                    const bool row_in = inside && by >= 0 && by < a.t_m;
#pragma unroll
                    for (int e = 0; e < PX; ++e) {
                        const int bx = bx0 + e;
                        tv[e] = (row_in && bx >= 0 && bx < a.t_m) ? a.t_blob[by * a.t_m + bx] : 0.f;
                    }
                } else {
                    tv = reinterpret_cast<const vec_t*>(a.t)[obase + (long long)co * hwv];
                }
                reinterpret_cast<vec_t*>(a.y)[obase + (long long)co * hwv] = acc[co];
                vec_t d;
#pragma unroll
                for (int e = 0; e < PX; ++e) {
                    float de;
#if defined(YNET_PRED_BCE_DIAG) && (YNET_PRED_BCE_DIAG & 4)
                    de = acc[co][e] - tv[e];      // (development build, WRONG results: no exp / log / rcp)
                    s += de;
#else
                    s += bce_element<true>(acc[co][e], tv[e], a.gs, de);
#endif
                    d[e] = de;
                }
                acc[co] = d;        // the accumulator now holds dy
                if (a.dy != nullptr) reinterpret_cast<vec_t*>(a.dy)[obase + (long long)co * hwv] = d;
            } else {
#pragma unroll
                for (int e = 0; e < PX; ++e) acc[co][e] = 0.f;
            }
        }
        acc_loss += (double)s;
        if (a.dx != nullptr) {
            vec_t* dp = reinterpret_cast<vec_t*>(a.dx + (long long)b * a.cin * (hwv * PX)) + p;
#pragma unroll 2
            for (int ci = 0; ci < a.cin; ++ci) {
                vec_t o;
#pragma unroll
                for (int e = 0; e < PX; ++e) o[e] = 0.f;
#if defined(YNET_PRED_BCE_DIAG) && (YNET_PRED_BCE_DIAG & 2)
#pragma unroll
                for (int e = 0; e < PX; ++e) o[e] = acc[ci % CT][e];      // (development build, WRONG results: no dgrad products)
#else
#pragma unroll
                for (int co = 0; co < CT; ++co) {
                    const float wv = w[ci * a.cout_pad + co];      // (zero in the padded columns)
#pragma unroll
                    for (int e = 0; e < PX; ++e) o[e] = __builtin_fmaf(acc[co][e], wv, o[e]);
                }
#endif
                if (a.relu_mask) {
#pragma unroll
                    for (int e = 0; e < PX; ++e) o[e] = ((xpos[e] >> (ci & 31)) & 1u) ? o[e] : 0.f;
                }
                dp[(long long)ci * hwv] = o;
            }
        }
    }
    acc_loss = wave_sum(acc_loss);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc_loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        a.partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
        __threadfence();
        last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last) {         // the last block to finish sums the partials in a fixed order: bitwise reproducible
        __threadfence();
        double t = 0.0;
        for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) t += ((volatile double*)a.partial)[i];
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            a.loss[0] = (float)(((ws[0] + ws[1]) + (ws[2] + ws[3])) / (double)a.n);
            *a.ticket = 0u;     // ready for the next launch on this workspace
        }
    }
}

// dx *= g[0] / expected, and nothing at all when the upstream gradient is the expected one.
__global__ __launch_bounds__(256) void bce_rescale_kernel(float* __restrict__ dx, const float* __restrict__ g,
                                                          float expected, long long n) {
    const float gv = g[0];
    if (gv == expected) return;
    const float f = gv / expected;
    const long long n4 = n >> 2;
    float4* d4 = reinterpret_cast<float4*>(dx);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 v = d4[i];
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        d4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dx[(n4 << 2) + threadIdx.x] *= f;
}

__global__ __launch_bounds__(256) void bce_finish_kernel(const double* __restrict__ partial, int nparts, long long n,
                                                         float* __restrict__ loss) {
    __shared__ double ws[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((ws[0] + ws[1]) + (ws[2] + ws[3])) / (double)n);
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                      const float* __restrict__ gout, float* __restrict__ dx,
                                                      long long n) {
    const float g = gout[0] / (float)n;
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    float4* d4 = reinterpret_cast<float4*>(dx);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 xv = x4[i], tv = t4[i];
        float4 o;
        o.x = (1.f / (1.f + expf(-xv.x)) - tv.x) * g;
        o.y = (1.f / (1.f + expf(-xv.y)) - tv.y) * g;
        o.z = (1.f / (1.f + expf(-xv.z)) - tv.z) * g;
        o.w = (1.f / (1.f + expf(-xv.w)) - tv.w) * g;
        d4[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        dx[i] = (1.f / (1.f + expf(-x[i])) - t[i]) * g;
    }
}

// ------------------------------------------------------------------------------------------------
// SoftArgmax2D  (utils/softargmax.py:55-81): one workgroup per (b,c) plane, ONE pass over HBM with
// an online (running-max) softmax; per-thread fp32 partials over <= a few hundred pixels, combined
// across the wavefront and the workgroup in fp64.  out[plane] = (E[x], E[y]).
// ------------------------------------------------------------------------------------------------
struct SoftAcc {
    float m;        // running max
    float s, sx, sy;
};

// exp(t) for t <= 0 in 6 instructions: v_exp_f32 on the rounded product t * log2(e), corrected to first order by the
// product's exact residual (fma) -- about 1 ulp, against ~12 instructions of libm's expf (range and denormal handling
// this kernel does not need: terms below 2^-126 of the plane's maximum vanish in the fp32 sums anyway).  t = -inf
// (running maximum not set yet, or a -inf logit) is clamped and comes out as 0.
__device__ __forceinline__ float exp_le0(float t) {
    t = fmaxf(t, -88.f);
    const float L = 1.44269502162933349609375f, Ll = 1.925963033500011e-8f;      // log2(e) = L + Ll
    const float p = t * L;
    const float r = __builtin_fmaf(t, L, -p) + t * Ll;
    const float e = __builtin_amdgcn_exp2f(p);
    return __builtin_fmaf(e, r * 0.693147182464599609375f, e);
}

__device__ __forceinline__ void soft_add(SoftAcc& a, float v, float px, float py) {
    if (v > a.m) {
        const float r = exp_le0(a.m - v);   // a.m = -inf on the first element -> 0
        a.s *= r;
        a.sx *= r;
        a.sy *= r;
        a.m = v;
    }
    const float e = exp_le0(v - a.m);
    a.s += e;
    a.sx += e * px;
    a.sy += e * py;
}

// one workgroup, one plane `p` -> out2[0..1] = (E[x], E[y])
__device__ __forceinline__ void softargmax_plane(const float* __restrict__ p, float* __restrict__ out2, int H, int W, float eps) {
    __shared__ float wm[4];
    __shared__ double wsum[4][3];
    const int tid = threadIdx.x;
    SoftAcc a{-INFINITY, 0.f, 0.f, 0.f};
    bool poison = false;        // a NaN logit: the reference's softmax makes the whole plane NaN
    const int n = H * W;
    if ((W & 3) == 0) {
        const int w4 = W >> 2, n4 = n >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        // (row, first column) of this thread's current vector, advanced by 256 vectors per step without a division
        const int step_r = 256 / w4, step_c = (256 - step_r * w4) << 2;
        int row = tid / w4, col = (tid - row * w4) << 2;
        auto advance = [&]() {
            row += step_r;
            col += step_c;
            if (col >= W) {
                col -= W;
                ++row;
            }
        };
        auto fold = [&](const float4 v) {
            const float chk = (v.x + v.y) + (v.z + v.w);      // NaN for a NaN logit (exp_le0's clamp would hide it)
            poison = poison || chk != chk;
            // rescale at most once per 16-byte vector
            const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            if (mx > a.m) {
                const float r = exp_le0(a.m - mx);
                a.s *= r;
                a.sx *= r;
                a.sy *= r;
                a.m = mx;
            }
            const float e0 = exp_le0(v.x - a.m), e1 = exp_le0(v.y - a.m), e2 = exp_le0(v.z - a.m), e3 = exp_le0(v.w - a.m);
            const float es = (e0 + e1) + (e2 + e3);
            a.s += es;
            a.sx += __builtin_fmaf((float)col, es, __builtin_fmaf(3.f, e3, __builtin_fmaf(2.f, e2, e1)));
            a.sy += es * (float)row;
            advance();
        };
        int i = tid;
        for (; i + 7 * 256 < n4; i += 8 * 256) {     // eight independent 16-byte loads in flight per thread
            const float4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
            const float4 v4 = p4[i + 1024], v5 = p4[i + 1280], v6 = p4[i + 1536], v7 = p4[i + 1792];
            fold(v0);
            fold(v1);
            fold(v2);
            fold(v3);
            fold(v4);
            fold(v5);
            fold(v6);
            fold(v7);
        }
        for (; i < n4; i += 256) fold(p4[i]);
    } else {
        for (int i = tid; i < n; i += 256) {
            const int row = i / W, col = i - row * W;
            poison = poison || p[i] != p[i];
            soft_add(a, p[i], (float)col, (float)row);
        }
    }
    if (poison) a.s = __builtin_nanf("");
    // combine: block max, then rescaled sums in fp64
    float m = wave_max(a.m);
    if ((tid & 63) == 0) wm[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    const double r = (a.m == -INFINITY) ? 0.0 : (double)exp_le0(a.m - m);
    double s = wave_sum((double)a.s * r), sx = wave_sum((double)a.sx * r), sy = wave_sum((double)a.sy * r);
    if ((tid & 63) == 0) {
        wsum[tid >> 6][0] = s;
        wsum[tid >> 6][1] = sx;
        wsum[tid >> 6][2] = sy;
    }
    __syncthreads();
    if (tid == 0) {
        s = (wsum[0][0] + wsum[1][0]) + (wsum[2][0] + wsum[3][0]);
        if (m == INFINITY) s = (double)__builtin_nanf("");      // a +inf logit: exp(inf - inf) = NaN in the reference
        sx = (wsum[0][1] + wsum[1][1]) + (wsum[2][1] + wsum[3][1]);
        sy = (wsum[0][2] + wsum[1][2]) + (wsum[2][2] + wsum[3][2]);
        const double inv = 1.0 / (s + (double)eps);
        out2[0] = (float)(sx * inv);
        out2[1] = (float)(sy * inv);
    }
}

__global__ __launch_bounds__(256) void softargmax_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         int C, long long bs, int H, int W, float eps) {
    const long long plane = blockIdx.x;
    softargmax_plane(x + (plane / C) * bs + (plane % C) * (long long)H * W, out + plane * 2, H, W, eps);
}

// The read-out of a training step (utils/train_epoch.py:118-126) in two launches instead of ~14:
//   pred_traj = softargmax(pred_traj_map) [B, P, 2];  pred_goal = softargmax(pred_goal_map[:, -1:]) [B, 1, 2]     (this kernel:
//   the B*P planes of the first map and the last plane of every image of the second, one workgroup each)
//   ADE[b] = mean_p |gt[b, p] - pred_traj[b, p]| / resize_factor;  FDE[b] = |gt[b, -1] - pred_goal[b, 0]| / resize_factor
//   (train_readout_finish_kernel; the reference's elementwise chain ((d / rf) ** 2).sum(2) ** 0.5 in the same fp32 order)
__global__ __launch_bounds__(256) void softargmax_two_kernel(const float* __restrict__ x0, long long bs0, int C0,
                                                             const float* __restrict__ x1, long long bs1, long long n0,
                                                             float* __restrict__ out0, float* __restrict__ out1, int H, int W, float eps) {
    const long long plane = blockIdx.x;
    if (plane < n0) softargmax_plane(x0 + (plane / C0) * bs0 + (plane % C0) * (long long)H * W, out0 + plane * 2, H, W, eps);
    else softargmax_plane(x1 + (plane - n0) * bs1, out1 + (plane - n0) * 2, H, W, eps);
}

__global__ __launch_bounds__(256) void train_readout_finish_kernel(const float* __restrict__ traj, const float* __restrict__ goal,
                                                                   const float* __restrict__ gt, int B, int P, float rf,
                                                                   float* __restrict__ ade, float* __restrict__ fde) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float acc = 0.f;
    for (int p = 0; p < P; ++p) {
        const float dx = (gt[(b * P + p) * 2] - traj[(b * P + p) * 2]) / rf, dy = (gt[(b * P + p) * 2 + 1] - traj[(b * P + p) * 2 + 1]) / rf;
        acc += __fsqrt_rn(dx * dx + dy * dy);
    }
    ade[b] = acc / (float)P;
    const float gx = (gt[(b * P + P - 1) * 2] - goal[b * 2]) / rf, gy = (gt[(b * P + P - 1) * 2 + 1] - goal[b * 2 + 1]) / rf;
    fde[b] = __fsqrt_rn(gx * gx + gy * gy);
}

// ------------------------------------------------------------------------------------------------
// 1x1 predictor + SoftArgmax2D in one pass (evaluate()'s K trajectory passes: models/ynet.py:469 followed by
// utils/softargmax.py:55-81 at utils/evaluate.py:259-262).  Unfused, the [B, pred, H, W] logits are written by the
// predictor and read back by the soft-argmax -- 2 x 2 GB per 256-image pass of the C5 sweep next to the 2.1 GB of the
// activation itself -- and nothing else ever looks at them.  Here the logits exist in accumulator registers only:
//   * the predictor is a [pixels x cin] x [cin x 32] product on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32
//     fma chains like the convolutions): a wave takes 128 consecutive pixels per step -- lane (r, h) loads the 16-byte quad
//     of pixels 4r .. 4r+3 of input plane 2s + h for each K-step s -- and runs four accumulator tiles (pixel 4i + j in
//     tile j, row i), the filter column of output channel r staying in registers;
//   * the result layout puts ONE output channel on every lane (column = lane & 31) with 16 x 4 pixels in its registers,
//     so the online soft-max state (running maximum, sum e, sum e x, sum e y) is four registers per lane; the two lane
//     halves are merged at the end of the wave's pixel range and written as one partial per (image, wave, channel);
//   * pred_softargmax_combine_kernel merges the partials of a plane in fp64 and applies the reference's eps.
// Same NaN / +inf semantics as softargmax_kernel.  Needs H*W % 128 == 0, 16-byte aligned planes, cin in {8, 16, 32},
// cout <= 32 (ynet_pred_softargmax_supported); everything else stays on the two launches.
// ------------------------------------------------------------------------------------------------
struct PredSoftArgs {
    const float* x;
    long long x_bs;
    const float* w;         // [cout][cin]: the 1x1 filter as the checkpoint stores it
    const float* bias;      // [cout] or NULL
    float* partial;         // [B][nchunk][32][4] = (m, s, sx, sy)
    int cout, H, W, nchunk, gpw;      // gpw = 128-pixel groups per wave, nchunk = waves per image
};

template <int CIN, int PT>
__global__ __launch_bounds__(256, PT == 4 ? 2 : 3) void pred_softargmax_kernel(const PredSoftArgs a) {
    // PT = accumulator tiles (pixels per lane and K-step): a wave takes 32 * PT consecutive pixels per step.  PT = 4 (16-byte
    // loads, 236 registers, two waves per SIMD) is the default for cin = 32: 536 us per 256-image pass at C5 against 586 us for
    // PT = 2 (168 registers, three waves) -- occupancy is not what bounds it: MFMA time (0.24 ms) and the soft-max vector code
    // (as much again) of one wave run back to back, and the input alone takes 0.39 ms at the streaming rate.
    constexpr int KS = CIN / 2, GP = 32 * PT;
    typedef float vec_t __attribute__((ext_vector_type(PT)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
    const int wgs_per_img = (a.nchunk + 3) >> 2;
    const int b = (int)blockIdx.x / wgs_per_img, ch = ((int)blockIdx.x % wgs_per_img) * 4 + wave;
    if (ch >= a.nchunk) return;
    const int HW = a.H * a.W, ngroups = HW / GP;
    const int gpw = a.gpw * (128 / GP);          // (a.gpw counts 128-pixel groups)
    const int g_lo = ch * gpw, g_hi = min(ngroups, g_lo + gpw);
    float wreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[s] = r < a.cout ? a.w[r * CIN + 2 * s + half] : 0.f;
    const float bias_v = (a.bias != nullptr && r < a.cout) ? a.bias[r] : 0.f;
    const float* xb = a.x + (long long)b * a.x_bs + (long long)half * HW + PT * r;
    const bool row_uniform = (a.W % GP) == 0;       // a pixel group lies inside one image row
    SoftAcc st{-INFINITY, 0.f, 0.f, 0.f};
    bool poison = false;
    vec_t v[KS];
    auto load_group = [&](int g) {
        const float* p = xb + (long long)g * GP;
#pragma unroll
        for (int s = 0; s < KS; ++s) v[s] = *reinterpret_cast<const vec_t*>(p + (long long)(2 * s) * HW);
    };
    if (g_lo < g_hi) load_group(g_lo);
    for (int g = g_lo; g < g_hi; ++g) {
        f32x16 acc[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = bias_v;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s][j], wreg[s], acc[j], 0, 0, 0);
        if (g + 1 < g_hi) load_group(g + 1);        // in flight while this group's logits are folded
        const int p0 = g * GP;
        const int row0 = p0 / a.W, col0 = p0 - row0 * a.W;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = (q & 3) + 8 * (q >> 2) + 4 * half;      // row of the accumulator tile = index of the pixel PT-tuple
            int row = row0, col = col0 + PT * i;
            if (!row_uniform) {
                const int pq = p0 + PT * i;
                row = pq / a.W;
                col = pq - row * a.W;
            }
            float lv[PT], mx = -INFINITY, chk = 0.f;
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                lv[j] = acc[j][q];
                mx = fmaxf(mx, lv[j]);
                chk += lv[j];
            }
            poison = poison || chk != chk;
            if (mx > st.m) {            // rescale at most once per pixel tuple
                const float rs = exp_le0(st.m - mx);
                st.s *= rs;
                st.sx *= rs;
                st.sy *= rs;
                st.m = mx;
            }
            float es = 0.f, ew = 0.f;
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const float e = exp_le0(lv[j] - st.m);
                es += e;
                ew = __builtin_fmaf((float)j, e, ew);
            }
            st.s += es;
            st.sx += __builtin_fmaf((float)col, es, ew);
            st.sy += es * (float)row;
        }
    }
    if (poison) st.s = __builtin_nanf("");
    // merge the two lane halves (same channel, different pixels), then one partial per (image, wave, channel)
    const float om = __shfl_xor(st.m, 32), os = __shfl_xor(st.s, 32), osx = __shfl_xor(st.sx, 32), osy = __shfl_xor(st.sy, 32);
    const float m = fmaxf(st.m, om);
    const float ra = st.m == -INFINITY ? 0.f : exp_le0(st.m - m), rb = om == -INFINITY ? 0.f : exp_le0(om - m);
    if (half == 0) {
        f32x4 o;
        o[0] = m;
        o[1] = st.s * ra + os * rb;
        o[2] = st.sx * ra + osx * rb;
        o[3] = st.sy * ra + osy * rb;
        *reinterpret_cast<f32x4*>(a.partial + (((long long)b * a.nchunk + ch) * 32 + r) * 4) = o;
    }
}

__global__ __launch_bounds__(256) void pred_softargmax_combine_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                      long long B, int cout, int nchunk, float eps) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= B * cout) return;
    const long long b = i / cout;
    const int co = (int)(i - b * cout);
    const float* p = partial + (b * nchunk * 32 + co) * 4;
    float m = -INFINITY;
    for (int k = 0; k < nchunk; ++k) m = fmaxf(m, p[(long long)k * 128]);
    double s = 0.0, sx = 0.0, sy = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        const float* q = p + (long long)k * 128;
        const double rk = q[0] == -INFINITY ? 0.0 : (double)exp_le0(q[0] - m);
        s += (double)q[1] * rk;
        sx += (double)q[2] * rk;
        sy += (double)q[3] * rk;
    }
    if (m == INFINITY) s = (double)__builtin_nanf("");      // a +inf logit: exp(inf - inf) = NaN in the reference
    const double inv = 1.0 / (s + (double)eps);
    out[i * 2 + 0] = (float)(sx * inv);
    out[i * 2 + 1] = (float)(sy * inv);
}

// ------------------------------------------------------------------------------------------------
// sigmoid(x[:, sel] / T)   (utils/evaluate.py:128-131, models/ynet.py:585-586): channel gather fused
// ------------------------------------------------------------------------------------------------
struct SigArgs {
    const float* x;
    float* y;
    int sel[8];
    int nsel, C;
    long long B, HW;
    float T;
};

__global__ void sigmoid_temp_kernel(const SigArgs a) {
    const long long total = a.B * a.nsel * a.HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long px = i % a.HW;
        const int k = (int)((i / a.HW) % a.nsel);
        const long long b = i / (a.HW * a.nsel);
        const float v = a.x[(b * a.C + a.sel[k]) * a.HW + px] / a.T;
        a.y[i] = 1.f / (1.f + expf(-v));
    }
}

// ------------------------------------------------------------------------------------------------
// get_patch + torch.stack  (utils/image_utils.py:40-63; utils/train_epoch.py:63-78):
//   out[n,y,x] = tmpl[cy - ry_n + y][cx - rx_n + x],  r = rint(coord) (round-half-even = np.round)
// coords: [N,2] (x,y) fp32 on the device; `status` (int, device) is set to 1 if a window leaves the
// template (the reference would silently produce a ragged patch and fail in torch.stack).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_patch_kernel(const float* __restrict__ tmpl, int SH, int SW,
                                                           const float* __restrict__ xy, float* __restrict__ out,
                                                           int H, int W, int* __restrict__ status) {
    const int n = blockIdx.y;
    const int rx = (int)rintf(xy[2 * n]), ry = (int)rintf(xy[2 * n + 1]);
    const int ox = SW / 2 - rx, oy = SH / 2 - ry;
    float* o = out + (long long)n * H * W;
    const int total = H * W;
    if (ox < 0 || oy < 0 || ox + W > SW || oy + H > SH) {
        // flagged AND zero-filled: the caller's buffer is never left uninitialised (the flag is read at the next sync point)
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicExch(status, 1);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) o[i] = 0.f;
        return;
    }
    if ((W & 3) == 0 && ((reinterpret_cast<uintptr_t>(o)) & 15) == 0) {      // 16-byte stores (the window start is unaligned)
        const int Wq = W >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (total >> 2); i += gridDim.x * 256) {
            const int y = i / Wq, x = (i - y * Wq) << 2;
            const float* src = tmpl + (long long)(oy + y) * SW + ox + x;
            reinterpret_cast<float4*>(o)[i] = make_float4(src[0], src[1], src[2], src[3]);
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        o[i] = tmpl[(long long)(oy + y) * SW + ox + x];
    }
}


// ------------------------------------------------------------------------------------------------
// get_patch WITHOUT the S x S templates: the window of create_dist_mat around (rx, ry) is the
// normalised Euclidean distance to that point, and the window of the Gaussian template is the kernlen x kernlen blob
// (a 3.8 KB table) placed at it -- nothing else of the 4.4 / 7.7 MB templates is ever read.
//   kind 0: out[n,y,x] = (float)( sqrt((double)((y-ry)^2 + (x-rx)^2)) / dmax * 2 )      utils/image_utils.py:30-37
//           (integer radicand, IEEE fp64 sqrt and division, then ONE rounding to fp32: bit-identical to the float64
//            NumPy template cast by torch.Tensor(...); dmax = the template's maximum = sqrt(2) * (S // 2) in fp64)
//   kind 1: out[n,y,x] = blob[y-ry+m/2][x-rx+m/2] inside the blob, 0 elsewhere                utils/image_utils.py:15-27
// The window must still lie inside the virtual template (the reference slices it): flagged + zero-filled otherwise.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heatmap_analytic_kernel(const float* __restrict__ xy, float* __restrict__ out, int H, int W,
                                                               int S, int kind, double dmax, const float* __restrict__ blob,
                                                               int m, int* __restrict__ status) {
    const int n = blockIdx.y;
    const int rx = (int)rintf(xy[2 * n]), ry = (int)rintf(xy[2 * n + 1]);
    const int ox = S / 2 - rx, oy = S / 2 - ry;
    float* o = out + (long long)n * H * W;
    const int total = H * W;
    if (ox < 0 || oy < 0 || ox + W > S || oy + H > S) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicExch(status, 1);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) o[i] = 0.f;
        return;
    }
    auto value = [&](int y, int x) -> float {
        const int dy = y - ry, dx = x - rx;
        if (kind == 0) return (float)(sqrt((double)((long long)dy * dy + (long long)dx * dx)) / dmax * 2.0);
        // the template carries the blob at [S/2 - m/2, S/2 + (m+1)/2): template index S/2 - ry + y -> blob row dy + m/2
        const int by = dy + m / 2, bx = dx + m / 2;
        return (by >= 0 && by < m && bx >= 0 && bx < m) ? blob[by * m + bx] : 0.f;
    };
    if ((W & 3) == 0 && ((reinterpret_cast<uintptr_t>(o)) & 15) == 0) {
        const int Wq = W >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (total >> 2); i += gridDim.x * 256) {
            const int y = i / Wq, x = (i - y * Wq) << 2;
            reinterpret_cast<float4*>(o)[i] = make_float4(value(y, x), value(y, x + 1), value(y, x + 2), value(y, x + 3));
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / W;
        o[i] = value(y, i - y * W);
    }
}

// ------------------------------------------------------------------------------------------------
// TTST (utils/evaluate.py:134-161, utils/kmeans.py:22-108): Lloyd's k-means of the N = 10000 goal samples of one
// person, one workgroup per person, the whole iteration on chip.  Points are pixel coordinates (integer valued), so
// cluster sums are exact in int32 whatever the summation order and the result does not depend on the thread count:
//   assign:  d_j = fl(fl(dx*dx) + fl(dy*dy)), first minimum wins          (pairwise_distance + argmin)
//   update:  c_j = sum_j / n_j (one fp32 division per coordinate)          (selected.mean(dim=0))
//   stop:    (sum_j sqrt(fl(ddx^2 + ddy^2)))^2 < tol, or iter_limit         (center_shift ** 2 < tol)
// An empty cluster needs the reference's torch.randint re-seed: the kernel reports it (status 1) and the host
// runs that person on its slow path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void kmeans2d_kernel(const float* __restrict__ points, const int* __restrict__ init_idx,
                                                        float* __restrict__ centers, int* __restrict__ status, int N, int K,
                                                        float tol, int iter_limit) {
    extern __shared__ int klds[];
    int* px = klds;                 // [N]
    int* py = klds + N;             // [N]
    float* cx = reinterpret_cast<float*>(klds + 2 * N);      // [32]
    float* cy = cx + 32;                                     // [32]
    int* sums = reinterpret_cast<int*>(cy + 32);             // [32][3]: sum x, sum y, count
    int* flag = sums + 96;                                   // [2]: 0 running, 1 converged / limit, 2 empty cluster
    const int person = blockIdx.x, tid = threadIdx.x;
    const float* P = points + (long long)person * N * 2;
    for (int i = tid; i < N; i += blockDim.x) {
        px[i] = (int)rintf(P[2 * i]);
        py[i] = (int)rintf(P[2 * i + 1]);
    }
    __syncthreads();
    if (tid < K) {
        const int j = init_idx[person * K + tid];
        cx[tid] = (float)px[j];
        cy[tid] = (float)py[j];
    }
    if (tid == 0) flag[0] = 0;
    int it = 0;
    for (;;) {
        if (tid < 3 * K) sums[tid] = 0;
        __syncthreads();
        for (int i = tid; i < N; i += blockDim.x) {
            const float x = (float)px[i], y = (float)py[i];
            int best = 0;
            float bd = __fadd_rn(__fmul_rn(__fsub_rn(x, cx[0]), __fsub_rn(x, cx[0])), __fmul_rn(__fsub_rn(y, cy[0]), __fsub_rn(y, cy[0])));
            for (int j = 1; j < K; ++j) {
                const float dx = __fsub_rn(x, cx[j]), dy = __fsub_rn(y, cy[j]);
                const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                if (d < bd) {
                    bd = d;
                    best = j;
                }
            }
            atomicAdd(&sums[3 * best], px[i]);
            atomicAdd(&sums[3 * best + 1], py[i]);
            atomicAdd(&sums[3 * best + 2], 1);
        }
        __syncthreads();
        if (tid == 0) {
            float shift = 0.f;
            int state = 0;
            for (int j = 0; j < K; ++j) {
                const int n = sums[3 * j + 2];
                if (n == 0) {
                    state = 2;
                    break;
                }
                const float nx = (float)sums[3 * j] / (float)n, ny = (float)sums[3 * j + 1] / (float)n;
                const float ddx = __fsub_rn(nx, cx[j]), ddy = __fsub_rn(ny, cy[j]);
                shift = __fadd_rn(shift, sqrtf(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy))));
                cx[j] = nx;
                cy[j] = ny;
            }
            ++it;
            if (state == 0 && (__fmul_rn(shift, shift) < tol || (iter_limit != 0 && it >= iter_limit))) state = 1;
            flag[0] = state;
            flag[1] = it;
        }
        __syncthreads();
        if (flag[0] != 0) break;
    }
    if (tid < K) {
        centers[((long long)person * K + tid) * 2] = cx[tid];
        centers[((long long)person * K + tid) * 2 + 1] = cy[tid];
    }
    if (tid == 0) status[person] = (flag[0] == 2 ? 1 : 0) | (flag[1] << 8);
}

// ------------------------------------------------------------------------------------------------
// Scene pre-processing, the part that needs neither OpenCV nor the segmentation backbone:
//   pad(images, division_factor)                          utils/image_utils.py:95-107  zero border at the bottom / right up to a
//                                                         multiple of the division factor (cv2.copyMakeBorder, BORDER_CONSTANT)
//   preprocess_image_for_segmentation(seg_mask = True)    utils/image_utils.py:74-81   one-hot planes of a label map
// The reference pads the LABEL map first (the border gets label 0) and one-hot encodes afterwards, so the border belongs to
// class 0; the fused kernel reproduces exactly that.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad2d_kernel(const float* __restrict__ x, float* __restrict__ y, long long N, int H, int W, int Hp, int Wp) {
    const long long total = N * Hp * Wp;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int px = (int)(i % Wp), py = (int)((i / Wp) % Hp);
        const long long n = i / ((long long)Wp * Hp);
        y[i] = (py < H && px < W) ? x[(n * H + py) * W + px] : 0.f;
    }
}

__global__ __launch_bounds__(256) void seg_onehot_pad_kernel(const int* __restrict__ lab, float* __restrict__ y, int H, int W, int Hp, int Wp, int C) {
    const long long total = (long long)C * Hp * Wp;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int px = (int)(i % Wp), py = (int)((i / Wp) % Hp), c = (int)(i / ((long long)Wp * Hp));
        const int v = (py < H && px < W) ? lab[(long long)py * W + px] : 0;      // the padded border carries label 0
        y[i] = v == c ? 1.f : 0.f;
    }
}

// resize(images, factor, seg_mask = True) (utils/image_utils.py:83-87): cv2.resize(..., INTER_NEAREST) of a label map, restated from
// OpenCV's published rule (imgproc resize.cpp, resizeNN): destination size cvRound(src * f) (the caller computes it, half-to-even),
// source column of destination column x = min(cvFloor(x * (1 / fx)), W - 1) with the product in double, rows alike.  PARITY UNPINNED:
// cv2 is absent from the image (like loralib), nothing reference-held can pin it; known-answer tests only.
__global__ __launch_bounds__(256) void resize_nearest_kernel(const int* __restrict__ src, int* __restrict__ dst, int H, int W, int Ho, int Wo,
                                                            double ifx, double ify) {
    const long long total = (long long)Ho * Wo;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)(i / Wo);
        int sx = (int)floor((double)x * ifx), sy = (int)floor((double)y * ify);
        sx = sx < W - 1 ? sx : W - 1;
        sy = sy < H - 1 ? sy : H - 1;
        dst[i] = src[(long long)sy * W + sx];
    }
}

// augment_data's image side (utils/data_utils.py:113-170): cv2.rotate(im, ROTATE_90_COUNTERCLOCKWISE) applied k times (= np.rot90(im, k))
// and cv2.flip(im, 1) (= np.fliplr) are pure index permutations -- N planes [H][W] of 32-bit words (int32 label maps, fp32 planes alike) ->
// [Ho][Wo], (Ho, Wo) = (W, H) for odd k.  One rotation: out[i][j] = in[j][W - 1 - i]; the flip is applied after the rotations.
__global__ __launch_bounds__(256) void rot90_flip_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, long long N, int H, int W, int k, int flip) {
    const int Wo = (k & 1) ? H : W;
    const long long plane = (long long)H * W, total = N * plane;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long n = t / plane;
        const int r = (int)(t - n * plane), i = r / Wo;
        int j = r - i * Wo;
        if (flip) j = Wo - 1 - j;
        int sy, sx;
        switch (k & 3) {
            case 0: sy = i; sx = j; break;
            case 1: sy = j; sx = W - 1 - i; break;
            case 2: sy = H - 1 - i; sx = W - 1 - j; break;
            default: sy = H - 1 - j; sx = i; break;
        }
        dst[t] = src[n * plane + (long long)sy * W + sx];
    }
}

// augment_data's coordinate side: rot() maps (x, y) -> ((x - x0 / 2, y - y0 / 2) . R) + (x0' / 2, y0' / 2) with R = [[c, s], [-s, c]],
// c = cos(-k pi / 2), s = sin(-k pi / 2) AS NumPy EVALUATES THEM (c is 6.1e-17, not 0, for odd k: the caller passes the doubles), fliplr()
// the same with R = [[-1, 0], [0, 1]]; float64 like the DataFrame columns.  xy [n][2] in place.
__global__ __launch_bounds__(256) void rot_coords_kernel(double* __restrict__ xy, long long n, double cx, double cy, double r00, double r01, double r10, double r11,
                                                         double ox, double oy) {
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const double x = xy[2 * t] - cx, y = xy[2 * t + 1] - cy;
        xy[2 * t] = (x * r00 + y * r10) + ox;
        xy[2 * t + 1] = (x * r01 + y * r11) + oy;
    }
}

// ------------------------------------------------------------------------------------------------
// The serial adapters' element-wise tail (models/ynet.py:24-26,64-66,117-131: nn.BatchNorm2d in front of the adapter's 1x1 conv, the residual add,
// the ReLU behind the sum) -- SURVEY 8(f)-2, outside every BASELINE configuration; round 6 moves them off ATen.
//   bn_reduce_kernel   per channel the sums a training-mode BatchNorm2d needs, over B x HW elements, as fp64 partials of 64 fixed slices per
//                      channel (bitwise reproducible): MODE 0 (sum x, sum x^2), MODE 1 (sum dy, sum dy * xhat)
//   bn_finish_kernel   mean, 1 / sqrt(var + eps) and the running statistics (momentum, unbiased variance) from the partials: F.batch_norm's rule
//   bn_apply_kernel    y = (x - mean) * invstd * gamma + beta
//   bn_bwd_kernel      training: dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)); evaluation: dx = dy * gamma * invstd
//   add_relu_kernel / relu_bwd_kernel   y = [relu](a + b); dx = y > 0 ? dy : 0
// ------------------------------------------------------------------------------------------------
#define BN_PARTS 64

template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, double* __restrict__ partial, int B, int C, long long HW) {
    __shared__ double w0[4], w1[4];
    const int c = blockIdx.x / BN_PARTS, part = blockIdx.x % BN_PARTS;
    const long long per = ((long long)B * HW + BN_PARTS - 1) / BN_PARTS, lo = part * per, hi = min(lo + per, (long long)B * HW);
    const float m = MODE == 1 ? mean[c] : 0.f, is = MODE == 1 ? invstd[c] : 0.f;
    double s0 = 0.0, s1 = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const long long b = i / HW, p = i - b * HW;
        const float xv = x[(b * C + c) * HW + p];
        if (MODE == 0) {
            s0 += (double)xv;
            s1 += (double)xv * (double)xv;
        } else {
            const float g = dy[(b * C + c) * HW + p];
            s0 += (double)g;
            s1 += (double)g * (double)((xv - m) * is);
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { w0[threadIdx.x >> 6] = s0; w1[threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (w0[0] + w0[1]) + (w0[2] + w0[3]);
        partial[2 * blockIdx.x + 1] = (w1[0] + w1[1]) + (w1[2] + w1[3]);
    }
}

__global__ void bn_finish_kernel(const double* __restrict__ partial, float* mean, float* invstd, float* running_mean, float* running_var, int C, double n, double eps,
                                 double momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int p = 0; p < BN_PARTS; ++p) { s0 += partial[2 * (c * BN_PARTS + p)]; s1 += partial[2 * (c * BN_PARTS + p) + 1]; }
    const double m = s0 / n;
    double var = s1 / n - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + eps));
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * (n > 1.0 ? var * n / (n - 1.0) : var));
}

__global__ void bn_invstd_kernel(const float* __restrict__ running_var, float* invstd, int C, double eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = (float)(1.0 / sqrt((double)running_var[c] + eps));
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, long long planes, int C,
                                                       long long HW) {
    const long long plane = blockIdx.y;
    if (plane >= planes) return;
    const int c = (int)(plane % C);
    const float m = mean[c], sc = invstd[c] * (gamma ? gamma[c] : 1.f), sh = beta ? beta[c] : 0.f;
    for (long long p = blockIdx.x * 256ll + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) y[plane * HW + p] = (x[plane * HW + p] - m) * sc + sh;
}

__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const double* __restrict__ partial, float* __restrict__ dx, float* dgamma, float* dbeta,
                                                     long long planes, int C, long long HW, double n, int train) {
    const long long plane = blockIdx.y;
    if (plane >= planes) return;
    const int c = (int)(plane % C);
    double s0 = 0.0, s1 = 0.0;
    if (partial) {
        for (int p = 0; p < BN_PARTS; ++p) { s0 += partial[2 * (c * BN_PARTS + p)]; s1 += partial[2 * (c * BN_PARTS + p) + 1]; }
        if (plane < C && blockIdx.x == 0 && threadIdx.x == 0) {
            if (dgamma) dgamma[c] = (float)s1;
            if (dbeta) dbeta[c] = (float)s0;
        }
    }
    const float m = mean[c], is = invstd[c], g = (gamma ? gamma[c] : 1.f) * is;
    const float mdy = train ? (float)(s0 / n) : 0.f, mdx = train ? (float)(s1 / n) : 0.f;
    for (long long p = blockIdx.x * 256ll + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
        const float gy = dy[plane * HW + p];
        dx[plane * HW + p] = train ? g * (gy - mdy - (x[plane * HW + p] - m) * is * mdx) : g * gy;
    }
}

__global__ __launch_bounds__(256) void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n, int relu) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = a[i] + b[i];
        y[i] = relu ? (v < 0.f ? 0.f : v) : v;      // (a NaN stays a NaN, as torch.relu)
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// The outermost ring of an up-convolution's low-resolution data gradient (ynet_upconv_dgrad_ring, include/ynet_hip.h): what the bilinear clamp and the zero padding of
// the up-sampled image add to the effective-filter convolution -- a 1 x 3 (rows 0, h-1) / 3 x 1 (columns 0, w-1) data gradient over D's border lines plus a term at the
// corners.  A workgroup owns one SEGMENT of 64 consecutive pixels of one border line of one image: the line's three tables [3][C4][cin] and the D tile [C4][66] go to
// LDS and the 3 C4-deep products run on the fp32 matrix cores (a wave: 16 pixels x 16 input channels per v_mfma_f32_16x16x4_f32 chain).  Row segments
// cover all columns and also add the column and corner terms of the two corner pixels (computed by the whole workgroup, 3 C4 + C4 products per channel); column
// segments cover rows 1 .. h-2.  Every ring pixel is written by exactly one workgroup.
#define RING_SEG 64
__global__ __launch_bounds__(256) void upconv_ring_kernel(const float* __restrict__ D, long long d_bs, const float* __restrict__ tab, const float* __restrict__ relu_of,
                                                          long long relu_bs, float* __restrict__ dx, long long dx_bs, int B, int C4, int cin, int h, int w,
                                                          int nseg_row, int nseg_col) {
    extern __shared__ float ring_smem[];
    float* T = ring_smem;                                  // [3][C4][cin]
    float* Dt = T + 3 * C4 * cin;                          // [C4][RING_SEG + 2]
    float* red = Dt + C4 * (RING_SEG + 2);                 // [8][cin]: partial sums of a corner's extra terms
    const int per_image = 2 * nseg_row + 2 * nseg_col;
    const int b = blockIdx.x / per_image, which = blockIdx.x % per_image;
    const bool is_row = which < 2 * nseg_row;
    const int side = is_row ? which / nseg_row : (which - 2 * nseg_row) / nseg_col;          // 0: row 0 / column 0; 1: row h-1 / column w-1
    const int seg = is_row ? which % nseg_row : (which - 2 * nseg_row) % nseg_col;
    const int len = is_row ? w : h;                        // pixels of the border line
    const int q0 = is_row ? seg * RING_SEG : 1 + seg * RING_SEG;      // first pixel of the segment along the line
    const int q_end = is_row ? w : h - 1;                  // column segments leave the corners to the rows
    const int fixed = is_row ? (side ? h - 1 : 0) : (side ? w - 1 : 0);
    const long long hw = (long long)h * w, tstride = (long long)C4 * cin;
    const float* Db = D + (long long)b * d_bs;
    const int tid = threadIdx.x;
    const float* tsrc = tab + (long long)((is_row ? 0 : 6) + side * 3) * tstride;
    for (int i = tid; i < 3 * C4 * cin; i += 256) T[i] = tsrc[i];
    for (int i = tid; i < C4 * (RING_SEG + 2); i += 256) {
        const int c = i / (RING_SEG + 2), k = i - c * (RING_SEG + 2), q = q0 - 1 + k;
        Dt[i] = (q >= 0 && q < len) ? Db[c * hw + (is_row ? (long long)fixed * w + q : (long long)q * w + fixed)] : 0.f;
    }
    __syncthreads();
    // out[ci][pixel] = sum over k = (tap, c') of table[k][ci] * D[c'][pixel - tap + 1] on the fp32 matrix cores: a wave owns 16 pixels of the segment, the A operand is
    // the table (row = input channel), the B operand the D tile (column = pixel): lane (m, kq) ends up with channels 4 kq .. 4 kq + 3 of pixel m -- consecutive lanes,
    // consecutive pixels of a border row
    const int lane = tid & 63, wv = tid >> 6, m = lane & 15, kq = lane >> 4;
    const int q = q0 + wv * 16 + m;
    for (int ct = 0; ct * 16 < cin; ++ct) {                // tiles of 16 input channels
        const int cia = ct * 16 + m;                       // the A operand's channel
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < 3; ++t)
            for (int s4 = 0; s4 < C4; s4 += 4) {
                const int c = s4 + kq;
                const float av = cia < cin ? T[((long long)t * C4 + c) * cin + cia] : 0.f;
                const float bv = Dt[c * (RING_SEG + 2) + wv * 16 + m + 2 - t];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
        if (q < q_end) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = ct * 16 + 4 * kq + e;
                if (ci >= cin) break;
                const long long o = (long long)ci * hw + (is_row ? (long long)fixed * w + q : (long long)q * w + fixed);
                if (relu_of == nullptr || relu_of[(long long)b * relu_bs + o] > 0.f) dx[(long long)b * dx_bs + o] += acc[e];
            }
        }
    }
    if (!is_row) return;
    // ---- the corner pixels of a row segment: + the column term (3 taps along the column) + the corner term, by the whole workgroup
    for (int corner = 0; corner < 2; ++corner) {
        const int jc = corner ? w - 1 : 0;
        if (jc < q0 || jc >= q0 + RING_SEG) continue;      // (workgroup-uniform)
        const float* tcol = tab + (long long)(6 + corner * 3) * tstride;
        const float* tx = tab + (long long)(12 + side * 2 + corner) * tstride;
        const int nterm = 4 * C4;                          // (tap 0..2 of the column term, 3 = the corner term) x c'
        for (int ci = tid & 31; ci < cin; ci += 32) {
            float part = 0.f;
            for (int k = tid >> 5; k < nterm; k += 8) {
                const int t = k / C4, c = k - t * C4;
                float d = 0.f;
                if (t < 3) {
                    const int ii = fixed - t + 1;
                    if (ii >= 0 && ii < h) d = Db[c * hw + (long long)ii * w + jc] * tcol[((long long)t * C4 + c) * cin + ci];
                } else {
                    d = Db[c * hw + (long long)fixed * w + jc] * tx[(long long)c * cin + ci];
                }
                part += d;
            }
            __syncthreads();
            red[(tid >> 5) * 32 + (tid & 31)] = part;
            __syncthreads();
            if (tid < 32) {
                float sum = 0.f;
                for (int k = 0; k < 8; ++k) sum += red[k * 32 + tid];
                const long long o = (long long)ci * hw + (long long)fixed * w + jc;
                if (relu_of == nullptr || relu_of[(long long)b * relu_bs + o] > 0.f) dx[(long long)b * dx_bs + o] += sum;
            }
        }
    }
}

extern "C" {

int ynet_maxpool2_fwd(const float* x, float* y, long long N, int H, int W, void* stream) {
    YNET_REQUIRE(x && y && N > 0 && H >= 2 && W >= 2, "maxpool2_fwd: bad arguments");
    const long long total = N * (H / 2) * (W / 2);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, plane_grid(N, total / N), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W);
    return ynet_check_launch("maxpool2_fwd");
}

int ynet_maxpool2_bwd(const float* x, const float* dy, float* dx, long long N, int H, int W, void* stream) {
    YNET_REQUIRE(x && dy && dx && N > 0 && H >= 2 && W >= 2, "maxpool2_bwd: bad arguments");
    const long long total = N * ((H + 1) / 2) * ((W + 1) / 2);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H, W);
    return ynet_check_launch("maxpool2_bwd");
}

int ynet_maxpool2_bwd_add(const float* x, const float* dy, const float* add0, const float* add1, float* dx, long long N,
                          int H, int W, int relu_mask, void* stream) {
    YNET_REQUIRE(x && dy && dx && N > 0 && H >= 2 && W >= 2, "maxpool2_bwd_add: bad arguments");
    YNET_REQUIRE((H & 1) == 0 && (W & 1) == 0, "maxpool2_bwd_add: H and W must be even (got %dx%d)", H, W);
    YNET_REQUIRE((((uintptr_t)x | (uintptr_t)dx | (uintptr_t)add0 | (uintptr_t)add1) & 7) == 0, "maxpool2_bwd_add: 8-byte aligned planes required");
    const long long total = N * (H / 2) * (W / 2);
    if (relu_mask)
        hipLaunchKernelGGL(maxpool2_bwd_add_kernel<true>, plane_grid(N, total / N), dim3(256), 0, (hipStream_t)stream, x, dy, add0, add1, dx, N, H, W);
    else
        hipLaunchKernelGGL(maxpool2_bwd_add_kernel<false>, plane_grid(N, total / N), dim3(256), 0, (hipStream_t)stream, x, dy, add0, add1, dx, N, H, W);
    return ynet_check_launch("maxpool2_bwd_add");
}

int ynet_maxpool2_bwd_add_code(const unsigned char* code, const float* dy, const float* add0, const float* add1, float* dx, long long N, int H, int W,
                               int relu_mask, void* stream) {
    YNET_REQUIRE(code && dy && dx && N > 0 && H >= 2 && W >= 2, "maxpool2_bwd_add_code: bad arguments");
    YNET_REQUIRE((H & 1) == 0 && (W & 1) == 0, "maxpool2_bwd_add_code: H and W must be even (got %dx%d)", H, W);
    YNET_REQUIRE((((uintptr_t)dx | (uintptr_t)add0 | (uintptr_t)add1) & 7) == 0, "maxpool2_bwd_add_code: 8-byte aligned planes required");
    const long long total = N * (H / 2) * (W / 2);
    const float* c = reinterpret_cast<const float*>(code);
    if (relu_mask)
        hipLaunchKernelGGL((maxpool2_bwd_add_kernel<true, true>), plane_grid(N, total / N), dim3(256), 0, (hipStream_t)stream, c, dy, add0, add1, dx, N, H, W);
    else
        hipLaunchKernelGGL((maxpool2_bwd_add_kernel<false, true>), plane_grid(N, total / N), dim3(256), 0, (hipStream_t)stream, c, dy, add0, add1, dx, N, H, W);
    return ynet_check_launch("maxpool2_bwd_add_code");
}

int ynet_upsample2x_fwd(const float* x, float* y, long long N, int H, int W, void* stream) {
    YNET_REQUIRE(x && y && N > 0 && H > 0 && W > 0, "upsample2x_fwd: bad arguments");
    const bool rows_ok = (W & 1) == 0 && (((uintptr_t)y) & 15) == 0 && (((uintptr_t)x) & 7) == 0;
    if (rows_ok && (H & 3) == 0 && (long long)H * W >= 64 * 64)
        hipLaunchKernelGGL(upsample2x_fwd_rows_kernel<4>, plane_grid(N, (long long)(H / 4) * (W / 2)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W);
    else if (rows_ok && (H & 1) == 0)
        hipLaunchKernelGGL(upsample2x_fwd_rows_kernel<2>, plane_grid(N, (long long)(H / 2) * (W / 2)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W);
    else if ((W & 1) == 0 && ((uintptr_t)y & 15) == 0)
        hipLaunchKernelGGL(upsample2x_fwd_vec_kernel, plane_grid(N, (long long)H * W), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W);
    else
        hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for(N * H * W * 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W);
    return ynet_check_launch("upsample2x_fwd");
}

static int upsample2x_bwd_impl(const float* dy, float* dx, const float* relu_of, long long N, int H, int W, void* stream) {
    YNET_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0, "upsample2x_bwd: bad arguments");
    const bool act8 = relu_of == nullptr || (((uintptr_t)relu_of) & 7) == 0, act16 = relu_of == nullptr || (((uintptr_t)relu_of) & 15) == 0;
    const bool rows_ok = (W & 1) == 0 && (((uintptr_t)dy) & 15) == 0 && (((uintptr_t)dx) & 7) == 0 && act8;
    if (rows_ok && (H & 3) == 0 && (long long)H * W >= 64 * 64)
        hipLaunchKernelGGL(upsample2x_bwd_rows_kernel<4>, plane_grid(N, (long long)(H / 4) * (W / 2)), dim3(256), 0, (hipStream_t)stream, dy, dx, relu_of, N, H, W);
    else if (rows_ok && (H & 1) == 0)
        hipLaunchKernelGGL(upsample2x_bwd_rows_kernel<2>, plane_grid(N, (long long)(H / 2) * (W / 2)), dim3(256), 0, (hipStream_t)stream, dy, dx, relu_of, N, H, W);
    else if ((W & 3) == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0 && act16)
        hipLaunchKernelGGL(upsample2x_bwd_vec_kernel, plane_grid(N, (long long)H * (W / 4)), dim3(256), 0, (hipStream_t)stream, dy, dx, relu_of, N, H, W);
    else
        hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(N * H * W, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, relu_of, N, H, W);
    return ynet_check_launch("upsample2x_bwd");
}

int ynet_upsample2x_bwd(const float* dy, float* dx, long long N, int H, int W, void* stream) {
    return upsample2x_bwd_impl(dy, dx, nullptr, N, H, W, stream);
}

// ... with the ReLU backward of the up-sampled activation applied to dx: dx = relu_of > 0 ? dx : 0 (relu_of [N][H][W], contiguous)
int ynet_upsample2x_bwd_relu(const float* dy, float* dx, const float* relu_of, long long N, int H, int W, void* stream) {
    YNET_REQUIRE(relu_of != nullptr, "upsample2x_bwd_relu: the activation is null");
    return upsample2x_bwd_impl(dy, dx, relu_of, N, H, W, stream);
}

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam / AdamW step (models/trainer.py:182) for a captured training step, all parameters in TWO launches: torch's
// fused multi-tensor form takes 6 launches = 0.18 ms for the 110 tensors of a fully trainable Y-Net (C1), alone on the GPU at the end
// of every step.  Same update rule and the same precision choices as torch's fused kernel (fused_adam_utils.cuh): bias corrections
// and the second-moment update in double, everything else in fp32; `step` counters stay the per-parameter fp32 device tensors of
// optimizer.state (incremented by the first launch), so optimizer.state_dict() and torch's own step() see a consistent state.
//   table [6][ntensors] of 64-bit values: param, grad, exp_avg, exp_avg_sq, step (pointers), numel
//   chunk c (1024 elements): tensor chunk_tensor[c], first element chunk_first[c]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_steps_kernel(const long long* __restrict__ table, int ntensors) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ntensors) {
        float* st = reinterpret_cast<float*>(table[4ll * ntensors + i]);
        *st += 1.f;
    }
}

__global__ __launch_bounds__(256) void adam_update_kernel(const long long* __restrict__ table, const int* __restrict__ chunk_tensor,
                                                          const long long* __restrict__ chunk_first, int ntensors, double lr, double beta1,
                                                          double beta2, double eps, double weight_decay, int adamw) {
    const int t = chunk_tensor[blockIdx.x];
    float* __restrict__ p = reinterpret_cast<float*>(table[t]);
    const float* __restrict__ g = reinterpret_cast<const float*>(table[1ll * ntensors + t]);
    float* __restrict__ m = reinterpret_cast<float*>(table[2ll * ntensors + t]);
    float* __restrict__ v = reinterpret_cast<float*>(table[3ll * ntensors + t]);
    const double step = (double)*reinterpret_cast<const float*>(table[4ll * ntensors + t]);
    const long long n = table[5ll * ntensors + t];
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - beta1);
    const long long i0 = chunk_first[blockIdx.x] + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + k * 256;
        if (i < n) {
            float pv = p[i], gv = g[i], mv = m[i], vv = v[i];
            if (weight_decay != 0.0) {
                if (adamw) pv -= (float)(lr * weight_decay) * pv;
                else gv += (float)((double)pv * weight_decay);
            }
            mv = mv + w1 * (gv - mv);                                            // lerp(exp_avg, grad, 1 - beta1), weight < 0.5
            vv = (float)(beta2 * (double)vv + (1.0 - beta2) * (double)gv * (double)gv);
            const float denom = (float)((double)(sqrtf(vv) / bc2_sqrt) + eps);
            pv -= step_size * mv / denom;
            p[i] = pv;
            m[i] = mv;
            v[i] = vv;
        }
    }
}

int ynet_adam_step(const long long* table, const int* chunk_tensor, const long long* chunk_first, int ntensors, int nchunks,
                   double lr, double beta1, double beta2, double eps, double weight_decay, int adamw, void* stream) {
    YNET_REQUIRE(table && chunk_tensor && chunk_first && ntensors > 0 && nchunks > 0, "adam_step: bad arguments");
    YNET_REQUIRE(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "adam_step: bad hyper-parameters");
    hipLaunchKernelGGL(adam_steps_kernel, dim3((ntensors + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, ntensors);
    hipLaunchKernelGGL(adam_update_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_first, ntensors, lr,
                       beta1, beta2, eps, weight_decay, adamw);
    return ynet_check_launch("adam_step");
}

// y[i] = sum_b x[b][i] in batch order: the backward of a batch-broadcast conv input (models/ynet.py:87 expands ONE scene to the batch;
// torch's reduce kernel runs this [B][n] -> [n] sum at ~1.1 TB/s, 0.8 ms per C4 step)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, float* __restrict__ y, int B, long long n4, long long bs4) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 a = x4[i];
        for (int b = 1; b < B; ++b) {
            const float4 v = x4[i + (long long)b * bs4];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        y4[i] = a;
    }
}

int ynet_batch_sum(const float* x, float* y, int B, long long n, long long batch_stride, void* stream) {
    YNET_REQUIRE(x && y && B > 0 && n > 0, "batch_sum: bad arguments");
    YNET_REQUIRE((n & 3) == 0 && (batch_stride & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0, "batch_sum: 16-byte aligned rows of a multiple of 4 floats");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, n / 4, batch_stride / 4);
    return ynet_check_launch("batch_sum");
}

int ynet_avgpool_pyramid(const float* x, float* const* outs, int nlev, long long N, int H, int W, void* stream) {
    YNET_REQUIRE(x && outs && nlev >= 1 && nlev <= 5, "avgpool_pyramid: 1..5 levels supported (got %d)", nlev);
    YNET_REQUIRE(N > 0 && H % 32 == 0 && W % 32 == 0 && H > 0 && W > 0,
                 "avgpool_pyramid: H and W must be multiples of 32 (got %dx%d)", H, W);
    PyrArgs a{};
    a.x = x;
    for (int i = 0; i < nlev; ++i) {
        YNET_REQUIRE(outs[i] != nullptr, "avgpool_pyramid: output %d is null", i);
        a.out[i] = outs[i];
    }
    a.nlev = nlev;
    a.H = H;
    a.W = W;
    a.N = N;
    const long long nblk = N * (H / 32) * (W / 32);
    hipLaunchKernelGGL(avgpool_pyramid_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
    return ynet_check_launch("avgpool_pyramid");
}

#define YNET_BCE_PARTS 1024

long long ynet_bce_workspace_bytes(void) { return YNET_BCE_PARTS * (long long)sizeof(double); }

/* workspace: [YNET_BCE_PARTS] doubles + a ticket counter that must be ZERO before the first launch (the kernel resets it) */
long long ynet_pred_bce_workspace_bytes(void) { return YNET_BCE_PARTS * (long long)sizeof(double) + 16; }

static int pred_bce_launch(const float* x, long long x_batch_stride, const float* wp, const float* bias, const float* target, const float* t_xy,
                           const float* t_blob, int t_m, int t_S, int t_H, int t_W, float* y, float* loss, float* dx, float* dy, void* workspace, int B,
                           int cin, int cout, long long HW, float expected_grad, int dx_relu_mask, void* stream) {
    YNET_REQUIRE(x && wp && (target || t_xy) && y && loss && workspace, "pred_bce: null pointer");
    YNET_REQUIRE(!t_xy || (t_blob && t_m > 0 && t_m <= t_S && t_H > 0 && t_W > 0 && (long long)t_H * t_W == HW && (t_W & 3) == 0 && t_S >= t_H && t_S >= t_W),
                 "pred_bce_blob: the target needs a blob table, 0 < kernlen <= S, H * W == HW, W %% 4 == 0 and S >= H, W (got kernlen %d, S %d, %dx%d, HW %lld)", t_m,
                 t_S, t_H, t_W, HW);
    YNET_REQUIRE(!dx_relu_mask || (dx != nullptr && cin <= 32), "pred_bce: the ReLU mask of dx needs dx and cin <= 32 (got cin = %d)", cin);
    YNET_REQUIRE(B > 0 && cin > 0 && cout > 0 && cout <= 32 && HW > 0 && (HW & 3) == 0, "pred_bce: bad shape B=%d cin=%d cout=%d HW=%lld (cout <= 32, HW %% 4 == 0)", B, cin, cout, HW);
    YNET_REQUIRE((((uintptr_t)x | (uintptr_t)target | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)dy) & 15) == 0 && (x_batch_stride & 3) == 0,
                 "pred_bce: tensors must be 16-byte aligned");
    YNET_REQUIRE(t_xy == nullptr || HW < (1ll << 31), "pred_bce_blob: H * W must stay below 2^31");
    PredBceArgs a{};
    a.x = x;
    a.x_bs = x_batch_stride;
    a.wp = wp;
    a.bias = bias;
    a.t = target;
    a.t_xy = t_xy;
    a.t_blob = t_blob;
    a.t_m = t_m;
    a.t_S = t_S;
    a.t_H = t_H;
    a.t_W = t_W;
    a.y = y;
    a.dx = dx;
    a.relu_mask = dx_relu_mask ? 1 : 0;
    a.dy = dy;
    a.partial = (double*)workspace;
    a.ticket = (unsigned*)((char*)workspace + YNET_BCE_PARTS * sizeof(double));
    a.loss = loss;
    a.cin = cin;
    a.cout = cout;
    a.cout_pad = ceil_div(cout, 64) * 64;
    a.B = B;
    a.hw4 = HW / 4;
    a.n = (long long)B * cout * HW;
    a.gs = expected_grad / (float)a.n;
    // exact channel counts for the two prediction horizons of the shipped configs (12 and 30 steps): no padded FMAs
    const int parts4 = grid_for((long long)B * a.hw4, 256, YNET_BCE_PARTS), parts2 = grid_for((long long)B * a.hw4 * 2, 256, YNET_BCE_PARTS);
    if (t_xy != nullptr) {
        if (cout == 12) hipLaunchKernelGGL((pred_bce_kernel<12, 4, true>), dim3(parts4), dim3(256), 0, (hipStream_t)stream, a);
        else if (cout <= 16) hipLaunchKernelGGL((pred_bce_kernel<16, 4, true>), dim3(parts4), dim3(256), 0, (hipStream_t)stream, a);
        else if (cout == 30) hipLaunchKernelGGL((pred_bce_kernel<30, 2, true>), dim3(parts2), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((pred_bce_kernel<32, 2, true>), dim3(parts2), dim3(256), 0, (hipStream_t)stream, a);
        return ynet_check_launch("pred_bce_blob");
    }
    if (cout == 12) hipLaunchKernelGGL((pred_bce_kernel<12, 4>), dim3(parts4), dim3(256), 0, (hipStream_t)stream, a);      // (2 pixels per thread: measured equal)
    else if (cout <= 16) hipLaunchKernelGGL((pred_bce_kernel<16, 4>), dim3(parts4), dim3(256), 0, (hipStream_t)stream, a);
    else if (cout == 30) hipLaunchKernelGGL((pred_bce_kernel<30, 2>), dim3(parts2), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((pred_bce_kernel<32, 2>), dim3(parts2), dim3(256), 0, (hipStream_t)stream, a);
    return ynet_check_launch("pred_bce");
}

int ynet_pred_bce(const float* x, long long x_batch_stride, const float* wp, const float* bias, const float* target,
                  float* y, float* loss, float* dx, float* dy, void* workspace, int B, int cin, int cout, long long HW,
                  float expected_grad, int dx_relu_mask, void* stream) {
    YNET_REQUIRE(target != nullptr, "pred_bce: null pointer");
    return pred_bce_launch(x, x_batch_stride, wp, bias, target, nullptr, nullptr, 0, 0, 0, 0, y, loss, dx, dy, workspace, B, cin, cout, HW, expected_grad,
                           dx_relu_mask, stream);
}

int ynet_pred_bce_blob(const float* x, long long x_batch_stride, const float* wp, const float* bias, const float* target_xy, const float* blob, int kernlen,
                       int S, int H, int W, float* y, float* loss, float* dx, float* dy, void* workspace, int B, int cin, int cout, float expected_grad,
                       int dx_relu_mask, void* stream) {
    YNET_REQUIRE(target_xy != nullptr && blob != nullptr, "pred_bce_blob: null pointer");
    return pred_bce_launch(x, x_batch_stride, wp, bias, nullptr, target_xy, blob, kernlen, S, H, W, y, loss, dx, dy, workspace, B, cin, cout,
                           (long long)H * W, expected_grad, dx_relu_mask, stream);
}

int ynet_bce_logits_fwd(const float* x, const float* t, long long n, float* loss, void* workspace, void* stream) {
    YNET_REQUIRE(x && t && loss && workspace && n > 0, "bce_logits_fwd: bad arguments");
    YNET_REQUIRE((((uintptr_t)x | (uintptr_t)t) & 15) == 0, "bce_logits_fwd: inputs must be 16-byte aligned");
    const int parts = grid_for(n / 4 + 1, 256, YNET_BCE_PARTS);
    hipLaunchKernelGGL(bce_fwd_kernel<false>, dim3(parts), dim3(256), 0, (hipStream_t)stream, x, t, n, (double*)workspace,
                       (float*)nullptr, 0.f);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, parts, n, loss);
    return ynet_check_launch("bce_logits_fwd");
}

int ynet_bce_logits_fwd_grad(const float* x, const float* t, long long n, float expected_grad, float* loss, float* dx,
                             void* workspace, void* stream) {
    YNET_REQUIRE(x && t && loss && dx && workspace && n > 0, "bce_logits_fwd_grad: bad arguments");
    YNET_REQUIRE((((uintptr_t)x | (uintptr_t)t | (uintptr_t)dx) & 15) == 0, "bce_logits_fwd_grad: buffers must be 16-byte aligned");
    YNET_REQUIRE(expected_grad != 0.f && expected_grad == expected_grad, "bce_logits_fwd_grad: the expected gradient must be non-zero");
    const int parts = grid_for(n / 4 + 1, 256, YNET_BCE_PARTS);
    hipLaunchKernelGGL(bce_fwd_kernel<true>, dim3(parts), dim3(256), 0, (hipStream_t)stream, x, t, n, (double*)workspace,
                       dx, expected_grad / (float)n);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, parts, n, loss);
    return ynet_check_launch("bce_logits_fwd_grad");
}

int ynet_bce_grad_rescale(float* dx, const float* grad_out, float expected_grad, long long n, void* stream) {
    YNET_REQUIRE(dx && grad_out && n > 0, "bce_grad_rescale: bad arguments");
    YNET_REQUIRE((((uintptr_t)dx) & 15) == 0, "bce_grad_rescale: dx must be 16-byte aligned");
    YNET_REQUIRE(expected_grad != 0.f && expected_grad == expected_grad, "bce_grad_rescale: the expected gradient must be non-zero");
    hipLaunchKernelGGL(bce_rescale_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, dx,
                       grad_out, expected_grad, n);
    return ynet_check_launch("bce_grad_rescale");
}

int ynet_bce_logits_bwd(const float* x, const float* t, const float* grad_out, float* dx, long long n, void* stream) {
    YNET_REQUIRE(x && t && grad_out && dx && n > 0, "bce_logits_bwd: bad arguments");
    YNET_REQUIRE((((uintptr_t)x | (uintptr_t)t | (uintptr_t)dx) & 15) == 0, "bce_logits_bwd: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, t, grad_out, dx, n);
    return ynet_check_launch("bce_logits_bwd");
}

int ynet_softargmax2d(const float* x, float* out, long long B, int C, long long batch_stride, int H, int W,
                      void* stream) {
    YNET_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "softargmax2d: bad arguments");
    const long long planes = B * C;
    YNET_REQUIRE(planes < (1ll << 31), "softargmax2d: too many planes");
    YNET_REQUIRE((W & 3) != 0 || ((((uintptr_t)x) & 15) == 0 && (batch_stride & 3) == 0 && (((long long)H * W) & 3) == 0),
                 "softargmax2d: input must be 16-byte aligned");
    hipLaunchKernelGGL(softargmax_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, out, C,
                       batch_stride, H, W, 1e-6f);
    return ynet_check_launch("softargmax2d");
}

int ynet_train_readout(const float* traj_map, long long traj_bs, const float* goal_map, long long goal_bs, int goal_channel,
                       const float* gt_future, float* pred_traj, float* pred_goal, float* ade, float* fde,
                       int B, int P, int H, int W, float resize_factor, void* stream) {
    YNET_REQUIRE(traj_map && goal_map && gt_future && pred_traj && pred_goal && ade && fde, "train_readout: null pointer");
    YNET_REQUIRE(B > 0 && P > 0 && H > 0 && W > 0 && goal_channel >= 0 && resize_factor != 0.f, "train_readout: bad arguments");
    YNET_REQUIRE((W & 3) != 0 || ((((uintptr_t)traj_map | (uintptr_t)goal_map) & 15) == 0 && ((traj_bs | goal_bs) & 3) == 0 && (((long long)H * W) & 3) == 0),
                 "train_readout: maps must be 16-byte aligned");
    const long long n0 = (long long)B * P, planes = n0 + B;
    YNET_REQUIRE(planes < (1ll << 31), "train_readout: too many planes");
    hipLaunchKernelGGL(softargmax_two_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, traj_map, traj_bs, P,
                       goal_map + (long long)goal_channel * H * W, goal_bs, n0, pred_traj, pred_goal, H, W, 1e-6f);
    hipLaunchKernelGGL(train_readout_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, pred_traj, pred_goal, gt_future,
                       B, P, resize_factor, ade, fde);
    return ynet_check_launch("train_readout");
}

int ynet_pad2d(const float* x, float* y, long long N, int H, int W, int Hp, int Wp, void* stream) {
    YNET_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W, "pad2d: bad arguments");
    hipLaunchKernelGGL(pad2d_kernel, dim3(grid_for(N * Hp * Wp, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, Hp, Wp);
    return ynet_check_launch("pad2d");
}

int ynet_seg_onehot_pad(const int* labels, float* y, int H, int W, int Hp, int Wp, int classes, void* stream) {
    YNET_REQUIRE(labels && y && H > 0 && W > 0 && Hp >= H && Wp >= W && classes > 0, "seg_onehot_pad: bad arguments");
    hipLaunchKernelGGL(seg_onehot_pad_kernel, dim3(grid_for((long long)classes * Hp * Wp, 256)), dim3(256), 0, (hipStream_t)stream,
                       labels, y, H, W, Hp, Wp, classes);
    return ynet_check_launch("seg_onehot_pad");
}

int ynet_resize_nearest(const int* labels, int* out, int H, int W, int Ho, int Wo, double fx, double fy, void* stream) {
    YNET_REQUIRE(labels && out && H > 0 && W > 0 && Ho > 0 && Wo > 0 && fx > 0.0 && fy > 0.0, "resize_nearest: bad arguments");
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for((long long)Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream,
                       labels, out, H, W, Ho, Wo, 1.0 / fx, 1.0 / fy);
    return ynet_check_launch("resize_nearest");
}

long long ynet_batchnorm_workspace_doubles(int C) { return C > 0 ? 2ll * C * BN_PARTS : 0; }

static inline dim3 bn_plane_grid(long long planes, long long HW) {
    long long gx = (HW + 1023) / 1024;
    return dim3((unsigned)(gx < 1 ? 1 : (gx > 64 ? 64 : gx)), (unsigned)planes);
}

int ynet_batchnorm2d_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                         double* workspace, int B, int C, long long HW, int train, double momentum, double eps, void* stream) {
    YNET_REQUIRE(x && y && save_mean && save_invstd && B > 0 && C > 0 && HW > 0 && (long long)B * C < 65536, "batchnorm2d_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (train) {
        YNET_REQUIRE(workspace != nullptr, "batchnorm2d_fwd: training mode needs ynet_batchnorm_workspace_doubles(C) doubles of workspace");
        hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(C * BN_PARTS), dim3(256), 0, st, x, nullptr, nullptr, nullptr, workspace, B, C, HW);
        hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, st, workspace, save_mean, save_invstd, running_mean, running_var, C, (double)B * (double)HW, eps, momentum);
        hipLaunchKernelGGL(bn_apply_kernel, bn_plane_grid((long long)B * C, HW), dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta, y, (long long)B * C, C, HW);
    } else {
        YNET_REQUIRE(running_mean && running_var, "batchnorm2d_fwd: evaluation mode needs the running statistics");
        hipLaunchKernelGGL(bn_invstd_kernel, dim3((C + 63) / 64), dim3(64), 0, st, running_var, save_invstd, C, eps);
        hipLaunchKernelGGL(bn_apply_kernel, bn_plane_grid((long long)B * C, HW), dim3(256), 0, st, x, running_mean, save_invstd, gamma, beta, y, (long long)B * C, C, HW);
    }
    return ynet_check_launch("batchnorm2d_fwd");
}

int ynet_batchnorm2d_bwd(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, float* dgamma, float* dbeta, double* workspace,
                         int B, int C, long long HW, int train, void* stream) {
    YNET_REQUIRE(dy && x && mean && invstd && dx && workspace && B > 0 && C > 0 && HW > 0 && (long long)B * C < 65536, "batchnorm2d_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(C * BN_PARTS), dim3(256), 0, st, x, dy, mean, invstd, workspace, B, C, HW);
    hipLaunchKernelGGL(bn_bwd_kernel, bn_plane_grid((long long)B * C, HW), dim3(256), 0, st, dy, x, mean, invstd, gamma, workspace, dx, dgamma, dbeta, (long long)B * C, C, HW,
                       (double)B * (double)HW, train ? 1 : 0);
    return ynet_check_launch("batchnorm2d_bwd");
}

int ynet_add_relu(const float* a, const float* b, float* y, long long n, int relu, void* stream) {
    YNET_REQUIRE(a && b && y && n > 0, "add_relu: bad arguments");
    hipLaunchKernelGGL(add_relu_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, relu ? 1 : 0);
    return ynet_check_launch("add_relu");
}

int ynet_relu_bwd(const float* dy, const float* y, float* dx, long long n, void* stream) {
    YNET_REQUIRE(dy && y && dx && n > 0, "relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    return ynet_check_launch("relu_bwd");
}

int ynet_upconv_dgrad_ring(const float* D, long long d_bs, const float* tables, const float* relu_of, long long relu_of_bs, float* dx, long long dx_bs,
                           int B, int C4, int cin, int h, int w, void* stream) {
    YNET_REQUIRE(D && tables && dx && B > 0 && C4 > 0 && (C4 & 3) == 0 && cin > 0 && h >= 2 && w >= 2, "upconv_dgrad_ring: bad arguments (C4 = 4 cout)");
    YNET_REQUIRE(d_bs >= (long long)C4 * h * w && dx_bs >= (long long)cin * h * w && (relu_of == nullptr || relu_of_bs >= (long long)cin * h * w),
                 "upconv_dgrad_ring: batch strides smaller than the images");
    const int lds = (3 * C4 * cin + C4 * (RING_SEG + 2) + 8 * 32) * 4;
    YNET_REQUIRE(lds <= 64 * 1024, "upconv_dgrad_ring: 4 cout %d x cin %d tables do not fit the 64 KB of LDS this kernel uses", C4, cin);
    const int nseg_row = (w + RING_SEG - 1) / RING_SEG, nseg_col = h > 2 ? (h - 2 + RING_SEG - 1) / RING_SEG : 0;
    const long long blocks = (long long)B * (2 * nseg_row + 2 * nseg_col);
    YNET_REQUIRE(blocks < (1ll << 31), "upconv_dgrad_ring: too many workgroups");
    hipLaunchKernelGGL(upconv_ring_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, D, d_bs, tables, relu_of, relu_of_bs, dx, dx_bs, B, C4, cin, h, w,
                       nseg_row, nseg_col);
    return ynet_check_launch("upconv_dgrad_ring");
}

int ynet_rot90_flip(const void* src, void* dst, long long N, int H, int W, int k, int flip, void* stream) {
    YNET_REQUIRE(src && dst && src != dst && N > 0 && H > 0 && W > 0 && k >= 0, "rot90_flip: bad arguments");
    YNET_REQUIRE(N * (long long)H * W < (1ll << 40), "rot90_flip: too many elements");
    hipLaunchKernelGGL(rot90_flip_kernel, dim3(grid_for(N * (long long)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned*)src, (unsigned*)dst, N, H, W, k & 3, flip ? 1 : 0);
    return ynet_check_launch("rot90_flip");
}

int ynet_rot_coords(double* xy, long long n, double cx, double cy, double r00, double r01, double r10, double r11, double ox, double oy, void* stream) {
    YNET_REQUIRE(xy && n > 0, "rot_coords: bad arguments");
    hipLaunchKernelGGL(rot_coords_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, xy, n, cx, cy, r00, r01, r10, r11, ox, oy);
    return ynet_check_launch("rot_coords");
}

static void pred_softargmax_plan(int H, int W, int* gpw, int* nchunk) {
    const int ngroups = (int)(((long long)H * W) >> 7);
    int g = ngroups / 32;
    if (g < 1) g = 1;
    *gpw = g;
    *nchunk = (ngroups + g - 1) / g;
}

int ynet_pred_softargmax_supported(int cin, int cout, int H, int W) {
    return (cin == 8 || cin == 16 || cin == 32) && cout >= 1 && cout <= 32 && H > 0 && W > 0 && (W & 3) == 0 &&
           (((long long)H * W) & 127) == 0 && (long long)H * W < (1ll << 30);
}

long long ynet_pred_softargmax_workspace_floats(long long B, int H, int W) {
    int gpw = 1, nchunk = 1;
    pred_softargmax_plan(H, W, &gpw, &nchunk);
    return B * nchunk * 128;
}

int ynet_pred_softargmax(const float* x, long long x_bs, const float* w, const float* bias, float* out, float* workspace,
                         long long B, int cin, int cout, int H, int W, void* stream) {
    YNET_REQUIRE(x && w && out && workspace && B > 0, "pred_softargmax: bad arguments");
    YNET_REQUIRE(ynet_pred_softargmax_supported(cin, cout, H, W), "pred_softargmax: cin %d cout %d %dx%d is not served (cin 8/16/32, cout <= 32, H*W %% 128 == 0, W %% 4 == 0)", cin, cout, H, W);
    YNET_REQUIRE((((uintptr_t)x | (uintptr_t)workspace) & 15) == 0 && (x_bs & 3) == 0, "pred_softargmax: x and the workspace must be 16-byte aligned");
    PredSoftArgs a{};
    a.x = x;
    a.x_bs = x_bs;
    a.w = w;
    a.bias = bias;
    a.partial = workspace;
    a.cout = cout;
    a.H = H;
    a.W = W;
    pred_softargmax_plan(H, W, &a.gpw, &a.nchunk);
    const long long blocks = B * ((a.nchunk + 3) / 4);
    YNET_REQUIRE(blocks < (1ll << 31), "pred_softargmax: too many workgroups");
    hipStream_t st = (hipStream_t)stream;
    static const int pt = getenv("YNET_PRED_SOFT_PT") ? atoi(getenv("YNET_PRED_SOFT_PT")) : 4;      // (2: the half-width variant, measured 8 % slower)
    if (cin == 32 && pt == 4) hipLaunchKernelGGL((pred_softargmax_kernel<32, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (cin == 32) hipLaunchKernelGGL((pred_softargmax_kernel<32, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (cin == 16) hipLaunchKernelGGL((pred_softargmax_kernel<16, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pred_softargmax_kernel<8, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    int rc = ynet_check_launch("pred_softargmax");
    if (rc) return rc;
    hipLaunchKernelGGL(pred_softargmax_combine_kernel, dim3((unsigned)((B * cout + 255) / 256)), dim3(256), 0, st, workspace, out, B, cout,
                       a.nchunk, 1e-6f);
    return ynet_check_launch("pred_softargmax(combine)");
}

int ynet_sigmoid_temp(const float* x, float* y, long long B, int C, long long HW, const int* sel, int nsel,
                      float temperature, void* stream) {
    YNET_REQUIRE(x && y && sel && B > 0 && C > 0 && HW > 0, "sigmoid_temp: bad arguments");
    YNET_REQUIRE(nsel >= 1 && nsel <= 8, "sigmoid_temp: 1..8 selected channels supported (got %d)", nsel);
    YNET_REQUIRE(temperature != 0.f, "sigmoid_temp: temperature must be non-zero");
    SigArgs a{};
    a.x = x;
    a.y = y;
    for (int i = 0; i < nsel; ++i) {
        YNET_REQUIRE(sel[i] >= 0 && sel[i] < C, "sigmoid_temp: channel %d out of range [0,%d)", sel[i], C);
        a.sel[i] = sel[i];
    }
    a.nsel = nsel;
    a.C = C;
    a.B = B;
    a.HW = HW;
    a.T = temperature;
    hipLaunchKernelGGL(sigmoid_temp_kernel, dim3(grid_for(B * nsel * HW, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return ynet_check_launch("sigmoid_temp");
}

int ynet_gather_patch(const float* tmpl, int SH, int SW, const float* xy, float* out, int N, int H, int W,
                      int* status, void* stream) {
    YNET_REQUIRE(tmpl && xy && out && status, "gather_patch: null pointer");
    YNET_REQUIRE(N > 0 && H > 0 && W > 0 && SH >= H && SW >= W, "gather_patch: bad shape N=%d %dx%d in %dx%d", N, H, W, SH, SW);
    int gx = (H * W + 255) / 256;
    if (gx > 64) gx = 64;
    for (int n0 = 0; n0 < N; n0 += 65535) {      // grid.y is limited to 65535 windows per launch
        const int n = N - n0 < 65535 ? N - n0 : 65535;
        hipLaunchKernelGGL(gather_patch_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, tmpl, SH, SW, xy + 2ll * n0,
                           out + (long long)n0 * H * W, H, W, status);
    }
    return ynet_check_launch("gather_patch");
}

int ynet_heatmap_analytic(const float* xy, float* out, int N, int H, int W, int S, int kind, double dmax,
                          const float* blob, int kernlen, int* status, void* stream) {
    YNET_REQUIRE(xy && out && status, "heatmap_analytic: null pointer");
    YNET_REQUIRE(N > 0 && H > 0 && W > 0 && S >= H && S >= W, "heatmap_analytic: bad shape N=%d %dx%d in %d", N, H, W, S);
    YNET_REQUIRE(kind == 0 ? dmax > 0.0 : (kind == 1 && blob != nullptr && kernlen > 0 && kernlen <= S),
                 "heatmap_analytic: kind 0 needs dmax > 0, kind 1 a blob table");
    int gx = (H * W + 255) / 256;
    if (gx > 64) gx = 64;
    for (int n0 = 0; n0 < N; n0 += 65535) {
        const int n = N - n0 < 65535 ? N - n0 : 65535;
        hipLaunchKernelGGL(heatmap_analytic_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, xy + 2ll * n0,
                           out + (long long)n0 * H * W, H, W, S, kind, dmax, blob, kernlen, status);
    }
    return ynet_check_launch("heatmap_analytic");
}

int ynet_kmeans2d(const float* points, const int* init_idx, float* centers, int* status, int P, int N, int K, float tol,
                  int iter_limit, void* stream) {
    YNET_REQUIRE(points && init_idx && centers && status, "kmeans2d: null pointer");
    YNET_REQUIRE(P > 0 && N > 0 && N <= 18000 && K >= 1 && K <= 32 && K <= N, "kmeans2d: bad shape P=%d N=%d K=%d (N <= 18000, K <= 32)", P, N, K);
    const int lds = (2 * N + 64 + 96 + 2) * 4;
    static bool attr_dev[YNET_MAX_DEV] = {false};
    bool& attr_set = attr_dev[ynet_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans2d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kmeans2d_kernel, dim3(P), dim3(1024), lds, (hipStream_t)stream, points, init_idx, centers, status, N, K, tol, iter_limit);
    return ynet_check_launch("kmeans2d");
}

}  // extern "C"
