// Filter / bias gradient of the 3x3 (1x1, 5x5) "same" convolution on the fp32 matrix cores.
//
// Replaces ATen convolution_backward's weight/bias outputs (K4 in SURVEY.md section 2.1) for the
// trainable convs of models/ynet.py (all of them in train_net=train/all, the adapted encoder convs
// in mosa_*, where dW then feeds ynet_lora_grad).
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} dy[b,co,y,x] * [yact[b,co,y,x] > 0] * x[b,ci,y+ky-P,x+kx-P]
//   db[co]             = sum_{b,y,x} dy[b,co,y,x] * [yact > 0]
//
// GEMM view per tap: M = 32 output channels (A = dy tile, LDS [co][pixel], odd stride),
// N = 32 input channels (B = x tile with halo, LDS [ci][row][col], odd channel stride),
// K = pixels, two per v_mfma_f32_32x32x2_f32.  db comes from one more MFMA per K-step against a
// B operand of ones.  The pixel dimension is split over the 4 waves of a workgroup and over
// `nsplit` workgroups; waves are summed through LDS (in wave order), workgroups through a partial buffer reduced by
// a second kernel in a fixed order, so the result is bitwise reproducible (no float atomics).
#include "ynet_common.h"

struct WgradArgs {
    YSrc src[YNET_MAX_SRC];   // x = virtual concat of the sources
    int nsrc, cin;
    const float* dy;
    long long dy_bs;
    const float* mask;        // post-ReLU activation of this conv (NULL: no ReLU)
    long long mask_bs;
    float* partial_w;         // [nsplit][cout*cin*KK]
    float* partial_b;         // [nsplit][cout] or NULL
    int B, H, W, cout;
    int tiles_x, tiles_y, ntiles, nsplit, co_blks, ci_blks, tap_groups;
};

template <int KS, int NT>
struct WgCfg {
    static constexpr int PAD = KS / 2, KK = KS * KS;
    static constexpr int TH = 8, TW = 32, NPIX = TH * TW;
    static constexpr int TROWS = TH + KS - 1, TCOLS = TW + KS - 1;
    static constexpr int XCH = (TROWS * TCOLS) | 1;          // odd channel stride -> conflict-free B reads
    static constexpr int DCH = NPIX + 1;                      // odd row stride     -> conflict-free A reads
    static constexpr int XS_FLOATS = 32 * XCH, DS_FLOATS = 32 * DCH;
    static constexpr int NACC = NT + 1;                       // + bias column
    static constexpr int RED_FLOATS = NACC * 16 * 64;         // one wave's accumulators
    static constexpr int LDS_FLOATS = (XS_FLOATS + DS_FLOATS) > RED_FLOATS ? (XS_FLOATS + DS_FLOATS) : RED_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

template <int KS, int NT>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradArgs a) {
    using C = WgCfg<KS, NT>;
    constexpr int PAD = C::PAD, KK = C::KK, TH = C::TH, TW = C::TW;
    constexpr int TROWS = C::TROWS, TCOLS = C::TCOLS, XCH = C::XCH, DCH = C::DCH, NACC = C::NACC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                   // [32 ci][XCH]
    float* ds = smem + C::XS_FLOATS;    // [32 co][DCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int bid = blockIdx.x;
    const int tg = bid % a.tap_groups;  // tap group (one filter row for 5x5; everything otherwise)
    bid /= a.tap_groups;
    const int cib = bid % a.ci_blks;
    bid /= a.ci_blks;
    const int cob = bid % a.co_blks;
    const int split = bid / a.co_blks;
    const int HW = a.H * a.W;
    const int ci0 = cib * 32, co0 = cob * 32;
    const int tap0 = tg * NT;
    const bool want_bias = (a.partial_b != nullptr) && cib == 0 && tg == 0;

    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    for (int tile = split; tile < a.ntiles; tile += a.nsplit) {
        int t = tile;
        const int txi = t % a.tiles_x;
        t /= a.tiles_x;
        const int tyi = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int x0 = txi * TW, y0 = tyi * TH;
        __syncthreads();
        // ---- stage x tile with halo: 32 input channels of this block
#pragma unroll 1
        for (int c = 0; c < 32; ++c) {
            const float* base = nullptr;
            const int cc = ci0 + c;
            if (cc < a.cin) {
                int s = 0, rel = cc;
                while (s < a.nsrc - 1 && rel >= a.src[s].c) {
                    rel -= a.src[s].c;
                    ++s;
                }
                base = a.src[s].p + (long long)b * a.src[s].bs + (long long)rel * HW;
            }
            for (int i = tid; i < TROWS * TCOLS; i += 256) {
                const int ty = i / TCOLS, tx = i - ty * TCOLS;
                const int gy = y0 + ty - PAD, gx = x0 + tx - PAD;
                float v = 0.f;
                if (base != nullptr && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = base[gy * a.W + gx];
                xs[c * XCH + i] = v;
            }
        }
        // ---- stage (masked) dy tile: 32 output channels
#pragma unroll 1
        for (int c = 0; c < 32; ++c) {
            const int co = co0 + c;
            const float* base = co < a.cout ? a.dy + (long long)b * a.dy_bs + (long long)co * HW : nullptr;
            const float* mbase = (co < a.cout && a.mask) ? a.mask + (long long)b * a.mask_bs + (long long)co * HW : nullptr;
            {
                const int i = tid;  // NPIX == 256 == blockDim
                const int ty = i / TW, tx = i - ty * TW;
                const int gy = y0 + ty, gx = x0 + tx;
                float v = 0.f;
                if (base != nullptr && gy < a.H && gx < a.W) {
                    v = base[gy * a.W + gx];
                    if (mbase != nullptr) v = mbase[gy * a.W + gx] > 0.f ? v : 0.f;
                }
                ds[c * DCH + i] = v;
            }
        }
        __syncthreads();
        // ---- MFMA: this wave owns rows 2*wave, 2*wave+1 of the tile
        const float* ap = ds + l31 * DCH + half;
        const float* bp = xs + l31 * XCH + half;
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int row = wave * 2 + rr;
#pragma unroll 4
            for (int xx = 0; xx < TW; xx += 2) {
                const float av = ap[row * TW + xx];
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    int ky, kx;
                    if (NT == KK) {
                        ky = t2 / KS;
                        kx = t2 % KS;
                    } else {
                        ky = tg;   // one filter row per tap group
                        kx = t2;
                    }
                    const float bv = bp[(row + ky) * TCOLS + xx + kx];
                    acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t2], 0, 0, 0);
                }
                if (want_bias) acc[NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, 1.0f, acc[NT], 0, 0, 0);
            }
        }
    }

    // ---- sum the 4 waves through LDS into wave 0 (waves 1,2,3 in turn: fixed order)
    float* red = smem;
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[(i * 16 + q) * 64 + lane] = acc[i][q];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][q] += red[(i * 16 + q) * 64 + lane];
        }
    }
    if (wave == 0) {
        const int ci = ci0 + l31;
        float* pw = a.partial_w + (long long)split * a.cout * a.cin * KK;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * half;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float v = acc[i][q];
                if (co < a.cout && ci < a.cin) pw[((long long)co * a.cin + ci) * KK + tap0 + i] = v;
            }
            if (want_bias) {
                const float v = acc[NT][q];
                if (co < a.cout && l31 == 0) a.partial_b[(long long)split * a.cout + co] = v;
            }
        }
    }
}

// out[i] = sum_s partial[s][i], fixed order
__global__ void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, long long n,
                                       int nsplit, int accumulate) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int s = 0;
        for (; s + 4 <= nsplit; s += 4) {
            s0 += partial[(long long)s * n + i];
            s1 += partial[(long long)(s + 1) * n + i];
            s2 += partial[(long long)(s + 2) * n + i];
            s3 += partial[(long long)(s + 3) * n + i];
        }
        for (; s < nsplit; ++s) s0 += partial[(long long)s * n + i];
        const float v = (s0 + s1) + (s2 + s3);
        out[i] = accumulate ? out[i] + v : v;
    }
}

template <int KS, int NT>
static int launch_wgrad(WgradArgs& a, float* dw, float* db, hipStream_t st) {
    using C = WgCfg<KS, NT>;
    a.tap_groups = C::KK / NT;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<KS, NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    const long long nblk = (long long)a.nsplit * a.co_blks * a.ci_blks * a.tap_groups;
    hipLaunchKernelGGL((wgrad_mfma_kernel<KS, NT>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, st, a);
    int rc = ynet_check_launch("conv2d_wgrad");
    if (rc) return rc;
    const long long nw = (long long)a.cout * a.cin * C::KK;
    int grid = (int)((nw + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid), dim3(256), 0, st, a.partial_w, dw, nw, a.nsplit, 0);
    if (db) hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, a.partial_b, db, (long long)a.cout, a.nsplit, 0);
    return ynet_check_launch("conv2d_wgrad(reduce)");
}

static int wgrad_plan(int B, int H, int W, int cout, int cin, int K, int* nsplit_out) {
    const int tiles = B * ceil_div(H, 8) * ceil_div(W, 32);
    const int blocks_per_split = ceil_div(cout, 32) * ceil_div(cin, 32) * (K == 5 ? 5 : 1);
    int nsplit = 768 / blocks_per_split;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > tiles) nsplit = tiles;
    if (nsplit > 256) nsplit = 256;
    *nsplit_out = nsplit;
    return tiles;
}

extern "C" {

// floats of workspace ynet_conv2d_wgrad needs for this problem
long long ynet_conv2d_wgrad_workspace_floats(int B, int H, int W, int cout, int cin, int K) {
    int nsplit;
    wgrad_plan(B, H, W, cout, cin, K, &nsplit);
    return (long long)nsplit * ((long long)cout * cin * K * K + cout);
}

int ynet_conv2d_wgrad(const float* const* src, const int* src_c, const long long* src_bs, int nsrc,
                      const float* dy, long long dy_bs, const float* mask, long long mask_bs,
                      float* dw, float* db, float* workspace, int B, int H, int W, int cout, int K,
                      void* stream) {
    YNET_REQUIRE(nsrc >= 1 && nsrc <= YNET_MAX_SRC, "conv2d_wgrad: 1..%d sources supported", YNET_MAX_SRC);
    YNET_REQUIRE(dy && dw && workspace, "conv2d_wgrad: null pointer");
    YNET_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0, "conv2d_wgrad: empty problem");
    WgradArgs a{};
    a.nsrc = nsrc;
    a.cin = 0;
    for (int i = 0; i < nsrc; ++i) {
        YNET_REQUIRE(src[i] != nullptr && src_c[i] > 0, "conv2d_wgrad: source %d is null/empty", i);
        a.src[i] = YSrc{src[i], src_c[i], src_bs[i]};
        a.cin += src_c[i];
    }
    a.dy = dy;
    a.dy_bs = dy_bs;
    a.mask = mask;
    a.mask_bs = mask_bs;
    a.B = B;
    a.H = H;
    a.W = W;
    a.cout = cout;
    a.tiles_x = ceil_div(W, 32);
    a.tiles_y = ceil_div(H, 8);
    a.ntiles = wgrad_plan(B, H, W, cout, a.cin, K, &a.nsplit);
    a.co_blks = ceil_div(cout, 32);
    a.ci_blks = ceil_div(a.cin, 32);
    a.partial_w = workspace;
    a.partial_b = db ? workspace + (long long)a.nsplit * cout * a.cin * K * K : nullptr;
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
        case 1: return launch_wgrad<1, 1>(a, dw, db, st);
        case 3: return launch_wgrad<3, 9>(a, dw, db, st);
        case 5: return launch_wgrad<5, 5>(a, dw, db, st);
        default: ynet_set_error("conv2d_wgrad: kernel size %d not supported (1, 3, 5)", K); return 1;
    }
}

}  // extern "C"
