// Filter / bias gradient of the 3x3 (1x1, 5x5) "same" convolution on the fp32 matrix cores.
//
// Replaces ATen convolution_backward's weight/bias outputs (K4 in SURVEY.md section 2.1) for the
// trainable convs of models/ynet.py (all of them in train_net=train/all, the adapted encoder convs
// in mosa_*, where dW then feeds ynet_lora_grad).
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} dy[b,co,y,x] * [yact[b,co,y,x] > 0] * x[b,ci,y+ky-P,x+kx-P]
//   db[co]             = sum_{b,y,x} dy[b,co,y,x] * [yact > 0]
//
// GEMM view (v_mfma_f32_32x32x2_f32): M = 32 output channels (A = masked dy tile, LDS [co][pixel], odd
// stride), N = 32 columns of the flattened (ci, kx) index (B = x tile with halo, LDS [ci][row][col], odd
// channel stride; flattening kx into N means Cin = 14 costs 2 column tiles instead of 3 padded taps),
// K = pixels, two per instruction; one accumulator tile per (column tile, ky).  db comes from one more
// MFMA per K-step against a B operand of ones.
// Pipeline: persistent workgroups walk 4x32-pixel tiles; the buffer_loads (hardware range check =
// zero padding) of tile t+1 are issued before the MFMA loop of tile t and written to LDS after it.
// Pixels are split over the 4 waves (summed through LDS in wave order) and over `nsplit` workgroups
// (partials reduced by a second kernel in fixed order): bitwise reproducible, no float atomics.
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
    int tiles_x, tiles_y, ntiles, nsplit, co_blks, ci_blks;
};

template <int KS>
struct WgCfg {
    static constexpr int PAD = KS / 2, KK = KS * KS;
    static constexpr int TH = 4, TW = 32, NPIX = TH * TW;
    static constexpr int TROWS = TH + KS - 1, TCOLS = TW + KS - 1, XPLANE = TROWS * TCOLS;
    static constexpr int CIB = KS == 5 ? 6 : 32;                // input channels per workgroup
    static constexpr int NB = (CIB * KS + 31) / 32;             // 32-wide column tiles of the (ci,kx) index
    static constexpr int XCH = XPLANE | 1;                      // odd channel stride -> spread banks
    static constexpr int DCH = NPIX + 1;                        // odd row stride
    static constexpr int XS_FLOATS = CIB * XCH + 64, DS_FLOATS = 32 * DCH;
    static constexpr int NACC = NB * KS + 1;                    // + bias column
    static constexpr int RED_FLOATS = NACC * 16 * 64;           // one wave's accumulators
    static constexpr int LDS_FLOATS = (XS_FLOATS + DS_FLOATS) > RED_FLOATS ? (XS_FLOATS + DS_FLOATS) : RED_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(XPLANE <= 320, "x plane must fit the two-pass staging");
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const float* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(p ? bytes : 0u);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
__device__ __forceinline__ float wg_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

struct WgTile {
    int b, x0, y0;
};

template <int KS, bool MASK>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradArgs a) {
    using C = WgCfg<KS>;
    constexpr int PAD = C::PAD, KK = C::KK, TH = C::TH, TW = C::TW, NPIX = C::NPIX;
    constexpr int TCOLS = C::TCOLS, XPLANE = C::XPLANE, XCH = C::XCH, DCH = C::DCH;
    constexpr int CIB = C::CIB, NB = C::NB, NACC = C::NACC;
    constexpr int XI = (XPLANE + 255) / 256;     // x elements per thread per channel (1, or 2 for 5x5)
    constexpr int DI = 32 * NPIX / 256;          // dy elements per thread per tile (16)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                   // [CIB][XCH] (+64 floats of slack read by unused lanes)
    float* ds = smem + C::XS_FLOATS;    // [32 co][DCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int bid = blockIdx.x;
    const int cib = bid % a.ci_blks;
    bid /= a.ci_blks;
    const int cob = bid % a.co_blks;
    const int split = bid / a.co_blks;
    const int HW = __builtin_amdgcn_readfirstlane(a.H * a.W);
    const unsigned plane_bytes = (unsigned)HW * 4u;
    const int ci0 = cib * CIB, co0 = cob * 32;
    const int ncib = min(CIB, a.cin - ci0);                 // input channels of this block
    const int nco = min(32, a.cout - co0);
    const bool want_bias = (a.partial_b != nullptr) && cib == 0;
    const int e0 = a.src[0].c, e1 = e0 + (a.nsrc > 1 ? a.src[1].c : 0), e2 = e1 + (a.nsrc > 2 ? a.src[2].c : 0);

    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    // B-operand column of this lane in each column tile: n = nb*32 + l31 -> (ci, kx)
    int boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + l31;
        const int ci = n / KS, kx = n - ci * KS;
        boff[nb] = (ci < ncib ? ci * XCH + kx : CIB * XCH) + half;     // unused columns read the slack
    }

    float xr[CIB][XI], dr[DI], mr[MASK ? DI : 1];
    auto decode = [&](int t) {
        WgTile c;
        c.x0 = (t % a.tiles_x) * TW;
        t /= a.tiles_x;
        c.y0 = (t % a.tiles_y) * TH;
        c.b = t / a.tiles_y;
        return c;
    };
    auto load_tile = [&](const WgTile& t) {
        unsigned xoff[XI], doff;      // byte offsets inside an image plane; past-the-end = zero fill
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int i = tid + k * 256;
            const int ty = i / TCOLS, tx = i - ty * TCOLS;
            const int gy = t.y0 + ty - PAD, gx = t.x0 + tx - PAD;
            const bool ok = i < XPLANE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            xoff[k] = ok ? (unsigned)(gy * a.W + gx) * 4u : 0xFFFFFFF0u;
        }
        {
            const int p = tid & (NPIX - 1), py = p / TW, px = p - py * TW;
            const int dy_y = t.y0 + py, dy_x = t.x0 + px;
            doff = (dy_y < a.H && dy_x < a.W) ? (unsigned)(dy_y * a.W + dy_x) * 4u : 0xFFFFFFF0u;
        }
        const float* sb0 = a.src[0].p + (long long)t.b * a.src[0].bs;
        const float* sb1 = a.nsrc > 1 ? a.src[1].p + (long long)t.b * a.src[1].bs : nullptr;
        const float* sb2 = a.nsrc > 2 ? a.src[2].p + (long long)t.b * a.src[2].bs : nullptr;
        const float* sb3 = a.nsrc > 3 ? a.src[3].p + (long long)t.b * a.src[3].bs : nullptr;
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            const int cc = ci0 + c;
            const float* base = nullptr;
            if (c < ncib) {
                base = cc < e0 ? sb0 + (long long)cc * HW
                     : cc < e1 ? sb1 + (long long)(cc - e0) * HW
                     : cc < e2 ? sb2 + (long long)(cc - e1) * HW
                               : sb3 + (long long)(cc - e2) * HW;
            }
            const __amdgpu_buffer_rsrc_t r = wg_rsrc(base, plane_bytes);
#pragma unroll
            for (int k = 0; k < XI; ++k) xr[c][k] = wg_load(r, xoff[k]);
        }
        // dy / mask: 32 output channels x 128 pixels = 16 elements per thread; thread (k, tid) holds
        // channel 2k + (tid >> 7), pixel tid & 127
        const int chalf = __builtin_amdgcn_readfirstlane(tid >> 7);
#pragma unroll
        for (int k = 0; k < DI; ++k) {
            const int c = 2 * k + chalf;
            const float* base = c < nco ? a.dy + (long long)t.b * a.dy_bs + (long long)(co0 + c) * HW : nullptr;
            dr[k] = wg_load(wg_rsrc(base, plane_bytes), doff);
            if (MASK) {
                const float* mb = c < nco ? a.mask + (long long)t.b * a.mask_bs + (long long)(co0 + c) * HW : nullptr;
                mr[MASK ? k : 0] = wg_load(wg_rsrc(mb, plane_bytes), doff);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int c = 0; c < CIB; ++c)
#pragma unroll
            for (int k = 0; k < XI; ++k) {
                const int i = tid + k * 256;
                if (i < XPLANE) xs[c * XCH + i] = xr[c][k];
            }
        const int chalf = tid >> 7, p = tid & (NPIX - 1);
#pragma unroll
        for (int k = 0; k < DI; ++k)
            ds[(2 * k + chalf) * DCH + p] = (!MASK || mr[MASK ? k : 0] > 0.f) ? dr[k] : 0.f;
    };

    int tile = split;
    if (tid < 64) xs[CIB * XCH + tid] = 0.f;      // slack read by unused B columns
    if (tile < a.ntiles) load_tile(decode(tile));
    for (; tile < a.ntiles; tile += a.nsplit) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + a.nsplit < a.ntiles) load_tile(decode(tile + a.nsplit));
        // ---- MFMA: this wave owns row `wave` of the tile (32 pixels = 16 K-steps)
        const float* ap = ds + l31 * DCH + wave * TW + half;
        const float* bp = xs + wave * TCOLS;
#pragma unroll 2
        for (int xx = 0; xx < TW; xx += 2) {
            const float av = ap[xx];
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float bv = bp[boff[nb] + ky * TCOLS + xx];
                    acc[nb * KS + ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb * KS + ky], 0, 0, 0);
                }
            if (want_bias) acc[NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, 1.0f, acc[NACC - 1], 0, 0, 0);
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
        float* pw = a.partial_w + (long long)split * a.cout * a.cin * KK;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (co >= a.cout) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = nb * 32 + l31;
                const int ci = n / KS, kx = n - ci * KS;
                if (ci >= ncib) continue;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
                    pw[((long long)co * a.cin + ci0 + ci) * KK + ky * KS + kx] = acc[nb * KS + ky][q];
            }
            if (want_bias && l31 == 0) a.partial_b[(long long)split * a.cout + co] = acc[NACC - 1][q];
        }
    }
}

// out[i] = sum_s partial[s][i], fixed order
__global__ void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, long long n,
                                       int nsplit) {
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
        out[i] = (s0 + s1) + (s2 + s3);
    }
}

template <int KS, bool MASK>
static int launch_wgrad_m(WgradArgs& a, float* dw, float* db, hipStream_t st) {
    using C = WgCfg<KS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<KS, MASK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    const long long nblk = (long long)a.nsplit * a.co_blks * a.ci_blks;
    hipLaunchKernelGGL((wgrad_mfma_kernel<KS, MASK>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, st, a);
    int rc = ynet_check_launch("conv2d_wgrad");
    if (rc) return rc;
    const long long nw = (long long)a.cout * a.cin * C::KK;
    int grid = (int)((nw + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid), dim3(256), 0, st, a.partial_w, dw, nw, a.nsplit);
    if (db) hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, a.partial_b, db, (long long)a.cout, a.nsplit);
    return ynet_check_launch("conv2d_wgrad(reduce)");
}

template <int KS>
static int launch_wgrad(WgradArgs& a, float* dw, float* db, hipStream_t st) {
    a.ci_blks = ceil_div(a.cin, WgCfg<KS>::CIB);
    return a.mask ? launch_wgrad_m<KS, true>(a, dw, db, st) : launch_wgrad_m<KS, false>(a, dw, db, st);
}

static int wgrad_cib(int K) { return K == 5 ? 6 : 32; }

static int wgrad_plan(int B, int H, int W, int cout, int cin, int K, int* nsplit_out) {
    const int tiles = B * ceil_div(H, 4) * ceil_div(W, 32);
    const int blocks_per_split = ceil_div(cout, 32) * ceil_div(cin, wgrad_cib(K));
    int nsplit = 512 / blocks_per_split;       // ~2 resident workgroups per CU
    if (nsplit < 1) nsplit = 1;
    if (nsplit > tiles) nsplit = tiles;
    if (nsplit > 512) nsplit = 512;
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
        a.src[i] = YSrc{src[i], src_c[i], src_bs[i], 0};
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
    a.tiles_y = ceil_div(H, 4);
    a.ntiles = wgrad_plan(B, H, W, cout, a.cin, K, &a.nsplit);
    a.co_blks = ceil_div(cout, 32);
    a.partial_w = workspace;
    a.partial_b = db ? workspace + (long long)a.nsplit * cout * a.cin * K * K : nullptr;
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
        case 1: return launch_wgrad<1>(a, dw, db, st);
        case 3: return launch_wgrad<3>(a, dw, db, st);
        case 5: return launch_wgrad<5>(a, dw, db, st);
        default: ynet_set_error("conv2d_wgrad: kernel size %d not supported (1, 3, 5)", K); return 1;
    }
}

}  // extern "C"
