// Adapter (LoRA / MoSA) gradients of a 3x3 convolution WITHOUT the full filter gradient.
//
// Replaces, for the adapted encoder convs of train_net = mosa_r (models/ynet.py:141-144, loralib 0.1.1 `Conv2d`),
// the chain  ATen convolution_backward(weight)  ->  dA = s B^T dWm,  dB = s dWm A^T  (K4 + K6 in SURVEY.md 2.1):
// the reference computes dW [cout, cin, 3, 3] -- 2 * B*H*W * cin * cout * 9 FLOP -- only to project it onto the
// rank-r factors.  With  Wm = W.view(3 cout, 3 cin),  flat column index u = 9 ci + 3 ky + kx  of one output channel's
// filter, ii(u) = u / (3 cin), n(u) = u % (3 cin)  (so that dWm[3 co + ii][n] = dW[co][u]),  X_u[p] = x[ci][p + tap]:
//     DP[m = ii*RQ + q][p] = sum_co  B[3 co + ii][q] * dy'[co][p]            dy' = dy where the activation was > 0
//     XP[m = ii*RQ + q][p] = sum_{u: ii(u) = ii}  A[q][n(u)] * X_u[p]
//     dB[3 co + ii][q] = s * sum_p dy'[co][p] * XP[m][p]
//     dA[q][n]         = s * sum_ii sum_p DP[m][p] * X_{ii*3cin + n}[p]
// i.e. the pixel sums run over 9 r projected planes instead of cout (resp. 9 cin) channels: (54 cin + 18 cout) r MACs
// per pixel instead of 9 cin cout -- 8x fewer at 64 -> 64, 3x at 14 -> 32 for r = 1 -- and the two rank-r GEMMs and the
// dW round trip through HBM disappear.  Same sums in a different order; dA / dB agree with the reference's to fp32
// rounding (tests/test_gpu_kernels.py).  RQ = 3 r; this file serves r = 1 (BASELINE.json's headline config), cin,
// cout <= 64, aligned planes with W % 4 == 0; everything else stays on ynet_conv2d_wgrad + ynet_lora_grad.
//
// lora_wgrad_kernel<TH, NTG>: persistent workgroups (256 threads) walk TH x 32 pixel tiles.  Per tile
//   0. the x tile with halo goes global -> LDS by `buffer_load_dwordx4 ... lds` (16-byte quads, rows of 40 floats, the
//      hardware range check writes the zero padding);
//   1. vector ALU, one thread per pixel (x NH = 256 / pixels slices of the channel loop): dy' (read from HBM, masked,
//      written to LDS) and DP -- independent of the x tile, so they overlap its DMA -- then XP from the x tile;
//   2. v_mfma_f32_16x16x4_f32 with K = 4 consecutive pixels: E[m][co] += XP[m][p] * dy'[co][p] and, for every block of
//      16 filter columns u, G[m][u] += DP[m][p] * X_u[p]; the four waves split the column blocks, the accumulators stay
//      in registers over the workgroup's whole tile walk (rows m >= 9 of the 16 x 16 tiles are not used).
// lora_wgrad_reduce_kernel sums the workgroups' partials in a fixed order (bitwise reproducible, no float atomics) and
// picks G's row ii(u) for every column:  dA[q][n] = s * sum_wg sum_ii G[ii*RQ + q][ii*3cin + n],  dB = s * sum_wg E.
#include "ynet_common.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

#define LW_RQ 3              // r * K rows of lora_A / columns of lora_B (r = 1, K = 3)
#define LW_ROWS 9            // projected planes: (ii, q)

struct LoraWgArgs {
    YSrc src[YNET_MAX_SRC];   // x = virtual concat of the sources
    int nsrc, cin;
    const float* dy;
    long long dy_bs;
    const float* mask;        // post-ReLU activation of this conv (NULL: no ReLU, or dy arrives already masked)
    long long mask_bs;
    const float* lora_a;      // [RQ][3 cin]
    const float* lora_b;      // [3 cout][RQ]
    float* partial;           // [gridDim.x][9 * (9 cin + cout)]
    int B, H, W, cout;
    int tiles_x, tiles_y, ntiles;
#ifdef YNET_LW_PROFILE
    unsigned long long* prof;      // development build: per-phase cycle sums
#endif
};

#ifdef YNET_LW_PROFILE
#define LW_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); pr[i] += t_ - tp; tp = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LW_STAMP(i) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* lw_lds_ptr_t;
typedef const __attribute__((address_space(4))) float* lw_const_f32;      // uniform reads -> scalar loads

template <int TH>
struct LwCfg {
    static constexpr int TW = 32, NPIX = TH * TW, NH = 256 / NPIX;
    static constexpr int TROWS = TH + 2, TCOLS = 40;                      // x rows [x0 - 4, x0 + 36): aligned quads
    static constexpr int XDATA = TROWS * TCOLS / 4;                       // quads of one channel's tile
    static constexpr int XQ = (XDATA + 1 + 6) / 8 * 8 + 1;                // + pad quads: = 1 mod 8 ...
    static constexpr int XCH = XQ * 4;                                    // ... so that the channel stride is 4 mod 32 floats
    static constexpr int DCH = NPIX + 4;                                  // dy' channel stride (4 mod 32)
    static constexpr int PL = NPIX + 4;                                   // projected-plane stride
    static_assert(XCH % 32 == 4 && DCH % 32 == 4, "bank-spread strides");
    static_assert(NH == 2 || NH == 4, "256 threads = NH slices of the tile's pixels");
    static constexpr int KSTEPS = NPIX / 4;
};

static inline int lw_xq(int th) { return ((th + 2) * 10 + 1 + 6) / 8 * 8 + 1; }
static inline int lw_lds_floats(int th, int cin, int cout) {
    const int npix = th * 32;
    return cin * lw_xq(th) * 4 + 64 + cout * (npix + 4) + 2 * LW_ROWS * (npix + 4) + cout * 12 + cin * 28;
}

__device__ __forceinline__ void lw_dma16(__amdgpu_buffer_rsrc_t r, const float* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lw_lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

// NTG: 16-column blocks of G per wave (the waves take blocks w, w + 4, ...); COS: output channels per channel slice;
// CIN_T / COUT_T: the channel counts as compile-time constants (0: read from the arguments) -- the encoder's own shapes get
// instantiations in which every channel bound folds away (left as run-time values the uniform loop conditions alone
// occupy ~200 scalar registers, spilled to vector lanes)
template <int TH, int NTG, int COS, int CIN_T, int COUT_T>
__global__ __launch_bounds__(256, (TH == 2 && COS <= 8) ? 3 : 2) void lora_wgrad_kernel(const LoraWgArgs a) {
    using C = LwCfg<TH>;
    constexpr int NPIX = C::NPIX, NH = C::NH, TCOLS = C::TCOLS, XQ = C::XQ, XCH = C::XCH, DCH = C::DCH, PL = C::PL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cin = CIN_T ? CIN_T : a.cin, cout = COUT_T ? COUT_T : a.cout;
    float* xs = smem;                                  // [cin][XCH] (+ 64 floats of slack for the last, partial DMA wave)
    float* ds = xs + cin * XCH + 64;                   // [cout][DCH]   dy' (masked output gradient)
    float* dpl = ds + cout * DCH;                      // [9][PL]       DP planes
    float* xpl = dpl + LW_ROWS * PL;                   // [9][PL]       XP planes
    float* tb = xpl + LW_ROWS * PL;                    // [cout][12]    lora_B rows 3 co .. 3 co + 2 (m = 3 ii + q), 16-byte rows
    float* ta = tb + cout * 12;                        // [cin][28]     lora_A coefficients of input channel ci's nine taps, [q][tap]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W;
    const unsigned plane_bytes = (unsigned)HW * 4u;
    const int ncol = 9 * cin;                          // filter columns u per output channel
    const int n3 = 3 * cin;
    const lw_const_f32 la = (lw_const_f32)a.lora_a;

    // ---- phase-2 roles: G blocks wave, wave + 4, ...; E block `wave` (16 output channels) if it exists
    const int NE = (cout + 15) >> 4;
    const int r16 = lane & 15, kq = lane >> 4;
    int gbase[NTG];                                    // LDS offset of this lane's B operand: column u's (ci, tap) + its K index
#pragma unroll
    for (int j = 0; j < NTG; ++j) {
        int u = (wave + 4 * j) * 16 + r16;
        u = u < ncol ? u : ncol - 1;                   // (columns past the end repeat the last one; never written out)
        const int ci = u / 9, t = u - 9 * ci, ky = t / 3, kx = t - 3 * ky;
        gbase[j] = ci * XCH + ky * TCOLS + kx + 3 + kq;          // tile column 0 (gx = x0 - 1) sits at LDS column 3
    }
    int eco = wave * 16 + r16;
    eco = eco < cout ? eco : cout - 1;
    const int ebase = eco * DCH + kq;
    const int arow = (r16 < LW_ROWS ? r16 : LW_ROWS - 1) * PL + kq;      // A operand: projected plane m = r16 (rows >= 9: unused)
    const bool has_e = wave < NE;

    f32x4 accg[NTG], acce = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NTG; ++j) accg[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- phase-1 roles: pixel p of the tile, slice h of the channel / column loops
    const int p = tid & (NPIX - 1);
    const int h = __builtin_amdgcn_readfirstlane(tid / NPIX);      // (a multiple of 64 pixels per slice: wave-uniform -> scalar loop control)
    const int prow = p >> 5, pcol = p & 31;
    const int xoff_p = prow * TCOLS + pcol + 3;

    // ---- x-tile DMA plan (tile independent part): quad q = tid + 256 k of the LDS image is quad `within` of channel q / XQ;
    // per tile only the window position is added and checked against the image
    constexpr int XI_MAX = TH == 4 ? 9 : 11;           // quads per thread: cin * XQ <= 256 * XI_MAX is checked by the host
    const int nquads = cin * XQ;
    int xrc[XI_MAX];                                   // tile row << 8 | (first column + 4), or -1: nothing to fetch (pad quad / past the image)
    unsigned xcoff[XI_MAX];                            // byte offset of the channel's plane inside its source
    int xsid[XI_MAX];                                  // source of the channel
#pragma unroll
    for (int k = 0; k < XI_MAX; ++k) {
        const int q = tid + k * 256;
        const int ch = q / XQ, within = q - ch * XQ;
        const int row = within / (TCOLS / 4), colp = (within - row * (TCOLS / 4)) * 4;      // column + 4
        int c = ch, sid = -1;
        if (q < nquads && within < C::XDATA) {
#pragma unroll
            for (int s_ = 0; s_ < YNET_MAX_SRC; ++s_) {
                if (sid < 0 && s_ < a.nsrc) {
                    if (c < a.src[s_].c) sid = s_;
                    else c -= a.src[s_].c;
                }
            }
        }
        xrc[k] = sid >= 0 ? (row << 8 | colp) : -1;
        xcoff[k] = (unsigned)c * plane_bytes;
        xsid[k] = sid;
    }

    // ---- coefficient tables in LDS (uniform-address 16-byte reads broadcast them to the lanes; as scalar operands the
    //      unrolled loops below need hundreds of SGPRs and spill them to vector lanes)
    for (int i = tid; i < cout * 12; i += 256) {
        const int co = i / 12, m = i - 12 * co;
        tb[i] = m < LW_ROWS ? a.lora_b[9 * co + m] : 0.f;
    }
    for (int i = tid; i < cin * 28; i += 256) {
        const int ci = i / 28, j = i - 28 * ci, q = j / 9, tp = j - 9 * q;
        const int ii = (9 * ci) / n3;                  // the third that holds the channel's first tap (and all nine, if it is a "whole" channel)
        const int n = 9 * ci + tp - ii * n3;
        ta[i] = (j < 27 && n < n3) ? a.lora_a[q * n3 + n] : 0.f;
    }

    // dy / mask of one tile, this thread's pixel and channel slice (co = h, h + NH, ...), into registers
    float dyv[COS], mkv[COS];
    auto fetch_dy = [&](int tile) {
        int t = tile;
        const int x0 = (t % a.tiles_x) * 32;
        t /= a.tiles_x;
        const int y0 = (t % a.tiles_y) * TH;
        const int b = t / a.tiles_y;
        const int gy = y0 + prow, gx = x0 + pcol;
        const bool inside = gy < H && gx < W;
        const long long pix = (long long)gy * W + gx;
        const float* dyp = a.dy + (long long)b * a.dy_bs + pix;
        const float* mkp = a.mask ? a.mask + (long long)b * a.mask_bs + pix : nullptr;
#pragma unroll
        for (int k = 0; k < COS; ++k) {
            const int co = h + k * NH;
            const bool ok = inside && co < cout;
            dyv[k] = ok ? dyp[(long long)co * HW] : 0.f;
            mkv[k] = (ok && mkp) ? mkp[(long long)co * HW] : 1.f;
        }
    };
    if ((int)blockIdx.x < a.ntiles) fetch_dy(blockIdx.x);
#ifdef YNET_LW_PROFILE
    unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = tp;
#endif

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t = tile;
        const int x0 = (t % a.tiles_x) * 32;
        t /= a.tiles_x;
        const int y0 = (t % a.tiles_y) * TH;
        const int b = t / a.tiles_y;

        // -- 0. queue the x tile (the previous tile's MFMA phase ended with a barrier: the image is free).  One DMA
        //       instruction per source (wave-uniform descriptor), issued by the lanes whose quad belongs to it; quads outside
        //       the image / pad quads go with the first source and the out-of-range marker (the range check writes zeros):
        //       every quad of the image is written exactly once
        {
            unsigned xo[XI_MAX];
#pragma unroll
            for (int k = 0; k < XI_MAX; ++k) {
                const int gy = y0 + (xrc[k] >> 8) - 1, gx = x0 + (xrc[k] & 255) - 4;
                const bool ok = xrc[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                xo[k] = ok ? xcoff[k] + (unsigned)(gy * W + gx) * 4u : 0x80000000u;
            }
#pragma unroll 1
            for (int s_ = 0; s_ < a.nsrc; ++s_) {
                const float* base = a.src[s_].p + (long long)b * a.src[s_].bs;
                const unsigned long long ub = (unsigned long long)base;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ub), hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(((unsigned long long)hi << 32) | lo), 0, (unsigned)a.src[s_].c * plane_bytes, 0x00020000);
#pragma unroll
                for (int k = 0; k < XI_MAX; ++k) {
                    if (k * 256 < nquads) {            // (uniform)
                        const int with_src = xo[k] == 0x80000000u ? 0 : xsid[k];
                        if (tid + k * 256 < nquads && with_src == s_) lw_dma16(r, xs + (k * 256 + wave * 64) * 4, xo[k]);
                    }
                }
            }
        }
        LW_STAMP(0);      // DMA issue
        // -- 1a. dy' and DP (vector ALU; does not need the x tile).  dy / mask of this tile were fetched into registers
        //        during the previous tile's MFMA phase.
        float dp[LW_ROWS], xp[LW_ROWS];
#pragma unroll
        for (int m = 0; m < LW_ROWS; ++m) dp[m] = xp[m] = 0.f;
#pragma unroll
        for (int k = 0; k < COS; ++k) {
            const int co = h + k * NH;
            if (co < cout) {                           // (uniform per slice)
                const float d = mkv[k] > 0.f ? dyv[k] : 0.f;
                ds[co * DCH + p] = d;
                const f32x4* cb = reinterpret_cast<const f32x4*>(tb + co * 12);
                const f32x4 c0 = cb[0], c1 = cb[1], c2 = cb[2];
#pragma unroll
                for (int m = 0; m < LW_ROWS; ++m)       // B[3 co + ii][q], m = 3 ii + q
                    dp[m] = __builtin_fmaf(m < 4 ? c0[m & 3] : (m < 8 ? c1[m & 3] : c2[m & 3]), d, dp[m]);
            }
        }
        LW_STAMP(1);      // DP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // x tile landed (every wave's DMAs), dy' complete
        LW_STAMP(2);      // DMA wait + barrier

        // -- 1b. XP from the x tile.  Third ii of the filter columns is u in [ii * 3cin, (ii + 1) * 3cin); the input channels
        //        whose nine taps lie wholly inside it go through the unrolled loop (slice h takes every NH-th of them: nine
        //        LDS reads at immediate offsets, 27 FMAs with scalar coefficients), the columns of the (at most two) channels
        //        that straddle a boundary one by one (slice 0).
        {
            const float* xb = xs + xoff_p;
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                const int lo_u = ii * n3, hi_u = lo_u + n3;
                const int ci_lo = (lo_u + 8) / 9, ci_hi = hi_u / 9;
                const int head_end = min(9 * ci_lo, hi_u);
                const int tail_begin = max(9 * ci_hi, head_end);
#pragma unroll 2
                for (int ci = ci_lo + h; ci < ci_hi; ci += NH) {
                    const float* xc = xb + ci * XCH;
                    const f32x4* cv = reinterpret_cast<const f32x4*>(ta + ci * 28);
                    float cf[28], xv[9];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const f32x4 c4 = cv[j];
                        cf[4 * j] = c4[0];
                        cf[4 * j + 1] = c4[1];
                        cf[4 * j + 2] = c4[2];
                        cf[4 * j + 3] = c4[3];
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t) xv[t] = xc[(t / 3) * TCOLS + (t % 3)];
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        s0 = __builtin_fmaf(cf[t], xv[t], s0);
                        s1 = __builtin_fmaf(cf[9 + t], xv[t], s1);
                        s2 = __builtin_fmaf(cf[18 + t], xv[t], s2);
                    }
                }
                if (h == 0) {
                    auto one = [&](int u) {
                        const int ci = u / 9, tp = u - 9 * ci, ky = tp / 3, kx = tp - 3 * ky, n = u - lo_u;
                        const float xv = xb[ci * XCH + ky * TCOLS + kx];
                        s0 = __builtin_fmaf(la[n], xv, s0);
                        s1 = __builtin_fmaf(la[n3 + n], xv, s1);
                        s2 = __builtin_fmaf(la[2 * n3 + n], xv, s2);
                    };
#pragma unroll 1
                    for (int u = lo_u; u < head_end; ++u) one(u);
#pragma unroll 1
                    for (int u = tail_begin; u < hi_u; ++u) one(u);
                }
                xp[3 * ii + 0] = s0;
                xp[3 * ii + 1] = s1;
                xp[3 * ii + 2] = s2;
            }
        }
        LW_STAMP(3);      // XP
        // the NH slices add their parts into the planes one after the other (fixed order)
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            if (h == hh) {
#pragma unroll
                for (int m = 0; m < LW_ROWS; ++m) {
                    if (hh == 0) {
                        dpl[m * PL + p] = dp[m];
                        xpl[m * PL + p] = xp[m];
                    } else {
                        dpl[m * PL + p] += dp[m];
                        xpl[m * PL + p] += xp[m];
                    }
                }
            }
            __syncthreads();
        }

        LW_STAMP(4);      // plane combine (NH barriers)
        // the next tile's dy / mask: in flight during the MFMA phase
        if (tile + (int)gridDim.x < a.ntiles) fetch_dy(tile + gridDim.x);

        LW_STAMP(5);      // dy prefetch issue
        // -- 2. pixel sums on the matrix cores: K-step s = pixels 4 s .. 4 s + 3 of tile row s / 8.  Fully unrolled: every
        //       LDS read is `base register + immediate`; the operands of step s + 1 are read before the MFMAs of step s.
        {
            const float* a_dp = dpl + arow;
            const float* a_xp = xpl + arow;
            const float* b_e = ds + ebase;
            const float* b_g[NTG];
#pragma unroll
            for (int j = 0; j < NTG; ++j) b_g[j] = xs + gbase[j];
            auto run = [&](auto with_e) {
                constexpr bool WE = decltype(with_e)::value;
                float ac, bc[NTG], ec = 0.f, dc = 0.f, an, bn[NTG], en = 0.f, dn = 0.f;
                ac = a_dp[0];
#pragma unroll
                for (int j = 0; j < NTG; ++j) bc[j] = b_g[j][0];
                if (WE) {
                    ec = a_xp[0];
                    dc = b_e[0];
                }
#pragma unroll
                for (int s_ = 0; s_ < C::KSTEPS; ++s_) {
                    if (s_ + 1 < C::KSTEPS) {
                        const int sn = s_ + 1, xo = (sn >> 3) * TCOLS + (sn & 7) * 4;
                        an = a_dp[4 * sn];
#pragma unroll
                        for (int j = 0; j < NTG; ++j) bn[j] = b_g[j][xo];
                        if (WE) {
                            en = a_xp[4 * sn];
                            dn = b_e[4 * sn];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // reads of the next step, THEN the MFMAs of this one
#pragma unroll
                    for (int j = 0; j < NTG; ++j) accg[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, bc[j], accg[j], 0, 0, 0);
                    if (WE) acce = __builtin_amdgcn_mfma_f32_16x16x4f32(ec, dc, acce, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ac = an;
#pragma unroll
                    for (int j = 0; j < NTG; ++j) bc[j] = bn[j];
                    if (WE) {
                        ec = en;
                        dc = dn;
                    }
                }
            };
            if (has_e) run(std::true_type{});
            else run(std::false_type{});
        }
        LW_STAMP(6);      // MFMA phase
        __syncthreads();                               // every wave is done with the tile images
        LW_STAMP(7);      // end barrier
    }
#ifdef YNET_LW_PROFILE
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(a.prof + i, pr[i]);
        atomicAdd(a.prof + 8, __builtin_amdgcn_s_memtime() - t_begin);
        atomicAdd(a.prof + 9, 1ull);
    }
#endif

    // ---- partial sums of this workgroup: G [9][9 cin], then E [9][cout]; D layout: column = lane & 15, row = 4 (lane >> 4) + reg
    float* pg = a.partial + (long long)blockIdx.x * (LW_ROWS * (ncol + cout));
    float* pe = pg + LW_ROWS * ncol;
#pragma unroll
    for (int j = 0; j < NTG; ++j) {
        const int u = (wave + 4 * j) * 16 + r16;
        if (u < ncol) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * kq + e;
                if (m < LW_ROWS) pg[m * ncol + u] = accg[j][e];
            }
        }
    }
    if (has_e) {
        const int co = wave * 16 + r16;
        if (co < cout) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * kq + e;
                if (m < LW_ROWS) pe[m * cout + co] = acce[e];
            }
        }
    }
}

// dA[q][n] = s * sum_wg sum_ii G_wg[3 ii + q][ii * 3cin + n];  dB[(3 co + ii) * 3 + q] = s * sum_wg E_wg[3 ii + q][co].
// Thread (o, g) of a block sums the workgroups wg = g mod 8 of output base + o, the eight chains are added in order.
__global__ __launch_bounds__(256) void lora_wgrad_reduce_kernel(const float* __restrict__ partial, int nwg, int cin, int cout,
                                                                float scale, float* __restrict__ d_a, float* __restrict__ d_b) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n3 = 3 * cin, ncol = 9 * cin, na = LW_RQ * n3, nb = 9 * cout;
    const long long stride = (long long)LW_ROWS * (ncol + cout);
    for (int base = blockIdx.x * 32; base < na + nb; base += gridDim.x * 32) {
        const int i = base + o;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (i < na) {
            const int q = i / n3, n = i - q * n3;
            const float* p0 = partial + (0 * LW_RQ + q) * ncol + 0 * n3 + n;
            const float* p1 = partial + (1 * LW_RQ + q) * ncol + 1 * n3 + n;
            const float* p2 = partial + (2 * LW_RQ + q) * ncol + 2 * n3 + n;
            int w = g;
            for (; w + 24 < nwg; w += 32) {            // 12 independent loads in flight (a dependent chain of ~64 L2 round trips otherwise)
                float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t0 += p0[(w + 8 * k) * stride];
                    t1 += p1[(w + 8 * k) * stride];
                    t2 += p2[(w + 8 * k) * stride];
                }
                s0 += t0;
                s1 += t1;
                s2 += t2;
            }
            for (; w < nwg; w += 8) {
                s0 += p0[w * stride];
                s1 += p1[w * stride];
                s2 += p2[w * stride];
            }
        } else if (i < na + nb) {
            const int f = i - na, co = f / 9, m = f - 9 * co;
            const float* p0 = partial + (long long)LW_ROWS * ncol + m * cout + co;
            int w = g;
            for (; w + 56 < nwg; w += 64) {
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t0 += p0[(w + 8 * k) * stride];
                    t1 += p0[(w + 32 + 8 * k) * stride];
                }
                s0 += t0;
                s1 += t1;
            }
            for (; w < nwg; w += 8) s0 += p0[w * stride];
        }
        red[g][o] = (s0 + s1) + s2;
        __syncthreads();
        if (g == 0 && i < na + nb) {
            float t = red[0][o];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += red[k][o];
            if (i < na) d_a[i] = scale * t;
            else d_b[i - na] = scale * t;
        }
        __syncthreads();
    }
}

// tile height by channels (LDS: at least two workgroups per CU); 0 = not served.  Measured at B = 32 (tools/lora_wgrad_bench.py):
// 32 -> 64 @ 64^2 takes 57 us with 2-row tiles (three resident workgroups) against 70 us with 4-row tiles.
static int lw_tile_rows(int cin, int cout) {
    if (cin <= 32 && cout <= 32) return 4;
    if (cin <= 64 && cout <= 64) return 2;
    return 0;
}

static int lw_grid(int ntiles, int lds_bytes) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    int per_cu = (160 * 1024) / lds_bytes;      // resident workgroups per CU by LDS (registers allow 2: __launch_bounds__(256, 2))
    static const int cap = getenv("YNET_LW_WG_PER_CU") ? atoi(getenv("YNET_LW_WG_PER_CU")) : 3;
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    const int slots = per_cu * cus;
    return ntiles < slots ? ntiles : slots;
}

template <int TH, int NTG, int COS, int CIN_T = 0, int COUT_T = 0>
static int launch_lora_wgrad(const LoraWgArgs& a, int grid, hipStream_t st) {
    static bool attr_dev[YNET_MAX_DEV] = {false};
    bool& attr_set = attr_dev[ynet_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lora_wgrad_kernel<TH, NTG, COS, CIN_T, COUT_T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int lds = lw_lds_floats(TH, a.cin, a.cout) * 4;
#ifdef YNET_LW_PROFILE
    static unsigned long long* prof_dev = nullptr;
    if (!prof_dev) (void)hipMalloc(&prof_dev, 128);
    (void)hipMemsetAsync(prof_dev, 0, 128, st);
    LoraWgArgs ap = a;
    ap.prof = prof_dev;
    hipLaunchKernelGGL((lora_wgrad_kernel<TH, NTG, COS, CIN_T, COUT_T>), dim3(grid), dim3(256), lds, st, ap);
    {
        unsigned long long h[16];
        (void)hipMemcpyAsync(h, prof_dev, 128, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        const double tot = (double)h[8];
        static const char* nm[8] = {"dma-issue", "DP", "dma-wait+barrier", "XP", "combine", "dy-prefetch", "MFMA", "end-barrier"};
        fprintf(stderr, "lora_wgrad<%d,%d,%d,%d,%d> grid %d tiles %d waves %llu avg cycles/wave %.0f:", TH, NTG, COS, CIN_T, COUT_T, grid, a.ntiles, h[9], tot / (double)h[9]);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h[i] / tot);
        fprintf(stderr, "\n");
    }
#else
    hipLaunchKernelGGL((lora_wgrad_kernel<TH, NTG, COS, CIN_T, COUT_T>), dim3(grid), dim3(256), lds, st, a);
#endif
    return ynet_check_launch("lora_conv2d_wgrad");
}

extern "C" {

// 1 if ynet_lora_conv2d_wgrad serves this layer (3x3, rank 1, cin / cout <= 64, W % 4 == 0)
int ynet_lora_conv2d_wgrad_supported(int cin, int cout, int K, int r, int W) {
    static const int on = getenv("YNET_LORA_WGRAD") ? atoi(getenv("YNET_LORA_WGRAD")) : 1;
    return (on && K == 3 && r == 1 && cin >= 1 && cout >= 1 && (W & 3) == 0 && lw_tile_rows(cin, cout) != 0) ? 1 : 0;
}

// 1 where the projected form is also the FASTER one at the encoder's shapes (measured, MI355X, B = 32): the layers with 64
// output channels (32 -> 64: 57 vs 69 us; 64 -> 64 @ 64^2: 83 vs 115; @ 32^2: 35 vs 43; @ 16^2: 26 vs 35).  At 14 -> 32 @ 256^2
// and 32 -> 32 @ 128^2 the per-pixel vector work of the projections (27 cin + 9 cout FMAs) costs more than the tuned full
// filter gradient saves (340 vs 254 us, 139 vs 117 us): those keep the two-call chain.  YNET_LORA_WGRAD=2: everywhere.
int ynet_lora_conv2d_wgrad_preferred(int cin, int cout, int K, int r, int W) {
    static const int on = getenv("YNET_LORA_WGRAD") ? atoi(getenv("YNET_LORA_WGRAD")) : 1;
    if (!ynet_lora_conv2d_wgrad_supported(cin, cout, K, r, W)) return 0;
    return (on >= 2 || cout > 32) ? 1 : 0;
}

long long ynet_lora_conv2d_wgrad_workspace_floats(int cin, int cout) {
    return 1024ll * LW_ROWS * (9ll * cin + cout);       // one slab per resident workgroup (2 per CU, <= 512 CUs)
}

int ynet_lora_conv2d_wgrad(const float* const* src, const int* src_c, const long long* src_bs, int nsrc,
                           const float* dy, long long dy_bs, const float* mask, long long mask_bs,
                           const float* lora_a, const float* lora_b, float scale, float* d_a, float* d_b,
                           float* workspace, int B, int H, int W, int cout, int K, int r, void* stream) {
    YNET_REQUIRE(nsrc >= 1 && nsrc <= YNET_MAX_SRC, "lora_conv2d_wgrad: 1..%d sources supported", YNET_MAX_SRC);
    YNET_REQUIRE(dy && lora_a && lora_b && d_a && d_b && workspace, "lora_conv2d_wgrad: null pointer");
    YNET_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0, "lora_conv2d_wgrad: empty problem");
    LoraWgArgs a{};
    a.nsrc = nsrc;
    a.cin = 0;
    auto misaligned = [](const void* p, long long bs) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0 || (bs & 3) != 0; };
    for (int i = 0; i < nsrc; ++i) {
        YNET_REQUIRE(src[i] != nullptr && src_c[i] > 0, "lora_conv2d_wgrad: source %d is null/empty", i);
        YNET_REQUIRE(!misaligned(src[i], src_bs[i]), "lora_conv2d_wgrad: source %d is not 16-byte aligned", i);
        a.src[i] = YSrc{src[i], src_c[i], src_bs[i], 0};
        a.cin += src_c[i];
    }
    YNET_REQUIRE(ynet_lora_conv2d_wgrad_supported(a.cin, cout, K, r, W), "lora_conv2d_wgrad: cin %d cout %d K %d r %d W %d is not served (3x3, r = 1, channels <= 64, W %% 4 == 0)", a.cin, cout, K, r, W);
    static const int th2_all = getenv("YNET_LW_TH2") ? atoi(getenv("YNET_LW_TH2")) : 0;      // (experiments: 2-row tiles everywhere)
    const int th = th2_all ? 2 : lw_tile_rows(a.cin, cout);
    a.dy = dy;
    a.dy_bs = dy_bs;
    a.mask = mask;
    a.mask_bs = mask_bs;
    a.lora_a = lora_a;
    a.lora_b = lora_b;
    a.partial = workspace;
    a.B = B;
    a.H = H;
    a.W = W;
    a.cout = cout;
    a.tiles_x = ceil_div(W, 32);
    a.tiles_y = ceil_div(H, th);
    a.ntiles = B * a.tiles_x * a.tiles_y;
    YNET_REQUIRE(a.cin * lw_xq(th) <= (th == 4 ? 9 : 11) * 256, "lora_conv2d_wgrad: x tile of %d channels exceeds the DMA plan", a.cin);
    const int grid = lw_grid(a.ntiles, lw_lds_floats(th, a.cin, cout) * 4);
    hipStream_t st = (hipStream_t)stream;
    const int ntg = ceil_div(ceil_div(9 * a.cin, 16), 4);
    int rc;
    // the encoder's shapes: channel counts known at compile time
    if (th == 2 && a.cin == 14 && cout == 32) rc = launch_lora_wgrad<2, 2, 8, 14, 32>(a, grid, st);
    else if (th == 2 && a.cin == 32 && cout == 32) rc = launch_lora_wgrad<2, 5, 8, 32, 32>(a, grid, st);
    else if (th == 2 && a.cin == 32 && cout == 64) rc = launch_lora_wgrad<2, 5, 16, 32, 64>(a, grid, st);
    else if (a.cin == 14 && cout == 32) rc = launch_lora_wgrad<4, 2, 16, 14, 32>(a, grid, st);
    else if (a.cin == 32 && cout == 32) rc = launch_lora_wgrad<4, 5, 16, 32, 32>(a, grid, st);
    else if (a.cin == 32 && cout == 64) rc = launch_lora_wgrad<4, 5, 32, 32, 64>(a, grid, st);
    else if (a.cin == 64 && cout == 64) rc = launch_lora_wgrad<2, 9, 16, 64, 64>(a, grid, st);
    else if (th == 4) {      // two channel slices
        if (cout <= 32) rc = ntg <= 2 ? launch_lora_wgrad<4, 2, 16>(a, grid, st) : launch_lora_wgrad<4, 5, 16>(a, grid, st);
        else rc = ntg <= 2 ? launch_lora_wgrad<4, 2, 32>(a, grid, st) : launch_lora_wgrad<4, 5, 32>(a, grid, st);
    } else {            // four channel slices
        rc = launch_lora_wgrad<2, 9, 16>(a, grid, st);
    }
    if (rc) return rc;
    const int nout = LW_RQ * 3 * a.cin + 9 * cout;
    hipLaunchKernelGGL(lora_wgrad_reduce_kernel, dim3(ceil_div(nout, 32)), dim3(256), 0, st, workspace, grid, a.cin, cout, scale, d_a, d_b);
    return ynet_check_launch("lora_conv2d_wgrad(reduce)");
}

}  // extern "C"
