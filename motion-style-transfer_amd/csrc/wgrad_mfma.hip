// Filter / bias gradient of the 3x3 (1x1, 5x5) "same" convolution on the fp32 matrix cores.
//
// Replaces ATen convolution_backward's weight/bias outputs (K4 in SURVEY.md section 2.1) for the
// trainable convs of models/ynet.py (all of them in train_net=train/all, the adapted encoder convs
// in mosa_*, where dW then feeds ynet_lora_grad).
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} dy[b,co,y,x] * [yact[b,co,y,x] > 0] * x[b,ci,y+ky-P,x+kx-P]
//   db[co]             = sum_{b,y,x} dy[b,co,y,x] * [yact > 0]
//
// wgrad_dma_kernel<MASK, TH_> (3x3, aligned planes; described at its definition): v_mfma_f32_16x16x4_f32 with K = 4
// consecutive pixels, LDS-DMA double buffer, waves split the 32 co x 32 ci x 9 output block.
// wgrad_mfma_kernel<K,MASK> (1x1, 5x5, unaligned planes): v_mfma_f32_32x32x2_f32, M = 32 output channels (A = masked
// dy tile, LDS [co][pixel], odd stride), N = 32 columns of the flattened (ci, kx) index (B = x tile with halo, LDS
// [ci][row][col], odd channel stride; flattening kx into N means Cin = 14 costs 2 column tiles instead of 3 padded
// taps), K = pixels, two per instruction; one accumulator tile per (column tile, ky); db from one more MFMA per K-step
// against a B operand of ones; persistent workgroups walk 4x32-pixel tiles, the buffer_loads (hardware range check =
// zero padding) of tile t+1 are issued before the MFMA loop of tile t and written to LDS after it; pixels are split
// over the 4 waves (summed through LDS in wave order).
// Both split the pixels over `nsplit` workgroups and reduce the partials with reduce_partials_kernel in a fixed
// order: bitwise reproducible, no float atomics.
#include "ynet_common.h"
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>

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
    int seg, nseg_y, nsegs;   // wgrad_roll_kernel: two-row steps per segment, segments per strip, segments in all
    int debug;                // timing ablations (YNET_WGRAD_DEBUG; results are wrong): 1 no DMA after a segment's start, 2 no MFMA loop
#ifdef YNET_WG_PROFILE
    unsigned long long* prof;      // development build: per-phase cycle sums
#endif
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


// ================================================================================================
// LDS-DMA generation of the 3x3 filter gradient (v_mfma_f32_16x16x4_f32).
//   * tiles of 2 x 32 pixels; x (with halo, as aligned 16-byte quads: rows of 40 floats), dy and the
//     ReLU mask go global -> LDS by `buffer_load_dwordx4 ... lds` into the OTHER half of a double
//     buffer while the MFMA loop reads this half: one barrier per tile, no staging registers;
//   * the four waves split the OUTPUT (16 co x 16 ci x 9 taps = 36 accumulator registers each), not the
//     pixels, so nothing is summed across waves when the block is a full 32 x 32; a partial block
//     (Cin = 14, Cout = 12 ...) hands its spare waves every 2nd / 4th K-step instead and those are
//     summed through LDS in fixed order;
//   * K = 4 pixels per instruction, every second one of an 8-pixel group: lane (r16, kq) reads channel r16,
//     pixels 8 g + 2 kq + {0, 1} as ONE 8-byte LDS read for the two K-steps of the group; channel strides are 4 mod 32
//     floats = conflict-free for ds_read_b64 (64 banks per 32 lanes; details at the reads);
//   * the mask is applied to the A operand of a group (second LDS read + 2 v_cndmask); db is a VALU
//     sum of that same operand.
// Needs W % 4 == 0 and 16-byte aligned planes; everything else stays on wgrad_mfma_kernel.
// ================================================================================================
typedef __attribute__((address_space(3))) void* wg_lds_ptr_t;
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t r, const float* lds, unsigned byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (wg_lds_ptr_t)lds, 16, byte_off, 0, 0, 0);
}

__device__ __forceinline__ void wg_dma16s(__amdgpu_buffer_rsrc_t r, const float* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (wg_lds_ptr_t)lds, 16, voff, soff, 0, 0);
}

template <bool MASK, int TH_>
struct WgDmaCfg {
    static constexpr int TH = TH_, TW = 32, TROWS = TH + 2, TCOLS = TW + 8;
    // quads per x / dy channel incl. pad quads, = 1 mod 8 so that the channel stride in floats is 4 mod 32
    static constexpr int XDATA = TROWS * TCOLS / 4, DDATA = TH * TW / 4;
    static constexpr int XQ = (XDATA + 1 + 6) / 8 * 8 + 1;
    static constexpr int XCH = XQ * 4;
    static constexpr int DQ = (DDATA + 1 + 6) / 8 * 8 + 1;
    static constexpr int DCH = DQ * 4;
    // + 128 floats of slack each: the last, partial DMA instruction of a tile image is issued by a whole wave
    // (no per-lane predicate = no vector-ALU compare); its surplus lanes carry the out-of-range marker and write
    // zeros into the slack
    static constexpr int XS = 32 * XCH + 128, DS = 32 * DCH + 128;
    static constexpr int BUF = XS + DS * (MASK ? 2 : 1);     // x tile, dy tile, ReLU-mask tile (the activation y)
    static constexpr int LDS_BYTES = 2 * BUF * 4;
    static constexpr int XI = (32 * XQ + 255) / 256;     // x DMA instructions per thread per tile
    static constexpr int DI = (32 * DQ + 255) / 256;     // dy DMA instructions (and mask loads) per thread per tile
    static constexpr int KSTEPS = TH * TW / 4;
    static_assert(XCH % 32 == 4 && DCH % 32 == 4, "bank-conflict-free channel strides");
    static_assert(3 * 37 * 64 <= 2 * BUF, "cross-wave reduction scratch fits");
    static_assert(KSTEPS % 16 == 0, "a wave runs KSTEPS / 2 / rpN groups of two K-steps in pairs (rpN <= 4)");
};

template <bool MASK, int TH_>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradArgs a) {
    using C = WgDmaCfg<MASK, TH_>;
    constexpr int TH = C::TH, TW = C::TW, TCOLS = C::TCOLS, XQ = C::XQ, XCH = C::XCH, DQ = C::DQ, DCH = C::DCH;
    constexpr int XI = C::XI, DI = C::DI, KK = 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto xs_of = [&](int b) { return smem + b * C::BUF; };
    auto ds_of = [&](int b) { return smem + b * C::BUF + C::XS; };
    auto ms_of = [&](int b) { return smem + b * C::BUF + C::XS + C::DS; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    int bid = blockIdx.x;
    const int cib = bid % a.ci_blks;
    bid /= a.ci_blks;
    const int cob = bid % a.co_blks;
    const int split = bid / a.co_blks;
    const int HW = __builtin_amdgcn_readfirstlane(a.H * a.W);
    const unsigned plane_bytes = (unsigned)HW * 4u;
    const int ci0 = cib * 32, co0 = cob * 32;
    const int ncib = min(32, a.cin - ci0), nco = min(32, a.cout - co0);
    const bool want_bias = (a.partial_b != nullptr) && cib == 0;
    // wave roles: (16-wide co block, 16-wide ci block, K-step residue)
    const int cbN = nco > 16 ? 2 : 1, ibN = ncib > 16 ? 2 : 1, rpN = 4 / (cbN * ibN);
    const int cb = wave % cbN, ib = (wave / cbN) % ibN, rp = wave / (cbN * ibN);

    // ---- tile-independent part of the DMA addressing
    // x: LDS image = 32 channels x XQ quads, linear in the quad index q = tid + 256 k
    int xsrc[XI];            // source of the quad's channel (-1: nothing to fetch)
    unsigned xcoff[XI];      // byte offset of the channel's plane inside that source's image
    int xrow[XI], xcol[XI];  // tile row (0 .. TH+1) and first column (-4, 0, 4 ...) of the quad
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int q = tid + k * 256;
        const int ch = q / XQ, within = q - ch * XQ;
        xrow[k] = within / (TCOLS / 4);
        xcol[k] = (within - xrow[k] * (TCOLS / 4)) * 4 - 4;
        int c = ci0 + ch, sid = -1;
        unsigned off = 0;
        if (q < 32 * XQ && within < C::XDATA && ch < ncib) {
#pragma unroll
            for (int s = 0; s < YNET_MAX_SRC; ++s) {
                if (sid < 0 && s < a.nsrc) {
                    if (c < a.src[s].c) {
                        sid = s;
                        off = (unsigned)c * plane_bytes;
                    } else {
                        c -= a.src[s].c;
                    }
                }
            }
        }
        xsrc[k] = sid;
        xcoff[k] = off;
    }
    unsigned dcoff[DI];
    int drow[DI], dcol[DI];  // drow < 0: pad quad / past the image
#pragma unroll
    for (int k = 0; k < DI; ++k) {
        const int q = tid + k * 256;
        const int ch = q / DQ, within = q - ch * DQ;
        const bool ok = q < 32 * DQ && within < C::DDATA;
        drow[k] = ok ? within / (TW / 4) : -1;
        dcol[k] = (within % (TW / 4)) * 4;
        dcoff[k] = (unsigned)ch * plane_bytes;       // channels >= nco fall outside the descriptor: zero
    }

    auto decode = [&](int t) {
        WgTile c;
        c.x0 = (t % a.tiles_x) * TW;
        t /= a.tiles_x;
        c.y0 = (t % a.tiles_y) * TH;
        c.b = t / a.tiles_y;
        return c;
    };
    // ---- fast path of the addressing (tiles whose halo rows lie inside the image, W % 32 == 0): the per-lane
    // offsets are STATIC (relative to the tile's top-left halo element / first pixel) and the tile position goes
    // into the scalar offset of the DMA, so a tile costs no vector-ALU instruction at all; tiles of the first /
    // last column use a second / third static set whose outer halo quads carry the out-of-range marker.  (A VALU
    // instruction issues once per ~MFMA slot of the other wave on the SIMD: the ~100 of the generic path below
    // took about as long as the tile's 144 MFMAs.)
    // static offsets for a tile of an interior column / the first column (left halo quads zeroed) / the last / both
    unsigned xs_in[XI], xs_l[XI], xs_r[XI], xs_lr[XI], dstat[DI];     // (separate arrays: a 2-D one selected by value goes to scratch)
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const bool ok = xsrc[k] >= 0;
        const unsigned in = ok ? xcoff[k] + (unsigned)(xrow[k] * a.W + xcol[k] + 4) * 4u : 0x80000000u;
        const bool lh = xcol[k] < 0, rh = xcol[k] >= TW;
        xs_in[k] = in;
        xs_l[k] = lh ? 0x80000000u : in;
        xs_r[k] = rh ? 0x80000000u : in;
        xs_lr[k] = (lh || rh) ? 0x80000000u : in;
    }
#pragma unroll
    for (int k = 0; k < DI; ++k) dstat[k] = drow[k] >= 0 ? dcoff[k] + (unsigned)(drow[k] * a.W + dcol[k]) * 4u : 0x80000000u;
    const int H = a.H, W = a.W;
    const bool regular = (W % TW) == 0 && (H % TH) == 0;

    // queue the DMAs of one tile given its per-lane offsets (+ scalar offsets xso / dso).  The arguments are re-read
    // through the kernarg segment (s_load) so that they do not occupy SGPRs across the MFMA loop (spills to VGPR
    // lanes are reloaded by v_readlane, a vector-ALU instruction).
    typedef const __attribute__((address_space(4))) WgradArgs* wg_kargs_t;
#ifdef YNET_WG_PROFILE
    unsigned long long prof_dma = 0;
#endif
    auto queue = [&](const WgTile& t, int buf, const unsigned* xo, unsigned xso, const unsigned* dofs, unsigned dso) {
        wg_kargs_t ka = (wg_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        float* xs = xs_of(buf);
        const int nsrc = ka->nsrc;
#ifdef YNET_WG_PROFILE
        const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
        constexpr int XFULL = 32 * XQ / 256, DFULL = 32 * DQ / 256;      // instructions issued by all four waves
        static_assert(32 * XQ - XFULL * 256 <= 64 && 32 * DQ - DFULL * 256 <= 64, "the partial instruction fits one wave");
        if (nsrc == 1) {        // lanes without data carry the marker: no predicate at all
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ka->src[0].p + (long long)t.b * ka->src[0].bs), 0, (unsigned)ka->src[0].c * plane_bytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < XFULL; ++k) wg_dma16s(r, xs + (k * 256 + wave * 64) * 4, xo[k], xso);
            if (XI > XFULL && wave == 0) wg_dma16s(r, xs + (XFULL * 256) * 4, xo[XI - 1], xso);
        } else {
#pragma unroll 1
            for (int s = 0; s < nsrc; ++s) {
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(ka->src[s].p + (long long)t.b * ka->src[s].bs), 0, (unsigned)ka->src[s].c * plane_bytes, 0x00020000);
#pragma unroll
                for (int k = 0; k < XI; ++k)
                    if (xsrc[k] == s) wg_dma16s(r, xs + (k * 256 + wave * 64) * 4, xo[k], xso);
            }
        }
        {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ka->dy + (long long)t.b * ka->dy_bs + (long long)co0 * HW), 0, (unsigned)nco * plane_bytes, 0x00020000);
            float* ds = ds_of(buf);
#pragma unroll
            for (int k = 0; k < DFULL; ++k) wg_dma16s(r, ds + (k * 256 + wave * 64) * 4, dofs[k], dso);
            if (DI > DFULL && wave == 0) wg_dma16s(r, ds + (DFULL * 256) * 4, dofs[DI - 1], dso);
        }
        if (MASK) {
            // the activation tile lands next to the dy tile; the select happens on the A operand, behind the MFMAs of
            // the previous K-step.  (Holding the mask quads in registers and rewriting the dy tile in LDS before the
            // barrier looked equal in an isolated launch but cost 7-19 % per launch inside the training step.)
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ka->mask + (long long)t.b * ka->mask_bs + (long long)co0 * HW), 0, (unsigned)nco * plane_bytes, 0x00020000);
            float* ms = ms_of(buf);
#pragma unroll
            for (int k = 0; k < DFULL; ++k) wg_dma16s(r, ms + (k * 256 + wave * 64) * 4, dofs[k], dso);
            if (DI > DFULL && wave == 0) wg_dma16s(r, ms + (DFULL * 256) * 4, dofs[DI - 1], dso);
        }
#ifdef YNET_WG_PROFILE
        prof_dma += __builtin_amdgcn_s_memtime() - q0;
#endif
    };
    // Queue the DMAs of a tile.  (Handing them out inside the MFMA loop instead was measured 6-15 % slower: a
    // buffer_load ... lds holds the wave's instruction stream for ~50-60 cycles wherever it is placed.)
    auto issue = [&](const WgTile& t, int buf) {
        // (the scalar offset of the x tile must not be negative: row y0-1 = 0 of the first tile column is left to the
        // generic path)
        const bool fast = regular && t.y0 >= 1 && t.y0 + TH + 1 <= H && (t.y0 - 1) * W + t.x0 - 4 >= 0;
        if (fast) {
            const unsigned xso = (unsigned)((t.y0 - 1) * W + t.x0 - 4) * 4u;
            const unsigned dso = (unsigned)(t.y0 * W + t.x0) * 4u;
            const bool at_left = t.x0 == 0, at_right = t.x0 + TW >= W;
            if (!at_left && !at_right) {
                queue(t, buf, xs_in, xso, dstat, dso);              // interior: no vector-ALU instruction
            } else {
                unsigned xo[XI];                                    // first / last column: one or two selects per quad
#pragma unroll
                for (int k = 0; k < XI; ++k) {
                    const unsigned l = at_right ? xs_lr[k] : xs_l[k];
                    xo[k] = at_left ? l : xs_r[k];
                }
                queue(t, buf, xo, xso, dstat, dso);
            }
        } else {
            unsigned xo[XI], dofs[DI];
#pragma unroll
            for (int k = 0; k < XI; ++k) {
                const int gy = t.y0 + xrow[k] - 1, gx = t.x0 + xcol[k];
                const bool ok = xsrc[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                xo[k] = ok ? xcoff[k] + (unsigned)(gy * W + gx) * 4u : 0x80000000u;
            }
#pragma unroll
            for (int k = 0; k < DI; ++k) {
                const int gy = t.y0 + drow[k], gx = t.x0 + dcol[k];
                const bool ok = drow[k] >= 0 && gy < H && gx < W;
                dofs[k] = ok ? dcoff[k] + (unsigned)(gy * W + gx) * 4u : 0x80000000u;
            }
            queue(t, buf, xo, 0u, dofs, 0u);
        }
    };

    f32x4 acc[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // ---- operand reads.  A K-step takes the pixels 8 g + 2 kq + par of a row (par = 0: the even pixels of an 8-pixel group g,
    // par = 1: the odd ones), so that one ds_read_b64 of the pair (8 g + 2 kq, + 1) serves lane (r16, kq) in BOTH K-steps of
    // the group: per group and lane 1 read of dy (+ 1 of the mask) and, per filter row, the three pairs Q0..Q2 at LDS
    // columns 8 g + 2 kq + 2 .. + 7 (tile column 0 = gx x0 - 1 sits at LDS column 3):
    //     par 0: taps kx = 0, 1, 2 = Q0.y, Q1.x, Q1.y        par 1: Q1.x, Q1.y, Q2.x
    // -- 10 (11) 8-byte reads per 18 MFMAs where 4-byte reads took 20 (22).  Banks: ds_read_b64 serves lanes {0-31} = 16
    // channels x kq {0, 1} in one cycle when their 32 pairs fall on 64 distinct banks: the two kq are neighbours (4
    // consecutive floats) and the channel strides are 4 mod 32 floats, so they do; a ds_read_b32 (32 banks) of 16 channels
    // at such a stride was 2-way conflicting whatever the pixel mapping (SQ_LDS_BANK_CONFLICT = half of the LDS cycles).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int NGRP = C::KSTEPS / 2;
    const int xoff = (ib * 16 + r16) * XCH + 2 * kq + 2;
    const int doff = (cb * 16 + r16) * DCH + 2 * kq;
    auto compute = [&](int buf) {
        const float* xb = xs_of(buf) + xoff;
        const float* ab = ds_of(buf) + doff;
        const float* mb = ms_of(buf) + doff;
        // (inline asm: hipcc fuses 8-byte reads off one base into ds_read2_b64, which is banked like two 4-byte reads and
        // takes 8 LDS cycles; the reads are therefore outside the compiler's lgkmcnt tracking and lds_wait() closes them)
#define WG_LD2(v, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off))
        auto lds_wait = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
        const unsigned xb_a = (unsigned)(size_t)(wg_lds_ptr_t)xb, ab_a = (unsigned)(size_t)(wg_lds_ptr_t)ab, mb_a = (unsigned)(size_t)(wg_lds_ptr_t)mb;
        auto rd = [&](int g, f32x2& av, f32x2& mv, f32x2* q) {
            const unsigned aa = ab_a + 32 * g;  // rows are 32 pixels = 4 groups: 8 g is the pixel offset in the tile
            WG_LD2(av, aa, 0);
            if (MASK) {
                const unsigned ma = mb_a + 32 * g;
                WG_LD2(mv, ma, 0);
            }
            const unsigned pa = xb_a + ((g >> 2) * TCOLS + (g & 3) * 8) * 4;
            WG_LD2(q[0], pa, 0);
            WG_LD2(q[1], pa, 8);
            WG_LD2(q[2], pa, 16);
            WG_LD2(q[3], pa, TCOLS * 4);
            WG_LD2(q[4], pa, TCOLS * 4 + 8);
            WG_LD2(q[5], pa, TCOLS * 4 + 16);
            WG_LD2(q[6], pa, TCOLS * 8);
            WG_LD2(q[7], pa, TCOLS * 8 + 8);
            WG_LD2(q[8], pa, TCOLS * 8 + 16);
        };
        f32x2 a_cur, m_cur = f32x2{1.f, 1.f}, q_cur[KK], a_nxt, m_nxt = f32x2{1.f, 1.f}, q_nxt[KK];
        int g = rp;
        rd(g, a_cur, m_cur, q_cur);
        lds_wait();
        if (MASK) {
            a_cur.x = m_cur.x > 0.f ? a_cur.x : 0.f;
            a_cur.y = m_cur.y > 0.f ? a_cur.y : 0.f;
        }
        // consume the first operands here: with reads still pending at the loop header hipcc waits for ALL LDS
        // reads (also the prefetch of the next group) in front of the second MFMA of every iteration
        asm volatile("" : "+v"(a_cur));
#pragma unroll
        for (int t = 0; t < KK; ++t) asm volatile("" : "+v"(q_cur[t]));
        // one group: queue the LDS reads of the following group into (a_n, q_n), then the 18 MFMAs of (a_c, q_c)
        auto step = [&](int gn, const f32x2& a_c, const f32x2* q_c, f32x2& a_n, f32x2& m_n, f32x2* q_n) {
            rd(gn, a_n, m_n, q_n);
            __builtin_amdgcn_sched_barrier(0);      // nothing moves across: reads of the next group, THEN the MFMAs of this one
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.x, q_c[ky * 3 + 0].y, acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.x, q_c[ky * 3 + 1].x, acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.x, q_c[ky * 3 + 1].y, acc[ky * 3 + 2], 0, 0, 0);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.y, q_c[ky * 3 + 1].x, acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.y, q_c[ky * 3 + 1].y, acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.y, q_c[ky * 3 + 2].x, acc[ky * 3 + 2], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            lds_wait();
            if (MASK) {      // VALU work on the fresh reads goes behind the MFMAs (their data has landed by then)
                a_n.x = m_n.x > 0.f ? a_n.x : 0.f;
                a_n.y = m_n.y > 0.f ? a_n.y : 0.f;
            }
            bsum += a_c.x + a_c.y;
            __builtin_amdgcn_sched_barrier(0);
        };
        // a wave runs NGRP / rpN groups (an even number): two per iteration, the register sets swapping roles
        // (the last group re-reads itself: no branch in the loop body)
#pragma unroll 1
        for (; g < NGRP; g += 2 * rpN) {
            step(g + rpN, a_cur, q_cur, a_nxt, m_nxt, q_nxt);
            step(g + 2 * rpN < NGRP ? g + 2 * rpN : g + rpN, a_nxt, q_nxt, a_cur, m_cur, q_cur);
        }
#undef WG_LD2
    };

    // The load cursor walks tiles split, split + nsplit, ...: its (x, y, image) coordinates advance by a fixed
    // mixed-radix step with carries -- scalar adds and compares; a division per tile is ~20 vector-ALU
    // instructions each (v_rcp based), which the MFMA stream of the other wave stretches to ~1 slot apiece.
    const int step_x = a.nsplit % a.tiles_x, step_q = a.nsplit / a.tiles_x;
    const int step_y = step_q % a.tiles_y, step_b = step_q / a.tiles_y;
    WgTile nt = decode(split);          // tile coordinates in tile units until handed to issue()
    nt.x0 /= TW;
    nt.y0 /= TH;
    auto next_tile = [&]() {
        WgTile t = nt;
        t.x0 *= TW;
        t.y0 *= TH;
        nt.x0 += step_x;
        const int cx = nt.x0 >= a.tiles_x ? 1 : 0;
        nt.x0 -= cx * a.tiles_x;
        nt.y0 += step_y + cx;
        const int cy = nt.y0 >= a.tiles_y ? 1 : 0;
        nt.y0 -= cy * a.tiles_y;
        nt.b += step_b + cy;
        return t;
    };
    int tile = split, buf = 0;
    if (tile < a.ntiles) issue(next_tile(), 0);
#ifdef YNET_WG_PROFILE
    unsigned long long pw = 0, pb = 0, pi = 0, pc = 0;
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    for (; tile < a.ntiles; tile += a.nsplit) {
#ifdef YNET_WG_PROFILE
        const unsigned long long p0 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef YNET_WG_PROFILE
        const unsigned long long p1 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef YNET_WG_PROFILE
        const unsigned long long p2 = __builtin_amdgcn_s_memtime();
#endif
        if (tile + a.nsplit < a.ntiles) issue(next_tile(), buf ^ 1);
#ifdef YNET_WG_PROFILE
        const unsigned long long p3 = __builtin_amdgcn_s_memtime();
#endif
        compute(buf);
#ifdef YNET_WG_PROFILE
        const unsigned long long p4 = __builtin_amdgcn_s_memtime();
        pw += p1 - p0; pb += p2 - p1; pi += p3 - p2; pc += p4 - p3;
#endif
        buf ^= 1;
    }
#ifdef YNET_WG_PROFILE
    if (lane == 0) {
        unsigned long long* pr = a.prof;
        atomicAdd(pr + 0, __builtin_amdgcn_s_memtime() - pt0);
        atomicAdd(pr + 1, pw); atomicAdd(pr + 2, pb); atomicAdd(pr + 3, pi); atomicAdd(pr + 4, pc); atomicAdd(pr + 5, 1ull);
        atomicAdd(pr + 6, prof_dma);
    }
#endif

    // ---- K-step residues of a partial block: waves rp = 1.. hand their sums to wave rp = 0 (fixed order)
    if (rpN > 1) {
        float* red = smem;
        __syncthreads();
        if (rp > 0) {
            float* w = red + (wave - cbN * ibN) * 37 * 64 + lane;
#pragma unroll
            for (int t = 0; t < KK; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) w[(t * 4 + e) * 64] = acc[t][e];
            w[36 * 64] = bsum;
        }
        __syncthreads();
        if (rp == 0) {
            for (int q = 1; q < rpN; ++q) {
                const float* w = red + (cb + cbN * (ib + ibN * q) - cbN * ibN) * 37 * 64 + lane;
#pragma unroll
                for (int t = 0; t < KK; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] += w[(t * 4 + e) * 64];
                bsum += w[36 * 64];
            }
        }
    }
    if (rp == 0) {
        float* pw = a.partial_w + (long long)split * a.cout * a.cin * KK;
        const int ci = ib * 16 + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = cb * 16 + 4 * kq + e;
            if (co < nco && ci < ncib) {
#pragma unroll
                for (int t = 0; t < KK; ++t) pw[((long long)(co0 + co) * a.cin + ci0 + ci) * KK + t] = acc[t][e];
            }
        }
        if (want_bias && ib == 0) {          // lane (r16, kq) summed pixels = kq mod 4 of channel cb*16 + r16
            float v = bsum;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int co = cb * 16 + r16;
            if (kq == 0 && co < nco) a.partial_b[(long long)split * a.cout + co0 + co] = v;
        }
    }
}

// ================================================================================================
// Rolling-row generation of the 3x3 filter gradient (maps of at least a few dozen rows, W % 32 == 0, H % 2 == 0).
// wgrad_dma_kernel is bound by its DMA instruction stream (a buffer_load ... lds costs the issuing wave 50-130 cycles; 9
// (12 masked) of them per 144 MFMAs), and two thirds of those fetch the x tile: 4 rows for 2 rows of pixels.  Here a
// workgroup walks DOWN a 32-column strip of one image, two rows per step, and keeps the x rows it has in LDS: a ring of
// three row PAIRS (pair m = image rows 2m - 1, 2m; the tile of rows 2j, 2j + 1 needs pairs j and j + 1, pair j + 2 arrives
// meanwhile) -- one pair = 3 DMA instructions per thread and step instead of 6.  Layout of a pair: [row][channel][44 floats]
// (10 data quads + 1 pad quad), so that a pair is ONE contiguous LDS range (the DMA writes linearly in the lane index) and
// the channel stride is 44 = 12 mod 32 floats: the 16 channels x 4 consecutive floats of an 8-byte operand read fall on 64
// distinct banks.  Strips are cut into segments of `seg` steps (work items for the persistent workgroups); a segment
// starts with two pairs.  Operand mapping, accumulators, reduction and output as in wgrad_dma_kernel.
// ================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (inline asm: hipcc fuses 8-byte reads off one base into ds_read2_b64, which is banked like two 4-byte reads and takes 8 LDS
// cycles; these reads are outside the compiler's lgkmcnt tracking and an explicit s_waitcnt lgkmcnt(0) closes them)
template <int OFF>
__device__ __forceinline__ void wg_ld2(f32x2& v, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <bool MASK>
struct WgRollCfg {
    static constexpr int TW = 32, XCS = 44, XRS = 32 * XCS, XPS = 2 * XRS, XRING = 3 * XPS;
    static constexpr int PQ = 2 * 32 * 11;                    // quads of a pair image
    static constexpr int XI = (PQ + 255) / 256;               // = 3 (the third by 192 threads)
    static constexpr int DDATA = 2 * TW / 4, DQ = (DDATA + 1 + 6) / 8 * 8 + 1, DCH = DQ * 4;
    static constexpr int DS = 32 * DCH + 128, DBUF = DS * (MASK ? 2 : 1);
    static constexpr int DI = (32 * DQ + 255) / 256;
    static constexpr int LDS_FLOATS = XRING + 2 * DBUF, LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int NGRP = 8;
    static_assert(DCH % 32 == 4 && (XCS * 4) % 16 == 0, "operand strides");
    static_assert(3 * 37 * 64 <= LDS_FLOATS, "cross-wave reduction scratch fits");
};

template <bool MASK>
__global__ __launch_bounds__(256, MASK ? 2 : 3) void wgrad_roll_kernel(const WgradArgs a) {
    using C = WgRollCfg<MASK>;
    constexpr int TW = C::TW, XCS = C::XCS, XRS = C::XRS, XPS = C::XPS, DQ = C::DQ, DCH = C::DCH, XI = C::XI, DI = C::DI, KK = 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto ds_of = [&](int b) { return smem + C::XRING + b * C::DBUF; };
    auto ms_of = [&](int b) { return smem + C::XRING + b * C::DBUF + C::DS; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    int bid = blockIdx.x;
    const int cib = bid % a.ci_blks;
    bid /= a.ci_blks;
    const int cob = bid % a.co_blks;
    const int split = bid / a.co_blks;
    const int H = a.H, W = a.W;
    const int HW = __builtin_amdgcn_readfirstlane(H * W);
    const unsigned plane_bytes = (unsigned)HW * 4u;
    const int ci0 = cib * 32, co0 = cob * 32;
    const int ncib = min(32, a.cin - ci0), nco = min(32, a.cout - co0);
    const bool want_bias = (a.partial_b != nullptr) && cib == 0;
    const int cbN = nco > 16 ? 2 : 1, ibN = ncib > 16 ? 2 : 1, rpN = 4 / (cbN * ibN);
    const int cb = wave % cbN, ib = (wave / cbN) % ibN, rp = wave / (cbN * ibN);

    // ---- x pair DMA: LDS quad q = tid + 256 k  ->  (row q / 352, channel (q % 352) / 11, quad q % 11)
    int xsrc[XI], xrow[XI], xcol[XI];   // source of the quad's channel (-1: pad quad / no channel), pair row, first column relative to x0
    unsigned xcoff[XI];                 // byte offset of the channel's plane inside that source's image
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int q = tid + k * 256;
        const int row = q / 352, within = q - row * 352, ch = within / 11, u = within - ch * 11;
        xrow[k] = row;
        xcol[k] = 4 * u - 4;
        int c = ci0 + ch, sid = -1;
        unsigned off = 0;
        if (q < C::PQ && u < 10 && ch < ncib) {
#pragma unroll
            for (int s = 0; s < YNET_MAX_SRC; ++s) {
                if (sid < 0 && s < a.nsrc) {
                    if (c < a.src[s].c) {
                        sid = s;
                        off = (unsigned)c * plane_bytes;
                    } else {
                        c -= a.src[s].c;
                    }
                }
            }
        }
        xsrc[k] = sid;
        xcoff[k] = off;
    }
    // static per-lane offsets relative to (first row of the pair, x0 - 4): interior / first / last / only tile column
    unsigned xs_in[XI], xs_l[XI], xs_r[XI], xs_lr[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const bool ok = xsrc[k] >= 0;
        const unsigned in = ok ? xcoff[k] + (unsigned)(xrow[k] * W + xcol[k] + 4) * 4u : 0x80000000u;
        const bool lh = xcol[k] < 0, rh = xcol[k] >= TW;
        xs_in[k] = in;
        xs_l[k] = lh ? 0x80000000u : in;
        xs_r[k] = rh ? 0x80000000u : in;
        xs_lr[k] = (lh || rh) ? 0x80000000u : in;
    }
    unsigned dstat[DI];                 // dy / mask tile: as in wgrad_dma_kernel (2 rows x 32 pixels per channel, stride DCH)
#pragma unroll
    for (int k = 0; k < DI; ++k) {
        const int q = tid + k * 256;
        const int ch = q / DQ, within = q - ch * DQ;
        const bool ok = q < 32 * DQ && within < C::DDATA;
        dstat[k] = ok ? (unsigned)ch * plane_bytes + (unsigned)((within / (TW / 4)) * W + (within % (TW / 4)) * 4) * 4u : 0x80000000u;
    }

    typedef const __attribute__((address_space(4))) WgradArgs* wg_kargs_t;
    // rows 2 m - 1, 2 m of image b, columns x0 - 4 .. x0 + 35 -> ring slot `slot`
    auto issue_pair = [&](int b, int x0, int m, int slot) {
        wg_kargs_t ka = (wg_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        float* xs = smem + slot * XPS;
        const int y0 = 2 * m - 1;
        const bool at_left = x0 == 0, at_right = x0 + TW >= W;
        unsigned xo[XI], xso;
        if (y0 >= 0 && y0 + 1 < H && y0 * W + x0 - 4 >= 0) {          // both rows inside the image: static offsets + a scalar offset
            xso = (unsigned)(y0 * W + x0 - 4) * 4u;
            if (!at_left && !at_right) {
#pragma unroll
                for (int k = 0; k < XI; ++k) xo[k] = xs_in[k];
            } else {
#pragma unroll
                for (int k = 0; k < XI; ++k) {
                    const unsigned l = at_right ? xs_lr[k] : xs_l[k];
                    xo[k] = at_left ? l : xs_r[k];
                }
            }
        } else {                                                       // first / last pair of the image (a row outside)
            xso = 0u;
#pragma unroll
            for (int k = 0; k < XI; ++k) {
                const int gy = y0 + xrow[k], gx = x0 + xcol[k];
                const bool ok = xsrc[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                xo[k] = ok ? xcoff[k] + (unsigned)(gy * W + gx) * 4u : 0x80000000u;
            }
        }
        const int nsrc = ka->nsrc;
        if (nsrc == 1) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ka->src[0].p + (long long)b * ka->src[0].bs), 0, (unsigned)ka->src[0].c * plane_bytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < XI - 1; ++k) wg_dma16s(r, xs + (k * 256 + wave * 64) * 4, xo[k], xso);
            if (wave < 3) wg_dma16s(r, xs + ((XI - 1) * 256 + wave * 64) * 4, xo[XI - 1], xso);      // quads 512 .. 703
        } else {
#pragma unroll 1
            for (int s = 0; s < nsrc; ++s) {
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(ka->src[s].p + (long long)b * ka->src[s].bs), 0, (unsigned)ka->src[s].c * plane_bytes, 0x00020000);
#pragma unroll
                for (int k = 0; k < XI; ++k)
                    if (xsrc[k] == s) wg_dma16s(r, xs + (k * 256 + wave * 64) * 4, xo[k], xso);
            }
        }
    };
    // rows 2 j, 2 j + 1 of dy (and of the activation) -> buffer `buf`
    auto issue_dy = [&](int b, int x0, int j, int buf) {
        wg_kargs_t ka = (wg_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        constexpr int DFULL = 32 * DQ / 256;
        static_assert(32 * DQ - DFULL * 256 <= 64, "the partial instruction fits one wave");
        const unsigned dso = (unsigned)(2 * j * W + x0) * 4u;
        {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ka->dy + (long long)b * ka->dy_bs + (long long)co0 * HW), 0, (unsigned)nco * plane_bytes, 0x00020000);
            float* ds = ds_of(buf);
#pragma unroll
            for (int k = 0; k < DFULL; ++k) wg_dma16s(r, ds + (k * 256 + wave * 64) * 4, dstat[k], dso);
            if (DI > DFULL && wave == 0) wg_dma16s(r, ds + (DFULL * 256) * 4, dstat[DI - 1], dso);
        }
        if (MASK) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ka->mask + (long long)b * ka->mask_bs + (long long)co0 * HW), 0, (unsigned)nco * plane_bytes, 0x00020000);
            float* ms = ms_of(buf);
#pragma unroll
            for (int k = 0; k < DFULL; ++k) wg_dma16s(r, ms + (k * 256 + wave * 64) * 4, dstat[k], dso);
            if (DI > DFULL && wave == 0) wg_dma16s(r, ms + (DFULL * 256) * 4, dstat[DI - 1], dso);
        }
    };

    f32x4 acc[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The K-step groups of a tile are unrolled with every LDS offset an immediate (group column, filter row, pair Q0..Q2): the
    // loop then holds 18 MFMAs + 10 (11) reads + ONE vector-ALU instruction (the packed bias sum) per group.  A vector-ALU
    // instruction of a wave whose SIMD neighbour streams MFMAs takes 40-70 cycles (tools/valu_under_mfma.hip); the rolled
    // loop's 9 per group (addresses of a run-time group index, scalar-style sums) held it at 0.73 of the MFMA rate with no
    // DMA in flight at all (YNET_WGRAD_DEBUG=1).  Per tile: 4 row addresses + 1 dy address.
    // A wave takes the groups rp, rp + RPN, ...: rp only shifts the column (32 bytes), which goes into the address registers.
    const unsigned x_lane = (unsigned)(size_t)(wg_lds_ptr_t)(smem + (ib * 16 + r16) * XCS + 2 * kq + 2) + 32u * rp;
    const unsigned d_lane = (unsigned)(size_t)(wg_lds_ptr_t)(smem + C::XRING + (cb * 16 + r16) * DCH + 2 * kq) + 32u * rp;
    f32x2 bsum2 = f32x2{0.f, 0.f};
    // tile of rows 2 j, 2 j + 1: x rows 2 j - 1 .. 2 j + 2 = rows 0, 1 of pair slot sa and of pair slot sb (byte offsets)
    auto compute = [&](auto rpn_tag, int buf, unsigned sa, unsigned sb) {
        constexpr int RPN = decltype(rpn_tag)::value, NG = C::NGRP / RPN;
        const unsigned ab = d_lane + (unsigned)(buf * C::DBUF * 4), mb = ab + C::DS * 4;
        unsigned pa[4];
        pa[0] = x_lane + sa;
        pa[1] = x_lane + sa + XRS * 4;
        pa[2] = x_lane + sb;
        pa[3] = x_lane + sb + XRS * 4;
        f32x2 av[2], mv[2], q[2][KK];
        mv[0] = mv[1] = f32x2{1.f, 1.f};
        auto rd = [&](auto g_tag, f32x2& a_, f32x2& m_, f32x2* q_) {
            constexpr int G = decltype(g_tag)::value, r = G >> 2, col = (G & 3) * 32;
            wg_ld2<32 * G>(a_, ab);
            if (MASK) wg_ld2<32 * G>(m_, mb);
            wg_ld2<col>(q_[0], pa[r]);
            wg_ld2<col + 8>(q_[1], pa[r]);
            wg_ld2<col + 16>(q_[2], pa[r]);
            wg_ld2<col>(q_[3], pa[r + 1]);
            wg_ld2<col + 8>(q_[4], pa[r + 1]);
            wg_ld2<col + 16>(q_[5], pa[r + 1]);
            wg_ld2<col>(q_[6], pa[r + 2]);
            wg_ld2<col + 8>(q_[7], pa[r + 2]);
            wg_ld2<col + 16>(q_[8], pa[r + 2]);
        };
        auto lds_wait = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
        auto select = [&](f32x2& a_, const f32x2& m_) {
            if (MASK) {
                a_.x = m_.x > 0.f ? a_.x : 0.f;
                a_.y = m_.y > 0.f ? a_.y : 0.f;
            }
        };
        auto mfmas = [&](const f32x2& a_c, const f32x2* q_c) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.x, q_c[ky * 3 + 0].y, acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.x, q_c[ky * 3 + 1].x, acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.x, q_c[ky * 3 + 1].y, acc[ky * 3 + 2], 0, 0, 0);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.y, q_c[ky * 3 + 1].x, acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.y, q_c[ky * 3 + 1].y, acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c.y, q_c[ky * 3 + 2].x, acc[ky * 3 + 2], 0, 0, 0);
            }
        };
        rd(std::integral_constant<int, 0>{}, av[0], mv[0], q[0]);
        lds_wait();
        select(av[0], mv[0]);
        // group i: queue the reads of group i + 1 into the other register set, then the 18 MFMAs of group i
        auto groups = [&](auto self, auto i_tag) -> void {
            constexpr int I = decltype(i_tag)::value, cur = I & 1, nxt = cur ^ 1;
#ifdef YNET_WG_EXPERIMENT      // (development: YNET_WGRAD_DEBUG & 4 = the MFMA loop without its LDS reads, stale operands)
            if constexpr (I + 1 < NG) if (!(a.debug & 4)) rd(std::integral_constant<int, (I + 1) * RPN>{}, av[nxt], mv[nxt], q[nxt]);
#else
            if constexpr (I + 1 < NG) rd(std::integral_constant<int, (I + 1) * RPN>{}, av[nxt], mv[nxt], q[nxt]);
#endif
            __builtin_amdgcn_sched_barrier(0);      // nothing moves across: reads of the next group, THEN the MFMAs of this one
            mfmas(av[cur], q[cur]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I + 1 < NG) {
                lds_wait();
                select(av[nxt], mv[nxt]);           // VALU work on the fresh reads goes behind the MFMAs
            }
            bsum2 += av[cur];                        // (v_pk_add_f32)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (I + 1 < NG) self(self, std::integral_constant<int, I + 1>{});
        };
        groups(groups, std::integral_constant<int, 0>{});
    };

    // ---- segments split, split + nsplit, ...: (image, strip, run of `seg` two-row steps)
    const int seg = a.seg, nseg_y = a.nseg_y, tiles_x = a.tiles_x, tiles_y = a.tiles_y;
    bool first = true;
    for (int sidx = split; sidx < a.nsegs; sidx += a.nsplit) {
        const int xt = sidx % tiles_x, t2 = sidx / tiles_x;
        const int ys = t2 % nseg_y, b = t2 / nseg_y;
        const int x0 = xt * TW, j0 = ys * seg, j1 = min(j0 + seg, tiles_y);
        if (!first) __syncthreads();       // every wave is done with the previous segment's rows and dy tiles
        first = false;
        issue_pair(b, x0, j0, 0);
        issue_pair(b, x0, j0 + 1, 1);
        issue_dy(b, x0, j0, 0);
        int sj = 0, buf = 0;
        for (int j = j0; j < j1; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int s1 = sj == 2 ? 0 : sj + 1, s2 = s1 == 2 ? 0 : s1 + 1;
            if (j + 1 < j1 && !(a.debug & 1)) {
                issue_pair(b, x0, j + 2, s2);
                issue_dy(b, x0, j + 1, buf ^ 1);
            }
            if (!(a.debug & 2)) {
                const unsigned sa = (unsigned)(sj * XPS * 4), sb = (unsigned)(s1 * XPS * 4);
                if (rpN == 1) compute(std::integral_constant<int, 1>{}, buf, sa, sb);
                else if (rpN == 2) compute(std::integral_constant<int, 2>{}, buf, sa, sb);
                else compute(std::integral_constant<int, 4>{}, buf, sa, sb);
            }
            buf ^= 1;
            sj = s1;
        }
    }

    // ---- K-step residues of a partial block: waves rp = 1.. hand their sums to wave rp = 0 (fixed order)
    float bsum = bsum2.x + bsum2.y;
    if (rpN > 1) {
        float* red = smem;
        __syncthreads();
        if (rp > 0) {
            float* w = red + (wave - cbN * ibN) * 37 * 64 + lane;
#pragma unroll
            for (int t = 0; t < KK; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) w[(t * 4 + e) * 64] = acc[t][e];
            w[36 * 64] = bsum2.x + bsum2.y;
        }
        __syncthreads();
        if (rp == 0) {
            for (int q = 1; q < rpN; ++q) {
                const float* w = red + (cb + cbN * (ib + ibN * q) - cbN * ibN) * 37 * 64 + lane;
#pragma unroll
                for (int t = 0; t < KK; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] += w[(t * 4 + e) * 64];
                bsum += w[36 * 64];
            }
        }
    }
    if (rp == 0) {
        float* pw = a.partial_w + (long long)split * a.cout * a.cin * KK;
        const int ci = ib * 16 + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = cb * 16 + 4 * kq + e;
            if (co < nco && ci < ncib) {
#pragma unroll
                for (int t = 0; t < KK; ++t) pw[((long long)(co0 + co) * a.cin + ci0 + ci) * KK + t] = acc[t][e];
            }
        }
        if (want_bias && ib == 0) {
            float v = bsum;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int co = cb * 16 + r16;
            if (kq == 0 && co < nco) a.partial_b[(long long)split * a.cout + co0 + co] = v;
        }
    }
}

// ================================================================================================
// Filter / bias gradient of the 1x1 predictors (32 -> pred_len; models/ynet.py:469): 0.75 .. 1.9 FLOP per byte, HBM-bound.
// A wave walks 16-pixel chunks: lane (r16, kq) reads ONE 16-byte quad per operand row -- pixels 4 kq .. 4 kq + 3 of the chunk
// for channel r16 (dy) / r16, r16 + 16 (x) -- and element e of the three quads is K-step e of v_mfma_f32_16x16x4_f32 (K-slot kq
// = pixel 4 kq + e for both operands), so nothing goes through LDS and every byte is read once; the next chunk's quads are in
// flight under the 8 (16) MFMAs of this one.  The four waves of a workgroup take consecutive chunks (256 contiguous bytes
// per channel row) and add their tiles through LDS in wave order; workgroup partials -> reduce_partials_kernel.
// (wgrad_mfma_kernel<1>, which stages tiles through LDS like the 5x5 case, ran at 1.9 TB/s.)
// ================================================================================================
struct Wg1x1Args {
    const float* x;
    long long x_bs;
    const float* dy;
    long long dy_bs;
    float* partial_w;      // [gridDim.x][cout * 32]
    float* partial_b;      // [gridDim.x][cout] or NULL
    int B, cout, HW, hw16;
    long long chunks;      // B * HW / 16
};

template <int CT>
__global__ __launch_bounds__(256) void wgrad1x1_stream_kernel(const Wg1x1Args a) {
    __shared__ float red[3][CT * 2 * 4 + CT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
    f32x4 acc[CT][2];
    float bs[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        bs[t] = 0.f;
    }
    const long long stride = (long long)gridDim.x * 4;
    long long c = (long long)blockIdx.x * 4 + wave;
    int b = (int)(c / a.hw16), pc = (int)(c - (long long)b * a.hw16);      // image, chunk inside the image
    const int sb = (int)(stride / a.hw16), sp = (int)(stride - (long long)sb * a.hw16);
    const long long HW = a.HW;
    auto load = [&](int bb, int pp, f32x4& xa, f32x4& xb, f32x4* d) {
        const float* xp = a.x + (long long)bb * a.x_bs + (long long)r16 * HW + pp * 16 + 4 * kq;
        xa = *reinterpret_cast<const f32x4*>(xp);
        xb = *reinterpret_cast<const f32x4*>(xp + 16 * HW);
        const float* dp = a.dy + (long long)bb * a.dy_bs + (long long)r16 * HW + pp * 16 + 4 * kq;
#pragma unroll
        for (int t = 0; t < CT; ++t)
            d[t] = (t * 16 + r16 < a.cout) ? *reinterpret_cast<const f32x4*>(dp + (long long)t * 16 * HW) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 xa, xb, d[CT], nxa, nxb, nd[CT];
    if (c < a.chunks) load(b, pc, xa, xb, d);
    while (c < a.chunks) {
        const long long cn = c + stride;
        int bn = b + sb, pn = pc + sp;
        if (pn >= a.hw16) {
            pn -= a.hw16;
            ++bn;
        }
        if (cn < a.chunks) load(bn, pn, nxa, nxb, nd);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[t][e], xa[e], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[t][e], xb[e], acc[t][1], 0, 0, 0);
                bs[t] += d[t][e];
            }
        xa = nxa;
        xb = nxb;
#pragma unroll
        for (int t = 0; t < CT; ++t) d[t] = nd[t];
        c = cn;
        b = bn;
        pc = pn;
    }
    // waves 1..3 -> wave 0, in order
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave - 1][(t * 2 + h) * 4 + e][lane] = acc[t][h][e];
            red[wave - 1][CT * 8 + t][lane] = bs[t];
        }
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < 3; ++w) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][h][e] += red[w][(t * 2 + h) * 4 + e][lane];
                bs[t] += red[w][CT * 8 + t][lane];
            }
        }
        float* pw = a.partial_w + (long long)blockIdx.x * a.cout * 32;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = t * 16 + 4 * kq + e;      // D[i][j]: row i = 4 kq + e in register e of lane (j = r16, kq)
                if (co < a.cout) {
                    pw[co * 32 + r16] = acc[t][0][e];
                    pw[co * 32 + 16 + r16] = acc[t][1][e];
                }
            }
        if (a.partial_b != nullptr) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                float v = bs[t];                         // lane (r16, kq) summed the pixels 4 kq .. 4 kq + 3 of channel 16 t + r16
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (kq == 0 && t * 16 + r16 < a.cout) a.partial_b[(long long)blockIdx.x * a.cout + t * 16 + r16] = v;
            }
        }
    }
}

// out[i] = sum_s partial[s][i] in a fixed order: thread (o, g) of a block sums the splits s = g mod 8 of output
// base + o (four independent chains, 128-byte coalesced rows), the eight g are then added in order through LDS.
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ partial, float* __restrict__ out, long long n,
                                                     int nsplit, int block, int nblocks) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (long long base = block * 32ll; base < n; base += (long long)nblocks * 32) {
        const long long i = base + o;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (i < n) {
            int s = g;
            for (; s + 24 < nsplit; s += 32) {
                s0 += partial[(long long)s * n + i];
                s1 += partial[(long long)(s + 8) * n + i];
                s2 += partial[(long long)(s + 16) * n + i];
                s3 += partial[(long long)(s + 24) * n + i];
            }
            for (; s < nsplit; s += 8) s0 += partial[(long long)s * n + i];
        }
        red[g][o] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0 && i < n) {
            float t = red[0][o];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += red[q][o];
            out[i] = t;
        }
        __syncthreads();
    }
}

// One launch for both outputs of a wgrad: blocks [0, gridDim.x - 1) reduce the filter partials, the last block the bias
// partials (when there are any) -- a second ~5 us launch per trainable conv otherwise.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial_w, float* __restrict__ dw, long long nw,
                                                              const float* __restrict__ partial_b, float* __restrict__ db, long long nb,
                                                              int nsplit) {
    const int wblocks = db ? (int)gridDim.x - 1 : (int)gridDim.x;
    if ((int)blockIdx.x < wblocks) reduce_partials_body(partial_w, dw, nw, nsplit, (int)blockIdx.x, wblocks);
    else reduce_partials_body(partial_b, db, nb, nsplit, 0, 1);
}

static void launch_reduce_partials(const WgradArgs& a, float* dw, float* db, long long nw, hipStream_t st) {
    int grid = (int)((nw + 31) / 32);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid + (db ? 1 : 0)), dim3(256), 0, st, a.partial_w, dw, nw, a.partial_b, db,
                       (long long)a.cout, a.nsplit);
}

template <int KS, bool MASK>
static int launch_wgrad_m(WgradArgs& a, float* dw, float* db, hipStream_t st) {
    using C = WgCfg<KS>;
    static bool attr_dev[YNET_MAX_DEV] = {false};
    bool& attr_set = attr_dev[ynet_device_slot()];
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
    launch_reduce_partials(a, dw, db, nw, st);
    return ynet_check_launch("conv2d_wgrad(reduce)");
}

template <bool MASK, int TH_>
static int launch_wgrad_dma(WgradArgs& a, float* dw, float* db, hipStream_t st) {
    using C = WgDmaCfg<MASK, TH_>;
    static bool attr_dev[YNET_MAX_DEV] = {false};
    bool& attr_set = attr_dev[ynet_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma_kernel<MASK, TH_>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
        if (getenv("YNET_DEBUG_OCC")) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_dma_kernel<MASK, TH_>, 256, C::LDS_BYTES);
            fprintf(stderr, "wgrad_dma_kernel<%d,%d>: %d bytes of LDS, %d workgroups per CU\n", (int)MASK, TH_, C::LDS_BYTES, per_cu);
        }
    }
    const long long nblk = (long long)a.nsplit * a.co_blks * a.ci_blks;
    static const bool timing = getenv("YNET_WGRAD_TIME") != nullptr;     // development aid: per-launch time of the main kernel
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
    }
#ifdef YNET_WG_PROFILE
    static unsigned long long* prof_dev = nullptr;
    if (!prof_dev) (void)hipMalloc(&prof_dev, 64);
    (void)hipMemsetAsync(prof_dev, 0, 64, st);
    a.prof = prof_dev;
#endif
    hipLaunchKernelGGL((wgrad_dma_kernel<MASK, TH_>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, st, a);
#ifdef YNET_WG_PROFILE
    {
        unsigned long long h[8];
        (void)hipMemcpyAsync(h, prof_dev, 64, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        const double tot = (double)h[0];
        fprintf(stderr, "wgrad_dma<%d> waves %llu avg cycles %.0f: vmcnt-wait %.1f%% barrier %.1f%% issue %.1f%% (of which DMA + descriptors %.1f%%) compute %.1f%% rest %.1f%%\n", (int)MASK,
                h[5], tot / (double)h[5], 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[6] / tot, 100.0 * h[4] / tot,
                100.0 * (tot - h[1] - h[2] - h[3] - h[4]) / tot);
    }
#endif
    if (timing) {
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "wgrad_dma_kernel<%d> B %d %dx%d cin %d cout %d nsplit %d tiles %d blocks %lld: %.1f us  %.1f TFLOP/s\n", (int)MASK, a.B, a.H, a.W,
                a.cin, a.cout, a.nsplit, a.ntiles, nblk, ms * 1e3f, 2.0 * a.B * a.H * a.W * (double)a.cin * a.cout * 9 / (ms * 1e9));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    int rc = ynet_check_launch("conv2d_wgrad");
    if (rc) return rc;
    const long long nw = (long long)a.cout * a.cin * 9;
    launch_reduce_partials(a, dw, db, nw, st);
    return ynet_check_launch("conv2d_wgrad(reduce)");
}

// 1x1, one 32-channel source, <= 32 output channels, no mask, whole 16-pixel chunks, 16-byte aligned planes
static const int WG1X1_BLOCKS = 512;
static bool wgrad1x1_stream_ok(const WgradArgs& a, int K) {
    static const int on = getenv("YNET_WGRAD_1X1_STREAM") ? atoi(getenv("YNET_WGRAD_1X1_STREAM")) : 1;
    return on && K == 1 && a.nsrc == 1 && a.cin == 32 && a.cout <= 32 && a.mask == nullptr && ((long long)a.H * a.W) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(a.src[0].p) & 15) == 0 && (a.src[0].bs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0 && (a.dy_bs & 3) == 0;
}

static int launch_wgrad1x1_stream(WgradArgs& a, float* dw, float* db, float* workspace, hipStream_t st) {
    Wg1x1Args k{};
    k.x = a.src[0].p;
    k.x_bs = a.src[0].bs;
    k.dy = a.dy;
    k.dy_bs = a.dy_bs;
    k.B = a.B;
    k.cout = a.cout;
    k.HW = a.H * a.W;
    k.hw16 = k.HW / 16;
    k.chunks = (long long)a.B * k.hw16;
    int grid = WG1X1_BLOCKS;
    if ((long long)grid * 4 > k.chunks) grid = (int)((k.chunks + 3) / 4);
    a.nsplit = grid;
    a.partial_w = k.partial_w = workspace;
    a.partial_b = k.partial_b = db ? workspace + (long long)grid * a.cout * 32 : nullptr;
    if (a.cout <= 16) hipLaunchKernelGGL(wgrad1x1_stream_kernel<1>, dim3(grid), dim3(256), 0, st, k);
    else hipLaunchKernelGGL(wgrad1x1_stream_kernel<2>, dim3(grid), dim3(256), 0, st, k);
    int rc = ynet_check_launch("conv2d_wgrad(1x1)");
    if (rc) return rc;
    launch_reduce_partials(a, dw, db, (long long)a.cout * 32, st);
    return ynet_check_launch("conv2d_wgrad(reduce)");
}

template <bool MASK>
static int launch_wgrad_roll(WgradArgs& a, float* dw, float* db, hipStream_t st) {
    using C = WgRollCfg<MASK>;
    static bool attr_dev[YNET_MAX_DEV] = {false};
    bool& attr_set = attr_dev[ynet_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_roll_kernel<MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
        if (getenv("YNET_DEBUG_OCC")) {
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_roll_kernel<MASK>, 256, C::LDS_BYTES);
            fprintf(stderr, "wgrad_roll_kernel<%d>: %d bytes of LDS, %d workgroups per CU\n", (int)MASK, C::LDS_BYTES, per_cu);
        }
    }
    const long long nblk = (long long)a.nsplit * a.co_blks * a.ci_blks;
    static const bool timing = getenv("YNET_WGRAD_TIME") != nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
    }
    hipLaunchKernelGGL((wgrad_roll_kernel<MASK>), dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, st, a);
    if (timing) {
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "wgrad_roll_kernel<%d> B %d %dx%d cin %d cout %d nsplit %d segments %d x %d steps, blocks %lld: %.1f us  %.1f TFLOP/s\n", (int)MASK, a.B,
                a.H, a.W, a.cin, a.cout, a.nsplit, a.nsegs, a.seg, nblk, ms * 1e3f, 2.0 * a.B * a.H * a.W * (double)a.cin * a.cout * 9 / (ms * 1e9));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    int rc = ynet_check_launch("conv2d_wgrad");
    if (rc) return rc;
    launch_reduce_partials(a, dw, db, (long long)a.cout * a.cin * 9, st);
    return ynet_check_launch("conv2d_wgrad(reduce)");
}

// Rolling-row kernel: W % 32 == 0, H even, and enough strip segments of >= 4 two-row steps for the persistent workgroups
// (16 / 8 / 4 steps: the longest that still gives three segments per workgroup; a segment start costs one extra row pair)
static bool wgrad_roll_plan(WgradArgs& a) {
    static const int on = getenv("YNET_WGRAD_ROLL") ? atoi(getenv("YNET_WGRAD_ROLL")) : 1;
    static const int forced = getenv("YNET_WGRAD_SEG") ? atoi(getenv("YNET_WGRAD_SEG")) : 0;
    if (!on || (a.W % 32) != 0 || (a.H & 1) != 0) return false;
    const int tiles_y = a.H / 2, strips = a.B * (a.W / 32);
    // without the mask tile three workgroups fit a CU (51 KB each): with two, 27 % of the MFMA slots stay empty even when no DMA
    // is issued at all (YNET_WGRAD_DEBUG=1) -- the waves of a workgroup meet at a barrier every 144 MFMAs
    static const int wg3 = getenv("YNET_WGRAD_WG3") ? atoi(getenv("YNET_WGRAD_WG3")) : 1;
    if (a.mask == nullptr && wg3) {
        const int blocks = a.co_blks * ceil_div(a.cin, 32);
        int n = 768 / blocks;
        if (n < 1) n = 1;
        if (n > a.ntiles) n = a.ntiles;
        a.nsplit = n;
    }
    int seg = 0;
    for (int s : {16, 8, 4})
        if (seg == 0 && (long long)strips * ceil_div(tiles_y, s) >= 3ll * a.nsplit) seg = s;
    if (seg == 0 && (long long)strips * ceil_div(tiles_y, 4) >= a.nsplit) seg = 4;
    if (forced > 0) seg = forced;
    if (seg == 0 || tiles_y < 2) return false;
    static const int debug = getenv("YNET_WGRAD_DEBUG") ? atoi(getenv("YNET_WGRAD_DEBUG")) : 0;
    a.debug = debug;
    a.seg = seg;
    a.nseg_y = ceil_div(tiles_y, seg);
    a.nsegs = strips * a.nseg_y;
    if (a.nsplit > a.nsegs) a.nsplit = a.nsegs;
    return true;
}

template <int KS>
static int launch_wgrad(WgradArgs& a, float* dw, float* db, hipStream_t st) {
    a.ci_blks = ceil_div(a.cin, WgCfg<KS>::CIB);
    return a.mask ? launch_wgrad_m<KS, true>(a, dw, db, st) : launch_wgrad_m<KS, false>(a, dw, db, st);
}

static int wgrad_cib(int K) { return K == 5 ? 6 : 32; }

// the 3x3 DMA path (2-row tiles) needs W % 4 == 0; alignment of the planes is checked per call
static bool wgrad_dma_shape(int W, int K) {
    static const int on = getenv("YNET_WGRAD_DMA") ? atoi(getenv("YNET_WGRAD_DMA")) : 1;
    return on && K == 3 && (W % 4) == 0;
}

static int wgrad_plan(int B, int H, int W, int cout, int cin, int K, int th, int* nsplit_out) {
    const int tiles = B * ceil_div(H, th) * ceil_div(W, 32);
    const int blocks_per_split = ceil_div(cout, 32) * ceil_div(cin, wgrad_cib(K));
    const int resident = th == 1 ? 768 : 512;  // 3 (one-row tiles: 45 KB of LDS) or 2 resident workgroups per CU
    int nsplit = resident / blocks_per_split;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > tiles) nsplit = tiles;
    if (nsplit > resident) nsplit = resident;
    *nsplit_out = nsplit;
    return tiles;
}

extern "C" {

// floats of workspace ynet_conv2d_wgrad needs for this problem
long long ynet_conv2d_wgrad_workspace_floats(int B, int H, int W, int cout, int cin, int K) {
    int n1 = 0, n2 = 0, n4 = 0;     // any tile height may be chosen at call time (alignment, env): size for the largest split
    wgrad_plan(B, H, W, cout, cin, K, 1, &n1);
    wgrad_plan(B, H, W, cout, cin, K, 2, &n2);
    wgrad_plan(B, H, W, cout, cin, K, 4, &n4);
    int n = n1 > n2 ? (n1 > n4 ? n1 : n4) : (n2 > n4 ? n2 : n4);
    if (K == 1 && n < WG1X1_BLOCKS) n = WG1X1_BLOCKS;      // (the streaming 1x1 kernel: one partial per workgroup)
    return (long long)n * ((long long)cout * cin * K * K + cout);
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
    bool dma = wgrad_dma_shape(W, K);
    {
        auto misaligned = [](const void* p, long long bs) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0 || (bs & 3) != 0; };
        for (int i = 0; i < nsrc; ++i) dma = dma && !misaligned(src[i], src_bs[i]);
        dma = dma && !misaligned(dy, dy_bs) && (mask == nullptr || !misaligned(mask, mask_bs));
    }
    const int th = dma ? 2 : 4;      // rows per tile (one-row tiles: 3 workgroups per CU, more halo traffic -- measured equal)
    a.tiles_x = ceil_div(W, 32);
    a.tiles_y = ceil_div(H, th);
    a.ntiles = wgrad_plan(B, H, W, cout, a.cin, K, th, &a.nsplit);
    a.co_blks = ceil_div(cout, 32);
    a.partial_w = workspace;
    hipStream_t st = (hipStream_t)stream;
    if (dma) {
        a.ci_blks = ceil_div(a.cin, 32);
        const int nsplit_dma = a.nsplit;
        const bool roll = wgrad_roll_plan(a);      // (may change nsplit: up to the 768 resident workgroups the workspace is sized for)
        if (!roll) a.nsplit = nsplit_dma;
        a.partial_b = db ? workspace + (long long)a.nsplit * cout * a.cin * K * K : nullptr;
        if (roll) return mask ? launch_wgrad_roll<true>(a, dw, db, st) : launch_wgrad_roll<false>(a, dw, db, st);
        return mask ? launch_wgrad_dma<true, 2>(a, dw, db, st) : launch_wgrad_dma<false, 2>(a, dw, db, st);
    }
    a.partial_b = db ? workspace + (long long)a.nsplit * cout * a.cin * K * K : nullptr;
    if (wgrad1x1_stream_ok(a, K)) return launch_wgrad1x1_stream(a, dw, db, workspace, st);
    switch (K) {
        case 1: return launch_wgrad<1>(a, dw, db, st);
        case 3: return launch_wgrad<3>(a, dw, db, st);
        case 5: return launch_wgrad<5>(a, dw, db, st);
        default: ynet_set_error("conv2d_wgrad: kernel size %d not supported (1, 3, 5)", K); return 1;
    }
}

}  // extern "C"
