// Stage A of the gated experiment of VERDICT r3 item 2: a 3x3 convolution whose fp32 operands arrive
// ALREADY SPLIT into three bf16 planes -- written by the PRODUCER's epilogue, so the consumer has no vector-ALU work at all -- with
// the partial products accumulated in fp32 on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate).
//
//   activation format   P[b][cblk = C / 8][plane 3][y][x] of 16-byte units = 8 consecutive channels of one pixel in bf16;
//                       plane 0 / 1 / 2 = hi / mid / lo with x == hi + mid + lo EXACTLY (round-to-nearest splits: 8 + 9 + 9
//                       significant bits >= 24).  6 bytes per element instead of 4 (the 1.5x traffic of the gate).
//   consumer            a workgroup stages its (4 + 2) x (128 + 2) pixel tile of 8 channels = 1 cblk x 3 planes per chunk by LDS-DMA
//                       (16 bytes per lane, one unit each: the LDS image IS the global format, zero padding by the range
//                       check) into a DOUBLE buffer, one chunk ahead of the MFMA loop (2 x 40 KB: two workgroups per CU); every
//                       lane's MFMA fragment is one ds_read_b128.  K = 16 of the instruction = 8 channels x TWO TAPS: the lane
//                       halves read the tile at tap 2p / 2p + 1 (the tenth tap is a zero filter: 10 % idle MFMA work), filters
//                       are split and laid out in fragment order on the host.  XCD-aware tile walk (halo rows re-read from L2).
//   products            NPROD = 9: all nine plane pairs (an exact restatement of the fp32 product, summed in fp32);
//                       NPROD = 6: without mid*lo, lo*mid, lo*lo -- terms below 2^-26 of the product with round-to-nearest splits.
//   epilogue (producer) ReLU, split of the fp32 accumulators into the three planes, 8-byte stores in the same format.
// Gate: >= 140 TFLOP/s fp32-equivalent on 32 -> 32 @ 256^2, B 32 including the epilogue, max error vs fp64 <= the fp32 FMA chain's.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/conv_b3s tools/conv_b3s_proto.hip && /tmp/conv_b3s [B Cin Cout H W]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define TR 4                               // output rows per workgroup (one per wave)
#define TC 128                             // output columns per workgroup (4 MFMA blocks of 32 pixels per wave)
#define LROWS (TR + 2)
#define LCOLS (TC + 2)
#define PLANE_U (LROWS * LCOLS)            // 780 16-byte units per (cblk, plane)
#define PLANE_P 832                        // ... padded to 13 wave instructions of 64 units
#define NIMG 3                             // plane images per 8-channel chunk
#define BUF_BYTES (NIMG * PLANE_P * 16)    // 39,936
#define LDS_BYTES (2 * BUF_BYTES)          // 79,872: double buffer, two workgroups per CU

struct Args {
    const u32x4* xp;       // input planes  P[b][Cin/8][3][H][W]
    const u32x4* wpk;      // [chunk of 8 channels][tap pair 5][plane][cb][64 lanes] filter fragments
    const float* bias;
    u32x4* yp;             // output planes P[b][Cout/8][3][H][W]
    int B, Cin, Cout, H, W, relu, ntiles, diag, skew;      // diag (timing ablations, wrong results): 1 no DMA after the first chunk, 2 no epilogue, 4 no filter loads after the first (the "no MFMAs" ablation, bit 8, was a run-time branch around every MFMA and is gone: it made hipcc shuttle the accumulators between AGPRs and VGPRs)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned lds_byte, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(uintptr_t)lds_byte, 16, voff, soff, 0, 0);
}

// round-to-nearest-even fp32 -> bf16 (as bits in the upper half of a float) and the exact remainder
__device__ __forceinline__ unsigned bf16_rn_bits(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;        // (finite inputs: activations after ReLU)
}

// two fp32 -> packed bf16 pair, round to nearest even (a plain cast: hipcc emits v_cvt_pk_bf16_f32 on gfx950)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

template <int NCB, int NPROD, bool SPREAD>
__global__ __launch_bounds__(256, 2) void conv_b3s_kernel(const Args a) {
    extern __shared__ u32x4 tile[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W;
    const int tiles_x = W / TC, tiles_y = H / TR;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);

    // ---- static DMA geometry of this thread: every plane image is 13 wave instructions of 64 units; this wave issues instructions
    // sub = wave + 4 k (k < 4) of EACH image, so the geometry is per k only: the offset inside an H x W plane RELATIVE to the tile
    // origin (row0 - 1, col0 - 1), and whether the unit is a left / right halo column (zero-filled on the image's edge) or a pad unit
    int rel[4];                            // ((r * W + col) * 16) or -1: never loaded
    unsigned leftbits = 0, rightbits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sub = wave + 4 * k, u = sub * 64 + lane;
        const int r = u / LCOLS, col = u - r * LCOLS;
        const bool ok = sub < 13 && u < PLANE_U;
        rel[k] = ok ? (r * W + col) * 16 : -1;
        if (ok && col == 0) leftbits |= 1u << k;
        if (ok && col == LCOLS - 1) rightbits |= 1u << k;
    }

    f32x16 acc[4][NCB];
    if ((int)blockIdx.x >= a.ntiles) return;
    const int nchunk = a.Cin / 8, ncblk_in = a.Cin / 8;
    const __amdgpu_buffer_rsrc_t rw = rsrc(a.wpk, (unsigned)(nchunk * 5 * 3 * NCB * 64 * 16));
    const unsigned char* lbase = reinterpret_cast<const unsigned char*>(tile);
    // fragment reads: lane (l & 31) = pixel column; the lane half picks the tap of the pair (tap 2p or 2p + 1; the tenth tap has a
    // zero filter and re-reads tap 8)
    unsigned tapoff[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        int t = 2 * p + (lane >> 5);
        t = t > 8 ? 8 : t;
        tapoff[p] = (unsigned)(((t / 3) * LCOLS + (t % 3) + wave * LCOLS + (lane & 31)) * 16);
    }
    // bias of this lane's output channels: four float4 per 32-channel block (channels 8 q + 4 (l >> 5) .. + 3), re-read per tile
    // (L1 hits; 16 registers held across the kernel made it spill)
    const __amdgpu_buffer_rsrc_t rb = rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);

    // XCD-aware walk (workgroups are dealt round-robin over the 8 XCDs): each XCD sweeps its own contiguous eighth of the tiles
    const bool xcd_walk = (gridDim.x & 7) == 0 && a.ntiles >= (int)gridDim.x;
    const int per_xcd = (a.ntiles + 7) >> 3;
    const int gstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tile_first = xcd_walk ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tile_end = xcd_walk ? min(a.ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd) : a.ntiles;
    if (tile_first >= tile_end) return;

    // issue the DMA of chunk c of tile t into buffer `buf`
    auto dma_chunk = [&](int t, int c, int buf) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int row0 = ty * TR, col0 = tx * TC;
        // rows above / below the image fall outside the per-plane descriptor by themselves (the offset wraps / exceeds H * W);
        // the halo COLUMNS of an edge tile would alias the neighbouring row and are sent to the out-of-range marker
        const unsigned edge = (col0 == 0 ? leftbits : 0u) | (col0 + TC == W ? rightbits : 0u);
        const int org = ((row0 - 1) * W + (col0 - 1)) * 16;
        unsigned vo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) vo[k] = (rel[k] < 0 || ((edge >> k) & 1u)) ? 0x80000000u : (unsigned)(rel[k] + org);
#pragma unroll
        for (int pl = 0; pl < NIMG; ++pl) {
            const u32x4* pbase = a.xp + ((long long)(b * ncblk_in + c) * 3 + pl) * HW;
            const __amdgpu_buffer_rsrc_t rp = rsrc(pbase, (unsigned)HW * 16u);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (wave + 4 * k < 13) dma16(rp, lds0 + (unsigned)(buf * BUF_BYTES + (pl * PLANE_P + (wave + 4 * k) * 64) * 16), vo[k], 0u);
        }
    };

    // The epilogue of tile T is spread over the chunks of tile T + 1 (a quarter of the output channels per chunk): its HBM writes
    // then run beside the next tile's MFMAs instead of in a burst of their own in which the matrix cores idle (measured: every
    // phase of the first version was additive -- 6 products: MFMA loop 105 us + DMA 60 + epilogue 54 + filter loads 39 + skeleton 62).
    f32x16 pend[SPREAD ? 4 : 1][SPREAD ? NCB : 1];
    int p_row0 = 0, p_col0 = 0, p_b = -1;
    auto store_group = [&](const f32x16 (&src)[4][NCB], const int cb, const int q) __attribute__((always_inline)) {
        const int ncblk_out = a.Cout / 8;
        const unsigned vo = (unsigned)((((p_row0 + wave) * W + p_col0 + (lane & 31)) * 16) + (lane >> 5) * 8);
        u32x4* obase = a.yp + ((long long)(p_b * ncblk_out + cb * 4 + q) * 3) * HW;
        const __amdgpu_buffer_rsrc_t ry = rsrc(obase, (unsigned)(3 * HW) * 16u);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            u32x2 h2, m2, l2;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                float v0 = src[nb][cb][4 * q + 2 * e2], v1 = src[nb][cb][4 * q + 2 * e2 + 1];      // (q: a constant after unrolling)
                if (a.relu) {
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                }
                const unsigned h = pk_bf16(v0, v1);
                const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
                const unsigned m = pk_bf16(r0, r1);
                const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
                h2[e2] = h;
                m2[e2] = m;
                l2[e2] = pk_bf16(s0, s1);
            }
            __builtin_amdgcn_raw_buffer_store_b64(h2, ry, vo + (unsigned)(nb * 32 * 16), 0u, 0);
            __builtin_amdgcn_raw_buffer_store_b64(m2, ry, vo + (unsigned)(nb * 32 * 16), (unsigned)HW * 16u, 0);
            __builtin_amdgcn_raw_buffer_store_b64(l2, ry, vo + (unsigned)(nb * 32 * 16), (unsigned)HW * 32u, 0);
        }
    };
    // the groups (cb, q) of the pending tile that chunk c of the current tile writes: g = cb * 4 + q with g % nspread == c
    const int ngroups = 4 * NCB, nspread = nchunk < ngroups ? nchunk : ngroups;
    auto store_part = [&](int c) {
        if constexpr (SPREAD) {
            if (p_b < 0 || (a.diag & 2)) return;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((cb * 4 + q) % nspread == c) store_group(pend, cb, q);
        }
    };

    if (a.skew && (blockIdx.x & 256))      // (experiment) every second workgroup starts late: the CU's two workgroups alternate phases
        for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(64);
    int lt = tile_first, lc = 0;           // load cursor, one chunk ahead of the compute cursor
    dma_chunk(lt, lc, 0);
    if (++lc == nchunk) { lc = 0; lt += gstride; }
    int buf = 0;
    bf16x8 wf[2][3][NCB];
    for (int tile_id = tile_first; tile_id < tile_end; tile_id += gstride) {
        const int tx = tile_id % tiles_x, ty = (tile_id / tiles_x) % tiles_y, b = tile_id / (tiles_x * tiles_y);
        const int row0 = ty * TR, col0 = tx * TC;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nb][cb][i] = 0.f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 bq = __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)((cb * 32 + q * 8 + (lane >> 5) * 4) * 4), 0u, 0);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned ub = bq[e];           // (scalar copy first: __builtin_bit_cast of a vector ELEMENT reads element 0)
                        acc[nb][cb][4 * q + e] = __builtin_bit_cast(float, ub);
                    }
            }
        for (int c = 0; c < nchunk; ++c) {
            auto load_w = [&](int p, int wb) {
                if ((a.diag & 4) && (c | p | (tile_id - tile_first))) return;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        wf[wb][pl][cb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                            rw, (unsigned)lane * 16u, (unsigned)((((c * 5 + p) * 3 + pl) * NCB + cb) * 64) * 16u, 0));
            };
            load_w(0, 0);
            // this wave's DMAs of the chunk about to be consumed have landed; after the barrier every wave's have, and every wave is
            // done reading the other buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (lt < tile_end) {
                if (!(a.diag & 1)) dma_chunk(lt, lc, buf ^ 1);
                if (++lc == nchunk) { lc = 0; lt += gstride; }
            }
            load_w(1, 1);
            if (c < nspread) store_part(c);
            __builtin_amdgcn_sched_barrier(0);
            // ---- fragment pipeline: the 12 pixel fragments of a tap pair live in one register set; small products first.
            // NPROD = 6 skips (mid, lo), (lo, mid), (lo, lo): i = filter plane, j = pixel plane.
            const unsigned char* xb = lbase + buf * BUF_BYTES;
            bf16x8 xf[4][3];
            auto read_x = [&](int p, int pl) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    xf[nb][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xb + tapoff[p] + (unsigned)((pl * PLANE_P + nb * 32) * 16)));
            };
            read_x(0, 2); read_x(0, 1); read_x(0, 0);
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                if (p >= 1 && p + 1 < 5) load_w(p + 1, (p + 1) & 1);
#pragma unroll
                for (int j = 2; j >= 0; --j) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 2; i >= 0; --i) {
                        if (NPROD == 6 && i + j > 2) continue;
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb)
                                acc[nb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[p & 1][i][cb], xf[nb][j], acc[nb][cb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 1 < 5) read_x(p + 1, j);
                }
            }
            buf ^= 1;
        }
        // ---- hand the finished tile over: its epilogue (the producer's side of the format: ReLU, exact three-way split by
        // v_cvt_pk_bf16_f32 -- round to nearest even, two elements per instruction --, 8-byte stores: a lane holds channels
        // 8 q + 4 (l >> 5) .. + 3 of pixel column l & 31 in registers 4 q .. 4 q + 3, half a unit per (q, plane)) runs in parts
        // beside the next tile's chunks
        p_row0 = row0;
        p_col0 = col0;
        p_b = b;
        if constexpr (SPREAD) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) pend[nb][cb] = acc[nb][cb];
        } else if (!(a.diag & 2) || acc[0][0][0] == 123.456f) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) store_group(acc, cb, q);
        }
    }
    if constexpr (SPREAD)
        for (int c = 0; c < nspread; ++c) store_part(c);      // the last tile's epilogue
}


// ================================================================================================
// Second form (round 4, after the first one's phases turned out additive): ONE workgroup per CU, filter fragments RESIDENT in
// LDS for the whole kernel (Cin / 8 x 15 KB: the first form re-read them from L2 per tile, 12 TB/s of L2 -> L1 traffic chip-wide),
// (4 + 2) x (64 + 2)-pixel tiles in a RING of three 21 KB chunk buffers filled two chunks ahead (the DMA latency is longer than
// one chunk's MFMA phase), progressive `s_waitcnt vmcnt(n)` instead of vmcnt(0).  RES=1 selects it.
// ================================================================================================
#define TC2 64
#define LCOLS2 (TC2 + 2)
#define PLANE_U2 (LROWS * LCOLS2)          // 396 units per plane image
#define PLANE_P2 448                       // ... padded to 7 wave instructions
#define BUF2_BYTES (3 * PLANE_P2 * 16)     // 21,504 per 8-channel chunk
#define NRING 3

template <int NPROD>
__global__ __launch_bounds__(256, 1) void conv_b3r_kernel(const Args a) {
    extern __shared__ u32x4 tile[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W;
    const int tiles_x = W / TC2, tiles_y = H / TR;
    const int nchunk = a.Cin / 8;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned wbytes = (unsigned)(nchunk * 15 * 64 * 16);       // resident filter fragments [chunk][pair 5][plane 3][64 lanes]
    const unsigned ring0 = lds0 + wbytes;

    // ---- filter fragments -> LDS, once
    {
        const __amdgpu_buffer_rsrc_t rwd = rsrc(a.wpk, wbytes);
        for (int j = wave; j < nchunk * 15; j += 4) dma16(rwd, lds0 + (unsigned)j * 1024u, (unsigned)(j * 1024 + lane * 16), 0u);
    }
    // ---- static DMA geometry: a chunk is 3 planes x 7 wave instructions; this wave issues j = wave + 4 k (k < 6, j < 21)
    int rel[6];
    unsigned leftbits = 0, rightbits = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int j = wave + 4 * k, sub = j % 7, u = sub * 64 + lane;
        const int r = u / LCOLS2, col = u - r * LCOLS2;
        const bool ok = j < 21 && u < PLANE_U2;
        rel[k] = ok ? (r * W + col) * 16 : -1;
        if (ok && col == 0) leftbits |= 1u << k;
        if (ok && col == LCOLS2 - 1) rightbits |= 1u << k;
    }
    unsigned tapoff[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        int t = 2 * p + (lane >> 5);
        t = t > 8 ? 8 : t;
        tapoff[p] = (unsigned)(((t / 3) * LCOLS2 + (t % 3) + wave * LCOLS2 + (lane & 31)) * 16);
    }
    const __amdgpu_buffer_rsrc_t rb = rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
    const bool xcd_walk = (gridDim.x & 7) == 0 && a.ntiles >= (int)gridDim.x;
    const int per_xcd = (a.ntiles + 7) >> 3;
    const int gstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tile_first = xcd_walk ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tile_end = xcd_walk ? min(a.ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd) : a.ntiles;

    auto dma_chunk = [&](int t, int c, int buf) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int row0 = ty * TR, col0 = tx * TC2;
        const unsigned edge = (col0 == 0 ? leftbits : 0u) | (col0 + TC2 == W ? rightbits : 0u);
        const int org = ((row0 - 1) * W + (col0 - 1)) * 16;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int j = wave + 4 * k;
            if (j < 21) {
                const int pl = j / 7, sub = j - pl * 7;
                const u32x4* pbase = a.xp + ((long long)(b * nchunk + c) * 3 + pl) * HW;
                const __amdgpu_buffer_rsrc_t rp = rsrc(pbase, (unsigned)HW * 16u);
                const unsigned vo = (rel[k] < 0 || ((edge >> k) & 1u)) ? 0x80000000u : (unsigned)(rel[k] + org);
                dma16(rp, ring0 + (unsigned)(buf * BUF2_BYTES + (pl * PLANE_P2 + sub * 64) * 16), vo, 0u);
            }
        }
    };
    // load cursor: two chunks ahead of the compute cursor
    int lt = tile_first, lc = 0, lbuf = 0, inflight = 0;
    auto advance_load = [&]() {
        if (lt < tile_end) {
            dma_chunk(lt, lc, lbuf);
            ++inflight;
            lbuf = lbuf + 1 == NRING ? 0 : lbuf + 1;
            if (++lc == nchunk) { lc = 0; lt += gstride; }
        }
    };
    float bias_r[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 bq = __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)((q * 8 + (lane >> 5) * 4) * 4), 0u, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned ub = bq[e];
            bias_r[4 * q + e] = __builtin_bit_cast(float, ub);
        }
    }
    if (tile_first < tile_end) {
        advance_load();
        advance_load();
    }
    const unsigned char* lbase = reinterpret_cast<const unsigned char*>(tile);
    int buf = 0;
    f32x16 acc[2], pend[2];
    int p_row0 = 0, p_col0 = 0, p_b = -1;
    // the producer's side of the format for the PREVIOUS tile: ReLU, three-way split, 8-byte stores.  It is issued right after the
    // barrier of the next tile's first chunk, so that the stores drain under that chunk's MFMAs: the `vmcnt` waits below count
    // loads AND stores, and a store issued just before a wait would be waited for.
    auto flush = [&]() {
        if (p_b < 0 || (a.diag & 2)) return;
        const int ncblk_out = a.Cout / 8;
        const unsigned vo = (unsigned)((((p_row0 + wave) * W + p_col0 + (lane & 31)) * 16) + (lane >> 5) * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4* obase = a.yp + ((long long)(p_b * ncblk_out + q) * 3) * HW;
            const __amdgpu_buffer_rsrc_t ry = rsrc(obase, (unsigned)(3 * HW) * 16u);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                u32x2 h2, m2, l2;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    float v0 = pend[nb][4 * q + 2 * e2], v1 = pend[nb][4 * q + 2 * e2 + 1];
                    if (a.relu) {
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                    }
                    const unsigned h = pk_bf16(v0, v1);
                    const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
                    const unsigned m = pk_bf16(r0, r1);
                    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
                    h2[e2] = h;
                    m2[e2] = m;
                    l2[e2] = pk_bf16(s0, s1);
                }
                __builtin_amdgcn_raw_buffer_store_b64(h2, ry, vo + (unsigned)(nb * 32 * 16), 0u, 0);
                __builtin_amdgcn_raw_buffer_store_b64(m2, ry, vo + (unsigned)(nb * 32 * 16), (unsigned)HW * 16u, 0);
                __builtin_amdgcn_raw_buffer_store_b64(l2, ry, vo + (unsigned)(nb * 32 * 16), (unsigned)HW * 32u, 0);
            }
        }
        p_b = -1;
    };
    for (int tile_id = tile_first; tile_id < tile_end; tile_id += gstride) {
        const int tx = tile_id % tiles_x, ty = (tile_id / tiles_x) % tiles_y, b = tile_id / (tiles_x * tiles_y);
        const int row0 = ty * TR, col0 = tx * TC2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = bias_r[i];
        for (int c = 0; c < nchunk; ++c) {
            // the chunk about to be consumed has landed when at most the NEXT chunk's DMAs of this wave are outstanding: loads return
            // in order among themselves (wave 0 issues 6 per chunk, the others 5); stores are counted too and may retire in any order
            // relative to the loads, which can only make this wait longer, never shorter than needed
            if (inflight < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (wave == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            __syncthreads();
            --inflight;
            advance_load();          // two chunks ahead, into the buffer whose readers all passed the barrier above
            if (c == 0) flush();     // the previous tile's epilogue: its stores drain under this chunk's MFMAs
            const unsigned char* xb = lbase + wbytes + buf * BUF2_BYTES;
            const unsigned char* wb = lbase + (unsigned)(c * 15) * 1024u + lane * 16;
            bf16x8 xf[2][3], wf[2][3];
            auto read_x = [&](int p, int pl) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    xf[nb][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xb + tapoff[p] + (unsigned)((pl * PLANE_P2 + nb * 32) * 16)));
            };
            auto read_w = [&](int p, int wbuf) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[wbuf][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wb + (unsigned)((p * 3 + pl) * 1024)));
            };
            read_w(0, 0);
            read_x(0, 2); read_x(0, 1); read_x(0, 0);
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                if (p + 1 < 5) read_w(p + 1, (p + 1) & 1);
#pragma unroll
                for (int j = 2; j >= 0; --j) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 2; i >= 0; --i) {
                        if (NPROD == 6 && i + j > 2) continue;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[p & 1][i], xf[nb][j], acc[nb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 1 < 5) read_x(p + 1, j);
                }
            }
            buf = buf + 1 == NRING ? 0 : buf + 1;
        }
        pend[0] = acc[0];
        pend[1] = acc[1];
        p_row0 = row0;
        p_col0 = col0;
        p_b = b;
    }
    flush();
}

// fp32 NCHW -> split planes (what the glue producers -- bilinear x2, patch gather, the previous conv -- would write)
__global__ void split_planes_kernel(const float* __restrict__ x, u32x4* __restrict__ p, int B, int C, int HW) {
    const long long n = (long long)B * (C / 8) * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int px = (int)(i % HW);
        const long long bc = i / HW;               // b * (C / 8) + cblk
        unsigned hi[8], mid[8], lo[8];
        for (int j = 0; j < 8; ++j) {
            const float v = x[(bc * 8 + j) * HW + px];
            hi[j] = bf16_rn_bits(v);
            const float r1 = v - __builtin_bit_cast(float, hi[j]);
            mid[j] = bf16_rn_bits(r1);
            lo[j] = bf16_rn_bits(r1 - __builtin_bit_cast(float, mid[j]));
        }
        u32x4 h, m, l;
        for (int j = 0; j < 4; ++j) {
            h[j] = (hi[2 * j] >> 16) | hi[2 * j + 1];
            m[j] = (mid[2 * j] >> 16) | mid[2 * j + 1];
            l[j] = (lo[2 * j] >> 16) | lo[2 * j + 1];
        }
        p[(bc * 3 + 0) * HW + px] = h;
        p[(bc * 3 + 1) * HW + px] = m;
        p[(bc * 3 + 2) * HW + px] = l;
    }
}

// ---------------------------------------------------------------- host
static float bf16_rn(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    float h;
    memcpy(&h, &u, 4);
    return h;
}
static void split3(float x, unsigned short out[3]) {
    for (int p = 0; p < 3; ++p) {
        const float h = bf16_rn(x);
        unsigned u;
        memcpy(&u, &h, 4);
        out[p] = (unsigned short)(u >> 16);
        x -= h;
    }
}
static float bf16_to_f(unsigned short s) {
    const unsigned u = (unsigned)s << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int NCB, int NPROD, bool SPREAD>
static void run(const Args& a, int grid, const char* what, const std::vector<float>& hx, const std::vector<float>& hw, const std::vector<float>& hb) {
    auto kern = conv_b3s_kernel<NCB, NPROD, SPREAD>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int B = a.B, Cin = a.Cin, Cout = a.Cout, H = a.H, W = a.W;
    const size_t nyu = (size_t)B * (Cout / 8) * 3 * H * W;
    CK(hipMemset(a.yp, 0xff, nyu * 16));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, a);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> hy(nyu * 8);
    CK(hipMemcpy(hy.data(), a.yp, nyu * 16, hipMemcpyDeviceToHost));
    double worst = 0, worst32 = 0, ref_max = 0, sum2 = 0, sum2_32 = 0;
    srand(7);
    const int NS = 20000;
    for (int n = 0; n < NS; ++n) {
        int bb = rand() % B, co = rand() % Cout, yy = rand() % H, xx = rand() % W;
        if (n < 2000) { yy = (n & 1) ? H - 1 - (n % 3) : n % 3; xx = ((n & 2) ? W - 1 - (n % 5) : (n % 5) + ((n & 4) ? TC - 2 : 0)) % W; }
        double s = hb[co];
        float s32 = hb[co];
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                const int iy = yy + t / 3 - 1, ix = xx + t % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float xv = hx[(((size_t)bb * Cin + ci) * H + iy) * W + ix], wv = hw[((size_t)co * Cin + ci) * 9 + t];
                s += (double)xv * wv;
                s32 = fmaf(xv, wv, s32);
            }
        if (s < 0) s = 0;
        if (s32 < 0) s32 = 0;
        const size_t u = ((((size_t)bb * (Cout / 8) + co / 8) * 3) * H + yy) * W + xx;
        double got = 0;
        for (int pl = 0; pl < 3; ++pl) got += bf16_to_f(hy[(u + (size_t)pl * H * W) * 8 + (co & 7)]);
        worst = fmax(worst, fabs(got - s));
        worst32 = fmax(worst32, fabs((double)s32 - s));
        sum2 += (got - s) * (got - s);
        sum2_32 += ((double)s32 - s) * ((double)s32 - s);
        ref_max = fmax(ref_max, fabs(s));
    }
    printf("%s  B=%d %d->%d %dx%d  vs fp64: max|err| %.3e (fp32 fma chain %.3e)  rms %.3e (fp32 chain %.3e)  max|y| %.3f\n", what, B, Cin, Cout, H, W,
           worst, worst32, sqrt(sum2 / NS), sqrt(sum2_32 / NS), ref_max);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, a);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 9;
    const double bytes = (double)B * H * W * (Cin + Cout) * 6.0;
    printf("%s  grid %d: %.1f us  %.1f TFLOP/s fp32-equivalent  (%.2f TB/s of split planes in + out)\n", what, grid, ms * 1e3, fl / ms / 1e9, bytes / ms / 1e9);
}

template <int NPROD>
static void run_res(const Args& a, const char* what) {
    auto kern = conv_b3r_kernel<NPROD>;
    const int lds = (a.Cin / 8) * 15 * 1024 + NRING * BUF2_BYTES;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    Args b = a;
    b.ntiles = a.B * (a.H / TR) * (a.W / TC2);
    int grid = 256;
    if (getenv("GRID")) grid = atoi(getenv("GRID"));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, b);
    CK(hipGetLastError());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, b);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = 2.0 * a.B * a.H * a.W * (double)a.Cin * a.Cout * 9;
    printf("%s  resident filters, ring of %d, grid %d, LDS %d: %.1f us  %.1f TFLOP/s fp32-equivalent\n", what, NRING, grid, lds, ms * 1e3, fl / ms / 1e9);
}

int main(int argc, char** argv) {
    int B = 32, Cin = 32, Cout = 32, H = 256, W = 256;
    if (argc >= 6) { B = atoi(argv[1]); Cin = atoi(argv[2]); Cout = atoi(argv[3]); H = atoi(argv[4]); W = atoi(argv[5]); }
    if (Cin % 8 || Cout % 32 || H % TR || W % TC || Cout > 64) { printf("unsupported shape\n"); return 1; }
    const int NCB = Cout / 32, nchunk = Cin / 8;
    const size_t nx = (size_t)B * Cin * H * W, nw = (size_t)Cout * Cin * 9;
    std::vector<float> hx(nx), hw(nw), hb(Cout);
    srand(1);
    for (auto& f : hx) { f = (float)rand() / (float)RAND_MAX; f = f < 0.4f ? 0.f : f * 2.f - 0.8f; }   // post-ReLU like
    for (auto& f : hw) f = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto& f : hb) f = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    std::vector<unsigned short> hpk((size_t)nchunk * 5 * 3 * NCB * 64 * 8, 0);
    for (int c = 0; c < nchunk; ++c)
        for (int p = 0; p < 5; ++p)
            for (int cb = 0; cb < NCB; ++cb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = cb * 32 + (l & 31), ci = c * 8 + j, t = 2 * p + (l >> 5);
                        unsigned short s3[3] = {0, 0, 0};
                        if (t < 9) split3(hw[((size_t)co * Cin + ci) * 9 + t], s3);
                        for (int pl = 0; pl < 3; ++pl)
                            hpk[((((size_t)(c * 5 + p) * 3 + pl) * NCB + cb) * 64 + l) * 8 + j] = s3[pl];
                    }
    float *dx, *db;
    u32x4 *dw, *dxp, *dyp;
    const size_t nxu = (size_t)B * (Cin / 8) * 3 * H * W, nyu = (size_t)B * (Cout / 8) * 3 * H * W;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&db, Cout * 4)); CK(hipMalloc(&dw, hpk.size() * 2));
    CK(hipMalloc(&dxp, nxu * 16)); CK(hipMalloc(&dyp, nyu * 16));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hpk.data(), hpk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_planes_kernel, dim3(4096), dim3(256), 0, 0, dx, dxp, B, Cin, H * W);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    Args a{dxp, dw, db, dyp, B, Cin, Cout, H, W, 1, B * (H / TR) * (W / TC), getenv("DIAG") ? atoi(getenv("DIAG")) : 0, getenv("SKEW") ? atoi(getenv("SKEW")) : 0};
    int grid = a.ntiles < 512 ? a.ntiles : 512;
    if (getenv("GRID")) grid = atoi(getenv("GRID"));
    if (getenv("RES") && atoi(getenv("RES")) && NCB == 1 && (W % TC2) == 0) {
        // (correctness of this form: the output planes are compared with the first form's below)
        run_res<9>(a, "9 products");
        std::vector<unsigned short> y_res(nyu * 8), y_ref(nyu * 8);
        CK(hipMemcpy(y_res.data(), dyp, nyu * 16, hipMemcpyDeviceToHost));
        auto k9 = conv_b3s_kernel<1, 9, false>;
        CK(hipFuncSetAttribute((const void*)k9, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        hipLaunchKernelGGL(k9, dim3(grid), dim3(256), LDS_BYTES, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y_ref.data(), dyp, nyu * 16, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < y_res.size(); ++i) bad += y_res[i] != y_ref[i];
        printf("resident-filter form vs first form (9 products): %zu of %zu bf16 words differ\n", bad, y_res.size());
        run_res<6>(a, "6 products");
        return bad ? 1 : 0;
    }
    const bool spread = getenv("SPREAD") && atoi(getenv("SPREAD"));
    if (NCB == 1) {
        if (spread) {
            run<1, 9, true>(a, grid, "9 products (spread epilogue)", hx, hw, hb);
            run<1, 6, true>(a, grid, "6 products (spread epilogue)", hx, hw, hb);
        } else {
            run<1, 9, false>(a, grid, "9 products", hx, hw, hb);
            run<1, 6, false>(a, grid, "6 products", hx, hw, hb);
        }
    } else {
        run<2, 9, false>(a, grid, "9 products", hx, hw, hb);
        run<2, 6, false>(a, grid, "6 products", hx, hw, hb);
    }
    return 0;
}
