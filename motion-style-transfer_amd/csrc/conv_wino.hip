// 3x3 convolution in the Winograd F(2x2, 3x3) form on the fp32 matrix cores (round 4; DESIGN.md section 4.20 / section 8 item 7;
// the measured prototype with its ablations is tools/conv_wino_proto.hip).
//
//   Y = A^T [ sum over cin (G g G^T) (.) (B^T d B) ] A       per 2x2 output block: 16 multiplies per input channel instead of 36,
//   i.e. 2.25x fewer v_mfma_f32_16x16x4_f32 than the implicit GEMM of conv_mfma.hip -- in fp32 throughout (the error against fp64 is
//   BELOW the direct form's: fewer additions reach an accumulator).  The same kernel serves the forward convolution and the data
//   gradient (flipped / transposed filter = the mode-1 packing of ynet_pack_weight).
//
//   MFMA        D[m = cout][n = block] += U[xi,nu][cout][cin 4] * V[xi,nu][cin 4][block]; a wave owns one PAIR of output rows x 32
//               columns = 16 blocks and NCB * 16 output channels: 16 (xi,nu) * NCB accumulators of 4 registers.  The lane that
//               needs V = B^T d B computes it from its own 4x4 patch (four 8-byte-aligned LDS reads, 16 packed additions).
//   filters     transformed once per weight version by wino_filter_kernel into MFMA fragment order; ALL of them stay in LDS for
//               the launch (64 * Cin * Cout bytes <= 64 KB: the shapes ynet_conv2d_winograd_supported admits).
//   staging     every wave stages ITS OWN four input rows of 8 channels per chunk (5 LDS-DMA instructions of 64 lanes x 16 bytes =
//               8 channels x 4 rows x 10 units exactly) into a private two-slot ring: no barrier in the loop.  The rows are shifted by
//               4 bytes in LDS (a 16-byte LDS-DMA takes a 4-byte aligned destination: tools/lds_dma_align_probe.hip), which puts
//               every patch's first column on an 8-byte boundary while the global units stay whole inside / outside the image; plane
//               pitch 640 bytes = 128 (mod 256): conflict-free.  Addresses are a static per-lane offset + a scalar offset.
//   scheduling  the two waves of a SIMD are served oldest-first, so a workgroup hands its row pairs out from a counter in LDS, one
//               ahead of the pair being computed (a static split left the younger four waves 25 % behind).
//   epilogue    Y = A^T M A packed over channel pairs, bias through M[1][1], ReLU, 8-byte stores; variants: through the ReLU backward of
//               the layer below (the activation fetched at the store addresses), + a precomputed additive term, + the 2x2 max-pooled copy
//               (a lane holds exactly the block it pools).
//   kernels     conv_wino_kernel<NCB, NCH, EM, NW>: one source, Cin = 8 NCH in {16, 32}, Cout = 16 NCB in {16, 32}, chunks of 8 channels;
//               EM 0 plain, 1 through a ReLU backward (float activation fetched), 2 the same from the 1-bit mask, 3 plain + that mask written.
//               conv_wino_cat_kernel<2, EPI>: up to three concatenated sources padded to multiples of 4 channels (<= 56), chunks of 4;
//               EPI 0 plain, 2 + additive term, 3 + pooled copy, 4 / 5 = 0 / 2 + the 1-bit mask written, 6 = 3 + one byte per pooled block
//               (arg-max and ReLU bits for the pool's backward).
//               conv_wino16_kernel<EPI> (the slice form), conv_wino_up_kernel<NCH> / conv_wino16_up_kernel (bilinear x2 inside): further down.
//               Wider convolutions are composed by the caller (ops.conv2d_raw): output-channel slices, or a second launch that adds onto
//               the first one's output.
#include "ynet_common.h"
#include "bce_element.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define WN_TH 16
#define WN_TW 32
#define WN_LQ 10                          // 16-byte units per staged row: columns x0 - 4 .. x0 + 35
#define WN_ROWF 40
#define WN_PLANE_F 160                    // 4 rows
#define WN_SLOT_BYTES (320 * 16 + 16)     // 8 channels x 4 rows x 10 units, + the 4-byte shift
#define WN_RING_BYTES (2 * WN_SLOT_BYTES)
#define WN_THREADS 512

struct WinoArgs {
    const float* x;        // [B] x (x_bs floats) : cin planes of H x W
    const f32x4* u;        // transformed filters in fragment order (ynet_winograd_filter)
    const float* bias;     // cout floats or NULL
    float* y;              // [B] x (y_bs floats) : cout planes
    long long x_bs, y_bs;
    int B, H, W, relu, ntiles;
    const float* emask;    // EM: [B] x (emask_bs floats), cout planes -- the post-ReLU activation whose backward is applied to y (y = emask > 0 ? y : 0)
    long long emask_bs;
    unsigned* wbits;       // the Winograd-native 1-bit mask [B][H / 2][W / 32][64 lanes] (NCB = 2): written (EM 3) or applied (EM 2)
    float* y2;             // EM 5 / 6 (NCB = 3): the destination of output blocks 1 and 2 (32 planes); block 0 goes to y (16 planes)
    long long y2_bs;
    // EM 7 (NCB = 2): the 1 x 1 predictor + BCE-with-logits + the predictor's data gradient in the epilogue (wino_epilogue_pred) -- y receives dX, the convolution's own
    // output is never written
    const float* pw;       // the predictor's packed filter [32 padded rows][pco_pad] (ynet_pack_weight, forward layout of a 1 x 1 filter)
    const float* pb;       // its bias or NULL
    int pco, pco_pad;      // predictor outputs (<= 16) and the packed filter's column padding
    const float* t_xy;     // the BLOB form of the target (pred_bce_kernel): plane (b, o) = the kernlen x kernlen blob at the rounded position t_xy[2 (b pco + o) ..]
    const float* t_blob;
    int t_m, t_S;
    float* logits;         // [B][pco][H][W]
    double* partial;       // [gridDim.x] loss partials, summed in a fixed order by the last workgroup
    unsigned* ticket;
    float* loss;
    long long n_loss;      // B pco H W
    float gs;              // expected upstream gradient / n_loss
};

__device__ __forceinline__ f32x2 wn_v01(f32x2 tl, f32x2 th) {      // (t0 - t2, t1 + t2)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(tl), "v"(th));
    return r;
}
__device__ __forceinline__ f32x2 wn_v23(f32x2 tl, f32x2 th) {      // (t2 - t1, t1 - t3)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(tl), "v"(th));
    return r;
}

// Y = A^T M A packed over channel pairs, the bias through M[1][1], ReLU, 8-byte stores at static-lane + scalar offsets.  EPI 1: the
// activation of the block's 2 x 2 x 4 outputs per lane is fetched before the transform it gates (y = act > 0 ? y : 0); EPI 2: a
// precomputed term is fetched the same way and added in front of the ReLU (y = relu(conv + bias + addend)); EPI 3: the 2 x 2 max-pooled
// copy of the output is written too -- a lane holds exactly the 2 x 2 block it pools (rm / sm_t: the pooled tensor, stp: the lane's
// static offset there, HW / 4 pixels per plane)
// BITS (round 5, the Winograd-native 1-bit ReLU mask: one 32-bit word per lane and unit at NCB = 2, bit ((cb * 2 + h) * 2 + k) * 4 + 2 * row + column
// = "output channel cb * 16 + 4 kq + 2 h + k of this lane's 2 x 2 block is positive"): 1 -- the forward launch of a conv -> ReLU -> conv chain returns
// the word of its outputs in *wbits; 2 -- the data gradient of the chain's second convolution is gated by that word (*wbits) instead of fetching
// the 16 activation quads of EPI 1 (268 MB less traffic and no fetch latency in front of the stores of a 32-channel launch at 256^2, B 32).
// S2D (round 6): the 2 x 2 block of a lane is stored SPACE-TO-DEPTH -- element (row r, column c) of the block goes to plane (2 r + c) * (16 NCB) + channel of a
// tensor [4 * 16 NCB][H / 2][W / 2] (st0 / so_t are then the lane's / the unit's offsets in a LOW-resolution plane): the layout in which the gradient of an
// up-convolution's output is consumed by ynet_upsample2x_conv2d_dgrad (a 3 x 3 convolution at the low resolution).
// CB0 / CB1: the output blocks [CB0, CB1) of the accumulators go to this destination, as its channels 0 .. 16 (CB1 - CB0) - 1 (the 48-channel launch of a data
// gradient with two destinations calls the epilogue once per destination).
template <int NCB, int EPI, int BITS = 0, bool S2D = false, int CB0 = 0, int CB1 = NCB>
__device__ __forceinline__ void wino_epilogue(f32x4 (&acc)[16][NCB], const f32x2 (&bias2)[NCB][2], float floor_v, __amdgpu_buffer_rsrc_t ry,
                                              __amdgpu_buffer_rsrc_t rm, unsigned st0, unsigned st1, unsigned so_t, unsigned sm_t, int HW, unsigned stp = 0,
                                              unsigned* wbits = nullptr, unsigned char* pcode = nullptr) {
    unsigned word = 0u;
    if constexpr (BITS == 2) word = *wbits;
#pragma unroll
    for (int cb = CB0; cb < CB1; ++cb) {
        constexpr int NB = CB1 - CB0;      // blocks of this destination
        const int cd = cb - CB0;           // the block's index there
        u32x2 mk[2][2][2];
        if constexpr (EPI == 1 || EPI == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned sm = sm_t + (unsigned)((cd * 16 + 2 * h + k) * HW * 4);
                    mk[h][k][0] = __builtin_amdgcn_raw_buffer_load_b64(rm, st0, sm, 0);
                    mk[h][k][1] = __builtin_amdgcn_raw_buffer_load_b64(rm, st1, sm, 0);
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x2 m[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                m[e] = h == 0 ? __builtin_shufflevector(acc[e][cb], acc[e][cb], 0, 1) : __builtin_shufflevector(acc[e][cb], acc[e][cb], 2, 3);
            m[5] = m[5] + bias2[cb][h];
            f32x2 r0[4], r1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r0[j] = m[j] + m[4 + j] + m[8 + j];
                r1[j] = m[4 + j] - m[8 + j] - m[12 + j];
            }
            const f32x2 o00 = r0[0] + r0[1] + r0[2], o01 = r0[1] - r0[2] - r0[3];
            const f32x2 o10 = r1[0] + r1[1] + r1[2], o11 = r1[1] - r1[2] - r1[3];
#pragma unroll
            for (int k = 0; k < 2; ++k) {      // channel cb * 16 + 4 kq + 2 h + k
                f32x2 row0 = {o00[k], o01[k]}, row1 = {o10[k], o11[k]};
                if constexpr (EPI == 2) {
                    row0 += __builtin_bit_cast(f32x2, mk[h][k][0]);
                    row1 += __builtin_bit_cast(f32x2, mk[h][k][1]);
                }
                // (v < floor ? floor : v -- a NaN stays a NaN as in torch's relu; fmaxf would return the floor)
                row0 = f32x2{row0[0] < floor_v ? floor_v : row0[0], row0[1] < floor_v ? floor_v : row0[1]};
                row1 = f32x2{row1[0] < floor_v ? floor_v : row1[0], row1[1] < floor_v ? floor_v : row1[1]};
                if constexpr (EPI == 1) {
                    const f32x2 m0 = __builtin_bit_cast(f32x2, mk[h][k][0]), m1 = __builtin_bit_cast(f32x2, mk[h][k][1]);
                    row0 = f32x2{m0[0] > 0.f ? row0[0] : 0.f, m0[1] > 0.f ? row0[1] : 0.f};
                    row1 = f32x2{m1[0] > 0.f ? row1[0] : 0.f, m1[1] > 0.f ? row1[1] : 0.f};
                }
                const int bit0 = ((cb * 2 + h) * 2 + k) * 4;
                if constexpr (BITS == 2) {
                    row0 = f32x2{(word >> bit0) & 1u ? row0[0] : 0.f, (word >> (bit0 + 1)) & 1u ? row0[1] : 0.f};
                    row1 = f32x2{(word >> (bit0 + 2)) & 1u ? row1[0] : 0.f, (word >> (bit0 + 3)) & 1u ? row1[1] : 0.f};
                }
                if constexpr (BITS == 1)      // (v > 0: NaN and -0 count as "not positive", as the float comparison of EPI 1 does)
                    word |= (row0[0] > 0.f ? 1u << bit0 : 0u) | (row0[1] > 0.f ? 2u << bit0 : 0u) | (row1[0] > 0.f ? 4u << bit0 : 0u) | (row1[1] > 0.f ? 8u << bit0 : 0u);
                if constexpr (S2D) {
                    const unsigned hw = (unsigned)(HW >> 2), ph = (unsigned)(NB * 16) * hw * 4u;
                    const unsigned sc = so_t + (unsigned)(cd * 16 + 2 * h + k) * hw * 4u;
                    // (scalars first: __builtin_bit_cast applied to a vector-element lvalue reads the vector's FIRST element -- hipcc stored row0[0] twice)
                    const float e00 = row0[0], e01 = row0[1], e10 = row1[0], e11 = row1[1];
                    // Two neighbouring lanes (low-resolution columns j, j + 1) trade one element per row so that the even one holds the column-phase-0 pair and the
                    // odd one the column-phase-1 pair: 2 eight-byte stores per lane instead of 4 four-byte ones (same 64-byte segments, half the instructions).
                    const bool odd = __lane_id() & 1u;
                    const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? e00 : e01), 0xB1, 0xF, 0xF, false));
                    const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? e10 : e11), 0xB1, 0xF, 0xF, false));
                    const f32x2 p0 = odd ? f32x2{g0, e01} : f32x2{e00, g0}, p1 = odd ? f32x2{g1, e11} : f32x2{e10, g1};
                    const unsigned vo = st0 + (odd ? ph - 4u : 0u);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p0), ry, vo, sc, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p1), ry, vo, sc + 2u * ph, 0);
                    continue;
                }
                const unsigned so = so_t + (unsigned)((cd * 16 + 2 * h + k) * HW * 4);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, row0), ry, st0, so, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, row1), ry, st1, so, 0);
                if constexpr (EPI == 3) {      // max over the block, NaN if any element is (torch's max_pool2d)
                    float pm = fmaxf(fmaxf(row0[0], row0[1]), fmaxf(row1[0], row1[1]));
                    if (__builtin_isunordered(row0[0], row0[1]) || __builtin_isunordered(row1[0], row1[1])) pm = __builtin_nanf("");
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pm), rm, stp, sm_t + (unsigned)((cd * 16 + 2 * h + k) * (HW >> 2) * 4), 0);
                    if constexpr (BITS == 3) {      // what the pool's backward needs of this block, in one byte: bits 0..1 the arg-max (first maximum in scan
                        float mx = row0[0];         // order, a NaN wins: maxpool2_bwd_add_kernel's rule), bits 2..5 "element is positive" (the ReLU backward)
                        unsigned arg = 0u;
                        if (row0[1] > mx || row0[1] != row0[1]) { mx = row0[1]; arg = 1u; }
                        if (row1[0] > mx || row1[0] != row1[0]) { mx = row1[0]; arg = 2u; }
                        if (row1[1] > mx || row1[1] != row1[1]) { mx = row1[1]; arg = 3u; }
                        const unsigned code = arg | (row0[0] > 0.f ? 4u : 0u) | (row0[1] > 0.f ? 8u : 0u) | (row1[0] > 0.f ? 16u : 0u) | (row1[1] > 0.f ? 32u : 0u);
                        pcode[(cd * 16 + 2 * h + k) * (HW >> 2)] = (unsigned char)code;
                    }
                }
            }
        }
    }
    if constexpr (BITS == 1) *wbits = word;
}

// EM 7 / 8: the last decoder convolution + ReLU with the predictor, the criterion and the predictor's data gradient in its epilogue (models/ynet.py:467,469
// `self.predictor(x)`, utils/train_epoch.py:93-94,105-106 `criterion(pred_map, gt_map) * loss_scale`) -- what pred_bce_kernel (glue.hip) does in a pass of its own
// over the 32 activation planes, done where those activations are still in registers: the convolution's output is never written, never read back.
//   y[c]       = relu(A^T M A + bias)                                 lane (n, kq): channels cb 16 + 4 kq + j (j = 0..3) of the 2 x 2 block n, 8 x 4 values
//   z[o]       = pb[o] + sum_c pw[o][c] y[c]                          on the matrix cores: k-step (cb, j) takes channel cb 16 + 4 kq' + j from lane group kq' -- the
//                                                                     lane's own value is the B operand; the result lands as o = 4 kq + i per lane
//   loss      += BCE-with-logits(z[o], t[o]);  dz[o] = (sigmoid(z[o]) - t[o]) gs
//   dX[c]      = y[c] > 0 ? sum_o pw[o][c] dz[o] : 0                  again on the matrix cores (k-step i takes o = 4 kq' + i), landing in y's own layout
// 64 matrix instructions per unit on top of the 256 of the convolution; the logits and dX leave through the stores the plain epilogue uses.
// wa / wb: the lane's A operands, [mb][cb][j] = pw[o = 16 mb + (lane & 15)][c = cb 16 + 4 (lane >> 4) + j] and [mb][cb][i] = pw[o = 16 mb + 4 (lane >> 4) + i][c = cb 16 + (lane & 15)].
// MB: blocks of 16 predictor outputs (1: <= 16 outputs, EM 7; 2: <= 32, EM 8 -- the 30 prediction steps of the long-term configs: 128 extra matrix instructions per unit);
// PosT: the rounded positions of the target planes in LDS -- int, or short where two blocks of operand tables leave the blob table and the positions 5.8 KB (MB = 2;
// 32767 = "this plane is all zero").
template <int MB, typename PosT>
__device__ __forceinline__ float wino_epilogue_pred(f32x4 (&acc)[16][2], const f32x2 (&bias2)[2][2], const f32x4* tab,      // tab: this lane's column of the operand tables in LDS
                                                    const WinoArgs& a, __amdgpu_buffer_rsrc_t rdx, __amdgpu_buffer_rsrc_t rlog, unsigned st_rm, unsigned so_rm, int b,
                                                    int py0, int px0, int kq, int H, int W, const float* blob_tab, const PosT* pos_lds) {
    const int HW = H * W;
    float yv[2][4][4];      // [cb][j = 2 h + k][p = 2 r + c]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x2 m[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                m[e] = h == 0 ? __builtin_shufflevector(acc[e][cb], acc[e][cb], 0, 1) : __builtin_shufflevector(acc[e][cb], acc[e][cb], 2, 3);
            m[5] = m[5] + bias2[cb][h];
            f32x2 r0[4], r1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r0[j] = m[j] + m[4 + j] + m[8 + j];
                r1[j] = m[4 + j] - m[8 + j] - m[12 + j];
            }
            const f32x2 o00 = r0[0] + r0[1] + r0[2], o01 = r0[1] - r0[2] - r0[3];
            const f32x2 o10 = r1[0] + r1[1] + r1[2], o11 = r1[1] - r1[2] - r1[3];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float v00 = o00[k], v01 = o01[k], v10 = o10[k], v11 = o11[k];
                // (v < 0 ? 0 : v -- a NaN stays a NaN as in torch's relu)
                yv[cb][2 * h + k][0] = v00 < 0.f ? 0.f : v00;
                yv[cb][2 * h + k][1] = v01 < 0.f ? 0.f : v01;
                yv[cb][2 * h + k][2] = v10 < 0.f ? 0.f : v10;
                yv[cb][2 * h + k][3] = v11 < 0.f ? 0.f : v11;
            }
        }
    // ---- the predictor: four independent accumulator chains (one per pixel of the block)
    // (one block of 16 outputs at a time: its four accumulators, the criterion on them, its logits stored -- only dz stays for the data gradient)
    const unsigned st1 = st_rm + (unsigned)(W * 4);
    float s = 0.f;
    f32x4 dx[2][4];      // the predictor's data gradient, in y's layout (accumulated block by block)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int p = 0; p < 4; ++p) dx[cb][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto block = [&](const int mb) {
        f32x4 z[4], dz[4];
        {
            const f32x4 pbv = tab[(4 * MB + mb) * 64];
#pragma unroll
            for (int p = 0; p < 4; ++p) z[p] = pbv;
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const f32x4 wa = tab[(mb * 2 + cb) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) z[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], yv[cb][j][p], z[p], 0, 0, 0);
        }
        // ---- the criterion on the lane's 4 outputs x 4 pixels; the target is the Gaussian blob at the rounded position (pred_bce_kernel's BLOB form)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 16 * mb + 4 * kq + i;
            const bool valid = o < a.pco;
            // (rounded position of plane (b, o) from LDS -- far outside for a plane whose window leaves the template)
            const int rx = valid ? (int)pos_lds[2 * (b * a.pco + o)] : 32767, ry = valid ? (int)pos_lds[2 * (b * a.pco + o) + 1] : 32767;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int by = py0 + (p >> 1) - ry + a.t_m / 2, bx = px0 + (p & 1) - rx + a.t_m / 2;
#if defined(YNET_PRED_EPI_DIAG) && (YNET_PRED_EPI_DIAG & 2)
                const float t = 0.f;      // (development build, WRONG results: no target lookup -- tools/ab_conv_pred_bce_epi.sh)
#else
                const float t = (by >= 0 && by < a.t_m && bx >= 0 && bx < a.t_m) ? blob_tab[by * a.t_m + bx] : 0.f;
#endif
                float de;
#if defined(YNET_PRED_EPI_DIAG) && (YNET_PRED_EPI_DIAG & 1)
                de = (z[p][i] - t) * a.gs;      // (development build, WRONG results: no exp / log / rcp)
                const float l = z[p][i] - t;
#else
                const float l = bce_element<true, true>(z[p][i], t, a.gs, de);
#endif
                s += valid ? l : 0.f;
                dz[p][i] = valid ? de : 0.f;
            }
        }
        // the logits: plane o = 16 mb + 4 kq + i at the lane's static offset (channel 4 kq) + 16 mb + i planes; planes >= pco are beyond the descriptor (dropped by the range check)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 row0 = {z[0][i], z[1][i]}, row1 = {z[2][i], z[3][i]};
            const unsigned so = so_rm + (unsigned)((16 * mb + i) * HW * 4);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, row0), rlog, st_rm, so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, row1), rlog, st1, so, 0);
        }
        // this block's part of the predictor's data gradient (k-step i takes output 16 mb + 4 kq' + i from lane group kq')
        const f32x4 wb0 = tab[(2 * MB + mb * 2) * 64], wb1 = tab[(2 * MB + mb * 2 + 1) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int p = 0; p < 4; ++p) dx[cb][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(cb == 0 ? wb0[i] : wb1[i], dz[p][i], dx[cb][p], 0, 0, 0);
    };
    if constexpr (MB == 1) {
        block(0);
    } else {
        // (a real loop over the blocks: unrolled, the two blocks' chains were interleaved and 36 registers spilled -- 754 against 637 us at 512^2, B 16)
#pragma unroll 1
        for (int mb = 0; mb < MB; ++mb) block(mb);
    }
    // ---- the ReLU backward of y applied to the data gradient, stored where y would have been
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 row0 = {yv[cb][j][0] > 0.f ? dx[cb][0][j] : 0.f, yv[cb][j][1] > 0.f ? dx[cb][1][j] : 0.f};
            const f32x2 row1 = {yv[cb][j][2] > 0.f ? dx[cb][2][j] : 0.f, yv[cb][j][3] > 0.f ? dx[cb][3][j] : 0.f};
            const unsigned so = so_rm + (unsigned)((cb * 16 + j) * HW * 4);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, row0), rdx, st_rm, so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, row1), rdx, st1, so, 0);
        }
    return s;
}

// one k-step of 4 input channels: V = B^T d B of the lane's patch (rows dl / dh), then 16 NCB MFMAs against the filter fragments at wl
template <int NCB, bool FIRST>
__device__ __forceinline__ void wino_kstep(f32x4 (&acc)[16][NCB], const f32x2 (&dl)[4], const f32x2 (&dh)[4], const f32x4* wl, int lane) {
    const f32x2 tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[1] - dl[3]};
    const f32x2 th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[1] - dh[3]};
    f32x2 v01[4], v23[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v01[i] = wn_v01(tl[i], th[i]);
        v23[i] = wn_v23(tl[i], th[i]);
    }
    // (inline asm is opaque to hipcc's hazard recognizer: the wait states between a vector write and the MFMA that reads it)
    asm volatile("s_nop 3" : "+v"(v01[0]), "+v"(v01[1]), "+v"(v01[2]), "+v"(v01[3]), "+v"(v23[0]), "+v"(v23[1]), "+v"(v23[2]), "+v"(v23[3]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 w[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) w[cb] = wl[(q * NCB + cb) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bv = e < 2 ? v01[q][e] : v23[q][e - 2];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float av = w[cb][e];
                if constexpr (FIRST) acc[q * 4 + e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else acc[q * 4 + e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[q * 4 + e][cb], 0, 0, 0);
            }
        }
    }
}

// NCB: 16-channel output blocks (cout = 16 NCB), NCH: chunks of 8 input channels (cin = 8 NCH; even: chunk c lives in slot c & 1 of the
// wave's ring, and the chunk two ahead -- of this pair or the next -- takes the slot just read)
template <int NCB, int NCH, int EM, int NW>
__global__ __launch_bounds__(NW * 64, 1) void conv_wino_kernel(const WinoArgs a) {
    extern __shared__ f32x4 smem[];
    constexpr int WQ = 8 * NCB * 64;      // units of one chunk's filters: [2 k-steps][4 quads of (xi,nu)][NCB][64 lanes]
    constexpr int NT = NW * 64;           // NW waves: 8, or 12 for the 16-channel form (64 accumulator registers: three waves per SIMD)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W;
    const int tiles_x = W / WN_TW, tiles_y = H / WN_TH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned ring0 = lds0 + (unsigned)(NCH * WQ * 16) + (unsigned)(wave * WN_RING_BYTES);

    // static DMA geometry of this lane: unit j * 64 + lane -> (channel of the chunk, row, unit of the row)
    unsigned rel[5], edge[5];             // edge bits: 1 top row, 2 bottom row, 4 left unit, 8 right unit
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int u = j * 64 + lane, plane = u / 40, rem = u - plane * 40, r = rem / WN_LQ, xq = rem - r * WN_LQ;
        rel[j] = (unsigned)((plane * HW + r * W + 4 * xq) * 4);
        edge[j] = (r == 0 ? 1u : 0u) | (r == 3 ? 2u : 0u) | (xq == 0 ? 4u : 0u) | (xq == WN_LQ - 1 ? 8u : 0u);
    }
    // the input descriptor starts one row and one unit BEFORE the tensor: the scalar offset of a chunk (its first output row and
    // column) is then never negative, and the lanes that would read in front of / behind a plane are exactly the edge lanes
    // (sent to an offset beyond the descriptor: zero fill)
    // (every descriptor spans ONE image -- base + b * batch stride in scalar registers, built where the image is known: a tensor of any
    //  size is addressed, and with fewer than 2^31 bytes per image the zero-fill offset 0x80000000 is beyond every descriptor whether
    //  or not the hardware adds the scalar offset before the range check)
    const unsigned lead = (unsigned)((W + 4) * 4);
    const unsigned x_img = (unsigned)(NCH * 8 * HW * 4) + lead, y_img = (unsigned)(NCB * 16 * HW * 4);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(a.u), 0, (unsigned)(NCH * WQ * 16), 0x00020000);

    // XCD-aware walk (workgroups are dealt round-robin over the 8 XCDs): each XCD sweeps its own contiguous eighth of the tiles
    const bool xcd_walk = (gridDim.x & 7) == 0 && a.ntiles >= (int)gridDim.x;
    const int per_xcd = (a.ntiles + 7) >> 3;
    const int gstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tile_first = xcd_walk ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tile_end = xcd_walk ? min(a.ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd) : a.ntiles;
    if (tile_first >= tile_end) {
        if constexpr (EM == 7 || EM == 8) {      // (a workgroup without tiles still takes its ticket: the last one to arrive sums the loss partials)
            if (tid == 0) {
                a.partial[blockIdx.x] = 0.0;
                __threadfence();
                if (atomicAdd(a.ticket, 1u) == gridDim.x - 1) {
                    __threadfence();
                    double t = 0.0;
                    for (unsigned i = 0; i < gridDim.x; ++i) t += ((volatile double*)a.partial)[i];
                    a.loss[0] = (float)(t / (double)a.n_loss);
                    *a.ticket = 0u;
                }
            }
        }
        return;
    }
    const int my_tiles = (tile_end - tile_first + gstride - 1) / gstride;
    const int total_units = my_tiles * 8;

    const int n = lane & 15, kq = lane >> 4;
    const float floor_v = a.relu ? 0.f : -INFINITY;
    f32x2 bias2[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            bias2[cb][h] = a.bias ? f32x2{a.bias[cb * 16 + 4 * kq + 2 * h], a.bias[cb * 16 + 4 * kq + 2 * h + 1]} : f32x2{0.f, 0.f};
    // (the bias loads are complete before the first DMA: hipcc would otherwise drain the DMA queue where the epilogue first reads them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(bias2[cb][h]));
    // static store offsets of this lane: output channel 4 kq (+ the rest by the scalar offset), column 2 n, rows 0 / 1 of the pair
    // (EM 4, the space-to-depth store: the lane's offset in a low-resolution plane -- channel 4 kq, column n)
    // (EM 5 / 6, NCB = 3: block 0 to a.y -- row-major / space-to-depth --, blocks 1 and 2 row-major to a.y2)
    const unsigned st_s2d = (unsigned)((4 * kq * (HW >> 2) + n) * 4), st_rm = (unsigned)((4 * kq * HW + 2 * n) * 4);
    const unsigned st0 = EM == 4 ? st_s2d : st_rm, st1 = st0 + (unsigned)(W * 4);

    // all transformed filters -> LDS, once; the workgroup's unit counter
#pragma unroll
    for (int j = 0; j < (NCH * WQ + NT - 1) / NT; ++j)
        if (j * NT + wave * 64 < NCH * WQ)      // (whole wave instructions: NCH * WQ is a multiple of 64)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(uintptr_t)(lds0 + (unsigned)((j * NT + wave * 64) * 16)), 16,
                                                     (unsigned)((j * NT + tid) * 16), 0, 0, 0);
    unsigned* unit_ctr = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(smem) + NCH * WQ * 16 + NW * WN_RING_BYTES);
    if (tid == 0) *unit_ctr = (unsigned)NW;
    // EM 7: the predictor's filter as the lanes' matrix operands (see wino_epilogue_pred), [5][64 lanes] x 16 bytes behind the unit counter: units 0 / 1 = wa[cb], 2 / 3 = wb[cb],
    // 4 = the bias of the lane's four outputs; then 8 doubles for the workgroup's loss partial
    f32x4* ptab = reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(smem) + NCH * WQ * 16 + NW * WN_RING_BYTES + 16);
    constexpr int PMB = EM == 8 ? 2 : 1;      // blocks of 16 predictor outputs (EM 7: one, EM 8: two); tables: units [mb 2 + cb] = wa, [2 PMB + mb 2 + cb] = wb, [4 PMB + mb] = bias
    if constexpr (EM == 7 || EM == 8) {
        if (tid < 64) {
            const int o_a = tid & 15, g = tid >> 4;
#pragma unroll
            for (int mb = 0; mb < PMB; ++mb) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    f32x4 va, vb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        va[j] = 16 * mb + o_a < a.pco ? a.pw[(cb * 16 + 4 * g + j) * a.pco_pad + 16 * mb + o_a] : 0.f;
                        vb[j] = 16 * mb + 4 * g + j < a.pco ? a.pw[(cb * 16 + o_a) * a.pco_pad + 16 * mb + 4 * g + j] : 0.f;
                    }
                    ptab[(mb * 2 + cb) * 64 + tid] = va;
                    ptab[(2 * PMB + mb * 2 + cb) * 64 + tid] = vb;
                }
                f32x4 vp;
#pragma unroll
                for (int j = 0; j < 4; ++j) vp[j] = (a.pb != nullptr && 16 * mb + 4 * g + j < a.pco) ? a.pb[16 * mb + 4 * g + j] : 0.f;
                ptab[(4 * PMB + mb) * 64 + tid] = vp;
            }
        }
        // ... then the blob table (EM 7) and the rounded position of every target plane (pred_bce_kernel's BLOB form: an all-zero plane when the H x W window around the
        // position would leave the S x S template -- encoded as a position far away)
        typedef typename std::conditional<EM == 7, int, short>::type PosT;
        float* blob_w = reinterpret_cast<float*>(ptab + 5 * PMB * 64) + 2 * NW + 4;
        PosT* pos_w = reinterpret_cast<PosT*>(blob_w + a.t_m * a.t_m);
        for (int i = tid; i < a.t_m * a.t_m; i += NT) blob_w[i] = a.t_blob[i];
        for (int i = tid; i < a.B * a.pco; i += NT) {
            const int rx = (int)rintf(a.t_xy[2 * i]), ry = (int)rintf(a.t_xy[2 * i + 1]);
            const int ox = a.t_S / 2 - rx, oy = a.t_S / 2 - ry;
            const bool inside = !(ox < 0 || oy < 0 || ox + W > a.t_S || oy + H > a.t_S);
            // (a position beyond +-32766 cannot be `inside`: the window would leave any template the host admits)
            pos_w[2 * i] = (PosT)(inside && rx > -32767 && rx < 32767 && ry > -32767 && ry < 32767 ? rx : 32767);
            pos_w[2 * i + 1] = (PosT)(inside && rx > -32767 && rx < 32767 && ry > -32767 && ry < 32767 ? ry : 32767);
        }
    }
    double acc_loss = 0.0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    auto next_unit = [&]() {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(unit_ctr, 1u);
        return (int)__builtin_amdgcn_readfirstlane(u);
    };

    auto dma_chunk = [&](int unit, int c) {       // the four input rows of row pair `unit` (tile unit / 8, pair unit % 8), chunk c -> slot c & 1
        const int t = tile_first + (unit >> 3) * gstride, slot = c & 1;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int y0 = ty * WN_TH + 2 * (unit & 7), x0 = tx * WN_TW;
        const unsigned em = (y0 == 0 ? 1u : 0u) | (y0 + 2 == H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + WN_TW == W ? 8u : 0u);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x + (long long)b * a.x_bs) - lead), 0, x_img, 0x00020000);
        const unsigned so = (unsigned)((c * 8 * HW + y0 * W + x0) * 4);
        const unsigned sb = ring0 + (unsigned)(slot * WN_SLOT_BYTES) + 4u;
        if (em == 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + (unsigned)(j * 1024)), 16, rel[j], so, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + (unsigned)(j * 1024)), 16,
                                                         (edge[j] & em) ? 0x80000000u : rel[j], so, 0, 0);
        }
    };

    int cur = wave, nxt = next_unit();
    if (cur < total_units) {
        dma_chunk(cur, 0);
        dma_chunk(cur, 1);
    }

    f32x4 acc[16][NCB];
    while (cur < total_units) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // chunk c has landed (the loads issued after it are those of the next chunk, if there is one)
            if (c + 1 < NCH || nxt < total_units) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const f32x4* wl = smem + c * WQ;
            const float* il = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(smem) + NCH * WQ * 16 + wave * WN_RING_BYTES +
                                                             (c & 1) * WN_SLOT_BYTES + 4);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // the lane's 4x4 patch of channel s * 4 + kq: staged rows 0 .. 3, floats 3 + 2n .. 6 + 2n of the row
                const float* ip = il + (s * 4 + kq) * WN_PLANE_F + 3 + 2 * n;
                f32x2 dl[4], dh[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dl[r] = *reinterpret_cast<const f32x2*>(ip + r * WN_ROWF);
                    dh[r] = *reinterpret_cast<const f32x2*>(ip + r * WN_ROWF + 2);
                }
                if (s == 1) {
                    // the slot is read out (this wave's own reads): it takes the chunk two ahead
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (c + 2 < NCH) dma_chunk(cur, c + 2);
                    else if (nxt < total_units) dma_chunk(nxt, c + 2 - NCH);
                }
                // V = B^T d B: rows, then the columns with source selection
                const f32x2 tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[1] - dl[3]};
                const f32x2 th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[1] - dh[3]};
                f32x2 v01[4], v23[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v01[i] = wn_v01(tl[i], th[i]);
                    v23[i] = wn_v23(tl[i], th[i]);
                }
                // (inline asm is opaque to hipcc's hazard recognizer: the wait states between a vector write and the MFMA that reads
                //  it are spent here)
                asm volatile("s_nop 3" : "+v"(v01[0]), "+v"(v01[1]), "+v"(v01[2]), "+v"(v01[3]), "+v"(v23[0]), "+v"(v23[1]), "+v"(v23[2]), "+v"(v23[3]));
                // 16 NCB MFMAs: (xi,nu) = 4 q + e, cout block cb; the pair's first k-step accumulates onto 0
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 w[NCB];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) w[cb] = wl[((s * 4 + q) * NCB + cb) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bv = e < 2 ? v01[q][e] : v23[q][e - 2];
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            const float av = w[cb][e];
                            if (c == 0 && s == 0) acc[q * 4 + e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            else acc[q * 4 + e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[q * 4 + e][cb], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- epilogue
        {
            const int t = tile_first + (cur >> 3) * gstride;
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
            const unsigned so_s2d = (unsigned)(((ty * (WN_TH / 2) + (cur & 7)) * (W >> 1) + tx * (WN_TW / 2)) * 4);
            const unsigned so_rm = (unsigned)(((ty * WN_TH + 2 * (cur & 7)) * W + tx * WN_TW) * 4);
            if constexpr (EM == 7 || EM == 8) {
                static_assert((EM != 7 && EM != 8) || NCB == 2, "the predictor epilogue is the 32-channel launch's");
                const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)b * a.y_bs, 0, y_img, 0x00020000);
                const __amdgpu_buffer_rsrc_t rlog =
                    __builtin_amdgcn_make_buffer_rsrc(a.logits + (long long)b * a.pco * HW, 0, (unsigned)(a.pco * HW * 4), 0x00020000);
                typedef typename std::conditional<EM == 7, int, short>::type PosT;
                const float* blob_lds = reinterpret_cast<const float*>(ptab + 5 * PMB * 64) + 2 * NW + 4;
                const float s_ = wino_epilogue_pred<PMB, PosT>(acc, bias2, ptab + lane, a, rdx, rlog, st_rm, so_rm, b, ty * WN_TH + 2 * (cur & 7), tx * WN_TW + 2 * n, kq, H, W,
                                                               blob_lds, reinterpret_cast<const PosT*>(blob_lds + a.t_m * a.t_m));
                acc_loss += (double)s_;
                cur = nxt;
                if (cur < total_units) nxt = next_unit();
                continue;
            }
            if constexpr (EM == 5 || EM == 6) {
                static_assert(EM < 5 || NCB == 3, "the two-destination epilogue is the 48-channel launch's");
                const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)b * a.y_bs, 0, (unsigned)(16 * HW * 4), 0x00020000);
                const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(a.y2 + (long long)b * a.y2_bs, 0, (unsigned)(32 * HW * 4), 0x00020000);
                if constexpr (EM == 6) wino_epilogue<NCB, 0, 0, true, 0, 1>(acc, bias2, floor_v, r0, r0, st_s2d, st_s2d, so_s2d, so_s2d, HW);
                else wino_epilogue<NCB, 0, 0, false, 0, 1>(acc, bias2, floor_v, r0, r0, st_rm, st_rm + (unsigned)(W * 4), so_rm, so_rm, HW);
                wino_epilogue<NCB, 0, 0, false, 1, 3>(acc, bias2, floor_v, r1, r1, st_rm, st_rm + (unsigned)(W * 4), so_rm, so_rm, HW);
                cur = nxt;
                if (cur < total_units) nxt = next_unit();
                continue;
            }
            const unsigned so_t = EM == 4 ? so_s2d : so_rm;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)b * a.y_bs, 0, y_img, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EM == 1 ? a.emask + (long long)b * a.emask_bs : a.y), 0, y_img, 0x00020000);
            unsigned* wb = (EM == 2 || EM == 3) ? a.wbits + (((long long)b * (H >> 1) + (ty * (WN_TH / 2) + (cur & 7))) * tiles_x + tx) * 64 + lane : nullptr;
            wino_epilogue<NCB, EM == 1 ? 1 : 0, EM == 2 ? 2 : (EM == 3 ? 1 : 0), EM == 4>(acc, bias2, floor_v, ry, rm, st0, st1, so_t, so_t, HW, 0, wb);
        }
        cur = nxt;
        if (cur < total_units) nxt = next_unit();
    }
    if constexpr (EM == 7 || EM == 8) {
        // the loss: lanes -> wave -> workgroup partial (fp64) -> the last workgroup to arrive sums the partials in index order (bitwise reproducible) and resets the ticket
        double* wsd = reinterpret_cast<double*>(ptab + 5 * PMB * 64);
        unsigned* last = reinterpret_cast<unsigned*>(wsd + NW);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc_loss += __shfl_xor(acc_loss, o, 64);
        if (lane == 0) wsd[wave] = acc_loss;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int i = 0; i < NW; ++i) t += wsd[i];
            a.partial[blockIdx.x] = t;
            __threadfence();
            *last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
        }
        __syncthreads();
        if (*last) {
            __threadfence();
            double t = 0.0;
            for (unsigned i = tid; i < gridDim.x; i += NT) t += ((volatile double*)a.partial)[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) wsd[wave] = t;
            __syncthreads();
            if (tid == 0) {
                double tt = 0.0;
                for (int i = 0; i < NW; ++i) tt += wsd[i];
                a.loss[0] = (float)(tt / (double)a.n_loss);
                *a.ticket = 0u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The multi-source form: the input is the (virtual) concatenation of up to three tensors with up to 56 channels in all -- the decoders'
// first convolutions, cat(up-sampled features 32, skip features 16 [, way-point map 1]) -> 32.  96-112 KB of transformed filters leave
// 6-8 KB of LDS per wave, so a chunk is 4 input channels (one MFMA k-step): 4 channels x 4 rows x 10 units = 160 units = 2.5 DMA
// instructions (the third with 32 lanes), two slots of 2.5 KB, the chunk two ahead issued right after a slot's patch is read.  Every
// source is padded to a multiple of 4 channels (zero planes by out-of-range offsets, zero filters); a chunk never straddles sources.
#define WC_SLOT_BYTES (160 * 16 + 16)
#define WC_RING_BYTES (2 * WC_SLOT_BYTES)
#define WC_MAX_SRC 3

struct WinoCatArgs {
    const float* x[WC_MAX_SRC];
    long long x_bs[WC_MAX_SRC];
    int x_c[WC_MAX_SRC];       // real channels of each source (its chunks: ceil(c / 4))
    int nsrc, nchunks;         // chunks of 4 channels over all sources
    const f32x4* u;
    const float* bias;
    float* y;
    long long y_bs;
    int B, H, W, relu, ntiles;
    const float* addend;       // EPI 2: [images] x (addend_bs floats), cout planes, image b % addend_bmod (b when the modulus is 0): y = relu(conv + bias + addend)
    long long addend_bs;
    int addend_bmod;
    float* pool;               // EPI 3: [B] x (pool_bs floats), cout planes of (H / 2) x (W / 2): the 2 x 2 max-pooled copy of y
    long long pool_bs;
    unsigned char* pcode;      // EPI_ 6: [B][cout][H / 2][W / 2] bytes: arg-max and ReLU bits of every pooled block (see wino_epilogue)
    unsigned* wbits;           // EPI 4 / 5 (= 0 / 2 + the Winograd-native 1-bit mask of the output written, see wino_epilogue)
};

template <int NCB, int EPI_>
__global__ __launch_bounds__(WN_THREADS, 1) void conv_wino_cat_kernel(const WinoCatArgs a) {
    constexpr int EPI = EPI_ == 4 ? 0 : (EPI_ == 5 ? 2 : (EPI_ == 6 ? 3 : EPI_));
    constexpr int BITS = EPI_ == 6 ? 3 : (EPI_ >= 4 ? 1 : 0);
    constexpr bool ADD = EPI == 2;
    extern __shared__ f32x4 smem[];
    constexpr int WQ = 4 * NCB * 64;      // units of one chunk's filters: [4 quads of (xi,nu)][NCB][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W, nchunks = a.nchunks;
    const int tiles_x = W / WN_TW, tiles_y = H / WN_TH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned wbytes = (unsigned)(nchunks * WQ * 16);
    const unsigned ring0 = lds0 + wbytes + (unsigned)(wave * WC_RING_BYTES);

    // static DMA geometry: unit j * 64 + lane -> (channel of the chunk, row, unit of the row); the third instruction has 32 lanes
    unsigned rel[3], edge[3], planeof[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = j * 64 + lane, plane = u / 40, rem = u - plane * 40, r = rem / WN_LQ, xq = rem - r * WN_LQ;
        rel[j] = (unsigned)((plane * HW + r * W + 4 * xq) * 4);
        edge[j] = (r == 0 ? 1u : 0u) | (r == 3 ? 2u : 0u) | (xq == 0 ? 4u : 0u) | (xq == WN_LQ - 1 ? 8u : 0u);
        planeof[j] = (unsigned)plane;
    }
    const unsigned lead = (unsigned)((W + 4) * 4);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(a.u), 0, wbytes, 0x00020000);
    const unsigned y_img = (unsigned)(NCB * 16 * HW * 4);      // (one image per descriptor, as in conv_wino_kernel)

    const bool xcd_walk = (gridDim.x & 7) == 0 && a.ntiles >= (int)gridDim.x;
    const int per_xcd = (a.ntiles + 7) >> 3;
    const int gstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tile_first = xcd_walk ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tile_end = xcd_walk ? min(a.ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd) : a.ntiles;
    if (tile_first >= tile_end) return;
    const int my_tiles = (tile_end - tile_first + gstride - 1) / gstride;
    const int total_units = my_tiles * 8;

    const int n = lane & 15, kq = lane >> 4;
    const float floor_v = a.relu ? 0.f : -INFINITY;
    f32x2 bias2[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            bias2[cb][h] = a.bias ? f32x2{a.bias[cb * 16 + 4 * kq + 2 * h], a.bias[cb * 16 + 4 * kq + 2 * h + 1]} : f32x2{0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(bias2[cb][h]));
    const unsigned st0 = (unsigned)((4 * kq * HW + 2 * n) * 4), st1 = st0 + (unsigned)(W * 4);
    const unsigned stp = (unsigned)((4 * kq * (HW >> 2) + n) * 4);      // (pooled copy: one float per lane and channel)

    // all transformed filters -> LDS, once (nchunks * WQ units, 512 per instruction); the workgroup's unit counter
    for (int j = 0; j < nchunks * WQ / WN_THREADS; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(uintptr_t)(lds0 + (unsigned)(j * 8192 + wave * 1024)), 16,
                                                 (unsigned)((j * WN_THREADS + tid) * 16), 0, 0, 0);
    unsigned* unit_ctr = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(smem) + wbytes + 8 * WC_RING_BYTES);
    if (tid == 0) *unit_ctr = 8u;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    auto next_unit = [&]() {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(unit_ctr, 1u);
        return (int)__builtin_amdgcn_readfirstlane(u);
    };

    // chunk c of row pair `unit` -> slot `slot` (the wave's running chunk count & 1): three DMA instructions, always
    auto dma_chunk = [&](int unit, int c, int slot) {
        const int t = tile_first + (unit >> 3) * gstride;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int y0 = ty * WN_TH + 2 * (unit & 7), x0 = tx * WN_TW;
        const unsigned em = (y0 == 0 ? 1u : 0u) | (y0 + 2 == H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + WN_TW == W ? 8u : 0u);
        // the chunk's source and its first channel there
        int s = 0, ch = c * 4;
        while (s + 1 < a.nsrc && ch >= ((a.x_c[s] + 3) & ~3)) {
            ch -= (a.x_c[s] + 3) & ~3;
            ++s;
        }
        const unsigned nvalid = (unsigned)min(4, a.x_c[s] - ch);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x[s] + (long long)b * a.x_bs[s]) - lead), 0,
            (unsigned)(a.x_c[s] * HW * 4) + lead, 0x00020000);
        const unsigned so = (unsigned)((ch * HW + y0 * W + x0) * 4);
        const unsigned sb = ring0 + (unsigned)(slot * WC_SLOT_BYTES) + 4u;
        if (em == 0 && nvalid == 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)sb, 16, rel[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + 1024u), 16, rel[1], so, 0, 0);
            if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + 2048u), 16, rel[2], so, 0, 0);
        } else {
            const unsigned v0 = ((edge[0] & em) || planeof[0] >= nvalid) ? 0x80000000u : rel[0];
            const unsigned v1 = ((edge[1] & em) || planeof[1] >= nvalid) ? 0x80000000u : rel[1];
            const unsigned v2 = ((edge[2] & em) || planeof[2] >= nvalid) ? 0x80000000u : rel[2];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)sb, 16, v0, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + 1024u), 16, v1, so, 0, 0);
            if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + 2048u), 16, v2, so, 0, 0);
        }
    };

    int cur = wave, nxt = next_unit();
    int g = 0;                              // the wave's running chunk count: chunk g lives in slot g & 1
    dma_chunk(cur, 0, 0);
    dma_chunk(cur, 1, 1);

    f32x4 acc[16][NCB];
    const unsigned char* ringp = reinterpret_cast<const unsigned char*>(smem) + wbytes + wave * WC_RING_BYTES;
    while (cur < total_units) {
        auto step = [&](int c, auto first) {
            // chunk c has landed (the loads issued after it are those of the next chunk, if there is one)
            if (c + 1 < nchunks || nxt < total_units) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int slot = g & 1;
            const float* ip = reinterpret_cast<const float*>(ringp + slot * WC_SLOT_BYTES + 4) + kq * WN_PLANE_F + 3 + 2 * n;
            f32x2 dl[4], dh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dl[r] = *reinterpret_cast<const f32x2*>(ip + r * WN_ROWF);
                dh[r] = *reinterpret_cast<const f32x2*>(ip + r * WN_ROWF + 2);
            }
            // the slot is read out (this wave's own reads): it takes the chunk two ahead
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c + 2 < nchunks) dma_chunk(cur, c + 2, slot);
            else if (nxt < total_units) dma_chunk(nxt, c + 2 - nchunks, slot);
            wino_kstep<NCB, decltype(first)::value>(acc, dl, dh, smem + c * WQ, lane);
            ++g;
        };
        step(0, std::true_type{});          // (the pair's first k-step accumulates onto 0: peeled, no run-time branch around the MFMAs)
        for (int c = 1; c < nchunks; ++c) step(c, std::false_type{});
        {
            const int t = tile_first + (cur >> 3) * gstride;
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
            const unsigned so_t = (unsigned)(((ty * WN_TH + 2 * (cur & 7)) * W + tx * WN_TW) * 4);
            const int ab = a.addend_bmod > 0 ? b % a.addend_bmod : b;
            const unsigned sa_t = EPI == 3 ? (unsigned)(((ty * (WN_TH / 2) + (cur & 7)) * (W >> 1) + tx * (WN_TW / 2)) * 4) : so_t;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)b * a.y_bs, 0, y_img, 0x00020000);
            const __amdgpu_buffer_rsrc_t ra =
                EPI == 3 ? __builtin_amdgcn_make_buffer_rsrc(a.pool + (long long)b * a.pool_bs, 0, y_img >> 2, 0x00020000)
                         : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ADD ? a.addend + (long long)ab * a.addend_bs : a.y), 0, y_img, 0x00020000);
            unsigned* wb = BITS == 1 ? a.wbits + (((long long)b * (H >> 1) + (ty * (WN_TH / 2) + (cur & 7))) * tiles_x + tx) * 64 + lane : nullptr;
            unsigned char* pc = BITS == 3 ? a.pcode + ((long long)b * (16 * NCB) * (HW >> 2) + ((stp + sa_t) >> 2)) : nullptr;      // (the pooled copy's element index)
            wino_epilogue<NCB, EPI, BITS>(acc, bias2, floor_v, ry, ra, st0, st1, so_t, sa_t, HW, stp, wb, pc);
        }
        cur = nxt;
        if (cur < total_units) nxt = next_unit();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The slice form (round 5): 16 output channels per workgroup, TWO row pairs (4 output rows x 32 columns) per wave and unit.  It serves
// what the two kernels above do not: layers with 48 / 64 / 128 output channels and up to 84 (padded) input channels -- the 64-channel
// layers at 64^2 (and 32^2 in evaluate()'s folded batches) --, and it replaces conv_wino_kernel<1, ...> (32 -> 16 at 256^2), whose
// MFMA pipes were 0.49 busy: that form did a whole patch transform, a whole staging step and a filter-fragment read per 16 MFMAs.
//   filters     ONE slice (16 output channels) of the transformed filter is resident per workgroup: 4 KB per chunk of 4 input channels,
//               at most 21 chunks; the slices of a layer are spread over the workgroups of an XCD (workgroup index / 8 modulo the number
//               of slices), which sweep the same tiles -- the input is fetched from HBM once and re-read from that XCD's L2.
//   unit        4 output rows x 32 columns: the two row pairs share two of their six input rows (6 staged rows instead of 8) and every
//               filter fragment read from LDS feeds 8 MFMAs instead of 4; 2 x 16 accumulators of 4 registers, as many as NCB = 2 above.
//   staging     per chunk one LDS-DMA instruction per input channel (60 lanes: 6 rows x 10 units), plane pitch 288 floats = 32 (mod 64)
//               banks: the patch reads of the four channel groups of a wave are conflict-free; two slots per wave, the chunk two ahead
//               issued when a slot's patch is in registers (as conv_wino_cat_kernel).
//   epilogue    wino_epilogue<1, EPI> once per row pair: plain / through a ReLU backward (the activation fetched) / + additive term /
//               + the 2 x 2 max-pooled copy.
#define W6_PLANE_BYTES 1152
#define W6_SLOT_BYTES (4 * W6_PLANE_BYTES + 16)
#define W6_RING_BYTES (2 * W6_SLOT_BYTES)
#define W6_MAX_CHUNKS 21
#define W6_TH 32

struct Wino16Args {
    const float* x[WC_MAX_SRC];
    long long x_bs[WC_MAX_SRC];
    int x_c[WC_MAX_SRC];
    int nsrc, nchunks;         // chunks of 4 channels over all sources (each padded to a multiple of 4)
    const f32x4* u;            // [slice][chunk][4 quads of (xi,nu)][64 lanes] (ynet_winograd16_filter)
    const float* bias;         // 16 * ns floats or NULL
    float* y;                  // [B] x (y_bs floats): 16 * ns planes
    long long y_bs;
    int ns;                    // slices of 16 output channels
    int B, H, W, relu, ntiles;
    const float* aux;          // EPI 1: the post-ReLU activation whose backward is applied to y; EPI 2: the additive term (image b % aux_bmod)
    long long aux_bs;
    int aux_bmod;
    float* pool;               // EPI 3: [B] x (pool_bs floats), 16 * ns planes of (H / 2) x (W / 2)
    long long pool_bs;
};

// one chunk (4 input channels) for both row pairs: V = B^T d B of the two patches (rows 0..3 / 2..5 of the six staged rows), then
// 4 filter-fragment reads feeding 2 x 16 MFMAs
template <bool FIRST>
__device__ __forceinline__ void wino16_kstep(f32x4 (&acc)[2][16][1], const f32x2 (&dl)[6], const f32x2 (&dh)[6], const f32x4* wl, int lane) {
    f32x2 v01[2][4], v23[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 tl[4] = {dl[2 * p] - dl[2 * p + 2], dl[2 * p + 1] + dl[2 * p + 2], dl[2 * p + 2] - dl[2 * p + 1], dl[2 * p + 1] - dl[2 * p + 3]};
        const f32x2 th[4] = {dh[2 * p] - dh[2 * p + 2], dh[2 * p + 1] + dh[2 * p + 2], dh[2 * p + 2] - dh[2 * p + 1], dh[2 * p + 1] - dh[2 * p + 3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v01[p][i] = wn_v01(tl[i], th[i]);
            v23[p][i] = wn_v23(tl[i], th[i]);
        }
    }
    // (inline asm is opaque to hipcc's hazard recognizer: the wait states between a vector write and the MFMA that reads it)
    asm volatile("s_nop 3"
                 : "+v"(v01[0][0]), "+v"(v01[0][1]), "+v"(v01[0][2]), "+v"(v01[0][3]), "+v"(v23[0][0]), "+v"(v23[0][1]), "+v"(v23[0][2]), "+v"(v23[0][3]),
                   "+v"(v01[1][0]), "+v"(v01[1][1]), "+v"(v01[1][2]), "+v"(v01[1][3]), "+v"(v23[1][0]), "+v"(v23[1][1]), "+v"(v23[1][2]), "+v"(v23[1][3]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 w = wl[q * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float bv = e < 2 ? v01[p][q][e] : v23[p][q][e - 2];
                if constexpr (FIRST) acc[p][q * 4 + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else acc[p][q * 4 + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], bv, acc[p][q * 4 + e][0], 0, 0, 0);
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(WN_THREADS, 1) void conv_wino16_kernel(const Wino16Args a) {
    extern __shared__ f32x4 smem[];
    constexpr int WQ = 4 * 64;            // units of one chunk's filters: [4 quads of (xi,nu)][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W, nchunks = a.nchunks;
    const int tiles_x = W / WN_TW, tiles_y = H / W6_TH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned wbytes = (unsigned)(nchunks * WQ * 16);
    const unsigned ring0 = lds0 + wbytes + (unsigned)(wave * W6_RING_BYTES);

    // static DMA geometry: lane l < 60 moves unit (row l / 10, unit l % 10) of a plane
    const int r6 = lane / WN_LQ, xq = lane - r6 * WN_LQ;
    const unsigned rel = (unsigned)((r6 * W + 4 * xq) * 4);
    const unsigned edge = (r6 == 0 ? 1u : 0u) | (r6 == 5 ? 2u : 0u) | (xq == 0 ? 4u : 0u) | (xq == WN_LQ - 1 ? 8u : 0u);
    const unsigned lead = (unsigned)((W + 4) * 4);
    const unsigned y_img = (unsigned)(16 * HW * 4);

    // workgroups of one XCD (blockIdx & 7) share its eighth of the tiles: index / 8 -> (slice, member of the slice's team)
    const int g8 = (int)(gridDim.x >> 3), idx = (int)(blockIdx.x >> 3);
    const int slice = idx % a.ns, gstride = g8 / a.ns;
    const int per_xcd = (a.ntiles + 7) >> 3;
    const int tile_first = (int)(blockIdx.x & 7) * per_xcd + idx / a.ns;
    const int tile_end = min(a.ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd);
    if (tile_first >= tile_end) return;
    const int my_tiles = (tile_end - tile_first + gstride - 1) / gstride;
    const int total_units = my_tiles * 8;

    const int n = lane & 15, kq = lane >> 4;
    const float floor_v = a.relu ? 0.f : -INFINITY;
    f32x2 bias2[1][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        bias2[0][h] = a.bias ? f32x2{a.bias[slice * 16 + 4 * kq + 2 * h], a.bias[slice * 16 + 4 * kq + 2 * h + 1]} : f32x2{0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(bias2[0][h]));
    const unsigned st0 = (unsigned)((4 * kq * HW + 2 * n) * 4), st1 = st0 + (unsigned)(W * 4);
    const unsigned stp = (unsigned)((4 * kq * (HW >> 2) + n) * 4);

    // this slice's transformed filters -> LDS, once; the workgroup's unit counter
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(a.u + (long long)slice * nchunks * WQ), 0, wbytes, 0x00020000);
    for (int j = 0; j * WN_THREADS + wave * 64 < nchunks * WQ; ++j)      // (whole wave instructions: a chunk is 256 units)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(uintptr_t)(lds0 + (unsigned)(j * 8192 + wave * 1024)), 16,
                                                 (unsigned)((j * WN_THREADS + tid) * 16), 0, 0, 0);
    unsigned* unit_ctr = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(smem) + wbytes + 8 * W6_RING_BYTES);
    if (tid == 0) *unit_ctr = 8u;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    auto next_unit = [&]() {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(unit_ctr, 1u);
        return (int)__builtin_amdgcn_readfirstlane(u);
    };

    // chunk c of unit `unit` -> slot: four DMA instructions (one per input channel of the chunk), always
    auto dma_chunk = [&](int unit, int c, int slot) {
        const int t = tile_first + (unit >> 3) * gstride;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int y0 = ty * W6_TH + 4 * (unit & 7), x0 = tx * WN_TW;
        const unsigned em = (y0 == 0 ? 1u : 0u) | (y0 + 4 == H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + WN_TW == W ? 8u : 0u);
        int s = 0, ch = c * 4;
        while (s + 1 < a.nsrc && ch >= ((a.x_c[s] + 3) & ~3)) {
            ch -= (a.x_c[s] + 3) & ~3;
            ++s;
        }
        const int nvalid = min(4, a.x_c[s] - ch);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x[s] + (long long)b * a.x_bs[s]) - lead), 0,
            (unsigned)(a.x_c[s] * HW * 4) + lead, 0x00020000);
        const unsigned so = (unsigned)((ch * HW + y0 * W + x0) * 4);
        const unsigned sb = ring0 + (unsigned)(slot * W6_SLOT_BYTES) + 4u;
        const unsigned v = (edge & em) ? 0x80000000u : rel;
        if (lane < 60) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + (unsigned)(p * W6_PLANE_BYTES)), 16, p < nvalid ? v : 0x80000000u,
                                                         so + (unsigned)(p * HW * 4), 0, 0);
        }
    };

    int cur = wave, nxt = next_unit();
    int g = 0;                              // the wave's running chunk count: chunk g lives in slot g & 1
    dma_chunk(cur, 0, 0);
    dma_chunk(cur, 1, 1);

    f32x4 acc[2][16][1];
    const unsigned char* ringp = reinterpret_cast<const unsigned char*>(smem) + wbytes + wave * W6_RING_BYTES;
    while (cur < total_units) {
        auto step = [&](int c, auto first) {
            if (c + 1 < nchunks || nxt < total_units) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int slot = g & 1;
            const float* ip = reinterpret_cast<const float*>(ringp + slot * W6_SLOT_BYTES + 4 + kq * W6_PLANE_BYTES) + 3 + 2 * n;
            f32x2 dl[6], dh[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                dl[r] = *reinterpret_cast<const f32x2*>(ip + r * WN_ROWF);
                dh[r] = *reinterpret_cast<const f32x2*>(ip + r * WN_ROWF + 2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c + 2 < nchunks) dma_chunk(cur, c + 2, slot);
            else if (nxt < total_units) dma_chunk(nxt, c + 2 - nchunks, slot);
            wino16_kstep<decltype(first)::value>(acc, dl, dh, smem + c * WQ, lane);
            ++g;
        };
        step(0, std::true_type{});
        for (int c = 1; c < nchunks; ++c) step(c, std::false_type{});
        {
            const int t = tile_first + (cur >> 3) * gstride;
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
            const int y0 = ty * W6_TH + 4 * (cur & 7), x0 = tx * WN_TW;
            const int ab = (EPI == 2 && a.aux_bmod > 0) ? b % a.aux_bmod : b;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)b * a.y_bs + (long long)slice * 16 * HW, 0, y_img, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm =
                EPI == 3 ? __builtin_amdgcn_make_buffer_rsrc(a.pool + (long long)b * a.pool_bs + (long long)slice * 16 * (HW >> 2), 0, y_img >> 2, 0x00020000)
                         : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((EPI == 1 || EPI == 2) ? a.aux + (long long)ab * a.aux_bs + (long long)slice * 16 * HW : a.y), 0,
                                                             y_img, 0x00020000);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned so_t = (unsigned)(((y0 + 2 * p) * W + x0) * 4);
                const unsigned sm_t = EPI == 3 ? (unsigned)((((y0 >> 1) + p) * (W >> 1) + (x0 >> 1)) * 4) : so_t;
                wino_epilogue<1, EPI>(acc[p], bias2, floor_v, ry, rm, st0, st1, so_t, sm_t, HW, stp);
            }
        }
        cur = nxt;
        if (cur < total_units) nxt = next_unit();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The up-convolution with its bilinear x2 inside (round 5): y = conv3x3(upsample_bilinear2d(x, 2, align_corners = false)) + bias, the
// decoders' `F.interpolate(x, scale_factor=2, mode='bilinear')` + `upsample_conv[i]` (models/ynet.py:463-464), for 32 -> 16 channels.
// The up-sampled tensor (4x the input) is never written: a wave stages the THREE low-resolution rows under its row pair (8 channels x 3
// rows x 6 units = 144 units per chunk: 2.25 LDS-DMA instructions instead of 5) and every lane builds its 4 x 4 high-resolution patch from
// its 3 x 3 low-resolution patch -- a 2 x 2 output block starts at an odd column / row of the up-sampled image, so the bilinear phase is the
// same for every block: rows (y0 - 1, y0) = (0.75, 0.25) L[i-1] + (0.25, 0.75) L[i], (y0 + 1, y0 + 2) = (0.75, 0.25) L[i] + (0.25, 0.75) L[i+1],
// columns alike.  Borders: the bilinear clamp (L[-1] = L[0]) happens in staging (rows: the lane's offset; columns: a select in
// registers), the convolution's zero padding of the UP-SAMPLED image is applied to the finished patch (row -1 / H, column -1 / W).
// Measured motive (tools/probe_up.py): 32 -> 16 at 256^2 sits under both roofs -- 144 us at B 32, 122 with its input in L2 -- and the
// bilinear pass in front of it takes another 74.
#define WU_LQ 6
#define WU_ROWF 24
#define WU_PLANE_F 72
#define WU_SLOT_BYTES (144 * 16)
#define WU_RING_BYTES (2 * WU_SLOT_BYTES)

struct WinoUpArgs {
    const float* x;        // [B] x (x_bs floats): 8 NCH planes of (H / 2) x (W / 2)
    const f32x4* u;        // transformed filters (ynet_winograd_filter, 16 output channels)
    const float* bias;     // 16 floats or NULL
    float* y;              // [B] x (y_bs floats): 16 planes of H x W
    long long x_bs, y_bs;
    int B, H, W, relu, ntiles;
};

template <int NCH>
__global__ __launch_bounds__(WN_THREADS, 1) void conv_wino_up_kernel(const WinoUpArgs a) {
    extern __shared__ f32x4 smem[];
    constexpr int NCB = 1, WQ = 8 * NCB * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W, Hl = H >> 1, Wl = W >> 1, HWl = Hl * Wl;
    const int tiles_x = W / WN_TW, tiles_y = H / WN_TH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned ring0 = lds0 + (unsigned)(NCH * WQ * 16) + (unsigned)(wave * WU_RING_BYTES);

    // static DMA geometry: unit j * 64 + lane -> (channel of the chunk, low-resolution row 0..2, unit of the row); the third instruction has 16 lanes
    unsigned rel[3], edge[3];             // edge bits: 1 row above, 2 row below, 4 left unit, 8 right unit
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = j * 64 + lane, plane = u / 18, rem = u - plane * 18, r = rem / WU_LQ, xq = rem - r * WU_LQ;
        rel[j] = (unsigned)((plane * HWl + r * Wl + 4 * xq) * 4);
        edge[j] = (r == 0 ? 1u : 0u) | (r == 2 ? 2u : 0u) | (xq == 0 ? 4u : 0u) | (xq == WU_LQ - 1 ? 8u : 0u);
    }
    const unsigned lead = (unsigned)((Wl + 4) * 4);
    const unsigned x_img = (unsigned)(NCH * 8 * HWl * 4) + lead, y_img = (unsigned)(NCB * 16 * HW * 4);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(a.u), 0, (unsigned)(NCH * WQ * 16), 0x00020000);

    const bool xcd_walk = (gridDim.x & 7) == 0 && a.ntiles >= (int)gridDim.x;
    const int per_xcd = (a.ntiles + 7) >> 3;
    const int gstride = xcd_walk ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int tile_first = xcd_walk ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tile_end = xcd_walk ? min(a.ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd) : a.ntiles;
    if (tile_first >= tile_end) return;
    const int my_tiles = (tile_end - tile_first + gstride - 1) / gstride;
    const int total_units = my_tiles * 8;

    const int n = lane & 15, kq = lane >> 4;
    const float floor_v = a.relu ? 0.f : -INFINITY;
    f32x2 bias2[NCB][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bias2[0][h] = a.bias ? f32x2{a.bias[4 * kq + 2 * h], a.bias[4 * kq + 2 * h + 1]} : f32x2{0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(bias2[0][h]));
    const unsigned st0 = (unsigned)((4 * kq * HW + 2 * n) * 4), st1 = st0 + (unsigned)(W * 4);

#pragma unroll
    for (int j = 0; j < (NCH * WQ + WN_THREADS - 1) / WN_THREADS; ++j)
        if (j * WN_THREADS + wave * 64 < NCH * WQ)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(uintptr_t)(lds0 + (unsigned)((j * WN_THREADS + wave * 64) * 16)), 16,
                                                     (unsigned)((j * WN_THREADS + tid) * 16), 0, 0, 0);
    unsigned* unit_ctr = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(smem) + NCH * WQ * 16 + 8 * WU_RING_BYTES);
    if (tid == 0) *unit_ctr = 8u;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    auto next_unit = [&]() {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(unit_ctr, 1u);
        return (int)__builtin_amdgcn_readfirstlane(u);
    };

    auto dma_chunk = [&](int unit, int c) {       // the three low-resolution rows under row pair `unit`, chunk c -> slot c & 1
        const int t = tile_first + (unit >> 3) * gstride, slot = c & 1;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int i = (ty * WN_TH >> 1) + (unit & 7), j0 = tx * (WN_TW / 2);
        const unsigned em = (i == 0 ? 1u : 0u) | (i + 1 == Hl ? 2u : 0u) | (j0 == 0 ? 4u : 0u) | (j0 + WN_TW / 2 == Wl ? 8u : 0u);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x + (long long)b * a.x_bs) - lead), 0, x_img, 0x00020000);
        const unsigned so = (unsigned)((c * 8 * HWl + i * Wl + j0) * 4);
        const unsigned sb = ring0 + (unsigned)(slot * WU_SLOT_BYTES);
        unsigned v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // the row above the image is row 0, the row below it the last row (bilinear clamp); units left / right of the image: zero
            // (their one float that matters, column -1 / W_low, is replaced by the clamp in registers)
            unsigned o = rel[j];
            if (edge[j] & em & 1u) o += (unsigned)(Wl * 4);
            if (edge[j] & em & 2u) o -= (unsigned)(Wl * 4);
            v[j] = (edge[j] & em & 12u) ? 0x80000000u : o;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)sb, 16, v[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + 1024u), 16, v[1], so, 0, 0);
        if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + 2048u), 16, v[2], so, 0, 0);
    };

    int cur = wave, nxt = next_unit();
    dma_chunk(cur, 0);
    dma_chunk(cur, 1);

    const f32x2 c7525 = {0.75f, 0.25f}, c2575 = {0.25f, 0.75f};
    f32x4 acc[16][NCB];
    while (cur < total_units) {
        // this unit's place in the image: which of the patch's border rows / columns lie outside the up-sampled image
        const int t = tile_first + (cur >> 3) * gstride;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int y0 = ty * WN_TH + 2 * (cur & 7), x0 = tx * WN_TW;
        const bool top = y0 == 0, bottom = y0 + 2 == H;
        const bool left = x0 == 0 && n == 0, right = x0 + WN_TW == W && n == 15;      // (per lane)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH || nxt < total_units) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const f32x4* wl = smem + c * WQ;
            const float* il = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(smem) + NCH * WQ * 16 + wave * WU_RING_BYTES + (c & 1) * WU_SLOT_BYTES);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // the lane's 3 x 3 low-resolution patch of channel s * 4 + kq: staged rows 0 .. 2, floats 3 + n .. 5 + n of the row
                const float* ip = il + (s * 4 + kq) * WU_PLANE_F + 3 + n;
                float x[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int m = 0; m < 3; ++m) x[r][m] = ip[r * WU_ROWF + m];
                if (s == 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (c + 2 < NCH) dma_chunk(cur, c + 2);
                    else if (nxt < total_units) dma_chunk(nxt, c + 2 - NCH);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {      // bilinear clamp of the columns left / right of the image
                    x[r][0] = left ? x[r][1] : x[r][0];
                    x[r][2] = right ? x[r][1] : x[r][2];
                }
                // rows: (y0 - 1, y0) and (y0 + 1, y0 + 2) of the up-sampled image at the three low-resolution columns
                f32x2 va[3], vb[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    va[m] = c7525 * x[0][m] + c2575 * x[1][m];
                    vb[m] = c7525 * x[1][m] + c2575 * x[2][m];
                    if (top) va[m][0] = 0.f;          // (the convolution's zero padding: row -1 / row H of the up-sampled image)
                    if (bottom) vb[m][1] = 0.f;
                }
                // columns: dl[R] = up-sampled columns (x0 + 2n - 1, x0 + 2n), dh[R] = (x0 + 2n + 1, x0 + 2n + 2) of patch row R
                f32x2 dl[4], dh[4];
#pragma unroll
                for (int R = 0; R < 4; ++R) {
                    const float v0 = R < 2 ? va[0][R] : vb[0][R - 2], v1 = R < 2 ? va[1][R] : vb[1][R - 2], v2 = R < 2 ? va[2][R] : vb[2][R - 2];
                    dl[R] = c7525 * v0 + c2575 * v1;
                    dh[R] = c7525 * v1 + c2575 * v2;
                    dl[R][0] = left ? 0.f : dl[R][0];       // (column -1 / column W of the up-sampled image)
                    dh[R][1] = right ? 0.f : dh[R][1];
                }
                const f32x2 tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[1] - dl[3]};
                const f32x2 th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[1] - dh[3]};
                f32x2 v01[4], v23[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v01[q] = wn_v01(tl[q], th[q]);
                    v23[q] = wn_v23(tl[q], th[q]);
                }
                asm volatile("s_nop 3" : "+v"(v01[0]), "+v"(v01[1]), "+v"(v01[2]), "+v"(v01[3]), "+v"(v23[0]), "+v"(v23[1]), "+v"(v23[2]), "+v"(v23[3]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = wl[((s * 4 + q) * NCB) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bv = e < 2 ? v01[q][e] : v23[q][e - 2];
                        if (c == 0 && s == 0) acc[q * 4 + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else acc[q * 4 + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], bv, acc[q * 4 + e][0], 0, 0, 0);
                    }
                }
            }
        }
        {
            const unsigned so_t = (unsigned)((y0 * W + x0) * 4);
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)b * a.y_bs, 0, y_img, 0x00020000);
            wino_epilogue<NCB, 0>(acc, bias2, floor_v, ry, ry, st0, st1, so_t, so_t, HW);
        }
        cur = nxt;
        if (cur < total_units) nxt = next_unit();
    }
}

// ... and in the slice form (64 -> 32 at 128^2, 64 -> 32 at 64^2: the decoders' levels 3 and 2): a unit of 4 up-sampled rows sits on FOUR
// low-resolution rows (4 channels x 4 rows x 6 units = 96 units per chunk: 1.5 LDS-DMA instructions instead of 4), the lane's 4 x 3 patch
// becomes the six up-sampled rows of its two 4 x 4 patches.  (The contiguous LDS image puts the four channel groups of a wave 96 floats
// apart: 2-way bank conflicts on its twelve 4-byte reads per chunk -- 24 LDS cycles on 1024 of MFMAs.)
#define WV_SLOT_BYTES (96 * 16)
#define WV_RING_BYTES (2 * WV_SLOT_BYTES)

struct Wino16UpArgs {
    const float* x;        // [B] x (x_bs floats): 4 nchunks planes of (H / 2) x (W / 2)
    const f32x4* u;        // [slice][chunk][4 quads][64 lanes] (ynet_winograd16_filter)
    const float* bias;     // 16 ns floats or NULL
    float* y;              // [B] x (y_bs floats): 16 ns planes of H x W
    long long x_bs, y_bs;
    int nchunks, ns, B, H, W, relu, ntiles;
};

__global__ __launch_bounds__(WN_THREADS, 1) void conv_wino16_up_kernel(const Wino16UpArgs a) {
    extern __shared__ f32x4 smem[];
    constexpr int WQ = 4 * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, HW = H * W, Hl = H >> 1, Wl = W >> 1, HWl = Hl * Wl, nchunks = a.nchunks;
    const int tiles_x = W / WN_TW, tiles_y = H / W6_TH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned wbytes = (unsigned)(nchunks * WQ * 16);
    const unsigned ring0 = lds0 + wbytes + (unsigned)(wave * WV_RING_BYTES);

    // static DMA geometry: unit j * 64 + lane (j = 1: 32 lanes) -> (channel of the chunk, low-resolution row 0..3, unit of the row)
    unsigned rel[2], edge[2];             // edge bits: 1 row above, 2 row below, 4 left unit, 8 right unit
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = j * 64 + lane, plane = u / 24, rem = u - plane * 24, r = rem / WU_LQ, xq = rem - r * WU_LQ;
        rel[j] = (unsigned)((plane * HWl + r * Wl + 4 * xq) * 4);
        edge[j] = (r == 0 ? 1u : 0u) | (r == 3 ? 2u : 0u) | (xq == 0 ? 4u : 0u) | (xq == WU_LQ - 1 ? 8u : 0u);
    }
    const unsigned lead = (unsigned)((Wl + 4) * 4);
    const unsigned x_img = (unsigned)(nchunks * 4 * HWl * 4) + lead, y_img = (unsigned)(16 * HW * 4);

    const int g8 = (int)(gridDim.x >> 3), idx = (int)(blockIdx.x >> 3);
    const int slice = idx % a.ns, gstride = g8 / a.ns;
    const int per_xcd = (a.ntiles + 7) >> 3;
    const int tile_first = (int)(blockIdx.x & 7) * per_xcd + idx / a.ns;
    const int tile_end = min(a.ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd);
    if (tile_first >= tile_end) return;
    const int my_tiles = (tile_end - tile_first + gstride - 1) / gstride;
    const int total_units = my_tiles * 8;

    const int n = lane & 15, kq = lane >> 4;
    const float floor_v = a.relu ? 0.f : -INFINITY;
    f32x2 bias2[1][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        bias2[0][h] = a.bias ? f32x2{a.bias[slice * 16 + 4 * kq + 2 * h], a.bias[slice * 16 + 4 * kq + 2 * h + 1]} : f32x2{0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(bias2[0][h]));
    const unsigned st0 = (unsigned)((4 * kq * HW + 2 * n) * 4), st1 = st0 + (unsigned)(W * 4);

    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(a.u + (long long)slice * nchunks * WQ), 0, wbytes, 0x00020000);
    for (int j = 0; j * WN_THREADS + wave * 64 < nchunks * WQ; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(uintptr_t)(lds0 + (unsigned)(j * 8192 + wave * 1024)), 16,
                                                 (unsigned)((j * WN_THREADS + tid) * 16), 0, 0, 0);
    unsigned* unit_ctr = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(smem) + wbytes + 8 * WV_RING_BYTES);
    if (tid == 0) *unit_ctr = 8u;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    auto next_unit = [&]() {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(unit_ctr, 1u);
        return (int)__builtin_amdgcn_readfirstlane(u);
    };

    auto dma_chunk = [&](int unit, int c, int slot) {       // the four low-resolution rows under unit `unit`, chunk c (4 channels) -> slot
        const int t = tile_first + (unit >> 3) * gstride;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int i = (ty * W6_TH >> 1) + 2 * (unit & 7), j0 = tx * (WN_TW / 2);
        const unsigned em = (i == 0 ? 1u : 0u) | (i + 2 == Hl ? 2u : 0u) | (j0 == 0 ? 4u : 0u) | (j0 + WN_TW / 2 == Wl ? 8u : 0u);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.x + (long long)b * a.x_bs) - lead), 0, x_img, 0x00020000);
        const unsigned so = (unsigned)((c * 4 * HWl + i * Wl + j0) * 4);
        const unsigned sb = ring0 + (unsigned)(slot * WV_SLOT_BYTES);
        unsigned v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned o = rel[j];
            if (edge[j] & em & 1u) o += (unsigned)(Wl * 4);       // (bilinear clamp: the row above the image is row 0, the row below it the last row)
            if (edge[j] & em & 2u) o -= (unsigned)(Wl * 4);
            v[j] = (edge[j] & em & 12u) ? 0x80000000u : o;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)sb, 16, v[0], so, 0, 0);
        if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(uintptr_t)(sb + 1024u), 16, v[1], so, 0, 0);
    };

    int cur = wave, nxt = next_unit();
    int g = 0;
    dma_chunk(cur, 0, 0);
    dma_chunk(cur, 1, 1);

    const f32x2 c7525 = {0.75f, 0.25f}, c2575 = {0.25f, 0.75f};
    f32x4 acc[2][16][1];
    const unsigned char* ringp = reinterpret_cast<const unsigned char*>(smem) + wbytes + wave * WV_RING_BYTES;
    while (cur < total_units) {
        const int t = tile_first + (cur >> 3) * gstride;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int y0 = ty * W6_TH + 4 * (cur & 7), x0 = tx * WN_TW;
        const bool top = y0 == 0, bottom = y0 + 4 == H;
        const bool left = x0 == 0 && n == 0, right = x0 + WN_TW == W && n == 15;
        auto step = [&](int c, auto first) {
            if (c + 1 < nchunks || nxt < total_units) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int slot = g & 1;
            const float* ip = reinterpret_cast<const float*>(ringp + slot * WV_SLOT_BYTES) + kq * 96 + 3 + n;
            float x[4][3];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 3; ++m) x[r][m] = ip[r * WU_ROWF + m];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c + 2 < nchunks) dma_chunk(cur, c + 2, slot);
            else if (nxt < total_units) dma_chunk(nxt, c + 2 - nchunks, slot);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[r][0] = left ? x[r][1] : x[r][0];
                x[r][2] = right ? x[r][1] : x[r][2];
            }
            f32x2 pv[3][3];      // pv[k][m]: up-sampled rows (y0 - 1 + 2k, y0 + 2k) at low-resolution column m
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int m = 0; m < 3; ++m) pv[k][m] = c7525 * x[k][m] + c2575 * x[k + 1][m];
            if (top) {
#pragma unroll
                for (int m = 0; m < 3; ++m) pv[0][m][0] = 0.f;
            }
            if (bottom) {
#pragma unroll
                for (int m = 0; m < 3; ++m) pv[2][m][1] = 0.f;
            }
            f32x2 dl[6], dh[6];
#pragma unroll
            for (int R = 0; R < 6; ++R) {
                const float v0 = pv[R >> 1][0][R & 1], v1 = pv[R >> 1][1][R & 1], v2 = pv[R >> 1][2][R & 1];
                dl[R] = c7525 * v0 + c2575 * v1;
                dh[R] = c7525 * v1 + c2575 * v2;
                dl[R][0] = left ? 0.f : dl[R][0];
                dh[R][1] = right ? 0.f : dh[R][1];
            }
            wino16_kstep<decltype(first)::value>(acc, dl, dh, smem + c * WQ, lane);
            ++g;
        };
        step(0, std::true_type{});
        for (int c = 1; c < nchunks; ++c) step(c, std::false_type{});
        {
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)b * a.y_bs + (long long)slice * 16 * HW, 0, y_img, 0x00020000);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned so_t = (unsigned)(((y0 + 2 * p) * W + x0) * 4);
                wino_epilogue<1, 0>(acc[p], bias2, floor_v, ry, ry, st0, st1, so_t, so_t, HW);
            }
        }
        cur = nxt;
        if (cur < total_units) nxt = next_unit();
    }
}

// U = G g G^T of every (cout, cin) pair, in the fragment order the kernels read: unit ((c4 * 4 + q) * NCB + cb) * 64 + lane holds
// (xi = q, nu = 0..3) of output channel col0 + cb * 16 + (lane & 15), PADDED input channel c4 * 4 + (lane >> 4) -- the input channels
// are the concatenation of up to three sources, each padded to a multiple of 4 (padded channels: zero filters); with one source of a
// multiple-of-4 channel count the padded channel is the channel.
// wp: a packed filter of ynet_pack_weight, [k][tap][m] with m padded to cols_pad (k = the conv's input channels, m = its outputs).
// nch_slice > 0: the slice-major order of conv_wino16_kernel instead -- unit ((cb * nch_slice + c4) * 4 + q) * 64 + lane, every 16-channel
// slice of the output contiguous.
__global__ void wino_filter_kernel(const float* __restrict__ wp, f32x4* __restrict__ u, int cols_pad, int col0, int ncb, int nunits, int c0, int c1, int c2,
                                   int nch_slice) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nunits) return;
    const int l = i & 63;
    const int cb = nch_slice > 0 ? (i >> 8) / nch_slice : (i >> 6) % ncb;
    const int q = nch_slice > 0 ? (i >> 6) & 3 : ((i >> 6) / ncb) & 3;
    const int c4 = nch_slice > 0 ? (i >> 8) % nch_slice : ((i >> 6) / ncb) >> 2;
    const int co = col0 + cb * 16 + (l & 15);
    // padded channel -> row of the packed filter, or none
    int cp = c4 * 4 + (l >> 4), row = -1, base = 0;
    const int cs[3] = {c0, c1, c2};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int pad = (cs[s] + 3) & ~3;
        if (row < 0 && cp >= 0 && cp < pad) {
            row = cp < cs[s] ? base + cp : -2;
            cp = -1;
        } else if (cp >= 0) {
            cp -= pad;
        }
        base += cs[s];
    }
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = row >= 0 ? wp[((long long)row * 9 + t) * cols_pad + co] : 0.f;
    // row q of G g (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]), then times G^T
    float gr[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        gr[j] = q == 0 ? g[0][j] : (q == 3 ? g[2][j] : 0.5f * ((g[0][j] + g[2][j]) + (q == 1 ? g[1][j] : -g[1][j])));
    f32x4 o;
    o[0] = gr[0];
    o[1] = 0.5f * ((gr[0] + gr[2]) + gr[1]);
    o[2] = 0.5f * ((gr[0] + gr[2]) - gr[1]);
    o[3] = gr[2];
    u[i] = o;
}

// The packed filter's column padding (conv_mfma.hip: YNET_COUT_PAD)
static int wino_cols_pad(int cols) { return ceil_div(cols, 64) * 64; }

static bool wino_shape_ok(int B, int H, int W, int cin, int cout, int K) {
    static const int on = getenv("YNET_WINOGRAD") ? atoi(getenv("YNET_WINOGRAD")) : 1;
    if (!on || K != 3 || B <= 0) return false;
    if (H % WN_TH || W % WN_TW || H < WN_TH || W < WN_TW) return false;
    static const int on48 = getenv("YNET_WINOGRAD48") ? atoi(getenv("YNET_WINOGRAD48")) : 1;      // (48 outputs: three blocks per wave, one wave per SIMD -- the two-destination data gradient)
    if ((cin != 16 && cin != 32) || (cout != 16 && cout != 32 && !(cout == 48 && on48))) return false;
    if (48ll * H * W * 4 + (W + 4) * 4 >= (1ll << 31)) return false;      // one image per buffer descriptor, below 2 GB (the batch is unbounded)
    static const int min_pixels = getenv("YNET_WINOGRAD_MIN") ? atoi(getenv("YNET_WINOGRAD_MIN")) : 128 * 128 * 8;      // (round 4: 16 -> 8 -- batch 10 at 128^2 is 320 tiles: 3.15 -> 3.00 ms per step there, nothing lost at batch 32)
    return (long long)B * H * W >= min_pixels;      // (one workgroup of eight waves per CU: small launches stay with the direct tiles)
}

template <int NCB, int NCH, int EM, int NW>
static int launch_wino_nw(WinoArgs& a, hipStream_t st) {
    // (EM 7: + the predictor tables, the loss scratch, the blob table and the positions of the target planes -- wino_pred_lds_bytes)
    const int lds = NCH * 8 * NCB * 64 * 16 + NW * WN_RING_BYTES + 16 +
                    (EM == 7 ? 5 * 64 * 16 + NW * 8 + 16 + (a.t_m * a.t_m + 2 * a.B * a.pco) * 4
                             : (EM == 8 ? 10 * 64 * 16 + NW * 8 + 16 + a.t_m * a.t_m * 4 + ((2 * a.B * a.pco * 2 + 15) & ~15) : 0));
    static bool attr_dev[YNET_MAX_DEV] = {false};
    static int cus_dev[YNET_MAX_DEV] = {0};
    const int slot = ynet_device_slot();
    if (!attr_dev[slot]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<NCB, NCH, EM, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (EM == 7 || EM == 8) ? 160 * 1024 : lds);
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus_dev[slot] = cus < 8 ? 8 : cus;
        attr_dev[slot] = true;
    }
    int grid = a.ntiles < cus_dev[slot] ? a.ntiles : cus_dev[slot];
    if (grid >= 8) grid &= ~7;          // (the XCD-aware walk wants a multiple of 8)
    hipLaunchKernelGGL((conv_wino_kernel<NCB, NCH, EM, NW>), dim3(grid), dim3(NW * 64), lds, st, a);
    return ynet_check_launch("conv2d_winograd");
}

// 16 output channels per workgroup leave 64 accumulator registers per wave: twelve waves (three per SIMD) fit -- measured SLOWER than eight
// (32 -> 16 @ 256^2, B 32: 123.5 against 130.5 TFLOP/s direct-equivalent, the captured C2 step 7.73 against 7.65 ms): the 16-channel form does
// the same patch reads, input transform and staging per k-step for half the MFMAs, and a third wave adds to that contention, not to the
// pipes' work.  Left as a switch (YNET_WINOGRAD_W12=1).
template <int NCB, int NCH, int EM>
static int launch_wino(WinoArgs& a, hipStream_t st) {
    static const int w12 = getenv("YNET_WINOGRAD_W12") ? atoi(getenv("YNET_WINOGRAD_W12")) : 0;
    if constexpr (NCB == 1) {
        if (w12) return launch_wino_nw<NCB, NCH, EM, 12>(a, st);
    }
    return launch_wino_nw<NCB, NCH, EM, 8>(a, st);
}

extern "C" {

int ynet_conv2d_winograd_supported(int B, int H, int W, int cin, int cout, int K) { return wino_shape_ok(B, H, W, cin, cout, K) ? 1 : 0; }

long long ynet_winograd_filter_floats(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return 16ll * (ceil_div(cin, 8) * 8) * (ceil_div(cout, 16) * 16);
}

int ynet_winograd_filter(const float* wp, float* u, int cin, int cout, int col0, int cols_total, void* stream) {
    YNET_REQUIRE(wp && u, "winograd_filter: null pointer");
    YNET_REQUIRE(cin > 0 && cin % 8 == 0 && cout > 0 && cout % 16 == 0, "winograd_filter: cin %d must be a multiple of 8, cout %d of 16", cin, cout);
    YNET_REQUIRE(col0 >= 0 && cols_total >= col0 + cout, "winograd_filter: output channels %d .. %d are not inside the filter's %d", col0, col0 + cout, cols_total);
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(u) & 15) == 0, "winograd_filter: the output must be 16-byte aligned");
    const int ncb = cout / 16, nunits = (cin / 4) * 4 * ncb * 64;
    hipLaunchKernelGGL(wino_filter_kernel, dim3(ceil_div(nunits, 256)), dim3(256), 0, (hipStream_t)stream, wp, reinterpret_cast<f32x4*>(u),
                       wino_cols_pad(cols_total), col0, ncb, nunits, cin, 0, 0, 0);
    return ynet_check_launch("winograd_filter");
}

// padded input channels of a concatenation: every source rounded up to a multiple of 4
static int wino_cat_padded(const int* src_c, int nsrc) {
    int n = 0;
    for (int i = 0; i < nsrc; ++i) n += (src_c[i] + 3) & ~3;
    return n;
}

static bool wino_cat_ok(int B, int H, int W, const int* src_c, int nsrc, int cout, int K) {
    static const int on = getenv("YNET_WINOGRAD") ? atoi(getenv("YNET_WINOGRAD")) : 1;
    static const int cat_on = getenv("YNET_WINOGRAD_CAT") ? atoi(getenv("YNET_WINOGRAD_CAT")) : 1;
    if (!on || !cat_on || K != 3 || B <= 0 || nsrc < 1 || nsrc > WC_MAX_SRC || cout != 32) return false;
    if (H % WN_TH || W % WN_TW || H < WN_TH || W < WN_TW) return false;
    for (int i = 0; i < nsrc; ++i)
        if (src_c[i] <= 0) return false;
    const int nch = wino_cat_padded(src_c, nsrc) / 4;
    if (56ll * H * W * 4 + (W + 4) * 4 >= (1ll << 31)) return false;      // one image of one source per buffer descriptor, below 2 GB
    if (nch < 2 || nch > 14) return false;          // 14 chunks of 8 KB of filters + eight 5 KB rings: 156 KB of LDS
    static const int min_pixels = getenv("YNET_WINOGRAD_MIN") ? atoi(getenv("YNET_WINOGRAD_MIN")) : 128 * 128 * 8;      // (round 4: 16 -> 8 -- batch 10 at 128^2 is 320 tiles: 3.15 -> 3.00 ms per step there, nothing lost at batch 32)
    return (long long)B * H * W >= min_pixels;
}

int ynet_conv2d_winograd_cat_supported(int B, int H, int W, const int* src_c, int nsrc, int cout, int K) {
    return (src_c != nullptr && wino_cat_ok(B, H, W, src_c, nsrc, cout, K)) ? 1 : 0;
}

long long ynet_winograd_filter_cat_floats(const int* src_c, int nsrc, int cout) {
    if (src_c == nullptr || nsrc < 1 || nsrc > WC_MAX_SRC || cout <= 0) return 0;
    return 16ll * wino_cat_padded(src_c, nsrc) * (ceil_div(cout, 16) * 16);
}

int ynet_winograd_filter_cat(const float* wp, float* u, const int* src_c, int nsrc, int cout, int col0, int cols_total, void* stream) {
    YNET_REQUIRE(wp && u && src_c, "winograd_filter_cat: null pointer");
    YNET_REQUIRE(nsrc >= 1 && nsrc <= WC_MAX_SRC && cout > 0 && cout % 16 == 0, "winograd_filter_cat: 1..%d sources, cout %d a multiple of 16", WC_MAX_SRC, cout);
    YNET_REQUIRE(col0 >= 0 && cols_total >= col0 + cout, "winograd_filter_cat: output channels %d .. %d are not inside the filter's %d", col0, col0 + cout, cols_total);
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(u) & 15) == 0, "winograd_filter_cat: the output must be 16-byte aligned");
    for (int i = 0; i < nsrc; ++i) YNET_REQUIRE(src_c[i] > 0, "winograd_filter_cat: source %d has no channels", i);
    const int ncb = cout / 16, nunits = wino_cat_padded(src_c, nsrc) * ncb * 64;
    hipLaunchKernelGGL(wino_filter_kernel, dim3(ceil_div(nunits, 256)), dim3(256), 0, (hipStream_t)stream, wp, reinterpret_cast<f32x4*>(u),
                       wino_cols_pad(cols_total), col0, ncb, nunits, src_c[0], nsrc > 1 ? src_c[1] : 0, nsrc > 2 ? src_c[2] : 0, 0);
    return ynet_check_launch("winograd_filter_cat");
}

static int wino_cat_launch(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                           long long dst_bs, int cout, int B, int H, int W, int relu, const float* addend, long long addend_bs, int addend_bmod, float* pool,
                           long long pool_bs, void* stream, const char* what, unsigned* wbits = nullptr, unsigned char* pcode = nullptr) {
    YNET_REQUIRE(src && src_c && src_bs && u && dst, "%s: null pointer", what);
    YNET_REQUIRE(wino_cat_ok(B, H, W, src_c, nsrc, cout, 3), "%s: shape B=%d %dx%d -> %d with %d sources is not served (ask ynet_conv2d_winograd_cat_supported)", what,
                 B, H, W, cout, nsrc);
    const long long HW = (long long)H * W;
    WinoCatArgs a{};
    for (int i = 0; i < nsrc; ++i) {
        YNET_REQUIRE(src[i] != nullptr && (reinterpret_cast<uintptr_t>(src[i]) & 15) == 0 && (src_bs[i] & 3) == 0 && (src_bs[i] == 0 || src_bs[i] >= src_c[i] * HW),
                     "%s: source %d must be 16-byte aligned with a batch stride of 0 (one image for the batch) or not smaller than its image", what, i);
        a.x[i] = src[i];
        a.x_bs[i] = src_bs[i];
        a.x_c[i] = src_c[i];
    }
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 7) == 0 && (reinterpret_cast<uintptr_t>(u) & 15) == 0 && (dst_bs & 1) == 0 && dst_bs >= cout * HW,
                 "%s: the output must be 8-byte aligned, its batch stride not smaller than the image", what);
    if (addend != nullptr) {
        YNET_REQUIRE(addend_bmod >= 0 && (reinterpret_cast<uintptr_t>(addend) & 7) == 0 && (addend_bs & 1) == 0 && addend_bs >= cout * HW,
                     "%s: the additive term must be 8-byte aligned, its image stride not smaller than the image, its modulus not negative", what);
    }
    if (pool != nullptr)
        YNET_REQUIRE(addend == nullptr && (reinterpret_cast<uintptr_t>(pool) & 3) == 0 && pool_bs >= cout * (HW / 4),
                     "%s: the pooled copy must have a batch stride not smaller than its image, and excludes an additive term", what);
    if (wbits != nullptr)
        YNET_REQUIRE(pool == nullptr && (reinterpret_cast<uintptr_t>(wbits) & 3) == 0, "%s: the 1-bit mask excludes the pooled copy and must be 4-byte aligned", what);
    if (pcode != nullptr) YNET_REQUIRE(pool != nullptr && cout == 32 && relu, "%s: the arg-max / ReLU bytes come with the pooled copy of a 32-channel ReLU output", what);
    a.wbits = wbits;
    a.pool = pool;
    a.pool_bs = pool_bs;
    a.pcode = pcode;
    a.nsrc = nsrc;
    a.nchunks = wino_cat_padded(src_c, nsrc) / 4;
    a.u = reinterpret_cast<const f32x4*>(u);
    a.bias = bias;
    a.y = dst;
    a.y_bs = dst_bs;
    a.B = B; a.H = H; a.W = W; a.relu = relu ? 1 : 0;
    a.ntiles = B * (H / WN_TH) * (W / WN_TW);
    a.addend = addend;
    a.addend_bs = addend_bs;
    a.addend_bmod = addend_bmod;
    const int lds = a.nchunks * 8192 + 8 * WC_RING_BYTES + 16;
    static bool attr_dev[YNET_MAX_DEV] = {false};
    static int cus_dev[YNET_MAX_DEV] = {0};
    const int slot = ynet_device_slot();
    if (!attr_dev[slot]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cat_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 14 * 8192 + 8 * WC_RING_BYTES + 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cat_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 14 * 8192 + 8 * WC_RING_BYTES + 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cat_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 14 * 8192 + 8 * WC_RING_BYTES + 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cat_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 14 * 8192 + 8 * WC_RING_BYTES + 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cat_kernel<2, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 14 * 8192 + 8 * WC_RING_BYTES + 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cat_kernel<2, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 14 * 8192 + 8 * WC_RING_BYTES + 16);
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus_dev[slot] = cus < 8 ? 8 : cus;
        attr_dev[slot] = true;
    }
    int grid = a.ntiles < cus_dev[slot] ? a.ntiles : cus_dev[slot];
    if (grid >= 8) grid &= ~7;
    if (wbits != nullptr && addend != nullptr) hipLaunchKernelGGL((conv_wino_cat_kernel<2, 5>), dim3(grid), dim3(WN_THREADS), lds, (hipStream_t)stream, a);
    else if (wbits != nullptr) hipLaunchKernelGGL((conv_wino_cat_kernel<2, 4>), dim3(grid), dim3(WN_THREADS), lds, (hipStream_t)stream, a);
    else if (addend != nullptr) hipLaunchKernelGGL((conv_wino_cat_kernel<2, 2>), dim3(grid), dim3(WN_THREADS), lds, (hipStream_t)stream, a);
    else if (pcode != nullptr) hipLaunchKernelGGL((conv_wino_cat_kernel<2, 6>), dim3(grid), dim3(WN_THREADS), lds, (hipStream_t)stream, a);
    else if (pool != nullptr) hipLaunchKernelGGL((conv_wino_cat_kernel<2, 3>), dim3(grid), dim3(WN_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_wino_cat_kernel<2, 0>), dim3(grid), dim3(WN_THREADS), lds, (hipStream_t)stream, a);
    return ynet_check_launch(what);
}

int ynet_conv2d_winograd_cat(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                             long long dst_bs, int cout, int B, int H, int W, int relu, void* stream) {
    return wino_cat_launch(src, src_c, src_bs, nsrc, u, bias, dst, dst_bs, cout, B, H, W, relu, nullptr, 0, 0, nullptr, 0, stream, "conv2d_winograd_cat");
}

int ynet_conv2d_winograd_cat_add(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                 long long dst_bs, int cout, int B, int H, int W, int relu, const float* addend, long long addend_bs, int addend_bmod,
                                 void* stream) {
    YNET_REQUIRE(addend != nullptr, "conv2d_winograd_cat_add: the additive term is null");
    return wino_cat_launch(src, src_c, src_bs, nsrc, u, bias, dst, dst_bs, cout, B, H, W, relu, addend, addend_bs, addend_bmod, nullptr, 0, stream, "conv2d_winograd_cat_add");
}

int ynet_conv2d_winograd_cat_pool(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                  long long dst_bs, float* pooled, long long pooled_bs, int cout, int B, int H, int W, int relu, void* stream) {
    YNET_REQUIRE(pooled != nullptr, "conv2d_winograd_cat_pool: the pooled output is null");
    return wino_cat_launch(src, src_c, src_bs, nsrc, u, bias, dst, dst_bs, cout, B, H, W, relu, nullptr, 0, 0, pooled, pooled_bs, stream, "conv2d_winograd_cat_pool");
}

int ynet_conv2d_winograd_cat_pool_code(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                       long long dst_bs, float* pooled, long long pooled_bs, unsigned char* code, int B, int H, int W, void* stream) {
    YNET_REQUIRE(pooled != nullptr && code != nullptr, "conv2d_winograd_cat_pool_code: the pooled output or its code plane is null");
    return wino_cat_launch(src, src_c, src_bs, nsrc, u, bias, dst, dst_bs, 32, B, H, W, 1, nullptr, 0, 0, pooled, pooled_bs, stream, "conv2d_winograd_cat_pool_code",
                           nullptr, code);
}

static int wino_launch_any(const float* src, long long src_bs, const float* u, const float* bias, float* dst, long long dst_bs, const float* emask,
                           long long emask_bs, int cin, int cout, int B, int H, int W, int relu, void* stream, const char* what, unsigned* wbits = nullptr,
                           bool wbits_apply = false, bool s2d = false, float* dst2 = nullptr, long long dst2_bs = 0, int split_mode = 0) {
    YNET_REQUIRE(src && u && dst, "%s: null pointer", what);
    YNET_REQUIRE(wino_shape_ok(B, H, W, cin, cout, 3), "%s: shape B=%d %dx%d %d -> %d is not served (ask ynet_conv2d_winograd_supported)", what, B, H, W, cin,
                 cout);
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0 && (reinterpret_cast<uintptr_t>(u) & 15) == 0 &&
                     (src_bs & 3) == 0 && (dst_bs & 1) == 0,
                 "%s: planes must be 16-byte (input, filters) / 8-byte (output) aligned", what);
    const long long HW = (long long)H * W;
    YNET_REQUIRE((src_bs == 0 || src_bs >= cin * HW) && dst_bs >= (split_mode ? 16 : cout) * HW, "%s: batch strides smaller than the images (input: 0 = one image for the batch)", what);
    if (split_mode)
        YNET_REQUIRE(cout == 48 && dst2 != nullptr && (reinterpret_cast<uintptr_t>(dst2) & 7) == 0 && (dst2_bs & 1) == 0 && dst2_bs >= 32 * HW && bias == nullptr && !relu &&
                         wbits == nullptr && !s2d,
                     "%s: two destinations are for the plain 48-channel data gradient (16 + 32 planes, 8-byte aligned)", what);
    if (emask != nullptr)
        YNET_REQUIRE((reinterpret_cast<uintptr_t>(emask) & 7) == 0 && (emask_bs & 1) == 0 && emask_bs >= cout * HW,
                     "%s: the activation must be 8-byte aligned, its batch stride not smaller than the image", what);
    WinoArgs a{src, reinterpret_cast<const f32x4*>(u), bias, dst, src_bs, dst_bs, B, H, W, relu ? 1 : 0, B * (H / WN_TH) * (W / WN_TW), emask, emask_bs, wbits, dst2, dst2_bs};
    hipStream_t st = (hipStream_t)stream;
    if (wbits != nullptr) {          // the 1-bit mask: 32 output channels only (one word per lane and unit)
        YNET_REQUIRE(cout == 32 && emask == nullptr && (reinterpret_cast<uintptr_t>(wbits) & 3) == 0, "%s: the 1-bit mask serves 32 output channels and excludes the float mask", what);
        if (wbits_apply) return cin == 32 ? launch_wino<2, 4, 2>(a, st) : launch_wino<2, 2, 2>(a, st);
        return cin == 32 ? launch_wino<2, 4, 3>(a, st) : launch_wino<2, 2, 3>(a, st);
    }
    if (s2d) {                       // the output stored space-to-depth: [4 cout][H / 2][W / 2] per image (a data gradient on its way through an up-convolution)
        YNET_REQUIRE(emask == nullptr && bias == nullptr && !relu, "%s: the space-to-depth store is for a plain data gradient", what);
        if (cout == 32) return cin == 32 ? launch_wino<2, 4, 4>(a, st) : launch_wino<2, 2, 4>(a, st);
        return cin == 32 ? launch_wino<1, 4, 4>(a, st) : launch_wino<1, 2, 4>(a, st);
    }
    if (cout == 48) {                // three output blocks per wave: 192 accumulator registers, FOUR waves per workgroup (one per SIMD, 512 registers each)
        YNET_REQUIRE(emask == nullptr && cin == 32, "%s: the 48-channel form is plain, from 32 input channels", what);
        if (split_mode == 1) return launch_wino_nw<3, 4, 5, 4>(a, st);
        if (split_mode == 2) return launch_wino_nw<3, 4, 6, 4>(a, st);
        return launch_wino_nw<3, 4, 0, 4>(a, st);
    }
    if (emask != nullptr) {
        if (cout == 32) return cin == 32 ? launch_wino<2, 4, 1>(a, st) : launch_wino<2, 2, 1>(a, st);
        return cin == 32 ? launch_wino<1, 4, 1>(a, st) : launch_wino<1, 2, 1>(a, st);
    }
    if (cout == 32) return cin == 32 ? launch_wino<2, 4, 0>(a, st) : launch_wino<2, 2, 0>(a, st);
    return cin == 32 ? launch_wino<1, 4, 0>(a, st) : launch_wino<1, 2, 0>(a, st);
}

int ynet_conv2d_winograd(const float* src, long long src_bs, const float* u, const float* bias, float* dst, long long dst_bs, int cin, int cout,
                         int B, int H, int W, int relu, void* stream) {
    return wino_launch_any(src, src_bs, u, bias, dst, dst_bs, nullptr, 0, cin, cout, B, H, W, relu, stream, "conv2d_winograd");
}

int ynet_conv2d_winograd_s2d(const float* src, long long src_bs, const float* u, float* dst, long long dst_bs, int cin, int cout, int B, int H, int W, void* stream) {
    return wino_launch_any(src, src_bs, u, nullptr, dst, dst_bs, nullptr, 0, cin, cout, B, H, W, 0, stream, "conv2d_winograd_s2d", nullptr, false, true);
}

// (the blob table and two ints per target plane live in LDS next to the filters and the staging rings: kernlen^2 + 2 B pred_cout words in what the 160 KB leave)
int ynet_conv2d_winograd_pred_bce_supported(int B, int H, int W, int cin, int cout, int pred_cout, int kernlen) {
    static const int on = getenv("YNET_CONV_PRED_BCE") ? atoi(getenv("YNET_CONV_PRED_BCE")) : 1;
    if (!(on && cin == 32 && cout == 32 && pred_cout >= 1 && pred_cout <= 32 && kernlen >= 1 && wino_shape_ok(B, H, W, cin, cout, 3))) return 0;
    const long long base = 4ll * 8 * 2 * 64 * 16 + 8ll * WN_RING_BYTES + 16 + 8 * 8 + 16;
    const long long lds = pred_cout <= 16 ? base + 5 * 64 * 16 + ((long long)kernlen * kernlen + 2ll * B * pred_cout) * 4      // one block of tables, the blob in LDS
                                          : base + 10 * 64 * 16 + (long long)kernlen * kernlen * 4 + ((2ll * B * pred_cout * 2 + 15) & ~15ll);      // two blocks, 16-bit positions
    return lds <= 160 * 1024 ? 1 : 0;
}

int ynet_conv2d_winograd_pred_bce_blob(const float* src, long long src_bs, const float* u, const float* bias, const float* pred_wp, const float* pred_bias, int pred_cout,
                                       const float* target_xy, const float* blob, int kernlen, int S, float* logits, float* loss, float* dx, long long dx_bs,
                                       void* workspace, int B, int H, int W, float expected_grad, void* stream) {
    YNET_REQUIRE(src && u && pred_wp && target_xy && blob && logits && loss && dx && workspace, "conv2d_winograd_pred_bce_blob: null pointer");
    YNET_REQUIRE(ynet_conv2d_winograd_pred_bce_supported(B, H, W, 32, 32, pred_cout, kernlen), "conv2d_winograd_pred_bce_blob: B=%d %dx%d with %d predictor outputs and a %d x %d blob is not served (32 -> 32, <= 32 outputs, tables within LDS; ask ..._supported)",
                 B, H, W, pred_cout, kernlen, kernlen);
    YNET_REQUIRE(kernlen > 0 && kernlen <= S && S >= H && S >= W, "conv2d_winograd_pred_bce_blob: the target needs a blob table with 0 < kernlen <= S and S >= H, W (got kernlen %d, S %d, %dx%d)", kernlen, S, H, W);
    const long long HW = (long long)H * W;
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(u) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(logits) & 7) == 0 && (src_bs & 3) == 0 && (dx_bs & 1) == 0 && (src_bs == 0 || src_bs >= 32 * HW) && dx_bs >= 32 * HW,
                 "conv2d_winograd_pred_bce_blob: planes must be 16-byte (input, filters) / 8-byte (outputs) aligned, batch strides not smaller than the images");
    WinoArgs a{src, reinterpret_cast<const f32x4*>(u), bias, dx, src_bs, dx_bs, B, H, W, 1, B * (H / WN_TH) * (W / WN_TW), nullptr, 0, nullptr, nullptr, 0};
    a.pw = pred_wp;
    a.pb = pred_bias;
    a.pco = pred_cout;
    a.pco_pad = ceil_div(pred_cout, 64) * 64;
    a.t_xy = target_xy;
    a.t_blob = blob;
    a.t_m = kernlen;
    a.t_S = S;
    a.logits = logits;
    a.partial = (double*)workspace;
    a.ticket = (unsigned*)((char*)workspace + 1024 * sizeof(double));      // (the layout of ynet_pred_bce_workspace_bytes(): 1024 partials, then the ticket)
    a.loss = loss;
    a.n_loss = (long long)B * pred_cout * HW;
    a.gs = expected_grad / (float)a.n_loss;
    return pred_cout <= 16 ? launch_wino_nw<2, 4, 7, 8>(a, (hipStream_t)stream) : launch_wino_nw<2, 4, 8, 8>(a, (hipStream_t)stream);
}

int ynet_conv2d_winograd_split_supported(int B, int H, int W, int cin) { return (cin == 32 && wino_shape_ok(B, H, W, cin, 48, 3)) ? 1 : 0; }

int ynet_conv2d_winograd_split(const float* src, long long src_bs, const float* u, float* dst0, long long dst0_bs, int dst0_s2d, float* dst1, long long dst1_bs, int cin,
                               int B, int H, int W, void* stream) {
    return wino_launch_any(src, src_bs, u, nullptr, dst0, dst0_bs, nullptr, 0, cin, 48, B, H, W, 0, stream, "conv2d_winograd_split", nullptr, false, false, dst1, dst1_bs,
                           dst0_s2d ? 2 : 1);
}

// ---- the Winograd-native 1-bit ReLU mask (round 5): one 32-bit word per lane and unit of the NCB = 2 tiling, i.e. per (image, row pair, 32-column tile, lane)
long long ynet_winograd_relu_bits_words(int B, int H, int W) {
    if (B <= 0 || H % WN_TH || W % WN_TW) return 0;
    return (long long)B * (H / 2) * (W / WN_TW) * 64;
}

int ynet_conv2d_winograd_relu_bits(const float* src, long long src_bs, const float* u, const float* bias, float* dst, long long dst_bs, int cin, int B, int H,
                                   int W, unsigned* bits_out, void* stream) {
    YNET_REQUIRE(bits_out != nullptr, "conv2d_winograd_relu_bits: the mask output is null");
    return wino_launch_any(src, src_bs, u, bias, dst, dst_bs, nullptr, 0, cin, 32, B, H, W, 1, stream, "conv2d_winograd_relu_bits", bits_out, false);
}

int ynet_conv2d_winograd_cat_relu_bits(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                       long long dst_bs, int B, int H, int W, const float* addend, long long addend_bs, int addend_bmod, unsigned* bits_out,
                                       void* stream) {
    YNET_REQUIRE(bits_out != nullptr, "conv2d_winograd_cat_relu_bits: the mask output is null");
    return wino_cat_launch(src, src_c, src_bs, nsrc, u, bias, dst, dst_bs, 32, B, H, W, 1, addend, addend_bs, addend_bmod, nullptr, 0, stream,
                           "conv2d_winograd_cat_relu_bits", bits_out);
}

int ynet_conv2d_winograd_dgrad_relu_bits(const float* dy, long long dy_bs, const float* u, float* dx, long long dx_bs, const unsigned* bits, int dy_c, int B, int H,
                                         int W, void* stream) {
    YNET_REQUIRE(bits != nullptr, "conv2d_winograd_dgrad_relu_bits: the mask is null");
    return wino_launch_any(dy, dy_bs, u, nullptr, dx, dx_bs, nullptr, 0, dy_c, 32, B, H, W, 0, stream, "conv2d_winograd_dgrad_relu_bits", const_cast<unsigned*>(bits), true);
}

int ynet_conv2d_winograd_dgrad_relu(const float* dy, long long dy_bs, const float* u, float* dx, long long dx_bs, const float* relu_of, long long relu_of_bs,
                                    int dy_c, int dx_c, int B, int H, int W, void* stream) {
    YNET_REQUIRE(relu_of != nullptr, "conv2d_winograd_dgrad_relu: the activation whose ReLU backward is applied is null");
    return wino_launch_any(dy, dy_bs, u, nullptr, dx, dx_bs, relu_of, relu_of_bs, dy_c, dx_c, B, H, W, 0, stream, "conv2d_winograd_dgrad_relu");
}

}  // extern "C"

// ---- the slice form (conv_wino16_kernel)
static bool wino16_ok(int B, int H, int W, const int* src_c, int nsrc, int cout, int K) {
    static const int on = getenv("YNET_WINOGRAD") ? atoi(getenv("YNET_WINOGRAD")) : 1;
    static const int on16 = getenv("YNET_WINOGRAD16") ? atoi(getenv("YNET_WINOGRAD16")) : 1;
    if (!on || !on16 || K != 3 || B <= 0 || nsrc < 1 || nsrc > WC_MAX_SRC) return false;
    if (cout != 16 && cout != 32 && cout != 64 && cout != 128) return false;      // 1 / 2 / 4 / 8 slices: a divisor of the 32 workgroups of an XCD
    if (H % W6_TH || W % WN_TW || H < W6_TH || W < WN_TW) return false;
    for (int i = 0; i < nsrc; ++i)
        if (src_c[i] <= 0) return false;
    const int nch = wino_cat_padded(src_c, nsrc) / 4;
    if (nch < 2 || nch > W6_MAX_CHUNKS) return false;      // 21 chunks of 4 KB of filters + eight 9.2 KB rings: 158 KB of LDS
    if (84ll * H * W * 4 + (W + 4) * 4 >= (1ll << 31)) return false;      // one image of one source / of the output per buffer descriptor
    // (40 K pixels: batch 10 at 64^2 is served -- 3.02 -> 2.92 ms per step there --, batch 32 at 32^2 is not: one 8-unit tile per workgroup on
    //  half the CUs takes 30-37 us per launch where the direct tiles take 27-30, and the 96 / 97-channel layers would be two launches)
    static const int min_pixels = getenv("YNET_WINOGRAD16_MIN") ? atoi(getenv("YNET_WINOGRAD16_MIN")) : (getenv("YNET_WINOGRAD_MIN") ? atoi(getenv("YNET_WINOGRAD_MIN")) : 64 * 64 * 10);
    return (long long)B * H * W >= min_pixels;
}

extern "C" {

int ynet_conv2d_winograd16_supported(int B, int H, int W, const int* src_c, int nsrc, int cout, int K) {
    return (src_c != nullptr && wino16_ok(B, H, W, src_c, nsrc, cout, K)) ? 1 : 0;
}

long long ynet_winograd16_filter_floats(const int* src_c, int nsrc, int cout) {
    if (src_c == nullptr || nsrc < 1 || nsrc > WC_MAX_SRC || cout <= 0) return 0;
    return 16ll * wino_cat_padded(src_c, nsrc) * (ceil_div(cout, 16) * 16);
}

int ynet_winograd16_filter(const float* wp, float* u, const int* src_c, int nsrc, int cout, int col0, int cols_total, void* stream) {
    YNET_REQUIRE(wp && u && src_c, "winograd16_filter: null pointer");
    YNET_REQUIRE(nsrc >= 1 && nsrc <= WC_MAX_SRC && cout > 0 && cout % 16 == 0, "winograd16_filter: 1..%d sources, cout %d a multiple of 16", WC_MAX_SRC, cout);
    YNET_REQUIRE(col0 >= 0 && cols_total >= col0 + cout, "winograd16_filter: output channels %d .. %d are not inside the filter's %d", col0, col0 + cout, cols_total);
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(u) & 15) == 0, "winograd16_filter: the output must be 16-byte aligned");
    for (int i = 0; i < nsrc; ++i) YNET_REQUIRE(src_c[i] > 0, "winograd16_filter: source %d has no channels", i);
    const int ncb = cout / 16, nch = wino_cat_padded(src_c, nsrc) / 4, nunits = nch * 4 * ncb * 64;
    hipLaunchKernelGGL(wino_filter_kernel, dim3(ceil_div(nunits, 256)), dim3(256), 0, (hipStream_t)stream, wp, reinterpret_cast<f32x4*>(u),
                       wino_cols_pad(cols_total), col0, ncb, nunits, src_c[0], nsrc > 1 ? src_c[1] : 0, nsrc > 2 ? src_c[2] : 0, nch);
    return ynet_check_launch("winograd16_filter");
}

int ynet_conv2d_winograd16(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                           long long dst_bs, int cout, int B, int H, int W, int relu, const float* relu_of, long long relu_of_bs, const float* addend,
                           long long addend_bs, int addend_bmod, float* pooled, long long pooled_bs, void* stream) {
    const char* what = "conv2d_winograd16";
    YNET_REQUIRE(src && src_c && src_bs && u && dst, "%s: null pointer", what);
    YNET_REQUIRE(wino16_ok(B, H, W, src_c, nsrc, cout, 3), "%s: shape B=%d %dx%d -> %d with %d sources is not served (ask ynet_conv2d_winograd16_supported)", what,
                 B, H, W, cout, nsrc);
    YNET_REQUIRE((relu_of != nullptr) + (addend != nullptr) + (pooled != nullptr) <= 1, "%s: at most one of relu_of / addend / pooled", what);
    const long long HW = (long long)H * W;
    Wino16Args a{};
    for (int i = 0; i < nsrc; ++i) {
        YNET_REQUIRE(src[i] != nullptr && (reinterpret_cast<uintptr_t>(src[i]) & 15) == 0 && (src_bs[i] & 3) == 0 && (src_bs[i] == 0 || src_bs[i] >= src_c[i] * HW),
                     "%s: source %d must be 16-byte aligned with a batch stride of 0 (one image for the batch) or not smaller than its image", what, i);
        a.x[i] = src[i];
        a.x_bs[i] = src_bs[i];
        a.x_c[i] = src_c[i];
    }
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 7) == 0 && (reinterpret_cast<uintptr_t>(u) & 15) == 0 && (dst_bs & 1) == 0 && dst_bs >= cout * HW,
                 "%s: the output must be 8-byte aligned, its batch stride not smaller than the image", what);
    if (relu_of != nullptr) {
        YNET_REQUIRE(bias == nullptr && !relu && (reinterpret_cast<uintptr_t>(relu_of) & 7) == 0 && (relu_of_bs & 1) == 0 && relu_of_bs >= cout * HW,
                     "%s: relu_of is for a data gradient (no bias, no ReLU); 8-byte aligned, batch stride not smaller than the image", what);
        a.aux = relu_of;
        a.aux_bs = relu_of_bs;
    }
    if (addend != nullptr) {
        YNET_REQUIRE(addend_bmod >= 0 && (reinterpret_cast<uintptr_t>(addend) & 7) == 0 && (addend_bs & 1) == 0 && addend_bs >= cout * HW,
                     "%s: the additive term must be 8-byte aligned, its image stride not smaller than the image, its modulus not negative", what);
        a.aux = addend;
        a.aux_bs = addend_bs;
        a.aux_bmod = addend_bmod;
    }
    if (pooled != nullptr)
        YNET_REQUIRE((reinterpret_cast<uintptr_t>(pooled) & 3) == 0 && pooled_bs >= cout * (HW / 4), "%s: the pooled copy's batch stride is smaller than its image", what);
    a.pool = pooled;
    a.pool_bs = pooled_bs;
    a.nsrc = nsrc;
    a.nchunks = wino_cat_padded(src_c, nsrc) / 4;
    a.u = reinterpret_cast<const f32x4*>(u);
    a.bias = bias;
    a.y = dst;
    a.y_bs = dst_bs;
    a.ns = cout / 16;
    a.B = B; a.H = H; a.W = W; a.relu = relu ? 1 : 0;
    a.ntiles = B * (H / W6_TH) * (W / WN_TW);
    const int lds = a.nchunks * 4096 + 8 * W6_RING_BYTES + 16;
    constexpr int lds_max = W6_MAX_CHUNKS * 4096 + 8 * W6_RING_BYTES + 16;
    static bool attr_dev[YNET_MAX_DEV] = {false};
    static int cus_dev[YNET_MAX_DEV] = {0};
    const int slot = ynet_device_slot();
    if (!attr_dev[slot]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus_dev[slot] = cus < 64 ? 64 : cus;
        attr_dev[slot] = true;
    }
    // one workgroup per CU; a slice's team inside an XCD is as large as that XCD has tiles for
    const int per_xcd = (a.ntiles + 7) / 8;
    int team = (cus_dev[slot] / 8) / a.ns;
    if (team > per_xcd) team = per_xcd;
    if (team < 1) team = 1;
    const int grid = 8 * a.ns * team;
    hipStream_t st = (hipStream_t)stream;
    if (relu_of != nullptr) hipLaunchKernelGGL((conv_wino16_kernel<1>), dim3(grid), dim3(WN_THREADS), lds, st, a);
    else if (addend != nullptr) hipLaunchKernelGGL((conv_wino16_kernel<2>), dim3(grid), dim3(WN_THREADS), lds, st, a);
    else if (pooled != nullptr) hipLaunchKernelGGL((conv_wino16_kernel<3>), dim3(grid), dim3(WN_THREADS), lds, st, a);
    else hipLaunchKernelGGL((conv_wino16_kernel<0>), dim3(grid), dim3(WN_THREADS), lds, st, a);
    return ynet_check_launch(what);
}

}  // extern "C"

// ---- the up-convolution with its bilinear x2 inside (conv_wino_up_kernel)
// (the slice form of it: cin a multiple of 4 up to 84, 16 / 32 / 64 outputs, 32-row tiles)
static bool wino16_up_ok(int B, int H, int W, int cin, int cout, int K) {
    static const int on = getenv("YNET_WINOGRAD") ? atoi(getenv("YNET_WINOGRAD")) : 1;
    static const int up_on = getenv("YNET_WINOGRAD_UP") ? atoi(getenv("YNET_WINOGRAD_UP")) : 1;
    static const int on16 = getenv("YNET_WINOGRAD16") ? atoi(getenv("YNET_WINOGRAD16")) : 1;
    if (!on || !up_on || !on16 || K != 3 || B <= 0 || cin < 8 || cin % 4 || cin / 4 > W6_MAX_CHUNKS) return false;
    if (cout != 16 && cout != 32 && cout != 64) return false;
    if (H % W6_TH || W % WN_TW || H < W6_TH || W < WN_TW) return false;
    if (84ll * H * W * 4 + (W + 4) * 4 >= (1ll << 31)) return false;
    static const int min_pixels = getenv("YNET_WINOGRAD16_MIN") ? atoi(getenv("YNET_WINOGRAD16_MIN")) : (getenv("YNET_WINOGRAD_MIN") ? atoi(getenv("YNET_WINOGRAD_MIN")) : 64 * 64 * 10);
    return (long long)B * H * W >= min_pixels;
}

static bool wino_up_ok(int B, int H, int W, int cin, int cout, int K) {
    static const int on = getenv("YNET_WINOGRAD") ? atoi(getenv("YNET_WINOGRAD")) : 1;
    static const int up_on = getenv("YNET_WINOGRAD_UP") ? atoi(getenv("YNET_WINOGRAD_UP")) : 1;
    if (!on || !up_on || K != 3 || B <= 0 || cin != 32 || cout != 16) return false;
    if (H % WN_TH || W % WN_TW || H < WN_TH || W < WN_TW) return false;      // (H, W: the UP-SAMPLED size; the input is H / 2 x W / 2)
    if (32ll * H * W * 4 + (W + 4) * 4 >= (1ll << 31)) return false;
    static const int min_pixels = getenv("YNET_WINOGRAD_MIN") ? atoi(getenv("YNET_WINOGRAD_MIN")) : 128 * 128 * 8;
    return (long long)B * H * W >= min_pixels;
}

extern "C" {

// 0: not served; 1: served, filter in ynet_winograd_filter's layout (32 -> 16); 2: served by the slice form, filter in ynet_winograd16_filter's layout
int ynet_upsample2x_conv2d_winograd_supported(int B, int H, int W, int cin, int cout, int K) {
    return wino_up_ok(B, H, W, cin, cout, K) ? 1 : (wino16_up_ok(B, H, W, cin, cout, K) ? 2 : 0);
}

int ynet_upsample2x_conv2d_winograd(const float* src, long long src_bs, const float* u, const float* bias, float* dst, long long dst_bs, int cin, int cout, int B,
                                    int H, int W, int relu, void* stream) {
    const char* what = "upsample2x_conv2d_winograd";
    YNET_REQUIRE(src && u && dst, "%s: null pointer", what);
    const bool plain = wino_up_ok(B, H, W, cin, cout, 3);
    YNET_REQUIRE(plain || wino16_up_ok(B, H, W, cin, cout, 3), "%s: shape B=%d %dx%d (up-sampled) %d -> %d is not served (ask ynet_upsample2x_conv2d_winograd_supported)",
                 what, B, H, W, cin, cout);
    const long long HW = (long long)H * W;
    YNET_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0 && (reinterpret_cast<uintptr_t>(u) & 15) == 0 &&
                     (src_bs & 3) == 0 && (dst_bs & 1) == 0,
                 "%s: planes must be 16-byte (input, filters) / 8-byte (output) aligned", what);
    YNET_REQUIRE((src_bs == 0 || src_bs >= cin * (HW / 4)) && dst_bs >= cout * HW, "%s: batch strides smaller than the images (input: 0 = one image for the batch)", what);
    static bool attr_dev[YNET_MAX_DEV] = {false};
    static int cus_dev[YNET_MAX_DEV] = {0};
    const int slot = ynet_device_slot();
    constexpr int lds = 4 * 8 * 64 * 16 + 8 * WU_RING_BYTES + 16;
    if (!attr_dev[slot]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_up_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino16_up_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  W6_MAX_CHUNKS * 4096 + 8 * WV_RING_BYTES + 16);
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus_dev[slot] = cus < 8 ? 8 : cus;
        attr_dev[slot] = true;
    }
    if (!plain) {      // the slice form: one slice of the filter per workgroup, the slices of a tile inside one XCD (as ynet_conv2d_winograd16)
        Wino16UpArgs w{src, reinterpret_cast<const f32x4*>(u), bias, dst, src_bs, dst_bs, cin / 4, cout / 16, B, H, W, relu ? 1 : 0, B * (H / W6_TH) * (W / WN_TW)};
        const int per_xcd = (w.ntiles + 7) / 8;
        int team = ((cus_dev[slot] < 64 ? 64 : cus_dev[slot]) / 8) / w.ns;
        if (team > per_xcd) team = per_xcd;
        if (team < 1) team = 1;
        hipLaunchKernelGGL(conv_wino16_up_kernel, dim3(8 * w.ns * team), dim3(WN_THREADS), w.nchunks * 4096 + 8 * WV_RING_BYTES + 16, (hipStream_t)stream, w);
        return ynet_check_launch(what);
    }
    WinoUpArgs a{src, reinterpret_cast<const f32x4*>(u), bias, dst, src_bs, dst_bs, B, H, W, relu ? 1 : 0, B * (H / WN_TH) * (W / WN_TW)};
    int grid = a.ntiles < cus_dev[slot] ? a.ntiles : cus_dev[slot];
    if (grid >= 8) grid &= ~7;
    hipLaunchKernelGGL((conv_wino_up_kernel<4>), dim3(grid), dim3(WN_THREADS), lds, (hipStream_t)stream, a);
    return ynet_check_launch(what);
}

}  // extern "C"
