/* libynet_hip.so — C ABI of the MI355X (gfx950) Y-Net forward/backward kernels.
 *
 * The reference (vita-epfl/motion-style-transfer) is pure Python on stock PyTorch and has no FFI
 * of its own: the seam is the nn.Module surface of models/ynet.py and the ATen ops it dispatches
 * (SURVEY.md section 8b).  Each entry point below therefore replaces one ATen op (or a fused group)
 * at the reference call site cited next to it.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned memory (PyTorch allocations); fp32, NCHW,
 *    contiguous inside one image plane; batch strides are explicit where concatenation is fused;
 *  - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); calls only enqueue
 *    work: no allocation, no synchronisation, no host<->device copies;
 *  - process state the library DOES keep (and nothing else): (1) per HIP device, set on first use -- the > 64 KB dynamic-LDS
 *    attribute of its kernels and the CU count its persistent grids are sized for; (2) the DEVELOPMENT switches YNET_* read from
 *    the environment ONCE, at the first call that consults them (function-local statics in csrc/conv_mfma.hip, conv_wino.hip,
 *    wgrad_mfma.hip, lora_wgrad.hip: dispatch overrides such as YNET_WINOGRAD, YNET_WINOGRAD16, YNET_WINOGRAD_MIN, YNET_CONV_R,
 *    YNET_KSPLIT_ITEMS -- the full table is DESIGN.md section 9).  A process that changes one of them after that first call keeps the
 *    old value: set them before loading the library; the defaults are what is tested and measured, no entry point's result depends
 *    on them beyond fp32 rounding (which kernel serves a shape), and the *_supported / *_plan queries report the dispatch in force;
 *  - return 0 on success, non-zero on error; ynet_last_error() returns the message (thread local);
 *  - results are bitwise reproducible run to run (no float atomics anywhere).
 */
#ifndef YNET_HIP_H
#define YNET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int ynet_abi_version(void);
const char* ynet_last_error(void);

/* ---- convolution ----------------------------------------------------------------------------
 * Filters are consumed in a packed layout [cin_pad16 + 16][K*K][cout_pad64] built by ynet_pack_weight
 * from the checkpoint layout [Cout][Cin][K][K] (models/ynet.py state-dict contract).
 *   mode 0: forward filter.   mode 1: data-gradient filter (taps flipped, cin/cout swapped).
 */
long long ynet_packed_weight_floats(int cout, int cin, int K, int mode);
int ynet_pack_weight(const float* w, float* wp, int cout, int cin, int K, int mode, void* stream);

/* nn.Conv2d(K in {1,3,5}, stride 1, padding K/2) [+ nn.ReLU] with the channel concatenation of its
 * input fused in — replaces conv2d at models/ynet.py:150,192-211,420-451,464,467,469 and the
 * torch.cat calls at models/ynet.py:387,466,574, utils/train_epoch.py:103-104,
 * utils/evaluate.py:259-260.  The same entry point computes the data gradient
 * (convolution_backward -> grad_input) when handed a mode-1 filter, the incoming gradient as the
 * source, `mask` = the post-ReLU activation of the layer (gradient kept where mask > 0) and one
 * destination per concatenated input (NULL = not wanted).
 *   src[i]:   nsrc (1..4) sources, src_c[i] channels, batch stride src_bs[i] elements (0 = broadcast);
 *             src_bmod (may be NULL): src_bmod[i] > 0 means source i holds that many images which repeat along
 *             the batch (image b % src_bmod[i]) — the K goal samples of utils/evaluate.py:248-266 share the
 *             encoder features of their trajectory without replicating them
 *   dst[i]:   ndst (1..4) destinations, dst_c[i] channels, batch stride dst_bs[i]
 *   cin = sum(src_c), cout = sum(dst_c); wp packed for (cout, cin); bias [cout] or NULL
 *   workspace: optional scratch of ynet_conv2d_workspace_floats(B, H, W, cout) floats (may be NULL / 0):
 *   lets small-map launches (8^2 .. 32^2) split their input-channel loop over more workgroups.
 */
long long ynet_conv2d_workspace_floats(int B, int H, int W, int cout);
int ynet_conv2d(const float* const* src, const int* src_c, const long long* src_bs, const int* src_bmod, int nsrc,
                const float* mask, long long mask_bs, const float* wp, const float* bias,
                float* const* dst, const int* dst_c, const long long* dst_bs, int ndst,
                int B, int H, int W, int K, int relu, float* workspace, long long workspace_floats,
                void* stream);
/* ynet_conv2d with ONE destination plus its 2 x 2 max-pooled copy written by the same epilogue: the last convolution of an encoder
 * stage and the nn.MaxPool2d(2, 2) that opens the next one (models/ynet.py:202,215) without the stand-alone pass over y.
 * pooled [B][cout][H/2][W/2] (batch stride pooled_bs), the first-maximum / NaN rule of ynet_maxpool2_fwd; H, W even.
 * ynet_conv2d_pool_supported: 1 for the shapes that take it (3x3, W % 4 == 0, maps large enough for two-row tiles); the call fails
 * elsewhere, callers then run ynet_conv2d + ynet_maxpool2_fwd. */
int ynet_conv2d_pool_supported(int B, int H, int W, int cout, int K);
int ynet_conv2d_pool(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* wp, const float* bias,
                     float* dst, int cout, long long dst_bs, float* pooled, long long pooled_bs,
                     int B, int H, int W, int K, int relu, void* stream);
/* ynet_conv2d as the data gradient of a layer whose INPUT was another layer's post-ReLU output (the conv -> ReLU -> conv chains
 * of models/ynet.py:192-211,420-451): dx = relu_of > 0 ? conv(dy [kept where mask > 0; mask may be NULL], wp) : 0, i.e. the ReLU
 * backward of the layer below is applied where its gradient is produced (the activation tile is fetched under the last MFMA
 * chunk of each output tile), and that layer's own dgrad / wgrad then run without a mask operand.  One source, one destination;
 * wp packed in mode 1; relu_of [B][dx_c][H][W] with batch stride relu_of_bs, 16-byte aligned.  workspace as for ynet_conv2d.
 * ynet_conv2d_dgrad_relu_supported: 1 where the mask is applied inside the convolution kernel (3x3, W % 4 == 0, maps large enough
 * for two-row tiles); elsewhere the call is still correct but costs a pass over dx, and callers keep the consumer-side mask. */
int ynet_conv2d_dgrad_relu_supported(int B, int H, int W, int dx_c, int K);
int ynet_conv2d_dgrad_relu(const float* dy, int dy_c, long long dy_bs, const float* mask, long long mask_bs, const float* wp,
                           float* dx, int dx_c, long long dx_bs, const float* relu_of, long long relu_of_bs,
                           int B, int H, int W, int K, float* workspace, long long workspace_floats, void* stream);
/* ynet_conv2d with one destination, no mask, plus a precomputed additive term: y = [relu](conv(cat(src...), wp) + bias +
 * addend[b % addend_bmod]), addend [images][cout][H][W] with batch stride addend_bs (addend_bmod = 0: one image per batch
 * item).  Convolution is linear in its input channels: the part over inputs that REPEAT along the batch -- the encoder
 * features that the K goal samples of a trajectory share in utils/evaluate.py:248-266 -- is computed once per trajectory
 * and added here, instead of K times inside the channel loop (dec.4.0 of the trajectory decoder: 32 of its 50 input
 * channels).  Served by the large-map 3x3 kernels only: ask ynet_conv2d_add_supported(B, H, W, cout, K) first. */
int ynet_conv2d_add_supported(int B, int H, int W, int cout, int K);
int ynet_conv2d_add(const float* const* src, const int* src_c, const long long* src_bs, const int* src_bmod, int nsrc,
                    const float* wp, const float* bias, float* dst, int cout, long long dst_bs,
                    int B, int H, int W, int K, int relu, const float* addend, long long addend_bs, int addend_bmod,
                    void* stream);

/* The 1-bit form of a ReLU mask between two convolutions (conv -> ReLU -> conv, models/ynet.py:196,206,421-445): the forward
 * convolution of the FIRST layer writes, next to its post-ReLU output y [B][cout][H][W], one bit per element (y > 0) in the register
 * layout of its own tiles -- ynet_conv2d_relu_bits_words(B, H, W, cout, K) uint32 words, 0 when the shape is not served --, and
 * the data gradient of the SECOND layer, whose result dx has that very shape and therefore the same tiling, applies it to what it
 * writes (dx = bit ? conv(dy) : 0): ynet_conv2d_dgrad_relu with 1/32 of the mask bytes and no wait for activation quads in the
 * epilogue.  Replaces the same ATen calls as ynet_conv2d / ynet_conv2d_dgrad_relu (conv2d + relu; convolution_backward +
 * threshold_backward). */
long long ynet_conv2d_relu_bits_words(int B, int H, int W, int cout, int K);
int ynet_conv2d_relu_bits(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* wp, const float* bias,
                          float* dst, int cout, long long dst_bs, unsigned* bits, int B, int H, int W, int K, void* stream);
int ynet_conv2d_dgrad_relu_bits(const float* dy, int dy_c, long long dy_bs, const float* mask, long long mask_bs, const float* wp,
                                float* dx, int dx_c, long long dx_bs, const unsigned* bits, int B, int H, int W, int K, void* stream);

/* The Winograd F(2x2, 3x3) generation of the 3x3 convolution (csrc/conv_wino.hip; round 4): 2.25x fewer fp32 MFMAs than the implicit
 * GEMM behind ynet_conv2d, fp32 throughout -- results differ from ynet_conv2d's by rounding only (the error against fp64 is smaller:
 * fewer additions reach an accumulator).  One source, one destination, no mask / epilogue variant; serves
 *   K = 3, cin in {16, 32}, cout in {16, 32}, H % 16 == 0, W % 32 == 0, B * H * W >= 128 * 128 * 8, an image below 2 GB (ynet_conv2d_winograd_supported;
 *   every tensor is addressed one image per buffer descriptor: any batch size)
 * i.e. the plain large-map convolutions and data gradients of both decoders (models/ynet.py:196,206: Conv2d(3x3) + ReLU; their
 * convolution_backward -> grad_input).  Replaces the same ATen calls as ynet_conv2d.
 *   ynet_winograd_filter        u = G g G^T of every (cout, cin) pair in MFMA fragment order, from a packed filter of ynet_pack_weight
 *                               (mode 0 for the forward convolution, mode 1 for the data gradient: cin / cout are the CONVOLUTION's
 *                               input / output channels either way), for the output channels [col0, col0 + cout) of a filter with
 *                               cols_total of them -- a convolution with 48 or 64 outputs runs as two launches over channel slices;
 *                               ynet_winograd_filter_floats(cin, cout) floats, 16-byte aligned; once per weight version.
 *   ynet_conv2d_winograd        dst[b][co] = [relu](conv3x3(src[b], filter) + bias[co]); src / dst: cin / cout planes of H x W per image,
 *                               batch strides in floats (>= the image; an input stride of 0 = one image for the whole batch), 16- / 8-byte
 *                               aligned; bias may be NULL. */
int ynet_conv2d_winograd_supported(int B, int H, int W, int cin, int cout, int K);
long long ynet_winograd_filter_floats(int cin, int cout);
int ynet_winograd_filter(const float* wp, float* u, int cin, int cout, int col0, int cols_total, void* stream);
int ynet_conv2d_winograd(const float* src, long long src_bs, const float* u, const float* bias, float* dst, long long dst_bs, int cin, int cout,
                         int B, int H, int W, int relu, void* stream);
/*   ynet_conv2d_winograd_s2d    (round 6) the plain data gradient with its output stored SPACE-TO-DEPTH: dst [4 cout][H / 2][W / 2] per image, element (2 i + r, 2 j + c)
 *                               of channel ch at plane (2 r + c) * cout + ch, position (i, j) -- a lane of this tiling holds exactly one such 2 x 2 block.  It is the layout
 *                               in which the gradient of an up-convolution's OUTPUT is consumed without the up-sampled tensor (see ynet_upconv_dgrad_ring). */
int ynet_conv2d_winograd_s2d(const float* src, long long src_bs, const float* u, float* dst, long long dst_bs, int cin, int cout, int B, int H, int W, void* stream);
/*   ynet_conv2d_winograd_split  (round 6) the plain data gradient of a convolution whose input was the concatenation [16 channels, 32 channels]
 *                               (models/ynet.py:466: cat(up-sampled features, skip features)) in ONE launch that reads and transforms dy once: u = the 48-column
 *                               transformed filter (ynet_winograd_filter(wp, u, cin, 48, col0, cols_total)); output channels 0..15 go to dst0 (row-major, or
 *                               space-to-depth as ynet_conv2d_winograd_s2d when dst0_s2d != 0), channels 16..47 to dst1.  Three output blocks per wave = 192
 *                               accumulator registers: four waves per workgroup, one per SIMD.  cin = 32; bit-identical to the two launches it replaces. */
/*   ynet_conv2d_winograd_pred_bce_blob  (round 6) the LAST decoder convolution with everything behind it in its epilogue:
 *                               y = relu(conv3x3(src, filter) + bias)  (32 -> 32, never written),  logits = pred_bias + pred_w y  (the 1 x 1 predictor, <= 32 outputs,
 *                               models/ynet.py:450-451,469),  loss = mean BCE-with-logits(logits, target)  (utils/train_epoch.py:93-94,105-106; the target in the blob form
 *                               of ynet_pred_bce_blob),  dx = y > 0 ? pred_w^T (sigmoid(logits) - target) expected_grad / n : 0  (the gradient of the loss with respect
 *                               to the convolution's PRE-activation output: what its data gradient consumes).  Replaces [ynet_conv2d_winograd -> ynet_pred_bce_blob]: the
 *                               32 activation planes are neither written nor read back; the predictor products run on the matrix cores.  u = ynet_winograd_filter(...,
 *                               32, 32, ...); pred_wp = ynet_pack_weight(1 x 1 filter, mode 0); workspace as ynet_pred_bce_workspace_bytes() (ticket zero before the
 *                               first launch); loss partials are summed in a fixed order (bitwise reproducible). */
int ynet_conv2d_winograd_pred_bce_supported(int B, int H, int W, int cin, int cout, int pred_cout, int kernlen);
int ynet_conv2d_winograd_pred_bce_blob(const float* src, long long src_bs, const float* u, const float* bias, const float* pred_wp, const float* pred_bias, int pred_cout,
                                       const float* target_xy, const float* blob, int kernlen, int S, float* logits, float* loss, float* dx, long long dx_bs,
                                       void* workspace, int B, int H, int W, float expected_grad, void* stream);
int ynet_conv2d_winograd_split_supported(int B, int H, int W, int cin);
int ynet_conv2d_winograd_split(const float* src, long long src_bs, const float* u, float* dst0, long long dst0_bs, int dst0_s2d, float* dst1, long long dst1_bs, int cin,
                               int B, int H, int W, void* stream);
/*   ynet_conv2d_winograd_dgrad_relu   the data gradient written THROUGH the ReLU backward of the layer below, as ynet_conv2d_dgrad_relu:
 *                               dx = relu_of > 0 ? conv3x3(dy, mode-1 filter) : 0, relu_of = that layer's post-ReLU output [B][dx_c][H][W]
 *                               (8-byte aligned), read by the epilogue at the addresses it stores to. */
int ynet_conv2d_winograd_dgrad_relu(const float* dy, long long dy_bs, const float* u, float* dx, long long dx_bs, const float* relu_of, long long relu_of_bs,
                                    int dy_c, int dx_c, int B, int H, int W, void* stream);
/*   ynet_conv2d_winograd_cat    the same convolution over the (virtual) concatenation of up to three sources -- the decoders' first convolutions,
 *                               conv(cat(up-sampled features, skip features[, way-point map])) (models/ynet.py:421-445) -- with 32 output channels and at
 *                               most 56 input channels after every source is padded to a multiple of 4 (ynet_conv2d_winograd_cat_supported);
 *                               ynet_winograd_filter_cat transforms the packed filter for that channel layout
 *                               (ynet_winograd_filter_cat_floats(src_c, nsrc, cout) floats). */
int ynet_conv2d_winograd_cat_supported(int B, int H, int W, const int* src_c, int nsrc, int cout, int K);
long long ynet_winograd_filter_cat_floats(const int* src_c, int nsrc, int cout);
int ynet_winograd_filter_cat(const float* wp, float* u, const int* src_c, int nsrc, int cout, int col0, int cols_total, void* stream);
int ynet_conv2d_winograd_cat(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                             long long dst_bs, int cout, int B, int H, int W, int relu, void* stream);
/*   ynet_conv2d_winograd_cat_add   ... + a precomputed term in front of the ReLU, as ynet_conv2d_add: y = [relu](conv(cat(src)) + bias + addend[b % addend_bmod])
 *                               (addend_bmod 0: image b) -- the K goal samples of utils/evaluate.py:248-283 share the skip-feature part of each decoder level's
 *                               first convolution; addend 8-byte aligned, image stride addend_bs floats. */
int ynet_conv2d_winograd_cat_add(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                 long long dst_bs, int cout, int B, int H, int W, int relu, const float* addend, long long addend_bs, int addend_bmod,
                                 void* stream);
/*   ynet_conv2d_winograd_cat_pool  ... + the 2 x 2 max-pooled copy of the output [B][cout][H/2][W/2] from the same launch, as ynet_conv2d_pool (the encoder's
 *                               conv + ReLU in front of MaxPool2d(2, 2), models/ynet.py:196-213): a lane of the Winograd tiling holds exactly the block it pools. */
int ynet_conv2d_winograd_cat_pool(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                  long long dst_bs, float* pooled, long long pooled_bs, int cout, int B, int H, int W, int relu, void* stream);
/*   ynet_conv2d_winograd_cat_pool_code  (round 5) the same launch for a 32-channel ReLU output, which also leaves what the pool's BACKWARD needs of every 2 x 2 block in one
 *                               byte, code [B][32][H/2][W/2]: bits 0..1 the arg-max (first maximum in window scan order, a NaN wins: ynet_maxpool2_bwd's rule), bits 2..5
 *                               "element is positive" in scan order (the ReLU backward ynet_maxpool2_bwd_add applies with relu_mask) -- ynet_maxpool2_bwd_add_code then
 *                               routes the pooled gradient without reading the full-resolution activation again. */
int ynet_conv2d_winograd_cat_pool_code(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                       long long dst_bs, float* pooled, long long pooled_bs, unsigned char* code, int B, int H, int W, void* stream);
/*   The Winograd-native 1-bit ReLU mask (round 5), the counterpart of ynet_conv2d_relu_bits / ynet_conv2d_dgrad_relu_bits for conv -> ReLU -> conv chains whose
 *   launches are Winograd ones with 32 channels in between: the FORWARD launch of the first convolution also writes one bit per output element (y > 0) -- one 32-bit
 *   word per lane and unit of the tiling both launches share, ynet_winograd_relu_bits_words(B, H, W) words --, and the data gradient of the second convolution
 *   (ynet_conv2d_winograd_dgrad_relu_bits) is gated by it instead of fetching the float activation (ynet_conv2d_winograd_dgrad_relu): bit-identical results,
 *   1 / 32 of the mask traffic.  ynet_conv2d_winograd_relu_bits = ynet_conv2d_winograd with relu 1 and 32 outputs; ynet_conv2d_winograd_cat_relu_bits =
 *   ynet_conv2d_winograd_cat (addend NULL) or ynet_conv2d_winograd_cat_add with relu 1. */
long long ynet_winograd_relu_bits_words(int B, int H, int W);
int ynet_conv2d_winograd_relu_bits(const float* src, long long src_bs, const float* u, const float* bias, float* dst, long long dst_bs, int cin, int B, int H, int W,
                                   unsigned* bits_out, void* stream);
int ynet_conv2d_winograd_cat_relu_bits(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                                       long long dst_bs, int B, int H, int W, const float* addend, long long addend_bs, int addend_bmod, unsigned* bits_out,
                                       void* stream);
int ynet_conv2d_winograd_dgrad_relu_bits(const float* dy, long long dy_bs, const float* u, float* dx, long long dx_bs, const unsigned* bits, int dy_c, int B, int H,
                                         int W, void* stream);
/*   ynet_conv2d_winograd16      the SLICE form of the same convolution (round 5): every workgroup keeps the transformed filter of 16 output channels in LDS
 *                               and a wave computes two row pairs (4 output rows x 32 columns) per unit -- for the layers the two kernels above do not
 *                               serve, 16 / 32 / 64 / 128 output channels from up to three concatenated sources of at most 84 padded input channels
 *                               (the 64-channel convolutions and data gradients at 64^2 of models/ynet.py:200-211, 420-445, and the 32 -> 16 up-convolution,
 *                               ynet.py:464), H % 32 == 0, W % 32 == 0, B * H * W >= 64 * 64 * 10 (ynet_conv2d_winograd16_supported).  At most ONE epilogue
 *                               variant: relu_of (a data gradient written through that activation's ReLU backward, as ynet_conv2d_winograd_dgrad_relu: bias
 *                               NULL, relu 0), addend (+ addend[b % addend_bmod] in front of the ReLU, as ynet_conv2d_winograd_cat_add) or pooled (the
 *                               2 x 2 max-pooled copy, as ynet_conv2d_winograd_cat_pool); the others NULL.  ynet_winograd16_filter writes the slice-major
 *                               filter (ynet_winograd16_filter_floats floats) from a packed filter of ynet_pack_weight, output channels [col0, col0 + cout)
 *                               of its cols_total.  Every tensor is addressed one image per buffer descriptor (any batch size; an image below 2 GB). */
int ynet_conv2d_winograd16_supported(int B, int H, int W, const int* src_c, int nsrc, int cout, int K);
long long ynet_winograd16_filter_floats(const int* src_c, int nsrc, int cout);
int ynet_winograd16_filter(const float* wp, float* u, const int* src_c, int nsrc, int cout, int col0, int cols_total, void* stream);
int ynet_conv2d_winograd16(const float* const* src, const int* src_c, const long long* src_bs, int nsrc, const float* u, const float* bias, float* dst,
                           long long dst_bs, int cout, int B, int H, int W, int relu, const float* relu_of, long long relu_of_bs, const float* addend,
                           long long addend_bs, int addend_bmod, float* pooled, long long pooled_bs, void* stream);
/*   ynet_upsample2x_conv2d_winograd   dst = [relu](conv3x3(upsample_bilinear2d(src, scale 2, align_corners = false)) + bias): the decoders'
 *                               `F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)` followed by `upsample_conv[i]` (models/ynet.py:463-464) as
 *                               ONE launch -- the up-sampled tensor (4x the input) is never written: every lane builds its 4 x 4 patch of it from the 3 x 3
 *                               low-resolution patch underneath (the bilinear phase is the same for every 2 x 2 output block), clamping at the image border as
 *                               ATen does and zero-padding the UP-SAMPLED image for the convolution.  src: cin planes of (H / 2) x (W / 2) per image, dst: cout
 *                               planes of H x W; u = ynet_winograd_filter(wp, u, cin, cout, 0, cout); serves cin 32 -> cout 16, H % 16 == 0, W % 32 == 0,
 *                               B * H * W >= 128 * 128 * 8 -- ynet_upsample2x_conv2d_winograd_supported returns 1 --, and in the SLICE form (it returns 2: u = ynet_winograd16_filter of the one source) cin a multiple of 4 up to 84,
 *                               cout in {16, 32, 64}, H % 32 == 0, B * H * W >= 64 * 64 * 10: the 64 -> 32 up-convolutions at 128^2 and 64^2.  H, W are the up-sampled size. */
int ynet_upsample2x_conv2d_winograd_supported(int B, int H, int W, int cin, int cout, int K);
int ynet_upsample2x_conv2d_winograd(const float* src, long long src_bs, const float* u, const float* bias, float* dst, long long dst_bs, int cin, int cout, int B,
                                    int H, int W, int relu, void* stream);

/* ---- ONE dispatching convolution entry (round 6) ----------------------------------------------------------------------------------
 * ynet_conv2d_auto = nn.Conv2d(K x K, padding K / 2) [+ nn.ReLU] (models/ynet.py:150,192-211,420-451,464,467) or its data gradient
 * (convolution_backward -> grad_input), with -- all optional -- the concatenation of its inputs (ynet.py:387,466,574), the bilinear x2 in
 * FRONT of it (`upsample2x`: F.interpolate(scale_factor=2) + upsample_conv, ynet.py:463-464; src is the low-resolution map, H x W the
 * up-sampled size), the MaxPool2d(2, 2) BEHIND it (`pooled`, ynet.py:202,215), the ReLU backward of the layer below a data gradient
 * (`relu_of`, or one of the 1-bit forms), and a precomputed additive term (`addend`, utils/evaluate.py:248-283).  This is SURVEY 8(b)'s
 * `ynet_conv2d_fwd(x, w, b, y, ..., x2, Cin2, pre_op)` / `ynet_conv2d_dgrad(dy, w, y_for_relu_mask, dx, ...)`: the caller describes the
 * operation, the library picks the kernel family (implicit GEMM / Winograd F(2x2,3x3) / its concatenated-source, slice and up-convolution
 * forms: every ynet_conv2d_* entry point above), splits wide layers into the launches those kernels serve, and keeps the TRANSFORMED filters
 * in a cache the caller owns.  It is the path bench.py measures: motion-style-transfer_amd/ops.py::conv2d_raw is a thin call of it.
 *
 *   operands     src / dst / mask / wp / bias / workspace: as ynet_conv2d (wp = ynet_pack_weight mode 0, or mode 1 for a data gradient);
 *                dst[i] may be NULL (those output channels are not wanted: a way-point map's gradient).
 *   filter cache `cache`: 16-byte aligned device memory of ynet_conv2d_auto_cache_floats(desc) floats (0: this call needs none);
 *                `cache_tag`: two 64-bit words in HOST memory, zero before the first call; `wp_version`: any number the caller changes
 *                whenever the contents of wp change.  The transforms run on `stream` when the tag does not match (first call, new
 *                version, or a plan that differs: another batch / raster size can choose another family); a cache belongs to ONE
 *                (layer, direction) and one stream order -- a caller that shares it between streams orders them itself (taken->transformed
 *                tells when a transform was enqueued).
 *   flags        YNET_AUTO_* below: development switches that restrict the choice (results then differ by fp32 rounding only).
 *   taken        (optional) what ran: the family, the launches with the template arguments their rocprofv3 kernel names carry, whether
 *                the optional outputs wbits_out / pool_code were WRITTEN (they are only when the launch taken is the Winograd one they
 *                belong to; a caller keeps them only then).
 *   errors       non-zero + ynet_last_error(): operand combinations no kernel serves (upsample2x outside
 *                ynet_upsample2x_conv2d_winograd_supported, addend outside ynet_conv2d_add_supported and the Winograd forms, ...). */
#define YNET_AUTO_NO_WINOGRAD 1u        /* implicit GEMM only */
#define YNET_AUTO_NO_WINOGRAD16 2u      /* no slice-form launches */
#define YNET_AUTO_WINOGRAD16_FOR_16 4u  /* 16-output launches on the slice form too (measured slower) */
#define YNET_AUTO_NO_POOL_CODE 8u       /* pool_code is never written */
#define YNET_AUTO_NO_RELU_WBITS 16u     /* wbits_out is never written, relu_wbits never read (relu_of's float activation instead) */
#define YNET_AUTO_NO_SPLIT48 32u        /* a [16, 32]-channel data gradient as two launches, not ynet_conv2d_winograd_split */
typedef struct YnetConvAuto {
    const float* src[4];
    int src_c[4];
    long long src_bs[4];
    int src_bmod[4];
    int nsrc;
    const float* mask;
    long long mask_bs;
    const float* wp;
    const float* bias;
    float* dst[4];
    int dst_c[4];
    long long dst_bs[4];
    int ndst;
    int B, H, W, K, relu;
    int upsample2x;
    const float* relu_of;
    long long relu_of_bs;
    float* pooled;
    long long pooled_bs;
    unsigned char* pool_code;
    const float* addend;
    long long addend_bs;
    int addend_bmod;
    unsigned* bits_out;
    const unsigned* relu_bits;
    unsigned* wbits_out;
    const unsigned* relu_wbits;
    float* cache;
    long long cache_floats;
    unsigned long long* cache_tag;
    unsigned long long wp_version;
    float* workspace;
    long long workspace_floats;
    unsigned flags;
    int dst_s2d[4];        /* dst_s2d[i] != 0: destination i (16 or 32 channels, a plain data gradient) may be written SPACE-TO-DEPTH -- element (2 i + r, 2 j + c) of
                              channel ch at plane (2 r + c) * dst_c + ch, position (i, j) of [4 dst_c][H / 2][W / 2]: the layout ynet_upconv_dgrad_ring and the
                              low-resolution data gradient of an up-convolution consume; honoured only where one Winograd launch writes the whole destination
                              (taken->wrote_s2d tells) */
} YnetConvAuto;
typedef struct YnetConvTaken {
    int family;            /* 0 implicit GEMM (conv_mfma_kernel / conv_dma_*), 1 conv_wino_kernel, 2 conv_wino_cat_kernel, 3 conv_wino16_kernel,
                              4 conv_wino_up_kernel, 5 conv_wino16_up_kernel */
    int variant;           /* the dispatcher's branch (csrc/conv_auto.cpp) */
    int nlaunch;
    int tmpl[4][3];        /* per launch: family 1 <NCB, NCH, EM>; family 2 <2, EPI>; family 3 <EPI> */
    int wrote_wbits, wrote_pool_code, transformed;
    int wrote_s2d;         /* bit i: destination i was written space-to-depth */
} YnetConvTaken;
long long ynet_conv2d_auto_cache_floats(const YnetConvAuto* desc);
long long ynet_conv2d_auto_workspace_floats(const YnetConvAuto* desc);
int ynet_conv2d_auto(const YnetConvAuto* desc, YnetConvTaken* taken, void* stream);
/* The choice ynet_conv2d_auto WOULD make for this descriptor (family, variant, number of launches), without launching anything: pointers are
 * looked at for their alignment and for NULL only.  A caller that must decide earlier what a later call will do -- the forward pass of a
 * conv -> ReLU -> conv chain leaves the mask in the layout of the kernel family that will write the data gradient -- asks the dispatcher itself. */
int ynet_conv2d_auto_plan(const YnetConvAuto* desc, YnetConvTaken* taken);

/* The backward of `upsample_conv[i](F.interpolate(x, scale_factor=2, mode='bilinear'))` (models/ynet.py:463-464) WITHOUT the up-sampled gradient (round 6;
 * VERDICT r5 item 3).  Up^T . conv^T is itself a 3 x 3 convolution at the LOW resolution over the space-to-depth output gradient D [4 cout][h][w]: output pixel
 * (2 i + py, 2 j + px) of the forward pass sees the 3 x 3 low-resolution patch around (i, j) through the effective filter
 *     Keff[(py, px, co)][ci][a][b] = sum_{ty, tx} M_py[a][ty] K[co][ci][ty][tx] M_px[b][tx],   M_0 = [[.75 .25 0] [.25 .75 .75] [0 0 .25]],  M_1 = [[.25 0 0] [.75 .75 .25] [0 .25 .75]]
 * so dx = the ordinary data gradient of a convolution with weight Keff [4 cout][cin][3][3] applied to D -- one ynet_conv2d_auto call (mode-1 packed Keff, relu_of = x) --
 * EXCEPT on the outermost ring of dx: there the bilinear clamp (L[-1] = L[0]) and the zero padding of the UP-SAMPLED image differ from that formula.  The difference is again
 * linear and thin: ynet_upconv_dgrad_ring adds it to the ring pixels of dx (masked by relu_of > 0 when given):
 *     rows 0 / h-1:  sum_{c', b} tables[s * 3 + b][c'][ci] D[c'][row][j - b + 1],  tables[s*3+b] = sum dM_s[py][ty] K M_px[b][tx]
 *     columns 0 / w-1:  tables[6 + s * 3 + a] alike;  the four corners additionally tables[12 + 2 sv + sh][c'][ci] D[c'][corner]
 *     dM_0 = .25 [[-1 1 0] [1 0 0]] (couples to L[0]),  dM_1 = .25 [[0 0 1] [0 1 -1]] (couples to L[h-1])
 * (motion-style-transfer_amd/ops.py::upconv_s2d_tables builds Keff and the 16 tables from the layer's filter; tests compare with autograd in fp64.)
 * D: [B][C4 = 4 cout][h][w] (batch stride d_bs); tables: [16][C4][cin] floats; dx [B][cin][h][w] is updated in place. */
int ynet_upconv_dgrad_ring(const float* D, long long d_bs, const float* tables, const float* relu_of, long long relu_of_bs, float* dx, long long dx_bs,
                           int B, int C4, int cin, int h, int w, void* stream);

/* Introspection for profiling: the instantiation the dispatcher uses for this problem, encoded as
 * rows | tiles << 8 | m16 << 16 | dma << 17 | x4 << 18 | log2(fold) << 19 | CC << 21  ->  conv_mfma_kernel<K, tiles,
 * rows, CC, mask, m16>, or with dma the LDS-DMA generation conv_dma_kernel<tiles, rows, CC, mask, x4, fold>, in a
 * rocprof trace (CC = input channels per staged chunk). */
int ynet_conv2d_plan(int B, int H, int W, int cout, int K);

/* convolution_backward -> grad_weight [Cout][Cin][K][K] and grad_bias [Cout] (db may be NULL).
 * x = concat(src...), dy = incoming gradient, mask = post-ReLU activation of this layer or NULL.
 * workspace: ynet_conv2d_wgrad_workspace_floats(...) floats.
 */
long long ynet_conv2d_wgrad_workspace_floats(int B, int H, int W, int cout, int cin, int K);
int ynet_conv2d_wgrad(const float* const* src, const int* src_c, const long long* src_bs, int nsrc,
                      const float* dy, long long dy_bs, const float* mask, long long mask_bs,
                      float* dw, float* db, float* workspace, int B, int H, int W, int cout, int K,
                      void* stream);

/* ---- MoSA / LoRA (loralib==0.1.1 Conv2d, call site models/ynet.py:141-144) -------------------
 * lora_a [r*K][cin*K], lora_b [cout*K][r*K], scale = lora_alpha / r (lora_alpha = 1).
 *   compose: w_eff = w + (lora_b @ lora_a).view(w.shape) * scale
 *   grad:    d_a = scale * lora_b^T @ dWm,  d_b = scale * dWm @ lora_a^T,  dWm = dw.view(cout*K, cin*K)
 */
int ynet_lora_compose(const float* w, const float* lora_a, const float* lora_b, float scale, float* w_eff,
                      int cout, int cin, int K, int r, void* stream);
int ynet_lora_grad(const float* dw, const float* lora_a, const float* lora_b, float scale, float* d_a, float* d_b,
                   int cout, int cin, int K, int r, void* stream);

/* lora_compose followed by ynet_pack_weight in mode 0 AND mode 1, in one launch: W_eff is never materialised, its
 * elements go straight into the two packed layouts.  wp_fwd / wp_dgrad: ynet_packed_weight_floats(cout, cin, K, 0 / 1)
 * floats each, ZERO-FILLED by the caller before the first use (the padding is not written; the buffers can be
 * reused for the same layer every step). */
int ynet_lora_compose_pack(const float* w, const float* lora_a, const float* lora_b, float scale, float* wp_fwd,
                           float* wp_dgrad, int cout, int cin, int K, int r, void* stream);
/* The same for n <= 48 layers in ONE launch (every argument a host array of n entries): all adapted convs of an encoder
 * are composed at the start of a step instead of one by one between its convolutions.  r[i] == 0: layer i has no adapter
 * (lora_a[i] / lora_b[i] may be NULL) and its weight is packed as it is -- ynet_pack_weight in both modes -- which is how
 * the 46 trainable convs of train_net = train / all (models/trainer.py:116-195) are re-packed after an optimizer step. */
int ynet_lora_compose_pack_multi(int n, const float* const* w, const float* const* lora_a, const float* const* lora_b,
                                 const float* scale, float* const* wp_fwd, float* const* wp_dgrad, const int* cout,
                                 const int* cin, const int* K, const int* r, void* stream);

/* Adapter gradients of a 3x3 loralib Conv2d WITHOUT the full filter gradient (models/ynet.py:141-144; replaces the chain
 * ynet_conv2d_wgrad -> ynet_lora_grad for the adapted convs of train_net = mosa_1):
 *   d_a [3 r][3 cin] = s * B^T * dWm,  d_b [3 cout][3 r] = s * dWm * A^T,  dWm = dW.view(3 cout, 3 cin)
 * computed from 9 r planes projected out of x (through lora_A) and out of dy (through lora_B): (54 cin + 18 cout) r MACs
 * per pixel instead of 9 cin cout, no dW round trip through HBM.  `mask`: the conv's post-ReLU output (dy is zeroed where
 * it is <= 0) or NULL.  Sources: the virtual concatenation of the conv's inputs, as for ynet_conv2d_wgrad; 16-byte aligned
 * planes, W % 4 == 0.  workspace: ynet_lora_conv2d_wgrad_workspace_floats(cin, cout) floats.
 * ynet_lora_conv2d_wgrad_supported: K == 3, r == 1, cin, cout <= 64, W % 4 == 0 (anything else: the two-call chain). */
int ynet_lora_conv2d_wgrad_supported(int cin, int cout, int K, int r, int W);
/* ... and where it is also the faster of the two paths (the layers with more than 32 output channels; measured) */
int ynet_lora_conv2d_wgrad_preferred(int cin, int cout, int K, int r, int W);
long long ynet_lora_conv2d_wgrad_workspace_floats(int cin, int cout);
int ynet_lora_conv2d_wgrad(const float* const* src, const int* src_c, const long long* src_bs, int nsrc,
                           const float* dy, long long dy_batch_stride, const float* mask, long long mask_batch_stride,
                           const float* lora_a, const float* lora_b, float scale, float* d_a, float* d_b,
                           float* workspace, int B, int H, int W, int cout, int K, int r, void* stream);

/* ---- pooling / resampling ------------------------------------------------------------------- */
/* nn.MaxPool2d(2,2) (models/ynet.py:202,215,326,340,354,367); N = B*C planes of H x W. */
int ynet_maxpool2_fwd(const float* x, float* y, long long N, int H, int W, void* stream);
int ynet_maxpool2_bwd(const float* x, const float* dy, float* dx, long long N, int H, int W, void* stream);
/* dx = maxpool2_bwd(x, dy) + add0 + add1 (addends may be NULL; even H, W): folds the skip-connection gradients of the
 * two decoders into the pool's backward instead of two autograd adds (utils/train_epoch.py:110 loss.backward()).
 * relu_mask != 0: dx is also zeroed where x <= 0 -- x is the post-ReLU output of the conv that receives dx as its output
 * gradient (models/ynet.py:196-211: nn.ReLU after every encoder conv), so the ReLU backward that conv's dgrad / wgrad
 * would apply by reading x again is applied here, where x is in registers anyway. */
int ynet_maxpool2_bwd_add(const float* x, const float* dy, const float* add0, const float* add1, float* dx, long long N,
                          int H, int W, int relu_mask, void* stream);
/* The same with the activation replaced by the code plane of ynet_conv2d_winograd_cat_pool_code ([N][H/2][W/2] bytes): bit-identical dx, x is not read. */
int ynet_maxpool2_bwd_add_code(const unsigned char* code, const float* dy, const float* add0, const float* add1, float* dx, long long N, int H, int W,
                               int relu_mask, void* stream);
/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) (models/ynet.py:463);
 * H, W are the LOW-resolution sizes in both directions. */
int ynet_upsample2x_fwd(const float* x, float* y, long long N, int H, int W, void* stream);
int ynet_upsample2x_bwd(const float* dy, float* dx, long long N, int H, int W, void* stream);
/* ... with the ReLU backward of the up-sampled activation (relu_of [N][H][W], contiguous: the post-ReLU output of the layer in
 * front of the interpolation, models/ynet.py:463) applied to dx: dx = relu_of > 0 ? dx : 0 */
int ynet_upsample2x_bwd_relu(const float* dy, float* dx, const float* relu_of, long long N, int H, int W, void* stream);
/* [x] + [AvgPool2d(2^i)(x) for i = 1..nlev] (utils/train_epoch.py:97-100, utils/evaluate.py:255-257);
 * outs[i-1] receives level i; H, W multiples of 32. */
int ynet_avgpool_pyramid(const float* x, float* const* outs, int nlev, long long N, int H, int W, void* stream);

/* ---- the serial adapters' element-wise tail (round 6; SURVEY 8(f)-2, outside every BASELINE configuration) ------------------------------ */
/* nn.BatchNorm2d of AdapterBlock / AdapterLayer.serial_layer[0] (models/ynet.py:24-26,64-66): F.batch_norm's rule on x [B][C][HW] contiguous.
 * train != 0: batch statistics (biased variance for the normalisation; running_mean / running_var, if given, move by `momentum` towards the batch
 * mean / the UNBIASED batch variance); train == 0: the running statistics.  save_mean / save_invstd [C] are written either way (what the backward
 * needs; in evaluation mode save_mean is not written: pass running_mean to the backward).  gamma / beta may be NULL (affine = False).  Sums are
 * fp64 over 64 fixed slices per channel: bitwise reproducible.  workspace: ynet_batchnorm_workspace_doubles(C) doubles. */
long long ynet_batchnorm_workspace_doubles(int C);
int ynet_batchnorm2d_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                         double* workspace, int B, int C, long long HW, int train, double momentum, double eps, void* stream);
/* native_batch_norm_backward: dgamma = sum dy * xhat, dbeta = sum dy (either may be NULL), dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)) in
 * training mode, dy * gamma * invstd in evaluation mode (mean = the running mean then). */
int ynet_batchnorm2d_bwd(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, float* dgamma, float* dbeta, double* workspace,
                         int B, int C, long long HW, int train, void* stream);
/* y = [relu](a + b): the adapters' residual add and the ReLU behind it (models/ynet.py:66,117-131); dx = y > 0 ? dy : 0 for both addends. */
int ynet_add_relu(const float* a, const float* b, float* y, long long n, int relu, void* stream);
int ynet_relu_bwd(const float* dy, const float* y, float* dx, long long n, void* stream);

/* ---- loss ----------------------------------------------------------------------------------- */
/* nn.BCEWithLogitsLoss() (models/trainer.py:206; utils/train_epoch.py:94,106): loss[0] = mean.
 * workspace: ynet_bce_workspace_bytes() bytes.  bwd: dx = (sigmoid(x) - t) * grad_out[0] / n. */
long long ynet_bce_workspace_bytes(void);
int ynet_bce_logits_fwd(const float* x, const float* t, long long n, float* loss, void* workspace, void* stream);
int ynet_bce_logits_bwd(const float* x, const float* t, const float* grad_out, float* dx, long long n, void* stream);
/* The training form: the same loss, and in the same pass dx = (sigmoid(x) - t) * expected_grad / n, where
 * expected_grad is the upstream gradient the caller expects (loss_scale in utils/train_epoch.py:94,106).
 * ynet_bce_grad_rescale then multiplies dx by grad_out[0] / expected_grad on the device -- a no-op launch
 * when the expectation was right, so the logits are read once per step instead of twice. */
int ynet_bce_logits_fwd_grad(const float* x, const float* t, long long n, float expected_grad, float* loss, float* dx,
                             void* workspace, void* stream);
int ynet_bce_grad_rescale(float* dx, const float* grad_out, float expected_grad, long long n, void* stream);

/* Predictor + criterion in ONE pass over the last decoder activation (models/ynet.py:450-451,469 `self.predictor(x)`
 * followed by utils/train_epoch.py:93-94,105-106 `criterion(pred_map, gt_map)`, and the predictor's dgrad of the
 * backward pass): y = conv1x1(x, w) + bias [B][cout][HW]; loss[0] = mean BCE-with-logits(y, target);
 * dy = (sigmoid(y) - target) * expected_grad / n (optional output, wanted when the predictor trains);
 * dx = conv1x1^T(dy) [B][cin][HW] (optional output: the gradient handed to decoder.4.2).  wp = ynet_pack_weight(w, mode 0)
 * of the 1x1 filter; cout <= 32, HW % 4 == 0, 16-byte aligned tensors.  workspace: ynet_pred_bce_workspace_bytes()
 * bytes, ZEROED before the first use (the kernel leaves its ticket counter at zero).  ynet_bce_grad_rescale corrects
 * dx / dy when the upstream gradient turns out not to be expected_grad.
 * dx_relu_mask != 0 (cin <= 32): dx is zeroed where x <= 0 -- the ReLU backward of decoder.4.2 (models/ynet.py:447-449),
 * whose output x is, applied while x streams through this kernel instead of by that conv's dgrad reading x again. */
long long ynet_pred_bce_workspace_bytes(void);
int ynet_pred_bce(const float* x, long long x_batch_stride, const float* wp, const float* bias, const float* target,
                  float* y, float* loss, float* dx, float* dy, void* workspace, int B, int cin, int cout, long long HW,
                  float expected_grad, int dx_relu_mask, void* stream);
/* The same pass with the target given by POSITION (round 5): plane (b, co) of the target is the H x W window of the S x S Gaussian template
 * around (x, y) = target_xy[2 * (b * cout + co) ..], i.e. what get_patch(gt_template, gt_future, H, W) holds (utils/train_epoch.py:68-72,
 * utils/image_utils.py:15-27,40-63) and ynet_heatmap_analytic(kind 1) writes: the kernlen x kernlen `blob` placed at the rounded position, zero
 * elsewhere, all zero when the window would leave the template.  The 12 target planes are computed, not read (88 -> 76 planes of traffic per pixel);
 * results bit-identical to ynet_pred_bce on the materialised target.  W % 4 == 0, kernlen <= S, S >= H, W. */
int ynet_pred_bce_blob(const float* x, long long x_batch_stride, const float* wp, const float* bias, const float* target_xy, const float* blob, int kernlen,
                       int S, int H, int W, float* y, float* loss, float* dx, float* dy, void* workspace, int B, int cin, int cout, float expected_grad,
                       int dx_relu_mask, void* stream);

/* ---- goal / trajectory read-out -------------------------------------------------------------- */
/* SoftArgmax2D.forward (utils/softargmax.py:55-81; models/ynet.py:582-583): x [B][C][H][W] with
 * batch stride `batch_stride` elements (so a channel slice such as pred_goal_map[:, -1:] needs no
 * copy) -> out [B][C][2] = (E[x], E[y]) in pixels. */
int ynet_softargmax2d(const float* x, float* out, long long B, int C, long long batch_stride, int H, int W,
                      void* stream);

/* The read-out of a training step (utils/train_epoch.py:118-126) in two launches:
 *   pred_traj [B][P][2] = SoftArgmax2D(traj_map [B][P][H][W]);  pred_goal [B][1][2] = SoftArgmax2D(goal_map[:, goal_channel])
 *   ade[b] = mean_p |gt_future[b][p] - pred_traj[b][p]| / resize_factor;  fde[b] = |gt_future[b][P-1] - pred_goal[b][0]| / resize_factor
 * (the reference: two soft-argmax calls and a dozen elementwise ATen launches). */
int ynet_train_readout(const float* traj_map, long long traj_batch_stride, const float* goal_map, long long goal_batch_stride,
                       int goal_channel, const float* gt_future, float* pred_traj, float* pred_goal, float* ade, float* fde,
                       int B, int P, int H, int W, float resize_factor, void* stream);
/* The 1x1 predictor (models/ynet.py:469: nn.Conv2d(decoder_channels[-1], pred_len, 1)) followed by SoftArgmax2D
 * (utils/softargmax.py:55-81) as evaluate() chains them for every trajectory sample (utils/evaluate.py:259-262:
 * `model.softargmax(model.pred_traj(...))`) in one pass: x [B][cin][H][W] (batch stride x_bs elements), w [cout][cin] the
 * filter as the checkpoint stores it, bias [cout] or NULL -> out [B][cout][2] = (E[x], E[y]) in pixels.  The logits are
 * never written.  workspace: ynet_pred_softargmax_workspace_floats(B, H, W) floats, 16-byte aligned.
 * ynet_pred_softargmax_supported: cin in {8, 16, 32}, cout <= 32, W % 4 == 0, H*W % 128 == 0; other shapes take
 * ynet_conv2d + ynet_softargmax2d. */
int ynet_pred_softargmax_supported(int cin, int cout, int H, int W);
long long ynet_pred_softargmax_workspace_floats(long long B, int H, int W);
int ynet_pred_softargmax(const float* x, long long x_bs, const float* w, const float* bias, float* out, float* workspace,
                         long long B, int cin, int cout, int H, int W, void* stream);
/* sigmoid(pred_goal_map[:, sel] / temperature) (utils/evaluate.py:128-131; models/ynet.py:585-586):
 * x [B][C][HW] -> y [B][nsel][HW]; sel is a HOST array of nsel (<= 8) channel indices. */
int ynet_sigmoid_temp(const float* x, float* y, long long B, int C, long long HW, const int* sel, int nsel,
                      float temperature, void* stream);

/* ---- heat-map construction -------------------------------------------------------------------- */
/* get_patch + torch.stack (utils/image_utils.py:40-63; utils/train_epoch.py:63-78;
 * utils/evaluate.py:112-114,250-253): out[n] = tmpl[SH/2 - ry : +H, SW/2 - rx : +W] with
 * (rx, ry) = rint(xy[n]) (round-half-even like np.round); xy [N][2] fp32 on the device.
 * *status (device int, zero it first) becomes 1 if any window leaves the template. */
int ynet_gather_patch(const float* tmpl, int SH, int SW, const float* xy, float* out, int N, int H, int W,
                      int* status, void* stream);

/* The same windows WITHOUT the S x S templates (create_dist_mat / create_gaussian_heatmap_template,
 * utils/image_utils.py:15-37, are functions of the distance to the rounded coordinate):
 *   kind 0: out[n,y,x] = (float)(sqrt((double)((y-ry)^2 + (x-rx)^2)) / dmax * 2), dmax = sqrt(2) * (S / 2) in fp64 --
 *           bit-identical to the float64 NumPy template cast to fp32;
 *   kind 1: the kernlen x kernlen Gaussian blob (`blob`, device, fp32 values of the template) placed at (rx, ry), 0 elsewhere.
 * S is the size of the virtual template: a window that would leave it is flagged in *status and zero-filled. */
int ynet_heatmap_analytic(const float* xy, float* out, int N, int H, int W, int S, int kind, double dmax,
                          const float* blob, int kernlen, int* status, void* stream);

/* ---- test-time sampling trick ------------------------------------------------------------------ */
/* kmeans (utils/kmeans.py:22-108) as evaluate() calls it for TTST (utils/evaluate.py:146-152): Lloyd's algorithm on
 * P independent sets of N 2-D points with integer-valued coordinates (sampled pixels), K clusters each.
 * points [P][N][2] fp32, init_idx [P][K] = indices of the initial centres (the host draws them with
 * np.random.choice, keeping the reference's RNG stream), centers [P][K][2] out.  Stops when (sum of centre
 * shifts)^2 < tol or after iter_limit iterations (0 = no limit).  status[p] = (iterations << 8) | empty, where
 * empty = 1 means a cluster lost all its points (the reference then re-seeds it with torch.randint): the centres of
 * that set are undefined and the caller must redo it on the host path.  N <= 18000, K <= 32. */
int ynet_kmeans2d(const float* points, const int* init_idx, float* centers, int* status, int P, int N, int K, float tol,
                  int iter_limit, void* stream);

/* ---- goal / waypoint sampling ------------------------------------------------------------------ */
/* torch.multinomial as utils/image_utils.py:110-135 (`sampling`) calls it, with a DOCUMENTED generator so that the CPU
 * restatement (oracle/ynet_oracle.py: device_multinomial) reproduces every draw from `seed` alone.
 * prob: `rows` rows of n non-negative floats, `row_stride` floats apart; out [rows][K] int64 element indices.
 * Philox4x32-10, key = (seed low word, seed high word), counter = (element or sample index, 0, row, stream);
 * u = ((x0 >> 5) * 2^26 + (x1 >> 6) + 0.5) * 2^-53 from the first two output words.
 *   replacement = 0 (stream 0): exponential race -- key_i = p_i / -log(u_i) in fp64, the K largest keys in descending
 *     order, ties to the smaller index; K <= min(48, n).
 *   replacement = 1 (stream 1): inverse CDF in fp64 -- the row is cut into 256 contiguous segments of ceil(n / 256)
 *     elements, summed sequentially; sample j = first element whose running sum >= u_j * total.
 * rel_threshold > 0 zeroes entries below rel_threshold * max(row) first (image_utils.py:113-118).
 * *status (device int, zero it first) becomes 1 if a row has too few (replacement: no) positive entries. */
int ynet_multinomial(const float* prob, long long rows, long long row_stride, int n, int K, int replacement,
                     float rel_threshold, unsigned long long seed, long long* out, int* status, void* stream);
/* The same with the seed read from device memory by the kernel (seed_dev: 8 bytes, 8-byte aligned): no per-call argument, so the
 * launch can be recorded into a hipGraph -- the captured evaluation sweep (utils/evaluate.py:248-266 as one graph per batch shape)
 * copies the seeds it draws on the host into a static device buffer before every replay. */
int ynet_multinomial_devseed(const float* prob, long long rows, long long row_stride, int n, int K, int replacement,
                             float rel_threshold, const unsigned long long* seed_dev, long long* out, int* status, void* stream);
/* Conditioned waypoint sampling prior (utils/evaluate.py:9-34 torch_multivariate_gaussian_heatmap, 198-211): for row r
 * (person r % n_persons) the anisotropic Gaussian centred at mean_xy[r] with its long axis along dist_xy[r], std
 * (|dist| + 5) / sigma_factor along it and that / ratio across (rot: axes swapped), on linspace(0, H, H) x
 * linspace(0, W, W), multiplied by the sigmoid map sig + (r % n_persons) * sig_batch_stride and normalised to sum 1
 * -> out_map [rows][H][W] (optional) and its expectation (sum x * map, sum y * map) -> out_xy [rows][2] (optional).
 * fp64 per pixel. */
int ynet_cws_prior(const float* sig, long long sig_batch_stride, int n_persons, const float* mean_xy, const float* dist_xy,
                   int rows, int H, int W, float sigma_factor, float ratio, int rot, float* out_map, float* out_xy,
                   void* stream);

/* ---- scene pre-processing without OpenCV / the segmentation backbone (the pinnable part) ------------- */
/* pad (utils/image_utils.py:95-107): N planes H x W -> Hp x Wp, zero border at the bottom / right (cv2.copyMakeBorder,
 * BORDER_CONSTANT); the caller rounds Hp, Wp up to the division factor (32). */
int ynet_pad2d(const float* x, float* y, long long N, int H, int W, int Hp, int Wp, void* stream);
/* pad + preprocess_image_for_segmentation(seg_mask = True) (utils/image_utils.py:74-81, 95-107): an int32 label map
 * [H][W] -> `classes` one-hot fp32 planes [classes][Hp][Wp]; the border is padded BEFORE the encoding, i.e. it is class 0. */
int ynet_seg_onehot_pad(const int* labels, float* y, int H, int W, int Hp, int Wp, int classes, void* stream);
/* resize(images, factor, seg_mask = True) (utils/image_utils.py:83-87: cv2.resize(image, (0, 0), fx = fy = factor, INTER_NEAREST)) of
 * an int32 label map [H][W] -> [Ho][Wo], Ho = cvRound(H * fy), Wo = cvRound(W * fx) computed by the caller (half-to-even);
 * out[y][x] = labels[min(floor(y * (1 / fy)), H - 1)][min(floor(x * (1 / fx)), W - 1)], products in double -- OpenCV's published
 * nearest-neighbour rule.  PARITY UNPINNED: cv2 is not in the image; restated from the published algorithm, known-answer tests only. */
int ynet_resize_nearest(const int* labels, int* out, int H, int W, int Ho, int Wo, double fx, double fy, void* stream);
/* augment_data's image side (utils/data_utils.py:113-170): cv2.rotate(image, ROTATE_90_COUNTERCLOCKWISE) applied k times (rot(),
 * lines 133-134) followed, if `flip`, by cv2.flip(image, 1) (fliplr(), line 162) -- pure index permutations, = np.rot90(image, k) and
 * np.fliplr: N planes [H][W] of 32-bit words (int32 label maps or fp32 planes) -> [Ho][Wo], (Ho, Wo) = (W, H) for odd k.
 * Bit-exact by construction; src and dst must not overlap. */
int ynet_rot90_flip(const void* src, void* dst, long long N, int H, int W, int k, int flip, void* stream);
/* augment_data's coordinate side (utils/data_utils.py:127-131,140-141 and 158-161,169-170), float64 like the DataFrame columns, in place
 * on xy [n][2]: (x, y) <- ((x - cx, y - cy) . [[r00, r01], [r10, r11]]) + (ox, oy).  rot(): c = cos(-k pi / 2), s = sin(-k pi / 2) as NumPy
 * evaluates them (passed in: c is 6.1e-17, not 0, for odd k), R = [[c, s], [-s, c]], (cx, cy) = half the image size before, (ox, oy) half
 * the size after the rotation; fliplr(): R = [[-1, 0], [0, 1]]. */
int ynet_rot_coords(double* xy, long long n, double cx, double cy, double r00, double r01, double r10, double r11, double ox, double oy, void* stream);

/* y[i] = sum over b of x[b * batch_stride + i], i < n, in batch order (bitwise reproducible): the backward of `semantic_img.expand(B, ...)`
 * (utils/train_epoch.py:87) and of the batch-broadcast scene features of Y-Net-Mod -- the gradient of a one-image tensor that every
 * trajectory of the batch read.  n and batch_stride multiples of 4, 16-byte aligned. */
int ynet_batch_sum(const float* x, float* y, int B, long long n, long long batch_stride, void* stream);

/* ---- optimizer step --------------------------------------------------------------------------- */
/* torch.optim.Adam / AdamW (models/trainer.py:182: Adam(lr)) for ALL parameters in two launches -- used inside captured training
 * steps, where torch's fused multi-tensor form costs 6 launches (0.18 ms alone on the GPU for a fully trainable Y-Net).  Same update
 * rule and precision choices as torch's fused kernel (no amsgrad / maximize); state stays optimizer.state's own tensors.
 *   table [6][ntensors] 64-bit: pointers to param, grad, exp_avg, exp_avg_sq (fp32, contiguous), step (ONE fp32, incremented here), numel
 *   chunk c = 1024 consecutive elements of tensor chunk_tensor[c] starting at element chunk_first[c]; all arrays on the device. */
int ynet_adam_step(const long long* table, const int* chunk_tensor, const long long* chunk_first, int ntensors, int nchunks,
                   double lr, double beta1, double beta2, double eps, double weight_decay, int adamw, void* stream);

/* ---- data-parallel exchange (new: the reference is single-process) --------------- */
/* One-shot all-reduce(SUM) of the flat trainable-gradient buffer among the GPUs of ONE node: every rank publishes its
 * buffer in a mailbox the other ranks map through HIP IPC (peer access over xGMI) and reads the N - 1 peers directly --
 * one hop instead of a ring's 2 (N - 1) -- summing in rank order, so all ranks obtain bit-identical results.
 *   ynet_comm_create(rank, world, max_floats, &comm)   allocate this rank's mailbox (world <= 16)
 *   ynet_comm_export(comm, handle)                     its IPC handle, ynet_comm_handle_bytes() bytes; exchange the handles
 *                                                      out of band (motion-style-transfer_amd/dist.py: all_gather_object)
 *   ynet_comm_connect(comm, handles)                   `world` handles in rank order (the own entry is ignored)
 *   ynet_allreduce_sum(comm, buf, n, stream)           in place, n <= max_floats; collective: every rank calls it the same
 *                                                      number of times; the call number lives on the device, so the
 *                                                      launch has no per-call argument and CAN be captured into a
 *                                                      hipGraph (utils/step_graph.py does); world == 1 launches too (sum = input)
 *   ynet_comm_status(comm)                             1 if a wait for a peer ever timed out (~20 s), else 0; synchronises.
 *                                                      A timed-out call does NOT leave buf un-reduced silently: the parts that
 *                                                      could not be reduced and the last element (the loss slot of
 *                                                      dist.DataParallel) are set to NaN
 *   ynet_comm_destroy(comm)
 * torch.distributed (RCCL) remains the default transport of dist.DataParallel; this path is selected with
 * YNET_ALLREDUCE=oneshot. */
long long ynet_comm_handle_bytes(void);
int ynet_comm_create(int rank, int world, long long max_floats, void** comm_out);
int ynet_comm_export(void* comm, void* handle_out);
int ynet_comm_connect(void* comm, const void* handles);
int ynet_allreduce_sum(void* comm, float* buf, long long n, void* stream);
int ynet_comm_status(void* comm);
int ynet_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* YNET_HIP_H */
