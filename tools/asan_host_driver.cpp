// Host-side AddressSanitizer driver (tests/test_host_asan.py; `make -C motion-style-transfer_amd/csrc asan`).
// Runs on the CPU, no GPU needed: calls every pure-host entry point of include/ynet_hip.h over a sweep of shapes and
// every launching entry point with arguments its validation must reject (null pointers, empty / oversized shapes), so
// that the instrumented host code -- argument checks, dispatch and workspace arithmetic, error formatting -- executes
// under ASan.  Exit code 0 = every rejection came back as a status code with a message, and ASan saw nothing.
#include <initializer_list>
#include <stdio.h>
#include <string.h>
#include "../include/ynet_hip.h"

static int failures = 0;
#define EXPECT_REJECT(call)                                                                      \
    do {                                                                                         \
        const int rc_ = (call);                                                                  \
        const char* msg_ = ynet_last_error();                                                    \
        if (rc_ == 0 || msg_ == nullptr || strlen(msg_) == 0) {                                  \
            fprintf(stderr, "NOT REJECTED: %s (rc %d)\n", #call, rc_);                           \
            ++failures;                                                                          \
        }                                                                                        \
    } while (0)

int main() {
    if (ynet_abi_version() != 1) ++failures;
    long long acc = 0;
    for (int cout : {1, 12, 16, 17, 32, 48, 64, 65, 130})
        for (int cin : {1, 6, 14, 32, 65, 130})
            for (int k : {1, 3, 5})
                for (int mode : {0, 1}) acc += ynet_packed_weight_floats(cout, cin, k, mode);
    for (int b : {1, 2, 10, 16, 32, 128, 256})
        for (int h : {1, 3, 8, 16, 32, 64, 96, 256, 512})
            for (int w : {1, 5, 8, 16, 32, 50, 64, 160, 256, 512})
                for (int cout : {1, 12, 16, 32, 48, 64, 130}) {
                    acc += ynet_conv2d_workspace_floats(b, h, w, cout);
                    for (int k : {1, 3, 5}) {
                        acc += ynet_conv2d_plan(b, h, w, cout, k);
                        acc += ynet_conv2d_wgrad_workspace_floats(b, h, w, cout, 32, k);
                    }
                }
    acc += ynet_bce_workspace_bytes() + ynet_pred_bce_workspace_bytes() + ynet_comm_handle_bytes();
    if (acc <= 0) ++failures;

    float dummy[4] = {0, 0, 0, 0};
    float* fp = dummy;
    const float* cfp = dummy;
    int one = 1, status = 0;
    long long bs = 4;
    const float* srcs[1] = {cfp};
    float* dsts[1] = {fp};
    const float* null_srcs[1] = {nullptr};
    EXPECT_REJECT(ynet_pack_weight(nullptr, fp, 4, 4, 3, 0, nullptr));
    EXPECT_REJECT(ynet_pack_weight(cfp, fp, 4, 4, 4, 0, nullptr));              // kernel size 4
    EXPECT_REJECT(ynet_pack_weight(cfp, fp, 4, 4, 3, 7, nullptr));              // mode 7
    EXPECT_REJECT(ynet_conv2d(srcs, &one, &bs, nullptr, 0, nullptr, 0, cfp, nullptr, dsts, &one, &bs, 1, 1, 4, 4, 3, 0, nullptr, 0, nullptr));   // no source
    EXPECT_REJECT(ynet_conv2d(srcs, &one, &bs, nullptr, 1, nullptr, 0, nullptr, nullptr, dsts, &one, &bs, 1, 1, 4, 4, 3, 0, nullptr, 0, nullptr)); // no filter
    EXPECT_REJECT(ynet_conv2d(null_srcs, &one, &bs, nullptr, 1, nullptr, 0, cfp, nullptr, dsts, &one, &bs, 1, 1, 4, 4, 3, 0, nullptr, 0, nullptr));
    EXPECT_REJECT(ynet_conv2d(srcs, &one, &bs, nullptr, 1, nullptr, 0, cfp, nullptr, dsts, &one, &bs, 1, 0, 4, 4, 3, 0, nullptr, 0, nullptr));    // B = 0
    EXPECT_REJECT(ynet_conv2d(srcs, &one, &bs, nullptr, 1, nullptr, 0, cfp, nullptr, dsts, &one, &bs, 9, 1, 4, 4, 3, 0, nullptr, 0, nullptr));    // 9 destinations
    EXPECT_REJECT(ynet_conv2d_add(srcs, &one, &bs, nullptr, 1, cfp, nullptr, fp, 32, 16, 1, 4, 4, 3, 1, nullptr, 16, 0, nullptr));      // no addend
    EXPECT_REJECT(ynet_conv2d_add(srcs, &one, &bs, nullptr, 1, cfp, nullptr, fp, 32, 16, 1, 8, 8, 3, 1, cfp, 16, 1, nullptr));          // a map the additive kernels do not serve
    for (int b : {1, 32, 256})
        for (int hw : {8, 32, 64, 256})
            for (int cout : {12, 32, 48, 64}) acc += ynet_conv2d_add_supported(b, hw, hw, cout, 3);
    EXPECT_REJECT(ynet_conv2d_pool(srcs, &one, &bs, 1, cfp, nullptr, fp, 4, 64, nullptr, 16, 1, 4, 4, 3, 1, nullptr));               // no pooled output
    EXPECT_REJECT(ynet_conv2d_pool(srcs, &one, &bs, 1, cfp, nullptr, fp, 4, 64, fp, 16, 1, 5, 4, 3, 1, nullptr));                    // odd H
    for (int b : {1, 32, 256})
        for (int hw : {8, 32, 64, 256})
            for (int c : {12, 32, 48, 64}) acc += ynet_conv2d_pool_supported(b, hw, hw, c, 3) + ynet_conv2d_pool_supported(b, hw + 1, hw, c, 3);
    EXPECT_REJECT(ynet_conv2d_dgrad_relu(cfp, 4, 64, nullptr, 0, cfp, fp, 4, 64, nullptr, 64, 1, 4, 4, 3, nullptr, 0, nullptr));     // no activation
    EXPECT_REJECT(ynet_conv2d_dgrad_relu(cfp, 4, 64, nullptr, 0, cfp, fp, 4, 64, cfp + 1, 64, 1, 4, 4, 3, nullptr, 0, nullptr));     // unaligned activation
    for (int b : {1, 32, 256})
        for (int hw : {8, 32, 64, 256})
            for (int c : {12, 32, 48, 64}) acc += ynet_conv2d_dgrad_relu_supported(b, hw, hw, c, 3) + ynet_conv2d_dgrad_relu_supported(b, hw, hw + 2, c, 3);
    for (int b : {1, 10, 32, 256})
        for (int hw : {8, 32, 64, 128, 256})
            for (int c : {12, 16, 32, 48, 64, 130}) acc += ynet_conv2d_relu_bits_words(b, hw, hw, c, 3) + ynet_conv2d_relu_bits_words(b, hw, hw + 2, c, 3) + ynet_conv2d_relu_bits_words(b, hw, hw, c, 5);
    EXPECT_REJECT(ynet_conv2d_relu_bits(srcs, &one, &bs, 1, cfp, nullptr, fp, 32, 64, nullptr, 8, 128, 128, 3, nullptr));                 // no mask words
    EXPECT_REJECT(ynet_conv2d_relu_bits(srcs, &one, &bs, 1, cfp, nullptr, fp, 64, 64, (unsigned*)dummy, 2, 16, 16, 3, nullptr));            // shape not served
    EXPECT_REJECT(ynet_conv2d_dgrad_relu_bits(cfp, 4, 64, nullptr, 0, cfp, fp, 32, 64, nullptr, 8, 128, 128, 3, nullptr));                 // no mask words
    EXPECT_REJECT(ynet_conv2d_dgrad_relu_bits(cfp, 4, 64, nullptr, 0, cfp, fp, 64, 64, (const unsigned*)dummy, 2, 16, 16, 3, nullptr));     // shape not served
    for (int b : {1, 10, 32, 256})
        for (int hw : {8, 32, 64, 128, 256, 512})
            for (int c : {8, 16, 24, 32, 48, 64}) acc += ynet_conv2d_winograd_supported(b, hw, hw, c, 32, 3) + ynet_conv2d_winograd_supported(b, hw, hw + 2, 32, c, 3) + ynet_conv2d_winograd_supported(b, hw, hw, c, c, 5) + ynet_winograd_filter_floats(c, c);
    EXPECT_REJECT(ynet_winograd_filter(nullptr, fp, 32, 32, 0, 32, nullptr));
    EXPECT_REJECT(ynet_winograd_filter(cfp, fp, 12, 32, 0, 32, nullptr));                                                             // cin not a multiple of 8
    EXPECT_REJECT(ynet_winograd_filter(cfp, fp, 32, 32, 32, 48, nullptr));                                                            // slice beyond the filter
    EXPECT_REJECT(ynet_conv2d_winograd(cfp, 32 * 65536, cfp, nullptr, fp, 32 * 65536, 32, 48, 32, 256, 256, 1, nullptr));               // cout not served
    EXPECT_REJECT(ynet_conv2d_winograd(cfp, 32 * 65536, cfp, nullptr, fp, 32 * 65536, 32, 32, 32, 250, 256, 1, nullptr));               // H not a multiple of 16
    EXPECT_REJECT(ynet_conv2d_winograd(cfp, 16 * 65536, cfp, nullptr, fp, 32 * 65536, 32, 32, 32, 256, 256, 1, nullptr));               // batch stride smaller than the image
    EXPECT_REJECT(ynet_conv2d_winograd(nullptr, 32 * 65536, cfp, nullptr, fp, 32 * 65536, 32, 32, 32, 256, 256, 1, nullptr));
    // the Winograd-native 1-bit ReLU mask
    for (int b : {0, 1, 10, 32})
        for (int hw : {8, 32, 250, 256}) acc += ynet_winograd_relu_bits_words(b, hw, hw) + ynet_winograd_relu_bits_words(b, hw, hw + 16);
    EXPECT_REJECT(ynet_conv2d_winograd_relu_bits(cfp, 32 * 65536, cfp, nullptr, fp, 32 * 65536, 32, 32, 256, 256, nullptr, nullptr));                    // no mask words
    EXPECT_REJECT(ynet_conv2d_winograd_relu_bits(cfp, 32 * 65536, cfp, nullptr, fp, 32 * 65536, 12, 32, 256, 256, (unsigned*)dummy, nullptr));           // cin not a multiple of 8
    EXPECT_REJECT(ynet_conv2d_winograd_relu_bits(cfp, 32 * 65536, cfp, nullptr, fp, 16 * 65536, 32, 32, 256, 256, (unsigned*)dummy, nullptr));           // output stride smaller than the image
    EXPECT_REJECT(ynet_conv2d_winograd_dgrad_relu_bits(cfp, 32 * 65536, cfp, fp, 32 * 65536, nullptr, 32, 32, 256, 256, nullptr));                      // no mask words
    EXPECT_REJECT(ynet_conv2d_winograd_dgrad_relu_bits(cfp, 32 * 65536, cfp, fp, 32 * 65536, (const unsigned*)dummy, 32, 32, 250, 256, nullptr));       // H not a multiple of 16
    EXPECT_REJECT(ynet_conv2d_winograd_dgrad_relu_bits(cfp, 32 * 65536, nullptr, fp, 32 * 65536, (const unsigned*)dummy, 32, 32, 256, 256, nullptr));   // no filters
    {
        const int cat3[3] = {32, 16, 1}, cat2[2] = {32, 32}, bad[2] = {32, 0};
        const long long bs3[3] = {32 * 65536, 16 * 65536, 65536};
        const float* s3[3] = {cfp, cfp, cfp};
        for (int b : {1, 10, 32})
            for (int hw : {32, 128, 256})
                acc += ynet_conv2d_winograd_cat_supported(b, hw, hw, cat3, 3, 32, 3) + ynet_conv2d_winograd_cat_supported(b, hw, hw, cat2, 2, 32, 3) +
                       ynet_conv2d_winograd_cat_supported(b, hw, hw, cat3, 3, 16, 3) + ynet_conv2d_winograd_cat_supported(b, hw, hw, bad, 2, 32, 3) +
                       ynet_winograd_filter_cat_floats(cat3, 3, 32);
        EXPECT_REJECT(ynet_winograd_filter_cat(cfp, fp, cat3, 4, 32, 0, 32, nullptr));                                                   // too many sources
        EXPECT_REJECT(ynet_winograd_filter_cat(cfp, fp, bad, 2, 32, 0, 32, nullptr));                                                    // an empty source
        EXPECT_REJECT(ynet_conv2d_winograd_cat(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, 16, 32, 256, 256, 1, nullptr));              // 16 outputs: not served
        EXPECT_REJECT(ynet_conv2d_winograd_cat(s3, cat2, bs3, 2, cfp, nullptr, fp, 32 * 65536, 32, 32, 256, 256, 1, nullptr));              // 64 inputs: too many filters for LDS
        EXPECT_REJECT(ynet_conv2d_winograd_cat(s3, cat3, bs3, 3, nullptr, nullptr, fp, 32 * 65536, 32, 32, 256, 256, 1, nullptr));          // no filters
        EXPECT_REJECT(ynet_conv2d_winograd_cat_add(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, 32, 32, 256, 256, 1, nullptr, 32 * 65536, 4, nullptr));   // no additive term
        EXPECT_REJECT(ynet_conv2d_winograd_cat_add(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, 32, 32, 256, 256, 1, cfp, 32 * 65536, -1, nullptr));      // negative modulus
        EXPECT_REJECT(ynet_conv2d_winograd_cat_relu_bits(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, 32, 256, 256, nullptr, 0, 0, nullptr, nullptr));              // no mask words
        EXPECT_REJECT(ynet_conv2d_winograd_cat_relu_bits(s3, cat2, bs3, 2, cfp, nullptr, fp, 32 * 65536, 32, 256, 256, nullptr, 0, 0, (unsigned*)dummy, nullptr));     // 64 inputs: too many filters for LDS
        EXPECT_REJECT(ynet_conv2d_winograd_cat_relu_bits(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, 32, 256, 256, cfp, 32 * 65536, -1, (unsigned*)dummy, nullptr)); // negative modulus
        EXPECT_REJECT(ynet_conv2d_winograd_cat_pool(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, nullptr, 32 * 16384, 32, 32, 256, 256, 1, nullptr));   // no pooled output
        EXPECT_REJECT(ynet_conv2d_winograd_cat_pool_code(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, fp, 32 * 16384, nullptr, 32, 256, 256, nullptr));   // no code plane
        EXPECT_REJECT(ynet_conv2d_winograd_cat_pool_code(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, nullptr, 32 * 16384, (unsigned char*)dummy, 32, 256, 256, nullptr));   // no pooled output
        EXPECT_REJECT(ynet_conv2d_winograd_cat_pool_code(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, fp, 16 * 16384, (unsigned char*)dummy, 32, 256, 256, nullptr));     // pooled stride too small
        EXPECT_REJECT(ynet_conv2d_winograd_cat_pool(s3, cat3, bs3, 3, cfp, nullptr, fp, 32 * 65536, fp, 16 * 16384, 32, 32, 256, 256, 1, nullptr));        // pooled stride too small
    }
    {   // the slice form (round 5)
        const int one64[1] = {64}, cat3[3] = {32, 64, 1}, bad[2] = {32, 0};
        const long long bs1[1] = {64 * 4096}, bs3[3] = {32 * 4096, 64 * 4096, 4096};
        const float* s1[1] = {cfp};
        const float* s3[3] = {cfp, cfp, cfp};
        for (int b : {1, 10, 32, 256})
            for (int hw : {32, 64, 128, 256})
                for (int co : {16, 32, 48, 64, 128})
                    acc += ynet_conv2d_winograd16_supported(b, hw, hw, one64, 1, co, 3) + ynet_conv2d_winograd16_supported(b, hw, hw, cat3, 3, co, 3) +
                           ynet_conv2d_winograd16_supported(b, hw, hw + 16, one64, 1, co, 3) + ynet_conv2d_winograd16_supported(b, hw, hw, bad, 2, co, 3) +
                           ynet_conv2d_winograd16_supported(b, hw, hw, one64, 1, co, 5) + ynet_winograd16_filter_floats(cat3, 3, co);
        EXPECT_REJECT(ynet_winograd16_filter(cfp, fp, cat3, 4, 64, 0, 64, nullptr));                                                     // too many sources
        EXPECT_REJECT(ynet_winograd16_filter(cfp, fp, one64, 1, 24, 0, 64, nullptr));                                                    // cout not a multiple of 16
        EXPECT_REJECT(ynet_winograd16_filter(cfp, fp, one64, 1, 64, 32, 64, nullptr));                                                   // slice beyond the filter
        EXPECT_REJECT(ynet_conv2d_winograd16(s3, cat3, bs3, 3, cfp, nullptr, fp, 64 * 4096, 64, 32, 64, 64, 1, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr));   // 97 -> 100 padded channels: too many filters for LDS
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, cfp, nullptr, fp, 64 * 4096, 48, 32, 64, 64, 1, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr));  // 48 outputs: not a divisor of an XCD's workgroups
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, cfp, nullptr, fp, 64 * 4096, 64, 32, 48, 64, 1, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr));  // H not a multiple of 32
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, cfp, nullptr, fp, 32 * 4096, 64, 32, 64, 64, 1, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr));  // output stride smaller than the image
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, cfp, nullptr, fp, 64 * 4096, 64, 32, 64, 64, 0, cfp, 64 * 4096, cfp, 64 * 4096, 0, nullptr, 0, nullptr));  // two epilogue variants
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, cfp, cfp, fp, 64 * 4096, 64, 32, 64, 64, 0, cfp, 64 * 4096, nullptr, 0, 0, nullptr, 0, nullptr));         // relu_of with a bias
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, cfp, nullptr, fp, 64 * 4096, 64, 32, 64, 64, 1, nullptr, 0, cfp, 64 * 4096, -1, nullptr, 0, nullptr));    // negative modulus
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, cfp, nullptr, fp, 64 * 4096, 64, 32, 64, 64, 1, nullptr, 0, nullptr, 0, 0, fp, 16 * 1024, nullptr));       // pooled stride too small
        EXPECT_REJECT(ynet_conv2d_winograd16(s1, one64, bs1, 1, nullptr, nullptr, fp, 64 * 4096, 64, 32, 64, 64, 1, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr));      // no filters
    }
    for (int b : {1, 10, 32})
        for (int hw : {32, 128, 256})
            for (int c : {16, 32, 64}) acc += ynet_upsample2x_conv2d_winograd_supported(b, hw, hw, c, 16, 3) + ynet_upsample2x_conv2d_winograd_supported(b, hw, hw + 8, 32, c, 3);
    EXPECT_REJECT(ynet_upsample2x_conv2d_winograd(cfp, 32 * 16384, cfp, nullptr, fp, 16 * 65536, 32, 32, 32, 256, 256, 0, nullptr));     // 32 outputs: not served
    EXPECT_REJECT(ynet_upsample2x_conv2d_winograd(cfp, 32 * 16384, cfp, nullptr, fp, 16 * 65536, 32, 16, 32, 250, 256, 0, nullptr));     // H not a multiple of 16
    EXPECT_REJECT(ynet_upsample2x_conv2d_winograd(cfp, 16 * 16384, cfp, nullptr, fp, 16 * 65536, 32, 16, 32, 256, 256, 0, nullptr));     // input stride smaller than the low-resolution image
    EXPECT_REJECT(ynet_upsample2x_conv2d_winograd(cfp, 32 * 16384, cfp, nullptr, fp, 8 * 65536, 32, 16, 32, 256, 256, 0, nullptr));      // output stride smaller than the image
    EXPECT_REJECT(ynet_upsample2x_conv2d_winograd(nullptr, 32 * 16384, cfp, nullptr, fp, 16 * 65536, 32, 16, 32, 256, 256, 0, nullptr));
    EXPECT_REJECT(ynet_conv2d_winograd_dgrad_relu(cfp, 32 * 65536, cfp, fp, 32 * 65536, nullptr, 32 * 65536, 32, 32, 32, 256, 256, nullptr));   // no activation
    EXPECT_REJECT(ynet_conv2d_winograd_dgrad_relu(cfp, 32 * 65536, cfp, fp, 32 * 65536, cfp, 16 * 65536, 32, 32, 32, 256, 256, nullptr));       // activation stride too small
    EXPECT_REJECT(ynet_conv2d_wgrad(srcs, &one, &bs, 1, nullptr, 0, nullptr, 0, fp, nullptr, fp, 1, 4, 4, 1, 3, nullptr));
    EXPECT_REJECT(ynet_conv2d_wgrad(srcs, &one, &bs, 7, cfp, 4, nullptr, 0, fp, nullptr, fp, 1, 4, 4, 1, 3, nullptr));
    EXPECT_REJECT(ynet_lora_compose(nullptr, cfp, cfp, 1.f, fp, 4, 4, 3, 1, nullptr));
    EXPECT_REJECT(ynet_lora_grad(cfp, cfp, cfp, 1.f, fp, nullptr, 4, 4, 3, 1, nullptr));
    EXPECT_REJECT(ynet_lora_compose_pack(cfp, cfp, cfp, 1.f, fp, fp, 4, 4, 2, 1, nullptr));
    EXPECT_REJECT(ynet_train_readout(cfp, 16, nullptr, 16, 0, cfp, fp, fp, fp, fp, 1, 1, 4, 4, 1.f, nullptr));           // no goal map
    EXPECT_REJECT(ynet_train_readout(cfp, 16, cfp, 16, 0, cfp, fp, fp, fp, fp, 1, 1, 4, 4, 0.f, nullptr));               // resize factor 0
    for (int cin : {1, 6, 14, 32, 33, 64, 65})
        for (int cout : {1, 16, 32, 64, 130})
            for (int r : {1, 4}) acc += ynet_lora_conv2d_wgrad_supported(cin, cout, 3, r, 64) + ynet_lora_conv2d_wgrad_preferred(cin, cout, 3, r, 64) + ynet_lora_conv2d_wgrad_workspace_floats(cin, cout);
    EXPECT_REJECT(ynet_lora_conv2d_wgrad(srcs, &one, &bs, 1, nullptr, 4, nullptr, 0, cfp, cfp, 1.f, fp, fp, fp, 1, 4, 4, 4, 3, 1, nullptr));   // no dy
    EXPECT_REJECT(ynet_lora_conv2d_wgrad(srcs, &one, &bs, 1, cfp, 4, nullptr, 0, cfp, cfp, 1.f, fp, fp, fp, 1, 4, 4, 4, 3, 4, nullptr));       // rank 4
    EXPECT_REJECT(ynet_lora_conv2d_wgrad(srcs, &one, &bs, 1, cfp, 4, nullptr, 0, cfp, cfp, 1.f, fp, fp, fp, 1, 4, 6, 4, 3, 1, nullptr));       // W % 4
    EXPECT_REJECT(ynet_lora_compose_pack_multi(0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    EXPECT_REJECT(ynet_lora_compose_pack_multi(99, srcs, srcs, srcs, cfp, dsts, dsts, &one, &one, &one, &one, nullptr));
    EXPECT_REJECT(ynet_adam_step(nullptr, &one, (const long long*)dummy, 1, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, nullptr));     // no table
    EXPECT_REJECT(ynet_adam_step((const long long*)dummy, &one, (const long long*)dummy, 1, 1, 1e-3, 1.5, 0.999, 1e-8, 0.0, 0, nullptr));   // beta1 >= 1
    EXPECT_REJECT(ynet_batch_sum(cfp, fp, 2, 6, 6, nullptr));           // n % 4
    EXPECT_REJECT(ynet_batch_sum(nullptr, fp, 2, 8, 8, nullptr));
    EXPECT_REJECT(ynet_maxpool2_fwd(nullptr, fp, 1, 4, 4, nullptr));
    EXPECT_REJECT(ynet_maxpool2_bwd(cfp, cfp, nullptr, 1, 4, 4, nullptr));
    EXPECT_REJECT(ynet_upsample2x_fwd(cfp, nullptr, 1, 2, 2, nullptr));
    EXPECT_REJECT(ynet_upsample2x_bwd(nullptr, fp, 1, 2, 2, nullptr));
    EXPECT_REJECT(ynet_upsample2x_bwd_relu(cfp, fp, nullptr, 1, 2, 2, nullptr));
    EXPECT_REJECT(ynet_avgpool_pyramid(cfp, dsts, 9, 1, 32, 32, nullptr));
    EXPECT_REJECT(ynet_bce_logits_fwd(cfp, cfp, 0, fp, fp, nullptr));
    EXPECT_REJECT(ynet_bce_logits_fwd_grad(cfp, nullptr, 4, 1.f, fp, fp, fp, nullptr));
    EXPECT_REJECT(ynet_maxpool2_bwd_add_code(nullptr, cfp, nullptr, nullptr, fp, 4, 8, 8, 1, nullptr));                          // no code plane
    EXPECT_REJECT(ynet_maxpool2_bwd_add_code((const unsigned char*)dummy, cfp, nullptr, nullptr, fp, 4, 7, 8, 1, nullptr));      // odd H
    EXPECT_REJECT(ynet_pred_bce(cfp, 4, cfp, nullptr, cfp, fp, fp, nullptr, nullptr, fp, 1, 4, 33, 4, 1.f, 0, nullptr));   // cout 33
    EXPECT_REJECT(ynet_pred_bce(cfp, 4, cfp, nullptr, cfp, fp, fp, nullptr, nullptr, fp, 1, 4, 12, 6, 1.f, 0, nullptr));   // HW % 4
    EXPECT_REJECT(ynet_pred_bce(cfp, 4, cfp, nullptr, nullptr, fp, fp, nullptr, nullptr, fp, 1, 4, 12, 4, 1.f, 0, nullptr));                            // no target
    EXPECT_REJECT(ynet_pred_bce_blob(cfp, 4096, cfp, nullptr, nullptr, cfp, 31, 400, 64, 64, fp, fp, nullptr, nullptr, fp, 1, 4, 12, 1.f, 0, nullptr));    // no positions
    EXPECT_REJECT(ynet_pred_bce_blob(cfp, 4096, cfp, nullptr, cfp, nullptr, 31, 400, 64, 64, fp, fp, nullptr, nullptr, fp, 1, 4, 12, 1.f, 0, nullptr));    // no blob table
    EXPECT_REJECT(ynet_pred_bce_blob(cfp, 4096, cfp, nullptr, cfp, cfp, 31, 16, 64, 64, fp, fp, nullptr, nullptr, fp, 1, 4, 12, 1.f, 0, nullptr));         // template smaller than the window
    EXPECT_REJECT(ynet_pred_bce_blob(cfp, 4096, cfp, nullptr, cfp, cfp, 31, 400, 64, 62, fp, fp, nullptr, nullptr, fp, 1, 4, 12, 1.f, 0, nullptr));        // W % 4
    EXPECT_REJECT(ynet_pred_bce_blob(cfp, 4096, cfp, nullptr, cfp, cfp, 0, 400, 64, 64, fp, fp, nullptr, nullptr, fp, 1, 4, 12, 1.f, 0, nullptr));         // empty blob
    EXPECT_REJECT(ynet_softargmax2d(nullptr, fp, 1, 1, 4, 2, 2, nullptr));
    EXPECT_REJECT(ynet_pred_softargmax(nullptr, 0, nullptr, nullptr, nullptr, nullptr, 1, 32, 12, 16, 32, nullptr));
    EXPECT_REJECT(ynet_pred_softargmax(fp, 32 * 17 * 23, fp, nullptr, fp, fp, 1, 32, 12, 17, 23, nullptr));      /* H*W % 128 != 0 */
    if (ynet_pred_softargmax_supported(24, 12, 16, 32) || ynet_pred_softargmax_supported(32, 33, 16, 32) || !ynet_pred_softargmax_supported(32, 30, 256, 256)) { fprintf(stderr, "pred_softargmax_supported\n"); return 1; }
    if (ynet_pred_softargmax_workspace_floats(4, 256, 256) != 4ll * 32 * 128) { fprintf(stderr, "pred_softargmax workspace\n"); return 1; }
    EXPECT_REJECT(ynet_sigmoid_temp(cfp, fp, 1, 4, 4, &one, 9, 1.f, nullptr));
    EXPECT_REJECT(ynet_sigmoid_temp(cfp, fp, 1, 4, 4, &one, 1, 0.f, nullptr));
    EXPECT_REJECT(ynet_gather_patch(cfp, 8, 8, cfp, fp, 1, 16, 16, &status, nullptr));      // window larger than the template
    EXPECT_REJECT(ynet_heatmap_analytic(cfp, fp, 1, 4, 4, 8, 0, 0.0, nullptr, 0, &status, nullptr));
    EXPECT_REJECT(ynet_heatmap_analytic(cfp, fp, 1, 4, 4, 8, 1, 1.0, nullptr, 31, &status, nullptr));
    EXPECT_REJECT(ynet_kmeans2d(cfp, &one, fp, &status, 1, 20000, 4, 1e-3f, 10, nullptr));
    EXPECT_REJECT(ynet_multinomial(cfp, 1, 4, 4, 99, 0, 0.f, 1ull, (long long*)dummy, &status, nullptr));
    EXPECT_REJECT(ynet_multinomial(cfp, 1, 4, 4, 1, 0, 2.f, 1ull, (long long*)dummy, &status, nullptr));
    EXPECT_REJECT(ynet_cws_prior(cfp, 4, 1, cfp, cfp, 1, 2, 2, 0.f, 1.f, 0, fp, fp, nullptr));
    EXPECT_REJECT(ynet_resize_nearest(nullptr, (int*)dummy, 4, 4, 2, 2, 0.5, 0.5, nullptr));
    EXPECT_REJECT(ynet_resize_nearest((const int*)dummy, (int*)dummy, 4, 4, 2, 2, 0.0, 0.5, nullptr));      // factor 0
    EXPECT_REJECT(ynet_upconv_dgrad_ring(nullptr, 0, cfp, nullptr, 0, fp, 0, 1, 64, 32, 8, 8, nullptr));
    EXPECT_REJECT(ynet_upconv_dgrad_ring(cfp, 64 * 64, cfp, nullptr, 0, fp, 32 * 64, 1, 62, 32, 8, 8, nullptr));        // C4 not a multiple of 4
    EXPECT_REJECT(ynet_upconv_dgrad_ring(cfp, 8, cfp, nullptr, 0, fp, 32 * 64, 1, 64, 32, 8, 8, nullptr));               // batch stride below the image
    EXPECT_REJECT(ynet_upconv_dgrad_ring(cfp, 512 * 64, cfp, nullptr, 0, fp, 64 * 64, 1, 512, 64, 8, 8, nullptr));       // tables beyond 64 KB of LDS
    EXPECT_REJECT(ynet_conv2d_winograd_s2d(nullptr, 0, cfp, fp, 0, 32, 16, 8, 256, 256, nullptr));
    EXPECT_REJECT(ynet_conv2d_winograd_s2d(cfp, 32ll * 64 * 64, cfp, fp, 16ll * 64 * 64, 32, 16, 1, 64, 64, nullptr));   // too few pixels for the Winograd kernels
    // the [16, 32]-channel data gradient in one launch and the last decoder convolution inside the predictor + criterion (round 6)
    if (ynet_conv2d_winograd_split_supported(8, 256, 256, 32) != 1 || ynet_conv2d_winograd_split_supported(8, 256, 256, 16) != 0) ++failures;
    EXPECT_REJECT(ynet_conv2d_winograd_split(nullptr, 0, cfp, fp, 0, 0, fp, 0, 32, 8, 256, 256, nullptr));
    EXPECT_REJECT(ynet_conv2d_winograd_split(cfp, 32ll * 65536, cfp, fp, 16ll * 65536, 1, nullptr, 0, 32, 8, 256, 256, nullptr));      // no second destination
    EXPECT_REJECT(ynet_conv2d_winograd_split(cfp, 32ll * 65536, cfp, fp, 16ll * 65536, 0, fp, 8, 32, 8, 256, 256, nullptr));            // its batch stride below the image
    if (ynet_conv2d_winograd_pred_bce_supported(32, 256, 256, 32, 32, 12, 31) != 1 || ynet_conv2d_winograd_pred_bce_supported(16, 512, 512, 32, 32, 30, 31) != 1 || ynet_conv2d_winograd_pred_bce_supported(32, 256, 256, 32, 32, 33, 31) != 0 ||
        ynet_conv2d_winograd_pred_bce_supported(512, 256, 256, 32, 32, 12, 31) != 0 || ynet_conv2d_winograd_pred_bce_supported(8, 256, 256, 16, 32, 12, 31) != 0)
        ++failures;      // (<= 32 predictor outputs, tables within LDS, 32 -> 32)
    EXPECT_REJECT(ynet_conv2d_winograd_pred_bce_blob(nullptr, 0, cfp, cfp, cfp, cfp, 12, cfp, cfp, 31, 400, fp, fp, fp, 0, fp, 8, 256, 256, 1.f, nullptr));
    EXPECT_REJECT(ynet_conv2d_winograd_pred_bce_blob(cfp, 32ll * 65536, cfp, cfp, cfp, cfp, 33, cfp, cfp, 31, 400, fp, fp, fp, 32ll * 65536, fp, 8, 256, 256, 1.f, nullptr));   // 33 outputs
    EXPECT_REJECT(ynet_conv2d_winograd_pred_bce_blob(cfp, 32ll * 65536, cfp, cfp, cfp, cfp, 12, cfp, cfp, 31, 128, fp, fp, fp, 32ll * 65536, fp, 8, 256, 256, 1.f, nullptr));   // template smaller than the window
    EXPECT_REJECT(ynet_batchnorm2d_fwd(nullptr, fp, nullptr, nullptr, nullptr, nullptr, fp, fp, nullptr, 1, 4, 16, 1, 0.1, 1e-5, nullptr));
    EXPECT_REJECT(ynet_batchnorm2d_fwd(cfp, fp, nullptr, nullptr, nullptr, nullptr, fp, fp, nullptr, 1, 4, 16, 1, 0.1, 1e-5, nullptr));      // training mode without a workspace
    EXPECT_REJECT(ynet_batchnorm2d_fwd(cfp, fp, nullptr, nullptr, nullptr, nullptr, fp, fp, nullptr, 1, 4, 16, 0, 0.1, 1e-5, nullptr));      // evaluation mode without running statistics
    EXPECT_REJECT(ynet_batchnorm2d_bwd(cfp, cfp, cfp, cfp, nullptr, fp, nullptr, nullptr, nullptr, 1, 4, 16, 1, nullptr));
    EXPECT_REJECT(ynet_add_relu(nullptr, cfp, fp, 4, 1, nullptr));
    EXPECT_REJECT(ynet_relu_bwd(cfp, cfp, nullptr, 4, nullptr));
    if (ynet_batchnorm_workspace_doubles(16) != 2ll * 16 * 64 || ynet_batchnorm_workspace_doubles(0) != 0) { fprintf(stderr, "batchnorm workspace\n"); ++failures; }
    EXPECT_REJECT(ynet_rot90_flip(nullptr, dummy, 1, 4, 4, 1, 0, nullptr));
    EXPECT_REJECT(ynet_rot90_flip(dummy, dummy, 1, 4, 4, 1, 0, nullptr));      // in place
    EXPECT_REJECT(ynet_rot_coords(nullptr, 4, 0, 0, 1, 0, 0, 1, 0, 0, nullptr));
    {   // ynet_conv2d_auto: the host half -- descriptor checks, the plan, the cache size -- without a launch
        EXPECT_REJECT(ynet_conv2d_auto(nullptr, nullptr, nullptr));
        YnetConvAuto d;
        memset(&d, 0, sizeof(d));
        EXPECT_REJECT(ynet_conv2d_auto(&d, nullptr, nullptr));                   // no sources / filter
        d.nsrc = 1; d.ndst = 1; d.src[0] = cfp; d.src_c[0] = 32; d.src_bs[0] = 32ll * 256 * 256; d.wp = cfp;
        d.dst[0] = fp; d.dst_c[0] = 32; d.dst_bs[0] = 32ll * 256 * 256; d.B = 8; d.H = 256; d.W = 256; d.K = 3; d.relu = 1;
        const long long need = ynet_conv2d_auto_cache_floats(&d);
        if (need != ynet_winograd_filter_floats(32, 32)) { fprintf(stderr, "conv2d_auto cache floats %lld\n", need); ++failures; }
        EXPECT_REJECT(ynet_conv2d_auto(&d, nullptr, nullptr));                   // a Winograd plan without a cache
        d.flags = YNET_AUTO_NO_WINOGRAD;
        if (ynet_conv2d_auto_cache_floats(&d) != 0) { fprintf(stderr, "conv2d_auto: implicit GEMM needs no cache\n"); ++failures; }
        d.flags = 0; d.K = 4;
        EXPECT_REJECT(ynet_conv2d_auto(&d, nullptr, nullptr));                   // K
        d.K = 3; d.upsample2x = 1; d.src_c[0] = 8; d.H = 32; d.W = 32;
        EXPECT_REJECT(ynet_conv2d_auto(&d, nullptr, nullptr));                   // no up-convolution kernel serves 8 channels at 32^2
        d.upsample2x = 0; d.H = 256; d.W = 256; d.src_c[0] = 32; d.addend = cfp; d.pooled = fp;
        EXPECT_REJECT(ynet_conv2d_auto(&d, nullptr, nullptr));                   // addend + pooled
        d.addend = nullptr; d.dst_c[0] = 64; d.dst_bs[0] = 64ll * 256 * 256; d.pooled = nullptr;
        if (ynet_conv2d_auto_cache_floats(&d) <= 0) { fprintf(stderr, "conv2d_auto: 32 -> 64 at 256^2 is a Winograd plan\n"); ++failures; }
        if (ynet_conv2d_auto_workspace_floats(&d) != 0) { fprintf(stderr, "conv2d_auto: no workspace for a large map\n"); ++failures; }
    }
    void* comm = nullptr;
    EXPECT_REJECT(ynet_comm_create(3, 2, 16, &comm));
    EXPECT_REJECT(ynet_comm_create(0, 99, 16, &comm));
    EXPECT_REJECT(ynet_comm_export(nullptr, dummy));
    EXPECT_REJECT(ynet_comm_connect(nullptr, dummy));
    EXPECT_REJECT(ynet_allreduce_sum(nullptr, fp, 4, nullptr));
    if (ynet_comm_destroy(nullptr) != 0) ++failures;
    printf("asan host driver: %d failures, checksum %lld\n", failures, acc);
    return failures ? 1 : 0;
}
