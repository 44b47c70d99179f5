// Shared declarations for the gfx950 Y-Net kernels (internal; the public C ABI is include/ynet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define YNET_MAX_SRC 4

// One operand of a (virtual) channel concatenation: NCHW fp32, `c` channels, batch stride `bs`
// elements (0 = the same image for every batch item, e.g. the semantic map of a scene).
struct YSrc {
    const float* p;
    int c;
    long long bs;
    int bmod;   // > 0: the source holds bmod images that repeat along the batch (image = b % bmod)
};
struct YDst {
    float* p;   // may be NULL: channels are computed but not stored (no gradient wanted)
    int c;
    long long bs;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void ynet_set_error(const char* fmt, ...);
int ynet_check_launch(const char* what);

#define YNET_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            ynet_set_error(__VA_ARGS__);     \
            return 1;                        \
        }                                    \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Per-device launch state (function attributes, resident-workgroup counts) is cached per HIP device: a process
// that drives several GPUs sets the > 64 KB LDS attribute and sizes its persistent grids on each of them.
#define YNET_MAX_DEV 16
static inline int ynet_device_slot() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (d >= 0 && d < YNET_MAX_DEV) ? d : 0;
}
