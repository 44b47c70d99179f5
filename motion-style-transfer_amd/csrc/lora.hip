// MoSA / LoRA adapter algebra on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Restates loralib==0.1.1 `Conv2d` (call site models/ynet.py:141-144; third-party, not in the
// reference tree -> "parity unpinned", see oracle/_stubs/loralib):
//   W_eff = W + (lora_B @ lora_A).view(W.shape) * (lora_alpha / r)          (compose, K5)
//   dA = s * B^T @ dWm,   dB = s * dWm @ A^T,   dWm = dW.view(Cout*k, Cin*k)  (grad,    K6)
// The .view is a FLAT reshape of the [Cout*k, Cin*k] product, so both are plain row-major GEMMs
// over the flat weight.  These are the "dense LoRA down/up projections" of the north star; they
// are tiny (inner dimension r*k = 3..12), one wavefront per 16x16 output tile, operands straight
// from global memory (L2-resident).
#include "ynet_common.h"

struct GemmArgs {
    const float* A;   // element (m,k) at A[m*sam + k*sak]
    const float* B;   // element (k,n) at B[k*sbk + n*sbn]
    const float* D;   // optional addend, row-major [M][N]
    float* C;         // row-major [M][N]:  C = alpha * A@B + D
    int M, N, K;
    long long sam, sak, sbk, sbn;
    float alpha;
};

__global__ __launch_bounds__(64) void small_gemm_mfma_kernel(const GemmArgs g) {
    const int lane = threadIdx.x;
    const int tiles_n = (g.N + 15) / 16;
    const int m0 = (blockIdx.x / tiles_n) * 16, n0 = (blockIdx.x % tiles_n) * 16;
    const int r = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int am = m0 + r, bn = n0 + r;
    for (int k0 = 0; k0 < g.K; k0 += 4) {
        const int k = k0 + kq;
        const float a = (am < g.M && k < g.K) ? g.A[am * g.sam + k * g.sak] : 0.f;
        const float b = (bn < g.N && k < g.K) ? g.B[k * g.sbk + bn * g.sbn] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + kq * 4 + q, n = n0 + r;
        if (m < g.M && n < g.N) {
            const long long i = (long long)m * g.N + n;
            g.C[i] = g.alpha * acc[q] + (g.D ? g.D[i] : 0.f);
        }
    }
}

// Two independent small GEMMs in one launch (blocks [0, tiles0) -> g0, the rest -> g1): dA and dB of one layer.
__global__ __launch_bounds__(64) void small_gemm2_mfma_kernel(const GemmArgs g0, const GemmArgs g1, int tiles0) {
    const bool first = (int)blockIdx.x < tiles0;
    const GemmArgs& g = first ? g0 : g1;
    const int blk = first ? (int)blockIdx.x : (int)blockIdx.x - tiles0;
    const int lane = threadIdx.x;
    const int tiles_n = (g.N + 15) / 16;
    const int m0 = (blk / tiles_n) * 16, n0 = (blk % tiles_n) * 16;
    const int r = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int am = m0 + r, bn = n0 + r;
    // the operand loads of 8 K-steps are issued together: one round trip to L2 per 8 MFMAs instead of one per MFMA
    // (K = Cout*k = 96 .. 192 here: 48 dependent round trips took 16-20 us per layer, on the critical path of the
    // encoder's backward pass)
    for (int k0 = 0; k0 < g.K; k0 += 32) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            a[u] = (am < g.M && k < g.K) ? g.A[am * g.sam + k * g.sak] : 0.f;
            b[u] = (bn < g.N && k < g.K) ? g.B[k * g.sbk + bn * g.sbn] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + kq * 4 + q, n = n0 + r;
        if (m < g.M && n < g.N) g.C[(long long)m * g.N + n] = g.alpha * acc[q];
    }
}

// W_eff = W + s * (B @ A) computed tile by tile and written straight into BOTH packed filter layouts of the conv
// kernels (forward [ci][tap][co] and dgrad [co][flipped tap][ci], see conv_mfma.hip: pack_weight_kernel); the zero
// padding of the packed buffers is the caller's (they are allocated zeroed once per layer and reused).
struct ComposePackArgs {
    const float* w;
    const float* lora_a;
    const float* lora_b;
    float* wp_fwd;
    float* wp_dgrad;
    int cout, cin, KK, K, r;
    int fwd_cols, dgrad_cols;      // padded column counts of the two packed layouts
    float scale;
};

__global__ __launch_bounds__(64) void lora_compose_pack_kernel(const ComposePackArgs g) {
    const int M = g.cout * g.K, N = g.cin * g.K, Kd = g.r * g.K;
    const int lane = threadIdx.x;
    const int tiles_n = (N + 15) / 16;
    const int m0 = ((int)blockIdx.x / tiles_n) * 16, n0 = ((int)blockIdx.x % tiles_n) * 16;
    const int r = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int am = m0 + r, bn = n0 + r;
    for (int k0 = 0; k0 < Kd; k0 += 4) {
        const int k = k0 + kq;
        const float a = (am < M && k < Kd) ? g.lora_b[am * Kd + k] : 0.f;
        const float b = (bn < N && k < Kd) ? g.lora_a[k * N + bn] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + kq * 4 + q, n = n0 + r;
        if (m < M && n < N) {
            const int flat = m * N + n;                       // index into the flat [Cout][Cin][K][K] weight
            const float v = g.scale * acc[q] + g.w[flat];
            const int co = flat / (g.cin * g.KK), rem = flat - co * g.cin * g.KK;
            const int ci = rem / g.KK, t = rem - ci * g.KK;
            g.wp_fwd[((long long)ci * g.KK + t) * g.fwd_cols + co] = v;
            g.wp_dgrad[((long long)co * g.KK + (g.KK - 1 - t)) * g.dgrad_cols + ci] = v;
        }
    }
}

// All adapted convs of a model in ONE launch (blockIdx.y = layer): the 9 composes of a mosa_* encoder are ~5 us launches
// each; issued one by one in front of their convs they sit on the critical path of the encoder's forward pass.  Rank 0
// (no adapter: lora_a / lora_b NULL) packs the plain weight -- the 46 trainable convs of train_net = train / all are
// re-packed after every optimizer step, 2 launches each when left to ynet_pack_weight.
#define YNET_LORA_MULTI_MAX 48
struct ComposePackMulti {
    ComposePackArgs layer[YNET_LORA_MULTI_MAX];
    int tiles[YNET_LORA_MULTI_MAX];
};
static_assert(sizeof(ComposePackMulti) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(64) void lora_compose_pack_multi_kernel(const ComposePackMulti mm) {
    const ComposePackArgs& g = mm.layer[blockIdx.y];
    if ((int)blockIdx.x >= mm.tiles[blockIdx.y]) return;
    const int M = g.cout * g.K, N = g.cin * g.K, Kd = g.r * g.K;
    const int lane = threadIdx.x;
    const int tiles_n = (N + 15) / 16;
    const int m0 = ((int)blockIdx.x / tiles_n) * 16, n0 = ((int)blockIdx.x % tiles_n) * 16;
    const int r = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int am = m0 + r, bn = n0 + r;
    for (int k0 = 0; k0 < Kd; k0 += 4) {
        const int k = k0 + kq;
        const float a = (am < M && k < Kd) ? g.lora_b[am * Kd + k] : 0.f;
        const float b = (bn < N && k < Kd) ? g.lora_a[k * N + bn] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + kq * 4 + q, n = n0 + r;
        if (m < M && n < N) {
            const int flat = m * N + n;
            const float v = g.scale * acc[q] + g.w[flat];
            const int co = flat / (g.cin * g.KK), rem = flat - co * g.cin * g.KK;
            const int ci = rem / g.KK, t = rem - ci * g.KK;
            g.wp_fwd[((long long)ci * g.KK + t) * g.fwd_cols + co] = v;
            g.wp_dgrad[((long long)co * g.KK + (g.KK - 1 - t)) * g.dgrad_cols + ci] = v;
        }
    }
}

static int run_gemm(const GemmArgs& g, hipStream_t st, const char* what) {
    const int tiles = ((g.M + 15) / 16) * ((g.N + 15) / 16);
    hipLaunchKernelGGL(small_gemm_mfma_kernel, dim3(tiles), dim3(64), 0, st, g);
    return ynet_check_launch(what);
}

extern "C" {

int ynet_lora_compose(const float* w, const float* lora_a, const float* lora_b, float scale, float* w_eff,
                      int cout, int cin, int K, int r, void* stream) {
    YNET_REQUIRE(w && lora_a && lora_b && w_eff, "lora_compose: null pointer");
    YNET_REQUIRE(cout > 0 && cin > 0 && K > 0 && r > 0, "lora_compose: bad shape");
    GemmArgs g{};
    g.M = cout * K;
    g.N = cin * K;
    g.K = r * K;
    g.A = lora_b;   // [M][Kd]
    g.sam = g.K;
    g.sak = 1;
    g.B = lora_a;   // [Kd][N]
    g.sbk = g.N;
    g.sbn = 1;
    g.D = w;
    g.C = w_eff;
    g.alpha = scale;
    return run_gemm(g, (hipStream_t)stream, "lora_compose");
}

int ynet_lora_grad(const float* dw, const float* lora_a, const float* lora_b, float scale, float* d_a, float* d_b,
                   int cout, int cin, int K, int r, void* stream) {
    YNET_REQUIRE(dw && lora_a && lora_b && d_a && d_b, "lora_grad: null pointer");
    YNET_REQUIRE(cout > 0 && cin > 0 && K > 0 && r > 0, "lora_grad: bad shape");
    const int M = cout * K, N = cin * K, Kd = r * K;
    GemmArgs ga{};   // dA[Kd][N] = s * B^T[Kd][M] @ dWm[M][N]
    ga.M = Kd;
    ga.N = N;
    ga.K = M;
    ga.A = lora_b;
    ga.sam = 1;
    ga.sak = Kd;
    ga.B = dw;
    ga.sbk = N;
    ga.sbn = 1;
    ga.C = d_a;
    ga.alpha = scale;
    GemmArgs gb{};   // dB[M][Kd] = s * dWm[M][N] @ A^T[N][Kd]
    gb.M = M;
    gb.N = Kd;
    gb.K = N;
    gb.A = dw;
    gb.sam = N;
    gb.sak = 1;
    gb.B = lora_a;
    gb.sbk = 1;
    gb.sbn = N;
    gb.C = d_b;
    gb.alpha = scale;
    const int tiles_a = ((ga.M + 15) / 16) * ((ga.N + 15) / 16), tiles_b = ((gb.M + 15) / 16) * ((gb.N + 15) / 16);
    hipLaunchKernelGGL(small_gemm2_mfma_kernel, dim3(tiles_a + tiles_b), dim3(64), 0, (hipStream_t)stream, ga, gb, tiles_a);
    return ynet_check_launch("lora_grad");
}

int ynet_lora_compose_pack(const float* w, const float* lora_a, const float* lora_b, float scale, float* wp_fwd,
                           float* wp_dgrad, int cout, int cin, int K, int r, void* stream) {
    YNET_REQUIRE(w && lora_a && lora_b && wp_fwd && wp_dgrad, "lora_compose_pack: null pointer");
    YNET_REQUIRE(cout > 0 && cin > 0 && (K == 1 || K == 3 || K == 5) && r > 0, "lora_compose_pack: bad shape");
    ComposePackArgs g{};
    g.w = w;
    g.lora_a = lora_a;
    g.lora_b = lora_b;
    g.wp_fwd = wp_fwd;
    g.wp_dgrad = wp_dgrad;
    g.cout = cout;
    g.cin = cin;
    g.K = K;
    g.KK = K * K;
    g.r = r;
    g.fwd_cols = (cout + 63) / 64 * 64;        // YNET_COUT_PAD of conv_mfma.hip
    g.dgrad_cols = (cin + 63) / 64 * 64;
    g.scale = scale;
    const int tiles = ((cout * K + 15) / 16) * ((cin * K + 15) / 16);
    hipLaunchKernelGGL(lora_compose_pack_kernel, dim3(tiles), dim3(64), 0, (hipStream_t)stream, g);
    return ynet_check_launch("lora_compose_pack");
}

// ynet_lora_compose_pack for `n` layers (n <= 48) in one launch; every argument is a HOST array of n entries.
// r[i] == 0: layer i has no adapter (lora_a[i] / lora_b[i] may be NULL), its weight is packed as it is.
int ynet_lora_compose_pack_multi(int n, const float* const* w, const float* const* lora_a, const float* const* lora_b,
                                 const float* scale, float* const* wp_fwd, float* const* wp_dgrad, const int* cout,
                                 const int* cin, const int* K, const int* r, void* stream) {
    YNET_REQUIRE(n >= 1 && n <= YNET_LORA_MULTI_MAX, "lora_compose_pack_multi: 1..%d layers per call (got %d)", YNET_LORA_MULTI_MAX, n);
    YNET_REQUIRE(w && lora_a && lora_b && scale && wp_fwd && wp_dgrad && cout && cin && K && r, "lora_compose_pack_multi: null pointer");
    ComposePackMulti mm{};
    int max_tiles = 0;
    for (int i = 0; i < n; ++i) {
        YNET_REQUIRE(w[i] && wp_fwd[i] && wp_dgrad[i] && (r[i] == 0 || (lora_a[i] && lora_b[i])), "lora_compose_pack_multi: null pointer (layer %d)", i);
        YNET_REQUIRE(cout[i] > 0 && cin[i] > 0 && (K[i] == 1 || K[i] == 3 || K[i] == 5) && r[i] >= 0, "lora_compose_pack_multi: bad shape (layer %d)", i);
        ComposePackArgs& g = mm.layer[i];
        g.w = w[i];
        g.lora_a = lora_a[i];
        g.lora_b = lora_b[i];
        g.wp_fwd = wp_fwd[i];
        g.wp_dgrad = wp_dgrad[i];
        g.cout = cout[i];
        g.cin = cin[i];
        g.K = K[i];
        g.KK = K[i] * K[i];
        g.r = r[i];
        g.fwd_cols = (cout[i] + 63) / 64 * 64;
        g.dgrad_cols = (cin[i] + 63) / 64 * 64;
        g.scale = scale[i];
        mm.tiles[i] = ((cout[i] * K[i] + 15) / 16) * ((cin[i] * K[i] + 15) / 16);
        if (mm.tiles[i] > max_tiles) max_tiles = mm.tiles[i];
    }
    hipLaunchKernelGGL(lora_compose_pack_multi_kernel, dim3(max_tiles, n), dim3(64), 0, (hipStream_t)stream, mm);
    return ynet_check_launch("lora_compose_pack_multi");
}

}  // extern "C"
