// Goal / waypoint sampling of evaluate() on the device:
//
//   multinomial_topk_kernel   torch.multinomial(prob, K, replacement=False) of utils/image_utils.py:110-135 with a
//                             DOCUMENTED counter-based generator, so that a CPU restatement (oracle/ynet_oracle.py
//                             device_multinomial) reproduces every draw from the seed alone
//   multinomial_cdf_kernel    the replacement=True form (TTST's 10000 thresholded goal samples, utils/evaluate.py:137-139)
//   cws_prior_kernel          torch_multivariate_gaussian_heatmap x sigmoid map, normalised, and its expectation
//                             (conditioned waypoint sampling, utils/evaluate.py:9-34, 172-224)
//
// Random numbers: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; the constants of Random123 / cuRAND / torch's
// device generator), key = (seed & 0xffffffff, seed >> 32), counter = (element, 0, row, stream) with stream 0 for the
// race without replacement and 1 for the inverse-CDF draws.  A 53-bit uniform in (0, 1) is built from the first two
// output words: u = ((x0 >> 5) * 2^26 + (x1 >> 6) + 0.5) * 2^-53.
//
// Without replacement (exponential race, the algorithm ATen's multinomial uses too): element i of row r gets the
// key  p_i / E_i,  E_i = -log(u_i)  (fp64);  the K samples are the indices of the K largest keys in DESCENDING key
// order (ties: smaller index first); p_i = 0 never wins.  With `rel_threshold` entries below threshold * max(row) are
// zeroed first (utils/image_utils.py:113-118).
// With replacement: the row is cut into 256 contiguous segments; cdf = (sequential fp64 sum of the segments before) +
// (sequential fp64 sum inside the segment); sample j takes the first element whose cdf >= u_j * total.
#include "ynet_common.h"
#include <math.h>

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned& o0, unsigned& o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0;
    o1 = c1;
}

__device__ __forceinline__ double philox_uniform(unsigned elem, unsigned row, unsigned stream, unsigned k0, unsigned k1) {
    unsigned x0, x1;
    philox4x32_10(elem, 0u, row, stream, k0, k1, x0, x1);
    return ((double)(x0 >> 5) * 67108864.0 + (double)(x1 >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// max of a row (block-wide), every thread gets it
__device__ float block_row_max(const float* __restrict__ p, int n, float* red) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, p[i]);
    m = wave_max_f(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return m;
}

// ------------------------------------------------------------------------------------------------
// K samples without replacement per row; one workgroup per row.  Every thread keeps the K best keys of its
// (strided) elements in a sorted list in LDS, then K rounds of a block-wide arg-max over the list heads merge them.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multinomial_topk_kernel(const float* __restrict__ prob, long long row_stride, int n,
                                                               int K, float rel_threshold, unsigned k0, unsigned k1,
                                                               long long* __restrict__ out, int* __restrict__ status,
                                                               const unsigned long long* __restrict__ seed_ptr) {
    if (seed_ptr != nullptr) {      // the seed as a device input (a captured evaluation sweep: ynet_multinomial_devseed)
        const unsigned long long sd = *seed_ptr;
        k0 = (unsigned)(sd & 0xffffffffull);
        k1 = (unsigned)(sd >> 32);
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* lkey = reinterpret_cast<double*>(lds_raw);                 // [K][256]
    int* lidx = reinterpret_cast<int*>(lkey + (size_t)K * 256);        // [K][256]
    __shared__ float red[4];
    __shared__ double wkey[4];
    __shared__ int widx[4], wthr[4];
    const int tid = threadIdx.x, row = blockIdx.x;
    const float* p = prob + (long long)row * row_stride;
    float cut = 0.f;
    if (rel_threshold > 0.f) cut = block_row_max(p, n, red) * rel_threshold;
    for (int k = 0; k < K; ++k) {
        lkey[k * 256 + tid] = -1.0;
        lidx[k * 256 + tid] = 0x7fffffff;
    }
    double kmin = -1.0;         // this thread's K-th best so far
    for (int i = tid; i < n; i += 256) {
        float pv = p[i];
        if (rel_threshold > 0.f && pv < cut) pv = 0.f;
        if (!(pv > 0.f)) continue;
        const double u = philox_uniform((unsigned)i, (unsigned)row, 0u, k0, k1);
        const double key = (double)pv / -log(u);
        if (key > kmin) {       // insert (descending; elements arrive in increasing index order, so ties keep the smaller index first)
            int pos = K - 1;
            while (pos > 0 && lkey[(pos - 1) * 256 + tid] < key) {
                lkey[pos * 256 + tid] = lkey[(pos - 1) * 256 + tid];
                lidx[pos * 256 + tid] = lidx[(pos - 1) * 256 + tid];
                --pos;
            }
            lkey[pos * 256 + tid] = key;
            lidx[pos * 256 + tid] = i;
            kmin = lkey[(K - 1) * 256 + tid];
        }
    }
    int head = 0;
    for (int k = 0; k < K; ++k) {
        double bk = head < K ? lkey[head * 256 + tid] : -1.0;
        int bi = head < K ? lidx[head * 256 + tid] : 0x7fffffff;
        int bt = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ok = __shfl_xor(bk, o, 64);
            const int oi = __shfl_xor(bi, o, 64), ot = __shfl_xor(bt, o, 64);
            if (ok > bk || (ok == bk && oi < bi)) {
                bk = ok;
                bi = oi;
                bt = ot;
            }
        }
        if ((tid & 63) == 0) {
            wkey[tid >> 6] = bk;
            widx[tid >> 6] = bi;
            wthr[tid >> 6] = bt;
        }
        __syncthreads();
        bk = wkey[0];
        bi = widx[0];
        bt = wthr[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wkey[w] > bk || (wkey[w] == bk && widx[w] < bi)) {
                bk = wkey[w];
                bi = widx[w];
                bt = wthr[w];
            }
        __syncthreads();
        if (tid == bt) ++head;
        if (tid == 0) {
            if (!(bk > 0.0)) {          // fewer than K entries with positive probability (torch raises here)
                atomicExch(status, 1);
                bi = 0;
            }
            out[(long long)row * K + k] = bi;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K samples WITH replacement per row (inverse CDF in fp64, fixed summation order); one workgroup per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multinomial_cdf_kernel(const float* __restrict__ prob, long long row_stride, int n, int K,
                                                              float rel_threshold, unsigned k0, unsigned k1,
                                                              long long* __restrict__ out, int* __restrict__ status,
                                                              const unsigned long long* __restrict__ seed_ptr) {
    if (seed_ptr != nullptr) {
        const unsigned long long sd = *seed_ptr;
        k0 = (unsigned)(sd & 0xffffffffull);
        k1 = (unsigned)(sd >> 32);
    }
    __shared__ float red[4];
    __shared__ double seg_off[257];      // exclusive sums of the segment totals, [256] = total
    const int tid = threadIdx.x, row = blockIdx.x;
    const float* p = prob + (long long)row * row_stride;
    float cut = 0.f;
    if (rel_threshold > 0.f) cut = block_row_max(p, n, red) * rel_threshold;
    const int seg = (n + 255) / 256;
    const int lo = min(n, tid * seg), hi = min(n, lo + seg);
    double s = 0.0;
    for (int i = lo; i < hi; ++i) {
        float pv = p[i];
        if (rel_threshold > 0.f && pv < cut) pv = 0.f;
        if (pv > 0.f) s += (double)pv;
    }
    seg_off[tid + 1] = s;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        seg_off[0] = 0.0;
        for (int t = 1; t <= 256; ++t) {
            run += seg_off[t];
            seg_off[t] = run;
        }
        if (!(run > 0.0)) atomicExch(status, 1);
    }
    __syncthreads();
    const double total = seg_off[256];
    for (int j = tid; j < K; j += 256) {
        const double target = philox_uniform((unsigned)j, (unsigned)row, 1u, k0, k1) * total;
        int a = 0, b = 256;             // first segment whose inclusive sum reaches the target
        while (a < b) {
            const int m = (a + b) >> 1;
            if (seg_off[m + 1] >= target) b = m;
            else a = m + 1;
        }
        const int t = min(a, 255);
        double run = seg_off[t];
        const int l2 = min(n, t * seg), h2 = min(n, l2 + seg);
        int pick = -1, last_pos = -1;
        for (int i = l2; i < h2; ++i) {
            float pv = p[i];
            if (rel_threshold > 0.f && pv < cut) pv = 0.f;
            if (pv > 0.f) {
                run += (double)pv;
                last_pos = i;
                if (run >= target) {
                    pick = i;
                    break;
                }
            }
        }
        if (pick < 0) pick = last_pos >= 0 ? last_pos : 0;
        out[(long long)row * K + j] = pick;
    }
}

// ------------------------------------------------------------------------------------------------
// Conditioned waypoint sampling prior.  For row n (person n % B of goal sample n / B):
//   k(x, y)  = exp(-0.5 * m^T T^-1 m),  m = (lin_W(x) - mean_x, lin_H(y) - mean_y),  lin_N(i) = i * N / (N - 1)
//   T        = R diag((d/sf/ratio)^2, (d/sf)^2) R^T,  d = |dist| + 5,  R = rotation by atan2(dist_x, dist_y) [, then 90 deg]
//   map      = sig * k / sum(sig * k)                                  (out_map, optional)
//   xy       = (sum x * map, sum y * map) over pixel indices x, y     (out_xy, optional)
// Everything per pixel in fp64 (the reference's fp32 chain is within 3e-6 pixels of the exact value; T^-1 is taken
// analytically as R diag(1/a, 1/b) R^T).  One workgroup per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cws_prior_kernel(const float* __restrict__ sig, long long sig_bs, int nb,
                                                        const float* __restrict__ mean_xy, const float* __restrict__ dist_xy,
                                                        int H, int W, float sigma_factor, float ratio, int rot,
                                                        float* __restrict__ out_map, float* __restrict__ out_xy) {
    __shared__ double wsum[4][3];
    const int tid = threadIdx.x, row = blockIdx.x;
    const float* s = sig + (long long)(row % nb) * sig_bs;
    const double mx = (double)mean_xy[2 * row], my = (double)mean_xy[2 * row + 1];
    const double dx = (double)dist_xy[2 * row], dy = (double)dist_xy[2 * row + 1];
    const double rad = atan2(dx, dy);
    double c = cos(rad), sn = sin(rad);
    // R = [[c, sn], [-sn, c]];  rot: R <- [[0, -1], [1, 0]] R = [[sn, -c], [c, sn]]
    double r00 = c, r01 = sn, r10 = -sn, r11 = c;
    if (rot) {
        r00 = sn;
        r01 = -c;
        r10 = c;
        r11 = sn;
    }
    const double d = sqrt(dx * dx + dy * dy) + 5.0;
    const double sa = d / (double)sigma_factor / (double)ratio, sb = d / (double)sigma_factor;
    const double ia = 1.0 / (sa * sa), ib = 1.0 / (sb * sb);
    // T^-1 = R diag(ia, ib) R^T
    const double t00 = r00 * r00 * ia + r01 * r01 * ib;
    const double t01 = r00 * r10 * ia + r01 * r11 * ib;
    const double t11 = r10 * r10 * ia + r11 * r11 * ib;
    const double stepx = W > 1 ? (double)W / (double)(W - 1) : 0.0, stepy = H > 1 ? (double)H / (double)(H - 1) : 0.0;
    const int n = H * W;
    double S = 0.0, Sx = 0.0, Sy = 0.0;
    for (int i = tid; i < n; i += 256) {
        const int y = i / W, x = i - y * W;
        const double X = (double)x * stepx - mx, Y = (double)y * stepy - my;
        const double q = X * (t00 * X + t01 * Y) + Y * (t01 * X + t11 * Y);
        const double v = (double)s[i] * exp(-0.5 * q);
        S += v;
        Sx += v * (double)x;
        Sy += v * (double)y;
    }
    S = wave_sum_d(S);
    Sx = wave_sum_d(Sx);
    Sy = wave_sum_d(Sy);
    if ((tid & 63) == 0) {
        wsum[tid >> 6][0] = S;
        wsum[tid >> 6][1] = Sx;
        wsum[tid >> 6][2] = Sy;
    }
    __syncthreads();
    S = (wsum[0][0] + wsum[1][0]) + (wsum[2][0] + wsum[3][0]);
    Sx = (wsum[0][1] + wsum[1][1]) + (wsum[2][1] + wsum[3][1]);
    Sy = (wsum[0][2] + wsum[1][2]) + (wsum[2][2] + wsum[3][2]);
    if (out_xy != nullptr && tid == 0) {
        out_xy[2 * row] = (float)(Sx / S);
        out_xy[2 * row + 1] = (float)(Sy / S);
    }
    if (out_map != nullptr) {
        float* o = out_map + (long long)row * n;
        const double inv = 1.0 / S;
        for (int i = tid; i < n; i += 256) {
            const int y = i / W, x = i - y * W;
            const double X = (double)x * stepx - mx, Y = (double)y * stepy - my;
            const double q = X * (t00 * X + t01 * Y) + Y * (t01 * X + t11 * Y);
            o[i] = (float)((double)s[i] * exp(-0.5 * q) * inv);
        }
    }
}

static int multinomial_impl(const float* prob, long long rows, long long row_stride, int n, int K, int replacement,
                            float rel_threshold, unsigned long long seed, const unsigned long long* seed_ptr, long long* out, int* status,
                            void* stream) {
    YNET_REQUIRE(prob && out && status, "multinomial: null pointer");
    YNET_REQUIRE(rows > 0 && rows < (1ll << 31) && n > 0 && K > 0, "multinomial: bad shape rows=%lld n=%d K=%d", rows, n, K);
    YNET_REQUIRE(rel_threshold >= 0.f && rel_threshold <= 1.f, "multinomial: rel_threshold must lie in [0, 1]");
    const unsigned k0 = (unsigned)(seed & 0xffffffffull), k1 = (unsigned)(seed >> 32);
    if (replacement) {
        hipLaunchKernelGGL(multinomial_cdf_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, prob, row_stride, n, K,
                           rel_threshold, k0, k1, out, status, seed_ptr);
        return ynet_check_launch("multinomial(replacement)");
    }
    YNET_REQUIRE(K <= 48 && K <= n, "multinomial: without replacement K <= min(48, n) is supported (got K=%d, n=%d)", K, n);
    const int lds = K * 256 * 12;
    static bool attr_dev[YNET_MAX_DEV] = {false};
    bool& attr_set = attr_dev[ynet_device_slot()];
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(multinomial_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  48 * 256 * 12);
        attr_set = true;
    }
    hipLaunchKernelGGL(multinomial_topk_kernel, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, prob, row_stride, n, K,
                       rel_threshold, k0, k1, out, status, seed_ptr);
    return ynet_check_launch("multinomial");
}

extern "C" {

int ynet_multinomial(const float* prob, long long rows, long long row_stride, int n, int K, int replacement,
                     float rel_threshold, unsigned long long seed, long long* out, int* status, void* stream) {
    return multinomial_impl(prob, rows, row_stride, n, K, replacement, rel_threshold, seed, nullptr, out, status, stream);
}

// ynet_multinomial with the seed read from device memory by the kernel (8 bytes, 8-byte aligned): the launch has no per-call
// argument and can be recorded into a hipGraph (utils/evaluate.py: the captured sweep copies the seeds it draws into a static buffer)
int ynet_multinomial_devseed(const float* prob, long long rows, long long row_stride, int n, int K, int replacement,
                             float rel_threshold, const unsigned long long* seed_dev, long long* out, int* status, void* stream) {
    YNET_REQUIRE(seed_dev != nullptr && (reinterpret_cast<uintptr_t>(seed_dev) & 7) == 0, "multinomial_devseed: null / unaligned seed pointer");
    return multinomial_impl(prob, rows, row_stride, n, K, replacement, rel_threshold, 0ull, seed_dev, out, status, stream);
}

int ynet_cws_prior(const float* sig, long long sig_batch_stride, int n_persons, const float* mean_xy, const float* dist_xy,
                   int rows, int H, int W, float sigma_factor, float ratio, int rot, float* out_map, float* out_xy,
                   void* stream) {
    YNET_REQUIRE(sig && mean_xy && dist_xy && (out_map || out_xy), "cws_prior: null pointer");
    YNET_REQUIRE(rows > 0 && n_persons > 0 && H > 0 && W > 0, "cws_prior: bad shape rows=%d persons=%d %dx%d", rows, n_persons, H, W);
    YNET_REQUIRE(sigma_factor != 0.f && ratio != 0.f, "cws_prior: sigma_factor and ratio must be non-zero");
    hipLaunchKernelGGL(cws_prior_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, sig, sig_batch_stride, n_persons,
                       mean_xy, dist_xy, H, W, sigma_factor, ratio, rot, out_map, out_xy);
    return ynet_check_launch("cws_prior");
}

}  // extern "C"
