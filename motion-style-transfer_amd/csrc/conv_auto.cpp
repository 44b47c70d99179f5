// ynet_conv2d_auto: ONE dispatching convolution entry (round 6; VERDICT r5 item 5, SURVEY 8(b)'s
// `ynet_conv2d_fwd(..., x2, Cin2, pre_op)` / `ynet_conv2d_dgrad(dy, w, y_for_relu_mask, dx, ...)`).
//
// nn.Conv2d(3x3 / 1x1 / 5x5, padding K/2) [+ ReLU] of models/ynet.py:150,192-211,420-451,464,467 and its data gradient, with the
// concatenation of the inputs (ynet.py:387,466,574), the 2x2 max-pool behind it (ynet.py:202,215), the bilinear x2 in front of it
// (ynet.py:463), the ReLU backward of the layer below and the shared-skip-term of utils/evaluate.py:248-283 as optional operands.
// The caller describes the operation; this file chooses among the kernel families of the library -- implicit GEMM (conv_mfma.hip),
// Winograd F(2x2, 3x3) with all filters resident (conv_wino_kernel), its concatenated-source form, its slice form and the
// up-convolution forms (conv_wino.hip) --, splits wide layers into the launches those kernels serve and keeps the TRANSFORMED
// filters in a caller-owned cache, re-transforming when the caller's filter version (or the plan) changes.  Until round 5 this
// composition lived in Python (ops.conv2d_raw): a maintainer binding include/ynet_hip.h alone reached the implicit GEMM only.
// Host code only: every launch goes through the library's own entry points.
#include "ynet_common.h"
#include "../../include/ynet_hip.h"
#include <stdint.h>
#include <string.h>

namespace {

struct Piece {
    float* ptr;
    int n;
    long long bs;
    int col0;
    int dst;             // index of the destination this piece belongs to
    int whole;           // the piece IS its destination (all its channels)
};

enum Kind { K_NONE = 0, K_WINO = 1, K_CAT = 2, K_W16 = 3 };

struct Launch {          // one Winograd launch of a plan: which filter it needs
    int kind;            // K_WINO: ynet_winograd_filter; K_CAT: ynet_winograd_filter_cat; K_W16: ynet_winograd16_filter
    int row0;            // first input-channel row of the packed filter
    int cs[4], ncs;      // channels per source (K_WINO: cs[0] = cin)
    int cout, col0, ctot;
    long long floats;    // size of the transformed filter
    long long offset;    // into the cache
};

struct Plan {
    int family;          // YNET_AUTO_*: 0 implicit GEMM, 1 winograd, 2 winograd_cat, 3 winograd16, 4 upsample2x + winograd, 5 upsample2x + winograd16
    int variant;         // which branch of the dispatcher (see run())
    Launch l[4];
    int nl;
    Piece pieces[4];
    int npieces;
    long long cache_floats;
};

inline bool al(const void* p, uintptr_t a) { return ((uintptr_t)p & (a - 1)) == 0; }

long long filter_floats(const Launch& q) {
    switch (q.kind) {
        case K_WINO: return ynet_winograd_filter_floats(q.cs[0], q.cout);
        case K_CAT: return ynet_winograd_filter_cat_floats(q.cs, q.ncs, q.cout);
        case K_W16: return ynet_winograd16_filter_floats(q.cs, q.ncs, q.cout);
    }
    return 0;
}

void add_launch(Plan& p, int kind, int row0, const int* cs, int ncs, int cout, int col0, int ctot) {
    Launch& q = p.l[p.nl++];
    q.kind = kind;
    q.row0 = row0;
    q.ncs = ncs;
    for (int i = 0; i < 4; ++i) q.cs[i] = i < ncs ? cs[i] : 0;
    q.cout = cout;
    q.col0 = col0;
    q.ctot = ctot;
    q.floats = (filter_floats(q) + 3) & ~3ll;
    q.offset = p.cache_floats;
    p.cache_floats += q.floats;
}

bool w16_ok(const YnetConvAuto* a, const int* cs, int ncs, int cout) {
    return !(a->flags & (YNET_AUTO_NO_WINOGRAD | YNET_AUTO_NO_WINOGRAD16)) && a->K == 3 && ncs >= 1 && ncs <= 3 &&
           ynet_conv2d_winograd16_supported(a->B, a->H, a->W, cs, ncs, cout, a->K);
}

bool srcs_plain(const YnetConvAuto* a) {      // no batch modulus (the Winograd kernels address image b)
    for (int i = 0; i < a->nsrc; ++i)
        if (a->src_bmod[i] > 0) return false;
    return true;
}

bool srcs_aligned(const YnetConvAuto* a) {
    for (int i = 0; i < a->nsrc; ++i)
        if (!al(a->src[i], 16) || (a->src_bs[i] & 3)) return false;
    return true;
}

// ---- the plan: a pure function of the descriptor (shapes, alignments, operands present, flags) -------------------------------------
int make_plan(const YnetConvAuto* a, Plan& p) {
    memset(&p, 0, sizeof(p));
    const bool wino = !(a->flags & YNET_AUTO_NO_WINOGRAD) && a->K == 3 && srcs_plain(a);
    const int B = a->B, H = a->H, W = a->W, K = a->K;
    int cs[4];
    for (int i = 0; i < 4; ++i) cs[i] = i < a->nsrc ? a->src_c[i] : 0;
    int ctot = 0;
    for (int i = 0; i < a->ndst; ++i) ctot += a->dst_c[i];

    if (a->upsample2x) {      // pre_op: bilinear x2 inside the convolution (models/ynet.py:463-464)
        YNET_REQUIRE(a->nsrc == 1 && a->ndst == 1 && !a->mask && !a->relu_of && !a->pooled && !a->addend && !a->bits_out && !a->relu_bits && !a->wbits_out && !a->relu_wbits,
                     "conv2d_auto: upsample2x takes one source, one destination and no other epilogue operand");
        const int s = wino && al(a->src[0], 16) ? ynet_upsample2x_conv2d_winograd_supported(B, H, W, cs[0], ctot, K) : 0;
        YNET_REQUIRE(s == 1 || s == 2, "conv2d_auto: upsample2x + conv is not served for cin %d cout %d at %dx%d (B %d): run ynet_upsample2x_fwd, then the convolution", cs[0], ctot, H, W, B);
        p.family = s == 1 ? 4 : 5;
        add_launch(p, s == 1 ? K_WINO : K_W16, 0, cs, 1, ctot, 0, ctot);
        return 0;
    }
    if (a->bits_out || a->relu_bits) {      // the direct tiles' 1-bit masks: implicit GEMM only
        p.family = 0;
        p.variant = a->bits_out ? 1 : 2;
        return 0;
    }
    if (a->addend) {      // y = [relu](conv(cat(src)) + bias + addend[b % mod]) (utils/evaluate.py:248-283)
        YNET_REQUIRE(a->ndst == 1 && !a->mask && !a->relu_of && !a->pooled, "conv2d_auto: addend takes one destination and no mask / relu_of / pooled");
        const bool ok = !(a->flags & YNET_AUTO_NO_WINOGRAD) && K == 3 && srcs_aligned(a) && al(a->addend, 8) && al(a->dst[0], 8) && !(a->dst_bs[0] & 1) && a->nsrc <= 3;
        bool plain = true;      // (the Winograd kernels take no per-source batch modulus)
        for (int i = 0; i < a->nsrc; ++i) plain = plain && a->src_bmod[i] <= 0;
        if (ok && plain && ctot == 32 && ynet_conv2d_winograd_cat_supported(B, H, W, cs, a->nsrc, ctot, K)) {
            p.family = 2;
            p.variant = 30;
            add_launch(p, K_CAT, 0, cs, a->nsrc, ctot, 0, ctot);
        } else if (ok && plain && w16_ok(a, cs, a->nsrc, ctot)) {
            p.family = 3;
            p.variant = 31;
            add_launch(p, K_W16, 0, cs, a->nsrc, ctot, 0, ctot);
        } else {
            YNET_REQUIRE(ynet_conv2d_add_supported(B, H, W, ctot, K), "conv2d_auto: a convolution with an additive term is not served at %dx%d (B %d, cout %d, K %d)", H, W, B, ctot, K);
            p.family = 0;
            p.variant = 32;
        }
        return 0;
    }
    if (a->pooled) {      // conv [+ ReLU] + MaxPool2d(2, 2) (models/ynet.py:202,215)
        YNET_REQUIRE(a->ndst == 1 && !a->mask && !a->relu_of, "conv2d_auto: pooled is for a forward convolution with one destination");
        const bool okp = wino && srcs_aligned(a) && al(a->dst[0], 8) && !(a->dst_bs[0] & 1) && a->nsrc <= 3;
        if (okp && ctot == 32 && ynet_conv2d_winograd_cat_supported(B, H, W, cs, a->nsrc, 32, K)) {
            p.family = 2;
            p.variant = 10;
            add_launch(p, K_CAT, 0, cs, a->nsrc, 32, 0, 32);
        } else if (okp && w16_ok(a, cs, a->nsrc, ctot)) {
            p.family = 3;
            p.variant = 11;
            add_launch(p, K_W16, 0, cs, a->nsrc, ctot, 0, ctot);
        } else {
            p.family = 0;
            p.variant = 12;
        }
        return 0;
    }
    // ---- one source of 16 / 32 / 64 channels: the plain convolutions and the data gradients
    if (wino && !a->mask && a->nsrc == 1 && al(a->src[0], 16) && !(a->src_bs[0] & 3) && (!a->relu_of || (al(a->relu_of, 8) && !(a->relu_of_bs & 1)))) {
        const int cin = cs[0];
        const long long HW = (long long)H * W;
        // destination channels in pieces of 32 / 16 (a 48- or 64-channel data gradient is several launches over slices of the filter;
        // the input is re-read from L2; pieces nobody wants are not computed)
        Piece pieces[8];
        int np = 0, c0 = 0;
        bool pieces_ok = true;
        for (int i = 0; i < a->ndst && pieces_ok; ++i) {
            if (!a->dst[i]) { c0 += a->dst_c[i]; continue; }
            int o = 0;
            const int c = a->dst_c[i];
            while (c - o >= 16 && (c - o) % 16 == 0 && np < 8) {
                const int n = c - o >= 32 ? 32 : 16;
                pieces[np++] = Piece{a->dst[i] + o * HW, n, a->dst_bs[i], c0 + o, i, n == c ? 1 : 0};
                o += n;
            }
            if (o != c) pieces_ok = false;
            c0 += c;
        }
        if (!pieces_ok) np = 0;
        auto piece_al = [&](int cnt) {
            for (int i = 0; i < cnt; ++i)
                if (!al(pieces[i].ptr, 8) || (pieces[i].bs & 1)) return false;
            return true;
        };
        if ((a->flags & YNET_AUTO_WINOGRAD16_FOR_16) && np == 1 && pieces[0].n == 16 && piece_al(1) && w16_ok(a, cs, 1, 16)) {
            p.family = 3;
            p.variant = 20;
            p.pieces[0] = pieces[0];
            p.npieces = 1;
            add_launch(p, K_W16, 0, cs, 1, 16, pieces[0].col0, ctot);
            return 0;
        }
        bool wide = false;
        for (int i = 0; i < a->ndst; ++i) wide = wide || (a->dst[i] && a->dst_c[i] >= 64);
        bool all_sup = np > 0 && np <= (a->relu_of ? 1 : 2) && piece_al(np);
        for (int i = 0; i < np && all_sup; ++i) all_sup = ynet_conv2d_winograd_supported(B, H, W, cin, pieces[i].n, K) != 0;
        // [16 channels, 32 channels] of a plain data gradient (the decoders' first convolution at the last level: cat(up-sampled 16, skip 32[, way-point map])):
        // ONE launch with three output blocks per wave -- dy is read and transformed once (ynet_conv2d_winograd_split)
        if (all_sup && !wide && np == 2 && !a->relu_of && !a->bias && !a->relu && !(a->flags & YNET_AUTO_NO_SPLIT48) && pieces[0].n == 16 && pieces[0].whole &&
            pieces[1].n == 32 && pieces[1].whole && pieces[1].col0 == pieces[0].col0 + 16 && ynet_conv2d_winograd_split_supported(B, H, W, cin)) {
            p.family = 1;
            p.variant = 23;
            p.npieces = 2;
            p.pieces[0] = pieces[0];
            p.pieces[1] = pieces[1];
            add_launch(p, K_WINO, 0, cs, 1, 48, pieces[0].col0, ctot);
            return 0;
        }
        if (all_sup && !wide) {
            p.family = 1;
            p.variant = 21;
            p.npieces = np;
            for (int i = 0; i < np; ++i) {
                p.pieces[i] = pieces[i];
                add_launch(p, K_WINO, 0, cs, 1, pieces[i].n, pieces[i].col0, ctot);
            }
            return 0;
        }
        // the slice form: 64 input channels, destinations of 48 / 64 channels (one launch per wanted destination over its filter slice)
        Piece wanted[4];
        int nw = 0;
        c0 = 0;
        for (int i = 0; i < a->ndst; ++i) {
            if (a->dst[i]) wanted[nw++] = Piece{a->dst[i], a->dst_c[i], a->dst_bs[i], c0, i, 1};
            c0 += a->dst_c[i];
        }
        bool w_ok = nw > 0 && (!a->relu_of || nw == 1);
        for (int i = 0; i < nw && w_ok; ++i) w_ok = al(wanted[i].ptr, 8) && !(wanted[i].bs & 1) && w16_ok(a, cs, 1, wanted[i].n);
        if (w_ok) {
            p.family = 3;
            p.variant = 22;
            p.npieces = nw;
            for (int i = 0; i < nw; ++i) {
                p.pieces[i] = wanted[i];
                add_launch(p, K_W16, 0, cs, 1, wanted[i].n, wanted[i].col0, ctot);
            }
            return 0;
        }
        if (all_sup && wide) {      // (YNET_AUTO_NO_WINOGRAD16: a 64-channel destination as two 32-channel launches)
            p.family = 1;
            p.variant = 21;
            p.npieces = np;
            for (int i = 0; i < np; ++i) {
                p.pieces[i] = pieces[i];
                add_launch(p, K_WINO, 0, cs, 1, pieces[i].n, pieces[i].col0, ctot);
            }
            return 0;
        }
    }
    // ---- several sources (or an odd channel count): the decoders' first convolutions
    if (wino && !a->mask && !a->relu_of && (a->nsrc > 1 || (cs[0] != 16 && cs[0] != 32)) && srcs_aligned(a) && a->nsrc <= 3) {
        int nwant = 0;
        for (int i = 0; i < a->ndst; ++i) nwant += a->dst[i] ? 1 : 0;
        if (nwant == 1 && a->ndst == 1 && al(a->dst[0], 8) && !(a->dst_bs[0] & 1)) {
            if (ctot == 32) {
                int rest[4], nr = 0;
                bool has_rest = cs[0] >= 32;
                if (has_rest) {
                    if (cs[0] - 32 > 0) rest[nr++] = cs[0] - 32;
                    for (int i = 1; i < a->nsrc; ++i)
                        if (cs[i] > 0) rest[nr++] = cs[i];
                }
                const bool cat_ok = ynet_conv2d_winograd_cat_supported(B, H, W, cs, a->nsrc, 32, K) != 0;
                if (!cat_ok && has_rest && nr > 0 && nr <= 3 && ynet_conv2d_winograd_supported(B, H, W, 32, 32, K) &&
                    ynet_conv2d_winograd_cat_supported(B, H, W, rest, nr, 32, K)) {
                    // 57 .. 88 input channels (64 / 65 -> 32 at 128^2): the first 32 channels into the destination, then the rest with
                    // the destination as the additive term in front of bias and ReLU (read and written by the same lane: in place)
                    p.family = 2;
                    p.variant = 40;
                    const int c32 = 32;
                    add_launch(p, K_WINO, 0, &c32, 1, 32, 0, 32);
                    add_launch(p, K_CAT, 32, rest, nr, 32, 0, 32);
                    return 0;
                }
                if (cat_ok) {
                    p.family = 2;
                    p.variant = 41;
                    add_launch(p, K_CAT, 0, cs, a->nsrc, 32, 0, 32);
                    return 0;
                }
            }
            if (w16_ok(a, cs, a->nsrc, ctot)) {
                p.family = 3;
                p.variant = 42;
                add_launch(p, K_W16, 0, cs, a->nsrc, ctot, 0, ctot);
                return 0;
            }
            for (int cut = 1; cut < a->nsrc; ++cut)      // more than 84 padded channels: leading sources, then the rest added in place
                if (w16_ok(a, cs, cut, ctot) && w16_ok(a, cs + cut, a->nsrc - cut, ctot)) {
                    int r0 = 0;
                    for (int i = 0; i < cut; ++i) r0 += cs[i];
                    p.family = 3;
                    p.variant = 43;
                    add_launch(p, K_W16, 0, cs, cut, ctot, 0, ctot);
                    add_launch(p, K_W16, r0, cs + cut, a->nsrc - cut, ctot, 0, ctot);
                    p.l[1].row0 = r0;
                    p.pieces[0].col0 = cut;      // (the cut)
                    return 0;
                }
        }
    }
    p.family = 0;
    p.variant = a->relu_of ? 3 : 0;
    return 0;
}

unsigned long long signature(const Plan& p) {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](long long v) {
        h ^= (unsigned long long)v;
        h *= 1099511628211ull;
    };
    mix(p.family);
    mix(p.variant);
    mix(p.nl);
    for (int i = 0; i < p.nl; ++i) {
        const Launch& q = p.l[i];
        mix(q.kind); mix(q.row0); mix(q.ncs);
        for (int j = 0; j < q.ncs; ++j) mix(q.cs[j]);
        mix(q.cout); mix(q.col0); mix(q.ctot); mix(q.offset);
    }
    return h ? h : 1;
}

int transform(const YnetConvAuto* a, const Plan& p, void* stream) {
    for (int i = 0; i < p.nl; ++i) {
        const Launch& q = p.l[i];
        const long long cols_pad = ((long long)q.ctot + 63) / 64 * 64;
        const float* wp = a->wp + (long long)q.row0 * a->K * a->K * cols_pad;
        float* u = a->cache + q.offset;
        int rc = 0;
        if (q.kind == K_WINO) rc = ynet_winograd_filter(wp, u, q.cs[0], q.cout, q.col0, q.ctot, stream);
        else if (q.kind == K_CAT) rc = ynet_winograd_filter_cat(wp, u, q.cs, q.ncs, q.cout, q.col0, q.ctot, stream);
        else rc = ynet_winograd16_filter(wp, u, q.cs, q.ncs, q.cout, q.col0, q.ctot, stream);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

long long ynet_conv2d_auto_cache_floats(const YnetConvAuto* a) {
    Plan p;
    if (!a || make_plan(a, p)) return -1;
    return p.cache_floats;
}

long long ynet_conv2d_auto_workspace_floats(const YnetConvAuto* a) {
    if (!a) return -1;
    if ((long long)a->B * a->H * a->W > 65536) return 0;      // small maps only (see ynet_conv2d_workspace_floats)
    int ctot = 0;
    for (int i = 0; i < a->ndst; ++i) ctot += a->dst_c[i];
    return ynet_conv2d_workspace_floats(a->B, a->H, a->W, ctot);
}

int ynet_conv2d_auto_plan(const YnetConvAuto* a, YnetConvTaken* taken) {
    YNET_REQUIRE(a != nullptr && taken != nullptr, "conv2d_auto_plan: null pointer");
    YNET_REQUIRE(a->nsrc >= 1 && a->nsrc <= 4 && a->ndst >= 1 && a->ndst <= 4, "conv2d_auto_plan: 1..4 sources and destinations are required");
    Plan p;
    if (int rc = make_plan(a, p)) return rc;
    memset(taken, 0, sizeof(*taken));
    taken->family = p.family;
    taken->variant = p.variant;
    taken->nlaunch = p.family ? (p.variant == 23 ? 1 : (p.npieces > 0 ? p.npieces : (p.nl > 0 ? p.nl : 1))) : 1;
    return 0;
}

int ynet_conv2d_auto(const YnetConvAuto* a, YnetConvTaken* taken, void* stream) {
    YNET_REQUIRE(a != nullptr, "conv2d_auto: null descriptor");
    YNET_REQUIRE(a->nsrc >= 1 && a->nsrc <= 4 && a->ndst >= 1 && a->ndst <= 4 && a->wp, "conv2d_auto: 1..4 sources, 1..4 destinations and a packed filter are required");
    YNET_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && (a->K == 1 || a->K == 3 || a->K == 5), "conv2d_auto: bad shape B %d %dx%d K %d", a->B, a->H, a->W, a->K);
    Plan p;
    if (int rc = make_plan(a, p)) return rc;
    YnetConvTaken t;
    memset(&t, 0, sizeof(t));
    t.family = p.family;
    t.variant = p.variant;
    // ---- the transformed filters: made on `stream` when the caller's tag does not match this plan / filter version
    if (p.nl) {
        YNET_REQUIRE(a->cache && a->cache_tag && a->cache_floats >= p.cache_floats && al(a->cache, 16),
                     "conv2d_auto: this launch keeps %lld floats of transformed filters (ynet_conv2d_auto_cache_floats): pass a 16-byte aligned cache of that size and its tag", p.cache_floats);
        const unsigned long long sig = signature(p);
        if (a->cache_tag[0] != sig || a->cache_tag[1] != a->wp_version) {
            if (int rc = transform(a, p, stream)) return rc;
            a->cache_tag[0] = sig;
            a->cache_tag[1] = a->wp_version;
            t.transformed = 1;
        }
    }
    const int B = a->B, H = a->H, W = a->W, K = a->K, relu = a->relu ? 1 : 0;
    int ctot = 0;
    for (int i = 0; i < a->ndst; ++i) ctot += a->dst_c[i];
    const float* u0 = p.nl ? a->cache + p.l[0].offset : nullptr;
    const float* u1 = p.nl > 1 ? a->cache + p.l[1].offset : nullptr;
    auto done = [&](int rc) {
        if (!rc && taken) *taken = t;
        return rc;
    };
    auto tag = [&](int i, int x, int y, int z) {
        if (i < 4) { t.tmpl[i][0] = x; t.tmpl[i][1] = y; t.tmpl[i][2] = z; }
        t.nlaunch = i + 1;
    };
    float* ws = a->workspace_floats > 0 ? a->workspace : nullptr;
    const long long nws = ws ? a->workspace_floats : 0;

    switch (p.family) {
    case 4:
    case 5:
        tag(0, p.family == 4 ? 4 : 0, 0, 0);
        return done(ynet_upsample2x_conv2d_winograd(a->src[0], a->src_bs[0], u0, a->bias, a->dst[0], a->dst_bs[0], a->src_c[0], ctot, B, H, W, relu, stream));
    case 1: {      // conv_wino_kernel<NCB, NCH, EM>, one launch per destination piece
        if (p.variant == 23) {      // two destinations, one launch (conv_wino_kernel<3, 4, 5 | 6, 4>)
            const Piece &q0 = p.pieces[0], &q1 = p.pieces[1];
            const int s2d = a->dst_s2d[q0.dst] ? 1 : 0;
            const int rc = ynet_conv2d_winograd_split(a->src[0], a->src_bs[0], u0, q0.ptr, q0.bs, s2d, q1.ptr, q1.bs, a->src_c[0], B, H, W, stream);
            if (rc) return rc;
            if (s2d) t.wrote_s2d |= 1 << q0.dst;
            tag(0, 3, a->src_c[0] / 8, s2d ? 6 : 5);
            return done(0);
        }
        const bool one32 = p.npieces == 1 && p.pieces[0].n == 32;
        unsigned* wb_out = (one32 && relu && !a->relu_of && !(a->flags & YNET_AUTO_NO_RELU_WBITS)) ? a->wbits_out : nullptr;
        const unsigned* wb_in = (one32 && a->relu_of && !relu && !a->bias && !(a->flags & YNET_AUTO_NO_RELU_WBITS)) ? a->relu_wbits : nullptr;
        const int em = wb_out ? 3 : (wb_in ? 2 : (a->relu_of ? 1 : 0));
        for (int i = 0; i < p.npieces; ++i) {
            const Piece& q = p.pieces[i];
            const float* u = a->cache + p.l[i].offset;
            const float* bias = a->bias ? a->bias + q.col0 : nullptr;
            int rc;
            // (a destination that asked for its gradient space-to-depth gets it when ONE plain 16 / 32-channel launch writes all of it)
            const bool s2d = a->dst_s2d[q.dst] && q.whole && em == 0 && !a->bias && !relu;
            if (s2d) {
                rc = ynet_conv2d_winograd_s2d(a->src[0], a->src_bs[0], u, q.ptr, q.bs, a->src_c[0], q.n, B, H, W, stream);
                if (rc) return rc;
                t.wrote_s2d |= 1 << q.dst;
                tag(i, q.n / 16, a->src_c[0] / 8, 4);
                continue;
            }
            if (wb_out) rc = ynet_conv2d_winograd_relu_bits(a->src[0], a->src_bs[0], u, bias, q.ptr, q.bs, a->src_c[0], B, H, W, wb_out, stream);
            else if (wb_in) rc = ynet_conv2d_winograd_dgrad_relu_bits(a->src[0], a->src_bs[0], u, q.ptr, q.bs, wb_in, a->src_c[0], B, H, W, stream);
            else if (a->relu_of) {
                YNET_REQUIRE(!a->bias && !relu, "conv2d_auto: relu_of is for a data gradient (no bias, no ReLU)");
                rc = ynet_conv2d_winograd_dgrad_relu(a->src[0], a->src_bs[0], u, q.ptr, q.bs, a->relu_of, a->relu_of_bs, a->src_c[0], q.n, B, H, W, stream);
            } else rc = ynet_conv2d_winograd(a->src[0], a->src_bs[0], u, bias, q.ptr, q.bs, a->src_c[0], q.n, B, H, W, relu, stream);
            if (rc) return rc;
            tag(i, q.n / 16, a->src_c[0] / 8, em);
        }
        t.wrote_wbits = wb_out ? 1 : 0;
        return done(0);
    }
    case 2: {      // conv_wino_cat_kernel<2, EPI>
        if (p.variant == 30) {
            unsigned* wb = (relu && !(a->flags & YNET_AUTO_NO_RELU_WBITS)) ? a->wbits_out : nullptr;
            tag(0, 2, wb ? 5 : 2, 0);
            t.wrote_wbits = wb ? 1 : 0;
            if (wb) return done(ynet_conv2d_winograd_cat_relu_bits(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], B, H, W, a->addend, a->addend_bs, a->addend_bmod, wb, stream));
            return done(ynet_conv2d_winograd_cat_add(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], 32, B, H, W, relu, a->addend, a->addend_bs, a->addend_bmod, stream));
        }
        if (p.variant == 10) {
            unsigned char* code = (relu && !(a->flags & YNET_AUTO_NO_POOL_CODE)) ? a->pool_code : nullptr;
            tag(0, 2, code ? 6 : 3, 0);
            t.wrote_pool_code = code ? 1 : 0;
            if (code) return done(ynet_conv2d_winograd_cat_pool_code(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], a->pooled, a->pooled_bs, code, B, H, W, stream));
            return done(ynet_conv2d_winograd_cat_pool(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], a->pooled, a->pooled_bs, 32, B, H, W, relu, stream));
        }
        unsigned* wb = (relu && !(a->flags & YNET_AUTO_NO_RELU_WBITS)) ? a->wbits_out : nullptr;
        t.wrote_wbits = wb ? 1 : 0;
        if (p.variant == 40) {
            const long long HW = (long long)H * W;
            int rc = ynet_conv2d_winograd(a->src[0], a->src_bs[0], u0, nullptr, a->dst[0], a->dst_bs[0], 32, 32, B, H, W, 0, stream);
            if (rc) return rc;
            const float* rs[4];
            int rc_[4], nr = 0;
            long long rb[4];
            if (a->src_c[0] > 32) { rs[nr] = a->src[0] + 32 * HW; rc_[nr] = a->src_c[0] - 32; rb[nr] = a->src_bs[0]; ++nr; }
            for (int i = 1; i < a->nsrc; ++i) { rs[nr] = a->src[i]; rc_[nr] = a->src_c[i]; rb[nr] = a->src_bs[i]; ++nr; }
            tag(0, 2, 4, 0);
            tag(1, 2, wb ? 5 : 2, 0);
            if (wb) return done(ynet_conv2d_winograd_cat_relu_bits(rs, rc_, rb, nr, u1, a->bias, a->dst[0], a->dst_bs[0], B, H, W, a->dst[0], a->dst_bs[0], 0, wb, stream));
            return done(ynet_conv2d_winograd_cat_add(rs, rc_, rb, nr, u1, a->bias, a->dst[0], a->dst_bs[0], 32, B, H, W, relu, a->dst[0], a->dst_bs[0], 0, stream));
        }
        tag(0, 2, wb ? 4 : 0, 0);
        if (wb) return done(ynet_conv2d_winograd_cat_relu_bits(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], B, H, W, nullptr, 0, 0, wb, stream));
        return done(ynet_conv2d_winograd_cat(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], 32, B, H, W, relu, stream));
    }
    case 3: {      // conv_wino16_kernel<EPI>
        if (p.variant == 31) {
            tag(0, 2, 0, 0);
            return done(ynet_conv2d_winograd16(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], ctot, B, H, W, relu, nullptr, 0, a->addend, a->addend_bs,
                                               a->addend_bmod, nullptr, 0, stream));
        }
        if (p.variant == 11) {
            tag(0, 3, 0, 0);
            return done(ynet_conv2d_winograd16(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], ctot, B, H, W, relu, nullptr, 0, nullptr, 0, 0, a->pooled,
                                               a->pooled_bs, stream));
        }
        if (p.variant == 20 || p.variant == 22) {
            for (int i = 0; i < p.npieces; ++i) {
                const Piece& q = p.pieces[i];
                if (a->relu_of) YNET_REQUIRE(!a->bias && !relu, "conv2d_auto: relu_of is for a data gradient (no bias, no ReLU)");
                int rc = ynet_conv2d_winograd16(a->src, a->src_c, a->src_bs, 1, a->cache + p.l[i].offset, a->bias ? a->bias + q.col0 : nullptr, q.ptr, q.bs, q.n, B, H, W, relu, a->relu_of,
                                                a->relu_of_bs, nullptr, 0, 0, nullptr, 0, stream);
                if (rc) return rc;
                tag(i, a->relu_of ? 1 : 0, 0, 0);
            }
            return done(0);
        }
        if (p.variant == 43) {
            const int cut = p.pieces[0].col0;
            int rc = ynet_conv2d_winograd16(a->src, a->src_c, a->src_bs, cut, u0, nullptr, a->dst[0], a->dst_bs[0], ctot, B, H, W, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, stream);
            if (rc) return rc;
            tag(0, 0, 0, 0);
            tag(1, 2, 0, 0);
            return done(ynet_conv2d_winograd16(a->src + cut, a->src_c + cut, a->src_bs + cut, a->nsrc - cut, u1, a->bias, a->dst[0], a->dst_bs[0], ctot, B, H, W, relu, nullptr, 0, a->dst[0],
                                               a->dst_bs[0], 0, nullptr, 0, stream));
        }
        tag(0, 0, 0, 0);
        return done(ynet_conv2d_winograd16(a->src, a->src_c, a->src_bs, a->nsrc, u0, a->bias, a->dst[0], a->dst_bs[0], ctot, B, H, W, relu, nullptr, 0, nullptr, 0, 0, nullptr, 0, stream));
    }
    default: break;
    }
    // ---- implicit GEMM (conv_mfma.hip)
    t.nlaunch = 1;
    const int* bmod = nullptr;
    for (int i = 0; i < a->nsrc; ++i)
        if (a->src_bmod[i] > 0) bmod = a->src_bmod;
    switch (p.variant) {
    case 1:
        YNET_REQUIRE(a->ndst == 1 && !a->mask && !a->relu_of && !a->pooled && relu && !bmod, "conv2d_auto: bits_out is for a forward ReLU convolution with one destination");
        return done(ynet_conv2d_relu_bits(a->src, a->src_c, a->src_bs, a->nsrc, a->wp, a->bias, a->dst[0], a->dst_c[0], a->dst_bs[0], a->bits_out, B, H, W, K, stream));
    case 2:
        YNET_REQUIRE(a->nsrc == 1 && a->ndst == 1 && !a->bias && !relu && !a->relu_of, "conv2d_auto: relu_bits is for a data gradient with one source and one destination");
        return done(ynet_conv2d_dgrad_relu_bits(a->src[0], a->src_c[0], a->src_bs[0], a->mask, a->mask_bs, a->wp, a->dst[0], a->dst_c[0], a->dst_bs[0], a->relu_bits, B, H, W, K, stream));
    case 12:
        YNET_REQUIRE(!bmod, "conv2d_auto: pooled takes no batch modulus");
        return done(ynet_conv2d_pool(a->src, a->src_c, a->src_bs, a->nsrc, a->wp, a->bias, a->dst[0], a->dst_c[0], a->dst_bs[0], a->pooled, a->pooled_bs, B, H, W, K, relu, stream));
    case 32:
        return done(ynet_conv2d_add(a->src, a->src_c, a->src_bs, bmod, a->nsrc, a->wp, a->bias, a->dst[0], ctot, a->dst_bs[0], B, H, W, K, relu, a->addend, a->addend_bs, a->addend_bmod, stream));
    case 3:
        YNET_REQUIRE(a->nsrc == 1 && a->ndst == 1 && !a->bias && !relu, "conv2d_auto: relu_of is for a data gradient with one source and one destination");
        return done(ynet_conv2d_dgrad_relu(a->src[0], a->src_c[0], a->src_bs[0], a->mask, a->mask_bs, a->wp, a->dst[0], a->dst_c[0], a->dst_bs[0], a->relu_of, a->relu_of_bs, B, H, W, K, ws,
                                           nws, stream));
    default:
        return done(ynet_conv2d(a->src, a->src_c, a->src_bs, bmod, a->nsrc, a->mask, a->mask_bs, a->wp, a->bias, a->dst, a->dst_c, a->dst_bs, a->ndst, B, H, W, K, relu, ws, nws, stream));
    }
}

}  // extern "C"
