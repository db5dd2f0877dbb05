// graph_exec_x.hip -- graph X (the Xception autoencoder, misc_py/modified_Xception.py:194-654, inference) behind the native executor of
// the C ABI: emd_graph_create(variant 2) / emd_graph_workspace_bytes / emd_graph_run (last row: "the static D / X
// executors").  The layer table in the reference's variable-creation order (TF names under scope "pellet"), the folding of the moving-
// statistics norms (float64), weight packing, and the launch sequence of the library's own entry points over the caller's workspace --
// the same kernel choices as emdenoise.xception.XceptionEngine in its split-bf16 mode, single stream: bit-identical to it
// (tests/test_graph_exec_gpu.py).  Batch statistics (the separable convs' norms, :302-323) are taken over the batch handed to
// emd_graph_run, as the reference takes them per tower.
#include <cstdlib>
#include <cstring>

#include "graph_common.hpp"

namespace emd {
namespace gx {

namespace {

constexpr int F00 = 32, F01 = 64, F1 = 128, F2 = 256, F4 = 728, F5 = 1024, F6 = 1536, F7 = 2048;   // modified_Xception.py:38-70
constexpr int N_MIDDLE = 16, ASPP_F = 256, ASPP_OUT = 32;
constexpr int DEC[8] = {728, 728, 512, 384, 256, 192, 128, 64};
constexpr double BN_EPS_X = 1e-3;

enum XKind { XCONV, XSEP, XDECONV, XBN };

struct XDecl {
    XKind kind;
    int cin, cout, k = 1, stride = 1, rate = 1;
    std::string scope, bn;   // conv / separable / transposed conv scope; the moving-statistics norm that follows (conv, deconv, bn)
};

// tf.variable_scope default-name uniquifier inside scope 'pellet' (modified_Xception.py:794)
struct XScope {
    std::map<std::string, int> n;
    std::string operator()(const std::string& base) {
        const int k = n[base]++;
        return k == 0 ? "pellet/" + base : "pellet/" + base + "_" + std::to_string(k);
    }
};

// layers of architecture() in graph-construction order (mirror of emdenoise.xception.declare_layers)
std::vector<XDecl> declare_x() {
    XScope sc;
    std::vector<XDecl> L;
    auto conv = [&](int cin, int cout, int k = 1, int stride = 1, int rate = 1, const char* name = nullptr, bool bn = true) {
        XDecl d{XCONV, cin, cout, k, stride, rate};
        d.scope = name ? std::string("pellet/") + name : sc("conv2d");
        if (bn) d.bn = sc("BatchNorm");
        L.push_back(d);
    };
    auto sep = [&](int cin, int cout, int stride = 1) {
        XDecl d{XSEP, cin, cout, 3, stride, 1};
        d.scope = sc("SeparableConv2d");
        L.push_back(d);
    };
    auto deconv = [&](int c) {
        XDecl d{XDECONV, c, c, 3, 2, 1};
        d.scope = sc("conv2d_transpose");
        d.bn = sc("BatchNorm");
        L.push_back(d);
    };
    conv(1, F00, 3, 2);                                   // entry flow (:356-473)
    conv(F00, F01, 3);
    int c = F01;
    for (int f : {F1, F2, F4}) {
        conv(c, f, 1, 2);
        sep(c, f); sep(f, f); sep(f, f, 2);
        c = f;
    }
    for (int i = 0; i < N_MIDDLE; ++i) { sep(c, c); sep(c, c); sep(c, c); }   // :475-491, :629-630
    conv(c, F5, 1, 2);                                    // exit flow (:493-535)
    sep(c, F4); sep(F4, F5); sep(F5, F5, 2);
    sep(F5, F6); sep(F6, F6, 2); sep(F6, F7);
    c = F7;
    conv(c, ASPP_F, 1, 1, 1, "1x1");                      // ASPP (:231-299)
    conv(c, ASPP_F, 3, 1, 3, "lowRate");
    conv(c, ASPP_F, 3, 1, 6, "mediumRate");
    conv(c, ASPP_F, 3, 1, 9, "highRate");
    conv(c, ASPP_F, 1, 1, 1, "imageLevel", false);        // created, its output is discarded (:268-285)
    {
        XDecl d{XBN, ASPP_F, ASPP_F};
        d.bn = sc("BatchNorm");
        L.push_back(d);
    }
    conv(5 * ASPP_F, ASPP_OUT, 1);
    conv(ASPP_OUT, DEC[0], 1);                            // decoder (:538-621)
    c = DEC[0];
    for (int i = 0; i < 3; ++i) { conv(c, DEC[1], 3); c = DEC[1]; }
    const int nblocks[6] = {3, 3, 3, 2, 2, 2};
    for (int s = 0; s < 6; ++s) {
        deconv(c);
        for (int i = 0; i < nblocks[s]; ++i) { conv(c, DEC[2 + s], 3); c = DEC[2 + s]; }
    }
    conv(c, 1, 3);
    return L;
}

struct XParams {
    XDecl d;
    Packed pw, phase[4];
    float *dw = nullptr, *beta = nullptr, *one = nullptr, *zero = nullptr;     // separable conv
    float *bias = nullptr, *g = nullptr, *h = nullptr, *gs = nullptr, *hs = nullptr;   // conv: bias; norm as affine; conv + bias + norm folded
    float *w9 = nullptr;                                                       // the 1-channel entry conv: [9][Cout]
    float *scale = nullptr, *shift = nullptr;                                  // deconv (bias + norm folded), lone norm
    float *wfin = nullptr;                                                     // final conv: [9][Cin]
    float pre_bias = 0.f, scale_f = 1.f, shift_f = 0.f;
};

}  // namespace

struct XGraph {
    std::vector<XParams> P;
};

XGraph* x_create(const WeightMap& w, std::vector<void*>& allocs, std::string* err) {
    XGraph* x = new XGraph();
    bool ok = true;
    for (const XDecl& d : declare_x()) {
        XParams p;
        p.d = d;
        if (d.kind == XCONV) {
            const float *wt, *bias;
            ok = fetch(w, d.scope + "/kernel", (long)d.k * d.k * d.cin * d.cout, &wt, err) && fetch(w, d.scope + "/bias", d.cout, &bias, err);
            if (!ok) break;
            std::vector<double> g, h;
            if (!d.bn.empty() && !(ok = bn_affine(w, d.bn, d.cout, BN_EPS_X, &g, &h, err))) break;
            if (d.cout == 1) {                       // final conv_block(.., 1): [3][3][Cin][1] == [9][Cin]
                p.wfin = upload(allocs, wt, 9 * (size_t)d.cin);
                p.pre_bias = bias[0];
                p.scale_f = (float)g[0];
                p.shift_f = (float)h[0];
                ok = p.wfin != nullptr;
            } else {
                if (d.cin == 1) {                    // entry conv of the 1-channel image: [3][3][1][Cout] == [9][Cout]
                    p.w9 = upload(allocs, wt, 9 * (size_t)d.cout);
                    ok = p.w9 != nullptr;
                } else {
                    ok = pack(allocs, wt, d.k * d.k, d.cin, d.cout, 0, &p.pw);
                }
                std::vector<double> one(d.cout, 1.0), b(bias, bias + d.cout);
                p.bias = upload_f(allocs, b);
                p.one = upload_f(allocs, one);
                ok = ok && p.bias && p.one;
                if (ok && !d.bn.empty()) {
                    std::vector<double> hs(d.cout);
                    for (int c = 0; c < d.cout; ++c) hs[c] = (double)bias[c] * g[c] + h[c];
                    p.g = upload_f(allocs, g); p.h = upload_f(allocs, h);
                    p.gs = upload_f(allocs, g); p.hs = upload_f(allocs, hs);
                    ok = p.g && p.h && p.gs && p.hs;
                }
            }
        } else if (d.kind == XDECONV) {
            const float *wt, *bias;
            std::vector<double> g, h;
            ok = fetch(w, d.scope + "/kernel", 9L * d.cout * d.cin, &wt, err) && fetch(w, d.scope + "/bias", d.cout, &bias, err) &&
                 bn_affine(w, d.bn, d.cout, BN_EPS_X, &g, &h, err);
            if (!ok) break;
            for (int ph = 0; ph < 4 && ok; ++ph) {   // [3][3][Cout][Cin] -> per output phase [taps][Cout][Cin]
                int ky[4], kx[4];
                const int nt = emd_deconv_phase_taps(ph, ky, kx);
                std::vector<float> sub((size_t)nt * d.cout * d.cin);
                for (int t = 0; t < nt; ++t)
                    memcpy(sub.data() + (size_t)t * d.cout * d.cin, wt + (size_t)(ky[t] * 3 + kx[t]) * d.cout * d.cin, sizeof(float) * d.cout * d.cin);
                ok = pack(allocs, sub.data(), nt, d.cin, d.cout, 1, &p.phase[ph]);
            }
            std::vector<double> sh(d.cout);
            for (int c = 0; c < d.cout; ++c) sh[c] = (double)bias[c] * g[c] + h[c];
            p.scale = upload_f(allocs, g);
            p.shift = upload_f(allocs, sh);
            ok = ok && p.scale && p.shift;
        } else if (d.kind == XSEP) {
            const float *dw, *pw, *beta;
            ok = fetch(w, d.scope + "/depthwise_weights", 9L * d.cin, &dw, err) && fetch(w, d.scope + "/pointwise_weights", (long)d.cin * d.cout, &pw, err) &&
                 fetch(w, d.scope + "/BatchNorm/beta", d.cout, &beta, err);
            if (!ok) break;
            std::vector<double> one(d.cout, 1.0), zero(d.cout, 0.0);
            p.dw = upload(allocs, dw, 9 * (size_t)d.cin);
            p.beta = upload(allocs, beta, (size_t)d.cout);
            p.one = upload_f(allocs, one);
            p.zero = upload_f(allocs, zero);
            ok = p.dw && p.beta && p.one && p.zero && pack(allocs, pw, 1, d.cin, d.cout, 0, &p.pw);
        } else {
            std::vector<double> g, h;
            ok = bn_affine(w, d.bn, d.cout, BN_EPS_X, &g, &h, err);
            if (!ok) break;
            p.scale = upload_f(allocs, g);
            p.shift = upload_f(allocs, h);
            ok = p.scale && p.shift;
        }
        if (!ok) {
            if (err->empty()) *err = "emd_graph_create: device allocation or upload failed";
            break;
        }
        x->P.push_back(p);
    }
    if (!ok) {
        if (err->empty()) *err = "emd_graph_create: device allocation or upload failed";
        delete x;
        return nullptr;
    }
    return x;
}

void x_destroy(XGraph* x) { delete x; }

namespace {

// an activation of the launch sequence: fp32 NHWC (T4) or a split32 tensor
struct XT {
    T4 t;                  // fp32: buf / ld / C
    void* sp = nullptr;    // split32: base pointer, pitch ld in 4-byte units
    int ld = 0;
    bool split = false;
    // a separable block whose norm + relu are applied by the NEXT depthwise kernel while it loads (defer): y raw, (scale, shift)
    float *pre_s = nullptr, *pre_t = nullptr;
};

struct XRun {
    XGraph* g;
    Arena* ar;
    hipStream_t st;
    bool dry;
    int B;
    int rc = EMD_OK;
    size_t pos = 0;

    void* raw(size_t bytes) {
        void* p = ar->alloc(bytes);
        if (!p && rc == EMD_OK) rc = emd::fail(EMD_E_INVALID, "emd_graph_run: workspace too small");
        return p;
    }
    float* vec(int n) { return static_cast<float*>(raw((size_t)n * 4)); }
    T4 E(int H, int W, int C) {
        T4 t;
        t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
        t.buf = static_cast<float*>(raw((size_t)B * H * W * C * 4));
        return t;
    }
    XT F(int H, int W, int C) {
        XT x;
        x.t = E(H, W, C);
        return x;
    }
    XT SP(int H, int W, int C) {
        XT x;
        x.split = true;
        x.ld = emd_split32_ld(C);
        x.t.B = B; x.t.H = H; x.t.W = W; x.t.C = C;
        x.sp = raw((size_t)B * H * W * x.ld * 4);
        return x;
    }
    void drop(XT& x) {
        ar->release(x.split ? x.sp : static_cast<void*>(x.t.buf));
        ar->release(x.pre_s);
        ar->release(x.pre_t);
        x.sp = nullptr; x.t.buf = nullptr; x.pre_s = x.pre_t = nullptr;
    }
    void call(int code) {
        if (code != EMD_OK && rc == EMD_OK) rc = code;
    }
    bool live() const { return !dry && rc == EMD_OK; }
    const XParams& next() { return g->P[pos++]; }

    // this conv / deconv layer runs on the LDS-DMA split32 kernels (xception.split_ok)
    static bool split_ok(const XDecl& d, long npix_in) {
        if (!(d.kind == XDECONV || (d.kind == XCONV && d.k == 3))) return false;
        const long m = (d.kind == XDECONV || d.stride == 1) ? npix_in : npix_in / (d.stride * d.stride);
        const int bn = d.cout <= 64 ? 64 : 128;
        return d.cout >= 32 && d.cin >= 32 && ((m + 255) / 256) * ((d.cout + bn - 1) / bn) >= 192;
    }
    bool next_takes_split(long npix_out) const { return pos < g->P.size() && split_ok(g->P[pos].d, npix_out); }

    // a as a split32 tensor (converted when it is fp32); *tmp receives a buffer to release afterwards
    const void* as_split(const XT& a, int* ld, void** tmp) {
        *tmp = nullptr;
        if (a.split) { *ld = a.ld; return a.sp; }
        *ld = emd_split32_ld(a.t.C);
        const long npix = (long)B * a.t.H * a.t.W;
        *tmp = raw((size_t)npix * *ld * 4);
        if (live()) call(emd_to_split32_f32(a.t.ptr(), a.t.ld, *tmp, *ld, npix, a.t.C, st));
        return *tmp;
    }

    // tf.layers.conv2d (+ bias) -> batch_then_activ: one launch (xception.conv_bn_relu)
    XT conv_bn_relu(const XT& a, const float* img, int S, const T4* out_opt) {
        const XParams& p = next();
        const XDecl& d = p.d;
        if (d.cin == 1 && d.k == 3 && !out_opt && d.cout % 32 == 0) {   // the 1-channel entry conv: fp32 FMAs, split32 output
            const int Ho = (S + d.stride - 1) / d.stride;
            XT r = SP(Ho, Ho, d.cout);
            if (live()) call(emd_conv3x3_cin1_f32(img, p.w9, p.gs, p.hs, r.sp, r.ld, B, S, S, d.cout, d.stride, EMD_ACT_RELU, 1, st));
            return r;
        }
        const int H = a.t.H, W = a.t.W;
        const int Ho = (H + d.stride - 1) / d.stride, Wo = (W + d.stride - 1) / d.stride;
        if (d.k == 3 && d.stride == 1 && d.rate == 1 && !out_opt && d.cin % 32 == 0 && d.cout <= 256 && Ho % 8 == 0 && Wo % 32 == 0 && a.split) {
            XT r = F(Ho, Wo, d.cout);
            if (live())
                call(emd_conv3x3_split32_f32(a.sp, a.ld, p.pw.hi, p.pw.lo, p.gs, p.hs, nullptr, nullptr, nullptr, 0, r.t.ptr(), r.t.ld, B, H, W,
                                             d.cin, d.cout, 1, 1, EMD_ACT_RELU, 0, st));
            return r;
        }
        XT r;
        r.t = out_opt ? *out_opt : E(Ho, Wo, d.cout);
        if (a.split && rc == EMD_OK) rc = emd::fail(EMD_E_UNSUPPORTED, "emd_graph_run: graph X: a split32 tensor reached an fp32 convolution");
        if (live()) {
            if (d.k == 1)
                call(emd_conv1x1_f32(a.t.ptr(), a.t.ld, p.pw.hi, p.pw.lo, p.gs, p.hs, nullptr, nullptr, nullptr, 0, r.t.ptr(), r.t.ld, B, H, W, d.cin,
                                     d.cout, d.stride, EMD_ACT_RELU, EMD_PREC_BF16X3, st));
            else
                call(emd_conv3x3_f32(a.t.ptr(), a.t.ld, p.pw.hi, p.pw.lo, p.gs, p.hs, nullptr, nullptr, nullptr, 0, r.t.ptr(), r.t.ld, B, H, W, d.cin,
                                     d.cout, d.stride, d.rate, EMD_ACT_RELU, EMD_PREC_BF16X3, st));
        }
        return r;
    }

    // conv3x3 + bias -> relu -> BN -> relu (:215-229): two-stage epilogue; split32 in / out where the GEMM is matrix-core bound
    XT conv_block(const XT& a) {
        const XParams& p = next();
        const XDecl& d = p.d;
        const int H = a.t.H, W = a.t.W;
        const long npix = (long)B * H * W;
        if (split_ok(d, npix)) {
            XT r = next_takes_split(npix) ? SP(H, W, d.cout) : F(H, W, d.cout);
            int ldx;
            void* tmp;
            const void* xs = as_split(a, &ldx, &tmp);
            if (live())
                call(emd_conv3x3_split32_f32(xs, ldx, p.pw.hi, p.pw.lo, p.one, p.bias, p.g, p.h, nullptr, 0, r.split ? r.sp : static_cast<void*>(r.t.ptr()),
                                             r.split ? r.ld : r.t.ld, B, H, W, d.cin, d.cout, 1, 1, EMD_ACT_RELU, r.split ? 1 : 0, st));
            ar->release(tmp);
            return r;
        }
        XT r = F(H, W, d.cout);
        if (a.split && rc == EMD_OK) rc = emd::fail(EMD_E_UNSUPPORTED, "emd_graph_run: graph X: a split32 tensor reached an fp32 convolution");
        if (live())
            call(emd_conv3x3_f32(a.t.ptr(), a.t.ld, p.pw.hi, p.pw.lo, p.one, p.bias, p.g, p.h, nullptr, 0, r.t.ptr(), r.t.ld, B, H, W, d.cin, d.cout, 1, 1,
                                 EMD_ACT_RELU, EMD_PREC_BF16X3, st));
        return r;
    }

    XT deconv(const XT& a) {
        const XParams& p = next();
        const XDecl& d = p.d;
        const int H = a.t.H, W = a.t.W;
        const long npix = (long)B * H * W;
        const uint16_t* hi[4] = {p.phase[0].hi, p.phase[1].hi, p.phase[2].hi, p.phase[3].hi};
        const uint16_t* lo[4] = {p.phase[0].lo, p.phase[1].lo, p.phase[2].lo, p.phase[3].lo};
        if (split_ok(d, npix)) {
            XT r = next_takes_split(4 * npix) ? SP(2 * H, 2 * W, d.cout) : F(2 * H, 2 * W, d.cout);
            int ldx;
            void* tmp;
            const void* xs = as_split(a, &ldx, &tmp);
            if (live())
                call(emd_deconv3x3s2_fused_split32_f32(xs, ldx, hi, lo, p.scale, p.shift, r.split ? r.sp : static_cast<void*>(r.t.ptr()),
                                                       r.split ? r.ld : r.t.ld, B, H, W, d.cin, d.cout, EMD_ACT_RELU, r.split ? 1 : 0, st));
            ar->release(tmp);
            return r;
        }
        XT r = F(2 * H, 2 * W, d.cout);
        if (a.split && rc == EMD_OK) rc = emd::fail(EMD_E_UNSUPPORTED, "emd_graph_run: graph X: a split32 tensor reached an fp32 transposed conv");
        if (live())
            call(emd_deconv3x3s2_f32(a.t.ptr(), a.t.ld, hi, lo, p.scale, p.shift, r.t.ptr(), r.t.ld, B, H, W, d.cin, d.cout, EMD_ACT_RELU, EMD_PREC_BF16X3, st));
        return r;
    }

    // depthwise -> pointwise (raw) -> batch-statistics BN (beta only) -> relu [+ res] (:302-323); defer: the norm + relu travel with the
    // raw output and the next depthwise kernel applies them while loading
    XT sep(const XT& a, const T4* res, bool defer) {
        const XParams& p = next();
        const XDecl& d = p.d;
        const int H = a.t.H, W = a.t.W;
        const int Ho = (H + d.stride - 1) / d.stride, Wo = (W + d.stride - 1) / d.stride;
        const long M = (long)B * Ho * Wo;
        XT y = F(Ho, Wo, d.cout);
        float *mean = vec(d.cout), *var = vec(d.cout), *scale = vec(d.cout), *shift = vec(d.cout);
        if (emd_conv1x1_split32_supported(M, d.cin, d.cout)) {
            const int ldd = emd_split32_ld(d.cin);
            void* dsp = raw((size_t)M * ldd * 4);
            void* ws = raw(emd_conv1x1_split32_stats_workspace_bytes(M, d.cout));
            if (live()) {
                if (a.pre_s)
                    call(emd_dw3x3_pre_split32_f32(a.t.ptr(), a.t.ld, a.pre_s, a.pre_t, p.dw, dsp, ldd, B, H, W, d.cin, d.stride, 1, st));
                else
                    call(emd_dw3x3_split32_f32(a.t.ptr(), a.t.ld, p.dw, dsp, ldd, B, H, W, d.cin, d.stride, 1, st));
                call(emd_conv1x1_split32_stats_fold_f32(dsp, ldd, p.pw.hi, p.pw.lo, p.one, p.zero, y.t.ptr(), y.t.ld, M, d.cin, d.cout, EMD_ACT_NONE, mean,
                                                        var, ws, nullptr, p.beta, (float)BN_EPS_X, scale, shift, st));
            }
            ar->release(dsp);
            ar->release(ws);
        } else {
            T4 tmp = E(Ho, Wo, d.cin);
            void* ws = raw(emd_conv_stats_workspace_bytes(M, d.cout));
            if (live()) {
                if (a.pre_s)
                    call(emd_dw3x3_pre_f32(a.t.ptr(), a.t.ld, a.pre_s, a.pre_t, p.dw, tmp.ptr(), tmp.ld, B, H, W, d.cin, d.stride, 1, st));
                else
                    call(emd_dw3x3_f32(a.t.ptr(), a.t.ld, p.dw, tmp.ptr(), tmp.ld, B, H, W, d.cin, d.stride, 1, st));
                // the statistics from the fp32 GEMM's epilogue (round 4; xception.py's sep does the same: identical bits)
                call(emd_conv1x1_stats_f32(tmp.ptr(), tmp.ld, p.pw.hi, p.pw.lo, p.one, p.zero, y.t.ptr(), y.t.ld, B, Ho, Wo, d.cin, d.cout, 1,
                                           EMD_PREC_BF16X3, 0, mean, var, ws, st));
                call(emd_bn_fold_f32(mean, var, nullptr, p.beta, (float)BN_EPS_X, scale, shift, d.cout, st));
            }
            ar->release(tmp.buf);
            ar->release(ws);
        }
        ar->release(mean);
        ar->release(var);
        if (defer && !res) {
            y.pre_s = scale;
            y.pre_t = shift;
            return y;
        }
        if (live())
            call(emd_affine_act_f32(y.t.ptr(), y.t.ld, scale, shift, res ? res->ptr() : nullptr, res ? res->ld : 0, y.t.ptr(), y.t.ld, M, d.cout,
                                    EMD_ACT_RELU, st));
        ar->release(scale);
        ar->release(shift);
        return y;
    }

    void forward(const float* in, float* out, int S) {
        XT none;
        XT e1 = conv_bn_relu(none, in, S, nullptr);            // entry flow: 1 -> 32 (stride 2), 32 -> 64
        XT e = conv_bn_relu(e1, nullptr, 0, nullptr);
        drop(e1);
        for (int blk = 0; blk < 3; ++blk) {
            XT res = conv_bn_relu(e, nullptr, 0, nullptr);
            XT m1 = sep(e, nullptr, true);
            XT m2 = sep(m1, nullptr, true);
            drop(m1);
            drop(e);
            e = sep(m2, &res.t, false);
            drop(m2);
            drop(res);
        }
        for (int i = 0; i < N_MIDDLE; ++i) {
            XT m1 = sep(e, nullptr, true);
            XT m2 = sep(m1, nullptr, true);
            drop(m1);
            XT y = sep(m2, &e.t, false);
            drop(m2);
            drop(e);
            e = y;
        }
        XT m;
        {   // exit flow
            XT res = conv_bn_relu(e, nullptr, 0, nullptr);
            XT m1 = sep(e, nullptr, true);
            drop(e);
            XT m2 = sep(m1, nullptr, true);
            drop(m1);
            m = sep(m2, &res.t, false);
            drop(m2);
            drop(res);
            XT m3 = sep(m, nullptr, true);
            drop(m);
            XT m4 = sep(m3, nullptr, true);
            drop(m3);
            m = sep(m4, nullptr, false);
            drop(m4);
        }
        // ASPP: the branches write into their slices of the 1280-channel concat
        XT cat = F(m.t.H, m.t.W, 5 * ASPP_F);
        T4 large{};
        for (int b = 0; b < 4; ++b) {
            T4 sl = cat.t.slice(b * ASPP_F, ASPP_F);
            conv_bn_relu(m, nullptr, 0, &sl);
            if (b == 3) large = sl;
        }
        drop(m);
        next();                                                // 'imageLevel': variables exist, output discarded (:268-285)
        {
            const XParams& p = next();                         // pooling = batch_then_activ(conv3x3_rateLarge)
            T4 sl = cat.t.slice(4 * ASPP_F, ASPP_F);
            if (live())
                call(emd_affine_act_f32(large.ptr(), large.ld, p.scale, p.shift, nullptr, 0, sl.ptr(), sl.ld, (long)B * sl.H * sl.W, ASPP_F, EMD_ACT_RELU, st));
        }
        XT d1 = conv_bn_relu(cat, nullptr, 0, nullptr);
        drop(cat);
        XT d = conv_bn_relu(d1, nullptr, 0, nullptr);          // decoder
        drop(d1);
        for (int i = 0; i < 3; ++i) {
            XT n = conv_block(d);
            drop(d);
            d = n;
        }
        const int nblocks[6] = {3, 3, 3, 2, 2, 2};
        for (int s = 0; s < 6; ++s) {
            XT n = deconv(d);
            drop(d);
            d = n;
            for (int i = 0; i < nblocks[s]; ++i) {
                XT q = conv_block(d);
                drop(d);
                d = q;
            }
        }
        const XParams& p = next();                             // conv_block(decoding, 1) then clip to [0,1] (:639-641)
        if (d.split && rc == EMD_OK) rc = emd::fail(EMD_E_UNSUPPORTED, "emd_graph_run: graph X: the final conv takes an fp32 tensor");
        if (live())
            call(emd_conv3x3_cout1_f32(d.t.ptr(), d.t.ld, p.wfin, p.scale_f, p.shift_f, out, B, d.t.H, d.t.W, p.d.cin, 2, p.pre_bias, 1, st));
        drop(d);
        if (pos != g->P.size() && rc == EMD_OK) rc = emd::fail(EMD_E_INVALID, "emd_graph_run: graph X: layer table and launch sequence disagree");
    }
};

}  // namespace

int x_forward(XGraph* x, Arena* ar, hipStream_t st, bool dry, const float* in, float* out, int B, int S) {
    XRun r{x, ar, st, dry, B};
    r.forward(in, out, S);
    return r.rc;
}

}  // namespace gx
}  // namespace emd
