// graph_exec_g.hip -- graph G's GENERATOR (the in-filling network of misc_py/gan-infilling-100.py:133-374, inference: moving-statistics
// batch norms, epsilon 0.01) behind the native executor of the C ABI: emd_graph_create(variant 3) / _workspace_bytes / _run.  The layer
// table in the reference's variable-creation order (TF names under "GAN/Gen" and "GAN/Gen/reg"), both batch norms of every separable conv
// folded into one affine (float64), weight packing, and the launch sequence of the library's own entry points -- the kernel choices of
// emdenoise.gan.GeneratorEngine in its split-bf16 mode, single stream: bit-identical to it (tests/test_graph_exec_gpu.py).
#include <cstdlib>
#include <cstring>

#include "graph_common.hpp"

namespace emd {
namespace gx {

namespace {

constexpr int GF0 = 32, GF1 = 64, GF2 = 64, GF3 = 32;            // gan-infilling-100.py: gen_features0..3
constexpr int NIN1 = 128, NIN2 = 256, NIN3 = 768, NOUT1 = 256, NOUT2 = 128, NOUT3 = 64;
constexpr int N_GLOBAL = 8, N_LOCAL = 3;
constexpr double BN_EPS_GEN = 0.01;                                // :167
constexpr float IN_EPS = 1e-3f;                                    // :146

struct GDecl {
    std::string key, scope, outer_bn;
    int cin, cout, k = 3, stride = 1;
    bool reflect = true;
};

// tf.variable_scope name uniquifier with a scope stack (emdenoise.gan._Scope)
struct GScope {
    std::vector<std::string> stack{"GAN/Gen"};
    std::map<std::string, int> counts;
    std::string unique(const std::string& base) {
        std::string parent = stack[0];
        for (size_t i = 1; i < stack.size(); ++i) parent += "/" + stack[i];
        const int k = counts[parent + "|" + base]++;
        return k == 0 ? parent + "/" + base : parent + "/" + base + "_" + std::to_string(k);
    }
};

struct GTable {
    std::vector<GDecl> L;
    std::string conv_scope, in_shift, in_scale;
};

// the generator's parameterised layers in creation order (:341-372; mirror of emdenoise.gan.declare_layers)
GTable declare_g() {
    GScope sc;
    GTable T;
    auto sep = [&](const std::string& key, int cin, int cout, int k = 3, int stride = 1, bool reflect = true) {
        GDecl d;
        d.key = key; d.cin = cin; d.cout = cout; d.k = k; d.stride = stride; d.reflect = reflect;
        d.scope = sc.unique("SeparableConv2d");
        d.outer_bn = sc.unique("BatchNorm");
        T.L.push_back(d);
    };
    auto middle = [&](const std::string& prefix, int f) {
        for (int j = 0; j < 3; ++j) sep(prefix + "_" + std::to_string(j), f, f);
    };
    sep("enc0", 1, GF0, 7);
    sep("enc1", GF0, GF1, 3, 2);
    sc.stack.push_back("reg");
    sep("nin_down0", GF1, NIN1, 3, 2);
    sep("nin_down1", NIN1, NIN2, 3, 2);
    sep("nin_down2", NIN2, NIN3, 3, 2);
    for (int i = 0; i < N_GLOBAL; ++i) middle("nin_mid" + std::to_string(i), NIN3);
    sep("nin_up0", NIN3, NOUT1, 3, 1, false);   // deconv_block: SAME (pad_size swallowed)
    sep("nin_up1", NOUT1, NOUT2, 3, 1, false);
    sep("nin_up2", NOUT2, NOUT3, 3, 1, false);
    for (int i = 0; i < N_LOCAL; ++i) middle("local" + std::to_string(i), GF2);
    sep("up", GF2, GF3, 3, 1, false);
    sep("last_sep", GF3, GF3);
    sc.stack.pop_back();
    T.conv_scope = sc.unique("Conv");
    T.in_shift = sc.unique("Variable");
    T.in_scale = sc.unique("Variable");
    return T;
}

struct GParams {
    GDecl d;
    float *dw = nullptr, *scale = nullptr, *shift = nullptr;   // [9][Cin]; both norms folded
    Packed pw;
    float *w49 = nullptr, *a = nullptr;                          // enc0: 7x7 taps; pointwise weights x folded scale
};

}  // namespace

struct GGraph {
    std::map<std::string, GParams> P;
    float *unit4 = nullptr, *zero4 = nullptr, *w_last = nullptr;
    float b_last = 0.f;
};

GGraph* g_create(const WeightMap& w, std::vector<void*>& allocs, std::string* err) {
    GGraph* g = new GGraph();
    const GTable T = declare_g();
    bool ok = true;
    for (const GDecl& d : T.L) {
        GParams p;
        p.d = d;
        const float *dw, *pw;
        ok = fetch(w, d.scope + "/depthwise_weights", (long)d.k * d.k * d.cin, &dw, err) && fetch(w, d.scope + "/pointwise_weights", (long)d.cin * d.cout, &pw, err);
        if (!ok) break;
        std::vector<double> s(d.cout, 1.0), t(d.cout, 0.0);
        for (const std::string& scope : {d.scope + "/BatchNorm", d.outer_bn}) {
            std::vector<double> gs, hs;
            if (!(ok = bn_affine(w, scope, d.cout, BN_EPS_GEN, &gs, &hs, err))) break;
            for (int c = 0; c < d.cout; ++c) {
                s[c] *= gs[c];
                t[c] = t[c] * gs[c] + hs[c];
            }
        }
        if (!ok) break;
        if (d.cin == 1) {
            std::vector<double> a(d.cout);
            for (int c = 0; c < d.cout; ++c) a[c] = (double)pw[c] * s[c];
            p.w49 = upload(allocs, dw, 49);
            p.a = upload_f(allocs, a);
            p.shift = upload_f(allocs, t);
            ok = p.w49 && p.a && p.shift;
        } else {
            p.dw = upload(allocs, dw, 9 * (size_t)d.cin);
            p.scale = upload_f(allocs, s);
            p.shift = upload_f(allocs, t);
            ok = p.dw && p.scale && p.shift && pack(allocs, pw, 1, d.cin, d.cout, 0, &p.pw);
        }
        if (!ok) break;
        g->P[d.key] = p;
    }
    if (ok) {
        const float *wl, *bl, *sh, *sc;
        ok = fetch(w, T.conv_scope + "/weights", 9L * GF3, &wl, err) && fetch(w, T.conv_scope + "/biases", 1, &bl, err) &&
             fetch(w, T.in_shift, 1, &sh, err) && fetch(w, T.in_scale, 1, &sc, err);
        if (ok && (sh[0] != 0.f || sc[0] != 1.f)) {
            *err = "emd_graph_create: the instance norm's shift / scale variables are frozen at 0 / 1 in the reference (gan-infilling-100.py:144-145)";
            ok = false;
        }
        if (ok) {
            const float u4[4] = {1.f, 0.f, 0.f, 0.f}, z4[4] = {0.f, 0.f, 0.f, 0.f};
            g->w_last = upload(allocs, wl, 9 * (size_t)GF3);   // [3][3][Cin][1] == [9][Cin]
            g->b_last = bl[0];
            g->unit4 = upload(allocs, u4, 4);
            g->zero4 = upload(allocs, z4, 4);
            ok = g->w_last && g->unit4 && g->zero4;
        }
    }
    if (!ok) {
        if (err->empty()) *err = "emd_graph_create: device allocation or upload failed";
        delete g;
        return nullptr;
    }
    return g;
}

void g_destroy(GGraph* g) { delete g; }

namespace {

struct GRun {
    GGraph* g;
    Arena* ar;
    hipStream_t st;
    bool dry;
    int B;
    int rc = EMD_OK;

    void* raw(size_t bytes) {
        void* p = ar->alloc(bytes);
        if (!p && rc == EMD_OK) rc = emd::fail(EMD_E_INVALID, "emd_graph_run: workspace too small");
        return p;
    }
    T4 E(int H, int W, int C) {
        T4 t;
        t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
        t.buf = static_cast<float*>(raw((size_t)B * H * W * C * 4));
        return t;
    }
    void drop(T4& t) {
        ar->release(t.buf);
        t.buf = nullptr;
    }
    void call(int code) {
        if (code != EMD_OK && rc == EMD_OK) rc = code;
    }
    bool live() const { return !dry && rc == EMD_OK; }

    // strided_conv_block / deconv_block (gan-infilling-100.py:205-243): the route choices of GeneratorEngine._sep
    T4 sep(const std::string& key, const T4& x, const T4* res) {
        const GParams& p = g->P[key];
        const GDecl& d = p.d;
        const int Ho = (x.H - 1) / d.stride + 1, Wo = (x.W - 1) / d.stride + 1;
        const long M = (long)B * Ho * Wo;
        T4 out = E(Ho, Wo, d.cout);
        const float* rp = res ? res->ptr() : nullptr;
        const int rl = res ? res->ld : 0;
        if (d.stride == 1 && emd_sep3x3_fused_supported(x.H, x.W, d.cin, d.cout, 1, 1) && x.H >= 2 && x.W >= 2) {
            if (live()) {
                if (d.reflect)
                    call(emd_sep3x3_fused_reflect_f32(x.ptr(), x.ld, p.dw, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, rp, rl, out.ptr(), out.ld, B,
                                                      x.H, x.W, d.cin, d.cout, EMD_ACT_LEAKY, EMD_PREC_BF16X3, st));
                else
                    call(emd_sep3x3_fused_f32(x.ptr(), x.ld, p.dw, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, rp, rl, out.ptr(), out.ld, B, x.H,
                                              x.W, d.cin, d.cout, EMD_ACT_LEAKY, EMD_PREC_BF16X3, st));
            }
            return out;
        }
        if (d.stride == 2 && d.reflect && x.H % 2 == 0 && x.W % 2 == 0 && emd_sep3x3_fused_supported(x.H, x.W, d.cin, d.cout, 2, 1)) {
            if (live())
                call(emd_sep3x3_fused_s2_reflect_f32(x.ptr(), x.ld, p.dw, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, rp, rl, out.ptr(), out.ld, B,
                                                     x.H, x.W, d.cin, d.cout, EMD_ACT_LEAKY, st));
            return out;
        }
        if (emd_conv1x1_split32_supported(M, d.cin, d.cout)) {
            const int ldd = emd_split32_ld(d.cin);
            void* dsp = raw((size_t)M * ldd * 4);
            if (live()) {
                if (d.reflect) call(emd_dw3x3_reflect_split32_f32(x.ptr(), x.ld, p.dw, dsp, ldd, B, x.H, x.W, d.cin, d.stride, st));
                else call(emd_dw3x3_split32_f32(x.ptr(), x.ld, p.dw, dsp, ldd, B, x.H, x.W, d.cin, d.stride, 1, st));
                call(emd_conv1x1_split32_f32(dsp, ldd, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, rp, rl, out.ptr(), out.ld, M, d.cin, d.cout,
                                             EMD_ACT_LEAKY, st));
            }
            ar->release(dsp);
            return out;
        }
        T4 tmp = E(Ho, Wo, d.cin);
        if (live()) {
            if (d.reflect) call(emd_dw3x3_reflect_f32(x.ptr(), x.ld, p.dw, tmp.ptr(), tmp.ld, B, x.H, x.W, d.cin, d.stride, st));
            else call(emd_dw3x3_f32(x.ptr(), x.ld, p.dw, tmp.ptr(), tmp.ld, B, x.H, x.W, d.cin, d.stride, 1, st));
            call(emd_conv1x1_f32(tmp.ptr(), tmp.ld, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, rp, rl, out.ptr(), out.ld, B, Ho, Wo, d.cin, d.cout,
                                 1, EMD_ACT_LEAKY, EMD_PREC_BF16X3, st));
        }
        drop(tmp);
        return out;
    }
    T4 middle(const std::string& prefix, T4& x) {   // consumes x
        T4 t0 = sep(prefix + "_0", x, nullptr);
        T4 t1 = sep(prefix + "_1", t0, nullptr);
        drop(t0);
        T4 y = sep(prefix + "_2", t1, &x);
        drop(t1);
        drop(x);
        return y;
    }
    T4 up(const std::string& key, T4& x, int size, const T4* res) {   // consumes x
        T4 u = E(size, size, x.C);
        if (live()) call(emd_resize_bilinear_f32(x.ptr(), x.ld, u.ptr(), u.ld, B, x.H, x.W, size, size, x.C, st));
        drop(x);
        T4 y = sep(key, u, res);
        drop(u);
        return y;
    }

    void forward(const float* in, float* out, int S) {
        const GParams& p0 = g->P["enc0"];
        const GParams& p1 = g->P["enc1"];
        const int So = (S - 1) / p1.d.stride + 1;
        T4 enc;
        if (p1.d.reflect && p1.d.stride == 2 && !emd_conv1x1_split32_supported((long)B * So * So, p1.d.cin, p1.d.cout)) {
            // enc0 = leaky(d7 * a + t) is an outer product of the 7x7 stencil of the 1-channel image: enc1's depthwise conv rebuilds it
            T4 d4 = E(S, S, 4), dd = E(So, So, p1.d.cin);
            enc = E(So, So, p1.d.cout);
            if (live()) {
                call(emd_cin1_k7_reflect_f32(in, p0.w49, g->unit4, g->zero4, d4.ptr(), d4.ld, B, S, S, 4, 0, st));
                call(emd_dw3x3_reflect_gen_f32(d4.ptr(), d4.ld, p0.a, p0.shift, 1, p1.dw, dd.ptr(), dd.ld, B, S, S, p1.d.cin, p1.d.stride, st));
                call(emd_conv1x1_f32(dd.ptr(), dd.ld, p1.pw.hi, p1.pw.lo, p1.scale, p1.shift, nullptr, nullptr, nullptr, 0, enc.ptr(), enc.ld, B, So, So,
                                     p1.d.cin, p1.d.cout, 1, EMD_ACT_LEAKY, EMD_PREC_BF16X3, st));
            }
            drop(d4);
            drop(dd);
        } else {
            T4 e0 = E(S, S, GF0);
            if (live()) call(emd_cin1_k7_reflect_f32(in, p0.w49, p0.a, p0.shift, e0.ptr(), e0.ld, B, S, S, GF0, 1, st));
            enc = sep("enc1", e0, nullptr);
            drop(e0);
        }
        T4 n0 = sep("nin_down0", enc, nullptr);
        T4 n1 = sep("nin_down1", n0, nullptr);
        drop(n0);
        T4 n = sep("nin_down2", n1, nullptr);
        drop(n1);
        for (int i = 0; i < N_GLOBAL; ++i) n = middle("nin_mid" + std::to_string(i), n);
        n = up("nin_up0", n, S / 8, nullptr);
        n = up("nin_up1", n, S / 4, nullptr);
        T4 e = up("nin_up2", n, S / 2, &enc);                  // enc += network_in_network(enc)  (:355)
        drop(enc);
        for (int i = 0; i < N_LOCAL; ++i) e = middle("local" + std::to_string(i), e);
        e = up("up", e, S, nullptr);
        T4 last = sep("last_sep", e, nullptr);
        drop(e);
        // tf.pad(REFLECT, 1) + 3x3 VALID conv to one channel + bias (:362-369), instance norm + tanh (:140-148, :372)
        const long npix = (long)S * S;
        float* rawimg = static_cast<float*>(raw((size_t)B * npix * 4));
        float *mean = static_cast<float*>(raw((size_t)B * 4)), *var = static_cast<float*>(raw((size_t)B * 4));
        size_t wsb = emd_bn_stats_workspace_bytes(npix, 1);
        if (wsb < 8) wsb = 8;
        void* ws = raw((size_t)B * wsb);
        if (live()) {
            call(emd_conv3x3_cout1_reflect_f32(last.ptr(), last.ld, g->w_last, g->b_last, rawimg, B, S, S, GF3, st));
            call(emd_bn_stats_images_f32(rawimg, 1, B, npix, 1, mean, var, ws, st));
            call(emd_instnorm_tanh_f32(rawimg, mean, var, out, B, npix, IN_EPS, st));
        }
        drop(last);
        ar->release(rawimg); ar->release(mean); ar->release(var); ar->release(ws);
    }
};

}  // namespace

int g_forward(GGraph* g, Arena* ar, hipStream_t st, bool dry, const float* in, float* out, int B, int S) {
    GRun r{g, ar, st, dry, B};
    r.forward(in, out, S);
    return r.rc;
}

}  // namespace gx
}  // namespace emd
